#!/bin/bash
# GPU call 18: diagnostic counter passes at one song and at 8 songs per handle
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
BATCH=1 bash tools/diag/pmc_diag.sh r03s
BATCH=8 bash tools/diag/pmc_diag.sh r03s
