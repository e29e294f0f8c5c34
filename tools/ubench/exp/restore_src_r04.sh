#!/bin/bash
# tools/ubench/exp/src_r04/ = round 4's csrc/ (+ its include/msd_amd.h, ABI 4) exactly as it was, with every experiment and
# ablation switch still inside -- what the micro-benchmarks under tools/ubench/ and the experiments build
# (`python music-spectrogram-diffusion_amd/build_native.py --experiments`) compile against.  Rounds 5 kept a 5 669-line
# copy of it in the tree; since round 6 it is RECONSTRUCTED from history: commit a65d5f7 (tag `r04-sources`) + the four
# include-path edits of src_r04_includes.patch.  Run from anywhere inside the repository:
#   bash tools/ubench/exp/restore_src_r04.sh
set -e
ROOT=$(git -C "$(dirname "$0")" rev-parse --show-toplevel)
DST=$ROOT/tools/ubench/exp/src_r04
REV=a65d5f7   # = tag r04-sources: the last commit whose csrc/ still carried the experiments
rm -rf "$DST"; mkdir -p "$DST"
for f in attention.h common.h elementwise.h gemm_f32.h gemm_h16.h msd_api.hip; do
  git -C "$ROOT" show $REV:music-spectrogram-diffusion_amd/csrc/$f > "$DST/$f"
done
git -C "$ROOT" show $REV:include/msd_amd.h > "$DST/msd_amd.h"
(cd "$DST" && patch -p1 --quiet < "$ROOT/tools/ubench/exp/src_r04_includes.patch")
echo "restored $DST from $REV ($(cat "$DST"/* | wc -l) lines)"
