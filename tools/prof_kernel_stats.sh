cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01b -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --batched-songs 0 --profile-steps 1 > $GRAFT_REPO_ROOT/gpurun_out/prof_r01b.log 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/prof_r01b -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/prof_r01b_kernel_stats.csv; tail -2 gpurun_out/prof_r01b.log | cut -c1-300; find gpurun_out/prof_r01b -name "*kernel_trace.csv" -delete; ls gpurun_out/prof_r01b/*
