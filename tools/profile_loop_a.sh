R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r1v
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r1v/prof -o trace -- python3 $R/tools/bench_loop_a.py trba 256 3 > $R/gpurun_out/r1v/prof.log 2>&1
cd $R
python3 tools/rocprof_summary.py gpurun_out/r1v/prof/trace_results.db 5 > gpurun_out/r1v/summary.md 2>&1
rm -rf gpurun_out/r1v/prof
tail -1 gpurun_out/r1v/prof.log
