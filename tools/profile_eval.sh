#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r03n_eval
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03n_eval/prof -o trace -- python3 $R/tools/bench_eval.py > $R/gpurun_out/r03n_eval/prof.log 2>&1
cd $R
python3 tools/rocprof_summary.py gpurun_out/r03n_eval/prof/trace_results.db 1 > gpurun_out/r03n_eval/summary.md 2>&1
rm -rf gpurun_out/r03n_eval/prof
