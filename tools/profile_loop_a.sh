#!/bin/bash
# usage (GPU box, repo root): bash tools/profile_loop_a.sh [trba|crnn|svtr] [tag]
# rocprofv3 kernel trace of tools/bench_loop_a.py (loop A: training the newest expert), summary -> gpurun_out/<tag>/summary.md
model=${1:-trba}; tag=${2:-r1v}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$tag/prof -o trace -- python3 $R/tools/bench_loop_a.py $model 256 3 > $R/gpurun_out/$tag/prof.log 2>&1
cd $R
python3 tools/rocprof_summary.py gpurun_out/$tag/prof/trace_results.db > gpurun_out/$tag/summary.md 2>&1
python3 tools/rocprof_step.py gpurun_out/$tag/prof/trace_results.db ${MARKER:-adam_kernel} ${STEP_LIST:+--list} > gpurun_out/$tag/step.md 2>&1
rm -rf gpurun_out/$tag/prof
tail -1 gpurun_out/$tag/prof.log
