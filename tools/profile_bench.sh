#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile_bench.sh <tag> [bench args...]
# rocprofv3 kernel trace of bench.py (1 warm-up + 2 timed router steps; with the loop-B pipeline one more expert forward is prefetched, so
# the whole-trace summary.md holds 3 router steps + 4 expert forwards INCLUDING the warm-up's one-off weight packing: use step.md --
# the launches between the last two adam_kernel launches, one steady-state step -- for per-step figures)
tag=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$tag/prof -o trace -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-isolated-pass --no-power-probe "$@" > $R/gpurun_out/$tag/prof.log 2>&1
cd $R
python3 tools/rocprof_summary.py gpurun_out/$tag/prof/trace_results.db > gpurun_out/$tag/summary.md 2>&1
python3 tools/rocprof_timeline.py gpurun_out/$tag/prof/trace_results.db > gpurun_out/$tag/timeline.txt 2>&1
python3 tools/rocprof_copies.py gpurun_out/$tag/prof/trace_results.db > gpurun_out/$tag/copies.txt 2>&1
python3 tools/rocprof_step.py gpurun_out/$tag/prof/trace_results.db ${MARKER:-adam_kernel} ${STEP_LIST:+--list} > gpurun_out/$tag/step.md 2>&1
rm -rf gpurun_out/$tag/prof
tail -1 gpurun_out/$tag/prof.log | cut -c1-300
