#!/bin/bash
# usage (GPU box, repo root): bash tools/refresh_evidence.sh <tag>   -- everything profiles/<round>* is regenerated from, in one call:
# the GPU suite, the default bench line, every other bench line, the kernel traces of the five workloads, the two PMC passes
tag=${1:-evidence}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$tag
cd $R
[ -z "$SKIP_TESTS" ] && { python -m pytest tests -m gpu -x -q > gpurun_out/$tag/pytest_gpu.log 2>&1; tail -2 gpurun_out/$tag/pytest_gpu.log; }
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$tag/smoke.log 2>&1; tail -1 gpurun_out/$tag/smoke.log
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_default_line.json; cp bench_detail.json gpurun_out/${tag}_bench_default_detail.json
python -c "import json; d=json.load(open('gpurun_out/${tag}_bench_default_line.json')); print('default', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('isolated', {}).get('frac'), d['extra'])"
bash tools/bench_all_lines.sh gpurun_out/${tag}_bench_lines.jsonl
bash tools/profile_bench.sh ${tag}_trba6
bash tools/profile_bench.sh ${tag}_trba6_serial --serial
bash tools/profile_loop_a.sh trba ${tag}_loopa
bash tools/profile_loop_a.sh svtr ${tag}_svtr_loopa
bash tools/profile_bench.sh ${tag}_svtr6_serial --model svtr --serial
bash tools/profile_bench.sh ${tag}_crnn3 --model crnn --experts 3
MARKER=adam_kernel bash tools/profile_bench.sh ${tag}_der --loop der
bash tools/profile_bench.sh ${tag}_trba6_fp16_serial --precision fp16 --serial
bash tools/pmc_pass.sh ${tag}_pmc
bash tools/pmc_pass.sh ${tag}_pmc_loopa --loop a
bash tools/pmc_pass.sh ${tag}_pmc_svtr --model svtr
bash tools/pmc_pass.sh ${tag}_pmc_svtr_a --model svtr --loop a
# then, in the build container: bash tools/evidence_to_profiles.sh $tag
