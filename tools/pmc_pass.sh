#!/bin/bash
# rocprofv3 PMC passes over bench.py (1 warm-up + 1 timed step), one counter group per run as the MI355X guide
# prescribes (FETCH_SIZE and WRITE_SIZE cannot share a pass).  Usage on the GPU box: bash tools/pmc_pass.sh <tag>
# Writes gpurun_out/<tag>/pmc_summary.json : {counter: {kernel: {"n": launches, "sum": total}}}.
# Extra arguments go to bench.py (e.g. `bash tools/pmc_pass.sh r04_pmc_loopa --loop a`: the loop-A step has its own G = 1 launches).
tag=${1:-pmc}
shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/$tag/p$i -o p -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-extra --no-power-probe "$@" > $R/gpurun_out/$tag/p$i.log 2>&1
done
cd $R
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("gpurun_out/$tag/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        a = agg[row["Counter_Name"]][row["Kernel_Name"].replace("(anonymous namespace)::", "")[:90]]
        a[0] += 1
        a[1] += float(row["Counter_Value"])
out = {c: {k: {"n": v[0], "sum": v[1]} for k, v in sorted(d.items(), key=lambda kv: -kv[1][1])[:28]} for c, d in agg.items()}
json.dump(out, open("gpurun_out/$tag/pmc_summary.json", "w"), indent=1)
for c, d in out.items():
    k, v = next(iter(d.items()))
    print(c, k[:60], v)
PY
rm -rf gpurun_out/$tag/p[0-9]
