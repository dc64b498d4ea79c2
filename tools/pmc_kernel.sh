#!/bin/bash
# PMC counters of ONE kernel of a micro-benchmark, separate rocprofv3 --pmc passes (MI355X_MICROARCH.md: slots per block).
# usage on the GPU box: bash tools/pmc_kernel.sh <tag> <kernel-name substring> <python script + args, relative to the repo root>
#   e.g. bash tools/pmc_kernel.sh r03_wino "ELi3ELi4EE" tools/bench_wino.py 2 6 --only-wino
tag=$1; kfilter=$2; shift 2
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
export NSHAPES=${NSHAPES:-1}
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE" "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCC_BUSY_sum TCC_TAG_STALL_sum TCC_READ_sum TCC_NORMAL_WRITEBACK_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"; do
  i=$((i+1))
  script=$1
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/$tag/p$i -o p -- python3 $R/$script "${@:2}" > $R/gpurun_out/$tag/p$i.log 2>&1
done
cd $R
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("gpurun_out/$tag/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "$kfilter" not in row["Kernel_Name"]:
            continue
        a = agg[row["Counter_Name"]]
        a[0] += 1
        a[1] += float(row["Counter_Value"])
dur = []
for f in glob.glob("gpurun_out/$tag/p2/**/*kernel_trace.csv", recursive=True):      # the pass that carries GRBM_GUI_ACTIVE
    for row in csv.DictReader(open(f)):
        if "$kfilter" in row["Kernel_Name"]:
            dur.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
with open("gpurun_out/$tag/pmc.txt", "w") as out:
    for k in sorted(agg):
        out.write(f"{k:40s} n={agg[k][0]:4d} avg={agg[k][1] / agg[k][0]:.6g}\n")
    if dur and "GRBM_GUI_ACTIVE" in agg:
        ns = sum(dur) / len(dur)
        ga = agg["GRBM_GUI_ACTIVE"][1] / agg["GRBM_GUI_ACTIVE"][0]
        out.write(f"kernel wall (same pass) {ns / 1e3:.1f} us; GRBM_GUI_ACTIVE / wall = {ga / ns:.3f} (summed over 8 XCDs: / 8 = effective GHz)\n")
print(open("gpurun_out/$tag/pmc.txt").read())
PY

rm -rf gpurun_out/$tag/p[0-9]
