set -x
mkdir -p gpurun_out/r1g
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r1g/fetch -o f -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timer > $R/gpurun_out/r1g/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r1g/write -o w -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timer > $R/gpurun_out/r1g/write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/r1g/mfma -o m -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timer > $R/gpurun_out/r1g/mfma.log 2>&1
cd $R
find gpurun_out/r1g -name "*.csv" | xargs ls -la
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for tag in ("fetch", "write", "mfma"):
    for f in glob.glob(f"gpurun_out/r1g/{tag}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
        with open(f) as fh:
            rd = csv.DictReader(fh)
            for row in rd:
                k = row["Kernel_Name"][:100]
                c = row["Counter_Name"]
                a = agg[k][c]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
        out[tag] = {k: {c: {"n": v[0], "sum": v[1]} for c, v in d.items()} for k, d in agg.items()}
json.dump(out, open("gpurun_out/r1g/pmc_summary.json", "w"), indent=1)
PY
find gpurun_out/r1g -name "*.csv" -size +1M -delete
ls -la gpurun_out/r1g
