R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp NSHAPES=1
for v in on off; do
  if [ $v = off ]; then export MRN_X3_NO_CLASS_ORDER=1; fi
  for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/pf; timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pf -o p -- python3 $R/tools/bench_conv_x3.py 2 6 --no-check --only-x3 > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("/tmp/pf/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "conv_x3_kernel" in row["Kernel_Name"]:
            a = agg[row["Counter_Name"]]; a[0] += 1; a[1] += float(row["Counter_Value"])
print("class_order=$v", {k: v[1] / v[0] for k, v in agg.items()})
PY
  done
done
