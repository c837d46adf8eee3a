#!/bin/bash
# instruction / wait counters of p_slice_kernel (EPZS, 1080p, one reference). Run on the GPU box from the repo root.
set -e
tag=${1:-pmc_slice}
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$n -- python3 $root/tools/time_slice.py --modes 3 --reps 1 > $out/$n.log 2>&1 || echo "failed: $set"
done
cd $root
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "p_slice_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c in sorted(acc):
    v = acc[c]
    print("   %-24s %16.0f  (n=%d)  per macroblock %.0f" % (c, sum(v) / len(v), len(v), sum(v) / len(v) / 8160))
PY
