#!/bin/bash
# VALU / LDS / wait counters of the integer-search kernel (default persistent form, and JMHIP_ME_KERNEL=pair). Run on the GPU box from the repo root.
set -e
tag=${1:-pmc_me}
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pers_$n -- python3 $root/bench.py --steps 2 --warmup 1 --cpu-mbs 0 > $out/pers_$n.log 2>&1 || echo "failed: $set"
  JMHIP_ME_KERNEL=pair timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pair_$n -- python3 $root/bench.py --steps 2 --warmup 1 --cpu-mbs 0 > $out/pair_$n.log 2>&1 || echo "failed: $set"
done
cd $root
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    kind = f.split("/")[-3].split("_")[0]
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "me_int" in k:
            acc[kind + ":" + k.split("(")[0][-24:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-24s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
