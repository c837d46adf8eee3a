#!/bin/bash
# Issue / wait / cache counters of me_sub_kernel (and, for comparison, frame_fused_kernel and interp_luma_kernel). Run on the GPU box from the repo root.
set -e
tag=${1:-pmc_sub}
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum" "TA_BUSY_sum TA_TA_BUSY_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$n -- python3 $root/bench.py --steps 2 --warmup 1 --cpu-mbs 0 > $out/$n.log 2>&1 || echo "failed: $set"
done
cd $root
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for name in ("me_sub_kernel", "frame_fused_kernel", "interp_luma_kernel", "me_int_pair_kernel"):
            if name in k:
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-32s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
