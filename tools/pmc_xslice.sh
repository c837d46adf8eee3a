#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes, --kernel-trace only) and SQ instruction / wait counters of the slice search's kernels:
# the exhaustive searches' sweeps over the frame kernels (me_xslice.hip: x_sim_kernel, me_int_pair_kernel<list>, me_sub_kernel<list>, x_skip_kernel)
# and the walkers' relaxation kernel (p_slice_relax_kernel). Run on the GPU box from the repo root:  bash tools/pmc_xslice.sh <tag> <modes>
#   per kernel: launches, counter per launch (average) and per call of jmhip_p_slice_search (sum over the call's launches / number of pictures)
set -e
tag=${1:-pmc_xslice}
modes=${2:--1}
frames=3
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp
sets=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES")
# third argument "icache": the instruction-cache view instead (a 14 k-instruction decision function on one wave per macroblock)
if [ "$3" = "icache" ]; then sets=("SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"); fi
for set in "${sets[@]}"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$n -- python3 $root/tools/time_slice.py --modes=$modes --frames $frames --clip bench > $out/$n.log 2>&1 || echo "failed: $set"
done
cd $root
python3 - "$out" "$frames" <<'PY'
import csv, glob, sys, collections, re
out, frames = sys.argv[1], int(sys.argv[2])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        k = re.sub(r"\(.*", "", k)
        if not any(s in k for s in ("x_", "me_int_pair", "me_sub", "p_slice", "ep_alias", "epzs_rows")): continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        unit = " KB" if c.endswith("_SIZE") else ""
        print("   %-18s launches %5d   per launch %14.1f%s   per picture %16.1f%s" % (c, len(v), sum(v) / len(v), unit, sum(v) / frames, unit))
PY
