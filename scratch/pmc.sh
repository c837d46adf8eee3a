#!/bin/bash
# usage: pmc.sh <outdir> <counters...>   (run on the GPU box from the repo root)
export TMPDIR=/tmp
out=$1; shift
mkdir -p gpurun_out/$out
timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $PWD/gpurun_out/$out -- python3 bench.py --steps 3 --warmup 1 --cpu-mbs 0 > gpurun_out/$out/log.txt 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/$out/*/*counter_collection.csv")[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
first=None
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"][:44]
    if first is None: first=r["Counter_Name"]
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Counter_Name"]==first: cnt[k]+=1
for k,v in acc.items():
    if "me_" in k or "interp" in k or "tq_" in k: print(k, cnt[k], {a:round(b/max(1,cnt[k])) for a,b in v.items()})
PY
