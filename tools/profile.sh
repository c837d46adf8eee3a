#!/bin/bash
# Run on the GPU box from the repo root:  bash tools/profile.sh <tag>
# 1) rocprofv3 --kernel-trace --stats of the default bench command  -> gpurun_out/<tag>/stats/
# 2) two PMC passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass, MI355X_MICROARCH.md "PMC slots") with
#    --kernel-trace only                                              -> gpurun_out/<tag>/pmc_{fetch,write}/
# 3) a per-kernel summary JSON                                        -> gpurun_out/<tag>/summary.json
# Copy what should be judged into profiles/ afterwards (gpurun_out/ is scratch).
set -e
tag=${1:-prof}
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --steps 20 --warmup 3 --cpu-mbs 0 > $out/bench_under_rocprof.json 2> $out/stats.log
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 $root/bench.py --steps 3 --warmup 1 --cpu-mbs 0 > $out/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 $root/bench.py --steps 3 --warmup 1 --cpu-mbs 0 > $out/pmc_write.log 2>&1
cd $root
python3 - "$out" <<'PY'
import csv, glob, json, re, sys, collections
out = sys.argv[1]
def pmc(d):
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob(out + "/" + d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]] += float(r["Counter_Value"]); cnt[r["Kernel_Name"]] += 1
    return {k: acc[k] / cnt[k] for k in acc}
fetch, write = pmc("pmc_fetch"), pmc("pmc_write")
stats = {}
for f in glob.glob(out + "/stats/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        stats[r["Name"]] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "pct": float(r["Percentage"])}
summ = {}
for k in sorted(set(stats) | set(fetch) | set(write)):
    m = re.search(r"(\w+)(<[^>]*>)?\(", k.replace("(anonymous namespace)::", ""))
    short = (m.group(1) + (m.group(2) or "")) if m else k
    summ[short] = {"stats": stats.get(k), "FETCH_SIZE_KB_raw_per_launch": fetch.get(k), "WRITE_SIZE_KB_raw_per_launch": write.get(k)}
json.dump(summ, open(out + "/summary.json", "w"), indent=1)
for k, v in summ.items():
    if any(s in k for s in ("me_", "interp", "tq_", "mc_", "finalize")): print(k, v)
PY
