#!/bin/bash
# scripts/pmc_counters.sh <tag> "<counters>" <python script + args>   (one rocprofv3 --pmc pass, kernel-trace only)
set -e
TAG=$1; CTRS=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $R/gpurun_out/pmc_$TAG -- python3 "$@" > $R/gpurun_out/pmc_$TAG.log 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_$TAG/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "panel" in k or "vit" in k or "conv" in k:
        print(k, {c: "%.4g" % (v / cnt[(k, c)]) for c, v in d.items()})
PY
