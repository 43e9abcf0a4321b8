#!/bin/bash
# End-of-milestone evidence, run on the GPU box from the repo root (through gpurun):
#   scripts/round_profiles.sh <tag>
# writes into gpurun_out/ (copy the summaries into profiles/ afterwards):
#   <tag>_bench.json               the default bench.py line (with cpu_baseline)
#   <tag>_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats of bench.py --pipeline off (one stream: a kernel's duration is its
#                                  own, which is what the bench line's `roofline` reports from its one-stream pass)
#   <tag>_bench_kernel_stats_two_streams.csv   the same for the default two-stream run (kernels share the chip: longer launches)
#   <tag>_pmc_hbm_traffic.json     FETCH_SIZE / WRITE_SIZE per kernel launch, separate --pmc passes, FETCH doubled
#                                  (gfx950 tallies 128-B requests at 64 B; MI355X_MICROARCH.md, HBM section)
set -e
TAG=${1:-rX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
python3 $R/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 600 $OUT/${TAG}_bench.json; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 $R/bench.py --pipeline off --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_$TAG.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof2_$TAG -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof2_$TAG.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$TAG -- python3 $R/bench.py --pipeline off --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch_$TAG.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$TAG -- python3 $R/bench.py --pipeline off --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_write_$TAG.log 2>&1
cd $R
python3 - <<PY
import csv, glob, json, collections, shutil
out = "$OUT"; tag = "$TAG"
shutil.copy(glob.glob(f"{out}/prof_{tag}/**/*kernel_stats.csv", recursive=True)[0], f"{out}/{tag}_bench_kernel_stats.csv")
shutil.copy(glob.glob(f"{out}/prof2_{tag}/**/*kernel_stats.csv", recursive=True)[0], f"{out}/{tag}_bench_kernel_stats_two_streams.csv")
def per_kernel(d, name):
    f = glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True)[0]
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name: continue
        k = r["Kernel_Name"].split("(")[0]
        tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    return tot, cnt
ft, fc = per_kernel(f"pmc_fetch_{tag}", "FETCH_SIZE")
wt, wc = per_kernel(f"pmc_write_{tag}", "WRITE_SIZE")
rows = []
for k in ft:
    rows.append({"kernel": k, "launches": fc[k], "fetch_MB_per_launch_x2_corrected": round(2 * ft[k] / fc[k] / 1024, 1),
                 "write_MB_per_launch": round(wt.get(k, 0.0) / max(wc.get(k, 1), 1) / 1024, 1)})
rows.sort(key=lambda r: -(r["fetch_MB_per_launch_x2_corrected"] + r["write_MB_per_launch"]) * r["launches"])
import hashlib
sha = hashlib.sha256(open("$R/maavss_amd/lib/libmaavss_hip.so", "rb").read()).hexdigest()
json.dump({"tag": tag, "lib_sha256": sha, "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 1; FETCH_SIZE (KB) doubled per "
                   "MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads), WRITE_SIZE as is", "kernels": rows[:40]},
          open(f"{out}/{tag}_pmc_hbm_traffic.json", "w"), indent=1)
for r in rows[:12]: print(r)
PY
