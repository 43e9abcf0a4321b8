#!/bin/bash
# Per-kernel timing of one bench.py run on the GPU box (run through gpurun from the repo root):
#   scripts/profile_kernels.sh <tag> [extra bench.py flags]
# writes gpurun_out/prof_<tag>/ (rocprofv3 --kernel-trace --stats, csv) and prints the top kernels.
set -e
TAG=${1:-x}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/prof_$TAG.log 2>&1
cd $R
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_$TAG/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:24]:
    print(r["Name"][:64].ljust(64), r["Calls"].rjust(5), ("%.1f" % (float(r["TotalDurationNs"]) / 1e3 / 3)).rjust(9), "us/step", ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(8), "us avg", r["Percentage"])
PY
