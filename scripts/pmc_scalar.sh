#!/bin/bash
# Scalar-unit load per kernel of one bench.py step (the scalar ALU is one per CU, shared by its waves):
#   scripts/pmc_scalar.sh <tag>  ->  gpurun_out/<tag>_scalar_pmc.json   (SALU / VALU / SMEM instruction counts, scalar-active cycles per wave cycle)
set -e
TAG=${1:-rX}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_INST_CYCLES_SALU --output-format csv -d $OUT/pmc_s_$TAG -- python3 $R/bench.py --pipeline off --steps 1 --warmup 1 --no-cpu-baseline "$@" > $OUT/pmc_s_$TAG.log 2>&1
cd $R
python3 - <<PY
import csv, glob, json, collections
f = glob.glob("$OUT/pmc_s_$TAG/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_INSTS_SALU":
        dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); n[k] += 1
rows = []
for k, c in agg.items():
    v = max(c["SQ_INSTS_VALU"], 1.0)
    rows.append({"kernel": k, "launches": n[k], "total_ms": round(dur[k] / 1e6, 3), "salu_per_valu": round(c["SQ_INSTS_SALU"] / v, 3),
                 "smem_per_valu": round(c["SQ_INSTS_SMEM"] / v, 4), "valu_per_mfma": round(v / max(c["SQ_INSTS_MFMA"], 1.0), 2),
                 "sca_active_over_wave_cycles": round(c["SQ_ACTIVE_INST_SCA"] / max(c["SQ_WAVE_CYCLES"], 1.0), 4),
                 "salu_cycles_over_wave_cycles": round(c["SQ_INST_CYCLES_SALU"] / max(c["SQ_WAVE_CYCLES"], 1.0), 4)})
rows.sort(key=lambda r: -r["total_ms"])
json.dump({"tag": "$TAG", "kernels": rows[:40]}, open("$OUT/${TAG}_scalar_pmc.json", "w"), indent=1)
for r in rows[:40]: print(r)
PY
