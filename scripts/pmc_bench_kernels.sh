#!/bin/bash
# Per-kernel matrix-pipe occupancy and held clock of one bench.py step (rocprofv3 --pmc, kernel-trace only):
#   scripts/pmc_bench_kernels.sh <tag> [bench.py flags]   ->  gpurun_out/<tag>_kernel_pmc.json
# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * cycles), cycles = GRBM_GUI_ACTIVE / 8 (summed over the 8 XCDs);
# clock_GHz = cycles / kernel duration (MI355X_MICROARCH.md, DVFS give-back).
set -e
TAG=${1:-r2}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_k_$TAG -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $OUT/pmc_k_$TAG.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_k2_$TAG -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $OUT/pmc_k2_$TAG.log 2>&1 || echo "LDS pass failed"
cd $R
python3 - <<PY
import csv, glob, json, collections
f = glob.glob("$OUT/pmc_k_$TAG/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        n[k] += 1; dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
f2 = glob.glob("$OUT/pmc_k2_$TAG/**/*counter_collection.csv", recursive=True)
agg2 = collections.defaultdict(lambda: collections.defaultdict(float))
if f2:
    for r in csv.DictReader(open(f2[0])):
        agg2[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
rows = []
for k, c in agg.items():
    if not n[k] or dur[k] <= 0: continue
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    rows.append({"kernel": k, "launches": n[k], "avg_us": round(dur[k] / n[k] / 1e3, 1), "clock_GHz": round(cyc / dur[k], 3),
                 "mfma_busy": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), 4) if cyc else None,
                 "valu_active_over_wave_cycles": round(c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"], 4) if c["SQ_WAVE_CYCLES"] else None,
                 "wait_inst_over_wave_cycles": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 4) if c["SQ_WAVE_CYCLES"] else None,
                 "wait_any_over_wave_cycles": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4) if c["SQ_WAVE_CYCLES"] else None,
                 "total_ms": round(dur[k] / 1e6, 3)})
    c2 = agg2.get(k)
    if c2 and c2.get("GRBM_GUI_ACTIVE"):
        cyc2 = c2["GRBM_GUI_ACTIVE"] / 8.0
        # SQ_LDS_IDX_ACTIVE counts LDS-array cycles summed over the 256 CUs
        rows[-1].update(lds_busy=round(c2["SQ_LDS_IDX_ACTIVE"] / (256.0 * cyc2), 4), lds_conflict_frac=round(c2["SQ_LDS_BANK_CONFLICT"] / max(c2["SQ_LDS_IDX_ACTIVE"], 1.0), 4),
                        insts_lds_per_mfma=round(c2["SQ_INSTS_LDS"] / max(c2["SQ_INSTS_MFMA"], 1.0), 3), insts_valu_per_mfma=round(c2["SQ_INSTS_VALU"] / max(c2["SQ_INSTS_MFMA"], 1.0), 3))
rows.sort(key=lambda r: -r["total_ms"])
json.dump({"tag": "$TAG", "note": "profiled pass (clocks run 2-3 % lower than un-profiled); warm-up + 1 timed step", "kernels": rows[:30]},
          open("$OUT/${TAG}_kernel_pmc.json", "w"), indent=1)
for r in rows[:24]: print(r)
PY
