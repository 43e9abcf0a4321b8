#!/bin/bash
# Counters for the K = 384 ViT GEMMs (scripts/gemm_bench.py), run on the GPU box from the repo root through gpurun:
#   scripts/gemm_pmc.sh <tag>
# rocprofv3 --pmc passes (kernel-trace only); per-kernel averages in gpurun_out/<tag>_gemm_pmc.json (copy into profiles/).
set -e
TAG=${1:-r2}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="FETCH_SIZE"
P2="WRITE_SIZE"
P3="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P4="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pmc_gemm_${TAG}_$i -- python3 $R/scripts/gemm_bench.py --reps 3 "$@" > $OUT/pmc_gemm_${TAG}_$i.log 2>&1 || echo "pass $i failed (see log)"
done
cd $R
python3 - <<PY
import csv, glob, json, collections
out, tag = "$OUT", "$TAG"
agg, cnt, dur = collections.defaultdict(lambda: collections.defaultdict(float)), collections.defaultdict(collections.Counter), collections.defaultdict(list)
for i in (1, 2, 3, 4):
    for f in glob.glob(f"{out}/pmc_gemm_{tag}_{i}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "vit_" not in k:
                continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
            if r["Counter_Name"] in ("GRBM_GUI_ACTIVE", "FETCH_SIZE"):
                dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
res = {}
for k in agg:
    c = {n: agg[k][n] / cnt[k][n] for n in agg[k]}
    d = sorted(dur[k])[len(dur[k]) // 2] if dur[k] else None
    row = {"median_duration_us_under_profiler": d / 1e3 if d else None, "counters_per_launch": c}
    if "FETCH_SIZE" in c:
        row["hbm_read_MB_x2_corrected"] = round(2 * c["FETCH_SIZE"] / 1024, 1)
    if "WRITE_SIZE" in c:
        row["hbm_write_MB"] = round(c["WRITE_SIZE"] / 1024, 1)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        row["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * c["GRBM_GUI_ACTIVE"] / 8 * 256), 3)   # per SIMD: 1024 SIMDs
    res[k] = row
json.dump({"tag": tag, "note": "rocprofv3 --kernel-trace --pmc passes over scripts/gemm_bench.py; FETCH_SIZE (KB) doubled per MI355X_MICROARCH.md", "kernels": res},
          open(f"{out}/{tag}_gemm_pmc.json", "w"), indent=1)
for k, v in res.items():
    print(k, {a: b for a, b in v.items() if a != "counters_per_launch"})
PY
