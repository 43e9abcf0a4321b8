#!/bin/bash
# Counter evidence for the attention kernel, run on the GPU box from the repo root through gpurun:
#   scripts/attn_pmc.sh <tag> [attn_bench.py flags]
# Three rocprofv3 --pmc passes (kernel-trace only; 8 SQ slots per pass) over scripts/attn_bench.py plus an un-profiled timing
# of random and all-zero operands (the DVFS give-back test of MI355X_MICROARCH.md), summarised in gpurun_out/<tag>_attn_pmc.json
# (copy into profiles/).
set -e
TAG=${1:-r2}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
python3 $R/scripts/attn_bench.py "$@" > $OUT/${TAG}_attn_random.txt
python3 $R/scripts/attn_bench.py --zeros "$@" > $OUT/${TAG}_attn_zeros.txt
cat $OUT/${TAG}_attn_random.txt $OUT/${TAG}_attn_zeros.txt
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVES"
P3="GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pmc_attn_${TAG}_$i -- python3 $R/scripts/attn_bench.py --iters 3 "$@" > $OUT/pmc_attn_${TAG}_$i.log 2>&1 || echo "pass $i failed (see log)"
done
cd $R
python3 - <<PY
import csv, glob, json, collections
out, tag = "$OUT", "$TAG"
agg, cnt, dur = collections.defaultdict(float), collections.Counter(), []
for i in (1, 2, 3):
    fs = glob.glob(f"{out}/pmc_attn_{tag}_{i}/**/*counter_collection.csv", recursive=True)
    if not fs:
        continue
    for r in csv.DictReader(open(fs[0])):
        if "vit_attn" not in r["Kernel_Name"] or "cls" in r["Kernel_Name"]:
            continue
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
        if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"):
            dur.append((r["Counter_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
c = {k: agg[k] / cnt[k] for k in agg}
res = {"tag": tag, "kernel": "vit_attn_kernel", "counters_per_launch": c,
       "unprofiled": {"random": open(f"{out}/{tag}_attn_random.txt").read().strip(), "zeros": open(f"{out}/{tag}_attn_zeros.txt").read().strip()}}
d = {}
wc = c.get("SQ_WAVE_CYCLES")
if wc:
    # SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
        if k in c: d[k + "_over_WAVE_CYCLES"] = round(c[k] / wc, 4)
if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c:
    # BUSY_CYCLES is summed over the SEs' SQs; MFMA busy over SIMDs: report the raw ratio and per-SIMD busy fraction below
    d["MFMA_BUSY_over_SQ_BUSY"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CYCLES"], 4)
gui = [x for n, x in dur if n == "GRBM_GUI_ACTIVE"]
if "GRBM_GUI_ACTIVE" in c and gui:
    ns = sum(gui) / len(gui)
    d["kernel_ns_profiled"] = ns
    d["effective_clock_GHz"] = round(c["GRBM_GUI_ACTIVE"] / 8.0 / ns, 3)       # GUI_ACTIVE is summed over the 8 XCDs
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        d["mfma_busy_fraction_per_simd"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0), 4)
if "SQ_LDS_BANK_CONFLICT" in c and "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"]:
    d["lds_bank_conflict_fraction"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
res["derived"] = d
json.dump(res, open(f"{out}/{tag}_attn_pmc.json", "w"), indent=1)
print(json.dumps(res["derived"], indent=1)); print({k: "%.4g" % v for k, v in c.items()})
PY
