#!/bin/bash
# Ablation table of vit_attn_kernel (f16, 512 frames x 785 tokens), run on the GPU box through gpurun after
# `make -C maavss_amd/csrc ablate`.  Each line removes parts of the loop (results are wrong by design); times on random
# and on all-zero operands (the latter is not DVFS-limited).  Output: gpurun_out/<tag>_attn_ablation.txt
TAG=${1:-r2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export MAAVSS_LIB=$R/maavss_amd/lib/libmaavss_ablate.so
OUT=$R/gpurun_out/${TAG}_attn_ablation.txt
: > $OUT
for m in 0 1 2 4 8 7 15 16 32 48 64 128 192 207 240; do
  for z in "" "--zeros"; do
    line=$(MAAVSS_ATTN_ABL=$m python3 $R/scripts/attn_bench.py --dtype 2 --iters 10 $z 2>/dev/null | grep vit_attn | sed 's/.*ntok=785: //')
    echo "mask $m ${z:-random}: $line" | tee -a $OUT
  done
done
