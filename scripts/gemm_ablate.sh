#!/bin/bash
# Ablation table of vit_ws_gemm_kernel (f16, 512 frames x 785 tokens), run on the GPU box through gpurun after
# `make -C maavss_amd/csrc ablate_ws`.  Each line removes parts of the loop (results are wrong by design): 1 epilogue,
# 2 panel fetch + deposit, 4 MFMAs, 8 fragment reads, 16 barrier.  Output: gpurun_out/<tag>_gemm_ablation.txt
TAG=${1:-r2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${TAG}_gemm_ablation.txt
: > $OUT
for m in 0 1 2 3 4 8 12 19 27; do
  if [ $m = 0 ]; then export MAAVSS_LIB=$R/maavss_amd/lib/libmaavss_hip.so; else export MAAVSS_LIB=$R/maavss_amd/lib/libmaavss_wsabl$m.so; fi
  for z in "" "--zeros"; do
    line=$(python3 $R/scripts/gemm_bench.py --reps 10 $z 2>/dev/null | grep -E "^(qkv|fc1)" | sed 's/.*weight-stationary//' | tr '\n' ' ')
    echo "mask $m ${z:-random}: $line" | tee -a $OUT
  done
done
