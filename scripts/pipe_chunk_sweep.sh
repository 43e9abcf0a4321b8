#!/bin/bash
# Same-box A/B of the two-stream pipeline and sweep of the ViT launch-group size.
# usage (GPU box): bash scripts/pipe_chunk_sweep.sh > gpurun_out/r3_pipe_chunk_sweep.txt
set -e
for rep in 1 2; do
  for mode in off on; do
    for chunk in 64 128 256 512; do
      python bench.py --pipeline $mode --vit-chunk $chunk --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | tail -1 | \
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print(f'rep $rep pipeline $mode vit-chunk $chunk: {d[\"value\"]:.1f} clips/s, {d[\"ms_per_step\"]:.2f} ms/step (second pass on one stream: {d.get(\"ms_per_step_one_stream\", d.get(\"ms_per_step_without_kernel_events\"))} ms)')"
    done
  done
done
