#!/bin/bash
# same box, alternating runs: fused decoder layers (csrc/decoder.hip) vs the per-stage kernels of rounds 1-4
for i in 1 2; do
  for v in 0 1; do
    SIMVG_DEC_UNFUSED=$v python bench.py --steps 30 --warmup 8 --no-extras --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('unfused=$v', j['value'], j['ms_per_step'], j['ms_per_step_p50'])"
  done
done
