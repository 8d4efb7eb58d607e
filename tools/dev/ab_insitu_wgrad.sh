# in-situ A/B of the wgrad variants (three alternating bench runs each): default plan / SIMVG_WG_FV=0 (even row partitions) /
# SIMVG_WGRAD_SQ=0 + SIMVG_WG_FV=0 (the tree before the second form)
mkdir -p gpurun_out/ab
for r in 1 2 3; do
  for v in new fv0 old; do
    unset SIMVG_WGRAD_SQ SIMVG_WG_FV
    if [ $v = fv0 ]; then export SIMVG_WG_FV=0; fi
    if [ $v = old ]; then export SIMVG_WG_FV=0 SIMVG_WGRAD_SQ=0; fi
    timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-forward-test --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['ms_per_step_p50'])"
  done
done
