mkdir -p gpurun_out/ab
for r in 1 2 3; do
  for v in sq base; do
    if [ $v = base ]; then export SIMVG_WGRAD_SQ=0; else unset SIMVG_WGRAD_SQ; fi
    timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-forward-test --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['ms_per_step_p50'])"
  done
done
