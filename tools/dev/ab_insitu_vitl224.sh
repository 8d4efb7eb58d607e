for r in 1 2 3; do
  for v in default off; do
    unset SIMVG_GEMM_224
    if [ $v = off ]; then export SIMVG_GEMM_224=0; fi
    timeout 300 python bench.py --vit large --batch 32 --steps 25 --warmup 6 --no-cpu-baseline --no-forward-test --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['ms_per_step_p50'])"
  done
done
