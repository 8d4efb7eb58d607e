# in-situ A/B of one environment switch: bash tools/dev/ab_insitu_env.sh SIMVG_WGRAD_ASSIGN 0   (three alternating bench runs:
# default environment / VAR=VALUE)
VAR=$1; VAL=$2
for r in 1 2 3; do
  for v in default set; do
    unset $VAR
    if [ $v = set ]; then export $VAR=$VAL; fi
    timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-forward-test --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['ms_per_step_p50'])"
  done
done
