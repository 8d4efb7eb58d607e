#!/bin/bash
# one-GPU cost of the gradient exchange, alternating sub-runs on one box (median step time of 24 steps each, host time to queue a step):
#   plain | SIMVG_FORCE_REDUCE=1 with SUM (issue cost only), 12 layer messages / grouped | with ncclAvg (production op)
cd $GRAFT_REPO_ROOT
show() { python -c 'import json,sys; j=json.loads(sys.argv[2]); print(sys.argv[1], "p50", j["ms_per_step_p50"], "mean", j["ms_per_step"], "host", j.get("host_ms_per_step"), "messages", j["reducer"].get("messages"))' "$1" "$2"; }
run() {
  local name=$1; shift
  show $name "$(env "$@" python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) \
        bench.py --gpus 1 --steps 24 --warmup 6 --no-cpu-baseline --no-forward-test --no-extras 2>/dev/null | tail -1)"
}
for rep in 1 2; do
  show plain "$(python bench.py --steps 24 --warmup 6 --no-cpu-baseline --no-forward-test --no-extras 2>/dev/null | tail -1)"
  run issue_sum_12 MASTER_ADDR=127.0.0.1 SIMVG_FORCE_REDUCE=1 SIMVG_REDUCE_OP=sum
  run issue_sum_1 MASTER_ADDR=127.0.0.1 SIMVG_FORCE_REDUCE=1 SIMVG_REDUCE_OP=sum SIMVG_REDUCE_GROUPS=12
  run avg_12 MASTER_ADDR=127.0.0.1 SIMVG_FORCE_REDUCE=1
  "$@"
done
