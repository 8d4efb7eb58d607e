#!/bin/bash
# kernels of a step with the gradient exchange forced on one rank (SIMVG_FORCE_REDUCE=1 under torch.distributed.run): which launches
# exist only because a reducer is live, and what they cost
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/kr && SIMVG_FORCE_REDUCE=1 MASTER_ADDR=127.0.0.1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kr -o p -- \
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-forward-test --no-extras > /tmp/kr.log 2>&1
tail -1 /tmp/kr.log | cut -c1-160
python - <<'PY'
import csv, glob, re
fs = glob.glob('/tmp/kr/**/*kernel_stats.csv', recursive=True)
for f in fs:
    rows = list(csv.DictReader(open(f)))
    steps = max(1, sum(int(r['Calls']) for r in rows if 'adam_kernel' in r['Name']) // 2)
    tot = sum(int(r['TotalDurationNs']) for r in rows)
    print(f, "steps", steps, f"kernel time per step {tot / steps / 1e6:.3f} ms")
    for r in rows:
        n = re.sub(r'\(anonymous namespace\)::', '', r['Name'])
        if re.search(r'nccl|rccl|Dev|index|gather|scatter|cat|Copy|copy|ln_param|foreach|fill', n, re.I):
            print(f"  {n[:100]:100s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:9.1f} us {int(r['TotalDurationNs'])/steps/1e6:7.3f} ms/step")
PY
