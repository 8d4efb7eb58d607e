#!/bin/bash
# The round's committed measurements in one call: bench line, rocprofv3 kernel stats of the same command (ViT-B bs 64), the
# ViT-L bs 32 line + kernel stats.  Usage on the GPU box:  bash tools/dev/round_profiles.sh r03_a   -> gpurun_out/<tag>_*
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py 2>gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench_line.json
rm -rf /tmp/ks && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o p -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-forward-test --no-extras > /tmp/ks.log 2>&1
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_kernel_stats.csv
python bench.py --vit large --batch 32 --steps 20 --warmup 5 --no-cpu-baseline --no-forward-test 2>/dev/null | tail -1 > gpurun_out/${TAG}_vitl_bench_line.json
rm -rf /tmp/kl && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kl -o p -- python bench.py --vit large --batch 32 --steps 6 --warmup 2 --no-cpu-baseline --no-forward-test --no-extras > /tmp/kl.log 2>&1
cp $(find /tmp/kl -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_vitl_kernel_stats.csv
python - $TAG <<'PY'
import json, sys
t = sys.argv[1]
for f in (f"gpurun_out/{t}_bench_line.json", f"gpurun_out/{t}_vitl_bench_line.json"):
    d = json.loads(open(f).read())
    print(f, {k: d.get(k) for k in ("value", "ms_per_step", "model_mfma_frac")}, "gemm_nt frac", d["roofline"]["frac"],
          "wgrad", (d.get("roofline_wgrad") or {}).get("frac"), "attn", (d.get("roofline_attn") or {}).get("frac"),
          "bf16", (d.get("bf16_line") or {}).get("value"), "hbm", {k: v.get("achieved") for k, v in (d.get("hbm_kernels") or {}).items() if isinstance(v, dict)})
PY
