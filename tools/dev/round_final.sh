#!/bin/bash
# A round's closing run in one call, in the order the bench line needs: PMC traffic of the built sources first (copied into
# profiles/ on the box, so that the bench line's `traffic` figures are stamped with the sources they were measured on), then the
# whole GPU test suite, smoke, the default bench line, rocprofv3 kernel stats of the same command, ViT-L, num_queries = 10.
# Usage on the GPU box:  bash tools/dev/round_final.sh r05_e   -> gpurun_out/<tag>_*
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 bash tools/dev/pmc_bench.sh > gpurun_out/${TAG}_pmc.log 2>&1 && cp gpurun_out/pmc/hbm_traffic.json profiles/gemm_nt_hbm_traffic.json && cp gpurun_out/pmc/hbm_traffic.json gpurun_out/${TAG}_hbm_traffic.json
tail -16 gpurun_out/${TAG}_pmc.log
timeout 1800 python -m pytest tests -m gpu -x -q -rs > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/${TAG}_pytest.log
tail -4 gpurun_out/${TAG}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1; tail -2 gpurun_out/${TAG}_smoke.log
bash tools/dev/round_profiles.sh $TAG
python bench.py --queries 10 --steps 20 --warmup 5 --no-cpu-baseline --no-forward-test --no-extras 2>/dev/null | tail -1 > gpurun_out/${TAG}_q10_bench_line.json
python -c "import json; d=json.load(open('gpurun_out/${TAG}_q10_bench_line.json')); print('q10', d['value'], d['ms_per_step'])"

