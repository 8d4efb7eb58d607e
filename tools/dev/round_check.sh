#!/bin/bash
# full GPU check of a round: pytest -m gpu, smoke, bench line, rocprofv3 kernel stats of the bench command, PMC traffic
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r02_a}
mkdir -p gpurun_out/$TAG
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/$TAG/pytest.log
tail -5 gpurun_out/$TAG/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/$TAG/smoke.log 2>&1; tail -2 gpurun_out/$TAG/smoke.log
timeout 600 python bench.py > gpurun_out/$TAG/bench_line.json 2> gpurun_out/$TAG/bench.err; cat gpurun_out/$TAG/bench_line.json
rm -rf /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-forward-test > gpurun_out/$TAG/bench_under_rocprof.log 2>&1
cp $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1) gpurun_out/$TAG/kernel_stats.csv
head -25 gpurun_out/$TAG/kernel_stats.csv | cut -c1-220
if [ "$2" = "pmc" ]; then timeout 900 bash tools/dev/pmc_bench.sh > gpurun_out/$TAG/pmc.log 2>&1; tail -40 gpurun_out/$TAG/pmc.log; fi
