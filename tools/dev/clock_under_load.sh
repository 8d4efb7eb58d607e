#!/bin/bash
# sclk / power while one GEMM shape runs back to back: usage clock_under_load.sh <nt|tn> [seconds] [reps] [shape]
which=${1:-nt}; secs=${2:-6}
case $which in
  nt|tn) cmd="python tools/dev/gemm_bench.py ${3:-80000} $which ${4:-fc1}" ;;
  *) echo "unknown"; exit 1 ;;
esac
$cmd > gpurun_out/clock_load_$which.log 2>&1 &
pid=$!
sleep 4     # import + warm-up
for i in $(seq 1 $secs); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|fclk|mclk" | tr -s ' ' | tr '\n' ';'; echo
  sleep 1
done
kill $pid 2>/dev/null; wait $pid 2>/dev/null
tail -3 gpurun_out/clock_load_$which.log
