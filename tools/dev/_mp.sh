cd tools/dev/probes
./mfma_power 1 > /tmp/mp.log 2>&1 &
pid=$!
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do sleep 0.7; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr -s ' ' | sed 's/GPU\[0\]\t\t: //' | tr '\n' ';'; echo; done
wait $pid; cat /tmp/mp.log
./mfma_power 0
