#!/bin/bash
# the fused decoder kernels of two builds side by side (rocprofv3 kernel stats of tools/dev/decoder_bench.py):  dec_ab.sh <libA> <libB>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in "$@"; do
  rm -rf /tmp/db_$L; SIMVG_HIP_LIB=simvg_amd/lib/libsimvg_$L.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/db_$L -o p -- python tools/dev/decoder_bench.py > /tmp/db_$L.log 2>&1
  echo "== $L"; grep "dec_" $(find /tmp/db_$L -name '*kernel_stats.csv' | head -1) | awk -F, '{n=$1; sub(/.*::/,"",n); printf "%-48s calls %s avg %.1f us\n", substr(n,1,46), $2, $4/1000}'
done
