#!/usr/bin/env bash
# One process per GPU of this node over RCCL:  bash tools/dist_test.sh <config> <gpus> [tools/test.py arguments ...]
set -e
if [ $# -lt 2 ]; then echo "usage: $0 <config> <gpus> [test.py arguments ...]" >&2; exit 2; fi
config=$1
gpus=$2
shift 2
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
exec python -m torch.distributed.run --nnodes=1 --nproc-per-node "$gpus" --master-addr 127.0.0.1 --master-port "${PORT:-29510}" \
    "$here/test.py" "$config" --launcher pytorch "$@"
