#!/usr/bin/env bash
# One process per GPU of this node over RCCL:  bash tools/dist_train.sh <config> <gpus> [tools/train.py arguments ...]
# (same positional interface as the reference's launcher; PORT overrides the rendezvous port, 127.0.0.1 is the rendezvous
# address because a container's hostname need not resolve)
set -e
if [ $# -lt 2 ]; then echo "usage: $0 <config> <gpus> [train.py arguments ...]" >&2; exit 2; fi
config=$1
gpus=$2
shift 2
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
exec python -m torch.distributed.run --nnodes=1 --nproc-per-node "$gpus" --master-addr 127.0.0.1 --master-port "${PORT:-29501}" \
    "$here/train.py" "$config" --launcher pytorch "$@"
