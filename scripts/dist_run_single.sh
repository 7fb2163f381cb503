#!/usr/bin/env bash
# The reference's scripts/dist_run_single.sh (`python ./scripts/launch.py --nproc_per_node N <entry> --cfg <yaml> --model-dir <dir>`):
# one process per GPU of this node over RCCL.
#   scripts/dist_run_single.sh <N> pretrain|vqa|vcr <cfg.yaml> <model-dir> [extra flags]
set -euo pipefail
n="$1"; task="$2"; cfg="$3"; dir="$4"; shift 4
cd "$(dirname "${BASH_SOURCE[0]}")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
export PYTHONPATH="$PWD${PYTHONPATH:+:$PYTHONPATH}"
cat > /tmp/vlb_entry_$$.py <<PY
import importlib, sys
importlib.import_module("vl-bert_amd.${task}.train_end2end").main(sys.argv[1:])
PY
python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port "${MASTER_PORT:-29577}" \
  /tmp/vlb_entry_$$.py --cfg "$cfg" --model-dir "$dir" --dist "$@"
rm -f /tmp/vlb_entry_$$.py
