#!/usr/bin/env bash
# The reference's scripts/nondist_run.sh (`python <entry> --cfg <yaml> --model-dir <dir>`) for the MI355X entry points.
#   scripts/nondist_run.sh pretrain|vqa|vcr <cfg.yaml> <model-dir> [extra flags, e.g. --steps 100 --dry-run]
# One GPU.  BASELINE config 1 (2-layer base, batch 4, 32 + 10) is this script on tests/fixtures/pretrain_small.yaml; without a GPU the
# program stops with an error (no CPU execution path) unless --dry-run is given, which resolves and prints the configuration.
set -euo pipefail
task="$1"; cfg="$2"; dir="$3"; shift 3
cd "$(dirname "${BASH_SOURCE[0]}")/.."
python -c "import importlib,sys; sys.exit(0 if importlib.import_module('vl-bert_amd.${task}.train_end2end').main(sys.argv[1:]) is not None else 0)" \
  --cfg "$cfg" --model-dir "$dir" "$@"
