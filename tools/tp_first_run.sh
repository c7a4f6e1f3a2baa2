#!/bin/bash
# First run on a multi-GPU MI355X node: bash tools/tp_first_run.sh [N=4] [extra tools/tp_first_run.py flags]
# Steps, outputs and what each refusal means: tools/tp_first_run.py (docstring).  Nothing in this repo has run on more than one GPU yet.
set -euo pipefail
cd "$(dirname "$0")/.."
N=${1:-4}; shift || true
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
exec python tools/tp_first_run.py --gpus "$N" "$@"
