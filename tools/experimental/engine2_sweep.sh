#!/bin/bash
# One GPU session over the engine-2 experiment builds: bash tools/experimental/engine2_sweep.sh <outdir> "<lib dirs>" "<models>" "<split specs or ->" "<env settings, comma-separated VAR=val, or ->"
#   per (library build, model, split, environment) one tools/engine2_timeline.py run, appended to <outdir>/timeline.jsonl
OUT=${1:-gpurun_out/e2}; LIBS=${2:-_lib}; MODELS=${3:-qwen3-4b}; SPLITS=${4:--}; ENVS=${5:--}
mkdir -p $OUT
for lib in $LIBS; do for m in $MODELS; do for sp in $SPLITS; do for ev in $ENVS; do
  arg=""; [ "$sp" != "-" ] && arg="--split $sp"
  evs=""; [ "$ev" != "-" ] && evs=$(echo $ev | tr ',' ' ')
  env PARO_LIB_DIR=$lib $evs timeout 120 python tools/experimental/engine2_timeline.py --model $m --layers 4 --reps 3 --tag "$lib/$ev" $arg >> $OUT/timeline.jsonl 2>> $OUT/timeline.err
done; done; done; done
