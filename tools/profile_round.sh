#!/bin/bash
# One GPU session that produces everything kept under profiles/ for a round:
#   bash tools/profile_round.sh r02       (run from the repo root on the GPU box; scratch under gpurun_out/<round>/)
# decode: bench lines + per-shape tables, rocprofv3 kernel trace of the bench command, HBM traffic (PMC, separate passes)
# prefill: TFLOP/s per linear at M = 2048 / 8192 / 65536 (auto = variant 4, and the round-1 kernel = variant 3),
#          rocprofv3 kernel trace + PMC passes of one GEMM run, ablation builds of variant 4, zero-fill (DVFS) check
# end to end: fused decode harness tokens/s + its kernel trace
set -u
R=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$R/prof
P=$ROOT/profiles
mkdir -p $OUT $P
export TMPDIR=/tmp
# ---- decode bench lines
for wl in qwen3-4b llama3-8b qwen3-0.6b; do
  timeout 400 python bench.py --workload $wl --per-shape > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
  tail -1 $OUT/bench_$wl.json > $P/${R}_bench_$wl.jsonl
  grep us_per_launch $OUT/bench_$wl.err >> $P/${R}_bench_$wl.jsonl
done
timeout 400 python bench.py --workload llama3-70b --layers 8 --per-shape --no-cpu-baseline --no-e2e > $OUT/bench_l70.json 2> $OUT/bench_l70.err
tail -1 $OUT/bench_l70.json > $P/${R}_bench_llama3-70b_8layers.jsonl; grep us_per_launch $OUT/bench_l70.err >> $P/${R}_bench_llama3-70b_8layers.jsonl
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e > $OUT/stats.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python $ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-e2e > $OUT/fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python $ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-e2e > $OUT/write.log 2>&1
cd $ROOT
S=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
[ -n "$S" ] && python tools/pmc_summary.py stats $S $P/${R}_bench_qwen3-4b_kernel_stats.csv
F=$(dirname $(find $OUT/fetch -name "*counter_collection.csv" | head -1))
W=$(dirname $(find $OUT/write -name "*counter_collection.csv" | head -1))
python tools/pmc_summary.py pmc $F $W qwen3-4b $P/${R}_pmc_bench_qwen3-4b.json
# ---- prefill
for m in llama3-8b qwen3-4b; do
  timeout 400 python tools/bench_gemm.py --model $m --rows 2048,8192,65536 --variants 0,3 > $P/${R}_prefill_$m.jsonl 2> $OUT/gemm_$m.err
done
timeout 300 python tools/bench_gemm.py --model llama3-8b --rows 8192,65536 --variants 0,1 --dtype bf16 > $P/${R}_prefill_llama3-8b_bf16.jsonl 2> $OUT/gemm_bf16.err
timeout 300 python tools/bench_gemm.py --model llama3-8b --only gate_up_proj,down_proj --rows 8192 --variants 3,4,41,42,43 > $P/${R}_gemm_ablation.jsonl 2> $OUT/gemm_abl.err
timeout 300 python tools/bench_gemm.py --model llama3-8b --only gate_up_proj --rows 8192 --variants 3,4 --fill zero >> $P/${R}_gemm_ablation.jsonl 2>> $OUT/gemm_abl.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/gemm_stats -o g -- python $ROOT/tools/bench_gemm.py --model llama3-8b --rows 65536 --variants 0 --rounds 1 --reps 2 > $OUT/gemm_stats.log 2>&1
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $pass | cut -c1-24 | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $pass --output-format csv -d $OUT/gemm_pmc_$tag -o p -- python $ROOT/tools/bench_gemm.py --model llama3-8b --only gate_up_proj --rows 8192 --variants 3,4 --rounds 1 --reps 2 > $OUT/gemm_pmc_$tag.log 2>&1
done
cd $ROOT
S=$(find $OUT/gemm_stats -name "*kernel_stats.csv" | head -1)
[ -n "$S" ] && python tools/pmc_summary.py stats $S $P/${R}_prefill_llama3-8b_M65536_kernel_stats.csv
python tools/pmc_gemm_summary.py $OUT/gemm_pmc_* > $P/${R}_gemm_pmc.json 2> $OUT/gemm_pmc_summary.err
# ---- end to end
rm -f $P/${R}_e2e.jsonl
for m in qwen3-4b llama3-8b qwen3-0.6b; do timeout 300 python tools/bench_e2e.py --model $m >> $P/${R}_e2e.jsonl 2>> $OUT/e2e.err; done
timeout 300 python tools/bench_e2e.py --model qwen3-4b --prompt 600 >> $P/${R}_e2e.jsonl 2>> $OUT/e2e.err
timeout 300 python tools/bench_e2e.py --model qwen3-4b --prompt 1900 >> $P/${R}_e2e.jsonl 2>> $OUT/e2e.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/e2e_stats -o e -- python $ROOT/tools/bench_e2e.py --model qwen3-4b --runs 1 --warmup 1 > $OUT/e2e_stats.log 2>&1
cd $ROOT
S=$(find $OUT/e2e_stats -name "*kernel_stats.csv" | head -1)
[ -n "$S" ] && python tools/pmc_summary.py stats $S $P/${R}_e2e_qwen3-4b_kernel_stats.csv
timeout 200 python tools/bench_fused.py --model qwen3-4b > $P/${R}_fused_vs_plain.jsonl 2>> $OUT/e2e.err
timeout 200 python tools/bench_attn.py --tmax 512 --positions 0,100,255,256,511 > $P/${R}_attn_decode.jsonl 2>> $OUT/e2e.err
timeout 200 python tools/bench_attn.py --tmax 2048 >> $P/${R}_attn_decode.jsonl 2>> $OUT/e2e.err
# ---- context probes: the vendor's dense kernels on the same shapes, grid-barrier cost, kernarg fetch latency, per-wave timeline
timeout 300 python tools/bench_vendor.py --model llama3-8b > $P/${R}_vendor_llama3-8b.jsonl 2>> $OUT/vendor.err
timeout 200 python tools/bench_vendor.py --model qwen3-4b --rows 1,8192 > $P/${R}_vendor_qwen3-4b.jsonl 2>> $OUT/vendor.err
[ -x tools/probes/barrier_probe ] && timeout 100 tools/probes/barrier_probe > $P/${R}_grid_barrier_probe.jsonl 2>> $OUT/vendor.err
[ -x tools/probes/coldcode_probe ] && timeout 60 tools/probes/coldcode_probe > $P/${R}_coldcode_probe.jsonl 2>> $OUT/vendor.err
[ -x tools/probes/kernarg_probe_sload ] && { timeout 60 tools/probes/kernarg_probe_sload; timeout 60 tools/probes/kernarg_probe_preload; } > $P/${R}_kernarg_probe.jsonl 2>> $OUT/vendor.err
if [ -f paroquant_amd/_lib_diag/libparo_mi355x.so ]; then
  rm -f $P/${R}_gemv_timeline.txt
  for spec in "qkv_proj 2 16" "gate_up_proj 8 8" "o_proj 1 16" "down_proj 1 16"; do
    set -- $spec
    PARO_LIB_DIR=_lib_diag PARO_GEMV_PD=31 timeout 100 python tools/timeline_gemv.py --model qwen3-4b --linear $1 --tpw $2 --waves $3 2>> $OUT/vendor.err | grep -v amdgpu.ids >> $P/${R}_gemv_timeline.txt
  done
fi
timeout 200 python tools/bench_fused.py --model llama3-70b --tp 8 2>> $OUT/vendor.err | grep '^{' > $P/${R}_fused_tp8.jsonl
# ---- tensor parallel: what one rank of llama3-70b costs per token at TP = 1 / 2 / 4 / 8 (upper bound of the node's tokens/s)
timeout 400 python tools/tp_rank_projection.py 2>> $OUT/vendor.err | grep '^{' > $P/${R}_tp_rank_projection.jsonl
mkdir -p $ROOT/gpurun_out/$R/profiles_copy && cp $P/${R}_* $ROOT/gpurun_out/$R/profiles_copy/
tail -1 $P/${R}_bench_qwen3-4b.jsonl | cut -c1-400; head -1 $P/${R}_bench_qwen3-4b.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline'], d.get('end_to_end'), d.get('cpu_baseline'))"
