#!/bin/bash
set -u
O=gpurun_out/r02
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "tp_sharded or full_size or launch_shapes or ksplit" > $O/s2_pytest.log 2>&1
tail -4 $O/s2_pytest.log
timeout 300 python tools/prefetch_probe.py --workload qwen3-4b > $O/s2_prefetch_q4b.jsonl 2> $O/s2_prefetch_q4b.err
cat $O/s2_prefetch_q4b.jsonl; tail -3 $O/s2_prefetch_q4b.err
timeout 300 python tools/prefetch_probe.py --workload llama3-8b --variants 0:0,1:64,2:64 > $O/s2_prefetch_l8b.jsonl 2> $O/s2_prefetch_l8b.err
cat $O/s2_prefetch_l8b.jsonl
timeout 200 python tools/bench_gemm.py --model llama3-8b --only gate_up_proj --rows 8192 --variants 3,4 --fill zero > $O/s2_gemm_zero.jsonl 2>&1
cut -c1-220 $O/s2_gemm_zero.jsonl
R=$(pwd)
cd /tmp
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $pass | cut -c1-24 | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $pass --output-format csv -d $R/$O/s2_pmc_$tag -o p -- python $R/tools/bench_gemm.py --model llama3-8b --only gate_up_proj --rows 8192 --variants 3,4 --rounds 1 --reps 2 > $R/$O/s2_pmc_$tag.log 2>&1
done
cd $R
python tools/pmc_gemm_summary.py $O/s2_pmc_* > $O/s2_pmc_summary.json 2>&1
cat $O/s2_pmc_summary.json | head -60
