#!/bin/bash
# The round's rocprofv3 evidence, run on the GPU box:  bash tools/profile_round.sh [B ...]   (default: 1 64)
# Writes gpurun_out/prof/: kernel stats of bench.py at batch B (the summaries bench.py's roofline must agree with), the
# PMC FETCH_SIZE / WRITE_SIZE passes of an eager forward (separate passes, no other trace domain) and their summary,
# and one pass of SQ instruction counters.
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/prof
mkdir -p $O
for B in ${@:-1 64}; do
  steps=3; [ "$B" -ge 16 ] && steps=1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b$B -- python3 bench.py --batch $B --steps $steps --warmup 1 --no-cpu-baseline --no-extras > $O/bench_b${B}_under_rocprof.json 2> $O/stats_b$B.err
  cp "$(find $O/stats_b$B -name '*kernel_stats.csv' | head -1)" $O/kernel_stats_b${B}_bf16.csv
  echo "stats B=$B done"
  # PMC passes over the SAME command (graph replays included; round 1's SIGSEGV under --pmc disappeared with the
  # hipMemsetAsync graph node: tools/pmc_graph_repro.py), algorithmic bytes from a plain eager pass
  python3 tools/pmc_workload.py $B $O/algo_b$B.json > $O/algo_b$B.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_b$B -- python3 bench.py --batch $B --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/fetch_b$B.log 2>&1
  echo "fetch B=$B done"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_b$B -- python3 bench.py --batch $B --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/write_b$B.log 2>&1
  echo "write B=$B done"
  python3 tools/pmc_summarize.py $O/fetch_b$B $O/write_b$B $O/algo_b$B.json > $O/pmc_traffic_b${B}_bf16.json
  if rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/sq_b$B -- python3 bench.py --batch $B --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/sq_b$B.log 2>&1; then
    python3 tools/pmc_sq_summarize.py $O/sq_b$B > $O/sq_counters_b${B}_bf16.txt 2>&1 || true
  else
    echo "sq pass failed (see $O/sq_b$B.log)"
  fi
  echo "B=$B done"
  # raw traces are large: keep the summaries only
  rm -rf $O/stats_b$B $O/fetch_b$B $O/write_b$B $O/sq_b$B
done
# the split-precision parity mode (f32 tensors, three f16 MFMAs per product) at batch 1: kernel stats of its bench line
if [ -n "$F32S" ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_f32s -- python3 bench.py --dtype f32s --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_b1_f32s_under_rocprof.json 2> $O/stats_f32s.err
  cp "$(find $O/stats_f32s -name '*kernel_stats.csv' | head -1)" $O/kernel_stats_b1_f32s.csv
  rm -rf $O/stats_f32s
  echo "f32s done"
fi
# BASELINE configs[4] (fp16 storage, batch 16, N = 100): kernel stats of the stochastic and the predictor-corrector sampler
if [ -n "$CONFIG4" ]; then
  for S in sde_ei pc; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4_$S -- python3 bench.py --batch 16 --N 100 --dtype f16 --sampler $S --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_config4_${S}_under_rocprof.json 2> $O/stats_c4_$S.err
    cp "$(find $O/stats_c4_$S -name '*kernel_stats.csv' | head -1)" $O/kernel_stats_config4_${S}_b16_f16.csv
    rm -rf $O/stats_c4_$S
    echo "config4 $S done"
  done
fi
