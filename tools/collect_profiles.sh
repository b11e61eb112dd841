#!/bin/bash
# Round-end evidence, all in one gpurun call (run from the repo root on the GPU box):
#   tools/collect_profiles.sh <tag>    ->  gpurun_out/<tag>/{bench_*.json, *_kernel_stats.csv, pmc_*_{fetch,write}.csv}
# Every rocprofv3 command has the program itself after `--`; counters are collected in their own passes.
set -e -o pipefail
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag
mkdir -p $out && cd /tmp && export TMPDIR=/tmp
echo "[collect] bench train (default flags)"; timeout -k 10 400 python3 $R/bench.py > $out/bench_train_default.json 2> $out/bench_train.log
echo "[collect] bench infer"; timeout -k 10 400 python3 $R/bench.py --mode infer > $out/bench_infer.json 2> $out/bench_infer.log
# Kernel statistics and counters are taken with the train step on ONE stream (DCS_WGRAD_SIDE=0): co-scheduled with the side
# stream's weight gradients a kernel's duration includes the time it shares the card, and bench.py's roofline is priced on the
# kernels' own durations (its instrumented pass runs on one stream as well).  The shipped two-stream step is traced separately.
export DCS_WGRAD_SIDE=0
for mode in train infer; do
  echo "[collect] kernel stats $mode"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$mode -- python3 $R/bench.py --mode $mode --no-cpu-baseline --no-native-line > $out/stats_$mode.log 2>&1
  cp $(find $out/stats_$mode -name "*kernel_stats.csv" | head -1) $out/${mode}_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "[collect] pmc $c $mode"
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_${mode}_$c -- python3 $R/bench.py --mode $mode --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-native-line > $out/pmc_${mode}_$c.log 2>&1
    cp $(find $out/pmc_${mode}_$c -name "*counter_collection.csv" | head -1) $out/pmc_${mode}_$c.csv
  done
  rm -rf $out/stats_$mode $out/pmc_${mode}_FETCH_SIZE $out/pmc_${mode}_WRITE_SIZE
done
for mode in train infer; do
  echo "[collect] pmc MFMA utilisation $mode"
  timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $out/pmc_${mode}_mfma -- python3 $R/bench.py --mode $mode --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-native-line > $out/pmc_${mode}_mfma.log 2>&1
  cp $(find $out/pmc_${mode}_mfma -name "*counter_collection.csv" | head -1) $out/pmc_${mode}_mfma.csv
  rm -rf $out/pmc_${mode}_mfma
done
echo "[collect] sustained fp32 MFMA peak + per-layer conv tables"
hipcc -O3 --offload-arch=gfx950 $R/tools/micro/mfma_peak.hip -o /tmp/mfma_peak 2>/dev/null && timeout -k 5 120 /tmp/mfma_peak > $out/mfma_peak.txt
timeout -k 10 300 python3 $R/tools/conv_layers_bench.py 32 256 > $out/conv_layers_train.txt 2>/dev/null
timeout -k 10 300 python3 $R/tools/conv_layers_bench.py 16 2000 > $out/conv_layers_infer.txt 2>/dev/null
timeout -k 10 300 python3 $R/tools/conv_layers_bench.py 32 256 f32 > $out/conv_layers_train_native.txt 2>/dev/null
timeout -k 10 300 python3 $R/tools/conv_layers_bench.py 16 2000 f32 > $out/conv_layers_infer_native.txt 2>/dev/null
echo "[collect] error of the three conv arithmetic modes against fp64"
timeout -k 10 300 python3 $R/tools/conv_precision_check.py > $out/conv_precision.txt 2>/dev/null
echo "[collect] kernel stats with the native fp32 MFMA (comparison)"
for mode in train infer; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_native_$mode -- python3 $R/bench.py --mode $mode --f32-mfma native --no-cpu-baseline --no-native-line > $out/stats_native_$mode.log 2>&1
  cp $(find $out/stats_native_$mode -name "*kernel_stats.csv" | head -1) $out/${mode}_native_kernel_stats.csv
  rm -rf $out/stats_native_$mode
done
timeout -k 10 120 python3 $R/tools/lstm_bench.py > $out/lstm_bench.txt 2>/dev/null
echo "[collect] kernel sequence of one replayed train step / inference pass"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/kt_train -- python3 $R/bench.py --no-cpu-baseline --no-native-line --steps 4 --warmup 2 > $out/kt_train.log 2>&1
python3 $R/tools/step_sequence.py $(find $out/kt_train -name "*kernel_trace.csv" | head -1) 6 > $out/step_sequence_train.txt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/kt_infer -- python3 $R/bench.py --mode infer --no-cpu-baseline --no-native-line --steps 4 --warmup 2 > $out/kt_infer.log 2>&1
python3 $R/tools/infer_sequence.py $(find $out/kt_infer -name "*kernel_trace.csv" | head -1) > $out/step_sequence_infer.txt
rm -rf $out/kt_train $out/kt_infer
unset DCS_WGRAD_SIDE
echo "[collect] the shipped two-stream train step: kernel stats and timeline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_side -- python3 $R/bench.py --no-cpu-baseline --no-native-line --no-sub-lines > $out/stats_side.log 2>&1
cp $(find $out/stats_side -name "*kernel_stats.csv" | head -1) $out/train_side_stream_kernel_stats.csv
python3 $R/tools/step_timeline.py $(find $out/stats_side -name "*kernel_trace.csv" | head -1) 8 > $out/step_timeline_train.txt
rm -rf $out/stats_side
echo "[collect] bf16 activation storage at B = 64 (BASELINE.json configs[4], per-GPU share) beside fp32 at the same batch"
timeout -k 10 400 python3 $R/bench.py --dtype bf16 --batch 64 --no-cpu-baseline > $out/bench_train_bf16_b64.json 2> $out/bench_bf16.log
timeout -k 10 400 python3 $R/bench.py --batch 64 --no-cpu-baseline > $out/bench_train_f32_b64.json 2> $out/bench_f32_b64.log
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/kt_bf16 -- python3 $R/bench.py --dtype bf16 --batch 64 --no-cpu-baseline --no-native-line --steps 4 --warmup 2 > $out/kt_bf16.log 2>&1
python3 $R/tools/step_sequence.py $(find $out/kt_bf16 -name "*kernel_trace.csv" | head -1) 6 > $out/step_sequence_train_bf16_b64.txt
rm -rf $out/kt_bf16
echo "[collect] done"; ls -la $out
