#!/usr/bin/env bash
# Collect the round's measurement artefacts on the GPU box (run through gpurun from the repo root):
#   kernel-trace summaries of the three single-GPU BASELINE configs, and separate FETCH_SIZE / WRITE_SIZE passes for main16 and
#   main14b_2 (counters in their own runs, MI355X_MICROARCH.md).  Results land under gpurun_out/<tag>/; the summaries that count are
#   then copied into profiles/ (profiles/summarize_pmc.py makes the per-kernel traffic JSON bench.py reads).
set -uo pipefail
tag=${1:-r3prof}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d $out/main16 -o main16 --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra > $out/main16_bench.json 2> $out/main16.err
echo "main16 stats rc=$?"
rocprofv3 --kernel-trace --stats -d $out/fwd64 -o fwd64 --output-format csv -- python3 bench.py --mode fwd --batch 64 --steps 8 --warmup 2 --no-cpu-baseline > $out/fwd64_bench.json 2> $out/fwd64.err
echo "fwd64 stats rc=$?"
rocprofv3 --kernel-trace --stats -d $out/m14 -o m14 --output-format csv -- python3 bench.py --model main14b_2 --steps 4 --warmup 1 --no-cpu-baseline > $out/m14_bench.json 2> $out/m14.err
echo "main14b_2 stats rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc16_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $out/pmc16_$c.json 2> $out/pmc16_$c.err
  echo "main16 $c rc=$?"
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc14_$c -- python3 bench.py --model main14b_2 --steps 2 --warmup 1 --no-cpu-baseline > $out/pmc14_$c.json 2> $out/pmc14_$c.err
  echo "main14b_2 $c rc=$?"
done
# in-kernel stamps (diagnostic builds, if present): where a dwgrad64bf tile spends its cycles and what clock the chip holds on random /
# all-zero operands; the per-step budget of the LSTM recurrences
P="$GRAFT_REPO_ROOT/audio-watermarking-deep-learning-watermarks-for-authenticating-speech_amd"
if [ -f "$P/libwm_hip_stamp.so" ]; then
  for m in random zeros random; do echo "== operands: $m"; python3 tests/diag_stamp_dw.py 256 $m 2>&1 | grep -v amdgpu.ids | grep -v "    wave"; done > $out/dwgrad_stamps.txt
fi
if [ -f "$P/libwm_lstm_stamp.so" ]; then python3 tests/diag_stamp_lstm.py 256 2>&1 | grep -v amdgpu.ids > $out/lstm_step_budget.txt; fi
# package power / clock level while one kernel family runs back to back (the power-limit evidence of DESIGN.md section 8)
for w in dwgrad lstm bnrelu; do echo "== $w"; timeout -k 5 60 python3 tests/diag_power.py $w 2>&1 | grep -v amdgpu.ids | tail -3; done > $out/power_clock.txt
# keep what travels back small: counter CSVs only
find $out -name "*kernel_trace.csv" -size +20M -delete
du -sh $out
