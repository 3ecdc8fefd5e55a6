#!/usr/bin/env bash
# Build libwm_hip.so for gfx950 (cross-compiles without a GPU).  Usage: csrc/build.sh [-j N]
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# -pragma-unroll-threshold: the hand-scheduled MFMA loops (conv64bf3, resblock_eval) must unroll completely; above the default
# 16K-instruction limit the compiler silently unrolls by 2 and the register-resident weight fragments land in scratch memory
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -mllvm -pragma-unroll-threshold=100000 ${WM_EXTRA_FLAGS:-}"
mkdir -p obj
pids=()
for f in conv64 bn small_convs lstm postproc stft_loss losses gconv; do
  if [ ! -f obj/$f.o ] || [ $f.hip -nt obj/$f.o ] || [ wm_common.hpp -nt obj/$f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o obj/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libwm_hip.so obj/*.o
echo "built $(cd .. && pwd)/libwm_hip.so"
