#!/bin/bash
# Builds the product library (HIP kernels + C ABI + host mirror) and the test oracle.
# Usage: ./build.sh [lib|oracle|all]
set -euo pipefail
cd "$(dirname "$0")"
what="${1:-all}"
PKG="ray-tracer_amd"
OUT="$PKG/lib"
mkdir -p "$OUT" "$PKG/build"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
COMMON="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function ${RTX_EXTRA_FLAGS:-}"
LIBNAME="${RTX_LIBNAME:-librtx.so}"
if [[ "$what" == "lib" || "$what" == "all" ]]; then
  # f64 parity kernels: no FMA contraction; f32 kernels: contraction on, and no SLP vectorisation:
  # packing pairs of f32 operations into v_pk_* instructions costs more register moves (operands must
  # sit in adjacent VGPRs; table values arrive in SGPRs) than it saves (glass frame 1.38 -> 1.18 ms).
  # The bounce kernels are units of their own, built without machine-level hoisting of loop
  # invariants: the kernel is one loop around a whole color_at node, and moving the literal
  # constants of its body (the coefficients of pow alone take 34 registers) in front of that loop
  # costs 46 vector registers and a wave of occupancy.
  LICM="${RTX_LICM_FLAGS--mllvm -disable-machine-licm}"
  "$HIPCC" --offload-arch=gfx950 $COMMON ${RTX_HIP_FLAGS:-} -ffp-contract=off -c "$PKG/csrc/render_f64.hip" -o "$PKG/build/render_f64.o" &
  "$HIPCC" --offload-arch=gfx950 $COMMON ${RTX_HIP_FLAGS:-} -ffp-contract=fast -fno-slp-vectorize -c "$PKG/csrc/render_f32.hip" -o "$PKG/build/render_f32.o" &
  "$HIPCC" --offload-arch=gfx950 $COMMON $LICM ${RTX_HIP_FLAGS:-} -ffp-contract=off -c "$PKG/csrc/bounce_f64.hip" -o "$PKG/build/bounce_f64.o" &
  "$HIPCC" --offload-arch=gfx950 $COMMON $LICM ${RTX_HIP_FLAGS:-} -ffp-contract=fast -fno-slp-vectorize -c "$PKG/csrc/bounce_f32.hip" -o "$PKG/build/bounce_f32.o" &
  "$HIPCC" --offload-arch=gfx950 $COMMON -c "$PKG/csrc/scratch.hip" -o "$PKG/build/scratch.o" &
  g++ $COMMON -ffp-contract=off -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c "$PKG/csrc/rtx_api.cpp" -o "$PKG/build/rtx_api.o" &
  g++ $COMMON -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c "$PKG/csrc/rtx_multi.cpp" -o "$PKG/build/rtx_multi.o" &
  g++ $COMMON -ffp-contract=off -fno-fast-math -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c "$PKG/host/rt.cpp" -o "$PKG/build/rt.o" &
  g++ $COMMON -ffp-contract=off -fno-fast-math -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c "$PKG/host/rt_capi.cpp" -o "$PKG/build/rt_capi.o" &
  for job in $(jobs -p); do wait "$job" || { echo "error: a compile step failed"; exit 1; }; done
  "$HIPCC" --offload-arch=gfx950 -shared -fPIC -pthread -o "$OUT/$LIBNAME" "$PKG/build/render_f64.o" "$PKG/build/render_f32.o" "$PKG/build/bounce_f64.o" "$PKG/build/bounce_f32.o" "$PKG/build/scratch.o" \
      "$PKG/build/rtx_api.o" "$PKG/build/rtx_multi.o" "$PKG/build/rt.o" "$PKG/build/rt_capi.o"
  echo "built $OUT/$LIBNAME"
fi
if [[ "$what" == "oracle" || "$what" == "all" ]]; then
  gcc -std=gnu11 -O2 -ffp-contract=off -fno-fast-math -Wall -fPIC -shared -pthread oracle/rt_oracle.c -o oracle/liboracle.so -lm
  echo "built oracle/liboracle.so"
fi
