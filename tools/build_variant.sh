#!/bin/bash
# One f64 kernel variant (RTX_ONLY_SF) with extra -D flags, linked with the objects of the last full build:
#   tools/build_variant.sh <name> <sf> [flags...]   ->  ray-tracer_amd/lib/librtx_<name>.so
set -euo pipefail
cd "$(dirname "$0")/.."
name="$1"; sf="$2"; shift 2
PKG="ray-tracer_amd"; B="$PKG/build/var_$name"; mkdir -p "$B"
COMMON="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical -DRTX_ONLY_SF=$sf $*"
/opt/rocm/bin/hipcc --offload-arch=gfx950 $COMMON -ffp-contract=off -c "$PKG/csrc/render_f64.hip" -o "$B/render_f64.o" &
/opt/rocm/bin/hipcc --offload-arch=gfx950 $COMMON -mllvm -disable-machine-licm -ffp-contract=off -c "$PKG/csrc/bounce_f64.hip" -o "$B/bounce_f64.o" &
g++ -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $* -ffp-contract=off -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c "$PKG/csrc/rtx_api.cpp" -o "$B/rtx_api.o" &
for job in $(jobs -p); do wait "$job" || { echo "compile failed"; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o "$PKG/lib/librtx_$name.so" "$B/render_f64.o" "$B/bounce_f64.o" \
  "$PKG/build/render_f32.o" "$PKG/build/bounce_f32.o" "$PKG/build/scratch.o" "$B/rtx_api.o" "$PKG/build/rtx_multi.o" "$PKG/build/rt.o" "$PKG/build/rt_capi.o"
echo "built $PKG/lib/librtx_$name.so"
