#!/bin/bash
# Times the headline workload with alternative builds of the library (RTX_LIB), interleaved.
# usage: ./tools_variants.sh "<bench args>" lib1.so lib2.so ...
args="$1"; shift
for round in 1 2; do
for lib in "$@"; do
  RTX_LIB="$PWD/ray-tracer_amd/lib/$lib" timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-22s frame %.4f ms  kernel %.4f ms  frac %.3f' % ('$lib', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
done; done
