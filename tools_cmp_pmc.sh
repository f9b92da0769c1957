#!/bin/bash
# Instruction and cycle counters of one scene under alternative builds of the library:
#   ./tools_cmp_pmc.sh "<bench args>" lib1.so lib2.so ...
set -uo pipefail
export TMPDIR=/tmp
args="$1"; shift
for lib in "$@"; do
  out="gpurun_out/cmp_${lib%.so}"
  rm -rf "$out"; mkdir -p "$out"
  RTX_LIB="$PWD/ray-tracer_amd/lib/$lib" timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY \
      --output-format csv -d "$out" -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra $args > /dev/null 2> "$out/err.txt" || { echo "pass failed for $lib"; exit 1; }
  python3 - "$out" "$lib" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:44]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    if "primary" in k or "bounce" in k:
        print(sys.argv[2], k, " ".join("%s=%.4g" % (c.replace("SQ_", ""), sum(v) / len(v)) for c, v in sorted(d.items())))
PY
done
