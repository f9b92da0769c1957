#!/bin/bash
# HBM traffic per kernel launch (FETCH_SIZE, WRITE_SIZE in separate passes) of one command for one build:
#   tools/pmc_mem.sh <tag> <lib.so> <command...>     (run on the GPU box; results under gpurun_out/mem_<tag>/)
set -uo pipefail
tag="$1"; lib="$2"; shift 2
out="gpurun_out/mem_${tag}"
mkdir -p "$out"
export TMPDIR=/tmp
export RTX_LIB="$PWD/ray-tracer_amd/lib/$lib"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 150 rocprofv3 --pmc $c --output-format csv -d "$out/$c" -o pmc -- "$@" > "$out/$c.log" 2>&1 || { echo "pmc pass $c failed"; tail -5 "$out/$c.log"; exit 1; }
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:56]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    # FETCH_SIZE counts 32-byte units on gfx950 although the counter is documented in KiB of 64-byte requests; WRITE_SIZE in KiB
    if m.get("FETCH_SIZE", 0) + m.get("WRITE_SIZE", 0) < 1000: continue
    print("%-56s FETCH_SIZE %12.0f  WRITE_SIZE %12.0f (KiB as reported)" % (k, m.get("FETCH_SIZE", 0), m.get("WRITE_SIZE", 0)))
PY
