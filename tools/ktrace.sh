#!/bin/bash
# Kernel durations (rocprofv3 --kernel-trace --stats) of one command for one build of the library:
#   tools/ktrace.sh <tag> <lib.so> <command...>      (run on the GPU box; results under gpurun_out/kt_<tag>/)
set -uo pipefail
tag="$1"; lib="$2"; shift 2
out="gpurun_out/kt_${tag}"
mkdir -p "$out"
export TMPDIR=/tmp
export RTX_LIB="$PWD/ray-tracer_amd/lib/$lib"
timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/t" -o trace -- "$@" > "$out/t.log" 2>&1 || { echo "trace failed"; tail -5 "$out/t.log"; exit 1; }
grep -h "frame" "$out/t.log"
python3 - "$out" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/t/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        print("%-60s calls %6s  avg %10.1f us  total %10.1f us  %5s%%" % (row["Name"][:60], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["TotalDurationNs"]) / 1e3, row["Percentage"]))
PY
