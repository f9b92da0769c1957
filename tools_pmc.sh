#!/bin/bash
# Counter passes over any command (run on the GPU box), one rocprofv3 run per pass:
#   ./tools_pmc.sh <tag> "<counters of pass 1>" ["<counters of pass 2>" ...] -- python3 tools/scene_times.py glass:f64
# Writes gpurun_out/pmc_<tag>/pmc.json (mean per dispatch and kernel) and prints it; with
# TRACE=1 a --kernel-trace --stats run comes first (gpurun_out/pmc_<tag>/trace).
set -uo pipefail
tag="$1"; shift
passes=()
while [[ $# -gt 0 && "$1" != "--" ]]; do passes+=("$1"); shift; done
shift
out="$PWD/gpurun_out/pmc_${tag}"
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
if [[ "${TRACE:-0}" == "1" ]]; then
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o trace -- "$@" > "$out/trace.out" 2> "$out/trace.err" || { echo "trace run failed"; tail -5 "$out/trace.err"; exit 1; }
fi
i=0
for pass in "${passes[@]}"; do
  i=$((i + 1))
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d "$out/pass$i" -o pmc -- "$@" > "$out/pass$i.out" 2> "$out/pass$i.err" || { echo "pmc pass failed: $pass"; tail -3 "$out/pass$i.err"; exit 1; }
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats")
    for row in csv.DictReader(open(f)):
        print("   %-60s calls %5s avg %10.1f us  min %10.1f  max %10.1f" % (row["Name"][:60], row["Calls"], float(row["AverageNs"]) / 1e3,
              float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
means = {}
for k, d in agg.items():
    if "rocclr" in k:
        continue
    print("== counters (mean per dispatch)", k)
    means[k] = {}
    for c, v in sorted(d.items()):
        print("   %-28s %.6g  (n=%d)" % (c, sum(v) / len(v), len(v)))
        means[k][c] = sum(v) / len(v)
json.dump(means, open(out + "/pmc.json", "w"), indent=1, sort_keys=True)
PY
