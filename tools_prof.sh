#!/bin/bash
# Profiling recipe for the headline workload (run on the GPU box):
#   ./tools_prof.sh <tag> [bench args...]
# writes rocprofv3 kernel-trace stats and PMC passes under gpurun_out/prof_<tag>/
set -uo pipefail
tag="$1"; shift
out="gpurun_out/prof_${tag}"
mkdir -p "$out"
export TMPDIR=/tmp
ARGS="--steps 50 --warmup 5 --no-cpu-baseline --no-extra $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o trace -- python3 bench.py $ARGS > "$out/bench.json" 2> "$out/trace.err" || { echo "trace run failed"; tail -5 "$out/trace.err"; exit 1; }
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
            "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo "$pass" | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d "$out/pmc_$name" -o pmc -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra $* > /dev/null 2> "$out/pmc_$name.err" || echo "pmc pass failed: $pass"
done
python3 - "$out" <<'PY'
import csv, glob, sys, os, collections
out = sys.argv[1]
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats", f)
    print(open(f).read()[:3000])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
import json
means = {}
for k, d in agg.items():
    print("== counters (mean per dispatch)", k)
    means[k] = {}
    for c, v in sorted(d.items()):
        print("   %-28s %.6g  (n=%d)" % (c, sum(v) / len(v), len(v)))
        means[k][c] = sum(v) / len(v)
json.dump(means, open(out + "/pmc.json", "w"), indent=1, sort_keys=True)
PY
