#!/bin/bash
# Profiling recipe for the headline workload (run on the GPU box):
#   ./tools_prof.sh <tag> [bench args...]
# writes rocprofv3 kernel-trace stats and PMC passes under gpurun_out/prof_<tag>/
set -uo pipefail
tag="$1"; shift
out="gpurun_out/prof_${tag}"
mkdir -p "$out"
export TMPDIR=/tmp
ARGS="--steps 50 --warmup 5 --ramp-ms 0 --no-cpu-baseline --no-extra $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o trace -- python3 bench.py $ARGS > "$out/bench.json" 2> "$out/trace.err" || { echo "trace run failed"; tail -5 "$out/trace.err"; exit 1; }
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
            "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH GRBM_GUI_ACTIVE" \
            "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo "$pass" | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d "$out/pmc_$name" -o pmc -- python3 bench.py --steps 5 --warmup 1 --ramp-ms 0 --no-cpu-baseline --no-extra $* > /dev/null 2> "$out/pmc_$name.err" || echo "pmc pass failed: $pass"
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
# derived figures per kernel: lanes active per vector instruction (divergence), share of the SIMDs'
# cycles the vector ALU was busy, resident waves per SIMD (occupancy) -- counters of different passes,
# so only as good as the kernel is repeatable
for k, m in means.items():
    d = {}
    if m.get("SQ_ACTIVE_INST_VALU") and m.get("SQ_THREAD_CYCLES_VALU"):
        d["active_lanes_per_valu_instruction"] = m["SQ_THREAD_CYCLES_VALU"] / m["SQ_ACTIVE_INST_VALU"]
    if m.get("GRBM_GUI_ACTIVE"):
        simd_quad_cycles = m["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0 / 4.0  # 8 XCDs report, 1024 SIMDs, counters tick every 4 cycles
        if m.get("SQ_ACTIVE_INST_VALU"):
            d["valu_busy"] = m["SQ_ACTIVE_INST_VALU"] / simd_quad_cycles
        if m.get("SQ_WAVE_CYCLES"):
            d["mean_resident_waves_per_simd"] = m["SQ_WAVE_CYCLES"] / simd_quad_cycles
        if m.get("SQ_INSTS"):
            d["instructions_per_simd_issue_slot"] = m["SQ_INSTS"] / simd_quad_cycles
    if d:
        m["derived"] = d
        print("== derived", k, " ".join("%s=%.3f" % kv for kv in d.items()))
json.dump(means, open(out + "/pmc.json", "w"), indent=1, sort_keys=True)
PY
