#!/bin/bash
# What the waves of one scene's kernels wait for: average latency of scalar, vector and LDS instructions
# (in-flight level / instructions), instruction-fetch level, LDS bank conflicts (run on the GPU box):
#   tools/pmc_waits.sh <tag> <scene:precision>        results under gpurun_out/waits_<tag>/
set -uo pipefail
tag="$1"; scene="$2"
out="gpurun_out/waits_${tag}"
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for pass in "SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" \
            "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVES"; do
  i=$((i + 1))
  timeout -s KILL 150 rocprofv3 --pmc $pass --output-format csv -d "$out/p$i" -o pmc -- python3 tools/scene_times.py "$scene" > "$out/p$i.log" 2>&1 || { echo "pmc pass $i failed"; tail -5 "$out/p$i.log"; exit 1; }
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:56]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    if m.get("SQ_WAVE_CYCLES", 0) < 1e6: continue
    print(k)
    for c in sorted(m): print("   %-26s %14.0f" % (c, m[c]))
    def lat(level, n): return m[level] / m[n] if m.get(n) else float("nan")
    print("   -> in-flight cycles per instruction: smem %.0f  vmem %.0f  lds %.0f  ifetch %.0f" % (
        lat("SQ_INST_LEVEL_SMEM", "SQ_INSTS_SMEM"), lat("SQ_INST_LEVEL_VMEM", "SQ_INSTS_VMEM"),
        lat("SQ_INST_LEVEL_LDS", "SQ_INSTS_LDS"), lat("SQ_IFETCH_LEVEL", "SQ_IFETCH")))
PY
