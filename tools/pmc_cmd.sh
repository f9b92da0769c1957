#!/bin/bash
# Lane activity, vector-ALU share and occupancy per kernel of one command for one build of the library:
#   tools/pmc_cmd.sh <tag> <lib.so> <command...>      (run on the GPU box; results under gpurun_out/pmc_<tag>/)
set -uo pipefail
tag="$1"; lib="$2"; shift 2
out="gpurun_out/pmc_${tag}"
mkdir -p "$out"
export TMPDIR=/tmp
export RTX_LIB="$PWD/ray-tracer_amd/lib/$lib"
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE \
    --output-format csv -d "$out/p1" -o pmc -- "$@" > "$out/p1.log" 2>&1 || { echo "pmc pass failed"; tail -5 "$out/p1.log"; exit 1; }
grep -h "frame" "$out/p1.log"
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p1/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:48]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    if m.get("SQ_INSTS_VALU", 0) < 1e4: continue
    quad = m["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0 / 4.0
    print("%-48s n=%d  waves %.0f  valu %.3g  valu/wave %.0f  salu/wave %.0f  lanes %.1f  valu_busy %.2f  waves/simd %.2f  gui_us %.0f" % (
        k, len(d["SQ_WAVES"]), m["SQ_WAVES"], m["SQ_INSTS_VALU"], m["SQ_INSTS_VALU"] / m["SQ_WAVES"], m["SQ_INSTS_SALU"] / m["SQ_WAVES"],
        m["SQ_THREAD_CYCLES_VALU"] / m["SQ_ACTIVE_INST_VALU"], m["SQ_ACTIVE_INST_VALU"] / quad, m["SQ_WAVE_CYCLES"] / quad,
        m["GRBM_GUI_ACTIVE"] / 8.0 / 2400.0))
PY
