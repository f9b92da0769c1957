#!/bin/bash
# Device time of every launch of one frame, by position in the frame (camera launch, then one
# launch per recursion level), from a rocprofv3 kernel trace (run on the GPU box):
#   ./tools/level_times.sh <tag> scene:precision
set -uo pipefail
tag="$1"; shift
out="$PWD/gpurun_out/levels_${tag}"
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$out" -o t -- python3 tools/scene_times.py "$@" > "$out/run.out" 2> "$out/run.err" || { echo "trace failed"; tail -3 "$out/run.err"; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
frames, cur = [], []
for s, e, n in rows:
    if "k_frame_begin" in n:
        if cur: frames.append(cur)
        cur = []
    elif "k_primary_camera" in n or "k_bounce" in n:
        cur.append((n, (e - s) / 1e3))
if cur: frames.append(cur)
frames = [f for f in frames if f and "k_primary_camera" in f[0][0]][2:]
by = collections.defaultdict(list)
for f in frames:
    for i, (n, us) in enumerate(f):
        by[i].append(us)
print("launch: median us over %d frames" % len(frames))
for i in sorted(by):
    v = sorted(by[i]); print("  %d: %9.1f" % (i, v[len(v) // 2]))
PY
