#!/bin/bash
# kernel + copy timeline of the end-to-end workload (GPU front-end, one lane): per tick, busy time by kind and the gaps
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/tl
rm -rf "$OUT"
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT" -- python3 "$ROOT/bench.py" --workload end_to_end --streams 8192 --steps 12 --warmup 2 --gpu-entropy --lanes 1 > /dev/null 2> "$OUT.err"
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
root = sys.argv[1]
ev = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0].split("::")[-1][:28]))
for f in glob.glob(os.path.join(root, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy_" + r.get("Direction", r.get("Name", "?"))[:20]))
ev.sort()
# ticks: from one k_aac_entropy_parse to the next
starts = [i for i, e in enumerate(ev) if e[2].startswith("k_aac_entropy_parse")]
rows = []
for a, b in zip(starts, starts[1:]):
    seg = ev[a:b]
    parse = seg[0][1] - seg[0][0]
    if parse < 0.4e6: continue  # small ticks
    busy = {}
    last_end = seg[0][0]; gap = 0
    for s, e, n in seg:
        busy[n] = busy.get(n, 0) + (e - s)
        if s > last_end: gap += s - last_end
        last_end = max(last_end, e)
    rows.append((ev[b][0] - seg[0][0], gap, busy))
if rows:
    n = len(rows)
    print("big ticks:", n, "period ms %.2f" % (sum(r[0] for r in rows) / n / 1e6), "gaps inside ms %.2f" % (sum(r[1] for r in rows) / n / 1e6))
    keys = {}
    for r in rows:
        for k, v in r[2].items(): keys[k] = keys.get(k, 0) + v
    for k, v in sorted(keys.items(), key=lambda kv: -kv[1])[:14]:
        print("   %-30s %.3f ms per tick" % (k, v / n / 1e6))
PY
rm -rf "$OUT"
