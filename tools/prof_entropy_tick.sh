#!/bin/bash
# front-end kernel times at a fixed tick size, by units per wave:  bash tools/prof_entropy_tick.sh <streams> shift...
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
S=$1; shift
cd /tmp && export TMPDIR=/tmp
for s in "$@"; do
  OUT=/tmp/prof_et_$s
  rm -rf "$OUT"
  SK_ENTROPY_LANE_SHIFT=$s rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$ROOT/tools/entropy_tick_bench.py" $S 4 > "$OUT.log" 2> "$OUT.err" || { tail -3 "$OUT.err"; exit 1; }
  f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
  echo "streams $S (x16 units) shift $s: $(tail -1 $OUT.log)"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_aac_entropy_parse" in r["Name"] or "k_aac_entropy_finish" in r["Name"]:
        print("   %-24s calls %s avg_us %.1f" % ("parse" if "parse" in r["Name"] else "finish", r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf "$OUT"
done
