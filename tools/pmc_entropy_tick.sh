#!/bin/bash
# PMC counters of the GPU front-end kernels at a fixed tick size:  bash tools/pmc_entropy_tick.sh <streams> <shift>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
S=$1; SH=$2
cd /tmp && export TMPDIR=/tmp
i=0
for group in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  OUT=/tmp/pmc_et_$i
  rm -rf "$OUT"
  SK_ENTROPY_LANE_SHIFT=$SH rocprofv3 --kernel-trace --pmc $group --output-format csv -d "$OUT" -- python3 "$ROOT/tools/entropy_tick_bench.py" $S 3 > /dev/null 2> "$OUT.err" || { tail -3 "$OUT.err"; exit 1; }
  f=$(find "$OUT" -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"]
    if "k_aac_entropy" not in k: continue
    name = "parse" if "parse" in k else "finish" if "finish" in k else "link" if "link" in k else "seal"
    acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    acc[name]["_ns"].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
for name, c in acc.items():
    print(name, "launch_us %.0f" % (sum(c["_ns"]) / len(c["_ns"]) / 1e3), {k: round(sum(v) / len(v)) for k, v in c.items() if k != "_ns"})
PY
  rm -rf "$OUT"
  i=$((i+1))
done
