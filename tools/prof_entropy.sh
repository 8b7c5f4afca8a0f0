#!/bin/bash
# kernel statistics of the end-to-end workload (GPU front-end) for several SK_ENTROPY_LANE_SHIFT values
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for s in "$@"; do
  OUT=$ROOT/gpurun_out/prof_ent_$s
  rm -rf "$OUT"
  SK_ENTROPY_LANE_SHIFT=$s rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$ROOT/bench.py" --workload end_to_end --streams 8192 --steps 12 --warmup 2 --gpu-entropy --lanes 1 > "$OUT.json" 2> "$OUT.err"
  f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
  echo "shift $s: $(python3 -c "import json,sys; d=json.loads(open('$OUT.json').read().strip().splitlines()[-1]); print(round(d['value']), round(d['scheduler']['gpu_tick_ms'],2), round(d['scheduler']['frames_per_tick']))")"
  grep -E "k_aac_entropy|k_aac_synth|k_fir|k_pack|k_row" "$f" | awk -F, '{gsub(/"/,""); printf "   %-60s calls %s avg_us %.1f\n", substr($1,1,60), $2, $4/1000}'
  cp "$f" "$ROOT/gpurun_out/ent_shift_${s}_kernel_stats.csv"
  rm -rf "$OUT"
done
