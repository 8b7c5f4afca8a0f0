#!/bin/bash
# Profiles the default bench workload on the GPU box: one rocprofv3 --kernel-trace --stats pass and three --pmc
# passes (counters in their own runs, never combined with sys/hip/hsa traces), then tools/summarize_pmc.py turns the
# CSVs into profiles/<tag>_*.  Usage (through gpurun):  bash tools/profile_bench.sh r01b [bench.py args...]
set -e
TAG=${1:-r01}
shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extras $*"
# (pass --separate-s16 to profile the chain with the separate conversion kernel)
# the stats pass runs the bench's own default step counts, so its per-kernel averages are those of the bench line
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --no-cpu-baseline $* > "$OUT/bench_stats.json" 2> "$OUT/stats.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2> "$OUT/pmc_write.err"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/pmc_sq" -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2> "$OUT/pmc_sq.err"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$OUT/pmc_inst" -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2> "$OUT/pmc_inst.err"
python3 "$ROOT/tools/summarize_pmc.py" "$OUT" "$TAG"
# gpurun copies back at most 64 MiB: keep the summaries (gpurun_out/profiles_<tag>/) and the bench line, drop the raw traces
cp "$OUT/bench_stats.json" "$ROOT/gpurun_out/profiles_$TAG/${TAG}_bench.json" 2>/dev/null || true
rm -rf "$OUT"
