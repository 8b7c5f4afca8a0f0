#!/bin/bash
# Soak of the scheduler's slow-tick regime on the CPU (no sanitizer, so timing is close to the real thing): builds
# tests/sched_stub.cpp and runs `slow` scenarios with drawn shapes in a few processes side by side.
#   tools/sched_soak.sh [repeats per process, default 400] [processes, default 3]
# A stall ends the process with the load generator's "[sk_loadgen] STALL" record on stderr (exit code 1).
set -e
cd "$(dirname "$0")/../tests"
exe=$(mktemp /tmp/sched_soak.XXXXXX)
g++ -O2 -g -std=c++17 -Wno-subobject-linkage -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o "$exe" sched_stub.cpp -lpthread
n=${1:-400}
procs=${2:-3}
pids=()
for s in $(seq 1 "$procs"); do
    "$exe" golden/aac/aac-stereo-48k.adts slow "$n" "$((s * 11))" > "/tmp/sched_soak_$s.log" 2>&1 &
    pids+=($!)
done
rc=0
for pid in "${pids[@]}"; do wait "$pid" || rc=1; done
grep -l STALL /tmp/sched_soak_*.log && rc=1
tail -n 1 /tmp/sched_soak_*.log
rm -f "$exe"
exit $rc
