#!/bin/bash
# A/B of library variants built by tools/build_ab.sh (soundkit_amd/ab/lib_<variant>.so), alternated on one box around the default build:
#   bash tools/ab_bench.sh "<bench args>" variant...   -> frames/s and the per-kernel launch times of each
ARGS=$1; shift
for v in base "$@" base; do
  if [ $v = base ]; then unset SOUNDKIT_AMD_LIB; else export SOUNDKIT_AMD_LIB=$PWD/soundkit_amd/ab/lib_$v.so; fi
  timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernels') or {d['roofline']['kernel']: d['roofline']}; print('$v', round(d['value']/1e6,1), {n: round(x['avg_launch_ms'],4) for n,x in k.items()})" || exit 1
done
