#!/bin/bash
# A/B of the FIR variants built by tools/build_ab.sh, alternated on one box: bash tools/ab_fir_bf16.sh "<bench args>" variant...
ARGS=$1; shift
for v in base "$@" base; do
  if [ $v = base ]; then unset SOUNDKIT_AMD_LIB; else export SOUNDKIT_AMD_LIB=$PWD/soundkit_amd/ab/lib_$v.so; fi
  timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print('$v', round(d['value']/1e6,1), {n: round(x['avg_launch_ms'],4) for n,x in k.items()})" || exit 1
done
