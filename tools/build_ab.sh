#!/bin/bash
# A/B builds of one HIP source: tools/build_ab.sh fir_bf16 SK_BF_ABLATE_LOAD SK_BF_ABLATE_STORE ...
# -> soundkit_amd/ab/lib_<macro>.so, selected at run time with SOUNDKIT_AMD_LIB=...
set -e
cd "$(dirname "$0")/../soundkit_amd/csrc"
make >/dev/null
src=$1; shift
mkdir -p ../ab
for X in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 -ffp-contract=off -D$X -c $src.hip -o build/${src}_$X.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../ab/lib_$X.so $(ls build/*.o | grep -v "_SK_" | grep -v "build/${src}\.o") build/${src}_$X.o
done
ls ../ab
