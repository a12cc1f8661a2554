#!/bin/bash
# A/B builds that differ in ONE translation unit: compiles <file>.hip with the given flags and links it with the
# in-tree objects of the other sources (csrc/build/*.o from __graft_entry__.build()).
# usage: tools/ab/build_file_variants.sh <file.hip> name "-DFLAG .." [name flags ...]
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
CSRC=$ROOT/vision-longformer_amd/csrc
F=$1; shift
BASE=$(basename $F .hip)
EXTRA=""
[ "$F" = "vil_attn_mfma.hip" ] && EXTRA="-fno-honor-nans"
OTHERS=$(ls $CSRC/build/*.o | grep -v "/$BASE.o")
while [ $# -ge 2 ]; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 $EXTRA $2 -c -o /tmp/ab_${BASE}_$1.o $CSRC/$F 2>&1 | grep -E "error|spill"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/ab/libvilattn_$1.so /tmp/ab_${BASE}_$1.o $OTHERS -L/opt/rocm/lib -lhipblaslt ) &
  shift 2
done
wait
ls -la $ROOT/tools/ab/*.so
