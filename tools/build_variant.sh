#!/bin/bash
# A second build of the library for same-box A/B runs (SYBL_LIBRARY=ab/<name>/libsybilgpu.so): the kernel TUs recompiled
# with extra defines, everything else linked from the in-tree objects.  usage: build_variant.sh <name> -DSYBL_...=... [...]
set -e
name=$1; shift
R=$(cd $(dirname $0)/.. && pwd)
S=$R/sybil_amd/csrc
O=$R/ab/$name
mkdir -p $O
make -s -j8 -C $S >/dev/null
KOBJ=""
for f in kernels_packed_0 kernels_packed_1 kernels_packed_2 kernels_packed_3 kernels_packed_4 kernels hashpacked; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-result -Wno-unused-value "$@" -c $S/$f.hip -o $O/$f.o ) &
  KOBJ="$KOBJ $O/$f.o"
done
wait
REST=""
for f in hashgroup hashfast distinct kernels_fast_0 kernels_fast_1 kernels_fast_2 kernels_fast_3 kernels_fast_4 engine planner table result render encode rccl loader gob writer re2lite; do REST="$REST $S/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libsybilgpu.so $KOBJ $REST -L/opt/rocm/lib -lrccl -lz -Wl,-rpath,/opt/rocm/lib
ls -la $O/libsybilgpu.so
