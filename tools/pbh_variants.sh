#!/bin/bash
# profiling builds of liblivesgpu.so with parts of k_pb_half switched off (timing probes only; results are wrong by construction):
#   bit 0: plain loads instead of non-temporal, bit 1: no DPP lane exchange, bit 2: no f64 division.   -> lives_amd/liblivesgpu_pbhN.so (LGPU_SO=...)
set -e
cd "$(dirname "$0")/../lives_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -mllvm -amdgpu-mfma-vgpr-form"
for v in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DPBH_VARIANT=$v -c pixbuf.hip -o build/pixbuf_v$v.o.tmp
  objs=$(ls build/*.o | grep -v pixbuf)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../liblivesgpu_pbh$v.so $objs build/pixbuf_v$v.o.tmp -ldl
  rm build/pixbuf_v$v.o.tmp
done
