#!/bin/bash
# Build liblivesgpu.so (gfx950 code objects only) in-tree: lives_amd/liblivesgpu.so
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# -amdgpu-mfma-vgpr-form: MFMA results in VGPRs (no v_accvgpr_read per accumulator in the VALU-bound chain kernel)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form ${LGPU_EXTRA_FLAGS}"
OBJ=build
mkdir -p $OBJ
pids=()
for f in runtime swizzle yuv effects resize stencil palette pixbuf fused; do
  if [ ! -f $OBJ/$f.o ] || [ $f.hip -nt $OBJ/$f.o ] || [ lgpu_common.h -nt $OBJ/$f.o ] || [ ../../include/lives_gpu.h -nt $OBJ/$f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o $OBJ/$f.o &
    pids+=($!)
  fi
done
for f in host_tables layer_seam dist; do
  if [ -f $f.cpp ] && { [ ! -f $OBJ/$f.o ] || [ $f.cpp -nt $OBJ/$f.o ] || [ ../../include/lives_gpu.h -nt $OBJ/$f.o ]; }; then
    g++ -O2 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -Wall -c $f.cpp -o $OBJ/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../liblivesgpu.so $OBJ/*.o -ldl -Wl,-soname,liblivesgpu.so
echo "built $(cd .. && pwd)/liblivesgpu.so"
