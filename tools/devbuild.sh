#!/bin/bash
# Quick A/B library with ONE (STATE_SIZE, CONTROL_SIZE) shape: build/ab/libgato_hip_dev.so (objects under $OBJ; build/ is
# git-ignored but travels to the GPU box with the gpurun snapshot, the package directory stays free of scratch libraries).
# OUT=name.so picks another file name under build/ab/ (several variants side by side).
# usage: [OUT=x.so] [OBJ=/tmp/dir] tools/devbuild.sh [S C] [extra hipcc flags] ; then GATO_HIP_LIB=$PWD/build/ab/libgato_hip_dev.so python tools/...
cd "$(dirname "$0")/../gato_python_amd/csrc" || exit 1
S=${1:-14}; C=${2:-7}; shift 2 2>/dev/null
OBJ=${OBJ:-/tmp/gato_dev}; OUT=${OUT:-libgato_hip_dev.so}
mkdir -p $OBJ ../../build/ab
pids=()
for f in gato_capi gato_assembly gato_pcg_resident gato_pcg_resident_dpp gato_pcg_resident_single gato_pcg_cg1 gato_pcg_stream gato_pcg_dma; do
  if [ ! -f $OBJ/$f.o ] || [ $f.hip -nt $OBJ/$f.o ] || [ gato_common.h -nt $OBJ/$f.o ] || [ gato_pcg_device.h -nt $OBJ/$f.o ] \
     || { { [ $f = gato_pcg_resident_dpp ] || [ $f = gato_pcg_resident_single ]; } && [ gato_pcg_resident.hip -nt $OBJ/$f.o ]; }; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall "-DGATO_SHAPES(X)=X($S,$C)" "$@" -c $f.hip -o $OBJ/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p || exit 1; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/ab/$OUT $OBJ/*.o
