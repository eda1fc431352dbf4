#!/bin/bash
# Quick A/B library with ONE (STATE_SIZE, CONTROL_SIZE) shape: gato_python_amd/libgato_hip_dev.so (objects under /tmp/gato_dev).
# usage: tools/devbuild.sh [S C] [extra hipcc flags] ; then GATO_HIP_LIB=$PWD/gato_python_amd/libgato_hip_dev.so python tools/...
cd "$(dirname "$0")/../gato_python_amd/csrc" || exit 1
S=${1:-14}; C=${2:-7}; shift 2 2>/dev/null
mkdir -p /tmp/gato_dev
pids=()
for f in gato_capi gato_assembly gato_pcg_resident gato_pcg_resident_dpp gato_pcg_cg1 gato_pcg_stream gato_pcg_dma; do
  if [ ! -f /tmp/gato_dev/$f.o ] || [ $f.hip -nt /tmp/gato_dev/$f.o ] || [ gato_common.h -nt /tmp/gato_dev/$f.o ] || [ gato_pcg_device.h -nt /tmp/gato_dev/$f.o ] \
     || { [ $f = gato_pcg_resident_dpp ] && [ gato_pcg_resident.hip -nt /tmp/gato_dev/$f.o ]; }; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall "-DGATO_SHAPES(X)=X($S,$C)" "$@" -c $f.hip -o /tmp/gato_dev/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p || exit 1; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libgato_hip_dev.so /tmp/gato_dev/*.o
