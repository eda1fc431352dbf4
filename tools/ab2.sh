#!/bin/bash
# Same-box A/B of the current library against another build given as $1 (path of a libgato_hip*.so), alternating, with
# tools/pcg_time.py.  usage: tools/ab2.sh gato_python_amd/libgato_hip_x.so [pcg_time options]
cd "$(dirname "$0")/.."
OTHER=$PWD/$1; shift
for rep in 1 2; do
  echo "== current (rep $rep)"; python tools/pcg_time.py "$@" 2>&1 | grep -v amdgpu.ids
  echo "== other   (rep $rep)"; GATO_HIP_LIB=$OTHER python tools/pcg_time.py "$@" 2>&1 | grep -v amdgpu.ids
done
