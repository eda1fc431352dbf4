#!/bin/bash
# Same-box A/B of two builds of the library: build/ab/libgato_hip_base.so (a build of an earlier commit, not tracked)
# against the current one, alternating, with tools/pcg_time.py.  usage: tools/ab.sh [pcg_time options]
cd "$(dirname "$0")/.."
for rep in 1 2; do
  echo "== base (rep $rep)"; GATO_HIP_LIB=$PWD/build/ab/libgato_hip_base.so python tools/pcg_time.py "$@" 2>&1 | grep -v amdgpu.ids
  echo "== new  (rep $rep)"; python tools/pcg_time.py "$@" 2>&1 | grep -v amdgpu.ids
done
