#!/bin/bash
# Diagnostic build of the library with in-kernel phase stamps (see tools/tap_timeline.py).
set -e
cd "$(dirname "$0")/.."
SRC=rethinking-flow-and-diffusion-bridge-models-for-speech-enhancement_amd/csrc
mkdir -p tools/_dbg
objs=""
for f in $SRC/*.hip $SRC/*.cpp; do
  o=tools/_dbg/$(basename "${f%.*}").o
  /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -DFDBM_STAMPS -c "$f" -o "$o" &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_dbg/libfdbm_hip_stamps.so $objs
rm -f tools/_dbg/*.o
echo built tools/_dbg/libfdbm_hip_stamps.so
