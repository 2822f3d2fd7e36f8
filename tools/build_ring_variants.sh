#!/bin/bash
# Diagnostic builds of the library with one part of the ring conv kernel switched off (tools/ring_timeline.py).
set -e
cd "$(dirname "$0")/.."
SRC=rethinking-flow-and-diffusion-bridge-models-for-speech-enhancement_amd/csrc
mkdir -p tools/_dbg
bash tools/build_stamps.sh > /dev/null
for v in ${VARIANTS:-NOXFORM NOPATCH NOWEIGHTS}; do
  mkdir -p tools/_dbg/$v
  /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -DFDBM_STAMPS -DRING_X_$v ${XFLAGS} -c $SRC/conv_ring.hip -o tools/_dbg/$v/conv_ring.o &
done
wait
objs=""
for f in $SRC/*.hip $SRC/*.cpp; do
  b=$(basename "${f%.*}")
  [ "$b" = conv_ring ] && continue
  /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -DFDBM_STAMPS -c "$f" -o tools/_dbg/$b.o &
  objs="$objs tools/_dbg/$b.o"
done
wait
for v in ${VARIANTS:-NOXFORM NOPATCH NOWEIGHTS}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_dbg/libfdbm_$v.so $objs tools/_dbg/$v/conv_ring.o
done
rm -f tools/_dbg/*.o
echo built variants
