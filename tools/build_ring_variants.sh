#!/bin/bash
# Diagnostic builds of the library with one part of the ring conv kernel switched off (tools/ring_timeline.py, tools/ring_check.py):
#   VARIANTS="NOXFORM NOPATCH NOWEIGHTS NOPF" (RING_X_<v> in conv_ring_impl.h); STAMPS=0 builds them without the in-kernel stamps
#   (for timing A/Bs) -> tools/_dbg/libfdbm_<v>.so
set -e
cd "$(dirname "$0")/.."
SRC=rethinking-flow-and-diffusion-bridge-models-for-speech-enhancement_amd/csrc
mkdir -p tools/_dbg
ST=-DFDBM_STAMPS
[ "${STAMPS:-1}" = 0 ] && ST=
[ -n "$ST" ] && bash tools/build_stamps.sh > /dev/null
FL="-x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on $ST"
for v in ${VARIANTS:-NOXFORM NOPATCH NOWEIGHTS}; do
  mkdir -p tools/_dbg/$v
  for r in conv_ring conv_ring8; do
    /opt/rocm/bin/hipcc $FL -DRING_X_$v ${XFLAGS} -c $SRC/$r.hip -o tools/_dbg/$v/$r.o &
  done
done
wait
objs=""
for f in $SRC/*.hip $SRC/*.cpp; do
  b=$(basename "${f%.*}")
  [ "$b" = conv_ring ] && continue
  [ "$b" = conv_ring8 ] && continue
  /opt/rocm/bin/hipcc $FL -c "$f" -o tools/_dbg/$b.o &
  objs="$objs tools/_dbg/$b.o"
done
wait
for v in ${VARIANTS:-NOXFORM NOPATCH NOWEIGHTS}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_dbg/libfdbm_$v.so $objs tools/_dbg/$v/conv_ring.o tools/_dbg/$v/conv_ring8.o
done
rm -f tools/_dbg/*.o
echo built variants
