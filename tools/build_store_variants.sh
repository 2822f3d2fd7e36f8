#!/bin/bash
# Variant libraries whose ring-kernel epilogue stores use another cache policy (see FDBM_RING_STORE_AUX in
# csrc/conv_ring_impl.h): tools/_dbg/libfdbm_store_{nt,sc1}.so, for `FDBM_HIP_LIB=... python bench.py`.
set -e
cd "$(dirname "$0")/.."
SRC=rethinking-flow-and-diffusion-bridge-models-for-speech-enhancement_amd/csrc
mkdir -p tools/_dbg
python - <<'PY'
import importlib.util
spec = importlib.util.spec_from_file_location('b', 'rethinking-flow-and-diffusion-bridge-models-for-speech-enhancement_amd/build.py')
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m); m.build(verbose=False)
PY
objs=""
for f in $SRC/*.o; do
  b=$(basename "$f")
  [ "$b" = conv_ring.o ] || [ "$b" = conv_ring8.o ] || objs="$objs $f"
done
for v in nt:2 sc1:16; do
  n=${v%%:*}; a=${v##*:}
  for k in conv_ring conv_ring8; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -DFDBM_RING_STORE_AUX=$a -c $SRC/$k.hip -o tools/_dbg/${k}_$n.o &
  done
done
wait
for n in nt sc1; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_dbg/libfdbm_store_$n.so $objs tools/_dbg/conv_ring_$n.o tools/_dbg/conv_ring8_$n.o
done
rm -f tools/_dbg/*.o
echo built tools/_dbg/libfdbm_store_nt.so tools/_dbg/libfdbm_store_sc1.so
