#!/bin/bash
# Build a SEPARATE library with the Winograd kernel's phase stamps compiled in (csrc/conv_wino.h FAVAE_WINO_TRACE) -> tools/experiments/libfavae_trace.so
# (run here, on the build container: hipcc cross-compiles; the .so travels with gpurun).  Then on the GPU box: python tools/wino_trace.py
set -e
cd "$(dirname "$0")/../fa-vae_amd/csrc"
B=/tmp/favae_trace_build; mkdir -p $B
for f in conv norm gemm blur ffl vq misc lpips trans prof; do
  if [ $f = conv ]; then X="-DFAVAE_WINO_TRACE"; else X=""; fi
  if [ $f = conv ] || [ ! -f $B/$f.o ] || [ $f.hip -nt $B/$f.o ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -w $X -c $f.hip -o $B/$f.o &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $B/*.o -o ../../tools/experiments/libfavae_trace.so
ls -la ../../tools/experiments/libfavae_trace.so
