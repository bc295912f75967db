#!/bin/bash
# Compile-time variants of render_rays.hip (-D switches) as copies of libia_hip.so under tools/_variants/, timed with tools/bench_render.py:
#   tools/ablate_render.sh build "name:-DX=1 -DY=0" ...   (CPU container)   /   tools/ablate_render.sh run name ...   (GPU box)
set -e
cd "$(dirname "$0")/.."
CS=invertavatar_amd/csrc
mkdir -p tools/_variants
if [ "$1" = build ]; then
  shift
  objs=$(ls $CS/build/*.o | grep -v "render_rays\.")
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Iinclude $flags -c $CS/render_rays.hip -o tools/_variants/render_$name.o
    hipcc -shared -fPIC --offload-arch=gfx950 $objs tools/_variants/render_$name.o -o tools/_variants/libia_render_$name.so
    rm tools/_variants/render_$name.o
  done
else
  shift
  for rep in 1 2; do
    echo "== library default"; python tools/bench_render.py | grep -v "^ "
    for name in "$@"; do echo "== $name"; IA_HIP_LIB=$PWD/tools/_variants/libia_render_$name.so python tools/bench_render.py | grep -v "^ "; done
  done
fi
