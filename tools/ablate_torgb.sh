#!/bin/bash
# Compile-time variants of conv1x1.hip (-D switches) as copies of libia_hip.so under tools/_variants/, timed with tools/bench_torgb.py:
#   tools/ablate_torgb.sh build "name:-DX=1" ...   (CPU container)   /   tools/ablate_torgb.sh run name ...   (GPU box)
set -e
cd "$(dirname "$0")/.."
CS=invertavatar_amd/csrc
mkdir -p tools/_variants
if [ "$1" = build ]; then
  shift
  objs=$(ls $CS/build/*.o | grep -v "conv1x1\.")
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Iinclude $flags -c $CS/conv1x1.hip -o tools/_variants/c1_$name.o
    hipcc -shared -fPIC --offload-arch=gfx950 $objs tools/_variants/c1_$name.o -o tools/_variants/libia_c1_$name.so
    rm tools/_variants/c1_$name.o
  done
else
  shift
  echo "== library default"; python tools/bench_torgb.py 2>&1 | grep "^I="
  echo "== z-grid kernel"; IA_TORGB_WIDE_P=2000000000 python tools/bench_torgb.py 2>&1 | grep "^I="
  for name in "$@"; do echo "== $name"; IA_HIP_LIB=$PWD/tools/_variants/libia_c1_$name.so python tools/bench_torgb.py 2>&1 | grep "^I="; done
fi
