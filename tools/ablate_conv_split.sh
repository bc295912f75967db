#!/bin/bash
# Builds ablated copies of libia_hip.so (conv_split.hip compiled with -DIA_ABLATE=n) into tools/_variants/ and, on a GPU box, times the
# large layers with each:   tools/ablate_conv_split.sh build   (CPU container)   /   tools/ablate_conv_split.sh run   (GPU box)
set -e
cd "$(dirname "$0")/.."
CS=invertavatar_amd/csrc
mkdir -p tools/_variants
if [ "$1" = build ]; then
  objs=$(ls $CS/build/*.o | grep -v conv_split)
  for n in ${ABL:-1 2 3 4 5 6 7}; do
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DIA_ABLATE=$n -c $CS/conv_split.hip -o tools/_variants/conv_split_a$n.o
    hipcc -shared -fPIC --offload-arch=gfx950 $objs tools/_variants/conv_split_a$n.o -o tools/_variants/libia_a$n.so
  done
else
  echo "== full"; python tools/bench_conv_split.py | grep "I="
  for n in ${ABL:-1 2 3 4 5 6 7}; do echo "== ablate $n"; IA_HIP_LIB=$PWD/tools/_variants/libia_a$n.so python tools/bench_conv_split.py | grep "I="; done
fi
