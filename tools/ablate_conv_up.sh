#!/bin/bash
# Ablated copies of libia_hip.so (conv_up.hip compiled with -DIA_UP_ABLATE=n) into tools/_variants/, timed on a GPU box:
#   tools/ablate_conv_up.sh build   (CPU container)   /   tools/ablate_conv_up.sh run   (GPU box)
set -e
cd "$(dirname "$0")/.."
CS=invertavatar_amd/csrc
mkdir -p tools/_variants
if [ "$1" = build ]; then
  objs=$(ls $CS/build/*.o | grep -v "conv_up\.")
  for n in ${ABL:-1 2 3 4}; do
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DIA_UP_ABLATE=$n -c $CS/conv_up.hip -o tools/_variants/conv_up_a$n.o
    hipcc -shared -fPIC --offload-arch=gfx950 $objs tools/_variants/conv_up_a$n.o -o tools/_variants/libia_up_a$n.so
  done
else
  echo "== full"; python tools/bench_upconv.py | grep "I="
  for n in ${ABL:-1 2 3 4}; do echo "== ablate $n"; IA_HIP_LIB=$PWD/tools/_variants/libia_up_a$n.so python tools/bench_upconv.py | grep "I="; done
fi
