#!/bin/bash
# On the GPU box: kernel times of the two-level quad pipeline for the regular build and every build/variants/lib_*.so, one box.
#   bash tools/ab_k15.sh OUTDIR [k] [reads]
O=${1:-gpurun_out/ab_k15}; K=${2:-15}; R=${3:-100000000}; mkdir -p $O
for L in kpal_amd/libkpal_hip.so build/variants/lib_*.so kpal_amd/libkpal_hip.so; do
  echo "== $L"; KPAL_HIP_LIBRARY=$L timeout 200 python tools/k15_probe.py $K $R 2>&1 | grep -v amdgpu
done > $O/k$K.txt 2>&1
cat $O/k$K.txt
