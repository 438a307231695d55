#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for D in 16 64; do
  for U in 0 4096; do
    echo "== D=$D bitonic_upto=$U" | tee -a gpurun_out/k3s.log
    RVA_K3_BITONIC_UPTO=$U timeout -k 10 120 python tools/k3_stamps.py $D 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/k3s.log || exit 1
    RVA_K3_BITONIC_UPTO=$U timeout -k 10 120 python tools/load_tail.py $D 40 2>&1 | grep "D=" | tee -a gpurun_out/k3s.log || exit 1
  done
done
