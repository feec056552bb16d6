#!/bin/bash
# Diagnostic builds of libdisco_hip.so with parts of conv3x3_mx_kernel switched off (results are WRONG by design; timing only):
#   MX_ABL=1 no LDS-DMA after the first chunk, 2 fragments read once per chunk, 3 both.  Use with DISCO_HIP_LIB=<path>.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/disentangledcolorization_amd/csrc
mkdir -p $C/build/ab
python -m disentangledcolorization_amd.build
for v in 1 2 3; do
  (/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Wno-unused-result -DMX_ABL=$v -c $C/conv_mx.hip -o $C/build/ab/conv_mx_$v.o &&
   objs=$(ls $C/build/*.o | grep -v conv_mx.o) &&
   /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/build/ab/libdisco_abl$v.so $objs $C/build/ab/conv_mx_$v.o && echo built abl$v) &
done
wait
