#!/bin/bash
# Diagnostic builds of libdisco_hip.so with parts of conv3x3_mx_kernel switched off (results are WRONG by design; timing only):
#   MX_ABL bits: 1 no LDS-DMA after the first chunk, 2 fragments read once per chunk, 4 no PIXEL pieces after a workgroup's first chunk,
#   8 no WEIGHT pieces, 16 no output stores, 32 no epilogue at all, 64 the latency loop on half of a long chain.  Builds the f16x3 translation unit (conv_mx_ar2.hip) per requested value
#   (and, for values with bit 32 or 64, the f16+fp6x2 one, conv_mx_ar3.hip, as well):
#       bash tools/build_ablations.sh 4 16        ->  tools/build/libdisco_abl{4,16}.so   (use with DISCO_HIP_LIB=<path>; tools/build/ travels with gpurun, csrc/build/ does not)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/disentangledcolorization_amd/csrc
mkdir -p $C/build/ab
python -m disentangledcolorization_amd.build
for v in "${@:-4 16}"; do
  (/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Wno-unused-result -DMX_ABL=$v -c $C/conv_mx_ar2.hip -o $C/build/ab/conv_mx_ar2_$v.o &&
   objs=$(ls $C/build/*.o | grep -v conv_mx_ar2.o) && extra= &&
   if [ $(( v & 96 )) -ne 0 ]; then
     /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Wno-unused-result -DMX_ABL=$v -c $C/conv_mx_ar3.hip -o $C/build/ab/conv_mx_ar3_$v.o &&
     objs=$(echo "$objs" | grep -v conv_mx_ar3.o) && extra=$C/build/ab/conv_mx_ar3_$v.o; fi &&
   /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/build/libdisco_abl$v.so $objs $C/build/ab/conv_mx_ar2_$v.o $extra && echo built abl$v) &
done
wait
