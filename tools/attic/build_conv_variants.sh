#!/bin/bash
# A/B builds of the conv kernel with the DEFERRED EPILOGUE on the main tile (conv_mx_kernel.h, MX_DEFER): tools/build/libdisco_conv_<name>.so,
# used through DISCO_HIP_LIB.   bash tools/build_conv_variants.sh defer "defer_s0 -DMX_DEFER_SLOT0=0 -DMX_DEFER_STEP=1" ...
#   each argument: "<name> [extra compiler flags]"; -DMX_DEFER=1 is always set; the f16x3 and f16+fp6x2 translation units are rebuilt
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/disentangledcolorization_amd/csrc
mkdir -p $C/build/ab $R/tools/build
python -m disentangledcolorization_amd.build
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Wno-unused-result -DMX_DEFER=1"
for spec in "$@"; do
  set -- $spec; v=$1; shift; extra="$*"
  (for ar in 2 3; do /opt/rocm/bin/hipcc $FL $extra -c $C/conv_mx_ar$ar.hip -o $C/build/ab/conv_mx_ar${ar}_$v.o & done; wait
   objs=$(ls $C/build/*.o | grep -v "conv_mx_ar2.o\|conv_mx_ar3.o")
   /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/build/libdisco_conv_$v.so $objs $C/build/ab/conv_mx_ar2_$v.o $C/build/ab/conv_mx_ar3_$v.o && echo built $v) &
done
wait
