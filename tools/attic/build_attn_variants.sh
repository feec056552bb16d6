#!/bin/bash
# Diagnostic / variant builds of attention_mfma.hip (attention_mfma_kernel): tools/build/libdisco_attn_<name>.so, used through DISCO_HIP_LIB
#   abl1 / abl2 / abl3: AM_ABL (no P V MFMAs / no softmax arithmetic / both): timing only, results wrong
#   novf: without -mllvm -amdgpu-mfma-vgpr-form (MFMA results in AGPRs: a v_accvgpr_read per score)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/disentangledcolorization_amd/csrc
mkdir -p $C/build/ab $R/tools/build
python -m disentangledcolorization_amd.build
objs=$(ls $C/build/*.o | grep -v "/attention_mfma.o")
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Wno-unused-result"
for v in "$@"; do
  case $v in
    novf) extra="" ;;
    abl*) extra="-mllvm -amdgpu-mfma-vgpr-form -DAM_ABL=${v#abl}" ;;
    kpt*) extra="-mllvm -amdgpu-mfma-vgpr-form -DAM_KPT=${v#kpt}" ;;
    *) extra="$EXTRA" ;;
  esac
  (/opt/rocm/bin/hipcc $FL $extra -c $C/attention_mfma.hip -o $C/build/ab/attention_mfma_$v.o &&
   /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/build/libdisco_attn_$v.so $objs $C/build/ab/attention_mfma_$v.o && echo built $v) &
done
wait
