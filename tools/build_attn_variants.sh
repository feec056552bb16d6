#!/bin/bash
# Diagnostic / variant builds of tokens.hip (attention_mfma_kernel): tools/build/libdisco_attn_<name>.so, used through DISCO_HIP_LIB
#   abl1 / abl2 / abl4 / abl3 / abl6: AM_ABL (no P V MFMAs / no softmax arithmetic / no K Q^T MFMAs / combinations): timing only
#   vf: -mllvm -amdgpu-mfma-vgpr-form (MFMA results in VGPRs: no v_accvgpr_read per score)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/disentangledcolorization_amd/csrc
mkdir -p $C/build/ab $R/tools/build
python -m disentangledcolorization_amd.build
objs=$(ls $C/build/*.o | grep -v "/tokens.o")
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Wno-unused-result"
for v in "$@"; do
  case $v in
    vf) extra="-mllvm -amdgpu-mfma-vgpr-form" ;;
    vfabl*) extra="-mllvm -amdgpu-mfma-vgpr-form -DAM_ABL=${v#vfabl}" ;;
    abl*) extra="-DAM_ABL=${v#abl}" ;;
    *) extra="$EXTRA" ;;
  esac
  (/opt/rocm/bin/hipcc $FL $extra -c $C/tokens.hip -o $C/build/ab/tokens_$v.o &&
   /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/build/libdisco_attn_$v.so $objs $C/build/ab/tokens_$v.o && echo built $v) &
done
wait
