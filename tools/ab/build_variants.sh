#!/bin/bash
# A/B builds of libvilattn.so with different tuning switches (see the #ifndef blocks in csrc/vil_attn_mfma*.hip)
# (whole-library single compile: -fno-honor-nans, which build() applies to the forward file only, is NOT set here;
#  pass it in the flags when the forward kernel is what is being compared)
# usage: tools/ab/build_variants.sh name "-DFLAG=.. -DFLAG=.." [name flags ...]
cd "$(dirname "$0")/../../vision-longformer_amd/csrc" || exit 1
SRCS="vil_attn_api.hip vil_attn_scalar.hip vil_attn_mfma.hip vil_attn_mfma_bwd.hip vil_attn_mfma_f32.hip vil_attn_dense.hip vil_layernorm.hip vil_attn_glo.hip vil_colsum.hip vil_wgrad.hip vil_gemm.hip vil_gemm_fused.hip vil_gemm_skinny.hip vil_sc2d_op.hip vil_patchify.hip vil_optim.hip"
while [ $# -ge 2 ]; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 $2 \
      -shared -o ../../tools/ab/libvilattn_$1.so $SRCS -L/opt/rocm/lib -lhipblaslt 2>&1 | grep -E "error|spill" &
  shift 2
done
wait
