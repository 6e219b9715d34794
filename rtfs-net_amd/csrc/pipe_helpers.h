// Device helpers shared by the pipelined pointwise kernels on padded channel rows (k_b2b.hip, k_s3f.hip).
#pragma once
#include "common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2a __attribute__((ext_vector_type(2), aligned(8)));
typedef _Float16 half2_ __attribute__((ext_vector_type(2)));

// Two adjacent floats through a raw buffer descriptor: address = descriptor base (this wave's first pixel of the sample, wave-uniform) +
// lane byte offset (one VGPR for the whole kernel) + row byte offset (an SGPR: one s_add per access).  The flat global_load form needs a
// 64-bit scalar base per channel row: 64 SGPR pairs per tile, which the first version of this kernel spilled (276 v_readlane per two tiles).
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const float* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ f32x2 ld2(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0));
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ half8 ld_h8(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {  // one 16-byte fragment piece (weight images)
    return __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}
__device__ __forceinline__ void st2(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, f32x2 v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rs, voff, soff, 0);
}

// f16 hi / lo split of two values: hi = RNE(v) (one v_cvt_pk_f16_f32), lo = (f16)(v - hi) (one v_fma_mix per value)
__device__ __forceinline__ void split2(float v0, float v1, unsigned& hi, unsigned& lo) {
    const half2_ h = __builtin_convertvector(f32x2{v0, v1}, half2_);
    hi = __builtin_bit_cast(unsigned, h);
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(v0), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(v1), "v"(hi));
    lo = l;
}

}  // namespace
