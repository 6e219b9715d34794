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
    // (s_nop 1: `lo` usually feeds a matrix instruction, which must not read a VGPR fewer than 2 wait states behind its VALU write; the compiler
    // pads its own VALU instructions but does not look at an inline-asm producer - tools/isa_check.py found seven such pairs in the library)
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\ts_nop 1" : "+v"(l) : "v"(v1), "v"(hi));
    lo = l;
}

// ---- VGPR-form matrix instructions by hand (k_s3f.hip, k_bnh.hip)
// A kernel whose big accumulator fills the 256 AGPRs has a 512-register budget, and with that budget the compiler selects the AGPR form for
// EVERY matrix instruction of the function: the small accumulators next to the big one would have to share its 256 registers (it shuffled
// them through v_accvgpr moves and spilled 160 registers).  These are therefore written as VGPR-form instructions by hand.  The compiler's
// hazard recogniser does not look inside inline assembly, so the wait states are here:
//   - VALU write of a source register -> matrix instruction: 2 wait states (found the hard way: a fragment register zeroed by v_mov one
//     instruction earlier was read stale, at random pixels) - every instruction carries its own s_nop 1;
//   - consecutive instructions on one accumulator (same opcode, destination = source C): none (hardware interlock);
//   - VALU read of the result: 11 wait states behind an 8-pass instruction - mfma_v_fence() (20) closes every sequence.
__device__ __forceinline__ void mfma_v0(f32x16& c, half8 a, half8 b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_v(f32x16& c, half8 a, half8 b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_v_fence(f32x16& c0, f32x16& c1) {
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(c0), "+v"(c1));
}
__device__ __forceinline__ void mfma_v_fence4(f32x16& c0, f32x16& c1, f32x16& c2, f32x16& c3) {
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
}
// One accumulator register -> VGPR.  Left to the compiler, the first VALU use of an accumulator tile copies all 16 registers of the tile to
// VGPRs at once - and it hoists the copies of all sixteen tiles (256 registers) to the top of the epilogue.
__device__ __forceinline__ float acc_rd(float v) {
    float o;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(o) : "a"(v));
    return o;
}

// ---- encoder conv on the fly (k_bnh.hip, k_s3f.hip): a0 tiles are rebuilt on the matrix cores from spectrogram patches
// Conv2d(2 -> 256, 3x3, 'same', no bias), TDAVNet/encoder.py:146-157.  GEMM view: K = 32 slots, slot (h, j) of K step 0 = tap j = dt*3 + df
// (j < 8) of input channel h (re / im), slot (h, 0) of step 1 = tap 8; the other slots of step 1 are zero.  The A operand is the image
// written by enc_stats_kernel (k_stft.hip); the B operand is built here from nine unaligned 8-byte loads per lane (this lane's two adjacent
// pixels at the nine window positions of ITS channel h), masked at the borders of the (T, F) plane.
struct PatchFrag {
    half8 h[2][2], l[2][2];  // [K step][pixel slot]
};
// sp: descriptor of this mixture's (2, T, F) spectrogram with num_records = its exact size (reads outside return 0, they are masked anyway)
// (+ 4 bytes: the pair that starts at the mixture's last float is then wholly in range - a pair that is partly out of range reads as zero as
// a whole; the extra float belongs to the next workspace tensor and only ever lands in a masked tap)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t spec_rsrc(const float* spec_b, int P) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(spec_b), 0, 2 * P * 4 + 4, 0x00020000);
}
__device__ __forceinline__ void patch_load(__amdgpu_buffer_rsrc_t sp, int p0, int h, int P, int F, f32x2 (&v)[9]) {
    const int base = h * P + p0;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        // offset -1 (the first pixels of the mixture: the pair's second float is spec[0], a live tap): read the pair at 0 and shift it
        const int o = base + (j / 3 - 1) * F + (j % 3 - 1);
        const bool m1 = o == -1;
        const f32x2 t = ld2(sp, (unsigned)(m1 ? 0 : o) * 4u, 0);
        v[j] = m1 ? f32x2{0.f, t.x} : t;
    }
}
// esc: power of two that brings the mixture's rms(a0) near 1 (the f16 hi / lo split keeps its low part only for |x| >= 2^-3 or so)
__device__ __forceinline__ void patch_build(const f32x2 (&v)[9], int p0, int T, int F, int P, float esc, PatchFrag& o) {
    const int t0 = p0 / F, f0 = p0 - t0 * F;
    const bool wrap = f0 + 1 == F;
    const int t1 = wrap ? t0 + 1 : t0, f1 = wrap ? 0 : f0 + 1;
    const bool in0 = p0 < P, in1 = p0 + 1 < P;
    float y0[9], y1[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int dt = j / 3, df = j % 3;
        const bool ok0 = in0 && (dt != 0 || t0 >= 1) && (dt != 2 || t0 <= T - 2) && (df != 0 || f0 >= 1) && (df != 2 || f0 <= F - 2);
        const bool ok1 = in1 && (dt != 0 || t1 >= 1) && (dt != 2 || t1 <= T - 2) && (df != 0 || f1 >= 1) && (df != 2 || f1 <= F - 2);
        y0[j] = ok0 ? v[j].x * esc : 0.f;
        y1[j] = ok1 ? v[j].y * esc : 0.f;
    }
    unsigned h0[4], l0[4], h1[4], l1[4];
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
        split2(y0[2 * jp], y0[2 * jp + 1], h0[jp], l0[jp]);
        split2(y1[2 * jp], y1[2 * jp + 1], h1[jp], l1[jp]);
    }
    o.h[0][0] = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(h0));
    o.l[0][0] = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(l0));
    o.h[0][1] = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(h1));
    o.l[0][1] = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(l1));
    unsigned a0[4] = {0, 0, 0, 0}, b0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0}, b1[4] = {0, 0, 0, 0};
    split2(y0[8], 0.f, a0[0], b0[0]);
    split2(y1[8], 0.f, a1[0], b1[0]);
    o.h[1][0] = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(a0));
    o.l[1][0] = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(b0));
    o.h[1][1] = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(a1));
    o.l[1][1] = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(b1));
}
// power of two nearest 1 / rms(a0) of a mixture from its (sum, sumsq) statistics, and 2^-8 / that (exponent arithmetic on the bits, wave-uniform)
__device__ __forceinline__ void rms_pow2(const double* st, double inv_count, float& esc, float& eisc) {
    const float ms = (float)(st[1] * inv_count);
    const int eb = (int)((__float_as_uint(ms) >> 23) & 0xFF) - 127;  // floor(log2(ms)); ms = 0 or denormal -> -127
    int e = -(eb >> 1);
    e = e < -40 ? -40 : (e > 40 ? 40 : e);
    esc = __uint_as_float((unsigned)(127 + e) << 23);
    eisc = __uint_as_float((unsigned)(127 - e - 8) << 23);
}

}  // namespace
