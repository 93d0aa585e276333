// Shared pieces of the split-operand matrix-core kernels (conv3d_x3.hip: persistent z-marching blocks; conv3d_deep.hip: the deep
// U-Net levels): vector types, the power-of-two scale of a bound, the fp16-pair split of four activations, the MFMA wrapper.
#pragma once
#include "common.h"

namespace rcmvs {

typedef __bf16 x3_bf16x8 __attribute__((ext_vector_type(8)));
typedef float x3_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int x3_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int x3_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned char x3_byte;

// power-of-two scale of a tensor bound: s = 2^e with s * m in [2^14, 2^15); 1 for m = 0, denormal or non-finite bounds.
// The same function serves the weights (pack time) and the activations (every launch, from the caller's bound).
__device__ __forceinline__ float x3_pow2_scale(float m, float& inv) {
    const int ex = (int)((__float_as_uint(m) >> 23) & 0xffu);
    int e = (ex == 0 || ex == 255) ? 0 : 14 - (ex - 127);
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    inv = __uint_as_float((unsigned)(127 - e) << 23);
    return __uint_as_float((unsigned)(127 + e) << 23);
}

// four fp32 -> the three bf16 piece quadruples (two dwords each).  18 VALU operations: the packing v_perm_b32 takes the high
// halves (= truncation), the remainders use packed subtractions.  The producers' VALU work is NOT hidden behind the consumers'
// MFMAs -- a wave issuing MFMAs back to back leaves a second wave on its SIMD ~10 % of the VALU issue rate (measured,
// tools/dev/coissue.hip) -- so every instruction here is paid for in matrix-pipe idle time.
typedef float x3_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void x3_split4(x3_f32x4 v, x3_u32x2& h, x3_u32x2& m, x3_u32x2& l) {
    const x3_u32x4 vb = __builtin_bit_cast(x3_u32x4, v);
#ifdef X3_FAKE_SPLIT       // timing experiment only (tools/dev): no split arithmetic, WRONG results -- the ceiling of a pre-split activation format
    h.x = vb[0]; h.y = vb[1]; m.x = vb[2]; m.y = vb[3]; l.x = vb[0]; l.y = vb[2];
    return;
#endif
    const x3_u32x4 hb = vb & 0xffff0000u;
    const x3_f32x4 r1 = v - __builtin_bit_cast(x3_f32x4, hb);                       // exact
    const x3_u32x4 r1b = __builtin_bit_cast(x3_u32x4, r1);
    const x3_u32x4 mb = r1b & 0xffff0000u;
    const x3_f32x4 r2 = r1 - __builtin_bit_cast(x3_f32x4, mb);                      // exact, <= 8 significant bits
    const x3_u32x4 lb = __builtin_bit_cast(x3_u32x4, r2);
    // v_perm_b32: (hi16 of b) << 16 | (hi16 of a)
    h.x = __builtin_amdgcn_perm(vb[1], vb[0], 0x07060302u); h.y = __builtin_amdgcn_perm(vb[3], vb[2], 0x07060302u);
    m.x = __builtin_amdgcn_perm(r1b[1], r1b[0], 0x07060302u); m.y = __builtin_amdgcn_perm(r1b[3], r1b[2], 0x07060302u);
    l.x = __builtin_amdgcn_perm(lb[1], lb[0], 0x07060302u); l.y = __builtin_amdgcn_perm(lb[3], lb[2], 0x07060302u);
}

// four fp32 (already multiplied by the power-of-two scale) -> the two fp16 piece quadruples: h = rne(xs), l = rne(xs - h).
// 2 v_cvt_pk_f16_f32 + 4 v_cvt_f32_f16 + 2 v_pk_add_f32 + 2 v_cvt_pk_f16_f32 (+ the 2 v_pk_mul_f32 of the scale at the call site)
typedef _Float16 x3_f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 x3_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void x3_split4h(x3_f32x4 xs, x3_u32x2& h, x3_u32x2& l) {
    const x3_f16x4 hh = __builtin_convertvector(xs, x3_f16x4);
    const x3_f32x4 r = xs - __builtin_convertvector(hh, x3_f32x4);                  // exact
    const x3_f16x4 ll = __builtin_convertvector(r, x3_f16x4);
    h = __builtin_bit_cast(x3_u32x2, hh);
    l = __builtin_bit_cast(x3_u32x2, ll);
}

template <int NP>
__device__ __forceinline__ x3_f32x4 x3_mfma(x3_u32x4 a, x3_u32x4 b, x3_f32x4 c) {
    if constexpr (NP == 3) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(x3_bf16x8, a), __builtin_bit_cast(x3_bf16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(x3_f16x8, a), __builtin_bit_cast(x3_f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ float x3_absmax4(float m, x3_f32x4 v) {
    return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}

}  // namespace rcmvs
