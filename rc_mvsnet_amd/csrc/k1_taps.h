// Shared device code of the K1 family (forward: warp_variance.hip, backward: warp_variance_bwd.hip):
// the per-(pixel, plane, view) coordinate chain and its exact-division helpers.  Forward and backward
// MUST sample at identical positions, so both include this one definition.
#pragma once
#include "common.h"

namespace rcmvs {

typedef float v4f __attribute__((ext_vector_type(4)));

// v_rcp_f32 refined by one Newton step (~0.5 ulp)
__device__ __forceinline__ float rcp_nr(float b) {
    float r = __builtin_amdgcn_rcpf(b);
    float e = fmaf(-b, r, 1.0f);
    return fmaf(e, r, r);
}

typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

template <int NCORR>
__device__ __forceinline__ float div_c(float a, float b, float r) {
    float q = a * r;
    float rem = fmaf(-b, q, a);
    q = fmaf(rem, r, q);
    if (NCORR >= 2) { rem = fmaf(-b, q, a); q = fmaf(rem, r, q); }
    return q;
}

struct K1Geom {
    float wm1, hm1, half_w, half_h, r_half_w, r_half_h;
    int w, h;
};

// one (pixel, plane, view): the sampling position in source pixels, in the reference's operation order (models/modules.py:326-333
// and grid_sample's align_corners un-normalisation), every division correctly rounded.  The position is what decides the tap pair,
// so every K1 kernel -- forward, backward, gather or window path -- takes it from here and samples at bit-identical positions.
__device__ __forceinline__ void k1_position(float rx, float ry, float rz, float t0, float t1, float t2, float d,
                                            const K1Geom& g, float& ix, float& iy) {
#pragma clang fp contract(off)
    const float px = rx * d + t0, py = ry * d + t1, pz = rz * d + t2;
    const float rpz = rcp_nr(pz);
    const float u = div_c<2>(px, pz, rpz), vv = div_c<2>(py, pz, rpz);
    const float gx = div_c<1>(u, g.half_w, g.r_half_w) - 1.0f;
    const float gy = div_c<1>(vv, g.half_h, g.r_half_h) - 1.0f;
    ix = ((gx + 1.0f) * 0.5f) * g.wm1;
    iy = ((gy + 1.0f) * 0.5f) * g.hm1;
}

// integer coordinates of the north-west tap (-4 for non-finite positions) + masked weights
__device__ __forceinline__ void k1_chain(float rx, float ry, float rz, float t0, float t1, float t2, float d,
                                         const K1Geom& g, int& xi, int& yi, v4f& wt) {
#pragma clang fp contract(off)
    float ix, iy;
    k1_position(rx, ry, rz, t0, t1, t2, d, g, ix, iy);
    const float x0 = floorf(ix), y0 = floorf(iy);
    const float wx1 = ix - x0, wx0 = (x0 + 1.0f) - ix;
    const float wy1 = iy - y0, wy0 = (y0 + 1.0f) - iy;
    // |coordinate| < 2^24: exact int conversion; false for NaN / inf
    const bool fin = (fabsf(ix) < 16777216.0f) && (fabsf(iy) < 16777216.0f);
    xi = fin ? (int)x0 : -4;
    yi = fin ? (int)y0 : -4;
    const bool vx0 = (unsigned)xi < (unsigned)g.w, vx1 = (unsigned)(xi + 1) < (unsigned)g.w;
    const bool vy0 = (unsigned)yi < (unsigned)g.h, vy1 = (unsigned)(yi + 1) < (unsigned)g.h;
    wt.x = (vx0 && vy0) ? wx0 * wy0 : 0.0f;
    wt.y = (vx1 && vy0) ? wx1 * wy0 : 0.0f;
    wt.z = (vx0 && vy1) ? wx0 * wy1 : 0.0f;
    wt.w = (vx1 && vy1) ? wx1 * wy1 : 0.0f;
}

// The same taps in FIXED-PATTERN form: the 2x2 footprint is moved inside the image -- (xc, yc) in [0, w-2] x [0, h-2] is its
// north-west corner -- and the four weights follow the texels: a footprint that hangs over a border by one texel keeps its inside
// taps' weights on the texels they belong to and gives the two texels it never had a weight of zero; a footprint wholly outside, or
// a non-finite position, gets four zeros (live = false: it may be pointed anywhere).  The four texels are then always
// base, base + 1 texel, base + 1 row, base + 1 row + 1 texel: one offset per record, the other three are immediates of the load.
// The products are the ones k1_chain forms (same operands, same roundings).
__device__ __forceinline__ void k1_tap_fixed(float ix, float iy, const K1Geom& g, int& xc, int& yc, v4f& wt, bool& live) {
#pragma clang fp contract(off)
    const float x0 = floorf(ix), y0 = floorf(iy);
    const float wx1 = ix - x0, wx0 = (x0 + 1.0f) - ix;
    const float wy1 = iy - y0, wy0 = (y0 + 1.0f) - iy;
    const bool fin = (fabsf(ix) < 16777216.0f) && (fabsf(iy) < 16777216.0f);
    const int xi = fin ? (int)x0 : -4, yi = fin ? (int)y0 : -4;
    const bool inx = (unsigned)xi < (unsigned)(g.w - 1), iny = (unsigned)yi < (unsigned)(g.h - 1);
    const float xa = inx ? wx0 : (xi == -1 ? wx1 : 0.0f), xb = inx ? wx1 : (xi == g.w - 1 ? wx0 : 0.0f);
    const float ya = iny ? wy0 : (yi == -1 ? wy1 : 0.0f), yb = iny ? wy1 : (yi == g.h - 1 ? wy0 : 0.0f);
    xc = min(max(xi, 0), g.w - 2);
    yc = min(max(yi, 0), g.h - 2);
    live = ((unsigned)(xi + 1) < (unsigned)(g.w + 1)) && ((unsigned)(yi + 1) < (unsigned)(g.h + 1));
    wt.x = xa * ya; wt.y = xb * ya; wt.z = xa * yb; wt.w = xb * yb;
}

// offsets (bytes, clamped in-bounds, view row base included) + weights
template <int C>
__device__ __forceinline__ void k1_tap(float rx, float ry, float rz, float t0, float t1, float t2, float d,
                                       const K1Geom& g, int vrow, v4i& o, v4f& wt) {
    int xi, yi;
    k1_chain(rx, ry, rz, t0, t1, t2, d, g, xi, yi, wt);
    const int xc0 = min(max(xi, 0), g.w - 1), xc1 = min(max(xi + 1, 0), g.w - 1);
    const int row0 = min(max(yi, 0), g.h - 1) * g.w + vrow, row1 = min(max(yi + 1, 0), g.h - 1) * g.w + vrow;
    o.x = (row0 + xc0) * (C * 4); o.y = (row0 + xc1) * (C * 4);
    o.z = (row1 + xc0) * (C * 4); o.w = (row1 + xc1) * (C * 4);
}

// min / max over the wave with DPP row shifts + four readlanes (no LDS traffic, no atomics); result is wave-uniform
template <bool MIN>
__device__ __forceinline__ int wave_reduce_i32(int v) {
#define RCMVS_DPP_STEP(CTRL) { const int t = __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); v = MIN ? min(v, t) : max(v, t); }
    RCMVS_DPP_STEP(0x111) RCMVS_DPP_STEP(0x112) RCMVS_DPP_STEP(0x114) RCMVS_DPP_STEP(0x118)     // row_shr 1, 2, 4, 8
#undef RCMVS_DPP_STEP
    const int a = __builtin_amdgcn_readlane(v, 15), b = __builtin_amdgcn_readlane(v, 31);
    const int c = __builtin_amdgcn_readlane(v, 47), d = __builtin_amdgcn_readlane(v, 63);
    return MIN ? min(min(a, b), min(c, d)) : max(max(a, b), max(c, d));
}

}  // namespace rcmvs
