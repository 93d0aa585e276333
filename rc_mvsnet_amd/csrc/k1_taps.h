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

// one (pixel, plane, view): integer coordinates of the north-west tap (-4 for non-finite positions) + masked weights.
// Shared by every K1 kernel so that all of them sample at bit-identical positions.
__device__ __forceinline__ void k1_chain(float rx, float ry, float rz, float t0, float t1, float t2, float d,
                                         const K1Geom& g, int& xi, int& yi, v4f& wt) {
#pragma clang fp contract(off)
    const float px = rx * d + t0, py = ry * d + t1, pz = rz * d + t2;
    const float rpz = rcp_nr(pz);
    const float u = div_c<2>(px, pz, rpz), vv = div_c<2>(py, pz, rpz);
    const float gx = div_c<1>(u, g.half_w, g.r_half_w) - 1.0f;
    const float gy = div_c<1>(vv, g.half_h, g.r_half_h) - 1.0f;
    const float ix = ((gx + 1.0f) * 0.5f) * g.wm1;
    const float iy = ((gy + 1.0f) * 0.5f) * g.hm1;
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
