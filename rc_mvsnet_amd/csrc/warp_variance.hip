// K1: fused plane-sweep homography warp + variance cost volume (gfx950).
//
// Replaces, per stage, (V-1) x homo_warping (models/modules.py:304-339: ~15 materialised
// [B,3,D,hw] intermediates + grid_sample) and the sum / square-sum / variance chain of
// DepthNet_eval.forward (models/casmvsnet.py:257-288).  One launch reads the V feature maps and
// the plane table and writes the variance volume exactly once.
//
// Mapping (wave = 64 lanes): channels-last everywhere.  A pixel's C channels are handled by
// C/4 adjacent lanes (one float4 each), a wave covers 1024 B of contiguous output per plane
// (256/C pixels of one image row) and a 256-thread block covers a 4-row tile x 8 (4 for C=8)
// planes, so every bilinear tap is a 16-byte load of 4 channels and the output store of a wave
// is one fully coalesced 1 KiB line.  See warp_variance_tp_kernel for the two-phase structure.
//
// Numerics: the coordinate chain and the accumulation follow the operation order of
// oracle/warp.py (itself the reference's op order) with fp contraction OFF and correctly rounded
// divisions, so the kernel is bit-comparable with the oracle; taps outside the source image, and
// non-finite coordinates (z == 0), contribute zero (grid_sample zeros padding, CUDA/HIP semantics).
#include <type_traits>
#include "common.h"
#include "k1_taps.h"
#include "k1_win.h"
#include "k1_pp.h"

namespace rcmvs {

constexpr int DK = 8;   // planes per thread

struct WarpCoord {
    int off[4];     // element offsets (pixel index * C) of the 4 taps, clamped in-bounds
    float wgt[4];   // tap weights, 0 where the tap is outside the image
};

__device__ __forceinline__ WarpCoord warp_taps(float rx, float ry, float rz, float tx, float ty, float tz,
                                               float d, float half_w, float half_h, float wm1, float hm1,
                                               int w, int h, int C) {
#pragma clang fp contract(off)
    // models/modules.py:326-331, then grid_sample's align_corners=True un-normalisation
    float px = rx * d + tx;
    float py = ry * d + ty;
    float pz = rz * d + tz;
    float u = px / pz;
    float v = py / pz;
    float gx = u / half_w - 1.0f;
    float gy = v / half_h - 1.0f;
    float ix = ((gx + 1.0f) / 2.0f) * wm1;
    float iy = ((gy + 1.0f) / 2.0f) * hm1;
    float x0 = floorf(ix), y0 = floorf(iy);
    float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
    float wx1 = ix - x0, wx0 = x1 - ix;
    float wy1 = iy - y0, wy0 = y1 - iy;
    // validity in the float domain (NaN / inf compare false)
    bool vx0 = (x0 >= 0.0f) && (x0 <= wm1);
    bool vx1 = (x1 >= 0.0f) && (x1 <= wm1);
    bool vy0 = (y0 >= 0.0f) && (y0 <= hm1);
    bool vy1 = (y1 >= 0.0f) && (y1 <= hm1);
    // clamp before the int conversion so huge / non-finite values cannot overflow
    int xi0 = (int)fminf(fmaxf(x0, 0.0f), wm1);
    int xi1 = (int)fminf(fmaxf(x1, 0.0f), wm1);
    int yi0 = (int)fminf(fmaxf(y0, 0.0f), hm1);
    int yi1 = (int)fminf(fmaxf(y1, 0.0f), hm1);
    WarpCoord t;
    t.off[0] = (yi0 * w + xi0) * C;
    t.off[1] = (yi0 * w + xi1) * C;
    t.off[2] = (yi1 * w + xi0) * C;
    t.off[3] = (yi1 * w + xi1) * C;
    t.wgt[0] = (vx0 && vy0) ? wx0 * wy0 : 0.0f;
    t.wgt[1] = (vx1 && vy0) ? wx1 * wy0 : 0.0f;
    t.wgt[2] = (vx0 && vy1) ? wx0 * wy1 : 0.0f;
    t.wgt[3] = (vx1 && vy1) ? wx1 * wy1 : 0.0f;
    return t;
}

__device__ __forceinline__ float4 bilerp4(const float* __restrict__ src, const WarpCoord& t, int q4) {
#pragma clang fp contract(off)
    float4 a = *reinterpret_cast<const float4*>(src + t.off[0] + q4);
    float4 b = *reinterpret_cast<const float4*>(src + t.off[1] + q4);
    float4 c = *reinterpret_cast<const float4*>(src + t.off[2] + q4);
    float4 d = *reinterpret_cast<const float4*>(src + t.off[3] + q4);
    float4 r;
    r.x = ((a.x * t.wgt[0] + b.x * t.wgt[1]) + c.x * t.wgt[2]) + d.x * t.wgt[3];
    r.y = ((a.y * t.wgt[0] + b.y * t.wgt[1]) + c.y * t.wgt[2]) + d.y * t.wgt[3];
    r.z = ((a.z * t.wgt[0] + b.z * t.wgt[1]) + c.z * t.wgt[2]) + d.z * t.wgt[3];
    r.w = ((a.w * t.wgt[0] + b.w * t.wgt[1]) + c.w * t.wgt[2]) + d.w * t.wgt[3];
    return r;
}

// Reference-order kernel (debug variant 2): every lane runs the full coordinate chain for its own
// (plane, view) with the compiler's IEEE division -- the straightforward transcription of
// oracle/warp.py that the production kernel below is checked against bit for bit.
// STORE_ONLY (debug variant 3) is a profiling ablation: no warp, just the output stream.
template <int C, bool STORE_ONLY>
__global__ __launch_bounds__(256) void warp_variance_ref_kernel(
    const float* __restrict__ feats, const float* __restrict__ rot, const float* __restrict__ trans,
    const float* __restrict__ planes, float* __restrict__ var, int V, int D, int h, int w, int tiles_x, int tiles_y) {
#pragma clang fp contract(off)
    constexpr int LPP = C / 4;          // lanes per pixel
    constexpr int TW = 256 / C;         // pixels per wave = tile width  (1 KiB of output per plane)
    constexpr int TH = 4;               // one wave per tile row
    const int b = blockIdx.z;
    const int k0 = blockIdx.y * DK;
    const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int q4 = (threadIdx.x % LPP) * 4;
    const int x = tx * TW + (threadIdx.x / LPP) % TW;
    const int y = ty * TH + threadIdx.x / (LPP * TW);
    if (x >= w || y >= h) return;
    const long long hw = (long long)h * w;
    const float fx = (float)x, fy = (float)y;
    const float wm1 = (float)(w - 1), hm1 = (float)(h - 1);
    const float half_w = wm1 / 2.0f, half_h = hm1 / 2.0f;
    const float2 pl = reinterpret_cast<const float2*>(planes)[(long long)b * hw + (long long)y * w + x];
    const float* fb = feats + (long long)b * V * hw * C;
    const float4 ref = *reinterpret_cast<const float4*>(fb + ((long long)y * w + x) * C + q4);
    float4 s[DK], sq[DK];
#pragma unroll
    for (int k = 0; k < DK; ++k) {
        s[k] = ref;
        sq[k] = make_float4(ref.x * ref.x, ref.y * ref.y, ref.z * ref.z, ref.w * ref.w);
    }
    for (int v = 1; v < V && !STORE_ONLY; ++v) {
        const float* r = rot + ((long long)b * (V - 1) + (v - 1)) * 9;
        const float* t = trans + ((long long)b * (V - 1) + (v - 1)) * 3;
        const float rx = (r[0] * fx + r[1] * fy) + r[2];
        const float ry = (r[3] * fx + r[4] * fy) + r[5];
        const float rz = (r[6] * fx + r[7] * fy) + r[8];
        const float t0 = t[0], t1 = t[1], t2 = t[2];
        const float* src = fb + (long long)v * hw * C;
#pragma unroll
        for (int k = 0; k < DK; ++k) {
            const float d = pl.x + (float)(k0 + k) * pl.y;
            WarpCoord tc = warp_taps(rx, ry, rz, t0, t1, t2, d, half_w, half_h, wm1, hm1, w, h, C);
            float4 val = bilerp4(src, tc, q4);
            s[k].x = s[k].x + val.x; s[k].y = s[k].y + val.y; s[k].z = s[k].z + val.z; s[k].w = s[k].w + val.w;
            sq[k].x = sq[k].x + val.x * val.x; sq[k].y = sq[k].y + val.y * val.y;
            sq[k].z = sq[k].z + val.z * val.z; sq[k].w = sq[k].w + val.w * val.w;
        }
    }
    const float fV = (float)V;
    float* ob = var + (((long long)b * D) * hw + (long long)y * w + x) * C + q4;
#pragma unroll
    for (int k = 0; k < DK; ++k) {
        if (k0 + k < D) {
            float4 m, o;
            if (STORE_ONLY) { o = s[k]; }
            else {
                m.x = s[k].x / fV; m.y = s[k].y / fV; m.z = s[k].z / fV; m.w = s[k].w / fV;
                o.x = sq[k].x / fV - m.x * m.x;
                o.y = sq[k].y / fV - m.y * m.y;
                o.z = sq[k].z / fV - m.z * m.z;
                o.w = sq[k].w / fV - m.w * m.w;
            }
            *reinterpret_cast<float4*>(ob + (long long)(k0 + k) * hw * C) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------
// K1 (production kernel): two-phase, plane-major, LDS tap table.
//   Phase A  every (pixel, plane, view) of the block's tile is handled by exactly one thread: the
//            coordinate chain runs once (not once per channel lane) and leaves FOUR ready-to-use
//            32-bit byte offsets (view base included) and four masked bilinear weights in LDS as
//            one int4 + one float4 record.  rot*(x,y,1) is computed once per (pixel, view).
//   Phase B  thread = (pixel, channel quad).  Per plane: start from the reference value, add every
//            source view's sample -- tap record broadcast-read from LDS, four 16-byte raw buffer
//            loads (SGPR descriptor + 32-bit offset: no 64-bit address arithmetic) -- form the
//            variance and STORE THE PLANE IMMEDIATELY (non-temporal), so stores are spread over
//            the kernel and few registers stay live.  With a compile-time view count (NVT = 2 for
//            the 3-view DTU setting, 4 for 5 views) the gathers of plane k+1 are issued before
//            plane k is blended (double buffer), which is what hides the L2 latency; NVT = 0 is
//            the general path (any V, views in LDS-sized chunks, sums carried in registers).
// Arithmetic: same operation order as oracle/warp.py, contraction off (FAST = false) -- the result
//   is bit-identical to the reference-order kernel (variant 0) in tests/test_gpu_parity.py.
//   Divisions: a/b for the projective divide = v_rcp_f32 + one Newton step + two fma-residual
//   corrections of the quotient (IEEE-exact for operands away from the exponent limits, without
//   the compiler's div_scale/div_fmas/div_fixup sequence and denormal-mode switches); divisions by
//   the constants (w-1)/2, (h-1)/2 and V use Markstein's single correction.  b == 0 gives NaN and
//   the tap is dropped exactly like the reference's inf coordinate.  FAST = true additionally
//   contracts the bilinear blend and the square-sum into FMAs (<= 2e-7 relative difference).
// ------------------------------------------------------------------------------------------
template <bool FAST>
__device__ __forceinline__ v4f blend4(v4f a, v4f b, v4f c, v4f d, v4f wt) {
    if (FAST) {
        v4f r = a * wt.x;
        r = __builtin_elementwise_fma(b, (v4f){wt.y, wt.y, wt.y, wt.y}, r);
        r = __builtin_elementwise_fma(c, (v4f){wt.z, wt.z, wt.z, wt.z}, r);
        r = __builtin_elementwise_fma(d, (v4f){wt.w, wt.w, wt.w, wt.w}, r);
        return r;
    } else {
#pragma clang fp contract(off)
        return ((a * wt.x + b * wt.y) + c * wt.z) + d * wt.w;
    }
}

template <int NV>
struct K1Fetch {
    v4f t[NV][4];
    v4f w[NV];
};

template <int NV, int DKB, int PIX>
__device__ __forceinline__ void k1_issue(K1Fetch<NV>& f, const v4i* lds_o, const v4f* lds_w, __amdgpu_buffer_rsrc_t rsrc,
                                         int k, int p, int q4b, int v0 = 0) {
#pragma unroll
    for (int va = 0; va < NV; ++va) {
        const int idx = ((v0 + va) * DKB + k) * PIX + p;
        const v4i o = lds_o[idx];
        f.w[va] = lds_w[idx];
        f.t[va][0] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.x + q4b, 0, 0));
        f.t[va][1] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.y + q4b, 0, 0));
        f.t[va][2] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.z + q4b, 0, 0));
        f.t[va][3] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.w + q4b, 0, 0));
    }
}

template <bool FAST>
__device__ __forceinline__ void k1_store_variance(v4f a, v4f a2, float fV, float rV, float* dst) {
#pragma clang fp contract(off)
    v4f m, o;
    m.x = div_c<1>(a.x, fV, rV); m.y = div_c<1>(a.y, fV, rV); m.z = div_c<1>(a.z, fV, rV); m.w = div_c<1>(a.w, fV, rV);
    o.x = div_c<1>(a2.x, fV, rV) - m.x * m.x;
    o.y = div_c<1>(a2.y, fV, rV) - m.y * m.y;
    o.z = div_c<1>(a2.z, fV, rV) - m.z * m.z;
    o.w = div_c<1>(a2.w, fV, rV) - m.w * m.w;
    __builtin_nontemporal_store(o, reinterpret_cast<v4f*>(dst));
}

// Common prologue of the two kernels below: geometry constants, the block's tile and the two thread identities.
template <int C>
struct K1Tile {
    static constexpr int LPP = C / 4;
    static constexpr int PIX = 256 / LPP;          // pixels per block
    static constexpr int TH = 4, TW = PIX / TH;
    static constexpr int GRP = 256 / PIX;          // phase-A threads per pixel
};

// one view's share of phase A for this thread: KPT planes of pixel (xa, ya) -> LDS records [slot][ka][pa]
template <int C, int DKB>
__device__ __forceinline__ void k1_phase_a(const float* __restrict__ rot, const float* __restrict__ trans, int b, int V, int view, int slot,
                                           float fxa, float fya, float2 pla, int k0, int hw, const K1Geom& g, v4i* lds_o, v4f* lds_w) {
#pragma clang fp contract(off)
    constexpr int PIX = K1Tile<C>::PIX, GRP = K1Tile<C>::GRP, KPT = DKB / GRP;
    const int pa = threadIdx.x % PIX, ga = threadIdx.x / PIX;
    const float* r = rot + ((long long)b * (V - 1) + (view - 1)) * 9;
    const float* t = trans + ((long long)b * (V - 1) + (view - 1)) * 3;
    const float rx = (r[0] * fxa + r[1] * fya) + r[2];
    const float ry = (r[3] * fxa + r[4] * fya) + r[5];
    const float rz = (r[6] * fxa + r[7] * fya) + r[8];
    const float t0 = t[0], t1 = t[1], t2 = t[2];
    const int vrow = view * hw;
#pragma unroll
    for (int kk = 0; kk < KPT; ++kk) {
        const int ka = ga + kk * GRP;
        const float d = pla.x + (float)(k0 + ka) * pla.y;
        v4i o;
        v4f wt;
        k1_tap<C>(rx, ry, rz, t0, t1, t2, d, g, vrow, o, wt);
        const int idx = (slot * DKB + ka) * PIX + pa;
        lds_o[idx] = o;
        lds_w[idx] = wt;
    }
}

// compile-time view count (2, 4 or 6 source views): straight-line body.  Kept apart from the runtime-view kernel below because sharing
// one function cost 30-40 VGPRs (128 / 156 / 153 against 98 / 116 / 116 at C = 8 / 16 / 32 with two source views) and 4 % of the time
// (profiles/r4_k1_walls.txt).
template <int C, int DKB, bool FAST, int NVT, int NG = (NVT > 4 ? 3 : NVT)>
__global__ __launch_bounds__(256) void warp_variance_tp_kernel(
    const float* __restrict__ feats, const float* __restrict__ rot, const float* __restrict__ trans,
    const float* __restrict__ planes, float* __restrict__ var, int V, int D, int h, int w, int tiles_x) {
#pragma clang fp contract(off)
    constexpr int LPP = K1Tile<C>::LPP, PIX = K1Tile<C>::PIX, TH = K1Tile<C>::TH, TW = K1Tile<C>::TW, GRP = K1Tile<C>::GRP;
    static_assert(NVT > 0 && DKB % GRP == 0, "DKB must be a multiple of 256/PIX");
    extern __shared__ __attribute__((aligned(16))) v4i lds_o[];          // [NVT][DKB][PIX] offsets, then weights
    v4f* lds_w = reinterpret_cast<v4f*>(lds_o + NVT * DKB * PIX);
    const int b = blockIdx.z;
    const int k0 = blockIdx.y * DKB;
    const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tx0 = (tile % tiles_x) * TW, ty0 = (tile / tiles_x) * TH;
    const int hw = h * w;
    K1Geom g;
    g.w = w; g.h = h;
    g.wm1 = (float)(w - 1); g.hm1 = (float)(h - 1);
    g.half_w = g.wm1 / 2.0f; g.half_h = g.hm1 / 2.0f;
    g.r_half_w = rcp_nr(g.half_w); g.r_half_h = rcp_nr(g.half_h);
    const float* fb = feats + (long long)b * V * hw * C;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fb), (short)0, 0x7fffffff, 0x00020000);
    // ---- phase-B identity
    const int p = threadIdx.x / LPP;
    const int q4b = (threadIdx.x % LPP) * 16;                    // byte offset of this lane's channel quad
    const int x = tx0 + p % TW, y = ty0 + p / TW;
    const bool inside = (x < w) && (y < h);
    v4f ref = (v4f){0.f, 0.f, 0.f, 0.f};
    if (inside) ref = *reinterpret_cast<const v4f*>(fb + ((long long)y * w + x) * C + (q4b >> 2));
    const float fV = (float)V, rV = rcp_nr(fV);
    float* ob = var + (((long long)b * D) * hw + (long long)y * w + x) * C + (q4b >> 2);
    // ---------------- phase A
    {
        const int pa = threadIdx.x % PIX;
        const int xa = min(tx0 + pa % TW, w - 1), ya = min(ty0 + pa / TW, h - 1);
        const float2 pla = reinterpret_cast<const float2*>(planes)[(long long)b * hw + ya * w + xa];
        for (int va = 0; va < NVT; ++va)
            k1_phase_a<C, DKB>(rot, trans, b, V, 1 + va, va, (float)xa, (float)ya, pla, k0, hw, g, lds_o, lds_w);
    }
    __syncthreads();
    if (!inside) return;
    // ---------------- phase B: double-buffered gathers over the flattened (plane, view group) sequence: the group after the current one
    // (same plane or the next) is in flight while the current one is blended.  NG = views per group: all of them for 2 / 4 source
    // views, 3 for the 7-view setting (two full tap sets of 6 views exceed the register file).  Views are accumulated in ascending
    // order whatever the grouping, so the result is bit-identical.
    constexpr int NGRP = NVT / NG, NS = DKB * NGRP;
    static_assert(NVT % NG == 0, "view groups must divide the view count");
    K1Fetch<NG> f0, f1;
    k1_issue<NG, DKB, PIX>(f0, lds_o, lds_w, rsrc, 0, p, q4b, 0);
    v4f a = ref, a2 = ref * ref;
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        const int k = st / NGRP, gi = st % NGRP;
        K1Fetch<NG>& cur = (st & 1) ? f1 : f0;
        K1Fetch<NG>& nxt = (st & 1) ? f0 : f1;
        if (st + 1 < NS) k1_issue<NG, DKB, PIX>(nxt, lds_o, lds_w, rsrc, (st + 1) / NGRP, p, q4b, ((st + 1) % NGRP) * NG);
        if (gi == 0) { a = ref; a2 = ref * ref; }
#pragma unroll
        for (int va = 0; va < NG; ++va) {
            v4f val = blend4<FAST>(cur.t[va][0], cur.t[va][1], cur.t[va][2], cur.t[va][3], cur.w[va]);
            a = a + val;
            if (FAST) a2 = __builtin_elementwise_fma(val, val, a2); else a2 = a2 + val * val;
        }
        if (gi == NGRP - 1 && k0 + k < D) k1_store_variance<FAST>(a, a2, fV, rV, ob + (long long)(k0 + k) * hw * C);
    }
}

// runtime view count (any V): views in LDS-sized chunks of VC, the sums carried in registers across the chunks
template <int C, int DKB, bool FAST>
__global__ __launch_bounds__(256) void warp_variance_mv_kernel(
    const float* __restrict__ feats, const float* __restrict__ rot, const float* __restrict__ trans,
    const float* __restrict__ planes, float* __restrict__ var, int V, int D, int h, int w, int tiles_x, int VC) {
#pragma clang fp contract(off)
    constexpr int LPP = K1Tile<C>::LPP, PIX = K1Tile<C>::PIX, TH = K1Tile<C>::TH, TW = K1Tile<C>::TW, GRP = K1Tile<C>::GRP;
    static_assert(DKB % GRP == 0, "DKB must be a multiple of 256/PIX");
    extern __shared__ __attribute__((aligned(16))) v4i lds_o[];          // [VC][DKB][PIX] offsets, then weights
    v4f* lds_w = reinterpret_cast<v4f*>(lds_o + VC * DKB * PIX);
    const int b = blockIdx.z;
    const int k0 = blockIdx.y * DKB;
    const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tx0 = (tile % tiles_x) * TW, ty0 = (tile / tiles_x) * TH;
    const int hw = h * w;
    K1Geom g;
    g.w = w; g.h = h;
    g.wm1 = (float)(w - 1); g.hm1 = (float)(h - 1);
    g.half_w = g.wm1 / 2.0f; g.half_h = g.hm1 / 2.0f;
    g.r_half_w = rcp_nr(g.half_w); g.r_half_h = rcp_nr(g.half_h);
    const float* fb = feats + (long long)b * V * hw * C;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fb), (short)0, 0x7fffffff, 0x00020000);
    const int p = threadIdx.x / LPP;
    const int q4b = (threadIdx.x % LPP) * 16;
    const int x = tx0 + p % TW, y = ty0 + p / TW;
    const bool inside = (x < w) && (y < h);
    v4f ref = (v4f){0.f, 0.f, 0.f, 0.f};
    if (inside) ref = *reinterpret_cast<const v4f*>(fb + ((long long)y * w + x) * C + (q4b >> 2));
    const float fV = (float)V, rV = rcp_nr(fV);
    float* ob = var + (((long long)b * D) * hw + (long long)y * w + x) * C + (q4b >> 2);
    const int pa = threadIdx.x % PIX;
    const int xa = min(tx0 + pa % TW, w - 1), ya = min(ty0 + pa / TW, h - 1);
    const float2 pla = reinterpret_cast<const float2*>(planes)[(long long)b * hw + ya * w + xa];
    v4f s[DKB], sq[DKB];
#pragma unroll
    for (int k = 0; k < DKB; ++k) { s[k] = ref; sq[k] = ref * ref; }
    for (int v0 = 1; v0 < V; v0 += VC) {
        const int nv = min(VC, V - v0);
        if (v0 > 1) __syncthreads();
        for (int va = 0; va < nv; ++va)
            k1_phase_a<C, DKB>(rot, trans, b, V, v0 + va, va, (float)xa, (float)ya, pla, k0, hw, g, lds_o, lds_w);
        __syncthreads();
        if (!inside) continue;
#pragma unroll
        for (int k = 0; k < DKB; ++k) {
            v4f a = s[k], a2 = sq[k];
            for (int va = 0; va < nv; ++va) {
                K1Fetch<1> f;
                k1_issue<1, DKB, PIX>(f, lds_o, lds_w, rsrc, k, p, q4b, va);
                v4f val = blend4<FAST>(f.t[0][0], f.t[0][1], f.t[0][2], f.t[0][3], f.w[0]);
                a = a + val;
                if (FAST) a2 = __builtin_elementwise_fma(val, val, a2); else a2 = a2 + val * val;
            }
            if (v0 + nv < V) { s[k] = a; sq[k] = a2; continue; }
            if (k0 + k < D) k1_store_variance<FAST>(a, a2, fV, rV, ob + (long long)(k0 + k) * hw * C);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Round 4 (profiles/r4_k1_walls.txt): the wave-specialised form of this kernel (producer waves running phase A of plane chunk i+1 into a
// double-buffered tap table while consumer waves gather / blend / store chunk i; written at the end of round 3) was timed and REMOVED:
// 81 / 162 / 115 us per stage with one producer wave, 60 / 100 / 77 us with two, against 42 / 57 / 39 us for the two-phase kernel --
// phase A is ~40 % of the VALU work, so one or two producer waves feeding four consumers are the bottleneck.
// ------------------------------------------------------------------------------------------
// Round 3 (profiles/r3_k1_schedule_variants.txt, bit-identical forms of the kernel above, two source views): bilinear weights re-read
// from the LDS record at blend time instead of carried with the taps (133-136 instead of 153-156 VGPRs) 155.6 us per scene; the same
// with two tap sets prefetched 181.2; the same capped at 128 VGPRs for four waves per SIMD (28-36 bytes of scratch) 173.2; against
// 139.2 for the kernel above in the same process.  Fewer registers / deeper prefetch do not help: code removed.
// What is NOT here any more (measured on the MI355X in round 2, profiles/r2_k1_pipelined_variants_timed.txt): the
// LDS-window variants of this kernel -- block-staged windows (78 / 102 / 84 us per stage), the persistent double-buffered
// form with buffer_load ... lds (137 / 164 / 116 us; 75 / 86 / 69 us with a 72-texel budget), its static-LDS-set form
// (94-111 / 105-120 / 84-97 us) and the register-held-window form (142 / 170 / 120 us) -- all bit-identical, all 1.7-3x
// slower than the two-phase kernel above (42 / 57 / 39 us): the tap table plus two windows leave room for 8 waves per CU,
// and the stage -> publish -> gather chain of a 32-pixel tile is latency-bound.  DESIGN.md section 4 has the analysis.
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// train-variant extra: warped RGB of every source view ++ source-only variance / V, written in
// the reference's NCDHW layout because the tensor crosses the module boundary
// (CascadeMVSNet.forward returns it, models/casmvsnet.py:231).  One thread per (pixel, plane).
// ------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void warp_noref_kernel(
    const float* __restrict__ feats, const float* __restrict__ imgs, const float* __restrict__ rot,
    const float* __restrict__ trans, const float* __restrict__ planes, float* __restrict__ out,
    int V, int D, int h, int w, int square_first) {
#pragma clang fp contract(off)
    const int b = blockIdx.z, k = blockIdx.y;
    const long long hw = (long long)h * w;
    long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hw) return;
    const int y = (int)(p / w), x = (int)(p % w);
    const float fx = (float)x, fy = (float)y;
    const float wm1 = (float)(w - 1), hm1 = (float)(h - 1);
    const float half_w = wm1 / 2.0f, half_h = hm1 / 2.0f;
    const float2 pl = reinterpret_cast<const float2*>(planes)[(long long)b * hw + p];
    const float d = pl.x + (float)k * pl.y;
    const int CT = 3 * (V - 1) + C;
    float* ob = out + (((long long)b * CT) * D + k) * hw + p;      // channel stride = D*hw
    const long long cs = (long long)D * hw;
    const float fV = (float)V;
    // one coordinate chain per source view; the C channel sums live in registers (view order = the oracle's)
    float s[C], sq[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { s[c] = 0.0f; sq[c] = 0.0f; }
    for (int v = 1; v < V; ++v) {
        const float* r = rot + ((long long)b * (V - 1) + (v - 1)) * 9;
        const float* t = trans + ((long long)b * (V - 1) + (v - 1)) * 3;
        const float rx = (r[0] * fx + r[1] * fy) + r[2];
        const float ry = (r[3] * fx + r[4] * fy) + r[5];
        const float rz = (r[6] * fx + r[7] * fy) + r[8];
        const WarpCoord tc = warp_taps(rx, ry, rz, t[0], t[1], t[2], d, half_w, half_h, wm1, hm1, w, h, 1);
        const float* im = imgs + ((long long)b * V + v) * hw * 3;
        for (int c = 0; c < 3; ++c) {
            float val = ((im[tc.off[0] * 3 + c] * tc.wgt[0] + im[tc.off[1] * 3 + c] * tc.wgt[1]) +
                         im[tc.off[2] * 3 + c] * tc.wgt[2]) + im[tc.off[3] * 3 + c] * tc.wgt[3];
            ob[(long long)((v - 1) * 3 + c) * cs] = val;
        }
        const float* src = feats + ((long long)b * V + v) * hw * C;
        const float* s0 = src + (long long)tc.off[0] * C;
        const float* s1 = src + (long long)tc.off[1] * C;
        const float* s2 = src + (long long)tc.off[2] * C;
        const float* s3 = src + (long long)tc.off[3] * C;
#pragma unroll
        for (int c = 0; c < C; c += 4) {
            const float4 a = *reinterpret_cast<const float4*>(s0 + c), bb = *reinterpret_cast<const float4*>(s1 + c);
            const float4 cc = *reinterpret_cast<const float4*>(s2 + c), dd = *reinterpret_cast<const float4*>(s3 + c);
            float val[4];
            val[0] = ((a.x * tc.wgt[0] + bb.x * tc.wgt[1]) + cc.x * tc.wgt[2]) + dd.x * tc.wgt[3];
            val[1] = ((a.y * tc.wgt[0] + bb.y * tc.wgt[1]) + cc.y * tc.wgt[2]) + dd.y * tc.wgt[3];
            val[2] = ((a.z * tc.wgt[0] + bb.z * tc.wgt[1]) + cc.z * tc.wgt[2]) + dd.z * tc.wgt[3];
            val[3] = ((a.w * tc.wgt[0] + bb.w * tc.wgt[1]) + cc.w * tc.wgt[2]) + dd.w * tc.wgt[3];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float vv = val[j];
                if (square_first) vv = vv * vv;
                s[c + j] = s[c + j] + vv;
                sq[c + j] = sq[c + j] + vv * vv;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float m = s[c] / fV;
        ob[(long long)(3 * (V - 1) + c) * cs] = sq[c] / fV - m * m;
    }
}

}  // namespace rcmvs

using namespace rcmvs;

extern "C" {

// variant: 0 = two-phase gather kernel (exact arithmetic), 1 = the same with FMA-contracted blend, 2 = reference-order kernel (one full
// coordinate chain per lane, compiler IEEE division: what variant 0 is held bit-identical to), 3 = store-only ablation,
// 5 / 6 = the LDS-window form (k1_win.h), 7 = the plane-pipelined gather form (k1_pp.h); k1_production_variant() picks 0, 5 or 7
static int k1_launch(const float* feats, const float* rot, const float* trans, const float* planes, float* var,
                     int B, int V, int C, int D, int h, int w, int variant, hipStream_t st, unsigned* stats = nullptr,
                     hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr) {
    RCMVS_REQUIRE(feats && rot && trans && planes && var, "warp_variance_fwd: null pointer");
    RCMVS_REQUIRE(B > 0 && D > 0 && h > 1 && w > 1, "warp_variance_fwd: bad sizes B=%d D=%d h=%d w=%d", B, D, h, w);
    RCMVS_REQUIRE(V >= 2 && V - 1 <= RCMVS_MAX_SRC_VIEWS, "warp_variance_fwd: V=%d unsupported", V);
    RCMVS_REQUIRE((long long)h * w * C < (1LL << 31), "warp_variance_fwd: feature map too large for 32-bit offsets");
    RCMVS_REQUIRE((variant >= 0 && variant <= 3) || (variant >= 5 && variant <= 7), "warp_variance_fwd: unknown variant %d", variant);
    RCMVS_REQUIRE(C == 8 || C == 16 || C == 32, "warp_variance_fwd: C must be 8, 16 or 32 (got %d)", C);
    if (variant == 7) {
        // plane-pipelined gather form (k1_pp.h)
        const int nsrc = V - 1;
        RCMVS_REQUIRE(nsrc == 2 || nsrc == 3 || nsrc == 4 || nsrc == 6, "warp_variance_fwd: the plane-pipelined form is built for 2, 3, 4 or 6 source views (got V=%d)", V);
        RCMVS_REQUIRE((long long)V * h * w * C * 4 < 0x7fffffffLL, "warp_variance_fwd: feature block too large for 32-bit offsets");
#define RCMVS_K1PP(NN) do { \
            if (C == 32) return k1_pp_launch_one<32, 8, NN>(feats, rot, trans, planes, var, B, V, D, h, w, st, ev0, ev1); \
            if (C == 16) return k1_pp_launch_one<16, 8, NN>(feats, rot, trans, planes, var, B, V, D, h, w, st, ev0, ev1); \
            return k1_pp_launch_one<8, 8, NN>(feats, rot, trans, planes, var, B, V, D, h, w, st, ev0, ev1); } while (0)
        if (nsrc == 2) RCMVS_K1PP(2);
        if (nsrc == 3) RCMVS_K1PP(3);
        if (nsrc == 4) RCMVS_K1PP(4);
        RCMVS_K1PP(6);
#undef RCMVS_K1PP
    }
    if (variant >= 5) {
        // window form (k1_win.h): 5 = source windows loaded ahead of the coordinate phase, 6 = after the fit test
        RCMVS_REQUIRE(V == 3, "warp_variance_fwd: the window form is built for two source views (got V=%d)", V);
        RCMVS_REQUIRE((long long)V * h * w * C * 4 < 0x7fffffffLL, "warp_variance_fwd: feature block too large for 32-bit offsets");
#define RCMVS_K1WIN(CC, DD, PP, RR) (variant == 5 ? k1_win_launch_one<CC, DD, 2, PP, RR, 1>(feats, rot, trans, planes, var, B, V, D, h, w, stats, st, ev0, ev1) : \
                                                    k1_win_launch_one<CC, DD, 2, PP, RR, 2>(feats, rot, trans, planes, var, B, V, D, h, w, stats, st, ev0, ev1))
        if (C == 32) return RCMVS_K1WIN(32, 4, 16, 8);
        if (C == 16) return RCMVS_K1WIN(16, 4, 32, 8);
        return RCMVS_K1WIN(8, 4, 64, 8);
#undef RCMVS_K1WIN
    }
    if (variant <= 1) {
        const bool fastm = variant == 1;
        const int LPP = C / 4, PIX = 256 / LPP;
        const int nsrc = V - 1;
        const int nvt = (nsrc == 2 || nsrc == 4 || nsrc == 6) ? nsrc : 0;
        // 6 source views: half the plane chunk so that the tap table of all views still fits 48 KB of LDS
        const int dkb = (nvt == 6) ? ((C == 8) ? 2 : (C == 16 ? 4 : 8)) : ((C == 8) ? 4 : 8);
        const size_t per_view = (size_t)32 * dkb * PIX;
        int VC = nvt ? nvt : (int)((48 * 1024) / per_view);
        if (VC > nsrc) VC = nsrc;
        const size_t lds = per_view * VC;
        const int TWp = PIX / 4;
        const int txp = (w + TWp - 1) / TWp, typ = (h + 3) / 4;
        RCMVS_REQUIRE((long long)V * h * w * C * 4 < 0x7fffffffLL, "warp_variance_fwd: feature block too large for 32-bit offsets");
        dim3 gridp(txp * typ, (D + dkb - 1) / dkb, B);
#define RCMVS_K1TP(CC, DD, FF, NN) RCMVS_LAUNCH_TIMED((warp_variance_tp_kernel<CC, DD, FF, NN>), gridp, dim3(256), lds, st, ev0, ev1, feats, rot, trans, planes, var, V, D, h, w, txp)
#define RCMVS_K1MV(CC, DD, FF) RCMVS_LAUNCH_TIMED((warp_variance_mv_kernel<CC, DD, FF>), gridp, dim3(256), lds, st, ev0, ev1, feats, rot, trans, planes, var, V, D, h, w, txp, VC)
#define RCMVS_K1TP_N(CC, DD, FF) do { if (nvt == 2) RCMVS_K1TP(CC, DD, FF, 2); else if (nvt == 4) RCMVS_K1TP(CC, DD, FF, 4); else RCMVS_K1MV(CC, DD, FF); } while (0)
#define RCMVS_K1TP_F(CC, DD) do { if (fastm) RCMVS_K1TP_N(CC, DD, true); else RCMVS_K1TP_N(CC, DD, false); } while (0)
#define RCMVS_K1TP_6(CC, DD) do { if (fastm) RCMVS_K1TP(CC, DD, true, 6); else RCMVS_K1TP(CC, DD, false, 6); } while (0)
        if (nvt == 6) {
            switch (C) {
                case 8:  RCMVS_K1TP_6(8, 2); break;
                case 16: RCMVS_K1TP_6(16, 4); break;
                case 32: RCMVS_K1TP_6(32, 8); break;
                default: return fail(-1, "warp_variance_fwd: C must be 8, 16 or 32 (got %d)", C);
            }
            return launch_status("warp_variance_fwd");
        }
        switch (C) {
            case 8:  RCMVS_K1TP_F(8, 4); break;
            case 16: RCMVS_K1TP_F(16, 8); break;
            case 32: RCMVS_K1TP_F(32, 8); break;
            default: return fail(-1, "warp_variance_fwd: C must be 8, 16 or 32 (got %d)", C);
        }
        return launch_status("warp_variance_fwd");
    }
    const int TW = 256 / C, TH = 4;
    const int tiles_x = (w + TW - 1) / TW, tiles_y = (h + TH - 1) / TH;
    dim3 grid(tiles_x * tiles_y, (D + DK - 1) / DK, B);
#define RCMVS_K1_LAUNCH(CC, VV) RCMVS_LAUNCH_TIMED((warp_variance_ref_kernel<CC, VV>), grid, dim3(256), 0, st, ev0, ev1, feats, rot, trans, planes, var, V, D, h, w, tiles_x, tiles_y)
#define RCMVS_K1_VARIANTS(CC) do { if (variant == 2) RCMVS_K1_LAUNCH(CC, false); else RCMVS_K1_LAUNCH(CC, true); } while (0)
    switch (C) {
        case 8:  RCMVS_K1_VARIANTS(8); break;
        case 16: RCMVS_K1_VARIANTS(16); break;
        case 32: RCMVS_K1_VARIANTS(32); break;
        default: return fail(-1, "warp_variance_fwd: C must be 8, 16 or 32 (got %d)", C);
    }
#undef RCMVS_K1_VARIANTS
#undef RCMVS_K1_LAUNCH
    return launch_status("warp_variance_fwd");
}

// which kernel a call takes.  No hint: the exact two-phase gather kernel (bit-identical to the reference-order kernel) -- the plain ABI entry.
// RCMVS_K1_FAST_BLEND (the caller accepts FMA-contracted blends, <= 2e-6 of the value range) opens the faster forms, chosen by measurement
// on the stage shapes of each view count (us per launch, profiles/r6_k1_views.txt; 0 two-phase, 1 its FMA build, 5 window, 7 plane-pipelined):
//   2 source views (DTU bench, 512 x 640)   stage 1 with RCMVS_K1_UNIFORM_PLANES: 5 (38.8 against 41.0 / 52.0 for 0 / 7); C = 8: 7 (50.4 against 52.7); else 0
//   3 (training)                            7 everywhere, stage 1 included (59 / 74 / 48 against 78 / 173 / 64 for 0: the run-time-view kernel)
//   4 (DTU evaluation, 1184 x 1600)         7 (stage 1 383 against 450, stage 3 379 against 487 on smooth tables); per-pixel C = 16: 1 (527 / 942 smooth / rough
//                                           against 529 / 1 138 for 7 and 614 / 941 for 0)
//   6 (Tanks and Temples, 1056 x 1920)      7 (721 / 767 / 537 against 1 263 / 1 203 / 983)
// other view counts: 0 (its run-time-view form).
static int k1_production_variant(int V, int C, int h, int w, int hint) {
    const bool small = (long long)V * h * w * C * 4 < 0x7fffffffLL;
    const int nsrc = V - 1;
    if (!small || !(hint & RCMVS_K1_FAST_BLEND)) return 0;
    const bool uniform = (hint & RCMVS_K1_UNIFORM_PLANES) != 0;
    if (nsrc == 2) return uniform ? 5 : (C == 8 ? 7 : 0);
    if (nsrc == 3 || nsrc == 6) return 7;
    if (nsrc == 4) return (C == 16 && !uniform) ? 1 : 7;
    return 0;
}

int rcmvs_warp_variance_fwd(const float* feats, const float* rot, const float* trans,
                            const float* planes, float* var,
                            int B, int V, int C, int D, int h, int w, void* stream) {
    return k1_launch(feats, rot, trans, planes, var, B, V, C, D, h, w, 0, as_stream(stream));
}

int rcmvs_warp_variance_hint_fwd(const float* feats, const float* rot, const float* trans,
                                 const float* planes, float* var,
                                 int B, int V, int C, int D, int h, int w, int hint, void* stream) {
    return k1_launch(feats, rot, trans, planes, var, B, V, C, D, h, w, k1_production_variant(V, C, h, w, hint), as_stream(stream));
}

int rcmvs_warp_variance_timed_fwd(const float* feats, const float* rot, const float* trans,
                                  const float* planes, float* var,
                                  int B, int V, int C, int D, int h, int w, int hint, void* start_event, void* stop_event, void* stream) {
    return k1_launch(feats, rot, trans, planes, var, B, V, C, D, h, w, k1_production_variant(V, C, h, w, hint), as_stream(stream), nullptr,
                     reinterpret_cast<hipEvent_t>(start_event), reinterpret_cast<hipEvent_t>(stop_event));
}

int rcmvs_debug_warp_variance_fwd(const float* feats, const float* rot, const float* trans,
                                  const float* planes, float* var,
                                  int B, int V, int C, int D, int h, int w, int variant, void* stream) {
    return k1_launch(feats, rot, trans, planes, var, B, V, C, D, h, w, variant, as_stream(stream));
}

int rcmvs_debug_warp_variance_win_fwd(const float* feats, const float* rot, const float* trans,
                                      const float* planes, float* var,
                                      int B, int V, int C, int D, int h, int w, int variant, unsigned* stats, void* stream) {
    RCMVS_REQUIRE(variant == 5 || variant == 6, "debug_warp_variance_win_fwd: variant must be 5 or 6 (got %d)", variant);
    return k1_launch(feats, rot, trans, planes, var, B, V, C, D, h, w, variant, as_stream(stream), stats);
}

int rcmvs_warp_noref_fwd(const float* feats, const float* imgs, const float* rot, const float* trans,
                         const float* planes, float* out,
                         int B, int V, int C, int D, int h, int w, int square_first, void* stream) {
    RCMVS_REQUIRE(feats && imgs && rot && trans && planes && out, "warp_noref_fwd: null pointer");
    RCMVS_REQUIRE(B > 0 && D > 0 && h > 1 && w > 1, "warp_noref_fwd: bad sizes");
    RCMVS_REQUIRE(C == 8 || C == 16 || C == 32, "warp_noref_fwd: C must be 8, 16 or 32 (got %d)", C);
    RCMVS_REQUIRE(V >= 2 && V - 1 <= RCMVS_MAX_SRC_VIEWS, "warp_noref_fwd: V=%d unsupported", V);
    dim3 grid((unsigned)cdiv((long long)h * w, 256), D, B);
#define RCMVS_NOREF(CC) hipLaunchKernelGGL((warp_noref_kernel<CC>), grid, dim3(256), 0, as_stream(stream), feats, imgs, rot, trans, planes, out, V, D, h, w, square_first)
    if (C == 8) RCMVS_NOREF(8); else if (C == 16) RCMVS_NOREF(16); else RCMVS_NOREF(32);
#undef RCMVS_NOREF
    return launch_status("warp_noref_fwd");
}

}  // extern "C"
