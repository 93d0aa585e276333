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

template <int C, int DKB, bool FAST, int NVT, int NG = (NVT == 0 ? 1 : (NVT > 4 ? 3 : NVT))>
__global__ __launch_bounds__(256) void warp_variance_tp_kernel(
    const float* __restrict__ feats, const float* __restrict__ rot, const float* __restrict__ trans,
    const float* __restrict__ planes, float* __restrict__ var, int V, int D, int h, int w, int tiles_x, int VC) {
#pragma clang fp contract(off)
    constexpr int LPP = C / 4;
    constexpr int PIX = 256 / LPP;          // pixels per block
    constexpr int TH = 4, TW = PIX / TH;
    constexpr int GRP = 256 / PIX;          // phase-A threads per pixel
    constexpr int KPT = DKB / GRP;          // planes per phase-A thread
    constexpr bool MULTI = (NVT == 0);
    static_assert(DKB % GRP == 0, "DKB must be a multiple of 256/PIX");
    extern __shared__ __attribute__((aligned(16))) v4i lds_o[];          // [nv][DKB][PIX] offsets, then weights
    const int nvmax = MULTI ? VC : NVT;
    v4f* lds_w = reinterpret_cast<v4f*>(lds_o + nvmax * DKB * PIX);
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
    // ---- phase-A identity
    const int pa = threadIdx.x % PIX, ga = threadIdx.x / PIX;
    const int xa = min(tx0 + pa % TW, w - 1), ya = min(ty0 + pa / TW, h - 1);
    const float fxa = (float)xa, fya = (float)ya;
    const float2 pla = reinterpret_cast<const float2*>(planes)[(long long)b * hw + ya * w + xa];

    v4f s[MULTI ? DKB : 1], sq[MULTI ? DKB : 1];
    if (MULTI) {
#pragma unroll
        for (int k = 0; k < DKB; ++k) { s[k] = ref; sq[k] = ref * ref; }
    }

    for (int v0 = 1; v0 < V; v0 += nvmax) {
        const int nv = MULTI ? min(VC, V - v0) : NVT;
        if (MULTI && v0 > 1) __syncthreads();
        // ---------------- phase A
        for (int va = 0; va < nv; ++va) {
            const float* r = rot + ((long long)b * (V - 1) + (v0 + va - 1)) * 9;
            const float* t = trans + ((long long)b * (V - 1) + (v0 + va - 1)) * 3;
            const float rx = (r[0] * fxa + r[1] * fya) + r[2];
            const float ry = (r[3] * fxa + r[4] * fya) + r[5];
            const float rz = (r[6] * fxa + r[7] * fya) + r[8];
            const float t0 = t[0], t1 = t[1], t2 = t[2];
            const int vrow = (v0 + va) * hw;
#pragma unroll
            for (int kk = 0; kk < KPT; ++kk) {
                const int ka = ga + kk * GRP;
                const float d = pla.x + (float)(k0 + ka) * pla.y;
                v4i o;
                v4f wt;
                k1_tap<C>(rx, ry, rz, t0, t1, t2, d, g, vrow, o, wt);
                const int idx = (va * DKB + ka) * PIX + pa;
                lds_o[idx] = o;
                lds_w[idx] = wt;
            }
        }
        __syncthreads();
        // ---------------- phase B
        if (!inside) continue;
        if constexpr (!MULTI) {
            // double-buffered gathers over the flattened (plane, view group) sequence: the group after the current one
            // (same plane or the next) is in flight while the current one is blended.  NG = views per group: all of them
            // for 2 / 4 source views, 3 for the 7-view setting (two full tap sets of 6 views exceed the register file).
            // Views are accumulated in ascending order whatever the grouping, so the result is bit-identical.
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
        } else {
#pragma unroll
            for (int k = 0; k < DKB; ++k) {
                v4f a = s[k], a2 = sq[k];
                for (int va = 0; va < nv; ++va) {
                    K1Fetch<1> f;
                    const int idx = (va * DKB + k) * PIX + p;
                    const v4i o = lds_o[idx];
                    f.w[0] = lds_w[idx];
                    f.t[0][0] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.x + q4b, 0, 0));
                    f.t[0][1] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.y + q4b, 0, 0));
                    f.t[0][2] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.z + q4b, 0, 0));
                    f.t[0][3] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.w + q4b, 0, 0));
                    v4f val = blend4<FAST>(f.t[0][0], f.t[0][1], f.t[0][2], f.t[0][3], f.w[0]);
                    a = a + val;
                    if (FAST) a2 = __builtin_elementwise_fma(val, val, a2); else a2 = a2 + val * val;
                }
                if (v0 + nv < V) { s[k] = a; sq[k] = a2; continue; }
                if (k0 + k < D) k1_store_variance<FAST>(a, a2, fV, rV, ob + (long long)(k0 + k) * hw * C);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// K1, LDS-staged variant: same two-phase structure and arithmetic as warp_variance_tp_kernel,
// but the bilinear taps are gathered from LDS instead of through the vector L1.
// Why: rocprof PMC on the tp kernel shows TCP_TOTAL_CACHE_ACCESSES = 8 taps x the output volume
// (18-24 M 64-byte accesses per launch), i.e. >= 30/40/30 us of vL1D time at one access per clock
// per CU for the three config-2 stages -- the gathers, not the HBM stream, are the wall.  A tile's
// samples land in a small source window (tile extent + the disparity sweep of DKB planes + 1), so:
//   A   per (pixel, plane, view): coordinate chain -> clamped integer tap coordinates + masked
//       weights; the block-wide bounding box of all contributing taps is reduced with LDS atomics;
//   A2  records rewritten to byte offsets inside the staged window (or to global offsets when the
//       window does not fit the LDS budget: block-uniform fallback to buffer loads);
//   S   the window rows are copied global -> LDS with coalesced 16-byte loads, each texel once;
//   B   as before, but each tap is a ds_read_b128.
// LDS traffic replaces ~8x-output of L1 traffic by ~1-2x-output of staging traffic.
// ------------------------------------------------------------------------------------------
template <int C, int DKB, bool FAST, int NV>
__global__ __launch_bounds__(256) void warp_variance_lds_kernel(
    const float* __restrict__ feats, const float* __restrict__ rot, const float* __restrict__ trans,
    const float* __restrict__ planes, float* __restrict__ var, int V, int D, int h, int w, int tiles_x, int patch_texels) {
#pragma clang fp contract(off)
    constexpr int LPP = C / 4;
    constexpr int PIX = 256 / LPP;
    constexpr int TH = 4, TW = PIX / TH;
    constexpr int GRP = 256 / PIX;
    constexpr int KPT = (DKB + GRP - 1) / GRP;   // planes per phase-A thread (threads with ga >= DKB idle when DKB < GRP)
    constexpr int C4 = C * 4;               // bytes per texel
    extern __shared__ __attribute__((aligned(16))) v4i lds_o[];          // [NV][DKB][PIX]
    v4f* lds_w = reinterpret_cast<v4f*>(lds_o + NV * DKB * PIX);         // [NV][DKB][PIX]
    int* lds_box = reinterpret_cast<int*>(lds_w + NV * DKB * PIX);       // [4 waves][NV][4] xmin xmax ymin ymax
    char* lds_patch = reinterpret_cast<char*>(lds_box + 16 * NV);        // [NV][patch_texels * C4]
    const int patch_bytes = patch_texels * C4;
    const unsigned patch_base = (unsigned)(lds_patch - reinterpret_cast<char*>(lds_o));

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
    const int pa = threadIdx.x % PIX, ga = threadIdx.x / PIX;
    const int xa = min(tx0 + pa % TW, w - 1), ya = min(ty0 + pa / TW, h - 1);
    const float fxa = (float)xa, fya = (float)ya;
    const float2 pla = reinterpret_cast<const float2*>(planes)[(long long)b * hw + ya * w + xa];
    const bool multi = (V - 1) > NV;

    v4f s[DKB], sq[DKB];
#pragma unroll
    for (int k = 0; k < DKB; ++k) { s[k] = ref; sq[k] = ref * ref; }

    for (int v0 = 1; v0 < V; v0 += NV) {
        const int nv = min(NV, V - v0);
        if (v0 > 1) __syncthreads();                             // previous chunk's phase B is done with LDS
        // ---------------- phase A: taps -> packed clamped coordinates + weights, bounding box
        for (int va = 0; va < nv; ++va) {
            const float* r = rot + ((long long)b * (V - 1) + (v0 + va - 1)) * 9;
            const float* t = trans + ((long long)b * (V - 1) + (v0 + va - 1)) * 3;
            const float rx = (r[0] * fxa + r[1] * fya) + r[2];
            const float ry = (r[3] * fxa + r[4] * fya) + r[5];
            const float rz = (r[6] * fxa + r[7] * fya) + r[8];
            const float t0 = t[0], t1 = t[1], t2 = t[2];
            int bx0 = 0x7fffffff, bx1 = -1, by0 = 0x7fffffff, by1 = -1;
#pragma unroll
            for (int kk = 0; kk < KPT; ++kk) {
                const int ka = ga + kk * GRP;
                if (ka >= DKB) continue;
                const float d = pla.x + (float)(k0 + ka) * pla.y;
                v4i o;
                v4f wt;
                int xi, yi;
                k1_chain(rx, ry, rz, t0, t1, t2, d, g, xi, yi, wt);
                const int xc0 = min(max(xi, 0), w - 1), xc1 = min(max(xi + 1, 0), w - 1);
                const int yc0 = min(max(yi, 0), h - 1), yc1 = min(max(yi + 1, 0), h - 1);
                const bool any = (wt.x != 0.0f) || (wt.y != 0.0f) || (wt.z != 0.0f) || (wt.w != 0.0f);
                const int idx = (va * DKB + ka) * PIX + pa;
                v4i rec;
                rec.x = xc0; rec.y = yc0; rec.z = (xc1 != xc0 ? 1 : 0) | (yc1 != yc0 ? 2 : 0) | (any ? 4 : 0); rec.w = 0;
                lds_o[idx] = rec;
                lds_w[idx] = wt;
                if (any) { bx0 = min(bx0, xc0); bx1 = max(bx1, xc1); by0 = min(by0, yc0); by1 = max(by1, yc1); }
                (void)o;
            }
            // wave-level reduction in registers (DPP + readlane), one plain LDS store per wave: LDS atomics -- even from a
            // single lane -- are expanded by the compiler into a 64-iteration scalar loop each (2.7 k SALU per wave, PMC)
            bx0 = wave_reduce_i32<true>(bx0); bx1 = wave_reduce_i32<false>(bx1);
            by0 = wave_reduce_i32<true>(by0); by1 = wave_reduce_i32<false>(by1);
            if ((threadIdx.x & 63) == 0)
                *reinterpret_cast<v4i*>(lds_box + ((threadIdx.x >> 6) * NV + va) * 4) = (v4i){bx0, bx1, by0, by1};
        }
        __syncthreads();
        // ---------------- phase A2: rewrite records to byte offsets (LDS window or global fallback)
        bool fits[NV];
        int px0[NV], py0[NV], pw[NV], ph[NV];
#pragma unroll
        for (int va = 0; va < NV; ++va) {
            v4i bb = *reinterpret_cast<const v4i*>(lds_box + va * 4);
#pragma unroll
            for (int wv = 1; wv < 4; ++wv) {
                const v4i t = *reinterpret_cast<const v4i*>(lds_box + (wv * NV + va) * 4);
                bb.x = min(bb.x, t.x); bb.y = max(bb.y, t.y); bb.z = min(bb.z, t.z); bb.w = max(bb.w, t.w);
            }
            px0[va] = bb.x; py0[va] = bb.z;
            pw[va] = bb.y - bb.x + 1; ph[va] = bb.w - bb.z + 1;
            const bool empty = bb.y < 0;
            if (empty) { px0[va] = 0; py0[va] = 0; pw[va] = 1; ph[va] = 1; }     // stage one (finite) texel
            fits[va] = (va < nv) && (pw[va] * ph[va] <= patch_texels);
        }
#pragma unroll
        for (int va = 0; va < NV; ++va) {
            if (va >= nv) continue;
#pragma unroll
            for (int kk = 0; kk < KPT; ++kk) {
                if (ga + kk * GRP >= DKB) continue;
                const int idx = (va * DKB + ga + kk * GRP) * PIX + pa;
                const v4i rec = lds_o[idx];
                const bool any = rec.z & 4;
                v4i o;
                if (fits[va]) {
                    const int lx = any ? rec.x - px0[va] : 0, ly = any ? rec.y - py0[va] : 0;
                    const int base = (ly * pw[va] + lx) * C4 + (int)patch_base + va * patch_bytes;
                    const int dx = (any && (rec.z & 1)) ? C4 : 0, dy = (any && (rec.z & 2)) ? pw[va] * C4 : 0;
                    o.x = base; o.y = base + dx; o.z = base + dy; o.w = base + dy + dx;
                } else {
                    const int base = ((v0 + va) * hw + rec.y * w + rec.x) * C4;
                    const int dx = (rec.z & 1) ? C4 : 0, dy = (rec.z & 2) ? w * C4 : 0;
                    o.x = base; o.y = base + dx; o.z = base + dy; o.w = base + dy + dx;
                }
                lds_o[idx] = o;
            }
        }
        // ---------------- stage the source windows (each texel once, coalesced rows)
#pragma unroll
        for (int va = 0; va < NV; ++va) {
            if (!fits[va]) continue;
            const int row4 = pw[va] * LPP;                        // float4s per window row
            const int n4 = row4 * ph[va];
            const float* src = fb + ((long long)(v0 + va) * hw + (long long)py0[va] * w + px0[va]) * C;
            v4f* dst = reinterpret_cast<v4f*>(lds_patch + va * patch_bytes);
            int row = threadIdx.x / row4, c4 = threadIdx.x - row * row4;   // one division per view, then incremental
            const int drow = 256 / row4, dc4 = 256 - drow * row4;
            for (int e = threadIdx.x; e < n4; e += 256) {
                dst[e] = *reinterpret_cast<const v4f*>(src + (long long)row * w * C + c4 * 4);
                c4 += dc4; row += drow;
                if (c4 >= row4) { c4 -= row4; ++row; }
            }
        }
        __syncthreads();
        // ---------------- phase B
        if (!inside) continue;
        const char* lds_bytes = reinterpret_cast<const char*>(lds_o) + q4b;
#pragma unroll
        for (int k = 0; k < DKB; ++k) {
            v4f a = s[k], a2 = sq[k];
#pragma unroll
            for (int va = 0; va < NV; ++va) {
                if (va >= nv) continue;
                const int idx = (va * DKB + k) * PIX + p;
                const v4i o = lds_o[idx];
                const v4f wt = lds_w[idx];
                v4f ta, tb, tc, td;
                if (fits[va]) {
                    ta = *reinterpret_cast<const v4f*>(lds_bytes + o.x);
                    tb = *reinterpret_cast<const v4f*>(lds_bytes + o.y);
                    tc = *reinterpret_cast<const v4f*>(lds_bytes + o.z);
                    td = *reinterpret_cast<const v4f*>(lds_bytes + o.w);
                } else {
                    ta = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.x + q4b, 0, 0));
                    tb = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.y + q4b, 0, 0));
                    tc = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.z + q4b, 0, 0));
                    td = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.w + q4b, 0, 0));
                }
                v4f val = blend4<FAST>(ta, tb, tc, td, wt);
                a = a + val;
                if (FAST) a2 = __builtin_elementwise_fma(val, val, a2); else a2 = a2 + val * val;
            }
            if (multi && v0 + nv < V) { s[k] = a; sq[k] = a2; continue; }
            if (k0 + k < D) k1_store_variance<FAST>(a, a2, fV, rV, ob + (long long)(k0 + k) * hw * C);
        }
    }
}

// ------------------------------------------------------------------------------------------
// K1, pipelined staged variant (debug variants 8 / 9; NOT measured yet -- logic verified bit-identical to the
// reference-order kernel on the CPU emulation, tests/test_emu_kernels_cpu.py).  The LDS-staged kernel above loses to
// the production kernel because every plane chunk is stage -> barrier -> gather (its waves wait half their cycles,
// profiles/r1_k1_pmc_summary.txt).  Here one block owns a tile for ALL plane chunks and keeps two LDS sets
// (tap table + source windows): while chunk i is blended out of set i & 1, the taps of chunk i + 1 are computed and
// its windows are loaded straight into the other set with buffer_load ... lds (no VGPRs in flight, the only wait is
// the vmcnt(0) in front of the barrier that publishes them).  Up to NV = 2 source views (config 2); anything else
// stays on the other kernels.
// ------------------------------------------------------------------------------------------
template <int C, int DKB, bool FAST, int NV, bool DMA = true>
__global__ __launch_bounds__(256) void warp_variance_ps_kernel(
    const float* __restrict__ feats, const float* __restrict__ rot, const float* __restrict__ trans,
    const float* __restrict__ planes, float* __restrict__ var, int V, int D, int h, int w, int tiles_x, int patch_texels, int pad) {
#pragma clang fp contract(off)
    constexpr int LPP = C / 4;
    constexpr int PIX = 256 / LPP;
    constexpr int TH = 4, TW = PIX / TH;
    constexpr int GRP = 256 / PIX;
    constexpr int KPT = (DKB + GRP - 1) / GRP;
    constexpr int C4 = C * 4;
    constexpr int TAB = NV * DKB * PIX;
    extern __shared__ __attribute__((aligned(16))) v4i lds_ps[];
    const int cs = C4 + (DMA ? 0 : pad);                         // bytes per staged texel (padding spreads the gathers over the LDS banks)
    const int patch_bytes = patch_texels * cs;
    const int set_v4 = TAB * 2 + 4 * NV + NV * (patch_bytes / 16);           // v4i units per set: offsets, weights, boxes, windows
    const int nv = V - 1;

    const int b = blockIdx.z;
    const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tx0 = (tile % tiles_x) * TW, ty0 = (tile / tiles_x) * TH;
    const int hw = h * w;
    const int nch = (D + DKB - 1) / DKB;
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
    const int pa = threadIdx.x % PIX, ga = threadIdx.x / PIX;
    const int xa = min(tx0 + pa % TW, w - 1), ya = min(ty0 + pa / TW, h - 1);
    const float fxa = (float)xa, fya = (float)ya;
    const float2 pla = reinterpret_cast<const float2*>(planes)[(long long)b * hw + ya * w + xa];
    const int lane = threadIdx.x & 63;

    // ---- phase A of chunk kc into set `st`: taps -> clamped coordinates + weights, per-wave bounding boxes
    auto phase_a = [&](int kc, int st) {
        v4i* lo = lds_ps + st * set_v4;
        v4f* lw = reinterpret_cast<v4f*>(lo + TAB);
        int* lbox = reinterpret_cast<int*>(lo + 2 * TAB);
        const int k0 = kc * DKB;
        for (int va = 0; va < nv; ++va) {
            const float* r = rot + ((long long)b * (V - 1) + va) * 9;
            const float* t = trans + ((long long)b * (V - 1) + va) * 3;
            const float rx = (r[0] * fxa + r[1] * fya) + r[2];
            const float ry = (r[3] * fxa + r[4] * fya) + r[5];
            const float rz = (r[6] * fxa + r[7] * fya) + r[8];
            const float t0 = t[0], t1 = t[1], t2 = t[2];
            int bx0 = 0x7fffffff, bx1 = -1, by0 = 0x7fffffff, by1 = -1;
#pragma unroll
            for (int kk = 0; kk < KPT; ++kk) {
                const int ka = ga + kk * GRP;
                if (ka >= DKB) continue;
                const float d = pla.x + (float)(k0 + ka) * pla.y;
                v4f wt;
                int xi, yi;
                k1_chain(rx, ry, rz, t0, t1, t2, d, g, xi, yi, wt);
                const int xc0 = min(max(xi, 0), w - 1), xc1 = min(max(xi + 1, 0), w - 1);
                const int yc0 = min(max(yi, 0), h - 1), yc1 = min(max(yi + 1, 0), h - 1);
                const bool any = (wt.x != 0.0f) || (wt.y != 0.0f) || (wt.z != 0.0f) || (wt.w != 0.0f);
                const int idx = (va * DKB + ka) * PIX + pa;
                v4i rec;
                rec.x = xc0; rec.y = yc0; rec.z = (xc1 != xc0 ? 1 : 0) | (yc1 != yc0 ? 2 : 0) | (any ? 4 : 0); rec.w = 0;
                lo[idx] = rec;
                lw[idx] = wt;
                if (any) { bx0 = min(bx0, xc0); bx1 = max(bx1, xc1); by0 = min(by0, yc0); by1 = max(by1, yc1); }
            }
            bx0 = wave_reduce_i32<true>(bx0); bx1 = wave_reduce_i32<false>(bx1);
            by0 = wave_reduce_i32<true>(by0); by1 = wave_reduce_i32<false>(by1);
            if (lane == 0) *reinterpret_cast<v4i*>(lbox + ((threadIdx.x >> 6) * NV + va) * 4) = (v4i){bx0, bx1, by0, by1};
        }
    };

    // ---- phase A2 + S of set `st` (after a barrier): block bounding box, records -> byte offsets, windows -> LDS (direct loads)
    constexpr int MAXP = 4;                                       // 16-byte pieces per thread and view in the register-staged form
    v4f held[NV][MAXP];
    int held_n4[NV], held_dst[NV];
    auto stage = [&](int st, bool* fits) {
        v4i* lo = lds_ps + st * set_v4;
        const int* lbox = reinterpret_cast<const int*>(lo + 2 * TAB);
        char* lpatch = reinterpret_cast<char*>(lo + 2 * TAB + 4 * NV);
        const unsigned patch_base = (unsigned)(lpatch - reinterpret_cast<char*>(lds_ps));
        int px0[NV], py0[NV], pw[NV], ph[NV];
#pragma unroll
        for (int va = 0; va < NV; ++va) {
            v4i bb = *reinterpret_cast<const v4i*>(lbox + va * 4);
#pragma unroll
            for (int wv = 1; wv < 4; ++wv) {
                const v4i t = *reinterpret_cast<const v4i*>(lbox + (wv * NV + va) * 4);
                bb.x = min(bb.x, t.x); bb.y = max(bb.y, t.y); bb.z = min(bb.z, t.z); bb.w = max(bb.w, t.w);
            }
            px0[va] = bb.x; py0[va] = bb.z;
            pw[va] = bb.y - bb.x + 1; ph[va] = bb.w - bb.z + 1;
            if (bb.y < 0) { px0[va] = 0; py0[va] = 0; pw[va] = 1; ph[va] = 1; }
            fits[va] = (va < nv) && (pw[va] * ph[va] <= patch_texels);
        }
#pragma unroll
        for (int va = 0; va < NV; ++va) {
            if (va >= nv) continue;
#pragma unroll
            for (int kk = 0; kk < KPT; ++kk) {
                if (ga + kk * GRP >= DKB) continue;
                const int idx = (va * DKB + ga + kk * GRP) * PIX + pa;
                const v4i rec = lo[idx];
                const bool any = rec.z & 4;
                v4i o;
                if (fits[va]) {
                    const int lx = any ? rec.x - px0[va] : 0, ly = any ? rec.y - py0[va] : 0;
                    const int base = (ly * pw[va] + lx) * cs + (int)patch_base + va * patch_bytes;
                    const int dx = (any && (rec.z & 1)) ? cs : 0, dy = (any && (rec.z & 2)) ? pw[va] * cs : 0;
                    o.x = base; o.y = base + dx; o.z = base + dy; o.w = base + dy + dx;
                } else {
                    const int base = ((1 + va) * hw + rec.y * w + rec.x) * C4;
                    const int dx = (rec.z & 1) ? C4 : 0, dy = (rec.z & 2) ? w * C4 : 0;
                    o.x = base; o.y = base + dx; o.z = base + dy; o.w = base + dy + dx;
                }
                lo[idx] = o;
            }
        }
#pragma unroll
        for (int va = 0; va < NV; ++va) {
            if (!fits[va]) continue;
            const int row4 = pw[va] * LPP;                        // 16-byte pieces per window row
            const int n4 = row4 * ph[va];
            const int src0 = (((1 + va) * hw) + py0[va] * w + px0[va]) * C4;      // byte offset of the window's first texel
            char* dst = lpatch + va * patch_bytes;
            if constexpr (DMA) {
                for (int e0 = (threadIdx.x & ~63); e0 < n4; e0 += 256) {          // one wave moves 64 consecutive pieces = 1 KB of LDS
                    const int e = e0 + lane;
                    if (e < n4) {
                        const int row = e / row4, c4 = e - row * row4;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst + (size_t)e0 * 16, 16, src0 + (row * w * C + c4 * 4) * 4, 0, 0, 0);
                    }
                }
            } else {
                // register-staged form: the loads are issued now, the LDS writes follow the blend of the current chunk
                held_n4[va] = n4;
                held_dst[va] = (int)(dst - reinterpret_cast<char*>(lds_ps));
#pragma unroll
                for (int i = 0; i < MAXP; ++i) {
                    const int e = threadIdx.x + i * 256;
                    const int row = e / row4, c4 = e - row * row4;
                    held[va][i] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, e < n4 ? src0 + (row * w * C + c4 * 4) * 4 : (int)0x80000000, 0, 0));
                }
            }
        }
        if constexpr (!DMA) {
#pragma unroll
            for (int va = 0; va < NV; ++va)
                if (!fits[va]) held_n4[va] = 0;
        }
    };
    auto stage_finish = [&]() {
        if constexpr (!DMA) {
#pragma unroll
            for (int va = 0; va < NV; ++va) {
#pragma unroll
                for (int i = 0; i < MAXP; ++i) {
                    const int e = threadIdx.x + i * 256;
                    if (e < held_n4[va]) {
                        // piece e = (texel t of the window, 16-byte quad q); texels sit cs bytes apart in LDS
                        const int t = e / LPP, q = e - t * LPP;
                        *reinterpret_cast<v4f*>(reinterpret_cast<char*>(lds_ps) + held_dst[va] + t * cs + q * 16) = held[va][i];
                    }
                }
            }
        }
    };

    // ---- phase B of chunk kc out of set `st`: a fits-only body (ds_reads only) and a mixed body, see warp_variance_pss_kernel
    auto phase_b_body = [&](int kc, int st, const bool* fits, auto mixed) {
        constexpr bool MIXED = decltype(mixed)::value;
        const v4i* lo = lds_ps + st * set_v4;
        const v4f* lw = reinterpret_cast<const v4f*>(lo + TAB);
        const char* lds_bytes = reinterpret_cast<const char*>(lds_ps) + q4b;
        const int k0 = kc * DKB;
#pragma unroll
        for (int k = 0; k < DKB; ++k) {
            v4f a = ref, a2 = ref * ref;
#pragma unroll
            for (int va = 0; va < NV; ++va) {
                if (va >= nv) continue;
                const int idx = (va * DKB + k) * PIX + p;
                const v4i o = lo[idx];
                const v4f wt = lw[idx];
                v4f ta, tb, tc, td;
                if (!MIXED || fits[va]) {
                    ta = *reinterpret_cast<const v4f*>(lds_bytes + o.x);
                    tb = *reinterpret_cast<const v4f*>(lds_bytes + o.y);
                    tc = *reinterpret_cast<const v4f*>(lds_bytes + o.z);
                    td = *reinterpret_cast<const v4f*>(lds_bytes + o.w);
                } else {
                    ta = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.x + q4b, 0, 0));
                    tb = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.y + q4b, 0, 0));
                    tc = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.z + q4b, 0, 0));
                    td = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.w + q4b, 0, 0));
                }
                v4f val = blend4<FAST>(ta, tb, tc, td, wt);
                a = a + val;
                if (FAST) a2 = __builtin_elementwise_fma(val, val, a2); else a2 = a2 + val * val;
            }
            if (k0 + k < D) k1_store_variance<FAST>(a, a2, fV, rV, ob + (long long)(k0 + k) * hw * C);
        }
    };
    auto phase_b = [&](int kc, int st, const bool* fits) {
        if (!inside) return;
        bool all = true;
#pragma unroll
        for (int va = 0; va < NV; ++va) all = all && (va >= nv || fits[va]);
        if (all) phase_b_body(kc, st, fits, std::false_type{});
        else phase_b_body(kc, st, fits, std::true_type{});
    };

    bool fits_cur[NV], fits_next[NV];
#pragma unroll
    for (int va = 0; va < NV; ++va) held_n4[va] = 0;
    phase_a(0, 0);
    __syncthreads();
    stage(0, fits_cur);
    stage_finish();
    for (int kc = 0; kc < nch; ++kc) {
        const int st = kc & 1;
        if (kc + 1 < nch) phase_a(kc + 1, st ^ 1);
        __builtin_amdgcn_s_waitcnt(0);                             // vmcnt(0): the windows of chunk kc have landed in LDS
        __syncthreads();                                           // ... for every wave; tables and boxes of chunk kc + 1 are visible
        if (kc + 1 < nch) stage(st ^ 1, fits_next);
        phase_b(kc, st, fits_cur);
        if (kc + 1 < nch) stage_finish();                          // register-staged form only: windows of chunk kc + 1 -> LDS
        __syncthreads();                                           // set `st` is free for the taps of chunk kc + 2
#pragma unroll
        for (int va = 0; va < NV; ++va) fits_cur[va] = fits_next[va];
    }
}

// ------------------------------------------------------------------------------------------
// K1, pipelined staged variant with STATIC LDS sets (debug variants 12 / 13).  Same algorithm as warp_variance_ps_kernel with
// direct-to-LDS loads, but the two sets are distinct __shared__ objects and the chunk loop is unrolled by two, so that every
// access names its object at compile time.  Why it matters: SIInsertWaitcnts makes a ds_read wait (vmcnt) for every outstanding
// buffer_load ... lds that MAY alias it; with one dynamic LDS block carved into sets it cannot tell the window being filled from
// the window being gathered and serialises them (ISA of variant 8: s_waitcnt vmcnt(3..0) in front of the first gathers of every
// chunk); with distinct objects it emits the loads and the gathers back to back (checked on the gfx950 ISA).  The window budget
// is a compile-time constant here (PTEX texels per view); NOT measured yet.
// ------------------------------------------------------------------------------------------
template <int C, int DKB, bool FAST, int PTEX>
__global__ __launch_bounds__(256) void warp_variance_pss_kernel(
    const float* __restrict__ feats, const float* __restrict__ rot, const float* __restrict__ trans,
    const float* __restrict__ planes, float* __restrict__ var, int V, int D, int h, int w, int tiles_x) {
#pragma clang fp contract(off)
    constexpr int NV = 2;
    constexpr int LPP = C / 4;
    constexpr int PIX = 256 / LPP;
    constexpr int TH = 4, TW = PIX / TH;
    constexpr int GRP = 256 / PIX;
    constexpr int KPT = (DKB + GRP - 1) / GRP;
    constexpr int C4 = C * 4;
    constexpr int TAB = NV * DKB * PIX;
    constexpr int PATCH_BYTES = PTEX * C4;
    __shared__ __attribute__((aligned(16))) v4i tab_o0[TAB];
    __shared__ __attribute__((aligned(16))) v4i tab_o1[TAB];
    __shared__ __attribute__((aligned(16))) v4f tab_w0[TAB];
    __shared__ __attribute__((aligned(16))) v4f tab_w1[TAB];
    __shared__ __attribute__((aligned(16))) int box0[16 * NV];
    __shared__ __attribute__((aligned(16))) int box1[16 * NV];
    __shared__ __attribute__((aligned(16))) char patch0[NV * PATCH_BYTES];
    __shared__ __attribute__((aligned(16))) char patch1[NV * PATCH_BYTES];
    const int nv = V - 1;

    const int b = blockIdx.z;
    const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tx0 = (tile % tiles_x) * TW, ty0 = (tile / tiles_x) * TH;
    const int hw = h * w;
    const int nch = (D + DKB - 1) / DKB;
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
    const int pa = threadIdx.x % PIX, ga = threadIdx.x / PIX;
    const int xa = min(tx0 + pa % TW, w - 1), ya = min(ty0 + pa / TW, h - 1);
    const float fxa = (float)xa, fya = (float)ya;
    const float2 pla = reinterpret_cast<const float2*>(planes)[(long long)b * hw + ya * w + xa];
    const int lane = threadIdx.x & 63;

    // every phase takes the arrays of ITS set as arguments; after inlining they are compile-time objects
    auto phase_a = [&](int kc, v4i* lo, v4f* lw, int* lbox) {
        const int k0 = kc * DKB;
        for (int va = 0; va < nv; ++va) {
            const float* r = rot + ((long long)b * (V - 1) + va) * 9;
            const float* t = trans + ((long long)b * (V - 1) + va) * 3;
            const float rx = (r[0] * fxa + r[1] * fya) + r[2];
            const float ry = (r[3] * fxa + r[4] * fya) + r[5];
            const float rz = (r[6] * fxa + r[7] * fya) + r[8];
            const float t0 = t[0], t1 = t[1], t2 = t[2];
            int bx0 = 0x7fffffff, bx1 = -1, by0 = 0x7fffffff, by1 = -1;
#pragma unroll
            for (int kk = 0; kk < KPT; ++kk) {
                const int ka = ga + kk * GRP;
                if (ka >= DKB) continue;
                const float d = pla.x + (float)(k0 + ka) * pla.y;
                v4f wt;
                int xi, yi;
                k1_chain(rx, ry, rz, t0, t1, t2, d, g, xi, yi, wt);
                const int xc0 = min(max(xi, 0), w - 1), xc1 = min(max(xi + 1, 0), w - 1);
                const int yc0 = min(max(yi, 0), h - 1), yc1 = min(max(yi + 1, 0), h - 1);
                const bool any = (wt.x != 0.0f) || (wt.y != 0.0f) || (wt.z != 0.0f) || (wt.w != 0.0f);
                const int idx = (va * DKB + ka) * PIX + pa;
                v4i rec;
                rec.x = xc0; rec.y = yc0; rec.z = (xc1 != xc0 ? 1 : 0) | (yc1 != yc0 ? 2 : 0) | (any ? 4 : 0); rec.w = 0;
                lo[idx] = rec;
                lw[idx] = wt;
                if (any) { bx0 = min(bx0, xc0); bx1 = max(bx1, xc1); by0 = min(by0, yc0); by1 = max(by1, yc1); }
            }
            bx0 = wave_reduce_i32<true>(bx0); bx1 = wave_reduce_i32<false>(bx1);
            by0 = wave_reduce_i32<true>(by0); by1 = wave_reduce_i32<false>(by1);
            if (lane == 0) *reinterpret_cast<v4i*>(lbox + ((threadIdx.x >> 6) * NV + va) * 4) = (v4i){bx0, bx1, by0, by1};
        }
    };

    // records -> byte offsets inside this set's window array (or global offsets), windows -> LDS by direct loads
    auto stage = [&](v4i* lo, const int* lbox, char* lpatch, bool* fits) {
        int px0[NV], py0[NV], pw[NV], ph[NV];
#pragma unroll
        for (int va = 0; va < NV; ++va) {
            v4i bb = *reinterpret_cast<const v4i*>(lbox + va * 4);
#pragma unroll
            for (int wv = 1; wv < 4; ++wv) {
                const v4i t = *reinterpret_cast<const v4i*>(lbox + (wv * NV + va) * 4);
                bb.x = min(bb.x, t.x); bb.y = max(bb.y, t.y); bb.z = min(bb.z, t.z); bb.w = max(bb.w, t.w);
            }
            px0[va] = bb.x; py0[va] = bb.z;
            pw[va] = bb.y - bb.x + 1; ph[va] = bb.w - bb.z + 1;
            if (bb.y < 0) { px0[va] = 0; py0[va] = 0; pw[va] = 1; ph[va] = 1; }
            fits[va] = (va < nv) && (pw[va] * ph[va] <= PTEX);
        }
#pragma unroll
        for (int va = 0; va < NV; ++va) {
            if (va >= nv) continue;
#pragma unroll
            for (int kk = 0; kk < KPT; ++kk) {
                if (ga + kk * GRP >= DKB) continue;
                const int idx = (va * DKB + ga + kk * GRP) * PIX + pa;
                const v4i rec = lo[idx];
                const bool any = rec.z & 4;
                v4i o;
                if (fits[va]) {
                    const int lx = any ? rec.x - px0[va] : 0, ly = any ? rec.y - py0[va] : 0;
                    const int base = (ly * pw[va] + lx) * C4 + va * PATCH_BYTES;
                    const int dx = (any && (rec.z & 1)) ? C4 : 0, dy = (any && (rec.z & 2)) ? pw[va] * C4 : 0;
                    o.x = base; o.y = base + dx; o.z = base + dy; o.w = base + dy + dx;
                } else {
                    const int base = ((1 + va) * hw + rec.y * w + rec.x) * C4;
                    const int dx = (rec.z & 1) ? C4 : 0, dy = (rec.z & 2) ? w * C4 : 0;
                    o.x = base; o.y = base + dx; o.z = base + dy; o.w = base + dy + dx;
                }
                lo[idx] = o;
            }
        }
#pragma unroll
        for (int va = 0; va < NV; ++va) {
            if (!fits[va]) continue;
            const int row4 = pw[va] * LPP;
            const int n4 = row4 * ph[va];
            const int src0 = (((1 + va) * hw) + py0[va] * w + px0[va]) * C4;
            char* dst = lpatch + va * PATCH_BYTES;
            for (int e0 = (threadIdx.x & ~63); e0 < n4; e0 += 256) {
                const int e = e0 + lane;
                if (e < n4) {
                    const int row = e / row4, c4 = e - row * row4;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst + (size_t)e0 * 16, 16, src0 + (row * w * C + c4 * 4) * 4, 0, 0, 0);
                }
            }
        }
    };

    // Two bodies: when every view's window fits (the block-uniform common case) the gathers are ds_reads only.  The mixed body
    // also issues global gathers for a view whose window did not fit; keeping it apart matters because its buffer loads share
    // destination registers with the ds_reads, which makes the compiler drain vmcnt -- and with it the windows in flight.
    auto phase_b_body = [&](int kc, const v4i* lo, const v4f* lw, const char* lpatch, const bool* fits, auto mixed) {
        constexpr bool MIXED = decltype(mixed)::value;
        const int k0 = kc * DKB;
        const char* pq = lpatch + q4b;
#pragma unroll
        for (int k = 0; k < DKB; ++k) {
            v4f a = ref, a2 = ref * ref;
#pragma unroll
            for (int va = 0; va < NV; ++va) {
                if (va >= nv) continue;
                const int idx = (va * DKB + k) * PIX + p;
                const v4i o = lo[idx];
                const v4f wt = lw[idx];
                v4f ta, tb, tc, td;
                if (!MIXED || fits[va]) {
                    ta = *reinterpret_cast<const v4f*>(pq + o.x);
                    tb = *reinterpret_cast<const v4f*>(pq + o.y);
                    tc = *reinterpret_cast<const v4f*>(pq + o.z);
                    td = *reinterpret_cast<const v4f*>(pq + o.w);
                } else {
                    ta = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.x + q4b, 0, 0));
                    tb = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.y + q4b, 0, 0));
                    tc = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.z + q4b, 0, 0));
                    td = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.w + q4b, 0, 0));
                }
                v4f val = blend4<FAST>(ta, tb, tc, td, wt);
                a = a + val;
                if (FAST) a2 = __builtin_elementwise_fma(val, val, a2); else a2 = a2 + val * val;
            }
            if (k0 + k < D) k1_store_variance<FAST>(a, a2, fV, rV, ob + (long long)(k0 + k) * hw * C);
        }
    };
    auto phase_b = [&](int kc, const v4i* lo, const v4f* lw, const char* lpatch, const bool* fits) {
        if (!inside) return;
        bool all = true;
#pragma unroll
        for (int va = 0; va < NV; ++va) all = all && (va >= nv || fits[va]);
        if (all) phase_b_body(kc, lo, lw, lpatch, fits, std::false_type{});
        else phase_b_body(kc, lo, lw, lpatch, fits, std::true_type{});
    };

    bool fits0[NV], fits1[NV];
    phase_a(0, tab_o0, tab_w0, box0);
    __syncthreads();
    stage(tab_o0, box0, patch0, fits0);
    for (int kc = 0; kc < nch; kc += 2) {
        // ---- even chunk: blend set 0 while set 1 is prepared
        if (kc + 1 < nch) phase_a(kc + 1, tab_o1, tab_w1, box1);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (kc + 1 < nch) stage(tab_o1, box1, patch1, fits1);
        phase_b(kc, tab_o0, tab_w0, patch0, fits0);
        __syncthreads();
        if (kc + 1 >= nch) break;
        // ---- odd chunk: blend set 1 while set 0 is prepared
        if (kc + 2 < nch) phase_a(kc + 2, tab_o0, tab_w0, box0);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (kc + 2 < nch) stage(tab_o0, box0, patch0, fits0);
        phase_b(kc + 1, tab_o1, tab_w1, patch1, fits1);
        __syncthreads();
    }
}

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

static int g_k1_variant = 0;     // profiling hook (rcmvs_debug_k1_variant)
static int g_k1_ps_dkb = 0, g_k1_ps_ptex = 0, g_k1_ps_pad = 0;     // tuning knobs of the pipelined staged variant (rcmvs_debug_k1_ps_config), 0 = default

extern "C" {

void rcmvs_debug_k1_variant(int v) { g_k1_variant = v; }
void rcmvs_debug_k1_ps_config(int dkb, int patch_texels, int texel_pad_bytes) { g_k1_ps_dkb = dkb; g_k1_ps_ptex = patch_texels; g_k1_ps_pad = texel_pad_bytes; }

int rcmvs_warp_variance_fwd(const float* feats, const float* rot, const float* trans,
                            const float* planes, float* var,
                            int B, int V, int C, int D, int h, int w, void* stream) {
    RCMVS_REQUIRE(feats && rot && trans && planes && var, "warp_variance_fwd: null pointer");
    RCMVS_REQUIRE(B > 0 && D > 0 && h > 1 && w > 1, "warp_variance_fwd: bad sizes B=%d D=%d h=%d w=%d", B, D, h, w);
    RCMVS_REQUIRE(V >= 2 && V - 1 <= RCMVS_MAX_SRC_VIEWS, "warp_variance_fwd: V=%d unsupported", V);
    RCMVS_REQUIRE((long long)h * w * C < (1LL << 31), "warp_variance_fwd: feature map too large for 32-bit offsets");
    const int TW = 256 / C, TH = 4;
    const int tiles_x = (w + TW - 1) / TW, tiles_y = (h + TH - 1) / TH;
    dim3 grid(tiles_x * tiles_y, (D + DK - 1) / DK, B);
    hipStream_t st = as_stream(stream);
    if (g_k1_variant >= 4 && g_k1_variant <= 7) {
        // LDS-staged kernel: bit0 = FMA blend, bit1 = deeper plane chunk
        const bool fastm = (g_k1_variant - 4) & 1, deep = (g_k1_variant - 4) & 2;
        const int LPP = C / 4, PIX = 256 / LPP;
        const int nvk = (V - 1) >= 2 ? 2 : 1;
        const int dkb = (C == 8) ? (deep ? 4 : 2) : (deep ? 8 : 4);
        const int ptex = (C == 32) ? (deep ? 192 : 128) : (C == 16 ? (deep ? 320 : 224) : (deep ? 512 : 384));
        const size_t lds = (size_t)nvk * dkb * PIX * 32 + 64 * nvk + (size_t)nvk * ptex * C * 4;
        RCMVS_REQUIRE(h <= 8191 && w <= 8191, "warp_variance_fwd: map too large");
        RCMVS_REQUIRE((long long)V * h * w * C * 4 < 0x7fffffffLL, "warp_variance_fwd: feature block too large for 32-bit offsets");
        const int TWl = PIX / 4;
        const int txl = (w + TWl - 1) / TWl, tyl = (h + 3) / 4;
        dim3 gridl(txl * tyl, (D + dkb - 1) / dkb, B);
#define RCMVS_K1L(CC, DD, FF, NN) do { \
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)warp_variance_lds_kernel<CC, DD, FF, NN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((warp_variance_lds_kernel<CC, DD, FF, NN>), gridl, dim3(256), lds, st, feats, rot, trans, planes, var, V, D, h, w, txl, ptex); } while (0)
#define RCMVS_K1L_N(CC, DD, FF) do { if (nvk == 2) RCMVS_K1L(CC, DD, FF, 2); else RCMVS_K1L(CC, DD, FF, 1); } while (0)
#define RCMVS_K1L_F(CC, DD) do { if (fastm) RCMVS_K1L_N(CC, DD, true); else RCMVS_K1L_N(CC, DD, false); } while (0)
        switch (C) {
            case 8:  if (deep) RCMVS_K1L_F(8, 4); else RCMVS_K1L_F(8, 2); break;
            case 16: if (deep) RCMVS_K1L_F(16, 8); else RCMVS_K1L_F(16, 4); break;
            case 32: if (deep) RCMVS_K1L_F(32, 8); else RCMVS_K1L_F(32, 4); break;
            default: return fail(-1, "warp_variance_fwd: C must be 8, 16 or 32 (got %d)", C);
        }
        return launch_status("warp_variance_fwd(lds)");
    }
    if (g_k1_variant == 12 || g_k1_variant == 13) {
        // pipelined staged kernel with static LDS sets (compile-time window budget): 12 = exact, 13 = FMA blend; <= 2 source views
        const bool fastm = g_k1_variant == 13;
        RCMVS_REQUIRE(V - 1 <= 2, "warp_variance_fwd: debug variant %d handles at most 2 source views (V=%d)", g_k1_variant, V);
        RCMVS_REQUIRE(h <= 8191 && w <= 8191, "warp_variance_fwd: map too large");
        RCMVS_REQUIRE((long long)V * h * w * C * 4 < 0x7fffffffLL, "warp_variance_fwd: feature block too large for 32-bit offsets");
        const int dkb = g_k1_ps_dkb ? g_k1_ps_dkb : (C == 8 ? 2 : 4);        // C = 8: 128 pixels per block, the 4-plane tap tables alone are 64 KB
        RCMVS_REQUIRE(dkb == 2 || dkb == 4, "warp_variance_fwd: static pipelined variant: plane chunk %d (2 or 4)", dkb);
        const int PIXs = 256 / (C / 4), TWs = PIXs / 4;
        const int txs = (w + TWs - 1) / TWs, tys = (h + 3) / 4;
        dim3 grids(txs * tys, 1, B);
        // window budgets (texels per view): two sets of tables + windows stay under 80 KB so that two blocks share a CU
#define RCMVS_K1PSS(CC, DD, FF, PP) hipLaunchKernelGGL((warp_variance_pss_kernel<CC, DD, FF, PP>), grids, dim3(256), 0, st, feats, rot, trans, planes, var, V, D, h, w, txs)
#define RCMVS_K1PSS_F(CC, DD, PP) do { if (fastm) RCMVS_K1PSS(CC, DD, true, PP); else RCMVS_K1PSS(CC, DD, false, PP); } while (0)
        switch (C) {
            case 8:  if (dkb == 2) RCMVS_K1PSS_F(8, 2, 320); else RCMVS_K1PSS_F(8, 4, 256); break;
            case 16: if (dkb == 2) RCMVS_K1PSS_F(16, 2, 224); else RCMVS_K1PSS_F(16, 4, 160); break;
            case 32:
                // K1_PS_PTEX <= 72 selects the 72-texel budget: 16 KB of tables + 36 KB of windows = three blocks per CU; the window
                // statistics put the median stage-1 window of a 4 x 8 tile over 4 planes at ~48 texels (profiles/r1_k1_window_stats.txt)
                if (dkb == 2) RCMVS_K1PSS_F(32, 2, 128);
                else if (g_k1_ps_ptex > 0 && g_k1_ps_ptex <= 72) RCMVS_K1PSS_F(32, 4, 72);
                else RCMVS_K1PSS_F(32, 4, 112);
                break;
            default: return fail(-1, "warp_variance_fwd: C must be 8, 16 or 32 (got %d)", C);
        }
        return launch_status("warp_variance_fwd(pss)");
    }
    if (g_k1_variant >= 8 && g_k1_variant <= 11) {
        // pipelined staged kernel (persistent over the plane chunks of a tile): 8 = exact, 9 = FMA blend, windows loaded straight
        // into LDS; 10 / 11 = the same with the windows held in registers across the blend (no buffer_load ... lds); <= 2 source views
        const bool fastm = g_k1_variant & 1, dma = g_k1_variant < 10;
        RCMVS_REQUIRE(V - 1 <= 2, "warp_variance_fwd: debug variant %d handles at most 2 source views (V=%d)", g_k1_variant, V);
        RCMVS_REQUIRE(h <= 8191 && w <= 8191, "warp_variance_fwd: map too large");
        RCMVS_REQUIRE((long long)V * h * w * C * 4 < 0x7fffffffLL, "warp_variance_fwd: feature block too large for 32-bit offsets");
        const int LPP = C / 4, PIX = 256 / LPP;
        const int dkb = g_k1_ps_dkb ? g_k1_ps_dkb : 4;
        const int ptex = g_k1_ps_ptex ? g_k1_ps_ptex : ((C == 32) ? 128 : (C == 16 ? 224 : 384));
        RCMVS_REQUIRE(dkb == 2 || dkb == 4 || dkb == 8, "warp_variance_fwd: pipelined variant: plane chunk %d (2, 4 or 8)", dkb);
        RCMVS_REQUIRE(ptex >= 16 && (ptex * C * 4) % 16 == 0, "warp_variance_fwd: pipelined variant: window budget %d texels", ptex);
        const int pad = dma ? 0 : g_k1_ps_pad;                  // direct-to-LDS loads land contiguously: padding only in the register-held form
        RCMVS_REQUIRE(g_k1_ps_pad == 0 || g_k1_ps_pad == 16 || g_k1_ps_pad == 32, "warp_variance_fwd: pipelined variant: texel padding %d (0, 16 or 32 bytes)", g_k1_ps_pad);
        const size_t set_bytes = (size_t)2 * dkb * PIX * 32 + 64 * 2 + (size_t)2 * ptex * (C * 4 + pad);
        const size_t lds = 2 * set_bytes;
        RCMVS_REQUIRE(lds <= 160 * 1024, "warp_variance_fwd: pipelined variant needs %zu bytes of LDS (chunk %d, %d texels)", lds, dkb, ptex);
        const int TWl = PIX / 4;
        const int txl = (w + TWl - 1) / TWl, tyl = (h + 3) / 4;
        dim3 gridp(txl * tyl, 1, B);
        RCMVS_REQUIRE(dma || ptex * LPP <= 256 * 4, "warp_variance_fwd: register-staged variant holds at most %d texels per view (asked %d)", 1024 / LPP, ptex);
#define RCMVS_K1PS_M(CC, DD, FF, MM) do { \
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)warp_variance_ps_kernel<CC, DD, FF, 2, MM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((warp_variance_ps_kernel<CC, DD, FF, 2, MM>), gridp, dim3(256), lds, st, feats, rot, trans, planes, var, V, D, h, w, txl, ptex, pad); } while (0)
#define RCMVS_K1PS(CC, DD, FF) do { if (dma) RCMVS_K1PS_M(CC, DD, FF, true); else RCMVS_K1PS_M(CC, DD, FF, false); } while (0)
#define RCMVS_K1PS_D(CC, FF) do { if (dkb == 2) RCMVS_K1PS(CC, 2, FF); else if (dkb == 4) RCMVS_K1PS(CC, 4, FF); else RCMVS_K1PS(CC, 8, FF); } while (0)
#define RCMVS_K1PS_F(CC) do { if (fastm) RCMVS_K1PS_D(CC, true); else RCMVS_K1PS_D(CC, false); } while (0)
        switch (C) {
            case 8:  RCMVS_K1PS_F(8); break;
            case 16: RCMVS_K1PS_F(16); break;
            case 32: RCMVS_K1PS_F(32); break;
            default: return fail(-1, "warp_variance_fwd: C must be 8, 16 or 32 (got %d)", C);
        }
        return launch_status("warp_variance_fwd(ps)");
    }
    if (g_k1_variant == 0 || g_k1_variant == 1) {
        // production kernel: variant 0 = exact arithmetic (default), 1 = FMA-contracted blend
        const bool fastm = g_k1_variant == 1;
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
#define RCMVS_K1TP(CC, DD, FF, NN) hipLaunchKernelGGL((warp_variance_tp_kernel<CC, DD, FF, NN>), gridp, dim3(256), lds, st, feats, rot, trans, planes, var, V, D, h, w, txp, VC)
#define RCMVS_K1TP_N(CC, DD, FF) do { if (nvt == 2) RCMVS_K1TP(CC, DD, FF, 2); else if (nvt == 4) RCMVS_K1TP(CC, DD, FF, 4); else RCMVS_K1TP(CC, DD, FF, 0); } while (0)
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
    // variants 2 (reference-order kernel, one tap computation per lane) and 3 (store-only ablation)
#define RCMVS_K1_LAUNCH(CC, VV) hipLaunchKernelGGL((warp_variance_ref_kernel<CC, VV>), grid, dim3(256), 0, st, feats, rot, trans, planes, var, V, D, h, w, tiles_x, tiles_y)
#define RCMVS_K1_VARIANTS(CC)                                                               \
    switch (g_k1_variant) {                                                                 \
        case 2: RCMVS_K1_LAUNCH(CC, false); break;                                              \
        case 3: RCMVS_K1_LAUNCH(CC, true); break;                                              \
        default: return fail(-1, "warp_variance_fwd: unknown debug variant %d", g_k1_variant); \
    }
    switch (C) {
        case 8:  RCMVS_K1_VARIANTS(8) break;
        case 16: RCMVS_K1_VARIANTS(16) break;
        case 32: RCMVS_K1_VARIANTS(32) break;
        default: return fail(-1, "warp_variance_fwd: C must be 8, 16 or 32 (got %d)", C);
    }
#undef RCMVS_K1_VARIANTS
#undef RCMVS_K1_LAUNCH
    return launch_status("warp_variance_fwd");
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
