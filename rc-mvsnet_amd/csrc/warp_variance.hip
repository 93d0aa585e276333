// K1: fused plane-sweep homography warp + variance cost volume (gfx950).
//
// Replaces, per stage, (V-1) x homo_warping (models/modules.py:304-339: ~15 materialised
// [B,3,D,hw] intermediates + grid_sample) and the sum / square-sum / variance chain of
// DepthNet_eval.forward (models/casmvsnet.py:257-288).  One launch reads the V feature maps and
// the plane table and writes the variance volume exactly once.
//
// Mapping (wave = 64 lanes): channels-last everywhere.  A pixel's C channels are handled by
// C/4 adjacent lanes (one float4 each), a wave covers 1024 B of contiguous output per plane
// (256/C pixels of one image row), a 256-thread block covers a 4-row tile, and each thread keeps
// DK = 8 planes x 4 channels of {sum, square-sum} in registers while it loops over the source
// views -- so every bilinear tap is a 16-byte load of 4 channels and the output store of a wave
// is one fully coalesced 1 KiB line.
//
// Numerics: the coordinate chain and the accumulation are compiled with fp contraction OFF and
// IEEE division, in the operation order of oracle/warp.py (itself the reference's op order), so
// the kernel is bit-comparable with the oracle; taps outside the source image, and non-finite
// coordinates (z == 0), contribute zero (grid_sample zeros padding, CUDA/HIP semantics).
#include "common.h"

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

// VARIANT: 0 = one tap computation per (lane, plane, view), exact arithmetic (bit-comparable with
//              the oracle);
//          4 = taps computed once per (pixel, plane, view) by ONE of the pixel's C/4 lanes and
//              broadcast to the others with ds_bpermute (cross-lane, no LDS storage), exact;
//          5 = 4 with FMA contraction in the bilinear blend and Markstein-style division by V
//              (q = x*r; q += fma(-V,q,x)*r) instead of the IEEE sequence -- <= 1 ulp from 4;
//          1,2,3 = ablations for profiling only (no gathers / no coordinate math / store only).
template <bool FAST>
__device__ __forceinline__ float4 bilerp4v(const float* __restrict__ src, int o0, int o1, int o2, int o3,
                                           float w0, float w1, float w2, float w3) {
    float4 a = *reinterpret_cast<const float4*>(src + o0);
    float4 b = *reinterpret_cast<const float4*>(src + o1);
    float4 c = *reinterpret_cast<const float4*>(src + o2);
    float4 d = *reinterpret_cast<const float4*>(src + o3);
    float4 r;
    if (FAST) {
        r.x = fmaf(d.x, w3, fmaf(c.x, w2, fmaf(b.x, w1, a.x * w0)));
        r.y = fmaf(d.y, w3, fmaf(c.y, w2, fmaf(b.y, w1, a.y * w0)));
        r.z = fmaf(d.z, w3, fmaf(c.z, w2, fmaf(b.z, w1, a.z * w0)));
        r.w = fmaf(d.w, w3, fmaf(c.w, w2, fmaf(b.w, w1, a.w * w0)));
    } else {
#pragma clang fp contract(off)
        r.x = ((a.x * w0 + b.x * w1) + c.x * w2) + d.x * w3;
        r.y = ((a.y * w0 + b.y * w1) + c.y * w2) + d.y * w3;
        r.z = ((a.z * w0 + b.z * w1) + c.z * w2) + d.z * w3;
        r.w = ((a.w * w0 + b.w * w1) + c.w * w2) + d.w * w3;
    }
    return r;
}

template <bool FAST>
__device__ __forceinline__ float div_by(float x, float fV, float rV) {
    if (FAST) {
        float q = x * rV;
        return fmaf(fmaf(-fV, q, x), rV, q);
    }
    return x / fV;
}

template <int C, int VARIANT>
__global__ __launch_bounds__(256) void warp_variance_kernel(
    const float* __restrict__ feats, const float* __restrict__ rot, const float* __restrict__ trans,
    const float* __restrict__ planes, float* __restrict__ var, int V, int D, int h, int w, int tiles_x, int tiles_y) {
#pragma clang fp contract(off)
    constexpr int LPP = C / 4;          // lanes per pixel
    constexpr int TW = 256 / C;         // pixels per wave = tile width  (1 KiB of output per plane)
    constexpr int TH = 4;               // one wave per tile row
    constexpr bool SHARED = (VARIANT == 4 || VARIANT == 5);
    constexpr bool FAST = (VARIANT == 5);
    constexpr int ROUNDS = DK / LPP;    // planes whose taps one lane computes per view (SHARED)
    const int b = blockIdx.z;
    const int k0 = blockIdx.y * DK;
    const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int lq = threadIdx.x % LPP;
    const int q4 = lq * 4;
    int x = tx * TW + (threadIdx.x / LPP) % TW;
    int y = ty * TH + threadIdx.x / (LPP * TW);
    const bool inside = (x < w) && (y < h);
    if (!SHARED && !inside) return;
    if (!inside) { x = w - 1; y = h - 1; }                     // SHARED: keep every lane alive for the shuffles
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
    const int lane = threadIdx.x & 63;
    const int grp = lane - lq;                                  // first lane of this pixel's group
    for (int v = 1; v < V; ++v) {
        const float* r = rot + ((long long)b * (V - 1) + (v - 1)) * 9;
        const float* t = trans + ((long long)b * (V - 1) + (v - 1)) * 3;
        const float rx = (r[0] * fx + r[1] * fy) + r[2];
        const float ry = (r[3] * fx + r[4] * fy) + r[5];
        const float rz = (r[6] * fx + r[7] * fy) + r[8];
        const float t0 = t[0], t1 = t[1], t2 = t[2];
        const float* src = fb + (long long)v * hw * C;
        if (SHARED) {
            // lane lq of the group computes planes k0 + lq + LPP*rr
            int pk[ROUNDS];
            float w0[ROUNDS], w1[ROUNDS], w2[ROUNDS], w3[ROUNDS];
#pragma unroll
            for (int rr = 0; rr < ROUNDS; ++rr) {
                const float d = pl.x + (float)(k0 + lq + LPP * rr) * pl.y;
                WarpCoord tc = warp_taps(rx, ry, rz, t0, t1, t2, d, half_w, half_h, wm1, hm1, w, h, C);
                // off[1]-off[0] is 0 or C, off[2]-off[0] is 0 or w*C: two flag bits in the low bits of the
                // base offset (a multiple of C >= 8)
                pk[rr] = tc.off[0] | (tc.off[1] != tc.off[0] ? 1 : 0) | (tc.off[2] != tc.off[0] ? 2 : 0);
                w0[rr] = tc.wgt[0]; w1[rr] = tc.wgt[1]; w2[rr] = tc.wgt[2]; w3[rr] = tc.wgt[3];
            }
#pragma unroll
            for (int k = 0; k < DK; ++k) {
                const int rr = k / LPP, sl = grp + (k % LPP);
                const int pkk = __shfl(pk[rr], sl);
                const float a0 = __shfl(w0[rr], sl), a1 = __shfl(w1[rr], sl), a2 = __shfl(w2[rr], sl), a3 = __shfl(w3[rr], sl);
                const int o0 = (pkk & ~3) + q4;
                const int dx = (pkk & 1) ? C : 0, dy = (pkk & 2) ? w * C : 0;
                float4 val = bilerp4v<FAST>(src, o0, o0 + dx, o0 + dy, o0 + dy + dx, a0, a1, a2, a3);
                if (FAST) {
                    s[k].x += val.x; s[k].y += val.y; s[k].z += val.z; s[k].w += val.w;
                    sq[k].x = fmaf(val.x, val.x, sq[k].x); sq[k].y = fmaf(val.y, val.y, sq[k].y);
                    sq[k].z = fmaf(val.z, val.z, sq[k].z); sq[k].w = fmaf(val.w, val.w, sq[k].w);
                } else {
                    s[k].x = s[k].x + val.x; s[k].y = s[k].y + val.y; s[k].z = s[k].z + val.z; s[k].w = s[k].w + val.w;
                    sq[k].x = sq[k].x + val.x * val.x; sq[k].y = sq[k].y + val.y * val.y;
                    sq[k].z = sq[k].z + val.z * val.z; sq[k].w = sq[k].w + val.w * val.w;
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < DK; ++k) {
                float4 val;
                if (VARIANT == 3) {
                    val = ref;
                } else {
                    const float d = pl.x + (float)(k0 + k) * pl.y;
                    WarpCoord tc;
                    if (VARIANT == 2) {                         // ablation: no coordinate math
                        int o = ((y * w + x) * C);
                        tc.off[0] = o; tc.off[1] = o; tc.off[2] = o; tc.off[3] = o;
                        tc.wgt[0] = 0.25f + d * 0.0f; tc.wgt[1] = 0.25f; tc.wgt[2] = 0.25f; tc.wgt[3] = 0.25f;
                    } else {
                        tc = warp_taps(rx, ry, rz, t0, t1, t2, d, half_w, half_h, wm1, hm1, w, h, C);
                    }
                    if (VARIANT == 1) {                         // ablation: no gathers
                        float sw = ((tc.wgt[0] + tc.wgt[1]) + tc.wgt[2]) + tc.wgt[3] + (float)(tc.off[0] & 1);
                        val = make_float4(ref.x * sw, ref.y * sw, ref.z * sw, ref.w * sw);
                    } else {
                        val = bilerp4(src, tc, q4);
                    }
                }
                s[k].x = s[k].x + val.x; s[k].y = s[k].y + val.y; s[k].z = s[k].z + val.z; s[k].w = s[k].w + val.w;
                sq[k].x = sq[k].x + val.x * val.x; sq[k].y = sq[k].y + val.y * val.y;
                sq[k].z = sq[k].z + val.z * val.z; sq[k].w = sq[k].w + val.w * val.w;
            }
        }
    }
    if (!inside) return;
    const float fV = (float)V, rV = 1.0f / fV;
    float* ob = var + (((long long)b * D) * hw + (long long)y * w + x) * C + q4;
#pragma unroll
    for (int k = 0; k < DK; ++k) {
        if (k0 + k < D) {
            float4 m, o;
            m.x = div_by<FAST>(s[k].x, fV, rV); m.y = div_by<FAST>(s[k].y, fV, rV);
            m.z = div_by<FAST>(s[k].z, fV, rV); m.w = div_by<FAST>(s[k].w, fV, rV);
            o.x = div_by<FAST>(sq[k].x, fV, rV) - m.x * m.x;
            o.y = div_by<FAST>(sq[k].y, fV, rV) - m.y * m.y;
            o.z = div_by<FAST>(sq[k].z, fV, rV) - m.z * m.z;
            o.w = div_by<FAST>(sq[k].w, fV, rV) - m.w * m.w;
            *reinterpret_cast<float4*>(ob + (long long)(k0 + k) * hw * C) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------
// K1 v6: two-phase, plane-major.
//   Phase A  every (pixel, plane, view) of the block's tile is handled by exactly one thread: the
//            full coordinate chain runs once (not once per channel lane) and leaves
//            {packed base offset, 4 masked weights} in LDS (structure-of-arrays, conflict-free).
//   Phase B  thread = (pixel, channel quad); for each plane: start from the reference value, add
//            every source view's bilinear sample (tap data broadcast-read from LDS, four 16-byte
//            gathers), form the variance and STORE THE PLANE IMMEDIATELY -- stores are spread over
//            the whole kernel and only 8 accumulator registers are live, so many waves fit and the
//            HBM write stream overlaps the arithmetic of other waves.
// Division: IEEE-correct quotients without the compiler's div_scale/div_fmas/div_fixup sequence
//   (and its denormal-mode switches): v_rcp_f32 refined by one Newton step, then two
//   fma-residual corrections of the quotient (Markstein).  The operands here are far from the
//   exponent limits, which is all the omitted scaling protects against; b == 0 yields NaN and
//   the tap is dropped exactly like the reference's inf coordinate.
// Source views are processed in chunks of VC (LDS budget); with more than one chunk the per-plane
//   sums persist in registers across chunks (T&T, 6 source views at C = 8).
// ------------------------------------------------------------------------------------------
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float rcp_nr(float b) {
    float r = __builtin_amdgcn_rcpf(b);
    float e = fmaf(-b, r, 1.0f);
    return fmaf(e, r, r);
}
// correctly rounded a / b given r ~ 1/b (rcp_nr)
__device__ __forceinline__ float div_cr(float a, float b, float r) {
    float q = a * r;
    float rem = fmaf(-b, q, a);
    q = fmaf(rem, r, q);
    rem = fmaf(-b, q, a);
    return fmaf(rem, r, q);
}

struct TapPack { int pk; float w0, w1, w2, w3; };

__device__ __forceinline__ TapPack warp_taps_fastdiv(float rx, float ry, float rz, float tx, float ty, float tz, float d,
                                                    float half_w, float half_h, float r_half_w, float r_half_h,
                                                    float wm1, float hm1, int w, int C) {
#pragma clang fp contract(off)
    float px = rx * d + tx;
    float py = ry * d + ty;
    float pz = rz * d + tz;
    float rpz = rcp_nr(pz);
    float u = div_cr(px, pz, rpz);
    float v = div_cr(py, pz, rpz);
    float gx = div_cr(u, half_w, r_half_w) - 1.0f;
    float gy = div_cr(v, half_h, r_half_h) - 1.0f;
    float ix = ((gx + 1.0f) * 0.5f) * wm1;
    float iy = ((gy + 1.0f) * 0.5f) * hm1;
    float x0 = floorf(ix), y0 = floorf(iy);
    float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
    float wx1 = ix - x0, wx0 = x1 - ix;
    float wy1 = iy - y0, wy0 = y1 - iy;
    bool vx0 = (x0 >= 0.0f) && (x0 <= wm1);
    bool vx1 = (x1 >= 0.0f) && (x1 <= wm1);
    bool vy0 = (y0 >= 0.0f) && (y0 <= hm1);
    bool vy1 = (y1 >= 0.0f) && (y1 <= hm1);
    int xi0 = (int)fminf(fmaxf(x0, 0.0f), wm1);
    int xi1 = (int)fminf(fmaxf(x1, 0.0f), wm1);
    int yi0 = (int)fminf(fmaxf(y0, 0.0f), hm1);
    int yi1 = (int)fminf(fmaxf(y1, 0.0f), hm1);
    TapPack t;
    // base offset is a multiple of C >= 8: two flag bits ride in its low bits (x step, y step)
    t.pk = ((yi0 * w + xi0) * C) | (xi1 != xi0 ? 1 : 0) | (yi1 != yi0 ? 2 : 0);
    t.w0 = (vx0 && vy0) ? wx0 * wy0 : 0.0f;
    t.w1 = (vx1 && vy0) ? wx1 * wy0 : 0.0f;
    t.w2 = (vx0 && vy1) ? wx0 * wy1 : 0.0f;
    t.w3 = (vx1 && vy1) ? wx1 * wy1 : 0.0f;
    return t;
}

template <int C, int TH, bool NT, bool FAST, bool MULTI>
__global__ __launch_bounds__(256) void warp_variance_v6_kernel(
    const float* __restrict__ feats, const float* __restrict__ rot, const float* __restrict__ trans,
    const float* __restrict__ planes, float* __restrict__ var, int V, int D, int h, int w, int tiles_x, int VC) {
#pragma clang fp contract(off)
    constexpr int LPP = C / 4;
    constexpr int PIX = 256 / LPP;      // pixels per block
    constexpr int TW = PIX / TH;
    extern __shared__ __attribute__((aligned(16))) int lds_i[];   // [VC][5][DK][PIX]
    float* lds_f = reinterpret_cast<float*>(lds_i);
    const int b = blockIdx.z;
    const int k0 = blockIdx.y * DK;
    const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tx0 = (tile % tiles_x) * TW, ty0 = (tile / tiles_x) * TH;
    const long long hw = (long long)h * w;
    const float wm1 = (float)(w - 1), hm1 = (float)(h - 1);
    const float half_w = wm1 / 2.0f, half_h = hm1 / 2.0f;
    const float r_half_w = rcp_nr(half_w), r_half_h = rcp_nr(half_h);
    const float* fb = feats + (long long)b * V * hw * C;
    const float2* plb = reinterpret_cast<const float2*>(planes) + (long long)b * hw;

    // phase-B identity of this thread
    const int p = threadIdx.x / LPP;
    const int q4 = (threadIdx.x % LPP) * 4;
    const int x = tx0 + p % TW, y = ty0 + p / TW;
    const bool inside = (x < w) && (y < h);
    float4 ref = make_float4(0.f, 0.f, 0.f, 0.f);
    if (inside) ref = *reinterpret_cast<const float4*>(fb + ((long long)y * w + x) * C + q4);
    const float fV = (float)V, rV = rcp_nr(fV);
    float* ob = var + (((long long)b * D) * hw + (long long)y * w + x) * C + q4;

    float4 s[MULTI ? DK : 1], sq[MULTI ? DK : 1];
    if (MULTI) {
#pragma unroll
        for (int k = 0; k < DK; ++k) {
            s[k] = ref;
            sq[k] = make_float4(ref.x * ref.x, ref.y * ref.y, ref.z * ref.z, ref.w * ref.w);
        }
    }

    for (int v0 = 1; v0 < V; v0 += VC) {
        const int nv = min(VC, V - v0);
        if (MULTI && v0 > 1) __syncthreads();                   // LDS reuse across chunks
        // ---------------- phase A: one (pixel, plane, view) per thread-iteration
        for (int i = threadIdx.x; i < PIX * DK * nv; i += 256) {
            const int pa = i % PIX, ka = (i / PIX) % DK, va = i / (PIX * DK);
            int xa = tx0 + pa % TW, ya = ty0 + pa / TW;
            xa = min(xa, w - 1); ya = min(ya, h - 1);
            const float fx = (float)xa, fy = (float)ya;
            const float* r = rot + ((long long)b * (V - 1) + (v0 + va - 1)) * 9;
            const float* t = trans + ((long long)b * (V - 1) + (v0 + va - 1)) * 3;
            const float rx = (r[0] * fx + r[1] * fy) + r[2];
            const float ry = (r[3] * fx + r[4] * fy) + r[5];
            const float rz = (r[6] * fx + r[7] * fy) + r[8];
            const float2 pl = plb[(long long)ya * w + xa];
            const float d = pl.x + (float)(k0 + ka) * pl.y;
            TapPack tp = warp_taps_fastdiv(rx, ry, rz, t[0], t[1], t[2], d, half_w, half_h, r_half_w, r_half_h, wm1, hm1, w, C);
            const int base = ((va * 5) * DK + ka) * PIX + pa;
            lds_i[base] = tp.pk;
            lds_f[base + 1 * DK * PIX] = tp.w0;
            lds_f[base + 2 * DK * PIX] = tp.w1;
            lds_f[base + 3 * DK * PIX] = tp.w2;
            lds_f[base + 4 * DK * PIX] = tp.w3;
        }
        __syncthreads();
        // ---------------- phase B: plane-major accumulation
        if (inside) {
#pragma unroll
            for (int k = 0; k < DK; ++k) {
                float4 a, a2;
                if (MULTI) { a = s[k]; a2 = sq[k]; }
                else { a = ref; a2 = make_float4(ref.x * ref.x, ref.y * ref.y, ref.z * ref.z, ref.w * ref.w); }
                for (int va = 0; va < nv; ++va) {
                    const int base = ((va * 5) * DK + k) * PIX + p;
                    const int pk = lds_i[base];
                    const float w0 = lds_f[base + 1 * DK * PIX], w1 = lds_f[base + 2 * DK * PIX];
                    const float w2 = lds_f[base + 3 * DK * PIX], w3 = lds_f[base + 4 * DK * PIX];
                    const float* src = fb + (long long)(v0 + va) * hw * C;
                    const int o0 = (pk & ~3) + q4;
                    const int dx = (pk & 1) ? C : 0, dy = (pk & 2) ? w * C : 0;
                    float4 val = bilerp4v<FAST>(src, o0, o0 + dx, o0 + dy, o0 + dy + dx, w0, w1, w2, w3);
                    if (FAST) {
                        a.x += val.x; a.y += val.y; a.z += val.z; a.w += val.w;
                        a2.x = fmaf(val.x, val.x, a2.x); a2.y = fmaf(val.y, val.y, a2.y);
                        a2.z = fmaf(val.z, val.z, a2.z); a2.w = fmaf(val.w, val.w, a2.w);
                    } else {
                        a.x = a.x + val.x; a.y = a.y + val.y; a.z = a.z + val.z; a.w = a.w + val.w;
                        a2.x = a2.x + val.x * val.x; a2.y = a2.y + val.y * val.y;
                        a2.z = a2.z + val.z * val.z; a2.w = a2.w + val.w * val.w;
                    }
                }
                if (MULTI && v0 + nv < V) { s[k] = a; sq[k] = a2; continue; }
                if (k0 + k < D) {
                    float mx = div_cr(a.x, fV, rV), my = div_cr(a.y, fV, rV), mz = div_cr(a.z, fV, rV), mw = div_cr(a.w, fV, rV);
                    v4f o;
                    o.x = div_cr(a2.x, fV, rV) - mx * mx;
                    o.y = div_cr(a2.y, fV, rV) - my * my;
                    o.z = div_cr(a2.z, fV, rV) - mz * mz;
                    o.w = div_cr(a2.w, fV, rV) - mw * mw;
                    v4f* dst = reinterpret_cast<v4f*>(ob + (long long)(k0 + k) * hw * C);
                    if (NT) __builtin_nontemporal_store(o, dst); else *dst = o;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// train-variant extra: warped RGB of every source view ++ source-only variance / V, written in
// the reference's NCDHW layout because the tensor crosses the module boundary
// (CascadeMVSNet.forward returns it, models/casmvsnet.py:231).  One thread per (pixel, plane).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void warp_noref_kernel(
    const float* __restrict__ feats, const float* __restrict__ imgs, const float* __restrict__ rot,
    const float* __restrict__ trans, const float* __restrict__ planes, float* __restrict__ out,
    int V, int C, int D, int h, int w, int square_first) {
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
    // channel loop outermost over 4-channel groups keeps registers small; taps are recomputed
    // per view only once (they do not depend on the channel)
    for (int v = 1; v < V; ++v) {
        const float* r = rot + ((long long)b * (V - 1) + (v - 1)) * 9;
        const float* t = trans + ((long long)b * (V - 1) + (v - 1)) * 3;
        const float rx = (r[0] * fx + r[1] * fy) + r[2];
        const float ry = (r[3] * fx + r[4] * fy) + r[5];
        const float rz = (r[6] * fx + r[7] * fy) + r[8];
        WarpCoord tc = warp_taps(rx, ry, rz, t[0], t[1], t[2], d, half_w, half_h, wm1, hm1, w, h, 1);
        const float* im = imgs + ((long long)b * V + v) * hw * 3;
        for (int c = 0; c < 3; ++c) {
            float val = ((im[tc.off[0] * 3 + c] * tc.wgt[0] + im[tc.off[1] * 3 + c] * tc.wgt[1]) +
                         im[tc.off[2] * 3 + c] * tc.wgt[2]) + im[tc.off[3] * 3 + c] * tc.wgt[3];
            ob[(long long)((v - 1) * 3 + c) * cs] = val;
        }
    }
    for (int c = 0; c < C; ++c) {
        float s = 0.0f, sq = 0.0f;
        for (int v = 1; v < V; ++v) {
            const float* r = rot + ((long long)b * (V - 1) + (v - 1)) * 9;
            const float* t = trans + ((long long)b * (V - 1) + (v - 1)) * 3;
            const float rx = (r[0] * fx + r[1] * fy) + r[2];
            const float ry = (r[3] * fx + r[4] * fy) + r[5];
            const float rz = (r[6] * fx + r[7] * fy) + r[8];
            WarpCoord tc = warp_taps(rx, ry, rz, t[0], t[1], t[2], d, half_w, half_h, wm1, hm1, w, h, 1);
            const float* src = feats + ((long long)b * V + v) * hw * C + c;
            float val = ((src[(long long)tc.off[0] * C] * tc.wgt[0] + src[(long long)tc.off[1] * C] * tc.wgt[1]) +
                         src[(long long)tc.off[2] * C] * tc.wgt[2]) + src[(long long)tc.off[3] * C] * tc.wgt[3];
            if (square_first) val = val * val;
            s = s + val;
            sq = sq + val * val;
        }
        float m = s / fV;
        ob[(long long)(3 * (V - 1) + c) * cs] = sq / fV - m * m;
    }
}

}  // namespace rcmvs

using namespace rcmvs;

static int g_k1_variant = 0;     // profiling hook (rcmvs_debug_k1_variant)

extern "C" {

void rcmvs_debug_k1_variant(int v) { g_k1_variant = v; }

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
    if (g_k1_variant >= 6 && g_k1_variant <= 13) {
        const int opt = g_k1_variant - 6;
        const int th = (opt & 1) ? 1 : 4;
        const bool nt = opt & 2, fastm = opt & 4;
        const int LPP = C / 4, PIX = 256 / LPP;
        const size_t per_view = (size_t)5 * DK * PIX * sizeof(float);
        int VC = (int)((64 * 1024) / per_view);
        if (VC > V - 1) VC = V - 1;
        const bool multi = VC < V - 1;
        const size_t lds = per_view * VC;
        const int TW6 = PIX / th;
        const int tx6 = (w + TW6 - 1) / TW6, ty6 = (h + th - 1) / th;
        dim3 grid6(tx6 * ty6, (D + DK - 1) / DK, B);
#define RCMVS_K1V6(CC, TT, NN, FF, MM) hipLaunchKernelGGL((warp_variance_v6_kernel<CC, TT, NN, FF, MM>), grid6, dim3(256), lds, st, feats, rot, trans, planes, var, V, D, h, w, tx6, VC)
#define RCMVS_K1V6_M(CC, TT, NN, FF) do { if (multi) RCMVS_K1V6(CC, TT, NN, FF, true); else RCMVS_K1V6(CC, TT, NN, FF, false); } while (0)
#define RCMVS_K1V6_F(CC, TT, NN) do { if (fastm) RCMVS_K1V6_M(CC, TT, NN, true); else RCMVS_K1V6_M(CC, TT, NN, false); } while (0)
#define RCMVS_K1V6_N(CC, TT) do { if (nt) RCMVS_K1V6_F(CC, TT, true); else RCMVS_K1V6_F(CC, TT, false); } while (0)
#define RCMVS_K1V6_T(CC) do { if (th == 1) RCMVS_K1V6_N(CC, 1); else RCMVS_K1V6_N(CC, 4); } while (0)
        switch (C) {
            case 8:  RCMVS_K1V6_T(8); break;
            case 16: RCMVS_K1V6_T(16); break;
            case 32: RCMVS_K1V6_T(32); break;
            default: return fail(-1, "warp_variance_fwd: C must be 8, 16 or 32 (got %d)", C);
        }
        return launch_status("warp_variance_fwd(v6)");
    }
#define RCMVS_K1_LAUNCH(CC, VV) hipLaunchKernelGGL((warp_variance_kernel<CC, VV>), grid, dim3(256), 0, st, feats, rot, trans, planes, var, V, D, h, w, tiles_x, tiles_y)
#define RCMVS_K1_VARIANTS(CC)                                                               \
    switch (g_k1_variant) {                                                                 \
        case 0: RCMVS_K1_LAUNCH(CC, 0); break;                                              \
        case 1: RCMVS_K1_LAUNCH(CC, 1); break;                                              \
        case 2: RCMVS_K1_LAUNCH(CC, 2); break;                                              \
        case 3: RCMVS_K1_LAUNCH(CC, 3); break;                                              \
        case 4: RCMVS_K1_LAUNCH(CC, 4); break;                                              \
        case 5: RCMVS_K1_LAUNCH(CC, 5); break;                                              \
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
    RCMVS_REQUIRE(B > 0 && D > 0 && h > 1 && w > 1 && C > 0, "warp_noref_fwd: bad sizes");
    RCMVS_REQUIRE(V >= 2 && V - 1 <= RCMVS_MAX_SRC_VIEWS, "warp_noref_fwd: V=%d unsupported", V);
    dim3 grid((unsigned)cdiv((long long)h * w, 256), D, B);
    hipLaunchKernelGGL(warp_noref_kernel, grid, dim3(256), 0, as_stream(stream), feats, imgs, rot, trans, planes, out,
                       V, C, D, h, w, square_first);
    return launch_status("warp_noref_fwd");
}

}  // extern "C"
