// K1 backward: gradient of the fused warp + variance cost volume with respect to the feature maps.
//
// Replaces the autograd graph PyTorch records for (V-1) x homo_warping (grid_sample backward,
// models/modules.py:333-337) and the sum / square-sum / variance chain of DepthNet.forward
// (models/casmvsnet.py:59-101), including the train variant's source-only variance
// (volume_feature_no_ref, :89-101).  homo_warping builds its sampling grid under torch.no_grad()
// (modules.py:313), so no gradient reaches the hypothesis planes or the cameras: the only
// differentiable inputs are the V feature maps.
//
// With f_0 the reference feature, f_v = bilinear sample of source view v, S = sum_v f_v (v = 0..V-1) and
// N = sum_{v>=1} f_v:
//   var   = sum f_v^2 / V - (S / V)^2        d var   / d f_v = (2 / V) (f_v - S / V)       v = 0..V-1
//   noref = sum_{v>=1} f_v^2 / V - (N / V)^2 d noref / d f_v = (2 / V) (f_v - N / V)       v = 1..V-1
// (the reference divides the source-only sums by V, not V-1).  The gradient of a bilinear sample
// is scattered to its four taps with the forward's masked weights.
//
// Mapping: the forward's two-phase structure (k1_taps.h) so that forward and backward sample at
// bit-identical positions.  One block owns a 4-row pixel tile and walks ALL plane chunks, so the
// reference-view gradient is a register accumulation and a plain store; source-view gradients are
// hardware fp32 atomic adds (global_atomic_add_f32) into a zero-initialised buffer -- unordered,
// like PyTorch's grid_sample backward.
//
// Run-length merging (NM = compile-time source-view count, up to 4): the kernel is bound by atomic throughput (251 M
// atomics for the 512x640x8 stage at 3 source views), and consecutive planes of one reference pixel hit overlapping 2x2
// footprints (same footprint 35-40 %, one-column shift 50-70 %, profiles/r1_k1_tuning_notes.txt).  Each lane therefore
// keeps the footprint of the previous plane pending in registers -- four byte offsets + four gradient quads per view --
// adds into it while the footprint repeats, slides it on a column shift and only flushes the texels that drop out.
//
// Coalesced scatter (round 2).  fp32 atomics are priced per 64-byte line an instruction touches, not per dword
// (tools/dev/atomic_rate.hip: 330 G dwords/s when the 64 lanes add to 64 consecutive floats = 4 lines, 80 G/s in the
// (pixel, channel quad) layout of the arithmetic, where the four component instructions each touch 16 lines; same-address lanes
// inside one instruction serialise: 50 G/s), and the scatter is 95 % of this kernel (arithmetic floor 0.12-0.19 ms of 2.6-3.9 ms per
// launch, tools/dev/k1_bwd_ablate.py).  So the gradient quads of a wave are transposed through a wave-private 1 KB LDS
// scratch before the scatter: the wave's 256 floats are [pixel][channel] in lane-major order, and instruction i takes floats
// 64 i .. 64 i + 63 -- 2 whole texels at C = 32, 4 at C = 16, 8 at C = 8 -- so an atomic instruction touches ~4 lines.  The
// run-length merging runs in the transposed domain (per lane: 4 items x source view, scalar gradients).
#include "common.h"
#include "k1_taps.h"

namespace rcmvs {

constexpr int BWD_DKB = 4;           // planes per chunk
constexpr int BWD_MAXSRC = RCMVS_MAX_SRC_VIEWS;

__device__ __forceinline__ void atomic_add4(float* p, v4f v) {
    unsafeAtomicAdd(p + 0, v.x);
    unsafeAtomicAdd(p + 1, v.y);
    unsafeAtomicAdd(p + 2, v.z);
    unsafeAtomicAdd(p + 3, v.w);
}

__device__ __forceinline__ void flush4(float* gb, int off, v4f g) {
    if (g.x != 0.0f || g.y != 0.0f || g.z != 0.0f || g.w != 0.0f) atomic_add4(gb + (off >> 2), g);
}

__device__ __forceinline__ void flush1(float* gb, int off, float g) {
    if (g != 0.0f) unsafeAtomicAdd(gb + (off >> 2), g);
}

// SCATTER = false (rcmvs_debug_warp_variance_bwd, variant bit 0): everything but the atomics -- the arithmetic floor of the kernel;
// the would-be scatter values are folded into the reference-view gradient so that nothing is optimised away (results are meaningless)
template <int C, int NM, bool SCATTER = true>
__global__ __launch_bounds__(256) void warp_variance_bwd_kernel(
    const float* __restrict__ feats, const float* __restrict__ rot, const float* __restrict__ trans,
    const float* __restrict__ planes, const float* __restrict__ gvar, const float* __restrict__ gnr,
    float* __restrict__ gfeats, int V, int D, int h, int w, int tiles_x) {
    constexpr int LPP = C / 4;
    constexpr int PIX = 256 / LPP;
    constexpr int TH = 4, TW = PIX / TH;
    constexpr int GRP = 256 / PIX;
    constexpr int DKB = BWD_DKB;
    extern __shared__ __attribute__((aligned(16))) v4i lds_o[];          // [nsrc][DKB][PIX]
    const int nsrc = V - 1;
    v4f* lds_w = reinterpret_cast<v4f*>(lds_o + nsrc * DKB * PIX);
    const int b = blockIdx.y;
    const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tx0 = (tile % tiles_x) * TW, ty0 = (tile / tiles_x) * TH;
    const int hw = h * w;
    K1Geom g;
    g.w = w; g.h = h;
    g.wm1 = (float)(w - 1); g.hm1 = (float)(h - 1);
    g.half_w = g.wm1 / 2.0f; g.half_h = g.hm1 / 2.0f;
    g.r_half_w = rcp_nr(g.half_w); g.r_half_h = rcp_nr(g.half_h);
    const float* fb = feats + (long long)b * V * hw * C;
    float* gb = gfeats + (long long)b * V * hw * C;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fb), (short)0, 0x7fffffff, 0x00020000);

    const int p = threadIdx.x / LPP;
    const int q4b = (threadIdx.x % LPP) * 16;
    const int x = tx0 + p % TW, y = ty0 + p / TW;
    const bool inside = (x < w) && (y < h);
    v4f ref = (v4f){0.f, 0.f, 0.f, 0.f};
    if (inside) ref = *reinterpret_cast<const v4f*>(fb + ((long long)y * w + x) * C + (q4b >> 2));
    const float rV = 1.0f / (float)V, c2 = 2.0f / (float)V;
    const long long gpix = (((long long)b * D) * hw + (long long)y * w + x) * C + (q4b >> 2);
    const int pa = threadIdx.x % PIX, ga = threadIdx.x / PIX;
    const int xa = min(tx0 + pa % TW, w - 1), ya = min(ty0 + pa / TW, h - 1);
    const float fxa = (float)xa, fya = (float)ya;
    const float2 pla = reinterpret_cast<const float2*>(planes)[(long long)b * hw + ya * w + xa];

    v4f gref = (v4f){0.f, 0.f, 0.f, 0.f};
    float sink = 0.0f;                                               // SCATTER = false: keeps the would-be scatter values alive
    auto flush = [&](int off, float gq) { if constexpr (SCATTER) flush1(gb, off, gq); else sink += gq * (float)(off & 4); };
    // transposed scatter domain: item i of this lane is float 64 i + lane of the wave's [pixel][channel] array
    const int lane = threadIdx.x & 63, wvb = (threadIdx.x >> 6) * (64 / LPP);
    float* scr = reinterpret_cast<float*>(lds_w + nsrc * DKB * PIX) + (threadIdx.x >> 6) * 256;      // wave-private 1 KB
    int tp_[4], tch[4];                                              // item -> block pixel index, channel byte offset
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int f = i * 64 + lane; tp_[i] = wvb + f / C; tch[i] = (f % C) * 4; }
    // pending footprints (run-length merging), one per (source view, item)
    constexpr int NMS = NM > 0 ? NM : 1;
    v4i pend_o[NMS][4];
    v4f pend_g[NMS][4];                                              // .x .y .z .w = the four taps
#pragma unroll
    for (int va = 0; va < NMS; ++va)
#pragma unroll
        for (int i = 0; i < 4; ++i) { pend_o[va][i] = (v4i){-1, -1, -1, -1}; pend_g[va][i] = (v4f){0.f, 0.f, 0.f, 0.f}; }
    for (int k0 = 0; k0 < D; k0 += DKB) {
        if (k0 > 0) __syncthreads();
        // ---- phase A: tap table of every (pixel, plane, source view) of this chunk
        for (int j = ga; j < nsrc * DKB; j += GRP) {
            const int va = j / DKB, ka = j % DKB;
            const float* r = rot + ((long long)b * nsrc + va) * 9;
            const float* t = trans + ((long long)b * nsrc + va) * 3;
            float rx, ry, rz;
            {
#pragma clang fp contract(off)
                rx = (r[0] * fxa + r[1] * fya) + r[2];
                ry = (r[3] * fxa + r[4] * fya) + r[5];
                rz = (r[6] * fxa + r[7] * fya) + r[8];
            }
            float d;
            {
#pragma clang fp contract(off)
                d = pla.x + (float)(k0 + ka) * pla.y;
            }
            v4i o;
            v4f wt;
            k1_tap<C>(rx, ry, rz, t[0], t[1], t[2], d, g, (va + 1) * hw, o, wt);
            lds_o[j * PIX + pa] = o;
            lds_w[j * PIX + pa] = wt;
        }
        __syncthreads();
        // ---- phase B: recompute the samples, form d var / d f_v, scatter (all lanes take part: a pixel outside the image
        // contributes zeros -- its taps were computed at the clamped position -- because the scatter is wave-cooperative)
#pragma unroll
        for (int k = 0; k < DKB; ++k) {
            if (k0 + k >= D) continue;
            v4f wv[BWD_MAXSRC];
            v4f s = ref, snr = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int va = 0; va < BWD_MAXSRC; ++va) {
                wv[va] = (v4f){0.f, 0.f, 0.f, 0.f};
                if (va >= nsrc) continue;
                const int idx = (va * DKB + k) * PIX + p;
                const v4i o = lds_o[idx];
                const v4f wt = lds_w[idx];
                const v4f ta = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.x + q4b, 0, 0));
                const v4f tb = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.y + q4b, 0, 0));
                const v4f tc = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.z + q4b, 0, 0));
                const v4f td = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.w + q4b, 0, 0));
                wv[va] = ((ta * wt.x + tb * wt.y) + tc * wt.z) + td * wt.w;
                s = s + wv[va];
                snr = snr + wv[va];
            }
            v4f gv = (v4f){0.f, 0.f, 0.f, 0.f}, gn = (v4f){0.f, 0.f, 0.f, 0.f};
            if (inside) {
                const long long go = gpix + (long long)(k0 + k) * hw * C;
                gv = *reinterpret_cast<const v4f*>(gvar + go);
                if (gnr) gn = *reinterpret_cast<const v4f*>(gnr + go);
            }
            const v4f mean = s * rV, mnr = snr * rV;
            gref = gref + gv * (ref - mean) * c2;
#pragma unroll
            for (int va = 0; va < BWD_MAXSRC; ++va) {
                if (va >= nsrc) continue;
                const v4f gw = (gv * (wv[va] - mean) + gn * (wv[va] - mnr)) * c2;
                // transpose: lane-major quads -> item-major scalars
                __builtin_amdgcn_wave_barrier();
                *reinterpret_cast<v4f*>(scr + lane * 4) = gw;
                __builtin_amdgcn_wave_barrier();
                float a[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = scr[i * 64 + lane];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int idx = (va * DKB + k) * PIX + tp_[i];
                    v4i o = lds_o[idx];
                    const v4f wt = lds_w[idx];
                    o += tch[i];                                     // this item's channel
                    const v4f c = wt * a[i];                         // contribution to the four taps
                    if constexpr (NM > 0) {
                        if (va < NM) {
                            v4i& P = pend_o[va < NM ? va : 0][i];
                            v4f& G = pend_g[va < NM ? va : 0][i];
                            if (o.x == P.x && o.y == P.y && o.z == P.z && o.w == P.w) {
                                G += c;
                            } else if (o.x == P.y && o.z == P.w) {       // one column to the right: west slots drop out
                                flush(P.x, G.x); flush(P.z, G.z);
                                G = (v4f){G.y + c.x, c.y, G.w + c.z, c.w}; P = o;
                            } else if (o.y == P.x && o.w == P.z) {       // one column to the left: east slots drop out
                                flush(P.y, G.y); flush(P.w, G.w);
                                G = (v4f){c.x, G.x + c.y, c.z, G.z + c.w}; P = o;
                            } else {
                                flush(P.x, G.x); flush(P.y, G.y); flush(P.z, G.z); flush(P.w, G.w);
                                G = c; P = o;
                            }
                        }
                    } else {
                        flush(o.x, c.x); flush(o.y, c.y); flush(o.z, c.z); flush(o.w, c.w);
                    }
                }
            }
        }
    }
    if constexpr (NM > 0) {
#pragma unroll
        for (int va = 0; va < NM; ++va)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                flush(pend_o[va][i].x, pend_g[va][i].x); flush(pend_o[va][i].y, pend_g[va][i].y);
                flush(pend_o[va][i].z, pend_g[va][i].z); flush(pend_o[va][i].w, pend_g[va][i].w);
            }
    }
    if (inside) {
        if constexpr (!SCATTER) gref.x += sink;
        *reinterpret_cast<v4f*>(gb + ((long long)y * w + x) * C + (q4b >> 2)) = gref;
    }
}

}  // namespace rcmvs

using namespace rcmvs;

static int k1_bwd_launch(const float* feats, const float* rot, const float* trans, const float* planes,
                         const float* grad_var, const float* grad_noref, float* grad_feats,
                         int B, int V, int C, int D, int h, int w, int variant, void* stream) {
    RCMVS_REQUIRE(feats && rot && trans && planes && grad_var && grad_feats, "warp_variance_bwd: null pointer");
    RCMVS_REQUIRE(B > 0 && D > 0 && h > 1 && w > 1, "warp_variance_bwd: bad sizes B=%d D=%d h=%d w=%d", B, D, h, w);
    RCMVS_REQUIRE(V >= 2 && V - 1 <= RCMVS_MAX_SRC_VIEWS, "warp_variance_bwd: V=%d unsupported", V);
    RCMVS_REQUIRE((long long)V * h * w * C * 4 < 0x7fffffffLL, "warp_variance_bwd: feature block too large for 32-bit offsets");
    RCMVS_REQUIRE(C == 8 || C == 16 || C == 32, "warp_variance_bwd: C must be 8, 16 or 32 (got %d)", C);
    const int LPP = C / 4, PIX = 256 / LPP, TW = PIX / 4;
    const int tiles_x = (w + TW - 1) / TW, tiles_y = (h + 3) / 4;
    const size_t lds = (size_t)(V - 1) * BWD_DKB * PIX * 32 + 4 * 1024;          // tap tables + the four waves' transpose scratch
    dim3 grid(tiles_x * tiles_y, B);
    hipStream_t st = as_stream(stream);
#define RCMVS_K1B_S(CC, NN, SC) do { \
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)warp_variance_bwd_kernel<CC, NN, SC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((warp_variance_bwd_kernel<CC, NN, SC>), grid, dim3(256), lds, st, feats, rot, trans, planes, grad_var, grad_noref, grad_feats, V, D, h, w, tiles_x); } while (0)
#define RCMVS_K1B(CC, NN) do { if (variant & 1) RCMVS_K1B_S(CC, NN, false); else RCMVS_K1B_S(CC, NN, true); } while (0)
#define RCMVS_K1B_N(CC) do { switch ((variant & 2) ? 99 : V - 1) { case 1: RCMVS_K1B(CC, 1); break; case 2: RCMVS_K1B(CC, 2); break; case 3: RCMVS_K1B(CC, 3); break; \
                                          case 4: RCMVS_K1B(CC, 4); break; default: RCMVS_K1B(CC, 0); } } while (0)
    switch (C) {
        case 8:  RCMVS_K1B_N(8); break;
        case 16: RCMVS_K1B_N(16); break;
        default: RCMVS_K1B_N(32); break;
    }
#undef RCMVS_K1B_N
#undef RCMVS_K1B
#undef RCMVS_K1B_S
    return launch_status("warp_variance_bwd");
}

extern "C" int rcmvs_warp_variance_bwd(const float* feats, const float* rot, const float* trans, const float* planes,
                                       const float* grad_var, const float* grad_noref, float* grad_feats,
                                       int B, int V, int C, int D, int h, int w, void* stream) {
    return k1_bwd_launch(feats, rot, trans, planes, grad_var, grad_noref, grad_feats, B, V, C, D, h, w, 0, stream);
}

extern "C" int rcmvs_debug_warp_variance_bwd(const float* feats, const float* rot, const float* trans, const float* planes,
                                             const float* grad_var, const float* grad_noref, float* grad_feats,
                                             int B, int V, int C, int D, int h, int w, int variant, void* stream) {
    return k1_bwd_launch(feats, rot, trans, planes, grad_var, grad_noref, grad_feats, B, V, C, D, h, w, variant, stream);
}
