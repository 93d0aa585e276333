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

// ---------------------------------------------------------------------------------------------------------------------------------
// Window form (round 4).  With smooth hypothesis planes the scatter of a 4 x 32 pixel tile lands, for four consecutive planes, in a
// rectangle of ~6 x 36 texels of each source view.  Such a tile accumulates its scatter in LDS -- one window of 8 x 40 texels per
// source view, anchored at the tile's minimum tap corner -- and flushes the window once per chunk: one global atomic per touched
// (texel, channel) instead of one per (pixel, plane, tap, channel), 64 consecutive floats per instruction.
//
// The window cells are DOUBLES: ds_add_f64 issues in 8.4 clocks per wave instruction, ds_add_f32 in 193 (it walks the lanes one by
// one), ds_add_u32 / u64 in 4.4 / 6.4 (tools/dev/lds_atomic_rate.hip) -- an fp32 window was 3x slower than the run-length kernel it
// was meant to replace (profiles/r4_k1_bwd.txt).  Sums inside a window are therefore exact to 2^-53; one rounding to fp32 at the flush.
//
// A block = (tile, group of 8 channels): the gradient of the variance is independent per channel, so C = 16 / 32 run as 2 / 4
// blocks per tile and the windows stay at 8 channels x 324 cells x 8 B = 20.7 KB per view (two blocks per CU at three source views).
// Cell layout [channel][8 x 40 texels + 4]: the channel stride 324 = 4 (mod 32) keeps the two channel quads of 16 consecutive pixels
// in disjoint banks for the adds and the 8 channels x 4 texels of a flush pass in disjoint banks for the reads.
//
// Which tiles: k1_tile_fits() evaluates every tap corner of the tile once more (the same k1_chain, so the same integers) and the tile
// takes the window form only if every (chunk, view) footprint of live taps fits its window -- then no tap can miss and the form
// needs no slow path.  The other tiles (noisy hypotheses, depth
// discontinuities, large scale changes) are left to the run-length kernel above, launched right after with skip_fit = 1: its blocks
// repeat the same test on their enclosing 4 x 32 tile and return when the window form has taken it.
struct K1Win {
    static constexpr int TH = 4, TW = 32, PIX = TH * TW, DKB = BWD_DKB, CG = 8;
    static constexpr int WY = 8, WX = TW + 8, WPIX = WY * WX, CHS = WPIX + 4;
    static constexpr int MAXCHUNK = 48;
    static_assert(CHS % 32 == 4, "window channel stride");
    static __host__ __device__ int chunks(int D) { return (D + DKB - 1) / DKB; }
    static size_t lds_bytes(int nsrc, int D) { return (size_t)nsrc * (DKB * PIX * 8 + CG * CHS * 8) + (size_t)chunks(D) * nsrc * 16 + 16; }
};

constexpr int K1_NONE = 0x7fffffff;

// minimum / maximum over each row of 16 lanes, valid in lanes 15, 31, 47, 63
template <bool MIN>
__device__ __forceinline__ int row_reduce_i32(int v) {
#define RCMVS_DPP_STEP(CTRL) { const int t = __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); v = MIN ? min(v, t) : max(v, t); }
    RCMVS_DPP_STEP(0x111) RCMVS_DPP_STEP(0x112) RCMVS_DPP_STEP(0x114) RCMVS_DPP_STEP(0x118)
#undef RCMVS_DPP_STEP
    return v;
}

// Extents of the live tap corners of the 4 x 32 tile at (tx0, ty0), per (chunk, source view): amin / amax [chunk][view][x, y] in
// LDS (amin = K1_NONE where the tile has no live tap).  Returns true when every footprint fits the window.  All 256 threads of the
// block call it; the answer depends on the tile alone (order-independent integer min / max), so every block that asks about a
// tile -- the channel groups of the window kernel, the sub-tiles of the run-length kernel -- gets the same one.
__device__ __forceinline__ bool k1_tile_fits(int* amin, int* amax, int* verdict, const float* __restrict__ rot, const float* __restrict__ trans,
                                             const float* __restrict__ planes, int b, int nsrc, int D, const K1Geom& g, int tx0, int ty0) {
    const int tid = threadIdx.x;
    const int nch = K1Win::chunks(D);
    for (int i = tid; i < nch * nsrc * 2; i += 256) { amin[i] = K1_NONE; amax[i] = -K1_NONE; }
    if (tid == 0) *verdict = 1;
    __syncthreads();
    const int pa = tid % K1Win::PIX, ga = tid / K1Win::PIX;
    const int xs = tx0 + pa % K1Win::TW, ys = ty0 + pa / K1Win::TW;
    const bool in_img = (xs < g.w) && (ys < g.h);
    const int xa = min(xs, g.w - 1), ya = min(ys, g.h - 1);
    const float fxa = (float)xa, fya = (float)ya;
    const float2 pla = reinterpret_cast<const float2*>(planes)[(long long)b * g.h * g.w + ya * g.w + xa];
    for (int e = ga; e < D * nsrc; e += 256 / K1Win::PIX) {
        const int va = e % nsrc, kk = e / nsrc, chunk = kk / K1Win::DKB;
        const float* r = rot + ((long long)b * nsrc + va) * 9;
        const float* t = trans + ((long long)b * nsrc + va) * 3;
        float rx, ry, rz, d;
        {
#pragma clang fp contract(off)
            rx = (r[0] * fxa + r[1] * fya) + r[2];
            ry = (r[3] * fxa + r[4] * fya) + r[5];
            rz = (r[6] * fxa + r[7] * fya) + r[8];
            d = pla.x + (float)kk * pla.y;
        }
        int xi, yi;
        v4f wt;
        k1_chain(rx, ry, rz, t[0], t[1], t[2], d, g, xi, yi, wt);
        const bool live = in_img && (wt.x != 0.0f || wt.y != 0.0f || wt.z != 0.0f || wt.w != 0.0f);
        const int nx = row_reduce_i32<true>(live ? xi : K1_NONE), ny = row_reduce_i32<true>(live ? yi : K1_NONE);
        const int mx = row_reduce_i32<false>(live ? xi : -K1_NONE), my = row_reduce_i32<false>(live ? yi : -K1_NONE);
        if ((tid & 15) == 15 && nx != K1_NONE) {
            const int o = (chunk * nsrc + va) * 2;
            atomicMin(&amin[o], nx); atomicMin(&amin[o + 1], ny);
            atomicMax(&amax[o], mx); atomicMax(&amax[o + 1], my);
        }
    }
    __syncthreads();
    for (int i = tid; i < nch * nsrc; i += 256)
        if (amin[2 * i] != K1_NONE && (amax[2 * i] - amin[2 * i] > K1Win::WX - 2 || amax[2 * i + 1] - amin[2 * i + 1] > K1Win::WY - 2)) *verdict = 0;
    __syncthreads();
    const bool ok = *verdict != 0;
    __syncthreads();                                                  // the caller may reuse the scratch at once
    return ok;
}


// SCATTER = false (rcmvs_debug_warp_variance_bwd, variant bit 0): everything but the atomics -- the arithmetic floor of the kernel;
// the would-be scatter values are folded into the reference-view gradient so that nothing is optimised away (results are meaningless)
template <int C, int NM, bool SCATTER = true>
__global__ __launch_bounds__(256) void warp_variance_bwd_kernel(
    const float* __restrict__ feats, const float* __restrict__ rot, const float* __restrict__ trans,
    const float* __restrict__ planes, const float* __restrict__ gvar, const float* __restrict__ gnr,
    float* __restrict__ gfeats, int V, int D, int h, int w, int tiles_x, int skip_fit) {
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
    if (skip_fit) {                                                   // the window form (below) has taken the tiles that fit its windows
        int* scratch = reinterpret_cast<int*>(lds_o);
        const int n = K1Win::chunks(D) * nsrc * 2;
        if (k1_tile_fits(scratch, scratch + n, scratch + 2 * n, rot, trans, planes, b, nsrc, D, g, tx0 / K1Win::TW * K1Win::TW, ty0)) return;
    }
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



__global__ __launch_bounds__(256) void warp_variance_bwd_win_kernel(
    const float* __restrict__ feats, const float* __restrict__ rot, const float* __restrict__ trans,
    const float* __restrict__ planes, const float* __restrict__ gvar, const float* __restrict__ gnr,
    float* __restrict__ gfeats, int V, int C, int D, int h, int w, int tiles_x, int dbg) {
    using K = K1Win;
    constexpr int PIX = K::PIX, TH = K::TH, TW = K::TW, DKB = K::DKB, WX = K::WX, WY = K::WY, CHS = K::CHS, WCELLS = K::CG * K::CHS;
    extern __shared__ __attribute__((aligned(16))) double win[];         // [nsrc][8][CHS]
    const int nsrc = V - 1;
    float2* tab = reinterpret_cast<float2*>(win + nsrc * WCELLS);         // [nsrc][DKB][PIX] sampling positions of the chunk
    int* amin = reinterpret_cast<int*>(tab + nsrc * DKB * PIX);           // [chunk][nsrc][2] window anchors
    int* amax = amin + K::chunks(D) * nsrc * 2;
    int* verdict = amax + K::chunks(D) * nsrc * 2;
    const int b = blockIdx.y, cg = blockIdx.z;
    const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tx0 = (tile % tiles_x) * TW, ty0 = (tile / tiles_x) * TH;
    const int hw = h * w;
    K1Geom g;
    g.w = w; g.h = h;
    g.wm1 = (float)(w - 1); g.hm1 = (float)(h - 1);
    g.half_w = g.wm1 / 2.0f; g.half_h = g.hm1 / 2.0f;
    g.r_half_w = rcp_nr(g.half_w); g.r_half_h = rcp_nr(g.half_h);
    if (!(dbg & 8)) { if (!k1_tile_fits(amin, amax, verdict, rot, trans, planes, b, nsrc, D, g, tx0, ty0)) return; }      // the run-length kernel takes this tile
    else { for (int i = threadIdx.x; i < K::chunks(D) * nsrc * 2; i += 256) amin[i] = (i & 1) ? ty0 - 2 : tx0 - 4; __syncthreads(); }

    const float* fb = feats + (long long)b * V * hw * C;
    float* gb = gfeats + (long long)b * V * hw * C;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fb), (short)0, 0x7fffffff, 0x00020000);
    const int tid = threadIdx.x;
    const int p = tid >> 1, q = tid & 1;
    const int ch0 = cg * 8 + q * 4;                                       // this lane's channel quad
    const int q4b = ch0 * 4, texel = C * 4;
    const int x = tx0 + p % TW, y = ty0 + p / TW;
    const bool inside = (x < w) && (y < h);
    v4f ref = (v4f){0.f, 0.f, 0.f, 0.f};
    if (inside) ref = *reinterpret_cast<const v4f*>(fb + ((long long)y * w + x) * C + ch0);
    const float rV = 1.0f / (float)V, c2 = 2.0f / (float)V;
    const long long gpix = (((long long)b * D) * hw + (long long)min(y, h - 1) * w + min(x, w - 1)) * C + ch0;
    const int pa = tid % PIX, ga = tid / PIX;
    const int xa = min(tx0 + pa % TW, w - 1), ya = min(ty0 + pa / TW, h - 1);
    const float fxa = (float)xa, fya = (float)ya;
    const float2 pla = reinterpret_cast<const float2*>(planes)[(long long)b * hw + ya * w + xa];

    for (int i = tid; i < nsrc * WCELLS; i += 256) win[i] = 0.0;
    v4f gref = (v4f){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0, chunk = 0; k0 < D; k0 += DKB, ++chunk) {
        // ---- phase A: sampling position of every (pixel, plane, source view) of this chunk
        for (int j = ga; j < nsrc * DKB; j += 256 / PIX) {
            const int va = j / DKB, ka = j % DKB;
            const float* r = rot + ((long long)b * nsrc + va) * 9;
            const float* t = trans + ((long long)b * nsrc + va) * 3;
            float rx, ry, rz, d;
            {
#pragma clang fp contract(off)
                rx = (r[0] * fxa + r[1] * fya) + r[2];
                ry = (r[3] * fxa + r[4] * fya) + r[5];
                rz = (r[6] * fxa + r[7] * fya) + r[8];
                d = pla.x + (float)(k0 + ka) * pla.y;
            }
            float ix, iy;
            k1_position(rx, ry, rz, t[0], t[1], t[2], d, g, ix, iy);
            tab[j * PIX + pa] = make_float2(ix, iy);
        }
        __syncthreads();
        // ---- phase B: recompute the samples, form d var / d f_v, add into the windows (no divergent branch: a pixel outside the image
        // or a plane beyond D carries a zero gradient, a tap without a live weight adds zeros at the window's first cell)
#pragma unroll
        for (int k = 0; k < DKB; ++k) {
            const bool on = inside && (k0 + k < D);
            const long long go = gpix + (long long)min(k0 + k, D - 1) * hw * C;
            v4f gv = *reinterpret_cast<const v4f*>(gvar + go);
            v4f gn = (v4f){0.f, 0.f, 0.f, 0.f};
            if (gnr) gn = *reinterpret_cast<const v4f*>(gnr + go);
            if (!on) { gv = (v4f){0.f, 0.f, 0.f, 0.f}; gn = gv; }
            v4f wv[BWD_MAXSRC], wts[BWD_MAXSRC];
            int cell[BWD_MAXSRC];
            v4f s = ref, snr = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int va = 0; va < BWD_MAXSRC; ++va) {
                wv[va] = (v4f){0.f, 0.f, 0.f, 0.f};
                if (va >= nsrc) continue;
                const float2 pos = tab[(va * DKB + k) * PIX + p];
                int xi, yi;
                k1_corner(pos.x, pos.y, g, xi, yi, wts[va]);
                v4i o;
                k1_offsets_rt(xi, yi, g, (va + 1) * hw, texel, o);
                if (dbg & 2) o = (v4i){0, 0, 0, 0};
                const v4f ta = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.x + q4b, 0, 0));
                const v4f tb = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.y + q4b, 0, 0));
                const v4f tc = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.z + q4b, 0, 0));
                const v4f td = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.w + q4b, 0, 0));
                wv[va] = ((ta * wts[va].x + tb * wts[va].y) + tc * wts[va].z) + td * wts[va].w;
                s = s + wv[va];
                snr = snr + wv[va];
                const int wx = xi - amin[(chunk * nsrc + va) * 2], wy = yi - amin[(chunk * nsrc + va) * 2 + 1];
                const bool hit = (unsigned)wx < (unsigned)(WX - 1) && (unsigned)wy < (unsigned)(WY - 1);
                cell[va] = hit ? wy * WX + wx : 0;
                if (!hit) wts[va] = (v4f){0.f, 0.f, 0.f, 0.f};           // cannot happen for a live tap of a tile that passed k1_tile_fits
            }
            const v4f mean = s * rV, mnr = snr * rV;
            gref = gref + gv * (ref - mean) * c2;
#pragma unroll
            for (int va = 0; va < BWD_MAXSRC; ++va) {
                if (va >= nsrc) continue;
                const v4f gw = (gv * (wv[va] - mean) + gn * (wv[va] - mnr)) * c2;
                double* c0 = win + va * WCELLS + (q * 4) * CHS + cell[va];
                if (dbg & 1) { gref = gref + gw * (float)cell[va]; continue; }
                // the sixteen values first, each in its own register pair, then the sixteen adds: an add that reads the pair the next
                // conversion writes holds that conversion back until the LDS has fetched its operand
                double cv[16];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    cv[j * 4 + 0] = (double)(gw[j] * wts[va].x); cv[j * 4 + 1] = (double)(gw[j] * wts[va].y);
                    cv[j * 4 + 2] = (double)(gw[j] * wts[va].z); cv[j * 4 + 3] = (double)(gw[j] * wts[va].w);
                }
                if (dbg & 32) { double t = 0.0; for (int i = 0; i < 16; ++i) t += cv[i]; gref.x += (float)t * (float)cell[va]; continue; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsafeAtomicAdd(c0 + j * CHS, cv[j * 4 + 0]);
                    unsafeAtomicAdd(c0 + j * CHS + 1, cv[j * 4 + 1]);
                    unsafeAtomicAdd(c0 + j * CHS + WX, cv[j * 4 + 2]);
                    unsafeAtomicAdd(c0 + j * CHS + WX + 1, cv[j * 4 + 3]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        // ---- flush: every touched (texel, channel) of the windows, 8 texels x 8 channels per instruction; leaves the windows zero
        for (int va = 0; va < nsrc; ++va) {
            const int ax = amin[(chunk * nsrc + va) * 2], ay = amin[(chunk * nsrc + va) * 2 + 1];
            if (ax == K1_NONE || (dbg & 16)) continue;
            double* wv_ = win + va * WCELLS;
            float* gv_ = gb + (long long)(va + 1) * hw * C + cg * 8;
            for (int f = tid; f < K::WPIX * 8; f += 256) {
                const int px = f >> 3, ch = f & 7;
                const double val = wv_[ch * CHS + px];
                if (val != 0.0) {
                    wv_[ch * CHS + px] = 0.0;
                    const int wy = px / WX, wx = px - wy * WX;
                    const int gx = ax + wx, gy = ay + wy;
                    if ((unsigned)gx < (unsigned)w && (unsigned)gy < (unsigned)h && !(dbg & 4)) unsafeAtomicAdd(gv_ + ((long long)gy * w + gx) * C + ch, (float)val);
                }
            }
        }
        // (the next chunk's phase A only writes the table, which phase B has finished reading; its phase B starts after a barrier)
    }
    if (inside) *reinterpret_cast<v4f*>(gb + ((long long)y * w + x) * C + ch0) = gref;
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
        hipLaunchKernelGGL((warp_variance_bwd_kernel<CC, NN, SC>), grid, dim3(256), lds, st, feats, rot, trans, planes, grad_var, grad_noref, grad_feats, V, D, h, w, tiles_x, skip_fit); } while (0)
#define RCMVS_K1B(CC, NN) do { if (variant & 1) RCMVS_K1B_S(CC, NN, false); else RCMVS_K1B_S(CC, NN, true); } while (0)
#define RCMVS_K1B_N(CC) do { switch ((variant & 2) ? 99 : V - 1) { case 1: RCMVS_K1B(CC, 1); break; case 2: RCMVS_K1B(CC, 2); break; case 3: RCMVS_K1B(CC, 3); break; \
                                          case 4: RCMVS_K1B(CC, 4); break; default: RCMVS_K1B(CC, 0); } } while (0)
    // window form first (unless the run-length form alone is asked for: variant bit 2; or timed without its scatter: bit 0; or the
    // windows / the chunk table do not fit), then the run-length form over the tiles the window form declined
    int skip_fit = 0;
    {
        const size_t wl = K1Win::lds_bytes(V - 1, D);
        if (!(variant & 5 & 0xff) && wl <= 160 * 1024 && K1Win::chunks(D) <= K1Win::MAXCHUNK) {
            const int wtx = (w + K1Win::TW - 1) / K1Win::TW;
            if (wl > 64 * 1024) (void)hipFuncSetAttribute((const void*)warp_variance_bwd_win_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wl);
            if (!(variant & 16))
                hipLaunchKernelGGL(warp_variance_bwd_win_kernel, dim3(wtx * ((h + K1Win::TH - 1) / K1Win::TH), B, C / K1Win::CG), dim3(256), wl, st,
                                   feats, rot, trans, planes, grad_var, grad_noref, grad_feats, V, C, D, h, w, wtx, variant >> 8);
            skip_fit = 1;
            if (variant & 8) return launch_status("warp_variance_bwd");            // test twins: bit 3 = the window kernel alone, bit 4 = the run-length kernel's share alone
        }
    }
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
