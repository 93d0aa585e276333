// Weight gradient of the 3x3x3 convolution family on v_mfma_f32_16x16x4_f32 (exact fp32), channels-last.
//
//   dW[tap][ci][co] = sum over output cells o of  x[S*o + tap - 1][ci] * dy[o][co]        (S = stride 1 | 2)
// which is what autograd computes for nn.Conv3d (models/modules.py:145) and, with the roles of the two
// tensors swapped, for nn.ConvTranspose3d (:189; see rcmvs.h).  GEMM view: K = output cells (4 per MFMA),
// N = output channels, M = (tap, input channel) -- the taps are multiplexed onto the M side so that a
// 16-row tile is full for every CI:
//   A[m][k]: lane l (m = l & 15, k = l >> 4) loads ONE float4 = input channels 4*cig .. 4*cig+3 of tap
//            `tapsel` at cell w0 + k, with (tapsel, cig) = (m / (CI/4), m % (CI/4)); component ja feeds MFMA ja,
//            so the four MFMAs of a step cover input channels 4*cig + ja (a row permutation of M, undone at
//            the flush).  One tile = 64/CI taps x all CI channels.
//   B[k][n]: lane l (n = l & 15) loads NJ = CO/16 consecutive floats dy[cell][NJ*n + jb]; component jb feeds
//            the MFMAs of column block jb (CO = 8 and 1 use the first CO lanes of one block).
// A wave keeps up to 128 accumulator registers = G tap groups; the remaining groups go to grid.y.  A wave
// walks whole output rows (b, od, oh) in steps of four cells, reading both operands straight through the
// vector L1 (the 27 taps re-read the same lines, so L1/L2 absorb the 27x reuse), and flushes with hardware
// fp32 atomics into the zero-filled [27][CI][CO] gradient: consecutive lanes (output channels) hit consecutive addresses,
// which the atomic units coalesce -- writing nn.Conv3d's (CO, CI, 27) layout directly made the flush 2-3x slower
// (summation order is not deterministic).
#include "common.h"
#include <cstdlib>

namespace rcmvs {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WgradDims {
    int B, D, H, W;       // x
    int Do, Ho, Wo;       // dy
};

template <int CI, int CO, int STRIDE>
struct WgradCfg {
    static constexpr int CQ = CI / 4;
    static constexpr int TPM = 16 / CQ;
    // CO = 8 fills only half of the 16 MFMA columns.  The idle half takes dy shifted by ONE cell along w: with A = x at tap
    // (kd, kh, kw), column 8 + co then accumulates sum x[cell + kw - 1] dy[cell + 1][co] = the gradient of tap (kd, kh, kw - 1).
    // Rows therefore only enumerate kw = 1, 2 (18 "row taps" instead of 27): kw = 1 rows yield kw = 1 (left half) and kw = 0
    // (right half), kw = 2 rows yield kw = 2 (their right half repeats kw = 1 and is dropped): 3/4 of the tile is useful.
    static constexpr bool PAIR = (CO == 8 && STRIDE == 1);
    static constexpr int NT = PAIR ? 18 : 27;                       // row taps
    static constexpr int NGRP = (NT + TPM - 1) / TPM;
    static constexpr int NJ = CO >= 16 ? CO / 16 : 1;
    static constexpr int GMAX = 4;                                   // accumulator groups per wave (x 16 NJ registers each)
    static constexpr int G = (GMAX / NJ) < 1 ? 1 : ((GMAX / NJ) < NGRP ? (GMAX / NJ) : NGRP);
    static constexpr int SPLITS = (NGRP + G - 1) / G;
};

// PLANE = the volume has a single plane (2-D layers run as D = 1): dead tap planes are skipped (kept out of the 3-D
// instantiation, whose schedule the extra branches would disturb: 64->64 1.4 -> 3.8 ms when they were unconditional).
constexpr int WG_WAVES = 8;       // row-walking waves per block: they are summed through LDS before the flush (fewer, larger flushes)
template <int CI, int CO, int STRIDE, bool PLANE>
__global__ __launch_bounds__(WG_WAVES * 64) void conv3d_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ dw, WgradDims dm, int rows,
                                                           int ldx, int ci_off, int ci_total, int wchunk) {
    using Cfg = WgradCfg<CI, CO, STRIDE>;
    constexpr int CQ = Cfg::CQ, TPM = Cfg::TPM, NGRP = Cfg::NGRP, NJ = Cfg::NJ, G = Cfg::G;
    static_assert(CI % 4 == 0 && CQ <= 16 && 16 % CQ == 0, "CI must be 4, 8, 16, 32 or 64");
    // x holds ldx channels per cell; this launch handles channels ci_off .. ci_off + CI - 1 of the ci_total the gradient has
    // (channel counts outside the power-of-two set, e.g. the renderer's 44, are covered by several launches)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int tapsel = m / CQ, cig = m % CQ;
    const int g0 = blockIdx.y * G;

    int tdx[G];                              // packed (valid << 6) | (td << 4) | (th << 2) | tw of this lane's tap per group
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
        const int rt = (g0 + gi) * TPM + tapsel;                  // row tap of this lane
        const bool valid = (g0 + gi) < NGRP && rt < Cfg::NT;
        const int td = Cfg::PAIR ? rt / 6 : rt / 9, th = Cfg::PAIR ? (rt / 2) % 3 : (rt / 3) % 3, tw = Cfg::PAIR ? 1 + (rt & 1) : rt % 3;
        tdx[gi] = valid ? (64 | (td << 4) | (th << 2) | tw) : 0;
    }
    f32x4 acc[G][4][NJ];
#pragma unroll
    for (int gi = 0; gi < G; ++gi)
#pragma unroll
        for (int ja = 0; ja < 4; ++ja)
#pragma unroll
            for (int jb = 0; jb < NJ; ++jb) acc[gi][ja][jb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // one-plane volumes (2-D layers run as D = 1): the tap planes td = 0 and td = 2 never see data; groups made only of
    // such taps are skipped wholesale (wave-uniform), which removes up to two thirds of the MFMAs
    unsigned gdead = 0;
    if (PLANE && STRIDE == 1) {
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            const int t0 = (g0 + gi) * TPM, t1 = t0 + TPM - 1;
            if (Cfg::PAIR ? (t1 < 6 || t0 >= 12) : (t1 < 9 || t0 >= 18)) gdead |= 1u << gi;
        }
    }
    const int w_lo = PLANE ? blockIdx.z * wchunk : 0, w_hi = PLANE ? min(dm.Wo, w_lo + wchunk) : dm.Wo;   // few rows: the row is split across grid.z
    const bool n_ok = NJ * m < CO;           // (m doubles as the B-side column index n = lane & 15)
    // x is read through a buffer descriptor with 32-bit byte offsets: out-of-range taps get an offset beyond num_records and
    // the hardware returns 0 -- no branches, no 64-bit address arithmetic in the inner loop (the first version spent ~10x
    // more VALU cycles on addresses than the MFMAs took)
    const long long xbytes = (long long)dm.B * dm.D * dm.H * dm.W * ldx * 4;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), (short)0, (int)xbytes, 0x00020000);
    constexpr int OOB = 0x7ffffff0;
    const int lane_off = (ci_off + cig * 4) * 4;
    const int cell_bytes = ldx * 4;
    for (int row = blockIdx.x * WG_WAVES + wave; row < rows; row += gridDim.x * WG_WAVES) {
        const int oh = row % dm.Ho;
        const int od = (row / dm.Ho) % dm.Do;
        const int b = row / (dm.Ho * dm.Do);
        const float* dyrow = dy + ((((long long)b * dm.Do + od) * dm.Ho + oh) * dm.Wo) * CO + (Cfg::PAIR ? (m & 7) : NJ * m);
        // per group: byte offset of (b, id, ih, iw = 0) for this lane's tap, or "row invalid"
        int rbase[G];
        unsigned rok = 0;                                       // bit gi: the tap's (d, h) row exists
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            const int t = tdx[gi];
            const int id = od * STRIDE + ((t >> 4) & 3) - 1;
            const int ih = oh * STRIDE + ((t >> 2) & 3) - 1;
            const bool ok = (t & 64) && (unsigned)id < (unsigned)dm.D && (unsigned)ih < (unsigned)dm.H;
            rbase[gi] = (((b * dm.D + id) * dm.H + ih) * dm.W + ((t & 3) - 1)) * cell_bytes + lane_off;
            rok |= ok ? (1u << gi) : 0u;
        }
        for (int w0 = w_lo; w0 < w_hi; w0 += 4) {
            const int ow = w0 + kq;
            const bool w_ok = ow < w_hi;                        // ragged last step: the cell contributes zero
            float bv[NJ];
            if constexpr (NJ == 4) {
                f32x4 t = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (w_ok) t = *reinterpret_cast<const f32x4*>(dyrow + (long long)ow * CO);
                bv[0] = t.x; bv[1] = t.y; bv[2] = t.z; bv[3] = t.w;
            } else if constexpr (NJ == 2) {
                float2 t = make_float2(0.f, 0.f);
                if (w_ok) t = *reinterpret_cast<const float2*>(dyrow + (long long)ow * CO);
                bv[0] = t.x; bv[1] = t.y;
            } else {
                if constexpr (Cfg::PAIR) {                          // columns 8..15: dy of the next cell along w (tap kw - 1)
                    const int sh = m >> 3;
                    bv[0] = (w_ok && ow + sh < dm.Wo) ? dyrow[(long long)(ow + sh) * CO] : 0.0f;
                } else {
                    bv[0] = (n_ok && w_ok) ? dyrow[(long long)ow * CO] : 0.0f;
                }
            }
            const int iw0 = ow * STRIDE;                         // input column of tap tw = 1 (the -1 is folded into rbase)
            f32x4 av[G];
#pragma unroll
            for (int gi = 0; gi < G; ++gi) {
                if (PLANE && ((gdead >> gi) & 1u)) { av[gi] = (f32x4){0.f, 0.f, 0.f, 0.f}; continue; }
                const int iw = iw0 + (tdx[gi] & 3) - 1;
                const bool ok = w_ok && ((rok >> gi) & 1u) && (unsigned)iw < (unsigned)dm.W;
                av[gi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? rbase[gi] + iw0 * cell_bytes : OOB, 0, 0));
            }
#pragma unroll
            for (int gi = 0; gi < G; ++gi) {
                if (PLANE && ((gdead >> gi) & 1u)) continue;
#pragma unroll
                for (int jb = 0; jb < NJ; ++jb) {
                    acc[gi][0][jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[gi].x, bv[jb], acc[gi][0][jb], 0, 0, 0);
                    acc[gi][1][jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[gi].y, bv[jb], acc[gi][1][jb], 0, 0, 0);
                    acc[gi][2][jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[gi].z, bv[jb], acc[gi][2][jb], 0, 0, 0);
                    acc[gi][3][jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[gi].w, bv[jb], acc[gi][3][jb], 0, 0, 0);
                }
            }
        }
    }
    // The block's eight waves hold the same accumulator set (they walked different rows): sum them through LDS so that ONE wave
    // flushes.  The flush is what the kernel waits for -- every flushing wave adds its whole set to the same [27][CI][CO] words
    // (round 3, profiles/r3_wgrad_flush.txt: 32 -> 8 at 48x128x160 with 1024 blocks x 4 flushing waves 718 us, 256 blocks 370 us, for the
    // same arithmetic) -- so the number of flushes, not of rows, sets the duration.  Four waves per block summed: 385 us; eight: 295 us; sixteen: 320 us.
    {
        constexpr int NACC = G * 4 * NJ;
        __shared__ f32x4 red[NACC * 64];
#pragma unroll 1
        for (int src = 1; src < WG_WAVES; ++src) {
            if (wave == src) {
#pragma unroll
                for (int gi = 0; gi < G; ++gi)
#pragma unroll
                    for (int ja = 0; ja < 4; ++ja)
#pragma unroll
                        for (int jb = 0; jb < NJ; ++jb) red[((gi * 4 + ja) * NJ + jb) * 64 + lane] = acc[gi][ja][jb];
            }
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int gi = 0; gi < G; ++gi)
#pragma unroll
                    for (int ja = 0; ja < 4; ++ja)
#pragma unroll
                        for (int jb = 0; jb < NJ; ++jb) acc[gi][ja][jb] += red[((gi * 4 + ja) * NJ + jb) * 64 + lane];
            }
            __syncthreads();
        }
        if (wave != 0) return;
    }
    // flush: D[mrow][n], lane holds rows 4*kq + r of column n = lane & 15
    const int co0 = NJ * m;
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
        if (g0 + gi >= NGRP || (PLANE && ((gdead >> gi) & 1u))) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int mrow = 4 * kq + r;
            const int rt = (g0 + gi) * TPM + mrow / CQ;
            if (rt >= Cfg::NT) continue;
            const int cq = mrow % CQ;
            int tap = rt;                                            // weight-gradient tap of this lane's column
            if constexpr (Cfg::PAIR) {
                const int kw = 1 + (rt & 1), half = m >> 3;
                tap = (rt / 6) * 9 + ((rt / 2) % 3) * 3 + (kw - half);
                if (half && kw == 2) continue;                       // right half of a kw = 2 row repeats kw = 1
            }
#pragma unroll
            for (int ja = 0; ja < 4; ++ja)
#pragma unroll
                for (int jb = 0; jb < NJ; ++jb) {
                    const int co = Cfg::PAIR ? (m & 7) : co0 + jb;
                    if (co < CO) unsafeAtomicAdd(dw + ((long long)tap * ci_total + ci_off + cq * 4 + ja) * CO + co, acc[gi][ja][jb][r]);
                }
        }
    }
}

// Weight gradient of the prob conv (8 -> 1).  With one output channel the general kernel would fill 1/16 of a tile, so
// the GEMM is turned around: M = input channel (8 of 16 rows), N = TAP (27 of 32 columns, two tiles), K = cells:
//   D[ci][tap] = sum_q x[q][ci] * dy[q - (tap - 1)]
// A is the unshifted activation (one float per lane), B the 1-channel gradient at the tap's offset.
__global__ __launch_bounds__(256) void conv3d_wgrad_c1_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              float* __restrict__ dw, WgradDims dm, int rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, kq = lane >> 4;
    int tdx[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int tap = t * 16 + m;
        tdx[t] = tap < 27 ? (64 | ((tap / 9) << 4) | (((tap / 3) % 3) << 2) | (tap % 3)) : 0;
    }
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const int ih = row % dm.H;
        const int id = (row / dm.H) % dm.D;
        const int b = row / (dm.H * dm.D);
        const float* xrow = x + ((((long long)b * dm.D + id) * dm.H + ih) * dm.W) * 8 + m;
        const float* dyb = dy + (long long)b * dm.D * dm.H * dm.W;
        for (int w0 = 0; w0 < dm.W; w0 += 4) {
            const int iw = w0 + kq;
            const bool w_ok = iw < dm.W;
            const float av = (w_ok && m < 8) ? xrow[(long long)iw * 8] : 0.0f;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int tt = tdx[t];
                const int od = id - ((tt >> 4) & 3) + 1, oh = ih - ((tt >> 2) & 3) + 1, ow = iw - (tt & 3) + 1;
                const bool ok = (tt & 64) && w_ok && (unsigned)od < (unsigned)dm.D && (unsigned)oh < (unsigned)dm.H && (unsigned)ow < (unsigned)dm.W;
                const float bv = ok ? dyb[((long long)od * dm.H + oh) * dm.W + ow] : 0.0f;
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[t], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int tap = t * 16 + m;                  // column n = lane & 15
        if (tap >= 27) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = 4 * kq + r;               // row
            if (ci < 8) unsafeAtomicAdd(dw + tap * 8 + ci, acc[t][r]);
        }
    }
}

// Weight gradient of the prob conv (8 -> 1), plane-marching form (round 2; replaces conv3d_wgrad_c1_kernel in production):
//   dW[kd][kh][kw][ci] = sum over output voxels q of  dy[q] * x[q + (kd - 1, kh - 1, kw - 1)][ci]
// The MFMA form above feeds every 16x16x4 MFMA from per-lane scalar gathers (1.3-2.3 TF: 0.5-0.9 ms per launch).  Here the
// structure of the forward kernel (conv3d_lds.hip, prob_conv_march_kernel) is reused: a block owns an 8 x 16 pixel tile, two
// threads per pixel (four input channels each), and marches over z with the x planes double-buffered in LDS; plane z of x meets
// dy[z + 1], dy[z], dy[z - 1] of the thread's pixel (kd = 0, 1, 2), i.e. 9 ds_read_b128 and 27 float4 FMAs per plane into 108
// register accumulators.  Blocks are persistent over (tile, z chunk) items and reduce once at the end: butterfly over the lanes
// of equal channel half, the four waves through LDS, 216 atomics per block.
constexpr int PW_TH = 8, PW_TW = 16, PW_HH = PW_TH + 2, PW_HW = PW_TW + 2, PW_STRIDE = 12, PW_ZC = 16;
constexpr int PW_PLANE = PW_HH * PW_HW * PW_STRIDE;          // floats per staged plane (8,640 B)
constexpr int PW_NLD = (PW_HH * PW_HW * 2 + 255) / 256;      // float4 per thread per plane

__global__ __launch_bounds__(256) void prob_wgrad_march_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               float* __restrict__ dw, int B, int D, int H, int W,
                                                               int tiles_w, int tiles_h, int nzc, int nitems) {
    __shared__ __attribute__((aligned(16))) float plane[2][PW_PLANE];
    __shared__ float red[4][2][108];
    const int cq = threadIdx.x & 1, pix = threadIdx.x >> 1;
    const int lw = pix % PW_TW, lh = pix / PW_TW;
    f32x4 acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = blockIdx.x; it < nitems; it += gridDim.x) {
        const int zc = it % nzc;
        const int r = it / nzc;
        const int tile = r % (tiles_w * tiles_h), b = r / (tiles_w * tiles_h);
        const int h0 = (tile / tiles_w) * PW_TH, w0 = (tile % tiles_w) * PW_TW, z0 = zc * PW_ZC;
        const int z1 = min(D, z0 + PW_ZC);                      // dy planes z0 .. z1-1 belong to this item; x planes z0-1 .. z1
        const float* xb = x + (long long)b * D * H * W * 8;
        int goff[PW_NLD], loff[PW_NLD];
#pragma unroll
        for (int i = 0; i < PW_NLD; ++i) {
            const int e = threadIdx.x + i * 256;
            const int v = e >> 1, c4 = e & 1;
            const int hh = v / PW_HW, hw_ = v - hh * PW_HW;
            const int ih = h0 + hh - 1, iw = w0 + hw_ - 1;
            const bool ok = e < PW_HH * PW_HW * 2 && ih >= 0 && ih < H && iw >= 0 && iw < W;
            goff[i] = ok ? (ih * W + iw) * 8 + c4 * 4 : -1;
            loff[i] = (e < PW_HH * PW_HW * 2) ? v * PW_STRIDE + c4 * 4 : -1;
        }
        float4 pf[PW_NLD];
        auto fetch = [&](int z) {
            const bool zin = z >= 0 && z < D;
            const float* xp = xb + (long long)z * H * W * 8;
#pragma unroll
            for (int i = 0; i < PW_NLD; ++i)
                pf[i] = (zin && goff[i] >= 0) ? *reinterpret_cast<const float4*>(xp + goff[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        auto stash = [&](int buf) {
#pragma unroll
            for (int i = 0; i < PW_NLD; ++i)
                if (loff[i] >= 0) *reinterpret_cast<float4*>(&plane[buf][loff[i]]) = pf[i];
        };
        const int oh = h0 + lh, ow = w0 + lw;
        const bool live = oh < H && ow < W;
        const float* dyp = dy + (long long)b * D * H * W + (long long)oh * W + ow;      // + z * H * W
        auto dy_at = [&](int z) { return (live && z >= z0 && z < z1) ? dyp[(long long)z * H * W] : 0.0f; };
        __syncthreads();                                        // the previous item's last plane is no longer read
        fetch(z0 - 1);
        stash(0);
        fetch(z0);
        float g_prev = 0.0f, g_cur = dy_at(z0 - 1), g_next = dy_at(z0);       // dy[z-1], dy[z], dy[z+1] for x plane z = z0 - 1
        __syncthreads();
        int buf = 0;
        for (int z = z0 - 1; z <= z1; ++z) {
            const float* tp0 = &plane[buf][(lh * PW_HW + lw) * PW_STRIDE + cq * 4];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(tp0 + (kh * PW_HW + kw) * PW_STRIDE);
                    acc[0 * 9 + kh * 3 + kw] += xv * g_next;    // kd = 0: x[q + (-1)] with q = z + 1
                    acc[1 * 9 + kh * 3 + kw] += xv * g_cur;
                    acc[2 * 9 + kh * 3 + kw] += xv * g_prev;
                }
            g_prev = g_cur; g_cur = g_next; g_next = dy_at(z + 2);
            if (z < z1) {
                stash(buf ^ 1);
                if (z + 2 <= z1) fetch(z + 2);
            }
            __syncthreads();
            buf ^= 1;
        }
    }
    // ---- reduce: lanes of equal channel half (xor 2 .. 32), then the four waves, then 216 atomics
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < 27; ++t) {
        f32x4 v = acc[t];
#pragma unroll
        for (int msk = 2; msk < 64; msk <<= 1) {
            v.x += __shfl_xor(v.x, msk); v.y += __shfl_xor(v.y, msk); v.z += __shfl_xor(v.z, msk); v.w += __shfl_xor(v.w, msk);
        }
        if (lane < 2) { red[wave][lane][t * 4 + 0] = v.x; red[wave][lane][t * 4 + 1] = v.y; red[wave][lane][t * 4 + 2] = v.z; red[wave][lane][t * 4 + 3] = v.w; }
    }
    __syncthreads();
    if (threadIdx.x < 216) {
        const int half = threadIdx.x / 108, e = threadIdx.x % 108;      // e = tap * 4 + component
        const float v = (red[0][half][e] + red[1][half][e]) + (red[2][half][e] + red[3][half][e]);
        unsafeAtomicAdd(dw + (e >> 2) * 8 + half * 4 + (e & 3), v);
    }
}

// depth head backward, element-wise part (models/casmvsnet.py:299-300): logits -> softmax p -> depth = sum p_k d_k
//   d loss / d logit_k = p_k * (d_k - depth) * g,   g = d loss / d depth,   d_k = planes.d0 + k * planes.delta
__global__ void depth_head_bwd_kernel(const float* __restrict__ prob, const float* __restrict__ planes,
                                      const float* __restrict__ depth, const float* __restrict__ gdepth,
                                      float* __restrict__ dlogits, int D, long long hw, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long pix = i % hw;
        const long long bk = i / hw;
        const int k = (int)(bk % D);
        const long long b = bk / D;
        const float2 pl = reinterpret_cast<const float2*>(planes)[b * hw + pix];
        const float dk = pl.x + (float)k * pl.y;
        dlogits[i] = prob[i] * (dk - depth[b * hw + pix]) * gdepth[b * hw + pix];
    }
}

// data gradient of the single-output-channel `prob` convolution (models/modules.py:489: Conv3d(8, 1, 3, pad 1)):
//   dx[i][c] = sum_tap dy[i - tap + 1] * w[0][c][tap]        one thread per voxel, CI outputs in registers,
// the 27 neighbours of the 1-channel gradient come through L1, the weights through scalar loads.
template <int CI>
__global__ __launch_bounds__(256) void conv3d_dgrad_c1_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                              float* __restrict__ dx, int B, int D, int H, int W) {
    const long long total = (long long)B * D * H * W;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int iw = (int)(i % W), ih = (int)((i / W) % H), id = (int)((i / ((long long)W * H)) % D);
    const long long b = i / ((long long)W * H * D);
    float acc[CI];
#pragma unroll
    for (int c = 0; c < CI; ++c) acc[c] = 0.0f;
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
        const int od = id - (tap / 9) + 1, oh = ih - ((tap / 3) % 3) + 1, ow = iw - (tap % 3) + 1;
        float g = 0.0f;
        if ((unsigned)od < (unsigned)D && (unsigned)oh < (unsigned)H && (unsigned)ow < (unsigned)W)
            g = dy[((b * D + od) * H + oh) * W + ow];
#pragma unroll
        for (int c = 0; c < CI; ++c) acc[c] = fmaf(g, w[c * 27 + tap], acc[c]);
    }
#pragma unroll
    for (int c = 0; c < CI; c += 4)
        *reinterpret_cast<f32x4*>(dx + i * CI + c) = (f32x4){acc[c], acc[c + 1], acc[c + 2], acc[c + 3]};
}

// packed accumulation layout [27][P][Q] -> the parameter's layout (Q, Pk, 27), Pk <= P (padding channels dropped), and the packed
// buffer is cleared for its next use: one launch instead of a zero fill before and a permute copy after every weight gradient.
__global__ __launch_bounds__(256) void wgrad_finish_kernel(float* __restrict__ packed, float* __restrict__ out, int P, int Q, int Pk) {
    const int n = 27 * P * Q;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
        const int q = e % Q, p = (e / Q) % P, t = e / (Q * P);
        const float v = packed[e];
        packed[e] = 0.0f;
        if (p < Pk) out[((long long)q * Pk + p) * 27 + t] = v;
    }
}

}  // namespace rcmvs

using namespace rcmvs;

template <int CI, int CO, int STRIDE>
static int wgrad_launch(const float* x, const float* dy, float* dw, const WgradDims& dm, int ldx, int ci_off, int ci_total, hipStream_t st) {
    using Cfg = WgradCfg<CI, CO, STRIDE>;
    const int rows = dm.B * dm.Do * dm.Ho;
    int gx = (rows + WG_WAVES - 1) / WG_WAVES;
    constexpr int gx_cap = 1024;
    if (gx > gx_cap) gx = gx_cap;
    if (dm.D == 1 && STRIDE == 1) {
        // few rows (one-plane volumes): split each row into chunks of >= 64 cells until there are ~512 waves (more waves cost
        // more than they gain: every wave ends with an atomic flush of its whole accumulator set)
        int wchunks = 1;
        while (gx * WG_WAVES * wchunks < 512 && dm.Wo / (wchunks * 2) >= 64) wchunks *= 2;
        const int wchunk = ((dm.Wo + wchunks - 1) / wchunks + 3) / 4 * 4;
        hipLaunchKernelGGL((conv3d_wgrad_kernel<CI, CO, STRIDE, true>), dim3(gx, Cfg::SPLITS, (dm.Wo + wchunk - 1) / wchunk), dim3(WG_WAVES * 64), 0, st,
                           x, dy, dw, dm, rows, ldx, ci_off, ci_total, wchunk);
    } else {
        hipLaunchKernelGGL((conv3d_wgrad_kernel<CI, CO, STRIDE, false>), dim3(gx, Cfg::SPLITS), dim3(WG_WAVES * 64), 0, st, x, dy, dw, dm, rows, ldx,
                           ci_off, ci_total, dm.Wo);
    }
    return launch_status("conv3d_wgrad");
}

extern "C" {

int rcmvs_wgrad_finish(float* packed, float* out, int P, int Q, int Pk, void* stream) {
    RCMVS_REQUIRE(packed && out && P > 0 && Q > 0 && Pk > 0 && Pk <= P, "wgrad_finish: bad arguments");
    const int n = 27 * P * Q;
    hipLaunchKernelGGL(wgrad_finish_kernel, dim3((n + 255) / 256 > 512 ? 512 : (n + 255) / 256), dim3(256), 0, as_stream(stream), packed, out, P, Q, Pk);
    return launch_status("wgrad_finish");
}

int rcmvs_conv3d_wgrad(const float* x, const float* dy, float* dw, int B, int D, int H, int W, int Ci, int Co, int stride,
                       void* stream) {
    RCMVS_REQUIRE(x && dy && dw, "conv3d_wgrad: null pointer");
    RCMVS_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, "conv3d_wgrad: bad sizes");
    RCMVS_REQUIRE(stride == 1 || stride == 2, "conv3d_wgrad: stride must be 1 or 2");
    RCMVS_REQUIRE((long long)B * D * H * W * Ci * 4 < 0x7ffffff0LL, "conv3d_wgrad: activation tensor too large for 32-bit offsets");
    WgradDims dm{B, D, H, W, (D - 1) / stride + 1, (H - 1) / stride + 1, (W - 1) / stride + 1};
    hipStream_t st = as_stream(stream);
    if (Ci == 8 && Co == 1 && stride == 1) {
        const int tw_ = (W + PW_TW - 1) / PW_TW, th_ = (H + PW_TH - 1) / PW_TH, nzc = (D + PW_ZC - 1) / PW_ZC;
        const long long items = (long long)B * tw_ * th_ * nzc;
        if (items < 0x7fffffffLL) {
            const int grid = items < 1024 ? (int)items : 1024;             // persistent blocks: each ends with a 108-value butterfly
            hipLaunchKernelGGL(prob_wgrad_march_kernel, dim3(grid), dim3(256), 0, st, x, dy, dw, B, D, H, W, tw_, th_, nzc, (int)items);
            return launch_status("conv3d_wgrad(prob, marching)");
        }
        const int rows = B * D * H;
        int gx = (rows + 3) / 4;
        if (gx > 2048) gx = 2048;
        hipLaunchKernelGGL(conv3d_wgrad_c1_kernel, dim3(gx), dim3(256), 0, st, x, dy, dw, dm, rows);
        return launch_status("conv3d_wgrad_c1");
    }
#define RCMVS_WG(CI, CO, S) if (Ci == CI && Co == CO && stride == S) return wgrad_launch<CI, CO, S>(x, dy, dw, dm, Ci, 0, Ci, st);
    RCMVS_WG(8, 8, 1) RCMVS_WG(16, 8, 1) RCMVS_WG(32, 8, 1)
    RCMVS_WG(16, 16, 1) RCMVS_WG(32, 32, 1) RCMVS_WG(64, 64, 1)
    RCMVS_WG(32, 16, 1) RCMVS_WG(64, 32, 1) RCMVS_WG(16, 32, 1) RCMVS_WG(8, 32, 1)      /* FeatureNet layers as one-plane volumes */
    RCMVS_WG(8, 16, 2) RCMVS_WG(16, 32, 2) RCMVS_WG(32, 64, 2)
#undef RCMVS_WG
    if (Co == 8 && stride == 1 && Ci % 4 == 0 && Ci < 64) {
        // other input widths (the renderer's CostReg reads 41 channels padded to 44): 32 + 8 + 4 channel slices
        int off = 0, rc = 0;
        while (off < Ci && rc == 0) {
            const int left = Ci - off;
            if (left >= 32)      { rc = wgrad_launch<32, 8, 1>(x, dy, dw, dm, Ci, off, Ci, st); off += 32; }
            else if (left >= 16) { rc = wgrad_launch<16, 8, 1>(x, dy, dw, dm, Ci, off, Ci, st); off += 16; }
            else if (left >= 8)  { rc = wgrad_launch<8, 8, 1>(x, dy, dw, dm, Ci, off, Ci, st); off += 8; }
            else                 { rc = wgrad_launch<4, 8, 1>(x, dy, dw, dm, Ci, off, Ci, st); off += 4; }
        }
        return rc;
    }
    return fail(-1, "conv3d_wgrad: unsupported (Ci=%d, Co=%d, stride=%d)", Ci, Co, stride);
}

int rcmvs_conv3d_dgrad_c1(const float* dy, const float* w, float* dx, int B, int D, int H, int W, int Ci, void* stream) {
    RCMVS_REQUIRE(dy && w && dx, "conv3d_dgrad_c1: null pointer");
    RCMVS_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, "conv3d_dgrad_c1: bad sizes");
    RCMVS_REQUIRE(Ci == 8, "conv3d_dgrad_c1: Ci=%d unsupported (the prob layer has 8 input channels)", Ci);
    const long long total = (long long)B * D * H * W;
    hipLaunchKernelGGL((conv3d_dgrad_c1_kernel<8>), dim3((unsigned)cdiv(total, 256LL)), dim3(256), 0, as_stream(stream), dy, w, dx, B, D, H, W);
    return launch_status("conv3d_dgrad_c1");
}

int rcmvs_depth_head_bwd(const float* prob, const float* planes, const float* depth, const float* grad_depth,
                         float* grad_logits, int B, int D, int h, int w, void* stream) {
    RCMVS_REQUIRE(prob && planes && depth && grad_depth && grad_logits, "depth_head_bwd: null pointer");
    RCMVS_REQUIRE(B > 0 && D > 0 && h > 0 && w > 0, "depth_head_bwd: bad sizes");
    const long long hw = (long long)h * w, total = hw * D * B;
    long long g = cdiv(total, 256LL);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(depth_head_bwd_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream), prob, planes, depth, grad_depth,
                       grad_logits, D, hw, total);
    return launch_status("depth_head_bwd");
}

}  // extern "C"
