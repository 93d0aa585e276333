// Stride-1 3x3x3 convolution on the bf16 matrix cores at fp32 accuracy ("x3" = three-way operand split), channels-last,
// gfx950 only.  Replaces Conv3d.forward of the stride-1 CostRegNet layers (models/modules.py:149-157, 470-501): conv0
// (Cin = 8/16/32 -> 8, half of the 3-D U-Net's flops), conv2 (16 -> 16) and conv4 (32 -> 32).
//
// Arithmetic.  fp32 MFMA runs at the fp32 vector rate (157 TF); v_mfma_f32_16x16x32_bf16 runs 16x faster.  Every fp32
// operand is split EXACTLY into three bf16 pieces by truncation, x = h + m + l (8 + 8 + 8 significant bits:
// h = x & 0xffff0000, m = (x - h) & 0xffff0000, l = x - h - m; both subtractions are exact), and the product is formed from
// six bf16 MFMAs with fp32 accumulation:   x*w ~= xh*wh + (xh*wm + xm*wh) + (xh*wl + xl*wh + xm*wm).
// The three dropped terms are bounded by 2^-23 |x||w| -- the size of ONE fp32 rounding of the product -- and the three
// magnitude classes are accumulated in separate fp32 accumulators that are added once at the end (small terms never meet a
// large partial sum).  Measured against an fp64 convolution the result is as close as the fp32 FMA-chain kernels
// (tests/test_gpu_parity.py::test_conv3d_x3_vs_fp64).  Six MFMAs at the bf16 rate = 2.6x the fp32 peak.
//
// GEMM view per wave:  D[16 x 16] += A[16 x 32] * B[32 x 16] with A = weights (register-stationary for the whole kernel),
//   B = activations read from LDS with one ds_read_b128 per piece, N = 16 output columns (voxels).
//   Cout = 16:  M = output channel.                                                     ("plain", X3_PL)
//   Cout = 8 :  M = (s, co) -- TWO output positions share one column of activations: position y0 + s (Cin = 16/32, X3_YT,
//               the K axis walks kh' = kh + s = 0..3) or x0 + 2n + s (Cin = 8, X3_XT, K walks kw' = kw + s = 0..3, columns
//               are every second voxel).  A holds the weight of tap k' - s, zero where that tap does not exist
//               (block-Toeplitz): the 16-row tile is full at 3/4 density instead of half empty.
//   K step = 32 = (32 / Cin) tap positions x Cin channels; lane (n = l & 15, kk = l >> 4) supplies channels 8*kk.. of its position.
//
// Data flow.  A block owns a TY x 32 output tile and marches over z: three input z-slices (halo TY+2 x 34 voxels, already
// split into the three bf16 piece planes) sit in an LDS ring; slice z+2 is fetched into registers (raw buffer loads, out of
// range -> 0 = the zero padding) before the MFMA phase of slice z and split + written to the ring after it, so each input
// voxel is read from L2/HBM once per block (x 1.3-1.6 halo) and split once instead of once per tap.
// Weights: K is cut into KSPLIT slices over the block's waves when the whole image does not fit the register file
// (Cin = 32); the partial 16 x 16 tiles are exchanged through LDS.
#include "common.h"

namespace rcmvs {

typedef __bf16 x3_bf16x8 __attribute__((ext_vector_type(8)));
typedef float x3_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int x3_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int x3_u32x2 __attribute__((ext_vector_type(2)));

enum { X3_XT = 0, X3_YT = 1, X3_PL = 2 };

template <int CIN, int COUT>
struct X3 {
    static constexpr int MODE = (COUT == 8) ? (CIN == 8 ? X3_XT : X3_YT) : X3_PL;
    static constexpr int MT = (COUT + 15) / 16;                          // 16-row tiles of A
    static constexpr int VB = CIN * 2;                                   // bytes per voxel per piece plane
    static constexpr int PPS = (MODE == X3_XT) ? 4 : (CIN >= 32 ? 1 : 32 / CIN);   // tap positions per K step
    static constexpr int HALVES = (CIN > 32) ? CIN / 32 : 1;            // K steps per position
    static constexpr int PPKD = (MODE == X3_PL) ? 9 : 12;               // positions per kd plane (3x3, or 4x3 / 3x4 Toeplitz)
    static constexpr int QC = (MODE == X3_XT) ? 4 : 3;                  // columns of the position grid
    static constexpr int SPK = ((PPKD + PPS - 1) / PPS) * HALVES;       // K steps per kd plane (a step never straddles kd)
    static constexpr int KSTEPS = 3 * SPK;
    static constexpr int KSPLIT = (KSTEPS * MT * 12 > 128) ? ((KSTEPS * MT * 12 > 256) ? 4 : 2) : 1;   // <= ~110 weight registers per wave
    static constexpr int KSW = (KSTEPS + KSPLIT - 1) / KSPLIT;          // K steps per wave
    static constexpr int TX = (CIN >= 32) ? 16 : 32;
    static constexpr int TY = (MODE == X3_XT) ? 8 : 4;
    static constexpr int TXP = TX + 2, TYP = TY + 2;
    static constexpr int ROWB = TXP * VB;                                // bytes per halo row
    static constexpr int PLB = TYP * ROWB;                               // bytes per piece plane
    static constexpr int SLB = 3 * PLB;                                  // bytes per z-slice (h, m, l planes)
    static constexpr int NSLOT = 4;                                      // ring: slices z-1, z, z+1 being read + z+2 being written
    static constexpr int NTX = (MODE == X3_XT) ? TX / 32 : TX / 16;      // n-tiles along x
    static constexpr int NTILE = ((MODE == X3_YT) ? TY / 2 : TY) * NTX;
    static constexpr int NG = 4 / KSPLIT;                                // tile groups (consumer waves that own different n-tiles)
    static constexpr int NTW = NTILE / NG;                               // n-tiles per consumer wave
    static constexpr int Q4 = CIN / 4;                                   // float4 per voxel
    static constexpr int NLOAD = TYP * TXP * Q4;                         // float4 per z-slice
    static constexpr int NPF = (NLOAD + 255) / 256;                      // float4 per producer thread per z-slice
    static constexpr int PARTB = NTILE * MT * KSPLIT * 1024;             // one buffer of (partial) output tiles
    static constexpr int LDSB = NSLOT * SLB + 2 * PARTB;
    static_assert(NTW % 2 == 0, "tiles are processed in pairs");
    static_assert(LDSB <= 160 * 1024, "LDS budget");
};

// ---- weight image: [K step][piece][m-tile][lane][8 bf16], the A fragment of v_mfma_f32_16x16x32_bf16 (row = lane & 15,
// k = 8 * (lane >> 4) + e), pieces split by truncation like the activations.
template <int CIN, int COUT>
__global__ void x3_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ img, int transposed) {
    using C = X3<CIN, COUT>;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= C::KSTEPS * C::MT * 64 * 8) return;
    const int e = t & 7, lane = (t >> 3) & 63, mt = (t >> 9) % C::MT, j = (t >> 9) / C::MT;
    const int m = lane & 15, kk = lane >> 4;
    const int kd = j / C::SPK, js = j % C::SPK;
    int q, ci;
    if (C::HALVES == 1) { q = js * C::PPS + kk / (4 / C::PPS); ci = (kk % (4 / C::PPS)) * 8 + e; }
    else { q = js / C::HALVES; ci = (js % C::HALVES) * 32 + kk * 8 + e; }
    float v = 0.0f;
    if (q < C::PPKD) {
        const int r = q / C::QC, c = q % C::QC;
        int co, kh, kw;
        if (C::MODE == X3_XT) { co = m & 7; kh = r; kw = c - (m >> 3); }
        else if (C::MODE == X3_YT) { co = m & 7; kh = r - (m >> 3); kw = c; }
        else { co = mt * 16 + m; kh = r; kw = c; }
        if (kh >= 0 && kh < 3 && kw >= 0 && kw < 3 && co < COUT) {
            const int tap = (kd * 3 + kh) * 3 + kw;
            v = transposed ? w[((long long)ci * COUT + co) * 27 + (transposed == 2 ? 26 - tap : tap)] : w[((long long)co * CIN + ci) * 27 + tap];
        }
    }
    const unsigned hb = __float_as_uint(v) & 0xffff0000u;
    const float r1 = v - __uint_as_float(hb);
    const unsigned mb = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(mb);
    const unsigned lb = __float_as_uint(r2) & 0xffff0000u;
    const long long base = (((long long)j * 3) * C::MT + mt) * 512 + lane * 8 + e;     // piece stride = MT * 512 shorts
    img[base] = (unsigned short)(hb >> 16);
    img[base + (long long)C::MT * 512] = (unsigned short)(mb >> 16);
    img[base + 2LL * C::MT * 512] = (unsigned short)(lb >> 16);
}

// four fp32 -> the three bf16 piece quadruples (two dwords each)
__device__ __forceinline__ void x3_split4(x3_f32x4 v, x3_u32x2& h, x3_u32x2& m, x3_u32x2& l) {
    unsigned hb[4], mb[4], lb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float f = v[i];
        hb[i] = __float_as_uint(f) & 0xffff0000u;
        const float r1 = f - __uint_as_float(hb[i]);
        mb[i] = __float_as_uint(r1) & 0xffff0000u;
        const float r2 = r1 - __uint_as_float(mb[i]);
        lb[i] = __float_as_uint(r2);
    }
    // v_perm_b32: (hi16 of b) << 16 | (hi16 of a)
    h.x = __builtin_amdgcn_perm(hb[1], hb[0], 0x07060302u); h.y = __builtin_amdgcn_perm(hb[3], hb[2], 0x07060302u);
    m.x = __builtin_amdgcn_perm(mb[1], mb[0], 0x07060302u); m.y = __builtin_amdgcn_perm(mb[3], mb[2], 0x07060302u);
    l.x = __builtin_amdgcn_perm(lb[1], lb[0], 0x07060302u); l.y = __builtin_amdgcn_perm(lb[3], lb[2], 0x07060302u);
}

// 512 threads = 4 consumer waves (MFMA only) + 4 producer waves (fetch / split / LDS stores, and the epilogue of the previous
// slice), one barrier per output z-slice.  Each SIMD hosts one consumer and one producer wave, so the producers' VALU / LDS /
// memory work runs in the shadow of the matrix pipe.
template <int CIN, int COUT>
__global__ __launch_bounds__(512) void conv3d_x3_kernel(
    const float* __restrict__ x, const x3_u32x4* __restrict__ wimg, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ res, float* __restrict__ y,
    int D, int H, int W, int tiles_x, int zchunk, int relu) {
    using C = X3<CIN, COUT>;
    constexpr int MT = C::MT, KSW = C::KSW, KSPLIT = C::KSPLIT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const partbase = smem + C::NSLOT * C::SLB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 4;
    const int n = lane & 15, kk = lane >> 4;
    const int b = blockIdx.z;
    const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
    const int x0 = (int)(tile % tiles_x) * C::TX, y0 = (int)(tile / tiles_x) * C::TY;
    const int zb = blockIdx.y * zchunk, ze = min(D, zb + zchunk);

    if (!producer) {
        // =============================== consumer: register-stationary weights of this wave's K slice
        const int ks = wave % KSPLIT, grp = wave / KSPLIT;
        x3_bf16x8 wr[KSW][3][MT];
        int kdj[KSW];      // kd plane of K step j (wave-uniform)
        int boff[KSW];     // this lane's byte offset of the B fragment inside a z-slice, tile origin excluded
#pragma unroll
        for (int j = 0; j < KSW; ++j) {
            const int jg = ks * KSW + j;
            const bool live = jg < C::KSTEPS;
            const int jc = live ? jg : 0;
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    x3_u32x4 v = wimg[((jc * 3 + p) * MT + mt) * 64 + lane];
                    if (!live) v = (x3_u32x4){0u, 0u, 0u, 0u};
                    wr[j][p][mt] = __builtin_bit_cast(x3_bf16x8, v);
                }
            kdj[j] = jc / C::SPK;
            const int js = jc % C::SPK;
            int q, ci0;
            if (C::HALVES == 1) { q = js * C::PPS + kk / (4 / C::PPS); ci0 = (kk % (4 / C::PPS)) * 8; }
            else { q = js / C::HALVES; ci0 = (js % C::HALVES) * 32 + kk * 8; }
            if (q >= C::PPKD) q = 0;
            const int r = q / C::QC, c = q % C::QC;
            boff[j] = r * C::ROWB + (c + n * (C::MODE == X3_XT ? 2 : 1)) * C::VB + ci0 * 2;
        }
        __syncthreads();          // prologue slices are in the ring
        int s0 = 0;               // ring slot of slice z-1
#pragma unroll 1
        for (int z = zb; z < ze; ++z) {
            int slotoff[3];
            slotoff[0] = s0 * C::SLB;
            slotoff[1] = ((s0 + 1) & 3) * C::SLB;
            slotoff[2] = ((s0 + 2) & 3) * C::SLB;
            x3_f32x4* part = reinterpret_cast<x3_f32x4*>(partbase + ((z - zb) & 1) * C::PARTB);
#pragma unroll
            for (int tp = 0; tp < C::NTW / 2; ++tp) {
                int toff[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int tl = grp + C::NG * (2 * tp + t);
                    const int trow = tl / C::NTX, tcol = tl % C::NTX;
                    toff[t] = ((C::MODE == X3_YT) ? 2 * trow : trow) * C::ROWB + tcol * ((C::MODE == X3_XT) ? 32 : 16) * C::VB;
                }
                x3_f32x4 acc[2][MT][3];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int a = 0; a < 3; ++a) acc[t][mt][a] = (x3_f32x4){0.f, 0.f, 0.f, 0.f};
                // software pipeline: the six B fragments of K step j+1 are read while the twelve MFMAs of step j run; the MFMA
                // order keeps >= 3 independent instructions between two uses of one accumulator
                x3_bf16x8 bq[2][2][3];
                {
                    const int a0 = boff[0] + slotoff[kdj[0]];
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int p = 0; p < 3; ++p) bq[0][t][p] = *reinterpret_cast<const x3_bf16x8*>(smem + a0 + toff[t] + p * C::PLB);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < KSW; ++j) {
                    const int cur = j & 1, nxt = cur ^ 1;
                    const bool pre = j + 1 < KSW;
                    const unsigned char* np0 = smem + (pre ? boff[j + 1] + slotoff[kdj[j + 1]] : 0) + toff[0];
                    const unsigned char* np1 = smem + (pre ? boff[j + 1] + slotoff[kdj[j + 1]] : 0) + toff[1];
                    // six slots: one B-fragment read of K step j+1, then two MFMAs per m-tile of step j; the fences pin this
                    // order (left alone, the scheduler sinks every read to just before its first use and exposes the LDS latency)
#define X3_MF(T, ACC, WP, BP) _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) acc[T][mt][ACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[j][WP][mt], bq[cur][T][BP], acc[T][mt][ACC], 0, 0, 0)
#define X3_LD(T, P) if (pre) bq[nxt][T][P] = *reinterpret_cast<const x3_bf16x8*>((T ? np1 : np0) + P * C::PLB)
                    X3_LD(0, 0); X3_MF(0, 2, 0, 2); X3_MF(1, 2, 0, 2); __builtin_amdgcn_sched_barrier(0);
                    X3_LD(1, 0); X3_MF(0, 1, 0, 1); X3_MF(1, 1, 0, 1); __builtin_amdgcn_sched_barrier(0);
                    X3_LD(0, 1); X3_MF(0, 0, 0, 0); X3_MF(1, 0, 0, 0); __builtin_amdgcn_sched_barrier(0);
                    X3_LD(1, 1); X3_MF(0, 2, 1, 1); X3_MF(1, 2, 1, 1); __builtin_amdgcn_sched_barrier(0);
                    X3_LD(0, 2); X3_MF(0, 1, 1, 0); X3_MF(1, 1, 1, 0); __builtin_amdgcn_sched_barrier(0);
                    X3_LD(1, 2); X3_MF(0, 2, 2, 0); X3_MF(1, 2, 2, 0); __builtin_amdgcn_sched_barrier(0);
#undef X3_MF
#undef X3_LD
                }
                // hand the (partial) tiles to the producers: [tile][m-tile][K slice][lane]
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const int tl = grp + C::NG * (2 * tp + t);
                        part[((tl * MT + mt) * KSPLIT + ks) * 64 + lane] = acc[t][mt][0] + (acc[t][mt][1] + acc[t][mt][2]);
                    }
            }
            __syncthreads();
            s0 = (s0 + 1) & 3;
        }
    } else {
        // =============================== producer
        const int pw = wave - 4, ptid = tid - 256;
        constexpr int OOB = 0x7ffffff0;
        const long long vol = (long long)D * H * W * CIN * 4;
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (long long)b * D * H * W * CIN), (short)0, (int)vol, 0x00020000);
        int goff[C::NPF], loff[C::NPF];
#pragma unroll
        for (int i = 0; i < C::NPF; ++i) {
            const int e = ptid + i * 256;
            const int v = e / C::Q4, c4 = e - v * C::Q4;
            const int hr = v / C::TXP, hc = v - hr * C::TXP;
            const int gy = y0 + hr - 1, gx = x0 + hc - 1;
            const bool ok = e < C::NLOAD && gy >= 0 && gy < H && gx >= 0 && gx < W;
            goff[i] = ok ? ((gy * W + gx) * CIN + c4 * 4) * 4 : OOB;
            loff[i] = (e < C::NLOAD) ? hr * C::ROWB + hc * C::VB + c4 * 8 : -1;
        }
        const int zstride = H * W * CIN * 4;
        auto fetch = [&](x3_f32x4 (&pf)[C::NPF], int z) {
            const bool zin = z >= 0 && z < D;
#pragma unroll
            for (int i = 0; i < C::NPF; ++i) {
                const int off = (zin && goff[i] != OOB) ? goff[i] + z * zstride : OOB;
                pf[i] = __builtin_bit_cast(x3_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
            }
        };
        auto stash = [&](const x3_f32x4 (&pf)[C::NPF], int slot) {
            unsigned char* sb = smem + slot * C::SLB;
#pragma unroll
            for (int i = 0; i < C::NPF; ++i) {
                if (loff[i] < 0) continue;
                x3_u32x2 h, m, l;
                x3_split4(pf[i], h, m, l);
                *reinterpret_cast<x3_u32x2*>(sb + loff[i]) = h;
                *reinterpret_cast<x3_u32x2*>(sb + C::PLB + loff[i]) = m;
                *reinterpret_cast<x3_u32x2*>(sb + 2 * C::PLB + loff[i]) = l;
            }
        };
        // epilogue constants of this lane (D layout of the 16 x 16 tile: column n, rows 4*kk .. 4*kk+3)
        const int sft = (C::MODE == X3_PL) ? 0 : (kk >> 1);
        x3_f32x4 sc[MT], sh[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int co0 = (C::MODE == X3_PL) ? mt * 16 + 4 * kk : (kk & 1) * 4;
            sc[mt] = scale ? *reinterpret_cast<const x3_f32x4*>(scale + co0) : (x3_f32x4){1.f, 1.f, 1.f, 1.f};
            sh[mt] = shift ? *reinterpret_cast<const x3_f32x4*>(shift + co0) : (x3_f32x4){0.f, 0.f, 0.f, 0.f};
        }
        auto epilogue = [&](int z) {
            const x3_f32x4* part = reinterpret_cast<const x3_f32x4*>(partbase + ((z - zb) & 1) * C::PARTB);
#pragma unroll
            for (int i = 0; i < (C::NTILE + 3) / 4; ++i) {
                const int tl = pw + 4 * i;
                if (tl >= C::NTILE) continue;
                const int trow = tl / C::NTX, tcol = tl % C::NTX;
                int oy, ox;
                if (C::MODE == X3_XT) { oy = y0 + trow; ox = x0 + tcol * 32 + 2 * n + sft; }
                else if (C::MODE == X3_YT) { oy = y0 + 2 * trow + sft; ox = x0 + tcol * 16 + n; }
                else { oy = y0 + trow; ox = x0 + tcol * 16 + n; }
                if (oy >= H || ox >= W) continue;
                const long long ov = (((long long)b * D + z) * H + oy) * W + ox;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const x3_f32x4* pp = part + (tl * MT + mt) * KSPLIT * 64 + lane;
                    x3_f32x4 v = pp[0];
#pragma unroll
                    for (int k = 1; k < KSPLIT; ++k) v += pp[k * 64];
                    const int co0 = (C::MODE == X3_PL) ? mt * 16 + 4 * kk : (kk & 1) * 4;
                    v = v * sc[mt] + sh[mt];
                    if (relu) v = __builtin_elementwise_max(v, (x3_f32x4){0.f, 0.f, 0.f, 0.f});
                    if (res) v += *reinterpret_cast<const x3_f32x4*>(res + ov * COUT + co0);
                    *reinterpret_cast<x3_f32x4*>(y + ov * COUT + co0) = v;
                }
            }
        };
        // prologue: slices zb-1, zb, zb+1 -> ring slots 0, 1, 2 (one batch of loads), slice zb+2 stays in registers
        x3_f32x4 pf[C::NPF];
        {
            x3_f32x4 p0[C::NPF], p1[C::NPF], p2[C::NPF];
            fetch(p0, zb - 1); fetch(p1, zb); fetch(p2, zb + 1);
            fetch(pf, zb + 2);
            stash(p0, 0); stash(p1, 1); stash(p2, 2);
        }
        __syncthreads();
#pragma unroll 1
        for (int z = zb; z < ze; ++z) {
            // slice z+2 (read by the consumers from the next iteration on) -> the slot that slice z-2 left
            if (z + 1 < ze) {
                stash(pf, (z - zb + 3) & 3);
                if (z + 2 < ze) fetch(pf, z + 3);
            }
            if (z > zb) epilogue(z - 1);
            __syncthreads();
        }
        epilogue(ze - 1);
    }
}

bool conv3d_x3_supported(int Ci, int Co) {
    return (Co == 8 && (Ci == 8 || Ci == 16 || Ci == 32)) || (Co == 16 && Ci == 16);
}

long long conv3d_x3_weight_floats(int Ci, int Co) {      // size of the x3 image in floats (it is stored as bf16 triples)
#define RCMVS_X3_SZ(CI, CO) if (Ci == CI && Co == CO) return (long long)X3<CI, CO>::KSTEPS * 3 * X3<CI, CO>::MT * 64 * 8 / 2;
    RCMVS_X3_SZ(8, 8) RCMVS_X3_SZ(16, 8) RCMVS_X3_SZ(32, 8) RCMVS_X3_SZ(16, 16)
#undef RCMVS_X3_SZ
    return 0;
}

int conv3d_x3_pack(const float* w, float* img, int Co, int Ci, int transposed, hipStream_t st) {
#define RCMVS_X3_PK(CI, CO) if (Ci == CI && Co == CO) { \
        const int nthr = X3<CI, CO>::KSTEPS * X3<CI, CO>::MT * 64 * 8; \
        hipLaunchKernelGGL((x3_pack_kernel<CI, CO>), dim3((nthr + 255) / 256), dim3(256), 0, st, w, reinterpret_cast<unsigned short*>(img), transposed); \
        return launch_status("conv3d_x3_pack"); }
    RCMVS_X3_PK(8, 8) RCMVS_X3_PK(16, 8) RCMVS_X3_PK(32, 8) RCMVS_X3_PK(16, 16)
#undef RCMVS_X3_PK
    return fail(-1, "conv3d_x3_pack: unsupported Ci=%d Co=%d", Ci, Co);
}

int conv3d_x3_launch(const float* x, const float* wimg, const float* scale, const float* shift, const float* res, float* y,
                     int B, int D, int H, int W, int Ci, int Co, int relu, hipStream_t st) {
    if ((long long)D * H * W * Ci * 4 >= 0x7ffffff0LL) return fail(-1, "conv3d_x3: input volume too large for 32-bit offsets");
#define RCMVS_X3_RUN(CI, CO) if (Ci == CI && Co == CO) { \
        using C = X3<CI, CO>; \
        const int tiles_x = (W + C::TX - 1) / C::TX, tiles_y = (H + C::TY - 1) / C::TY; \
        /* z chunks: enough blocks to fill 256 CUs a few times over, but at least 8 slices per chunk (3-slice prologue) */ \
        int zchunk = D; \
        while (zchunk > 8 && (long long)tiles_x * tiles_y * B * ((D + zchunk - 1) / zchunk) < 1024) zchunk = (zchunk + 1) / 2; \
        dim3 grid(tiles_x * tiles_y, (D + zchunk - 1) / zchunk, B); \
        static bool attr_set = false; \
        if (!attr_set) { (void)hipFuncSetAttribute((const void*)conv3d_x3_kernel<CI, CO>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDSB); attr_set = true; } \
        hipLaunchKernelGGL((conv3d_x3_kernel<CI, CO>), grid, dim3(512), C::LDSB, st, x, reinterpret_cast<const x3_u32x4*>(wimg), scale, shift, res, y, D, H, W, tiles_x, zchunk, relu); \
        return launch_status("conv3d_x3"); }
    RCMVS_X3_RUN(8, 8) RCMVS_X3_RUN(16, 8) RCMVS_X3_RUN(32, 8) RCMVS_X3_RUN(16, 16)
#undef RCMVS_X3_RUN
    return fail(-1, "conv3d_x3: unsupported Ci=%d Co=%d", Ci, Co);
}

}  // namespace rcmvs
