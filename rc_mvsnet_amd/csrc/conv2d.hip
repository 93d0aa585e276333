// 2-D convolution family of the feature pyramid (FeatureNet, arch_mode='fpn', models/modules.py:363-464),
// channels-last, for inference: conv + folded BatchNorm (or bias) + ReLU, and the FPN merge
//   intra = nearest_upsample_x2(prev) + conv1x1(lateral) + bias          (modules.py:448-455)
// fused as an "upsample-add" epilogue.  Same mapping as conv3d_lds.hip: one thread per output pixel,
// all Cout accumulators in registers, weights wave-uniform (scalar loads, v_pk_fma with SGPR operands),
// input halo tile staged in LDS 8 channels at a time at a conflict-free padded stride.  The outputs
// (B*V, h, w, C) are exactly the channels-last maps K1 consumes, so no layout pass is needed.
// gfx950 only.
#include "common.h"
#include "x3_pieces.h"

namespace rcmvs {

typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int C2_CK = 8;                 // channels staged per pass
constexpr int C2_STRIDE = C2_CK + 4;     // floats per staged pixel

// TH x TW threads (TH*TW == 256), PPT vertically stacked output pixels per thread (rows ly and ly + TH): with two
// pixels every scalar weight feeds two FMAs, which halves the scalar-cache traffic that bounds the Cout = 32 layers.
// The tap loops stay rolled (UKY / UKX = 1) where a fully unrolled body would need more weights than there are SGPRs:
// the compiler otherwise hoists all K*K*CK*CO scalar loads and spills them through v_writelane/v_readlane (2560 spill
// instructions against 288 packed FMAs in the 8 -> 8 layer: 95 -> 24 us once rolled).
// RGB3 (first layer only, CI = 4): x is the network's input as it arrives, (N, 3, H, W) planar -- the halo tile is staged straight from
// the three planes (zero fourth channel), which saves the NCHW -> NHWC4 pass over the images (rcmvs_rgb_to_nhwc4: a launch and 27 MB).
template <int CI, int CO, int K, int S, int TH, int TW, int PPT, int UKY, int UKX, bool RGB3 = false>
__global__ __launch_bounds__(256) void conv2d_lds_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ up, float* __restrict__ y,
    int H, int W, int Ho, int Wo, int tiles_w, int relu) {
    static_assert(!RGB3 || CI == 4, "planar RGB input feeds the 4-channel first layer");
    static_assert(TH * TW == 256, "tile must have 256 threads");
    constexpr int PAD = K / 2;
    constexpr int HH = (TH * PPT - 1) * S + K, HW = (TW - 1) * S + K;     // halo tile
    constexpr int CKW = (CI < C2_CK ? CI : C2_CK) * CO;                   // scalar weights per tap and channel chunk
    // UKY / UKX: unroll counts of the two tap loops, chosen per layer from measurements (rcmvs_conv2d_fwd's table)
    extern __shared__ __attribute__((aligned(16))) float tile[];          // [HH*HW][C2_STRIDE]
    const int n = blockIdx.y;
    const unsigned t2 = xcd_remap(blockIdx.x, gridDim.x);
    const int tw = t2 % tiles_w, th = t2 / tiles_w;
    const int oy0 = th * TH * PPT, ox0 = tw * TW;
    const int lx = threadIdx.x % TW, ly = threadIdx.x / TW;
    const int ox = ox0 + lx;
    const float* xb = x + (long long)n * H * W * (RGB3 ? 3 : CI);
    const int iy0 = oy0 * S - PAD, ix0 = ox0 * S - PAD;

    // output channel PAIRS per v_pk_fma_f32: acc2[co / 2] += splat(x) * (w[co], w[co + 1]) -- the splat is an operand selector of the
    // packed instruction, the weight pair two neighbouring SGPRs: half the VALU instructions of the scalar form, the same arithmetic
    // per accumulator (round 3; fpn_fused.hip likewise)
    static_assert(CO % 2 == 0, "output channels are accumulated in pairs");
    typedef float f2v __attribute__((ext_vector_type(2)));
    f2v acc2[PPT][CO / 2];
#pragma unroll
    for (int p = 0; p < PPT; ++p)
#pragma unroll
        for (int c = 0; c < CO / 2; ++c) acc2[p][c] = (f2v){0.0f, 0.0f};

    for (int c0 = 0; c0 < CI; c0 += C2_CK) {
        const int ck = (CI - c0 < C2_CK) ? (CI - c0) : C2_CK;       // multiple of 4
        const int q = ck >> 2;
        if (c0 > 0) __syncthreads();
        for (int e = threadIdx.x; e < HH * HW * q; e += 256) {
            const int v = e / q, c4 = e - v * q;
            const int hx = v % HW, hy = v / HW;
            const int iy = iy0 + hy, ix = ix0 + hx;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                if constexpr (RGB3) {
                    const long long pix = (long long)iy * W + ix, hw = (long long)H * W;
                    val = make_float4(xb[pix], xb[hw + pix], xb[2 * hw + pix], 0.0f);
                } else {
                    val = *reinterpret_cast<const float4*>(xb + ((long long)iy * W + ix) * CI + c0 + c4 * 4);
                }
            }
            *reinterpret_cast<float4*>(tile + v * C2_STRIDE + c4 * 4) = val;
        }
        __syncthreads();
#pragma unroll UKY
        for (int ky = 0; ky < K; ++ky) {
#pragma unroll UKX
            for (int kx = 0; kx < K; ++kx) {
                const float* wt = wp + ((long long)(ky * K + kx) * CI + c0) * CO;
#pragma unroll
                for (int c4 = 0; c4 < C2_CK / 4; ++c4) {
                    if (c4 * 4 < ck) {
                        f4v xv[PPT];
#pragma unroll
                        for (int p = 0; p < PPT; ++p)
                            xv[p] = *reinterpret_cast<const f4v*>(tile + (((ly + p * TH) * S + ky) * HW + (lx * S + kx)) * C2_STRIDE + c4 * 4);
                        if constexpr (PPT > 1 || CO <= 8) {
                            // weights consumed in memory order ([ci][co], co fastest): one s_load_dwordx16 = 16 output
                            // channels of one input channel, used up before the next block is needed
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
#pragma unroll
                                for (int co = 0; co < CO; co += 2) {
                                    if constexpr (K > 1) {
                                        const f2v wv = (f2v){wt[(c4 * 4 + j) * CO + co], wt[(c4 * 4 + j) * CO + co + 1]};
#pragma unroll
                                        for (int p = 0; p < PPT; ++p) acc2[p][co / 2] = __builtin_elementwise_fma((f2v){xv[p][j], xv[p][j]}, wv, acc2[p][co / 2]);
                                    } else {      // 1x1 layers: the packed form makes the optimiser hoist every weight of the layer (233 SGPRs spilled to VGPR lanes)
                                        const float w0 = wt[(c4 * 4 + j) * CO + co], w1 = wt[(c4 * 4 + j) * CO + co + 1];
#pragma unroll
                                        for (int p = 0; p < PPT; ++p) {
                                            acc2[p][co / 2].x = fmaf(xv[p][j], w0, acc2[p][co / 2].x);
                                            acc2[p][co / 2].y = fmaf(xv[p][j], w1, acc2[p][co / 2].y);
                                        }
                                    }
                                }
                            }
                        } else {
                            // one pixel, wide Cout: four input channels per accumulator back to back (measured faster
                            // for the 32 -> 32 / 32 -> 16 3x3 layers: 69 vs 118 us)
#pragma unroll
                            for (int co = 0; co < CO; co += 2) {
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    acc2[0][co / 2] = __builtin_elementwise_fma((f2v){xv[0][j], xv[0][j]}, (f2v){wt[(c4 * 4 + j) * CO + co], wt[(c4 * 4 + j) * CO + co + 1]}, acc2[0][co / 2]);
                            }
                        }
                    }
                }
            }
        }
    }
    float acc[PPT][CO];
#pragma unroll
    for (int p = 0; p < PPT; ++p)
#pragma unroll
        for (int c = 0; c < CO / 2; ++c) { acc[p][2 * c] = acc2[p][c].x; acc[p][2 * c + 1] = acc2[p][c].y; }
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
        const int oy = oy0 + ly + p * TH;
        if (!(oy < Ho && ox < Wo)) continue;
        const long long op = ((long long)n * Ho + oy) * Wo + ox;
        float* yp = y + op * CO;
        const float* upp = up ? up + (((long long)n * (Ho / 2) + oy / 2) * (Wo / 2) + ox / 2) * CO : nullptr;
        if (scale) {
#pragma unroll
            for (int co = 0; co < CO; ++co) acc[p][co] = acc[p][co] * scale[co];
        }
        if (shift) {
#pragma unroll
            for (int co = 0; co < CO; ++co) acc[p][co] = acc[p][co] + shift[co];
        }
        if (upp) {                                                      // F.interpolate(intra) + inner(conv)
#pragma unroll
            for (int co = 0; co < CO; co += 4) {
                const f4v u4 = *reinterpret_cast<const f4v*>(upp + co);
                acc[p][co] = u4.x + acc[p][co]; acc[p][co + 1] = u4.y + acc[p][co + 1];
                acc[p][co + 2] = u4.z + acc[p][co + 2]; acc[p][co + 3] = u4.w + acc[p][co + 3];
            }
        }
        if (relu) {
#pragma unroll
            for (int co = 0; co < CO; ++co) acc[p][co] = fmaxf(acc[p][co], 0.0f);
        }
#pragma unroll
        for (int co = 0; co < CO; co += 4)
            *reinterpret_cast<float4*>(yp + co) = make_float4(acc[p][co], acc[p][co + 1], acc[p][co + 2], acc[p][co + 3]);
    }
}

// ---- 1x1 layers (round 4): a streaming kernel.  The tile kernel above stages a halo in LDS 8 channels at a time behind barriers, which a
// 1x1 conv has no use for: the 16 -> 32 lateral merge ran 22 us per scene for 55 MB of traffic and 0.13 GFLOP (fewer than two blocks per
// CU, four staging rounds each).  Here a thread owns one pixel: its CI channels arrive with CI / 4 16-byte loads, the CI x CO weights are
// wave-uniform scalar operands consumed one input channel at a time (CO SGPRs live), all CO accumulators stay in registers; bias /
// BatchNorm, the nearest x2 up-add of the FPN merge, ReLU and -- optionally -- the block's (max |y|)^2 into a bound vector (one atomic max
// per block: the bound of the variance volume, csrc/absmax.hip) follow in the epilogue.  Same accumulation order (ascending input channel,
// fmaf) as the tile kernel's 1x1 form: bit-identical results.
template <int CI, int CO, bool TRS>
__global__ __launch_bounds__(256) void conv1x1_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ up, float* __restrict__ y, float* __restrict__ ysq, long long npix, int H, int W, int relu) {
    __shared__ float red[4];
    __shared__ __attribute__((aligned(16))) float tr[TRS ? 256 * (CO + 4) : 4];
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool live = p < npix;
    float acc[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) acc[co] = 0.0f;
    float m = 0.0f;
    if (live) {
        f4v xv[CI / 4];
#pragma unroll
        for (int c = 0; c < CI / 4; ++c) xv[c] = *reinterpret_cast<const f4v*>(x + p * CI + c * 4);
#pragma unroll 2
        for (int c = 0; c < CI / 4; ++c) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* wt = wp + (c * 4 + j) * CO;
#pragma unroll
                for (int co = 0; co < CO; ++co) acc[co] = fmaf(xv[c][j], wt[co], acc[co]);
            }
        }
        if (scale) {
#pragma unroll
            for (int co = 0; co < CO; ++co) acc[co] = acc[co] * scale[co];
        }
        if (shift) {
#pragma unroll
            for (int co = 0; co < CO; ++co) acc[co] = acc[co] + shift[co];
        }
        if (up) {                                                       // F.interpolate(intra) + inner(conv): pixel (n, oy, ox) <- up[n, oy / 2, ox / 2]
            const long long hw = (long long)H * W;
            const long long n = p / hw;
            const int r = (int)(p - n * hw), oy = r / W, ox = r - oy * W;
            const float* upp = up + ((n * (H / 2) + oy / 2) * (W / 2) + ox / 2) * CO;
#pragma unroll
            for (int co = 0; co < CO; co += 4) {
                const f4v u4 = *reinterpret_cast<const f4v*>(upp + co);
                acc[co] = u4.x + acc[co]; acc[co + 1] = u4.y + acc[co + 1]; acc[co + 2] = u4.z + acc[co + 2]; acc[co + 3] = u4.w + acc[co + 3];
            }
        }
        if (relu) {
#pragma unroll
            for (int co = 0; co < CO; ++co) acc[co] = fmaxf(acc[co], 0.0f);
        }
        if (ysq) {
#pragma unroll
            for (int co = 0; co < CO; ++co) m = fmaxf(m, fabsf(acc[co]));
        }
    }
    // store through a wave-private LDS transpose: a lane holds its pixel's CO channels (CO * 4 contiguous bytes), so storing from the
    // registers would put 64 scattered 16-byte pieces into every store instruction (measured: 21 us for the 16 -> 32 layer, no faster than
    // the tile kernel); after the transpose instruction i of a wave writes float4 i * 64 + lane of the wave's 64 x CO block: 1 KiB contiguous.
    // (TRS = false: plain per-lane stores, for maps so small that the launch is latency-bound and the LDS round trip only adds to it:
    // the 32 -> 32 output conv of stage 1 at 3 x 128 x 160, 11.8 against 13.3 us)
    if constexpr (!TRS) {
        if (live) {
            float* yp = y + p * CO;
#pragma unroll
            for (int co = 0; co < CO; co += 4) *reinterpret_cast<float4*>(yp + co) = make_float4(acc[co], acc[co + 1], acc[co + 2], acc[co + 3]);
        }
    } else {
        constexpr int STR = CO + 4;                                     // floats per pixel row in LDS (16-byte aligned, bank-staggered)
        float* tw = tr + (threadIdx.x >> 6) * 64 * STR;
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int co = 0; co < CO; co += 4) *reinterpret_cast<f4v*>(tw + lane * STR + co) = (f4v){acc[co], acc[co + 1], acc[co + 2], acc[co + 3]};
        __builtin_amdgcn_wave_barrier();
        const long long wbase = (long long)blockIdx.x * 256 + (threadIdx.x & ~63);      // first pixel of this wave
#pragma unroll
        for (int i = 0; i < CO / 4; ++i) {
            const int e = i * 64 + lane, px = e / (CO / 4), q = e % (CO / 4);
            const f4v v = *reinterpret_cast<const f4v*>(tw + px * STR + q * 4);
            if (wbase + px < npix) *reinterpret_cast<f4v*>(y + (wbase + px) * CO + q * 4) = v;
        }
    }
    if (ysq) {
#pragma unroll
        for (int k = 32; k > 0; k >>= 1) m = fmaxf(m, __shfl_xor(m, k));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            atomicMax(reinterpret_cast<unsigned int*>(ysq) + (blockIdx.x & 63) * 16, __float_as_uint(m * m));
        }
    }
}

template <int CI, int CO>
static int conv1x1_launch_t(const float* x, const float* wp, const float* scale, const float* shift, const float* up, float* y, float* ysq,
                            int N, int H, int W, int relu, hipStream_t st) {
    const long long npix = (long long)N * H * W;
    if (npix >= 131072)
        hipLaunchKernelGGL((conv1x1_kernel<CI, CO, true>), dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, x, wp, scale, shift, up, y, ysq, npix, H, W, relu);
    else
        hipLaunchKernelGGL((conv1x1_kernel<CI, CO, false>), dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, x, wp, scale, shift, up, y, ysq, npix, H, W, relu);
    return launch_status("conv1x1");
}

// (Co,Ci,K,K) -> [K*K][Cip][Co], input channels zero-padded to Cip
__global__ void pack_weight2d_kernel(const float* __restrict__ w, float* __restrict__ packed, int Co, int Ci, int Cip, int KK) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= KK * Cip * Co) return;
    int co = t % Co, ci = (t / Co) % Cip, tap = t / (Co * Cip);
    packed[t] = (ci < Ci) ? w[((long long)co * Ci + ci) * KK + tap] : 0.0f;
}

// NCHW (3 channels) -> channels-last padded to 4
__global__ __launch_bounds__(256) void rgb_to_nhwc4_kernel(const float* __restrict__ x, float* __restrict__ y, long long HW) {
    const int n = blockIdx.y;
    long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const float* xb = x + (long long)n * 3 * HW;
    *reinterpret_cast<float4*>(y + ((long long)n * HW + p) * 4) = make_float4(xb[p], xb[HW + p], xb[2 * HW + p], 0.0f);
}

template <int CI, int CO, int K, int S, int TH, int TW, int PPT, int UKY, int UKX, bool RGB3 = false>
static int conv2d_launch_t(const float* x, const float* wp, const float* scale, const float* shift, const float* up, float* y,
                           int N, int H, int W, int relu, hipStream_t st) {
    constexpr int PAD = K / 2;
    const int Ho = (H + 2 * PAD - K) / S + 1, Wo = (W + 2 * PAD - K) / S + 1;
    const int tiles_w = (Wo + TW - 1) / TW, tiles_h = (Ho + TH * PPT - 1) / (TH * PPT);
    constexpr int HH = (TH * PPT - 1) * S + K, HW = (TW - 1) * S + K;
    const size_t lds = (size_t)HH * HW * C2_STRIDE * sizeof(float);
    hipLaunchKernelGGL((conv2d_lds_kernel<CI, CO, K, S, TH, TW, PPT, UKY, UKX, RGB3>), dim3(tiles_w * tiles_h, N), dim3(256), lds, st, x, wp, scale,
                       shift, up, y, H, W, Ho, Wo, tiles_w, relu);
    return launch_status("conv2d");
}

// ---- 1x1 layers on the matrix cores (round 4), exact: three bf16 pieces per operand by truncation, six v_mfma_f32_16x16x32_bf16 per
// product (the arithmetic of conv3d_x3.hip, NP = 3).  A 1x1 conv is the one layer whose B fragment needs no staging: lane (n, kq) of a
// wave owns pixel 16 t + n of n-tile t and loads channels 8 kq .. 8 kq + 7 of it (4 kq .. 4 kq + 3 with Ci = 16) -- exactly its fragment of v_mfma (column n).
// M = the 32 output channels (two m-tiles, weight fragments built once per wave from the fp32 [Ci][Co] image and kept in registers),
// K = Ci (one k-step: v_mfma_f32_16x16x32_bf16 for Ci = 32, v_mfma_f32_16x16x16_bf16 -- four channels per lane -- for Ci = 16), 12 MFMAs per 16 pixels; the output fragment of a lane is four channels of
// its pixel: float4 stores, and the epilogue (BatchNorm / bias, nearest x2 up-add, ReLU, squared bound) works on that float4.
// No LDS, no barrier; a wave walks n-tiles with the next tile's loads in flight.
template <int CI>
__global__ __launch_bounds__(256) void conv1x1_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ up, float* __restrict__ y, float* __restrict__ ysq, long long npix, int H, int W, int relu) {
    constexpr int CO = 32;
    constexpr int KL = CI / 4;                                           // input channels per lane: 8 (K = 32: v_mfma_f32_16x16x32_bf16) or 4 (K = 16: v_mfma_f32_16x16x16_bf16)
    static_assert(CI == 32 || CI == 16, "one k-step: Ci = 32 or 16");
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef unsigned int frag_t __attribute__((ext_vector_type(KL / 2)));        // KL bf16
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, kq = lane >> 4;
    // weight fragments: row m = co % 16, k = KL kq + i = input channel
    frag_t A[2][3];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        unsigned hb[KL], mb[KL], lb[KL];
#pragma unroll
        for (int i = 0; i < KL; ++i) {
            const float v = wp[(kq * KL + i) * CO + mt * 16 + n];
            hb[i] = __float_as_uint(v) & 0xffff0000u;
            const float r1 = v - __uint_as_float(hb[i]);
            mb[i] = __float_as_uint(r1) & 0xffff0000u;
            const float r2 = r1 - __uint_as_float(mb[i]);
            lb[i] = __float_as_uint(r2) & 0xffff0000u;
        }
#pragma unroll
        for (int j = 0; j < KL / 2; ++j) {
            A[mt][0][j] = (hb[2 * j] >> 16) | hb[2 * j + 1];
            A[mt][1][j] = (mb[2 * j] >> 16) | mb[2 * j + 1];
            A[mt][2][j] = (lb[2 * j] >> 16) | lb[2 * j + 1];
        }
    }
    auto mfma = [](frag_t a, frag_t b, x3_f32x4 c) -> x3_f32x4 {
        if constexpr (KL == 8) return x3_mfma<3>(a, b, c);
        else return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
    };
    const long long ntiles = (npix + 15) / 16, stride = (long long)gridDim.x * 4;
    x3_f32x4 xq[KL / 4];
    auto fetch = [&](long long tile) {
        const long long p = tile * 16 + n;
        const bool in = tile < ntiles && p < npix;
#pragma unroll
        for (int c = 0; c < KL / 4; ++c) xq[c] = in ? *reinterpret_cast<const x3_f32x4*>(x + p * CI + kq * KL + c * 4) : (x3_f32x4){0.f, 0.f, 0.f, 0.f};
    };
    float vmax = 0.0f;
    long long tile = (long long)blockIdx.x * 4 + wave;
    fetch(tile);
    const long long hw = (long long)H * W;
    for (; tile < ntiles; tile += stride) {
        frag_t bq[3];
        {
            x3_u32x2 h0, m0, l0;
            x3_split4(xq[0], h0, m0, l0);
            if constexpr (KL == 8) {
                x3_u32x2 h1, m1, l1;
                x3_split4(xq[KL / 4 - 1], h1, m1, l1);
                bq[0] = (frag_t){h0.x, h0.y, h1.x, h1.y}; bq[1] = (frag_t){m0.x, m0.y, m1.x, m1.y}; bq[2] = (frag_t){l0.x, l0.y, l1.x, l1.y};
            } else { bq[0] = (frag_t){h0.x, h0.y}; bq[1] = (frag_t){m0.x, m0.y}; bq[2] = (frag_t){l0.x, l0.y}; }
        }
        fetch(tile + stride);
        const long long p = tile * 16 + n;
        const bool live = p < npix;
        long long upo = 0;
        if (up && live) {                                                // F.interpolate(intra) + inner(conv): pixel (b, oy, ox) <- up[b, oy / 2, ox / 2]
            const long long b = p / hw;
            const int r = (int)(p - b * hw), oy = r / W, ox = r - oy * W;
            upo = ((b * (H / 2) + oy / 2) * (W / 2) + ox / 2) * CO;
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            x3_f32x4 acc[3] = {(x3_f32x4){0.f, 0.f, 0.f, 0.f}, (x3_f32x4){0.f, 0.f, 0.f, 0.f}, (x3_f32x4){0.f, 0.f, 0.f, 0.f}};
            acc[2] = mfma(A[mt][0], bq[2], acc[2]);
            acc[1] = mfma(A[mt][0], bq[1], acc[1]);
            acc[0] = mfma(A[mt][0], bq[0], acc[0]);
            acc[2] = mfma(A[mt][1], bq[1], acc[2]);
            acc[1] = mfma(A[mt][1], bq[0], acc[1]);
            acc[2] = mfma(A[mt][2], bq[0], acc[2]);
            const int co = mt * 16 + kq * 4;                             // this lane's four output channels of pixel p
            x3_f32x4 v = acc[0] + (acc[1] + acc[2]);
            if (scale) v = v * *reinterpret_cast<const x3_f32x4*>(scale + co);
            if (shift) v = v + *reinterpret_cast<const x3_f32x4*>(shift + co);
            if (live) {
                if (up) v = *reinterpret_cast<const x3_f32x4*>(up + upo + co) + v;
                if (relu) v = __builtin_elementwise_max(v, (x3_f32x4){0.f, 0.f, 0.f, 0.f});
                *reinterpret_cast<x3_f32x4*>(y + p * CO + co) = v;
                vmax = x3_absmax4(vmax, v);
            }
        }
    }
    if (ysq) {
#pragma unroll
        for (int k = 32; k > 0; k >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, k));
        if (lane == 0) red[wave] = vmax;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            atomicMax(reinterpret_cast<unsigned int*>(ysq) + (blockIdx.x & 63) * 16, __float_as_uint(m * m));
        }
    }
}

template <int CI>
static int conv1x1_mfma_launch_t(const float* x, const float* wp, const float* scale, const float* shift, const float* up, float* y, float* ysq,
                                 int N, int H, int W, int relu, hipStream_t st) {
    const long long npix = (long long)N * H * W, ntiles = (npix + 15) / 16;
    const long long blocks = (ntiles + 3) / 4 < 2048 ? (ntiles + 3) / 4 : 2048;          // (a wave walks ~2-4 n-tiles on the FeatureNet maps)
    hipLaunchKernelGGL(conv1x1_mfma_kernel<CI>, dim3((unsigned)blocks), dim3(256), 0, st, x, wp, scale, shift, up, y, ysq, npix, H, W, relu);
    return launch_status("conv1x1_mfma");
}

}  // namespace rcmvs

using namespace rcmvs;

extern "C" {

int rcmvs_rgb_to_nhwc4(const float* x, float* y, int N, int H, int W, void* stream) {
    RCMVS_REQUIRE(x && y && N > 0 && H > 0 && W > 0, "rgb_to_nhwc4: bad arguments");
    long long HW = (long long)H * W;
    hipLaunchKernelGGL(rgb_to_nhwc4_kernel, dim3((unsigned)cdiv(HW, 256), N), dim3(256), 0, as_stream(stream), x, y, HW);
    return launch_status("rgb_to_nhwc4");
}

int rcmvs_pack_conv2d_weight(const float* w, float* packed, int Co, int Ci, int Cip, int K, void* stream) {
    RCMVS_REQUIRE(w && packed && Co > 0 && Ci > 0 && Cip >= Ci && K > 0, "pack_conv2d_weight: bad arguments");
    int n = K * K * Cip * Co;
    hipLaunchKernelGGL(pack_weight2d_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), w, packed, Co, Ci, Cip, K * K);
    return launch_status("pack_conv2d_weight");
}

int rcmvs_conv1x1_fwd(const float* x, const float* w_packed, const float* scale, const float* shift, const float* up_add,
                      float* y, float* ysq_absmax, int N, int H, int W, int Ci, int Co, int relu, void* stream) {
    RCMVS_REQUIRE(x && w_packed && y, "conv1x1_fwd: null pointer");
    RCMVS_REQUIRE(N > 0 && H > 0 && W > 0, "conv1x1_fwd: bad sizes");
    RCMVS_REQUIRE(!up_add || (H % 2 == 0 && W % 2 == 0), "conv1x1_fwd: the up-add merge needs even H and W (got %d x %d)", H, W);
    hipStream_t st = as_stream(stream);
    if (Ci == 16 && Co == 32) return conv1x1_launch_t<16, 32>(x, w_packed, scale, shift, up_add, y, ysq_absmax, N, H, W, relu, st);
    if (Ci == 32 && Co == 32) return conv1x1_launch_t<32, 32>(x, w_packed, scale, shift, up_add, y, ysq_absmax, N, H, W, relu, st);
    if (Ci == 8 && Co == 32) return conv1x1_launch_t<8, 32>(x, w_packed, scale, shift, up_add, y, ysq_absmax, N, H, W, relu, st);
    return fail(-1, "conv1x1_fwd: unsupported layer Ci=%d Co=%d (16 -> 32, 32 -> 32, 8 -> 32)", Ci, Co);
}

int rcmvs_conv1x1_mfma_fwd(const float* x, const float* w_packed, const float* scale, const float* shift, const float* up_add,
                           float* y, float* ysq_absmax, int N, int H, int W, int Ci, int Co, int relu, void* stream) {
    RCMVS_REQUIRE(x && w_packed && y, "conv1x1_mfma_fwd: null pointer");
    RCMVS_REQUIRE(N > 0 && H > 0 && W > 0, "conv1x1_mfma_fwd: bad sizes");
    RCMVS_REQUIRE(!up_add || (H % 2 == 0 && W % 2 == 0), "conv1x1_mfma_fwd: the up-add merge needs even H and W (got %d x %d)", H, W);
    hipStream_t st = as_stream(stream);
    if (Ci == 16 && Co == 32) return conv1x1_mfma_launch_t<16>(x, w_packed, scale, shift, up_add, y, ysq_absmax, N, H, W, relu, st);
    if (Ci == 32 && Co == 32) return conv1x1_mfma_launch_t<32>(x, w_packed, scale, shift, up_add, y, ysq_absmax, N, H, W, relu, st);
    return fail(-1, "conv1x1_mfma_fwd: unsupported layer Ci=%d Co=%d (16 -> 32, 32 -> 32)", Ci, Co);
}

int rcmvs_conv2d_fwd(const float* x, const float* w_packed, const float* scale, const float* shift, const float* up_add,
                     float* y, int N, int H, int W, int Ci, int Co, int K, int stride, int relu, void* stream) {
    RCMVS_REQUIRE(x && w_packed && y, "conv2d_fwd: null pointer");
    RCMVS_REQUIRE(N > 0 && H > 0 && W > 0, "conv2d_fwd: bad sizes");
    hipStream_t st = as_stream(stream);
    if (K == 1 && stride == 1 && Co == 32 && (Ci == 8 || Ci == 16 || Ci == 32) && (!up_add || (H % 2 == 0 && W % 2 == 0)))
        return rcmvs_conv1x1_fwd(x, w_packed, scale, shift, up_add, y, nullptr, N, H, W, Ci, Co, relu, stream);      // the streaming 1x1 kernel (bit-identical to the tile kernel's 1x1 form)
    // Ci == 3: the first layer on the planar (N, 3, H, W) input itself; w_packed is the layer's weight packed with Cip = 4
    if (Ci == 3 && Co == 8 && K == 3 && stride == 1)
        return conv2d_launch_t<4, 8, 3, 1, 16, 16, 1, 1, 3, true>(x, w_packed, scale, shift, up_add, y, N, H, W, relu, st);
#define RCMVS_C2(CI, CO, KK, SS, TH, TW, PPT, UY, UX)                                                             \
    if (Ci == CI && Co == CO && K == KK && stride == SS)                                                          \
        return conv2d_launch_t<CI, CO, KK, SS, TH, TW, PPT, UY, UX>(x, w_packed, scale, shift, up_add, y, N, H, W, relu, st);
    // the 13 layers of FeatureNet(base_channels=8, fpn, 3 stages): tile, pixels per thread and tap-loop unrolling per layer
    // (us per 3-view scene at 512x640 in the comments, profiles/r1_run7_kernel_stats.csv)
    RCMVS_C2(4, 8, 3, 1, 16, 16, 1, 1, 3)     /* conv0.0  18 */  RCMVS_C2(8, 8, 3, 1, 16, 16, 1, 1, 1)    /* conv0.1  24 */
    RCMVS_C2(8, 16, 5, 2, 8, 32, 1, 1, 1)     /* conv1.0  45 */  RCMVS_C2(16, 16, 3, 1, 16, 16, 2, 1, 1)  /* conv1.1/2  2 x 30 */
    RCMVS_C2(16, 32, 5, 2, 8, 32, 1, 5, 5)    /* conv2.0  98 */  RCMVS_C2(32, 32, 3, 1, 16, 16, 1, 3, 3)  /* conv2.1/2  2 x 76 */
    RCMVS_C2(32, 32, 1, 1, 16, 16, 1, 1, 1)   /* out1     14 */  RCMVS_C2(16, 32, 1, 1, 16, 16, 2, 1, 1)  /* inner1   23 */
    RCMVS_C2(32, 16, 3, 1, 16, 16, 1, 3, 3)   /* out2     47 */  RCMVS_C2(8, 32, 1, 1, 16, 16, 2, 1, 1)   /* inner2   59 */
    RCMVS_C2(32, 8, 3, 1, 16, 16, 1, 1, 1)    /* out3     91 */
#undef RCMVS_C2
    return fail(-1, "conv2d_fwd: unsupported layer Ci=%d Co=%d K=%d stride=%d", Ci, Co, K, stride);
}

}  // extern "C"
