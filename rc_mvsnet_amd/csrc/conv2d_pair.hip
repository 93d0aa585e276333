// Two consecutive 3x3 stride-1 Conv2d blocks of FeatureNet (conv + BatchNorm(eval) + ReLU, twice: conv1.1 -> conv1.2, 16 -> 16 -> 16 at half resolution;
// models/modules.py:372-379,413-424) in ONE launch (round 6).  As two launches of the planar split-bf16 kernel they cost 16.5 us each for 15.7 MB in and
// 15.7 MB out: tiny layers whose time is the launch, the first round trip and the tick structure, not traffic or arithmetic.  Here a block owns an
// 8 x 30 pixel tile of one view: the 12 x 34 input halo is loaded once and split EXACTLY into three bf16 pieces in LDS (x = h + m + l by truncation,
// six v_mfma_f32_16x16x32_bf16 per product, three magnitude classes in separate accumulators: the arithmetic of conv3d_x3.hip, fp32-exact, no bound
// needed), the first layer is evaluated on the 10 x 32 tile the second one needs (zero outside the image: the second layer's zero padding), its
// output goes to LDS as three pieces again, the second layer reads it from there and stores the tile.  The intermediate map never reaches memory.
// GEMM per wave: D[16 x 16] += A[16 x 32] B[32 x 16], M = output channel, N = 16 pixels of a row, K step = two taps x 16 channels (five steps, the
// tenth tap slot is zero); a layer's fragments (5 x 3 pieces) are register-stationary while that layer runs.  gfx950 only.
// What bounds it: the tile's latency chain (halo round trip -> split -> first layer -> LDS -> second layer -> store) with two blocks per CU (72 KB of
// LDS each) and 1 024 tiles in two rounds -- not traffic (31 MB) and not the 286 K MFMAs per view.
#include "common.h"
#include "x3_pieces.h"
#include <atomic>

namespace rcmvs {

constexpr int P2_TH = 8, P2_TW = 30;                            // output tile
constexpr int P2_MH = P2_TH + 2, P2_MW = 34;                    // intermediate tile: 10 rows x 32 columns computed, 34 stored (the last two stay zero: B fragments of the
                                                                // second n-tile read them under taps that fall outside the 30 output columns' needs)
constexpr int P2_IH = P2_TH + 4, P2_IW = 34;                    // input halo
constexpr int P2_C = 16, P2_PB = P2_C * 2;                      // bytes per pixel per piece plane
constexpr int P2_IPL = P2_IH * P2_IW * P2_PB, P2_MPL = P2_MH * P2_MW * P2_PB;
constexpr int P2_LDS = 3 * P2_IPL + 3 * P2_MPL;
constexpr int P2_KS = 5;                                        // K steps per layer
constexpr int P2_NLD = (P2_IH * P2_IW * (P2_C / 4) + 255) / 256;
constexpr long long P2_IMG_HALFS = 2LL * P2_KS * 3 * 64 * 8;    // two layers x [K step][piece][lane][8 bf16]

long long conv2d_pair_weight_floats() { return P2_IMG_HALFS / 2; }

// wa, wb: Conv2d weights (16, 16, 3, 3) of the two layers -> A fragments (row = lane & 15 = output channel, k = 8 (lane >> 4) + e: tap slot kk >> 1 of the
// step, channel 8 (kk & 1) + e), three bf16 pieces by truncation
__global__ void conv2d_pair_pack_kernel(const float* __restrict__ wa, const float* __restrict__ wb, unsigned short* __restrict__ img) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * P2_KS * 64 * 8) return;
    const int e = t & 7, lane = (t >> 3) & 63, j = (t >> 9) % P2_KS, layer = (t >> 9) / P2_KS;
    const int co = lane & 15, kk = lane >> 4;
    const int tap = 2 * j + (kk >> 1), ci = (kk & 1) * 8 + e;
    const float* w = layer ? wb : wa;
    const float v = tap < 9 ? w[(co * P2_C + ci) * 9 + tap] : 0.0f;
    const unsigned hb = __float_as_uint(v) & 0xffff0000u;
    const float r1 = v - __uint_as_float(hb);
    const unsigned mb = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(mb);
    const unsigned lb = __float_as_uint(r2) & 0xffff0000u;
    const long long base = (((long long)layer * P2_KS + j) * 3) * 512 + lane * 8 + e;
    img[base] = (unsigned short)(hb >> 16);
    img[base + 512] = (unsigned short)(mb >> 16);
    img[base + 1024] = (unsigned short)(lb >> 16);
}

__global__ __launch_bounds__(256, 2) void conv2d_pair_kernel(
    const float* __restrict__ x, const x3_u32x4* __restrict__ wimg, const float* __restrict__ sa, const float* __restrict__ ha,
    const float* __restrict__ sb, const float* __restrict__ hb, float* __restrict__ y, int H, int W, int tiles_w, int tiles) {
    extern __shared__ __attribute__((aligned(16))) x3_byte smem[];
    x3_byte* const ib = smem;                                   // input halo: three piece planes
    x3_byte* const mb = smem + 3 * P2_IPL;                      // intermediate tile: three piece planes
    constexpr int OOB = 0x7ffffff0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kk = lane >> 4;
    const int view = blockIdx.y;
    const unsigned t2 = xcd_remap(blockIdx.x, gridDim.x);
    const int tw = t2 % tiles_w, th = t2 / tiles_w;
    const int h0 = th * P2_TH, w0 = tw * P2_TW;
    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (long long)view * H * W * P2_C), (short)0, H * W * P2_C * 4, 0x00020000);
    // ---- input halo: rows h0 - 2 .. h0 + 9, columns w0 - 2 .. w0 + 31, requested first
    x3_u32x4 pf[P2_NLD];
    int ls[P2_NLD];
#pragma unroll
    for (int i = 0; i < P2_NLD; ++i) {
        const int e = tid + i * 256;
        const int pix = e >> 2, c4 = e & 3;
        const int r = pix / P2_IW, c = pix - r * P2_IW;
        const int ih = h0 - 2 + r, iw = w0 - 2 + c;
        const bool has = e < P2_IH * P2_IW * 4;
        pf[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (has && ih >= 0 && ih < H && iw >= 0 && iw < W) ? ((ih * W + iw) * P2_C + c4 * 4) * 4 : OOB, 0, 0);
        ls[i] = has ? pix * P2_PB + c4 * 8 : -1;
    }
    // ---- first layer's fragments
    x3_u32x4 A[P2_KS][3];
#pragma unroll
    for (int j = 0; j < P2_KS; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p) A[j][p] = wimg[(j * 3 + p) * 64 + lane];
    // per-lane tap offsets of the K steps: lane (n, kk) supplies channels 8 (kk & 1).. of the pixel shifted by tap 2 j + (kk >> 1)
    int toff_i[P2_KS], toff_m[P2_KS];
#pragma unroll
    for (int j = 0; j < P2_KS; ++j) {
        const int tap = min(2 * j + (kk >> 1), 8), dy = tap / 3, dx = tap - 3 * dy;
        toff_i[j] = (dy * P2_IW + dx) * P2_PB + (kk & 1) * 16;
        toff_m[j] = (dy * P2_MW + dx) * P2_PB + (kk & 1) * 16;
    }
    const int g4 = kk;                                          // D fragment: rows (= output channels) 4 g4 .. 4 g4 + 3 of pixel n
    const x3_f32x4 sca = *reinterpret_cast<const x3_f32x4*>(sa + 4 * g4), sha = *reinterpret_cast<const x3_f32x4*>(ha + 4 * g4);
    const x3_f32x4 scb = *reinterpret_cast<const x3_f32x4*>(sb + 4 * g4), shb = *reinterpret_cast<const x3_f32x4*>(hb + 4 * g4);
    // the two stored-but-not-computed columns of the intermediate tile are zero
    if (tid < P2_MH * 2 * 3) {
        const int p = tid / (P2_MH * 2), r = (tid / 2) % P2_MH, c = 32 + (tid & 1);
        x3_u32x4* q = reinterpret_cast<x3_u32x4*>(mb + p * P2_MPL + (r * P2_MW + c) * P2_PB);
        q[0] = (x3_u32x4){0u, 0u, 0u, 0u};
        q[1] = (x3_u32x4){0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < P2_NLD; ++i) {
        x3_u32x2 h, m, l;
        x3_split4(__builtin_bit_cast(x3_f32x4, pf[i]), h, m, l);
        if (ls[i] >= 0) {
            *reinterpret_cast<x3_u32x2*>(ib + ls[i]) = h;
            *reinterpret_cast<x3_u32x2*>(ib + P2_IPL + ls[i]) = m;
            *reinterpret_cast<x3_u32x2*>(ib + 2 * P2_IPL + ls[i]) = l;
        }
    }
    __syncthreads();
    // two n-tiles at a time (16 pixels of a row each, from byte offsets base[t] of their first pixel under tap (0, 0)): six MFMAs per K step and tile, ordered
    // so that an accumulator meets its next MFMA four issues later (a lone n-tile chains three dependent MFMAs per step: 28.1 us per launch on the three
    // half-resolution maps of a DTU scene against 26.1 with two tiles; the two launches it replaces: 33.0 -- profiles/r6_conv2d_pair.txt)
    auto gemm2 = [&](const x3_byte* src, int plane, const int (&base)[2], const int (&toff)[P2_KS], x3_f32x4 (&out)[2]) {
        x3_f32x4 a0[2], a1[2], a2[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) a0[t] = a1[t] = a2[t] = (x3_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < P2_KS; ++j) {
            x3_u32x4 bh[2], bm[2], bl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const x3_byte* q = src + base[t] + toff[j];
                bh[t] = *reinterpret_cast<const x3_u32x4*>(q); bm[t] = *reinterpret_cast<const x3_u32x4*>(q + plane); bl[t] = *reinterpret_cast<const x3_u32x4*>(q + 2 * plane);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) a0[t] = x3_mfma<3>(A[j][0], bh[t], a0[t]);
#pragma unroll
            for (int t = 0; t < 2; ++t) a1[t] = x3_mfma<3>(A[j][0], bm[t], a1[t]);
#pragma unroll
            for (int t = 0; t < 2; ++t) a2[t] = x3_mfma<3>(A[j][0], bl[t], a2[t]);
#pragma unroll
            for (int t = 0; t < 2; ++t) a1[t] = x3_mfma<3>(A[j][1], bh[t], a1[t]);
#pragma unroll
            for (int t = 0; t < 2; ++t) a2[t] = x3_mfma<3>(A[j][2], bh[t], a2[t]);
#pragma unroll
            for (int t = 0; t < 2; ++t) a2[t] = x3_mfma<3>(A[j][1], bm[t], a2[t]);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) out[t] = a0[t] + (a1[t] + a2[t]);
    };
    // ---- first layer on the 10 x 32 intermediate tile (rows h0 - 1 .., columns w0 - 1 ..): 20 n-tiles = 10 rows x 2, a wave takes rows wave, wave + 4, wave + 8
    // (both n-tiles of a row together: waves 0 / 1 three rows, waves 2 / 3 two)
#pragma unroll 1
    for (int r = wave; r < P2_MH; r += 4) {
        const int base[2] = {(r * P2_IW + n) * P2_PB, (r * P2_IW + 16 + n) * P2_PB};
        x3_f32x4 acc[2];
        gemm2(ib, P2_IPL, base, toff_i, acc);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int c0 = 16 * t;
            x3_f32x4 v = acc[t] * sca + sha;
            v = __builtin_elementwise_max(v, (x3_f32x4){0.f, 0.f, 0.f, 0.f});
            const int oh = h0 - 1 + r, ow = w0 - 1 + c0 + n;
            if (oh < 0 || oh >= H || ow < 0 || ow >= W) v = (x3_f32x4){0.f, 0.f, 0.f, 0.f};      // the second layer's zero padding
            x3_u32x2 h, m, l;
            x3_split4(v, h, m, l);
            const int o = (r * P2_MW + c0 + n) * P2_PB + g4 * 8;
            *reinterpret_cast<x3_u32x2*>(mb + o) = h;
            *reinterpret_cast<x3_u32x2*>(mb + P2_MPL + o) = m;
            *reinterpret_cast<x3_u32x2*>(mb + 2 * P2_MPL + o) = l;
        }
    }
    // ---- second layer's fragments over the first one's
#pragma unroll
    for (int j = 0; j < P2_KS; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p) A[j][p] = wimg[((P2_KS + j) * 3 + p) * 64 + lane];
    __syncthreads();
    // ---- second layer on the 8 x 30 output tile: rows wave, wave + 4 (the second n-tile of a row holds 14 pixels)
    float* yb = y + (long long)view * H * W * P2_C;
#pragma unroll 1
    for (int r = wave; r < P2_TH; r += 4) {
        const int base[2] = {(r * P2_MW + n) * P2_PB, (r * P2_MW + 16 + n) * P2_PB};
        x3_f32x4 acc[2];
        gemm2(mb, P2_MPL, base, toff_m, acc);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int c0 = 16 * t;
            x3_f32x4 v = acc[t] * scb + shb;
            v = __builtin_elementwise_max(v, (x3_f32x4){0.f, 0.f, 0.f, 0.f});
            const int oh = h0 + r, ow = w0 + c0 + n;
            if (c0 + n < P2_TW && oh < H && ow < W) *reinterpret_cast<x3_f32x4*>(yb + ((long long)oh * W + ow) * P2_C + 4 * g4) = v;
        }
    }
}

// x (N, H, W, 16) -> y (N, H, W, 16) = relu(bn_b(conv_b(relu(bn_a(conv_a(x))))));  wimg: conv2d_pair_pack's image
int conv2d_pair_launch(const float* x, const float* wimg, const float* sa, const float* ha, const float* sb, const float* hb, float* y,
                       int N, int H, int W, hipStream_t st) {
    if ((long long)H * W * P2_C * 4 >= 0x7ffffff0LL || N > 65535) return fail(-1, "conv2d_pair: map too large for 32-bit offsets");
    static std::atomic<bool> raised[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fail(-1, "conv2d_pair: cannot query the device");
    if (!raised[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)conv2d_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, P2_LDS) != hipSuccess)
            return fail(-1, "conv2d_pair: cannot raise the dynamic LDS limit to %d bytes", P2_LDS);
        raised[dev].store(true, std::memory_order_release);
    }
    const int tw_ = (W + P2_TW - 1) / P2_TW, th_ = (H + P2_TH - 1) / P2_TH;
    hipLaunchKernelGGL(conv2d_pair_kernel, dim3(tw_ * th_, N), dim3(256), P2_LDS, st, x, reinterpret_cast<const x3_u32x4*>(wimg), sa, ha, sb, hb, y, H, W, tw_, tw_ * th_);
    return launch_status("conv2d_pair");
}

}  // namespace rcmvs

using namespace rcmvs;

extern "C" {

long long rcmvs_conv2d_pair_weight_floats(void) { return conv2d_pair_weight_floats(); }

int rcmvs_pack_conv2d_pair(const float* wa, const float* wb, float* image, void* stream) {
    RCMVS_REQUIRE(wa && wb && image, "pack_conv2d_pair: null pointer");
    hipLaunchKernelGGL(conv2d_pair_pack_kernel, dim3((2 * P2_KS * 64 * 8 + 255) / 256), dim3(256), 0, as_stream(stream), wa, wb, reinterpret_cast<unsigned short*>(image));
    return launch_status("pack_conv2d_pair");
}

int rcmvs_conv2d_pair_fwd(const float* x, const float* image, const float* scale_a, const float* shift_a, const float* scale_b, const float* shift_b,
                          float* y, int N, int H, int W, int C, void* stream) {
    RCMVS_REQUIRE(x && image && scale_a && shift_a && scale_b && shift_b && y, "conv2d_pair_fwd: null pointer");
    RCMVS_REQUIRE(N > 0 && H > 0 && W > 0, "conv2d_pair_fwd: bad sizes");
    RCMVS_REQUIRE(C == P2_C, "conv2d_pair_fwd: built for 16 channels (got %d)", C);
    return conv2d_pair_launch(x, image, scale_a, shift_a, scale_b, shift_b, y, N, H, W, as_stream(stream));
}

}  // extern "C"
