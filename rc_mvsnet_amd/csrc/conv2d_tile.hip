// FeatureNet's 32-channel 3x3 layers as tile kernels (round 6; 32 -> 32 = conv2.1, conv2.2 at quarter resolution: eight waves, wave = (m-tile, row pair), below).
// The two 32 -> 16 layers at half resolution: conv1.0 -- the 5x5 stride-2 Conv2d(8, 16) + BatchNorm + ReLU as a 3x3
// layer on the space-to-depth view of the full-resolution map (models/modules.py:374-375,416-417) -- and the FPN output conv out2 (models/modules.py:437,452).
// On the planar split-bf16 kernel of conv3d_x3.hip (persistent blocks marching over "planes" = views, producer / consumer waves) each costs 23.6 us per DTU
// scene for 47 MB and 0.83 M MFMAs: the tick skeleton, not the work (profiles/r6_session2.txt).  Here a block owns an 8 x 32 pixel tile of one view: the 10 x 34
// input halo is loaded once and split EXACTLY into three bf16 pieces in LDS (x = h + m + l by truncation, six v_mfma_f32_16x16x32_bf16 per product, three
// magnitude classes in separate accumulators: the arithmetic of conv3d_x3.hip, fp32-exact, no bound needed), a wave owns two rows = four n-tiles of 16 pixels,
// M = 16 output channels, K step = one tap x 32 channels, the 27 weight fragments (9 x 3 pieces) register-stationary.  64-byte voxels: the bank swizzle of
// conv3d_x3.hip.  Epilogue: BatchNorm scale / shift and ReLU (both optional), and -- optionally -- the SQUARE of max|y| into a bound vector (the bound of the
// variance volume built from the map, csrc/absmax.hip).  gfx950 only.
#include "common.h"
#include "x3_pieces.h"
#include <atomic>

namespace rcmvs {

constexpr int T2_TH = 8, T2_TW = 32;
constexpr int T2_IH = T2_TH + 2, T2_IW = T2_TW + 2;
constexpr int T2_CI = 32, T2_PB = T2_CI * 2;                      // bytes per voxel per piece plane
constexpr int T2_IPL = T2_IH * T2_IW * T2_PB;
constexpr int T2_LDS = 3 * T2_IPL + 64;
constexpr int T2_KS = 9;
constexpr long long t2_img_halfs(int co) { return (long long)T2_KS * 3 * (co / 16) * 64 * 8; }     // [K step = tap][piece][m-tile][lane][8 bf16]
__device__ __forceinline__ int t2_swz(int c) { return ((c >> 2) & 1) * 32; }

long long conv2d_tile_weight_floats(int co) { return (co == 16 || co == 32) ? t2_img_halfs(co) / 2 : 0; }

// w: Conv2d weight (Co, 32, 3, 3), Co = 16 or 32 -> A fragments (row = lane & 15 = output channel of the m-tile, k = 8 (lane >> 4) + e = input channel), three bf16 pieces by truncation
__global__ void conv2d_tile_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ img, int MT) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T2_KS * MT * 64 * 8) return;
    const int e = t & 7, lane = (t >> 3) & 63, mt = (t >> 9) % MT, j = (t >> 9) / MT;
    const int co = mt * 16 + (lane & 15), ci = (lane >> 4) * 8 + e;
    const float v = w[(co * T2_CI + ci) * 9 + j];
    const unsigned hb = __float_as_uint(v) & 0xffff0000u;
    const float r1 = v - __uint_as_float(hb);
    const unsigned mb = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(mb);
    const unsigned lb = __float_as_uint(r2) & 0xffff0000u;
    const long long base = (((long long)j * 3) * MT + mt) * 512 + lane * 8 + e;       // piece stride = MT x 512
    img[base] = (unsigned short)(hb >> 16);
    img[base + MT * 512] = (unsigned short)(mb >> 16);
    img[base + 2 * MT * 512] = (unsigned short)(lb >> 16);
}

// S2D: x is physically (N, 2H, 2W, 8) and is read through the space-to-depth view (channel (py, px, c) of voxel (y, x) = channel c of pixel (2y + py, 2x + px))
// CO = 32: 512 threads, wave = (m-tile wave & 1, row pair wave >> 1) -- every B fragment is read once per m-tile --, one block per CU (240 tiles of a DTU scene's quarter-resolution maps: one round)
template <bool S2D, int CO>
__global__ __launch_bounds__(CO * 16, CO == 16 ? 2 : 1) void conv2d_tile_kernel(
    const float* __restrict__ x, const x3_u32x4* __restrict__ wimg, const float* __restrict__ scale, const float* __restrict__ shift,
    float* __restrict__ y, int H, int W, int tiles_w, int relu, float* __restrict__ ysq) {
    extern __shared__ __attribute__((aligned(16))) x3_byte smem[];
    x3_byte* const ib = smem;
    float* const redmax = reinterpret_cast<float*>(smem + 3 * T2_IPL);
    constexpr int OOB = 0x7ffffff0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kk = lane >> 4;
    constexpr int MT = CO / 16, NW = 4 * MT, NHALF = MT;          // m-tiles, waves, thread groups of 256 that share the halo rows
    const int mt = MT == 2 ? (wave & 1) : 0, rg = MT == 2 ? (wave >> 1) : wave;
    const int view = blockIdx.y;
    const unsigned t2 = xcd_remap(blockIdx.x, gridDim.x);
    const int tw = t2 % tiles_w, th = t2 / tiles_w;
    const int h0 = th * T2_TH, w0 = tw * T2_TW;
    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (long long)view * H * W * T2_CI), (short)0, H * W * T2_CI * 4, 0x00020000);
    // ---- input halo: rows h0 - 1 .. h0 + 8, columns w0 - 1 .. w0 + 32.  Thread (column tid / 8, float4 tid % 8) takes columns 0 .. 31 of every row, the first 160
    // threads the two remaining columns (a flat element index costs two divisions per load, and these kernels are bound by their vector instructions)
    auto offset_of = [&](int r, int c, int c4) -> int {
        const int ih = h0 - 1 + r, iw = w0 - 1 + c;
        if (!(ih >= 0 && ih < H && iw >= 0 && iw < W)) return OOB;
        if constexpr (S2D) return (((2 * ih + (c4 >> 2)) * (2 * W) + 2 * iw + ((c4 >> 1) & 1)) * 8 + (c4 & 1) * 4) * 4;
        else return ((ih * W + iw) * T2_CI + c4 * 4) * 4;
    };
    constexpr int NR = T2_IH / NHALF;                             // rows per thread: group g of 256 threads takes rows g, g + NHALF, ...
    x3_u32x4 pf[NR + 1];
    const int lc = (tid & 255) >> 3, l4 = tid & 7, r0 = tid >> 8;
#pragma unroll
    for (int r = 0; r < NR; ++r) pf[r] = __builtin_amdgcn_raw_buffer_load_b128(xrs, offset_of(r0 + r * NHALF, lc, l4), 0, 0);
    const int er = tid >> 4, ec = 32 + ((tid >> 3) & 1);          // (threads 0 .. 159: row tid / 16, column 32 + (tid / 8) % 2)
    pf[NR] = __builtin_amdgcn_raw_buffer_load_b128(xrs, tid < T2_IH * 16 ? offset_of(er, ec, l4) : OOB, 0, 0);
    // ---- fragments, epilogue constants
    x3_u32x4 A[T2_KS][3];
#pragma unroll
    for (int j = 0; j < T2_KS; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p) A[j][p] = wimg[((j * 3 + p) * MT + mt) * 64 + lane];
    const int co0 = mt * 16 + 4 * kk;
    const x3_f32x4 sc = scale ? *reinterpret_cast<const x3_f32x4*>(scale + co0) : (x3_f32x4){1.f, 1.f, 1.f, 1.f};
    const x3_f32x4 sh = shift ? *reinterpret_cast<const x3_f32x4*>(shift + co0) : (x3_f32x4){0.f, 0.f, 0.f, 0.f};
    auto park = [&](x3_u32x4 v, int r, int c) {
        x3_u32x2 h, m, l;
        x3_split4(__builtin_bit_cast(x3_f32x4, v), h, m, l);
        x3_byte* q = ib + (r * T2_IW + c) * T2_PB + ((l4 * 8) ^ t2_swz(c));
        *reinterpret_cast<x3_u32x2*>(q) = h;
        *reinterpret_cast<x3_u32x2*>(q + T2_IPL) = m;
        *reinterpret_cast<x3_u32x2*>(q + 2 * T2_IPL) = l;
    };
#pragma unroll
    for (int r = 0; r < NR; ++r) park(pf[r], r0 + r * NHALF, lc);
    if (tid < T2_IH * 16) park(pf[NR], er, ec);
    __syncthreads();
    // ---- rows 2 rg, 2 rg + 1 (rg = this wave's row pair): the two n-tiles of a row together (independent accumulator chains); lane (n, kk) supplies channels 8 kk .. of pixel (row + dy, column + dx)
    float* yb = y + (long long)view * H * W * CO;
    float vmax = 0.0f;
#pragma unroll 1
    for (int rr = 0; rr < 2; ++rr) {
        const int row = 2 * rg + rr;
        x3_f32x4 a0[2], a1[2], a2[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) a0[t] = a1[t] = a2[t] = (x3_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < T2_KS; ++j) {
            const int dy = j / 3, dx = j - 3 * dy;
            x3_u32x4 bh[2], bm[2], bl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int c = 16 * t + n + dx;
                const x3_byte* q = ib + ((row + dy) * T2_IW + c) * T2_PB + ((kk * 16) ^ t2_swz(c));
                bh[t] = *reinterpret_cast<const x3_u32x4*>(q); bm[t] = *reinterpret_cast<const x3_u32x4*>(q + T2_IPL); bl[t] = *reinterpret_cast<const x3_u32x4*>(q + 2 * T2_IPL);
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
        for (int t = 0; t < 2; ++t) {
            x3_f32x4 v = (a0[t] + (a1[t] + a2[t])) * sc + sh;
            if (relu) v = __builtin_elementwise_max(v, (x3_f32x4){0.f, 0.f, 0.f, 0.f});
            const int oh = h0 + row, ow = w0 + 16 * t + n;
            if (oh < H && ow < W) {
                *reinterpret_cast<x3_f32x4*>(yb + ((long long)oh * W + ow) * CO + co0) = v;
                vmax = x3_absmax4(vmax, v);
            }
        }
    }
    if (ysq) {
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, m));
        if (lane == 0) redmax[wave] = vmax;
        __syncthreads();
        if (tid == 0) {
            float m = redmax[0];
#pragma unroll
            for (int i = 1; i < NW; ++i) m = fmaxf(m, redmax[i]);
            atomicMax(reinterpret_cast<unsigned int*>(ysq) + ((blockIdx.x + blockIdx.y * gridDim.x) & 63) * 16, __float_as_uint(m * m));
        }
    }
}

// x (N, H, W, 32) [s2d: (N, 2H, 2W, 8)] -> y (N, H, W, Co) = [relu]([scale *] conv3x3(x) [+ shift]), Co = 16 or 32 (no s2d form); wimg: conv2d_tile_pack's image
int conv2d_tile_launch(const float* x, const float* wimg, const float* scale, const float* shift, float* y, int N, int H, int W, int Co, int s2d, int relu,
                       float* ysq, hipStream_t st) {
    if ((long long)H * W * T2_CI * 4 >= 0x7ffffff0LL || N > 65535) return fail(-1, "conv2d_tile: map too large for 32-bit offsets");
    if (!(Co == 16 || (Co == 32 && !s2d))) return fail(-1, "conv2d_tile: built for 32 -> 16 (plain or space-to-depth) and 32 -> 32 (got Co=%d s2d=%d)", Co, s2d);
    static std::atomic<bool> raised[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fail(-1, "conv2d_tile: cannot query the device");
    if (!raised[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)conv2d_tile_kernel<false, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, T2_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)conv2d_tile_kernel<true, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, T2_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)conv2d_tile_kernel<false, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, T2_LDS) != hipSuccess)
            return fail(-1, "conv2d_tile: cannot raise the dynamic LDS limit to %d bytes", T2_LDS);
        raised[dev].store(true, std::memory_order_release);
    }
    const int tw_ = (W + T2_TW - 1) / T2_TW, th_ = (H + T2_TH - 1) / T2_TH;
    const x3_u32x4* wi = reinterpret_cast<const x3_u32x4*>(wimg);
    if (Co == 32) hipLaunchKernelGGL((conv2d_tile_kernel<false, 32>), dim3(tw_ * th_, N), dim3(512), T2_LDS, st, x, wi, scale, shift, y, H, W, tw_, relu, ysq);
    else if (s2d) hipLaunchKernelGGL((conv2d_tile_kernel<true, 16>), dim3(tw_ * th_, N), dim3(256), T2_LDS, st, x, wi, scale, shift, y, H, W, tw_, relu, ysq);
    else hipLaunchKernelGGL((conv2d_tile_kernel<false, 16>), dim3(tw_ * th_, N), dim3(256), T2_LDS, st, x, wi, scale, shift, y, H, W, tw_, relu, ysq);
    return launch_status("conv2d_tile");
}

}  // namespace rcmvs

using namespace rcmvs;

extern "C" {

long long rcmvs_conv2d_tile_weight_floats(int Co) { return conv2d_tile_weight_floats(Co); }

int rcmvs_pack_conv2d_tile(const float* w, float* image, int Co, void* stream) {
    RCMVS_REQUIRE(w && image, "pack_conv2d_tile: null pointer");
    RCMVS_REQUIRE(Co == 16 || Co == 32, "pack_conv2d_tile: built for 16 or 32 output channels (got %d)", Co);
    hipLaunchKernelGGL(conv2d_tile_pack_kernel, dim3((T2_KS * (Co / 16) * 64 * 8 + 255) / 256), dim3(256), 0, as_stream(stream), w, reinterpret_cast<unsigned short*>(image), Co / 16);
    return launch_status("pack_conv2d_tile");
}

int rcmvs_conv2d_tile_fwd(const float* x, const float* image, const float* scale, const float* shift, float* y, int N, int H, int W, int Co, int s2d, int relu,
                          float* ysq_absmax, void* stream) {
    RCMVS_REQUIRE(x && image && y, "conv2d_tile_fwd: null pointer");
    RCMVS_REQUIRE(N > 0 && H > 0 && W > 0, "conv2d_tile_fwd: bad sizes");
    return conv2d_tile_launch(x, image, scale, shift, y, N, H, W, Co, s2d, relu, ysq_absmax, as_stream(stream));
}

}  // extern "C"
