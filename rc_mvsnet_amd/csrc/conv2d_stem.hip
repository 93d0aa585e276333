// FeatureNet's first block in ONE launch (round 6): conv0.0 (3 -> 8) -> conv0.1 (8 -> 8), each a 3x3 stride-1 Conv2d + BatchNorm(eval) + ReLU
// (models/modules.py:372-379,413-415), from the planar (N, 3, H, W) images to the channels-last (N, H, W, 8) map.  As two launches (the
// scalar-weight tile kernel of conv2d.hip, then the planar split-bf16 kernel of conv3d_x3.hip) they cost 18.5 + 24.4 us per DTU scene for
// 11.8 MB in, 31.5 MB written, read again and 31.5 MB out -- at 2.3 - 2.6 TB/s, bound by their launch / tick structure.  Here a block owns a
// 14 x 30 pixel tile of one view:
//   1. the 18 x 34 RGB halo goes to LDS (fp32, planar);
//   2. the first layer is evaluated on the 16 x 32 tile the second one needs -- a thread owns two vertically adjacent pixels (36 LDS reads for 54
//      taps), all eight channels, weights wave-uniform scalar operands, v_pk_fma_f32 on channel pairs in the tap order of conv2d_lds_kernel
//      (fp32 FMA chain: the arithmetic of the launch it replaces) -- BatchNorm + ReLU, zero outside the image (the second layer's padding),
//      split EXACTLY into three bf16 pieces (x = h + m + l by truncation) and parked in LDS: the 8-channel map never reaches memory;
//   3. the second layer runs on the matrix cores from there: six v_mfma_f32_16x16x32_bf16 per product, three magnitude classes in separate
//      accumulators (the arithmetic of conv3d_x3.hip, fp32-exact).  GEMM per wave: D[16 x 16] += A[16 x 32] B[32 x 16] with M = (s, co) -- the
//      outputs at columns 2n and 2n + 1 share lane n's activations --, N = 16 column pairs of a row, K step = one kernel row: four columns
//      kw' = kw + s x 8 channels (block-Toeplitz A: zero where kw' - s falls outside 0..2), three steps.
// gfx950 only.
#include "common.h"
#include "x3_pieces.h"

namespace rcmvs {

constexpr int ST_TH = 14, ST_TW = 30;                           // output tile
constexpr int ST_MH = ST_TH + 2, ST_MC = 32, ST_MW = 34;        // intermediate tile: 16 rows x 32 columns computed, 34 stored (the B fragments of column pair 15 reach columns 32, 33: zeros)
constexpr int ST_IH = ST_TH + 4, ST_IW = 34, ST_IS = 48;        // RGB halo; row stride in floats (two rows further = 32 banks further: the two half-waves of a read never collide)
constexpr int ST_RGB = 3 * ST_IH * ST_IS * 4;                   // bytes
constexpr int ST_MPL = ST_MH * ST_MW * 16;                      // bytes of one piece plane of the intermediate tile (8 channels x bf16 per pixel)
constexpr int ST_LDS = ST_RGB + 3 * ST_MPL;
constexpr int ST_KS = 3;                                        // K steps of the second layer (one per kernel row)
constexpr long long ST_IMG_HALFS = (long long)ST_KS * 3 * 64 * 8;
static_assert(ST_MH * ST_MC == 512, "two intermediate pixels per thread");

long long conv2d_stem_weight_floats() { return ST_IMG_HALFS / 2; }

// wb: Conv2d weight (8, 8, 3, 3) of the second layer -> A fragments [K step = kh][piece][lane][8]: row = lane & 15 = (s, co), k = 8 kk + ci with
// kk = lane >> 4 = kw' = kw + s; three bf16 pieces by truncation
__global__ void conv2d_stem_pack_kernel(const float* __restrict__ wb, unsigned short* __restrict__ img) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ST_KS * 64 * 8) return;
    const int ci = t & 7, lane = (t >> 3) & 63, kh = t >> 9;
    const int m = lane & 15, s = m >> 3, co = m & 7, kw = (lane >> 4) - s;
    const float v = (kw >= 0 && kw < 3) ? wb[((co * 8 + ci) * 3 + kh) * 3 + kw] : 0.0f;
    const unsigned hb = __float_as_uint(v) & 0xffff0000u;
    const float r1 = v - __uint_as_float(hb);
    const unsigned mb = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(mb);
    const unsigned lb = __float_as_uint(r2) & 0xffff0000u;
    const long long base = ((long long)kh * 3) * 512 + lane * 8 + ci;
    img[base] = (unsigned short)(hb >> 16);
    img[base + 512] = (unsigned short)(mb >> 16);
    img[base + 1024] = (unsigned short)(lb >> 16);
}

typedef float st_f2 __attribute__((ext_vector_type(2)));
#ifndef ST_ABL
#define ST_ABL 0            // timing ablations (tools/dev/build_variant.sh conv2d_stem ... -DST_ABL=mask, WRONG results): 1 no first-layer FMAs, 2 no MFMAs, 4 no image loads, 8 no split
#endif

__global__ __launch_bounds__(256, 2) void conv2d_stem_kernel(
    const float* __restrict__ x, const float* __restrict__ wa, const float* __restrict__ sa, const float* __restrict__ ha,
    const x3_u32x4* __restrict__ wimg, const float* __restrict__ sb, const float* __restrict__ hb, float* __restrict__ y, int H, int W, int tiles_w) {
    extern __shared__ __attribute__((aligned(16))) x3_byte smem[];
    float* const rgb = reinterpret_cast<float*>(smem);          // [3][ST_IH][ST_IS]
    x3_byte* const mb = smem + ST_RGB;                          // intermediate tile: three piece planes [ST_MH][ST_MW][16 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int view = blockIdx.y;
    const unsigned t2 = xcd_remap(blockIdx.x, gridDim.x);
    const int tw = t2 % tiles_w, th = t2 / tiles_w;
    const int h0 = th * ST_TH, w0 = tw * ST_TW;
    // ---- 1. RGB halo: rows h0 - 2 .. h0 + 15, columns w0 - 2 .. w0 + 31 of the three planes (zeros outside the image: the first layer's padding).
    // The 54 (channel, row) lines of 34 floats: columns 0 .. 31 by (line = i * 8 + tid / 32, column = tid % 32), columns 32, 33 by the first 108 threads
    // (a flat element index costs two divisions per load: a third of the kernel's vector instructions, and the kernel is bound by those)
    {
        __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (long long)view * 3 * H * W), (short)0, 3 * H * W * 4, 0x00020000);
        constexpr int OOB = 0x7ffffff0, NL = 3 * ST_IH, NI = (NL + 7) / 8;
        float v[NI + 1];
        auto fetch = [&](int line, int c) -> float {
            const int ch = (line >= ST_IH) + (line >= 2 * ST_IH), r = line - ch * ST_IH;
            const int ih = h0 - 2 + r, iw = w0 - 2 + c;
            const bool in = !(ST_ABL & 4) && line < NL && ih >= 0 && ih < H && iw >= 0 && iw < W;
            return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, in ? ((ch * H + ih) * W + iw) * 4 : OOB, 0, 0));
        };
#pragma unroll
        for (int i = 0; i < NI; ++i) v[i] = fetch(i * 8 + (tid >> 5), tid & 31);
        v[NI] = fetch(tid < 2 * NL ? (tid >> 1) : NL, 32 + (tid & 1));
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (i * 8 + (tid >> 5) < NL) rgb[(i * 8 + (tid >> 5)) * ST_IS + (tid & 31)] = v[i];
        if (tid < 2 * NL) rgb[(tid >> 1) * ST_IS + 32 + (tid & 1)] = v[NI];
    }
    // the two stored-but-not-computed columns of the intermediate tile
    if (tid < ST_MH * 2 * 3) {
        const int p = tid / (ST_MH * 2), r = (tid >> 1) % ST_MH, c = ST_MC + (tid & 1);
        *reinterpret_cast<x3_u32x4*>(mb + p * ST_MPL + (r * ST_MW + c) * 16) = (x3_u32x4){0u, 0u, 0u, 0u};
    }
    __syncthreads();
    // ---- 2. first layer: intermediate pixels (rows 2 pr, 2 pr + 1; column c) = image pixel (h0 - 1 + row, w0 - 1 + c)
    {
        const int c = tid & 31, pr = tid >> 5;
        st_f2 acc[2][4];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[p][q] = (st_f2){0.0f, 0.0f};
        // (tap order ky, kx, channel as in conv2d_lds_kernel; the row loop stays rolled: an unrolled body wants 216 weights in SGPRs at once)
        const float* src = rgb + (2 * pr) * ST_IS + c;
#pragma unroll 1
        for (int ky = 0; ky < ((ST_ABL & 1) ? 0 : 3); ++ky) {
            float xv[2][3][3];                                      // [pixel][kx][channel]
#pragma unroll
            for (int ch = 0; ch < 3; ++ch)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    xv[0][kx][ch] = src[(ch * ST_IH + ky) * ST_IS + kx];
                    xv[1][kx][ch] = src[(ch * ST_IH + ky + 1) * ST_IS + kx];
                }
            const float* wt = wa + ky * 3 * 4 * 8;                   // [tap][4 input channels, the fourth one zero][8]
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int ch = 0; ch < 3; ++ch)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const st_f2 wv = (st_f2){wt[(kx * 4 + ch) * 8 + 2 * q], wt[(kx * 4 + ch) * 8 + 2 * q + 1]};
#pragma unroll
                        for (int p = 0; p < 2; ++p) acc[p][q] = __builtin_elementwise_fma((st_f2){xv[p][kx][ch], xv[p][kx][ch]}, wv, acc[p][q]);
                    }
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r = 2 * pr + p;
            const int oh = h0 - 1 + r, ow = w0 - 1 + c;
            // ReLU and the second layer's zero padding in one v_med3_f32: median(v, 0, +inf) = max(v, 0) inside the image, median(v, 0, 0) = 0 outside
            const float lim = (oh >= 0 && oh < H && ow >= 0 && ow < W) ? __builtin_inff() : 0.0f;
            x3_f32x4 v[2];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v[q >> 1][2 * (q & 1)] = __builtin_amdgcn_fmed3f(fmaf(acc[p][q].x, sa[2 * q], ha[2 * q]), 0.0f, lim);
                v[q >> 1][2 * (q & 1) + 1] = __builtin_amdgcn_fmed3f(fmaf(acc[p][q].y, sa[2 * q + 1], ha[2 * q + 1]), 0.0f, lim);
            }
            x3_u32x2 h0_, m0_, l0_, h1_, m1_, l1_;
            if (ST_ABL & 8) { h0_ = m0_ = l0_ = (x3_u32x2){__float_as_uint(v[0][0]), __float_as_uint(v[0][1])}; h1_ = m1_ = l1_ = (x3_u32x2){__float_as_uint(v[1][0]), __float_as_uint(v[1][1])}; }
            else {
            x3_split4(v[0], h0_, m0_, l0_);
            x3_split4(v[1], h1_, m1_, l1_);
            }
            x3_byte* q = mb + (r * ST_MW + c) * 16;
            *reinterpret_cast<x3_u32x4*>(q) = (x3_u32x4){h0_.x, h0_.y, h1_.x, h1_.y};
            *reinterpret_cast<x3_u32x4*>(q + ST_MPL) = (x3_u32x4){m0_.x, m0_.y, m1_.x, m1_.y};
            *reinterpret_cast<x3_u32x4*>(q + 2 * ST_MPL) = (x3_u32x4){l0_.x, l0_.y, l1_.x, l1_.y};
        }
    }
    // ---- second layer's fragments (requested before the barrier)
    x3_u32x4 A[ST_KS][3];
#pragma unroll
    for (int j = 0; j < ST_KS; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p) A[j][p] = wimg[(j * 3 + p) * 64 + lane];
    const int n = lane & 15, kk = lane >> 4;
    const int s = kk >> 1, c4 = (kk & 1) * 4;                   // D fragment: rows 4 kk .. 4 kk + 3 = (s, channels c4 .. c4 + 3) of column pair n
    const x3_f32x4 scb = *reinterpret_cast<const x3_f32x4*>(sb + c4), shb = *reinterpret_cast<const x3_f32x4*>(hb + c4);
    __syncthreads();
    // ---- 3. second layer on the 14 x 30 output tile: a wave takes rows wave, wave + 4, wave + 8, wave + 12, two at a time (independent accumulator chains)
    const int col = 2 * n + s;
    float* yb = y + ((long long)view * H * W + (long long)h0 * W + w0 + col) * 8 + c4;
    const bool colin = col < ST_TW && w0 + col < W;
    const int boff = (2 * n + kk) * 16;                         // lane (n, kk): the 8 channels of column 2 n + kk
#pragma unroll 1
    for (int r0 = wave; r0 < ST_TH; r0 += 8) {
        const int rows[2] = {r0, r0 + 4};
        x3_f32x4 a0[2], a1[2], a2[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) a0[u] = a1[u] = a2[u] = (x3_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < ((ST_ABL & 2) ? 0 : ST_KS); ++j) {
            x3_u32x4 bh[2], bm[2], bl[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const x3_byte* q = mb + (min(rows[u], ST_TH - 1) + j) * (ST_MW * 16) + boff;
                bh[u] = *reinterpret_cast<const x3_u32x4*>(q); bm[u] = *reinterpret_cast<const x3_u32x4*>(q + ST_MPL); bl[u] = *reinterpret_cast<const x3_u32x4*>(q + 2 * ST_MPL);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) a0[u] = x3_mfma<3>(A[j][0], bh[u], a0[u]);
#pragma unroll
            for (int u = 0; u < 2; ++u) a1[u] = x3_mfma<3>(A[j][0], bm[u], a1[u]);
#pragma unroll
            for (int u = 0; u < 2; ++u) a2[u] = x3_mfma<3>(A[j][0], bl[u], a2[u]);
#pragma unroll
            for (int u = 0; u < 2; ++u) a1[u] = x3_mfma<3>(A[j][1], bh[u], a1[u]);
#pragma unroll
            for (int u = 0; u < 2; ++u) a2[u] = x3_mfma<3>(A[j][2], bh[u], a2[u]);
#pragma unroll
            for (int u = 0; u < 2; ++u) a2[u] = x3_mfma<3>(A[j][1], bm[u], a2[u]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            x3_f32x4 v = (a0[u] + (a1[u] + a2[u])) * scb + shb;
            v = __builtin_elementwise_max(v, (x3_f32x4){0.f, 0.f, 0.f, 0.f});
            if (colin && rows[u] < ST_TH && h0 + rows[u] < H) *reinterpret_cast<x3_f32x4*>(yb + (long long)rows[u] * W * 8) = v;
        }
    }
}

// x (N, 3, H, W) planar -> y (N, H, W, 8) = relu(bn_b(conv_b(relu(bn_a(conv_a(x))))));  wa: the first layer's weight as rcmvs_pack_conv2d_weight
// writes it with the input channels padded to four ([9][4][8]); wimg: conv2d_stem_pack's image of the second layer
int conv2d_stem_launch(const float* x, const float* wa, const float* sa, const float* ha, const float* wimg, const float* sb, const float* hb, float* y,
                       int N, int H, int W, hipStream_t st) {
    if ((long long)H * W * 8 * 4 >= 0x7ffffff0LL || N > 65535) return fail(-1, "conv2d_stem: map too large");
    const int tw_ = (W + ST_TW - 1) / ST_TW, th_ = (H + ST_TH - 1) / ST_TH;
    hipLaunchKernelGGL(conv2d_stem_kernel, dim3(tw_ * th_, N), dim3(256), ST_LDS, st, x, wa, sa, ha, reinterpret_cast<const x3_u32x4*>(wimg), sb, hb, y, H, W, tw_);
    return launch_status("conv2d_stem");
}

}  // namespace rcmvs

using namespace rcmvs;

extern "C" {

long long rcmvs_conv2d_stem_weight_floats(void) { return conv2d_stem_weight_floats(); }

int rcmvs_pack_conv2d_stem(const float* wb, float* image, void* stream) {
    RCMVS_REQUIRE(wb && image, "pack_conv2d_stem: null pointer");
    hipLaunchKernelGGL(conv2d_stem_pack_kernel, dim3((ST_KS * 64 * 8 + 255) / 256), dim3(256), 0, as_stream(stream), wb, reinterpret_cast<unsigned short*>(image));
    return launch_status("pack_conv2d_stem");
}

int rcmvs_conv2d_stem_fwd(const float* x, const float* w_a_packed, const float* scale_a, const float* shift_a, const float* image_b, const float* scale_b,
                          const float* shift_b, float* y, int N, int H, int W, void* stream) {
    RCMVS_REQUIRE(x && w_a_packed && scale_a && shift_a && image_b && scale_b && shift_b && y, "conv2d_stem_fwd: null pointer");
    RCMVS_REQUIRE(N > 0 && H > 0 && W > 0, "conv2d_stem_fwd: bad sizes");
    return conv2d_stem_launch(x, w_a_packed, scale_a, shift_a, image_b, scale_b, shift_b, y, N, H, W, as_stream(stream));
}

}  // extern "C"
