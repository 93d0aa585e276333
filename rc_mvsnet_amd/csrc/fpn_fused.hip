// Last FPN level of FeatureNet in one kernel (models/modules.py:448-462, arch_mode='fpn'):
//   intra = nearest_upsample_x2(prev) + inner2(conv0)        1x1 conv 8 -> 32 + bias, at full resolution
//   out   = out3(intra)                                      3x3 conv 32 -> 8, no bias
// Unfused, `intra` (32 channels at full resolution: 126 MB for three 512x640 views) is written by the merge kernel and read
// back by the output conv -- more bytes than everything else FeatureNet moves.  Here the 3x3 conv's halo tile of `intra` is
// formed on the fly in LDS, 8 channels at a time, from the 8-channel lateral map (staged once per block) and the half-
// resolution previous level, so `intra` never exists in memory: reads 31 + 31 MB, writes 31 MB.  The arithmetic order of both
// convolutions is the one of conv2d_lds_kernel (conv2d.hip), so the result is bit-identical to the two-kernel path.
// gfx950 only.
#include "common.h"

namespace rcmvs {

typedef float f4v __attribute__((ext_vector_type(4)));

template <int CL, int CM, int CO>
__global__ __launch_bounds__(256) void fpn_out_fused_kernel(
    const float* __restrict__ lat, const float* __restrict__ up, const float* __restrict__ w_in, const float* __restrict__ b_in,
    const float* __restrict__ w_out, float* __restrict__ y, int H, int W, int tiles_w) {
    static_assert(CL == 8 && CM % 8 == 0 && CO % 4 == 0, "lateral map has 8 channels");
    constexpr int TH = 16, TW = 16, K = 3, HH = TH + 2, HW = TW + 2, NP = HH * HW;
    constexpr int CK = 8, ST = CK + 4;                                  // channels per pass, floats per staged pixel
    __shared__ __attribute__((aligned(16))) float lat_s[NP * ST];       // lateral halo tile, all 8 channels
    __shared__ __attribute__((aligned(16))) float tile[NP * ST];        // 8 channels of intra
    __shared__ float w_s[CL * CM];
    __shared__ float b_s[CM];
    const int n = blockIdx.y;
    const unsigned t2 = xcd_remap(blockIdx.x, gridDim.x);
    const int tw = t2 % tiles_w, th = t2 / tiles_w;
    const int oy0 = th * TH, ox0 = tw * TW;
    const int lx = threadIdx.x % TW, ly = threadIdx.x / TW;
    const int Hh = H / 2, Wh = W / 2;
    const float* lb = lat + (long long)n * H * W * CL;
    const float* ub = up + (long long)n * Hh * Wh * CM;

    for (int e = threadIdx.x; e < CL * CM; e += 256) w_s[e] = w_in[e];
    if (threadIdx.x < CM) b_s[threadIdx.x] = b_in[threadIdx.x];
    for (int e = threadIdx.x; e < NP * 2; e += 256) {
        const int v = e >> 1, c4 = e & 1;
        const int iy = oy0 - 1 + v / HW, ix = ox0 - 1 + v % HW;
        f4v val = (f4v){0.f, 0.f, 0.f, 0.f};
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) val = *reinterpret_cast<const f4v*>(lb + ((long long)iy * W + ix) * CL + c4 * 4);
        *reinterpret_cast<f4v*>(lat_s + v * ST + c4 * 4) = val;
    }

    // output channel PAIRS per v_pk_fma_f32: acc2[co / 2] += splat(x) * (w[co], w[co + 1]) -- the splat is an operand selector of the
    // packed instruction, the weight pair two neighbouring SGPRs: half the VALU instructions of the scalar form, same arithmetic
    typedef float f2v __attribute__((ext_vector_type(2)));
    f2v acc2[CO / 2];
#pragma unroll
    for (int c = 0; c < CO / 2; ++c) acc2[c] = (f2v){0.0f, 0.0f};

    for (int c0 = 0; c0 < CM; c0 += CK) {
        __syncthreads();                                                // lat_s / w_s ready (first pass); tile free (later passes)
        for (int e = threadIdx.x; e < NP * 2; e += 256) {
            const int v = e >> 1, c4 = e & 1;
            const int iy = oy0 - 1 + v / HW, ix = ox0 - 1 + v % HW;
            f4v val = (f4v){0.f, 0.f, 0.f, 0.f};                        // the 3x3 conv zero-pads intra, not its inputs
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                const f4v x0 = *reinterpret_cast<const f4v*>(lat_s + v * ST), x1 = *reinterpret_cast<const f4v*>(lat_s + v * ST + 4);
                const f4v u4 = *reinterpret_cast<const f4v*>(ub + ((long long)(iy >> 1) * Wh + (ix >> 1)) * CM + c0 + c4 * 4);
                const float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int co = c0 + c4 * 4 + k;
                    float a = 0.0f;
#pragma unroll
                    for (int ci = 0; ci < CL; ++ci) a = fmaf(xs[ci], w_s[ci * CM + co], a);
                    a = a + b_s[co];
                    val[k] = u4[k] + a;
                }
            }
            *reinterpret_cast<f4v*>(tile + v * ST + c4 * 4) = val;
        }
        __syncthreads();
#pragma unroll 1
        for (int ky = 0; ky < K; ++ky) {
#pragma unroll 1
            for (int kx = 0; kx < K; ++kx) {
                const float* wt = w_out + ((long long)(ky * K + kx) * CM + c0) * CO;
#pragma unroll
                for (int c4 = 0; c4 < CK / 4; ++c4) {
                    const f4v xv = *reinterpret_cast<const f4v*>(tile + ((ly + ky) * HW + (lx + kx)) * ST + c4 * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int co = 0; co < CO; co += 2)
                            acc2[co / 2] = __builtin_elementwise_fma((f2v){xv[j], xv[j]}, (f2v){wt[(c4 * 4 + j) * CO + co], wt[(c4 * 4 + j) * CO + co + 1]}, acc2[co / 2]);
                }
            }
        }
    }
    const int oy = oy0 + ly, ox = ox0 + lx;
    if (oy < H && ox < W) {
        float* yp = y + (((long long)n * H + oy) * W + ox) * CO;
#pragma unroll
        for (int co = 0; co < CO; co += 4) *reinterpret_cast<float4*>(yp + co) = make_float4(acc2[co / 2].x, acc2[co / 2].y, acc2[co / 2 + 1].x, acc2[co / 2 + 1].y);
    }
}


// ------------------------------------------------------------------------------------------------------------------------------
// The same level with the two convolutions FOLDED (round 4).  The fused kernel above is VALU-bound (62 % busy, 82 us per scene for
// three 512x640 views): per output pixel 9 x 32 x 8 multiply-adds of the 3x3 conv plus the 1x1 conv of its 18 x 18 halo.  Both
// convolutions are linear and nothing sits between them, so
//   out3(up2(prev) + inner2(lat) + b) = (W3 o W2) * lat                  a 3x3 conv 8 -> 8 on the lateral map         (WB, 576 weights)
//                                     + W3 * up2(prev)                   on a nearest-upsampled map: per output parity (py, px) the
//                                                                        nine taps fall on a 2 x 2 block of `prev`, whose combined
//                                                                        weights are sums of W3 taps                   (WA, 4 x 4 x 32 x 8)
//                                     + sum over the taps INSIDE the image of W3[tap] b   (zero padding applies to the merged map,
//                                                                        so border pixels miss some taps: nine classes, BS)
// = 576 + 1024 multiply-adds per pixel instead of 2304 + 324, and neither the merged map nor its halo is ever formed.  H and W are even,
// so two taps that share a source pixel of `prev` are inside or outside the image together and the zero-filled tiles handle the borders.
// The tables are built once per weight set on the host side (ops.pack_fpn_folded, fp64 products); the result differs from the
// two-kernel path by fp32 rounding only (1e-6 relative: test_fpn_out_folded_matches_the_unfused_path).
// Mapping: a block owns 16 x 16 outputs; wave w owns the 64 pixels of parity class (py, px) = (w >> 1, w & 1), so its WA block is
// wave-uniform (scalar-cache operands of v_pk_fma_f32, as above).  LDS: the lateral halo tile de-interleaved by pixel parity
// (4 x 9 x 9 pixels: a wave's nine taps are unit-stride reads of one parity plane each) and the 10 x 10 x 32 tile of `prev`;
// strides 12 / 160 and 36 / 416 floats make every ds_read_b128 lane group conflict-free (exhaustive search over the strides).
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int FF_LST = 12, FF_LRS = 160, FF_LPL = 9 * FF_LRS;          // lateral tile: floats per pixel / row / parity plane
constexpr int FF_UST = 36, FF_URS = 416;                               // `prev` tile: floats per pixel / row
constexpr int FF_WB = 0, FF_BS = 576, FF_WA = 576 + 72, FF_TABLE = 576 + 72 + 4096;

__global__ __launch_bounds__(256) void fpn_out_folded_kernel(
    const float* __restrict__ lat, const float* __restrict__ up, const float* __restrict__ tab, float* __restrict__ y, float* __restrict__ ysq, int H, int W, int tiles_w) {
    typedef float f2v __attribute__((ext_vector_type(2)));
    __shared__ float red[4];
    __shared__ __attribute__((aligned(16))) float lat_s[4 * FF_LPL];
    __shared__ __attribute__((aligned(16))) float up_s[10 * FF_URS];
    const int n = blockIdx.y;
    const unsigned t2 = xcd_remap(blockIdx.x, gridDim.x);
    const int tw = t2 % tiles_w, th = t2 / tiles_w;
    const int oy0 = th * 16, ox0 = tw * 16;
    const int Hh = H / 2, Wh = W / 2;
    const float* lb = lat + (long long)n * H * W * 8;
    const float* ub = up + (long long)n * Hh * Wh * 32;
    for (int e = threadIdx.x; e < 18 * 18 * 2; e += 256) {              // lateral halo: image pixel (oy0 - 1 + hy, ox0 - 1 + hx), zero outside
        const int v = e >> 1, c4 = e & 1;
        const int hy = v / 18, hx = v % 18;
        const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
        f4v val = (f4v){0.f, 0.f, 0.f, 0.f};
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) val = *reinterpret_cast<const f4v*>(lb + ((long long)iy * W + ix) * 8 + c4 * 4);
        *reinterpret_cast<f4v*>(lat_s + ((hy & 1) * 2 + (hx & 1)) * FF_LPL + (hy >> 1) * FF_LRS + (hx >> 1) * FF_LST + c4 * 4) = val;
    }
    for (int e = threadIdx.x; e < 10 * 10 * 8; e += 256) {              // `prev` tile: half-resolution pixel (oy0 / 2 - 1 + ur, ox0 / 2 - 1 + uc)
        const int v = e >> 3, c4 = e & 7;
        const int ur = v / 10, uc = v % 10;
        const int iy = oy0 / 2 - 1 + ur, ix = ox0 / 2 - 1 + uc;
        f4v val = (f4v){0.f, 0.f, 0.f, 0.f};
        if (iy >= 0 && iy < Hh && ix >= 0 && ix < Wh) val = *reinterpret_cast<const f4v*>(ub + ((long long)iy * Wh + ix) * 32 + c4 * 4);
        *reinterpret_cast<f4v*>(up_s + ur * FF_URS + uc * FF_UST + c4 * 4) = val;
    }
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int py = wv >> 1, px = wv & 1;
    const int li = threadIdx.x & 7, lj = (threadIdx.x >> 3) & 7;
    const int oy = oy0 + 2 * lj + py, ox = ox0 + 2 * li + px;
    // bias of the lateral conv through the taps that are inside the image
    const int cy = oy == 0 ? 0 : (oy == H - 1 ? 2 : 1), cx = ox == 0 ? 0 : (ox == W - 1 ? 2 : 1);
    const float* bs = tab + FF_BS + (cy * 3 + cx) * 8;
    f2v acc2[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc2[c] = (f2v){bs[2 * c], bs[2 * c + 1]};
    __syncthreads();
    // ---- (W3 o W2) * lat: 3x3, 8 -> 8
#pragma unroll 1
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll 1
        for (int kx = 0; kx < 3; ++kx) {
            const int hy = py + ky, hx = px + kx;                        // halo coordinates minus (2 lj, 2 li): wave-uniform parity plane
            const float* xp = lat_s + ((hy & 1) * 2 + (hx & 1)) * FF_LPL + (lj + (hy >> 1)) * FF_LRS + (li + (hx >> 1)) * FF_LST;
            const f4v x0 = *reinterpret_cast<const f4v*>(xp), x1 = *reinterpret_cast<const f4v*>(xp + 4);
            const float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
            const float* wt = tab + FF_WB + (ky * 3 + kx) * 64;
#pragma unroll
            for (int ci = 0; ci < 8; ++ci)
#pragma unroll
                for (int co = 0; co < 8; co += 2)
                    acc2[co / 2] = __builtin_elementwise_fma((f2v){xs[ci], xs[ci]}, (f2v){wt[ci * 8 + co], wt[ci * 8 + co + 1]}, acc2[co / 2]);
        }
    }
    // ---- W3 * up2(prev): the 2 x 2 block of `prev` under this parity class, 32 -> 8
    const float* wa = tab + FF_WA + wv * 1024;
#pragma unroll 1
    for (int ry = 0; ry < 2; ++ry) {
#pragma unroll 1
        for (int rx = 0; rx < 2; ++rx) {
            const float* xp = up_s + (lj + py + ry) * FF_URS + (li + px + rx) * FF_UST;
            const float* wt = wa + (ry * 2 + rx) * 256;
#pragma unroll 2
            for (int c4 = 0; c4 < 8; ++c4) {
                const f4v xv = *reinterpret_cast<const f4v*>(xp + c4 * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int co = 0; co < 8; co += 2)
                        acc2[co / 2] = __builtin_elementwise_fma((f2v){xv[j], xv[j]}, (f2v){wt[(c4 * 4 + j) * 8 + co], wt[(c4 * 4 + j) * 8 + co + 1]}, acc2[co / 2]);
            }
        }
    }
    if (oy < H && ox < W) {
        float* yp = y + (((long long)n * H + oy) * W + ox) * 8;
        *reinterpret_cast<float4*>(yp) = make_float4(acc2[0].x, acc2[0].y, acc2[1].x, acc2[1].y);
        *reinterpret_cast<float4*>(yp + 4) = make_float4(acc2[2].x, acc2[2].y, acc2[3].x, acc2[3].y);
    }
    if (ysq) {                                                          // (max |y|)^2 of the block: one atomic max into slot (block & 63)
        float m = 0.0f;
        if (oy < H && ox < W) {
#pragma unroll
            for (int c = 0; c < 4; ++c) m = fmaxf(m, fmaxf(fabsf(acc2[c].x), fabsf(acc2[c].y)));
        }
#pragma unroll
        for (int k = 32; k > 0; k >>= 1) m = fmaxf(m, __shfl_xor(m, k));
        if ((threadIdx.x & 63) == 0) red[wv] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            atomicMax(reinterpret_cast<unsigned int*>(ysq) + ((blockIdx.x + blockIdx.y * gridDim.x) & 63) * 16, __float_as_uint(m * m));
        }
    }
}

}  // namespace rcmvs

using namespace rcmvs;

extern "C" int rcmvs_fpn_out_fused(const float* lat, const float* up, const float* w_inner, const float* b_inner, const float* w_out,
                                   float* y, int N, int H, int W, int CL, int CM, int CO, void* stream) {
    RCMVS_REQUIRE(lat && up && w_inner && b_inner && w_out && y, "fpn_out_fused: null pointer");
    RCMVS_REQUIRE(N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "fpn_out_fused: H and W must be even (got %d x %d)", H, W);
    RCMVS_REQUIRE(CL == 8 && CM == 32 && CO == 8, "fpn_out_fused: unsupported channels %d -> %d -> %d (8 -> 32 -> 8 only)", CL, CM, CO);
    const int tiles_w = (W + 15) / 16, tiles_h = (H + 15) / 16;
    hipLaunchKernelGGL((fpn_out_fused_kernel<8, 32, 8>), dim3(tiles_w * tiles_h, N), dim3(256), 0, as_stream(stream), lat, up, w_inner,
                       b_inner, w_out, y, H, W, tiles_w);
    return launch_status("fpn_out_fused");
}

extern "C" int rcmvs_fpn_out_folded(const float* lat, const float* up, const float* tables, float* y, float* ysq_absmax, int N, int H, int W, void* stream) {
    RCMVS_REQUIRE(lat && up && tables && y, "fpn_out_folded: null pointer");
    RCMVS_REQUIRE(N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "fpn_out_folded: H and W must be even (got %d x %d)", H, W);
    const int tiles_w = (W + 15) / 16, tiles_h = (H + 15) / 16;
    hipLaunchKernelGGL(fpn_out_folded_kernel, dim3(tiles_w * tiles_h, N), dim3(256), 0, as_stream(stream), lat, up, tables, y, ysq_absmax, H, W, tiles_w);
    return launch_status("fpn_out_folded");
}
