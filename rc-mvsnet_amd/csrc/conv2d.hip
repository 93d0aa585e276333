// 2-D convolution family of the feature pyramid (FeatureNet, arch_mode='fpn', models/modules.py:363-464),
// channels-last, for inference: conv + folded BatchNorm (or bias) + ReLU, and the FPN merge
//   intra = nearest_upsample_x2(prev) + conv1x1(lateral) + bias          (modules.py:448-455)
// fused as an "upsample-add" epilogue.  Same mapping as conv3d_lds.hip: one thread per output pixel,
// all Cout accumulators in registers, weights wave-uniform (scalar loads, v_pk_fma with SGPR operands),
// input halo tile staged in LDS 8 channels at a time at a conflict-free padded stride.  The outputs
// (B*V, h, w, C) are exactly the channels-last maps K1 consumes, so no layout pass is needed.
// gfx950 only.
#include "common.h"

namespace rcmvs {

constexpr int C2_CK = 8;                 // channels staged per pass
constexpr int C2_STRIDE = C2_CK + 4;     // floats per staged pixel

// TH x TW output tile (TH*TW == 256)
template <int CI, int CO, int K, int S, int TH, int TW>
__global__ __launch_bounds__(256) void conv2d_lds_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ up, float* __restrict__ y,
    int H, int W, int Ho, int Wo, int tiles_w, int relu) {
    static_assert(TH * TW == 256, "tile must have 256 pixels");
    constexpr int PAD = K / 2;
    constexpr int HH = (TH - 1) * S + K, HW = (TW - 1) * S + K;     // halo tile
    extern __shared__ __attribute__((aligned(16))) float tile[];    // [HH*HW][C2_STRIDE]
    const int n = blockIdx.y;
    const unsigned t2 = xcd_remap(blockIdx.x, gridDim.x);
    const int tw = t2 % tiles_w, th = t2 / tiles_w;
    const int oy0 = th * TH, ox0 = tw * TW;
    const int lx = threadIdx.x % TW, ly = threadIdx.x / TW;
    const int oy = oy0 + ly, ox = ox0 + lx;
    const bool inside = oy < Ho && ox < Wo;
    const float* xb = x + (long long)n * H * W * CI;
    const int iy0 = oy0 * S - PAD, ix0 = ox0 * S - PAD;

    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = 0.0f;

    for (int c0 = 0; c0 < CI; c0 += C2_CK) {
        const int ck = (CI - c0 < C2_CK) ? (CI - c0) : C2_CK;       // multiple of 4
        const int q = ck >> 2;
        if (c0 > 0) __syncthreads();
        for (int e = threadIdx.x; e < HH * HW * q; e += 256) {
            const int v = e / q, c4 = e - v * q;
            const int hx = v % HW, hy = v / HW;
            const int iy = iy0 + hy, ix = ix0 + hx;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iy >= 0 && iy < H && ix >= 0 && ix < W)
                val = *reinterpret_cast<const float4*>(xb + ((long long)iy * W + ix) * CI + c0 + c4 * 4);
            *reinterpret_cast<float4*>(tile + v * C2_STRIDE + c4 * 4) = val;
        }
        __syncthreads();
        for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const float* tp = tile + ((ly * S + ky) * HW + (lx * S + kx)) * C2_STRIDE;
                const float* wt = wp + ((long long)(ky * K + kx) * CI + c0) * CO;
#pragma unroll
                for (int c4 = 0; c4 < C2_CK / 4; ++c4) {
                    if (c4 * 4 < ck) {
                        const float4 xv = *reinterpret_cast<const float4*>(tp + c4 * 4);
#pragma unroll
                        for (int co = 0; co < CO; ++co) {
                            acc[co] = fmaf(xv.x, wt[(c4 * 4 + 0) * CO + co], acc[co]);
                            acc[co] = fmaf(xv.y, wt[(c4 * 4 + 1) * CO + co], acc[co]);
                            acc[co] = fmaf(xv.z, wt[(c4 * 4 + 2) * CO + co], acc[co]);
                            acc[co] = fmaf(xv.w, wt[(c4 * 4 + 3) * CO + co], acc[co]);
                        }
                    }
                }
            }
    }
    if (!inside) return;
    const long long op = ((long long)n * Ho + oy) * Wo + ox;
    float* yp = y + op * CO;
    const float* upp = up ? up + (((long long)n * (Ho / 2) + oy / 2) * (Wo / 2) + ox / 2) * CO : nullptr;
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        float v = acc[co];
        if (scale) v = v * scale[co];
        if (shift) v = v + shift[co];
        if (upp) v = upp[co] + v;                                   // F.interpolate(intra) + inner(conv)
        if (relu) v = fmaxf(v, 0.0f);
        acc[co] = v;
    }
#pragma unroll
    for (int co = 0; co < CO; co += 4)
        *reinterpret_cast<float4*>(yp + co) = make_float4(acc[co], acc[co + 1], acc[co + 2], acc[co + 3]);
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

template <int CI, int CO, int K, int S, int TH, int TW>
static int conv2d_launch_t(const float* x, const float* wp, const float* scale, const float* shift, const float* up, float* y,
                           int N, int H, int W, int relu, hipStream_t st) {
    constexpr int PAD = K / 2;
    const int Ho = (H + 2 * PAD - K) / S + 1, Wo = (W + 2 * PAD - K) / S + 1;
    const int tiles_w = (Wo + TW - 1) / TW, tiles_h = (Ho + TH - 1) / TH;
    constexpr int HH = (TH - 1) * S + K, HW = (TW - 1) * S + K;
    const size_t lds = (size_t)HH * HW * C2_STRIDE * sizeof(float);
    hipLaunchKernelGGL((conv2d_lds_kernel<CI, CO, K, S, TH, TW>), dim3(tiles_w * tiles_h, N), dim3(256), lds, st, x, wp, scale,
                       shift, up, y, H, W, Ho, Wo, tiles_w, relu);
    return launch_status("conv2d");
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

int rcmvs_conv2d_fwd(const float* x, const float* w_packed, const float* scale, const float* shift, const float* up_add,
                     float* y, int N, int H, int W, int Ci, int Co, int K, int stride, int relu, void* stream) {
    RCMVS_REQUIRE(x && w_packed && y, "conv2d_fwd: null pointer");
    RCMVS_REQUIRE(N > 0 && H > 0 && W > 0, "conv2d_fwd: bad sizes");
    hipStream_t st = as_stream(stream);
#define RCMVS_C2(CI, CO, KK, SS, TH, TW)                                                                          \
    if (Ci == CI && Co == CO && K == KK && stride == SS)                                                          \
        return conv2d_launch_t<CI, CO, KK, SS, TH, TW>(x, w_packed, scale, shift, up_add, y, N, H, W, relu, st);
    // the 13 layers of FeatureNet(base_channels=8, fpn, 3 stages)
    RCMVS_C2(4, 8, 3, 1, 16, 16) RCMVS_C2(8, 8, 3, 1, 16, 16) RCMVS_C2(8, 16, 5, 2, 8, 32) RCMVS_C2(16, 16, 3, 1, 16, 16)
    RCMVS_C2(16, 32, 5, 2, 8, 32) RCMVS_C2(32, 32, 3, 1, 16, 16) RCMVS_C2(32, 32, 1, 1, 16, 16) RCMVS_C2(16, 32, 1, 1, 16, 16)
    RCMVS_C2(32, 16, 3, 1, 16, 16) RCMVS_C2(8, 32, 1, 1, 16, 16) RCMVS_C2(32, 8, 3, 1, 16, 16)
#undef RCMVS_C2
    return fail(-1, "conv2d_fwd: unsupported layer Ci=%d Co=%d K=%d stride=%d", Ci, Co, K, stride);
}

}  // extern "C"
