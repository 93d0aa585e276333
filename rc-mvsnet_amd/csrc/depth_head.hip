// K4: prob conv (8 -> 1, 3x3x3) + softmax over planes + soft-argmin depth + photometric
// confidence, fused: the logit volume, the probability volume and the two regression sums never
// leave the chip unless the caller asks for `prob`.
// Replaces CostRegNet.prob (models/modules.py:489,500), F.softmax, depth_regression (x2),
// F.pad + avg_pool3d and torch.gather in DepthNet_eval.forward (models/casmvsnet.py:293-309).
//
// A 256-thread block owns 32 consecutive pixels x all D planes.  Phase 1: every thread computes
// logits for (pixel = t % 32, plane = t / 32 + 8 i): 27 taps x 8 channels, two 16-byte loads per
// tap, weights wave-uniform -- B*D*h*w-way parallel, so stage 1 (20 480 pixels) still fills the
// chip.  The logits go to an LDS column [k][pixel] (conflict-free).  Phase 2: one thread per
// pixel runs the softmax / regression / confidence over its column in place.
#include "common.h"

namespace rcmvs {

constexpr int HEAD_PX = 32;
constexpr int HEAD_THREADS = 256;
constexpr int HEAD_KPAR = HEAD_THREADS / HEAD_PX;   // planes computed concurrently per pixel

__global__ __launch_bounds__(HEAD_THREADS) void depth_head_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ planes,
    float* __restrict__ depth, float* __restrict__ conf, float* __restrict__ prob, int D, int h, int w) {
    extern __shared__ __attribute__((aligned(16))) float col[];   // [D][HEAD_PX]
    const int b = blockIdx.y;
    const long long hw = (long long)h * w;
    const int px = threadIdx.x % HEAD_PX;
    const int kpar = threadIdx.x / HEAD_PX;
    long long p = (long long)blockIdx.x * HEAD_PX + px;
    const bool active = p < hw;
    if (!active) p = hw - 1;
    const int y = (int)(p / w), xx = (int)(p % w);
    const float* xb = x + (long long)b * D * hw * 8;

    for (int k = kpar; k < D; k += HEAD_KPAR) {
        float acc = 0.0f;
        for (int kd = 0; kd < 3; ++kd) {
            int id = k + kd - 1;
            if (id < 0 || id >= D) continue;
            for (int kh = 0; kh < 3; ++kh) {
                int ih = y + kh - 1;
                if (ih < 0 || ih >= h) continue;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    int iw = xx + kw - 1;
                    if (iw < 0 || iw >= w) continue;
                    const float* xp = xb + (((long long)id * h + ih) * w + iw) * 8;
                    const float* wt = wp + ((kd * 3 + kh) * 3 + kw) * 8;
                    float4 a = *reinterpret_cast<const float4*>(xp);
                    float4 c = *reinterpret_cast<const float4*>(xp + 4);
                    acc = fmaf(a.x, wt[0], acc); acc = fmaf(a.y, wt[1], acc);
                    acc = fmaf(a.z, wt[2], acc); acc = fmaf(a.w, wt[3], acc);
                    acc = fmaf(c.x, wt[4], acc); acc = fmaf(c.y, wt[5], acc);
                    acc = fmaf(c.z, wt[6], acc); acc = fmaf(c.w, wt[7], acc);
                }
            }
        }
        col[k * HEAD_PX + px] = acc;
    }
    __syncthreads();
    if (kpar != 0) return;

    float mx = -INFINITY;
    for (int k = 0; k < D; ++k) mx = fmaxf(mx, col[k * HEAD_PX + px]);
    // softmax (exp(x - max) / sum), in place
    float sum = 0.0f;
    for (int k = 0; k < D; ++k) {
        float e = expf(col[k * HEAD_PX + px] - mx);
        col[k * HEAD_PX + px] = e;
        sum += e;
    }
    const float2 pl = reinterpret_cast<const float2*>(planes)[(long long)b * hw + p];
    float dsum = 0.0f, isum = 0.0f;
    for (int k = 0; k < D; ++k) {
        float pk = col[k * HEAD_PX + px] / sum;
        col[k * HEAD_PX + px] = pk;
        float dk = pl.x + (float)k * pl.y;
        dsum += pk * dk;
        isum += pk * (float)k;
        if (prob && active) prob[((long long)b * D + k) * hw + p] = pk;
    }
    int idx = (int)isum;                       // .long(): truncation
    idx = idx < 0 ? 0 : (idx > D - 1 ? D - 1 : idx);
    float c = 0.0f;                            // ((p[i-1] + p[i]) + p[i+1]) + p[i+2], zero padded
#pragma unroll
    for (int j = -1; j <= 2; ++j) {
        int kk = idx + j;
        c += (kk >= 0 && kk < D) ? col[kk * HEAD_PX + px] : 0.0f;
    }
    if (active) {
        depth[(long long)b * hw + p] = dsum;
        conf[(long long)b * hw + p] = c;
    }
}

}  // namespace rcmvs

using namespace rcmvs;

extern "C" int rcmvs_depth_head_fwd(const float* x, const float* w_prob, const float* planes,
                                    float* depth, float* conf, float* prob,
                                    int B, int D, int h, int w, void* stream) {
    RCMVS_REQUIRE(x && w_prob && planes && depth && conf, "depth_head_fwd: null pointer");
    RCMVS_REQUIRE(B > 0 && D > 0 && h > 0 && w > 0, "depth_head_fwd: bad sizes");
    size_t lds = (size_t)D * HEAD_PX * sizeof(float);
    RCMVS_REQUIRE(lds <= 160 * 1024, "depth_head_fwd: D=%d needs %zu B of LDS (max 160 KiB)", D, lds);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)depth_head_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail((int)e, "depth_head_fwd: cannot raise dynamic LDS to %zu", lds);
    }
    dim3 grid((unsigned)cdiv((long long)h * w, HEAD_PX), B);
    hipLaunchKernelGGL(depth_head_kernel, grid, dim3(HEAD_THREADS), lds, as_stream(stream), x, w_prob, planes, depth, conf,
                       prob, D, h, w);
    return launch_status("depth_head_fwd");
}
