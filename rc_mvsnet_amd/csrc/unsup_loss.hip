// Self-supervised photometric loss of the training step: inverse warp of every source
// image into the reference view through the estimated depth, photometric + gradient smooth-L1, SSIM, image-aware depth
// smoothness, per-pixel best source view -- forward (three scalars per stage) and backward (d loss / d depth).
// Reference: losses/homography.py:6-200, losses/modules.py:6-82, losses/unsup_loss.py:14-94.  gfx950 only.
//
// The reference runs this as ~150 ATen launches per stage (index gathers, slices, pools, topk); here a stage is
//   forward : Vs x (warp, terms) + smoothness + best-view count + finalize        = 2 Vs + 3 launches
//   backward: scalars + smoothness + Vs x (SSIM window coefficients, view)        = 2 Vs + 2 launches
// Every kernel is one thread per pixel over (B, H, W), HBM-bound on ~10 floats per pixel; the per-pixel arithmetic
// lives in unsup_loss_math.h (shared with the CPU harness of tests/test_unsup_loss_cpu.py).  Scalars never leave the
// device: sums are fp64 atomics (one per block and term), the per-view losses, counts and backward factors are read by
// the consuming kernels from device memory, so a training step has no host synchronisation here.
#include "common.h"
#include "unsup_loss_math.h"

namespace rcmvs {

constexpr int UL_BLOCK = 256;

__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// block-reduce NT per-thread partials and add them to out[0..NT) with one fp64 atomic each
template <int NT>
__device__ inline void block_accumulate(const float* part, double* out) {
    __shared__ double red[NT][UL_BLOCK / WAVE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const double s = wave_sum((double)part[t]);
        if (lane == 0) red[t][wave] = s;
    }
    __syncthreads();
    if (threadIdx.x < NT) {
        double s = 0.0;
        for (int w = 0; w < UL_BLOCK / WAVE; ++w) s += red[threadIdx.x][w];
        unsafeAtomicAdd(out + threadIdx.x, s);
    }
}

struct Pix { int b, y, x; bool ok; long long p; };
__device__ inline Pix pixel_of(int B, int H, int W) {
    Pix q;
    const long long gid = (long long)blockIdx.x * UL_BLOCK + threadIdx.x;
    const long long plane = (long long)H * W;
    q.ok = gid < (long long)B * plane;
    const long long g = q.ok ? gid : 0;
    q.b = (int)(g / plane);
    const int r = (int)(g - (long long)q.b * plane);
    q.y = r / W; q.x = r - q.y * W;
    q.p = g;
    return q;
}

__global__ __launch_bounds__(UL_BLOCK) void inverse_warp_kernel(const float* __restrict__ src, const float* __restrict__ depth,
                                                                 const float* __restrict__ coef, float* __restrict__ warped,
                                                                 float* __restrict__ mask, int B, int H, int W) {
    const Pix q = pixel_of(B, H, W);
    if (!q.ok) return;
    const float* img = src + (long long)q.b * H * W * 3;
    const ul::Taps t = ul::inv_warp_taps(coef + q.b * 12, q.x, q.y, depth[q.p], H, W);
#pragma unroll
    for (int c = 0; c < 3; ++c) warped[q.p * 3 + c] = ul::tap_value(t, img, c);
    mask[q.p] = t.mask;
}

__global__ __launch_bounds__(UL_BLOCK) void photo_terms_kernel(const float* __restrict__ warped, const float* __restrict__ ref,
                                                                const float* __restrict__ mask, double* __restrict__ sums,
                                                                int B, int H, int W) {
    const Pix q = pixel_of(B, H, W);
    float part[4] = {0.f, 0.f, 0.f, 0.f};
    if (q.ok) {
        const long long o = (long long)q.b * H * W;
        ul::photo_terms(warped + o * 3, ref + o * 3, mask + o, q.y, q.x, H, W, part);
    }
    block_accumulate<4>(part, sums);
}

__global__ __launch_bounds__(UL_BLOCK) void smooth_terms_kernel(const float* __restrict__ depth, const float* __restrict__ ref,
                                                                 double* __restrict__ sums, int B, int H, int W) {
    const Pix q = pixel_of(B, H, W);
    float part[2] = {0.f, 0.f};
    if (q.ok) {
        const long long o = (long long)q.b * H * W;
        ul::smooth_terms(depth + o, ref + o * 3, q.y, q.x, H, W, part);
    }
    block_accumulate<2>(part, sums);
}

// counts[v] = number of pixels whose cheapest valid view is v
__global__ __launch_bounds__(UL_BLOCK) void best_view_kernel(const float* __restrict__ masks, const double* __restrict__ sums,
                                                              int* __restrict__ counts, int B, int Vs, int H, int W) {
    const Pix q = pixel_of(B, H, W);
    float L[RCMVS_UNSUP_MAX_VIEWS];
    for (int v = 0; v < Vs; ++v) L[v] = ul::view_loss(sums + v * 4, B, H, W);
    const int best = q.ok ? ul::best_view(L, masks, (long long)B * H * W, q.p, Vs) : -1;
    for (int v = 0; v < Vs; ++v) {
        const unsigned long long m = __ballot(best == v);
        if ((threadIdx.x & 63) == 0 && m) atomicAdd(counts + v, __popcll(m));
    }
}

// out[0..3] = reconstr, ssim, smooth, 12 r + 6 s + 0.18 m ; out[4 + v] = L_v
__global__ void unsup_finalize_kernel(const double* __restrict__ sums, const int* __restrict__ counts, float* __restrict__ out,
                                      int B, int Vs, int H, int W) {
    if (threadIdx.x != 0) return;
    const double n = (double)B * H * W;
    double rec = 0.0, ssim = 0.0;
    for (int v = 0; v < Vs; ++v) {
        const float L = ul::view_loss(sums + v * 4, B, H, W);
        out[4 + v] = L;
        rec += (double)L * counts[v];
        if (v < 2) ssim += sums[v * 4 + 3] / ((double)B * (H - 2) * (W - 2) * 3);
    }
    rec /= n;
    const double* ss = sums + Vs * 4;
    const double smooth = ss[0] / ((double)B * H * (W - 1)) + ss[1] / ((double)B * (H - 1) * W);
    out[0] = (float)rec; out[1] = (float)ssim; out[2] = (float)smooth;
    out[3] = (float)(12.0 * rec + 6.0 * ssim + 0.18 * smooth);
}

// kbuf[v * 4 + {0,1,2,3}] = per-element factors of the photo / dx / dy / ssim sums of view v; kbuf[Vs * 4 + {0,1}] = smoothness
__global__ void unsup_bwd_scalars_kernel(const float* __restrict__ gout, const int* __restrict__ counts, float* __restrict__ kbuf,
                                         int B, int Vs, int H, int W) {
    const int v = threadIdx.x;
    const double n = (double)B * H * W;
    if (v < Vs) {
        const double share = 0.5 * (double)gout[0] * ((double)counts[v] / n);
        kbuf[v * 4 + 0] = (float)(share / (n * 3));
        kbuf[v * 4 + 1] = (float)(share / ((double)B * H * (W - 1) * 3));
        kbuf[v * 4 + 2] = (float)(share / ((double)B * (H - 1) * W * 3));
        kbuf[v * 4 + 3] = v < 2 ? (float)((double)gout[1] / ((double)B * (H - 2) * (W - 2) * 3)) : 0.0f;
    }
    if (v == 0) {
        kbuf[Vs * 4 + 0] = (float)((double)gout[2] / ((double)B * H * (W - 1)));
        kbuf[Vs * 4 + 1] = (float)((double)gout[2] / ((double)B * (H - 1) * W));
    }
}

__global__ __launch_bounds__(UL_BLOCK) void smooth_bwd_kernel(const float* __restrict__ depth, const float* __restrict__ ref,
                                                               const float* __restrict__ k, float* __restrict__ gdepth,
                                                               int B, int H, int W) {
    const Pix q = pixel_of(B, H, W);
    if (!q.ok) return;
    const long long o = (long long)q.b * H * W;
    gdepth[q.p] = ul::smooth_grad(depth + o, ref + o * 3, k, q.y, q.x, H, W);
}

// one thread per SSIM window (B, H-2, W-2)
__global__ __launch_bounds__(UL_BLOCK) void ssim_coef_kernel(const float* __restrict__ warped, const float* __restrict__ ref,
                                                              const float* __restrict__ mask, float* __restrict__ coef,
                                                              int B, int H, int W) {
    const Pix q = pixel_of(B, H - 2, W - 2);
    if (!q.ok) return;
    const long long o = (long long)q.b * H * W;
    float c9[9];
    ul::ssim_coefs(warped + o * 3, ref + o * 3, mask + o, q.y + 1, q.x + 1, W, c9);
#pragma unroll
    for (int i = 0; i < 9; ++i) coef[q.p * 9 + i] = c9[i];
}

__global__ __launch_bounds__(UL_BLOCK) void view_bwd_kernel(const float* __restrict__ src, const float* __restrict__ depth,
                                                             const float* __restrict__ cf, const float* __restrict__ warped,
                                                             const float* __restrict__ ref, const float* __restrict__ mask,
                                                             const float* __restrict__ coef, const float* __restrict__ k,
                                                             float* __restrict__ gdepth, int B, int H, int W) {
    const Pix q = pixel_of(B, H, W);
    if (!q.ok) return;
    const long long o = (long long)q.b * H * W;
    const float* img = src + o * 3;
    const ul::Taps t = ul::inv_warp_taps(cf + q.b * 12, q.x, q.y, depth[q.p], H, W);
    const float* cw = coef + (long long)q.b * (H - 2) * (W - 2) * 9;
    float g = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c)
        g += ul::photo_grad(warped + o * 3, ref + o * 3, mask + o, cw, k, q.y, q.x, c, H, W) * ul::tap_ddepth(t, img, c);
    gdepth[q.p] += g;
}

// masked smooth-L1 (losses/aug_loss.py:58-59, losses/sl1loss.py:9-13): sums = { sum sl1(pred - target), count } over mask > 0.5
__global__ __launch_bounds__(UL_BLOCK) void masked_sl1_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                               const float* __restrict__ mask, double* __restrict__ sums, long long n) {
    float part[2] = {0.f, 0.f};
    for (long long i = (long long)blockIdx.x * UL_BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * UL_BLOCK)
        if (mask[i] > 0.5f) { part[0] += ul::sl1(pred[i] - target[i]); part[1] += 1.0f; }
    block_accumulate<2>(part, sums);
}

__global__ __launch_bounds__(UL_BLOCK) void masked_sl1_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                                   const float* __restrict__ mask, const double* __restrict__ sums,
                                                                   const float* __restrict__ gout, float* __restrict__ gpred, long long n) {
    const float k = (float)((double)gout[0] / sums[1]);
    for (long long i = (long long)blockIdx.x * UL_BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * UL_BLOCK)
        gpred[i] = mask[i] > 0.5f ? k * ul::sl1_grad(pred[i] - target[i]) : 0.0f;
}

static inline unsigned blocks_for(long long n) { return (unsigned)((n + UL_BLOCK - 1) / UL_BLOCK); }

}  // namespace rcmvs

using namespace rcmvs;

extern "C" int rcmvs_inverse_warp(const float* src, const float* depth, const float* coef, float* warped, float* mask,
                                  int B, int H, int W, void* stream) {
    RCMVS_REQUIRE(src && depth && coef && warped && mask, "inverse_warp: null pointer");
    RCMVS_REQUIRE(B > 0 && H >= 2 && W >= 2, "inverse_warp: bad dims B=%d H=%d W=%d", B, H, W);
    hipLaunchKernelGGL(inverse_warp_kernel, dim3(blocks_for((long long)B * H * W)), dim3(UL_BLOCK), 0, as_stream(stream),
                       src, depth, coef, warped, mask, B, H, W);
    return launch_status("inverse_warp");
}

extern "C" int rcmvs_unsup_loss_fwd(const float* ref, const float* srcs, const float* depth, const float* coef,
                                    float* warped, float* masks, double* sums, int* counts, float* out,
                                    int B, int Vs, int H, int W, void* stream) {
    RCMVS_REQUIRE(ref && srcs && depth && coef && warped && masks && sums && counts && out, "unsup_loss_fwd: null pointer");
    RCMVS_REQUIRE(B > 0 && H >= 3 && W >= 3, "unsup_loss_fwd: bad dims B=%d H=%d W=%d (SSIM needs 3x3 windows)", B, H, W);
    RCMVS_REQUIRE(Vs >= 1 && Vs <= RCMVS_UNSUP_MAX_VIEWS, "unsup_loss_fwd: %d source views (1..%d)", Vs, RCMVS_UNSUP_MAX_VIEWS);
    hipStream_t st = as_stream(stream);
    const long long n = (long long)B * H * W;
    hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * (Vs * 4 + 2), st);
    if (e == hipSuccess) e = hipMemsetAsync(counts, 0, sizeof(int) * Vs, st);
    if (e != hipSuccess) return fail((int)e, "unsup_loss_fwd: memset: %s", hipGetErrorString(e));
    const dim3 grid(blocks_for(n)), block(UL_BLOCK);
    for (int v = 0; v < Vs; ++v) {
        hipLaunchKernelGGL(inverse_warp_kernel, grid, block, 0, st, srcs + v * n * 3, depth, coef + v * B * 12,
                           warped + v * n * 3, masks + v * n, B, H, W);
        hipLaunchKernelGGL(photo_terms_kernel, grid, block, 0, st, warped + v * n * 3, ref, masks + v * n, sums + v * 4, B, H, W);
    }
    hipLaunchKernelGGL(smooth_terms_kernel, grid, block, 0, st, depth, ref, sums + Vs * 4, B, H, W);
    hipLaunchKernelGGL(best_view_kernel, grid, block, 0, st, masks, sums, counts, B, Vs, H, W);
    hipLaunchKernelGGL(unsup_finalize_kernel, dim3(1), dim3(64), 0, st, sums, counts, out, B, Vs, H, W);
    return launch_status("unsup_loss_fwd");
}

extern "C" int rcmvs_unsup_loss_bwd(const float* ref, const float* srcs, const float* depth, const float* coef,
                                    const float* warped, const float* masks, const int* counts, const float* gout,
                                    float* ssim_ws, float* kbuf, float* gdepth, int B, int Vs, int H, int W, void* stream) {
    RCMVS_REQUIRE(ref && srcs && depth && coef && warped && masks && counts && gout && ssim_ws && kbuf && gdepth,
                  "unsup_loss_bwd: null pointer");
    RCMVS_REQUIRE(B > 0 && H >= 3 && W >= 3, "unsup_loss_bwd: bad dims B=%d H=%d W=%d", B, H, W);
    RCMVS_REQUIRE(Vs >= 1 && Vs <= RCMVS_UNSUP_MAX_VIEWS, "unsup_loss_bwd: %d source views (1..%d)", Vs, RCMVS_UNSUP_MAX_VIEWS);
    hipStream_t st = as_stream(stream);
    const long long n = (long long)B * H * W;
    const dim3 grid(blocks_for(n)), block(UL_BLOCK);
    hipLaunchKernelGGL(unsup_bwd_scalars_kernel, dim3(1), dim3(64), 0, st, gout, counts, kbuf, B, Vs, H, W);
    hipLaunchKernelGGL(smooth_bwd_kernel, grid, block, 0, st, depth, ref, kbuf + Vs * 4, gdepth, B, H, W);
    for (int v = 0; v < Vs; ++v) {
        if (v < 2)
            hipLaunchKernelGGL(ssim_coef_kernel, dim3(blocks_for((long long)B * (H - 2) * (W - 2))), block, 0, st,
                               warped + v * n * 3, ref, masks + v * n, ssim_ws, B, H, W);
        hipLaunchKernelGGL(view_bwd_kernel, grid, block, 0, st, srcs + v * n * 3, depth, coef + v * B * 12, warped + v * n * 3,
                           ref, masks + v * n, ssim_ws, kbuf + v * 4, gdepth, B, H, W);
    }
    return launch_status("unsup_loss_bwd");
}

extern "C" int rcmvs_masked_sl1_fwd(const float* pred, const float* target, const float* mask, double* sums, long long n, void* stream) {
    RCMVS_REQUIRE(pred && target && mask && sums, "masked_sl1_fwd: null pointer");
    RCMVS_REQUIRE(n > 0, "masked_sl1_fwd: n=%lld", n);
    hipStream_t st = as_stream(stream);
    hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * 2, st);
    if (e != hipSuccess) return fail((int)e, "masked_sl1_fwd: memset: %s", hipGetErrorString(e));
    const unsigned nb = blocks_for(n) < 2048u ? blocks_for(n) : 2048u;
    hipLaunchKernelGGL(masked_sl1_kernel, dim3(nb), dim3(UL_BLOCK), 0, st, pred, target, mask, sums, n);
    return launch_status("masked_sl1_fwd");
}

extern "C" int rcmvs_masked_sl1_bwd(const float* pred, const float* target, const float* mask, const double* sums,
                                    const float* gout, float* grad_pred, long long n, void* stream) {
    RCMVS_REQUIRE(pred && target && mask && sums && gout && grad_pred, "masked_sl1_bwd: null pointer");
    RCMVS_REQUIRE(n > 0, "masked_sl1_bwd: n=%lld", n);
    const unsigned nb = blocks_for(n) < 2048u ? blocks_for(n) : 2048u;
    hipLaunchKernelGGL(masked_sl1_bwd_kernel, dim3(nb), dim3(UL_BLOCK), 0, as_stream(stream), pred, target, mask, sums, gout, grad_pred, n);
    return launch_status("masked_sl1_bwd");
}
