// Training-mode BatchNorm (+ReLU, + skip add) around the 3-D convolutions, channels-last.
//
// Conv3d.forward / Deconv3d.forward in train mode (models/modules.py:149-157,196-204) are
// conv -> BatchNorm3d with BATCH statistics -> ReLU, and CostRegNet adds the skip after the block
// (:497-499).  The convolution runs on the kernels of conv3d*.hip with an identity epilogue; this file
// holds the rest, forward and backward:
//   bn_stats          per-channel sum / sum of squares of the conv output        (one read)
//   scale_shift_relu  z = [relu](y * scale[c] + shift[c]) [+ residual]          (one read, one write)
//   bn_bwd_reduce     dbeta[c] = sum g, dgamma[c] = sum g * xhat,  g = dz * [z > 0]
//   bn_bwd_apply      dy = gamma * invstd * (g - dbeta / N - xhat * dgamma / N)
// Tensors are (rows, C) with C a multiple of 4; a lane owns one float4 of channels and walks rows, so
// every access is a coalesced 16-byte vector.  Per-lane fp32 partial sums cover a few dozen rows; they
// are combined in fp64 (LDS tree, then one fp64 atomic per channel and block), which keeps the batch
// statistics of a million-voxel volume accurate to fp32 round-off.  The fp64 sums are what a
// SyncBatchNorm all-reduce exchanges (train_rcmvsnet.py:525), so the multi-GPU path needs no extra kernel.
#include "common.h"
#include <cstdlib>

namespace rcmvs {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int BN_BLOCK = 256;

// combine the block's per-lane float4 partials that belong to the same channel quad, in fp64
template <int NACC>
__device__ __forceinline__ void block_reduce_to_global(const v4f (&part)[NACC], int q, int nq, double* out, int C) {
    __shared__ double red[BN_BLOCK * 4];
    for (int a = 0; a < NACC; ++a) {
        __syncthreads();
        red[threadIdx.x * 4 + 0] = part[a].x; red[threadIdx.x * 4 + 1] = part[a].y;
        red[threadIdx.x * 4 + 2] = part[a].z; red[threadIdx.x * 4 + 3] = part[a].w;
        __syncthreads();
        // threads 0 .. nq*4-1 each own one channel: sum over the lanes t = q, q + nq, q + 2 nq, ...
        if ((int)threadIdx.x < nq * 4) {
            const int qq = threadIdx.x >> 2, comp = threadIdx.x & 3;
            double s = 0.0;
            for (int t = qq; t < BN_BLOCK; t += nq) s += red[t * 4 + comp];
            unsafeAtomicAdd(out + (long long)a * C + qq * 4 + comp, s);
        }
    }
    (void)q;
}

__global__ __launch_bounds__(BN_BLOCK) void bn_stats_kernel(const float* __restrict__ x, double* __restrict__ sums,
                                                            long long rows, int C) {
    const int nq = C / 4;                                  // float4 per row; BN_BLOCK % nq == 0 (checked by the host)
    const int q = threadIdx.x % nq;
    const long long rpb = BN_BLOCK / nq;                   // rows per block iteration
    v4f acc[2] = {(v4f){0.f, 0.f, 0.f, 0.f}, (v4f){0.f, 0.f, 0.f, 0.f}};
    const long long stride = (long long)gridDim.x * rpb;
    long long r = (long long)blockIdx.x * rpb + threadIdx.x / nq;
    for (; r + 3 * stride < rows; r += 4 * stride) {           // four independent loads in flight per lane
        v4f v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const v4f*>(x + (r + k * stride) * C + q * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { acc[0] += v[k]; acc[1] += v[k] * v[k]; }
    }
    for (; r < rows; r += stride) {
        const v4f v = *reinterpret_cast<const v4f*>(x + r * C + q * 4);
        acc[0] += v;
        acc[1] += v * v;
    }
    block_reduce_to_global<2>(acc, q, nq, sums, C);
    if (blockIdx.x == 0 && threadIdx.x == 0) unsafeAtomicAdd(sums + 2 * C, (double)rows);      // the row count travels with the sums (SyncBatchNorm all-reduces 2C + 1 doubles)
}

__global__ __launch_bounds__(BN_BLOCK) void scale_shift_relu_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, const float* __restrict__ res,
                                                                   float* __restrict__ y, long long n4, int C, int relu) {
    const int nq = C / 4;
    for (long long i = (long long)blockIdx.x * BN_BLOCK + threadIdx.x; i < n4; i += (long long)gridDim.x * BN_BLOCK) {
        const int q = (int)(i % nq);
        v4f v = reinterpret_cast<const v4f*>(x)[i];
        if (scale) v = v * reinterpret_cast<const v4f*>(scale)[q] + reinterpret_cast<const v4f*>(shift)[q];
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (res) v += reinterpret_cast<const v4f*>(res)[i];
        reinterpret_cast<v4f*>(y)[i] = v;
    }
}

__device__ __forceinline__ v4f relu_mask(v4f g, v4f z) {
    g.x = z.x > 0.f ? g.x : 0.f; g.y = z.y > 0.f ? g.y : 0.f;
    g.z = z.z > 0.f ? g.z : 0.f; g.w = z.w > 0.f ? g.w : 0.f;
    return g;
}

__global__ __launch_bounds__(BN_BLOCK) void bn_bwd_reduce_kernel(const float* __restrict__ y, const float* __restrict__ dz,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                double* __restrict__ sums, long long rows, int C, int relu) {
    const int nq = C / 4;
    const int q = threadIdx.x % nq;
    const long long rpb = BN_BLOCK / nq;
    const v4f sc = reinterpret_cast<const v4f*>(scale)[q], sh = reinterpret_cast<const v4f*>(shift)[q];
    const v4f mu = reinterpret_cast<const v4f*>(mean)[q], is = reinterpret_cast<const v4f*>(invstd)[q];
    v4f acc[2] = {(v4f){0.f, 0.f, 0.f, 0.f}, (v4f){0.f, 0.f, 0.f, 0.f}};
    const long long stride = (long long)gridDim.x * rpb;
    long long r = (long long)blockIdx.x * rpb + threadIdx.x / nq;
    for (; r + stride < rows; r += 2 * stride) {               // two rows (four loads) in flight per lane
        v4f v[2], g[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            v[k] = *reinterpret_cast<const v4f*>(y + (r + k * stride) * C + q * 4);
            g[k] = *reinterpret_cast<const v4f*>(dz + (r + k * stride) * C + q * 4);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (relu) g[k] = relu_mask(g[k], v[k] * sc + sh);
            acc[0] += g[k];
            acc[1] += g[k] * ((v[k] - mu) * is);
        }
    }
    for (; r < rows; r += stride) {
        const v4f v = *reinterpret_cast<const v4f*>(y + r * C + q * 4);
        v4f g = *reinterpret_cast<const v4f*>(dz + r * C + q * 4);
        if (relu) g = relu_mask(g, v * sc + sh);
        acc[0] += g;
        acc[1] += g * ((v - mu) * is);
    }
    block_reduce_to_global<2>(acc, q, nq, sums, C);
}

// coef = [dbeta / N (C), dgamma / N (C)]
__global__ __launch_bounds__(BN_BLOCK) void bn_bwd_apply_kernel(const float* __restrict__ y, const float* __restrict__ dz,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               const float* __restrict__ coef, float* __restrict__ dy,
                                                               long long n4, int C, int relu) {
    const int nq = C / 4;
    for (long long i = (long long)blockIdx.x * BN_BLOCK + threadIdx.x; i < n4; i += (long long)gridDim.x * BN_BLOCK) {
        const int q = (int)(i % nq);
        const v4f sc = reinterpret_cast<const v4f*>(scale)[q], sh = reinterpret_cast<const v4f*>(shift)[q];
        const v4f mu = reinterpret_cast<const v4f*>(mean)[q], is = reinterpret_cast<const v4f*>(invstd)[q];
        const v4f a = reinterpret_cast<const v4f*>(coef)[q], b = reinterpret_cast<const v4f*>(coef + C)[q];
        const v4f v = reinterpret_cast<const v4f*>(y)[i];
        v4f g = reinterpret_cast<const v4f*>(dz)[i];
        if (relu) g = relu_mask(g, v * sc + sh);
        reinterpret_cast<v4f*>(dy)[i] = sc * (g - a - ((v - mu) * is) * b);
    }
}

// fp64 sums -> everything the block needs, in one launch (replaces a dozen 32-element tensor ops):
// batch mean / biased variance / invstd, the folded scale = gamma*invstd and shift = beta - mean*scale, and the
// running-statistics update of nn.BatchNorm (running_var takes the unbiased variance).
// The sums are CONSUMED: one block reads them, then clears all 2C + 1 doubles, so the caller can keep ONE accumulation buffer per
// BatchNorm module alive across steps instead of zero-filling a fresh one per call (round 3: ~250 fp64 fill launches per training
// iteration); the row count is handed on in *count_out for the backward pass.
__global__ void bn_finalize_kernel(double* __restrict__ sums, double* __restrict__ count_out,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
                                   float* __restrict__ mean, float* __restrict__ var, float* __restrict__ invstd,
                                   float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ running_mean, float* __restrict__ running_var, int C) {
    const double n = sums[2 * C];
    __syncthreads();
    if (threadIdx.x == 0) { count_out[0] = n; sums[2 * C] = 0.0; }
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const double s1 = sums[c], s2 = sums[C + c];
    sums[c] = 0.0; sums[C + c] = 0.0;
    const double m = s1 / n;
    double v = s2 / n - m * m;
    v = v < 0.0 ? 0.0 : v;
    const float is = (float)(1.0 / sqrt(v + (double)eps));
    const float mf = (float)m;
    mean[c] = mf; var[c] = (float)v; invstd[c] = is;
    const float sc = gamma[c] * is;
    scale[c] = sc;
    shift[c] = beta[c] - mf * sc;
    if (running_mean) {
        running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mf;
        running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)(v * (n / (n - 1.0)));
    }
    }
}

// backward: this replica's parameter gradients from its own sums, the normalisation coefficients from the totals
// (the local sums are consumed -- cleared -- like the forward ones; `total` may be the same buffer)
__global__ void bn_bwd_finalize_kernel(double* __restrict__ local, const double* __restrict__ total,
                                       const double* __restrict__ count, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       float* __restrict__ coef, int C) {
    const double n = *count;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const double t0 = total[c], t1 = total[C + c], l0 = local[c], l1 = local[C + c];
        local[c] = 0.0; local[C + c] = 0.0;
        dbeta[c] = (float)l0;
        dgamma[c] = (float)l1;
        coef[c] = (float)(t0 / n);
        coef[C + c] = (float)(t1 / n);
    }
}

// ---- fused forms (round 3): finalize + apply in ONE launch, forward and backward.  A training iteration has 134 BatchNorm layers x
// two passes; the two one-block finalize launches per layer were 268 launches of 3 us that the host -- which issues ~2400 launches per
// iteration and is what the GPU waits for since the weight-gradient flush was fixed -- can do without.  Every thread derives the
// folded scale / shift (backward: the two coefficients) of its own channel quad from the fp64 sums (a few dozen flops, the same
// expressions as the finalize kernels: bit-identical statistics); block 0 also writes the statistics the backward pass needs, the
// running-statistics update and the parameter gradients.  The sums cannot be cleared here (other blocks are still reading them):
// the caller alternates between TWO accumulation buffers per layer and this kernel clears the one the previous call consumed.
__global__ __launch_bounds__(BN_BLOCK) void bn_norm_fwd_kernel(const float* __restrict__ x, const double* __restrict__ sums, double* __restrict__ clear,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
                                                              float* __restrict__ stats, double* __restrict__ count_out,
                                                              float* __restrict__ running_mean, float* __restrict__ running_var,
                                                              const float* __restrict__ res, float* __restrict__ y, long long n4, int C, int relu) {
    const int nq = C / 4;
    const double n = sums[2 * C];
    const int q = threadIdx.x % nq;                        // (gridDim.x * BN_BLOCK) % nq == 0: a thread keeps its channel quad
    v4f sc, sh;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = q * 4 + k;
        const double m = sums[c] / n;
        double v = sums[C + c] / n - m * m;
        v = v < 0.0 ? 0.0 : v;
        const float is = (float)(1.0 / sqrt(v + (double)eps));
        const float mf = (float)m;
        const float s = gamma[c] * is;
        sc[k] = s;
        sh[k] = beta[c] - mf * s;
        if (blockIdx.x == 0 && (int)threadIdx.x < nq) {    // one thread per channel quad records the layer's statistics
            stats[c] = mf; stats[C + c] = (float)v; stats[2 * C + c] = is; stats[3 * C + c] = s; stats[4 * C + c] = sh[k];
            if (running_mean) {
                running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mf;
                running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)(v * (n / (n - 1.0)));
            }
        }
    }
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) count_out[0] = n;
        for (int c = threadIdx.x; c < 2 * C + 1; c += BN_BLOCK) clear[c] = 0.0;
    }
    for (long long i = (long long)blockIdx.x * BN_BLOCK + threadIdx.x; i < n4; i += (long long)gridDim.x * BN_BLOCK) {
        v4f v = reinterpret_cast<const v4f*>(x)[i] * sc + sh;
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (res) v += reinterpret_cast<const v4f*>(res)[i];
        reinterpret_cast<v4f*>(y)[i] = v;
    }
}

__global__ __launch_bounds__(BN_BLOCK) void bn_norm_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dz,
                                                              const float* __restrict__ stats, const double* __restrict__ local,
                                                              const double* __restrict__ total, const double* __restrict__ count, double* __restrict__ clear,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dy,
                                                              long long n4, int C, int relu) {
    const int nq = C / 4;
    const double n = *count;
    const int q = threadIdx.x % nq;
    const v4f mu = reinterpret_cast<const v4f*>(stats)[q], is = reinterpret_cast<const v4f*>(stats + 2 * C)[q];
    const v4f sc = reinterpret_cast<const v4f*>(stats + 3 * C)[q], sh = reinterpret_cast<const v4f*>(stats + 4 * C)[q];
    v4f a, b;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = q * 4 + k;
        a[k] = (float)(total[c] / n);
        b[k] = (float)(total[C + c] / n);
        if (blockIdx.x == 0 && (int)threadIdx.x < nq) { dbeta[c] = (float)local[c]; dgamma[c] = (float)local[C + c]; }
    }
    if (blockIdx.x == 0)
        for (int c = threadIdx.x; c < 2 * C; c += BN_BLOCK) clear[c] = 0.0;
    for (long long i = (long long)blockIdx.x * BN_BLOCK + threadIdx.x; i < n4; i += (long long)gridDim.x * BN_BLOCK) {
        const v4f v = reinterpret_cast<const v4f*>(y)[i];
        v4f g = reinterpret_cast<const v4f*>(dz)[i];
        if (relu) g = relu_mask(g, v * sc + sh);
        reinterpret_cast<v4f*>(dy)[i] = sc * (g - a - ((v - mu) * is) * b);
    }
}

static inline unsigned grid_for(long long work_items) {
    long long g = cdiv(work_items, (long long)BN_BLOCK);
    return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}
// the two reductions end with one fp64 atomic per channel and BLOCK on the same 2C words: same-address atomics serialise (~90 per
// microsecond), so their grid is capped lower than the element-wise kernels'
static inline unsigned grid_for_reduce(long long work_items) {
    constexpr long long cap = 256;      // (sweep: profiles/r3_bn_reduce_grid.txt)
    long long g = cdiv(work_items, (long long)BN_BLOCK);
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace rcmvs

using namespace rcmvs;

extern "C" {

int rcmvs_bn_stats(const float* x, double* sums, long long rows, int C, void* stream) {
    RCMVS_REQUIRE(x && sums && rows > 0, "bn_stats: bad arguments");
    RCMVS_REQUIRE(C >= 4 && C % 4 == 0 && BN_BLOCK % (C / 4) == 0, "bn_stats: C=%d must be 4, 8, 16, 32, 64 ...", C);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(grid_for_reduce(rows * (C / 4) / 8)), dim3(BN_BLOCK), 0, as_stream(stream), x, sums, rows, C);
    return launch_status("bn_stats");
}

int rcmvs_bn_finalize(double* sums, double* count, const float* gamma, const float* beta, float eps, float momentum,
                      float* mean, float* var, float* invstd, float* scale, float* shift,
                      float* running_mean, float* running_var, int C, void* stream) {
    RCMVS_REQUIRE(sums && count && gamma && beta && mean && var && invstd && scale && shift && C > 0, "bn_finalize: bad arguments");
    RCMVS_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize: running_mean and running_var go together");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(64), 0, as_stream(stream), sums, count, gamma, beta, eps, momentum,
                       mean, var, invstd, scale, shift, running_mean, running_var, C);
    return launch_status("bn_finalize");
}

int rcmvs_bn_bwd_finalize(double* local_sums, const double* total_sums, const double* count, float* dgamma, float* dbeta,
                          float* coef, int C, void* stream) {
    RCMVS_REQUIRE(local_sums && total_sums && count && dgamma && dbeta && coef && C > 0, "bn_bwd_finalize: bad arguments");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(1), dim3(64), 0, as_stream(stream), local_sums, total_sums, count,
                       dgamma, dbeta, coef, C);
    return launch_status("bn_bwd_finalize");
}

int rcmvs_scale_shift_relu(const float* x, const float* scale, const float* shift, const float* residual, float* y,
                           long long rows, int C, int relu, void* stream) {
    RCMVS_REQUIRE(x && y && rows > 0, "scale_shift_relu: bad arguments");
    RCMVS_REQUIRE((scale == nullptr) == (shift == nullptr), "scale_shift_relu: scale and shift go together");
    RCMVS_REQUIRE(C >= 4 && C % 4 == 0, "scale_shift_relu: C=%d must be a multiple of 4", C);
    const long long n4 = rows * (C / 4);
    hipLaunchKernelGGL(scale_shift_relu_kernel, dim3(grid_for(n4 / 4)), dim3(BN_BLOCK), 0, as_stream(stream), x, scale, shift, residual, y, n4, C, relu);
    return launch_status("scale_shift_relu");
}

int rcmvs_bn_bwd_reduce(const float* y, const float* dz, const float* scale, const float* shift, const float* mean,
                        const float* invstd, double* sums, long long rows, int C, int relu, void* stream) {
    RCMVS_REQUIRE(y && dz && scale && shift && mean && invstd && sums && rows > 0, "bn_bwd_reduce: bad arguments");
    RCMVS_REQUIRE(C >= 4 && C % 4 == 0 && BN_BLOCK % (C / 4) == 0, "bn_bwd_reduce: C=%d unsupported", C);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(grid_for_reduce(rows * (C / 4) / 8)), dim3(BN_BLOCK), 0, as_stream(stream),
                       y, dz, scale, shift, mean, invstd, sums, rows, C, relu);
    return launch_status("bn_bwd_reduce");
}

int rcmvs_bn_bwd_apply(const float* y, const float* dz, const float* scale, const float* shift, const float* mean,
                       const float* invstd, const float* coef, float* dy, long long rows, int C, int relu, void* stream) {
    RCMVS_REQUIRE(y && dz && scale && shift && mean && invstd && coef && dy && rows > 0, "bn_bwd_apply: bad arguments");
    RCMVS_REQUIRE(C >= 4 && C % 4 == 0, "bn_bwd_apply: C=%d unsupported", C);
    const long long n4 = rows * (C / 4);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(n4 / 4)), dim3(BN_BLOCK), 0, as_stream(stream),
                       y, dz, scale, shift, mean, invstd, coef, dy, n4, C, relu);
    return launch_status("bn_bwd_apply");
}

int rcmvs_bn_norm_fwd(const float* x, const double* sums, double* clear, const float* gamma, const float* beta, float eps, float momentum,
                      float* stats, double* count, float* running_mean, float* running_var, const float* residual, float* y,
                      long long rows, int C, int relu, void* stream) {
    RCMVS_REQUIRE(x && sums && clear && gamma && beta && stats && count && y && rows > 0, "bn_norm_fwd: bad arguments");
    RCMVS_REQUIRE(sums != clear, "bn_norm_fwd: the buffer to clear must be the OTHER accumulation buffer of the layer");
    RCMVS_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_norm_fwd: running_mean and running_var go together");
    RCMVS_REQUIRE(C >= 4 && C % 4 == 0 && BN_BLOCK % (C / 4) == 0, "bn_norm_fwd: C=%d must be 4, 8, 16, 32, 64 ...", C);
    const long long n4 = rows * (C / 4);
    hipLaunchKernelGGL(bn_norm_fwd_kernel, dim3(grid_for(n4 / 4)), dim3(BN_BLOCK), 0, as_stream(stream), x, sums, clear, gamma, beta, eps, momentum,
                       stats, count, running_mean, running_var, residual, y, n4, C, relu);
    return launch_status("bn_norm_fwd");
}

int rcmvs_bn_norm_bwd(const float* y, const float* dz, const float* stats, const double* local_sums, const double* total_sums,
                      const double* count, double* clear, float* dgamma, float* dbeta, float* dy, long long rows, int C, int relu, void* stream) {
    RCMVS_REQUIRE(y && dz && stats && local_sums && total_sums && count && clear && dgamma && dbeta && dy && rows > 0, "bn_norm_bwd: bad arguments");
    RCMVS_REQUIRE(local_sums != clear && total_sums != clear, "bn_norm_bwd: the buffer to clear must be the OTHER accumulation buffer of the layer");
    RCMVS_REQUIRE(C >= 4 && C % 4 == 0 && BN_BLOCK % (C / 4) == 0, "bn_norm_bwd: C=%d unsupported", C);
    const long long n4 = rows * (C / 4);
    hipLaunchKernelGGL(bn_norm_bwd_kernel, dim3(grid_for(n4 / 4)), dim3(BN_BLOCK), 0, as_stream(stream), y, dz, stats, local_sums, total_sums, count,
                       clear, dgamma, dbeta, dy, n4, C, relu);
    return launch_status("bn_norm_bwd");
}

}  // extern "C"
