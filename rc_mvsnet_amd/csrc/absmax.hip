// Bound of max|x| of a tensor in the slot format the split-operand convolution kernels take as `x_absmax` (conv3d_x3.hip: 64 slots,
// 16 floats apart, the bound is their maximum; one atomic max per block into slot (block & 63)).  Used where the producer of a
// tensor does not maintain the bound itself: the feature maps in front of the warp + variance kernel -- var = E[f^2] - E[f]^2 <=
// max f^2, so `square` publishes the squared maximum as the bound of the variance volume (models/casmvsnet.py:288).
#include "common.h"

namespace rcmvs {

__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long long n4, long long n, int square, float* __restrict__ amax) {
    __shared__ float red[4];
    float m = 0.0f;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (blockIdx.x == 0 && threadIdx.x < (unsigned)(n - 4 * n4)) m = fmaxf(m, fabsf(x[4 * n4 + threadIdx.x]));      // tail (n not a multiple of 4)
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) m = fmaxf(m, __shfl_xor(m, k));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (square) m = m * m;
        atomicMax(reinterpret_cast<unsigned int*>(amax) + (blockIdx.x & 63) * 16, __float_as_uint(m));
    }
}

}  // namespace rcmvs

using namespace rcmvs;

extern "C" int rcmvs_absmax_fwd(const float* x, long long n, int square, float* amax, void* stream) {
    RCMVS_REQUIRE(x && amax && n > 0, "absmax_fwd: bad arguments");
    const long long n4 = n / 4;
    const long long want = (n4 + 255) / 256;
    const unsigned grid = (unsigned)(want < 1 ? 1 : (want > 1024 ? 1024 : want));
    hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(256), 0, as_stream(stream), x, n4, n, square, amax);
    return launch_status("absmax_fwd");
}
