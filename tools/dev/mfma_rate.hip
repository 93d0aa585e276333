// micro-benchmark: issue rate of v_mfma_f32_16x16x32_bf16 from 1 / 2 / 3 waves per SIMD, with 2 / 4 / 6 / 12 independent accumulators
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k(const uint4* a, f32x4* out, int iters) {
    __shared__ char pad[64 * 1024];
    if (threadIdx.x == 9999) pad[0] = 1;
    bf16x8 av = __builtin_bit_cast(bf16x8, a[threadIdx.x & 63]), bv = __builtin_bit_cast(bf16x8, a[64 + (threadIdx.x & 63)]);
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 12 / NACC; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[i], 0, 0, 0);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int threads, const uint4* a, f32x4* out) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, a, out, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, a, out, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf_per_simd = (double)iters * 12 * (threads / 256.0);
    printf("waves/SIMD %d  independent accumulators %2d: %.1f ns per MFMA per SIMD (%.1f cycles @2.4GHz) -> %.0f TF\n", threads / 256, NACC,
           ms * 1e6 / mf_per_simd, ms * 1e6 / mf_per_simd * 2.4, 256.0 * 4 * mf_per_simd * 16384 / (ms * 1e-3) / 1e12);
}
int main() {
    uint4* a; f32x4* out;
    hipMalloc(&a, 128 * 16); hipMemset(a, 0, 128 * 16); hipMalloc(&out, 256 * 1024 * 16);
    for (int th : {256, 512, 768}) { run<2>(th, a, out); run<4>(th, a, out); run<6>(th, a, out); run<12>(th, a, out); }
    return 0;
}
