// micro-benchmark: how many VALU instructions of the SAME wave hide behind its v_mfma_f32_16x16x32_bf16 stream?
// F fillers (independent v_fma_f32 / v_and+v_sub pairs) after every MFMA; 4 or 8 waves per block (1 or 2 per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int F>
__global__ void k(const uint4* a, float* out, int iters, unsigned long long* t) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    bf16x8 av = __builtin_bit_cast(bf16x8, a[threadIdx.x & 63]), bv = __builtin_bit_cast(bf16x8, a[64 + (threadIdx.x & 63)]);
    f32x4 acc[6];
    for (int i = 0; i < 6; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = (float)threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[i], 0, 0, 0);
#pragma unroll
                for (int f = 0; f < F; ++f) { const int j = (i * F + f) & 7; x[j] = __builtin_fmaf(x[j], 1.0001f, 0.5f); }
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    f32x4 s = acc[0]; for (int i = 1; i < 6; ++i) s += acc[i];
    float sx = 0; for (int i = 0; i < 8; ++i) sx += x[i];
    out[blockIdx.x * 512 + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + sx;
    if (threadIdx.x == 0) t[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
}
template <int F>
void run(uint4* a, float* out, unsigned long long* t, int iters, int threads) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<F>, dim3(256), dim3(threads), 0, 0, a, out, iters, t);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<F>, dim3(256), dim3(threads), 0, 0, a, out, iters, t);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; (void)hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("waves/SIMD %d  fillers/MFMA %d: kernel %.1f us, %.2f cycles per MFMA per wave\n", threads / 256, F, ms * 1e3, h / (iters * 12.0));
}
template <int F>
__global__ void k32(const uint4* a, float* out, int iters, unsigned long long* t) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    bf16x8 av = __builtin_bit_cast(bf16x8, a[threadIdx.x & 63]), bv = __builtin_bit_cast(bf16x8, a[64 + (threadIdx.x & 63)]);
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = (float)threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i], 0, 0, 0);
#pragma unroll
                for (int f = 0; f < F; ++f) { const int j = (i * F + f) & 7; x[j] = __builtin_fmaf(x[j], 1.0001f, 0.5f); }
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    f32x16 s = acc[0]; for (int i = 1; i < 4; ++i) s += acc[i];
    float sx = 0; for (int i = 0; i < 8; ++i) sx += x[i];
    for (int e = 0; e < 16; ++e) sx += s[e];
    out[blockIdx.x * 512 + threadIdx.x] = sx;
    if (threadIdx.x == 0) t[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
}
template <int F>
void run32(uint4* a, float* out, unsigned long long* t, int iters, int threads) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k32<F>, dim3(256), dim3(threads), 0, 0, a, out, iters, t);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k32<F>, dim3(256), dim3(threads), 0, 0, a, out, iters, t);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; (void)hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("32x32x16: waves/SIMD %d  fillers/MFMA %d: kernel %.1f us, %.2f cycles per MFMA per wave\n", threads / 256, F, ms * 1e3, h / (iters * 12.0));
}
int main() {
    uint4* a; float* out; unsigned long long* t;
    (void)hipMalloc(&a, 128 * 16); (void)hipMemset(a, 0, 128 * 16); (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&t, 256 * 16);
    const int iters = 2000;
    for (int th : {256, 512}) {
        run<0>(a, out, t, iters, th); run<1>(a, out, t, iters, th); run<2>(a, out, t, iters, th); run<3>(a, out, t, iters, th);
        run<4>(a, out, t, iters, th); run<6>(a, out, t, iters, th);
    }
    for (int th : {256, 512}) { run32<0>(a, out, t, iters, th); run32<2>(a, out, t, iters, th); run32<4>(a, out, t, iters, th); run32<6>(a, out, t, iters, th); run32<8>(a, out, t, iters, th); run32<12>(a, out, t, iters, th); }
    return 0;
}
