// micro-benchmark 2: does out-of-place accumulation (vDst != SrcC) or vDst == SrcB cost MFMA issue rate?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF_INPLACE(ACC, A, B) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))
#define MF_OOP(D, A, B, C) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(D) : "v"(A), "v"(B), "v"(C))
#define MF_DB(DB, A, C) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %0, %2" : "+v"(DB) : "v"(A), "v"(C))
template <int MODE>
__global__ void k(const uint4* a, f32x4* out, int iters) {
    __shared__ char pad[64 * 1024];
    if (threadIdx.x == 9999) pad[0] = 1;
    bf16x8 av[6], bv;
#pragma unroll
    for (int i = 0; i < 6; ++i) av[i] = __builtin_bit_cast(bf16x8, a[(threadIdx.x + i) & 127]);
    bv = __builtin_bit_cast(bf16x8, a[64 + (threadIdx.x & 63)]);
    f32x4 x[6], y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { x[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; y[i] = x[i]; }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {            // in place, 6 chains, distinct A registers
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 6; ++i) MF_INPLACE(x[i], av[i], bv);
        } else if (MODE == 1) {     // out of place: x -> y -> x
#pragma unroll
            for (int i = 0; i < 6; ++i) MF_OOP(y[i], av[i], bv, x[i]);
#pragma unroll
            for (int i = 0; i < 6; ++i) MF_OOP(x[i], av[i], bv, y[i]);
        } else {                    // vDst == SrcB, SrcC elsewhere (the pattern the compiler produced)
#pragma unroll
            for (int i = 0; i < 6; ++i) { y[i] = x[i]; }
#pragma unroll
            for (int i = 0; i < 6; ++i) MF_DB(y[i], av[i], x[(i + 3) % 6]);
#pragma unroll
            for (int i = 0; i < 6; ++i) MF_DB(x[i], av[i], y[(i + 3) % 6]);
        }
    }
    asm volatile("s_nop 15\n s_nop 15");
    f32x4 s = x[0] + y[0];
#pragma unroll
    for (int i = 1; i < 6; ++i) s += x[i] + y[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, const uint4* a, f32x4* out) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, a, out, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, a, out, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %.2f ns per MFMA per SIMD\n", name, ms * 1e6 / (iters * 12.0));
}
int main() {
    uint4* a; f32x4* out;
    hipMalloc(&a, 128 * 16); hipMemset(a, 0, 128 * 16); hipMalloc(&out, 256 * 1024 * 16);
    run<0>("in place (vDst == SrcC)", a, out);
    run<1>("out of place (vDst != SrcC)", a, out);
    run<2>("vDst == SrcB, SrcC elsewhere", a, out);
    return 0;
}
