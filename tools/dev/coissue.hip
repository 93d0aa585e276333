// micro-benchmark: can a second wave on the same SIMD issue VALU work while the first one issues MFMAs back to back?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// mode bit 2: s_setprio 3 in the VALU waves; bit 3: s_setprio 3 in the MFMA waves
// 512 threads: waves 0-3 (one per SIMD) run MFMA, waves 4-7 run VALU fma chains (mode bit 0: mfma on, bit 1: valu on)
template <int NOP>
__global__ void k(const uint4* a, float* out, int iters, int mode, unsigned long long* t) {
    __shared__ char pad[100 * 1024];
    if (threadIdx.x == 9999) pad[0] = 1;
    const int wave = threadIdx.x >> 6;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (mode & 1) {
            if (mode & 8) __builtin_amdgcn_s_setprio(3);
            bf16x8 av = __builtin_bit_cast(bf16x8, a[threadIdx.x & 63]), bv = __builtin_bit_cast(bf16x8, a[64 + (threadIdx.x & 63)]);
            f32x4 acc[6];
            for (int i = 0; i < 6; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[i], 0, 0, 0);
                        if (NOP == 1) asm volatile("s_nop 1");
                        if (NOP == 2) asm volatile("s_nop 3");
                        if (NOP == 3) asm volatile("s_nop 7");
                        if (NOP == 4) { asm volatile("s_nop 7"); asm volatile("s_nop 1"); }
                        if (NOP == 5) __builtin_amdgcn_s_sleep(1);
                        if (NOP) __builtin_amdgcn_sched_barrier(0);
                    }
            }
            f32x4 s = acc[0]; for (int i = 1; i < 6; ++i) s += acc[i];
            out[blockIdx.x * 512 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
        }
        if (threadIdx.x == 0) t[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - t0;
    } else {
        if (mode & 2) {
            if (mode & 4) __builtin_amdgcn_s_setprio(3);
            float x[8];
            for (int i = 0; i < 8; ++i) x[i] = (float)threadIdx.x + i;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 6; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], 1.0001f, 0.5f);       // 48 independent-ish VALU ops per iteration
            }
            float s = 0; for (int i = 0; i < 8; ++i) s += x[i];
            out[blockIdx.x * 512 + threadIdx.x] = s;
        }
        if (threadIdx.x == 256) t[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memtime() - t0;
    }
}
template <int NOP>
void run(uint4* a, float* out, unsigned long long* t, int iters, int mode) {

        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<NOP>, dim3(256), dim3(512), 0, 0, a, out, iters, mode, t);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<NOP>, dim3(256), dim3(512), 0, 0, a, out, iters, mode, t);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
        printf("nop %d mode %d (%s%s): kernel %.1f us; block 0 MFMA wave %.2f units per MFMA, VALU wave %.2f units per VALU op\n", NOP, mode, (mode & 1) ? "mfma " : "", (mode & 2) ? "valu" : "",
               ms * 1e3, (mode & 1) ? h[0] / (iters * 12.0) : 0.0, (mode & 2) ? h[1] / (iters * 48.0) : 0.0);
    
}
int main() {
    uint4* a; float* out; unsigned long long* t;
    hipMalloc(&a, 128 * 16); hipMemset(a, 0, 128 * 16); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&t, 256 * 16);
    const int iters = 4000;
    run<0>(a, out, t, iters, 1); run<0>(a, out, t, iters, 2); run<0>(a, out, t, iters, 3);
    run<1>(a, out, t, iters, 1); run<1>(a, out, t, iters, 3);
    run<2>(a, out, t, iters, 1); run<2>(a, out, t, iters, 3);
    run<3>(a, out, t, iters, 1); run<3>(a, out, t, iters, 3);
    run<4>(a, out, t, iters, 1); run<4>(a, out, t, iters, 3);
    run<5>(a, out, t, iters, 1); run<5>(a, out, t, iters, 3);
    return 0;
}
