// VALU issue-rate microbench (gfx950): cycles per wave-instruction of packed / plain fp32 ops, independent chains, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v2f __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int OP>
__global__ __launch_bounds__(1024) void rate_kernel(float* out, long long* cyc, int iters) {
    __shared__ float ldsbuf[4096]; ldsbuf[threadIdx.x] = 1.0f;
    v2f a[8];
    float s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (v2f){(float)threadIdx.x * 1e-3f + i, 1.0f + i}; s[i] = (float)threadIdx.x * 1e-3f + i; }
    v2f m = (v2f){1.0001f, 0.9999f}, c = (v2f){1e-6f, -1e-6f};
    float ms = 1.0001f, cs = 1e-6f;
    unsigned long long mask = 0x5555555555555555ull; int addr = (threadIdx.x & 63) * 16; typedef float v4f_ __attribute__((ext_vector_type(4))); v4f_ q[2] = {};
    asm volatile("s_mov_b64 vcc, 0x55" ::: "vcc");
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) { asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(s[i]) : "v"(s[(i+1)&7]), "v"(ms) : "vcc"); }
                if (OP == 1) { asm volatile("v_cmp_lt_f32_e64 %3, %1, %2\n v_cndmask_b32_e64 %0, %0, %2, %3" : "+v"(s[i]) : "v"(s[(i+1)&7]), "v"(ms), "s"(mask)); }
                if (OP == 2) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(s[i]) : "v"(ms)); }
                if (OP == 3) { asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(s[i]) : "v"(ms)); }
                if (OP == 4) { asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(ms), "v"(cs)); }
                if (OP == 5) { asm volatile("v_or_b32 %0, %0, %1" : "+v"(s[i]) : "v"(ms)); }
                if (OP == 6) { asm volatile("v_sub_u32 %0, %0, %1" : "+v"(s[i]) : "v"(ms)); }
                if (OP == 7) { asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(a[i]) : "v"(m)); }
                if (OP == 8) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(m)); }
                if (OP == 9) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "s"(mask)); }
                if (OP == 10) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i]) : "v"(s[(i + 1) & 7]), "s"(addr)); }
                if (OP == 11) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i]) : "v"(s[(i + 1) & 7]), "v"(ms)); }
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += a[i].x + a[i].y + s[i] + q[i & 1].x + (float)(mask & 1);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    float* out; long long* cyc;
    CK(hipMalloc(&out, 256 * 1024 * 4)); CK(hipMalloc(&cyc, 8));
    const char* names[] = {"cmp vcc + cndmask vcc", "cmp_e64 sgpr + cndmask_e64", "cndmask vcc (vcc set by s_mov first)", "cndmask_e64 vcc explicit", "v_bfi_b32", "v_or_b32", "v_sub_u32", "v_pk_add_f32 neg", "v_pk_fma_f32 v,v,v", "v_pk_fma_f32 v,SGPR pair,v", "v_fma_f32 v,SGPR,v", "v_fma_f32 v,v,v"};
    const int iters = 2000;
    for (int op = 0; op < 12; ++op)
        for (int wps = 1; wps <= 4; wps += (wps == 1 ? 2 : 1)) {
            const int threads = 256 * wps;      // one block per CU, wps waves per SIMD
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            auto launch = [&]() {
                switch (op) {
#define L(O) case O: hipLaunchKernelGGL(rate_kernel<O>, dim3(256), dim3(threads), 0, 0, out, cyc, iters); break;
                    L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11)
                }
            };
            launch(); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            long long hc; CK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
            const double ninst = (double)iters * 32;     // per wave
            // s_memtime ticks at 100 MHz: use the event time and an assumed 2.4 GHz instead
            printf("%-26s waves/SIMD %d: %7.1f us  -> %5.2f clk (2.4 GHz) per wave-instruction per SIMD slot; memtime ticks %lld\n", names[op], wps, ms * 1e3,
                   ms * 1e-3 * 2.4e9 / (ninst * wps), hc);
        }
    return 0;
}
