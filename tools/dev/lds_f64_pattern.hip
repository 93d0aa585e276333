// micro-benchmark: ds_add_f64 in the access pattern of the K1-backward window kernel (csrc/warp_variance_bwd.hip): lane = (pixel p, channel quad q),
// 16 adds per item at cells c0 + j * CHS + {0, 1, WX, WX + 1}, c0 = q * 4 * CHS + row * WX + p.
//  mode 0: as the kernel issues them (j outer, the four taps inner: the east tap of lane p is the west tap lane p + 1 hit one instruction earlier)
//  mode 1: taps outer, j inner (four instructions between the two)
//  mode 2: pixels two cells apart (no cell shared between neighbouring lanes)
//  mode 3: 64 consecutive doubles per instruction (the reference rate)
//  mode 4: mode 0 with the four waves of a block in the same window rows (adds of different waves meet in the same cells)
//  mode 5: mode 0 with an uneven tap sequence (a duplicate every 11 pixels, the second half of the row one window row down)
//  mode 6: mode 0 with fractional values that change every iteration
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int WX = 40, CHS = 8 * 40 + 4;
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
    __shared__ double buf[3 * 8 * CHS];
    for (int i = threadIdx.x; i < 3 * 8 * CHS; i += 256) buf[i] = 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int p = lane >> 1, q = lane & 1;
    const double v0 = (double)(lane + 1);
    for (int it = 0; it < iters; ++it) {
        const int va = it % 3, row = MODE == 4 ? it % 6 : (wv + it) % 6;
        const int pp = MODE == 5 ? p - p / 11 + (p >= 16 ? WX : 0) : p;
        const double v = MODE == 6 ? 0.37 * lane + 1e-3 * it : v0;
        double* c0 = buf + va * 8 * CHS + (MODE == 3 ? 0 : (q * 4) * CHS + row * WX + (MODE == 2 ? (p & 15) * 2 : pp));
        if (MODE == 0 || MODE >= 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { unsafeAtomicAdd(c0 + j * CHS, v); unsafeAtomicAdd(c0 + j * CHS + 1, v); unsafeAtomicAdd(c0 + j * CHS + WX, v); unsafeAtomicAdd(c0 + j * CHS + WX + 1, v); }
        } else if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) unsafeAtomicAdd(c0 + j * CHS + (t & 1) + (t >> 1) * WX, v);
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) unsafeAtomicAdd(c0 + j * 64 + lane, v);
        }
    }
    __syncthreads();
    double acc = 0.0;
    for (int i = threadIdx.x; i < 3 * 8 * CHS; i += 256) acc += buf[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int MODE>
void run(double* out) {
    const int iters = 500, blocks = 512;                    // two blocks per CU, as the kernel
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_cu = (blocks / 256.0) * 4.0 * iters * 16;
    printf("mode %d: %.3f ms, %.1f clocks per wave instruction per CU at 2.4 GHz\n", MODE, ms, ms * 1e6 / instr_per_cu * 2.4);
}
int main() {
    double* out; (void)hipMalloc(&out, 512 * 256 * 8);
    run<0>(out); run<1>(out); run<2>(out); run<3>(out); run<4>(out); run<5>(out); run<6>(out);
    return 0;
}
