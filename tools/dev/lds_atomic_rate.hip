// micro-benchmark: LDS atomic throughput on MI355X (clocks per wave instruction, per CU) against plain LDS traffic.
//  mode 0: ds_add_f32, lanes -> 64 consecutive floats (conflict-free)
//  mode 1: ds_add_f32, lanes 4 floats apart (4-way bank conflict)
//  mode 2: ds_write_b32, consecutive (no atomic)
//  mode 3: ds_read_b32 + add + ds_write_b32, consecutive (what a non-atomic accumulation costs)
//  mode 4: ds_add_f32, the 16 lanes of a row on one address
//  mode 5: ds_add_u32, consecutive          mode 6: ds_add_u64, consecutive        mode 7: ds_add_u32, 16 lanes of a row on one address
//  mode 8: ds_add_f64, consecutive          mode 9: ds_add_rtn_f32 (result used)
// 1024 blocks x 256 threads (four blocks resident per CU), each thread issues ITERS x 16 operations.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float buf[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) buf[i] = 0.0f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* base = buf + wv * 2048;
    unsigned* ub = reinterpret_cast<unsigned*>(base);
    unsigned long long* lb = reinterpret_cast<unsigned long long*>(base);
    float acc = 0.0f;
    const float v = (float)(lane + 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE == 0) unsafeAtomicAdd(base + j * 64 + lane, v);
            else if (MODE == 1) unsafeAtomicAdd(base + ((lane * 4 + j) & 2047), v);
            else if (MODE == 2) { base[j * 64 + lane] = v; __builtin_amdgcn_sched_barrier(0); }
            else if (MODE == 3) base[j * 64 + lane] += v;
            else if (MODE == 4) unsafeAtomicAdd(base + j * 64 + (lane >> 4), v);
            else if (MODE == 5) atomicAdd(ub + j * 64 + lane, (unsigned)lane);
            else if (MODE == 6) atomicAdd(lb + j * 64 + lane, (unsigned long long)lane);
            else if (MODE == 7) atomicAdd(ub + j * 64 + (lane >> 4), (unsigned)lane);
            else if (MODE == 8) unsafeAtomicAdd(reinterpret_cast<double*>(base) + j * 64 + lane, (double)v);
            else acc += unsafeAtomicAdd(base + j * 64 + lane, v);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 8192; i += 256) acc += buf[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int MODE>
void run(float* out) {
    const int iters = 500, blocks = 1024;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_cu = (blocks / 256.0) * 4.0 * iters * 16;            // wave instructions one CU executed
    printf("mode %d: %.3f ms, %.2f ns per wave instruction per CU (= %.1f clocks at 2.4 GHz)\n", MODE, ms, ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.4);
}
int main() {
    float* out; (void)hipMalloc(&out, 1024 * 256 * 4);
    run<0>(out); run<1>(out); run<2>(out); run<3>(out); run<4>(out); run<5>(out); run<6>(out); run<7>(out); run<8>(out); run<9>(out);
    return 0;
}
