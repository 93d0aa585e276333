// micro-benchmark: fp32 global atomic add throughput on MI355X as a function of the access pattern of a wave instruction.
//  mode 0: lanes -> 64 consecutive floats (256 B, 4 lines per instruction)
//  mode 1: the K1-backward pattern: lane (pixel p, quad q) adds component j of its float4 at ((p * LPP + q) * 4 + j): four
//          instructions walk j, each touching every 16-byte slot of the same lines (lanes 16 B apart)
//  mode 2: mode 0 but every lane pair hits the same address (in-wave duplicates)
//  mode 3: random texel (32 B) per lane pair, components as mode 1 (scattered lines)
// Each thread issues N atomics into a buffer of `span` bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k(float* buf, int mode, int iters, unsigned span_floats) {
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned wave = gid >> 6, lane = gid & 63;
    unsigned s = gid * 2654435761u + 12345u;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        const unsigned wbase = ((wave * 977u + it * 131u) * 256u) % span_floats;      // a wave-private-ish 1 KB region per step
        if (mode == 0) {
            for (int j = 0; j < 4; ++j) unsafeAtomicAdd(buf + (wbase + j * 64 + lane) % span_floats, 1.0f);
        } else if (mode == 1) {
            for (int j = 0; j < 4; ++j) unsafeAtomicAdd(buf + (wbase + lane * 4 + j) % span_floats, 1.0f);
        } else if (mode == 2) {
            for (int j = 0; j < 4; ++j) unsafeAtomicAdd(buf + (wbase + j * 64 + (lane >> 1)) % span_floats, 1.0f);
        } else {
            const unsigned tex = ((s >> 8) % (span_floats / 8)) * 8;          // texel chosen per lane pair
            const unsigned t2 = __shfl(tex, lane & ~1u);
            for (int j = 0; j < 4; ++j) unsafeAtomicAdd(buf + t2 + (lane & 1) * 4 + j, 1.0f);
        }
    }
}
int main() {
    const unsigned span = 64u << 20;         // 64 MB target (inside the 256 MB MALL, beyond L2)
    float* buf; (void)hipMalloc(&buf, span); (void)hipMemset(buf, 0, span);
    const int iters = 64, blocks = 256 * 8, threads = 256;
    for (int mode = 0; mode < 4; ++mode) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, buf, mode, iters, span / 4);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, buf, mode, iters, span / 4);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)blocks * threads * iters * 4;
        printf("mode %d: %.3f ms, %.1f G atomic dwords/s, %.1f G wave-instructions/s\n", mode, ms, n / ms / 1e6, n / 64 / ms / 1e6);
    }
    return 0;
}
