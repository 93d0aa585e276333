// correctness probe: ds_add_f64 when lanes of one instruction share an address.  Every wave adds, ITERS times, a value per lane to
// cell (lane >> SHIFT) of its own row; lanes other than the first of each group add `filler` (0.0 or 1.0).  Expected sums are exact.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(double* out, int shift, double filler, int iters) {
    __shared__ double buf[4 * 64];
    buf[threadIdx.x] = 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool first = (lane & ((1 << shift) - 1)) == 0;
    for (int it = 0; it < iters; ++it) unsafeAtomicAdd(buf + wv * 64 + (lane >> shift), first ? 0.37 + lane : filler);
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = buf[threadIdx.x];
}
int main() {
    double* out; (void)hipMalloc(&out, 64 * 256 * 8);
    static double host[64 * 256];
    for (int shift = 0; shift <= 6; ++shift)
        for (int f = 0; f < 2; ++f) {
            const double filler = f ? 1.0 : 0.0;
            const int iters = 100;
            hipLaunchKernelGGL(k, dim3(64), dim3(256), 0, 0, out, shift, filler, iters);
            (void)hipMemcpy(host, out, sizeof(host), hipMemcpyDeviceToHost);
            int bad = 0; double worst = 0.0;
            for (int b = 0; b < 64; ++b) for (int t = 0; t < 256; ++t) {
                const int cell = t & 63;
                double want = 0.0;
                if (cell < (64 >> shift)) want = iters * ((0.37 + (cell << shift)) + filler * ((1 << shift) - 1));
                const double got = host[b * 256 + t];
                const double err = got > want ? got - want : want - got;
                if (err > 1e-9 * (want > 1 ? want : 1)) { ++bad; if (err > worst) worst = err; }
            }
            printf("lanes per address %2d, filler %.0f: %d wrong cells of %d (worst error %.3g)\n", 1 << shift, filler, bad, 64 * 256, worst);
        }
    return 0;
}
