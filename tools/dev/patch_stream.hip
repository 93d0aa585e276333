// Developer probe (GPU box): what does HBM deliver to 256 persistent blocks that march over z reading a small (TY+2) x (TX+2) patch of every plane
// (the access pattern of the z-streaming conv kernels: rows of (TX+2) * C * 4 contiguous bytes, one row per image row, one patch per plane) and
// writing a TY x TX patch -- against the same bytes laid out tile-major (a tick's patch contiguous)?   hipcc --offload-arch=gfx950 -O3 patch_stream.hip -o patch_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
struct P { int D, H, W, C, TY, TX, tiles_x, ntiles, mode, qd, wr; };   // mode 0: patch rows, 1: tile-major contiguous
template <int QD>
__global__ __launch_bounds__(512) void k(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ sink, P p) {
    const int tid = threadIdx.x, nblk = gridDim.x, bid = blockIdx.x;
    const int r = ((nblk & 7) == 0) ? (bid & 7) * (nblk >> 3) + (bid >> 3) : bid;
    const long long T = (long long)p.ntiles * p.D;
    const long long lo = T * r / nblk, hi = T * (r + 1) / nblk;
    const int Q4 = p.C / 4, TYP = p.TY + 2, TXP = p.TX + 2, NE = TYP * TXP * Q4, NO = p.TY * p.TX * (8 / 4);
    f4 q[QD][6];
    f4 acc = {0, 0, 0, 0};
    auto fetch = [&](f4 (&d)[6], long long s) {
        if (s >= hi) return;
        const int tile = (int)(s / p.D), z = (int)(s % p.D);
        const int y0 = (tile / p.tiles_x) * p.TY, x0 = (tile % p.tiles_x) * p.TX;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int e = tid + i * 512;
            if (e >= NE) break;
            long long off;
            if (p.mode == 0) {
                const int v = e / Q4, c4 = e % Q4, hy = v / TXP, hx = v % TXP;
                int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
                gy = gy < 0 ? 0 : (gy >= p.H ? p.H - 1 : gy); gx = gx < 0 ? 0 : (gx >= p.W ? p.W - 1 : gx);
                off = ((((long long)z * p.H + gy) * p.W + gx) * p.C + c4 * 4);
            } else off = ((long long)s * NE + e) * 4 % ((long long)p.D * p.H * p.W * p.C - 4);
            d[i] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(x + off));
        }
    };
#pragma unroll
    for (int j = 0; j < QD; ++j) fetch(q[j], lo + j);
    for (long long s = lo; s < hi; s += QD) {
#pragma unroll
        for (int j = 0; j < QD; ++j) {
            if (s + j >= hi) break;
#pragma unroll
            for (int i = 0; i < 6; ++i) if (tid + i * 512 < NE) acc += q[j][i];
            fetch(q[j], s + j + QD);
            if (p.wr && tid < NO) {
                const long long ss = s + j;
                const int tile = (int)(ss / p.D), z = (int)(ss % p.D);
                long long off;
                if (p.mode == 0) {
                    const int v = tid / 2, c4 = tid % 2, oy = (tile / p.tiles_x) * p.TY + v / p.TX, ox = (tile % p.tiles_x) * p.TX + v % p.TX;
                    off = (((long long)z * p.H + (oy < p.H ? oy : p.H - 1)) * p.W + (ox < p.W ? ox : p.W - 1)) * 8 + c4 * 4;
                } else off = (ss * NO + tid) * 4;
                *reinterpret_cast<f4*>(y + off) = acc;
            }
        }
    }
    if (acc.x == 12345.678f) sink[bid] = acc.y;
}
int main(int argc, char** argv) {
    struct { int D, H, W, C, TY, TX; const char* name; } shapes[] = {{48, 128, 160, 32, 8, 32, "S1 32ch"}, {32, 256, 320, 16, 8, 32, "S2 16ch"}, {8, 512, 640, 8, 16, 32, "S3 8ch"}, {32, 256, 320, 16, 16, 64, "S2 16ch big tile"}};
    for (auto& sh : shapes) {
        P p; p.D = sh.D; p.H = sh.H; p.W = sh.W; p.C = sh.C; p.TY = sh.TY; p.TX = sh.TX;
        p.tiles_x = (p.W + p.TX - 1) / p.TX; p.ntiles = p.tiles_x * ((p.H + p.TY - 1) / p.TY);
        const size_t nin = (size_t)p.D * p.H * p.W * p.C, nout = (size_t)p.D * p.H * p.W * 8;
        if ((p.TY + 2) * (p.TX + 2) * p.C / 4 > 6 * 512) { printf("%s: patch too large\n", sh.name); continue; }
        float *x, *y, *sink;
        hipMalloc(&x, nin * 4 + 4096); hipMalloc(&y, nout * 4 + ((size_t)64 << 20)); hipMalloc(&sink, 4096);
        hipMemset(x, 0, nin * 4); hipMemset(y, 0, nout * 4);
        for (int mode = 0; mode < 2; ++mode) for (int wr = 0; wr < 2; ++wr) for (int qd = 2; qd <= 4; qd += 2) {
            p.mode = mode; p.wr = wr; p.qd = qd;
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            auto run = [&]() { if (qd == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, x, y, sink, p); else hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, x, y, sink, p); };
            for (int i = 0; i < 3; ++i) run();
            hipEventRecord(a); for (int i = 0; i < 20; ++i) run(); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double us = ms * 1e3 / 20;
            const double rd = (double)p.ntiles * p.D * (p.TY + 2) * (p.TX + 2) * p.C * 4, wrb = wr ? (double)nout * 4 : 0, alg = (double)nin * 4 + wrb;
            printf("%-18s mode %d (%s) wr %d qd %d: %7.1f us   requested %6.1f MB -> %5.2f TB/s   algorithmic %6.1f MB -> %5.2f TB/s\n", sh.name, mode, mode ? "tile-major" : "patch rows", wr, qd, us,
                   (rd + wrb) / 1e6, (rd + wrb) / us * 1e-6, alg / 1e6, alg / us * 1e-6);
        }
        hipFree(x); hipFree(y); hipFree(sink);
    }
    return 0;
}
