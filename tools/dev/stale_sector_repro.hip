// Developer probe (GPU box, standalone: hipcc --offload-arch=gfx950 -O3 -o stale_sector_repro stale_sector_repro.hip):
// an attempt at a MINIMAL reproducer of round 3's two-stream defect (profiles/r3_two_streams.txt (i)-(k)) outside the pipeline.
// What the pipeline did at the spot that went wrong, per scene and per stream:
//   ... heavy kernels ... -> W: a small map (320 KB) written by every other lane (a softmax/regress head, 2 lanes per pixel), reading a
//   10 MB volume -> X: an unrelated 30 MB kernel -> R: a kernel that reads the map with 4 bilinear taps per output pixel -> heavy kernels
// and the map's allocator block held OTHER small values the scene before (the head's second output; the two blocks swap roles).
// R checks every tap against the value W wrote in this scene and counts the taps that saw anything else.
//   mode 0: two streams, plain loads   mode 1: ONE stream (same scenes, same order)   mode 2: two streams, agent-scope (sc1) loads in R
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int HP = 256, WP = 320, HW = HP * WP, D = 32;

__global__ __launch_bounds__(256) void head_kernel(const float* __restrict__ vol, float* __restrict__ depth, float* __restrict__ conf, float value) {
    const int j = threadIdx.x % 2;
    const long long p = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / 2;
    if (p >= HW) return;
    float s = 0.f;
    for (int k = j; k < D; k += 2) s += vol[(long long)k * HW + p];
    s += __shfl_xor(s, 1);
    if (j == 0) {
        depth[p] = value + 0.0f * s;
        conf[p] = 0.25f + 0.0f * s;
    }
}

__global__ __launch_bounds__(256) void stream_kernel(const float4* __restrict__ a, float4* __restrict__ b, long long n, float add) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float4 v = a[i];
        v.x += add; v.y += add; v.z += add; v.w += add;
        b[i] = v;
    }
}

template <bool AGENT>
__device__ __forceinline__ float ld(const float* p) {
    if constexpr (AGENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}

template <bool AGENT>
__global__ __launch_bounds__(256) void planes_kernel(const float* __restrict__ prev, float2* __restrict__ planes, float want, unsigned* bad, unsigned* first) {
    const int H = 2 * HP, W = 2 * WP;
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= H * W) return;
    int y = t / W, x = t % W;
    float sy = 0.5f * ((float)y + 0.5f) - 0.5f, sx = 0.5f * ((float)x + 0.5f) - 0.5f;
    sy = sy < 0.f ? 0.f : sy; sx = sx < 0.f ? 0.f : sx;
    int y0 = (int)sy, x0 = (int)sx;
    y0 = y0 > HP - 1 ? HP - 1 : y0; x0 = x0 > WP - 1 ? WP - 1 : x0;
    int y1 = y0 + 1 > HP - 1 ? HP - 1 : y0 + 1, x1 = x0 + 1 > WP - 1 ? WP - 1 : x0 + 1;
    float a = ld<AGENT>(prev + y0 * WP + x0), b = ld<AGENT>(prev + y0 * WP + x1), c = ld<AGENT>(prev + y1 * WP + x0), d = ld<AGENT>(prev + y1 * WP + x1);
    int nb = (a != want) + (b != want) + (c != want) + (d != want);
    if (nb) {
        unsigned k = atomicAdd(bad, (unsigned)nb);
        if (k == 0) { first[0] = (unsigned)t; first[1] = __float_as_uint(a != want ? a : (b != want ? b : (c != want ? c : d))); }
    }
    planes[t] = make_float2(a + b - c - d + want, 1.0f);
}

struct Slot {
    hipStream_t st;
    float *vol, *small[2], *big[4];
    float2* planes;
    unsigned *bad, *first;
};

int main(int argc, char** argv) {
    const int scenes = argc > 1 ? atoi(argv[1]) : 400;
    const long long BIG = 80ll << 20;          // the 80 MiB volumes of stages 2 / 3
    for (int mode = 0; mode < 3; ++mode) {
        Slot s[2];
        for (int k = 0; k < 2; ++k) {
            CK(hipStreamCreateWithFlags(&s[k].st, hipStreamNonBlocking));
            CK(hipMalloc(&s[k].vol, sizeof(float) * D * HW));
            CK(hipMemset(s[k].vol, 0, sizeof(float) * D * HW));
            for (int i = 0; i < 2; ++i) { CK(hipMalloc(&s[k].small[i], sizeof(float) * HW)); CK(hipMemset(s[k].small[i], 0, sizeof(float) * HW)); }
            for (int i = 0; i < 4; ++i) { CK(hipMalloc(&s[k].big[i], BIG)); CK(hipMemset(s[k].big[i], 0, BIG)); }
            CK(hipMalloc(&s[k].planes, sizeof(float2) * 4 * HW));
            CK(hipMalloc(&s[k].bad, 8)); CK(hipMemset(s[k].bad, 0, 8));
            CK(hipMalloc(&s[k].first, 8)); CK(hipMemset(s[k].first, 0, 8));
        }
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        const long long n4 = BIG / 16;
        for (int i = 0; i < scenes; ++i) {
            Slot& q = s[i % 2];
            hipStream_t st = mode == 1 ? s[0].st : q.st;
            const int sc = i / 2;                                   // this slot's scene number
            float* depth = q.small[sc % 2]; float* conf = q.small[(sc + 1) % 2];       // the two blocks swap roles every scene
            const float value = 500.0f + (float)(sc % 97);
            // "stage 2": two big streaming kernels, the head, an unrelated mid-size kernel, the reader, "stage 3": four big kernels
            hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, st, (const float4*)q.big[0], (float4*)q.big[1], n4, 1.0f);
            hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, st, (const float4*)q.big[1], (float4*)q.big[2], n4, 1.0f);
            hipLaunchKernelGGL(head_kernel, dim3((HW * 2 + 255) / 256), dim3(256), 0, st, q.vol, depth, conf, value);
            hipLaunchKernelGGL(stream_kernel, dim3(1024), dim3(256), 0, st, (const float4*)q.big[2], (float4*)q.big[3], n4 / 3, 1.0f);
            if (mode == 2) hipLaunchKernelGGL(planes_kernel<true>, dim3((4 * HW + 255) / 256), dim3(256), 0, st, depth, q.planes, value, q.bad, q.first);
            else           hipLaunchKernelGGL(planes_kernel<false>, dim3((4 * HW + 255) / 256), dim3(256), 0, st, depth, q.planes, value, q.bad, q.first);
            for (int r = 0; r < 4; ++r)
                hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, st, (const float4*)q.big[r % 4], (float4*)q.big[(r + 1) % 4], n4, 1.0f);
        }
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned bad[2], first[2][2];
        for (int k = 0; k < 2; ++k) { CK(hipMemcpy(&bad[k], s[k].bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(first[k], s[k].first, 8, hipMemcpyDeviceToHost)); }
        float f0, f1; memcpy(&f0, &first[0][1], 4); memcpy(&f1, &first[1][1], 4);
        printf("mode %d (%s): %d scenes, %.3f ms per scene; taps that saw a value other than this scene's: slot 0: %u (first at output pixel %u, saw %g), slot 1: %u (pixel %u, saw %g)\n",
               mode, mode == 0 ? "two streams, plain loads" : (mode == 1 ? "one stream" : "two streams, agent-scope loads"), scenes, ms / scenes,
               bad[0], first[0][0], f0, bad[1], first[1][0], f1);
        for (int k = 0; k < 2; ++k) {
            CK(hipStreamDestroy(s[k].st)); CK(hipFree(s[k].vol)); CK(hipFree(s[k].planes)); CK(hipFree(s[k].bad)); CK(hipFree(s[k].first));
            for (int i = 0; i < 2; ++i) CK(hipFree(s[k].small[i]));
            for (int i = 0; i < 4; ++i) CK(hipFree(s[k].big[i]));
        }
    }
    return 0;
}
