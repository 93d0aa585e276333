// Depth-map fusion filter: geometric-consistency check of a reference depth map against its
// source views, photometric mask, averaged depth, back-projection to world points and ordered compaction of the surviving
// points.  Reference: eval_rcmvsnet_dtu.py:281-338 (reproject_with_depth, check_geometric_consistency) and :369-425 (the
// per-view body of filter_depth), which run on the CPU in numpy + cv2.remap, one process per scan.  gfx950 only.
//
// One thread per reference pixel walks all N source views (the matrices are wave-uniform, the only gathers are the four
// bilinear taps of each source depth map), so a reference view is ONE launch instead of ~40 numpy passes per source view;
// every depth map of the scan stays resident in HBM and is addressed by view index.  The chain is float64 with the
// reference's float32 cast points (fusion_math.h) -- MI355X runs fp64 FMA at half the fp32 rate, and the kernel is bound by
// its ~30 B / pixel of output anyway.  Compaction keeps numpy's row-major order of `array[mask]`: per-block counts, one
// single-block scan, scatter.
#include "common.h"
#include "fusion_math.h"

namespace rcmvs {

constexpr int FU_BLOCK = 256;

struct SrcIdx { int v[RCMVS_FUSE_MAX_SRC]; };

__global__ __launch_bounds__(FU_BLOCK) void fuse_view_kernel(
    const float* __restrict__ depth_all, int ref_idx, SrcIdx src, const float* __restrict__ conf, const float* __restrict__ img,
    const double* __restrict__ mats, float prob_thresh, int num_consistent, double dist_thresh, float depth_thresh,
    unsigned char* __restrict__ masks, float* __restrict__ depth_avg, float* __restrict__ xyz, unsigned char* __restrict__ rgb,
    float* __restrict__ dbg_depth, unsigned char* __restrict__ dbg_geo, float* __restrict__ dbg_xy, int N, int H, int W) {
    const int p = blockIdx.x * FU_BLOCK + threadIdx.x;
    const int plane = H * W;
    if (p >= plane) return;
    const int y = p / W, x = p - y * W;
    const float d_ref = depth_all[(long long)ref_idx * plane + p];
    int geo_sum = 0;
    float acc = 0.0f;                                            // python sum(): 0 + d_1 + d_2 + ... in float32
    for (int n = 0; n < N; ++n) {
        const fu::Reproj r = fu::reproject(mats, mats + fu::REF_MATS + n * fu::SRC_MATS, depth_all + (long long)src.v[n] * plane,
                                           H, W, x, y, d_ref, dist_thresh, depth_thresh);
        geo_sum += r.ok ? 1 : 0;
        acc += r.depth;
        if (dbg_depth) dbg_depth[(long long)n * plane + p] = r.depth;
        if (dbg_geo) dbg_geo[(long long)n * plane + p] = r.ok ? 1 : 0;
        if (dbg_xy) { dbg_xy[((long long)n * plane + p) * 2] = r.x_src; dbg_xy[((long long)n * plane + p) * 2 + 1] = r.y_src; }
    }
    const double avg = (double)(acc + d_ref) / (double)(geo_sum + 1);
    const bool photo = conf[p] > prob_thresh, geo = geo_sum >= num_consistent;
    masks[p] = photo; masks[plane + p] = geo; masks[2 * plane + p] = photo && geo;
    depth_avg[p] = (float)avg;
    double wpt[3];
    fu::world_point(mats, x, y, avg, wpt);
    xyz[p * 3 + 0] = (float)wpt[0]; xyz[p * 3 + 1] = (float)wpt[1]; xyz[p * 3 + 2] = (float)wpt[2];
    if (img) {
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[p * 3 + c] = (unsigned char)(int)(img[p * 3 + c] * 255.0f);
    }
}

// ---- ordered compaction ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(FU_BLOCK) void count_kernel(const unsigned char* __restrict__ mask, int* __restrict__ offsets, long long n) {
    const long long i = (long long)blockIdx.x * FU_BLOCK + threadIdx.x;
    const bool f = i < n && mask[i] != 0;
    const int c = __syncthreads_count(f);
    if (threadIdx.x == 0) offsets[blockIdx.x] = c;
}

// in place: counts[0..nblk) -> exclusive prefix sums, counts[nblk] = total.  One block of 1024 threads.
__global__ __launch_bounds__(1024) void scan_kernel(int* __restrict__ counts, int nblk) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int chunk = (nblk + 1023) / 1024;
    const int lo = t * chunk, hi = min(lo + chunk, nblk);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += counts[i];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;                                        // exclusive prefix of this thread's chunk
    for (int i = lo; i < hi; ++i) { const int c = counts[i]; counts[i] = run; run += c; }
    if (t == 1023) counts[nblk] = part[1023];
}

__global__ __launch_bounds__(FU_BLOCK) void scatter_kernel(const unsigned char* __restrict__ mask, const float* __restrict__ xyz,
                                                           const unsigned char* __restrict__ rgb, const int* __restrict__ offsets,
                                                           float* __restrict__ out_xyz, unsigned char* __restrict__ out_rgb, long long n) {
    __shared__ int wave_base[FU_BLOCK / WAVE];
    const long long i = (long long)blockIdx.x * FU_BLOCK + threadIdx.x;
    const bool f = i < n && mask[i] != 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long b = __ballot(f);
    const int before = __popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) wave_base[wave] = __popcll(b);
    __syncthreads();
    int base = offsets[blockIdx.x];
    for (int w = 0; w < wave; ++w) base += wave_base[w];
    if (f) {
        const long long o = (long long)base + before;
#pragma unroll
        for (int c = 0; c < 3; ++c) out_xyz[o * 3 + c] = xyz[i * 3 + c];
        if (rgb) {
#pragma unroll
            for (int c = 0; c < 3; ++c) out_rgb[o * 3 + c] = rgb[i * 3 + c];
        }
    }
}

}  // namespace rcmvs

using namespace rcmvs;

extern "C" int rcmvs_fuse_view(const float* depth_all, int ref_idx, const int* src_idx_host, const float* conf, const float* img,
                               const double* mats, float prob_thresh, int num_consistent, double dist_thresh, float depth_thresh,
                               unsigned char* masks, float* depth_avg, float* xyz, unsigned char* rgb,
                               float* dbg_depth, unsigned char* dbg_geo, float* dbg_xy, int N, int H, int W, void* stream) {
    RCMVS_REQUIRE(depth_all && src_idx_host && conf && mats && masks && depth_avg && xyz, "fuse_view: null pointer");
    RCMVS_REQUIRE((img == nullptr) == (rgb == nullptr), "fuse_view: img and rgb must be given together");
    RCMVS_REQUIRE(N >= 1 && N <= RCMVS_FUSE_MAX_SRC, "fuse_view: %d source views (1..%d)", N, RCMVS_FUSE_MAX_SRC);
    RCMVS_REQUIRE(H > 0 && W > 0 && (long long)H * W < (1ll << 30) && ref_idx >= 0, "fuse_view: bad dims H=%d W=%d ref=%d", H, W, ref_idx);
    SrcIdx s;
    for (int n = 0; n < RCMVS_FUSE_MAX_SRC; ++n) s.v[n] = n < N ? src_idx_host[n] : 0;
    for (int n = 0; n < N; ++n) RCMVS_REQUIRE(s.v[n] >= 0, "fuse_view: negative source index");
    hipLaunchKernelGGL(fuse_view_kernel, dim3((H * W + FU_BLOCK - 1) / FU_BLOCK), dim3(FU_BLOCK), 0, as_stream(stream),
                       depth_all, ref_idx, s, conf, img, mats, prob_thresh, num_consistent, dist_thresh, depth_thresh,
                       masks, depth_avg, xyz, rgb, dbg_depth, dbg_geo, dbg_xy, N, H, W);
    return launch_status("fuse_view");
}

extern "C" int rcmvs_compact_points(const unsigned char* mask, const float* xyz, const unsigned char* rgb, float* out_xyz,
                                    unsigned char* out_rgb, int* block_offsets, long long n, void* stream) {
    RCMVS_REQUIRE(mask && xyz && out_xyz && block_offsets, "compact_points: null pointer");
    RCMVS_REQUIRE((rgb == nullptr) == (out_rgb == nullptr), "compact_points: rgb and out_rgb must be given together");
    RCMVS_REQUIRE(n > 0 && n < (1ll << 31), "compact_points: n=%lld", n);
    hipStream_t st = as_stream(stream);
    const int nblk = (int)((n + FU_BLOCK - 1) / FU_BLOCK);
    hipLaunchKernelGGL(count_kernel, dim3(nblk), dim3(FU_BLOCK), 0, st, mask, block_offsets, n);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, block_offsets, nblk);
    hipLaunchKernelGGL(scatter_kernel, dim3(nblk), dim3(FU_BLOCK), 0, st, mask, xyz, rgb, block_offsets, out_xyz, out_rgb, n);
    return launch_status("compact_points");
}
