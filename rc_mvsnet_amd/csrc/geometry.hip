// librcmvs_hip.so: version / error plumbing, layout converters, homography composition and the
// per-pixel hypothesis-plane table.  gfx950 only.
#include "common.h"

namespace rcmvs {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

// ------------------------------------------------------------------------------------------
// layout: (N,C,S) <-> (N,S,C).  One thread per (s, channel-quad); the channels-last side is
// accessed as float4 (fully coalesced), the planar side as 4 scalar accesses that are
// contiguous across the lanes sharing a quad index.  C % 4 == 0.
// ------------------------------------------------------------------------------------------
template <bool TO_LAST>
__global__ __launch_bounds__(256) void layout_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                      int C, long long S) {
    const int Q = C >> 2;
    const int n = blockIdx.y;
    // lanes: s fastest inside a group of 64 so the planar accesses coalesce
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long sblk = t / (64LL * Q);
    int r = (int)(t % (64LL * Q));
    int q = r / 64;
    long long s = sblk * 64 + (r % 64);
    if (s >= S) return;
    const float* pl = (TO_LAST ? src : dst) + (long long)n * C * S;   // planar side base (read or write)
    if (TO_LAST) {
        float4 v;
        v.x = pl[(4LL * q + 0) * S + s];
        v.y = pl[(4LL * q + 1) * S + s];
        v.z = pl[(4LL * q + 2) * S + s];
        v.w = pl[(4LL * q + 3) * S + s];
        *reinterpret_cast<float4*>(dst + ((long long)n * S + s) * C + 4 * q) = v;
    } else {
        float4 v = *reinterpret_cast<const float4*>(src + ((long long)n * S + s) * C + 4 * q);
        float* o = dst + (long long)n * C * S;
        o[(4LL * q + 0) * S + s] = v.x;
        o[(4LL * q + 1) * S + s] = v.y;
        o[(4LL * q + 2) * S + s] = v.z;
        o[(4LL * q + 3) * S + s] = v.w;
    }
}

// channels not a multiple of 4 (e.g. RGB): scalar fallback, one thread per element of the
// channels-last side.
template <bool TO_LAST>
__global__ __launch_bounds__(256) void layout_scalar_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                             int C, long long S) {
    const int n = blockIdx.y;
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= S * C) return;
    long long s = t / C;
    int c = (int)(t % C);
    long long planar = ((long long)n * C + c) * S + s, last = ((long long)n * S + s) * C + c;
    if (TO_LAST) dst[last] = src[planar]; else dst[planar] = src[last];
}

template <bool TO_LAST>
static int layout_launch(const float* src, float* dst, int N, int C, long long S, void* stream) {
    RCMVS_REQUIRE(src && dst && N > 0 && C > 0 && S > 0, "layout: bad arguments N=%d C=%d S=%lld", N, C, S);
    if (C % 4 == 0) {
        long long threads = cdiv(S, 64) * 64 * (C / 4);
        dim3 grid((unsigned)cdiv(threads, 256), N);
        hipLaunchKernelGGL(layout_kernel<TO_LAST>, grid, dim3(256), 0, as_stream(stream), src, dst, C, S);
    } else {
        dim3 grid((unsigned)cdiv(S * C, 256), N);
        hipLaunchKernelGGL(layout_scalar_kernel<TO_LAST>, grid, dim3(256), 0, as_stream(stream), src, dst, C, S);
    }
    return launch_status("layout");
}

// ------------------------------------------------------------------------------------------
// homography composition, fp64 on the device (one thread per (b, source view)).
// ------------------------------------------------------------------------------------------
__device__ static void fold_intrinsics(const float* p, double A[4][4]) {
    // p: (2,4,4) -- [0] extrinsic, [1][:3][:3] intrinsic.  A[:3,:4] = K @ E[:3,:4]; A[3] = E[3].
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += (double)p[16 + i * 4 + k] * (double)p[k * 4 + j];
            A[i][j] = s;
        }
    for (int j = 0; j < 4; ++j) A[3][j] = (double)p[12 + j];
}

// (blockIdx.y = cascade stage when the three stages' projection tensors are composed in ONE launch: rcmvs_compose_homography_stages)
struct ProjPtrs { const float* p[4]; };
// zero / zero_n: an optional float buffer the launch clears on the side (the cascade's activation-bound rows: one fill launch per scene less)
__global__ void compose_homography_kernel(ProjPtrs pp, float* __restrict__ rot,
                                          float* __restrict__ trans, int B, int V, float* __restrict__ zero, int zero_n) {
    for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; i < zero_n; i += gridDim.x * gridDim.y * blockDim.x) zero[i] = 0.0f;
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * (V - 1)) return;
    const float* proj = pp.p[blockIdx.y];
    rot += (long long)blockIdx.y * B * (V - 1) * 9;
    trans += (long long)blockIdx.y * B * (V - 1) * 3;
    int b = t / (V - 1), v = 1 + t % (V - 1);
    double R[4][4], S[4][4], inv[4][4];
    fold_intrinsics(proj + ((long long)b * V + 0) * 32, R);
    fold_intrinsics(proj + ((long long)b * V + v) * 32, S);
    // Gauss-Jordan with partial pivoting on [R | I]
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) inv[i][j] = (i == j) ? 1.0 : 0.0;
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        double best = fabs(R[c][c]);
        for (int r = c + 1; r < 4; ++r)
            if (fabs(R[r][c]) > best) { best = fabs(R[r][c]); piv = r; }
        if (piv != c)
            for (int j = 0; j < 4; ++j) {
                double t0 = R[c][j]; R[c][j] = R[piv][j]; R[piv][j] = t0;
                double t1 = inv[c][j]; inv[c][j] = inv[piv][j]; inv[piv][j] = t1;
            }
        double d = 1.0 / R[c][c];
        for (int j = 0; j < 4; ++j) { R[c][j] *= d; inv[c][j] *= d; }
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            double f = R[r][c];
            for (int j = 0; j < 4; ++j) { R[r][j] -= f * R[c][j]; inv[r][j] -= f * inv[c][j]; }
        }
    }
    float* ro = rot + (long long)t * 9;
    float* tr = trans + (long long)t * 3;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += S[i][k] * inv[k][j];
            if (j < 3) ro[i * 3 + j] = (float)s; else tr[i] = (float)s;
        }
    }
}

// ------------------------------------------------------------------------------------------
// hypothesis planes: planes[b,y,x] = {d_0, delta}
// ------------------------------------------------------------------------------------------
__device__ static inline float ld_agent(const float* p) {
#if defined(__AMDGCN__)
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    return *p;              // host pass / CPU emulator of the tests
#endif
}

__device__ static inline float bilinear_up(const float* __restrict__ p, int hp, int wp, float sh, float sw, int Y, int X) {
#pragma clang fp contract(off)
    // ATen upsample_bilinear2d, align_corners=False: src = max(scale*(dst+0.5)-0.5, 0)
    float sy = sh * ((float)Y + 0.5f) - 0.5f;
    float sx = sw * ((float)X + 0.5f) - 0.5f;
    sy = sy < 0.f ? 0.f : sy;
    sx = sx < 0.f ? 0.f : sx;
    int y0 = (int)sy, x0 = (int)sx;
    y0 = y0 > hp - 1 ? hp - 1 : y0;
    x0 = x0 > wp - 1 ? wp - 1 : x0;
    int y1 = y0 + 1 > hp - 1 ? hp - 1 : y0 + 1;
    int x1 = x0 + 1 > wp - 1 ? wp - 1 : x0 + 1;
    float ly1 = sy - (float)y0, lx1 = sx - (float)x0;
    float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
    // The previous stage's depth map is read at agent scope (sc1 loads).  Round 3, profiles/r3_two_streams.txt: with two scenes in
    // flight on two HIP streams this kernel was the FIRST op of every corrupted scene -- it read 32-byte pieces of the depth map's
    // previous contents (a reused allocator block) although the depth kernel before it on the same stream had completed (explicit
    // event dependencies did not change that; a release fence at the end of the writer did not either; these loads did: 0 of 1200
    // scenes against 7 %).  The map is 80-330 KB, the kernel is bound by its stores; the loads cost nothing measurable.
    // Round 4 (profiles/r4_two_streams_ab.txt): a one-lane agent-scope acquire at the top of EVERY inference kernel with plain loads
    // here does NOT remove the corruption (84 of 90 rounds) and costs 18 % on one stream: the stale data is not in the reader's vector
    // L1, the mechanism stays unexplained, and the binding refuses a second stream per process (ops._stream(); profiles/r6_two_streams.txt).
    float top = lx0 * ld_agent(p + y0 * wp + x0) + lx1 * ld_agent(p + y0 * wp + x1);
    float bot = lx0 * ld_agent(p + y1 * wp + x0) + lx1 * ld_agent(p + y1 * wp + x1);
    return ly0 * top + ly1 * bot;
}

__global__ __launch_bounds__(256) void planes_kernel(const float* __restrict__ prev, const float* __restrict__ dv,
                                                      float* __restrict__ planes, int hp, int wp, int H, int W,
                                                      int scale, int D, float ratio, int ND) {
#pragma clang fp contract(off)
    const int h = H / scale, w = W / scale;
    const int b = blockIdx.y;
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= h * w) return;
    int y = t / w, x = t % w;
    float d0, delta;
    if (prev == nullptr) {
        float lo = dv[(long long)b * ND], hi = dv[(long long)b * ND + ND - 1];
        d0 = lo;
        delta = (hi - lo) / (float)(D - 1);
    } else {
        // casmvsnet.py:357-359 -- python doubles, batch item 0, divided by ND (192), not ND-1
        double itv = ((double)dv[ND - 1] - (double)dv[0]) / (double)ND;
        float half = (float)((double)D / 2.0 * ((double)ratio * itv));
        const float* p = prev + (long long)b * hp * wp;
        float sh = (float)hp / (float)H, sw = (float)wp / (float)W;
        float fd = (float)(D - 1);
        if (scale == 1) {
            float c = bilinear_up(p, hp, wp, sh, sw, y, x);
            float cmin = c - half, cmax = c + half;
            d0 = cmin;
            delta = (cmax - cmin) / fd;
        } else {
            // trilinear down-sampling by an even integer factor = 0.5/0.5 blend of the two
            // centre rows / columns of each scale x scale block (W first, then H)
            int Y0 = y * scale + scale / 2 - 1, X0 = x * scale + scale / 2 - 1;
            float mn[2][2], it[2][2];
            for (int j = 0; j < 2; ++j)
                for (int i = 0; i < 2; ++i) {
                    float c = bilinear_up(p, hp, wp, sh, sw, Y0 + j, X0 + i);
                    float cmin = c - half, cmax = c + half;
                    mn[j][i] = cmin;
                    it[j][i] = (cmax - cmin) / fd;
                }
            d0 = 0.5f * (0.5f * mn[0][0] + 0.5f * mn[0][1]) + 0.5f * (0.5f * mn[1][0] + 0.5f * mn[1][1]);
            delta = 0.5f * (0.5f * it[0][0] + 0.5f * it[0][1]) + 0.5f * (0.5f * it[1][0] + 0.5f * it[1][1]);
        }
    }
    float2 o = make_float2(d0, delta);
    reinterpret_cast<float2*>(planes)[(long long)b * h * w + t] = o;
}

}  // namespace rcmvs

using namespace rcmvs;

extern "C" {

int rcmvs_version(void) { return RCMVS_VERSION; }
const char* rcmvs_last_error_string(void) { return err_buf(); }

int rcmvs_nchw_to_nhwc(const float* src, float* dst, int N, int C, long long S, void* stream) {
    return layout_launch<true>(src, dst, N, C, S, stream);
}
int rcmvs_nhwc_to_nchw(const float* src, float* dst, int N, int C, long long S, void* stream) {
    return layout_launch<false>(src, dst, N, C, S, stream);
}

int rcmvs_compose_homography(const float* proj, float* rot, float* trans, int B, int V, void* stream) {
    RCMVS_REQUIRE(proj && rot && trans, "compose_homography: null pointer");
    RCMVS_REQUIRE(B > 0 && V >= 2 && V - 1 <= RCMVS_MAX_SRC_VIEWS, "compose_homography: B=%d V=%d", B, V);
    const int n = B * (V - 1);
    ProjPtrs pp{{proj, nullptr, nullptr, nullptr}};
    hipLaunchKernelGGL(compose_homography_kernel, dim3((n + 63) / 64), dim3(64), 0, as_stream(stream), pp, rot, trans, B, V, (float*)nullptr, 0);
    return launch_status("compose_homography");
}

int rcmvs_compose_homography_stages(const float* proj0, const float* proj1, const float* proj2, const float* proj3, int nstage,
                                    float* rot, float* trans, int B, int V, float* zero, long long zero_n, void* stream) {
    RCMVS_REQUIRE(zero_n >= 0 && zero_n < (1LL << 24) && (zero || zero_n == 0), "compose_homography_stages: bad side buffer (%lld floats)", zero_n);
    RCMVS_REQUIRE(rot && trans && nstage >= 1 && nstage <= 4, "compose_homography_stages: bad arguments (1..4 stages)");
    RCMVS_REQUIRE(B > 0 && V >= 2 && V - 1 <= RCMVS_MAX_SRC_VIEWS, "compose_homography_stages: B=%d V=%d", B, V);
    ProjPtrs pp{{proj0, proj1, proj2, proj3}};
    for (int s = 0; s < nstage; ++s) RCMVS_REQUIRE(pp.p[s], "compose_homography_stages: stage %d has no projection tensor", s);
    const int n = B * (V - 1);
    hipLaunchKernelGGL(compose_homography_kernel, dim3((n + 63) / 64, nstage), dim3(zero_n ? 256 : 64), 0, as_stream(stream), pp, rot, trans, B, V, zero, (int)zero_n);
    return launch_status("compose_homography_stages");
}

int rcmvs_hypothesis_planes(const float* prev_depth, const float* depth_values, float* planes,
                            int B, int hp, int wp, int H, int W, int scale,
                            int D, float ratio, int ND, void* stream) {
    RCMVS_REQUIRE(depth_values && planes, "hypothesis_planes: null pointer");
    RCMVS_REQUIRE(B > 0 && H > 0 && W > 0 && D >= 2 && ND >= 2, "hypothesis_planes: bad sizes");
    RCMVS_REQUIRE(scale == 1 || scale == 2 || scale == 4, "hypothesis_planes: scale must be 1, 2 or 4 (got %d)", scale);
    RCMVS_REQUIRE(H % scale == 0 && W % scale == 0, "hypothesis_planes: H,W must be multiples of scale");
    RCMVS_REQUIRE(prev_depth == nullptr || (hp > 0 && wp > 0), "hypothesis_planes: prev_depth needs hp,wp");
    int h = H / scale, w = W / scale;
    dim3 grid((h * w + 255) / 256, B);
    hipLaunchKernelGGL(planes_kernel, grid, dim3(256), 0, as_stream(stream), prev_depth, depth_values, planes,
                       hp, wp, H, W, scale, D, ratio, ND);
    return launch_status("hypothesis_planes");
}

}  // extern "C"
