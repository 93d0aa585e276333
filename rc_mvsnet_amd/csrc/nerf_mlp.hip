// NeRF MLP of the rendering-consistency branch (Renderer_ours.forward, use_viewdirs=True,
// models/render_models.py:192-220; Embedder.embed :45-49; run_network_mvs, models/renderer.py:42-63)
// as a chain of exact-fp32 MFMA GEMMs (v_mfma_f32_16x16x4_f32) with fused epilogues.  gfx950 only.
//
// Per point (M = rays x samples): X0 = [ndc, sin(ndc 2^j), cos(ndc 2^j)]_{j<10} (63, padded to 64);
//   bias = W_b feat20 + b_b;  h = relu((W_i h + b_i) * bias) for i = 0..5 with the input re-attached
//   after i = 4;  sigma = relu(w_a h + b_a);  f = W_f h + b_f;  hv = relu(W_v [f, dir] + b_v);
//   rgb = sigmoid(W_r hv + b_r);  raw = [rgb, sigma].
// GEMM mapping (same fragment scheme as conv3d_mfma.hip): MFMA rows = 16 output features, columns =
//   16 points, K = input features in chunks of 16 (lane (n, kq) loads 4 consecutive features of
//   point n with one 16-byte load and feeds component j to MFMA j); the accumulator fragment of a
//   lane is 4 consecutive output features of one point -> float4 epilogue and store, which lets a
//   layer write directly into a column window of a wider buffer (the skip / view concatenations are
//   never copied).  One wave = 64 points x all output features.
// Workspace per point (floats): XS 192 = [X0 (64) | h after layer 4 (128)], BIAS 128, HA 128, HB 128,
//   XV 144 = [f (128) | dir (3) | 0], HV 64  -> 784.
#include "common.h"

namespace rcmvs {

typedef float f32x4m __attribute__((ext_vector_type(4)));

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2 };

// layer table of the packed blob: {Cout, Kp}; image = [Kp/16][MT][64][4] floats, then MT*16 biases
struct MlpLayer { int cout, kp; };
__host__ __device__ inline int mlp_mt(int cout) { return (cout + 15) / 16; }
__host__ __device__ inline long long mlp_layer_floats(int cout, int kp) {
    return (long long)(kp / 16) * mlp_mt(cout) * 64 * 4 + mlp_mt(cout) * 16;
}
// order: pts_bias, L0, L1, L2, L3, L4, L5, alpha, feature, views, rgb
static const MlpLayer kLayers[11] = {{128, 32}, {128, 64}, {128, 128}, {128, 128}, {128, 128}, {128, 128}, {128, 192},
                                     {1, 128}, {128, 128}, {64, 144}, {3, 64}};

constexpr int WS_XS = 0, WS_BIAS = 192, WS_HA = 320, WS_HB = 448, WS_XV = 576, WS_HV = 720, WS_ROW = 784;

// dense (Cout, K) row-major weight + (Cout) bias -> MFMA image with optional column remap:
//   packed k index kk (< Kp) reads source column  (kk < split ? kk : kk - gap)  when that is a valid
//   source column and kk is not inside [split_lo, split) (the zero pad between the two parts).
__global__ void pack_linear_kernel(const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ packed,
                                   int Cout, int K, int Kp, int part0, int pad0) {
    // source columns [0, part0) -> packed [0, part0); packed [part0, part0+pad0) = 0;
    // source columns [part0, K) -> packed [part0+pad0, ...); remaining packed columns = 0
    const int MT = mlp_mt(Cout);
    const long long nimg = (long long)(Kp / 16) * MT * 64 * 4;
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nimg) {
        int j = (int)(t & 3);
        long long r = t >> 2;
        int lane = (int)(r % 64); r /= 64;
        int mt = (int)(r % MT);
        int chunk = (int)(r / MT);
        int m = lane & 15, kq = lane >> 4;
        int co = mt * 16 + m, kk = chunk * 16 + kq * 4 + j;
        int src = -1;
        if (kk < part0) src = kk;
        else if (kk >= part0 + pad0) src = kk - pad0;
        float v = 0.0f;
        if (co < Cout && src >= 0 && src < K) v = w[(long long)co * K + src];
        packed[t] = v;
    } else if (t < nimg + MT * 16) {
        int co = (int)(t - nimg);
        packed[t] = (co < Cout) ? bias[co] : 0.0f;
    }
}

// positional encoding + view-direction / padding fill
__global__ __launch_bounds__(256) void mlp_embed_kernel(const float* __restrict__ ndc, const float* __restrict__ dirs,
                                                         const float* __restrict__ w2c, float* __restrict__ feat,
                                                         float* __restrict__ ws, int M, int S, int ldf, int nfeat) {
    int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float* xs = ws + (long long)m * WS_ROW + WS_XS;
    float p[3] = {ndc[m * 3 + 0], ndc[m * 3 + 1], ndc[m * 3 + 2]};
    xs[0] = p[0]; xs[1] = p[1]; xs[2] = p[2];
    float f = 1.0f;
    for (int j = 0; j < 10; ++j) {
        for (int c = 0; c < 3; ++c) {
            const float a = p[c] * f;
            xs[3 + j * 3 + c] = sinf(a);
            xs[33 + j * 3 + c] = cosf(a);
        }
        f *= 2.0f;
    }
    xs[63] = 0.0f;
    // view direction: normalise the ray direction, rotate by w2c_ref[:3,:3] (renderer.py:141-152,172-177)
    const int ray = m / S;
    const float dx = dirs[ray * 3 + 0], dy = dirs[ray * 3 + 1], dz = dirs[ray * 3 + 2];
    const float nrm = sqrtf((dx * dx + dy * dy) + dz * dz);
    const float ux = dx / nrm, uy = dy / nrm, uz = dz / nrm;
    float* xv = ws + (long long)m * WS_ROW + WS_XV;
    xv[128] = (ux * w2c[0] + uy * w2c[1]) + uz * w2c[2];
    xv[129] = (ux * w2c[4] + uy * w2c[5]) + uz * w2c[6];
    xv[130] = (ux * w2c[8] + uy * w2c[9]) + uz * w2c[10];
    for (int c = 131; c < 144; ++c) xv[c] = 0.0f;
    for (int c = nfeat; c < ldf; ++c) feat[(long long)m * ldf + c] = 0.0f;
}

// Y[m][ycol + co] = act((X[m][:Kp] . W[co][:] + b[co]) * RM[m][co])
template <int MT, int ACT, bool ROWMUL>
__global__ __launch_bounds__(256) void linear_mfma_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ wimg,
                                                           const float* __restrict__ RM, int ldrm, float* __restrict__ Y,
                                                           int ldy, int M, int Kp, int Cout) {
    constexpr int NT = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const long long row0 = ((long long)blockIdx.x * 4 + wave) * (NT * 16);
    if (row0 >= M) return;
    f32x4m acc[NT][MT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[t][mt] = (f32x4m){0.f, 0.f, 0.f, 0.f};
    long long rows[NT];
    bool rv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { rows[t] = row0 + t * 16 + n; rv[t] = rows[t] < M; }
    const int chunks = Kp / 16;
    const float* wl = wimg + lane * 4;
    for (int c = 0; c < chunks; ++c) {
        float4 b[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
            b[t] = rv[t] ? *reinterpret_cast<const float4*>(X + rows[t] * ldx + c * 16 + kq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 a[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const float4*>(wl + ((long long)c * MT + mt) * 256);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].x, b[t].x, acc[t][mt], 0, 0, 0);
                acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].y, b[t].y, acc[t][mt], 0, 0, 0);
                acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].z, b[t].z, acc[t][mt], 0, 0, 0);
                acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].w, b[t].w, acc[t][mt], 0, 0, 0);
            }
    }
    const float* bias = wimg + (long long)chunks * MT * 256;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m0 = mt * 16 + kq * 4;
        if (m0 >= Cout) continue;
        const float4 bb = *reinterpret_cast<const float4*>(bias + m0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (!rv[t]) continue;
            float v[4] = {acc[t][mt][0] + bb.x, acc[t][mt][1] + bb.y, acc[t][mt][2] + bb.z, acc[t][mt][3] + bb.w};
            if (ROWMUL) {
                const float4 r4 = *reinterpret_cast<const float4*>(RM + rows[t] * ldrm + m0);
                v[0] *= r4.x; v[1] *= r4.y; v[2] *= r4.z; v[3] *= r4.w;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (ACT == ACT_RELU) v[j] = fmaxf(v[j], 0.0f);
                if (ACT == ACT_SIGMOID) v[j] = 1.0f / (1.0f + expf(-v[j]));
            }
            float* yp = Y + rows[t] * ldy + m0;
            if (m0 + 4 <= Cout && (ldy & 3) == 0) *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
            else
                for (int j = 0; j < 4 && m0 + j < Cout; ++j) yp[j] = v[j];
        }
    }
}

template <int MT, int ACT, bool ROWMUL>
static int linear_launch(const float* X, int ldx, const float* wimg, const float* RM, int ldrm, float* Y, int ldy, int M,
                         int Kp, int Cout, hipStream_t st) {
    dim3 grid((unsigned)cdiv(M, 256));
    hipLaunchKernelGGL((linear_mfma_kernel<MT, ACT, ROWMUL>), grid, dim3(256), 0, st, X, ldx, wimg, RM, ldrm, Y, ldy, M, Kp, Cout);
    return launch_status("nerf linear");
}

}  // namespace rcmvs

using namespace rcmvs;

extern "C" {

long long rcmvs_nerf_weight_floats(void) {
    long long n = 0;
    for (int i = 0; i < 11; ++i) n += mlp_layer_floats(kLayers[i].cout, kLayers[i].kp);
    return n;
}

long long rcmvs_nerf_workspace_floats(long long M) { return M * WS_ROW; }

/* wb [host]: 22 device pointers, (weight, bias) pairs in the order
 * pts_bias, pts_linears.0 .. .5, alpha_linear, feature_linear, views_linears.0, rgb_linear */
int rcmvs_pack_nerf_weights(const float* const* wb, float* blob, void* stream) {
    RCMVS_REQUIRE(wb && blob, "pack_nerf_weights: null pointer");
    // source K, split point and zero pad between the parts (skip layer: [63 | pad 1 | 128], views: [128 | 3 | pad])
    static const int srcK[11] = {20, 63, 128, 128, 128, 128, 191, 128, 128, 131, 64};
    static const int part0[11] = {20, 63, 128, 128, 128, 128, 63, 128, 128, 131, 64};
    static const int pad0[11] = {0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0};
    long long off = 0;
    for (int i = 0; i < 11; ++i) {
        RCMVS_REQUIRE(wb[2 * i] && wb[2 * i + 1], "pack_nerf_weights: null layer pointer");
        long long n = mlp_layer_floats(kLayers[i].cout, kLayers[i].kp);
        hipLaunchKernelGGL(pack_linear_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, as_stream(stream), wb[2 * i],
                           wb[2 * i + 1], blob + off, kLayers[i].cout, srcK[i], kLayers[i].kp, part0[i], pad0[i]);
        int rc = launch_status("pack_nerf_weights");
        if (rc) return rc;
        off += n;
    }
    return 0;
}

int rcmvs_nerf_mlp_fwd(const float* ndc, float* feat, int ldf, const float* dirs, const float* w2c_ref,
                       const float* weights, float* workspace, float* raw, int N, int S, void* stream) {
    RCMVS_REQUIRE(ndc && feat && dirs && w2c_ref && weights && workspace && raw, "nerf_mlp_fwd: null pointer");
    RCMVS_REQUIRE(N > 0 && S > 0 && ldf == 32, "nerf_mlp_fwd: feat must have a row stride of 32 floats (20 used)");
    hipStream_t st = as_stream(stream);
    const int M = N * S;
    const float* L[11];
    long long off = 0;
    for (int i = 0; i < 11; ++i) { L[i] = weights + off; off += mlp_layer_floats(kLayers[i].cout, kLayers[i].kp); }
    float* ws = workspace;
    hipLaunchKernelGGL(mlp_embed_kernel, dim3((M + 255) / 256), dim3(256), 0, st, ndc, dirs, w2c_ref, feat, ws, M, S, ldf, 20);
    int rc = launch_status("nerf embed");
    if (rc) return rc;
    float *XS = ws + WS_XS, *BI = ws + WS_BIAS, *HA = ws + WS_HA, *HB = ws + WS_HB, *XV = ws + WS_XV, *HV = ws + WS_HV;
    const int R = WS_ROW;
    // bias = pts_bias(feat)
    if ((rc = linear_launch<8, ACT_NONE, false>(feat, ldf, L[0], nullptr, 0, BI, R, M, 32, 128, st))) return rc;
    // trunk
    if ((rc = linear_launch<8, ACT_RELU, true>(XS, R, L[1], BI, R, HA, R, M, 64, 128, st))) return rc;       // 0
    if ((rc = linear_launch<8, ACT_RELU, true>(HA, R, L[2], BI, R, HB, R, M, 128, 128, st))) return rc;      // 1
    if ((rc = linear_launch<8, ACT_RELU, true>(HB, R, L[3], BI, R, HA, R, M, 128, 128, st))) return rc;      // 2
    if ((rc = linear_launch<8, ACT_RELU, true>(HA, R, L[4], BI, R, HB, R, M, 128, 128, st))) return rc;      // 3
    if ((rc = linear_launch<8, ACT_RELU, true>(HB, R, L[5], BI, R, XS + 64, R, M, 128, 128, st))) return rc; // 4 -> skip buffer
    if ((rc = linear_launch<8, ACT_RELU, true>(XS, R, L[6], BI, R, HA, R, M, 192, 128, st))) return rc;      // 5
    // heads
    if ((rc = linear_launch<1, ACT_RELU, false>(HA, R, L[7], nullptr, 0, raw + 3, 4, M, 128, 1, st))) return rc;   // sigma
    if ((rc = linear_launch<8, ACT_NONE, false>(HA, R, L[8], nullptr, 0, XV, R, M, 128, 128, st))) return rc;      // feature
    if ((rc = linear_launch<4, ACT_RELU, false>(XV, R, L[9], nullptr, 0, HV, R, M, 144, 64, st))) return rc;       // views
    return linear_launch<1, ACT_SIGMOID, false>(HV, R, L[10], nullptr, 0, raw, 4, M, 64, 3, st);                  // rgb
}

}  // extern "C"
