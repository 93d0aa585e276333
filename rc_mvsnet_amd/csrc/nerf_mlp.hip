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
__global__ __launch_bounds__(256) void mlp_embed_rows_kernel(const float* __restrict__ ndc, const float* __restrict__ dirs,
                                                              const float* __restrict__ w2c, float* __restrict__ feat,
                                                              float* __restrict__ ws, int M, int S, int ldf, int nfeat, int row, int xs_off, int xv_off) {
    int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float* xs = ws + (long long)m * row + xs_off;
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
    float* xv = ws + (long long)m * row + xv_off;
    xv[128] = (ux * w2c[0] + uy * w2c[1]) + uz * w2c[2];
    xv[129] = (ux * w2c[4] + uy * w2c[5]) + uz * w2c[6];
    xv[130] = (ux * w2c[8] + uy * w2c[9]) + uz * w2c[10];
    for (int c = 131; c < 144; ++c) xv[c] = 0.0f;
    for (int c = nfeat; c < ldf; ++c) feat[(long long)m * ldf + c] = 0.0f;
}

// stand-alone forward of the MLP (Renderer_ours.forward(x), models/render_models.py:192-220): x rows are already
// [embedded point (63) | point feature (20) | view direction (3)]; they are copied into the workspace layout of the fused forward
__global__ __launch_bounds__(256) void mlp_scatter_rows_kernel(const float* __restrict__ x, int ldx, float* __restrict__ feat, float* __restrict__ ws,
                                                                long long M, int ldf, int row, int xs_off, int xv_off) {
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const float* xr = x + m * ldx;
    float* xs = ws + m * row + xs_off;
    for (int c = 0; c < 63; ++c) xs[c] = xr[c];
    xs[63] = 0.0f;
    float* fr = feat + m * ldf;
    for (int c = 0; c < 20; ++c) fr[c] = xr[63 + c];
    for (int c = 20; c < ldf; ++c) fr[c] = 0.0f;
    float* xv = ws + m * row + xv_off;
    xv[128] = xr[83]; xv[129] = xr[84]; xv[130] = xr[85];
    for (int c = 131; c < 144; ++c) xv[c] = 0.0f;
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


// ======================================================================================================================
// Training: activations of every layer are kept (one row of TW_ROW floats per point) and the backward pass runs on the same
// fp32 MFMA fragment scheme -- autograd of Renderer_ours.forward (models/render_models.py:192-220), no library GEMM.
//   data gradient of a layer:  dX = dZ W          -> linear_bwd_kernel (A = the transposed weight image, B = dZ formed on the
//       fly from dY, the ReLU mask of Y and the multiplicative bias; dZ is written back over dY for the weight gradient, and
//       the gradient of the multiplicative bias, dY mask Z = dY mask Y / bias, is accumulated on the way)
//   weight gradient:           dW = dZ^T X, db = column sums of dZ -> wgrad_kernel (LDS-staged 64-point tiles, one partial
//       result per point chunk) + wgrad_reduce_kernel (deterministic sum over chunks, scatter into the parameter's shape)
// Row layout of the training workspace (floats): X0 64 | h4 128 (= the 192-wide input of layer 5) | bias 128 | h0 h1 h2 h3 h5
//   5 x 128 | XV 144 = [f 128 | dir 3 | 0] | HV 64.
constexpr int TW_XS = 0, TW_BIAS = 192, TW_H0 = 320, TW_H1 = 448, TW_H2 = 576, TW_H3 = 704, TW_H5 = 832, TW_XV = 960, TW_HV = 1104, TW_ROW = 1168;
// gradient workspace row: dXS 192 (dX0 unused | dh4) | dBIAS 128 | GA 128 | GB 128 | dXV 144 | dHV 64 | head 8 = [dz_rgb 3, 0 | dz_sigma, 0 0 0]
constexpr int GW_XS = 0, GW_BIAS = 192, GW_A = 320, GW_B = 448, GW_XV = 576, GW_HV = 720, GW_HEAD = 784, GW_ROW = 792;

// transposed image of a layer for the data gradient: "output" rows = packed input columns kp < Kp, "K" = output channels
// co < Cp (Cout rounded up to 16); same [chunk][MT][lane][4] fragment format as pack_linear_kernel, bias block zero.
__global__ void pack_linear_t_kernel(const float* __restrict__ w, float* __restrict__ packed, int Cout, int K, int Kp, int part0, int pad0) {
    const int Cp = (Cout + 15) / 16 * 16, MT = Kp / 16;
    const long long nimg = (long long)(Cp / 16) * MT * 64 * 4;
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nimg) {
        int j = (int)(t & 3);
        long long r = t >> 2;
        int lane = (int)(r % 64); r /= 64;
        int mt = (int)(r % MT);
        int chunk = (int)(r / MT);
        int kp = mt * 16 + (lane & 15), co = chunk * 16 + (lane >> 4) * 4 + j;
        int src = -1;
        if (kp < part0) src = kp;
        else if (kp >= part0 + pad0) src = kp - pad0;
        packed[t] = (co < Cout && src >= 0 && src < K) ? w[(long long)co * K + src] : 0.0f;
    } else if (t < nimg + MT * 16) {
        packed[t] = 0.0f;
    }
}

// dZ = dY [* (Y > 0)] [* RM]  (written back over dY),  dRM (+)= dY (Y > 0) Y / RM,  OUT (+)= dZ W   (W through its transposed image)
// G (M, ldg): dY in, dZ out.  Cout = valid columns of G.  MT = 16-column tiles of OUT (0: no data gradient wanted).
template <int MT, bool RELU, bool ROWMUL>
__global__ __launch_bounds__(256) void linear_bwd_kernel(float* __restrict__ G, int ldg, const float* __restrict__ Y, int ldy,
                                                          const float* __restrict__ RM, int ldrm, float* __restrict__ dRM, int lddrm, int drm_acc,
                                                          const float* __restrict__ wimg, float* __restrict__ OUT, int ldo, int out_acc,
                                                          int M, int Cout) {
    constexpr int NT = 4;
    constexpr int MTA = MT > 0 ? MT : 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const long long row0 = ((long long)blockIdx.x * 4 + wave) * (NT * 16);
    if (row0 >= M) return;
    f32x4m acc[NT][MTA];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mt = 0; mt < MTA; ++mt) acc[t][mt] = (f32x4m){0.f, 0.f, 0.f, 0.f};
    long long rows[NT];
    bool rv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { rows[t] = row0 + t * 16 + n; rv[t] = rows[t] < M; }
    const int chunks = (Cout + 15) / 16;
    const float* wl = wimg + lane * 4;
    for (int c = 0; c < chunks; ++c) {
        const int col = c * 16 + kq * 4;
        float4 b[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            b[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rv[t] && col < Cout) {
                float4 g = *reinterpret_cast<const float4*>(G + rows[t] * ldg + col);
                if (RELU || ROWMUL) {
                    const float4 yv = *reinterpret_cast<const float4*>(Y + rows[t] * ldy + col);
                    if (RELU) { g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f; g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f; }
                    if (ROWMUL) {
                        const float4 rm = *reinterpret_cast<const float4*>(RM + rows[t] * ldrm + col);
                        // d(bias) = g * Z with Z = Y / bias where the unit is active (g = 0 elsewhere)
                        float4 dr = make_float4(g.x != 0.f ? g.x * (yv.x / rm.x) : 0.f, g.y != 0.f ? g.y * (yv.y / rm.y) : 0.f,
                                                g.z != 0.f ? g.z * (yv.z / rm.z) : 0.f, g.w != 0.f ? g.w * (yv.w / rm.w) : 0.f);
                        float* dp = dRM + rows[t] * lddrm + col;
                        if (drm_acc) { const float4 o = *reinterpret_cast<const float4*>(dp); dr.x += o.x; dr.y += o.y; dr.z += o.z; dr.w += o.w; }
                        *reinterpret_cast<float4*>(dp) = dr;
                        g.x *= rm.x; g.y *= rm.y; g.z *= rm.z; g.w *= rm.w;
                    }
                    *reinterpret_cast<float4*>(G + rows[t] * ldg + col) = g;
                }
                b[t] = g;
            }
        }
        if (MT == 0) continue;
        float4 a[MTA];
#pragma unroll
        for (int mt = 0; mt < MTA; ++mt) a[mt] = *reinterpret_cast<const float4*>(wl + ((long long)c * MTA + mt) * 256);
#pragma unroll
        for (int mt = 0; mt < MTA; ++mt)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].x, b[t].x, acc[t][mt], 0, 0, 0);
                acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].y, b[t].y, acc[t][mt], 0, 0, 0);
                acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].z, b[t].z, acc[t][mt], 0, 0, 0);
                acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].w, b[t].w, acc[t][mt], 0, 0, 0);
            }
    }
    if (MT == 0) return;
#pragma unroll
    for (int mt = 0; mt < MTA; ++mt) {
        const int m0 = mt * 16 + kq * 4;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (!rv[t]) continue;
            float* op = OUT + rows[t] * ldo + m0;
            float4 v = make_float4(acc[t][mt][0], acc[t][mt][1], acc[t][mt][2], acc[t][mt][3]);
            if (out_acc) { const float4 o = *reinterpret_cast<const float4*>(op); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *reinterpret_cast<float4*>(op) = v;
        }
    }
}

// heads: dz_rgb = d(rgb) rgb (1 - rgb) (sigmoid), dz_sigma = d(sigma) [sigma > 0] (ReLU); raw / draw (M,4) = [rgb, sigma]
__global__ __launch_bounds__(256) void mlp_head_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ draw, float* __restrict__ gw, int M) {
    int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const float4 r = reinterpret_cast<const float4*>(raw)[m], d = reinterpret_cast<const float4*>(draw)[m];
    float* h = gw + (long long)m * GW_ROW + GW_HEAD;
    *reinterpret_cast<float4*>(h) = make_float4(d.x * r.x * (1.0f - r.x), d.y * r.y * (1.0f - r.y), d.z * r.z * (1.0f - r.z), 0.0f);
    *reinterpret_cast<float4*>(h + 4) = make_float4(r.w > 0.0f ? d.w : 0.0f, 0.0f, 0.0f, 0.0f);
}

// partial weight gradient of one chunk of points: P[chunk][co][k] = sum_m A[m][co] B[m][k]  (co < CAp, k in this block's 64 columns),
// pdb[chunk][co] = sum_m A[m][co].  A (M, lda) with ca4 readable columns (multiple of 4), B (M, ldb) with cb4 readable columns.
constexpr int WG_P = 64, WG_AS = 144, WG_BS = 80;
__global__ __launch_bounds__(256) void wgrad_kernel(const float* __restrict__ A, int lda, int ca4, const float* __restrict__ B, int ldb, int cb4,
                                                     float* __restrict__ P, float* __restrict__ pdb, int M, int CAp, int CBp, int stages) {
    __shared__ __attribute__((aligned(16))) float As[WG_P * WG_AS];
    __shared__ __attribute__((aligned(16))) float Bs[WG_P * WG_BS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = blockIdx.x, kb = blockIdx.y * 64;            // this block's columns of B
    const int i16 = lane & 15, kq = lane >> 4;
    f32x4m acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4m){0.f, 0.f, 0.f, 0.f};
    float4 dbs = make_float4(0.f, 0.f, 0.f, 0.f);                  // column sums of the A columns this thread stages (fixed: tid % 32)
    const int acol = (tid & 31) * 4;
    const bool cact = 2 * wave * 16 < CAp;                          // this wave's first co tile exists
    for (int s = 0; s < stages; ++s) {
        const long long m0 = ((long long)chunk * stages + s) * WG_P;
        __syncthreads();
        // stage A: 64 rows x 128 columns = 2048 float4, thread -> column group tid % 32, rows tid / 32 + 8 i
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = (tid >> 5) + 8 * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m0 + r < M && acol < ca4) v = *reinterpret_cast<const float4*>(A + (m0 + r) * lda + acol);
            *reinterpret_cast<float4*>(As + r * WG_AS + acol) = v;
            dbs.x += v.x; dbs.y += v.y; dbs.z += v.z; dbs.w += v.w;
        }
        // stage B: 64 rows x 64 columns = 1024 float4, thread -> column group tid % 16, rows tid / 16 + 16 i
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (tid >> 4) + 16 * i, c = (tid & 15) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m0 + r < M && kb + c < cb4) v = *reinterpret_cast<const float4*>(B + (m0 + r) * ldb + kb + c);
            *reinterpret_cast<float4*>(Bs + r * WG_BS + c) = v;
        }
        __syncthreads();
        if (!cact) continue;
#pragma unroll 4
        for (int p = 0; p < WG_P; p += 4) {
            float a[2], b[4];
#pragma unroll
            for (int t = 0; t < 2; ++t) a[t] = As[(p + kq) * WG_AS + (2 * wave + t) * 16 + i16];
#pragma unroll
            for (int t = 0; t < 4; ++t) b[t] = Bs[(p + kq) * WG_BS + t * 16 + i16];
#pragma unroll
            for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                for (int tb = 0; tb < 4; ++tb) acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
        }
    }
    // partial tile: rows co = (2 wave + ta) 16 + 4 kq + r, columns k = kb + tb 16 + i16
    float* Pc = P + (long long)chunk * CAp * CBp;
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = (2 * wave + ta) * 16 + 4 * kq + r, k = kb + tb * 16 + i16;
                if (co < CAp && k < CBp) Pc[(long long)co * CBp + k] = acc[ta][tb][r];
            }
    if (blockIdx.y == 0) {
        __syncthreads();
        float* red = As;                                            // [8 row groups][128 columns]
        *reinterpret_cast<float4*>(red + (tid >> 5) * 128 + acol) = dbs;
        __syncthreads();
        if (tid < 128 && tid < CAp) {
            float sum = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) sum += red[g * 128 + tid];
            pdb[(long long)chunk * CAp + tid] = sum;
        }
    }
}

// dW[co][ks] = sum_chunk P[chunk][co][kmap(ks)], db[co] = sum_chunk pdb[chunk][co]; kmap undoes the packed column layout
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ P, const float* __restrict__ pdb, float* __restrict__ dW,
                                                            float* __restrict__ db, int nch, int Cout, int K, int CAp, int CBp, int part0, int pad0) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < Cout * K) {
        const int co = t / K, ks = t - co * K;
        const int kp = ks < part0 ? ks : ks + pad0;
        float s = 0.f;
        for (int c = 0; c < nch; ++c) s += P[((long long)c * CAp + co) * CBp + kp];
        dW[t] = s;
    } else if (t < Cout * K + Cout) {
        const int co = t - Cout * K;
        float s = 0.f;
        for (int c = 0; c < nch; ++c) s += pdb[(long long)c * CAp + co];
        db[co] = s;
    }
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

// row layout of an activation workspace: offsets of X0|h4 (192), bias, the outputs of trunk layers 0..3 and 5, XV, HV
struct MlpRowLayout { int row, xs, bias, h[4], h5, xv, hv; };
static const MlpRowLayout kInferLayout = {WS_ROW, WS_XS, WS_BIAS, {WS_HA, WS_HB, WS_HA, WS_HB}, WS_HA, WS_XV, WS_HV};
static const MlpRowLayout kTrainLayout = {TW_ROW, TW_XS, TW_BIAS, {TW_H0, TW_H1, TW_H2, TW_H3}, TW_H5, TW_XV, TW_HV};

static int nerf_chain(const MlpRowLayout& lay, float* feat, int ldf, const float* weights, float* ws, float* raw, int M, hipStream_t st);

static int nerf_forward(const MlpRowLayout& lay, const float* ndc, float* feat, int ldf, const float* dirs, const float* w2c_ref,
                        const float* weights, float* ws, float* raw, int N, int S, hipStream_t st) {
    const int M = N * S;
    hipLaunchKernelGGL(mlp_embed_rows_kernel, dim3((M + 255) / 256), dim3(256), 0, st, ndc, dirs, w2c_ref, feat, ws, M, S, ldf, 20, lay.row, lay.xs, lay.xv);
    int rc = launch_status("nerf embed");
    if (rc) return rc;
    return nerf_chain(lay, feat, ldf, weights, ws, raw, M, st);
}

// the eleven layers on rows whose inputs (X0, feat, dir) are in place
static int nerf_chain(const MlpRowLayout& lay, float* feat, int ldf, const float* weights, float* ws, float* raw, int M, hipStream_t st) {
    const float* L[11];
    long long off = 0;
    for (int i = 0; i < 11; ++i) { L[i] = weights + off; off += mlp_layer_floats(kLayers[i].cout, kLayers[i].kp); }
    int rc;
    const int R = lay.row;
    float *XS = ws + lay.xs, *BI = ws + lay.bias, *XV = ws + lay.xv, *HV = ws + lay.hv, *H5 = ws + lay.h5;
    float* H[4] = {ws + lay.h[0], ws + lay.h[1], ws + lay.h[2], ws + lay.h[3]};
    // bias = pts_bias(feat)
    if ((rc = linear_launch<8, ACT_NONE, false>(feat, ldf, L[0], nullptr, 0, BI, R, M, 32, 128, st))) return rc;
    // trunk
    if ((rc = linear_launch<8, ACT_RELU, true>(XS, R, L[1], BI, R, H[0], R, M, 64, 128, st))) return rc;       // 0
    if ((rc = linear_launch<8, ACT_RELU, true>(H[0], R, L[2], BI, R, H[1], R, M, 128, 128, st))) return rc;    // 1
    if ((rc = linear_launch<8, ACT_RELU, true>(H[1], R, L[3], BI, R, H[2], R, M, 128, 128, st))) return rc;    // 2
    if ((rc = linear_launch<8, ACT_RELU, true>(H[2], R, L[4], BI, R, H[3], R, M, 128, 128, st))) return rc;    // 3
    if ((rc = linear_launch<8, ACT_RELU, true>(H[3], R, L[5], BI, R, XS + 64, R, M, 128, 128, st))) return rc; // 4 -> skip buffer
    if ((rc = linear_launch<8, ACT_RELU, true>(XS, R, L[6], BI, R, H5, R, M, 192, 128, st))) return rc;        // 5
    // heads
    if ((rc = linear_launch<1, ACT_RELU, false>(H5, R, L[7], nullptr, 0, raw + 3, 4, M, 128, 1, st))) return rc;   // sigma
    if ((rc = linear_launch<8, ACT_NONE, false>(H5, R, L[8], nullptr, 0, XV, R, M, 128, 128, st))) return rc;      // feature
    if ((rc = linear_launch<4, ACT_RELU, false>(XV, R, L[9], nullptr, 0, HV, R, M, 144, 64, st))) return rc;       // views
    return linear_launch<1, ACT_SIGMOID, false>(HV, R, L[10], nullptr, 0, raw, 4, M, 64, 3, st);                  // rgb
}

int rcmvs_nerf_mlp_fwd(const float* ndc, float* feat, int ldf, const float* dirs, const float* w2c_ref,
                       const float* weights, float* workspace, float* raw, int N, int S, void* stream) {
    RCMVS_REQUIRE(ndc && feat && dirs && w2c_ref && weights && workspace && raw, "nerf_mlp_fwd: null pointer");
    RCMVS_REQUIRE(N > 0 && S > 0 && ldf == 32, "nerf_mlp_fwd: feat must have a row stride of 32 floats (20 used)");
    return nerf_forward(kInferLayout, ndc, feat, ldf, dirs, w2c_ref, weights, workspace, raw, N, S, as_stream(stream));
}

/* Renderer_ours.forward(x) on its own: x (M, ldx >= 86) rows = [embedded point 63 | feature 20 | view direction 3] -> raw (M, 4) = [rgb, sigma].
 * feat32: scratch (M, 32); workspace: rcmvs_nerf_workspace_floats(M) floats. */
int rcmvs_nerf_mlp_embedded_fwd(const float* x, int ldx, const float* weights, float* workspace, float* feat32, float* raw, long long M, void* stream) {
    RCMVS_REQUIRE(x && weights && workspace && feat32 && raw, "nerf_mlp_embedded_fwd: null pointer");
    RCMVS_REQUIRE(M > 0 && M < (1LL << 31) / WS_ROW && ldx >= 86, "nerf_mlp_embedded_fwd: bad sizes (M = %lld, ldx = %d)", M, ldx);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(mlp_scatter_rows_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, x, ldx, feat32, workspace, M, 32, kInferLayout.row, kInferLayout.xs, kInferLayout.xv);
    int rc = launch_status("nerf scatter");
    if (rc) return rc;
    return nerf_chain(kInferLayout, feat32, 32, weights, workspace, raw, (int)M, st);
}

long long rcmvs_nerf_train_workspace_floats(long long M) { return M * TW_ROW; }

int rcmvs_nerf_mlp_train_fwd(const float* ndc, float* feat, int ldf, const float* dirs, const float* w2c_ref,
                             const float* weights, float* workspace, float* raw, int N, int S, void* stream) {
    RCMVS_REQUIRE(ndc && feat && dirs && w2c_ref && weights && workspace && raw, "nerf_mlp_train_fwd: null pointer");
    RCMVS_REQUIRE(N > 0 && S > 0 && ldf == 32, "nerf_mlp_train_fwd: feat must have a row stride of 32 floats (20 used)");
    return nerf_forward(kTrainLayout, ndc, feat, ldf, dirs, w2c_ref, weights, workspace, raw, N, S, as_stream(stream));
}

/* scratch of the backward pass: gradient rows, transposed weight images, partial weight gradients */
static const int kSrcK[11] = {20, 63, 128, 128, 128, 128, 191, 128, 128, 131, 64};
static const int kPart0[11] = {20, 63, 128, 128, 128, 128, 63, 128, 128, 131, 64};
static const int kPad0[11] = {0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0};
static inline long long mlp_t_floats(int i) {       // transposed image of layer i
    const int cp = (kLayers[i].cout + 15) / 16 * 16, mt = kLayers[i].kp / 16;
    return (long long)(cp / 16) * mt * 256 + mt * 16;
}
static inline int wg_chunks(long long M) { long long st = cdiv(M, WG_P); return (int)(st < 256 ? st : 256); }
long long rcmvs_nerf_bwd_workspace_floats(long long M) {
    long long n = M * GW_ROW;
    for (int i = 0; i < 11; ++i) n += mlp_t_floats(i);
    n += (long long)wg_chunks(M) * (128 * 192 + 128);
    return n;
}

/* Backward of the MLP.  wb: the 22 parameter pointers (as for rcmvs_pack_nerf_weights); tws: the workspace
 * rcmvs_nerf_mlp_train_fwd filled; raw / draw (M,4): outputs and their gradient; gws: rcmvs_nerf_bwd_workspace_floats(M)
 * floats; dfeat (M, ldf): gradient of the 20 feature columns (other columns zero); dwb: 22 output pointers, gradients in the
 * parameters' own shapes.  Every output is written (not accumulated). */
int rcmvs_nerf_mlp_bwd(const float* const* wb, const float* feat, int ldf, const float* tws, const float* raw, const float* draw,
                       float* gws, float* dfeat, float* const* dwb, int N, int S, void* stream) {
    RCMVS_REQUIRE(wb && feat && tws && raw && draw && gws && dfeat && dwb, "nerf_mlp_bwd: null pointer");
    RCMVS_REQUIRE(N > 0 && S > 0 && ldf == 32, "nerf_mlp_bwd: feat must have a row stride of 32 floats");
    hipStream_t st = as_stream(stream);
    const int M = N * S;
    int rc;
    float* gw = gws;
    float* timg[11];
    float* cur = gws + (long long)M * GW_ROW;
    for (int i = 0; i < 11; ++i) {
        RCMVS_REQUIRE(wb[2 * i] && dwb[2 * i] && dwb[2 * i + 1], "nerf_mlp_bwd: null layer pointer");
        timg[i] = cur;
        const long long n = mlp_t_floats(i);
        if (i != 1) {      // layer 0 of the trunk has no data gradient (its input is the positional encoding)
            hipLaunchKernelGGL(pack_linear_t_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, wb[2 * i], cur, kLayers[i].cout, kSrcK[i],
                               kLayers[i].kp, kPart0[i], kPad0[i]);
            if ((rc = launch_status("nerf pack_t"))) return rc;
        }
        cur += n;
    }
    const int nch = wg_chunks(M);
    const int stages = (int)cdiv(cdiv(M, WG_P), nch);
    float* P = cur;
    float* pdb = P + (long long)nch * 128 * 192;
    const int R = TW_ROW, GR = GW_ROW;
    const float *XS = tws + TW_XS, *BI = tws + TW_BIAS, *H0 = tws + TW_H0, *H1 = tws + TW_H1, *H2 = tws + TW_H2, *H3 = tws + TW_H3,
                *H4 = tws + TW_XS + 64, *H5 = tws + TW_H5, *XV = tws + TW_XV, *HV = tws + TW_HV;
    float *gXS = gw + GW_XS, *gBI = gw + GW_BIAS, *gA = gw + GW_A, *gB = gw + GW_B, *gXV = gw + GW_XV, *gHV = gw + GW_HV, *gHD = gw + GW_HEAD;
    // weight + bias gradient of layer i from dZ (A, ca4 readable columns) and the layer input (B, cb4 readable columns)
    auto wgrad = [&](int i, const float* A, int lda, int ca4, const float* B, int ldb, int cb4) -> int {
        const int CAp = (kLayers[i].cout + 15) / 16 * 16, CBp = kLayers[i].kp;
        hipLaunchKernelGGL(wgrad_kernel, dim3(nch, (CBp + 63) / 64), dim3(256), 0, st, A, lda, ca4, B, ldb, cb4, P, pdb, M, CAp, CBp, stages);
        int r = launch_status("nerf wgrad");
        if (r) return r;
        const int tot = kLayers[i].cout * kSrcK[i] + kLayers[i].cout;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, P, pdb, dwb[2 * i], dwb[2 * i + 1], nch, kLayers[i].cout,
                           kSrcK[i], CAp, CBp, kPart0[i], kPad0[i]);
        return launch_status("nerf wgrad reduce");
    };
    const dim3 grid((unsigned)cdiv(M, 256)), blk(256);
#define RCMVS_BWD(MT, RELU, ROWMUL, G, LDG, Yp, LDY, RMp, DRM, DACC, IMG, OUT, LDO, OACC, COUT)                                          \
    hipLaunchKernelGGL((linear_bwd_kernel<MT, RELU, ROWMUL>), grid, blk, 0, st, G, LDG, Yp, LDY, RMp, R, DRM, GR, DACC, IMG, OUT, LDO, OACC, M, COUT); \
    if ((rc = launch_status("nerf linear bwd"))) return rc;
    hipLaunchKernelGGL(mlp_head_bwd_kernel, grid, blk, 0, st, raw, draw, gw, M);
    if ((rc = launch_status("nerf head bwd"))) return rc;
    // rgb: dHV = dz_rgb W_r
    RCMVS_BWD(4, false, false, gHD, GR, nullptr, 0, nullptr, nullptr, 0, timg[10], gHV, GR, 0, 3)
    if ((rc = wgrad(10, gHD, GR, 4, HV, R, 64))) return rc;
    // views: dz_v = dHV [HV > 0]; dXV = dz_v W_v
    RCMVS_BWD(9, true, false, gHV, GR, HV, R, nullptr, nullptr, 0, timg[9], gXV, GR, 0, 64)
    if ((rc = wgrad(9, gHV, GR, 64, XV, R, 144))) return rc;
    // feature (no activation): dh5 = df W_f; alpha: dh5 += dz_sigma w_a
    RCMVS_BWD(8, false, false, gXV, GR, nullptr, 0, nullptr, nullptr, 0, timg[8], gA, GR, 0, 128)
    if ((rc = wgrad(8, gXV, GR, 128, H5, R, 128))) return rc;
    RCMVS_BWD(8, false, false, gHD + 4, GR, nullptr, 0, nullptr, nullptr, 0, timg[7], gA, GR, 1, 1)
    if ((rc = wgrad(7, gHD + 4, GR, 4, H5, R, 128))) return rc;
    // trunk layer 5: input [X0 | h4] (192), output h5
    RCMVS_BWD(12, true, true, gA, GR, H5, R, BI, gBI, 0, timg[6], gXS, GR, 0, 128)
    if ((rc = wgrad(6, gA, GR, 128, XS, R, 192))) return rc;
    // layer 4: dY = dXS[:, 64:], Y = h4, input h3
    RCMVS_BWD(8, true, true, gXS + 64, GR, H4, R, BI, gBI, 1, timg[5], gB, GR, 0, 128)
    if ((rc = wgrad(5, gXS + 64, GR, 128, H3, R, 128))) return rc;
    RCMVS_BWD(8, true, true, gB, GR, H3, R, BI, gBI, 1, timg[4], gA, GR, 0, 128)          // layer 3: input h2
    if ((rc = wgrad(4, gB, GR, 128, H2, R, 128))) return rc;
    RCMVS_BWD(8, true, true, gA, GR, H2, R, BI, gBI, 1, timg[3], gB, GR, 0, 128)          // layer 2: input h1
    if ((rc = wgrad(3, gA, GR, 128, H1, R, 128))) return rc;
    RCMVS_BWD(8, true, true, gB, GR, H1, R, BI, gBI, 1, timg[2], gA, GR, 0, 128)          // layer 1: input h0
    if ((rc = wgrad(2, gB, GR, 128, H0, R, 128))) return rc;
    RCMVS_BWD(0, true, true, gA, GR, H0, R, BI, gBI, 1, timg[1], nullptr, 0, 0, 128)       // layer 0: input X0, no data gradient
    if ((rc = wgrad(1, gA, GR, 128, XS, R, 64))) return rc;
    // pts_bias (no activation): dfeat = dBIAS W_b
    RCMVS_BWD(2, false, false, gBI, GR, nullptr, 0, nullptr, nullptr, 0, timg[0], dfeat, ldf, 0, 128)
#undef RCMVS_BWD
    return wgrad(0, gBI, GR, 128, feat, ldf, 32);
}

}  // extern "C"
