// Rendering-consistency branch (Rendering_Consistency_Net.forward, models/render_consist_net.py:54-76):
// plane-axis resize feeding the neural-volume U-Net, Gaussian-Uniform ray sampler, point-feature
// gathers and volumetric compositing.  The NeRF MLP lives in nerf_mlp.hip.  gfx950 only.
#include "common.h"

namespace rcmvs {

typedef float v4f_r __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// F.interpolate(size=[Do,h,w], mode='trilinear', align_corners=True) when only the plane axis changes
// (models/render_models.py:756): src = dst*(D-1)/(Do-1), lerp of the two neighbouring planes.
// NCDHW in, channels-last out with the channel count padded to Cp (zeros) so that the first 3-D
// conv reads 16-byte channel vectors.  One thread per (b, do, y, x).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resize_planes_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             int C, int Cp, int D, int Do, int hw) {
#pragma clang fp contract(off)
    // The NCDHW reads are coalesced per channel (consecutive pixels); the channels-last rows of the output are Cp floats apart, so
    // writing them from the thread that owns the pixel touches 64 lines per store instruction (1.44 ms for the 461 MB output of
    // config 3).  The block's 256 x Cp outputs are one contiguous chunk: stage them in LDS (row stride Cp + 1: conflict-free both
    // ways) and write the chunk with consecutive lanes on consecutive float4.
    extern __shared__ float rsz_slab[];                             // [256][Cp + 1]
    const int b = blockIdx.z, od = blockIdx.y;
    const int p0 = blockIdx.x * 256, p = p0 + threadIdx.x;
    const int npix = min(256, hw - p0);
    const int ld = Cp + 1;
    const float scale = (Do > 1) ? (float)(D - 1) / (float)(Do - 1) : 0.0f;
    const float srcf = scale * (float)od;
    int i0 = (int)srcf;
    i0 = i0 > D - 1 ? D - 1 : i0;
    const int i1 = i0 + 1 > D - 1 ? D - 1 : i0 + 1;
    const float l1 = srcf - (float)i0, l0 = 1.0f - l1;
    const float* xb = x + (long long)b * C * D * hw;
    if (p < hw) {
        float* row = rsz_slab + threadIdx.x * ld;
        for (int c = 0; c < Cp; ++c) {
            float v = 0.0f;
            if (c < C) v = l0 * xb[((long long)c * D + i0) * hw + p] + l1 * xb[((long long)c * D + i1) * hw + p];
            row[c] = v;
        }
    }
    __syncthreads();
    float* yo = y + (((long long)b * Do + od) * hw + p0) * Cp;       // the block's contiguous chunk: npix * Cp floats
    if ((Cp & 3) == 0) {
        const int q = Cp >> 2;
        for (int e = threadIdx.x; e < npix * q; e += 256) {
            const int pp = e / q, c = (e - pp * q) * 4;
            const float* r = rsz_slab + pp * ld + c;
            *reinterpret_cast<float4*>(yo + (long long)e * 4) = make_float4(r[0], r[1], r[2], r[3]);
        }
    } else {
        for (int e = threadIdx.x; e < npix * Cp; e += 256) yo[e] = rsz_slab[(e / Cp) * ld + e % Cp];
    }
}

// backward of resize_planes_kernel w.r.t. its input: g (B,Do,hw,ldg) channels-last -> gx (B,C,D,hw) NCDHW.
// One block per (b, input plane j, 256 pixels), one thread per pixel: every output plane whose two source planes
// include j contributes with the forward's weight (same float arithmetic), so the adjoint is exact and needs no
// atomics.  Each contributing plane's (256 x ldg) slab is staged through LDS with coalesced reads (the channels-last
// rows are ldg floats apart) and the NCDHW writes are coalesced per channel.
constexpr int RSZ_MAXC = 64;
__global__ __launch_bounds__(256) void resize_planes_bwd_kernel(const float* __restrict__ g, float* __restrict__ gx,
                                                                 int C, int ldg, int D, int Do, int hw) {
#pragma clang fp contract(off)
    extern __shared__ float slab[];                                // [256][ldg + 1]
    const int b = blockIdx.z, j = blockIdx.y;
    const int p0 = blockIdx.x * 256;
    const int npix = min(256, hw - p0);
    const int lds_ld = ldg + 1;
    const float scale = (Do > 1) ? (float)(D - 1) / (float)(Do - 1) : 0.0f;
    int klo = 0, khi = Do - 1;
    if (scale > 0.0f) {
        klo = (int)floorf((float)(j - 1) / scale) - 1;
        khi = (int)ceilf((float)(j + 1) / scale) + 1;
        klo = klo < 0 ? 0 : klo;
        khi = khi > Do - 1 ? Do - 1 : khi;
    }
    float acc[RSZ_MAXC];
#pragma unroll
    for (int c = 0; c < RSZ_MAXC; ++c) acc[c] = 0.0f;
    for (int k = klo; k <= khi; ++k) {
        const float srcf = scale * (float)k;
        int i0 = (int)srcf;
        i0 = i0 > D - 1 ? D - 1 : i0;
        const int i1 = i0 + 1 > D - 1 ? D - 1 : i0 + 1;
        if (i0 != j && i1 != j) continue;                           // block-uniform
        const float l1 = srcf - (float)i0, l0 = 1.0f - l1;
        __syncthreads();
        const float* gs = g + (((long long)b * Do + k) * hw + p0) * ldg;
        for (int e = threadIdx.x; e < npix * ldg; e += 256) slab[(e / ldg) * lds_ld + (e % ldg)] = gs[e];
        __syncthreads();
        const float* row = slab + threadIdx.x * lds_ld;
#pragma unroll
        for (int c = 0; c < RSZ_MAXC; ++c) {
            if (c < C) {
                const float v = row[c];
                if (i0 == j) acc[c] = acc[c] + l0 * v;
                if (i1 == j) acc[c] = acc[c] + l1 * v;
            }
        }
    }
    if ((int)threadIdx.x < npix) {
        float* go = gx + ((long long)b * C * D + j) * hw + p0 + threadIdx.x;
#pragma unroll
        for (int c = 0; c < RSZ_MAXC; ++c)
            if (c < C) go[(long long)c * D * hw] = acc[c];
    }
}

// ------------------------------------------------------------------------------------------------
// Gaussian-Uniform sampler (models/render_utils.py:86-108,149-243,112-146).  One 128-thread block per
// ray; samples are sorted with an LDS bitonic network (power-of-two S <= 1024 handled by padding
// with +inf).  cam = [K(9) | c2w(16) | w2c_ref(16) | K_ref(9) | near | far].
// ------------------------------------------------------------------------------------------------
constexpr int GU_THREADS = 128;

__device__ __forceinline__ float torch_linspace01(int i, int steps) {
    // at::linspace(0, 1, steps): step = 1/(steps-1); symmetric evaluation (RangeFactories kernel)
    const float step = 1.0f / (float)(steps - 1);
    return (i < steps / 2) ? step * (float)i : 1.0f - step * (float)(steps - 1 - i);
}

__global__ __launch_bounds__(GU_THREADS) void gu_sample_kernel(
    const float* __restrict__ pseudo_depth, const float* __restrict__ img0, const int* __restrict__ pix,
    const float* __restrict__ eps, const float* __restrict__ u, const float* __restrict__ cam,
    float* __restrict__ z, float* __restrict__ pts, float* __restrict__ ndc, float* __restrict__ dirs,
    float* __restrict__ rays_depth, float* __restrict__ target, int N, int S, int H, int W, int SP) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float sbuf[];   // [SP]
    const int ray = blockIdx.x;
    const int px = pix[ray], py = pix[N + ray];
    const float* K = cam;
    const float* c2w = cam + 9;
    const float* w2c = cam + 25;
    const float* Kr = cam + 41;
    const float near = cam[50], far = cam[51];
    const float mu = pseudo_depth[(long long)py * W + px];
    // ray direction: [(x-cx)/fx, (y-cy)/fy, 1] @ c2w[:3,:3]^T ; origin c2w[:3,3]
    const float d0 = ((float)px - K[2]) / K[0], d1 = ((float)py - K[5]) / K[4], d2 = 1.0f;
    const float rdx = (d0 * c2w[0] + d1 * c2w[1]) + d2 * c2w[2];
    const float rdy = (d0 * c2w[4] + d1 * c2w[5]) + d2 * c2w[6];
    const float rdz = (d0 * c2w[8] + d1 * c2w[9]) + d2 * c2w[10];
    const float ox = c2w[3], oy = c2w[7], oz = c2w[11];
    if (threadIdx.x == 0) {
        dirs[ray * 3 + 0] = rdx; dirs[ray * 3 + 1] = rdy; dirs[ray * 3 + 2] = rdz;
        rays_depth[ray] = mu;
        for (int c = 0; c < 3; ++c) target[ray * 3 + c] = img0[((long long)c * H + py) * W + px];
    }
    const bool gaussian = ray < N / 2;
    if (gaussian) {
        const float sigma = fminf(fabsf(far - mu), fabsf(mu - near)) / 3.0f;
        for (int i = threadIdx.x; i < SP; i += GU_THREADS)
            sbuf[i] = (i < S) ? mu + sigma * eps[(long long)ray * S + i] : INFINITY;
        __syncthreads();
        for (int k = 2; k <= SP; k <<= 1) {                    // bitonic sort, ascending
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = threadIdx.x; i < SP; i += GU_THREADS) {
                    const int l = i ^ j;
                    if (l > i) {
                        const float a = sbuf[i], bb = sbuf[l];
                        const bool up = ((i & k) == 0);
                        if ((a > bb) == up) { sbuf[i] = bb; sbuf[l] = a; }
                    }
                }
                __syncthreads();
            }
        }
    } else {
        for (int i = threadIdx.x; i < S; i += GU_THREADS) {
            const float t = torch_linspace01(i, S);
            const float lin = near * (1.0f - t) + far * t;
            float lo = lin, hi = lin;
            if (i > 0) { const float tm = torch_linspace01(i - 1, S); lo = 0.5f * (lin + (near * (1.0f - tm) + far * tm)); }
            if (i < S - 1) { const float tp = torch_linspace01(i + 1, S); hi = 0.5f * ((near * (1.0f - tp) + far * tp) + lin); }
            sbuf[i] = lo + (hi - lo) * u[(long long)(ray - N / 2) * S + i];
        }
        __syncthreads();
    }
    const float inv_w = (float)(W - 1), inv_h = (float)(H - 1);
    for (int i = threadIdx.x; i < S; i += GU_THREADS) {
        const float zz = sbuf[i];
        const long long o = (long long)ray * S + i;
        z[o] = zz;
        const float wx = ox + zz * rdx, wy = oy + zz * rdy, wz = oz + zz * rdz;
        pts[o * 3 + 0] = wx; pts[o * 3 + 1] = wy; pts[o * 3 + 2] = wz;
        // get_ndc_coordinate: p = pts @ R^T + T ; q = p @ K^T ; xy / z / (W-1,H-1) ; (z-near)/(far-near)
        const float cx = ((wx * w2c[0] + wy * w2c[1]) + wz * w2c[2]) + w2c[3];
        const float cy = ((wx * w2c[4] + wy * w2c[5]) + wz * w2c[6]) + w2c[7];
        const float cz = ((wx * w2c[8] + wy * w2c[9]) + wz * w2c[10]) + w2c[11];
        const float qx = (cx * Kr[0] + cy * Kr[1]) + cz * Kr[2];
        const float qy = (cx * Kr[3] + cy * Kr[4]) + cz * Kr[5];
        const float qz = (cx * Kr[6] + cy * Kr[7]) + cz * Kr[8];
        ndc[o * 3 + 0] = (qx / qz + 0.0f) / inv_w;
        ndc[o * 3 + 1] = (qy / qz + 0.0f) / inv_h;
        ndc[o * 3 + 2] = (qz - near) / (far - near);
    }
}

// ------------------------------------------------------------------------------------------------
// Point features (models/renderer.py:154-166; render_utils.py:247-279,304-330): per point
//   [0:8)        trilinear sample of the 8-channel volume (channels-last (Dv,hv,wv,8)) at ndc*2-1,
//                zeros padding, align_corners=True;
//   [8+4i:12+4i) bilinear (border padding) RGB of image i at the point's projection with pose i, and
//                the strict in-bounds mask.
// One thread per point; ldf = row stride of `feat` in floats (>= 8 + 4*nimg; extra columns untouched).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void point_feats_kernel(
    const float* __restrict__ vol, const float* __restrict__ imgs, const float* __restrict__ poses,
    const float* __restrict__ pts, const float* __restrict__ ndc, float* __restrict__ feat,
    int M, int Dv, int hv, int wv, int nimg, int H, int W, int ldf) {
#pragma clang fp contract(off)
    int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float* fo = feat + (long long)m * ldf;
    {
        const float gx = ndc[m * 3 + 0] * 2.0f - 1.0f, gy = ndc[m * 3 + 1] * 2.0f - 1.0f, gz = ndc[m * 3 + 2] * 2.0f - 1.0f;
        const float ix = ((gx + 1.0f) / 2.0f) * (float)(wv - 1);
        const float iy = ((gy + 1.0f) / 2.0f) * (float)(hv - 1);
        const float iz = ((gz + 1.0f) / 2.0f) * (float)(Dv - 1);
        const float x0 = floorf(ix), y0 = floorf(iy), z0 = floorf(iz);
        float acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = 0.0f;
        for (int dz = 0; dz < 2; ++dz)
            for (int dy = 0; dy < 2; ++dy)
                for (int dx = 0; dx < 2; ++dx) {
                    const float xx = x0 + dx, yy = y0 + dy, zz = z0 + dz;
                    const float wx = dx ? (ix - x0) : ((x0 + 1.0f) - ix);
                    const float wy = dy ? (iy - y0) : ((y0 + 1.0f) - iy);
                    const float wz = dz ? (iz - z0) : ((z0 + 1.0f) - iz);
                    const bool ok = (xx >= 0.0f) && (xx <= (float)(wv - 1)) && (yy >= 0.0f) && (yy <= (float)(hv - 1)) &&
                                    (zz >= 0.0f) && (zz <= (float)(Dv - 1));
                    if (!ok) continue;
                    const float wgt = (wx * wy) * wz;
                    const float* vp = vol + (((long long)(int)zz * hv + (int)yy) * wv + (int)xx) * 8;
                    const float4 a = *reinterpret_cast<const float4*>(vp);
                    const float4 b = *reinterpret_cast<const float4*>(vp + 4);
                    acc[0] = acc[0] + a.x * wgt; acc[1] = acc[1] + a.y * wgt; acc[2] = acc[2] + a.z * wgt; acc[3] = acc[3] + a.w * wgt;
                    acc[4] = acc[4] + b.x * wgt; acc[5] = acc[5] + b.y * wgt; acc[6] = acc[6] + b.z * wgt; acc[7] = acc[7] + b.w * wgt;
                }
#pragma unroll
        for (int c = 0; c < 8; ++c) fo[c] = acc[c];
    }
    const float wx_ = pts[m * 3 + 0], wy_ = pts[m * 3 + 1], wz_ = pts[m * 3 + 2];
    const float sw = (float)(W - 1), sh = (float)(H - 1);
    for (int i = 0; i < nimg; ++i) {
        const float* w2c = poses + i * 25;
        const float* K = w2c + 16;
        const float cx = ((wx_ * w2c[0] + wy_ * w2c[1]) + wz_ * w2c[2]) + w2c[3];
        const float cy = ((wx_ * w2c[4] + wy_ * w2c[5]) + wz_ * w2c[6]) + w2c[7];
        const float cz = ((wx_ * w2c[8] + wy_ * w2c[9]) + wz_ * w2c[10]) + w2c[11];
        const float qx = (cx * K[0] + cy * K[1]) + cz * K[2];
        const float qy = (cx * K[3] + cy * K[4]) + cz * K[5];
        const float qz = (cx * K[6] + cy * K[7]) + cz * K[8];
        const float gx = ((qx / qz + 0.0f) / sw) * 2.0f - 1.0f;
        const float gy = ((qy / qz + 0.0f) / sh) * 2.0f - 1.0f;
        const float mask = ((gx > -1.0f) && (gx < 1.0f) && (gy > -1.0f) && (gy < 1.0f)) ? 1.0f : 0.0f;
        // border padding: clip the pixel coordinate, then plain bilinear
        float ix = ((gx + 1.0f) / 2.0f) * sw, iy = ((gy + 1.0f) / 2.0f) * sh;
        ix = fminf(fmaxf(ix, 0.0f), sw);
        iy = fminf(fmaxf(iy, 0.0f), sh);
        if (!(ix == ix)) ix = 0.0f;                              // NaN (qz == 0) -> ATen clips NaN to 0 via min/max order
        if (!(iy == iy)) iy = 0.0f;
        const float x0 = floorf(ix), y0 = floorf(iy);
        const float w1x = ix - x0, w0x = (x0 + 1.0f) - ix, w1y = iy - y0, w0y = (y0 + 1.0f) - iy;
        const int xi = (int)x0, yi = (int)y0;
        const bool x1ok = xi + 1 <= W - 1, y1ok = yi + 1 <= H - 1;
        const float* ib = imgs + (long long)i * 3 * H * W;
        for (int c = 0; c < 3; ++c) {
            const float* pl = ib + (long long)c * H * W;
            float v = pl[(long long)yi * W + xi] * (w0x * w0y);
            v = v + (x1ok ? pl[(long long)yi * W + xi + 1] * (w1x * w0y) : 0.0f);
            v = v + (y1ok ? pl[(long long)(yi + 1) * W + xi] * (w0x * w1y) : 0.0f);
            v = v + ((x1ok && y1ok) ? pl[(long long)(yi + 1) * W + xi + 1] * (w1x * w1y) : 0.0f);
            fo[8 + 4 * i + c] = v;
        }
        fo[8 + 4 * i + 3] = mask;
    }
}

// ------------------------------------------------------------------------------------------------
// Compositing (models/renderer.py:18-26,65-93): alpha = 1-exp(-sigma); T = exclusive cumprod of
// (1 - alpha + 1e-10); w = alpha*T; rgb = sum w*c; depth = sum w*z.  One wave (64 lanes) per ray:
// each lane owns a contiguous run of samples, multiplies it locally, an inclusive wave scan with
// __shfl_up gives the prefix of the runs before it, and the final sums are wave reductions.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void composite_kernel(const float* __restrict__ raw, const float* __restrict__ z,
                                                         float* __restrict__ rgb, float* __restrict__ depth,
                                                         float* __restrict__ weights, float* __restrict__ alpha, int N, int S) {
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= N) return;
    const int per = (S + 63) / 64;                    // samples per lane (2 for S = 128)
    const int s0 = lane * per;
    float prod = 1.0f;
    for (int i = 0; i < per; ++i) {
        const int sidx = s0 + i;
        if (sidx < S) {
            const float a = 1.0f - expf(-raw[((long long)ray * S + sidx) * 4 + 3]);
            prod *= (1.0f - a + 1e-10f);
        }
    }
    // inclusive scan of lane products, then shift to exclusive
    float scan = prod;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float t = __shfl_up(scan, off);
        if (lane >= off) scan *= t;
    }
    float T = __shfl_up(scan, 1);
    if (lane == 0) T = 1.0f;
    float r = 0.f, g = 0.f, bch = 0.f, dsum = 0.f;
    for (int i = 0; i < per; ++i) {
        const int sidx = s0 + i;
        if (sidx < S) {
            const long long o = (long long)ray * S + sidx;
            const float4 c = *reinterpret_cast<const float4*>(raw + o * 4);
            const float a = 1.0f - expf(-c.w);
            const float wgt = a * T;
            alpha[o] = a;
            weights[o] = wgt;
            r += wgt * c.x; g += wgt * c.y; bch += wgt * c.z; dsum += wgt * z[o];
            T *= (1.0f - a + 1e-10f);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        r += __shfl_xor(r, off); g += __shfl_xor(g, off); bch += __shfl_xor(bch, off); dsum += __shfl_xor(dsum, off);
    }
    if (lane == 0) {
        rgb[ray * 3 + 0] = r; rgb[ray * 3 + 1] = g; rgb[ray * 3 + 2] = bch;
        depth[ray] = dsum;
    }
}

// ------------------------------------------------------------------------------------------------
// Backward of the rendering tail (what autograd records for renderer.py:18-26,65-93 and for the trilinear
// grid_sample of render_utils.py:304-330).
// composite_bwd: one thread per ray, two serial passes over its S samples.  With t_j = 1 - alpha_j + 1e-10,
//   T_i = prod_{j<i} t_j, w_i = alpha_i T_i and G_i = g_w_i + g_rgb . c_i + g_depth z_i:
//   d/d c_i = w_i g_rgb,   d/d alpha_i = g_alpha_i + T_i (G_i - Q_i),   Q_i = G_{i+1} alpha_{i+1} + t_{i+1} Q_{i+1}
//   (division-free reverse recurrence: no 1/t blow-up where alpha -> 1),   d/d sigma_i = (1 - alpha_i) d/d alpha_i.
// T_i of the forward pass is parked in graw[...,3] between the passes.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void composite_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ z,
                                                           const float* __restrict__ g_rgb, const float* __restrict__ g_depth,
                                                           const float* __restrict__ g_w, const float* __restrict__ g_alpha,
                                                           float* __restrict__ graw, int N, int S) {
    const int ray = blockIdx.x * 64 + threadIdx.x;
    if (ray >= N) return;
    const float gr = g_rgb ? g_rgb[ray * 3 + 0] : 0.f, gg = g_rgb ? g_rgb[ray * 3 + 1] : 0.f, gb = g_rgb ? g_rgb[ray * 3 + 2] : 0.f;
    const float gd = g_depth ? g_depth[ray] : 0.f;
    const long long base = (long long)ray * S;
    float T = 1.0f;
    for (int i = 0; i < S; ++i) {
        const float a = 1.0f - expf(-raw[(base + i) * 4 + 3]);
        graw[(base + i) * 4 + 3] = T;
        T *= (1.0f - a + 1e-10f);
    }
    float Q = 0.0f;
    for (int i = S - 1; i >= 0; --i) {
        const long long o = base + i;
        const float4 c = *reinterpret_cast<const float4*>(raw + o * 4);
        const float a = 1.0f - expf(-c.w);
        const float Ti = graw[o * 4 + 3];
        const float wgt = a * Ti;
        const float G = (g_w ? g_w[o] : 0.f) + (gr * c.x + gg * c.y + gb * c.z) + gd * z[o];
        const float da = (g_alpha ? g_alpha[o] : 0.f) + Ti * (G - Q);
        *reinterpret_cast<float4*>(graw + o * 4) = make_float4(wgt * gr, wgt * gg, wgt * gb, (1.0f - a) * da);
        Q = G * a + (1.0f - a + 1e-10f) * Q;
    }
}

// point_feats_bwd: gradient of the 8 volume channels of every point, scattered to the 8 trilinear corners with the
// forward's weights (hardware fp32 atomics into the zero-filled volume gradient).  The image taps, the mask and the
// point coordinates carry no gradient (inputs / no_grad in the reference).
__global__ __launch_bounds__(256) void point_feats_bwd_kernel(const float* __restrict__ ndc, const float* __restrict__ gfeat,
                                                              float* __restrict__ gvol, int M, int Dv, int hv, int wv, int ldg) {
#pragma clang fp contract(off)
    int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const float gx = ndc[m * 3 + 0] * 2.0f - 1.0f, gy = ndc[m * 3 + 1] * 2.0f - 1.0f, gz = ndc[m * 3 + 2] * 2.0f - 1.0f;
    const float ix = ((gx + 1.0f) / 2.0f) * (float)(wv - 1);
    const float iy = ((gy + 1.0f) / 2.0f) * (float)(hv - 1);
    const float iz = ((gz + 1.0f) / 2.0f) * (float)(Dv - 1);
    const float x0 = floorf(ix), y0 = floorf(iy), z0 = floorf(iz);
    float g[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) g[c] = gfeat[(long long)m * ldg + c];
    for (int dz = 0; dz < 2; ++dz)
        for (int dy = 0; dy < 2; ++dy)
            for (int dx = 0; dx < 2; ++dx) {
                const float xx = x0 + dx, yy = y0 + dy, zz = z0 + dz;
                const float wx = dx ? (ix - x0) : ((x0 + 1.0f) - ix);
                const float wy = dy ? (iy - y0) : ((y0 + 1.0f) - iy);
                const float wz = dz ? (iz - z0) : ((z0 + 1.0f) - iz);
                const bool ok = (xx >= 0.0f) && (xx <= (float)(wv - 1)) && (yy >= 0.0f) && (yy <= (float)(hv - 1)) &&
                                (zz >= 0.0f) && (zz <= (float)(Dv - 1));
                if (!ok) continue;
                const float wgt = (wx * wy) * wz;
                float* vp = gvol + (((long long)(int)zz * hv + (int)yy) * wv + (int)xx) * 8;
#pragma unroll
                for (int c = 0; c < 8; ++c) unsafeAtomicAdd(vp + c, g[c] * wgt);
            }
}

}  // namespace rcmvs

using namespace rcmvs;

extern "C" {

int rcmvs_resize_planes_fwd(const float* x, float* y, int B, int C, int Cp, int D, int Do, int h, int w, void* stream) {
    RCMVS_REQUIRE(x && y, "resize_planes_fwd: null pointer");
    RCMVS_REQUIRE(B > 0 && C > 0 && Cp >= C && D > 0 && Do > 0 && h > 0 && w > 0, "resize_planes_fwd: bad sizes");
    dim3 grid((h * w + 255) / 256, Do, B);
    RCMVS_REQUIRE(Cp <= 63, "resize_planes_fwd: at most 63 (padded) channels (the block stages 256 x (Cp + 1) floats in LDS)");
    hipLaunchKernelGGL(resize_planes_kernel, grid, dim3(256), (size_t)256 * (Cp + 1) * sizeof(float), as_stream(stream), x, y, C, Cp, D, Do, h * w);
    return launch_status("resize_planes_fwd");
}

int rcmvs_resize_planes_bwd(const float* g, float* gx, int B, int C, int ldg, int D, int Do, int h, int w, void* stream) {
    RCMVS_REQUIRE(g && gx, "resize_planes_bwd: null pointer");
    RCMVS_REQUIRE(B > 0 && C > 0 && ldg >= C && D > 0 && Do > 0 && h > 0 && w > 0, "resize_planes_bwd: bad sizes");
    RCMVS_REQUIRE(C <= RSZ_MAXC && ldg <= 63, "resize_planes_bwd: at most %d channels (row stride <= 63)", RSZ_MAXC);
    dim3 grid((h * w + 255) / 256, D, B);
    hipLaunchKernelGGL(resize_planes_bwd_kernel, grid, dim3(256), (size_t)256 * (ldg + 1) * sizeof(float), as_stream(stream), g, gx, C,
                       ldg, D, Do, h * w);
    return launch_status("resize_planes_bwd");
}

int rcmvs_gu_sample_fwd(const float* pseudo_depth, const float* img0, const int* pix,
                        const float* eps, const float* u, const float* cam,
                        float* z, float* pts, float* ndc, float* dirs, float* rays_depth, float* target,
                        int N, int S, int H, int W, void* stream) {
    RCMVS_REQUIRE(pseudo_depth && img0 && pix && eps && u && cam && z && pts && ndc && dirs && rays_depth && target,
                  "gu_sample_fwd: null pointer");
    RCMVS_REQUIRE(N > 0 && (N % 2) == 0 && S >= 2 && S <= 4096 && H > 1 && W > 1, "gu_sample_fwd: bad sizes N=%d S=%d", N, S);
    int SP = 1;
    while (SP < S) SP <<= 1;
    hipLaunchKernelGGL(gu_sample_kernel, dim3(N), dim3(GU_THREADS), SP * sizeof(float), as_stream(stream), pseudo_depth, img0,
                       pix, eps, u, cam, z, pts, ndc, dirs, rays_depth, target, N, S, H, W, SP);
    return launch_status("gu_sample_fwd");
}

int rcmvs_point_feats_fwd(const float* volume, const float* imgs, const float* poses,
                          const float* pts, const float* ndc, float* feat,
                          int M, int Dv, int hv, int wv, int nimg, int H, int W, int ldf, void* stream) {
    RCMVS_REQUIRE(volume && imgs && poses && pts && ndc && feat, "point_feats_fwd: null pointer");
    RCMVS_REQUIRE(M > 0 && Dv > 0 && hv > 0 && wv > 0 && nimg >= 0 && H > 1 && W > 1 && ldf >= 8 + 4 * nimg,
                  "point_feats_fwd: bad sizes");
    hipLaunchKernelGGL(point_feats_kernel, dim3((M + 255) / 256), dim3(256), 0, as_stream(stream), volume, imgs, poses, pts,
                       ndc, feat, M, Dv, hv, wv, nimg, H, W, ldf);
    return launch_status("point_feats_fwd");
}

int rcmvs_point_feats_bwd(const float* ndc, const float* grad_feat, float* grad_volume,
                          int M, int Dv, int hv, int wv, int ldg, void* stream) {
    RCMVS_REQUIRE(ndc && grad_feat && grad_volume, "point_feats_bwd: null pointer");
    RCMVS_REQUIRE(M > 0 && Dv > 0 && hv > 0 && wv > 0 && ldg >= 8, "point_feats_bwd: bad sizes");
    hipLaunchKernelGGL(point_feats_bwd_kernel, dim3((M + 255) / 256), dim3(256), 0, as_stream(stream), ndc, grad_feat, grad_volume,
                       M, Dv, hv, wv, ldg);
    return launch_status("point_feats_bwd");
}

int rcmvs_composite_bwd(const float* raw, const float* z, const float* grad_rgb, const float* grad_depth,
                        const float* grad_weights, const float* grad_alpha, float* grad_raw, int N, int S, void* stream) {
    RCMVS_REQUIRE(raw && z && grad_raw, "composite_bwd: null pointer");
    RCMVS_REQUIRE(N > 0 && S > 0, "composite_bwd: bad sizes");
    hipLaunchKernelGGL(composite_bwd_kernel, dim3((N + 63) / 64), dim3(64), 0, as_stream(stream), raw, z, grad_rgb, grad_depth,
                       grad_weights, grad_alpha, grad_raw, N, S);
    return launch_status("composite_bwd");
}

int rcmvs_composite_fwd(const float* raw, const float* z, float* rgb, float* depth,
                        float* weights, float* alpha, int N, int S, void* stream) {
    RCMVS_REQUIRE(raw && z && rgb && depth && weights && alpha, "composite_fwd: null pointer");
    RCMVS_REQUIRE(N > 0 && S > 0, "composite_fwd: bad sizes");
    hipLaunchKernelGGL(composite_kernel, dim3((N + 3) / 4), dim3(256), 0, as_stream(stream), raw, z, rgb, depth, weights,
                       alpha, N, S);
    return launch_status("composite_fwd");
}

}  // extern "C"
