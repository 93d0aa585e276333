// Depth head on the matrix cores: the prob conv (8 -> 1, 3x3x3; models/modules.py:489,500) in the fp16-pair arithmetic of
// conv3d_x3.hip (two fp16 pieces per operand after an exact power-of-two pre-scale, v_mfma_f32_16x16x32_f16, three MFMAs per
// product), for callers that hand over a bound of max|x| -- the B = 1 inference scene.  gfx950 only.
//
// Why.  The VALU form (conv3d_lds.hip, prob_conv_march_plain_kernel) needs 108 v_pk_fma_f32 (4.7 issue clocks each), 18 ds_read_b128 and 18
// scalar weight loads per plane and wave and runs at ~1400 clocks per plane and wave (profiles/r4_prob_pair.txt).
// One output channel is a poor GEMM (M = 1), so the GEMM here is built differently:
//   M = (kd, kw)   -- the NINE partial sums  P[kd][kw][h][q] = sum_{kh, c} x[z][h + kh][q][c] W[kd][kh][kw][c]  of an input plane z,
//                     one per output plane it feeds (kd) and per column shift still to be applied (kw); rows m = 4 g + kw, g = lane group,
//   N = 16 halo columns q of one row,   K = (kh, c) = 24 of 32.
// One MFMA triple per (row, 16 columns) and plane: 18 per wave and plane = 288 clocks of the matrix pipe instead of ~510 VALU clocks of packed FMAs.
// The output is  out[z + 1 - kd][h][w] = sum_kw P[kd][kw][h][w + kw]  (halo column q = w + 1 - 1 + kw): two DPP row shifts and two adds;
// n-tiles start every 14 columns so that the shifted lanes stay inside a 16-lane row.
// The three kd of a plane belong to three different output planes.  The weight image exists in three ROTATIONS (plane i of a block
// uses rotation i mod 3, which puts kd into lane group (kd - i) mod 3): a given output plane then always accumulates in the same
// lane group, the MFMA's own C input does the rolling sum over the three input planes, and after plane i the group (2 - i) mod 3
// holds the finished output plane i - 2, is read out and cleared.
// Staging as in the VALU form (a block owns an 8 x 32 pixel tile and marches over z, planes double-buffered in LDS, next plane
// prefetched into registers), except that a plane is split into its two fp16 pieces on the way into LDS (10.9 KB instead of 16.3).
// FUSE_D = 8 (the cascade's last stage): the logits go to LDS instead of memory and the softmax / soft-argmin / confidence of
// depth_head.hip finish in the same launch.
#include "common.h"
#include "x3_pieces.h"
#include <type_traits>

namespace rcmvs {

constexpr int PP_TH = 8, PP_TW = 32, PP_HH = PP_TH + 2, PP_HW = PP_TW + 2;
constexpr int PP_PIECE = PP_HH * PP_HW * 16;               // bytes of one piece plane (8 fp16 per voxel)
constexpr int PP_BUF = 2 * PP_PIECE;                       // hi + lo
constexpr int PP_NLD = (PP_HH * PP_HW * 2 + 255) / 256;    // float4 per thread per plane
constexpr int PP_NJ = 3, PP_STEP = 14;                     // n-tiles per row, columns between their starts

long long prob_pair_weight_floats() { return 4 + 3 * 2 * 64 * 4; }        // header + [rotation][piece][lane][8 fp16]

// w (1, 8, 3, 3, 3) -> the three rotated A fragments (row = lane & 15 = 4 g + kw, k = 8 (lane >> 4) + c = (kh, c)); wsc as in conv3d_x3.hip
__global__ void prob_pair_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ img, const float* __restrict__ wsc) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 3 * 64 * 8) return;
    const int c = t & 7, lane = (t >> 3) & 63, r = t >> 9;
    const int m = lane & 15, kh = lane >> 4, g = m >> 2, kw = m & 3;
    float v = 0.0f;
    if (g < 3 && kw < 3 && kh < 3) {
        const int kd = (g + r) % 3;
        v = w[c * 27 + (kd * 3 + kh) * 3 + kw];
    }
    const float sw = wsc[0];
    if (t == 0) { float* hdr = reinterpret_cast<float*>(img); hdr[0] = sw; hdr[1] = wsc[1]; hdr[2] = 0.0f; hdr[3] = 0.0f; }
    const float vs = v * sw;
    const _Float16 h = (_Float16)vs;
    const _Float16 l = (_Float16)(vs - (float)h);
    const int base = 8 + ((r * 2) * 64 + lane) * 8 + c;
    img[base] = __builtin_bit_cast(unsigned short, h);
    img[base + 64 * 8] = __builtin_bit_cast(unsigned short, l);
}

int prob_pair_pack(const float* w, float* img, const float* wsc, hipStream_t st) {
    hipLaunchKernelGGL(prob_pair_pack_kernel, dim3(6), dim3(256), 0, st, w, reinterpret_cast<unsigned short*>(img), wsc);
    return launch_status("prob_pair_pack");
}

template <int FUSE_D>
__global__ __launch_bounds__(256, 4) void prob_pair_kernel(
    const float* __restrict__ x, const x3_u32x4* __restrict__ wimg, const float* __restrict__ xmax, float* __restrict__ y,
    int D, int H, int W, int tiles_w, int tiles_h, int zchunk,
    const float* __restrict__ planes, float* __restrict__ depth, float* __restrict__ conf) {
    constexpr int LGB = FUSE_D > 0 ? FUSE_D * 256 * 4 : 0;
    __shared__ __attribute__((aligned(16))) x3_byte smem[2 * PP_BUF + LGB + 16];
    constexpr int OOB = 0x7ffffff0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const int b = blockIdx.z, zc = blockIdx.y;
    const unsigned t2 = xcd_remap(blockIdx.x, gridDim.x);
    const int tw = t2 % tiles_w, th = t2 / tiles_w;
    const int h0 = th * PP_TH, w0 = tw * PP_TW, z0 = zc * zchunk;
    const int z1 = min(D, z0 + zchunk);                          // outputs z0 .. z1-1, input planes z0-1 .. z1
    const int nplanes = z1 - z0 + 2;
    const int plane_bytes = H * W * 32;
    const float xmax_lane = xmax[lane * 16];
    const float whdr = reinterpret_cast<const float*>(wimg)[1];
    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (long long)b * D * H * W * 8), (short)0, D * plane_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(y ? y + (long long)b * D * H * W : nullptr, (short)0, y ? D * H * W * 4 : 0, 0x00020000);
    // this thread's share of a plane's halo: element e = (halo voxel, float4 half)
    int goff[PP_NLD], ls[PP_NLD];
#pragma unroll
    for (int i = 0; i < PP_NLD; ++i) {
        const int e = tid + i * 256;
        const int v = e >> 1, c4 = e & 1;
        const int hh = v / PP_HW, hw_ = v - hh * PP_HW;
        const int ih = h0 + hh - 1, iw = w0 + hw_ - 1;
        const bool has = e < PP_HH * PP_HW * 2;
        goff[i] = (has && ih >= 0 && ih < H && iw >= 0 && iw < W) ? ((ih * W + iw) * 8 + c4 * 4) * 4 : OOB;
        ls[i] = has ? v * 16 + c4 * 8 : -1;
    }
    x3_u32x4 pf[PP_NLD];
    auto fetch = [&](int z) {
        const bool zin = z >= 0 && z < D;
        const int zoff = zin ? z * plane_bytes : 0;
#pragma unroll
        for (int i = 0; i < PP_NLD; ++i) pf[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, zin ? goff[i] : OOB, zoff, 0);
    };
    // the three rotations of the weight fragments (constant for the launch)
    x3_u32x4 A[3][2];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int p = 0; p < 2; ++p) A[r][p] = wimg[1 + (r * 2 + p) * 64 + lane];
    fetch(z0 - 1);
    float bound = xmax_lane;
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) bound = fmaxf(bound, __shfl_xor(bound, m));
    float xinv;
    const float xs_scale = x3_pow2_scale(bound, xinv);
    const float unscale = xinv * whdr;
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PP_NLD; ++i) {
            x3_u32x2 h, l;
            x3_split4h(__builtin_bit_cast(x3_f32x4, pf[i]) * xs_scale, h, l);
            if (ls[i] >= 0) {
                *reinterpret_cast<x3_u32x2*>(smem + buf * PP_BUF + ls[i]) = h;
                *reinterpret_cast<x3_u32x2*>(smem + buf * PP_BUF + PP_PIECE + ls[i]) = l;
            }
        }
    };
    stash(0);
    fetch(z0);
    __syncthreads();
    // B fragments: lane (n, kq) reads the 8 channels of halo voxel (row 2 wave + rr + kh, column 14 j + n), kh = kq (kq = 3: the weights
    // are zero there; it re-reads kh = 0 so that the operand is a finite number)
    int baddr[2][PP_NJ], ooff[2][PP_NJ];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int j = 0; j < PP_NJ; ++j) {
            const int row = 2 * wave + rr, q = min(PP_STEP * j + n, PP_HW - 1);
            baddr[rr][j] = ((row + (kq < 3 ? kq : 0)) * PP_HW + q) * 16;
            const int wl = PP_STEP * j + n, oh = h0 + row, ow = w0 + wl;        // output pixel of this lane (lanes n < 14 of the group that finished)
            const bool ok = n < PP_STEP && wl < PP_TW && oh < H && ow < W;
            ooff[rr][j] = FUSE_D > 0 ? (ok ? (row * PP_TW + wl) * 4 : -1) : (ok ? (oh * W + ow) * 4 : OOB);
        }
    x3_f32x4 acc[2][PP_NJ];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int j = 0; j < PP_NJ; ++j) acc[rr][j] = (x3_f32x4){0.f, 0.f, 0.f, 0.f};
    float* const lg = reinterpret_cast<float*>(smem + 2 * PP_BUF);              // [FUSE_D][256] logits of the tile (+ one dummy word)

    // plane i of the block (input plane z0 - 1 + i) with rotation R = i mod 3; finishes output plane z0 + i - 2 in lane group (2 - R) mod 3
    auto body = [&](auto rtag, int i, int buf) {
        constexpr int R = decltype(rtag)::value, G = (2 - R + 3) % 3;
        const x3_byte* pb = smem + buf * PP_BUF;
#pragma unroll
        for (int j = 0; j < PP_NJ; ++j) {                           // (the two rows of a column tile alternate: no MFMA waits for the one before it)
            x3_u32x4 bh[2], bl[2];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                bh[rr] = *reinterpret_cast<const x3_u32x4*>(pb + baddr[rr][j]);
                bl[rr] = *reinterpret_cast<const x3_u32x4*>(pb + PP_PIECE + baddr[rr][j]);
            }
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) acc[rr][j] = x3_mfma<2>(A[R][0], bh[rr], acc[rr][j]);
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) acc[rr][j] = x3_mfma<2>(A[R][0], bl[rr], acc[rr][j]);
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) acc[rr][j] = x3_mfma<2>(A[R][1], bh[rr], acc[rr][j]);
        }
        const int o = i - 2;                                        // finished output plane (local)
        const bool mine = kq == G;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int j = 0; j < PP_NJ; ++j) {
                x3_f32x4& a = acc[rr][j];
                const float a0 = a[0], a1 = a[1], a2 = a[2];        // (copies: __builtin_bit_cast of a vector ELEMENT reads element 0)
                const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a1), 0x101, 0xf, 0xf, true));   // row_shl:1
                const float s2 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a2), 0x102, 0xf, 0xf, true));   // row_shl:2
                const float v = ((a0 + s1) + s2) * unscale;
                if (o >= 0 && o < z1 - z0) {
                    if constexpr (FUSE_D > 0) {
                        if (mine && ooff[rr][j] >= 0) lg[o * 256 + ooff[rr][j] / 4] = v;
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yrs, mine ? ooff[rr][j] : OOB, (z0 + o) * H * W * 4, 0);
                    }
                }
                a[0] = mine ? 0.0f : a[0]; a[1] = mine ? 0.0f : a[1]; a[2] = mine ? 0.0f : a[2];
            }
        if (i + 1 < nplanes) {
            stash(buf ^ 1);                                         // plane i + 1 (fetched during the previous plane)
            if (i + 2 < nplanes) fetch(z0 - 1 + i + 2);
        }
        __syncthreads();
    };
    using R0 = std::integral_constant<int, 0>;
    using R1 = std::integral_constant<int, 1>;
    using R2 = std::integral_constant<int, 2>;
    for (int i = 0; i < nplanes; i += 3) {
        body(R0{}, i, i & 1);
        if (i + 1 < nplanes) body(R1{}, i + 1, (i + 1) & 1);
        if (i + 2 < nplanes) body(R2{}, i + 2, i & 1);
    }
    if constexpr (FUSE_D > 0) {
        // softmax over the planes, soft-argmin depth, confidence window (models/casmvsnet.py:293-309; the arithmetic of depth_head.hip)
        const int lw = tid % PP_TW, lh = tid / PP_TW;
        const int oh = h0 + lh, ow = w0 + lw;
        const bool live = oh < H && ow < W;
        const int ooffp = live ? (oh * W + ow) * 4 : OOB;
        float v[FUSE_D];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < FUSE_D; ++k) { v[k] = lg[k * 256 + tid]; mx = fmaxf(mx, v[k]); }
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < FUSE_D; ++k) { v[k] = expf(v[k] - mx); sum += v[k]; }
        const long long hw = (long long)H * W;
        const long long pix = live ? (long long)oh * W + ow : 0;
        const float2 pl = reinterpret_cast<const float2*>(planes)[(long long)b * hw + pix];
        float dsum = 0.0f, isum = 0.0f;
#pragma unroll
        for (int k = 0; k < FUSE_D; ++k) {
            v[k] = v[k] / sum;
            if (y) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[k]), yrs, ooffp, k * H * W * 4, 0);
            dsum = fmaf(v[k], fmaf((float)k, pl.y, pl.x), dsum);
            isum = fmaf(v[k], (float)k, isum);
        }
        int idx = (int)isum;                       // .long(): truncation
        idx = idx < 0 ? 0 : (idx > FUSE_D - 1 ? FUSE_D - 1 : idx);
        float c = 0.0f;                            // p[i-1] + p[i] + p[i+1] + p[i+2], zero padded
#pragma unroll
        for (int k = 0; k < FUSE_D; ++k)
            if (k >= idx - 1 && k <= idx + 2) c += v[k];
        if (live) {
            depth[(long long)b * hw + pix] = dsum;
            conf[(long long)b * hw + pix] = c;
        }
    }
}

// z chunk: as for the VALU form (conv3d_lds.hip): slots = CUs x 5 resident blocks, minimise rounds x (c + 2)
static int prob_pair_zchunk(int tiles, int D) {
    static int slots = 0;
    if (!slots) {
        hipDeviceProp_t prop;
        int dev = 0;
        slots = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256) * 5;
    }
    long long best = -1; int zc = D < 8 ? D : 8;
    for (int c = D < 16 ? D : 16; c >= 2; --c) {
        const long long blocks = (long long)tiles * ((D + c - 1) / c);
        const long long cost = ((blocks + slots - 1) / slots) * (c + 2);
        if (best < 0 || cost < best) { best = cost; zc = c; }
    }
    return zc;
}

bool prob_pair_supported(int D, int H, int W) { return D > 0 && (long long)D * H * W * 32 < 0x7ffffff0LL; }

// logits (or, with D = 8, the whole head: depth, conf and optionally the probabilities) of x (B, D, H, W, 8); zc_force: test hook
int prob_pair_launch(const float* x, const float* wimg, const float* xmax, float* y, const float* planes, float* depth, float* conf,
                     int B, int D, int H, int W, bool fuse, int zc_force, hipStream_t st) {
    if (!prob_pair_supported(D, H, W)) return fail(-1, "prob_pair: volume too large for 32-bit offsets");
    const int tw_ = (W + PP_TW - 1) / PP_TW, th_ = (H + PP_TH - 1) / PP_TH;
    if (fuse) {
        if (D != 8) return fail(-1, "prob_pair: the one-launch head needs D = 8 (got %d)", D);
        hipLaunchKernelGGL(prob_pair_kernel<8>, dim3(tw_ * th_, 1, B), dim3(256), 0, st, x, reinterpret_cast<const x3_u32x4*>(wimg), xmax, y, D, H, W, tw_, th_, D,
                           planes, depth, conf);
        return launch_status("depth_head(pair, fused)");
    }
    const int zc = zc_force ? (zc_force < D ? zc_force : D) : prob_pair_zchunk(B * tw_ * th_, D);
    hipLaunchKernelGGL(prob_pair_kernel<0>, dim3(tw_ * th_, (D + zc - 1) / zc, B), dim3(256), 0, st, x, reinterpret_cast<const x3_u32x4*>(wimg), xmax, y, D, H, W, tw_, th_, zc,
                       (const float*)nullptr, (float*)nullptr, (float*)nullptr);
    return launch_status("prob_pair");
}

}  // namespace rcmvs
