// K4: prob conv (8 -> 1, 3x3x3) + softmax over planes + soft-argmin depth + photometric
// confidence.  Replaces CostRegNet.prob (models/modules.py:489,500), F.softmax, depth_regression (x2),
// F.pad + avg_pool3d and torch.gather in DepthNet_eval.forward (models/casmvsnet.py:293-309).
//
// Production path (round 3): ONE launch, depth_head_fused_kernel.  A block owns a pixel tile and ALL D planes of it: the
// 8-channel volume is read once (no z halo between chunks), the logits of the tile stay in LDS, and the softmax / soft-argmin /
// index / 4-tap confidence window run on them in the same block -- the logit volume is neither written nor re-read, the
// probability volume is written only when the caller asks for it (training: rcmvs_depth_head_bwd needs it).
// The older two-launch path (plane-marching prob conv of conv3d_lds.hip writing logits into `prob`, then
// softmax_regress_kernel in place) stays behind rcmvs_debug_depth_head_fwd(variant = 1) as the cross-check, and serves D > 64.
#include "common.h"

namespace rcmvs {

int conv3d_lds_launch(const float* x, const float* wp, const float* scale, const float* shift, const float* res, float* y,
                      int B, int D, int H, int W, int Ci, int Co, int relu, hipStream_t st, int lds_cfg);   // conv3d_lds.hip

// LP adjacent lanes share one pixel and own the planes k = j, j + LP, ... (at most 16 each, held in registers): the logit
// column is read ONCE with all of a lane's loads in flight, max / sum / soft-argmin / window sums are combined across the
// LP lanes with shuffles, the probabilities are written once.  LP = 1, 2, 4 for D <= 16, 32, 64 -- the 128x160 stage has
// only 20 k pixels, so spreading a pixel over four lanes is also what fills the machine there.
template <int LP>
__global__ __launch_bounds__(256) void softmax_regress_kernel(float* __restrict__ prob, const float* __restrict__ planes,
                                                               float* __restrict__ depth, float* __restrict__ conf, int D, long long hw) {
    constexpr int MAXK = 16;
    const int b = blockIdx.y;
    const int j = threadIdx.x % LP;
    const long long p = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / LP;
    const bool live = p < hw;
    float* col = prob + (long long)b * D * hw + (live ? p : 0);  // element k at col[k*hw]
    float v[MAXK];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAXK; ++i) {
        const int k = j + i * LP;
        v[i] = (k < D) ? col[(long long)k * hw] : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
#pragma unroll
    for (int m = 1; m < LP; m <<= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXK; ++i) {
        v[i] = (j + i * LP < D) ? expf(v[i] - mx) : 0.0f;
        sum += v[i];
    }
#pragma unroll
    for (int m = 1; m < LP; m <<= 1) sum += __shfl_xor(sum, m);
    const float2 pl = reinterpret_cast<const float2*>(planes)[(long long)b * hw + (live ? p : 0)];
    float dsum = 0.0f, isum = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXK; ++i) {
        const int k = j + i * LP;
        v[i] = v[i] / sum;
        if (k < D) {
            if (live) col[(long long)k * hw] = v[i];
            dsum += v[i] * (pl.x + (float)k * pl.y);
            isum += v[i] * (float)k;
        }
    }
#pragma unroll
    for (int m = 1; m < LP; m <<= 1) { dsum += __shfl_xor(dsum, m); isum += __shfl_xor(isum, m); }
    int idx = (int)isum;                       // .long(): truncation
    idx = idx < 0 ? 0 : (idx > D - 1 ? D - 1 : idx);
    float c = 0.0f;                            // p[i-1] + p[i] + p[i+1] + p[i+2], zero padded
#pragma unroll
    for (int i = 0; i < MAXK; ++i) {
        const int k = j + i * LP;
        if (k < D && k >= idx - 1 && k <= idx + 2) c += v[i];
    }
#pragma unroll
    for (int m = 1; m < LP; m <<= 1) c += __shfl_xor(c, m);
    if (live && j == 0) {
        depth[(long long)b * hw + p] = dsum;
        conf[(long long)b * hw + p] = c;
    }
}


// ---- fused head ----------------------------------------------------------------------------------------------------------
// Block = 256 threads = ZS z-chunk groups x PX pixels (PX = 256 / ZS; tile = PX / 32 rows x 32 columns).  Group g marches over
// the planes [g * ZC - 1, g * ZC + ZC] of its chunk with the rolling-accumulator scheme of prob_conv_march_kernel (plane z
// feeds the kd = 0 / 1 / 2 terms of out[z + 1] / out[z] / out[z - 1]; planes double-buffered in the group's own LDS slabs,
// one block barrier per plane) and leaves logit[z][pixel] in LDS.  ZS = 4 / 2 / 1 for D <= 64 / 32 / 16: the 128 x 160
// stage has only 20 k pixels, the z split is what fills the machine there (320 blocks of four chunk waves instead of 80).
// Softmax phase: thread (j, pixel) owns the planes k = j, j + ZS, ... exactly like softmax_regress_kernel<LP = ZS> and the
// partial max / sums are combined in the same butterfly order (through LDS: the ZS lanes of a pixel sit in different waves),
// so depth / confidence / probabilities are bit-identical to the two-launch path.
constexpr int DH_TW = 32, DH_HW = DH_TW + 2, DH_STRIDE = 12;      // 8 channels in a 12-float row stride: conflict-free ds_read_b128
template <int ZS>
struct DhCfg {
    static constexpr int PX = 256 / ZS, TH = PX / DH_TW, HH = TH + 2;
    static constexpr int PLANE = HH * DH_HW * DH_STRIDE;              // floats per staged plane
    static constexpr int NLD = (HH * DH_HW * 2 + PX - 1) / PX;        // float4 per thread per plane
    static constexpr int MAXK = 16;                                    // planes per softmax thread (D <= 16 ZS)
};

template <int ZS>
__global__ __launch_bounds__(256) void depth_head_fused_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ planes,
    float* __restrict__ depth, float* __restrict__ conf, float* __restrict__ prob, int D, int H, int W, int tiles_w) {
    using C = DhCfg<ZS>;
    typedef float f2v __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) float dh_smem[];
    float* const slabs = dh_smem;                                      // [ZS][2][PLANE]
    float* const logit = dh_smem + ZS * 2 * C::PLANE;                  // [D][PX]
    float* const red = logit + D * C::PX;                              // [4][256] reduction scratch
    const int b = blockIdx.y;
    const unsigned t2 = xcd_remap(blockIdx.x, gridDim.x);
    const int tw = t2 % tiles_w, th = t2 / tiles_w;
    const int h0 = th * C::TH, w0 = tw * DH_TW;
    const int grp = threadIdx.x / C::PX, pt = threadIdx.x % C::PX;     // z-chunk group, thread inside the group
    const int ZC = (D + ZS - 1) / ZS;
    const int z0 = grp * ZC, z1 = min(D, z0 + ZC);                     // this group's outputs z0 .. z1-1 (empty when z0 >= D)
    const int lw = pt % DH_TW, lh = pt / DH_TW;
    const float* xb = x + (long long)b * D * H * W * 8;
    float* const plane0 = slabs + grp * 2 * C::PLANE;
    int goff[C::NLD], loff[C::NLD];
#pragma unroll
    for (int i = 0; i < C::NLD; ++i) {
        const int e = pt + i * C::PX;
        const int v = e >> 1, c4 = e & 1;
        const int hh = v / DH_HW, hw_ = v - hh * DH_HW;
        const int ih = h0 + hh - 1, iw = w0 + hw_ - 1;
        const bool ok = e < C::HH * DH_HW * 2 && ih >= 0 && ih < H && iw >= 0 && iw < W;
        goff[i] = ok ? (ih * W + iw) * 8 + c4 * 4 : -1;
        loff[i] = (e < C::HH * DH_HW * 2) ? v * DH_STRIDE + c4 * 4 : -1;
    }
    float4 pf[C::NLD];
    auto fetch = [&](int z) {
        const bool zin = z >= 0 && z < D;
        const float* xp = xb + (long long)z * H * W * 8;
#pragma unroll
        for (int i = 0; i < C::NLD; ++i)
            pf[i] = (zin && goff[i] >= 0) ? *reinterpret_cast<const float4*>(xp + goff[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < C::NLD; ++i)
            if (loff[i] >= 0) *reinterpret_cast<float4*>(&plane0[buf * C::PLANE + loff[i]]) = pf[i];
    };
    fetch(z0 - 1);
    stash(0);
    fetch(z0);
    __syncthreads();
    f2v acc_prev = (f2v){0.f, 0.f}, acc_cur = (f2v){0.f, 0.f};
    int buf = 0;
    // every group runs the same ZC + 2 iterations (block barriers inside); a ragged last chunk just computes planes nobody keeps
    for (int it = 0; it < ZC + 2; ++it) {
        const int z = z0 - 1 + it;
        f2v acc_next = (f2v){0.f, 0.f};
        const float* tp0 = &plane0[buf * C::PLANE + (lh * DH_HW + lw) * DH_STRIDE];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const float* tp = tp0 + (kh * DH_HW + kw) * DH_STRIDE;
                const float4 xa = *reinterpret_cast<const float4*>(tp), xc = *reinterpret_cast<const float4*>(tp + 4);
                const f2v x01 = (f2v){xa.x, xa.y}, x23 = (f2v){xa.z, xa.w}, x45 = (f2v){xc.x, xc.y}, x67 = (f2v){xc.z, xc.w};
#pragma unroll
                for (int kd = 0; kd < 3; ++kd) {
                    const float* wt = wp + ((kd * 3 + kh) * 3 + kw) * 8;
                    f2v a = (kd == 0) ? acc_next : (kd == 1 ? acc_cur : acc_prev);
                    a = __builtin_elementwise_fma(x01, (f2v){wt[0], wt[1]}, a);
                    a = __builtin_elementwise_fma(x23, (f2v){wt[2], wt[3]}, a);
                    a = __builtin_elementwise_fma(x45, (f2v){wt[4], wt[5]}, a);
                    a = __builtin_elementwise_fma(x67, (f2v){wt[6], wt[7]}, a);
                    if (kd == 0) acc_next = a; else if (kd == 1) acc_cur = a; else acc_prev = a;
                }
            }
        const int zo = z - 1;                                          // complete now
        if (zo >= z0 && zo < z1) logit[zo * C::PX + pt] = acc_prev.x + acc_prev.y;
        acc_prev = acc_cur; acc_cur = acc_next;
        if (it + 1 < ZC + 2) {
            stash(buf ^ 1);                                            // plane z + 1 (fetched during the previous iteration)
            if (it + 2 < ZC + 2) fetch(z + 2);
        }
        __syncthreads();
        buf ^= 1;
    }
    // ---- softmax / regression / confidence on the tile's logits (same arithmetic and order as softmax_regress_kernel<ZS>)
    const int j = grp;                                                 // lane index inside the pixel's group of ZS
    const int oh = h0 + lh, ow = w0 + lw;
    const bool live = oh < H && ow < W;
    const long long hw = (long long)H * W, p = (long long)oh * W + ow;
    float v[C::MAXK];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < C::MAXK; ++i) {
        const int k = j + i * ZS;
        v[i] = (k < D) ? logit[k * C::PX + pt] : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
    // butterfly over the ZS lanes of a pixel (xor 1, then xor 2), through LDS
    auto xor_read = [&](float* arr, float mine, int m) { arr[threadIdx.x] = mine; __syncthreads(); const float o = arr[((j ^ m) * C::PX) + pt]; __syncthreads(); return o; };
#pragma unroll
    for (int m = 1; m < ZS; m <<= 1) mx = fmaxf(mx, xor_read(red, mx, m));
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < C::MAXK; ++i) {
        v[i] = (j + i * ZS < D) ? expf(v[i] - mx) : 0.0f;
        sum += v[i];
    }
#pragma unroll
    for (int m = 1; m < ZS; m <<= 1) sum += xor_read(red, sum, m);
    const float2 pl = live ? reinterpret_cast<const float2*>(planes)[(long long)b * hw + p] : make_float2(0.f, 0.f);
    float dsum = 0.0f, isum = 0.0f;
#pragma unroll
    for (int i = 0; i < C::MAXK; ++i) {
        const int k = j + i * ZS;
        v[i] = v[i] / sum;
        if (k < D) {
            if (live && prob) prob[((long long)b * D + k) * hw + p] = v[i];
            dsum += v[i] * (pl.x + (float)k * pl.y);
            isum += v[i] * (float)k;
        }
    }
#pragma unroll
    for (int m = 1; m < ZS; m <<= 1) { dsum += xor_read(red, dsum, m); isum += xor_read(red + 256, isum, m); }
    int idx = (int)isum;                       // .long(): truncation
    idx = idx < 0 ? 0 : (idx > D - 1 ? D - 1 : idx);
    float c = 0.0f;                            // p[i-1] + p[i] + p[i+1] + p[i+2], zero padded
#pragma unroll
    for (int i = 0; i < C::MAXK; ++i) {
        const int k = j + i * ZS;
        if (k < D && k >= idx - 1 && k <= idx + 2) c += v[i];
    }
#pragma unroll
    for (int m = 1; m < ZS; m <<= 1) c += xor_read(red, c, m);
    if (live && j == 0) {
        depth[(long long)b * hw + p] = dsum;
        conf[(long long)b * hw + p] = c;
    }
}

template <int ZS>
static int depth_head_fused_launch(const float* x, const float* wp, const float* planes, float* depth, float* conf, float* prob,
                                   int B, int D, int h, int w, hipStream_t st) {
    using C = DhCfg<ZS>;
    const int tiles_w = (w + DH_TW - 1) / DH_TW, tiles_h = (h + C::TH - 1) / C::TH;
    const size_t lds = (size_t)(ZS * 2 * C::PLANE + D * C::PX + 2 * 256) * sizeof(float);
    if (lds > 64 * 1024) {      // raise the dynamic-LDS limit once per device (races are benign: the same value is written)
        static bool raised[64];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fail(-1, "depth_head_fwd: cannot query the device");
        if (!raised[dev]) {
            (void)hipFuncSetAttribute((const void*)depth_head_fused_kernel<ZS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            raised[dev] = true;
        }
    }
    hipLaunchKernelGGL(depth_head_fused_kernel<ZS>, dim3(tiles_w * tiles_h, B), dim3(256), lds, st, x, wp, planes, depth, conf, prob, D, h, w, tiles_w);
    return launch_status("depth_head_fwd(fused)");
}


// ---- strip-mined head (round 3, variant 3) ---------------------------------------------------------------------------------
// Both forms above feed every v_pk_fma_f32 of the prob conv from two v_readlane_b32 (the 216 scalar weights are hoisted out of the
// plane loop and spilled: 108 packed FMAs + 216 lane reads per pixel and plane).  Here a thread owns a STRIP of four vertically
// adjacent pixels, so a weight pair, once in SGPRs, feeds four FMAs, and the 3 x 3 neighbourhoods of the four pixels overlap
// (18 voxel reads for four outputs instead of 36); the tap-column loop is kept rolled with an opaque weight pointer, so the 72
// weights of one column stay in SGPRs for their 144 FMAs and nothing is spilled.
// A WAVE is the unit: tile of 8 rows x 32 columns (lanes 0-31: rows 0-3, lanes 32-63: rows 4-7; lane & 31 = column), one z chunk,
// its own LDS slab (10 x 34 halo voxels, 12-float stride), no block barrier in the march -- the waves of a block drift apart and
// cover each other's scalar-load and staging latency.  Block = TPB tiles x ZS z chunks: D <= 16: 4 tiles x 1 chunk, the logits of
// a thread's four pixels stay in registers; D <= 32: 1 tile x 4 chunks, D <= 64: 1 tile x 6 chunks, logits of the tile through LDS
// and one thread per pixel for the softmax.  Softmax arithmetic: sequential over the planes (softmax_regress_kernel<1> order).
constexpr int DS_TH = 8, DS_TW = 32, DS_HH = DS_TH + 2, DS_HW = DS_TW + 2, DS_STRIDE = 12;
constexpr int DS_SLAB = DS_HH * DS_HW * DS_STRIDE;                    // floats per wave slab (16,320 B)
constexpr int DS_NLD = (DS_HH * DS_HW * 2 + 63) / 64;                 // float4 per lane per plane

// softmax + soft-argmin + index + 4-tap confidence of one pixel's logit column v[0..D) (D <= MAXK); writes prob if asked
template <int MAXK>
__device__ __forceinline__ void dh_pixel_softmax(float (&v)[MAXK], int D, float2 pl, float* __restrict__ probcol, long long hw,
                                                 float& depth_out, float& conf_out) {
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < MAXK; ++k) if (k < D) mx = fmaxf(mx, v[k]);
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < MAXK; ++k) { v[k] = (k < D) ? expf(v[k] - mx) : 0.0f; sum += v[k]; }
    float dsum = 0.0f, isum = 0.0f;
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
        v[k] = v[k] / sum;
        if (k < D) {
            if (probcol) probcol[(long long)k * hw] = v[k];
            dsum += v[k] * (pl.x + (float)k * pl.y);
            isum += v[k] * (float)k;
        }
    }
    int idx = (int)isum;                       // .long(): truncation
    idx = idx < 0 ? 0 : (idx > D - 1 ? D - 1 : idx);
    float c = 0.0f;                            // p[i-1] + p[i] + p[i+1] + p[i+2], zero padded
#pragma unroll
    for (int k = 0; k < MAXK; ++k) if (k < D && k >= idx - 1 && k <= idx + 2) c += v[k];
    depth_out = dsum;
    conf_out = c;
}

template <int ZS, int TPB>
__global__ __launch_bounds__(64 * ZS * TPB) void depth_head_strip_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ planes,
    float* __restrict__ depth, float* __restrict__ conf, float* __restrict__ prob, int D, int H, int W, int tiles_w, int ntiles) {
    typedef float f2v __attribute__((ext_vector_type(2)));
    constexpr int MAXK = (ZS == 1) ? 16 : 1;                           // logits a thread keeps in registers per pixel (ZS = 1 only)
    extern __shared__ __attribute__((aligned(16))) float ds_smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = blockIdx.y;
    const int tsub = wave / ZS, grp = wave % ZS;                       // tile inside the block, z chunk
    const unsigned t2 = xcd_remap(blockIdx.x, gridDim.x) * TPB + tsub;
    const bool tile_ok = (int)t2 < ntiles;
    const int tw = t2 % tiles_w, th = t2 / tiles_w;
    const int h0 = th * DS_TH, w0 = tw * DS_TW;
    const int ZC = (D + ZS - 1) / ZS;
    const int z0 = grp * ZC, z1 = min(D, z0 + ZC);
    float* const slab = ds_smem + wave * DS_SLAB;
    float* const logit = ds_smem + ZS * TPB * DS_SLAB;                 // [D][256] (ZS > 1 only; TPB = 1 then)
    const long long hw = (long long)H * W;
    const float* xb = x + (long long)b * D * hw * 8;
    const int c = lane & 31, r0 = 4 * (lane >> 5);
    int goff[DS_NLD], loff[DS_NLD];
#pragma unroll
    for (int i = 0; i < DS_NLD; ++i) {
        const int e = lane + i * 64;
        const int v = e >> 1, c4 = e & 1;
        const int hh = v / DS_HW, hw_ = v - hh * DS_HW;
        const int ih = h0 + hh - 1, iw = w0 + hw_ - 1;
        const bool ok = tile_ok && e < DS_HH * DS_HW * 2 && ih >= 0 && ih < H && iw >= 0 && iw < W;
        goff[i] = ok ? (ih * W + iw) * 8 + c4 * 4 : -1;
        loff[i] = (e < DS_HH * DS_HW * 2) ? v * DS_STRIDE + c4 * 4 : -1;
    }
    float4 pf[DS_NLD];
    auto fetch = [&](int z) {
        const bool zin = z >= 0 && z < D;
        const float* xp = xb + (long long)z * hw * 8;
#pragma unroll
        for (int i = 0; i < DS_NLD; ++i)
            pf[i] = (zin && goff[i] >= 0) ? *reinterpret_cast<const float4*>(xp + goff[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    float lg[4][MAXK];                                                 // ZS = 1: the strip's logits
    f2v acc_prev[4], acc_cur[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) { acc_prev[s] = (f2v){0.f, 0.f}; acc_cur[s] = (f2v){0.f, 0.f}; }
    fetch(z0 - 1);
    const int nit = (z0 < z1) ? (z1 - z0) + 2 : 0;                     // planes z0 - 1 .. z1 (an empty ragged chunk does nothing)
    for (int it = 0; it < nit; ++it) {
        const int z = z0 - 1 + it;
        __builtin_amdgcn_wave_barrier();                               // (every lane has finished reading the previous plane)
#pragma unroll
        for (int i = 0; i < DS_NLD; ++i)
            if (loff[i] >= 0) *reinterpret_cast<float4*>(&slab[loff[i]]) = pf[i];
        __builtin_amdgcn_wave_barrier();                               // wave-private slab: in-order LDS, no block barrier
        if (it + 1 < nit) fetch(z + 1);                                // in flight during this plane's FMAs
        f2v acc_next[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) acc_next[s] = (f2v){0.f, 0.f};
#pragma unroll 1
        for (int kw = 0; kw < 3; ++kw) {
            const float* wq = wp + kw * 8;
            asm volatile("" : "+r"(wq));                               // opaque: the 72 weights of this tap column are loaded HERE, not hoisted
            float4 xa[6], xc[6];
            const float* tp = slab + (r0 * DS_HW + c + kw) * DS_STRIDE;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                xa[j] = *reinterpret_cast<const float4*>(tp + j * DS_HW * DS_STRIDE);
                xc[j] = *reinterpret_cast<const float4*>(tp + j * DS_HW * DS_STRIDE + 4);
            }
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kd = 0; kd < 3; ++kd) {
                    const float* wt = wq + (kd * 3 + kh) * 24;
                    const f2v w01 = (f2v){wt[0], wt[1]}, w23 = (f2v){wt[2], wt[3]}, w45 = (f2v){wt[4], wt[5]}, w67 = (f2v){wt[6], wt[7]};
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const float4 A = xa[s + kh], Cc = xc[s + kh];
                        f2v a = (kd == 0) ? acc_next[s] : (kd == 1 ? acc_cur[s] : acc_prev[s]);
                        a = __builtin_elementwise_fma((f2v){A.x, A.y}, w01, a);
                        a = __builtin_elementwise_fma((f2v){A.z, A.w}, w23, a);
                        a = __builtin_elementwise_fma((f2v){Cc.x, Cc.y}, w45, a);
                        a = __builtin_elementwise_fma((f2v){Cc.z, Cc.w}, w67, a);
                        if (kd == 0) acc_next[s] = a; else if (kd == 1) acc_cur[s] = a; else acc_prev[s] = a;
                    }
                }
        }
        const int zo = z - 1;                                          // complete now
        if (zo >= z0 && zo < z1) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float v = acc_prev[s].x + acc_prev[s].y;
                if constexpr (ZS == 1) {
#pragma unroll
                    for (int k = 0; k < MAXK; ++k) if (k == zo) lg[s][k] = v;
                } else {
                    logit[zo * 256 + (r0 + s) * DS_TW + c] = v;
                }
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) { acc_prev[s] = acc_cur[s]; acc_cur[s] = acc_next[s]; }
    }
    if constexpr (ZS == 1) {
        if (!tile_ok) return;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int oh = h0 + r0 + s, ow = w0 + c;
            if (oh >= H || ow >= W) continue;
            const long long p = (long long)oh * W + ow;
            const float2 pl = reinterpret_cast<const float2*>(planes)[(long long)b * hw + p];
            float d_, c_;
            dh_pixel_softmax<MAXK>(lg[s], D, pl, prob ? prob + (long long)b * D * hw + p : nullptr, hw, d_, c_);
            depth[(long long)b * hw + p] = d_;
            conf[(long long)b * hw + p] = c_;
        }
    } else {
        __syncthreads();                                               // the tile's logits are complete
        if (threadIdx.x < 256 && tile_ok) {
            const int px = threadIdx.x, oh = h0 + px / DS_TW, ow = w0 + px % DS_TW;
            if (oh < H && ow < W) {
                const long long p = (long long)oh * W + ow;
                const float2 pl = reinterpret_cast<const float2*>(planes)[(long long)b * hw + p];
                float v[64];
#pragma unroll
                for (int k = 0; k < 64; ++k) v[k] = (k < D) ? logit[k * 256 + px] : 0.0f;
                float d_, c_;
                dh_pixel_softmax<64>(v, D, pl, prob ? prob + (long long)b * D * hw + p : nullptr, hw, d_, c_);
                depth[(long long)b * hw + p] = d_;
                conf[(long long)b * hw + p] = c_;
            }
        }
    }
}

template <int ZS, int TPB>
static int depth_head_strip_launch(const float* x, const float* wp, const float* planes, float* depth, float* conf, float* prob,
                                   int B, int D, int h, int w, hipStream_t st) {
    const int tiles_w = (w + DS_TW - 1) / DS_TW, tiles_h = (h + DS_TH - 1) / DS_TH;
    const int ntiles = tiles_w * tiles_h;
    const size_t lds = (size_t)(ZS * TPB * DS_SLAB + (ZS > 1 ? D * 256 : 0)) * sizeof(float);
    if (lds > 64 * 1024) {
        static bool raised[64];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fail(-1, "depth_head_fwd: cannot query the device");
        if (!raised[dev]) {
            (void)hipFuncSetAttribute((const void*)depth_head_strip_kernel<ZS, TPB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            raised[dev] = true;
        }
    }
    hipLaunchKernelGGL((depth_head_strip_kernel<ZS, TPB>), dim3((ntiles + TPB - 1) / TPB, B), dim3(64 * ZS * TPB), lds, st,
                       x, wp, planes, depth, conf, prob, D, h, w, tiles_w, ntiles);
    return launch_status("depth_head_fwd(strip)");
}

}  // namespace rcmvs

using namespace rcmvs;

// variant 0 = production = the two-launch path while the fused kernel is slower on the hardware (137.6 vs 134.0 us per scene,
// profiles/r3_depth_head_ab.txt: the prob conv is bound by the v_readlane traffic of its spilled scalar weights, not by the logit
// round trip); 1 = two-launch explicitly; 2 = fused single launch (D <= 64).  The two-launch forms need `prob` (logit scratch).
static int depth_head_dispatch(const float* x, const float* w_prob, const float* planes, float* depth, float* conf, float* prob,
                               int B, int D, int h, int w, int variant, hipStream_t st) {
    RCMVS_REQUIRE(x && w_prob && planes && depth && conf, "depth_head_fwd: null pointer");
    RCMVS_REQUIRE(B > 0 && D > 0 && h > 0 && w > 0, "depth_head_fwd: bad sizes");
    RCMVS_REQUIRE(D <= 64, "depth_head_fwd: at most 64 depth hypotheses per stage (got %d)", D);
    RCMVS_REQUIRE(variant >= 0 && variant <= 3, "depth_head_fwd: unknown variant %d", variant);
    if (variant == 3) {
        if (D <= 16) return depth_head_strip_launch<1, 4>(x, w_prob, planes, depth, conf, prob, B, D, h, w, st);
        if (D <= 32) return depth_head_strip_launch<4, 1>(x, w_prob, planes, depth, conf, prob, B, D, h, w, st);
        return depth_head_strip_launch<6, 1>(x, w_prob, planes, depth, conf, prob, B, D, h, w, st);
    }
    RCMVS_REQUIRE((long long)h * w * 8 < (1LL << 31), "depth_head_fwd: plane too large for 32-bit offsets");
    if (variant == 2 || (variant == 0 && !prob)) {
        if (D <= 16) return depth_head_fused_launch<1>(x, w_prob, planes, depth, conf, prob, B, D, h, w, st);
        if (D <= 32) return depth_head_fused_launch<2>(x, w_prob, planes, depth, conf, prob, B, D, h, w, st);
        return depth_head_fused_launch<4>(x, w_prob, planes, depth, conf, prob, B, D, h, w, st);
    }
    RCMVS_REQUIRE(prob, "depth_head_fwd (two-launch variant): prob is required, it doubles as the logit scratch");
    int rc = conv3d_lds_launch(x, w_prob, nullptr, nullptr, nullptr, prob, B, D, h, w, 8, 1, 0, st, 0);
    if (rc) return rc;
    const long long hw = (long long)h * w;
    if (D <= 16)      hipLaunchKernelGGL(softmax_regress_kernel<1>, dim3((unsigned)cdiv(hw, 256), B), dim3(256), 0, st, prob, planes, depth, conf, D, hw);
    else if (D <= 32) hipLaunchKernelGGL(softmax_regress_kernel<2>, dim3((unsigned)cdiv(hw * 2, 256), B), dim3(256), 0, st, prob, planes, depth, conf, D, hw);
    else              hipLaunchKernelGGL(softmax_regress_kernel<4>, dim3((unsigned)cdiv(hw * 4, 256), B), dim3(256), 0, st, prob, planes, depth, conf, D, hw);
    return launch_status("depth_head_fwd");
}

extern "C" int rcmvs_depth_head_fwd(const float* x, const float* w_prob, const float* planes,
                                    float* depth, float* conf, float* prob,
                                    int B, int D, int h, int w, void* stream) {
    return depth_head_dispatch(x, w_prob, planes, depth, conf, prob, B, D, h, w, 0, as_stream(stream));
}

extern "C" int rcmvs_debug_depth_head_fwd(const float* x, const float* w_prob, const float* planes,
                                          float* depth, float* conf, float* prob,
                                          int B, int D, int h, int w, int variant, void* stream) {
    return depth_head_dispatch(x, w_prob, planes, depth, conf, prob, B, D, h, w, variant, as_stream(stream));
}
