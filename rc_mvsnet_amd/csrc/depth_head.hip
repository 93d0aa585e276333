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

}  // namespace rcmvs

using namespace rcmvs;

// variant 0 = production (fused single launch for D <= 64), 1 = the two-launch path (needs `prob` as the logit scratch)
static int depth_head_dispatch(const float* x, const float* w_prob, const float* planes, float* depth, float* conf, float* prob,
                               int B, int D, int h, int w, int variant, hipStream_t st) {
    RCMVS_REQUIRE(x && w_prob && planes && depth && conf, "depth_head_fwd: null pointer");
    RCMVS_REQUIRE(B > 0 && D > 0 && h > 0 && w > 0, "depth_head_fwd: bad sizes");
    RCMVS_REQUIRE(D <= 64, "depth_head_fwd: at most 64 depth hypotheses per stage (got %d)", D);
    RCMVS_REQUIRE(variant == 0 || variant == 1, "depth_head_fwd: unknown variant %d", variant);
    RCMVS_REQUIRE((long long)h * w * 8 < (1LL << 31), "depth_head_fwd: plane too large for 32-bit offsets");
    if (variant == 0) {
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
