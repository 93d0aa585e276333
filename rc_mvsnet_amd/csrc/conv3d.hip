// K2/K3: 3-D convolution family (3x3x3, pad 1; stride 1 / stride 2 / transposed stride 2),
// channels-last, with the BatchNorm(eval) scale/shift, ReLU and skip-add fused in the epilogue.
// Replaces Conv3d.forward / Deconv3d.forward (models/modules.py:149-157,196-204) and the
// additions of CostRegNet.forward (:497-499).
//
// Two code paths:
//   * conv3d_direct_kernel  -- generic: one thread per output voxel, all Co accumulators in
//     registers, 16-byte channel-vector loads of the input, weights through wave-uniform
//     (scalar-cache) loads.  Handles every mode, any size; used for the strided / transposed
//     layers and as the fallback.
//   * conv3d_mfma_kernel (conv3d_mfma.hip) -- stride-1 layers on v_mfma_f32_16x16x4_f32.
#include "common.h"

namespace rcmvs {

enum ConvMode { CONV_S1 = 0, CONV_S2 = 1, CONV_T2 = 2 };

struct ConvDims {
    int B, D, H, W;        // input
    int Do, Ho, Wo;        // output
};

template <int CI, int CO, int MODE>
__global__ __launch_bounds__(256) void conv3d_direct_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ res, float* __restrict__ y,
    ConvDims dm, int relu) {
    constexpr int VW = (CI % 4 == 0) ? 4 : 1;      // input channel vector width
    const long long nvox = (long long)dm.B * dm.Do * dm.Ho * dm.Wo;
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nvox) return;
    int ow = (int)(t % dm.Wo);
    long long r = t / dm.Wo;
    int oh = (int)(r % dm.Ho); r /= dm.Ho;
    int od = (int)(r % dm.Do);
    int b = (int)(r / dm.Do);

    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = 0.0f;

    const float* xb = x + (long long)b * dm.D * dm.H * dm.W * CI;
    for (int kd = 0; kd < 3; ++kd) {
        int id;
        bool vd;
        if (MODE == CONV_T2) { int n = od + 1 - kd; id = n >> 1; vd = (n >= 0) && !(n & 1) && id < dm.D; }
        else { id = (MODE == CONV_S2 ? 2 * od : od) + kd - 1; vd = id >= 0 && id < dm.D; }
        if (!vd) continue;
        for (int kh = 0; kh < 3; ++kh) {
            int ih;
            bool vh;
            if (MODE == CONV_T2) { int n = oh + 1 - kh; ih = n >> 1; vh = (n >= 0) && !(n & 1) && ih < dm.H; }
            else { ih = (MODE == CONV_S2 ? 2 * oh : oh) + kh - 1; vh = ih >= 0 && ih < dm.H; }
            if (!vh) continue;
            for (int kw = 0; kw < 3; ++kw) {
                int iw;
                bool vw;
                if (MODE == CONV_T2) { int n = ow + 1 - kw; iw = n >> 1; vw = (n >= 0) && !(n & 1) && iw < dm.W; }
                else { iw = (MODE == CONV_S2 ? 2 * ow : ow) + kw - 1; vw = iw >= 0 && iw < dm.W; }
                if (!vw) continue;
                const float* xp = xb + (((long long)id * dm.H + ih) * dm.W + iw) * CI;
                const float* wt = wp + (long long)((kd * 3 + kh) * 3 + kw) * CI * CO;
#pragma unroll 2
                for (int c0 = 0; c0 < CI; c0 += VW) {
                    float xv[VW];
                    if (VW == 4) {
                        float4 v4 = *reinterpret_cast<const float4*>(xp + c0);
                        xv[0] = v4.x; xv[1 % VW] = v4.y; xv[2 % VW] = v4.z; xv[3 % VW] = v4.w;
                    } else {
                        xv[0] = xp[c0];
                    }
#pragma unroll
                    for (int j = 0; j < VW; ++j)
#pragma unroll
                        for (int co = 0; co < CO; ++co)
                            acc[co] = fmaf(xv[j], wt[(c0 + j) * CO + co], acc[co]);
                }
            }
        }
    }
    float* yp = y + t * CO;
    const float* rp = res ? res + t * CO : nullptr;
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        float v = acc[co];
        if (scale) v = v * scale[co] + shift[co];
        if (relu) v = fmaxf(v, 0.0f);
        if (rp) v += rp[co];
        acc[co] = v;
    }
    if (CO % 4 == 0) {
#pragma unroll
        for (int co = 0; co < CO; co += 4)
            *reinterpret_cast<float4*>(yp + co) = make_float4(acc[co], acc[co + 1], acc[co + 2], acc[co + 3]);
    } else {
#pragma unroll
        for (int co = 0; co < CO; ++co) yp[co] = acc[co];
    }
}

// weight repack: conv (Co,Ci,27) / deconv (Ci,Co,27) -> [27][Ci][Co]; transposed == 2 additionally flips the taps
// (the adjoint of a stride-1 conv: data gradient on the forward kernels)
__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ packed, int Co, int Ci, int transposed) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int n = 27 * Ci * Co;
    if (t >= n) return;
    int co = t % Co, ci = (t / Co) % Ci, tap = t / (Co * Ci);
    long long src = transposed ? ((long long)ci * Co + co) * 27 + (transposed == 2 ? 26 - tap : tap) : ((long long)co * Ci + ci) * 27 + tap;
    packed[t] = w[src];
}

template <int MODE>
static int direct_dispatch(const float* x, const float* wp, const float* scale, const float* shift, const float* res,
                           float* y, const ConvDims& dm, int Ci, int Co, int relu, hipStream_t st) {
    const long long nvox = (long long)dm.B * dm.Do * dm.Ho * dm.Wo;
    dim3 grid((unsigned)cdiv(nvox, 256)), block(256);
#define RCMVS_CONV_CASE(CI, CO)                                                                                   \
    if (Ci == CI && Co == CO) {                                                                                   \
        hipLaunchKernelGGL((conv3d_direct_kernel<CI, CO, MODE>), grid, block, 0, st, x, wp, scale, shift, res, y, \
                           dm, relu);                                                                             \
        return launch_status("conv3d_direct");                                                                    \
    }
    // CostRegNet (base 8) with 8/16/32-channel inputs, and the renderer's CostReg (41 -> 8)
    RCMVS_CONV_CASE(8, 8) RCMVS_CONV_CASE(16, 8) RCMVS_CONV_CASE(32, 8) RCMVS_CONV_CASE(41, 8) RCMVS_CONV_CASE(44, 8)
    RCMVS_CONV_CASE(8, 16) RCMVS_CONV_CASE(8, 32) RCMVS_CONV_CASE(8, 48) RCMVS_CONV_CASE(16, 16) RCMVS_CONV_CASE(16, 32) RCMVS_CONV_CASE(32, 32)
    RCMVS_CONV_CASE(32, 64) RCMVS_CONV_CASE(64, 64) RCMVS_CONV_CASE(64, 32) RCMVS_CONV_CASE(32, 16)
    RCMVS_CONV_CASE(8, 1)
#undef RCMVS_CONV_CASE
    return fail(-1, "conv3d: unsupported channel pair Ci=%d Co=%d", Ci, Co);
}

// conv3d_mfma.hip
int conv3d_mfma_launch(const float* x, const float* wm, const float* scale, const float* shift, const float* res,
                       float* y, int B, int D, int H, int W, int Ci, int Co, int mode, int relu, hipStream_t st, float* ymax = nullptr);
bool conv3d_mfma_supported(int Ci, int Co, int mode);
int pack_weight_mfma_launch(const float* w, float* packed, int Co, int Ci, int transposed, hipStream_t st);
long long mfma_weight_floats_host(int Ci, int Co);

// conv3d_lds.hip
bool conv3d_lds_supported(int Ci, int Co, int stride);
int conv3d_lds_launch(const float* x, const float* wp, const float* scale, const float* shift, const float* res, float* y,
                      int B, int D, int H, int W, int Ci, int Co, int relu, hipStream_t st, int lds_cfg);

// deconv3d_lds.hip
bool deconv3d_lds_supported(int Ci, int Co);
int deconv3d_lds_launch(const float* x, const float* wp, const float* scale, const float* shift, const float* res, float* y,
                        int B, int D, int H, int W, int Ci, int Co, int relu, hipStream_t st);

// conv3d_x3.hip (bf16 matrix cores, three-way split operands); kind: 0 stride-1 conv, 1 stride-2 conv, 2 transposed stride-2,
// 3 planar stride-1 (the kd = 1 taps only: every z-plane on its own -- exact for a one-plane volume)
enum { X3_KIND_PLANAR = 3, X3_NKINDS = 4 };
bool conv3d_x3_supported(int Ci, int Co, int kind);
long long conv3d_x3_weight_floats(int Ci, int Co, int kind);
int conv3d_x3_pack(const float* w, float* img, int Co, int Ci, int kind, int transposed, hipStream_t st);
int conv3d_x3_launch(const float* x, const float* wimg, const float* scale, const float* shift, const float* res, float* y,
                     int B, int D, int H, int W, int Ci, int Co, int kind, int relu, hipStream_t st, int max_blocks, int s2d = 0,
                     const float* xmax = nullptr, float* ymax = nullptr);
// the two-piece fp16 form of the same kernels (conv3d_x3.hip, NP = 2): needs a bound of max|x| (xmax of conv3d_x3_launch)
bool conv3d_x3h_supported(int Ci, int Co, int kind);
long long conv3d_x3h_weight_floats(int Ci, int Co, int kind);
int conv3d_x3h_pack(const float* w, float* img, int Co, int Ci, int kind, int transposed, const float* wsc, hipStream_t st);
int conv3d_x3_wscale(const float* w, int n, float* out, hipStream_t st);

// packed weight blob = [27][Ci][Co] (direct kernels), then the fp32-MFMA image when the pair has one, then the x3 images of
// the four kinds (each present when the pair has that kernel), then the fp16-pair images of the four kinds, then 4 floats
// {weight scale, 1 / scale, -, -} (what conv3d_x3_wscale computed for the pack kernels), then -- (Ci, Co) = (8, 1) only -- the
// depth head's fp16-pair image (prob_pair.hip)
static inline long long direct_weight_floats(int Ci, int Co) { return 27LL * Ci * Co; }
static inline long long x3_image_offset(int Ci, int Co, int kind) {
    long long off = direct_weight_floats(Ci, Co) + (conv3d_mfma_supported(Ci, Co, 0) ? mfma_weight_floats_host(Ci, Co) : 0);
    for (int k = 0; k < kind; ++k) off += conv3d_x3_weight_floats(Ci, Co, k);
    return off;
}
static inline long long x3h_image_offset(int Ci, int Co, int kind) {
    long long off = x3_image_offset(Ci, Co, X3_NKINDS);
    for (int k = 0; k < kind; ++k) off += conv3d_x3h_weight_floats(Ci, Co, k);
    return off;
}
// prob_pair.hip: the depth head's prob conv (8 -> 1) on the matrix cores; its weight image closes the blob of that channel pair
long long prob_pair_weight_floats();
int prob_pair_pack(const float* w, float* img, const float* wsc, hipStream_t st);
static inline bool has_prob_pair(int Ci, int Co) { return Ci == 8 && Co == 1; }
static inline long long prob_pair_image_offset(int Ci, int Co) { return x3h_image_offset(Ci, Co, X3_NKINDS) + 4; }
long long prob_pair_blob_offset() { return prob_pair_image_offset(8, 1); }           // (for depth_head.hip)
// conv11_prob.hip: the last transposed layer (16 -> 8) and the prob conv in one pass
bool conv11_prob_supported(int Dt, int Ht, int Wt);
int conv11_prob_launch(const float* t, const float* w11img, const float* scale, const float* shift, const float* res, const float* wprobimg,
                       const float* tmax, const float* rmax, const float* coef, float* y, int B, int Dt, int Ht, int Wt, int zc_force, hipStream_t st,
                       const float* planes, float* depth, float* conf);

}  // namespace rcmvs

using namespace rcmvs;

// `impl` of the rcmvs_debug_* twins (tests, A/B benches; 0 = the production dispatch): bit 0 = direct kernels only;
// bits 1-3 and 5 = tuning configuration of the LDS-halo kernel; bit 4 = fp32-MFMA kernel before the LDS kernel where both
// exist; bit 6 = skip the split-bf16 MFMA kernels (conv3d_x3.hip); bits 8-15 = block count of the x3 kernels; bits 16-23 = z chunk
// of the marching prob conv; bit 25 = its generic (predicated) form where the plain one would run.  Passed by value: the library keeps no dispatch state.
struct ConvImpl {
    bool direct, prefer_lds, no_x3, ysq;
    int lds_cfg, x3_blocks;
    explicit ConvImpl(int on) : direct(on & 1), prefer_lds(!(on & 16)), no_x3((on >> 6) & 1), ysq((on >> 24) & 1),
                                lds_cfg(((on >> 1) & 7) | (((on >> 5) & 1) << 3) | (((on >> 25) & 1) << 4) | (((on >> 16) & 0xff) << 8)), x3_blocks((on >> 8) & 0xff) {}
};

extern "C" {

long long rcmvs_packed_weight_floats(int Co, int Ci) {
    if (Co <= 0 || Ci <= 0) return -1;
    long long n = direct_weight_floats(Ci, Co);
    if (conv3d_mfma_supported(Ci, Co, 0)) n += mfma_weight_floats_host(Ci, Co);
    for (int k = 0; k < X3_NKINDS; ++k) n += conv3d_x3_weight_floats(Ci, Co, k) + conv3d_x3h_weight_floats(Ci, Co, k);
    return n + 4 + (has_prob_pair(Ci, Co) ? prob_pair_weight_floats() : 0);
}

// Which kernel family a forward call lands on (shared by the dispatchers and by the selective weight pack below, so the two cannot
// disagree).  planar = stride-1 conv of a one-plane volume; scaled = the caller supplied an activation bound.
enum ConvSel { SEL_X3_PLANAR, SEL_X3H, SEL_X3, SEL_LDS, SEL_MFMA, SEL_DIRECT };
static ConvSel conv_select(int Ci, int Co, int mode, bool planar, bool scaled, const ConvImpl& im) {
    const bool x3 = !im.direct && !im.no_x3;
    if (mode == CONV_T2) {
        if (scaled && x3 && conv3d_x3h_supported(Ci, Co, CONV_T2)) return SEL_X3H;
        if (x3 && conv3d_x3_supported(Ci, Co, CONV_T2)) return SEL_X3;
        if (deconv3d_lds_supported(Ci, Co) && !im.direct) return SEL_LDS;
        if (conv3d_mfma_supported(Ci, Co, CONV_T2) && !im.direct) return SEL_MFMA;
        return SEL_DIRECT;
    }
    const int stride = mode == CONV_S1 ? 1 : 2;
    if (planar && x3 && conv3d_x3_supported(Ci, Co, X3_KIND_PLANAR)) return SEL_X3_PLANAR;      // one plane: the kd = 0, 2 taps only see padding
    if (scaled && x3 && conv3d_x3h_supported(Ci, Co, mode)) return SEL_X3H;
    if (x3 && conv3d_x3_supported(Ci, Co, mode)) return SEL_X3;
    if (im.prefer_lds && conv3d_lds_supported(Ci, Co, stride) && !im.direct) return SEL_LDS;
    if (conv3d_mfma_supported(Ci, Co, mode) && !im.direct) return SEL_MFMA;
    if (conv3d_lds_supported(Ci, Co, stride) && !im.direct) return SEL_LDS;
    return SEL_DIRECT;
}

// Images of a packed-weight blob as a bit mask: 1 = direct / LDS-halo layout, 2 = fp32-MFMA fragments, 4 << k = split-bf16 image of
// kind k, 256 << k = fp16-pair image of kind k, 1 << 16 = the depth head's fp16-pair image (8 -> 1 only).
enum { IMG_DIRECT = 1, IMG_MFMA = 2, IMG_X3 = 4, IMG_X3H = 256, IMG_PROB_PAIR = 1 << 16, IMG_ALL = 0x7fffffff };

int rcmvs_conv3d_images(int Co, int Ci, int stride, int transposed, int planar) {
    // the image the PRODUCTION dispatch (rcmvs_conv3d_fwd / rcmvs_deconv3d_fwd, no activation bound) reads for this layer
    const int mode = transposed ? CONV_T2 : (stride == 1 ? CONV_S1 : CONV_S2);
    switch (conv_select(Ci, Co, mode, !transposed && stride == 1 && planar, false, ConvImpl(0))) {
        case SEL_X3_PLANAR: return IMG_X3 << X3_KIND_PLANAR;
        case SEL_X3: return IMG_X3 << mode;
        case SEL_X3H: return IMG_X3H << mode;
        case SEL_MFMA: return IMG_MFMA;
        default: return IMG_DIRECT;
    }
}

// images: mask of the images to write (IMG_ALL = every image this channel pair has).  Training re-packs ~200 weights per iteration
// for one forward kernel each: the full blob costs up to 8 launches per weight, the selected image one (round 3).
int rcmvs_pack_conv3d_weight_sel(const float* w, float* packed, int Co, int Ci, int transposed, int images, void* stream) {
    RCMVS_REQUIRE(w && packed && Co > 0 && Ci > 0, "pack_conv3d_weight: bad arguments");
    int n = 27 * Ci * Co;
    int rc = 0;
    if (images & IMG_DIRECT) {
        hipLaunchKernelGGL(pack_weight_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), w, packed, Co, Ci, transposed);
        rc = launch_status("pack_conv3d_weight");
        if (rc) return rc;
    }
    if ((images & IMG_MFMA) && conv3d_mfma_supported(Ci, Co, 0)) {
        rc = pack_weight_mfma_launch(w, packed + direct_weight_floats(Ci, Co), Co, Ci, transposed, as_stream(stream));
        if (rc) return rc;
    }
    for (int k = 0; k < X3_NKINDS; ++k) {         // a ConvTranspose3d weight (transposed == 1) feeds the transposed kernel only, and vice versa
        if (!(images & (IMG_X3 << k)) || !conv3d_x3_supported(Ci, Co, k) || ((k == 2) != (transposed == 1))) continue;
        rc = conv3d_x3_pack(w, packed + x3_image_offset(Ci, Co, k), Co, Ci, k, transposed, as_stream(stream));
        if (rc) return rc;
    }
    float* wsc = packed + x3h_image_offset(Ci, Co, X3_NKINDS);
    bool scaled = false;
    for (int k = 0; k < X3_NKINDS; ++k) {
        if (!(images & (IMG_X3H << k)) || !conv3d_x3h_supported(Ci, Co, k) || ((k == 2) != (transposed == 1))) continue;
        if (!scaled) { rc = conv3d_x3_wscale(w, n, wsc, as_stream(stream)); if (rc) return rc; scaled = true; }
        rc = conv3d_x3h_pack(w, packed + x3h_image_offset(Ci, Co, k), Co, Ci, k, transposed, wsc, as_stream(stream));
        if (rc) return rc;
    }
    if ((images & IMG_PROB_PAIR) && has_prob_pair(Ci, Co) && !transposed) {
        if (!scaled) { rc = conv3d_x3_wscale(w, n, wsc, as_stream(stream)); if (rc) return rc; }
        rc = prob_pair_pack(w, packed + prob_pair_image_offset(Ci, Co), wsc, as_stream(stream));
        if (rc) return rc;
    }
    return 0;
}

int rcmvs_pack_conv3d_weight(const float* w, float* packed, int Co, int Ci, int transposed, void* stream) {
    return rcmvs_pack_conv3d_weight_sel(w, packed, Co, Ci, transposed, IMG_ALL, stream);
}

// xmax / ymax (the `scaled` entry points): device scalars; xmax = a bound of max|x| (selects the fp16-pair matrix-core form where the
// channel pair has one), ymax = receives max|y| (the matrix-core kernels -- split-operand and fp32 -- maintain it: an error elsewhere)
static int conv3d_dispatch(const float* x, const float* w_packed, const float* scale, const float* shift,
                           const float* residual, float* y,
                           int B, int D, int H, int W, int Ci, int Co, int stride, int relu, void* stream, const ConvImpl& im,
                           const float* xmax = nullptr, float* ymax = nullptr) {
    RCMVS_REQUIRE(x && w_packed && y, "conv3d_fwd: null pointer");
    RCMVS_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, "conv3d_fwd: bad sizes");
    RCMVS_REQUIRE(stride == 1 || stride == 2, "conv3d_fwd: stride must be 1 or 2");
    RCMVS_REQUIRE((scale == nullptr) == (shift == nullptr), "conv3d_fwd: scale and shift go together");
    ConvDims dm{B, D, H, W, (D - 1) / stride + 1, (H - 1) / stride + 1, (W - 1) / stride + 1};
    hipStream_t st = as_stream(stream);
    const int mode = stride == 1 ? CONV_S1 : CONV_S2;
    const ConvSel sel = conv_select(Ci, Co, mode, stride == 1 && D == 1, xmax != nullptr, im);
    RCMVS_REQUIRE(!im.ysq || (sel == SEL_X3_PLANAR && ymax), "conv3d_scaled_fwd: the squared output bound (impl bit 24) is kept by the planar matrix-core kernels only");
    if (sel == SEL_X3_PLANAR)
        return conv3d_x3_launch(x, w_packed + x3_image_offset(Ci, Co, X3_KIND_PLANAR), scale, shift, residual, y, B, D, H, W, Ci, Co, X3_KIND_PLANAR, relu, st, im.x3_blocks, im.ysq ? 2 : 0, nullptr, ymax);
    if (sel == SEL_X3H)
        return conv3d_x3_launch(x, w_packed + x3h_image_offset(Ci, Co, mode), scale, shift, residual, y, B, D, H, W, Ci, Co, mode, relu, st, im.x3_blocks, 0, xmax, ymax);
    if (sel == SEL_X3)
        return conv3d_x3_launch(x, w_packed + x3_image_offset(Ci, Co, mode), scale, shift, residual, y, B, D, H, W, Ci, Co, mode, relu, st, im.x3_blocks, 0, nullptr, ymax);
    if (sel == SEL_MFMA)
        return conv3d_mfma_launch(x, w_packed + direct_weight_floats(Ci, Co), scale, shift, residual, y, B, D, H, W, Ci, Co,
                                  mode, relu, st, ymax);
    RCMVS_REQUIRE(!ymax, "conv3d_scaled_fwd: no matrix-core kernel for Ci=%d Co=%d stride=%d, the output bound cannot be maintained", Ci, Co, stride);
    if (sel == SEL_LDS)
        return conv3d_lds_launch(x, w_packed, scale, shift, residual, y, B, D, H, W, Ci, Co, relu, st, im.lds_cfg);
    if (stride == 1) return direct_dispatch<CONV_S1>(x, w_packed, scale, shift, residual, y, dm, Ci, Co, relu, st);
    return direct_dispatch<CONV_S2>(x, w_packed, scale, shift, residual, y, dm, Ci, Co, relu, st);
}

int rcmvs_conv3d_fwd(const float* x, const float* w_packed, const float* scale, const float* shift,
                     const float* residual, float* y,
                     int B, int D, int H, int W, int Ci, int Co, int stride, int relu, void* stream) {
    return conv3d_dispatch(x, w_packed, scale, shift, residual, y, B, D, H, W, Ci, Co, stride, relu, stream, ConvImpl(0));
}

int rcmvs_conv3d_scaled_fwd(const float* x, const float* x_absmax, const float* w_packed, const float* scale, const float* shift,
                            const float* residual, float* y, float* y_absmax,
                            int B, int D, int H, int W, int Ci, int Co, int stride, int relu, int impl, void* stream) {
    return conv3d_dispatch(x, w_packed, scale, shift, residual, y, B, D, H, W, Ci, Co, stride, relu, stream, ConvImpl(impl), x_absmax, y_absmax);
}

int rcmvs_conv2d_s2d_fwd(const float* x, const float* w_packed, const float* scale, const float* shift, float* y,
                         int N, int H, int W, int C, int Co, int relu, void* stream) {
    RCMVS_REQUIRE(x && w_packed && y, "conv2d_s2d_fwd: null pointer");
    RCMVS_REQUIRE(N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "conv2d_s2d_fwd: H and W must be even (got %d x %d)", H, W);
    RCMVS_REQUIRE((scale == nullptr) == (shift == nullptr), "conv2d_s2d_fwd: scale and shift go together");
    RCMVS_REQUIRE(conv3d_x3_supported(4 * C, Co, X3_KIND_PLANAR), "conv2d_s2d_fwd: no planar split-bf16 kernel for %d -> %d channels", 4 * C, Co);
    return conv3d_x3_launch(x, w_packed + x3_image_offset(4 * C, Co, X3_KIND_PLANAR), scale, shift, nullptr, y, N, 1, H / 2, W / 2, 4 * C, Co,
                            X3_KIND_PLANAR, relu, as_stream(stream), 0, 1);
}

int rcmvs_debug_conv3d_fwd(const float* x, const float* w_packed, const float* scale, const float* shift,
                           const float* residual, float* y,
                           int B, int D, int H, int W, int Ci, int Co, int stride, int relu, int impl, void* stream) {
    return conv3d_dispatch(x, w_packed, scale, shift, residual, y, B, D, H, W, Ci, Co, stride, relu, stream, ConvImpl(impl));
}

static int deconv3d_dispatch(const float* x, const float* w_packed, const float* scale, const float* shift,
                             const float* residual, float* y,
                             int B, int D, int H, int W, int Ci, int Co, int relu, void* stream, const ConvImpl& im,
                             const float* xmax = nullptr, float* ymax = nullptr) {
    RCMVS_REQUIRE(x && w_packed && y, "deconv3d_fwd: null pointer");
    RCMVS_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, "deconv3d_fwd: bad sizes");
    RCMVS_REQUIRE((scale == nullptr) == (shift == nullptr), "deconv3d_fwd: scale and shift go together");
    ConvDims dm{B, D, H, W, 2 * D, 2 * H, 2 * W};
    const ConvSel sel = conv_select(Ci, Co, CONV_T2, false, xmax != nullptr, im);
    if (sel == SEL_X3H)
        return conv3d_x3_launch(x, w_packed + x3h_image_offset(Ci, Co, CONV_T2), scale, shift, residual, y, B, D, H, W, Ci, Co, CONV_T2, relu,
                                as_stream(stream), im.x3_blocks, 0, xmax, ymax);
    if (sel == SEL_X3)
        return conv3d_x3_launch(x, w_packed + x3_image_offset(Ci, Co, CONV_T2), scale, shift, residual, y, B, D, H, W, Ci, Co, CONV_T2, relu,
                                as_stream(stream), im.x3_blocks, 0, nullptr, ymax);
    if (sel == SEL_MFMA)
        return conv3d_mfma_launch(x, w_packed + direct_weight_floats(Ci, Co), scale, shift, residual, y, B, D, H, W, Ci, Co,
                                  CONV_T2, relu, as_stream(stream), ymax);
    RCMVS_REQUIRE(!ymax, "deconv3d_scaled_fwd: no matrix-core kernel for Ci=%d Co=%d, the output bound cannot be maintained", Ci, Co);
    if (sel == SEL_LDS)
        return deconv3d_lds_launch(x, w_packed, scale, shift, residual, y, B, D, H, W, Ci, Co, relu, as_stream(stream));
    return direct_dispatch<CONV_T2>(x, w_packed, scale, shift, residual, y, dm, Ci, Co, relu, as_stream(stream));
}

int rcmvs_deconv3d_scaled_fwd(const float* x, const float* x_absmax, const float* w_packed, const float* scale, const float* shift,
                              const float* residual, float* y, float* y_absmax,
                              int B, int D, int H, int W, int Ci, int Co, int relu, int impl, void* stream) {
    return deconv3d_dispatch(x, w_packed, scale, shift, residual, y, B, D, H, W, Ci, Co, relu, stream, ConvImpl(impl), x_absmax, y_absmax);
}

int rcmvs_conv11_prob_fwd(const float* t, const float* t_absmax, const float* w11_packed, const float* scale, const float* shift,
                          const float* res, const float* res_absmax, const float* coef, const float* wprob_packed, float* logits,
                          const float* planes, float* depth, float* conf,
                          int B, int Dt, int Ht, int Wt, int zchunk, void* stream) {
    RCMVS_REQUIRE(t && t_absmax && w11_packed && scale && shift && res && res_absmax && coef && wprob_packed && (logits || depth), "conv11_prob_fwd: null pointer");
    RCMVS_REQUIRE(!depth || (planes && conf && Dt == 4), "conv11_prob_fwd: the one-launch head (depth != NULL) needs planes, conf and 2 Dt = 8 planes");
    RCMVS_REQUIRE(B > 0 && Dt > 0 && Ht > 0 && Wt > 0, "conv11_prob_fwd: bad sizes");
    RCMVS_REQUIRE(zchunk >= 0 && zchunk % 2 == 0, "conv11_prob_fwd: the z chunk must be even (got %d)", zchunk);
    RCMVS_REQUIRE(conv11_prob_supported(Dt, Ht, Wt), "conv11_prob_fwd: volume %dx%dx%d too large for 32-bit offsets", Dt, Ht, Wt);
    return conv11_prob_launch(t, w11_packed + x3h_image_offset(16, 8, CONV_T2), scale, shift, res, wprob_packed + prob_pair_image_offset(8, 1),
                              t_absmax, res_absmax, coef, logits, B, Dt, Ht, Wt, zchunk, as_stream(stream), planes, depth, conf);
}

int rcmvs_deconv3d_fwd(const float* x, const float* w_packed, const float* scale, const float* shift,
                       const float* residual, float* y,
                       int B, int D, int H, int W, int Ci, int Co, int relu, void* stream) {
    return deconv3d_dispatch(x, w_packed, scale, shift, residual, y, B, D, H, W, Ci, Co, relu, stream, ConvImpl(0));
}

int rcmvs_debug_deconv3d_fwd(const float* x, const float* w_packed, const float* scale, const float* shift,
                             const float* residual, float* y,
                             int B, int D, int H, int W, int Ci, int Co, int relu, int impl, void* stream) {
    return deconv3d_dispatch(x, w_packed, scale, shift, residual, y, B, D, H, W, Ci, Co, relu, stream, ConvImpl(impl));
}

}  // extern "C"
