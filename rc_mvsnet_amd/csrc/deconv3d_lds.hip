// Transposed 3x3x3 convolution (stride 2, pad 1, output_pad 1) with few output channels -- conv11 of the 3-D U-Nets
// (16 -> 8 at full resolution, models/modules.py:486) and, in training, the data gradient of conv1 (8 -> 16, stride 2).
// Channels-last, LDS-staged input tile + wave-uniform scalar weights.  gfx950 only.
//
// An output voxel o = 2 q + p (p = parity per axis) receives, per axis, tap 1 of input q when p = 0 and taps {2 of q,
// 0 of q + 1} when p = 1, i.e. 1 / 2 / 4 / 8 taps for the 8 parity classes (27 in total).  One thread per OUTPUT voxel
// (the direct kernel) makes the tap set lane-dependent, which forces per-lane weight loads; here a lane is an INPUT
// cell q and a wave walks parity classes, so the tap set -- and with it every weight -- is wave-uniform:
//   block = 2 x 4 x 8 input cells (64 lanes = one wave per class pass) -> 4 x 8 x 16 outputs,
//   halo  = 3 x 5 x 9 input cells x CI channels staged once in LDS at a padded stride,
//   the 8 classes are dealt to the 4 waves as {7}, {6,1,0}, {5,2}, {3,4}: 8 / 7 / 6 / 6 taps each.
// Epilogue as everywhere: [relu](v * scale + shift) + residual.  The kernel is bound by its 8-channel output stream.
#include "common.h"

namespace rcmvs {

typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int DQ_D = 2, DQ_H = 4, DQ_W = 8;                        // input cells per block (64 = one wave)
constexpr int DH_D = DQ_D + 1, DH_H = DQ_H + 1, DH_W = DQ_W + 1;   // + the q + 1 neighbours
constexpr int DH_VOX = DH_D * DH_H * DH_W;                         // 135

template <int CI, int CO, int PD, int PH, int PW>
__device__ __forceinline__ void deconv_class(const float* __restrict__ tile, const float* __restrict__ wp,
                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                             const float* __restrict__ res, float* __restrict__ y,
                                             int qd, int qh, int qw, int gd, int gh, int gw, bool q_ok,
                                             long long ybase, int Ho, int Wo, int relu) {
    constexpr int STRIDE = CI + 4;
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = 0.0f;
    // the tap loop stays rolled: 128 scalar weights per tap already fill the SGPR file (an unrolled class-7 body would
    // want 1024 of them and spill through v_writelane / v_readlane)
    constexpr int NTAPS = (1 << PD) * (1 << PH) * (1 << PW);
#pragma unroll 1
    for (int t = 0; t < NTAPS; ++t) {
        // odd axis: bit = 0 -> tap 2 of cell q, bit = 1 -> tap 0 of cell q + 1; even axis: tap 1 of cell q
        const int c = PW ? (t & 1) : 0, b = PH ? ((t >> PW) & 1) : 0, a = PD ? ((t >> (PW + PH)) & 1) : 0;
        const int td = PD ? (a ? 0 : 2) : 1, th = PH ? (b ? 0 : 2) : 1, tw = PW ? (c ? 0 : 2) : 1;
        const float* tp = tile + (((qd + a) * DH_H + (qh + b)) * DH_W + (qw + c)) * STRIDE;
        const float* wt = wp + (long long)((td * 3 + th) * 3 + tw) * CI * CO;
#pragma unroll 2
        for (int c4 = 0; c4 < CI / 4; ++c4) {
            const f4v xv = *reinterpret_cast<const f4v*>(tp + c4 * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int co = 0; co < CO; ++co) acc[co] = fmaf(xv[j], wt[(c4 * 4 + j) * CO + co], acc[co]);
        }
    }
    if (!q_ok) return;
    const long long o = ybase + (((long long)(2 * gd + PD) * Ho + (2 * gh + PH)) * Wo + (2 * gw + PW)) * CO;
    if (scale) {
#pragma unroll
        for (int co = 0; co < CO; ++co) acc[co] = acc[co] * scale[co] + shift[co];
    }
    if (relu) {
#pragma unroll
        for (int co = 0; co < CO; ++co) acc[co] = fmaxf(acc[co], 0.0f);
    }
#pragma unroll
    for (int co = 0; co < CO; co += 4) {
        f4v v = (f4v){acc[co], acc[co + 1], acc[co + 2], acc[co + 3]};
        if (res) v += *reinterpret_cast<const f4v*>(res + o + co);
        *reinterpret_cast<f4v*>(y + o + co) = v;
    }
}

template <int CI, int CO>
__global__ __launch_bounds__(256) void deconv3d_lds_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ res, float* __restrict__ y,
    int D, int H, int W, int tiles_w, int tiles_h, int relu) {
    constexpr int STRIDE = CI + 4;
    __shared__ __attribute__((aligned(16))) float tile[DH_VOX * STRIDE];
    const int b = blockIdx.z, td = blockIdx.y;
    const unsigned t2 = xcd_remap(blockIdx.x, gridDim.x);
    const int tw = t2 % tiles_w, th = t2 / tiles_w;
    const int d0 = td * DQ_D, h0 = th * DQ_H, w0 = tw * DQ_W;
    const float* xb = x + (long long)b * D * H * W * CI;
    for (int e = threadIdx.x; e < DH_VOX * (CI / 4); e += 256) {
        const int v = e / (CI / 4), c4 = e - v * (CI / 4);
        const int hw_ = v % DH_W, hh = (v / DH_W) % DH_H, hd = v / (DH_W * DH_H);
        const int id = d0 + hd, ih = h0 + hh, iw = w0 + hw_;
        f4v val = (f4v){0.f, 0.f, 0.f, 0.f};
        if (id < D && ih < H && iw < W) val = *reinterpret_cast<const f4v*>(xb + (((long long)id * H + ih) * W + iw) * CI + c4 * 4);
        *reinterpret_cast<f4v*>(tile + v * STRIDE + c4 * 4) = val;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qw = lane % DQ_W, qh = (lane / DQ_W) % DQ_H, qd = lane / (DQ_W * DQ_H);
    const int gd = d0 + qd, gh = h0 + qh, gw = w0 + qw;
    const bool q_ok = gd < D && gh < H && gw < W;
    const int Ho = 2 * H, Wo = 2 * W;
    const long long ybase = (long long)b * (2 * D) * Ho * Wo * CO;
#define RCMVS_DC(PD, PH, PW) deconv_class<CI, CO, PD, PH, PW>(tile, wp, scale, shift, res, y, qd, qh, qw, gd, gh, gw, q_ok, ybase, Ho, Wo, relu)
    if (wave == 0)      { RCMVS_DC(1, 1, 1); }
    else if (wave == 1) { RCMVS_DC(1, 1, 0); RCMVS_DC(0, 0, 1); RCMVS_DC(0, 0, 0); }
    else if (wave == 2) { RCMVS_DC(1, 0, 1); RCMVS_DC(0, 1, 0); }
    else                { RCMVS_DC(0, 1, 1); RCMVS_DC(1, 0, 0); }
#undef RCMVS_DC
}

bool deconv3d_lds_supported(int Ci, int Co) { return Ci == 16 && Co == 8; }

int deconv3d_lds_launch(const float* x, const float* wp, const float* scale, const float* shift, const float* res, float* y,
                        int B, int D, int H, int W, int Ci, int Co, int relu, hipStream_t st) {
    (void)Ci; (void)Co;
    const int tiles_w = (W + DQ_W - 1) / DQ_W, tiles_h = (H + DQ_H - 1) / DQ_H, tiles_d = (D + DQ_D - 1) / DQ_D;
    dim3 grid(tiles_w * tiles_h, tiles_d, B);
    hipLaunchKernelGGL((deconv3d_lds_kernel<16, 8>), grid, dim3(256), 0, st, x, wp, scale, shift, res, y, D, H, W, tiles_w, tiles_h, relu);
    return launch_status("deconv3d_lds");
}

}  // namespace rcmvs
