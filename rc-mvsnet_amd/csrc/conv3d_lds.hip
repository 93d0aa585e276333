// Stride-1 3x3x3 convolution with few output channels (Cout = 8: conv0 of every 3-D U-Net), channels-last,
// LDS-staged halo tile + wave-uniform (scalar-register) weights.  gfx950 only.
//
// Why not MFMA here: fp32 MFMA runs at the fp32 VALU rate on gfx950 and a 16x16 tile is half empty at
// Cout = 8, while `v_fmac_f32 acc, x, s_weight` has no such waste and takes its weight operand straight
// from an SGPR (weights are identical for every lane).  Why LDS: with one thread per voxel the 16-byte
// channel-vector loads of neighbouring lanes are Cin*4 bytes apart, i.e. every load instruction touches
// 64 cache lines and uses 16 bytes of each -- the direct kernel is vL1D-access bound at Cin = 32/16
// (rocprof: 23.7 TF at stage 1).  Here a block stages the (2+2) x (8+2) x (16+2) input halo of its
// 2 x 8 x 16 output tile once per 16-channel chunk with coalesced loads, at a padded per-voxel stride
// (chunk + 4 floats) that makes the per-lane ds_read_b128 bank-conflict free; every tap is then one LDS
// read feeding 4 x Cout FMAs.
#include "common.h"

namespace rcmvs {

constexpr int LT_D = 2, LT_H = 8, LT_W = 16;                       // output tile (256 voxels, thread = voxel, w fastest)
constexpr int LH_D = LT_D + 2, LH_H = LT_H + 2, LH_W = LT_W + 2;   // input halo tile
constexpr int LH_VOX = LH_D * LH_H * LH_W;                         // 720

template <int CI, int CO>
__global__ __launch_bounds__(256) void conv3d_lds_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ res, float* __restrict__ y,
    int D, int H, int W, int tiles_w, int tiles_h, int relu) {
    constexpr int CK = (CI >= 16) ? 16 : CI;                        // channel chunk staged at a time
    constexpr int STRIDE = CK + 4;                                  // floats per staged voxel (padding kills bank conflicts)
    extern __shared__ __attribute__((aligned(16))) float tile[];    // [LH_VOX][STRIDE]
    const int b = blockIdx.z;
    const int td = blockIdx.y;
    const unsigned t2 = xcd_remap(blockIdx.x, gridDim.x);
    const int tw = t2 % tiles_w, th = t2 / tiles_w;
    const int d0 = td * LT_D, h0 = th * LT_H, w0 = tw * LT_W;
    const int lw = threadIdx.x % LT_W, lh = (threadIdx.x / LT_W) % LT_H, ld = threadIdx.x / (LT_W * LT_H);
    const int od = d0 + ld, oh = h0 + lh, ow = w0 + lw;
    const bool inside = od < D && oh < H && ow < W;
    const float* xb = x + (long long)b * D * H * W * CI;

    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = 0.0f;

    for (int c0 = 0; c0 < CI; c0 += CK) {
        const int ck = (CI - c0 < CK) ? (CI - c0) : CK;            // channels in this chunk (multiple of 4)
        const int q = ck >> 2;                                      // float4 per voxel
        if (c0 > 0) __syncthreads();
        // ---- stage the halo tile of this channel chunk (zero outside the volume)
        for (int e = threadIdx.x; e < LH_VOX * q; e += 256) {
            const int v = e / q, c4 = e - v * q;
            const int hw_ = v % LH_W, hh = (v / LH_W) % LH_H, hd = v / (LH_W * LH_H);
            const int id = d0 + hd - 1, ih = h0 + hh - 1, iw = w0 + hw_ - 1;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (id >= 0 && id < D && ih >= 0 && ih < H && iw >= 0 && iw < W)
                val = *reinterpret_cast<const float4*>(xb + (((long long)id * H + ih) * W + iw) * CI + c0 + c4 * 4);
            *reinterpret_cast<float4*>(tile + v * STRIDE + c4 * 4) = val;
        }
        __syncthreads();
        // ---- 27 taps from LDS, weights wave-uniform
        for (int kd = 0; kd < 3; ++kd)
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float* tp = tile + (((ld + kd) * LH_H + (lh + kh)) * LH_W + (lw + kw)) * STRIDE;
                    const float* wt = wp + ((long long)((kd * 3 + kh) * 3 + kw) * CI + c0) * CO;
#pragma unroll
                    for (int c4 = 0; c4 < CK / 4; ++c4) {
                        if (c4 * 4 < ck) {
                            const float4 xv = *reinterpret_cast<const float4*>(tp + c4 * 4);
#pragma unroll
                            for (int co = 0; co < CO; ++co) {
                                acc[co] = fmaf(xv.x, wt[(c4 * 4 + 0) * CO + co], acc[co]);
                                acc[co] = fmaf(xv.y, wt[(c4 * 4 + 1) * CO + co], acc[co]);
                                acc[co] = fmaf(xv.z, wt[(c4 * 4 + 2) * CO + co], acc[co]);
                                acc[co] = fmaf(xv.w, wt[(c4 * 4 + 3) * CO + co], acc[co]);
                            }
                        }
                    }
                }
    }
    if (!inside) return;
    const long long ov = (((long long)b * D + od) * H + oh) * W + ow;
    float* yp = y + ov * CO;
    const float* rp = res ? res + ov * CO : nullptr;
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        float v = acc[co];
        if (scale) v = v * scale[co] + shift[co];
        if (relu) v = fmaxf(v, 0.0f);
        if (rp) v += rp[co];
        acc[co] = v;
    }
#pragma unroll
    for (int co = 0; co < CO; co += 4)
        *reinterpret_cast<float4*>(yp + co) = make_float4(acc[co], acc[co + 1], acc[co + 2], acc[co + 3]);
}

bool conv3d_lds_supported(int Ci, int Co, int stride) {
    return stride == 1 && Co == 8 && (Ci == 8 || Ci == 16 || Ci == 32 || Ci == 44);
}

int conv3d_lds_launch(const float* x, const float* wp, const float* scale, const float* shift, const float* res, float* y,
                      int B, int D, int H, int W, int Ci, int Co, int relu, hipStream_t st) {
    const int tiles_w = (W + LT_W - 1) / LT_W, tiles_h = (H + LT_H - 1) / LT_H, tiles_d = (D + LT_D - 1) / LT_D;
    dim3 grid(tiles_w * tiles_h, tiles_d, B), block(256);
    const int ck = Ci >= 16 ? 16 : Ci;
    const size_t lds = (size_t)LH_VOX * (ck + 4) * sizeof(float);
#define RCMVS_LDS_CASE(CI)                                                                                              \
    if (Ci == CI) {                                                                                                     \
        hipLaunchKernelGGL((conv3d_lds_kernel<CI, 8>), grid, block, lds, st, x, wp, scale, shift, res, y, D, H, W, tiles_w, \
                           tiles_h, relu);                                                                              \
        return launch_status("conv3d_lds");                                                                             \
    }
    RCMVS_LDS_CASE(8) RCMVS_LDS_CASE(16) RCMVS_LDS_CASE(32) RCMVS_LDS_CASE(44)
#undef RCMVS_LDS_CASE
    (void)Co;
    return fail(-1, "conv3d_lds: unsupported Ci=%d", Ci);
}

}  // namespace rcmvs
