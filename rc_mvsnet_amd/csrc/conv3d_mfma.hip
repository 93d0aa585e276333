// 3x3x3 convolution family as an implicit GEMM on v_mfma_f32_16x16x4_f32 (exact fp32: the MFMA is
// a k-ordered fmaf chain, cdna_hip_programming.md section 3), channels-last, for the layers with
// Cout >= 16 (conv1..conv9 of the 3-D U-Nets).  gfx950 only.
//
// GEMM view:  M = output channels (16 per m-tile), N = output "cells" (16 per n-tile),
//             K = (tap, input channel).
//   A[m][k] = weight, lane l supplies row m = l & 15, k-slot kq = l >> 4;
//   B[k][n] = input activation of cell n at the tap's offset, lane l supplies column n = l & 15,
//             k-slot kq = l >> 4;
//   D[m][n]: lane l holds rows (l >> 4) * 4 + r (r = 0..3) of column n = l & 15, i.e. FOUR
//             CONSECUTIVE OUTPUT CHANNELS of one cell -> the epilogue (BN scale/shift, ReLU,
//             skip-add) and the store are one float4 per lane.
// k-permutation: a lane loads VEC (= 4, or 2 when Cin = 8) consecutive input channels
//   cin = chunk*4*VEC + kq*VEC + j  with one 16/8-byte load and feeds component j to MFMA j; the
//   packed weights use the same order, so each (tap, chunk) costs one vector load per operand.
// A "cell" is an output voxel (conv) or an input-grid voxel (transposed conv, where the 8 output
//   parity classes 2*cell + p are separate grid.z slices so that the tap set is wave-uniform).
// One wave = NT n-tiles x one m-tile; grid.y = m-tiles.  Cells are flattened over (b, d, h, w), so
//   small deep-level volumes (e.g. 6x16x20) still fill whole tiles.
#include "common.h"

namespace rcmvs {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { MF_S1 = 0, MF_S2 = 1, MF_T2 = 2 };

struct MfmaDims {
    int B, D, H, W;      // input volume
    int Dg, Hg, Wg;      // cell grid (= output volume for conv, input volume for transposed)
    int Do, Ho, Wo;      // output volume
    long long cells;     // B*Dg*Hg*Wg
};

// floats of the MFMA weight image for (Ci, Co):  [27][chunks][mtiles][64 lanes][VEC]
__host__ __device__ inline int mfma_vec(int Ci) { return Ci >= 16 ? 4 : 2; }
__host__ __device__ inline long long mfma_weight_floats(int Ci, int Co) {
    int vec = mfma_vec(Ci);
    int chunks = Ci / (4 * vec), mtiles = (Co + 15) / 16;
    return 27LL * chunks * mtiles * 64 * vec;
}

__global__ void pack_weight_mfma_kernel(const float* __restrict__ w, float* __restrict__ packed, int Co, int Ci, int transposed) {
    const int vec = mfma_vec(Ci);
    const int chunks = Ci / (4 * vec), mtiles = (Co + 15) / 16;
    long long n = 27LL * chunks * mtiles * 64 * vec;
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    int j = (int)(t % vec);
    long long r = t / vec;
    int lane = (int)(r % 64); r /= 64;
    int mt = (int)(r % mtiles); r /= mtiles;
    int chunk = (int)(r % chunks);
    int tap = (int)(r / chunks);
    int m = lane & 15, kq = lane >> 4;
    int co = mt * 16 + m, ci = chunk * 4 * vec + kq * vec + j;
    float v = 0.0f;
    if (co < Co) v = transposed ? w[((long long)ci * Co + co) * 27 + (transposed == 2 ? 26 - tap : tap)] : w[((long long)co * Ci + ci) * 27 + tap];
    packed[t] = v;
}

// KS = K-split: with KS = 4 the block's four waves share ONE (n-tile, m-tile) and take every fourth tap each, then add
// their accumulators through LDS.  The deep U-Net levels have only a few hundred tiles (6x16x20 cells = 120 n-tiles): one
// wave per tile leaves most SIMDs empty and every wave walks a 27-tap dependent chain of load -> MFMA; splitting the taps
// gives 4x the waves and a 4x shorter chain (measured: the 64->64 level 49 -> 19 us, the stride-2 32->64 level 27 -> 13 us).
// (Round 3: nine waves of three taps, and a four-way split of the transposed form's parity classes, were measured: 64->64 14.6 -> 16.9 us,
// 32->64 11.4 -> 14.3, 64->32 transposed 14.8 -> 14.2 -- the tap chain is no longer what these launches wait for.)
template <int CIN, int COUT, int MODE, int NT, int KS = 1>
__global__ __launch_bounds__(256) void conv3d_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ wm, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ res, float* __restrict__ y, MfmaDims dm, int relu, float* __restrict__ ymax) {
    constexpr int VEC = (CIN >= 16) ? 4 : 2;
    constexpr int CHUNKS = CIN / (4 * VEC);
    constexpr int MTILES = (COUT + 15) / 16;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const int mt = blockIdx.y;
    const int pc = (MODE == MF_T2) ? blockIdx.z : 0;                 // output parity class (transposed only)
    const int pd = (pc >> 2) & 1, ph = (pc >> 1) & 1, pw = pc & 1;
    static_assert(KS == 1 || NT == 1, "K-split works on single-tile waves");
    const long long tile0 = (KS == 1) ? ((long long)blockIdx.x * 4 + wave) * NT : (long long)blockIdx.x;
    if (KS == 1 && tile0 * 16 >= dm.cells) return;
    int tap_turn = 0;                                                  // K-split: taps are dealt round-robin to the waves

    // per n-tile cell coordinates of this lane
    int cb[NT], cd[NT], ch[NT], cw[NT];
    bool cv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        long long c = (tile0 + t) * 16 + n;
        cv[t] = c < dm.cells;
        long long cc = cv[t] ? c : 0;
        cw[t] = (int)(cc % dm.Wg); cc /= dm.Wg;
        ch[t] = (int)(cc % dm.Hg); cc /= dm.Hg;
        cd[t] = (int)(cc % dm.Dg);
        cb[t] = (int)(cc / dm.Dg);
    }
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    constexpr int MF_OOB = 0x7ffffff0;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), (short)0,
                                                                    (int)((long long)dm.B * dm.D * dm.H * dm.W * CIN * 4), 0x00020000);
    const float* wbase = wm + ((long long)mt * 64 + lane) * VEC;
    constexpr long long W_TAP_STRIDE = (long long)CHUNKS * MTILES * 64 * VEC;
    constexpr long long W_CHUNK_STRIDE = (long long)MTILES * 64 * VEC;

    for (int kd = 0; kd < 3; ++kd) {
        if (MODE == MF_T2 && ((pd + 1 - kd) & 1)) continue;          // wave-uniform: tap does not hit this parity
        if (MODE == MF_S1 && dm.D == 1 && kd != 1) continue;         // single-plane volume = a 2-D 3x3 conv (FeatureNet 32->32)
        for (int kh = 0; kh < 3; ++kh) {
            if (MODE == MF_T2 && ((ph + 1 - kh) & 1)) continue;
            for (int kw = 0; kw < 3; ++kw) {
                if (MODE == MF_T2 && ((pw + 1 - kw) & 1)) continue;
                if (KS > 1 && (tap_turn++ % KS) != wave) continue;      // wave-uniform
                const int tap = (kd * 3 + kh) * 3 + kw;
                // activation operand through the buffer descriptor: a 32-bit byte offset per n-tile, pushed past num_records
                // for taps outside the volume so the bounds check returns 0 -- unconditional loads the scheduler can hoist
                // across taps (the exec-masked global loads of the first version were scheduling barriers)
                int off[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    int id, ih, iw;
                    if (MODE == MF_S1) { id = cd[t] + kd - 1; ih = ch[t] + kh - 1; iw = cw[t] + kw - 1; }
                    else if (MODE == MF_S2) { id = 2 * cd[t] + kd - 1; ih = 2 * ch[t] + kh - 1; iw = 2 * cw[t] + kw - 1; }
                    else { id = cd[t] + ((pd + 1 - kd) >> 1); ih = ch[t] + ((ph + 1 - kh) >> 1); iw = cw[t] + ((pw + 1 - kw) >> 1); }
                    const bool ok = cv[t] && id >= 0 && id < dm.D && ih >= 0 && ih < dm.H && iw >= 0 && iw < dm.W;
                    off[t] = ok ? ((((cb[t] * dm.D + id) * dm.H + ih) * dm.W + iw) * CIN + kq * VEC) * 4 : MF_OOB;
                }
                const float* wt = wbase + (long long)tap * W_TAP_STRIDE;
#pragma unroll
                for (int c = 0; c < CHUNKS; ++c) {
                    float a[VEC];
                    if (VEC == 4) {
                        float4 a4 = *reinterpret_cast<const float4*>(wt + c * W_CHUNK_STRIDE);
                        a[0] = a4.x; a[1] = a4.y; a[2 % VEC] = a4.z; a[3 % VEC] = a4.w;
                    } else {
                        float2 a2 = *reinterpret_cast<const float2*>(wt + c * W_CHUNK_STRIDE);
                        a[0] = a2.x; a[1] = a2.y;
                    }
                    float bv[NT][VEC];
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        if (VEC == 4) {
                            const f32x4 b4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[t] + c * 4 * VEC * 4, 0, 0));
                            bv[t][0] = b4.x; bv[t][1] = b4.y; bv[t][2 % VEC] = b4.z; bv[t][3 % VEC] = b4.w;
                        } else {
                            typedef float f32x2 __attribute__((ext_vector_type(2)));
                            const f32x2 b2 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, off[t] + c * 4 * VEC * 4, 0, 0));
                            bv[t][0] = b2.x; bv[t][1] = b2.y;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < VEC; ++j)
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], bv[t][j], acc[t], 0, 0, 0);
                }
            }
        }
    }

    if constexpr (KS > 1) {
        __shared__ f32x4 red[KS][64];
        red[wave][lane] = acc[0];
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int k = 1; k < KS; ++k) acc[0] += red[k][lane];
    }
    // epilogue: this lane owns output channels m0..m0+3 of cell n of every n-tile
    const int m0 = mt * 16 + kq * 4;
    if (m0 >= COUT) return;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (scale) { sc = *reinterpret_cast<const float4*>(scale + m0); sh = *reinterpret_cast<const float4*>(shift + m0); }
    float vmax = 0.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (!cv[t]) continue;
        long long ov;
        if (MODE == MF_T2)
            ov = (((long long)cb[t] * dm.Do + 2 * cd[t] + pd) * dm.Ho + 2 * ch[t] + ph) * dm.Wo + 2 * cw[t] + pw;
        else
            ov = (((long long)cb[t] * dm.Do + cd[t]) * dm.Ho + ch[t]) * dm.Wo + cw[t];
        float4 v = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
        if (scale) { v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w; }
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (res) {
            float4 r4 = *reinterpret_cast<const float4*>(res + ov * COUT + m0);
            v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
        }
        *reinterpret_cast<float4*>(y + ov * COUT + m0) = v;
        vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    // optional bound of the stored outputs (the slot format of conv3d_x3.hip: 64 slots, 16 floats apart, the bound is their maximum):
    // one atomic max per finishing wave -- what the fp16-pair form of the NEXT layer scales its activations with
    if (ymax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
        if (lane == 0) atomicMax(reinterpret_cast<unsigned int*>(ymax) + ((blockIdx.x * 4 + wave) & 63) * 16, __float_as_uint(vmax));
    }
}

bool conv3d_mfma_supported(int Ci, int Co, int mode) {
    (void)mode;
    if (Co < 16 || (Co % 16)) return false;
    return Ci == 8 || Ci == 16 || Ci == 32 || Ci == 64;
}

template <int MODE>
static int mfma_dispatch(const float* x, const float* wm, const float* scale, const float* shift, const float* res,
                         float* y, const MfmaDims& dm, int Ci, int Co, int relu, hipStream_t st, float* ymax) {
    const long long ntiles = cdiv(dm.cells, 16);
    // few tiles (deep U-Net levels): one n-tile per wave so that every SIMD gets work
    const bool small = ntiles < 4096;
    const int nt = small ? 1 : 4;
    // small launches: one (n-tile, m-tile) per BLOCK, taps split over its four waves (transposed convs keep one wave per
    // tile: a parity class has 1-8 taps, too few to split)
    const bool ksplit = small && MODE != MF_T2;
    dim3 grid((unsigned)(ksplit ? ntiles : cdiv(ntiles, 4LL * nt)), (Co + 15) / 16, MODE == MF_T2 ? 8 : 1), block(256);
#define RCMVS_MFMA_CASE(CI, CO)                                                                                     \
    if (Ci == CI && Co == CO) {                                                                                     \
        if (ksplit)     hipLaunchKernelGGL((conv3d_mfma_kernel<CI, CO, MODE, 1, 4>), grid, block, 0, st, x, wm, scale, shift, res, y, dm, relu, ymax); \
        else if (small) hipLaunchKernelGGL((conv3d_mfma_kernel<CI, CO, MODE, 1>), grid, block, 0, st, x, wm, scale, shift, res, y, dm, relu, ymax); \
        else            hipLaunchKernelGGL((conv3d_mfma_kernel<CI, CO, MODE, 4>), grid, block, 0, st, x, wm, scale, shift, res, y, dm, relu, ymax); \
        return launch_status("conv3d_mfma");                                                                        \
    }
    RCMVS_MFMA_CASE(8, 16) RCMVS_MFMA_CASE(8, 32) RCMVS_MFMA_CASE(8, 48) RCMVS_MFMA_CASE(16, 16) RCMVS_MFMA_CASE(16, 32) RCMVS_MFMA_CASE(32, 32)
    RCMVS_MFMA_CASE(32, 64) RCMVS_MFMA_CASE(64, 64) RCMVS_MFMA_CASE(64, 32) RCMVS_MFMA_CASE(32, 16)
#undef RCMVS_MFMA_CASE
    (void)nt;
    return fail(-1, "conv3d_mfma: unsupported channel pair Ci=%d Co=%d", Ci, Co);
}

// mode: 0 stride-1 conv, 1 stride-2 conv, 2 transposed stride-2
int conv3d_mfma_launch(const float* x, const float* wm, const float* scale, const float* shift, const float* res,
                       float* y, int B, int D, int H, int W, int Ci, int Co, int mode, int relu, hipStream_t st, float* ymax) {
    MfmaDims dm;
    dm.B = B; dm.D = D; dm.H = H; dm.W = W;
    if (mode == MF_T2) { dm.Dg = D; dm.Hg = H; dm.Wg = W; dm.Do = 2 * D; dm.Ho = 2 * H; dm.Wo = 2 * W; }
    else {
        int s = mode == MF_S2 ? 2 : 1;
        dm.Do = (D - 1) / s + 1; dm.Ho = (H - 1) / s + 1; dm.Wo = (W - 1) / s + 1;
        dm.Dg = dm.Do; dm.Hg = dm.Ho; dm.Wg = dm.Wo;
    }
    dm.cells = (long long)B * dm.Dg * dm.Hg * dm.Wg;
    if ((long long)B * D * H * W * Ci * 4 >= 0x7ffffff0LL) return fail(-1, "conv3d_mfma: input volume too large for 32-bit offsets");
    if (mode == MF_S1) return mfma_dispatch<MF_S1>(x, wm, scale, shift, res, y, dm, Ci, Co, relu, st, ymax);
    if (mode == MF_S2) return mfma_dispatch<MF_S2>(x, wm, scale, shift, res, y, dm, Ci, Co, relu, st, ymax);
    return mfma_dispatch<MF_T2>(x, wm, scale, shift, res, y, dm, Ci, Co, relu, st, ymax);
}

int pack_weight_mfma_launch(const float* w, float* packed, int Co, int Ci, int transposed, hipStream_t st) {
    long long n = mfma_weight_floats(Ci, Co);
    hipLaunchKernelGGL(pack_weight_mfma_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, w, packed, Co, Ci, transposed);
    return launch_status("pack_weight_mfma");
}

long long mfma_weight_floats_host(int Ci, int Co) { return mfma_weight_floats(Ci, Co); }

}  // namespace rcmvs
