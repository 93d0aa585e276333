// FeatureNet's last level with its two convolutions folded (fpn_fused.hip, round 4: out3(up2(prev) + inner2(lat) + b) =
// (W3 o W2) * lat  +  per-parity 2 x 2 blocks of W3 on `prev`  +  border-class bias; models/modules.py:413-452) on the matrix cores,
// exact: operands split into three bf16 pieces by truncation, six v_mfma_f32_16x16x32_bf16 per product (the arithmetic of
// conv3d_x3.hip, NP = 3), no bound needed.  gfx950 only.
//
// Why.  The VALU form spends 800 v_pk_fma_f32 per thread (4.7 issue clocks each: 24 us of the chip for the three views) and runs at
// 57 us per scene for 3.1 GFLOP and 94 MB.  As a GEMM per 16 PIXEL PAIRS (x = 2 n + px) of one output row:
//   M = (px, co) = 16 rows,  N = 16 pairs,
//   K = (dy, e, ci): the lateral map's 3 rows x 4 columns 2 n - 1 + e x 8 channels (weight (W3 o W2)[dy][e - px], zero outside 0..2)   3 k-steps
//     + (ry, e3, cj): `prev`'s 2 rows x 3 columns n - 1 + e3 x 32 channels (weight WA[py][px][ry][e3 - px], zero outside 0..1)         6 k-steps
// = 54 MFMAs per 32 pixels; the output fragment of a lane is four channels of one pixel: float4 stores, 1 KB contiguous per wave.
// Blocks are persistent (two per CU): a wave keeps the 27 weight fragments of its row parity in registers for the whole launch
// (re-loading them per tile would move 108 KB per 256 pixels through the vector L1), tiles of 8 x 32 pixels are staged through LDS
// (lateral tile de-interleaved by column parity so that a lane group reads 16 consecutive 16-byte voxels; `prev` voxels 80 bytes
// apart: conflict-free), the next tile's loads are in flight while the current one is computed.
#include "common.h"
#include "x3_pieces.h"

namespace rcmvs {

constexpr int FM_TY = 8, FM_TX = 32;
constexpr int FM_LH = FM_TY + 2, FM_LW = FM_TX + 2;            // lateral halo: 10 x 34 pixels
constexpr int FM_LROW = (FM_LW / 2) * 16;                      // bytes of one parity row of a piece plane (17 voxels x 8 bf16)
constexpr int FM_LPIECE = 2 * FM_LH * FM_LROW;                 // bytes of a lateral piece plane
constexpr int FM_UH = FM_TY / 2 + 2, FM_UW = FM_TX / 2 + 2;    // `prev` halo: 6 x 18 pixels
constexpr int FM_UVS = 80;                                     // bytes of a `prev` voxel in a piece plane (32 bf16 + 16)
constexpr int FM_UPIECE = FM_UH * FM_UW * FM_UVS;
constexpr int FM_UBASE = 3 * FM_LPIECE;
constexpr int FM_LDS = 3 * FM_LPIECE + 3 * FM_UPIECE;
constexpr int FM_NL = (FM_LH * FM_LW * 2 + 255) / 256;         // float4 per thread: lateral halo
constexpr int FM_NU = (FM_UH * FM_UW * 8 + 255) / 256;         // float4 per thread: `prev` halo
constexpr int FM_KSTEPS = 3 + 2 * 6;
constexpr int FM_IMG_FLOATS = FM_KSTEPS * 3 * 64 * 4;          // [k-step][piece][lane][8 bf16]
constexpr int FMT_WB = 0, FMT_BS = 576, FMT_WA = 576 + 72;     // layout of the fp32 tables (ops.pack_fpn_folded, fpn_fused.hip)

long long fpn_folded_mfma_floats() { return FM_IMG_FLOATS + 72; }        // image + the nine border-class biases

// fp32 tables -> A fragments (row m = lane & 15 = (px, co), k = 8 (lane >> 4) + i), three bf16 pieces by truncation; biases copied
__global__ void fpn_folded_mfma_pack_kernel(const float* __restrict__ tab, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 72) out[FM_IMG_FLOATS + t] = tab[FMT_BS + t];
    if (t >= FM_KSTEPS * 64 * 8) return;
    const int i = t & 7, lane = (t >> 3) & 63, j = t >> 9;
    const int m = lane & 15, kq = lane >> 4, px = m >> 3, co = m & 7;
    float v = 0.0f;
    if (j < 3) {
        const int dx = kq - px;
        if (dx >= 0 && dx <= 2) v = tab[FMT_WB + ((j * 3 + dx) * 8 + i) * 8 + co];
    } else {
        const int jj = j - 3, py = jj / 6, ry = (jj % 6) / 3, e3 = jj % 3, rx = e3 - px;
        if (rx >= 0 && rx <= 1) v = tab[FMT_WA + ((((py * 2 + px) * 2 + ry) * 2 + rx) * 32 + kq * 8 + i) * 8 + co];
    }
    const unsigned hb = __float_as_uint(v) & 0xffff0000u;
    const float r1 = v - __uint_as_float(hb);
    const unsigned mb = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(mb);
    const unsigned lb = __float_as_uint(r2) & 0xffff0000u;
    unsigned short* img = reinterpret_cast<unsigned short*>(out);
    const int base = (j * 3) * 512 + lane * 8 + i;
    img[base] = (unsigned short)(hb >> 16);
    img[base + 512] = (unsigned short)(mb >> 16);
    img[base + 1024] = (unsigned short)(lb >> 16);
}

__global__ __launch_bounds__(256, 2) void fpn_folded_mfma_kernel(
    const float* __restrict__ lat, const float* __restrict__ up, const x3_u32x4* __restrict__ img, const float* __restrict__ bias,
    float* __restrict__ y, float* __restrict__ ysq, int N, int H, int W, int tiles_w, int tiles_h) {
    __shared__ __attribute__((aligned(16))) x3_byte smem[FM_LDS];
    __shared__ float red[4];
    constexpr int OOB = 0x7ffffff0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const int py = wave & 1, rh = wave >> 1;                     // this wave: output rows py + 4 rh and py + 4 rh + 2 of the tile
    const int Hh = H / 2, Wh = W / 2;
    const int ntiles = N * tiles_h * tiles_w;
    __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(lat), (short)0, (int)((long long)N * H * W * 32), 0x00020000);
    __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(up), (short)0, (int)((long long)N * Hh * Wh * 128), 0x00020000);
    // weight fragments of this row parity: 3 (lateral rows) + 6 (`prev`: 2 rows x 3 columns) k-steps x 3 pieces
    x3_u32x4 A[9][3];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int p = 0; p < 3; ++p) A[k][p] = img[((k < 3 ? k : 3 + py * 6 + (k - 3)) * 3 + p) * 64 + lane];
    // this thread's shares of the two halo tiles (tile-independent parts)
    int lhy[FM_NL], lhx[FM_NL], lc4[FM_NL], lls[FM_NL], uur[FM_NU], uuc[FM_NU], uc4[FM_NU], uls[FM_NU];
#pragma unroll
    for (int i = 0; i < FM_NL; ++i) {
        const int e = tid + i * 256, v = e >> 1;
        lc4[i] = e & 1; lhy[i] = v / FM_LW; lhx[i] = v % FM_LW;
        lls[i] = e < FM_LH * FM_LW * 2 ? ((lhx[i] & 1) * FM_LH + lhy[i]) * FM_LROW + (lhx[i] >> 1) * 16 + lc4[i] * 8 : -1;
    }
#pragma unroll
    for (int i = 0; i < FM_NU; ++i) {
        const int e = tid + i * 256, v = e >> 3;
        uc4[i] = e & 7; uur[i] = v / FM_UW; uuc[i] = v % FM_UW;
        uls[i] = e < FM_UH * FM_UW * 8 ? FM_UBASE + v * FM_UVS + uc4[i] * 8 : -1;
    }
    x3_u32x4 pfl[FM_NL], pfu[FM_NU];
    auto fetch = [&](int tile) {
        const int tw = tile % tiles_w, th = (tile / tiles_w) % tiles_h, b = tile / (tiles_w * tiles_h);
        const int y0 = th * FM_TY, x0 = tw * FM_TX;
#pragma unroll
        for (int i = 0; i < FM_NL; ++i) {
            const int iy = y0 - 1 + lhy[i], ix = x0 - 1 + lhx[i];
            const bool in = lls[i] >= 0 && iy >= 0 && iy < H && ix >= 0 && ix < W;
            pfl[i] = __builtin_amdgcn_raw_buffer_load_b128(lrs, in ? (((b * H + iy) * W + ix) * 8 + lc4[i] * 4) * 4 : OOB, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < FM_NU; ++i) {
            const int iy = y0 / 2 - 1 + uur[i], ix = x0 / 2 - 1 + uuc[i];
            const bool in = uls[i] >= 0 && iy >= 0 && iy < Hh && ix >= 0 && ix < Wh;
            pfu[i] = __builtin_amdgcn_raw_buffer_load_b128(urs, in ? (((b * Hh + iy) * Wh + ix) * 32 + uc4[i] * 4) * 4 : OOB, 0, 0);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < FM_NL; ++i) {
            x3_u32x2 h, m, l;
            x3_split4(__builtin_bit_cast(x3_f32x4, pfl[i]), h, m, l);
            if (lls[i] >= 0) {
                *reinterpret_cast<x3_u32x2*>(smem + lls[i]) = h;
                *reinterpret_cast<x3_u32x2*>(smem + FM_LPIECE + lls[i]) = m;
                *reinterpret_cast<x3_u32x2*>(smem + 2 * FM_LPIECE + lls[i]) = l;
            }
        }
#pragma unroll
        for (int i = 0; i < FM_NU; ++i) {
            x3_u32x2 h, m, l;
            x3_split4(__builtin_bit_cast(x3_f32x4, pfu[i]), h, m, l);
            if (uls[i] >= 0) {
                *reinterpret_cast<x3_u32x2*>(smem + uls[i]) = h;
                *reinterpret_cast<x3_u32x2*>(smem + FM_UPIECE + uls[i]) = m;
                *reinterpret_cast<x3_u32x2*>(smem + 2 * FM_UPIECE + uls[i]) = l;
            }
        }
    };
    float vmax = 0.0f;
    int tile = blockIdx.x;
    if (tile < ntiles) fetch(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        stash();
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
        const int tw = tile % tiles_w, th = (tile / tiles_w) % tiles_h, b = tile / (tiles_w * tiles_h);
        const int y0 = th * FM_TY, x0 = tw * FM_TX;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int r = py + 4 * rh + 2 * t;                   // local output row
            x3_f32x4 acc[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] = (x3_f32x4){0.f, 0.f, 0.f, 0.f};
            auto six = [&](const x3_u32x4 (&a)[3], const x3_u32x4 (&bq)[3]) {
                acc[2] = x3_mfma<3>(a[0], bq[2], acc[2]);
                acc[1] = x3_mfma<3>(a[0], bq[1], acc[1]);
                acc[0] = x3_mfma<3>(a[0], bq[0], acc[0]);
                acc[2] = x3_mfma<3>(a[1], bq[1], acc[2]);
                acc[1] = x3_mfma<3>(a[1], bq[0], acc[1]);
                acc[2] = x3_mfma<3>(a[2], bq[0], acc[2]);
            };
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {                     // lateral map: halo row r + dy, halo column 2 n + kq
                const int off = ((kq & 1) * FM_LH + r + dy) * FM_LROW + (n + (kq >> 1)) * 16;
                x3_u32x4 bq[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) bq[p] = *reinterpret_cast<const x3_u32x4*>(smem + p * FM_LPIECE + off);
                six(A[dy], bq);
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) {                        // `prev`: halo row (r >> 1) + py + ry, halo column n + e3, channels 8 kq ..
                const int ry = k / 3, e3 = k % 3;
                const int off = FM_UBASE + (((r >> 1) + py + ry) * FM_UW + n + e3) * FM_UVS + kq * 16;
                x3_u32x4 bq[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) bq[p] = *reinterpret_cast<const x3_u32x4*>(smem + p * FM_UPIECE + off);
                six(A[3 + k], bq);
            }
            // lane (n, kq): pixel x = 2 n + (kq >> 1), channels 4 (kq & 1) .. + 3
            const int oy = y0 + r, ox = x0 + 2 * n + (kq >> 1);
            if (oy < H && ox < W) {
                const int cy = oy == 0 ? 0 : (oy == H - 1 ? 2 : 1), cx = ox == 0 ? 0 : (ox == W - 1 ? 2 : 1);
                const x3_f32x4 bs = *reinterpret_cast<const x3_f32x4*>(bias + (cy * 3 + cx) * 8 + (kq & 1) * 4);
                const x3_f32x4 v = (acc[0] + (acc[1] + acc[2])) + bs;
                *reinterpret_cast<x3_f32x4*>(y + (((long long)b * H + oy) * W + ox) * 8 + (kq & 1) * 4) = v;
                vmax = x3_absmax4(vmax, v);
            }
        }
        __syncthreads();                                         // the tile is consumed: the next one may be parked
    }
    if (ysq) {                                                   // (max |y|)^2 over the block's tiles: one atomic max into slot (block & 63)
#pragma unroll
        for (int k = 32; k > 0; k >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, k));
        if (lane == 0) red[wave] = vmax;
        __syncthreads();
        if (tid == 0) {
            const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            atomicMax(reinterpret_cast<unsigned int*>(ysq) + (blockIdx.x & 63) * 16, __float_as_uint(m * m));
        }
    }
}

}  // namespace rcmvs

using namespace rcmvs;

extern "C" long long rcmvs_fpn_folded_mfma_floats(void) { return fpn_folded_mfma_floats(); }

extern "C" int rcmvs_fpn_folded_mfma_pack(const float* tables, float* image, void* stream) {
    RCMVS_REQUIRE(tables && image, "fpn_folded_mfma_pack: null pointer");
    hipLaunchKernelGGL(fpn_folded_mfma_pack_kernel, dim3((FM_KSTEPS * 64 * 8 + 255) / 256), dim3(256), 0, as_stream(stream), tables, image);
    return launch_status("fpn_folded_mfma_pack");
}

extern "C" int rcmvs_fpn_out_folded_mfma(const float* lat, const float* up, const float* image, float* y, float* ysq_absmax, int N, int H, int W, void* stream) {
    RCMVS_REQUIRE(lat && up && image && y, "fpn_out_folded_mfma: null pointer");
    RCMVS_REQUIRE(N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "fpn_out_folded_mfma: H and W must be even (got %d x %d)", H, W);
    RCMVS_REQUIRE((long long)N * H * W * 32 < 0x7ffffff0LL, "fpn_out_folded_mfma: maps too large for 32-bit offsets");
    const int tiles_w = (W + FM_TX - 1) / FM_TX, tiles_h = (H + FM_TY - 1) / FM_TY;
    static int cus = 0;
    if (!cus) {
        hipDeviceProp_t prop;
        int dev = 0;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    const long long ntiles = (long long)N * tiles_w * tiles_h;
    const int blocks = (int)(ntiles < 2LL * cus ? ntiles : 2LL * cus);
    hipLaunchKernelGGL(fpn_folded_mfma_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), lat, up, reinterpret_cast<const x3_u32x4*>(image),
                       image + FM_IMG_FLOATS, y, ysq_absmax, N, H, W, tiles_w, tiles_h);
    return launch_status("fpn_out_folded_mfma");
}
