// conv11 + prob in one pass (round 6): the last transposed convolution of the 3-D U-Net (16 -> 8, stride 2, BatchNorm, ReLU, + conv0 skip:
// models/modules.py:486,497-499) and the prob conv on its output (8 -> 1, 3x3x3: models/modules.py:489,500), so that the 8-channel full-resolution
// volume between them never reaches HBM.  In the two-launch form that volume is written once (84 MB at stages 2 / 3 of a DTU scene) and read
// once by the depth head: 168 of the ~270 MB the two launches move.  gfx950 only; fp16-pair arithmetic (conv3d_x3.hip: two fp16 pieces per
// operand after an exact power-of-two pre-scale, v_mfma_f32_16x16x32_f16, three MFMAs per product), for callers that hand over activation bounds
// -- the B = 1 inference scene.
//
// A block (256 threads) owns a 12 x 28 pixel tile of the output and marches over a z chunk.  A STEP s takes the input cell planes s, s + 1
// (9 x 17 cells of 16 channels: the tile's cells, one halo cell row / column before and two after, split once into fp16 pieces, ring of three
// slots in LDS) and produces the two output planes 2 s, 2 s + 1 of the 14 x 30 halo tile the prob conv needs:
//   transposed conv   the GEMM of conv3d_x3.hip's transposed kind with ITS weight image (fragments register-stationary): N = 16 input cells of a
//                     cell row, M = (output parity class (pd, ph, pw), co) = 4 m-tiles (pd, ph) x 16 rows (pw, co), K step = (input plane kd,
//                     cell row r) x {columns c = 0, 1} x 16 channels; 9 of the 16 (K step, m-tile) fragments hold taps (kd <= pd, r <= ph).  A wave
//                     owns two of the tile's eight cell rows: 54 MFMAs per step.
//   epilogue          un-scale x BatchNorm scale + shift, ReLU, + skip (conv0's output, 16 B per lane: a wave reads a contiguous 1 KiB row
//                     segment; the loads are issued before the MFMAs), zero outside the volume (the prob conv's zero padding), split into fp16
//                     pieces, 8 B per lane and piece into the plane's LDS buffer (a wave writes one 512-byte row).
//   prob conv         prob_pair.hip's GEMM with ITS weight image: M = (kd, kw) partial sums, K = (kh, c), three rotations of the weight
//                     fragments so that an output plane accumulates in one lane group over its three input planes; the kw shifts are two DPP
//                     row shifts.  18 MFMAs per wave and plane; a finished plane of logits is stored (4 B per lane).
// The bound of the intermediate volume (the scale of its fp16 pieces) is not measured -- it would need a pass over a tensor that no longer
// exists -- but derived:  max|x8| <= max|conv0| + c1 max|t| + c2,  c1 = max_co |scale_co| max_p sum_{ci, taps of class p} |w|, c2 = max_co |shift_co|
// (coef = {c1, c2}, computed by the caller from the layer's parameters).  It is loose by the usual factor between an L1 bound and a typical sum
// (~ 2^3): the fp16 pair keeps 22 significant bits of every value down to 2^-18 of the bound, so that costs nothing measurable.
#include "common.h"
#include "x3_pieces.h"
#include <atomic>
#include <type_traits>

#ifndef CP_ABL
#define CP_ABL 0            // tools/dev/build_c11_variants.sh builds timing ablations (WRONG results): bit 0 no skip loads, 1 no transposed-conv MFMAs,
#endif                      // 2 no prob conv, 3 no logit stores, 4 no input fetch / stash after the prologue, 5 no epilogue LDS stores

namespace rcmvs {

constexpr int CP_TH = 12, CP_TW = 28;                          // output pixels per tile (even; TW = 2 x 14: two n-tiles of the prob GEMM)
constexpr int CP_HH = CP_TH + 2, CP_HWU = CP_TW + 2;           // halo tile of the intermediate volume: 14 rows x 30 columns used ...
constexpr int CP_HW = 32;                                      // ... of 32 stored per row: a 512-byte row keeps the rows of a B fragment in different banks
constexpr int CP_XPIECE = CP_HH * CP_HW * 16;                  // bytes of one piece plane (8 fp16 per voxel)
constexpr int CP_XBUF = 2 * CP_XPIECE;                         // hi + lo
constexpr int CP_CR = CP_TH / 2 + 2, CP_CC = CP_TW / 2 + 2;    // cells whose outputs the halo tile needs: 8 rows x 16 columns (= one n-tile per row)
constexpr int CP_IR = CP_CR + 1, CP_IC = CP_CC + 1;            // input cells read: 9 x 17
constexpr int CP_IPIECE = CP_IR * CP_IC * 32;                  // bytes of one piece plane of the input (16 fp16 per cell)
constexpr int CP_IBUF = 2 * CP_IPIECE;
constexpr int CP_NLD = (CP_IR * CP_IC * 4 + 255) / 256;        // float4 per thread per input plane
constexpr int CP_NJ = 2, CP_STEP = 14;                         // n-tiles of the prob GEMM per row, columns between their starts
constexpr int CP_WFRAG = 9 + 6;                                // weight fragments parked in LDS (1 KiB each): the low pieces of the transposed conv's nine live
                                                               // fragments and the prob conv's three rotations x two pieces (60 registers otherwise: 256 + scratch)
constexpr int CP_LGW = 3 * CP_TW * 4;                          // bytes of a wave's three logit rows of one plane
constexpr int cp_lds(int fuse_d) { return 2 * CP_XBUF + 2 * CP_IBUF + CP_WFRAG * 1024 + fuse_d * 4 * CP_LGW + 16; }
// index of the live fragment (K step j = (kd, r), m-tile mt = (pd, ph)) among the nine: kd <= pd and r <= ph
__host__ __device__ constexpr int cp_live_index(int j, int mt) {
    int k = 0;
    for (int jj = 0; jj < 4; ++jj)
        for (int m = 0; m < 4; ++m) {
            if (!((jj >> 1) <= (m >> 1) && (jj & 1) <= (m & 1))) continue;
            if (jj == j && m == mt) return k;
            ++k;
        }
    return -1;
}
static_assert(cp_live_index(3, 3) == 8, "nine live fragments");
static_assert(CP_CC == 16 && CP_CR == 8, "one n-tile per cell row, two cell rows per wave");

// FUSE_D = 8 (the cascade's last stage; one chunk = the whole depth): the logits stay in LDS and the softmax / soft-argmin / confidence of
// depth_head.hip finish in the same launch (depth, conf; the logits are not stored)
template <int FUSE_D>
__global__ __launch_bounds__(256, 2) void conv11_prob_kernel(
    const float* __restrict__ t, const x3_u32x4* __restrict__ w11, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ res, const x3_u32x4* __restrict__ wprob, const float* __restrict__ tmax, const float* __restrict__ rmax,
    const float* __restrict__ coef, float* __restrict__ y, int Dt, int Ht, int Wt, int tiles_w, int zchunk,
    const float* __restrict__ planes, float* __restrict__ depth, float* __restrict__ conf) {
    extern __shared__ __attribute__((aligned(16))) x3_byte smem[];
    x3_byte* const xb = smem;                                   // two plane buffers of the intermediate volume: [pd]
    x3_byte* const ib = smem + 2 * CP_XBUF;                     // two input planes (plane s in slot s & 1: plane s + 2 is parked over plane s once the step's MFMAs are done)
    x3_byte* const wl = ib + 2 * CP_IBUF;                       // weight fragments [CP_WFRAG][64 lanes][16 B]
    constexpr int OOB = 0x7ffffff0;
    const int D = 2 * Dt, H = 2 * Ht, W = 2 * Wt;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const int b = blockIdx.z, zc = blockIdx.y;
    const unsigned t2 = xcd_remap(blockIdx.x, gridDim.x);
    const int tw = t2 % tiles_w, th = t2 / tiles_w;
    const int h0 = th * CP_TH, w0 = tw * CP_TW, i0 = h0 / 2, j0 = w0 / 2;
    const int z0 = zc * zchunk, z1 = min(D, z0 + zchunk);      // output planes z0 .. z1 - 1 (z0, zchunk even); intermediate planes z0 - 1 .. z1
    const int nplanes = z1 - z0 + 2;
    const int s0 = z0 / 2 - 1;                                  // the step that makes plane z0 - 1
    // ---- bounds and scales (the first requests of the kernel)
    const float tmax_lane = tmax[lane * 16], rmax_lane = rmax[lane * 16];
    const float c1 = coef[0], c2 = coef[1];
    const float whdr11 = reinterpret_cast<const float*>(w11)[1], whdrp = reinterpret_cast<const float*>(wprob)[1];
    __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(t + (long long)b * Dt * Ht * Wt * 16), (short)0, Dt * Ht * Wt * 64, 0x00020000);
    __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(res + (long long)b * D * H * W * 8), (short)0, D * H * W * 32, 0x00020000);
    __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(y ? y + (long long)b * D * H * W : nullptr, (short)0, y ? D * H * W * 4 : 0, 0x00020000);
    const int iplane_bytes = Ht * Wt * 64;
    // this thread's share of an input plane: element e = (cell, float4 of its 16 channels)
    int goff[CP_NLD], ls[CP_NLD];
#pragma unroll
    for (int i = 0; i < CP_NLD; ++i) {
        const int e = tid + i * 256;
        const int cell = e >> 2, c4 = e & 3;
        const int rr = cell / CP_IC, cc = cell - rr * CP_IC;
        const int ir = i0 - 1 + rr, ic = j0 - 1 + cc;
        const bool has = e < CP_IR * CP_IC * 4;
        goff[i] = (has && ir >= 0 && ir < Ht && ic >= 0 && ic < Wt) ? ((ir * Wt + ic) * 16 + c4 * 4) * 4 : OOB;
        ls[i] = has ? cell * 32 + c4 * 8 : -1;
    }
    x3_u32x4 pf[CP_NLD];
    auto fetch_to = [&](x3_u32x4 (&dst)[CP_NLD], int z) {
        const bool zin = z >= 0 && z < Dt;
        const int zoff = zin ? z * iplane_bytes : 0;
#pragma unroll
        for (int i = 0; i < CP_NLD; ++i) dst[i] = __builtin_amdgcn_raw_buffer_load_b128(trs, zin ? goff[i] : OOB, zoff, 0);
    };
    auto fetch = [&](int z) { fetch_to(pf, z); };
    // ---- weights: the 9 live fragments of the transposed conv (K step j = (kd, r), m-tile mt = (pd, ph): kd <= pd and r <= ph), two pieces each;
    // the three rotations of the prob conv's fragment
    x3_u32x4 A11[4][4];                                         // (high pieces: registers; low pieces and the prob conv's fragments: LDS, read per use)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
            if ((j >> 1) <= (mt >> 1) && (j & 1) <= (mt & 1)) {
                A11[j][mt] = w11[1 + ((j * 2 + 0) * 4 + mt) * 64 + lane];
                if (wave == 0) *reinterpret_cast<x3_u32x4*>(wl + cp_live_index(j, mt) * 1024 + lane * 16) = w11[1 + ((j * 2 + 1) * 4 + mt) * 64 + lane];
            }
    if (wave == 1) {
#pragma unroll
        for (int r = 0; r < 6; ++r) *reinterpret_cast<x3_u32x4*>(wl + (9 + r) * 1024 + lane * 16) = wprob[1 + r * 64 + lane];
    }
    const x3_byte* const wll = wl + lane * 16;
    float* const lg = reinterpret_cast<float*>(wl + CP_WFRAG * 1024 + wave * (FUSE_D * CP_LGW));      // (FUSE_D) this wave's logits [plane][3 rows x 28]
    const int sf = max(s0, 0);                                  // first step that exists
    x3_u32x4 pf1[CP_NLD];                                       // (prologue only: both input planes of the first step fly together -- one round trip, not two)
    fetch(sf);
    fetch_to(pf1, sf + 1);
    float bt = tmax_lane, br = rmax_lane;
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) { bt = fmaxf(bt, __shfl_xor(bt, m)); br = fmaxf(br, __shfl_xor(br, m)); }
    float tinv, x8inv;
    const float ts_scale = x3_pow2_scale(bt, tinv);
    const float xs8 = x3_pow2_scale(br + (c1 * bt + c2), x8inv);
    const float unscale11 = tinv * whdr11, unscalep = x8inv * whdrp;
    auto stash_from = [&](const x3_u32x4 (&src)[CP_NLD], int slot) {
#pragma unroll
        for (int i = 0; i < CP_NLD; ++i) {
            x3_u32x2 h, l;
            x3_split4h(__builtin_bit_cast(x3_f32x4, src[i]) * ts_scale, h, l);
            if (ls[i] >= 0) {
                *reinterpret_cast<x3_u32x2*>(ib + slot * CP_IBUF + ls[i]) = h;
                *reinterpret_cast<x3_u32x2*>(ib + slot * CP_IBUF + CP_IPIECE + ls[i]) = l;
            }
        }
    };
    auto stash = [&](int slot) { stash_from(pf, slot); };
    stash(sf & 1);
    stash_from(pf1, (sf + 1) & 1);
    // ---- transposed conv: lane geometry.  B fragment of K step (kd, r): 8 channels (half kk & 1) of cell (row cr + r, column n + c), c = kk >> 1
    const int g4 = kq;                                          // D fragment: rows 4 g4 .. 4 g4 + 3 of an m-tile = (pw = g4 >> 1, co0 = 4 (g4 & 1))
    const int pw = g4 >> 1, co0 = (g4 & 1) * 4;
    int ibaddr[2][2];                                           // [n-tile t][r]
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 2; ++r) ibaddr[tt][r] = ((2 * wave + tt + r) * CP_IC + n + (kq >> 1)) * 32 + (kq & 1) * 16;
    x3_f32x4 sc11 = *reinterpret_cast<const x3_f32x4*>(scale + co0) * unscale11;          // exact (power of two)
    const x3_f32x4 sh11 = *reinterpret_cast<const x3_f32x4*>(shift + co0);
    // epilogue geometry of (n-tile tt, m-tile mt = (pd, ph)): halo row 2 cr + ph - 1, halo column 2 n + pw - 1
    const int hc = 2 * n + pw - 1;
    const int ow = w0 - 1 + hc;
    const bool colok = hc >= 0 && hc < CP_HWU && ow >= 0 && ow < W;
    // ---- prob conv: lane geometry (prob_pair.hip): lane (n, kq) reads the 8 channels of halo voxel (row + kh, column 14 j + n), kh = kq
    int pbaddr[3][CP_NJ], ooff[3][CP_NJ];
#pragma unroll
    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
        for (int j = 0; j < CP_NJ; ++j) {
            const int row = 3 * wave + rr, q = min(CP_STEP * j + n, CP_HWU - 1);
            pbaddr[rr][j] = ((row + (kq < 3 ? kq : 0)) * CP_HW + q) * 16;
            const int wl = CP_STEP * j + n, oh = h0 + row, owp = w0 + wl;
            const bool ok = n < CP_STEP && wl < CP_TW && oh < H && owp < W;
            ooff[rr][j] = ok ? (oh * W + owp) * 4 : OOB;
        }
    x3_f32x4 pacc[3][CP_NJ];
#pragma unroll
    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
        for (int j = 0; j < CP_NJ; ++j) pacc[rr][j] = (x3_f32x4){0.f, 0.f, 0.f, 0.f};
    // both plane buffers start as zeros: voxels of the halo tile that lie outside the image (the prob conv's zero padding) are never written
    for (int o = tid * 16; o < 2 * CP_XBUF; o += 256 * 16) *reinterpret_cast<x3_u32x4*>(xb + o) = (x3_u32x4){0u, 0u, 0u, 0u};
    __syncthreads();

    // vmcnt counts loads AND stores, in order: a wait for a load also waits for every store issued before it, and a logit store's acknowledgement
    // takes microseconds under the skip volume's read stream (the first form of this kernel -- stores at the end of a step, the next step's loads
    // behind them -- lost 15 - 25 us per launch to that: profiles/r6_conv11_prob.txt).  So the loads of step s + 1 (skip values, input plane s + 3)
    // are requested at the END of step s, in front of the prob conv's stores of step s: nothing waits for a store but the next request point.
    // The row of a (n-tile, m-tile) pair is the same for the whole wave (halo row 4 wave + 2 tt + ph - 1), the column and the channel quad are the lane's
    // and the same for every pair: the lane part of every address below is ONE register computed once, the rest is scalar arithmetic (the first form
    // recomputed rows, bounds and offsets per lane, pair and step: 700 VALU instructions per wave and step against 90 MFMAs, profiles/r6_final_pmc.txt)
    const int rq_lane = (colok && !(CP_ABL & 1)) ? (ow * 8 + co0) * 4 : OOB;          // skip values: byte offset inside a row of the skip volume
    const bool wr_lane = colok && hc >= 0 && hc < CP_HWU && !(CP_ABL & 32);             // this lane's column of the halo tile lies inside the image: it is written
    const int xo_lane = hc * 16 + co0 * 2;                                              // its byte offset inside a row of the LDS plane
    x3_u32x4 rq[2][4];
    auto request = [&](int s, int need) {                       // need = 0: nothing (out-of-range offsets: no traffic)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int pd = mt >> 1, ph = mt & 1;
                const int hr = 2 * (2 * wave + tt) + ph - 1, oh = h0 - 1 + hr;                     // (scalar)
                const bool rowok = hr >= 0 && hr < CP_HH && oh >= 0 && oh < H && ((need >> pd) & 1);
                rq[tt][mt] = __builtin_amdgcn_raw_buffer_load_b128(rrs, rowok ? rq_lane : OOB, rowok ? ((2 * s + pd) * H + oh) * (W * 32) : 0, 0);
            }
        if (!(CP_ABL & 16)) fetch(need ? s + 2 : -1);
    };
    // ---- one step: intermediate planes 2 s + pd for the pd in `need` (bit pd) -> xb[pd]; (s_next, need_next): the step after it (need_next = 0: none)
    auto produce = [&](int s, int need, int need_next) {
        x3_f32x4 acc[2][4];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[tt][mt] = (x3_f32x4){0.f, 0.f, 0.f, 0.f};
        // K steps in two groups (kd = 0: j = 0, 1; kd = 1: j = 2, 3), the three products of the pair (hi x hi, hi x lo, lo x hi) outermost: an
        // accumulator meets its next MFMA 12 / 6 MFMAs later instead of 2 (single wave on the matrix pipe: a dependent MFMA stalls the issue)
#pragma unroll
        for (int kd = 0; kd < 2; ++kd) {
            const x3_byte* pb = ib + ((s + kd) & 1) * CP_IBUF;
            x3_u32x4 bh[2][2], bl[2][2];                        // [r][n-tile]
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    bh[r][tt] = *reinterpret_cast<const x3_u32x4*>(pb + ibaddr[tt][r]);
                    bl[r][tt] = *reinterpret_cast<const x3_u32x4*>(pb + CP_IPIECE + ibaddr[tt][r]);
                }
#pragma unroll
            for (int cls = 0; cls < 3; ++cls)
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        const int j = kd * 2 + r;
                        if (!(kd <= (mt >> 1) && r <= (mt & 1))) continue;          // (compile time: a zero fragment)
                        if (!((need >> (mt >> 1)) & 1)) continue;                   // (uniform: a plane this chunk does not need)
                        if (CP_ABL & 2) continue;
                        const x3_u32x4 a = cls == 2 ? *reinterpret_cast<const x3_u32x4*>(wll + cp_live_index(j, mt) * 1024) : A11[j][mt];
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt) acc[tt][mt] = x3_mfma<2>(a, cls == 1 ? bl[r][tt] : bh[r][tt], acc[tt][mt]);
                    }
        }
        __syncthreads();                                        // the previous step's planes have been read
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int pd = mt >> 1, ph = mt & 1;
                if (!((need >> pd) & 1)) continue;
                const int hr = 2 * (2 * wave + tt) + ph - 1, oh = h0 - 1 + hr;                     // (scalar)
                if (!(hr >= 0 && hr < CP_HH && oh >= 0 && oh < H)) continue;                      // a row outside the halo tile, or outside the image:
                x3_f32x4 v = acc[tt][mt] * sc11 + sh11;                                            // there the planes keep the zeros of the prologue
                v = __builtin_elementwise_max(v, (x3_f32x4){0.f, 0.f, 0.f, 0.f});                 // (the prob conv's zero padding)
                v += __builtin_bit_cast(x3_f32x4, rq[tt][mt]);
                x3_u32x2 h, l;
                x3_split4h(v * xs8, h, l);
                if (wr_lane) {
                    x3_byte* q = xb + (pd * CP_XBUF + hr * (CP_HW * 16)) + xo_lane;
                    *reinterpret_cast<x3_u32x2*>(q) = h;
                    *reinterpret_cast<x3_u32x2*>(q + CP_XPIECE) = l;
                }
            }
        if (!(CP_ABL & 16)) stash(s & 1);                       // plane s + 2 over plane s (every wave is past the step's MFMAs: the barrier above)
        request(s + 1, need_next);
        __syncthreads();
    };
    // ---- the prob conv over intermediate plane i of the chunk (global plane z0 - 1 + i) in xb[buf], rotation R = i mod 3; finishes the logits
    // of output plane z0 + i - 2 in lane group (2 - R) mod 3.  zero: the plane lies outside the volume (padding): nothing to add
    auto body = [&](auto rtag, int i, int buf, bool zero) {
        constexpr int R = decltype(rtag)::value, G = (2 - R + 3) % 3;
        const x3_byte* pb = xb + buf * CP_XBUF;
        if (!zero && !(CP_ABL & 4)) {
            const x3_u32x4 aph = *reinterpret_cast<const x3_u32x4*>(wll + (9 + R * 2) * 1024), apl = *reinterpret_cast<const x3_u32x4*>(wll + (9 + R * 2 + 1) * 1024);
            x3_u32x4 bh[CP_NJ][3], bl[CP_NJ][3];
#pragma unroll
            for (int j = 0; j < CP_NJ; ++j)
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) {
                    bh[j][rr] = *reinterpret_cast<const x3_u32x4*>(pb + pbaddr[rr][j]);
                    bl[j][rr] = *reinterpret_cast<const x3_u32x4*>(pb + CP_XPIECE + pbaddr[rr][j]);
                }
#pragma unroll
            for (int cls = 0; cls < 3; ++cls)
#pragma unroll
                for (int j = 0; j < CP_NJ; ++j)
#pragma unroll
                    for (int rr = 0; rr < 3; ++rr) pacc[rr][j] = x3_mfma<2>(cls == 2 ? apl : aph, cls == 1 ? bl[j][rr] : bh[j][rr], pacc[rr][j]);
        }
        const int o = i - 2;                                    // finished output plane (local)
        const bool mine = kq == G;
        const bool store = o >= 0 && o < z1 - z0 && !(CP_ABL & 8);
        const int zoff = store ? (z0 + o) * H * W * 4 : 0;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int j = 0; j < CP_NJ; ++j) {
                x3_f32x4& a = pacc[rr][j];
                const float a0 = a[0], a1 = a[1], a2 = a[2];
                const float q1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a1), 0x101, 0xf, 0xf, true));   // row_shl:1
                const float q2 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a2), 0x102, 0xf, 0xf, true));   // row_shl:2
                const float v = ((a0 + q1) + q2) * unscalep;
                if constexpr (FUSE_D > 0) {
                    if (mine && store && n < CP_STEP) lg[o * (3 * CP_TW) + rr * CP_TW + CP_STEP * j + n] = v;
                } else {
                    // (always issued -- a plane that is not stored goes to the out-of-range offset: the waits of the loads count a fixed number of younger stores)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yrs, (mine && store) ? ooff[rr][j] : OOB, zoff, 0);
                }
                a[0] = mine ? 0.0f : a[0]; a[1] = mine ? 0.0f : a[1]; a[2] = mine ? 0.0f : a[2];
            }
    };
    using R0 = std::integral_constant<int, 0>;
    using R1 = std::integral_constant<int, 1>;
    using R2 = std::integral_constant<int, 2>;
    // plane i: i = 0 is the pd = 1 plane of step s0, then (pd = 0, pd = 1) of the steps s0 + 1, ...; the last plane is a pd = 0 one.
    // Step s0 + st makes planes 2 st - 1 (pd = 0; st >= 1) and 2 st (pd = 1; st <= nst - 2) of the chunk.
    const int nst = (z1 - z0) / 2 + 2;
    auto need_of = [&](int st) -> int {
        const int s = s0 + st;
        if (st < 0 || st >= nst || s < 0 || s >= Dt) return 0;
        return (st >= 1 ? 1 : 0) | (st <= nst - 2 ? 2 : 0);
    };
    request(sf, need_of(sf - s0));                              // the first step's skip values and input plane sf + 2
    for (int i = 0; i < nplanes; ++i) {
        const int pd = (i & 1) ? 0 : 1;
        const int st = (i + 1) >> 1, s = s0 + st;
        const bool real = s >= 0 && s < Dt;
        if (real && (i == 0 || (i & 1))) produce(s, need_of(st), need_of(st + 1));
        const int r3 = i % 3;
        if (r3 == 0) body(R0{}, i, pd, !real);
        else if (r3 == 1) body(R1{}, i, pd, !real);
        else body(R2{}, i, pd, !real);
    }
    if constexpr (FUSE_D > 0) {
        // softmax over the planes, soft-argmin depth, confidence window (models/casmvsnet.py:293-309; the arithmetic of depth_head.hip); a wave
        // finishes the 3 x 28 pixels whose logits it produced itself
        __builtin_amdgcn_wave_barrier();
        const long long hw = (long long)H * W;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int pix = lane + 64 * pass;
            const int lr = pix / CP_TW, lc = pix - lr * CP_TW;
            const int oh = h0 + 3 * wave + lr, owp = w0 + lc;
            const bool live = pix < 3 * CP_TW && oh < H && owp < W;
            const int pi = min(pix, 3 * CP_TW - 1);
            float v[FUSE_D];
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < FUSE_D; ++k) { v[k] = lg[k * (3 * CP_TW) + pi]; mx = fmaxf(mx, v[k]); }
            float sum = 0.0f;
#pragma unroll
            for (int k = 0; k < FUSE_D; ++k) { v[k] = expf(v[k] - mx); sum += v[k]; }
            const long long gp = live ? (long long)oh * W + owp : 0;
            const float2 pl = reinterpret_cast<const float2*>(planes)[(long long)b * hw + gp];
            float dsum = 0.0f, isum = 0.0f;
#pragma unroll
            for (int k = 0; k < FUSE_D; ++k) {
                v[k] = v[k] / sum;
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
                depth[(long long)b * hw + gp] = dsum;
                conf[(long long)b * hw + gp] = c;
            }
        }
    }
}

bool conv11_prob_supported(int Dt, int Ht, int Wt) {
    return Dt > 0 && Ht > 0 && Wt > 0 && 8LL * Dt * Ht * Wt * 32 < 0x7ffffff0LL;
}

// z chunk (even): slots = CUs x 2 resident blocks; minimise rounds x steps per block (a chunk that is not the whole depth pays two halo steps)
static int conv11_prob_zchunk(long long tiles, int D) {
    static int slots = 0;
    if (!slots) {
        hipDeviceProp_t prop;
        int dev = 0;
        slots = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256) * 2;
    }
    long long best = -1; int zc = D;
    for (int c = D; c >= 2; c -= 2) {
        const long long blocks = tiles * ((D + c - 1) / c);
        const long long cost = ((blocks + slots - 1) / slots) * (c / 2 + (c < D ? 2 : 0));
        if (best < 0 || cost < best) { best = cost; zc = c; }
    }
    return zc;
}

// t (B, Dt, Ht, Wt, 16) -> logits (B, 2 Dt, 2 Ht, 2 Wt).  w11img: the fp16-pair image of the transposed 16 -> 8 layer (conv3d_x3h_pack), wprobimg:
// prob_pair_pack's image; tmax / rmax: bounds of max|t| / max|res| (ABSMAX slot format); coef: {c1, c2} on the device (see the header)
int conv11_prob_launch(const float* t, const float* w11img, const float* scale, const float* shift, const float* res, const float* wprobimg,
                       const float* tmax, const float* rmax, const float* coef, float* y, int B, int Dt, int Ht, int Wt, int zc_force, hipStream_t st,
                       const float* planes, float* depth, float* conf) {
    if (!conv11_prob_supported(Dt, Ht, Wt)) return fail(-1, "conv11_prob: volume too large for 32-bit offsets");
    const int D = 2 * Dt, H = 2 * Ht, W = 2 * Wt;
    const int tw_ = (W + CP_TW - 1) / CP_TW, th_ = (H + CP_TH - 1) / CP_TH;
    static std::atomic<bool> raised[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fail(-1, "conv11_prob: cannot query the device");
    if (!raised[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)conv11_prob_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, cp_lds(0)) != hipSuccess ||
            hipFuncSetAttribute((const void*)conv11_prob_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, cp_lds(8)) != hipSuccess)
            return fail(-1, "conv11_prob: cannot raise the dynamic LDS limit to %d bytes", cp_lds(8));
        raised[dev].store(true, std::memory_order_release);
    }
    if (depth) {            // the whole head in this launch: D = 8, one chunk
        if (D != 8 || !planes || !conf) return fail(-1, "conv11_prob: the one-launch head needs D = 8 (got %d), planes, depth and conf", D);
        hipLaunchKernelGGL(conv11_prob_kernel<8>, dim3(tw_ * th_, 1, B), dim3(256), cp_lds(8), st, t, reinterpret_cast<const x3_u32x4*>(w11img), scale, shift,
                           res, reinterpret_cast<const x3_u32x4*>(wprobimg), tmax, rmax, coef, (float*)nullptr, Dt, Ht, Wt, tw_, D, planes, depth, conf);
        return launch_status("conv11_prob (one-launch head)");
    }
    if (!y) return fail(-1, "conv11_prob: no output");
    int zc = zc_force > 0 ? zc_force : conv11_prob_zchunk((long long)B * tw_ * th_, D);
    zc = (zc + 1) & ~1;
    if (zc > D) zc = D;
    hipLaunchKernelGGL(conv11_prob_kernel<0>, dim3(tw_ * th_, (D + zc - 1) / zc, B), dim3(256), cp_lds(0), st, t, reinterpret_cast<const x3_u32x4*>(w11img), scale, shift,
                       res, reinterpret_cast<const x3_u32x4*>(wprobimg), tmax, rmax, coef, y, Dt, Ht, Wt, tw_, zc, (const float*)nullptr, (float*)nullptr, (float*)nullptr);
    return launch_status("conv11_prob");
}

}  // namespace rcmvs
