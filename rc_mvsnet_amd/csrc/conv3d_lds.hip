// Stride-1 3x3x3 convolution with few output channels (Cout = 8: conv0 of every 3-D U-Net), channels-last,
// LDS-staged halo tile + wave-uniform (scalar-register) weights.  gfx950 only.
//
// Why not MFMA here: fp32 MFMA runs at the fp32 VALU rate on gfx950 and a 16x16 tile is half empty at
// Cout = 8, while `v_fmac_f32 acc, x, s_weight` has no such waste and takes its weight operand straight
// from an SGPR (weights are identical for every lane).  Why LDS: with one thread per voxel the 16-byte
// channel-vector loads of neighbouring lanes are Cin*4 bytes apart, i.e. every load instruction touches
// 64 cache lines and uses 16 bytes of each -- the direct kernel is vL1D-access bound at Cin = 32/16
// (rocprof: 23.7 TF at stage 1).  Here a block stages the (2+2) x (8+2) x (16+2) input halo of its
// 2 x 8 x 16 output tile once per 8-channel chunk with coalesced loads, at a padded per-voxel stride
// (chunk + 4 floats) that makes the per-lane ds_read_b128 bank-conflict free; every tap is then one LDS
// read feeding 4 x Cout FMAs.
#include "common.h"

namespace rcmvs {

constexpr int LT_D = 2, LT_H = 8, LT_W = 16;                       // output tile (256 voxels, thread = voxel, w fastest)
constexpr int LH_D = LT_D + 2, LH_H = LT_H + 2, LH_W = LT_W + 2;   // input halo tile
constexpr int LH_VOX = LH_D * LH_H * LH_W;                         // 720

// SPLIT = threads per output voxel: the CO output channels are divided between SPLIT wave groups
// (threads [256*g, 256*g+256) own channels [g*CO/SPLIT, (g+1)*CO/SPLIT)), which multiplies the waves per
// LDS byte -- the tile costs 34-58 KB, so at SPLIT = 1 only two 4-wave blocks fit a CU and every
// s_waitcnt on the shared LDS/scalar-load counter is exposed.
// PPT = output voxels per thread, stacked in depth (d and d + LT_D): every scalar weight then feeds PPT FMAs, which halves
// the scalar-load traffic and the s_waitcnt stalls behind it (the Cin = 32 / 16 conv0 layers sit at 44-70 % VALU busy).
template <int CI, int CO, int SPLIT, int CKT, int PPT = 1>
__global__ __launch_bounds__(256 * SPLIT) void conv3d_lds_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ res, float* __restrict__ y,
    int D, int H, int W, int tiles_w, int tiles_h, int relu) {
    constexpr int CK = (CI < CKT) ? CI : CKT;                       // channel chunk staged at a time
    constexpr int STRIDE = CK + 4;                                  // floats per staged voxel (padding kills bank conflicts)
    constexpr int TD = LT_D * PPT;                                  // output tile depth
    constexpr int HD = TD + 2;                                      // halo depth
    constexpr int HVOX = HD * LH_H * LH_W;
    static_assert(PPT == 1 || CO > 1, "the single-channel path keeps one voxel per thread");
    extern __shared__ __attribute__((aligned(16))) float tile[];    // [HVOX][STRIDE]
    const int b = blockIdx.z;
    const int td = blockIdx.y;
    const unsigned t2 = xcd_remap(blockIdx.x, gridDim.x);
    const int tw = t2 % tiles_w, th = t2 / tiles_w;
    const int d0 = td * TD, h0 = th * LT_H, w0 = tw * LT_W;
    constexpr int NT = 256 * SPLIT;                                // threads per block
    constexpr int COT = CO / SPLIT;                                 // output channels per thread
    const int vox = threadIdx.x % 256;
    // wave-uniform by construction; readfirstlane makes that provable so the weights stay on scalar loads
    const int grp = (SPLIT == 1) ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x / 256);
    const int cob = grp * COT;
    const int lw = vox % LT_W, lh = (vox / LT_W) % LT_H, ld = vox / (LT_W * LT_H);
    const int od = d0 + ld, oh = h0 + lh, ow = w0 + lw;
    const float* xb = x + (long long)b * D * H * W * CI;

    float acc[PPT][COT];
#pragma unroll
    for (int p = 0; p < PPT; ++p)
#pragma unroll
        for (int c = 0; c < COT; ++c) acc[p][c] = 0.0f;
    // single output channel (prob conv): two partial sums over even / odd input channels, so that channel pairs go
    // through v_pk_fma_f32 (x pair from the float4, weight pair from consecutive SGPRs) instead of 216 scalar FMAs
    typedef float f2v __attribute__((ext_vector_type(2)));
    f2v acc2 = (f2v){0.f, 0.f};

    for (int c0 = 0; c0 < CI; c0 += CK) {
        const int ck = (CI - c0 < CK) ? (CI - c0) : CK;            // channels in this chunk (multiple of 4)
        const int q = ck >> 2;                                      // float4 per voxel
        if (c0 > 0) __syncthreads();
        // ---- stage the halo tile of this channel chunk (zero outside the volume)
        for (int e = threadIdx.x; e < HVOX * q; e += NT) {
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
                    const float* wt = wp + ((long long)((kd * 3 + kh) * 3 + kw) * CI + c0) * CO + cob;
#pragma unroll
                    for (int c4 = 0; c4 < CK / 4; ++c4) {
                        if (c4 * 4 < ck) {
                            float4 xv[PPT];
#pragma unroll
                            for (int p = 0; p < PPT; ++p)
                                xv[p] = *reinterpret_cast<const float4*>(tp + p * (LT_D * LH_H * LH_W * STRIDE) + c4 * 4);
                            if constexpr (CO == 1) {
                                const f2v w01 = (f2v){wt[c4 * 4 + 0], wt[c4 * 4 + 1]}, w23 = (f2v){wt[c4 * 4 + 2], wt[c4 * 4 + 3]};
                                acc2 = __builtin_elementwise_fma((f2v){xv[0].x, xv[0].y}, w01, acc2);
                                acc2 = __builtin_elementwise_fma((f2v){xv[0].z, xv[0].w}, w23, acc2);
                                continue;
                            }
#pragma unroll
                            for (int co = 0; co < COT; ++co) {
                                const float w0 = wt[(c4 * 4 + 0) * CO + co], w1 = wt[(c4 * 4 + 1) * CO + co];
                                const float w2 = wt[(c4 * 4 + 2) * CO + co], w3 = wt[(c4 * 4 + 3) * CO + co];
#pragma unroll
                                for (int p = 0; p < PPT; ++p) {
                                    acc[p][co] = fmaf(xv[p].x, w0, acc[p][co]);
                                    acc[p][co] = fmaf(xv[p].y, w1, acc[p][co]);
                                    acc[p][co] = fmaf(xv[p].z, w2, acc[p][co]);
                                    acc[p][co] = fmaf(xv[p].w, w3, acc[p][co]);
                                }
                            }
                        }
                    }
                }
    }
    if constexpr (CO == 1) acc[0][0] = acc2.x + acc2.y;
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
        const int odp = od + p * LT_D;
        if (!(odp < D && oh < H && ow < W)) continue;
        const long long ov = (((long long)b * D + odp) * H + oh) * W + ow;
        float* yp = y + ov * CO + cob;
        const float* rp = res ? res + ov * CO + cob : nullptr;
#pragma unroll
        for (int co = 0; co < COT; ++co) {
            float v = acc[p][co];
            if (scale) v = v * scale[cob + co] + shift[cob + co];
            if (relu) v = fmaxf(v, 0.0f);
            if (rp) v += rp[co];
            acc[p][co] = v;
        }
        if constexpr (COT % 4 == 0) {
#pragma unroll
            for (int co = 0; co < COT; co += 4)
                *reinterpret_cast<float4*>(yp + co) = make_float4(acc[p][co], acc[p][co + 1], acc[p][co + 2], acc[p][co + 3]);
        } else {
#pragma unroll
            for (int co = 0; co < COT; ++co) yp[co] = acc[p][co];
        }
    }
}

// ---- prob conv (8 -> 1), plane-marching form ---------------------------------------------------------------------------
// The tile kernel above reads every input voxel 27 times from LDS for the single output channel (54 ds_read_b128 per voxel) and
// stages a 4 x 10 x 18 halo per 2 x 8 x 16 outputs (2.8x the tile).  Here a block owns an 8 x 32 pixel tile and MARCHES over z:
// input plane z is staged once (10 x 34 halo, 1.33x) and read 9 times (18 ds_read_b128 per voxel); its 3x3 neighbourhood feeds the
// three kd taps at once -- plane z contributes the kd = 0 term of out[z + 1], the kd = 1 term of out[z] and the kd = 2 term of
// out[z - 1] -- into three rolling accumulators, and out[z - 1] is complete when plane z is done.  Planes are double buffered in
// LDS (next plane prefetched into registers during the FMAs): one barrier per plane.  Weights are wave-uniform scalar loads;
// channel pairs go through v_pk_fma_f32 as in the tile kernel.
constexpr int PM_TH = 8, PM_TW = 32, PM_HH = PM_TH + 2, PM_HW = PM_TW + 2, PM_STRIDE = 12;
constexpr int PM_PLANE = PM_HH * PM_HW * PM_STRIDE;          // floats per staged plane (16,320 B)
constexpr int PM_NLD = (PM_HH * PM_HW * 2 + 255) / 256;      // float4 per thread per plane

typedef float pm_f2v __attribute__((ext_vector_type(2)));
// One halo row of a plane against its 72 weights (taps (kd, kh, 0..2) x 8 channels, 24 consecutive floats per kd at wrow, wrow + 72,
// wrow + 144): 36 v_pk_fma_f32 into the three rolling accumulators (see prob_conv_march_kernel for why this is hand-placed).
__device__ __forceinline__ void pm_row_fma(pm_f2v& acc_next, pm_f2v& acc_cur, pm_f2v& acc_prev, const pm_f2v (&xq)[3][4], const float* wrow) {
    typedef pm_f2v f2v;
#if defined(__AMDGCN__)
    asm volatile("s_load_dwordx16 s[16:31], %[wp], 0x0\n\ts_load_dwordx8 s[32:39], %[wp], 0x40\n\t"
                 "s_load_dwordx16 s[40:55], %[wp], 0x120\n\ts_load_dwordx8 s[56:63], %[wp], 0x160\n\t"
                 "s_load_dwordx16 s[64:79], %[wp], 0x240\n\ts_load_dwordx8 s[80:87], %[wp], 0x280\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "v_pk_fma_f32 %[a0], %[x00], s[16:17], %[a0]\n\tv_pk_fma_f32 %[a1], %[x00], s[40:41], %[a1]\n\tv_pk_fma_f32 %[a2], %[x00], s[64:65], %[a2]\n\t"
                 "v_pk_fma_f32 %[a0], %[x01], s[18:19], %[a0]\n\tv_pk_fma_f32 %[a1], %[x01], s[42:43], %[a1]\n\tv_pk_fma_f32 %[a2], %[x01], s[66:67], %[a2]\n\t"
                 "v_pk_fma_f32 %[a0], %[x02], s[20:21], %[a0]\n\tv_pk_fma_f32 %[a1], %[x02], s[44:45], %[a1]\n\tv_pk_fma_f32 %[a2], %[x02], s[68:69], %[a2]\n\t"
                 "v_pk_fma_f32 %[a0], %[x03], s[22:23], %[a0]\n\tv_pk_fma_f32 %[a1], %[x03], s[46:47], %[a1]\n\tv_pk_fma_f32 %[a2], %[x03], s[70:71], %[a2]\n\t"
                 "v_pk_fma_f32 %[a0], %[x10], s[24:25], %[a0]\n\tv_pk_fma_f32 %[a1], %[x10], s[48:49], %[a1]\n\tv_pk_fma_f32 %[a2], %[x10], s[72:73], %[a2]\n\t"
                 "v_pk_fma_f32 %[a0], %[x11], s[26:27], %[a0]\n\tv_pk_fma_f32 %[a1], %[x11], s[50:51], %[a1]\n\tv_pk_fma_f32 %[a2], %[x11], s[74:75], %[a2]\n\t"
                 "v_pk_fma_f32 %[a0], %[x12], s[28:29], %[a0]\n\tv_pk_fma_f32 %[a1], %[x12], s[52:53], %[a1]\n\tv_pk_fma_f32 %[a2], %[x12], s[76:77], %[a2]\n\t"
                 "v_pk_fma_f32 %[a0], %[x13], s[30:31], %[a0]\n\tv_pk_fma_f32 %[a1], %[x13], s[54:55], %[a1]\n\tv_pk_fma_f32 %[a2], %[x13], s[78:79], %[a2]\n\t"
                 "v_pk_fma_f32 %[a0], %[x20], s[32:33], %[a0]\n\tv_pk_fma_f32 %[a1], %[x20], s[56:57], %[a1]\n\tv_pk_fma_f32 %[a2], %[x20], s[80:81], %[a2]\n\t"
                 "v_pk_fma_f32 %[a0], %[x21], s[34:35], %[a0]\n\tv_pk_fma_f32 %[a1], %[x21], s[58:59], %[a1]\n\tv_pk_fma_f32 %[a2], %[x21], s[82:83], %[a2]\n\t"
                 "v_pk_fma_f32 %[a0], %[x22], s[36:37], %[a0]\n\tv_pk_fma_f32 %[a1], %[x22], s[60:61], %[a1]\n\tv_pk_fma_f32 %[a2], %[x22], s[84:85], %[a2]\n\t"
                 "v_pk_fma_f32 %[a0], %[x23], s[38:39], %[a0]\n\tv_pk_fma_f32 %[a1], %[x23], s[62:63], %[a1]\n\tv_pk_fma_f32 %[a2], %[x23], s[86:87], %[a2]\n\t"
                 : [a0] "+v"(acc_next), [a1] "+v"(acc_cur), [a2] "+v"(acc_prev)
                 : [wp] "s"(wrow),
                   [x00] "v"(xq[0][0]), [x01] "v"(xq[0][1]), [x02] "v"(xq[0][2]), [x03] "v"(xq[0][3]),
                   [x10] "v"(xq[1][0]), [x11] "v"(xq[1][1]), [x12] "v"(xq[1][2]), [x13] "v"(xq[1][3]),
                   [x20] "v"(xq[2][0]), [x21] "v"(xq[2][1]), [x22] "v"(xq[2][2]), [x23] "v"(xq[2][3])
                 : "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87");
#else       // host pass of hipcc / the CPU emulation of the tests (tests/emu): the same 36 fused multiply-adds in the same order
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            const float* wt = wrow + kw * 8 + 2 * pr;
            acc_next = __builtin_elementwise_fma(xq[kw][pr], (f2v){wt[0], wt[1]}, acc_next);
            acc_cur = __builtin_elementwise_fma(xq[kw][pr], (f2v){wt[72], wt[73]}, acc_cur);
            acc_prev = __builtin_elementwise_fma(xq[kw][pr], (f2v){wt[144], wt[145]}, acc_prev);
        }
#endif
}

// Occupancy: 32.6 KB of LDS per block = five blocks per CU, and the register budget is held to five waves per SIMD to match
// (round 3: at 107 VGPRs four blocks were resident, and the 1280 blocks of the two large stages ran as 1024 + 256 -- two rounds
// for 1.25 rounds of work).  The z chunk is chosen per launch (prob_zchunk below) so that the grid fills those 1280 slots once.
__global__ __launch_bounds__(256, 5) void prob_conv_march_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ res, float* __restrict__ y, int D, int H, int W, int tiles_w, int tiles_h, int relu, int zchunk) {
    typedef float f2v __attribute__((ext_vector_type(2)));
    __shared__ __attribute__((aligned(16))) float plane[2][PM_PLANE];
    const int b = blockIdx.z, zc = blockIdx.y;
    const unsigned t2 = xcd_remap(blockIdx.x, gridDim.x);
    const int tw = t2 % tiles_w, th = t2 / tiles_w;
    const int h0 = th * PM_TH, w0 = tw * PM_TW, z0 = zc * zchunk;
    const int z1 = min(D, z0 + zchunk);                          // outputs z0 .. z1-1, input planes z0-1 .. z1
    const int lw = threadIdx.x % PM_TW, lh = threadIdx.x / PM_TW;
    const float* xb = x + (long long)b * D * H * W * 8;
    // this thread's share of a plane's halo: element e = (halo voxel, float4 half)
    int goff[PM_NLD], loff[PM_NLD];
#pragma unroll
    for (int i = 0; i < PM_NLD; ++i) {
        const int e = threadIdx.x + i * 256;
        const int v = e >> 1, c4 = e & 1;
        const int hh = v / PM_HW, hw_ = v - hh * PM_HW;
        const int ih = h0 + hh - 1, iw = w0 + hw_ - 1;
        const bool ok = e < PM_HH * PM_HW * 2 && ih >= 0 && ih < H && iw >= 0 && iw < W;
        goff[i] = ok ? (ih * W + iw) * 8 + c4 * 4 : -1;
        loff[i] = (e < PM_HH * PM_HW * 2) ? v * PM_STRIDE + c4 * 4 : -1;
    }
    float4 pf[PM_NLD];
    auto fetch = [&](int z) {
        const bool zin = z >= 0 && z < D;
        const float* xp = xb + (long long)z * H * W * 8;
#pragma unroll
        for (int i = 0; i < PM_NLD; ++i)
            pf[i] = (zin && goff[i] >= 0) ? *reinterpret_cast<const float4*>(xp + goff[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PM_NLD; ++i)
            if (loff[i] >= 0) *reinterpret_cast<float4*>(&plane[buf][loff[i]]) = pf[i];
    };
    fetch(z0 - 1);
    stash(0);
    fetch(z0);
    __syncthreads();
    f2v acc_prev = (f2v){0.f, 0.f}, acc_cur = (f2v){0.f, 0.f};      // out[z-1] and out[z] while plane z is processed
    const int oh = h0 + lh, ow = w0 + lw;
    const bool live = oh < H && ow < W;
    int buf = 0;
    for (int z = z0 - 1; z <= z1; ++z) {
        f2v acc_next = (f2v){0.f, 0.f};                             // out[z+1]
        const float* tp0 = &plane[buf][(lh * PM_HW + lw) * PM_STRIDE];
        // One halo row at a time, as ONE block of hand-placed instructions: the row's 72 weights -- taps (kd, kh, 0..2) x 8
        // channels, 24 consecutive floats per kd -- are fetched with six scalar loads into s[16:87] (scalar cache), then the
        // 36 v_pk_fma_f32 of the row take them as their scalar operand.  Left to the optimiser (plain wp[...] indexing, rounds
        // 1-2) all 216 loop-invariant weights are hoisted out of the plane loop, the SGPR file overflows, 164 of them are parked
        // in VGPR lanes and every plane pays ~160 v_readlane_b32 next to its 108 v_pk_fma_f32: the kernel ran at 41 % of its
        // VALU floor (profiles/r3_prob_conv.txt); fencing the loads (asm loads, sched_barrier) only moved the spills.
        // Scalar loads return out of order: one s_waitcnt lgkmcnt(0) per row -- the fragment reads of the NEXT row are issued in
        // front of it, so LDS and scalar-cache latency overlap -- and five waves per SIMD cover the rest.
        f2v xr[2][3][4];
        auto read_row = [&](int kh, f2v (&q)[3][4]) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const float* tp = tp0 + (kh * PM_HW + kw) * PM_STRIDE;
                const float4 xa = *reinterpret_cast<const float4*>(tp), xc = *reinterpret_cast<const float4*>(tp + 4);
                q[kw][0] = (f2v){xa.x, xa.y}; q[kw][1] = (f2v){xa.z, xa.w}; q[kw][2] = (f2v){xc.x, xc.y}; q[kw][3] = (f2v){xc.z, xc.w};
            }
        };
        read_row(0, xr[0]);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            if (kh < 2) read_row(kh + 1, xr[(kh + 1) & 1]);
            f2v (&xq)[3][4] = xr[kh & 1];
            pm_row_fma(acc_next, acc_cur, acc_prev, xq, wp + kh * 24);
        }
        const int zo = z - 1;                                       // complete now
        if (live && zo >= z0 && zo < z1) {
            float v = acc_prev.x + acc_prev.y;
            if (scale) v = v * scale[0] + shift[0];
            if (relu) v = fmaxf(v, 0.0f);
            const long long ov = (((long long)b * D + zo) * H + oh) * W + ow;
            if (res) v += res[ov];
            y[ov] = v;
        }
        acc_prev = acc_cur; acc_cur = acc_next;
        if (z < z1) {
            stash(buf ^ 1);                                         // plane z+1 (fetched during the previous iteration)
            if (z + 2 <= z1) fetch(z + 2);
        }
        __syncthreads();
        buf ^= 1;
    }
}

// ---- the same march for the depth head (no BatchNorm, no skip tensor: models/modules.py:489 `prob`), round 4 -------------------------
// What the generic kernel above spends next to its 108 v_pk_fma_f32 per plane and wave (profiles/r3_prob_conv.txt: 178 VALU) is
// mostly bookkeeping of its predicates: the in-range tests of the three halo loads, of the LDS stores and of the output live in
// 64-bit masks, the row block's 72 scalar weights leave no SGPRs for them, and every plane restores them with ~24 v_readlane_b32.
// Here nothing is predicated: halo loads and the logit store go through buffer descriptors (out of range -> zero / dropped), a lane
// with no halo element writes the PAD float4 of a voxel it owns (the row stride of 12 floats has one), planes are float4 arrays
// (ds_write_b128 instead of the conflicting ds_write2_b32 pairs).
// FUSE_D = 8: the last stage of the cascade (D = 8 in ONE chunk): the z loop is unrolled, the eight logits of a pixel stay in
// registers and the softmax / soft-argmin / 4-plane confidence of softmax_regress_kernel (depth_head.hip; same operations in the
// same order: bit-identical) finish in the same thread -- no logit round trip, no second launch; `prob` is optional then.
template <int FUSE_D>
__global__ __launch_bounds__(256, 5) void prob_conv_march_plain_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, float* __restrict__ y, int D, int H, int W, int tiles_w, int tiles_h, int zchunk,
    const float* __restrict__ planes, float* __restrict__ depth, float* __restrict__ conf) {
    typedef pm_f2v f2v;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    __shared__ float4 plane4[2][PM_PLANE / 4];
    constexpr int OOB = 0x7ffffff0;
    const int b = blockIdx.z, zc = blockIdx.y;
    const unsigned t2 = xcd_remap(blockIdx.x, gridDim.x);
    const int tw = t2 % tiles_w, th = t2 / tiles_w;
    const int h0 = th * PM_TH, w0 = tw * PM_TW, z0 = zc * zchunk;
    const int z1 = min(D, z0 + zchunk);                          // outputs z0 .. z1-1, input planes z0-1 .. z1
    const int lw = threadIdx.x % PM_TW, lh = threadIdx.x / PM_TW;
    const int plane_bytes = H * W * 32;
    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (long long)b * D * H * W * 8), (short)0, D * plane_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(y ? y + (long long)b * D * H * W : nullptr, (short)0, y ? D * H * W * 4 : 0, 0x00020000);
    // this thread's share of a plane's halo: element e = (halo voxel, float4 half)
    int goff[PM_NLD], l4[PM_NLD];
#pragma unroll
    for (int i = 0; i < PM_NLD; ++i) {
        const int e = threadIdx.x + i * 256;
        const int v = e >> 1, c4 = e & 1;
        const int hh = v / PM_HW, hw_ = v - hh * PM_HW;
        const int ih = h0 + hh - 1, iw = w0 + hw_ - 1;
        const bool has = e < PM_HH * PM_HW * 2;
        goff[i] = (has && ih >= 0 && ih < H && iw >= 0 && iw < W) ? ((ih * W + iw) * 8 + c4 * 4) * 4 : OOB;
        l4[i] = has ? v * 3 + c4 : (threadIdx.x >> 1) * 3 + 2;    // no element: the pad float4 of a voxel this thread writes anyway
    }
    u32x4 pf[PM_NLD];
    auto fetch = [&](int z) {
        const bool zin = z >= 0 && z < D;
        const int zoff = zin ? z * plane_bytes : 0;
#pragma unroll
        for (int i = 0; i < PM_NLD; ++i) pf[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, zin ? goff[i] : OOB, zoff, 0);
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PM_NLD; ++i) plane4[buf][l4[i]] = __builtin_bit_cast(float4, pf[i]);
    };
    fetch(z0 - 1);
    stash(0);
    fetch(z0);
    __syncthreads();
    f2v acc_prev = (f2v){0.f, 0.f}, acc_cur = (f2v){0.f, 0.f};      // out[z-1] and out[z] while plane z is processed
    const int oh = h0 + lh, ow = w0 + lw;
    const bool live = oh < H && ow < W;
    const int ooff = live ? (oh * W + ow) * 4 : OOB;
    float lg[FUSE_D > 0 ? FUSE_D : 1];
    auto plane_step = [&](int z, int buf) {
        f2v acc_next = (f2v){0.f, 0.f};                             // out[z+1]
        const float* tp0 = reinterpret_cast<const float*>(&plane4[buf][0]) + (lh * PM_HW + lw) * PM_STRIDE;
        f2v xr[2][3][4];
        auto read_row = [&](int kh, f2v (&q)[3][4]) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const float* tp = tp0 + (kh * PM_HW + kw) * PM_STRIDE;
                const float4 xa = *reinterpret_cast<const float4*>(tp), xc = *reinterpret_cast<const float4*>(tp + 4);
                q[kw][0] = (f2v){xa.x, xa.y}; q[kw][1] = (f2v){xa.z, xa.w}; q[kw][2] = (f2v){xc.x, xc.y}; q[kw][3] = (f2v){xc.z, xc.w};
            }
        };
        read_row(0, xr[0]);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            if (kh < 2) read_row(kh + 1, xr[(kh + 1) & 1]);
            pm_row_fma(acc_next, acc_cur, acc_prev, xr[kh & 1], wp + kh * 24);
        }
        const float v = acc_prev.x + acc_prev.y;                    // out[z - 1] is complete now
        acc_prev = acc_cur; acc_cur = acc_next;
        return v;
    };
    if constexpr (FUSE_D == 0) {
        int buf = 0;
        for (int z = z0 - 1; z <= z1; ++z) {
            const float v = plane_step(z, buf);
            const int zo = z - 1;
            if (zo >= z0 && zo < z1) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yrs, ooff, zo * H * W * 4, 0);
            if (z < z1) {
                stash(buf ^ 1);                                     // plane z+1 (fetched during the previous iteration)
                if (z + 2 <= z1) fetch(z + 2);
            }
            __syncthreads();
            buf ^= 1;
        }
    } else {
#pragma unroll
        for (int z = -1; z <= FUSE_D; ++z) {                        // (z0 = 0, z1 = D = FUSE_D)
            const float v = plane_step(z, (z + 1) & 1);
            if (z >= 1) lg[z - 1] = v;
            if (z < FUSE_D) {
                stash(z & 1);
                if (z + 2 <= FUSE_D) fetch(z + 2);
            }
            __syncthreads();
        }
        // softmax over the planes, soft-argmin depth, confidence window (models/casmvsnet.py:293-309; depth_head.hip, LP = 1)
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < FUSE_D; ++k) mx = fmaxf(mx, lg[k]);
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < FUSE_D; ++k) { lg[k] = expf(lg[k] - mx); sum += lg[k]; }
        const long long hw = (long long)H * W;
        const long long pix = live ? (long long)oh * W + ow : 0;
        const float2 pl = reinterpret_cast<const float2*>(planes)[(long long)b * hw + pix];
        float dsum = 0.0f, isum = 0.0f;
#pragma unroll
        for (int k = 0; k < FUSE_D; ++k) {
            lg[k] = lg[k] / sum;
            if (y) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(lg[k]), yrs, ooff, k * H * W * 4, 0);
            dsum = fmaf(lg[k], fmaf((float)k, pl.y, pl.x), dsum);
            isum = fmaf(lg[k], (float)k, isum);
        }
        int idx = (int)isum;                       // .long(): truncation
        idx = idx < 0 ? 0 : (idx > FUSE_D - 1 ? FUSE_D - 1 : idx);
        float c = 0.0f;                            // p[i-1] + p[i] + p[i+1] + p[i+2], zero padded
#pragma unroll
        for (int k = 0; k < FUSE_D; ++k)
            if (k >= idx - 1 && k <= idx + 2) c += lg[k];
        if (live) {
            depth[(long long)b * hw + pix] = dsum;
            conf[(long long)b * hw + pix] = c;
        }
    }
}

// z chunk of the marching prob conv: a block of chunk length c stages c + 2 planes for c output planes; the grid runs in
// ceil(blocks / slots) rounds (slots = CUs x 5 resident blocks).  Minimise rounds x (c + 2), ties to the longer chunk.
static int prob_zchunk(int tiles, int D) {
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

bool conv3d_lds_supported(int Ci, int Co, int stride) {
    return stride == 1 && ((Co == 8 && (Ci == 8 || Ci == 16 || Ci == 32 || Ci == 44)) || (Co == 1 && Ci == 8) ||
                           (Co == 16 && Ci == 16));
}

// g_lds_cfg: test / tuning override passed by the caller (0 = tuned default; bit0 = 16-channel chunks, bit1 = force split,
// bit2 = force no split, bit3 = one voxel per thread)
int conv3d_lds_launch(const float* x, const float* wp, const float* scale, const float* shift, const float* res, float* y,
                      int B, int D, int H, int W, int Ci, int Co, int relu, hipStream_t st, int g_lds_cfg) {
    const int ppt = (Co == 8 && Ci >= 32 && !(g_lds_cfg & 8)) ? 2 : 1;   // two voxels per thread pay at Cin >= 32 (298 -> 250 us; 16 -> 8: 235 vs 243)
    const int tiles_w = (W + LT_W - 1) / LT_W, tiles_h = (H + LT_H - 1) / LT_H, tiles_d = (D + LT_D * ppt - 1) / (LT_D * ppt);
    // tuned on MI355X (tools/conv_bench.py): 8-channel chunks keep the per-pass weight set (6.9 KB) in the scalar
    // cache; splitting Cout over two wave groups pays only when there are >= 4 chunk passes (Cin >= 32)
    const int ckt = (g_lds_cfg & 1) ? 16 : 8;
    const int split = (g_lds_cfg & 2) ? 2 : ((g_lds_cfg & 4) ? 1 : ((Ci >= 32 || Co >= 16) ? 2 : 1));
    dim3 grid(tiles_w * tiles_h, tiles_d, B), block(256 * split);
    const int ck = Ci < ckt ? Ci : ckt;
    const size_t lds = (size_t)(LT_D * ppt + 2) * LH_H * LH_W * (ck + 4) * sizeof(float);
#define RCMVS_LDS_LAUNCH(CI, SP, CK) do { \
        if (ppt == 2) { if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)conv3d_lds_kernel<CI, 8, SP, CK, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                        hipLaunchKernelGGL((conv3d_lds_kernel<CI, 8, SP, CK, 2>), grid, block, lds, st, x, wp, scale, shift, res, y, D, H, W, tiles_w, tiles_h, relu); } \
        else hipLaunchKernelGGL((conv3d_lds_kernel<CI, 8, SP, CK>), grid, block, lds, st, x, wp, scale, shift, res, y, D, H, W, tiles_w, tiles_h, relu); } while (0)
#define RCMVS_LDS_CASE(CI)                                                                  \
    if (Ci == CI) {                                                                         \
        if (split == 2) { if (ckt == 8) RCMVS_LDS_LAUNCH(CI, 2, 8); else RCMVS_LDS_LAUNCH(CI, 2, 16); } \
        else            { if (ckt == 8) RCMVS_LDS_LAUNCH(CI, 1, 8); else RCMVS_LDS_LAUNCH(CI, 1, 16); } \
        return launch_status("conv3d_lds");                                                 \
    }
    if (Co == 16 && Ci == 16) {                                 // conv2 of the U-Nets: 16 accumulators, 64 scalar weights per (tap, 4 ch)
        if (!(g_lds_cfg & 8)) {                                 // two voxels per thread (bit 5 of the rcmvs_debug_conv3d_fwd impl: one)
            const int td2 = (D + 2 * LT_D - 1) / (2 * LT_D);
            const size_t lds2 = (size_t)(2 * LT_D + 2) * LH_H * LH_W * 12 * sizeof(float);
            dim3 grid2(tiles_w * tiles_h, td2, B);
            hipLaunchKernelGGL((conv3d_lds_kernel<16, 16, 2, 8, 2>), grid2, dim3(512), lds2, st, x, wp, scale, shift, res, y, D, H, W, tiles_w, tiles_h, relu);
            return launch_status("conv3d_lds(16x16, 2 voxels)");
        }
        if (split == 2) hipLaunchKernelGGL((conv3d_lds_kernel<16, 16, 2, 8>), grid, block, lds, st, x, wp, scale, shift, res, y, D, H, W, tiles_w, tiles_h, relu);
        else            hipLaunchKernelGGL((conv3d_lds_kernel<16, 16, 1, 8>), grid, block, lds, st, x, wp, scale, shift, res, y, D, H, W, tiles_w, tiles_h, relu);
        return launch_status("conv3d_lds(16x16)");
    }
    if (Co == 1 && Ci == 8 && !(g_lds_cfg & 1)) {               // prob conv 8 -> 1 (logits, (B,D,H,W) since Co = 1): plane-marching kernel
        const int tw_ = (W + PM_TW - 1) / PM_TW, th_ = (H + PM_TH - 1) / PM_TH;
        const int zc_force = (g_lds_cfg >> 8) & 0xff;           // test / tuning override of the z chunk
        const int zc = zc_force ? (zc_force < D ? zc_force : D) : prob_zchunk(B * tw_ * th_, D);
        if (!scale && !shift && !res && !relu && !(g_lds_cfg & 16) && (long long)D * H * W * 32 < 0x7ffffff0LL) {      // the depth head's form (bit 4 of lds_cfg: the generic kernel)
            hipLaunchKernelGGL(prob_conv_march_plain_kernel<0>, dim3(tw_ * th_, (D + zc - 1) / zc, B), dim3(256), 0, st, x, wp, y, D, H, W, tw_, th_, zc,
                               (const float*)nullptr, (float*)nullptr, (float*)nullptr);
            return launch_status("conv3d_lds(prob, marching, plain)");
        }
        hipLaunchKernelGGL(prob_conv_march_kernel, dim3(tw_ * th_, (D + zc - 1) / zc, B), dim3(256), 0, st, x, wp, scale, shift, res, y,
                           D, H, W, tw_, th_, relu, zc);
        return launch_status("conv3d_lds(prob, marching)");
    }
    if (Co == 1 && Ci == 8) {                                   // the tile kernel form (bit 0 of lds_cfg: A/B and cross-check)
        dim3 block1(256);
        hipLaunchKernelGGL((conv3d_lds_kernel<8, 1, 1, 8>), grid, block1, (size_t)LH_VOX * 12 * sizeof(float), st, x, wp, scale, shift,
                           res, y, D, H, W, tiles_w, tiles_h, relu);
        return launch_status("conv3d_lds(prob)");
    }
    RCMVS_LDS_CASE(8) RCMVS_LDS_CASE(16) RCMVS_LDS_CASE(32) RCMVS_LDS_CASE(44)
#undef RCMVS_LDS_CASE
#undef RCMVS_LDS_LAUNCH
    (void)Co;
    return fail(-1, "conv3d_lds: unsupported Ci=%d", Ci);
}

// depth head with D = 8 in one launch: prob conv + softmax + soft-argmin + confidence (prob_conv_march_plain_kernel<8>); prob may be null
bool prob_head_fused_supported(int D, int H, int W) { return D == 8 && (long long)D * H * W * 32 < 0x7ffffff0LL; }
int prob_head_fused_launch(const float* x, const float* wp, const float* planes, float* depth, float* conf, float* prob,
                           int B, int D, int H, int W, hipStream_t st) {
    if (!prob_head_fused_supported(D, H, W)) return fail(-1, "prob_head_fused: D = %d is not the fused form", D);
    const int tw_ = (W + PM_TW - 1) / PM_TW, th_ = (H + PM_TH - 1) / PM_TH;
    hipLaunchKernelGGL(prob_conv_march_plain_kernel<8>, dim3(tw_ * th_, 1, B), dim3(256), 0, st, x, wp, prob, D, H, W, tw_, th_, D, planes, depth, conf);
    return launch_status("depth_head(fused, D = 8)");
}

}  // namespace rcmvs
