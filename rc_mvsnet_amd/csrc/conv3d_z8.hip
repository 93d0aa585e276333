// conv0 of the cost regularisation (3x3x3, stride 1, Cin = 8 / 16 -> Cout = 8; models/modules.py:472) for a B = 1 inference scene, fp16-pair
// arithmetic (conv3d_x3.hip, NP = 2: the SAME weight image and fragment maps) -- without the producer / consumer split.  gfx950 only.
//
// Why.  The tick trace of the split kernel (profiles/r3_x3_tick_trace.txt) reads: consumer MFMAs end at 1 374 of a
// 3 593-clock tick; the producer wave that shares the SIMD then needs ~2 000 clocks for 45 VALU instructions, 6 LDS stores and 12 load
// requests -- its work is tiny, but VALU work does not overlap MFMAs issued on the same SIMD and a lone wave has nothing to hide its own
// latencies behind: a tick is MFMA time PLUS a latency chain.  Here all eight waves are alike: each owns one n-tile (16 columns x 2 rows) of
// an 8 x 32 output tile (Cin = 8: two n-tiles of 32 columns x 1 row of a 16 x 32 tile) and the whole K range (weights register-stationary: 9 / 18 k-steps x two
// pieces), each stages its share of the next input plane (split into fp16 pieces on the way into a two-slot LDS ring), one barrier per
// input plane.  A SIMD hosts two such waves: while one waits for LDS or memory the other issues MFMAs.
// Work is a STREAM of planes: a block takes a contiguous range of the flattened (tile, z) steps; an item (tile, z range of n planes) needs
// the n + 2 input planes around it -- minus the planes outside the volume (round 6: the zero planes z = -1 and z = D are never fed; the tick
// that feeds plane D - 1 stores the last TWO output planes), so a whole tile costs D ticks and only a cut inside a tile costs two extra
// (the ring, the register queue and the loads run on across item boundaries: no latency is exposed when the block moves to its next tile).
// Tick t: park stream plane t + 1 (requested two ticks earlier), request stream plane t + 3, feed
// stream plane t to the three rolling accumulators (kd = 0, 1, 2: output planes t, t - 1, t - 2 of the item) and store the one it completes.
// Measured on the way (16 -> 8 at 32 x 256 x 320; profiles/r4_z8.txt): three planes read per output plane instead of rolling accumulators
// 74.3 us (LDS bandwidth is not the bound); a cursor object instead of the closed-form stream plan 72.2; with every memory, LDS and MFMA
// instruction switched off 36.9 -- the skeleton of a tick (split arithmetic, epilogue, barrier) is half of it.
// Cin = 32 -> 8 (stage 1): 36 k-steps x two pieces are 288 registers, so the LOW pieces of the weights live in LDS (36 KB, one ds_read_b128
// per use), a plane's loads fly one tick instead of two and the B fragments are not read ahead: 89 -> 77 us with 15 spilled registers left.
// (A first form with wave = (n-tile, K half), 4 x 32 tiles and the halves meeting through LDS one tick later measured 96 - 100 us: removed.)
#include "common.h"
#include "x3_pieces.h"
#include <atomic>
#include <cstdlib>

namespace rcmvs {

template <int CIN, int COUT>
struct Z8 {
    // fragment maps of the x3 image: Cout = 8 fills the 16 rows of an m-tile with two output positions -- M = (shift along x, co) for Cin = 8
    // ("XT": an n-tile is 32 columns of one row), (shift along y, co) for Cin = 16 ("YT": 16 columns of two rows); Cout = 16: M = co ("PL": 16 columns of one row)
    static constexpr bool XT = COUT == 8 && CIN == 8, YT = COUT == 8 && CIN >= 16, PL = COUT == 16;
    static constexpr bool ALO = CIN == 32;           // Cin = 32: 36 k-steps x two pieces are 288 registers -- the low pieces of the weights live in LDS (36 KB, read per use)
#ifndef Z8_PL_NTW
#define Z8_PL_NTW 1         // (conv2, us per launch at stages 1 / 2 / 3: 18.4 / 26.2 / 25.7 with one n-tile per wave = 4 x 32 tiles, 20.4 / 27.5 / 28.7 with two)
#endif
    static constexpr int NTW = YT ? 1 : (PL ? Z8_PL_NTW : 2);           // n-tiles per wave (XT with one: 52.2 against 47.3 us at stage 3) (two where the weights leave the registers: a tick of 27 - 45 MFMAs per wave is mostly barrier)
    static constexpr int NTT = 8 * NTW;              // n-tiles per tick
    static constexpr int TY = PL ? NTT / 2 : NTT, TX = 32;      // XT: a row per n-tile; YT: row pairs, two side by side; PL: rows, two side by side
    static constexpr int RSTEP = PL ? 4 : 8;         // tile rows between the two n-tiles of a wave
    static constexpr int VB = CIN * 2;               // bytes per voxel per piece plane
    static constexpr int Q4 = CIN / 4;               // float4 per voxel
    static constexpr int PPS = 32 / CIN;             // tap positions per k-step
    static constexpr int QR = YT ? 4 : 3, QC = XT ? 4 : 3, PPKD = QR * QC;
    static constexpr int SPK = (PPKD + PPS - 1) / PPS;      // k-steps per plane
    static constexpr int KSTEPS = 3 * SPK;
    static constexpr int TYP = TY + 2, TXP = TX + 2;
    static constexpr int ROWB = TXP * VB, PLB = TYP * ROWB, SLB = 2 * PLB;
    static constexpr int NSLOT = 2;                  // the plane being read + the one being parked
    static constexpr int NLD = (TYP * TXP * Q4 + 511) / 512;
    static constexpr int CS = XT ? 2 : 1;
    static constexpr int ALOB = ALO ? KSTEPS * 1024 : 0;
    static constexpr int LDS = NSLOT * SLB + ALOB + 64;
    static_assert((COUT == 8 && (CIN == 8 || CIN == 16 || CIN == 32)) || (COUT == 16 && CIN == 16), "conv0 of stages 3, 2 and 1, conv2");
    static_assert(LDS <= 160 * 1024, "LDS budget");
};

// voxel-internal byte swizzle of the Cin = 32 layout (64-byte voxels: lanes n, n + 4 of a ds_read_b128 lane group would collide; conv3d_x3.hip)
template <int CIN> __device__ __forceinline__ int z8_swz(int hc) { return CIN == 32 ? ((hc >> 2) & 1) * 32 : 0; }

struct Z8Dims {
    int B, D, H, W;
    int tiles_x, ntiles;          // tile grid of one batch element
    int relu;
};

// The block's stream in closed form: its steps [lo, hi) of the flattened (tile, z) sequence fall into items -- a first one (tile0, planes zb0 ..
// zb0 + nz0 - 1), full tiles, a tail.  Item-local input plane ip stands for input plane zb - 1 + ip; an item feeds ip = a0 .. a1 - 1 with
// a0 = 1 when it starts at the volume's first plane (zb = 0) and a1 = nz + 1 when it ends at its last (zb + nz = D), else 0 / nz + 2: a full
// tile is D stream planes (ip = 1 .. D), a tail nz + 1 (ip = 1 .. nz + 1).  Stream position s -> (tile, ip, zb, nz) costs a dozen scalar
// instructions (a cursor object with its 64-bit divisions inlined four times per loop body was most of a tick's 700 instructions: 37 of
// 75 us with every memory, LDS and MFMA instruction switched off).
struct Z8Plan { int nticks, L0, a00, tile0, zb0, nz0, last, nz_last, D; unsigned inv; };
__device__ __forceinline__ bool z8_entry(const Z8Plan& p, int s, int& tile, int& ip, int& zb, int& nz) {
    if (s >= p.nticks) return false;
    if (s < p.L0) { tile = p.tile0; ip = s + p.a00; zb = p.zb0; nz = p.nz0; return true; }
    const int x = s - p.L0;
    const int k1 = p.D == 1 ? x : (int)__umulhi((unsigned)x, p.inv);      // x / D, exact for x < 2^16 (host-checked; D = 1 has no 32-bit reciprocal)
    tile = p.tile0 + 1 + k1; ip = x - k1 * p.D + 1; zb = 0;
    nz = (1 + k1 == p.last) ? p.nz_last : p.D;
    return true;
}

template <int CIN, int COUT>
__global__ __launch_bounds__(512) void conv3d_z8_kernel(
    const float* __restrict__ x, const x3_u32x4* __restrict__ wimg, const float* __restrict__ scale, const float* __restrict__ shift,
    float* __restrict__ y, Z8Dims dm, const float* __restrict__ xmax, float* __restrict__ ymax) {
    using C = Z8<CIN, COUT>;
    constexpr int KSTEPS = C::KSTEPS, NLD = C::NLD, OOB = 0x7ffffff0;
    extern __shared__ __attribute__((aligned(16))) x3_byte smem[];
    x3_byte* const alo = smem + C::NSLOT * C::SLB;              // (Cin = 32) low pieces of the weight fragments, [k-step][lane][16 B]
    float* const redmax = reinterpret_cast<float*>(alo + C::ALOB);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kk = lane >> 4;
    // ---- the block's share of the step sequence (neighbouring ranges on one XCD: block ids b, b + 8, ... share an L2)
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int r = ((nblk & 7) == 0) ? (bid & 7) * (nblk >> 3) + (bid >> 3) : bid;
    const long long T = (long long)dm.B * dm.ntiles * dm.D;
    const long long lo = T * r / nblk, hi = T * (r + 1) / nblk;
    if (hi <= lo) return;
    Z8Plan pl;
    {
        const int S = (int)(hi - lo);
        pl.tile0 = (int)(lo / dm.D); pl.zb0 = (int)(lo % dm.D);
        pl.nz0 = min(dm.D - pl.zb0, S);
        const int R = S - pl.nz0, full = R / dm.D, tail = R % dm.D;
        const int nitems = 1 + full + (tail > 0);
        pl.last = nitems - 1; pl.nz_last = tail > 0 ? tail : dm.D;
        pl.a00 = pl.zb0 == 0 ? 1 : 0;
        pl.L0 = pl.nz0 + 2 - pl.a00 - (pl.zb0 + pl.nz0 == dm.D ? 1 : 0); pl.D = dm.D;
        pl.inv = (unsigned)(0x100000000ull / (unsigned)dm.D) + 1u;
        pl.nticks = pl.L0 + full * dm.D + (tail > 0 ? tail + 1 : 0);
    }

    // ---- scales, weights (register-stationary), this lane's B-fragment offsets inside a slice
    float bound = xmax[lane * 16];
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) bound = fmaxf(bound, __shfl_xor(bound, m));
    float xinv;
    const float xs_scale = x3_pow2_scale(bound, xinv);
    const float unscale = xinv * reinterpret_cast<const float*>(wimg)[1];
    x3_u32x4 wr[KSTEPS][C::ALO ? 1 : 2];
    int boff[C::SPK];              // (the same for the three kd weight sets of a plane)
    const int toff = C::XT ? wave * C::ROWB : (wave >> 1) * (C::YT ? 2 : 1) * C::ROWB + (wave & 1) * 16 * C::VB;      // this wave's (first) n-tile
#pragma unroll
    for (int j = 0; j < KSTEPS; ++j) {
#pragma unroll
        for (int p = 0; p < (C::ALO ? 1 : 2); ++p) wr[j][p] = wimg[1 + (j * 2 + p) * 64 + lane];
        if (C::ALO && wave == (j & 7)) *reinterpret_cast<x3_u32x4*>(alo + (j * 64 + lane) * 16) = wimg[1 + (j * 2 + 1) * 64 + lane];      // (visible after the barrier in front of the first tick)
    }
#pragma unroll
    for (int js = 0; js < C::SPK; ++js) {
        int q = js * C::PPS + kk / (4 / C::PPS);
        const int ci0 = (kk % (4 / C::PPS)) * 8;
        if (q >= C::PPKD) q = 0;                       // (padding slot of the last k-step of a plane: its weights are zero)
        const int hc = q % C::QC + n * C::CS;          // halo column of this lane's voxel (tile column offsets are multiples of 16: same swizzle)
        boff[js] = toff + (q / C::QC) * C::ROWB + hc * C::VB + ((ci0 * 2) ^ z8_swz<CIN>(hc));
    }
    const int co0 = C::PL ? kk * 4 : (kk & 1) * 4;
    const x3_f32x4 sc = (scale ? *reinterpret_cast<const x3_f32x4*>(scale + co0) : (x3_f32x4){1.f, 1.f, 1.f, 1.f}) * unscale;
    const x3_f32x4 sh = shift ? *reinterpret_cast<const x3_f32x4*>(shift + co0) : (x3_f32x4){0.f, 0.f, 0.f, 0.f};
    // this lane's output voxel inside the tile
    const int oyl = C::XT ? wave : (C::YT ? 2 * (wave >> 1) + (kk >> 1) : (wave >> 1)), oxl = C::XT ? 2 * n + (kk >> 1) : (wave & 1) * 16 + n;      // (n-tile i of the wave: RSTEP i rows further down)

    // ---- staging shares of a plane: element e = (halo voxel, float4 of its channels)
    int loff[NLD], hyx[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        // (a thread whose last share lies past the end of the plane repeats its previous one -- the same load, the same pieces to the same LDS
        // address -- so that stash() is straight-line code: 1 - 2 us per launch against exec-masked stores)
        static_assert(NLD * 512 - C::TYP * C::TXP * C::Q4 <= 512 && C::TYP * C::TXP * C::Q4 >= 512, "the repeated share exists");
        const int e0 = tid + i * 512, e = e0 >= C::TYP * C::TXP * C::Q4 ? e0 - 512 : e0, v = e / C::Q4, c4 = e % C::Q4;
        const int hy = v / C::TXP, hx = v % C::TXP;
        loff[i] = hy * C::ROWB + hx * C::VB + ((c4 * 8) ^ z8_swz<CIN>(hx));
        hyx[i] = (hy << 20) | (hx << 8) | c4;
    }
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), (short)0, (int)((long long)dm.B * dm.D * dm.H * dm.W * CIN * 4), 0x00020000);
    const int zstride = dm.H * dm.W * CIN * 4;
    __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(y, (short)0, (int)((long long)dm.B * dm.D * dm.H * dm.W * COUT * 4), 0x00020000);

    // fetch side: request stream plane sf (zeros past the end of the stream)
    int goff[NLD];
    int f_tile = -1;
    auto fetch = [&](x3_f32x4 (&q)[NLD], int sf) {
        int tile, ip, zb, nz;
        const bool valid = z8_entry(pl, sf, tile, ip, zb, nz);
        bool zin = false;
        int zoff = 0;
        if (valid) {
            if (tile != f_tile) {
                f_tile = tile;
                const int b = tile / dm.ntiles, t = tile % dm.ntiles;
                const int y0 = (t / dm.tiles_x) * C::TY, x0 = (t % dm.tiles_x) * C::TX;
                const int base = (((b * dm.D) * dm.H + (y0 - 1)) * dm.W + (x0 - 1)) * CIN * 4;
#pragma unroll
                for (int i = 0; i < NLD; ++i) {
                    const int hy = hyx[i] >> 20, hx = (hyx[i] >> 8) & 0xfff, c4 = hyx[i] & 0xff;
                    const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
                    goff[i] = (gy >= 0 && gy < dm.H && gx >= 0 && gx < dm.W) ? base + ((hy * dm.W + hx) * CIN + c4 * 4) * 4 : OOB;
                }
            }
            const int z = zb - 1 + ip;
            zin = z >= 0 && z < dm.D;
            zoff = zin ? z * zstride : 0;
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) q[i] = __builtin_bit_cast(x3_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, zin ? goff[i] : OOB, zoff, 0));
    };
    auto stash = [&](const x3_f32x4 (&q)[NLD], int slot) {
        x3_byte* sb = smem + slot * C::SLB;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            x3_u32x2 h, l;
            x3_split4h(q[i] * xs_scale, h, l);
            *reinterpret_cast<x3_u32x2*>(sb + loff[i]) = h;
            *reinterpret_cast<x3_u32x2*>(sb + C::PLB + loff[i]) = l;
        }
    };

    // compute side
    int ob[C::NTW];               // byte offset of this lane's float4 of the current output plane, per n-tile of the wave
#pragma unroll
    for (int i = 0; i < C::NTW; ++i) ob[i] = OOB;
    const int ostep = dm.H * dm.W * COUT * 4;
    float vmax = 0.0f;
    constexpr int QD = C::ALO ? 1 : 2;                   // ticks of flight of a plane's loads (Cin = 32: one register set is all there is room for)
    x3_f32x4 pq[QD][NLD];
    // stream planes 0 and 1 now, plane 0 parked before the first tick, plane 2 requested
    fetch(pq[0], 0);
    if constexpr (QD == 2) fetch(pq[1], 1);
    stash(pq[0], 0);
    fetch(pq[0], QD);
    __syncthreads();
    // Rolling accumulators (as in the depth head's marching conv): the fragments of input plane i are read from LDS ONCE and feed all three
    // kd weight sets -- acc[.][kd] holds output plane i - kd of the item; after plane i output i - 2 is complete, the sets move up.
    // A third of the LDS reads of "three planes per output plane" (LDS bandwidth, not the matrix pipe, bounded that form: 36 KB per
    // n-tile and output plane against 48 MFMA clocks per 2 KB), and the ring is two slots.  One fp32 accumulator per output plane
    // (hh, hl and lh products alike): consecutive MFMAs go to different sets, none waits for the one before it.
    x3_f32x4 acc[C::NTW][3];
    auto tick = [&](int t, x3_f32x4 (&q)[NLD]) {          // q = the register set of stream plane t + 1
        stash(q, (t + 1) & 1);
        fetch(q, t + 1 + QD);
        int ctile, ip, czb, cnz;                           // ip = plane of the item (0 .. nz + 1): input plane zb - 1 + ip
        z8_entry(pl, t, ctile, ip, czb, cnz);              // (t < nticks: the loop's own bound)
        {
            if (ip == (czb == 0 ? 1 : 0)) {                // first plane of the item: empty accumulators, where this lane's voxels of output plane zb go
#pragma unroll
                for (int i = 0; i < C::NTW; ++i)
#pragma unroll
                    for (int c = 0; c < 3; ++c) acc[i][c] = (x3_f32x4){0.f, 0.f, 0.f, 0.f};
                const int b = ctile / dm.ntiles, tl = ctile % dm.ntiles;
#pragma unroll
                for (int i = 0; i < C::NTW; ++i) {
                    const int oy = (tl / dm.tiles_x) * C::TY + oyl + C::RSTEP * i, ox = (tl % dm.tiles_x) * C::TX + oxl;
                    ob[i] = (oy < dm.H && ox < dm.W) ? (((((b * dm.D) + czb) * dm.H + oy) * dm.W + ox) * COUT + co0) * 4 : OOB;
                }
            }
            // (an input plane feeds output planes ip - kd; those outside 0 .. nz - 1 are computed too and never stored: no branches in the MFMA stream)
            const x3_byte* sb = smem + (t & 1) * C::SLB;
            constexpr int NBQ = C::ALO ? 1 : 2;            // (Cin = 32: no registers for reading a k-step ahead)
            x3_u32x4 bq[NBQ][C::NTW][2];
            auto read_b = [&](int buf, int js) {
#pragma unroll
                for (int i = 0; i < C::NTW; ++i) {
                    const x3_byte* pb = sb + boff[js] + i * C::RSTEP * C::ROWB;
                    bq[buf][i][0] = *reinterpret_cast<const x3_u32x4*>(pb);
                    bq[buf][i][1] = *reinterpret_cast<const x3_u32x4*>(pb + C::PLB);
                }
            };
            if (NBQ == 2) read_b(0, 0);
#pragma unroll
            for (int js = 0; js < C::SPK; ++js) {
                if (NBQ == 1) read_b(0, js);
                else if (js + 1 < C::SPK) read_b((js + 1) & 1, js + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int p = 0; p < 3; ++p) {              // product hh, hl, lh
#pragma unroll
                    for (int kd = 0; kd < 3; ++kd) {
                        const int j = kd * C::SPK + js;
                        x3_u32x4 a;
                        if constexpr (C::ALO) a = p == 2 ? *reinterpret_cast<const x3_u32x4*>(alo + (j * 64 + lane) * 16) : wr[j][0];
                        else a = wr[j][p == 2 ? 1 : 0];
#pragma unroll
                        for (int i = 0; i < C::NTW; ++i)
                            acc[i][kd] = x3_mfma<2>(a, bq[js & (NBQ - 1)][i][p == 1 ? 1 : 0], acc[i][kd]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ip >= 2) {                                 // output plane ip - 2 is complete
#pragma unroll
                for (int i = 0; i < C::NTW; ++i) {
                    x3_f32x4 v = acc[i][2] * sc + sh;
                    if (dm.relu) v = __builtin_elementwise_max(v, (x3_f32x4){0.f, 0.f, 0.f, 0.f});
                    // (a buffer store with an out-of-range offset for lanes outside the volume: a fixed number of vector-memory operations per
                    // tick, so the compiler's s_waitcnt bookkeeping can wait for the two-tick-old plane with a partial count)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(x3_u32x4, v), yrs, ob[i], 0, 0);
                    const bool in = ob[i] != OOB;
                    vmax = in ? x3_absmax4(vmax, v) : vmax;
                    ob[i] = in ? ob[i] + ostep : OOB;
                }
            }
            if (ip == cnz && czb + cnz == dm.D) {          // the item ends at the volume's last plane: plane D (zeros) is not fed, output plane D - 1 is complete as well
#pragma unroll
                for (int i = 0; i < C::NTW; ++i) {
                    x3_f32x4 v = acc[i][1] * sc + sh;
                    if (dm.relu) v = __builtin_elementwise_max(v, (x3_f32x4){0.f, 0.f, 0.f, 0.f});
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(x3_u32x4, v), yrs, ob[i], 0, 0);
                    vmax = ob[i] != OOB ? x3_absmax4(vmax, v) : vmax;
                }
            }
#pragma unroll
            for (int i = 0; i < C::NTW; ++i) { acc[i][2] = acc[i][1]; acc[i][1] = acc[i][0]; acc[i][0] = (x3_f32x4){0.f, 0.f, 0.f, 0.f}; }
        }
        __syncthreads();
    };
    for (int t = 0; t < pl.nticks; t += 2) {
        tick(t, pq[QD - 1]);
        if (t + 1 < pl.nticks) tick(t + 1, pq[0]);
    }
    if (ymax) {
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, m));
        if (lane == 0) redmax[wave] = vmax;
        __syncthreads();
        if (tid == 0) {
            float m = redmax[0];
#pragma unroll
            for (int i = 1; i < 8; ++i) m = fmaxf(m, redmax[i]);
            atomicMax(reinterpret_cast<unsigned int*>(ymax) + (blockIdx.x & 63) * 16, __float_as_uint(m));
        }
    }
}

bool conv3d_z8_supported(int Ci, int Co, int kind) {
    // conv2 (16 -> 16) runs here too since the planes outside the volume are no longer fed (same-box, us per launch at stages 1 / 2 / 3: 20.3 / 27.4 / 28.0
    // against 20.0 / 30.0 / 31.6 on the split kernel); RCMVS_Z8_CONV2=0 sends it back (A/B, and the test of the split kernel's 16 -> 16 pair form)
    static const bool conv2 = [] { const char* e = getenv("RCMVS_Z8_CONV2"); return !e || e[0] != '0'; }();
    return kind == 0 && ((Co == 8 && (Ci == 8 || Ci == 16 || Ci == 32)) || (conv2 && Ci == 16 && Co == 16));
}

template <int CIN, int COUT>
static int z8_launch_t(const float* x, const float* wimg, const float* scale, const float* shift, float* y, const Z8Dims& dm, int n_cu, int dev,
                       const float* xmax, float* ymax, hipStream_t st) {
    using C = Z8<CIN, COUT>;
    constexpr int MAXDEV = 64;
    static std::atomic<bool> raised[MAXDEV];
    if (C::LDS > 64 * 1024 && !raised[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)conv3d_z8_kernel<CIN, COUT>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS) != hipSuccess)
            return fail(-1, "conv3d_z8: cannot raise the dynamic LDS limit to %d bytes", C::LDS);
        raised[dev].store(true, std::memory_order_release);
    }
    const long long T = (long long)dm.B * dm.ntiles * dm.D;
    const int blocks = (int)(T < n_cu ? T : n_cu);
    hipLaunchKernelGGL((conv3d_z8_kernel<CIN, COUT>), dim3(blocks), dim3(512), C::LDS, st, x, reinterpret_cast<const x3_u32x4*>(wimg), scale, shift, y, dm, xmax, ymax);
    return launch_status("conv3d_z8");
}

// x (B, D, H, W, Ci) -> y (B, D, H, W, Co); wimg = the x3h image of the pair (conv3d_x3h_pack); xmax required, ymax optional
int conv3d_z8_launch(const float* x, const float* wimg, const float* scale, const float* shift, float* y,
                     int B, int D, int H, int W, int Ci, int Co, int relu, hipStream_t st, int max_blocks, const float* xmax, float* ymax) {
    if (!xmax) return fail(-1, "conv3d_z8: the fp16-pair form needs a bound of max|x|");
    if ((long long)B * D * H * W * (Ci > Co ? Ci : Co) * 4 >= 0x7ffffff0LL) return fail(-1, "conv3d_z8: tensor too large for 32-bit offsets");
    constexpr int MAXDEV = 64;
    static std::atomic<int> cu_of[MAXDEV];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return fail(-1, "conv3d_z8: cannot query the device");
    if (cu_of[dev] == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return fail(-1, "conv3d_z8: cannot query the device");
        cu_of[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    Z8Dims dm;
    dm.B = B; dm.D = D; dm.H = H; dm.W = W; dm.relu = relu;
    const int ty = Co == 16 ? Z8<16, 16>::TY : (Ci == 8 ? Z8<8, 8>::TY : Z8<16, 8>::TY);      // (32 -> 8: as 16 -> 8)
    dm.tiles_x = (W + 31) / 32;
    dm.ntiles = dm.tiles_x * ((H + ty - 1) / ty);
    const int n_blk = max_blocks > 0 ? max_blocks : cu_of[dev].load();
    if (((long long)B * dm.ntiles * D + n_blk - 1) / n_blk + 2 >= 65536)
        return 1;        // too many steps per block for the 16-bit stream arithmetic: not taken (conv3d_x3_launch goes on to the split kernel)
    if (Ci == 8 && Co == 8) return z8_launch_t<8, 8>(x, wimg, scale, shift, y, dm, n_blk, dev, xmax, ymax, st);
    if (Ci == 16 && Co == 8) return z8_launch_t<16, 8>(x, wimg, scale, shift, y, dm, n_blk, dev, xmax, ymax, st);
    if (Ci == 32 && Co == 8) return z8_launch_t<32, 8>(x, wimg, scale, shift, y, dm, n_blk, dev, xmax, ymax, st);
    if (Ci == 16 && Co == 16) return z8_launch_t<16, 16>(x, wimg, scale, shift, y, dm, n_blk, dev, xmax, ymax, st);
    return fail(-1, "conv3d_z8: unsupported Ci=%d Co=%d", Ci, Co);
}

}  // namespace rcmvs
