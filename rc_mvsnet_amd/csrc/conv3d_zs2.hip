// conv1 of the cost regularisation (3x3x3, stride 2, 8 -> 16; models/modules.py:473) for a B = 1 inference scene on the z-streaming scheme of conv3d_z8.hip:
// fp16-pair arithmetic (the x3h weight image and fragment map of conv3d_x3.hip's stride-2 kind, NP = 2), eight alike waves, no producer / consumer
// split.  gfx950 only.  Round 6.
//
// Why.  On the split kernel this layer runs 21 / 34 / 34 us per stage for 39 / 105 / 105 MB (3 TB/s): its tick trace (profiles/r6_session2.txt) shows the
// consumers done after 1 500 of 2 790 clocks and the four producer waves parking, requesting and finishing until 2 630 -- and with every memory, LDS and
// MFMA instruction switched off the launch still takes 22.7 of 30.2 us: the tick skeleton of 64-voxel output tiles is the cost, not the work.
// Here a tick is one OUTPUT plane of an 8 x 16 tile = the input-plane pair (2 z, 2 z + 1) of its 17 x 33 halo (36 KB per tick and block):
//   * every wave owns one output row (one n-tile of 16 columns) and the whole K range: M = output channel, K step = four tap positions x 8 channels
//     (three steps per plane, the last one three quarters padding), weights register-stationary (9 K steps x two pieces = 72 registers);
//   * rolling accumulators (two magnitude classes each, as on the split kernel: hh / hl + lh): plane 2 z feeds output z (kd = 1), plane 2 z + 1 feeds output z (kd = 2) and output z + 1 (kd = 0); after the pair output z is
//     complete and stored.  An item that starts inside a tile (z_b > 0) opens with a lead tick for plane 2 z_b - 1 alone (nothing stored); planes
//     outside the volume are never requested: a whole tile costs Do ticks;
//   * a thread stages 2 x 3 float4 of the next pair (split into fp16 pieces on the way into a four-plane LDS ring; requested one tick earlier -- a
//     tick is microseconds here), one barrier per tick.
// Work is the flattened (tile, output plane) sequence cut into one contiguous range per block, as in conv3d_z8.hip.
#include "common.h"
#include "x3_pieces.h"
#include <atomic>
#include <cstdlib>

namespace rcmvs {

namespace zs2 {
constexpr int CIN = 8, COUT = 16;
#ifndef ZS2_TX
#define ZS2_TX 16           // (32: 19.7 / 29.2 / 28.8 us per stage against 16.9 / 28.0 / 26.5 -- more, smaller ticks fill the pipeline of a block that has five to ten of them)
#endif
constexpr int TY = 8, TX = ZS2_TX, NTW = TX / 16;         // output tile; n-tiles (16 columns of one row) per wave
constexpr int HY = 2 * TY + 1, HX = 2 * TX + 1;           // input halo of one plane
constexpr int VB = CIN * 2;                               // bytes per voxel per piece plane
constexpr int ROWB = HX * VB, PLB = HY * ROWB, SLB = 2 * PLB;      // a plane = its high-piece plane + its low-piece plane
constexpr int NSLOT = 4;                                  // the pair being read + the pair being parked
constexpr int Q4 = CIN / 4;
constexpr int NE = HY * HX * Q4, NLD = (NE + 511) / 512;  // float4 per plane, per thread
constexpr int SPK = 3, KSTEPS = 9;                        // K steps per plane (positions 0 .. 8 four at a time), per tap cube
constexpr int LDS = NSLOT * SLB + 64;
static_assert(NLD * 512 - NE <= 512 && NE >= 512, "the repeated share exists");
static_assert(LDS <= 160 * 1024, "LDS budget");
}

struct ZS2Dims {
    int B, D, H, W, Do, Ho, Wo;
    int tiles_x, ntiles;          // tile grid of one batch element (output space)
    int relu;
};

// The block's stream: its steps [lo, hi) of the flattened (tile, output plane) sequence fall into a first item (tile0, planes zb0 .. zb0 + nz0 - 1, with a
// lead tick in front when zb0 > 0), full tiles, a tail.  Stream position s -> (tile, zo): the tick feeds input planes 2 zo, 2 zo + 1 (zo = zb0 - 1 on the
// lead tick, which feeds plane 2 zb0 - 1 only and stores nothing).
struct ZS2Plan { int nticks, L0, lead0, tile0, zb0, Do; unsigned inv; };
__device__ __forceinline__ bool zs2_entry(const ZS2Plan& p, int s, int& tile, int& zo, bool& lead, bool& first) {
    if (s >= p.nticks) return false;
    if (s < p.L0) { tile = p.tile0; zo = p.zb0 - p.lead0 + s; lead = p.lead0 && s == 0; first = s == 0; return true; }
    const int x = s - p.L0;
    const int k1 = p.Do == 1 ? x : (int)__umulhi((unsigned)x, p.inv);      // x / Do, exact for x < 2^16 (host-checked)
    tile = p.tile0 + 1 + k1; zo = x - k1 * p.Do; lead = false; first = zo == 0;
    return true;
}

__global__ __launch_bounds__(512) void conv3d_zs2_kernel(
    const float* __restrict__ x, const x3_u32x4* __restrict__ wimg, const float* __restrict__ scale, const float* __restrict__ shift,
    float* __restrict__ y, ZS2Dims dm, const float* __restrict__ xmax, float* __restrict__ ymax) {
    using namespace zs2;
    constexpr int OOB = 0x7ffffff0;
    extern __shared__ __attribute__((aligned(16))) x3_byte smem[];
    float* const redmax = reinterpret_cast<float*>(smem + NSLOT * SLB);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kk = lane >> 4;
    // ---- the block's share of the step sequence (neighbouring ranges on one XCD: block ids b, b + 8, ... share an L2)
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int r = ((nblk & 7) == 0) ? (bid & 7) * (nblk >> 3) + (bid >> 3) : bid;
    const long long T = (long long)dm.B * dm.ntiles * dm.Do;
    const long long lo = T * r / nblk, hi = T * (r + 1) / nblk;
    if (hi <= lo) return;
    ZS2Plan pl;
    {
        const int S = (int)(hi - lo);
        pl.tile0 = (int)(lo / dm.Do); pl.zb0 = (int)(lo % dm.Do);
        const int nz0 = min(dm.Do - pl.zb0, S);
        pl.lead0 = pl.zb0 > 0 ? 1 : 0;
        pl.L0 = nz0 + pl.lead0; pl.Do = dm.Do;
        pl.inv = (unsigned)(0x100000000ull / (unsigned)dm.Do) + 1u;
        pl.nticks = S + pl.lead0;
    }

    // ---- scales, weights (register-stationary), this lane's B-fragment offsets inside a plane
    float bound = xmax[lane * 16];
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) bound = fmaxf(bound, __shfl_xor(bound, m));
    float xinv;
    const float xs_scale = x3_pow2_scale(bound, xinv);
    const float unscale = xinv * reinterpret_cast<const float*>(wimg)[1];
    x3_u32x4 wr[KSTEPS][2];
#pragma unroll
    for (int j = 0; j < KSTEPS; ++j)
#pragma unroll
        for (int p = 0; p < 2; ++p) wr[j][p] = wimg[1 + (j * 2 + p) * 64 + lane];
    int boff[SPK];                 // lane (n, kk) supplies the 8 channels of position q = 4 js + kk = (q / 3, q % 3) for output (row wave, column n of the n-tile)
#pragma unroll
    for (int js = 0; js < SPK; ++js) {
        int q = js * 4 + kk;
        if (q >= 9) q = 0;                               // (padding slots: their weights are zero)
        boff[js] = (2 * wave + q / 3) * ROWB + (2 * n + q % 3) * VB;
    }
    const int co0 = kk * 4;
    const x3_f32x4 sc = (scale ? *reinterpret_cast<const x3_f32x4*>(scale + co0) : (x3_f32x4){1.f, 1.f, 1.f, 1.f}) * unscale;
    const x3_f32x4 sh = shift ? *reinterpret_cast<const x3_f32x4*>(shift + co0) : (x3_f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- staging shares of a plane: element e = (halo voxel, float4 of its channels); a thread whose last share lies past the end repeats its previous one
    int loff[NLD], hyx[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e0 = tid + i * 512, e = e0 >= NE ? e0 - 512 : e0, v = e / Q4, c4 = e % Q4;
        const int hy = v / HX, hx = v % HX;
        loff[i] = hy * ROWB + hx * VB + c4 * 8;
        hyx[i] = (hy << 20) | (hx << 8) | c4;
    }
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), (short)0, (int)((long long)dm.B * dm.D * dm.H * dm.W * CIN * 4), 0x00020000);
    const int zstride = dm.H * dm.W * CIN * 4;
    __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(y, (short)0, (int)((long long)dm.B * dm.Do * dm.Ho * dm.Wo * COUT * 4), 0x00020000);

    // fetch side: request the plane pair of stream entry sf (zeros for planes outside the volume, for the even plane of a lead tick, past the end of the stream)
    int goff[NLD];
    int f_tile = -1;
    auto fetch = [&](x3_f32x4 (&q)[2 * NLD], int sf) {
        int tile, zo;
        bool lead, first;
        const bool valid = zs2_entry(pl, sf, tile, zo, lead, first);
        bool in0 = false, in1 = false;
        int zoff0 = 0, zoff1 = 0;
        if (valid) {
            if (tile != f_tile) {
                f_tile = tile;
                const int b = tile / dm.ntiles, t = tile % dm.ntiles;
                const int y0 = (t / dm.tiles_x) * TY, x0 = (t % dm.tiles_x) * TX;
                const int base = (((b * dm.D) * dm.H + (2 * y0 - 1)) * dm.W + (2 * x0 - 1)) * CIN * 4;
#pragma unroll
                for (int i = 0; i < NLD; ++i) {
                    const int hy = hyx[i] >> 20, hx = (hyx[i] >> 8) & 0xfff, c4 = hyx[i] & 0xff;
                    const int gy = 2 * y0 - 1 + hy, gx = 2 * x0 - 1 + hx;
                    goff[i] = (gy >= 0 && gy < dm.H && gx >= 0 && gx < dm.W) ? base + ((hy * dm.W + hx) * CIN + c4 * 4) * 4 : OOB;
                }
            }
            const int z0 = 2 * zo, z1 = 2 * zo + 1;
            in0 = !lead && z0 < dm.D;                     // (zo >= 0 off the lead tick)
            in1 = z1 < dm.D;
            zoff0 = in0 ? z0 * zstride : 0;
            zoff1 = in1 ? z1 * zstride : 0;
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) q[i] = __builtin_bit_cast(x3_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, in0 ? goff[i] : OOB, zoff0, 0));
#pragma unroll
        for (int i = 0; i < NLD; ++i) q[NLD + i] = __builtin_bit_cast(x3_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, in1 ? goff[i] : OOB, zoff1, 0));
    };
    auto stash = [&](const x3_f32x4 (&q)[2 * NLD], int pair) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            x3_byte* sb = smem + (2 * pair + h) * SLB;
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                x3_u32x2 hp, lp;
                x3_split4h(q[h * NLD + i] * xs_scale, hp, lp);
                *reinterpret_cast<x3_u32x2*>(sb + loff[i]) = hp;
                *reinterpret_cast<x3_u32x2*>(sb + PLB + loff[i]) = lp;
            }
        }
    };

    // compute side
    int ob[NTW];                  // byte offset of this lane's float4 of the current output plane, per n-tile of the wave
#pragma unroll
    for (int i = 0; i < NTW; ++i) ob[i] = OOB;
    const int ostep = dm.Ho * dm.Wo * COUT * 4;
    float vmax = 0.0f;
    x3_f32x4 pq[2 * NLD];
    fetch(pq, 0);
    stash(pq, 0);
    fetch(pq, 1);
    __syncthreads();
    x3_f32x4 cur[NTW][2], nxt[NTW][2];                    // [.][magnitude class]: the hh products / the hl + lh products (small terms never meet a large partial sum)
    for (int t = 0; t < pl.nticks; ++t) {
        stash(pq, (t + 1) & 1);                            // stream entry t + 1 (requested during tick t - 1)
        fetch(pq, t + 2);
        int ctile, zo;
        bool lead, first;
        zs2_entry(pl, t, ctile, zo, lead, first);          // (t < nticks: the loop's own bound)
        if (first) {                                       // first tick of an item: empty accumulators, where this lane's voxels of the first stored plane go
            const int b = ctile / dm.ntiles, tl = ctile % dm.ntiles;
            const int oy = (tl / dm.tiles_x) * TY + wave, zf = lead ? zo + 1 : zo;
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                cur[i][0] = cur[i][1] = nxt[i][0] = nxt[i][1] = (x3_f32x4){0.f, 0.f, 0.f, 0.f};
                const int ox = (tl % dm.tiles_x) * TX + 16 * i + n;
                ob[i] = (oy < dm.Ho && ox < dm.Wo) ? (((((b * dm.Do) + zf) * dm.Ho + oy) * dm.Wo + ox) * COUT + co0) * 4 : OOB;
            }
        }
        const x3_byte* s0 = smem + (2 * (t & 1)) * SLB;
        // even plane: kd = 1 -> output zo; odd plane: kd = 2 -> output zo, kd = 0 -> output zo + 1
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const x3_byte* sb = s0 + h * SLB;
#pragma unroll
            for (int js = 0; js < SPK; ++js) {
                x3_u32x4 bq[NTW][2];
#pragma unroll
                for (int i = 0; i < NTW; ++i) {
                    const x3_byte* pb = sb + boff[js] + i * 32 * VB;
                    bq[i][0] = *reinterpret_cast<const x3_u32x4*>(pb);
                    bq[i][1] = *reinterpret_cast<const x3_u32x4*>(pb + PLB);
                }
#pragma unroll
                for (int p = 0; p < 3; ++p) {              // product hh, hl, lh
                    if (h == 0) {
                        const x3_u32x4 a = wr[1 * SPK + js][p == 2 ? 1 : 0];
#pragma unroll
                        for (int i = 0; i < NTW; ++i) cur[i][p > 0] = x3_mfma<2>(a, bq[i][p == 1 ? 1 : 0], cur[i][p > 0]);
                    } else {
                        const x3_u32x4 a2 = wr[2 * SPK + js][p == 2 ? 1 : 0], a0 = wr[0 * SPK + js][p == 2 ? 1 : 0];
#pragma unroll
                        for (int i = 0; i < NTW; ++i) {
                            cur[i][p > 0] = x3_mfma<2>(a2, bq[i][p == 1 ? 1 : 0], cur[i][p > 0]);
                            nxt[i][p > 0] = x3_mfma<2>(a0, bq[i][p == 1 ? 1 : 0], nxt[i][p > 0]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            x3_f32x4 v = (cur[i][0] + cur[i][1]) * sc + sh;
            if (dm.relu) v = __builtin_elementwise_max(v, (x3_f32x4){0.f, 0.f, 0.f, 0.f});
            const bool in = !lead && ob[i] != OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(x3_u32x4, v), yrs, in ? ob[i] : OOB, 0, 0);
            vmax = in ? x3_absmax4(vmax, v) : vmax;
            ob[i] = in ? ob[i] + ostep : ob[i];
            cur[i][0] = nxt[i][0]; cur[i][1] = nxt[i][1];
            nxt[i][0] = nxt[i][1] = (x3_f32x4){0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
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

// RCMVS_ZS2=0: conv1 stays on the split kernel of conv3d_x3.hip (A/B, and the test of that kernel's stride-2 pair form)
bool conv3d_zs2_supported(int Ci, int Co, int kind) {
    static const bool on = [] { const char* e = getenv("RCMVS_ZS2"); return !e || e[0] != '0'; }();
    return on && kind == 1 && Ci == zs2::CIN && Co == zs2::COUT;
}

// x (B, D, H, W, 8) -> y (B, Do, Ho, Wo, 16), stride 2; wimg = the x3h image of the pair's stride-2 kind (conv3d_x3h_pack); xmax required, ymax optional.
// Returns 1 when the volume is not taken (the caller goes on to the split kernel).
int conv3d_zs2_launch(const float* x, const float* wimg, const float* scale, const float* shift, float* y,
                      int B, int D, int H, int W, int relu, hipStream_t st, int max_blocks, const float* xmax, float* ymax) {
    using namespace zs2;
    if (!xmax) return fail(-1, "conv3d_zs2: the fp16-pair form needs a bound of max|x|");
    ZS2Dims dm;
    dm.B = B; dm.D = D; dm.H = H; dm.W = W; dm.relu = relu;
    dm.Do = (D - 1) / 2 + 1; dm.Ho = (H - 1) / 2 + 1; dm.Wo = (W - 1) / 2 + 1;
    if ((long long)B * D * H * W * CIN * 4 >= 0x7ffffff0LL || (long long)B * dm.Do * dm.Ho * dm.Wo * COUT * 4 >= 0x7ffffff0LL)
        return fail(-1, "conv3d_zs2: tensor too large for 32-bit offsets");
    constexpr int MAXDEV = 64;
    static std::atomic<int> cu_of[MAXDEV];
    static std::atomic<bool> raised[MAXDEV];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return fail(-1, "conv3d_zs2: cannot query the device");
    if (cu_of[dev] == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return fail(-1, "conv3d_zs2: cannot query the device");
        cu_of[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    if (!raised[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)conv3d_zs2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
            return fail(-1, "conv3d_zs2: cannot raise the dynamic LDS limit to %d bytes", LDS);
        raised[dev].store(true, std::memory_order_release);
    }
    dm.tiles_x = (dm.Wo + TX - 1) / TX;
    dm.ntiles = dm.tiles_x * ((dm.Ho + TY - 1) / TY);
    const long long T = (long long)B * dm.ntiles * dm.Do;
    const int n_blk = max_blocks > 0 ? max_blocks : cu_of[dev].load();
    if ((T + n_blk - 1) / n_blk + 2 >= 65536) return 1;        // too many steps per block for the 16-bit stream arithmetic
    const int blocks = (int)(T < n_blk ? T : n_blk);
    hipLaunchKernelGGL(conv3d_zs2_kernel, dim3(blocks), dim3(512), LDS, st, x, reinterpret_cast<const x3_u32x4*>(wimg), scale, shift, y, dm, xmax, ymax);
    return launch_status("conv3d_zs2");
}

}  // namespace rcmvs
