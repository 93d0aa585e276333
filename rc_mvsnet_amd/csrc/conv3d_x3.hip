// 3x3x3 convolution family on the bf16 matrix cores at fp32 accuracy ("x3" = three-way operand split), channels-last,
// gfx950 only.  Replaces Conv3d.forward / Deconv3d.forward of the CostRegNet layers (models/modules.py:149-157, 196-204,
// 470-501): stride 1 (conv0 Cin = 8/16/32 -> 8, conv2 16 -> 16), stride 2 (conv1 8 -> 16, conv3 16 -> 32) and the
// transposed stride-2 layer with the most work (conv11 16 -> 8).
//
// Arithmetic.  fp32 MFMA runs at the fp32 vector rate (157 TF); v_mfma_f32_16x16x32_bf16 runs 16x faster.  Every fp32
// operand is split EXACTLY into three bf16 pieces by truncation, x = h + m + l (8 + 8 + 8 significant bits:
// h = x & 0xffff0000, m = (x - h) & 0xffff0000, l = x - h - m; both subtractions are exact), and the product is formed from
// six bf16 MFMAs with fp32 accumulation:   x*w ~= xh*wh + (xh*wm + xm*wh) + (xh*wl + xl*wh + xm*wm).
// The three dropped terms are bounded by 2^-23 |x||w| -- the size of ONE fp32 rounding of the product -- and the three
// magnitude classes are accumulated in separate fp32 accumulators that are added once at the end (small terms never meet a
// large partial sum).  Measured against an fp64 convolution the result is closer than the fp32 FMA-chain kernels
// (tests/test_gpu_parity.py::test_conv3d_x3_vs_fp64).  Six MFMAs at the bf16 rate = 2.6x the fp32 peak.
//
// GEMM view per wave:  D[16 x 16] += A[16 x 32] * B[32 x 16] with A = weights (register-stationary for the whole kernel),
//   B = activations read from LDS with one ds_read_b128 per piece, N = 16 columns = 16 voxels of the "tile grid" (output
//   voxels for the convolutions, INPUT cells for the transposed convolution).
//   conv, Cout >= 16:  M = output channel.                                                              (map X3_PL)
//   conv, Cout = 8  :  M = (s, co) -- TWO output positions share one column of activations: y0 + s (Cin = 16/32, X3_YT,
//               the K axis walks kh' = kh + s = 0..3) or x0 + 2n + s (Cin = 8, X3_XT, K walks kw' = kw + s = 0..3, columns
//               are every second voxel).  A holds the weight of tap k' - s, zero where that tap does not exist
//               (block-Toeplitz): the 16-row tile is full at 3/4 density instead of half empty.
//   transposed:        M = (output parity class p in 0..7, co): the eight outputs 2*cell + p of a cell share its 2x2x2
//               input neighbourhood; A holds tap p - 2*nb + 1 per axis (zero where it falls outside 0..2).
//   K step = 32 = (32 / Cin) tap positions x Cin channels; lane (n = l & 15, kk = l >> 4) supplies channels 8*kk.. of its position.
//
// Data flow.  512 threads = NCW consumer waves (MFMA + ds_read only) + NPW producer waves (4 + 4; 6 + 2 for the layers whose weight
// image exceeds 4 x 128 registers: 32 -> 32 and the transposed 32 -> 16).  A block owns a TY x TX tile of the tile grid and
// marches over z.  The input z-slices it needs (halo included, already split into the three bf16 piece planes) sit in an LDS ring
// of 2 NKD slices; during tick s the consumers work on step s while the producers (1) split + store the slice(s) step s+1 adds
// (fetched into registers two ticks earlier with raw buffer loads, out of range -> 0 = the zero padding), (2) issue the loads of
// step s+3 and (3) finish the K-split partial tiles of step s-1: one barrier per step, every input voxel is read once per block
// (x halo) and split once instead of once per tap.  Blocks are persistent (one per CU): a block walks its (batch, xy tile, z chunk)
// work items back to back with the ring running across the item boundary.
// Weights: K (and M) are cut over the consumer waves until a wave's share fits ~100 registers; with a K cut the partial
// 16 x 16 tiles go through LDS and the producers finish them (sum, BN scale/shift, ReLU, skip-add, store).
// Kinds: stride 1, stride 2, transposed stride 2, and "planar" (kd = 1 taps only: every z-plane on its own -- what a 3x3x3 conv
// of a one-plane volume reduces to; the 2-D FeatureNet layers run as such volumes).
// What bounds it (measured): on gfx950 VALU work -- of another wave on the SIMD or interleaved in the same
// wave -- does not overlap v_mfma_f32_16x16x32_bf16, so a step costs MFMA time + producer VALU time; the producers are therefore
// kept to the bare split (22 VALU per float4) and table-driven addressing.
#include "common.h"

namespace rcmvs {

typedef __bf16 x3_bf16x8 __attribute__((ext_vector_type(8)));
typedef float x3_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int x3_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int x3_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned char x3_byte;

// stride-1 conv, stride-2 conv, transposed stride-2 conv, planar stride-1 conv (kd = 1 taps only: a 3x3 conv of every z-plane,
// what a 3x3x3 conv of a one-plane volume reduces to -- the FeatureNet layers, models/modules.py:413-424)
enum { X3_S1 = 0, X3_S2 = 1, X3_T2 = 2, X3_P1 = 3, X3_KINDS = 4 };
__host__ __device__ constexpr bool x3_unit(int kind) { return kind == X3_S1 || kind == X3_P1; }       // stride-1 geometry in the plane
enum { X3_XT = 0, X3_YT = 1, X3_PL = 2 };        // how an n-tile maps to voxels (see above)

template <int CIN, int COUT, int KIND>
struct X3 {
    static constexpr int MAP = (x3_unit(KIND) && COUT == 8) ? (CIN == 8 ? X3_XT : X3_YT) : X3_PL;
    static constexpr int MROWS = (KIND == X3_T2) ? 8 * COUT : (COUT == 8 ? 16 : COUT);
    static constexpr int MT_ALL = MROWS / 16;                            // 16-row tiles of A
    static constexpr int VB = CIN * 2;                                   // bytes per voxel per piece plane
    static constexpr int PPS = (CIN >= 32) ? 1 : 32 / CIN;               // tap positions per K step
    static constexpr int HALVES = (CIN > 32) ? CIN / 32 : 1;            // K steps per position
    static constexpr int NKD = (KIND == X3_T2) ? 2 : (KIND == X3_P1 ? 1 : 3);                  // input z-planes one output step reads
    static constexpr int QR = (KIND == X3_T2) ? 2 : (MAP == X3_YT ? 4 : 3);   // position grid of one plane: rows ...
    static constexpr int QC = (KIND == X3_T2) ? 2 : (MAP == X3_XT ? 4 : 3);   // ... and columns
    static constexpr int PPKD = QR * QC;
    static constexpr int SPK = ((PPKD + PPS - 1) / PPS) * HALVES;       // K steps per plane (a step never straddles planes)
    static constexpr int KSTEPS = NKD * SPK;
    static constexpr int WREG = KSTEPS * MT_ALL * 12;                    // registers the whole weight image would take
    // wave roles: 4 consumer + 4 producer waves; layers whose weight image exceeds 4 x 128 registers take 6 consumer waves (K cut
    // three ways, M two ways) and 2 producer waves -- their volumes are small and the producers have little to do
    static constexpr int NCW = (WREG > 512) ? 6 : 4;
    static constexpr int NPW = 8 - NCW;
    static constexpr int MSPLIT = (MT_ALL >= 2 && WREG > 128) ? 2 : 1;
    static constexpr int KSPLIT = (NCW == 6) ? 3 : ((WREG / MSPLIT > 256) ? 4 / MSPLIT : ((WREG / MSPLIT > 128) ? 2 : 1));
    static constexpr int MT = MT_ALL / MSPLIT;                           // m-tiles per consumer wave
    static constexpr int KSW = (KSTEPS + KSPLIT - 1) / KSPLIT;          // K steps per consumer wave
    static constexpr int TX = ((x3_unit(KIND) && CIN >= 32) || (KIND == X3_S2 && CIN >= 16) || NCW == 6) ? 16 : 32;
    static constexpr int TY = (MAP == X3_XT) ? 8 : ((KIND == X3_S2 || NCW == 6) ? 2 : 4);
    static constexpr int CS = (MAP == X3_XT || KIND == X3_S2) ? 2 : 1;   // voxels between neighbouring columns
    static constexpr int RS = (MAP == X3_YT || KIND == X3_S2) ? 2 : 1;   // halo rows between neighbouring tile rows
    static constexpr int TYP = x3_unit(KIND) ? TY + 2 : (KIND == X3_S2 ? 2 * TY + 1 : TY + 1);
    static constexpr int TXP = x3_unit(KIND) ? TX + 2 : (KIND == X3_S2 ? 2 * TX + 1 : TX + 1);
    static constexpr int ROWB = TXP * VB;                                // bytes per halo row
    static constexpr int PLB = TYP * ROWB;                               // bytes per piece plane
    static constexpr int SLB = 3 * PLB;                                  // bytes per z-slice (h, m, l planes)
    static constexpr int ZADV = (KIND == X3_S2) ? 2 : 1;                 // input slices consumed per step
    static constexpr int NSLOT = 2 * NKD;                                // ring: the planes being read + the ones being written (up to NKD when the next item starts)
    static constexpr int NTX = (MAP == X3_XT) ? TX / 32 : TX / 16;       // n-tiles along x
    static constexpr int NTILE = ((MAP == X3_YT) ? TY / 2 : TY) * NTX;
    static constexpr int NG = NCW / (KSPLIT * MSPLIT);                     // consumer waves that own different n-tiles
    static constexpr int NTW = NTILE / NG;                               // n-tiles per consumer wave
    static constexpr int TP = (MT <= 2 && NTW % 2 == 0) ? 2 : 1;         // n-tiles in flight (independent accumulators)
    static constexpr int Q4 = CIN / 4;                                   // float4 per voxel
    static constexpr int NLOAD = TYP * TXP * Q4;                         // float4 per z-slice
    static constexpr int NPF = (NLOAD + NPW * 64 - 1) / (NPW * 64);                      // float4 per producer thread per z-slice
    static constexpr int PARTB = (KSPLIT > 1) ? NTILE * MT_ALL * KSPLIT * 1024 : 0;   // one buffer of partial output tiles
    static constexpr int LDSB = NSLOT * SLB + 2 * PARTB;
    static_assert(NCW % (KSPLIT * MSPLIT) == 0 && WREG / (KSPLIT * MSPLIT) <= 128, "weight slice per wave");
    static_assert(NTILE % NG == 0 && NTW % TP == 0, "tiles per wave");
    static_assert(MAP != X3_XT || PPS == 4, "XT packs the four kw' positions of 8 channels into one K step");
    static_assert(LDSB <= 160 * 1024, "LDS budget");
};

// which (position, channel) a lane slot of K step j holds: q = position index inside the plane (>= PPKD: padding), ci0 = first channel
template <class C>
__host__ __device__ inline void x3_kslot(int js, int kk, int& q, int& ci0) {
    if (C::HALVES == 1) { q = js * C::PPS + kk / (4 / C::PPS); ci0 = (kk % (4 / C::PPS)) * 8; }
    else { q = js / C::HALVES; ci0 = (js % C::HALVES) * 32 + kk * 8; }
}

// LDS bank swizzle of a voxel's bytes inside a piece plane, as an XOR mask on the byte offset inside the voxel (hc = halo column).
// ds_read_b128 is served in four 16-lane groups ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH "LDS"), banks = (addr / 4) % 64.  With
// Cin = 32 a voxel is 64 B, lane (n, kk) reads 16 B at 64 n + 16 kk and lanes n, n + 4 (n + 12) of a group collide: measured 43 %
// of the LDS cycles.  Flipping bit 1 of the 16-byte slot on every second group of four columns makes all four groups
// conflict-free for each of the three tap columns (exhaustive check: tests/test_x3_swizzle_cpu.py); other layouts read conflict-free as they are.
// With Cin = 64 a voxel is 128 B and a lane group sees only four distinct 16-byte slots (4-way conflicts); the slot's low three bits
// (channel half, kk) are XORed with a per-column-pair value found by exhaustive search over the lane groups, both channel halves and
// the three tap columns: table[hc >> 1] = 0 1 2 4 5 6 2 6 0 (3 bits each, packed below), conflict-free for the 18 halo columns.
template <class C, int CIN, int KIND>
__host__ __device__ inline int x3_swz(int hc) {
    if (CIN == 32 && (x3_unit(KIND) || KIND == X3_T2)) return ((hc >> 2) & 1) * 32;       // transposed: same geometry (columns one voxel apart, tap columns 0..1)
    if (CIN == 64 && x3_unit(KIND)) return (int)((0xCB5888u >> (3 * (hc >> 1))) & 7u) * 16;
    return 0;
}

// ---- weight image: [K step][piece][m-tile][lane][8 bf16], the A fragment of v_mfma_f32_16x16x32_bf16 (row = lane & 15,
// k = 8 * (lane >> 4) + e), pieces split by truncation like the activations.
// transposed: 0 = Conv3d weight (Co,Ci,27); 1 = ConvTranspose3d weight (Ci,Co,27) (X3_T2 only); 2 = (Ci,Co,27) with flipped
// taps = the adjoint of a stride-1 conv (training data gradient).
template <int CIN, int COUT, int KIND>
__global__ void x3_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ img, int transposed) {
    using C = X3<CIN, COUT, KIND>;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= C::KSTEPS * C::MT_ALL * 64 * 8) return;
    const int e = t & 7, lane = (t >> 3) & 63, mt = (t >> 9) % C::MT_ALL, j = (t >> 9) / C::MT_ALL;
    const int m = mt * 16 + (lane & 15), kk = lane >> 4;
    const int kd = j / C::SPK, js = j % C::SPK;
    int q, ci;
    x3_kslot<C>(js, kk, q, ci);
    ci += e;
    float v = 0.0f;
    if (q < C::PPKD) {
        const int r = q / C::QC, c = q % C::QC;
        int co, td, th, tw;       // output channel and tap per axis (planar: kd is the centre plane of the 3x3x3 weight)
        if (KIND == X3_T2) {
            const int p = m / COUT;
            co = m % COUT;
            td = ((p >> 2) & 1) - 2 * kd + 1; th = ((p >> 1) & 1) - 2 * r + 1; tw = (p & 1) - 2 * c + 1;
        } else if (C::MAP == X3_XT) { co = m & 7; td = kd; th = r; tw = c - (m >> 3); }
        else if (C::MAP == X3_YT) { co = m & 7; td = kd; th = r - (m >> 3); tw = c; }
        else { co = m; td = kd; th = r; tw = c; }
        if (KIND == X3_P1) td = 1;
        if (td >= 0 && td < 3 && th >= 0 && th < 3 && tw >= 0 && tw < 3 && co < COUT) {
            const int tap = (td * 3 + th) * 3 + tw;
            v = transposed ? w[((long long)ci * COUT + co) * 27 + (transposed == 2 ? 26 - tap : tap)] : w[((long long)co * CIN + ci) * 27 + tap];
        }
    }
    const unsigned hb = __float_as_uint(v) & 0xffff0000u;
    const float r1 = v - __uint_as_float(hb);
    const unsigned mb = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(mb);
    const unsigned lb = __float_as_uint(r2) & 0xffff0000u;
    const long long base = (((long long)j * 3) * C::MT_ALL + mt) * 512 + lane * 8 + e;     // piece stride = MT_ALL * 512 shorts
    img[base] = (unsigned short)(hb >> 16);
    img[base + (long long)C::MT_ALL * 512] = (unsigned short)(mb >> 16);
    img[base + 2LL * C::MT_ALL * 512] = (unsigned short)(lb >> 16);
}

// four fp32 -> the three bf16 piece quadruples (two dwords each).  18 VALU operations: the packing v_perm_b32 takes the high
// halves (= truncation), the remainders use packed subtractions.  The producers' VALU work is NOT hidden behind the consumers'
// MFMAs -- a wave issuing MFMAs back to back leaves a second wave on its SIMD ~10 % of the VALU issue rate (measured,
// tools/dev/coissue.hip) -- so every instruction here is paid for in matrix-pipe idle time.
typedef float x3_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void x3_split4(x3_f32x4 v, x3_u32x2& h, x3_u32x2& m, x3_u32x2& l) {
    const x3_u32x4 vb = __builtin_bit_cast(x3_u32x4, v);
    const x3_u32x4 hb = vb & 0xffff0000u;
    const x3_f32x4 r1 = v - __builtin_bit_cast(x3_f32x4, hb);                       // exact
    const x3_u32x4 r1b = __builtin_bit_cast(x3_u32x4, r1);
    const x3_u32x4 mb = r1b & 0xffff0000u;
    const x3_f32x4 r2 = r1 - __builtin_bit_cast(x3_f32x4, mb);                      // exact, <= 8 significant bits
    const x3_u32x4 lb = __builtin_bit_cast(x3_u32x4, r2);
    // v_perm_b32: (hi16 of b) << 16 | (hi16 of a)
    h.x = __builtin_amdgcn_perm(vb[1], vb[0], 0x07060302u); h.y = __builtin_amdgcn_perm(vb[3], vb[2], 0x07060302u);
    m.x = __builtin_amdgcn_perm(r1b[1], r1b[0], 0x07060302u); m.y = __builtin_amdgcn_perm(r1b[3], r1b[2], 0x07060302u);
    l.x = __builtin_amdgcn_perm(lb[1], lb[0], 0x07060302u); l.y = __builtin_amdgcn_perm(lb[3], lb[2], 0x07060302u);
}

struct X3Dims {
    int D, H, W;        // input volume
    int Do, Ho, Wo;     // output volume
    int Dt;             // extent of the tile grid in z (= Do for the convolutions, D for the transposed convolution)
    int tiles_x, zchunk, relu;
    int B, ntiles, nchunks, nitems;   // work items = B x xy tiles x z chunks
    int s2d;            // planar kind only: the input is physically (B, D, 2H, 2W, CIN / 4) and is read through a space-to-depth view
                        // (channel (py, px, c) of voxel (y, x) = channel c of pixel (2y + py, 2x + px)): a 5x5 stride-2 layer as a 3x3 one
};

// output voxel + first channel of the float4 a lane holds for (tile tl, m-tile mtg, step z); false = outside the volume
template <class C, int COUT, int KIND>
__device__ __forceinline__ bool x3_out_coord(const X3Dims& dm, int b, int x0, int y0, int z, int tl, int mtg, int n, int g4, long long& ov, int& co0) {
    const int trow = tl / C::NTX, tcol = tl % C::NTX;
    int oz = z, oy, ox;
    if (KIND == X3_T2) {
        const int m0 = mtg * 16 + 4 * g4, p = m0 / COUT;
        co0 = m0 % COUT;
        oz = 2 * z + ((p >> 2) & 1); oy = 2 * (y0 + trow) + ((p >> 1) & 1); ox = 2 * (x0 + tcol * 16 + n) + (p & 1);
    } else if (C::MAP == X3_XT) { oy = y0 + trow; ox = x0 + tcol * 32 + 2 * n + (g4 >> 1); co0 = (g4 & 1) * 4; }
    else if (C::MAP == X3_YT) { oy = y0 + 2 * trow + (g4 >> 1); ox = x0 + tcol * 16 + n; co0 = (g4 & 1) * 4; }
    else { oy = y0 + trow; ox = x0 + tcol * 16 + n; co0 = mtg * 16 + 4 * g4; }
    ov = (((long long)b * dm.Do + oz) * dm.Ho + oy) * dm.Wo + ox;
    return oz < dm.Do && oy < dm.Ho && ox < dm.Wo;
}

// One work item = (batch, xy tile, z chunk) of the tile grid.  A block walks its items back to back: the ring keeps running
// across the item boundary (the first planes of the next item are staged during the last step of the current one), so weights
// are loaded once per block and the ring prologue is paid once per block instead of once per item.
struct X3Item { int b, x0, y0, zb, ze; };
template <class C>
__device__ __forceinline__ X3Item x3_item(const X3Dims& dm, int it) {
    X3Item w;
    const int zc = it % dm.nchunks;
    const int r = it / dm.nchunks;
    const int tile = r % dm.ntiles;
    w.b = r / dm.ntiles;
    w.x0 = (tile % dm.tiles_x) * C::TX;
    w.y0 = (tile / dm.tiles_x) * C::TY;
    w.zb = zc * dm.zchunk;
    w.ze = min(dm.Dt, w.zb + dm.zchunk);
    return w;
}
// a step of the block's schedule: item + z inside it; advance() moves to the next step (next item after the last z)
template <class C>
struct X3Step {
    int it, z;
    bool live, first;
    X3Item w;
    __device__ __forceinline__ void start(const X3Dims& dm, int it0) {
        it = it0; live = it < dm.nitems; first = true;
        if (live) { w = x3_item<C>(dm, it); z = w.zb; }
    }
    __device__ __forceinline__ void advance(const X3Dims& dm, int stride) {
        if (!live) return;
        first = false;
        if (++z >= w.ze) {
            it += stride; live = it < dm.nitems; first = true;
            if (live) { w = x3_item<C>(dm, it); z = w.zb; }
        }
    }
};

template <int CIN, int COUT, int KIND>
__global__ __launch_bounds__(512) void conv3d_x3_kernel(
    const float* __restrict__ x, const x3_u32x4* __restrict__ wimg, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ res, float* __restrict__ y, X3Dims dm) {
    using C = X3<CIN, COUT, KIND>;
    constexpr int MT = C::MT, KSW = C::KSW, KSPLIT = C::KSPLIT, TP = C::TP, NSLOT = C::NSLOT, NKD = C::NKD, ZADV = C::ZADV;
    extern __shared__ __attribute__((aligned(16))) x3_byte smem[];
    x3_byte* const partbase = smem + NSLOT * C::SLB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= C::NCW;
    const int n = lane & 15, kk = lane >> 4;
    // item order: with a block count that is a multiple of 8 the blocks of one XCD (every 8th block id, MI355X_MICROARCH
    // "workgroup dispatch") take one contiguous eighth of the items, so neighbouring tiles share that XCD's L2; otherwise plain
    // round robin.  Either way block `bid` walks items it_first, it_first + it_stride, ... < it_limit.
    const int nblk = gridDim.x, bid = blockIdx.x;
    int it_first, it_stride, it_limit;
    if ((nblk & 7) == 0) {
        const int xcd = bid & 7;
        const int lo = (int)((long long)dm.nitems * xcd / 8), hi = (int)((long long)dm.nitems * (xcd + 1) / 8);
        it_first = lo + (bid >> 3); it_stride = nblk >> 3; it_limit = hi;
    } else {
        it_first = bid; it_stride = nblk; it_limit = dm.nitems;
    }
    X3Dims dmx = dm;
    dmx.nitems = it_limit;                      // the step iterators stop at the end of this block's range
    const bool relu = dm.relu != 0;
    if (it_first >= it_limit) return;            // (more blocks than items in this XCD's range)

    if (!producer) {
        // =============================== consumer: register-stationary weights of this wave's (K, M) slice
        const int ks = wave % KSPLIT, ms = (wave / KSPLIT) % C::MSPLIT, grp = wave / (KSPLIT * C::MSPLIT);
        x3_bf16x8 wr[KSW][3][MT];
        int kdj[KSW];      // input plane of K step j (wave-uniform)
        int boff[KSW];     // this lane's byte offset of the B fragment inside a z-slice, tile origin excluded
#pragma unroll
        for (int j = 0; j < KSW; ++j) {
            const int jg = ks * KSW + j;
            const bool live = jg < C::KSTEPS;
            const int jc = live ? jg : 0;
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    x3_u32x4 v = wimg[((jc * 3 + p) * C::MT_ALL + ms * MT + mt) * 64 + lane];
                    if (!live) v = (x3_u32x4){0u, 0u, 0u, 0u};
                    wr[j][p][mt] = __builtin_bit_cast(x3_bf16x8, v);
                }
            kdj[j] = jc / C::SPK;
            int q, ci0;
            x3_kslot<C>(jc % C::SPK, kk, q, ci0);
            if (q >= C::PPKD) q = 0;
            const int hc = q % C::QC + n * C::CS;      // halo column of this lane's voxel (tile column offsets are multiples of 16: same swizzle)
            boff[j] = (q / C::QC) * C::ROWB + hc * C::VB + ((ci0 * 2) ^ x3_swz<C, CIN, KIND>(hc));
        }
        x3_f32x4 sc[MT], sh[MT];      // epilogue constants (used when this wave finishes its own tiles: KSPLIT == 1)
        if constexpr (KSPLIT == 1) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                long long ov; int co0;
                x3_out_coord<C, COUT, KIND>(dm, 0, 0, 0, 0, 0, ms * MT + mt, n, kk, ov, co0);
                sc[mt] = scale ? *reinterpret_cast<const x3_f32x4*>(scale + co0) : (x3_f32x4){1.f, 1.f, 1.f, 1.f};
                sh[mt] = shift ? *reinterpret_cast<const x3_f32x4*>(shift + co0) : (x3_f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        X3Step<C> st;
        st.start(dmx, it_first);
        __syncthreads();          // the planes of the first step are in the ring
        int s0 = 0;               // ring slot of the first input plane of the current step
        int tick = 0;
#pragma unroll 1
        while (st.live) {
            const int z = st.z, x0 = st.w.x0, y0 = st.w.y0, b = st.w.b;
            int slotoff[NKD];
#pragma unroll
            for (int k = 0; k < NKD; ++k) slotoff[k] = ((s0 + k) % NSLOT) * C::SLB;
#pragma unroll
            for (int tp = 0; tp < C::NTW / TP; ++tp) {
                int toff[TP], tl[TP];
#pragma unroll
                for (int t = 0; t < TP; ++t) {
                    tl[t] = grp + C::NG * (TP * tp + t);
                    toff[t] = (tl[t] / C::NTX) * C::RS * C::ROWB + (tl[t] % C::NTX) * 16 * C::CS * C::VB;
                }
                // skip-connection values of the tiles this wave finishes itself: loaded before the MFMA phase
                x3_f32x4 rv[TP][MT];
                long long ovv[TP][MT];
                bool okv[TP][MT];
                if constexpr (KSPLIT == 1) {
#pragma unroll
                    for (int t = 0; t < TP; ++t)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            int co0;
                            okv[t][mt] = x3_out_coord<C, COUT, KIND>(dm, b, x0, y0, z, tl[t], ms * MT + mt, n, kk, ovv[t][mt], co0);
                            ovv[t][mt] = ovv[t][mt] * COUT + co0;
                            rv[t][mt] = (res && okv[t][mt]) ? *reinterpret_cast<const x3_f32x4*>(res + ovv[t][mt]) : (x3_f32x4){0.f, 0.f, 0.f, 0.f};
                        }
                }
                x3_f32x4 acc[TP][MT][3];
#pragma unroll
                for (int t = 0; t < TP; ++t)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int a = 0; a < 3; ++a) acc[t][mt][a] = (x3_f32x4){0.f, 0.f, 0.f, 0.f};
                // software pipeline: the B fragments of K step j+1 are read while the MFMAs of step j run; the MFMA order keeps
                // >= 3 independent instructions between two uses of one accumulator
                x3_bf16x8 bq[2][TP][3];
                {
                    const int a0 = boff[0] + slotoff[kdj[0]];
#pragma unroll
                    for (int t = 0; t < TP; ++t)
#pragma unroll
                        for (int p = 0; p < 3; ++p) bq[0][t][p] = *reinterpret_cast<const x3_bf16x8*>(smem + a0 + toff[t] + p * C::PLB);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < KSW; ++j) {
                    const int cur = j & 1, nxt = cur ^ 1;
                    const bool pre = j + 1 < KSW;
                    const int na = pre ? boff[j + 1] + slotoff[kdj[j + 1]] : 0;
                    // six slots: one B-fragment read of K step j+1, then the MFMAs of one product class of step j; the fences pin
                    // this order (left alone, the scheduler sinks every read to just before its first use and exposes the LDS latency)
#define X3_MF(ACC, WP, BP) _Pragma("unroll") for (int t = 0; t < TP; ++t) _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) \
        acc[t][mt][ACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[j][WP][mt], bq[cur][t][BP], acc[t][mt][ACC], 0, 0, 0)
#define X3_LD(I) if (pre && (I) < TP * 3) bq[nxt][((I) / 3) % TP][(I) % 3] = *reinterpret_cast<const x3_bf16x8*>(smem + na + toff[((I) / 3) % TP] + ((I) % 3) * C::PLB)
                    X3_LD(0); X3_MF(2, 0, 2); __builtin_amdgcn_sched_barrier(0);
                    X3_LD(1); X3_MF(1, 0, 1); __builtin_amdgcn_sched_barrier(0);
                    X3_LD(2); X3_MF(0, 0, 0); __builtin_amdgcn_sched_barrier(0);
                    X3_LD(3); X3_MF(2, 1, 1); __builtin_amdgcn_sched_barrier(0);
                    X3_LD(4); X3_MF(1, 1, 0); __builtin_amdgcn_sched_barrier(0);
                    X3_LD(5); X3_MF(2, 2, 0); __builtin_amdgcn_sched_barrier(0);
#undef X3_MF
#undef X3_LD
                }
#pragma unroll
                for (int t = 0; t < TP; ++t)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        x3_f32x4 v = acc[t][mt][0] + (acc[t][mt][1] + acc[t][mt][2]);
                        if constexpr (KSPLIT > 1) {
                            // hand the partial tile to the producers: [tile][m-tile][K slice][lane]
                            x3_f32x4* part = reinterpret_cast<x3_f32x4*>(partbase + (tick & 1) * C::PARTB);
                            part[((tl[t] * C::MT_ALL + ms * MT + mt) * KSPLIT + ks) * 64 + lane] = v;
                        } else if (okv[t][mt]) {
                            v = v * sc[mt] + sh[mt];
                            if (relu) v = __builtin_elementwise_max(v, (x3_f32x4){0.f, 0.f, 0.f, 0.f});
                            v += rv[t][mt];
                            *reinterpret_cast<x3_f32x4*>(y + ovv[t][mt]) = v;
                        }
                    }
            }
            st.advance(dmx, it_stride);
            s0 = (s0 + (st.first ? NKD : ZADV)) % NSLOT;      // a new item starts right behind the last plane of the previous one
            ++tick;
            __syncthreads();
        }
    } else {
        // =============================== producer
        const int pw = wave - C::NCW, ptid = tid - C::NCW * 64;
        constexpr int OOB = 0x7ffffff0;
        const long long vol = (long long)dm.B * dm.D * dm.H * dm.W * CIN * 4;
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), (short)0, (int)vol, 0x00020000);
        // which float4 of a halo slice this thread moves: halo row / column and offsets (item-invariant), and -- recomputed when the
        // fetch side moves to another item -- the absolute byte offset of that float4 in input plane 0 (OOB outside the volume)
        int hrc[C::NPF], grel[C::NPF], loff[C::NPF], goff[C::NPF];
#pragma unroll
        for (int i = 0; i < C::NPF; ++i) {
            const int e = ptid + i * (C::NPW * 64);
            const int v = e / C::Q4, c4 = e - v * C::Q4;
            const int hr = v / C::TXP, hc = v - hr * C::TXP;
            hrc[i] = (e < C::NLOAD) ? ((hr << 16) | hc) : -1;
            if (dm.s2d) {
                constexpr int CP = CIN / 4, QP = CP / 4 > 0 ? CP / 4 : 1;      // physical channels, float4 per physical pixel
                const int par = c4 / QP, cq = c4 - par * QP;
                grel[i] = (((2 * hr + (par >> 1)) * (2 * dm.W) + 2 * hc + (par & 1)) * CP + cq * 4) * 4;
            } else {
                grel[i] = ((hr * dm.W + hc) * CIN + c4 * 4) * 4;
            }
            loff[i] = (e < C::NLOAD) ? hr * C::ROWB + hc * C::VB + ((c4 * 8) ^ x3_swz<C, CIN, KIND>(hc)) : -1;
            goff[i] = OOB;
        }
        const int zstride = dm.H * dm.W * CIN * 4;
        int goff_item = -1;
        auto set_item = [&](const X3Step<C>& t) {
            if (t.it == goff_item) return;
            goff_item = t.it;
            const X3Item& w = t.w;
            const int hy0 = x3_unit(KIND) ? w.y0 - 1 : (KIND == X3_S2 ? 2 * w.y0 - 1 : w.y0);
            const int hx0 = x3_unit(KIND) ? w.x0 - 1 : (KIND == X3_S2 ? 2 * w.x0 - 1 : w.x0);
            const int base = dm.s2d ? ((w.b * dm.D * 2 * dm.H + 2 * hy0) * (2 * dm.W) + 2 * hx0) * CIN
                                    : (w.b * dm.D * dm.H + hy0) * dm.W * CIN * 4 + hx0 * CIN * 4;
#pragma unroll
            for (int i = 0; i < C::NPF; ++i) {
                const int gy = hy0 + (hrc[i] >> 16), gx = hx0 + (hrc[i] & 0xffff);
                const bool ok = hrc[i] >= 0 && gy >= 0 && gy < dm.H && gx >= 0 && gx < dm.W;
                goff[i] = ok ? base + grel[i] : OOB;
            }
        };
        // fetch input plane zi of the current fetch item's halo tile: one add per float4 (an OOB entry stays out of range: the sum
        // of two offsets below 2^31 does not wrap, and the buffer bounds check compares unsigned)
        auto fetch = [&](x3_f32x4 (&pf)[C::NPF], int zi) {
            const bool zin = zi >= 0 && zi < dm.D;
            const int zoff = zi * zstride;
#pragma unroll
            for (int i = 0; i < C::NPF; ++i)
                pf[i] = __builtin_bit_cast(x3_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, zin ? goff[i] + zoff : OOB, 0, 0));
        };
        auto stash = [&](const x3_f32x4 (&pf)[C::NPF], int slot) {
            x3_byte* sb = smem + slot * C::SLB;
#pragma unroll
            for (int i = 0; i < C::NPF; ++i) {
                if (loff[i] < 0) continue;
                x3_u32x2 h, m, l;
                x3_split4(pf[i], h, m, l);
                *reinterpret_cast<x3_u32x2*>(sb + loff[i]) = h;
                *reinterpret_cast<x3_u32x2*>(sb + C::PLB + loff[i]) = m;
                *reinterpret_cast<x3_u32x2*>(sb + 2 * C::PLB + loff[i]) = l;
            }
        };
        auto zin0 = [&](int z) { return (KIND == X3_S1) ? z - 1 : (KIND == X3_S2 ? 2 * z - 1 : z); };      // (planar: the plane itself)   // first input plane of step z
        // the planes a step adds to the ring: all NKD on the first step of an item, the last ZADV afterwards
        auto new_planes = [&](const X3Step<C>& t, int& first_plane) { first_plane = t.first ? zin0(t.z) : zin0(t.z) + NKD - ZADV; return t.first ? NKD : ZADV; };
        // finish the K-split tiles of a step: sum the partial tiles, BN scale/shift, ReLU, skip-add, store.  Two halves: epi_open at
        // the start of the tick works out where this wave's (tile, m-tile) units go and issues the skip-connection loads; epi_close
        // at the end of the tick (a stash and a fetch later) does the arithmetic, so the loads' latency is off the tick's critical path
        constexpr int NEU = (KSPLIT > 1) ? (C::NTILE * C::MT_ALL + C::NPW - 1) / C::NPW : 1;
        x3_f32x4 esc[NEU], esh[NEU], erv[NEU];
        long long eov[NEU];            // element offset of the unit's float4 in y / res; < 0 = nothing to store
        if constexpr (KSPLIT > 1) {
#pragma unroll
            for (int i = 0; i < NEU; ++i) {
                const int u = min(pw + C::NPW * i, C::NTILE * C::MT_ALL - 1);
                long long ov; int co0;
                x3_out_coord<C, COUT, KIND>(dm, 0, 0, 0, 0, 0, u % C::MT_ALL, n, kk, ov, co0);      // co0 depends on the m-tile and the lane only
                esc[i] = scale ? *reinterpret_cast<const x3_f32x4*>(scale + co0) : (x3_f32x4){1.f, 1.f, 1.f, 1.f};
                esh[i] = shift ? *reinterpret_cast<const x3_f32x4*>(shift + co0) : (x3_f32x4){0.f, 0.f, 0.f, 0.f};
                eov[i] = -1;
            }
        }
        auto epi_open = [&](const X3Item& w, int z) {
            if constexpr (KSPLIT > 1) {
#pragma unroll
                for (int i = 0; i < NEU; ++i) {
                    const int u = pw + C::NPW * i;                  // (tile, m-tile) unit of this producer wave
                    long long ov; int co0;
                    const bool ok = u < C::NTILE * C::MT_ALL && x3_out_coord<C, COUT, KIND>(dm, w.b, w.x0, w.y0, z, u / C::MT_ALL, u % C::MT_ALL, n, kk, ov, co0);
                    eov[i] = ok ? ov * COUT + co0 : -1;
                    erv[i] = (res && ok) ? *reinterpret_cast<const x3_f32x4*>(res + eov[i]) : (x3_f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
        };
        auto epi_close = [&](int buf) {
            if constexpr (KSPLIT > 1) {
                const x3_f32x4* part = reinterpret_cast<const x3_f32x4*>(partbase + buf * C::PARTB);
#pragma unroll
                for (int i = 0; i < NEU; ++i) {
                    if (eov[i] < 0) continue;
                    const x3_f32x4* pp = part + (pw + C::NPW * i) * KSPLIT * 64 + lane;
                    x3_f32x4 v = pp[0];
#pragma unroll
                    for (int k = 1; k < KSPLIT; ++k) v += pp[k * 64];
                    v = v * esc[i] + esh[i];
                    if (relu) v = __builtin_elementwise_max(v, (x3_f32x4){0.f, 0.f, 0.f, 0.f});
                    v += erv[i];
                    *reinterpret_cast<x3_f32x4*>(y + eov[i]) = v;
                }
            }
        };
        // schedule: during tick s the consumers compute step s while the producers (1) store into the ring the planes step s+1
        // adds, (2) issue the loads of the planes step s+3 adds -- two ticks of flight time, a load that misses to HBM under
        // load takes longer than one tick -- and (3) finish step s-1.  Register queue: buffer (t & 1) holds step t's planes from
        // tick t-3 until they are stored in tick t-1.
        X3Step<C> s_cur, s_n1, s_n2, s_n3;
        x3_f32x4 pfq[2][NKD][C::NPF];
        int wslot = 0;
        auto fetch_step = [&](const X3Step<C>& t, x3_f32x4 (&q)[NKD][C::NPF]) {
            if (!t.live) return;
            int p;
            const int c = new_planes(t, p);
            set_item(t);
#pragma unroll
            for (int k = 0; k < NKD; ++k) if (k < c) fetch(q[k], p + k);
        };
        auto stash_step = [&](const X3Step<C>& t, const x3_f32x4 (&q)[NKD][C::NPF]) {
            if (!t.live) return;
            int p;
            const int c = new_planes(t, p);
#pragma unroll
            for (int k = 0; k < NKD; ++k) if (k < c) stash(q[k], (wslot + k) % NSLOT);
            wslot = (wslot + c) % NSLOT;
        };
        s_cur.start(dmx, it_first);
        s_n1 = s_cur; s_n1.advance(dmx, it_stride);
        s_n2 = s_n1; s_n2.advance(dmx, it_stride);
        // prologue: the planes of step 0 straight into the ring; steps 1 and 2 into the register queue
        fetch_step(s_cur, pfq[0]);
        fetch_step(s_n1, pfq[1]);
        stash_step(s_cur, pfq[0]);
        fetch_step(s_n2, pfq[0]);
        __syncthreads();
        int tick = 0;
        X3Item done_w = s_cur.w;
        int done_z = -1;
#pragma unroll 1
        while (s_cur.live) {
            // step (tick + 1) sits in buffer ((tick + 1) & 1); that buffer then takes step (tick + 3)
            s_n3 = s_n2; s_n3.advance(dmx, it_stride);
            if (done_z >= 0) epi_open(done_w, done_z);
            if (tick & 1) { stash_step(s_n1, pfq[0]); fetch_step(s_n3, pfq[0]); }
            else          { stash_step(s_n1, pfq[1]); fetch_step(s_n3, pfq[1]); }
            if (done_z >= 0) epi_close((tick + 1) & 1);
            done_w = s_cur.w; done_z = s_cur.z;
            s_cur = s_n1; s_n1 = s_n2; s_n2 = s_n3;
            ++tick;
            __syncthreads();
        }
        if (done_z >= 0) { epi_open(done_w, done_z); epi_close((tick + 1) & 1); }
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------
#define RCMVS_X3_LIST(X) X(8, 8, X3_S1) X(16, 8, X3_S1) X(32, 8, X3_S1) X(16, 16, X3_S1) X(8, 16, X3_S2) X(16, 32, X3_S2) X(16, 8, X3_T2) \
    X(8, 8, X3_P1) X(16, 16, X3_P1) X(32, 32, X3_P1) X(32, 16, X3_P1) X(64, 32, X3_P1) X(32, 32, X3_S1) X(32, 16, X3_T2)

bool conv3d_x3_supported(int Ci, int Co, int kind) {
#define X3_CASE(CI, CO, K) if (Ci == CI && Co == CO && kind == K) return true;
    RCMVS_X3_LIST(X3_CASE)
#undef X3_CASE
    return false;
}

long long conv3d_x3_weight_floats(int Ci, int Co, int kind) {      // size of one x3 image in floats (it is stored as bf16 triples)
#define X3_CASE(CI, CO, K) if (Ci == CI && Co == CO && kind == K) return (long long)X3<CI, CO, K>::KSTEPS * 3 * X3<CI, CO, K>::MT_ALL * 64 * 8 / 2;
    RCMVS_X3_LIST(X3_CASE)
#undef X3_CASE
    return 0;
}

int conv3d_x3_pack(const float* w, float* img, int Co, int Ci, int kind, int transposed, hipStream_t st) {
#define X3_CASE(CI, CO, K) if (Ci == CI && Co == CO && kind == K) { \
        const int nthr = X3<CI, CO, K>::KSTEPS * X3<CI, CO, K>::MT_ALL * 64 * 8; \
        hipLaunchKernelGGL((x3_pack_kernel<CI, CO, K>), dim3((nthr + 255) / 256), dim3(256), 0, st, w, reinterpret_cast<unsigned short*>(img), transposed); \
        return launch_status("conv3d_x3_pack"); }
    RCMVS_X3_LIST(X3_CASE)
#undef X3_CASE
    return fail(-1, "conv3d_x3_pack: unsupported Ci=%d Co=%d kind=%d", Ci, Co, kind);
}

int conv3d_x3_launch(const float* x, const float* wimg, const float* scale, const float* shift, const float* res, float* y,
                     int B, int D, int H, int W, int Ci, int Co, int kind, int relu, hipStream_t st, int max_blocks, int s2d) {
    if (s2d && (kind != X3_P1 || Ci % 16 != 0)) return fail(-1, "conv3d_x3: the space-to-depth view needs the planar kind and Ci a multiple of 16");
    if ((long long)B * D * H * W * Ci * 4 >= 0x7ffffff0LL) return fail(-1, "conv3d_x3: input tensor too large for 32-bit offsets");
    // per-device facts (a process may drive several GPUs, e.g. nn.DataParallel replicas): CU count, and whether the kernel's
    // dynamic-LDS limit has been raised on that device.  Races are benign (the same values are written).
    constexpr int MAXDEV = 64;
    static int cu_of[MAXDEV];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return fail(-1, "conv3d_x3: cannot query the device");
    if (cu_of[dev] == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return fail(-1, "conv3d_x3: cannot query the device");
        cu_of[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int n_cu = cu_of[dev];
    const int n_blk = max_blocks > 0 ? max_blocks : n_cu;      // max_blocks: test / tuning hook (few blocks = many items per block)
    X3Dims dm;
    dm.B = B; dm.D = D; dm.H = H; dm.W = W; dm.relu = relu; dm.s2d = s2d;
    if (kind == X3_T2) { dm.Do = 2 * D; dm.Ho = 2 * H; dm.Wo = 2 * W; }
    else { const int s = kind == X3_S2 ? 2 : 1; dm.Do = (D - 1) / s + 1; dm.Ho = (H - 1) / s + 1; dm.Wo = (W - 1) / s + 1; }
    const int gh = kind == X3_T2 ? H : dm.Ho, gw = kind == X3_T2 ? W : dm.Wo;      // tile grid
    dm.Dt = kind == X3_T2 ? D : dm.Do;
#define X3_CASE(CI, CO, K) if (Ci == CI && Co == CO && kind == K) { \
        using C = X3<CI, CO, K>; \
        const int tiles_x = (gw + C::TX - 1) / C::TX, tiles_y = (gh + C::TY - 1) / C::TY; \
        dm.tiles_x = tiles_x; dm.ntiles = tiles_x * tiles_y; \
        /* one persistent block per CU walks its share of the (batch, tile, z chunk) items: pick the chunk length that minimises the \
           longest block (items per block x (steps per item + ~1.5 steps for the extra planes an item start loads)) */ \
        long long best = -1; int zchunk = dm.Dt; \
        for (int zc = dm.Dt; zc >= (C::NKD > 1 ? 2 : 1); --zc) { \
            const int nch = (dm.Dt + zc - 1) / zc; \
            if (nch > 1 && (dm.Dt + nch - 1) / nch != zc) continue;            /* only balanced splits */ \
            const long long items = (long long)B * dm.ntiles * nch; \
            const long long per_blk = (items + n_blk - 1) / n_blk; \
            const long long cost = per_blk * (2 * zc + (C::NKD > 1 ? 3 : 0)); \
            if (best < 0 || cost < best) { best = cost; zchunk = zc; } \
        } \
        dm.zchunk = zchunk; dm.nchunks = (dm.Dt + zchunk - 1) / zchunk; \
        const long long items = (long long)B * dm.ntiles * dm.nchunks; \
        if (items >= 0x7fffffffLL) return fail(-1, "conv3d_x3: too many work items"); \
        dm.nitems = (int)items; \
        dim3 grid((unsigned)(dm.nitems < n_blk ? dm.nitems : n_blk)); \
        static bool attr_set[MAXDEV]; \
        if (!attr_set[dev]) { (void)hipFuncSetAttribute((const void*)conv3d_x3_kernel<CI, CO, K>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDSB); attr_set[dev] = true; } \
        hipLaunchKernelGGL((conv3d_x3_kernel<CI, CO, K>), grid, dim3(512), C::LDSB, st, x, reinterpret_cast<const x3_u32x4*>(wimg), scale, shift, res, y, dm); \
        return launch_status("conv3d_x3"); }
    RCMVS_X3_LIST(X3_CASE)
#undef X3_CASE
    return fail(-1, "conv3d_x3: unsupported Ci=%d Co=%d kind=%d", Ci, Co, kind);
}

}  // namespace rcmvs
