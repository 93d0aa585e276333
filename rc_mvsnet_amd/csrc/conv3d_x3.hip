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
// Second arithmetic (round 3, NP = 2): TWO fp16 pieces per operand after an exact power-of-two pre-scale, THREE MFMAs per product.
//   The caller hands over an upper bound of max|x| (a device scalar that the producing kernel's epilogue maintained with one
//   atomic max per wave: `ymax` below); s = 2^e puts that bound into [2^14, 2^15).  xs = s x (exact), h = fp16(xs) and
//   l = fp16(xs - h), both round-to-nearest-even: |xs - h - l| <= 2^-22 |xs| while l is a normal fp16 number, i.e. for
//   |x| >= 2^-17 max|x|, and <= 2^-25 in scaled units (2^-39 max|x|) below that.  Weights are split the same way at pack time with
//   their own scale.  Product = xh wh + (xh wl + xl wh) on v_mfma_f32_16x16x32_f16 (the products are exact, 11 x 11 bits; the
//   dropped xl wl is <= 2^-22 |x||w|), two magnitude classes in separate fp32 accumulators, un-scaled by the exact factor
//   2^-(e_x + e_w) folded into the BatchNorm scale of the epilogue.  Half the matrix-pipe work of the three-piece form, a 12-
//   instead of 22-instruction split per float4, two thirds of the LDS ring.  Accuracy against fp64: tests/test_gpu_parity.py::
//   test_conv3d_x3h_*.  Since the end of round 3 this is what the CostRegNet layers of a B = 1 inference scene run on (casmvsnet.py:
//   bound of the variance volume from the feature maps, every layer keeps the bound of its stored outputs: -4.3 % per scene at unchanged
//   depth parity).  The three-piece bf16 form (NP = 3: exact split, no scale, no bound needed) serves callers without a bound: batches,
//   the planar FeatureNet layers, training (forward and data gradients), stand-alone calls, RCMVS_FP16_PAIR=0.
// What bounds it (measured): on gfx950 VALU work -- of another wave on the SIMD or interleaved in the same
// wave -- does not overlap v_mfma_f32_16x16x32_bf16, so a step costs MFMA time + producer VALU time; the producers are therefore
// kept to the bare split (22 VALU per float4) and table-driven addressing.
#include "common.h"
#include "x3_pieces.h"
#include <atomic>
#include <cstdlib>

#ifndef X3_ABLATION
#define X3_ABLATION 0       // tools/dev/x3_test.hip builds with 1: the phase-ablation switches of X3Dims::dbg (timing experiments only)
#endif

namespace rcmvs {

#if X3_ABLATION
int x3_ablation_mask = 0;   // bit 0: no MFMAs, 1: no split / ring stores, 2: no input loads, 3: no output stores, 4: no B-fragment reads
long long* x3_trace_buf = nullptr;      // device buffer [64 ticks][8 stamps] of s_memtime values of block 0 (consumer wave 0: 0-2, producer wave NCW: 4-7)
#define X3_STAMP(SLOT) do { if (dm.trace && blockIdx.x == 0 && lane == 0 && s < 64) dm.trace[s * 8 + (SLOT)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define X3_STAMP(SLOT) do { } while (0)
#endif
#define X3_DBG(BIT) (X3_ABLATION == 1 && ((dm.dbg >> (BIT)) & 1))       // (X3_ABLATION == 2: time stamps only, no switches)


// stride-1 conv, stride-2 conv, transposed stride-2 conv, planar stride-1 conv (kd = 1 taps only: a 3x3 conv of every z-plane,
// what a 3x3x3 conv of a one-plane volume reduces to -- the FeatureNet layers, models/modules.py:413-424)
enum { X3_S1 = 0, X3_S2 = 1, X3_T2 = 2, X3_P1 = 3, X3_KINDS = 4 };
__host__ __device__ constexpr bool x3_unit(int kind) { return kind == X3_S1 || kind == X3_P1; }       // stride-1 geometry in the plane
enum { X3_XT = 0, X3_YT = 1, X3_PL = 2 };        // how an n-tile maps to voxels (see above)

template <int CIN, int COUT, int KIND, int NP = 3>
struct X3 {
    static constexpr int NPIECE = NP;                                    // 3 = bf16 pieces (exact split), 2 = scaled fp16 pieces
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
    static constexpr int WREG = KSTEPS * MT_ALL * 4 * NP;                // registers the whole weight image would take
    // Transposed kind: output parity class (pd, ph, pw) only reaches the input cells nb <= p per axis (tap p - 2 nb + 1 must be 0..2), so
    // whole 16 x 32 weight fragments are zero: bit (j * MT_ALL + mt) of LIVEMASK = fragment (K step j, m-tile mt) holds a tap.
    // Computed with the pack kernel's own index arithmetic.
    static constexpr bool frag_live(int j, int mt) {
        if (KIND != X3_T2) return true;
        const int kd = j / SPK, js = j % SPK;
        for (int lane = 0; lane < 64; ++lane) {
            const int m = mt * 16 + (lane & 15), kk = lane >> 4;
            const int q = (HALVES == 1) ? js * PPS + kk / (4 / PPS) : js / HALVES;
            if (q >= PPKD) continue;
            const int r = q / QC, c = q % QC, p = m / COUT;
            const int td = ((p >> 2) & 1) - 2 * kd + 1, th = ((p >> 1) & 1) - 2 * r + 1, tw = (p & 1) - 2 * c + 1;
            if (td >= 0 && td < 3 && th >= 0 && th < 3 && tw >= 0 && tw < 3) return true;
        }
        return false;
    }
    static constexpr unsigned long long live_mask() {
        unsigned long long m = 0;
        for (int j = 0; j < KSTEPS; ++j)
            for (int mt = 0; mt < MT_ALL; ++mt)
                if (KSTEPS * MT_ALL > 64 || frag_live(j, mt)) m |= 1ULL << ((j * MT_ALL + mt) & 63);
        return m;
    }
    static constexpr int popcount64(unsigned long long v) { int n = 0; for (; v; v &= v - 1) ++n; return n; }
    static constexpr unsigned long long LIVEMASK = live_mask();
    static constexpr int NLIVE = (KSTEPS * MT_ALL > 64) ? KSTEPS * MT_ALL : popcount64(LIVEMASK);
    // the zero fragments are skipped (no registers, no MFMAs) where the live ones fit ONE wave's budget: every consumer wave then
    // holds the same compile-time pattern (a K / M cut would make the pattern depend on the wave: branches around the MFMAs)
    static constexpr bool SKIP = (KIND == X3_T2) && NLIVE < KSTEPS * MT_ALL && NLIVE * 4 * NP <= 128;
    static constexpr bool live(int j, int mt) { return !SKIP || ((LIVEMASK >> (j * MT_ALL + mt)) & 1ULL); }
    // wave roles: 4 consumer + 4 producer waves; layers whose weight image exceeds 4 x 128 registers take 6 consumer waves (K cut
    // three ways, M two ways) and 2 producer waves -- their volumes are small and the producers have little to do
    static constexpr int NCW = (KSTEPS * MT_ALL * 12 > 512) ? 6 : 4;     // (decided on the three-piece image: both forms share the tile geometry)
    static constexpr int NPW = 8 - NCW;                                   // (two consumer + six producer waves on the two-piece 16 -> 8 / 8 -> 8 layers: 117.6 -> 136-141 us, 58.6 -> 70 us: not adopted)
    static constexpr int MSPLIT = (MT_ALL >= 2 && WREG > 128 && !SKIP) ? 2 : 1;
    static constexpr int KSPLIT = SKIP ? 1 : ((NCW == 6) ? 3 : ((WREG / MSPLIT > 256) ? 4 / MSPLIT : ((WREG / MSPLIT > 128) ? 2 : 1)));
    static constexpr int MT = MT_ALL / MSPLIT;                           // m-tiles per consumer wave
    static constexpr int KSW = (KSTEPS + KSPLIT - 1) / KSPLIT;          // K steps per consumer wave
    static constexpr int TX = ((x3_unit(KIND) && CIN >= 32) || (KIND == X3_S2 && CIN >= 16) || NCW == 6) ? 16 : 32;
#ifndef X3_P1_TY32
#define X3_P1_TY32 8
#endif
    static constexpr int TY = (MAP == X3_XT) ? 8 : ((KIND == X3_P1 && CIN == 32 && NCW == 4) ? X3_P1_TY32 : ((KIND == X3_S2 || NCW == 6) ? 2 : 4));      // (TY = 6 on the two-piece form, whose smaller ring leaves the LDS for it: halo 1.59x -> 1.42x, 32 -> 8 89.0 -> 85.1 us, 16 -> 8 116.4 -> 115.0: not worth a third tile geometry)
    static constexpr int CS = (MAP == X3_XT || KIND == X3_S2) ? 2 : 1;   // voxels between neighbouring columns
    static constexpr int RS = (MAP == X3_YT || KIND == X3_S2) ? 2 : 1;   // halo rows between neighbouring tile rows
    static constexpr int TYP = x3_unit(KIND) ? TY + 2 : (KIND == X3_S2 ? 2 * TY + 1 : TY + 1);
    static constexpr int TXP = x3_unit(KIND) ? TX + 2 : (KIND == X3_S2 ? 2 * TX + 1 : TX + 1);
    static constexpr int ROWB = TXP * VB;                                // bytes per halo row
    static constexpr int PLB = TYP * ROWB;                               // bytes per piece plane
    static constexpr int SLB = NP * PLB;                                 // bytes per z-slice (h, m, l planes / h, l planes)
    static constexpr int ZADV = (KIND == X3_S2) ? 2 : 1;                 // input slices consumed per step
    static constexpr int NSLOT = 2 * NKD;                                // ring: the planes being read + the ones being written (up to NKD when the next item starts)
    static constexpr int NTX = (MAP == X3_XT) ? TX / 32 : TX / 16;       // n-tiles along x
    static constexpr int NTILE = ((MAP == X3_YT) ? TY / 2 : TY) * NTX;
    static constexpr int NG = NCW / (KSPLIT * MSPLIT);                     // consumer waves that own different n-tiles
    static constexpr int NTW = NTILE / NG;                               // n-tiles per consumer wave
    static constexpr int TP = (MT <= 2 && NTW % 2 == 0) ? 2 : 1;         // n-tiles in flight (independent accumulators)
    // B-fragment prefetch distance in K steps.  A consumer wave is alone on its SIMD, so the ~300 cycles between a ds_read_b128 and
    // its data (four waves' bursts queue in the LDS) must be covered by the MFMAs of the K steps in between: 6 (4, 2) MFMAs of 17
    // cycles per n-tile and m-tile and K step.  One step ahead (round 2) left every K step waiting on the LDS -- which is why
    // halving the MFMA count (NP = 2) changed nothing (profiles/r3_x3_prefetch_distance.txt).  Bounded by the register budget.
    // who finishes the K-split tiles (sum of the partial tiles, BN, ReLU, skip-add, store): with three pieces the consumers' MFMA
    // phase is the longer half of a tick and the producers have the slack; with two pieces it is the other way round
    // (profiles/r3_x3_tick_trace.txt)
    static constexpr bool EPI_CONSUMER = (NP == 2);
    static constexpr int NEW = EPI_CONSUMER ? NCW : NPW;                 // waves the (tile, m-tile) units are dealt to
    static constexpr int MFMA_PER_KSTEP = (SKIP ? (TP * NLIVE + KSTEPS - 1) / KSTEPS : TP * MT) * (NP == 3 ? 6 : 3);      // (average over the K steps when fragments are skipped)
    static constexpr int PD_WANT = (MFMA_PER_KSTEP >= 20) ? 1 : ((MFMA_PER_KSTEP >= 10) ? 2 : ((MFMA_PER_KSTEP >= 6) ? 3 : 4));
    static constexpr int REG_FIXED = (SKIP ? NLIVE * NP * 4 : ((KSTEPS + KSPLIT - 1) / KSPLIT) * NP * MT * 4) + TP * MT * NP * 4 + 28;      // weights + accumulators + the rest
    static constexpr int PD_FIT = (244 - REG_FIXED) / (TP * NP * 4) - 1;
    static constexpr int PD_CAP = PD_WANT < PD_FIT ? PD_WANT : PD_FIT;
    static constexpr int PD = PD_CAP < 1 ? 1 : (PD_CAP > (KSTEPS + KSPLIT - 1) / KSPLIT - 1 && (KSTEPS + KSPLIT - 1) / KSPLIT > 1 ? (KSTEPS + KSPLIT - 1) / KSPLIT - 1 : PD_CAP);
    static constexpr int Q4 = CIN / 4;                                   // float4 per voxel
    static constexpr int NLOAD = TYP * TXP * Q4;                         // float4 per z-slice
    static constexpr int NPF = (NLOAD + NPW * 64 - 1) / (NPW * 64);                      // float4 per producer thread per z-slice
    static constexpr int PARTB = (KSPLIT > 1) ? NTILE * MT_ALL * KSPLIT * 1024 : 0;   // one buffer of partial output tiles
    static constexpr int LDSB = NSLOT * SLB + 2 * PARTB;
    static_assert(NCW % (KSPLIT * MSPLIT) == 0 && (SKIP ? NLIVE * 4 * NP : WREG / (KSPLIT * MSPLIT)) <= 128, "weight slice per wave");
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
// NP = 2: image = a 16-byte header {s_w, 1 / s_w, 0, 0} (s_w = the power of two that puts max|w| into [2^14, 2^15), read from
// `wsc`, which x3_wscale_kernel filled) followed by the fp16 pieces of s_w * w in the same fragment order.
template <int CIN, int COUT, int KIND, int NP>
__global__ void x3_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ img, int transposed, const float* __restrict__ wsc) {
    using C = X3<CIN, COUT, KIND, NP>;
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
    if constexpr (NP == 3) {
        const unsigned hb = __float_as_uint(v) & 0xffff0000u;
        const float r1 = v - __uint_as_float(hb);
        const unsigned mb = __float_as_uint(r1) & 0xffff0000u;
        const float r2 = r1 - __uint_as_float(mb);
        const unsigned lb = __float_as_uint(r2) & 0xffff0000u;
        const long long base = (((long long)j * 3) * C::MT_ALL + mt) * 512 + lane * 8 + e;     // piece stride = MT_ALL * 512 shorts
        img[base] = (unsigned short)(hb >> 16);
        img[base + (long long)C::MT_ALL * 512] = (unsigned short)(mb >> 16);
        img[base + 2LL * C::MT_ALL * 512] = (unsigned short)(lb >> 16);
    } else {
        const float sw = wsc[0];
        if (t == 0) { float* hdr = reinterpret_cast<float*>(img); hdr[0] = sw; hdr[1] = wsc[1]; hdr[2] = 0.0f; hdr[3] = 0.0f; }
        const float vs = v * sw;                                       // exact (power of two)
        const _Float16 h = (_Float16)vs;                               // round to nearest even
        const _Float16 l = (_Float16)(vs - (float)h);
        const long long base = 8 + (((long long)j * 2) * C::MT_ALL + mt) * 512 + lane * 8 + e;
        img[base] = __builtin_bit_cast(unsigned short, h);
        img[base + (long long)C::MT_ALL * 512] = __builtin_bit_cast(unsigned short, l);
    }
}

// one block: out[0] = scale of max|w| over n weights, out[1] = its inverse
__global__ void x3_wscale_kernel(const float* __restrict__ w, int n, float* __restrict__ out) {
    __shared__ float red[256];
    float m = 0.0f;
    for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(w[i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + k]);
        __syncthreads();
    }
    if (threadIdx.x == 0) { float inv; out[0] = x3_pow2_scale(red[0], inv); out[1] = inv; }
}

struct X3Dims {
    int D, H, W;        // input volume
    int Do, Ho, Wo;     // output volume
    int Dt;             // extent of the tile grid in z (= Do for the convolutions, D for the transposed convolution)
    int tiles_x, zchunk, relu;
    int B, ntiles, nchunks, nitems;   // work items = B x xy tiles x z chunks
    int itemcap, stepcap;   // capacities of the block's schedule tables in LDS (items / steps per block, rounded up)
    int bal;                // 1 = balanced schedule: block r walks steps [T r / n, T (r + 1) / n) of the flattened (batch, tile, z) sequence (see the kernel)
    int dbg;            // phase-ablation mask (0 in the library; X3_ABLATION builds only)
    long long* trace;   // s_memtime stamps of block 0 (nullptr in the library; X3_ABLATION builds only)
    int place;          // wave placement of the 4 + 4 forms (see the kernel): 0 = one consumer and one producer per SIMD, 1 = consumers on SIMDs 0-1, producers on 2-3
    int ysq;            // ymax receives the SQUARE of max|y| (the bound of a variance volume from the bound of its samples: FeatureNet's output convs)
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
// The block's schedule lives in two LDS tables, built once by all threads (round 3: the tick loops used to advance four copies of
// an item iterator with its integer divisions and branches -- ~500 mostly scalar instructions per tick in the producer waves,
// which is what bounded every layer: profiles/r3_x3_phase_ablation.txt):
//   itab[k] = {b, x0, y0, zb | ze << 16}     item k of this block (global item it_first + k it_stride)
//   tab[s]  = k << 16 | first << 15 | z      step s: item, first step of its item?, z of the tile grid
// A tick reads the descriptors of the steps it touches (s - 1: epilogue, s: compute, s + 1: ring stores, s + 3: loads).
struct X3Sched {
    const int4* itab;
    const int* tab;
    int nsteps;
    __device__ __forceinline__ int desc(int s) const { return (s >= 0 && s < nsteps) ? __builtin_amdgcn_readfirstlane(tab[s]) : -1; }
};
__device__ __forceinline__ int x3_desc_item(int d) { return d >> 16; }
__device__ __forceinline__ bool x3_desc_first(int d) { return (d >> 15) & 1; }
__device__ __forceinline__ int x3_desc_z(int d) { return d & 0x7fff; }

// xmax (NP = 2 only): bound of max|x| -- 64 slots, 16 floats apart, the bound is their maximum (the scale of the activation split is
// derived from it); ymax (optional, both forms): the same structure for max|y| over the stored outputs, maintained with ONE atomic
// max per block into slot (block & 63) (same-address atomics serialise at ~90 per microsecond: one per wave into one word cost
// 6-20 us per launch) -- the next layer's xmax.  The caller zero-fills it; |y| as an IEEE bit pattern orders like an unsigned integer.
template <int CIN, int COUT, int KIND, int NP>
__global__ __launch_bounds__(512) void conv3d_x3_kernel(
    const float* __restrict__ x, const x3_u32x4* __restrict__ wimg, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ res, float* __restrict__ y, X3Dims dm,
    const float* __restrict__ xmax, float* __restrict__ ymax) {
    using C = X3<CIN, COUT, KIND, NP>;
    constexpr int MT = C::MT, KSW = C::KSW, KSPLIT = C::KSPLIT, TP = C::TP, NSLOT = C::NSLOT, NKD = C::NKD, ZADV = C::ZADV;
    extern __shared__ __attribute__((aligned(16))) x3_byte smem[];
    x3_byte* const partbase = smem + NSLOT * C::SLB;
    int4* const itab_w = reinterpret_cast<int4*>(partbase + 2 * C::PARTB);
    int* const tab_w = reinterpret_cast<int*>(itab_w + dm.itemcap);
    float* const redmax = reinterpret_cast<float*>(tab_w + dm.stepcap);        // 8 floats: the waves' output maxima
    const int lane = threadIdx.x & 63;
    // Wave placement (dm.place, 4 + 4 forms): hardware wave h runs on SIMD h & 3, and VALU work does not overlap MFMAs issued on the same
    // SIMD -- not another wave's either.  With roles dealt by hardware id (consumers 0-3, producers 4-7) every SIMD hosts one of each and
    // a tick is MFMA time PLUS producer time.  place = 1 deals the logical ids so that consumers sit on SIMDs 0 and 1 (two each) and
    // producers on SIMDs 2 and 3: the MFMA phase takes twice as long and the producers run beside it.
    const int hwave = threadIdx.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane((dm.place && C::NCW == 4) ? ((hwave & 1) | ((hwave & 4) >> 1) | ((hwave & 2) << 1)) : hwave);
    const int tid = wave * 64 + lane;
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
    const bool relu = dm.relu != 0;
    // ---- schedule tables
    int nloc;
    if (dm.bal) {
        // Balanced schedule (round 4).  With (batch, tile, z chunk) items dealt to the blocks, a block's work is a whole number of items:
        // the mid-size layers have 1-5 items per block and the longest block sets the launch time (16 -> 16 at 16 x 128 x 160: 160 tiles,
        // 3 items of 4 steps on some blocks = 16.5 step-equivalents where the average is 10).  Here the steps of ALL tiles form one
        // sequence (tile-major, z inside) and block r takes the contiguous range [T r / n, T (r + 1) / n): every block gets the same
        // number of steps (+- 1), a block's first and last item are partial tiles, and the ring runs across the item boundaries as
        // before (an item is any z range of a tile).  Cuts that would leave a one-step fragment at a tile's start or end move by one.
        const long long T = (long long)dm.B * dm.ntiles * dm.Dt;
        const int r = ((nblk & 7) == 0) ? (bid & 7) * (nblk >> 3) + (bid >> 3) : bid;      // an XCD's blocks (bid & 7) take neighbouring ranges
        auto cut = [&](int i) -> long long {
            long long c = T * i / nblk;
            if (dm.Dt >= 4 && i > 0 && i < nblk) {
                const int m = (int)(c % dm.Dt);
                if (m == 1) c -= 1; else if (m == dm.Dt - 1) c += 1;
            }
            return c;
        };
        const long long lo = cut(r), hi = cut(r + 1);
        if (hi <= lo) return;
        const int t0 = (int)(lo / dm.Dt), t1 = (int)((hi - 1) / dm.Dt);
        nloc = t1 - t0 + 1;
        for (int k = tid; k < nloc; k += 512) {
            const int t = t0 + k;
            const int zb = (k == 0) ? (int)(lo - (long long)t0 * dm.Dt) : 0;
            const int ze = (t == t1) ? (int)(hi - (long long)t1 * dm.Dt) : dm.Dt;
            const int tile = t % dm.ntiles;
            itab_w[k] = make_int4(t / dm.ntiles, (tile % dm.tiles_x) * C::TX, (tile / dm.tiles_x) * C::TY, zb | (ze << 16));
        }
    } else {
        if (it_first >= it_limit) return;            // (more blocks than items in this XCD's range)
        nloc = (it_limit - it_first + it_stride - 1) / it_stride;
        for (int k = tid; k < nloc; k += 512) {
            const X3Item w = x3_item<C>(dm, it_first + k * it_stride);
            itab_w[k] = make_int4(w.b, w.x0, w.y0, w.zb | (w.ze << 16));
        }
    }
    __syncthreads();
    X3Sched sch;
    sch.itab = itab_w; sch.tab = tab_w;
    {
        int total = 0, mine = tid;      // this thread fills steps tid, tid + 512, ...
        for (int k = 0; k < nloc; ++k) {
            const int zz = itab_w[k].w, zb = zz & 0xffff, len = (zz >> 16) - zb;
            while (mine < total + len) {           // step `mine` belongs to item k
                tab_w[mine] = (k << 16) | ((mine == total) ? (1 << 15) : 0) | (zb + (mine - total));
                mine += 512;
            }
            total += len;
        }
        sch.nsteps = total;
    }
    __syncthreads();
    const int nsteps = sch.nsteps;
    // NP = 2: activation scale from the caller's bound, weight scale from the image header; `unscale` = 2^-(e_x + e_w) is exact
    float xs_scale = 1.0f, unscale = 1.0f;
    if constexpr (NP == 2) {
        float bound = xmax[lane * 16];
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) bound = fmaxf(bound, __shfl_xor(bound, m));
        float xinv;
        xs_scale = x3_pow2_scale(bound, xinv);
        unscale = xinv * reinterpret_cast<const float*>(wimg)[1];
        wimg += 1;                                // skip the 16-byte header
    }
    float vmax = 0.0f;                            // max |stored output| seen by this thread (ymax)

    constexpr int OOB = 0x7ffffff0;
    const int ew = max(C::EPI_CONSUMER ? wave : wave - C::NCW, 0);       // index among the waves that finish tiles (the others never call epi_*)
    // Finishing the K-split tiles of a step (sum of the partial tiles, BN scale/shift, ReLU, skip-add, store) falls to the waves with
    // the slack in a tick (C::EPI_CONSUMER): the producers in the three-piece form (as in round 2); the consumers in the two-piece
    // form, whose MFMA phase is half as long -- a consumer then waits at the tick barrier for the producers anyway, whose ring stores
    // cannot overlap the MFMAs of the wave they share a SIMD with (profiles/r3_x3_tick_trace.txt).  epi_open works out where this wave's (tile, m-tile) units go and issues the skip-connection
    // loads, epi_close does the arithmetic.  Inside an item the output offsets just advance by one z step of the tile grid; they are
    // recomputed when the item changes.  Output and skip tensor go through buffer descriptors with 32-bit byte offsets (host-checked
    // < 2^31): a unit outside the volume, a missing skip tensor and a step that does not exist are out-of-range offsets.
    constexpr int NEU = (KSPLIT > 1) ? (C::NTILE * C::MT_ALL + C::NEW - 1) / C::NEW : 1;
    x3_f32x4 esc[NEU], esh[NEU], erv[NEU];
    int eob[NEU];                  // byte offset of the unit's float4 in y / res; OOB = nothing to store
    int epi_item = -1;
    const int ostep = (KIND == X3_T2 ? 2 : 1) * dm.Ho * dm.Wo * COUT * 4;
    const int ybytes = (int)((long long)dm.B * dm.Do * dm.Ho * dm.Wo * COUT * 4);
    __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(y, (short)0, ybytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(res ? res : y), (short)0, res ? ybytes : 0, 0x00020000);
    if constexpr (KSPLIT > 1) {
#pragma unroll
        for (int i = 0; i < NEU; ++i) {
            const int u = min(ew + C::NEW * i, C::NTILE * C::MT_ALL - 1);
            long long ov; int co0;
            x3_out_coord<C, COUT, KIND>(dm, 0, 0, 0, 0, 0, u % C::MT_ALL, n, kk, ov, co0);      // co0 depends on the m-tile and the lane only
            esc[i] = (scale ? *reinterpret_cast<const x3_f32x4*>(scale + co0) : (x3_f32x4){1.f, 1.f, 1.f, 1.f}) * unscale;
            esh[i] = shift ? *reinterpret_cast<const x3_f32x4*>(shift + co0) : (x3_f32x4){0.f, 0.f, 0.f, 0.f};
            eob[i] = OOB;
        }
    }
    auto epi_open = [&](int d) {
        if constexpr (KSPLIT > 1) {
            if (d >= 0) {
                const int k = x3_desc_item(d);
                if (k == epi_item) {
#pragma unroll
                    for (int i = 0; i < NEU; ++i) if (eob[i] != OOB) eob[i] += ostep;
                } else {
                    epi_item = k;
                    const int4 w = sch.itab[k];
                    const int z = x3_desc_z(d);
#pragma unroll
                    for (int i = 0; i < NEU; ++i) {
                        const int u = ew + C::NEW * i;                  // (tile, m-tile) unit of this wave
                        long long ov; int co0;
                        const bool ok = u < C::NTILE * C::MT_ALL && x3_out_coord<C, COUT, KIND>(dm, w.x, w.y, w.z, z, u / C::MT_ALL, u % C::MT_ALL, n, kk, ov, co0);
                        eob[i] = ok ? (int)((ov * COUT + co0) * 4) : OOB;
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < NEU; ++i) eob[i] = OOB;
            }
#pragma unroll
            for (int i = 0; i < NEU; ++i)          // (no skip tensor: a zero-length descriptor, every lane out of range)
                erv[i] = __builtin_bit_cast(x3_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrs, eob[i], 0, 0));
        }
    };
    auto epi_close = [&](int buf) {
        if constexpr (KSPLIT > 1) {
            const x3_f32x4* part = reinterpret_cast<const x3_f32x4*>(partbase + buf * C::PARTB);
#pragma unroll
            for (int i = 0; i < NEU; ++i) {
                const x3_f32x4* pp = part + min(ew + C::NEW * i, C::NTILE * C::MT_ALL - 1) * KSPLIT * 64 + lane;
                x3_f32x4 v = pp[0];
#pragma unroll
                for (int k = 1; k < KSPLIT; ++k) v += pp[k * 64];
                v = v * esc[i] + esh[i];
                if (relu) v = __builtin_elementwise_max(v, (x3_f32x4){0.f, 0.f, 0.f, 0.f});
                v += erv[i];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(x3_u32x4, v), yrs, X3_DBG(3) ? OOB : eob[i], 0, 0);
                if (ymax && eob[i] != OOB) vmax = x3_absmax4(vmax, v);
            }
        }
    };
    if (!producer) {
        // =============================== consumer: register-stationary weights of this wave's (K, M) slice
        const int ks = wave % KSPLIT, ms = (wave / KSPLIT) % C::MSPLIT, grp = wave / (KSPLIT * C::MSPLIT);
        x3_u32x4 wr[KSW][NP][MT];          // raw fragment bits (bf16 or fp16 pieces)
        int kdj[KSW];      // input plane of K step j (wave-uniform)
        int boff[KSW];     // this lane's byte offset of the B fragment inside a z-slice, tile origin excluded
#pragma unroll
        for (int j = 0; j < KSW; ++j) {
            const int jg = ks * KSW + j;
            const bool live = jg < C::KSTEPS;
            const int jc = live ? jg : 0;
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    x3_u32x4 v = (x3_u32x4){0u, 0u, 0u, 0u};
                    if (C::live(C::SKIP ? j : 0, C::SKIP ? mt : 0)) v = wimg[((jc * NP + p) * C::MT_ALL + ms * MT + mt) * 64 + lane];      // (SKIP: one K / M slice, j and mt are the global indices)
                    if (!live) v = (x3_u32x4){0u, 0u, 0u, 0u};
                    wr[j][p][mt] = v;
                }
            kdj[j] = jc / C::SPK;
            int q, ci0;
            x3_kslot<C>(jc % C::SPK, kk, q, ci0);
            if (q >= C::PPKD) q = 0;
            const int hc = q % C::QC + n * C::CS;      // halo column of this lane's voxel (tile column offsets are multiples of 16: same swizzle)
            boff[j] = (q / C::QC) * C::ROWB + hc * C::VB + ((ci0 * 2) ^ x3_swz<C, CIN, KIND>(hc));
        }
        // epilogue constants (used when this wave finishes its own tiles: KSPLIT == 1).  Transposed kind: an m-tile is 16 / COUT parity
        // classes, every tile sees the same channels
        constexpr int SCN = (KIND == X3_T2 && 16 % COUT == 0) ? 1 : MT;
        x3_f32x4 sc[SCN], sh[SCN];
        if constexpr (KSPLIT == 1) {
#pragma unroll
            for (int mt = 0; mt < SCN; ++mt) {
                long long ov; int co0;
                x3_out_coord<C, COUT, KIND>(dm, 0, 0, 0, 0, 0, ms * MT + mt, n, kk, ov, co0);
                sc[mt] = (scale ? *reinterpret_cast<const x3_f32x4*>(scale + co0) : (x3_f32x4){1.f, 1.f, 1.f, 1.f}) * unscale;
                sh[mt] = shift ? *reinterpret_cast<const x3_f32x4*>(shift + co0) : (x3_f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        // tile offsets inside a slice and tile ids of this wave: step-invariant
        int toffv[C::NTW], tlv[C::NTW];
#pragma unroll
        for (int i = 0; i < C::NTW; ++i) {
            tlv[i] = grp + C::NG * i;
            toffv[i] = (tlv[i] / C::NTX) * C::RS * C::ROWB + (tlv[i] % C::NTX) * 16 * C::CS * C::VB;
        }
        // KSPLIT == 1: this wave stores its own tiles.  Byte offset of the lane's float4 per (n-tile, m-tile) in y / the skip tensor
        // (OOB outside the volume): worked out on the first step of an item, advanced by one z step of the tile grid afterwards
        // (round 2 recomputed the 64-bit voxel index for every tile of every step: ~50 VALU per float4 -- on the transposed layer, 8 float4
        // per lane and step, the epilogue was a quarter of the tick: profiles/r3_x3_tick_trace.txt)
        int cob[KSPLIT == 1 ? C::NTW : 1][MT];
        __syncthreads();          // the planes of the first step are in the ring
        int s0 = 0;               // ring slot of the first input plane of the current step
#pragma unroll 1
        for (int s = 0; s < nsteps; ++s) {
            if (wave == 0) X3_STAMP(0);
            const int d = sch.desc(s);
            if constexpr (KSPLIT > 1 && C::EPI_CONSUMER) { if (s > 0) epi_close((s + 1) & 1); }      // finish step s - 1 (its partial tiles were complete at the barrier, its skip loads in flight since before it)
            const int4 itm = sch.itab[x3_desc_item(d)];
            const int z = x3_desc_z(d), x0 = itm.y, y0 = itm.z, b = itm.x;
            int slotoff[NKD];
#pragma unroll
            for (int k = 0; k < NKD; ++k) slotoff[k] = ((s0 + k) % NSLOT) * C::SLB;
            if constexpr (KSPLIT == 1) {
                if (x3_desc_first(d)) {
#pragma unroll
                    for (int i = 0; i < C::NTW; ++i)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            long long ov; int co0;
                            const bool ok = x3_out_coord<C, COUT, KIND>(dm, b, x0, y0, z, tlv[i], ms * MT + mt, n, kk, ov, co0);
                            cob[i][mt] = ok ? (int)((ov * COUT + co0) * 4) : OOB;
                        }
                } else {
#pragma unroll
                    for (int i = 0; i < C::NTW; ++i)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) if (cob[i][mt] != OOB) cob[i][mt] += ostep;
                }
            }
#pragma unroll
            for (int tp = 0; tp < C::NTW / TP; ++tp) {
                int toff[TP], tl[TP];
#pragma unroll
                for (int t = 0; t < TP; ++t) { tl[t] = tlv[TP * tp + t]; toff[t] = toffv[TP * tp + t]; }
                // skip-connection values of the tiles this wave finishes itself: loaded before the MFMA phase
                x3_f32x4 rv[TP][MT];
                if constexpr (KSPLIT == 1) {
#pragma unroll
                    for (int t = 0; t < TP; ++t)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)          // (no skip tensor: a zero-length descriptor, zeros)
                            rv[t][mt] = __builtin_bit_cast(x3_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrs, cob[TP * tp + t][mt], 0, 0));
                }
                x3_f32x4 acc[TP][MT][NP];          // one accumulator per magnitude class
#pragma unroll
                for (int t = 0; t < TP; ++t)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int a = 0; a < NP; ++a) acc[t][mt][a] = (x3_f32x4){0.f, 0.f, 0.f, 0.f};
                // software pipeline: the B fragments of K step j+1 are read while the MFMAs of step j run; the MFMA order keeps
                // >= 3 independent instructions between two uses of one accumulator
                constexpr int PD = C::PD;
                x3_u32x4 bq[PD + 1][TP][NP];
#pragma unroll
                for (int jj = 0; jj < PD; ++jj) {
                    if (jj >= KSW) break;
                    const int a0 = boff[jj] + slotoff[kdj[jj]];
#pragma unroll
                    for (int t = 0; t < TP; ++t)
#pragma unroll
                        for (int p = 0; p < NP; ++p) bq[jj][t][p] = *reinterpret_cast<const x3_u32x4*>(smem + a0 + toff[t] + p * C::PLB);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < KSW; ++j) {
                    const int cur = j % (PD + 1), nxt = (j + PD) % (PD + 1);
                    const bool pre = j + PD < KSW;
                    const int na = pre ? boff[j + PD < KSW ? j + PD : 0] + slotoff[kdj[j + PD < KSW ? j + PD : 0]] : 0;
                    // slots: B-fragment reads of K step j+PD, then the MFMAs of one product class of step j; the fences pin
                    // this order (left alone, the scheduler sinks every read to just before its first use and exposes the LDS latency)
#define X3_MF(ACC, WP, BP) _Pragma("unroll") for (int t = 0; t < TP; ++t) _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) \
        if (!X3_DBG(0) && C::live(C::SKIP ? j : 0, C::SKIP ? mt : 0)) acc[t][mt][ACC] = x3_mfma<NP>(wr[j][WP][mt], bq[cur][t][BP], acc[t][mt][ACC])
#define X3_LD(I) if (pre && (I) < TP * NP && !X3_DBG(4)) bq[nxt][((I) / NP) % TP][(I) % NP] = *reinterpret_cast<const x3_u32x4*>(smem + na + toff[((I) / NP) % TP] + ((I) % NP) * C::PLB)
                    if constexpr (NP == 3) {
                        X3_LD(0); X3_MF(2, 0, 2); __builtin_amdgcn_sched_barrier(0);
                        X3_LD(1); X3_MF(1, 0, 1); __builtin_amdgcn_sched_barrier(0);
                        X3_LD(2); X3_MF(0, 0, 0); __builtin_amdgcn_sched_barrier(0);
                        X3_LD(3); X3_MF(2, 1, 1); __builtin_amdgcn_sched_barrier(0);
                        X3_LD(4); X3_MF(1, 1, 0); __builtin_amdgcn_sched_barrier(0);
                        X3_LD(5); X3_MF(2, 2, 0); __builtin_amdgcn_sched_barrier(0);
                    } else {          // three slots: wh xl, wh xh, wl xh; the (up to) four fragment reads of K step j+1 spread over them
                        X3_LD(0); X3_LD(1); X3_MF(1, 0, 1); __builtin_amdgcn_sched_barrier(0);
                        X3_LD(2); X3_MF(0, 0, 0); __builtin_amdgcn_sched_barrier(0);
                        X3_LD(3); X3_MF(1, 1, 0); __builtin_amdgcn_sched_barrier(0);
                    }
#undef X3_MF
#undef X3_LD
                }
                if (wave == 0 && tp == C::NTW / TP - 1) X3_STAMP(1);
#pragma unroll
                for (int t = 0; t < TP; ++t)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        x3_f32x4 v;
                        if constexpr (NP == 3) v = acc[t][mt][0] + (acc[t][mt][1] + acc[t][mt][2]);
                        else v = acc[t][mt][0] + acc[t][mt][1];
                        if constexpr (KSPLIT > 1) {
                            // hand the partial tile to the producers: [tile][m-tile][K slice][lane]
                            x3_f32x4* part = reinterpret_cast<x3_f32x4*>(partbase + (s & 1) * C::PARTB);
                            part[((tl[t] * C::MT_ALL + ms * MT + mt) * KSPLIT + ks) * 64 + lane] = v;
                        } else {
                            v = v * sc[mt % SCN] + sh[mt % SCN];
                            if (relu) v = __builtin_elementwise_max(v, (x3_f32x4){0.f, 0.f, 0.f, 0.f});
                            v += rv[t][mt];
                            const int ob = cob[TP * tp + t][mt];
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(x3_u32x4, v), yrs, X3_DBG(3) ? OOB : ob, 0, 0);
                            if (ymax && ob != OOB) vmax = x3_absmax4(vmax, v);
                        }
                    }
            }
            const int dn = sch.desc(s + 1);
            s0 = (s0 + ((dn >= 0 && x3_desc_first(dn)) ? NKD : ZADV)) % NSLOT;      // a new item starts right behind the last plane of the previous one
            if constexpr (KSPLIT > 1 && C::EPI_CONSUMER) epi_open(d);       // where step s goes + its skip-connection loads: consumed after the barrier
            if (wave == 0) X3_STAMP(2);
            __syncthreads();
        }
        if constexpr (KSPLIT > 1 && C::EPI_CONSUMER) epi_close((nsteps + 1) & 1);
    } else {
        // =============================== producer
        const int pw = wave - C::NCW, ptid = tid - C::NCW * 64;
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
        auto set_item = [&](int k) {
            if (k == goff_item) return;
            goff_item = k;
            const int4 w = sch.itab[k];
            const int wb = w.x, wx0 = w.y, wy0 = w.z;
            const int hy0 = x3_unit(KIND) ? wy0 - 1 : (KIND == X3_S2 ? 2 * wy0 - 1 : wy0);
            const int hx0 = x3_unit(KIND) ? wx0 - 1 : (KIND == X3_S2 ? 2 * wx0 - 1 : wx0);
            const int base = dm.s2d ? ((wb * dm.D * 2 * dm.H + 2 * hy0) * (2 * dm.W) + 2 * hx0) * CIN
                                    : (wb * dm.D * dm.H + hy0) * dm.W * CIN * 4 + hx0 * CIN * 4;
#pragma unroll
            for (int i = 0; i < C::NPF; ++i) {
                const int gy = hy0 + (hrc[i] >> 16), gx = hx0 + (hrc[i] & 0xffff);
                const bool ok = hrc[i] >= 0 && gy >= 0 && gy < dm.H && gx >= 0 && gx < dm.W;
                goff[i] = ok ? base + grel[i] : OOB;
            }
        };
        // A tick of a producer wave issues a FIXED number of vector-memory operations, whatever the step needs: NKD planes of loads
        // (the planes a step does not add, steps that do not exist, lanes outside the volume = out-of-range buffer offsets: the load
        // returns 0 without touching memory), and the tick loop is unrolled by two so that the register queue alternates statically.
        // Only then can the compiler's s_waitcnt bookkeeping -- vmcnt counts in issue order -- wait for the two-tick-old planes with
        // a partial count; with data-dependent counts it fell back to vmcnt(0) in front of every use, i.e. it waited for the loads
        // issued a few hundred cycles earlier (profiles/r3_x3_tick_trace.txt: producer 4400 of a 4600-cycle tick).  (Loading only
        // the ZADV planes every step adds and fetching the leading planes of an item synchronously was measured too: better on the
        // long conv0 items, worse on the short items of the deep levels, 1.503 against 1.474 ms per scene.)
        auto fetch = [&](x3_f32x4 (&pf)[C::NPF], int zi, bool want) {
            const bool zin = want && zi >= 0 && zi < dm.D && !X3_DBG(2);
            const int zoff = zin ? zi * zstride : 0;
#pragma unroll
            for (int i = 0; i < C::NPF; ++i)
                pf[i] = __builtin_bit_cast(x3_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, zin ? goff[i] : OOB, zoff, 0));
        };
        auto stash = [&](const x3_f32x4 (&pf)[C::NPF], int slot) {
            x3_byte* sb = smem + slot * C::SLB;
#pragma unroll
            for (int i = 0; i < C::NPF; ++i) {
                if (loff[i] < 0 || X3_DBG(1)) continue;
                if constexpr (NP == 3) {
                    x3_u32x2 h, m, l;
                    x3_split4(pf[i], h, m, l);
                    *reinterpret_cast<x3_u32x2*>(sb + loff[i]) = h;
                    *reinterpret_cast<x3_u32x2*>(sb + C::PLB + loff[i]) = m;
                    *reinterpret_cast<x3_u32x2*>(sb + 2 * C::PLB + loff[i]) = l;
                } else {
                    x3_u32x2 h, l;
                    x3_split4h(pf[i] * xs_scale, h, l);
                    *reinterpret_cast<x3_u32x2*>(sb + loff[i]) = h;
                    *reinterpret_cast<x3_u32x2*>(sb + C::PLB + loff[i]) = l;
                }
            }
        };
        auto zin0 = [&](int z) { return (KIND == X3_S1) ? z - 1 : (KIND == X3_S2 ? 2 * z - 1 : z); };      // first input plane of step z (planar: the plane itself)
        // schedule: during tick s the consumers compute step s while the producers (1) store into the ring the planes step s+1
        // adds, (2) issue the loads of the planes step s+3 adds -- two ticks of flight time, a load that misses to HBM under
        // load takes longer than one tick -- and (3) finish step s-1.  Register queue: buffer (t & 1) holds step t's planes from
        // tick t-3 until they are stored in tick t-1.  The planes a step adds to the ring: all NKD on the first step of an item,
        // the last ZADV afterwards.
        x3_f32x4 pfq[2][NKD][C::NPF];
        int wslot = 0;
        auto fetch_step = [&](int d, x3_f32x4 (&q)[NKD][C::NPF]) {
            bool first = false;
            int p = 0;
            if (d >= 0) {
                set_item(x3_desc_item(d));
                first = x3_desc_first(d);
                p = first ? zin0(x3_desc_z(d)) : zin0(x3_desc_z(d)) + NKD - ZADV;
            }
#pragma unroll
            for (int k = 0; k < NKD; ++k) fetch(q[k], p + k, d >= 0 && (k < ZADV || first));
        };
        auto stash_step = [&](int d, const x3_f32x4 (&q)[NKD][C::NPF]) {
            if (d < 0) return;
            const bool first = x3_desc_first(d);
#pragma unroll
            for (int k = 0; k < NKD; ++k) if (k < ZADV || first) stash(q[k], (wslot + k) % NSLOT);
            wslot = (wslot + (first ? NKD : ZADV)) % NSLOT;
        };
        // one tick: step (s + 1) sits in buffer qa = pfq[(s + 1) & 1]; that buffer then takes step (s + 3)
        auto tick = [&](int s, x3_f32x4 (&qa)[NKD][C::NPF]) {
            if (pw == 0) X3_STAMP(4);
            const int d1 = sch.desc(s + 1), d3 = sch.desc(s + 3);
            if constexpr (KSPLIT > 1 && !C::EPI_CONSUMER) epi_open(sch.desc(s - 1));      // step s - 1: where it goes + its skip loads
            stash_step(d1, qa);
            if (pw == 0) X3_STAMP(5);
            fetch_step(d3, qa);
            if (pw == 0) X3_STAMP(6);
            if constexpr (KSPLIT > 1 && !C::EPI_CONSUMER) epi_close((s + 1) & 1);
            if (pw == 0) X3_STAMP(7);
            __syncthreads();
        };
        // prologue: the planes of step 0 straight into the ring; steps 1 and 2 into the register queue
        fetch_step(sch.desc(0), pfq[0]);
        fetch_step(sch.desc(1), pfq[1]);
        stash_step(sch.desc(0), pfq[0]);
        fetch_step(sch.desc(2), pfq[0]);
        __syncthreads();
#pragma unroll 1
        for (int s = 0; s < nsteps; s += 2) {
            tick(s, pfq[1]);
            if (s + 1 < nsteps) tick(s + 1, pfq[0]);
        }
        if constexpr (KSPLIT > 1 && !C::EPI_CONSUMER) { epi_open(sch.desc(nsteps - 1)); epi_close((nsteps + 1) & 1); }
    }
    // ---- bound of the output: one atomic max per block
    if (ymax) {
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, m));
        if (lane == 0) redmax[wave] = vmax;
        __syncthreads();
        if (tid == 0) {
            float m = redmax[0];
#pragma unroll
            for (int i = 1; i < 8; ++i) m = fmaxf(m, redmax[i]);
            if (dm.ysq) m = m * m;
            atomicMax(reinterpret_cast<unsigned int*>(ymax) + (blockIdx.x & 63) * 16, __float_as_uint(m));
        }
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------
#define RCMVS_X3_LIST(X) X(8, 8, X3_S1) X(16, 8, X3_S1) X(32, 8, X3_S1) X(16, 16, X3_S1) X(8, 16, X3_S2) X(16, 32, X3_S2) X(16, 8, X3_T2) \
    X(8, 8, X3_P1) X(16, 16, X3_P1) X(32, 32, X3_P1) X(32, 16, X3_P1) X(64, 32, X3_P1) X(32, 32, X3_S1) X(32, 16, X3_T2)
// the two-piece fp16 form: the 3-D kinds (the CostRegNet layers, whose producers maintain the activation bound)
#define RCMVS_X3H_LIST(X) X(8, 8, X3_S1) X(16, 8, X3_S1) X(32, 8, X3_S1) X(16, 16, X3_S1) X(8, 16, X3_S2) X(16, 32, X3_S2) X(16, 8, X3_T2) \
    X(32, 32, X3_S1) X(32, 16, X3_T2)

bool conv3d_x3_supported(int Ci, int Co, int kind) {
#define X3_CASE(CI, CO, K) if (Ci == CI && Co == CO && kind == K) return true;
    RCMVS_X3_LIST(X3_CASE)
#undef X3_CASE
    return false;
}

// conv3d_deep.hip: the deep U-Net levels (32 -> 64 stride 2, 64 -> 64, 64 -> 32 transposed) in the same arithmetic, one tile-owning block
// per 32 cells instead of z-marching persistent blocks; they share the x3h slot of the packed-weight blob and of the dispatch
bool conv3d_deep_supported(int Ci, int Co, int kind);
long long conv3d_deep_weight_floats(int Ci, int Co, int kind);
int conv3d_deep_pack(const float* w, float* img, int Co, int Ci, int kind, int transposed, const float* wsc, hipStream_t st);
int conv3d_deep_launch(const float* x, const float* wimg, const float* scale, const float* shift, const float* res, float* y,
                       int B, int D, int H, int W, int Ci, int Co, int kind, int relu, hipStream_t st, const float* xmax, float* ymax);

// conv3d_z8.hip: conv0 of stages 2 / 3 (Cin = 16 / 8 -> 8) and conv2 (16 -> 16), stride 1, no skip tensor, without the producer / consumer split; same image
bool conv3d_z8_supported(int Ci, int Co, int kind);
// conv3d_zs2.hip: conv1 (8 -> 16, stride 2) of a B = 1 inference scene on the same z-streaming scheme
bool conv3d_zs2_supported(int Ci, int Co, int kind);
int conv3d_zs2_launch(const float* x, const float* wimg, const float* scale, const float* shift, float* y,
                      int B, int D, int H, int W, int relu, hipStream_t st, int max_blocks, const float* xmax, float* ymax);
int conv3d_z8_launch(const float* x, const float* wimg, const float* scale, const float* shift, float* y,
                     int B, int D, int H, int W, int Ci, int Co, int relu, hipStream_t st, int max_blocks, const float* xmax, float* ymax);

bool conv3d_x3h_supported(int Ci, int Co, int kind) {
    if (conv3d_deep_supported(Ci, Co, kind)) return true;
#define X3_CASE(CI, CO, K) if (Ci == CI && Co == CO && kind == K) return true;
    RCMVS_X3H_LIST(X3_CASE)
#undef X3_CASE
    return false;
}

long long conv3d_x3_weight_floats(int Ci, int Co, int kind) {      // size of one x3 image in floats (it is stored as bf16 triples)
#define X3_CASE(CI, CO, K) if (Ci == CI && Co == CO && kind == K) return (long long)X3<CI, CO, K>::KSTEPS * 3 * X3<CI, CO, K>::MT_ALL * 64 * 8 / 2;
    RCMVS_X3_LIST(X3_CASE)
#undef X3_CASE
    return 0;
}

long long conv3d_x3h_weight_floats(int Ci, int Co, int kind) {     // header (4 floats) + the fp16 pairs
    if (conv3d_deep_supported(Ci, Co, kind)) return conv3d_deep_weight_floats(Ci, Co, kind);
#define X3_CASE(CI, CO, K) if (Ci == CI && Co == CO && kind == K) return 4 + (long long)X3<CI, CO, K, 2>::KSTEPS * 2 * X3<CI, CO, K, 2>::MT_ALL * 64 * 8 / 2;
    RCMVS_X3H_LIST(X3_CASE)
#undef X3_CASE
    return 0;
}

int conv3d_x3_pack(const float* w, float* img, int Co, int Ci, int kind, int transposed, hipStream_t st) {
#define X3_CASE(CI, CO, K) if (Ci == CI && Co == CO && kind == K) { \
        const int nthr = X3<CI, CO, K>::KSTEPS * X3<CI, CO, K>::MT_ALL * 64 * 8; \
        hipLaunchKernelGGL((x3_pack_kernel<CI, CO, K, 3>), dim3((nthr + 255) / 256), dim3(256), 0, st, w, reinterpret_cast<unsigned short*>(img), transposed, (const float*)nullptr); \
        return launch_status("conv3d_x3_pack"); }
    RCMVS_X3_LIST(X3_CASE)
#undef X3_CASE
    return fail(-1, "conv3d_x3_pack: unsupported Ci=%d Co=%d kind=%d", Ci, Co, kind);
}

// wsc: two device floats {scale, 1 / scale} of the weight tensor (conv3d_x3_wscale)
int conv3d_x3h_pack(const float* w, float* img, int Co, int Ci, int kind, int transposed, const float* wsc, hipStream_t st) {
    if (conv3d_deep_supported(Ci, Co, kind)) return conv3d_deep_pack(w, img, Co, Ci, kind, transposed, wsc, st);
#define X3_CASE(CI, CO, K) if (Ci == CI && Co == CO && kind == K) { \
        const int nthr = X3<CI, CO, K, 2>::KSTEPS * X3<CI, CO, K, 2>::MT_ALL * 64 * 8; \
        hipLaunchKernelGGL((x3_pack_kernel<CI, CO, K, 2>), dim3((nthr + 255) / 256), dim3(256), 0, st, w, reinterpret_cast<unsigned short*>(img), transposed, wsc); \
        return launch_status("conv3d_x3h_pack"); }
    RCMVS_X3H_LIST(X3_CASE)
#undef X3_CASE
    return fail(-1, "conv3d_x3h_pack: unsupported Ci=%d Co=%d kind=%d", Ci, Co, kind);
}

int conv3d_x3_wscale(const float* w, int n, float* out, hipStream_t st) {
    hipLaunchKernelGGL(x3_wscale_kernel, dim3(1), dim3(256), 0, st, w, n, out);
    return launch_status("conv3d_x3_wscale");
}

template <int CI, int CO, int K, int NP>
static int x3_launch_t(const float* x, const float* wimg, const float* scale, const float* shift, const float* res, float* y,
                       X3Dims dm, int gh, int gw, int n_blk, int dev, const float* xmax, float* ymax, hipStream_t st) {
    using C = X3<CI, CO, K, NP>;
    const int tiles_x = (gw + C::TX - 1) / C::TX, tiles_y = (gh + C::TY - 1) / C::TY;
    dm.tiles_x = tiles_x; dm.ntiles = tiles_x * tiles_y;
    // one persistent block per CU walks its share of the (batch, tile, z chunk) items: pick the chunk length that minimises the
    // longest block (items per block x (steps per item + ~1.5 steps for the extra planes an item start loads))
    long long best = -1; int zchunk = dm.Dt;
    for (int zc = dm.Dt; zc >= (C::NKD > 1 ? 2 : 1); --zc) {
        const int nch = (dm.Dt + zc - 1) / zc;
        if (nch > 1 && (dm.Dt + nch - 1) / nch != zc) continue;            // only balanced splits
        const long long items = (long long)dm.B * dm.ntiles * nch;
        const long long per_blk = (items + n_blk - 1) / n_blk;
        const long long cost = per_blk * (2 * zc + (C::NKD > 1 ? 3 : 0));
        if (best < 0 || cost < best) { best = cost; zchunk = zc; }
    }
    dm.zchunk = zchunk; dm.nchunks = (dm.Dt + zchunk - 1) / zchunk;
    const long long items = (long long)dm.B * dm.ntiles * dm.nchunks;
    if (items >= 0x7fffffffLL) return fail(-1, "conv3d_x3: too many work items");
    dm.nitems = (int)items;
    // the balanced schedule (see the kernel): every block the same number of steps; taken when the model says it is shorter than the best
    // chunked split (cost in half steps: steps + 1.5 per item start, a block of a balanced launch starts ceil(steps / Dt) + 1 items at most)
    dm.bal = 0;
    {
        constexpr int mode = 2;                                             // by the cost model (0 / 1 = off / forced were A/B settings of round 4: profiles/HISTORY.md)
        const long long T = (long long)dm.B * dm.ntiles * dm.Dt;
        const long long spb = (T + n_blk - 1) / n_blk + 1;                   // steps of the longest block
        const long long ipb = (spb + dm.Dt - 1) / dm.Dt + 1;                 // item starts of a block, at most
        const long long cost_bal = 2 * spb + (C::NKD > 1 ? 3 : 0) * ipb;
        if (T < 0x7fffffffLL && spb < 32768 && dm.Dt < 32768 && (mode == 1 || (mode == 2 && cost_bal < best))) {      // the step table packs z into 15 bits
            dm.bal = 1;
            dm.itemcap = (int)((ipb + 1 + 3) & ~3LL);
            dm.stepcap = (int)((spb + 2 + 3) & ~3LL);
            const size_t ldsb = (size_t)C::LDSB + (size_t)dm.itemcap * 16 + (size_t)dm.stepcap * 4 + 64;
            if (ldsb <= 160 * 1024) {
                const int grid_b = (int)(T < n_blk ? T : n_blk);
                static std::atomic<bool> attr_set_b[64];
                if (!attr_set_b[dev].load(std::memory_order_acquire)) { (void)hipFuncSetAttribute((const void*)conv3d_x3_kernel<CI, CO, K, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set_b[dev].store(true, std::memory_order_release); }
                hipLaunchKernelGGL((conv3d_x3_kernel<CI, CO, K, NP>), dim3((unsigned)grid_b), dim3(512), ldsb, st, x, reinterpret_cast<const x3_u32x4*>(wimg), scale, shift, res, y, dm, xmax, ymax);
                return launch_status("conv3d_x3");
            }
            dm.bal = 0;
        }
    }
    // schedule tables of a block (LDS, behind the ring and the partial tiles): items per block (an XCD's share is split over
    // grid / 8 blocks: + 1 for the remainders), steps per block; more blocks than CUs when the tables would not fit
    int grid_n = dm.nitems < n_blk ? dm.nitems : n_blk;
    size_t lds = 0;
    for (;;) {
        const long long per_blk = (items + grid_n - 1) / grid_n + 2;
        dm.itemcap = (int)((per_blk + 3) & ~3LL);
        dm.stepcap = (int)(((long long)dm.itemcap * dm.zchunk + 3) & ~3LL);
        lds = (size_t)C::LDSB + (size_t)dm.itemcap * 16 + (size_t)dm.stepcap * 4 + 64;
        if (lds <= 160 * 1024 || grid_n >= dm.nitems) break;
        grid_n = grid_n * 2 < dm.nitems ? grid_n * 2 : dm.nitems;
    }
    if (lds > 160 * 1024) return fail(-1, "conv3d_x3: the schedule of a block does not fit the LDS (%zu bytes)", lds);
    if (dm.zchunk >= 32768 || dm.itemcap >= 32768) return fail(-1, "conv3d_x3: schedule descriptor fields overflow");
    dim3 grid((unsigned)grid_n);
    static std::atomic<bool> attr_set[64];
    if (!attr_set[dev].load(std::memory_order_acquire)) { (void)hipFuncSetAttribute((const void*)conv3d_x3_kernel<CI, CO, K, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set[dev].store(true, std::memory_order_release); }
    hipLaunchKernelGGL((conv3d_x3_kernel<CI, CO, K, NP>), grid, dim3(512), lds, st, x, reinterpret_cast<const x3_u32x4*>(wimg), scale, shift, res, y, dm, xmax, ymax);
    return launch_status("conv3d_x3");
}

// xmax != nullptr selects the two-piece fp16 form (wimg must then be the x3h image of the pair); ymax is optional in both forms
int conv3d_x3_launch(const float* x, const float* wimg, const float* scale, const float* shift, const float* res, float* y,
                     int B, int D, int H, int W, int Ci, int Co, int kind, int relu, hipStream_t st, int max_blocks, int s2d,
                     const float* xmax, float* ymax) {
    if (xmax && conv3d_deep_supported(Ci, Co, kind)) return conv3d_deep_launch(x, wimg, scale, shift, res, y, B, D, H, W, Ci, Co, kind, relu, st, xmax, ymax);
    if (xmax && !res && !s2d && conv3d_z8_supported(Ci, Co, kind)) {
        const int rc = conv3d_z8_launch(x, wimg, scale, shift, y, B, D, H, W, Ci, Co, relu, st, max_blocks, xmax, ymax);
        if (rc != 1) return rc;        // (1 = not taken: volumes with more than 64 k steps per block stay on the split kernel)
    }
    if (xmax && !res && !s2d && conv3d_zs2_supported(Ci, Co, kind)) {
        const int rc = conv3d_zs2_launch(x, wimg, scale, shift, y, B, D, H, W, relu, st, max_blocks, xmax, ymax);
        if (rc != 1) return rc;
    }
    const int ysq = (s2d >> 1) & 1;            // `s2d` carries two flags: bit 0 = space-to-depth view of the input, bit 1 = square the output bound
    s2d &= 1;
    if (s2d && (kind != X3_P1 || Ci % 16 != 0)) return fail(-1, "conv3d_x3: the space-to-depth view needs the planar kind and Ci a multiple of 16");
    if ((long long)B * D * H * W * Ci * 4 >= 0x7ffffff0LL) return fail(-1, "conv3d_x3: input tensor too large for 32-bit offsets");
    {
        const long long so = kind == X3_T2 ? 8 : (kind == X3_S2 ? 1 : 1);        // output voxels per input voxel (upper bound)
        if ((long long)B * D * H * W * so * Co * 4 >= 0x7ffffff0LL) return fail(-1, "conv3d_x3: output tensor too large for 32-bit offsets");
    }
    // per-device facts (a process may drive several GPUs, e.g. nn.DataParallel replicas): CU count, and whether the kernel's
    // dynamic-LDS limit has been raised on that device (atomics: the library is re-entrant).
    constexpr int MAXDEV = 64;
    static std::atomic<int> cu_of[MAXDEV];
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
    dm.B = B; dm.D = D; dm.H = H; dm.W = W; dm.relu = relu; dm.s2d = s2d; dm.ysq = ysq; dm.dbg = 0; dm.trace = nullptr;
    dm.place = 0;
#if X3_ABLATION
    dm.dbg = x3_ablation_mask;
    dm.trace = x3_trace_buf;
#endif
    if (kind == X3_T2) { dm.Do = 2 * D; dm.Ho = 2 * H; dm.Wo = 2 * W; }
    else { const int s = kind == X3_S2 ? 2 : 1; dm.Do = (D - 1) / s + 1; dm.Ho = (H - 1) / s + 1; dm.Wo = (W - 1) / s + 1; }
    const int gh = kind == X3_T2 ? H : dm.Ho, gw = kind == X3_T2 ? W : dm.Wo;      // tile grid
    dm.Dt = kind == X3_T2 ? D : dm.Do;
    dm.tiles_x = dm.zchunk = dm.ntiles = dm.nchunks = dm.nitems = dm.itemcap = dm.stepcap = 0;
    if (xmax) {
#define X3_CASE(CI, CO, K) if (Ci == CI && Co == CO && kind == K) return x3_launch_t<CI, CO, K, 2>(x, wimg, scale, shift, res, y, dm, gh, gw, n_blk, dev, xmax, ymax, st);
        RCMVS_X3H_LIST(X3_CASE)
#undef X3_CASE
        return fail(-1, "conv3d_x3 (fp16 pair form): unsupported Ci=%d Co=%d kind=%d", Ci, Co, kind);
    }
#define X3_CASE(CI, CO, K) if (Ci == CI && Co == CO && kind == K) return x3_launch_t<CI, CO, K, 3>(x, wimg, scale, shift, res, y, dm, gh, gw, n_blk, dev, xmax, ymax, st);
    RCMVS_X3_LIST(X3_CASE)
#undef X3_CASE
    return fail(-1, "conv3d_x3: unsupported Ci=%d Co=%d kind=%d", Ci, Co, kind);
}

}  // namespace rcmvs
