// The deep levels of the 3-D U-Nets (conv5 32->64 stride 2, conv6 64->64, conv7 64->32 transposed, and conv3 16->32 stride 2 / conv4 32->32 / conv9 32->16 transposed one level up;
// models/modules.py:470-501 in the reference) on the fp16-pair matrix-core arithmetic of conv3d_x3.hip (two fp16 pieces per operand after an exact power-of-two
// pre-scale, three v_mfma_f32_16x16x32_f16 per product), for callers that hand over a bound of max|x|.  gfx950 only.
//
// Why a kernel of their own.  These volumes are tiny (a DTU scene: 1 920 - 5 120 cells of 64 channels) and the weight tensors are
// large (64 x 64 x 27 = 442 KB): the z-marching persistent blocks of conv3d_x3.hip have nothing to march over (1 - 6 planes), and
// the fp32-MFMA kernel (conv3d_mfma.hip), whose blocks own 16 cells x 16 channels, re-reads 110 KB of weights and 110 KB of
// activations per block through the vector L1 and runs fp32 MFMAs at the VALU rate: 11 - 15 us per launch, nine launches per scene.
// Here a block owns 32 cells (4 x 8 of one plane = two MFMA n-tiles) x ALL output channels:
//   * the input halo of the tile is loaded once (raw buffer loads, out of range -> 0 = the zero padding), split once into its two
//     fp16 pieces and parked in LDS (two piece planes, voxel stride = 2 Cin + 16 bytes so that lanes a voxel apart read different
//     banks); B fragments are one ds_read_b128 per piece,
//   * the weight image streams ONCE per block from L2 straight into A fragments (a ring of PF k-steps of prefetch per wave):
//     442 KB per block through a 64 B/clk vector L1 is the floor of the launch (~2.9 us); the 3 x 54 x 8 MFMAs of a block are ~2 us,
//   * conv / stride-2 conv: wave = (m-tile, K half); the two K halves of a tile meet through LDS.  Transposed conv: the (output
//     parity class, m-tile) pairs (taps per class 1, 2, 2, 4, 2, 4, 4, 8) are dealt to the waves so that every SIMD gets 13 - 14
//     tap-tiles; a wave finishes its pairs one after the other and owns their tiles outright,
//   * k-steps whose input plane lies outside the volume (the kd = 0, 2 taps of a one-plane volume, ...) are dropped from the wave's
//     step range: neither their weights nor their zeros are touched.
// Epilogue as everywhere: un-scale (exact, folded into the BatchNorm scale), BN scale / shift, ReLU, skip-add, float4 store, and the
// bound of the stored outputs (one atomic max per block) for the next layer.
#include "common.h"
#include "x3_pieces.h"
#include <atomic>
#include <type_traits>

namespace rcmvs {

enum { DP_S1 = 0, DP_S2 = 1, DP_T2 = 2 };

// taps of output parity class p = (pd, ph, pw) of the transposed convolution: per axis one tap (k = 1, input offset 0) for parity 0,
// two (k = 0 -> input offset 1, k = 2 -> input offset 0) for parity 1
__host__ __device__ constexpr int dp_ntaps(int p) { return (((p >> 2) & 1) + 1) * (((p >> 1) & 1) + 1) * ((p & 1) + 1); }
__host__ __device__ constexpr int dp_first_sum(int p) { int s = 0; for (int q = 0; q < p; ++q) s += dp_ntaps(q); return s; }
// taps in front of class p: 0 1 3 5 9 11 15 19 (a table: the kernel asks with a run-time class)
__host__ __device__ constexpr int dp_first(int p) { return (int)((0x130F0B0905030100ull >> (8 * p)) & 0xffu); }
static_assert(dp_first(1) == dp_first_sum(1) && dp_first(2) == dp_first_sum(2) && dp_first(3) == dp_first_sum(3) && dp_first(4) == dp_first_sum(4) &&
              dp_first(5) == dp_first_sum(5) && dp_first(6) == dp_first_sum(6) && dp_first(7) == dp_first_sum(7) && dp_first(0) == 0, "class offsets");
// tap i of class p -> kernel index per axis (kd, kh, kw); i = (id, ih, iw) with one bit per odd axis
__host__ __device__ inline void dp_class_tap(int p, int i, int& kd, int& kh, int& kw) {
    const int pd = (p >> 2) & 1, ph = (p >> 1) & 1, pw = p & 1;
    const int iw = i & pw, ih = (i >> pw) & ph, id = (i >> (pw + ph)) & pd;
    kd = pd ? 2 * id : 1; kh = ph ? 2 * ih : 1; kw = pw ? 2 * iw : 1;
}

template <int CIN, int COUT, int KIND, int MS = 1>
struct Deep {
    // cell tile: 4 x 8 cells of one (b, d) plane, n-tile t = rows 2t, 2t + 1.  (NT = 4, 8 x 8 cells, for the 32 -> 32 layer -- 8x the cells, a quarter of
    // the weights -- measured 9.0 / 20.9 us per launch against 8.9 / 17.3, and NT = 1, 2 x 8 cells, 11.9 / 24.7: 32 cells is the sweet spot between the
    // blocks in flight and the halo overhead; a persistent form with register-stationary weights measured 22.4: DESIGN.md section 8)
    static constexpr int NT = 2;
    static constexpr int TH = 2 * NT, TW = 8;
    static constexpr int MT = COUT / 16;
    static constexpr int HALVES = CIN >= 32 ? CIN / 32 : 1;      // k-steps (K = 32 input channels) per tap
    static constexpr int TPS = CIN >= 32 ? 1 : 32 / CIN;         // taps per k-step (Cin = 16: a k-step is two taps x 16 channels; the 28th tap is a zero)
    static constexpr int HD = KIND == DP_T2 ? 2 : 3;
    static constexpr int HH = KIND == DP_S1 ? TH + 2 : (KIND == DP_S2 ? 2 * TH + 1 : TH + 1);
    static constexpr int HW = KIND == DP_S1 ? TW + 2 : (KIND == DP_S2 ? 2 * TW + 1 : TW + 1);
    static constexpr int CS = KIND == DP_S2 ? 2 : 1;    // halo voxels per cell step
    static constexpr int NVOX = HD * HH * HW;
    static constexpr int VS = CIN * 2 + 16;             // bytes of a voxel in a piece plane
    static constexpr int PLANE = NVOX * VS;
    static constexpr int KSTEPS = (27 + TPS - 1) / TPS * HALVES;
    static constexpr int UNITS = NVOX * (CIN / 8);      // 8-channel units of the halo (32 B in, 16 + 16 B out)
    static constexpr int NLD = (UNITS + 511) / 512;
    static constexpr int PF = KIND == DP_T2 ? 6 : 8;    // k-steps of weight prefetch per ring (8 registers per step)
    // MS = 2 (64 output channels, conv5 / conv6, when twice the tiles still fit one round of blocks: 60 tiles at stage 1): the m-tiles of a tile are cut over two blocks -- each
    // streams its share of the weight image only (the vector L1's 64 B/clk is the floor of these launches) and halves its K ranges once more: 9.0 / 10.2 us against 10.2 / 12.8;
    // with 160 tiles (stages 2 / 3) the 320 blocks are a second round: 13.5 / 15.4 against 10.5 / 13.0
    static constexpr int MSPLIT = (KIND != DP_T2 && MT == 4) ? MS : 1;
    static constexpr int MB = MT / MSPLIT;                     // m-tiles per block
    static constexpr int KP = KIND == DP_T2 ? 1 : 8 / MB;      // convolutions: wave = (m-tile, one of KP parts of the K range); the parts meet through LDS
    static constexpr int PARTB = KIND == DP_T2 ? 0 : (KP - 1) * MB * NT * 1024;
    static constexpr int LDS = 2 * PLANE + PARTB + 64;
    static constexpr long long IMG_HALFS = 8 + (long long)KSTEPS * 2 * MT * 512;
    static_assert((CIN % 32 == 0 || (CIN == 16 && KIND != DP_T2)) && COUT % 16 == 0 && (KIND == DP_T2 ? (MT == 2 || MT == 1) : (MT == 4 || MT == 2)), "wave layout");
};

// ---- weight image: a 16-byte header {s_w, 1 / s_w, 0, 0}, then [k-step][piece][m-tile][lane][8 fp16]: the A fragment of
// v_mfma_f32_16x16x32_f16 (row = lane & 15, k = 8 (lane >> 4) + e).  k-step order: (tap, channel half) for the convolutions;
// (parity class, tap of the class, channel half) for the transposed convolution.  transposed: as in conv3d_x3.hip.
template <int CIN, int COUT, int KIND>
__global__ void deep_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ img, int transposed, const float* __restrict__ wsc) {
    using C = Deep<CIN, COUT, KIND>;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= C::KSTEPS * C::MT * 512) return;
    const int e = t & 7, lane = (t >> 3) & 63, mt = (t >> 9) % C::MT, j = (t >> 9) / C::MT;
    const int co = mt * 16 + (lane & 15), kq = lane >> 4;
    int tap, half;
    if (KIND == DP_T2) {
        int p = 0;
        while (p < 7 && j >= dp_first(p + 1) * C::HALVES) ++p;
        const int jj = j - dp_first(p) * C::HALVES;
        int kd, kh, kw;
        dp_class_tap(p, jj / C::HALVES, kd, kh, kw);
        tap = (kd * 3 + kh) * 3 + kw;
        half = jj % C::HALVES;
    } else if (C::TPS == 2) { tap = 2 * j + (kq >> 1); half = 0; }
    else { tap = j / C::HALVES; half = j % C::HALVES; }
    const int ci = C::TPS == 2 ? (kq & 1) * 8 + e : half * 32 + kq * 8 + e;
    float v = 0.0f;
    if (tap < 27) v = transposed ? w[((long long)ci * COUT + co) * 27 + (transposed == 2 ? 26 - tap : tap)] : w[((long long)co * CIN + ci) * 27 + tap];
    const float sw = wsc[0];
    if (t == 0) { float* hdr = reinterpret_cast<float*>(img); hdr[0] = sw; hdr[1] = wsc[1]; hdr[2] = 0.0f; hdr[3] = 0.0f; }
    const float vs = v * sw;                                       // exact (power of two)
    const _Float16 h = (_Float16)vs;                               // round to nearest even
    const _Float16 l = (_Float16)(vs - (float)h);
    const long long base = 8 + (((long long)j * 2) * C::MT + mt) * 512 + lane * 8 + e;
    img[base] = __builtin_bit_cast(unsigned short, h);
    img[base + (long long)C::MT * 512] = __builtin_bit_cast(unsigned short, l);
}

struct DeepDims {
    int B, D, H, W;        // input volume
    int Dg, Hg, Wg;        // cell grid (output volume of the convolutions, input volume of the transposed convolution)
    int Do, Ho, Wo;        // output volume
    int tiles_h, tiles_w, relu;
};

// Work of a wave: up to three SEGMENTS, each one (m-tile, contiguous k-step range, output parity class) with its own accumulators,
// finished (or handed over) before the next one starts.
//   convolutions: one segment = (m-tile wave % MT, K part wave / MT: halves for 64 output channels, quarters for 32); the parts meet through LDS.
//   transposed:   (class, m-tile) pairs dealt so that the two waves of every SIMD get 13 - 14 tap-tiles (taps per class 8 4 4 4 2 2 2 1):
//                 waves 0 1: class 7; 2 3: classes 3, 1, 0; 4 5: classes 5, 2; 6 7: classes 6, 4; m-tile = wave & 1.
__device__ __forceinline__ int dp_t2_segments(int wave, int (&cls)[3]) {
    const int g = wave >> 1;
    cls[0] = g == 0 ? 7 : (g == 1 ? 3 : (g == 2 ? 5 : 6));
    cls[1] = g == 1 ? 1 : (g == 2 ? 2 : 4);
    cls[2] = 0;
    return g == 0 ? 1 : (g == 1 ? 3 : 2);
}

template <int CIN, int COUT, int KIND, int MS = 1>
__global__ __launch_bounds__(512) void conv3d_deep_kernel(
    const float* __restrict__ x, const x3_u32x4* __restrict__ wimg, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ res, float* __restrict__ y, DeepDims dm, const float* __restrict__ xmax, float* __restrict__ ymax) {
    using C = Deep<CIN, COUT, KIND, MS>;
    constexpr int MT = C::MT, NT = C::NT, PF = C::PF, VS = C::VS, PLANE = C::PLANE, HALVES = C::HALVES;
    constexpr int OOB = 0x7ffffff0;
    extern __shared__ __attribute__((aligned(16))) x3_byte smem[];
    x3_byte* const part = smem + 2 * PLANE;
    float* const redmax = reinterpret_cast<float*>(part + C::PARTB);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    int bt = blockIdx.x;
    const int tw = bt % dm.tiles_w; bt /= dm.tiles_w;
    const int th = bt % dm.tiles_h; bt /= dm.tiles_h;
    const int dg = bt % dm.Dg, b = bt / dm.Dg;
    const int h0 = th * C::TH, w0 = tw * C::TW;
    // origin of the halo in the input volume
    const int zd0 = KIND == DP_S1 ? dg - 1 : (KIND == DP_S2 ? 2 * dg - 1 : dg);
    const int zh0 = KIND == DP_S1 ? h0 - 1 : (KIND == DP_S2 ? 2 * h0 - 1 : h0);
    const int zw0 = KIND == DP_S1 ? w0 - 1 : (KIND == DP_S2 ? 2 * w0 - 1 : w0);
    const float xmax_lane = xmax[lane * 16];                         // the first requests of the kernel: bound and weight scale
    const float whdr = reinterpret_cast<const float*>(wimg)[1];

    // ---- segments.  Taps whose input plane lies outside the volume contribute zeros and are dropped: in tap order (kd-major; of the
    // two kd choices of an odd class the one that reads plane dg + 1 first) they are a prefix (first plane below the volume) and / or
    // a suffix (last plane above it) of a segment's range.
    int seg_cls[3] = {0, 0, 0}, nseg = 1, mt0;
    if constexpr (KIND == DP_T2 && MT == 1) { seg_cls[0] = wave; mt0 = 0; }                 // 16 output channels: a wave per parity class (1 - 8 taps: the MFMAs are nothing here)
    else if constexpr (KIND == DP_T2) { nseg = dp_t2_segments(wave, seg_cls); mt0 = wave & 1; }
    else mt0 = blockIdx.y * C::MB + wave % C::MB;
    const int kp = KIND == DP_T2 ? 0 : wave / C::MB;                 // which part of the K range (convolutions)
    const int mtl = KIND == DP_T2 ? 0 : wave % C::MB;                // (m-tile inside the block's share)
    auto seg_range = [&](int sg, int& j0, int& nsteps) {
        if constexpr (KIND == DP_T2) {
            const int cls = seg_cls[sg], nt = dp_ntaps(cls);
            const bool drop = (cls & 4) && zd0 + 1 >= dm.D;          // the k = 0 taps of an odd plane read input plane dg + 1
            j0 = (dp_first(cls) + (drop ? nt / 2 : 0)) * HALVES;
            nsteps = (drop ? nt / 2 : nt) * HALVES;
        } else {
            // (two taps per step: a step that straddles the boundary is kept -- its dropped tap reads the halo's zeros)
            const int lo = (zd0 < 0 ? 9 : 0) * HALVES / C::TPS, hi = ((zd0 + 2 >= dm.D ? 18 : 27) * HALVES + C::TPS - 1) / C::TPS, per = (hi - lo + C::KP - 1) / C::KP;
            j0 = min(lo + kp * per, hi);
            nsteps = min(per, hi - j0);                              // (>= 1: a range holds 9 steps at least)
        }
    };
    // byte offset of step j's tap inside a piece plane
    auto tap_offset = [&](int cls, int j) -> int {
        const int half = j % HALVES;
        int od, oh, ow;
        if constexpr (KIND == DP_T2) {
            int kd, kh, kw;
            dp_class_tap(cls, j / HALVES - dp_first(cls), kd, kh, kw);
            od = kd == 0; oh = kh == 0; ow = kw == 0;
        } else { const int tp = j / HALVES; od = tp / 9; oh = (tp / 3) % 3; ow = tp % 3; }
        return ((od * C::HH + oh) * C::HW + ow) * VS + half * 64;
    };
    // (two taps per step: offset of tap tp, the zero 28th one mapped to the 27th)
    auto tap_offset_of = [&](int tp) -> int { tp = min(tp, 26); return (((tp / 9) * C::HH + (tp / 3) % 3) * C::HW + tp % 3) * VS; };

    // ---- weights: A fragments straight from the image (L2), a ring of PF k-steps
    __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<x3_u32x4*>(wimg + 1), (short)0, (int)(C::KSTEPS * 2 * MT * 1024), 0x00020000);
    // (transposed form: TWO rings -- segment s runs on ring s & 1 while the first PF steps of segment s + 1 are already on their way
    // into the other one; a wave's segments are short and each would otherwise open with a full round trip to L2)
    constexpr int NR = KIND == DP_T2 ? 2 : 1;
    x3_u32x4 ah[NR][PF], al[NR][PF];
    const int alane = lane * 16 + mt0 * 1024;
    auto load_a = [&](auto ring, int slot, int j0, int nsteps, int jj) {     // jj >= nsteps: out-of-range offsets, zeros without traffic
        constexpr int R = decltype(ring)::value;
        const int so = (j0 + jj) * (2 * MT * 1024);
        const int vo = jj < nsteps ? alane : OOB;
        ah[R][slot] = __builtin_amdgcn_raw_buffer_load_b128(wrs, vo, so, 0);
        al[R][slot] = __builtin_amdgcn_raw_buffer_load_b128(wrs, vo == OOB ? OOB : vo + MT * 1024, so, 0);
    };
    using Ring0 = std::integral_constant<int, 0>;
    using Ring1 = std::integral_constant<int, NR - 1>;

    // ---- halo: load (the vector-memory path of the CU takes these requests first: the block cannot start before they are back),
    // then the first PF steps of weights, then split and park
    float unscale;
    {
        __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), (short)0,
                                                                       (int)((long long)dm.B * dm.D * dm.H * dm.W * CIN * 4), 0x00020000);
        x3_f32x4 pf[C::NLD][2];
        int lo_[C::NLD];
#pragma unroll
        for (int i = 0; i < C::NLD; ++i) {
            const int u = tid + i * 512;
            const int vox = u / (CIN / 8), cu = u % (CIN / 8);
            const int hw_ = vox % C::HW, hh = (vox / C::HW) % C::HH, hd = vox / (C::HW * C::HH);
            const int id = zd0 + hd, ih = zh0 + hh, iw = zw0 + hw_;
            const bool in = u < C::UNITS && id >= 0 && id < dm.D && ih >= 0 && ih < dm.H && iw >= 0 && iw < dm.W;
            const int off = in ? ((((b * dm.D + id) * dm.H + ih) * dm.W + iw) * CIN + cu * 8) * 4 : OOB;
            pf[i][0] = __builtin_bit_cast(x3_f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0));
            pf[i][1] = __builtin_bit_cast(x3_f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, in ? off + 16 : OOB, 0, 0));
            lo_[i] = u < C::UNITS ? vox * VS + cu * 16 : -1;
        }
        {
            int j0, nsteps;
            seg_range(0, j0, nsteps);
#pragma unroll
            for (int u = 0; u < PF; ++u) load_a(Ring0{}, u, j0, nsteps, u);
        }
        float bound = xmax_lane;
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) bound = fmaxf(bound, __shfl_xor(bound, m));
        float xinv;
        const float xs_scale = x3_pow2_scale(bound, xinv);
        unscale = xinv * whdr;
#pragma unroll
        for (int i = 0; i < C::NLD; ++i) {
            x3_u32x2 h0_, l0_, h1_, l1_;
            x3_split4h(pf[i][0] * xs_scale, h0_, l0_);
            x3_split4h(pf[i][1] * xs_scale, h1_, l1_);
            if (lo_[i] >= 0) {
                *reinterpret_cast<x3_u32x4*>(smem + lo_[i]) = (x3_u32x4){h0_.x, h0_.y, h1_.x, h1_.y};
                *reinterpret_cast<x3_u32x4*>(smem + PLANE + lo_[i]) = (x3_u32x4){l0_.x, l0_.y, l1_.x, l1_.y};
            }
        }
    }
    __syncthreads();

    // B fragments: lane (n, kq) reads 16 bytes of cell (row 2t + (n >> 3), column n & 7) at the tap's offset, one step ahead (two
    // register sets, pinned with sched_barrier: left alone the scheduler sinks every ds_read to just before its MFMAs)
    static_assert(PF % 2 == 0, "the B double buffer alternates with the step parity");
    int bb[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bb[t] = (((2 * t + (n >> 3)) * C::CS) * C::HW + (n & 7) * C::CS) * VS + (C::TPS == 2 ? (kq & 1) : kq) * 16;
    float vmax = 0.0f;
#pragma unroll
    for (int sg = 0; sg < (KIND == DP_T2 ? 3 : 1); ++sg) {           // (unrolled: the ring of a segment is a compile-time choice)
        if (sg >= nseg) break;
        const int cls = seg_cls[sg];
        int j0, nsteps;
        seg_range(sg, j0, nsteps);
        x3_f32x4 acc[NT][3];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[t][c] = (x3_f32x4){0.f, 0.f, 0.f, 0.f};
        x3_u32x4 bq[2][NT][2];
        const int jl = j0 + min(lane, nsteps - 1);
        const int entv = C::TPS == 2 ? tap_offset_of(2 * jl) : tap_offset(cls, jl);      // lane jj: the tap offset of step jj (past the end: the last step's)
        const int entv2 = C::TPS == 2 ? tap_offset_of(2 * jl + 1) : 0;                   // ... and of the step's second tap (lanes kq = 2, 3 read that one)
        auto read_b = [&](int buf, int jj) {
            int boff = __builtin_amdgcn_readlane(entv, jj);
            if constexpr (C::TPS == 2) { const int b2 = __builtin_amdgcn_readlane(entv2, jj); boff = (kq >> 1) ? b2 : boff; }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                bq[buf][t][0] = *reinterpret_cast<const x3_u32x4*>(smem + bb[t] + boff);
                bq[buf][t][1] = *reinterpret_cast<const x3_u32x4*>(smem + PLANE + bb[t] + boff);
            }
        };
        auto run = [&](auto ring, auto other) {
            constexpr int R = decltype(ring)::value;
            if (sg + 1 < nseg) {                                     // (transposed form only: nseg = 1 otherwise)
                int j0n, nn;
                seg_range(sg + 1, j0n, nn);
#pragma unroll
                for (int u = 0; u < PF; ++u) load_a(other, u, j0n, nn, u);
            }
            auto mfmas = [&](int slot, int buf) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc[t][0] = x3_mfma<2>(ah[R][slot], bq[buf][t][0], acc[t][0]);
                    acc[t][1] = x3_mfma<2>(ah[R][slot], bq[buf][t][1], acc[t][1]);
                    acc[t][2] = x3_mfma<2>(al[R][slot], bq[buf][t][0], acc[t][2]);
                }
            };
            read_b(0, 0);
            int base = 0;
            for (; base + PF <= nsteps; base += PF) {                // full groups: a fixed number of loads per step, so the waits can be counted
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    read_b((u + 1) & 1, base + u + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    mfmas(u, u & 1);
                    __builtin_amdgcn_sched_barrier(0);
                    load_a(ring, u, j0, nsteps, base + u + PF);
                }
            }
#pragma unroll
            for (int u = 0; u < PF; ++u)
                if (base + u < nsteps) {
                    read_b((u + 1) & 1, base + u + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    mfmas(u, u & 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
        };
        if (NR == 1 || !(sg & 1)) run(Ring0{}, Ring1{});
        else run(Ring1{}, Ring0{});

        // ---- hand-over of the K halves (convolutions), epilogue
        x3_f32x4 tot[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) tot[t] = acc[t][0] + (acc[t][1] + acc[t][2]);
        bool finisher = true;
        if constexpr (KIND != DP_T2) {
            finisher = kp == 0;
            if (!finisher) {
#pragma unroll
                for (int t = 0; t < NT; ++t) *reinterpret_cast<x3_f32x4*>(part + ((((kp - 1) * C::MB + mtl) * NT + t) * 64 + lane) * 16) = tot[t];
            }
            __syncthreads();
            if (finisher) {
#pragma unroll
                for (int q = 0; q < C::KP - 1; ++q)
#pragma unroll
                    for (int t = 0; t < NT; ++t) tot[t] += *reinterpret_cast<const x3_f32x4*>(part + (((q * C::MB + mtl) * NT + t) * 64 + lane) * 16);
            }
        }
        if (finisher) {
            const int pd = (cls >> 2) & 1, ph = (cls >> 1) & 1, pw = cls & 1;
            const int co = mt0 * 16 + kq * 4;
            x3_f32x4 sc = (x3_f32x4){unscale, unscale, unscale, unscale}, sh = (x3_f32x4){0.f, 0.f, 0.f, 0.f};
            if (scale) {
                sc = *reinterpret_cast<const x3_f32x4*>(scale + co) * unscale;        // exact (power of two)
                sh = *reinterpret_cast<const x3_f32x4*>(shift + co);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int ch = h0 + 2 * t + (n >> 3), cw = w0 + (n & 7);
                if (ch >= dm.Hg || cw >= dm.Wg) continue;
                const int od = KIND == DP_T2 ? 2 * dg + pd : dg, oh = KIND == DP_T2 ? 2 * ch + ph : ch, ow = KIND == DP_T2 ? 2 * cw + pw : cw;
                const long long ov = ((((long long)b * dm.Do + od) * dm.Ho + oh) * dm.Wo + ow) * COUT + co;
                x3_f32x4 v = tot[t] * sc + sh;
                if (dm.relu) v = __builtin_elementwise_max(v, (x3_f32x4){0.f, 0.f, 0.f, 0.f});
                if (res) v += *reinterpret_cast<const x3_f32x4*>(res + ov);
                *reinterpret_cast<x3_f32x4*>(y + ov) = v;
                vmax = x3_absmax4(vmax, v);
            }
        }
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

// ---- host side -------------------------------------------------------------------------------------------------------
// (32 -> 32 stride 1 = conv4, one level up: 15 - 41 k cells; the z-marching kernel, whose blocks have 2 - 12 planes to march over, took 14.7 / 20.7 us
// per launch, this one 8.9 / 17.3: -13 us per scene, +1 % in bench.py on one box; 32 -> 16 transposed = conv9: 17.4 / 25.1 -> 8.8 / 18.5 us, +2 %;
// 16 -> 32 stride 2 = conv3, a k-step of two taps x 16 channels: 11.6 / 16.7 -> 7.9 / 15.2 us, +0.6 %)
#define RCMVS_DEEP_LIST(X) X(32, 64, DP_S2) X(64, 64, DP_S1) X(64, 32, DP_T2) X(32, 32, DP_S1) X(32, 16, DP_T2) X(16, 32, DP_S2)

bool conv3d_deep_supported(int Ci, int Co, int kind) {
    if (Ci == 16 && Co == 32) {                    // RCMVS_DEEP3=0: conv3 (16 -> 32 stride 2) stays on the z-marching kernel (A/B)
        static const bool on = [] { const char* e = getenv("RCMVS_DEEP3"); return !e || e[0] != '0'; }();
        if (!on) return false;
    }
    if (Ci == 32 && Co == 16) {                    // RCMVS_DEEP9=0: conv9 (32 -> 16 transposed) stays on the z-marching kernel (A/B)
        static const bool on = [] { const char* e = getenv("RCMVS_DEEP9"); return !e || e[0] != '0'; }();
        if (!on) return false;
    }
    if (Ci == 32 && Co == 32) {                    // RCMVS_DEEP4=0: conv4 stays on the z-marching kernel (A/B; read once: pack and launch must agree)
        static const bool on = [] { const char* e = getenv("RCMVS_DEEP4"); return !e || e[0] != '0'; }();
        if (!on) return false;
    }
#define DP_CASE(CI, CO, K) if (Ci == CI && Co == CO && kind == K) return true;
    RCMVS_DEEP_LIST(DP_CASE)
#undef DP_CASE
    return false;
}

long long conv3d_deep_weight_floats(int Ci, int Co, int kind) {
#define DP_CASE(CI, CO, K) if (Ci == CI && Co == CO && kind == K) return Deep<CI, CO, K>::IMG_HALFS / 2;
    RCMVS_DEEP_LIST(DP_CASE)
#undef DP_CASE
    return 0;
}

// wsc: two device floats {scale, 1 / scale} of the weight tensor (conv3d_x3_wscale)
int conv3d_deep_pack(const float* w, float* img, int Co, int Ci, int kind, int transposed, const float* wsc, hipStream_t st) {
#define DP_CASE(CI, CO, K) if (Ci == CI && Co == CO && kind == K) { \
        const int nthr = Deep<CI, CO, K>::KSTEPS * Deep<CI, CO, K>::MT * 512; \
        hipLaunchKernelGGL((deep_pack_kernel<CI, CO, K>), dim3((nthr + 255) / 256), dim3(256), 0, st, w, reinterpret_cast<unsigned short*>(img), transposed, wsc); \
        return launch_status("conv3d_deep_pack"); }
    RCMVS_DEEP_LIST(DP_CASE)
#undef DP_CASE
    return fail(-1, "conv3d_deep_pack: unsupported Ci=%d Co=%d kind=%d", Ci, Co, kind);
}

template <int CI, int CO, int K, int MS>
static int deep_launch_ms(const float* x, const float* wimg, const float* scale, const float* shift, const float* res, float* y,
                          const DeepDims& dm, long long blocks, int dev, const float* xmax, float* ymax, hipStream_t st) {
    using C = Deep<CI, CO, K, MS>;
    constexpr int MAXDEV = 64;
    static std::atomic<bool> raised[MAXDEV];                    // per device: dynamic-LDS limit of this instantiation
    if (C::LDS > 64 * 1024 && !raised[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)conv3d_deep_kernel<CI, CO, K, MS>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS) != hipSuccess)
            return fail(-1, "conv3d_deep: cannot raise the dynamic LDS limit to %d bytes", C::LDS);
        raised[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((conv3d_deep_kernel<CI, CO, K, MS>), dim3((unsigned)blocks, C::MSPLIT), dim3(512), C::LDS, st, x, reinterpret_cast<const x3_u32x4*>(wimg),
                       scale, shift, res, y, dm, xmax, ymax);
    return launch_status("conv3d_deep");
}

template <int CI, int CO, int K>
static int deep_launch_t(const float* x, const float* wimg, const float* scale, const float* shift, const float* res, float* y,
                         const DeepDims& dm_in, int dev, const float* xmax, float* ymax, hipStream_t st) {
    using C = Deep<CI, CO, K>;
    DeepDims dm = dm_in;
    dm.tiles_h = (dm.Hg + C::TH - 1) / C::TH; dm.tiles_w = (dm.Wg + C::TW - 1) / C::TW;
    const long long blocks = (long long)dm.B * dm.Dg * dm.tiles_h * dm.tiles_w;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return fail(-1, "conv3d_deep: bad grid");
    if constexpr (K != DP_T2 && CO == 64) {                      // two blocks per tile while they fit one round (one block per CU: 512 threads, > 80 KB of LDS)
        static std::atomic<int> cu_of[64];
        if (cu_of[dev] == 0) {
            hipDeviceProp_t prop;
            cu_of[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        }
        if (2 * blocks <= cu_of[dev].load()) return deep_launch_ms<CI, CO, K, 2>(x, wimg, scale, shift, res, y, dm, blocks, dev, xmax, ymax, st);
    }
    return deep_launch_ms<CI, CO, K, 1>(x, wimg, scale, shift, res, y, dm, blocks, dev, xmax, ymax, st);
}

int conv3d_deep_launch(const float* x, const float* wimg, const float* scale, const float* shift, const float* res, float* y,
                       int B, int D, int H, int W, int Ci, int Co, int kind, int relu, hipStream_t st, const float* xmax, float* ymax) {
    if (!xmax) return fail(-1, "conv3d_deep: the fp16-pair form needs a bound of max|x|");
    DeepDims dm;
    dm.B = B; dm.D = D; dm.H = H; dm.W = W; dm.relu = relu;
    if (kind == DP_T2) { dm.Do = 2 * D; dm.Ho = 2 * H; dm.Wo = 2 * W; dm.Dg = D; dm.Hg = H; dm.Wg = W; }
    else {
        const int s = kind == DP_S2 ? 2 : 1;
        dm.Do = (D - 1) / s + 1; dm.Ho = (H - 1) / s + 1; dm.Wo = (W - 1) / s + 1;
        dm.Dg = dm.Do; dm.Hg = dm.Ho; dm.Wg = dm.Wo;
    }
    if ((long long)B * D * H * W * Ci * 4 >= 0x7ffffff0LL) return fail(-1, "conv3d_deep: input tensor too large for 32-bit offsets");
    dm.tiles_h = dm.tiles_w = 0;                   // (per instantiation: deep_launch_t)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fail(-1, "conv3d_deep: cannot query the device");
#define DP_CASE(CI, CO, K) if (Ci == CI && Co == CO && kind == K) return deep_launch_t<CI, CO, K>(x, wimg, scale, shift, res, y, dm, dev, xmax, ymax, st);
    RCMVS_DEEP_LIST(DP_CASE)
#undef DP_CASE
    return fail(-1, "conv3d_deep: unsupported Ci=%d Co=%d kind=%d", Ci, Co, kind);
}

}  // namespace rcmvs
