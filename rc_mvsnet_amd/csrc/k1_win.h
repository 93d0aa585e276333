// K1, window form (round 5): fused plane-sweep warp + variance with the source footprint of a tile staged in LDS.
//
// Why: the gather kernel pulls 8 x 16 B of taps through the texture addresser / vector L1 for every 16 B it stores
// (64 B / clock / CU: ~100 us per scene for the loads alone).  The 2x2 footprints of a 4-row tile over a chunk of consecutive
// planes overlap almost completely, so the block loads the union ONCE -- wide, coalesced, straight into LDS
// (buffer_load ... lds, no VGPR round trip) -- and takes the taps from LDS (ds_read_b128: ~4x the L1's bandwidth).
//
// Structure of a block (256 threads, tile = PIX pixels x DKB planes, NVT source views):
//   origin   wave 0 evaluates a cheap approximate position of the tile's centre pixel at the chunk's middle plane and hands it to the
//            block through LDS (one early barrier): the window of view v is WR rows x WP texels around it.
//            MODE 1 issues the window loads here, so they fly while phase A computes.  A wave-load is one scalar row offset
//            plus a per-lane column offset computed once per chunk: one vector add per load.
//   phase A  one thread per (pixel, plane, view): exact position (k1_position), fixed-pattern taps (k1_tap_fixed), record =
//            four weights + the texel offset twice: in the feature map (gather path) and in the window.  A record "fits" if
//            its 2x2 footprint lies inside the window (records with four zero weights always fit).  The block takes the window
//            path only if every record of every wave fits (one ballot per wave, four flags in LDS): a per-tile fallback.
//   phase B  thread = (pixel, channel quad), per plane: record broadcast-read from LDS, four 16-byte taps per view from the
//            window (ds_read_b128, immediate offsets) or from the feature map (raw buffer loads: one offset register, the other
//            three taps are the instruction's immediate / scalar offsets), FMA blend, variance, non-temporal store.
// Positions are bit-identical to the reference-order kernel; the blend and the variance are FMA-contracted and the mean is a
// multiplication by 1/V: results differ from the reference-order kernel by a few 1e-7 of the value range (tests bound it).
// MODE 1 = window loads issued ahead of phase A (the production form for pixel-invariant hypothesis planes: stage 1 of the cascade),
// MODE 2 = issued after the fit test (no wasted loads on tiles that fall back).  Measurements: profiles/r5_k1_window.txt.
// Two source views only.  Round 6 built the any-view-count form (the views two at a time, each group with its own origins, windows, records
// and vote, the plane sums carried in registers: 168 VGPRs, three blocks per CU) and timed it at the reference's other view counts
// (profiles/r6_k1_views.txt): V = 4 73 us against 59 for the plane-pipelined gather form, V = 5 at 296 x 400 547 against 383 (38 % of the
// (tile, group) pairs fit the 16 x 8 window at that resolution), V = 7 1 118 against 721 -- it lost everywhere and was removed.
#pragma once
#include <atomic>
#include "k1_taps.h"

namespace rcmvs {

template <int C, int DKB, int NVT, int WP, int WR>
struct K1Win {
    static constexpr int LPP = C / 4;               // lanes per pixel (one float4 each)
    static constexpr int PIX = 256 / LPP;           // pixels per block
    static constexpr int TH = 4, TW = PIX / TH;     // tile: 4 rows x TW pixels, one wave per row
    static constexpr int NREC = NVT * DKB * PIX;    // records per chunk
    static constexpr int NCH = NREC / 256;          // chains per thread and chunk
    static constexpr int TEXB = C * 4;              // bytes per texel
    static constexpr int WBYTES = WR * WP * TEXB;   // window bytes per view
    static constexpr int NLD = NVT * WBYTES / 1024; // 1 KiB wave-loads per chunk
    static constexpr int LPV = NLD / NVT;           // ... per view
    static constexpr int TPW = 1024 / TEXB;         // texels per wave-load
    static constexpr int SEG = WP / TPW;            // wave-loads per window row
    // LDS layout (bytes)
    static constexpr int OFF_W = 0;                         // v4f weights[NREC]
    static constexpr int OFF_G = OFF_W + NREC * 16;         // int  feature-map byte offset[NREC]
    static constexpr int OFF_L = OFF_G + NREC * 4;          // int  window byte offset[NREC]
    static constexpr int OFF_F = OFF_L + NREC * 4;          // int  fit flags[4], then the window origins (x, y per view; + pad to 1 KiB)
    static constexpr int OFF_WIN = (OFF_F + 16 + 8 * NVT + 1023) / 1024 * 1024;
    static constexpr int LDS_BYTES = OFF_WIN + NVT * WBYTES;
    static_assert(NREC % 256 == 0, "records must divide over the threads");
    static_assert(WBYTES % 1024 == 0 && LPV % 4 == 0, "the window loads of a view must divide over the four waves");
    static_assert(WP % TPW == 0, "a wave-load must stay inside one window row");
};

template <int C, int DKB, int NVT, int WP, int WR, int MODE>
__global__ __launch_bounds__(256) void warp_variance_win_kernel(
    const float* __restrict__ feats, const float* __restrict__ rot, const float* __restrict__ trans,
    const float* __restrict__ planes, float* __restrict__ var, int V, int D, int h, int w, int tiles_x, unsigned* __restrict__ stats) {
    using W = K1Win<C, DKB, NVT, WP, WR>;
    constexpr int LPP = W::LPP, PIX = W::PIX, TH = W::TH, TW = W::TW, TEXB = W::TEXB;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    v4f* rec_w = reinterpret_cast<v4f*>(lds + W::OFF_W);
    int* rec_g = reinterpret_cast<int*>(lds + W::OFF_G);
    int* rec_l = reinterpret_cast<int*>(lds + W::OFF_L);
    int* fitf = reinterpret_cast<int*>(lds + W::OFF_F);
    int* orgf = fitf + 4;
    char* win = lds + W::OFF_WIN;
    const int b = blockIdx.z;
    const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tx0 = (tile % tiles_x) * TW, ty0 = (tile / tiles_x) * TH;
    const int hw = h * w;
    K1Geom g;
    g.w = w; g.h = h;
    g.wm1 = (float)(w - 1); g.hm1 = (float)(h - 1);
    g.half_w = g.wm1 / 2.0f; g.half_h = g.hm1 / 2.0f;
    g.r_half_w = rcp_nr(g.half_w); g.r_half_h = rcp_nr(g.half_h);
    const float* fb = feats + (long long)b * V * hw * C;
    // real size: window loads past the end of the last view rely on out-of-range lanes returning zero instead of faulting
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fb), (short)0, V * hw * TEXB, 0x00020000);
    const float* rotb = rot + (long long)b * (V - 1) * 9;
    const float* trb = trans + (long long)b * (V - 1) * 3;
    const int lane = threadIdx.x & 63;
    const int swave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));

    // the per-thread global loads first (plane table at the phase-A pixel, reference feature at the phase-B pixel, plane table at
    // the tile's centre), so that their latency is spent under the first origin / window loads
    const int pa = threadIdx.x % PIX;
    const int xa = min(tx0 + pa % TW, w - 1), ya = min(ty0 + pa / TW, h - 1);
    const float2 pla = reinterpret_cast<const float2*>(planes)[(long long)b * hw + ya * w + xa];
    const int p = threadIdx.x / LPP;
    const int q4b = (threadIdx.x % LPP) * 16;                    // byte offset of this lane's channel quad
    const int x = tx0 + p % TW, y = ty0 + p / TW;
    const bool inside = (x < w) && (y < h);
    const v4f ref = *reinterpret_cast<const v4f*>(fb + ((long long)min(y, h - 1) * w + min(x, w - 1)) * C + (q4b >> 2));
    const v4f ref2 = ref * ref;
    const float rV = rcp_nr((float)V);
    const long long pstride = (long long)hw * C;
    float* ob = var + ((long long)b * D * hw + (long long)min(y, h - 1) * w + min(x, w - 1)) * C + (q4b >> 2);
    const int pitch = w * TEXB;
    const int xcen = min(tx0 + TW / 2, w - 1), ycen = min(ty0 + TH / 2, h - 1);
    const float fxc = (float)xcen, fyc = (float)ycen, fxa = (float)xa, fya = (float)ya;
    float2 plc = make_float2(0.f, 0.f);
    plc = reinterpret_cast<const float2*>(planes)[(long long)b * hw + ycen * w + xcen];

    struct Taps { v4f t[NVT][4]; v4f w[NVT]; };

    {
        const int k0 = blockIdx.y * DKB;
        // ---------------- window origins (uniform) and, MODE 1, the window loads
        int ox[NVT], oy[NVT];
        auto fill = [&]() {
            const int rw = (4 % W::SEG == 0) ? swave / W::SEG : 0, sw = (4 % W::SEG == 0) ? swave % W::SEG : 0;
            // The byte offset of a wave-load is uniform (view, row, origin column + segment) + lane * 16 and may be NEGATIVE left of the
            // first row: a huge unsigned offset, which the descriptor's size turns into zeros (texels no record points at).  It all goes
            // into the VECTOR offset (one v_add with a scalar): the hardware adds the instruction's scalar offset zero-extended, so a
            // negative value there reads far outside the buffer (round 5: wrong results on 5x3 images; the emulation models it now).
#pragma unroll
            for (int j = 0; j < W::NLD / 4; ++j) {
                const int va = j / (W::LPV / 4), jj = j % (W::LPV / 4);   // compile-time: wave s takes loads s, s + 4, ... of every view
                int r, c0 = 0;
                if (4 % W::SEG == 0) r = rw + (4 / W::SEG) * jj;
                else { const int l = swave + 4 * jj; r = l / W::SEG; c0 = (l % W::SEG) * W::TPW; }
                const int L = va * W::LPV + swave + 4 * jj;         // wave-uniform load id = 1 KiB slot of the window
                // rows outside the image are never referenced (records point at clamped, in-image footprints): bring a valid row
                const int yy = min(max(oy[va] + r, 0), h - 1);
                const int off = (((va + 1) * h + yy) * w + ox[va] + c0 + ((4 % W::SEG == 0) ? sw * W::TPW : 0)) * TEXB;      // scalar, any sign
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, win + L * 1024, 16, lane * 16 + off, 0, 0, 0);
            }
        };
        // The origin must be ONE value for the whole block: the window rows are loaded by all four waves and every record is tested
        // against it.  Wave 0 computes it and hands it over through LDS, so that the block agrees by construction (four waves each
        // deriving it from their own loads of rot / trans / planes would agree only as long as those loads return the same bytes).
        if (swave == 0) {
            const float dmid = plc.x + ((float)k0 + 0.5f * (float)(DKB - 1)) * plc.y;
#pragma unroll
            for (int va = 0; va < NVT; ++va) {
                const float* r = rotb + va * 9;
                const float* t = trb + va * 3;
                const float pz = fmaf(fmaf(r[6], fxc, fmaf(r[7], fyc, r[8])), dmid, t[2]);
                const float rp = __builtin_amdgcn_rcpf(pz);
                const float cx = fmaf(fmaf(r[0], fxc, fmaf(r[1], fyc, r[2])), dmid, t[0]) * rp;
                const float cy = fmaf(fmaf(r[3], fxc, fmaf(r[4], fyc, r[5])), dmid, t[1]) * rp;
                // centre the window on the footprint of the tile's centre: clamp so that the int conversion is defined
                const float cxc = fminf(fmaxf(cx, -65536.0f), 65536.0f), cyc = fminf(fmaxf(cy, -65536.0f), 65536.0f);
                if (lane == 0) { orgf[2 * va] = (int)floorf(cxc) - (WP / 2 - 1); orgf[2 * va + 1] = (int)floorf(cyc) - (WR / 2 - 1); }
            }
        }
        __syncthreads();
#pragma unroll
        for (int va = 0; va < NVT; ++va) {
            ox[va] = __builtin_amdgcn_readfirstlane(orgf[2 * va]);
            oy[va] = __builtin_amdgcn_readfirstlane(orgf[2 * va + 1]);
        }
        if (MODE == 1) fill();

        // ---------------- phase A: one record per (pixel, plane, view)
        bool all_fit = true;
#pragma unroll
        for (int j = 0; j < W::NCH; ++j) {
#pragma clang fp contract(off)
            const int c = threadIdx.x + 256 * j;
            const int ka = (c / PIX) % DKB, va = c / (PIX * DKB);
            const float* r = rotb + va * 9;
            const float* t = trb + va * 3;
            const float rx = (r[0] * fxa + r[1] * fya) + r[2];
            const float ry = (r[3] * fxa + r[4] * fya) + r[5];
            const float rz = (r[6] * fxa + r[7] * fya) + r[8];
            const float d = pla.x + (float)(k0 + ka) * pla.y;
            float ix, iy;
            k1_position(rx, ry, rz, t[0], t[1], t[2], d, g, ix, iy);
            int xc, yc;
            v4f wt;
            bool live;
            k1_tap_fixed(ix, iy, g, xc, yc, wt, live);
            rec_w[c] = wt;
            rec_g[c] = (((va + 1) * h + yc) * w + xc) * TEXB;
            {
                int oxv = ox[0], oyv = oy[0];
#pragma unroll
                for (int q = 1; q < NVT; ++q) if (va == q) { oxv = ox[q]; oyv = oy[q]; }
                const int lx = xc - oxv, ly = yc - oyv;
                const bool fits = ((unsigned)lx < (unsigned)(WP - 1)) && ((unsigned)ly < (unsigned)(WR - 1));
                rec_l[c] = va * W::WBYTES + ((live && fits) ? (ly * WP + lx) * TEXB : 0);
                all_fit = all_fit && (fits || !live);
            }
        }
        bool use_win = false;
        {
            const int wave_fit = __all(all_fit);
            if (lane == 0) fitf[swave] = wave_fit;
            if (MODE == 1) __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this wave's window loads have landed
        }
        __syncthreads();
        {
            const v4i ff = *reinterpret_cast<const v4i*>(fitf);
            use_win = __builtin_amdgcn_readfirstlane((ff.x & ff.y & ff.z & ff.w) != 0);
            if (MODE == 2 && use_win) {
                fill();
                __builtin_amdgcn_s_waitcnt(0x0F70);
                __syncthreads();
            }
            if (stats && threadIdx.x == 0) { atomicAdd(stats, 1u); if (use_win) atomicAdd(stats + 1, 1u); }
        }

        // ---------------- phase B
        if (inside) {
            auto blend_store = [&](const Taps& f, int k) {
                v4f a = ref, a2 = ref2;
#pragma unroll
                for (int va = 0; va < NVT; ++va) {
                    const v4f wt = f.w[va];
                    v4f val = f.t[va][0] * wt.x;
                    val = __builtin_elementwise_fma(f.t[va][1], (v4f){wt.y, wt.y, wt.y, wt.y}, val);
                    val = __builtin_elementwise_fma(f.t[va][2], (v4f){wt.z, wt.z, wt.z, wt.z}, val);
                    val = __builtin_elementwise_fma(f.t[va][3], (v4f){wt.w, wt.w, wt.w, wt.w}, val);
                    a = a + val;
                    a2 = __builtin_elementwise_fma(val, val, a2);
                }
                if (k0 + k < D) {
                    const v4f m = a * rV;
                    const v4f o = __builtin_elementwise_fma(a2, (v4f){rV, rV, rV, rV}, -(m * m));
                    __builtin_nontemporal_store(o, reinterpret_cast<v4f*>(ob + (k0 + k) * pstride));
                }
            };
            Taps f0, f1;
            if (use_win) {
                auto issue = [&](Taps& f, int k) {
#pragma unroll
                    for (int va = 0; va < NVT; ++va) {
                        const int idx = (va * DKB + k) * PIX + p;
                        const char* base = win + rec_l[idx] + q4b;
                        f.w[va] = rec_w[idx];
                        f.t[va][0] = *reinterpret_cast<const v4f*>(base);
                        f.t[va][1] = *reinterpret_cast<const v4f*>(base + TEXB);
                        f.t[va][2] = *reinterpret_cast<const v4f*>(base + WP * TEXB);
                        f.t[va][3] = *reinterpret_cast<const v4f*>(base + WP * TEXB + TEXB);
                    }
                };
                issue(f0, 0);
#pragma unroll
                for (int k = 0; k < DKB; ++k) {
                    Taps& cur = (k & 1) ? f1 : f0;
                    Taps& nxt = (k & 1) ? f0 : f1;
                    if (k + 1 < DKB) issue(nxt, k + 1);
                    blend_store(cur, k);
                }
            } else {
                auto issue = [&](Taps& f, int k) {
#pragma unroll
                    for (int va = 0; va < NVT; ++va) {
                        const int idx = (va * DKB + k) * PIX + p;
                        const int o = rec_g[idx] + q4b;
                        f.w[va] = rec_w[idx];
                        f.t[va][0] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, 0, 0));
                        f.t[va][1] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o + TEXB, 0, 0));
                        f.t[va][2] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, pitch, 0));
                        f.t[va][3] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o + TEXB, pitch, 0));
                    }
                };
                issue(f0, 0);
#pragma unroll
                for (int k = 0; k < DKB; ++k) {
                    Taps& cur = (k & 1) ? f1 : f0;
                    Taps& nxt = (k & 1) ? f0 : f1;
                    if (k + 1 < DKB) issue(nxt, k + 1);
                    blend_store(cur, k);
                }
            }
        }
    }
}

// host side: launch one instantiation
template <int C, int DKB, int NVT, int WP, int WR, int MODE>
static int k1_win_launch_one(const float* feats, const float* rot, const float* trans, const float* planes, float* var,
                             int B, int V, int D, int h, int w, unsigned* stats, hipStream_t st, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr) {
    using W = K1Win<C, DKB, NVT, WP, WR>;
    const int txp = (w + W::TW - 1) / W::TW, typ = (h + W::TH - 1) / W::TH;
    dim3 grid(txp * typ, (D + DKB - 1) / DKB, B);
    const size_t lds = (size_t)W::LDS_BYTES;
    auto kern = warp_variance_win_kernel<C, DKB, NVT, WP, WR, MODE>;
    if (lds > 48 * 1024) {             // (the debug geometries of C = 8; the production launch of stage 1 needs 39 KB)
        static std::atomic<bool> raised[64];        // per instantiation and device
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev >= 0 && dev < 64 && !raised[dev].load(std::memory_order_acquire)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return fail(-1, "warp_variance_fwd (window form): cannot raise the dynamic LDS limit to %zu bytes", lds);
            raised[dev].store(true, std::memory_order_release);
        }
    }
    RCMVS_LAUNCH_TIMED(kern, grid, dim3(256), lds, st, ev0, ev1, feats, rot, trans, planes, var, V, D, h, w, txp, stats);
    return launch_status("warp_variance_fwd (window form)");
}

}  // namespace rcmvs
