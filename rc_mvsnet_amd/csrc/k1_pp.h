// K1, plane-pipelined gather form (round 5): fused plane-sweep warp + variance for per-pixel hypothesis planes (stages 2 / 3).
//
// Why: the two-phase kernel (warp_variance_tp_kernel) computes ALL coordinate chains of its chunk first (pure VALU, ~1 us), then gathers
// (texture addresser / L1 bound).  The resident blocks of a CU start together and take the same time, so they stay in step: while they
// compute nothing is loaded, while they gather the VALU idles -- the kernel's time is the SUM of its VALU and its TA time
// (profiles/r4_k1_walls.txt: "the parts add up almost without overlap").  Here the two are interleaved plane by plane INSIDE every
// wave: the gathers of plane k are issued, the coordinate chains of plane k + 1 are computed while they fly, then plane k is
// blended and stored.  The tap records of one plane (20 bytes per pixel and view) are double-buffered in LDS: 2.5-10 KB per block.
//
//   records   thread c of the plane's PIX * NVT chains: exact position (k1_position), fixed-pattern taps (k1_tap_fixed) ->
//             four weights + one byte offset.  With fewer chains per plane than threads (C = 16: 128, C = 32: 64) the thread groups
//             take the planes in turn.  rot * (x, y, 1) is computed once per block (it does not depend on the plane).
//   gathers   thread = (pixel, channel quad): record broadcast-read from LDS, four raw buffer loads per view (one offset register;
//             the other taps are the instruction's immediate / scalar offsets), FMA blend, variance, non-temporal store.
// Positions are bit-identical to the reference-order kernel; blend and variance are FMA-contracted (~4e-7 of the value range).
#pragma once
#include "k1_taps.h"

namespace rcmvs {

// View counts (round 6): NVT = 2 / 3 / 4 / 6 source views (V = 3 DTU bench, 4 training, 5 DTU evaluation, 7 Tanks and Temples).  The gathers of
// a plane are issued in groups of NG views (all of them up to four; 3 + 3 for six: two full tap sets of six views exceed the register
// file); the views are accumulated in ascending order whatever the grouping.  Chains per plane that do not tile the block (C = 8 with
// three views: 384; C = 16 / 32 with three or six: 192 / 96) leave the surplus threads of the record phase idle.
template <int C, int DKB, int NVT, int NG = (NVT <= 4 ? NVT : 3)>
__global__ __launch_bounds__(256) void warp_variance_pp_kernel(
    const float* __restrict__ feats, const float* __restrict__ rot, const float* __restrict__ trans,
    const float* __restrict__ planes, float* __restrict__ var, int V, int D, int h, int w, int tiles_x) {
    constexpr int LPP = C / 4, PIX = 256 / LPP, TH = 4, TW = PIX / TH, TEXB = C * 4;
    constexpr int CPP = PIX * NVT;                       // chains (= records) per plane
    constexpr int NCH = (CPP + 255) / 256;               // chains per thread and plane when every thread works on every plane
    constexpr int GROUPS = (CPP < 256) ? 256 / CPP : 1;  // thread groups that take the planes in turn otherwise
    constexpr int NGRP = NVT / NG;
    static_assert(NVT % NG == 0, "view groups must divide the view count");
    __shared__ __attribute__((aligned(16))) v4f rec_w[2][CPP];
    __shared__ int rec_g[2][CPP];
    const int b = blockIdx.z;
    const int k0 = blockIdx.y * DKB;
    const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tx0 = (tile % tiles_x) * TW, ty0 = (tile / tiles_x) * TH;
    const int hw = h * w;
    K1Geom g;
    g.w = w; g.h = h;
    g.wm1 = (float)(w - 1); g.hm1 = (float)(h - 1);
    g.half_w = g.wm1 / 2.0f; g.half_h = g.hm1 / 2.0f;
    g.r_half_w = rcp_nr(g.half_w); g.r_half_h = rcp_nr(g.half_h);
    const float* fb = feats + (long long)b * V * hw * C;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fb), (short)0, V * hw * TEXB, 0x00020000);
    const float* rotb = rot + (long long)b * (V - 1) * 9;
    const float* trb = trans + (long long)b * (V - 1) * 3;
    // ---- record identity of this thread: chain c = (pixel pa, view va); the plane-independent part of the chain
    const int pa = threadIdx.x % PIX;
    const int xa = min(tx0 + pa % TW, w - 1), ya = min(ty0 + pa / TW, h - 1);
    const float2 pla = reinterpret_cast<const float2*>(planes)[(long long)b * hw + ya * w + xa];
    const int grp = threadIdx.x / CPP;                   // (0 when CPP >= 256; >= GROUPS: a surplus thread of the record phase)
    float rx[NCH], ry[NCH], rz[NCH], t0[NCH], t1[NCH], t2[NCH];
    {
#pragma clang fp contract(off)
        const float fxa = (float)xa, fya = (float)ya;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = (threadIdx.x % CPP) + 256 * j;
            const int va = min(c / PIX, NVT - 1);                // (surplus chains of a ragged plane: any valid view, never stored)
            const float* r = rotb + va * 9;
            const float* t = trb + va * 3;
            rx[j] = (r[0] * fxa + r[1] * fya) + r[2];
            ry[j] = (r[3] * fxa + r[4] * fya) + r[5];
            rz[j] = (r[6] * fxa + r[7] * fya) + r[8];
            t0[j] = t[0]; t1[j] = t[1]; t2[j] = t[2];
        }
    }
    auto records = [&](int k, int buf) {                 // the records of plane k0 + k -> rec_*[buf]
        if (CPP < 256 && (k % GROUPS) != grp) return;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
#pragma clang fp contract(off)
            const int c = (threadIdx.x % CPP) + 256 * j;
            if (CPP % 256 != 0 && CPP > 256 && c >= CPP) continue;
            const int va = c / PIX;
            const float d = pla.x + (float)(k0 + k) * pla.y;
            float ix, iy;
            k1_position(rx[j], ry[j], rz[j], t0[j], t1[j], t2[j], d, g, ix, iy);
            int xc, yc;
            v4f wt;
            bool live;
            k1_tap_fixed(ix, iy, g, xc, yc, wt, live);
            rec_w[buf][c] = wt;
            rec_g[buf][c] = (((va + 1) * h + yc) * w + xc) * TEXB;
        }
    };
    // ---- gather identity
    const int p = threadIdx.x / LPP;
    const int q4b = (threadIdx.x % LPP) * 16;            // byte offset of this lane's channel quad
    const int x = tx0 + p % TW, y = ty0 + p / TW;
    const bool inside = (x < w) && (y < h);
    const v4f ref = *reinterpret_cast<const v4f*>(fb + ((long long)min(y, h - 1) * w + min(x, w - 1)) * C + (q4b >> 2));
    const v4f ref2 = ref * ref;
    const float rV = rcp_nr((float)V);
    const long long pstride = (long long)hw * C;
    float* ob = var + (((long long)b * D + k0) * hw + (long long)min(y, h - 1) * w + min(x, w - 1)) * C + (q4b >> 2);
    const int pitch = w * TEXB;

    records(0, 0);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < DKB; ++k) {
        v4f a = ref, a2 = ref2;
#pragma unroll
        for (int gi = 0; gi < NGRP; ++gi) {
            v4f tp[NG][4], wt[NG];
            if (inside) {
#pragma unroll
                for (int vg = 0; vg < NG; ++vg) {
                    const int va = gi * NG + vg;
                    const int o = rec_g[k & 1][va * PIX + p] + q4b;
                    wt[vg] = rec_w[k & 1][va * PIX + p];
                    tp[vg][0] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, 0, 0));
                    tp[vg][1] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o + TEXB, 0, 0));
                    tp[vg][2] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, pitch, 0));
                    tp[vg][3] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o + TEXB, pitch, 0));
                }
            }
            if (gi == 0 && k + 1 < DKB) records(k + 1, (k + 1) & 1);    // ... while the gathers fly
            if (inside) {
#pragma unroll
                for (int vg = 0; vg < NG; ++vg) {
                    v4f val = tp[vg][0] * wt[vg].x;
                    val = __builtin_elementwise_fma(tp[vg][1], (v4f){wt[vg].y, wt[vg].y, wt[vg].y, wt[vg].y}, val);
                    val = __builtin_elementwise_fma(tp[vg][2], (v4f){wt[vg].z, wt[vg].z, wt[vg].z, wt[vg].z}, val);
                    val = __builtin_elementwise_fma(tp[vg][3], (v4f){wt[vg].w, wt[vg].w, wt[vg].w, wt[vg].w}, val);
                    a = a + val;
                    a2 = __builtin_elementwise_fma(val, val, a2);
                }
            }
        }
        if (inside && k0 + k < D) {
            const v4f m = a * rV;
            const v4f o = __builtin_elementwise_fma(a2, (v4f){rV, rV, rV, rV}, -(m * m));
            __builtin_nontemporal_store(o, reinterpret_cast<v4f*>(ob + k * pstride));
        }
        if (k + 1 < DKB) __syncthreads();
    }
}

template <int C, int DKB, int NVT>
static int k1_pp_launch_one(const float* feats, const float* rot, const float* trans, const float* planes, float* var,
                            int B, int V, int D, int h, int w, hipStream_t st, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr) {
    constexpr int PIX = 256 / (C / 4), TW = PIX / 4;
    const int txp = (w + TW - 1) / TW, typ = (h + 3) / 4;
    dim3 grid(txp * typ, (D + DKB - 1) / DKB, B);
    RCMVS_LAUNCH_TIMED((warp_variance_pp_kernel<C, DKB, NVT>), grid, dim3(256), 0, st, ev0, ev1, feats, rot, trans, planes, var, V, D, h, w, txp);
    return launch_status("warp_variance_fwd (plane-pipelined form)");
}

}  // namespace rcmvs
