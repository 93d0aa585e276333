// Experiments of the K1 lab (not product code).  lab_launch(variant >= 20, ...).
#pragma once
namespace rcmvs {

// ---- ablations of the production kernel (NVT = 2): 20 = as is, 21 = no gathers (tap data made from the offsets), 22 = no coordinate
// chains (own pixel, constant weights), 23 = no stores, 24 = 21 + 22
template <int C, int DKB, int ABL, int THP = 4>
__global__ __launch_bounds__(256, (ABL == 33 ? 5 : 1)) void lab_tp_kernel(
    const float* __restrict__ feats, const float* __restrict__ rot, const float* __restrict__ trans,
    const float* __restrict__ planes, float* __restrict__ var, int V, int D, int h, int w, int tiles_x) {
#pragma clang fp contract(off)
    constexpr int NVT = 2;
    constexpr int LPP = C / 4;
    constexpr int PIX = 256 / LPP;
    constexpr int TH = THP, TW = PIX / TH;
    constexpr int GRP = 256 / PIX;
    constexpr int KPT = DKB / GRP;
    constexpr bool NOG = (ABL == 21 || ABL == 24), NOA = (ABL == 22 || ABL == 24), NOS = (ABL == 23);
    constexpr bool SMALLST = (ABL == 28), AHEAD2 = (ABL == 29), PLAINST = (ABL == 30), BUFST = (ABL == 31 || ABL == 32);
    constexpr bool PRIO_B = (ABL == 25), PRIO_A = (ABL == 26), PRIO_LD = (ABL == 27);
    if (PRIO_A) __builtin_amdgcn_s_setprio(3);
    extern __shared__ __attribute__((aligned(16))) v4i lds_o[];
    v4f* lds_w = reinterpret_cast<v4f*>(lds_o + NVT * DKB * PIX);
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
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fb), (short)0, 0x7fffffff, 0x00020000);
    const int p = threadIdx.x / LPP;
    const int q4b = (threadIdx.x % LPP) * 16;
    const int x = tx0 + p % TW, y = ty0 + p / TW;
    const bool inside = (x < w) && (y < h);
    v4f ref = (v4f){0.f, 0.f, 0.f, 0.f};
    if (inside) ref = *reinterpret_cast<const v4f*>(fb + ((long long)y * w + x) * C + (q4b >> 2));
    const float fV = (float)V, rV = rcp_nr(fV);
    float* ob = var + (((long long)b * D) * hw + (long long)y * w + x) * C + (q4b >> 2);
    if (SMALLST) ob = var + ((((long long)y * w + x) * C + (q4b >> 2)) & 0x7ffff);
    __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(var + (long long)b * D * hw * C, (short)0, 0x7fffffff, 0x00020000);
    const int pa = threadIdx.x % PIX, ga = threadIdx.x / PIX;
    const int xa = min(tx0 + pa % TW, w - 1), ya = min(ty0 + pa / TW, h - 1);
    const float fxa = (float)xa, fya = (float)ya;
    const float2 pla = reinterpret_cast<const float2*>(planes)[(long long)b * hw + ya * w + xa];
    for (int va = 0; va < NVT; ++va) {
        const float* r = rot + ((long long)b * (V - 1) + va) * 9;
        const float* t = trans + ((long long)b * (V - 1) + va) * 3;
        const float rx = (r[0] * fxa + r[1] * fya) + r[2];
        const float ry = (r[3] * fxa + r[4] * fya) + r[5];
        const float rz = (r[6] * fxa + r[7] * fya) + r[8];
        const float t0 = t[0], t1 = t[1], t2 = t[2];
        const int vrow = (1 + va) * hw;
#pragma unroll
        for (int kk = 0; kk < KPT; ++kk) {
            const int ka = ga + kk * GRP;
            const float d = pla.x + (float)(k0 + ka) * pla.y;
            v4i o;
            v4f wt;
            if (NOA) {
                const int base = (vrow + ya * w + xa) * (C * 4);
                o = (v4i){base, base, base, base};
                wt = (v4f){0.25f, 0.25f, 0.25f, d};
            } else {
                k1_tap<C>(rx, ry, rz, t0, t1, t2, d, g, vrow, o, wt);
            }
            const int idx = (va * DKB + ka) * PIX + pa;
            lds_o[idx] = o;
            lds_w[idx] = wt;
        }
    }
    __syncthreads();
    if (PRIO_A) __builtin_amdgcn_s_setprio(0);
    if (PRIO_B) __builtin_amdgcn_s_setprio(3);
    if (!inside) return;
    K1Fetch<NVT> f0, f1;
    auto issue = [&](K1Fetch<NVT>& f, int k) {
#pragma unroll
        for (int va = 0; va < NVT; ++va) {
            const int idx = (va * DKB + k) * PIX + p;
            const v4i o = lds_o[idx];
            f.w[va] = lds_w[idx];
            if (NOG) {
                f.t[va][0] = (v4f){(float)(o.x + q4b), 1.f, 2.f, 3.f};
                f.t[va][1] = (v4f){(float)(o.y + q4b), 1.f, 2.f, 3.f};
                f.t[va][2] = (v4f){(float)(o.z + q4b), 1.f, 2.f, 3.f};
                f.t[va][3] = (v4f){(float)(o.w + q4b), 1.f, 2.f, 3.f};
            } else {
                f.t[va][0] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.x + q4b, 0, 0));
                f.t[va][1] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.y + q4b, 0, 0));
                f.t[va][2] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.z + q4b, 0, 0));
                f.t[va][3] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.w + q4b, 0, 0));
            }
        }
    };
    K1Fetch<NVT> f2;
    issue(f0, 0);
    if (AHEAD2) issue(f1, 1);
#pragma unroll
    for (int k = 0; k < DKB; ++k) {
        K1Fetch<NVT>& cur = AHEAD2 ? ((k % 3 == 0) ? f0 : (k % 3 == 1) ? f1 : f2) : ((k & 1) ? f1 : f0);
        K1Fetch<NVT>& nxt = AHEAD2 ? (((k + 2) % 3 == 0) ? f0 : ((k + 2) % 3 == 1) ? f1 : f2) : ((k & 1) ? f0 : f1);
        if (PRIO_LD) __builtin_amdgcn_s_setprio(3);
        if (AHEAD2) { if (k + 2 < DKB) issue(nxt, k + 2); } else
        if (k + 1 < DKB) issue(nxt, k + 1);
        if (PRIO_LD) __builtin_amdgcn_s_setprio(0);
        v4f a = ref, a2 = ref * ref;
#pragma unroll
        for (int va = 0; va < NVT; ++va) {
            v4f val = blend4<false>(cur.t[va][0], cur.t[va][1], cur.t[va][2], cur.t[va][3], cur.w[va]);
            a = a + val;
            a2 = a2 + val * val;
        }
        if (k0 + k < D) {
            if (NOS) { if (a.x == 123456.75f) k1_store_variance<false>(a, a2, fV, rV, ob + (long long)(k0 + k) * hw * C); }
            else if (PLAINST || BUFST) {
                v4f m, o;
                m.x = div_c<1>(a.x, fV, rV); m.y = div_c<1>(a.y, fV, rV); m.z = div_c<1>(a.z, fV, rV); m.w = div_c<1>(a.w, fV, rV);
                o.x = div_c<1>(a2.x, fV, rV) - m.x * m.x; o.y = div_c<1>(a2.y, fV, rV) - m.y * m.y;
                o.z = div_c<1>(a2.z, fV, rV) - m.z * m.z; o.w = div_c<1>(a2.w, fV, rV) - m.w * m.w;
                if (PLAINST) *reinterpret_cast<v4f*>(ob + (long long)(k0 + k) * hw * C) = o;
                else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, o), orsrc, (((y * w + x) * C) << 2) + q4b, (k0 + k) * hw * C * 4, ABL == 32 ? 2 : 0);
            }
            else k1_store_variance<false>(a, a2, fV, rV, ob + (SMALLST ? 0 : (long long)(k0 + k) * hw * C));
        }
    }
}

}  // namespace rcmvs

static int lab_launch(int variant, const float* feats, const float* rot, const float* trans, const float* planes, float* var,
                      int B, int V, int C, int D, int h, int w, hipStream_t st) {
    using namespace rcmvs;
    if (variant >= 20 && variant <= 33) {
        const int LPP = C / 4, PIX = 256 / LPP;
        const int dkb = (C == 8) ? 4 : 8;
        const size_t lds = (size_t)32 * dkb * PIX * 2;
        const int TWp = PIX / 4;
        const int txp = (w + TWp - 1) / TWp, typ = (h + 3) / 4;
        dim3 grid(txp * typ, (D + dkb - 1) / dkb, B);
#define LAB_TP(CC, DD, AA) hipLaunchKernelGGL((lab_tp_kernel<CC, DD, AA>), grid, dim3(256), lds, st, feats, rot, trans, planes, var, V, D, h, w, txp)
#define LAB_TP_A(CC, DD) do { switch (variant) { case 20: LAB_TP(CC, DD, 20); break; case 21: LAB_TP(CC, DD, 21); break; case 22: LAB_TP(CC, DD, 22); break; \
                                                 case 23: LAB_TP(CC, DD, 23); break; case 24: LAB_TP(CC, DD, 24); break; case 25: LAB_TP(CC, DD, 25); break; case 26: LAB_TP(CC, DD, 26); break; case 27: LAB_TP(CC, DD, 27); break; case 28: LAB_TP(CC, DD, 28); break; case 29: LAB_TP(CC, DD, 29); break; case 30: LAB_TP(CC, DD, 30); break; case 31: LAB_TP(CC, DD, 31); break; case 32: LAB_TP(CC, DD, 32); break; default: LAB_TP(CC, DD, 33); break; } } while (0)
        if (C == 8) LAB_TP_A(8, 4); else if (C == 16) LAB_TP_A(16, 8); else LAB_TP_A(32, 8);
        return launch_status("lab_tp");
    }
    if (variant == 34 || variant == 35 || variant == 36) {
        const int LPP = C / 4, PIX = 256 / LPP;
        const int dkb = (C == 8) ? 4 : 8;
        const size_t lds = (size_t)32 * dkb * PIX * 2;
        const int th = variant == 34 ? 2 : (variant == 35 ? 8 : 16);
        const int TWp = PIX / th;
        if (TWp < 1) return fail(-1, "tile too tall");
        const int txp = (w + TWp - 1) / TWp, typ = (h + th - 1) / th;
        dim3 grid(txp * typ, (D + dkb - 1) / dkb, B);
#define LAB_TH(CC, DD, TT) hipLaunchKernelGGL((lab_tp_kernel<CC, DD, 20, TT>), grid, dim3(256), lds, st, feats, rot, trans, planes, var, V, D, h, w, txp)
#define LAB_TH_T(CC, DD) do { if (th == 2) LAB_TH(CC, DD, 2); else if (th == 8) LAB_TH(CC, DD, 8); else LAB_TH(CC, DD, 16); } while (0)
        if (C == 8) LAB_TH_T(8, 4); else if (C == 16) LAB_TH_T(16, 8); else LAB_TH_T(32, 8);
        return launch_status("lab_th");
    }
    return fail(-1, "lab: unknown variant %d", variant);
}
