// Per-pixel arithmetic of the self-supervised photometric loss (losses/homography.py:6-200, losses/modules.py:6-82,
// losses/unsup_loss.py:14-94 of the reference).  Plain C++: unsup_loss.hip calls these from its kernels, and
// tests/test_unsup_loss_cpu.py compiles the same header with g++ into a loop harness that is compared with the oracle
// -- the harness checks this arithmetic on a machine without a GPU, it is not a product path.
//
// Layouts: images (B, H, W, 3) channels-last fp32, depth / mask (B, H, W).  One source view at a time.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define RCMVS_HD __host__ __device__ inline
#else
#define RCMVS_HD inline
#endif

namespace rcmvs {
namespace ul {

constexpr float SSIM_C1 = 0.01f * 0.01f;
constexpr float SSIM_C2 = 0.03f * 0.03f;

// ---------------------------------------------------------------------------------------------------------------------
// inverse warp.  coef[12] = { M (3x3 row-major), t (3) } with  p = M (x, y, 1)^T d + t,  M = K_ref R_rel K_ref^-1,
// t = K_ref t_rel  (homography.py:9-56 composed on the host in fp64; the reference keeps K_ref for the projection).
// ---------------------------------------------------------------------------------------------------------------------
struct Taps {
    int o00, o10, o01, o11;   // pixel offsets (y * W + x) of (y0,x0), (y1,x0), (y0,x1), (y1,x1), clamped into the image
    float fx, fy;             // x1c - x, y1c - y against the CLAMPED corner (homography.py:187-190)
    float mask;               // 1 when (x0 >= 0, x1 <= W-1, y0 >= 0, y0 <= H-1) before clamping (homography.py:148)
    float dxdd, dydd;         // d x / d depth, d y / d depth
};

RCMVS_HD int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

#pragma clang fp contract(off)
RCMVS_HD Taps inv_warp_taps(const float* cf, int xi, int yi, float d, int H, int W) {
    const float xf = (float)xi, yf = (float)yi;
    const float m0 = cf[0] * xf + cf[1] * yf + cf[2];
    const float m1 = cf[3] * xf + cf[4] * yf + cf[5];
    const float m2 = cf[6] * xf + cf[7] * yf + cf[8];
    const float px = m0 * d + cf[9], py = m1 * d + cf[10], pz = m2 * d + cf[11];
    const float den = pz + 1e-10f;
    const float xs = px / den, ys = py / den;
    Taps t;
    t.dxdd = (m0 - xs * m2) / den;
    t.dydd = (m1 - ys * m2) / den;
    // _spatial_transformer normalises to [-1, 1] and _bilinear_sample maps back (homography.py:112-113,144-145)
    float x = xs / (float)(W - 1) * 2.0f - 1.0f;
    float y = ys / (float)(H - 1) * 2.0f - 1.0f;
    x = (x + 1.0f) * ((float)W - 1.0f) / 2.0f;
    y = (y + 1.0f) * ((float)H - 1.0f) / 2.0f;
    // keep the float -> int conversion defined for wild coordinates (the reference's is not); such pixels are masked
    const float big = 1.0e9f;
    const float xfl = floorf(fminf(fmaxf(x, -big), big)), yfl = floorf(fminf(fmaxf(y, -big), big));
    const int x0 = (int)xfl, y0 = (int)yfl;
    const int x1 = x0 + 1, y1 = y0 + 1;
    t.mask = (x0 >= 0 && x1 <= W - 1 && y0 >= 0 && y0 <= H - 1) ? 1.0f : 0.0f;
    const int x0c = clampi(x0, 0, W - 1), x1c = clampi(x1, 0, W - 1);
    const int y0c = clampi(y0, 0, H - 1), y1c = clampi(y1, 0, H - 1);
    t.o00 = y0c * W + x0c; t.o10 = y1c * W + x0c; t.o01 = y0c * W + x1c; t.o11 = y1c * W + x1c;
    t.fx = (float)x1c - x;
    t.fy = (float)y1c - y;
    return t;
}

// bilinear value of channel c; img points at this batch item's (H, W, 3) image
RCMVS_HD float tap_value(const Taps& t, const float* img, int c) {
    const float wa = t.fx * t.fy, wb = t.fx * (1.0f - t.fy), wc = (1.0f - t.fx) * t.fy, wd = (1.0f - t.fx) * (1.0f - t.fy);
    return wa * img[t.o00 * 3 + c] + wb * img[t.o10 * 3 + c] + wc * img[t.o01 * 3 + c] + wd * img[t.o11 * 3 + c];
}

// d value / d depth of channel c (through the sampling position only; the weights are linear in fx, fy and
// d fx / d x = d fy / d y = -1)
RCMVS_HD float tap_ddepth(const Taps& t, const float* img, int c) {
    const float pa = img[t.o00 * 3 + c], pb = img[t.o10 * 3 + c], pc = img[t.o01 * 3 + c], pd = img[t.o11 * 3 + c];
    const float dfx = t.fy * pa + (1.0f - t.fy) * pb - t.fy * pc - (1.0f - t.fy) * pd;
    const float dfy = t.fx * pa - t.fx * pb + (1.0f - t.fx) * pc - (1.0f - t.fx) * pd;
    return -(dfx * t.dxdd + dfy * t.dydd);
}

// ---------------------------------------------------------------------------------------------------------------------
// photometric + gradient smooth-L1 and SSIM terms of one source view (modules.py:6-42,70-81)
// ---------------------------------------------------------------------------------------------------------------------
RCMVS_HD float sl1(float z) { const float a = fabsf(z); return a < 1.0f ? 0.5f * a * a : a - 0.5f; }
RCMVS_HD float sl1_grad(float z) { return z < -1.0f ? -1.0f : (z > 1.0f ? 1.0f : z); }

struct Ssim { float val, a, b, c; };   // val = clamp((1 - S) / 2, 0, 1);  d val / d y_p = (a + b y_p + c x_p) per window pixel

// x = reference image, y = warped image (the argument order of unsup_loss.py:72); p points at the window CENTRE
RCMVS_HD Ssim ssim_window(const float* x, const float* y, int W, int c) {
    float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const float xv = x[(dy * W + dx) * 3 + c], yv = y[(dy * W + dx) * 3 + c];
            sx += xv; sy += yv; sxx += xv * xv; syy += yv * yv; sxy += xv * yv;
        }
    const float k = 1.0f / 9.0f;
    const float mx = sx * k, my = sy * k;
    const float vx = sxx * k - mx * mx, vy = syy * k - my * my, cxy = sxy * k - mx * my;
    const float n1 = 2.0f * mx * my + SSIM_C1, n2 = 2.0f * cxy + SSIM_C2;
    const float d1 = mx * mx + my * my + SSIM_C1, d2 = vx + vy + SSIM_C2;
    const float S = (n1 * n2) / (d1 * d2);
    const float raw = (1.0f - S) / 2.0f;
    Ssim r;
    r.val = fminf(fmaxf(raw, 0.0f), 1.0f);
    const float gate = (raw >= 0.0f && raw <= 1.0f) ? -0.5f * k : 0.0f;          // d val / d S, times the 1/9 of the means
    const float inv = 1.0f / (d1 * d2);
    const float dS_dmy = (2.0f * mx * n2 - 2.0f * mx * n1) * inv - S * (2.0f * my / d1 - 2.0f * my / d2);
    const float dS_deyy = -S / d2;
    const float dS_dexy = 2.0f * n1 * inv;
    r.a = gate * dS_dmy; r.b = gate * 2.0f * dS_deyy; r.c = gate * dS_dexy;
    return r;
}

RCMVS_HD float mask_window(const float* m, int W) {
    float s = 0.f;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) s += m[dy * W + dx];
    return s * (1.0f / 9.0f);
}

// sums[0..3] += photometric, x-gradient, y-gradient smooth-L1 and SSIM contributions of pixel (yi, xi).
// warped / ref / mask point at this batch item.
RCMVS_HD void photo_terms(const float* warped, const float* ref, const float* mask, int yi, int xi, int H, int W, float* sums) {
    const int p = yi * W + xi;
    const float m0 = mask[p];
    for (int c = 0; c < 3; ++c) {
        const float a0 = warped[p * 3 + c] * m0, b0 = ref[p * 3 + c] * m0;
        sums[0] += sl1(a0 - b0);
        if (xi + 1 < W) {
            const float m1 = mask[p + 1];
            sums[1] += sl1((warped[(p + 1) * 3 + c] * m1 - a0) - (ref[(p + 1) * 3 + c] * m1 - b0));
        }
        if (yi + 1 < H) {
            const float m1 = mask[p + W];
            sums[2] += sl1((warped[(p + W) * 3 + c] * m1 - a0) - (ref[(p + W) * 3 + c] * m1 - b0));
        }
    }
    if (yi >= 1 && yi <= H - 2 && xi >= 1 && xi <= W - 2) {
        const float mw = mask_window(mask + p, W);
        for (int c = 0; c < 3; ++c) sums[3] += mw * ssim_window(ref + p * 3, warped + p * 3, W, c).val;
    }
}

// SSIM backward coefficients of the window centred at (yi, xi), 1 <= yi <= H-2, 1 <= xi <= W-2: out[9] =
// {a, b, c} x 3 channels, already multiplied by the window's mask average (k_ssim is applied by the consumer)
RCMVS_HD void ssim_coefs(const float* warped, const float* ref, const float* mask, int yi, int xi, int W, float* out) {
    const int p = yi * W + xi;
    const float mw = mask_window(mask + p, W);
    for (int c = 0; c < 3; ++c) {
        const Ssim s = ssim_window(ref + p * 3, warped + p * 3, W, c);
        out[c * 3 + 0] = mw * s.a; out[c * 3 + 1] = mw * s.b; out[c * 3 + 2] = mw * s.c;
    }
}

// helper: the masked difference image e = (warped - ref) * mask and its forward differences
RCMVS_HD float masked_diff(const float* warped, const float* ref, const float* mask, int p, int c) {
    return warped[p * 3 + c] * mask[p] - ref[p * 3 + c] * mask[p];
}
RCMVS_HD float fwd_diff(const float* warped, const float* ref, const float* mask, int p, int q, int c) {
    // (a(q) - a(p)) - (b(q) - b(p)) with the same association as photo_terms
    const float a0 = warped[p * 3 + c] * mask[p], b0 = ref[p * 3 + c] * mask[p];
    return (warped[q * 3 + c] * mask[q] - a0) - (ref[q * 3 + c] * mask[q] - b0);
}

// d (k[0] photo_sum + k[1] dx_sum + k[2] dy_sum + k[3] ssim_sum) / d warped[(yi, xi), c]
// coef = SSIM window coefficients (H-2, W-2, 9) of this batch item.
RCMVS_HD float photo_grad(const float* warped, const float* ref, const float* mask, const float* coef, const float* k,
                          int yi, int xi, int c, int H, int W) {
    const int p = yi * W + xi;
    float ga = k[0] * sl1_grad(masked_diff(warped, ref, mask, p, c));
    if (xi + 1 < W) ga -= k[1] * sl1_grad(fwd_diff(warped, ref, mask, p, p + 1, c));
    if (xi >= 1)    ga += k[1] * sl1_grad(fwd_diff(warped, ref, mask, p - 1, p, c));
    if (yi + 1 < H) ga -= k[2] * sl1_grad(fwd_diff(warped, ref, mask, p, p + W, c));
    if (yi >= 1)    ga += k[2] * sl1_grad(fwd_diff(warped, ref, mask, p - W, p, c));
    float g = ga * mask[p];
    if (k[3] != 0.0f) {
        const float yv = warped[p * 3 + c], xv = ref[p * 3 + c];
        float s = 0.f;
        for (int wy = yi - 1; wy <= yi + 1; ++wy) {
            if (wy < 1 || wy > H - 2) continue;
            for (int wx = xi - 1; wx <= xi + 1; ++wx) {
                if (wx < 1 || wx > W - 2) continue;
                const float* q = coef + ((wy - 1) * (W - 2) + (wx - 1)) * 9 + c * 3;
                s += q[0] + q[1] * yv + q[2] * xv;
            }
        }
        g += k[3] * s;
    }
    return g;
}

// ---------------------------------------------------------------------------------------------------------------------
// image-aware depth smoothness (modules.py:56-67, lambda = 1)
// ---------------------------------------------------------------------------------------------------------------------
RCMVS_HD float edge_weight(const float* img, int p, int q) {
    const float s = fabsf(img[p * 3] - img[q * 3]) + fabsf(img[p * 3 + 1] - img[q * 3 + 1]) + fabsf(img[p * 3 + 2] - img[q * 3 + 2]);
    return expf(-(s / 3.0f));
}
RCMVS_HD float signf(float v) { return v > 0.f ? 1.0f : (v < 0.f ? -1.0f : 0.0f); }

RCMVS_HD void smooth_terms(const float* depth, const float* img, int yi, int xi, int H, int W, float* sums) {
    const int p = yi * W + xi;
    if (xi + 1 < W) sums[0] += fabsf((depth[p] - depth[p + 1]) * edge_weight(img, p, p + 1));
    if (yi + 1 < H) sums[1] += fabsf((depth[p] - depth[p + W]) * edge_weight(img, p, p + W));
}

// d (k[0] sum_x + k[1] sum_y) / d depth[(yi, xi)]
RCMVS_HD float smooth_grad(const float* depth, const float* img, const float* k, int yi, int xi, int H, int W) {
    const int p = yi * W + xi;
    float g = 0.f;
    if (xi + 1 < W) { const float w = edge_weight(img, p, p + 1); g += k[0] * signf((depth[p] - depth[p + 1]) * w) * w; }
    if (xi >= 1)    { const float w = edge_weight(img, p - 1, p); g -= k[0] * signf((depth[p - 1] - depth[p]) * w) * w; }
    if (yi + 1 < H) { const float w = edge_weight(img, p, p + W); g += k[1] * signf((depth[p] - depth[p + W]) * w) * w; }
    if (yi >= 1)    { const float w = edge_weight(img, p - W, p); g -= k[1] * signf((depth[p - W] - depth[p]) * w) * w; }
    return g;
}

// ---------------------------------------------------------------------------------------------------------------------
// per-view scalar loss and the best-view selection (unsup_loss.py:66-88)
// ---------------------------------------------------------------------------------------------------------------------
// L_v = 0.5 photo + 0.5 (grad_x + grad_y) from this view's sums over a (B, H, W, 3) image
RCMVS_HD float view_loss(const double* s, int B, int H, int W) {
    const double n = (double)B * H * W * 3, nx = (double)B * H * (W - 1) * 3, ny = (double)B * (H - 1) * W * 3;
    return (float)(0.5 * (s[0] / n) + 0.5 * (s[1] / nx + s[2] / ny));
}

// index of the view that wins pixel p, or -1 when every view is masked there
RCMVS_HD int best_view(const float* L, const float* masks, long long plane, long long p, int Vs) {
    int best = -1;
    float bv = 0.f;
    for (int v = 0; v < Vs; ++v) {
        const float val = L[v] + 1.0e4f * (1.0f - masks[v * plane + p]);
        if (best < 0 || val < bv) { best = v; bv = val; }
    }
    return bv < 1.0e4f ? best : -1;
}

}  // namespace ul
}  // namespace rcmvs
