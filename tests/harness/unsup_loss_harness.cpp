// CPU loop harness around rc_mvsnet_amd/csrc/unsup_loss_math.h: the same per-pixel arithmetic the HIP kernels of
// unsup_loss.hip execute, driven by plain loops so tests/test_unsup_loss_cpu.py can compare it with the oracle on a
// machine without a GPU.  Test infrastructure only -- nothing in the package loads this.
#include <cstring>
#include <vector>
#include "../../rc_mvsnet_amd/csrc/unsup_loss_math.h"

using namespace rcmvs;

extern "C" void h_inverse_warp(const float* src, const float* depth, const float* coef, float* warped, float* mask, int B, int H, int W) {
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const long long p = ((long long)b * H + y) * W + x;
                const ul::Taps t = ul::inv_warp_taps(coef + b * 12, x, y, depth[p], H, W);
                for (int c = 0; c < 3; ++c) warped[p * 3 + c] = ul::tap_value(t, src + (long long)b * H * W * 3, c);
                mask[p] = t.mask;
            }
}

extern "C" void h_unsup_loss_fwd(const float* ref, const float* srcs, const float* depth, const float* coef, float* warped,
                                 float* masks, double* sums, int* counts, float* out, int B, int Vs, int H, int W) {
    const long long n = (long long)B * H * W;
    std::memset(sums, 0, sizeof(double) * (Vs * 4 + 2));
    std::memset(counts, 0, sizeof(int) * Vs);
    for (int v = 0; v < Vs; ++v) {
        h_inverse_warp(srcs + v * n * 3, depth, coef + v * B * 12, warped + v * n * 3, masks + v * n, B, H, W);
        for (int b = 0; b < B; ++b)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    float part[4] = {0, 0, 0, 0};
                    const long long o = (long long)b * H * W;
                    ul::photo_terms(warped + (v * n + o) * 3, ref + o * 3, masks + v * n + o, y, x, H, W, part);
                    for (int t = 0; t < 4; ++t) sums[v * 4 + t] += part[t];
                }
    }
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float part[2] = {0, 0};
                const long long o = (long long)b * H * W;
                ul::smooth_terms(depth + o, ref + o * 3, y, x, H, W, part);
                sums[Vs * 4] += part[0]; sums[Vs * 4 + 1] += part[1];
            }
    std::vector<float> L(Vs);
    for (int v = 0; v < Vs; ++v) L[v] = ul::view_loss(sums + v * 4, B, H, W);
    for (long long p = 0; p < n; ++p) {
        const int best = ul::best_view(L.data(), masks, n, p, Vs);
        if (best >= 0) counts[best]++;
    }
    double rec = 0, ssim = 0;
    for (int v = 0; v < Vs; ++v) {
        out[4 + v] = L[v];
        rec += (double)L[v] * counts[v];
        if (v < 2) ssim += sums[v * 4 + 3] / ((double)B * (H - 2) * (W - 2) * 3);
    }
    rec /= (double)n;
    const double smooth = sums[Vs * 4] / ((double)B * H * (W - 1)) + sums[Vs * 4 + 1] / ((double)B * (H - 1) * W);
    out[0] = (float)rec; out[1] = (float)ssim; out[2] = (float)smooth; out[3] = (float)(12 * rec + 6 * ssim + 0.18 * smooth);
}

extern "C" void h_unsup_loss_bwd(const float* ref, const float* srcs, const float* depth, const float* coef, const float* warped,
                                 const float* masks, const int* counts, const float* gout, float* gdepth, int B, int Vs, int H, int W) {
    const long long n = (long long)B * H * W;
    std::vector<float> ws((size_t)B * (H - 2) * (W - 2) * 9);
    float ks[2] = {(float)((double)gout[2] / ((double)B * H * (W - 1))), (float)((double)gout[2] / ((double)B * (H - 1) * W))};
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const long long o = (long long)b * H * W;
                gdepth[o + y * W + x] = ul::smooth_grad(depth + o, ref + o * 3, ks, y, x, H, W);
            }
    for (int v = 0; v < Vs; ++v) {
        const double share = 0.5 * (double)gout[0] * ((double)counts[v] / (double)n);
        float k[4] = {(float)(share / ((double)n * 3)), (float)(share / ((double)B * H * (W - 1) * 3)),
                      (float)(share / ((double)B * (H - 1) * W * 3)),
                      v < 2 ? (float)((double)gout[1] / ((double)B * (H - 2) * (W - 2) * 3)) : 0.0f};
        for (int b = 0; b < B; ++b) {
            const long long o = (long long)b * H * W;
            if (v < 2)
                for (int y = 1; y <= H - 2; ++y)
                    for (int x = 1; x <= W - 2; ++x)
                        ul::ssim_coefs(warped + (v * n + o) * 3, ref + o * 3, masks + v * n + o, y, x, W,
                                       ws.data() + (((long long)b * (H - 2) + (y - 1)) * (W - 2) + (x - 1)) * 9);
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    const ul::Taps t = ul::inv_warp_taps(coef + (v * B + b) * 12, x, y, depth[o + y * W + x], H, W);
                    float g = 0;
                    for (int c = 0; c < 3; ++c)
                        g += ul::photo_grad(warped + (v * n + o) * 3, ref + o * 3, masks + v * n + o,
                                            ws.data() + (long long)b * (H - 2) * (W - 2) * 9, k, y, x, c, H, W) *
                             ul::tap_ddepth(t, srcs + (v * n + o) * 3, c);
                    gdepth[o + y * W + x] += g;
                }
        }
    }
}
