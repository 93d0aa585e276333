// Standalone developer harness for csrc/conv3d_x3.hip (GPU box, no torch): accuracy against an fp64 CPU convolution on small
// volumes, then HIP-event timings at the config-2 layer shapes.   hipcc --offload-arch=gfx950 -O3 tools/dev/x3_test.hip -o x3_test
// -DX3_ABLATION=1 adds the phase-ablation switches (X3_DBG=<mask>; the switches themselves slow the MFMA loop: compare within that build only)
#include "../../rc_mvsnet_amd/csrc/conv3d_x3.hip"
#include "../../rc_mvsnet_amd/csrc/conv3d_deep.hip"      // (conv3d_x3_launch hands the deep-level and Cout = 8 / conv2 shapes on to these two)
#include "../../rc_mvsnet_amd/csrc/conv3d_z8.hip"
#include <vector>
#include <random>
#include <cmath>
#include <cstdlib>
namespace rcmvs {
char* err_buf() { static char b[512]; return b; }
int fail(int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(err_buf(), 512, fmt, ap); va_end(ap); fprintf(stderr, "FAIL: %s\n", err_buf()); return code; }
}
using namespace rcmvs;
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

// kind 0/1: w (Co,Ci,27), stride 1/2.  kind 2: ConvTranspose3d stride 2, pad 1, output_padding 1, w (Ci,Co,27).  kind 3: planar (kd = 1 taps only)
static void ref_conv(int kind, const std::vector<float>& x, const std::vector<float>& w, const std::vector<float>& sc, const std::vector<float>& sh,
                     const std::vector<float>& res, std::vector<double>& y, std::vector<float>& y32, int D, int H, int W, int Do, int Ho, int Wo, int Ci, int Co, int relu) {
    for (int z = 0; z < Do; ++z) for (int yy = 0; yy < Ho; ++yy) for (int xx = 0; xx < Wo; ++xx) for (int co = 0; co < Co; ++co) {
        double a = 0; float a32 = 0.f;
        for (int kd = 0; kd < 3; ++kd) for (int kh = 0; kh < 3; ++kh) for (int kw = 0; kw < 3; ++kw) {
            int iz, iy, ix;
            if (kind == 3 && kd != 1) continue;
            if (kind == 2) {
                int nz = z + 1 - kd, ny = yy + 1 - kh, nx = xx + 1 - kw;
                if (nz < 0 || ny < 0 || nx < 0 || (nz & 1) || (ny & 1) || (nx & 1)) continue;
                iz = nz >> 1; iy = ny >> 1; ix = nx >> 1;
            } else if (kind == 3) { iz = z; iy = yy + kh - 1; ix = xx + kw - 1;
            } else { int s = kind == 1 ? 2 : 1; iz = s * z + kd - 1; iy = s * yy + kh - 1; ix = s * xx + kw - 1; }
            if (iz < 0 || iz >= D || iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            for (int ci = 0; ci < Ci; ++ci) {
                float xv = x[(((size_t)iz * H + iy) * W + ix) * Ci + ci];
                float wv = kind == 2 ? w[((size_t)ci * Co + co) * 27 + (kd * 3 + kh) * 3 + kw] : w[((size_t)co * Ci + ci) * 27 + (kd * 3 + kh) * 3 + kw];
                a += (double)xv * wv; a32 = fmaf(xv, wv, a32);
            }
        }
        size_t o = (((size_t)z * Ho + yy) * Wo + xx) * Co + co;
        double v = a * sc[co] + sh[co]; float v32 = a32 * sc[co] + sh[co];
        if (relu) { v = v > 0 ? v : 0; v32 = v32 > 0 ? v32 : 0; }
        v += res[o]; v32 += res[o];
        y[o] = v; y32[o] = v32;
    }
}

static int run_case(int kind, int Ci, int Co, int D, int H, int W, bool check, int reps, int blocks = 0) {
    const bool nores = getenv("X3_NORES") != nullptr;
    std::mt19937 rng(Ci * 131 + Co * 7 + D + H + W + kind);
    std::normal_distribution<float> nd(0.f, 1.f);
    int Do, Ho, Wo;
    if (kind == 2) { Do = 2 * D; Ho = 2 * H; Wo = 2 * W; } else { int s = kind == 1 ? 2 : 1; Do = (D - 1) / s + 1; Ho = (H - 1) / s + 1; Wo = (W - 1) / s + 1; }
    size_t nx = (size_t)D * H * W * Ci, ny = (size_t)Do * Ho * Wo * Co, nw = (size_t)Co * Ci * 27;
    std::vector<float> x(nx), w(nw), sc(Co), sh(Co), res(ny);
    for (auto& v : x) v = nd(rng) * (check ? expf(nd(rng)) : 1.f);
    for (auto& v : w) v = nd(rng) * 0.05f;
    for (auto& v : sc) v = 0.5f + fabsf(nd(rng));
    for (auto& v : sh) v = 0.1f * nd(rng);
    for (auto& v : res) v = nd(rng);
    float *dx, *dw, *dimg, *dsc, *dsh, *dres, *dy, *dmax;
    const bool pair = getenv("X3_NP") && atoi(getenv("X3_NP")) == 2 && conv3d_x3h_supported(Ci, Co, kind);     // the fp16-pair form
    long long imgf = pair ? conv3d_x3h_weight_floats(Ci, Co, kind) : conv3d_x3_weight_floats(Ci, Co, kind);
    CK(hipMalloc(&dmax, 4096 + 64 + 4096)); CK(hipMemset(dmax, 0, 4096 + 64 + 4096));                 // bound vector (64 slots, 16 floats apart) + the weight-scale scratch
    { float m = 0.f; for (auto v : x) m = fmaxf(m, fabsf(v)); std::vector<float> h(1024 + 16, 0.f); h[0] = m; CK(hipMemcpy(dmax, h.data(), 4096 + 64, hipMemcpyHostToDevice)); }
    const float* xmx = pair ? dmax : nullptr;
    float* ymx = getenv("X3_YMAX") ? dmax + 1024 + 16 : nullptr;       // X3_YMAX=1: the kernel also maintains the bound of its output
    CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&dimg, imgf * 4)); CK(hipMalloc(&dsc, Co * 4)); CK(hipMalloc(&dsh, Co * 4));
    CK(hipMalloc(&dres, ny * 4)); CK(hipMalloc(&dy, ny * 4));
    CK(hipMemcpy(dx, x.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, w.data(), nw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsc, sc.data(), Co * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsh, sh.data(), Co * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dres, res.data(), ny * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dy, 0xff, ny * 4));
    if (pair) { if (conv3d_x3_wscale(dw, (int)nw, dmax + 1024, 0) || conv3d_x3h_pack(dw, dimg, Co, Ci, kind, kind == 2 ? 1 : 0, dmax + 1024, 0)) return 1; }
    else if (conv3d_x3_pack(dw, dimg, Co, Ci, kind, kind == 2 ? 1 : 0, 0)) return 1;
    if (conv3d_x3_launch(dx, dimg, dsc, dsh, dres, dy, 1, D, H, W, Ci, Co, kind, 1, 0, blocks, 0, xmx, ymx)) return 1;
    CK(hipDeviceSynchronize());
    int bad = 0;
    if (check) {
        std::vector<double> yr(ny); std::vector<float> y32(ny), yg(ny);
        ref_conv(kind, x, w, sc, sh, res, yr, y32, D, H, W, Do, Ho, Wo, Ci, Co, 1);
        CK(hipMemcpy(yg.data(), dy, ny * 4, hipMemcpyDeviceToHost));
        double eg = 0, e32 = 0, mag = 0; size_t worst = 0;
        for (size_t i = 0; i < ny; ++i) {
            double d = fabs(yg[i] - yr[i]); if (!(d <= eg)) { eg = d; worst = i; }
            e32 = fmax(e32, fabs(y32[i] - yr[i])); mag = fmax(mag, fabs(yr[i]));
        }
        printf("check kind=%d Ci=%d Co=%d %dx%dx%d: max|y|=%.3f  max err x3 = %.3e (at %zu)  max err fp32 fma chain = %.3e\n", kind, Ci, Co, D, H, W, mag, eg, worst, e32);
        if (!(eg <= 4 * e32 + 1e-6 * mag)) { bad = 1; printf("   MISMATCH: gpu %.6f ref %.6f\n", yg[worst], yr[worst]); }
    }
#if X3_ABLATION
    if (reps > 0 && getenv("X3_TRACE")) {      // s_memtime stamps of block 0 (ticks 8..55): where a tick's time goes
        long long* dtr; CK(hipMalloc(&dtr, 64 * 8 * 8)); CK(hipMemset(dtr, 0, 64 * 8 * 8));
        x3_trace_buf = dtr;
        conv3d_x3_launch(dx, dimg, dsc, dsh, nores ? nullptr : dres, dy, 1, D, H, W, Ci, Co, kind, 1, 0, blocks, 0, xmx, ymx);
        CK(hipDeviceSynchronize());
        x3_trace_buf = nullptr;
        std::vector<long long> tr(64 * 8); CK(hipMemcpy(tr.data(), dtr, 64 * 8 * 8, hipMemcpyDeviceToHost)); hipFree(dtr);
        double acc[8] = {0}; int cnt = 0;
        for (int s = 8; s < 56; ++s) {
            if (!tr[(s + 1) * 8] || !tr[s * 8 + 7]) continue;
            const long long t0 = tr[s * 8];
            acc[0] += tr[s * 8 + 1] - t0; acc[1] += tr[s * 8 + 2] - t0; acc[2] += tr[(s + 1) * 8] - t0;            // consumer: MFMA phase end, before barrier, next tick start
            acc[4] += tr[s * 8 + 5] - tr[s * 8 + 4]; acc[5] += tr[s * 8 + 6] - tr[s * 8 + 4]; acc[6] += tr[s * 8 + 7] - tr[s * 8 + 4];
            acc[7] += tr[s * 8 + 4] - t0; ++cnt;
        }
        if (cnt) printf("trace kind=%d Ci=%d Co=%d (s_memtime units; mean over %d ticks): consumer MFMA-end %.0f, pre-barrier %.0f, tick %.0f | producer start-skew %.0f, stash-done %.0f, fetch-done %.0f, epilogue-done %.0f\n",
                        kind, Ci, Co, cnt, acc[0] / cnt, acc[1] / cnt, acc[2] / cnt, acc[7] / cnt, acc[4] / cnt, acc[5] / cnt, acc[6] / cnt);
    }
#endif
    if (reps > 0) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#if X3_ABLATION
        x3_ablation_mask = getenv("X3_DBG") ? atoi(getenv("X3_DBG")) : 0;
#endif
        for (int i = 0; i < 3; ++i) conv3d_x3_launch(dx, dimg, dsc, dsh, nores ? nullptr : dres, dy, 1, D, H, W, Ci, Co, kind, 1, 0, blocks, 0, xmx, ymx);
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) conv3d_x3_launch(dx, dimg, dsc, dsh, nores ? nullptr : dres, dy, 1, D, H, W, Ci, Co, kind, 1, 0, blocks, 0, xmx, ymx);
#if X3_ABLATION
        x3_ablation_mask = 0;
#endif
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double us = ms * 1e3 / reps, fl = 2.0 * (kind == 3 ? 9 : 27) * Ci * Co * (kind == 2 ? (double)D * H * W : (double)Do * Ho * Wo);
        printf("time  kind=%d Ci=%d Co=%d %dx%dx%d %s: %8.1f us  %6.1f TF (fp32-equivalent)\n", kind, Ci, Co, D, H, W, pair ? "fp16 pair  " : "bf16 triple", us, fl / us / 1e6);
    }
    hipFree(dmax); hipFree(dx); hipFree(dw); hipFree(dimg); hipFree(dsc); hipFree(dsh); hipFree(dres); hipFree(dy);
    return bad;
}

int main(int argc, char** argv) {
    int bad = 0;
    const int cases[13][3] = {{3, 64, 32}, {0, 32, 32}, {2, 32, 16}, {0, 8, 8}, {0, 16, 8}, {0, 32, 8}, {0, 16, 16}, {1, 8, 16}, {1, 16, 32}, {2, 16, 8}, {3, 8, 8}, {3, 16, 16}, {3, 32, 32}};
    for (auto& p : cases) {
        if (getenv("X3_NOCHECK")) break;
        bad |= run_case(p[0], p[1], p[2], 8, 8, 32, true, 0);
        bad |= run_case(p[0], p[1], p[2], 11, 13, 45, true, 0);     // ragged: partial tiles in x and y, z not a multiple of the chunk
        bad |= run_case(p[0], p[1], p[2], 3, 20, 70, true, 0);
        bad |= run_case(p[0], p[1], p[2], 10, 21, 70, true, 0, 3);     // three persistent blocks: several items per block, hand-over across tiles and z chunks
    }
    if (argc > 1 && atoi(argv[1]) == 0) return bad;
    if (argc > 1 && atoi(argv[1]) == 4) {        // FeatureNet layers: 3 views as planes
        run_case(3, 8, 8, 3, 512, 640, false, 20); run_case(3, 16, 16, 3, 256, 320, false, 20); run_case(3, 32, 32, 3, 128, 160, false, 20); run_case(3, 64, 32, 3, 128, 160, false, 20);
        run_case(0, 32, 32, 12, 32, 40, false, 20); run_case(0, 32, 32, 8, 64, 80, false, 20); run_case(0, 32, 32, 2, 128, 160, false, 20);
        run_case(2, 32, 16, 6, 16, 20, false, 20); run_case(2, 32, 16, 4, 32, 40, false, 20); run_case(2, 32, 16, 1, 64, 80, false, 20);
        return 0;
    }
    if (argc > 1 && atoi(argv[1]) == 3) {
        const int blk = getenv("X3_BLOCKS") ? atoi(getenv("X3_BLOCKS")) : 0;
        run_case(0, 32, 8, 48, 128, 160, false, 20, blk); run_case(0, 16, 8, 32, 256, 320, false, 20, blk); run_case(0, 8, 8, 8, 512, 640, false, 20, blk);
        run_case(0, 16, 16, 16, 128, 160, false, 20, blk); run_case(1, 8, 16, 32, 256, 320, false, 20, blk); run_case(2, 16, 8, 16, 128, 160, false, 20, blk);
        return 0;
    }
    run_case(0, 32, 8, 48, 128, 160, false, 20);
    run_case(0, 16, 8, 32, 256, 320, false, 20);
    run_case(0, 8, 8, 8, 512, 640, false, 20);
    run_case(0, 16, 16, 16, 128, 160, false, 20);
    run_case(1, 8, 16, 48, 128, 160, false, 20);
    run_case(1, 8, 16, 32, 256, 320, false, 20);
    run_case(1, 8, 16, 8, 512, 640, false, 20);
    run_case(1, 16, 32, 16, 128, 160, false, 20);
    run_case(2, 16, 8, 24, 64, 80, false, 20);
    run_case(2, 16, 8, 16, 128, 160, false, 20);
    run_case(2, 16, 8, 4, 256, 320, false, 20);
    printf(bad ? "X3 TEST FAILED\n" : "X3 TEST OK\n");
    return bad;
}
