// K1 lab: standalone timing / bit-comparison harness for warp+variance kernel experiments (no Python, no torch).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/dev/_bin/k1_lab tools/dev/k1_lab/lab.hip
//   tools/dev/_bin/k1_lab [variant ...]        (data: tools/dev/k1_lab/data, written by gen_data.py)
// Variants 0-7 are rcmvs_debug_warp_variance_fwd's; >= 20 are the experiments of lab_*.h.  Every variant's output is
// compared bit for bit with variant 0's.
#include "../../../rc_mvsnet_amd/csrc/warp_variance.hip"
#include <vector>
#include <string>
#include <cstring>
#include <cstdlib>

namespace rcmvs {
char* err_buf() { static thread_local char b[512]; return b; }
int fail(int code, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(err_buf(), 512, fmt, ap); va_end(ap);
    return code;
}
}  // namespace rcmvs

#include "lab_kernels.h"
#include "lab_win.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

static std::vector<float> read_f32(const std::string& path, size_t n) {
    std::vector<float> v(n);
    FILE* f = fopen(path.c_str(), "rb");
    if (!f || fread(v.data(), 4, n, f) != n) { fprintf(stderr, "cannot read %s\n", path.c_str()); exit(1); }
    fclose(f);
    return v;
}

int main(int argc, char** argv) {
    std::vector<int> variants;
    for (int i = 1; i < argc; ++i) variants.push_back(atoi(argv[i]));
    if (variants.empty()) variants = {0};
    const char* dd = getenv("K1_LAB_DATA");
    const std::string dir = dd ? dd : "tools/dev/k1_lab/data";
    const int V = 3, H = 512, W = 640;
    const int only = getenv("K1_STAGE") ? atoi(getenv("K1_STAGE")) : 0;
    const int R = getenv("K1_REPS") ? atoi(getenv("K1_REPS")) : 20;
    struct St { int C, D, sc; } st[3] = {{32, 48, 4}, {16, 32, 2}, {8, 8, 1}};
    std::vector<double> tot(variants.size(), 0.0);
    for (int s = 0; s < 3; ++s) {
        if (only && only != s + 1) continue;
        const int C = st[s].C, D = st[s].D, h = H / st[s].sc, w = W / st[s].sc;
        const size_t nf = (size_t)V * h * w * C, nv = (size_t)D * h * w * C;
        std::vector<float> feats(nf);
        unsigned long long z = 0x9E3779B97F4A7C15ull * (s + 1);
        for (auto& f : feats) { z = z * 6364136223846793005ull + 1442695040888963407ull; f = (float)((int)(z >> 40) - (1 << 23)) / (float)(1 << 22); }
        auto planes = read_f32(dir + "/planes_s" + std::to_string(s + 1) + ".bin", (size_t)h * w * 2);
        auto rt = read_f32(dir + "/rt_s" + std::to_string(s + 1) + ".bin", (size_t)(V - 1) * 12);
        std::vector<float> rot((V - 1) * 9), trans((V - 1) * 3);
        for (int v = 0; v < V - 1; ++v) { memcpy(&rot[v * 9], &rt[v * 12], 36); memcpy(&trans[v * 3], &rt[v * 12 + 9], 12); }
        float *d_f, *d_r, *d_t, *d_p, *d_o;
        unsigned* d_st;
        const size_t nst = 8 * 8 * 16384;             // room for the per-block timelines of variants 70+
        CK(hipMalloc(&d_st, nst));
        CK(hipMalloc(&d_f, nf * 4)); CK(hipMalloc(&d_r, rot.size() * 4)); CK(hipMalloc(&d_t, trans.size() * 4));
        CK(hipMalloc(&d_p, planes.size() * 4)); CK(hipMalloc(&d_o, nv * 4));
        CK(hipMemcpy(d_f, feats.data(), nf * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_r, rot.data(), rot.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_t, trans.data(), trans.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_p, planes.data(), planes.size() * 4, hipMemcpyHostToDevice));
        std::vector<float> ref(nv), out(nv);
        CK(hipMemset(d_o, 0xff, nv * 4));
        if (rcmvs_debug_warp_variance_fwd(d_f, d_r, d_t, d_p, d_o, 1, V, C, D, h, w, 0, nullptr)) { fprintf(stderr, "ref: %s\n", rcmvs::err_buf()); return 1; }
        CK(hipMemcpy(ref.data(), d_o, nv * 4, hipMemcpyDeviceToHost));
        const double nbytes = 4.0 * ((double)(V - 1) * C * h * w + (double)C * h * w + (double)D * h * w + (double)C * D * h * w);
        for (size_t vi = 0; vi < variants.size(); ++vi) {
            const int v = variants[vi];
            unsigned* stp = d_st;           // statistics in the verification run only: 10^4 same-address atomics cost ~20 ns each
            auto run = [&]() -> int {
                if (v < 20) return rcmvs_debug_warp_variance_fwd(d_f, d_r, d_t, d_p, d_o, 1, V, C, D, h, w, v, nullptr);
                if (v >= 40) return lab_win_launch(v, d_f, d_r, d_t, d_p, d_o, 1, V, C, D, h, w, stp, nullptr);
                return lab_launch(v, d_f, d_r, d_t, d_p, d_o, 1, V, C, D, h, w, nullptr);
            };
            CK(hipMemset(d_o, 0xff, nv * 4));
            CK(hipMemset(d_st, 0, nst));
            int rc = run();
            if (rc) { printf("C=%2d variant %3d: launch failed: %s\n", C, v, rcmvs::err_buf()); continue; }
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(out.data(), d_o, nv * 4, hipMemcpyDeviceToHost));
            size_t bad = 0, first = 0;
            double maxd = 0, maxr = 0;
            unsigned hst[2] = {0, 0};
            CK(hipMemcpy(hst, d_st, 8, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < nv; ++i) {
                if (memcmp(&out[i], &ref[i], 4)) { if (!bad) first = i; ++bad; }
                const double dd = fabs((double)out[i] - (double)ref[i]);
                if (!(dd <= maxd)) maxd = dd;               // NaN-propagating
                if (fabs((double)ref[i]) > maxr) maxr = fabs((double)ref[i]);
            }
            stp = nullptr;
            for (int i = 0; i < 3; ++i) run();
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, nullptr));
            for (int i = 0; i < R; ++i) run();
            CK(hipEventRecord(e1, nullptr));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / R;
            tot[vi] += us;
            printf("S%d C=%2d D=%2d %dx%d variant %3d: %8.1f us %8.1f GB/s  %s", s + 1, C, D, h, w, v, us, nbytes / us / 1e3, bad ? "MISMATCH" : "bit-identical");
            if (bad) printf(" (%zu of %zu words, first at %zu: %g vs %g; max |d| %.3g = %.2e of max |ref|)", bad, nv, first, out[first], ref[first], maxd, maxd / maxr);
            if (hst[0]) printf("  [window path: %u of %u blocks]", hst[1], hst[0]);
            printf("\n");
            fflush(stdout);
        }
        (void)hipFree(d_f); (void)hipFree(d_r); (void)hipFree(d_t); (void)hipFree(d_p); (void)hipFree(d_o);
    }
    for (size_t vi = 0; vi < variants.size(); ++vi)
        printf("variant %3d: total %.1f us/scene -> %.3f of 8 TB/s\n", variants[vi], tot[vi], 457441280.0 / tot[vi] / 1e3 / 8000.0);
    return 0;
}
