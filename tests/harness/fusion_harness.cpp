// CPU loop harness around rc_mvsnet_amd/csrc/fusion_math.h (the per-pixel arithmetic of fusion.hip) for
// tests/test_fusion_cpu.py.  Test infrastructure only -- nothing in the package loads this.
#include "../../rc_mvsnet_amd/csrc/fusion_math.h"

using namespace rcmvs;

extern "C" void h_fuse_view(const float* depth_all, int ref_idx, const int* src_idx, const float* conf, const float* img,
                            const double* mats, float prob_thresh, int num_consistent, double dist_thresh, float depth_thresh,
                            unsigned char* masks, float* depth_avg, float* xyz, unsigned char* rgb, float* dbg_depth,
                            unsigned char* dbg_geo, float* dbg_xy, int N, int H, int W) {
    const int plane = H * W;
    for (int p = 0; p < plane; ++p) {
        const int y = p / W, x = p - y * W;
        const float d_ref = depth_all[(long long)ref_idx * plane + p];
        int geo_sum = 0;
        float acc = 0.0f;
        for (int n = 0; n < N; ++n) {
            const fu::Reproj r = fu::reproject(mats, mats + fu::REF_MATS + n * fu::SRC_MATS, depth_all + (long long)src_idx[n] * plane,
                                               H, W, x, y, d_ref, dist_thresh, depth_thresh);
            geo_sum += r.ok ? 1 : 0;
            acc += r.depth;
            if (dbg_depth) dbg_depth[(long long)n * plane + p] = r.depth;
            if (dbg_geo) dbg_geo[(long long)n * plane + p] = r.ok ? 1 : 0;
            if (dbg_xy) { dbg_xy[((long long)n * plane + p) * 2] = r.x_src; dbg_xy[((long long)n * plane + p) * 2 + 1] = r.y_src; }
        }
        const double avg = (double)(acc + d_ref) / (double)(geo_sum + 1);
        const bool photo = conf[p] > prob_thresh, geo = geo_sum >= num_consistent;
        masks[p] = photo; masks[plane + p] = geo; masks[2 * plane + p] = photo && geo;
        depth_avg[p] = (float)avg;
        double w[3];
        fu::world_point(mats, x, y, avg, w);
        for (int c = 0; c < 3; ++c) xyz[p * 3 + c] = (float)w[c];
        if (img) for (int c = 0; c < 3; ++c) rgb[p * 3 + c] = (unsigned char)(int)(img[p * 3 + c] * 255.0f);
    }
}
