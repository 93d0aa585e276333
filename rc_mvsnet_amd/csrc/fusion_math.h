// Per-pixel arithmetic of the depth-map fusion filter (eval_rcmvsnet_dtu.py:281-338 and the
// per-view body of filter_depth :369-425 of the reference).  Plain C++ shared by fusion.hip and by the CPU loop harness of
// tests/test_fusion_cpu.py (test infrastructure).
//
// The reference does this in numpy: float32 camera matrices (inverted / multiplied in float32 on the host, kept that way
// here), promoted to float64 for the per-pixel chain, with float32 casts at fixed points and cv2.remap(INTER_LINEAR) for
// the source-depth lookup.  The chain below keeps float64 and the same cast points, so masks agree except at thresholds'
// knife edges.  cv2.remap (opencv-python 4.5.5.62, requirements.txt:31) is not part of the reference tree and is absent
// from this image; remap_linear restates its published algorithm (imgwarp.cpp: coordinates rounded to 1/32 pixel,
// float weight table, BORDER_CONSTANT = 0).
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define RCMVS_HD __host__ __device__ inline
#else
#define RCMVS_HD inline
#endif

namespace rcmvs {
namespace fu {

// matrices, row-major doubles (values are the reference's float32 results promoted)
constexpr int REF_MATS = 9 + 9 + 12;          // K_ref^-1, K_ref, (E_ref^-1)[:3,:4]
constexpr int SRC_MATS = 12 + 9 + 9 + 12;     // (E_src E_ref^-1)[:3,:4], K_src, K_src^-1, (E_ref E_src^-1)[:3,:4]

#pragma clang fp contract(off)
RCMVS_HD void mul3(const double* m, double a, double b, double c, double* o) {
    o[0] = m[0] * a + m[1] * b + m[2] * c;
    o[1] = m[3] * a + m[4] * b + m[5] * c;
    o[2] = m[6] * a + m[7] * b + m[8] * c;
}
RCMVS_HD void mul34(const double* m, const double* v, double* o) {     // (3x4) * (v, 1)
    o[0] = m[0] * v[0] + m[1] * v[1] + m[2] * v[2] + m[3];
    o[1] = m[4] * v[0] + m[5] * v[1] + m[6] * v[2] + m[7];
    o[2] = m[8] * v[0] + m[9] * v[1] + m[10] * v[2] + m[11];
}

// cv2.remap(img, x, y, INTER_LINEAR) with the default BORDER_CONSTANT (0) for one float32 pixel
RCMVS_HD float remap_linear(const float* img, int H, int W, float x, float y) {
    const float fx32 = x * 32.0f, fy32 = y * 32.0f;
    if (!(fabsf(fx32) < 1.0e9f) || !(fabsf(fy32) < 1.0e9f)) return 0.0f;      // NaN / far away: outside the image
    const int sx = (int)rintf(fx32), sy = (int)rintf(fy32);                  // cvRound: nearest, ties to even
    const int ix = sx >> 5, iy = sy >> 5;
    const float ax = (float)(sx & 31) * (1.0f / 32.0f), ay = (float)(sy & 31) * (1.0f / 32.0f);
    if (ix >= W || ix + 1 < 0 || iy >= H || iy + 1 < 0) return 0.0f;
    const float w0 = (1.0f - ay) * (1.0f - ax), w1 = (1.0f - ay) * ax, w2 = ay * (1.0f - ax), w3 = ay * ax;
    const bool x0 = ix >= 0, x1 = ix + 1 < W, y0 = iy >= 0, y1 = iy + 1 < H;
    const float s0 = (x0 && y0) ? img[iy * W + ix] : 0.0f;
    const float s1 = (x1 && y0) ? img[iy * W + ix + 1] : 0.0f;
    const float s2 = (x0 && y1) ? img[(iy + 1) * W + ix] : 0.0f;
    const float s3 = (x1 && y1) ? img[(iy + 1) * W + ix + 1] : 0.0f;
    return s0 * w0 + s1 * w1 + s2 * w2 + s3 * w3;
}

struct Reproj { float depth, x_src, y_src; bool ok; };

// reproject_with_depth + check_geometric_consistency for reference pixel (x, y) against one source view
// (eval_rcmvsnet_dtu.py:281-338).  rm = REF_MATS doubles, sm = SRC_MATS doubles.
RCMVS_HD Reproj reproject(const double* rm, const double* sm, const float* depth_src, int H, int W, int x, int y,
                          float d_ref, double dist_thresh, float depth_thresh) {
    const double d = (double)d_ref;
    double p[3], q[3], k[3];
    mul3(rm, (double)x * d, (double)y * d, d, p);                      // reference camera space
    mul34(sm, p, q);                                                   // source camera space
    mul3(sm + 12, q[0], q[1], q[2], k);
    const double xs = k[0] / k[2], ys = k[1] / k[2];
    Reproj r;
    r.x_src = (float)xs; r.y_src = (float)ys;
    const double s = (double)remap_linear(depth_src, H, W, r.x_src, r.y_src);
    mul3(sm + 21, xs * s, ys * s, s, p);                               // back into source camera space at the sampled depth
    mul34(sm + 30, p, q);                                              // reference camera space
    const float d_rep = (float)q[2];
    mul3(rm + 9, q[0], q[1], q[2], k);
    const float xr = (float)(k[0] / k[2]), yr = (float)(k[1] / k[2]);
    const double dx = (double)xr - (double)x, dy = (double)yr - (double)y;
    const double dist = sqrt(dx * dx + dy * dy);
    const float rel = fabsf(d_rep - d_ref) / d_ref;
    r.ok = (dist < dist_thresh) && (rel < depth_thresh);
    r.depth = r.ok ? d_rep : 0.0f;
    return r;
}

// world point of reference pixel (x, y) at depth d (eval_rcmvsnet_dtu.py:418-421), before the cast to float32
RCMVS_HD void world_point(const double* rm, int x, int y, double d, double* out) {
    double p[3];
    mul3(rm, (double)x * d, (double)y * d, d, p);
    mul34(rm + 18, p, out);
}

}  // namespace fu
}  // namespace rcmvs
