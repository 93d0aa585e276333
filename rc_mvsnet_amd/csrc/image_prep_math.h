// Per-pixel arithmetic of the input-image preparation of the evaluation loader (SURVEY.md section 8f rank 4;
// datasets/dtu_test.py:107-112,127-145,171-191 of the reference): uint8 -> float32 / 255, bilinear resize with cv2.resize's
// INTER_LINEAR conventions, ImageNet normalisation, channels-first.  Plain C++ shared by image_prep.hip and the CPU loop
// harness of tests/test_dataset_cpu.py (test infrastructure).
//
// cv2.resize (opencv-python 4.5.5.62, requirements.txt:31) is absent from the reference tree and from this image; the
// coordinate rule below restates its published algorithm (modules/imgproc/src/resize.cpp, float32 path): source position
// (d + 0.5) * scale - 0.5 in float, floor + fraction, fraction zeroed where the horizontal tap pair leaves the row, rows
// clamped for the vertical pair, horizontal pass first.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define RCMVS_HD __host__ __device__ inline
#else
#define RCMVS_HD inline
#endif

namespace rcmvs {
namespace ip {

struct Tap { int i0, i1; float w0, w1; };

#pragma clang fp contract(off)
// horizontal rule: out-of-row taps collapse onto the border pixel with weight 1
RCMVS_HD Tap tap_x(int d, double scale, int n) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { s = 0; f = 0.0f; }
    Tap t;
    if (s >= n - 1) { t.i0 = n - 1; t.i1 = n - 1; t.w0 = 1.0f; t.w1 = 0.0f; return t; }
    t.i0 = s; t.i1 = s + 1; t.w0 = 1.0f - f; t.w1 = f;
    return t;
}
// vertical rule: the fraction is kept and the two rows are clamped
RCMVS_HD Tap tap_y(int d, double scale, int n) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    const int s = (int)floorf(f);
    f -= (float)s;
    Tap t;
    t.i0 = s < 0 ? 0 : (s > n - 1 ? n - 1 : s);
    t.i1 = s + 1 < 0 ? 0 : (s + 1 > n - 1 ? n - 1 : s + 1);
    t.w0 = 1.0f - f; t.w1 = f;
    return t;
}

// one output value: src (H, W, 3) uint8, channel c, output pixel (y, x) of an (h, w) image
RCMVS_HD float prepared_pixel(const unsigned char* src, int H, int W, int h, int w, int y, int x, int c, float mean, float stdv) {
    const double sx = 1.0 / ((double)w / (double)W), sy = 1.0 / ((double)h / (double)H);
    const Tap tx = tap_x(x, sx, W), ty = tap_y(y, sy, H);
    const float k = 255.0f;
    const float a00 = (float)src[(ty.i0 * W + tx.i0) * 3 + c] / k, a01 = (float)src[(ty.i0 * W + tx.i1) * 3 + c] / k;
    const float a10 = (float)src[(ty.i1 * W + tx.i0) * 3 + c] / k, a11 = (float)src[(ty.i1 * W + tx.i1) * 3 + c] / k;
    float v;
    if (h == H && w == W) {
        v = a00;                                                   // cv2.resize copies when the size is unchanged
    } else {
        const float r0 = tx.i0 == tx.i1 ? a00 * 1.0f : a00 * tx.w0 + a01 * tx.w1;
        const float r1 = tx.i0 == tx.i1 ? a10 * 1.0f : a10 * tx.w0 + a11 * tx.w1;
        v = r0 * ty.w0 + r1 * ty.w1;
    }
    return (v - mean) / stdv;                                      // transforms.Normalize
}

}  // namespace ip
}  // namespace rcmvs
