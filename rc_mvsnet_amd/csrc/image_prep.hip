// Input-image preparation of the evaluation loader on the device: the decoded uint8 image
// (H, W, 3) -> float32 / 255 -> bilinear resize to the network size (cv2.resize conventions) -> ImageNet normalisation ->
// channels-first (3, h, w).  Reference: datasets/dtu_test.py:107-112 (read_img), :127-145 (scale_mvs_input),
// :78-81 (ToTensor + Normalize), which cost the reference's single loader worker ~30 ms per 1200 x 1600 image; here the
// host only decodes the JPEG and uploads 5.8 MB of bytes.  One thread per output pixel, all three channels, stores
// coalesced per channel plane; HBM-bound at (3 B read + 12 B written) per pixel.  gfx950 only.
#include "common.h"
#include "image_prep_math.h"

namespace rcmvs {

__global__ __launch_bounds__(256) void prepare_image_kernel(const unsigned char* __restrict__ src, float* __restrict__ out,
                                                             int H, int W, int h, int w, float m0, float m1, float m2,
                                                             float s0, float s1, float s2) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= h * w) return;
    const int y = p / w, x = p - y * w;
    out[p] = ip::prepared_pixel(src, H, W, h, w, y, x, 0, m0, s0);
    out[h * w + p] = ip::prepared_pixel(src, H, W, h, w, y, x, 1, m1, s1);
    out[2 * h * w + p] = ip::prepared_pixel(src, H, W, h, w, y, x, 2, m2, s2);
}

}  // namespace rcmvs

using namespace rcmvs;

extern "C" int rcmvs_prepare_image(const unsigned char* src, float* out, int H, int W, int h, int w, const float* mean_host,
                                   const float* std_host, void* stream) {
    RCMVS_REQUIRE(src && out && mean_host && std_host, "prepare_image: null pointer");
    RCMVS_REQUIRE(H > 0 && W > 0 && h > 0 && w > 0 && (long long)h * w < (1ll << 30) && (long long)H * W < (1ll << 29),
                  "prepare_image: bad dims %dx%d -> %dx%d", H, W, h, w);
    for (int c = 0; c < 3; ++c) RCMVS_REQUIRE(std_host[c] != 0.0f, "prepare_image: zero std");
    hipLaunchKernelGGL(prepare_image_kernel, dim3((h * w + 255) / 256), dim3(256), 0, as_stream(stream), src, out, H, W, h, w,
                       mean_host[0], mean_host[1], mean_host[2], std_host[0], std_host[1], std_host[2]);
    return launch_status("prepare_image");
}
