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

// The train variant's small images (models/casmvsnet.py:60-62,148-150: F.interpolate(imgs, (h, w), mode="bilinear", align_corners=False), then the
// channels-last view the warp kernels read): (N, 3, H, W) planar -> (N, h, w, 3), ATen's upsample_bilinear2d arithmetic (source index
// max(scale (dst + 0.5) - 0.5, 0) with scale = H / h in fp32, the two row blends before the column blend, no contraction) so that the result
// equals torch's on the same GPU.  One thread per output pixel.
__global__ __launch_bounds__(256) void resize_rgb_cl_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int h, int w) {
#pragma clang fp contract(off)
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= h * w) return;
    const int n = blockIdx.y;
    const int oy = p / w, ox = p - oy * w;
    const float rh = (float)H / (float)h, rw = (float)W / (float)w;
    const float h1r = fmaxf(rh * ((float)oy + 0.5f) - 0.5f, 0.0f), w1r = fmaxf(rw * ((float)ox + 0.5f) - 0.5f, 0.0f);
    const int h1 = (int)h1r, w1 = (int)w1r;
    const int h1p = h1 < H - 1 ? 1 : 0, w1p = w1 < W - 1 ? 1 : 0;
    const float h1l = h1r - (float)h1, w1l = w1r - (float)w1;
    const float h0l = 1.0f - h1l, w0l = 1.0f - w1l;
    const float* xp = x + (long long)n * 3 * H * W;
    float* yp = y + ((long long)n * h * w + p) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* pc = xp + (long long)c * H * W + (long long)h1 * W + w1;
        yp[c] = h0l * (w0l * pc[0] + w1l * pc[w1p]) + h1l * (w0l * pc[(long long)h1p * W] + w1l * pc[(long long)h1p * W + w1p]);
    }
}

}  // namespace rcmvs

using namespace rcmvs;

extern "C" int rcmvs_resize_rgb_cl(const float* x, float* y, int N, int H, int W, int h, int w, void* stream) {
    RCMVS_REQUIRE(x && y, "resize_rgb_cl: null pointer");
    RCMVS_REQUIRE(N > 0 && H > 0 && W > 0 && h > 0 && w > 0 && (long long)h * w < (1ll << 30) && N <= 65535, "resize_rgb_cl: bad dims N=%d %dx%d -> %dx%d", N, H, W, h, w);
    hipLaunchKernelGGL(resize_rgb_cl_kernel, dim3((h * w + 255) / 256, N), dim3(256), 0, as_stream(stream), x, y, H, W, h, w);
    return launch_status("resize_rgb_cl");
}

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
