// CPU loop harness around rc_mvsnet_amd/csrc/image_prep_math.h for tests/test_dataset_cpu.py.  Test infrastructure only.
#include "../../rc_mvsnet_amd/csrc/image_prep_math.h"

extern "C" void h_prepare_image(const unsigned char* src, float* out, int H, int W, int h, int w, const float* mean, const float* stdv) {
    for (int c = 0; c < 3; ++c)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x)
                out[(c * h + y) * w + x] = rcmvs::ip::prepared_pixel(src, H, W, h, w, y, x, c, mean[c], stdv[c]);
}
