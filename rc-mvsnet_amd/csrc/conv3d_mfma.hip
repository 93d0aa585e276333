// stride-1 3x3x3 convolution on v_mfma_f32_16x16x4_f32 (placeholder: dispatch disabled until the
// kernel lands; rcmvs_conv3d_fwd falls through to the direct kernel).
#include "common.h"

namespace rcmvs {

bool conv3d_mfma_supported(int, int, int, int, int) { return false; }

int conv3d_mfma_launch(const float*, const float*, const float*, const float*, const float*, float*, int, int, int, int,
                       int, int, int, hipStream_t) {
    return fail(-2, "conv3d_mfma: not built");
}

}  // namespace rcmvs
