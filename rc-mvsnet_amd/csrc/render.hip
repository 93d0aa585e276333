// Rendering-consistency branch kernels (placeholders until implemented).
#include "common.h"
using namespace rcmvs;
extern "C" {
int rcmvs_resize_planes_fwd(const float*, float*, int, int, int, int, int, int, void*) { return fail(-2, "resize_planes_fwd: not implemented"); }
int rcmvs_gu_sample_fwd(const float*, const float*, const int*, const float*, const float*, const float*, float*, float*, float*, float*, float*, float*, int, int, int, int, void*) { return fail(-2, "gu_sample_fwd: not implemented"); }
int rcmvs_point_feats_fwd(const float*, const float*, const float*, const float*, const float*, float*, int, int, int, int, int, int, int, void*) { return fail(-2, "point_feats_fwd: not implemented"); }
int rcmvs_nerf_mlp_fwd(const float*, const float*, const float*, const float*, float*, int, int, void*) { return fail(-2, "nerf_mlp_fwd: not implemented"); }
long long rcmvs_nerf_weight_floats(void) { return 0; }
int rcmvs_composite_fwd(const float*, const float*, float*, float*, float*, float*, int, int, void*) { return fail(-2, "composite_fwd: not implemented"); }
}
