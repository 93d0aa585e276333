// K4: prob conv (8 -> 1, 3x3x3) + softmax over planes + soft-argmin depth + photometric
// confidence, fused: the logit volume, the probability volume and the two regression sums never
// leave the chip unless the caller asks for `prob`.
// Replaces CostRegNet.prob (models/modules.py:489,500), F.softmax, depth_regression (x2),
// F.pad + avg_pool3d and torch.gather in DepthNet_eval.forward (models/casmvsnet.py:293-309).
//
// Round 4: with D = 8 (the last stage) ONE launch: the marching prob conv keeps a pixel's eight logits in registers and finishes
// them itself (conv3d_lds.hip, prob_conv_march_plain_kernel<8>; bit-identical to the two launches; `prob` optional).  Otherwise:
// Two launches: (1) the prob conv runs on the LDS-staged halo kernel of conv3d_lds.hip (Cout = 1: the
// logits land in the caller's (B,D,h,w) probability buffer); (2) one thread per pixel turns its logit
// column into probabilities IN PLACE (max, exp, sum, divide -- coalesced plane-major accesses, the
// column stays in L2) and accumulates depth = sum p*d, index = sum p*k and the 4-tap confidence window.
// The logit volume is D*h*w*4 B = 4-10 MB per stage, so the round trip is noise next to the 31-84 MB
// input volume, which is read exactly once.
// Round 3 (profiles/r3_depth_head_ab.txt): two single-launch forms were written, verified and timed -- (a) logits of a pixel tile kept
// in LDS for all D planes with the softmax in the same block, one pixel per thread: 134.6 us per scene against 129.9 for the two
// launches here; (b) strips of four pixels per thread on wave-private LDS slabs with the tap-column loop rolled so that a column's 72
// weights stay in SGPRs: 234.4 us (two waves per SIMD, three exposed scalar-load latencies per plane).  The prob conv is bound by
// the delivery of its 216 wave-uniform weights (the compiler hoists and spills them: two v_readlane per v_pk_fma), not by the logit
// round trip; both forms were removed again.
#include "common.h"

namespace rcmvs {

int conv3d_lds_launch(const float* x, const float* wp, const float* scale, const float* shift, const float* res, float* y,
                      int B, int D, int H, int W, int Ci, int Co, int relu, hipStream_t st, int lds_cfg);   // conv3d_lds.hip
bool prob_head_fused_supported(int D, int H, int W);                                                       // conv3d_lds.hip
bool prob_pair_supported(int D, int H, int W);                                                             // prob_pair.hip
int prob_pair_launch(const float* x, const float* wimg, const float* xmax, float* y, const float* planes, float* depth, float* conf,
                     int B, int D, int H, int W, bool fuse, int zc_force, hipStream_t st);
long long prob_pair_blob_offset();                                                                         // conv3d.hip
int prob_head_fused_launch(const float* x, const float* wp, const float* planes, float* depth, float* conf, float* prob,
                           int B, int D, int H, int W, hipStream_t st);

// LP adjacent lanes share one pixel and own the planes k = j, j + LP, ... (at most 16 each, held in registers): the logit
// column is read ONCE with all of a lane's loads in flight, max / sum / soft-argmin / window sums are combined across the
// LP lanes with shuffles, the probabilities are written once.  LP = 1, 2, 4 for D <= 16, 32, 64 -- the 128x160 stage has
// only 20 k pixels, so spreading a pixel over four lanes is also what fills the machine there.
// KEEP = false: the caller does not want the probability volume (inference: only depth and confidence leave the head) -- the logits are
// read and nothing is written back.
template <int LP, bool KEEP = true>
__global__ __launch_bounds__(256) void softmax_regress_kernel(float* __restrict__ prob, const float* __restrict__ planes,
                                                               float* __restrict__ depth, float* __restrict__ conf, int D, long long hw) {
    constexpr int MAXK = 16;
    const int b = blockIdx.y;
    const int j = threadIdx.x % LP;
    const long long p = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / LP;
    const bool live = p < hw;
    float* col = prob + (long long)b * D * hw + (live ? p : 0);  // element k at col[k*hw]
    float v[MAXK];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAXK; ++i) {
        const int k = j + i * LP;
        v[i] = (k < D) ? col[(long long)k * hw] : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
#pragma unroll
    for (int m = 1; m < LP; m <<= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXK; ++i) {
        v[i] = (j + i * LP < D) ? expf(v[i] - mx) : 0.0f;
        sum += v[i];
    }
#pragma unroll
    for (int m = 1; m < LP; m <<= 1) sum += __shfl_xor(sum, m);
    const float2 pl = reinterpret_cast<const float2*>(planes)[(long long)b * hw + (live ? p : 0)];
    float dsum = 0.0f, isum = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXK; ++i) {
        const int k = j + i * LP;
        v[i] = v[i] / sum;
        if (k < D) {
            if (KEEP && live) col[(long long)k * hw] = v[i];
            dsum = fmaf(v[i], fmaf((float)k, pl.y, pl.x), dsum);       // (explicit: the one-launch form of conv3d_lds.hip must round alike)
            isum = fmaf(v[i], (float)k, isum);
        }
    }
#pragma unroll
    for (int m = 1; m < LP; m <<= 1) { dsum += __shfl_xor(dsum, m); isum += __shfl_xor(isum, m); }
    int idx = (int)isum;                       // .long(): truncation
    idx = idx < 0 ? 0 : (idx > D - 1 ? D - 1 : idx);
    float c = 0.0f;                            // p[i-1] + p[i] + p[i+1] + p[i+2], zero padded
#pragma unroll
    for (int i = 0; i < MAXK; ++i) {
        const int k = j + i * LP;
        if (k < D && k >= idx - 1 && k <= idx + 2) c += v[i];
    }
#pragma unroll
    for (int m = 1; m < LP; m <<= 1) c += __shfl_xor(c, m);
    if (live && j == 0) {
        depth[(long long)b * hw + p] = dsum;
        conf[(long long)b * hw + p] = c;
    }
}

}  // namespace rcmvs

using namespace rcmvs;

/* impl (tests, A/B): bit 0 = two launches even where the one-launch form exists (D = 8); bit 1 = the generic marching prob conv;
 * bit 2 = the VALU prob conv although a bound was given; bit 3 = `prob` is scratch only (the probabilities are not written back: production
 * callers that want depth and confidence only); bits 8-15 = z chunk of the prob conv (0 = chosen per launch).
 * xmax: bound of max|x| (ABSMAX slot format) -> the matrix-core prob conv of prob_pair.hip (fp16 pairs); NULL -> the exact VALU form */
static int depth_head_fwd(const float* x, const float* xmax, const float* w_prob, const float* planes, float* depth, float* conf, float* prob,
                          int B, int D, int h, int w, int impl, void* stream) {
    RCMVS_REQUIRE(x && w_prob && planes && depth && conf, "depth_head_fwd: null pointer");
    RCMVS_REQUIRE(B > 0 && D > 0 && h > 0 && w > 0, "depth_head_fwd: bad sizes");
    RCMVS_REQUIRE(D <= 64, "depth_head_fwd: at most 64 depth hypotheses per stage (got %d)", D);
    hipStream_t st = as_stream(stream);
    const int zc_force = (impl >> 8) & 0xff;
    const bool pair = xmax && !(impl & 4) && prob_pair_supported(D, h, w);
    int rc;
    if (pair && D == 8 && !(impl & 1))
        return prob_pair_launch(x, w_prob + prob_pair_blob_offset(), xmax, prob, planes, depth, conf, B, D, h, w, true, 0, st);
    if (!pair && !(impl & 1) && prob_head_fused_supported(D, h, w)) return prob_head_fused_launch(x, w_prob, planes, depth, conf, prob, B, D, h, w, st);
    RCMVS_REQUIRE(prob, "depth_head_fwd: prob is required for D = %d (it doubles as the logit scratch; only the D = 8 form runs without)", D);
    if (pair) rc = prob_pair_launch(x, w_prob + prob_pair_blob_offset(), xmax, prob, nullptr, nullptr, nullptr, B, D, h, w, false, zc_force, st);
    else rc = conv3d_lds_launch(x, w_prob, nullptr, nullptr, nullptr, prob, B, D, h, w, 8, 1, 0, st, ((impl & 2) ? 16 : 0) | (zc_force << 8));
    if (rc) return rc;
    const long long hw = (long long)h * w;
    const bool keep = !(impl & 8);
#define RCMVS_SOFTMAX(LP) do { if (keep) hipLaunchKernelGGL((softmax_regress_kernel<LP, true>), dim3((unsigned)cdiv(hw * LP, 256), B), dim3(256), 0, st, prob, planes, depth, conf, D, hw); \
                               else hipLaunchKernelGGL((softmax_regress_kernel<LP, false>), dim3((unsigned)cdiv(hw * LP, 256), B), dim3(256), 0, st, prob, planes, depth, conf, D, hw); } while (0)
    if (D <= 16) RCMVS_SOFTMAX(1);
    else if (D <= 32) RCMVS_SOFTMAX(2);
    else RCMVS_SOFTMAX(4);
#undef RCMVS_SOFTMAX
    return launch_status("depth_head_fwd");
}

/* the second launch of the head on its own: logits (B, D, h, w) -> probabilities in place (keep != 0), depth, confidence.  For producers that
 * leave the logits themselves (rcmvs_conv11_prob_fwd). */
extern "C" int rcmvs_softmax_head_fwd(float* prob, const float* planes, float* depth, float* conf, int B, int D, int h, int w, int keep, void* stream) {
    RCMVS_REQUIRE(prob && planes && depth && conf, "softmax_head_fwd: null pointer");
    RCMVS_REQUIRE(B > 0 && D > 0 && h > 0 && w > 0, "softmax_head_fwd: bad sizes");
    RCMVS_REQUIRE(D <= 64, "softmax_head_fwd: at most 64 depth hypotheses per stage (got %d)", D);
    hipStream_t st = as_stream(stream);
    const long long hw = (long long)h * w;
#define RCMVS_SOFTMAX(LP) do { if (keep) hipLaunchKernelGGL((softmax_regress_kernel<LP, true>), dim3((unsigned)cdiv(hw * LP, 256), B), dim3(256), 0, st, prob, planes, depth, conf, D, hw); \
                               else hipLaunchKernelGGL((softmax_regress_kernel<LP, false>), dim3((unsigned)cdiv(hw * LP, 256), B), dim3(256), 0, st, prob, planes, depth, conf, D, hw); } while (0)
    if (D <= 16) RCMVS_SOFTMAX(1);
    else if (D <= 32) RCMVS_SOFTMAX(2);
    else RCMVS_SOFTMAX(4);
#undef RCMVS_SOFTMAX
    return launch_status("softmax_head_fwd");
}

extern "C" int rcmvs_depth_head_fwd(const float* x, const float* w_prob, const float* planes, float* depth, float* conf, float* prob,
                                    int B, int D, int h, int w, void* stream) {
    return depth_head_fwd(x, nullptr, w_prob, planes, depth, conf, prob, B, D, h, w, 0, stream);
}
extern "C" int rcmvs_depth_head_scaled_fwd(const float* x, const float* x_absmax, const float* w_prob, const float* planes, float* depth, float* conf,
                                           float* prob, int B, int D, int h, int w, int impl, void* stream) {
    return depth_head_fwd(x, x_absmax, w_prob, planes, depth, conf, prob, B, D, h, w, impl, stream);
}
