/*
 * rcmvs.h -- C ABI of librcmvs_hip.so: the MI355X (gfx950) plane-sweep hot path of RC-MVSNet.
 *
 * The reference (Boese0601/RC-MVSNet) has no FFI layer: its hot path is a composition of ATen
 * ops inside Python nn.Modules.  Each entry point below replaces one
 * such composition; the reference call site it replaces is cited (paths relative to the
 * reference repository root).  The Python modules in rc_mvsnet_amd/ bind these with ctypes.
 *
 * Conventions
 *   - every function is extern "C", returns int: 0 = ok, <0 = bad argument (see
 *     rcmvs_last_error_string), >0 = a hipError_t from the launch;
 *   - every pointer is a DEVICE pointer owned by the caller unless marked [host]; nothing is
 *     allocated, freed or synchronised inside; kernels are enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the null stream); calls are re-entrant;
 *   - all tensors are dense fp32.  Feature maps and volumes are CHANNELS-LAST inside the
 *     library:  maps (B,h,w,C), volumes (B,D,h,w,C)  ("NHWC"/"NDHWC"); the layout entry
 *     points convert from/to the reference's NCHW / NCDHW at the module boundary;
 *   - `planes` is the per-pixel hypothesis-plane table (B,h,w,2) = {d_0, delta}: plane k of a
 *     pixel lies at depth d_0 + k*delta (what models/modules.py:549-588 materialises as a
 *     (B,D,H,W) tensor).
 */
#ifndef RCMVS_H
#define RCMVS_H

#ifdef __cplusplus
extern "C" {
#endif

#define RCMVS_VERSION 106          /* 0.1.6 -- 106: + rcmvs_conv2d_stem_fwd, rcmvs_conv2d_tile_fwd (+ pack, floats each).  105: + rcmvs_conv11_prob_fwd, rcmvs_softmax_head_fwd, rcmvs_resize_rgb_cl, rcmvs_warp_variance_timed_fwd, rcmvs_conv2d_pair_fwd (+ pack, floats).  104: rcmvs_warp_variance_fwd is the exact kernel again for every V and C (bit-identical to the reference-order variant 2);
                                      the FMA-contracted forms are opted into with RCMVS_K1_FAST_BLEND of rcmvs_warp_variance_hint_fwd; the plane-pipelined form
                                      (variant 7) takes 2, 3, 4 or 6 source views.
                                      103: + rcmvs_warp_variance_hint_fwd, rcmvs_debug_warp_variance_win_fwd; rcmvs_debug_warp_variance_fwd takes variants 0-3, 5-7; rcmvs_warp_variance_fwd is FMA-contracted for V = 3, C = 8;
                                      rcmvs_compose_homography_stages gained (zero, zero_n) before its stream argument in 102 (not listed then).  102: rcmvs_depth_head_fwd accepts prob == NULL for D = 8; 101: rcmvs_bn_stats writes 2C + 1 doubles (the row count joined the sums: one SyncBatchNorm message);
                                      rcmvs_debug_warp_variance_fwd takes variants 0-3 only.  A caller built against 100 must be rebuilt: check
                                      rcmvs_version() >= the RCMVS_VERSION it was compiled with. */
#define RCMVS_MAX_SRC_VIEWS 10     /* V-1 */

int         rcmvs_version(void);
const char* rcmvs_last_error_string(void);   /* thread-local, valid until the next failing call */

/* ---- layout (module boundary) -------------------------------------------------------- */
/* (N,C,S) -> (N,S,C) with S = h*w or D*h*w, and back. */
int rcmvs_nchw_to_nhwc(const float* src, float* dst, int N, int C, long long S, void* stream);
int rcmvs_nhwc_to_nchw(const float* src, float* dst, int N, int C, long long S, void* stream);

/* ---- homography  (models/casmvsnet.py:267-270, models/modules.py:314-316) ------------- */
/* proj (B,V,2,4,4): [..,0,:,:] extrinsic, [..,1,:3,:3] intrinsic.  For every source view
 * v=1..V-1:  P = (K_v E_v) * inverse(K_0 E_0)  (4x4 with last row 0,0,0,1), evaluated in fp64
 * on the device and rounded to fp32:  rot (B,V-1,9) row-major 3x3, trans (B,V-1,3). */
int rcmvs_compose_homography(const float* proj, float* rot, float* trans, int B, int V, void* stream);
/* The same for up to four cascade stages in ONE launch (their projection tensors differ only in the intrinsics scale, casmvsnet.py:
 * 376-381): rot (nstage, B, V-1, 9), trans (nstage, B, V-1, 3); unused proj pointers may be NULL.  zero / zero_n: an optional float
 * buffer (NULL / 0 = none) the launch clears on the side -- the scene's activation-bound rows, so that the first launch of a scene also
 * does its one fill. */
int rcmvs_compose_homography_stages(const float* proj0, const float* proj1, const float* proj2, const float* proj3, int nstage,
                                    float* rot, float* trans, int B, int V, float* zero, long long zero_n, void* stream);

/* ---- hypothesis planes  (models/casmvsnet.py:357-359,383-404; modules.py:549-588) ------ */
/* Stage 1 (prev_depth == NULL): d_0 = depth_values[b,0], delta = (dv[b,ND-1]-dv[b,0])/(D-1).
 * Later stages: prev_depth (B,hp,wp) is bilinearly up-sampled to (H,W) (align_corners=False),
 * the range  c -/+ D/2 * ratio * (dv[0,ND-1]-dv[0,0])/ND  is formed per full-res pixel and
 * averaged over the scale x scale block of each stage pixel (the reference's trilinear
 * down-sampling with integer scale 1 or 2).  planes (B,H/scale,W/scale,2). */
int rcmvs_hypothesis_planes(const float* prev_depth, const float* depth_values, float* planes,
                            int B, int hp, int wp, int H, int W, int scale,
                            int D, float ratio, int ND, void* stream);

/* ---- K1: fused warp + variance cost volume --------------------------------------------- */
/* replaces homo_warping (models/modules.py:304-339) x (V-1) plus the sum / square-sum /
 * variance chain of DepthNet_eval.forward (models/casmvsnet.py:257-288).
 *   feats (B,V,h,w,C) channels-last, view 0 = reference;  rot/trans from
 *   rcmvs_compose_homography;  planes (B,h,w,2);  var (B,D,h,w,C) = sum(x^2)/V - (sum(x)/V)^2.
 * C in {8,16,32}. */
int rcmvs_warp_variance_fwd(const float* feats, const float* rot, const float* trans,
                            const float* planes, float* var,
                            int B, int V, int C, int D, int h, int w, void* stream);

/* Test / profiling twin of rcmvs_warp_variance_fwd with an explicit code variant (stateless, re-entrant): 0 = the production
 * kernel, 1 = production with FMA-contracted blend (<= 2e-7 relative), 2 = reference-order kernel (one full coordinate chain per
 * lane, compiler IEEE division -- the kernel the production one is held bit-identical to), 3 = store-only ablation,
 * 5 / 6 = the LDS-window form (V = 3 only), 7 = the plane-pipelined gather form (V - 1 in {2, 3, 4, 6});
 * both: same sampling positions as 2, FMA-contracted blend: <= 2e-6 of the value range from 2.  rcmvs_warp_variance_fwd itself runs 0. */
int rcmvs_debug_warp_variance_fwd(const float* feats, const float* rot, const float* trans,
                                  const float* planes, float* var,
                                  int B, int V, int C, int D, int h, int w, int variant, void* stream);

/* rcmvs_warp_variance_fwd with a hint about the plane table.  RCMVS_K1_UNIFORM_PLANES: the hypothesis planes are the same (or nearly
 * the same) for every pixel -- stage 1 of the cascade (models/modules.py:549-566) -- so the 2x2 footprints of a tile over a chunk of planes
 * fit a small source window: with two source views the kernel stages that window in LDS and takes the taps from there (csrc/k1_win.h;
 * tiles whose footprints do not fit fall back to gathers one by one, so the hint never changes results beyond the kernel's FMA-level
 * tolerance, ~4e-7 of the value range, only the speed; other view counts ignore this bit).  It takes effect together with RCMVS_K1_FAST_BLEND.
 * RCMVS_K1_FAST_BLEND: the caller accepts FMA-contracted blend / variance arithmetic (same sampling positions, results within 2e-6 of the value
 * range of rcmvs_warp_variance_fwd's): the call then runs the fastest kernel measured for its view count and channel count (the window form,
 * the plane-pipelined gather form csrc/k1_pp.h, or the FMA build of the two-phase kernel; the table is in csrc/warp_variance.hip).
 * hint 0 = rcmvs_warp_variance_fwd (the exact two-phase kernel, bit-identical to the reference's operation order). */
#define RCMVS_K1_UNIFORM_PLANES 1
#define RCMVS_K1_FAST_BLEND 2
int rcmvs_warp_variance_hint_fwd(const float* feats, const float* rot, const float* trans,
                                 const float* planes, float* var,
                                 int B, int V, int C, int D, int h, int w, int hint, void* stream);

/* rcmvs_warp_variance_hint_fwd whose kernel leaves its own start / stop timestamps in two caller-owned hipEvent_t (hipExtLaunchKernelGGL; either may be
 * NULL): the launch duration a profiler reports, without the marker packets that event records around a launch add.  For measurement (bench.py's
 * roofline object); the results are those of the hint entry. */
int rcmvs_warp_variance_timed_fwd(const float* feats, const float* rot, const float* trans,
                                  const float* planes, float* var,
                                  int B, int V, int C, int D, int h, int w, int hint, void* start_event, void* stop_event, void* stream);

/* The window-form kernel (csrc/k1_win.h) with its tile statistics: variant 5 = source windows loaded
 * ahead of the coordinate phase, 6 = after the fit test.  stats (device pointer, two unsigned, caller-zeroed, NULL = none):
 * stats[0] += thread blocks launched, stats[1] += blocks whose tile took the LDS-window path. */
int rcmvs_debug_warp_variance_win_fwd(const float* feats, const float* rot, const float* trans,
                                      const float* planes, float* var,
                                      int B, int V, int C, int D, int h, int w, int variant, unsigned* stats, void* stream);

/* train-variant extra (models/casmvsnet.py:59,82,89-101): volume_feature_no_ref, NCDHW like
 * the reference returns it: out (B, 3(V-1)+C, D, h, w) = warped RGB of each source view
 * (imgs (B,V,h,w,3) channels-last, already resized to the stage) then the source-only
 * variance divided by V.  square_first != 0 reproduces the eval-mode quirk (:92-96). */
int rcmvs_warp_noref_fwd(const float* feats, const float* imgs, const float* rot, const float* trans,
                         const float* planes, float* out,
                         int B, int V, int C, int D, int h, int w, int square_first, void* stream);

/* K1 backward: gradient w.r.t. the feature maps (the only differentiable inputs: homo_warping
 * builds its grid under torch.no_grad(), models/modules.py:313).  Replaces the autograd graph of
 * grid_sample x (V-1) + the variance chain (models/casmvsnet.py:59-101).
 *   grad_var   (B,D,h,w,C) channels-last: d loss / d var;
 *   grad_noref (B,D,h,w,C) channels-last or NULL: d loss / d (source-only variance), i.e. the last C
 *              channels of volume_feature_no_ref (train mode, square_first == 0);
 *   grad_feats (B,V,h,w,C): MUST be zero-filled by the caller; views 1.. are accumulated with
 *              hardware fp32 atomics (unordered, like grid_sample's backward), view 0 is stored. */
int rcmvs_warp_variance_bwd(const float* feats, const float* rot, const float* trans, const float* planes,
                            const float* grad_var, const float* grad_noref, float* grad_feats,
                            int B, int V, int C, int D, int h, int w, void* stream);
/* Test / ablation twin of rcmvs_warp_variance_bwd (no reference counterpart).  variant bit 0: everything but the atomic scatter (timing
 * floor; the source-view gradients are NOT produced); bit 1: no run-length merging of consecutive planes' footprints. */
int rcmvs_debug_warp_variance_bwd(const float* feats, const float* rot, const float* trans, const float* planes,
                            const float* grad_var, const float* grad_noref, float* grad_feats,
                            int B, int V, int C, int D, int h, int w, int variant, void* stream);

/* ---- K2/K3: 3-D convolution family, channels-last, fused epilogue ----------------------- */
/* weight packing (host-visible layout change, done once per weight update):
 *   conv   w (Co,Ci,3,3,3) -> packed [27][Ci][Co]      (nn.Conv3d,          modules.py:145)
 *   deconv w (Ci,Co,3,3,3) -> packed [27][Ci][Co]      (nn.ConvTranspose3d, modules.py:189); transposed == 2 also
 *          flips the taps: the adjoint of a stride-1 nn.Conv3d with weight (Ci,Co,3,3,3) (data gradient)
 * followed, for channel pairs served by the MFMA kernels (Co a multiple of 16), by the
 * fragment-ordered image [27][Ci/(4*VEC)][Co/16][64 lanes][VEC] those kernels read. */
long long rcmvs_packed_weight_floats(int Co, int Ci);   /* size of `packed` in floats ([27][Ci][Co] + MFMA image) */
int rcmvs_pack_conv3d_weight(const float* w, float* packed, int Co, int Ci, int transposed, void* stream);
/* Selective form (no reference counterpart: the reference's weights are consumed in place by cuDNN).  A blob holds several images of
 * the same weight, one per kernel family (bit 0: direct / LDS-halo layout, bit 1: fp32 matrix-core fragments, 4 << k: split-bf16 image
 * of kind k = 0 stride 1, 1 stride 2, 2 transposed, 3 planar; 256 << k: fp16-pair image of kind k).  rcmvs_conv3d_images tells which ONE
 * rcmvs_conv3d_fwd (transposed = 0; planar = the volume is one plane deep) / rcmvs_deconv3d_fwd (transposed = 1) reads for a layer;
 * rcmvs_pack_conv3d_weight_sel writes only the images in `images` (training re-packs every weight every step for one call each).
 * Calling a forward entry on a blob whose image was not written is undefined: the Python host keeps the mask next to the blob and checks. */
int rcmvs_conv3d_images(int Co, int Ci, int stride, int transposed, int planar);
int rcmvs_pack_conv3d_weight_sel(const float* w, float* packed, int Co, int Ci, int transposed, int images, void* stream);

/* y = epilogue(conv3d(x, w, k=3, pad=1, stride)),  x (B,D,H,W,Ci) -> y (B,Do,Ho,Wo,Co),
 * Do = (D-1)/stride+1 ...;   epilogue(v) = [relu](v*scale[co] + shift[co]) + residual
 * (scale/shift/residual may be NULL).  Replaces Conv3d.forward = conv+BN(eval)+ReLU
 * (models/modules.py:149-157) and the skip adds of CostRegNet.forward (:497-499). */
int rcmvs_conv3d_fwd(const float* x, const float* w_packed, const float* scale, const float* shift,
                     const float* residual, float* y,
                     int B, int D, int H, int W, int Ci, int Co, int stride, int relu, void* stream);

/* y = epilogue(conv_transpose3d(x, w, k=3, stride=2, pad=1, output_pad=1)),
 * x (B,D,H,W,Ci) -> y (B,2D,2H,2W,Co).  Replaces Deconv3d.forward (models/modules.py:196-204). */
/* A 5x5 stride-2 (pad 2) 2-D convolution + BN(eval) + ReLU (models/modules.py:53-59 with the FeatureNet arguments of :418-423) as a
 * 3x3 stride-1 convolution of the space-to-depth view of its input, on the planar split-bf16 kernel, WITHOUT materialising that view:
 * x (N,H,W,C) channels-last, H and W even; w_packed = rcmvs_pack_conv3d_weight of the (Co, 4C, 3,3,3) weight whose middle depth slice
 * holds the re-indexed 5x5 taps (tap k = 2t + parity per axis, channel order (row parity, column parity, c)); y (N,H/2,W/2,Co).
 * Supported (4C, Co): (32,16), (64,32). */
int rcmvs_conv2d_s2d_fwd(const float* x, const float* w_packed, const float* scale, const float* shift, float* y,
                         int N, int H, int W, int C, int Co, int relu, void* stream);

int rcmvs_deconv3d_fwd(const float* x, const float* w_packed, const float* scale, const float* shift,
                       const float* residual, float* y,
                       int B, int D, int H, int W, int Ci, int Co, int relu, void* stream);

/* Test / A-B twins of the two entry points above with an explicit kernel selection `impl` (stateless: the library keeps no
 * dispatch state).  0 = the production dispatch; bit 0 = direct (one thread per voxel) kernels only; bits 1-3 and 5 = tuning
 * configuration of the LDS-halo kernel; bit 4 = fp32-MFMA kernel before the LDS kernel where both exist; bit 6 = skip the
 * split-bf16 MFMA kernels (the fp32 FMA-chain kernels: the comparator of tests/test_gpu_parity.py::test_conv3d_x3_*);
 * bits 8-15 = cap on the persistent blocks of the split-bf16 kernels (0 = one per CU). */
int rcmvs_debug_conv3d_fwd(const float* x, const float* w_packed, const float* scale, const float* shift,
                           const float* residual, float* y,
                           int B, int D, int H, int W, int Ci, int Co, int stride, int relu, int impl, void* stream);
int rcmvs_debug_deconv3d_fwd(const float* x, const float* w_packed, const float* scale, const float* shift,
                             const float* residual, float* y,
                             int B, int D, int H, int W, int Ci, int Co, int relu, int impl, void* stream);

/* Activation bounds (round 3).  A bound is a device vector of RCMVS_ABSMAX_FLOATS floats: 64 slots, 16 floats apart, the bound is the
 * maximum of the slots (the kernels that maintain one issue ONE atomic max per block into slot block & 63: same-address atomics
 * serialise).  The caller zero-fills it.  rcmvs_absmax_fwd computes the bound of a tensor nobody maintained one for: max|x| over n
 * floats, or its square (square != 0: the bound of a variance from the bound of its samples, models/casmvsnet.py:288). */
#define RCMVS_ABSMAX_FLOATS 1024
int rcmvs_absmax_fwd(const float* x, long long n, int square, float* amax, void* stream);
/* The same two layers with activation bounds.  x_absmax: bound of max|x| in the format above (NULL = none known); where the channel
 * pair has the kernel, the convolution then runs on the fp16 matrix cores with TWO fp16 pieces per fp32 operand after an exact
 * power-of-two pre-scale derived from the bound (|x - (h + l) / s| <= 2^-22 |x| down to 2^-17 of the bound, 2^-39 of the bound below
 * that; three MFMAs per product instead of the six of the exact three-piece bf16 split; csrc/conv3d_x3.hip).  y_absmax (NULL = not
 * wanted): bound vector that receives max|y| -- the caller zero-fills it and hands it to the next layer as its
 * x_absmax; only the split-operand matrix-core kernels maintain it (error for channel pairs without one).  impl: the kernel
 * selector of rcmvs_debug_conv3d_fwd (0 = production); bit 24 (planar layers only): y_absmax receives the SQUARE of max|y| -- the bound of
 * a variance volume from the bound of its samples, so FeatureNet's output convs leave the bound the cost regularisation needs and no
 * pass over the feature maps is required.  A bound that is too small by a factor 2^k costs k bits of fp16 range at the
 * top (overflow to inf beyond 2^1), one that is too large only raises the absolute floor: pass a true upper bound. */
int rcmvs_conv3d_scaled_fwd(const float* x, const float* x_absmax, const float* w_packed, const float* scale, const float* shift,
                            const float* residual, float* y, float* y_absmax,
                            int B, int D, int H, int W, int Ci, int Co, int stride, int relu, int impl, void* stream);
/* The last transposed layer of the 3-D U-Net and the prob conv in ONE pass (csrc/conv11_prob.hip): replaces
 *   x = conv0 + self.conv11(x);  x = self.prob(x)        (models/modules.py:497-500; Deconv3d 16 -> 8 + BatchNorm + ReLU, Conv3d 8 -> 1 without bias)
 * for a B = 1 inference scene, fp16-pair arithmetic -- the 8-channel full-resolution volume between the two layers never reaches memory.
 *   t (B,Dt,Ht,Wt,16) with t_absmax (bound of max|t|, RCMVS_ABSMAX_FLOATS slot format);  w11_packed = rcmvs_pack_conv3d_weight(Co = 8, Ci = 16,
 *   transposed = 1);  scale / shift: the folded BatchNorm of conv11 (8 floats each);  res (B,2Dt,2Ht,2Wt,8) = conv0's output with res_absmax;
 *   coef: two device floats {c1, c2} with max|conv11 output before the skip| <= c1 max|t| + c2 (c1 = max_co |scale_co| max over the 8 output parity
 *   classes of sum_{ci, taps of the class} |w|, c2 = max_co |shift_co|): the scale of the intermediate volume's fp16 pieces comes from
 *   max|res| + c1 max|t| + c2;  wprob_packed = rcmvs_pack_conv3d_weight(Co = 1, Ci = 8);  logits (B,2Dt,2Ht,2Wt): what rcmvs_softmax_head_fwd takes.
 *   With 2 Dt = 8 planes (the cascade's last stage) the whole head fits the launch: depth != NULL (then planes (B,2Ht,2Wt,2) and conf too) ->
 *   softmax, soft-argmin and the confidence window finish in the same kernel (the arithmetic of rcmvs_softmax_head_fwd); logits may then be NULL.
 *   zchunk: output planes per block (even; 0 = chosen per launch).  Results: within the fp16-pair tolerance of the two-launch form
 *   (rcmvs_deconv3d_scaled_fwd + the prob conv of rcmvs_depth_head_scaled_fwd); tests/test_gpu_parity.py::test_conv11_prob_*. */
int rcmvs_conv11_prob_fwd(const float* t, const float* t_absmax, const float* w11_packed, const float* scale, const float* shift,
                          const float* res, const float* res_absmax, const float* coef, const float* wprob_packed, float* logits,
                          const float* planes, float* depth, float* conf,
                          int B, int Dt, int Ht, int Wt, int zchunk, void* stream);
int rcmvs_deconv3d_scaled_fwd(const float* x, const float* x_absmax, const float* w_packed, const float* scale, const float* shift,
                              const float* residual, float* y, float* y_absmax,
                              int B, int D, int H, int W, int Ci, int Co, int relu, int impl, void* stream);

/* ---- training: the 3-D blocks with BATCH statistics, forward and backward ---------------- */
/* Conv3d / Deconv3d in train mode = conv -> BatchNorm3d(batch stats) -> ReLU (models/modules.py:149-157,
 * 196-204).  The convolution is rcmvs_conv3d_fwd / rcmvs_deconv3d_fwd with a NULL epilogue; autograd's
 * backward of the block is rcmvs_bn_bwd_* + the data gradient (the same forward kernels on re-packed
 * weights: dgrad(conv s1) = conv with flipped, transposed weights; dgrad(conv s2) = deconv; dgrad(deconv) =
 * conv s2) + rcmvs_conv3d_wgrad.  All tensors channels-last, rows = B*D*H*W.
 *   rcmvs_bn_stats:          sums[0..C) += sum_rows x, sums[C..2C) += sum_rows x^2, sums[2C] += rows     (fp64; SINCE VERSION 101 these 2C + 1
 *                            doubles are what a SyncBatchNorm all-reduce exchanges).  The buffer starts at zero ONCE: the
 *                            finalize calls below clear what they consumed, so one buffer per layer serves every step.
 *   rcmvs_scale_shift_relu:  y = [relu](x*scale[c] + shift[c]) + residual   (scale/shift/residual may be NULL)
 *   rcmvs_bn_bwd_reduce:     sums[0..C) += sum g, sums[C..2C) += sum g*xhat,  g = dz*[y*scale+shift > 0] (relu) or dz
 *   rcmvs_bn_bwd_apply:      dy = scale * (g - coef[c] - xhat*coef[C+c]),  coef = {dbeta/N, dgamma/N},
 *                            xhat = (y-mean)*invstd, scale = gamma*invstd */
int rcmvs_bn_stats(const float* x, double* sums, long long rows, int C, void* stream);
/* sums (2C + 1, fp64: consumed and cleared) -> batch mean / biased var / invstd, scale = gamma*invstd, shift = beta - mean*scale, and
 * (when given) the momentum update of running_mean / running_var (unbiased), exactly nn.BatchNorm's bookkeeping; *count receives the
 * row count (sums[2C]) for the backward pass. */
int rcmvs_bn_finalize(double* sums, double* count, const float* gamma, const float* beta, float eps, float momentum,
                      float* mean, float* var, float* invstd, float* scale, float* shift,
                      float* running_mean, float* running_var, int C, void* stream);
/* local sums (2C: consumed and cleared) -> dgamma, dbeta of this replica; total (all-reduced) sums / *count -> coef for rcmvs_bn_bwd_apply */
int rcmvs_bn_bwd_finalize(double* local_sums, const double* total_sums, const double* count, float* dgamma, float* dbeta,
                          float* coef, int C, void* stream);
int rcmvs_scale_shift_relu(const float* x, const float* scale, const float* shift, const float* residual, float* y,
                           long long rows, int C, int relu, void* stream);
/* Fused forms, one launch instead of two (what the training path calls):
 *   rcmvs_bn_norm_fwd = rcmvs_bn_finalize + rcmvs_scale_shift_relu: y = [relu](x*scale + shift) + residual with scale / shift derived from
 *     `sums` (2C + 1 doubles, NOT modified); stats (5 x C floats) receives mean | biased var | invstd | scale | shift, *count the row
 *     count; running statistics updated when given.
 *   rcmvs_bn_norm_bwd = rcmvs_bn_bwd_finalize + rcmvs_bn_bwd_apply, with `stats` as written by the forward form.
 * Neither can clear the sums it reads (other blocks are still reading them): the caller alternates between TWO accumulation buffers
 * per layer and passes the one the PREVIOUS call of that layer consumed as `clear` (2C + 1 / 2C doubles, zeroed on return). */
int rcmvs_bn_norm_fwd(const float* x, const double* sums, double* clear, const float* gamma, const float* beta, float eps, float momentum,
                      float* stats, double* count, float* running_mean, float* running_var, const float* residual, float* y,
                      long long rows, int C, int relu, void* stream);
int rcmvs_bn_norm_bwd(const float* y, const float* dz, const float* stats, const double* local_sums, const double* total_sums,
                      const double* count, double* clear, float* dgamma, float* dbeta, float* dy, long long rows, int C, int relu, void* stream);
int rcmvs_bn_bwd_reduce(const float* y, const float* dz, const float* scale, const float* shift, const float* mean,
                        const float* invstd, double* sums, long long rows, int C, int relu, void* stream);
int rcmvs_bn_bwd_apply(const float* y, const float* dz, const float* scale, const float* shift, const float* mean,
                       const float* invstd, const float* coef, float* dy, long long rows, int C, int relu, void* stream);
/* dw[27][Ci][Co] += sum_o x[stride*o + tap - 1][ci] * dy[o][co]   (dw zero-filled by the caller, fp32 atomics).
 *   x (B,D,H,W,Ci), dy (B,Do,Ho,Wo,Co) with Do = (D-1)/stride+1 ...  nn.Conv3d weight grad = dw permuted to
 *   (Co,Ci,27); for nn.ConvTranspose3d(Cin,Cout) call with x := grad of the (large) output, dy := the (small)
 *   input, stride 2: dw[27][Cout][Cin] -> permute to (Cin,Cout,27). */
int rcmvs_conv3d_wgrad(const float* x, const float* dy, float* dw, int B, int D, int H, int W, int Ci, int Co, int stride,
                       void* stream);
/* packed gradient [27][P][Q] (as rcmvs_conv3d_wgrad accumulates it) -> out (Q, Pk, 27) = nn.Conv3d's weight layout (Co, Ci, 27) /
 * nn.ConvTranspose3d's (Cin, Cout, 27) with the roles swapped; rows p >= Pk (padding channels) are dropped; `packed` is ZEROED, ready to
 * accumulate the next gradient of that shape (what autograd does with a permute + reshape copy, plus the zero fill before). */
int rcmvs_wgrad_finish(float* packed, float* out, int P, int Q, int Pk, void* stream);
/* data gradient of the 1-output-channel prob conv (modules.py:489): dy (B,D,H,W), w (1,Ci,3,3,3) as stored, dx (B,D,H,W,Ci) */
int rcmvs_conv3d_dgrad_c1(const float* dy, const float* w, float* dx, int B, int D, int H, int W, int Ci, void* stream);
/* softmax + soft-argmin backward (casmvsnet.py:299-300): grad_logits = prob * (d_k - depth) * grad_depth */
int rcmvs_depth_head_bwd(const float* prob, const float* planes, const float* depth, const float* grad_depth,
                         float* grad_logits, int B, int D, int h, int w, void* stream);

/* ---- 2-D feature pyramid (FeatureNet fpn, models/modules.py:363-464), channels-last, inference -------- */
/* imgs (N,3,H,W) NCHW -> (N,H,W,4) with a zero 4th channel (16-byte input vectors). */
int rcmvs_rgb_to_nhwc4(const float* x, float* y, int N, int H, int W, void* stream);
/* w (Co,Ci,K,K) -> packed [K*K][Cip][Co], input channels zero-padded to Cip. */
int rcmvs_pack_conv2d_weight(const float* w, float* packed, int Co, int Ci, int Cip, int K, void* stream);
/* y = [relu]( up2(up_add) + conv2d(x, w, K, pad K/2, stride) * scale + shift ),  x (N,H,W,Ci) -> y (N,Ho,Wo,Co);
 * scale / shift / up_add may be NULL (shift alone = plain bias; up_add (N,Ho/2,Wo/2,Co) is added after
 * nearest x2 up-sampling: the FPN merge `F.interpolate(intra) + inner(conv)`, modules.py:448-455).
 * Replaces Conv2d.forward = conv + BN(eval) + ReLU (modules.py:53-59) and the bare 1x1 / 3x3 output convs.
 * Ci == 3 (3 -> 8, K = 3, stride 1: FeatureNet's first layer): x is the planar (N, 3, H, W) image batch as the network receives it and
 * w_packed the layer's weight packed with Cip = 4 -- no rcmvs_rgb_to_nhwc4 pass in front. */
int rcmvs_conv2d_fwd(const float* x, const float* w_packed, const float* scale, const float* shift, const float* up_add,
                     float* y, int N, int H, int W, int Ci, int Co, int K, int stride, int relu, void* stream);
/* The 1x1 layers of the pyramid (out1 32 -> 32, inner1 16 -> 32, inner2 8 -> 32; models/modules.py:437-447) on a streaming kernel: same
 * arguments and results as rcmvs_conv2d_fwd with K = 1 (which routes here), plus ysq_absmax (may be NULL): a zero-filled bound vector
 * (RCMVS_ABSMAX_FLOATS floats) that receives (max |y|)^2 -- the bound of the variance volume built from the map. */
int rcmvs_conv1x1_fwd(const float* x, const float* w_packed, const float* scale, const float* shift, const float* up_add,
                      float* y, float* ysq_absmax, int N, int H, int W, int Ci, int Co, int relu, void* stream);
/* the same layer on the matrix cores, exact (three bf16 pieces per operand, six MFMAs per product): 16 -> 32 and 32 -> 32; equal to
 * rcmvs_conv1x1_fwd up to fp32 summation order */
int rcmvs_conv1x1_mfma_fwd(const float* x, const float* w_packed, const float* scale, const float* shift, const float* up_add,
                      float* y, float* ysq_absmax, int N, int H, int W, int Ci, int Co, int relu, void* stream);
/* Last FPN level in one launch: y = conv3x3(up2(up) + conv1x1(lat) + b_inner) without materialising the 32-channel
 * intermediate (models/modules.py:448-462: `intra_feat = F.interpolate(intra_feat) + self.inner2(conv0)`,
 * `self.out3(intra_feat)`).  lat (N,H,W,CL), up (N,H/2,W/2,CM), w_inner packed [1][CL][CM], b_inner (CM), w_out packed
 * [9][CM][CO], y (N,H,W,CO); H, W even; channels 8 -> 32 -> 8.  Bit-identical to the two rcmvs_conv2d_fwd calls it replaces. */
int rcmvs_fpn_out_fused(const float* lat, const float* up, const float* w_inner, const float* b_inner, const float* w_out,
                        float* y, int N, int H, int W, int CL, int CM, int CO, void* stream);
/* The same level (8 -> 32 -> 8 channels) with the two convolutions folded into one 3x3 conv 8 -> 8 on `lat` plus, per output parity, a
 * 2x2 conv 32 -> 8 on `up` (the 3x3 taps that fall on the same pixel of the nearest-upsampled map summed), plus the lateral bias through
 * the taps inside the image: 1600 instead of 2628 multiply-adds per pixel.  tables (RCMVS_FPN_FOLDED_FLOATS floats, fp32):
 *   [0, 576)     WB[ky][kx][ci][co] = sum_cm w_inner[cm][ci] w_out[co][cm][ky][kx]
 *   [576, 648)   BS[cy][cx][co]     = sum over the taps inside the image (cy / cx = 0 first, 1 interior, 2 last row / column) of sum_cm b[cm] w_out[co][cm][ky][kx]
 *   [648, 4744)  WA[py][px][ry][rx][cm][co] = sum of w_out[co][cm][ky][kx] over the taps (ky, kx) with ((py + ky - 1) >> 1) - ((py - 1) >> 1) == ry, same in x
 * Equal to rcmvs_fpn_out_fused up to fp32 rounding (1e-6 relative). */
#define RCMVS_FPN_FOLDED_FLOATS 4744
int rcmvs_fpn_out_folded(const float* lat, const float* up, const float* tables, float* y, float* ysq_absmax, int N, int H, int W, void* stream);
/* ysq_absmax (may be NULL): bound vector (RCMVS_ABSMAX_FLOATS floats, zero-filled by the caller) that receives the square of max|y|. */
/* The same level on the matrix cores, exact (three bf16 pieces per operand, six MFMAs per product; csrc/fpn_folded_mfma.hip):
 * rcmvs_fpn_folded_mfma_pack turns the RCMVS_FPN_FOLDED_FLOATS tables into an image of rcmvs_fpn_folded_mfma_floats() floats,
 * rcmvs_fpn_out_folded_mfma takes that image in place of the tables; same arguments and result (to fp32 rounding) otherwise. */
long long rcmvs_fpn_folded_mfma_floats(void);
int rcmvs_fpn_folded_mfma_pack(const float* tables, float* image, void* stream);
int rcmvs_fpn_out_folded_mfma(const float* lat, const float* up, const float* image, float* y, float* ysq_absmax, int N, int H, int W, void* stream);


/* ---- K4: prob conv + softmax + soft-argmin + photometric confidence ---------------------- */
/* replaces CostRegNet.prob (models/modules.py:489,500), F.softmax, depth_regression and the
 * confidence gather of DepthNet_eval.forward (models/casmvsnet.py:293-309).
 *   x (B,D,h,w,8) channels-last, w_prob packed [27][8][1];  depth, conf (B,h,w);
 *   prob (B,D,h,w): receives the logits, then the probabilities in place; required except for D = 8 (the cascade's last stage),
 *   which runs as one launch with the logits in registers -- there it may be NULL (probabilities not stored). */
int rcmvs_depth_head_fwd(const float* x, const float* w_prob, const float* planes,
                         float* depth, float* conf, float* prob,
                         int B, int D, int h, int w, void* stream);
/* the same with an activation bound and a dispatch override.
 *   x_absmax (may be NULL): bound of max|x| in the RCMVS_ABSMAX_FLOATS slot format (rcmvs_absmax_fwd, or the y_absmax a scaled
 *     convolution left) -> the prob conv runs on the matrix cores in fp16 pairs (csrc/prob_pair.hip; w_prob must be the full blob
 *     of rcmvs_pack_conv3d_weight(Co = 1, Ci = 8)); NULL -> the exact fp32 form, as rcmvs_depth_head_fwd.
 *   impl (0 in production; tests and A/B timing): bit 0 = two launches also for D = 8, bit 1 = the generic (predicated) marching
 *     prob conv instead of the depth head's plain one, bit 2 = the fp32 form although a bound was given, bits 8-15 = z chunk;
 *     bit 3 (a production flag): prob is scratch only -- the probabilities are not written back (callers that want depth and confidence only). */
int rcmvs_depth_head_scaled_fwd(const float* x, const float* x_absmax, const float* w_prob, const float* planes,
                                float* depth, float* conf, float* prob,
                                int B, int D, int h, int w, int impl, void* stream);
/* The head's second half on its own (F.softmax, depth_regression, the confidence gather: models/casmvsnet.py:293-309) for producers that leave
 * the logits themselves (rcmvs_conv11_prob_fwd): prob (B,D,h,w) holds the logits on entry and, with keep != 0, the probabilities on return. */
int rcmvs_softmax_head_fwd(float* prob, const float* planes, float* depth, float* conf, int B, int D, int h, int w, int keep, void* stream);

/* ---- rendering-consistency branch --------------------------------------------------------- */
/* F.interpolate(size=[Do,h,w], trilinear, align_corners=True) along the plane axis only
 * (models/render_models.py:756), NCDHW in -> channels-last out with the channel count zero-padded
 * to Cp >= C (so the first conv reads 16-byte vectors):  x (B,C,D,h,w) -> y (B,Do,h,w,Cp). */
int rcmvs_resize_planes_fwd(const float* x, float* y, int B, int C, int Cp, int D, int Do, int h, int w, void* stream);
/* adjoint of rcmvs_resize_planes_fwd w.r.t. x (autograd of F.interpolate(trilinear, align_corners=True) along D,
 * models/render_models.py:756): g (B,Do,h,w,ldg) channels-last (first C channels used) -> gx (B,C,D,h,w). */
int rcmvs_resize_planes_bwd(const float* g, float* gx, int B, int C, int ldg, int D, int Do, int h, int w, void* stream);

/* Gaussian-Uniform ray sampler + world/NDC points (models/render_utils.py:86-108,149-243,
 * 112-146).  Random draws are inputs: pix (2,N) int32 rows x,y; eps (N,S); u (N/2,S).
 * cam = [K(9) | c2w(16) | w2c_ref(16) | K_ref(9) | near | far]  (52 floats, device).
 * Outputs: z (N,S) sorted Gaussian / stratified-uniform depths, pts (N,S,3), ndc (N,S,3),
 * dirs (N,3), rays_depth (N), target (N,3) from img0 (3,H,W) un-normalised. */
int rcmvs_gu_sample_fwd(const float* pseudo_depth, const float* img0, const int* pix,
                        const float* eps, const float* u, const float* cam,
                        float* z, float* pts, float* ndc, float* dirs, float* rays_depth, float* target,
                        int N, int S, int H, int W, void* stream);

/* point features (models/renderer.py:154-166, render_utils.py:247-279,304-330):
 * feat (M, ldf) row-major, columns [0, 8+4*nimg) written = trilinear(volume (Dv,hv,wv,8)
 * channels-last at ndc*2-1, zeros pad, align_corners) ++ per image i: bilinear border RGB of
 * imgs (nimg,3,H,W) at the projection with poses (nimg,25) = [w2c(16) | K(9)] ++ strict
 * in-bounds mask. */
int rcmvs_point_feats_fwd(const float* volume, const float* imgs, const float* poses,
                          const float* pts, const float* ndc, float* feat,
                          int M, int Dv, int hv, int wv, int nimg, int H, int W, int ldf, void* stream);

/* NeRF MLP (models/render_models.py:45-49,192-220; renderer.py:42-63,141-152): positional
 * encoding of ndc (10 freqs), 6x128 trunk with multiplicative feature bias, skip after layer 4,
 * sigma / rgb heads, view direction = normalised ray direction rotated by w2c_ref[:3,:3].
 *   rcmvs_pack_nerf_weights: wb [host array] of 22 device pointers, (weight, bias) of pts_bias,
 *     pts_linears.0..5, alpha_linear, feature_linear, views_linears.0, rgb_linear -> blob of
 *     rcmvs_nerf_weight_floats() floats (MFMA fragment images + biases).
 *   rcmvs_nerf_mlp_fwd: ndc (M,3); feat (M,ldf=32) with 20 used columns (padding is zeroed here);
 *     dirs (N,3) raw ray directions; w2c_ref (4,4); workspace of rcmvs_nerf_workspace_floats(M)
 *     floats; raw (M,4) = [rgb(3), sigma];  M = N*S. */
long long rcmvs_nerf_weight_floats(void);
long long rcmvs_nerf_workspace_floats(long long M);
int rcmvs_pack_nerf_weights(const float* const* wb, float* blob, void* stream);
int rcmvs_nerf_mlp_fwd(const float* ndc, float* feat, int ldf, const float* dirs, const float* w2c_ref,
                       const float* weights, float* workspace, float* raw, int N, int S, void* stream);
/* Renderer_ours.forward(x) / RenderNet.forward(x) called on their own (models/render_models.py:192-220,538-565): the rows of x are already
 * [embedded point (63) | point feature (20) | view direction (3)] (x (M, ldx >= 86)); feat32 = scratch (M, 32); raw (M, 4) = [rgb, sigma]. */
int rcmvs_nerf_mlp_embedded_fwd(const float* x, int ldx, const float* weights, float* workspace, float* feat32, float* raw, long long M, void* stream);

/* NeRF MLP in training (autograd of Renderer_ours.forward, models/render_models.py:192-220, called through
 * run_network_mvs, models/renderer.py:42-63; replaces the 11 nn.Linear forward/backward pairs of the reference).
 *   rcmvs_nerf_mlp_train_fwd: as rcmvs_nerf_mlp_fwd, but every layer's activations are kept in `workspace`
 *     (rcmvs_nerf_train_workspace_floats(M) floats) for the backward pass.
 *   rcmvs_nerf_mlp_bwd: wb = the 22 parameter pointers (host array, order of rcmvs_pack_nerf_weights); feat / tws / raw as
 *     given to / filled by the forward call; draw (M,4) = gradient of raw; gws = scratch of
 *     rcmvs_nerf_bwd_workspace_floats(M) floats; outputs (all written, not accumulated): dfeat (M,ldf) = gradient of the
 *     feature columns, dwb = 22 device pointers receiving the weight / bias gradients in the parameters' own shapes. */
long long rcmvs_nerf_train_workspace_floats(long long M);
long long rcmvs_nerf_bwd_workspace_floats(long long M);
int rcmvs_nerf_mlp_train_fwd(const float* ndc, float* feat, int ldf, const float* dirs, const float* w2c_ref,
                             const float* weights, float* workspace, float* raw, int N, int S, void* stream);
int rcmvs_nerf_mlp_bwd(const float* const* wb, const float* feat, int ldf, const float* tws, const float* raw, const float* draw,
                       float* gws, float* dfeat, float* const* dwb, int N, int S, void* stream);

/* compositing (models/renderer.py:18-26,65-93): alpha = 1-exp(-sigma), T = exclusive cumprod
 * of (1-alpha+1e-10), w = alpha*T;  rgb (N,3), depth (N), weights (N,S), alpha (N,S). */
int rcmvs_composite_fwd(const float* raw, const float* z, float* rgb, float* depth,
                        float* weights, float* alpha, int N, int S, void* stream);
/* backward of rcmvs_composite_fwd: gradients of rgb (N,3), depth (N), weights (N,S), alpha (N,S) (any may be NULL)
 * -> grad_raw (N,S,4) = d/d [rgb, sigma]  (autograd of renderer.py:18-26,65-93; division-free reverse recurrence). */
int rcmvs_composite_bwd(const float* raw, const float* z, const float* grad_rgb, const float* grad_depth,
                        const float* grad_weights, const float* grad_alpha, float* grad_raw, int N, int S, void* stream);
/* backward of rcmvs_point_feats_fwd w.r.t. the neural volume: grad_feat (M, ldg) (first 8 columns used) scattered to
 * grad_volume (Dv,hv,wv,8), which the caller zero-fills (fp32 atomics).  Images / coordinates carry no gradient. */
int rcmvs_point_feats_bwd(const float* ndc, const float* grad_feat, float* grad_volume,
                          int M, int Dv, int hv, int wv, int ldg, void* stream);

/* ---- self-supervised photometric loss -------------------------------------------------
 * Replaces losses/homography.py:6-200 (inverse_warping + _bilinear_sample), losses/modules.py:6-82 (SSIM,
 * depth_smoothness, compute_reconstr_loss) and the per-stage body of UnSupLoss.forward (losses/unsup_loss.py:14-94).
 * Images are channels-last (B,H,W,3) at the stage resolution; depth / mask (B,H,W); all device pointers.
 * coef (Vs,B,12) = per source view and batch item {M 3x3 row-major, t 3} with p = M (x,y,1)^T d + t, where
 * M = K_ref R_rel K_ref^-1, t = K_ref t_rel, R_rel = R_src R_ref^T, t_rel = t_src - R_rel t_ref (homography.py:9-56;
 * the reference projects with the REFERENCE view's intrinsics and that is kept). */
#define RCMVS_UNSUP_MAX_VIEWS 8
/* inverse_warping of one source image: warped (B,H,W,3), mask (B,H,W) in {0,1}; coef (B,12). */
int rcmvs_inverse_warp(const float* src, const float* depth, const float* coef, float* warped, float* mask,
                       int B, int H, int W, void* stream);
/* One stage of UnSupLoss.forward for Vs source views: srcs (Vs,B,H,W,3) -> warped (Vs,B,H,W,3), masks (Vs,B,H,W) (kept
 * for the backward), out[0..3] = {reconstr_loss, ssim_loss, smooth_loss, 12 r + 6 s + 0.18 m}, out[4+v] = the scalar
 * reconstruction loss of view v.  sums (4 Vs + 2 doubles) and counts (Vs ints: pixels won by each view) are workspace
 * the call zero-fills; counts feed the backward.  No host synchronisation. */
int rcmvs_unsup_loss_fwd(const float* ref, const float* srcs, const float* depth, const float* coef,
                         float* warped, float* masks, double* sums, int* counts, float* out,
                         int B, int Vs, int H, int W, void* stream);
/* Backward of rcmvs_unsup_loss_fwd w.r.t. depth (the images carry no gradient): gout = 3 device floats, the gradients of
 * {reconstr_loss, ssim_loss, smooth_loss}; grad_depth (B,H,W) is overwritten.  Workspace: ssim_ws (B,H-2,W-2,9) floats,
 * kbuf (4 Vs + 2) floats. */
int rcmvs_unsup_loss_bwd(const float* ref, const float* srcs, const float* depth, const float* coef,
                         const float* warped, const float* masks, const int* counts, const float* gout,
                         float* ssim_ws, float* kbuf, float* grad_depth, int B, int Vs, int H, int W, void* stream);
/* Masked smooth-L1 mean (losses/aug_loss.py:58-59 and losses/sl1loss.py:9-13: F.smooth_l1_loss(pred[mask], target[mask])):
 * sums[0] = sum of smooth_l1(pred - target) over elements with mask > 0.5, sums[1] = their number (fp64, zero-filled by the
 * call); the loss is sums[0] / sums[1].  Backward: grad_pred = gout[0] / sums[1] * clamp(pred - target, -1, 1) on the mask. */
int rcmvs_masked_sl1_fwd(const float* pred, const float* target, const float* mask, double* sums, long long n, void* stream);
int rcmvs_masked_sl1_bwd(const float* pred, const float* target, const float* mask, const double* sums,
                         const float* gout, float* grad_pred, long long n, void* stream);

/* ---- depth-map fusion filter -----------------------------------------------------------
 * Replaces reproject_with_depth / check_geometric_consistency (eval_rcmvsnet_dtu.py:281-338, eval_rcmvsnet_tanks.py:206-262)
 * and the per-reference-view body of filter_depth (:369-425) for one reference view and its N source views.
 * depth_all: every depth map of the scan, (n_views,H,W) fp32, resident on the device; ref_idx / src_idx_host (HOST array of
 * N ints) index it.  conf (H,W) photometric confidence, img (H,W,3) fp32 in [0,1] (may be NULL together with rgb).
 * mats (device, doubles, row-major; the reference's float32 numpy results promoted):
 *   [K_ref^-1 (9)][K_ref (9)][(E_ref^-1)[:3,:4] (12)] then per source view
 *   [(E_src E_ref^-1)[:3,:4] (12)][K_src (9)][K_src^-1 (9)][(E_ref E_src^-1)[:3,:4] (12)].
 * Outputs: masks (3,H,W) u8 = photo (conf > prob_thresh), geo (consistent views >= num_consistent), final;
 * depth_avg (H,W) = (sum of masked reprojected depths + depth_ref) / (consistent views + 1); xyz (H,W,3) world point of every
 * pixel at depth_avg, cast to fp32; rgb (H,W,3) u8 = (img * 255) truncated.  Optional per-source outputs (NULL to skip):
 * dbg_depth (N,H,W) the masked reprojected depth, dbg_geo (N,H,W) u8 the per-view consistency mask,
 * dbg_xy (N,H,W,2) the fp32 sampling position in the source view (the x2d_src / y2d_src the reference returns). */
#define RCMVS_FUSE_MAX_SRC 16
int rcmvs_fuse_view(const float* depth_all, int ref_idx, const int* src_idx_host, const float* conf, const float* img,
                    const double* mats, float prob_thresh, int num_consistent, double dist_thresh, float depth_thresh,
                    unsigned char* masks, float* depth_avg, float* xyz, unsigned char* rgb,
                    float* dbg_depth, unsigned char* dbg_geo, float* dbg_xy, int N, int H, int W, void* stream);
/* numpy's xyz[mask], rgb[mask] (row-major order kept): out_xyz (n,3) / out_rgb (n,3) sized for the worst case;
 * block_offsets = workspace of ceil(n / 256) + 1 ints whose LAST element receives the number of points kept. */
int rcmvs_compact_points(const unsigned char* mask, const float* xyz, const unsigned char* rgb, float* out_xyz,
                         unsigned char* out_rgb, int* block_offsets, long long n, void* stream);

/* ---- evaluation loader: image preparation ------------------------------------------------
 * Replaces read_img (/255), scale_mvs_input's cv2.resize and ToTensor + Normalize of datasets/dtu_test.py:78-81,107-112,
 * 127-145 (same in datasets/tanks.py): src = decoded image (H,W,3) uint8 on the device -> out (3,h,w) fp32 =
 * (bilinear_resize(src / 255) - mean[c]) / std[c] with cv2.resize's INTER_LINEAR coordinate rule (a copy when the size is
 * unchanged).  mean_host / std_host: HOST arrays of 3 floats. */
/* Two consecutive 3x3 stride-1 Conv2d blocks (conv + BatchNorm(eval) + ReLU, twice) of FeatureNet's trunk in one launch (csrc/conv2d_pair.hip): replaces
 *   x = self.conv1[1](x); x = self.conv1[2](x)        (models/modules.py:372-379,413-424; 16 -> 16 -> 16 channels)
 * exact split-bf16 matrix-core arithmetic like the planar kernels it replaces; the intermediate map stays in LDS.
 *   rcmvs_pack_conv2d_pair: wa, wb (16,16,3,3) Conv2d weights of the two layers -> image of rcmvs_conv2d_pair_weight_floats() floats;
 *   x, y (N,H,W,16) channels-last; scale / shift: the folded BatchNorm of each layer (16 floats each).  C must be 16. */
long long rcmvs_conv2d_pair_weight_floats(void);
int rcmvs_pack_conv2d_pair(const float* wa, const float* wb, float* image, void* stream);
int rcmvs_conv2d_pair_fwd(const float* x, const float* image, const float* scale_a, const float* shift_a, const float* scale_b, const float* shift_b,
                          float* y, int N, int H, int W, int C, void* stream);

/* FeatureNet's first block in one launch (csrc/conv2d_stem.hip): replaces
 *   conv0 = self.conv0(x)        (models/modules.py:372-373,413-415: Conv2d(3, 8, 3, 1) -> Conv2d(8, 8, 3, 1), each conv + BatchNorm(eval) + ReLU)
 * x (N,3,H,W) planar images as the reference hands them over -> y (N,H,W,8) channels-last.  First layer: fp32 FMA chain in the tap order of
 * rcmvs_conv2d_fwd (w_a_packed = that layer's weight from rcmvs_pack_conv2d_weight with the input channels padded to 4); second layer: exact
 * split-bf16 matrix-core arithmetic like the planar kernel it replaces (image_b from rcmvs_pack_conv2d_stem: (8,8,3,3) Conv2d weight ->
 * rcmvs_conv2d_stem_weight_floats() floats); the 8-channel map between the two layers stays in LDS.  scale / shift: the folded BatchNorm of
 * each layer (8 floats each). */
long long rcmvs_conv2d_stem_weight_floats(void);
int rcmvs_pack_conv2d_stem(const float* wb, float* image, void* stream);
int rcmvs_conv2d_stem_fwd(const float* x, const float* w_a_packed, const float* scale_a, const float* shift_a, const float* image_b, const float* scale_b,
                          const float* shift_b, float* y, int N, int H, int W, void* stream);

/* FeatureNet's 32-channel 3x3 layers as tile kernels (csrc/conv2d_tile.hip; Co = 16 or 32): replaces
 *   x = self.conv2[1](x); x = self.conv2[2](x)   (Co = 32, s2d = 0, one call per layer: models/modules.py:376-377,418-419 -- Conv2d(32, 32, 3, 1) + BatchNorm(eval) + ReLU)
 *   x = self.conv1[0](conv0)                  (s2d = 1: models/modules.py:374,416 -- Conv2d(8, 16, 5, stride=2, padding=2) + BatchNorm(eval) + ReLU as a 3x3 layer on the
 *                                              space-to-depth view: x is the (N,2H,2W,8) map, the weight the (16,32,3,3) re-indexed one, tap k = 2t + parity)
 *   out = self.out2(intra_feat)               (s2d = 0: models/modules.py:437,452 -- Conv2d(32, 16, 3, padding=1, bias=False); x (N,H,W,32))
 * y (N,H,W,Co) channels-last; exact split-bf16 matrix-core arithmetic like the planar kernel they replace.  scale / shift: folded BatchNorm (Co floats each) or
 * NULL; ysq_absmax (RCMVS_ABSMAX_FLOATS floats, zero-filled) or NULL: receives (max|y|)^2, the bound of the variance volume built from y.
 *   rcmvs_pack_conv2d_tile: w (Co,32,3,3) -> image of rcmvs_conv2d_tile_weight_floats(Co) floats. */
long long rcmvs_conv2d_tile_weight_floats(int Co);
int rcmvs_pack_conv2d_tile(const float* w, float* image, int Co, void* stream);
int rcmvs_conv2d_tile_fwd(const float* x, const float* image, const float* scale, const float* shift, float* y, int N, int H, int W, int Co, int s2d, int relu,
                          float* ysq_absmax, void* stream);

/* The train variant's small images: F.interpolate(imgs, (h, w), mode="bilinear", align_corners=False) (models/casmvsnet.py:60-62,148-150) fused with
 * the channels-last transpose the warp kernels want: x (N,3,H,W) planar -> y (N,h,w,3); ATen's upsample_bilinear2d arithmetic. */
int rcmvs_resize_rgb_cl(const float* x, float* y, int N, int H, int W, int h, int w, void* stream);
int rcmvs_prepare_image(const unsigned char* src, float* out, int H, int W, int h, int w, const float* mean_host,
                        const float* std_host, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RCMVS_H */
