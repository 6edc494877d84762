/* C ABI of libmvster_hip.so -- the gfx950 kernels of the MVSTER cost-volume hot path.
 *
 * The reference (JeffWang987/MVSTER) is pure Python/PyTorch and has no FFI of its own; the
 * drop-in boundary is its Python module API (models/__init__.py:2, models/MVS4Net.py:9-111),
 * mirrored by the `mvster_amd` package, which calls the entry points below through ctypes
 * with raw device pointers (mvster_amd/_lib.py).  Each entry point names the reference code it
 * replaces (file:line under the reference tree).
 *
 * Conventions: every pointer is a DEVICE pointer to contiguous fp32 unless stated otherwise;
 * buffers are caller-owned; nothing is allocated, copied to the host or synchronised inside;
 * `stream` is a hipStream_t (NULL = default stream); the return value is 0 or a negative
 * MVSTER_ERR_* code.  All functions are stateless and re-entrant.
 */
#ifndef MVSTER_HIP_H
#define MVSTER_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define MVSTER_OK 0
#define MVSTER_ERR_NULL (-1)        /* a required pointer is NULL */
#define MVSTER_ERR_SHAPE (-2)       /* inconsistent or out-of-range sizes */
#define MVSTER_ERR_UNSUPPORTED (-3) /* no kernel instance for this channel / tile combination */
#define MVSTER_ERR_LAUNCH (-4)      /* hipGetLastError() != hipSuccess after the launch */

/* proj_matrices [B,N,2,4,4] (extrinsic, intrinsic) -> rt [B,N-1,12]: rows 0..2 of
 * src_P @ inverse(ref_P) as rot(9)+trans(3), with X_P = K[:3,:3] @ E[:3,:4] (row 3 from E).
 * Replaces models/mvs4net_utils.py:1032-1035 (K@[R|t]) and :24-26 (inverse + matmul). */
int mvster_relative_projection(const float* proj_matrices, float* rt, int B, int N, void* stream);

/* Same for several cascade stages in one launch: proj_matrices = HOST array of nstage device pointers
 * (each [B,N,2,4,4]); rt [nstage,B,N-1,12]. */
int mvster_relative_projection_multi(const float* const* proj_matrices, int nstage, float* rt, int B, int N,
                                     void* stream);

/* imgs = HOST array of N device pointers, each [B,3,H,W] (the reference's `imgs` list, MVS4Net.py:60) ->
 * out [N*B,H,W,4] channels-last RGB0, view-major: the batch FPN4 runs on. */
int mvster_pack_images(const float* const* imgs, int N, float* out, int B, int H, int W, void* stream);

/* The three launches a forward starts with in one (MVS4Net.py:60-76): mvster_pack_images (imgs -> packed),
 * mvster_relative_projection_multi (proj_matrices -> rt) and mvster_init_range (depth_values [B,ndv] -> hypo [B,D,h,w], the
 * first stage's hypotheses; h*w <= H*W).  Same arithmetic as the three, bit for bit. */
int mvster_forward_prologue(const float* const* imgs, int N, float* packed, int B, int H, int W,
                            const float* const* proj_matrices, int nstage, float* rt, const float* depth_values, int ndv,
                            float* hypo, int D, int h, int w, int inverse, void* stream);

/* Fused homography warp + group-wise (or squared-difference) correlation + epipolar attention
 * aggregation over all NV source views.  Channels-last features:
 *   ref_feat [B,h,w,C] (batch stride given), src_feat view v / batch b at
 *   src_feat + v*src_view_stride + b*src_batch_stride as [Hs,Ws,C];
 *   rt [B,NV,12]; hypo [B,D,h,w]; out [B,D,h,w,G]; wsum_out optional [B,D,h,w].
 * group_cor=0 requires G == C.  variant: 0 = choose (the wave-local kernel whenever group_cor, D in {4, 8}
 * and C in {8, 16, 32, 64}); 1 = one thread per (pixel, d); 2 = workgroup-level lane split (C >= 16);
 * 3 = wave-local.  All forms return the same bits.  Replaces models/mvs4net_utils.py:13-59 (homo_warping),
 * :1037-1042 (correlation), :1048-1060 (attention aggregation). */
int mvster_warp_agg_fwd(const float* ref_feat, const float* src_feat, const float* rt, const float* hypo,
                        float* out, float* wsum_out, int B, int NV, int C, int G, int D, int h, int w, int Hs,
                        int Ws, long ref_batch_stride, long src_view_stride, long src_batch_stride, int group_cor,
                        int attn_fuse_d, float attn_temp, int variant, void* stream);

/* mvster_warp_agg_fwd with the stage's depth-hypothesis scheduling fused in: one launch instead of
 * (mvster_schedule_inverse_range | mvster_init_range) + mvster_warp_agg_fwd.  mode 1: the hypotheses are
 * schedule_inverse_range(inv_min, inv_max) of the previous stage's bounds [B,h/2,w/2] (models/mvs4net_utils.py:79-86);
 * mode 2: init_inverse_range of depth_values [B,ndv] (:71-77).  hypo_out [B,D,h,w] receives them, bit-identical to the
 * scheduler kernels' (the stage's selection and the API's `hypo_depth` read it).  Group correlation on the wave-local
 * kernel only: D in {4, 8}, (C, G) in {(8,4), (16,4), (32,8), (64,8)} -- the shipped cascade; MVSTER_ERR_UNSUPPORTED (-3)
 * otherwise: run the two launches.  Other arguments as mvster_warp_agg_fwd.  Reference call site: models/MVS4Net.py:88-99. */
int mvster_warp_agg_fwd_sched(const float* ref_feat, const float* src_feat, const float* rt, const float* inv_min,
                              const float* inv_max, const float* depth_values, int ndv, float* hypo_out, float* out,
                              float* wsum_out, int B, int NV, int C, int G, int D, int h, int w, int Hs, int Ws,
                              long ref_batch_stride, long src_view_stride, long src_batch_stride, int attn_fuse_d,
                              float attn_temp, int mode, void* stream);

/* Backward of mvster_warp_agg_fwd w.r.t. the features (the sampling grid carries no gradient,
 * models/mvs4net_utils.py:23).  grad_out/out [B,D,h,w,G], wsum [B,D,h,w] from the forward; both attention forms,
 * D <= 16.  grad_ref [B,h,w,C] is written; grad_src [NV][B,Hs,Ws,C] must be zero-initialised.
 * windows / win_org: scratch of the sizes mvster_warp_agg_bwd_scratch() reports (floats / ints, uninitialised).
 * win_org is required (it also holds the operand maxima of the fixed-point scale of the LDS accumulators).  With
 * `windows` the source gradient is accumulated without floating-point atomics for every tap inside a workgroup's
 * scatter window (integer LDS accumulation, dense windows, a gather pass in fixed order: bit-reproducible); with
 * windows == NULL the windows are flushed with global fp32 atomics.  Taps outside a window (strongly rotated views)
 * always use global atomics.
 * Autograd of models/mvs4net_utils.py:1036-1060. */
int mvster_warp_agg_bwd(const float* ref_feat, const float* src_feat, const float* rt, const float* hypo,
                        const float* out, const float* wsum, const float* grad_out, float* grad_ref,
                        float* grad_src, float* windows, int* win_org, int B, int NV, int C, int G, int D, int h, int w,
                        int Hs, int Ws, long ref_batch_stride, long src_view_stride, long src_batch_stride, int group_cor,
                        int attn_fuse_d, float attn_temp, void* stream);
int mvster_warp_agg_bwd_scratch(int B, int NV, int C, int G, int D, int h, int w, int attn_fuse_d, long* window_floats,
                                long* origin_ints);

/* mvster_warp_agg_bwd with the source-view gradient accumulated by a SORTED SCATTER instead of scatter windows + global
 * atomics (the adjoint of F.grid_sample in homo_warping, models/mvs4net_utils.py:13-59): the (pixel, hypothesis, view)
 * samples are counting-sorted by 32x32 source tile (count, prefix sums, 48-byte records appended per touched tile), then one
 * workgroup per (batch, view, 8-channel block, tile) accumulates its records in a 64-bit fixed-point LDS window and writes
 * the tile with plain stores.  Bit-reproducible, independent of the smoothness of the depth maps; grad_src needs NO zero
 * fill (every texel is written exactly once).  Same arguments as mvster_warp_agg_bwd except the scratch: rec (*rec_floats
 * floats: the worst case of 4 records per sample and channel block) and ints (*ints ints) as sized by the _scratch query.
 * MVSTER_ERR_UNSUPPORTED (from both) for source maps of more than 2048 tiles or 8190 texels a side: use mvster_warp_agg_bwd. */
int mvster_warp_agg_bwd_sorted_scratch(int B, int NV, int C, int D, int h, int w, int Hs, int Ws, long* rec_floats, long* ints);
int mvster_warp_agg_bwd_sorted(const float* ref_feat, const float* src_feat, const float* rt, const float* hypo,
                               const float* out, const float* wsum, const float* grad_out, float* grad_ref, float* grad_src,
                               float* rec, int* ints, int B, int NV, int C, int G, int D, int h, int w, int Hs, int Ws,
                               long ref_batch_stride, long src_view_stride, long src_batch_stride, int group_cor,
                               int attn_fuse_d, float attn_temp, void* stream);

/* depth_values [B,ndv] (columns 0 and ndv-1 used) -> out [B,D,h,w].
 * inverse=1: models/mvs4net_utils.py:71-77 (init_inverse_range); inverse=0: :61-69 (init_range). */
int mvster_init_range(const float* depth_values, int ndv, float* out, int B, int D, int h, int w, int inverse,
                      void* stream);

/* inv_min, inv_max [B,h/2,w/2] -> out [B,D,h,w].  models/mvs4net_utils.py:79-86. */
int mvster_schedule_inverse_range(const float* inv_min, const float* inv_max, float* out, int B, int D, int h,
                                  int w, void* stream);

/* cur_depth [B,h/2,w/2], interval [B] (device) -> out [B,D,h,w].  models/mvs4net_utils.py:88-99. */
int mvster_schedule_range(const float* cur_depth, const float* interval, float* out, int B, int D, int h, int w,
                          void* stream);

/* (optional 1x1x1 `prob` head) + softmax over D + first-max argmax + gather + max-prob confidence
 * + inverse-depth bounds.  Either logits [B,D,h,w] or feat [B,D,h,w,CF] (+ prob_w [CF], prob_b [1]).
 * attn [B,D,h,w], depth/conf/inv_min/inv_max [B,h,w] (conf, inv_* optional; inv_* need D >= 3),
 * logits_out optional [B,D,h,w].  models/mvs4net_utils.py:900 and :1068-1088. */
int mvster_select_depth(const float* logits, const float* feat, const float* prob_w, const float* prob_b, int CF,
                        const float* hypo, float* attn, float* depth, float* conf, float* inv_min, float* inv_max,
                        float* logits_out, int B, int D, int h, int w, float split_itv, void* stream);

/* Backward of mvster_select_depth's differentiable part (training): the 1x1x1 `prob` head + softmax over depth with respect
 * to the feature volume and the head's parameters, given gattn = d L / d attn [B,D,h,w] (depth, confidence and the
 * inverse bounds carry no gradient: argmax / detached in the reference, models/mvs4net_utils.py:1068-1088).
 * dfeat [B,D,h,w,CF] = dlogit * prob_w; partial [B * ceil(h*w/256), CF + 1] = per-workgroup sums of dlogit * feat (d prob_w)
 * and of dlogit (d prob_b): the caller adds the rows.  CF = 8, D <= 16. */
int mvster_select_depth_bwd(const float* attn, const float* gattn, const float* feat, const float* prob_w, float* dfeat,
                            float* partial, int B, int D, int h, int w, int CF, void* stream);

/* in [B,hi,wi] -> out [B,ho,wo], bilinear, align_corners=True.  models/mvs4net_utils.py:1077. */
int mvster_upsample_bilinear(const float* in, float* out, int B, int hi, int wi, int ho, int wo, void* stream);

/* n <= 8 maps of one batch to one output size in one launch: ins / outs / his / wis = HOST arrays (device pointers,
 * input sizes); ins[k] [B,his[k],wis[k]] -> outs[k] [B,ho,wo].  The coarse stages' confidence maps (MVS4Net.py:1077 per stage). */
int mvster_upsample_bilinear_multi(const float* const* ins, float* const* outs, const int* his, const int* wis, int n, int B,
                                   int ho, int wo, void* stream);

/* Channels-last implicit-GEMM convolution on the fp32 matrix cores with a fused
 * scale/shift (+ReLU, +skip) epilogue.  in [B,Di,Hi,Wi,cin]; wpk = weights packed by
 * mvster_amd/conv_plan.py; scale/shift [16*ntiles]; skip optional; zeros = >=16 B of zeros;
 * geom = HOST int32 array (layout: conv_plan.GEOM), woff = HOST int64 per-class weight offsets.
 * prob_w [8] / prob_b [1] (optional, cout == 8, variants 0/2): fuse the 1x1x1 `prob` head of reg2d
 * (models/mvs4net_utils.py:900) -- `out` is then the [B,Do,Ho,Wo] logits volume instead of the feature volume.
 * variant 0 = direct (operands from L1), 1 = LDS-staged input patch (ordinary convs, cin % 16 == 0,
 * mt in {2,4}: the workgroup tile is 2*mt rows x 32 columns), 2 = direct with the 4 waves of a workgroup
 * splitting K (small deep layers; cin >= 16, mt*nt <= 4), 5 = persistent workgroups with LDS-DMA double-buffered
 * input patches and workgroup-resident weights (ordinary convs, cin in {16, 32, 64}, cout % 16 == 0, kernels
 * (1|3)x3x3 and 1x5x5, in-plane stride 1 or 2, mt = 2; bits 8.. of `variant` = workgroups per CU, 0 = default),
 * 6 = persistent 1x1x1 kernel with all packed weights resident in LDS and the input read once (cin in {32, 64},
 * cout % 4 == 0, optional same-shape skip; nt is ignored: a wave walks all N tiles),
 * 7 = variant 5 with eight waves per workgroup in ping-pong (3x3, cin 16 / 32; measured no faster, kept for the record),
 * 8 = Winograd F(2x2, 3x3) on the persistent LDS-DMA frame: 1x3x3 stride 1 pad 1, cin in {16, 32}, cout % 16 == 0,
 * optional same-shape skip; `wpk` must be the array written by mvster_pack_wino_weights (not bit-identical to the other
 * variants: fp32 with a different operation order),
 * 9 = the same transform for deep layers: (1|3)x3x3 stride 1, cin in {16, 32, 64}; patch slices and transformed weights
 * stream through LDS rings one depth tap and 16-channel chunk at a time.
 * Conv3d/ConvTranspose3d/BatchNorm3d/ReLU of reg2d/reg3d (models/mvs4net_utils.py:870-965) and
 * Conv2d/BatchNorm2d/ReLU/upsample-add of FPN4 (:419-502). */
int mvster_conv_mfma(const float* in, const float* wpk, const float* scale, const float* shift, const float* skip,
                     const float* zeros, const float* prob_w, const float* prob_b, float* out, const int* geom,
                     int ngeom, const long* woff, int cin, int mt, int nt, int variant, void* stream);

/* 3x3 (pad 1, stride 1) convolution with 8 output channels and cin in {4, 8} on the fp32 VALU (packed FMA),
 * for the narrow full-resolution layers where the 16-wide MFMA tile is half padding.  in [NB,H,W,cin],
 * w [3,3,cin,8], scale/shift [8], skip optional [NB,H,W,8], out [NB,H,W,8] (depth slices folded into NB).
 * FPN4 conv0 (models/mvs4net_utils.py:427-428), reg2d conv0 (:875), the composed out4 tail (:459). */
int mvster_conv_small(const float* in, const float* w, const float* scale, const float* shift, const float* skip,
                      float* out, int NB, int H, int W, int cin, int relu, void* stream);

/* The same layers (same arguments, same result up to fp32 summation order) on the fp32 matrix cores: the empty half of
 * the 16-wide N tile holds the NEXT pixel's outputs (weights shifted by one tap column over a four-column K axis: 24
 * instead of 36 MFMAs per 32 pixels), persistent workgroups, inputs and the skip tile streamed through an LDS-DMA ring
 * three tiles ahead (csrc/conv_narrow.hip).  mt: tile rows / 4 (2, 4; 0 = by size); wpc: workgroups per CU (0 = default). */
int mvster_conv_narrow(const float* in, const float* w, const float* scale, const float* shift, const float* skip,
                       float* out, int NB, int H, int W, int cin, int relu, int mt, int wpc, void* stream);

/* The 8 -> 4 form of the same kernel (round 6): in [NB,H,W,8], w [3,3,8,8] with output columns 4..7 zero, scale / shift [8]
 * -> out [NB,H,W,4]; no skip.  The input gradient of reg2d's conv0 in training (autograd of models/mvs4net_utils.py:875). */
int mvster_conv_narrow4(const float* in, const float* w, const float* scale, const float* shift, float* out, int NB, int H,
                        int W, int relu, int mt, int wpc, void* stream);

/* FPN4.conv0 in ONE launch (round 6; models/mvs4net_utils.py:427-428: 3 -> 8 -> 8 channels, each conv3x3 + BatchNorm + ReLU):
 * in [NB,H,W,4] (RGB0), w1 [3,3,4,8], w2 [3,3,8,8], scale / shift [8] each -> out [NB,H,W,8].  The 8-channel intermediate
 * stays in LDS (14 x 64-pixel tiles, the first layer computed on the 16 x 66 halo tile, zero outside the image), so the pair
 * moves 78 MB instead of 183 MB at 5 x 512 x 640.  Same result as two mvster_conv_narrow calls up to fp32 summation order.
 * wpc: workgroups per CU (0 = default).  csrc/conv_narrow.hip. */
int mvster_conv_narrow_pair(const float* in, const float* w1, const float* scale1, const float* shift1, const float* w2,
                            const float* scale2, const float* shift2, float* out, int NB, int H, int W, int relu1, int relu2,
                            int wpc, void* stream);

/* ConvTranspose3d (1,3,3), stride (1,2,2), padding (0,1,1), output_padding (0,1,1) + BatchNorm scale/shift +
 * ReLU + skip add for (cin, cout) in {(16,8), (32,16)} on the VALU (the layers are HBM-bound).  in [NB,Hi,Wi,cin],
 * w [3,3,cin,cout], skip optional [NB,2Hi,2Wi,cout]; prob_w/prob_b optional (cout == 8): fuse the 1x1x1 head,
 * out = logits [NB,2Hi,2Wi] instead of [NB,2Hi,2Wi,cout].  reg2d conv9 / conv11 (models/mvs4net_utils.py:890-900). */
int mvster_deconv_small(const float* in, const float* w, const float* scale, const float* shift, const float* skip,
                        const float* prob_w, const float* prob_b, float* out, int NB, int Hi, int Wi, int cin,
                        int cout, int relu, void* stream);

/* reg2d's last layer and the depth selection in one launch (models/mvs4net_utils.py:897-900 and :1068-1088):
 * mvster_deconv_small (ConvTranspose 16 -> 8 + BatchNorm + ReLU + skip, fused 1x1x1 `prob` head) followed by
 * mvster_select_depth, bit for bit, without the logits volume going through HBM.  in [B*D,Hi,Wi,16] (slice b*D + d),
 * skip [B*D,2Hi,2Wi,8] or null, hypo [B,D,2Hi,2Wi] -> attn [B,D,2Hi,2Wi], depth / conf / inv_min / inv_max [B,2Hi,2Wi]
 * (conf and inv_* optional; inv_* need D >= 3), logits_out [B,D,2Hi,2Wi] optional.  2 <= D <= 16. */
int mvster_deconv_select(const float* in, const float* w, const float* scale, const float* shift, const float* skip,
                         const float* prob_w, const float* prob_b, const float* hypo, float* attn, float* depth, float* conf,
                         float* inv_min, float* inv_max, float* logits_out, int B, int D, int Hi, int Wi, int cin, int relu,
                         float split_itv, void* stream);

/* FPN4 top-down tail, re-associated: G [NB,H/2,W/2,9*CO] = 1x1 conv of the half-resolution top-down
 * map with the 9 taps of the output conv stacked on the channel axis; vb [9,CO] = the taps applied to the
 * lateral conv's bias; P [NB,H,W,CO] = sum over in-bounds taps of (bilinear x2 align_corners upsample of
 * G_tap at p+tap, + vb[tap]).  Together with a 3x3 conv of the lateral input (composed weights, skip-add P)
 * this equals out4(F.interpolate(f) + inner3(c0)) of models/mvs4net_utils.py:488-489 without the
 * full-resolution 64-channel intermediate.  workspace: optional [NB,H,W/2,3*CO]; when given the gather runs as
 * two separable passes (vertical into the workspace, then horizontal), otherwise as one 36-tap pass. */
int mvster_fpn_tail_gather(const float* G, const float* vb, float* P, float* workspace, int NB, int H, int W,
                           int CO, void* stream);

/* Adjoint of mvster_fpn_tail_gather with respect to G (training): gP [NB,H,W,CO] -> gG [NB,H/2,W/2,pitch], channels
 * 9*CO..pitch-1 zero-filled (pitch >= 9*CO, multiple of 4); a gather over the <= 6x6 full-resolution positions that read
 * a half-resolution pixel, no atomics.  CO in {8, 16}. */
int mvster_fpn_tail_gather_bwd(const float* gP, float* gG, int NB, int H, int W, int CO, int pitch, void* stream);

/* Lateral 1x1 conv + top-down add of one FPN level, for a top-down map that is only held at the coarser
 * resolution (models/mvs4net_utils.py:485, after pushing the next level's 1x1 "tap" conv through it):
 * out [NB,H,W,CO] = bias [CO] + A [CO,CI] x [NB,H,W,CI] + bilinear x2 align_corners upsample of q [NB,H/2,W/2,CO].
 * (CI, CO) in {(16,72), (8,72)}. */
int mvster_fpn_lateral_up(const float* x, const float* A, const float* bias, const float* q, float* out, int NB,
                          int H, int W, int CI, int CO, void* stream);

/* mvster_fpn_lateral_up (16 -> 72) + mvster_fpn_tail_gather (CO = 8) of the finest FPN level in ONE launch: a workgroup builds
 * the half-resolution 72-channel patch its 8 x 32 output tile gathers from straight into LDS (lateral product on MFMA tiles,
 * bilinear x2 of q in the epilogue), so the 118 MB map between the two kernels never touches HBM.  x [NB,H/2,W/2,16],
 * A [72,16], bias [72], q [NB,H/4,W/4,72], vb [9,8] -> P [NB,H,W,8].  H, W multiples of 4, H >= 16, W >= 64, CI = 16;
 * MVSTER_ERR_UNSUPPORTED otherwise.  models/mvs4net_utils.py:485-489 (section 4.3 of DESIGN.md). */
int mvster_fpn_tail_fused(const float* x, const float* A, const float* bias, const float* q, const float* vb, float* P,
                          int NB, int H, int W, int CI, void* stream);

/* Packed-weight refresh on the device (training, once per layer and optimizer step): writes the fragment order
 * [K/16][N/16][64][4] that mvster_conv_mfma reads, Bm[tap*cin_pad + ci][n] = w[n*s_n + ci*s_c + kz*s_z + ky*s_y + kx*s_x]
 * (element strides of the parameter tensor; flip = 1 mirrors the taps: the input-gradient form of a stride-1 layer),
 * zero for ci >= cin and n >= cout.  wpk holds ceil(kd*kh*kw*cin_pad/16) * ceil(cout/16) * 256 floats. */
int mvster_pack_conv_weights(const float* w, float* wpk, int cout, int cin, int cin_pad, int kd, int kh, int kw, long s_n,
                             long s_c, long s_z, long s_y, long s_x, int flip, void* stream);

/* Transformed weights of the Winograd F(2x2, 3x3) kernels (variants 8 / 9 of mvster_conv_mfma, which take this array as
 * their `wpk`): U = G g G^T per (cout, cin, kz) of w [cout, cin, kd, 3, 3], kd in {1, 3} (element strides; flip = 1 mirrors
 * the taps), stored in the same fragment order with the 16 transform points in place of the in-plane taps:
 * kd * 16 * cin_pad/16 K steps x ceil(cout/16) x 256 floats.  Stands in for the weight side of nn.Conv2d(3x3) (models/mvs4net_utils.py:116-123). */
int mvster_pack_wino_weights(const float* w, float* wpk, int cout, int cin, int cin_pad, int kd, long s_n, long s_c, long s_z,
                             long s_y, long s_x, int flip, void* stream);

/* The same for every record of a DEVICE table in ONE launch (the training step re-transforms the weights of all its
 * Winograd layers after each optimizer update): 88-byte records {const float* w; float* wpk; long s_n, s_c, s_z, s_y, s_x;
 * int cout, cin_raw, cin_pad, kd, flip, ntile, first_block, pad}, ntile = ceil(cout / 16), first_block = prefix sum of
 * ceil(kd*16*cin_pad*ntile*16 / 256) over the earlier records, total_blocks = their sum. */
int mvster_pack_wino_batch(const void* descs, int ndesc, int total_blocks, void* stream);

/* The same refresh for a transposed layer (output-parity classes): w [cin, cout, kd, kh, kw] contiguous, ktot = kd*kh*kw
 * <= 27; class c packs the taps taps[c*27 .. c*27 + ntaps[c]) (flattened indices, input-offset order) at float offset
 * woff[c] of wpk. */
int mvster_pack_conv_weights_classes(const float* w, float* wpk, int cout, int cin, int cin_pad, int ktot, int nclass,
                                     const int* ntaps, const long* woff, const int* taps, void* stream);

/* Weight gradient of a channels-last convolution (training): for every kernel tap
 *   dW[tap][co][ci] = sum_o gy[o][co] * x[o*s - p + tap][ci]      (zero padding)
 * x [B,Di,Hi,Wi,CI], gy [B,Do,Ho,Wo,CO] with (Do,Ho,Wo) the conv output size for (k,s,p); CI, CO <= 64.
 * Workgroup slot n of `partial` [nblk][kd*kh*kw][COP][CIP] (COP/CIP = CO/CI rounded up to 16, 48 -> 64)
 * receives the sum over the output rows that workgroup visited; the caller adds the nblk slots.  packed = 1 (CI <= 8):
 * 16/CIP taps share one N tile (CIP = 4 or 8), partial [nblk][ceil(taps/(16/CIP))][COP][16], column = (tap % TPN)*CIP + ci.
 * With x and gy swapped it is the weight gradient of the transposed convolution.  Replaces autograd's conv weight gradients
 * of nn.Conv3d / nn.Conv2d / nn.ConvTranspose3d (models/mvs4net_utils.py:116-123, :224-251, :870-965, :419-502). */
int mvster_conv_wgrad(const float* x, const float* gy, float* partial, int nblk, int B, int Di, int Hi, int Wi, int CI,
                      int Do, int Ho, int Wo, int CO, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph,
                      int pw, int packed, void* stream);

/* Slot count (`nblk`) to give `partial` for a layer: > 0 where the persistent weight-gradient kernel applies (3x3 / 3x3x3,
 * stride 1, 16 / 32 / 64 channels on both sides: it fills exactly its workgroup count of slots) and for the 5x5 stride-2
 * layers (one resident round of workgroups), 0 = caller's choice.  The caller may use fewer slots, never more. */
int mvster_conv_wgrad_slots(int CI, int CO, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw, int packed);

/* Finish of the weight gradient: adds the nblk slots (fixed order) and writes dW in the parameter's layout in one launch.
 * Slot element (g, row, col) of [ngrp][cop][width]: co = row; cip = 0: tap = g, ci = col; cip = 4 / 8 (packed): tap =
 * g*(16/cip) + col/cip, ci = col % cip.  Kept when tap < ntaps, co < co_lim, ci < ci_lim; flip mirrors the taps
 * (tap -> ntaps-1-tap); dw is [co_lim][ci_lim][ntaps], or [ci_lim][co_lim][ntaps] with swap (the mirrored narrow-output
 * form and nothing else needs both). */
int mvster_conv_wgrad_finish(const float* partial, float* dw, int nblk, int ngrp, int cop, int width, int ntaps, int cip,
                             int co_lim, int ci_lim, int swap, int flip, void* stream);

/* `count` finishes in ceil(count / 56) launches: recs = HOST array of 56-byte records {const float* partial; float* dw; int
 * nblk, ngrp, cop, width, ntaps, cip, co_lim, ci_lim, swap, flip} -- the arguments of mvster_conv_wgrad_finish.  The
 * training step defers the finishes of all its layers to the end of the backward pass (nothing reads a weight gradient
 * before) and issues them together. */
int mvster_conv_wgrad_finish_batch(const void* recs, int count, void* stream);

/* Training-mode BatchNorm + ReLU on channels-last activations (C a power of two, 4..64): the elementwise half of the
 * reference's conv -> BatchNorm -> ReLU blocks (models/mvs4net_utils.py:116-123, :224-251) and its autograd.
 * x [groups*rows, C]: `groups` independent statistics groups of `rows` rows each (the reference normalises every
 * view's batch on its own, MVS4Net.py:65-68); scale = gamma * rstd, shift = beta - mean * scale, mean, rstd are
 * [groups, C] (batch statistics from the caller).
 *   stats:       per-workgroup slots partial[g][n][0][c] / [g][n][1][c] = sum d and sum d^2, d = x - x[first row of g];
 *                a finishing kernel adds the slots in order (fp64) and writes out [5][groups][C] = mean, biased var, rstd,
 *                scale, shift; running_mean / running_var (optional) get the groups' exponential-average updates in
 *                order (momentum, unbiased variance) and num_batches_tracked (optional, device int64) += groups --
 *                what `groups` sequential nn.BatchNorm calls would do
 *   fwd:         y = relu(x*scale + shift) (+ skip)                      (relu = 0: affine only; skip optional: the U-Net's
 *                same-shape skip connection, added after the activation, models/mvs4net_utils.py:893-895)
 *   bwd_reduce:  slots as above of sum g_ and sum g_*xh; the finishing kernel writes sums [groups][2][C] and the parameter
 *                gradients dbeta [C] = sum_g sum g_, dgamma [C] = sum_g sum g_*xh
 *   bwd_apply:   dx = scale * (g_ - sums[g][0]/rows - xh * sums[g][1]/rows),   g_ = gy * (y > 0), xh = (x - mean) * rstd;
 *                frozen = 1 (statistics are constants): dx = scale * g_
 * partial: [groups][mvster_bn_slots(rows, C, groups)][2][C] floats of scratch.  stats and bwd_reduce are ONE launch each:
 * the last workgroup to arrive (ticket: one device int, 0 before the call and 0 again after it) adds the slots in a fixed
 * order in fp64 and writes the results (slots published with write-through stores: no L2 write-back); groups*2C <= 2048. */
int mvster_bn_relu_fwd(const float* x, const float* scale, const float* shift, const float* skip, float* y, long rows,
                       int C, int relu, int groups, void* stream);
int mvster_bn_slots(long rows, int C, int groups);
int mvster_bn_stats(const float* x, const float* weight, const float* bias, float* running_mean, float* running_var,
                    long* num_batches_tracked, float* partial, float* out, int* ticket, long rows, int C, int groups,
                    float eps, float momentum, void* stream);
int mvster_bn_relu_bwd_reduce(const float* x, const float* gy, const float* scale, const float* shift, const float* mean,
                              const float* rstd, float* partial, float* sums, float* dgamma, float* dbeta, int* ticket,
                              long rows, int C, int relu, int groups, void* stream);
int mvster_bn_relu_bwd_apply(const float* x, const float* gy, const float* scale, const float* shift, const float* mean,
                             const float* rstd, const float* sums, float* dx, long rows, int C, int relu, int groups,
                             int frozen, void* stream);

/* out [C] = column sums of the channels-last x [rows, C], C in {4, 8, 16, 32, 64}: the bias gradient of the reference's
 * convolutions with bias (sum of the output gradient over every voxel; FPN laterals models/mvs4net_utils.py:485-487,
 * monocular heads :846-848).  partial: mvster_bn_slots(rows, C, 1) * 2 * C floats, ticket as above.  One launch. */
int mvster_col_sum(const float* x, float* partial, float* out, int* ticket, long rows, int C, void* stream);

/* Training-mode BatchNorm as the step runs it since round 6: TWO launches per pass without a serial tail -- the reduction
 * kernel only writes its per-workgroup slots, the apply kernel's workgroups sum them in their prologue (every workgroup its
 * group's, in the same fixed order, fp64), one extra workgroup does the running-average updates (forward) / the parameter
 * gradients (backward).  train_fwd = mvster_bn_stats + mvster_bn_relu_fwd (same `out` pack [5][groups][C], same running
 * updates); train_bwd = mvster_bn_relu_bwd_reduce + _apply (pack = the forward's `out`).  partial: groups *
 * mvster_bn_train_slots(rows, C, groups) * 2 * C floats.  models/mvs4net_utils.py:116-123, :224-251 under autograd. */
int mvster_bn_train_slots(long rows, int C, int groups);
int mvster_bn_train_fwd(const float* x, const float* weight, const float* bias, float* running_mean, float* running_var,
                        long* num_batches_tracked, const float* skip, float* partial, float* y, float* out, long rows, int C,
                        int relu, int groups, float eps, float momentum, void* stream);
int mvster_bn_train_bwd(const float* x, const float* gy, const float* pack, float* partial, float* dgamma, float* dbeta, float* dx,
                        long rows, int C, int relu, int groups, void* stream);

/* The same BatchNorm passes for SMALL tensors in one launch each way (most layers of the step: coarse stages, deep U-Net
 * levels): a thread keeps its <= 8 (backward: 4) float4 of x (and gy) in registers across the reduction, the last workgroup
 * to arrive finishes and publishes the statistics and releases the others, which spin on a flag -- a grid barrier among at
 * most 128 resident workgroups.  fwd_fused = mvster_bn_stats + mvster_bn_relu_fwd (same `out`, same running updates);
 * bwd_fused = mvster_bn_relu_bwd_reduce + _apply (pack = the forward's `out`).  partial: groups * 128 * 2 * C floats; sync: 3
 * device ints, zero before and after.  mvster_bn_fused_ok(rows, C, groups, backward) says whether a tensor fits (16 / 8 MB). */
int mvster_bn_fused_ok(long rows, int C, int groups, int backward);
int mvster_bn_fwd_fused(const float* x, const float* skip, float* y, const float* weight, const float* bias, float* running_mean,
                        float* running_var, long* num_batches_tracked, float* partial, float* out, int* sync, long rows, int C,
                        int relu, int groups, float eps, float momentum, void* stream);
int mvster_bn_bwd_fused(const float* x, const float* gy, const float* pack, float* partial, float* sums, float* dgamma,
                        float* dbeta, float* dx, int* sync, long rows, int C, int relu, int groups, void* stream);

/* Bilinear x2 up-sampling (align_corners=True) of a channels-last map, in [B,h,w,C] -> out [B,2h,2w,C], and its
 * adjoint gout [B,2h,2w,C] -> gin [B,h,w,C] as a gather (no atomics): the FPN top-down path in training
 * (models/mvs4net_utils.py:488-496 under autograd).  C % 4 == 0. */
int mvster_upsample2x_cl_fwd(const float* in, float* out, int B, int h, int w, int C, void* stream);
int mvster_upsample2x_cl_bwd(const float* gout, float* gin, int B, int h, int w, int C, void* stream);
/* Nearest x2 (F.interpolate(scale_factor=2, mode="nearest"), the mono head's up-sampling, models/mvs4net_utils.py:858):
 * backward = 0: in [B,h,w,C] -> out [B,2h,2w,C]; backward = 1: the adjoint, in = gout [B,2h,2w,C] -> out = gin [B,h,w,C]. */
int mvster_upsample2x_nearest_cl(const float* in, float* out, int B, int h, int w, int C, int backward, void* stream);

/* Sinkhorn optimal-transport loss per pixel and its gradient, fused (discrete form, ot_continous=False):
 * attn, hypo [B,D,HW], gt [B,HW] -> loss_pix [B,HW], jac [B,D,HW] = d loss_pix / d attn.  2 <= D <= 16,
 * iters <= 16.  Replaces the per-pixel part of `sinkhorn` (models/mvs4net_utils.py:1096-1142) and its autograd;
 * the masked mean over pixels stays with the caller. */
int mvster_sinkhorn(const float* attn, const float* hypo, const float* gt, float* loss_pix, float* jac, int B, int D,
                    long HW, int iters, float eps, void* stream);
/* The continuous form (ot_continous=True, models/mvs4net_utils.py:1111-1123): D + 1 target columns, all the mass on
 * the extra one, whose cost is the distance of every bin to the ground truth's fractional bin position (10 where
 * mask <= 0.5).  mask [B,HW] float; 3 <= D <= 8, iters <= 16. */
int mvster_sinkhorn_continuous(const float* attn, const float* hypo, const float* gt, const float* mask, float* loss_pix,
                               float* jac, int B, int D, long HW, int iters, float eps, void* stream);

/* The per-pixel terms of one stage of the training loss around the OT term, one pass (MVS4net_loss,
 * models/MVS4Net.py:131-151: the mask compare, F.l1_loss(mono_depth[mask], depth_gt[mask]) :136-137, mask_out_of_range
 * :141-147 and the masked mean of the OT loss).  hypo [B,D,HW] (D >= 3), gt, mask (float, > 0.5 = valid), loss_pix
 * [B,HW] (from mvster_sinkhorn*), mono [B,HW] or NULL -> terms [5][B*HW]: valid, valid*|mono-gt|, valid*(no hypothesis
 * within |t(hypo_2)-t(hypo_1)| of gt; t = 1/x when inverse_depth), valid*loss_pix, valid*sign(mono-gt).  The caller
 * sums the planes: l1 = S1/S0, range ratio = S2/S0, ot = S3/S0. */
int mvster_stage_loss_terms(const float* hypo, const float* gt, const float* mask, const float* loss_pix,
                            const float* mono, float* terms, int B, int D, long HW, int inverse_depth, void* stream);

/* Geometric-consistency filter of one reference view against NS source views, fused (test_mvs4.py:273-328 per
 * view pair + the sums of filter_depth :362-385).  depth_ref [H,W], depth_src [NS,H,W]; ref_mats = inv(K_ref)[9],
 * K_ref[9]; view_mats [NS][42] = (E_src inv(E_ref))[3x4], K_src[3x3], inv(K_src)[3x3], (E_ref inv(E_src))[3x4], all
 * row major, float32 values widened to double (the reference's dtype flow).  mask_sum [H,W] = number of consistent
 * views, depth_sum [H,W] = sum of their reprojected depths; view_mask / view_depth / x_src / y_src [NS,H,W] are
 * optional per-view outputs (the returns of check_geometric_consistency).  Source depth is sampled like
 * cv2.remap(INTER_LINEAR): 1/32-pixel coordinates, constant-0 border. */
int mvster_geo_filter(const float* depth_ref, const float* depth_src, const double* ref_mats, const double* view_mats,
                      int* mask_sum, float* depth_sum, unsigned char* view_mask, float* view_depth, float* x_src,
                      float* y_src, int NS, int H, int W, float pix_thres, float rel_thres, void* stream);

/* ---- training-step glue (csrc/train_glue.hip): the reference's chains of small tensor expressions, one launch each ---- */

/* One stage of MVS4net_loss around the OT term (models/MVS4Net.py:126-153), forward: hypo [B,D,HW] (D >= 3), gt, mask
 * (float, > 0.5 = valid), loss_pix [B,HW] (from mvster_sinkhorn*), mono [B,HW] or NULL, total_in: device scalar or NULL ->
 * planes [2][B*HW] (valid, valid*sign(mono-gt): for the backward), partial [mvster_stage_loss_slots(B*HW)][4] (scratch),
 * out [6] = #valid, l1 = mean|mono-gt| (:136-137), out-of-range ratio (:141-147), ot = masked mean of loss_pix,
 * weighted = w_stage*(w_l1*l1 + w_ot*ot) (:151), total = total_in + weighted.  Two launches, deterministic. */
int mvster_stage_loss_slots(long n);
int mvster_stage_loss_fwd(const float* hypo, const float* gt, const float* mask, const float* loss_pix, const float* mono,
                          const float* total_in, float* planes, float* partial, float* out, int B, int D, long HW,
                          int inverse_depth, float w_l1, float w_ot, float w_stage, void* stream);

/* Its backward: jac [B,D,HW] = d loss_pix / d attn (mvster_sinkhorn*), planes / out from the forward; g_total / g_l1 /
 * g_ot = device scalars (NULL = 0), gradients of out[5] / out[1] / out[3]; w_l1 = w_stage*l1ot_lw[0], w_ot =
 * w_stage*l1ot_lw[1] -> g_attn [B,D,HW] (or NULL), g_mono [B,HW] (or NULL).  One launch. */
int mvster_stage_loss_bwd(const float* jac, const float* planes, const float* out, const float* g_total, const float* g_l1,
                          const float* g_ot, float w_l1, float w_ot, float* g_attn, float* g_mono, int B, int D, long HW,
                          void* stream);

/* Monocular head, disparity -> depth (models/mvs4net_utils.py:858-866): depth [B,HW] = 1 / (1/d_max[b] + (1/d_min[b] -
 * 1/d_max[b]) * sigmoid(z)); sig [B,HW] = sigmoid(z) is kept for the backward, gz = g * d depth / d z. */
int mvster_mono_depth_fwd(const float* z, const float* dmin, const float* dmax, float* depth, float* sig, int B, long HW,
                          void* stream);
int mvster_mono_depth_bwd(const float* g, const float* depth, const float* sig, const float* dmin, const float* dmax, float* gz,
                          int B, long HW, void* stream);

/* out [NB,H,W,Ca+Cb] = concat(nearest x2 up-sampling of a [NB,H/2,W/2,Ca], b [NB,H,W,Cb]) -- the input of the monocular
 * head's 3x3 convolutions (models/mvs4net_utils.py:854-857) -- and the adjoint g -> ga, gb.  H, W even; Ca, Cb % 4 == 0. */
int mvster_upcat_fwd(const float* a, const float* b, float* out, int NB, int H, int W, int Ca, int Cb, void* stream);
int mvster_upcat_bwd(const float* g, float* ga, float* gb, int NB, int H, int W, int Ca, int Cb, void* stream);

/* Composed weights of the re-associated finest FPN level (out4(up(f) + inner3(c0)), models/mvs4net_utils.py:496-498):
 * wo [CO,CM,3,3] = out4.weight, wi [CM,CI] = inner3.weight, bi [CM] = inner3.bias -> wg [9*CO,CM] (row tap*CO+o =
 * wo[o,:,tap]), wc [CO,CI,3,3] = sum_c wo[o,c,tap] wi[c,i], vb [9,CO] = sum_c wo[o,c,tap] bi[c]; and the adjoint
 * (g_wg / g_wc / g_vb may be NULL) -> g_wo, g_wi, g_bi.  One single-workgroup launch each. */
int mvster_fine_weights_fwd(const float* wo, const float* wi, const float* bi, float* wg, float* wc, float* vb, int CO, int CM,
                            int CI, void* stream);
int mvster_fine_weights_bwd(const float* wo, const float* wi, const float* bi, const float* g_wg, const float* g_wc,
                            const float* g_vb, float* g_wo, float* g_wi, float* g_bi, int CO, int CM, int CI, void* stream);

/* Adam update (torch.optim.Adam semantics with L2 weight decay, no amsgrad; train_mvs4.py:367) of `count` fp32 tensors:
 * params / grads = HOST arrays of `count` device pointers, sizes / state_offs = host int arrays (elements; offsets of a
 * tensor's moments in the flat exp_avg / exp_avg_sq buffers).  step_cells [2] device floats: cell 0 = number of updates
 * done so far, + 1 afterwards (cell 1 scratch); lr = one DEVICE float (a schedule reaches a captured step by rewriting
 * it between replays).  ceil(count / 128) launches (+ 1 when that is odd); the pointers travel as
 * kernel arguments, so a captured step records them with the launch. */
int mvster_fused_adam(const void* const* params, const void* const* grads, const int* sizes, const int* state_offs, int count,
                      float* exp_avg, float* exp_avg_sq, float* step_cells, const float* lr, double beta1, double beta2,
                      double eps, double weight_decay, void* stream);

/* Batched gather, one launch: for every record r of the DEVICE table `descs` (32-byte records {const float* src; float* dst;
 * const int* idx; int n; int first_block}), dst[i] = idx[i] > 0 ? src[idx[i] - 1] : 0, i < n (n % 4 == 0); first_block =
 * prefix sum of ceil(n / 1024), total_blocks their sum.  Refreshes all packed / permuted weight forms of the training step
 * from their parameters after an optimizer update (host-side plumbing: the reference's layers read the parameters directly). */
int mvster_gather_batch(const void* descs, int ndesc, int total_blocks, void* stream);

/* Name of the kernel the most recent mvster_conv_mfma / mvster_conv_small / mvster_deconv_small / mvster_conv_wgrad /
 * mvster_warp_agg_fwd / mvster_warp_agg_bwd (first pass) call on the calling host thread launched, in the profiler's spelling with template arguments (e.g. "conv_lds_kernel<2, 1, 3, 1, 3>");
 * "" before the first call.  The pointer stays valid for the life of the library.  (bench.py attributes HIP-event timings
 * with it; there is no reference counterpart -- the reference's dispatch lives inside cuDNN.) */
const char* mvster_last_kernel(void);

/* Build flags of the loaded library: bit 0 = probe build (make -C mvster_amd/csrc probes).  The product library (0) reads no
 * environment variable and returns MVSTER_ERR_UNSUPPORTED for the kernel forms kept for the record only: warp variants 4
 * (pixel-major) and 5 (LDS-staged source windows), convolution variant 7 (ping-pong). */
int mvster_build_flags(void);

/* One v_mfma_f32_16x16x4_f32: A [16,4], B [4,16] -> D [16,16] (row major).  Test hook that pins the
 * fragment layout the convolution kernels assume. */
int mvster_mfma_probe(const float* A, const float* B, float* D, void* stream);

#ifdef __cplusplus
}
#endif
#endif
