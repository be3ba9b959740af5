/* robir_hip.h -- C ABI of librobir_hip.so: hand-written gfx950 (MI355X) kernels for RobIR's per-ray forward
 * renderer hot path.
 *
 * Conventions ("C-ABI layer"):
 *   - every pointer is a DEVICE pointer (hipMalloc'ed / a torch tensor's data_ptr()) unless marked HOST;
 *   - tensors are dense row-major fp32 unless stated; masks are uint8 (0/1); indices are int32 / int64 as declared;
 *   - the library never allocates, never synchronises and keeps no state: workspaces are caller-provided,
 *     kernels are enqueued on the given stream (rb_stream_t == hipStream_t, 0 = default stream);
 *   - every entry point returns 0 on success, non-zero on error (text via rb_last_error(), thread local)
 *     and never throws.
 *
 * Each group cites the reference interface (ingra14m/RobIR, file:line) it replaces.  The reference has no FFI
 * of its own for this path (it is pure PyTorch); INTEGRATION.md shows the ctypes binding a maintainer adds.
 */
#ifndef ROBIR_HIP_H
#define ROBIR_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define RB_ABI_VERSION 8

typedef void* rb_stream_t; /* hipStream_t */

int rb_abi_version(void);

/* Activation-range sentinel of the kernels that carry fp32 operands as f16 PIECES (the exact three-piece "f16x6" kernels of the default
 * policy; the (hi, lo) pairs of the legacy split-precision kernels).  The leading piece is a half: an operand whose magnitude reaches
 * 65504 no longer fits (it saturates -- round-toward-zero conversion -- and the pieces silently lose precision).  Every such kernel
 * therefore tracks the largest leading piece it consumed and, when one saturates (or is inf; NaN inputs are not an overflow), stores 1 into a per-family
 * word of a process-wide block of pinned, mapped host memory.  rb_range_check reads and clears the words:
 *   mask bit 0 light-visibility (rb_dvis_fused* / rb_dvis_stream*), 1 visibility MLP (rb_vis_x6_points), 2 SDF net (rb_sdf_x6*_points,
 *   rb_sdf_value_grad_x6*_points), 3 colour net (rb_color_x6*_points), 4 512-wide nets (rb_wide_x6*), 5 CESR nets (rb_cesr_net_x6_points);
 *   the legacy library's split-precision kernels (robir_hip_legacy.h) report into the same families of ITS block.
 * synchronize != 0: wait for `stream` first, so every kernel enqueued on it so far has reported; 0: no wait -- reports what
 * completed kernels have flagged (free of charge; call it at the next natural sync point for a complete answer).
 * A set bit means: re-run that family with the f32-input MFMA kernels (rb_*_mlp_points; ROBIR_MLP_PRECISION=fp32 / ROBIR_VIS_PRECISION=fp32).  Nothing like this
 * exists in the reference (PyTorch fp32 throughout); it guards the precision mode this library adds. */
int rb_range_check(int synchronize, rb_stream_t stream, unsigned* mask_out);
const char* rb_last_error(void);

/* ------------------------------------------------------------------------------------------------------------
 * Weight packing.  A network is handed to the MLP kernels as the concatenation of its packed layers.
 * Packed layer (n_pad, k_pad multiples of 16) = n_pad/16 chunks of (16 + 16*k_pad) floats:
 *   [bias of the 16 neurons] [kb][lane 0..63][r 0..3] = W[16*jb + (lane&15)][16*kb + 4*(lane>>4) + r]
 * i.e. exactly the A-operand order of v_mfma_f32_16x16x4_f32 (robir_amd/csrc/mlp_engine.h).
 * Replaces: nn.Linear / weight_norm parameter storage, model/neus_model.py:350-381,
 *           model/implicit_differentiable_renderer.py:186-193,241-248, model/sg_envmap_material.py:52-68.
 * k_perm (device int32[k_pad], may be NULL): packed input column k reads source column k_perm[k] (-1: zero).
 * ------------------------------------------------------------------------------------------------------------ */
long rb_packed_layer_floats(int n_pad, int k_pad);
int rb_pack_layer(const float* W, const float* b, int n_out, int k_in, int n_pad, int k_pad, const int* k_perm,
                  float w_scale, float* out, rb_stream_t stream);
/* exact-operand packing ("f16x6"): every weight * 2^scale_log2 as three halves h, m, l with w = h + m 2^-11 + l 2^-22 exactly;
 * chunk = 16 bias floats + [kb][h|m|l][lane][8 halves] = 16 + 24*k_pad floats; k_pad % 32 == 0. */
long rb_packed_layer_x6_floats(int n_pad, int k_pad);
int rb_pack_layer_x6(const float* W, const float* b, int n_out, int k_in, int n_pad, int k_pad, const int* perm,
                     int scale_log2, float* out, rb_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Feature construction (positional encodings), accurate sinf/cosf.
 * Replaces: model/embedder.py:7-55 (get_embedder), model/neus_model.py:14-57,71-91 (IPE).
 * ------------------------------------------------------------------------------------------------------------ */
/* X[M,64] = [PE10(x*scale) | extra[M] or 0];  jvp!=0: X[4M,64] with rows (PE, dPE/dx, dPE/dy, dPE/dz) per point */
int rb_feat_pe10(const float* x, long M, float scale, const float* extra, int jvp, float* X, rb_stream_t stream);
/* X[M,64] = full-covariance IPE(x, var*I) (60) [+ noise[M,60]*noise_scale] | 0 x4 */
int rb_feat_ipe(const float* x, long M, float var, const float* noise, float noise_scale, float* X,
                rb_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Fused MLPs (fp32 MFMA, activations resident in registers across all layers).
 * ------------------------------------------------------------------------------------------------------------ */
/* SDFNetwork.forward / .gradient (model/neus_model.py:385-438), ImplicitNetworkMy.forward/.gradient (:788-818).
 * X from rb_feat_pe10 (jvp rows for modes 2,3).  Wp packed [64->256, 256->256 x2, 256->208, 272->256, 256->256 x3,
 * 256->272] (modes 1,3) or [... , 256->16] (modes 0,2).
 * mode 0: out0[M] = sdf*out_scale          mode 1: out0[M,257] = (sdf,feat)*out_scale
 * mode 2: + grad[M,3] = d sdf/d(x*scale) * grad_scale (forward-mode), out0[M]      mode 3: same with out0[M,257]
 * Softplus(beta=100) is evaluated with the hardware exp/log/rcp (a few ulp); modes 4 and 6 are modes 0 and 2 with the
 * library expf/log1pf instead -- the octree build uses them because its split / hit thresholds must fall exactly where
 * the reference's do. */
/* The same op at the reference's precision (the default policy): value rows on the f32-input MFMA with the sigmoid of every hidden
 * pre-activation kept (k_sdf_mlp<5>), one pass over the transposed layers (k_sdf_back_f32; Wt / w8row from the host mirror's
 * packing.pack_sdf_back), the encoding's Jacobian (k_pe_grad_points) -- 2 x the value pass's MACs instead of the 4 x of the three
 * tangent rows per point of rb_sdf_mlp_points mode 3.  scratch: rb_sdf_value_grad_f32_scratch_floats(M) floats. */
/* The SDF value pass on EXACT fp32 operands (three f16 pieces per operand, six MFMA products per multiply-add in three fp32 accumulators:
 * not narrower than an fp32 fma chain; csrc/sdf_x6.hip) -- the default policy's value kernel.  x [M,3] points (encoded in the kernel,
 * x in_scale), Wp = the host mirror's packing.pack_sdf_x6(full = mode); mode 0: out0 [M] signed distances, 1: out0 [M,257].
 * rb_sdf_value_grad_x6_points: rb_sdf_value_grad_f32_points with that value pass and the pass over the transposed layers on exact
 * operands too (csrc/sdf_back_x6.hip; Wt / w8row = packing.pack_sdf_back_x6; same scratch). */
/* two_tile = 0: one 16-row tile per wave, rounds of 64 rows (csrc/sdf_x6.hip); 1: TWO tiles per wave (csrc/sdf_x6t.hip + x6t_engine.h,
 * round 4): a weight fragment read from the LDS feeds two MFMAs per product, a pass of the weights serves 128 rows -- same blob, same
 * arithmetic, the products of a class summed part by part: the two forms agree to fp32 summation order, each with itself bit for bit.
 * The host mirror takes the two-tile form where it needs fewer than two thirds of the one-tile form's passes over the persistent grid
 * (16385-32768 rows, > 49152 rows on 256 compute units: ops.sdf_two_tile); elsewhere rounds of 64 rows fill the chip better. */
int rb_sdf_x6_points(const float* x, long M, float in_scale, const float* Wp, int mode, float out_scale, float* out0, int two_tile,
                     int n_workgroups, rb_stream_t stream);
/* The colour net on exact three-piece operands (csrc/color_x6.hip; two_tile = 1: csrc/color_x6t.hip, as above; Wp = packing.pack_color_x6):
 * the arguments of rb_color_mlp_points. */
int rb_color_x6_points(const float* feat, long feat_stride, float feat_scale, const float* x, float x_scale, const float* view,
                       const float* normal, long M, const float* Wp, float* rgb, int two_tile, int n_workgroups, rb_stream_t stream);
/* The visibility MLP on exact three-piece operands (csrc/vis_x6.hip; Wp = packing.pack_vis_x6): the arguments of rb_vis_mlp_points. */
int rb_vis_x6_points(const float* p, const float* d, long M, int rep, const float* Wp, float* logits, int n_workgroups, rb_stream_t stream);
/* The 512-wide ReLU nets (SparseAE encoder: raw latent [M,32]; indirect-illumination decoder: raw SG outputs [M,144]) on exact
 * three-piece operands (csrc/wide_x6.hip; Wp = packing.pack_wide_x6): the arguments of rb_wide_mlp_points. */
int rb_wide_x6_points(const float* x, const float* extra, long M, const float* Wp, int encoder, float* Y, int n_workgroups, rb_stream_t stream);
int rb_wide_x6(const float* X /* feature rows [M,64] */, long M, const float* Wp, int encoder, float* Y, int n_workgroups, rb_stream_t stream);
/* The CESR nets on exact three-piece operands (csrc/cesr_x6.hip; Wp = packing.pack_softplus512_x6): the arguments of rb_cesr_net_points
 * (kind 0: normal_net on PE10(x), 2: shadow_net on (point, one-hot label) rows). */
int rb_cesr_net_x6_points(const float* x, long M, int kind, int n_label, const float* Wp, float* Y, int n_workgroups, rb_stream_t stream);
/* The CESR nets in PLAIN f16 (csrc/cesr_f16.hip, round 6): ONE f16 MFMA product per multiply-add, f16 weights (the h pieces of the
 * exact-operand blob: Wp = packing.pack_softplus512_f16 = [16 biases per chunk of the stream][1 KB fragments, k-block major]), f16
 * activations truncated between the layers, fp32 accumulation -- the labelled THROUGHPUT mode BASELINE.json configs[4] names
 * ("fp16 MLP weights on MFMA", ROBIR_PRECISION=f16): NARROWER than the reference's fp32, never a parity claim.  The arguments of
 * rb_cesr_net_x6_points + tiles = 16-row tiles per wave the library was built with (3). */
int rb_cesr_net_f16_points(const float* x, long M, int kind, int n_label, const float* Wp, float* Y, int tiles, int n_workgroups,
                           rb_stream_t stream);
/* two_tile = 1: two tiles per wave in the value and the reverse pass (csrc/sdf_x6t.hip, sdf_back_x6t.hip); Wt then = the transposed layers
 * packed with W3^T's K padded to 256 (packing.pack_sdf_back_x6(two_tile=True)).  Same scratch. */
int rb_sdf_value_grad_x6_points(const float* x, long M, float in_scale, const float* Wp, const float* Wt, const float* w8row,
                                float out_scale, float grad_scale, float* out0, float* grad, float* scratch, int two_tile,
                                rb_stream_t stream);
long rb_sdf_value_grad_f32_scratch_floats(long M);
int rb_sdf_value_grad_f32_points(const float* x, long M, float in_scale, const float* Wp, const float* Wt, const float* w8row,
                                 float out_scale, float grad_scale, float* out0, float* grad, float* scratch, rb_stream_t stream);
/* rb_sdf_mlp with the positional encoding fused (every mode, tangent rows of the forward-mode gradient included): x [M,3],
 * evaluated at x * in_scale; bit-identical to rb_feat_pe10 (jvp for modes 2, 3, 6) + rb_sdf_mlp. */
int rb_sdf_mlp_points(const float* x, long M, float in_scale, const float* Wp, int mode, float out_scale, float grad_scale,
                      float* out0, float* grad, rb_stream_t stream);
/* The small MLPs that take an encoded point straight from the points (encoding fused, load_features_pe10x / load_features_vis in
 * csrc/mlp_engine.h; each bit-identical to the row form named):
 *   rb_vis_mlp_points / _h3_points   VisNetwork.forward(p, d): rep consecutive directions per point   (= rb_feat_vis + rb_vis_mlp[_h3])
 *   rb_linear_pe10_256               the 64 -> 256 first-layer halves of the light-visibility net        (= rb_feat_pe10 + rb_linear_64_256)
 *   rb_wide_mlp_points / _h3_points  64 -> 512 x4 nets on [PE10(x) | extra]: indirect-illumination lobes (extra = hdr_shift [M]) and
 *                                    SparseAE encoders (extra NULL)                                       (= rb_feat_pe10 + rb_illum_mlp / rb_ae_encode / rb_wide_mlp_h3) */
int rb_vis_mlp_points(const float* p, const float* d, long M, int rep, const float* Wp, float* logits, rb_stream_t stream);
int rb_linear_pe10_256(const float* x, long M, const float* Wp, float* Y, rb_stream_t stream);
int rb_wide_mlp_points(const float* x, const float* extra, long M, const float* Wp, int encoder, float* Y, rb_stream_t stream);
/* CESR nets straight from the points (training/train_cesr.py:106-110,331-352): kind 0 = normal_net on PE10(x) [M rows], kind 2 =
 * shadow_net on (point, one-hot label) rows [M = points * n_label rows]; = rb_feat_pe10 + rb_cesr_net[_h3]. */
int rb_cesr_net_points(const float* x, long M, int kind, int n_label, const float* Wp, float* Y, rb_stream_t stream);
/* f32-input-MFMA form of the same (rb_feat_color + rb_color_mlp without the assembled [M,304] rows). */
int rb_color_mlp_points(const float* feat, long feat_stride, float feat_scale, const float* x, float x_scale, const float* view,
                        const float* normal, long M, const float* Wp, float* rgb, rb_stream_t stream);
/* raw[M,24,6] -> lgt_sgs[M,24,7] (implicit_differentiable_renderer.py:208-218). */
int rb_illum_decode(const float* raw, long M, float* sgs, rb_stream_t stream);
/* CESR-stage nets (training/train_cesr.py:106-110; SDFNetwork with multires 0, model/neus_model.py:312-417):
 * kind 0: normal_net X[M,64] (PE10 | 0) -> Y[M,3];   kind 1: shadow_net X[M,192] ([PE10 | one-hot 128 | 0]) -> Y[M,2];
 * kind 2: shadow_net on (point, label) pairs: X = Xp[M/n_label,64] point features, row = point*n_label + label,
 *         the one-hot block is synthesised in registers -> Y[M,2].
 * Wp packed [K0P->512, 512->512 x2, 512->N3P, 528->512 (cols [lin3 | input]), 512->512 x3, 512->16]. */
int rb_cesr_net(const float* X, long M, int kind, int n_label, const float* Wp, float* Y, rb_stream_t stream);
/* SparseAE (model/sg_envmap_material.py:40-99): encoder X[M,64] -> raw latent[M,32]
 * (packed [64->512, 512->512 x3, 512->32]); latent = act(raw*(1-var)) [+ lat2 = latent + noise*noise_scale];
 * decoder latent[M,32] -> Y[M,n_out] (packed [32->128, 128->128, 128->16]). act: 0 sigmoid, 1 softplus, 2 none (SparseAE.encode, :96-99). */
int rb_ae_encode(const float* X, long M, const float* Wp, float* raw_latent, rb_stream_t stream);
int rb_ae_latent(const float* raw, long M, const float* var, int act, const float* noise, float noise_scale, float* lat,
                 float* lat2, rb_stream_t stream);
int rb_ae_decode(const float* lat, long M, const float* Wp, int n_out, int sigmoid_out, float* Y, rb_stream_t stream);
/* y = a + s*b (n floats) */
int rb_axpy(const float* a, const float* b, float s, long n, float* y, rb_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Light-SG ("diffuse") visibility -- get_diffuse_visibility, model/sg_render.py:111-195.
 * rb_dvis_dirs: per chunk c (C chunks share the light lgt[L,7] but have their own draws u_theta/u_phi[C,L,nsamp]):
 *   dirs[C*L*nsamp,3], wdir[C*L*nsamp] = exp(lambda(d.axis-1)), wsum[C*L] = sum_s wdir + 1e-6.
 *   direct = 0: lgt holds RAW light SGs (render_with_sg's call chain: lobe normalised twice, |lambda|, sg_render.py:364-366,126);
 *   direct = 1: lgt[:, :3] / lgt[:, 3] are the lgtSGLobes / lgtSGLambdas of a direct get_diffuse_visibility call
 *   (normalised once, lambda as given and only clamped >= 1e-4 for the cone, :126-134,179).
 * rb_dvis_fused: one workgroup per point.  A[n,256] = W0[:, :63].PE10(p)+b0 and Bd[C*L*nsamp,256] = W0[:,63:].PE10(d)
 *   (rb_linear_64_256), Whid = packed [256->256 x3], wlast[2,256], blast[2] row-major; chunk_id[n] int32 or NULL.
 *   vis_out[n,L] (the reference returns the transpose [L,n]); eval_count (may be NULL) += surviving (p,d) pairs.
 *   precision 0: hidden layers on the f32-input MFMA (exact fp32 fma chain), Whid from rb_pack_layer;
 *   precision 5: the first-generation split-precision kernel -- legacy library only (include/robir_hip_legacy.h).
 * The production path is rb_dvis_fused_x6t / rb_dvis_stream_x6 below.
 * ------------------------------------------------------------------------------------------------------------ */
int rb_dvis_dirs(const float* lgt, int L, int nsamp, int C, int direct, const float* u_theta, const float* u_phi, float thr,
                 float* dirs, float* wdir, float* wsum, rb_stream_t stream);
int rb_dvis_fused(const float* normals, const int* chunk_id, long n, const float* A, const float* Bd, const float* dirs,
                  const float* wdir, const float* wsum, const float* Whid, const float* wlast, const float* blast, int L,
                  int nsamp, int argmax_vis, int precision, int scale_log2, float* vis_out,
                  unsigned long long* eval_count, rb_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * BRDF-lobe ("specular") visibility -- get_specular_visibility, single-view branch, model/sg_render.py:198-301.
 * rb_spec_vis_sample: sharp[n] = clip(warped lambda, 0.1, 50), chunk_min[C] (uint32 bit pattern of the per-chunk
 *   minimum: the reference's batch-global sharpness.min()), dirs[n*nsamp,3], wts[n*nsamp], front[n*nsamp] (uint8).
 * rb_spec_vis_reduce: logits[n*nsamp,2] of the visibility MLP -> bvis[n]; inv: use softmax[...,0] (indirect pass).
 * ------------------------------------------------------------------------------------------------------------ */
/* The light-visibility stage with EXACT fp32 operands on the f16 matrix pipe ("f16x6", csrc/vis_diffuse_x6t.hip + x6t_engine.h; the
 * DEFAULT and bench.py's headline kernel): every operand as three halves (h + m 2^-11 + l 2^-22 = the fp32 value exactly), the six
 * partial products of weight >= 2^-22 in three fp32 accumulators by weight class -- not narrower than the reference's fp32
 * (VisNetwork, model/implicit_differentiable_renderer.py:241-258, evaluated by nn.Linear in fp32).  Since round 6 the two outer
 * products of the 2^-22 class (h.xl, l.xh) are formed from bf8 copies of their operands on v_mfma_f32_16x16x128_f8f6f4 (twice the f16
 * rate; the other four products stay exact; error against float64 unchanged: DESIGN.md section 6).  W49 = the three hidden layers and
 * the 256->2 output layer (rows padded to 16) packed by rb_pack_layer_x6 back to back = 49 chunks, then re-arranged per half chunk
 * of 128 K as [k-block 0..3][h | m][lane][8 halves] (8 KB), [h8: 2 planes][lane][16 bytes] (2 KB), [l8] (2 KB) with byte 4 j + r of
 * a lane's 32 = the e5m2 rounding of half 4 (j % 2) + r of k-block 4 G + j / 2 (robir_amd/packing.py: repack_x6_chunks_fp8): the same
 * size; scale_log2 = 8 NAMES this layout (a library built with -DXT_FP8=0 takes rb_pack_layer_x6's own layout and scale_log2 = 0; a
 * mismatch is refused).  One workgroup of
 * four waves per point, TWO 16-sample tiles per wave (a weight fragment read from the LDS feeds both tiles, a pass of the weights
 * serves 128 samples), output layer on the matrix pipe.  Other arguments as rb_dvis_fused; vis_out agrees with its precision 0 to
 * fp32 summation order. */
int rb_dvis_fused_x6t(const float* normals, const int* chunk_id, long n, const float* A, const float* Bd, const float* dirs,
                      const float* wdir, const float* wsum, const float* W49, int L, int nsamp, int argmax_vis, int scale_log2,
                      float* vis_out, unsigned long long* eval_count, rb_stream_t stream);
/* The same arithmetic as THREE launches on `stream` (the form for launches up to ~8192 points, e.g. one 1024-pixel chunk):
 *   cull      one workgroup per point: n.d > 1e-6 survivors compacted into a global list of 16-sample tiles;
 *   stream    a PERSISTENT grid (n_workgroups; <= 0: one per CU) walks the tile list eight tiles per round, whatever point
 *             they belong to (balanced over the CUs for any n; the tile count is read from device memory: no host sync);
 *   reduce    one workgroup per point: SG-weighted mean per lobe in the fixed sample order.
 * Every pair goes through rb_dvis_fused_x6t's instruction sequence: vis_out is bit-identical to it.  Caller-provided scratch (device):
 *   pair_j[n*L*nsamp] u16, pair_vis[n*L*nsamp] f32, tile_info[n*L*nsamp/16][2] i32, point_info[n][2] i32, counters[2] u64.
 * L*nsamp must be a multiple of 16 (other shapes: rb_dvis_fused_x6t). */
int rb_dvis_stream_x6(const float* normals, const int* chunk_id, long n, const float* A, const float* Bd, const float* dirs,
                      const float* wdir, const float* wsum, const float* W49, int L, int nsamp, int argmax_vis, int scale_log2,
                      unsigned short* pair_j, float* pair_vis, int* tile_info, int* point_info, unsigned long long* counters,
                      int n_workgroups, float* vis_out, unsigned long long* eval_count, rb_stream_t stream);
/* The same stage in PLAIN f16 (csrc/vis_diffuse_f16t.hip, round 4): ONE f16 MFMA product per multiply-add, fp32 accumulation, weights
 * = the round-to-nearest f16 of the fp32 weights (the h pieces of the blob rb_dvis_stream_x6 takes), activations truncated to f16
 * between the layers -- the labelled throughput mode BASELINE.json configs[4] names, NARROWER than the reference's fp32, never a
 * default (ROBIR_PRECISION=f16 only).  Arguments and scratch of rb_dvis_stream_x6. */
int rb_dvis_stream_f16(const float* normals, const int* chunk_id, long n, const float* A, const float* Bd, const float* dirs,
                       const float* wdir, const float* wsum, const float* W49, int L, int nsamp, int argmax_vis, int scale_log2,
                       unsigned short* pair_j, float* pair_vis, int* tile_info, int* point_info, unsigned long long* counters,
                       int n_workgroups, float* vis_out, unsigned long long* eval_count, rb_stream_t stream);
/* scale_log2 of rb_dvis_stream_f16 names the weight blob: 0 = rb_dvis_stream_x6's exact-operand blob (its h pieces read in place, the
 * round-4 kernel), 1 = the f16 blob (49 x 16 bias floats, then the h fragments of the 49 chunks, 8 KB each: what W49h below is) and the
 * second-generation kernel (two chunks per barrier step; the same bits).
 *
 * The f16 mode in the POINT-BLOCK form (csrc/vis_diffuse_f16p.hip, round 5; ABI 7): a 16-sample tile = SIXTEEN consecutive points x ONE
 * direction of their chunk, kept when any of the sixteen faces it (lanes that do not are computed and dropped): a round of sixteen tiles
 * reads 16 rows of A and 16 rows of Bd by whole-row LDS-DMA copies instead of 256 Bd rows sixteen cache lines at a time -- the gather
 * that one MFMA per multiply-add cannot hide.  Three launches on `stream` (cull per point block, persistent grid over the rounds,
 * reduce per block); every pair goes through rb_dvis_stream_f16's instruction sequence: vis_out is bit-identical to it.
 *   chunk_id must be ASCENDING (the renderer's hit points are); if it is not, vis_out comes back NaN (never wrong numbers).
 *   items_max >= ceil(n / 16) + (number of distinct chunk ids) - 1.  Scratch (device, caller-provided): entries[items_max * L*nsamp] u32,
 *   pair_vis[items_max * L*nsamp * 16] f32, round_info[items_max * L*nsamp / 16][4] i32, item_info[items_max][4] i32, counters[4] u64. */
int rb_dvis_pblock_f16(const float* normals, const int* chunk_id, long n, const float* A, const float* Bd, const float* dirs,
                       const float* wdir, const float* wsum, const float* W49h, int L, int nsamp, int argmax_vis, int items_max,
                       unsigned* entries, float* pair_vis, int* round_info, int* item_info, unsigned long long* counters,
                       int n_workgroups, float* vis_out, unsigned long long* eval_count, rb_stream_t stream);
/* ------------------------------------------------------------------------------------------------------------
 * Traced light visibility -- OctreeVisModel (model/octree_tracing.py:63-85) as the VisModel of get_diffuse_visibility
 * (model/sg_render.py:111-195), the mode `trace_vis` switches on (training/train_pbr.py:409-410): csrc/octree_vis.hip.
 * rb_dvis_octree: same inputs / vis_out[n,L] as rb_dvis_fused_x6t with points[n,3] instead of the MLP rows and the octree
 *   tables of rb_octree_cast_*; the surviving (point, direction) pairs of each chunk, in the reference's order, are traced in
 *   lock-step batches of `batch_pairs` (reference: 2 000 000, sg_render.py:158) with max_iter (32) -- per batch the step size
 *   (0.01 beyond 100 000 rays, else 0.005) and the per-iteration fine-march count follow utils/octree.py:542-549.
 *   chunk_id must be ascending (points of a chunk contiguous).  No host synchronisation.  Scratch (device, caller-provided):
 *   pcount[n] i32, prank[n] i32, chunk_tab[4*n_chunks+4] i64, group_tab[2*max_groups] i64, counters[34*max_groups] i32,
 *   pair_p[cap] i32, pair_j[cap] u16, t_st[cap] f32, leaf_st[cap] i32, act_st[cap] u8, grp[cap] i32 (cap = n*L*nsamp),
 *   point_span[2n] i64, layout[4 + 8192] i64 (out: [0] pairs traced; [4 + 2 b], [5 + 2 b] = 32-byte octree records read / ray-iterations counted by workgroup b mod 4096 -- per-workgroup slots instead of device-wide atomics, the caller adds them up).  max_groups >= sum over chunks of
 *   ceil(pairs / batch_pairs) (<= n_chunks * ceil(points per chunk * L*nsamp / batch_pairs)).
 * rb_octree_cast_grouped: the grouped lock-step secondary cast for explicit rays: group g = rays group_start[g] ..
 *   group_start[g+1]-1 (device array of G+1 offsets) advances on its own schedule, exactly as if each group were a
 *   separate rb_octree_cast_* call with max_iter.  Outputs as OctreeTracing.forward.  Scratch: gsize[G] i64, grp[R] i32,
 *   t_st[R] f32, leaf_st[R] i32, act_st[R] u8, counters[34*G] i32.
 * ------------------------------------------------------------------------------------------------------------ */
/* alive_a .. n_alive (all six non-NULL): the rays still active are compacted -- stably, by a prefix sum: list order = (point, direction)
 * order -- between the lock-step iterations, so that an iteration reads and steps only live rays in full waves (all six NULL: the plain
 * form, which walks every pair 33 times).  Bit-identical vis_out.  alive_a, alive_b int32[cap], flags uint8[cap + 8] (cap = size of pair_p,
 * < 2^31), blk_cnt int32[cap / 2048 + 2], blk_off int64[cap / 2048 + 2], n_alive int64[40] (list sizes + per-iteration cursors). */
int rb_dvis_octree(const float* node, const float* nrm, long B, const float* root_min, const float* root_size, const int* res,
                   const float* points, const float* normals, const int* chunk_id, long n, int n_chunks, const float* dirs,
                   const float* wdir, const float* wsum, int L, int nsamp, int argmax_vis, long batch_pairs, int max_iter,
                   int* pcount, int* prank, long* chunk_tab, long* group_tab, int max_groups, int* counters, int* pair_p,
                   unsigned short* pair_j, float* t_st, int* leaf_st, unsigned char* act_st, int* grp, long* point_span,
                   long* layout, int* alive_a, int* alive_b, unsigned char* flags, int* blk_cnt, long* blk_off, long* n_alive,
                   float* vis_out, unsigned long long* eval_count, rb_stream_t stream);
int rb_octree_cast_grouped(const float* node, const float* nrm, long B, const float* root_min, const float* root_size,
                           const int* res, const float* origins, const float* dirs, long R, const long* group_start, int G,
                           int max_iter, float clamp_dt, long* gsize, int* grp, float* t_st, int* leaf_st,
                           unsigned char* act_st, int* counters, float* x_out, unsigned char* hit_out, float* t_out,
                           rb_stream_t stream);
int rb_spec_vis_sample(const float* normal, const float* view, const float* rough, const int* chunk_id, long n,
                       int n_chunks, int nsamp, const float* u_theta, const float* u_phi, float* sharp,
                       unsigned* chunk_min, float* dirs, float* wts, unsigned char* front, rb_stream_t stream);
/* The same sampling stage for a caller that passes its own lobes / lambdas -- the reference's exact signature
 * get_specular_visibility(points, normals, viewdirs, VisModel, lgtSGLobes [n,3], lgtSGLambdas [n,1], nsamp, ...), model/sg_render.py:198-223:
 * the cone is still built around the reflection of the view about the normal (:204-207), its opening from clip(lambdas, 0.1, 50) and the
 * batch-global (per chunk_id) minimum (:219-223), the sample weights exp(sharp (d . lobes - 1)) from the passed lobes AS GIVEN (:281). */
int rb_spec_vis_sample_lobes(const float* normal, const float* view, const float* lobes, const float* lambdas, const int* chunk_id,
                             long n, int n_chunks, int nsamp, const float* u_theta, const float* u_phi, float* sharp,
                             unsigned* chunk_min, float* dirs, float* wts, unsigned char* front, rb_stream_t stream);
int rb_spec_vis_reduce(const float* logits, const unsigned char* front, const float* wts, long n, int nsamp, int inv,
                       int argmax_vis, int testing, float* bvis, rb_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * SG shading -- render_with_sg (model/sg_render.py:343-565) incl. lambda_trick (:84-104), hemisphere_int (:62-81).
 * lgt: [M,7] (per_point_lgt=0) or [n,M,7]; light_vis[n,M] or NULL (comp_vis=False); metallic[n] or NULL;
 * indir_integral[n,3] or NULL (replaces the diffuse term).  f0: DEVICE pointer to the scalar specular reflectance
 * (|specular_reflectance|, sg_envmap_material.py:160; read by the kernel, no host copy).  Outputs [n,3]; out_shadow may be NULL.
 * ------------------------------------------------------------------------------------------------------------ */
int rb_sg_shade(const float* normal, const float* view, const float* lgt, int per_point_lgt, int M, const float* f0,
                const float* rough, const float* albedo, const float* metallic, const float* light_vis,
                const float* bvis, const float* indir_integral, int lin_diff, long n, float* out_rgb, float* out_spec,
                float* out_diff, float* out_shadow, rb_stream_t stream);
/* render_envmap_sg (model/sg_render.py:26-42): rgb[n,3] = sum_k |mu_k| exp(|lambda_k| (d . lobe_k/|lobe_k| - 1)), lgt[M,7] */
int rb_envmap_sg(const float* lgt, int M, const float* dirs, long n, float* rgb, rb_stream_t stream);
/* render_envmap (model/sg_render.py:45-59): bilinear lookup of env[H,W,3] (lat-long, row-major) along dirs[n,3]; same
 * coordinates and border handling as the reference's F.grid_sample(align_corners=True). */
int rb_envmap_lookup(const float* env, int H, int W, const float* dirs, long n, float* rgb, rb_stream_t stream);
/* y = x/(|x|+eps) (mode 0) or x/max(|x|,eps) (mode 1) on rows of 3 */
int rb_normalize3(const float* x, long n, float eps, int mode, float* y, rb_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Octree over the SDF zero set + lock-step sphere tracer.
 * Replaces: utils/octree.py:124-199,217-265,377-438,459-471,493-592 (OctreeSDF build / query / cast /
 * multi_step_cast / fast_volume_render / first_nonzero=torch_scatter.scatter_min), model/octree_tracing.py:31-60.
 * Tables: node[B][8] floats = {min.xyz, bits(first_child int32, -1 leaf)}, {size.xyz, sdf_val}; nrm[B][3] unit
 * gradient; children of a split node are contiguous; base grid cell (ix,iy,iz) = (ix*res[1]+iy)*res[2]+iz.
 * root_min[3], root_size[3], res[3] are HOST arrays.
 * Build steps (host loops over <= 4 levels; sdf/grad come from rb_sdf_mlp on `centre` rows):
 *   base_grid -> { mark_split (flag, exclusive rank, *total = number of splits) -> subdivide } x levels -> store_cells.
 * Cast: a lock-step batch is (a) one workgroup of rb_octree_cast_batched when it has <= 1024 rays -- R_total rays are
 *   cut into consecutive batches of `batch` rays, origins[n_batches,3] (per_ray_origin=0) or [R_total,3];
 *   sched (may be NULL): int32[n_batches, sched_cap, 2] receives (n_active, multi_samp) per iteration --
 *   or (b) init / iter / finish launches for one batch of any size (counters: int32[>= max iterations + 2], zeroed).
 * max_iter <= 0: primary rays (run until no ray is active); max_iter = 32: secondary rays (origin + 0.005 d,
 *   break after max_iter+1 iterations, still-active rays are hits).  step: 0.001 primary, 0.005 / 0.01 secondary.
 * clamp_dt = 10*min_step.  Outputs: x[R,3] = t*d + o, hit[R] uint8, t[R].
 * ------------------------------------------------------------------------------------------------------------ */
int rb_octree_base_grid(const float* root_min, const float* root_size, const int* res, float* node, float* centre,
                        rb_stream_t stream);
int rb_octree_mark_split(const float* node, long first, long count, const float* sdf, float thr, int* flag, int* rank,
                         int* scan_tmp, int* total, rb_stream_t stream);
int rb_octree_subdivide(float* node, float* centre, long first, long count, const int* flag, const int* rank,
                        long new_first, rb_stream_t stream);
int rb_octree_store_cells(float* node, float* nrm, long first, long count, const float* sdf, const float* grad,
                          rb_stream_t stream);
int rb_octree_cast_batched(const float* node, const float* nrm, long B, const float* root_min, const float* root_size,
                           const int* res, const float* origins, int per_ray_origin, const float* dirs, long R_total,
                           int batch, int max_iter, double step, float clamp_dt, float* x_out, unsigned char* hit_out,
                           float* t_out, int* sched, int sched_cap, rb_stream_t stream);
int rb_octree_cast_init(const float* node, const float* nrm, long B, const float* root_min, const float* root_size,
                        const int* res, const float* origins, const float* dirs, long R, int max_iter, float* t,
                        int* leaf, unsigned char* active, int* counters, rb_stream_t stream);
int rb_octree_cast_iter(const float* node, const float* nrm, long B, const float* root_min, const float* root_size,
                        const int* res, const float* origins, const float* dirs, long R, int max_iter, double step,
                        int it_first, int it_count, float* t, int* leaf, unsigned char* active, int* counters,
                        rb_stream_t stream);
int rb_octree_cast_finish(const float* node, const float* nrm, long B, const float* root_min, const float* root_size,
                          const int* res, const float* origins, const float* dirs, long R, int max_iter, float clamp_dt,
                          const float* t, const int* leaf, float* x_out, unsigned char* hit_out, float* t_out,
                          rb_stream_t stream);
/* init + every iteration + finish of one lock-step batch of any size in ONE launch (persistent grid with a grid barrier instead of
 * one launch per iteration): counters[it_limit + 1] int32 (output: rays active at the start of each iteration; it_limit = max_iter + 1,
 * or max_total when max_iter <= 0: iterate until no ray is active) and arrive[1024] 64-bit words zeroed by the caller (passed as int*,
 * 8-byte aligned): the barrier's slots, 2 x 512 selected by epoch parity -- workgroup g publishes {epoch, its active count} in slot
 * [epoch & 1][g], so the grid is capped at 512 workgroups.  The same results as the per-iteration launches bit for bit.
 * The barrier spins, so the grid must be co-resident: the launch is a COOPERATIVE one (hipLaunchCooperativeKernel; grid <= what the
 * occupancy query of the current device allows, cached per device id).  Returns 0 = done, 2 = the runtime refused the launch (another
 * stream or rank holds compute units; cooperative launches unsupported): NOTHING ran, take rb_octree_cast_init / _iter / _finish;
 * any other value = error. */
int rb_octree_cast_coop(const float* node, const float* nrm, long B, const float* root_min, const float* root_size,
                        const int* res, const float* origins, const float* dirs, long R, int max_iter, double step, int max_total,
                        float clamp_dt, float* t, int* leaf, unsigned char* active, int* counters, int* arrive, float* x_out,
                        unsigned char* hit_out, float* t_out, rb_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Ray generation / points / tone mapping.
 * rb_camera_rays_dev: get_camera_params + lift, 4x4 pose (utils/rend_util.py:51-97).  pose_dev: 16 floats row-major
 *   camera-to-world, K_dev: 9 floats row-major intrinsics, both in DEVICE memory (no host copy on the per-chunk path);
 *   uv[N,2] (x = column, y = row) -> dirs[N,3].
 * rb_points_along: pts = o + t*d (implicit_differentiable_renderer.py:324); origins [N/batch,3] or per ray.
 * rb_tonemap: ACESToneMapping hdr_mode 0 (model/color_correction.py:52-60,116-134): mode 0 hdr2ldr, 1 ldr2hdr,
 *   2 ldr2hdr(x^2.2); rows of 3 channels, shift[n] (stride 1) or one value (stride 0), clamped to [1e-4,1].
 * ------------------------------------------------------------------------------------------------------------ */
int rb_camera_rays_dev(const float* pose_dev, const float* K_dev, const float* uv, long N, float* dirs, rb_stream_t stream);
int rb_points_along(const float* origins, int per_ray_origin, long batch, const float* dirs, const float* t, long N,
                    float* pts, rb_stream_t stream);
/* ACESToneMapping (model/color_correction.py:31-73,116-134): mode = op + 16 * curve; op 0 hdr2ldr, 1 ldr2hdr, 2 ldr2hdr(x^2.2);
 * curve 0 = hdr_mode 0 (scale_aces, every shipped conf), 1 = hdr_mode 1 (warp_aces), 2 = hdr_mode 2 (ln_space), 3 = identity. */
/* K row sets src[k] [n, src_width[k]] of the hit pixels idx[n] (int64, ascending or not) -> K consecutive blocks [N, dst_width[k]] of
 * `flat` (block k starts at N * sum of the widths before it; the caller pre-fills the defaults), one launch: the per-pixel outputs of
 * IDRNetwork.forward (implicit_differentiable_renderer.py:420-470, `x[mask] = values` per output).  src_width 1 broadcasts over the
 * destination's columns.  src / src_width / dst_width are HOST arrays. */
int rb_scatter_rows(const float* const* src, const int* src_width, const int* dst_width, int K, const long* idx, long n, long N,
                    float* flat, rb_stream_t stream);
int rb_tonemap(const float* x, long n, const float* shift, int shift_stride, int mode, float* y, rb_stream_t stream);
/* Element-wise heads of the hooks / networks:
 * rb_material_decode: EnvmapMaterialNetwork outputs from the spec-AE's [n,5] (model/sg_envmap_material.py:205-211);
 * rb_abs_scale: y = (take_abs ? |x| : x)*s (implicit_differentiable_renderer.py:220 abs; training/train_pbr.py:365 `* 2 pi`);
 * rb_softmax2: torch.softmax(logits[n,2], -1)[..., which];
 * rb_lin_diff_combine: diffuse*albedo/pi + specular on [n,3] (training/train_cesr.py:523-524). */
int rb_material_decode(const float* brdf, const float* brdf_r, long n, float* albedo, float* rough, float* metal,
                       float* albedo_r, float* rough_r, float* metal_r, rb_stream_t stream);
int rb_abs_scale(const float* x, long n, float s, int take_abs, float* y, rb_stream_t stream);
int rb_softmax2(const float* logits, long n, int which, float* p, rb_stream_t stream);
int rb_lin_diff_combine(const float* diffuse, const float* albedo, const float* spec, long n, float* rgb,
                        rb_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Secondary rays / NeuS compositing -- IDRNetwork.trace_radiance (implicit_differentiable_renderer.py:566-650),
 * ImplicitNetworkMy.borrow_color / volume_render (model/neus_model.py:828-871), render_core weights
 * (model/sdf_render.py:209-240).
 * rb_sphere_dirs: u1,u2[n*nsamp] uniform draws -> dirs[n*nsamp,3]; back[n*nsamp] = (n.d < 0) with n = normal/max(|n|,1e-4);
 *   cosw = relu(n.d); origins[n,3] = points + 0.005 n.
 * rb_borrow_points: x[m*ns,3] = 2 p + dir t_k, dirs[m*ns,3] = dir = -view/|view|, tk[ns] device.
 * rb_neus_composite: alpha_k = clip((sig(s_k inv_s) - sig(s_{k+1} inv_s) + 1e-5)/(sig(s_k inv_s) + 1e-5), lo, hi) [* mask],
 *   (last sample uses s_{ns-1} twice), w_k = alpha_k prod_{j<k}(1 - alpha_j + eps); rgb[m,3] = sum w_k color_k
 *   (color, mask, rgb, weights may be NULL).
 * rb_trace_integrate: out[n,3] = sum_k rad cosw / max(#front-facing, 1e-4).
 * ------------------------------------------------------------------------------------------------------------ */
int rb_sphere_dirs(const float* u1, const float* u2, const float* normals, const float* points, long n, int nsamp,
                   float* dirs, unsigned char* back, float* cosw, float* origins, rb_stream_t stream);
int rb_borrow_points(const float* points, const float* view, const float* tk, long m, int ns, float* x, float* dirs,
                     rb_stream_t stream);
int rb_neus_composite(const float* sdf, const float* color, const float* mask, long m, int ns, float inv_s, float lo,
                      float hi, float eps, float* rgb, float* weights, rb_stream_t stream);
int rb_trace_integrate(const float* rad, const float* cosw, const unsigned char* back, long n, int nsamp, float* out,
                       rb_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * NeuS ray-march with hierarchical sampling -- the non-MLP steps of render_neus (model/sdf_render.py:263-374):
 * rb_neus_coarse_z  z[R,n] = near + (far-near)*lin[n]                                   (:279-283)
 * rb_neus_jitter_z  perturb > 0: z[r,:] += (u[r] - 0.5) * 2.0 / n, u[R] = the ray's one uniform draw (:293-295)
 * rb_ray_points     pts[R*n,3] = o + d*z (dirs[R*n,3] = d per sample, may be NULL)      (:313, :192-196)
 * rb_neus_upsample  up_sample + sample_pdf(det=True): z_new[R,n_new]; u[n_new] = the deterministic quantiles;
 *                   wtmp[R,n] scratch                                                   (:70-114, :37-67)
 * rb_neus_merge     cat_z_vals: merge ascending z lists, sdf follows (sdf_new NULL on the last step, sdf_out may be NULL) (:117-132)
 * rb_neus_mid_z     zmid = z + dz/2, last dz = sample_dist                              (:186-189)
 * rb_neus_finish    render_core compositing + render_neus epilogue: alpha from consecutive mid-point SDFs
 *                   (sdf = column 0 of an [R*n, sdf_stride] matrix), inside-sphere mask, weights, white background,
 *                   normal (set to 1 where acc > 0.8, sic), dist clipped to [near,far];
 *                   gerr[2] += (sum relax*(|grad|-1)^2, sum relax)                      (:203-260, :354-374)
 * rb_surface_points / rb_surface_finish: NormalTrainRunner.get_neus_surface (training/train_normal.py:239-286).
 * ------------------------------------------------------------------------------------------------------------ */
int rb_neus_coarse_z(const float* near, const float* far, const float* lin, long R, int n, float* z, rb_stream_t stream);
int rb_neus_jitter_z(const float* u, long R, int n, float* z, rb_stream_t stream);
int rb_ray_points(const float* o, const float* d, const float* z, long R, int n, float* pts, float* dirs,
                  rb_stream_t stream);
int rb_neus_upsample(const float* o, const float* d, const float* z, const float* sdf, long R, int n, int n_new,
                     float inv_s, float radius, const float* u, float* wtmp, float* z_new, rb_stream_t stream);
int rb_neus_merge(const float* z_old, const float* sdf_old, int n, const float* z_new, const float* sdf_new, int m,
                  long R, float* z_out, float* sdf_out, rb_stream_t stream);
int rb_neus_mid_z(const float* z, long R, int n, float sample_dist, float* zmid, rb_stream_t stream);
/* z / rays_d / sample_dist / cos_anneal: NULL z = stage-2 alpha from the neighbouring mid-point SDFs (model/sdf_render.py:
 * 206-218); z[R,n] (section starts) + rays_d[R,3] = stage-1 alpha, SDF extrapolated half a section along the ray with the
 * annealed cosine (neus/volume_render/sdf_render.py:172-190) -- for rendering directly from stage-1 checkpoints. */
/* Stage-2 weights from the mid-point SDFs alone (same expressions as rb_neus_finish: bit-identical weights); keep[j] = weights[j]
 * != 0, *count += kept.  Callers that do not need the eikonal term skip gradient + colour where the weight is exactly zero. */
int rb_neus_weights(const float* sdf, const float* pts, long R, int n, float inv_s, float radius, float* weights,
                    unsigned char* keep, unsigned long long* count, rb_stream_t stream);
int rb_neus_finish(const float* sdf, long sdf_stride, const float* color, const float* grad, const float* pts,
                   const float* zmid, const float* near, const float* far, long R, int n, float inv_s, float radius,
                   int white, const float* z, const float* rays_d, float sample_dist, float cos_anneal, float* rgb,
                   float* dist, float* acc, float* normal, float* weights, float* gerr, rb_stream_t stream);
int rb_surface_points(const float* p, const float* dir, const float* tk, long m, int ns, float* xs, rb_stream_t stream);
int rb_surface_finish(const float* sdf, const float* grad, const float* xs, const float* p, const float* pred_n, long m,
                      int ns, float s, float* x_out, float* n_out, float* gerr, rb_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * IDR sphere tracer (the use_octree=False ray tracer) -- RayTracing.forward / sphere_tracing / ray_sampler / secant in
 * eval mode (model/ray_tracing.py:26-297; get_sphere_intersection, utils/rend_util.py:141-163).  The host alternates
 * these per-ray state updates with rb_sdf_mlp (mode 0) on pts[2N,3] (start points, then end points).
 * state_f[6N] = acc_s, acc_e, cur_s, cur_e, nxt_s, nxt_e; state_b[4N] = un_s, un_e, bad_s, bad_e; ctrl int32[2]
 * (zero before op 0).  cam_stride 0 = one camera centre cam[3], 3 = one origin per ray cam[N,3].
 * op: 0 init (param = squared bounding-sphere radius) | 1 / 2 take sdf2[2N] for the unfinished / the
 * line-search rows | 3 loop top (param = sdf threshold) | 4 step | 5 back-step (param = (1-line_search_step)/2^k) |
 * 6 close iteration.  After the loop: hit = acc_s < acc_e, dist = acc_s, points = pts[:N]; rays with un_s set go
 * through the sampler: rb_raytrace_samples (z[m,n], P[m*n,3]), rb_sdf_mlp, rb_raytrace_pick (first negative sample,
 * minimal-SDF fallback for rays without surface, bracket[4m] = z_lo, z_hi, sdf_lo, sdf_hi), then n_rootfind_steps x
 * (rb_raytrace_secant, rb_sdf_mlp on pmid).
 * ------------------------------------------------------------------------------------------------------------ */
int rb_raytrace_step(int op, const float* cam, int cam_stride, const float* dirs, long N, float param, const float* sdf2, float* state_f,
                     unsigned char* state_b, float* pts, int* ctrl, rb_stream_t stream);
int rb_raytrace_samples(const float* cam, int cam_stride, const float* dirs, const float* lo, const float* hi, const float* lin, long m,
                        int n_steps, float* z, float* P, rb_stream_t stream);
int rb_raytrace_pick(const float* sdf, const float* z, const float* P, const unsigned char* obj, long m, int n,
                     float* out_pts, float* out_dist, unsigned char* out_hit, float* bracket, rb_stream_t stream);
int rb_raytrace_secant(const float* cam, int cam_stride, const float* dirs, const unsigned char* on, const float* smid, long m, int phase,
                       float* bracket, float* zp, float* pmid, rb_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Helper functions of the overlaid reference modules (ABI 8; robir_amd/csrc/surface.hip).  The fused forward never calls these;
 * they back the PUBLIC helper names a caller of the reference can import.
 * rb_pe_encode        PE.embed / Embedder.embed for any (input_dims d, bands): out[n, (include_input ? d : 0) + 2 d n_freq] =
 *                     [x | sin(x f_0) | cos(x f_0) | ...], freq[n_freq] DEVICE band values (model/neus_model.py:136-184,
 *                     model/embedder.py:7-38)
 * rb_expected_sin     y = exp(-var/2) sin(x), yvar = relu((1 - exp(-2 var) cos(2x))/2 - y^2) (yvar may be NULL), arguments wrapped
 *                     mod 100 pi beyond 100 pi (model/neus_model.py:14-24)
 * rb_tonemap_curve    the free functions of model/color_correction.py:31-73, NO clamp of t: curve 0 aces_fn(x), 1 aces_inv(x),
 *                     2 warp_aces_fn, 3 warp_aces_inv, 4 scale_aces_fn, 5 scale_aces_inv, 6 identity_fn, 7 ln_space_fn, 8 ln_space_inv;
 *                     t = shift[(i / width) * shift_stride] for element i (shift may be NULL for curves 0, 1, 6)
 * rb_sample_pdf       sample_pdf (model/sdf_render.py:37-67): bins[R,n], weights[R,n-1], u[n_s] (u_stride 0) or u[R,n_s]
 *                     (u_stride n_s, any order) -> samples[R,n_s]; cdf[R,n] is written as a by-product
 * rb_neus_core_aux    render_core's `dists`, `cdf` (= sigmoid(sdf inv_s)) and `inside_sphere` entries (model/sdf_render.py:186-225);
 *                     each output may be NULL
 * rb_sample_dirs      IDRNetwork.sample_dirs (model/implicit_differentiable_renderer.py:548-564), one row per (normal, theta, phi)
 * rb_intersect_sphere OctreeVisModel.intersect_sphere (model/octree_tracing.py:70-76)
 * ------------------------------------------------------------------------------------------------------------ */
int rb_pe_encode(const float* x, long n, int d, const float* freq, int n_freq, int include_input, float* out,
                 rb_stream_t stream);
int rb_expected_sin(const float* x, const float* var, long n, float* y, float* yvar, rb_stream_t stream);
int rb_tonemap_curve(const float* x, long n, int width, const float* shift, int shift_stride, int curve, float* y,
                     rb_stream_t stream);
int rb_sample_pdf(const float* bins, const float* weights, long R, int n, const float* u, long u_stride, int n_s,
                  float* cdf, float* samples, rb_stream_t stream);
int rb_neus_core_aux(const float* sdf, long sdf_stride, const float* pts, const float* z, long R, int n, float inv_s,
                     float radius, float sample_dist, float* dists, float* cdf, float* inside, rb_stream_t stream);
int rb_sample_dirs(const float* normals, const float* theta, const float* phi, long n, float* dirs, rb_stream_t stream);
int rb_intersect_sphere(const float* origins, const float* dirs, long n, float radius, float* out, rb_stream_t stream);

/* Host-only helper (no GPU work): one PIZ-compressed OpenEXR chunk -> 16-bit words, channel-major
 * ([channel][line][pixel][word]); chan = n_ch rows of (pixels per line, lines, words per pixel: 1 HALF, 2 FLOAT/UINT).
 * Used by robir_amd/exr.py to read the relighting environment maps that EnvmapMaterialNetwork.load_light
 * (model/sg_envmap_material.py:266-268) reads through imageio in the reference. */
int rb_exr_piz_decode(const unsigned char* src, long n_src, const int* chan, int n_ch, unsigned short* out, long n_out);

#ifdef __cplusplus
}
#endif
#ifdef RB_LEGACY        /* the legacy library is a superset: its translation units see both sets of prototypes */
#include "robir_hip_legacy.h"
#endif
#endif /* ROBIR_HIP_H */
