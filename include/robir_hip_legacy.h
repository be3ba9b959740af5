/* include/robir_hip_legacy.h -- the RETIRED entry points of librobir_hip (round 5).
 *
 * None of these is in the default library `robir_amd/librobir_hip.so` (ABI version 8, include/robir_hip.h) and no default precision
 * policy calls them.  `make -C robir_amd/csrc legacy` builds `robir_amd/librobir_hip_legacy.so`, a SUPERSET of the default library compiled
 * with -DRB_LEGACY, which exports them next to everything robir_hip.h declares; robir_amd/_lib.py loads it on demand
 * (`ROBIR_PRECISION=split`, the bit-identity tests that compare kernel generations, `ROBIR_SDF_FUSED_PE=0`).  What lives here:
 *   - the split-precision family of rounds 1-2 ((hi, lo) f16 operand pairs, 22 bits): k_dvis_v2 / k_dvis3_* (rb_dvis_fused_v2,
 *     rb_dvis_stream), k_sdf_ring / k_sdf_ring8 + k_sdf_back (rb_sdf_*ring*, rb_sdf_value_grad*), k_color_ring8, k_wide_ring, and their
 *     first generation mlp_kernels_h3.hip (rb_*_h3*), rb_pack_layer_h3; rb_dvis_fused with precision = 5;
 *   - round 3's one-tile exact-operand light-visibility kernel (rb_dvis_fused_x6; round 4's two-tile kernel replaced it);
 *   - the feature-row forms of the f32-input MFMA kernels (rb_feat_vis / _color / _color_tail, rb_vis_mlp, rb_sdf_mlp, rb_color_mlp,
 *     rb_illum_mlp, rb_linear_64_256): every network kernel of the default path takes points / directions and encodes them itself.
 * Reference lines each one mirrors: as stated for its successor in robir_hip.h. */
#ifndef ROBIR_HIP_LEGACY_H
#define ROBIR_HIP_LEGACY_H
#include "robir_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Second-generation kernel of the same stage (csrc/vis_diffuse_v2.hip): W49 = the three hidden layers and the 256->2 output
 * layer (rows padded to 16) packed by rb_pack_layer_h3 back to back = 49 chunks; two sample tiles per wave, one workgroup
 * per CU, output layer on the matrix pipe.  Same arguments and results (to fp32 rounding of the output layer). */
int rb_dvis_fused_v2(const float* normals, const int* chunk_id, long n, const float* A, const float* Bd, const float* dirs,
                     const float* wdir, const float* wsum, const float* W49, int L, int nsamp, int argmax_vis, int scale_log2,
                     float* vis_out, unsigned long long* eval_count, rb_stream_t stream);

/* Third generation of the same stage (csrc/vis_diffuse_v3.hip): three launches on `stream` --
 *   cull      one workgroup per point: n.d > 1e-6 survivors compacted into a global list of 16-sample tiles;
 *   stream    a PERSISTENT grid (n_workgroups; <= 0: one per CU) walks the tile list eight tiles per round, whatever point
 *             they belong to (balanced over the CUs for any n; the tile count is read from device memory: no host sync);
 *   reduce    one workgroup per point: SG-weighted mean per lobe in the fixed sample order.
 * Per pair the instruction sequence is that of rb_dvis_fused_v2: vis_out is bit-identical.  Caller-provided scratch (device):
 *   pair_j[n*L*nsamp] u16, pair_vis[n*L*nsamp] f32, tile_info[n*L*nsamp/16][2] i32, point_info[n][2] i32, counters[2] u64.
 * L*nsamp must be a multiple of 16. */
int rb_dvis_stream(const float* normals, const int* chunk_id, long n, const float* A, const float* Bd, const float* dirs,
                   const float* wdir, const float* wsum, const float* W49, int L, int nsamp, int argmax_vis, int scale_log2,
                   unsigned short* pair_j, float* pair_vis, int* tile_info, int* point_info, unsigned long long* counters,
                   int n_workgroups, float* vis_out, unsigned long long* eval_count, rb_stream_t stream);

/* Profiling aid: with RB_V2_TIMED=1 in the environment rb_dvis_fused_v2 runs an instrumented build that accumulates
 * shader-clock totals of wave 0 per phase (prologue, ring start, row gather, hidden layers, head, final reduction);
 * this call copies the six totals to out8[0..5] and clears them.  Returns non-zero on a HIP error. */
int rb_dvis_v2_debug(unsigned long long* out8);

/* The same stage with EXACT fp32 operands on the f16 matrix pipe ("f16x6", csrc/vis_diffuse_x6.hip): every operand as three
 * halves (h + m 2^-11 + l 2^-22 = the fp32 value exactly), the six partial products of weight >= 2^-22 in three fp32
 * accumulators by weight class -- not narrower than the reference's fp32 (VisNetwork, model/implicit_differentiable_renderer.py:
 * 241-258 evaluated by nn.Linear in fp32).  W49 = the same 49 chunks packed by rb_pack_layer_x6.  Same arguments as
 * rb_dvis_fused_v2; vis_out agrees with precision 0 of rb_dvis_fused to fp32 summation order. */
int rb_dvis_fused_x6(const float* normals, const int* chunk_id, long n, const float* A, const float* Bd, const float* dirs,
                     const float* wdir, const float* wsum, const float* W49, int L, int nsamp, int argmax_vis, int scale_log2,
                     float* vis_out, unsigned long long* eval_count, rb_stream_t stream);

/* Split-precision (f16x3) form of modes 0..3: Wp = the nine layers packed by rb_pack_layer_h3 with k_pad 64, 256, 256, 256,
 * 288 (skip layer: [208 | 64 | 16 zero slots]), 256 x4 and one scale 2^scale_log2. */
int rb_sdf_mlp_h3(const float* X, long M, const float* Wp, int mode, int scale_log2, float out_scale, float grad_scale,
                  float* out0, float* grad, rb_stream_t stream);

/* Split-precision (f16x3) form: Wp = the five layers packed by rb_pack_layer_h3 (first layer k_pad 320, same permutation). */
int rb_color_mlp_h3(const float* X, long M, const float* Wp, int scale_log2, float* rgb, rb_stream_t stream);

/* The same with the 48 encoded columns computed IN the kernel from x / view / normal [M,3] (embedview_fn fused into the network,
 * model/neus_model.py:535-545): no tail rows, no rb_feat_color_tail launch; bit-identical rgb. */
int rb_color_mlp_h3_points(const float* feat, long feat_stride, float feat_scale, const float* x, float x_scale, const float* view,
                           const float* normal, long M, const float* Wp, int scale_log2, float* rgb, rb_stream_t stream);

int rb_color_mlp_h3_two(const float* feat, long feat_stride, float feat_scale, const float* tail, long M, const float* Wp,
                        int scale_log2, float* rgb, rb_stream_t stream);

/* Same with split-precision (f16x3) layers: Wp = the five layers packed by rb_pack_layer_h3 with scale 2^scale_log2. */
int rb_vis_mlp_h3(const float* X, long M, const float* Wp, int scale_log2, float* logits, rb_stream_t stream);

int rb_vis_mlp_h3_points(const float* p, const float* d, long M, int rep, const float* Wp, int scale_log2, float* logits,
                         rb_stream_t stream);

/* Split-precision (f16x3) form of rb_illum_mlp (encoder = 0, raw[M,144]) and rb_ae_encode (encoder = 1, raw_latent[M,32]):
 * Wp = the five layers packed by rb_pack_layer_h3 with one scale. */
int rb_wide_mlp_h3(const float* X, long M, const float* Wp, int encoder, int scale_log2, float* Y, rb_stream_t stream);

int rb_wide_mlp_h3_points(const float* x, const float* extra, long M, const float* Wp, int encoder, int scale_log2, float* Y,
                          rb_stream_t stream);

/* Split-precision (f16x3) form: Wp = the nine layers packed by rb_pack_layer_h3 (skip layer k_pad 544). */
int rb_cesr_net_h3(const float* X, long M, int kind, int n_label, const float* Wp, int scale_log2, float* Y,
                   rb_stream_t stream);

int rb_cesr_net_h3_points(const float* x, long M, int kind, int n_label, const float* Wp, int scale_log2, float* Y, rb_stream_t stream);

/* Second generation of rb_sdf_mlp_h3 (csrc/sdf_ring.hip): same arguments, packed weights and results (to fp32 rounding); the net
 * is one cyclic chunk stream through an LDS-DMA ring, activations and hi/lo splits run between the MFMAs, workgroups are
 * persistent (n_workgroups <= 0: one per compute unit). */
int rb_sdf_mlp_ring(const float* X, long M, const float* Wp, int mode, int scale_log2, float out_scale, float grad_scale,
                    float* out0, float* grad, int n_workgroups, rb_stream_t stream);

/* The same two ops with the positional encoding FUSED into the network kernel (SDFNetwork.forward = embed_fn + layers,
 * model/neus_model.py:385-417; model/embedder.py:17-38): x [M,3] points, evaluated at x * in_scale -- no feature rows, no encoding
 * kernel.  The four lanes that share a point evaluate its 30 sine / cosine pairs between them once per round (the sincosf calls of
 * rb_feat_pe10), so outputs are bit-identical to the row forms above.
 *   rb_sdf_points_ring        mode 0 = signed distance [M], 1 = all 257 outputs [M,257]          (= rb_feat_pe10 + rb_sdf_mlp_ring)
 *   rb_sdf_points_ring_jvp    mode 2 / 3 = the same + the forward-mode gradient (small batches)   (= rb_feat_pe10(jvp) + rb_sdf_mlp_ring)
 *   rb_sdf_value_grad_points  all outputs + d sdf / dx; grad_scale multiplies the gradient           (= rb_feat_pe10 + rb_sdf_value_grad) */
int rb_sdf_points_ring(const float* x, long M, float in_scale, const float* Wp, int mode, int scale_log2, float out_scale,
                       float* out0, int n_workgroups, rb_stream_t stream);

int rb_sdf_points_ring_jvp(const float* x, long M, float in_scale, const float* Wp, int mode, int scale_log2, float out_scale,
                           float grad_scale, float* out0, float* grad, int n_workgroups, rb_stream_t stream);

/* Value rows (modes 0, 1 and the value pass of rb_sdf_value_grad) run as eight waves of one 16-row tile per workgroup -- two
 * waves per SIMD, csrc/sdf_ring8.hip -- or as four waves of two tiles (csrc/sdf_ring.hip): same results.  Selects 8 (default)
 * or 4 for the calling process; returns the previous setting. */
int rb_sdf_ring_waves(int waves);

int rb_sdf_value_grad(const float* X, long M, const float* Wp, const float* Wb, const float* w8row, int scale_log2,
                      float out_scale, float grad_scale, float* out0, float* grad, float* scratch, int n_workgroups,
                      rb_stream_t stream);

int rb_sdf_value_grad_points(const float* x, long M, float in_scale, const float* Wp, const float* Wb, const float* w8row,
                             int scale_log2, float out_scale, float grad_scale, float* out0, float* grad, float* scratch,
                             int n_workgroups, rb_stream_t stream);

/* All 257 outputs and the input gradient of the signed distance in REVERSE mode (csrc/sdf_back.hip; model/neus_model.py:440-452
 * is autograd too): one value pass (rb_sdf_mlp_ring's kernel, which also stores sigmoid(100 z) of every hidden pre-activation)
 * and one row vector per point back through the transposed layers -- twice the matrix work of the values instead of the four
 * times of the forward-mode rows of mode 3; same results to fp32 rounding.
 *   X [M,64] rb_feat_pe10 rows (value rows only);  Wp as rb_sdf_mlp_ring;  out0 [M,257], grad [M,3] as mode 3;
 *   Wb   the transposed layers packed by rb_pack_layer_h3 in the order W7^T, W6^T, W5^T, W4^T (320 x 256: rows 0..192 the
 *        columns of layer 3's outputs, 208..270 those of the skip features, the rest zero), W3^T (256 x 224), W2^T, W1^T,
 *        W0^T (64 x 256), followed by >= 2 KB of padding;  w8row [256] = row 0 of layer 8 (the distance output);
 *   scratch  rb_sdf_value_grad_scratch_floats(M) floats (8.5 KB per point: the sigmoid blob and the feature gradients).
 * Callers bound the scratch by evaluating large M in slabs. */
long rb_sdf_value_grad_scratch_floats(long M);

/* The same network on the eight-wave chunk-stream machine of rb_sdf_points_ring (csrc/color_ring8.hip: persistent workgroups, the
 * five layers as one cyclic stream of 65 chunks through an LDS-DMA ring, two waves per SIMD): bit-identical rgb, the default for
 * batches that fill the chip (n_workgroups <= 0: one workgroup per compute unit). */
int rb_color_ring_points(const float* feat, long feat_stride, float feat_scale, const float* x, float x_scale, const float* view,
                         const float* normal, long M, const float* Wp, int scale_log2, float* rgb, int n_workgroups,
                         rb_stream_t stream);

/* ... on feature rows X[M,64] (rb_feat_pe10 / rb_feat_ipe): the arguments and bits of rb_wide_mlp_h3 */
int rb_wide_mlp_ring(const float* X, long M, const float* Wp, int encoder, int scale_log2, float* Y, int n_workgroups, rb_stream_t stream);

int rb_wide_mlp_ring_points(const float* x, const float* extra, long M, const float* Wp, int encoder, int scale_log2, float* Y,
                            int n_workgroups, rb_stream_t stream);

/* The split-precision 512-wide nets on the chunk-stream machine (csrc/wide_ring.h: persistent workgroups of four waves, the net as one
 * cyclic stream of 16-neuron chunks through an LDS-DMA ring, activation + hi/lo split between the next chunk's MFMAs): the same
 * arguments and bit-identical outputs as rb_cesr_net_h3_points / rb_wide_mlp_h3_points, the default for batches that fill the chip.
 * n_workgroups <= 0: one workgroup per compute unit. */
int rb_cesr_net_ring_points(const float* x, long M, int kind, int n_label, const float* Wp, int scale_log2, float* Y, int n_workgroups,
                            rb_stream_t stream);

/* split-precision packing: weights (and bias) scaled by 2^scale_log2, stored as hi/lo half pairs; k_pad % 32 == 0;
 * same size as the fp32 packing. */
int rb_pack_layer_h3(const float* W, const float* b, int n_out, int k_in, int n_pad, int k_pad, const int* perm,
                     int scale_log2, float* out, rb_stream_t stream);

/* X[M,128] = [PE10(p[i/rep]) | PE10(d[i]) | 0 0]   VisNetwork input; p holds M/rep points, d holds M directions */
int rb_feat_vis(const float* p, const float* d, long M, int rep, float* X, rb_stream_t stream);

/* X[M,304] = [feat[M,256 @feat_stride]*feat_scale | x*x_scale | PE4(view) | normal | 0 x15]   colour-net input */
int rb_feat_color(const float* x, float x_scale, const float* view, const float* normal, const float* feat,
                  long feat_stride, float feat_scale, long M, float* X, rb_stream_t stream);

/* The same net reading its 304 input columns from two places: 0..255 = feat[i*feat_stride + 0..255] * feat_scale (the SDF net's
 * output rows; 4-byte alignment suffices), 256..303 = tail[i*48 + 0..47] written by rb_feat_color_tail ([x*x_scale | PE4(view) |
 * normal | 0 x15]) -- no assembled [M,304] rows.  Results equal rb_feat_color + rb_color_mlp_h3 bit for bit. */
int rb_feat_color_tail(const float* x, float x_scale, const float* view, const float* normal, long M, float* tail, rb_stream_t stream);

/* VisNetwork.forward (implicit_differentiable_renderer.py:250-258): X[M,128] -> logits[M,2].
 * Wp: packed [128->256, 256->256 x3, 256->16]. */
int rb_vis_mlp(const float* X, long M, const float* Wp, float* logits, rb_stream_t stream);

/* One linear layer X[M,64] -> Y[M,256] (packed 64->256); used to factor the visibility net's first layer. */
int rb_linear_64_256(const float* X, long M, const float* Wp, float* Y, rb_stream_t stream);

int rb_sdf_mlp(const float* X, long M, const float* Wp, int mode, float out_scale, float grad_scale, float* out0,
               float* grad, rb_stream_t stream);

/* NeuS RenderingNetwork.forward (model/neus_model.py:535-560): X[M,304] -> rgb[M,3] (sigmoid applied).
 * Wp packed [304->256 (columns permuted to the rb_feat_color order), 256->256 x3, 256->16]. */
int rb_color_mlp(const float* X, long M, const float* Wp, float* rgb, rb_stream_t stream);

/* IndirctIllumNetwork.lobe_layer (implicit_differentiable_renderer.py:186-193,206): X[M,64] -> raw[M,144].
 * Wp packed [64->512, 512->512 x3, 512->144]. */
int rb_illum_mlp(const float* X, long M, const float* Wp, float* raw, rb_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ROBIR_HIP_LEGACY_H */
