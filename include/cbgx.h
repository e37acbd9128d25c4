/*
 * cbgx.h -- C ABI of libcbgx.so: the MI355X (gfx950) implementation of CBGBench's
 * per-diffusion-step E(3)-equivariant message passing.
 *
 * The reference is pure Python; the natives it calls on this path are the
 * un-vendored torch_cluster / torch_scatter wheels and ATen.  Each entry point
 * below names the reference interface it replaces (paths relative to the
 * reference tree).  All pointers are DEVICE pointers unless said otherwise, all
 * float tensors are contiguous fp32 row-major, `stream` is a hipStream_t passed
 * as void* (NULL = default stream).  Every call is asynchronous on `stream`,
 * re-entrant, keeps no pointer after it returns and never frees or allocates
 * device memory: the caller owns inputs, outputs and the workspace.
 *
 * The only state the library owns, besides the optional profiling hook at the end of this file:
 * one auxiliary non-blocking HIP stream and two events per HOST THREAD AND CALLER STREAM (at most eight per thread, created on the
 * first multi-layer forward on that stream; a ninth caller stream takes over the oldest entry's stream and events -- nothing is
 * destroyed or waited for, so this is legal under stream capture; results stay correct, only the overlap of that caller's node
 * stages with its other calls is lost), and the process-wide workgroup limit of cbgx_set_edge_workgroups (a relaxed atomic int:
 * the one value shared between host threads).  cbgx_unitransformer_forward{,_cached} fork the node stage of
 * layer l+1 onto it next to the h2x block of layer l and join it back before returning work to `stream`, so
 * from the caller's point of view everything is still ordered on `stream` (hipGraph capture of `stream`
 * records the fork/join).  One device per host thread (the one-process-per-GPU model): a thread that switches
 * devices falls back to the serial schedule.  CBGX_OVERLAP=0 in the environment disables the auxiliary stream.
 *
 * Return value: 0 on success, negative CBGX_E_* on failure (never abort());
 * cbgx_last_error() gives a thread-local message.
 *
 * Arithmetic contract: inputs, outputs and accumulators are fp32.  The node projection, the query MLP's second Linear and the
 * radial-basis part of the edge pre-activation run on the f16 matrix pipe as "split-f16" products (every fp32 operand v is carried
 * as hi = f16(v 2^k), lo = f16(v 2^k - hi) with 2^k an exact power of two chosen per weight column / per table at pack time and per
 * row of activations at run time; three f16 products hi hi + hi lo + lo hi, fp32 accumulation, 2^-k applied to the fp32 result),
 * everything else on fp32 MFMA / VALU.  The result is fp32-grade over the range the scale exponents cover: error <= ~2^-21 of
 * sum |a||b| per dot product, measured 5 - 8e-8 against 1.3 - 2.5e-7 for a plain fp32 FMA chain over weight scales 1e-4 .. 30 and
 * activations 1e-3 .. 1e5 (tests/test_splitf16_range.py, tests/test_gpu_range.py; |h| > 65 504 is fine).  That is the supported and
 * tested range.  The scale exponents are clamped so that every factor stays a normal fp32 number (rbf tables: 2^-40 .. 2^20, node
 * tables 2^-100 .. 2^60, activation rows 2^-100 .. 2^100); outside of what they can compensate (rbf columns above 2^54, rows whose
 * largest entry is denormal) accuracy degrades gradually towards plain f16 -- never an error code, and finite inputs never give a
 * non-finite result because of the scaling.
 * Non-finite inputs propagate as in the reference.
 *
 * Node order (as produced by compose_context, repo/modules/common.py:189-214):
 * nodes sorted by graph; graph g owns rows [graph_ptr[g], graph_ptr[g+1]).
 */
#ifndef CBGX_H
#define CBGX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: the packed-weight layout carries the power-of-two scales of the split-f16 tables (range-safe arithmetic, below); a blob
 * packed by a version-1 library is not understood by version 2 and vice versa -- re-pack with the library that consumes it.
 * 4 (round 5): cbgx_unitransformer_backward with grad_h_out == NULL also prunes the classifier head's backward to the ligand rows
 * (round 4 changed that without a bump), the cbgx_targetdiff_train_noise / _loss / _loss_backward exports exist, and the workspace
 * layout of cbgx_unitransformer_forward changed: a caller built against version 3 must not load this library. */
#define CBGX_ABI_VERSION 6

#define CBGX_OK 0
#define CBGX_E_INVALID (-1)   /* bad argument (shape, NULL pointer, unsupported hyper-parameter) */
/* Unsupported hyper-parameters -- the encoder options of unitransformer.py:17-39 that no shipped config sets and this library does not
 * implement: num_x2h / num_h2x != 1, num_blocks != 1, ew_type in {'r', 'm'} (x2h_attention.py:70-76; the 'r' branch of
 * h2x_attention.py:54 is broken in the reference), cutoff_mode in {'radius', 'hybrid'} (unitransformer.py:77,83 reference undefined
 * names there), x2h_out_fc=True (x2h_attention.py:93-94), n_heads != 16, node_feat_dim != 128, k != 32, num_r_gaussian != 20,
 * act_fn != 'relu', norm=False, and a `time:` embedding (context_emb.py:190-195).  The host mirror raises ValueError naming the option
 * (cbgbench_amd/unitransformer.py; texts pinned by tests/test_host.py), the C entry points return CBGX_E_INVALID. */
#define CBGX_E_WORKSPACE (-2) /* workspace too small */
#define CBGX_E_HIP (-3)       /* a HIP runtime call / kernel launch failed */

/* Fixed hyper-parameters of every shipped diffusion config
 * (configs/{denovo,linker,frag,scaffold,sidechain}/train/{targetdiff,diffbp,diffsbdd}.yml:3-7 and the
 * defaults at repo/modules/e3nn/unitransformer.py:17-39). The kernels are specialised for them. */
#define CBGX_HIDDEN 128
#define CBGX_HEADS 16
#define CBGX_NUM_GAUSSIANS 20
#define CBGX_KNN 32
#define CBGX_EDGE_TYPES 4
#define CBGX_GATE_HIDDEN 160

int cbgx_abi_version(void);
const char *cbgx_last_error(void);

/* ---- scheduling hint (changes no result) -----------------------------------------------------------------------
 * The fused x2h edge kernel is persistent: one 8-wave workgroup per CU, all registers and 150 KB of LDS, so nothing else runs
 * beside it, while the node kernels of the same call are HBM-bound and leave the matrix cores idle.  A caller that keeps TWO
 * forward calls in flight on two of its streams (two resident batches; every call gets its own auxiliary stream) can let the
 * node kernels of one call run under the edge kernel of the other by leaving a few CUs free: n = upper bound of the edge
 * kernel's workgroups (a multiple of 8; e.g. 224 of 256), 0 = default (one per CU).  Process-wide; returns the previous value.
 * cbgbench_amd.TargetDiff.sample_many / bench.py --streams 2 use it. */
int cbgx_set_edge_workgroups(int n);

/* ---- weights -------------------------------------------------------------------------------
 * The library consumes one packed fp32 blob built from the reference state_dict tensors
 * (SURVEY.md A.2 key names).  `tensors` is a HOST array of DEVICE pointers in this order:
 *   [0..5]   denoiser.dist_emb.1.net.{0.weight,0.bias,1.weight,1.bias,3.weight,3.bias}
 *   then for each layer l, 36 pointers:
 *     x2h_layers.0.hk_func.net.{0.weight,0.bias,1.weight,1.bias,3.weight,3.bias}, hv_func (6), hq_func (6),
 *     h2x_layers.0.xk_func (6), xv_func (6), xq_func (6)
 *   then classifier.{0.weight,0.bias,2.weight,2.bias}
 * i.e. 6 + 36*num_layers + 4 pointers.  Re-pack whenever the parameters change. */
size_t cbgx_packed_weights_floats(int num_layers, int num_classes);
int cbgx_pack_weights(const float *const *tensors, int num_tensors, int num_layers, int num_classes,
                      float *packed, void *stream);

/* ---- whole denoiser call ---------------------------------------------------------------------
 * Replaces UniTransformer.forward (repo/modules/e3nn/unitransformer.py:102-123) with
 * cutoff_mode='knn', k=32, ew_type='global', num_blocks=1, num_x2h=num_h2x=1, relu+LayerNorm.
 * x[N,3], h[N,128], graph_ptr[B+1] int32, lig_flag[N]/gen_flag[N] uint8 (0/1).
 * Outputs x_out[N,3], h_out[N,128], logits[N,num_classes] (logits may be NULL).
 * h_out may be NULL when the caller only consumes x_out and the logits of ligand rows (what the samplers do,
 * targetdiff.py:164-165): the last two x2h blocks are then restricted to the nodes whose features can still reach
 * those outputs (receptive-field pruning) and logits are defined on lig_flag rows only. */
size_t cbgx_workspace_bytes(int n_nodes, int n_graphs);
int cbgx_unitransformer_forward(const float *packed, int num_layers, int num_classes,
                                const float *x, const float *h, const int32_t *graph_ptr,
                                const uint8_t *lig_flag, const uint8_t *gen_flag,
                                int n_nodes, int n_graphs,
                                float *x_out, float *h_out, float *logits,
                                void *workspace, size_t workspace_bytes, void *stream);

/* Same call with a *static-context cache* for samplers, where the protein half of (x, h) is identical in every one of
 * the T denoiser calls of a run (protein atoms never move, unitransformer.py:182; no time embedding in the shipped
 * configs).  static_h1 / static_h2 [N,128]: the features leaving layer 0 / layer 1 when the same call is made on the
 * ligand-free pockets (rows of ligand atoms are ignored) -- obtain them with two cbgx_unitransformer_forward calls of
 * num_layers = 1 and 2 on the protein rows alone and scatter to the composed row order.  A protein node whose 32
 * neighbours contain no ligand atom sees exactly the ligand-free pocket in layer 0 (and, one hop further, in layer 1),
 * so those rows are copied from the cache and the first two x2h blocks run only on the rows that differ.
 * Optional graph part (all four or none; NULL = recompute): static_nbr [N,32] / static_deg [N] / static_ew [N,32] = the
 * ligand-free pockets' cbgx_knn_graph lists (in composed row numbers) and cbgx_edge_gate values, static_r32sq [N] = squared
 * distance to the last (32nd) of those neighbours, +inf where deg < 32.  A protein node whose nearest ligand atom is not
 * closer than that keeps its cached list and gate; only the others get fresh ones.  static_nbr rows must be what cbgx_knn_graph
 * writes: ids of nodes of the SAME graph in the first static_deg[i] slots, -1 in the rest (an entry outside its graph is treated as
 * padding by the list kernels; it is never dereferenced).
 * Results are bit-identical to cbgx_unitransformer_forward.  Requires num_layers >= 4 (otherwise the cache is ignored).
 * flags: CBGX_FWD_H_ON_SOURCES -- the caller reads h_out only on the rows A1 = gen_flag | lig_flag | in-neighbours of gen_flag rows
 *   (what an H2X stack run on the same coordinates reads: DiffBP's CoMPredictor, diffbp.py:79-101; pass the same rows to
 *   cbgx_h2x_stack_forward).  The last two x2h blocks are then pruned to the receptive field of those rows exactly as when
 *   h_out is NULL; rows of h_out outside A1 are left unwritten.  Without the flag h_out is defined on every row (no pruning). */
#define CBGX_FWD_H_ON_SOURCES 1u
int cbgx_unitransformer_forward_cached(const float *packed, int num_layers, int num_classes,
                                       const float *x, const float *h, const int32_t *graph_ptr,
                                       const uint8_t *lig_flag, const uint8_t *gen_flag, int n_nodes, int n_graphs,
                                       const float *static_h1, const float *static_h2,
                                       const int32_t *static_nbr, const int32_t *static_deg,
                                       const float *static_ew, const float *static_r32sq,
                                       float *x_out, float *h_out, float *logits, unsigned flags,
                                       void *workspace, size_t workspace_bytes, void *stream);

/* ---- stages (also what the parity tests call one by one) ------------------------------------- */

/* torch_cluster.knn_graph(x, k, batch, loop=False, flow='source_to_target') as called at
 * unitransformer.py:80.  nbr[N,32] int32: neighbours of centre i in ascending (squared distance,
 * index) order, -1 padded; deg[N] = min(k, n_graph-1).  Edge (src=nbr[i][s], dst=i). k must be 32. */
int cbgx_knn_graph(const float *x, const int32_t *graph_ptr, int n_graphs, int n_nodes, int k,
                   int32_t *nbr, int32_t *deg, void *stream);

/* Global distance gate, unitransformer.py:109-112 + repo/modules/embs/dist_emb.py:6-9:
 * e_w[N,32] = sigmoid(MLP_{20->160->1}(rbf(|x_i - x_nbr|))), 0 in padded slots. */
int cbgx_edge_gate(const float *packed, const float *x, const int32_t *nbr, const int32_t *deg,
                   int n_nodes, float *e_w, void *stream);

/* X2HAttention.forward, repo/modules/attention/x2h_attention.py:43-97 (includes the residual):
 * h_out = h + sum_e softmax_e(q_i.k_e/sqrt(8)) * v_e * e_w.  `layer` selects denoiser.blocks[layer]. */
int cbgx_x2h_attention(const float *packed, int layer, const float *x, const float *h,
                       const int32_t *nbr, const int32_t *deg, const uint8_t *lig_flag, const float *e_w,
                       int n_nodes, float *h_out, void *workspace, size_t workspace_bytes, void *stream);

/* H2XAttention.forward, repo/modules/attention/h2x_attention.py:34-73, plus the masked update of
 * E3DualAttentionLayer.forward (unitransformer.py:178-184): x_out = x + delta_x * gen_flag.
 * Only gen_flag nodes are computed (the others cannot move); delta_x[N,3] (may be NULL) receives the attention
 * output for gen_flag nodes and 0 elsewhere. */
int cbgx_h2x_attention(const float *packed, int layer, const float *x, const float *h,
                       const int32_t *nbr, const int32_t *deg, const uint8_t *lig_flag,
                       const uint8_t *gen_flag, const float *e_w, int n_nodes,
                       float *x_out, float *delta_x, void *workspace, size_t workspace_bytes, void *stream);

/* classifier head, unitransformer.py:46-51,119-120: Linear -> softplus - ln2 -> Linear. */
int cbgx_classifier(const float *packed, int num_layers, int num_classes, const float *h, int n_nodes,
                    float *logits, void *workspace, size_t workspace_bytes, void *stream);

/* ---- a stack of H2X blocks on its own graph: DiffBP's CoMPredictor -----------------------------------
 * CoMPredictor.forward (repo/models/diffusion/diffbp.py:79-101): kNN graph + gate from x, then num_layers
 * H2XAttention blocks with x_out = x_out + delta_x * gen_flag, h fixed.  `tensors` for the pack: HOST array of
 * DEVICE pointers com_head.dist_emb.1.net.{0.weight,0.bias,1.weight,1.bias,3.weight,3.bias} followed, per layer,
 * by com_head.h2xattentions.{l}.{xk_func,xv_func,xq_func}.net.{0.weight,...,3.bias} (6 + 18*num_layers). */
size_t cbgx_packed_h2x_stack_floats(int num_layers);
int cbgx_pack_h2x_stack(const float *const *tensors, int num_tensors, int num_layers, float *packed, void *stream);
int cbgx_h2x_stack_forward(const float *packed, int num_layers, const float *x, const float *h,
                           const int32_t *graph_ptr, const uint8_t *lig_flag, const uint8_t *gen_flag,
                           int n_nodes, int n_graphs, float *x_out, void *workspace, size_t workspace_bytes,
                           void *stream);

/* ---- TargetDiff step prologue / epilogue ----------------------------------------------------------
 * The per-step work around the denoiser in TargetDiff.sample (repo/models/diffusion/targetdiff.py:150-182).
 * prologue: x[lig_rows[a]] = x_lig[a];  h[lig_rows[a]] = ligand_atom_emb(c_lig[a]) + ligand_indicator(1)
 *           (PLContextEmbedder.forward, repo/modules/context_emb.py:179-230; no time embedding in shipped configs).
 *           lig_emb_w [128,C], lig_emb_b [128], ind_w [128,1], ind_b [128] are the reference parameters
 *           context_embedder.{ligand_atom_emb,ligand_indicator}.{weight,bias}; c_lig [n_lig,C] float (one-hot).
 * epilogue: x_next = posterior sample of positions (CTNVPScheduler.backward_remove_noise 'denoise',
 *           diffusion_scheduler.py:144-165), c_next / v_next = posterior sample of atom types by Gumbel-argmax
 *           (TypeVPScheduler.backward_remove_noise, :367-378, 407-441; categorical.py:26-32); atoms with
 *           gen_lig == 0 keep position and type.  x_den [N,3] / logits [N,C] are the denoiser outputs, read at
 *           lig_rows.  `tables` = HOST array of 7 DEVICE pointers into the reference's frozen schedule tables:
 *           pos_scheduler.{posterior_mean_c0_coef, posterior_mean_ct_coef, posterior_logvar},
 *           type_scheduler.{log_alphas_v, log_one_minus_alphas_v, log_alphas_cumprod_v,
 *           log_one_minus_alphas_cumprod_v}.  t is the (batch-uniform) step index; eps ~ N(0,1) [n_lig,3] and
 *           u ~ U(0,1) [n_lig,C] are the noise draws (the reference draws randn_like then rand_like). */
int cbgx_targetdiff_prologue(const float *x_lig, const float *c_lig, const int32_t *lig_rows, int n_lig,
                             int num_classes, const float *lig_emb_w, const float *lig_emb_b, const float *ind_w,
                             const float *ind_b, float *x, float *h, void *stream);
int cbgx_targetdiff_epilogue(const float *x_den, const float *logits, const int32_t *lig_rows, const float *x_lig,
                             const float *c_lig, const uint8_t *gen_lig, int n_lig, int num_classes, int t,
                             int num_timesteps, const float *const *tables, const float *eps, const float *u,
                             float *x_next, float *c_next, int32_t *v_next, void *stream);

/* Epilogue of step t and prologue of step t - 1 in one launch (round 5): cbgx_targetdiff_epilogue's outputs x_next / c_next, and
 * in the same launch the composed rows of the NEXT denoiser call -- x[lig_rows] = x_next, h[lig_rows] = ligand_atom_emb(c_next) +
 * ligand_indicator(1) -- exactly what cbgx_targetdiff_prologue would write from (x_next, c_next).  Same arithmetic, bit for bit,
 * as the two calls it replaces (targetdiff.py:164-182 followed by :155-158 of the next iteration); x_den must not alias x. */
int cbgx_targetdiff_step_boundary(const float *x_den, const float *logits, const int32_t *lig_rows, const float *x_lig,
                                  const float *c_lig, const uint8_t *gen_lig, int n_lig, int num_classes, int t,
                                  int num_timesteps, const float *const *tables, const float *eps, const float *u,
                                  float *x_next, float *c_next, const float *lig_emb_w, const float *lig_emb_b,
                                  const float *ind_w, const float *ind_b, float *x, float *h, void *stream);

/* Trajectory-resident variants: the ligand state lives in traj_x [T+1, n_lig, 3] / traj_c [T+1, n_lig, C] (slot s+1 = the
 * state entering step s, slot 0 = the final state) and the step index in a device int (*t_dev).  The prologue reads slot
 * *t_dev + 1; the epilogue reads slot *t_dev + 1, writes slot *t_dev and then decrements *t_dev.  No argument changes
 * from step to step, so one captured hipGraph of {prologue_traj, forward, noise draw, epilogue_traj} can be replayed for
 * all T steps -- the launch-bound regime of small batches (TargetDiff.sample(..., use_graph=True)). */
int cbgx_targetdiff_prologue_traj(const float *traj_x, const float *traj_c, const int32_t *t_dev,
                                  const int32_t *lig_rows, int n_lig, int num_classes, const float *lig_emb_w,
                                  const float *lig_emb_b, const float *ind_w, const float *ind_b, float *x, float *h,
                                  void *stream);
int cbgx_targetdiff_epilogue_traj(const float *x_den, const float *logits, const int32_t *lig_rows, float *traj_x,
                                  float *traj_c, const uint8_t *gen_lig, int n_lig, int num_classes, int32_t *t_dev,
                                  const float *const *tables, const float *eps, const float *u, void *stream);

/* ---- DiffBP / DiffSBDD: the per-step arithmetic around the network calls, one launch per step -------------------
 * cbgx_diffbp_epilogue: diffbp.py:262-297 after the denoiser and the CoMPredictor stack -- zero-COM noise estimate,
 *   mean shift of the H2X stack (CoMPredictor.forward, diffbp.py:79-101), score step on the positions
 *   (CTNVPScheduler.backward_remove_noise(type='score'), diffusion_scheduler.py:154-158) and the absorbing-state type
 *   step (MaskTypeSchedule.backward_remove_noise, :475-496).  x_den / x_com / x_in: [N,3] denoiser output, H2X-stack
 *   output and their common input; lig_ptr [B+1]: ligand atoms of graph g are [lig_ptr[g], lig_ptr[g+1]) (ligand arrays are
 *   sorted by graph); tables: alphas_cumprod [T], betas [T] of the position scheduler; eps [n_lig,3], u [n_lig].
 * cbgx_diffsbdd_step: one iteration of diffsbdd.py:296-304 after the denoiser (sample_p_zs_given_zt for positions with
 *   the ligand's centre of mass removed and the pocket translated with it, and for types; diffusion_scheduler.py:1012-1040).
 *   inv_alpha, coef, sigma: 1 / alpha_ts, sigma2_ts / alpha_ts / sigma_t and sigma_ts sigma_s / sigma_t of the step.  Updates
 *   the composed x IN PLACE (pocket rows of every graph, ligand rows) and writes h on ligand rows from the new types, so
 *   that the next denoiser call needs no prologue; shift [B,3] (may be NULL) receives the removed mean per graph.
 *   frame_shift [B,3] (in / out, may be NULL) selects the FRAMED mode: the pocket rows of x are left where they are (so the
 *   static-context cache of cbgx_unitransformer_forward_cached holds for the whole run) and the translation the reference applies
 *   to the pocket is accumulated in frame_shift instead: true position = position in x - frame_shift[graph].  x_den is then the
 *   denoiser's output in the frame of x (the network is translation-equivariant), x_lig / x_next stay in true coordinates, the
 *   ligand rows of x get x_next + frame_shift.  Start with frame_shift = 0. */
int cbgx_diffbp_epilogue(const float *x_den, const float *x_com, const float *x_in, const float *logits,
                         const int32_t *lig_rows, const int32_t *lig_ptr, const float *x_lig, const float *c_lig,
                         const uint8_t *gen_lig, int n_lig, int n_graphs, int num_classes, int t, int num_timesteps,
                         const float *alphas_cumprod, const float *betas, int absorbing_state, const float *eps,
                         const float *u, float *x_next, float *c_next, void *stream);
int cbgx_diffsbdd_step(const float *x_den, const float *logits, const int32_t *graph_ptr, const int32_t *lig_rows,
                       const int32_t *lig_ptr, const uint8_t *lig_flag, const float *x_lig, const float *c_lig, int n_lig,
                       int n_graphs, int num_classes, float inv_alpha, float coef, float sigma, int update_positions,
                       int update_types, const float *eps_x, const float *eps_c, const float *lig_emb_w,
                       const float *lig_emb_b, const float *ind_w, const float *ind_b, float *x_next, float *c_next,
                       float *x, float *h, float *shift, float *frame_shift, void *stream);

/* ---- training: taped forward and backward -----------------------------------------------------------
 * train.py:185-189 runs `loss_dict, _ = model(batch); loss.backward()`; autograd walks UniTransformer.forward
 * (unitransformer.py:102-123) backwards through every X2HAttention / H2XAttention (x2h_attention.py:43-97,
 * h2x_attention.py:34-73), the gate and the classifier.  libcbgx replaces that pair with
 *   cbgx_unitransformer_forward_train : same outputs as cbgx_unitransformer_forward and a *tape*
 *       (caller-owned, cbgx_train_tape_bytes) holding the kNN lists, the gate, the per-layer x / h inputs and the node
 *       stage of every block (projection [N,640] + folded query [N,16,128]: 177 KB per node and layer).  h_out != NULL: no
 *       pruning.  h_out == NULL (round 6): the caller promises that its loss reads x_out on gen_flag rows and logits on
 *       lig_flag rows only -- and will call the backward with grad_h_out == NULL; the last two x2h blocks, the neighbour
 *       projections of the h2x blocks and the classifier then run on the receptive field of those rows, as in the sampling
 *       forward (rows of `logits` outside gen | lig | nbr(gen) are not written);
 *   cbgx_unitransformer_backward      : given dL/dx_out [N,3], dL/dh_out [N,128], dL/dlogits [N,C] (each may be
 *       NULL = zero) it recomputes the per-edge intermediates block by block (nothing per-edge is ever stored) and
 *       writes dL/dh_in [N,128] (may be NULL) and the gradient of every parameter tensor: `grads` is a HOST array
 *       of DEVICE pointers, same order and shapes as the `tensors` of cbgx_pack_weights (6 + 36*L + 4), each
 *       OVERWRITTEN.  No gradient is produced for the input coordinates (they are data: targetdiff.py:87-101).
 *       PRECONDITION of grad_h_out == NULL: the caller's loss reads x_out on gen_flag rows and logits on lig_flag rows only
 *       (what TargetDiff.get_loss / DiffSBDD.get_loss do, targetdiff.py:103-121); the backward of the last x2h blocks is
 *       then pruned to the receptive field of those rows, the classifier head's backward walks the lig_flag rows only, and
 *       gradient entries on other rows are ignored.  A loss that touches other rows must pass a (possibly all-zero) grad_h_out
 *       (and must have run the forward with h_out != NULL): the backward is then exact for ANY loss -- since round 6 it still
 *       prunes, but around the support of the caller's gradients, which it finds itself (the rows of grad_h_out / grad_logits
 *       with a non-zero entry are marked on the device and join the seed of the receptive field; DiffBP's centre-of-mass head
 *       reads h_out on the movable atoms and their neighbours only: + 9.6 % on its training step).
 * Both take the larger training workspace (cbgx_train_workspace_bytes).  Neighbour-row gradients are accumulated with
 * fp32 atomics, so results are reproducible only up to summation order (as with the reference's torch_scatter on GPU).
 * CBGX_BX_EDGE_ROWS=1 in the environment (read at every backward call; ABI unchanged, the workspace always holds the buffers:
 * 32 KB per node) selects the edge-row mode of the x2h blocks instead: every edge's contribution goes to its own row and the rows of
 * every source node are summed in a fixed order (csrc/train_scatter.hip) -- dL/dh then has the same bits in every run, the
 * coordinate gradient still uses atomics, and a training step is 1 - 3 % slower.  Same mathematics, same tolerances in the tests. */
/* cbgx_unitransformer_forward_train_ex (ABI 5): the same with `flags`; CBGX_FWD_H_ON_SOURCES = h_out is wanted on
 * A1 = gen_flag | lig_flag | in-neighbours of gen_flag rows only (zero elsewhere) -- the taped forward prunes as with h_out == NULL,
 * and the caller's dL/dh_out must vanish outside A1 (DiffBP: the centre-of-mass head reads h_out on the movable atoms and their
 * neighbours in the same k-nearest-neighbour graph). */
size_t cbgx_train_tape_bytes(int n_nodes, int num_layers);
size_t cbgx_train_workspace_bytes(int n_nodes);
int cbgx_unitransformer_forward_train(const float *packed, int num_layers, int num_classes,
                                      const float *x, const float *h, const int32_t *graph_ptr,
                                      const uint8_t *lig_flag, const uint8_t *gen_flag, int n_nodes, int n_graphs,
                                      float *x_out, float *h_out, float *logits, void *tape, size_t tape_bytes,
                                      void *workspace, size_t workspace_bytes, void *stream);
int cbgx_unitransformer_backward(const float *packed, int num_layers, int num_classes, const void *tape,
                                 size_t tape_bytes, const uint8_t *lig_flag, const uint8_t *gen_flag, int n_nodes,
                                 const float *grad_x_out, const float *grad_h_out, const float *grad_logits,
                                 float *const *grads, int num_grads, float *grad_h_in, void *workspace,
                                 size_t workspace_bytes, void *stream);
/* Backward of one attention block (stage-level parity tests).  Inputs as the forward stage; `grads` = 18 DEVICE
 * pointers {hk,hv,hq}_func (x2h) / {xk,xv,xq}_func (h2x) x net.{0.weight,0.bias,1.weight,1.bias,3.weight,3.bias}.
 * x2h: grad_h includes the residual path.  h2x: grad_x includes the identity path of x_out = x + delta_x * gen_flag.
 * grad_e_w [N,32] is the gradient with respect to the gate values. */
int cbgx_unitransformer_forward_train_ex(const float *packed, int num_layers, int num_classes, const float *x, const float *h,
                                         const int32_t *graph_ptr, const uint8_t *lig_flag, const uint8_t *gen_flag,
                                         int n_nodes, int n_graphs, float *x_out, float *h_out, float *logits, unsigned flags,
                                         void *tape, size_t tape_bytes, void *workspace, size_t workspace_bytes, void *stream);
int cbgx_x2h_attention_backward(const float *packed, int layer, const float *x, const float *h,
                                const int32_t *nbr, const int32_t *deg, const uint8_t *lig_flag, const float *e_w,
                                int n_nodes, const float *grad_h_out, float *grad_h, float *grad_x, float *grad_e_w,
                                float *const *grads, void *workspace, size_t workspace_bytes, void *stream);
int cbgx_h2x_attention_backward(const float *packed, int layer, const float *x, const float *h,
                                const int32_t *nbr, const int32_t *deg, const uint8_t *lig_flag,
                                const uint8_t *gen_flag, const float *e_w, int n_nodes, const float *grad_x_out,
                                float *grad_h, float *grad_x, float *grad_e_w, float *const *grads, void *workspace,
                                size_t workspace_bytes, void *stream);

/* The same pair for a stack of H2X blocks on its own graph: DiffBP's CoMPredictor inside `DiffBP.get_loss`
 * (repo/models/diffusion/diffbp.py:79-101, 195-198).  The tape holds the stack's kNN lists, gate and per-layer
 * coordinates; h is the same tensor in every layer.  backward: grad_x_out [N,3] -> grad_h [N,128] (overwritten) and
 * `grads` = 6 + 18*L DEVICE pointers in the order of cbgx_pack_h2x_stack's `tensors`, each overwritten.  No gradient for the
 * input coordinates (they are the noised data). */
size_t cbgx_h2x_stack_tape_bytes(int n_nodes, int num_layers);
int cbgx_h2x_stack_forward_train(const float *packed, int num_layers, const float *x, const float *h,
                                 const int32_t *graph_ptr, const uint8_t *lig_flag, const uint8_t *gen_flag,
                                 int n_nodes, int n_graphs, float *x_out, void *tape, size_t tape_bytes,
                                 void *workspace, size_t workspace_bytes, void *stream);
int cbgx_h2x_stack_backward(const float *packed, int num_layers, const void *tape, size_t tape_bytes, const float *h,
                            const uint8_t *lig_flag, const uint8_t *gen_flag, int n_nodes, const float *grad_x_out,
                            float *const *grads, int num_grads, float *grad_h, void *workspace,
                            size_t workspace_bytes, void *stream);

/* TargetDiff's training arithmetic around the denoiser call (targetdiff.py:82-124), one launch each instead of the ~260 small
 * launches the same formulas take as tensor operations (a training step is bound by its launch count as much as by its kernels).
 * Index tensors are int64 as PyTorch holds them (no conversion launches); gen / masks are bytes.  Preconditions the library cannot
 * check without reading device memory: 0 <= batch[a] < n_graphs (the per-graph sums are indexed by it), 0 <= t[g] < T, types < C.
 * cbgx_targetdiff_train_noise: q(x_t | x_0) on gen rows (CTNVPScheduler.forward_add_noise, diffusion_scheduler.py:117-134) and
 *   q(v_t | v_0) by Gumbel-argmax (TypeVPScheduler.forward_add_noise, :339-365).  t [B] per graph, batch [n_lig] graph of each
 *   ligand atom; eps [n_lig,3] ~ N(0,1), u [n_lig,C] ~ U(0,1) are inputs.  Outputs x_t [n_lig,3], c_t [n_lig,C] one-hot, v_t [n_lig].
 * cbgx_targetdiff_loss: position loss (type 'denoise': sum of squares against x0, :185-201) and atom-type loss (KL between
 *   q(v_{t-1} | v_t, v_0) and q(v_{t-1} | v_t, softmax(logits)); the decoder NLL for graphs at t = 0; :380-441), each the mean over
 *   the gen atoms of a graph, then over the graphs (targetdiff.py:103-121).  x_out [N,3] / logits [N,C] are the denoiser outputs,
 *   lig_rows [n_lig] the composed row of each ligand atom; tables = {log_alphas_v, log_one_minus_alphas_v, log_alphas_cumprod_v,
 *   log_one_minus_alphas_cumprod_v} [T].  losses [2] = {pos, atom}; x_pred [n_lig,3], c_pred [n_lig,C] = softmax(logits) (the
 *   `results` of the reference; may be NULL); grad_pos [n_lig,3], grad_logit [n_lig,C] receive d loss_pos / d x_out[row] and
 *   d loss_atom / d logits[row].  n_graphs <= 4096.
 * cbgx_targetdiff_loss_backward: grad_x_out [N,3] = *g_loss_pos * grad_pos and grad_logits [N,C] = *g_loss_atom * grad_logit on
 *   ligand rows, zero on protein rows; sort_idx [N] is the composition permutation (composed row r holds entry sort_idx[r] of
 *   cat(protein, ligand)); g_loss_* are DEVICE scalars (NULL = 0). */
int cbgx_targetdiff_train_noise(const float *x0, const int64_t *v0, const int64_t *t, const int64_t *batch,
                                const uint8_t *gen, int n_lig, int num_classes, const float *alphas_cumprod,
                                const float *log_alphas_cumprod, const float *log_one_minus_alphas_cumprod,
                                const float *eps, const float *u, float *x_t, float *c_t, int64_t *v_t, void *stream);
int cbgx_targetdiff_loss(const float *x_out, const float *logits, const int64_t *lig_rows, const float *x0,
                         const int64_t *v0, const int64_t *v_t, const int64_t *t, const int64_t *batch, const uint8_t *gen,
                         int n_lig, int n_graphs, int num_classes, const float *const *tables, float *losses, float *x_pred,
                         float *c_pred, float *grad_pos, float *grad_logit, void *stream);
int cbgx_targetdiff_loss_backward(const float *grad_pos, const float *grad_logit, const int64_t *sort_idx, int n_protein,
                                  int n_nodes, int num_classes, const float *g_loss_pos, const float *g_loss_atom,
                                  float *grad_x_out, float *grad_logits, void *stream);
/* cbgx_diffbp_loss (ABI 5): DiffBP's four training losses around its two network calls -- zero-COM noise prediction + centre-of-mass
 *   prediction (CoMPredictor.forward, diffbp.py:79-101), get_score_loss x 2 (diffusion_scheduler.py:203-218), MaskTypeSchedule.get_loss
 *   (:499-511: cross_entropy of the softmax OUTPUT, as the reference has it), xs_mean (:166-183) and interior_loss (diffbp.py:18-28)
 *   -- in two launches, with their gradients with respect to the network outputs.  All [N,.] arrays are in the COMPOSED row order
 *   (per graph: protein rows, then ligand rows; graph_ptr [B+1]; sort_idx [N] as above; lig_flag [N]); x_in = the composed input
 *   positions (protein positions and x_t), x_stack = the centre-of-mass head's output positions.  pos_noise / com_noise [n_lig,3],
 *   v0, type_flag (the mask of the type loss), gen [n_lig] in LIGAND order; t [B].  Ligands of at most 48 atoms (beyond that the
 *   reference restricts every protein atom to its 48 nearest ligand atoms: *bad is set to 1 and the host must take its tensor path).
 *   losses [4] = {pos, atom, com, inter}; scal [2] = {1 / D_gen, 1 / D_type}, the graph-count divisors of the masked means;
 *   gstats [8 n_graphs] scratch.  Gradient pieces, [N,3] / [N,C], zero on protein rows, to be combined by the caller with the
 *   upstream gradients g_* of the four losses:
 *     d L / d x_out   = g_pos scal[0] a_pos + g_inter a_int        d L / d x_stack = g_com scal[0] b_com + g_inter b_int
 *     d L / d logits  = g_atom scal[1] z_atom */
int cbgx_diffbp_loss(const float *x_out, const float *x_in, const float *x_stack, const float *logits, const int64_t *sort_idx,
                     const int32_t *graph_ptr, const uint8_t *lig_flag, const float *pos_noise, const float *com_noise,
                     const int64_t *v0, const uint8_t *type_flag, const uint8_t *gen, const int64_t *t, int n_protein, int n_lig,
                     int n_graphs, int num_classes, const float *alphas_cumprod, const float *betas, float rho, float gamma,
                     float *losses, float *scal, float *gstats, float *a_pos, float *a_int, float *b_com, float *b_int,
                     float *z_atom, int32_t *bad, void *stream);

/* cbgx_compose_plan (ABI 6): the index work of compose_context (repo/modules/common.py:189-214) -- sort_idx [N] = the STABLE argsort of
 *   cat(batch_protein, batch_ligand) (graph ids in [0, n_graphs), int64), and what its callers derive from it: batch_idx [N] the sorted ids,
 *   lig_flag [N] (1 = ligand atom), lig_rows [n_ligand] the composed row of every ligand atom, graph_ptr [n_graphs + 1] the CSR offsets --
 *   as a counting sort in three launches.  Exact for any order of the ids (arrays that are not non-decreasing, which no collated batch is,
 *   take a slow rank pass); ids outside [0, n_graphs) are dropped.  scratch: 4 n_graphs + 1 ints. */
int cbgx_compose_plan(const int64_t *batch_protein, const int64_t *batch_ligand, int n_protein, int n_ligand, int n_graphs,
                      int32_t *scratch, int64_t *sort_idx, int64_t *batch_idx, uint8_t *lig_flag, int64_t *lig_rows,
                      int32_t *graph_ptr, void *stream);

/* cbgx_diffsbdd_train_noise / cbgx_diffsbdd_loss (ABI 6): the tensor operations of DiffSBDD.get_loss (training mode) around its denoiser
 *   call, diffsbdd.py:91-195.  All graph-wise sums are taken per graph of the COMPOSED order (graph_ptr [B+1], sort_idx [N] as above), in a
 *   fixed order.  x0 [n_lig,3], x_protein [n_protein,3], v0 [n_lig] class indices (the one-hot / 4 features are formed inside), eps_x
 *   [n_lig,3] / eps_c [n_lig,C] the Gaussian draws, gen [n_lig], t [B] integer time steps in [0, T]; alpha_table / sigma_table [T+1] =
 *   sqrt(sigmoid(-gamma)), sqrt(sigmoid(gamma)) of the predefined schedule (schedule_utils.py:60-96).
 *   _train_noise: ligand centred on its mean, z_t for coordinates and types with the pocket re-centred on the noisy ligand
 *   (diffusion_scheduler.py:740-790) -> x_t, x_protein_t, c_t; gdata [4 B] = per graph {ligand atoms, KL prior of the coordinates, KL prior
 *   of the types (:846-868), -log p(c | z_0) [t == 0] (:930-945)}: every term of the loss that does not depend on the network.
 *   _loss: x_out [N,3] / logits [N,C] = the denoiser's outputs (composed order); losses [2] = {pos, atom} = mean over graphs of
 *   0.5 sum(err^2) [t != 0] / (n dim) + the t == 0 terms + KL (:886-900); glosses [2 B] scratch; x_pred / c_pred = the outputs on the
 *   ligand rows, in ligand order; grad_pos [n_lig,3] / grad_logit [n_lig,C] = d losses / d those rows -- scatter them with
 *   cbgx_targetdiff_loss_backward. */
int cbgx_diffsbdd_train_noise(const float *x0, const float *x_protein, const int64_t *v0, const float *eps_x, const float *eps_c,
                              const uint8_t *gen, const int64_t *t, const int64_t *sort_idx, const int32_t *graph_ptr, int n_protein,
                              int n_lig, int n_graphs, int num_classes, const float *alpha_table, const float *sigma_table,
                              int num_timesteps, float *x_t, float *x_protein_t, float *c_t, float *gdata, void *stream);
int cbgx_diffsbdd_loss(const float *x_out, const float *logits, const float *eps_x, const float *eps_c, const int64_t *t,
                       const int64_t *sort_idx, const int32_t *graph_ptr, int n_protein, int n_lig, int n_graphs, int num_classes,
                       const float *gdata, float *glosses, float *losses, float *x_pred, float *c_pred, float *grad_pos,
                       float *grad_logit, void *stream);

/* cbgx_embed_compose / cbgx_embed_compose_backward (ABI 6): the input side of a training step -- PLContextEmbedder
 *   (repo/modules/context_emb.py:137-230, the shipped configuration: Linear atom / residue / ligand-indicator embeddings, no time / vec)
 *   and compose_context (repo/modules/common.py:189-214) -- as one launch, and the embedder's weight gradients as two.
 *     protein row:  h = W_pa feat + b_pa + W_res onehot(aa) + b_res + b_ind           ligand row:  h = W_la c + b_la + W_ind + b_ind
 *     x = cat(x_protein, x_ligand)[sort_idx],  h likewise,  gen_flag = cat(gen_protein, gen_ligand)[sort_idx]
 *   protein_feat [n_protein, feat_dim] fp32, protein_aa [n_protein] class indices (an index outside [0, num_aa) embeds as no residue),
 *   ligand_feat [n_ligand, lig_dim] fp32 (one-hot or noised types); sort_idx [N] as above; gen_protein may be NULL (all 0), gen_flag may be
 *   NULL (not written).  params: the eight tensors of the embedder in nn.Linear layout -- protein_atom_emb.weight [128, feat_dim], .bias,
 *   residue_emb.weight [128, num_aa], .bias, ligand_atom_emb.weight [128, lig_dim], .bias, ligand_indicator.weight [128, 1], .bias.
 *   feat_dim + num_aa + lig_dim + 2 <= 120.  Outputs x [N,3], h [N,128] and ext [N,128], the rows
 *     ext = [ feat | onehot(aa) | 1 if protein | c | 1 if ligand | 0 ... ]
 *   that the backward multiplies with: h = ext . Wext for the stacked weights, so  dWext[c][j] = sum_r grad_h[r][c] ext[r][j]  holds
 *   every parameter's gradient.  cbgx_embed_compose_backward: partial = scratch of `groups` slabs of 128 x 128 floats (1 <= groups <= 256:
 *   the rows are split into that many partial sums, added in slab order -- reproducible); grad_out [128 (feat_dim + num_aa + lig_dim + 2)]
 *   = dW_pa [128, feat_dim] | dW_res [128, num_aa] | u [128] | dW_la [128, lig_dim] | v [128]  back to back, with
 *   db_pa = db_res = u,  db_la = dW_ind[:, 0] = v,  db_ind = u + v.  Coordinates and features get no gradient (they are data). */
int cbgx_embed_compose(const float *x_protein, const float *x_ligand, const float *protein_feat, const int64_t *protein_aa,
                       const float *ligand_feat, const int64_t *sort_idx, const uint8_t *gen_protein, const uint8_t *gen_ligand,
                       int n_protein, int n_ligand, int feat_dim, int num_aa, int lig_dim, const float *const *params, float *x,
                       float *h, float *ext, uint8_t *gen_flag, void *stream);
int cbgx_embed_compose_backward(const float *grad_h, const float *ext, int n_nodes, int feat_dim, int num_aa, int lig_dim,
                                float *partial, int groups, float *grad_out, void *stream);

/* ---- measurement hook (bench.py) ----------------------------------------------------------------
 * Between cbgx_profile_begin() and cbgx_profile_end() every kernel launch is bracketed by HIP events on
 * its own stream.  cbgx_profile_end() synchronises them and returns, per kernel class, the summed
 * device time in ms and the launch count.  Classes: 0 knn, 1 gate, 2 node GEMM, 3 node query fold,
 * 4 x2h edge kernel over all nodes, 5 h2x edge kernel (node list), 6 x2h edge kernel over a node list (pruned last
 * layers), 7 x2h edge backward over all nodes, 8 h2x edge backward, 9 training GEMMs, 10 x2h edge backward over a node list
 * 11 the ordered gather of the x2h backward's edge-row mode (CBGX_PROFILE_CLASSES = 12).  Not thread-safe with concurrent launches
 * from other threads; process-wide. */
#define CBGX_PROFILE_CLASSES 12
int cbgx_profile_begin(int max_launches);
int cbgx_profile_end(double *ms_by_class, int *launches_by_class, int num_classes);

#ifdef __cplusplus
}
#endif
#endif /* CBGX_H */
