/*
 * dpf_hip.h -- C ABI of libdpf_hip.so: the MI355X (gfx950) implementation of
 * dpf-nets' per-point flow decoder and structural losses.
 *
 * Drop-in boundary.  Every entry point takes raw DEVICE pointers, plain ints
 * and a HIP stream (hipStream_t passed as void*); no torch / ATen types.  The
 * caller owns every buffer (inputs, outputs and scratch); nothing here
 * allocates, frees or synchronises.  All launches are asynchronous on
 * `stream`.  Return value: 0 on success, otherwise the hipError_t of the failed
 * launch, or a negative DPF_E* code for an argument the kernels do not support
 * (the reference returned void and, for Chamfer, checked nothing:
 * nndistance.cu:125-128).
 *
 * File:line citations are into the reference tree (Regenerator/dpf-nets).
 */
#ifndef DPF_HIP_H
#define DPF_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *dpf_stream_t; /* hipStream_t */

#define DPF_EINVAL (-1)  /* bad size / null pointer */
#define DPF_ENOSUP (-2)  /* shape the kernels are not built for (e.g. F != 64) */

/* ------------------------------------------------------------------------ *
 * Structural losses.  Same argument order and meaning as the reference's C
 * launchers, which lib/metrics/pytorch_structural_losses/src/structural_loss.cpp
 * binds (:35, :50, :67, :97, :118).  Point clouds are point-major fp32
 * (B, n, 3) contiguous; indices int32.
 * ------------------------------------------------------------------------ */

/* replaces nndistance(...)      src/nndistance.cuh:1, nndistance.cu:125-128.
 * result[b,j]  = min_k |xyz[b,j]-xyz2[b,k]|^2, result_i = its FIRST argmin;
 * result2/_i the same with the clouds swapped.  Bit-exact contract:
 * d = (dx*dx + dy*dy) + dz*dz, dx = xyz2 - xyz, no FMA contraction.
 * Non-finite input (NaN / Inf coordinates, or coordinates whose squared
 * distances overflow): the result of the reference's own loop,
 * nndistance.cu:5-122 -- candidates in batches of 512, the first of a batch
 * taken unconditionally, strict '<' inside a batch (:26, 116), strict '>'
 * across batches (:120) -- so a NaN distance at candidate 0 gives (NaN, 0), at
 * candidate 512k it makes batch k lose, elsewhere it is skipped.  Every
 * implementation below (scan, matrix-core filter, strided, _cd, pairwise)
 * detects such input and runs that loop for the affected queries
 * (csrc/nn_refscan.h); NaN payloads are not specified. */
int dpf_nndistance(int b, int n, const float *xyz, int m, const float *xyz2,
                   float *result, int *result_i, float *result2, int *result2_i,
                   dpf_stream_t stream);

/* Which kernel serves dpf_nndistance(_strided) for rank-sized batches of 1024..8192-point clouds (< 512 waves of the scan):
 * the LDS-staged one-query-per-lane kernel (csrc/chamfer.hip nn_small_kernel, r04) or the scalar-load scan.  Same bits.
 * mode: -1 = by size (default; env DPF_NN_SMALL), 0 = never, 1 = whenever the clouds fit.  Returns the previous mode. */
int dpf_nn_small_mode(int mode);

/* dpf_nndistance with explicit strides (in floats) between consecutive clouds of
 * each set; stride 0 broadcasts one cloud over the batch -- the "expand +
 * contiguous" copy of pairwise_CD (lib/networks/utils.py:104-107) disappears. */
int dpf_nndistance_strided(int b, int n, const float *xyz, long xyz_stride, int m,
                           const float *xyz2, long xyz2_stride, float *result, int *result_i,
                           float *result2, int *result2_i, dpf_stream_t stream);

/* Same results as dpf_nndistance, bit for bit, matrix-core filtered: one bf16
 * MFMA per 32x32 pairs bounds every distance to within 2^-14*R2, and only the
 * candidates that can still be the fp32-exact minimiser (or tie with it) are
 * evaluated with the exact formula.  The fragments are built inside the kernel:
 * `workspace` is unused (kept for the ABI; _workspace_bytes returns 0).  Both
 * clouds under 32 points -> the brute-force kernel. */
size_t dpf_nndistance_mfma_workspace_bytes(int b, int n, int m);
int dpf_nndistance_mfma(int b, int n, const float *xyz, int m, const float *xyz2,
                        float *result, int *result_i, float *result2, int *result2_i,
                        void *workspace, size_t workspace_bytes, dpf_stream_t stream);

/* dpf_nndistance's contract and bits; the matrix-core filtered kernel where it is the faster
 * one (>= 1e8 pair evaluations and >= 64 workgroups of 256 queries), the VALU scan otherwise.
 * This is what the Python mirror calls by default. */
int dpf_nndistance_auto(int b, int n, const float *xyz, int m, const float *xyz2,
                        float *result, int *result_i, float *result2, int *result2_i,
                        dpf_stream_t stream);
int dpf_nndistance_strided_auto(int b, int n, const float *xyz, long xyz_stride, int m,
                                const float *xyz2, long xyz2_stride, float *result,
                                int *result_i, float *result2, int *result2_i,
                                dpf_stream_t stream);   /* dpf_nndistance_strided, same choice */

/* replaces nndistancegrad(...)  src/nndistance.cuh:2, nndistance.cu:149-154.
 * grad_xyz1 / grad_xyz2 are fully overwritten (the zero-fill happens on
 * `stream`, not on the null stream as at nndistance.cu:150-151). */
int dpf_nndistancegrad(int b, int n, const float *xyz1, int m, const float *xyz2,
                       const float *grad_dist1, const int *idx1,
                       const float *grad_dist2, const int *idx2,
                       float *grad_xyz1, float *grad_xyz2, dpf_stream_t stream);

/* replaces approxmatch(...)     src/approxmatch.cuh:6, approxmatch.cu:299-307.
 * match: (b, m, n) fp32, fully overwritten.  temp: (b, 2*(n+m)) fp32 scratch
 * (remainL | remainR | ratioL | ratioR per cloud, approxmatch.cu:4). */
int dpf_approxmatch(int b, int n, int m, const float *xyz1, const float *xyz2,
                    float *match, float *temp, dpf_stream_t stream);

/* dpf_approxmatch with `match` written ONCE: the nine levels' ratio vectors and the passes' operand records are kept in
 * `workspace` and the matching is materialised by a final pass (4*n*m instead of 68*n*m bytes of HBM traffic per
 * cloud).  dpf_approxmatch_workspace_bytes covers both kernel families below: ~ 36*(n+m) + 16*(n+2m) + 48*m + n/2 bytes
 * per cloud for the packed-VALU one (bit-identical to dpf_approxmatch) + ~ 64*n + 350*m for the matrix-core one.
 * NULL / short workspace -> the read-modify-write path. */
size_t dpf_approxmatch_workspace_bytes(int b, int n, int m);
/* r05 / r06: with a workspace the 27 level passes run on the matrix cores (expanded-form squared distance, two chained MFMAs
 * per 32 x 32 pairs whose large terms are exact integer arithmetic on a per-call grid; csrc/emd.hip) when every point of
 * both clouds, centred on cloud 1's centroid c, has log2(e) |x - c|^2 <= 16 and is finite -- decided per call on the device;
 * otherwise, and after dpf_emd_set_matrix_path(0), the packed-VALU kernels run, whose results are bit-identical to
 * dpf_approxmatch.  The matrix-core results are within the tolerance contract (cost 1e-4; the exp2 arguments within 2e-5 of
 * float64 at the steepest level on unit-size clouds, measured by dpf_debug_emd_exponents), not bit-identical.  DETERMINISM of
 * the matrix-core kernels is a property of the compiled code, not of the source: two builds of r05 returned run-to-run
 * differing bits (csrc/emd.hip, opaque_zero).  r06 found why: the compiler's vectoriser had written a packed fp32 instruction form
 * (low half reading the high word of a VGPR pair) that gfx950 executes wrongly in lanes 48-63 while another wave of the SIMD issues
 * MFMAs (tools/ubench/pk_vs_mfma_forms.hip).  The Makefile refuses to link ANY object that holds that form, or packed
 * fp32 beside MFMAs, or an emd.o whose MFMAs do not have the properties of the build that repeats (tools/mfma_overlap_check.py);
 * tests/test_gpu_emd.py holds the repeat tests, tests/test_gpu_interference.py runs the kernels beside an MFMA-issuing kernel of
 * another stream.  Returns the previous setting. */
int dpf_emd_set_matrix_path(int on);
int dpf_approxmatch_ws(int b, int n, int m, const float *xyz1, const float *xyz2,
                       float *match, float *temp, void *workspace, size_t workspace_bytes,
                       dpf_stream_t stream);

/* dpf_approxmatch_ws followed by dpf_matchcost in one call (what MatchCostFunction.forward does,
 * match_cost.py:20-22): the pass that materialises `match` also accumulates
 * cost[b] = sum match * |xyz1 - xyz2| (the distance is at hand), saving matchcost's pass over
 * the (b, m, n) matching.  Fixed-order partial sums (deterministic); agrees with dpf_matchcost
 * to fp32 summation order.  The workspace (dpf_approxmatch_workspace_bytes) is required. */
int dpf_approxmatch_cost_ws(int b, int n, int m, const float *xyz1, const float *xyz2,
                            float *match, float *temp, float *cost, void *workspace,
                            size_t workspace_bytes, dpf_stream_t stream);

/* replaces matchcost(...)       src/approxmatch.cuh:7, approxmatch.cu:309-316.
 * out: (b,) = sum_{l,k} match[b,l,k] * |xyz1[b,k]-xyz2[b,l]|_2. */
int dpf_matchcost(int b, int n, int m, const float *xyz1, const float *xyz2,
                  const float *match, float *out, dpf_stream_t stream);

/* replaces matchcostgrad(...)   src/approxmatch.cuh:8, approxmatch.cu:318-326. */
int dpf_matchcostgrad(int b, int n, int m, const float *xyz1, const float *xyz2,
                      const float *match, float *grad1, float *grad2,
                      dpf_stream_t stream);

/* The same two gradients with `match` read ONCE instead of twice (one workgroup per
 * 256 columns walks all rows; per-row lane sums by a halving butterfly; partials in
 * `workspace`, added in a fixed order).  Deterministic; agrees with
 * dpf_matchcostgrad to fp32 summation order.  NULL / short workspace -> that. */
size_t dpf_matchcostgrad_workspace_bytes(int b, int n, int m);
int dpf_matchcostgrad_ws(int b, int n, int m, const float *xyz1, const float *xyz2,
                         const float *match, float *grad1, float *grad2,
                         void *workspace, size_t workspace_bytes, dpf_stream_t stream);

/* Chamfer caller reductions, lib/networks/evaluating.py:112:
 * cd[b] = mean_j dist1[b,j] + mean_k dist2[b,k]  (the reference then takes
 * .mean() over the batch).  Deterministic: one workgroup per cloud, fixed tree. */
int dpf_chamfer_reduce(int b, int n, int m, const float *dist1, const float *dist2,
                       float *cd, dpf_stream_t stream);

/* f_score's reductions, lib/networks/utils.py:38-42, from the two distance rows of one nn_distance call:
 * out[b] = 2 P R / (P + R + 1e-7), P = 100 * mean(dist2[b] < threshold), R = 100 * mean(dist1[b] < threshold).
 * One launch instead of ~10 elementwise / reduction launches; counts are integers (exact). */
int dpf_fscore_reduce(int b, int n, int m, const float *dist1, const float *dist2, float threshold,
                      float *out, dpf_stream_t stream);

/* nn_distance and its callers' reduction in one call (lib/networks/evaluating.py:110-113): the four outputs of
 * dpf_nndistance (same bits) plus cd[b] = mean(result[b]) + mean(result2[b]).  With the matrix-core kernel the
 * workgroups publish fixed-order sums of their distances and the LAST workgroup of every cloud (a ticket per cloud,
 * device-coherent stores / loads on both sides) adds them in tile order: no second launch.  workspace:
 * dpf_nndistance_cd_workspace_bytes bytes, caller-owned; its first b words are the tickets -- they must be zero on
 * entry and are zero again on exit: pass tickets_are_zero = 1 for a workspace that was cleared once and is kept,
 * 0 to have this call clear them.  Small problems: dpf_nndistance_auto + dpf_chamfer_reduce. */
size_t dpf_nndistance_cd_workspace_bytes(int b, int n, int m);
int dpf_nndistance_cd(int b, int n, const float *xyz, int m, const float *xyz2,
                      float *result, int *result_i, float *result2, int *result2_i, float *cd,
                      void *workspace, size_t workspace_bytes, int tickets_are_zero, dpf_stream_t stream);

/* pairwise_CD, lib/networks/utils.py:90-117 (called three times per generative evaluation, evaluating.py:245-247):
 * cds[i, j] = mean(dist1) + mean(dist2) of nn_distance(clouds1[i], clouds2[j]) for ALL pairs in ONE launch (the last
 * workgroup of every pair adds the workgroups' fixed-order partial sums; + a memset of the pairs' tickets).  clouds1 (n1, n, 3), clouds2 (n2, m, 3), cds (n1, n2),
 * all point-major fp32; the reference's per-i `expand + contiguous + nn_distance` loop (utils.py:104-107) and its
 * (N2, n) intermediates disappear.  Distances are those of dpf_nndistance (bit-exact minima); the means are summed
 * per 512-query workgroup, then in ascending tile order.  n, m >= 32; n1 <= 32767, n2 <= 65535.
 * A row-sharded evaluation passes its own rows of clouds1 and gathers the (rows, n2) blocks. */
size_t dpf_pairwise_cd_workspace_bytes(int n1, int n2, int n, int m);
int dpf_pairwise_cd(int n1, int n2, int n, int m, const float *clouds1, const float *clouds2, float *cds,
                    void *workspace, size_t workspace_bytes, dpf_stream_t stream);

/* ------------------------------------------------------------------------ *
 * Per-point conditional affine-coupling flow (eval-mode BatchNorm), i.e.
 * LocalCondRNVPDecoder.forward (lib/networks/decoders.py:54-72) over
 * CondRealNVPFlow3D.forward (lib/networks/flows.py:95-117), F = 64.
 *
 * Weights arrive as one "canonical" fp32 block per coupling layer, layers in
 * DIRECT order (decoders.py:58-64).  dpf_flow_canon_floats(G) floats per layer:
 *
 *   for br in (logvar, mu):                               offsets in floats
 *     W0   [64][2]   sd0.weight, columns = the (up to) two keep channels,
 *                    zero column if the layer keeps one            flows.py:26/61
 *     BN0  gamma[64] beta[64] running_mean[64] running_var[64]     flows.py:27/62
 *     W1   [64][64]  sd1.weight                                    flows.py:29/64
 *     BN1  running_mean[64] running_var[64]  (affine=False)        flows.py:30/65
 *     W2   [2][64]   sd2.weight, rows = the (up to) two warp channels,
 *                    zero row if the layer warps one               flows.py:49/84
 *     b2   [4]       sd2.bias in [0..1], [2..3] unused
 *     for s in (w, b):                                             flows.py:33-45/68-80
 *       Wf0 [64][G]  film_{s}0.weight
 *       BNf gamma[64] beta[64] running_mean[64] running_var[64]
 *       Wf1 [64][64] film_{s}1.weight
 *       bf1 [64]     film_{s}1.bias
 *
 * meta: int32 [n_layers][4] = {keep_a, keep_b, warp_a, warp_b}, -1 = absent
 * (keep_inds / warp_inds of flows.py:17-23), device memory.
 * ------------------------------------------------------------------------ */

#define DPF_FLOW_F 64

/* precision of the 64x64 conditioner contraction on the matrix cores */
#define DPF_PREC_BF16   1  /* one bf16 MFMA product                      ~3e-3 rel */
#define DPF_PREC_BF16X3 2  /* hi/lo split, 3 bf16 MFMA products          ~1e-5 rel */
#define DPF_PREC_BF16X6 3  /* hi/mid/lo split, 6 bf16 MFMA products      fp32-class */
#define DPF_PREC_F16X3  4  /* hi/lo split in fp16 (11 + 11 bits), 3 MFMA products ~1e-6 rel; the activations'
                            * split costs 5 instead of 8 VALU per pair (v_cvt_pkrtz_f16_f32 + v_fma_mixlo/hi_f16).
                            * Hidden activations and W1 must stay below 65504 in magnitude (fp16 range). */

#define DPF_MODE_DIRECT  0 /* p_out = sqrt(eps+exp(logvar))*p + mu     flows.py:113 */
#define DPF_MODE_INVERSE 1 /* p_out = (p-mu)/sqrt(eps+exp(logvar))     flows.py:115 */

size_t dpf_flow_canon_floats(int G);
size_t dpf_flow_packed_bytes(int n_layers, int G, int precision);
size_t dpf_flow_film_floats(int n_layers, int B);

/* canonical fp32 weights -> `packed`: MFMA-fragment-ordered bf16 parts with
 * BatchNorm folded (dpf_flow_forward), followed by the transposed fp32
 * conditioner weights (dpf_flow_film).  Run once per weight version (not per
 * batch).  The film weights of the n_layers-layer prefix of a longer packed
 * buffer are NOT at the prefix's offset: pack a stack with the n_layers it
 * will be run with. */
int dpf_flow_pack(int n_layers, int G, int precision, const float *canon,
                  const int *meta, void *packed, dpf_stream_t stream);

/* per-cloud FiLM conditioner (flows.py:33-45,68-80,100-101,105-106) for every
 * layer at once: g (B, G) -> film (dpf_flow_film_floats floats). */
int dpf_flow_film(int n_layers, int B, int G, int precision, const void *packed,
                  const float *g, float *film, float flow_eps, dpf_stream_t stream);

/* The fused L-layer stack.  p_in / p_out / sum_logvar: (B, 3, N) fp32
 * channel-major as the networks use (flows.py:95).  Optional per-layer lists
 * ps / mus / logvars: (n_layers, B, 3, N) fp32 in DIRECT layer order for both
 * modes (decoders.py:61-70), or NULL to skip materialising them.
 * sum_logvar = sum over layers of logvar (what sum(logvars) gives
 * losses.py:13); may be NULL.  p_out_pointmajor: optional (B, N, 3) copy of
 * p_out in the layout the structural losses take (the
 * `.transpose(1, 2).contiguous()` of lib/networks/evaluating.py:110), or NULL. */
int dpf_flow_forward(int n_layers, int B, int N, int mode, int precision,
                     const void *packed, const int *meta, const float *film, const float *p_in,
                     float *p_out, float *p_out_pointmajor, float *sum_logvar,
                     float *ps, float *mus, float *logvars,
                     float flow_eps, dpf_stream_t stream);
/* dpf_flow_forward in direct mode with `reparameterize` (lib/networks/models.py:76-79, called at :212 / :118) fused into
 * the prologue: `noise` (B,3,N) is what torch.randn_like drew (the generator stream stays the caller's), the stack starts
 * from z = noise * exp(0.5 * lv0) + mu0 with mu0 / lv0 read through (batch, channel, point) element strides (the models'
 * stride-0 expansions, models.py:153-158 / 203-209); z_out (B,3,N, may be NULL) receives z = p_prior_samples[0]. */
int dpf_flow_forward_base(int n_layers, int B, int N, int precision, const void *packed, const int *meta,
                          const float *film, const float *noise, const float *mu0, long mu_sb, long mu_sc, long mu_sn,
                          const float *lv0, long lv_sb, long lv_sc, long lv_sn, float *z_out, float *p_out,
                          float *p_out_pointmajor, float *sum_logvar, float *ps, float *mus, float *logvars,
                          float flow_eps, dpf_stream_t stream);

/* Tiling of dpf_flow_forward / _base (results agree to rounding; both are held to the same goldens):
 * a wave owns 32 points (v_mfma_f32_32x32x16, csrc/flow.hip) or, for small per-GPU batches at f16x3 -- at most one tile per
 * SIMD of the chip, B * ceil(N / 16) <= 1024, e.g. the 4 clouds a rank of an 8-GPU job holds of BASELINE's 32 -- 16 points
 * (v_mfma_f32_16x16x32, csrc/flow16.hip).  mode: -1 = by size (default; env DPF_FLOW_TILE16), 0 = never, 1 = whenever
 * the precision allows.  Returns the previous mode.  Process-wide; meant for tests and measurements. */
int dpf_flow_set_tile16(int mode);
long dpf_flow_tile16_launches(void);   /* calls of dpf_flow_forward / _base served by the 16-point kernel so far */

/* ---------------------------------------------------------------------------
 * Training mode (model.train()) of one coupling layer: batch-statistics
 * BatchNorm in both conditioner SharedDot stacks (flows.py:27,30,62,65), and the
 * backward pass autograd derives from CondRealNVPFlow3D.forward (flows.py:95-117;
 * lib/networks/training.py:55 calls loss.backward()).  precision: DPF_PREC_BF16X3 or
 * DPF_PREC_BF16X6 (of the forward contraction and its recomputation; see
 * csrc/flow_train.hip).
 *
 * tcanon, per layer dpf_flow_train_canon_floats() floats, per branch
 * (logvar, mu) 4484 floats:
 *     W0 [64][nk] @0 (sd0.weight as stored, nk = 1 or 2 keep channels, zeros up to 128)
 *     gamma0[64] @128   beta0[64] @192   W1[64][64] @256
 *     W2 [nw][64] @4352 (zeros up to 128)                b2[nw] @4480 (zeros up to 4)
 * -- every parameter tensor appears flattened exactly as PyTorch stores it, so the
 * block is one concatenation and its gradient splits into per-parameter slices
 * (BN1 is affine=False and the running statistics are not inputs in training mode).
 * dcanon has the same layout.
 *
 * fm / dfm: the FiLM vectors (cw, cb) of flows.py:100-101 / 105-106 that the
 * per-cloud conditioner nets produced for this batch, [branch][w|b][B][64];
 * those nets (B x 64 tensors) stay with the caller.
 *
 * film_l: dpf_flow_train_film_floats(B) floats; its first B*512 floats are the
 * block dpf_flow_forward(n_layers = 1, precision, packed = packed_l, film = film_l)
 * takes.
 * stats_l: dpf_flow_train_stats_floats() floats, [branch][6][64] =
 * mean0, rstd0, mean1, rstd1, unbiased batch var0, unbiased batch var1 (what
 * BatchNorm1d feeds its running statistics with), then 8 floats of input moments.
 * ------------------------------------------------------------------------ */
size_t dpf_flow_train_canon_floats(void);
size_t dpf_flow_train_packed_bytes(int n_layers, int precision);
size_t dpf_flow_train_stats_floats(void);
size_t dpf_flow_train_film_floats(int B);
size_t dpf_flow_train_workspace_bytes(int B, int N);

/* W1 / W1^T MFMA fragments of every layer (once per optimizer step) */
int dpf_flow_train_pack(int n_layers, int precision, const float *tcanon, void *packed,
                        dpf_stream_t stream);

/* Training-mode forward of an n_layers stack (layers in DIRECT order in every array; the mode
 * decides the order they run in).  meta_host / meta_dev: the int32 [n_layers][4] keep/warp table
 * in host and in device memory.  tcanon (L, canon_floats), packed (from dpf_flow_train_pack; the
 * input-layer fragments are completed here), fm (L,[branch][w|b][B][64]).  Outputs: ps / mus /
 * logvars (L,B,3,N) in DIRECT order, stats (L, stats_floats), film (L, film_floats(B)) -- stats,
 * film and packed are what the backward call needs back. */
int dpf_flow_train_forward(int n_layers, int B, int N, int mode, int precision,
                           const int *meta_host, const int *meta_dev, const float *tcanon,
                           void *packed, const float *fm, const float *p_in, float *ps, float *mus,
                           float *logvars, float *stats, float *film, float flow_eps,
                           void *workspace, dpf_stream_t stream);

/* Backward of the stack.  ps / mus / logvars: the forward call's three (L,B,3,N) outputs, unmodified (the
 * per-point mu and logvar of a layer are read back rather than recomputed).  g_ps / g_mus / g_lvs: (L,B,3,N)
 * gradients w.r.t. the three output lists (g_mus, g_lvs may be NULL = zero).  Overwrites dp_in (B,3,N),
 * dcanon (L, canon_floats) and dfm (L,[branch][w|b][B][64]); dp_tmp: (B,3,N) scratch. */
int dpf_flow_train_backward(int n_layers, int B, int N, int mode, int precision,
                            const int *meta_host, const float *tcanon, const void *packed,
                            const float *film, const float *stats, const float *p_in,
                            const float *ps, const float *mus, const float *logvars,
                            const float *g_ps, const float *g_mus,
                            const float *g_lvs, float *dp_in, float *dp_tmp, float *dcanon,
                            float *dfm, float flow_eps, void *workspace, dpf_stream_t stream);

/* The same with ONE (B,3,N) gradient pointer per layer and output list (host arrays of n_layers
 * device pointers): autograd produces a gradient per output tensor, and training.py's loss only
 * touches ps[0], mus[0] and the logvars, so nothing is stacked into (L,B,3,N) blocks and unused
 * outputs cost no memory traffic.  Any table, and any entry, may be NULL (zero gradient). */
int dpf_flow_train_backward_lists(int n_layers, int B, int N, int mode, int precision,
                                  const int *meta_host, const float *tcanon, const void *packed,
                                  const float *film, const float *stats, const float *p_in,
                                  const float *ps, const float *mus, const float *logvars,
                                  const float *const *g_ps,
                                  const float *const *g_mus, const float *const *g_lvs,
                                  float *dp_in, float *dp_tmp, float *dcanon, float *dfm,
                                  float flow_eps, void *workspace, dpf_stream_t stream);

/* library identification: returns e.g. "dpf_hip gfx950 r1" */
const char *dpf_version(void);

/* ---- PointNet cloud encoder + max over the points (eval-mode BatchNorm) ---------------------
 * replaces PointNetCloudEncoder.forward (lib/networks/encoders.py:27-28) for the architecture
 * every config uses, SharedDot(no bias).BatchNorm1d.ReLU x 4 over 3 -> 64 -> 128 -> 256 -> 512
 * (encoders.py:15-25, configs pc_enc_*), together with the torch.max(features, dim=2)[0] the
 * models apply to it (lib/networks/models.py:85,131,175).
 *
 * canon: dpf_encoder_canon_floats() = 176 064 fp32, per layer l = 0..3
 *   W[cout][cin] (SharedDot.weight[0]) | bn.weight | bn.bias | bn.running_mean | bn.running_var.
 * dpf_encoder_pack folds BatchNorm and lays the weights out as bf16 MFMA fragments (once per weight
 * version and precision) into `packed` (dpf_encoder_packed_bytes(precision)).
 * dpf_encoder_forward: x (B,3,N) channel-major -> gmax (B,512) fully overwritten; feat, when not
 * NULL, also receives the per-point features (B,512,N) (what the module's forward returns). */
size_t dpf_encoder_canon_floats(void);
size_t dpf_encoder_packed_bytes(int precision);
int dpf_encoder_pack(int precision, const float *canon, void *packed, dpf_stream_t stream);
int dpf_encoder_forward(int B, int N, int precision, const void *packed, const float *x,
                        float *gmax, float *feat, dpf_stream_t stream);

/* ---- PointNet cloud encoder, TRAINING mode (batch-statistics BatchNorm) + max over the points, and backward ----
 * replaces, under model.train(), PointNetCloudEncoder.forward (lib/networks/encoders.py:27-28) followed by
 * torch.max(features, dim=2)[0] (lib/networks/models.py:131), and the gradients autograd derives for the
 * encoder's parameters (lib/networks/training.py:55).  precision: DPF_PREC_BF16X6 or DPF_PREC_BF16X3 for the
 * forward contractions (they decide the ReLU masks and the argmax); gradient contractions are bf16x3; fp32
 * accumulation; every reduction in a fixed order (deterministic).
 *
 * canon: the block of dpf_encoder_forward (running_mean / running_var slots are not read).
 * ws: dpf_encoder_train_workspace_bytes(B, N) bytes; the forward pass leaves the pre-BatchNorm activations,
 *   the batch statistics and the argmax there, the backward pass of the SAME step reads them.
 * forward: x (B,3,N) -> pooled (B,512).  batch_stats (optional, 2 * 960 floats): per layer mean[C] | biased
 *   var[C], layers in order.  running (optional): HOST array of 8 DEVICE pointers running_mean_0,
 *   running_var_0, ... updated in place as torch.nn.BatchNorm1d does with `momentum` (unbiased variance).
 *   B * N must be >= 2.
 * backward: g_pooled (B,512) = d loss / d pooled, pooled = the forward's output -> dcanon (canon layout: dW,
 *   d gamma, d beta per layer; the running-statistics slots are left untouched); dx (optional, (B,3,N)) = d loss / d x. */
size_t dpf_encoder_train_workspace_bytes(int B, int N);
int dpf_encoder_train_forward(int B, int N, int precision, const float *canon, const float *x, void *ws, float *pooled,
                              float *batch_stats, float *const *running, float momentum, dpf_stream_t stream);
int dpf_encoder_train_backward(int B, int N, const float *canon, const float *x, void *ws, const float *pooled,
                               const float *g_pooled, float *dcanon, float *dx, dpf_stream_t stream);

/* The training-mode entry points (dpf_flow_train_forward / _backward[_lists], dpf_encoder_train_forward / _backward) issue
 * hundreds of small dependent launches per call; a call whose scalars AND pointers have been seen twice is recorded once
 * (stream capture) and from then on replayed as one hipGraph -- a training loop presents the same addresses every step.
 * A replay needs the stored key bytes to match, not only their hash.  Only kernel nodes are recorded (no memset nodes).
 * DPF_TRAIN_GRAPH=0 in the environment switches that off; dpf_train_graph_set_enabled(on) does so at run time and returns
 * the previous state.  dpf_train_graph_replays(): calls served by a replay so far.  dpf_train_graph_stats(out[5]):
 * {replays, eager calls, recordings, evictions, keys that could not be captured} -- evictions growing while replays stand
 * still means the loop presents more than 8 live pointer sets per entry point, or its allocator does not settle. */
/* Test hook of the matrix-core Chamfer filter (csrc/chamfer_mfma.hip): the surrogate s(q, c) = |c - mu|^2 - 2 (q - mu).(c - mu)
 * of every pair of ONE pair of clouds ((nq, 3) queries, (nc, 3) candidates), formed exactly as the filter forms it; s is
 * (nq, nc) floats, out4 = {mu_x, mu_y, mu_z, R2}.  The filter's exactness rests on |s - exact| <= 2^-14 R2. */
int dpf_debug_nn_surrogate(int nq, const float *q, int nc, const float *c, float *s, float *out4, dpf_stream_t stream);
/* Test hook of the matrix-core approx-EMD passes (csrc/emd.hip): out (m, n) = the exp2 arguments -4^j log2(e) |xyz2_l - xyz1_k|^2
 * of ONE cloud pair at annealing level j (7 .. -1), formed exactly as the passes form them (same records, level vectors and
 * MFMAs); meta8 (8 device floats or NULL) = {centroid xyz, out-of-range flag (then `out` is untouched), 2^g, T, 0, 0}.
 * Workspace as for dpf_approxmatch_ws(1, n, m). */
int dpf_debug_emd_exponents(int n, int m, const float *xyz1, const float *xyz2, int level_j, float *out, float *meta8,
                            void *workspace, size_t workspace_bytes, dpf_stream_t stream);
long dpf_train_graph_replays(void);
/* Workgroups of the training backward pass that gave up waiting for role workgroups of their own launch (pass 1: the column sums
 * of the layer above; pass 2, small batches: the per-cloud totals and BatchNorm-backward means of pass 1) and did the sums
 * themselves -- process-wide.  0 as long as the role workgroups, placed first in the grid, are dispatched first. */
long dpf_train_colsum_fallbacks(void);
void dpf_train_graph_stats(long *out);
int dpf_train_graph_set_enabled(int on);
/* Diagnostics: per-kernel time of the training stack's launches.  (1, NULL, NULL) starts collecting -- every launch of the
 * per-layer kernels is bracketed by HIP events on its stream; graph recording / replay is off while it collects --,
 * (0, us[8], calls[8]) synchronises the device, returns the summed time in microseconds and the number of launches per kernel
 * id and stops.  ids: 0 tstats_x, 1 tstats_h1, 2 tfold, 3 flow_kernel (L = 1), 4 tbwd1, 5 tbwd2, 6 tcolsum, 7 tbwd3f. */
int dpf_train_kernel_times(int enable, double *us_out, long *calls_out);

/* ---- fused AMSGrad-Adam step over one flat fp32 buffer -----------------------------------------
 * replaces the per-parameter update of the reference's optimizer, lib/networks/optimizers.py:52-74 (state['step'] and the
 * bias corrections :48, :63-64 stay with the caller): exp_avg / exp_avg_sq / max_exp_avg_sq / p updated in place in ONE pass,
 * with the roundings of the reference's op sequence as PyTorch-ROCm executes it (bit-identical, csrc/adam.hip).
 * max_exp_avg_sq = NULL: plain Adam (:59).  The hyper-parameters are the Python doubles of the parameter group;
 * bias_correction1 = 1 - beta1^step, bias_correction2 = sqrt(1 - beta2^step) (both non-zero).  Buffers 16-byte aligned. */
int dpf_adam_step(size_t n, float *p, const float *g, float *exp_avg, float *exp_avg_sq, float *max_exp_avg_sq, double lr,
                  double beta1, double beta2, double eps, double weight_decay, double bias_correction1, double bias_correction2,
                  dpf_stream_t stream);

/* ---- latent prior flow: GlobalRNVPDecoder on (B, G) codes, eval-mode BatchNorm ---------------
 * replaces GlobalRNVPDecoder.forward (lib/networks/decoders.py:21-38): n_steps = 2 * n_flows
 * RealNVPFlow steps (lib/networks/flows.py:198-213) in ONE launch, both modes.
 *
 * codes (HOST array, n_steps ints): which coordinates a step warps -- RealNVPFlowCouple's
 * patterns (flows.py:224-233): 0 even, 1 odd, 2 first half, 3 second half; G must be even.
 * canon: per step dpf_gprior_canon_floats(G, n_features) fp32, for T in (mu, logvar), K = G/2:
 *   mlp0.weight [n_features][K] | bn.weight | bn.bias | bn.running_mean | bn.running_var |
 *   mlp1.weight [K][n_features] | mlp1.bias [K]                      (flows.py:176-196).
 * dpf_gprior_pack folds BatchNorm and transposes the weights (once per weight version).
 * dpf_gprior_forward: g (B,G) -> gs, mus, lvs (n_steps,B,G) in DIRECT order whatever the mode
 * (mu / logvar zero on a step's kept coordinates, as the reference's lists), sum_lv (B,G) the sum of
 * the logvars over the steps, g_out (B,G) the end of the chain (gs[n_steps-1] direct, gs[0]
 * inverse); every output may be NULL.  eps: RealNVPFlow's logvar floor (flows.py:164,201). */
size_t dpf_gprior_canon_floats(int G, int n_features);
size_t dpf_gprior_packed_floats(int n_steps, int G, int n_features);
int dpf_gprior_pack(int n_steps, int G, int n_features, float bn_eps, const float *canon,
                    float *packed, dpf_stream_t stream);
int dpf_gprior_forward(int n_steps, int B, int G, int n_features, int mode, const int *codes,
                       const float *packed, const float *g, float *gs, float *mus, float *lvs,
                       float *sum_lv, float *g_out, float eps, dpf_stream_t stream);

/* ---- PointFlowNLL (lib/networks/losses.py:11-15) in one pass ---------------------------------
 * out[0] = 0.5 * ( sum_{b,c,n} [ sum_lv + lv0 + (s0 - mu0)^2 / exp(lv0) ] / B + log(2 pi) * C * N ).
 * s0 (B,C,N) contiguous: the cloud at the base of the flow (samples[0]); sum_lv (B,C,N) contiguous or
 * NULL: the flow's summed log-variances (dpf_flow_forward's sum_logvar); mu0 / lv0: the base
 * distribution (mus[0], logvars[0]) read through element strides (batch, channel, point) -- the
 * reference's stride-0 expansions (models.py:108-117) are never materialised.  Deterministic.
 * workspace: dpf_pointflow_nll_workspace_floats() fp32. */
size_t dpf_pointflow_nll_workspace_floats(void);
int dpf_pointflow_nll(int B, int C, int N, const float *s0, const float *mu0, long mu_sb, long mu_sc,
                      long mu_sn, const float *lv0, long lv_sb, long lv_sc, long lv_sn,
                      const float *sum_lv, float *workspace, float *out, dpf_stream_t stream);
/* Backward of dpf_pointflow_nll for a scalar upstream gradient grad_out[0] (device): d_s0 and d_sum_lv (B,C,N)
 * contiguous, and -- when non-NULL -- d_mu0 / d_lv0 as full (B,C,N) tensors (the caller's expand-backward reduces
 * them); any output pointer may be NULL.  One launch; with it the flow NLL of a training step (losses.py:48) is a single
 * autograd node over HIP kernels. */
int dpf_pointflow_nll_backward(int B, int C, int N, const float *s0, const float *mu0, long mu_sb, long mu_sc,
                               long mu_sn, const float *lv0, long lv_sb, long lv_sc, long lv_sn,
                               const float *grad_out, float *d_s0, float *d_sum_lv, float *d_mu0, float *d_lv0,
                               dpf_stream_t stream);

/* ---- latent prior flow, TRAINING mode (BatchNorm1d on the statistics of the B rows) -----------
 * replaces GlobalRNVPDecoder.forward under model.train() (decoders.py:21-38, flows.py:198-213)
 * and the backward autograd derives from it; 4 launches per step forward, 5 backward, all issued
 * by the one call.  canon: the UNPACKED canonical block of dpf_gprior_pack (the running statistics
 * in it are not read) when params_only == 0; with params_only != 0 the same without the two
 * running-statistics vectors of each net (W0 | bn.weight | bn.bias | W1 | b1: what an optimizer
 * updates, so a caller can keep all parameters in one buffer of exactly this layout and all
 * gradients in its twin).  codes, mode, eps as dpf_gprior_forward; B >= 2.
 * forward: g (B,G) -> gs, mus, lvs (S,B,G) DIRECT order, all required; save_h (S,B,2*n_features)
 *   the pre-BatchNorm activations and save_stats (S,2,2*n_features) = batch mean | biased variance per
 *   hidden unit (mu net, then logvar net) -- for the backward and for the caller's running-statistics
 *   update (nn.BatchNorm1d: momentum 0.1, unbiased variance).
 * backward: d_gs, d_mus, d_lvs (S,B,G) or NULL -> dg (B,G) and dcanon (same layout as canon: weight,
 *   BatchNorm affine and bias gradients, zeros in the running-statistics slots), both overwritten.
 * workspace: dpf_gprior_train_workspace_floats(B, G, n_features) fp32. */
size_t dpf_gprior_train_workspace_floats(int B, int G, int n_features);
int dpf_gprior_train_forward(int n_steps, int B, int G, int n_features, int mode, const int *codes,
                             int params_only, const float *canon, const float *g, float *gs, float *mus, float *lvs,
                             float *save_h, float *save_stats, float *workspace, float bn_eps,
                             float eps, dpf_stream_t stream);
int dpf_gprior_train_backward(int n_steps, int B, int G, int n_features, int mode, const int *codes,
                              int params_only, const float *canon, const float *g, const float *gs, const float *mus,
                              const float *lvs, const float *save_h, const float *save_stats,
                              const float *d_gs, const float *d_mus, const float *d_lvs, float *dg,
                              float *dcanon, float *workspace, float bn_eps, float eps,
                              dpf_stream_t stream);

/* BatchNorm running statistics after a training-mode forward (nn.BatchNorm1d: running = (1 - momentum) * running +
 * momentum * batch, unbiased batch variance; lib/networks/flows.py:27,30,35,42 in train()): all 8 n_layers BatchNorm1d layers
 * of the stack in one launch.  running_mean / running_var (8 n_layers, 64) and num_batches_tracked (8 n_layers, int64): rows
 * [0, 4 n_layers) the FiLM nets in (layer, branch, w|b) order with batch statistics film_mean / film_uvar (4 n_layers, 64),
 * rows [4 n_layers, 8 n_layers) BN0, BN1 of (layer, branch) from the `stats` block of dpf_flow_train_forward. */
int dpf_flow_train_update_running(int n_layers, double momentum, const float *film_mean, const float *film_uvar,
                                  const float *stats, float *running_mean, float *running_var,
                                  long long *num_batches_tracked, dpf_stream_t stream);

/* ---- FiLM conditioner nets of the coupling stack, training mode (csrc/film_train.hip) ---------------------
 * Replaces, for model.train(), the K = 4 n_layers per-cloud conditioner sub-nets of CondRealNVPFlow3D
 * (lib/networks/flows.py:33-45, 68-80: Linear(G, 64, bias=False) . BatchNorm1d over the B clouds . Swish . Linear(64, 64))
 * and what autograd derives from them (lib/networks/training.py:55) -- ~12 tensor-op launches forward and ~25 backward per
 * optimizer step in the reference's formulation, ONE launch each way here.  All arrays fp32, contiguous:
 *   g (B, G); W0 (K, 64, G); gamma, beta (K, 64); W1 (K, 64, 64) [out][in]; b1 (K, 64)
 * forward:  fm (K, B, 64) = the stack's conditioner input (dpf_flow_train_forward's `fm`); saved for the backward: xhat
 *   (K, B, 64) and rstd (K, 64); mean / uvar (K, 64) = batch mean and UNBIASED batch variance for the running-statistics
 *   update (nn.BatchNorm1d, momentum 0.1).
 * backward: dfm (K, B, 64) -> dW0 (K, 64, G), dgamma, dbeta (K, 64), dW1 (K, 64, 64), db1 (K, 64): overwritten, or added to
 *   (accumulate != 0: the flat gradient store); dg_part (K, B, G) or NULL: every sub-net's share of d loss / d g (the caller
 *   sums over K).  B <= dpf_film_train_max_batch() and G % 4 == 0, else DPF_ENOSUP (the caller keeps the tensor ops). */
int dpf_film_train_max_batch(void);
int dpf_film_train_forward(int K, int B, int G, const float *g, const float *W0, const float *gamma, const float *beta,
                           const float *W1, const float *b1, float bn_eps, float *fm, float *xhat, float *rstd,
                           float *mean, float *uvar, dpf_stream_t stream);
int dpf_film_train_backward(int K, int B, int G, const float *g, const float *W0, const float *gamma, const float *beta,
                            const float *W1, const float *xhat, const float *rstd, const float *dfm, float *dW0,
                            float *dgamma, float *dbeta, float *dW1, float *db1, float *dg_part, int accumulate,
                            dpf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DPF_HIP_H */
