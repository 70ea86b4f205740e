/*
 * dgg_hip.h -- C ABI of libdgg_hip.so, the MI355X (gfx950) implementation of the Differentiable Graph
 * Generator hot path of avishkarsaha/learning-adaptive-neighborhoods-for-gnns.
 *
 * The reference has no FFI: its hot path is a chain of ATen calls on dense [N,N] tensors inside
 * DGG_LearnableK_debug.forward (reference dgm.py:1178-1292) and the graph-conv layers of model.py.  Each entry
 * point below names the reference lines it replaces.  All pointers are DEVICE pointers (hipMalloc / torch
 * caching allocator), fp32 / int32 / int64, row-major and contiguous; `stream` is a hipStream_t passed as
 * void* (NULL = default stream).  Every function returns 0 on success or a DGG_ERR_* code, with a message
 * available from dgg_last_error() (thread-local).  No function allocates, synchronises or keeps state; work
 * buffers are caller-provided, so calls may be captured in a hipGraph.
 *
 * Sparse adjacency layout (ELL): idx int32 [rows, K], K <= 64, entry r of a row = its rank-r candidate in
 * descending score order (ties: lower column first); idx = -1 marks an empty slot.  Row-sharded multi-GPU
 * use passes the local row range [row0, row1) together with GLOBAL-length per-node arrays.
 *
 * Reference-side binding: see INTEGRATION.md (ctypes stub a maintainer would add next to dgm.py).
 */
#ifndef DGG_HIP_H
#define DGG_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGG_OK 0
#define DGG_ERR_ARG 1          /* invalid argument        (reference: bare assert, dgm.py:1187-1188)            */
#define DGG_ERR_UNSUPPORTED 2  /* unsupported shape/mode  (reference: Exception("mode not found"), dgm.py:1727) */
#define DGG_ERR_HIP 3          /* HIP runtime / launch failure                                                  */

/* noise_mode: how the Gumbel perturbation of dgm.py:1211-1229 is supplied */
#define DGG_NOISE_NONE 0      /* args.perturb_edge_prob == False                                      */
#define DGG_NOISE_EXPLICIT 1  /* caller passes G (dense, ld = ldG): what gumbel_sample(log_p, G) takes */
#define DGG_NOISE_HASH 2      /* counter-based Gumbel(0,0.3) keyed on (seed, i, j)                    */
#define DGG_NOISE_HASH_SYM 3  /* keyed on (seed, min(i,j), max(i,j)), zero diagonal (dgm.py:1216-1223) */
#define DGG_NOISE_RANKED 4    /* counter-based, same iid Gumbel(0,0.3) law, generated per row in decreasing order
                               * (Renyi spacings + keyed column permutation): all-pairs top-K in O(N*~150) */
#define DGG_NOISE_RANKED_SYM 5 /* symmetric (G_ij = G_ji, zero diagonal: symmetric_noise=True, dgm.py:1216-1223), same law per unordered
                               * pair; every pair is owned by one endpoint, which generates the noises of its pairs in decreasing
                               * order: all-pairs top-K in O(N*~300) instead of the N^2 hash sweep of DGG_NOISE_HASH_SYM.
                               * All-pairs candidates only (dgg_allpairs_topk) */

/* activations of dgg_linear_*: */
#define DGG_ACT_NONE 0
#define DGG_ACT_LEAKY 1 /* nn.LeakyReLU(0.01), dgm.py:1099 */
#define DGG_ACT_RELU 2  /* torch.relu, model.py:598 */

/* select_top_k modes (dgm.py:1402-1435) */
#define DGG_MODE_K_TIMES_EDGE_PROB 0
#define DGG_MODE_K_ONLY 1

const char *dgg_last_error(void);
int dgg_abi_version(void);

/* ---- dense layers (fp32 MFMA) ------------------------------------------------------------------------------
 * y[N,out] = act(x[N,d] W^T + b).  w_layout 0: W[out][d] (nn.Linear: node_encode_for_edges / node_encode_for_k,
 * dgm.py:1097-1100, 1123-1126, applied at 1609 / 1566); w_layout 1: W[d][out] (GCNConv.W, model.py:583, 594-598;
 * GraphConvolution.weight, model.py:25, 41).  b may be NULL. */
int dgg_linear_fwd(const float *x, int64_t N, int d, const float *W, const float *b, int out, int w_layout, int act,
                   float *y, void *stream);
/* autograd of the above.  ws: dgg_linear_bwd_ws_floats(N, d, out) floats of workspace.  dx (nullable) is
 * overwritten; dW (layout of W) and db (nullable) are accumulated into. */
size_t dgg_linear_bwd_ws_floats(int64_t N, int d, int out);
int dgg_linear_bwd(const float *x, int64_t N, int d, const float *W, int out, int w_layout, int act, const float *y,
                   const float *dy, float *dx, float *dW, float *db, float *ws, void *stream);
/* stacks the weights of nseg <= 8 layers (layout 0: [out, d] nn.Linear; 1: [d, out] GCNConv, model.py:583) and their biases (NULL:
 * zeros) into the row-stacked form dgg_linear_fwd_multi reads: Wcat [sum out, d], bcat [sum out]; one launch */
int dgg_linear_pack_weights(int nseg, const float *const *W, const float *const *b, const int *seg_out, const int *seg_layout, int d,
                            float *Wcat, float *bcat, void *stream);
/* Several layers that read the SAME input in one pass over it (node_encode_for_edges + node_encode_for_k of one DGG,
 * dgm.py:1097-1100 / 1123-1126 applied at 1609 / 1566, and the GCNConv projection x W, model.py:596): Wcat [sum out_s, d] /
 * bcat [sum out_s] (nullable) = the layers' weights in nn.Linear layout stacked by rows; layer s has out_s outputs (multiples
 * of 32, at most 256 in total, not 224), activation act_s and destination y_s [N,out_s].  seg_out / seg_act / y are HOST arrays.
 * Bit-identical to nseg calls of dgg_linear_fwd. */
int dgg_linear_fwd_multi(const float *x, int64_t N, int d, const float *Wcat, const float *bcat, int nseg, const int *seg_out,
                         const int *seg_act, float *const *y, void *stream);
/* Their weight gradients in one pass over the shared input B [N,M2] (M2 <= 128):
 *   C_s[M1_s,M2] += (A_s * act_s'(Y_s))^T B,  colsum_s[M1_s] += column sums of the masked A_s (bias gradient; nullable)
 * A_s = d loss / d y_s [N,M1_s] (M1_s multiples of 32, at most 256 in total), Y_s = y_s (nullable: no activation),
 * c_layout_s 0: C_s stored [M1_s][M2], 1: [M2][M1_s] (GCNConv.W).  A / M1 / Y / act / C / c_layout / colsum are HOST arrays.
 * ws: dgg_gemm_tn_multi_ws_floats(N, sum M1_s, M2) floats. */
size_t dgg_gemm_tn_multi_ws_floats(int64_t N, int M1_total, int M2);
int dgg_gemm_tn_multi(int nseg, const float *const *A, const int *M1, const float *const *Y, const int *act, const float *B,
                      int64_t N, int M2, float *const *C, const int *c_layout, float *const *colsum, float *ws, void *stream);
/* Several independent small products over the same N rows in ONE launch: C_p[M1_p,M2_p] += A_p[N,M1_p]^T B_p[N,M2_p], colsum_p
 * (nullable) += column sums of A_p -- the k-net's three weight gradients (k_embed 32x65, k_mu 16x32, k_project 1x16 at h = 64:
 * autograd of dgm.py:1576-1577, 2051-2063).  At most 256 rows of output in total (each M1_p padded to 32), M2_p <= 128;
 * A / M1 / B / M2 / C / colsum are HOST arrays; ws: dgg_gemm_tn_multi_ws_floats(N, padded total M1, 128) floats. */
int dgg_gemm_tn_pairs(int npair, const float *const *A, const int *M1, const float *const *B, const int *M2, int64_t N, float *const *C,
                      float *const *colsum, float *ws, void *stream);
/* C[M1,M2] += A[N,M1]^T B[N,M2] (c_layout 1: C stored [M2][M1]); colsum (nullable,[M1]) += column sums of A.
 * Weight gradients of the per-node layers (autograd of dgm.py:1576-1577).  ws: dgg_gemm_tn_ws_floats(N, M1, M2)
 * floats (per-chunk partial blocks, summed by a second kernel: no same-address atomics storm). */
size_t dgg_gemm_tn_ws_floats(int64_t N, int M1, int M2);
int dgg_gemm_tn_acc(const float *A, const float *B, int64_t N, int M1, int M2, float *C, int c_layout, float *colsum,
                    float *ws, void *stream);

/* ---- learned degree k (k_estimate_net, dgm.py:1472-1586; LearnableKEncoder.forward, dgm.py:2051-2063) ------ */
/* mu_sd[0] = mean(deg), mu_sd[1] = unbiased std(deg)   (dgm.py:1568-1570; deg replaces in_adj.to_dense().sum(-1)) */
int dgg_degree_stats(const float *deg, int64_t N, float *mu_sd, void *ws, void *stream);
size_t dgg_degree_stats_ws_bytes(void); /* device workspace of dgg_degree_stats (partial sums) */
/* mode "x" (dgm.py:1562-1586): xk = leaky(node_encode_for_k(x)) [N,h]; W1/b1 = k_embed.0 [h2][h+1]; Wmu/bmu =
 * k_net.k_mu [h4][h2]; Wp/bp = k_net.k_project [h4]/[1].  Saves (nullable) z [N,h2], u [N] (pre-relu), feat
 * [N,h+1] = [xk | normalised degree] for the backward. */
int dgg_knet_x_fwd(const float *xk, int64_t N, int h, const float *deg, const float *mu_sd, const float *W1,
                   const float *b1, int h2, const float *Wmu, const float *bmu, int h4, const float *Wp, const float *bp,
                   float *k, float *z_save, float *u_save, float *feat_save, void *stream);
/* The k-net of mode "x" (k_estimate_net dgm.py:1562-1586 + LearnableKEncoder dgm.py:2051-2063) on the fp32 matrix cores, latent_dim h
 * in {16, 32, 64}; h2 = h/2, h4 = h/4.  Forward: k [N] and u [N] (the pre-ReLU output: all the backward needs besides xk); same bits
 * as dgg_knet_x_fwd. */
int dgg_knet_x_fwd_mfma(const float *xk, int64_t N, int h, const float *deg, const float *mu_sd, const float *W1, const float *b1,
                        const float *Wmu, const float *bmu, const float *Wp, const float *bp, float *k, float *u_save, void *stream);
/* Backward in ONE pass over xk (layer 1 is re-run): dxk [N,h] and every parameter gradient -- gW1 [h2, h+1] (k_embed.0.weight), gb1 [h2],
 * gWmu [h4, h2], gbmu [h4], gWp [h4], gbp [1] -- are OVERWRITTEN (weight-gradient partials leave the kernel as one plain-store slab per
 * workgroup, a reduce launch sums them and runs the parameter-sized tail; no float atomics, nothing to zero).  ws:
 * dgg_knet_x_bwd_ws_bytes bytes of scratch (0: width not supported).  xk, dxk 16-byte aligned.  out_act = 1: dxk is returned multiplied by
 * LeakyReLU'(xk), i.e. as the gradient of the PRE-activation of the layer that produced xk (node_encode_for_k, dgm.py:1123-1126), so
 * that its weight-gradient product needs no second pass over xk for the mask. */
size_t dgg_knet_x_bwd_ws_bytes(int64_t N, int h);
int dgg_knet_x_bwd_reg(const float *xk, int64_t N, int h, const float *deg, const float *mu_sd, const float *W1, const float *b1,
                       const float *Wmu, const float *bmu, const float *Wp, const float *u, const float *dk, float *dxk, float *gW1,
                       float *gb1, float *gWmu, float *gbmu, float *gWp, float *gbp, int out_act, void *ws, void *stream);
/* per-node part of the backward: dk -> dkp [N], dm [N,h4], dpre1 [N,h2], dxk [N,h], m [N,h4] (recomputed) */
int dgg_knet_x_bwd_nodes(int64_t N, int h, const float *mu_sd, const float *W1, int h2, const float *Wmu, int h4,
                         const float *Wp, const float *bmu, const float *z, const float *u, const float *dk, float *dkp,
                         float *dm, float *dpre1, float *dxk, float *m_out, void *stream);
/* degree-only modes in general (dgm.py:1492-1526): nd = (deg - mu) / (sd + eps), k = relu(k_project(k_mu(
 * input_degree_project(nd))) * sd + mu) + 1.  "input_deg": mu_sd = NULL, constants dmean/dstd = args.deg_mean/deg_std,
 * eps = 1e-5; "learn_normalized_degree": mu_sd = device [2] from dgg_degree_stats, eps = 0.  u_save (nullable): pre-relu */
int dgg_knet_deg_fwd(const float *deg, int64_t N, const float *mu_sd, float dmean, float dstd, float eps, const float *Wd,
                     const float *bd, const float *Wmu, const float *bmu, int h4, const float *Wp, const float *bp, float *k,
                     float *u_save, void *stream);
/* latent_dim > 128 (PPI configuration: 2048): the three layers of mode "x" run as GEMMs (dgg_linear_fwd/bwd) on
 * feat [N,h+1] = [xk | nd] -- the same k-ordered fmaf chains -- between these elementwise ends:
 * feat assembly; u = kp*sd + mu, k = relu(u) + 1; dkp = dk * sd * [u > 0] */
int dgg_knet_feat(const float *xk, const float *deg, const float *mu_sd, int64_t N, int h, float *feat, void *stream);
int dgg_knet_out_fwd(const float *kp, const float *mu_sd, int64_t N, float *k, float *u, void *stream);
int dgg_knet_out_bwd(const float *u, const float *dk, const float *mu_sd, int64_t N, float *dkp, void *stream);
/* backward: S[0] += sum_i dkp_i, S[1] += sum_i dkp_i nd_i with dkp = dk * sd * [u > 0]; the net is affine in nd, so every
 * parameter gradient is a combination of these two sums (see dgg_amd/dgm.py, _KnetDegFn) */
int dgg_knet_deg_bwd_sums(const float *deg, int64_t N, const float *mu_sd, float dmean, float dstd, float eps, const float *u,
                          const float *dk, float *S, void *stream);
/* mode "input_deg" (dgm.py:1509-1526) alone: Wd/bd = input_degree_project [3]/[3]; Wmu [h4][3] */
int dgg_knet_input_deg_fwd(const float *deg, int64_t N, float dmean, float dstd, const float *Wd, const float *bd,
                           const float *Wmu, const float *bmu, int h4, const float *Wp, const float *bp, float *k,
                           void *stream);

/* ---- pair scoring + per-row top-K ---------------------------------------------------------------------------
 * edge_prob_net "u-v-dist" (dgm.py:1607-1627) + perturbation (dgm.py:1211-1229) + torch.sort (dgm.py:1404), kept
 * to the K best per row.  xp [N,h] = leaky(node_encode_for_edges(x)); t = -0.05 (dgm.py:1618).
 * All-pairs candidates (complete in_adj): rows [row0,row1) of the N x N score matrix; outputs [row1-row0, K].
 * algo: 0 auto, 1 exhaustive, 2 the MFMA sweep (unperturbed scores), 4 guess-and-verify (per-pair hash noise); identical results.
 * workspace: dgg_allpairs_workspace_bytes().
 * k_limit (nullable, [row1-row0]): the learned k of each row.  The soft top-k ramp (dgm.py:1412-1420) is exactly 0.0f
 * for ranks r >= ceil(k + 8.5), so with k_limit the ranks from ceil(k + 8.5) + 1 on are returned as idx = -1, val = 0
 * (weights, outputs and gradients are unchanged) and the ranked-noise search stops as soon as the kept ranks are
 * settled. */
int dgg_allpairs_topk(const float *xp, int64_t N, int h, int64_t row0, int64_t row1, float t, int noise_mode,
                      const float *G, int64_t ldG, uint32_t s0, uint32_t s1, int K, int32_t *idx, float *val,
                      const float *k_limit, int algo, void *workspace, size_t ws_bytes, void *stream);
/* Walk statistics of the ranked search (noise_mode DGG_NOISE_RANKED): how deep a row walks depends on the DATA -- about
 * L exp(spread of 0.05 ||xp_i - xp_j|| over the row / 0.3) ranks -- so callers measure it, or estimate it on a row sample before they
 * choose this generator over the per-pair hash sweep.  Runs the search of dgg_allpairs_topk on every `stride`-th row of [row0, row1),
 * each walk cut after `max_blocks` blocks of 64 ranks (0: no cut), and writes nothing but six 64-bit counters:
 * probe[0..4] += rows walked, blocks visited, candidates gathered, candidates scored in full, rows cut by the budget;
 * probe[5] = max(probe[5], blocks of a row).  The caller zeroes probe. */
int dgg_allpairs_ranked_probe(const float *xp, int64_t N, int h, int64_t row0, int64_t row1, float t, uint32_t s0, uint32_t s1,
                              const float *k_limit, int stride, int max_blocks, unsigned long long *probe, const float *lpub, void *stream);
/* dgg_allpairs_topk (noise_mode DGG_NOISE_RANKED, K = 64, learned k required) fused with dgg_softk_fwd: the ramp is applied to the
 * settled list while it is still in registers; additionally writes w [row1-row0,64] and rs [row1-row0].  Same bits as the two calls. */
int dgg_allpairs_topk_ranked_softk(const float *xp, int64_t N, int h, int64_t row0, int64_t row1, float t, uint32_t s0, uint32_t s1,
                                   const float *k, int mode, int32_t *idx, float *val, float *w, float *rs, void *stream);
/* the same, with the noise seed {s0, s1} read from DEVICE memory when the kernel starts: a captured hipGraph of the step draws fresh
 * noise on every replay once the caller advances seed_dev between replays (the reference samples fresh noise per forward, dgm.py:1226) */
int dgg_allpairs_topk_ranked_softk_dseed(const float *xp, int64_t N, int h, int64_t row0, int64_t row1, float t, const uint32_t *seed_dev,
                                         const float *k, int mode, int32_t *idx, float *val, float *w, float *rs, void *stream);
/* the same two calls with the rows' upper bounds of log p_ij over their OTHER nodes j != i (lpub [row1-row0], nullable; from
 * dgg_allpairs_rowmin_bound): the walk's stop test and its per-candidate cut use  G + lpub[i]  in place of the distance-free  G + 1e-8  --
 * legitimate because the walk visits the row's own column FIRST (the ranked generator keeps the diagonal outside its rank sequence) --
 * so that fewer ranks are walked when the latent distances spread over several noise scales; the result is the same bits.  lpub is
 * also a trailing argument of dgg_allpairs_ranked_probe, dgg_allpairs_topk_ranked_wide and dgg_allpairs_topk_anywide.
 * seed_dev != NULL: the seed is read from device memory (s0, s1 ignored). */
int dgg_allpairs_topk_ranked_softk_lp(const float *xp, int64_t N, int h, int64_t row0, int64_t row1, float t, uint32_t s0, uint32_t s1,
                                      const uint32_t *seed_dev, const float *lpub, const float *k, int mode, int32_t *idx, float *val, float *w,
                                      float *rs, void *stream);
/* ---- rows wider than 64 ranks: CHUNKED rows ------------------------------------------------------------------------------------
 * The learned degree k = relu(k_net * std + mean) + 1 is unbounded (dgm.py:1580-1584) and select_top_k ramps over the whole dense row
 * (dgm.py:1402-1421): a row carries weight on its first ceil(k_i + 8.5) ranks whatever k_i.  Node i then owns the M_i = ceil(L_i / 64)
 * consecutive 64-entry CHUNKS [cptr[i], cptr[i+1]) of idx / val / w / ahat ([chunks, 64] arrays), L_i = ceil(k_i + 8.5) + 1; rank r of the
 * row is entry r % 64 of its chunk r / 64.  With every M_i = 1 the layout IS the [rows, 64] list of the calls above, and every chunk is
 * a row of that layout to the entry points that take `cnode` (node of every chunk) below.
 * dgg_chunk_layout: k [rows] -> cptr int32 [rows+1], cnode int32 [ccap], meta int32 [4 + 384] = {chunks NEEDED in total, max M_i, flags, 0,
 * scratch of the two-pass scan};
 * flags bit 0: a row needs more than 64 * maxm ranks (the row is cut there: harmless when 64 * maxm covers every column of the graph,
 * otherwise callers must raise; maxm <= 2^20), bit 2: a learned degree is NaN, bit 1: more than ccap chunks -- cptr is then CLAMPED to
 * ccap (rows beyond it own no chunk, a row that straddles it is cut), so that no consumer of the layout leaves arrays of ccap chunks;
 * call again with a larger capacity.  sticky (nullable, int32[1]): the flags are also ORed into it -- meta is rewritten by every call
 * (every replay of a captured hipGraph), sticky keeps every overflow until the host clears it. */
int dgg_chunk_layout(const float *k, int64_t rows, int maxm, int64_t ccap, int32_t *cptr, int32_t *cnode, int32_t *meta, int32_t *sticky,
                     void *stream);
/* dgg_allpairs_topk_ranked_softk[_dseed] on chunked rows: the ranked search settles L_i ranks of row i in up to `maxm` descending 64-lane
 * lists per wavefront (maxm >= max M_i; rows of more than 32 chunks are SKIPPED: dgg_allpairs_topk_anywide settles them); idx / val / w [chunks,64], rs [row1-row0] (lane-wise sums over the chunks, then the wavefront
 * butterfly).  ccap >= cptr[rows]: the chunks the arrays hold (a capacity fixed ahead of the learned degrees, e.g. inside a captured
 * hipGraph); chunks beyond the last one are written empty (idx -1, weight 0) and dgg_chunk_layout gives them node 0, so every consumer
 * may walk all ccap chunks.  seed_dev != NULL: the seed is read from device memory, s0 / s1 are ignored.  w (and rs) may be NULL. */
int dgg_allpairs_topk_ranked_wide(const float *xp, int64_t N, int h, int64_t row0, int64_t row1, float t, uint32_t s0, uint32_t s1,
                                  const uint32_t *seed_dev, const float *k, int mode, int maxm, const int32_t *cptr, int64_t ccap, int32_t *idx,
                                  float *val, float *w, float *rs, const float *lpub, void *stream);
/* Chunked rows of ANY width and every noise generator (dgg_topk_anywide.hip; reference dgm.py:1402-1421 on the dense row, 1580-1584 the
 * unbounded learned degree, 1211-1231 perturb_edge_prob / symmetric_noise): row i keeps its L_i = ceil(k_i + 8.5) + 1 best columns by the
 * perturbed score in the chunks [cptr[i], cptr[i+1]) of idx / val / w [ccap,64], rs [row1-row0], with the ramp and the row sums of
 * dgg_allpairs_topk_ranked_wide (same bits as the oracle).  noise_mode 0 (unperturbed), 2 (per-pair hash), 3 (symmetric per-pair hash):
 * every row of [row0,row1), min_m = 0, spare chunks of a fixed capacity written empty.  noise_mode 4 (ranked generator): only the rows of
 * MORE than min_m chunks (call dgg_allpairs_topk_ranked_wide for the others: with maxm > 32 it leaves the rows beyond 32 chunks to this
 * entry, min_m = 32).  Every row owns a threshold buffer of 128 keys per chunk in `workspace` (dgg_allpairs_anywide_ws_bytes(ccap, rows, N, h)
 * bytes, which also hold the front ends' candidate lists and, for unperturbed scores, the scratch -- 64 keys per node, at most 32 MB -- of the
 * segmented re-scan of the rows the radius sweep cannot settle); a candidate is dropped only once L_i better ones of its row are known.
 * maxm >= max M_i (sizes the sort's LDS). */
size_t dgg_allpairs_anywide_ws_bytes(int64_t ccap, int64_t rows, int64_t N, int h);
int dgg_allpairs_topk_anywide(const float *xp, int64_t N, int h, int64_t row0, int64_t row1, float t, int noise_mode, uint32_t s0, uint32_t s1,
                              const uint32_t *seed_dev, const float *k, int mode, int maxm, int min_m, const int32_t *cptr, int64_t ccap,
                              int32_t *idx, float *val, float *w, float *rs, const float *lpub, void *workspace, size_t ws_bytes, void *stream);
/* lpub [row1-row0]: for every row i a rigorous UPPER bound of log p_ij = log(exp(t ||xp_i - xp_j||) + 1e-8) over all j != i (reference
 * dgm.py:1618-1623: the scores before the perturbation), from an fp16-MFMA lower bound of the squared distance to the row's nearest OTHER
 * node (one sweep over all N^2 pairs on the matrix cores, dgg_topk_sweep.hip).  The noise generators bound a pair's log-score by its
 * noise alone (log p' <= G + 1e-8); with lpub they use G + lpub[i]: fewer ranks walked / fewer candidates scored on latents whose
 * distances spread over several noise scales.  latent_dim in {16, 32, 64, 128}, t < 0; workspace: dgg_allpairs_rowmin_ws_bytes. */
size_t dgg_allpairs_rowmin_ws_bytes(int64_t rows, int64_t N, int h);
int dgg_allpairs_rowmin_bound(const float *xp, int64_t N, int h, int64_t row0, int64_t row1, float t, float *lpub, void *workspace, size_t ws_bytes,
                              void *stream);
/* bytes of device workspace the pruned path needs for (N, h): a bf16 copy of xp plus discounted squared norms;
 * 0 when the pruned path does not apply (explicit noise, K != 64, latent_dim not in {16,32,64,128}) */
size_t dgg_allpairs_workspace_bytes(int64_t N, int h, int noise_mode, int K);
/* diagnostics of the unperturbed path (noise_mode 0, N >= 8192; the reference default perturb_edge_prob=False, dgm.py:1230-1231):
 * byte offset inside the workspace of its control block { int32 rows_redone_by_the_fallback; int32 stats_on; uint64 phase_a_hits,
 * phase_a_hits_inside_the_tight_radius, phase_b_hits } (the three sums only with DGG_SWEEP_STATS=1), readable after the stream
 * has drained */
size_t dgg_allpairs_sweep_ctl_offset_bytes(int64_t rows, int64_t N, int h);
/* diagnostics of the ranked symmetric path (noise_mode 5): byte offset inside the workspace of its control block { float pilot_sum;
 * int32 rows_redone_by_tier_2; float guessed_threshold; int32 rows_redone_by_tier_3; int32 err; int32 pad[3]; uint64 emitted_pairs,
 * scored_candidates (both only with DGG_RSYM_STATS=1), ... }, readable after the stream has drained.  err != 0: more rows needed the
 * dense tier than the workspace holds -- their idx is -1 and the caller must not use the result (the host mirror raises) */
size_t dgg_allpairs_rsym_ctl_offset_bytes(int64_t rows, int64_t N);
/* The debug class's LITERAL dgg_hard output, reference dgm.py:1294-1311 (return_hard_or_soft): ones.scatter_(-1, idxs, edge_p > 0.5)
 * with edge_p the already unsorted soft adjacency and idxs the sort permutation of the perturbed scores over ALL N columns, then
 * (hard - soft).detach() + soft.  Opt-in compatibility path, N <= 8192 (full per-row ranking; O(N^2 log^2 N)).  Candidates: all pairs
 * (rowptr NULL) or the CSR entries (non-candidates: probability 0, perturbed exp(log(1e-8) + G)); scorer u-v-dist (dgm.py:1618-1623).
 * idx / w [N,K]: the soft ELL adjacency (dgg_softk_fwd).  Outputs [N,K], compacted to the front, -1 / 0 padded: hidx = columns of the
 * ones, hval = their forward value (1 - s) + s, hsrc = the soft ELL slot at that column (-1: none) that receives the gradient. */
int dgg_literal_hard_fwd(const float *xp, int64_t N, int h, const int64_t *rowptr, const int32_t *col, float t, int noise_mode,
                         const float *G, int64_t ldG, uint32_t s0, uint32_t s1, const int32_t *idx, const float *w, int K, float threshold,
                         int32_t *hidx, float *hval, int32_t *hsrc, void *stream);
/* its backward: dw [N,K] (OVERWRITTEN) = cotangent of hval routed to the soft slots hsrc */
int dgg_literal_hard_bwd(const int32_t *hsrc, const float *g, int64_t N, int K, float *dw, void *stream);
/* candidates = stored entries of in_adj as CSR (the live class's semantics, dgm.py:1613-1614) */
int dgg_edgelist_topk(const float *xp, int64_t N, int h, const int64_t *rowptr, const int32_t *col, float t,
                      int noise_mode, const float *G, int64_t ldG, uint32_t s0, uint32_t s1, int K, int32_t *idx,
                      float *val, void *stream);
/* dgg_edgelist_topk + dgg_softk_fwd in ONE launch (same bits as the two calls; dgm.py:1404-1420 on the sorted list while it is
 * still in registers): k [N] learned degrees, mode as dgg_softk_fwd -> additionally w [N,K], rs [N].  latent_dim 16 / 32 / 64 / 128.
 * overflow (nullable, int32[1] on the device, ORed into, never cleared): becomes 1 when some row has more than K candidates AND a
 * learned degree with k + 8.5 > K, i.e. when the K-wide list dropped a rank the reference still weights (dgm.py:1410-1420). */
int dgg_edgelist_topk_softk(const float *xp, int64_t N, int h, const int64_t *rowptr, const int32_t *col, float t,
                            int noise_mode, const float *G, int64_t ldG, uint32_t s0, uint32_t s1, int K, const float *k,
                            int mode, int32_t *idx, float *val, float *w, float *rs, int32_t *overflow, void *stream);

/* ---- edge-MLP scorers on a candidate edge list (dgm.py:1628-1725: u-v-A_uv, u-v-deg, u-v-deg-dist, edge_conv, A_uv) --
 * The reference evaluates sigmoid(W2 act(W1 [x_u, x_v, extras] + b1) + b2) per edge (edge_encode, dgm.py:1101-1105;
 * edge_conv_* 1107-1109; adj_project 1117).  By linearity the first layer is split into per-node products
 * AB = xp [Wa | Wb]^T ([N, 2*hw], computed with dgg_linear_fwd) and per-edge terms:
 *   z_o = A[u][o] + B[v][o] (+ deg_u wdu_o + deg_v wdv_o) (+ ex_e wex_o) + b1_o;  p_e = sigmoid(sum_o act(z_o) w2_o + b2)
 * erow/col [E]: the coalesced COO candidate list (row-major); deg (nullable): row sums of in_adj (dgm.py:1653);
 * ex_mode 0 none, 1 ex_in[e] (a_uv, dgm.py:1636), 2 exp(t_ex ||xp_u - xp_v||) (dgm.py:1684-1686, needs xp);
 * act 1 LeakyReLU, 0 identity; b2 is a device pointer (the parameter).  ex_out (nullable, [E]): the extra used. */
int dgg_edge_mlp_fwd(const float *AB, const float *xp, int64_t N, int h, int hw, const int32_t *erow, const int32_t *col,
                     int64_t E, const float *deg, const float *ex_in, int ex_mode, float t_ex, const float *wdu,
                     const float *wdv, const float *wex, const float *b1, const float *w2, const float *b2, int act,
                     float *p_edge, float *ex_out, void *stream);
/* perturbation (dgm.py:1211-1229) + torch.sort (dgm.py:1404) kept to the K best per row, on given edge probabilities
 * p_edge [E] (CSR order, columns of a row ascending).  eid [N,K]: index of each selected candidate in col[] (-1 empty) */
int dgg_edgelist_topk_p(const float *p_edge, int64_t N, const int64_t *rowptr, const int32_t *col, int noise_mode,
                        const float *G, int64_t ldG, uint32_t s0, uint32_t s1, int K, int32_t *idx, float *val, int32_t *eid,
                        void *stream);
/* autograd of dgg_edge_mlp_fwd for the selected entries: dval (wrt the stored score) -> dAB [N,2*hw] and
 * dpar [5*hw+1] = [dwdu | dwdv | dwex | db1 | dw2 | db2] (both ACCUMULATED into: caller zeroes), dex [N,K] (nullable,
 * overwritten; gradient wrt the per-edge extra).  ex [E] as written by the forward (nullable when ex_mode was 0). */
int dgg_edge_mlp_bwd(const float *AB, int64_t N, int hw, const int64_t *rowptr, const int32_t *idx, const int32_t *eid,
                     const float *val, const float *dval, int K, const float *deg, const float *ex, const float *wdu,
                     const float *wdv, const float *wex, const float *b1, const float *w2, const float *b2, int act,
                     int perturb, float *dAB, float *dpar, float *dex, void *stream);
/* the same for an ELL block whose payload partition (dgg_partp_build of (idx, w), with its entry -> record map: dgg_partp_has_map(N))
 * is at hand: the neighbour-side sums d B_j WITHOUT float atomics -- every selected entry's term is stored as a row of dz_rec in
 * record order (the records of a destination node are consecutive) and a second kernel adds each node's rows.  w [N,K]: the weights
 * the partition was built from; dz_rec: dz_rows * hw floats of scratch, dz_rows >= the number of records (the number of candidate
 * edges bounds it); hw a multiple of 4. */
int dgg_edge_mlp_bwd_partp(const float *AB, int64_t N, int hw, const int32_t *idx, const int32_t *eid, const float *val, const float *dval,
                           const float *w, int K, const float *deg, const float *ex, const float *wdu, const float *wdv, const float *wex,
                           const float *b1, const float *w2, const float *b2, int act, int perturb, const void *partp_ws, int64_t ncols,
                           float *dz_rec, int64_t dz_rows, float *dAB, float *dpar, float *dex, void *stream);

/* ---- CSR-valued adjacency (variable row length) + the `DGG` class "for ICLR" (dgm.py:1730-1815) ---------------------
 * `DGG.forward` keeps every candidate edge (weight rank * (ramp + 1), dgm.py:1804-1807), so its output has the sparsity
 * of in_adj (rows up to 168 wide on Cora) and is carried as values [E] on the CSR pattern (rowptr int64 [N+1], col
 * int32 [E], columns of a row ascending).  dgg_edge_mlp_bwd takes the same pattern through its rowptr argument. */
/* dgm.py:1791-1812: S_i = sum_j rank_ij; k_i = degree_decoder(S_i) = leaky(S_i w + b) (w, b device pointers);
 * pos_e = rank position of edge e in its row under torch.sort(descending) (ties: lower column first);
 * out_e = rank_e * ((1 - 0.5 (1 + tanh(pos_e - k_i))) + 1) */
int dgg_csr_rank_ramp_fwd(const float *p, const int64_t *rowptr, const int32_t *col, int64_t N, const float *w, const float *b,
                          float *out, float *S, float *k, int32_t *pos, void *stream);
/* g = d loss / d out -> dp [E] (direct + through S -> k), dkz [N] = d loss / d (S_i w + b) (dw = <dkz, S>, db = sum dkz) */
int dgg_csr_rank_ramp_bwd(const float *p, const int64_t *rowptr, int64_t N, const float *w, const float *b, const float *S,
                          const float *k, const int32_t *pos, const float *g, float *dp, float *dkz, void *stream);
/* select_top_k of DGG_LearnableK_debug (dgm.py:1402-1435) on rows of ANY width: the CSR counterpart of dgg_edgelist_topk_p +
 * dgg_softk_fwd for graphs whose rows have more candidates than the ELL width and whose learned degrees may exceed it.
 * p [E] edge probabilities (edge_prob_net, dgm.py:1607-1725), k [N]; noise_mode DGG_NOISE_NONE / EXPLICIT (G [N, ldG]) / HASH / HASH_SYM
 * (dgm.py:1213-1229); mode 0 k_times_edge_prob, 1 k_only.  -> w [E] = p' * ramp(pos - k) (mode 0) or the ramp, pp [E] = p', pos [E]
 * = position of the entry in its row's descending sort (ties: lower column first). */
int dgg_csr_softk_fwd(const float *p, const int64_t *rowptr, const int32_t *col, int64_t N, const float *k, int noise_mode, const float *G,
                      int64_t ldG, uint32_t s0, uint32_t s1, int mode, float *w, float *pp, int32_t *pos, void *stream);
/* g = d loss / d w -> dp [E] (through the perturbation when perturb != 0), dk [N] */
int dgg_csr_softk_bwd(const float *p, const float *pp, const int64_t *rowptr, int64_t N, const float *k, const int32_t *pos, int perturb,
                      int mode, const float *g, float *dp, float *dk, void *stream);
/* `DGG_Ablations.forward` (dgm.py:1927-1962): edge_rank = sigmoid(sigmoid(score) + noise), noise ~ U(-1,1) per stored edge
 * (dgm.py:1930-1933; the caller draws the noise); out [E] */
int dgg_csr_noisy_sigmoid_fwd(const float *p, const float *noise, int64_t E, float *out, void *stream);
/* dp_e = g_e out_e (1 - out_e) */
int dgg_csr_noisy_sigmoid_bwd(const float *out, const float *g, int64_t E, float *dp, void *stream);
/* fixed k (dgm.py:1940-1942, `srt_edge_rank[:, k:] = 0`): out_e = p_e if fewer than kcut entries of the row sort before e
 * under (rank desc, column asc), else 0; pos as in dgg_csr_rank_ramp_fwd */
int dgg_csr_rank_cut_fwd(const float *p, const int64_t *rowptr, const int32_t *col, int64_t N, int kcut, float *out, int32_t *pos,
                         void *stream);
int dgg_csr_rank_cut_bwd(const int32_t *pos, const float *g, int64_t E, int kcut, float *dp, void *stream);
/* raw edge probabilities as the adjacency -- debug_step 0/1 (dgm.py:1202-1209, 1240-1246) and k-select mode `edge_p-cdf`
 * (dgm.py:1368-1401 scatters the unsorted probabilities back) of DGG_LearnableK_debug.  u-v-dist scorer on the stored
 * entries (dgm.py:1613-1627): p_e = exp(t ||xp_u - xp_v||); backward: dxp [N,h] accumulated (caller zeroes) */
int dgg_csr_uvdist_fwd(const float *xp, const int64_t *rowptr, const int32_t *col, int64_t N, int h, float t, float *p, void *stream);
int dgg_csr_uvdist_bwd(const float *xp, const int64_t *rowptr, const int32_t *col, int64_t N, int h, float t, const float *p,
                       const float *dp, float *dxp, void *stream);
/* normalize_adj of the *_DGG_00 wrappers (model.py:1340-1352): rs = row sums, ahat_e = rs_i^-1/2 w_e rs_j^-1/2 */
int dgg_csr_row_sum(const float *vals, const int64_t *rowptr, int64_t N, float *rs, void *stream);
int dgg_csr_normalize_fwd(const int64_t *rowptr, const int32_t *col, const float *w, const float *rs, int64_t N, float *ahat,
                          void *stream);
/* autograd of the two: dA (wrt ahat) -> dw; da_ws [N] zeroed by the caller */
int dgg_csr_norm_bwd(const int64_t *rowptr, const int32_t *col, const float *w, const float *rs, const float *dA, int64_t N,
                     float *da_ws, float *dw, void *stream);
/* torch.mm(adj, x) (model.py:594) on the CSR pattern and its autograd (dA [E]; dX nullable, accumulated into) */
int dgg_csr_spmm_fwd(const int64_t *rowptr, const int32_t *col, const float *a, const float *X, int64_t N, int F, float *Y,
                     void *stream);
int dgg_csr_spmm_bwd(const int64_t *rowptr, const int32_t *col, const float *a, const float *X, const float *dY, int64_t N, int F,
                     float *dA, float *dX, void *stream);
/* GATConv_DGG (model.py:534-577): softmax(dim=1) of the dense logit matrix whose explicit entries are L [E] on the CSR
 * pattern and whose other N - cnt_i entries per row are the logit 0 that -1e20 * 0 produces (model.py:565-567):
 * att [E] on the pattern, bg [N] = the weight every non-listed node of a row receives.  Backward: datt, dbg -> dL. */
int dgg_csr_bg_softmax_fwd(const float *L, const int64_t *rowptr, int64_t N, float *att, float *bg, void *stream);
int dgg_csr_bg_softmax_bwd(const float *att, const float *bg, const int64_t *rowptr, int64_t N, const float *datt, const float *dbg,
                           float *dL, void *stream);
/* Training-mode dropout of that dense attention matrix (model.py:570, F.dropout(attention, p)): every one of the N x N pairs is kept
 * with probability 1 - p -- the non-listed pairs, which all carry bg_i, included.  The mask is counter-based (pair (i, j) kept iff
 * hash24(s0, s1 ^ 0x9E3779B9 (i + 1), j) >= p 2^24; the reference draws from torch's generator: same law, another realisation), so
 * no N x N tensor exists: out_r = sum_s keep(r, s) X_s (transpose 0; the forward's sum of the kept rows of h) or sum_s keep(s, r) X_s
 * (transpose 1; its backward).  X, out [N,F], F <= 64.  dgg_pair_keep: the same mask on listed pairs (erow, col) [E] -> 1.0 / 0.0. */
int dgg_masked_dense_sum(const float *X, int64_t N, int F, float p, uint32_t s0, uint32_t s1, int transpose, float *out, void *stream);
int dgg_pair_keep(const int32_t *erow, const int32_t *col, int64_t E, float p, uint32_t s0, uint32_t s1, float *out, void *stream);
/* selection only, from a dense score matrix [R,N] (test entry: torch.sort(pert_edge_p)[:, :K], dgm.py:1404) */
int dgg_select_scores(const float *scores, int64_t R, int64_t N, int K, int32_t *idx, float *val, void *stream);

/* ---- smooth first-k ramp, normalisation, aggregation ---------------------------------------------------------
 * w_ir = val_ir * (1 - 0.5(1 + tanh(r - k_i)))  (mode 0, dgm.py:1410-1420)  or the ramp alone (mode 1, 1427-1434);
 * mode 3: the forward value of the straight-through hard adjacency `(hard - soft).detach() + soft` with hard = the ramp
 * mask at the selected columns (the well-defined form of dgg_hard, dgm.py:343-346; its gradient is mode 0's, so
 * dgg_softk_bwd is called with mode 0);  rs_i = sum_r w_ir. */
int dgg_softk_fwd(const int32_t *idx, const float *val, const float *k, int64_t N, int K, int mode, float *w, float *rs,
                  void *stream);
/* ahat_ir = rs_i^-1/2 w_ir rs_j^-1/2   (normalize_adj, model.py:1205-1219; rs has GLOBAL length, rows are
 * nodes row0..row0+N-1) */
int dgg_ell_normalize_fwd(const int32_t *idx, const float *w, const float *rs, int64_t N, int K, int64_t row0,
                          float *ahat, void *stream);
/* Y[N,F] = A X   (torch.mm(adj, x) model.py:594, 67; torch.spmm model.py:34); X has GLOBAL rows */
int dgg_ell_spmm_fwd(const int32_t *idx, const float *ahat, const float *X, int64_t N, int K, int F, float *Y,
                     void *stream);
/* Y = act(A X) with the activation fused into the aggregation: act 0 none, 2 ReLU.  This is GCNConv evaluated as
 * relu(A (x W)) instead of relu((A x) W) (model.py:594-598; equal up to fp32 reassociation): when out_features <
 * in_features the F = out_features wide projected rows are gathered instead of the in_features wide inputs.  Rows of
 * 16 / 32 / 64 features are gathered four / two / ... per wave-instruction (F/4 lanes each, 16-byte loads). */
int dgg_ell_spmm_act_fwd(const int32_t *idx, const float *ahat, const float *X, int64_t N, int K, int F, int act, float *Y,
                         void *stream);
/* dp = dy * act'(y) elementwise (cotangent of a fused activation epilogue; act as in dgg_linear_fwd) */
int dgg_act_bwd(const float *y, const float *dy, int64_t n, int act, float *dp, void *stream);
/* dA_ir = <dY_i, X_j> (overwritten); dX_j += ahat_ir dY_i (nullable; accumulated with fp32 atomics).
 * skip_zero != 0: entries whose value is exactly 0 get dA = 0 without the dot product (valid when the adjacency
 * comes from dgg_softk_fwd: a zero weight is a saturated ramp, whose gradient is zero as well) */
int dgg_ell_spmm_bwd(const int32_t *idx, const float *ahat, const float *X, const float *dY, int64_t N, int K, int F,
                     int skip_zero, float *dA, float *dX, void *stream);
/* normalisation backward, phase 1: da (GLOBAL length, zeroed by caller) += d loss / d rs^-1/2 */
int dgg_norm_bwd_da(const int32_t *idx, const float *w, const float *rs, const float *dA, int64_t N, int K, int64_t row0,
                    float *da, void *stream);
/* phase 2 + ramp backward: dval [N,K], dk [N].  normalized = 0: dA is the cotangent of w itself.
 * mode 2: no ramp (dval = dw, dk untouched, val/k may be NULL): plain normalize_adj backward */
int dgg_softk_bwd(const int32_t *idx, const float *val, const float *k, const float *rs, const float *dA, const float *da,
                  int64_t N, int K, int64_t row0, int mode, int normalized, float *dval, float *dk, void *stream);
/* the same for a NORMALISED adjacency whose `da_cols` (GLOBAL length) holds the neighbour-side sums only -- what the column
 * kernels of the partitioned backward leave (dgg_ell_conv_bwd_partp) -- the row side rs_i^1/2 sum_r dA_ir ahat_ir is formed inside
 * (ahat_rows [N,K]: the normalised values).  Reference: autograd of normalize_adj + select_top_k, model.py:1205-1219, dgm.py:1402-1421 */
int dgg_softk_bwd_rows(const int32_t *idx, const float *val, const float *k, const float *rs, const float *dA, const float *da_cols,
                       const float *ahat_rows, int64_t N, int K, int64_t row0, int mode, float *dval, float *dk, void *stream);
/* score backward to the projected features: dxp [Nglobal,h] += ... (zeroed by caller; fp32 atomics) */
int dgg_edge_bwd(const float *xp, int64_t N, int h, const int32_t *idx, const float *val, const float *dval, int K,
                 int64_t row0, float t, int perturb, float *dxp, void *stream);
/* ROW pass of the same for wide latents (h a multiple of 64, <= 2048; the PPI configuration runs the generator at latent 2048,
 * train_ppi.py:44) without float atomics: dxp_rows [N,h] (OVERWRITTEN) = sum_r dd_ir (xp_i - xp_j), dd [N,K] (overwritten) = the
 * per-entry coefficient (d loss / d dist_ir) / dist_ir.  The neighbour side follows from dd with the transposed aggregation:
 * dxp_j -= (DD^T xp)_j - (sum_i dd_ij) xp_j  (dgg_ell_spmm_t_part(a = dd, dY = xp)).  Autograd of dgm.py:1618-1623. */
int dgg_edge_bwd_wide_rows(const float *xp, int64_t N, int h, const int32_t *idx, const float *val, const float *dval, int K,
                           int64_t row0, float t, int perturb, float *dxp_rows, float *dd, void *stream);
/* the same one 256-feature slice of the gathered rows at a time (h a multiple of 256; a slice of a PPI graph's xp stays in an XCD's
 * L2, whole 8 KB rows do not); ws: dgg_edge_bwd_wide_rows_ws_floats(N, K, h) floats */
size_t dgg_edge_bwd_wide_rows_ws_floats(int64_t N, int K, int h);
int dgg_edge_bwd_wide_rows_sliced(const float *xp, int64_t N, int h, const int32_t *idx, const float *val, const float *dval, int K,
                                  int64_t row0, float t, int perturb, float *ws, float *dxp_rows, float *dd, void *stream);

/* ---- column-side backward terms without global float atomics (dgg_scatter.hip) -----------------------------------
 * The ACTIVE entries (idx >= 0, w != 0) of an ELL block are partitioned once per forward by destination bucket; the two
 * backward terms that land on the neighbour j = idx[i][r] (score backward, normalisation backward) are then accumulated
 * per bucket in LDS.  Same results as dgg_edge_bwd / dgg_norm_bwd_da up to summation order. */
size_t dgg_part_ws_bytes(int64_t rows, int K, int64_t ncols);   /* 0: partitioned path not applicable */
int dgg_part_build(const int32_t *idx, const float *w, int64_t rows, int K, int64_t ncols, void *ws, void *stream);
/* PAYLOAD partition: 16-byte records (row*64 + r, j, w_ir rs_i^-1/2, score_ir), no slot map.  With the two per-entry scalars in
 * the record every column-walking kernel of the backward works from its own coalesced record stream; the only value that
 * still crosses between row order and record order is dA.  val [rows,K]: the scores; rs_rows [rows]: row sums of the block's
 * own rows.  dgg_partp_ws_bytes: 0 = not applicable (more than 2^20 destination nodes or 2^25 rows). */
size_t dgg_partp_ws_bytes(int64_t rows, int K, int64_t ncols);
int dgg_partp_build(const int32_t *idx, const float *w, const float *val, const float *rs_rows, int64_t rows, int K, int64_t ncols,
                    void *ws, void *stream);
/* dgg_partp_build with dgg_ell_normalize_fwd fused: rs_all [ncols] = row sums of every node -> ahat [rows,K] (same bits) */
int dgg_partp_build_norm(const int32_t *idx, const float *w, const float *val, const float *rs_rows, int64_t rows, int K, int64_t ncols,
                         const float *rs_all, float *ahat, void *ws, void *stream);
/* dgg_partp_build_norm in two parts (phase 1: count + scan + fill -- ahat complete, records in bucket order; phase 2: the per-bucket
 * sort -- records in node order + the CSC pointer, read only by the backward; phase 0: both), so that the sort can run on a second
 * stream beside the forward aggregation */
int dgg_partp_build_phase(const int32_t *idx, const float *w, const float *val, const float *rs_rows, int64_t rows, int K, int64_t ncols,
                          const float *rs_all, float *ahat, void *ws, int phase, void *stream);
/* Where the pieces of a built payload partition live inside its workspace (byte offsets; for tests and tools that want to look at the
 * records): out[0] = bucket starts (int32 [nb+1]), out[1] = CSC node pointer (int32 [ncols+1]: the records of destination node j are
 * recs[nodeptr[j] .. nodeptr[j+1])), out[2] = records in node order (16 bytes each: row*64 + r, j, bits of w_ir rs_i^-1/2, bits of the
 * score), out[3] = number of buckets, out[4] = bucket width in nodes, out[5] = rows per counting workgroup. */
int dgg_partp_describe(int64_t rows, int K, int64_t ncols, int64_t *out6);
/* 1 when the partition of a block of `rows` rows carries the slot -> record map (blocks whose record-ordered dA fits an XCD's L2): the
 * backward may then skip the row-major dA (dA = NULL in dgg_ell_conv_bwd_partp[_ext] and dgg_softk_edge_bwd_partp[_phase]) */
int dgg_partp_has_map(int64_t rows);
/* dgg_ell_conv_bwd_part on a payload partition (ahat_ir = record payload * rs_j^-1/2, bit-identical to dgg_ell_normalize_fwd);
 * also writes dA_rec [rows*K] = dA in record order.  dA, dH, da: caller zeroes.  dA may be NULL: the row-major copy (one scattered
 * 4-byte store per entry) is then not written, and dgg_softk_edge_bwd_partp -- called with dA = NULL as well -- reads dA_rec through the
 * slot -> record map the partition's sort leaves in its workspace. */
int dgg_ell_conv_bwd_partp(const float *G, const float *H, int64_t rows, int K, int F, const void *partp_ws, int64_t ncols,
                           const float *rs, float *dA, float *dA_rec, float *dH, float *da, void *stream);
/* the same when the normalised adjacency has OTHER consumers besides this aggregation (GCN_DGG hands it to its second layer as well,
 * reference model.py:1266-1290): dA_ext [rows,K] (nullable) = their cotangent, slot for slot; it is added per record, so dA, dA_rec
 * and da hold the totals and the score backward runs once.  Entries outside the partition (weight 0) are not read. */
int dgg_ell_conv_bwd_partp_ext(const float *G, const float *H, int64_t rows, int K, int F, const void *partp_ws, int64_t ncols,
                               const float *rs, const float *dA_ext, float *dA, float *dA_rec, float *dH, float *da, void *stream);
/* dgg_softk_edge_bwd_part on a payload partition: the column kernel recomputes d loss / d score (ramp + normalisation chain,
 * dgm.py:1410-1420, model.py:1215-1218) from dA_rec and the per-row scalars the row kernel leaves in rowinfo_ws (4*rows floats).
 * out_act = 1 (mode 0 only): dxp is returned multiplied by LeakyReLU'(xp) -- the gradient of the pre-activation of node_encode_for_edges
 * (dgm.py:1097-1100) -- so that the weight-gradient product needs no pass over xp for the mask. */
int dgg_softk_edge_bwd_partp(const float *xp, int64_t rows, int h, const int32_t *idx, const float *val, const float *k, const float *rs,
                             const float *dA, const float *dA_rec, const float *da, const float *ahat_rows, int K, int64_t row0, float t,
                             int perturb, int mode, int normalized, const void *partp_ws, int64_t ncols, float *rowinfo_ws, float *dk,
                             float *dxp, int out_act, void *stream);
/* dgg_softk_edge_bwd_partp in two parts (phase 1: the row kernel -- dk and the rows' own side of dxp complete; phase 2: the
 * per-destination kernel; phase 0: both), so that the k-net backward, which needs only dk, can run on a second stream beside phase 2 */
int dgg_softk_edge_bwd_partp_phase(const float *xp, int64_t rows, int h, const int32_t *idx, const float *val, const float *k, const float *rs,
                                   const float *dA, const float *dA_rec, const float *da, const float *ahat_rows, int K, int64_t row0, float t,
                                   int perturb, int mode, int normalized, const void *partp_ws, int64_t ncols, float *rowinfo_ws, float *dk,
                                   float *dxp, int out_act, int phase, void *stream);
/* ---- the same steps on CHUNKED rows (rows wider than 64 ranks: dgg_chunk_layout / dgg_allpairs_topk_ranked_wide) -----------------------
 * idx / w / val / ahat / dA / dA_ext are [chunks,64]; `cnode` int32 [chunks] = node of every chunk, `cptr` int32 [rows+1] = first chunk of
 * every node; per-node arrays (rs, k, dk, G, Y) are [rows] / [rows,F].  The reference has no width limit (select_top_k ramps over the dense
 * row, dgm.py:1402-1421; normalize_adj and torch.mm(adj, x) act on dense [N,N] tensors, model.py:1205-1219, 594).
 * dgg_partp_build_chunked = dgg_partp_build_phase; the SORTED records carry (chunk*64 + entry, SOURCE NODE of the chunk, w rs_i^-1/2,
 * score): the destination of a record is implied by the CSC pointer, and the per-destination kernels read G_i / xp_i at the source node. */
int dgg_partp_build_chunked(const int32_t *idx, const float *w, const float *val, const float *rs_nodes, int64_t chunks, const int32_t *cnode,
                            int64_t ncols, const float *rs_all, float *ahat, void *ws, int phase, void *stream);
/* Y [rows,F] = act(A X) with row i = the chunks [cptr[i], cptr[i+1]), walked in rank order (dgg_ell_spmm_act_fwd) */
int dgg_ell_spmm_act_fwd_chunked(const int32_t *idx, const float *ahat, const float *X, int64_t rows, const int32_t *cptr, int F, int act, float *Y,
                                 void *stream);
/* dgg_ell_conv_bwd_partp_ext on a partition built by dgg_partp_build_chunked; dA is required (no slot -> record map) */
int dgg_ell_conv_bwd_partp_chunked(const float *G, const float *H, int64_t chunks, int F, const void *partp_ws, int64_t ncols, const float *rs,
                                   const float *dA_ext, float *dA, float *dA_rec, float *dH, float *da, void *stream);
/* dgg_softk_edge_bwd_partp_phase on chunked rows: one wavefront per NODE walks its chunks (the normalised form first sums dA ahat over
 * the whole row); rowinfo_ws: 4 * chunks floats, per chunk (rs_i^-1/2, d loss / d rs_i, k_i - 64 m, 0) with m the chunk's position in its
 * row, so that the per-destination kernel forms rank - k from an entry's index inside its chunk; k, dk [rows] */
int dgg_softk_edge_bwd_partp_chunked(const float *xp, int64_t rows, const int32_t *cptr, int64_t chunks, int h, const int32_t *idx, const float *val,
                                     const float *k, const float *rs, const float *dA, const float *dA_rec, const float *da, const float *ahat_rows,
                                     int64_t row0, float t, int perturb, int mode, int normalized, const void *partp_ws, int64_t ncols,
                                     float *rowinfo_ws, float *dk, float *dxp, int out_act, int phase, void *stream);
/* dA [rows,64] by rows (chunked rows: [chunks,64]) -> dA_rec [rows*64] in the record order of a built payload partition (K = 64): for a
 * caller that holds d loss / d w row-major -- the generator used as a separate module, whose output feeds other layers -- and runs the
 * score backward (dgg_softk_edge_bwd_partp[_chunked], normalized = 0) on it */
int dgg_partp_gather_rec(const float *dA, int64_t rows, int64_t ncols, const void *partp_ws, float *dA_rec, void *stream);
/* GCNII layer epilogue (GraphConvolution.forward, model.py:36-44): out = theta * sw + (1 - theta) * r (+ inp), sw = support W,
 * r = (1 - alpha) * hi + alpha * h0 (h0 NULL: r = hi; inp NULL: no residual).  Backward: dsw = theta g, dhi, dh0 (NULL with h0);
 * the residual input's gradient is g itself. */
int dgg_gcnii_epilogue_fwd(const float *sw, const float *hi, const float *h0, const float *inp, int64_t n, float theta, float alpha,
                           float *out, void *stream);
int dgg_gcnii_epilogue_bwd(const float *g, int64_t n, float theta, float alpha, float *dsw, float *dhi, float *dh0, void *stream);
/* dX [ncols,F] += A^T dY through the partition (autograd of torch.mm(adj, x) w.r.t. x, model.py:594, when the conv input is
 * a learned activation): runs of equal destination are reduced in registers, one flush per run.  a [rows,K] on the pattern
 * the partition was built from; F a multiple of 64 (else DGG_ERR_UNSUPPORTED: use dgg_ell_spmm_bwd's atomic dX). */
int dgg_ell_spmm_t_part(const float *a, const float *dY, int64_t rows, int K, int F, const void *part_ws, int64_t ncols, float *dX,
                        void *stream);
/* coef_ws: rows*K (+ ncols for dgg_edge_bwd_part) floats; dxp [ncols,h] / da [ncols] zeroed by the caller;
 * latent_dim in {16,32,64,128} */
int dgg_edge_bwd_part(const float *xp, int64_t rows, int h, const int32_t *idx, const float *val, const float *dval, int K,
                      int64_t row0, float t, int perturb, const void *part_ws, int64_t ncols, float *coef_ws, float *dxp,
                      void *stream);
/* dgg_softk_bwd (modes 0 / 1) and dgg_edge_bwd_part in one call: lane r of a row's wavefront owns entry r in both kernels, so
 * d loss / d score is formed in registers inside the row kernel of the score backward.  dval (nullable) [rows,K] receives it
 * as well; dk [rows] is written; the other arguments as in the two separate calls */
/* ahat_rows (nullable, [rows,K]; normalized only): `da` then holds the NEIGHBOUR-side sums alone (as written by
 * dgg_ell_conv_bwd_part) and the row side, sqrt(rs_i) sum_r dA_ir ahat_ir, is added inside the kernel */
int dgg_softk_edge_bwd_part(const float *xp, int64_t rows, int h, const int32_t *idx, const float *val, const float *k, const float *rs,
                            const float *dA, const float *da, const float *ahat_rows, int K, int64_t row0, float t, int perturb,
                            int mode, int normalized, const void *part_ws, int64_t ncols, float *coef_ws, float *dval, float *dk,
                            float *dxp, void *stream);
/* Backward of Z = A H (H [ncols,F] = projected features, G [rows,F] = d loss / d (A H)) through the partition, ONE gathered
 * row of G per active entry for all three column-walking terms: dA_ir = <G_i, H_j> (autograd of torch.mm(adj, x) wrt adj,
 * model.py:594), dH_j += ahat_ir G_i (wrt x) and da_j += dA_ir w_ir a_i (neighbour side of the normalize_adj backward,
 * model.py:1215-1218; nullable, needs rs).  dA [rows,K] is written for the ACTIVE entries only (caller zeroes), dH [ncols,F]
 * and da [ncols] are accumulated into (caller zeroes).  F in {16,32,64,128}, 16-byte aligned rows. */
int dgg_ell_conv_bwd_part(const float *G, const float *H, const float *ahat, int64_t rows, int K, int F, const void *part_ws,
                          int64_t ncols, const float *rs, float *dA, float *dH, float *da, void *stream);
int dgg_norm_bwd_da_part(const int32_t *idx, const float *w, const float *rs, const float *dA, int64_t rows, int K, int64_t row0,
                         const void *part_ws, int64_t ncols, float *coef_ws, float *da, void *stream);
/* SDDMM of dgg_ell_spmm_bwd (dA only) fused with the row side of dgg_norm_bwd_da_part: one pass over the row instead of
 * two (ONE launch).  dA [rows,K] overwritten, da [ncols] zeroed by the caller, coef_ws rows*K floats.  Returns
 * DGG_ERR_UNSUPPORTED unless F is 128 or 256 with 16-byte aligned rows (then use the two separate calls).  Follow with
 * dgg_norm_da_cols_part (the neighbour-side sums of the coefficients, through the partition). */
int dgg_ell_sddmm_norm_part(const int32_t *idx, const float *ahat, const float *w, const float *rs, const float *X,
                            const float *dY, int64_t rows, int K, int F, int64_t row0, int skip_zero, const void *part_ws,
                            int64_t ncols, float *coef_ws, float *dA, float *da, void *stream);
int dgg_norm_da_cols_part(const void *part_ws, int64_t rows, int K, int64_t ncols, const float *coef_ws, float *da, void *stream);

/* ---- bf16 matrix-core path of the GCNII layer product (BASELINE configs[4]: "bf16 fwd+bwd, MFMA feature-projection GEMM") ----
 * GraphConvolution.forward (model.py:32-44): out = theta * (support @ weight) + (1 - theta) * r (+ input).  The product (and its
 * autograd) runs on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; operands are bf16 copies made by dgg_pack_bf16.
 * dgg_pack_bf16: fp32 src [R,C] -> bf16 dst; transpose 0: dst [R][ld] (ld >= C), 1: dst [C][ld] (ld >= R); padding is zeroed.
 * dgg_gemm_nt_bf16: C[M,N] (fp32) = scale * A[M,K] B[N,K]^T, both operands bf16 with K contiguous, K a multiple of 64
 *   (d support = theta g W^T: A = g, B = weight as stored;  d weight = theta support^T g: A = support^T, B = g^T, K = padded n).
 * dgg_gcnii_gemm_bf16: the forward with the epilogue fused; S bf16 [n,K], Wt bf16 [F,K] = weight^T; hi/h0/inp fp32 [n,F]
 *   (r = h0 ? (1 - alpha) hi + alpha h0 : hi;  h0, inp nullable). */
int dgg_pack_bf16(const float *src, int64_t R, int64_t C, int transpose, void *dst, int64_t ld, void *stream);
/* both layouts from one read of src (a weight is the B operand of the forward as W^T and of d support as stored):
 * dst [R][ld] (ld >= C) and dstT [C][ldT] (ldT >= R) */
int dgg_pack_bf16_both(const float *src, int64_t R, int64_t C, void *dst, int64_t ld, void *dstT, int64_t ldT, void *stream);
int dgg_gemm_nt_bf16(const void *A, const void *B, int64_t M, int64_t N, int64_t K, float scale, float *C, void *stream);
int dgg_gcnii_gemm_bf16(const void *S, const void *Wt, int64_t n, int64_t F, int64_t K, const float *hi, const float *h0, const float *inp,
                        float theta, float alpha, float *out, void *stream);
/* The VARIANT layer (support = cat[hi, h0], model.py:37-40) without forming the concatenation: the contraction range [0,F1) of the
 * A operand is S1 = bf16(hi) [n,F1], the range [F1,K) is S2 = bf16(h0) [n,K-F1] (F1 a multiple of 64; h0 is the same tensor in every
 * layer of a GCNII stack, model.py:724, so the caller packs it once per forward).  Otherwise as dgg_gcnii_gemm_bf16. */
int dgg_gcnii_gemm_bf16_split(const void *S1, const void *S2, const void *Wt, int64_t n, int64_t F, int64_t K, int64_t F1, const float *hi,
                              const float *h0, const float *inp, float theta, float alpha, float *out, void *stream);
/* Backward of the variant layer w.r.t. hi and h0 in ONE product (autograd of model.py:37-44):
 *   [dhi | dh0] = theta * Gp W^T + [(1-theta)(1-alpha) | (1-theta) alpha] * g
 * Gp bf16 [n,F] = bf16(g), Wp bf16 [2F,F] = the weight as stored, g fp32 [n,F]; dhi, dh0 fp32 [n,F]: no [n,2F] intermediate, no
 * slicing adds, no separate epilogue pass. */
/* C[M,N] = scale * [A ; A2] B^T: rows [0,M1) of the A operand in A [M1,K], rows [M1,M) in A2 [M-M1,K] (M1 a multiple of 128): the weight
 * gradient of the variant layer, theta * cat[hi, h0]^T g (autograd of model.py:41), from the two transposed halves in one product. */
int dgg_gemm_nt_bf16_rows2(const void *A, const void *A2, int64_t M1, const void *B, int64_t M, int64_t N, int64_t K, float scale, float *C,
                           void *stream);
int dgg_gcnii_dsupport_bf16(const void *Gp, const void *Wp, int64_t n, int64_t F, const float *g, float theta, float alpha, float *dhi,
                            float *dh0, void *stream);

/* ---- the fused GCNII stack on bf16 operands (reference model.py:32-44 layer, 716-731 / 942-957 stack loop:
 * dropout -> layer -> ReLU, h0 = the first activation, one adjacency for every layer).  Per layer, forward:
 *   dgg_ell_spmm_fwd_bf16          hi = A xd and bf16(hi) in the same pass (no pack of the product operand)
 *   dgg_gcnii_gemm_bf16_split_act  xd' = dropout(relu(theta [hi|h0] W + (1-theta)((1-alpha) hi + alpha h0) + xd)) -- the layer's
 *                                  ReLU (model.py:724) and the NEXT layer's dropout (model.py:722) in the product's epilogue; the
 *                                  dropout mask is counter-based (element e kept iff hash24(s0, s1, e) >= drop_p 2^24): the backward
 *                                  reads it off the stored output (xd' != 0), nothing else is saved
 *   dgg_dropout_hash               the same mask on a tensor no product produces (the stack's input h0; `accumulate`: out += ...,
 *                                  used with the same seeds for the gradient flowing back through that dropout)
 * backward:
 *   dgg_gcnii_gout_pack            g = g_in * (xd' != 0 ? 1/(1-p) : 0) as fp32 (twice on request: the second copy is the buffer the
 *                                  transposed aggregation accumulates into -- the residual add) and as the two bf16 operand packs
 * ([d hi | d h0] is dgg_gcnii_dsupport_bf16 -- accumulating d h0 inside its epilogue measured 40 us slower per product than a separate
 * add --, the weight gradient dgg_gemm_nt_bf16_rows2; d A and A^T d hi are dgg_ell_spmm_bwd / dgg_ell_spmm_t_part) */
/* (Yb rows have stride ldyb >= F bf16 elements: the copy can be written straight into the left half of a [n, 2F] product operand) */
int dgg_ell_spmm_fwd_bf16(const int32_t *idx, const float *ahat, const float *X, int64_t N, int K, int F, float *Y, void *Yb, int64_t ldyb,
                          void *stream);
/* the layer's epilogue as a separate pass after the plain product dgg_gemm_nt_bf16 (measured faster than inside the product's epilogue) */
int dgg_gcnii_stack_epilogue(const float *sw, const float *hi, const float *h0, const float *inp, int64_t n, float theta, float alpha,
                             float drop_p, uint32_t s0, uint32_t s1, float *out, void *outb, void *stream);
/* `outb` (nullable): bf16 copy of the output, for the kernels that gather it next */
int dgg_gcnii_gemm_bf16_split_act(const void *S1, const void *S2, const void *Wt, int64_t n, int64_t F, int64_t K, int64_t F1, const float *hi,
                                  const float *h0, const float *inp, float theta, float alpha, int relu, float drop_p, uint32_t s0,
                                  uint32_t s1, float *out, void *outb, void *stream);
int dgg_dropout_hash(const float *x, int64_t n, float p, uint32_t s0, uint32_t s1, int accumulate, float *out, void *outb, void *stream);
/* The stack's three gather kernels on bf16 COPIES of what they gather (the activation in the aggregation and the SDDMM, d hi in the
 * transposed aggregation): at F = 2048 they are bound by L2 bandwidth, the copies halve their bytes; accumulation stays fp32.
 *   dgg_ell_spmm_fwd_b16     Y = A Xb (+ bf16(Y));  dgg_ell_sddmm_b16   dA = <dYb_i, Xb_j>;  dgg_ell_spmm_t_part_b16   dX += A^T dYb
 *   dgg_gcnii_dsupport_bf16_b  dgg_gcnii_dsupport_bf16 that also leaves bf16(d hi) in dhib (dhi may then be NULL); accumulate_dh0:
 *                              dh0 += its half (the sum over the stack's layers without an add pass per layer) */
int dgg_ell_spmm_fwd_b16(const int32_t *idx, const float *ahat, const void *Xb, int64_t N, int K, int F, float *Y, void *Yb, int64_t ldyb,
                         void *stream);
int dgg_ell_sddmm_b16(const int32_t *idx, const float *ahat, const void *Xb, const void *dYb, int64_t N, int K, int F, int skip_zero, float *dA,
                      void *stream);
int dgg_ell_spmm_t_part_b16(const float *a, const void *dYb, int64_t rows, int K, int F, const void *part_ws, int64_t ncols, float *dX,
                            void *stream);
/* dgg_ell_sddmm_b16 one 512-feature slice of the gathered rows at a time (F a multiple of 512; a slice of a PPI graph's activation
 * stays in an XCD's L2, whole rows do not); ws: dgg_ell_sddmm_b16_ws_floats(N, K, F) floats; accumulate: dA += instead of =.
 * dA == NULL: the slices' sums stay in ws (overwritten, or added to what it holds: accumulate) and dgg_ell_sddmm_slices_sum adds the
 * slices into dA after the last call of a series */
int dgg_ell_sddmm_slices_sum(const float *ws, int64_t N, int K, int F, float *dA, int accumulate, void *stream);
size_t dgg_ell_sddmm_b16_ws_floats(int64_t N, int K, int F);
int dgg_ell_sddmm_b16_sliced(const int32_t *idx, const float *ahat, const void *Xb, const void *dYb, int64_t N, int K, int F, int skip_zero,
                             float *ws, float *dA, int accumulate, void *stream);
int dgg_gcnii_dsupport_bf16_b(const void *Gp, const void *Wp, int64_t n, int64_t F, const float *g, float theta, float alpha, float *dhi,
                              float *dh0, void *dhib, int accumulate_dh0, void *stream);
int dgg_gcnii_gout_pack(const float *gin, const float *xd, float scale, int64_t n, int64_t F, float *g, void *Gp, void *GT, int64_t ldT,
                        float *g2, void *stream);

/* ---- dense all-pairs alternates: DGG_LearnableK_SDD (dgm.py:259-351, dist_fn="metric") and DGG_StraightThrough
 * (dgm.py:140-182 + 63-100), noise off.  Rows are a softmax over ALL N columns, outputs are dense [B,N,N]: O(N^2) by
 * definition, written for batches of small graphs (N <= 8192).
 *   prob = exp(-t dist(xq_i, xq_j)); y = softmax_j(log(prob) / temp); pos = position under (y desc, column asc)
 *   ramp 0 (SDD, dgm.py:315-346): f = sigmoid((hs_start - interval pos) + interval (k_i - 1)); out = y f, hard: (f - y f) + y f
 *   ramp 1 (ST, dgm.py:83-98):    out = y, hard: (1[pos < kfix] - y) + y
 * xq [B,N,h]; t device scalar (nn.Parameter); k [B*N] (ramp 0); out, y [B,N,N] fp32, pos [B,N,N] int32 */
int dgg_dense_rows_fwd(const float *xq, int B, int64_t N, int h, const float *t, float temp, int ramp, const float *k, int kfix,
                       float hs_start, float interval, int hard, float *out, float *y, int32_t *pos, void *stream);
/* g = d loss / d out [B,N,N] -> Cm [B,N,N] (coefficient of (xq_i - xq_j) from row i; feeds dgg_dense_pairs_dx),
 * dk [B*N] (ramp 0, else NULL), dt_rows [B*N] (their sum = d loss / d t) */
int dgg_dense_rows_bwd(const float *xq, int B, int64_t N, int h, const float *t, float temp, int ramp, const float *k, float hs_start,
                       float interval, const float *y, const int32_t *pos, const float *g, float *Cm, float *dk, float *dt_rows,
                       void *stream);
/* d loss / d xq_i = sum_j (C_ij + C_ji)(xq_i - xq_j)  (autograd of torch.cdist, dgm.py:275 / 157) */
int dgg_dense_pairs_dx(const float *xq, int B, int64_t N, int h, const float *Cm, float *dxq, void *stream);
/* nn.Softmax(dim=-1) over the latent features (SDD input_project, dgm.py:217-221): out [rows,h]; dz = out (g - <out, g>) */
int dgg_feat_softmax_fwd(const float *z, int64_t rows, int h, float *out, void *stream);
int dgg_feat_softmax_bwd(const float *out, const float *g, int64_t rows, int h, float *dz, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DGG_HIP_H */
