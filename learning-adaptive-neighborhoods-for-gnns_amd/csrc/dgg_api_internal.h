// dgg_api_internal.h -- error plumbing shared by the translation units of libdgg_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DGG_OK 0
#define DGG_ERR_ARG 1          // invalid argument (reference style: bare assert, dgm.py:1187-1188)
#define DGG_ERR_UNSUPPORTED 2  // shape/mode outside what the kernels implement (reference: Exception("mode not found"), dgm.py:1727)
#define DGG_ERR_HIP 3          // a HIP runtime call or kernel launch failed

#define DGG_CHUNK_MAXM 32       // chunked rows, ranked search with register lists: at most 32 chunks of 64 ranks per row (2048 ranks)
#define DGG_CHUNK_MAXM_ANY (1 << 20)   // chunked rows of ANY width (dgg_topk_anywide.hip: threshold buffers in memory): up to 2^26 ranks

int dgg_set_error(int code, const char *msg);
int dgg_check_launch(const char *what);
int dgg_check_hip(hipError_t e, const char *what);

int dgg_allpairs_topk_exhaustive_impl(const float *xp, int64_t N, int h, int64_t row0, int64_t row1, float t,
                                      int noise_mode, const float *G, int64_t ldG, uint32_t s0, uint32_t s1, int K,
                                      int32_t *idx, float *val, hipStream_t st);
// unperturbed scores, two-phase guess-sweep-verify (dgg_topk_sweep.hip)
int dgg_allpairs_topk_sweep_impl(const float *xp, int64_t N, int h, int64_t row0, int64_t row1, float t, int K, const float *klim, int32_t *idx, float *val,
                                 void *workspace, size_t ws_bytes, hipStream_t st);
size_t dgg_allpairs_sweep_ws_bytes(int64_t rows, int64_t N, int h);
size_t dgg_allpairs_sweep_ctl_offset(int64_t rows, int64_t N, int h);
bool dgg_allpairs_sweep_supported(int h, int noise_mode, int K);

int dgg_allpairs_topk_gv_impl(const float *xp, int64_t N, int h, int64_t row0, int64_t row1, float t, int noise_mode,
                              uint32_t s0, uint32_t s1, int K, int32_t *idx, float *val, void *workspace, size_t ws_bytes,
                              hipStream_t st);
size_t dgg_allpairs_gv_ws_bytes(int64_t rows, int64_t N);
bool dgg_allpairs_gv_supported(int h, int noise_mode, int K);

// ranked symmetric noise (dgg_topk_rsym.hip)
int dgg_allpairs_topk_rsym_impl(const float *xp, int64_t N, int h, int64_t row0, int64_t row1, float t, uint32_t s0, uint32_t s1, int K,
                                const float *klim, int32_t *idx, float *val, void *workspace, size_t ws_bytes, hipStream_t st);
size_t dgg_allpairs_rsym_ws_bytes(int64_t rows, int64_t N);
bool dgg_allpairs_rsym_supported(int h, int K);

int dgg_allpairs_topk_ranked_impl(const float *xp, int64_t N, int h, int64_t row0, int64_t row1, float t, uint32_t s0,
                                  uint32_t s1, int K, const float *klim, int32_t *idx, float *val, hipStream_t st, int softk_mode = 0,
                                  float *w = nullptr, float *rs = nullptr, const uint32_t *seed_dev = nullptr, const float *lpub = nullptr);
// partition of the forward (dgg_scatter.hip): slot map [rows*K] inside the workspace (NULL if unsupported); column pass
const int *dgg_part_slotmap(const void *part_ws, int64_t rows, int K, int64_t ncols);
int dgg_norm_da_cols_impl(const void *part_ws, int64_t rows, int K, int64_t ncols, const float *coef_ws, float *da,
                          hipStream_t st);
extern "C" size_t dgg_part_ws_bytes(int64_t rows, int K, int64_t ncols);
// payload partition (dgg_scatter.hip): nodeptr [ncols + 1] and the entry -> record map recpos [rows * 64] inside a built workspace
// (0 on success; recpos is there only when dgg_partp_has_map(rows))
int dgg_partp_internal_ptrs(const void *partp_ws, int64_t rows, int K, int64_t ncols, const int **nodeptr, const int **recpos);
int dgg_klimit_truncate_impl(const float *klim, int64_t rows, int K, int32_t *idx, float *val, hipStream_t st);

// unperturbed chunked rows, front end (dgg_topk_sweep.hip): radius per row from a sampled sweep, one full fp16-MFMA sweep; the columns
// inside a row's radius are appended to its dgg_plain_wide_sublists() sub-lists of `cslot` * M_i / sublists slots each (cand), their counts
// to ncand [rows * sublists]; rad [rows]: every pair that is not listed has d^2 > rad[i]
size_t dgg_plain_wide_front_ws_bytes(int64_t rows, int64_t N, int h);
int dgg_plain_wide_sublists(void);
int dgg_plain_wide_front_impl(const float *xp, int64_t N, int h, int64_t row0, int64_t row1, const float *klim, const int32_t *cptr, int cslot,
                              uint32_t *cand, int32_t *ncand, float *rad, void *ws, hipStream_t st);
