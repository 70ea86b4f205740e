// dgg_capi.hip -- error plumbing, library info and the all-pairs dispatcher of libdgg_hip.so
#include "dgg_api_internal.h"
#include <string.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

int dgg_set_error(int code, const char *msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}
int dgg_check_hip(hipError_t e, const char *what) {
    if (e == hipSuccess) return DGG_OK;
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return DGG_ERR_HIP;
}
int dgg_check_launch(const char *what) { return dgg_check_hip(hipGetLastError(), what); }

extern "C" {

const char *dgg_last_error(void) { return g_err; }
int dgg_abi_version(void) { return 1; }

// algo: 0 auto, 1 exhaustive (every pair scored with the canonical arithmetic), 2 pruned fast path
int dgg_allpairs_topk(const float *xp, int64_t N, int h, int64_t row0, int64_t row1, float t, int noise_mode,
                      const float *G, int64_t ldG, uint32_t s0, uint32_t s1, int K, int32_t *idx, float *val, int algo,
                      void *workspace, size_t ws_bytes, void *stream) {
    if (K < 1 || K > 64) return dgg_set_error(DGG_ERR_UNSUPPORTED, "ELL width K must be in [1,64]");
    if (row0 < 0 || row1 > N || row0 > row1) return dgg_set_error(DGG_ERR_ARG, "allpairs_topk: bad row range");
    if (noise_mode < 0 || noise_mode > 3) return dgg_set_error(DGG_ERR_ARG, "allpairs_topk: bad noise_mode");
    if (noise_mode == 1 && !G) return dgg_set_error(DGG_ERR_ARG, "explicit noise requested but G is NULL");
    return dgg_allpairs_topk_exhaustive_impl(xp, N, h, row0, row1, t, noise_mode, G, ldG, s0, s1, K, idx, val,
                                             (hipStream_t)stream);
}

}  // extern "C"
