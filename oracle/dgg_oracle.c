/*
 * dgg_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, fp32) of the reference's Differentiable Graph Generator hot path:
 *   DGG_LearnableK_debug.forward            /root/reference/dgm.py:1178-1292
 *     edge_prob_net  mode "u-v-dist"        dgm.py:1607-1627
 *     Gumbel perturbation                   dgm.py:1211-1231, gumbel_sample 14-29
 *     k_estimate_net mode "x" / "input_deg" dgm.py:1562-1586 / 1509-1526, LearnableKEncoder 2029-2063
 *     select_top_k   "k_times_edge_prob" / "k_only"   dgm.py:1402-1435
 *     edge_prob_net  edge-MLP modes          dgm.py:1628-1725 (u-v-A_uv, u-v-deg, u-v-deg-dist, edge_conv, A_uv)
 *     k_estimate_net "learn_normalized_degree" / "gcn-x-deg"   dgm.py:1492-1507 / 1528-1560
 *     debug_step 0 / 1, select_top_k "edge_p-cdf": raw edge probabilities returned   dgm.py:1202-1209, 1240-1246, 1368-1401
 *     dgg_hard as the straight-through value (ramp - soft) + soft                     dgm.py:343-346 (the
 *                                             debug class's own return_hard_or_soft is NOT restated, parity unpinned for it)
 *   DGG.forward ("for ICLR", every edge kept) dgm.py:1758-1815 on a CSR-valued adjacency
 *   DGG_Ablations.forward                    dgm.py:1904-1968 (noisy second sigmoid, learned or fixed k)
 *   DGG_LearnableK_SDD.forward / DGG_StraightThrough.forward (dist_fn="metric", noise off)   dgm.py:259-351 / 140-182
 *   normalize_adj                           model.py:1205-1219
 *   GCNConv / GraphConvolution aggregation  model.py:580-599, 32-44
 * in the sparse "top-K per row" formulation that SURVEY.md section 0/8(a) shows to be bit-identical to the
 * dense N x N reference formulation (the tanh ramp is exactly 0.0f beyond rank k+8.5).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (dgg_amd/) never does.  Parity is PINNED: tests/test_oracle_golden.py checks every function
 * here against fixtures produced by importing the reference itself (tests/golden/make_golden.py).
 *
 * Canonical arithmetic.  Every floating-point step is written as an explicit sequence of IEEE-754
 * binary32 operations (add, mul, fmaf, div, sqrt) in a fixed order, with our own exp/log/tanh
 * polynomials, so that the HIP kernels -- which restate the same sequences independently -- agree
 * with this file bit-for-bit on scores and therefore on top-k indices.  Against the reference
 * (torch CPU: vectorised reductions, Sleef transcendentals) values agree to a few ulp; near-ties
 * can therefore swap ranks, which the golden tests canonicalise.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORA_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------ */
/* canonical transcendental functions                                                          */
/* ------------------------------------------------------------------------------------------ */
static inline float f_from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t bits_from_f(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* exp: argument clamped to [-87, 88]; n = rint(x*log2e); two-step Cody-Waite; degree-5 Horner (Cephes
 * expf coefficients) evaluated with fmaf; scaling by an exactly representable power of two. */
ORA_API float ora_exp(float x) {
    if (x > 88.0f) x = 88.0f;
    if (x < -87.0f) x = -87.0f;
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float q = 1.9875691500e-4f;
    q = fmaf(q, r, 1.3981999507e-3f);
    q = fmaf(q, r, 8.3334519073e-3f);
    q = fmaf(q, r, 4.1665795894e-2f);
    q = fmaf(q, r, 1.6666665459e-1f);
    q = fmaf(q, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = fmaf(q, r2, r);
    y = y + 1.0f;
    int32_t e = (int32_t)n;
    return y * f_from_bits((uint32_t)(e + 127) << 23);
}

/* log for positive normal x (Cephes logf scheme, fixed fmaf sequence). */
ORA_API float ora_log(float x) {
    uint32_t ux = bits_from_f(x);
    int32_t e = (int32_t)(ux >> 23) - 126;                 /* x = m * 2^e, m in [0.5, 1) */
    float m = f_from_bits((ux & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = m + m; }
    m = m - 1.0f;
    float z = m * m;
    float q = 7.0376836292e-2f;
    q = fmaf(q, m, -1.1514610310e-1f);
    q = fmaf(q, m, 1.1676998740e-1f);
    q = fmaf(q, m, -1.2420140846e-1f);
    q = fmaf(q, m, 1.4249322787e-1f);
    q = fmaf(q, m, -1.6668057665e-1f);
    q = fmaf(q, m, 2.0000714765e-1f);
    q = fmaf(q, m, -2.4999993993e-1f);
    q = fmaf(q, m, 3.3333331174e-1f);
    float y = (q * m) * z;
    float fe = (float)e;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(z, -0.5f, y);
    float r = m + y;
    r = fmaf(fe, 0.693359375f, r);
    return r;
}

/* tanh: |z| < 0.625 -> odd polynomial (Cephes tanhf); else 1 - 2/(exp(2|z|)+1); saturates to +-1 for |z| > 9.1 */
ORA_API float ora_tanh(float x) {
    float ax = fabsf(x);
    float r;
    if (ax < 0.625f) {
        float z = x * x;
        float q = -5.70498872745e-3f;
        q = fmaf(q, z, 2.06390887954e-2f);
        q = fmaf(q, z, -5.37397155531e-2f);
        q = fmaf(q, z, 1.33314422036e-1f);
        q = fmaf(q, z, -3.33332819422e-1f);
        q = q * z;
        return fmaf(q, x, x);
    }
    if (ax > 9.1f) r = 1.0f;
    else {
        float e = ora_exp(ax + ax);
        r = 1.0f - 2.0f / (e + 1.0f);
    }
    return x < 0.0f ? -r : r;
}

/* ------------------------------------------------------------------------------------------ */
/* counter-based noise: U_ij from a 2-multiply keyed hash, G = -0.3 * log(-log(U))             */
/* (the reference samples torch.distributions.Gumbel(0, 0.3), dgm.py:1149-1151, 1226; its      */
/*  stream cannot exist at N=100k, so the at-scale generator is defined here; parity at small  */
/*  N uses explicit noise tensors instead)                                                     */
/* ------------------------------------------------------------------------------------------ */
static inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
ORA_API void ora_rowkey(uint32_t s0, uint32_t s1, uint32_t i, uint32_t *k1, uint32_t *k2) {
    *k1 = mix32(i ^ s0);
    *k2 = mix32(*k1 ^ s1 ^ 0x9E3779B9U);
}
/* 24-bit uniform integer for ordered pair (i,j); symmetric: keyed on (min,max). */
ORA_API uint32_t ora_pair_u24(uint32_t s0, uint32_t s1, uint32_t i, uint32_t j, int symmetric) {
    uint32_t a = i, b = j;
    if (symmetric && j < i) { a = j; b = i; }
    uint32_t k1, k2;
    ora_rowkey(s0, s1, a, &k1, &k2);
    uint32_t x = b ^ k1;
    x *= 0x7feb352dU; x ^= x >> 15; x += k2; x *= 0x846ca68bU;
    return x >> 8;
}
ORA_API float ora_gumbel_u24(uint32_t u24) {
    if (u24 == 0) u24 = 1;                        /* U in [2^-24, 1-2^-24] */
    float U = (float)u24 * 5.9604644775390625e-8f; /* exact */
    float a = -ora_log(U);
    float b = ora_log(a);
    return -0.3f * b;
}
ORA_API float ora_noise(uint32_t s0, uint32_t s1, uint32_t i, uint32_t j, int symmetric) {
    if (symmetric && i == j) return 0.0f;          /* dgm.py:1218-1222: diagonal of the symmetric G stays 0 */
    return ora_gumbel_u24(ora_pair_u24(s0, s1, i, j, symmetric));
}

/* ------------------------------------------------------------------------------------------ */
/* "ranked" counter-based noise (noise_mode 4): the same iid Gumbel(0,0.3) law, generated per  */
/* row in DECREASING order so that a top-k search can stop early.                              */
/*   - the order statistics U_(1) >= U_(2) >= ... of N iid uniforms satisfy (Renyi)            */
/*         -log U_(s) = sum_{t<=s} E_t / (N - t + 1),   E_t iid Exp(1)                          */
/*     the prefix sums are kept in exact 64-bit fixed point (2^-40), so their value does not   */
/*     depend on the summation order (sequential here, wavefront scan on the GPU);             */
/*   - rank s is assigned to column sigma_i(r'), the s-th element < N of a keyed bijection of  */
/*     [0, 2^b) walked in order r' = 0, 1, 2, ... (b = ceil(log2 N)): a pseudo-random          */
/*     permutation per row, independent of the values, hence the row is iid again.             */
/* ------------------------------------------------------------------------------------------ */
static inline int ranked_bits(int64_t N) { int b = 6; while (((int64_t)1 << b) < N) b++; return b; }
static inline uint32_t ranked_sigma(uint32_t r, uint32_t k1, uint32_t k2, uint32_t k3, int b) {
    const uint32_t mask = (b >= 32) ? 0xffffffffu : ((1u << b) - 1u);
    const int hb = (b + 1) / 2;
    uint32_t x = (r ^ k1) & mask;
    x = (x * 0x9E3779B1u) & mask; x ^= x >> hb;
    x = (x + k2) & mask; x = (x * 0x85EBCA77u) & mask; x ^= x >> hb;
    x = (x * 0xC2B2AE3Du + k3) & mask; x ^= x >> hb;
    return x;
}
static inline uint64_t ranked_term(uint32_t k1, uint32_t k3, uint32_t s, int64_t N) {   /* s = 1-based rank */
    uint32_t v = mix32(mix32(s + k3) ^ k1) >> 8;
    if (v == 0) v = 1;
    float V = (float)v * 5.9604644775390625e-8f;
    float E = -ora_log(V);
    float term = E / (float)(N - (int64_t)s + 1);
    return (uint64_t)(term * 1099511627776.0f);                 /* floor(term * 2^40), exact */
}
static inline float ranked_gumbel(uint64_t S) {                 /* S = fixed-point prefix sum */
    uint32_t q = (uint32_t)(S >> 16);
    if (q == 0) q = 1;
    float L = (float)q * 5.9604644775390625e-8f;                /* -log U_(s), resolution 2^-24 */
    return -0.3f * ora_log(L);
}
/* noise of row i for every column: G[j], j < N.
 * Round 6: the DIAGONAL has its own, independent variate; the ranked sequence covers the n = N - 1 OTHER columns (slot c of the keyed
 * bijection -> column c + (c >= i)).  The joint law is unchanged -- order statistics of n iid uniforms on a random permutation of the
 * off-diagonal columns, one more independent sample on the diagonal: N iid Gumbel(0, 0.3) -- and the walk of the HIP search now visits
 * the row's own column (the one pair at distance 0) FIRST, so every rank it has not reached is another node and the search may bound
 * its score with the row's nearest-neighbour distance (dgg_allpairs_rowmin_bound).
 * Diagonal: E = -log V from the generator's own exponential variate (ranked_term with rank 1 of 1: the term is floor(E 2^40), no
 * division), G = -0.3 log E in the fixed point of ranked_gumbel: -0.3 log(-log V) is Gumbel(0, 0.3). */
#define ORA_RANKED_DIAG_KEY 0xA5A5A5A5u
ORA_API void ora_ranked_row(uint32_t s0, uint32_t s1, uint32_t i, int64_t N, float *G) {
    uint32_t k1, k2;
    ora_rowkey(s0, s1, i, &k1, &k2);
    const uint32_t k3 = mix32(k2 ^ 0x68E31DA4u);
    const int64_t n = N - 1;
    const int b = ranked_bits(n);
    uint64_t S = 0;
    uint32_t s = 0;
    if ((int64_t)i < N) G[i] = ranked_gumbel(ranked_term(k1, k3 ^ ORA_RANKED_DIAG_KEY, 1, 1));
    for (uint64_t r = 0; r < ((uint64_t)1 << b); r++) {
        uint32_t c = ranked_sigma((uint32_t)r, k1, k2, k3, b);
        if ((int64_t)c >= n) continue;
        s++;
        S += ranked_term(k1, k3, s, n);
        G[(int64_t)c + ((int64_t)c >= (int64_t)i ? 1 : 0)] = ranked_gumbel(S);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* dense layers: acc = 0; acc = fmaf(x_c, w_c, acc) for c ascending; + bias; activation        */
/* (nn.Linear + LeakyReLU dgm.py:1097-1100, 1123-1130; GCNConv mm model.py:594-598)            */
/* act: 0 none, 1 leaky_relu(0.01), 2 relu.  w_layout 0: W[out][in] (nn.Linear), 1: W[in][out] */
/* ------------------------------------------------------------------------------------------ */
static inline float act_fwd(float v, int act) {
    if (act == 1) return v > 0.0f ? v : 0.01f * v;
    if (act == 2) return v > 0.0f ? v : 0.0f;
    return v;
}
ORA_API void ora_linear(const float *x, int64_t N, int d, const float *W, const float *b, int out,
                        int w_layout, int act, float *y) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; i++) {
        const float *xi = x + i * d;
        for (int o = 0; o < out; o++) {
            float acc = 0.0f;
            if (w_layout == 0) { const float *w = W + (int64_t)o * d; for (int c = 0; c < d; c++) acc = fmaf(xi[c], w[c], acc); }
            else { for (int c = 0; c < d; c++) acc = fmaf(xi[c], W[(int64_t)c * out + o], acc); }
            if (b) acc = acc + b[o];
            y[i * out + o] = act_fwd(acc, act);
        }
    }
}
/* backward of y = act(x W^T + b): given dy and y (post-activation), returns dx (optional), dW, db (accumulated
 * in double, written as float). */
ORA_API void ora_linear_bwd(const float *x, int64_t N, int d, const float *W, int out, int w_layout, int act,
                            const float *y, const float *dy, float *dx, float *dW, float *db) {
    double *aW = (double *)calloc((size_t)out * d, sizeof(double));
    double *ab = (double *)calloc((size_t)out, sizeof(double));
    float *dp = (float *)malloc(sizeof(float) * out);
    for (int64_t i = 0; i < N; i++) {
        for (int o = 0; o < out; o++) {
            float g = dy[i * out + o], v = y[i * out + o];
            if (act == 1) g = v > 0.0f ? g : 0.01f * g;
            else if (act == 2) g = v > 0.0f ? g : 0.0f;
            dp[o] = g; ab[o] += g;
        }
        const float *xi = x + i * d;
        for (int o = 0; o < out; o++) { double g = dp[o]; if (g != 0.0) for (int c = 0; c < d; c++) aW[(int64_t)o * d + c] += g * xi[c]; }
        if (dx) for (int c = 0; c < d; c++) {
            double s = 0.0;
            for (int o = 0; o < out; o++) s += (double)dp[o] * (w_layout == 0 ? W[(int64_t)o * d + c] : W[(int64_t)c * out + o]);
            dx[i * d + c] = (float)s;
        }
    }
    for (int o = 0; o < out; o++) for (int c = 0; c < d; c++) {
        float v = (float)aW[(int64_t)o * d + c];
        if (w_layout == 0) dW[(int64_t)o * d + c] = v; else dW[(int64_t)c * out + o] = v;
    }
    if (db) for (int o = 0; o < out; o++) db[o] = (float)ab[o];
    free(aW); free(ab); free(dp);
}

/* mean and unbiased std of the prior degree (dgm.py:1568-1570), accumulated in double, rounded once */
ORA_API void ora_degree_stats(const float *deg, int64_t N, float *mu, float *sd) {
    double s = 0.0;
    for (int64_t i = 0; i < N; i++) s += deg[i];
    double m = s / (double)N, v = 0.0;
    for (int64_t i = 0; i < N; i++) { double t = deg[i] - m; v += t * t; }
    *mu = (float)m;
    *sd = (float)sqrt(v / (double)(N - 1));
}

/* k_estimate_net mode "x" (dgm.py:1562-1586) after the node encoder: xk = leaky(node_encode_for_k(x)).
 * W1 [h2][h+1], b1[h2] = k_embed.0; Wmu [h4][h2], bmu = k_net.k_mu; Wp [h4], bp = k_net.k_project.
 * Saves z (post-leaky, [N,h2]), m ([N,h4]) and u = kp*sd+mu ([N]) when the pointers are non-NULL. */
ORA_API void ora_knet_x(const float *xk, int64_t N, int h, const float *deg, float mu, float sd,
                        const float *W1, const float *b1, int h2, const float *Wmu, const float *bmu, int h4,
                        const float *Wp, const float *bp, float *k, float *z_save, float *m_save, float *u_save) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; i++) {
        float z[256], m[128];
        float nd = (deg[i] - mu) / (sd + 1e-5f);
        for (int o = 0; o < h2; o++) {
            const float *w = W1 + (int64_t)o * (h + 1);
            float acc = 0.0f;
            for (int c = 0; c < h; c++) acc = fmaf(xk[i * h + c], w[c], acc);
            acc = fmaf(nd, w[h], acc);
            acc = acc + b1[o];
            z[o] = acc > 0.0f ? acc : 0.01f * acc;
        }
        for (int o = 0; o < h4; o++) {
            float acc = 0.0f;
            for (int c = 0; c < h2; c++) acc = fmaf(z[c], Wmu[o * h2 + c], acc);
            m[o] = acc + bmu[o];
        }
        float acc = 0.0f;
        for (int c = 0; c < h4; c++) acc = fmaf(m[c], Wp[c], acc);
        float kp = acc + bp[0];
        float u = kp * sd; u = u + mu;
        k[i] = (u > 0.0f ? u : 0.0f) + 1.0f;
        if (z_save) memcpy(z_save + i * h2, z, sizeof(float) * h2);
        if (m_save) memcpy(m_save + i * h4, m, sizeof(float) * h4);
        if (u_save) u_save[i] = u;
    }
}
/* backward of ora_knet_x: dk -> dxk [N,h], dW1,db1,dWmu,dbmu,dWp,dbp */
ORA_API void ora_knet_x_bwd(const float *xk, int64_t N, int h, const float *deg, float mu, float sd,
                            const float *W1, int h2, const float *Wmu, int h4, const float *Wp,
                            const float *z, const float *m, const float *u, const float *dk,
                            float *dxk, float *dW1, float *db1, float *dWmu, float *dbmu, float *dWp, float *dbp) {
    double *aW1 = calloc((size_t)h2 * (h + 1), 8), *ab1 = calloc(h2, 8), *aWmu = calloc((size_t)h4 * h2, 8),
           *abmu = calloc(h4, 8), *aWp = calloc(h4, 8), abp = 0.0;
    for (int64_t i = 0; i < N; i++) {
        float nd = (deg[i] - mu) / (sd + 1e-5f);
        double dkp = u[i] > 0.0f ? (double)dk[i] * sd : 0.0;
        abp += dkp;
        double dm[128], dz[256];
        for (int c = 0; c < h4; c++) { aWp[c] += dkp * m[i * h4 + c]; dm[c] = dkp * Wp[c]; abmu[c] += dm[c]; }
        for (int c = 0; c < h2; c++) dz[c] = 0.0;
        for (int o = 0; o < h4; o++) for (int c = 0; c < h2; c++) { aWmu[o * h2 + c] += dm[o] * z[i * h2 + c]; dz[c] += dm[o] * Wmu[o * h2 + c]; }
        for (int c = 0; c < h; c++) dxk[i * h + c] = 0.0f;
        for (int o = 0; o < h2; o++) {
            double g = z[i * h2 + o] > 0.0f ? dz[o] : 0.01 * dz[o];
            ab1[o] += g;
            for (int c = 0; c < h; c++) { aW1[(int64_t)o * (h + 1) + c] += g * xk[i * h + c]; dxk[i * h + c] += (float)(g * W1[(int64_t)o * (h + 1) + c]); }
            aW1[(int64_t)o * (h + 1) + h] += g * nd;
        }
    }
    for (int64_t t = 0; t < (int64_t)h2 * (h + 1); t++) dW1[t] = (float)aW1[t];
    for (int t = 0; t < h2; t++) db1[t] = (float)ab1[t];
    for (int t = 0; t < h4 * h2; t++) dWmu[t] = (float)aWmu[t];
    for (int t = 0; t < h4; t++) { dbmu[t] = (float)abmu[t]; dWp[t] = (float)aWp[t]; }
    dbp[0] = (float)abp;
    free(aW1); free(ab1); free(aWmu); free(abmu); free(aWp);
}

/* k_estimate_net mode "input_deg" (dgm.py:1509-1526): constants deg_mean/deg_std from args;
 * Wd[3],bd[3] = input_degree_project (Linear(1,3)); Wmu [h4][3], bmu; Wp[h4], bp. */
ORA_API void ora_knet_input_deg(const float *deg, int64_t N, float dmean, float dstd, const float *Wd, const float *bd,
                                const float *Wmu, const float *bmu, int h4, const float *Wp, const float *bp, float *k) {
    for (int64_t i = 0; i < N; i++) {
        float nd = (deg[i] - dmean) / (dstd + 1e-5f);
        float in3[3], m[128];
        for (int o = 0; o < 3; o++) { float acc = fmaf(nd, Wd[o], 0.0f); in3[o] = acc + bd[o]; }
        for (int o = 0; o < h4; o++) { float acc = 0.0f; for (int c = 0; c < 3; c++) acc = fmaf(in3[c], Wmu[o * 3 + c], acc); m[o] = acc + bmu[o]; }
        float acc = 0.0f;
        for (int c = 0; c < h4; c++) acc = fmaf(m[c], Wp[c], acc);
        float kp = acc + bp[0];
        float u = kp * dstd; u = u + dmean;
        k[i] = (u > 0.0f ? u : 0.0f) + 1.0f;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* pair score (dgm.py:1613-1623 + 1211-1229)                                                   */
/* ------------------------------------------------------------------------------------------ */
/* squared-distance accumulation order: one fmaf chain over the features in ascending order for h <= 128; for wider
 * latents (the PPI configuration runs the DGG at 2048) 64 interleaved chains -- feature c goes to chain c mod 64 -- combined
 * by the xor butterfly of a wavefront, i.e. what 64 lanes reading coalesced 256-byte segments compute */
static inline float pair_dist(const float *a, const float *b, int h) {
    if (h <= 128) {
        float d2 = 0.0f;
        for (int c = 0; c < h; c++) { float df = a[c] - b[c]; d2 = fmaf(df, df, d2); }
        return sqrtf(d2);
    }
    float s[64], t[64];
    for (int l = 0; l < 64; l++) s[l] = 0.0f;
    for (int c = 0; c < h; c++) { float df = a[c] - b[c]; s[c & 63] = fmaf(df, df, s[c & 63]); }
    for (int off = 32; off >= 1; off >>= 1) {
        for (int l = 0; l < 64; l++) t[l] = s[l] + s[l ^ off];
        memcpy(s, t, sizeof(s));
    }
    return sqrtf(s[0]);
}
ORA_API float ora_pair_score(const float *xi, const float *xj, int h, float t, int perturb, float G) {
    float p = ora_exp(t * pair_dist(xi, xj, h));
    if (!perturb) return p;
    float lp = ora_log(p + 1e-8f);
    return ora_exp(lp + G);
}

/* selection order: score descending, column ascending; a strict total order */
typedef struct { float v; int32_t j; } cand_t;
static inline int better(float v, int32_t j, float v2, int32_t j2) { return v > v2 || (v == v2 && j < j2); }
static void topk_insert(cand_t *list, int *cnt, int K, float v, int32_t j) {
    int n = *cnt;
    if (n == K && !better(v, j, list[K - 1].v, list[K - 1].j)) return;
    int pos = n < K ? n : K - 1;
    while (pos > 0 && better(v, j, list[pos - 1].v, list[pos - 1].j)) { list[pos] = list[pos - 1]; pos--; }
    list[pos].v = v; list[pos].j = j;
    if (n < K) *cnt = n + 1;
}

/* ------------------------------------------------------------------------------------------ */
/* "ranked symmetric" counter-based noise (noise_mode 5): G_ij = G_ji iid Gumbel(0,0.3) per     */
/* UNORDERED pair, zero diagonal (the reference's symmetric_noise=True, dgm.py:1216-1223),      */
/* generated so that every node can list its largest noises first:                              */
/*   - the pair {i,j} is OWNED by one endpoint: with delta = (j - i) mod N and hmax = (N-1)/2,  */
/*     i owns it when delta <= hmax, j owns it when delta >= N - hmax, and for even N the       */
/*     antipodal pair (delta = N/2) belongs to the smaller index.  Node o owns the offsets      */
/*     1..n_o, n_o = hmax (+1 for even N and o < N/2): partner (o + delta) mod N;               */
/*   - the n_o noises of an owner are the ranked generator above on n_o "columns" (Renyi order  */
/*     statistics in 2^-40 fixed point, rank s placed at offset sigma_o(.) + 1), keyed by o.    */
/* Owners are independent and each unordered pair has exactly one, so the matrix is iid on its  */
/* upper triangle and symmetric.                                                                */
/* ------------------------------------------------------------------------------------------ */
static inline int64_t rsym_owned(int64_t N, int64_t o) {
    return (N - 1) / 2 + ((((N & 1) == 0) && o < N / 2) ? 1 : 0);
}
/* rows [row0,row1) of the noise matrix, G[(i-row0)*N + j].  Every owner's whole sequence is walked (the rank of a pair
 * inside its owner's sequence has no closed form): O(N^2 / 2) whatever the number of rows. */
ORA_API void ora_ranked_sym_block(uint32_t s0, uint32_t s1, int64_t N, int64_t row0, int64_t row1, float *G) {
    for (int64_t i = row0; i < row1; i++) G[(i - row0) * N + i] = 0.0f;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t o = 0; o < N; o++) {
        const int64_t n = rsym_owned(N, o);
        if (n <= 0) continue;
        uint32_t k1, k2;
        ora_rowkey(s0, s1, (uint32_t)o, &k1, &k2);
        const uint32_t k3 = mix32(k2 ^ 0x68E31DA4u);
        const int b = ranked_bits(n);
        const int own = o >= row0 && o < row1;
        uint64_t S = 0;
        uint32_t s = 0;
        for (uint64_t r = 0; r < ((uint64_t)1 << b); r++) {
            uint32_t c = ranked_sigma((uint32_t)r, k1, k2, k3, b);
            if ((int64_t)c >= n) continue;
            s++;
            S += ranked_term(k1, k3, s, n);
            int64_t p = o + (int64_t)c + 1;
            if (p >= N) p -= N;
            const float g = ranked_gumbel(S);
            if (own) G[(o - row0) * N + p] = g;
            if (p >= row0 && p < row1) G[(p - row0) * N + o] = g;
        }
    }
}

/* noise_mode: 0 none (perturb_edge_prob False), 1 explicit G (row-major [N][N], row i at G + i*N), 4 ranked counter-based,
 *             2 counter-based (s0,s1), 3 counter-based symmetric, 5 ranked symmetric */
ORA_API void ora_allpairs_topk(const float *xp, int64_t N, int h, int64_t row0, int64_t row1, float t,
                               int noise_mode, const float *G, uint32_t s0, uint32_t s1,
                               int K, int32_t *idx, float *val) {
    float *gsym = NULL;
    if (noise_mode == 5 && row1 > row0) {
        gsym = (float *)malloc(sizeof(float) * (size_t)(row1 - row0) * (size_t)N);
        ora_ranked_sym_block(s0, s1, N, row0, row1, gsym);
    }
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t i = row0; i < row1; i++) {
        cand_t *list = (cand_t *)malloc(sizeof(cand_t) * (size_t)(K > 0 ? K : 1));   /* (rows wider than the 64-wide list: any K) */
        int cnt = 0;
        float *grow = NULL;
        if (noise_mode == 4) { grow = (float *)malloc(sizeof(float) * (size_t)N); ora_ranked_row(s0, s1, (uint32_t)i, N, grow); }
        for (int64_t j = 0; j < N; j++) {
            float g = 0.0f;
            if (noise_mode == 1) g = G[i * N + j];
            else if (noise_mode == 4) g = grow[j];
            else if (noise_mode == 5) g = gsym[(i - row0) * N + j];
            else if (noise_mode >= 2) g = ora_noise(s0, s1, (uint32_t)i, (uint32_t)j, noise_mode == 3);
            float v = ora_pair_score(xp + i * h, xp + j * h, h, t, noise_mode != 0, g);
            topk_insert(list, &cnt, K, v, (int32_t)j);
        }
        free(grow);
        for (int r = 0; r < K; r++) {
            idx[(i - row0) * K + r] = r < cnt ? list[r].j : -1;
            val[(i - row0) * K + r] = r < cnt ? list[r].v : 0.0f;
        }
        free(list);
    }
    free(gsym);
}

/* candidates restricted to a CSR graph (the live class's in_adj semantics, dgm.py:1613-1627).
 * G (noise_mode 1) is dense [N][N]. */
ORA_API void ora_edgelist_topk(const float *xp, int64_t N, int h, const int64_t *rowptr, const int32_t *col, float t,
                               int noise_mode, const float *G, uint32_t s0, uint32_t s1,
                               int K, int32_t *idx, float *val) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < N; i++) {
        cand_t *list = (cand_t *)malloc(sizeof(cand_t) * (size_t)(K > 0 ? K : 1));   /* (rows wider than the 64-wide list: any K) */
        int cnt = 0;
        for (int64_t e = rowptr[i]; e < rowptr[i + 1]; e++) {
            int32_t j = col[e];
            float g = 0.0f;
            if (noise_mode == 1) g = G[i * N + j];
            else if (noise_mode >= 2) g = ora_noise(s0, s1, (uint32_t)i, (uint32_t)j, noise_mode == 3);
            float v = ora_pair_score(xp + i * h, xp + (int64_t)j * h, h, t, noise_mode != 0, g);
            topk_insert(list, &cnt, K, v, j);
        }
        for (int r = 0; r < K; r++) {
            idx[i * K + r] = r < cnt ? list[r].j : -1;
            val[i * K + r] = r < cnt ? list[r].v : 0.0f;
        }
        free(list);
    }
}

/* selection only: dense score rows [R][N] -> top-K (bit-exact target of the HIP selection kernel) */
ORA_API void ora_select_scores(const float *scores, int64_t R, int64_t N, int K, int32_t *idx, float *val) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < R; i++) {
        cand_t *list = (cand_t *)malloc(sizeof(cand_t) * (size_t)(K > 0 ? K : 1));   /* (rows wider than the 64-wide list: any K) */
        int cnt = 0;
        for (int64_t j = 0; j < N; j++) topk_insert(list, &cnt, K, scores[i * N + j], (int32_t)j);
        for (int r = 0; r < K; r++) { idx[i * K + r] = r < cnt ? list[r].j : -1; val[i * K + r] = r < cnt ? list[r].v : 0.0f; }
        free(list);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* smooth first-k ramp (dgm.py:1410-1417 / 1427-1434), row sums, normalisation (model.py:1215-1218) */
/* mode 0: k_times_edge_prob  w = s * f ; mode 1: k_only  w = f  (only where idx >= 0)          */
/* ------------------------------------------------------------------------------------------ */
static inline float ramp(float r, float k) {
    float th = ora_tanh(r - k);
    float a = 1.0f + th;
    a = 0.5f * a;
    return 1.0f - a;
}
/* row sum in the xor-butterfly order of a 64-lane wavefront reduction (slots >= K are zero) */
static float butterfly_sum(const float *w, int K) {
    float s[64], t[64];
    for (int l = 0; l < 64; l++) s[l] = 0.0f;
    /* slot l holds w[l] + w[l+64] + w[l+128] + ... (a row wider than 64 ranks: lane l of the wavefront adds its entry of every
     * 64-rank chunk in chunk order, then the butterfly) */
    for (int l = 0; l < K; l++) s[l & 63] = (l < 64) ? w[l] : s[l & 63] + w[l];
    for (int off = 32; off >= 1; off >>= 1) {
        for (int l = 0; l < 64; l++) t[l] = s[l] + s[l ^ off];
        memcpy(s, t, sizeof(s));
    }
    return s[0];
}
ORA_API void ora_softk(const int32_t *idx, const float *val, const float *k, int64_t N, int K, int mode,
                       float *w, float *rs) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; i++) {
        for (int r = 0; r < K; r++) {
            float f = ramp((float)r, k[i]);
            float v = f;                                   /* mode 1: the ramp alone */
            if (mode == 0 || mode == 3) {                  /* mode 3: straight-through value (hard - soft) + soft, hard = ramp */
                float a = val[i * K + r] * f;
                v = mode == 0 ? a : (f - a) + a;
            }
            w[i * K + r] = idx[i * K + r] >= 0 ? v : 0.0f;
        }
        rs[i] = butterfly_sum(w + i * K, K);
    }
}
/* A_hat = (a_i * w) * a_j with a = 1/sqrt(rs)  (row sums on BOTH sides, model.py:1215-1218) */
ORA_API void ora_normalize(const int32_t *idx, const float *w, const float *rs, int64_t N, int K, float *ahat) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; i++) {
        float ai = 1.0f / sqrtf(rs[i]);
        for (int r = 0; r < K; r++) {
            int32_t j = idx[i * K + r];
            if (j < 0) { ahat[i * K + r] = 0.0f; continue; }
            float aj = 1.0f / sqrtf(rs[j]);
            ahat[i * K + r] = (ai * w[i * K + r]) * aj;
        }
    }
}
/* Y_i = sum_r ahat_ir * X[idx_ir], r ascending, fmaf chain (torch.mm(adj, x), model.py:594 / spmm 34) */
ORA_API void ora_spmm(const int32_t *idx, const float *ahat, const float *X, int64_t N, int K, int F, float *Y) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; i++) {
        for (int c = 0; c < F; c++) Y[i * F + c] = 0.0f;
        for (int r = 0; r < K; r++) {
            int32_t j = idx[i * K + r];
            if (j < 0) continue;
            float a = ahat[i * K + r];
            for (int c = 0; c < F; c++) Y[i * F + c] = fmaf(a, X[(int64_t)j * F + c], Y[i * F + c]);
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* backward (autograd of the above; accumulations in double, single-threaded and simple)       */
/* ------------------------------------------------------------------------------------------ */
/* dA_ir = <dY_i, X_j>; dX_j += ahat_ir * dY_i */
ORA_API void ora_spmm_bwd(const int32_t *idx, const float *ahat, const float *X, const float *dY, int64_t N, int K,
                          int F, float *dA, float *dX) {
    double *ax = dX ? calloc((size_t)N * F, 8) : NULL;
    for (int64_t i = 0; i < N; i++) for (int r = 0; r < K; r++) {
        int32_t j = idx[i * K + r];
        if (j < 0) { dA[i * K + r] = 0.0f; continue; }
        double s = 0.0;
        for (int c = 0; c < F; c++) s += (double)dY[i * F + c] * X[(int64_t)j * F + c];
        dA[i * K + r] = (float)s;
        if (ax) { double a = ahat[i * K + r]; for (int c = 0; c < F; c++) ax[(int64_t)j * F + c] += a * dY[i * F + c]; }
    }
    if (ax) { for (int64_t t = 0; t < N * F; t++) dX[t] = (float)ax[t]; free(ax); }
}
/* normalisation + ramp backward: dA (wrt ahat) -> dval (wrt sorted scores), dk */
ORA_API void ora_softk_norm_bwd(const int32_t *idx, const float *val, const float *k, const float *w, const float *rs,
                                const float *dA, int64_t N, int K, int mode, float *dval, float *dk) {
    double *da = calloc((size_t)N, 8);
    double *a = malloc(sizeof(double) * N);
    for (int64_t i = 0; i < N; i++) a[i] = 1.0 / sqrt((double)rs[i]);
    for (int64_t i = 0; i < N; i++) for (int r = 0; r < K; r++) {
        int32_t j = idx[i * K + r];
        if (j < 0) continue;
        double g = dA[i * K + r], ww = w[i * K + r];
        da[i] += g * ww * a[j];
        da[j] += g * ww * a[i];
    }
    for (int64_t i = 0; i < N; i++) {
        double drs = -0.5 * da[i] * a[i] / (double)rs[i];
        double sk = 0.0;
        for (int r = 0; r < K; r++) {
            int32_t j = idx[i * K + r];
            if (j < 0) { dval[i * K + r] = 0.0f; continue; }
            double dw = (double)dA[i * K + r] * a[i] * a[j] + drs;
            double th = ora_tanh((float)r - k[i]);
            double f = 1.0 - 0.5 * (1.0 + th);
            double dfdk = 0.5 * (1.0 - th * th);
            if (mode == 0) { dval[i * K + r] = (float)(dw * f); sk += dw * val[i * K + r] * dfdk; }
            else { dval[i * K + r] = 0.0f; sk += dw * dfdk; }
        }
        dk[i] = (float)sk;
    }
    free(da); free(a);
}
/* score backward to the projected features: dxp [N,h] (dgm.py:1613-1623, 1213-1229 under autograd).
 * G explicit dense or counter-based as in ora_allpairs_topk (needed to recompute p from the stored score). */
ORA_API void ora_edge_bwd_rows(const float *xp, int64_t N, int64_t R, int h, const int32_t *idx, const float *val,
                               const float *dval, int K, float t, int perturb, float *dxp);
ORA_API void ora_edge_bwd(const float *xp, int64_t N, int h, const int32_t *idx, const float *val, const float *dval,
                          int K, float t, int perturb, float *dxp) {
    ora_edge_bwd_rows(xp, N, N, h, idx, val, dval, K, t, perturb, dxp);
}
/* same, for the first R rows of an N-node problem (idx/val/dval are [R,K]; columns index all N nodes) */
ORA_API void ora_edge_bwd_rows(const float *xp, int64_t N, int64_t R, int h, const int32_t *idx, const float *val,
                               const float *dval, int K, float t, int perturb, float *dxp) {
    double *acc = calloc((size_t)N * h, 8);
    for (int64_t i = 0; i < R; i++) for (int r = 0; r < K; r++) {
        int32_t j = idx[i * K + r];
        if (j < 0) continue;
        double g = dval[i * K + r];
        if (g == 0.0) continue;
        const float *a = xp + i * h, *b = xp + (int64_t)j * h;
        float dist = pair_dist(a, b, h);
        if (dist == 0.0f) continue;                         /* vector_norm backward at 0 is 0 */
        float p = ora_exp(t * dist);
        double dp;
        if (perturb) dp = g * (double)val[i * K + r] / ((double)p + 1e-8);   /* d exp(log(p+1e-8)+G) / dp */
        else dp = g;
        double dd = dp * (double)t * (double)p / (double)dist;
        for (int c = 0; c < h; c++) {
            double df = (double)a[c] - (double)b[c];
            acc[i * h + c] += dd * df;
            acc[(int64_t)j * h + c] -= dd * df;
        }
    }
    for (int64_t tix = 0; tix < N * h; tix++) dxp[tix] = (float)acc[tix];
    free(acc);
}

/* ------------------------------------------------------------------------------------------ */
/* edge-MLP scorers on a candidate edge list (dgm.py:1628-1719)               */
/*   u-v-A_uv 1628-1644, u-v-deg 1645-1670, u-v-deg-dist 1671-1702, edge_conv 1703-1719        */
/* The reference evaluates  sigmoid(W2 act(W1 [x_u, x_v, extras] + b1) + b2)  per edge.  The    */
/* first layer is split by linearity into per-NODE products AB = xp [Wa | Wb]^T ([N, 2*hw],     */
/* computed by ora_linear) plus the per-edge extras; canonical order per hidden unit o:         */
/*   z = A[u][o] + B[v][o]; z = fma(deg_u, wdu[o], z); z = fma(deg_v, wdv[o], z);               */
/*   z = fma(ex_e, wex[o], z); z += b1[o]; hid = act(z); s = fma(hid, w2[o], s)  (o ascending)  */
/*   p = 1 / (1 + exp(-(s + b2)))                                                               */
/* ex_mode: 0 none, 1 per-edge array ex_in (a_uv), 2 exp(t_ex * ||xp_u - xp_v||) (u-v-deg-dist) */
/* act: 1 LeakyReLU (edge_encode), 0 identity (edge_conv: theta(v-u) + phi(u) is linear)        */
/* ------------------------------------------------------------------------------------------ */
static inline float mlp_edge_p(const float *AB, int hw, int64_t u, int64_t v, const float *deg, float ex, int has_ex,
                               const float *wdu, const float *wdv, const float *wex, const float *b1, const float *w2,
                               float b2, int act) {
    const float *A = AB + u * 2 * hw, *B = AB + v * 2 * hw + hw;
    float s = 0.0f;
    for (int o = 0; o < hw; o++) {
        float z = A[o] + B[o];
        if (deg) { z = fmaf(deg[u], wdu[o], z); z = fmaf(deg[v], wdv[o], z); }
        if (has_ex) z = fmaf(ex, wex[o], z);
        z = z + b1[o];
        float hid = (act == 1) ? (z > 0.0f ? z : 0.01f * z) : z;
        s = fmaf(hid, w2[o], s);
    }
    s = s + b2;
    return 1.0f / (1.0f + ora_exp(-s));
}
ORA_API void ora_edge_mlp_fwd(const float *AB, const float *xp, int64_t N, int h, int hw, const int32_t *erow,
                              const int32_t *col, int64_t E, const float *deg, const float *ex_in, int ex_mode, float t_ex,
                              const float *wdu, const float *wdv, const float *wex, const float *b1, const float *w2,
                              float b2, int act, float *p_edge, float *ex_out) {
    (void)N;
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < E; e++) {
        int64_t u = erow[e], v = col[e];
        float ex = 0.0f;
        if (ex_mode == 1) ex = ex_in[e];
        else if (ex_mode == 2) ex = ora_exp(t_ex * pair_dist(xp + u * h, xp + v * h, h));
        if (ex_out) ex_out[e] = ex;
        p_edge[e] = mlp_edge_p(AB, hw, u, v, deg, ex, ex_mode != 0, wdu, wdv, wex, b1, w2, b2, act);
    }
}
/* perturbation + per-row top-K on given edge probabilities; columns of a row ascending (coalesced COO), ties by
 * column; eid = index of the selected candidate in col[] (-1 for empty slots) */
ORA_API void ora_edgelist_topk_p(const float *p_edge, int64_t N, const int64_t *rowptr, const int32_t *col, int noise_mode,
                                 const float *G, uint32_t s0, uint32_t s1, int K, int32_t *idx, float *val, int32_t *eid) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < N; i++) {
        cand_t *list = (cand_t *)malloc(sizeof(cand_t) * (size_t)(K > 0 ? K : 1));   /* (rows wider than the 64-wide list: any K) */
        int cnt = 0;
        for (int64_t e = rowptr[i]; e < rowptr[i + 1]; e++) {
            int32_t j = col[e];
            float g = 0.0f, v = p_edge[e];
            if (noise_mode == 1) g = G[i * N + j];
            else if (noise_mode >= 2) g = ora_noise(s0, s1, (uint32_t)i, (uint32_t)j, noise_mode == 3);
            if (noise_mode != 0) v = ora_exp(ora_log(v + 1e-8f) + g);
            topk_insert(list, &cnt, K, v, (int32_t)(e - rowptr[i]));           /* payload: position in the row */
        }
        for (int r = 0; r < K; r++) {
            idx[i * K + r] = r < cnt ? col[rowptr[i] + list[r].j] : -1;
            val[i * K + r] = r < cnt ? list[r].v : 0.0f;
            eid[i * K + r] = r < cnt ? (int32_t)(rowptr[i] + list[r].j) : -1;
        }
        free(list);
    }
}
/* backward of the scorer for the selected entries: dval (wrt the stored score) -> dAB [N, 2*hw], parameter gradients
 * dpar = [dwdu hw | dwdv hw | dwex hw | db1 hw | dw2 hw | db2 1], dex [N,K] (wrt the per-edge extra).  Double
 * accumulation, single-threaded and simple. */
ORA_API void ora_edge_mlp_bwd(const float *AB, int64_t N, int hw, const int32_t *idx, const int32_t *eid, const float *val,
                              const float *dval, int K, const float *deg, const float *ex, const float *wdu,
                              const float *wdv, const float *wex, const float *b1, const float *w2, float b2, int act,
                              int perturb, float *dAB, float *dpar, float *dex) {
    double *acc = calloc((size_t)N * 2 * hw, 8), *par = calloc((size_t)5 * hw + 1, 8);
    double *z = malloc(sizeof(double) * hw), *hid = malloc(sizeof(double) * hw);
    for (int64_t i = 0; i < N; i++) for (int r = 0; r < K; r++) {
        int32_t j = idx[i * K + r];
        if (dex) dex[i * K + r] = 0.0f;
        if (j < 0) continue;
        double g = dval[i * K + r];
        double exv = ex ? (double)ex[eid[i * K + r]] : 0.0;
        const float *A = AB + i * 2 * hw, *B = AB + (int64_t)j * 2 * hw + hw;
        double s = 0.0;
        for (int o = 0; o < hw; o++) {
            double zz = (double)A[o] + (double)B[o];
            if (deg) zz += (double)deg[i] * wdu[o] + (double)deg[j] * wdv[o];
            if (ex) zz += exv * wex[o];
            zz += b1[o];
            z[o] = zz;
            hid[o] = (act == 1) ? (zz > 0.0 ? zz : 0.01 * zz) : zz;
            s += hid[o] * w2[o];
        }
        s += b2;
        double p = 1.0 / (1.0 + exp(-s));
        double dp = perturb ? g * (double)val[i * K + r] / (p + 1e-8) : g;
        double ds = dp * p * (1.0 - p);
        double de = 0.0;
        for (int o = 0; o < hw; o++) {
            double dh = ds * w2[o];
            double dz = (act == 1) ? (z[o] > 0.0 ? dh : 0.01 * dh) : dh;
            acc[i * 2 * hw + o] += dz;
            acc[(int64_t)j * 2 * hw + hw + o] += dz;
            if (deg) { par[o] += dz * deg[i]; par[hw + o] += dz * deg[j]; }
            if (ex) { par[2 * hw + o] += dz * exv; de += dz * wex[o]; }
            par[3 * hw + o] += dz;
            par[4 * hw + o] += ds * hid[o];
        }
        par[5 * hw] += ds;
        if (dex) dex[i * K + r] = (float)de;
    }
    for (int64_t e = 0; e < N * 2 * hw; e++) dAB[e] = (float)acc[e];
    for (int e = 0; e < 5 * hw + 1; e++) dpar[e] = (float)par[e];
    free(acc); free(par); free(z); free(hid);
}

/* ------------------------------------------------------------------------------------------ */
/* degree-only k-net modes (dgm.py:1492-1526): "input_deg" (constants deg_mean/deg_std, eps    */
/* 1e-5) and "learn_normalized_degree" (batch mean/std of deg, no eps).  Forward with saved    */
/* pre-relu u; backward: the whole net is affine in the scalar nd_i, so every parameter        */
/* gradient is a combination of S0 = sum_i dkp_i and S1 = sum_i dkp_i nd_i (dkp = dk*sd*[u>0]). */
/* ------------------------------------------------------------------------------------------ */
ORA_API void ora_knet_deg(const float *deg, int64_t N, float mu, float sd, float eps, const float *Wd, const float *bd,
                          const float *Wmu, const float *bmu, int h4, const float *Wp, const float *bp, float *k,
                          float *u_save) {
    for (int64_t i = 0; i < N; i++) {
        float nd = (deg[i] - mu) / (sd + eps);
        float in3[3], m[128];
        for (int o = 0; o < 3; o++) { float acc = fmaf(nd, Wd[o], 0.0f); in3[o] = acc + bd[o]; }
        for (int o = 0; o < h4; o++) { float acc = 0.0f; for (int c = 0; c < 3; c++) acc = fmaf(in3[c], Wmu[o * 3 + c], acc); m[o] = acc + bmu[o]; }
        float acc = 0.0f;
        for (int c = 0; c < h4; c++) acc = fmaf(m[c], Wp[c], acc);
        float kp = acc + bp[0];
        float u = kp * sd; u = u + mu;
        k[i] = (u > 0.0f ? u : 0.0f) + 1.0f;
        if (u_save) u_save[i] = u;
    }
}
/* -> S[0] = sum dkp, S[1] = sum dkp * nd (double accumulation) */
ORA_API void ora_knet_deg_bwd_sums(const float *deg, int64_t N, float mu, float sd, float eps, const float *u,
                                   const float *dk, float *S) {
    double s0 = 0.0, s1 = 0.0;
    for (int64_t i = 0; i < N; i++) {
        double dkp = u[i] > 0.0f ? (double)dk[i] * sd : 0.0;
        double nd = ((double)deg[i] - mu) / ((double)sd + eps);
        s0 += dkp; s1 += dkp * nd;
    }
    S[0] = (float)s0; S[1] = (float)s1;
}

/* ------------------------------------------------------------------------------------------ */
/* CSR adjacency (variable row length): the `DGG` class "for ICLR" keeps EVERY candidate edge  */
/* (weight rank * (ramp + 1), dgm.py:1758-1815), so its output has the sparsity of in_adj and  */
/* rows are not bounded by the ELL width.  Row reductions: lane-strided partial sums (entry e   */
/* of the row goes to slot e mod 64, sequentially) followed by the xor-butterfly of a wavefront */
/* ------------------------------------------------------------------------------------------ */
static float strided_butterfly_sum(const float *v, int64_t n) {
    float s[64], t[64];
    for (int l = 0; l < 64; l++) s[l] = 0.0f;
    for (int64_t e = 0; e < n; e++) s[e & 63] = s[e & 63] + v[e];
    for (int off = 32; off >= 1; off >>= 1) {
        for (int l = 0; l < 64; l++) t[l] = s[l] + s[l ^ off];
        memcpy(s, t, sizeof(s));
    }
    return s[0];
}
ORA_API void ora_csr_row_sum(const float *vals, const int64_t *rowptr, int64_t N, float *rs) {
    for (int64_t i = 0; i < N; i++) rs[i] = strided_butterfly_sum(vals + rowptr[i], rowptr[i + 1] - rowptr[i]);
}
/* A_hat = (a_i * w) * a_j, a = 1/sqrt(rs)   (model.py:1340-1352 normalize_adj of GCN_DGG_00, same as 1205-1219) */
ORA_API void ora_csr_normalize(const int64_t *rowptr, const int32_t *col, const float *w, const float *rs, int64_t N,
                               float *ahat) {
    for (int64_t i = 0; i < N; i++) {
        float ai = 1.0f / sqrtf(rs[i]);
        for (int64_t e = rowptr[i]; e < rowptr[i + 1]; e++) ahat[e] = (ai * w[e]) * (1.0f / sqrtf(rs[col[e]]));
    }
}
ORA_API void ora_csr_spmm(const int64_t *rowptr, const int32_t *col, const float *a, const float *X, int64_t N, int F,
                          float *Y) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < N; i++) {
        for (int c = 0; c < F; c++) Y[i * F + c] = 0.0f;
        for (int64_t e = rowptr[i]; e < rowptr[i + 1]; e++)
            for (int c = 0; c < F; c++) Y[i * F + c] = fmaf(a[e], X[(int64_t)col[e] * F + c], Y[i * F + c]);
    }
}
ORA_API void ora_csr_spmm_bwd(const int64_t *rowptr, const int32_t *col, const float *a, const float *X, const float *dY,
                              int64_t N, int F, float *dA, float *dX) {
    double *acc = calloc((size_t)N * F, 8);
    for (int64_t i = 0; i < N; i++) for (int64_t e = rowptr[i]; e < rowptr[i + 1]; e++) {
        int64_t j = col[e];
        double s = 0.0;
        for (int c = 0; c < F; c++) { s += (double)dY[i * F + c] * X[j * F + c]; acc[j * F + c] += (double)a[e] * dY[i * F + c]; }
        dA[e] = (float)s;
    }
    if (dX) for (int64_t q = 0; q < N * F; q++) dX[q] = (float)acc[q];
    free(acc);
}
/* backward of ora_csr_row_sum + ora_csr_normalize: dA (wrt ahat) -> dw */
ORA_API void ora_csr_norm_bwd(const int64_t *rowptr, const int32_t *col, const float *w, const float *rs, const float *dA,
                              int64_t N, float *dw) {
    double *da = calloc((size_t)N, 8), *a = malloc(sizeof(double) * N);
    for (int64_t i = 0; i < N; i++) a[i] = 1.0 / sqrt((double)rs[i]);
    for (int64_t i = 0; i < N; i++) for (int64_t e = rowptr[i]; e < rowptr[i + 1]; e++) {
        double g = (double)dA[e] * w[e];
        da[i] += g * a[col[e]];
        da[col[e]] += g * a[i];
    }
    for (int64_t i = 0; i < N; i++) {
        double drs = -0.5 * da[i] * a[i] / (double)rs[i];
        for (int64_t e = rowptr[i]; e < rowptr[i + 1]; e++) dw[e] = (float)((double)dA[e] * a[i] * a[col[e]] + drs);
    }
    free(da); free(a);
}
/* `DGG.forward` after the edge ranks (dgm.py:1791-1812): S_i = sum_j rank_ij; k_i = leaky(S_i * w + b) (degree_decoder,
 * Linear(1,1) + LeakyReLU); position of every edge in its row sorted by (rank desc, column asc);
 * out_e = rank_e * ((1 - 0.5 (1 + tanh(pos - k))) + 1).  Saves S, k, pos. */
ORA_API void ora_csr_rank_ramp(const float *p, const int64_t *rowptr, const int32_t *col, int64_t N, float w, float b,
                               float *out, float *S, float *k, int32_t *pos) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < N; i++) {
        int64_t e0 = rowptr[i], e1 = rowptr[i + 1];
        S[i] = strided_butterfly_sum(p + e0, e1 - e0);
        float z = S[i] * w; z = z + b;
        k[i] = z > 0.0f ? z : 0.01f * z;
        for (int64_t e = e0; e < e1; e++) {
            int c = 0;
            for (int64_t q = e0; q < e1; q++) c += better(p[q], col[q], p[e], col[e]);
            pos[e] = c;
            float f = ramp((float)c, k[i]);
            f = f + 1.0f;
            out[e] = p[e] * f;
        }
    }
}
/* backward: g = d out -> dp (incl. the path through S -> k), dkz_i = d loss / d (S_i w + b) (for dw, db) */
ORA_API void ora_csr_rank_ramp_bwd(const float *p, const int64_t *rowptr, int64_t N, float w, float b, const float *S,
                                   const float *k, const int32_t *pos, const float *g, float *dp, float *dkz) {
    for (int64_t i = 0; i < N; i++) {
        double dk = 0.0;
        for (int64_t e = rowptr[i]; e < rowptr[i + 1]; e++) {
            double th = ora_tanh((float)pos[e] - k[i]);
            dk += (double)g[e] * p[e] * 0.5 * (1.0 - th * th);
        }
        double z = (double)S[i] * w + b;
        double dz = z > 0.0 ? dk : 0.01 * dk;
        dkz[i] = (float)dz;
        for (int64_t e = rowptr[i]; e < rowptr[i + 1]; e++) {
            double th = ora_tanh((float)pos[e] - k[i]);
            double f = 1.0 - 0.5 * (1.0 + th) + 1.0;
            dp[e] = (float)((double)g[e] * f + dz * w);
        }
    }
}
/* `DGG_Ablations.forward` (dgm.py:1930-1933): edge_rank = sigmoid(sigmoid(score) + noise) */
ORA_API void ora_csr_noisy_sigmoid(const float *p, const float *noise, int64_t E, float *out) {
    for (int64_t e = 0; e < E; e++) { float z = p[e] + noise[e]; out[e] = 1.0f / (1.0f + ora_exp(-z)); }
}
/* fixed k (dgm.py:1940-1942): the kcut best entries of a row keep their rank, the others become 0 */
ORA_API void ora_csr_rank_cut(const float *p, const int64_t *rowptr, const int32_t *col, int64_t N, int kcut, float *out,
                              int32_t *pos) {
    for (int64_t i = 0; i < N; i++) for (int64_t e = rowptr[i]; e < rowptr[i + 1]; e++) {
        int c = 0;
        for (int64_t q = rowptr[i]; q < rowptr[i + 1]; q++) c += better(p[q], col[q], p[e], col[e]);
        pos[e] = c;
        out[e] = c < kcut ? p[e] : 0.0f;
    }
}
/* raw edge probabilities as the adjacency (debug_step 0/1, edge_p-cdf): u-v-dist scorer on the stored entries */
ORA_API void ora_csr_uvdist(const float *xp, const int64_t *rowptr, const int32_t *col, int64_t N, int h, float t, float *p) {
    for (int64_t i = 0; i < N; i++) for (int64_t e = rowptr[i]; e < rowptr[i + 1]; e++)
        p[e] = ora_exp(t * pair_dist(xp + i * h, xp + (int64_t)col[e] * h, h));
}
ORA_API void ora_csr_uvdist_bwd(const float *xp, const int64_t *rowptr, const int32_t *col, int64_t N, int h, float t,
                                const float *p, const float *dp, float *dxp) {
    double *acc = calloc((size_t)N * h, 8);
    for (int64_t i = 0; i < N; i++) for (int64_t e = rowptr[i]; e < rowptr[i + 1]; e++) {
        int64_t j = col[e];
        double d2 = 0.0;
        for (int c = 0; c < h; c++) { double df = (double)xp[i * h + c] - xp[j * h + c]; d2 += df * df; }
        if (!(d2 > 0.0)) continue;
        double coef = (double)dp[e] * p[e] * t / sqrt(d2);
        for (int c = 0; c < h; c++) {
            double v = coef * ((double)xp[i * h + c] - xp[j * h + c]);
            acc[i * h + c] += v; acc[j * h + c] -= v;
        }
    }
    for (int64_t q = 0; q < N * h; q++) dxp[q] = (float)acc[q];
    free(acc);
}
/* ora_edge_mlp_bwd on a CSR-valued adjacency: entry e of row i is (i, col[e]) with cotangent dval[e] */
ORA_API void ora_edge_mlp_bwd_csr(const float *AB, int64_t N, int hw, const int64_t *rowptr, const int32_t *col,
                                  const float *dval, const float *b1, const float *w2, float b2, int act, float *dAB,
                                  float *dpar) {
    double *acc = calloc((size_t)N * 2 * hw, 8), *par = calloc((size_t)5 * hw + 1, 8);
    double *z = malloc(sizeof(double) * hw), *hid = malloc(sizeof(double) * hw);
    for (int64_t i = 0; i < N; i++) for (int64_t e = rowptr[i]; e < rowptr[i + 1]; e++) {
        int64_t j = col[e];
        const float *A = AB + i * 2 * hw, *B = AB + j * 2 * hw + hw;
        double s = 0.0;
        for (int o = 0; o < hw; o++) {
            double zz = (double)A[o] + (double)B[o] + b1[o];
            z[o] = zz; hid[o] = (act == 1) ? (zz > 0.0 ? zz : 0.01 * zz) : zz;
            s += hid[o] * w2[o];
        }
        s += b2;
        double pp = 1.0 / (1.0 + exp(-s));
        double ds = (double)dval[e] * pp * (1.0 - pp);
        for (int o = 0; o < hw; o++) {
            double dh = ds * w2[o];
            double dz = (act == 1) ? (z[o] > 0.0 ? dh : 0.01 * dh) : dh;
            acc[i * 2 * hw + o] += dz; acc[j * 2 * hw + hw + o] += dz;
            par[3 * hw + o] += dz; par[4 * hw + o] += ds * hid[o];
        }
        par[5 * hw] += ds;
    }
    for (int64_t q = 0; q < N * 2 * hw; q++) dAB[q] = (float)acc[q];
    for (int q = 0; q < 5 * hw + 1; q++) dpar[q] = (float)par[q];
    free(acc); free(par); free(z); free(hid);
}

/* ------------------------------------------------------------------------------------------ */
/* Dense all-pairs alternates: DGG_LearnableK_SDD (dgm.py:259-351, dist_fn="metric", noise off) */
/* and DGG_StraightThrough (dgm.py:140-182 + 63-100).  Both return a DENSE [B,N,N] adjacency    */
/* whose rows are a softmax over ALL N columns, so the backward couples every pair: these are   */
/* O(N^2) by definition (the reference's use is B small graphs of few nodes).                   */
/*   prob = exp(-t dist); log_p = log(prob); y = softmax(log_p / temp) over the row             */
/*   pos_j = position of column j in the row sorted by (y desc, column asc)                     */
/*   ramp 0 (SDD, dgm.py:315-346): f = sigmoid((hs_start - interval pos) + interval (k_i - 1)); */
/*            soft: out = y f; hard: out = (f - y f) + y f                                      */
/*   ramp 1 (ST, dgm.py:83-98): soft: out = y; hard: out = (1[pos < kfix] - y) + y              */
/* Row reductions in the order of a 64-lane wavefront (strided partial sums + xor butterfly).   */
/* ------------------------------------------------------------------------------------------ */
static float strided_butterfly_max(const float *v, int64_t n) {
    float m = -INFINITY;
    for (int64_t e = 0; e < n; e++) m = v[e] > m ? v[e] : m;
    return m;
}
ORA_API void ora_dense_rows_fwd(const float *xq, int B, int64_t N, int h, float t, float temp, int ramp, const float *k,
                                int kfix, float hs_start, float interval, int hard, float *out, float *y, int32_t *pos) {
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t bi = 0; bi < (int64_t)B * N; bi++) {
        const float *X = xq + (bi / N) * N * h, *xi = xq + bi * h;
        float *yr = y + bi * N, *orow = out + bi * N;
        int32_t *pr = pos + bi * N;
        for (int64_t j = 0; j < N; j++) {
            float d = pair_dist(xi, X + j * h, h);
            float p = ora_exp(-t * d);
            yr[j] = ora_log(p) / temp;
        }
        float m = strided_butterfly_max(yr, N);
        for (int64_t j = 0; j < N; j++) yr[j] = ora_exp(yr[j] - m);
        float Z = strided_butterfly_sum(yr, N);
        for (int64_t j = 0; j < N; j++) yr[j] = yr[j] / Z;
        for (int64_t j = 0; j < N; j++) {
            int c = 0;
            for (int64_t q = 0; q < N; q++) c += better(yr[q], (int32_t)q, yr[j], (int32_t)j);
            pr[j] = c;
            if (ramp == 0) {
                float xs = hs_start - interval * (float)c;
                float sh = (k[bi] - 1.0f) * interval;
                float z = xs + sh;
                float f = 1.0f / (1.0f + ora_exp(-z));
                float a = yr[j] * f;
                orow[j] = hard ? (f - a) + a : a;
            } else {
                float ind = c < kfix ? 1.0f : 0.0f;
                orow[j] = hard ? (ind - yr[j]) + yr[j] : yr[j];
            }
        }
    }
}
/* g = d loss / d out -> Cm [B,N,N] (coefficient of (x_i - x_j) in d loss / d xq_i contributed by row i), dk [B*N] (ramp 0),
 * dt_rows [B*N] (sum = d loss / d t) */
ORA_API void ora_dense_rows_bwd(const float *xq, int B, int64_t N, int h, float t, float temp, int ramp, const float *k,
                                float hs_start, float interval, const float *y, const int32_t *pos, const float *g, float *Cm,
                                float *dk, float *dt_rows) {
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t bi = 0; bi < (int64_t)B * N; bi++) {
        const float *X = xq + (bi / N) * N * h, *xi = xq + bi * h;
        const float *yr = y + bi * N, *gr = g + bi * N;
        const int32_t *pr = pos + bi * N;
        double S = 0.0, dkk = 0.0;
        double *dy = malloc(sizeof(double) * N);
        for (int64_t j = 0; j < N; j++) {
            if (ramp == 0) {
                double z = ((double)hs_start - (double)interval * pr[j]) + ((double)k[bi] - 1.0) * interval;
                double f = 1.0 / (1.0 + exp(-z));
                dy[j] = gr[j] * f;
                dkk += (double)gr[j] * yr[j] * f * (1.0 - f) * interval;
            } else dy[j] = gr[j];
            S += yr[j] * dy[j];
        }
        double dt = 0.0;
        for (int64_t j = 0; j < N; j++) {
            double dlp = yr[j] * (dy[j] - S) / temp;       /* = d loss / d (-t d_ij): log(exp(.)) is the identity */
            double d = pair_dist(xi, X + j * h, h);
            dt += dlp * (-d);
            Cm[bi * N + j] = d > 0.0 ? (float)(-t * dlp / d) : 0.0f;
        }
        if (dk) dk[bi] = (float)dkk;
        dt_rows[bi] = (float)dt;
        free(dy);
    }
}
/* d loss / d xq_i = sum_j (C_ij + C_ji) (xq_i - xq_j)   (row i's own terms + its appearances as a column) */
ORA_API void ora_dense_pairs_dx(const float *xq, int B, int64_t N, int h, const float *Cm, float *dxq) {
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t bi = 0; bi < (int64_t)B * N; bi++) {
        int64_t b = bi / N, i = bi % N;
        const float *X = xq + b * N * h, *C0 = Cm + b * N * N;
        for (int c = 0; c < h; c++) {
            double s = 0.0;
            for (int64_t j = 0; j < N; j++) s += ((double)C0[i * N + j] + C0[j * N + i]) * ((double)X[i * h + c] - X[j * h + c]);
            dxq[bi * h + c] = (float)s;
        }
    }
}
/* nn.Softmax(dim=-1) over the latent features (input_project of the SDD class, dgm.py:217-221) */
ORA_API void ora_feat_softmax(const float *z, int64_t rows, int h, float *out) {
    for (int64_t r = 0; r < rows; r++) {
        const float *zr = z + r * h;
        float *o = out + r * h;
        float m = strided_butterfly_max(zr, h);
        for (int c = 0; c < h; c++) o[c] = ora_exp(zr[c] - m);
        float Z = strided_butterfly_sum(o, h);
        for (int c = 0; c < h; c++) o[c] = o[c] / Z;
    }
}
ORA_API void ora_feat_softmax_bwd(const float *out, const float *g, int64_t rows, int h, float *dz) {
    for (int64_t r = 0; r < rows; r++) {
        double S = 0.0;
        for (int c = 0; c < h; c++) S += (double)out[r * h + c] * g[r * h + c];
        for (int c = 0; c < h; c++) dz[r * h + c] = (float)(out[r * h + c] * (g[r * h + c] - S));
    }
}
