// dgg_common.h -- device-side building blocks shared by the gfx950 kernels.
//
// Canonical arithmetic: every score-path value is produced by an explicit, fixed sequence of IEEE-754
// binary32 operations (__fadd_rn/__fmul_rn/__fmaf_rn/__fdiv_rn/__fsqrt_rn) so that results do not
// depend on compiler contraction or on hardware transcendental approximations, and top-k indices are
// reproducible bit-for-bit.  The sequences restate the reference's math:
//   exp(t*||u-v||), log(p+1e-8)+G, exp(.)          reference dgm.py:1618-1623, 1213-1229
//   1 - 0.5*(1 + tanh(r - k))                      reference dgm.py:1410-1414
// (the reference evaluates them with torch CPU kernels; agreement with it is to a few ulp).
#pragma once
#include <utility>
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DGG_WAVE 64

namespace dgg {

// Index of this wavefront inside its workgroup, as a SCALAR: `threadIdx.x >> 6` is the same value in every lane, but the compiler
// cannot know that, so everything derived from it (row index, row pointers) lives in vector registers and "wave-uniform" loads
// become vector loads (the row's own 64 features of allpairs_topk_ranked: 64 registers and 16 vector loads per block; the kernel
// went from 334 to 257 us with this one change).  readfirstlane makes the uniformity explicit.
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }


__device__ __forceinline__ float f_from_bits(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t bits_from_f(float f) { return __float_as_uint(f); }

// IEEE correctly rounded sqrt / div.  NOTE: HIP's __fsqrt_rn maps to the NATIVE (approximate) sqrt unless
// OCML_BASIC_ROUNDED_OPERATIONS is defined; sqrtf and '/' are correctly rounded under hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt.
__device__ __forceinline__ float c_sqrt(float x) { return sqrtf(x); }

// exp: clamp to [-87, 88]; n = rint(x*log2e); Cody-Waite; degree-5 Horner; exact power-of-two scaling
__device__ __forceinline__ float c_exp(float x) {
    x = fminf(x, 88.0f);
    x = fmaxf(x, -87.0f);
    float n = __builtin_rintf(__fmul_rn(x, 1.44269504088896341f));
    float r = __fmaf_rn(n, -0.693359375f, x);
    r = __fmaf_rn(n, 2.12194440e-4f, r);
    float q = 1.9875691500e-4f;
    q = __fmaf_rn(q, r, 1.3981999507e-3f);
    q = __fmaf_rn(q, r, 8.3334519073e-3f);
    q = __fmaf_rn(q, r, 4.1665795894e-2f);
    q = __fmaf_rn(q, r, 1.6666665459e-1f);
    q = __fmaf_rn(q, r, 5.0000001201e-1f);
    float r2 = __fmul_rn(r, r);
    float y = __fmaf_rn(q, r2, r);
    y = __fadd_rn(y, 1.0f);
    int e = (int)n;
    return __fmul_rn(y, f_from_bits((uint32_t)(e + 127) << 23));
}

// log of a positive normal float
__device__ __forceinline__ float c_log(float x) {
    uint32_t ux = bits_from_f(x);
    int e = (int)(ux >> 23) - 126;
    float m = f_from_bits((ux & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = __fadd_rn(m, m); }
    m = __fadd_rn(m, -1.0f);
    float z = __fmul_rn(m, m);
    float q = 7.0376836292e-2f;
    q = __fmaf_rn(q, m, -1.1514610310e-1f);
    q = __fmaf_rn(q, m, 1.1676998740e-1f);
    q = __fmaf_rn(q, m, -1.2420140846e-1f);
    q = __fmaf_rn(q, m, 1.4249322787e-1f);
    q = __fmaf_rn(q, m, -1.6668057665e-1f);
    q = __fmaf_rn(q, m, 2.0000714765e-1f);
    q = __fmaf_rn(q, m, -2.4999993993e-1f);
    q = __fmaf_rn(q, m, 3.3333331174e-1f);
    float y = __fmul_rn(__fmul_rn(q, m), z);
    float fe = (float)e;
    y = __fmaf_rn(fe, -2.12194440e-4f, y);
    y = __fmaf_rn(z, -0.5f, y);
    float r = __fadd_rn(m, y);
    r = __fmaf_rn(fe, 0.693359375f, r);
    return r;
}

__device__ __forceinline__ float c_tanh(float x) {
    float ax = fabsf(x);
    if (ax < 0.625f) {
        float z = __fmul_rn(x, x);
        float q = -5.70498872745e-3f;
        q = __fmaf_rn(q, z, 2.06390887954e-2f);
        q = __fmaf_rn(q, z, -5.37397155531e-2f);
        q = __fmaf_rn(q, z, 1.33314422036e-1f);
        q = __fmaf_rn(q, z, -3.33332819422e-1f);
        q = __fmul_rn(q, z);
        return __fmaf_rn(q, x, x);
    }
    float r;
    if (ax > 9.1f) r = 1.0f;
    else {
        float e = c_exp(__fadd_rn(ax, ax));
        r = __fadd_rn(1.0f, -__fdiv_rn(2.0f, __fadd_rn(e, 1.0f)));
    }
    return x < 0.0f ? -r : r;
}

// smooth first-k ramp 1 - 0.5*(1 + tanh(r - k))   (reference dgm.py:1412-1414, w = 1)
__device__ __forceinline__ float c_ramp(float r, float k) {
    float th = c_tanh(__fadd_rn(r, -k));
    float a = __fadd_rn(1.0f, th);
    a = __fmul_rn(0.5f, a);
    return __fadd_rn(1.0f, -a);
}

// ---- counter-based noise ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
// dropout mask of the fused GCNII stack: element e of a layer's activation is KEPT iff its 24-bit hash reaches thr24 = p * 2^24
// (counter-based: the backward recomputes the mask instead of storing it)
__device__ __forceinline__ bool drop_keep(uint32_t s0, uint32_t s1, uint32_t e, uint32_t thr24) {
    const uint32_t x = mix32(mix32(e ^ s0) ^ s1);
    return (x >> 8) >= thr24;
}
__device__ __forceinline__ void rowkey(uint32_t s0, uint32_t s1, uint32_t i, uint32_t &k1, uint32_t &k2) {
    k1 = mix32(i ^ s0);
    k2 = mix32(k1 ^ s1 ^ 0x9E3779B9U);
}
__device__ __forceinline__ uint32_t pair_u24_keyed(uint32_t k1, uint32_t k2, uint32_t b) {
    uint32_t x = b ^ k1;
    x *= 0x7feb352dU; x ^= x >> 15; x += k2; x *= 0x846ca68bU;
    return x >> 8;
}
__device__ __forceinline__ uint32_t pair_u24(uint32_t s0, uint32_t s1, uint32_t i, uint32_t j, bool symmetric) {
    uint32_t a = i, b = j;
    if (symmetric && j < i) { a = j; b = i; }
    uint32_t k1, k2;
    rowkey(s0, s1, a, k1, k2);
    return pair_u24_keyed(k1, k2, b);
}
__device__ __forceinline__ float gumbel_u24(uint32_t u24) {
    if (u24 == 0) u24 = 1;
    float U = __fmul_rn((float)u24, 5.9604644775390625e-8f);
    float a = -c_log(U);
    float b = c_log(a);
    return __fmul_rn(-0.3f, b);
}
__device__ __forceinline__ float pair_noise(uint32_t s0, uint32_t s1, uint32_t i, uint32_t j, bool symmetric) {
    if (symmetric && i == j) return 0.0f;
    return gumbel_u24(pair_u24(s0, s1, i, j, symmetric));
}

// ---- "ranked" counter-based noise (noise_mode 4): same iid Gumbel(0,0.3) law, generated per row in decreasing
// order.  -log U_(s) = sum_{t<=s} E_t/(n-t+1) (Renyi), prefix sums in exact 2^-40 fixed point (order-independent);
// rank s sits at slot sigma_i(r'), the s-th element < n of a keyed bijection of [0, 2^b) walked in order.
// Round 6: the sequence covers the n = N - 1 OTHER columns (slot c -> column c + (c >= i)); the row's own column has an independent
// variate (DGG_RANKED_DIAG_KEY: -log V from the same exponential generator, no division, then the same -0.3 log) and is visited FIRST
// by the searches (position 0 of the walk): every rank a search has not reached is another node, so its score is bounded by the row's
// nearest-neighbour distance (dgg_allpairs_rowmin_bound) and not just by its noise.  Oracle: ora_ranked_row.
#define DGG_RANKED_DIAG_KEY 0xA5A5A5A5u
// Number of leading ranks of a row that can carry a non-zero soft top-k weight: the ramp 1 - 0.5 (1 + tanh(r - k))
// (dgm.py:1412-1420) is exactly 0.0f in fp32 for r - k >= 8.5, so ranks r >= ceil(k + 8.5) never matter (+1 margin).
__host__ __device__ inline int klimit_len(float k, int K) {
    const float L = ceilf(k + 8.5f) + 1.0f;
    if (!(L < (float)K)) return K;                               // also NaN
    return L < 1.0f ? 1 : (int)L;
}
__device__ __forceinline__ int ranked_bits(int64_t N) { int b = 6; while (((int64_t)1 << b) < N) b++; return b; }
__device__ __forceinline__ uint32_t ranked_sigma(uint32_t r, uint32_t k1, uint32_t k2, uint32_t k3, int b) {
    const uint32_t mask = (b >= 32) ? 0xffffffffu : ((1u << b) - 1u);
    const int hb = (b + 1) / 2;
    uint32_t x = (r ^ k1) & mask;
    x = (x * 0x9E3779B1u) & mask; x ^= x >> hb;
    x = (x + k2) & mask; x = (x * 0x85EBCA77u) & mask; x ^= x >> hb;
    x = (x * 0xC2B2AE3Du + k3) & mask; x ^= x >> hb;
    return x;
}
__device__ __forceinline__ uint64_t ranked_term(uint32_t k1, uint32_t k3, uint32_t s, int64_t N) {   // s: 1-based rank
    uint32_t v = mix32(mix32(s + k3) ^ k1) >> 8;
    if (v == 0) v = 1;
    float V = __fmul_rn((float)v, 5.9604644775390625e-8f);
    float E = -c_log(V);
    float term = __fdiv_rn(E, (float)(N - (int64_t)s + 1));
    return (uint64_t)__fmul_rn(term, 1099511627776.0f);         // floor(term * 2^40), exact
}
__device__ __forceinline__ float ranked_gumbel(uint64_t S) {
    uint32_t q = (uint32_t)(S >> 16);
    if (q == 0) q = 1;
    float L = __fmul_rn((float)q, 5.9604644775390625e-8f);
    return __fmul_rn(-0.3f, c_log(L));
}

// squared distance of two rows in the canonical accumulation order, evaluated by ONE thread: a single ascending fmaf chain
// for h <= 128; for wider latents 64 interleaved chains (feature c -> chain c mod 64) combined by the xor butterfly of a
// wavefront -- the order the wave-parallel kernels produce with coalesced loads (oracle: pair_dist)
__device__ __forceinline__ float pair_d2_thread(const float *__restrict__ a, const float *__restrict__ b, int h) {
    if (h <= 128) {
        float d2 = 0.0f;
        for (int c = 0; c < h; c++) {
            const float df = __fadd_rn(a[c], -b[c]);
            d2 = __fmaf_rn(df, df, d2);
        }
        return d2;
    }
    float s[64];
#pragma unroll
    for (int l = 0; l < 64; l++) s[l] = 0.0f;
    for (int c0 = 0; c0 < h; c0 += 64) {
#pragma unroll
        for (int l = 0; l < 64; l++) {
            if (c0 + l < h) {
                const float df = __fadd_rn(a[c0 + l], -b[c0 + l]);
                s[l] = __fmaf_rn(df, df, s[l]);
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
        for (int l = 0; l < 64; l++) {
            if ((l & off) == 0) { const float t = __fadd_rn(s[l], s[l ^ off]); s[l] = t; s[l ^ off] = t; }
        }
    }
    return s[0];
}

// score of a pair given the distance (reference dgm.py:1623, 1213-1229)
__device__ __forceinline__ float score_from_dist(float dist, float t, bool perturb, float G) {
    float p = c_exp(__fmul_rn(t, dist));
    if (!perturb) return p;
    float lp = c_log(__fadd_rn(p, 1e-8f));
    return c_exp(__fadd_rn(lp, G));
}

// selection key: (score desc, column asc) packed so that a larger 64-bit key is better.
// scores are non-negative finite floats, whose bit patterns are monotone.
__device__ __forceinline__ uint64_t make_key(float v, int32_t col) {
    return ((uint64_t)bits_from_f(v) << 32) | (uint32_t)(0x7fffffff - col);
}
__device__ __forceinline__ float key_val(uint64_t k) { return f_from_bits((uint32_t)(k >> 32)); }
__device__ __forceinline__ int32_t key_col(uint64_t k) { return 0x7fffffff - (int32_t)(uint32_t)(k & 0xffffffffu); }
// empty slot: key 0 (value +0.0, column 0x7fffffff) -- ranks below every real candidate except an exact
// zero score at a larger column, which cannot exist.
#define DGG_EMPTY_KEY 0ull

// broadcast from a WAVE-UNIFORM lane index: v_readlane (scalar path) instead of a ds_bpermute round trip
__device__ __forceinline__ float bcast(float v, int uniform_lane) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), uniform_lane));
}
__device__ __forceinline__ int32_t bcast(int32_t v, int uniform_lane) { return __builtin_amdgcn_readlane(v, uniform_lane); }

__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl(lo, src, 64);
    hi = __shfl(hi, src, 64);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int mask) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_xor(lo, mask, 64);
    hi = __shfl_xor(hi, mask, 64);
    return ((uint64_t)hi << 32) | lo;
}

// v of lane U of every 16-lane row (DPP row_share: one vector instruction, no LDS crossbar -- __shfl is a ds_bpermute), for all
// U = 0..15 at once: out[U] = v[16 (lane / 16) + U]
template <int... U>
__device__ __forceinline__ void row16_all_impl(int v, int (&out)[16], std::integer_sequence<int, U...>) {
    ((out[U] = __builtin_amdgcn_update_dpp(0, v, 0x150 + U, 0xF, 0xF, false)), ...);
}
__device__ __forceinline__ void row16_all(int v, int (&out)[16]) { row16_all_impl(v, out, std::make_integer_sequence<int, 16>{}); }
// the first N of them: out[U] = v of lane U of the caller's 16-lane row, U < N
template <int N, int... U>
__device__ __forceinline__ void row16_first_impl(int v, int (&out)[N], std::integer_sequence<int, U...>) {
    ((out[U] = __builtin_amdgcn_update_dpp(0, v, 0x150 + U, 0xF, 0xF, false)), ...);
}
template <int N>
__device__ __forceinline__ void row16_first(int v, int (&out)[N]) { row16_first_impl<N>(v, out, std::make_integer_sequence<int, N>{}); }

// lane ^ D exchange without the LDS crossbar: DPP quad_perm / row shifts / row_ror for D < 16,
// v_permlane16_swap / v_permlane32_swap (gfx950) for D = 16 / 32.  ~1-3 VALU ops instead of a ds_bpermute round trip.
template <int D>
__device__ __forceinline__ uint32_t xor_shfl(uint32_t v, int lane) {
    if constexpr (D == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
    else if constexpr (D == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    else if constexpr (D == 4) {
        int t = __builtin_amdgcn_update_dpp((int)v, (int)v, 0x104, 0xF, 0x5, false);                      // row_shl:4 -> banks 0,2
        return (uint32_t)__builtin_amdgcn_update_dpp(t, (int)v, 0x114, 0xF, 0xA, false);                  // row_shr:4 -> banks 1,3
    } else if constexpr (D == 8) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xF, 0xF, true); // row_ror:8
    else if constexpr (D == 16) {
        auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (lane & 16) ? r[0] : r[1];
    } else {
        auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return (lane & 32) ? r[0] : r[1];
    }
}
template <int D>
__device__ __forceinline__ uint64_t xor_shfl_u64(uint64_t v, int lane) {
    uint32_t lo = xor_shfl<D>((uint32_t)v, lane), hi = xor_shfl<D>((uint32_t)(v >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}

// one compare-exchange stage of the bitonic network: block size KB, distance D; DESC = final order descending
template <int KB, int D, bool DESC>
__device__ __forceinline__ uint64_t bitonic_stage(uint64_t key, int lane) {
    uint64_t other = xor_shfl_u64<D>(key, lane);
    bool up = ((lane & KB) == 0) || KB == 64;       // this block is sorted in the final direction
    bool lower = ((lane & D) == 0);
    bool take_max = DESC ? (up == lower) : (up != lower);
    bool gt = key > other;
    return (gt == take_max) ? key : other;
}
template <int KB, int D, bool DESC>
__device__ __forceinline__ uint64_t bitonic_block(uint64_t key, int lane) {
    key = bitonic_stage<KB, D, DESC>(key, lane);
    if constexpr (D > 1) key = bitonic_block<KB, D / 2, DESC>(key, lane);
    return key;
}
// 64-lane bitonic sort by key; DESC: lane 0 ends with the largest key, else lane 0 the smallest
template <bool DESC>
__device__ __forceinline__ uint64_t wave_sort(uint64_t key, int lane) {
    key = bitonic_block<2, 1, DESC>(key, lane);
    key = bitonic_block<4, 2, DESC>(key, lane);
    key = bitonic_block<8, 4, DESC>(key, lane);
    key = bitonic_block<16, 8, DESC>(key, lane);
    key = bitonic_block<32, 16, DESC>(key, lane);
    key = bitonic_block<64, 32, DESC>(key, lane);
    return key;
}
__device__ __forceinline__ uint64_t wave_sort_desc(uint64_t key, int lane) { return wave_sort<true>(key, lane); }
// merge a descending-sorted list with an ASCENDING-sorted candidate vector, keep the best 64 (descending)
__device__ __forceinline__ uint64_t wave_merge_top64_asc(uint64_t a_desc, uint64_t b_asc, int lane) {
    uint64_t key = a_desc > b_asc ? a_desc : b_asc;          // bitonic sequence holding the 64 largest
    return bitonic_block<64, 32, true>(key, lane);
}
// convenience: both inputs descending (one crossbar reversal)
__device__ __forceinline__ uint64_t wave_merge_top64(uint64_t a, uint64_t b_sorted_desc, int lane) {
    return wave_merge_top64_asc(a, shfl_u64(b_sorted_desc, 63 - lane), lane);
}

// order-insensitive 64-lane sum on the DPP / permlane network (no LDS crossbar); every lane gets the total
__device__ __forceinline__ float wave_sum_dpp(float v, int lane) {
    v += __uint_as_float(xor_shfl<32>(__float_as_uint(v), lane));
    v += __uint_as_float(xor_shfl<16>(__float_as_uint(v), lane));
    v += __uint_as_float(xor_shfl<8>(__float_as_uint(v), lane));
    v += __uint_as_float(xor_shfl<4>(__float_as_uint(v), lane));
    v += __uint_as_float(xor_shfl<2>(__float_as_uint(v), lane));
    v += __uint_as_float(xor_shfl<1>(__float_as_uint(v), lane));
    return v;
}

__device__ __forceinline__ float wave_sum_butterfly(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = __fadd_rn(v, __shfl_xor(v, off, 64));
    return v;
}

}  // namespace dgg
