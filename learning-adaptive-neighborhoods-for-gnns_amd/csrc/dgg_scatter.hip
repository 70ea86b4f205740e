// dgg_scatter.hip -- column-side ("transposed") accumulations of the backward pass WITHOUT global float atomics.
//
// Two backward terms land on the NEIGHBOUR of an ELL entry (i, r) -> j = idx[i][r] rather than on its owner row:
//   score backward      dxp_j -= dd_ir (xp_i - xp_j)          autograd of ||xp_u - xp_v||, reference dgm.py:1613-1623
//   normalisation       da_j  += dA_ir w_ir a_i               autograd of D^-1/2 A D^-1/2, reference model.py:1215-1218
// Scattering them with global fp32 atomics runs at the atomic rate (~1.3 TB/s for 256-B rows, 17x less for scalars;
// MI355X_MICROARCH.md "Global float atomics"): 0.9 + 0.2 ms at N = 100k.  Instead the active entries are PARTITIONED
// once per forward into destination order (a CSC view of the ELL block): two counting passes with LDS histograms over
// buckets of 128 consecutive nodes (one global integer atomic per workgroup and bucket), then one workgroup per bucket
// orders its records by node.  The column-side kernels walk the sorted records in fixed chunks (balanced whatever the
// in-degree skew), reduce each run of equal destinations in registers and flush once per run: ~18x fewer float atomics
// than the entry-wise scatter (N + nnz/64 runs instead of nnz), 16-byte gathers of the source rows.
#include "dgg_common.h"
#include "dgg_api_internal.h"

using namespace dgg;

namespace {

constexpr int BS = 128;            // destination nodes per bucket
constexpr int PR = 256;            // rows per partition workgroup (4 wavefronts x 64 rows)

struct PartHdr {                   // workspace: [bstart NB+1][cursor NB][tmp rows*K int2][slot rows*K int][recs rows*K int2][cstart]
    int *bstart, *cursor;
    int2 *tmp;                     // bucket-ordered records (src = row*64 + r, dst = j)
    int *slot;                     // slot[row*K + r] = position of the entry in recs, -1 if inactive: lets the row-side
    int2 *recs;                    // kernels write their per-entry coefficient in RECORD order (coalesced reads later)
    int *cstart;                   // recs: the records ordered by destination node; cstart[c]: first record in [64 c, 64 c + 48] that
};                                 // starts a RUN of equal destination, else 64 c (rows*K/64 + 2 entries): chunks that split only hubs' runs
__host__ __device__ inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }
inline PartHdr part_layout(void *ws, int64_t nb, int64_t nrec) {
    char *w = reinterpret_cast<char *>(ws);
    PartHdr p;
    p.bstart = reinterpret_cast<int *>(w);
    p.cursor = reinterpret_cast<int *>(w + align256((size_t)(nb + 1) * 4));
    p.tmp = reinterpret_cast<int2 *>(w + align256((size_t)(nb + 1) * 4) + align256((size_t)nb * 4));
    p.slot = reinterpret_cast<int *>(reinterpret_cast<char *>(p.tmp) + align256((size_t)nrec * sizeof(int2)));
    p.recs = reinterpret_cast<int2 *>(reinterpret_cast<char *>(p.slot) + align256((size_t)nrec * sizeof(int)));
    p.cstart = reinterpret_cast<int *>(reinterpret_cast<char *>(p.recs) + align256((size_t)nrec * sizeof(int2)));
    return p;
}

// pass 1 (FILL = false): per-bucket totals.  pass 2 (FILL = true): bucket-sorted records (src = row*64 + r, dst = j)
template <bool FILL>
__global__ __launch_bounds__(256) void part_pass(const int32_t *__restrict__ idx, const float *__restrict__ w, int64_t rows,
                                                 int K, int nb, int *__restrict__ gcount, int2 *__restrict__ recs,
                                                 int *__restrict__ slotmap) {
    extern __shared__ int lds[];                                 // hist[nb] (+ base[nb] when filling)
    int *hist = lds, *base = lds + nb;
    const int tid = threadIdx.x, lane = tid & 63, wave = dgg::wave_id();
    for (int b = tid; b < nb; b += 256) hist[b] = 0;
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * PR + wave * 64;
    constexpr int UQ = 8;                                        // rows in flight per wavefront (loads before LDS atomics)
    // A wavefront whose 64 rows all exist (every one but the last) loads WITHOUT predicates: a predicated load is compiled
    // into a branch with its own s_waitcnt vmcnt(0), i.e. one load in flight at a time instead of 2*UQ.
    const bool full = r0 + 64 <= rows && K == 64;                // wave-uniform
    // entry code: destination j >= 0 (active), -1 (inactive entry), -2 (no entry)
    auto load_rows = [&](int q0, int32_t(&jj)[UQ]) {
        if (full) {
#pragma unroll
            for (int u = 0; u < UQ; u++) {
                const int64_t e = (r0 + q0 + u) * 64 + lane;
                const int32_t j = idx[e];
                const float wv = w[e];
                jj[u] = (j >= 0 && wv != 0.0f) ? j : -1;
            }
        } else {
#pragma unroll
            for (int u = 0; u < UQ; u++) {
                const int64_t i = r0 + q0 + u;
                jj[u] = -2;
                if (i < rows && lane < K) {
                    const int32_t j = idx[i * K + lane];
                    jj[u] = (j >= 0 && w[i * K + lane] != 0.0f) ? j : -1;
                }
            }
        }
    };
    for (int q0 = 0; q0 < 64; q0 += UQ) {
        int32_t jj[UQ];
        load_rows(q0, jj);
#pragma unroll
        for (int u = 0; u < UQ; u++)
            if (jj[u] >= 0) atomicAdd(&hist[jj[u] / BS], 1);
    }
    __syncthreads();
    for (int b = tid; b < nb; b += 256) {
        const int c = hist[b];
        if (FILL) { base[b] = c ? atomicAdd(&gcount[b], c) : 0; hist[b] = 0; }
        else if (c) atomicAdd(&gcount[b], c);
    }
    if (!FILL) return;
    __syncthreads();
    for (int q0 = 0; q0 < 64; q0 += UQ) {
        int32_t jj[UQ];
        load_rows(q0, jj);
#pragma unroll
        for (int u = 0; u < UQ; u++) {
            const int64_t i = r0 + q0 + u;
            if (jj[u] >= 0) {
                const int b = jj[u] / BS;
                const int slot = base[b] + atomicAdd(&hist[b], 1);
                recs[slot] = make_int2((int)(i * 64 + lane), jj[u]);
            } else if (jj[u] == -1) {
                slotmap[i * K + lane] = -1;                      // active entries: written by part_sort
            }
        }
    }
}

// pass 3: order the records of one bucket by destination node (counting sort on LDS counters); records the final
// position of every entry in slotmap
__global__ __launch_bounds__(1024) void part_sort(const int *__restrict__ bstart, const int2 *__restrict__ tmp, int K,
                                                 int2 *__restrict__ recs, int *__restrict__ slotmap) {
    __shared__ int cnt[BS], base[BS];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int e0 = bstart[b], e1 = bstart[b + 1];
    if (tid < BS) cnt[tid] = 0;
    __syncthreads();
    for (int e = e0 + tid; e < e1; e += 1024) atomicAdd(&cnt[tmp[e].y - b * BS], 1);
    __syncthreads();
    if (tid < 64) {                                              // exclusive scan of 128 counters by one wavefront
        const int c0 = cnt[2 * tid], c1 = cnt[2 * tid + 1];
        int incl = c0 + c1;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off, 64);
            if (tid >= off) incl += v;
        }
        base[2 * tid] = incl - c0 - c1;
        base[2 * tid + 1] = incl - c1;
    }
    __syncthreads();
    if (tid < BS) cnt[tid] = 0;
    __syncthreads();
    for (int e = e0 + tid; e < e1; e += 1024) {
        const int2 rec = tmp[e];
        const int jl = rec.y - b * BS;
        const int pos = e0 + base[jl] + atomicAdd(&cnt[jl], 1);
        recs[pos] = rec;
        slotmap[(int64_t)(rec.x >> 6) * K + (rec.x & 63)] = pos;
    }
}

// run-aligned chunk starts (see PartHdr::cstart): one thread per chunk.  A run that goes on for more than 48 records past the
// nominal start (a hub's) is SPLIT there instead: the consumer adds the pieces of such a run atomically.
__global__ void part_chunk_starts(const int *__restrict__ bstart, int nb, const int2 *__restrict__ recs, int nchunks, int chunk,
                                  int *__restrict__ cstart) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > nchunks) return;
    const int nnz = bstart[nb];
    const int64_t x = (int64_t)c * chunk;
    int e = x < nnz ? (int)x : nnz;
    if (e > 0) {
        const int lim = e + 48 < nnz ? e + 48 : nnz;
        while (e < lim && recs[e].y == recs[e - 1].y) e++;
        if (e < nnz && recs[e].y == recs[e - 1].y) e = (int)x;
    }
    cstart[c] = e;
}

// exclusive scan of the bucket totals (single workgroup): bstart[0..nb], cursor[b] = bstart[b]
__global__ __launch_bounds__(1024) void part_scan(int *__restrict__ bstart, int *__restrict__ cursor, int nb) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (nb + 1023) / 1024;
    const int lo = tid * per, hi = lo + per < nb ? lo + per : nb;
    int s = 0;
    for (int b = lo; b < hi; b++) s += cursor[b];                // cursor holds the totals of pass 1
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;                                     // exclusive prefix of this thread's segment
    for (int b = lo; b < hi; b++) {
        const int c = cursor[b];
        bstart[b] = run;
        cursor[b] = run;
        run += c;
    }
    if (tid == 1023) bstart[nb] = part[1023];
}

// ---- score backward, row side: coefficient dd_ir for every entry, dxp_i (own row: plain store) -----------------------
// H/4 lanes per neighbour (16-byte loads), 256/H neighbours per wave-instruction
// Arguments of the ramp + normalisation backward (dgg_softk_bwd) when it is evaluated inside edge_bwd_rows (FUSE): lane r of a
// row's wavefront owns entry r in both kernels, so d loss / d score is formed in registers instead of a [rows,K] round trip.
struct SoftkArgs {
    const float *k, *rs, *dA, *da;
    int mode, normalized;
    float *dval_out, *dk;                                        // dval_out nullable (diagnostics)
    // ahat_rows != NULL: `da` holds only the NEIGHBOUR-side sums (conv_bwd_cols); the row side
    // da_i += sum_r dA_ir w_ir a_j = sqrt(rs_i) sum_r dA_ir ahat_ir is added here, in registers
    const float *ahat_rows;
    // payload partition: no slot map, no coefficient hand-over; instead (a_i, d loss / d rs_i, k_i, 0) per row for edge_bwd_node
    float4 *rowinfo;
    // payload partition, dA == NULL: the row-major dA was never written (conv_bwd_node's 4.1 M scattered stores); entry (i, r) reads
    // dA_rec[recpos[i*64 + r]] instead -- the map is written by pp_sort, which runs beside the forward aggregation
    const int *recpos;
    const float *dA_rec;
};
template <int H, bool FUSE, bool PAY = false>
__global__ __launch_bounds__(256) void edge_bwd_rows(const float *__restrict__ xp, int64_t rows, const int32_t *__restrict__ idx,
                                                     const float *__restrict__ val, const float *__restrict__ dval, int K,
                                                     int64_t row0, float t, int perturb, const int *__restrict__ slotmap,
                                                     float *__restrict__ coef, float *__restrict__ dxp, SoftkArgs sk) {
    constexpr int LPR = H / 4;                                   // lanes per neighbour
    constexpr int NPI = 64 / LPR;                                // neighbours per wave-instruction
    const int lane = threadIdx.x & 63, c4 = lane % LPR, slot = lane / LPR;
    const int64_t i = (int64_t)blockIdx.x * 4 + dgg::wave_id();
    if (i >= rows) return;
    const int64_t gi = row0 + i;
    const int32_t jl = lane < K ? idx[i * K + lane] : -1;
    const float vl = lane < K ? val[i * K + lane] : 0.0f;
    float gl;
    if (FUSE) {                                                  // same arithmetic as softk_bwd_kernel (dgg_ell.hip), modes 0 / 1
        const int lc = lane < K ? lane : K - 1;
        const bool viamap = PAY && sk.recpos != nullptr;         // (kernel-uniform)
        float dw = viamap ? 0.0f : sk.dA[i * K + lc];
        const int rpos = viamap ? sk.recpos[i * 64 + lc] : 0;
        const bool live = lane < K && jl >= 0;
        if (sk.normalized) {
            const float rsi = sk.rs[gi];
            const float ai = __fdiv_rn(1.0f, c_sqrt(rsi)), aj = __fdiv_rn(1.0f, c_sqrt(sk.rs[jl >= 0 ? jl : gi]));
            float dai = sk.da[gi];
            if (sk.ahat_rows) {
                const float ah = lane < K ? sk.ahat_rows[i * K + lane] : 0.0f;
                // entries outside the partition (ahat == 0: empty slot or saturated ramp) were never written by conv_bwd_node:
                // with the payload partition dA is NOT zero-filled by the caller, so their dA is masked here
                if (viamap) dw = sk.dA_rec[ah != 0.0f ? rpos : 0];   // (unconditional gather; entries outside the partition have no map entry)
                if (PAY && ah == 0.0f) dw = 0.0f;
                float rp = dw * ah;
                rp = wave_sum_dpp(rp, lane);
                dai += rp * sqrtf(rsi);
            }
            const float drs = -0.5f * dai * ai / rsi;
            dw = dw * ai * aj + drs;
            if (PAY && lane == 0) sk.rowinfo[i] = make_float4(ai, drs, sk.k[i], 0.0f);
        } else if (PAY && lane == 0) {
            sk.rowinfo[i] = make_float4(1.0f, 0.0f, sk.k[i], 0.0f);
        }
        const float th = c_tanh((float)lane - sk.k[i]);
        const float f = 1.0f - 0.5f * (1.0f + th);
        const float dfdk = 0.5f * (1.0f - th * th);
        gl = (live && sk.mode == 0) ? dw * f : 0.0f;
        float skp = live ? (sk.mode == 0 ? dw * vl * dfdk : dw * dfdk) : 0.0f;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) skp += __shfl_xor(skp, off, 64);
        if (lane == 0) sk.dk[i] = skp;
        if (sk.dval_out && lane < K) sk.dval_out[i * K + lane] = gl;
    } else {
        gl = lane < K ? dval[i * K + lane] : 0.0f;
    }
    const float4 xi = *reinterpret_cast<const float4 *>(xp + gi * H + 4 * c4);
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float mycoef = 0.0f;
    // Per iteration: NBT batches of NPI neighbours, all gathers issued UNCONDITIONALLY (inactive slots re-read the own row and
    // are masked arithmetically) before any is consumed: a predicated gather is a branch + s_waitcnt vmcnt(0), i.e. one gather
    // in flight per wavefront.  Three passes over the batch registers: (1) squared distances, each handed to the lane that OWNS
    // the entry (lane r = entry r: it already holds the entry's cotangent and score); (2) the scalar chain sqrt / exp / two
    // divisions ONCE per entry -- evaluated per batch it ran on all H/4 lanes of a neighbour, 16 times per row, and made this
    // kernel VALU-bound (435 VALU instructions per 8 entries, 200 us); (3) the coefficient goes back to the neighbour's lanes.
    // Same operations per entry and the same accumulation order: identical bits.
    constexpr int NBT = (32 / NPI) < 1 ? 1 : 32 / NPI;           // 32 entries per iteration (two iterations for K = 64)
    const bool myact = lane < K && jl >= 0 && gl != 0.0f;
    for (int r0 = 0; r0 < K; r0 += NBT * NPI) {
        const bool mine = lane >= r0 && lane < r0 + NBT * NPI;
        if (__ballot(mine && myact) == 0ull) continue;           // wave-uniform: no active entry in this group
        float4 xj[NBT];
#pragma unroll
        for (int b = 0; b < NBT; b++) {
            const int r = r0 + b * NPI + slot;
            const int rr = r < 64 ? r : 63;
            const int32_t jb = __shfl(jl, rr, 64);
            const bool ab = __shfl((int)myact, rr, 64) != 0 && r < K;
            xj[b] = *reinterpret_cast<const float4 *>(xp + (ab ? (int64_t)jb : gi) * H + 4 * c4);
        }
        float myd2 = 0.0f;
#pragma unroll
        for (int b = 0; b < NBT; b++) {
            const float4 d = make_float4(xi.x - xj[b].x, xi.y - xj[b].y, xi.z - xj[b].z, xi.w - xj[b].w);   // 0 when inactive
            float d2 = d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
            if (LPR > 16) d2 += __uint_as_float(xor_shfl<16>(__float_as_uint(d2), lane));
            if (LPR > 8) d2 += __uint_as_float(xor_shfl<8>(__float_as_uint(d2), lane));
            if (LPR > 4) d2 += __uint_as_float(xor_shfl<4>(__float_as_uint(d2), lane));
            if (LPR > 2) d2 += __uint_as_float(xor_shfl<2>(__float_as_uint(d2), lane));
            d2 += __uint_as_float(xor_shfl<1>(__float_as_uint(d2), lane));
            // entry r0 + b*NPI + s was reduced by slot s: its owner lane fetches it from that slot's first lane
            const float tq = __shfl(d2, (lane % NPI) * LPR, 64);
            if ((lane - r0) / NPI == b && mine) myd2 = tq;
        }
        float mydd = 0.0f;
        if (mine && myact && myd2 != 0.0f) {                     // vector_norm backward at 0 is 0 (self loop)
            const float dist = sqrtf(myd2);
            const float p = c_exp(t * dist);
            const float dp = perturb ? gl * vl / (p + 1e-8f) : gl;
            mydd = dp * t * p / dist;
        }
        if (!PAY && mine) mycoef = mydd;                         // lane r owns entry r
#pragma unroll
        for (int b = 0; b < NBT; b++) {
            const int r = r0 + b * NPI + slot;
            const float dd = __shfl(mydd, r < 64 ? r : 63, 64);  // (outside any select: see conv_bwd_node)
            const float4 d = make_float4(xi.x - xj[b].x, xi.y - xj[b].y, xi.z - xj[b].z, xi.w - xj[b].w);
            acc.x += dd * d.x; acc.y += dd * d.y; acc.z += dd * d.z; acc.w += dd * d.w;
        }
    }
    // sum the NPI neighbour slots (lanes with equal c4)
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
        acc.x += __shfl_xor(acc.x, off, 64); acc.y += __shfl_xor(acc.y, off, 64);
        acc.z += __shfl_xor(acc.z, off, 64); acc.w += __shfl_xor(acc.w, off, 64);
    }
    if (slot == 0) *reinterpret_cast<float4 *>(dxp + gi * H + 4 * c4) = acc;
    if (!PAY && lane < K) {
        const int sl = slotmap[i * K + lane];
        if (sl >= 0) coef[sl] = mycoef;                          // record order
    }
}

// ---- the row kernel on CHUNKED rows (rows wider than 64 ranks; dgg_chunk_layout) ---------------------------------------------------
// edge_bwd_rows<H, FUSE = true, PAY = true> for a node whose row is the chunks [cptr[i], cptr[i+1]) of the [chunks,64] arrays: one
// wavefront per NODE walks its chunks.  The normalisation backward needs the row's complete sum_r dA_ir ahat_ir before any entry's
// cotangent can be formed, so the normalised form makes a first pass over dA / ahat of all chunks (two streamed arrays), then the main
// pass per chunk (ramp backward in registers, gathers as above).  rowinfo is written per CHUNK as (a_i, d loss / d rs_i, k_i - 64 m,
// 0): the per-destination kernel forms rank - k from the entry's index inside its chunk (64 m and k_i - 64 m are exact in fp32, so
// (float)lane - (k_i - 64 m) and (float)(64 m + lane) - k_i are the same rounding of the same real number).
template <int H>
__global__ __launch_bounds__(256) void edge_bwd_rows_chunked(const float *__restrict__ xp, int64_t rows, const int32_t *__restrict__ cptr,
                                                             const int32_t *__restrict__ idx, const float *__restrict__ val, int64_t row0,
                                                             float t, int perturb, float *__restrict__ dxp, SoftkArgs sk) {
    constexpr int LPR = H / 4;
    constexpr int NPI = 64 / LPR;
    constexpr int NBT = (32 / NPI) < 1 ? 1 : 32 / NPI;
    const int lane = threadIdx.x & 63, c4 = lane % LPR, slot = lane / LPR;
    const int64_t i = (int64_t)blockIdx.x * 4 + dgg::wave_id();
    if (i >= rows) return;
    const int64_t gi = row0 + i;
    const int cb = __builtin_amdgcn_readfirstlane(cptr[i]), ce = __builtin_amdgcn_readfirstlane(cptr[i + 1]);
    const float ki = sk.k[i];
    float ai = 1.0f, drs = 0.0f;
    if (sk.normalized) {
        const float rsi = sk.rs[gi];
        ai = __fdiv_rn(1.0f, c_sqrt(rsi));
        float dai = sk.da[gi];
        if (sk.ahat_rows) {
            float rp = 0.0f;
            for (int c = cb; c < ce; c++) {
                const float ah = sk.ahat_rows[(int64_t)c * 64 + lane];
                float dw = sk.dA[(int64_t)c * 64 + lane];
                if (ah == 0.0f) dw = 0.0f;                       // (entries outside the partition were never written)
                rp += dw * ah;
            }
            rp = wave_sum_dpp(rp, lane);
            dai += rp * sqrtf(rsi);
        }
        drs = -0.5f * dai * ai / rsi;
    }
    const float4 xi = *reinterpret_cast<const float4 *>(xp + gi * H + 4 * c4);
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float skp = 0.0f;
    for (int c = cb; c < ce; c++) {
        const int64_t e = (int64_t)c * 64 + lane;
        const int32_t jl = idx[e];
        const float vl = val[e];
        const bool live = jl >= 0;
        float dw = sk.dA[e];
        const float kshift = ki - (float)(64 * (c - cb));
        if (sk.normalized) {
            const float aj = __fdiv_rn(1.0f, c_sqrt(sk.rs[live ? jl : gi]));
            if (sk.ahat_rows && sk.ahat_rows[e] == 0.0f) dw = 0.0f;
            dw = dw * ai * aj + drs;
        }
        if (lane == 0) sk.rowinfo[c] = make_float4(ai, drs, kshift, 0.0f);
        const float th = c_tanh((float)lane - kshift);
        const float f = 1.0f - 0.5f * (1.0f + th);
        const float dfdk = 0.5f * (1.0f - th * th);
        const float gl = (live && sk.mode == 0) ? dw * f : 0.0f;
        skp += live ? (sk.mode == 0 ? dw * vl * dfdk : dw * dfdk) : 0.0f;
        const bool myact = live && gl != 0.0f;
        if (__ballot(myact) == 0ull) continue;                   // (wave-uniform: a chunk of saturated ranks or beyond the ramp)
        for (int r0 = 0; r0 < 64; r0 += NBT * NPI) {
            const bool mine = lane >= r0 && lane < r0 + NBT * NPI;
            if (__ballot(mine && myact) == 0ull) continue;
            float4 xj[NBT];
#pragma unroll
            for (int b = 0; b < NBT; b++) {
                const int r = r0 + b * NPI + slot;
                const int rr = r < 64 ? r : 63;
                const int32_t jb = __shfl(jl, rr, 64);
                const bool ab = __shfl((int)myact, rr, 64) != 0 && r < 64;
                xj[b] = *reinterpret_cast<const float4 *>(xp + (ab ? (int64_t)jb : gi) * H + 4 * c4);
            }
            float myd2 = 0.0f;
#pragma unroll
            for (int b = 0; b < NBT; b++) {
                const float4 d = make_float4(xi.x - xj[b].x, xi.y - xj[b].y, xi.z - xj[b].z, xi.w - xj[b].w);
                float d2 = d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
                if (LPR > 16) d2 += __uint_as_float(xor_shfl<16>(__float_as_uint(d2), lane));
                if (LPR > 8) d2 += __uint_as_float(xor_shfl<8>(__float_as_uint(d2), lane));
                if (LPR > 4) d2 += __uint_as_float(xor_shfl<4>(__float_as_uint(d2), lane));
                if (LPR > 2) d2 += __uint_as_float(xor_shfl<2>(__float_as_uint(d2), lane));
                d2 += __uint_as_float(xor_shfl<1>(__float_as_uint(d2), lane));
                const float tq = __shfl(d2, (lane % NPI) * LPR, 64);
                if ((lane - r0) / NPI == b && mine) myd2 = tq;
            }
            float mydd = 0.0f;
            if (mine && myact && myd2 != 0.0f) {
                const float dist = sqrtf(myd2);
                const float p = c_exp(t * dist);
                const float dp = perturb ? gl * vl / (p + 1e-8f) : gl;
                mydd = dp * t * p / dist;
            }
#pragma unroll
            for (int b = 0; b < NBT; b++) {
                const int r = r0 + b * NPI + slot;
                const float dd = __shfl(mydd, r < 64 ? r : 63, 64);
                const float4 d = make_float4(xi.x - xj[b].x, xi.y - xj[b].y, xi.z - xj[b].z, xi.w - xj[b].w);
                acc.x += dd * d.x; acc.y += dd * d.y; acc.z += dd * d.z; acc.w += dd * d.w;
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) skp += __shfl_xor(skp, off, 64);
    if (lane == 0) sk.dk[i] = skp;
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
        acc.x += __shfl_xor(acc.x, off, 64); acc.y += __shfl_xor(acc.y, off, 64);
        acc.z += __shfl_xor(acc.z, off, 64); acc.w += __shfl_xor(acc.w, off, 64);
    }
    if (slot == 0) *reinterpret_cast<float4 *>(dxp + gi * H + 4 * c4) = acc;
}

// ---- score backward, column side: chunks of CH destination-ordered records per group of H/4 lanes --------------------
// acc_j = sum_e dd_e (xp_j - xp_{i_e}) over the run of records with destination j: -sum_e dd_e xp_{i_e} accumulated record by
// record, (sum_e dd_e) xp_j added at the flush (one extra row gather per run).
constexpr int CH = 64;
template <int H>
__global__ __launch_bounds__(256) void edge_bwd_cols(const float *__restrict__ xp, const int *__restrict__ bstart, int nb,
                                                     const int2 *__restrict__ recs, const float *__restrict__ coef,
                                                     int64_t row0, float *__restrict__ dxp) {
    constexpr int LPR = H / 4;
    const int lane = threadIdx.x & 63, c4 = lane % LPR, gbase = lane - c4;
    const int64_t gid = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LPR;
    const int nnz = bstart[nb];
    const int64_t cbeg = gid * CH;
    if (cbeg >= nnz) return;
    const int cend = cbeg + CH < nnz ? (int)cbeg + CH : nnz;
    int cur = -1;
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float sacc = 0.0f;
    // Records are globally ordered by destination, so only the FIRST and the LAST run of a chunk can continue in a neighbouring
    // chunk (another lane group): those are flushed with float atomics, every other run has this group as its only writer and
    // is flushed with one 16-byte read-modify-write per lane.  (With G ranks a rank's records spread over G times more
    // destinations -- ~4 records per run instead of ~30 at G = 8 -- and atomics for every run would dominate the kernel.)
    const int shared_lo = cbeg > 0 ? recs[cbeg - 1].y : -1, shared_hi = cend < nnz ? recs[cend].y : -1;
    auto flush = [&]() {
        if (cur >= 0) {
            // the run's share of  dxp_j += (sum_e dd_e) xp_j  is linear in the partial sum, so every group adds its own part here
            const float4 xj = *reinterpret_cast<const float4 *>(xp + (int64_t)cur * H + 4 * c4);
            acc.x = fmaf(sacc, xj.x, acc.x); acc.y = fmaf(sacc, xj.y, acc.y);
            acc.z = fmaf(sacc, xj.z, acc.z); acc.w = fmaf(sacc, xj.w, acc.w);
            float *o = dxp + (int64_t)cur * H + 4 * c4;
            if (cur == shared_lo || cur == shared_hi) {
                atomicAdd(o + 0, acc.x); atomicAdd(o + 1, acc.y); atomicAdd(o + 2, acc.z); atomicAdd(o + 3, acc.w);
            } else {
                float4 v = *reinterpret_cast<float4 *>(o);
                v.x += acc.x; v.y += acc.y; v.z += acc.z; v.w += acc.w;
                *reinterpret_cast<float4 *>(o) = v;
            }
        }
    };
    for (int eb = (int)cbeg; eb < cend; eb += LPR) {             // LPR records per batch, one per lane of the group
        const int e = eb + c4;
        const int2 myrec = e < cend ? recs[e] : make_int2(0, -1);
        const float mycf = e < cend ? coef[e] : 0.0f;
#pragma unroll
        for (int u0 = 0; u0 < LPR; u0 += 4) {
            int src[4], dst[4];
            float cf[4];
            float4 xi[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                src[u] = __shfl(myrec.x, gbase + u0 + u, 64);
                dst[u] = __shfl(myrec.y, gbase + u0 + u, 64);
                cf[u] = __shfl(mycf, gbase + u0 + u, 64);
                // unconditional (padding records point at row 0 and carry coefficient 0): four loads back to back
                xi[u] = *reinterpret_cast<const float4 *>(xp + (row0 + (src[u] >> 6)) * H + 4 * c4);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (dst[u] < 0) continue;
                if (dst[u] != cur) {
                    flush();
                    cur = dst[u];
                    acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    sacc = 0.0f;
                }
                acc.x -= cf[u] * xi[u].x; acc.y -= cf[u] * xi[u].y; acc.z -= cf[u] * xi[u].z; acc.w -= cf[u] * xi[u].w;
                sacc += cf[u];
            }
        }
    }
    flush();
}
// ---- transposed SpMM through the partition: dX_j += sum_{(i,r) -> j} a_ir dY_i ------------------------------------------
// (autograd of torch.mm(adj, x) w.r.t. x, model.py:594; needed whenever the conv input is itself a learned activation: second
// GCNConv of GCN_DGG, every GCNII layer).  Same walk as edge_bwd_cols: fixed chunks of destination-ordered records per
// 16-lane group, runs of equal destination reduced in registers, one flush per run; grid.y = blocks of 64 features.
// B16: dY is a bf16 copy [rows, F] of the cotangent (the fused GCNII stack: half the gathered bytes), accumulation in fp32 as before
template <bool B16 = false>
__global__ __launch_bounds__(256) void spmm_t_cols(const void *__restrict__ dYv, int F, const int *__restrict__ cstart, int nchunks,
                                                   const int2 *__restrict__ recs, const float *__restrict__ a, int K,
                                                   float *__restrict__ dX) {
    const float *__restrict__ dY = static_cast<const float *>(dYv);
    const uint16_t *__restrict__ dYb = static_cast<const uint16_t *>(dYv);
    constexpr int LPR = 16;
    const int lane = threadIdx.x & 63, c4 = lane % LPR;
    const int64_t gid = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LPR;
    const int f0 = blockIdx.y * 64 + 4 * c4;
    if (gid >= nchunks) return;
    // run-aligned chunk (PartHdr::cstart): but for the pieces of a hub's run, every destination row is written by exactly one lane
    // group -- a plain add.  (With fixed chunks of 64 records the two runs at a chunk's ends were added atomically: two thirds of all
    // runs at PPI's ~29 records per destination, and those atomics, not the gathers, were the kernel's time: 56 of 59 us.)
    const int cbeg = cstart[gid], cend = cstart[gid + 1];
    if (cbeg >= cend) return;
    const int nnz = cstart[nchunks];
    const int first = recs[cbeg].y, last = recs[cend - 1].y;
    const int shared_lo = (cbeg > 0 && recs[cbeg - 1].y == first) ? first : -1;
    const int shared_hi = (cend < nnz && recs[cend].y == last) ? last : -1;
    int cur = -1;
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    auto flush = [&]() {
        if (cur >= 0) {
            float *o = dX + (int64_t)cur * F + f0;
            if (cur == shared_lo || cur == shared_hi) {
                atomicAdd(o + 0, acc.x); atomicAdd(o + 1, acc.y); atomicAdd(o + 2, acc.z); atomicAdd(o + 3, acc.w);
            } else {
                float4 v = *reinterpret_cast<float4 *>(o);
                v.x += acc.x; v.y += acc.y; v.z += acc.z; v.w += acc.w;
                *reinterpret_cast<float4 *>(o) = v;
            }
        }
    };
    for (int eb = cbeg; eb < cend; eb += LPR) {
        const int e = eb + c4;
        const int2 myrec = e < cend ? recs[e] : make_int2(0, -1);
        // record source = row*64 + r: the coefficient lives at a[row*K + r]
        const float mycf = e < cend ? a[(int64_t)(myrec.x >> 6) * K + (myrec.x & 63)] : 0.0f;
        // all 16 rows of the batch are requested before the first is used
        int src[LPR], dst[LPR], cfb[LPR];                         // (row broadcasts: the shuffles of this loop were 3 ds_bpermute per record and slice)
        float4 g[LPR];
        dgg::row16_all(myrec.x, src);
        dgg::row16_all(myrec.y, dst);
        dgg::row16_all(__float_as_int(mycf), cfb);
#pragma unroll
        for (int u = 0; u < LPR; u++) {
            if constexpr (B16) {
                const uint2 h_ = *reinterpret_cast<const uint2 *>(dYb + (int64_t)(src[u] >> 6) * F + f0);   // unconditional
                g[u] = make_float4(__uint_as_float(h_.x << 16), __uint_as_float(h_.x & 0xffff0000u), __uint_as_float(h_.y << 16),
                                   __uint_as_float(h_.y & 0xffff0000u));
            } else {
                g[u] = *reinterpret_cast<const float4 *>(dY + (int64_t)(src[u] >> 6) * F + f0);          // unconditional
            }
        }
#pragma unroll
        for (int u = 0; u < LPR; u++) {
            if (dst[u] < 0) continue;
            if (dst[u] != cur) {
                flush();
                cur = dst[u];
                acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            const float cf = __int_as_float(cfb[u]);
            acc.x = fmaf(cf, g[u].x, acc.x); acc.y = fmaf(cf, g[u].y, acc.y);
            acc.z = fmaf(cf, g[u].z, acc.z); acc.w = fmaf(cf, g[u].w, acc.w);
        }
    }
    flush();
}

// ---- graph-conv backward on PROJECTED features through the partition -------------------------------------------------------
// Z = act(A H) with H = X W aggregated AFTER the projection ((A X) W = A (X W): reference model.py:594-598 aggregates the
// d-wide X and projects afterwards; here the F-wide H, F = out_features <= d, is gathered instead -- half the bytes at
// 128 -> 64, 22x fewer on Cora's 1433 -> 64).  For the cotangent G = dZ * act'(Z) and every destination-ordered record
// e = (i, r) -> j, ONE gathered row G_i serves all three column-walking terms of the backward:
//     dA_ir  = <G_i, H_j>                      (SDDMM: autograd wrt the adjacency values; H_j is the same line for a whole run)
//     dH_j  += ahat_ir G_i                     (transposed SpMM: autograd wrt the aggregated features)
//     da_j  += dA_ir w_ir a_i = sqrt(rs_j) sum_e dA_e ahat_e   (neighbour side of the normalize_adj backward, model.py:1215-1218)
// F/4 lanes per record (16-byte loads), 256/F records per wave-instruction, chunks of CH records per lane group as in
// edge_bwd_cols: runs of equal destination are reduced in registers, one flush per run, float atomics only for the two runs of
// a chunk that can continue in a neighbouring chunk.
template <int F>
__global__ __launch_bounds__(256) void conv_bwd_cols(const float *__restrict__ G, const float *__restrict__ Hm,
                                                     const float *__restrict__ a, int K, const int *__restrict__ bstart, int nb,
                                                     const int2 *__restrict__ recs, const float *__restrict__ rs,
                                                     float *__restrict__ dA, float *__restrict__ dH, float *__restrict__ da) {
    constexpr int LPR = F / 4;
    const int lane = threadIdx.x & 63, c4 = lane % LPR, gbase = lane - c4;
    const int64_t gid = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LPR;
    const int nnz = bstart[nb];
    const int64_t cbeg = gid * CH;
    if (cbeg >= nnz) return;
    const int cend = cbeg + CH < nnz ? (int)cbeg + CH : nnz;
    int cur = -1;
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float sda = 0.0f;
    const int shared_lo = cbeg > 0 ? recs[cbeg - 1].y : -1, shared_hi = cend < nnz ? recs[cend].y : -1;   // see edge_bwd_cols
    auto flush = [&]() {
        if (cur >= 0) {
            float *o = dH + (int64_t)cur * F + 4 * c4;
            const bool shared = cur == shared_lo || cur == shared_hi;
            if (shared) {
                atomicAdd(o + 0, acc.x); atomicAdd(o + 1, acc.y); atomicAdd(o + 2, acc.z); atomicAdd(o + 3, acc.w);
            } else {
                float4 v = *reinterpret_cast<float4 *>(o);
                v.x += acc.x; v.y += acc.y; v.z += acc.z; v.w += acc.w;
                *reinterpret_cast<float4 *>(o) = v;
            }
            if (da && c4 == 0) {
                const float v = sda * sqrtf(rs[cur]);
                if (shared) atomicAdd(da + cur, v);
                else da[cur] += v;
            }
        }
    };
    for (int eb = (int)cbeg; eb < cend; eb += LPR) {
        const int e = eb + c4;
        const int2 myrec = e < cend ? recs[e] : make_int2(0, -1);
        const float mycf = e < cend ? a[(int64_t)(myrec.x >> 6) * K + (myrec.x & 63)] : 0.0f;
#pragma unroll
        for (int u0 = 0; u0 < LPR; u0 += 4) {
            int src[4], dst[4];
            float cf[4];
            float4 g[4], hj[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                src[u] = __shfl(myrec.x, gbase + u0 + u, 64);
                dst[u] = __shfl(myrec.y, gbase + u0 + u, 64);
                cf[u] = __shfl(mycf, gbase + u0 + u, 64);
                // both unconditional (padding records: row 0 / node 0, coefficient 0): eight loads back to back; the H_j line is
                // the same for a whole run of records, i.e. an L1 hit after the run's first record
                g[u] = *reinterpret_cast<const float4 *>(G + (int64_t)(src[u] >> 6) * F + 4 * c4);
                hj[u] = *reinterpret_cast<const float4 *>(Hm + (int64_t)(dst[u] < 0 ? 0 : dst[u]) * F + 4 * c4);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                float dot = g[u].x * hj[u].x;
                dot = fmaf(g[u].y, hj[u].y, dot); dot = fmaf(g[u].z, hj[u].z, dot); dot = fmaf(g[u].w, hj[u].w, dot);
                if (LPR > 16) dot += __uint_as_float(xor_shfl<16>(__float_as_uint(dot), lane));
                if (LPR > 8) dot += __uint_as_float(xor_shfl<8>(__float_as_uint(dot), lane));
                if (LPR > 4) dot += __uint_as_float(xor_shfl<4>(__float_as_uint(dot), lane));
                dot += __uint_as_float(xor_shfl<2>(__float_as_uint(dot), lane));
                dot += __uint_as_float(xor_shfl<1>(__float_as_uint(dot), lane));
                if (dst[u] < 0) continue;
                if (dst[u] != cur) {
                    flush();
                    cur = dst[u];
                    acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    sda = 0.0f;
                }
                if (c4 == 0) dA[(int64_t)(src[u] >> 6) * K + (src[u] & 63)] = dot;
                acc.x = fmaf(cf[u], g[u].x, acc.x); acc.y = fmaf(cf[u], g[u].y, acc.y);
                acc.z = fmaf(cf[u], g[u].z, acc.z); acc.w = fmaf(cf[u], g[u].w, acc.w);
                sda = fmaf(dot, cf[u], sda);
            }
        }
    }
    flush();
}

// ---- one wavefront per DESTINATION node (CSC walk of the payload records) ------------------------------------------------
// (A chunked walk of the payload records -- fixed chunks of 64 records per lane group, runs of equal destination reduced in
// registers, as conv_bwd_cols / edge_bwd_cols do on the slot-map partition -- was built first: ~100 registers of run-tracking
// state per lane, 4 wavefronts per SIMD, 258 + 190 us; removed.)  With the CSC pointer of the payload partition a wavefront owns ALL records of one
// destination j: H_j / xp_j are loaded once, there is no run logic, the sums are plain stores (no atomics, no zero fill for
// dH / da), and the kernel has the shape of the forward SpMM: F/4 lanes per record, 256/F records per wave-instruction,
// NBT batches in flight.  (In-degree skew: a node with thousands of incoming edges is walked by one wavefront.)
// EXT: the adjacency is read by MORE consumers than this aggregation (GCN_DGG feeds the same normalised adjacency to its second
// layer, model.py:1266-1290): their cotangent dA_ext [rows,K] (same slots as dA) is added to G_i . H_j per record, so that dA,
// dA_rec and da carry the total and the score backward needs no second pass.
// Nodes with very many incoming edges (a citation graph's hubs: 300 on the Pubmed shape, where the whole kernel is 20 000 wavefronts
// of ~5 records): `nmain` > 0 appends one WORKGROUP per node to the grid; the wavefront-per-node part then skips nodes with more than
// NODE_LONG records and the appended workgroup of such a node (the others leave at once) walks a quarter of the records per wavefront
// and meets in LDS.  Used for graphs of at most NODE_SPLIT_MAX nodes (the appended workgroups cost ~5 us per 100 000 nodes).
constexpr int NODE_LONG = 128;
constexpr int64_t NODE_SPLIT_MAX = 65536;

template <int F, bool EXT>
__device__ __forceinline__ void conv_bwd_walk(const float *__restrict__ G, int K, int pb, int pe, int p1, const int4 *__restrict__ recs,
                                              const float4 hj, const float aj, float *__restrict__ dA, float *__restrict__ dA_rec,
                                              const float *__restrict__ dA_ext, int lane, float4 &acc, float &sda, const int wide = 0) {
    // (wide: chunked rows -- a record's first word is chunk * 64 + entry and its second word the SOURCE NODE of the chunk, see pp_sort)
    constexpr int LPR = F / 4, NPI = 64 / LPR, NBT = 4, PER = NBT * NPI;          // PER records per iteration (<= 64)
    const int c4 = lane % LPR, slot = lane / LPR;
    // ROWSHARE (groups of 16 lanes = DPP rows, F = 64): record q = b * NPI + slot of an iteration is LOADED by lane 16 slot + b, a lane
    // of the group that works on it in batch b -- its fields then reach the group by DPP row_share (one vector instruction each) instead
    // of ds_bpermute through the LDS crossbar, and the reduced dot product is already in the lane that stores it (203 -> 199 us at
    // N = 100k; the same change in edge_bwd_walk measured nothing and was not kept)
    constexpr bool ROWSHARE = LPR == 16;
    for (int e0 = pb; e0 < pe; e0 += PER) {
        // lane l: record e0 + l (ROWSHARE: e0 + (l % 16) * NPI + l / 16; unconditional, clamped load; a lane past the range marks its copy invalid)
        const int myq = ROWSHARE ? c4 * NPI + slot : lane;
        const bool have = (ROWSHARE ? c4 < NBT : lane < PER) && e0 + myq < pe;
        int4 myrec = recs[have ? e0 + myq : p1 - 1];
        if (!have) myrec.y = -1;
        int src[NBT];
        float cf[NBT];
        float4 g[NBT];
        int bx[NBT], by[NBT], bz[NBT];
        if constexpr (ROWSHARE) { dgg::row16_first<NBT>(myrec.x, bx); dgg::row16_first<NBT>(myrec.y, by); dgg::row16_first<NBT>(myrec.z, bz); }
#pragma unroll
        for (int b = 0; b < NBT; b++) {
            const int q = b * NPI + slot;                        // record of this lane's group in batch b
            src[b] = ROWSHARE ? bx[b] : __shfl(myrec.x, q, 64);
            const int dst = ROWSHARE ? by[b] : __shfl(myrec.y, q, 64);
            // (every shuffle OUTSIDE the select: `c ? shfl : 0` would run the shuffle under a divergent mask, and a lane that is
            // masked off there hands 0 to the lanes that read from it)
            const float wa = __int_as_float(ROWSHARE ? bz[b] : __shfl(myrec.z, q, 64));
            cf[b] = dst >= 0 ? __fmul_rn(wa, aj) : 0.0f;
            if (dst < 0) src[b] = -1;
            g[b] = *reinterpret_cast<const float4 *>(G + (int64_t)(src[b] < 0 ? 0 : (wide ? dst : (src[b] >> 6))) * F + 4 * c4);   // unconditional
        }
        float ext[NBT];
        if constexpr (EXT) {
#pragma unroll
            for (int b = 0; b < NBT; b++) ext[b] = src[b] >= 0 ? dA_ext[(int64_t)(src[b] >> 6) * K + (src[b] & 63)] : 0.0f;
        }
        float mydot = 0.0f;
#pragma unroll
        for (int b = 0; b < NBT; b++) {
            float dot = g[b].x * hj.x;
            dot = fmaf(g[b].y, hj.y, dot); dot = fmaf(g[b].z, hj.z, dot); dot = fmaf(g[b].w, hj.w, dot);
            if (LPR > 16) dot += __uint_as_float(xor_shfl<16>(__float_as_uint(dot), lane));
            if (LPR > 8) dot += __uint_as_float(xor_shfl<8>(__float_as_uint(dot), lane));
            if (LPR > 4) dot += __uint_as_float(xor_shfl<4>(__float_as_uint(dot), lane));
            dot += __uint_as_float(xor_shfl<2>(__float_as_uint(dot), lane));
            dot += __uint_as_float(xor_shfl<1>(__float_as_uint(dot), lane));
            if constexpr (EXT) dot += ext[b];
            // (write-through store that does not stay in the XCD's L2: these 4.1 M scattered dwords are never touched again by this
            //  kernel, and left in L2 they push out the G rows the gathers hit: 202 -> 193 us; nontemporal: 196)
            if (dA && src[b] >= 0 && c4 == 0)                    // (dA == NULL: the row kernel reads dA_rec through the slot -> record map)
                __hip_atomic_store(&dA[(int64_t)(src[b] >> 6) * K + (src[b] & 63)], dot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // back to the lane that loaded the record: record order, one store per iteration
            if constexpr (ROWSHARE) {
                if (c4 == b) mydot = dot;                        // (every lane of the group holds the reduced value)
            } else {
                const float tq = __shfl(dot, (lane % NPI) * LPR, 64);
                if (lane / NPI == b) mydot = tq;
            }
            acc.x = fmaf(cf[b], g[b].x, acc.x); acc.y = fmaf(cf[b], g[b].y, acc.y);
            acc.z = fmaf(cf[b], g[b].z, acc.z); acc.w = fmaf(cf[b], g[b].w, acc.w);
            sda = fmaf(dot, cf[b], sda);
        }
        if (have) dA_rec[e0 + myq] = mydot;
    }
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
        acc.x += __shfl_xor(acc.x, off, 64); acc.y += __shfl_xor(acc.y, off, 64);
        acc.z += __shfl_xor(acc.z, off, 64); acc.w += __shfl_xor(acc.w, off, 64);
        sda += __shfl_xor(sda, off, 64);
    }
}

template <int F, bool EXT = false>
__global__ __launch_bounds__(256) void conv_bwd_node(const float *__restrict__ G, const float *__restrict__ Hm, int K, int64_t ncols,
                                                     const int *__restrict__ nodeptr, const int4 *__restrict__ recs,
                                                     const float *__restrict__ rs, float *__restrict__ dA, float *__restrict__ dA_rec,
                                                     float *__restrict__ dH, float *__restrict__ da, const float *__restrict__ dA_ext = nullptr,
                                                     unsigned nmain = 0, int wide = 0) {
    constexpr int LPR = F / 4, PER = 4 * (64 / LPR);
    __shared__ float4 part[4][LPR];
    __shared__ float parts[4];
    const int lane = threadIdx.x & 63, c4 = lane % LPR, slot = lane / LPR, wave = dgg::wave_id();
    const bool longmode = nmain != 0 && blockIdx.x >= nmain;      // (block-uniform)
    const int64_t j = longmode ? (int64_t)(blockIdx.x - nmain) : (int64_t)blockIdx.x * 4 + wave;
    if (j >= ncols) return;
    const int p0 = nodeptr[j], p1 = nodeptr[j + 1];
    const bool islong = nmain != 0 && p1 - p0 > NODE_LONG;
    if (islong != longmode) return;                               // (a long node belongs to its appended workgroup; block-uniform there)
    const float4 hj = *reinterpret_cast<const float4 *>(Hm + j * F + 4 * c4);
    const float rsj = rs[j];
    const float aj = __fdiv_rn(1.0f, c_sqrt(rsj));                // as normalize_fwd_kernel: wa * aj == ahat bit for bit
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float sda = 0.0f;
    if (!longmode) {
        conv_bwd_walk<F, EXT>(G, K, p0, p1, p1, recs, hj, aj, dA, dA_rec, dA_ext, lane, acc, sda, wide);
    } else {
        const int chunk = ((p1 - p0 + 4 * PER - 1) / (4 * PER)) * PER;
        const int pb = p0 + wave * chunk, pe = pb + chunk < p1 ? pb + chunk : p1;
        conv_bwd_walk<F, EXT>(G, K, pb < p1 ? pb : p1, pe, p1, recs, hj, aj, dA, dA_rec, dA_ext, lane, acc, sda, wide);
        if (slot == 0) part[wave][c4] = acc;
        if (lane == 0) parts[wave] = sda;
        __syncthreads();
        if (wave != 0) return;
        acc = part[0][c4];
        sda = parts[0];
#pragma unroll
        for (int w_ = 1; w_ < 4; w_++) {
            const float4 v = part[w_][c4];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            sda += parts[w_];
        }
    }
    if (slot == 0) *reinterpret_cast<float4 *>(dH + j * F + 4 * c4) = acc;
    if (da && lane == 0) da[j] = sda * sqrtf(rsj);
}

// score backward, column side, one wavefront per destination node.  For record e = (i, r) -> j the lane that loaded it forms
//     dw = dA_e a_i a_j + drs_i,  dval = dw * ramp(r - k_i)                 (softk_bwd_kernel, dgg_ell.hip; dgm.py:1410-1420)
// from the per-row scalars (a_i, drs_i, k_i) that edge_bwd_rows<PAY> left in rowinfo; the distance part
//     dd = dval [v / (p + 1e-8)] t p / dist,  p = exp(t dist)               (autograd of dgm.py:1213-1229, 1618-1623)
// needs the gathered row xp_i; acc_j = sum_e dd_e (xp_j - xp_i).  The lane that
// loaded a record forms d loss / d score for it (PER records in parallel), the distance part runs per lane group.
// dxp_j is the exclusive property of this wavefront: rows inside [row0, row0 + rows) already hold the row-side term written by
// edge_bwd_rows (read-modify-write), the others are written plainly (no zero fill needed).
template <int H>
__device__ __forceinline__ void edge_bwd_walk(const float *__restrict__ xp, int64_t j, int pb, int pe, int p1, const int4 *__restrict__ recs,
                                              const float *__restrict__ dA_rec, const float4 *__restrict__ rowinfo, const float4 xj, const float aj,
                                              int64_t row0, float t, int perturb, int lane, float4 &acc, const int wide = 0) {
    // (wide: chunked rows -- rowinfo is per CHUNK with the degree shifted by the chunk's first rank, k_i - 64 m, so that the entry's
    //  index inside its chunk still gives rank - k; the record's second word is the source node of the chunk, see pp_sort)
    constexpr int LPR = H / 4, NPI = 64 / LPR, NBT = (32 / NPI) < 1 ? 1 : 32 / NPI, PER = NBT * NPI;   // 32 records per iteration
    const int c4 = lane % LPR, slot = lane / LPR;
    for (int e0 = pb; e0 < pe; e0 += PER) {
        const bool have = lane < PER && e0 + lane < pe;
        const int ec = have ? e0 + lane : p1 - 1;                // unconditional, clamped loads
        int4 myrec = recs[ec];
        if (!have) myrec.y = -1;
        float mydval;
        {                                                        // d loss / d score of this lane's own record
            const float4 info = rowinfo[myrec.x >> 6];
            const float dw = dA_rec[ec] * info.x * aj + info.y;
            const float th = c_tanh((float)(myrec.x & 63) - info.z);
            mydval = have ? dw * (1.0f - 0.5f * (1.0f + th)) : 0.0f;
        }
        float4 xi[NBT];
#pragma unroll
        for (int b = 0; b < NBT; b++) {
            const int q = b * NPI + slot;
            const int src = __shfl(myrec.x, q, 64);             // every shuffle outside a select (see conv_bwd_node)
            const int dst = __shfl(myrec.y, q, 64);
            // unconditional gather; an inactive slot re-reads xp_j, whose distance to itself is 0 (dd = 0)
            xi[b] = *reinterpret_cast<const float4 *>(xp + (dst >= 0 ? row0 + (wide ? dst : (src >> 6)) : j) * H + 4 * c4);
        }
        // squared distances, each handed to the lane that loaded the record; the scalar chain (sqrt, exp, two divisions) then runs
        // ONCE per record instead of on all H/4 lanes of every batch (the kernel was VALU-bound: 515 VALU per 16 records)
        float myd2 = 0.0f;
#pragma unroll
        for (int b = 0; b < NBT; b++) {
            const float4 d = make_float4(xi[b].x - xj.x, xi[b].y - xj.y, xi[b].z - xj.z, xi[b].w - xj.w);
            float d2 = d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
            if (LPR > 16) d2 += __uint_as_float(xor_shfl<16>(__float_as_uint(d2), lane));
            if (LPR > 8) d2 += __uint_as_float(xor_shfl<8>(__float_as_uint(d2), lane));
            if (LPR > 4) d2 += __uint_as_float(xor_shfl<4>(__float_as_uint(d2), lane));
            d2 += __uint_as_float(xor_shfl<2>(__float_as_uint(d2), lane));
            d2 += __uint_as_float(xor_shfl<1>(__float_as_uint(d2), lane));
            const float tq = __shfl(d2, (lane % NPI) * LPR, 64);
            if (lane / NPI == b) myd2 = tq;
        }
        float mydd = 0.0f;
        if (mydval != 0.0f && myd2 != 0.0f) {                    // vector_norm backward at 0 is 0 (self loop)
            const float dist = sqrtf(myd2);
            const float p = c_exp(t * dist);
            const float dp = perturb ? mydval * __int_as_float(myrec.w) / (p + 1e-8f) : mydval;
            mydd = dp * t * p / dist;
        }
#pragma unroll
        for (int b = 0; b < NBT; b++) {
            const float dd = __shfl(mydd, b * NPI + slot, 64);
            const float4 d = make_float4(xi[b].x - xj.x, xi[b].y - xj.y, xi[b].z - xj.z, xi[b].w - xj.w);
            acc.x -= dd * d.x; acc.y -= dd * d.y; acc.z -= dd * d.z; acc.w -= dd * d.w;
        }
    }
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
        acc.x += __shfl_xor(acc.x, off, 64); acc.y += __shfl_xor(acc.y, off, 64);
        acc.z += __shfl_xor(acc.z, off, 64); acc.w += __shfl_xor(acc.w, off, 64);
    }
}

// (nmain: long nodes in appended workgroups, as conv_bwd_node)
template <int H>
__global__ __launch_bounds__(256) void edge_bwd_node(const float *__restrict__ xp, int64_t ncols, const int *__restrict__ nodeptr,
                                                     const int4 *__restrict__ recs, const float *__restrict__ dA_rec,
                                                     const float4 *__restrict__ rowinfo, const float *__restrict__ rs, int normalized,
                                                     int64_t row0, int64_t rows, float t, int perturb, float *__restrict__ dxp, int out_act,
                                                     unsigned nmain = 0, int wide = 0) {
    constexpr int LPR = H / 4, NPI = 64 / LPR, NBT = (32 / NPI) < 1 ? 1 : 32 / NPI, PER = NBT * NPI;
    __shared__ float4 part[4][LPR];
    const int lane = threadIdx.x & 63, c4 = lane % LPR, slot = lane / LPR, wave = dgg::wave_id();
    const bool longmode = nmain != 0 && blockIdx.x >= nmain;      // (block-uniform)
    const int64_t j = longmode ? (int64_t)(blockIdx.x - nmain) : (int64_t)blockIdx.x * 4 + wave;
    if (j >= ncols) return;
    const int p0 = nodeptr[j], p1 = nodeptr[j + 1];
    const bool islong = nmain != 0 && p1 - p0 > NODE_LONG;
    if (islong != longmode) return;
    const float4 xj = *reinterpret_cast<const float4 *>(xp + j * H + 4 * c4);
    const float aj = normalized ? __fdiv_rn(1.0f, c_sqrt(rs[j])) : 1.0f;
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (!longmode) {
        edge_bwd_walk<H>(xp, j, p0, p1, p1, recs, dA_rec, rowinfo, xj, aj, row0, t, perturb, lane, acc, wide);
    } else {
        const int chunk = ((p1 - p0 + 4 * PER - 1) / (4 * PER)) * PER;
        const int pb = p0 + wave * chunk, pe = pb + chunk < p1 ? pb + chunk : p1;
        edge_bwd_walk<H>(xp, j, pb < p1 ? pb : p1, pe, p1, recs, dA_rec, rowinfo, xj, aj, row0, t, perturb, lane, acc, wide);
        if (slot == 0) part[wave][c4] = acc;
        __syncthreads();
        if (wave != 0) return;
        acc = part[0][c4];
#pragma unroll
        for (int w_ = 1; w_ < 4; w_++) {
            const float4 v = part[w_][c4];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    if (slot == 0) {
        float4 *o = reinterpret_cast<float4 *>(dxp + j * H + 4 * c4);
        if (j >= row0 && j < row0 + rows) {
            float4 v = *o;
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        if (out_act == 1) {                                      // d loss / d (pre-activation of xp): LeakyReLU'(xp_j), xp_j is at hand
            acc.x *= xj.x > 0.0f ? 1.0f : 0.01f; acc.y *= xj.y > 0.0f ? 1.0f : 0.01f;
            acc.z *= xj.z > 0.0f ? 1.0f : 0.01f; acc.w *= xj.w > 0.0f ? 1.0f : 0.01f;
        }
        *o = acc;
    }
}

// ---- F/4 lanes per DESTINATION node (64/(F/4) nodes per wavefront) ---------------------------------------------------------
// The one-wavefront-per-node kernels above pay a fixed cost per node (pointer, H_j, stores, a 16-slot batch however short the
// list): right when a node has tens of records, wasteful when it has a handful -- a rank of 8 holds 1/8 of the rows but the
// destination set stays all N nodes, ~5 records each.  Here a group of F/4 lanes walks one node's records NBT at a time, so a
// wavefront covers 64/(F/4) nodes per instruction and needs no cross-group reduction.  Same arithmetic per record; the sums of
// a node accumulate in record order (a different order from the wavefront kernel: equal within rounding, not bit for bit).
template <int F, int NBT>
__global__ __launch_bounds__(256) void conv_bwd_nodeg(const float *__restrict__ G, const float *__restrict__ Hm, int K, int64_t ncols,
                                                      const int *__restrict__ nodeptr, const int4 *__restrict__ recs,
                                                      const float *__restrict__ rs, float *__restrict__ dA, float *__restrict__ dA_rec,
                                                      float *__restrict__ dH, float *__restrict__ da) {
    constexpr int LPR = F / 4, NPW = 64 / LPR;
    const int lane = threadIdx.x & 63, c4 = lane % LPR;
    const int64_t jraw = ((int64_t)blockIdx.x * 4 + dgg::wave_id()) * NPW + lane / LPR;
    const bool live = jraw < ncols;
    const int64_t j = live ? jraw : ncols - 1;                    // a group past the end shadows the last node and stores nothing
    const int p0 = nodeptr[j], p1 = live ? nodeptr[j + 1] : p0;
    const float4 hj = *reinterpret_cast<const float4 *>(Hm + j * F + 4 * c4);
    const float rsj = rs[j];
    const float aj = __fdiv_rn(1.0f, c_sqrt(rsj));
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float sda = 0.0f;
    int trips = (p1 - p0 + NBT - 1) / NBT;                         // the wavefront runs the longest list of its groups (DPP reductions
#pragma unroll                                                    //  below want every lane in the loop)
    for (int off = LPR; off < 64; off <<= 1) trips = max(trips, __shfl_xor(trips, off, 64));
    for (int it = 0, e0 = p0; it < trips; it++, e0 += NBT) {
        int src[NBT];
        float cf[NBT];
        float4 g[NBT];
#pragma unroll
        for (int b = 0; b < NBT; b++) {
            const bool have = e0 + b < p1;
            const int4 r = recs[have ? e0 + b : (p1 > p0 ? p1 - 1 : 0)];      // same address across the group; clamped, unconditional
            src[b] = have ? r.x : -1;
            cf[b] = have ? __fmul_rn(__int_as_float(r.z), aj) : 0.0f;
            g[b] = *reinterpret_cast<const float4 *>(G + (int64_t)(have ? (r.x >> 6) : 0) * F + 4 * c4);
        }
#pragma unroll
        for (int b = 0; b < NBT; b++) {
            float dot = g[b].x * hj.x;
            dot = fmaf(g[b].y, hj.y, dot); dot = fmaf(g[b].z, hj.z, dot); dot = fmaf(g[b].w, hj.w, dot);
            if (LPR > 16) dot += __uint_as_float(xor_shfl<16>(__float_as_uint(dot), lane));
            if (LPR > 8) dot += __uint_as_float(xor_shfl<8>(__float_as_uint(dot), lane));
            if (LPR > 4) dot += __uint_as_float(xor_shfl<4>(__float_as_uint(dot), lane));
            dot += __uint_as_float(xor_shfl<2>(__float_as_uint(dot), lane));
            dot += __uint_as_float(xor_shfl<1>(__float_as_uint(dot), lane));
            if (src[b] >= 0 && c4 == (b % LPR)) {
                if (dA) __hip_atomic_store(&dA[(int64_t)(src[b] >> 6) * K + (src[b] & 63)], dot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (see conv_bwd_node)
                dA_rec[e0 + b] = dot;
            }
            acc.x = fmaf(cf[b], g[b].x, acc.x); acc.y = fmaf(cf[b], g[b].y, acc.y);
            acc.z = fmaf(cf[b], g[b].z, acc.z); acc.w = fmaf(cf[b], g[b].w, acc.w);
            sda = fmaf(dot, cf[b], sda);
        }
    }
    if (!live) return;
    *reinterpret_cast<float4 *>(dH + j * F + 4 * c4) = acc;
    if (da && c4 == 0) da[j] = sda * sqrtf(rsj);
}

template <int H, int NBT>
__global__ __launch_bounds__(256) void edge_bwd_nodeg(const float *__restrict__ xp, int64_t ncols, const int *__restrict__ nodeptr,
                                                      const int4 *__restrict__ recs, const float *__restrict__ dA_rec,
                                                      const float4 *__restrict__ rowinfo, const float *__restrict__ rs, int normalized,
                                                      int64_t row0, int64_t rows, float t, int perturb, float *__restrict__ dxp, int out_act) {
    constexpr int LPR = H / 4, NPW = 64 / LPR;
    static_assert(NBT <= LPR, "one lane of the group per record of a batch");
    const int lane = threadIdx.x & 63, c4 = lane % LPR, gbase = lane - c4;
    const int64_t jraw = ((int64_t)blockIdx.x * 4 + dgg::wave_id()) * NPW + lane / LPR;
    const bool live = jraw < ncols;
    const int64_t j = live ? jraw : ncols - 1;
    const int p0 = nodeptr[j], p1 = live ? nodeptr[j + 1] : p0;
    const float4 xj = *reinterpret_cast<const float4 *>(xp + j * H + 4 * c4);
    const float aj = normalized ? __fdiv_rn(1.0f, c_sqrt(rs[j])) : 1.0f;
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    int trips = (p1 - p0 + NBT - 1) / NBT;
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) trips = max(trips, __shfl_xor(trips, off, 64));
    for (int it = 0, e0 = p0; it < trips; it++, e0 += NBT) {
        // lane c4 of the group forms d loss / d score of record e0 + c4 % NBT (lanes >= NBT repeat one: harmless, same value)
        const int eb = e0 + c4 % NBT;
        const bool mine = eb < p1;
        const int ec = mine ? eb : (p1 > p0 ? p1 - 1 : 0);
        const int4 myrec = recs[ec];
        float mydval;
        {
            const float4 info = rowinfo[myrec.x >> 6];
            const float dw = dA_rec[ec] * info.x * aj + info.y;
            const float th = c_tanh((float)(myrec.x & 63) - info.z);
            mydval = mine ? dw * (1.0f - 0.5f * (1.0f + th)) : 0.0f;
        }
        float4 xi[NBT];
#pragma unroll
        for (int b = 0; b < NBT; b++) {
            const int src = __shfl(myrec.x, gbase + b, 64);      // every shuffle outside a select (see conv_bwd_node)
            // unconditional gather; an inactive slot reads xp_j, whose distance to itself is 0
            xi[b] = *reinterpret_cast<const float4 *>(xp + (e0 + b < p1 ? row0 + (src >> 6) : j) * H + 4 * c4);
        }
        // lane c4 of the group owns record e0 + c4 % NBT: it takes that record's squared distance (every lane of the group holds
        // all NBT of them after the reduction) and runs the scalar chain once
        float myd2 = 0.0f;
#pragma unroll
        for (int b = 0; b < NBT; b++) {
            const float4 d = make_float4(xi[b].x - xj.x, xi[b].y - xj.y, xi[b].z - xj.z, xi[b].w - xj.w);
            float d2 = d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
            if (LPR > 16) d2 += __uint_as_float(xor_shfl<16>(__float_as_uint(d2), lane));
            if (LPR > 8) d2 += __uint_as_float(xor_shfl<8>(__float_as_uint(d2), lane));
            if (LPR > 4) d2 += __uint_as_float(xor_shfl<4>(__float_as_uint(d2), lane));
            d2 += __uint_as_float(xor_shfl<2>(__float_as_uint(d2), lane));
            d2 += __uint_as_float(xor_shfl<1>(__float_as_uint(d2), lane));
            if (c4 % NBT == b) myd2 = d2;
        }
        float mydd = 0.0f;
        if (mydval != 0.0f && myd2 != 0.0f) {
            const float dist = sqrtf(myd2);
            const float p = c_exp(t * dist);
            const float dp = perturb ? mydval * __int_as_float(myrec.w) / (p + 1e-8f) : mydval;
            mydd = dp * t * p / dist;
        }
#pragma unroll
        for (int b = 0; b < NBT; b++) {
            const float dd = __shfl(mydd, gbase + b, 64);
            const float4 d = make_float4(xi[b].x - xj.x, xi[b].y - xj.y, xi[b].z - xj.z, xi[b].w - xj.w);
            acc.x -= dd * d.x; acc.y -= dd * d.y; acc.z -= dd * d.z; acc.w -= dd * d.w;
        }
    }
    if (!live) return;
    float4 *o = reinterpret_cast<float4 *>(dxp + j * H + 4 * c4);
    if (j >= row0 && j < row0 + rows) {
        const float4 v = *o;
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (out_act == 1) {
        acc.x *= xj.x > 0.0f ? 1.0f : 0.01f; acc.y *= xj.y > 0.0f ? 1.0f : 0.01f;
        acc.z *= xj.z > 0.0f ? 1.0f : 0.01f; acc.w *= xj.w > 0.0f ? 1.0f : 0.01f;
    }
    *o = acc;
}

// ---- normalisation backward: row side (da_i, per-entry coefficient), column side (bucket sums) -----------------------
__global__ __launch_bounds__(256) void norm_da_rows(const int32_t *__restrict__ idx, const float *__restrict__ w,
                                                    const float *__restrict__ rs, const float *__restrict__ dA, int64_t rows,
                                                    int K, int64_t row0, const int *__restrict__ slotmap,
                                                    float *__restrict__ coef, float *__restrict__ da) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + dgg::wave_id();
    if (i >= rows) return;
    float rowpart = 0.0f, cf = 0.0f;
    if (lane < K) {
        const int32_t j = idx[i * K + lane];
        if (j >= 0) {
            const float g = dA[i * K + lane] * w[i * K + lane];
            if (g != 0.0f) {
                const float ai = 1.0f / sqrtf(rs[row0 + i]), aj = 1.0f / sqrtf(rs[j]);
                rowpart = g * aj;
                cf = g * ai;
            }
        }
        const int sl = slotmap[i * K + lane];
        if (sl >= 0) coef[sl] = cf;                              // record order
    }
    rowpart = wave_sum_dpp(rowpart, lane);
    if (lane == 0) da[row0 + i] = rowpart;
}
__global__ __launch_bounds__(256) void norm_da_cols(int64_t ncols, const int *__restrict__ bstart, const int2 *__restrict__ recs,
                                                    const float *__restrict__ coef, float *__restrict__ da) {
    __shared__ float ssum[BS];
    const int tid = threadIdx.x, b = blockIdx.x;
    if (tid < BS) ssum[tid] = 0.0f;
    __syncthreads();
    // records are ordered by destination: the 64 consecutive records of a wavefront form a few runs (~30 records per destination),
    // summed by a segmented scan across the lanes; only the last lane of a run touches the LDS counter
    const int e0 = bstart[b], e1 = bstart[b + 1], lane = tid & 63;
    for (int eb = e0 + (tid - lane); eb < e1; eb += 256) {      // eb: first record of this wavefront's batch (wave-uniform)
        const int e = eb + lane;
        const bool have = e < e1;
        const int dst = have ? recs[e].y : -1;
        float v = have ? coef[e] : 0.0f;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float vo = __shfl_up(v, off, 64);
            const int dd = __shfl_up(dst, off, 64);
            if (lane >= off && dd == dst) v += vo;
        }
        const int dn = __shfl_down(dst, 1, 64);
        if (have && (lane == 63 || dn != dst) && v != 0.0f) atomicAdd(&ssum[dst - b * BS], v);
    }
    __syncthreads();
    if (tid < BS) {
        const int64_t j = (int64_t)b * BS + tid;
        if (j < ncols && ssum[tid] != 0.0f) da[j] += ssum[tid];
    }
}

inline int64_t nbuckets(int64_t ncols) { return (ncols + BS - 1) / BS; }

// ---- payload partition, second form (round 4): table-driven offsets, no global atomics, one LDS atomic per record and pass ----
// The first form (part_pass<.., true> + part_scan + part_sort_p) was latency-bound: 391 workgroups of four wavefronts with a chain
// of dependent loads each (wait share 0.76, 186 us for 360 MB).  Here
//   pp_count   every workgroup of PPR rows leaves its per-bucket counts in a table T [nwg][nb]         (reads idx, w)
//   pp_scan    per bucket: exclusive prefix of T over the workgroups (in place) and the bucket total   (table-sized)
//   pp_fill    bucket starts = scan of the totals (every workgroup for itself, nb is a few hundred), then each active entry takes
//              its rank inside (workgroup, bucket) from ONE returning LDS add and is stored at start[b] + T[wg][b] + rank; ahat fused
//   pp_sort    one workgroup per bucket holds the bucket's records in REGISTERS, ranks them inside their node with one returning
//              LDS add, scans the node counts and stores every record once (buckets beyond the register budget re-read)
// Buckets are `width` nodes wide (pp_width).  Record order inside a node depends on the arrival order of the LDS adds, as before.
constexpr int PP_T = 1024;                                       // threads of a pp_sort workgroup
constexpr int PP_RPT = 16;                                       // records per thread held in registers by pp_sort
constexpr int PP_UF = 8;                                         // loads in flight per thread on its over-capacity path
inline int pp_threads() {                                        // threads of a pp_count / pp_fill workgroup (16 rows per wavefront)
    static const int t = [] { const char *e = getenv("DGG_PP_THREADS"); const int v = e ? atoi(e) : 0; return (v == 256 || v == 512 || v == 1024) ? v : 1024; }();
    return t;
}
// Bucket width in destination nodes.  One pp_sort workgroup per bucket and CU: the bucket count is made a MULTIPLE OF THE CU COUNT
// (from below) so that the last round of sort workgroups is full -- 391 power-of-two buckets on 256 CUs ran a 1.5-round tail -- with
// the expected records of a bucket (~ rows * 0.7 K * width / ncols) within 3/4 of what the sort keeps in registers.  The bucket of a
// node is j / width = umulhi(j, ceil(2^32 / width)) (exact while ncols * width < 2^32; wider problems take a power of two).
inline int pp_width(int64_t rows, int K, int64_t ncols) {
    static const int forced = [] { const char *e = getenv("DGG_PP_WIDTH"); return e ? atoi(e) : 0; }();
    const double per_node = (double)rows * (0.7 * K) / (double)(ncols > 0 ? ncols : 1);
    int64_t wmax = (int64_t)(0.75 * PP_T * PP_RPT / (per_node > 1e-9 ? per_node : 1e-9));
    wmax = wmax < 32 ? 32 : (wmax > 4096 ? 4096 : wmax);
    int64_t w = wmax;
    for (int m = 1; m <= 16; m++) {                               // smallest multiple of 256 buckets whose width fits
        const int64_t cand = (ncols + 256 * m - 1) / (256 * m);
        if (cand <= wmax) { w = cand < 32 ? 32 : cand; break; }
    }
    if (forced >= 32 && forced <= 4096) w = forced;
    while ((ncols + w - 1) / w > 4096) w *= 2;                   // LDS histograms: at most 4096 buckets
    if ((double)ncols * (double)w >= 4.0e9) {                    // the reciprocal multiply would not be exact: power of two
        int64_t p2 = 32;
        while (p2 < w) p2 *= 2;
        w = p2;
    }
    return (int)w;
}
inline uint32_t pp_recip(int width) { return (uint32_t)((((uint64_t)1 << 32) + (uint64_t)width - 1) / (uint64_t)width); }
__device__ __forceinline__ int pp_bucket(int j, uint32_t rcp) { return (int)__umulhi((uint32_t)j, rcp); }

struct PartP2 {                    // workspace: [bstart NB+1][totals NB][nodeptr NB*PBS+1][T nwg*NB][ainv ncols][tmp][recs][recpos]
    int *bstart, *totals, *nodeptr, *T;
    int *recpos;                   // [rows*64]: position of entry (row, r) among the node-ordered records (written by pp_sort for the
                                   // active entries only): the row kernel of the score backward reads dA_rec through it
    float *ainv;                   // rs_j^-1/2 of every destination node (normalize_adj fused into the fill pass)
    int4 *tmp, *recs;
    int width;                     // destination nodes per bucket
    uint32_t rcp;                  // ceil(2^32 / width)
    int64_t nb, nwg;
};
// The slot -> record map is built (one more scattered 4-byte store per record in pp_sort) only for blocks whose record-ordered dA
// fits an XCD's L2 -- where gathering through it beats the row-major scatter (see ShardedDGGConv._backward); DGG_DA_MAP=0/1 forces.
static bool pp_builds_map(int64_t rows) {
    static const int forced = [] { const char *e = getenv("DGG_DA_MAP"); return e ? atoi(e) : -1; }();
    return forced >= 0 ? forced != 0 : rows * 64 * 4 <= (8 << 20);
}
inline size_t partp2_layout(PartP2 &p, void *ws, int64_t rows, int K, int64_t ncols) {
    p.width = pp_width(rows, K, ncols);
    p.rcp = pp_recip(p.width);
    p.nb = (ncols + p.width - 1) / p.width;
    p.nwg = (rows + pp_threads() / 4 - 1) / (pp_threads() / 4);
    char *w = reinterpret_cast<char *>(ws);
    size_t o = 0;
    p.bstart = reinterpret_cast<int *>(w + o); o += align256((size_t)(p.nb + 1) * 4);
    p.totals = reinterpret_cast<int *>(w + o); o += align256((size_t)p.nb * 4);
    p.nodeptr = reinterpret_cast<int *>(w + o); o += align256((size_t)(p.nb * p.width + 1) * 4);
    p.T = reinterpret_cast<int *>(w + o); o += align256((size_t)p.nwg * p.nb * 4);
    p.ainv = reinterpret_cast<float *>(w + o); o += align256((size_t)ncols * 4);
    p.tmp = reinterpret_cast<int4 *>(w + o); o += align256((size_t)rows * K * sizeof(int4));
    p.recs = reinterpret_cast<int4 *>(w + o); o += align256((size_t)rows * K * sizeof(int4));
    p.recpos = reinterpret_cast<int *>(w + o); o += (size_t)rows * 64 * sizeof(int);
    return o;
}

// rows [r0, r0 + 16) of one wavefront as 4 x (16 bytes per lane): lane l of load u holds entries 4*(l%16) .. +3 of row r0 + 4u + l/16
__device__ __forceinline__ void pp_load16(const int32_t *__restrict__ idx, const float *__restrict__ w, int64_t rows, int64_t r0, int lane,
                                          int4 (&ji)[4], float4 (&wv)[4]) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int64_t i = r0 + 4 * u + (lane >> 4);
        const int64_t e = (i < rows ? i : rows - 1) * 64 + 4 * (lane & 15);
        ji[u] = *reinterpret_cast<const int4 *>(idx + e);
        wv[u] = *reinterpret_cast<const float4 *>(w + e);
        if (i >= rows) ji[u] = make_int4(-1, -1, -1, -1);
    }
}
__device__ __forceinline__ int pp_get(const int4 &v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }
__device__ __forceinline__ float pp_getf(const float4 &v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }

template <int THREADS>
__global__ __launch_bounds__(THREADS) void pp_count(const int32_t *__restrict__ idx, const float *__restrict__ w, int64_t rows, int K, int nb,
                                                    uint32_t rcp, int *__restrict__ T, const float *__restrict__ rs_all, int64_t ncols,
                                                    float *__restrict__ ainv) {
    extern __shared__ int lds[];
    int *hist = lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = dgg::wave_id();
    for (int b = tid; b < nb; b += THREADS) hist[b] = 0;
    if (rs_all) {                                                // this workgroup's slice of a_j = rs_j^-1/2 (the bits normalize_fwd_kernel forms)
        const int64_t per = (ncols + gridDim.x - 1) / gridDim.x;
        const int64_t j1 = ((int64_t)blockIdx.x + 1) * per < ncols ? ((int64_t)blockIdx.x + 1) * per : ncols;
        for (int64_t j = (int64_t)blockIdx.x * per + tid; j < j1; j += THREADS) ainv[j] = __fdiv_rn(1.0f, c_sqrt(rs_all[j]));
    }
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * (THREADS / 4) + wave * 16;
    if (K == 64) {
        int4 ji[4];
        float4 wv[4];
        pp_load16(idx, w, rows, r0, lane, ji, wv);
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int j = pp_get(ji[u], c);
                if (j >= 0 && pp_getf(wv[u], c) != 0.0f) atomicAdd(&hist[pp_bucket(j, rcp)], 1);
            }
    } else {
        for (int q = 0; q < 16; q++) {
            const int64_t i = r0 + q;
            if (i < rows && lane < K) {
                const int j = idx[i * K + lane];
                if (j >= 0 && w[i * K + lane] != 0.0f) atomicAdd(&hist[pp_bucket(j, rcp)], 1);
            }
        }
    }
    __syncthreads();
    int *row = T + (int64_t)blockIdx.x * nb;
    for (int b = tid; b < nb; b += THREADS) row[b] = hist[b];
}

// per bucket: T[wg][b] <- sum of T[wg'][b] over wg' < wg; totals[b] = sum over all workgroups.  One workgroup of 1024 threads
// covers 32 buckets (a 128-byte segment of every table row) x 32 slices of the workgroup range.
__global__ __launch_bounds__(1024) void pp_scan(int *__restrict__ T, int nwg, int nb, int *__restrict__ totals) {
    __shared__ int part[32][33];
    const int c = threadIdx.x & 31, s = threadIdx.x >> 5;
    const int b = blockIdx.x * 32 + c;
    const int per = (nwg + 31) / 32;
    const int lo = s * per, hi = lo + per < nwg ? lo + per : nwg;
    constexpr int CHK = 16;                                      // table entries in flight per thread (independent loads)
    int v[CHK];
    int sum = 0;
    for (int g0 = lo; g0 < hi; g0 += CHK) {
#pragma unroll
        for (int q = 0; q < CHK; q++) v[q] = (b < nb && g0 + q < hi) ? T[(int64_t)(g0 + q) * nb + b] : 0;
#pragma unroll
        for (int q = 0; q < CHK; q++) sum += v[q];
    }
    part[s][c] = sum;
    __syncthreads();
    int run = 0, tot = 0;
#pragma unroll
    for (int q = 0; q < 32; q++) {
        const int pv = part[q][c];
        if (q < s) run += pv;
        tot += pv;
    }
    if (b < nb && s == 0) totals[b] = tot;
    for (int g0 = lo; g0 < hi; g0 += CHK) {
        if (per > CHK) {                                         // (a slice longer than one chunk: re-read, still CHK loads in flight)
#pragma unroll
            for (int q = 0; q < CHK; q++) v[q] = (b < nb && g0 + q < hi) ? T[(int64_t)(g0 + q) * nb + b] : 0;
        }
#pragma unroll
        for (int q = 0; q < CHK; q++) {
            if (b < nb && g0 + q < hi) T[(int64_t)(g0 + q) * nb + b] = run;
            run += v[q];
        }
    }
}

// exclusive scan of nb values in LDS by the whole workgroup: out[b] = sum of in[< b]; returns the grand total in every thread
template <int THREADS>
__device__ __forceinline__ int pp_block_scan(const int *__restrict__ in_g, int *__restrict__ out, int nb, int *__restrict__ scratch) {
    const int tid = threadIdx.x;
    const int per = (nb + THREADS - 1) / THREADS;
    const int lo = tid * per, hi = lo + per < nb ? lo + per : nb;
    int s = 0;
    for (int b = lo; b < hi; b++) s += in_g[b];
    // wave scan, then the wave totals
    int incl = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if ((tid & 63) >= off) incl += v;
    }
    if ((tid & 63) == 63) scratch[tid >> 6] = incl;
    __syncthreads();
    int wbase = 0, total = 0;
    for (int q = 0; q < THREADS / 64; q++) {
        const int v = scratch[q];
        if (q < (tid >> 6)) wbase += v;
        total += v;
    }
    int run = wbase + incl - s;
    for (int b = lo; b < hi; b++) {
        const int v = in_g[b];
        out[b] = run;
        run += v;
    }
    __syncthreads();
    return total;
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void pp_fill(const int32_t *__restrict__ idx, const float *__restrict__ w, const float *__restrict__ val,
                                                   const float *__restrict__ rs_rows, int64_t rows, int K, int nb, uint32_t rcp,
                                                   const int *__restrict__ T, const int *__restrict__ totals, int *__restrict__ bstart,
                                                   int4 *__restrict__ recs4, const float *__restrict__ ainv, float *__restrict__ ahat_out,
                                                   const int32_t *__restrict__ cnode = nullptr) {
    // (cnode != NULL: chunked rows -- every "row" of idx / w / val is a 64-entry chunk of node cnode[row]; rs_rows is indexed by node)
    extern __shared__ int lds[];                                 // hist[nb], base[nb], scratch[16]
    int *hist = lds, *base = lds + nb, *scratch = lds + 2 * nb;
    const int tid = threadIdx.x, lane = tid & 63, wave = dgg::wave_id();
    const int64_t r0 = (int64_t)blockIdx.x * (THREADS / 4) + wave * 16;
    int4 ji[4];
    if (K == 64) {                                               // the wavefront's 16 rows of idx: in flight across the scan below
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int64_t i = r0 + 4 * u + (lane >> 4);
            ji[u] = *reinterpret_cast<const int4 *>(idx + (i < rows ? i : rows - 1) * 64 + 4 * (lane & 15));
            if (i >= rows) ji[u] = make_int4(-1, -1, -1, -1);
        }
    }
    for (int b = tid; b < nb; b += THREADS) hist[b] = 0;
    const int total = pp_block_scan<THREADS>(totals, base, nb, scratch);
    if (blockIdx.x == 0) {
        for (int b = tid; b < nb; b += THREADS) bstart[b] = base[b];
        if (tid == 0) bstart[nb] = total;
    }
    const int *trow = T + (int64_t)blockIdx.x * nb;
    for (int b = tid; b < nb; b += THREADS) base[b] += trow[b];
    __syncthreads();
    if (K == 64) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int64_t i = r0 + 4 * u + (lane >> 4);
            const bool row_ok = i < rows;
            const int64_t e = (row_ok ? i : rows - 1) * 64 + 4 * (lane & 15);
            const float4 wv = *reinterpret_cast<const float4 *>(w + e);
            const float4 vv = *reinterpret_cast<const float4 *>(val + e);
            const int64_t ic = row_ok ? i : rows - 1;
            const float ai = __fdiv_rn(1.0f, c_sqrt(rs_rows[cnode ? (int64_t)cnode[ic] : ic]));
            // a_j of the four entries gathered BEFORE the entry loop: inside it there is then no load to wait for (a wait there
            // would also wait for the record store of the previous entry: one memory round trip per entry)
            float4 aj = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (ahat_out) {
                aj.x = ainv[ji[u].x >= 0 ? ji[u].x : 0]; aj.y = ainv[ji[u].y >= 0 ? ji[u].y : 0];
                aj.z = ainv[ji[u].z >= 0 ? ji[u].z : 0]; aj.w = ainv[ji[u].w >= 0 ? ji[u].w : 0];
            }
            float4 ah = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll 1
            for (int c = 0; c < 4; c++) {                        // (not unrolled: 16 entries in flight cost 130 registers and an occupancy step)
                const int j = pp_get(ji[u], c);
                const float wx = pp_getf(wv, c);
                if (j >= 0 && wx != 0.0f) {
                    const int b = pp_bucket(j, rcp);
                    const int slot = base[b] + atomicAdd(&hist[b], 1);
                    // wa = a_i * w exactly as normalize_fwd_kernel forms it (dgg_ell.hip), so that wa * a_j == ahat bit for bit
                    const float wa = __fmul_rn(ai, wx);
                    recs4[slot] = make_int4((int)(i * 64 + 4 * (lane & 15) + c), j, (int)__float_as_uint(wa), (int)__float_as_uint(pp_getf(vv, c)));
                    const float a = __fmul_rn(wa, pp_getf(aj, c));
                    if (c == 0) ah.x = a; else if (c == 1) ah.y = a; else if (c == 2) ah.z = a; else ah.w = a;
                }
            }
            if (ahat_out && row_ok) *reinterpret_cast<float4 *>(ahat_out + i * 64 + 4 * (lane & 15)) = ah;
        }
    } else {
        for (int q = 0; q < 16; q++) {
            const int64_t i = r0 + q;
            if (i < rows && lane < K) {
                const int j = idx[i * K + lane];
                const float wx = w[i * K + lane];
                float a = 0.0f;
                if (j >= 0 && wx != 0.0f) {
                    const int b = pp_bucket(j, rcp);
                    const int slot = base[b] + atomicAdd(&hist[b], 1);
                    const float wa = __fmul_rn(__fdiv_rn(1.0f, c_sqrt(rs_rows[cnode ? (int64_t)cnode[i] : i])), wx);
                    recs4[slot] = make_int4((int)(i * 64 + lane), j, (int)__float_as_uint(wa), (int)__float_as_uint(val[i * K + lane]));
                    if (ahat_out) a = __fmul_rn(wa, ainv[j]);
                }
                if (ahat_out) ahat_out[i * K + lane] = a;
            }
        }
    }
}

// one workgroup per bucket: records -> node order; nodeptr for the bucket's nodes
// WIDE (chunked rows): the destination j of a sorted record is implied by its position (nodeptr), so its second word is rewritten to
// the SOURCE NODE cnode[chunk] -- what the per-destination kernels need to find G_i / xp_i of a chunk (no lookup there)
template <bool WIDE>
__global__ __launch_bounds__(PP_T) void pp_sort(const int *__restrict__ bstart, const int4 *__restrict__ tmp, int4 *__restrict__ recs,
                                                int *__restrict__ nodeptr, int *__restrict__ recpos, int nb, int PBS,
                                                const int32_t *__restrict__ cnode = nullptr) {
    extern __shared__ int lds[];                                 // cnt[PBS], base[PBS], scratch[16]
    int *cnt = lds, *base = lds + PBS, *scratch = lds + 2 * PBS;
    const int tid = threadIdx.x, b = blockIdx.x;
    const int e0 = bstart[b], e1 = bstart[b + 1], n = e1 - e0;
    const int o0 = e0, o1 = e1;
    for (int q = tid; q < PBS; q += PP_T) cnt[q] = 0;
    __syncthreads();
    const bool inreg = n <= PP_T * PP_RPT;                       // workgroup-uniform
    int4 rec[PP_RPT];
    int rank[PP_RPT];
    int srcnode[PP_RPT];
    if (inreg) {
#pragma unroll
        for (int u = 0; u < PP_RPT; u++) {
            const int e = e0 + u * PP_T + tid;
            rec[u] = tmp[e < e1 ? e : (n > 0 ? e1 - 1 : 0)];
            if (e >= e1) rec[u].y = -1;
        }
        // (WIDE: the source node of every record's chunk, all 16 lookups of a thread in flight together -- fetched one by one in front
        //  of each record's store they were sixteen dependent round trips: the wide sort took 607 us at k ~ 130)
        if (WIDE) {
#pragma unroll
            for (int u = 0; u < PP_RPT; u++) srcnode[u] = cnode[rec[u].y >= 0 ? (rec[u].x >> 6) : 0];
        }
#pragma unroll
        for (int u = 0; u < PP_RPT; u++) rank[u] = rec[u].y >= 0 ? atomicAdd(&cnt[rec[u].y - b * PBS], 1) : 0;
    } else {
        // a bucket beyond the register capacity (full 64-entry lists -- the hash / unperturbed generators keep every rank -- or
        // a few nodes that very many rows select): two passes over its records, PP_UF loads of a thread in flight at a time
        // (one load per iteration made this path 4x the in-register one: 35 -> 140-170 us for those generators)
        for (int eb = e0; eb < e1; eb += PP_T * PP_UF) {
            int jy[PP_UF];
#pragma unroll
            for (int u = 0; u < PP_UF; u++) {
                const int e = eb + u * PP_T + tid;
                jy[u] = e < e1 ? tmp[e].y : -1;
            }
#pragma unroll
            for (int u = 0; u < PP_UF; u++)
                if (jy[u] >= 0) atomicAdd(&cnt[jy[u] - b * PBS], 1);
        }
    }
    __syncthreads();
    pp_block_scan<PP_T>(cnt, base, PBS, scratch);
    for (int q = tid; q < PBS; q += PP_T) nodeptr[(int64_t)b * PBS + q] = o0 + base[q];
    if (b == nb - 1 && tid == 0) nodeptr[(int64_t)nb * PBS] = o1;
    if (inreg) {
#pragma unroll
        for (int u = 0; u < PP_RPT; u++)
            if (rec[u].y >= 0) {
                const int pos = o0 + base[rec[u].y - b * PBS] + rank[u];
                if (WIDE) rec[u].y = srcnode[u];
                recs[pos] = rec[u];
                if (recpos) recpos[rec[u].x] = pos;
            }
    } else {
        for (int q = tid; q < PBS; q += PP_T) cnt[q] = 0;
        __syncthreads();
        for (int eb = e0; eb < e1; eb += PP_T * PP_UF) {
            int4 r[PP_UF];
#pragma unroll
            for (int u = 0; u < PP_UF; u++) {
                const int e = eb + u * PP_T + tid;
                r[u] = tmp[e < e1 ? e : e1 - 1];
                if (e >= e1) r[u].y = -1;
            }
#pragma unroll
            for (int u = 0; u < PP_UF; u++)
                if (r[u].y >= 0) {
                    const int jl = r[u].y - b * PBS;
                    const int pos = o0 + base[jl] + atomicAdd(&cnt[jl], 1);
                    if (WIDE) r[u].y = cnode[r[u].x >> 6];
                    recs[pos] = r[u];
                    if (recpos) recpos[r[u].x] = pos;
                }
        }
    }
}

// dA_rec[e] = dA[slot of record e]: a row-major cotangent [rows,64] brought into the record order of a built payload partition (for
// callers that hold d loss / d w by rows -- the generator as a separate module -- and want the per-destination score backward)
__global__ void pp_gather_rec(const int *__restrict__ bstart, int nb, const int4 *__restrict__ recs, const float *__restrict__ dA,
                              float *__restrict__ dA_rec) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < bstart[nb]) dA_rec[e] = dA[recs[e].x];
}

}  // namespace

const int *dgg_part_slotmap(const void *part_ws, int64_t rows, int K, int64_t ncols) {
    if (!part_ws || dgg_part_ws_bytes(rows, K, ncols) == 0) return nullptr;
    return part_layout(const_cast<void *>(part_ws), nbuckets(ncols), rows * K).slot;
}

int dgg_norm_da_cols_impl(const void *part_ws, int64_t rows, int K, int64_t ncols, const float *coef_ws, float *da,
                          hipStream_t st) {
    const int64_t nb = nbuckets(ncols);
    PartHdr p = part_layout(const_cast<void *>(part_ws), nb, rows * K);
    hipLaunchKernelGGL(norm_da_cols, dim3((unsigned)nb), dim3(256), 0, st, ncols, p.bstart, p.recs, coef_ws, da);
    return dgg_check_launch("norm_da_cols");
}

extern "C" {

// bytes of workspace for the partition of an ELL block of `rows` x K entries over `ncols` destination nodes;
// 0 when the partitioned path does not apply (too many buckets for an LDS histogram)
size_t dgg_part_ws_bytes(int64_t rows, int K, int64_t ncols) {
    const int64_t nb = nbuckets(ncols);
    // 8 bytes of dynamic LDS per bucket in the fill pass (64 KiB without opting in to more); record ids are 32-bit
    if (nb > 8192 || K > 64 || K < 1 || rows * 64 >= ((int64_t)1 << 31)) return 0;
    return align256((size_t)(nb + 1) * 4) + align256((size_t)nb * 4) + align256((size_t)rows * K * sizeof(int2)) +
           align256((size_t)rows * K * sizeof(int)) + align256((size_t)rows * K * sizeof(int2)) + (size_t)(rows * K / CH + 2) * sizeof(int);
}

// Partition the ACTIVE entries (idx >= 0, w != 0) of idx [rows,K] by destination bucket.
int dgg_part_build(const int32_t *idx, const float *w, int64_t rows, int K, int64_t ncols, void *ws, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    const int64_t nb = nbuckets(ncols);
    if (dgg_part_ws_bytes(rows, K, ncols) == 0 || !ws) return dgg_set_error(DGG_ERR_UNSUPPORTED, "part_build: unsupported size or NULL workspace");
    if (rows == 0) return 0;
    PartHdr p = part_layout(ws, nb, rows * K);
    if (dgg_check_hip(hipMemsetAsync(p.cursor, 0, (size_t)nb * 4, st), "part memset") != 0) return DGG_ERR_HIP;
    const unsigned grid = (unsigned)((rows + PR - 1) / PR);
    hipLaunchKernelGGL(part_pass<false>, dim3(grid), dim3(256), (size_t)nb * 4, st, idx, w, rows, K, (int)nb, p.cursor, p.tmp, p.slot);
    hipLaunchKernelGGL(part_scan, dim3(1), dim3(1024), 0, st, p.bstart, p.cursor, (int)nb);
    hipLaunchKernelGGL(part_pass<true>, dim3(grid), dim3(256), (size_t)nb * 8, st, idx, w, rows, K, (int)nb, p.cursor, p.tmp, p.slot);
    hipLaunchKernelGGL(part_sort, dim3((unsigned)nb), dim3(1024), 0, st, p.bstart, p.tmp, K, p.recs, p.slot);
    const int nchunks = (int)((rows * K + CH - 1) / CH);
    hipLaunchKernelGGL(part_chunk_starts, dim3((unsigned)(nchunks / 256 + 1)), dim3(256), 0, st, p.bstart, (int)nb, p.recs, nchunks, CH, p.cstart);
    return dgg_check_launch("part_build");
}

// score backward through the partition: same result as dgg_edge_bwd (up to summation order), no global float atomics.
// coef_ws: rows*K + ncols floats.  dxp [ncols,h] zeroed by the caller.
static int edge_bwd_part_impl(const float *xp, int64_t rows, int h, const int32_t *idx, const float *val, const float *dval, int K,
                              int64_t row0, float t, int perturb, const void *part_ws, int64_t ncols, float *coef_ws, float *dxp,
                              const SoftkArgs *sk, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (rows == 0) return 0;
    const int64_t nb = nbuckets(ncols);
    PartHdr p = part_layout(const_cast<void *>(part_ws), nb, rows * K);
    const unsigned gr = (unsigned)((rows + 3) / 4);
    // upper bound of the chunk count (the live count bstart[nb] is read on the device: no host sync)
    const int64_t ngroups = (rows * K + CH - 1) / CH;
#define DGG_EDGE_PART(HH)                                                                                                  \
    if (sk) hipLaunchKernelGGL((edge_bwd_rows<HH, true>), dim3(gr), dim3(256), 0, st, xp, rows, idx, val, dval, K, row0, t, perturb, p.slot, coef_ws, dxp, *sk); \
    else hipLaunchKernelGGL((edge_bwd_rows<HH, false>), dim3(gr), dim3(256), 0, st, xp, rows, idx, val, dval, K, row0, t, perturb, p.slot, coef_ws, dxp, SoftkArgs{}); \
    hipLaunchKernelGGL(edge_bwd_cols<HH>, dim3((unsigned)((ngroups * (HH / 4) + 255) / 256)), dim3(256), 0, st, xp, p.bstart,   \
                       (int)nb, p.recs, coef_ws, row0, dxp)
    switch (h) {
        case 16: DGG_EDGE_PART(16); break;
        case 32: DGG_EDGE_PART(32); break;
        case 64: DGG_EDGE_PART(64); break;
        case 128: DGG_EDGE_PART(128); break;
        default: return dgg_set_error(DGG_ERR_UNSUPPORTED, "edge_bwd_part supports latent_dim in {16,32,64,128}");
    }
#undef DGG_EDGE_PART
    return dgg_check_launch("edge_bwd_part");
}

int dgg_edge_bwd_part(const float *xp, int64_t rows, int h, const int32_t *idx, const float *val, const float *dval, int K,
                      int64_t row0, float t, int perturb, const void *part_ws, int64_t ncols, float *coef_ws, float *dxp,
                      void *stream) {
    return edge_bwd_part_impl(xp, rows, h, idx, val, dval, K, row0, t, perturb, part_ws, ncols, coef_ws, dxp, nullptr, stream);
}

// dgg_softk_bwd (modes 0 / 1) + dgg_edge_bwd_part in one call: d loss / d score is formed inside the row kernel of the score
// backward.  dval (nullable) receives it as well; dk [rows] is written.
int dgg_softk_edge_bwd_part(const float *xp, int64_t rows, int h, const int32_t *idx, const float *val, const float *k, const float *rs,
                            const float *dA, const float *da, const float *ahat_rows, int K, int64_t row0, float t, int perturb,
                            int mode, int normalized, const void *part_ws, int64_t ncols, float *coef_ws, float *dval, float *dk,
                            float *dxp, void *stream) {
    if (mode != 0 && mode != 1) return dgg_set_error(DGG_ERR_ARG, "softk_edge_bwd_part: mode must be 0 (k_times) or 1 (k_only)");
    if (!k || !dA || !dk || (normalized && (!rs || !da))) return dgg_set_error(DGG_ERR_ARG, "softk_edge_bwd_part: missing operand");
    if (ahat_rows && !normalized) return dgg_set_error(DGG_ERR_ARG, "softk_edge_bwd_part: ahat_rows is an operand of the normalised form");
    const SoftkArgs sk{k, rs, dA, da, mode, normalized, dval, dk, ahat_rows, nullptr};
    return edge_bwd_part_impl(xp, rows, h, idx, val, nullptr, K, row0, t, perturb, part_ws, ncols, coef_ws, dxp, &sk, stream);
}

// dX [ncols,F] += A^T dY for an ELL block (a [rows,K] on the pattern the partition was built from); F a multiple of 64,
// dY rows 16-byte aligned.  dX is ACCUMULATED into (caller zeroes).
int dgg_ell_spmm_t_part(const float *a, const float *dY, int64_t rows, int K, int F, const void *part_ws, int64_t ncols, float *dX,
                        void *stream) {
    if (F % 64 != 0 || (reinterpret_cast<uintptr_t>(dY) % 16) != 0)
        return dgg_set_error(DGG_ERR_UNSUPPORTED, "ell_spmm_t_part: feature width must be a multiple of 64 (16-byte aligned rows)");
    if (rows == 0) return 0;
    const int64_t nb = nbuckets(ncols);
    PartHdr p = part_layout(const_cast<void *>(part_ws), nb, rows * K);
    const int64_t ngroups = (rows * K + CH - 1) / CH;
    hipLaunchKernelGGL(spmm_t_cols<false>, dim3((unsigned)((ngroups * 16 + 255) / 256), (unsigned)(F / 64)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const void *>(dY), F, p.cstart, (int)ngroups, p.recs, a, K, dX);
    return dgg_check_launch("ell_spmm_t_part");
}
// the same gathering a bf16 COPY of the cotangent (dYb [rows, F] bf16, 8-byte aligned rows)
int dgg_ell_spmm_t_part_b16(const float *a, const void *dYb, int64_t rows, int K, int F, const void *part_ws, int64_t ncols, float *dX,
                            void *stream) {
    if (F % 64 != 0 || (reinterpret_cast<uintptr_t>(dYb) % 8) != 0)
        return dgg_set_error(DGG_ERR_UNSUPPORTED, "ell_spmm_t_part_b16: feature width must be a multiple of 64 (8-byte aligned rows)");
    if (rows == 0) return 0;
    const int64_t nb = nbuckets(ncols);
    PartHdr p = part_layout(const_cast<void *>(part_ws), nb, rows * K);
    const int64_t ngroups = (rows * K + CH - 1) / CH;
    hipLaunchKernelGGL(spmm_t_cols<true>, dim3((unsigned)((ngroups * 16 + 255) / 256), (unsigned)(F / 64)), dim3(256), 0, (hipStream_t)stream, dYb,
                       F, p.cstart, (int)ngroups, p.recs, a, K, dX);
    return dgg_check_launch("ell_spmm_t_part_b16");
}

// Backward of Z = A H through the partition, one gather of G per record (conv_bwd_cols): dA [rows,K] (entries outside the
// partition are NOT written: caller zeroes), dH [ncols,F] and da [ncols] (nullable) accumulated into (caller zeroes).
int dgg_ell_conv_bwd_part(const float *G, const float *H, const float *ahat, int64_t rows, int K, int F, const void *part_ws,
                          int64_t ncols, const float *rs, float *dA, float *dH, float *da, void *stream) {
    if ((F != 16 && F != 32 && F != 64 && F != 128) || (reinterpret_cast<uintptr_t>(G) % 16) || (reinterpret_cast<uintptr_t>(H) % 16) ||
        (reinterpret_cast<uintptr_t>(dH) % 16))
        return dgg_set_error(DGG_ERR_UNSUPPORTED, "ell_conv_bwd_part: feature width must be 16, 32, 64 or 128 (16-byte aligned rows)");
    if (da && !rs) return dgg_set_error(DGG_ERR_ARG, "ell_conv_bwd_part: da needs the row sums");
    if (!dgg_part_slotmap(part_ws, rows, K, ncols)) return dgg_set_error(DGG_ERR_ARG, "ell_conv_bwd_part: no partition");
    if (rows == 0) return 0;
    const int64_t nb = nbuckets(ncols);
    PartHdr p = part_layout(const_cast<void *>(part_ws), nb, rows * K);
    const int64_t ngroups = (rows * K + CH - 1) / CH;
    hipStream_t st = (hipStream_t)stream;
#define DGG_CONV_COLS(FF)                                                                                                  \
    hipLaunchKernelGGL(conv_bwd_cols<FF>, dim3((unsigned)((ngroups * (FF / 4) + 255) / 256)), dim3(256), 0, st, G, H, ahat, K, \
                       p.bstart, (int)nb, p.recs, rs, dA, dH, da)
    switch (F) {
        case 16: DGG_CONV_COLS(16); break;
        case 32: DGG_CONV_COLS(32); break;
        case 64: DGG_CONV_COLS(64); break;
        default: DGG_CONV_COLS(128); break;
    }
#undef DGG_CONV_COLS
    return dgg_check_launch("ell_conv_bwd_part");
}

// ---- payload partition (16-byte records carrying wa = w * rs_i^-1/2 and the score; no slot map) ---------------------------
int dgg_partp_build_norm(const int32_t *idx, const float *w, const float *val, const float *rs_rows, int64_t rows, int K, int64_t ncols,
                         const float *rs_all, float *ahat, void *ws, void *stream);
int dgg_partp_build_phase(const int32_t *idx, const float *w, const float *val, const float *rs_rows, int64_t rows, int K, int64_t ncols,
                          const float *rs_all, float *ahat, void *ws, int phase, void *stream);
size_t dgg_partp_ws_bytes(int64_t rows, int K, int64_t ncols) {
    if (K > 64 || K < 1 || rows * 64 >= ((int64_t)1 << 31) || ncols < 1) return 0;          // 32-bit record ids
    PartP2 p;
    const size_t bytes = partp2_layout(p, nullptr, rows, K, ncols);
    if (p.nb > 4096 || p.width > 4096) return 0;                                             // LDS histograms / node counters
    return bytes;
}

// Partition the ACTIVE entries (idx >= 0, w != 0) of an ELL block by destination, records = (row*64 + r, j, w * rs_i^-1/2, val).
// val [rows,K] = the scores (dgg_allpairs_topk / dgg_edgelist_topk), rs_rows [rows] = the row sums of the block's OWN rows.
int dgg_partp_build(const int32_t *idx, const float *w, const float *val, const float *rs_rows, int64_t rows, int K, int64_t ncols,
                    void *ws, void *stream) {
    return dgg_partp_build_norm(idx, w, val, rs_rows, rows, K, ncols, nullptr, nullptr, ws, stream);
}

// the same with normalize_adj fused: rs_all [ncols] (row sums of EVERY node) -> ahat [rows,K] = rs_i^-1/2 w rs_j^-1/2, the bits
// of dgg_ell_normalize_fwd (entries outside the partition: 0)
int dgg_partp_build_norm(const int32_t *idx, const float *w, const float *val, const float *rs_rows, int64_t rows, int K, int64_t ncols,
                         const float *rs_all, float *ahat, void *ws, void *stream) {
    return dgg_partp_build_phase(idx, w, val, rs_rows, rows, K, ncols, rs_all, ahat, ws, 0, stream);
}

// the same in two parts, so that a caller can run the second beside other work on another stream: phase 1 = count + scan + fill
// (ahat is complete, the records sit in bucket order), phase 2 = the per-bucket sort (records in node order + nodeptr: what the
// backward's column kernels read; the forward aggregation does not need it); phase 0 = both
int dgg_partp_build_phase(const int32_t *idx, const float *w, const float *val, const float *rs_rows, int64_t rows, int K, int64_t ncols,
                          const float *rs_all, float *ahat, void *ws, int phase, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (phase < 0 || phase > 2) return dgg_set_error(DGG_ERR_ARG, "partp_build_phase: phase is 0 (all), 1 (count + fill) or 2 (sort)");
    if (dgg_partp_ws_bytes(rows, K, ncols) == 0 || !ws) return dgg_set_error(DGG_ERR_UNSUPPORTED, "partp_build: unsupported size or NULL workspace");
    if (!val || !rs_rows) return dgg_set_error(DGG_ERR_ARG, "partp_build: the payload needs the scores and the row sums");
    if ((rs_all == nullptr) != (ahat == nullptr)) return dgg_set_error(DGG_ERR_ARG, "partp_build_norm: rs_all and ahat go together");
    if (K == 64 && ((reinterpret_cast<uintptr_t>(idx) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(val) |
                     reinterpret_cast<uintptr_t>(ahat)) % 16))
        return dgg_set_error(DGG_ERR_ARG, "partp_build: idx / w / val / ahat must be 16-byte aligned");
    if (rows == 0) return 0;
    PartP2 p;
    partp2_layout(p, ws, rows, K, ncols);
    const int nb = (int)p.nb, nwg = (int)p.nwg, pbs = p.width;
    if (phase == 2) {
        hipLaunchKernelGGL(pp_sort<false>, dim3((unsigned)nb), dim3(PP_T), (size_t)(2 * pbs + 16) * 4, st, p.bstart, p.tmp, p.recs, p.nodeptr, pp_builds_map(rows) ? p.recpos : nullptr, nb, pbs, (const int32_t *)nullptr);
        return dgg_check_launch("partp_build");
    }
#define DGG_PP_PASS(TT)                                                                                                      \
    hipLaunchKernelGGL(pp_count<TT>, dim3((unsigned)nwg), dim3(TT), (size_t)nb * 4, st, idx, w, rows, K, nb, p.rcp, p.T, rs_all, ncols, \
                       p.ainv);                                                                                              \
    hipLaunchKernelGGL(pp_scan, dim3((unsigned)((nb + 31) / 32)), dim3(1024), 0, st, p.T, nwg, nb, p.totals);                \
    hipLaunchKernelGGL(pp_fill<TT>, dim3((unsigned)nwg), dim3(TT), (size_t)(2 * nb + 16) * 4, st, idx, w, val, rs_rows, rows, K, nb, p.rcp, \
                       p.T, p.totals, p.bstart, p.tmp, p.ainv, ahat)
    switch (pp_threads()) {
        case 256: DGG_PP_PASS(256); break;
        case 512: DGG_PP_PASS(512); break;
        default: DGG_PP_PASS(1024); break;
    }
#undef DGG_PP_PASS
    if (phase == 1) return dgg_check_launch("partp_build");
    hipLaunchKernelGGL(pp_sort<false>, dim3((unsigned)nb), dim3(PP_T), (size_t)(2 * pbs + 16) * 4, st, p.bstart, p.tmp, p.recs, p.nodeptr, pp_builds_map(rows) ? p.recpos : nullptr, nb, pbs, (const int32_t *)nullptr);
    return dgg_check_launch("partp_build");
}

// 1 when dgg_partp_build leaves the slot -> record map for a block of `rows` rows (then dgg_ell_conv_bwd_partp / dgg_softk_edge_bwd_partp
// may be called with dA = NULL)
int dgg_partp_has_map(int64_t rows) { return pp_builds_map(rows) ? 1 : 0; }

}  // extern "C"
int dgg_partp_internal_ptrs(const void *partp_ws, int64_t rows, int K, int64_t ncols, const int **nodeptr, const int **recpos) {
    if (!partp_ws || dgg_partp_ws_bytes(rows, K, ncols) == 0 || !pp_builds_map(rows)) return 1;
    PartP2 p;
    partp2_layout(p, const_cast<void *>(partp_ws), rows, K, ncols);
    *nodeptr = p.nodeptr;
    *recpos = p.recpos;
    return 0;
}
extern "C" {
int dgg_partp_describe(int64_t rows, int K, int64_t ncols, int64_t *out6) {
    if (!out6 || dgg_partp_ws_bytes(rows, K, ncols) == 0) return dgg_set_error(DGG_ERR_UNSUPPORTED, "partp_describe: no payload partition for this shape");
    PartP2 p;
    partp2_layout(p, nullptr, rows, K, ncols);
    out6[0] = reinterpret_cast<char *>(p.bstart) - static_cast<char *>(nullptr);
    out6[1] = reinterpret_cast<char *>(p.nodeptr) - static_cast<char *>(nullptr);
    out6[2] = reinterpret_cast<char *>(p.recs) - static_cast<char *>(nullptr);
    out6[3] = p.nb;
    out6[4] = p.width;
    out6[5] = pp_threads() / 4;
    return 0;
}

// which node kernel: a group of lanes per node when the lists are short (fewer than 16 records per node on average; a rank of G
// holds 64/G of them), a wavefront per node otherwise.  DGG_NODE_GROUPS=0/1 forces one (measurement only).
// long nodes in appended workgroups (conv_bwd_node / edge_bwd_node): graphs of at most NODE_SPLIT_MAX nodes; DGG_NODE_SPLIT=0/1 forces
static bool node_split(int64_t ncols) {
    static const int forced = [] { const char *e = getenv("DGG_NODE_SPLIT"); return e ? atoi(e) : -1; }();
    return forced >= 0 ? forced != 0 : ncols <= NODE_SPLIT_MAX;
}
static bool node_groups(int64_t nrec, int64_t ncols) {
    static const int forced = [] { const char *e = getenv("DGG_NODE_GROUPS"); return e ? atoi(e) : -1; }();
    return forced >= 0 ? forced != 0 : nrec < 16 * ncols;
}

// dgg_ell_conv_bwd_part on a payload partition: ahat comes from the records; additionally dA_rec [rows*K] = dA in record order
// (for dgg_softk_edge_bwd_partp).  dA [rows,K], dH [ncols,F], da [ncols] as in dgg_ell_conv_bwd_part (caller zeroes all three).
int dgg_ell_conv_bwd_partp_ext(const float *G, const float *H, int64_t rows, int K, int F, const void *partp_ws, int64_t ncols,
                               const float *rs, const float *dA_ext, float *dA, float *dA_rec, float *dH, float *da, void *stream);
int dgg_ell_conv_bwd_partp(const float *G, const float *H, int64_t rows, int K, int F, const void *partp_ws, int64_t ncols,
                           const float *rs, float *dA, float *dA_rec, float *dH, float *da, void *stream) {
    return dgg_ell_conv_bwd_partp_ext(G, H, rows, K, F, partp_ws, ncols, rs, nullptr, dA, dA_rec, dH, da, stream);
}
// the same with an additional cotangent of the normalised adjacency from its other consumers (dA_ext [rows,K], nullable; entries
// outside the partition are not read): dA / dA_rec / da then hold the totals
int dgg_ell_conv_bwd_partp_ext(const float *G, const float *H, int64_t rows, int K, int F, const void *partp_ws, int64_t ncols,
                               const float *rs, const float *dA_ext, float *dA, float *dA_rec, float *dH, float *da, void *stream) {
    if ((F != 16 && F != 32 && F != 64 && F != 128) || (reinterpret_cast<uintptr_t>(G) % 16) || (reinterpret_cast<uintptr_t>(H) % 16) ||
        (reinterpret_cast<uintptr_t>(dH) % 16))
        return dgg_set_error(DGG_ERR_UNSUPPORTED, "ell_conv_bwd_partp: feature width must be 16, 32, 64 or 128 (16-byte aligned rows)");
    if (!rs || !partp_ws || !dA_rec || dgg_partp_ws_bytes(rows, K, ncols) == 0) return dgg_set_error(DGG_ERR_ARG, "ell_conv_bwd_partp: missing operand");
    if (rows == 0) return 0;
    PartP2 p;
    partp2_layout(p, const_cast<void *>(partp_ws), rows, K, ncols);
    hipStream_t st = (hipStream_t)stream;
    const bool grouped = !dA_ext && node_groups(rows * K, ncols);
    const unsigned nmain = (unsigned)((ncols + 3) / 4);
    const bool split = node_split(ncols);                         // long nodes in appended workgroups (see conv_bwd_node)
    const dim3 gridn(split ? nmain + (unsigned)ncols : nmain);
#define DGG_CONV_COLS_P(FF)                                                                                                \
    if (dA_ext)                                                                                                            \
        hipLaunchKernelGGL((conv_bwd_node<FF, true>), gridn, dim3(256), 0, st, G, H, K, ncols, p.nodeptr, p.recs,          \
                           rs, dA, dA_rec, dH, da, dA_ext, split ? nmain : 0u);                                            \
    else if (grouped)                                                                                                      \
        hipLaunchKernelGGL((conv_bwd_nodeg<FF, 4>), dim3((unsigned)((ncols + 4 * (256 / FF) - 1) / (4 * (256 / FF)))), dim3(256), 0, st, G, H, K, \
                           ncols, p.nodeptr, p.recs, rs, dA, dA_rec, dH, da);                                              \
    else                                                                                                                   \
        hipLaunchKernelGGL((conv_bwd_node<FF, false>), gridn, dim3(256), 0, st, G, H, K, ncols, p.nodeptr, p.recs,         \
                           rs, dA, dA_rec, dH, da, (const float *)nullptr, split ? nmain : 0u)
    switch (F) {
        case 16: DGG_CONV_COLS_P(16); break;
        case 32: DGG_CONV_COLS_P(32); break;
        case 64: DGG_CONV_COLS_P(64); break;
        default: DGG_CONV_COLS_P(128); break;
    }
#undef DGG_CONV_COLS_P
    return dgg_check_launch("ell_conv_bwd_partp");
}

int dgg_softk_edge_bwd_partp_phase(const float *xp, int64_t rows, int h, const int32_t *idx, const float *val, const float *k, const float *rs,
                                   const float *dA, const float *dA_rec, const float *da, const float *ahat_rows, int K, int64_t row0, float t,
                                   int perturb, int mode, int normalized, const void *partp_ws, int64_t ncols, float *rowinfo_ws, float *dk,
                                   float *dxp, int out_act, int phase, void *stream);
// dgg_softk_edge_bwd_part on a payload partition: the row kernel hands (a_i, d loss / d rs_i, k_i) per row to the column kernel
// (rowinfo_ws: 4*rows floats), which recomputes d loss / d score from dA_rec in record order -- no slot map, no per-entry
// coefficient hand-over.  dk [rows] written, dxp [ncols,h] zeroed by the caller.
int dgg_softk_edge_bwd_partp(const float *xp, int64_t rows, int h, const int32_t *idx, const float *val, const float *k, const float *rs,
                             const float *dA, const float *dA_rec, const float *da, const float *ahat_rows, int K, int64_t row0, float t,
                             int perturb, int mode, int normalized, const void *partp_ws, int64_t ncols, float *rowinfo_ws, float *dk,
                             float *dxp, int out_act, void *stream) {
    return dgg_softk_edge_bwd_partp_phase(xp, rows, h, idx, val, k, rs, dA, dA_rec, da, ahat_rows, K, row0, t, perturb, mode, normalized,
                                          partp_ws, ncols, rowinfo_ws, dk, dxp, out_act, 0, stream);
}

// the same in two parts: phase 1 = the row kernel (dk and the rows' own side of dxp are complete: the k-net backward can start on
// another stream), phase 2 = the per-destination kernel (dxp complete); phase 0 = both
int dgg_softk_edge_bwd_partp_phase(const float *xp, int64_t rows, int h, const int32_t *idx, const float *val, const float *k, const float *rs,
                                   const float *dA, const float *dA_rec, const float *da, const float *ahat_rows, int K, int64_t row0, float t,
                                   int perturb, int mode, int normalized, const void *partp_ws, int64_t ncols, float *rowinfo_ws, float *dk,
                                   float *dxp, int out_act, int phase, void *stream) {
    if (phase < 0 || phase > 2) return dgg_set_error(DGG_ERR_ARG, "softk_edge_bwd_partp_phase: phase is 0 (all), 1 (rows) or 2 (nodes)");
    if (mode != 0 && mode != 1) return dgg_set_error(DGG_ERR_ARG, "softk_edge_bwd_partp: mode must be 0 (k_times) or 1 (k_only)");
    if (out_act != 0 && (out_act != 1 || mode != 0))             // (mode 1 launches no node kernel: nothing would apply the mask)
        return dgg_set_error(DGG_ERR_ARG, "softk_edge_bwd_partp: out_act is 0 or 1 (LeakyReLU), mode 0 only");
    if (!k || !dA_rec || !dk || !rowinfo_ws || (normalized && (!rs || !da))) return dgg_set_error(DGG_ERR_ARG, "softk_edge_bwd_partp: missing operand");
    if (!dA && !(normalized && ahat_rows))       // (the slot -> record map covers the partition's entries; ahat_rows tells which those are)
        return dgg_set_error(DGG_ERR_ARG, "softk_edge_bwd_partp: dA may be NULL only in the normalised form with ahat_rows");
    if (!dA && !pp_builds_map(rows))
        return dgg_set_error(DGG_ERR_ARG, "softk_edge_bwd_partp: dA is NULL but the partition of a block this large carries no slot -> record "
                                          "map (dgg_partp_has_map)");
    if (ahat_rows && !normalized) return dgg_set_error(DGG_ERR_ARG, "softk_edge_bwd_partp: ahat_rows is an operand of the normalised form");
    if (!partp_ws || dgg_partp_ws_bytes(rows, K, ncols) == 0) return dgg_set_error(DGG_ERR_ARG, "softk_edge_bwd_partp: no partition");
    if (rows == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    PartP2 p;
    partp2_layout(p, const_cast<void *>(partp_ws), rows, K, ncols);
    const SoftkArgs sk{k, rs, dA, da, mode, normalized, nullptr, dk, ahat_rows, reinterpret_cast<float4 *>(rowinfo_ws),
                       dA ? nullptr : p.recpos, dA_rec};
    const unsigned gr = (unsigned)((rows + 3) / 4);
    const bool grouped = node_groups(rows * K, ncols);
    const unsigned nmain = (unsigned)((ncols + 3) / 4);
    const bool split = node_split(ncols);
#define DGG_EDGE_PARTP(HH)                                                                                                  \
    if (phase != 2)                                                                                                          \
        hipLaunchKernelGGL((edge_bwd_rows<HH, true, true>), dim3(gr), dim3(256), 0, st, xp, rows, idx, val, nullptr, K, row0, t, perturb, nullptr, \
                           nullptr, dxp, sk);                                                                                \
    if (phase == 1) {                                                                                                        \
    } else if (mode == 0 && grouped)                                                                                         \
        hipLaunchKernelGGL((edge_bwd_nodeg<HH, 4>), dim3((unsigned)((ncols + 4 * (256 / HH) - 1) / (4 * (256 / HH)))), dim3(256), 0, st, xp, \
                           ncols, p.nodeptr, p.recs, dA_rec, reinterpret_cast<const float4 *>(rowinfo_ws), rs, normalized, row0, rows, t, \
                           perturb, dxp, out_act);                                                                           \
    else if (mode == 0)                                                                                                      \
        hipLaunchKernelGGL(edge_bwd_node<HH>, dim3(split ? nmain + (unsigned)ncols : nmain), dim3(256), 0, st, xp, ncols, p.nodeptr, p.recs, \
                           dA_rec, reinterpret_cast<const float4 *>(rowinfo_ws), rs, normalized, row0, rows, t, perturb, dxp, out_act,  \
                           split ? nmain : 0u)
    switch (h) {
        case 16: DGG_EDGE_PARTP(16); break;
        case 32: DGG_EDGE_PARTP(32); break;
        case 64: DGG_EDGE_PARTP(64); break;
        case 128: DGG_EDGE_PARTP(128); break;
        default: return dgg_set_error(DGG_ERR_UNSUPPORTED, "softk_edge_bwd_partp supports latent_dim in {16,32,64,128}");
    }
#undef DGG_EDGE_PARTP
    return dgg_check_launch("softk_edge_bwd_partp");
}

// ---- chunked rows (rows wider than 64 ranks, dgg_chunk_layout): the same three steps with the node of every chunk -----------------
// dgg_partp_build_phase on the [chunks,64] arrays of chunked rows: rs_nodes [nodes of the block] is indexed through cnode [chunks];
// the SORTED records carry (chunk * 64 + entry, SOURCE NODE of the chunk, w rs_i^-1/2, score) -- the destination is implied by nodeptr
int dgg_partp_build_chunked(const int32_t *idx, const float *w, const float *val, const float *rs_nodes, int64_t chunks, const int32_t *cnode,
                            int64_t ncols, const float *rs_all, float *ahat, void *ws, int phase, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    const int K = 64;
    if (phase < 0 || phase > 2) return dgg_set_error(DGG_ERR_ARG, "partp_build_chunked: phase is 0 (all), 1 (count + fill) or 2 (sort)");
    if (dgg_partp_ws_bytes(chunks, K, ncols) == 0 || !ws) return dgg_set_error(DGG_ERR_UNSUPPORTED, "partp_build_chunked: unsupported size or NULL workspace");
    if (!val || !rs_nodes || !cnode) return dgg_set_error(DGG_ERR_ARG, "partp_build_chunked: the payload needs the scores, the row sums and the chunk nodes");
    if ((rs_all == nullptr) != (ahat == nullptr)) return dgg_set_error(DGG_ERR_ARG, "partp_build_chunked: rs_all and ahat go together");
    if ((reinterpret_cast<uintptr_t>(idx) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(val) | reinterpret_cast<uintptr_t>(ahat)) % 16)
        return dgg_set_error(DGG_ERR_ARG, "partp_build_chunked: idx / w / val / ahat must be 16-byte aligned");
    if (chunks == 0) return 0;
    PartP2 p;
    partp2_layout(p, ws, chunks, K, ncols);
    const int nb = (int)p.nb, nwg = (int)p.nwg, pbs = p.width;
    if (phase != 2) {
#define DGG_PP_PASS(TT)                                                                                                      \
    hipLaunchKernelGGL(pp_count<TT>, dim3((unsigned)nwg), dim3(TT), (size_t)nb * 4, st, idx, w, chunks, K, nb, p.rcp, p.T, rs_all, ncols, \
                       p.ainv);                                                                                              \
    hipLaunchKernelGGL(pp_scan, dim3((unsigned)((nb + 31) / 32)), dim3(1024), 0, st, p.T, nwg, nb, p.totals);                \
    hipLaunchKernelGGL(pp_fill<TT>, dim3((unsigned)nwg), dim3(TT), (size_t)(2 * nb + 16) * 4, st, idx, w, val, rs_nodes, chunks, K, nb, p.rcp, \
                       p.T, p.totals, p.bstart, p.tmp, p.ainv, ahat, cnode)
        switch (pp_threads()) {
            case 256: DGG_PP_PASS(256); break;
            case 512: DGG_PP_PASS(512); break;
            default: DGG_PP_PASS(1024); break;
        }
#undef DGG_PP_PASS
    }
    if (phase != 1)
        hipLaunchKernelGGL(pp_sort<true>, dim3((unsigned)nb), dim3(PP_T), (size_t)(2 * pbs + 16) * 4, st, p.bstart, p.tmp, p.recs, p.nodeptr,
                           (int *)nullptr, nb, pbs, cnode);
    return dgg_check_launch("partp_build_chunked");
}

// dgg_ell_conv_bwd_partp_ext on a partition built by dgg_partp_build_chunked: G [nodes of the block, F] is read at the record's source
// node; dA [chunks,64] (entries outside the partition are not written), dA_rec [chunks*64], dA_ext [chunks,64] (nullable)
int dgg_ell_conv_bwd_partp_chunked(const float *G, const float *H, int64_t chunks, int F, const void *partp_ws, int64_t ncols, const float *rs,
                                   const float *dA_ext, float *dA, float *dA_rec, float *dH, float *da, void *stream) {
    const int K = 64;
    if ((F != 16 && F != 32 && F != 64 && F != 128) || (reinterpret_cast<uintptr_t>(G) % 16) || (reinterpret_cast<uintptr_t>(H) % 16) ||
        (reinterpret_cast<uintptr_t>(dH) % 16))
        return dgg_set_error(DGG_ERR_UNSUPPORTED, "ell_conv_bwd_partp_chunked: feature width must be 16, 32, 64 or 128 (16-byte aligned rows)");
    if (!rs || !partp_ws || !dA_rec || !dA || dgg_partp_ws_bytes(chunks, K, ncols) == 0) return dgg_set_error(DGG_ERR_ARG, "ell_conv_bwd_partp_chunked: missing operand");
    if (chunks == 0) return 0;
    PartP2 p;
    partp2_layout(p, const_cast<void *>(partp_ws), chunks, K, ncols);
    hipStream_t st = (hipStream_t)stream;
    const unsigned nmain = (unsigned)((ncols + 3) / 4);
    const bool split = node_split(ncols);
    const dim3 gridn(split ? nmain + (unsigned)ncols : nmain);
#define DGG_CONV_COLS_C(FF)                                                                                                \
    if (dA_ext)                                                                                                            \
        hipLaunchKernelGGL((conv_bwd_node<FF, true>), gridn, dim3(256), 0, st, G, H, K, ncols, p.nodeptr, p.recs,          \
                           rs, dA, dA_rec, dH, da, dA_ext, split ? nmain : 0u, 1);                                         \
    else                                                                                                                   \
        hipLaunchKernelGGL((conv_bwd_node<FF, false>), gridn, dim3(256), 0, st, G, H, K, ncols, p.nodeptr, p.recs,         \
                           rs, dA, dA_rec, dH, da, (const float *)nullptr, split ? nmain : 0u, 1)
    switch (F) {
        case 16: DGG_CONV_COLS_C(16); break;
        case 32: DGG_CONV_COLS_C(32); break;
        case 64: DGG_CONV_COLS_C(64); break;
        default: DGG_CONV_COLS_C(128); break;
    }
#undef DGG_CONV_COLS_C
    return dgg_check_launch("ell_conv_bwd_partp_chunked");
}

// dgg_softk_edge_bwd_partp_phase on chunked rows: `rows` NODES with chunks [cptr[i], cptr[i+1]) of idx / val / dA / ahat_rows [chunks,64];
// k, dk [rows]; rowinfo_ws 4 * chunks floats; dA is required (no slot -> record map for chunked rows)
int dgg_softk_edge_bwd_partp_chunked(const float *xp, int64_t rows, const int32_t *cptr, int64_t chunks, int h, const int32_t *idx, const float *val,
                                     const float *k, const float *rs, const float *dA, const float *dA_rec, const float *da, const float *ahat_rows,
                                     int64_t row0, float t, int perturb, int mode, int normalized, const void *partp_ws, int64_t ncols,
                                     float *rowinfo_ws, float *dk, float *dxp, int out_act, int phase, void *stream) {
    const int K = 64;
    if (phase < 0 || phase > 2) return dgg_set_error(DGG_ERR_ARG, "softk_edge_bwd_partp_chunked: phase is 0 (all), 1 (rows) or 2 (nodes)");
    if (mode != 0 && mode != 1) return dgg_set_error(DGG_ERR_ARG, "softk_edge_bwd_partp_chunked: mode must be 0 (k_times) or 1 (k_only)");
    if (out_act != 0 && (out_act != 1 || mode != 0)) return dgg_set_error(DGG_ERR_ARG, "softk_edge_bwd_partp_chunked: out_act is 0 or 1 (LeakyReLU), mode 0 only");
    if (!k || !cptr || !dA || !dA_rec || !dk || !rowinfo_ws || (normalized && (!rs || !da)))
        return dgg_set_error(DGG_ERR_ARG, "softk_edge_bwd_partp_chunked: missing operand");
    if (ahat_rows && !normalized) return dgg_set_error(DGG_ERR_ARG, "softk_edge_bwd_partp_chunked: ahat_rows is an operand of the normalised form");
    if (!partp_ws || dgg_partp_ws_bytes(chunks, K, ncols) == 0) return dgg_set_error(DGG_ERR_ARG, "softk_edge_bwd_partp_chunked: no partition");
    if (rows == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    PartP2 p;
    partp2_layout(p, const_cast<void *>(partp_ws), chunks, K, ncols);
    const SoftkArgs sk{k, rs, dA, da, mode, normalized, nullptr, dk, ahat_rows, reinterpret_cast<float4 *>(rowinfo_ws), nullptr, dA_rec};
    const unsigned gr = (unsigned)((rows + 3) / 4);
    const unsigned nmain = (unsigned)((ncols + 3) / 4);
    const bool split = node_split(ncols);
#define DGG_EDGE_PARTC(HH)                                                                                                   \
    if (phase != 2)                                                                                                          \
        hipLaunchKernelGGL((edge_bwd_rows_chunked<HH>), dim3(gr), dim3(256), 0, st, xp, rows, cptr, idx, val, row0, t, perturb, dxp, sk); \
    if (phase != 1 && mode == 0)                                                                                             \
        hipLaunchKernelGGL(edge_bwd_node<HH>, dim3(split ? nmain + (unsigned)ncols : nmain), dim3(256), 0, st, xp, ncols, p.nodeptr, p.recs, \
                           dA_rec, reinterpret_cast<const float4 *>(rowinfo_ws), rs, normalized, row0, rows, t, perturb, dxp, out_act,  \
                           split ? nmain : 0u, 1)
    switch (h) {
        case 16: DGG_EDGE_PARTC(16); break;
        case 32: DGG_EDGE_PARTC(32); break;
        case 64: DGG_EDGE_PARTC(64); break;
        case 128: DGG_EDGE_PARTC(128); break;
        default: return dgg_set_error(DGG_ERR_UNSUPPORTED, "softk_edge_bwd_partp_chunked supports latent_dim in {16,32,64,128}");
    }
#undef DGG_EDGE_PARTC
    return dgg_check_launch("softk_edge_bwd_partp_chunked");
}

// dA [rows,64] (row-major; chunked rows: [chunks,64]) -> dA_rec [rows*64] in the record order of a built payload partition (K = 64)
int dgg_partp_gather_rec(const float *dA, int64_t rows, int64_t ncols, const void *partp_ws, float *dA_rec, void *stream) {
    if (!dA || !dA_rec || !partp_ws || dgg_partp_ws_bytes(rows, 64, ncols) == 0) return dgg_set_error(DGG_ERR_ARG, "partp_gather_rec: missing operand or no partition");
    if (rows == 0) return 0;
    PartP2 p;
    partp2_layout(p, const_cast<void *>(partp_ws), rows, 64, ncols);
    hipLaunchKernelGGL(pp_gather_rec, dim3((unsigned)((rows * 64 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p.bstart, (int)p.nb, p.recs, dA, dA_rec);
    return dgg_check_launch("partp_gather_rec");
}

// normalisation backward phase 1 through the partition; da [ncols] zeroed by the caller
int dgg_norm_bwd_da_part(const int32_t *idx, const float *w, const float *rs, const float *dA, int64_t rows, int K, int64_t row0,
                         const void *part_ws, int64_t ncols, float *coef_ws, float *da, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (rows == 0) return 0;
    const int64_t nb = nbuckets(ncols);
    PartHdr p = part_layout(const_cast<void *>(part_ws), nb, rows * K);
    hipLaunchKernelGGL(norm_da_rows, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, idx, w, rs, dA, rows, K, row0, p.slot, coef_ws, da);
    hipLaunchKernelGGL(norm_da_cols, dim3((unsigned)nb), dim3(256), 0, st, ncols, p.bstart, p.recs, coef_ws, da);
    return dgg_check_launch("norm_bwd_da_part");
}

}  // extern "C"
