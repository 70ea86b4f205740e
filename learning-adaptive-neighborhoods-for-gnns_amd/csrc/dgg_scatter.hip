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

struct PartHdr {                   // workspace: [bstart NB+1][cursor NB][tmp rows*K int2][slot rows*K int][recs rows*K int2]
    int *bstart, *cursor;
    int2 *tmp;                     // bucket-ordered records (src = row*64 + r, dst = j)
    int *slot;                     // slot[row*K + r] = position of the entry in recs, -1 if inactive: lets the row-side
    int2 *recs;                    // kernels write their per-entry coefficient in RECORD order (coalesced reads later)
};                                 // recs: the records ordered by destination node
__host__ __device__ inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }
inline PartHdr part_layout(void *ws, int64_t nb, int64_t nrec) {
    char *w = reinterpret_cast<char *>(ws);
    PartHdr p;
    p.bstart = reinterpret_cast<int *>(w);
    p.cursor = reinterpret_cast<int *>(w + align256((size_t)(nb + 1) * 4));
    p.tmp = reinterpret_cast<int2 *>(w + align256((size_t)(nb + 1) * 4) + align256((size_t)nb * 4));
    p.slot = reinterpret_cast<int *>(reinterpret_cast<char *>(p.tmp) + align256((size_t)nrec * sizeof(int2)));
    p.recs = reinterpret_cast<int2 *>(reinterpret_cast<char *>(p.slot) + align256((size_t)nrec * sizeof(int)));
    return p;
}

// pass 1 (FILL = false): per-bucket totals.  pass 2 (FILL = true): bucket-sorted records (src = row*64 + r, dst = j)
template <bool FILL>
__global__ __launch_bounds__(256) void part_pass(const int32_t *__restrict__ idx, const float *__restrict__ w, int64_t rows,
                                                 int K, int nb, int *__restrict__ gcount, int2 *__restrict__ recs,
                                                 int *__restrict__ slotmap) {
    extern __shared__ int lds[];                                 // hist[nb] (+ base[nb] when filling)
    int *hist = lds, *base = lds + nb;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
};
template <int H, bool FUSE>
__global__ __launch_bounds__(256) void edge_bwd_rows(const float *__restrict__ xp, int64_t rows, const int32_t *__restrict__ idx,
                                                     const float *__restrict__ val, const float *__restrict__ dval, int K,
                                                     int64_t row0, float t, int perturb, const int *__restrict__ slotmap,
                                                     float *__restrict__ coef, float *__restrict__ dxp, SoftkArgs sk) {
    constexpr int LPR = H / 4;                                   // lanes per neighbour
    constexpr int NPI = 64 / LPR;                                // neighbours per wave-instruction
    const int lane = threadIdx.x & 63, c4 = lane % LPR, slot = lane / LPR;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= rows) return;
    const int64_t gi = row0 + i;
    const int32_t jl = lane < K ? idx[i * K + lane] : -1;
    const float vl = lane < K ? val[i * K + lane] : 0.0f;
    float gl;
    if (FUSE) {                                                  // same arithmetic as softk_bwd_kernel (dgg_ell.hip), modes 0 / 1
        const int lc = lane < K ? lane : K - 1;
        float dw = sk.dA[i * K + lc];
        const bool live = lane < K && jl >= 0;
        if (sk.normalized) {
            const float rsi = sk.rs[gi];
            const float ai = __fdiv_rn(1.0f, c_sqrt(rsi)), aj = __fdiv_rn(1.0f, c_sqrt(sk.rs[jl >= 0 ? jl : gi]));
            float dai = sk.da[gi];
            if (sk.ahat_rows) {
                float rp = lane < K ? dw * sk.ahat_rows[i * K + lane] : 0.0f;
                rp = wave_sum_dpp(rp, lane);
                dai += rp * sqrtf(rsi);
            }
            const float drs = -0.5f * dai * ai / rsi;
            dw = dw * ai * aj + drs;
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
    constexpr int NBT = 2;                                       // batches (gathers in flight per lane) per iteration; 4 measured the same
    // NBT batches of NPI neighbours per iteration, all gathers issued UNCONDITIONALLY (inactive slots re-read the own row and
    // are masked arithmetically) before either is consumed: a predicated gather is a branch + s_waitcnt vmcnt(0), i.e. one
    // gather in flight per wavefront.
    for (int r0 = 0; r0 < K; r0 += NBT * NPI) {
        int32_t j[NBT];
        float g[NBT], v[NBT];
        bool act[NBT];
#pragma unroll
        for (int b = 0; b < NBT; b++) {
            const int r = r0 + b * NPI + slot;
            const int rr = r < 64 ? r : 63;
            j[b] = __shfl(jl, rr, 64);
            g[b] = __shfl(gl, rr, 64);
            v[b] = __shfl(vl, rr, 64);
            act[b] = r < K && j[b] >= 0 && g[b] != 0.0f;
        }
        bool any = false;
#pragma unroll
        for (int b = 0; b < NBT; b++) any = any || act[b];
        if (__ballot(any) == 0ull) continue;        // wave-uniform
        float4 xj[NBT];
#pragma unroll
        for (int b = 0; b < NBT; b++) xj[b] = *reinterpret_cast<const float4 *>(xp + (act[b] ? (int64_t)j[b] : gi) * H + 4 * c4);
#pragma unroll
        for (int b = 0; b < NBT; b++) {
            const float4 d = make_float4(xi.x - xj[b].x, xi.y - xj[b].y, xi.z - xj[b].z, xi.w - xj[b].w);   // 0 when inactive
            float d2 = d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
            if (LPR > 16) d2 += __uint_as_float(xor_shfl<16>(__float_as_uint(d2), lane));
            if (LPR > 8) d2 += __uint_as_float(xor_shfl<8>(__float_as_uint(d2), lane));
            if (LPR > 4) d2 += __uint_as_float(xor_shfl<4>(__float_as_uint(d2), lane));
            if (LPR > 2) d2 += __uint_as_float(xor_shfl<2>(__float_as_uint(d2), lane));
            d2 += __uint_as_float(xor_shfl<1>(__float_as_uint(d2), lane));
            float dd = 0.0f;
            if (act[b] && d2 != 0.0f) {                          // vector_norm backward at 0 is 0 (self loop)
                const float dist = sqrtf(d2);
                const float p = c_exp(t * dist);
                const float dp = perturb ? g[b] * v[b] / (p + 1e-8f) : g[b];
                dd = dp * t * p / dist;
            }
            acc.x += dd * d.x; acc.y += dd * d.y; acc.z += dd * d.z; acc.w += dd * d.w;
            // hand the coefficient to the lane that owns entry r (lane r): gather from the first lane of each slot
#pragma unroll
            for (int s2 = 0; s2 < NPI; s2++) {
                const float cs = bcast(dd, s2 * LPR);
                if (lane == r0 + b * NPI + s2) mycoef = cs;
            }
        }
    }
    // sum the NPI neighbour slots (lanes with equal c4)
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
        acc.x += __shfl_xor(acc.x, off, 64); acc.y += __shfl_xor(acc.y, off, 64);
        acc.z += __shfl_xor(acc.z, off, 64); acc.w += __shfl_xor(acc.w, off, 64);
    }
    if (slot == 0) *reinterpret_cast<float4 *>(dxp + gi * H + 4 * c4) = acc;
    if (lane < K) {
        const int sl = slotmap[i * K + lane];
        if (sl >= 0) coef[sl] = mycoef;                          // record order
    }
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
__global__ __launch_bounds__(256) void spmm_t_cols(const float *__restrict__ dY, int F, const int *__restrict__ bstart, int nb,
                                                   const int2 *__restrict__ recs, const float *__restrict__ a, int K,
                                                   float *__restrict__ dX) {
    constexpr int LPR = 16;
    const int lane = threadIdx.x & 63, c4 = lane % LPR, gbase = lane - c4;
    const int64_t gid = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LPR;
    const int f0 = blockIdx.y * 64 + 4 * c4;
    const int nnz = bstart[nb];
    const int64_t cbeg = gid * CH;
    if (cbeg >= nnz) return;
    const int cend = cbeg + CH < nnz ? (int)cbeg + CH : nnz;
    int cur = -1;
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const int shared_lo = cbeg > 0 ? recs[cbeg - 1].y : -1, shared_hi = cend < nnz ? recs[cend].y : -1;   // see edge_bwd_cols
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
    for (int eb = (int)cbeg; eb < cend; eb += LPR) {
        const int e = eb + c4;
        const int2 myrec = e < cend ? recs[e] : make_int2(0, -1);
        // record source = row*64 + r: the coefficient lives at a[row*K + r]
        const float mycf = e < cend ? a[(int64_t)(myrec.x >> 6) * K + (myrec.x & 63)] : 0.0f;
#pragma unroll
        for (int u0 = 0; u0 < LPR; u0 += 4) {
            int src[4], dst[4];
            float cf[4];
            float4 g[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                src[u] = __shfl(myrec.x, gbase + u0 + u, 64);
                dst[u] = __shfl(myrec.y, gbase + u0 + u, 64);
                cf[u] = __shfl(mycf, gbase + u0 + u, 64);
                g[u] = *reinterpret_cast<const float4 *>(dY + (int64_t)(src[u] >> 6) * F + f0);     // unconditional
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (dst[u] < 0) continue;
                if (dst[u] != cur) {
                    flush();
                    cur = dst[u];
                    acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
                acc.x = fmaf(cf[u], g[u].x, acc.x); acc.y = fmaf(cf[u], g[u].y, acc.y);
                acc.z = fmaf(cf[u], g[u].z, acc.z); acc.w = fmaf(cf[u], g[u].w, acc.w);
            }
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

// ---- normalisation backward: row side (da_i, per-entry coefficient), column side (bucket sums) -----------------------
__global__ __launch_bounds__(256) void norm_da_rows(const int32_t *__restrict__ idx, const float *__restrict__ w,
                                                    const float *__restrict__ rs, const float *__restrict__ dA, int64_t rows,
                                                    int K, int64_t row0, const int *__restrict__ slotmap,
                                                    float *__restrict__ coef, float *__restrict__ da) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
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
    if (nb > 16384 || K > 64) return 0;
    return align256((size_t)(nb + 1) * 4) + align256((size_t)nb * 4) + align256((size_t)rows * K * sizeof(int2)) +
           align256((size_t)rows * K * sizeof(int)) + (size_t)rows * K * sizeof(int2);
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
    const SoftkArgs sk{k, rs, dA, da, mode, normalized, dval, dk, ahat_rows};
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
    hipLaunchKernelGGL(spmm_t_cols, dim3((unsigned)((ngroups * 16 + 255) / 256), (unsigned)(F / 64)), dim3(256), 0, (hipStream_t)stream, dY,
                       F, p.bstart, (int)nb, p.recs, a, K, dX);
    return dgg_check_launch("ell_spmm_t_part");
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
