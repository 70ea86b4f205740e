// dgg_bf16.hip -- bf16 matrix-core path of the GCNII layer product (BASELINE configs[4]: "PPI multi-graph batched GCNII_DGG,
// bf16 fwd+bwd, MFMA feature-projection GEMM").
//
// Replaces, for the reduced-precision variant of GraphConvolution / DenseGraphConvolution (reference model.py:32-44, 65-77;
// train_ppi.py:43-44: hidden 2048, 9 layers, variant => support is [n, 4096]):
//     out = theta * (support @ weight) + (1 - theta) * r (+ input)            model.py:41-44
// and its autograd (d support = theta g W^T, d weight = theta support^T g).  Operands are rounded to bf16 (8 significant bits),
// products accumulate in fp32 on v_mfma_f32_32x32x16_bf16; the epilogue and every tensor outside the product stay fp32.
//
// One kernel shape serves all three products: C[M,N] = scale * A[M,K] B[N,K]^T with both operands K-CONTIGUOUS in bf16
// ("NT").  The operands are produced by dgg_pack_bf16 (fp32 -> bf16 copy, optionally transposed, rows zero-padded to a
// multiple of 64 so that the contraction never needs a tail):
//     forward     A = support       [n, K]        B = weight^T [out, K]
//     d support   A = g             [n, out]      B = weight   [K, out]   (as stored)
//     d weight    A = support^T     [K, n_pad]    B = g^T      [out, n_pad]
// Tile: 128 x 128 (or 64 x 128), K in steps of 64; 512 threads = 8 wavefronts of 64 x 32 in the ring loop (two per SIMD), 256 threads =
// 4 wavefronts of 64 x 64 otherwise.  Two main loops:
//   RING (default)  the operands go global -> LDS directly (global_load_lds_dwordx4, no staging registers) into a ring of FOUR
//                   stages, three K-steps in flight ahead of the one being multiplied: a K-step is 16 MFMAs per wavefront
//                   (~0.3 us) and one workgroup per CU is all the 224-576 tiles of a PPI graph give, so a one-step-ahead
//                   register prefetch paid a memory round trip per step (1.3 us per step on the forward product).  An LDS-DMA
//                   instruction writes its wavefront's 64 x 16 B contiguously, so the stage is an UNPADDED [row][8 chunks] image
//                   and the bank spread comes from an XOR swizzle applied to the SOURCE chunk and again on the read:
//                   slot = chunk ^ ((row >> 1) & 7) -- conflict-free for the lane groups of ds_read_b128.  The fragment reads of
//                   sub-step s + 1 are issued before the MFMAs of sub-step s; ONE barrier per K-step, placed mid-step.
//   staged          double-buffered LDS image filled through registers (row stride 144 B), the next step's loads in flight
//                   during the current step's MFMAs (DGG_BF16_RING=0).
#include "dgg_common.h"
#include "dgg_api_internal.h"

#include <cstdlib>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int BN = 128, BK = 64, LDS_STRIDE = BK + 8;   // bf16 elements per staged row (144 bytes)

struct GcniiEpi {
    const float *hi, *h0, *inp;    // EPI 1: r = h0 ? (1 - alpha) hi + alpha h0 : hi;  out = theta C + (1 - theta) r (+ inp)
    float theta, alpha;
    // EPI 2 / 3 (one half of d support of the variant layer, [d hi | d h0] = theta g W^T + [c1 | c2] g; a launch per half, the half's
    // rows of W as the B operand): out = scale * acc + c1 * g (EPI 3: + the value C holds -- d h0 summed over the stack's layers
    // without an add pass per layer); fp32 to C (nullable in EPI 2), bf16 to outb (nullable).  g, C, outb are [M, N].
    const float *g;
    float c1;
    // fused GCNII stack (EPI 1): the layer's activation and the NEXT layer's dropout in the same store --
    // out = keep(e) ? relu(.) / (1 - p) : 0 with the counter-based mask of dgg_common.h (drop_keep); relu 0: neither
    int relu;
    uint32_t drop_thr24, s0, s1;
    float drop_scale;
    // bf16 copy of the output for the kernels that GATHER it afterwards (the next layer's aggregation and the SDDMM read the
    // activation, the transposed aggregation reads d hi): EPI 1: of `out` [n,N]; EPI 2: of d hi.  Nullable.
    __bf16 *outb;
};

// AM: 32-row MFMA blocks per wavefront (2: 128-row tiles; 1: 64-row tiles, used when 128-row tiles would not fill the chip twice)
// A2 / ksplit > 0: the contraction range [ksplit, K) of the A operand comes from a SECOND matrix A2 [M, K - ksplit] (A is then
// [M, ksplit]): the variant layer's support cat[hi, h0] (model.py:37-40) is never formed.
// A2 / ksplit < 0: the ROWS [-ksplit, M) of the A operand come from A2 [M + ksplit, K] (the weight gradient [hi | h0]^T g as one
// product over the two transposed halves).  Splits are multiples of the tile (64 in K, 128 in M).
#ifndef DGG_BF16_ABL
#define DGG_BF16_ABL 0
#endif
#ifndef DGG_BF16_NST
#define DGG_BF16_NST 4
#endif
// One LDS-DMA wave-instruction: lane l's 16 bytes at `gsrc` go to LDS byte address lds_dst + 16 l (lds_dst wave-uniform, in M0).
// Written as inline assembly, not __builtin_amdgcn_global_load_lds: the compiler books the builtin as a possible LDS access of a
// FLAT instruction and from then on waits lgkmcnt(0) before every use of a ds_read result -- the fragment prefetch below would
// be serialised again.  The compiler does not count these loads: the s_waitcnt vmcnt(N) that retire them are written out below.
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// WC: wavefronts across the tile's 128 columns (2: four wavefronts of 32 AM x 64, 256 threads; 4: eight of 32 AM x 32, 512 threads --
// two per SIMD, so that one's fragment reads and barrier waits run under the other's MFMAs; ring only)
template <int EPI, int AM, bool RING, int WC = 2>
__global__ __launch_bounds__(128 * WC) void gemm_nt_bf16(const __bf16 *__restrict__ A, const __bf16 *__restrict__ B, int M, int N, int K,
                                                         float scale, float *__restrict__ C, GcniiEpi ep, const __bf16 *__restrict__ A2, int ksplit) {
    static_assert(WC == 2 || (WC == 4 && RING), "eight wavefronts: ring loop only");
    constexpr int BM = 64 * AM, NB = 4 / WC, NW = 2 * WC;
    constexpr int NST = AM >= 4 ? 3 : DGG_BF16_NST, STAGE = (BM + BN) * 128;   // ring: stages of [BM + BN rows][8 chunks of 16 B] (256-row tiles: 3 x 48 KB)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[RING ? NST * STAGE : 2 * (BM + BN) * LDS_STRIDE * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = dgg::wave_id(), li = lane & 31, hh = lane >> 5;
    const int wr = wave / WC, wc = wave % WC;
    const int m0 = blockIdx.y * (64 * AM), n0 = blockIdx.x * BN;
    f32x16 acc[AM][NB];
#pragma unroll
    for (int a = 0; a < AM; a++)
#pragma unroll
        for (int b = 0; b < NB; b++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[a][b][q] = 0.0f;
    // fp32 terms of the fused epilogues, fetched BEFORE the main loop (rows / columns beyond the matrix clamped: no branches):
    // one workgroup per CU leaves nothing to overlap an epilogue's loads with, and fetched after the loop they cost 20 us per tile
    float add[EPI != 0 ? AM : 1][16][NB];
    auto epi_terms = [&]() {
      if constexpr (EPI != 0) {
        const float omt = 1.0f - ep.theta, oma = 1.0f - ep.alpha;
        const float *__restrict__ e_hi = ep.hi, *__restrict__ e_h0 = ep.h0, *__restrict__ e_inp = ep.inp, *__restrict__ e_g = ep.g;
#pragma unroll
        for (int a = 0; a < AM; a++) {
            const int rbase = m0 + wr * 32 * AM + a * 32 + 4 * hh;
#pragma unroll
            for (int b = 0; b < NB; b++) {
                int col = n0 + wc * 32 * NB + b * 32 + li;
                col = col < N ? col : N - 1;
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    int row = rbase + (q & 3) + 8 * (q >> 2);
                    row = row < M ? row : M - 1;
                    if constexpr (EPI >= 2) {
                        add[a][q][b] = ep.c1 * e_g[(int64_t)row * N + col];
                        if constexpr (EPI == 3) add[a][q][b] += C[(int64_t)row * N + col];
                    } else {
                        const int64_t o = (int64_t)row * N + col;
                        const float r = e_h0 ? oma * e_hi[o] + ep.alpha * e_h0[o] : e_hi[o];
                        add[a][q][b] = omt * r + (e_inp ? e_inp[o] : 0.0f);
                    }
                }
            }
        }
        // (pins the terms HERE: left alone the compiler sinks every load to its use after the main loop, one memory round trip per element)
#pragma unroll
        for (int a = 0; a < AM; a++)
#pragma unroll
            for (int q = 0; q < 16; q++)
#pragma unroll
                for (int b = 0; b < NB; b++) asm volatile("" ::"v"(add[a][q][b]));
      }
    };
    if constexpr (RING) {
        // piece = 8 rows x 128 B = one LDS-DMA wave-instruction; wavefront w fills the pieces w, w + 4, ... of either operand.
        // lane -> (row lane / 8 of the piece, slot lane % 8); it fetches the chunk that belongs in that slot: slot ^ ((row >> 1) & 7).
        // Rows beyond M / N are clamped to a valid row: they only reach accumulators that are never stored.
        constexpr int PA = BM / 8 / NW, PB = BN / 8 / NW;       // pieces per wavefront and stage
        const bool msec = A2 != nullptr && ksplit < 0 && m0 >= -ksplit;             // (block-uniform: the splits are tile multiples)
        int rowA[PA], rowB[PB], chA[PA], chB[PB];
#pragma unroll
        for (int q = 0; q < PA; q++) {
            const int r = (wave + NW * q) * 8 + (lane >> 3);
            int gr = m0 + r < M ? m0 + r : M - 1;
            if (msec) gr -= -ksplit;
            rowA[q] = gr;
            chA[q] = ((lane & 7) ^ ((r >> 1) & 7)) * 8;
        }
#pragma unroll
        for (int q = 0; q < PB; q++) {
            const int r = (wave + NW * q) * 8 + (lane >> 3);
            rowB[q] = n0 + r < N ? n0 + r : N - 1;
            chB[q] = ((lane & 7) ^ ((r >> 1) & 7)) * 8;
        }
        const unsigned lds0 = (unsigned)(uintptr_t)smem, wave_u = (unsigned)__builtin_amdgcn_readfirstlane(wave);
        auto issue = [&](int kt, int st) {
            const int k0 = kt * BK;
            const bool ksec = A2 != nullptr && ksplit > 0 && k0 >= ksplit;
            const __bf16 *Ab = (ksec || msec) ? A2 : A;
            const int lda = (A2 && ksplit > 0) ? (ksec ? K - ksplit : ksplit) : K, ka = ksec ? k0 - ksplit : k0;
            const unsigned sa = lds0 + (unsigned)(st * STAGE) + wave_u * 1024u, sb = sa + BM * 128;
#pragma unroll
            for (int q = 0; q < PA; q++) glds16(Ab + (int64_t)rowA[q] * lda + ka + chA[q], sa + q * (NW * 1024));
#pragma unroll
            for (int q = 0; q < PB; q++) glds16(B + (int64_t)rowB[q] * K + k0 + chB[q], sb + q * (NW * 1024));
        };
        // fragment reads: lane (li, hh) of K-sub-step ks wants chunk 2 ks + hh of row R = 32 x + li; (R >> 1) & 7 = (li >> 1) & 7
        const int swz = (li >> 1) & 7;
        int ko[BK / 16];
#pragma unroll
        for (int ks = 0; ks < BK / 16; ks++) ko[ks] = ((2 * ks + hh) ^ swz) * 16;
        const int arow = (wr * 32 * AM + li) * 128, brow = BM * 128 + (wc * 32 * NB + li) * 128;
        const int nk = K / BK;
        static_assert(NST == 3 || NST == 4, "ring depth");
        static_assert(BK / 16 == 4, "the K-step is four MFMA sub-steps");
        // fragments of one sub-step: AM row blocks of A, two column blocks of B.  Two sets: the reads of sub-step s + 1 are issued
        // before the MFMAs of sub-step s (with one set the compiler could only re-issue a read after the MFMAs that consume the
        // register, and every sub-step exposed an LDS round trip: 0.58 us per K-step against 0.3 us of matrix time).
        struct Frags { bf16x8 a[AM], b[NB]; };
        auto rd = [&](Frags &f, const unsigned char *sb, int ks) {
#pragma unroll
            for (int b = 0; b < NB; b++) f.b[b] = *reinterpret_cast<const bf16x8 *>(sb + brow + b * 32 * 128 + ko[ks]);
#pragma unroll
            for (int a = 0; a < AM; a++) f.a[a] = *reinterpret_cast<const bf16x8 *>(sb + arow + a * 32 * 128 + ko[ks]);
        };
        auto mm = [&](const Frags &f) {
#if DGG_BF16_ABL != 1
#pragma unroll
            for (int a = 0; a < AM; a++)
#pragma unroll
                for (int b = 0; b < NB; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[a], f.b[b], acc[a][b], 0, 0, 0);
#endif
        };
#pragma unroll
        for (int q = 0; q < NST - 1; q++)
            if (q < nk) issue(q, q);
        epi_terms();                                            // (behind the first stages' loads: one memory round trip for both)
        // stage 0 has landed once all but the loads of the newer stages of every wavefront are done; the barrier publishes it
        if (nk >= NST - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * (PA + PB)) : "memory");
        else if (NST > 3 && nk == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA + PB) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        Frags f0, f1;
        rd(f0, smem, 0);
        int st = 0;
        for (int kt = 0; kt < nk; kt++) {
            const unsigned char *sb = smem + st * STAGE;
            const int stn = st == NST - 1 ? 0 : st + 1;
            rd(f1, sb, 1);
            mm(f0);
            rd(f0, sb, 2);
            mm(f1);
            // MID-step: publish stage kt + 1 (every wavefront waits for its own pieces, then the barrier) so that its first
            // fragments can be fetched during the last sub-step below; every wavefront is also through with stage kt - 1, whose
            // slot the loads of stage kt + NST - 1 now overwrite
            if (kt + 1 < nk) {
                if (NST > 3 && kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA + PB) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
#if DGG_BF16_ABL != 2
                if (kt + NST - 1 < nk) issue(kt + NST - 1, st == 0 ? NST - 1 : st - 1);
#endif
            }
            rd(f1, sb, 3);
            mm(f0);
            if (kt + 1 < nk) rd(f0, smem + stn * STAGE, 0);
            mm(f1);
            st = stn;
        }
    } else {
    __bf16 (*As)[BM * LDS_STRIDE] = reinterpret_cast<__bf16 (*)[BM * LDS_STRIDE]>(smem);
    __bf16 (*Bs)[BN * LDS_STRIDE] = reinterpret_cast<__bf16 (*)[BN * LDS_STRIDE]>(smem + 2 * BM * LDS_STRIDE * 2);
    uint4 ra[2 * AM], rb[4];
    // 128 rows x 8 chunks of 16 bytes per operand and K-step: 4 chunks per thread (rows beyond M / N read row 0 and are zeroed)
    auto load = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int c = tid + q * 256, row = c >> 3, kc = c & 7;
            const bool vb = n0 + row < N;
            rb[q] = *reinterpret_cast<const uint4 *>(B + (int64_t)(vb ? n0 + row : 0) * K + k0 + kc * 8);
            if (!vb) rb[q] = make_uint4(0, 0, 0, 0);
            if (q < 2 * AM) {
                const bool va = m0 + row < M;
                const bool ksec = A2 != nullptr && ksplit > 0 && k0 >= ksplit;       // (block-uniform: the splits are tile multiples)
                const bool msec = A2 != nullptr && ksplit < 0 && m0 >= -ksplit;
                const __bf16 *Ab = (ksec || msec) ? A2 : A;
                const int lda = (A2 && ksplit > 0) ? (ksec ? K - ksplit : ksplit) : K, ka = ksec ? k0 - ksplit : k0;
                const int ra_row = (va ? m0 + row : (msec ? -ksplit : 0)) - (msec ? -ksplit : 0);
                ra[q] = *reinterpret_cast<const uint4 *>(Ab + (int64_t)ra_row * lda + ka + kc * 8);
                if (!va) ra[q] = make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int c = tid + q * 256, row = c >> 3, kc = c & 7;
            if (q < 2 * AM) *reinterpret_cast<uint4 *>(&As[buf][row * LDS_STRIDE + kc * 8]) = ra[q];
            *reinterpret_cast<uint4 *>(&Bs[buf][row * LDS_STRIDE + kc * 8]) = rb[q];
        }
    };
    load(0);
    epi_terms();
    store(0);
    __syncthreads();
    const int nk = K / BK;
    for (int kt = 0; kt < nk; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load((kt + 1) * BK);
        const __bf16 *as = &As[buf][(wr * 32 * AM + li) * LDS_STRIDE + hh * 8];
        const __bf16 *bs = &Bs[buf][(wc * 64 + li) * LDS_STRIDE + hh * 8];
#pragma unroll
        for (int ks = 0; ks < BK / 16; ks++) {
            const bf16x8 b0 = *reinterpret_cast<const bf16x8 *>(bs + ks * 16);
            const bf16x8 b1 = *reinterpret_cast<const bf16x8 *>(bs + 32 * LDS_STRIDE + ks * 16);
#pragma unroll
            for (int a = 0; a < AM; a++) {
                const bf16x8 af = *reinterpret_cast<const bf16x8 *>(as + a * 32 * LDS_STRIDE + ks * 16);
                acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b0, acc[a][0], 0, 0, 0);
                acc[a][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b1, acc[a][1], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) store(buf ^ 1);
        __syncthreads();
    }
    }
    // epilogue: accumulator register q of block (a, b) holds C[row = (q & 3) + 8 (q >> 2) + 4 hh][col = li] of the 32 x 32 block;
    // the fp32 terms of the fused epilogues were fetched before the main loop (epi_terms).  A tile that lies inside the matrix -- all
    // but the last row of tiles -- stores without per-element bounds tests (224 branches in the other form).
    auto store_tile = [&](auto fullc) {
        constexpr bool FULL = decltype(fullc)::value;
#pragma unroll
        for (int a = 0; a < AM; a++) {
            const int rbase = m0 + wr * 32 * AM + a * 32 + 4 * hh;
#pragma unroll
            for (int b = 0; b < NB; b++) {
                const int col = n0 + wc * 32 * NB + b * 32 + li;
                if (!FULL && col >= N) continue;
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const int row = rbase + (q & 3) + 8 * (q >> 2);
                    if (!FULL && row >= M) continue;
                    const float v = scale * acc[a][b][q];
                    if constexpr (EPI >= 2) {
                        const float o_ = v + add[a][q][b];
                        if (EPI == 3 || C) C[(int64_t)row * N + col] = o_;
                        if (ep.outb) ep.outb[(int64_t)row * N + col] = (__bf16)o_;
                    } else if constexpr (EPI == 1) {
                        float o_ = ep.theta * v + add[a][q][b];
                        if (ep.relu) {
                            o_ = o_ > 0.0f ? o_ : 0.0f;
                            if (ep.drop_thr24)
                                o_ = dgg::drop_keep(ep.s0, ep.s1, (uint32_t)((int64_t)row * N + col), ep.drop_thr24) ? o_ * ep.drop_scale : 0.0f;
                        }
                        C[(int64_t)row * N + col] = o_;
                        if (ep.outb) ep.outb[(int64_t)row * N + col] = (__bf16)o_;
                    } else {
                        C[(int64_t)row * N + col] = v;
                    }
                }
            }
        }
    };
    if (m0 + BM <= M && n0 + BN <= N) store_tile(std::true_type{});
    else store_tile(std::false_type{});
}

// fp32 [R, Cc] -> bf16.  transpose 0: dst [R][ld] (ld >= Cc, columns Cc..ld-1 zero);  transpose 1: dst [Cc][ld] (ld >= R, zero padded)
__global__ __launch_bounds__(256) void pack_bf16_kernel(const float *__restrict__ src, int R, int Cc, int transpose, __bf16 *__restrict__ dst,
                                                        int ld) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    if (!transpose) {
        const int c = blockIdx.x * 32 + tx;
        for (int r = blockIdx.y * 32 + ty; r < R && r < blockIdx.y * 32 + 32; r += 8)
            if (c < ld) dst[(int64_t)r * ld + c] = (__bf16)(c < Cc ? src[(int64_t)r * Cc + c] : 0.0f);
        return;
    }
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < Cc) ? src[(int64_t)r * Cc + c] : 0.0f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;                       // dst row = source column
        if (c < Cc && r < ld) dst[(int64_t)c * ld + r] = (__bf16)tile[tx][i];
    }
}

// both layouts of one fp32 matrix in one pass: dst [R][ld] and dstT [Cc][ldT] (zero padded as pack_bf16_kernel does)
__global__ __launch_bounds__(256) void pack_bf16_both_kernel(const float *__restrict__ src, int R, int Cc, __bf16 *__restrict__ dst, int ld,
                                                             __bf16 *__restrict__ dstT, int ldT) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        const float v = (r < R && c < Cc) ? src[(int64_t)r * Cc + c] : 0.0f;
        tile[i][tx] = v;
        if (r < R && c < ld) dst[(int64_t)r * ld + c] = (__bf16)v;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;                       // dstT row = source column
        if (c < Cc && r < ldT) dstT[(int64_t)c * ldT + r] = (__bf16)tile[tx][i];
    }
}

// The packs on 64 x 64 tiles with 16-byte accesses (Cc a multiple of 4, ld of 4, ldT of 16, aligned pointers; the 32 x 32 kernels
// above, one 4-byte load and one 2-byte store per thread and element, moved a 4096 x 2048 weight at 3.2 TB/s): a thread loads float4s
// (256 B per row segment), stores 4 bf16 of the plain layout and, through an LDS tile, 16 consecutive bf16 of one transposed row.
template <bool PLAIN, bool TRANS>
__global__ __launch_bounds__(256) void pack_bf16_t64(const float *__restrict__ src, int R, int Cc, __bf16 *__restrict__ dst, int ld,
                                                     __bf16 *__restrict__ dstT, int ldT) {
    __shared__ float tile[TRANS ? 64 : 1][65];
    const int tid = threadIdx.x, c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int rl = (tid >> 4) + 16 * i, cl = (tid & 15) * 4, r = r0 + rl, c = c0 + cl;
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (r < R && c < Cc) v = *reinterpret_cast<const float4 *>(src + (int64_t)r * Cc + c);
        if constexpr (TRANS) { tile[rl][cl] = v.x; tile[rl][cl + 1] = v.y; tile[rl][cl + 2] = v.z; tile[rl][cl + 3] = v.w; }
        if constexpr (PLAIN) {
            if (r < R && c < ld) {
                union { __bf16 h[4]; uint2 u; } o;
                o.h[0] = (__bf16)v.x; o.h[1] = (__bf16)v.y; o.h[2] = (__bf16)v.z; o.h[3] = (__bf16)v.w;
                *reinterpret_cast<uint2 *>(dst + (int64_t)r * ld + c) = o.u;
            }
        }
    }
    if constexpr (TRANS) {
        __syncthreads();
        const int cT = c0 + (tid >> 2), rb = (tid & 3) * 16;      // transposed row = source column; 16 consecutive source rows
        if (cT < Cc && r0 + rb < ldT) {
            union { __bf16 h[16]; uint4 u[2]; } o;
#pragma unroll
            for (int k = 0; k < 16; k++) o.h[k] = (__bf16)tile[rb + k][tid >> 2];
            uint4 *d = reinterpret_cast<uint4 *>(dstT + (int64_t)cT * ldT + r0 + rb);
            d[0] = o.u[0];
            d[1] = o.u[1];
        }
    }
}
__host__ inline bool pack_fast_ok(const void *src, int64_t Cc, const void *dst, int64_t ld, const void *dstT, int64_t ldT) {
    return Cc % 4 == 0 && (reinterpret_cast<uintptr_t>(src) % 16) == 0 && (!dst || (ld % 4 == 0 && (reinterpret_cast<uintptr_t>(dst) % 8) == 0)) &&
           (!dstT || (ldT % 16 == 0 && (reinterpret_cast<uintptr_t>(dstT) % 16) == 0)) && !getenv("DGG_PACK_SLOW");
}

// Backward through a stack layer's relu + dropout, and the operand packs of the two products that follow, in ONE pass over the
// gradient: g = g_in * (xd != 0 ? scale : 0) (xd = the layer's stored output: zero where the ReLU or the dropout zeroed it), written as
// fp32 [n,F], as bf16 [n,F] (A operand of [d hi | d h0]) and as bf16 transposed [F, ldT] (B operand of the weight gradient; columns
// n..ldT-1 zero).  Replaces threshold_backward + masked_scale + two pack_bf16 launches (+ the residual add, which the caller gets
// for free by accumulating A^T d hi into a copy of g).
__global__ __launch_bounds__(256) void gcnii_gout_pack_kernel(const float *__restrict__ gin, const float *__restrict__ xd, float scale, int n,
                                                             int F, float *__restrict__ g, __bf16 *__restrict__ Gp, __bf16 *__restrict__ GT,
                                                             int ldT, float *__restrict__ g2) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        float v = 0.0f;
        if (r < n && c < F) {
            const int64_t o = (int64_t)r * F + c;
            v = xd[o] != 0.0f ? gin[o] * scale : 0.0f;
            g[o] = v;
            if (g2) g2[o] = v;                                   // (a second fp32 copy: the buffer A^T d hi accumulates into -- the residual)
            Gp[o] = (__bf16)v;
        }
        tile[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;                       // GT row = feature, column = node
        if (c < F && r < ldT) GT[(int64_t)c * ldT + r] = (__bf16)tile[tx][i];
    }
}

// gcnii_gout_pack_kernel on 64 x 64 tiles with 16-byte accesses (F and ldT multiples of 64, 16-byte aligned pointers), as pack_bf16_t64
__global__ __launch_bounds__(256) void gcnii_gout_pack_t64(const float *__restrict__ gin, const float *__restrict__ xd, float scale, int n, int F,
                                                           float *__restrict__ g, __bf16 *__restrict__ Gp, __bf16 *__restrict__ GT, int ldT,
                                                           float *__restrict__ g2) {
    __shared__ float tile[64][65];
    const int tid = threadIdx.x, c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int rl = (tid >> 4) + 16 * i, cl = (tid & 15) * 4, r = r0 + rl, c = c0 + cl;
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (r < n) {
            const int64_t o = (int64_t)r * F + c;
            const float4 gi = *reinterpret_cast<const float4 *>(gin + o), x = *reinterpret_cast<const float4 *>(xd + o);
            v = make_float4(x.x != 0.0f ? gi.x * scale : 0.0f, x.y != 0.0f ? gi.y * scale : 0.0f, x.z != 0.0f ? gi.z * scale : 0.0f,
                            x.w != 0.0f ? gi.w * scale : 0.0f);
            *reinterpret_cast<float4 *>(g + o) = v;
            if (g2) *reinterpret_cast<float4 *>(g2 + o) = v;
            union { __bf16 h[4]; uint2 u; } b;
            b.h[0] = (__bf16)v.x; b.h[1] = (__bf16)v.y; b.h[2] = (__bf16)v.z; b.h[3] = (__bf16)v.w;
            *reinterpret_cast<uint2 *>(Gp + o) = b.u;
        }
        tile[rl][cl] = v.x; tile[rl][cl + 1] = v.y; tile[rl][cl + 2] = v.z; tile[rl][cl + 3] = v.w;
    }
    __syncthreads();
    const int cT = c0 + (tid >> 2), rb = (tid & 3) * 16;          // GT row = feature; 16 consecutive nodes (zeros beyond n)
    if (r0 + rb < ldT) {
        union { __bf16 h[16]; uint4 u[2]; } o;
#pragma unroll
        for (int k = 0; k < 16; k++) o.h[k] = (__bf16)tile[rb + k][tid >> 2];
        uint4 *d = reinterpret_cast<uint4 *>(GT + (int64_t)cT * ldT + r0 + rb);
        d[0] = o.u[0];
        d[1] = o.u[1];
    }
}

int launch_gemm(const __bf16 *A, const __bf16 *B, int M, int N, int K, float scale, float *C, const GcniiEpi *ep, hipStream_t st,
                int epi = -1, const __bf16 *A2 = nullptr, int ksplit = 0) {
    if (K % BK != 0 || K < BK) return dgg_set_error(DGG_ERR_UNSUPPORTED, "gemm_nt_bf16: the contraction length must be a multiple of 64 (pack with padding)");
    if (A2 && ksplit > 0 && (ksplit % BK != 0 || ksplit >= K)) return dgg_set_error(DGG_ERR_UNSUPPORTED, "gemm_nt_bf16: the operand split must be a multiple of 64 inside (0, K)");
    if (A2 && ksplit <= 0 && (ksplit == 0 || (-ksplit) % 128 != 0 || -ksplit >= M)) return dgg_set_error(DGG_ERR_UNSUPPORTED, "gemm_nt_bf16: the row split must be a multiple of 128 inside (0, M)");
    if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(A2)) % 16) return dgg_set_error(DGG_ERR_ARG, "gemm_nt_bf16: operands must be 16-byte aligned");
    if (M == 0 || N == 0) return 0;
    if (epi < 0) epi = ep ? 1 : 0;
    // 128-row or 64-row tiles: whichever needs less time in ROUNDS of one workgroup per CU (the ring's LDS admits one).  A 64-row tile takes
    // 0.68 of a 128-row tile's time (measured: it carries 1.5x the LDS traffic per MFMA), so it pays only while it saves a round:
    // n = 591 (PPI's smallest graph): 160 small tiles in one round against 80 large ones; n = 1300: 336 small tiles are TWO rounds (58 us)
    // where 176 large ones are one (45 us) -- the former rule (small below 192 large tiles) chose the 58.
    const int64_t nt = (N + BN - 1) / BN, big = nt * ((M + 127) / 128), sml = nt * ((M + 63) / 64);
    bool small = (double)((sml + 255) / 256) * 0.68 < (double)((big + 255) / 256);
    { const char *e = getenv("DGG_BF16_TILE"); if (e) small = atoi(e) == 64 ? true : (atoi(e) == 128 ? false : small); }
    dim3 grid((unsigned)((N + BN - 1) / BN), (unsigned)((M + (small ? 63 : 127)) / (small ? 64 : 128)));
    const GcniiEpi e0 = ep ? *ep : GcniiEpi{};
    bool ring = true;
    { const char *e = getenv("DGG_BF16_RING"); if (e && atoi(e) == 0) ring = false; }
    bool eight = ring;                                           // eight wavefronts per workgroup (ring loop only)
    { const char *e = getenv("DGG_BF16_WAVES"); if (e && atoi(e) == 4) eight = false; }
    // 256-row tiles (a wavefront owns 128 x 32: five fragment reads per four MFMAs instead of three per two, half the barriers and 3/4 of the
    // operand traffic per flop) for the plain products that make at least two full rounds of 128-row tiles: the weight gradients (M = 2F)
    bool tall = !small && epi == 0 && eight && M % 256 == 0 && big >= 512 && !getenv("DGG_BF16_TILE");
    { const char *e = getenv("DGG_BF16_TALL"); if (e && atoi(e) == 0) tall = false; }
    if (tall) {
        grid.y = (unsigned)(M / 256);
        hipLaunchKernelGGL((gemm_nt_bf16<0, 4, true, 4>), grid, dim3(512), 0, st, A, B, M, N, K, scale, C, e0, A2, ksplit);
        return dgg_check_launch("gemm_nt_bf16");
    }
#define DGG_BF16_LAUNCH2(E, AMV, RG, WCV) hipLaunchKernelGGL((gemm_nt_bf16<E, AMV, RG, WCV>), grid, dim3(128 * WCV), 0, st, A, B, M, N, K, scale, C, e0, A2, ksplit)
#define DGG_BF16_LAUNCH(E, AMV) do { if (eight) DGG_BF16_LAUNCH2(E, AMV, true, 4); else if (ring) DGG_BF16_LAUNCH2(E, AMV, true, 2); else DGG_BF16_LAUNCH2(E, AMV, false, 2); } while (0)
    if (epi == 3) { if (small) DGG_BF16_LAUNCH(3, 1); else DGG_BF16_LAUNCH(3, 2); }
    else if (epi == 2) { if (small) DGG_BF16_LAUNCH(2, 1); else DGG_BF16_LAUNCH(2, 2); }
    else if (epi == 1) { if (small) DGG_BF16_LAUNCH(1, 1); else DGG_BF16_LAUNCH(1, 2); }
    else { if (small) DGG_BF16_LAUNCH(0, 1); else DGG_BF16_LAUNCH(0, 2); }
#undef DGG_BF16_LAUNCH
#undef DGG_BF16_LAUNCH2
    return dgg_check_launch("gemm_nt_bf16");
}

}  // namespace

extern "C" {

int dgg_pack_bf16(const float *src, int64_t R, int64_t Cc, int transpose, void *dst, int64_t ld, void *stream) {
    if (R <= 0 || Cc <= 0) return 0;
    if (ld < (transpose ? R : Cc)) return dgg_set_error(DGG_ERR_ARG, "pack_bf16: leading dimension smaller than the row length");
    hipStream_t st_ = (hipStream_t)stream;
    if (transpose && pack_fast_ok(src, Cc, nullptr, 0, dst, ld)) {
        hipLaunchKernelGGL((pack_bf16_t64<false, true>), dim3((unsigned)((Cc + 63) / 64), (unsigned)((ld + 63) / 64)), dim3(256), 0, st_, src, (int)R,
                           (int)Cc, nullptr, 0, reinterpret_cast<__bf16 *>(dst), (int)ld);
        return dgg_check_launch("pack_bf16");
    }
    if (!transpose && pack_fast_ok(src, Cc, dst, ld, nullptr, 0)) {
        hipLaunchKernelGGL((pack_bf16_t64<true, false>), dim3((unsigned)((ld + 63) / 64), (unsigned)((R + 63) / 64)), dim3(256), 0, st_, src, (int)R,
                           (int)Cc, reinterpret_cast<__bf16 *>(dst), (int)ld, nullptr, 0);
        return dgg_check_launch("pack_bf16");
    }
    const int64_t cols = transpose ? Cc : ld;                    // plain copy also writes the zero padding columns
    const dim3 grid((unsigned)((cols + 31) / 32), (unsigned)(((transpose ? (ld > R ? ld : R) : R) + 31) / 32));
    hipLaunchKernelGGL(pack_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, (int)R, (int)Cc, transpose,
                       reinterpret_cast<__bf16 *>(dst), (int)ld);
    return dgg_check_launch("pack_bf16");
}

// dgg_pack_bf16 in both layouts from ONE read of src: dst [R][ld] (ld >= Cc) and dstT [Cc][ldT] (ldT >= R), zero padded
int dgg_pack_bf16_both(const float *src, int64_t R, int64_t Cc, void *dst, int64_t ld, void *dstT, int64_t ldT, void *stream) {
    if (R <= 0 || Cc <= 0) return 0;
    if (ld < Cc || ldT < R) return dgg_set_error(DGG_ERR_ARG, "pack_bf16_both: leading dimension smaller than the row length");
    if (pack_fast_ok(src, Cc, dst, ld, dstT, ldT)) {
        hipLaunchKernelGGL((pack_bf16_t64<true, true>), dim3((unsigned)((ld + 63) / 64), (unsigned)((ldT + 63) / 64)), dim3(256), 0, (hipStream_t)stream,
                           src, (int)R, (int)Cc, reinterpret_cast<__bf16 *>(dst), (int)ld, reinterpret_cast<__bf16 *>(dstT), (int)ldT);
        return dgg_check_launch("pack_bf16_both");
    }
    const dim3 grid((unsigned)((ld + 31) / 32), (unsigned)((ldT + 31) / 32));
    hipLaunchKernelGGL(pack_bf16_both_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, (int)R, (int)Cc, reinterpret_cast<__bf16 *>(dst),
                       (int)ld, reinterpret_cast<__bf16 *>(dstT), (int)ldT);
    return dgg_check_launch("pack_bf16_both");
}

// C[M,N] (fp32) = scale * A[M,K] B[N,K]^T, A and B bf16 with K contiguous (K a multiple of 64)
int dgg_gemm_nt_bf16(const void *A, const void *B, int64_t M, int64_t N, int64_t K, float scale, float *C, void *stream) {
    return launch_gemm(reinterpret_cast<const __bf16 *>(A), reinterpret_cast<const __bf16 *>(B), (int)M, (int)N, (int)K, scale, C, nullptr,
                       (hipStream_t)stream);
}

// GCNII layer (model.py:36-44) with the product on the bf16 matrix cores and the epilogue fused:
//   out[n,F] = theta * (S Wt^T) + (1 - theta) * r (+ inp),  r = h0 ? (1 - alpha) hi + alpha h0 : hi
// S bf16 [n,K], Wt bf16 [F,K] (the weight transposed), hi / h0 / inp fp32 [n,F] (h0, inp nullable)
int dgg_gcnii_gemm_bf16(const void *S, const void *Wt, int64_t n, int64_t F, int64_t K, const float *hi, const float *h0, const float *inp,
                        float theta, float alpha, float *out, void *stream) {
    if (!hi) return dgg_set_error(DGG_ERR_ARG, "gcnii_gemm_bf16: hi is required");
    const GcniiEpi ep{hi, h0, inp, theta, alpha};
    return launch_gemm(reinterpret_cast<const __bf16 *>(S), reinterpret_cast<const __bf16 *>(Wt), (int)n, (int)F, (int)K, 1.0f, out, &ep,
                       (hipStream_t)stream);
}

// The VARIANT layer (support = cat[hi, h0], model.py:37-40) without forming the concatenation: S1 = bf16(hi) [n,F1], S2 = bf16(h0)
// [n,K-F1] are the two halves of the A operand (F1 a multiple of 64); otherwise as dgg_gcnii_gemm_bf16
int dgg_gcnii_gemm_bf16_split(const void *S1, const void *S2, const void *Wt, int64_t n, int64_t F, int64_t K, int64_t F1, const float *hi,
                              const float *h0, const float *inp, float theta, float alpha, float *out, void *stream) {
    if (!hi || !S2) return dgg_set_error(DGG_ERR_ARG, "gcnii_gemm_bf16_split: hi and the second operand half are required");
    const GcniiEpi ep{hi, h0, inp, theta, alpha};
    return launch_gemm(reinterpret_cast<const __bf16 *>(S1), reinterpret_cast<const __bf16 *>(Wt), (int)n, (int)F, (int)K, 1.0f, out, &ep,
                       (hipStream_t)stream, 1, reinterpret_cast<const __bf16 *>(S2), (int)F1);
}

// dgg_gcnii_gemm_bf16_split with the layer's ReLU and the next layer's dropout in the epilogue (fused GCNII stack): relu 0 = neither;
// drop_p in [0,1): out = keep(e) ? relu(.) / (1 - drop_p) : 0, keep(e) the counter-based mask of seeds (s0, s1) on element e = row*F + col
int dgg_gcnii_gemm_bf16_split_act(const void *S1, const void *S2, const void *Wt, int64_t n, int64_t F, int64_t K, int64_t F1, const float *hi,
                                  const float *h0, const float *inp, float theta, float alpha, int relu, float drop_p, uint32_t s0,
                                  uint32_t s1, float *out, void *outb, void *stream) {
    if (!hi || !S2) return dgg_set_error(DGG_ERR_ARG, "gcnii_gemm_bf16_split_act: hi and the second operand half are required");
    if (!(drop_p >= 0.0f && drop_p < 1.0f) || n * F >= ((int64_t)1 << 32)) return dgg_set_error(DGG_ERR_ARG, "gcnii_gemm_bf16_split_act: drop_p in [0,1), n*F < 2^32");
    GcniiEpi ep{hi, h0, inp, theta, alpha};
    ep.relu = relu; ep.drop_thr24 = relu ? (uint32_t)(drop_p * 16777216.0f) : 0u; ep.s0 = s0; ep.s1 = s1; ep.drop_scale = 1.0f / (1.0f - drop_p);
    ep.outb = reinterpret_cast<__bf16 *>(outb);
    return launch_gemm(reinterpret_cast<const __bf16 *>(S1), reinterpret_cast<const __bf16 *>(Wt), (int)n, (int)F, (int)K, 1.0f, out, &ep,
                       (hipStream_t)stream, 1, reinterpret_cast<const __bf16 *>(S2), (int)F1);
}

// see gcnii_gout_pack_kernel: gin, xd, g fp32 [n,F]; Gp bf16 [n,F]; GT bf16 [F, ldT], ldT >= n a multiple of 64 (columns beyond n zeroed);
// g2 (nullable): a second fp32 copy of g
int dgg_gcnii_gout_pack(const float *gin, const float *xd, float scale, int64_t n, int64_t F, float *g, void *Gp, void *GT, int64_t ldT,
                        float *g2, void *stream) {
    if (n <= 0 || F <= 0) return 0;
    if (ldT < n || ldT % 64 != 0 || F % 64 != 0) return dgg_set_error(DGG_ERR_ARG, "gcnii_gout_pack: F and ldT multiples of 64, ldT >= n");
    const uintptr_t al = reinterpret_cast<uintptr_t>(gin) | reinterpret_cast<uintptr_t>(xd) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(g2) |
                         reinterpret_cast<uintptr_t>(GT) | (reinterpret_cast<uintptr_t>(Gp) << 1);
    if (al % 16 == 0 && !getenv("DGG_PACK_SLOW")) {
        hipLaunchKernelGGL(gcnii_gout_pack_t64, dim3((unsigned)(F / 64), (unsigned)(ldT / 64)), dim3(256), 0, (hipStream_t)stream, gin, xd, scale, (int)n,
                           (int)F, g, reinterpret_cast<__bf16 *>(Gp), reinterpret_cast<__bf16 *>(GT), (int)ldT, g2);
        return dgg_check_launch("gcnii_gout_pack");
    }
    const dim3 grid((unsigned)((F + 31) / 32), (unsigned)((ldT + 31) / 32));
    hipLaunchKernelGGL(gcnii_gout_pack_kernel, grid, dim3(256), 0, (hipStream_t)stream, gin, xd, scale, (int)n, (int)F, g,
                       reinterpret_cast<__bf16 *>(Gp), reinterpret_cast<__bf16 *>(GT), (int)ldT, g2);
    return dgg_check_launch("gcnii_gout_pack");
}

// Backward of the variant layer w.r.t. its two inputs: [d hi | d h0] = theta * Gp W^T + [c1 | c2] * g with
// c1 = (1 - theta)(1 - alpha), c2 = (1 - theta) alpha (model.py:41-44): Gp bf16 [n,F] (= bf16(g)), Wp bf16 [2F, F] (the weight as
// stored), g fp32 [n,F].  One product per half (rows [0,F) / [F,2F) of Wp as the B operand), epilogue fused: no [n,2F] intermediate,
// no slicing adds.  dhi fp32 [n,F] (nullable when dhib is given), dhib = bf16(d hi) (nullable) for the transposed aggregation that
// gathers it next, dh0 fp32 [n,F]; accumulate_dh0: dh0 += its half instead of =.
static int dsupport_halves(const void *Gp, const void *Wp, int64_t n, int64_t F, const float *g, float theta, float alpha, float *dhi, float *dh0,
                           void *dhib, int accumulate_dh0, void *stream) {
    if (!g || !dh0 || (!dhi && !dhib)) return dgg_set_error(DGG_ERR_ARG, "gcnii_dsupport_bf16: g, dh0 and one of dhi / dhib are required");
    GcniiEpi ep{};
    ep.g = g; ep.c1 = (1.0f - theta) * (1.0f - alpha);
    ep.outb = reinterpret_cast<__bf16 *>(dhib);
    const __bf16 *W = reinterpret_cast<const __bf16 *>(Wp);
    int rc = launch_gemm(reinterpret_cast<const __bf16 *>(Gp), W, (int)n, (int)F, (int)F, theta, dhi, &ep, (hipStream_t)stream, 2);
    if (rc) return rc;
    ep.c1 = (1.0f - theta) * alpha;
    ep.outb = nullptr;
    return launch_gemm(reinterpret_cast<const __bf16 *>(Gp), W + F * F, (int)n, (int)F, (int)F, theta, dh0, &ep, (hipStream_t)stream,
                       accumulate_dh0 ? 3 : 2);
}
int dgg_gcnii_dsupport_bf16_b(const void *Gp, const void *Wp, int64_t n, int64_t F, const float *g, float theta, float alpha, float *dhi,
                              float *dh0, void *dhib, int accumulate_dh0, void *stream) {
    return dsupport_halves(Gp, Wp, n, F, g, theta, alpha, dhi, dh0, dhib, accumulate_dh0, stream);
}
int dgg_gcnii_dsupport_bf16(const void *Gp, const void *Wp, int64_t n, int64_t F, const float *g, float theta, float alpha, float *dhi,
                            float *dh0, void *stream) {
    if (!dhi) return dgg_set_error(DGG_ERR_ARG, "gcnii_dsupport_bf16: dhi is required");
    return dsupport_halves(Gp, Wp, n, F, g, theta, alpha, dhi, dh0, nullptr, 0, stream);
}

// C[M,N] = scale * [A ; A2] B^T with the rows [0, M1) of the A operand in A [M1,K] and the rows [M1, M) in A2 [M-M1,K] (M1 a multiple
// of 128): the weight gradient of the variant layer, theta * [hi | h0]^T g, from the two transposed halves in one product
int dgg_gemm_nt_bf16_rows2(const void *A, const void *A2, int64_t M1, const void *B, int64_t M, int64_t N, int64_t K, float scale, float *C,
                           void *stream) {
    if (!A2) return dgg_set_error(DGG_ERR_ARG, "gemm_nt_bf16_rows2: the second operand block is required");
    return launch_gemm(reinterpret_cast<const __bf16 *>(A), reinterpret_cast<const __bf16 *>(B), (int)M, (int)N, (int)K, scale, C, nullptr,
                       (hipStream_t)stream, 0, reinterpret_cast<const __bf16 *>(A2), -(int)M1);
}

}  // extern "C"
