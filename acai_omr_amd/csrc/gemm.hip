// C[M,N] = epi(A[M,K] . W[N,K]^T + bias) (+ residual): the nn.Linear of every projection on the path
// (reference: acai_omr/models/models.py:29,57,204,205,428,655-660; kv_caching.py:193,215,244).
//
// gfx950 design: 128x128 block tile, 4 waves (2x2), each wave owns 64x64 = 2x2 MFMA 32x32 tiles in
// 64 accumulator VGPRs.  Both operands are K-contiguous ("NT"), which is exactly the MFMA A/B lane
// layout (lane (r,h) holds 8 consecutive k of row r), so A and W tiles are staged as 128-byte row
// segments (bf16: BK = 64, fp32: BK = 32) with a 144-byte LDS pitch: ds_read_b128 of 32 rows at one
// column offset is then bank-conflict free (36*r mod 64 distinct over each 16-lane group).
// bf16 -> v_mfma_f32_32x32x16_bf16; fp32 -> v_mfma_f32_32x32x2_f32 (exact fp32 fma chain; one
// ds_read_b128 feeds four K=2 MFMAs because lane-half h takes k = 8s+4h+j, j = 0..3).
// Global -> register prefetch of tile t+1 is issued before the MFMAs of tile t (issue-early /
// write-late staging), one LDS stage, ~3 workgroups per CU cover each other's barriers.
#include <stdlib.h>

#include "common.h"
#include <type_traits>

namespace {

constexpr int BM = 128, BN = 128, ROWB = 128, PITCH = 144;

struct GemmArgs {
    const void *A, *W;
    const float *bias, *residual;
    void *C;
    int lda, ldw, ldr, ldc, M, N, K, out_dtype, flags;
    int vec_epi; // host-checked: C (and the residual) allow 16-byte row accesses -> LDS-transposed epilogue
    int ksplit;  // > 0: split-K over `ksplit` workgroups per tile, fp32 atomic accumulation into C (weight gradients: K = rows)
    // aux_mode 1: `aux` (dtype of C) also receives the pre-activation while C gets GELU of it (training forward keeps both);
    // aux_mode 2: C = round(acc) * gelu'(aux)  (the dX GEMM of linear2 applies the GELU derivative of the saved pre-activation)
    // aux_mode 3 (round 4, what the training steps use): as 1, but `aux` receives gelu'(pre-activation) - the forward epilogue holds
    //             Phi(-|a|) anyway - and aux_mode 4: C = round(acc) * aux, ONE multiply per element in the backward epilogue, which was
    //             VALU-bound on the derivative (tools/ablate_gelu_grad.py)
    void *aux;
    int ldaux, aux_mode;
    // columns [0, scale_cols) of (A.W^T + bias) are multiplied by col_scale before rounding (the in-projection's q for the attention kernels)
    int scale_cols;
    float col_scale;
    // dW form (TA && TB): also accumulate the column sums of the token-major A operand (= the bias gradient next to dW = dY^T X) into colsum[M]
    float *colsum;
    int *colsum_done;   // host-side out flag of the dW dispatch: the launched kernel formed `colsum` itself (acai_gemm_dw runs a separate pass otherwise)
    // EPI == 1 (cross K/V prefill scatter)
    const int32_t *row_seq, *row_pos, *seq_len;
    const int64_t *seq_off;
    void *k_out, *v_out;
    int E, dh, dhp;
};

template <typename T, bool FAST>
__device__ __forceinline__ uint4 load_chunk(const T *base, int ld, int row, int rows, int k0, int K) {
    constexpr int EPC = 16 / sizeof(T);
    uint4 r = make_uint4(0, 0, 0, 0);
    if (row >= rows) return r;
    if constexpr (FAST) {
        if (k0 < K) r = *reinterpret_cast<const uint4 *>(base + (size_t)row * ld + k0);
    } else {
        union { uint4 v; T e[EPC]; } u;
        u.v = r;
#pragma unroll
        for (int e = 0; e < EPC; ++e)
            if (k0 + e < K) u.e[e] = base[(size_t)row * ld + k0 + e];
        r = u.v;
    }
    return r;
}

// ---- epilogue shared by both main loops: C/D layout col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5) ----------------
template <int EPI, bool ACCUM>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs &g, const f32x16 (&acc)[2][2], int bm0, int bn0, int wm, int wn, int lr, int lh) {
    constexpr bool TA = ACCUM, TB = ACCUM;
    const bool do_gelu = g.flags & ACAI_GEMM_GELU, do_round = g.flags & ACAI_GEMM_ROUND_BF16;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = bn0 + wn * 64 + j * 32 + lr;
        if (col >= g.N) continue;
        const float bv = g.bias ? g.bias[col] : 0.f;
        const float cs = col < g.scale_cols ? g.col_scale : 1.0f;
        int kv = 0, hh = 0, dd = 0;
        if constexpr (EPI == 1) {
            kv = col / g.E;
            const int e = col - kv * g.E;
            hh = e / g.dh;
            dd = e - hh * g.dh;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = bm0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (row >= g.M) continue;
                float v = acc[i][j][e] + bv;
                if constexpr (EPI == 0) {
                    v *= cs;
                    if (do_round) v = round_bf16(v);
                    if (g.aux_mode == 1 || g.aux_mode == 3) {
                        const float keep = g.aux_mode == 1 ? v : gelu_erf_grad(v);
                        if (g.out_dtype == ACAI_BF16) reinterpret_cast<bf16_t *>(g.aux)[(size_t)row * g.ldaux + col] = f2bf(keep);
                        else reinterpret_cast<float *>(g.aux)[(size_t)row * g.ldaux + col] = keep;
                    } else if (g.aux_mode == 2 || g.aux_mode == 4) {
                        const float a0 = g.out_dtype == ACAI_BF16 ? bf2f(reinterpret_cast<const bf16_t *>(g.aux)[(size_t)row * g.ldaux + col])
                                                                  : reinterpret_cast<const float *>(g.aux)[(size_t)row * g.ldaux + col];
                        v *= g.aux_mode == 2 ? gelu_erf_grad(a0) : a0;
                    }
                    if (do_gelu) {
                        v = gelu_erf(v);
                        if (do_round) v = round_bf16(v);
                    }
                    if (g.residual) v += g.residual[(size_t)row * g.ldr + col];
                    if (g.out_dtype == ACAI_BF16)
                        reinterpret_cast<bf16_t *>(g.C)[(size_t)row * g.ldc + col] = f2bf(v);
                    else
                        reinterpret_cast<float *>(g.C)[(size_t)row * g.ldc + col] = v;
                } else {
                    const int b = g.row_seq[row], s = g.row_pos[row];
                    const int64_t off = g.seq_off[b] + ((int64_t)hh * g.seq_len[b] + s) * g.dhp + dd;
                    void *dst = kv ? g.v_out : g.k_out;
                    if (g.out_dtype == ACAI_BF16)
                        reinterpret_cast<bf16_t *>(dst)[off] = f2bf(v);
                    else
                        reinterpret_cast<float *>(dst)[off] = v;
                }
            }
    }
}

// Split-K accumulation epilogue (weight gradients): fp32 atomics straight from the C/D layout - a wave instruction adds two 128-byte row
// segments, the shape the atomic units take at full rate.  One 32x32 block at a time behind a scheduling barrier: left to itself the compiler
// materialises all 64 row addresses first, which cost the register-staged dW kernel 430 registers (one wave per SIMD).
__device__ __forceinline__ void gemm_accum_epilogue(const GemmArgs &g, const f32x16 (&acc)[2][2], int bm0, int bn0, int wm, int wn, int lr, int lh) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = bn0 + wn * 64 + j * 32 + lr, row0 = bm0 + wm * 64 + i * 32 + 4 * lh;
            if (col < g.N) {
                float *base = reinterpret_cast<float *>(g.C) + (size_t)row0 * g.ldc + col;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int ro = (e & 3) + 8 * (e >> 2);
                    if (row0 + ro < g.M) atomicAdd(base + (size_t)ro * g.ldc, acc[i][j][e]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
}

// Transposed storage (backward GEMMs: dX = dY.W reads W as [K][N]; dW = dY^T.X reads both operands reduction-major):
// the operand is stored [k][row] with `row` contiguous.  A thread still owns one 16-byte global chunk (EPC consecutive
// rows at one k) and scatters its elements into EPC LDS rows, so the LDS image and the MFMA loop are unchanged.
template <typename T, bool FAST>
__device__ __forceinline__ uint4 load_chunk_t(const T *base, int ld, int k, int K, int row0, int rows) {
    constexpr int EPC = 16 / sizeof(T);
    uint4 r = make_uint4(0, 0, 0, 0);
    if (k >= K) return r;
    if constexpr (FAST) {
        if (row0 < rows) r = *reinterpret_cast<const uint4 *>(base + (size_t)k * ld + row0);
    } else {
        union { uint4 v; T e[EPC]; } u;
        u.v = r;
#pragma unroll
        for (int e = 0; e < EPC; ++e)
            if (row0 + e < rows) u.e[e] = base[(size_t)k * ld + row0 + e];
        r = u.v;
    }
    return r;
}

template <typename T, int EPI, bool FAST, bool TA = false, bool TB = false>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs g) {
    constexpr int EPC = 16 / sizeof(T);     // elements per 16-byte chunk
    constexpr int BK = ROWB / sizeof(T);    // k elements per tile
    // A transposed operand keeps its natural [k][128 rows] image in LDS (16-byte stores, no scatter); its MFMA fragments are
    // read with ds_read_b64_tr_b16 (bf16: two 4x16 transposing reads per fragment; pitch 320 B puts the 4 k-rows of both
    // 16-lane groups of a half-wave on disjoint banks) or one ds_read_b32 per K=2 MFMA (fp32).
    constexpr int PT = sizeof(T) == 2 ? 320 : 528;
    constexpr int SZA = TA ? BK * PT : BM * PITCH, SZB = TB ? BK * PT : BN * PITCH;
    __shared__ __attribute__((aligned(16))) unsigned char lds[SZA + SZB];
    unsigned char *ldsA = lds, *ldsB = lds + SZA;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;

    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD a
    // contiguous run of tiles along N (they share the same A panel in that XCD's L2).
    const int nbn = (g.N + BN - 1) / BN, nbm = (g.M + BM - 1) / BM;
    const int nwg = nbn * nbm;
    int pid = blockIdx.x;
    int kslice = 0;
    if (g.ksplit > 1) {
        kslice = pid / nwg;
        pid -= kslice * nwg;
    }
    {
        const int q = nwg / 8, r = nwg % 8, xcd = pid % 8, idx = pid / 8;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bm0 = (pid / nbn) * BM, bn0 = (pid % nbn) * BN;

    const T *A = reinterpret_cast<const T *>(g.A);
    const T *W = reinterpret_cast<const T *>(g.W);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    uint4 ra[4], rb[4];
    int nkt = (g.K + BK - 1) / BK, kt_begin = 0;
    if (g.ksplit > 1) {
        const int per = (nkt + g.ksplit - 1) / g.ksplit;
        kt_begin = kslice * per;
        nkt = min(nkt, kt_begin + per);
        if (kt_begin >= nkt) return;  // uniform per workgroup, before any barrier
    }

    // transposed staging: chunk c -> k index c / (128/EPC), row group c % (128/EPC) (consecutive lanes walk the contiguous dim)
    constexpr int RG = 128 / EPC;  // 16-byte row groups per 128-row tile
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i, row = c >> 3, k0 = kt * BK + (c & 7) * EPC;
            if constexpr (TA) ra[i] = load_chunk_t<T, FAST>(A, g.lda, kt * BK + c / RG, g.K, bm0 + (c % RG) * EPC, g.M);
            else ra[i] = load_chunk<T, FAST>(A, g.lda, bm0 + row, g.M, k0, g.K);
            if constexpr (TB) rb[i] = load_chunk_t<T, FAST>(W, g.ldw, kt * BK + c / RG, g.K, bn0 + (c % RG) * EPC, g.N);
            else rb[i] = load_chunk<T, FAST>(W, g.ldw, bn0 + row, g.N, k0, g.K);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i, row = c >> 3, cb = (c & 7) * 16;
            if constexpr (TA) *reinterpret_cast<uint4 *>(ldsA + (c / RG) * PT + (c % RG) * 16) = ra[i];
            else *reinterpret_cast<uint4 *>(ldsA + row * PITCH + cb) = ra[i];
            if constexpr (TB) *reinterpret_cast<uint4 *>(ldsB + (c / RG) * PT + (c % RG) * 16) = rb[i];
            else *reinterpret_cast<uint4 *>(ldsB + row * PITCH + cb) = rb[i];
        }
    };

    // fragment of 32 rows starting at `rowbase`, k-step s (bf16: k = 16s + 8h + j; fp32: k = 8s + 4h + e), from a row-major image
    // (ds_read_b128) or from a transposed operand's natural image
    auto frag = [&](const unsigned char *img, bool trans, int rowbase, int s) -> uint4 {
        if (!trans) return *reinterpret_cast<const uint4 *>(img + (rowbase + lr) * PITCH + s * 32 + lh * 16);
        if constexpr (sizeof(T) == 2) {
            typedef __attribute__((ext_vector_type(4))) short s4;
            typedef __attribute__((address_space(3))) s4 *lds_s4;
            const int i16 = lane & 15, g1 = (lane >> 4) & 1;  // lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3
            const unsigned char *p0 = img + (16 * s + 8 * lh + (i16 >> 2)) * PT + (rowbase + 16 * g1 + 4 * (i16 & 3)) * 2;
            const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(p0));
            const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(p0 + 4 * PT));
            union { s4 v[2]; uint4 u; } r;
            r.v[0] = lo;
            r.v[1] = hi;
            return r.u;
        } else {
            const unsigned char *p0 = img + (8 * s + 4 * lh) * PT + (rowbase + lr) * 4;
            uint4 r;
            r.x = *reinterpret_cast<const uint32_t *>(p0);
            r.y = *reinterpret_cast<const uint32_t *>(p0 + PT);
            r.z = *reinterpret_cast<const uint32_t *>(p0 + 2 * PT);
            r.w = *reinterpret_cast<const uint32_t *>(p0 + 3 * PT);
            return r;
        }
    };

    load_tile(kt_begin);
    store_tile();
    __syncthreads();
    for (int kt = kt_begin; kt < nkt; ++kt) {
        if (kt + 1 < nkt) load_tile(kt + 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            uint4 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = frag(ldsA, TA, wm * 64 + i * 32, s);
                fb[i] = frag(ldsB, TB, wn * 64 + i * 32, s);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (sizeof(T) == 2) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]),
                                                                            __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
                    } else {
                        const f32x4 a4 = __builtin_bit_cast(f32x4, fa[i]), b4 = __builtin_bit_cast(f32x4, fb[j]);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b4[e], acc[i][j], 0, 0, 0);
                    }
                }
        }
        __syncthreads();
        if (kt + 1 < nkt) {
            store_tile();
            __syncthreads();
        }
    }

    if constexpr (EPI == 0 && TA && TB) {
        if (g.ksplit > 0) {
            gemm_accum_epilogue(g, acc, bm0, bn0, wm, wn, lr, lh);
            return;
        }
    }
    gemm_epilogue<EPI, TA && TB>(g, acc, bm0, bn0, wm, wn, lr, lh);
}

// ---- LDS-transposed epilogue of the LDS-DMA kernels -----------------------------------------------------------------------------------
// The MFMA C/D layout gives a lane one column and 16 scattered rows: storing from it means 2- or 4-byte stores in 64/128-byte runs, which
// bounds the short-K GEMMs (K = 512..768: 8-12 K-steps per output tile).  Each wave therefore transposes its 64x64 block through `stg`
// (its share of the now free LDS stages), 32 rows at a time, and writes 16 bytes per lane along the rows.  Bias and bf16 rounding are applied
// on the way in; GELU (optionally keeping the pre-activation in `aux`), the GELU-derivative product and the fp32 residual on the way out.
template <int ROWS = 32>   // rows staged per pass: 32 (8.5 KB of LDS per wave) or 16 (4.25 KB: the persistent kernel has one free stage)
__device__ __forceinline__ void gemm_vec_epilogue(const GemmArgs &g, const f32x16 (&acc)[2][2], float *stg, int bm0, int bn0, int wm, int wn, int lane) {
    constexpr int EP = 68;  // floats per staged row (64 + 4: 16-byte aligned rows, conflict-light)
    constexpr int NP = 32 / ROWS;   // passes per 32-row block
    const int lr = lane & 31, lh = lane >> 5;
    const int ldc_e = g.ldc, n_valid = g.N - (bn0 + wn * 64);
    const bool do_gelu = g.flags & ACAI_GEMM_GELU, do_round = g.flags & ACAI_GEMM_ROUND_BF16;
    // The bf16 rounding on the way into LDS is skipped when the value goes straight to a bf16 store, which rounds the same number the same way
    // (measured: no change in tile time -- the epilogue is a latency chain, not VALU issue).
    const bool pre_round = do_round && (do_gelu || g.aux_mode != 0 || g.residual != nullptr || g.out_dtype != ACAI_BF16);
    // `a4`: the saved pre-activation of these four columns (aux_mode 2, loaded by the caller in one access per lane)
    auto finish = [&](f32x4 &v, int row, int col, const float (&a4)[4]) {   // four consecutive columns of one row, after bias / rounding
        if (g.aux_mode == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= gelu_erf_grad(a4[e]);
        } else if (g.aux_mode == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= a4[e];
        }
        if (do_gelu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = gelu_erf(v[e]);
                if (do_round) v[e] = round_bf16(v[e]);
            }
        }
        if (g.residual) v += *reinterpret_cast<const f32x4 *>(g.residual + (size_t)row * g.ldr + col);
    };
#pragma unroll
    for (int ip = 0; ip < 2 * NP; ++ip) {
        const int i = ip / NP, pass = ip % NP;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = bn0 + wn * 64 + j * 32 + lr;
            const float bv = (g.bias && col < g.N) ? g.bias[col] : 0.f;
            const float cs = col < g.scale_cols ? g.col_scale : 1.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (NP == 2 && (e >> 3) != pass) continue;   // 16-row passes: registers 0..7 hold rows 0..15, 8..15 rows 16..31
                float v = (acc[i][j][e] + bv) * cs;
                if (pre_round) v = round_bf16(v);
                stg[((e & 3) + 8 * ((e >> 2) & (NP == 2 ? 1 : 3)) + 4 * lh) * EP + j * 32 + lr] = v;
            }
        }
        // same wave wrote and now reads: LDS operations of a wave complete in order
        const int row_base = bm0 + wm * 64 + i * 32 + pass * ROWS, col_base = bn0 + wn * 64;
        if (g.out_dtype == ACAI_BF16) {
#pragma unroll
            for (int it = 0; it < ROWS / 8; ++it) {
                const int idx = it * 64 + lane, r = idx >> 3, c8 = (idx & 7) * 8;
                const int row = row_base + r;
                if (row < g.M && c8 < n_valid) {
                    f32x4 v0 = *reinterpret_cast<const f32x4 *>(stg + r * EP + c8), v1 = *reinterpret_cast<const f32x4 *>(stg + r * EP + c8 + 4);
                    float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
                    bf16_t *auxp = reinterpret_cast<bf16_t *>(g.aux) + (size_t)row * g.ldaux + col_base + c8;
                    if (g.aux_mode == 2 || g.aux_mode == 4) {        // saved pre-activation / derivative: ONE 16-byte load per lane
                        const uint4 r4 = *reinterpret_cast<const uint4 *>(auxp);
                        a0[0] = __uint_as_float(r4.x << 16); a0[1] = __uint_as_float(r4.x & 0xFFFF0000u);
                        a0[2] = __uint_as_float(r4.y << 16); a0[3] = __uint_as_float(r4.y & 0xFFFF0000u);
                        a1[0] = __uint_as_float(r4.z << 16); a1[1] = __uint_as_float(r4.z & 0xFFFF0000u);
                        a1[2] = __uint_as_float(r4.w << 16); a1[3] = __uint_as_float(r4.w & 0xFFFF0000u);
                    } else if (g.aux_mode == 1) {  // keep the pre-activation: ONE 16-byte store per lane (was two 8-byte ones: the tail is store-issue bound)
                        uint4 p4;
                        p4.x = pack_bf16(v0[0], v0[1]); p4.y = pack_bf16(v0[2], v0[3]); p4.z = pack_bf16(v1[0], v1[1]); p4.w = pack_bf16(v1[2], v1[3]);
                        *reinterpret_cast<uint4 *>(auxp) = p4;
                    } else if (g.aux_mode == 3) {  // keep gelu'(pre-activation)
                        uint4 p4;
                        p4.x = pack_bf16(gelu_erf_grad(v0[0]), gelu_erf_grad(v0[1])); p4.y = pack_bf16(gelu_erf_grad(v0[2]), gelu_erf_grad(v0[3]));
                        p4.z = pack_bf16(gelu_erf_grad(v1[0]), gelu_erf_grad(v1[1])); p4.w = pack_bf16(gelu_erf_grad(v1[2]), gelu_erf_grad(v1[3]));
                        *reinterpret_cast<uint4 *>(auxp) = p4;
                    }
                    finish(v0, row, col_base + c8, a0);
                    finish(v1, row, col_base + c8 + 4, a1);
                    uint4 o;
                    o.x = pack_bf16(v0[0], v0[1]); o.y = pack_bf16(v0[2], v0[3]); o.z = pack_bf16(v1[0], v1[1]); o.w = pack_bf16(v1[2], v1[3]);
                    *reinterpret_cast<uint4 *>(reinterpret_cast<bf16_t *>(g.C) + (size_t)row * ldc_e + col_base + c8) = o;
                }
            }
        } else {
#pragma unroll
            for (int it = 0; it < ROWS / 4; ++it) {
                const int idx = it * 64 + lane, r = idx >> 4, c4 = (idx & 15) * 4;
                const int row = row_base + r;
                if (row < g.M && c4 < n_valid) {
                    f32x4 v0 = *reinterpret_cast<const f32x4 *>(stg + r * EP + c4);
                    float a0[4] = {0.f, 0.f, 0.f, 0.f};
                    float *auxp = reinterpret_cast<float *>(g.aux) + (size_t)row * g.ldaux + col_base + c4;
                    if (g.aux_mode == 2 || g.aux_mode == 4) {
                        const f32x4 r4 = *reinterpret_cast<const f32x4 *>(auxp);
#pragma unroll
                        for (int e = 0; e < 4; ++e) a0[e] = r4[e];
                    } else if (g.aux_mode == 1) {
                        *reinterpret_cast<f32x4 *>(auxp) = v0;
                    } else if (g.aux_mode == 3) {
                        *reinterpret_cast<f32x4 *>(auxp) = f32x4{gelu_erf_grad(v0[0]), gelu_erf_grad(v0[1]), gelu_erf_grad(v0[2]), gelu_erf_grad(v0[3])};
                    }
                    finish(v0, row, col_base + c4, a0);
                    *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(g.C) + (size_t)row * ldc_e + col_base + c4) = v0;
                }
            }
        }
    }
}

// ---- NT main loop with direct-to-LDS staging (global_load_lds, 16 B per lane) ------------------------------------------------
// For row-major operands with K % BK == 0: no staging VGPRs, two LDS stages (64 KB), tile t+1 in flight while tile t feeds the MFMAs,
// one barrier per K-step.  An LDS-DMA wave instruction writes 1 KiB linearly (lane l -> base + 16 l = row l/8, slot l%8 of an unpadded
// 128-byte row), so the bank-conflict fix is an XOR swizzle applied on BOTH sides: lane l fetches the global chunk (l%8) ^ ((row/2)%8) and
// the fragment read of logical chunk c goes to slot c ^ ((row/2)%8).  Rows beyond M / N are clamped (their outputs are never stored).
// WM = waves along M: 2 -> 128x128 tile, 4 waves (two workgroups per CU); 4 -> 256x128 tile, 8 waves (one workgroup per CU).  A CU takes in
// only ~55 GB/s through the LDS-DMA path, which caps the 128x128 tile (64 flop per staged byte) near 0.9 PF; the 256-row tile stages
// 25 % fewer bytes per flop and is used whenever it still yields >= 2 tiles per CU.
template <typename T, int EPI, int WM>
__global__ __launch_bounds__(WM * 128) __attribute__((amdgpu_waves_per_eu(2))) void gemm_nt_glds_kernel(GemmArgs g) {
    constexpr int BK = ROWB / sizeof(T);
    constexpr int BMT = WM * 64, NW = WM * 2;          // tile rows, waves
    constexpr int GA = BMT / 8 / NW, GW = BN / 8 / NW;  // 8-row groups (1 KiB LDS-DMA instructions) per wave and K-step: A, W
    constexpr int STAGE = (BMT + BN) * ROWB;           // 32 / 48 KB
    // Two DISTINCT LDS objects, addressed statically (the K loop is unrolled by two): the compiler's wait-count pass can then prove that
    // the fragment reads of one stage do not alias the LDS-DMA in flight into the other.  With one array and a run-time stage index it
    // put `s_waitcnt vmcnt(0)` in front of the first ds_read of every K-step, i.e. it drained tile t+1 before computing tile t.
    __shared__ __attribute__((aligned(16))) unsigned char lds0[STAGE];
    __shared__ __attribute__((aligned(16))) unsigned char lds1[STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
    const int nbn = (g.N + BN - 1) / BN, nbm = (g.M + BMT - 1) / BMT, nwg = nbn * nbm;
    int pid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = pid % 8, idx = pid / 8;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bm0 = (pid / nbn) * BMT, bn0 = (pid % nbn) * BN;
    const T *A = reinterpret_cast<const T *>(g.A);
    const T *W = reinterpret_cast<const T *>(g.W);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // this lane's source chunk inside an 8-row group: row l/8; the 16-byte slot is XORed with bits 1..3 of the tile row (see SWZ below)
    const int grow = lane >> 3;
    const T *srcA[GA], *srcW[GW];
#pragma unroll
    for (int i = 0; i < GA; ++i) {
        const int row = (wave * GA + i) * 8 + grow;
        const int gslot = (lane & 7) ^ ((row >> 1) & 7);
        srcA[i] = A + (size_t)min(bm0 + row, g.M - 1) * g.lda + gslot * (16 / sizeof(T));
    }
#pragma unroll
    for (int i = 0; i < GW; ++i) {
        const int row = (wave * GW + i) * 8 + grow;
        const int gslot = (lane & 7) ^ ((row >> 1) & 7);
        srcW[i] = W + (size_t)min(bn0 + row, g.N - 1) * g.ldw + gslot * (16 / sizeof(T));
    }
    typedef __attribute__((address_space(3))) void *lds_ptr;
    typedef const __attribute__((address_space(1))) void *glb_ptr;
    auto issue = [&](int kt, unsigned char *base) {
#pragma unroll
        for (int i = 0; i < GA; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr)(srcA[i] + (size_t)kt * BK), (lds_ptr)(base + (wave * GA + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < GW; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr)(srcW[i] + (size_t)kt * BK), (lds_ptr)(base + BMT * ROWB + (wave * GW + i) * 1024), 16, 0, 0);
    };
    // Fragment read of logical 16-byte chunk c of tile row r goes to slot c ^ ((r >> 1) & 7).  A 256-byte bank row holds two tile rows, and a
    // ds_read_b128 is served 16 lanes (= 16 consecutive tile rows, one k-half) per cycle: bit 0 of r picks the half of the bank row, bits 1..3 the
    // slot inside it, so the 16 lanes hit 16 different 16-byte bank groups (slot ^ (r & 7) paired rows r and r + 8: 2-way conflicts).
    // Fragments are double-buffered: the reads of k-slice s+1 are issued before the MFMAs of slice s.
    auto compute = [&](const unsigned char *sa) {
        const unsigned char *sb = sa + BMT * ROWB;
        uint4 fa[2][2], fb[2][2];
        auto frags = [&](int s, uint4 (&xa)[2], uint4 (&xb)[2]) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ra = wm * 64 + i * 32 + lr, rb = wn * 64 + i * 32 + lr;
                xa[i] = *reinterpret_cast<const uint4 *>(sa + ra * ROWB + (((s * 2 + lh) ^ ((ra >> 1) & 7)) << 4));
                xb[i] = *reinterpret_cast<const uint4 *>(sb + rb * ROWB + (((s * 2 + lh) ^ ((rb >> 1) & 7)) << 4));
            }
        };
        frags(0, fa[0], fb[0]);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s + 1 < 4) frags(s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (sizeof(T) == 2) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[s & 1][i]), __builtin_bit_cast(bf16x8, fb[s & 1][j]), acc[i][j], 0, 0, 0);
                    } else {
                        const f32x4 a4 = __builtin_bit_cast(f32x4, fa[s & 1][i]), b4 = __builtin_bit_cast(f32x4, fb[s & 1][j]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b4[e], acc[i][j], 0, 0, 0);
                    }
                }
        }
    };
    const int nkt = g.K / BK;
    issue(0, lds0);
    __syncthreads();  // with an LDS-DMA in flight this is s_waitcnt vmcnt(0) + s_barrier
    int kt = 0;
    for (; kt + 2 <= nkt; kt += 2) {
        issue(kt + 1, lds1);
        compute(lds0);
        __syncthreads();  // tile kt+1 has landed (vmcnt(0)) and every wave is done reading stage 0
        if (kt + 2 < nkt) issue(kt + 2, lds0);
        compute(lds1);
        __syncthreads();
    }
    if (kt < nkt) {  // odd tile count: the last tile sits in stage 0
        compute(lds0);
        __syncthreads();
    }
    // ---- epilogue (see gemm_vec_epilogue) ------------------------------------------------------------------------------------------------
    if (EPI != 0 || !g.vec_epi) {
        gemm_epilogue<EPI, false>(g, acc, bm0, bn0, wm, wn, lr, lh);
        return;
    }
    // every wave is past the last barrier: both stages are free; a wave stages 32 rows x 68 floats = 8.5 KB
    float *stg = reinterpret_cast<float *>((wave < WM ? lds0 : lds1) + (wave % WM) * (STAGE / WM));
    gemm_vec_epilogue<32>(g, acc, stg, bm0, bn0, wm, wn, lane);
}

// One 16-byte-per-lane LDS-DMA instruction from inline asm (hidden from the compiler's wait-count pass, see gemm_nt_glds3_kernel): M0 carries
// the wave-uniform LDS destination; it is saved and restored around the instruction, so the asm has no reserved-register clobber.
__device__ __forceinline__ void glds16(uint32_t lds_dst, const void *src) {
    uint32_t m0_save;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_save)
                 : "s"(lds_dst), "v"(src)
                 : "memory");
}

// the same with the source as a uniform base (SGPR pair) + a 32-bit byte offset per lane
__device__ __forceinline__ void glds16s(uint32_t lds_dst, const void *sbase, uint32_t voff) {
    uint32_t m0_save;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_save)
                 : "s"(lds_dst), "v"(voff), "s"(sbase)
                 : "memory");
}

// four pieces of one 32 KB unit (consecutive 1 KiB destinations 8 KiB apart: wave w's pieces of a [256][128 B] image) in one block: one M0
// save / restore instead of four
__device__ __forceinline__ void glds16s_x4(uint32_t lds_dst, const void *sbase, uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3) {
    uint32_t m0_save;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %6\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %6\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %6\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %6\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(m0_save)
                 : "s"(lds_dst), "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(sbase)
                 : "memory", "scc");
}

// The same through a buffer resource (buffer_load_dwordx4 ... offen lds): lanes whose offset lies beyond the resource's num_records write
// ZEROS into LDS (measured, tools/experiments/lds_dma_oob.hip: per dword) - a ragged last K-tile of the token-major operands needs no
// remainder launch.  (Only the VGPR offset is range-checked, not an SGPR offset: the K-step's advance goes into the resource's base.)
__device__ __forceinline__ void blds16(uint32_t lds_dst, __amdgpu_buffer_rsrc_t rs, uint32_t voff) {
    uint32_t m0_save;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_save)
                 : "s"(lds_dst), "v"(voff), "s"(rs)
                 : "memory");
}

__device__ __forceinline__ void blds16_x4(uint32_t lds_dst, __amdgpu_buffer_rsrc_t rs, uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3) {
    uint32_t m0_save;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %6, 0 offen lds\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %6, 0 offen lds\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %6, 0 offen lds\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %6, 0 offen lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(m0_save)
                 : "s"(lds_dst), "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(rs)
                 : "memory", "scc");
}

// ---- 256x128 tile, 8 waves, THREE LDS stages (144 KB): two K-tiles in flight ------------------------------------------------------------
// The two-stage kernels are latency-bound, not MFMA-bound (PMC: MFMA busy 36 %, a third of the wave cycles in s_waitcnt, L2 hit rate 64 %):
// with one tile in flight per workgroup a CU has 64 KB outstanding, and 64 KB x 256 CUs / ~1.2 us of loaded L2/fabric latency is exactly the
// ~13.7 TB/s staging rate observed at 0.88 PF.  Throughput = (bytes in flight) x (flop per staged byte) / latency, so this variant keeps
// 2 x 48 KB in flight per CU on a tile with 1.33x the flop per byte.  One barrier per K-step:
//     s_waitcnt vmcnt(6)   this wave's six LDS-DMA instructions of tile t have landed (those of t+1 may still fly)
//     s_barrier            everybody's have; and everybody is done reading tile t-1, whose stage is reused next
//     issue tile t+2 -> stage (t+2) % 3;  MFMAs of tile t from stage t % 3
// (wait and barrier are one asm block: the fence inside __syncthreads() would drain vmcnt to 0).
template <typename T, int EPI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void gemm_nt_glds3_kernel(GemmArgs g) {
    constexpr int WM = 4;
    constexpr int BK = ROWB / sizeof(T);
    constexpr int BMT = WM * 64, NW = WM * 2;          // tile rows, waves
    constexpr int GA = BMT / 8 / NW, GW = BN / 8 / NW;  // 8-row groups (1 KiB LDS-DMA instructions) per wave and K-step: A, W
    constexpr int STAGE = (BMT + BN) * ROWB;           // 32 / 48 KB
    // DISTINCT LDS objects, addressed statically (the K loop is unrolled by the stage count): the compiler's wait-count pass can then prove that
    // the fragment reads of one stage do not alias the LDS-DMA in flight into the other.  With one array and a run-time stage index it
    // put `s_waitcnt vmcnt(0)` in front of the first ds_read of every K-step, i.e. it drained tile t+1 before computing tile t.
    __shared__ __attribute__((aligned(16))) unsigned char lds0[STAGE];
    __shared__ __attribute__((aligned(16))) unsigned char lds1[STAGE];
    __shared__ __attribute__((aligned(16))) unsigned char lds2[STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
    const int nbn = (g.N + BN - 1) / BN, nbm = (g.M + BMT - 1) / BMT, nwg = nbn * nbm;
    int pid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = pid % 8, idx = pid / 8;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bm0 = (pid / nbn) * BMT, bn0 = (pid % nbn) * BN;
    const T *A = reinterpret_cast<const T *>(g.A);
    const T *W = reinterpret_cast<const T *>(g.W);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // this lane's source chunk inside an 8-row group: row l/8; the 16-byte slot is XORed with bits 1..3 of the tile row (see SWZ below)
    const int grow = lane >> 3;
    const T *srcA[GA], *srcW[GW];
#pragma unroll
    for (int i = 0; i < GA; ++i) {
        const int row = (wave * GA + i) * 8 + grow;
        const int gslot = (lane & 7) ^ ((row >> 1) & 7);
        srcA[i] = A + (size_t)min(bm0 + row, g.M - 1) * g.lda + gslot * (16 / sizeof(T));
    }
#pragma unroll
    for (int i = 0; i < GW; ++i) {
        const int row = (wave * GW + i) * 8 + grow;
        const int gslot = (lane & 7) ^ ((row >> 1) & 7);
        srcW[i] = W + (size_t)min(bn0 + row, g.N - 1) * g.ldw + gslot * (16 / sizeof(T));
    }
    typedef __attribute__((address_space(3))) void *lds_ptr;
    typedef const __attribute__((address_space(1))) void *glb_ptr;
    // The LDS-DMA instructions are issued from inline asm: the compiler's wait-count pass tracks only a few LDS-DMA writers and, past that,
    // guards every LDS read with s_waitcnt vmcnt(0) - which would drain the two tiles in flight.  Hidden from it, the only vmcnt waits in the
    // loop are the counted ones below (no other vector-memory instruction is issued between the prologue and the epilogue).
    auto issue = [&](int kt, unsigned char *base) {
        const uint32_t lbase = (uint32_t)(uintptr_t)(lds_ptr)base;
#pragma unroll
        for (int i = 0; i < GA; ++i) {
            const uint32_t dst = __builtin_amdgcn_readfirstlane(lbase + (wave * GA + i) * 1024);
            glds16(dst, srcA[i] + (size_t)kt * BK);
        }
#pragma unroll
        for (int i = 0; i < GW; ++i) {
            const uint32_t dst = __builtin_amdgcn_readfirstlane(lbase + BMT * ROWB + (wave * GW + i) * 1024);
            glds16(dst, srcW[i] + (size_t)kt * BK);
        }
    };
    // Fragment read of logical 16-byte chunk c of tile row r goes to slot c ^ ((r >> 1) & 7).  A 256-byte bank row holds two tile rows, and a
    // ds_read_b128 is served 16 lanes (= 16 consecutive tile rows, one k-half) per cycle: bit 0 of r picks the half of the bank row, bits 1..3 the
    // slot inside it, so the 16 lanes hit 16 different 16-byte bank groups (slot ^ (r & 7) paired rows r and r + 8: 2-way conflicts).
    // Fragments are double-buffered: the reads of k-slice s+1 are issued before the MFMAs of slice s.
    auto compute = [&](const unsigned char *sa) {
        const unsigned char *sb = sa + BMT * ROWB;
        uint4 fa[2][2], fb[2][2];
        auto frags = [&](int s, uint4 (&xa)[2], uint4 (&xb)[2]) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ra = wm * 64 + i * 32 + lr, rb = wn * 64 + i * 32 + lr;
                xa[i] = *reinterpret_cast<const uint4 *>(sa + ra * ROWB + (((s * 2 + lh) ^ ((ra >> 1) & 7)) << 4));
                xb[i] = *reinterpret_cast<const uint4 *>(sb + rb * ROWB + (((s * 2 + lh) ^ ((rb >> 1) & 7)) << 4));
            }
        };
        frags(0, fa[0], fb[0]);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s + 1 < 4) frags(s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (sizeof(T) == 2) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[s & 1][i]), __builtin_bit_cast(bf16x8, fb[s & 1][j]), acc[i][j], 0, 0, 0);
                    } else {
                        const f32x4 a4 = __builtin_bit_cast(f32x4, fa[s & 1][i]), b4 = __builtin_bit_cast(f32x4, fb[s & 1][j]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b4[e], acc[i][j], 0, 0, 0);
                    }
                }
        }
    };
    static_assert(GA + GW == 6, "the counted wait below assumes six LDS-DMA instructions per wave and tile");
    const int nkt = g.K / BK;
    issue(0, lds0);
    if (nkt > 1) issue(1, lds1);
    auto step = [&](int kt, const unsigned char *cur, unsigned char *nxt2) {
        if (kt + 1 < nkt)
            asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + 2 < nkt) issue(kt + 2, nxt2);
        compute(cur);
    };
    for (int kt = 0; kt < nkt; kt += 3) {
        step(kt, lds0, lds2);
        if (kt + 1 < nkt) step(kt + 1, lds1, lds0);
        if (kt + 2 < nkt) step(kt + 2, lds2, lds1);
    }
    __syncthreads();  // all tiles consumed, nothing in flight: the stages become the epilogue's staging space
    // ---- epilogue (see gemm_vec_epilogue) ------------------------------------------------------------------------------------------------
    if (EPI != 0 || !g.vec_epi) {
        gemm_epilogue<EPI, false>(g, acc, bm0, bn0, wm, wn, lr, lh);
        return;
    }
    // every wave is past the last barrier: both stages are free; a wave stages 32 rows x 68 floats = 8.5 KB
    float *stg = reinterpret_cast<float *>((wave < WM ? lds0 : lds1) + (wave % WM) * (STAGE / WM));
    gemm_vec_epilogue<32>(g, acc, stg, bm0, bn0, wm, wn, lane);
}

// ---- dW = dY^T . X with direct-to-LDS staging (bf16) ----------------------------------------------------------------------------------
// Both operands are stored contraction-major ([Kc][M] and [Kc][N], Kc = tokens): a K-step stages 64 token rows x 128 columns of each in its
// natural image (256-byte rows, four rows per 1 KiB LDS-DMA instruction) and the MFMA fragments - eight consecutive tokens of one column -
// come from ds_read_b64_tr_b16.  A transposing read covers 4 token rows x 64 bytes per 32 lanes; with 256-byte rows those four rows share
// their banks, so the 64-byte block index is XORed with (token row & 3) on both sides (DMA source address / fragment address).  Two LDS
// stages as distinct objects, split-K over the tokens with fp32 atomics into the gradient (as the register-staged form it replaces, which
// ran at ~0.4 PF with one stage and two barriers per K-step).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gemm_tn_glds_kernel(GemmArgs g) {
    typedef bf16_t T;
    constexpr int BKT = 64, STAGE = 2 * BKT * 256;  // 32 KB: A image then W image
    __shared__ __attribute__((aligned(16))) unsigned char lds0[STAGE];
    __shared__ __attribute__((aligned(16))) unsigned char lds1[STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
    const int nbn = (g.N + BN - 1) / BN, nbm = (g.M + BM - 1) / BM, nwg = nbn * nbm;
    int pid = blockIdx.x, kslice = 0;
    if (g.ksplit > 1) {
        kslice = pid / nwg;
        pid -= kslice * nwg;
    }
    {
        const int q = nwg / 8, r = nwg % 8, xcd = pid % 8, idx = pid / 8;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bm0 = (pid / nbn) * BM, bn0 = (pid % nbn) * BN;
    int nkt = g.K / BKT, kt_begin = 0;
    if (g.ksplit > 1) {
        const int per = (nkt + g.ksplit - 1) / g.ksplit;
        kt_begin = kslice * per;
        nkt = min(nkt, kt_begin + per);
        if (kt_begin >= nkt) return;  // uniform per workgroup, before any barrier
    }
    const T *A = reinterpret_cast<const T *>(g.A);
    const T *W = reinterpret_cast<const T *>(g.W);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // LDS-DMA instruction j = wave * 4 + i covers token rows 4j .. 4j+3 of the stage; lane l lands at row 4j + l/16, 16-byte unit l%16, which
    // holds logical unit (l%16) ^ ((row & 3) << 2).  Columns beyond M / N are clamped to the last whole chunk (their outputs are never stored).
    const T *srcA[4], *srcW[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 4 + (lane >> 4);
        const int u = (lane & 15) ^ ((r & 3) << 2);
        srcA[i] = A + (size_t)r * g.lda + min(bm0 + u * 8, g.M - 8);
        srcW[i] = W + (size_t)r * g.ldw + min(bn0 + u * 8, g.N - 8);
    }
    typedef __attribute__((address_space(3))) void *lds_ptr;
    typedef const __attribute__((address_space(1))) void *glb_ptr;
    auto issue = [&](int kt, unsigned char *base) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((glb_ptr)(srcA[i] + (size_t)kt * BKT * g.lda), (lds_ptr)(base + (wave * 4 + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr)(srcW[i] + (size_t)kt * BKT * g.ldw), (lds_ptr)(base + BKT * 256 + (wave * 4 + i) * 1024), 16, 0, 0);
        }
    };
    typedef __attribute__((ext_vector_type(4))) short s4;
    typedef __attribute__((address_space(3))) s4 *lds_s4;
    const int i16 = lane & 15, g1 = (lane >> 4) & 1;
    // fragment of output rows rowbase .. rowbase+31, contraction slice s: element j of lane half h is token 16 s + 8 h + j
    auto frag = [&](const unsigned char *img, int rowbase, int s) -> uint4 {
        const int m = 16 * s + 8 * lh + (i16 >> 2), bc = (rowbase + 16 * g1 + 4 * (i16 & 3)) * 2;
        const int o = m * 256 + ((((bc >> 6) ^ (m & 3))) << 6) + (bc & 63);   // rows m and m + 4 share (m & 3)
        union { s4 v[2]; uint4 u; } r;
        r.v[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(img + o));
        r.v[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(img + o + 4 * 256));
        return r.u;
    };
    auto compute = [&](const unsigned char *sa) {
        const unsigned char *sb = sa + BKT * 256;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            uint4 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = frag(sa, wm * 64 + i * 32, s);
                fb[i] = frag(sb, wn * 64 + i * 32, s);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
        }
    };
    issue(kt_begin, lds0);
    __syncthreads();
    int kt = kt_begin;
    for (; kt + 2 <= nkt; kt += 2) {
        issue(kt + 1, lds1);
        compute(lds0);
        __syncthreads();
        if (kt + 2 < nkt) issue(kt + 2, lds0);
        compute(lds1);
        __syncthreads();
    }
    if (kt < nkt) compute(lds0);
    gemm_accum_epilogue(g, acc, bm0, bn0, wm, wn, lr, lh);
}

// ---- persistent 256x128 kernel: the three-stage LDS-DMA ring runs ACROSS output tiles ------------------------------------------------------
// The per-tile fixed cost of the three-stage kernel is ~9 us (workgroup dispatch, address set-up, the first DMA round trip, epilogue): two
// thirds of a K = 512 tile (8 K-steps, 4.3 us of MFMA).  Here one workgroup per CU walks a list of tiles (XCD-contiguous ranges, as the
// non-persistent order) and its producer cursor simply keeps going: while tile i is in its last K-steps and its epilogue, the first two
// K-tiles of tile i+1 are already in flight.  The epilogue stages through the ring slot the last K-step just freed (16 rows per pass), behind
// one extra barrier; the step after an epilogue drains vmcnt to 0 (output stores share the counter with the DMA).
template <typename T, int EPI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void gemm_nt_pers_kernel(GemmArgs g) {
    constexpr int BK = ROWB / sizeof(T);
    constexpr int WM = 4, BMT = 256, NW = 8, GA = BMT / 8 / NW, GW = BN / 8 / NW, STAGE = (BMT + BN) * ROWB;
    static_assert(GA + GW == 6, "counted wait assumes six LDS-DMA instructions per wave and K-tile");
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
    const int nbn = (g.N + BN - 1) / BN, nbm = (g.M + BMT - 1) / BMT, ntiles = nbn * nbm;
    // my tiles: XCD x = blockIdx % 8 owns a contiguous range, its workgroups interleave inside it
    const int w = blockIdx.x, nwg = gridDim.x, xcd = w % 8, j0 = w / 8;
    const int per_xcd = (nwg - xcd + 7) / 8;
    const int tq = ntiles / 8, tr = ntiles % 8;
    const int t_start = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq, t_count = tq + (xcd < tr ? 1 : 0);
    const int n_my = j0 < t_count ? (t_count - j0 + per_xcd - 1) / per_xcd : 0;
    const int nkt = g.K / BK, total = n_my * nkt;
    if (total == 0) return;
    const T *A = reinterpret_cast<const T *>(g.A);
    const T *W = reinterpret_cast<const T *>(g.W);
    typedef __attribute__((address_space(3))) void *lds_ptr;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr)lds;

    // ---- producer cursor ----
    const int grow = lane >> 3;
    const T *srcA[GA], *srcW[GW];
    int p_tile = 0, p_kt = 0;
    auto set_tile = [&](int ti) {
        const int t = t_start + j0 + ti * per_xcd, bm0 = (t / nbn) * BMT, bn0 = (t % nbn) * BN;
#pragma unroll
        for (int i = 0; i < GA; ++i) {
            const int row = (wave * GA + i) * 8 + grow, gslot = (lane & 7) ^ ((row >> 1) & 7);
            srcA[i] = A + (size_t)min(bm0 + row, g.M - 1) * g.lda + gslot * (16 / sizeof(T));
        }
#pragma unroll
        for (int i = 0; i < GW; ++i) {
            const int row = (wave * GW + i) * 8 + grow, gslot = (lane & 7) ^ ((row >> 1) & 7);
            srcW[i] = W + (size_t)min(bn0 + row, g.N - 1) * g.ldw + gslot * (16 / sizeof(T));
        }
    };
    auto issue = [&](int slot) {   // next K-tile of the producer cursor -> ring slot
        const uint32_t lb = lds_base + slot * STAGE;
#pragma unroll
        for (int i = 0; i < GA; ++i) {
            const uint32_t dst = __builtin_amdgcn_readfirstlane(lb + (wave * GA + i) * 1024);
            glds16(dst, srcA[i] + (size_t)p_kt * BK);
        }
#pragma unroll
        for (int i = 0; i < GW; ++i) {
            const uint32_t dst = __builtin_amdgcn_readfirstlane(lb + BMT * ROWB + (wave * GW + i) * 1024);
            glds16(dst, srcW[i] + (size_t)p_kt * BK);
        }
        if (++p_kt == nkt) {
            p_kt = 0;
            if (++p_tile < n_my) set_tile(p_tile);
        }
    };

    f32x16 acc[2][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    };
    auto compute = [&](const unsigned char *sa) {
        const unsigned char *sb = sa + BMT * ROWB;
        uint4 fa[2][2], fb[2][2];
        auto frags = [&](int s, uint4 (&xa)[2], uint4 (&xb)[2]) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ra = wm * 64 + i * 32 + lr, rb = wn * 64 + i * 32 + lr;
                xa[i] = *reinterpret_cast<const uint4 *>(sa + ra * ROWB + (((s * 2 + lh) ^ ((ra >> 1) & 7)) << 4));
                xb[i] = *reinterpret_cast<const uint4 *>(sb + rb * ROWB + (((s * 2 + lh) ^ ((rb >> 1) & 7)) << 4));
            }
        };
        frags(0, fa[0], fb[0]);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s + 1 < 4) frags(s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (sizeof(T) == 2) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[s & 1][i]), __builtin_bit_cast(bf16x8, fb[s & 1][j]), acc[i][j], 0, 0, 0);
                    } else {
                        const f32x4 a4 = __builtin_bit_cast(f32x4, fa[s & 1][i]), b4 = __builtin_bit_cast(f32x4, fb[s & 1][j]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b4[e], acc[i][j], 0, 0, 0);
                    }
                }
        }
    };

    set_tile(0);
    issue(0);
    if (total > 1) issue(1);
    zero_acc();
    int c_tile = 0, c_kt = 0, slot = 0;
    bool drained = false;   // the previous step ended with an epilogue: its stores are still counted in vmcnt
    for (int q = 0; q < total; ++q) {
        if (!drained && q + 1 < total)
            asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        drained = false;
        if (q + 2 < total) issue(slot == 0 ? 2 : slot - 1);   // slot of step q + 2 = (slot + 2) % 3
        compute(lds + slot * STAGE);
        if (++c_kt == nkt) {
            c_kt = 0;
            const int t = t_start + j0 + c_tile * per_xcd, bm0 = (t / nbn) * BMT, bn0 = (t % nbn) * BN;
            if (EPI != 0 || !g.vec_epi) {
                gemm_epilogue<EPI, false>(g, acc, bm0, bn0, wm, wn, lr, lh);
            } else {
                asm volatile("s_barrier" ::: "memory");   // every wave is done reading this slot: it becomes the staging space
                float *stg = reinterpret_cast<float *>(lds + slot * STAGE + wave * (STAGE / NW));
                gemm_vec_epilogue<16>(g, acc, stg, bm0, bn0, wm, wn, lane);
            }
            zero_acc();
            ++c_tile;
            drained = true;
        }
        slot = slot == 2 ? 0 : slot + 1;
    }
}

// ---- 256x256 tile, 8 waves x (128 x 64), two 64 KB LDS-DMA stages -----------------------------------------------------------------------
// A CU takes in at most ~60-70 GB/s through the LDS-DMA path.  A 256x128 K-step stages 48 KB for 1024 MFMA cycles per SIMD (~0.54 us at
// 1.9 GHz): 89 GB/s would be needed, so those kernels run staging-bound at ~45 % MFMA utilisation however deep the ring is.  The 256x256
// tile stages 64 KB for 2048 MFMA cycles (59 GB/s): the first shape on which the matrix cores, not the staging path, can set the pace.
template <typename T, int EPI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void gemm_nt_256_kernel(GemmArgs g) {
    constexpr int BK = ROWB / sizeof(T);
    constexpr int BT = 256, STAGE = 2 * BT * ROWB;   // 64 KB: A image (256 rows) then W image (256 rows)
    __shared__ __attribute__((aligned(16))) unsigned char lds0[STAGE];
    __shared__ __attribute__((aligned(16))) unsigned char lds1[STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, lr = lane & 31, lh = lane >> 5;
    const int nbn = (g.N + BT - 1) / BT, nbm = (g.M + BT - 1) / BT, nwg = nbn * nbm;
    int pid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = pid % 8, idx = pid / 8;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bm0 = (pid / nbn) * BT, bn0 = (pid % nbn) * BT;
    const T *A = reinterpret_cast<const T *>(g.A);
    const T *W = reinterpret_cast<const T *>(g.W);
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int grow = lane >> 3;
    const T *srcA[4], *srcW[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + grow, gslot = (lane & 7) ^ ((row >> 1) & 7);
        srcA[i] = A + (size_t)min(bm0 + row, g.M - 1) * g.lda + gslot * (16 / sizeof(T));
        srcW[i] = W + (size_t)min(bn0 + row, g.N - 1) * g.ldw + gslot * (16 / sizeof(T));
    }
    typedef __attribute__((address_space(3))) void *lds_ptr;
    typedef const __attribute__((address_space(1))) void *glb_ptr;
    auto issue = [&](int kt, unsigned char *base) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((glb_ptr)(srcA[i] + (size_t)kt * BK), (lds_ptr)(base + (wave * 4 + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr)(srcW[i] + (size_t)kt * BK), (lds_ptr)(base + BT * ROWB + (wave * 4 + i) * 1024), 16, 0, 0);
        }
    };
    auto compute = [&](const unsigned char *sa) {
        const unsigned char *sb = sa + BT * ROWB;
        uint4 fa[2][4], fb[2][2];
        auto frags = [&](int s, uint4 (&xa)[4], uint4 (&xb)[2]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ra = wm * 128 + i * 32 + lr;
                xa[i] = *reinterpret_cast<const uint4 *>(sa + ra * ROWB + (((s * 2 + lh) ^ ((ra >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int rb = wn * 64 + j * 32 + lr;
                xb[j] = *reinterpret_cast<const uint4 *>(sb + rb * ROWB + (((s * 2 + lh) ^ ((rb >> 1) & 7)) << 4));
            }
        };
        frags(0, fa[0], fb[0]);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s + 1 < 4) frags(s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (sizeof(T) == 2) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[s & 1][i]), __builtin_bit_cast(bf16x8, fb[s & 1][j]), acc[i][j], 0, 0, 0);
                    } else {
                        const f32x4 a4 = __builtin_bit_cast(f32x4, fa[s & 1][i]), b4 = __builtin_bit_cast(f32x4, fb[s & 1][j]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b4[e], acc[i][j], 0, 0, 0);
                    }
                }
        }
    };
    const int nkt = g.K / BK;
    issue(0, lds0);
    __syncthreads();
    int kt = 0;
    for (; kt + 2 <= nkt; kt += 2) {
        issue(kt + 1, lds1);
        compute(lds0);
        __syncthreads();
        if (kt + 2 < nkt) issue(kt + 2, lds0);
        compute(lds1);
        __syncthreads();
    }
    if (kt < nkt) {
        compute(lds0);
        __syncthreads();
    }
    // epilogue: the wave's 128 x 64 block as two 64 x 64 halves through the shared routines (constant indices only: a run-time index
    // into acc would push all 128 accumulator registers to scratch)
    auto epi_half = [&](auto ihc) {
        constexpr int ih = decltype(ihc)::value;
        f32x16 half[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) half[i][j] = acc[2 * ih + i][j];
        const int bmh = bm0 + wm * 128 + ih * 64;
        if (EPI != 0 || !g.vec_epi) {
            gemm_epilogue<EPI, false>(g, half, bmh, bn0, 0, wn, lr, lh);
        } else {
            float *stg = reinterpret_cast<float *>((wave < 4 ? lds0 : lds1) + (wave & 3) * (STAGE / 4));
            gemm_vec_epilogue<32>(g, half, stg, bmh, bn0, 0, wn, lane);
        }
    };
    epi_half(std::integral_constant<int, 0>{});
    epi_half(std::integral_constant<int, 1>{});
}

// ---- persistent 256x256 kernel on a ring of five 32 KB half-stages (round 2) ------------------------------------------------------------
// The short-K GEMMs of the training steps (K = 512..768: 8-12 K-tiles per output tile) are bound by what a CU takes in through the LDS-DMA
// path - 33 GB/s from the Infinity Cache, 66-73 from its XCD's L2, and the 256x128 persistent ring sustains ~35 with ~75 % L2 hits - not by
// the matrix cores (31 % busy).  A 256x256 tile needs 64 KB per K-tile for twice the MFMA work of the 256x128 tile's 48 KB: 1.5x the flops
// per staged byte.  Three whole 64 KB stages do not fit the 160 KB of LDS, so the ring runs on HALF-stages: unit 2q is the A image
// ([256 rows][128 B], the layout of the other LDS-DMA kernels) of K-step q, unit 2q+1 its W image, unit u lives in slot u % 5.  While
// K-step q is computed (two slots), units A(q+1), W(q+1), A(q+2) are in flight in the other three: 96 KB, as much as the three-stage ring.
// Per step: s_waitcnt vmcnt(4) (only the four DMA instructions of the youngest unit may be outstanding), ONE raw barrier, issue W(q+2) and
// A(q+3) into the two slots step q-1 just freed, 32 MFMAs per wave.  The ring runs ACROSS output tiles (XCD-contiguous tile ranges, one
// workgroup per CU); the epilogue stages through the two slots its last K-step read (behind one extra barrier).
template <typename T, int EPI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void gemm_nt_pers256_kernel(GemmArgs g) {
    static_assert(sizeof(T) == 2 && EPI == 0, "bf16 row-major operands, plain epilogue");
    constexpr int BK = ROWB / sizeof(T);
    constexpr int BT = 256, UNIT = BT * ROWB, NSLOT = 5;   // 32 KB per unit
    __shared__ __attribute__((aligned(16))) unsigned char lds[NSLOT * UNIT];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, lr = lane & 31, lh = lane >> 5;
    const int nbn = (g.N + BT - 1) / BT, nbm = (g.M + BT - 1) / BT, ntiles = nbn * nbm;
    // my tiles: XCD x = blockIdx % 8 owns a contiguous range, its workgroups interleave inside it
    const int w = blockIdx.x, nwg = gridDim.x, xcd = w % 8, j0 = w / 8;
    const int per_xcd = (nwg - xcd + 7) / 8;
    const int tq = ntiles / 8, tr = ntiles % 8;
    const int t_start = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq, t_count = tq + (xcd < tr ? 1 : 0);
    const int n_my = j0 < t_count ? (t_count - j0 + per_xcd - 1) / per_xcd : 0;
    const int nkt = g.K / BK, total = n_my * nkt;
    if (total == 0) return;
    const T *A = reinterpret_cast<const T *>(g.A);
    const T *W = reinterpret_cast<const T *>(g.W);
    typedef __attribute__((address_space(3))) void *lds_ptr;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr)lds;

    // ---- producer: one cursor per operand (A runs one K-step ahead of W) ----
    const int grow = lane >> 3;
    const T *srcA[4], *srcW[4];
    int a_tile = 0, a_kt = 0, w_tile = 0, w_kt = 0, p_slot = 0;
    auto tile_origin = [&](int ti, int &bm0, int &bn0) {
        const int t = t_start + j0 + ti * per_xcd;
        bm0 = (t / nbn) * BT;
        bn0 = (t % nbn) * BT;
    };
    auto set_a = [&](int ti) {
        int bm0, bn0;
        tile_origin(ti, bm0, bn0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (wave * 4 + i) * 8 + grow, gslot = (lane & 7) ^ ((row >> 1) & 7);
            srcA[i] = A + (size_t)min(bm0 + row, g.M - 1) * g.lda + gslot * (16 / sizeof(T));
        }
    };
    auto set_w = [&](int ti) {
        int bm0, bn0;
        tile_origin(ti, bm0, bn0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (wave * 4 + i) * 8 + grow, gslot = (lane & 7) ^ ((row >> 1) & 7);
            srcW[i] = W + (size_t)min(bn0 + row, g.N - 1) * g.ldw + gslot * (16 / sizeof(T));
        }
    };
    // next A / W unit -> slot p_slot; nothing once the work list is exhausted.  Two functions with a fixed call order (A W A | W A | W A ...),
    // not one that picks the operand from the unit's parity: indexed that way the source-pointer arrays went to scratch memory, and every
    // reload drained vmcnt to 0 in front of the next DMA
    int a_done = 0, w_done = 0;   // K-steps issued per operand
    auto next_slot = [&]() { p_slot = p_slot == NSLOT - 1 ? 0 : p_slot + 1; };
    auto issue_a = [&]() {
        if (a_done >= total) return;
        const uint32_t lb = lds_base + p_slot * UNIT;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(__builtin_amdgcn_readfirstlane(lb + (wave * 4 + i) * 1024), srcA[i] + (size_t)a_kt * BK);
        if (++a_kt == nkt) {
            a_kt = 0;
            if (++a_tile < n_my) set_a(a_tile);
        }
        ++a_done;
        next_slot();
    };
    auto issue_w = [&]() {
        if (w_done >= total) return;
        const uint32_t lb = lds_base + p_slot * UNIT;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(__builtin_amdgcn_readfirstlane(lb + (wave * 4 + i) * 1024), srcW[i] + (size_t)w_kt * BK);
        if (++w_kt == nkt) {
            w_kt = 0;
            if (++w_tile < n_my) set_w(w_tile);
        }
        ++w_done;
        next_slot();
    };

    f32x16 acc[4][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    };
    auto compute = [&](const unsigned char *sa, const unsigned char *sb) {
        uint4 fa[2][4], fb[2][2];
        auto frags = [&](int s4, uint4 (&xa)[4], uint4 (&xb)[2]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ra = wm * 128 + i * 32 + lr;
                xa[i] = *reinterpret_cast<const uint4 *>(sa + ra * ROWB + (((s4 * 2 + lh) ^ ((ra >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int rb = wn * 64 + j * 32 + lr;
                xb[j] = *reinterpret_cast<const uint4 *>(sb + rb * ROWB + (((s4 * 2 + lh) ^ ((rb >> 1) & 7)) << 4));
            }
        };
        frags(0, fa[0], fb[0]);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            if (s4 + 1 < 4) frags(s4 + 1, fa[(s4 + 1) & 1], fb[(s4 + 1) & 1]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[s4 & 1][i]), __builtin_bit_cast(bf16x8, fb[s4 & 1][j]), acc[i][j], 0, 0, 0);
        }
    };

    set_a(0);
    set_w(0);
    issue_a();   // A(0)
    issue_w();   // W(0)
    issue_a();   // A(1)
    zero_acc();
    int c_tile = 0, c_kt = 0, slot_a = 0;   // slot of A(q); W(q) sits in the next one
    bool drained = false;   // the previous step ended with an epilogue: its stores are still counted in vmcnt
    for (int q = 0; q < total; ++q) {
        // A(q), W(q) have landed: of everything issued, only A(q+1) - the youngest unit, four instructions per wave - may be outstanding
        if (!drained && q + 1 < total)
            asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        drained = false;
        issue_w();   // W(q+1) -> the slot A(q-1) left
        issue_a();   // A(q+2) -> the slot W(q-1) left
        const int slot_w = slot_a == NSLOT - 1 ? 0 : slot_a + 1;
        compute(lds + slot_a * UNIT, lds + slot_w * UNIT);
        if (++c_kt == nkt) {
            c_kt = 0;
            int bm0, bn0;
            tile_origin(c_tile, bm0, bn0);
            // the wave's 128 x 64 block as two 64 x 64 halves through the shared routines (constant indices only: a run-time index into
            // acc would push all 128 accumulator registers to scratch)
            const bool vec = g.vec_epi;
            if (vec) asm volatile("s_barrier" ::: "memory");   // every wave is done reading these two slots: they become the staging space
            auto epi_half = [&](auto ihc) {
                constexpr int ih = decltype(ihc)::value;
                f32x16 half[2][2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) half[i][j] = acc[2 * ih + i][j];
                const int bmh = bm0 + wm * 128 + ih * 64;
                if (!vec) {
                    gemm_epilogue<EPI, false>(g, half, bmh, bn0, 0, wn, lr, lh);
                } else {
                    float *stg = reinterpret_cast<float *>(lds + (wave < 4 ? slot_a : slot_w) * UNIT + (wave & 3) * (UNIT / 4));
                    gemm_vec_epilogue<16>(g, half, stg, bmh, bn0, 0, wn, lane);
                }
            };
            epi_half(std::integral_constant<int, 0>{});
            epi_half(std::integral_constant<int, 1>{});
            zero_acc();
            ++c_tile;
            drained = true;
        }
        slot_a = slot_a + 2 >= NSLOT ? slot_a + 2 - NSLOT : slot_a + 2;
    }
}

// ---- register epilogue of the swapped 16x16x32 layout (gemm_nt_pp_kernel) ----------------------------------------------------------------
// Lane (lm = lane & 15, lq = lane >> 4) of accumulator [mb][nb] holds C[m0 + mb*16 + lm][n0 + nb*16 + 4*lq + e], e = 0..3: four consecutive
// columns of one row.  fp32 rows are stored / loaded 16 bytes per lane as they stand (16 rows x 64 contiguous bytes per wave instruction).
// bf16: the two packed dwords of blocks nb = 2p and 2p+1 are exchanged between neighbouring 16-lane rows (v_permlane16_swap: odd rows of the
// first operand <-> even rows of the second), after which lane row r holds EIGHT consecutive columns: block 2p + (r & 1), columns 8 (r >> 1) ..;
// the exchange is an involution, so a 16-byte bf16 load at the swapped position followed by the same exchange yields the original layout
// (the saved pre-activation of aux_mode 2).  Every global access is a raw buffer access whose out-of-range lanes (rows >= M, columns >= N:
// offset forced beyond num_records) are dropped by the hardware range check: no divergent branch, and the instruction count per tile is exact
// (the caller's counted vmcnt wait relies on it).  Arithmetic and its order are gemm_vec_epilogue's.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
__device__ __forceinline__ void pp_swap2(uint32_t &x, uint32_t &y) {
    const u32x2_t r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    x = r[0];
    y = r[1];
}
// MODE (compile-time: one kernel instantiation per epilogue form - with every form in one body the kernel was > 100 KB of code, more than the
// instruction cache, and a plain bf16 epilogue took ~10 us per tile walking around the other forms' blocks)
enum { PP_OBF = 1, PP_GELU = 2, PP_AUX1 = 4, PP_AUX2 = 8, PP_RES = 16, PP_SCALE = 32, PP_DEFER = 64, PP_AUX3 = 128, PP_AUX4 = 256 };
template <int MODE>
__device__ __forceinline__ void gemm_pp_epilogue(const GemmArgs &g, const f32x4 (&acc)[8][4], int m0, int n0, int lane, float bv, void *dst = nullptr,
                                                 int ldd = 0) {
    const int lm = lane & 15, lq = lane >> 4;
    constexpr bool obf = MODE & PP_OBF, do_gelu = MODE & PP_GELU, has_res = MODE & PP_RES, has_scale = MODE & PP_SCALE;
    constexpr int aux_mode = (MODE & PP_AUX1) ? 1 : ((MODE & PP_AUX2) ? 2 : ((MODE & PP_AUX3) ? 3 : ((MODE & PP_AUX4) ? 4 : 0)));
    const bool do_round = g.flags & ACAI_GEMM_ROUND_BF16;
    const bool pre_round = do_round && (do_gelu || aux_mode != 0 || has_res || !obf);
    constexpr int es = obf ? 2 : 4;
    // (dst / ldd: the deferred forms write their first, plain bf16 pass somewhere else than C - see gemm_nt_pp_kernel)
    void *const Cp = dst ? dst : g.C;
    const int ldc = dst ? ldd : g.ldc;
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(Cp, 0, (int)((size_t)g.M * ldc * es), 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(has_res ? g.residual : reinterpret_cast<const float *>(g.C)), 0,
                                                                        has_res ? (int)((size_t)g.M * g.ldr * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(aux_mode ? g.aux : g.C, 0, aux_mode ? (int)((size_t)g.M * g.ldaux * es) : 0, 0x00020000);
    constexpr uint32_t OOB = 0xFFFFFFF0u;
    // swapped (bf16) position of this lane inside a block pair: block 2p + (lq & 1), columns 8 (lq >> 1) .. + 7
    const int sw_col = n0 + (lq & 1) * 16 + (lq >> 1) * 8;
    f32x4 bias4[4];
    bool colok[4];
    auto take_bias = [&](int p) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int nb = 2 * p + t, col = n0 + nb * 16 + 4 * lq;
            colok[nb] = col < g.N;
            // lane L of `bv` holds bias[n0 + L] (zero beyond N / without a bias), loaded by the caller a K-step ahead: a vector-memory load HERE
            // would be waited for behind the LDS-DMA just issued (vmcnt counts in order) - 3.6 us per tile on the decoder's in-projection
#pragma unroll
            for (int e = 0; e < 4; ++e) bias4[nb][e] = __shfl(bv, nb * 16 + 4 * lq + e);
        }
    };
    // Loop order.  Default: block pair p outermost (only its eight per-column constants are live), then two halves of four 16-row blocks.
    // P_INNER (the two GELU forms, bf16 output): pairs of 16-row blocks outermost, p innermost - a lane's two 64-byte halves of a 128-byte
    // line (the stores of C and of the kept pre-activation, the loads of the saved pre-activation) are then issued back to back.  With p
    // outermost they were half an epilogue apart, and under these forms' 1.6 GB of stores / loads per launch L2 evicted lines half written
    // and fetched lines twice (tools/pmc_gemm.sh, M = 131072, N = 3072, K = 512: forward WRITE_SIZE 2.24 GB for 1.61 GB of output - 743 -> 633 us
    // with this order; gelu' form FETCH_SIZE 2.2 GB for 0.94 GB of operands, WRITE_SIZE 1.08 GB for 0.81 GB).
    constexpr bool P_INNER = obf && (aux_mode != 0);   // (the plain bf16 forms measured the same either way: their lines survive in L2)
    if constexpr (P_INNER) {
        take_bias(0);
        take_bias(1);
    }
#pragma unroll
    for (int it_o = 0; it_o < 4; ++it_o) {
        if constexpr (!P_INNER) {
            if ((it_o & 1) == 0) take_bias(it_o >> 1);
        }
        // Every load of this group first (the saved pre-activation of aux_mode 2, the fp32 residual: 4 or 8 x 16 bytes per lane), then
        // the arithmetic and the stores: load -> use -> store per 16 rows made a chain of 16 dependent memory round trips per epilogue
        // (~30 us per tile on the decoder's gelu' GEMM).
        // (four (block, p) slots at a time: all eight blocks of a p spilled the residual form)
        u32x4_t prex[8][2], prer[8][2];
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            const int p = P_INNER ? (sl & 1) : (it_o >> 1);
            const int mb = P_INNER ? 2 * it_o + (sl >> 1) : 4 * (it_o & 1) + sl;
            const int row = m0 + mb * 16 + lm;
            const bool rowok = row < g.M;
            if constexpr (aux_mode == 2 || aux_mode == 4) {
                if constexpr (obf) {
                    const bool swok = rowok && (sw_col + p * 32) < g.N;
                    prex[mb][P_INNER ? p : 0] = __builtin_amdgcn_raw_buffer_load_b128(rx, swok ? (uint32_t)(((size_t)row * g.ldaux + sw_col + p * 32) * 2) : OOB, 0, 0);
                } else {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int nb = 2 * p + t, col = n0 + nb * 16 + 4 * lq;
                        prex[mb][t] = __builtin_amdgcn_raw_buffer_load_b128(rx, (rowok && colok[nb]) ? (uint32_t)(((size_t)row * g.ldaux + col) * 4) : OOB, 0, 0);
                    }
                }
            }
            if constexpr (has_res) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int nb = 2 * p + t, col = n0 + nb * 16 + 4 * lq;
                    prer[mb][t] = __builtin_amdgcn_raw_buffer_load_b128(rr, (rowok && colok[nb]) ? (uint32_t)(((size_t)row * g.ldr + col) * 4) : OOB, 0, 0);
                }
            }
        }
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            const int p = P_INNER ? (sl & 1) : (it_o >> 1);
            const int mb = P_INNER ? 2 * it_o + (sl >> 1) : 4 * (it_o & 1) + sl;
            const int row = m0 + mb * 16 + lm;
            const bool rowok = row < g.M;
            f32x4 v[2];
            uint32_t rp[2][2] = {};
            const bool swok = rowok && (sw_col + p * 32) < g.N;
            const uint32_t off_sw_c = swok ? (uint32_t)(((size_t)row * ldc + sw_col + p * 32) * 2) : OOB;
            const uint32_t off_sw_x = swok ? (uint32_t)(((size_t)row * g.ldaux + sw_col + p * 32) * 2) : OOB;
            float a0[2][4] = {};
            if constexpr (aux_mode == 2 || aux_mode == 4) {
                if constexpr (obf) {
                    const u32x4_t L = prex[mb][P_INNER ? p : 0];
                    uint32_t x0 = L[0], x1 = L[1], y0 = L[2], y1 = L[3];
                    pp_swap2(x0, y0);
                    pp_swap2(x1, y1);
                    a0[0][0] = __uint_as_float(x0 << 16); a0[0][1] = __uint_as_float(x0 & 0xFFFF0000u);
                    a0[0][2] = __uint_as_float(x1 << 16); a0[0][3] = __uint_as_float(x1 & 0xFFFF0000u);
                    a0[1][0] = __uint_as_float(y0 << 16); a0[1][1] = __uint_as_float(y0 & 0xFFFF0000u);
                    a0[1][2] = __uint_as_float(y1 << 16); a0[1][3] = __uint_as_float(y1 & 0xFFFF0000u);
                } else {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const u32x4_t L = prex[mb][t];
#pragma unroll
                        for (int e = 0; e < 4; ++e) a0[t][e] = __uint_as_float(L[e]);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int nb = 2 * p + t;
                v[t] = acc[mb][nb] + bias4[nb];
                if constexpr (has_scale) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[t][e] *= (n0 + nb * 16 + 4 * lq + e) < g.scale_cols ? g.col_scale : 1.0f;
                }
                if (pre_round) {
                    // rounded in PAIRS: one v_cvt_pk_bf16_f32 per two values and a shift / mask to widen them again (per value it was one
                    // conversion + one shift: the GELU forms' epilogues are VALU-bound); the packed words are also what aux_mode 1 stores
                    rp[t][0] = pack_bf16(v[t][0], v[t][1]);
                    rp[t][1] = pack_bf16(v[t][2], v[t][3]);
                    v[t][0] = __uint_as_float(rp[t][0] << 16); v[t][1] = __uint_as_float(rp[t][0] & 0xFFFF0000u);
                    v[t][2] = __uint_as_float(rp[t][1] << 16); v[t][3] = __uint_as_float(rp[t][1] & 0xFFFF0000u);
                }
            }
            [[maybe_unused]] float dg[2][4];   // aux_mode 3: gelu'(pre-activation), formed with the GELU below from one Phi(-|a|)
            if constexpr (aux_mode == 3) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float gv;
                        gelu_erf_both(v[t][e], gv, dg[t][e]);
                        v[t][e] = gv;
                    }
            }
            if constexpr (aux_mode == 1 || aux_mode == 3) {   // keep the pre-activation (1) / its GELU derivative (3)
                if constexpr (obf) {
                    uint32_t x0, x1, y0, y1;
                    if constexpr (aux_mode == 3) {
                        x0 = pack_bf16(dg[0][0], dg[0][1]); x1 = pack_bf16(dg[0][2], dg[0][3]); y0 = pack_bf16(dg[1][0], dg[1][1]); y1 = pack_bf16(dg[1][2], dg[1][3]);
                    } else if (pre_round) {
                        x0 = rp[0][0]; x1 = rp[0][1]; y0 = rp[1][0]; y1 = rp[1][1];
                    } else {
                        x0 = pack_bf16(v[0][0], v[0][1]); x1 = pack_bf16(v[0][2], v[0][3]); y0 = pack_bf16(v[1][0], v[1][1]); y1 = pack_bf16(v[1][2], v[1][3]);
                    }
                    pp_swap2(x0, y0);
                    pp_swap2(x1, y1);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{x0, x1, y0, y1}, rx, off_sw_x, 0, 0);
                } else {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int nb = 2 * p + t, col = n0 + nb * 16 + 4 * lq;
                        const uint32_t o = (rowok && colok[nb]) ? (uint32_t)(((size_t)row * g.ldaux + col) * 4) : OOB;
                        if constexpr (aux_mode == 3) __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{__float_as_uint(dg[t][0]), __float_as_uint(dg[t][1]), __float_as_uint(dg[t][2]), __float_as_uint(dg[t][3])}, rx, o, 0, 0);
                        else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v[t]), rx, o, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int nb = 2 * p + t, col = n0 + nb * 16 + 4 * lq;
                if constexpr (aux_mode == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[t][e] *= gelu_erf_grad(a0[t][e]);
                }
                if constexpr (aux_mode == 4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[t][e] *= a0[t][e];
                }
                if constexpr (do_gelu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if constexpr (aux_mode != 3) v[t][e] = gelu_erf(v[t][e]);   // (3: done above, together with the derivative)
                        if (do_round && !obf) v[t][e] = round_bf16(v[t][e]);   // (bf16 output: the pack below is that rounding)
                    }
                }
                if constexpr (has_res) v[t] += __builtin_bit_cast(f32x4, prer[mb][t]);
                if constexpr (!obf) {
                    const uint32_t o = (rowok && colok[nb]) ? (uint32_t)(((size_t)row * ldc + col) * 4) : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v[t]), rc, o, 0, 0);
                }
            }
            if constexpr (obf) {
                uint32_t x0 = pack_bf16(v[0][0], v[0][1]), x1 = pack_bf16(v[0][2], v[0][3]), y0 = pack_bf16(v[1][0], v[1][1]), y1 = pack_bf16(v[1][2], v[1][3]);
                pp_swap2(x0, y0);
                pp_swap2(x1, y1);
#ifdef ACAI_GEMM_ABLATE
                const int dbg = g.flags >> 8;
#else
                constexpr int dbg = 0;
#endif
                if (dbg & 128) asm volatile("" ::"v"(x0), "v"(x1), "v"(y0), "v"(y1));
                else __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{x0, x1, y0, y1}, rc, (dbg & 64) ? (off_sw_c & 0xFFFF0u) : off_sw_c, 0, 0);
            }
        }
    }
}

// ---- persistent 256x256 ring with PING-PONG wave groups and a REGISTER epilogue (round 3) -----------------------------------------------
// What round 2 measured on the short-K GEMMs of the training steps (K = 512..768): the main loop of the rings runs at ~40-45 % of the matrix
// pipe and the epilogue (LDS transposition + stores, nothing else running on the CU meanwhile) adds another 50 % on top.  Both have one cause:
// the two waves of a SIMD run the same program in lock-step behind one barrier per K-step - both read fragments, both issue MFMAs, both
// transpose and store - so the matrix pipe idles whenever "the" wave does anything else.  This kernel keeps pers256's ring (five 32 KB
// half-stage slots, LDS-DMA from inline asm, counted vmcnt waits, persistent XCD-contiguous tile lists) and changes three things:
//  1. Ping-pong: a K-step is four barrier-separated segments per wave - R(kh=0): 12 ds_read_b128, M(kh=0): 32 MFMAs, R(kh=1), M(kh=1) - and
//     waves 4-7 (the lower 128 rows of the tile; they share SIMDs with waves 0-3) run ONE SEGMENT BEHIND waves 0-3: while one wave of a
//     SIMD issues its 32 MFMAs the other reads its fragments, issues its LDS-DMA or runs its epilogue.  Fragments are single-buffered (the
//     partner's MFMAs cover the read latency), so the loop needs 48 fragment registers instead of 96.
//  2. v_mfma_f32_16x16x32_bf16 with the operands SWAPPED (W fragment as the A operand): D[n][m], a lane holds FOUR CONSECUTIVE COLUMNS of one
//     output row.  fp32 outputs store 16 bytes per lane straight from the accumulators; bf16 outputs exchange the packed halves of two
//     neighbouring 16-column blocks between the 16-lane rows (v_permlane16_swap) and also store 16 bytes per lane - 16 rows x 64 contiguous
//     bytes per wave instruction, no LDS transposition, no epilogue barrier, no staging space.  (The 16x16x32 shape also holds a higher clock
//     than 32x32x16 at equal cycles per flop on this chip: MI355X_MICROARCH.md, DVFS item 7.)
//  3. The epilogue of tile i runs in the first R segment of tile i+1 of the same wave - i.e. beside the partner's MFMAs - and its stores stay
//     in flight: they are raw buffer stores (out-of-range lanes dropped by the range check, so the instruction COUNT is exact) issued between
//     the segment's two LDS-DMA groups, and the K-step's counted wait allows for them (vmcnt counts stores and LDS-DMA together, in order).
// Segment / barrier ledger (b_s = barrier at the end of global segment s; waves 0-3 = G0, waves 4-7 = G1):
//     G0: R(q,0) = seg 4q, M(q,0) = 4q+1, R(q,1) = 4q+2, M(q,1) = 4q+3;    G1: the same one segment later (4q+1 .. 4q+4)
//     slots of K-step q are last read in seg 4q+3 (G1, reads waited with lgkmcnt(0) before b_{4q+3}) and refilled from seg 4q+4 on;
//     every wave waits for ITS pieces of K-step q+1 (vmcnt) before b_{4q+3}; the first read of K-step q+1 is in seg 4q+4.
template <typename T, int MODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void gemm_nt_pp_kernel(GemmArgs g) {
    static_assert(sizeof(T) == 2, "bf16 row-major operands");
    constexpr int BK = ROWB / sizeof(T);
    constexpr int BT = 256, UNIT = BT * ROWB, NSLOT = 5;   // 32 KB per unit
    __shared__ __attribute__((aligned(16))) unsigned char lds[NSLOT * UNIT];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, grp = wm;
    const int lm = lane & 15, lq = lane >> 4;
    const int nbn = (g.N + BT - 1) / BT, nbm = (g.M + BT - 1) / BT, ntiles = nbn * nbm;
    const int w = blockIdx.x, nwg = gridDim.x, xcd = w % 8, j0 = w / 8;
    const int per_xcd = (nwg - xcd + 7) / 8;
    const int tq = ntiles / 8, tr = ntiles % 8;
    const int t_start = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq, t_count = tq + (xcd < tr ? 1 : 0);
    const int n_my = j0 < t_count ? (t_count - j0 + per_xcd - 1) / per_xcd : 0;
    const int nkt = g.K / BK, total = n_my * nkt;
    if (total == 0) return;
    const T *A = reinterpret_cast<const T *>(g.A);
    const T *W = reinterpret_cast<const T *>(g.W);
    typedef __attribute__((address_space(3))) void *lds_ptr;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr)lds;

    // timing ablations (-DACAI_GEMM_ABLATE builds only - tools/ablate_pp.py, tools/build_variant.sh; ACAI_GEMM_DEBUG bits, results are wrong):
    // 1 no counted waits, 2 no LDS-DMA, 4 no epilogue, 8 no MFMAs, 16 no fragment reads, 32 start-up skew of the workgroups.  The production
    // build compiles them out (a stray high flag bit or a leftover environment variable silently dropped stores).
#ifdef ACAI_GEMM_ABLATE
    const int dbg = g.flags >> 8;
#else
    constexpr int dbg = 0;
#endif
    // ---- producer (as pers256): one cursor per operand, A runs one K-step ahead of W ----
    // Sources as a uniform base (SGPR pair: the tile's first row, advanced per K-step) + a 32-bit byte offset per lane: (row within the tile,
    // clamped to the operand's last row) x pitch + the swizzled 16-byte slot.  24-bit multiplies (the host checks pitch < 2^24 bytes).
    const int grow = lane >> 3;
    const int r0 = wave * 32 + grow;                                  // this lane's tile row for piece i: r0 + 8 i
    const uint32_t g0 = (uint32_t)(((lane & 7) ^ (grow >> 1)) << 4);   // its slot for even i; odd i: ^ 64
    const uint32_t pitchA = (uint32_t)g.lda * 2, pitchW = (uint32_t)g.ldw * 2;
    uint32_t offA[4], offW[4];
    const T *tileA = A, *tileW = W;
    int a_tile = 0, a_kt = 0, w_tile = 0, w_kt = 0, p_slot = 0;
    auto tile_origin = [&](int ti, int &bm0, int &bn0) {
        const int t = t_start + j0 + ti * per_xcd;
        bm0 = (t / nbn) * BT;
        bn0 = (t % nbn) * BT;
    };
    auto set_a = [&](int ti) {
        int bm0, bn0;
        tile_origin(ti, bm0, bn0);
        tileA = A + (size_t)bm0 * g.lda;
        const int last = g.M - 1 - bm0;
#pragma unroll
        for (int i = 0; i < 4; ++i) offA[i] = __umul24(min(r0 + 8 * i, last), pitchA) + (g0 ^ ((i & 1) << 6));
    };
    auto set_w = [&](int ti) {
        int bm0, bn0;
        tile_origin(ti, bm0, bn0);
        tileW = W + (size_t)bn0 * g.ldw;
        const int last = g.N - 1 - bn0;
#pragma unroll
        for (int i = 0; i < 4; ++i) offW[i] = __umul24(min(r0 + 8 * i, last), pitchW) + (g0 ^ ((i & 1) << 6));
    };
    int a_done = 0, w_done = 0;
    auto next_slot = [&]() { p_slot = p_slot == NSLOT - 1 ? 0 : p_slot + 1; };
    auto issue_a = [&]() -> bool {
        if (a_done >= total) return false;
        const uint32_t lb = lds_base + p_slot * UNIT;
        const T *base = tileA + (size_t)a_kt * BK;   // uniform
        if (!(dbg & 2)) glds16s_x4(__builtin_amdgcn_readfirstlane(lb + wave * 4096), base, offA[0], offA[1], offA[2], offA[3]);
        if (++a_kt == nkt) {
            a_kt = 0;
            if (++a_tile < n_my) set_a(a_tile);
        }
        ++a_done;
        next_slot();
        return true;
    };
    auto issue_w = [&]() -> bool {
        if (w_done >= total) return false;
        const uint32_t lb = lds_base + p_slot * UNIT;
        const T *base = tileW + (size_t)w_kt * BK;
        if (!(dbg & 2)) glds16s_x4(__builtin_amdgcn_readfirstlane(lb + wave * 4096), base, offW[0], offW[1], offW[2], offW[3]);
        if (++w_kt == nkt) {
            w_kt = 0;
            if (++w_tile < n_my) set_w(w_tile);
        }
        ++w_done;
        next_slot();
        return true;
    };

    // ---- consumer: swapped 16x16x32 MFMAs; lane (lm, lq) of accumulator [mb][nb] holds C[mb*16 + lm][nb*16 + 4*lq + 0..3] ----
    f32x4 acc[8][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // logical 16-byte chunk c = 4 kh + lq of tile row r sits in slot c ^ ((r >> 1) & 7); (r >> 1) & 7 = (lm >> 1) & 7 for every row this lane
    // reads (the rows differ by multiples of 16).  16 lanes of a ds_read_b128 group then cover 16 distinct 16-byte bank groups.
    const int swz = (lm >> 1) & 7;
    const int offk0 = (lq ^ swz) << 4, offk1 = ((4 + lq) ^ swz) << 4;
    const int a_off = (wm * 128 + lm) * ROWB, w_off = (wn * 64 + lm) * ROWB;
    uint4 fa[8], fw[4];
    auto reads = [&](const unsigned char *sa, const unsigned char *sb, int offk) {
#pragma unroll
        for (int j = 0; j < 4; ++j) fw[j] = *reinterpret_cast<const uint4 *>(sb + w_off + j * 16 * ROWB + offk);
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const uint4 *>(sa + a_off + i * 16 * ROWB + offk);
    };
    auto mfmas = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fw[j]), __builtin_bit_cast(bf16x8, fa[i]), acc[i][j], 0, 0, 0);
    };
#define PP_BAR()                                      \
    do {                                              \
        __builtin_amdgcn_sched_barrier(0);            \
        asm volatile("s_barrier" ::: "memory");       \
        __builtin_amdgcn_sched_barrier(0);            \
    } while (0)
#define PP_LGKM0()                                               \
    do {                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
        __builtin_amdgcn_sched_barrier(0);                       \
    } while (0)
    // counted wait of K-step q: everything up to W(q+1) has landed; younger and allowed in flight: the epilogue's stores (n_st of them when an
    // epilogue ran in this K-step) and the four pieces of A(q+2) (when it exists)
    auto wait_step = [&](int allow) {
        switch (allow) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
            case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
            case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
            case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
            case 32: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
            case 36: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
    };
    constexpr int n_st = (MODE & PP_OBF) ? ((MODE & (PP_AUX1 | PP_AUX3)) ? 32 : 16) : 32;   // store instructions per wave and tile (host: vec_epi only)

    // The bias of the tile in flight, one column per lane (lane L: bias[bn0 + wn*64 + L]); the epilogue distributes it with ds_bpermute and
    // issues no vector-memory load of its own.  A compiler-visible load anywhere in this loop is waited for with vmcnt(0) at its first use,
    // i.e. behind whatever LDS-DMA is in flight then (vmcnt counts in order): 2.5-3.6 us per tile, measured at three placements.  So the load is
    // an asm statement the wait-count pass does not see, issued in front of the K-step's LDS-DMA; the counted wait at the end of the same K-step
    // retires it together with everything older than W(q+1), and it is read at the top of the NEXT K-step.
    // Rule for every such untracked load in this kernel (bias, deferred chunks): ONE load statement and ONE take statement per loop iteration,
    // both unconditional and in straight-line code, the value handed from the load's "=v" output to the take's "+v" operand.  A first version
    // loaded under `if (tile start)` into a register tied through the asm statements: the branches' merge points made the register allocator
    // copy the value right behind the load statement - i.e. read it before the data had landed.  (Accumulator registers as destinations would
    // be out of the compiler's reach, but a kernel that touches AGPRs gets its 256 registers split 128 / 128 on gfx950: 100+ spills here.)
    uint32_t bias_raw = 0;
    int c_tile = 0, c_kt = 0, slot_a = 0;
    auto load_bias = [&]() {      // this K-step's tile
        int bm0, bn0;
        tile_origin(c_tile < n_my ? c_tile : n_my - 1, bm0, bn0);
        const uint32_t off = (uint32_t)min(bn0 + wn * 64 + lane, g.N - 1) * 4;
        const float *bp = g.bias ? g.bias : reinterpret_cast<const float *>(g.W);   // (no bias: any valid address; the value is not used)
        asm volatile("global_load_dword %0, %1, %2" : "=v"(bias_raw) : "v"(g.bias ? off : 0u), "s"(bp) : "memory");
    };
    // ---- PP_DEFER: the expensive half of a GELU epilogue leaves the tile boundary -----------------------------------------------------------
    // The GELU forms (forward: C = gelu(a), aux = a; backward: C = v gelu'(aux)) spend ~10k cycles of VALU per wave and tile at the boundary,
    // beside ONE 512-cycle MFMA segment of the partner wave: 2 x 10k exposed per tile against 16k of MFMA time (K = 512).  Deferred form: the
    // boundary epilogue only stores the bf16-rounded linear output (a into aux / v into C - both are rounded to bf16 before the GELU step in
    // the undeferred arithmetic too, so nothing changes numerically), and the GELU step runs in CHUNKS of 8 columns x 16 rows per lane - one
    // 16-byte piece of the SAME lane's own stores, read back from L2 - one chunk per R segment of the next tile, i.e. beside the partner's MFMAs
    // for the whole tile.  A chunk's piece is loaded one segment ahead by an untracked asm load (as the bias is: a compiler-visible load would be
    // waited for behind the LDS-DMA in flight), in FRONT of that segment's LDS-DMA, so waiting for it (vmcnt(4): only the DMA pieces are
    // younger) never waits for a DMA.
    // MEASURED (round 3, tools/bench_pp.py, same box): slower - MAE decoder lin1 + GELU 833 against 686 us, its gelu' GEMM 962 against 825 us,
    // the encoder's 283 / 304 against 236 / 254 us.  A chunk is ~1100 cycles of VALU in an R segment whose partner's M segment is 512: every
    // barrier interval stretches to the chunk, 32 intervals per tile, where the boundary form pays its 2 x 10k once - the two waves of a SIMD
    // share one VALU port, so 20k VALU cycles per SIMD and tile cannot hide behind 16k of MFMA however they are cut - and the chunks come back
    // from L2 / MALL (0.8 GB more reads on lin1).  Off by default (variant 8 / ACAI_GEMM_PP_DEFER=1 select it); the test keeps it runnable.
    constexpr bool DEFER = MODE & PP_DEFER;
    constexpr int NL = (MODE & PP_AUX2) ? 2 : 1;     // loads per chunk
    // two chunk slots: X is loaded in R(q,0) and finished in R(q,1), Y is loaded in R(q,1) and finished in R(q+1,0); *2: the saved
    // pre-activation of the backward form.  Loads and takes are unconditional (an idle slot reloads chunk 0 and drops its result), see the rule
    // at the bias load.
    u32x4_t cX = {0, 0, 0, 0}, cX2 = {0, 0, 0, 0}, cY = {0, 0, 0, 0}, cY2 = {0, 0, 0, 0};
    int d_m0 = 0, d_n0 = 0, d_next = 16;   // tile being drained (its wave block's origin) and its next chunk to load (16: nothing to drain)
    int idX = -1, idY = -1;                // chunk id in each slot (-1: idle)
    // one chunk per R segment: a tile has 2 nkt of them for its predecessor's 16 chunks (host: nkt >= 8, i.e. K >= 512 - every MLP of the path)
    const __amdgpu_buffer_rsrc_t d_rc = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, (int)((size_t)g.M * g.ldc * 2), 0x00020000);
    auto chunk_off = [&](int c, int ld, bool &ok) -> uint32_t {
        const int row = d_m0 + (c & 7) * 16 + lm, col = d_n0 + (lq & 1) * 16 + (lq >> 1) * 8 + (c >> 3) * 32;
        ok = c >= 0 && row < g.M && col < g.N;
        return ok ? (uint32_t)row * (uint32_t)(ld * 2) + (uint32_t)col * 2 : 0u;
    };
    auto chunk_load = [&](int c, u32x4_t &L, u32x4_t &A2) {
        bool ok;
        if constexpr (MODE & PP_AUX2) {
            const uint32_t oc = chunk_off(c, g.ldc, ok), ox = chunk_off(c, g.ldaux, ok);
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(L) : "v"(oc), "s"(g.C) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(A2) : "v"(ox), "s"(g.aux) : "memory");
        } else {
            const uint32_t ox = chunk_off(c, g.ldaux, ok);
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(L) : "v"(ox), "s"(g.aux) : "memory");
        }
    };
    // wait for the slot's loads: only the LDS-DMA pieces issued behind them (four, or none at the end of the work list) may stay outstanding
    auto chunk_take = [&](bool dma_behind, u32x4_t &L, u32x4_t &A2) {
        // (the statement that carries the registers is unconditional: with one take per branch the merge point copied the in-flight registers)
        if (!dma_behind) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (MODE & PP_AUX2) asm volatile("s_waitcnt vmcnt(4)" : "+v"(L), "+v"(A2)::"memory");
        else asm volatile("s_waitcnt vmcnt(4)" : "+v"(L)::"memory");
    };
    auto chunk_finish = [&](int c, const u32x4_t &L, const u32x4_t &A2) {
        bool ok;
        const uint32_t oc = chunk_off(c, g.ldc, ok);
        u32x4_t o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float lo = __uint_as_float(L[i] << 16), hi = __uint_as_float(L[i] & 0xFFFF0000u);
            if constexpr (MODE & PP_AUX2) {
                lo *= gelu_erf_grad(__uint_as_float(A2[i] << 16));
                hi *= gelu_erf_grad(__uint_as_float(A2[i] & 0xFFFF0000u));
            } else {
                lo = gelu_erf(lo);
                hi = gelu_erf(hi);
            }
            o[i] = pack_bf16(lo, hi);
        }
        __builtin_amdgcn_raw_buffer_store_b128(o, d_rc, ok ? oc : 0xFFFFFFF0u, 0, 0);
    };

    set_a(0);
    set_w(0);
    issue_a();   // A(0)
    issue_w();   // W(0)
    issue_a();   // A(1)
    zero_acc();
    if (total > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PP_BAR();              // b_{-1}: K-step 0 has landed for everybody
    if (dbg & 32) {
        for (int i = 0; i < (int)(blockIdx.x / 8 % 8); ++i) __builtin_amdgcn_s_sleep(32);   // ~ (w/8 % 8) x 1 us
    }
    if (grp == 1) PP_BAR();   // G1 runs one segment behind
    bool dma_prev = true;    // the segment before the first one of the loop issued A(1)
    for (int q = 0; q <= total; ++q) {
        const bool live = q < total;    // the extra pass q == total only runs the last tile's epilogue (and keeps the barrier count)
        const int slot_w = slot_a == NSLOT - 1 ? 0 : slot_a + 1;
        const unsigned char *sa = lds + slot_a * UNIT, *sb = lds + slot_w * UNIT;
        // ---- R(q, 0) ----
        const bool boundary = c_kt == 0 && q > 0;
        if (!live) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (no counted wait came after the last K-step's bias / chunk loads)
        asm volatile("" : "+v"(bias_raw));   // take: the load of the K-step before was retired by that K-step's counted wait
        const float bv = g.bias ? __uint_as_float(bias_raw) : 0.f;
        int allow = q + 2 < total ? 4 : 0;
        if constexpr (DEFER) {
            // slot Y (loaded in R(q-1,1), in front of A(q+1)'s pieces): take and finish - at a tile boundary this is the old tile's last chunk
            chunk_take(live && dma_prev, cY, cY2);
            if (idY >= 0) chunk_finish(idY, cY, cY2);
            if (boundary) {
                // the plain bf16 pass of the tile just finished; its chunks start below
                int bm0, bn0;
                tile_origin(c_tile - 1, bm0, bn0);
                if (!(dbg & 4)) {
                    if constexpr (MODE & PP_AUX2) gemm_pp_epilogue<PP_OBF>(g, acc, bm0 + wm * 128, bn0 + wn * 64, lane, bv);
                    else gemm_pp_epilogue<PP_OBF>(g, acc, bm0 + wm * 128, bn0 + wn * 64, lane, bv, g.aux, g.ldaux);
                }
                zero_acc();
                d_m0 = bm0 + wm * 128;
                d_n0 = bn0 + wn * 64;
                d_next = 0;
            }
            load_bias();
            idX = (live && d_next < 16) ? d_next++ : -1;
            chunk_load(idX, cX, cX2);
            dma_prev = issue_w();   // W(q+1) -> the slot A(q-1) left
        } else {
            load_bias();
            issue_w();   // W(q+1) -> the slot A(q-1) left
            if (boundary) {
                int bm0, bn0;
                tile_origin(c_tile - 1, bm0, bn0);
                if (!(dbg & 4)) {
                    gemm_pp_epilogue<MODE>(g, acc, bm0 + wm * 128, bn0 + wn * 64, lane, bv);
                    allow += n_st;
                }
                zero_acc();
            }
        }
        if (live && !(dbg & 16)) reads(sa, sb, offk0);   // (after the epilogue: the fragment registers are not live across it)
        PP_LGKM0();
        PP_BAR();
        // ---- M(q, 0) ----
        if (live && !(dbg & 8)) mfmas();
        PP_BAR();
        // ---- R(q, 1) ----
        if constexpr (DEFER) {
            chunk_take(dma_prev, cX, cX2);   // slot X, loaded in R(q,0) in front of W(q+1)'s pieces
            if (idX >= 0) {
                chunk_finish(idX, cX, cX2);
                allow += 1;                  // its store is younger than W(q+1) ...
            }
            idY = (live && d_next < 16) ? d_next++ : -1;
            chunk_load(idY, cY, cY2);
            allow += NL;                     // ... and so are slot Y's loads
            dma_prev = issue_a();            // A(q+2) -> the slot W(q-1) left
        } else {
            issue_a();   // A(q+2) -> the slot W(q-1) left (the K-step's eight LDS-DMA pieces are split over its two R segments)
        }
        if (live && !(dbg & 16)) reads(sa, sb, offk1);
        PP_LGKM0();
        if (grp == 1 && q + 1 < total && !(dbg & 1)) wait_step(allow);
        PP_BAR();
        // ---- M(q, 1) ----
        if (live && !(dbg & 8)) mfmas();
        if (grp == 0 && q + 1 < total && !(dbg & 1)) wait_step(allow);
        PP_BAR();
        if (live && ++c_kt == nkt) {
            c_kt = 0;
            ++c_tile;
        }
        slot_a = slot_a + 2 >= NSLOT ? slot_a + 2 - NSLOT : slot_a + 2;
    }
    if (grp == 0) PP_BAR();
#undef PP_BAR
#undef PP_LGKM0
    if constexpr (DEFER) {
        // the last tile's chunks: nothing else is in flight any more, so plain (compiler-tracked) loads, all requested before the first is used
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(g.aux, 0, (int)((size_t)g.M * g.ldaux * 2), 0x00020000);
        u32x4_t Ls[16], As[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            bool ok;
            if constexpr (MODE & PP_AUX2) {
                Ls[c] = __builtin_amdgcn_raw_buffer_load_b128(d_rc, chunk_off(c, g.ldc, ok), 0, 0);
                As[c] = __builtin_amdgcn_raw_buffer_load_b128(rx, chunk_off(c, g.ldaux, ok), 0, 0);
            } else {
                Ls[c] = __builtin_amdgcn_raw_buffer_load_b128(rx, chunk_off(c, g.ldaux, ok), 0, 0);
                As[c] = Ls[c];
            }
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) chunk_finish(c, Ls[c], As[c]);
    }
}

// ---- dW = dY^T X on the three-stage ring (round 2) -------------------------------------------------------------------------------------
// gemm_tn_glds_kernel keeps ONE 64-token tile in flight per workgroup (two stages, `__syncthreads()` drains the LDS-DMA): a K-step of a
// 128x128 tile is ~0.2 us of MFMA work against >= 1 us of loaded fabric latency, and two co-resident workgroups do not cover it (0.61-0.69 PF
// on the MAE's weight gradients).  This form is the NT ring's structure on the token-major operands: 256 (dY columns) x 128 (X columns) tile,
// 8 waves as 4 x 2, THREE 48 KB stages (two [64 tokens][128 col] images of dY, one of X, each in gemm_tn_glds_kernel's swizzled layout and
// read with the same transposing fragment reads), two K-tiles in flight behind a counted `s_waitcnt vmcnt(6)` and ONE raw barrier per K-step,
// LDS-DMA issued from inline asm so the compiler's wait-count pass never sees it.  Split-K over the tokens, fp32 atomics per 32x32 block.
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void gemm_tn_ring_kernel(GemmArgs g) {
    typedef bf16_t T;
    constexpr int BKT = 64, IMG = BKT * 256, STAGE = 3 * IMG, BMT = 256;   // 16 KB per image, 48 KB per stage
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
    const int nbn = (g.N + BN - 1) / BN, nbm = (g.M + BMT - 1) / BMT, nwg = nbn * nbm;
    int pid = blockIdx.x, kslice = 0;
    if (g.ksplit > 1) {
        kslice = pid / nwg;
        pid -= kslice * nwg;
    }
    const int bm0 = (pid / nbn) * BMT, bn0 = (pid % nbn) * BN;
    int nkt = (g.K + BKT - 1) / BKT, kt_begin = 0;   // the last K-tile may be ragged: its missing token rows arrive as zeros (blds16)
    if (g.ksplit > 1) {
        const int per = (nkt + g.ksplit - 1) / g.ksplit;
        kt_begin = kslice * per;
        nkt = min(nkt, kt_begin + per);
        if (kt_begin >= nkt) return;  // uniform per workgroup, before any barrier
    }
    const int nk = nkt - kt_begin;
    const T *A = reinterpret_cast<const T *>(g.A);
    const T *W = reinterpret_cast<const T *>(g.W);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // LDS-DMA instruction j (0..15) of an image covers token rows 4j .. 4j+3; lane l lands at row 4j + l/16, 16-byte unit l%16, which holds logical
    // unit (l%16) ^ ((row & 3) << 2).  Wave w issues j = 2w, 2w+1 of each of the three images: six instructions per wave and K-tile.
    // Columns beyond M / N are clamped to the last whole chunk (their outputs are never stored).
    uint32_t src[6];   // byte offsets from the K-tile's first token row (the host checks that they fit 32 bits)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 4 + (lane >> 4);
        const int u = (lane & 15) ^ ((r & 3) << 2);
        src[i] = (uint32_t)r * (uint32_t)(g.lda * 2) + (uint32_t)min(bm0 + u * 8, g.M - 8) * 2;
        src[2 + i] = (uint32_t)r * (uint32_t)(g.lda * 2) + (uint32_t)min(bm0 + 128 + u * 8, g.M - 8) * 2;
        src[4 + i] = (uint32_t)r * (uint32_t)(g.ldw * 2) + (uint32_t)min(bn0 + u * 8, g.N - 8) * 2;
    }
    typedef __attribute__((address_space(3))) void *lds_ptr;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr)lds;
    auto issue = [&](int kt, int slot) {
        const uint32_t lb = lds_base + slot * STAGE;
        const int rows = min(g.K - kt * BKT, BKT);   // token rows of this K-tile that exist
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(A + (size_t)kt * BKT * g.lda), 0, rows * g.lda * 2, 0x00020000);
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(W + (size_t)kt * BKT * g.ldw), 0, rows * g.ldw * 2, 0x00020000);
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint32_t dst = __builtin_amdgcn_readfirstlane(lb + m * IMG + (wave * 2 + i) * 1024);
                blds16(dst, m == 2 ? rw : ra, src[2 * m + i]);
            }
    };
    typedef __attribute__((ext_vector_type(4))) short s4;
    typedef __attribute__((address_space(3))) s4 *lds_s4;
    const int i16 = lane & 15, g1 = (lane >> 4) & 1;
    // fragment of output rows rowbase .. rowbase+31 of one 128-column image, contraction slice s: element j of lane half h is token 16 s + 8 h + j
    auto frag = [&](const unsigned char *img, int rowbase, int s) -> uint4 {
        const int m = 16 * s + 8 * lh + (i16 >> 2), bc = (rowbase + 16 * g1 + 4 * (i16 & 3)) * 2;
        const int o = m * 256 + ((((bc >> 6) ^ (m & 3))) << 6) + (bc & 63);   // rows m and m + 4 share (m & 3)
        union { s4 v[2]; uint4 u; } r;
        r.v[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(img + o));
        r.v[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(img + o + 4 * 256));
        return r.u;
    };
    const int a_img = (wm >> 1) * IMG, a_row = (wm & 1) * 64;   // this wave's 64 dY columns: image wm / 2, rows (wm % 2) * 64 ..
    auto compute = [&](const unsigned char *st) {
        const unsigned char *sa = st + a_img, *sb = st + 2 * IMG;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            uint4 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = frag(sa, a_row + i * 32, s);
                fb[i] = frag(sb, wn * 64 + i * 32, s);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
        }
    };
    issue(kt_begin, 0);
    if (nk > 1) issue(kt_begin + 1, 1);
    int slot = 0;
    for (int q = 0; q < nk; ++q) {
        if (q + 1 < nk)
            asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");   // this wave's six pieces of tile q have landed (tile q + 1 may still fly); everybody's have
        else
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (q + 2 < nk) issue(kt_begin + q + 2, slot == 0 ? 2 : slot - 1);   // the stage read in step q - 1: every wave is past it
        compute(lds + slot * STAGE);
        slot = slot == 2 ? 0 : slot + 1;
    }
    gemm_accum_epilogue(g, acc, bm0, bn0, wm, wn, lr, lh);
}

// ---- dW = dY^T X with ping-pong wave groups on the ring of half-stages (round 3) ---------------------------------------------------------
// gemm_tn_ring_kernel (256 x 128 tile, three 48 KB stages, both waves of a SIMD in lock-step behind one barrier per K-step) ran the MAE's
// weight gradients at 0.71-0.89 PF.  This is gemm_nt_pp_kernel's structure on the token-major operands: 256 (dY columns) x 256 (X columns)
// tile, 8 waves as 2 x 4 (128 x 64 each, 16x16x32 MFMAs: 128 accumulator registers), K-step = 64 tokens, a unit = one operand's
// [64 tokens][256 columns] image (32 KB, as two [64][128-column] images of 256-byte rows), five unit slots, the dY unit two K-steps ahead and
// the X unit one, LDS-DMA from inline asm behind counted waits, and the two wave groups one segment apart (R: fragment reads + this
// segment's four LDS-DMA pieces, M: 32 MFMAs; two of each per K-step).  Fragments - eight consecutive tokens of one column - come from
// ds_read_b64_tr_b16 out of the natural image, whose 16-byte chunk c of token row r sits in slot c ^ (((r & 3) << 2) | ((r >> 2) & 3)): the
// four token rows of a transposing read land in four different 64-byte blocks and the two 16-lane groups of a half-wave (token rows 8 apart)
// in different 32-byte halves (cdna_hip_programming.md T10, image (b)).  One (tile, K-slice) per workgroup, split-K over the tokens so that
// about one workgroup per CU exists, fp32 atomics straight from the accumulators: the un-swapped 16x16 layout gives a lane one column and
// four consecutive rows, so a wave instruction adds four 64-byte row segments.
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void gemm_tn_pp_kernel(GemmArgs g) {
    typedef bf16_t T;
    constexpr int BKT = 64, BT = 256, UNIT = BKT * BT * 2, NSLOT = 5, IMG = BKT * 256;   // 32 KB per unit, 16 KB per 128-column image
    __shared__ __attribute__((aligned(16))) unsigned char lds[NSLOT * UNIT];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, grp = wm;
    const int nbn = (g.N + BT - 1) / BT, nbm = (g.M + BT - 1) / BT, nwg = nbn * nbm;
    int pid = blockIdx.x, kslice = 0;
    if (g.ksplit > 1) {
        kslice = pid / nwg;
        pid -= kslice * nwg;
    }
    const int bm0 = (pid / nbn) * BT, bn0 = (pid % nbn) * BT;
    int nkt = (g.K + BKT - 1) / BKT, kt_begin = 0;   // the last K-step may be ragged: its missing token rows arrive as zeros (blds16_x4)
    if (g.ksplit > 1) {
        const int per = (nkt + g.ksplit - 1) / g.ksplit;
        kt_begin = kslice * per;
        nkt = min(nkt, kt_begin + per);
        if (kt_begin >= nkt) return;  // uniform per workgroup, before any barrier
    }
    const int total = nkt - kt_begin;
    const T *A = reinterpret_cast<const T *>(g.A) + (size_t)kt_begin * BKT * g.lda;
    const T *W = reinterpret_cast<const T *>(g.W) + (size_t)kt_begin * BKT * g.ldw;
    const int rows_left0 = g.K - kt_begin * BKT;   // token rows from this slice's first K-step to the end of the operands
    typedef __attribute__((address_space(3))) void *lds_ptr;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr)lds;

    // ---- producer: piece j (0..31) of a unit = image j / 16, token rows 4 (j % 16) .. + 3; wave w issues pieces 4w .. 4w+3 (4 KB of LDS in a row).
    // Lane l of a piece: token row 4 (j % 16) + l / 16, LDS slot l % 16 <- logical chunk (l % 16) ^ swz(row).  Columns beyond the operand
    // are clamped to its last whole chunk (their outputs are never stored).
    uint32_t offA[4], offW[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = wave * 4 + i, row = 4 * (j & 15) + (lane >> 4);
        const int ch = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
        const int col = 128 * (j >> 4) + 8 * ch;
        offA[i] = (uint32_t)row * (uint32_t)(g.lda * 2) + (uint32_t)min(bm0 + col, g.M - 8) * 2;
        offW[i] = (uint32_t)row * (uint32_t)(g.ldw * 2) + (uint32_t)min(bn0 + col, g.N - 8) * 2;
    }
    int a_done = 0, w_done = 0, p_slot = 0;
    auto next_slot = [&]() { p_slot = p_slot == NSLOT - 1 ? 0 : p_slot + 1; };
    auto issue_a = [&]() {
        if (a_done >= total) return;
        blds16_x4(__builtin_amdgcn_readfirstlane(lds_base + p_slot * UNIT + wave * 4096),
                  __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(A + (size_t)a_done * BKT * g.lda), 0, min(rows_left0 - a_done * BKT, BKT) * g.lda * 2, 0x00020000),
                  offA[0], offA[1], offA[2], offA[3]);
        ++a_done;
        next_slot();
    };
    auto issue_w = [&]() {
        if (w_done >= total) return;
        blds16_x4(__builtin_amdgcn_readfirstlane(lds_base + p_slot * UNIT + wave * 4096),
                  __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(W + (size_t)w_done * BKT * g.ldw), 0, min(rows_left0 - w_done * BKT, BKT) * g.ldw * 2, 0x00020000),
                  offW[0], offW[1], offW[2], offW[3]);
        ++w_done;
        next_slot();
    };

    // ---- consumer: D[m][n] += sum_tokens dY[token][m] X[token][n]; lane (col = lane & 15, lq = lane >> 4) of accumulator [mb][nb] holds
    // C[mb*16 + 4*lq + e][nb*16 + col], e = 0..3 ----
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // transposing fragment read of the 16-column block at image column c0, contraction slice kh (32 tokens): lane 4q'+p' of a 16-lane group
    // addresses token row t0 + q', columns c0 + 4p' .. + 3, t0 = 32 kh + 8 (lane >> 4) (+ 4 for the second read); lane i receives column i.
    // Everything lane-dependent is folded into two byte offsets per kh (first / second read); the block's column enters as XOR (c0 / 8) << 4.
    typedef __attribute__((ext_vector_type(4))) short s4;
    typedef __attribute__((address_space(3))) s4 *lds_s4;
    const int l16 = lane & 15, qp = l16 >> 2, pp = l16 & 3, lq = lane >> 4;
    int fro[2][2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int h4 = 0; h4 < 2; ++h4) {
            const int row = 32 * kh + 8 * lq + 4 * h4 + qp;
            fro[kh][h4] = row * 256 + ((((pp >> 1)) ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4) + 8 * (pp & 1);
        }
    uint4 fa[8], fw[4];
    auto frag = [&](const unsigned char *img, int c0, int kh) -> uint4 {
        union { s4 v[2]; uint4 u; } r;
        r.v[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(img + (fro[kh][0] ^ ((c0 >> 3) << 4))));
        r.v[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(img + (fro[kh][1] ^ ((c0 >> 3) << 4))));
        return r.u;
    };
    auto reads = [&](const unsigned char *sa, const unsigned char *sb, int kh) {
        // this wave's 128 dY columns are image wm of the dY unit; its 64 X columns are columns (wn & 1) * 64 .. of image wn >> 1 of the X unit
#pragma unroll
        for (int j = 0; j < 4; ++j) fw[j] = frag(sb + (wn >> 1) * IMG, (wn & 1) * 64 + j * 16, kh);
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] = frag(sa + wm * IMG, i * 16, kh);
    };
    // Bias gradient beside the weight gradient (g.colsum): the column sums of dY are one more product of the dY fragments already in registers,
    // against a B operand of ones.  The 16-column fragments of a 128-column group are dealt over the tile columns (workgroups bn) and the four
    // waves that share them: fragment i belongs to the workgroup with bn % nbn == i % nbn and its wave wn == (i / nbn) % 4, slot i / (4 nbn) -
    // at most two extra MFMAs per 32 and wave (one from nbn = 2 on), instead of a separate pass over dY (colsum_vec_kernel: 1.9 ms per MAE step).
    const bool do_cs = g.colsum != nullptr;
    const int cs_b = pid % nbn;
    f32x4 csacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const uint4 ones8 = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);
    auto cs_mine = [&](int i) -> bool { return (i % nbn) == cs_b && ((i / nbn) & 3) == wn && i / (4 * nbn) < 2; };
    auto mfmas = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fw[j]), acc[i][j], 0, 0, 0);
        if (do_cs) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (cs_mine(i)) {   // (wave-uniform)
                    if (i / (4 * nbn) == 0) csacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, ones8), csacc[0], 0, 0, 0);
                    else csacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, ones8), csacc[1], 0, 0, 0);
                }
        }
    };
#define PP_BAR()                                      \
    do {                                              \
        __builtin_amdgcn_sched_barrier(0);            \
        asm volatile("s_barrier" ::: "memory");       \
        __builtin_amdgcn_sched_barrier(0);            \
    } while (0)
#define PP_LGKM0()                                               \
    do {                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
        __builtin_amdgcn_sched_barrier(0);                       \
    } while (0)
    issue_a();   // A(0)
    issue_w();   // W(0)
    issue_a();   // A(1)
    if (total > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PP_BAR();
    if (grp == 1) PP_BAR();   // G1 runs one segment behind
    int slot_a = 0;
    for (int q = 0; q < total; ++q) {
        const int slot_w = slot_a == NSLOT - 1 ? 0 : slot_a + 1;
        const unsigned char *sa = lds + slot_a * UNIT, *sb = lds + slot_w * UNIT;
        // ---- R(q, 0) ----
        issue_w();   // X(q+1) -> the slot dY(q-1) left
        reads(sa, sb, 0);
        PP_LGKM0();
        PP_BAR();
        // ---- M(q, 0) ----
        mfmas();
        PP_BAR();
        // ---- R(q, 1) ----
        issue_a();   // dY(q+2) -> the slot X(q-1) left
        reads(sa, sb, 1);
        PP_LGKM0();
        if (grp == 1 && q + 1 < total) {
            if (q + 2 < total) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        PP_BAR();
        // ---- M(q, 1) ----
        mfmas();
        if (grp == 0 && q + 1 < total) {
            if (q + 2 < total) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        PP_BAR();
        slot_a = slot_a + 2 >= NSLOT ? slot_a + 2 - NSLOT : slot_a + 2;
    }
    if (grp == 0) PP_BAR();
#undef PP_BAR
#undef PP_LGKM0
    if (do_cs && (lane & 15) == 0) {   // every column n of the 16x16 result holds the same sums: lanes with n = 0 add rows 4 lq .. 4 lq + 3
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (cs_mine(i)) {
                const f32x4 v = i / (4 * nbn) == 0 ? csacc[0] : csacc[1];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int m = bm0 + wm * 128 + i * 16 + 4 * lq + e;
                    if (m < g.M) atomicAdd(g.colsum + m, v[e]);
                }
            }
    }
    // ---- split-K accumulation: four 64-byte row segments per wave instruction ----
    const int m0 = bm0 + wm * 128 + 4 * lq, n0 = bn0 + wn * 64 + (lane & 15);
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const int col = n0 + nb * 16;
            if (col < g.N) {
                float *base = reinterpret_cast<float *>(g.C) + (size_t)(m0 + mb * 16) * g.ldc + col;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (m0 + mb * 16 + e < g.M) atomicAdd(base + (size_t)e * g.ldc, acc[mb][nb][e]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// epilogue form of a launch as gemm_nt_pp_kernel's MODE, and the forms it is instantiated for (the others keep variant 6)
static inline int pp_mode(const GemmArgs &g) {
    return (g.out_dtype == ACAI_BF16 ? PP_OBF : 0) | ((g.flags & ACAI_GEMM_GELU) ? PP_GELU : 0) | (g.aux_mode == 1 ? PP_AUX1 : 0) | (g.aux_mode == 2 ? PP_AUX2 : 0) |
           (g.aux_mode == 3 ? PP_AUX3 : 0) | (g.aux_mode == 4 ? PP_AUX4 : 0) |
           (g.residual ? PP_RES : 0) | (g.scale_cols > 0 ? PP_SCALE : 0);
}
static inline bool pp_mode_ok(int m) {
    return m == PP_OBF || m == (PP_OBF | PP_SCALE) || m == (PP_OBF | PP_GELU | PP_AUX1) || m == (PP_OBF | PP_AUX2) || m == 0 || m == PP_RES ||
           m == (PP_OBF | PP_GELU | PP_AUX3) || m == (PP_OBF | PP_AUX4);
}

int g_gemm_variant = getenv("ACAI_GEMM_VARIANT") ? atoi(getenv("ACAI_GEMM_VARIANT")) : 0;

template <typename T, int EPI, bool TA = false, bool TB = false>
int launch(const GemmArgs &g, hipStream_t st) {
    constexpr int EPC = 16 / sizeof(T);
    // 16-byte chunks run along K for row-major operands and along the row index for transposed ones
    const bool fast = (g.lda % EPC == 0) && (g.ldw % EPC == 0) && aligned16(g.A) && aligned16(g.W) && (TA ? g.M % EPC == 0 : g.K % EPC == 0) &&
                      (TB ? g.N % EPC == 0 : g.K % EPC == 0);
    int nwg = cdiv(g.M, BM) * cdiv(g.N, BN);
    GemmArgs h = g;
    if (TA && TB && EPI == 0) {
        // dW = dY^T X: few output tiles, K = number of rows (1e4..1e5): split K until ~3 workgroups per CU exist
        constexpr int BKE = ROWB / (int)sizeof(T);
        const int nkt = cdiv(g.K, BKE);
        int ks = cdiv(768, nwg);
        ks = ks < 1 ? 1 : (ks > nkt / 4 ? (nkt / 4 < 1 ? 1 : nkt / 4) : ks);
        h.ksplit = ks;
        nwg *= ks;
    }
    h.vec_epi = (EPI == 0) && (g.ldc % 8 == 0) && (g.N % 8 == 0) && aligned16(g.C) && (!g.residual || (g.ldr % 4 == 0 && aligned16(g.residual))) &&
                (!g.aux_mode || (g.ldaux % 8 == 0 && aligned16(g.aux)));
    constexpr int BKG = ROWB / (int)sizeof(T);
    static const bool no_glds = getenv("ACAI_GEMM_NO_GLDS") != nullptr;
    static const bool no_tn = getenv("ACAI_GEMM_NO_TN_GLDS") != nullptr;   // A/B aid
    if constexpr (TA && TB && EPI == 0 && sizeof(T) == 2) {
        if (fast && !no_tn && !no_glds && h.ksplit > 0 && g.K >= 64 && g.M % 8 == 0 && g.N % 8 == 0 && g.M >= 8 && g.N >= 8) {
            // token counts that are not a multiple of 64 (16 x 513 decoder tokens): the LDS-DMA kernel takes the whole 64-token tiles, the
            // register-staged kernel accumulates the remaining rows into the same gradient
            // (round 3: the ring and ping-pong kernels take the ragged last tile themselves - out-of-range rows of a buffer LDS-DMA arrive as
            // zeros; only the two-stage kernel still leaves the remainder to the register-staged one)
            const int k64 = g.K - g.K % 64;
            GemmArgs m = h;
            m.K = k64;
            bool whole = false;   // the launched kernel covered every token row
            // the three-stage ring (256 x 128 tiles, one workgroup per CU) when the reduction is long enough to fill it: split K until ~256 workgroups exist
            static const bool no_ring = getenv("ACAI_GEMM_TN_RING") && atoi(getenv("ACAI_GEMM_TN_RING")) == 0;   // A/B aid
            const int tiles_r = cdiv(g.M, 256) * cdiv(g.N, BN), nkt_r = cdiv(g.K, 64);
            int ks_r = 256 / tiles_r;            // one resident workgroup per CU: never more workgroups than CUs (a second, nearly empty round doubles the time)
            if (ks_r < 1) ks_r = 1;
            if (ks_r > nkt_r / 8) ks_r = nkt_r / 8;
            // round 3: the ping-pong form (256 x 256 tiles) where the operands' lane offsets fit 32 bits
            static const int tn_pp = getenv("ACAI_GEMM_TN_PP") ? atoi(getenv("ACAI_GEMM_TN_PP")) : 1;   // A/B aid
            const int tiles_p = cdiv(g.M, 256) * cdiv(g.N, 256);
            int ks_p = 256 / tiles_p;
            if (ks_p < 1) ks_p = 1;
            if (ks_p > nkt_r / 8) ks_p = nkt_r / 8;
            const bool pp_fits = (size_t)64 * g.lda * 2 + (size_t)g.M * 2 < 0xFFFFFF00ull && (size_t)64 * g.ldw * 2 + (size_t)g.N * 2 < 0xFFFFFF00ull;
            // (short K-slices keep the 256 x 128 ring: with fewer than ~24 K-steps per workgroup the atomics of the larger tile's extra
            // splits cost more than the main loop gains - encoder dWo, 768 x 768 from 32768 tokens: 82 us on the ring, 89 us here)
            float *const want_cs = g.colsum;   // only the ping-pong kernel forms the column sums itself
            m.colsum = nullptr;
            if (tn_pp && !no_ring && ks_p >= 1 && nkt_r >= 16 && pp_fits && nkt_r / ks_p >= 24) {
                m.ksplit = ks_p;
                m.K = g.K;
                m.colsum = want_cs;
                if (g.colsum_done) *g.colsum_done = want_cs != nullptr;   // (a per-call out flag: a process global raced between devices' backward threads)
                whole = true;
                hipLaunchKernelGGL(gemm_tn_pp_kernel, dim3(tiles_p * ks_p), dim3(512), 0, st, m);
            } else if (!no_ring && ks_r >= 1 && nkt_r >= 16 && pp_fits) {
                m.ksplit = ks_r;
                m.K = g.K;
                whole = true;
                hipLaunchKernelGGL(gemm_tn_ring_kernel, dim3(tiles_r * ks_r), dim3(512), 0, st, m);
            } else
            hipLaunchKernelGGL(gemm_tn_glds_kernel, dim3(nwg), dim3(256), 0, st, m);
            if (k64 < g.K && !whole) {
                GemmArgs t = h;
                t.A = reinterpret_cast<const T *>(g.A) + (size_t)k64 * g.lda;
                t.W = reinterpret_cast<const T *>(g.W) + (size_t)k64 * g.ldw;
                t.K = g.K - k64;
                t.ksplit = 1;
                hipLaunchKernelGGL((gemm_nt_kernel<T, 0, true, true, true>), dim3(cdiv(g.M, BM) * cdiv(g.N, BN)), dim3(256), 0, st, t);
            }
            ACAI_LAUNCH_CHECK("acai_gemm");
            return 0;
        }
    }
    if (fast && !TA && !TB && g.K % BKG == 0 && !no_glds) {
        // Row-major LDS-DMA kernels.  variant (acai_gemm_set_variant / ACAI_GEMM_VARIANT; tests and A/B runs): 0 auto, 1 128x128 two-stage,
        // 2 256x128 two-stage, 3 256x128 three-stage, 4 256x128 persistent three-stage ring, 5 256x256 two-stage, 6 persistent 256x256 ring of
        // half-stages, 7 ping-pong ring with the register epilogue, 8 = 7 with the GELU forms' deferred epilogue.
        static const int n_cu = [] {
            int dev = 0, n = 256;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
            return n > 0 ? n : 256;
        }();
        // A few rows past a multiple of 256 (the teacher-forced decoder stream: 16 x 513 = 8208 = 32 x 256 + 16 rows) can cost a whole round of tiles:
        // N = 4096 is 528 tiles of 256 x 256 on 256 CUs (three rounds for 2.06 of work: 86 us against 65 at M = 8192), N = 1024 is 520 tiles of
        // 128 x 128 on 512 resident workgroups (two rounds for 1.02: 33.5 us against 25.7; K = 4096: 99 against 74).  Where the full 256-row part
        // alone saves such a round, it is launched by itself (the auto rule then picks what it picks for a round number) and the remainder rows
        // follow as a second, small launch of the same epilogue form (tools/bench_gemm_tf.py).  ACAI_GEMM_SPLIT_REM=0: A/B aid.
        if constexpr (sizeof(T) == 2 && EPI == 0) {
            static const bool split_rem = !(getenv("ACAI_GEMM_SPLIT_REM") && atoi(getenv("ACAI_GEMM_SPLIT_REM")) == 0);
            const int rem = g.M % 256, Mm = g.M - rem;
            if (split_rem && g_gemm_variant == 0 && rem > 0 && rem <= 32 && Mm >= 4096) {
                const int nbn256 = cdiv(g.N, 256), full256 = cdiv(g.M, 256) * nbn256, main256 = (Mm / 256) * nbn256;
                // (measured, tools/bench_gemm_tf.py, M = 8208: N = 4096 90.5 -> 81.5 us, N = 2048 59.7 -> 49.8.  The other case - N = 1024, where the
                // 8192-row part is exactly one round of 128 x 128 tiles - LOSES: the 16-row remainder takes 16 us at K = 1024 and 46 us at K = 4096 on
                // the eight workgroups of the generic kernel, 41.7 against 33.5 us and 121 against 101 us in all; it would need a skinny kernel with
                // the training epilogue forms, so that rule is not applied)
                const bool saves_round = main256 >= n_cu && cdiv(main256, n_cu) < cdiv(full256, n_cu);
                if (saves_round) {
                    const size_t esz = g.out_dtype == ACAI_BF16 ? 2 : 4;
                    GemmArgs m = g, t = g;
                    m.M = Mm;
                    t.M = rem;
                    t.A = reinterpret_cast<const unsigned char *>(g.A) + (size_t)Mm * g.lda * sizeof(T);
                    t.C = reinterpret_cast<unsigned char *>(g.C) + (size_t)Mm * g.ldc * esz;
                    if (g.residual) t.residual = g.residual + (size_t)Mm * g.ldr;
                    if (g.aux) t.aux = reinterpret_cast<unsigned char *>(g.aux) + (size_t)Mm * g.ldaux * esz;
                    const int rc = launch<T, EPI, TA, TB>(m, st);
                    return rc ? rc : launch<T, EPI, TA, TB>(t, st);
                }
            }
        }
        const int nwg4 = cdiv(g.M, 256) * cdiv(g.N, BN), nwg256 = cdiv(g.M, 256) * cdiv(g.N, 256);
        int v = g_gemm_variant;
        bool defer = false;   // variant 8 = 7 with the GELU forms' deferred epilogue (PP_DEFER: measured slower, kept as an experiment - see there)
        if (v == 8) {
            v = 7;
            defer = true;
        }
        // auto (tools/bench_gemm.py, bf16): up to 24 K-tiles the persistent ring wins (0.61-0.71 PF on K = 512..768 against 0.53-0.64 for
        // one tile per workgroup); from 64 K-tiles the 256x256 tile does (1.00-1.02 PF at 4096^3 / 8192^3 against 0.95-0.98); between, the
        // three-stage 256x128 kernel; small problems keep two 128x128 workgroups per CU.
        const int ktiles = g.K / BKG;
        // (fp32: the 256x256 tile never beats the three-stage kernel - 102 against 137 TF at 32768 x 768 x 3072 - and a K-tile is 32 floats)
        // The 256x256 tile only when its tiles fill whole rounds of the chip (260 tiles on 256 CUs are two rounds: 16416 x 1024 x 4096 took
        // 240 us on it against 170 us on the 128x128 kernel, tools/bench_gemm_tf.py); long-K problems of only a few rounds then go to the
        // 128x128 kernel, whose two co-resident workgroups cover each other (170 against 192 us for the three-stage tile there).
        const int rounds256 = cdiv(nwg256, n_cu);
        const bool fills256 = nwg256 >= n_cu && nwg256 * 100 >= rounds256 * n_cu * 85;
        if (v == 0)
            v = nwg4 >= 512 ? (ktiles <= 24 ? 4 : (sizeof(T) == 2 && ktiles >= 64 ? (fills256 ? 5 : (nwg4 < 4 * n_cu ? 1 : 3)) : 3)) : 1;
        // Mid-size problems at 16-24 K-tiles (the teacher-forced decoder stream: 8208 rows, K = 1024): two co-resident 128x128 workgroups beat the
        // persistent ring until it has ~8 rounds of tiles to run across (tools/bench_gemm_tf.py: N = 3072 89 -> 80 us, N = 4096 116 -> 100 us at
        // M = 8208; still ahead at M = 20000)
        if (g_gemm_variant == 0 && v == 4 && ktiles >= 16 && nwg4 < 2048) v = 1;
        // The persistent 256x256 ring of half-stages (6) where its tiles fill whole rounds of the chip (>= 95 %): 1.5x the flops per staged byte
        // pays on the decoder's 131072-token GEMMs and on N = 3072 (tools/bench_mae_gemms.py: lin1 + GELU 977 -> 868 us, lin2 537 -> 481,
        // dX of the in-projection 346 -> 307); it loses where a quarter of the last round idles (N = 768: 384 tiles) and ties on K = N = 512.
        if constexpr (sizeof(T) == 2 && EPI == 0) {
            static const bool no_p256 = getenv("ACAI_GEMM_NO_P256") != nullptr;   // A/B aid
            if (g_gemm_variant == 0 && !no_p256 && (v == 4 || v == 3) && ktiles < 64 && nwg256 >= n_cu && nwg256 * 100 >= rounds256 * n_cu * 95 &&
                !(g.N <= 512 && ktiles <= 8))
                v = 6;
        }
        // Round 3: wherever one of the large-tile kernels was chosen, the ping-pong ring (7) replaces it when the launch has one of its epilogue
        // forms (tools/bench_pp.py, MAE step shapes, same box: every shape faster - 0.82-1.02 PF on the plain / residual forms against 0.32-0.84).
        if constexpr (sizeof(T) == 2 && EPI == 0) {
            static const bool no_pp = getenv("ACAI_GEMM_NO_PP") != nullptr;   // A/B aid
            if (g_gemm_variant == 0 && !no_pp && v >= 3 && v <= 6) v = 7;
            // ... and it replaces the 128x128 kernel on mid-size problems when it needs fewer rounds of the chip: 128x128 tiles run two workgroups
            // per CU, a 256x256 tile (four of them) takes ~1.54 of their tile times on the ring.  The teacher-forced decoder stream (M = 16 x 513
            // = 8208 rows: 520 tiles of 128x128 at N = 1024, two rounds for 1.02 rounds of work) is the case: 37.9 -> 31.8 us at K = 1024,
            // 111.8 -> 96.0 us at K = 4096, N = 3072 76.7 -> 60.7 us (tools/bench_gemm_tf.py); M = 8192 (exactly one round) keeps the 128x128 kernel.
            if (g_gemm_variant == 0 && !no_pp && v == 1 && nwg256 >= 8) {
                const int rounds1 = cdiv(nwg, 2 * n_cu), rounds7 = cdiv(nwg256, n_cu);
                if (rounds7 * 154 < rounds1 * 100) v = 7;
            }
        }
        if (v == 7) {   // the register epilogue addresses C / residual / aux through 32-bit buffer offsets
            const size_t lim = 0xFFFFFF00ull, esz = g.out_dtype == ACAI_BF16 ? 2 : 4;
            const bool fits = (size_t)g.M * g.ldc * esz < lim && (!g.residual || (size_t)g.M * g.ldr * 4 < lim) && (!g.aux_mode || (size_t)g.M * g.ldaux * esz < lim);
            const bool src32 = (size_t)g.lda * 2 < (1u << 24) && (size_t)g.ldw * 2 < (1u << 24);   // 24-bit multiplies in the lane offsets of the LDS-DMA sources
            if (!(sizeof(T) == 2 && EPI == 0) || !fits || !src32 || !h.vec_epi || !pp_mode_ok(pp_mode(h))) v = 6;
        }
        if ((v == 5 || v == 6 || v == 7) && nwg256 < 8) v = 1;
        if (v == 4 && nwg4 < 8) v = 1;
        switch (v) {
            case 7:
                if constexpr (sizeof(T) == 2 && EPI == 0) {
#ifdef ACAI_GEMM_ABLATE
                    if (const char *d = getenv("ACAI_GEMM_DEBUG")) h.flags |= atoi(d) << 8;
#endif
                    const dim3 grid(nwg256 < n_cu ? nwg256 : n_cu), block(512);
                    int mode = pp_mode(h);
                    // the GELU forms can defer their GELU step into the next tile's R segments (see PP_DEFER) when a tile has enough of them;
                    // off by default: 833 against 686 us on the MAE decoder's lin1, 962 against 825 us on its gelu' GEMM (tools/bench_pp.py)
                    static const bool env_defer = getenv("ACAI_GEMM_PP_DEFER") && atoi(getenv("ACAI_GEMM_PP_DEFER")) != 0;   // A/B aid
                    if ((defer || env_defer) && g.K / BKG >= 8 && (mode == (PP_OBF | PP_GELU | PP_AUX1) || mode == (PP_OBF | PP_AUX2))) mode |= PP_DEFER;
                    switch (mode) {
#define PP_CASE(M) case (M): hipLaunchKernelGGL((gemm_nt_pp_kernel<T, (M)>), grid, block, 0, st, h); break
                        PP_CASE(PP_OBF | PP_GELU | PP_AUX1 | PP_DEFER);
                        PP_CASE(PP_OBF | PP_AUX2 | PP_DEFER);
                        PP_CASE(PP_OBF);
                        PP_CASE(PP_OBF | PP_SCALE);
                        PP_CASE(PP_OBF | PP_GELU | PP_AUX1);
                        PP_CASE(PP_OBF | PP_AUX2);
                        PP_CASE(PP_OBF | PP_GELU | PP_AUX3);
                        PP_CASE(PP_OBF | PP_AUX4);
                        PP_CASE(0);
                        PP_CASE(PP_RES);
#undef PP_CASE
                        default: break;   // unreachable: pp_mode_ok() sent every other form to variant 6
                    }
                    break;
                }
            case 6:
                if constexpr (sizeof(T) == 2 && EPI == 0) {
                    hipLaunchKernelGGL((gemm_nt_pers256_kernel<T, EPI>), dim3(nwg256 < n_cu ? nwg256 : n_cu), dim3(512), 0, st, h);
                    break;
                }
            case 5: hipLaunchKernelGGL((gemm_nt_256_kernel<T, EPI>), dim3(nwg256), dim3(512), 0, st, h); break;
            case 4: hipLaunchKernelGGL((gemm_nt_pers_kernel<T, EPI>), dim3(nwg4 < n_cu ? nwg4 : n_cu), dim3(512), 0, st, h); break;
            case 3: hipLaunchKernelGGL((gemm_nt_glds3_kernel<T, EPI>), dim3(nwg4), dim3(512), 0, st, h); break;
            case 2: hipLaunchKernelGGL((gemm_nt_glds_kernel<T, EPI, 4>), dim3(nwg4), dim3(512), 0, st, h); break;
            default: hipLaunchKernelGGL((gemm_nt_glds_kernel<T, EPI, 2>), dim3(nwg), dim3(256), 0, st, h); break;
        }
    }
    else if (fast)
        hipLaunchKernelGGL((gemm_nt_kernel<T, EPI, true, TA, TB>), dim3(nwg), dim3(256), 0, st, h);
    else
        hipLaunchKernelGGL((gemm_nt_kernel<T, EPI, false, TA, TB>), dim3(nwg), dim3(256), 0, st, h);
    ACAI_LAUNCH_CHECK("acai_gemm");
    return 0;
}

}  // namespace

extern "C" int acai_gemm_set_variant(int variant) {
    ACAI_CHECK_ARG(variant >= 0 && variant <= 8, "acai_gemm_set_variant: 0 (auto) .. 8");
    g_gemm_variant = variant;
    return 0;
}

extern "C" int acai_gemm_nt_ex(const void *A, int lda, const void *W, int ldw, const float *bias, const float *residual, int ldr,
                               void *C, int ldc, void *aux, int ldaux, int aux_mode, int M, int N, int K, int in_dtype, int out_dtype, int flags,
                               int scale_cols, float col_scale, void *stream) {
    ACAI_CHECK_ARG(A && W && C, "acai_gemm_nt: null operand");
    ACAI_CHECK_ARG(M >= 0 && N > 0 && K > 0, "acai_gemm_nt: bad shape M=%d N=%d K=%d", M, N, K);
    ACAI_CHECK_ARG(lda >= K && ldw >= K && ldc >= N && (!residual || ldr >= N), "acai_gemm_nt: leading dimension smaller than row");
    ACAI_CHECK_ARG((in_dtype == ACAI_F32 || in_dtype == ACAI_BF16) && (out_dtype == ACAI_F32 || out_dtype == ACAI_BF16),
                   "acai_gemm_nt: bad dtype");
    ACAI_CHECK_ARG(aux_mode >= 0 && aux_mode <= 4 && (aux_mode == 0 || (aux && ldaux >= N)), "acai_gemm_nt_ex: bad aux operand (mode %d)", aux_mode);
    ACAI_CHECK_ARG((aux_mode != 1 && aux_mode != 3) || (flags & ACAI_GEMM_GELU), "acai_gemm_nt_ex: aux_mode 1 / 3 keep the pre-activation / its derivative of a GELU epilogue");
    ACAI_CHECK_ARG((aux_mode != 2 && aux_mode != 4) || !(flags & ACAI_GEMM_GELU), "acai_gemm_nt_ex: aux_mode 2 / 4 (GELU derivative) exclude the GELU flag");
    ACAI_CHECK_ARG((flags & ~(ACAI_GEMM_GELU | ACAI_GEMM_ROUND_BF16)) == 0, "acai_gemm_nt: unknown flag bits 0x%x", flags);
    if (M == 0) return 0;
    GemmArgs g{};
    g.A = A; g.W = W; g.bias = bias; g.residual = residual; g.C = C;
    g.lda = lda; g.ldw = ldw; g.ldr = ldr; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.out_dtype = out_dtype; g.flags = flags;
    g.aux = aux; g.ldaux = ldaux; g.aux_mode = aux_mode;
    ACAI_CHECK_ARG(scale_cols >= 0 && scale_cols <= N, "acai_gemm_nt_ex: scale_cols %d outside [0, N]", scale_cols);
    g.scale_cols = scale_cols; g.col_scale = col_scale;
    return in_dtype == ACAI_BF16 ? launch<bf16_t, 0>(g, (hipStream_t)stream) : launch<float, 0>(g, (hipStream_t)stream);
}

extern "C" int acai_gemm_nt(const void *A, int lda, const void *W, int ldw, const float *bias, const float *residual, int ldr,
                            void *C, int ldc, int M, int N, int K, int in_dtype, int out_dtype, int flags, void *stream) {
    return acai_gemm_nt_ex(A, lda, W, ldw, bias, residual, ldr, C, ldc, nullptr, 0, 0, M, N, K, in_dtype, out_dtype, flags, 0, 1.0f, stream);
}

// General form for the backward pass: C[M,N] = op(A) . op(W)^T (+bias) (+residual), logical A [M,K], logical W [N,K];
// trans_a: A is stored [K][M] (lda = row stride of that storage); trans_w: W is stored [K][N].
//   dX = dY . W       : A = dY [M,N'] row-major,  W stored [N'][K'] = "[K_red][N_out]" -> trans_w = 1
//   dW = dY^T . X     : A = dY stored [M_red][N] -> trans_a = 1;  W-operand = X stored [M_red][K] -> trans_w = 1
extern "C" int acai_gemm(const void *A, int lda, int trans_a, const void *W, int ldw, int trans_w, const float *bias, const float *residual,
                         int ldr, void *C, int ldc, int M, int N, int K, int in_dtype, int out_dtype, int flags, void *stream) {
    ACAI_CHECK_ARG(A && W && C, "acai_gemm: null operand");
    ACAI_CHECK_ARG((flags & ~(ACAI_GEMM_GELU | ACAI_GEMM_ROUND_BF16)) == 0, "acai_gemm: unknown flag bits 0x%x", flags);
    ACAI_CHECK_ARG(M >= 0 && N > 0 && K > 0, "acai_gemm: bad shape M=%d N=%d K=%d", M, N, K);
    ACAI_CHECK_ARG(lda >= (trans_a ? M : K) && ldw >= (trans_w ? N : K) && ldc >= N && (!residual || ldr >= N), "acai_gemm: leading dimension smaller than row");
    ACAI_CHECK_ARG((in_dtype == ACAI_F32 || in_dtype == ACAI_BF16) && (out_dtype == ACAI_F32 || out_dtype == ACAI_BF16), "acai_gemm: bad dtype");
    ACAI_CHECK_ARG(!(trans_a && trans_w) || (out_dtype == ACAI_F32 && !bias && !residual && !flags),
                   "acai_gemm: trans_a && trans_w accumulates into an fp32 C (no bias / residual / flags)");
    if (M == 0) return 0;
    GemmArgs g{};
    g.A = A; g.W = W; g.bias = bias; g.residual = residual; g.C = C;
    g.lda = lda; g.ldw = ldw; g.ldr = ldr; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.out_dtype = out_dtype; g.flags = flags;
    hipStream_t st = (hipStream_t)stream;
    const int sel = (trans_a ? 2 : 0) | (trans_w ? 1 : 0);
    if (in_dtype == ACAI_BF16) {
        switch (sel) {
            case 0: return launch<bf16_t, 0, false, false>(g, st);
            case 1: return launch<bf16_t, 0, false, true>(g, st);
            case 2: return launch<bf16_t, 0, true, false>(g, st);
            default: return launch<bf16_t, 0, true, true>(g, st);
        }
    }
    switch (sel) {
        case 0: return launch<float, 0, false, false>(g, st);
        case 1: return launch<float, 0, false, true>(g, st);
        case 2: return launch<float, 0, true, false>(g, st);
        default: return launch<float, 0, true, true>(g, st);
    }
}

// dW[M][N] += dY^T X and, optionally, db[M] += column sums of dY: the two parameter gradients of an nn.Linear from one pass over dY
// (autograd of F.linear, models.py:29,57,...; torch computes them as mm + sum).  bf16 operands [K tokens][.], fp32 accumulators (zero them for
// fresh gradients).  The ping-pong weight-gradient kernel forms the sums from the dY fragments it holds; other shapes take acai_colsum.
extern "C" int acai_gemm_dw(const void *dY, int ldy, const void *X, int ldx, float *dW, int lddw, float *db, int M, int N, int K, int dtype, void *stream) {
    ACAI_CHECK_ARG(dY && X && dW, "acai_gemm_dw: null operand");
    ACAI_CHECK_ARG(M > 0 && N > 0 && K > 0 && ldy >= M && ldx >= N && lddw >= N, "acai_gemm_dw: bad shape M=%d N=%d K=%d", M, N, K);
    ACAI_CHECK_ARG(dtype == ACAI_F32 || dtype == ACAI_BF16, "acai_gemm_dw: bad dtype");
    GemmArgs g{};
    g.A = dY; g.W = X; g.C = dW; g.lda = ldy; g.ldw = ldx; g.ldc = lddw; g.M = M; g.N = N; g.K = K;
    g.out_dtype = ACAI_F32;
    g.colsum = dtype == ACAI_BF16 ? db : nullptr;
    int colsum_done = 0;   // set by the dispatch when the launched kernel also formed the column sums (else: a separate pass below)
    g.colsum_done = &colsum_done;
    hipStream_t st = (hipStream_t)stream;
    const int rc = dtype == ACAI_BF16 ? launch<bf16_t, 0, true, true>(g, st) : launch<float, 0, true, true>(g, st);
    if (rc) return rc;
    if (db && !colsum_done) return acai_colsum(dY, ldy, db, K, M, dtype, stream);
    return 0;
}

extern "C" int acai_cross_kv_prefill(const void *mem, int ldm, const void *Wkv, int ldw, const float *bkv, const int32_t *row_seq,
                                     const int32_t *row_pos, const int64_t *seq_off, const int32_t *seq_len, void *k_out,
                                     void *v_out, int M, int E, int H, int dh, int dhp, int dtype, int flags, void *stream) {
    ACAI_CHECK_ARG(mem && Wkv && row_seq && row_pos && seq_off && seq_len && k_out && v_out, "acai_cross_kv_prefill: null operand");
    ACAI_CHECK_ARG(E == H * dh && dhp >= dh && ldm >= E && ldw >= E, "acai_cross_kv_prefill: bad dims E=%d H=%d dh=%d dhp=%d", E, H, dh, dhp);
    ACAI_CHECK_ARG(dtype == ACAI_F32 || dtype == ACAI_BF16, "acai_cross_kv_prefill: bad dtype");
    if (M == 0) return 0;
    GemmArgs g{};
    g.A = mem; g.W = Wkv; g.bias = bkv; g.lda = ldm; g.ldw = ldw; g.M = M; g.N = 2 * E; g.K = E;
    g.out_dtype = dtype; g.flags = flags;
    g.row_seq = row_seq; g.row_pos = row_pos; g.seq_off = seq_off; g.seq_len = seq_len; g.k_out = k_out; g.v_out = v_out;
    g.E = E; g.dh = dh; g.dhp = dhp;
    return dtype == ACAI_BF16 ? launch<bf16_t, 1>(g, (hipStream_t)stream) : launch<float, 1>(g, (hipStream_t)stream);
}
