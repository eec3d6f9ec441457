// Varlen attention BACKWARD (autograd of the SDPA inside nn.MultiheadAttention: MAE pre-training backward,
// acai_omr/train/pre_train.py:59; teacher-forced backward, acai_omr/train/omr_teacher_force_train.py:118).
//
// Flash-style recompute from Q, K, V and the forward's log-sum-exp; no N x N tensor, no atomics, deterministic.
// Two kernels with the forward's MFMA machinery (32x32 tiles; the lane-owned side sits in registers as the B operand, the
// streamed side in LDS as the A operand, natural [row][d] image only - the transposed fragments come from ds_read_b64_tr_b16 /
// ds_read_b32; an accumulator tile is the next MFMA's B operand without touching LDS):
//   attn_bwd_dq  : workgroup = 128 queries (query on the lane), streams 64-key tiles:
//                    S^T = K Q^T, P^T = 2^(c S^T - lse),  dP^T = V dO^T,  dS^T = P^T o (dP^T - delta),  dQ^T += K^T dS^T
//   attn_bwd_dkv : workgroup = 128 keys (key on the lane), streams 64-query tiles:
//                    S = Q K^T,  P,  dV^T += dO^T P,  dP = dO V^T,  dS = P o (dP - delta),  dK^T += Q^T dS
// delta[q] = sum_d dO[q,d] O[q,d] is formed in the prologue of attn_bwd_dq (which runs first) and published for attn_bwd_dkv.  S and P are recomputed in both kernels (7 products instead
// of 5) - the price of having no cross-workgroup reduction.  fp32: v_mfma_f32_32x32x2_f32, bf16: v_mfma_f32_32x32x16_bf16.
#include "attn_bwd_args.h"
#include <type_traits>
#ifndef ACAI_DKV_STRAIGHT
#define ACAI_DKV_STRAIGHT 0
#endif
#ifndef ACAI_DKV_WAVES
#define ACAI_DKV_WAVES 3
#endif

namespace {

constexpr int TT = 64;   // streamed-side rows per tile
constexpr int OB = 128;  // lane-owned rows per workgroup

template <typename T, bool FAST>
__device__ __forceinline__ uint4 ld16(const T *base, int ld, int row, int rows, int d0, int dh) {
    constexpr int EPC = 16 / sizeof(T);
    uint4 r = make_uint4(0, 0, 0, 0);
    if (row >= rows) return r;
    if constexpr (FAST) {
        if (d0 < dh) r = *reinterpret_cast<const uint4 *>(base + (size_t)row * ld + d0);
    } else {
        union { uint4 v; T e[EPC]; } u;
        u.v = r;
#pragma unroll
        for (int e = 0; e < EPC; ++e)
            if (d0 + e < dh) u.e[e] = base[(size_t)row * ld + d0 + e];
        r = u.v;
    }
    return r;
}

// acc (32 x 32, rows in registers) += A(rows r0.. of an LDS tile, contraction-contiguous) . B(register fragments)
template <typename T, int DHP, int NS>
__device__ __forceinline__ void mma_rows(f32x16 &acc, const unsigned char *tile, int r0, int lr, int lh, const uint4 (&bf)[NS]) {
    typedef TileLayout<sizeof(T), DHP> TL;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const uint4 af = *reinterpret_cast<const uint4 *>(tile + TL::off(r0 + lr, 2 * s + lh));
        if constexpr (sizeof(T) == 2) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bf[s]), acc, 0, 0, 0);
        } else {
            const f32x4 a4 = __builtin_bit_cast(f32x4, af), b4 = __builtin_bit_cast(f32x4, bf[s]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b4[e], acc, 0, 0, 0);
        }
    }
}

// two independent products that share their loop structure (S and dP): the MFMAs alternate between the two accumulators, so no link of a
// dependent chain waits out the full MFMA latency
template <typename T, int DHP, int NS>
__device__ __forceinline__ void mma_rows2(f32x16 &acc0, const unsigned char *tile0, const uint4 (&bf0)[NS], f32x16 &acc1, const unsigned char *tile1,
                                          const uint4 (&bf1)[NS], int r0, int lr, int lh) {
    typedef TileLayout<sizeof(T), DHP> TL;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int o = TL::off(r0 + lr, 2 * s + lh);
        const uint4 a0 = *reinterpret_cast<const uint4 *>(tile0 + o), a1 = *reinterpret_cast<const uint4 *>(tile1 + o);
        if constexpr (sizeof(T) == 2) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, bf0[s]), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, bf1[s]), acc1, 0, 0, 0);
        } else {
            const f32x4 x0 = __builtin_bit_cast(f32x4, a0), y0 = __builtin_bit_cast(f32x4, bf0[s]);
            const f32x4 x1 = __builtin_bit_cast(f32x4, a1), y1 = __builtin_bit_cast(f32x4, bf1[s]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0[e], y0[e], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[e], y1[e], acc1, 0, 0, 0);
            }
        }
    }
}

// acc[c][lane] += tile^T . X: `tile` is a natural [row][col] LDS image (pitch bytes); the 32 contraction rows start at row r0,
// the 32 output rows are its columns c0..c0+31; X is a 32x32 accumulator tile whose register rows are the contraction rows.
// bf16: two 4-row x 16-col transposing reads (ds_read_b64_tr_b16) build the A fragment whose element j is contraction row
// 16 s2 + 8 (j>>2) + 4 lh + (j&3) - the order of X's registers 8 s2 + j.  fp32: one ds_read_b32 per K=2 MFMA.
template <typename T, int DHP>
__device__ __forceinline__ void mma_acc(f32x16 &acc, const unsigned char *tile, int r0, int c0, int lane, const f32x16 &x) {
    typedef TileLayout<sizeof(T), DHP> TL;
    constexpr int pitch = TL::PITCH;
    const int lr = lane & 31, lh = lane >> 5;
    if constexpr (sizeof(T) == 2) {
        typedef __attribute__((ext_vector_type(4))) short s4;
        typedef __attribute__((address_space(3))) s4 *lds_s4;
        const int i16 = lane & 15, g1 = (lane >> 4) & 1;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            uint4 xf;
            xf.x = pack_bf16(x[8 * s2 + 0], x[8 * s2 + 1]);
            xf.y = pack_bf16(x[8 * s2 + 2], x[8 * s2 + 3]);
            xf.z = pack_bf16(x[8 * s2 + 4], x[8 * s2 + 5]);
            xf.w = pack_bf16(x[8 * s2 + 6], x[8 * s2 + 7]);
            const int row = r0 + 16 * s2 + 4 * lh + (i16 >> 2), chunk = (c0 >> 3) + 2 * g1 + ((i16 & 3) >> 1), sub = 8 * (i16 & 1);
            union { s4 v[2]; uint4 u; } af;
            af.v[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(tile + TL::off(row, chunk) + sub));
            af.v[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(tile + TL::off(row + 8, chunk) + sub));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af.u), __builtin_bit_cast(bf16x8, xf), acc, 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const unsigned char *p = tile + (r0 + 8 * g4 + 4 * lh) * pitch + (c0 + lr) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(*reinterpret_cast<const float *>(p + e * pitch), x[4 * g4 + e], acc, 0, 0, 0);
        }
    }
}

// Streams 64-row tiles of two [rows][dh] operands (natural image) from global memory through registers into an LDS stage: the loads of
// tile t+1 are issued before the MFMAs of tile t and written to the other stage after them (issue-early / write-late), so a tile
// costs one barrier.  Per-thread global pointers advance by one tile per call; interior tiles load unguarded.
template <typename T, int DHP, bool FAST>
struct TileStager {
    typedef TileLayout<sizeof(T), DHP> TL;
    static constexpr int ES = sizeof(T), EPC = 16 / ES, RP = TL::PITCH, CPR = DHP / EPC, NCH = 64 * CPR / 256;
    const T *A, *B;
    int srow[NCH], soff[NCH], lda, ldb, dh;
    uint32_t ga[NCH], gb[NCH];   // FAST: this thread's byte offsets inside a tile (beyond every resource for chunks past d_h)
    uint4 ra[NCH], rb[NCH];
    __device__ __forceinline__ void init(const T *A_, int lda_, const T *B_, int ldb_, int tid, int dh_, int first_tile) {
        A = A_; B = B_; lda = lda_; ldb = ldb_; dh = dh_;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + 256 * i, row = c / CPR, cc = c % CPR;
            srow[i] = row;
            soff[i] = TL::off(row, cc);
            const bool dok = cc * EPC < dh;
            ga[i] = dok ? (uint32_t)((row * lda + cc * EPC) * ES) : 0xFFFFFFF0u;
            gb[i] = dok ? (uint32_t)((row * ldb + cc * EPC) * ES) : 0xFFFFFFF0u;
        }
    }
    // FAST (round 4): buffer loads through a resource rebuilt per tile in SGPRs - base = the tile's first row, num_records = the rows that exist
    // (only the VGPR offset is range-checked) - so rows past the sequence end and chunks past d_h read as zeros with no exec-mask branch, no
    // v_cndmask and no 64-bit pointer arithmetic per tile (the guarded global loads cost ~40 VALU / SALU instructions per tile on loops that
    // have no issue slots to spare).  The host checks that an operand's rows x pitch fits the 32-bit num_records.
    __device__ __forceinline__ void load(int t, int rows) {
        if constexpr (FAST) {
            const int left = rows - t * 64;
            const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(A + (size_t)t * 64 * lda), 0, left > 0 ? left * lda * ES : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(B + (size_t)t * 64 * ldb), 0, left > 0 ? left * ldb * ES : 0, 0x00020000);
            typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rsa, ga[i], 0, 0), y = __builtin_amdgcn_raw_buffer_load_b128(rsb, gb[i], 0, 0);
                ra[i] = make_uint4(x[0], x[1], x[2], x[3]);
                rb[i] = make_uint4(y[0], y[1], y[2], y[3]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int cc = (threadIdx.x + 256 * i) % CPR;
                ra[i] = ld16<T, false>(A, lda, t * 64 + srow[i], rows, cc * EPC, dh);
                rb[i] = ld16<T, false>(B, ldb, t * 64 + srow[i], rows, cc * EPC, dh);
            }
        }
    }
    __device__ __forceinline__ void store(unsigned char *stage) const {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            *reinterpret_cast<uint4 *>(stage + soff[i]) = ra[i];
            *reinterpret_cast<uint4 *>(stage + 64 * RP + soff[i]) = rb[i];
        }
    }
};


// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int DHP, bool FAST, bool DROP, bool PRE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void attn_bwd_dq_kernel(BwdArgs a) {
    constexpr int ES = sizeof(T);
    constexpr int RP = TileLayout<ES, DHP>::PITCH;   // pitch of the natural [row][d] tiles
    constexpr int NS = DHP * ES / 32, NDB = DHP / 32, STAGE = 2 * TT * RP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // two stages of {K tile, V tile}

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 31, lh = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q_start = a.cu_q[b], lq = a.cu_q[b + 1] - q_start;
    const int k_start = a.cu_k[b], lk = a.cu_k[b + 1] - k_start;
    const int q0 = ((a.tail256 & 1) ? (lq / 256) * 256 : 0) + blockIdx.x * OB;   // tail256 bit 0: only the rows past the last full 256-query block (attn_bwd64w.hip has the full ones)
    if (q0 >= lq) return;
    const int dh = a.dh;
    const T *Q = reinterpret_cast<const T *>(a.q) + (size_t)q_start * a.ldq + h * dh;
    const T *K = reinterpret_cast<const T *>(a.k) + (size_t)k_start * a.ldk + h * dh;
    const T *V = reinterpret_cast<const T *>(a.v) + (size_t)k_start * a.ldv + h * dh;
    const T *DO = reinterpret_cast<const T *>(a.dout) + (size_t)q_start * a.lddo + h * dh;
    T *DQ = reinterpret_cast<T *>(a.dq) + (size_t)q_start * a.lddq + h * dh;

    const int my_q = q0 + wave * 32 + lr;
    uint4 qf[NS], dof[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        qf[s] = ld16<T, FAST>(Q, a.ldq, my_q, lq, (s * 32 + lh * 16) / ES, dh);
        dof[s] = ld16<T, FAST>(DO, a.lddo, my_q, lq, (s * 32 + lh * 16) / ES, dh);
    }
    const size_t sidx = (size_t)h * a.total_q + q_start + (my_q < lq ? my_q : 0);
    const float lse = a.lse[sidx];
    // delta[q] = sum_d dO[q,d] O[q,d]: this lane holds its half of the row of dO already; O is read the same way, the two halves meet with one
    // cross-lane add, and the value is published for the dK/dV kernel that follows in the stream (no separate pass over O and dO)
    float dlt = 0.f;
    {
        const T *O = reinterpret_cast<const T *>(a.o) + (size_t)q_start * a.ldo + h * dh;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const uint4 of = ld16<T, FAST>(O, a.ldo, my_q, lq, (s * 32 + lh * 16) / ES, dh);
            if constexpr (ES == 2) {
                const uint32_t ow[4] = {of.x, of.y, of.z, of.w}, dw[4] = {dof[s].x, dof[s].y, dof[s].z, dof[s].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dlt = fmaf(__uint_as_float(ow[e] << 16), __uint_as_float(dw[e] << 16), dlt);
                    dlt = fmaf(__uint_as_float(ow[e] & 0xffff0000u), __uint_as_float(dw[e] & 0xffff0000u), dlt);
                }
            } else {
                const f32x4 o4 = __builtin_bit_cast(f32x4, of), d4 = __builtin_bit_cast(f32x4, dof[s]);
#pragma unroll
                for (int e = 0; e < 4; ++e) dlt = fmaf(o4[e], d4[e], dlt);
            }
        }
        dlt += __shfl_xor(dlt, 32);
        if (lh == 0 && my_q < lq) const_cast<float *>(a.delta)[sidx] = -dlt;   // negated: it is the dK/dV kernel's dP accumulator start, as loaded
    }

    f32x16 dqacc[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) dqacc[d][e] = 0.f;

    int nkt = (lk + TT - 1) / TT;
    if (a.causal) nkt = min(nkt, (min(q0 + OB, lq) - 1) / TT + 1);
    // PRE: q arrives as q * log2(e) / sqrt(d_h) (see acai_attn_varlen_fwd), so K . Q^T is the exponent's first term itself; the score
    // accumulators then START at -lse and the MFMA leaves (score - lse): one exp2 per probability and nothing else
    const float c = PRE ? 1.0f : a.scale_log2e;
    f32x16 sinit, pinit;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        sinit[e] = PRE ? -lse : 0.f;
        pinit[e] = DROP ? 0.f : -dlt;   // without dropout dP starts at -delta (a per-lane scalar here): dP - delta leaves the MFMA for free
    }
    const float dsub = DROP ? dlt : 0.f;
    TileStager<T, DHP, FAST> stg;
    stg.init(K, a.ldk, V, a.ldv, tid, dh, 0);
    stg.load(0, lk);
    stg.store(smem);
    __syncthreads();

    auto drop_dp = [&](int kt, int kb, f32x16 &dpacc) {  // dP = mask/(1-p) o (dO V^T)
        const uint32_t rrow = (uint32_t)(h * a.total_q + q_start + my_q);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const uint32_t key = (uint32_t)(kt * TT + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh);
            dpacc[e] = drop_keep(a.drop_seed, rrow, key, a.drop_thr) ? dpacc[e] * a.drop_scale : 0.f;
        }
    };

    // ---- fast loop: the leading tiles with every key valid (not causal).  One basic block per tile: the four S / dP products of both key
    // blocks are in flight before the first exponential is needed, and the common tile carries 1 exp2 + 1 multiply per score and one pack per
    // two (PRE).  The masked tiles live in their own loop below: sharing one loop body, the two paths' accumulator chains met in copies
    // (16 v_mov per tile) and split the tile into blocks the scheduler could not cross.
    constexpr bool STRAIGHT = !(ES == 4 && DHP == 64);   // fp32 d_h = 64 would spill with both blocks live: it stays in the general loop
    const int n_fast = (STRAIGHT && !a.causal) ? min(nkt, lk / TT) : 0;
    // a wave whose 32 queries all lie past the sequence end (513 decoder tokens: the fifth 128-query block holds ONE row) only helps with the
    // staging, in a loop of its own (see attn_fwd_kernel)
    const bool wave_active = q0 + wave * 32 < lq;
    if (!wave_active) {
        for (int kt = 0; kt < nkt; ++kt) {
            if (kt + 1 < nkt) {
                stg.load(kt + 1, lk);
                stg.store(smem + ((kt + 1) & 1) * STAGE);
            }
            __syncthreads();
        }
    } else {
    // (the LDS stages are compile-time offsets - the loop is unrolled by two: chosen by `kt & 1`, every swizzled fragment address cost a VALU add
    // per tile on top of its lane offset, 9-22 instructions per tile on a loop bound by the issue port)
    auto fast_tile = [&](int kt, const unsigned char *cur, unsigned char *nxt) {
        const unsigned char *ldsK = cur, *ldsV = cur + TT * RP;
        if (kt + 1 < nkt) stg.load(kt + 1, lk);
        f32x16 s0 = sinit, p0 = pinit, s1 = sinit, p1 = pinit;
        mma_rows2<T, DHP, NS>(s0, ldsK, qf, p0, ldsV, dof, 0, lr, lh);    // S^T[key][q] and dP^T[key][q] (- delta)
        mma_rows2<T, DHP, NS>(s1, ldsK, qf, p1, ldsV, dof, 32, lr, lh);
        if constexpr (DROP) {
            drop_dp(kt, 0, p0);
            drop_dp(kt, 1, p1);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) s0[e] = (PRE ? fast_exp2(s0[e]) : fast_exp2(fmaf(s0[e], c, -lse))) * (p0[e] - dsub);   // dS^T
#pragma unroll
        for (int d = 0; d < NDB; ++d) mma_acc<T, DHP>(dqacc[d], ldsK, 0, d * 32, lane, s0);  // dQ^T += K^T dS^T
#pragma unroll
        for (int e = 0; e < 16; ++e) s1[e] = (PRE ? fast_exp2(s1[e]) : fast_exp2(fmaf(s1[e], c, -lse))) * (p1[e] - dsub);
#pragma unroll
        for (int d = 0; d < NDB; ++d) mma_acc<T, DHP>(dqacc[d], ldsK, 32, d * 32, lane, s1);
        if (kt + 1 < nkt) stg.store(nxt);
        __syncthreads();
    };
    {
        int kt = 0;
        for (; kt + 1 < n_fast; kt += 2) {
            fast_tile(kt, smem, smem + STAGE);
            fast_tile(kt + 1, smem + STAGE, smem);
        }
        if (kt < n_fast) fast_tile(kt, smem, smem + STAGE);
    }
    // ---- general loop: ragged last tile, causal tiles ------------------------------------------------------------------------------------
    for (int kt = n_fast; kt < nkt; ++kt) {
        const unsigned char *ldsK = smem + (kt & 1) * STAGE, *ldsV = ldsK + TT * RP;
        if (kt + 1 < nkt) stg.load(kt + 1, lk);
        const int key_lim = a.causal ? min(lk, my_q + 1) : lk;
        const bool interior = (kt + 1) * TT <= (a.causal ? min(lk, q0 + wave * 32 + 1) : lk);  // every key valid for the whole wave
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (kt * TT + kb * 32 >= lk) continue;   // a 32-key block entirely past the sequence end
            f32x16 sacc, dpacc = pinit;
#pragma unroll
            for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
            mma_rows2<T, DHP, NS>(sacc, ldsK, qf, dpacc, ldsV, dof, kb * 32, lr, lh);
            if constexpr (DROP) drop_dp(kt, kb, dpacc);
            if (interior) {
#pragma unroll
                for (int e = 0; e < 16; ++e) sacc[e] = fast_exp2(fmaf(sacc[e], c, -lse)) * (dpacc[e] - dsub);   // dS^T
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = kt * TT + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    const float pr = key < key_lim ? fast_exp2(sacc[e] * c - lse) : 0.f;
                    sacc[e] = pr * (dpacc[e] - dsub);                        // dS^T
                }
            }
#pragma unroll
            for (int d = 0; d < NDB; ++d) mma_acc<T, DHP>(dqacc[d], ldsK, kb * 32, d * 32, lane, sacc);  // dQ^T += K^T dS^T
        }
        if (kt + 1 < nkt) stg.store(smem + ((kt + 1) & 1) * STAGE);
        __syncthreads();
    }
    }   // wave_active
    if (my_q < lq) {
        T *row = DQ + (size_t)my_q * a.lddq;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d0 = d * 32 + 8 * g4 + 4 * lh;   // registers 4 g4 .. 4 g4 + 3 are four consecutive d: one 8- / 16-byte store
                if constexpr (FAST) {
                    if (d0 < dh) {
                        if constexpr (ES == 2) {
                            uint2 pk;
                            pk.x = pack_bf16(dqacc[d][4 * g4 + 0] * a.scale, dqacc[d][4 * g4 + 1] * a.scale);
                            pk.y = pack_bf16(dqacc[d][4 * g4 + 2] * a.scale, dqacc[d][4 * g4 + 3] * a.scale);
                            *reinterpret_cast<uint2 *>(row + d0) = pk;
                        } else {
                            *reinterpret_cast<float4 *>(row + d0) = make_float4(dqacc[d][4 * g4 + 0] * a.scale, dqacc[d][4 * g4 + 1] * a.scale,
                                                                               dqacc[d][4 * g4 + 2] * a.scale, dqacc[d][4 * g4 + 3] * a.scale);
                        }
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (d0 + e < dh) DT<T>::st(row + d0 + e, dqacc[d][4 * g4 + e] * a.scale);
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int DHP, bool FAST, bool DROP, bool PRE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((sizeof(T) == 4 && DHP == 64) ? 1 : ((sizeof(T) == 2 && DHP == 32) ? ACAI_DKV_WAVES : 2)))) void attn_bwd_dkv_kernel(BwdArgs a) {
    constexpr int ES = sizeof(T);
    constexpr int RP = TileLayout<ES, DHP>::PITCH;
    constexpr int NS = DHP * ES / 32, NDB = DHP / 32, STAGE = 2 * TT * RP + 2 * TT * (int)sizeof(float);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // two stages of {Q tile, dO tile, lse[64], delta[64]}

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 31, lh = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q_start = a.cu_q[b], lq = a.cu_q[b + 1] - q_start;
    const int k_start = a.cu_k[b], lk = a.cu_k[b + 1] - k_start;
    const int k0 = ((a.tail256 & 2) ? (lk / 256) * 256 : 0) + blockIdx.x * OB;   // tail256 bit 1: only the keys past the last full 256-key block
    if (k0 >= lk) return;
    const int dh = a.dh;
    const T *Q = reinterpret_cast<const T *>(a.q) + (size_t)q_start * a.ldq + h * dh;
    const T *K = reinterpret_cast<const T *>(a.k) + (size_t)k_start * a.ldk + h * dh;
    const T *V = reinterpret_cast<const T *>(a.v) + (size_t)k_start * a.ldv + h * dh;
    const T *DO = reinterpret_cast<const T *>(a.dout) + (size_t)q_start * a.lddo + h * dh;
    T *DK = reinterpret_cast<T *>(a.dk) + (size_t)k_start * a.lddk + h * dh;
    T *DV = reinterpret_cast<T *>(a.dv) + (size_t)k_start * a.lddv + h * dh;

    const int my_k = k0 + wave * 32 + lr;
    uint4 kf[NS], vf[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        kf[s] = ld16<T, FAST>(K, a.ldk, my_k, lk, (s * 32 + lh * 16) / ES, dh);
        vf[s] = ld16<T, FAST>(V, a.ldv, my_k, lk, (s * 32 + lh * 16) / ES, dh);
    }
    f32x16 dkacc[NDB], dvacc[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) dkacc[d][e] = dvacc[d][e] = 0.f;

    const int nqt = (lq + TT - 1) / TT;
    const int qt0 = a.causal ? k0 / TT : 0;  // queries before this key block never attend to it
    TileStager<T, DHP, FAST> stg;
    stg.init(Q, a.ldq, DO, a.lddo, tid, dh, qt0);
    const float *lse_row = a.lse + (size_t)h * a.total_q + q_start, *dlt_row = a.delta + (size_t)h * a.total_q + q_start;
    const float c = PRE ? 1.0f : a.scale_log2e;   // PRE: see attn_bwd_dq_kernel
    // one query row's statistics per thread: threads 0..63 (and, redundantly, 128..191) fetch lse, 64..127 (192..255) -delta.  Every thread
    // issues the load - under a `tid < 64` branch the compiler put s_waitcnt vmcnt in front of the tile's first MFMAs, which made every wave
    // wait out its own prefetch of the next tile
    float r_stat = 0.f;
    const float *stat_row = (tid & 64) ? dlt_row : lse_row;
    auto load_stats = [&](int qt) {
        const int qq = qt * TT + (tid & 63);
        r_stat = stat_row[qq < lq ? qq : 0];
    };
    // both go to LDS NEGATED (attn_bwd_dq published -delta already): they are accumulator start values.  The sign flip sits here, a tile
    // after the load, not next to it
    auto store_stats = [&](unsigned char *stage) {
        if (tid < 2 * TT) reinterpret_cast<float *>(stage + 2 * TT * RP)[tid] = (tid & 64) ? r_stat : -r_stat;
    };
    if (qt0 < nqt) {
        stg.load(qt0, lq);
        load_stats(qt0);
        stg.store(smem);
        store_stats(smem);
    }
    // Every global load so far (the K / V fragments above all) is complete on EVERY path into the loops.  Without this the path around the `if`
    // leaves them pending as far as the compiler's wait-count bookkeeping knows, and it then puts s_waitcnt vmcnt(1) / vmcnt(0) in front of the
    // first MFMAs of every tile that read those fragments - which waits out the prefetch of the next tile issued a few instructions earlier
    // (PMC: 40 % of the wave cycles parked in s_waitcnt).
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt and lgkmcnt untouched
    __syncthreads();

    // S[q][key] and dP[q][key] of one 32-query block.  The 16 query rows a lane holds are rows 8 g4 + 4 lh + (0..3): their statistics come as
    // one 16-byte LDS read each, straight into the accumulators' start values: dP starts at -delta (dP - delta leaves the MFMA for free), and
    // with PRE the score starts at -lse (the MFMA leaves score - lse: one exp2 per probability)
    auto s_dp = [&](const unsigned char *cur, int qb, f32x16 &sacc, f32x16 &dpacc, f32x4 (&nlse4)[4], f32x4 (&ndl4)[4]) {
        const float *ldsNlse = reinterpret_cast<const float *>(cur + 2 * TT * RP), *ldsNdl = ldsNlse + TT;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            nlse4[g4] = *reinterpret_cast<const f32x4 *>(ldsNlse + qb * 32 + 8 * g4 + 4 * lh);
            ndl4[g4] = *reinterpret_cast<const f32x4 *>(ldsNdl + qb * 32 + 8 * g4 + 4 * lh);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            sacc[e] = PRE ? nlse4[e >> 2][e & 3] : 0.f;
            dpacc[e] = DROP ? 0.f : ndl4[e >> 2][e & 3];
        }
        mma_rows2<T, DHP, NS>(sacc, cur, kf, dpacc, cur + TT * RP, vf, qb * 32, lr, lh);
    };
    // P and dS of the block (in place of S and dP), then dV^T += dO^T P and dK^T += Q^T dS.  mode 0: every (query, key) of the block is valid
    auto ds_dkv = [&](const unsigned char *cur, int qt, int qb, f32x16 &sacc, f32x16 &dpacc, const f32x4 (&nlse4)[4], const f32x4 (&ndl4)[4], bool inter) {
        const float s_off = PRE ? 0.f : 1.f;   // PRE: -lse is already inside sacc
        if (DROP) {  // keep mask of (query row, my key): dP is masked, and so is the P that multiplies dO for dV
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ql = qb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh, qq = qt * TT + ql;
                const bool ok = qq < lq && my_k < lk && (!a.causal || my_k <= qq);
                const float pr = ok ? fast_exp2(fmaf(sacc[e], c, s_off * nlse4[e >> 2][e & 3])) : 0.f;
                const float mk = drop_keep(a.drop_seed, (uint32_t)(h * a.total_q + q_start + qq), (uint32_t)my_k, a.drop_thr) ? a.drop_scale : 0.f;
                sacc[e] = pr * mk;                                                  // dropped P (for dV)
                dpacc[e] = pr * (dpacc[e] * mk + ndl4[e >> 2][e & 3]);              // dS
            }
        } else if (inter) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float pr = PRE ? fast_exp2(sacc[e]) : fast_exp2(fmaf(sacc[e], c, nlse4[e >> 2][e & 3]));
                sacc[e] = pr;                                           // P
                dpacc[e] = pr * dpacc[e];                               // dS (delta already subtracted)
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ql = qb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh, qq = qt * TT + ql;
                const bool ok = qq < lq && my_k < lk && (!a.causal || my_k <= qq);
                const float pr = ok ? fast_exp2(fmaf(sacc[e], c, s_off * nlse4[e >> 2][e & 3])) : 0.f;
                sacc[e] = pr;
                dpacc[e] = pr * dpacc[e];
            }
        }
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
            mma_acc<T, DHP>(dvacc[d], cur + TT * RP, qb * 32, d * 32, lane, sacc);   // dV^T += dO^T P
            mma_acc<T, DHP>(dkacc[d], cur, qb * 32, d * 32, lane, dpacc);            // dK^T += Q^T dS
        }
    };
    auto advance = [&](int qt) {
        if (qt + 1 < nqt) {
            unsigned char *nxt = smem + ((qt + 1 - qt0) & 1) * STAGE;
            stg.store(nxt);
            store_stats(nxt);
        }
        __syncthreads();
    };

    // ---- fast loop (not causal, every key of the workgroup valid): the leading full query tiles, without masks.  Its own loop for the reason
    // given in attn_bwd_dq_kernel.  ACAI_DKV_STRAIGHT: both 32-query blocks in one basic block (needs the registers of two waves per SIMD at
    // d_h = 32; three waves per SIMD with one block at a time measured faster - tools/ab_attn.sh)
    const int n_fast = (!a.causal && k0 + OB <= lk) ? lq / TT : 0;
    constexpr bool STRAIGHT = ACAI_DKV_STRAIGHT && !(ES == 2 && DHP == 64);
    // (n_fast > 0 only without the causal mask, i.e. qt0 = 0: tile qt sits in stage qt & 1.  The stages are compile-time offsets, the loop
    // unrolled by two - see attn_bwd_dq_kernel)
    auto fast_tile = [&](int qt, const unsigned char *cur, unsigned char *nxt) {
        if (qt + 1 < nqt) {
            stg.load(qt + 1, lq);
            load_stats(qt + 1);
        }
        if constexpr (STRAIGHT) {
            f32x16 s0, p0, s1, p1;
            f32x4 l0[4], n0[4], l1[4], n1[4];
            s_dp(cur, 0, s0, p0, l0, n0);
            s_dp(cur, 1, s1, p1, l1, n1);
            ds_dkv(cur, qt, 0, s0, p0, l0, n0, true);
            ds_dkv(cur, qt, 1, s1, p1, l1, n1, true);
        } else {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                f32x16 sacc, dpacc;
                f32x4 nlse4[4], ndl4[4];
                s_dp(cur, qb, sacc, dpacc, nlse4, ndl4);
                ds_dkv(cur, qt, qb, sacc, dpacc, nlse4, ndl4, true);
            }
        }
        if (qt + 1 < nqt) {
            stg.store(nxt);
            store_stats(nxt);
        }
        __syncthreads();
    };
    if constexpr (ES == 2 && DHP == 64) {
        // (bf16 d_h = 64 holds 256 registers: unrolled, the allocator spilt and the backward measured 0-2 % slower - it keeps the rolled loop)
        for (int qt = 0; qt < n_fast; ++qt) fast_tile(qt, smem + (qt & 1) * STAGE, smem + ((qt + 1) & 1) * STAGE);
    } else {
        int qt = 0;
        for (; qt + 1 < n_fast; qt += 2) {
            fast_tile(qt, smem, smem + STAGE);
            fast_tile(qt + 1, smem + STAGE, smem);
        }
        if (qt < n_fast) fast_tile(qt, smem, smem + STAGE);
    }
    // ---- general loop: ragged tiles, causal tiles -----------------------------------------------------------------------------------------
    for (int qt = max(qt0, n_fast); qt < nqt; ++qt) {
        const unsigned char *cur = smem + ((qt - qt0) & 1) * STAGE;
        if (qt + 1 < nqt) {
            stg.load(qt + 1, lq);
            load_stats(qt + 1);
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            if (qt * TT + qb * 32 >= lq) continue;   // a 32-query block entirely past the sequence end (513 tokens: the last tile holds one row)
            f32x16 sacc, dpacc;
            f32x4 nlse4[4], ndl4[4];
            s_dp(cur, qb, sacc, dpacc, nlse4, ndl4);
            // interior block: every query of the tile exists, every key of the WAVE exists and (causal) lies at or before the block's first query
            const bool inter = (qt + 1) * TT <= lq && k0 + wave * 32 + 32 <= lk && (!a.causal || k0 + wave * 32 + 31 <= qt * TT + qb * 32);
            ds_dkv(cur, qt, qb, sacc, dpacc, nlse4, ndl4, inter);
        }
        advance(qt);
    }
    if (my_k < lk) {
        T *rk = DK + (size_t)my_k * a.lddk, *rv = DV + (size_t)my_k * a.lddv;
        const float ksc = PRE ? 0.6931471805599453f : a.scale;   // PRE: dK = dS^T Q = dS^T Q' sqrt(d_h) / log2(e), times 1/sqrt(d_h)
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d0 = d * 32 + 8 * g4 + 4 * lh;   // registers 4 g4 .. 4 g4 + 3 are four consecutive d: one 8- / 16-byte store each
                if constexpr (FAST) {
                    if (d0 < dh) {
                        if constexpr (ES == 2) {
                            float gk[4], gv[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                gk[e] = dkacc[d][4 * g4 + e] * ksc;
                                gv[e] = dvacc[d][4 * g4 + e];
                            }
                            if (a.accum_dkv) {   // += : the gradient another pass over the same K / V left here (added in fp32, one rounding)
                                const uint2 ok = *reinterpret_cast<const uint2 *>(rk + d0), ov = *reinterpret_cast<const uint2 *>(rv + d0);
                                gk[0] += __uint_as_float(ok.x << 16); gk[1] += __uint_as_float(ok.x & 0xffff0000u);
                                gk[2] += __uint_as_float(ok.y << 16); gk[3] += __uint_as_float(ok.y & 0xffff0000u);
                                gv[0] += __uint_as_float(ov.x << 16); gv[1] += __uint_as_float(ov.x & 0xffff0000u);
                                gv[2] += __uint_as_float(ov.y << 16); gv[3] += __uint_as_float(ov.y & 0xffff0000u);
                            }
                            uint2 pk, pv;
                            pk.x = pack_bf16(gk[0], gk[1]);
                            pk.y = pack_bf16(gk[2], gk[3]);
                            pv.x = pack_bf16(gv[0], gv[1]);
                            pv.y = pack_bf16(gv[2], gv[3]);
                            *reinterpret_cast<uint2 *>(rk + d0) = pk;
                            *reinterpret_cast<uint2 *>(rv + d0) = pv;
                        } else {
                            *reinterpret_cast<float4 *>(rk + d0) = make_float4(dkacc[d][4 * g4 + 0] * ksc, dkacc[d][4 * g4 + 1] * ksc,
                                                                              dkacc[d][4 * g4 + 2] * ksc, dkacc[d][4 * g4 + 3] * ksc);
                            *reinterpret_cast<float4 *>(rv + d0) = make_float4(dvacc[d][4 * g4 + 0], dvacc[d][4 * g4 + 1], dvacc[d][4 * g4 + 2], dvacc[d][4 * g4 + 3]);
                        }
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (d0 + e < dh) {
                            DT<T>::st(rk + d0 + e, dkacc[d][4 * g4 + e] * ksc);
                            DT<T>::st(rv + d0 + e, dvacc[d][4 * g4 + e]);
                        }
                }
            }
    }
}

// ---- two lane-owned 32-row blocks per wave (round 3) ---------------------------------------------------------------------------------------
// The training steps' hot form - bf16, q prescaled, no dropout, no causal mask, d_h <= 32 - with a wave owning 64 queries (dQ) or 64 keys
// (dK / dV): every fragment it reads from the streamed tile (ds_read_b128 of the rows, ds_read_b64_tr_b16 of the transposed image) feeds
// TWO MFMAs, one per owned block, and the tile loop's fixed costs (staging, statistics reads, waits, barrier, branches) are spent once per
// 4096 scores instead of 2048.  The kernels are bound by the issue port (section 5 of DESIGN.md): dQ 126 -> ~108, dK/dV 179 -> ~150
// instructions per 2048 scores.  Costs the third wave per SIMD (~210-230 registers).  Same arithmetic, same summation order per output as the
// one-block kernels above (which keep every other case: causal masks, dropout, fp32, d_h = 64).
template <int NQ, int NS>
__device__ __forceinline__ void mma_rows2q(f32x16 (&acc0)[NQ], const unsigned char *tile0, const uint4 (&bf0)[NQ][NS], f32x16 (&acc1)[NQ],
                                           const unsigned char *tile1, const uint4 (&bf1)[NQ][NS], int r0, int lr, int lh) {
    typedef TileLayout<2, 32> TL;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int o = TL::off(r0 + lr, 2 * s + lh);
        const uint4 a0 = *reinterpret_cast<const uint4 *>(tile0 + o), a1 = *reinterpret_cast<const uint4 *>(tile1 + o);   // one read each, NQ products
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            acc0[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, bf0[j][s]), acc0[j], 0, 0, 0);
            acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, bf1[j][s]), acc1[j], 0, 0, 0);
        }
    }
}

// acc[j] += tile^T . x[j] (see mma_acc): the transposed A fragment is read once for the NQ accumulator tiles
template <int NQ>
__device__ __forceinline__ void mma_accq(f32x16 (&acc)[NQ], const unsigned char *tile, int r0, int lane, const f32x16 (&x)[NQ]) {
    typedef TileLayout<2, 32> TL;
    typedef __attribute__((ext_vector_type(4))) short s4;
    typedef __attribute__((address_space(3))) s4 *lds_s4;
    const int lh = lane >> 5, i16 = lane & 15, g1 = (lane >> 4) & 1;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const int row = r0 + 16 * s2 + 4 * lh + (i16 >> 2), chunk = 2 * g1 + ((i16 & 3) >> 1), sub = 8 * (i16 & 1);
        union { s4 v[2]; uint4 u; } af;
        af.v[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(tile + TL::off(row, chunk) + sub));
        af.v[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(tile + TL::off(row + 8, chunk) + sub));
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            uint4 xf;
            xf.x = pack_bf16(x[j][8 * s2 + 0], x[j][8 * s2 + 1]);
            xf.y = pack_bf16(x[j][8 * s2 + 2], x[j][8 * s2 + 3]);
            xf.z = pack_bf16(x[j][8 * s2 + 4], x[j][8 * s2 + 5]);
            xf.w = pack_bf16(x[j][8 * s2 + 6], x[j][8 * s2 + 7]);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af.u), __builtin_bit_cast(bf16x8, xf), acc[j], 0, 0, 0);
        }
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void attn_bwd_dq2_kernel(BwdArgs a) {
    typedef bf16_t T;
    constexpr int NQ = 2, DHP = 32, RP = TileLayout<2, DHP>::PITCH, NS = 2, STAGE = 2 * TT * RP, QBW = 32 * NQ, QBG = 4 * QBW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // two stages of {K tile, V tile}
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 31, lh = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q_start = a.cu_q[b], lq = a.cu_q[b + 1] - q_start;
    const int k_start = a.cu_k[b], lk = a.cu_k[b + 1] - k_start;
    const int q0 = blockIdx.x * QBG;
    if (q0 >= lq) return;
    const int dh = a.dh;
    const T *Q = reinterpret_cast<const T *>(a.q) + (size_t)q_start * a.ldq + h * dh;
    const T *K = reinterpret_cast<const T *>(a.k) + (size_t)k_start * a.ldk + h * dh;
    const T *V = reinterpret_cast<const T *>(a.v) + (size_t)k_start * a.ldv + h * dh;
    const T *DO = reinterpret_cast<const T *>(a.dout) + (size_t)q_start * a.lddo + h * dh;
    const T *O = reinterpret_cast<const T *>(a.o) + (size_t)q_start * a.ldo + h * dh;
    T *DQ = reinterpret_cast<T *>(a.dq) + (size_t)q_start * a.lddq + h * dh;

    int my_q[NQ];
    uint4 qf[NQ][NS], dof[NQ][NS];
    f32x16 dqacc[NQ], sinit[NQ], pinit[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        my_q[j] = q0 + wave * QBW + j * 32 + lr;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            qf[j][s] = ld16<T, true>(Q, a.ldq, my_q[j], lq, s * 16 + lh * 8, dh);
            dof[j][s] = ld16<T, true>(DO, a.lddo, my_q[j], lq, s * 16 + lh * 8, dh);
        }
        const size_t sidx = (size_t)h * a.total_q + q_start + (my_q[j] < lq ? my_q[j] : 0);
        const float lse = a.lse[sidx];
        float dlt = 0.f;   // delta[q] = sum_d dO[q,d] O[q,d], published (negated) for the dK/dV kernel - see attn_bwd_dq_kernel
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const uint4 of = ld16<T, true>(O, a.ldo, my_q[j], lq, s * 16 + lh * 8, dh);
            const uint32_t ow[4] = {of.x, of.y, of.z, of.w}, dw[4] = {dof[j][s].x, dof[j][s].y, dof[j][s].z, dof[j][s].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dlt = fmaf(__uint_as_float(ow[e] << 16), __uint_as_float(dw[e] << 16), dlt);
                dlt = fmaf(__uint_as_float(ow[e] & 0xffff0000u), __uint_as_float(dw[e] & 0xffff0000u), dlt);
            }
        }
        dlt += __shfl_xor(dlt, 32);
        if (lh == 0 && my_q[j] < lq) const_cast<float *>(a.delta)[sidx] = -dlt;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            dqacc[j][e] = 0.f;
            sinit[j][e] = -lse;     // the score accumulators start at -lse, dP at -delta: the MFMAs leave (score - lse) and (dP - delta)
            pinit[j][e] = -dlt;
        }
    }
    const int nkt = (lk + TT - 1) / TT;
    TileStager<T, DHP, true> stg;
    stg.init(K, a.ldk, V, a.ldv, tid, dh, 0);
    stg.load(0, lk);
    stg.store(smem);
    __syncthreads();
    const bool wave_active = q0 + wave * QBW < lq;
    if (!wave_active) {   // (only helps with the staging: see attn_fwd_kernel)
        for (int kt = 0; kt < nkt; ++kt) {
            if (kt + 1 < nkt) {
                stg.load(kt + 1, lk);
                stg.store(smem + ((kt + 1) & 1) * STAGE);
            }
            __syncthreads();
        }
    } else {
        auto tile = [&](int kt, const unsigned char *cur, unsigned char *nxt, bool masked) {
            const unsigned char *ldsK = cur, *ldsV = cur + TT * RP;
            if (kt + 1 < nkt) stg.load(kt + 1, lk);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (masked && kt * TT + kb * 32 >= lk) continue;   // a 32-key block entirely past the sequence end
                f32x16 sc[NQ], dp[NQ];
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    sc[j] = sinit[j];
                    dp[j] = pinit[j];
                }
                mma_rows2q<NQ, NS>(sc, ldsK, qf, dp, ldsV, dof, kb * 32, lr, lh);    // S^T[key][q] - lse and dP^T[key][q] - delta
#pragma unroll
                for (int j = 0; j < NQ; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float pr = fast_exp2(sc[j][e]);
                        if (masked) pr = (kt * TT + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) < lk ? pr : 0.f;
                        sc[j][e] = pr * dp[j][e];   // dS^T
                    }
                mma_accq<NQ>(dqacc, ldsK, kb * 32, lane, sc);   // dQ^T += K^T dS^T
            }
            if (kt + 1 < nkt) stg.store(nxt);
            __syncthreads();
        };
        const int n_fast = lk / TT;   // tiles with every key valid
        int kt = 0;
        for (; kt + 1 < n_fast; kt += 2) {   // (unrolled by two: the LDS stages are compile-time offsets)
            tile(kt, smem, smem + STAGE, false);
            tile(kt + 1, smem + STAGE, smem, false);
        }
        if (kt < n_fast) {
            tile(kt, smem, smem + STAGE, false);
            ++kt;
        }
        for (; kt < nkt; ++kt) tile(kt, smem + (kt & 1) * STAGE, smem + ((kt + 1) & 1) * STAGE, true);
    }
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        if (my_q[j] >= lq) continue;
        T *row = DQ + (size_t)my_q[j] * a.lddq;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int d0 = 8 * g4 + 4 * lh;   // registers 4 g4 .. 4 g4 + 3 are four consecutive d: one 8-byte store
            if (d0 < dh) {
                uint2 pk;
                pk.x = pack_bf16(dqacc[j][4 * g4 + 0] * a.scale, dqacc[j][4 * g4 + 1] * a.scale);
                pk.y = pack_bf16(dqacc[j][4 * g4 + 2] * a.scale, dqacc[j][4 * g4 + 3] * a.scale);
                *reinterpret_cast<uint2 *>(row + d0) = pk;
            }
        }
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void attn_bwd_dkv2_kernel(BwdArgs a) {
    typedef bf16_t T;
    constexpr int NK = 2, DHP = 32, RP = TileLayout<2, DHP>::PITCH, NS = 2, STAGE = 2 * TT * RP + 2 * TT * (int)sizeof(float), KBW = 32 * NK, KBG = 4 * KBW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // two stages of {Q tile, dO tile, -lse[64], -delta[64]}
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 31, lh = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q_start = a.cu_q[b], lq = a.cu_q[b + 1] - q_start;
    const int k_start = a.cu_k[b], lk = a.cu_k[b + 1] - k_start;
    const int k0 = blockIdx.x * KBG;
    if (k0 >= lk) return;
    const int dh = a.dh;
    const T *Q = reinterpret_cast<const T *>(a.q) + (size_t)q_start * a.ldq + h * dh;
    const T *K = reinterpret_cast<const T *>(a.k) + (size_t)k_start * a.ldk + h * dh;
    const T *V = reinterpret_cast<const T *>(a.v) + (size_t)k_start * a.ldv + h * dh;
    const T *DO = reinterpret_cast<const T *>(a.dout) + (size_t)q_start * a.lddo + h * dh;
    T *DK = reinterpret_cast<T *>(a.dk) + (size_t)k_start * a.lddk + h * dh;
    T *DV = reinterpret_cast<T *>(a.dv) + (size_t)k_start * a.lddv + h * dh;

    int my_k[NK];
    uint4 kf[NK][NS], vf[NK][NS];
    f32x16 dkacc[NK], dvacc[NK];
#pragma unroll
    for (int j = 0; j < NK; ++j) {
        my_k[j] = k0 + wave * KBW + j * 32 + lr;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            kf[j][s] = ld16<T, true>(K, a.ldk, my_k[j], lk, s * 16 + lh * 8, dh);
            vf[j][s] = ld16<T, true>(V, a.ldv, my_k[j], lk, s * 16 + lh * 8, dh);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) dkacc[j][e] = dvacc[j][e] = 0.f;
    }
    const int nqt = (lq + TT - 1) / TT;
    TileStager<T, DHP, true> stg;
    stg.init(Q, a.ldq, DO, a.lddo, tid, dh, 0);
    const float *lse_row = a.lse + (size_t)h * a.total_q + q_start, *dlt_row = a.delta + (size_t)h * a.total_q + q_start;
    // one query row's statistics per thread (every thread issues the load, see attn_bwd_dkv_kernel); -delta arrives negated, lse is negated here
    float r_stat = 0.f;
    const float *stat_row = (tid & 64) ? dlt_row : lse_row;
    auto load_stats = [&](int qt) {
        const int qq = qt * TT + (tid & 63);
        r_stat = stat_row[qq < lq ? qq : 0];
    };
    auto store_stats = [&](unsigned char *stage) {
        if (tid < 2 * TT) reinterpret_cast<float *>(stage + 2 * TT * RP)[tid] = (tid & 64) ? r_stat : -r_stat;
    };
    stg.load(0, lq);
    load_stats(0);
    stg.store(smem);
    store_stats(smem);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the K / V fragments are complete on every path into the loop (see attn_bwd_dkv_kernel)
    __syncthreads();
    const bool wave_active = k0 + wave * KBW < lk;   // a wave whose 64 keys all lie past the end only stages
    auto tile = [&](int qt, const unsigned char *cur, unsigned char *nxt, bool masked) {
        if (qt + 1 < nqt) {
            stg.load(qt + 1, lq);
            load_stats(qt + 1);
        }
        if (wave_active) {
            const float *ldsNlse = reinterpret_cast<const float *>(cur + 2 * TT * RP), *ldsNdl = ldsNlse + TT;
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                if (masked && qt * TT + qb * 32 >= lq) continue;   // a 32-query block entirely past the sequence end
                // the 16 query rows a lane holds are rows 8 g4 + 4 lh + (0..3): their -lse / -delta as the accumulators' start values, read
                // once for both key blocks
                f32x16 nl, nd;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 l4 = *reinterpret_cast<const f32x4 *>(ldsNlse + qb * 32 + 8 * g4 + 4 * lh);
                    const f32x4 d4 = *reinterpret_cast<const f32x4 *>(ldsNdl + qb * 32 + 8 * g4 + 4 * lh);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        nl[4 * g4 + i] = l4[i];
                        nd[4 * g4 + i] = d4[i];
                    }
                }
                f32x16 sc[NK], dp[NK];
#pragma unroll
                for (int j = 0; j < NK; ++j) {
                    sc[j] = nl;
                    dp[j] = nd;
                }
                mma_rows2q<NK, NS>(sc, cur, kf, dp, cur + TT * RP, vf, qb * 32, lr, lh);   // S[q][key] - lse and dP[q][key] - delta
#pragma unroll
                for (int j = 0; j < NK; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float pr = fast_exp2(sc[j][e]);
                        if (masked) {
                            const int qq = qt * TT + qb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                            pr = (qq < lq && my_k[j] < lk) ? pr : 0.f;
                        }
                        sc[j][e] = pr;                 // P
                        dp[j][e] = pr * dp[j][e];      // dS
                    }
                mma_accq<NK>(dvacc, cur + TT * RP, qb * 32, lane, sc);   // dV^T += dO^T P
                mma_accq<NK>(dkacc, cur, qb * 32, lane, dp);             // dK^T += Q^T dS
            }
        }
        if (qt + 1 < nqt) {
            stg.store(nxt);
            store_stats(nxt);
        }
        __syncthreads();
    };
    {
        const int n_fast = (k0 + KBG <= lk) ? lq / TT : 0;   // every key of the workgroup and every query of the tile valid
        int qt = 0;
        for (; qt + 1 < n_fast; qt += 2) {
            tile(qt, smem, smem + STAGE, false);
            tile(qt + 1, smem + STAGE, smem, false);
        }
        if (qt < n_fast) {
            tile(qt, smem, smem + STAGE, false);
            ++qt;
        }
        for (; qt < nqt; ++qt) tile(qt, smem + (qt & 1) * STAGE, smem + ((qt + 1) & 1) * STAGE, true);
    }
#pragma unroll
    for (int j = 0; j < NK; ++j) {
        if (my_k[j] >= lk) continue;
        T *rk = DK + (size_t)my_k[j] * a.lddk, *rv = DV + (size_t)my_k[j] * a.lddv;
        const float ksc = 0.6931471805599453f;   // dK = dS^T Q = dS^T Q' sqrt(d_h) / log2(e), times 1 / sqrt(d_h)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int d0 = 8 * g4 + 4 * lh;
            if (d0 < dh) {
                uint2 pk, pv;
                pk.x = pack_bf16(dkacc[j][4 * g4 + 0] * ksc, dkacc[j][4 * g4 + 1] * ksc);
                pk.y = pack_bf16(dkacc[j][4 * g4 + 2] * ksc, dkacc[j][4 * g4 + 3] * ksc);
                pv.x = pack_bf16(dvacc[j][4 * g4 + 0], dvacc[j][4 * g4 + 1]);
                pv.y = pack_bf16(dvacc[j][4 * g4 + 2], dvacc[j][4 * g4 + 3]);
                *reinterpret_cast<uint2 *>(rk + d0) = pk;
                *reinterpret_cast<uint2 *>(rv + d0) = pv;
            }
        }
    }
}

// the shapes the one-pass kernel takes (beyond bf16 / prescaled / no dropout / aligned operands, which the dispatch checks by its template arguments)
inline bool bwd1p_enabled() {   // ACAI_ATTN_BWD_1P=0: keep the two-kernel form (read once per process)
    static const bool on = !(getenv("ACAI_ATTN_BWD_1P") && atoi(getenv("ACAI_ATTN_BWD_1P")) == 0);
    return on;
}
inline bool bwd1p_shape_ok(int dh, int causal, int accum, int B, int H, int max_q, int max_k, int total_q, int total_k) {
    // long sequences only (short ones keep the two-kernel form: a workgroup's prologue weighs more); 32-bit byte offsets into one sequence's
    // [max_q][H][32] fp32 rows
    (void)B; (void)total_q; (void)total_k;
    return dh == 32 && !causal && !accum && max_k >= 512 && max_q >= 512 && (long long)(max_q + 64) * H * 128 < 0x7FFFFF00ll;
}

template <typename T, int DHP>
int launch_bwd(const BwdArgs &a, int B, int max_q, int max_k, bool pre, void *ws, size_t ws_bytes, int total_k, hipStream_t st) {
    constexpr int ES = sizeof(T), EPC = 16 / ES;
    const bool fast = (a.dh % EPC == 0) && (a.ldq % EPC == 0) && (a.ldk % EPC == 0) && (a.ldv % EPC == 0) && (a.lddo % EPC == 0) && (a.ldo % EPC == 0) &&
                      aligned16(a.q) && aligned16(a.k) && aligned16(a.v) && aligned16(a.dout) && aligned16(a.o) &&
                      (a.lddq % EPC == 0) && (a.lddk % EPC == 0) && (a.lddv % EPC == 0) && aligned16(a.dq) && aligned16(a.dk) && aligned16(a.dv) &&   // vector stores of the gradients too
                      (size_t)max_q * (size_t)(a.ldq > a.lddo ? a.ldq : a.lddo) * ES < 0x7FFFFF00ull && (size_t)max_k * (size_t)(a.ldk > a.ldv ? a.ldk : a.ldv) * ES < 0x7FFFFF00ull;   // the staging's 32-bit buffer resources
    constexpr int RP = TileLayout<ES, DHP>::PITCH;
    const size_t lds_dq = 2 * (2 * TT * RP), lds_dkv = 2 * (2 * TT * RP + 2 * TT * sizeof(float));   // two stages each
    if (pre && !fast) return acai_set_err(-1, "acai_attn_varlen_bwd: q_prescaled needs 16-byte aligned operands and d_h %% %d == 0", EPC);
    if (a.accum_dkv && !(ES == 2 && fast)) return acai_set_err(-1, "acai_attn_varlen_bwd: accumulating dk / dv needs bf16 and 16-byte aligned operands");
    auto launch_pair = [&](auto drop, auto fst, auto pr) {
        constexpr bool D = decltype(drop)::value, F = decltype(fst)::value, P = decltype(pr)::value;
        static bool attr[ACAI_MAX_DEV] = {};  // one latch per instantiation and device
        if (acai_first_on_device(attr)) {
            hipFuncSetAttribute(reinterpret_cast<const void *>(attn_bwd_dkv_kernel<T, DHP, F, D, P>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void *>(attn_bwd_dq_kernel<T, DHP, F, D, P>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        }
        dim3 gq(cdiv(max_q, OB), a.H, B), gk(cdiv(max_k, OB), a.H, B);
        if constexpr (sizeof(T) == 2 && DHP == 32 && F && !D && P) {
            // the training steps' form on long sequences: two lane-owned blocks per wave (ACAI_ATTN_NQ=1: the one-block kernels, A/B aid)
            static const int nq_env = getenv("ACAI_ATTN_NQ") ? atoi(getenv("ACAI_ATTN_NQ")) : 2;
            static bool attr2[ACAI_MAX_DEV] = {};
            if (acai_first_on_device(attr2)) {
                hipFuncSetAttribute(reinterpret_cast<const void *>(attn_bwd_dkv2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
                hipFuncSetAttribute(reinterpret_cast<const void *>(attn_bwd_dq2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            }
            // one pass over the scores (attn_bwd1p.hip) when the caller lent a workspace and the sequences are long (ACAI_ATTN_BWD_1P=0: keep the two kernels below, whose dQ is bit-reproducible - the one-pass form adds the key
            // blocks' contributions to a query's gradient in arrival order)
            if (bwd1p_enabled() && bwd1p_shape_ok(a.dh, a.causal, a.accum_dkv, B, a.H, max_q, max_k, a.total_q, total_k) && ws && ws_bytes >= acai_attn_bwd1p_workspace(a.total_q, a.H)) {
                // (when the host can tell that every sequence is max_k keys long, a multiple of 512, the partial-block launch is left out)
                const bool eq = total_k > 0 && (long long)B * max_k == (long long)total_k && (long long)B * max_q == (long long)a.total_q;
                acai_attn_bwd1p_launch(a, B, max_k, eq && max_k % 512 == 0 ? 0 : 1, eq ? 1 : 0, ws, st);
                return;
            }
            if (nq_env == 2 && !a.causal && !a.accum_dkv && max_q >= 512 && max_k >= 512) {
                hipLaunchKernelGGL(attn_bwd_dq2_kernel, dim3(cdiv(max_q, 2 * OB), a.H, B), dim3(256), lds_dq, st, a);
                hipLaunchKernelGGL(attn_bwd_dkv2_kernel, dim3(cdiv(max_k, 2 * OB), a.H, B), dim3(256), lds_dkv, st, a);
                return;
            }
        }
        if constexpr (sizeof(T) == 2 && DHP == 64 && F && !D && P) {
            // the training steps' d_h = 64 form: one wave per SIMD, two lane-owned blocks per wave (attn_bwd64w.hip) over every sequence's full
            // 256-row blocks; the rows past them (none for the encoder's 256-multiples, one for the decoder's 513 tokens) stay with the
            // one-block kernels below, offset by tail256.  ACAI_ATTN64_BWD_WIDE: 0 off, 1 dQ only, 2 dK/dV only, 3 both; unset = auto.
            // Measured (tools/bench_cross_train_attn.py, 16 x 16 heads, same box): the wide dQ form is 1-5 % faster than the one-block kernel
            // (encoder 4096 x 4096: 1533 against 1561 us; decoder cross 512 x 4096: 184 against 210 us).  The wide dK/dV form is 10 % faster on the
            // encoder (1886 against 2091 us; 16 x 12 heads: backward 2.51 against 2.63 ms) once its K / V fragments were pinned to the accumulator
            // half (before that it spilt to scratch and ran 2894 us), but SLOWER where the streamed query loop is short (decoder cross attention,
            // 513 queries: 9 tiles per workgroup, 663 against 607 us for the pair - one wave per SIMD has nobody to cover a workgroup's prologue):
            // auto takes it from 2048 streamed queries on.  Why the gains are small: every d_h = 64 attention kernel executes ~1 PFLOP/s of MFMA
            // work, the rate this chip sustains at its loaded clock.
            static const int wide_env = getenv("ACAI_ATTN64_BWD_WIDE") ? atoi(getenv("ACAI_ATTN64_BWD_WIDE")) : -1;
            if (wide_env && !a.causal && a.dh == 64) {
                const bool wq = (wide_env & 1) && max_q >= 256, wk = (wide_env & 2) && max_k >= 256 && !a.accum_dkv && (wide_env > 0 || max_q >= 2048);
                // when every sequence is max_q long (B * max_q rows in all) the host knows whether a tail exists; keys: only for self-attention
                const bool eq_q = (long long)B * max_q == (long long)a.total_q, eq_k = eq_q && a.cu_k == a.cu_q && max_k == max_q;
                BwdArgs t = a;
                if (wq) acai_attn_bwd64w_dq_launch(a, B, max_q, st);
                if (!wq || !eq_q || max_q % 256) {
                    t.tail256 = wq ? 1 : 0;
                    const int rows = wq ? (eq_q ? max_q % 256 : 255) : max_q;
                    hipLaunchKernelGGL((attn_bwd_dq_kernel<T, DHP, F, D, P>), dim3(cdiv(rows, OB), a.H, B), dim3(256), lds_dq, st, t);
                }
                if (wk) acai_attn_bwd64w_dkv_launch(a, B, max_k, eq_k ? 1 : 0, st);
                if (!wk || !eq_k || max_k % 256) {
                    t.tail256 = wk ? 2 : 0;
                    const int rows = wk ? (eq_k ? max_k % 256 : 255) : max_k;
                    hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, DHP, F, D, P>), dim3(cdiv(rows, OB), a.H, B), dim3(256), lds_dkv, st, t);
                }
                return;
            }
        }
        hipLaunchKernelGGL((attn_bwd_dq_kernel<T, DHP, F, D, P>), gq, dim3(256), lds_dq, st, a);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, DHP, F, D, P>), gk, dim3(256), lds_dkv, st, a);
    };
    if (pre) {
        if (a.drop_thr) launch_pair(std::true_type{}, std::true_type{}, std::true_type{});
        else launch_pair(std::false_type{}, std::true_type{}, std::true_type{});
    } else if (a.drop_thr) {
        if (fast) launch_pair(std::true_type{}, std::true_type{}, std::false_type{});
        else launch_pair(std::true_type{}, std::false_type{}, std::false_type{});
    } else {
        if (fast) launch_pair(std::false_type{}, std::true_type{}, std::false_type{});
        else launch_pair(std::false_type{}, std::false_type{}, std::false_type{});
    }
    ACAI_LAUNCH_CHECK("acai_attn_varlen_bwd");
    return 0;
}

}  // namespace

extern "C" int acai_attn_varlen_bwd_ws(const void *q, int ldq, const void *k, int ldk, const void *v, int ldv, const void *o, int ldo,
                                       const void *dout, int lddo, void *dq, int lddq, void *dk, int lddk, void *dv, int lddv, const float *lse,
                                       float *delta, const int32_t *cu_q, const int32_t *cu_k, int B, int H, int dh, int max_q, int max_k,
                                       int total_q, int total_k, int causal, int dtype, float dropout_p, uint32_t dropout_seed, int q_prescaled,
                                       void *workspace, size_t workspace_bytes, void *stream) {
    ACAI_CHECK_ARG(q && k && v && o && dout && dq && dk && dv && lse && delta && cu_q && cu_k, "acai_attn_varlen_bwd: null operand");
    ACAI_CHECK_ARG(B > 0 && H > 0 && dh > 0 && dh <= 64 && max_q > 0 && max_k > 0 && total_q > 0 && B <= 65535 && H <= 65535,
                   "acai_attn_varlen_bwd: bad dims");
    ACAI_CHECK_ARG(total_k >= 0 && (workspace || !workspace_bytes), "acai_attn_varlen_bwd_ws: bad workspace / total_k");
    BwdArgs a{};
    a.q = q; a.k = k; a.v = v; a.o = o; a.dout = dout; a.dq = dq; a.dk = dk; a.dv = dv; a.lse = lse; a.delta = delta; a.cu_q = cu_q; a.cu_k = cu_k;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.lddo = lddo; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
    ACAI_CHECK_ARG((causal & ~3) == 0, "acai_attn_varlen_bwd: causal takes bit 0 (causal mask) and bit 1 (accumulate dk / dv)");
    a.H = H; a.dh = dh; a.causal = causal & 1; a.accum_dkv = (causal >> 1) & 1; a.total_q = total_q;
    ACAI_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "acai_attn_varlen_bwd: dropout_p out of range");
    a.drop_thr = (uint32_t)((double)dropout_p * 4294967296.0); a.drop_seed = dropout_seed; a.drop_scale = 1.0f / (1.0f - dropout_p);
    a.scale = 1.0f / sqrtf((float)dh);
    a.scale_log2e = 1.4426950408889634f * a.scale;
    hipStream_t st = (hipStream_t)stream;
    const bool pre = q_prescaled != 0;
    if (dtype == ACAI_BF16)
        return dh <= 32 ? launch_bwd<bf16_t, 32>(a, B, max_q, max_k, pre, workspace, workspace_bytes, total_k, st)
                        : launch_bwd<bf16_t, 64>(a, B, max_q, max_k, pre, nullptr, 0, total_k, st);
    if (dtype == ACAI_F32)
        return dh <= 32 ? launch_bwd<float, 32>(a, B, max_q, max_k, pre, nullptr, 0, total_k, st) : launch_bwd<float, 64>(a, B, max_q, max_k, pre, nullptr, 0, total_k, st);
    return acai_set_err(-1, "acai_attn_varlen_bwd: bad dtype %d", dtype);
}

extern "C" size_t acai_attn_varlen_bwd_workspace_bytes(int B, int H, int dh, int max_q, int max_k, int total_q, int total_k, int causal, int dtype,
                                                       float dropout_p, int q_prescaled) {
    if (dtype != ACAI_BF16 || !q_prescaled || dropout_p != 0.f || B <= 0 || H <= 0 || total_q <= 0) return 0;
    if (!bwd1p_enabled()) return 0;
    return bwd1p_shape_ok(dh, causal & 1, (causal >> 1) & 1, B, H, max_q, max_k, total_q, total_k) ? acai_attn_bwd1p_workspace(total_q, H) : 0;
}

extern "C" int acai_attn_varlen_bwd(const void *q, int ldq, const void *k, int ldk, const void *v, int ldv, const void *o, int ldo,
                                    const void *dout, int lddo, void *dq, int lddq, void *dk, int lddk, void *dv, int lddv, const float *lse,
                                    float *delta, const int32_t *cu_q, const int32_t *cu_k, int B, int H, int dh, int max_q, int max_k,
                                    int total_q, int causal, int dtype, float dropout_p, uint32_t dropout_seed, int q_prescaled, void *stream) {
    return acai_attn_varlen_bwd_ws(q, ldq, k, ldk, v, ldv, o, ldo, dout, lddo, dq, lddq, dk, lddk, dv, lddv, lse, delta, cu_q, cu_k, B, H, dh, max_q, max_k,
                                   total_q, 0, causal, dtype, dropout_p, dropout_seed, q_prescaled, nullptr, 0, stream);
}
