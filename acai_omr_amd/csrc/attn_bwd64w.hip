// Varlen attention BACKWARD, bf16, d_h = 64, q prescaled, no dropout, no causal mask: ONE WAVE PER SIMD, two lane-owned 32-row blocks per wave
// (autograd of the SDPA of every encoder layer and of the teacher-forced decoder's cross attention: acai_omr/models/models.py:29-33, 351-360,
// 466-482; backward call sites acai_omr/train/omr_teacher_force_train.py:118, acai_omr/train/pre_train.py:59).
//
// Same two-pass, atomic-free decomposition and the same products / fragment layouts as attn_bwd.hip (dQ: query on the lane, streams 64-key
// tiles of K and V; dK / dV: key on the lane, streams 64-query tiles of Q and dO with their -lse / -delta), rebuilt the way attn_fwd64w.hip
// rebuilt the forward: at d_h = 64 the one-block kernels take a fresh 1 KB LDS fragment for every 32x32x16 MFMA, and two co-resident waves
// that meet at one barrier per tile run "products -> exponentials -> products" in step.  Here a wave owns 64 rows (every fragment read feeds
// two MFMAs) and the whole 512-entry register file: the gradient accumulators are asm-owned ACCUMULATOR registers (C / D of inline-asm MFMAs),
// everything the VALU touches lives in the architectural half (file built with -mllvm -amdgpu-mfma-vgpr-form).  With one wave per SIMD only
// the wave's own instruction order overlaps the VALU with the matrix pipe, so the block loop is a STAGGERED software pipeline over the two
// owned blocks j = 0, 1, pinned gap by gap with sched_barrier (one MFMA + <= 4 single-issue fillers + <= 1 LDS read per gap):
//
//   dQ, 32-key block b (24 gaps):  g0-7   S0, dP0 of block b        beside  dS1 of block b-1 (exp2, multiply, pack)
//                                  g8-11  dQ1 += K(b-1)^T dS1(b-1)
//                                  g12-19 S1, dP1 of block b        beside  dS0 of block b
//                                  g20-23 dQ0 += K(b)^T dS0(b)      beside  the start of dS1 of block b
//   dK/dV, 32-query block b (32 gaps): g0-7 S0, dP0 | g8-15 dV1, dK1 of block b-1 | g16-23 S1, dP1 | g24-31 dV0, dK0 of block b
//
// so a score tile is exponentiated while the OTHER owned block's products run, each (S, dP) register set is single-buffered, and the row
// fragments (held across the two S / dP phases) and transposed fragments (held from a block's own gradient product to the other block's one
// iteration later, two alternating register sets) are read from LDS once per block.  K / V (Q / dO) tiles travel global -> registers -> LDS one
// tile and a half ahead through a three-slot ring (tile t+2's loads are issued in tile t's first block, written to the slot tile t-1 left
// in its second block), one barrier per tile.  The accumulator start values carry the row constants as in attn_bwd.hip: S starts at -lse,
// dP at -delta, so P = 2^S' and dS = P dP' are one exp2 + one multiply per score.
// Register halves: the dK / dV kernel needs 128 (gradients) + 64 (K / V fragments) + 32 (transposed fragments) accumulator-half registers and ~210
// architectural ones; left to the allocator the K / V fragments ended up partly architectural and the kernel spilt to scratch (2894 us against the
// one-block kernel's 2142 on 16 x 16 x 4096^2) - its S / dP MFMAs are therefore inline asm with the B operand constrained to "a" (1886 us, no spill).
#include "attn_bwd_args.h"

#include <type_traits>

namespace {

typedef TileLayout<2, 64> TL;
constexpr int KT = 64, PITCH = 128, SLOT = KT * PITCH;   // one 64-row x 64-col bf16 tile: 8 KB
constexpr int NT = 256, NCH = KT * 8 / NT;               // threads per workgroup; 16-byte chunks per thread and operand tile
constexpr int RBG = 256;                                 // lane-owned rows per workgroup (4 waves x 2 blocks x 32)
typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((address_space(3))) s4 *lds_s4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define ACAI_SB() __builtin_amdgcn_sched_barrier(0)

// acc (32 x 32 fp32, accumulator registers) += A . B on the matrix pipe (see attn_fwd64w.hip).  The A operand - a transposed LDS fragment that
// only this MFMA reads - is asked for in the accumulator half too (ds_read writes it there directly): the architectural half is what runs out.
#ifndef ACAI_BWD64W_AV
#define ACAI_BWD64W_AV 1
#endif
// A VALU write (the compiler's v_accvgpr_mov / v_accvgpr_write copies that assemble an operand tuple sit directly in front of the asm statement)
// needs two wait states before an MFMA reads the register; hipcc inserts them for its own MFMAs and knows nothing about an asm one: without
// the s_nop the gradients came out wrong in a few 32 x 32 blocks (tools/dbg_bwd64w.py: "s_nop 1" in front is exact, any s_nop behind is not).
#ifndef ACAI_BWD64W_PRE
#define ACAI_BWD64W_PRE 1
#endif
#ifndef ACAI_BWD64W_POST
#define ACAI_BWD64W_POST 0
#endif
#define ACAI_STR2(x) #x
#define ACAI_STR(x) ACAI_STR2(x)
#if ACAI_BWD64W_PRE
#define ACAI_PRE_NOP "s_nop " ACAI_STR(ACAI_BWD64W_PRE) "\n\t"
#else
#define ACAI_PRE_NOP ""
#endif
#if ACAI_BWD64W_POST
#define ACAI_POST_NOP "\n\ts_nop " ACAI_STR(ACAI_BWD64W_POST)
#else
#define ACAI_POST_NOP ""
#endif
__device__ __forceinline__ void mma_acc(f32x16 &c, const uint4 &a, const uint4 &b) {
    const u32x4 av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
    if constexpr (ACAI_BWD64W_AV) asm volatile(ACAI_PRE_NOP "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" ACAI_POST_NOP : "+a"(c) : "a"(av), "v"(bv));
    else asm volatile(ACAI_PRE_NOP "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" ACAI_POST_NOP : "+a"(c) : "v"(av), "v"(bv));
}
__device__ __forceinline__ f32x16 mma(const uint4 &af, const uint4 &bf, const f32x16 &c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bf), c, 0, 0, 0);
}
// S / dP chains of the dK / dV kernel: D and C architectural (the VALU reads them two MFMA gaps later at the earliest), the B operand - a K / V fragment of
// an owned block, live for the whole kernel - PINNED to the accumulator half: left to the allocator, part of those 64 registers stayed architectural
// and the kernel spilt to scratch (ACAI_BWD64W_DKV_ASM=0: the builtin form, A/B aid).
#ifndef ACAI_BWD64W_DKV_ASM
#define ACAI_BWD64W_DKV_ASM 1
#endif
__device__ __forceinline__ void mma_ab0(f32x16 &d, const uint4 &a, const uint4 &b, const f32x16 &c) {   // d = A . B + c
    if constexpr (ACAI_BWD64W_DKV_ASM) {
        const u32x4 av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(av), "a"(bv), "v"(c));   // (no s_nop: operands come from LDS reads, AGPR constants and MFMAs; _asmcheck verifies)
    } else
        d = mma(a, b, c);
}
__device__ __forceinline__ void mma_ab(f32x16 &d, const uint4 &a, const uint4 &b) {   // d += A . B
    if constexpr (ACAI_BWD64W_DKV_ASM) {
        const u32x4 av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(av), "a"(bv));
    } else
        d = mma(a, b, d);
}

// ---- the VALU stream of one (owned block, streamed block) pair, as numbered single-issue operations -------------------------------------------
// dQ form (40 operations): e(i): s[i] = 2^s[i] (MASK: 0 for streamed rows past the end); m(i): s[i] *= p[i]; k(n): x[n] = bf16 pair (s[2n], s[2n+1]).
// Skewed so that no multiply directly follows its exponential:  e0 e1 | e2 e3 m0 m1 k0 | e4 e5 m2 m3 k1 | ... | e14 e15 m12 m13 k6 | m14 m15 k7
template <int OP, bool MASK>
__device__ __forceinline__ void dq_op(f32x16 &s, const f32x16 &p, uint32_t (&x)[8], int row0, int lh, int rows) {
    constexpr int r = OP - 2, step = OP < 2 ? 0 : 1 + r / 5, w = OP < 2 ? OP : r % 5;
    if constexpr (step == 0 || (step < 8 && w < 2)) {
        constexpr int i = 2 * step + w;
        float v = fast_exp2(s[i]);
        if constexpr (MASK) v = (row0 + (i & 3) + 8 * (i >> 2) + 4 * lh) < rows ? v : 0.f;
        s[i] = v;
    } else if constexpr (step < 8 ? w < 4 : w < 2) {
        constexpr int i = 2 * (step - 1) + (step < 8 ? w - 2 : w);
        s[i] *= p[i];
    } else {
        constexpr int n = step - 1;
        x[n] = pack_bf16(s[2 * n], s[2 * n + 1]);
    }
}
template <int OP, int END, bool MASK>
__device__ __forceinline__ void dq_ops(f32x16 &s, const f32x16 &p, uint32_t (&x)[8], int row0, int lh, int rows) {
    if constexpr (OP < END) {
        dq_op<OP, MASK>(s, p, x, row0, lh, rows);
        dq_ops<OP + 1, END, MASK>(s, p, x, row0, lh, rows);
    }
}

// dK / dV form (48 operations): e(i): s[i] = P = 2^s[i] (MASK: 0 unless the streamed query row and the lane's key both exist); m(i): p[i] = dS = s[i] p[i];
// k(n): xp[n] = bf16 pair of P, xs[n] = bf16 pair of dS.   e0 e1 | e2 e3 m0 m1 kP0 kS0 | ... | e14 e15 m12 m13 kP6 kS6 | m14 m15 kP7 kS7
template <int OP, bool MASK>
__device__ __forceinline__ void dkv_op(f32x16 &s, f32x16 &p, uint32_t (&xp)[8], uint32_t (&xs)[8], int row0, int lh, int rows, bool lane_ok) {
    constexpr int r = OP - 2, step = OP < 2 ? 0 : 1 + r / 6, w = OP < 2 ? OP : r % 6;
    if constexpr (step == 0 || (step < 8 && w < 2)) {
        constexpr int i = 2 * step + w;
        float v = fast_exp2(s[i]);
        if constexpr (MASK) v = (lane_ok && (row0 + (i & 3) + 8 * (i >> 2) + 4 * lh) < rows) ? v : 0.f;
        s[i] = v;
    } else if constexpr (step < 8 ? w < 4 : w < 2) {
        constexpr int i = 2 * (step - 1) + (step < 8 ? w - 2 : w);
        p[i] *= s[i];
    } else if constexpr (step < 8 ? w == 4 : w == 2) {
        constexpr int n = step - 1;
        xp[n] = pack_bf16(s[2 * n], s[2 * n + 1]);
    } else {
        constexpr int n = step - 1;
        xs[n] = pack_bf16(p[2 * n], p[2 * n + 1]);
    }
}
template <int OP, int END, bool MASK>
__device__ __forceinline__ void dkv_ops(f32x16 &s, f32x16 &p, uint32_t (&xp)[8], uint32_t (&xs)[8], int row0, int lh, int rows, bool lane_ok) {
    if constexpr (OP < END) {
        dkv_op<OP, MASK>(s, p, xp, xs, row0, lh, rows, lane_ok);
        dkv_ops<OP + 1, END, MASK>(s, p, xp, xs, row0, lh, rows, lane_ok);
    }
}

__device__ __forceinline__ uint4 x4(const uint32_t (&x)[8], int h) { return make_uint4(x[4 * h], x[4 * h + 1], x[4 * h + 2], x[4 * h + 3]); }

typedef std::true_type Y;
typedef std::false_type N;

// XCD-aware block order of a one-dimensional grid (see attn_fwd64w.hip): the blocks of one (sequence, head) stream the same operand tiles
__device__ __forceinline__ int xcd_vid(bool xcd) {   // (xcd = false: a ragged batch keeps the plain order, see attn_bwd1p.hip)
    int vid = blockIdx.x;
    const int per = gridDim.x >> 3;
    if (xcd && vid < (per << 3)) vid = (vid & 7) * per + (vid >> 3);
    return vid;
}

// Staging of two [rows][64] bf16 operands, 64-row tiles, through registers (buffer loads: rows past `rows` read as zeros)
struct Stager2 {
    const bf16_t *A, *B;
    int lda, ldb, rows;
    int soff[NCH];
    uint32_t ga[NCH], gb[NCH];
    u32x4 ra[NCH], rb[NCH];
    __device__ __forceinline__ void init(const bf16_t *A_, int lda_, const bf16_t *B_, int ldb_, int rows_, int tid) {
        A = A_; B = B_; lda = lda_; ldb = ldb_; rows = rows_;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int cidx = tid + NT * i, row = cidx >> 3, cc = cidx & 7;
            soff[i] = TL::off(row, cc);
            ga[i] = (uint32_t)(row * lda * 2 + cc * 16);
            gb[i] = (uint32_t)(row * ldb * 2 + cc * 16);
        }
    }
    __device__ __forceinline__ void load(int t) {
        const int left = rows - t * KT;
        const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(A + (size_t)t * KT * lda), 0, left > 0 ? left * lda * 2 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(B + (size_t)t * KT * ldb), 0, left > 0 ? left * ldb * 2 : 0, 0x00020000);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(r0, ga[i], 0, 0);
            rb[i] = __builtin_amdgcn_raw_buffer_load_b128(r1, gb[i], 0, 0);
        }
    }
    __device__ __forceinline__ void store(unsigned char *sa, unsigned char *sb) const {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            *reinterpret_cast<u32x4 *>(sa + soff[i]) = ra[i];
            *reinterpret_cast<u32x4 *>(sb + soff[i]) = rb[i];
        }
    }
};

// =====================================================================================================================================
// dQ: workgroup = 256 queries of one (sequence, head); wave = 64 queries (two lane-owned blocks); streams K and V
// =====================================================================================================================================
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_bwd64w_dq_kernel(BwdArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[6 * SLOT];   // K ring: slots 0..2; V ring: slots 3..5
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int vid = xcd_vid(a.nblk > 0), nblk = a.nblk < 0 ? -a.nblk : a.nblk;
    const int qb = vid % nblk, h = (vid / nblk) % a.H, b = vid / (nblk * a.H);
    const int q_start = a.cu_q[b], lq = a.cu_q[b + 1] - q_start;
    const int k_start = a.cu_k[b], lk = a.cu_k[b + 1] - k_start;
    const int q0 = qb * RBG;
    if (q0 + RBG > lq) return;   // full blocks only: the rows past a sequence's last full block belong to attn_bwd_dq_kernel (tail256)

    const bf16_t *Q = reinterpret_cast<const bf16_t *>(a.q) + (size_t)q_start * a.ldq + h * 64;
    const bf16_t *K = reinterpret_cast<const bf16_t *>(a.k) + (size_t)k_start * a.ldk + h * 64;
    const bf16_t *V = reinterpret_cast<const bf16_t *>(a.v) + (size_t)k_start * a.ldv + h * 64;
    const bf16_t *DO = reinterpret_cast<const bf16_t *>(a.dout) + (size_t)q_start * a.lddo + h * 64;
    const bf16_t *O = reinterpret_cast<const bf16_t *>(a.o) + (size_t)q_start * a.ldo + h * 64;
    bf16_t *DQ = reinterpret_cast<bf16_t *>(a.dq) + (size_t)q_start * a.lddq + h * 64;
    const int nkt = (lk + KT - 1) / KT;

    // ---- lane-owned rows: Q and dO fragments (B operands), -lse and -delta as accumulator start values ---------------------------------------
    int my_q[2];
    uint4 qf[2][4], dof[2][4];
    f32x16 sinit[2], pinit[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        my_q[j] = q0 + wave * 64 + j * 32 + lr;   // (< lq: the block is full)
        float dlt = 0.f;   // delta[q] = sum_d dO[q,d] O[q,d], published (negated) for the dK / dV kernels - see attn_bwd_dq_kernel
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qf[j][s] = *reinterpret_cast<const uint4 *>(Q + (size_t)my_q[j] * a.ldq + s * 16 + lh * 8);
            dof[j][s] = *reinterpret_cast<const uint4 *>(DO + (size_t)my_q[j] * a.lddo + s * 16 + lh * 8);
            const uint4 of = *reinterpret_cast<const uint4 *>(O + (size_t)my_q[j] * a.ldo + s * 16 + lh * 8);
            const uint32_t ow[4] = {of.x, of.y, of.z, of.w}, dw[4] = {dof[j][s].x, dof[j][s].y, dof[j][s].z, dof[j][s].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dlt = fmaf(__uint_as_float(ow[e] << 16), __uint_as_float(dw[e] << 16), dlt);
                dlt = fmaf(__uint_as_float(ow[e] & 0xffff0000u), __uint_as_float(dw[e] & 0xffff0000u), dlt);
            }
        }
        dlt += __shfl_xor(dlt, 32);
        const size_t sidx = (size_t)h * a.total_q + q_start + my_q[j];
        const float lse = a.lse[sidx];
        if (lh == 0) const_cast<float *>(a.delta)[sidx] = -dlt;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            sinit[j][e] = -lse;
            pinit[j][e] = -dlt;
        }
    }

    Stager2 stg;
    stg.init(K, a.ldk, V, a.ldv, lk, tid);

    // ---- fragment addresses inside a tile ------------------------------------------------------------------------------------------------------
    int raddr[4];   // row fragments: row lr of the 32-row block, 16-byte chunk 2 s + lh
#pragma unroll
    for (int s = 0; s < 4; ++s) raddr[s] = TL::off(lr, 2 * s + lh);
    // transposed fragment (k-step s2, d block d): two 4-row x 16-d transposing reads, rows L and L + 8 with L = 4 lh + (i16 >> 2) (+ 16 s2)
    const int i16 = lane & 15, g1 = (lane >> 4) & 1;
    int taddr[2][2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) taddr[d][r2] = TL::off(4 * lh + (i16 >> 2) + 8 * r2, d * 4 + 2 * g1 + ((i16 & 3) >> 1)) + 8 * (i16 & 1);
    auto read_r = [&](const unsigned char *blk, int s) -> uint4 { return *reinterpret_cast<const uint4 *>(blk + raddr[s]); };
    auto read_t_half = [&](const unsigned char *blk, int i, int r2) -> s4 {   // half r2 of transposed fragment i = (d block i & 1, k-step i >> 1)
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(blk + (i >> 1) * 16 * PITCH + taddr[i & 1][r2]));
    };
    union TF { s4 v[2]; uint4 u; };

    f32x16 dq[2][2];   // [owned block][d block], accumulator registers
    f32x16 sc[2], dp[2];
    uint32_t xf[2][8];
    uint4 kr[4], vr[4];
    TF ktA[4], ktB[4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) dq[j][d][e] = 0.f;

    // One block iteration (see the file header).  kblk / vblk: LDS address of the 32-key block b inside its tile (for the transposed reads of K);
    // knext / vnext: block b + 1 (row fragment reloads).  ktC receives K(b)^T, ktP holds K(b-1)^T.
    //   FIRST: no block b-1 (no dS1 / dQ1 of it);  BODY: block b exists (false: the drain call, only dS1 / dQ1 of block b-1)
    //   MP / MC: mask the probabilities of block b-1 / b (keys past the end);  STAGE 1: issue tile (t+2)'s loads, 2: write them to k_dst / v_dst
    auto block_iter = [&](auto first_, auto body_, auto mp_, auto mc_, auto stage_, int key0, const unsigned char *kblk, const unsigned char *knext,
                          const unsigned char *vnext, TF (&ktC)[4], TF (&ktP)[4], int tnext = 0, unsigned char *k_dst = nullptr, unsigned char *v_dst = nullptr) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_)::value, BODY = decltype(body_)::value, MP = decltype(mp_)::value, MC = decltype(mc_)::value;
        constexpr int STAGE = decltype(stage_)::value;
        // g0-7: S0, dP0 of block b | dS1 of block b-1: operations 12..39 | K(b)^T fragment halves
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if constexpr (BODY) sc[0] = mma(kr[s], qf[0][s], s == 0 ? sinit[0] : sc[0]);
            if constexpr (!FIRST) {
                if (s == 0) dq_ops<12, 16, MP>(sc[1], dp[1], xf[1], key0 - 32, lh, lk);
                if (s == 1) dq_ops<20, 24, MP>(sc[1], dp[1], xf[1], key0 - 32, lh, lk);
                if (s == 2) dq_ops<28, 31, MP>(sc[1], dp[1], xf[1], key0 - 32, lh, lk);
                if (s == 3) dq_ops<34, 37, MP>(sc[1], dp[1], xf[1], key0 - 32, lh, lk);
            }
            if constexpr (BODY) {   // (both halves in one statement: read a gap apart, the 4-register tuple was assembled through copies)
                ktC[s].v[0] = read_t_half(kblk, s, 0);
                ktC[s].v[1] = read_t_half(kblk, s, 1);
            }
            ACAI_SB();
            if constexpr (BODY) dp[0] = mma(vr[s], dof[0][s], s == 0 ? pinit[0] : dp[0]);
            if constexpr (!FIRST) {
                if (s == 0) dq_ops<16, 20, MP>(sc[1], dp[1], xf[1], key0 - 32, lh, lk);
                if (s == 1) dq_ops<24, 28, MP>(sc[1], dp[1], xf[1], key0 - 32, lh, lk);
                if (s == 2) dq_ops<31, 34, MP>(sc[1], dp[1], xf[1], key0 - 32, lh, lk);
                if (s == 3) dq_ops<37, 40, MP>(sc[1], dp[1], xf[1], key0 - 32, lh, lk);
            }
            ACAI_SB();
        }
        // g8-11: dQ1 += K(b-1)^T dS1(b-1) | dS0 of block b: operations 0..11 from g9
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (!FIRST) mma_acc(dq[1][i & 1], ktP[i].u, x4(xf[1], i >> 1));
            if constexpr (BODY) {
                if constexpr (STAGE == 1) {
                    if (i == 0) stg.load(tnext);
                }
                if (i == 1) dq_ops<0, 4, MC>(sc[0], dp[0], xf[0], key0, lh, lk);
                if (i == 2) dq_ops<4, 8, MC>(sc[0], dp[0], xf[0], key0, lh, lk);
                if (i == 3) dq_ops<8, 12, MC>(sc[0], dp[0], xf[0], key0, lh, lk);
            }
            ACAI_SB();
        }
        if constexpr (BODY) {
            // g12-19: S1, dP1 of block b | dS0 of block b: operations 12..39 | row fragments of block b + 1 behind their last use
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                sc[1] = mma(kr[s], qf[1][s], s == 0 ? sinit[1] : sc[1]);
                if (s == 0) dq_ops<12, 16, MC>(sc[0], dp[0], xf[0], key0, lh, lk);
                if (s == 1) dq_ops<20, 24, MC>(sc[0], dp[0], xf[0], key0, lh, lk);
                if (s == 2) dq_ops<28, 31, MC>(sc[0], dp[0], xf[0], key0, lh, lk);
                if (s == 3) dq_ops<34, 37, MC>(sc[0], dp[0], xf[0], key0, lh, lk);
                kr[s] = read_r(knext, s);
                ACAI_SB();
                dp[1] = mma(vr[s], dof[1][s], s == 0 ? pinit[1] : dp[1]);
                if (s == 0) dq_ops<16, 20, MC>(sc[0], dp[0], xf[0], key0, lh, lk);
                if (s == 1) dq_ops<24, 28, MC>(sc[0], dp[0], xf[0], key0, lh, lk);
                if (s == 2) dq_ops<31, 34, MC>(sc[0], dp[0], xf[0], key0, lh, lk);
                if (s == 3) dq_ops<37, 40, MC>(sc[0], dp[0], xf[0], key0, lh, lk);
                vr[s] = read_r(vnext, s);
                ACAI_SB();
            }
            // g20-23: dQ0 += K(b)^T dS0(b) | dS1 of block b: operations 0..11 from g21
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                mma_acc(dq[0][i & 1], ktC[i].u, x4(xf[0], i >> 1));
                if constexpr (STAGE == 2) {
                    if (i == 0) stg.store(k_dst, v_dst);
                }
                if (i == 1) dq_ops<0, 4, MC>(sc[1], dp[1], xf[1], key0, lh, lk);
                if (i == 2) dq_ops<4, 8, MC>(sc[1], dp[1], xf[1], key0, lh, lk);
                if (i == 3) dq_ops<8, 12, MC>(sc[1], dp[1], xf[1], key0, lh, lk);
                ACAI_SB();
            }
        }
    };
    auto kslot = [&](int t) -> unsigned char * { return lds + (t % 3) * SLOT; };
    auto vslot = [&](int t) -> unsigned char * { return lds + (3 + t % 3) * SLOT; };
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    typedef std::integral_constant<int, 2> S2;

    if (nkt > 0) {
        // ---- prologue: tiles 0 and 1 into the rings, the row fragments of block 0 ---------------------------------------------------------------
        stg.load(0);
        stg.store(kslot(0), vslot(0));
        stg.load(1);
        stg.store(kslot(1), vslot(1));
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kr[s] = read_r(kslot(0), s);
            vr[s] = read_r(vslot(0), s);
        }
        // tile t: block 2t (ktA <- K^T, dQ1 of block 2t-1 from ktB) issues the loads of tile t+2; block 2t+1 (ktB <- K^T, dQ1 from ktA) writes them
        // to the slots tile t-1 left; one barrier per tile
        auto tile = [&](auto first_, auto mask_, int t, const unsigned char *k0, const unsigned char *v0, const unsigned char *k1, const unsigned char *v1,
                        unsigned char *k_dst, unsigned char *v_dst) __attribute__((always_inline)) {
            constexpr bool M = decltype(mask_)::value;
            block_iter(first_, Y{}, std::integral_constant<bool, M>{}, std::integral_constant<bool, M>{}, S1{}, t * KT, k0, k0 + 32 * PITCH, v0 + 32 * PITCH, ktA, ktB, t + 2);
            block_iter(N{}, Y{}, std::integral_constant<bool, M>{}, std::integral_constant<bool, M>{}, S2{}, t * KT + 32, k0 + 32 * PITCH, k1, v1, ktB, ktA, 0, k_dst, v_dst);
            __syncthreads();
        };
        int t = 0;
        if (nkt > 4) {   // tile 0 (no block -1), then rounds of three tiles with static ring slots and no masks while three more tiles follow
            tile(Y{}, N{}, 0, lds + 0 * SLOT, lds + 3 * SLOT, lds + 1 * SLOT, lds + 4 * SLOT, lds + 2 * SLOT, lds + 5 * SLOT);
            tile(N{}, N{}, 1, lds + 1 * SLOT, lds + 4 * SLOT, lds + 2 * SLOT, lds + 5 * SLOT, lds + 0 * SLOT, lds + 3 * SLOT);
            tile(N{}, N{}, 2, lds + 2 * SLOT, lds + 5 * SLOT, lds + 0 * SLOT, lds + 3 * SLOT, lds + 1 * SLOT, lds + 4 * SLOT);
            for (t = 3; t + 3 < nkt; t += 3) {
                tile(N{}, N{}, t, lds + 0 * SLOT, lds + 3 * SLOT, lds + 1 * SLOT, lds + 4 * SLOT, lds + 2 * SLOT, lds + 5 * SLOT);
                tile(N{}, N{}, t + 1, lds + 1 * SLOT, lds + 4 * SLOT, lds + 2 * SLOT, lds + 5 * SLOT, lds + 0 * SLOT, lds + 3 * SLOT);
                tile(N{}, N{}, t + 2, lds + 2 * SLOT, lds + 5 * SLOT, lds + 0 * SLOT, lds + 3 * SLOT, lds + 1 * SLOT, lds + 4 * SLOT);
            }
        } else {
            tile(Y{}, Y{}, 0, kslot(0), vslot(0), kslot(1), vslot(1), kslot(2), vslot(2));
            t = 1;
        }
        for (; t < nkt; ++t)   // the remaining tiles (the ragged last one among them): masked probabilities, ring slots from t
            tile(N{}, Y{}, t, kslot(t), vslot(t), kslot(t + 1), vslot(t + 1), kslot(t + 2), vslot(t + 2));
        // ---- drain: dS1 and dQ1 of the last block ---------------------------------------------------------------------------------------------------
        block_iter(N{}, N{}, Y{}, Y{}, S0{}, nkt * KT, lds, lds, lds, ktA, ktB);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last accumulator MFMAs are asm statements: no compiler-tracked hazard in front of the reads below

#pragma unroll
    for (int j = 0; j < 2; ++j) {
        bf16_t *row = DQ + (size_t)my_q[j] * a.lddq;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                uint2 pk;   // registers 4 g4 .. 4 g4 + 3 are four consecutive d: one 8-byte store
                pk.x = pack_bf16(dq[j][d][4 * g4 + 0] * a.scale, dq[j][d][4 * g4 + 1] * a.scale);
                pk.y = pack_bf16(dq[j][d][4 * g4 + 2] * a.scale, dq[j][d][4 * g4 + 3] * a.scale);
                *reinterpret_cast<uint2 *>(row + d * 32 + 8 * g4 + 4 * lh) = pk;
            }
    }
}


// =====================================================================================================================================
// dK / dV: workgroup = 256 keys of one (sequence, head); wave = 64 keys (two lane-owned blocks); streams Q, dO and their -lse / -delta
// =====================================================================================================================================
constexpr int QSLOT = 2 * SLOT + 2 * KT * (int)sizeof(float);   // one ring slot: Q tile, dO tile, -lse[64], -delta[64]

__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_bwd64w_dkv_kernel(BwdArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * QSLOT];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int vid = xcd_vid(a.nblk > 0), nblk = a.nblk < 0 ? -a.nblk : a.nblk;
    const int kb = vid % nblk, h = (vid / nblk) % a.H, b = vid / (nblk * a.H);
    const int q_start = a.cu_q[b], lq = a.cu_q[b + 1] - q_start;
    const int k_start = a.cu_k[b], lk = a.cu_k[b + 1] - k_start;
    const int k0 = kb * RBG;
    if (k0 + RBG > lk) return;   // full blocks only: the keys past a sequence's last full block belong to attn_bwd_dkv_kernel (tail256)

    const bf16_t *Q = reinterpret_cast<const bf16_t *>(a.q) + (size_t)q_start * a.ldq + h * 64;
    const bf16_t *K = reinterpret_cast<const bf16_t *>(a.k) + (size_t)k_start * a.ldk + h * 64;
    const bf16_t *V = reinterpret_cast<const bf16_t *>(a.v) + (size_t)k_start * a.ldv + h * 64;
    const bf16_t *DO = reinterpret_cast<const bf16_t *>(a.dout) + (size_t)q_start * a.lddo + h * 64;
    bf16_t *DK = reinterpret_cast<bf16_t *>(a.dk) + (size_t)k_start * a.lddk + h * 64;
    bf16_t *DV = reinterpret_cast<bf16_t *>(a.dv) + (size_t)k_start * a.lddv + h * 64;
    const int nqt = (lq + KT - 1) / KT;

    int my_k[2];
    uint4 kf[2][4], vf[2][4];   // B operands of S = Q K^T and dP = dO V^T
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        my_k[j] = k0 + wave * 64 + j * 32 + lr;   // (< lk: the block is full)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kf[j][s] = *reinterpret_cast<const uint4 *>(K + (size_t)my_k[j] * a.ldk + s * 16 + lh * 8);
            vf[j][s] = *reinterpret_cast<const uint4 *>(V + (size_t)my_k[j] * a.ldv + s * 16 + lh * 8);
        }
    }

    Stager2 stg;
    stg.init(Q, a.ldq, DO, a.lddo, lq, tid);
    // one query row's statistics per thread: threads 0..63 (and, redundantly, 128..191) fetch lse, 64..127 (192..255) -delta (every thread issues
    // the load: see attn_bwd_dkv_kernel); both go to LDS negated - they are accumulator start values
    const float *stat_row = ((tid & 64) ? a.delta : a.lse) + (size_t)h * a.total_q + q_start;
    float r_stat = 0.f;
    auto load_stats = [&](int t) {
        const int qq = t * KT + (tid & 63);
        r_stat = stat_row[qq < lq ? qq : 0];
    };
    auto store_stats = [&](unsigned char *slot) {
        if (tid < 2 * KT) reinterpret_cast<float *>(slot + 2 * SLOT)[tid] = (tid & 64) ? r_stat : -r_stat;
    };

    int raddr[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) raddr[s] = TL::off(lr, 2 * s + lh);
    const int i16 = lane & 15, g1 = (lane >> 4) & 1;
    int taddr[2][2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) taddr[d][r2] = TL::off(4 * lh + (i16 >> 2) + 8 * r2, d * 4 + 2 * g1 + ((i16 & 3) >> 1)) + 8 * (i16 & 1);
    auto read_r = [&](const unsigned char *blk, int s) -> uint4 { return *reinterpret_cast<const uint4 *>(blk + raddr[s]); };
    union TF { s4 v[2]; uint4 u; };
    auto read_t = [&](const unsigned char *blk, int i) -> TF {   // transposed fragment i = (d block i & 1, k-step i >> 1) of the 32-row block at blk
        TF f;
        f.v[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(blk + (i >> 1) * 16 * PITCH + taddr[i & 1][0]));
        f.v[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(blk + (i >> 1) * 16 * PITCH + taddr[i & 1][1]));
        return f;
    };
    // the 16 query rows a lane's accumulator registers hold are rows 8 g4 + 4 lh + (0..3) of the block: group g4 of their start values
    // (constant register indices on every path: a run-time vector index sends the whole tile to scratch memory)
    auto read_init = [&](f32x16 &acc, const float *stat, int g4) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(stat + 8 * g4 + 4 * lh);
        switch (g4) {
            case 0: acc[0] = v[0]; acc[1] = v[1]; acc[2] = v[2]; acc[3] = v[3]; break;
            case 1: acc[4] = v[0]; acc[5] = v[1]; acc[6] = v[2]; acc[7] = v[3]; break;
            case 2: acc[8] = v[0]; acc[9] = v[1]; acc[10] = v[2]; acc[11] = v[3]; break;
            default: acc[12] = v[0]; acc[13] = v[1]; acc[14] = v[2]; acc[15] = v[3]; break;
        }
    };

    f32x16 dk[2][2], dv[2][2];   // [owned block][d block], accumulator registers
    f32x16 sc[2], dp[2];
    f32x16 nl, nd;               // -lse / -delta of the streamed block's 16 rows this lane's accumulator registers hold: start values of both owned blocks' chains
    uint32_t xp[2][8], xs[2][8];
    uint4 qr[4], dor[4];
    TF qt[4], dot[4];            // Q^T / dO^T fragments: read once per block, behind the LAST product of the block before (accumulator half: see mma_acc)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) dk[j][d][e] = dv[j][d][e] = 0.f;

    // One block iteration (see the file header).  blk: LDS address of the 32-query block b inside its slot's Q tile (dO tile: + SLOT);
    // next / nstat: block b + 1 (row fragments, start values).
    //   FIRST: no block b-1;  BODY: block b exists (false: the drain call);  MP / MC: mask the probabilities of block b-1 / b (queries past the end)
    //   STAGE 1: issue tile (t+2)'s loads, 2: write them to dst
    auto block_iter = [&](auto first_, auto body_, auto mp_, auto mc_, auto stage_, int row0, const unsigned char *blk, const unsigned char *next, const float *nstat,
                          int tnext = 0, unsigned char *dst = nullptr) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_)::value, BODY = decltype(body_)::value, MP = decltype(mp_)::value, MC = decltype(mc_)::value;
        constexpr int STAGE = decltype(stage_)::value;
        // g0-7: S0, dP0 of block b | P1, dS1 of block b-1: operations 24..47
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if constexpr (BODY) {
                if (s == 0) mma_ab0(sc[0], qr[s], kf[0][s], nl);
                else mma_ab(sc[0], qr[s], kf[0][s]);
            }
            if constexpr (!FIRST) {
                if (s == 0) dkv_ops<24, 27, MP>(sc[1], dp[1], xp[1], xs[1], row0 - 32, lh, lq, true);
                if (s == 1) dkv_ops<30, 33, MP>(sc[1], dp[1], xp[1], xs[1], row0 - 32, lh, lq, true);
                if (s == 2) dkv_ops<36, 39, MP>(sc[1], dp[1], xp[1], xs[1], row0 - 32, lh, lq, true);
                if (s == 3) dkv_ops<42, 45, MP>(sc[1], dp[1], xp[1], xs[1], row0 - 32, lh, lq, true);
            }
            ACAI_SB();
            if constexpr (BODY) {
                if (s == 0) mma_ab0(dp[0], dor[s], vf[0][s], nd);
                else mma_ab(dp[0], dor[s], vf[0][s]);
            }
            if constexpr (!FIRST) {
                if (s == 0) dkv_ops<27, 30, MP>(sc[1], dp[1], xp[1], xs[1], row0 - 32, lh, lq, true);
                if (s == 1) dkv_ops<33, 36, MP>(sc[1], dp[1], xp[1], xs[1], row0 - 32, lh, lq, true);
                if (s == 2) dkv_ops<39, 42, MP>(sc[1], dp[1], xp[1], xs[1], row0 - 32, lh, lq, true);
                if (s == 3) dkv_ops<45, 48, MP>(sc[1], dp[1], xp[1], xs[1], row0 - 32, lh, lq, true);
            }
            ACAI_SB();
        }
        // g8-15: dV1 += dO(b-1)^T P1, dK1 += Q(b-1)^T dS1, each fragment then replaced by block b's | P0, dS0 of block b: operations 0..23 from g9
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (!FIRST) mma_acc(dv[1][i & 1], dot[i].u, x4(xp[1], i >> 1));
            if constexpr (BODY) {
                if constexpr (STAGE == 1) {
                    if (i == 0) {
                        stg.load(tnext);
                        load_stats(tnext);
                    }
                }
                if (i == 1) dkv_ops<4, 8, MC>(sc[0], dp[0], xp[0], xs[0], row0, lh, lq, true);
                if (i == 2) dkv_ops<12, 15, MC>(sc[0], dp[0], xp[0], xs[0], row0, lh, lq, true);
                if (i == 3) dkv_ops<18, 21, MC>(sc[0], dp[0], xp[0], xs[0], row0, lh, lq, true);
                dot[i] = read_t(blk + SLOT, i);
            }
            ACAI_SB();
            if constexpr (!FIRST) mma_acc(dk[1][i & 1], qt[i].u, x4(xs[1], i >> 1));
            if constexpr (BODY) {
                if (i == 0) dkv_ops<0, 4, MC>(sc[0], dp[0], xp[0], xs[0], row0, lh, lq, true);
                if (i == 1) dkv_ops<8, 12, MC>(sc[0], dp[0], xp[0], xs[0], row0, lh, lq, true);
                if (i == 2) dkv_ops<15, 18, MC>(sc[0], dp[0], xp[0], xs[0], row0, lh, lq, true);
                if (i == 3) dkv_ops<21, 24, MC>(sc[0], dp[0], xp[0], xs[0], row0, lh, lq, true);
                qt[i] = read_t(blk, i);
            }
            ACAI_SB();
        }
        if constexpr (BODY) {
            // g16-23: S1, dP1 of block b | P0, dS0 of block b: operations 24..47 | row fragments of block b + 1 behind their last use
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (s == 0) mma_ab0(sc[1], qr[s], kf[1][s], nl);
                else mma_ab(sc[1], qr[s], kf[1][s]);
                if (s == 0) dkv_ops<24, 27, MC>(sc[0], dp[0], xp[0], xs[0], row0, lh, lq, true);
                if (s == 1) dkv_ops<30, 33, MC>(sc[0], dp[0], xp[0], xs[0], row0, lh, lq, true);
                if (s == 2) dkv_ops<36, 39, MC>(sc[0], dp[0], xp[0], xs[0], row0, lh, lq, true);
                if (s == 3) dkv_ops<42, 45, MC>(sc[0], dp[0], xp[0], xs[0], row0, lh, lq, true);
                qr[s] = read_r(next, s);
                ACAI_SB();
                if (s == 0) mma_ab0(dp[1], dor[s], vf[1][s], nd);
                else mma_ab(dp[1], dor[s], vf[1][s]);
                if (s == 0) dkv_ops<27, 30, MC>(sc[0], dp[0], xp[0], xs[0], row0, lh, lq, true);
                if (s == 1) dkv_ops<33, 36, MC>(sc[0], dp[0], xp[0], xs[0], row0, lh, lq, true);
                if (s == 2) dkv_ops<39, 42, MC>(sc[0], dp[0], xp[0], xs[0], row0, lh, lq, true);
                if (s == 3) dkv_ops<45, 48, MC>(sc[0], dp[0], xp[0], xs[0], row0, lh, lq, true);
                dor[s] = read_r(next + SLOT, s);
                ACAI_SB();
            }
            // g24-31: dV0 += dO(b)^T P0, dK0 += Q(b)^T dS0 | P1, dS1 of block b: operations 0..23 from g25 | start values of block b + 1
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                mma_acc(dv[0][i & 1], dot[i].u, x4(xp[0], i >> 1));
                if constexpr (STAGE == 2) {
                    if (i == 0) {
                        stg.store(dst, dst + SLOT);
                        store_stats(dst);
                    }
                }
                if (i == 1) dkv_ops<4, 8, MC>(sc[1], dp[1], xp[1], xs[1], row0, lh, lq, true);
                if (i == 2) dkv_ops<12, 15, MC>(sc[1], dp[1], xp[1], xs[1], row0, lh, lq, true);
                if (i == 3) dkv_ops<18, 21, MC>(sc[1], dp[1], xp[1], xs[1], row0, lh, lq, true);
                read_init(nl, nstat, i);
                ACAI_SB();
                mma_acc(dk[0][i & 1], qt[i].u, x4(xs[0], i >> 1));
                if (i == 0) dkv_ops<0, 4, MC>(sc[1], dp[1], xp[1], xs[1], row0, lh, lq, true);
                if (i == 1) dkv_ops<8, 12, MC>(sc[1], dp[1], xp[1], xs[1], row0, lh, lq, true);
                if (i == 2) dkv_ops<15, 18, MC>(sc[1], dp[1], xp[1], xs[1], row0, lh, lq, true);
                if (i == 3) dkv_ops<21, 24, MC>(sc[1], dp[1], xp[1], xs[1], row0, lh, lq, true);
                read_init(nd, nstat + KT, i);
                ACAI_SB();
            }
        }
    };
    auto slot = [&](int t) -> unsigned char * { return lds + (t % 3) * QSLOT; };
    auto stats = [&](unsigned char *sl) -> const float * { return reinterpret_cast<const float *>(sl + 2 * SLOT); };
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    typedef std::integral_constant<int, 2> S2;

    if (nqt > 0) {
        // ---- prologue: tiles 0 and 1 into the ring, the row fragments and start values of block 0 --------------------------------------------------
        stg.load(0);
        load_stats(0);
        stg.store(slot(0), slot(0) + SLOT);
        store_stats(slot(0));
        stg.load(1);
        load_stats(1);
        stg.store(slot(1), slot(1) + SLOT);
        store_stats(slot(1));
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qr[s] = read_r(slot(0), s);
            dor[s] = read_r(slot(0) + SLOT, s);
            read_init(nl, stats(slot(0)), s);
            read_init(nd, stats(slot(0)) + KT, s);
        }
        // tile t: block 2t issues the loads of tile t+2; block 2t+1 writes them to the slot tile t-1 left; one barrier per tile
        auto tile = [&](auto first_, auto mask_, int t, unsigned char *s0, unsigned char *s1, unsigned char *dst) __attribute__((always_inline)) {
            constexpr bool M = decltype(mask_)::value;
            block_iter(first_, Y{}, std::integral_constant<bool, M>{}, std::integral_constant<bool, M>{}, S1{}, t * KT, s0, s0 + 32 * PITCH, stats(s0) + 32, t + 2);
            block_iter(N{}, Y{}, std::integral_constant<bool, M>{}, std::integral_constant<bool, M>{}, S2{}, t * KT + 32, s0 + 32 * PITCH, s1, stats(s1), 0, dst);
            __syncthreads();
        };
        int t = 0;
        if (nqt > 4) {
            tile(Y{}, N{}, 0, lds + 0 * QSLOT, lds + 1 * QSLOT, lds + 2 * QSLOT);
            tile(N{}, N{}, 1, lds + 1 * QSLOT, lds + 2 * QSLOT, lds + 0 * QSLOT);
            tile(N{}, N{}, 2, lds + 2 * QSLOT, lds + 0 * QSLOT, lds + 1 * QSLOT);
            for (t = 3; t + 3 < nqt; t += 3) {
                tile(N{}, N{}, t, lds + 0 * QSLOT, lds + 1 * QSLOT, lds + 2 * QSLOT);
                tile(N{}, N{}, t + 1, lds + 1 * QSLOT, lds + 2 * QSLOT, lds + 0 * QSLOT);
                tile(N{}, N{}, t + 2, lds + 2 * QSLOT, lds + 0 * QSLOT, lds + 1 * QSLOT);
            }
        } else {
            tile(Y{}, Y{}, 0, slot(0), slot(1), slot(2));
            t = 1;
        }
        for (; t < nqt; ++t) tile(N{}, Y{}, t, slot(t), slot(t + 1), slot(t + 2));
        // ---- drain: P1 / dS1 and dV1 / dK1 of the last block ------------------------------------------------------------------------------------------
        block_iter(N{}, N{}, Y{}, Y{}, S0{}, nqt * KT, lds, lds, stats(lds));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // (asm MFMAs: no compiler-tracked hazard in front of the accumulator reads below)

    const float ksc = 0.6931471805599453f;   // dK = dS^T Q = dS^T Q' sqrt(d_h) / log2(e), times 1 / sqrt(d_h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        bf16_t *rk = DK + (size_t)my_k[j] * a.lddk, *rv = DV + (size_t)my_k[j] * a.lddv;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                uint2 pk, pv;
                pk.x = pack_bf16(dk[j][d][4 * g4 + 0] * ksc, dk[j][d][4 * g4 + 1] * ksc);
                pk.y = pack_bf16(dk[j][d][4 * g4 + 2] * ksc, dk[j][d][4 * g4 + 3] * ksc);
                pv.x = pack_bf16(dv[j][d][4 * g4 + 0], dv[j][d][4 * g4 + 1]);
                pv.y = pack_bf16(dv[j][d][4 * g4 + 2], dv[j][d][4 * g4 + 3]);
                *reinterpret_cast<uint2 *>(rk + d * 32 + 8 * g4 + 4 * lh) = pk;
                *reinterpret_cast<uint2 *>(rv + d * 32 + 8 * g4 + 4 * lh) = pv;
            }
    }
}

}  // namespace

void acai_attn_bwd64w_dq_launch(const BwdArgs &a, int B, int max_q, hipStream_t st) {
    BwdArgs w = a;
    w.nblk = max_q / RBG;
    const int grid = w.nblk * a.H * B;
    if (!acai_xcd_order((long long)B * max_q == (long long)a.total_q)) w.nblk = -w.nblk;   // (nblk < 0: plain block order)
    if (grid > 0) hipLaunchKernelGGL(attn_bwd64w_dq_kernel, dim3(grid), dim3(NT), 0, st, w);
}

void acai_attn_bwd64w_dkv_launch(const BwdArgs &a, int B, int max_k, int equal_len, hipStream_t st) {
    BwdArgs w = a;
    w.nblk = max_k / RBG;
    const int grid = w.nblk * a.H * B;
    if (!acai_xcd_order(equal_len != 0)) w.nblk = -w.nblk;
    if (grid > 0) hipLaunchKernelGGL(attn_bwd64w_dkv_kernel, dim3(grid), dim3(NT), 0, st, w);
}
