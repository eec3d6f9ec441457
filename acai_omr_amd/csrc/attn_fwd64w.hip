// Varlen flash attention forward, bf16, d_h = 64, q prescaled, no dropout, no causal mask: ONE WAVE PER SIMD, 64 queries per wave
// (reference call sites: acai_omr/models/models.py:29-33, 351-360 encoder self-attention; :466-482 teacher-forced decoder cross attention).
//
// Why this form (tools/ablate_attn64.py, profiles/r04_attn64_ablation.txt): at d_h = 64 the 32-queries-per-wave kernels are bound by the
// LDS, not by the VALU and not by the matrix pipe - every 32x32x16 MFMA takes a fresh 1 KB A fragment (K rows or V^T) from LDS, 8 waves
// per CU pull 128 B/clk of fragments plus the tiles' 16-byte stores against a 256 B/clk array, and with the fragment reads removed the
// same loop runs at the MFMA floor with all its exponentials and staging in place.  Two lane-owned 32-query blocks per wave halve the
// bytes per MFMA (each fragment feeds two products); at d_h = 64 that needs ~330 registers per lane, i.e. one wave per SIMD with the whole
// 512-entry file: the output accumulators (64) and row sums live in the ACCUMULATOR half as the C/D operands of inline-asm MFMAs
// ("+a"), everything the VALU touches - score tiles, probabilities, fragments - in the architectural half.  The file is built with
// -mllvm -amdgpu-mfma-vgpr-form so that the score MFMAs (builtins: the compiler tracks their MFMA -> VALU hazards) write VGPRs; without it
// every MFMA result lands in the accumulator half and each exponential pays a v_accvgpr_read.
//
// With one wave per SIMD nothing but the wave's own instruction order overlaps the VALU with the matrix pipe, so the loop is software-
// pipelined over 32-key blocks exactly as attn_fwd64_kernel's (S two blocks ahead of P V, the exponentials of block j+1 behind the
// MFMAs of blocks j+2 and j; order pinned by sched_barrier), 16 MFMA gaps per block iteration with 2 exp2 + 1 pack + <= 2 LDS reads each.
// Fragment registers are reloaded for the NEXT block iteration right behind the second MFMA that read them (~15 gaps of latency cover);
// for that to run across tile boundaries the K and V rings have three slots: during tile t, K(t+1) and V(t) are computed on, K(t+2) and
// V(t+1) are resident for the reloads, K(t+3) and V(t+2) are on their way through registers and are written at the tile's end.
// One barrier per tile.  Probabilities are 2^score against a zero reference; a row sum outside (2^-100, 2^100) sends the workgroup
// through a plain two-pass loop (exact row maxima, then the products again).
#include "attn_args.h"

#include <type_traits>

namespace {

typedef TileLayout<2, 64> TL;
constexpr int KT = 64, PITCH = 128, SLOT = KT * PITCH;   // one 64-row x 64-col bf16 tile: 8 KB
constexpr int NT = 256, NCH = KT * 8 / NT;               // threads per workgroup; 16-byte chunks per thread and operand tile
constexpr int QBG = 256;                                 // queries per workgroup (4 waves x 2 blocks x 32)
typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((address_space(3))) s4 *lds_s4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define ACAI_SB() __builtin_amdgcn_sched_barrier(0)

// O (32 x 32 fp32, accumulator registers) += A . B on the matrix pipe.  Inline asm so that C/D sit in the accumulator half; the operands are
// LDS-read fragments (the compiler's lgkmcnt wait precedes the statement) and probabilities packed a block iteration earlier.
__device__ __forceinline__ void mma_acc(f32x16 &c, const uint4 &a, const uint4 &b) {
    const u32x4 av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(av), "v"(bv));
}
__device__ __forceinline__ void mma16_acc(f32x4 &c, const uint4 &a, const uint4 &b) {
    const u32x4 av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(av), "v"(bv));
}

// ABL (timing ablations, -DACAI_ATTN64_ABLATE builds only; results are wrong): 1 no exp2, 2 no pack, 4 no LDS fragment reads, 8 no staging /
// barrier in the steady loop, 16 no S MFMAs, 32 no P V MFMAs, 64 no row-sum MFMAs, 128 no barrier, 256 no global loads, 512 no LDS stores of the staged tiles
template <int ABL>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_fwd64w_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[6 * SLOT];   // K ring: slots 0..2; V ring: slots 3..5

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    // XCD-aware block order.  Workgroups are dealt round-robin over the 8 XCDs by linear id, and the nqb query blocks of one (sequence, head)
    // stream the same K / V (1 MB at 4096 keys): with (block, head, sequence) taken straight from the id, every XCD's 4 MB L2 saw the K / V of all
    // ~16 pairs in flight and served the tiles from the Infinity Cache.  Ids congruent mod 8 (one XCD, dispatched together) get a contiguous run of
    // virtual ids instead, so a pair's blocks share one L2.  (Placement is a speed matter only.)
    // nqb < 0: a RAGGED batch, plain order - the static split leaves the XCD with the longest sequences working alone at the end (attn_bwd1p.hip).
    int vid = blockIdx.x;
    const int nqb = a.nqb < 0 ? -a.nqb : a.nqb;
    if (a.nqb > 0) {
        const int per = gridDim.x >> 3;
        if (vid < (per << 3)) vid = (vid & 7) * per + (vid >> 3);
    }
    const int qb = vid % nqb, h = (vid / nqb) % a.H, b = vid / (nqb * a.H);
    const int q_start = a.cu_q[b], lq = a.cu_q[b + 1] - q_start;
    const int k_start = a.cu_k[b], lk = a.cu_k[b + 1] - k_start;
    const int q0 = qb * QBG;
    if (q0 >= lq || (a.tail && lq - q0 <= a.tail)) return;   // a last block of <= a.tail (32) rows belongs to attn_fwd64_tail_kernel

    const bf16_t *Q = reinterpret_cast<const bf16_t *>(a.q) + (size_t)q_start * a.ldq + h * 64;
    const bf16_t *K = reinterpret_cast<const bf16_t *>(a.k) + (size_t)k_start * a.ldk + h * 64;
    const bf16_t *V = reinterpret_cast<const bf16_t *>(a.v) + (size_t)k_start * a.ldv + h * 64;
    bf16_t *O = reinterpret_cast<bf16_t *>(a.out) + (size_t)q_start * a.ldo + h * 64;
    const int nkt = (lk + KT - 1) / KT;

    // ---- Q fragments (B operand of S^T = K Q^T): lane (q = lr, half lh) of block j keeps d = 16 s + 8 lh .. + 7 ----------------------------
    int my_q[2];
    uint4 qf[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        my_q[j] = q0 + wave * 64 + j * 32 + lr;
#pragma unroll
        for (int s = 0; s < 4; ++s)
            qf[j][s] = my_q[j] < lq ? *reinterpret_cast<const uint4 *>(Q + (size_t)my_q[j] * a.ldq + s * 16 + lh * 8) : make_uint4(0, 0, 0, 0);
    }

    // ---- staging: thread -> NCH chunks of a K and of a V tile, through registers (buffer loads: rows past the sequence end read as zeros) ---
    int soff[NCH];
    uint32_t gk[NCH], gv[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int cidx = tid + NT * i, row = cidx >> 3, cc = cidx & 7;
        soff[i] = TL::off(row, cc);
        gk[i] = (uint32_t)(row * a.ldk * 2 + cc * 16);
        gv[i] = (uint32_t)(row * a.ldv * 2 + cc * 16);
    }
    u32x4 rk[NCH], rv[NCH];
    auto load_k = [&](int t) {
        const int rows = lk - t * KT;
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(K + (size_t)t * KT * a.ldk), 0, rows > 0 ? rows * a.ldk * 2 : 0, 0x00020000);
#pragma unroll
        for (int i = 0; i < NCH; ++i) rk[i] = __builtin_amdgcn_raw_buffer_load_b128(r, gk[i], 0, 0);
    };
    auto load_v = [&](int t) {
        const int rows = lk - t * KT;
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(V + (size_t)t * KT * a.ldv), 0, rows > 0 ? rows * a.ldv * 2 : 0, 0x00020000);
#pragma unroll
        for (int i = 0; i < NCH; ++i) rv[i] = __builtin_amdgcn_raw_buffer_load_b128(r, gv[i], 0, 0);
    };
    auto store_k = [&](unsigned char *slot) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) *reinterpret_cast<u32x4 *>(slot + soff[i]) = rk[i];
    };
    auto store_v = [&](unsigned char *slot) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) *reinterpret_cast<u32x4 *>(slot + soff[i]) = rv[i];
    };

    // ---- fragment addresses inside a tile (32-key block, k-step and ring slot are compile-time offsets in the hot loop) ---------------------
    int kaddr[4];   // K rows: key lr of the block, 16-byte chunk 2 s + lh
#pragma unroll
    for (int s = 0; s < 4; ++s) kaddr[s] = TL::off(lr, 2 * s + lh);
    // V^T fragment of k-step s2 and d block d: two 4-key x 16-d transposing reads, key rows L and L + 8 with L = 4 lh + (i16 >> 2) (+ 16 s2)
    const int i16 = lane & 15, g1 = (lane >> 4) & 1;
    int vaddr[2][2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) vaddr[d][r2] = TL::off(4 * lh + (i16 >> 2) + 8 * r2, d * 4 + 2 * g1 + ((i16 & 3) >> 1)) + 8 * (i16 & 1);
    auto read_k = [&](const unsigned char *kblock, int s) -> uint4 { return *reinterpret_cast<const uint4 *>(kblock + kaddr[s]); };
    auto read_v = [&](const unsigned char *vblock, int d, int s2) -> uint4 {
        union { s4 v[2]; uint4 u; } f;
        f.v[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(vblock + s2 * 16 * PITCH + vaddr[d][0]));
        f.v[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(vblock + s2 * 16 * PITCH + vaddr[d][1]));
        return f.u;
    };
    auto mma = [&](const uint4 &af, const uint4 &bf, const f32x16 &c) -> f32x16 {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bf), c, 0, 0, 0);
    };
    auto pv4 = [&](const uint32_t (&pf)[8], int s2) -> uint4 { return make_uint4(pf[4 * s2], pf[4 * s2 + 1], pf[4 * s2 + 2], pf[4 * s2 + 3]); };

    const uint32_t selw = (((lane >> 4) & 1) == ((lane >> 3) & 1)) ? 0x3F803F80u : 0u;   // row-sum selector (see attn_fwd_kernel)
    const uint4 sel = make_uint4(selw, selw, selw, selw);

    f32x16 oacc[2][2];          // [query block][d block], accumulator registers
    f32x4 lsum[2];              // [query block], accumulator registers
    f32x16 sA[2], sB[2];        // score tiles of the two query blocks: one pair being written by the MFMAs, one being exponentiated
    uint32_t pfA[2][8], pfB[2][8];
    uint4 kf[4], vf[4];         // K row fragments (k-steps 0..3) and V^T fragments ((d0, lo), (d1, lo), (d0, hi), (d1, hi)) of the CURRENT block iteration
    f32x16 zero16;
#pragma unroll
    for (int e = 0; e < 16; ++e) zero16[e] = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        oacc[j][0] = oacc[j][1] = zero16;
        lsum[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // One block iteration = 16 MFMA gaps.
    //   DO_S: sx[j] = S^T of the 32-key block whose fragments are in kf; each kf[s] is then reloaded from `knext` (the next iteration's block)
    //   DO_E: P = 2^sy[j], packed into pfy[j] (MASK: keys >= lk get probability zero; key0 = first key of sy's block)
    //   DO_P: O[j] += V^T P(pfx[j]) for the block whose fragments are in vf, and its row sums; each vf[i] is then reloaded from `vnext`
    //   k_dst / v_dst (non-null in the hot loop's second iteration of a tile): the staged K / V tile registers are written there in gap 12
    auto block_iter = [&](auto do_s_, auto do_e_, auto mask_, auto do_p_, auto rl_k_, auto rl_v_, f32x16 (&sx)[2], const unsigned char *knext, f32x16 (&sy)[2],
                          uint32_t (&pfy)[2][8], int key0, const uint32_t (&pfx)[2][8], const unsigned char *vnext, auto stage_, unsigned char *k_dst = nullptr,
                          unsigned char *v_dst = nullptr) {
        constexpr bool STAGE = decltype(stage_)::value;
        constexpr bool DO_S = decltype(do_s_)::value, DO_E = decltype(do_e_)::value, MASK = decltype(mask_)::value, DO_P = decltype(do_p_)::value;
        constexpr bool RL_K = decltype(rl_k_)::value, RL_V = decltype(rl_v_)::value;
        auto E = [&](int j, int g) {
            if constexpr (DO_E) {
                float p0 = (ABL & 1) ? sy[j][2 * g] : fast_exp2(sy[j][2 * g]), p1 = (ABL & 1) ? sy[j][2 * g + 1] : fast_exp2(sy[j][2 * g + 1]);
                if constexpr (MASK) {
                    const int ka = key0 + ((2 * g) & 3) + 8 * ((2 * g) >> 2) + 4 * lh;
                    p0 = ka < lk ? p0 : 0.f;
                    p1 = ka + 1 < lk ? p1 : 0.f;
                }
                if constexpr (ABL & 2) {
                    asm volatile("" ::"v"(p1));
                    pfy[j][g] = __float_as_uint(p0);
                } else
                    pfy[j][g] = pack_bf16(p0, p1);
            }
        };
        auto opaque4 = [&](uint4 &f) { asm volatile("" : "+v"(f.x), "+v"(f.y), "+v"(f.z), "+v"(f.w)); };
        if constexpr (DO_S && (ABL & 16) != 0) asm volatile("" : "+v"(sx[0]), "+v"(sx[1]));
        if constexpr (DO_P && (ABL & 32) != 0)
            asm volatile("" ::"v"(pfx[0][0]), "v"(pfx[0][1]), "v"(pfx[0][2]), "v"(pfx[0][3]), "v"(pfx[0][4]), "v"(pfx[0][5]), "v"(pfx[0][6]), "v"(pfx[0][7]), "v"(pfx[1][0]),
                         "v"(pfx[1][1]), "v"(pfx[1][2]), "v"(pfx[1][3]), "v"(pfx[1][4]), "v"(pfx[1][5]), "v"(pfx[1][6]), "v"(pfx[1][7]));
#pragma unroll
        for (int s = 0; s < 4; ++s) {   // gaps 0..7: the score products, fragment s feeds both query blocks
            if constexpr (DO_S && !(ABL & 16)) sx[0] = mma(kf[s], qf[0][s], s == 0 ? zero16 : sx[0]);
            E(0, s);
            ACAI_SB();
            if constexpr (DO_S) {
                if constexpr (!(ABL & 16)) sx[1] = mma(kf[s], qf[1][s], s == 0 ? zero16 : sx[1]);
                if constexpr (RL_K) {
                    if constexpr (ABL & 4) opaque4(kf[s]);
                    else kf[s] = read_k(knext, s);
                }
            }
            E(1, s);
            ACAI_SB();
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // gaps 8..15: O += V^T P; fragment i = (d block i & 1, key half i >> 1)
            if constexpr (STAGE && !(ABL & 8)) {
                if (i == 2) {   // tile t's K(t+3) / V(t+2) registers -> their ring slots, in FRONT of the last four fragment reloads (see `tile`)
                    if constexpr (ABL & 512) {
#pragma unroll
                        for (int c = 0; c < NCH; ++c) asm volatile("" ::"v"(rk[c]), "v"(rv[c]));
                    } else {
                        store_k(k_dst);
                        store_v(v_dst);
                    }
                    ACAI_SB();
                }
            }
            if constexpr (DO_P && !(ABL & 32)) mma_acc(oacc[0][i & 1], vf[i], pv4(pfx[0], i >> 1));
            E(0, 4 + i);
            ACAI_SB();
            if constexpr (DO_P) {
                if constexpr (!(ABL & 32)) mma_acc(oacc[1][i & 1], vf[i], pv4(pfx[1], i >> 1));
                if constexpr (RL_V) {
                    if constexpr (ABL & 4) opaque4(vf[i]);
                    else vf[i] = read_v(vnext, i & 1, i >> 1);
                }
                if constexpr (!(ABL & 64)) mma16_acc(lsum[i >> 1], sel, pv4(pfx[i >> 1], i & 1));   // row sums: gaps 8..11 query block 0 (key halves lo, hi), 12..15 query block 1
            }
            E(1, 4 + i);
            ACAI_SB();
        }
    };
    typedef std::true_type Y;
    typedef std::false_type N;
    auto kslot = [&](int t) -> unsigned char * { return lds + (t % 3) * SLOT; };
    auto vslot = [&](int t) -> unsigned char * { return lds + (3 + t % 3) * SLOT; };

    // ---- prologue: K(0..2), V(0..1) into the rings; S(0); P(0, block 0); fragments of K(1) block 0 and V(0) block 0 --------------------------
    load_k(0);
    load_v(0);
    store_k(kslot(0));
    store_v(vslot(0));
    load_k(1);
    load_v(1);
    store_k(kslot(1));
    store_v(vslot(1));
    load_k(2);
    store_k(kslot(2));
    __syncthreads();
    if (nkt > 0) {
#pragma unroll
        for (int s = 0; s < 4; ++s) kf[s] = read_k(kslot(0), s);
        block_iter(Y{}, N{}, N{}, N{}, Y{}, N{}, sA, kslot(0) + 32 * PITCH, sB, pfB, 0, pfA, lds, N{});      // S(0, b0); kf <- K(0) block 1
        block_iter(Y{}, N{}, N{}, N{}, Y{}, N{}, sB, kslot(1), sA, pfA, 0, pfB, lds, N{});                   // S(0, b1); kf <- K(1) block 0
        if (nkt == 1) block_iter(N{}, Y{}, Y{}, N{}, N{}, N{}, sB, lds, sA, pfA, 0, pfB, lds, N{});          // P(0, b0)
        else block_iter(N{}, Y{}, N{}, N{}, N{}, N{}, sB, lds, sA, pfA, 0, pfB, lds, N{});
#pragma unroll
        for (int i = 0; i < 4; ++i) vf[i] = read_v(vslot(0), i & 1, i >> 1);
    }
    __syncthreads();   // K(0) has been read by every wave: tile 0's staging may overwrite its slot

    // ---- steady state: tile t computes S(t+1) beside P(t, b1) / P(t+1, b0) and O += V(t)^T P(t); K(t+3), V(t+2) pass through registers ------
    auto tile = [&](auto mask_, int t, const unsigned char *k1, const unsigned char *k2, const unsigned char *v0, const unsigned char *v1, unsigned char *k_dst,
                    unsigned char *v_dst) {
        if constexpr (!(ABL & 8) && !(ABL & 256)) {
            load_k(t + 3);
            load_v(t + 2);
        }
        // S(t+1, b0) | P(t, b1) | O += V(t, b0)^T P(t, b0);   kf <- K(t+1) block 1, vf <- V(t) block 1
        block_iter(Y{}, Y{}, N{}, Y{}, Y{}, Y{}, sA, k1 + 32 * PITCH, sB, pfB, 0, pfA, v0 + 32 * PITCH, N{});
        // S(t+1, b1) | P(t+1, b0) | O += V(t, b1)^T P(t, b1);  kf <- K(t+2) block 0, vf <- V(t+1) block 0;  gap 12: K(t+3), V(t+2) -> LDS
        block_iter(Y{}, Y{}, mask_, Y{}, Y{}, Y{}, sB, k2, sA, pfA, (t + 1) * KT, pfB, v1, Y{}, k_dst, v_dst);
        if constexpr (!(ABL & 8)) {
            // The tile's barrier WITHOUT draining the LDS queue: __syncthreads() waits lgkmcnt(0), i.e. for the fragment reloads issued a few
            // instructions earlier - with one wave per SIMD that emptied the pipeline once per tile (tools/ablate_attn64.py: staging + barrier
            // 22 % of the kernel).  LDS operations complete in order, and exactly four reads (the two halves of vf[2] and of vf[3]) are issued
            // behind the stores of gap 12: lgkmcnt(4) covers the stores and leaves those reads in flight across the barrier.
            __builtin_amdgcn_s_waitcnt(0xC07F | (4 << 8));
            if constexpr (!(ABL & 128)) __builtin_amdgcn_s_barrier();
        }
    };
    int t = 0;
    for (; t + 4 < nkt; t += 3) {   // three tiles, none of them next to the last one: static ring slots, no masks
        tile(N{}, t, lds + 1 * SLOT, lds + 2 * SLOT, lds + 3 * SLOT, lds + 4 * SLOT, lds + 0 * SLOT, lds + 5 * SLOT);
        tile(N{}, t + 1, lds + 2 * SLOT, lds + 0 * SLOT, lds + 4 * SLOT, lds + 5 * SLOT, lds + 1 * SLOT, lds + 3 * SLOT);
        tile(N{}, t + 2, lds + 0 * SLOT, lds + 1 * SLOT, lds + 5 * SLOT, lds + 3 * SLOT, lds + 2 * SLOT, lds + 4 * SLOT);
    }
    for (; t + 1 < nkt; ++t)        // the remaining steady tiles: P(t+1, b0) may belong to the ragged last tile
        tile(Y{}, t, kslot(t + 1), kslot(t + 2), vslot(t), vslot(t + 1), kslot(t), vslot(t + 2));
    // ---- drain: the last tile's P(t, b1) and both of its P V products --------------------------------------------------------------------
    if (nkt > 0) {
        block_iter(N{}, Y{}, Y{}, Y{}, N{}, Y{}, sA, lds, sB, pfB, t * KT + 32, pfA, vslot(t) + 32 * PITCH, N{});
        block_iter(N{}, N{}, N{}, Y{}, N{}, N{}, sA, lds, sB, pfA, 0, pfB, lds, N{});
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last accumulator MFMAs are asm statements: no compiler-tracked hazard in front of the reads below

    // rows 0..7 (lanes 0..31) hold the sum of query n = lane & 15, rows 8..15 (lanes 32..63) that of query n + 16 (see attn_fwd_kernel)
    float l_run[2], m_ref[2] = {0.f, 0.f};
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float full = __shfl(lsum[j][0], (lane & 16) ? 32 + (lane & 15) : (lane & 15));
        bad |= ABL == 0 && nkt > 0 && (!(full < 1.2e30f) || !(full > 1.0e-30f));
        l_run[j] = lh == 0 ? full : 0.f;
    }
    if (__syncthreads_or(bad)) {
        // ---- rare: a probability outside fp32's 2^+-100 window.  Exact row maxima, then the products again - plain loops, one tile in LDS ------
        float m[2] = {-1.0e30f, -1.0e30f};
        for (int kt = 0; kt < nkt; ++kt) {
            load_k(kt);
            __syncthreads();
            store_k(lds);
            __syncthreads();
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x16 s = zero16;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) s = mma(read_k(lds + kb * 32 * PITCH, ks), qf[j][ks], s);
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int key = kt * KT + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                        m[j] = key < lk ? fmaxf(m[j], s[e]) : m[j];
                    }
                }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            m[j] = fmaxf(m[j], __shfl_xor(m[j], 32));
            m_ref[j] = m[j];
            oacc[j][0] = oacc[j][1] = zero16;
            lsum[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        for (int kt = 0; kt < nkt; ++kt) {
            load_k(kt);
            load_v(kt);
            __syncthreads();
            store_k(lds);
            store_v(lds + 3 * SLOT);
            __syncthreads();
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x16 s = zero16;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) s = mma(read_k(lds + kb * 32 * PITCH, ks), qf[j][ks], s);
                    uint32_t pf[8];
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        const int ka = kt * KT + kb * 32 + ((2 * g) & 3) + 8 * ((2 * g) >> 2) + 4 * lh;
                        const float p0 = ka < lk ? fast_exp2(s[2 * g] - m[j]) : 0.f, p1 = ka + 1 < lk ? fast_exp2(s[2 * g + 1] - m[j]) : 0.f;
                        pf[g] = pack_bf16(p0, p1);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const uint4 vfr = read_v(lds + 3 * SLOT + kb * 32 * PITCH, i & 1, i >> 1);
                        asm volatile("s_nop 1" ::: "memory");
                        mma_acc(oacc[j][i & 1], vfr, make_uint4(pf[4 * (i >> 1)], pf[4 * (i >> 1) + 1], pf[4 * (i >> 1) + 2], pf[4 * (i >> 1) + 3]));
                    }
                    mma16_acc(lsum[j], sel, make_uint4(pf[0], pf[1], pf[2], pf[3]));
                    mma16_acc(lsum[j], sel, make_uint4(pf[4], pf[5], pf[6], pf[7]));
                }
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float full = __shfl(lsum[j][0], (lane & 16) ? 32 + (lane & 15) : (lane & 15));
            l_run[j] = lh == 0 ? full : 0.f;
        }
    }

    // ---- normalise and store: lane owns query my_q[j], registers hold d = db*32 + (e&3) + 8*(e>>2) + 4*lh ------------------------------------
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float l_tot = l_run[j] + __shfl_xor(l_run[j], 32);
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        if (my_q[j] < lq) {
            if (a.lse && lh == 0) a.lse[(size_t)h * a.total_q + q_start + my_q[j]] = m_ref[j] + log2f(l_tot);
            bf16_t *orow = O + (size_t)my_q[j] * a.ldo;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    uint2 p;
                    p.x = pack_bf16(oacc[j][d][4 * g4 + 0] * inv, oacc[j][d][4 * g4 + 1] * inv);
                    p.y = pack_bf16(oacc[j][d][4 * g4 + 2] * inv, oacc[j][d][4 * g4 + 3] * inv);
                    *reinterpret_cast<uint2 *>(orow + d * 32 + 8 * g4 + 4 * lh) = p;
                }
        }
    }
}

}  // namespace

void acai_attn_fwd64w_launch(const AttnArgs &a, int B, int max_q, hipStream_t st) {
    AttnArgs w = a;
    w.nqb = cdiv(max_q, QBG);
    const int grid = w.nqb * a.H * B;
    if (!acai_xcd_order((long long)B * max_q == (long long)a.total_q)) w.nqb = -w.nqb;
#ifdef ACAI_ATTN64_ABLATE
    const int abl = getenv("ACAI_ATTN64_ABL") ? atoi(getenv("ACAI_ATTN64_ABL")) : 0;
#define ACAI_ABL_CASE(X) case X: hipLaunchKernelGGL(attn_fwd64w_kernel<X>, dim3(grid), dim3(NT), 0, st, w); return;
    switch (abl) {
        ACAI_ABL_CASE(1) ACAI_ABL_CASE(3) ACAI_ABL_CASE(4) ACAI_ABL_CASE(8) ACAI_ABL_CASE(12) ACAI_ABL_CASE(16) ACAI_ABL_CASE(32) ACAI_ABL_CASE(48)
        ACAI_ABL_CASE(64) ACAI_ABL_CASE(15) ACAI_ABL_CASE(79) ACAI_ABL_CASE(112) ACAI_ABL_CASE(124) ACAI_ABL_CASE(7) ACAI_ABL_CASE(120) ACAI_ABL_CASE(76) ACAI_ABL_CASE(128) ACAI_ABL_CASE(256) ACAI_ABL_CASE(512) ACAI_ABL_CASE(384) ACAI_ABL_CASE(640) ACAI_ABL_CASE(768) ACAI_ABL_CASE(896)
        default: break;
    }
#endif
    hipLaunchKernelGGL(attn_fwd64w_kernel<0>, dim3(grid), dim3(NT), 0, st, w);
}
