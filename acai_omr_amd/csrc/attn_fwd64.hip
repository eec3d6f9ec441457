// Varlen flash attention forward, bf16, d_h = 64 - the SDPA of every encoder layer and of the teacher-forced decoder
// (reference: acai_omr/models/models.py:29-33, 351-360 encoder self-attention; :466-482 decoder self / cross attention) in the training
// steps' form: q prescaled by log2(e) / sqrt(d_h), no dropout, no causal mask.  Everything else stays in attn_varlen.hip.
//
// Same products and fragment layouts as attn_fwd_kernel (query on the lane, S^T = K Q^T, O^T = V^T P^T, P straight from the score
// accumulators, row sums by a 16x16x32 MFMA against a 0/1 selector), rebuilt around what bounded that kernel at d_h = 64: its tile was
// "QK^T chain -> s_nop -> 32 exponentials -> P V" in program order, an in-order wave cannot issue its exponentials under its own MFMAs, and
// two co-resident waves that meet at one barrier per tile run those phases in step - the matrix pipe sat idle ~50 % of the time
// (profiles/r04_attn64_pmc_summary.txt).  Here the loop is SOFTWARE-PIPELINED over 32-key blocks and the order is pinned in the source
// (sched_barrier between groups):
//     block iteration j:   S(j+2) = K Q^T   4 MFMAs   beside the first half of  P(j+1) = 2^S(j+1)  (8 exp2 + 4 packs)
//                          O += V^T P(j)    4 MFMAs   beside the second half
// so every MFMA has ~2 exponentials + 1 pack + 1-2 LDS reads of independent work behind it, and the score tile a wave exponentiates was
// finished four MFMAs earlier.  K is therefore read one 64-key tile ahead of V: two 2-slot rings (K(t+1) and V(t) are live during tile t),
// one barrier per tile, register staging through buffer loads (rows past the sequence end arrive as zeros: no exec-mask branches, no 64-bit
// pointer arithmetic in the loop; the resource is rebuilt per tile in SGPRs because only the VGPR offset is range-checked).
// Reference "maximum" zero as in the two-block d_h = 32 form: probabilities are 2^score as they stand; a row sum outside (2^-100, 2^100)
// makes the workgroup take the exact row maxima in a pre-pass and run the same loop again with the score accumulators starting at -m.
#include "attn_args.h"

#include <type_traits>

namespace {

typedef TileLayout<2, 64> TL;
constexpr int KT = 64, PITCH = 128, SLOT = KT * PITCH;   // one 64-row x 64-col bf16 tile: 8 KB
typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((address_space(3))) s4 *lds_s4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define ACAI_SB() __builtin_amdgcn_sched_barrier(0)

// ABL (timing ablations, -DACAI_ATTN64_ABLATE builds only; results are wrong): 1 no exp2, 2 no pack, 4 fragments from registers (no LDS reads),
// 8 no staging / barrier in the steady loop, 16 no S MFMAs, 32 no P V MFMAs, 64 no row-sum MFMAs
template <int NW, int ABL = 0>   // NW = waves per workgroup, 32 queries each
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2))) void attn_fwd64_kernel(AttnArgs a) {
    constexpr int NT = 64 * NW, QBG = 32 * NW, NCH = KT * 8 / NT;   // 16-byte chunks per thread and operand tile
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * SLOT];   // K ring: slots 0, 1; V ring: slots 2, 3

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q_start = a.cu_q[b], lq = a.cu_q[b + 1] - q_start;
    const int k_start = a.cu_k[b], lk = a.cu_k[b + 1] - k_start;
    const int q0 = blockIdx.x * QBG;
    if (q0 >= lq) return;

    const bf16_t *Q = reinterpret_cast<const bf16_t *>(a.q) + (size_t)q_start * a.ldq + h * 64;
    const bf16_t *K = reinterpret_cast<const bf16_t *>(a.k) + (size_t)k_start * a.ldk + h * 64;
    const bf16_t *V = reinterpret_cast<const bf16_t *>(a.v) + (size_t)k_start * a.ldv + h * 64;
    bf16_t *O = reinterpret_cast<bf16_t *>(a.out) + (size_t)q_start * a.ldo + h * 64;
    const int my_q = q0 + wave * 32 + lr;
    const int nkt = (lk + KT - 1) / KT;

    // ---- Q fragments (B operand of S^T = K Q^T): lane (q = lr, half lh) keeps d = 16 s + 8 lh .. + 7 ---------------------------------------
    uint4 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
        qf[s] = my_q < lq ? *reinterpret_cast<const uint4 *>(Q + (size_t)my_q * a.ldq + s * 16 + lh * 8) : make_uint4(0, 0, 0, 0);

    // ---- staging: thread -> NCH chunks of the K and of the V tile ------------------------------------------------------------------------
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
    auto load_k = [&](int t) {   // rows of tile t past the end of the sequence (and whole tiles past it) arrive as zeros
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

    // ---- fragment addresses inside a tile (the 32-key block, the k-step and the ring slot are compile-time offsets) -----------------------
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

    const uint32_t selw = (((lane >> 4) & 1) == ((lane >> 3) & 1)) ? 0x3F803F80u : 0u;   // row-sum selector (see attn_fwd_kernel)
    const uint4 sel = make_uint4(selw, selw, selw, selw);

    f32x16 oacc[2], sA, sB, minit;
    f32x4 lsum[2];
    uint32_t pfA[8], pfB[8];
    float m_ref = 0.f, l_run = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) minit[e] = 0.f;

    uint4 fixfrag = make_uint4(selw, lane, selw, tid);
    if constexpr (ABL & 4) asm volatile("" : "+v"(fixfrag.x), "+v"(fixfrag.y), "+v"(fixfrag.z), "+v"(fixfrag.w));
    auto read_k = [&](const unsigned char *kblock, int s) -> uint4 {
        if constexpr (ABL & 4) return fixfrag;
        return *reinterpret_cast<const uint4 *>(kblock + kaddr[s]);
    };
    auto read_v = [&](const unsigned char *vblock, int d, int s2) -> uint4 {
        if constexpr (ABL & 4) return fixfrag;
        union { s4 v[2]; uint4 u; } f;
        f.v[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(vblock + s2 * 16 * PITCH + vaddr[d][0]));
        f.v[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(vblock + s2 * 16 * PITCH + vaddr[d][1]));
        return f.u;
    };
    auto mma = [&](const uint4 &af, const uint4 &bf, const f32x16 &c) -> f32x16 {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bf), c, 0, 0, 0);
    };
    auto pv4 = [&](const uint32_t (&pf)[8], int s2) -> uint4 { return make_uint4(pf[4 * s2], pf[4 * s2 + 1], pf[4 * s2 + 2], pf[4 * s2 + 3]); };

    // One block iteration.  DO_S: sx = S^T of the 32-key block at `kblock` (K ring).  DO_E: P = 2^sy, packed into pfy (MASK: keys >= lk get
    // probability zero; key0 = first key of sy's block).  DO_P: O += V^T P(pfx) for the block at `vblock` (V ring), and its row sums.
    auto block_iter = [&](auto do_s_, auto do_e_, auto mask_, auto do_p_, f32x16 &sx, const unsigned char *kblock, f32x16 &sy, uint32_t (&pfy)[8],
                          int key0, const uint32_t (&pfx)[8], const unsigned char *vblock) {
        constexpr bool DO_S = decltype(do_s_)::value, DO_E = decltype(do_e_)::value, MASK = decltype(mask_)::value, DO_P = decltype(do_p_)::value;
        auto E = [&](int g) {
            if constexpr (DO_E) {
                float p0 = (ABL & 1) ? sy[2 * g] : fast_exp2(sy[2 * g]), p1 = (ABL & 1) ? sy[2 * g + 1] : fast_exp2(sy[2 * g + 1]);
                if constexpr (MASK) {
                    const int ka = key0 + ((2 * g) & 3) + 8 * ((2 * g) >> 2) + 4 * lh;
                    p0 = ka < lk ? p0 : 0.f;
                    p1 = ka + 1 < lk ? p1 : 0.f;
                }
                if constexpr (ABL & 2) {
                    asm volatile("" ::"v"(p1));
                    pfy[g] = __float_as_uint(p0);
                } else
                    pfy[g] = pack_bf16(p0, p1);
            }
        };
        constexpr bool DO_SM = DO_S && !(ABL & 16), DO_PM = DO_P && !(ABL & 32), DO_RS = DO_P && !(ABL & 64);
        if constexpr (DO_P && (ABL & 32) != 0) asm volatile("" ::"v"(pfx[0]), "v"(pfx[1]), "v"(pfx[2]), "v"(pfx[3]), "v"(pfx[4]), "v"(pfx[5]), "v"(pfx[6]), "v"(pfx[7]));
        uint4 kf0, kf1, kf2, kf3, v00, v10, v01, v11;
        if constexpr (DO_S) {
            kf0 = read_k(kblock, 0);
            kf1 = read_k(kblock, 1);
        }
        E(0);
        ACAI_SB();
        if constexpr (DO_S) {
            if constexpr (DO_SM) sx = mma(kf0, qf[0], minit);
            kf2 = read_k(kblock, 2);
        }
        E(1);
        ACAI_SB();
        if constexpr (DO_S) {
            if constexpr (DO_SM) sx = mma(kf1, qf[1], sx);
            kf3 = read_k(kblock, 3);
        }
        E(2);
        ACAI_SB();
        if constexpr (DO_SM) sx = mma(kf2, qf[2], sx);
        if constexpr (DO_P) v00 = read_v(vblock, 0, 0);
        E(3);
        ACAI_SB();
        if constexpr (DO_SM) sx = mma(kf3, qf[3], sx);
        if constexpr (DO_P) v10 = read_v(vblock, 1, 0);
        E(4);
        ACAI_SB();
        if constexpr (DO_P) {
            if constexpr (DO_PM) oacc[0] = mma(v00, pv4(pfx, 0), oacc[0]);
            v01 = read_v(vblock, 0, 1);
        }
        E(5);
        ACAI_SB();
        if constexpr (DO_P) {
            if constexpr (DO_PM) oacc[1] = mma(v10, pv4(pfx, 0), oacc[1]);
            v11 = read_v(vblock, 1, 1);
        }
        E(6);
        ACAI_SB();
        if constexpr (DO_RS) lsum[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, sel), __builtin_bit_cast(bf16x8, pv4(pfx, 0)), lsum[0], 0, 0, 0);
        if constexpr (DO_PM) oacc[0] = mma(v01, pv4(pfx, 1), oacc[0]);
        E(7);
        ACAI_SB();
        if constexpr (DO_PM) oacc[1] = mma(v11, pv4(pfx, 1), oacc[1]);
        if constexpr (DO_RS) lsum[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, sel), __builtin_bit_cast(bf16x8, pv4(pfx, 1)), lsum[1], 0, 0, 0);
        ACAI_SB();
    };
    typedef std::true_type Y;
    typedef std::false_type N;

    for (int attempt = 0;; ++attempt) {
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
        lsum[0] = lsum[1] = f32x4{0.f, 0.f, 0.f, 0.f};

        // ---- prologue: K(0), V(0), K(1) into the rings; S(0); P(0, block 0) ---------------------------------------------------------------
        load_k(0);
        load_v(0);
        store_k(lds);
        store_v(lds + 2 * SLOT);
        load_k(1);
        store_k(lds + SLOT);
        __syncthreads();
        if (nkt > 0) {
            block_iter(Y{}, N{}, N{}, N{}, sA, lds, sB, pfB, 0, pfA, lds);
            block_iter(Y{}, N{}, N{}, N{}, sB, lds + 32 * PITCH, sA, pfA, 0, pfB, lds);
            if (nkt == 1) block_iter(N{}, Y{}, Y{}, N{}, sB, lds, sA, pfA, 0, pfB, lds);
            else block_iter(N{}, Y{}, N{}, N{}, sB, lds, sA, pfA, 0, pfB, lds);
        }
        __syncthreads();   // K(0) has been read by every wave: tile 0's staging may overwrite its slot

        // ---- steady state: tile t computes S(t+1) beside P(t, block 1) / P(t+1, block 0) and O += V(t)^T P(t) ------------------------------
        // KS = ring slot of K(t+1) = (t+1) & 1, VS = ring slot of V(t) = t & 1; the staging of K(t+2) goes to K slot t & 1, V(t+1) to V slot (t+1) & 1
        auto tile = [&](auto mask_, int t, const unsigned char *kS, const unsigned char *vP, unsigned char *k_dst, unsigned char *v_dst) {
            if constexpr (!(ABL & 8)) {
                load_k(t + 2);
                load_v(t + 1);
            }
            block_iter(Y{}, Y{}, N{}, Y{}, sA, kS, sB, pfB, 0, pfA, vP);                                   // S(t+1, b0) | P(t, b1) | O += V(t, b0)^T P(t, b0)
            block_iter(Y{}, Y{}, mask_, Y{}, sB, kS + 32 * PITCH, sA, pfA, (t + 1) * KT, pfB, vP + 32 * PITCH);   // S(t+1, b1) | P(t+1, b0) | O += V(t, b1)^T P(t, b1)
            if constexpr (!(ABL & 8)) {
                store_k(k_dst);
                store_v(v_dst);
                __syncthreads();
            }
        };
        int t = 0;
        for (; t + 3 < nkt; t += 2) {   // tiles t and t + 1 are both followed by at least two more: no mask anywhere
            tile(N{}, t, lds + SLOT, lds + 2 * SLOT, lds, lds + 3 * SLOT);
            tile(N{}, t + 1, lds, lds + 3 * SLOT, lds + SLOT, lds + 2 * SLOT);
        }
        for (; t + 1 < nkt; ++t) {      // the last one or two steady tiles: P(t+1, b0) may belong to the ragged last tile
            const int ks = (t + 1) & 1, vs = t & 1;
            tile(Y{}, t, lds + ks * SLOT, lds + (2 + vs) * SLOT, lds + (ks ^ 1) * SLOT, lds + (2 + (vs ^ 1)) * SLOT);
        }
        // ---- drain: the last tile's P(t, b1) and both of its P V products ------------------------------------------------------------------
        if (nkt > 0) {
            const unsigned char *vP = lds + (2 + (t & 1)) * SLOT;
            block_iter(N{}, Y{}, Y{}, Y{}, sA, lds, sB, pfB, t * KT + 32, pfA, vP);
            block_iter(N{}, N{}, N{}, Y{}, sA, lds, sB, pfA, 0, pfB, vP + 32 * PITCH);
        }
        // rows 0..7 (lanes 0..31) hold the sum of query n = lane & 15, rows 8..15 (lanes 32..63) that of query n + 16 (see attn_fwd_kernel)
        const float full = __shfl(lsum[0][0] + lsum[1][0], (lane & 16) ? 32 + (lane & 15) : (lane & 15));
        const bool bad = ABL == 0 && nkt > 0 && (!(full < 1.2e30f) || !(full > 1.0e-30f));
        l_run = lh == 0 ? full : 0.f;
        if (attempt > 0 || !__syncthreads_or(bad)) break;

        // ---- exact row maxima (rare: a probability outside fp32's 2^+-100 window), then the same loop with the scores starting at -m -------
        float m = -1.0e30f;
        for (int kt = 0; kt < nkt; ++kt) {
            load_k(kt);
            __syncthreads();
            store_k(lds);
            __syncthreads();
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                f32x16 s;
#pragma unroll
                for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) s = mma(read_k(lds + kb * 32 * PITCH, ks), qf[ks], s);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = kt * KT + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    m = key < lk ? fmaxf(m, s[e]) : m;
                }
            }
        }
        m = fmaxf(m, __shfl_xor(m, 32));
        m_ref = m;
#pragma unroll
        for (int e = 0; e < 16; ++e) minit[e] = -m;
        __syncthreads();
    }

    // ---- normalise and store: lane owns query my_q, registers hold d = db*32 + (e&3) + 8*(e>>2) + 4*lh --------------------------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (my_q < lq) {
        if (a.lse && lh == 0) a.lse[(size_t)h * a.total_q + q_start + my_q] = m_ref + log2f(l_tot);
        bf16_t *orow = O + (size_t)my_q * a.ldo;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                uint2 p;
                p.x = pack_bf16(oacc[d][4 * g4 + 0] * inv, oacc[d][4 * g4 + 1] * inv);
                p.y = pack_bf16(oacc[d][4 * g4 + 2] * inv, oacc[d][4 * g4 + 3] * inv);
                *reinterpret_cast<uint2 *>(orow + d * 32 + 8 * g4 + 4 * lh) = p;
            }
    }
}

// ---- query tails of <= 32 rows (the 513th token of the teacher-forced decoder stream: models.py:531-540, omr_teacher_force_train.py:25) -------
// A sequence whose last 256-query block holds r <= 32 rows would occupy a whole CU of attn_fwd64w_kernel for the full key loop with one of its
// four SIMDs at work (513 queries x 4096 keys: 245 us against 146 us at 512 queries, tools/bench_cross_train_attn.py).  Here the KEYS of such
// a tail are split over the 8 waves of one workgroup instead: wave w takes tiles w, w + 8, ... through a wave-private LDS region (no
// barrier in the loop), with an ordinary online softmax (running maximum per wave - a tail is not worth a restart path), and the eight
// partial (m, l, O) triples meet in LDS at the end.  1/8 of the key loop per tail, 256 small workgroups for the decoder's 16 x 16 pairs.
__global__ __launch_bounds__(512) void attn_fwd64_tail_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[8 * 2 * SLOT];   // per wave: K tile, V tile (16 KB); reused for the partial results
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q_start = a.cu_q[b], lq = a.cu_q[b + 1] - q_start;
    const int k_start = a.cu_k[b], lk = a.cu_k[b + 1] - k_start;
    const int qbase = (lq / 256) * 256, r = lq - qbase;
    if (r == 0 || r > 32) return;   // (whole workgroup: no barrier has been reached)
    const bf16_t *Q = reinterpret_cast<const bf16_t *>(a.q) + (size_t)q_start * a.ldq + h * 64;
    const bf16_t *K = reinterpret_cast<const bf16_t *>(a.k) + (size_t)k_start * a.ldk + h * 64;
    const bf16_t *V = reinterpret_cast<const bf16_t *>(a.v) + (size_t)k_start * a.ldv + h * 64;
    bf16_t *O = reinterpret_cast<bf16_t *>(a.out) + (size_t)q_start * a.ldo + h * 64;
    const int my_q = qbase + lr, nkt = (lk + KT - 1) / KT;
    uint4 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = lr < r ? *reinterpret_cast<const uint4 *>(Q + (size_t)my_q * a.ldq + s * 16 + lh * 8) : make_uint4(0, 0, 0, 0);
    unsigned char *ldsK = lds + wave * 2 * SLOT, *ldsV = ldsK + SLOT;
    const int i16 = lane & 15, g1 = (lane >> 4) & 1;
    f32x16 oacc[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
    float m_run = -1.0e30f, l_run = 0.f;   // running maximum of the lane's query (log2 domain) and this lane-half's partial row sum
    u32x4 rk[8], rv[8];
    auto load_tile = [&](int t) {   // one wave stages both 8 KB tiles: chunk c = lane + 64 i -> row c / 8, 16-byte chunk c % 8 (tiles past the end: zeros)
        const int rows = lk - t * KT;
        const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(K + (size_t)t * KT * a.ldk), 0, rows > 0 ? rows * a.ldk * 2 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(V + (size_t)t * KT * a.ldv), 0, rows > 0 ? rows * a.ldv * 2 : 0, 0x00020000);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = lane + 64 * i, row = c >> 3, cc = c & 7;
            rk[i] = __builtin_amdgcn_raw_buffer_load_b128(rK, (uint32_t)(row * a.ldk * 2 + cc * 16), 0, 0);
            rv[i] = __builtin_amdgcn_raw_buffer_load_b128(rV, (uint32_t)(row * a.ldv * 2 + cc * 16), 0, 0);
        }
    };
    load_tile(wave);
    for (int t = wave; t < nkt; t += 8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = lane + 64 * i, off = TL::off(c >> 3, c & 7);
            *reinterpret_cast<u32x4 *>(ldsK + off) = rk[i];
            *reinterpret_cast<u32x4 *>(ldsV + off) = rv[i];
        }
        load_tile(t + 8);   // the wave's next tile travels while this one is computed on (a lone load -> compute chain took ~6 us per tile)
        // (wave-private image: the wave's own LDS operations complete in order - no barrier between its stores and its reads, nor between
        // this tile's reads and the next tile's stores)
        f32x16 sacc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int e = 0; e < 16; ++e) sacc[kb][e] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const uint4 kf = *reinterpret_cast<const uint4 *>(ldsK + TL::off(kb * 32 + lr, 2 * s + lh));
                sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[s]), sacc[kb], 0, 0, 0);
            }
        }
        float tmax = -1.0e30f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = t * KT + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                sacc[kb][e] = key < lk ? sacc[kb][e] : -1.0e30f;
                tmax = fmaxf(tmax, sacc[kb][e]);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m2 = fmaxf(m_run, tmax), alpha = fast_exp2(m_run - m2);   // (every tile holds at least one key: m2 is a real score)
        m_run = m2;
        l_run *= alpha;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[d][e] *= alpha;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            uint32_t pf[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float p0 = fast_exp2(sacc[kb][2 * g] - m_run), p1 = fast_exp2(sacc[kb][2 * g + 1] - m_run);   // masked keys: 2^(-1e30) = 0
                pf[g] = pack_bf16(p0, p1);
                l_run += round_bf16(p0) + round_bf16(p1);   // the sum of what multiplies V, as the MFMA row sums of the main kernels
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const int vrow = kb * 32 + 16 * s2 + 4 * lh + (i16 >> 2), vchunk = d * 4 + 2 * g1 + ((i16 & 3) >> 1), vsub = 8 * (i16 & 1);
                    union { s4 v[2]; uint4 u; } vf;
                    vf.v[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(ldsV + TL::off(vrow, vchunk) + vsub));
                    vf.v[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(ldsV + TL::off(vrow + 8, vchunk) + vsub));
                    const uint4 pb = make_uint4(pf[4 * s2], pf[4 * s2 + 1], pf[4 * s2 + 2], pf[4 * s2 + 3]);
                    oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf.u), __builtin_bit_cast(bf16x8, pb), oacc[d], 0, 0, 0);
                }
        }
    }
    // ---- the eight partial results meet in LDS: wave w's region holds O_w[q][d] (fp32, 8 KB), then m_w[q], l_w[q] ------------------------------
    float *part = reinterpret_cast<float *>(ldsK);
    const float l_w = l_run + __shfl_xor(l_run, 32);
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) part[lr * 64 + d * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh] = oacc[d][e];
    if (lh == 0) {
        part[2048 + lr] = m_run;
        part[2048 + 32 + lr] = l_w;
    }
    __syncthreads();
    const int q = tid >> 4, d0 = (tid & 15) * 4;   // 512 threads x 4 outputs
    float m = -1.0e30f;
#pragma unroll
    for (int w = 0; w < 8; ++w) m = fmaxf(m, reinterpret_cast<const float *>(lds + w * 2 * SLOT)[2048 + q]);
    float l = 0.f, o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const float *pw = reinterpret_cast<const float *>(lds + w * 2 * SLOT);
        const float f = fast_exp2(pw[2048 + q] - m);   // a wave without tiles: m_w = -1e30, l_w = 0, O_w = 0 -> f * 0
        l += f * pw[2048 + 32 + q];
        const f32x4 ov = *reinterpret_cast<const f32x4 *>(pw + q * 64 + d0);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += f * ov[e];
    }
    if (q < r) {
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        uint2 pk;
        pk.x = pack_bf16(o[0] * inv, o[1] * inv);
        pk.y = pack_bf16(o[2] * inv, o[3] * inv);
        *reinterpret_cast<uint2 *>(O + (size_t)(qbase + q) * a.ldo + d0) = pk;
        if (a.lse && d0 == 0) a.lse[(size_t)h * a.total_q + q_start + qbase + q] = m + log2f(l);
    }
}

}  // namespace

int acai_attn_fwd64_launch(const AttnArgs &a, int B, int max_q, hipStream_t st) {
    static const int nw_env = getenv("ACAI_ATTN64_NW") ? atoi(getenv("ACAI_ATTN64_NW")) : 4;
#ifdef ACAI_ATTN64_ABLATE
    const int abl = getenv("ACAI_ATTN64_ABL") ? atoi(getenv("ACAI_ATTN64_ABL")) : 0;
#define ACAI_ABL_CASE(X) case X: hipLaunchKernelGGL((attn_fwd64_kernel<4, X>), dim3(cdiv(max_q, 128), a.H, B), dim3(256), 0, st, a); return 0;
    switch (abl) {
        ACAI_ABL_CASE(1) ACAI_ABL_CASE(3) ACAI_ABL_CASE(4) ACAI_ABL_CASE(8) ACAI_ABL_CASE(12) ACAI_ABL_CASE(16) ACAI_ABL_CASE(32) ACAI_ABL_CASE(48)
        ACAI_ABL_CASE(64) ACAI_ABL_CASE(15) ACAI_ABL_CASE(79) ACAI_ABL_CASE(112) ACAI_ABL_CASE(124) ACAI_ABL_CASE(7)
        default: break;
    }
#endif
    // 0: the 32-queries-per-wave kernel everywhere; 1 (default): 64 queries per wave, one wave per SIMD, with a 128-query tail launch
    static const int wide_env = getenv("ACAI_ATTN64_WIDE") ? atoi(getenv("ACAI_ATTN64_WIDE")) : 1;
    if (wide_env) {
        AttnArgs w = a;
        // A sequence's last 256-query block goes to the tail kernel when it holds <= 32 rows.  When every sequence is max_q long (B * max_q rows
        // in all) the host knows whether such a tail exists and launches only what is needed; were that reading of the arguments wrong, the
        // wide kernel (tail = 0) still covers every row.
        const bool all_equal = (long long)B * max_q == (long long)a.total_q;
        const int rem = max_q % 256;
        const bool need_tail = !all_equal || (rem > 0 && rem <= 32);
        w.tail = need_tail ? 32 : 0;
        // (equal lengths: the wide grid covers the full blocks only - workgroups that exit at once still wait for a whole free CU each, one
        // per (sequence, head): 192 against 147 us on the decoder's 513-query cross attention)
        const int wide_q = (all_equal && need_tail) ? max_q - rem : max_q;
        if (wide_q > 0) acai_attn_fwd64w_launch(w, B, wide_q, st);
        if (need_tail) hipLaunchKernelGGL(attn_fwd64_tail_kernel, dim3(1, a.H, B), dim3(512), 0, st, w);
        return 0;
    }
    if (nw_env == 8)
        hipLaunchKernelGGL((attn_fwd64_kernel<8>), dim3(cdiv(max_q, 256), a.H, B), dim3(512), 0, st, a);
    else
        hipLaunchKernelGGL((attn_fwd64_kernel<4>), dim3(cdiv(max_q, 128), a.H, B), dim3(256), 0, st, a);
    return 0;
}
