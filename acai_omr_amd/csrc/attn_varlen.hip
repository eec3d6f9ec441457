// Flash-style varlen attention forward on packed ragged streams (no padding, no mask tensor):
// replaces nn.MultiheadAttention's SDPA inside nn.TransformerEncoderLayer / nn.TransformerDecoderLayer
// (reference: acai_omr/models/models.py:30-34,186-190,351-360,422-426; masks M:70-73, M:468).
//
// gfx950 design.  Workgroup = 4 waves = 128 queries of one (sequence, head); each wave owns 32 queries
// and streams 64-key K/V tiles that the workgroup stages in LDS (both in their natural [key][d] image; the
// V^T fragments are read with ds_read_b64_tr_b16 - bf16 - or one ds_read_b32 per K=2 MFMA - fp32).  Both products are "swapped" so that a query lives on a LANE and keys/d live in REGISTERS:
//   S^T[key][q] = K . Q^T   A = K rows (ds_read_b128), B = Q fragments held in registers for the whole kernel
//   O^T[d][q]   = V^T . P^T A = V^T rows (LDS),          B = P, taken straight from the S^T accumulators
// With the 32x32 MFMA C/D layout (col = lane&31, row = (reg&3)+8*(reg>>2)+4*(lane>>5)) the softmax row
// statistics are per-lane scalars (31 v_max + one cross-half exchange), the rescale of O is a per-lane
// multiply, and P never touches LDS: the accumulator registers 8s..8s+7, packed to bf16, ARE the B
// fragment of k-step s when the A fragment takes keys 16s + 8(j>>2) + 4h + (j&3) (fp32: one register per
// K=2 MFMA with A key (i&3)+8(i>>2)+4h).  bf16 -> v_mfma_f32_32x32x16_bf16, fp32 -> v_mfma_f32_32x32x2_f32.
// Online softmax in fp32 with exp2 and a finite -1e30 floor; global loads of tile t+1 are issued before
// the MFMAs of tile t (issue-early / write-late) into the other of two LDS stages: one barrier per tile.
#include "attn_args.h"

#ifndef ACAI_ATTN_QK_FIRST
#define ACAI_ATTN_QK_FIRST 0
#endif
#ifndef ACAI_FWD_WAVES
#define ACAI_FWD_WAVES 2
#endif
#ifndef ACAI_ATTN_PSUM4
#define ACAI_ATTN_PSUM4 1
#endif
#ifndef ACAI_ATTN_MFMA_SUM
#define ACAI_ATTN_MFMA_SUM 1
#endif

namespace {

constexpr int KT = 64;   // keys per tile
constexpr int QB = 128;  // queries per workgroup

// NQ = 32-query blocks per wave (1 or 2).  NQ = 2: a wave owns 64 queries and every K / V^T fragment it reads from LDS feeds two MFMAs, one per
// block; the tile loop's fixed costs (fragment reads, staging, waits, barrier, branches: ~150 of ~600 issue cycles per 2048 scores at d_h = 32)
// are spent once per 4096 scores.  Costs the third wave per SIMD (~200 registers).
template <typename T, int DHP, bool FAST, bool DROP, bool PRE, int NQ = 1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ACAI_FWD_WAVES))) void attn_fwd_kernel(AttnArgs a) {
    constexpr int QBW = 32 * NQ, QBG = 4 * QBW;   // queries per wave / per workgroup
    constexpr int ES = sizeof(T);
    constexpr int EPC = 16 / ES;                 // elements per 16-byte chunk
    typedef TileLayout<ES, DHP> TL;              // natural [key][d] image of the K and V tiles (swizzled bf16 / padded fp32)
    constexpr int KPITCH = TL::PITCH;
    constexpr int NS = DHP * ES / 32;            // 16-byte fragments per lane along d (per lane-half)
    constexpr int NDB = DHP / 32;                // 32-wide d blocks of the output
    constexpr int CPR = DHP / EPC;               // 16-byte chunks per K/V row
    constexpr int NCH = KT * CPR / 256;          // chunks per thread per operand
    constexpr int STAGE = 2 * KT * KPITCH;       // one K tile + one V tile
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];   // two stages: tile t+1 is written while tile t is read

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform (SGPR): tile-class branches are scalar
    const int lr = lane & 31, lh = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q_start = a.cu_q[b], lq = a.cu_q[b + 1] - q_start;
    const int k_start = a.cu_k[b], lk = a.cu_k[b + 1] - k_start;
    const int q0 = blockIdx.x * QBG;
    if (q0 >= lq) return;  // whole workgroup exits together: no barrier has been reached yet

    const T *Q = reinterpret_cast<const T *>(a.q) + (size_t)q_start * a.ldq + h * a.dh;
    const T *K = reinterpret_cast<const T *>(a.k) + (size_t)k_start * a.ldk + h * a.dh;
    const T *V = reinterpret_cast<const T *>(a.v) + (size_t)k_start * a.ldv + h * a.dh;
    T *O = reinterpret_cast<T *>(a.out) + (size_t)q_start * a.ldo + h * a.dh;
    const int dh = a.dh;
    // PRE: q arrives multiplied by log2(e) / sqrt(d_h) (the in-projection's epilogue did it before the one rounding), so K . Q^T already is the
    // score in the log2 domain and no per-score multiply is left
    const float c = PRE ? 1.0f : a.scale_log2e;

    auto load16 = [&](const T *base, int ld, int row, int rows, int d0) -> uint4 {
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
    };

    // ---- Q fragments: lane (q = lr, half lh) keeps d = (32 s + 16 lh)/ES .. for s = 0..NS-1 -----------------
    int my_q[NQ];
    uint4 qf[NQ][NS];
    f32x16 oacc[NQ][NDB];
    float m_run[NQ], l_run[NQ];  // reference maximum (log2 domain) and this lane-half's partial row sum
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        my_q[j] = q0 + wave * QBW + j * 32 + lr;
#pragma unroll
        for (int s = 0; s < NS; ++s) qf[j][s] = load16(Q, a.ldq, my_q[j], lq, (s * 32 + lh * 16) / ES);
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[j][d][e] = 0.f;
        m_run[j] = -1.0e30f;
        l_run[j] = 0.f;
    }
    // Row sums stay on the VALU.  Measured alternative: one extra MFMA per 16 keys with an all-ones A operand replaces the 32 adds per tile,
    // but the kernel got 20 % SLOWER (2.0 -> 2.46 ms at d_h = 32): the four dependent MFMAs per tile on one accumulator hold the wave's
    // issue port longer than the adds they replace.

    int nkt = (lk + KT - 1) / KT;
    if (a.causal) {
        const int last_q = min(q0 + QBG, lq) - 1;
        nkt = min(nkt, last_q / KT + 1);
    }

    // staging: thread -> NCH 16-byte chunks of the K and of the V tile; global pointers advance by one tile per call
    uint4 rk[NCH], rv[NCH];
    const T *kp[NCH], *vp[NCH];
    int srow[NCH], soff[NCH];
    bool dok[NCH];
    auto stage_at = [&](int tile) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int cidx = tid + 256 * i, row = cidx / CPR, cc = cidx % CPR;
            srow[i] = row;
            soff[i] = TL::off(row, cc);
            dok[i] = cc * EPC < dh;
            kp[i] = K + ((size_t)tile * KT + row) * a.ldk + cc * EPC;
            vp[i] = V + ((size_t)tile * KT + row) * a.ldv + cc * EPC;
        }
    };
    auto load_tile = [&](int kt) {
        if constexpr (FAST) {
            const bool full = (kt + 1) * KT <= lk;  // wave-uniform: interior tiles load unguarded
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const bool ok = dok[i] && (full || kt * KT + srow[i] < lk);
                rk[i] = rv[i] = make_uint4(0, 0, 0, 0);
                if (ok) {
                    rk[i] = *reinterpret_cast<const uint4 *>(kp[i]);
                    rv[i] = *reinterpret_cast<const uint4 *>(vp[i]);
                }
                kp[i] += (size_t)KT * a.ldk;
                vp[i] += (size_t)KT * a.ldv;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int cidx = tid + 256 * i, d0 = (cidx % CPR) * EPC;
                rk[i] = load16(K, a.ldk, kt * KT + srow[i], lk, d0);
                rv[i] = load16(V, a.ldv, kt * KT + srow[i], lk, d0);
            }
        }
    };
    auto store_tile = [&](unsigned char *stage) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            *reinterpret_cast<uint4 *>(stage + soff[i]) = rk[i];
            *reinterpret_cast<uint4 *>(stage + KT * KPITCH + soff[i]) = rv[i];
        }
    };

    // S^T += K . Q^T of the two 32-key blocks of a tile, on top of whatever `sacc` holds.  d-slice outer, key block inner: consecutive MFMAs
    // alternate between the two accumulators (a chain of NS dependent MFMAs on one accumulator waits out the full MFMA latency at every link)
    auto qk = [&](f32x16 (&sacc)[NQ][2], const unsigned char *ldsK, int kb_lo = 0, int kb_hi = 2) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int kb = kb_lo; kb < kb_hi; ++kb) {
                const uint4 kf = *reinterpret_cast<const uint4 *>(ldsK + TL::off(kb * 32 + lr, 2 * s + lh));   // one read, NQ products
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    if constexpr (ES == 2) {
                        sacc[j][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf),
                                                                              __builtin_bit_cast(bf16x8, qf[j][s]), sacc[j][kb], 0, 0, 0);
                    } else {
                        const f32x4 k4 = __builtin_bit_cast(f32x4, kf), q4 = __builtin_bit_cast(f32x4, qf[j][s]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) sacc[j][kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(k4[e], q4[e], sacc[j][kb], 0, 0, 0);
                    }
                }
            }
        }
    };
    // O^T += V^T . P^T with P taken straight from the score accumulators (dropout, if any, masks the P that multiplies V only: the
    // normaliser uses the undropped probabilities)
    // MFSUM (bf16, no dropout, fast loop): the row sums leave the VALU.  The packed P fragment of 16 keys, read as the B operand of a 16x16x32
    // MFMA, is [k-group g = lane >> 4][n = lane & 15] with g = 0 / 2 the two key halves of query n and g = 1 / 3 those of query n + 16; against
    // an A operand of ones in the k-groups {0, 2} for rows 0..7 and {1, 3} for rows 8..15 the product's rows 0..7 are the 16-key sum of query n
    // and rows 8..15 that of query n + 16 - four 16-cycle MFMAs (8 issue cycles each) per tile on two alternating accumulators instead of 17
    // v_pk_add_f32 behind the tile's last MFMA.  The sums are those of the bf16-ROUNDED probabilities, i.e. of exactly what multiplies V.
    constexpr bool MFSUM = ACAI_ATTN_MFMA_SUM && ES == 2 && !DROP;
    f32x4 lsum[NQ][2];
#pragma unroll
    for (int j = 0; j < NQ; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) lsum[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint32_t selw = (((lane >> 4) & 1) == ((lane >> 3) & 1)) ? 0x3F803F80u : 0u;
    const uint4 sel = make_uint4(selw, selw, selw, selw);
    auto pv = [&](f32x16 (&sacc)[NQ][2], const unsigned char *ldsV, int kt, bool sum = false, int kb_lo = 0, int kb_hi = 2) {
        if constexpr (DROP) {
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const uint32_t rrow = (uint32_t)(h * a.total_q + q_start + my_q[j]);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const uint32_t key = (uint32_t)(kt * KT + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh);
                        sacc[j][kb][e] = drop_keep(a.drop_seed, rrow, key, a.drop_thr) ? sacc[j][kb][e] * a.drop_scale : 0.f;
                    }
            }
        }
#pragma unroll
        for (int kb = kb_lo; kb < kb_hi; ++kb) {
            if constexpr (ES == 2) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    uint4 pf[NQ];
#pragma unroll
                    for (int j = 0; j < NQ; ++j) {
                        pf[j].x = pack_bf16(sacc[j][kb][8 * s2 + 0], sacc[j][kb][8 * s2 + 1]);
                        pf[j].y = pack_bf16(sacc[j][kb][8 * s2 + 2], sacc[j][kb][8 * s2 + 3]);
                        pf[j].z = pack_bf16(sacc[j][kb][8 * s2 + 4], sacc[j][kb][8 * s2 + 5]);
                        pf[j].w = pack_bf16(sacc[j][kb][8 * s2 + 6], sacc[j][kb][8 * s2 + 7]);
                        if constexpr (MFSUM) {
                            if (sum) lsum[j][s2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, sel), __builtin_bit_cast(bf16x8, pf[j]), lsum[j][s2], 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int d = 0; d < NDB; ++d) {
                        // A element j = V[key 16 s2 + 8 (j>>2) + 4 lh + (j&3)][d]: two 4-key x 16-d transposing reads of the natural V tile
                        // (lane 4q+p of a 16-lane group addresses key row q, columns 4p..4p+3; lane i receives column i)
                        typedef __attribute__((ext_vector_type(4))) short s4;
                        typedef __attribute__((address_space(3))) s4 *lds_s4;
                        const int i16 = lane & 15, g1 = (lane >> 4) & 1;
                        const int vrow = kb * 32 + 16 * s2 + 4 * lh + (i16 >> 2), vchunk = d * 4 + 2 * g1 + ((i16 & 3) >> 1), vsub = 8 * (i16 & 1);
                        union { s4 v[2]; uint4 u; } vf;
                        vf.v[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(ldsV + TL::off(vrow, vchunk) + vsub));
                        vf.v[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(ldsV + TL::off(vrow + 8, vchunk) + vsub));
#pragma unroll
                        for (int j = 0; j < NQ; ++j)   // one V^T fragment, NQ products
                            oacc[j][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf.u),
                                                                                __builtin_bit_cast(bf16x8, pf[j]), oacc[j][d], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
#pragma unroll
                    for (int d = 0; d < NDB; ++d) {
                        // A for MFMA step i = 4 g4 + e: V[key e + 8 g4 + 4 lh][d]: one conflict-free ds_read_b32 per K=2 MFMA
                        const unsigned char *vr = ldsV + (kb * 32 + 8 * g4 + 4 * lh) * KPITCH + (d * 32 + lr) * 4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float vv = *reinterpret_cast<const float *>(vr + e * KPITCH);
#pragma unroll
                            for (int j = 0; j < NQ; ++j) oacc[j][d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, sacc[j][kb][4 * g4 + e], oacc[j][d], 0, 0, 0);
                        }
                    }
                }
            }
        }
    };

    stage_at(0);
    load_tile(0);
    store_tile(lds);
    __syncthreads();

    // ---- fast loop: the leading tiles whose every key is attended by every query of the WORKGROUP -------------------------------------
    // At d_h <= 64 the VALU issue port, not the MFMA, bounds this kernel (PMC: 85 % of the SIMD cycles issue, 22 % of them MFMA), so the
    // common tile carries nothing but 1 exp2 + 1 add per score and one bf16 pack per two (PRE) - no maximum, no rescale, no compare:
    //   * the reference maximum is the row maximum of tile 0, taken once in front of the loop;
    //   * PRE: the score accumulators START at -m, so the MFMA leaves (score - m) and the exponential reads it directly;
    //   * fp32 holds 2^(score - m) until score - m reaches 2^7: a row sum above 2^80 (or inf / NaN) sets `bad`, and a workgroup with a bad
    //     lane starts over in the general loop below, which keeps a running maximum.  (LayerNorm-ed activations never come near; the
    //     general loop is also what the masked tiles take.)
    // Keeping the rescale out of this loop matters beyond the compare: as `if (raise) O *= alpha` in front of the P.V MFMAs it made the
    // compiler keep two copies of O (16 v_mov_b64 per tile behind an s_nop that waits out the last MFMA).
    int n_fast = min(nkt, (a.causal ? min(lk, q0 + 1) : lk) / KT);
    bool bad = false;
    // ZREF (two blocks per wave): the reference "maximum" of the fast loop is ZERO - probabilities are 2^score as they stand, the score
    // accumulators start at the inline constant 0 instead of a 16-register vector of -m per block (the registers the second block needs at
    // d_h = 64), and no tile-0 maximum is taken.  fp32 / bf16 hold 2^score for scores in (-126, 127) (log2 domain: +-85 in natural units,
    // LayerNorm-ed activations stay within a few tens); a row sum beyond 2^100 or below 2^-100 (or inf / NaN) restarts the workgroup in
    // the general loop like any other overflow.  O / l does not depend on the reference.
    constexpr bool ZREF = NQ == 2 && PRE && MFSUM;
    if (ZREF && n_fast > 0) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) m_run[j] = 0.f;
    } else if (n_fast > 0) {
        f32x16 s0[NQ][2];
#pragma unroll
        for (int j = 0; j < NQ; ++j)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) s0[j][kb][e] = 0.f;
        qk(s0, lds);
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            float tmax = -1.0e30f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) tmax = fmaxf(tmax, s0[j][kb][e]);
            tmax *= c;
            m_run[j] = fmaxf(tmax, __shfl_xor(tmax, 32));
        }
    }
    f32x16 minit[NQ];   // PRE: start value of the score accumulators
#pragma unroll
    for (int j = 0; j < NQ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) minit[j][e] = (PRE && !ZREF) ? -m_run[j] : 0.f;
    // A wave whose 32 queries all lie past the end of the sequence (513 decoder tokens: the fifth 128-query block holds ONE row) only helps
    // with the staging: it skips the products and the softmax, which leaves the CU's issue slots to the co-resident workgroups.  Its own loop,
    // so that the working waves' loop stays one basic block.
    const bool wave_active = q0 + wave * QBW < lq;
    if (!wave_active) {
        for (int kt = 0; kt < n_fast; ++kt) {
            if (kt + 1 < nkt) {
                load_tile(kt + 1);
                store_tile(lds + ((kt + 1) & 1) * STAGE);
            }
            __syncthreads();
        }
    } else {
        // One tile of the fast loop.  `cur` / `nxt` are the LDS stages as COMPILE-TIME offsets (the loop below is unrolled by two): with the
        // stage chosen by `kt & 1` every swizzled fragment address cost a v_add / v_or per tile on top of its lane offset - 14 VALU instructions
        // per tile at d_h = 32 and 27 at d_h = 64 on a loop that is bound by the issue port; as constants they fold into the reads' offset fields.
        auto fast_tile = [&](int kt, const unsigned char *cur, unsigned char *nxt) {
            const unsigned char *ldsK = cur, *ldsV = cur + KT * KPITCH;
            if (kt + 1 < nkt) load_tile(kt + 1);
            f32x16 sacc[NQ][2];
            // NQ = 1: both 32-key blocks of the tile in flight (the second block's products run under the first block's exponentials).
            // NQ = 2 at d_h = 64: one key block at a time for both query blocks - four score tiles at once do not fit the registers
            constexpr int KSTEP = (NQ == 2 && DHP == 64) ? 1 : 2;
#pragma unroll
            for (int k0 = 0; k0 < 2; k0 += KSTEP) {
#pragma unroll
                for (int j = 0; j < NQ; ++j)
#pragma unroll
                    for (int kb = k0; kb < k0 + KSTEP; ++kb) sacc[j][kb] = minit[j];
                qk(sacc, ldsK, k0, k0 + KSTEP);
#if ACAI_ATTN_QK_FIRST
                __builtin_amdgcn_sched_barrier(0);   // all four S^T MFMAs first: the second key block's run under the first block's exponentials
#endif
                // four independent partial sums: one running sum made a chain of 32 dependent v_add_f32 per tile (a dependent add issues every
                // ~6.6 cycles instead of 4: +80 cycles per tile on a loop whose floor is ~520)
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kb = k0; kb < k0 + KSTEP; ++kb)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const float p = PRE ? fast_exp2(sacc[j][kb][e]) : fast_exp2(fmaf(sacc[j][kb][e], c, -m_run[j]));
                            sacc[j][kb][e] = p;
                            if constexpr (!MFSUM) ps[ACAI_ATTN_PSUM4 ? (e & 3) : 0] += p;
                        }
                    if constexpr (!MFSUM) {
                        const float psum = (ps[0] + ps[1]) + (ps[2] + ps[3]);
                        bad |= !(psum < 1.2e24f);   // 2^80; also true for inf and NaN
                        l_run[j] += psum;
                    }
                }
                pv(sacc, ldsV, kt, true, k0, k0 + KSTEP);
            }
            if (kt + 1 < nkt) store_tile(nxt);
            __syncthreads();   // one barrier per tile: the other stage was last read in iteration kt-1
        };
        int kt = 0;
        for (; kt + 1 < n_fast; kt += 2) {
            fast_tile(kt, lds, lds + STAGE);
            fast_tile(kt + 1, lds + STAGE, lds);
        }
        if (kt < n_fast) fast_tile(kt, lds, lds + STAGE);
    }
    if constexpr (MFSUM) {
        // rows 0..7 (any register of lanes 0..31) hold the sum of query n = lane & 15, rows 8..15 (lanes 32..63) that of query n + 16; l_run is
        // a per-lane-half partial sum (the halves meet after the loops), so the whole sum goes to the lower half.  One check for the whole
        // loop: a probability beyond 2^100 (or inf / NaN) shows in its row's sum
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const float full = __shfl(lsum[j][0][0] + lsum[j][1][0], (lane & 16) ? 32 + (lane & 15) : (lane & 15));
            if (wave_active && n_fast > 0) {
                bad |= !(full < 1.2e30f) || (ZREF && !(full > 1.0e-30f));
                l_run[j] = lh == 0 ? full : 0.f;
            }
        }
    }
    int kt0 = n_fast;
    if (n_fast > 0 && __syncthreads_or(bad)) {   // start over with a running maximum (all waves: the barrier count must match)
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
#pragma unroll
            for (int d = 0; d < NDB; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) oacc[j][d][e] = 0.f;
            m_run[j] = -1.0e30f;
            l_run[j] = 0.f;
        }
        kt0 = 0;
        stage_at(0);
        load_tile(0);
        store_tile(lds);
        __syncthreads();
    }

    // ---- general loop: masked tiles (ragged end, causal diagonal) and restarted workgroups: online softmax with a running maximum ------
    if (!wave_active) {
        for (int kt = kt0; kt < nkt; ++kt) {
            if (kt + 1 < nkt) {
                load_tile(kt + 1);
                store_tile(lds + ((kt + 1) & 1) * STAGE);
            }
            __syncthreads();
        }
    } else
    for (int kt = kt0; kt < nkt; ++kt) {
        const unsigned char *ldsK = lds + (kt & 1) * STAGE, *ldsV = ldsK + KT * KPITCH;
        if (kt + 1 < nkt) load_tile(kt + 1);
        f32x16 sacc[NQ][2];
#pragma unroll
        for (int j = 0; j < NQ; ++j)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) sacc[j][kb][e] = 0.f;
        qk(sacc, ldsK);

        // ---- mask + online softmax (per-lane query) ----------------------------------------------------
        const int wave_lim = a.causal ? min(lk, q0 + wave * QBW + 1) : lk;  // keys < wave_lim are valid for ALL lanes of the wave
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int key_lim = a.causal ? min(lk, my_q[j] + 1) : lk;  // keys < key_lim are attended
            // Lazy rescale: the running maximum is only raised (and O, l rescaled) when some lane's tile maximum exceeds it by more than 2^8;
            // otherwise probabilities are taken against the stale maximum (p <= 256: exact in the final O / l ratio up to rounding).
            auto raise_max = [&](float tmax) {
                if (__builtin_amdgcn_ballot_w64(tmax > m_run[j] + 8.0f) != 0) {  // wave-uniform
                    const float m2 = fmaxf(m_run[j], tmax);
                    const float alpha = fast_exp2(m_run[j] - m2);
                    m_run[j] = m2;
                    l_run[j] *= alpha;
#pragma unroll
                    for (int d = 0; d < NDB; ++d)
#pragma unroll
                        for (int e = 0; e < 16; ++e) oacc[j][d][e] *= alpha;
                }
            };
            float psum = 0.f;
            if ((kt + 1) * KT <= wave_lim) {
                float tmax = -1.0e30f;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) tmax = fmaxf(tmax, sacc[j][kb][e]);
                tmax *= c;
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
                raise_max(tmax);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float p = fast_exp2(fmaf(sacc[j][kb][e], c, -m_run[j]));
                        sacc[j][kb][e] = p;
                        psum += p;
                    }
            } else {
                float tmax = -1.0e30f;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int key = kt * KT + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                        const float sv = key < key_lim ? sacc[j][kb][e] * c : -1.0e30f;
                        sacc[j][kb][e] = sv;
                        tmax = fmaxf(tmax, sv);
                    }
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
                raise_max(tmax);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        // masked entries: gate on the mask value itself (a fully masked row keeps m = -1e30 and exp2(0) = 1 would be wrong)
                        const float p = sacc[j][kb][e] > -0.5e30f ? fast_exp2(sacc[j][kb][e] - m_run[j]) : 0.f;
                        sacc[j][kb][e] = p;
                        psum += p;
                    }
            }
            l_run[j] += psum;
        }
        pv(sacc, ldsV, kt);
        if (kt + 1 < nkt) store_tile(lds + ((kt + 1) & 1) * STAGE);
        __syncthreads();   // one barrier per tile: stage (kt+1)&1 was last read in iteration kt-1
    }

    // ---- normalise and store: lane owns query my_q, registers hold d = db*32 + (e&3) + 8*(e>>2) + 4*lh ------
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const float l_tot = l_run[j] + __shfl_xor(l_run[j], 32);
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        if (a.lse && my_q[j] < lq && lh == 0) a.lse[(size_t)h * a.total_q + q_start + my_q[j]] = m_run[j] + log2f(l_tot);
        if (my_q[j] < lq) {
            T *orow = O + (size_t)my_q[j] * a.ldo;
#pragma unroll
            for (int d = 0; d < NDB; ++d)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int d0 = d * 32 + 8 * g4 + 4 * lh;
                    if constexpr (FAST) {
                        if (d0 < dh) {
                            if constexpr (ES == 2) {
                                uint2 p;
                                p.x = pack_bf16(oacc[j][d][4 * g4 + 0] * inv, oacc[j][d][4 * g4 + 1] * inv);
                                p.y = pack_bf16(oacc[j][d][4 * g4 + 2] * inv, oacc[j][d][4 * g4 + 3] * inv);
                                *reinterpret_cast<uint2 *>(orow + d0) = p;
                            } else {
                                float4 p = make_float4(oacc[j][d][4 * g4 + 0] * inv, oacc[j][d][4 * g4 + 1] * inv,
                                                       oacc[j][d][4 * g4 + 2] * inv, oacc[j][d][4 * g4 + 3] * inv);
                                *reinterpret_cast<float4 *>(orow + d0) = p;
                            }
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (d0 + e < dh) DT<T>::st(orow + d0 + e, oacc[j][d][4 * g4 + e] * inv);
                    }
                }
        }
    }
}

template <typename T, int DHP>
int launch(const AttnArgs &a, int B, int max_q, bool pre, hipStream_t st) {
    constexpr int EPC = 16 / sizeof(T);
    const bool fast = (a.dh % EPC == 0) && (a.ldq % EPC == 0) && (a.ldk % EPC == 0) && (a.ldv % EPC == 0) && (a.ldo % EPC == 0) &&
                      aligned16(a.q) && aligned16(a.k) && aligned16(a.v) && aligned16(a.out);
    dim3 grid(cdiv(max_q, QB), a.H, B);
    // train-mode attention dropout is its own instantiation, so the common kernel carries no hash code / registers; the prescaled-q form
    // (training path) exists for 16-byte-aligned operands only
    if (pre) {
        if (!fast) return acai_set_err(-1, "acai_attn_varlen_fwd: q_prescaled needs 16-byte aligned operands and d_h %% %d == 0", EPC);
        if (a.drop_thr)
            hipLaunchKernelGGL((attn_fwd_kernel<T, DHP, true, true, true>), grid, dim3(256), 0, st, a);
        else if constexpr (sizeof(T) == 2) {
            // the training steps' form: two query blocks per wave (ACAI_ATTN_NQ=1: one, A/B aid) when the sequences are long enough to fill the chip
            static const int nq_env = getenv("ACAI_ATTN_NQ") ? atoi(getenv("ACAI_ATTN_NQ")) : 2;
            // (d_h = 64 keeps one block: two need 256 registers + 55 spilt even with the zero reference, and measured 5 % slower - 0.97 against 0.92 ms)
            bool two = false;
            if constexpr (DHP == 64) {
                // d_h = 64 exactly, no causal mask: the software-pipelined kernel of attn_fwd64.hip (ACAI_ATTN64=0: this file's kernel, A/B aid)
                static const int f64_env = getenv("ACAI_ATTN64") ? atoi(getenv("ACAI_ATTN64")) : 1;
                if (f64_env && a.dh == 64 && !a.causal) {
                    const int rc = acai_attn_fwd64_launch(a, B, max_q, st);
                    ACAI_LAUNCH_CHECK("acai_attn_varlen_fwd");
                    return rc;
                }
            }
            if constexpr (DHP == 32) {
                two = nq_env == 2 && max_q >= 512;
                if (two) hipLaunchKernelGGL((attn_fwd_kernel<T, DHP, true, false, true, 2>), dim3(cdiv(max_q, 2 * QB), a.H, B), dim3(256), 0, st, a);
            }
            if (!two) hipLaunchKernelGGL((attn_fwd_kernel<T, DHP, true, false, true>), grid, dim3(256), 0, st, a);
        } else
            hipLaunchKernelGGL((attn_fwd_kernel<T, DHP, true, false, true>), grid, dim3(256), 0, st, a);
    } else if (a.drop_thr) {
        if (fast)
            hipLaunchKernelGGL((attn_fwd_kernel<T, DHP, true, true, false>), grid, dim3(256), 0, st, a);
        else
            hipLaunchKernelGGL((attn_fwd_kernel<T, DHP, false, true, false>), grid, dim3(256), 0, st, a);
    } else if (fast)
        hipLaunchKernelGGL((attn_fwd_kernel<T, DHP, true, false, false>), grid, dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((attn_fwd_kernel<T, DHP, false, false, false>), grid, dim3(256), 0, st, a);
    ACAI_LAUNCH_CHECK("acai_attn_varlen_fwd");
    return 0;
}

}  // namespace

// q_prescaled: q already carries the factor log2(e) / sqrt(dh) (acai_gemm_nt_ex's column scale on the in-projection); the kernel then
// takes K . Q^T as the score in the log2 domain.  `lse` has the same meaning either way.
extern "C" int acai_attn_varlen_fwd(const void *q, int ldq, const void *k, int ldk, const void *v, int ldv, void *out, int ldo,
                                    const int32_t *cu_q, const int32_t *cu_k, int B, int H, int dh, int max_q, int causal,
                                    int dtype, float *lse, int total_q, float dropout_p, uint32_t dropout_seed, int q_prescaled, void *stream) {
    ACAI_CHECK_ARG(q && k && v && out && cu_q && cu_k, "acai_attn_varlen_fwd: null operand");
    ACAI_CHECK_ARG(B > 0 && H > 0 && dh > 0 && dh <= 64 && max_q > 0, "acai_attn_varlen_fwd: bad dims B=%d H=%d dh=%d max_q=%d (dh <= 64)", B, H, dh, max_q);
    ACAI_CHECK_ARG(ldq >= H * dh && ldk >= H * dh && ldv >= H * dh && ldo >= H * dh, "acai_attn_varlen_fwd: row stride smaller than H*dh");
    ACAI_CHECK_ARG(B <= 65535 && H <= 65535, "acai_attn_varlen_fwd: grid too large");
    ACAI_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "acai_attn_varlen_fwd: dropout_p out of range");
    AttnArgs a{q, k, v, out, cu_q, cu_k, ldq, ldk, ldv, ldo, H, dh, causal, 0.f, (uint32_t)((double)dropout_p * 4294967296.0), dropout_seed,
               1.0f / (1.0f - dropout_p), lse, total_q};
    a.scale_log2e = 1.4426950408889634f / sqrtf((float)dh);
    hipStream_t st = (hipStream_t)stream;
    const bool pre = q_prescaled != 0;
    if (dtype == ACAI_BF16) return dh <= 32 ? launch<bf16_t, 32>(a, B, max_q, pre, st) : launch<bf16_t, 64>(a, B, max_q, pre, st);
    if (dtype == ACAI_F32) return dh <= 32 ? launch<float, 32>(a, B, max_q, pre, st) : launch<float, 64>(a, B, max_q, pre, st);
    return acai_set_err(-1, "acai_attn_varlen_fwd: bad dtype %d", dtype);
}
