// Varlen attention BACKWARD in ONE PASS, bf16, d_h = 32, q prescaled, no dropout, no causal mask; ragged sequences (the MAE decoder's
// self-attention: acai_omr/models/models.py:186-190; backward call site acai_omr/train/pre_train.py:59).
//
// attn_bwd.hip runs two kernels - dQ with the query on the lane, dK / dV with the key on the lane - and each recomputes S and P: seven products and two
// exponentials per score, and at d_h = 32 those kernels are bound by VALU issue (one exp2 + one multiply + packs per score AND kernel).  This kernel
// computes P and dS ONCE per score with the key on the lane - dK / dV accumulate in registers - and forms the query gradient from the same dS:
//     dQ^T[d][q] += K^T[d][key] dS^T[key][q]
// whose contraction runs over the lane dimension: dS crosses LDS once (the wave writes the [key][q] image of its own 32 keys and reads it back with the
// transposing fragment read as the MFMA's B operand; the K^T fragments are read the same way, once, from a prologue image of the wave's K rows, and
// stay in registers), and a query's gradient is a sum over the workgroups that own its keys: the four waves of a workgroup (512 keys) add their
// partial tiles in LDS and the workgroup adds the result to an fp32 buffer with no-return float atomics, 128 contiguous bytes per query row and 32
// lanes (measured 1.35 TB/s chip-wide, profiles/r04_atomic_rate.txt; 4096 keys / 512 = 8 adds per element: 2.15 GB per MAE decoder layer = 1.58 ms of
// atomic-unit time, inside the kernel's 2.9 ms).  A first small kernel forms delta and zeroes that buffer, a last one scales it into the bf16 gradient.
// The sum over key blocks is in arrival order: dQ is reproducible to fp32 rounding, not bit for bit (ACAI_ATTN_BWD_1P=0 keeps the two-pass kernels,
// which are).  Measured: 32 x 16 heads x 4096^2 3.00 ms against 3.78 for the two kernels; DESIGN.md section 5 has the counters and ablations.
//
// One wave per SIMD, four lane-owned 32-key blocks per wave: dK / dV 128 accumulator registers, the K / V fragments (B operands of S and dP) 64 and the
// K^T fragments 32 more, pinned there by inline-asm MFMAs (file built with -mllvm -amdgpu-mfma-vgpr-form), and a software pipeline over the work items w = (32-query block,
// owned block j): slot w issues the four S / dP MFMAs of item w+1 and the six gradient MFMAs of item w-1 (dV, dK, dQ) while the VALU turns item w's
// scores into P and dS - 48 numbered single-issue operations dealt over the slot's ten MFMA gaps.  EVERY tile runs the same code (the first tile's
// "previous item" multiplies zero packs): with first- / last-tile variants of the tile body the accumulator tuples met in register copies and
// scratch round trips at every region boundary (hundreds of spills - what sank the first version of this kernel).  And since with one wave per SIMD
// nothing issues beside the wave's own stream - every instruction of any kind is a four-cycle issue turn (profiles/r04_valu_issue.txt) - the steady
// state carries no padding s_nop, no register copies and no address arithmetic that a rotating offset or a scalar operand can replace.
#include "attn_bwd_args.h"

#include <type_traits>

namespace {

typedef TileLayout<2, 32> TL;
constexpr int QT = 64, PITCH = 64, TILE = QT * PITCH;          // one 64-row x 32-col bf16 tile: 4 KB
constexpr int NT = 256;
constexpr int KBW = 128, KBG = 4 * KBW;                         // keys per wave (4 blocks of 32) and per workgroup
constexpr int QSLOT = 2 * TILE + 2 * QT * (int)sizeof(float);   // ring slot: Q tile, dO tile, -lse[64], -delta[64]
constexpr int DSREG = 2 * 32 * PITCH;                           // per wave: two [32 key][32 q] bf16 images of dS (item parity)
constexpr int KIMG = 4 * KBW * PITCH;                           // the workgroup's 512 K rows, natural image: 32 KB (read transposed once, in the prologue)
constexpr int PROW = 36;                                        // floats per row of a partial dQ tile [32 q][32 d] (+4: 16-byte aligned rows, fewer bank conflicts)
constexpr int PTILE = 32 * PROW * (int)sizeof(float);           // 4608 B
constexpr int PSET = 2 * 4 * PTILE;                             // one set of partial tiles: [query block][wave]
constexpr int LDS_DS = 3 * QSLOT, LDS_P = LDS_DS + 4 * DSREG, LDS_TOTAL = LDS_P + 3 * PSET;   // three sets by tile % 3; set 0 holds the K image first
static_assert(KIMG <= PSET && LDS_TOTAL <= 160 * 1024, "LDS map");
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((address_space(3))) s4 *lds_s4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define ACAI_SB() __builtin_amdgcn_sched_barrier(0)
#ifndef ACAI_1P_ABL   // timing ablations (tools/build_variant.sh ... -DACAI_1P_ABL=bits; WRONG results): 1 no barrier in the loop, 2 no atomics, 4 no sums,
#define ACAI_1P_ABL 0 // 8 no flush, 16 no dS image traffic, 32 no statistics reads
#endif

// acc (accumulator registers) += A . B, both operands architectural.  NOP: "s_nop 1" in front (a VALU write - the compiler's tuple copies at region
// edges sit directly in front of an asm statement - needs two wait states before an MFMA reads the register: attn_bwd64w.hip).  The steady-state
// loop goes without: with one wave per SIMD EVERY instruction, an s_nop too, costs a four-cycle issue turn, and there no VALU instruction writes an
// MFMA operand within two wait states of its use - which _asmcheck verifies on every build.
template <bool NOP>
__device__ __forceinline__ void mma_acc(f32x16 &c, const uint4 &a, const uint4 &b) {
    const u32x4 av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
    if constexpr (NOP) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(av), "v"(bv));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(av), "v"(bv));
}
// the dQ products: A = a K^T fragment of an owned block, pinned to the accumulator half like the K / V fragments (dK / dV 128 registers, dQ^T 16,
// K / V fragments 64, K^T fragments 32: 240 of 256); B = a dS^T fragment fresh from LDS
template <bool NOP>
__device__ __forceinline__ void mma_acca(f32x16 &c, const u32x4 &av, const uint4 &b) {
    const u32x4 bv = {b.x, b.y, b.z, b.w};
    if constexpr (NOP) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "a"(av), "v"(bv));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "a"(av), "v"(bv));
}
template <bool NOP>
__device__ __forceinline__ void mma_acca0(f32x16 &c, const u32x4 &av, const uint4 &b) {   // c = A . B (tied operand: the same registers, no copy at the loop edge)
    const u32x4 bv = {b.x, b.y, b.z, b.w};
    if constexpr (NOP) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "+a"(c) : "a"(av), "v"(bv));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "+a"(c) : "a"(av), "v"(bv));
}
// S / dP chains: D and C architectural (the VALU reads them a whole slot later), B - a K / V fragment of an owned block, live for the whole kernel -
// pinned to the accumulator half
__device__ __forceinline__ u32x4 pin_acc(const uint4 &x) {   // a value defined in, and only ever read from, the accumulator half
    u32x4 r = {x.x, x.y, x.z, x.w};
    asm volatile("; pinned %0" : "+a"(r));
    return r;
}
template <bool NOP>
__device__ __forceinline__ void mma_ab0(f32x16 &d, const uint4 &a, const u32x4 &bv, const f32x16 &c) {   // d = A . B + c
    const u32x4 av = {a.x, a.y, a.z, a.w};
    if constexpr (NOP) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(av), "a"(bv), "v"(c));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(av), "a"(bv), "v"(c));
}
template <bool NOP>
__device__ __forceinline__ void mma_ab(f32x16 &d, const uint4 &a, const u32x4 &bv) {   // d += A . B
    const u32x4 av = {a.x, a.y, a.z, a.w};
    if constexpr (NOP) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(av), "a"(bv));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(av), "a"(bv));
}

// The VALU stream of one work item as 48 numbered single-issue operations (attn_bwd64w.hip: dkv_op):
//   e(i): s[i] = P = 2^s[i] (MASK: 0 unless the query row exists); m(i): p[i] = dS = s[i] p[i]; kP(n) / kS(n): bf16 pairs of P / dS
//   e0 e1 | e2 e3 m0 m1 kP0 kS0 | ... | e14 e15 m12 m13 kP6 kS6 | m14 m15 kP7 kS7
template <int OP, bool MASK, bool KMSK>
__device__ __forceinline__ void e_op(f32x16 &s, f32x16 &p, uint32_t (&xp)[8], uint32_t (&xs)[8], int row0, int lh, int rows, uint32_t km) {
    constexpr int r = OP - 2, step = OP < 2 ? 0 : 1 + r / 6, w = OP < 2 ? OP : r % 6;
    if constexpr (step == 0 || (step < 8 && w < 2)) {
        constexpr int i = 2 * step + w;
        float v = fast_exp2(s[i]);
        if constexpr (MASK) v = (row0 + (i & 3) + 8 * (i >> 2) + 4 * lh) < rows ? v : 0.f;
        s[i] = v;
    } else if constexpr (step < 8 ? w < 4 : w < 2) {
        constexpr int i = 2 * (step - 1) + (step < 8 ? w - 2 : w);
        p[i] *= s[i];
    } else if constexpr (step < 8 ? w == 4 : w == 2) {
        constexpr int n = step - 1;
        xp[n] = pack_bf16(s[2 * n], s[2 * n + 1]);
        if constexpr (KMSK) xp[n] &= km;   // (the partial-block body: a lane whose key does not exist contributes P = 0 and dS = 0)
    } else {
        constexpr int n = step - 1;
        xs[n] = pack_bf16(p[2 * n], p[2 * n + 1]);
        if constexpr (KMSK) xs[n] &= km;
    }
}
template <int OP, int END, bool MASK, bool KMSK>
__device__ __forceinline__ void e_ops(f32x16 &s, f32x16 &p, uint32_t (&xp)[8], uint32_t (&xs)[8], int row0, int lh, int rows, uint32_t km) {
    if constexpr (OP < END) {
        e_op<OP, MASK, KMSK>(s, p, xp, xs, row0, lh, rows, km);
        e_ops<OP + 1, END, MASK, KMSK>(s, p, xp, xs, row0, lh, rows, km);
    }
}
// operations of MFMA gap G of a slot, dealt by issue TIME (an exp2 holds the issue port 8 cycles, the rest 4; an MFMA wants its predecessor 32 cycles
// back): 3 3 4 5 5 5 5 5 6 7 - the first gaps also carry the slot's LDS instructions, the last ones nothing else
template <int G, bool MASK, bool KMSK>
__device__ __forceinline__ void e_gap(f32x16 &s, f32x16 &p, uint32_t (&xp)[8], uint32_t (&xs)[8], int row0, int lh, int rows, uint32_t km) {
    constexpr int start[11] = {0, 3, 6, 10, 15, 20, 25, 30, 35, 41, 48};
    e_ops<start[G], start[G + 1], MASK, KMSK>(s, p, xp, xs, row0, lh, rows, km);
}

__device__ __forceinline__ uint4 x4(const uint32_t (&x)[8], int h) { return make_uint4(x[4 * h], x[4 * h + 1], x[4 * h + 2], x[4 * h + 3]); }

typedef std::true_type Y;
typedef std::false_type N;
template <int V> using I = std::integral_constant<int, V>;

union TF { s4 v[2]; uint4 u; };

// ---- delta[h][q] = -sum_d dO[q, d] O[q, d] (the dP accumulators' start value), and the zero fill of the fp32 query-gradient buffer --------
__global__ __launch_bounds__(256) void bwd1p_delta_kernel(BwdArgs a, float *dq32, int rows) {
    const int t = blockIdx.x * 256 + threadIdx.x, sub = t & 3, pair = t >> 2;   // four lanes per (row, head): 16 bytes each
    const int row = pair / a.H, h = pair % a.H;
    if (row >= rows + 64) return;
    float4 *z = reinterpret_cast<float4 *>(dq32 + ((size_t)row * a.H + h) * 32 + sub * 8);
    z[0] = make_float4(0.f, 0.f, 0.f, 0.f);
    z[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row >= rows) return;   // (the workspace's 64 padding rows)
    const uint4 o = *reinterpret_cast<const uint4 *>(reinterpret_cast<const bf16_t *>(a.o) + (size_t)row * a.ldo + h * 32 + sub * 8);
    const uint4 d = *reinterpret_cast<const uint4 *>(reinterpret_cast<const bf16_t *>(a.dout) + (size_t)row * a.lddo + h * 32 + sub * 8);
    const uint32_t ow[4] = {o.x, o.y, o.z, o.w}, dw[4] = {d.x, d.y, d.z, d.w};
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        s = fmaf(__uint_as_float(ow[e] << 16), __uint_as_float(dw[e] << 16), s);
        s = fmaf(__uint_as_float(ow[e] & 0xffff0000u), __uint_as_float(dw[e] & 0xffff0000u), s);
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if (sub == 0) const_cast<float *>(a.delta)[(size_t)h * a.total_q + row] = -s;
}

// ---- dq[q][h * 32 + d] = bf16(scale * dq32[q][h][d]) --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bwd1p_scale_kernel(BwdArgs a, const float *dq32, int rows) {
    const int t = blockIdx.x * 256 + threadIdx.x, sub = t & 3, pair = t >> 2;
    const int row = pair / a.H, h = pair % a.H;
    if (row >= rows) return;
    const float4 *s = reinterpret_cast<const float4 *>(dq32 + ((size_t)row * a.H + h) * 32 + sub * 8);
    const float4 x = s[0], y = s[1];
    uint4 r;
    r.x = pack_bf16(x.x * a.scale, x.y * a.scale);
    r.y = pack_bf16(x.z * a.scale, x.w * a.scale);
    r.z = pack_bf16(y.x * a.scale, y.y * a.scale);
    r.w = pack_bf16(y.z * a.scale, y.w * a.scale);
    *reinterpret_cast<uint4 *>(reinterpret_cast<bf16_t *>(a.dq) + (size_t)row * a.lddq + h * 32 + sub * 8) = r;
}

// One workgroup's work: 512 keys (k0 ...) of sequence b, head h.  KM = false: a FULL block.  KM = true: the keys past a sequence's last full block -
// lanes whose key does not exist read the last key's row and their P / dS packs are ANDed to zero (16 more VALU operations per item; the unmasked
// values never leave the lane's registers).
template <bool KM>
__device__ __forceinline__ void bwd1p_body(const BwdArgs &a, float *dq32, unsigned char *lds, int b, int h, int k0) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int q_start = a.cu_q[b], lq = a.cu_q[b + 1] - q_start;
    const int k_start = a.cu_k[b], lk = a.cu_k[b + 1] - k_start;

    const bf16_t *Q = reinterpret_cast<const bf16_t *>(a.q) + (size_t)q_start * a.ldq + h * 32;
    const bf16_t *K = reinterpret_cast<const bf16_t *>(a.k) + (size_t)k_start * a.ldk + h * 32;
    const bf16_t *V = reinterpret_cast<const bf16_t *>(a.v) + (size_t)k_start * a.ldv + h * 32;
    const bf16_t *DO = reinterpret_cast<const bf16_t *>(a.dout) + (size_t)q_start * a.lddo + h * 32;
    bf16_t *DK = reinterpret_cast<bf16_t *>(a.dk) + (size_t)k_start * a.lddk + h * 32;
    bf16_t *DV = reinterpret_cast<bf16_t *>(a.dv) + (size_t)k_start * a.lddv + h * 32;
    float *DQ32 = dq32 + ((size_t)q_start * a.H + h) * 32;   // [query][H][32]
    const int ldq32 = a.H * 32;
    const int nqt = (lq + QT - 1) / QT;

    // ---- fragment addresses -----------------------------------------------------------------------------------------------------------------------
    int raddr[2];   // row fragments: row lr of a 32-row block, 16-byte chunk 2 s + lh
#pragma unroll
    for (int s = 0; s < 2; ++s) raddr[s] = TL::off(lr, 2 * s + lh);
    // transposed fragment of k-step s2: two 4-row x 16-col transposing reads, rows L and L + 8 with L = 4 lh + (i16 >> 2) (+ 16 s2)
    const int i16 = lane & 15, g1 = (lane >> 4) & 1;
    int taddr[2];
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2) taddr[r2] = TL::off(4 * lh + (i16 >> 2) + 8 * r2, 2 * g1 + ((i16 & 3) >> 1)) + 8 * (i16 & 1);
    auto read_r = [&](const unsigned char *blk, int s) -> uint4 { return *reinterpret_cast<const uint4 *>(blk + raddr[s]); };
    auto read_t = [&](const unsigned char *blk, int s2) -> TF {
        TF f;
        f.v[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(blk + s2 * 16 * PITCH + taddr[0]));
        f.v[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(blk + s2 * 16 * PITCH + taddr[1]));
        return f;
    };

    // The dS images have their own chunk swizzle, f(row) = ((row >> 1) ^ ((row >> 3) & 1)) & 3 (the same for rows 16 apart): the 16-byte writes of a pack (lane = row lr, one chunk
    // column) collide pairwise under the tiles' (row >> 2) & 3 - measured 64 against 52 cycles an instruction with four waves writing
    // (tools/experiments/lds_patterns.hip), SQ_LDS_BANK_CONFLICT 783 cycles a tile - while the transposing reads only need aligned chunk pairs to
    // stay pairs, which any XOR swizzle keeps.
    auto ds_off = [&](int row, int chunk) -> int { return row * PITCH + ((chunk ^ (((row >> 1) ^ ((row >> 3) & 1)) & 3)) << 4); };
    int dsw[2], dst_addr[2];
#pragma unroll
    for (int P = 0; P < 2; ++P) dsw[P] = ds_off(lr, 2 * P + lh);
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2) dst_addr[r2] = ds_off(4 * lh + (i16 >> 2) + 8 * r2, 2 * g1 + ((i16 & 3) >> 1)) + 8 * (i16 & 1);
    auto read_td = [&](const unsigned char *img, int s2) -> TF {
        TF f;
        f.v[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(img + s2 * 16 * PITCH + dst_addr[0]));
        f.v[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(img + s2 * 16 * PITCH + dst_addr[1]));
        return f;
    };

    // ---- lane-owned keys: K / V fragments (B operands of S = Q K^T, dP = dO V^T) --------------------------------------------------------------------
    u32x4 kf[4][2], vf[4][2];
    int kmask = 0;   // KM: bit j = the lane's key of owned block j exists
    unsigned char *kimg = lds + LDS_P + wave * (KBW * PITCH);   // the wave's 128 K rows, natural image, for the prologue only (partial-tile set 0 later)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int key = k0 + wave * KBW + j * 32 + lr;   // (KM = false: < lk, the block is full)
        if constexpr (KM) {
            kmask |= (key < lk ? 1 : 0) << j;
            key = key < lk ? key : lk - 1;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const uint4 kk = *reinterpret_cast<const uint4 *>(K + (size_t)key * a.ldk + s * 16 + lh * 8);
            const uint4 vv = *reinterpret_cast<const uint4 *>(V + (size_t)key * a.ldv + s * 16 + lh * 8);
            *reinterpret_cast<uint4 *>(kimg + j * 32 * PITCH + raddr[s]) = kk;
            kf[j][s] = pin_acc(kk);
            vf[j][s] = pin_acc(vv);
        }
    }
    // K^T fragments (A operand of dQ^T = K^T dS^T) of the owned blocks: the transposing read of that image (wave-private: the wave's own LDS
    // operations complete in order - no barrier between the stores and these reads)
    u32x4 ktf[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) ktf[j][s2] = pin_acc(read_t(kimg + j * 32 * PITCH, s2).u);

    // ---- staging of the Q / dO tiles (one 16-byte chunk of each per thread) and their statistics (wave 0 / 2: -lse, wave 1 / 3: -delta, lane = row;
    // the two pairs store the same values).  Buffer loads: rows past the sequence's end read as zero. ----------------------------------------------
    const int srow = tid >> 2, scc = tid & 3, soff = TL::off(srow, scc);
    const uint32_t gq = (uint32_t)(srow * a.ldq * 2 + scc * 16), gd = (uint32_t)(srow * a.lddo * 2 + scc * 16);
    const __amdgpu_buffer_rsrc_t rQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(Q), 0, lq * a.ldq * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(DO), 0, lq * a.lddo * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rS =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(((wave & 1) ? a.delta : a.lse) + (size_t)h * a.total_q + q_start), 0, lq * 4, 0x00020000);
    const float stat_sign = (wave & 1) ? 1.f : -1.f;   // (delta arrives negated)
    const int stat_off = 2 * TILE + ((wave & 1) * QT + lane) * 4;
    u32x4 rq, rd;
    float r_stat = 0.f;
    auto load_tile = [&](int t) {
        rq = __builtin_amdgcn_raw_buffer_load_b128(rQ, gq + (uint32_t)(t * QT * a.ldq * 2), 0, 0);
        rd = __builtin_amdgcn_raw_buffer_load_b128(rD, gd + (uint32_t)(t * QT * a.lddo * 2), 0, 0);
        r_stat = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rS, (uint32_t)((t * QT + lane) * 4), 0, 0));
    };
    auto store_tile = [&](int slot) {
        *reinterpret_cast<u32x4 *>(lds + slot + soff) = rq;
        *reinterpret_cast<u32x4 *>(lds + slot + TILE + soff) = rd;
        *reinterpret_cast<float *>(lds + slot + stat_off) = r_stat * stat_sign;
    };
    auto read_init = [&](f32x16 &acc, int stat, int g4) {   // (constant register indices on every path)
        const f32x4 v = *reinterpret_cast<const f32x4 *>(lds + stat + (8 * g4 + 4 * lh) * 4);
        switch (g4) {
            case 0: acc[0] = v[0]; acc[1] = v[1]; acc[2] = v[2]; acc[3] = v[3]; break;
            case 1: acc[4] = v[0]; acc[5] = v[1]; acc[6] = v[2]; acc[7] = v[3]; break;
            case 2: acc[8] = v[0]; acc[9] = v[1]; acc[10] = v[2]; acc[11] = v[3]; break;
            default: acc[12] = v[0]; acc[13] = v[1]; acc[14] = v[2]; acc[15] = v[3]; break;
        }
    };

    f32x16 dk[4], dv[4];      // [owned block], accumulator registers
    f32x16 dqa;               // partial dQ^T of the 32-query block in flight, accumulator registers
    f32x16 sc[2], dp[2];      // by item parity
    uint32_t xp[2][8], xs[2][8];
    f32x16 nl, nd;            // -lse / -delta of the query block whose S / dP chains are being issued
    uint4 qr[2], dor[2];      // its row fragments
    TF qt[2], dot[2];         // transposed fragments [k-step] of the query block whose gradient products are being issued
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) dk[j][e] = dv[j][e] = 0.f;

    unsigned char *dsreg = lds + LDS_DS + wave * DSREG;

    // ---- the query gradient.  A wave's partial dQ^T tile of a 32-query block (its 128 keys) goes to LDS as [32 q][32 d] fp32 ("flush"); the four
    // waves' tiles are summed and added to the fp32 gradient with no-return float atomics, lanes along d: one instruction adds two 128-byte row
    // segments.  Three buffer sets by tile % 3: tile t's tiles are complete at tile t+1's closing barrier (the second query block's flush is in tile
    // t+1's second slot), are added during tile t+2's first slot, and are overwritten from tile t+3 on - one barrier per tile serves the ring and this.
    const int qcol = (lr & 19) | ((lr & 4) << 1) | ((lr & 8) >> 1);     // the query behind column lr of the dS image (bits 2 and 3 swapped)
    const int pflush_v = wave * PTILE + (qcol * PROW + 4 * lh) * 4;     // flush: lane = query qcol, registers 4 g4 + i = d 8 g4 + 4 lh + i
    // sum: lane = d lr of query rows wave + 8 lh (+ 4 (k & 1) + 16 (k >> 1)): the two half-waves read rows 8 apart = 32 banks apart (PROW = 36)
    const int pred_v = ((wave + 8 * lh) * PROW + lr) * 4;
    const int row4 = 4 * ldq32 * 4;                                     // bytes between gradient rows q and q + 4
    const uint32_t avoff = (uint32_t)(((wave + 8 * lh) * ldq32 + lr) * 4);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(DQ32, 0, lq > 0 ? ((lq - 1) * ldq32 + 32) * 4 : 0, 0x00020000);
    float ra[8][4];   // partial-tile elements between their LDS read and their add (a few gaps: the register allocator sees short live ranges)
    auto flush_piece = [&](int pset_qb, int g4) {   // (constant g4 at every call)
        if constexpr (ACAI_1P_ABL & 8) return;
        unsigned char *pt = lds + pset_qb + pflush_v + 32 * g4;
        switch (g4) {
            case 0: *reinterpret_cast<f32x4 *>(pt) = f32x4{dqa[0], dqa[1], dqa[2], dqa[3]}; break;
            case 1: *reinterpret_cast<f32x4 *>(pt) = f32x4{dqa[4], dqa[5], dqa[6], dqa[7]}; break;
            case 2: *reinterpret_cast<f32x4 *>(pt) = f32x4{dqa[8], dqa[9], dqa[10], dqa[11]}; break;
            default: *reinterpret_cast<f32x4 *>(pt) = f32x4{dqa[12], dqa[13], dqa[14], dqa[15]}; break;
        }
    };
    auto flush_tile = [&](int pset_qb) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) flush_piece(pset_qb, g4);
    };
    auto reduce_plain = [&](int t, int pset) {   // outside the steady state: rows past the sequence's end are dropped by the buffer's range check
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const int rown = 4 * (n & 1) + 16 * ((n >> 1) & 1);
            const unsigned char *pp = lds + pset + pred_v + (n >> 2) * 4 * PTILE + rown * PROW * 4;
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) sum += *reinterpret_cast<const float *>(pp + w * PTILE);
            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(sum, rA, avoff + (uint32_t)((t * QT + (n >> 2) * 32 + rown) * ldq32 * 4), 0, 0);
        }
    };

    // ---- one slot: E(item w) | A(item w+1) | C(item w-1), ten MFMA gaps ------------------------------------------------------------------------------
    //   jA / jC: owned blocks of items w+1 / w-1;  pE = w & 1: register parity of item w (A writes !pE, C reads !pE) - COMPILE-TIME constants
    //   LOADROW: row fragments / start values of the next query block (LDS offsets nextrow / nextstat) replace the current ones, each behind A's last use
    //   LOADT: the transposed fragments of the query block at LDS offset tblk replace the current ones, each behind C's last use of it
    //   FLUSH: the finished partial dQ^T tile goes to LDS (offset pflush: set and query block) in front of gap 7, where item w-1 - the first owned
    //   block of the NEXT query block - starts the accumulator afresh
    //   REDUCE: the eight elements per thread of the partial tiles in set pred are summed and added to the gradient rows at byte offset arow: element n
    //   is read in gap n of the tile's first slot and added four gaps later (the last two in the second slot's first gaps)
    auto slot = [&](auto hasE_, auto hasA_, auto hasC_, auto mask_, auto loadrow_, auto loadt_, auto flush_, auto reduce_, auto pE_, auto jA_, auto jC_,
                    int rowE, int nextrow, int nextstat, int tblk, int pflush, int pred, int arow) __attribute__((always_inline)) {
        constexpr bool HE = decltype(hasE_)::value, HA = decltype(hasA_)::value, HC = decltype(hasC_)::value, M = decltype(mask_)::value;
        constexpr bool LOADROW = decltype(loadrow_)::value, LOADT = decltype(loadt_)::value, FLUSH = decltype(flush_)::value;
        constexpr int REDUCE = decltype(reduce_)::value;
        constexpr int pE = decltype(pE_)::value, jA = decltype(jA_)::value, jC = decltype(jC_)::value;
        constexpr bool NP = M || !HE || !HA || !HC;
   // outside the steady-state loop the compiler's tuple copies at region edges may sit in front of an MFMA: pad
        const uint32_t kmE = KM ? 0u - ((kmask >> ((jA + 3) & 3)) & 1u) : ~0u;   // all ones unless the lane's key of item w's owned block (the one before A's) does not exist
        f32x16 &sE = sc[pE], &pEd = dp[pE], &sA = sc[pE ^ 1], &pA = dp[pE ^ 1];
        uint32_t(&xpE)[8] = xp[pE], (&xsE)[8] = xs[pE], (&xpC)[8] = xp[pE ^ 1], (&xsC)[8] = xs[pE ^ 1];
        unsigned char *dsimg = dsreg + (pE ^ 1) * (32 * PITCH);
        const unsigned char *rowp = lds + nextrow, *tp = lds + tblk;
        TF dst[2];   // dS^T fragments of item w-1
        auto red = [&](auto g_) __attribute__((always_inline)) {   // REDUCE: 1 = the tile's first slot, 2 = its second (the last two elements' adds)
            constexpr int g = decltype(g_)::value;
            if constexpr (ACAI_1P_ABL & 4) return;
            if constexpr (REDUCE == 1 && g < 8) {
                const unsigned char *pp = lds + pred + pred_v + (g >> 2) * 4 * PTILE + (4 * (g & 1) + 16 * ((g >> 1) & 1)) * PROW * 4;
#pragma unroll
                for (int w = 0; w < 4; ++w) ra[g][w] = *reinterpret_cast<const float *>(pp + w * PTILE);
            }
            constexpr int n = REDUCE == 1 ? g - 4 : g + 6;
            if constexpr ((REDUCE == 1 && g >= 4) || (REDUCE == 2 && g < 2)) {
                const float sum = (ra[n][0] + ra[n][1]) + (ra[n][2] + ra[n][3]);
                if constexpr (ACAI_1P_ABL & 2) asm volatile("" ::"v"(sum));
                else __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(sum, rA, avoff, arow + (8 * (n >> 2) + (n & 1) + 4 * ((n >> 1) & 1)) * row4, 0);
            }
        };
        // gap 0: A, S k-step 0 | the second half of the dS image of item w-1 -> LDS (the first went out in its own E phase, gap 6: LDS writes move
        // ~36 bytes a clock per wave, two in a row stall the wave's next LDS instruction).  Lane = key: its 32 q values as the two 16-byte packs the
        // dK product reads - q in the order 16 P + 8 gl + 4 lh + i at column 16 P + 8 lh + 4 gl + i (bits 2 and 3 swapped), which the flush undoes
        if constexpr (HA) mma_ab0<NP>(sA, qr[0], kf[jA][0], nl);
        ACAI_SB();
        if constexpr (HE) e_gap<0, M, KM>(sE, pEd, xpE, xsE, rowE, lh, lq, kmE);
        if constexpr (HC && !(ACAI_1P_ABL & 16)) *reinterpret_cast<uint4 *>(dsimg + dsw[1]) = x4(xsC, 1);
        if constexpr (LOADROW) qr[0] = read_r(rowp, 0);
        red(I<0>{});
        ACAI_SB();
        // gap 1: C, dV k-step 0
        if constexpr (HC) mma_acc<NP>(dv[jC], dot[0].u, x4(xpC, 0));
        ACAI_SB();
        if constexpr (HE) e_gap<1, M, KM>(sE, pEd, xpE, xsE, rowE, lh, lq, kmE);
        if constexpr (HC && !(ACAI_1P_ABL & 16)) dst[0] = read_td(dsimg, 0);
        if constexpr (HC && (ACAI_1P_ABL & 16)) dst[0].u = make_uint4(xsC[0], xsC[1], xsC[2], xsC[3]);
        if constexpr (LOADROW) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                if constexpr (!(ACAI_1P_ABL & 32)) read_init(nl, nextstat, g4);
        }
        if constexpr (LOADT) dot[0] = read_t(tp + TILE, 0);
        red(I<1>{});
        ACAI_SB();
        // gap 2: A, dP k-step 0
        if constexpr (HA) mma_ab0<NP>(pA, dor[0], vf[jA][0], nd);
        ACAI_SB();
        if constexpr (HE) e_gap<2, M, KM>(sE, pEd, xpE, xsE, rowE, lh, lq, kmE);
        if constexpr (HC && !(ACAI_1P_ABL & 16)) dst[1] = read_td(dsimg, 1);
        if constexpr (HC && (ACAI_1P_ABL & 16)) dst[1].u = make_uint4(xsC[4], xsC[5], xsC[6], xsC[7]);
        if constexpr (LOADROW) dor[0] = read_r(rowp + TILE, 0);
        if constexpr (FLUSH) flush_piece(pflush, 0);   // (the block's last dQ MFMA is three MFMAs back; the next one that writes dqa is gap 7's)
        red(I<2>{});
        ACAI_SB();
        // gap 3: C, dV k-step 1
        if constexpr (HC) mma_acc<NP>(dv[jC], dot[1].u, x4(xpC, 1));
        ACAI_SB();
        if constexpr (HE) e_gap<3, M, KM>(sE, pEd, xpE, xsE, rowE, lh, lq, kmE);
        if constexpr (LOADROW) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                if constexpr (!(ACAI_1P_ABL & 32)) read_init(nd, nextstat + QT * 4, g4);
        }
        if constexpr (LOADT) dot[1] = read_t(tp + TILE, 1);
        if constexpr (FLUSH) flush_piece(pflush, 1);
        red(I<3>{});
        ACAI_SB();
        // gap 4: A, S k-step 1
        if constexpr (HA) mma_ab<NP>(sA, qr[1], kf[jA][1]);
        ACAI_SB();
        if constexpr (HE) e_gap<4, M, KM>(sE, pEd, xpE, xsE, rowE, lh, lq, kmE);
        if constexpr (LOADROW) qr[1] = read_r(rowp, 1);
        if constexpr (FLUSH) flush_piece(pflush, 2);
        red(I<4>{});
        ACAI_SB();
        // gap 5: C, dK k-step 0
        if constexpr (HC) mma_acc<NP>(dk[jC], qt[0].u, x4(xsC, 0));
        ACAI_SB();
        if constexpr (HE) e_gap<5, M, KM>(sE, pEd, xpE, xsE, rowE, lh, lq, kmE);
        if constexpr (LOADT) qt[0] = read_t(tp, 0);
        if constexpr (FLUSH) flush_piece(pflush, 3);
        red(I<5>{});
        ACAI_SB();
        // gap 6: A, dP k-step 1
        if constexpr (HA) mma_ab<NP>(pA, dor[1], vf[jA][1]);
        ACAI_SB();
        if constexpr (HE) e_gap<6, M, KM>(sE, pEd, xpE, xsE, rowE, lh, lq, kmE);
        if constexpr (HE && !(ACAI_1P_ABL & 16)) *reinterpret_cast<uint4 *>(dsreg + pE * (32 * PITCH) + dsw[0]) = x4(xsE, 0);   // (its last pair is packed first thing in this gap)
        if constexpr (LOADROW) dor[1] = read_r(rowp + TILE, 1);
        red(I<6>{});
        ACAI_SB();
        // gap 7: C, dQ k-step 0 (the first owned block of a query block starts the partial tile)
        if constexpr (HC) {
            if constexpr (jC == 0) mma_acca0<NP>(dqa, ktf[jC][0], dst[0].u);
            else mma_acca<NP>(dqa, ktf[jC][0], dst[0].u);
        }
        ACAI_SB();
        if constexpr (HE) e_gap<7, M, KM>(sE, pEd, xpE, xsE, rowE, lh, lq, kmE);
        red(I<7>{});
        ACAI_SB();
        // gap 8: C, dK k-step 1 (between the two dQ k-steps: a dependent MFMA directly behind its producer waits for the whole pass)
        if constexpr (HC) mma_acc<NP>(dk[jC], qt[1].u, x4(xsC, 1));
        ACAI_SB();
        if constexpr (HE) e_gap<8, M, KM>(sE, pEd, xpE, xsE, rowE, lh, lq, kmE);
        if constexpr (LOADT) qt[1] = read_t(tp, 1);
        red(I<8>{});
        ACAI_SB();
        // gap 9: C, dQ k-step 1
        if constexpr (HC) mma_acca<NP>(dqa, ktf[jC][1], dst[1].u);
        ACAI_SB();
        if constexpr (HE) e_gap<9, M, KM>(sE, pEd, xpE, xsE, rowE, lh, lq, kmE);
        red(I<9>{});
        ACAI_SB();
    };

    if (nqt > 0) {
        // ---- prologue: partial-tile sets 1 and 2 zeroed (the first two tiles "add" the tiles of tiles -2 and -1 to the first rows: zeros); tiles 0
        // and 1 into the ring; fragments and start values of query block 0 ----------------------------------------------------------------------------
#pragma unroll
        for (int e = 0; e < 16; ++e) dqa[e] = 0.f;
#pragma unroll
        for (int sq = 2; sq < 6; ++sq) flush_tile(LDS_P + sq * 4 * PTILE);
        load_tile(0);
        store_tile(0);
        load_tile(1);
        store_tile(QSLOT);
        __syncthreads();
        {
            const unsigned char *s0 = lds;
            qr[0] = read_r(s0, 0); qr[1] = read_r(s0, 1);
            dor[0] = read_r(s0 + TILE, 0); dor[1] = read_r(s0 + TILE, 1);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                read_init(nl, 2 * TILE, g4);
                read_init(nd, 2 * TILE + QT * 4, g4);
            }
            qt[0] = read_t(s0, 0); qt[1] = read_t(s0, 1);
            dot[0] = read_t(s0 + TILE, 0); dot[1] = read_t(s0 + TILE, 1);
        }
        // item -1: only A(item 0).  Its "gradient products" in the first real slot multiply ZERO packs, and the partial tile it "finishes" is zero
#pragma unroll
        for (int n = 0; n < 8; ++n) xp[1][n] = xs[1][n] = 0u;
        *reinterpret_cast<uint4 *>(dsreg + 32 * PITCH + dsw[0]) = make_uint4(0u, 0u, 0u, 0u);   // (the half of item -1's dS image its E phase would have written)
        slot(N{}, Y{}, N{}, N{}, N{}, N{}, N{}, I<0>{}, I<1>{}, I<0>{}, I<0>{}, 0, 0, 0, 0, 0, 0, 0);

        // One tile = 8 items (query blocks qb = 0, 1 x owned blocks j = 0..3).  Slot of item (qb, j): E(qb, j) | A(next item) | C(previous item).
        //   (0, 0): the partial tiles of tile t-2 are added to the gradient
        //   (qb, 2): A issues (qb, 3), the block's last: the row fragments / start values of the next query block follow
        //   (qb', 0): C issues the previous block's (., 3), the last user of its transposed fragments: the new block's follow
        //   (qb', 1): C issues (qb', 0), which starts a new partial dQ tile: the finished one of the previous block is flushed first
        int o0 = 0, o1 = QSLOT, o2 = 2 * QSLOT;                              // ring slots of tiles t, t+1, t+2
        int p0 = LDS_P, p1 = LDS_P + PSET, p2 = LDS_P + 2 * PSET;            // partial-tile sets of tiles t, t+1 (= t-2), t+2 (= t-1)
        auto tile = [&](auto mask_, int t) __attribute__((always_inline)) {
            const int r0 = t * QT, arow = (t >= 2 ? t - 2 : 0) * (QT / 4) * row4;
            load_tile(t + 2);
            slot(Y{}, Y{}, Y{}, mask_, N{}, Y{}, N{}, I<1>{}, I<0>{}, I<1>{}, I<3>{}, r0, 0, 0, o0, 0, p1, arow);                                       // (0, 0)
            slot(Y{}, Y{}, Y{}, mask_, N{}, N{}, Y{}, I<2>{}, I<1>{}, I<2>{}, I<0>{}, r0, 0, 0, 0, p2 + 4 * PTILE, 0, arow);                               // (0, 1)
            slot(Y{}, Y{}, Y{}, mask_, Y{}, N{}, N{}, I<0>{}, I<0>{}, I<3>{}, I<1>{}, r0, o0 + 32 * PITCH, o0 + 2 * TILE + 32 * 4, 0, 0, 0, 0);         // (0, 2)
            slot(Y{}, Y{}, Y{}, mask_, N{}, N{}, N{}, I<0>{}, I<1>{}, I<0>{}, I<2>{}, r0, 0, 0, 0, 0, 0, 0);                                            // (0, 3)
            slot(Y{}, Y{}, Y{}, mask_, N{}, Y{}, N{}, I<0>{}, I<0>{}, I<1>{}, I<3>{}, r0 + 32, 0, 0, o0 + 32 * PITCH, 0, 0, 0);                         // (1, 0)
            slot(Y{}, Y{}, Y{}, mask_, N{}, N{}, Y{}, I<0>{}, I<1>{}, I<2>{}, I<0>{}, r0 + 32, 0, 0, 0, p0, 0, 0);                                      // (1, 1)
            store_tile(o2);
            slot(Y{}, Y{}, Y{}, mask_, Y{}, N{}, N{}, I<0>{}, I<0>{}, I<3>{}, I<1>{}, r0 + 32, o1, o1 + 2 * TILE, 0, 0, 0, 0);                          // (1, 2)
            slot(Y{}, Y{}, Y{}, mask_, N{}, N{}, N{}, I<0>{}, I<1>{}, I<0>{}, I<2>{}, r0 + 32, 0, 0, 0, 0, 0, 0);                                       // (1, 3)
            if constexpr (!(ACAI_1P_ABL & 1)) __syncthreads();
            const int ot = o0, pt = p0;
            o0 = o1; o1 = o2; o2 = ot;
            p0 = p1; p1 = p2; p2 = pt;
        };
        // (the next tile's first query block is read during this tile's last slots and the next tile's first: it was stored during tile t-1 and
        // published by that tile's barrier; tile t+2 is stored in this tile's sixth slot into the slot tile t-1 left; past the last tile the ring holds
        // zeros - buffer loads beyond the end - and the extra A of the last slot feeds nothing)
        int t = 0;
        for (; t + 1 < nqt; ++t) tile(N{}, t);
        tile(Y{}, t);   // the last tile: query rows past the end give probability zero
        // ---- drain: C of the last item, the flush of the last query block, the last two tiles' sums (after the rotation: p2 = the last tile's set,
        // p1 = the one before) ---------------------------------------------------------------------------------------------------------------------
        slot(N{}, N{}, Y{}, N{}, N{}, N{}, N{}, I<0>{}, I<0>{}, I<0>{}, I<3>{}, 0, 0, 0, 0, 0, 0, 0);
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // (the last dQ MFMA was issued just above: let it write back before the flush reads)
        flush_tile(p2 + 4 * PTILE);
        __syncthreads();
        if (nqt >= 2) reduce_plain(nqt - 2, p1);
        reduce_plain(nqt - 1, p2);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // (asm MFMAs: no compiler-tracked hazard in front of the accumulator reads below)

    const float ksc = 0.6931471805599453f;   // dK = dS^T Q = dS^T Q' sqrt(d_h) / log2(e), times 1 / sqrt(d_h)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int key = k0 + wave * KBW + j * 32 + lr;
        if (KM && key >= lk) continue;
        bf16_t *rk = DK + (size_t)key * a.lddk, *rv = DV + (size_t)key * a.lddv;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            uint2 pk, pv;
            pk.x = pack_bf16(dk[j][4 * g4 + 0] * ksc, dk[j][4 * g4 + 1] * ksc);
            pk.y = pack_bf16(dk[j][4 * g4 + 2] * ksc, dk[j][4 * g4 + 3] * ksc);
            pv.x = pack_bf16(dv[j][4 * g4 + 0], dv[j][4 * g4 + 1]);
            pv.y = pack_bf16(dv[j][4 * g4 + 2], dv[j][4 * g4 + 3]);
            *reinterpret_cast<uint2 *>(rk + 8 * g4 + 4 * lh) = pk;
            *reinterpret_cast<uint2 *>(rv + 8 * g4 + 4 * lh) = pv;
        }
    }
}

// Grid: slots x H x B with slots = max_k / 512 (+ 1 when a sequence may end inside a block), the key-block slot fastest: the workgroups of one
// (sequence, head) run together and stream the same Q / dO tiles.  tail256 != 0 (equal-length batch): the XCD-aware block order of attn_fwd64w.hip on
// top - each XCD takes a contiguous eighth of the (sequence, head) pairs and its L2 serves the shared tiles.  On a RAGGED batch that static split
// leaves the XCD with the longest sequences working alone at the end, so the plain order is used.  Measured (one box): equal-length 32 x 16 x 4096^2
// 2.97-3.00 ms with the XCD order, 3.10 without; 16 sequences of 1024 ... 9216 tokens 2.53 ms plain, 3.47 with the XCD order; and with the SEQUENCE
// index fastest and the slots descending (longest workgroups first) 5.95 / 3.21 ms - the workgroups of a (sequence, head) must run together.
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_bwd1p_kernel(BwdArgs a, float *dq32) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    int vid = blockIdx.x;
    if (a.tail256) {
        const int per = gridDim.x >> 3;
        if (vid < (per << 3)) vid = (vid & 7) * per + (vid >> 3);
    }
    const int kb = vid % a.nblk, h = (vid / a.nblk) % a.H, b = vid / (a.nblk * a.H);
    const int lk = a.cu_k[b + 1] - a.cu_k[b], nfull = lk / KBG;
    if (kb < nfull) bwd1p_body<false>(a, dq32, lds, b, h, kb * KBG);
    else if (kb == nfull && lk > nfull * KBG) bwd1p_body<true>(a, dq32, lds, b, h, nfull * KBG);
}

}  // namespace

// (64 rows more than the gradient: a workgroup's first two tiles "add" the zero-filled partial sets to the sequence's first 64 rows without a range
// check - for a sequence shorter than that, or the last one, those adds of 0.0 land in the next sequence's rows or in the padding)
size_t acai_attn_bwd1p_workspace(int total_q, int H) { return ((size_t)total_q + 64) * H * 32 * sizeof(float); }

// partial_blocks: 0 = the host knows every sequence is a whole number of 512-key blocks (no slot for the keys past the last full block);
// equal_len: the host knows all sequences have max_q queries and max_k keys (the XCD-aware block order)
void acai_attn_bwd1p_launch(const BwdArgs &a, int B, int max_k, int partial_blocks, int equal_len, void *workspace, hipStream_t st) {
    BwdArgs w = a;
    w.nblk = max_k / KBG + (partial_blocks ? 1 : 0);
    w.tail256 = equal_len ? 1 : 0;   // (this kernel's use of the field: the XCD-aware block order, for batches known to be equal-length)
    float *dq32 = reinterpret_cast<float *>(workspace);
    static bool attr[ACAI_MAX_DEV] = {};
    if (acai_first_on_device(attr))
        hipFuncSetAttribute(reinterpret_cast<const void *>(attn_bwd1p_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    const int pairs4 = (a.total_q + 64) * a.H * 4;
    hipLaunchKernelGGL(bwd1p_delta_kernel, dim3(cdiv(pairs4, 256)), dim3(256), 0, st, w, dq32, a.total_q);
    hipLaunchKernelGGL(attn_bwd1p_kernel, dim3(w.nblk * a.H * B), dim3(NT), LDS_TOTAL, st, w, dq32);
    hipLaunchKernelGGL(bwd1p_scale_kernel, dim3(cdiv(a.total_q * a.H * 4, 256)), dim3(256), 0, st, w, dq32, a.total_q);
}
