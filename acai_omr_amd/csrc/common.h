// Shared device/host helpers for the gfx950 kernels.  CDNA4 only: wave = 64 lanes, MFMA 32x32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/acai_omr_hip.h"

typedef uint16_t bf16_t;  // raw bf16 storage
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

extern char g_acai_err[512];
int acai_set_err(int code, const char *fmt, ...);

#define ACAI_CHECK_ARG(cond, ...)                       \
    do {                                                \
        if (!(cond)) return acai_set_err(-1, __VA_ARGS__); \
    } while (0)

#define ACAI_LAUNCH_CHECK(name)                                                        \
    do {                                                                               \
        hipError_t e_ = hipGetLastError();                                             \
        if (e_ != hipSuccess) return acai_set_err((int)e_, "%s: %s", name, hipGetErrorString(e_)); \
    } while (0)

// ---- bf16 <-> f32 (round-to-nearest-even; plain casts keep NaN a NaN on gfx950) --------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ float round_bf16(float f) { return bf2f(f2bf(f)); }
typedef __attribute__((ext_vector_type(2))) float f32x2;
// two floats -> one dword of bf16 (RNE) with a single v_cvt_pk_bf16_f32 (separate scalar casts cost 2 cvt + shift + or)
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

// 2^x as the bare v_exp_f32 (exp2f() adds a denormal-range fix-up of ~4 VALU ops per call; softmax arguments are <= 0 and
// results below 2^-126 may flush to zero)
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Counter-based dropout RNG: a 32-bit avalanche hash of (seed, row, col), so that any kernel can regenerate the keep mask of an
// element in any thread layout (forward: key rows in registers; dK/dV backward: query rows in registers).  ~10 integer ops / element.
__device__ __forceinline__ uint32_t drop_hash(uint32_t seed, uint32_t row, uint32_t col) {
    uint32_t x = (row * 0x9E3779B1u) ^ (col * 0x85EBCA77u) ^ seed;
    x ^= x >> 15; x *= 0x2C1B3C6Du;
    x ^= x >> 12; x *= 0x297A2D39u;
    x ^= x >> 15;
    return x;
}
// keep probability 1 - p as a 32-bit threshold: keep iff hash >= thr
__device__ __forceinline__ bool drop_keep(uint32_t seed, uint32_t row, uint32_t col, uint32_t thr) { return drop_hash(seed, row, col) >= thr; }

// streamed-once 16-byte load (non-temporal: does not displace reusable lines in L2 / Infinity Cache)
__device__ __forceinline__ uint4 ld_nt16(const void *p) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(v[0], v[1], v[2], v[3]);
}

// LDS image of a [rows][DHP] attention operand tile (K, V, Q, dO) in its natural row-major order.
//   bf16: unpadded rows (64 / 128 bytes) with an XOR swizzle of the 16-byte chunk index by row bits, chosen so that BOTH access patterns are
//         bank-conflict free: ds_read_b128 of one chunk column by 16 consecutive rows, and ds_read_b64_tr_b16 of a 4-row x 64-byte block.
//         (The padded pitch used before cost 25-38 % of the LDS cycles in conflicts on the transposing reads - PMC SQ_LDS_BANK_CONFLICT.)
//   fp32: rows padded by 16 bytes (the K=2 MFMA path reads single floats down a column).
template <int ES, int DHP>
struct TileLayout {
    static constexpr int PITCH = ES == 2 ? DHP * 2 : DHP * 4 + 16;
    __device__ static __forceinline__ int swz(int row) {
        if constexpr (ES != 2) return 0;
        else if constexpr (DHP == 32) return (row >> 2) & 3;
        else return (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
    }
    // byte offset of 16-byte chunk `chunk` of row `row`
    __device__ static __forceinline__ int off(int row, int chunk) { return row * PITCH + ((chunk ^ swz(row)) << 4); }
};

// GELU (exact-erf form, torch's F.gelu(approximate="none")) and its derivative through ONE function: h(x) = 0.5 erfc(|x| / sqrt 2) = Phi(-|x|),
// computed as 2^(t q(t) - 1), t = |x| / sqrt 2, q a degree-5 polynomial fitted (weighted minimax, weight erfc(t) t ln 2, i.e. the absolute error
// of erf) to log2(erfc(t)) / t on [0, 4.3]: |error of erf| <= 2.4e-7 in exact arithmetic.  Then
//     gelu(x)  = max(x, 0) - |x| h        (x >= 0: x - x h = x Phi(x);  x < 0: x h = x Phi(x))
//     gelu'(x) = Phi(x) + x phi(x),  Phi(x) = x < 0 ? h : 1 - h
// Branch-free and single-path: 2 + 6 FMAs + one exp2 + 2 = 11 vector instructions per GELU.  (Rounds 1-2 evaluated erf to 6e-8 through two
// polynomial pieces and a select - 26 instructions per GELU; the GELU epilogues of the MLP GEMMs were VALU-bound on it: 402 M elements per
// decoder layer.)  Measured against the exact function (tools/check_erf.py, fp32 emulation): max |error| of gelu 5.0e-7 over [-8, 8] (the
// two-piece form: 3.8e-7); over ALL finite bf16 inputs the bf16-rounded result differs from the correctly rounded one for 133 inputs in
// |x| < 8, every one of them below -3.5 where |gelu| < 1e-3 (two-piece form: 108).  t is clamped at 12: the polynomial is monotone up to there
// (2^-258 flushes to zero), beyond it would turn upward.
__device__ __forceinline__ float acai_half_erfc_abs(float x) {
    // (v_med3_f32, not fminf: the IEEE minimum first quiets its operand with an extra v_max per element)
    const float t = __builtin_amdgcn_fmed3f(fabsf(x) * 0.70710678118654752440f, 0.0f, 12.0f);
    float q = 1.420412202e-04f;
    q = fmaf(q, t, -3.664269981e-03f);
    q = fmaf(q, t, 3.089617088e-02f);
    q = fmaf(q, t, -1.496994187e-01f);
    q = fmaf(q, t, -9.181654744e-01f);
    q = fmaf(q, t, -1.627925070e+00f);
    return __builtin_amdgcn_exp2f(fmaf(q, t, -1.0f));
}

// exact-erf GELU, as torch's F.gelu(approximate="none")
// (max(x, 0) on the bit pattern - v_max_i32: a negative float is a negative integer; fmaxf first quiets an operand it cannot prove canonical -
// every widened bf16 - with one more v_max per element, and v_med3(x, 0, inf) is folded back into that pair)
__device__ __forceinline__ float gelu_erf(float x) {
    return fmaf(-fabsf(x), acai_half_erfc_abs(x), __int_as_float(max(__float_as_int(x), 0)));
}

// d/dx of the exact-erf GELU: Phi(x) + x exp(-x^2 / 2) / sqrt(2 pi)   (log2(1 / sqrt(2 pi)) = -1.3257480647)
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float h = acai_half_erfc_abs(x);
    const float phi = x < 0.0f ? h : 1.0f - h;
    return fmaf(x, __builtin_amdgcn_exp2f(fmaf(x * x, -0.72134752044448170368f, -1.3257480647361593f)), phi);
}

// GELU and its derivative of the same argument through ONE h = Phi(-|x|): the forward epilogue of linear1 that also keeps gelu'(a) for the
// backward pass (aux_mode 3) pays one more exp2 + 5 VALU on top of the GELU itself instead of a second polynomial.
__device__ __forceinline__ void gelu_erf_both(float x, float &g, float &dg) {
    const float h = acai_half_erfc_abs(x);
    g = fmaf(-fabsf(x), h, __int_as_float(max(__float_as_int(x), 0)));
    const float phi = x < 0.0f ? h : 1.0f - h;
    dg = fmaf(x, __builtin_amdgcn_exp2f(fmaf(x * x, -0.72134752044448170368f, -1.3257480647361593f)), phi);
}

template <typename T> struct DT;
template <> struct DT<float> {
    static constexpr int id = ACAI_F32;
    __device__ static __forceinline__ float ld(const float *p) { return *p; }
    __device__ static __forceinline__ void st(float *p, float v) { *p = v; }
};
template <> struct DT<bf16_t> {
    static constexpr int id = ACAI_BF16;
    __device__ static __forceinline__ float ld(const bf16_t *p) { return bf2f(*p); }
    __device__ static __forceinline__ void st(bf16_t *p, float v) { *p = f2bf(v); }
};

// Wave-wide sum on the DPP network (no LDS crossbar): quad swaps, half-row and row mirrors give every lane its 16-lane row's sum, two
// row broadcasts carry it across the four rows into lane 63, one v_readlane hands it to every lane through an SGPR.  7 dependent VALU
// steps of a few cycles each instead of 6 ds_bpermute round trips (~100 cycles each) - the LayerNorm-on-load of the decode GEMVs sits
// on the step's critical path 37 times per token.
__device__ __forceinline__ float wave_sum_dpp(float v) {
#define ACAI_DPP_ADD(CTRL, ROWMASK) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xF, false))
    ACAI_DPP_ADD(0xB1, 0xF);    // quad_perm [1,0,3,2]
    ACAI_DPP_ADD(0x4E, 0xF);    // quad_perm [2,3,0,1]
    ACAI_DPP_ADD(0x141, 0xF);   // row_half_mirror
    ACAI_DPP_ADD(0x140, 0xF);   // row_mirror: every lane now holds its row's sum
    ACAI_DPP_ADD(0x142, 0xA);   // row_bcast15 into rows 1 and 3
    ACAI_DPP_ADD(0x143, 0xC);   // row_bcast31 into rows 2 and 3: lane 63 holds the wave's sum
#undef ACAI_DPP_ADD
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
// Once-per-DEVICE latch for hipFuncSetAttribute and the like (function attributes are per device; a per-process `static bool` skipped the
// call for every device after the first when one process drives several GPUs).  Usage: static bool seen[ACAI_MAX_DEV]; if (acai_first_on_device(seen)) ...
constexpr int ACAI_MAX_DEV = 64;
// XCD-aware block order of the one-dimensional attention grids: on for batches the host knows to be equal-length, off for ragged ones (the static
// split of the (sequence, head) pairs over the XCDs leaves the one with the longest sequences working alone at the end: attn_bwd1p.hip has the
// measurement).  ACAI_XCD_ORDER=0 / 1 forces it off / on (A/B aid, read once).
static inline bool acai_xcd_order(bool equal_len) {
    static const int force = getenv("ACAI_XCD_ORDER") ? atoi(getenv("ACAI_XCD_ORDER")) : -1;
    return force < 0 ? equal_len : force != 0;
}
static inline bool acai_first_on_device(bool (&seen)[ACAI_MAX_DEV]) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= ACAI_MAX_DEV) return true;   // unknown device: just do the (idempotent) call
    if (seen[d]) return false;
    seen[d] = true;
    return true;
}
static inline bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }
