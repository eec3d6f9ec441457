// KV-cached greedy decode step: the HBM-bound half of the path.
// Reference: CachedTransformerDecoderLayer.cached_forward (acai_omr/models/kv_caching.py:190-223),
// KVCache.update (K:83-109), CachedTransformerDecoder.cached_generate (K:292-302),
// OMRDecoder.cached_generate (acai_omr/models/models.py:518-528), ViTOMR.cached_get_next_token (M:575-583),
// cached_greedy_generate loop body (M:603-611).
//
// gfx950 design: per step the bytes that matter are the decoder weights (read once) and, dominating, the
// cross-attention K/V of every sequence (12 layers x 2 x S x 1024 elements each).  Everything is a
// streaming kernel with 16-byte loads, fp32 accumulation and no host involvement: positions, lengths and
// the finished flags live in device memory, so one captured hipGraph replays for every token.
//   skinny_gemm   : y[B,N] = x[B,K] . W[N,K]^T, 8 lanes per weight row (128 contiguous bytes per row per
//                   wave instruction), x staged once per workgroup in LDS (bf16 or fp32), v_dot2_f32_bf16.
//   decode_attn   : flash-decoding split over keys: grid (split, head, sequence), 8 lanes per key
//                   (dh 64 bf16 = 128 B), online softmax per lane group, in-wave + LDS combine, one
//                   (m, l, o[dh]) partial per workgroup; attn_combine merges the splits.
#include <stdlib.h>

#include "common.h"

#include <mutex>
#include <unordered_map>

namespace {

constexpr int SK_KC = 1024;  // k elements of x staged in LDS per pass
constexpr int SK_BC = 8;     // batch rows per pass

struct SkinnyArgs {
    const float *x;        // [B, ldx] fp32
    const void *W;         // [N, ldw]
    const float *bias;     // [N] or null
    const float *residual; // [B, ldr] or null
    float *y;              // [B, ldy]
    int ldx, ldw, ldr, ldy, B, N, K, flags;
    // optional KV append (self-attention in_proj): columns E..3E also go to the caches at position step[1]
    void *k_cache, *v_cache;
    const int32_t *step;
    int E, H, dh, dhp, Tmax;
    // optional fused LayerNorm (MFMA kernel only): x := LN(x) on load (dim == K); the per-row (mean, rstd) can be
    // published for a later launch; residual := LN(residual) from published statistics
    const float *ln_w, *ln_b;
    float ln_eps;
    float *stats_out;        // [B][2]
    const float *rln_w, *rln_b, *rstats;
    int x_bf16, y_bf16;      // activation in / out stored as bf16 (x: row stride ldx in bf16 elements)
    int rows_per_block;      // MFMA kernel: weight rows per workgroup (set by the launcher)
    int w_cached;            // non-zero: default-policy (cacheable) weight loads instead of non-temporal ones (ACAI_SKINNY_NT, an A/B aid)
    const float *ln2_w, *ln2_b;   // chain kernel only: a SECOND LayerNorm applied to the normalised row (last layer's norm3, then the stack's final norm)
    float ln2_eps;
    unsigned long long *stamps;   // diagnostic (acai_debug_stamps): [workgroup][8] s_memrealtime stamps (100 MHz) of the kernel's stages, else null
};

// diagnostic stamp buffer: [launch][1024 workgroups][8] (tools/stamp_decode.py); one slot per skinny launch, handed out in launch order
static unsigned long long *g_stamps = nullptr;
static int g_stamp_cap = 0, g_stamp_next = 0;

template <typename TW, bool FAST>
__global__ __launch_bounds__(256) void skinny_gemm_kernel(SkinnyArgs a) {
    constexpr int ES = sizeof(TW), EPC = 16 / ES;
    __shared__ __attribute__((aligned(16))) unsigned char xs_raw[SK_BC * SK_KC * ES];
    TW *xs = reinterpret_cast<TW *>(xs_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = lane & 7, rsub = lane >> 3;
    const int n = blockIdx.x * 32 + wave * 8 + rsub;
    const TW *W = reinterpret_cast<const TW *>(a.W);
    const bool row_ok = n < a.N;

    for (int b0 = 0; b0 < a.B; b0 += SK_BC) {
        const int nb = min(SK_BC, a.B - b0);
        float acc[SK_BC];
#pragma unroll
        for (int b = 0; b < SK_BC; ++b) acc[b] = 0.f;
        for (int k0 = 0; k0 < a.K; k0 += SK_KC) {
            const int kc = min(SK_KC, a.K - k0);
            __syncthreads();  // previous pass has finished reading xs
            for (int i = tid; i < SK_BC * SK_KC; i += 256) {
                const int b = i / SK_KC, k = i - b * SK_KC;
                const float v = (b < nb && k < kc) ? a.x[(size_t)(b0 + b) * a.ldx + k0 + k] : 0.f;
                DT<TW>::st(xs + i, v);  // bf16 mode: autocast's input cast
            }
            __syncthreads();
            if (row_ok) {
                const int nsteps = (kc + 8 * EPC - 1) / (8 * EPC);
#pragma unroll 4
                for (int s = 0; s < nsteps; ++s) {
                    const int kl = (s * 8 + kq) * EPC;  // local k of this lane's chunk
                    uint4 wv = make_uint4(0, 0, 0, 0);
                    if constexpr (FAST) {
                        if (kl < kc) wv = *reinterpret_cast<const uint4 *>(W + (size_t)n * a.ldw + k0 + kl);
                    } else {
                        union { uint4 v; TW e[EPC]; } u;
                        u.v = wv;
#pragma unroll
                        for (int e = 0; e < EPC; ++e)
                            if (kl + e < kc) u.e[e] = W[(size_t)n * a.ldw + k0 + kl + e];
                        wv = u.v;
                    }
#pragma unroll
                    for (int b = 0; b < SK_BC; ++b) {
                        const uint4 xv = *reinterpret_cast<const uint4 *>(xs + b * SK_KC + kl);
                        if constexpr (ES == 2) {
                            acc[b] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, wv.x), __builtin_bit_cast(bf16x2, xv.x), acc[b], false);
                            acc[b] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, wv.y), __builtin_bit_cast(bf16x2, xv.y), acc[b], false);
                            acc[b] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, wv.z), __builtin_bit_cast(bf16x2, xv.z), acc[b], false);
                            acc[b] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, wv.w), __builtin_bit_cast(bf16x2, xv.w), acc[b], false);
                        } else {
                            const f32x4 w4 = __builtin_bit_cast(f32x4, wv), x4 = __builtin_bit_cast(f32x4, xv);
                            acc[b] = fmaf(w4[0], x4[0], acc[b]);
                            acc[b] = fmaf(w4[1], x4[1], acc[b]);
                            acc[b] = fmaf(w4[2], x4[2], acc[b]);
                            acc[b] = fmaf(w4[3], x4[3], acc[b]);
                        }
                    }
                }
            }
        }
        // reduce over the 8 lanes of a row; afterwards lane kq owns batch row b0 + kq
        float mine = 0.f;
#pragma unroll
        for (int b = 0; b < SK_BC; ++b) {
            float v = acc[b];
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 4);
            if (kq == b) mine = v;
        }
        const int b = b0 + kq;
        if (row_ok && kq < nb) {
            float v = mine + (a.bias ? a.bias[n] : 0.f);
            const bool rnd = a.flags & ACAI_GEMM_ROUND_BF16;
            if (rnd) v = round_bf16(v);
            if (a.flags & ACAI_GEMM_GELU) {
                v = gelu_erf(v);
                if (rnd) v = round_bf16(v);
            }
            if (a.k_cache && n >= a.E) {  // KVCache.update (K:94-95): position = entries already cached
                const int kv = (n - a.E) / a.E, e = (n - a.E) - kv * a.E, hh = e / a.dh, dd = e - hh * a.dh;
                const size_t off = (((size_t)b * a.H + hh) * a.Tmax + a.step[1]) * a.dhp + dd;
                DT<TW>::st(reinterpret_cast<TW *>(kv ? a.v_cache : a.k_cache) + off, v);
            }
            if (a.residual) v += a.residual[(size_t)b * a.ldr + n];
            a.y[(size_t)b * a.ldy + n] = v;
        }
    }
}

// ---- bf16 weights: MFMA skinny GEMM ---------------------------------------------------------------------------
// Workgroup = 16 weight rows x all of K; wave w owns K/4 of it, so every lane streams its 16-byte fragments of the
// weight rows straight into VGPRs (8 loads in flight per lane, issued BEFORE anything else) and
// v_mfma_f32_16x16x32_bf16 does the K reduction: A = W[16 rows][32 k], B = x^T[32 k][16 batch columns], D[row][batch]
// in fp32.  No cross-lane shuffles.  While the weight loads fly, the four waves build the bf16 activation image in
// LDS (pitch K*2+16 bytes: conflict-free ds_read_b128 over 16 rows): each wave reads whole rows of x into registers
// once, optionally applies the LayerNorm (post-LN decoder: x = LN(z)) from in-register statistics, rounds to bf16
// (= autocast's input cast).  The 4 K-slices meet in LDS once.  Fusions that remove launches from the decode step:
// LN on load, published (mean, rstd) for the residual path of a later launch, bf16 activations in / out.
constexpr int SKM_MAXK = 4096;

// 16-byte non-temporal load: decoder weights (and K/V) are read exactly once per decode step
template <int NV>
__device__ __forceinline__ void skm_row_stats(const float4 (&v)[NV], int K, int lane, float eps, float &mean, float &rstd) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
        if (j * 256 + lane * 4 < K) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    mean = wave_sum(s) / (float)K;
    float qq = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
        if (j * 256 + lane * 4 < K) {
            const float d0 = v[j].x - mean, d1 = v[j].y - mean, d2 = v[j].z - mean, d3 = v[j].w - mean;
            qq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    rstd = 1.0f / sqrtf(wave_sum(qq) / (float)K + eps);
}

// NV = float4 registers per lane for one fp32 activation row (K <= 256 * NV); NW = waves per workgroup = K slices
// (NW = K/256 puts a slice's 8 weight fragments per lane in flight at once: one HBM round trip per workgroup)
template <bool XBF16, int NV, int NW>
__global__ __launch_bounds__(64 * NW) void skinny_mfma_kernel(SkinnyArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int R = a.rows_per_block;                            // weight rows of this workgroup (4, 8 or 16)
    const int n0 = blockIdx.x * R;
    const int K = a.K, Kw = K / NW, kbase = wave * Kw, nch = Kw >> 5;
    const int pitch = K * 2 + 16;
    const bool row_ok = r < R && n0 + r < a.N;
    const bf16_t *Wrow = reinterpret_cast<const bf16_t *>(a.W) + (size_t)(row_ok ? n0 + r : 0) * a.ldw + kbase + 8 * q;
    float *red = reinterpret_cast<float *>(smem);              // [NW][256] floats
    unsigned char *xs = smem + NW * 1024;                      // [rows][pitch] bf16 activation image
    auto stamp = [&](int k) {
        if (a.stamps && tid == 0) a.stamps[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + k] = __builtin_amdgcn_s_memrealtime();
    };
    stamp(0);

    // batch tiles of 16 rows: one per workgroup along grid.y (B > 16: GRPO rollouts, max_batch_size = 32 inference) - the tiles of a weight
    // row block re-read its weights from L2 instead of queueing four latency chains inside one workgroup (64 rows: 24 -> 17 us per launch)
    for (int bt = blockIdx.y * 16; bt < a.B; bt += 16 * gridDim.y) {
        const int nb = min(16, a.B - bt);
        // 1. first batch of weight fragments.  Loads return in issue order, so whatever is requested first is waited for first: the activation
        // rows (whose consumer chain - statistics, LDS image, barrier - is the long one) are requested BEFORE the weights and the epilogue
        // operands, which are only needed after the barrier (+1-2 % tokens/s over weights-first on the same box).
        uint4 wf[8];
        bool w_requested = false;
        auto request_weights = [&]() {
            if (w_requested) return;
            w_requested = true;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                wf[c] = make_uint4(0, 0, 0, 0);
                if (c < nch && row_ok) wf[c] = a.w_cached ? *reinterpret_cast<const uint4 *>(Wrow + 32 * c) : ld_nt16(Wrow + 32 * c);
            }
        };
        constexpr bool X_FIRST = !XBF16 && NV <= 4;   // (the bf16-input and NV = 16 paths keep weights first)
        // 1b. wave 0 also fetches everything its epilogue needs now, so that nothing is loaded after the reduction
        float e_bias[4] = {0.f, 0.f, 0.f, 0.f}, e_res[4] = {0.f, 0.f, 0.f, 0.f}, e_rw[4] = {1.f, 1.f, 1.f, 1.f}, e_rb[4] = {0.f, 0.f, 0.f, 0.f};
        float rmean = 0.f, rrstd = 1.f;
        bool e_requested = false;
        auto request_epilogue = [&]() {
            if (e_requested) return;
            e_requested = true;
            if (wave == 0) {
                const int b = bt + r;
                const bool col_ok = r < nb;
                if (a.rln_w && col_ok) {
                    rmean = a.rstats[b * 2];
                    rrstd = a.rstats[b * 2 + 1];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int n = n0 + 4 * q + i;
                    if (4 * q + i < R && n < a.N) {
                        if (a.bias) e_bias[i] = a.bias[n];
                        if (a.residual && col_ok) e_res[i] = a.residual[(size_t)b * a.ldr + n];
                        if (a.rln_w) {
                            e_rw[i] = a.rln_w[n];
                            e_rb[i] = a.rln_b[n];
                        }
                    }
                }
            }
        };
        if (!X_FIRST) {
            request_weights();
            request_epilogue();
        }
        // 2. activation image: wave w takes rows w, w+4 (together), then w+8, w+12; lane takes 4-element groups
        if constexpr (XBF16) {
            // plain copy of the bf16 rows (global loads and LDS stores do not alias: the compiler hoists the loads of a row)
            for (int b0 = wave; b0 < nb; b0 += NW) {
                const bf16_t *xr0 = reinterpret_cast<const bf16_t *>(a.x) + (size_t)(bt + b0) * a.ldx;
#pragma unroll
                for (int j = 0; j < SKM_MAXK / 512; ++j) {
                    const int k = j * 512 + lane * 8;
                    if (k < K) *reinterpret_cast<uint4 *>(xs + b0 * pitch + k * 2) = *reinterpret_cast<const uint4 *>(xr0 + k);
                }
            }
        } else if constexpr (NV <= 4) {
            for (int b0 = wave; b0 < nb; b0 += 2 * NW) {
                const bool two = b0 + NW < nb;
                const float *xr0 = a.x + (size_t)(bt + b0) * a.ldx, *xr1 = a.x + (size_t)(bt + (two ? b0 + NW : b0)) * a.ldx;
                float4 v0[NV], v1[NV], lw[NV], lb[NV];
#pragma unroll
                for (int j = 0; j < NV; ++j)
                    if (j * 256 + lane * 4 < K) {
                        v0[j] = *reinterpret_cast<const float4 *>(xr0 + j * 256 + lane * 4);
                        v1[j] = *reinterpret_cast<const float4 *>(xr1 + j * 256 + lane * 4);
                        if (a.ln_w) {
                            lw[j] = *reinterpret_cast<const float4 *>(a.ln_w + j * 256 + lane * 4);
                            lb[j] = *reinterpret_cast<const float4 *>(a.ln_b + j * 256 + lane * 4);
                        }
                    }
                request_weights();   // behind this wave's activation rows
                request_epilogue();
                if (a.ln_w) {
                    // both rows' sum and sum of squares ride the same 6 cross-lane steps (4 independent chains); one-pass variance
                    // E[x^2] - mean^2 is accurate to ~1e-6 relative for O(1) activations and this path rounds to bf16 right after
                    float t0 = 0.f, u0 = 0.f, t1 = 0.f, u1 = 0.f;
#pragma unroll
                    for (int j = 0; j < NV; ++j)
                        if (j * 256 + lane * 4 < K) {
                            t0 += (v0[j].x + v0[j].y) + (v0[j].z + v0[j].w);
                            u0 += (v0[j].x * v0[j].x + v0[j].y * v0[j].y) + (v0[j].z * v0[j].z + v0[j].w * v0[j].w);
                            t1 += (v1[j].x + v1[j].y) + (v1[j].z + v1[j].w);
                            u1 += (v1[j].x * v1[j].x + v1[j].y * v1[j].y) + (v1[j].z * v1[j].z + v1[j].w * v1[j].w);
                        }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) {
                        t0 += __shfl_xor(t0, o);
                        u0 += __shfl_xor(u0, o);
                        t1 += __shfl_xor(t1, o);
                        u1 += __shfl_xor(u1, o);
                    }
                    const float invK = 1.0f / (float)K;
                    const float m0 = t0 * invK, m1 = t1 * invK;
                    const float s0 = 1.0f / sqrtf(fmaxf(u0 * invK - m0 * m0, 0.f) + a.ln_eps), s1 = 1.0f / sqrtf(fmaxf(u1 * invK - m1 * m1, 0.f) + a.ln_eps);
                    if (lane == 0 && a.stats_out && blockIdx.x == 0) {
                        a.stats_out[(bt + b0) * 2] = m0;
                        a.stats_out[(bt + b0) * 2 + 1] = s0;
                        if (two) {
                            a.stats_out[(bt + b0 + NW) * 2] = m1;
                            a.stats_out[(bt + b0 + NW) * 2 + 1] = s1;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < NV; ++j)
                        if (j * 256 + lane * 4 < K) {
                            v0[j].x = (v0[j].x - m0) * s0 * lw[j].x + lb[j].x; v0[j].y = (v0[j].y - m0) * s0 * lw[j].y + lb[j].y;
                            v0[j].z = (v0[j].z - m0) * s0 * lw[j].z + lb[j].z; v0[j].w = (v0[j].w - m0) * s0 * lw[j].w + lb[j].w;
                            v1[j].x = (v1[j].x - m1) * s1 * lw[j].x + lb[j].x; v1[j].y = (v1[j].y - m1) * s1 * lw[j].y + lb[j].y;
                            v1[j].z = (v1[j].z - m1) * s1 * lw[j].z + lb[j].z; v1[j].w = (v1[j].w - m1) * s1 * lw[j].w + lb[j].w;
                        }
                }
#pragma unroll
                for (int j = 0; j < NV; ++j)
                    if (j * 256 + lane * 4 < K) {
                        const int kb = (j * 256 + lane * 4) * 2;
                        *reinterpret_cast<uint2 *>(xs + b0 * pitch + kb) = make_uint2(pack_bf16(v0[j].x, v0[j].y), pack_bf16(v0[j].z, v0[j].w));
                        if (two) *reinterpret_cast<uint2 *>(xs + (b0 + NW) * pitch + kb) = make_uint2(pack_bf16(v1[j].x, v1[j].y), pack_bf16(v1[j].z, v1[j].w));
                    }
            }
        } else {
            for (int b = wave; b < nb; b += NW) {
                const float *xr = a.x + (size_t)(bt + b) * a.ldx;
                float4 v[NV];
#pragma unroll
                for (int j = 0; j < NV; ++j)
                    if (j * 256 + lane * 4 < K) v[j] = *reinterpret_cast<const float4 *>(xr + j * 256 + lane * 4);
                float mean = 0.f, rstd = 1.f;
                if (a.ln_w) {
                    skm_row_stats<NV>(v, K, lane, a.ln_eps, mean, rstd);
                    if (lane == 0 && a.stats_out && blockIdx.x == 0) {
                        a.stats_out[(bt + b) * 2] = mean;
                        a.stats_out[(bt + b) * 2 + 1] = rstd;
                    }
                }
#pragma unroll
                for (int j = 0; j < NV; ++j)
                    if (j * 256 + lane * 4 < K) {
                        const int k = j * 256 + lane * 4;
                        if (a.ln_w) {
                            const float4 w4 = *reinterpret_cast<const float4 *>(a.ln_w + k), b4 = *reinterpret_cast<const float4 *>(a.ln_b + k);
                            v[j].x = (v[j].x - mean) * rstd * w4.x + b4.x; v[j].y = (v[j].y - mean) * rstd * w4.y + b4.y;
                            v[j].z = (v[j].z - mean) * rstd * w4.z + b4.z; v[j].w = (v[j].w - mean) * rstd * w4.w + b4.w;
                        }
                        *reinterpret_cast<uint2 *>(xs + b * pitch + k * 2) = make_uint2(pack_bf16(v[j].x, v[j].y), pack_bf16(v[j].z, v[j].w));
                    }
            }
        }
        request_weights();   // waves without an activation row of this tile
        request_epilogue();
        stamp(1);
        __syncthreads();
        stamp(2);
        // 3. MFMA over this wave's K slice; batch columns >= nb read row 0 (their outputs are never stored)
        const unsigned char *xfrag = xs + (r < nb ? r : 0) * pitch + (kbase + 8 * q) * 2;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int c0 = 0; c0 < nch; c0 += 8) {
            uint4 wn[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {  // next batch of weight fragments (none when NW covers K in one batch)
                wn[c] = make_uint4(0, 0, 0, 0);
                if constexpr (NW * 256 < SKM_MAXK)
                    if (c0 + 8 + c < nch && row_ok) wn[c] = a.w_cached ? *reinterpret_cast<const uint4 *>(Wrow + 32 * (c0 + 8 + c)) : ld_nt16(Wrow + 32 * (c0 + 8 + c));
            }
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c0 + c < nch) {
                    const uint4 xf = *reinterpret_cast<const uint4 *>(xfrag + 64 * (c0 + c));
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[c]), __builtin_bit_cast(bf16x8, xf), acc, 0, 0, 0);
                }
#pragma unroll
            for (int c = 0; c < 8; ++c) wf[c] = wn[c];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) red[wave * 256 + lane * 4 + i] = acc[i];
        stamp(3);
        __syncthreads();
        stamp(4);
        if (wave == 0) {
            // D layout: col (batch) = lane & 15, row (weight row) = 4 * (lane >> 4) + i
            const int b = bt + r;
            const bool col_ok = r < nb;
            const bool rnd = a.flags & ACAI_GEMM_ROUND_BF16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = n0 + 4 * q + i;
                if (4 * q + i >= R || n >= a.N || !col_ok) continue;
                float v = e_bias[i];
#pragma unroll
                for (int w = 0; w < NW; ++w) v += red[w * 256 + lane * 4 + i];
                if (rnd) v = round_bf16(v);
                if (a.flags & ACAI_GEMM_GELU) {
                    v = gelu_erf(v);
                    if (rnd) v = round_bf16(v);
                }
                if (a.k_cache && n >= a.E) {
                    const int kv = (n - a.E) / a.E, e = (n - a.E) - kv * a.E, hh = e / a.dh, dd = e - hh * a.dh;
                    const size_t off = (((size_t)b * a.H + hh) * a.Tmax + a.step[1]) * a.dhp + dd;
                    reinterpret_cast<bf16_t *>(kv ? a.v_cache : a.k_cache)[off] = f2bf(v);
                }
                if (a.residual) v += a.rln_w ? (e_res[i] - rmean) * rrstd * e_rw[i] + e_rb[i] : e_res[i];
                if (a.y_bf16)
                    reinterpret_cast<bf16_t *>(a.y)[(size_t)b * a.ldy + n] = f2bf(v);
                else
                    a.y[(size_t)b * a.ldy + n] = v;
            }
        }
        stamp(5);
        __syncthreads();
    }
}

// ---- the decode chain's GEMV at its two hot shapes (K = 256 * NW: K = 1024 fp32 activations, K = 4096 bf16 activations) ------------------
// Same arithmetic and fusions as skinny_mfma_kernel (bit-identical results), rebuilt around what in-kernel stamps showed on MI355X
// (tools/stamp_decode.py): of a 5-7 us launch, 2.7-5.2 us passed before the activation image was complete and ~1 us in the epilogue, because
//   * the compiler fetched the 200-byte argument struct in SIX dependent scalar-load stages (each a cold round trip after a kernel boundary):
//     here every argument is forced into SGPRs by one batch of s_loads and one wait;
//   * loads sat inside per-lane branches, so the first use of the activation rows waited with vmcnt(0) for the weight fragments (HBM) and the
//     epilogue operands as well: here every load is unconditional (clamped address, or a buffer load whose out-of-range lanes return zero), in
//     straight-line code, so the compiler's counted waits are exact - activations first, weights stay in flight across the barrier, epilogue
//     operands are requested after the barrier and land under the weight wait;
//   * one wave reduced and finished all 4 x 64 outputs: here wave i finishes accumulator register i of every lane (4 waves in parallel).
typedef __attribute__((vector_size(16))) unsigned int skm_v4u;

template <bool XBF16, int LN, int NW, bool WFIRST = false>   // LN: 0 = none, 1 = LayerNorm on load, 2 = two LayerNorms in a row (unembed: norm3 of the last layer, then the final norm)
__global__ __launch_bounds__(64 * NW) void skinny_chain_kernel(SkinnyArgs a) {
    constexpr bool HAS_LN = LN > 0;
    constexpr int K = 256 * NW, PITCH = K * 2 + 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    asm volatile("" ::"s"(a.x), "s"(a.W), "s"(a.bias), "s"(a.residual), "s"(a.y), "s"(a.ldx), "s"(a.ldw), "s"(a.ldr), "s"(a.ldy), "s"(a.B), "s"(a.N),
                 "s"(a.flags), "s"(a.k_cache), "s"(a.v_cache), "s"(a.step));
    asm volatile("" ::"s"(a.E), "s"(a.H), "s"(a.dh), "s"(a.dhp), "s"(a.Tmax), "s"(a.ln_w), "s"(a.ln_b), "s"(a.ln_eps), "s"(a.stats_out), "s"(a.rln_w),
                 "s"(a.rln_b), "s"(a.rstats), "s"(a.y_bf16), "s"(a.rows_per_block), "s"(a.stamps));
    if constexpr (LN == 2) asm volatile("" ::"s"(a.ln2_w), "s"(a.ln2_b), "s"(a.ln2_eps));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int R = a.rows_per_block, n0 = blockIdx.x * R, bt = blockIdx.y * 16;
    const int nb = min(16, a.B - bt);
    float *red = reinterpret_cast<float *>(smem);   // [NW][256]
    unsigned char *xs = smem + NW * 1024;           // [rows][PITCH] bf16 activation image
    auto stamp = [&](int k) {
        if (a.stamps && tid == 0) a.stamps[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + k] = __builtin_amdgcn_s_memrealtime();
    };
    stamp(0);

    // weight fragments: lane (r, q) of wave w owns 8 x 16 B of row n0 + r at k = 256 w + 8 q + 32 c.  Buffer loads: lanes without a row are
    // out of range and read zeros - no branch.  Non-temporal: every weight byte is read once per step.
    const bool row_ok = r < R && n0 + r < a.N;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.W), 0, (int)((size_t)a.N * a.ldw * 2), 0x00020000);
    const unsigned woff = row_ok ? (unsigned)(((size_t)(n0 + r) * a.ldw + wave * 256 + 8 * q) * 2) : 0xFFF00000u;
    skm_v4u wf[8];
    auto request_weights = [&]() {
#pragma unroll
        for (int c = 0; c < 8; ++c) wf[c] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, woff + 64 * c, 0, 2);
    };

    if constexpr (XBF16) {
        // activation rows are bf16 already: the [nb][K] image is copied in half-row chunks (4 KB), wave w takes chunk w (and w + 16 when the
        // tile has more than 8 rows): every wave has 4 loads in flight, none idles.  A wave without a chunk copies chunk 0 again (same bytes
        // to the same place) instead of branching around its loads.
        static_assert(!XBF16 || NW == 16, "bf16 activations: K = 4096");
        auto copy = [&](int ch, bool first) {
            const int c = ch < 2 * nb ? ch : 0;
            const bf16_t *xr = reinterpret_cast<const bf16_t *>(a.x) + (size_t)(bt + (c >> 1)) * a.ldx + (c & 1) * (K / 2) + lane * 8;
            unsigned char *xd = xs + (c >> 1) * PITCH + ((c & 1) * (K / 2) + lane * 8) * 2;
            static_assert(!XBF16 || K == 4096, "four 16-byte pieces per lane and chunk");   // (named registers: an array here went to scratch)
            if (first && WFIRST) request_weights();   // A/B: the HBM round trip (long) requested ahead of the L2 one
            const uint4 x0 = *reinterpret_cast<const uint4 *>(xr), x1 = *reinterpret_cast<const uint4 *>(xr + 512),
                        x2 = *reinterpret_cast<const uint4 *>(xr + 1024), x3 = *reinterpret_cast<const uint4 *>(xr + 1536);
            if (first && !WFIRST) request_weights();
            __builtin_amdgcn_sched_barrier(0);
            *reinterpret_cast<uint4 *>(xd) = x0;
            *reinterpret_cast<uint4 *>(xd + 1024) = x1;
            *reinterpret_cast<uint4 *>(xd + 2048) = x2;
            *reinterpret_cast<uint4 *>(xd + 3072) = x3;
        };
        copy(wave, true);
        if (nb > 8) copy(wave + 16, false);
    } else {
        static_assert(XBF16 || NW == 4, "fp32 activations: K = 1024");
        // wave w builds rows w and w + 4 (then w + 8, w + 12 when the tile has more than 8 rows); lane takes 4 consecutive k per 256
        auto build = [&](int ra, int rb, bool first) {
            const bool oka = ra < nb, okb = rb < nb;
            const float *pa = a.x + (size_t)(bt + (oka ? ra : 0)) * a.ldx + lane * 4, *pb = a.x + (size_t)(bt + (okb ? rb : 0)) * a.ldx + lane * 4;
            float4 va[4], vb[4], lw[4], lb[4], lw2[4], lb2[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                va[j] = *reinterpret_cast<const float4 *>(pa + j * 256);
                vb[j] = *reinterpret_cast<const float4 *>(pb + j * 256);
            }
            if constexpr (HAS_LN) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    lw[j] = *reinterpret_cast<const float4 *>(a.ln_w + j * 256 + lane * 4);
                    lb[j] = *reinterpret_cast<const float4 *>(a.ln_b + j * 256 + lane * 4);
                }
            }
            if constexpr (LN == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    lw2[j] = *reinterpret_cast<const float4 *>(a.ln2_w + j * 256 + lane * 4);
                    lb2[j] = *reinterpret_cast<const float4 *>(a.ln2_b + j * 256 + lane * 4);
                }
            }
            if (first) request_weights();   // behind this wave's activation rows: loads return in issue order
            __builtin_amdgcn_sched_barrier(0);   // every load above is in flight before anything is waited for (hipcc otherwise sinks the LN / weight loads below the statistics)
            if constexpr (HAS_LN) {
                // both rows' sum and sum of squares ride the same 6 cross-lane steps; one-pass variance (as skinny_mfma_kernel)
                float t0 = 0.f, u0 = 0.f, t1 = 0.f, u1 = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    t0 += (va[j].x + va[j].y) + (va[j].z + va[j].w);
                    u0 += (va[j].x * va[j].x + va[j].y * va[j].y) + (va[j].z * va[j].z + va[j].w * va[j].w);
                    t1 += (vb[j].x + vb[j].y) + (vb[j].z + vb[j].w);
                    u1 += (vb[j].x * vb[j].x + vb[j].y * vb[j].y) + (vb[j].z * vb[j].z + vb[j].w * vb[j].w);
                }
                t0 = wave_sum_dpp(t0);   // four independent chains on the DPP network
                u0 = wave_sum_dpp(u0);
                t1 = wave_sum_dpp(t1);
                u1 = wave_sum_dpp(u1);
                const float invK = 1.0f / (float)K;
                const float m0 = t0 * invK, m1 = t1 * invK;
                const float s0 = 1.0f / sqrtf(fmaxf(u0 * invK - m0 * m0, 0.f) + a.ln_eps), s1 = 1.0f / sqrtf(fmaxf(u1 * invK - m1 * m1, 0.f) + a.ln_eps);
                if (lane == 0 && a.stats_out && blockIdx.x == 0) {
                    if (oka) {
                        a.stats_out[(bt + ra) * 2] = m0;
                        a.stats_out[(bt + ra) * 2 + 1] = s0;
                    }
                    if (okb) {
                        a.stats_out[(bt + rb) * 2] = m1;
                        a.stats_out[(bt + rb) * 2 + 1] = s1;
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    va[j].x = (va[j].x - m0) * s0 * lw[j].x + lb[j].x; va[j].y = (va[j].y - m0) * s0 * lw[j].y + lb[j].y;
                    va[j].z = (va[j].z - m0) * s0 * lw[j].z + lb[j].z; va[j].w = (va[j].w - m0) * s0 * lw[j].w + lb[j].w;
                    vb[j].x = (vb[j].x - m1) * s1 * lw[j].x + lb[j].x; vb[j].y = (vb[j].y - m1) * s1 * lw[j].y + lb[j].y;
                    vb[j].z = (vb[j].z - m1) * s1 * lw[j].z + lb[j].z; vb[j].w = (vb[j].w - m1) * s1 * lw[j].w + lb[j].w;
                }
            }
            if constexpr (LN == 2) {   // the second norm on the fp32 rows in registers (two-pass variance: the rows are O(1) after the first)
                float t0 = 0.f, t1 = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    t0 += (va[j].x + va[j].y) + (va[j].z + va[j].w);
                    t1 += (vb[j].x + vb[j].y) + (vb[j].z + vb[j].w);
                }
                const float invK = 1.0f / (float)K;
                const float m0 = wave_sum_dpp(t0) * invK, m1 = wave_sum_dpp(t1) * invK;
                float u0 = 0.f, u1 = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    va[j].x -= m0; va[j].y -= m0; va[j].z -= m0; va[j].w -= m0;
                    vb[j].x -= m1; vb[j].y -= m1; vb[j].z -= m1; vb[j].w -= m1;
                    u0 += (va[j].x * va[j].x + va[j].y * va[j].y) + (va[j].z * va[j].z + va[j].w * va[j].w);
                    u1 += (vb[j].x * vb[j].x + vb[j].y * vb[j].y) + (vb[j].z * vb[j].z + vb[j].w * vb[j].w);
                }
                const float s0 = 1.0f / sqrtf(wave_sum_dpp(u0) * invK + a.ln2_eps), s1 = 1.0f / sqrtf(wave_sum_dpp(u1) * invK + a.ln2_eps);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    va[j].x = va[j].x * s0 * lw2[j].x + lb2[j].x; va[j].y = va[j].y * s0 * lw2[j].y + lb2[j].y;
                    va[j].z = va[j].z * s0 * lw2[j].z + lb2[j].z; va[j].w = va[j].w * s0 * lw2[j].w + lb2[j].w;
                    vb[j].x = vb[j].x * s1 * lw2[j].x + lb2[j].x; vb[j].y = vb[j].y * s1 * lw2[j].y + lb2[j].y;
                    vb[j].z = vb[j].z * s1 * lw2[j].z + lb2[j].z; vb[j].w = vb[j].w * s1 * lw2[j].w + lb2[j].w;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kb = (j * 256 + lane * 4) * 2;
                if (oka) *reinterpret_cast<uint2 *>(xs + ra * PITCH + kb) = make_uint2(pack_bf16(va[j].x, va[j].y), pack_bf16(va[j].z, va[j].w));
                if (okb) *reinterpret_cast<uint2 *>(xs + rb * PITCH + kb) = make_uint2(pack_bf16(vb[j].x, vb[j].y), pack_bf16(vb[j].z, vb[j].w));
            }
        };
        build(wave, wave + 4, true);
        if (nb > 8) build(wave + 8, wave + 12, false);
    }
    stamp(1);
    __syncthreads();
    stamp(2);

    // epilogue operands of the output this lane will finish (wave i < 4 finishes accumulator register i): requested now, used after the
    // reduction - they land while the weight fragments are waited for
    const int oi = wave & 3;
    const int on = n0 + 4 * q + oi, ob = bt + r;
    const bool out_ok = wave < 4 && 4 * q + oi < R && on < a.N && r < nb;
    float e_bias = 0.f, e_res = 0.f, e_rw = 1.f, e_rb = 0.f, rmean = 0.f, rrstd = 1.f;
    if (wave < 4) {
        const int cn = out_ok ? on : 0, cb = out_ok ? ob : 0;
        if (a.bias) e_bias = a.bias[cn];
        if (a.residual) e_res = a.residual[(size_t)cb * a.ldr + cn];
        if (a.rln_w) {
            e_rw = a.rln_w[cn];
            e_rb = a.rln_b[cn];
            rmean = a.rstats[cb * 2];
            rrstd = a.rstats[cb * 2 + 1];
        }
    }
    // MFMA over this wave's K slice; batch columns >= nb read row 0 (their outputs are never stored)
    const unsigned char *xfrag = xs + (r < nb ? r : 0) * PITCH + (wave * 256 + 8 * q) * 2;
    uint4 xf[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) xf[c] = *reinterpret_cast<const uint4 *>(xfrag + 64 * c);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 8; ++c) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[c]), __builtin_bit_cast(bf16x8, xf[c]), acc, 0, 0, 0);
    *reinterpret_cast<f32x4 *>(red + wave * 256 + lane * 4) = acc;
    stamp(3);
    __syncthreads();
    stamp(4);
    if (wave < 4) {
        // D layout: col (batch) = lane & 15, row (weight row) = 4 * (lane >> 4) + register
        float v = e_bias;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[w * 256 + lane * 4 + oi];
        const bool rnd = a.flags & ACAI_GEMM_ROUND_BF16;
        if (rnd) v = round_bf16(v);
        if (a.flags & ACAI_GEMM_GELU) {
            v = gelu_erf(v);
            if (rnd) v = round_bf16(v);
        }
        if (out_ok) {
            if (a.k_cache && on >= a.E) {   // KVCache.update (K:94-95): position = entries already cached
                const int kv = (on - a.E) / a.E, e = (on - a.E) - kv * a.E, hh = e / a.dh, dd = e - hh * a.dh;
                const size_t off = (((size_t)ob * a.H + hh) * a.Tmax + a.step[1]) * a.dhp + dd;
                reinterpret_cast<bf16_t *>(kv ? a.v_cache : a.k_cache)[off] = f2bf(v);
            }
            if (a.residual) v += a.rln_w ? (e_res - rmean) * rrstd * e_rw + e_rb : e_res;
            if (a.y_bf16)
                reinterpret_cast<bf16_t *>(a.y)[(size_t)ob * a.ldy + on] = f2bf(v);
            else
                a.y[(size_t)ob * a.ldy + on] = v;
        }
    }
    stamp(5);
}

static inline bool skinny_mfma_ok(const SkinnyArgs &a) {
    return (a.K % 256 == 0) && a.K <= SKM_MAXK && (a.ldw % 8 == 0) && (a.ldx % 8 == 0) && aligned16(a.W) && aligned16(a.x) &&
           (!a.ln_w || (aligned16(a.ln_w) && aligned16(a.ln_b)));
}

template <typename TW>
int launch_skinny(const SkinnyArgs &a, hipStream_t st) {
    if constexpr (sizeof(TW) == 2) {
        if (skinny_mfma_ok(a)) {
            const int rows = a.B < 16 ? ((a.B + 3) & ~3) : 16;
            const bool wide = a.x_bf16 && a.K == 4096;  // 16 K-slices: every weight fragment of the workgroup in flight at once
            const size_t lds = (wide ? 16 : 4) * 1024 + (size_t)rows * (a.K * 2 + 16);
            static bool attr_done[ACAI_MAX_DEV] = {};
            if (acai_first_on_device(attr_done)) {  // opt in to > 64 KB of dynamic LDS (K = 4096 activation image); per device
                hipFuncSetAttribute(reinterpret_cast<const void *>(skinny_mfma_kernel<false, 4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
                hipFuncSetAttribute(reinterpret_cast<const void *>(skinny_mfma_kernel<false, 16, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
                hipFuncSetAttribute(reinterpret_cast<const void *>(skinny_mfma_kernel<true, 1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
                hipFuncSetAttribute(reinterpret_cast<const void *>(skinny_mfma_kernel<true, 1, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
            }
            // one CU ingests only ~25 GB/s from HBM: spread a small weight matrix over >= ~200 workgroups by giving each
            // fewer than 16 rows (the unused MFMA rows load nothing)
            SkinnyArgs b = a;
            static const int rpb = getenv("ACAI_SKINNY_ROWS") ? atoi(getenv("ACAI_SKINNY_ROWS")) : 0;
            b.w_cached = 0;
            // weight cache policy (A/B aid): 0 = default-policy loads for every matrix, 2 = default policy below 8 MB (candidates for the
            // Infinity Cache across steps) and non-temporal above, unset = non-temporal everywhere
            static const int ntm = getenv("ACAI_SKINNY_NT") ? atoi(getenv("ACAI_SKINNY_NT")) : 1;
            if (ntm == 0 || (ntm == 2 && (size_t)a.N * a.K * 2 < (8u << 20))) b.w_cached = 1;
            b.rows_per_block = rpb ? rpb : (a.N >= 2560 ? 16 : (a.N >= 1600 ? 8 : 4));
            b.stamps = nullptr;
            if (g_stamps && g_stamp_next < g_stamp_cap) b.stamps = g_stamps + (size_t)(g_stamp_next++) * 1024 * 8;
            // (the K = 4096 form holds a 131 KB activation image: one workgroup per CU, so its batch tiles stay a loop inside the workgroup)
            static const bool no_chain = getenv("ACAI_SKINNY_CHAIN") && atoi(getenv("ACAI_SKINNY_CHAIN")) == 0;   // A/B aid
            const bool chain_ok = !no_chain && (size_t)a.N * a.ldw * 2 < 0xFFF00000u && (a.ldx % 8 == 0);   // (row statistics are only published with a LayerNorm on load, as in skinny_mfma_kernel)
            if (a.ln2_w && !(chain_ok && a.K == 1024 && !a.x_bf16 && a.ln_w)) return acai_set_err(-1, "skinny_gemm: the double LayerNorm needs the chain kernel (K = 1024, fp32 activations)");
            if (chain_ok && ((a.K == 1024 && !a.x_bf16) || (a.K == 4096 && a.x_bf16 && !a.ln_w))) {
                static bool attr2[ACAI_MAX_DEV] = {};
                if (acai_first_on_device(attr2)) {
                    hipFuncSetAttribute(reinterpret_cast<const void *>(skinny_chain_kernel<true, 0, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
                    hipFuncSetAttribute(reinterpret_cast<const void *>(skinny_chain_kernel<true, 0, 16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
                }
                const dim3 cgrid(cdiv(a.N, b.rows_per_block), cdiv(a.B, 16));
                const int nw = a.x_bf16 ? 16 : 4;
                const size_t clds = (size_t)nw * 1024 + (size_t)rows * (a.K * 2 + 16);
                static const bool wfirst = getenv("ACAI_LIN2_WFIRST") && atoi(getenv("ACAI_LIN2_WFIRST")) == 1;   // A/B aid
                if (a.x_bf16 && wfirst)
                    hipLaunchKernelGGL((skinny_chain_kernel<true, 0, 16, true>), cgrid, dim3(1024), clds, st, b);
                else if (a.x_bf16)
                    hipLaunchKernelGGL((skinny_chain_kernel<true, 0, 16>), cgrid, dim3(1024), clds, st, b);
                else if (a.ln_w && a.ln2_w)
                    hipLaunchKernelGGL((skinny_chain_kernel<false, 2, 4>), cgrid, dim3(256), clds, st, b);
                else if (a.ln_w)
                    hipLaunchKernelGGL((skinny_chain_kernel<false, 1, 4>), cgrid, dim3(256), clds, st, b);
                else
                    hipLaunchKernelGGL((skinny_chain_kernel<false, 0, 4>), cgrid, dim3(256), clds, st, b);
                ACAI_LAUNCH_CHECK("skinny_chain");
                return 0;
            }
            const dim3 grid(cdiv(a.N, b.rows_per_block), wide ? 1 : cdiv(a.B, 16));
            if (wide)
                hipLaunchKernelGGL((skinny_mfma_kernel<true, 1, 16>), grid, dim3(1024), lds, st, b);
            else if (a.x_bf16)
                hipLaunchKernelGGL((skinny_mfma_kernel<true, 1, 4>), grid, dim3(256), lds, st, b);
            else if (a.K <= 1024)
                hipLaunchKernelGGL((skinny_mfma_kernel<false, 4, 4>), grid, dim3(256), lds, st, b);
            else
                hipLaunchKernelGGL((skinny_mfma_kernel<false, 16, 4>), grid, dim3(256), lds, st, b);
            ACAI_LAUNCH_CHECK("skinny_mfma");
            return 0;
        }
    }
    if (a.ln_w || a.rln_w || a.x_bf16 || a.y_bf16) return acai_set_err(-1, "skinny_gemm: fused LayerNorm needs the bf16 MFMA path (K %% 128 == 0, 16-byte aligned operands)");
    constexpr int EPC = 16 / sizeof(TW);
    const bool fast = (a.K % EPC == 0) && (a.ldw % EPC == 0) && aligned16(a.W);
    dim3 grid(cdiv(a.N, 32));
    if (fast)
        hipLaunchKernelGGL((skinny_gemm_kernel<TW, true>), grid, dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((skinny_gemm_kernel<TW, false>), grid, dim3(256), 0, st, a);
    ACAI_LAUNCH_CHECK("skinny_gemm");
    return 0;
}

// ---- decode attention ---------------------------------------------------------------------------------------
struct DAttnArgs {
    const float *q;   // [B, ldq] fp32, head h at column h*dh
    const void *kc, *vc;
    const int64_t *seq_off;  // per-sequence element offset (ragged cross K/V) or null
    const int32_t *seq_len;  // per-sequence length (cross) or null
    const int32_t *step;     // self-attention: length = step[1] + 1, layout [B][H][Tmax][dhp]
    float *partial;          // [B][H][nsplit][dhp + 2]
    int ldq, H, dh, dhp, Tmax, chunk, nsplit;
    float scale_log2e;
    float *out;              // nsplit == 1: the workgroup writes softmax(qK^T)V itself to out[b, h*dh + d] (no combine launch)
    int ldo, round_out;
    unsigned *tickets;       // [B*H] arrival counters (zero between launches): the LAST workgroup of a (b, h) merges the splits
};

// LPK = lanes per key = dhp * sizeof(TC) / 16.  RAGGED = cross attention over the ragged encoder memory (the dominant
// HBM stream of a decode step); !RAGGED = self attention over the [B][H][Tmax][dhp] cache.  Two instantiations so that
// rocprof reports them as separate kernels.
template <typename TC, int LPK, bool RAGGED, int U = 2>
__global__ __launch_bounds__(256) void decode_attn_kernel(DAttnArgs a) {
    constexpr int EPC = 16 / sizeof(TC), KPW = 64 / LPK;
    __shared__ float red[4][2 + 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = lane % LPK, kg = lane / LPK;
    const int split = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    int len, hstride;
    size_t base;
    if constexpr (RAGGED) {
        len = a.seq_len[b];
        hstride = len * a.dhp;
        base = (size_t)a.seq_off[b] + (size_t)h * hstride;
    } else {
        len = a.step[1] + 1;
        hstride = a.Tmax * a.dhp;
        base = ((size_t)b * a.H + h) * hstride;
    }
    float *part = a.partial + (((size_t)b * a.H + h) * a.nsplit + split) * (a.dhp + 2);
    const int c0 = split * a.chunk, c1 = min(len, c0 + a.chunk);
    const bool fused_merge = a.tickets && a.out && a.nsplit > 1;
    if (c0 >= len) {  // empty split: neutral element (m = -1e30, l = 0, o = 0)
        if (tid < a.dhp + 2) {
            if (fused_merge) __hip_atomic_store(part + tid, tid == 0 ? -1.0e30f : 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else part[tid] = tid == 0 ? -1.0e30f : 0.f;
        }
        if (!fused_merge) return;
        // the neutral stores of wave 1 (elements 64, 65) must be drained before wave 0 takes the ticket below
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const TC *Kp = reinterpret_cast<const TC *>(a.kc) + base;
    const TC *Vp = reinterpret_cast<const TC *>(a.vc) + base;

    float qf[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        const int d = kq * EPC + e;
        qf[e] = d < a.dh ? a.q[(size_t)b * a.ldq + h * a.dh + d] : 0.f;
    }
    float m = -1.0e30f, l = 0.f, acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;

    // software pipeline: the U key groups of iteration i+1 are requested before iteration i is computed (a wave that computes has no load
    // in flight otherwise: PMC showed the VALU busy a third of the time and the waves waiting on memory for half of it)
    uint4 kn[U], vn[U];
    auto request = [&](int key0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int key = key0 + u * 4 * KPW;
            kn[u] = vn[u] = make_uint4(0, 0, 0, 0);
            if (key < c1) {
                // every K/V byte is read exactly once per step: non-temporal loads (streaming cache policy)
                kn[u] = ld_nt16(Kp + (size_t)key * a.dhp + kq * EPC);
                vn[u] = ld_nt16(Vp + (size_t)key * a.dhp + kq * EPC);
            }
        }
    };
    request(c0 + wave * KPW + kg);
    for (int key0 = c0 + wave * KPW + kg; key0 < c1; key0 += 4 * KPW * U) {
        uint4 kk[U], vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            kk[u] = kn[u];
            vv[u] = vn[u];
        }
        if (key0 + 4 * KPW * U < c1) request(key0 + 4 * KPW * U);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int key = key0 + u * 4 * KPW;
            float kf[EPC], vf[EPC];
            if constexpr (sizeof(TC) == 2) {
                const uint32_t kw[4] = {kk[u].x, kk[u].y, kk[u].z, kk[u].w}, vw[4] = {vv[u].x, vv[u].y, vv[u].z, vv[u].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    kf[2 * e] = __uint_as_float(kw[e] << 16);
                    kf[2 * e + 1] = __uint_as_float(kw[e] & 0xffff0000u);
                    vf[2 * e] = __uint_as_float(vw[e] << 16);
                    vf[2 * e + 1] = __uint_as_float(vw[e] & 0xffff0000u);
                }
            } else {
                const f32x4 k4 = __builtin_bit_cast(f32x4, kk[u]), v4 = __builtin_bit_cast(f32x4, vv[u]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    kf[e] = k4[e];
                    vf[e] = v4[e];
                }
            }
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < EPC; ++e) s = fmaf(qf[e], kf[e], s);
#pragma unroll
            for (int o = 1; o < LPK; o <<= 1) s += __shfl_xor(s, o);
            if (key < c1) {  // uniform inside a lane group
                s *= a.scale_log2e;
                const float mn = fmaxf(m, s), al = fast_exp2(m - mn), p = fast_exp2(s - mn);
                m = mn;
                l = l * al + p;
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc[e] = acc[e] * al + p * vf[e];
            }
        }
    }
    // merge the KPW lane groups of this wave (lanes with equal kq)
    float mw = m;
#pragma unroll
    for (int o = LPK; o < 64; o <<= 1) mw = fmaxf(mw, __shfl_xor(mw, o));
    const float f = fast_exp2(m - mw);
    l *= f;
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] *= f;
#pragma unroll
    for (int o = LPK; o < 64; o <<= 1) {
        l += __shfl_xor(l, o);
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] += __shfl_xor(acc[e], o);
    }
    if (kg == 0) {
        if (kq == 0) {
            red[wave][0] = mw;
            red[wave][1] = l;
        }
#pragma unroll
        for (int e = 0; e < EPC; ++e) red[wave][2 + kq * EPC + e] = acc[e];
    }
    __syncthreads();
    if (wave != 0) return;
    // Wave 0 finishes alone - lane d owns output dim d - so the tail needs no workgroup barrier: combine the four waves, publish the split's
    // partial (write-through), drain, take the ticket, and (last arrival only) merge all splits.
    const int d = lane;
    const float M = fmaxf(fmaxf(red[0][0], red[1][0]), fmaxf(red[2][0], red[3][0]));
    float v = 0.f, lsum = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float f = fast_exp2(red[w][0] - M);
        v += red[w][2 + d] * f;
        lsum += red[w][1] * f;
    }
    if (a.nsplit == 1 && a.out) {
        if (d < a.dh) {
            float o = v / lsum;
            if (a.round_out) o = round_bf16(o);
            a.out[(size_t)b * a.ldo + h * a.dh + d] = o;
        }
        return;
    }
    if (c0 < len && d < a.dhp) {
        // fused merge: write-through (sc1) stores, so the hand-off needs no release fence (an L2 write-back per workgroup)
        if (fused_merge) {
            __hip_atomic_store(part + 2 + d, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d == 0) {
                __hip_atomic_store(part, M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(part + 1, lsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            part[2 + d] = v;
            if (d == 0) {
                part[0] = M;
                part[1] = lsum;
            }
        }
    }
    if (!fused_merge) return;
    // In-launch merge of the split partials (placement-independent hand-off, write-through form): the partials were stored sc1 (agent-scope
    // atomic stores) by this wave, which drains them, then its lane 0 takes a ticket with an agent-scope atomic add; the wave that draws
    // nsplit-1 reads every partial with sc1 loads (agent-scope atomic loads bypass this CU's L1) and merges.  The counter re-arms itself.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int last = 0;
    if (lane == 0) {
        unsigned *cnt = a.tickets + (size_t)b * a.H + h;
        const unsigned t = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = t == (unsigned)(a.nsplit - 1);
        if (last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    last = __builtin_amdgcn_readfirstlane(last);
    if (!last || d >= a.dhp) return;
    float *p = a.partial + ((size_t)b * a.H + h) * a.nsplit * (a.dhp + 2);
    auto ld = [&](int i) { return __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    float Mm = -1.0e30f, lm = 0.f, om = 0.f;
    // the merge sits on the step's critical path: request all (max, sum, value) triples of up to 8 splits before touching any of them
    // (a rolled loop issues one dependent L2 round trip after another: 3 x nsplit of them)
    for (int s0 = 0; s0 < a.nsplit; s0 += 8) {
        float pm[8], pl[8], po[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool in = s0 + u < a.nsplit;
            const int pb = (in ? s0 + u : s0) * (a.dhp + 2);
            pm[u] = in ? ld(pb) : -1.0e30f;
            pl[u] = in ? ld(pb + 1) : 0.f;
            po[u] = in ? ld(pb + 2 + d) : 0.f;
        }
        float Mc = Mm;
#pragma unroll
        for (int u = 0; u < 8; ++u) Mc = fmaxf(Mc, pm[u]);
        const float resc = fast_exp2(Mm - Mc);
        lm *= resc;
        om *= resc;
        Mm = Mc;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float w = fast_exp2(pm[u] - Mm);
            lm += pl[u] * w;
            om += po[u] * w;
        }
    }
    if (d < a.dh) {
        float o = om / lm;
        if (a.round_out) o = round_bf16(o);
        a.out[(size_t)b * a.ldo + h * a.dh + d] = o;
    }
}

// ---- cross attention of a rollout GROUP (GRPO, models.py:883-891 / 988-1049) on the matrix cores (bf16, d_h padded to 64) --------------
// `group` consecutive decode rows share one image's cross K/V (the engine stores it once).  A workgroup owns (split, head, image, tile of
// 16 rows) and streams its K/V chunk ONCE for all of them: the HBM stream of a step no longer grows with the group size.  (A VALU form -
// GT dot products and softmax updates per key and lane group - was built first and measured SLOWER than letting the rows alias the
// stored K/V through the per-row kernel: 3.7-4.8 ms against 3.25 ms per step at 8 x 8; it is gone.)
// Up to 16 rollout rows of one image are the 16 columns of v_mfma_f32_16x16x32_bf16.  A wave owns 32-key tiles of the workgroup's chunk:
//   S[key][g]  = K . Q^T      A = K rows, loaded from global memory straight in the A layout (lane = key, 16 B = 8 dims), B = Q^T in registers
//   O^T[d][g] += V^T . P      B = P taken from the S accumulators as they stand (keys 4q+j of both 16-key halves = contraction slots 8q+j),
//                             A = V^T read with ds_read_b64_tr_b16 from the wave's private 4 KB image of the V tile (no barrier in the loop)
// so the K/V stream is read once per IMAGE and the per-key VALU work is the softmax of 8 scores per lane.  The running maximum of a query is
// kept equal across the four lanes that share its column (two shuffles when it is raised, lazily); partials / tickets / merge as above.
__global__ __launch_bounds__(256) void decode_attn_gmfma_kernel(DAttnArgs a, int group, int gtiles) {
    typedef bf16_t TC;
    typedef TileLayout<2, 64> TL;
    constexpr int GT = 16, KT = 32, DHP = 64;
    __shared__ __attribute__((aligned(16))) unsigned char vlds[4][KT * DHP * 2];
    __shared__ float red[4][GT][2 + 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int split = blockIdx.x, h = blockIdx.y, img = blockIdx.z / gtiles, gt = blockIdx.z % gtiles;
    const int row0 = img * group + gt * GT, ng = min(GT, group - gt * GT);
    const int len = a.seq_len[row0], hstride = len * DHP;
    const size_t base = (size_t)a.seq_off[row0] + (size_t)h * hstride;
    const int c0 = split * a.chunk, c1 = min(len, c0 + a.chunk);
    const bool fused_merge = a.tickets && a.out && a.nsplit > 1;
    auto part_of = [&](int g) { return a.partial + (((size_t)(row0 + g) * a.H + h) * a.nsplit + split) * (DHP + 2); };
    if (c0 >= len) {  // empty split: neutral elements
        if (tid < DHP + 2)
            for (int g = 0; g < ng; ++g) {
                if (fused_merge) __hip_atomic_store(part_of(g) + tid, tid == 0 ? -1.0e30f : 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else part_of(g)[tid] = tid == 0 ? -1.0e30f : 0.f;
            }
        if (!fused_merge) return;
    }
    const TC *Kp = reinterpret_cast<const TC *>(a.kc) + base;
    const TC *Vp = reinterpret_cast<const TC *>(a.vc) + base;

    // Q^T fragments: lane (g = r16, kq) holds dims db * 32 + kq * 8 + 0..7 of query row0 + g (bf16, as the reference's autocast SDPA input)
    uint4 qb[2];
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        float t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int d = db * 32 + kq * 8 + e;
            t[e] = (r16 < ng && d < a.dh) ? a.q[(size_t)(row0 + r16) * a.ldq + h * a.dh + d] : 0.f;
        }
        qb[db] = make_uint4(pack_bf16(t[0], t[1]), pack_bf16(t[2], t[3]), pack_bf16(t[4], t[5]), pack_bf16(t[6], t[7]));
    }
    float m = -1.0e30f, l = 0.f;
    f32x4 oacc[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) oacc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned char *vimg = vlds[wave];
    typedef __attribute__((ext_vector_type(4))) short s4;
    typedef __attribute__((address_space(3))) s4 *lds_s4;

    for (int key0 = c0 + wave * KT; key0 < c1; key0 += 4 * KT) {
        uint4 kf[2][2], vv[4];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int key = key0 + sub * 16 + r16;
#pragma unroll
            for (int db = 0; db < 2; ++db) kf[sub][db] = key < c1 ? ld_nt16(Kp + (size_t)key * DHP + db * 32 + kq * 8) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane + 64 * i, row = c >> 3, key = key0 + row;
            vv[i] = key < c1 ? ld_nt16(Vp + (size_t)key * DHP + (c & 7) * 8) : make_uint4(0, 0, 0, 0);
        }
        f32x4 sc[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            sc[sub] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int db = 0; db < 2; ++db)
                sc[sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kf[sub][db]), __builtin_bit_cast(bf16x8, qb[db]), sc[sub], 0, 0, 0);
        }
        // this lane: keys key0 + 16 sub + 4 kq + j of query r16
        float tmax = -1.0e30f;
        bool ok[2][4];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ok[sub][j] = key0 + sub * 16 + 4 * kq + j < c1;
                sc[sub][j] *= a.scale_log2e;
                if (ok[sub][j]) tmax = fmaxf(tmax, sc[sub][j]);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));   // equal on the four lanes of a query column
        if (__ballot(tmax > m + 8.0f)) {
            const float mn = fmaxf(m, tmax), al = fast_exp2(m - mn);
            m = mn;
            l *= al;
#pragma unroll
            for (int d = 0; d < 4; ++d) oacc[d] *= al;
        }
        float p[2][4];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                p[sub][j] = ok[sub][j] ? fast_exp2(sc[sub][j] - m) : 0.f;
                l += p[sub][j];
            }
        const uint4 pf = make_uint4(pack_bf16(p[0][0], p[0][1]), pack_bf16(p[0][2], p[0][3]), pack_bf16(p[1][0], p[1][1]), pack_bf16(p[1][2], p[1][3]));
        // V tile -> this wave's LDS image (the previous tile's transposing reads were consumed by its MFMAs: same wave, in order)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane + 64 * i;
            *reinterpret_cast<uint4 *>(vimg + TL::off(c >> 3, c & 7)) = vv[i];
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int vrow = 4 * kq + (r16 >> 2), vchunk = d * 2 + ((r16 & 3) >> 1), vsub = 8 * (r16 & 1);
            union { s4 v[2]; uint4 u; } vf;
            vf.v[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(vimg + TL::off(vrow, vchunk) + vsub));
            vf.v[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(vimg + TL::off(vrow + 16, vchunk) + vsub));
            oacc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vf.u), __builtin_bit_cast(bf16x8, pf), oacc[d], 0, 0, 0);
        }
    }
    // wave result: column g = r16; l summed over the four lanes of the column; O^T rows d = 16 dblk + 4 kq + j
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    if (kq == 0) {
        red[wave][r16][0] = m;
        red[wave][r16][1] = l;
    }
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[wave][r16][2 + d * 16 + 4 * kq + j] = oacc[d][j];
    __syncthreads();
    if (tid < DHP + 2 && c0 < len) {
        for (int g = 0; g < ng; ++g) {
            const float M = fmaxf(fmaxf(red[0][g][0], red[1][g][0]), fmaxf(red[2][g][0], red[3][g][0]));
            float v = 0.f, lsum = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float f = fast_exp2(red[w][g][0] - M);
                v += red[w][g][tid] * f;
                lsum += red[w][g][1] * f;
            }
            if (a.nsplit == 1 && a.out) {
                const int d = tid - 2;
                if (d >= 0 && d < a.dh) {
                    float o = v / lsum;
                    if (a.round_out) o = round_bf16(o);
                    a.out[(size_t)(row0 + g) * a.ldo + h * a.dh + d] = o;
                }
            } else if (fused_merge) {
                __hip_atomic_store(part_of(g) + tid, tid == 0 ? M : v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                part_of(g)[tid] = tid == 0 ? M : v;
            }
        }
    }
    if (fused_merge) {   // see decode_attn_kernel: write-through partials, one ticket per (first row of the tile, head)
        __shared__ int s_last;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            unsigned *cnt = a.tickets + (size_t)row0 * a.H + h;
            const unsigned t = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = t == (unsigned)(a.nsplit - 1);
            if (last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = last;
        }
        __syncthreads();
        if (s_last) {
            // wave w merges rows w, w + 4, ...; all partial triples of up to 8 splits are requested before any is used
            const int d = tid & 63;
            for (int g = tid >> 6; g < ng; g += 4) {
                float *pp = a.partial + ((size_t)(row0 + g) * a.H + h) * a.nsplit * (DHP + 2);
                auto ld = [&](int i) { return __hip_atomic_load(pp + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
                float M = -1.0e30f, ls = 0.f, o = 0.f;
                for (int s0 = 0; s0 < a.nsplit; s0 += 8) {
                    float pm[8], pl[8], po[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const bool in = s0 + u < a.nsplit;
                        const int base = (in ? s0 + u : s0) * (DHP + 2);
                        pm[u] = in ? ld(base) : -1.0e30f;
                        pl[u] = in ? ld(base + 1) : 0.f;
                        po[u] = in ? ld(base + 2 + d) : 0.f;
                    }
                    float Mc = M;
#pragma unroll
                    for (int u = 0; u < 8; ++u) Mc = fmaxf(Mc, pm[u]);
                    const float resc = fast_exp2(M - Mc);
                    ls *= resc;
                    o *= resc;
                    M = Mc;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float w = fast_exp2(pm[u] - M);
                        ls += pl[u] * w;
                        o += po[u] * w;
                    }
                }
                if (d < a.dh) {
                    float v = o / ls;
                    if (a.round_out) v = round_bf16(v);
                    a.out[(size_t)(row0 + g) * a.ldo + h * a.dh + d] = v;
                }
            }
        }
    }
}

// one wave per (b, h): out[b, h*dh + d] = sum_s o_s[d] 2^(m_s - M) / sum_s l_s 2^(m_s - M)
__global__ __launch_bounds__(64) void attn_combine_kernel(const float *partial, float *out, int ldo, int H, int dh, int dhp,
                                                          int nsplit, int round_out) {
    const int h = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
    const float *p = partial + ((size_t)b * H + h) * nsplit * (dhp + 2);
    float M = -1.0e30f;
    for (int s = 0; s < nsplit; ++s) M = fmaxf(M, p[s * (dhp + 2)]);
    float l = 0.f, o = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float w = fast_exp2(p[s * (dhp + 2)] - M);
        l += p[s * (dhp + 2) + 1] * w;
        if (d < dhp) o += p[s * (dhp + 2) + 2 + d] * w;
    }
    if (d < dh) {
        float v = o / l;
        if (round_out) v = round_bf16(v);  // SDPA output is bf16 under autocast
        out[(size_t)b * ldo + h * dh + d] = v;
    }
}

// x[b,:] = vocab_embedding[token_b] + pos_embedding[t]; token from `tokens` or seqs[b, t-1] (quirk Q1: M:576)
__global__ __launch_bounds__(256) void embed_kernel(const float *emb, const float *pos, const int64_t *tokens, const int64_t *seqs,
                                                    const int32_t *step, int max_len, float *x, int E) {
    const int b = blockIdx.x, t = step[0];
    const int64_t tok = tokens ? tokens[b] : seqs[(size_t)b * max_len + t - 1];
    for (int i = threadIdx.x; i < E; i += 256) x[(size_t)b * E + i] = emb[(size_t)tok * E + i] + pos[(size_t)t * E + i];
}

__global__ void set_step_kernel(int32_t *step, int t) { step[0] = t; }

// cached_get_next_token (M:579-581) + loop bookkeeping (M:606-611).  One workgroup, wave w takes rows w, w+4, ...
// With `emb`: the wave that chose row b's token also writes the NEXT step's input x[b] = vocab_embedding[token] + pos_embedding[t + 1]
// (quirk Q1: the token at index t is embedded with position t + 1, M:576), so a token step needs no embed launch of its own.
__global__ __launch_bounds__(1024) void argmax_logprob_kernel(const float *logits, int V, int B, int64_t *seqs, float *logprobs,
                                                             int max_len, int32_t *step, int32_t *finished, int eos, int round_lp,
                                                             int bookkeeping, const float *emb, const float *pos, float *x, int E, int Tmax) {
    __shared__ int unfinished[16];   // up to 16 waves: one row per wave for the usual batch sizes (the rows of a wave run back to back)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = step[0];
    int cnt = 0;
    const int nw = blockDim.x >> 6;
    for (int b = wave; b < B; b += nw) {
        const float *lg = logits + (size_t)b * V;
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = lane; i < V; i += 64) {
            const float v = lg[i];
            if (v > best) {  // strided scan keeps the lowest index per lane on ties
                best = v;
                bi = i;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {  // argmax, first index on ties (torch.argmax on CPU)
            const float ov = __shfl_xor(best, o);
            const int oi = __shfl_xor(bi, o);
            if (ov > best || (ov == best && oi < bi)) {
                best = ov;
                bi = oi;
            }
        }
        float se = 0.f;
        for (int i = lane; i < V; i += 64) se += expf(lg[i] - best);
        se = wave_sum(se);
        float lp = -logf(se);  // logit[argmax] - logsumexp
        if (round_lp) lp = round_bf16(lp);
        if (bookkeeping) {
            int fin = finished[b];
            if (bi == eos) fin = 1;
            if (lane == 0) {
                seqs[(size_t)b * max_len + t] = bi;
                logprobs[(size_t)b * max_len + t] = lp;
                finished[b] = fin;
            }
            cnt += fin ? 0 : 1;
            if (emb && t + 1 < Tmax)
                for (int i = lane * 4; i < E; i += 256) {
                    const float4 ev = *reinterpret_cast<const float4 *>(emb + (size_t)bi * E + i), pv = *reinterpret_cast<const float4 *>(pos + (size_t)(t + 1) * E + i);
                    *reinterpret_cast<float4 *>(x + (size_t)b * E + i) = make_float4(ev.x + pv.x, ev.y + pv.y, ev.z + pv.z, ev.w + pv.w);
                }
        } else if (lane == 0) {
            seqs[b] = bi;
            logprobs[b] = lp;
        }
    }
    if (lane == 0) unfinished[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < nw; ++w) tot += unfinished[w];
        if (bookkeeping) finished[B] = tot;
        step[0] = t + 1;
        step[1] = step[1] + 1;
    }
}

// GRPOViTOMR.cached_forward_rollout_policy (M:988-1049), one sampling step: top-k filter, softmax with temperature over the kept logits,
// draw from that distribution, log-prob of the drawn token under the UN-tempered softmax of the kept logits (the reference takes
// log_softmax(top_k_logits), M:1017).  torch.multinomial's Philox stream is not reproducible here; the draw is the inverse CDF of a caller
// supplied uniform u[b][t] over the kept logits in descending order (ties: lower vocabulary index first), so a step is a pure function of
// (logits, u) that the oracle restates.  One wave per row: k rounds of a wave-wide arg-max build the sorted top-k (k <= 64).
__global__ __launch_bounds__(256) void sample_logprob_kernel(const float *logits, int V, int B, int64_t *seqs, float *logprobs, int max_len,
                                                             const int32_t *step, int32_t *finished, int eos, int round_lp,
                                                             const float *uniforms, int top_k, float inv_temperature, const float *emb,
                                                             const float *pos, float *xnext, int E, int Tmax) {
    __shared__ float sv[4][64];
    __shared__ int si[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    const int t = step[0];
    const float *lg = logits + (size_t)b * V;
    // lane owns vocabulary entries lane, lane + 64, ... (V <= 512)
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (lane + 64 * j < V) ? lg[lane + 64 * j] : -INFINITY;
    const int k = min(top_k, V);
    for (int r = 0; r < k; ++r) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (v[j] > best) {   // ascending j = ascending index: first maximum wins
                best = v[j];
                bi = lane + 64 * j;
            }
        if (best == -INFINITY) bi = 0x7fffffff;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o);
            const int oi = __shfl_xor(bi, o);
            if (ov > best || (ov == best && oi < bi)) {
                best = ov;
                bi = oi;
            }
        }
        if ((bi & 63) == lane) {   // owner removes the winner
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (bi == lane + 64 * j) v[j] = -INFINITY;
        }
        if (lane == 0) {
            sv[wave][r] = best;
            si[wave][r] = bi;
        }
    }
    // same wave wrote and reads: LDS operations of a wave complete in order
    const bool in = lane < k;
    const float x = in ? sv[wave][lane] : -INFINITY, m = sv[wave][0];
    const float pT = in ? expf((x - m) * inv_temperature) : 0.f;   // softmax(top_k_logits / temperature), unnormalised
    const float p1 = in ? expf(x - m) : 0.f;                       // softmax(top_k_logits), unnormalised
    const float sumT = wave_sum(pT), sum1 = wave_sum(p1);
    float cdf = pT;                                                // inclusive prefix sum over the lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float up = __shfl_up(cdf, o);
        if (lane >= o) cdf += up;
    }
    const float target = uniforms[(size_t)b * max_len + t] * sumT;
    const unsigned long long hit = __ballot(in && cdf > target);
    const int r = hit ? __builtin_ctzll(hit) : k - 1;              // rounding at the top of the CDF: last kept entry
    const int tok = si[wave][r];
    float lp = (sv[wave][r] - m) - logf(sum1);
    if (round_lp) lp = round_bf16(lp);
    if (lane == 0) {
        seqs[(size_t)b * max_len + t] = tok;
        logprobs[(size_t)b * max_len + t] = lp;
        if (tok == eos) finished[b] = 1;
    }
    if (emb && t + 1 < Tmax)   // next step's input (see argmax_logprob_kernel)
        for (int i = lane * 4; i < E; i += 256) {
            const float4 ev = *reinterpret_cast<const float4 *>(emb + (size_t)tok * E + i), pv = *reinterpret_cast<const float4 *>(pos + (size_t)(t + 1) * E + i);
            *reinterpret_cast<float4 *>(xnext + (size_t)b * E + i) = make_float4(ev.x + pv.x, ev.y + pv.y, ev.z + pv.z, ev.w + pv.w);
        }
}

// loop bookkeeping after a sampling step: unfinished count, advance position and cache length
__global__ __launch_bounds__(64) void sample_bookkeeping_kernel(int B, int32_t *step, int32_t *finished) {
    int cnt = 0;
    for (int b = threadIdx.x; b < B; b += 64) cnt += finished[b] ? 0 : 1;
    cnt = (int)wave_sum((float)cnt);
    if (threadIdx.x == 0) {
        finished[B] = cnt;
        step[0] = step[0] + 1;
        step[1] = step[1] + 1;
    }
}

__global__ void advance_cache_kernel(int32_t *step) { step[1] = step[1] + 1; }

// rollout groups: B rows = B / group images x group rows (bf16, dhp = 64; needs the in-launch merge or a single split)
inline int launch_dattn_group(const DAttnArgs &a, int B, int group, hipStream_t st) {
    const int gtm = cdiv(group, 16);
    hipLaunchKernelGGL(decode_attn_gmfma_kernel, dim3(a.nsplit, a.H, (B / group) * gtm), dim3(256), 0, st, a, group, gtm);
    ACAI_LAUNCH_CHECK("decode_attn_gmfma");
    return 0;
}

template <typename TC>
int launch_dattn(const DAttnArgs &a, int B, hipStream_t st) {
    const int lpk = a.dhp * (int)sizeof(TC) / 16;
    static const int dattn_u = getenv("ACAI_DATTN_U") ? atoi(getenv("ACAI_DATTN_U")) : 2;   // key groups in flight per lane (A/B aid)
    dim3 grid(a.nsplit, a.H, B);
#define ACAI_DA(L)                                                                                        \
    case L:                                                                                               \
        if (a.seq_off && L == 8 && dattn_u == 4) hipLaunchKernelGGL((decode_attn_kernel<TC, 8, true, 4>), grid, dim3(256), 0, st, a);  \
        else if (a.seq_off && L == 8 && dattn_u == 3) hipLaunchKernelGGL((decode_attn_kernel<TC, 8, true, 3>), grid, dim3(256), 0, st, a);  \
        else if (a.seq_off) hipLaunchKernelGGL((decode_attn_kernel<TC, L, true>), grid, dim3(256), 0, st, a);  \
        else hipLaunchKernelGGL((decode_attn_kernel<TC, L, false>), grid, dim3(256), 0, st, a);           \
        break;
    switch (lpk) {
        ACAI_DA(1) ACAI_DA(2) ACAI_DA(4) ACAI_DA(8) ACAI_DA(16)
        default: return acai_set_err(-1, "decode_attn: dhp=%d unsupported", a.dhp);
    }
#undef ACAI_DA
    ACAI_LAUNCH_CHECK("decode_attn");
    return 0;
}

// The in-launch merge of the split partials (decode_attn_kernel: write-through stores, one agent-scope ticket per (sequence, head), the last
// arriver loads every partial) is a hand-off MEASURED on gfx950 / ROCm 7.2 with this kernel at TWO resident workgroups per CU - not an
// architectural guarantee (MI355X_MICROARCH.md, "Valid forms").  If a toolchain change moves the kernel's register count so that the residency
// is no longer the one it was validated at, the step falls back to the separate combine launch by itself (tickets ignored) instead of running
// the hand-off in a regime nobody tested.  ACAI_DATTN_MERGE=1 / 0 forces either path (A/B aid).
template <typename TC>
bool dattn_merge_validated(int dhp) {
    static int cached[5] = {-1, -1, -1, -1, -1};   // per lanes-per-key variant: 1, 2, 4, 8, 16
    static const int force = getenv("ACAI_DATTN_MERGE") ? atoi(getenv("ACAI_DATTN_MERGE")) : -1;
    if (force >= 0) return force != 0;
    const int lpk = dhp * (int)sizeof(TC) / 16;
    int idx = lpk == 1 ? 0 : lpk == 2 ? 1 : lpk == 4 ? 2 : lpk == 8 ? 3 : lpk == 16 ? 4 : -1;
    if (idx < 0) return false;
    if (cached[idx] < 0) {
        auto occ = [](int L, int &n) -> hipError_t {
            switch (L) {
#define ACAI_OCC(LL) case LL: return hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void *>(&decode_attn_kernel<TC, LL, true>), 256, 0);
                ACAI_OCC(1) ACAI_OCC(2) ACAI_OCC(4) ACAI_OCC(8) ACAI_OCC(16)
#undef ACAI_OCC
            }
            return hipErrorUnknown;
        };
        if (getenv("ACAI_DATTN_MERGE_DEBUG"))
            for (int L = 1; L <= 16; L *= 2) {
                int m = 0;
                const hipError_t e2 = occ(L, m);
                fprintf(stderr, "decode_attn_kernel<%d-byte cache, %d lanes per key>: hipOccupancyMaxActiveBlocksPerMultiprocessor = %d (err %d)\n", (int)sizeof(TC), L, m, (int)e2);
            }
        int n = 0;
        const hipError_t e = occ(lpk, n);
        // The residency this build was validated at (rounds 2-4: determinism test, 512-step soak, every decode parity test), as the occupancy
        // API reports it on gfx950 / ROCm 7.2: 7 workgroups of 256 threads per CU (the launch itself puts 2 on a CU: 512 workgroups).  Round 4's
        // first form of this check compared against 2, the API said 7, and the headline step silently took the separate combine launch
        // (0.712 against 0.689 ms) until the profile showed attn_combine_kernel back in it.
        cached[idx] = (e == hipSuccess && n >= 6 && n <= 8) ? 1 : 0;
    }
    return cached[idx] == 1;
}

// Host-side record of "d->x holds the chained step's input embedding" per decoder state (keyed by the x buffer): acai_decode_embed sets it,
// acai_decode_step / acai_decode_sample_step require it (they no longer embed by themselves: their input is what the previous step's argmax
// kernel wrote) and keep it, acai_decode_logits / acai_decode_hidden clear it (they overwrite x).  A C-ABI caller that mixes the stepwise and
// the chained entry points without re-embedding gets an argument error instead of a step on stale input.  (Replays of a captured graph do
// not pass through here: the capture-time call is what is checked - INTEGRATION.md.)
static std::mutex g_xv_mu;
static std::unordered_map<const void *, bool> g_x_valid;
static void x_valid_set(const AcaiDecoder *d, bool v) {
    std::lock_guard<std::mutex> lk(g_xv_mu);
    g_x_valid[d->x] = v;
}
static bool x_valid_get(const AcaiDecoder *d) {
    std::lock_guard<std::mutex> lk(g_xv_mu);
    auto it = g_x_valid.find(d->x);
    return it != g_x_valid.end() && it->second;
}

int check_decoder(const AcaiDecoder *d) {
    ACAI_CHECK_ARG(d && d->layers, "decoder: null descriptor");
    ACAI_CHECK_ARG(d->B > 0 && d->L > 0 && d->E == d->H * d->dh && d->dhp >= d->dh && d->dhp <= 64 && (d->dhp & (d->dhp - 1)) == 0 &&
                       d->dhp * (d->dtype == ACAI_BF16 ? 2 : 4) >= 16,
                   "decoder: bad dims B=%d L=%d E=%d H=%d dh=%d dhp=%d", d->B, d->L, d->E, d->H, d->dh, d->dhp);
    ACAI_CHECK_ARG(d->dtype == ACAI_F32 || d->dtype == ACAI_BF16, "decoder: bad dtype");
    ACAI_CHECK_ARG(d->self_chunk > 0 && d->cross_chunk > 0 && d->self_nsplit > 0 && d->cross_nsplit > 0 &&
                       (long)d->self_chunk * d->self_nsplit >= d->Tmax,
                   "decoder: attention split does not cover the cache");
    ACAI_CHECK_ARG(d->cross_off && d->cross_len && d->step && d->x && d->xn && d->qkv && d->attn && d->proj && d->hid && d->partial,
                   "decoder: null buffer");
    return 0;
}

template <typename TW>
int decode_core(const AcaiDecoder *d, const int64_t *tokens, hipStream_t st, bool do_embed = true, bool do_unembed = true) {
    const int B = d->B, E = d->E, H = d->H, F = d->F;
    const int rnd = (d->flags & ACAI_GEMM_ROUND_BF16) ? ACAI_GEMM_ROUND_BF16 : 0;
    const float sc = 1.4426950408889634f / sqrtf((float)d->dh);
    int rc;
    if (do_embed) {
        hipLaunchKernelGGL(embed_kernel, dim3(B), dim3(256), 0, st, d->emb, d->pos, tokens, d->seqs, d->step, d->max_len, d->x, E);
        ACAI_LAUNCH_CHECK("embed");
    }

    auto skinny = [&](const float *x, int ldx, const void *W, const float *bias, const float *res, float *y, int ldy, int N, int K,
                      int flags, const AcaiDecLayer *kvl) -> int {
        SkinnyArgs s{};
        s.x = x; s.W = W; s.bias = bias; s.residual = res; s.y = y;
        s.ldx = ldx; s.ldw = K; s.ldr = E; s.ldy = ldy; s.B = B; s.N = N; s.K = K; s.flags = flags;
        if (kvl) {
            s.k_cache = kvl->k_self; s.v_cache = kvl->v_self; s.step = d->step;
            s.E = E; s.H = H; s.dh = d->dh; s.dhp = d->dhp; s.Tmax = d->Tmax;
        }
        return launch_skinny<TW>(s, st);
    };
    auto attend = [&](const float *q, int ldq, const void *kc, const void *vc, bool cross) -> int {
        DAttnArgs a{};
        a.q = q; a.kc = kc; a.vc = vc; a.ldq = ldq; a.H = H; a.dh = d->dh; a.dhp = d->dhp; a.Tmax = d->Tmax;
        a.partial = d->partial; a.scale_log2e = sc;
        if (cross) {
            a.seq_off = d->cross_off; a.seq_len = d->cross_len; a.chunk = d->cross_chunk; a.nsplit = d->cross_nsplit;
        } else {
            a.step = d->step; a.chunk = d->self_chunk; a.nsplit = d->self_nsplit;
        }
        // rollout groups (bf16, d_h padded to 64): one K/V stream per image through the matrix-core kernel; otherwise the rows simply alias
        // the stored K/V through the per-row kernel (ACAI_DECODE_GROUP_KERNEL=0 forces that form: A/B aid)
        static const bool no_group = getenv("ACAI_DECODE_GROUP_KERNEL") && atoi(getenv("ACAI_DECODE_GROUP_KERNEL")) == 0;
        const int group = (!no_group && sizeof(TW) == 2 && d->dhp == 64 && cross && d->cross_group > 1 && B % d->cross_group == 0) ? d->cross_group : 1;
        if (a.nsplit == 1) {
            a.out = d->attn; a.ldo = E; a.round_out = rnd ? 1 : 0;
            return group > 1 ? launch_dattn_group(a, B, group, st) : launch_dattn<TW>(a, B, st);
        }
        if (d->tickets && (group > 1 || dattn_merge_validated<TW>(d->dhp))) {
            a.out = d->attn; a.ldo = E; a.round_out = rnd ? 1 : 0; a.tickets = d->tickets;
            return group > 1 ? launch_dattn_group(a, B, group, st) : launch_dattn<TW>(a, B, st);
        }
        int r = launch_dattn<TW>(a, B, st);
        if (r) return r;
        hipLaunchKernelGGL(attn_combine_kernel, dim3(H, B), dim3(64), 0, st, d->partial, d->attn, E, H, d->dh, d->dhp, a.nsplit, rnd ? 1 : 0);
        ACAI_LAUNCH_CHECK("attn_combine");
        return 0;
    };

    // Fused path (bf16, MFMA skinny GEMM): the residual stream is kept PRE-LayerNorm (z) and every consumer applies the
    // LayerNorm on load, so a layer is 6 GEMV + 2 attention (+2 combine) launches instead of 17.
    SkinnyArgs probe{};
    probe.x = d->x; probe.W = d->layers[0].self_in_w; probe.K = E; probe.ldw = E; probe.ldx = E;
    const bool fused = sizeof(TW) == 2 && d->stats && rnd && skinny_mfma_ok(probe) && (F % 256 == 0) && F <= SKM_MAXK;
    bool hid_bf16 = false, hid_in_bf16 = false;
    auto skinny_ln = [&](const float *x, int ldx, const void *W, const float *bias, float *y, int ldy, int N, int K, int flags,
                         const AcaiDecLayer *kvl, const float *lnw, const float *lnb, float *stats_out, const float *res,
                         const float *rlnw, const float *rlnb, const float *rstats) -> int {
        SkinnyArgs s{};
        s.x = x; s.W = W; s.bias = bias; s.residual = res; s.y = y;
        s.ldx = ldx; s.ldw = K; s.ldr = E; s.ldy = ldy; s.B = B; s.N = N; s.K = K; s.flags = flags;
        s.ln_w = lnw; s.ln_b = lnb; s.ln_eps = 1e-5f; s.stats_out = stats_out; s.rln_w = rlnw; s.rln_b = rlnb; s.rstats = rstats;
        s.y_bf16 = hid_bf16 && (flags & ACAI_GEMM_ROUND_BF16);
        s.x_bf16 = hid_in_bf16 && (flags & ACAI_GEMM_ROUND_BF16);
        if (kvl) {
            s.k_cache = kvl->k_self; s.v_cache = kvl->v_self; s.step = d->step;
            s.E = E; s.H = H; s.dh = d->dh; s.dhp = d->dhp; s.Tmax = d->Tmax;
        }
        return launch_skinny<TW>(s, st);
    };
    if (fused) {
        float *st0 = d->stats, *st1 = d->stats + 2 * B, *st2 = d->stats + 4 * B;
        float *zin = d->x, *z1 = d->proj, *z2 = d->xn;
        const float *lnw = nullptr, *lnb = nullptr;  // LayerNorm still to be applied to zin (norm3 of the previous layer)
        for (int l = 0; l < d->L; ++l) {
            const AcaiDecLayer *ly = d->layers + l;
            if ((rc = skinny_ln(zin, E, ly->self_in_w, ly->self_in_b, d->qkv, 3 * E, 3 * E, E, rnd, ly, lnw, lnb, st0, nullptr, nullptr, nullptr, nullptr))) return rc;
            if ((rc = attend(d->qkv, 3 * E, ly->k_self, ly->v_self, false))) return rc;
            if ((rc = skinny_ln(d->attn, E, ly->self_out_w, ly->self_out_b, z1, E, E, E, rnd, nullptr, nullptr, nullptr, nullptr, zin, lnw, lnb, st0))) return rc;
            if ((rc = skinny_ln(z1, E, ly->cross_q_w, ly->cross_q_b, d->qkv, 3 * E, E, E, rnd, nullptr, ly->n1_w, ly->n1_b, st1, nullptr, nullptr, nullptr, nullptr))) return rc;
            if ((rc = attend(d->qkv, 3 * E, ly->k_cross, ly->v_cross, true))) return rc;
            if ((rc = skinny_ln(d->attn, E, ly->cross_out_w, ly->cross_out_b, z2, E, E, E, rnd, nullptr, nullptr, nullptr, nullptr, z1, ly->n1_w, ly->n1_b, st1))) return rc;
            hid_bf16 = true;   // linear1 -> GELU output is bf16 under autocast anyway: store it as such (half the x bytes of linear2)
            if ((rc = skinny_ln(z2, E, ly->lin1_w, ly->lin1_b, d->hid, F, F, E, rnd | ACAI_GEMM_GELU, nullptr, ly->n2_w, ly->n2_b, st2, nullptr, nullptr, nullptr, nullptr))) return rc;
            hid_bf16 = false;
            hid_in_bf16 = true;
            if ((rc = skinny_ln(d->hid, F, ly->lin2_w, ly->lin2_b, zin, E, E, F, rnd, nullptr, nullptr, nullptr, nullptr, z2, ly->n2_w, ly->n2_b, st2))) return rc;
            hid_in_bf16 = false;
            lnw = ly->n3_w;
            lnb = ly->n3_b;
        }
        // x = norm3(z3) of the last layer, then the stack's final norm (eps 1e-6): both fused into the unembed GEMV's load when the chain
        // kernel takes it (E = 1024) - no stand-alone LayerNorm launch is left in a token step
        static const bool no_chain2 = getenv("ACAI_SKINNY_CHAIN") && atoi(getenv("ACAI_SKINNY_CHAIN")) == 0;
        if (do_unembed && d->fn_w && E == 1024 && !no_chain2 && lnw) {
            SkinnyArgs s{};
            s.x = zin; s.W = d->unembed_w; s.bias = d->unembed_b; s.y = d->logits;
            s.ldx = E; s.ldw = E; s.ldy = d->V; s.B = B; s.N = d->V; s.K = E; s.flags = rnd;
            s.ln_w = lnw; s.ln_b = lnb; s.ln_eps = 1e-5f;
            s.ln2_w = d->fn_w; s.ln2_b = d->fn_b; s.ln2_eps = 1e-6f;
            return launch_skinny<TW>(s, st);
        }
        if ((rc = acai_layernorm_fwd(zin, lnw, lnb, 1e-5f, d->proj, nullptr, B, E, st))) return rc;
        if (do_unembed && d->fn_w) {
            SkinnyArgs s{};
            s.x = d->proj; s.W = d->unembed_w; s.bias = d->unembed_b; s.y = d->logits;
            s.ldx = E; s.ldw = E; s.ldy = d->V; s.B = B; s.N = d->V; s.K = E; s.flags = rnd;
            s.ln_w = d->fn_w; s.ln_b = d->fn_b; s.ln_eps = 1e-6f;
            return launch_skinny<TW>(s, st);
        }
        if (d->fn_w) {
            if ((rc = acai_layernorm_fwd(d->proj, d->fn_w, d->fn_b, 1e-6f, d->xn, nullptr, B, E, st))) return rc;
        } else {
            hipError_t e = hipMemcpyAsync(d->xn, d->proj, sizeof(float) * (size_t)B * E, hipMemcpyDeviceToDevice, st);
            if (e != hipSuccess) return acai_set_err((int)e, "hipMemcpyAsync: %s", hipGetErrorString(e));
        }
    } else {
    for (int l = 0; l < d->L; ++l) {
        const AcaiDecLayer *ly = d->layers + l;
        // self attention (K:193-208)
        if ((rc = skinny(d->x, E, ly->self_in_w, ly->self_in_b, nullptr, d->qkv, 3 * E, 3 * E, E, rnd, ly))) return rc;
        if ((rc = attend(d->qkv, 3 * E, ly->k_self, ly->v_self, false))) return rc;
        if ((rc = skinny(d->attn, E, ly->self_out_w, ly->self_out_b, d->x, d->proj, E, E, E, rnd, nullptr))) return rc;
        if ((rc = acai_layernorm_fwd(d->proj, ly->n1_w, ly->n1_b, 1e-5f, d->x, nullptr, B, E, st))) return rc;
        // cross attention (K:212-220)
        if ((rc = skinny(d->x, E, ly->cross_q_w, ly->cross_q_b, nullptr, d->qkv, 3 * E, E, E, rnd, nullptr))) return rc;
        if ((rc = attend(d->qkv, 3 * E, ly->k_cross, ly->v_cross, true))) return rc;
        if ((rc = skinny(d->attn, E, ly->cross_out_w, ly->cross_out_b, d->x, d->proj, E, E, E, rnd, nullptr))) return rc;
        if ((rc = acai_layernorm_fwd(d->proj, ly->n2_w, ly->n2_b, 1e-5f, d->x, nullptr, B, E, st))) return rc;
        // feed forward (K:222)
        if ((rc = skinny(d->x, E, ly->lin1_w, ly->lin1_b, nullptr, d->hid, F, F, E, rnd | ACAI_GEMM_GELU, nullptr))) return rc;
        if ((rc = skinny(d->hid, F, ly->lin2_w, ly->lin2_b, d->x, d->proj, E, E, F, rnd, nullptr))) return rc;
        if ((rc = acai_layernorm_fwd(d->proj, ly->n3_w, ly->n3_b, 1e-5f, d->x, nullptr, B, E, st))) return rc;
    }
    if (d->fn_w) {
        if ((rc = acai_layernorm_fwd(d->x, d->fn_w, d->fn_b, 1e-6f, d->xn, nullptr, B, E, st))) return rc;
    } else {
        hipError_t e = hipMemcpyAsync(d->xn, d->x, sizeof(float) * (size_t)B * E, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return acai_set_err((int)e, "hipMemcpyAsync: %s", hipGetErrorString(e));
    }
    }
    if (do_unembed) {
        SkinnyArgs s{};
        s.x = d->xn; s.W = d->unembed_w; s.bias = d->unembed_b; s.y = d->logits;
        s.ldx = E; s.ldw = E; s.ldy = d->V; s.B = B; s.N = d->V; s.K = E; s.flags = rnd;
        if ((rc = launch_skinny<TW>(s, st))) return rc;
    }
    return 0;
}

}  // namespace

// Stand-alone entry points for the module-level API (CachedMultiheadAttention.cached_forward K:123-140,
// F.linear on (B,1,E) K:193,215): the same kernels acai_decode_step chains.
extern "C" int acai_skinny_gemm(const float *x, int ldx, const void *W, int ldw, const float *bias, const float *residual, int ldr,
                                float *y, int ldy, int B, int N, int K, int dtype, int flags, void *stream) {
    ACAI_CHECK_ARG(x && W && y && B > 0 && N > 0 && K > 0 && ldx >= K && ldw >= K && ldy >= N, "acai_skinny_gemm: bad arguments");
    SkinnyArgs s{};
    s.x = x; s.W = W; s.bias = bias; s.residual = residual; s.y = y;
    s.ldx = ldx; s.ldw = ldw; s.ldr = ldr; s.ldy = ldy; s.B = B; s.N = N; s.K = K; s.flags = flags;
    if (dtype == ACAI_BF16) return launch_skinny<bf16_t>(s, (hipStream_t)stream);
    if (dtype == ACAI_F32) return launch_skinny<float>(s, (hipStream_t)stream);
    return acai_set_err(-1, "acai_skinny_gemm: bad dtype %d", dtype);
}

// The full option set of the decode GEMV (what acai_decode_step chains): LayerNorm on load, published row statistics,
// LayerNorm of the residual, bf16 activations in / out.  Exposed so that tests and micro-benchmarks can hit each fusion.
extern "C" int acai_skinny_gemm_ex(const void *x, int ldx, int x_dtype, const void *W, int ldw, const float *bias, const float *residual,
                                   int ldr, void *y, int ldy, int y_dtype, int B, int N, int K, int dtype, int flags, const float *ln_w,
                                   const float *ln_b, float ln_eps, float *stats_out, const float *rln_w, const float *rln_b,
                                   const float *rstats, void *stream) {
    ACAI_CHECK_ARG(x && W && y && B > 0 && N > 0 && K > 0 && ldx >= K && ldw >= K && ldy >= N, "acai_skinny_gemm_ex: bad arguments");
    ACAI_CHECK_ARG(!rln_w || (rln_b && rstats && residual), "acai_skinny_gemm_ex: residual LayerNorm needs weights, bias, statistics and a residual");
    SkinnyArgs s{};
    s.x = (const float *)x; s.W = W; s.bias = bias; s.residual = residual; s.y = (float *)y;
    s.ldx = ldx; s.ldw = ldw; s.ldr = ldr; s.ldy = ldy; s.B = B; s.N = N; s.K = K; s.flags = flags;
    s.ln_w = ln_w; s.ln_b = ln_b; s.ln_eps = ln_eps; s.stats_out = stats_out; s.rln_w = rln_w; s.rln_b = rln_b; s.rstats = rstats;
    s.x_bf16 = x_dtype == ACAI_BF16; s.y_bf16 = y_dtype == ACAI_BF16;
    if (dtype == ACAI_BF16) return launch_skinny<bf16_t>(s, (hipStream_t)stream);
    if (dtype == ACAI_F32) return launch_skinny<float>(s, (hipStream_t)stream);
    return acai_set_err(-1, "acai_skinny_gemm_ex: bad dtype %d", dtype);
}

extern "C" int acai_decode_attn(const float *q, int ldq, const void *kc, const void *vc, const int64_t *seq_off, const int32_t *seq_len,
                                float *partial, float *out, int ldo, int B, int H, int dh, int dhp, int chunk, int nsplit, int dtype,
                                int round_out, uint32_t *tickets, void *stream) {
    ACAI_CHECK_ARG(q && kc && vc && seq_off && seq_len && partial, "acai_decode_attn: null operand");
    ACAI_CHECK_ARG(B > 0 && H > 0 && dh > 0 && dhp >= dh && dhp <= 64 && (dhp & (dhp - 1)) == 0 && chunk > 0 && nsplit > 0,
                   "acai_decode_attn: bad dims");
    DAttnArgs a{};
    a.q = q; a.kc = kc; a.vc = vc; a.seq_off = seq_off; a.seq_len = seq_len; a.partial = partial;
    a.ldq = ldq; a.H = H; a.dh = dh; a.dhp = dhp; a.chunk = chunk; a.nsplit = nsplit;
    a.scale_log2e = 1.4426950408889634f / sqrtf((float)dh);
    hipStream_t st = (hipStream_t)stream;
    if (tickets && out) {
        a.out = out; a.ldo = ldo; a.round_out = round_out; a.tickets = tickets;
        return dtype == ACAI_BF16 ? launch_dattn<bf16_t>(a, B, st) : launch_dattn<float>(a, B, st);
    }
    int rc = dtype == ACAI_BF16 ? launch_dattn<bf16_t>(a, B, st) : launch_dattn<float>(a, B, st);
    if (rc || !out) return rc;  // out == NULL: partials only (lets a benchmark time the streaming kernel alone)
    hipLaunchKernelGGL(attn_combine_kernel, dim3(H, B), dim3(64), 0, st, partial, out, ldo, H, dh, dhp, nsplit, round_out);
    ACAI_LAUNCH_CHECK("attn_combine");
    return 0;
}

// Diagnostic: from now on every MFMA skinny launch (up to cap_launches, 1024 workgroups each) writes s_memrealtime stamps of its stages
// into buf[launch][workgroup][8]; buf = NULL switches it off and rewinds the slot counter.  Not part of the product path.
extern "C" int acai_debug_stamps(void *buf, int cap_launches) {
    g_stamps = (unsigned long long *)buf;
    g_stamp_cap = buf ? cap_launches : 0;
    g_stamp_next = 0;
    return 0;
}

extern "C" int acai_decode_merge_in_launch(int dtype, int dhp) {
    ACAI_CHECK_ARG((dtype == ACAI_BF16 || dtype == ACAI_F32) && dhp > 0 && dhp <= 64 && (dhp & (dhp - 1)) == 0, "acai_decode_merge_in_launch: bad dtype / dhp");
    return dtype == ACAI_BF16 ? (dattn_merge_validated<bf16_t>(dhp) ? 1 : 0) : (dattn_merge_validated<float>(dhp) ? 1 : 0);
}

extern "C" int acai_decode_hidden(const AcaiDecoder *d, const float *x_in, void *stream) {
    int rc = check_decoder(d);
    if (rc) return rc;
    ACAI_CHECK_ARG(x_in, "acai_decode_hidden: null input");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemcpyAsync(d->x, x_in, sizeof(float) * (size_t)d->B * d->E, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return acai_set_err((int)e, "hipMemcpyAsync: %s", hipGetErrorString(e));
    rc = d->dtype == ACAI_BF16 ? decode_core<bf16_t>(d, nullptr, st, false, false) : decode_core<float>(d, nullptr, st, false, false);
    if (rc) return rc;
    hipLaunchKernelGGL(advance_cache_kernel, dim3(1), dim3(1), 0, st, d->step);
    ACAI_LAUNCH_CHECK("advance_cache");
    x_valid_set(d, false);
    return 0;
}

// x = vocab_embedding[seqs[:, t-1]] + pos_embedding[t] for the armed state (t = step[0]): run once after arming; every later step's input is
// written by the previous step's argmax / sampling kernel.
extern "C" int acai_decode_embed(const AcaiDecoder *d, void *stream) {
    int rc = check_decoder(d);
    if (rc) return rc;
    ACAI_CHECK_ARG(d->emb && d->pos && d->seqs && d->max_len > 1, "acai_decode_embed: decoder has no embedding / sequence state");
    hipLaunchKernelGGL(embed_kernel, dim3(d->B), dim3(256), 0, (hipStream_t)stream, (const float *)d->emb, (const float *)d->pos, (const int64_t *)nullptr,
                       (const int64_t *)d->seqs, (const int32_t *)d->step, d->max_len, d->x, d->E);
    ACAI_LAUNCH_CHECK("embed");
    x_valid_set(d, true);
    return 0;
}

extern "C" int acai_decode_step(const AcaiDecoder *d, void *stream) {
    int rc = check_decoder(d);
    if (rc) return rc;
    ACAI_CHECK_ARG(d->emb && d->pos && d->unembed_w && d->logits, "acai_decode_step: decoder has no embedding / unembed");
    ACAI_CHECK_ARG(d->seqs && d->logprobs && d->finished && d->max_len > 1, "acai_decode_step: null sequence state");
    hipStream_t st = (hipStream_t)stream;
    // the step's input x was written by the previous step's argmax (or by acai_decode_embed after arming) when E allows 16-byte rows
    const bool chained = (d->E % 4 == 0);
    ACAI_CHECK_ARG(!chained || x_valid_get(d), "acai_decode_step: x does not hold this step's input embedding - call acai_decode_embed after arming the "
                                               "sequence state and after every acai_decode_logits / acai_decode_hidden");
    rc = d->dtype == ACAI_BF16 ? decode_core<bf16_t>(d, nullptr, st, !chained) : decode_core<float>(d, nullptr, st, !chained);
    if (rc) return rc;
    hipLaunchKernelGGL(argmax_logprob_kernel, dim3(1), dim3(d->B > 8 ? 1024 : (d->B > 4 ? 512 : 256)), 0, st, d->logits, d->V, d->B, d->seqs, d->logprobs, d->max_len, d->step,
                       d->finished, d->eos, (d->flags & ACAI_GEMM_ROUND_BF16) ? 1 : 0, 1, chained ? (const float *)d->emb : nullptr, (const float *)d->pos, d->x, d->E, d->Tmax);
    ACAI_LAUNCH_CHECK("argmax_logprob");
    return 0;
}

extern "C" int acai_decode_sample_step(const AcaiDecoder *d, const float *uniforms, int top_k, float temperature, void *stream) {
    int rc = check_decoder(d);
    if (rc) return rc;
    ACAI_CHECK_ARG(d->emb && d->pos && d->unembed_w && d->logits, "acai_decode_sample_step: decoder has no embedding / unembed");
    ACAI_CHECK_ARG(d->seqs && d->logprobs && d->finished && d->max_len > 1, "acai_decode_sample_step: null sequence state");
    ACAI_CHECK_ARG(uniforms && top_k >= 1 && top_k <= 64 && temperature > 0.f && d->V <= 512,
                   "acai_decode_sample_step: needs uniforms, 1 <= top_k <= 64, temperature > 0, vocabulary <= 512 (top_k=%d V=%d)", top_k, d->V);
    hipStream_t st = (hipStream_t)stream;
    const bool chained = (d->E % 4 == 0);
    ACAI_CHECK_ARG(!chained || x_valid_get(d), "acai_decode_sample_step: x does not hold this step's input embedding - call acai_decode_embed after arming "
                                               "the sequence state and after every acai_decode_logits / acai_decode_hidden");
    rc = d->dtype == ACAI_BF16 ? decode_core<bf16_t>(d, nullptr, st, !chained) : decode_core<float>(d, nullptr, st, !chained);
    if (rc) return rc;
    hipLaunchKernelGGL(sample_logprob_kernel, dim3(cdiv(d->B, 4)), dim3(256), 0, st, d->logits, d->V, d->B, d->seqs, d->logprobs, d->max_len, d->step,
                       d->finished, d->eos, (d->flags & ACAI_GEMM_ROUND_BF16) ? 1 : 0, uniforms, top_k, 1.0f / temperature,
                       chained ? (const float *)d->emb : nullptr, (const float *)d->pos, d->x, d->E, d->Tmax);
    hipLaunchKernelGGL(sample_bookkeeping_kernel, dim3(1), dim3(64), 0, st, d->B, d->step, d->finished);
    ACAI_LAUNCH_CHECK("sample_logprob");
    return 0;
}

extern "C" int acai_decode_logits(const AcaiDecoder *d, const int64_t *tokens, int time_step, void *stream) {
    int rc = check_decoder(d);
    if (rc) return rc;
    ACAI_CHECK_ARG(d->emb && d->pos && d->unembed_w && d->logits, "acai_decode_logits: decoder has no embedding / unembed");
    ACAI_CHECK_ARG(tokens && time_step >= 0 && time_step < d->Tmax, "acai_decode_logits: bad tokens / time_step %d", time_step);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(set_step_kernel, dim3(1), dim3(1), 0, st, d->step, time_step);
    rc = d->dtype == ACAI_BF16 ? decode_core<bf16_t>(d, tokens, st) : decode_core<float>(d, tokens, st);
    if (rc) return rc;
    hipLaunchKernelGGL(advance_cache_kernel, dim3(1), dim3(1), 0, st, d->step);
    ACAI_LAUNCH_CHECK("advance_cache");
    x_valid_set(d, false);
    return 0;
}
