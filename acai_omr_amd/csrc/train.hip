// Row kernels of the backward pass and the two losses (HBM-bound, fp32 math):
//   LayerNorm backward (autograd of nn.LayerNorm, models.py:33 and the norm1/2/3 of every block), exact-erf GELU forward /
//   backward as stand-alone passes (training keeps the pre-activation), column sums (bias gradients), row scatter-add
//   (gradients of nn.Embedding, pos_embedding slices, MAE shuffle), MAELoss (models.py:271-288) and OMRCELoss
//   (models.py:784-796) forward + backward in one pass each.
#include "common.h"
#include <algorithm>

namespace {

// ---- LayerNorm backward ------------------------------------------------------------------------------------------------
// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * w,  xhat = (x - mean) * rstd;  stats[row] = (mean, rstd) for the
// parameter-gradient pass.  One wave per row, statistics recomputed from x (two-pass, as the forward).
__global__ __launch_bounds__(256) void ln_bwd_dx_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ dy,
                                                        float eps, float *dx, float *stats, int rows, int dim) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + (size_t)row * dim, *gr = dy + (size_t)row * dim;
    float s = 0.f;
    for (int i = lane; i < dim; i += 64) s += xr[i];
    const float mean = wave_sum(s) / (float)dim;
    float q = 0.f;
    for (int i = lane; i < dim; i += 64) {
        const float d = xr[i] - mean;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)dim + eps);
    float sg = 0.f, sgx = 0.f;
    for (int i = lane; i < dim; i += 64) {
        const float g = gr[i] * w[i], xh = (xr[i] - mean) * rstd;
        sg += g;
        sgx += g * xh;
    }
    sg = wave_sum(sg) / (float)dim;
    sgx = wave_sum(sgx) / (float)dim;
    for (int i = lane; i < dim; i += 64) {
        const float g = gr[i] * w[i], xh = (xr[i] - mean) * rstd;
        dx[(size_t)row * dim + i] = rstd * (g - sg - xh * sgx);
    }
    if (lane == 0) {
        stats[row * 2] = mean;
        stats[row * 2 + 1] = rstd;
    }
}

// dw[c] += sum_r dy[r,c] * xhat[r,c], db[c] += sum_r dy[r,c]; grid (col tiles of 64, row chunks), fp32 atomics per workgroup.
__global__ __launch_bounds__(256) void ln_bwd_param_kernel(const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ stats,
                                                           float *dw, float *db, int rows, int dim, int rows_per_block) {
    __shared__ float sw[4][64], sb[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float aw = 0.f, ab = 0.f;
    if (c < dim)
        for (int r = r0 + wave; r < r1; r += 4) {
            const float g = dy[(size_t)r * dim + c];
            aw += g * (x[(size_t)r * dim + c] - stats[r * 2]) * stats[r * 2 + 1];
            ab += g;
        }
    sw[wave][lane] = aw;
    sb[wave][lane] = ab;
    __syncthreads();
    if (wave == 0 && c < dim) {
        atomicAdd(dw + c, sw[0][lane] + sw[1][lane] + sw[2][lane] + sw[3][lane]);
        atomicAdd(db + c, sb[0][lane] + sb[1][lane] + sb[2][lane] + sb[3][lane]);
    }
}

// Fused form for dim = NV * 256 (the path's 512 / 768 / 1024): one pass over x and dy.  A wave owns a row in registers (lane = 4 consecutive
// columns per 256-column slab: 1 KiB per wave instruction), writes dx (fp32 and, optionally, the bf16 copy the next GEMM reads) and keeps
// per-lane partial dw / db over all rows of its chunk; the four waves' partials meet in LDS and leave as one atomic per column and workgroup.
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_fused_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ dy, float eps,
                                                           float *__restrict__ dx, bf16_t *__restrict__ dx_bf16, float *dw, float *db, float *dxsum,
                                                           int rows, int rows_per_block) {
    // dxsum (optional): column sums of dx as the consuming Linear sees it (the bf16 copy when one is written) = that Linear's bias gradient,
    // formed here while dx is in registers instead of by a colsum launch that reads dx back from HBM
    constexpr int DIM = NV * 256;
    __shared__ float red[4][DIM];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    f32x4 wv[NV], aw[NV], ab[NV], ac[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        wv[j] = *reinterpret_cast<const f32x4 *>(w + j * 256 + lane * 4);
        aw[j] = ab[j] = ac[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float inv_dim = 1.0f / (float)DIM;
    for (int r = r0 + wave; r < r1; r += 4) {
        f32x4 xv[NV], gv[NV];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            xv[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(x + (size_t)r * DIM + j * 256 + lane * 4));
            gv[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(dy + (size_t)r * DIM + j * 256 + lane * 4));
            s += (xv[j][0] + xv[j][1]) + (xv[j][2] + xv[j][3]);
        }
        const float mean = wave_sum(s) * inv_dim;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xv[j][e] -= mean;
                q += xv[j][e] * xv[j][e];
            }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * inv_dim + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = xv[j][e] * rstd, d = gv[j][e], g = d * wv[j][e];
                xv[j][e] = xh;
                aw[j][e] += d * xh;
                ab[j][e] += d;
                gv[j][e] = g;
                sg += g;
                sgx += g * xh;
            }
        sg = wave_sum(sg) * inv_dim;
        sgx = wave_sum(sgx) * inv_dim;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rstd * (gv[j][e] - sg - xv[j][e] * sgx);
            *reinterpret_cast<f32x4 *>(dx + (size_t)r * DIM + j * 256 + lane * 4) = o;
            if (dx_bf16) {
                uint2 pk;
                pk.x = pack_bf16(o[0], o[1]);
                pk.y = pack_bf16(o[2], o[3]);
                *reinterpret_cast<uint2 *>(dx_bf16 + (size_t)r * DIM + j * 256 + lane * 4) = pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) ac[j][e] += round_bf16(o[e]);
            } else {
                ac[j] += o;
            }
        }
    }
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
        float *dst = pass == 0 ? dw : (pass == 1 ? db : dxsum);
        if (!dst) continue;  // uniform
#pragma unroll
        for (int j = 0; j < NV; ++j) *reinterpret_cast<f32x4 *>(&red[wave][j * 256 + lane * 4]) = pass == 0 ? aw[j] : (pass == 1 ? ab[j] : ac[j]);
        __syncthreads();
        for (int c = threadIdx.x; c < DIM; c += 256) atomicAdd(dst + c, (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]));
        __syncthreads();
    }
}

// ---- GELU ----------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const T *__restrict__ a, T *h, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) DT<T>::st(h + i, gelu_erf(DT<T>::ld(a + i)));
}
template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T *__restrict__ a, const T *__restrict__ dh, T *da, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float x = DT<T>::ld(a + i);
        const float g = 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * expf(-0.5f * x * x);
        DT<T>::st(da + i, DT<T>::ld(dh + i) * g);
    }
}

// ---- column sum: out[c] += sum_r x[r,c] ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T *__restrict__ x, int ld, float *out, int rows, int cols, int rows_per_block) {
    __shared__ float s[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float acc = 0.f;
    if (c < cols)
        for (int r = r0 + wave; r < r1; r += 4) acc += DT<T>::ld(x + (size_t)r * ld + c);
    s[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && c < cols) atomicAdd(out + c, s[0][lane] + s[1][lane] + s[2][lane] + s[3][lane]);
}

// 16-byte form: a lane owns EPC consecutive columns, a wave 64*EPC columns of one row (1 KiB per wave instruction), the four waves interleave
// rows, four rows in flight per lane.
template <typename T>
__global__ __launch_bounds__(256) void colsum_vec_kernel(const T *__restrict__ x, int ld, float *out, int rows, int cols, int rows_per_block) {
    constexpr int EPC = 16 / sizeof(T);
    __shared__ float s[4][64 * EPC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = (blockIdx.x * 64 + lane) * EPC;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
    if (c0 < cols) {
        const T *p = x + c0;
        auto add = [&](const uint4 &v) {
            union { uint4 u; T e[EPC]; } t;
            t.u = v;
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] += DT<T>::ld(&t.e[e]);
        };
        int r = r0 + wave;
        for (; r + 12 < r1; r += 16) {
            const uint4 v0 = ld_nt16(p + (size_t)r * ld);
            const uint4 v1 = ld_nt16(p + (size_t)(r + 4) * ld);
            const uint4 v2 = ld_nt16(p + (size_t)(r + 8) * ld);
            const uint4 v3 = ld_nt16(p + (size_t)(r + 12) * ld);
            add(v0); add(v1); add(v2); add(v3);
        }
        for (; r < r1; r += 4) add(*reinterpret_cast<const uint4 *>(p + (size_t)r * ld));
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) s[wave][lane * EPC + e] = acc[e];
    __syncthreads();
    for (int c = threadIdx.x; c < 64 * EPC; c += 256) {
        const int col = blockIdx.x * 64 * EPC + c;
        if (col < cols) atomicAdd(out + col, (s[0][c] + s[1][c]) + (s[2][c] + s[3][c]));
    }
}

// ---- dst[idx[r], :] += src[r, :] -------------------------------------------------------------------------------------------
// shared >= 0: only row index `shared` may occur more than once (the MAE mask token, models.py:219-241): every other row is a plain
// read-modify-write (no atomics: the fp32 atomic path moves ~1 TB/s, a third of the plain rate), the shared row goes through a block-level
// column sum and one atomic per column and workgroup.
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float *__restrict__ src, const int32_t *__restrict__ idx, float *dst, int rows, int dim,
                                                               int shared) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (shared < 0) {
        const int row = blockIdx.x * 4 + wave;
        if (row >= rows) return;
        const float *s = src + (size_t)row * dim;
        float *d = dst + (size_t)idx[row] * dim;
        for (int i = lane; i < dim; i += 64) atomicAdd(d + i, s[i]);
        return;
    }
    // 64 rows per workgroup, 16 per wave
    __shared__ float red[4][1024];
    const int r0 = blockIdx.x * 64 + wave * 16, nv = dim / 4;   // dim % 4 == 0, dim <= 1024 * ... handled by the column loop
    for (int c4 = lane; c4 < nv; c4 += 64) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int r = r0; r < min(rows, r0 + 16); ++r) {
            const int j = idx[r];
            const f32x4 v = *reinterpret_cast<const f32x4 *>(src + (size_t)r * dim + c4 * 4);
            if (j == shared) {
                acc += v;
            } else {
                f32x4 *d = reinterpret_cast<f32x4 *>(dst + (size_t)j * dim + c4 * 4);
                *d = *d + v;
            }
        }
        if (c4 * 4 < 1024) *reinterpret_cast<f32x4 *>(&red[wave][c4 * 4]) = acc;
        else
            for (int e = 0; e < 4; ++e) atomicAdd(dst + (size_t)shared * dim + c4 * 4 + e, acc[e]);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < min(dim, 1024); c += 256) {
        const float v = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
        if (v != 0.f) atomicAdd(dst + (size_t)shared * dim + c, v);
    }
}

// ---- MAELoss fwd + bwd: loss += mask * mean_d((pred - that)^2) * inv_count, dpred = 2 (pred - that) / D * mask * inv_count * gscale
__global__ __launch_bounds__(256) void mae_loss_kernel(const float *__restrict__ pred, const float *__restrict__ target, const unsigned char *__restrict__ mask,
                                                       float inv_count, float *loss, float *dpred, int rows, int dim, int rows_per_wave) {
    // A wave walks `rows_per_wave` rows and keeps its loss share in a register; the four waves meet in LDS and the workgroup issues ONE
    // atomic (one atomic per row on a single address serialised ~1e5 of them: 1.26 ms of a 0.1 ms kernel).
    __shared__ float part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = (blockIdx.x * 4 + wave) * rows_per_wave, r1 = min(rows, r0 + rows_per_wave);
    float total = 0.f;
    for (int row = r0; row < r1; ++row) {
        const float *p = pred + (size_t)row * dim, *t = target + (size_t)row * dim;
        if (mask[row] == 0) {
            if (dpred)
                for (int i = lane; i < dim; i += 64) dpred[(size_t)row * dim + i] = 0.f;
            continue;
        }
        float s = 0.f;
        for (int i = lane; i < dim; i += 64) s += t[i];
        const float mean = wave_sum(s) / (float)dim;
        float q = 0.f;
        for (int i = lane; i < dim; i += 64) {
            const float d = t[i] - mean;
            q += d * d;
        }
        const float var = wave_sum(q) / (float)(dim - 1);  // Tensor.var default: unbiased (models.py:281)
        const float rs = 1.0f / sqrtf(var + 1.0e-6f);
        float e = 0.f;
        for (int i = lane; i < dim; i += 64) {
            const float d = p[i] - (t[i] - mean) * rs;
            e += d * d;
            if (dpred) dpred[(size_t)row * dim + i] = 2.0f * d / (float)dim * inv_count;
        }
        total += wave_sum(e) / (float)dim;
    }
    if (lane == 0) part[wave] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = (part[0] + part[1]) + (part[2] + part[3]);
        if (v != 0.f) atomicAdd(loss, v * inv_count);
    }
}

// ---- OMRCELoss fwd + bwd: rows with target == ignore contribute nothing; mean over the others ------------------------------
// label smoothing eps (nn.CrossEntropyLoss: row loss = (1 - eps) * nll + eps / V * sum_c -log p_c; d/dlogit_c = p_c - (1 - eps) [c == t] - eps / V)
__global__ __launch_bounds__(256) void ce_loss_kernel(const float *__restrict__ logits, int ld, const int64_t *__restrict__ target, int ignore,
                                                      float inv_count, float *loss, float *dlogits, int rows, int V, float eps) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *lg = logits + (size_t)row * ld;
    const int64_t tg = target[row];
    if (tg == ignore) {
        if (dlogits)
            for (int i = lane; i < V; i += 64) dlogits[(size_t)row * V + i] = 0.f;
        return;
    }
    float m = -INFINITY;
    for (int i = lane; i < V; i += 64) m = fmaxf(m, lg[i]);
    m = wave_max(m);
    float se = 0.f, sl = 0.f;
    for (int i = lane; i < V; i += 64) {
        se += expf(lg[i] - m);
        sl += lg[i];
    }
    se = wave_sum(se);
    const float lse = m + logf(se);
    const float uni = eps / (float)V;
    if (dlogits)
        for (int i = lane; i < V; i += 64) dlogits[(size_t)row * V + i] = (expf(lg[i] - lse) - (i == tg ? 1.f - eps : 0.f) - uni) * inv_count;
    if (eps != 0.f) sl = wave_sum(sl);
    if (lane == 0) {
        float v = (1.f - eps) * (lse - lg[tg]);
        if (eps != 0.f) v += eps * (lse - sl / (float)V);
        atomicAdd(loss, v * inv_count);
    }
}

// out = residual + keep(row, col) * x / (1 - p)   (nn.Dropout after a projection, then the residual add; also its own backward:
// call it on dy with residual = NULL).  x in T, residual / out fp32 or T.
template <typename T, typename TO>
__global__ __launch_bounds__(256) void dropout_add_kernel(const T *__restrict__ x, const float *__restrict__ res, TO *out, int rows, int cols,
                                                          uint32_t thr, float scale, uint32_t seed) {
    const long n = (long)rows * cols;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int r = (int)(i / cols), c = (int)(i - (long)r * cols);
        float v = drop_keep(seed, (uint32_t)r, (uint32_t)c, thr) ? DT<T>::ld(x + i) * scale : 0.f;
        if (res) v += res[i];
        DT<TO>::st(out + i, v);
    }
}

static inline int grid1d(long n) {
    long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

extern "C" int acai_layernorm_bwd(const float *x, const float *w, const float *dy, float eps, float *dx, void *dx_bf16, float *dw, float *db,
                                  float *dxsum, float *stats, int rows, int dim, void *stream) {
    ACAI_CHECK_ARG(x && w && dy && dx && stats && rows >= 0 && dim > 0, "acai_layernorm_bwd: bad arguments");
    ACAI_CHECK_ARG((dw == nullptr) == (db == nullptr), "acai_layernorm_bwd: dw and db come together");
    if (rows == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (dim % 256 == 0 && dim <= 1024 && aligned16(x) && aligned16(dy) && aligned16(w) && aligned16(dx) && (!dx_bf16 || aligned16(dx_bf16))) {
        const int rpb = std::max(4, cdiv(cdiv(rows, 2048), 4) * 4);   // ~2048 workgroups, whole groups of 4 rows (one per wave)
        const dim3 grid(cdiv(rows, rpb));
        bf16_t *xb = (bf16_t *)dx_bf16;
        switch (dim / 256) {
            case 1: hipLaunchKernelGGL(ln_bwd_fused_kernel<1>, grid, dim3(256), 0, st, x, w, dy, eps, dx, xb, dw, db, dxsum, rows, rpb); break;
            case 2: hipLaunchKernelGGL(ln_bwd_fused_kernel<2>, grid, dim3(256), 0, st, x, w, dy, eps, dx, xb, dw, db, dxsum, rows, rpb); break;
            case 3: hipLaunchKernelGGL(ln_bwd_fused_kernel<3>, grid, dim3(256), 0, st, x, w, dy, eps, dx, xb, dw, db, dxsum, rows, rpb); break;
            default: hipLaunchKernelGGL(ln_bwd_fused_kernel<4>, grid, dim3(256), 0, st, x, w, dy, eps, dx, xb, dw, db, dxsum, rows, rpb); break;
        }
        ACAI_LAUNCH_CHECK("acai_layernorm_bwd");
        return 0;
    }
    ACAI_CHECK_ARG(!dx_bf16 && !dxsum, "acai_layernorm_bwd: the bf16 copy / column sums of dx need dim %% 256 == 0, dim <= 1024 and 16-byte aligned operands");
    hipLaunchKernelGGL(ln_bwd_dx_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, x, w, dy, eps, dx, stats, rows, dim);
    if (dw && db) {
        const int rpb = 2048;
        hipLaunchKernelGGL(ln_bwd_param_kernel, dim3(cdiv(dim, 64), cdiv(rows, rpb)), dim3(256), 0, st, x, dy, stats, dw, db, rows, dim, rpb);
    }
    ACAI_LAUNCH_CHECK("acai_layernorm_bwd");
    return 0;
}

extern "C" int acai_gelu_fwd(const void *a, void *h, int64_t n, int dtype, void *stream) {
    ACAI_CHECK_ARG(a && h && n >= 0, "acai_gelu_fwd: bad arguments");
    if (n == 0) return 0;
    if (dtype == ACAI_BF16) hipLaunchKernelGGL(gelu_fwd_kernel<bf16_t>, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)a, (bf16_t *)h, (long)n);
    else hipLaunchKernelGGL(gelu_fwd_kernel<float>, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, (const float *)a, (float *)h, (long)n);
    ACAI_LAUNCH_CHECK("acai_gelu_fwd");
    return 0;
}

extern "C" int acai_gelu_bwd(const void *a, const void *dh, void *da, int64_t n, int dtype, void *stream) {
    ACAI_CHECK_ARG(a && dh && da && n >= 0, "acai_gelu_bwd: bad arguments");
    if (n == 0) return 0;
    if (dtype == ACAI_BF16)
        hipLaunchKernelGGL(gelu_bwd_kernel<bf16_t>, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)a, (const bf16_t *)dh, (bf16_t *)da, (long)n);
    else
        hipLaunchKernelGGL(gelu_bwd_kernel<float>, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, (const float *)a, (const float *)dh, (float *)da, (long)n);
    ACAI_LAUNCH_CHECK("acai_gelu_bwd");
    return 0;
}

extern "C" int acai_colsum(const void *x, int ld, float *out, int rows, int cols, int dtype, void *stream) {
    ACAI_CHECK_ARG(x && out && rows >= 0 && cols > 0 && ld >= cols, "acai_colsum: bad arguments");
    if (rows == 0) return 0;
    const int es = dtype == ACAI_BF16 ? 2 : 4, epc = 16 / es;
    if (cols % epc == 0 && ld % epc == 0 && aligned16(x)) {
        const int gx = cdiv(cols, 64 * epc);
        const int rpb = std::max(64, cdiv(cdiv(rows, std::max(1, 2048 / gx)), 16) * 16);
        dim3 grid(gx, cdiv(rows, rpb));
        if (dtype == ACAI_BF16) hipLaunchKernelGGL(colsum_vec_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t *)x, ld, out, rows, cols, rpb);
        else hipLaunchKernelGGL(colsum_vec_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float *)x, ld, out, rows, cols, rpb);
        ACAI_LAUNCH_CHECK("acai_colsum");
        return 0;
    }
    const int rpb = 2048;
    dim3 grid(cdiv(cols, 64), cdiv(rows, rpb));
    if (dtype == ACAI_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t *)x, ld, out, rows, cols, rpb);
    else hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float *)x, ld, out, rows, cols, rpb);
    ACAI_LAUNCH_CHECK("acai_colsum");
    return 0;
}

extern "C" int acai_scatter_add_rows(const float *src, const int32_t *idx, float *dst, int rows, int dim, int shared_row, void *stream) {
    ACAI_CHECK_ARG(src && idx && dst && rows >= 0 && dim > 0, "acai_scatter_add_rows: bad arguments");
    if (rows == 0) return 0;
    const bool uniq = shared_row >= 0 && dim % 4 == 0 && aligned16(src) && aligned16(dst);
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(uniq ? cdiv(rows, 64) : cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, src, idx, dst, rows, dim,
                       uniq ? shared_row : -1);
    ACAI_LAUNCH_CHECK("acai_scatter_add_rows");
    return 0;
}

extern "C" int acai_mae_loss(const float *pred, const float *target, const unsigned char *mask, float inv_count, float *loss, float *dpred,
                             int rows, int dim, void *stream) {
    ACAI_CHECK_ARG(pred && target && mask && loss && rows >= 0 && dim > 1, "acai_mae_loss: bad arguments");
    if (rows == 0) return 0;
    const int rpw = rows >= 65536 ? 16 : (rows >= 4096 ? 4 : 1);
    hipLaunchKernelGGL(mae_loss_kernel, dim3(cdiv(rows, 4 * rpw)), dim3(256), 0, (hipStream_t)stream, pred, target, mask, inv_count, loss, dpred, rows, dim, rpw);
    ACAI_LAUNCH_CHECK("acai_mae_loss");
    return 0;
}

extern "C" int acai_ce_loss(const float *logits, int ld, const int64_t *target, int ignore_index, float inv_count, float label_smoothing, float *loss,
                            float *dlogits, int rows, int V, void *stream) {
    ACAI_CHECK_ARG(logits && target && loss && rows >= 0 && V > 0 && ld >= V && label_smoothing >= 0.f && label_smoothing <= 1.f, "acai_ce_loss: bad arguments");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(ce_loss_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, logits, ld, target, ignore_index, inv_count, loss, dlogits, rows, V,
                       label_smoothing);
    ACAI_LAUNCH_CHECK("acai_ce_loss");
    return 0;
}

extern "C" int acai_dropout_add(const void *x, const float *residual, void *out, int rows, int cols, float p, uint32_t seed, int x_dtype,
                                int out_dtype, void *stream) {
    ACAI_CHECK_ARG(x && out && rows >= 0 && cols > 0 && p >= 0.f && p < 1.f, "acai_dropout_add: bad arguments");
    if (rows == 0) return 0;
    const uint32_t thr = (uint32_t)((double)p * 4294967296.0);
    const float scale = 1.0f / (1.0f - p);
    const long n = (long)rows * cols;
    hipStream_t st = (hipStream_t)stream;
    if (x_dtype == ACAI_BF16 && out_dtype == ACAI_BF16)
        hipLaunchKernelGGL((dropout_add_kernel<bf16_t, bf16_t>), dim3(grid1d(n)), dim3(256), 0, st, (const bf16_t *)x, residual, (bf16_t *)out, rows, cols, thr, scale, seed);
    else if (x_dtype == ACAI_BF16)
        hipLaunchKernelGGL((dropout_add_kernel<bf16_t, float>), dim3(grid1d(n)), dim3(256), 0, st, (const bf16_t *)x, residual, (float *)out, rows, cols, thr, scale, seed);
    else if (out_dtype == ACAI_BF16)
        hipLaunchKernelGGL((dropout_add_kernel<float, bf16_t>), dim3(grid1d(n)), dim3(256), 0, st, (const float *)x, residual, (bf16_t *)out, rows, cols, thr, scale, seed);
    else
        hipLaunchKernelGGL((dropout_add_kernel<float, float>), dim3(grid1d(n)), dim3(256), 0, st, (const float *)x, residual, (float *)out, rows, cols, thr, scale, seed);
    ACAI_LAUNCH_CHECK("acai_dropout_add");
    return 0;
}

// ---- fused multi-tensor AdamW ---------------------------------------------------------------------------------------------------
// One launch updates every parameter of every group (the reference steps torch.optim.AdamW over 200-300 tensors: pre_train.py:105,
// omr_teacher_force_train.py:207 with the layer-wise LR groups of models.py:761-781).  Decoupled weight decay, bias correction and update
// in torch's order of operations:  p *= 1 - lr wd;  m += (g - m)(1 - b1);  v = b2 v + (1 - b2) g g;  p -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps).
// HBM-bound: 28 bytes per parameter.  A workgroup owns one chunk of one tensor (chunk table built by the host once per parameter set).
namespace {
__global__ __launch_bounds__(256) void adamw_kernel(const AcaiAdamWTensor *__restrict__ tensors, const AcaiAdamWGroup *__restrict__ groups,
                                                    const int32_t *__restrict__ chunk_tensor, const int64_t *__restrict__ chunk_off, int chunk_elems,
                                                    float grad_scale) {
    const AcaiAdamWTensor t = tensors[chunk_tensor[blockIdx.x]];
    const AcaiAdamWGroup h = groups[t.group];
    const int64_t off = chunk_off[blockIdx.x];
    const int64_t n = min((int64_t)chunk_elems, t.n - off);
    float *p = t.p + off, *m = t.m + off, *v = t.v + off;
    const float *g = t.g + off;
    const float decay = 1.0f - h.lr * h.weight_decay, step_size = h.lr / t.bias_c1, omb1 = 1.0f - h.beta1, omb2 = 1.0f - h.beta2;
    auto upd = [&](float &pp, float gg, float &mm, float &vv) {
        gg *= grad_scale;
        pp *= decay;
        mm += (gg - mm) * omb1;
        vv = vv * h.beta2 + omb2 * gg * gg;
        pp -= step_size * (mm / (sqrtf(vv) / t.bias_c2_sqrt + h.eps));
    };
    if (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0) {
        const int64_t n4 = n >> 2;
        for (int64_t i = threadIdx.x; i < n4; i += 256) {
            f32x4 pv = reinterpret_cast<f32x4 *>(p)[i], mv = reinterpret_cast<f32x4 *>(m)[i], vv = reinterpret_cast<f32x4 *>(v)[i];
            const f32x4 gv = reinterpret_cast<const f32x4 *>(g)[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = pv[e], b = mv[e], c = vv[e];
                upd(a, gv[e], b, c);
                pv[e] = a; mv[e] = b; vv[e] = c;
            }
            reinterpret_cast<f32x4 *>(p)[i] = pv;
            reinterpret_cast<f32x4 *>(m)[i] = mv;
            reinterpret_cast<f32x4 *>(v)[i] = vv;
        }
        for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) upd(p[i], g[i], m[i], v[i]);
    } else {
        for (int64_t i = threadIdx.x; i < n; i += 256) upd(p[i], g[i], m[i], v[i]);
    }
}
}  // namespace

extern "C" int acai_adamw_step(const AcaiAdamWTensor *tensors, const AcaiAdamWGroup *groups, const int32_t *chunk_tensor, const int64_t *chunk_off,
                               int n_chunks, int chunk_elems, float grad_scale, void *stream) {
    ACAI_CHECK_ARG(tensors && groups && chunk_tensor && chunk_off && n_chunks >= 0 && chunk_elems > 0 && chunk_elems % 4 == 0,
                   "acai_adamw_step: bad arguments");
    if (n_chunks == 0) return 0;
    hipLaunchKernelGGL(adamw_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, tensors, groups, chunk_tensor, chunk_off, chunk_elems, grad_scale);
    ACAI_LAUNCH_CHECK("acai_adamw_step");
    return 0;
}

// ---- operand copies of the fp32 master weights, all tensors in ONE launch -----------------------------------------------------------------
// After an optimizer step every cached bf16 operand copy of every parameter is stale (engine.WeightCache: autocast's weight cast done once
// per parameter version): the bf16 copy [rows][cols] (forward operand), the TRANSPOSED bf16 copy [cols][rows] (dX = dY . W as a row-major
// GEMM) and, for biases, the bf16-rounded fp32 vector.  Through ATen that was one cast / copy launch per tensor and kind (~300 launches of
// ~5-13 us per MAE step, ~2 ms).  Here: a table of entries, 64 x 64 tiles, the transposition through LDS.  HBM-bound, ~1 GB per step.
namespace {
__global__ __launch_bounds__(256) void cast_weights_kernel(const AcaiCastEntry *__restrict__ table, int n_entries) {
    __shared__ uint16_t tl[64][66];
    const int tile = blockIdx.x, tid = threadIdx.x;
    int lo = 0, hi = n_entries - 1;   // last entry with tile0 <= tile
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].tile0 <= tile) lo = mid; else hi = mid - 1;
    }
    const AcaiCastEntry e = table[lo];
    const int tiles_c = (e.cols + 63) >> 6, lt = tile - e.tile0;
    const int r0 = (lt / tiles_c) * 64, c0 = (lt % tiles_c) * 64;
    const int ty = tid >> 4, tx = tid & 15;
    const bool vec = (e.cols & 3) == 0;
    bf16_t *d16 = reinterpret_cast<bf16_t *>(e.dst16), *d16t = reinterpret_cast<bf16_t *>(e.dst16t);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 16 * i, c = c0 + tx * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < e.rows) {
            const float *sp = e.src + (size_t)r * e.cols + c;
            if (vec && c < e.cols) {
                const f32x4 q = *reinterpret_cast<const f32x4 *>(sp);
                v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c + j < e.cols) v[j] = sp[j];
            }
            bf16_t b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = f2bf(v[j]);
            if (d16) {
                bf16_t *dp = d16 + (size_t)r * e.cols + c;
                if (vec && c < e.cols) {
                    uint2 o;
                    o.x = (uint32_t)b[0] | ((uint32_t)b[1] << 16);
                    o.y = (uint32_t)b[2] | ((uint32_t)b[3] << 16);
                    *reinterpret_cast<uint2 *>(dp) = o;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (c + j < e.cols) dp[j] = b[j];
                }
            }
            if (e.dst32r) {
                float *dp = e.dst32r + (size_t)r * e.cols + c;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c + j < e.cols) dp[j] = bf2f(b[j]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) tl[ty + 16 * i][tx * 4 + j] = b[j];
        }
    }
    if (!d16t) return;   // (entry-uniform: the whole workgroup leaves)
    __syncthreads();
    const bool vect = (e.rows & 3) == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cl = ty + 16 * i, c = c0 + cl, r = r0 + tx * 4;   // output row c, output columns r .. r+3
        if (c >= e.cols || r >= e.rows) continue;
        bf16_t *dp = d16t + (size_t)c * e.rows + r;
        if (vect) {
            uint2 o;
            o.x = (uint32_t)tl[tx * 4 + 0][cl] | ((uint32_t)tl[tx * 4 + 1][cl] << 16);
            o.y = (uint32_t)tl[tx * 4 + 2][cl] | ((uint32_t)tl[tx * 4 + 3][cl] << 16);
            *reinterpret_cast<uint2 *>(dp) = o;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (r + j < e.rows) dp[j] = tl[tx * 4 + j][cl];
        }
    }
}
}  // namespace

extern "C" int acai_cast_weights(const AcaiCastEntry *table, int n_entries, int n_tiles, void *stream) {
    ACAI_CHECK_ARG(table && n_entries > 0 && n_tiles > 0, "acai_cast_weights: bad arguments");
    hipLaunchKernelGGL(cast_weights_kernel, dim3(n_tiles), dim3(256), 0, (hipStream_t)stream, table, n_entries);
    ACAI_LAUNCH_CHECK("acai_cast_weights");
    return 0;
}
