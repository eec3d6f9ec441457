// HBM-bound row kernels: LayerNorm, patchify (nn.Unfold), row gather; plus ABI plumbing (errors, version,
// hipGraph helpers).  One wave per row, 16-byte vector accesses where the row allows it.
#include <stdarg.h>

#include "common.h"

char g_acai_err[512] = "";

int acai_set_err(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_acai_err, sizeof(g_acai_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" int acai_version(void) { return ACAI_ABI_VERSION; }
extern "C" const char *acai_last_error(void) { return g_acai_err; }

namespace {

// ---- LayerNorm: torch nn.LayerNorm semantics (biased variance, eps inside the sqrt) ---------------------
// Two-pass in fp32 (mean, then centred sum of squares) so that fp32 parity with torch holds to ~1e-7.
template <bool VEC>
__global__ __launch_bounds__(256) void layernorm_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                        const float *__restrict__ b, float eps, float *out_f32,
                                                        bf16_t *out_bf16, int rows, int dim) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + (size_t)row * dim;
    float s = 0.f;
    if constexpr (VEC) {
        for (int i = lane * 4; i < dim; i += 256) {
            const float4 v = *reinterpret_cast<const float4 *>(xr + i);
            s += (v.x + v.y) + (v.z + v.w);
        }
    } else {
        for (int i = lane; i < dim; i += 64) s += xr[i];
    }
    const float mean = wave_sum(s) / (float)dim;
    float q = 0.f;
    if constexpr (VEC) {
        for (int i = lane * 4; i < dim; i += 256) {
            const float4 v = *reinterpret_cast<const float4 *>(xr + i);
            const float a = v.x - mean, c = v.y - mean, d = v.z - mean, e = v.w - mean;
            q += (a * a + c * c) + (d * d + e * e);
        }
    } else {
        for (int i = lane; i < dim; i += 64) {
            const float a = xr[i] - mean;
            q += a * a;
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)dim + eps);
    if constexpr (VEC) {
        for (int i = lane * 4; i < dim; i += 256) {
            const float4 v = *reinterpret_cast<const float4 *>(xr + i);
            const float4 wv = *reinterpret_cast<const float4 *>(w + i);
            const float4 bv = *reinterpret_cast<const float4 *>(b + i);
            float4 y;
            y.x = (v.x - mean) * rstd * wv.x + bv.x;
            y.y = (v.y - mean) * rstd * wv.y + bv.y;
            y.z = (v.z - mean) * rstd * wv.z + bv.z;
            y.w = (v.w - mean) * rstd * wv.w + bv.w;
            if (out_f32) *reinterpret_cast<float4 *>(out_f32 + (size_t)row * dim + i) = y;
            if (out_bf16) {
                uint2 p;
                p.x = pack_bf16(y.x, y.y);
                p.y = pack_bf16(y.z, y.w);
                *reinterpret_cast<uint2 *>(out_bf16 + (size_t)row * dim + i) = p;
            }
        }
    } else {
        for (int i = lane; i < dim; i += 64) {
            const float y = (xr[i] - mean) * rstd * w[i] + b[i];
            if (out_f32) out_f32[(size_t)row * dim + i] = y;
            if (out_bf16) out_bf16[(size_t)row * dim + i] = f2bf(y);
        }
    }
}

// ---- patchify: (1,H,W) image -> rows of P*P pixels, patch (i,j) -> row i*wp + j, pixels (kh,kw) row-major.
// One thread per output element group of 4 along kw (P % 4 == 0) or per element.
template <typename TO>
__global__ __launch_bounds__(256) void patchify_kernel(const float *__restrict__ img, int H, int W, int P, TO *out, int ld,
                                                       int row0, int hp, int wp) {
    const int PP = P * P;
    const long total = (long)hp * wp * PP;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        // consecutive threads walk kw fastest, then patch column j, then kh, then patch row i: image reads are
        // contiguous along a pixel row (wp*P floats), writes are P-element runs.
        const int kw = idx % P;
        long t = idx / P;
        const int j = t % wp;
        t /= wp;
        const int kh = t % P;
        const int i = t / P;
        const float v = img[(size_t)(i * P + kh) * W + j * P + kw];
        DT<TO>::st(out + (size_t)(row0 + i * wp + j) * ld + kh * P + kw, v);
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ table, const int32_t *__restrict__ idx,
                                                          const float *__restrict__ add, float *out, int rows, int dim) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *src = table + (size_t)idx[row] * dim;
    const float *ad = add ? add + (size_t)row * dim : nullptr;
    float *dst = out + (size_t)row * dim;
    if constexpr (VEC) {
        for (int i = lane * 4; i < dim; i += 256) {
            float4 v = *reinterpret_cast<const float4 *>(src + i);
            if (ad) {
                const float4 a = *reinterpret_cast<const float4 *>(ad + i);
                v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
            }
            *reinterpret_cast<float4 *>(dst + i) = v;
        }
    } else {
        for (int i = lane; i < dim; i += 64) dst[i] = src[i] + (ad ? ad[i] : 0.f);
    }
}

// OMREncoder.interpolate_pe (M:291-302) = aten upsample_bilinear2d, align_corners = False, on the (Hin, Win, E) table; one wave per output
// grid cell, lanes over E.  BWD = the transposed stencil (four float atomics per element: the table is tiny and L2-resident).
struct Bilin { int y0, y1, x0, x1; float ly, lx; };
__device__ __forceinline__ Bilin bilin_of(int oy, int ox, int Hin, int Win, float sh, float sw) {
    Bilin b;
    const float fy = fmaxf(sh * ((float)oy + 0.5f) - 0.5f, 0.f), fx = fmaxf(sw * ((float)ox + 0.5f) - 0.5f, 0.f);
    b.y0 = min((int)fy, Hin - 1);
    b.x0 = min((int)fx, Win - 1);
    b.y1 = b.y0 + (b.y0 < Hin - 1 ? 1 : 0);
    b.x1 = b.x0 + (b.x0 < Win - 1 ? 1 : 0);
    b.ly = fy - (float)b.y0;
    b.lx = fx - (float)b.x0;
    return b;
}

template <bool BWD>
__global__ __launch_bounds__(256) void pe_interp_kernel(const float *__restrict__ src, float *dst, int Hin, int Win, int E, int Hout, int Wout,
                                                        float sh, float sw) {
    const int lane = threadIdx.x & 63, cell = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cell >= Hout * Wout) return;
    const int oy = cell / Wout, ox = cell - oy * Wout;
    const Bilin b = bilin_of(oy, ox, Hin, Win, sh, sw);
    const float hy = 1.f - b.ly, hx = 1.f - b.lx;
    const size_t r00 = ((size_t)b.y0 * Win + b.x0) * E, r01 = ((size_t)b.y0 * Win + b.x1) * E, r10 = ((size_t)b.y1 * Win + b.x0) * E,
                 r11 = ((size_t)b.y1 * Win + b.x1) * E, ro = (size_t)cell * E;
    for (int e = lane; e < E; e += 64) {
        if constexpr (!BWD) {
            // aten: h0lambda * (w0lambda * v00 + w1lambda * v01) + h1lambda * (w0lambda * v10 + w1lambda * v11)
            dst[ro + e] = hy * (hx * src[r00 + e] + b.lx * src[r01 + e]) + b.ly * (hx * src[r10 + e] + b.lx * src[r11 + e]);
        } else {
            const float g = src[ro + e];
            atomicAdd(dst + r00 + e, hy * hx * g);
            atomicAdd(dst + r01 + e, hy * b.lx * g);
            atomicAdd(dst + r10 + e, b.ly * hx * g);
            atomicAdd(dst + r11 + e, b.ly * b.lx * g);
        }
    }
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float *__restrict__ x, bf16_t *y, long n) {
    long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    const long stride = (long)gridDim.x * 1024;
    for (; i + 3 < n; i += stride) {
        const float4 v = *reinterpret_cast<const float4 *>(x + i);
        uint2 p;
        p.x = pack_bf16(v.x, v.y);
        p.y = pack_bf16(v.z, v.w);
        *reinterpret_cast<uint2 *>(y + i) = p;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) y[(n & ~3L) + threadIdx.x] = f2bf(x[(n & ~3L) + threadIdx.x]);
}

}  // namespace

extern "C" int acai_cast_f32_bf16(const float *x, void *y, int64_t n, void *stream) {
    ACAI_CHECK_ARG(x && y && n >= 0 && aligned16(x) && aligned16(y), "acai_cast_f32_bf16: bad arguments");
    if (n == 0) return 0;
    const long groups = (n / 4 + 255) / 256;
    const int grid = (int)(groups < 1 ? 1 : (groups > 4096 ? 4096 : groups));
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (bf16_t *)y, (long)n);
    ACAI_LAUNCH_CHECK("acai_cast_f32_bf16");
    return 0;
}

extern "C" int acai_layernorm_fwd(const float *x, const float *w, const float *b, float eps, float *out_f32, void *out_bf16,
                                  int rows, int dim, void *stream) {
    ACAI_CHECK_ARG(x && w && b && (out_f32 || out_bf16), "acai_layernorm_fwd: null operand");
    ACAI_CHECK_ARG(rows >= 0 && dim > 0, "acai_layernorm_fwd: bad shape rows=%d dim=%d", rows, dim);
    if (rows == 0) return 0;
    const bool vec = (dim % 4 == 0) && aligned16(x) && aligned16(w) && aligned16(b) && (!out_f32 || aligned16(out_f32)) &&
                     (!out_bf16 || aligned16(out_bf16));
    hipStream_t st = (hipStream_t)stream;
    if (vec)
        hipLaunchKernelGGL(layernorm_kernel<true>, dim3(cdiv(rows, 4)), dim3(256), 0, st, x, w, b, eps, out_f32, (bf16_t *)out_bf16, rows, dim);
    else
        hipLaunchKernelGGL(layernorm_kernel<false>, dim3(cdiv(rows, 4)), dim3(256), 0, st, x, w, b, eps, out_f32, (bf16_t *)out_bf16, rows, dim);
    ACAI_LAUNCH_CHECK("acai_layernorm_fwd");
    return 0;
}

extern "C" int acai_patchify(const float *img, int H, int W, int P, void *out, int ld, int row0, int out_dtype, void *stream) {
    ACAI_CHECK_ARG(img && out && P > 0 && H >= P && W >= P && ld >= P * P && row0 >= 0, "acai_patchify: bad arguments");
    const int hp = H / P, wp = W / P;
    const long total = (long)hp * wp * P * P;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipStream_t st = (hipStream_t)stream;
    if (out_dtype == ACAI_BF16)
        hipLaunchKernelGGL(patchify_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, img, H, W, P, (bf16_t *)out, ld, row0, hp, wp);
    else if (out_dtype == ACAI_F32)
        hipLaunchKernelGGL(patchify_kernel<float>, dim3(grid), dim3(256), 0, st, img, H, W, P, (float *)out, ld, row0, hp, wp);
    else
        return acai_set_err(-1, "acai_patchify: bad dtype %d", out_dtype);
    ACAI_LAUNCH_CHECK("acai_patchify");
    return 0;
}

extern "C" int acai_gather_rows(const float *table, const int32_t *idx, const float *add, float *out, int rows, int dim, void *stream) {
    ACAI_CHECK_ARG(table && idx && out && rows >= 0 && dim > 0, "acai_gather_rows: bad arguments");
    if (rows == 0) return 0;
    const bool vec = (dim % 4 == 0) && aligned16(table) && aligned16(out) && (!add || aligned16(add));
    hipStream_t st = (hipStream_t)stream;
    if (vec)
        hipLaunchKernelGGL(gather_rows_kernel<true>, dim3(cdiv(rows, 4)), dim3(256), 0, st, table, idx, add, out, rows, dim);
    else
        hipLaunchKernelGGL(gather_rows_kernel<false>, dim3(cdiv(rows, 4)), dim3(256), 0, st, table, idx, add, out, rows, dim);
    ACAI_LAUNCH_CHECK("acai_gather_rows");
    return 0;
}

extern "C" int acai_pe_interp_fwd(const float *table, int Hin, int Win, int E, float *out, int Hout, int Wout, void *stream) {
    ACAI_CHECK_ARG(table && out && Hin > 0 && Win > 0 && E > 0 && Hout > 0 && Wout > 0, "acai_pe_interp_fwd: bad arguments");
    hipLaunchKernelGGL(pe_interp_kernel<false>, dim3(cdiv(Hout * Wout, 4)), dim3(256), 0, (hipStream_t)stream, table, out, Hin, Win, E, Hout, Wout,
                       (float)Hin / (float)Hout, (float)Win / (float)Wout);
    ACAI_LAUNCH_CHECK("acai_pe_interp_fwd");
    return 0;
}

extern "C" int acai_pe_interp_bwd(const float *dout, int Hout, int Wout, int E, float *dtable, int Hin, int Win, void *stream) {
    ACAI_CHECK_ARG(dout && dtable && Hin > 0 && Win > 0 && E > 0 && Hout > 0 && Wout > 0, "acai_pe_interp_bwd: bad arguments");
    hipLaunchKernelGGL(pe_interp_kernel<true>, dim3(cdiv(Hout * Wout, 4)), dim3(256), 0, (hipStream_t)stream, dout, dtable, Hin, Win, E, Hout, Wout,
                       (float)Hin / (float)Hout, (float)Win / (float)Wout);
    ACAI_LAUNCH_CHECK("acai_pe_interp_bwd");
    return 0;
}

// ---- hipGraph helpers ------------------------------------------------------------------------------------
extern "C" int acai_graph_begin(void *stream) {
    hipError_t e = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) return acai_set_err((int)e, "hipStreamBeginCapture: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int acai_graph_end(void *stream, void **graph_exec_out) {
    ACAI_CHECK_ARG(graph_exec_out, "acai_graph_end: null out");
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamEndCapture((hipStream_t)stream, &graph);
    if (e != hipSuccess) return acai_set_err((int)e, "hipStreamEndCapture: %s", hipGetErrorString(e));
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (e != hipSuccess) return acai_set_err((int)e, "hipGraphInstantiate: %s", hipGetErrorString(e));
    *graph_exec_out = (void *)exec;
    return 0;
}

extern "C" int acai_graph_launch(void *graph_exec, void *stream) {
    hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
    if (e != hipSuccess) return acai_set_err((int)e, "hipGraphLaunch: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int acai_graph_destroy(void *graph_exec) {
    if (!graph_exec) return 0;
    hipError_t e = hipGraphExecDestroy((hipGraphExec_t)graph_exec);
    if (e != hipSuccess) return acai_set_err((int)e, "hipGraphExecDestroy: %s", hipGetErrorString(e));
    return 0;
}


// ---- hardware-assumption probe: out-of-range lanes of a buffer LDS-DMA (see include/acai_omr_hip.h) -------------------------------------------
namespace {
__global__ __launch_bounds__(64) void lds_dma_oob_probe_kernel(const uint32_t *src, int valid_bytes, uint32_t *out) {
    __shared__ __attribute__((aligned(16))) uint32_t img[64 * 4];
    for (int i = threadIdx.x; i < 256; i += 64) img[i] = 0xFFFFFFFFu;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(src), 0, valid_bytes, 0x00020000);
    const uint32_t voff = threadIdx.x * 16;
    typedef __attribute__((address_space(3))) void *lds_ptr;
    const uint32_t dst = (uint32_t)(uintptr_t)(lds_ptr)img;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_waitcnt vmcnt(0)" ::"s"(dst), "v"(voff), "s"(rs) : "memory", "m0");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = img[i];
}
}  // namespace

extern "C" int acai_debug_lds_dma_oob(const void *src, int valid_bytes, void *out, void *stream) {
    ACAI_CHECK_ARG(src && out && valid_bytes >= 0 && valid_bytes <= 1024, "acai_debug_lds_dma_oob: bad arguments");
    hipLaunchKernelGGL(lds_dma_oob_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, reinterpret_cast<const uint32_t *>(src), valid_bytes,
                       reinterpret_cast<uint32_t *>(out));
    ACAI_LAUNCH_CHECK("acai_debug_lds_dma_oob");
    return 0;
}
