// Bicubic antialiased resize of one C x H x W fp32 image (the resize of DynamicResize / PatchDivisibleResize,
// acai_omr/utils/utils.py:325-330,351-356; torchvision's float path = aten `_upsample_bicubic2d_aa`, align_corners = False).
// Separable as aten does it: width first into `tmp` (C x H x OW), then height.  Every output sample owns its window:
//     scale = in / out, support = 2 scale (scale >= 1) or 2, centre = scale (i + 0.5),
//     taps xmin .. xmin + xsize - 1, weight_j = cubic_{a = -0.5}((j + xmin - centre + 0.5) / max(scale, 1)), normalised by their sum,
// with aten's mixed float / double evaluation of the window bounds, so that a tap never falls on the other side of a boundary.
// HBM-bound and tiny next to the encoder (one 512 x 2048 output reads ~4-40 MB): one thread per output sample, threads along x.
#include "common.h"

namespace {

struct Axis {
    float scale, support, invscale;
    int in_size, max_taps;
};

__device__ __forceinline__ float cubic_aa(float x) {
    const float a = -0.5f;
    x = fabsf(x);
    if (x < 1.f) return ((a + 2.f) * x - (a + 3.f)) * x * x + 1.f;
    if (x < 2.f) return (((x - 5.f) * x + 8.f) * x - 4.f) * a;
    return 0.f;
}

__device__ __forceinline__ void window(const Axis &ax, int i, int &xmin, int &xsize, float &center) {
    center = (float)((double)ax.scale * ((double)i + 0.5));
    const long long lo = (long long)((double)(center - ax.support) + 0.5);
    const long long hi = (long long)((double)(center + ax.support) + 0.5);
    xmin = (int)(lo > 0 ? lo : 0);
    int n = (int)(hi < ax.in_size ? hi : ax.in_size) - xmin;
    xsize = n < 0 ? 0 : (n > ax.max_taps ? ax.max_taps : n);
}

__device__ __forceinline__ float tap_weight(const Axis &ax, int j, int xmin, float center) {
    return cubic_aa((float)(((double)((float)(j + xmin) - center) + 0.5) * (double)ax.invscale));
}

// VERTICAL = false: in [rows = C*H][W] -> out [rows][OW] along x;  VERTICAL = true: in [C][H][OW] -> out [C][OH][OW] along y
template <bool VERTICAL>
__global__ __launch_bounds__(256) void resize_axis_kernel(const float *__restrict__ in, float *__restrict__ out, Axis ax, int out_size, int width,
                                                          int clamp01) {
    const int x = blockIdx.x * 256 + threadIdx.x;   // column of the OUTPUT row
    if (x >= width) return;
    const int i = VERTICAL ? (int)blockIdx.y : x;   // output index along the resized axis
    int xmin, xsize;
    float center;
    window(ax, i, xmin, xsize, center);
    float total = 0.f;
    for (int j = 0; j < xsize; ++j) total += tap_weight(ax, j, xmin, center);
    const float inv = total != 0.f ? 1.f / total : 0.f;
    const float *src;
    size_t stride;
    if (VERTICAL) {
        src = in + ((size_t)blockIdx.z * ax.in_size + xmin) * width + x;
        stride = width;
    } else {
        src = in + (size_t)blockIdx.y * ax.in_size + xmin;
        stride = 1;
    }
    float t = 0.f;
    for (int j = 0; j < xsize; ++j) {
        const float w = tap_weight(ax, j, xmin, center) * inv;
        t = j == 0 ? src[0] * w : t + src[(size_t)j * stride] * w;
    }
    if (clamp01) t = fminf(fmaxf(t, 0.f), 1.f);
    if (VERTICAL)
        out[((size_t)blockIdx.z * out_size + i) * width + x] = t;
    else
        out[(size_t)blockIdx.y * out_size + x] = t;
}

Axis make_axis(int in_size, int out_size) {
    Axis a;
    a.scale = (float)in_size / (float)out_size;
    a.support = a.scale >= 1.f ? 2.f * a.scale : 2.f;
    a.invscale = a.scale >= 1.f ? 1.f / a.scale : 1.f;
    a.in_size = in_size;
    a.max_taps = (int)ceilf(a.support) * 2 + 1;
    return a;
}

}  // namespace

extern "C" int acai_resize_bicubic_aa(const float *img, int C, int H, int W, float *tmp, float *out, int OH, int OW, int clamp01, void *stream) {
    ACAI_CHECK_ARG(img && tmp && out, "acai_resize_bicubic_aa: null operand");
    ACAI_CHECK_ARG(C > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "acai_resize_bicubic_aa: bad shape C=%d H=%d W=%d -> %d x %d", C, H, W, OH, OW);
    ACAI_CHECK_ARG((long long)C * H <= 65535 && OH <= 65535 && C <= 65535, "acai_resize_bicubic_aa: C*H, OH and C are grid dimensions (<= 65535)");
    hipStream_t st = (hipStream_t)stream;
    const Axis ax = make_axis(W, OW), ay = make_axis(H, OH);
    hipLaunchKernelGGL(resize_axis_kernel<false>, dim3((OW + 255) / 256, C * H, 1), dim3(256), 0, st, img, tmp, ax, OW, OW, 0);
    ACAI_LAUNCH_CHECK("acai_resize_bicubic_aa (width)");
    hipLaunchKernelGGL(resize_axis_kernel<true>, dim3((OW + 255) / 256, OH, C), dim3(256), 0, st, tmp, out, ay, OH, OW, clamp01);
    ACAI_LAUNCH_CHECK("acai_resize_bicubic_aa (height)");
    return 0;
}
