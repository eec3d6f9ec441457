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

// Where the height pass puts its samples when it feeds the encoder directly ("direct packing into the token stream"): the
// crop window (DynamicResize's centre crop to the positional grid) of the resized image goes out as nn.Unfold(P, P) rows - sample (y, x) of
// the window -> row row0 + (y / P) (cw / P) + x / P, column (y % P) P + x % P - in fp32 or bf16: no image tensor, no patchify launch, no cast.
struct PatchOut {
    void *base;
    int ld, row0, P, top, left, ch, cw, bf16;
};

__device__ __forceinline__ float in_sample(const float *p) { return *p; }
__device__ __forceinline__ float in_sample(const unsigned char *p) { return (float)*p / 255.0f; }   // v2.ToDtype(torch.float32, scale=True) of a uint8 image

// VERTICAL = false: in [rows = C*H][W] -> out [rows][OW] along x;  VERTICAL = true: in [C][H][OW] -> out [C][OH][OW] along y
template <bool VERTICAL, typename TI = float, bool PATCHES = false>
__global__ __launch_bounds__(256) void resize_axis_kernel(const TI *__restrict__ in, float *__restrict__ out, Axis ax, int out_size, int width,
                                                          int clamp01, PatchOut po = PatchOut{}) {
    const int x = blockIdx.x * 256 + threadIdx.x;   // column of the OUTPUT row
    if (x >= width) return;
    const int i = VERTICAL ? (int)blockIdx.y : x;   // output index along the resized axis
    int xmin, xsize;
    float center;
    window(ax, i, xmin, xsize, center);
    float total = 0.f;
    for (int j = 0; j < xsize; ++j) total += tap_weight(ax, j, xmin, center);
    const float inv = total != 0.f ? 1.f / total : 0.f;
    const TI *src;
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
        t = j == 0 ? in_sample(src) * w : t + in_sample(src + (size_t)j * stride) * w;
    }
    if (clamp01) t = fminf(fmaxf(t, 0.f), 1.f);
    if constexpr (PATCHES) {
        const int y = i - po.top, xx = x - po.left;
        if (y < 0 || y >= po.ch || xx < 0 || xx >= po.cw) return;
        const size_t row = (size_t)po.row0 + (size_t)(y / po.P) * (po.cw / po.P) + xx / po.P;
        const int col = (y % po.P) * po.P + xx % po.P;
        if (po.bf16) reinterpret_cast<bf16_t *>(po.base)[row * po.ld + col] = f2bf(t);
        else reinterpret_cast<float *>(po.base)[row * po.ld + col] = t;
    } else if (VERTICAL)
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
    hipLaunchKernelGGL((resize_axis_kernel<false, float, false>), dim3((OW + 255) / 256, C * H, 1), dim3(256), 0, st, img, tmp, ax, OW, OW, 0, PatchOut{});
    ACAI_LAUNCH_CHECK("acai_resize_bicubic_aa (width)");
    hipLaunchKernelGGL((resize_axis_kernel<true, float, false>), dim3((OW + 255) / 256, OH, C), dim3(256), 0, st, tmp, out, ay, OH, OW, clamp01, PatchOut{});
    ACAI_LAUNCH_CHECK("acai_resize_bicubic_aa (height)");
    return 0;
}

// One grayscale image (the path's NUM_CHANNELS = 1), fp32 in [0, 1] or uint8 (scaled by 1/255 on load: the reference's ToImage -> ToDtype(float32,
// scale=True), acai_omr/train/pre_train.py:56, omr_teacher_force_train.py:278-282), resized to OH x OW (bicubic, antialias, clamp) and written
// as the Unfold(P, P) rows row0 .. of the packed patch stream `patches` [rows][ld >= P*P] (fp32 or bf16), cropped to the window
// (top, left, ch, cw) of the resized image (DynamicResize's centre crop; the whole image: 0, 0, OH, OW).  ch and cw are multiples of P.
extern "C" int acai_resize_to_patches(const void *img, int in_u8, int H, int W, float *tmp, void *patches, int ld, int row0, int OH, int OW, int top,
                                      int left, int ch, int cw, int P, int out_dtype, int clamp01, void *stream) {
    ACAI_CHECK_ARG(img && tmp && patches, "acai_resize_to_patches: null operand");
    ACAI_CHECK_ARG(H > 0 && W > 0 && OH > 0 && OW > 0 && P > 0 && ld >= P * P && row0 >= 0, "acai_resize_to_patches: bad shape %d x %d -> %d x %d, P = %d, ld = %d", H, W, OH, OW, P, ld);
    ACAI_CHECK_ARG(top >= 0 && left >= 0 && ch > 0 && cw > 0 && top + ch <= OH && left + cw <= OW && ch % P == 0 && cw % P == 0,
                   "acai_resize_to_patches: crop window (%d, %d, %d, %d) outside %d x %d or not a multiple of the patch size", top, left, ch, cw, OH, OW);
    ACAI_CHECK_ARG(H <= 65535 && OH <= 65535, "acai_resize_to_patches: H and OH are grid dimensions (<= 65535)");
    ACAI_CHECK_ARG(out_dtype == ACAI_F32 || out_dtype == ACAI_BF16, "acai_resize_to_patches: bad output dtype");
    hipStream_t st = (hipStream_t)stream;
    const Axis ax = make_axis(W, OW), ay = make_axis(H, OH);
    if (in_u8)
        hipLaunchKernelGGL((resize_axis_kernel<false, unsigned char, false>), dim3((OW + 255) / 256, H, 1), dim3(256), 0, st,
                           reinterpret_cast<const unsigned char *>(img), tmp, ax, OW, OW, 0, PatchOut{});
    else
        hipLaunchKernelGGL((resize_axis_kernel<false, float, false>), dim3((OW + 255) / 256, H, 1), dim3(256), 0, st, reinterpret_cast<const float *>(img), tmp, ax, OW, OW,
                           0, PatchOut{});
    ACAI_LAUNCH_CHECK("acai_resize_to_patches (width)");
    const PatchOut po{patches, ld, row0, P, top, left, ch, cw, out_dtype == ACAI_BF16 ? 1 : 0};
    hipLaunchKernelGGL((resize_axis_kernel<true, float, true>), dim3((OW + 255) / 256, OH, 1), dim3(256), 0, st, tmp, nullptr, ay, OH, OW, clamp01, po);
    ACAI_LAUNCH_CHECK("acai_resize_to_patches (height)");
    return 0;
}
