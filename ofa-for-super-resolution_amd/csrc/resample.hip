// resample.hip -- PIL's 8-bit bicubic resize on the GPU, bit for bit (ofasr_bicubic_resize_u8).
//
// The reference makes its LR training / validation images on the host with PIL:  Scale(1/2), Scale(1/4) =
// img.resize(size, Image.BICUBIC) on the uint8 HR crop (ofa/imagenet_codebase/data_providers/div2k_setxx.py:354-380,
// called per sample at :288-298).  At multi-thousand images/s per GPU that host loop is the next bottleneck (SURVEY.md
// 8f rank 4), so the same arithmetic runs here on uint8 planes already in HBM:
//   Pillow src/libImaging/Resample.c -- precompute_coeffs (double), normalize_coeffs_8bpc (22-bit fixed point, round
//   half away from zero), ImagingResampleHorizontal_8bpc then Vertical_8bpc (int32 accumulate, + 2^21, >> 22, clip).
// The coefficient tables are computed ON THE DEVICE in double with explicitly rounded operations (no FMA contraction:
// Pillow's x86-64 build has none), so they equal the host library's; integer passes are exact.  Pinned by
// oracle/pil_bicubic.py, which tests/test_resample.py pins to Pillow and to the reference's own LR images.
#include "ofasr_common.h"

namespace ofasr {

constexpr int RS_PREC = 32 - 8 - 2;
constexpr int RS_KMAX = 40;   // taps per output index: ceil(2 * scale) * 2 + 1; scale <= 8 supported

__device__ __forceinline__ double rs_bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) {   // ((a + 2.0) * x - (a + 3.0)) * x * x + 1
        const double t = __dsub_rn(__dmul_rn(a + 2.0, x), a + 3.0);
        return __dadd_rn(__dmul_rn(__dmul_rn(t, x), x), 1.0);
    }
    if (x < 2.0) {   // (((x - 5) * x + 8) * x - 4) * a
        const double t = __dadd_rn(__dmul_rn(__dsub_rn(x, 5.0), x), 8.0);
        return __dmul_rn(__dsub_rn(__dmul_rn(t, x), 4.0), a);
    }
    return 0.0;
}

// table[xx] = {xmin, count, k[0..ksize)}: int32, row stride 2 + ksize
__global__ void rs_coeff_kernel(int* __restrict__ table, int in_size, int out_size, int ksize) {
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    if (xx >= out_size) return;
    const double scale = __ddiv_rn((double)in_size, (double)out_size);
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = __dmul_rn(2.0, filterscale);
    const double ss = __ddiv_rn(1.0, filterscale);
    const double center = __dmul_rn(__dadd_rn((double)xx, 0.5), scale);
    int xmin = (int)__dadd_rn(__dsub_rn(center, support), 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)__dadd_rn(__dadd_rn(center, support), 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    int* row = table + (long long)xx * (2 + ksize);
    double w[RS_KMAX];
    double ww = 0.0;
    for (int x = 0; x < xmax && x < RS_KMAX; ++x) {
        w[x] = rs_bicubic(__dmul_rn(__dadd_rn(__dsub_rn((double)(x + xmin), center), 0.5), ss));
        ww = __dadd_rn(ww, w[x]);
    }
    for (int x = 0; x < ksize; ++x) {
        int k = 0;
        if (x < xmax && x < RS_KMAX) {
            const double v = ww != 0.0 ? __ddiv_rn(w[x], ww) : w[x];
            const double sc = __dmul_rn(v, (double)(1 << RS_PREC));
            k = v < 0 ? (int)__dadd_rn(-0.5, sc) : (int)__dadd_rn(0.5, sc);
        }
        row[2 + x] = k;
    }
    row[0] = xmin;
    row[1] = xmax;
}

__device__ __forceinline__ uint8_t rs_clip8(int v) {
    v >>= RS_PREC;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// dst[p][y][xx] = clip8(sum_x src[p][y][xmin + x] * k[x] + 2^21)
__global__ void rs_horizontal_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                     const int* __restrict__ table, int ksize, long long rows, int in_w, int out_w) {
    const long long total = rows * out_w;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long r = e / out_w;
        const int xx = (int)(e - r * out_w);
        const int* row = table + (long long)xx * (2 + ksize);
        const int xmin = row[0], cnt = row[1];
        const uint8_t* s = src + r * in_w + xmin;
        int acc = 1 << (RS_PREC - 1);
        for (int x = 0; x < cnt; ++x) acc += (int)s[x] * row[2 + x];
        dst[e] = rs_clip8(acc);
    }
}

// dst[p][yy][x] = clip8(sum_y src[p][ymin + y][x] * k[y] + 2^21)
__global__ void rs_vertical_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                   const int* __restrict__ table, int ksize, long long planes, int in_h, int out_h, int w) {
    const long long total = planes * out_h * w;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(e % w);
        const long long t = e / w;
        const int yy = (int)(t % out_h);
        const long long p = t / out_h;
        const int* row = table + (long long)yy * (2 + ksize);
        const int ymin = row[0], cnt = row[1];
        const uint8_t* s = src + (p * in_h + ymin) * (long long)w + x;
        int acc = 1 << (RS_PREC - 1);
        for (int y = 0; y < cnt; ++y) acc += (int)s[(long long)y * w] * row[2 + y];
        dst[e] = rs_clip8(acc);
    }
}

static int rs_ksize(int64_t in_size, int64_t out_size) {
    double scale = (double)in_size / (double)out_size;
    if (scale < 1.0) scale = 1.0;
    const double support = 2.0 * scale;
    int c = (int)support;
    if ((double)c < support) ++c;   // ceil
    return 2 * c + 1;
}
static size_t rs_align(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace ofasr

using namespace ofasr;

OFASR_EXPORT size_t ofasr_bicubic_resize_u8_workspace(int64_t planes, int64_t in_h, int64_t in_w, int64_t out_h,
                                                      int64_t out_w) {
    if (planes <= 0 || in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0) return 0;
    const size_t tw = rs_align((size_t)out_w * (2 + rs_ksize(in_w, out_w)) * sizeof(int));
    const size_t th = rs_align((size_t)out_h * (2 + rs_ksize(in_h, out_h)) * sizeof(int));
    return tw + th + rs_align((size_t)planes * in_h * out_w);
}

OFASR_EXPORT int ofasr_bicubic_resize_u8(const void* src, void* dst, int64_t planes, int64_t in_h, int64_t in_w,
                                         int64_t out_h, int64_t out_w, void* workspace, size_t workspace_bytes,
                                         void* stream) {
    const char* name = "ofasr_bicubic_resize_u8";
    OFASR_REQUIRE(src && dst, OFASR_ERR_INVALID_ARG, "%s: null pointer", name);
    OFASR_REQUIRE(planes > 0 && in_h > 0 && in_w > 0 && out_h > 0 && out_w > 0, OFASR_ERR_INVALID_ARG, "%s: bad shape", name);
    OFASR_REQUIRE(in_h <= (1 << 20) && in_w <= (1 << 20) && planes <= (1 << 24), OFASR_ERR_UNSUPPORTED, "%s: too large", name);
    const int kw = rs_ksize(in_w, out_w), kh = rs_ksize(in_h, out_h);
    OFASR_REQUIRE(kw <= RS_KMAX && kh <= RS_KMAX, OFASR_ERR_UNSUPPORTED, "%s: down-scale factor above 8 not supported", name);
    const size_t need = ofasr_bicubic_resize_u8_workspace(planes, in_h, in_w, out_h, out_w);
    OFASR_REQUIRE(workspace && workspace_bytes >= need, OFASR_ERR_WORKSPACE, "%s: workspace %zu B < required %zu B", name,
                  workspace_bytes, need);
    hipStream_t st = as_stream(stream);
    char* ws = (char*)workspace;
    int* tw = (int*)ws;
    const size_t tw_bytes = rs_align((size_t)out_w * (2 + kw) * sizeof(int));
    int* th = (int*)(ws + tw_bytes);
    const size_t th_bytes = rs_align((size_t)out_h * (2 + kh) * sizeof(int));
    uint8_t* mid = (uint8_t*)(ws + tw_bytes + th_bytes);
    OFASR_LAUNCH(rs_coeff_kernel, dim3((unsigned)cdiv(out_w, 64)), dim3(64), 0, st, tw, (int)in_w, (int)out_w, kw);
    OFASR_LAUNCH(rs_coeff_kernel, dim3((unsigned)cdiv(out_h, 64)), dim3(64), 0, st, th, (int)in_h, (int)out_h, kh);
    int rc = check_launch(name);
    if (rc) return rc;
    const long long rows = planes * in_h;
    {
        const long long total = rows * out_w;
        const long long blocks = cdiv(total, 256);
        prof_note((double)rows * (double)(in_w + out_w), 0.0);
        OFASR_LAUNCH(rs_horizontal_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, st,
                     (const uint8_t*)src, mid, (const int*)tw, kw, rows, (int)in_w, (int)out_w);
    }
    {
        const long long total = planes * out_h * out_w;
        const long long blocks = cdiv(total, 256);
        prof_note((double)planes * (double)out_w * (double)(in_h + out_h), 0.0);
        OFASR_LAUNCH(rs_vertical_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, st,
                     (const uint8_t*)mid, (uint8_t*)dst, (const int*)th, kh, planes, (int)in_h, (int)out_h, (int)out_w);
    }
    return check_launch(name);
}
