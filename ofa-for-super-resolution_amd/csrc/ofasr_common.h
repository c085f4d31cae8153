// ofasr_common.h -- shared device/host helpers for the gfx950 kernels.  HIP only, CDNA4 only.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <tuple>
#include <utility>
#include <stdint.h>
#include <stdio.h>
#include "../../include/ofasr.h"

#define OFASR_EXPORT extern "C" __attribute__((visibility("default")))

namespace ofasr {

void set_error(const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return OFASR_ERR_LAUNCH;
    }
    return OFASR_OK;
}

// ---- launch bookkeeping ------------------------------------------------------------------------
// Every kernel launch of the library goes through OFASR_LAUNCH.  Each launch site template instance owns a
// LaunchSite (its name is the fully resolved, demangled kernel symbol the runtime reports for the host stub -- the
// name rocprofv3 prints, minus the parameter list) with a launch counter --
// tests assert the routing of a call through it (ofasr_debug_launch_count) -- and, while profiling is switched on
// (ofasr_profile_enable), the launch carries its own start / stop events (hipExtLaunchKernelGGL: the two events are
// attached to the kernel's dispatch packet, so their difference is the kernel's begin -> end time as the GPU's
// completion signal records it -- the duration rocprofv3's kernel trace prints -- with no extra barrier packet on the
// stream: launch gaps are not counted and kernels of the two streams overlap exactly as in the un-profiled step).
// bench.py's per-kernel table and `roofline` therefore describe the kernels the timed step really ran, in the regime
// it ran them, whichever stream (caller's or the library's side stream) they ran on.  prof_note() attaches
// algorithmic bytes / flops to the next launch of this host thread.  With profiling off the cost per launch is one
// relaxed increment.
struct LaunchSite {
    const char* name;
    unsigned long long count;
    LaunchSite* next;
};
LaunchSite* register_site(const void* host_fn, const char* fallback);
template <auto F> inline LaunchSite* launch_site(const char* spelled) {
    static LaunchSite* s = register_site(reinterpret_cast<const void*>(F), spelled);
    return s;
}
extern int g_profile_on;
void prof_note(double bytes, double flops);
struct ProfEvents {
    hipEvent_t t0, t1;
};
ProfEvents* prof_begin(LaunchSite* s);
// kernel<<<grid, block, shmem, stream>>>(args...) with the two events attached to the dispatch itself; the actual
// arguments are converted to the kernel's formal parameter types first, as the triple-chevron launch does
template <typename... F, size_t... I>
inline void ext_launch_tuple(void (*kernel)(F...), dim3 grid, dim3 block, unsigned shmem, hipStream_t st, ProfEvents* ev,
                             std::tuple<F...>& formals, std::index_sequence<I...>) {
    void* ptrs[sizeof...(F) ? sizeof...(F) : 1] = {(void*)&std::get<I>(formals)...};
    (void)hipExtLaunchKernel(reinterpret_cast<const void*>(kernel), grid, block, ptrs, shmem, st, ev->t0, ev->t1, 0);
}
template <size_t I, typename Fi, typename Tup> inline Fi ext_arg(Tup& actuals) {
    // trailing parameters a call leaves to their defaults: every default in csrc/ is the value-initialised object
    // (nullptr, InputXf{}, BwdXf{}, BnFold{}, StatOut{nullptr, 0})
    if constexpr (I < std::tuple_size<Tup>::value) return static_cast<Fi>(std::get<I>(actuals));
    else return Fi{};
}
template <typename... F, size_t... I, typename Tup>
inline void ext_launch_conv(void (*kernel)(F...), dim3 grid, dim3 block, unsigned shmem, hipStream_t st, ProfEvents* ev,
                            Tup& actuals, std::index_sequence<I...> seq) {
    std::tuple<F...> formals{ext_arg<I, F>(actuals)...};
    ext_launch_tuple(kernel, grid, block, shmem, st, ev, formals, seq);
}
template <typename... F, typename... A>
inline void ext_launch(void (*kernel)(F...), dim3 grid, dim3 block, unsigned shmem, hipStream_t st, ProfEvents* ev,
                       A&&... a) {
    static_assert(sizeof...(F) >= sizeof...(A), "kernel argument count");
    auto actuals = std::forward_as_tuple(a...);
    ext_launch_conv(kernel, grid, block, shmem, st, ev, actuals, std::index_sequence_for<F...>{});
}

#define OFASR_LAUNCH(kernel, grid, block, shmem, stream, ...)                                              \
    do {                                                                                                   \
        ::ofasr::LaunchSite* _site = ::ofasr::launch_site<&kernel>(#kernel);                               \
        __atomic_fetch_add(&_site->count, 1ull, __ATOMIC_RELAXED);                                         \
        ::ofasr::ProfEvents* _ev = ::ofasr::g_profile_on ? ::ofasr::prof_begin(_site) : nullptr;           \
        if (_ev)                                                                                           \
            ::ofasr::ext_launch(kernel, grid, block, shmem, stream, _ev, __VA_ARGS__);                     \
        else                                                                                               \
            hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                           \
    } while (0)

#define OFASR_REQUIRE(cond, code, ...)      \
    do {                                    \
        if (!(cond)) {                      \
            ofasr::set_error(__VA_ARGS__);  \
            return (code);                  \
        }                                   \
    } while (0)

// ---- 16-bit element types carried as raw bits -------------------------------------------------
struct bf16_t { uint16_t v; };
struct f16_t { uint16_t v; };

__device__ __forceinline__ float to_float(float x) { return x; }
__device__ __forceinline__ float to_float(bf16_t x) { return __uint_as_float(((uint32_t)x.v) << 16); }
__device__ __forceinline__ float to_float(f16_t x) {
    _Float16 h;
    __builtin_memcpy(&h, &x.v, 2);
    return (float)h;
}

template <typename T> __device__ __forceinline__ T from_float(float x);
template <> __device__ __forceinline__ float from_float<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_float<bf16_t>(float x) {
    // plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN stays NaN) on gfx950
    __bf16 b = (__bf16)x;
    bf16_t r;
    __builtin_memcpy(&r.v, &b, 2);
    return r;
}
template <> __device__ __forceinline__ f16_t from_float<f16_t>(float x) {
    _Float16 h = (_Float16)x;
    f16_t r;
    __builtin_memcpy(&r.v, &h, 2);
    return r;
}

// the pair as ONE vector conversion: v_cvt_pk_bf16_f32 dst, lo, hi (RNE, as the scalar cast).  Converting the halves
// separately and OR-ing them costs three instructions per pair (cvt, cvt-or via SDWA, shift).
typedef float ofasr_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 ofasr_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 ofasr_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
    const ofasr_f32x2 f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, ofasr_bf16x2));
}
__device__ __forceinline__ uint32_t pack2_f16(float lo, float hi) {
    const ofasr_f32x2 f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, ofasr_f16x2));
}
// the two 16-bit values of a packed word -> fp32
template <typename T> __device__ __forceinline__ void unpack2(uint32_t w, float& lo, float& hi);
template <> __device__ __forceinline__ void unpack2<bf16_t>(uint32_t w, float& lo, float& hi) {
    lo = __uint_as_float(w << 16);
    hi = __uint_as_float(w & 0xffff0000u);
}
template <> __device__ __forceinline__ void unpack2<f16_t>(uint32_t w, float& lo, float& hi) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 v = __builtin_bit_cast(h2, w);
    lo = (float)v.x;
    hi = (float)v.y;
}
template <typename T> __device__ __forceinline__ uint32_t pack2(float lo, float hi);
template <> __device__ __forceinline__ uint32_t pack2<bf16_t>(float lo, float hi) { return pack2_bf16(lo, hi); }
template <> __device__ __forceinline__ uint32_t pack2<f16_t>(float lo, float hi) { return pack2_f16(lo, hi); }

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// wave-wide sum (64 lanes); result valid in every lane
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int ES> struct uint_of;
template <> struct uint_of<1> { using type = uint8_t; };
template <> struct uint_of<2> { using type = uint16_t; };
template <> struct uint_of<4> { using type = uint32_t; };
template <> struct uint_of<8> { using type = uint64_t; };

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Optional BatchNorm + ReLU6 applied to an operand as a kernel reads it (the composite MB block never materialises
// the activated tensor): v -> min(max((v - mean[c]) * scale[c] + beta[c], 0), 6) with beta = shift + mean*scale.
// Zero padding / out-of-range elements stay exactly zero.  All pointers null = plain read.
struct InputXf {
    const float* scale;
    const float* shift;
    const float* mean;
};

// BatchNorm(+ReLU6) BACKWARD applied to an operand as a kernel reads it: the gradient dy of the pre-BN tensor y is
//     t = y - mean;  pre = t*scale + beta;  dz = (0 < pre < 6) ? da : 0;  dy = scale*dz - ka - t*kbi
// (ka = scale * mean(dz), kbi = scale * invstd * mean(dz * xhat); both 0 for eval-mode BN) formed from the incoming
// gradient da and y on the fly, so the "apply" pass of the BN backward -- read da, read y, write dy -- never runs and dy
// is a pass of its own no more.  The per-channel sums behind ka / kbi come from bn_bwd_reduce_coef.  The kernel that
// consumes dy on the input-gradient chain also stores it once (dy_out, may be null) for the weight-gradient kernel on the
// side stream: compared with the separate pass the chain moves one tensor less (no second read of da) and loses a
// launch, and the side stream reads what it read before.  y == nullptr: plain read.
struct BwdXf {
    const void* y;        // the pre-BN tensor, same shape and element type as the gradient operand
    const float* mean;
    const float* scale;   // gamma * invstd
    const float* shift;   // beta - mean * scale
    const float* ka;
    const float* kbi;
    void* dy_out;         // where the consumer leaves dy (same shape / element type), or nullptr
    // Optional: the consumer folds the reduction pass's partial slabs itself (no coefficient launch between the two).
    // fold_partial[(q * C + c) * 2 + {0, 1}] = (sum dz, sum dz * xhat) of part q < fold_P (bn_bwd_reduce_kernel's layout),
    // folded in fp64 in the order q = 0 .. fold_P-1 exactly as bn_bwd_coef_kernel does; ka / kbi above are then ignored.
    // ONE designated unit per channel also writes dgamma / dbeta and (coef_ka, coef_kbi) -- the latter for a later kernel
    // that takes finished coefficients (the side stream's weight gradient).
    const double* fold_partial;
    int fold_P, fold_C, fold_training;
    double fold_M;
    const float* fold_invstd;
    float* fold_dgamma;
    float* fold_dbeta;
    float* coef_ka;
    float* coef_kbi;
};
#if defined(__HIPCC__)
// (ka, kbi) of channel c from the partial slabs; `writer`: this unit also publishes dgamma / dbeta / the coefficients
__device__ __forceinline__ void bwdxf_fold(const BwdXf& bx, int c, float sc, bool writer, float& ka, float& kbi) {
    double s = 0.0, sx = 0.0;
    for (int q0 = 0; q0 < bx.fold_P; q0 += 8) {      // the loads of a round requested together, summed in order
        double a[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = q0 + j < bx.fold_P ? q0 + j : bx.fold_P - 1;
            a[j] = bx.fold_partial[((long long)q * bx.fold_C + c) * 2];
            b[j] = bx.fold_partial[((long long)q * bx.fold_C + c) * 2 + 1];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s += q0 + j < bx.fold_P ? a[j] : 0.0;
            sx += q0 + j < bx.fold_P ? b[j] : 0.0;
        }
    }
    ka = bx.fold_training ? (float)((double)sc * (s / bx.fold_M)) : 0.f;
    kbi = bx.fold_training ? (float)((double)sc * (double)bx.fold_invstd[c] * (sx / bx.fold_M)) : 0.f;
    if (writer) {
        if (bx.fold_dgamma) bx.fold_dgamma[c] = (float)sx;
        if (bx.fold_dbeta) bx.fold_dbeta[c] = (float)s;
        if (bx.coef_ka) bx.coef_ka[c] = ka;
        if (bx.coef_kbi) bx.coef_kbi[c] = kbi;
    }
}
#endif
// The REDUCTION pass of the same backward folded into the kernel that produces da (the gradient of the activated
// tensor): the producer reads y at the positions it writes and leaves, per channel c and producer unit p (a pixel tile,
// a plane slab), partial[c * P + p] = (sum dz, sum dz * (y - mean)) with dz as above and da as stored; bn_bwd_coef_cp
// folds the P partials of a channel in fp64 in a fixed order into dgamma, dbeta, ka, kbi.  The chain loses the pass
// "read da, read y" and its launch; the producer reads y (one tensor) more.
struct BwdStatOut {
    const void* y;
    const float* mean;
    const float* scale;
    const float* shift;
    float2* partial;
    int P;
};
int bn_bwd_coef_cp(const float2* partial, int64_t P, int64_t C, double count, int training, const float* scale,
                   const float* invstd, float* dgamma, float* dbeta, float* ka, float* kbi, void* stream);
bool pwconv_dgrad_bstat_supported(const void* dy, const void* y, const void* dx, const float* w, int64_t ldw, int64_t Cin,
                                  int64_t Cout, int64_t HW, int dtype);
int pwconv_dgrad_bstat(const void* dy, const float* w, int64_t ldw, void* dx, int64_t N, int64_t Cin, int64_t Cout,
                       int64_t HW, int dtype, BwdStatOut bs, void* stream);
// reduction pass of the BN(+ReLU6) backward + a C-thread kernel that turns the partials into dgamma, dbeta and the
// (ka, kbi) of BwdXf.  workspace: ofasr_bn_act_bwd_workspace(N, C) bytes.
int bn_bwd_reduce_coef(const void* dy, const void* x, const float* scale, const float* shift, const float* mean,
                       const float* invstd, float* dgamma, float* dbeta, float* ka, float* kbi, int64_t N, int64_t C,
                       int64_t HW, int act, int training, int dtype, void* workspace, size_t workspace_bytes,
                       void* stream);
// only the reduction pass: the partial slabs stay in `workspace` ([P][C][2] doubles, *P_out parts) for a consumer that
// folds them itself (BwdXf::fold_partial)
int bn_bwd_reduce_only(const void* dy, const void* x, const float* scale, const float* shift, const float* mean,
                       const float* invstd, int64_t N, int64_t C, int64_t HW, int act, int dtype, void* workspace,
                       size_t workspace_bytes, int* P_out, void* stream);

// A consumer kernel can fold the producer's epilogue partials itself (no finalize launch between the two): every
// block derives scale / mean / beta of the channels it reads from partial[c * P + 0..P) in fp64 in a fixed order, and
// one designated block also writes mean | invstd | scale | shift (kept for the backward) and the running statistics.
// cp == nullptr: not used (the InputXf pointers hold finalized values).
struct BnFold {
    const float2* cp;
    int P;
    double M, momentum, eps;
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    float* mean;
    float* invstd;
    float* scale;
    float* shift;
    int64_t* counters[3];   // num_batches_tracked of the block's BNs, bumped once by the designated writer (may be null)
};

#if defined(__HIPCC__)
// BatchNorm statistics epilogue: v[reg] is this lane's partial sum for accumulator register reg (row (reg & 3) + 8 * (reg >> 2) + 4 * h).
// A reduce-scatter over the 32 lanes of the half-wave (16 shuffles instead of 16 x 5) leaves in v[0] the total of
// register rho(c) = 8*b4 + 4*b3 + 2*b2 + b1 (bits of the lane's column c), identical in the two lanes of a b0 pair.
__device__ __forceinline__ void half_wave_row_sums(float (&v)[16], float (&w)[16], int c) {
    // both quantities level by level, so the 2 x (8 + 4 + 2 + 1 + 1) cross-lane moves form 5 dependent rounds, not 10
    const bool b4 = c & 16, b3 = c & 8, b2 = c & 4, b1 = c & 2;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float kv = b4 ? v[r + 8] : v[r], sv = b4 ? v[r] : v[r + 8];
        const float kw = b4 ? w[r + 8] : w[r], sw = b4 ? w[r] : w[r + 8];
        v[r] = kv + __shfl_xor(sv, 16, 64);
        w[r] = kw + __shfl_xor(sw, 16, 64);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float kv = b3 ? v[r + 4] : v[r], sv = b3 ? v[r] : v[r + 4];
        const float kw = b3 ? w[r + 4] : w[r], sw = b3 ? w[r] : w[r + 4];
        v[r] = kv + __shfl_xor(sv, 8, 64);
        w[r] = kw + __shfl_xor(sw, 8, 64);
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const float kv = b2 ? v[r + 2] : v[r], sv = b2 ? v[r] : v[r + 2];
        const float kw = b2 ? w[r + 2] : w[r], sw = b2 ? w[r] : w[r + 2];
        v[r] = kv + __shfl_xor(sv, 4, 64);
        w[r] = kw + __shfl_xor(sw, 4, 64);
    }
    {
        const float kv = b1 ? v[1] : v[0], sv = b1 ? v[0] : v[1];
        const float kw = b1 ? w[1] : w[0], sw = b1 ? w[0] : w[1];
        v[0] = kv + __shfl_xor(sv, 2, 64);
        w[0] = kw + __shfl_xor(sw, 2, 64);
    }
    v[0] += __shfl_xor(v[0], 1, 64);
    w[0] += __shfl_xor(w[0], 1, 64);
}
__device__ __forceinline__ int half_wave_row_reg(int c) { return ((c >> 4) & 1) * 8 + ((c >> 3) & 1) * 4 + ((c >> 2) & 1) * 2 + ((c >> 1) & 1); }

#endif

// Optional BatchNorm statistics of the tensor a conv kernel writes: every producer unit p (a pixel tile, a plane slab)
// leaves its per-channel (sum, sum of squares) at partial[c * P + p] -- fp32 over <= a few thousand values -- and
// bn_finalize_cp folds the P partials of a channel in fp64 in a fixed order.  partial == nullptr: no statistics.
struct StatOut {
    float2* partial;
    int P;
};
// number of producer units per channel of the kernels that take a StatOut
int dwconv_stat_units(int64_t N, int64_t H, int64_t W, int K, int dtype);
int pwconv_stat_units(int64_t N, int64_t Cin, int64_t HW);
// mean | invstd | scale | shift (+ running statistics, + up to three counters) from [C][P] partials
int bn_finalize_cp(const float2* partial, int64_t P, int64_t C, double count, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, double momentum, double eps, int training, float* mean,
                   float* invstd, float* scale, float* shift, int64_t* k0, int64_t* k1, int64_t* k2, void* stream);

int bn_fwd_cp(const void* x, const void* residual, void* y, const float2* partial, int64_t P, const float* gamma,
              const float* beta, float* running_mean, float* running_var, double momentum, double eps, int training,
              float* stats, int64_t N, int64_t C, int64_t HW, int act, int dtype, void* stream);

// ktransform.hip: the kernel transforms of all blocks of an MB stack in one launch per phase (fwd: filter; bwd: chain + matrices)
struct KtJob {
    const float* w_max;
    const int* ks;
    int nsteps;
    const float* const* mats;
    int transform;
    int64_t C;
    float* f;                 // fwd
    const float* df;          // bwd
    float* dw_max;
    float* const* dmats;
    void* ws;
    size_t ws_bytes;
};
int ktransform_fwd_batch(const KtJob* jobs, int n, void* stream);
int ktransform_bwd_batch(const KtJob* jobs, int n, void* stream);

// conv_thin.hip: the static convs with a 3-channel side (head / stem); conv2d.hip and conv2d_f32.hip route to them.
// "out": the THIN tensor is the result (forward with Cout thin, input gradient with Cin thin);
// "in": the THIN tensor is the operand; dgrad = 1 reads the [Cout][Cin][K][K] weights transposed and mirrored.
bool conv_thin_enabled();
bool conv_thin_out_supported(int64_t Ct, int64_t Cw, int K, int64_t W, int dtype, const void* x, const void* y);
int conv_thin_out_units(int64_t N, int64_t Ct, int64_t Cw, int64_t H, int64_t W, int dtype);
int conv_thin_out(const void* x, const float* w, void* y, int64_t N, int64_t Ct, int64_t Cw, int64_t H, int64_t W, int K,
                  int dtype, int dgrad, StatOut so, void* stream);
bool conv_thin_wgrad_supported(int64_t Ct, int64_t Cw, int K, int64_t H, int64_t W, int dtype, const void* a, const void* b);
size_t conv_thin_wgrad_workspace(int64_t N, int64_t Ct, int64_t Cw, int64_t H, int64_t W, int K, int dtype);
int conv_thin_wgrad(const void* dy, const void* x, float* dw, int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W, int K,
                    int dtype, void* workspace, size_t workspace_bytes, void* stream);
bool conv_thin_in_supported(int64_t Ct, int64_t Cw, int K, int64_t W, int dtype, const void* x, const void* y);
int conv_thin_in_units(int64_t N, int64_t Ct, int64_t Cw, int64_t H, int64_t W, int dtype);
int conv_thin_in(const void* x, const float* w, void* y, int64_t N, int64_t Ct, int64_t Cw, int64_t H, int64_t W, int K,
                 int dtype, int dgrad, StatOut so, void* stream);

// internal (not exported) variants used by mbconv.hip; they return OFASR_ERR_UNSUPPORTED (and launch nothing) when
// the vector / aligned 16-bit kernels that implement the fused read do not apply to the shape.
bool dwconv_xf_supported(const void* x, const void* y, int64_t H, int64_t W, int K, int dtype);
bool dwconv_stat_supported(const void* x, const void* y, int64_t H, int64_t W, int K, int dtype);
int dwconv_fwd_stat(const void* x, const float* f, void* y, int64_t N, int64_t C, int64_t H, int64_t W, int K, int dtype,
                    StatOut so, void* stream);
int dwconv_fwd_xf(const void* x, const float* f, void* y, int64_t N, int64_t C, int64_t H, int64_t W, int K, int dtype,
                  InputXf xf, void* stream, StatOut so = StatOut{nullptr, 0}, BnFold fold = BnFold{});
int dwconv_wgrad_xf(const void* dy, const void* x, float* df, int64_t N, int64_t C, int64_t H, int64_t W, int K,
                    int dtype, InputXf xf, void* workspace, size_t workspace_bytes, void* stream);
bool dwconv_wgrad_bx_supported(const void* da, const void* x, const void* y, int64_t N, int64_t C, int64_t H, int64_t W,
                               int K, int dtype);
int dwconv_wgrad_xf_bx(const void* da, const void* x, float* df, int64_t N, int64_t C, int64_t H, int64_t W, int K,
                       int dtype, InputXf xf, BwdXf bx, void* stream);
int dwconv_dgrad_bx(const void* da, const float* f, void* dx, int64_t N, int64_t C, int64_t H, int64_t W, int K,
                    int dtype, BwdXf bx, void* stream);
// ofasr_bn_finalize that also bumps up to three num_batches_tracked counters (thread 0)
int bn_finalize_bump(const void* workspace, int64_t n_partials, int64_t C, double count, const float* gamma,
                     const float* beta, float* running_mean, float* running_var, double momentum, double eps,
                     int training, float* mean, float* invstd, float* scale, float* shift, int64_t* k0, int64_t* k1,
                     int64_t* k2, void* stream);
// dx = W^T dy + addend (the identity shortcut's gradient joins in the epilogue of the expand conv's input gradient)
int pwconv_dgrad_add(const void* dy, const float* w, int64_t ldw, void* dx, const void* addend, int64_t N, int64_t Cin,
                     int64_t Cout, int64_t HW, int dtype, void* stream);
bool pwconv_dgrad_bx_supported(const void* da, const void* y, const void* dx, const void* addend, const float* w,
                               int64_t ldw, int64_t Cin, int64_t Cout, int64_t HW, int dtype);
int pwconv_dgrad_add_bx(const void* da, const float* w, int64_t ldw, void* dx, const void* addend, int64_t N, int64_t Cin,
                        int64_t Cout, int64_t HW, int dtype, BwdXf bx, void* stream);
bool pwconv_xf_supported(const void* x, const void* y, int64_t HW, int dtype);
int pwconv_fwd_xf(const void* x, const float* w, int64_t ldw, void* y, int64_t N, int64_t Cin, int64_t Cout, int64_t HW,
                  int dtype, InputXf xf, void* stream, StatOut so = StatOut{nullptr, 0});
// the same with the input BN's finalize folded into the kernel; only for the shapes pwconv_fold_supported accepts
bool pwconv_fold_supported(const void* x, const void* y, const float* w, int64_t ldw, int64_t Cin, int64_t HW, int dtype);
int pwconv_fwd_fold(const void* x, const float* w, int64_t ldw, void* y, int64_t N, int64_t Cin, int64_t Cout,
                    int64_t HW, int dtype, BnFold fold, void* stream, StatOut so = StatOut{nullptr, 0});
// plain forward (no input transform) that leaves the output's statistics partials
int pwconv_fwd_stat(const void* x, const float* w, int64_t ldw, void* y, int64_t N, int64_t Cin, int64_t Cout,
                    int64_t HW, int dtype, StatOut so, void* stream);
bool pwconv_wgrad_bx_supported(const void* da, const void* y, const void* x, int64_t Cin, int64_t Cout, int64_t HW, int dtype);
int pwconv_wgrad_bx(const void* da, const void* x, float* dw, int64_t ldw, int64_t N, int64_t Cin, int64_t Cout, int64_t HW,
                    int dtype, BwdXf bx, void* workspace, size_t workspace_bytes, void* stream);
int pwconv_wgrad_xf(const void* dy, const void* x, float* dw, int64_t ldw, int64_t N, int64_t Cin, int64_t Cout,
                    int64_t HW, int dtype, InputXf xf, void* workspace, size_t workspace_bytes, void* stream);

}  // namespace ofasr
