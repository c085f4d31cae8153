// conv_thin.hip -- the static convolutions with a THIN side (3 channels): the RGB head 64 -> 3 at HR resolution
// (reference ofa/elastic_nn/networks/ofa_mbs4.py:123) and the stem 3 -> 64 (ofa_mbs4.py:65), forward, input gradient and
// weight gradient, 16-bit and fp32 activations.
//
// These layers are HBM-bound by intensity (the head: 10.1 GFLOP over 140.5 MB in 16 bits = 72 flop/B, ridge 312), yet on
// the implicit-GEMM kernels of conv2d.hip / conv2d_f32.hip they occupy a 32-row matrix tile with 3 channels and re-read
// their input per kernel row (3.6-5.2x the algorithmic traffic, 0.07-0.16 of the HBM roof).  Here the thin side is
// packed with the kernel COLUMN into the matrix dimension, so that a 16-row tile is 15/16 full (3 channels x 5 taps),
// the wide tensor streams through a workgroup exactly once, and the remaining 1-D tap sum is a shift-and-add of a tiny
// fp32 image in LDS:
//
//   thin OUTPUT (head forward, stem input gradient)      ct_out_kernel
//       P[(t, kx)][col] = sum_ky sum_c  W[t][c][ky][kx] * X[c][y + ky - p][col]            (matrix cores, K = k * C)
//       out[t][y][x]    = sum_kx P[(t, kx)][x + kx - p]                                    (5 adds per output)
//     a workgroup owns a column strip of one image and walks down a row segment with a ring of k input rows in LDS
//     (every input row is requested from HBM once per segment, 3 rows ahead, staged through registers); 16-bit B
//     operands are transposing LDS reads of the [channel][column] rows as they lie in NCHW.
//   thin INPUT (stem forward, head input gradient)       ct_in_kernel
//       out[c][y][x] = sum_(t, ky) sum_kx W[c][t][ky][kx] * In[t][y + ky - p][x + kx - p]
//     k-slots = (t, ky) x 8 consecutive columns (16-bit: a B fragment is 8 adjacent input columns, no transpose;
//     the 3 slots beyond the k taps carry zero weights); the 64-channel result goes through LDS to 16-byte row stores.
//   thin WEIGHT GRADIENT                                  ct_wg_kernel + ct_wg_reduce_kernel
//       G[(t, sx)][c][sy] = sum_(n, y, x) Thin[t][y - sy][x - sx] * Wide[c][y][x]
//     k = 32 consecutive pixels of a row: the wide operand's fragment is one 16-byte global load per lane (no LDS),
//     the thin operand's 5 row shifts x 5 column shifts come from a small LDS image; a wave keeps the k x (C / 16)
//     accumulator tiles of all kernel rows over its rows, waves and workgroups are folded in a fixed order.
//
// All three take explicit weight strides / a flip flag, so that the forward of one layer and the input gradient of the
// other are the same kernel.  Entry points: csrc/conv2d.hip and csrc/conv2d_f32.hip route thin shapes here
// (conv_thin_*_supported); unsupported shapes keep the implicit-GEMM kernels.
#include <stdlib.h>
#include "ofasr_common.h"

namespace ofasr {

typedef __attribute__((ext_vector_type(4))) float ct_f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 ct_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 ct_f16x8;
typedef __attribute__((ext_vector_type(4))) short ct_s16x4;
typedef __attribute__((ext_vector_type(8))) short ct_s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int ct_u32x4;
typedef __attribute__((address_space(3))) ct_s16x4 ct_lds_s16x4;

constexpr int CT_ROWB = 288;             // bytes per channel row of a staged image row: 18 chunks of 16 B (72 dwords = 8 mod 64:
                                         // the 8 channel rows a half-wave's transposing read touches fall on disjoint banks)
constexpr int CT_CHUNKS = 18;
constexpr int CT_D = 4;                  // image rows requested ahead of the one being computed
constexpr int CT_PSTRIDE = 148;          // floats per row of the P image (4 * 148 = 16 mod 32: the two 16-lane groups of a
                                         // half-wave store to disjoint banks)

template <typename T> struct CtElem;
template <> struct CtElem<bf16_t> { static constexpr bool is16 = true; static constexpr int CW = 128, EPC = 8; };
template <> struct CtElem<f16_t> { static constexpr bool is16 = true; static constexpr int CW = 128, EPC = 8; };
template <> struct CtElem<float> { static constexpr bool is16 = false; static constexpr int CW = 64, EPC = 4; };
// CW: output columns per strip; EPC: elements per 16-byte chunk.  A staged row holds CT_CHUNKS * EPC columns starting at
// column x0 - EPC (one chunk of left halo, one of right halo).

template <typename T> __device__ __forceinline__ ct_f32x4 ct_mma16(ct_s16x8 a, ct_s16x8 b, ct_f32x4 c);
template <> __device__ __forceinline__ ct_f32x4 ct_mma16<bf16_t>(ct_s16x8 a, ct_s16x8 b, ct_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ct_bf16x8, a), __builtin_bit_cast(ct_bf16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ ct_f32x4 ct_mma16<f16_t>(ct_s16x8 a, ct_s16x8 b, ct_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ct_f16x8, a), __builtin_bit_cast(ct_f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ ct_f32x4 ct_mma32(float a, float b, ct_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ ct_s16x4 ct_tr_read(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((ct_lds_s16x4*)p);
}

// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for vmcnt(0), i.e. for the image rows that
// were requested three rows ahead precisely so that nobody has to wait for them
__device__ __forceinline__ void ct_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename T> __device__ __forceinline__ unsigned short ct_bits16(float v) { return from_float<T>(v).v; }

// weight of (thin channel t, wide channel c, kernel row ky, kernel column kx) in the [Cout][Cin][k][k] parameter:
// forward of the head: t = co, c = ci; input gradient of the stem: t = ci, c = co with both taps mirrored
struct CtW {
    const float* w;
    int st, sc;      // element strides of t and c
    int flip;
};
template <int K> __device__ __forceinline__ int ct_widx(const CtW& W, int t, int c, int ky, int kx) {
    const int tap = W.flip ? (K - 1 - ky) * K + (K - 1 - kx) : ky * K + kx;
    return t * W.st + c * W.sc + tap;
}
// the whole parameter (Ct * Cw * K * K floats, < 26 KB) copied to LDS with coalesced requests: every lane then gathers the
// ~80 elements of its matrix fragments from there.  Gathering them from global memory is 80 wave-instructions of 64
// scattered 4-byte requests per wave -- measured: a third of the whole kernel's time.
__device__ __forceinline__ void ct_stage_weights(float* dst, const float* w, int count, int tid, int nthreads) {
    for (int i = tid; i < count; i += nthreads) dst[i] = w[i];
}

struct CtGeom {
    int N, H, W;        // images, rows, columns
    int Ct, Cw;         // thin / wide channel counts
    int strips, segs, RS;   // column strips per row, row segments per image, rows per segment
};

// ------------------------------------------------------------------------------------------------------------------
// thin OUTPUT: X [N][Cw][H][W] -> out [N][Ct][H][W], Ct * K <= 16, Cw = 32 * NC
//
// 16-bit: 9 waves, wave w owns column tile w of the strip (9 tiles of 16 columns cover the 128 + 2 * (k / 2) columns P is
//   needed at) and chains the k * NC matrix instructions of all kernel rows into one accumulator tile; the ring holds
//   k + 1 rows and the P image is double-buffered, so a row costs ONE workgroup barrier.
// fp32: 8 waves, items (column tile, kernel row, 32-channel chunk) dealt round robin (5 * k * NC items of 8 matrix
//   instructions: the kernel is bound by the fp32 matrix rate, 10.7 GFLOP at 157 TFLOP/s), one P plane per (kernel row,
//   chunk), two barriers per row.
// ------------------------------------------------------------------------------------------------------------------
template <typename T> struct CtOut {
    static constexpr bool is16 = CtElem<T>::is16;
    static constexpr int THREADS = is16 ? 576 : 512;
    static constexpr int NT = is16 ? 9 : 5;                // 16-column tiles
    static constexpr int PSTR = is16 ? CT_PSTRIDE : 84;    // floats per P row (4 * PSTR = 16 mod 32)
    static constexpr int ring_rows(int K) { return is16 ? K + 1 : K; }
    static constexpr int planes(int K, int NC) { return is16 ? 2 : K * NC; }
    static constexpr size_t lds_bytes(int K, int NC) {
        return (size_t)ring_rows(K) * 32 * NC * CT_ROWB + (size_t)planes(K, NC) * 16 * PSTR * sizeof(float);
    }
};

template <typename T, int K, int NC>
__global__ __launch_bounds__(CtOut<T>::THREADS) void ct_out_kernel(const T* __restrict__ X, CtW Wt, T* __restrict__ out,
                                                                   CtGeom g, StatOut so) {
    using CO = CtOut<T>;
    constexpr bool is16 = CO::is16;
    constexpr int THREADS = CO::THREADS, WAVES = THREADS / 64, NT = CO::NT, PSTR = CO::PSTR;
    constexpr int CW = CtElem<T>::CW, EPC = CtElem<T>::EPC, P = K / 2, CWD = 32 * NC;
    constexpr int ROW_BYTES = CWD * CT_ROWB;                  // one staged image row
    constexpr int RR = CO::ring_rows(K);
    constexpr int NCH = CWD * CT_CHUNKS;                      // 16-byte chunks per staged row
    constexpr int SLOTS = (NCH + THREADS - 1) / THREADS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    float* Pb = reinterpret_cast<float*>(smem + RR * ROW_BYTES);   // [planes][16][PSTR]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // The strips of one (image, segment) request each other's edge lines (a 16-byte halo chunk lies in the neighbour's
    // 128-byte line), so they must share an L2: workgroups go to the 8 XCDs round robin by blockIdx, hence the strips of
    // a unit are 8 apart in blockIdx and run side by side on one XCD (measured without: 1.95x the algorithmic reads).
    int strip, unit;
    {
        const int b = blockIdx.x, units = g.N * g.segs;
        if (units % 8 == 0) {
            const int grp = b / (8 * g.strips), rem = b % (8 * g.strips);
            strip = rem / 8;
            unit = grp * 8 + rem % 8;
        } else {
            strip = b % g.strips;
            unit = b / g.strips;
        }
    }
    const int seg = unit % g.segs;
    const int n = unit / g.segs;
    const int x0 = strip * CW;
    const int ys = seg * g.RS, ye = min(ys + g.RS, g.H);
    const long long plane = (long long)g.H * g.W;
    const T* Xn = X + (long long)n * g.Cw * plane;

    // ---- the weight operand: A[m = t * K + kx][k = (ky, c)], kept in registers for the whole block ----------------
    const int m = lane & 15, kg = lane >> 4;
    const bool mvalid = m < g.Ct * K;
    const int mt = mvalid ? m / K : 0, mkx = m % K;   // rows beyond Ct * K are zero rows (select after a valid read)
    float* wl = reinterpret_cast<float*>(ring);       // the ring is not in use yet
    ct_stage_weights(wl, Wt.w, g.Ct * g.Cw * K * K, tid, THREADS);
    __syncthreads();
    ct_s16x8 a16[is16 ? K * NC : 1];
    float a32[is16 ? 1 : K * 8 * NC];
    if constexpr (is16) {
#pragma unroll
        for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int cc = 0; cc < NC; ++cc) {
                ct_s16x8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    // the transposing reads hand lane group kg the channels 4 kg .. 4 kg + 3 (first read) and
                    // 16 + 4 kg .. (second read) of the 32-channel chunk
                    const int c = 32 * cc + (e < 4 ? 4 * kg + e : 16 + 4 * kg + (e - 4));
                    const float wv = wl[ct_widx<K>(Wt, mt, c, ky, mkx)];
                    v[e] = (short)ct_bits16<T>(mvalid ? wv : 0.f);
                }
                a16[ky * NC + cc] = v;
            }
    } else {
        const int perm = (kg == 1) ? 2 : (kg == 2 ? 1 : kg);   // lane group -> channel of a 4-channel step (bank-disjoint halves)
#pragma unroll
        for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int s = 0; s < 8 * NC; ++s) {
                const float wv = wl[ct_widx<K>(Wt, mt, 4 * s + perm, ky, mkx)];
                a32[ky * 8 * NC + s] = mvalid ? wv : 0.f;
            }
    }
    __syncthreads();   // the gathers are done: the ring may be filled

    // ---- staging of input rows: chunk q * THREADS + tid of a row, requested CT_D rows ahead ---------------------------
    int ch_off[SLOTS];        // byte offset of the chunk inside a staged row (LDS)
    long long g_off[SLOTS];   // element offset of the chunk inside an image row of X (channel * plane + column)
    bool col_ok[SLOTS];
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) {
        // slots beyond the row's chunk count wrap around: such a lane requests and writes a chunk some other lane handles
        // as well (same bytes to the same place).  A slot that is skipped under a branch instead leaves its request
        // "pending" on that path in the compiler's wait-count analysis, which then drains ALL requests at every re-use.
        const int id = (q * THREADS + tid) % NCH;
        const int c = id / CT_CHUNKS, jc = id % CT_CHUNKS;
        const int col = x0 - EPC + jc * EPC;
        col_ok[q] = col >= 0 && col < g.W;
        ch_off[q] = c * CT_ROWB + jc * 16;
        g_off[q] = (long long)c * plane + (col_ok[q] ? col : 0);
    }
    ct_u32x4 stage[CT_D][SLOTS];
    auto request = [&](int d, int row) {
        // rows above / below the image are zero (handled at the LDS write); the request itself goes to a clamped row,
        // and rows beyond what this segment needs re-request the last needed row (a cache hit, no HBM traffic)
        int r = min(row, min(ye - 1 + P, g.H - 1));
        r = max(r, 0);
#pragma unroll
        for (int q = 0; q < SLOTS; ++q)
            stage[d][q] = *reinterpret_cast<const ct_u32x4*>(Xn + g_off[q] + (long long)r * g.W);
    };
    auto deposit = [&](int d, int row) {
        const bool row_ok = row >= 0 && row < g.H;
        char* slot = ring + ((row + RR * 4096) % RR) * ROW_BYTES;
#pragma unroll
        for (int q = 0; q < SLOTS; ++q) {
            ct_u32x4 v = stage[d][q];
            if (!(row_ok && col_ok[q])) v = ct_u32x4{0u, 0u, 0u, 0u};
            *reinterpret_cast<ct_u32x4*>(slot + ch_off[q]) = v;
        }
    };

    // prologue: rows ys - P .. ys + P into the ring, rows ys + P + 1 .. ys + P + CT_D - 1 requested
    {
        // all k rows requested together (one memory latency, not k of them), then written to their slots
        ct_u32x4 pro[K][SLOTS];
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int r = max(min(ys - P + i, g.H - 1), 0);
#pragma unroll
            for (int q = 0; q < SLOTS; ++q)
                pro[i][q] = *reinterpret_cast<const ct_u32x4*>(Xn + g_off[q] + (long long)r * g.W);
        }
#pragma unroll
        for (int d = 1; d < CT_D; ++d) request((2 * P + d) % CT_D, ys + P + d);   // input row i lives in set (i - (ys - P)) % CT_D
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int row = ys - P + i;
            const bool row_ok = row >= 0 && row < g.H;
            char* slot = ring + ((row + RR * 4096) % RR) * ROW_BYTES;
#pragma unroll
            for (int q = 0; q < SLOTS; ++q) {
                ct_u32x4 v = pro[i][q];
                if (!(row_ok && col_ok[q])) v = ct_u32x4{0u, 0u, 0u, 0u};
                *reinterpret_cast<ct_u32x4*>(slot + ch_off[q]) = v;
            }
        }
    }
    __syncthreads();

    float ssum = 0.f, ssq = 0.f;            // BatchNorm statistics of what this thread stores (StatOut)
    const int o_t = tid / CW, o_x = tid % CW;  // this thread's output (thin channel, column) -- Ct * CW <= THREADS
    const bool o_ok = tid < g.Ct * CW && x0 + o_x < g.W;

    // The row loop is unrolled CT_D times so that the staging register sets are named, and its body is straight-line code
    // (whole trips only: the up to CT_D - 1 rows beyond the segment are computed from re-requested rows and not stored).
    // A skip-or-leave branch per row makes the compiler assume requests still in flight on some path and drain them
    // all -- vmcnt(0) -- at every re-use of a staging register, which serialises the kernel on the memory latency.
    const int trips = (ye - ys + CT_D - 1) / CT_D;
    for (int it = 0; it < trips; ++it) {
        const int y = ys + it * CT_D;
#pragma unroll
        for (int d = 0; d < CT_D; ++d) {
            const int r = y + d;
            const bool o_row = o_ok && r < ye;
            {
                // request input row r + P + CT_D into the set that held row r + P (deposited during row r - 1)
                request((2 * P + d) % CT_D, r + P + CT_D);
                __builtin_amdgcn_sched_barrier(0);   // keep the requests at the top of the row: that is their head start
                if constexpr (is16) {
                    // the ring has k + 1 slots: row r + P + 1 replaces row r - P - 1, which nobody reads any more
                    deposit((2 * P + d + 1) % CT_D, r + P + 1);
                    ct_s16x8 bf[K * NC];
                    const int l16 = lane & 15, q4 = l16 >> 2, p4 = l16 & 3;
                    const int lane_off = (4 * kg + q4) * CT_ROWB + 32 * wave + 8 * p4;
#pragma unroll
                    for (int ky = 0; ky < K; ++ky) {
                        const char* base = ring + ((r + ky - P + RR * 4096) % RR) * ROW_BYTES + lane_off;
#pragma unroll
                        for (int cc = 0; cc < NC; ++cc) {
                            const ct_s16x4 lo = ct_tr_read(base + (32 * cc) * CT_ROWB);
                            const ct_s16x4 hi = ct_tr_read(base + (32 * cc + 16) * CT_ROWB);
                            bf[ky * NC + cc] = ct_s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        }
                    }
                    ct_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < K * NC; ++i) {
                        if (i & 1) acc1 = ct_mma16<T>(a16[i], bf[i], acc1);
                        else acc0 = ct_mma16<T>(a16[i], bf[i], acc0);
                    }
                    float* pr = Pb + ((r & 1) * 16 + 4 * kg) * PSTR + 16 * wave + (lane & 15);
#pragma unroll
                    for (int i = 0; i < 4; ++i) pr[i * PSTR] = acc0[i] + acc1[i];
                    ct_lds_barrier();
                    if (o_row) {
                        float v = 0.f;
                        const float* ps = Pb + ((r & 1) * 16 + o_t * K) * PSTR + o_x + EPC - P;
#pragma unroll
                        for (int kx = 0; kx < K; ++kx) v += ps[kx * PSTR + kx];
                        const T ov = from_float<T>(v);
                        out[((long long)n * g.Ct + o_t) * plane + (long long)r * g.W + x0 + o_x] = ov;
                        const float vs = to_float(ov);
                        ssum += vs;
                        ssq += vs * vs;
                    }
                } else {
                    // items (column tile t, kernel row ky, chunk cc) dealt round robin: item (ky * NC + cc) * NT + t to wave
                    // ((ky * NC + cc) * NT + t) % 8; (ky, cc) unrolled so that the weight registers are named
                    const int perm = (kg == 1) ? 2 : (kg == 2 ? 1 : kg);
#pragma unroll
                    for (int ky = 0; ky < K; ++ky)
#pragma unroll
                        for (int cc = 0; cc < NC; ++cc)
                            for (int t = (wave + WAVES * K * NC * NT - (ky * NC + cc) * NT) % WAVES; t < NT; t += WAVES) {
                                const char* base = ring + ((r + ky - P + RR * 4096) % RR) * ROW_BYTES +
                                                   (32 * cc + perm) * CT_ROWB + (16 * t + (lane & 15)) * 4;
                                float bv[8];
#pragma unroll
                                for (int s = 0; s < 8; ++s) bv[s] = *reinterpret_cast<const float*>(base + (4 * s) * CT_ROWB);
                                ct_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                                for (int s = 0; s < 8; s += 2) {
                                    acc0 = ct_mma32(a32[ky * 8 * NC + 8 * cc + s], bv[s], acc0);
                                    acc1 = ct_mma32(a32[ky * 8 * NC + 8 * cc + s + 1], bv[s + 1], acc1);
                                }
                                float* pr = Pb + ((ky * NC + cc) * 16 + 4 * kg) * PSTR + 16 * t + (lane & 15);
#pragma unroll
                                for (int i = 0; i < 4; ++i) pr[i * PSTR] = acc0[i] + acc1[i];
                            }
                    ct_lds_barrier();
                    if (o_row) {
                        float v = 0.f;
                        const float* ps = Pb + (o_t * K) * PSTR + o_x + EPC - P;
#pragma unroll
                        for (int pl = 0; pl < K * NC; ++pl)
#pragma unroll
                            for (int kx = 0; kx < K; ++kx) v += ps[(pl * 16 + kx) * PSTR + kx];
                        const T ov = from_float<T>(v);
                        out[((long long)n * g.Ct + o_t) * plane + (long long)r * g.W + x0 + o_x] = ov;
                        const float vs = to_float(ov);
                        ssum += vs;
                        ssq += vs * vs;
                    }
                    deposit((2 * P + d + 1) % CT_D, r + P + 1);   // into the slot of row r - P, free after the barrier above
                    ct_lds_barrier();
                }
            }
        }
    }

    if (so.partial != nullptr) {
        // per-channel partial of this block: threads [t * CW, (t + 1) * CW) own channel t
        __syncthreads();
        float* red = Pb;   // the P image is free now
        const float a = wave_sum(ssum), q = wave_sum(ssq);
        if (lane == 0) {
            red[wave] = a;
            red[16 + wave] = q;
        }
        __syncthreads();
        constexpr int WPC = CW / 64;   // waves per channel
        if (tid < g.Ct) {
            float s0 = 0.f, q0 = 0.f;
            for (int w = 0; w < WPC; ++w) {
                s0 += red[tid * WPC + w];
                q0 += red[16 + tid * WPC + w];
            }
            so.partial[(long long)tid * so.P + blockIdx.x] = make_float2(s0, q0);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// thin INPUT: In [N][Ct][H][W] -> out [N][Cw][H][W], Ct * K <= 16, Cw = 16 * NM
//
// D[pixel][channel] = A[pixel][k] * B[k][channel]: the PIXELS are the matrix rows, so a lane's 4 accumulator registers are
// 4 adjacent columns of one output channel (one 8 / 16-byte LDS store), and the thin operand needs no transpose:
//   16-bit: k-slot (combo (t, ky), e) = In[t][y + ky - p][x - p + e], e = 0..7 -- 8 ADJACENT input columns per lane (five
//           4-byte LDS reads and a funnel shift for odd starts); the weights of slots e >= k are zero;
//   fp32:   k = (t, ky, kx), one LDS dword per lane and step.
// The thin rows of the whole segment (+ halo) are loaded once into LDS (a few tens of KB), so the row loop has no global
// request in it; the wide result goes through a double-buffered [channel][column] LDS tile to 16-byte row stores.
// ------------------------------------------------------------------------------------------------------------------
constexpr int CT_IN_THREADS = 512;
constexpr int CT_OSTR = 272;   // bytes per channel row of the output tile: 68 dwords = 4 mod 32 (2-way at worst, the minimum)
constexpr int CT_IN_MAXRS = 64;

template <typename T> struct CtIn {
    static constexpr bool is16 = CtElem<T>::is16;
    static constexpr int CTILES = CtElem<T>::CW / 16;        // column tiles per strip: 8 (16-bit), 4 (fp32)
    static constexpr int WPT = 8 / CTILES;                   // waves per column tile: 1 / 2
    static constexpr size_t lds_bytes(int K, int Ct, int Cw, int RS) {
        return (size_t)(RS + K - 1) * Ct * CT_ROWB + 2 * (size_t)Cw * CT_OSTR + 64;
    }
};

template <typename T, int K, int NM>
__global__ __launch_bounds__(CT_IN_THREADS) void ct_in_kernel(const T* __restrict__ In, CtW Wt, T* __restrict__ out, CtGeom g,
                                                              StatOut so) {
    using CI = CtIn<T>;
    constexpr bool is16 = CI::is16;
    constexpr int CW = CtElem<T>::CW, EPC = CtElem<T>::EPC, P = K / 2, CWD = 16 * NM;
    constexpr int CTILES = CI::CTILES, WPT = CI::WPT, MPW = NM / WPT;   // channel tiles per wave
    constexpr int KS = is16 ? 4 : (4 * K * K + 3) / 4;                 // k-steps (Ct <= 4: 16 combos / 4 K^2 products at most)
    static_assert(NM % WPT == 0, "channel tiles split evenly over the waves of a column tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    int strip, unit;
    {
        const int b = blockIdx.x, units = g.N * g.segs;
        if (units % 8 == 0) {   // (as ct_out_kernel: the strips of a unit on one XCD -- they share the thin rows in L2)
            const int grp = b / (8 * g.strips), rem = b % (8 * g.strips);
            strip = rem / 8;
            unit = grp * 8 + rem % 8;
        } else {
            strip = b % g.strips;
            unit = b / g.strips;
        }
    }
    const int seg = unit % g.segs, n = unit / g.segs;
    const int x0 = strip * CW;
    const int ys = seg * g.RS, ye = min(ys + g.RS, g.H);
    const int nrows = ye - ys + K - 1;                      // thin rows held: ys - P .. ye - 1 + P
    const long long plane = (long long)g.H * g.W;
    char* thin = smem;                                      // [nrows][Ct][CT_ROWB]
    char* obuf = smem + (size_t)(g.RS + K - 1) * g.Ct * CT_ROWB;   // 2 x [CWD][CT_OSTR]
    float* wl = reinterpret_cast<float*>(obuf);             // the parameter, staged through the (not yet used) output tiles

    // ---- thin rows of the segment -> LDS (all requests in flight together), the weights -> LDS -> registers -----------
    {
        const T* In_n = In + (long long)n * g.Ct * plane;
        const int total = nrows * g.Ct * CT_CHUNKS;
        for (int id0 = 0; id0 < total; id0 += 4 * CT_IN_THREADS) {
            ct_u32x4 v[4];
            int off[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int id = min(id0 + u * CT_IN_THREADS + tid, total - 1);   // (the tail repeats the last chunk)
                const int jc = id % CT_CHUNKS, rc = id / CT_CHUNKS;
                const int t = rc % g.Ct, row = ys - P + rc / g.Ct;
                const int col = x0 - EPC + jc * EPC;
                const bool ok = row >= 0 && row < g.H && col >= 0 && col < g.W;
                const long long src = (long long)t * plane + (long long)(ok ? row : 0) * g.W + (ok ? col : 0);
                v[u] = *reinterpret_cast<const ct_u32x4*>(In_n + src);
                if (!ok) v[u] = ct_u32x4{0u, 0u, 0u, 0u};
                off[u] = rc * CT_ROWB + jc * 16;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) *reinterpret_cast<ct_u32x4*>(thin + off[u]) = v[u];
        }
        ct_stage_weights(wl, Wt.w, g.Ct * g.Cw * K * K, tid, CT_IN_THREADS);
    }
    __syncthreads();

    const int ctile = wave % CTILES;              // this wave's column tile
    const int mbase = (wave / CTILES) * MPW;      // and its first channel tile
    const int l16 = lane & 15, kg = lane >> 4;
    const int ncomb = g.Ct * K;
    // B operand (weights): lane (channel l16 of the tile, group kg)
    ct_s16x8 b16[is16 ? MPW * KS : 1];
    float b32[is16 ? 1 : MPW * KS];
#pragma unroll
    for (int mi = 0; mi < MPW; ++mi) {
        const int c = 16 * (mbase + mi) + l16;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if constexpr (is16) {
                const int comb = 4 * s + kg;
                const bool cv = comb < ncomb;
                const int t = cv ? comb / K : 0, ky = comb % K;
                ct_s16x8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float wv = wl[ct_widx<K>(Wt, t, c, ky, e < K ? e : 0)];
                    v[e] = (short)ct_bits16<T>((cv && e < K) ? wv : 0.f);
                }
                b16[mi * KS + s] = v;
            } else {
                const int k = 4 * s + kg;
                const bool kv = k < ncomb * K;
                const int kk = kv ? k : 0;
                const float wv = wl[ct_widx<K>(Wt, kk / (K * K), c, (kk / K) % K, kk % K)];
                b32[mi * KS + s] = kv ? wv : 0.f;
            }
        }
    }
    __syncthreads();   // the output tiles may be written now

    // A operand addressing: pixel l16 of the column tile; LDS column index of its first tap
    const int j0 = 16 * ctile + l16 + EPC - P;
    int a_off[KS];      // byte offset of the step's (row, channel) inside the thin image, relative to output row 0 of the segment
    if constexpr (is16) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int comb = min(4 * s + kg, ncomb - 1);
            a_off[s] = ((comb % K) * g.Ct + comb / K) * CT_ROWB + (j0 & ~1) * 2;
        }
    } else {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k = min(4 * s + kg, ncomb * K - 1);
            a_off[s] = (((k / K) % K) * g.Ct + k / (K * K)) * CT_ROWB + (j0 + k % K) * 4;
        }
    }
    const int sh = (j0 & 1) * 16;

    float ssum[MPW], ssq[MPW];
#pragma unroll
    for (int mi = 0; mi < MPW; ++mi) ssum[mi] = ssq[mi] = 0.f;
    bool px_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) px_ok[i] = x0 + 16 * ctile + 4 * kg + i < g.W;

    for (int r = ys; r < ye; ++r) {
        const char* rowp = thin + (r - ys) * g.Ct * CT_ROWB;   // thin row r - P is held row r - ys
        ct_f32x4 acc[MPW];
#pragma unroll
        for (int mi = 0; mi < MPW; ++mi) acc[mi] = ct_f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (is16) {
            ct_s16x8 af[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const unsigned int* q = reinterpret_cast<const unsigned int*>(rowp + a_off[s]);
                unsigned int dw[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) dw[i] = q[i];
                ct_u32x4 pk;
#pragma unroll
                for (int i = 0; i < 4; ++i) pk[i] = __builtin_amdgcn_alignbit(dw[i + 1], dw[i], sh);
                af[s] = __builtin_bit_cast(ct_s16x8, pk);
            }
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int mi = 0; mi < MPW; ++mi) acc[mi] = ct_mma16<T>(af[s], b16[mi * KS + s], acc[mi]);
        } else {
            const int ks_used = (ncomb * K + 3) / 4;   // 19 of the 25 steps for 3 thin channels (wave-uniform)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s < ks_used) {
                    const float av = *reinterpret_cast<const float*>(rowp + a_off[s]);
#pragma unroll
                    for (int mi = 0; mi < MPW; ++mi) acc[mi] = ct_mma32(av, b32[mi * KS + s], acc[mi]);
                }
            }
        }
        // accumulator register i = pixel 4 kg + i of the tile, lane's channel l16: one 8 / 16-byte store per channel tile
        char* ob = obuf + (r & 1) * (CWD * CT_OSTR);
#pragma unroll
        for (int mi = 0; mi < MPW; ++mi) {
            char* dst = ob + (16 * (mbase + mi) + l16) * CT_OSTR + (16 * ctile + 4 * kg) * (int)sizeof(T);
            float vs[4];
            if constexpr (is16) {
                uint2 pk;
                pk.x = pack2<T>(acc[mi][0], acc[mi][1]);
                pk.y = pack2<T>(acc[mi][2], acc[mi][3]);
                *reinterpret_cast<uint2*>(dst) = pk;
                unpack2<T>(pk.x, vs[0], vs[1]);
                unpack2<T>(pk.y, vs[2], vs[3]);
            } else {
                *reinterpret_cast<ct_f32x4*>(dst) = acc[mi];
#pragma unroll
                for (int i = 0; i < 4; ++i) vs[i] = acc[mi][i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = px_ok[i] ? vs[i] : 0.f;
                ssum[mi] += v;
                ssq[mi] += v * v;
            }
        }
        ct_lds_barrier();
        // read the tile out row-wise: CWD channels x 16 chunks of 16 bytes
        T* orow = out + (long long)n * g.Cw * plane + (long long)r * g.W + x0;
#pragma unroll
        for (int u = 0; u < (CWD * 16 + CT_IN_THREADS - 1) / CT_IN_THREADS; ++u) {
            const int id = u * CT_IN_THREADS + tid;
            const int c = id >> 4, ch = id & 15;
            if (id < CWD * 16 && x0 + ch * EPC < g.W) {
                const ct_u32x4 v = *reinterpret_cast<const ct_u32x4*>(ob + c * CT_OSTR + ch * 16);
                *reinterpret_cast<ct_u32x4*>(orow + (long long)c * plane + ch * EPC) = v;
            }
        }
        // (no second barrier: the next row writes the other tile; the row after that is behind the next barrier)
    }

    if (so.partial != nullptr) {
        // lane (channel l16, pixel group kg): fold the 4 pixel groups, then the column tiles of the workgroup
        __syncthreads();
        float* red = reinterpret_cast<float*>(obuf);   // [wave][MPW][16][2]
#pragma unroll
        for (int mi = 0; mi < MPW; ++mi) {
            float a = ssum[mi], q = ssq[mi];
            a += __shfl_xor(a, 16, 64); q += __shfl_xor(q, 16, 64);
            a += __shfl_xor(a, 32, 64); q += __shfl_xor(q, 32, 64);
            if (kg == 0) {
                red[((wave * MPW + mi) * 16 + l16) * 2] = a;
                red[((wave * MPW + mi) * 16 + l16) * 2 + 1] = q;
            }
        }
        __syncthreads();
        if (tid < CWD) {
            const int mt = tid >> 4, ch = tid & 15;
            float a = 0.f, q = 0.f;
            for (int ct = 0; ct < CTILES; ++ct) {        // the waves that own channel tile mt, in a fixed order
                const int w = (mt / MPW) * CTILES + ct, mi = mt % MPW;
                a += red[((w * MPW + mi) * 16 + ch) * 2];
                q += red[((w * MPW + mi) * 16 + ch) * 2 + 1];
            }
            so.partial[(long long)tid * so.P + blockIdx.x] = make_float2(a, q);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// thin WEIGHT GRADIENT:  G[(t, sxi)][c][syi] = sum over the block's pixels of Thin[t][y - syi + p][x - sxi + p] * Wide[c][y][x]
//
// k = 32 consecutive pixels of a row.  B (wide): lane (channel l16 of a 16-channel tile, group kg) requests the 8 pixels
// xb + 8 kg .. +7 of its channel row straight from global memory -- 16 bytes (two requests in fp32), no LDS; the requests
// of an item are issued CT_WPD items ahead.  A (thin): lane (row m = t * k + sxi, group kg) reads the same 8 pixels shifted
// by sxi - p from the LDS image of the segment's thin rows (16-bit: five dwords and a funnel shift).  A wave walks the
// items (row, 32-pixel chunk) i = wave, wave + 8, ... of its block with the k x NT accumulator tiles of ALL kernel rows
// in registers; at the end the waves add their tiles into one LDS slab in wave order and the block writes the slab;
// ct_wg_reduce_kernel sums the slabs in block order into the parameter's gradient (deterministic, no atomics).
// ------------------------------------------------------------------------------------------------------------------
constexpr int CT_WG_THREADS = 512;
constexpr int CT_WPD = 3;          // items whose wide fragments are in flight ahead of the one being multiplied

struct CtWgGeom {
    int N, H, W, Ct, Cw;
    int segs, RS;          // row segments per image, rows per segment
    int nchunks;           // 32-pixel chunks per row
    int trow;              // bytes per (row, channel) of the thin LDS image: (W + 16 + pad) elements
};

template <typename T, int K, int NT>
__global__ __launch_bounds__(CT_WG_THREADS) void ct_wg_kernel(const T* __restrict__ Thin, const T* __restrict__ Wide,
                                                              float* __restrict__ slabs, CtWgGeom g) {
    constexpr bool is16 = CtElem<T>::is16;
    constexpr int EPC = CtElem<T>::EPC, P = K / 2, ES = (int)sizeof(T);
    constexpr int BV = is16 ? 1 : 2;                  // 16-byte requests per lane and channel tile
    constexpr int HALO = 8;                           // thin image: column index = column + HALO
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int seg = blockIdx.x % g.segs, n = blockIdx.x / g.segs;
    const int ys = seg * g.RS, ye = min(ys + g.RS, g.H);
    const int nheld = ye - ys + K - 1;                // thin rows ys - P .. ye - 1 + P
    const long long plane = (long long)g.H * g.W;
    char* thin = smem;

    // ---- the thin rows of the segment -> LDS: [held row][t][trow], zero outside the image --------------------------
    {
        const T* Tn = Thin + (long long)n * g.Ct * plane;
        const int cpr = g.trow / 16;                  // 16-byte chunks per (row, channel)
        const int total = nheld * g.Ct * cpr;
        for (int id0 = 0; id0 < total; id0 += 4 * CT_WG_THREADS) {
            ct_u32x4 v[4];
            int off[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int id = min(id0 + u * CT_WG_THREADS + tid, total - 1);
                const int jc = id % cpr, rc = id / cpr;
                const int t = rc % g.Ct, row = ys - P + rc / g.Ct;
                const int col = jc * EPC - HALO;
                const bool ok = row >= 0 && row < g.H && col >= 0 && col < g.W;
                const long long src = (long long)t * plane + (long long)(ok ? row : 0) * g.W + (ok ? col : 0);
                v[u] = *reinterpret_cast<const ct_u32x4*>(Tn + src);
                if (!ok) v[u] = ct_u32x4{0u, 0u, 0u, 0u};
                off[u] = rc * g.trow + jc * 16;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) *reinterpret_cast<ct_u32x4*>(thin + off[u]) = v[u];
        }
    }
    __syncthreads();

    const int l16 = lane & 15, kg = lane >> 4;
    const bool mvalid = l16 < g.Ct * K;
    const int mt = mvalid ? l16 / K : 0, msx = l16 % K;          // this lane's A row: thin channel, column shift index
    // A addressing: first element = column xb + 8 kg - (msx - P), image index + HALO
    const int ja = 8 * kg - msx + P + HALO;                      // + xb
    const int a_lane = mt * g.trow + (is16 ? (ja & ~1) * 2 : ja * 4);
    const int sh = (ja & 1) * 16;
    const T* Wn = Wide + (long long)n * g.Cw * plane;
    const long long b_lane = (long long)l16 * plane + 8 * kg;     // + tile * 16 * plane + row * W + xb

    ct_f32x4 acc[K][NT];
#pragma unroll
    for (int sy = 0; sy < K; ++sy)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[sy][nt] = ct_f32x4{0.f, 0.f, 0.f, 0.f};

    const int nitems = (ye - ys) * g.nchunks;
    const int per_wave = (nitems + 7) / 8;
    const int trips = (per_wave + CT_WPD - 1) / CT_WPD;
    ct_u32x4 bst[CT_WPD][NT][BV];
    auto request = [&](int d, int it) {
        const int i = min(wave + 8 * it, nitems - 1);            // items beyond the block's last repeat it (and contribute zero)
        const int row = ys + i / g.nchunks, xb = 32 * (i % g.nchunks);
        const bool ok = xb + 8 * kg < g.W;                        // (W % 8 == 0: the 8 pixels are all in or all out)
        const T* src = Wn + b_lane + (long long)row * g.W + (ok ? xb : 0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int q = 0; q < BV; ++q)
                bst[d][nt][q] = *reinterpret_cast<const ct_u32x4*>(src + (long long)nt * 16 * plane + q * 4);
    };
#pragma unroll
    for (int d = 0; d < CT_WPD - 1; ++d) request(d, d);

    for (int tr = 0; tr < trips; ++tr) {
#pragma unroll
        for (int d = 0; d < CT_WPD; ++d) {
            const int it = tr * CT_WPD + d;
            request((d + CT_WPD - 1) % CT_WPD, it + CT_WPD - 1);
            __builtin_amdgcn_sched_barrier(0);
            const int i = wave + 8 * it;
            const bool live = i < nitems;
            const int ic = min(i, nitems - 1);
            const int rrel = ic / g.nchunks, xb = 32 * (ic % g.nchunks);
            const bool ok = live && xb + 8 * kg < g.W;
            const char* abase = thin + a_lane + xb * ES;
            if constexpr (is16) {
                ct_s16x8 bf[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    ct_u32x4 v = bst[d][nt][0];
                    if (!ok) v = ct_u32x4{0u, 0u, 0u, 0u};
                    bf[nt] = __builtin_bit_cast(ct_s16x8, v);
                }
#pragma unroll
                for (int sy = 0; sy < K; ++sy) {
                    // thin row y - (sy - P) is held row (y - ys) + K - 1 - sy
                    const unsigned int* q = reinterpret_cast<const unsigned int*>(abase + (rrel + K - 1 - sy) * g.Ct * g.trow);
                    unsigned int dw[5];
#pragma unroll
                    for (int j = 0; j < 5; ++j) dw[j] = q[j];
                    ct_u32x4 pk;
#pragma unroll
                    for (int j = 0; j < 4; ++j) pk[j] = mvalid ? __builtin_amdgcn_alignbit(dw[j + 1], dw[j], sh) : 0u;
                    const ct_s16x8 af = __builtin_bit_cast(ct_s16x8, pk);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[sy][nt] = ct_mma16<T>(af, bf[nt], acc[sy][nt]);
                }
            } else {
                float bv[NT][8];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int j = 0; j < 8; ++j) bv[nt][j] = ok ? __uint_as_float(bst[d][nt][j >> 2][j & 3]) : 0.f;
#pragma unroll
                for (int sy = 0; sy < K; ++sy) {
                    const float* q = reinterpret_cast<const float*>(abase + (rrel + K - 1 - sy) * g.Ct * g.trow);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float av = mvalid ? q[j] : 0.f;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[sy][nt] = ct_mma32(av, bv[nt][j], acc[sy][nt]);
                    }
                }
            }
        }
    }

    // ---- the waves' tiles -> one slab [sy][nt][m = 4 kg + i][c = l16], added in wave order ---------------------------
    __syncthreads();
    float* slab = reinterpret_cast<float*>(smem);   // the thin image is not needed any more
    for (int w = 0; w < CT_WG_THREADS / 64; ++w) {
        if (wave == w) {
#pragma unroll
            for (int sy = 0; sy < K; ++sy)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float* p = slab + ((sy * NT + nt) * 16 + 4 * kg + i) * 16 + l16;
                        *p = (w == 0 ? 0.f : *p) + acc[sy][nt][i];
                    }
        }
        __syncthreads();
    }
    float* dst = slabs + (long long)blockIdx.x * (K * NT * 256);
    for (int i = tid; i < K * NT * 256; i += CT_WG_THREADS) dst[i] = slab[i];
}

// dw[co][ci][ky][kx] = sum over the blocks' slabs.  role 0 (head: thin = dY, wide = X): co = t, ci = c, (ky, kx) = (sy, sx);
// role 1 (stem: thin = X, wide = dY): ci = t, co = c, (ky, kx) = (k - 1 - sy, k - 1 - sx).
// A block folds 16 consecutive slab elements: thread (element tid & 15, group tid >> 4) adds the slabs group, group + 16, ...
// in that order with all its requests in flight, then the 16 groups are added in group order (fixed order: deterministic).
template <int K>
__global__ __launch_bounds__(256) void ct_wg_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw, int nslabs,
                                                           int Ct, int Cw, int NT, int role) {
    __shared__ float red[256];
    const int el = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;                 // slab element ((sy * NT + nt) * 16 + m) * 16 + c16
    const long long stride = (long long)K * NT * 256;
    float s = 0.f;
    for (int b0 = grp; b0 < nslabs; b0 += 16 * 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = slabs[(long long)min(b0 + 16 * j, nslabs - 1) * stride + e];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += b0 + 16 * j < nslabs ? v[j] : 0.f;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (grp == 0) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += red[16 * j + el];
        const int c16 = e & 15, m = (e >> 4) & 15, nt = (e >> 8) % NT, sy = (e >> 8) / NT;
        if (m < Ct * K) {
            const int tch = m / K, sx = m % K, c = 16 * nt + c16;
            const int ky = role ? K - 1 - sy : sy, kx = role ? K - 1 - sx : sx;
            const int co = role ? c : tch, ci = role ? tch : c, Cin = role ? Ct : Cw;
            dw[((co * Cin + ci) * K + ky) * K + kx] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
static int ct_num_cus() {
    static int n = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t p;
            if (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) cus = p.multiProcessorCount;
        }
        (void)hipGetLastError();
        return cus;
    }();
    return n;
}

template <typename T> static CtGeom ct_geom(int64_t N, int64_t Ct, int64_t Cw, int64_t H, int64_t W) {
    CtGeom g{};
    g.N = (int)N; g.H = (int)H; g.W = (int)W; g.Ct = (int)Ct; g.Cw = (int)Cw;
    g.strips = (int)cdiv(W, CtElem<T>::CW);
    // one workgroup per CU: as few row segments as that allows (every segment re-reads 2 * (k / 2) halo rows)
    const int64_t units = N * g.strips;
    int64_t segs = cdiv(ct_num_cus(), units);
    int64_t RS = cdiv(H, segs < 1 ? 1 : segs);
    if (RS < 8) RS = H < 8 ? H : 8;
    g.RS = (int)RS;
    g.segs = (int)cdiv(H, RS);
    return g;
}

bool conv_thin_enabled() {
    static const bool on = [] {
        const char* e = getenv("OFASR_CONV_THIN");   // A/B switch: 0 keeps the 3-channel convs on the implicit-GEMM kernels
        return !(e && e[0] == '0');
    }();
    return on;
}

template <typename T> static CtGeom ct_geom_in(int64_t N, int64_t Ct, int64_t Cw, int64_t H, int64_t W) {
    CtGeom g = ct_geom<T>(N, Ct, Cw, H, W);
    if (g.RS > CT_IN_MAXRS) {       // the thin rows of a whole segment live in LDS
        g.RS = CT_IN_MAXRS;
        g.segs = (int)cdiv(H, g.RS);
    }
    return g;
}

bool conv_thin_in_supported(int64_t Ct, int64_t Cw, int K, int64_t W, int dtype, const void* x, const void* y) {
    if (!conv_thin_enabled() || !(K == 3 || K == 5) || Ct * K > 16 || Ct > 4 || !(Cw == 32 || Cw == 64)) return false;
    const int epc = dtype == OFASR_F32 ? 4 : 8;
    if (W % epc != 0) return false;
    return ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
}

int conv_thin_in_units(int64_t N, int64_t Ct, int64_t Cw, int64_t H, int64_t W, int dtype) {
    const CtGeom g = dtype == OFASR_F32 ? ct_geom_in<float>(N, Ct, Cw, H, W) : ct_geom_in<bf16_t>(N, Ct, Cw, H, W);
    return g.N * g.segs * g.strips;
}

template <typename T, int K, int NM>
static void ct_in_launch(const void* x, CtW Wt, void* y, const CtGeom& g, StatOut so, hipStream_t st) {
    const size_t lds = CtIn<T>::lds_bytes(K, g.Ct, g.Cw, g.RS);
    static bool attr = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ct_in_kernel<T, K, NM>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        return true;
    }();
    (void)attr;
    OFASR_LAUNCH((ct_in_kernel<T, K, NM>), dim3((unsigned)(g.N * g.segs * g.strips)), dim3(CT_IN_THREADS), lds, st, (const T*)x, Wt,
                 (T*)y, g, so);
}

// out[c] = sum_t W(c, t) * In[t]: forward of a conv with Cin = Ct (dgrad = 0, w is [Cw][Ct][K][K]) or input gradient of a
// conv with Cout = Ct (dgrad = 1, w is [Ct][Cw][K][K], taps mirrored)
int conv_thin_in(const void* x, const float* w, void* y, int64_t N, int64_t Ct, int64_t Cw, int64_t H, int64_t W, int K,
                 int dtype, int dgrad, StatOut so, void* stream) {
    hipStream_t st = as_stream(stream);
    CtW Wt{w, dgrad ? (int)(Cw * K * K) : K * K, dgrad ? K * K : (int)(Ct * K * K), dgrad ? 1 : 0};
    const double es = dtype == OFASR_F32 ? 4.0 : 2.0;
    prof_note(es * (double)N * (double)H * (double)W * (double)(Ct + Cw), 2.0 * (double)N * (double)H * (double)W * (double)Ct * (double)Cw * K * K);
#define OFASR_CTI(TT, KK, NMM)                                            \
    {                                                                     \
        const CtGeom g = ct_geom_in<TT>(N, Ct, Cw, H, W);                 \
        ct_in_launch<TT, KK, NMM>(x, Wt, y, g, so, st);                   \
    }
#define OFASR_CTI_T(TT)                                                   \
    if (K == 5) {                                                         \
        if (Cw == 64) OFASR_CTI(TT, 5, 4) else OFASR_CTI(TT, 5, 2)        \
    } else {                                                              \
        if (Cw == 64) OFASR_CTI(TT, 3, 4) else OFASR_CTI(TT, 3, 2)        \
    }
    if (dtype == OFASR_BF16) { OFASR_CTI_T(bf16_t) }
    else if (dtype == OFASR_F16) { OFASR_CTI_T(f16_t) }
    else { OFASR_CTI_T(float) }
#undef OFASR_CTI_T
#undef OFASR_CTI
    return check_launch("conv_thin_in");
}

static CtWgGeom ct_wg_geom(int64_t N, int64_t Ct, int64_t Cw, int64_t H, int64_t W, int es) {
    CtWgGeom g{};
    g.N = (int)N; g.H = (int)H; g.W = (int)W; g.Ct = (int)Ct; g.Cw = (int)Cw;
    g.nchunks = (int)cdiv(W, 32);
    // bytes per (row, channel): columns -8 .. 32 * nchunks + 8 + 2 (the funnel shift reads one dword further)
    g.trow = (int)((32 * g.nchunks + 24) * es + 15) / 16 * 16;
    int64_t segs = cdiv(ct_num_cus(), N);
    int64_t RS = cdiv(H, segs < 1 ? 1 : segs);
    // a block's fixed costs (thin image, the waves' fold, its 20 KB slab) want >= 8 items per wave
    const int64_t min_rows = cdiv(64, g.nchunks);
    if (RS < min_rows) RS = H < min_rows ? H : min_rows;
    while (RS > 4 && (RS + 4) * Ct * g.trow > 96 * 1024) RS = (RS + 1) / 2;
    g.RS = (int)RS;
    g.segs = (int)cdiv(H, RS);
    return g;
}

bool conv_thin_wgrad_supported(int64_t Ct, int64_t Cw, int K, int64_t H, int64_t W, int dtype, const void* a, const void* b) {
    if (!conv_thin_enabled() || !(K == 3 || K == 5) || Ct * K > 16 || Ct > 4 || !(Cw == 32 || Cw == 64)) return false;
    // a lane's wide fragment is 8 consecutive pixels of a row in either dtype: they must be all inside or all outside it
    if (W % 8 != 0 || W > 1024) return false;
    return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
}

size_t conv_thin_wgrad_workspace(int64_t N, int64_t Ct, int64_t Cw, int64_t H, int64_t W, int K, int dtype) {
    const CtWgGeom g = ct_wg_geom(N, Ct, Cw, H, W, dtype == OFASR_F32 ? 4 : 2);
    return (size_t)g.N * g.segs * K * (Cw / 16) * 256 * sizeof(float);
}

template <typename T, int K, int NT>
static void ct_wg_launch(const void* thin, const void* wide, float* slabs, const CtWgGeom& g, hipStream_t st) {
    size_t lds = (size_t)(g.RS + K - 1) * g.Ct * g.trow;
    const size_t slab = (size_t)K * NT * 256 * sizeof(float);
    if (lds < slab) lds = slab;
    static bool attr = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ct_wg_kernel<T, K, NT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        return true;
    }();
    (void)attr;
    OFASR_LAUNCH((ct_wg_kernel<T, K, NT>), dim3((unsigned)(g.N * g.segs)), dim3(CT_WG_THREADS), lds, st, (const T*)thin,
                 (const T*)wide, slabs, g);
}

// dw [Cout][Cin][K][K] of a conv with a thin side.  role 0: Cout thin (thin = dy, wide = x); role 1: Cin thin (thin = x, wide = dy)
int conv_thin_wgrad(const void* dy, const void* x, float* dw, int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W, int K,
                    int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    const int role = Cin < Cout ? 1 : 0;
    const int64_t Ct = role ? Cin : Cout, Cw = role ? Cout : Cin;
    const void* thin = role ? x : dy;
    const void* wide = role ? dy : x;
    const size_t need = conv_thin_wgrad_workspace(N, Ct, Cw, H, W, K, dtype);
    OFASR_REQUIRE(workspace && workspace_bytes >= need, OFASR_ERR_WORKSPACE, "conv_thin_wgrad: workspace %zu B < required %zu B",
                  workspace_bytes, need);
    hipStream_t st = as_stream(stream);
    const CtWgGeom g = ct_wg_geom(N, Ct, Cw, H, W, dtype == OFASR_F32 ? 4 : 2);
    const double es = dtype == OFASR_F32 ? 4.0 : 2.0;
    prof_note(es * (double)N * (double)H * (double)W * (double)(Ct + Cw), 2.0 * (double)N * (double)H * (double)W * (double)Ct * (double)Cw * K * K);
#define OFASR_CTW(TT, KK, NTT) ct_wg_launch<TT, KK, NTT>(thin, wide, (float*)workspace, g, st);
#define OFASR_CTW_T(TT)                                                   \
    if (K == 5) {                                                         \
        if (Cw == 64) OFASR_CTW(TT, 5, 4) else OFASR_CTW(TT, 5, 2)        \
    } else {                                                              \
        if (Cw == 64) OFASR_CTW(TT, 3, 4) else OFASR_CTW(TT, 3, 2)        \
    }
    if (dtype == OFASR_BF16) { OFASR_CTW_T(bf16_t) }
    else if (dtype == OFASR_F16) { OFASR_CTW_T(f16_t) }
    else { OFASR_CTW_T(float) }
#undef OFASR_CTW_T
#undef OFASR_CTW
    int rc = check_launch("conv_thin_wgrad");
    if (rc) return rc;
    const unsigned rblocks = (unsigned)(K * (Cw / 16) * 256 / 16);
    if (K == 5)
        OFASR_LAUNCH(ct_wg_reduce_kernel<5>, dim3(rblocks), dim3(256), 0, st, (const float*)workspace, dw, g.N * g.segs, (int)Ct,
                     (int)Cw, (int)(Cw / 16), role);
    else
        OFASR_LAUNCH(ct_wg_reduce_kernel<3>, dim3(rblocks), dim3(256), 0, st, (const float*)workspace, dw, g.N * g.segs, (int)Ct,
                     (int)Cw, (int)(Cw / 16), role);
    return check_launch("conv_thin_wgrad_reduce");
}

bool conv_thin_out_supported(int64_t Ct, int64_t Cw, int K, int64_t W, int dtype, const void* x, const void* y) {
    if (!conv_thin_enabled() || !(K == 3 || K == 5) || Ct * K > 16 || Ct > 4 || !(Cw == 32 || Cw == 64)) return false;
    const int epc = dtype == OFASR_F32 ? 4 : 8;
    if (W % epc != 0) return false;
    return ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
}

int conv_thin_out_units(int64_t N, int64_t Ct, int64_t Cw, int64_t H, int64_t W, int dtype) {
    const CtGeom g = dtype == OFASR_F32 ? ct_geom<float>(N, Ct, Cw, H, W) : ct_geom<bf16_t>(N, Ct, Cw, H, W);
    return g.N * g.segs * g.strips;
}

template <typename T, int K, int NC>
static void ct_out_launch(const void* x, CtW Wt, void* y, const CtGeom& g, StatOut so, hipStream_t st) {
    const size_t lds = CtOut<T>::lds_bytes(K, NC);
    static bool attr = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ct_out_kernel<T, K, NC>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        return true;
    }();
    (void)attr;
    OFASR_LAUNCH((ct_out_kernel<T, K, NC>), dim3((unsigned)(g.N * g.segs * g.strips)), dim3(CtOut<T>::THREADS), lds, st, (const T*)x, Wt,
                 (T*)y, g, so);
}

// out[t] = sum_c W(t, c) * X[c]: forward of a conv with Cout = Ct (flip = 0, w is [Ct][Cw][K][K]) or input gradient of a
// conv with Cin = Ct (flip = 1, w is [Cw][Ct][K][K])
int conv_thin_out(const void* x, const float* w, void* y, int64_t N, int64_t Ct, int64_t Cw, int64_t H, int64_t W, int K,
                  int dtype, int dgrad, StatOut so, void* stream) {
    hipStream_t st = as_stream(stream);
    CtW Wt{w, dgrad ? K * K : (int)(Cw * K * K), dgrad ? (int)(Ct * K * K) : K * K, dgrad ? 1 : 0};
    const double es = dtype == OFASR_F32 ? 4.0 : 2.0;
    prof_note(es * (double)N * (double)H * (double)W * (double)(Ct + Cw), 2.0 * (double)N * (double)H * (double)W * (double)Ct * (double)Cw * K * K);
#define OFASR_CTO(TT, KK, NCC)                                            \
    {                                                                     \
        const CtGeom g = ct_geom<TT>(N, Ct, Cw, H, W);                    \
        ct_out_launch<TT, KK, NCC>(x, Wt, y, g, so, st);                  \
    }
#define OFASR_CTO_T(TT)                                                   \
    if (K == 5) {                                                         \
        if (Cw == 64) OFASR_CTO(TT, 5, 2) else OFASR_CTO(TT, 5, 1)        \
    } else {                                                              \
        if (Cw == 64) OFASR_CTO(TT, 3, 2) else OFASR_CTO(TT, 3, 1)        \
    }
    if (dtype == OFASR_BF16) { OFASR_CTO_T(bf16_t) }
    else if (dtype == OFASR_F16) { OFASR_CTO_T(f16_t) }
    else { OFASR_CTO_T(float) }
#undef OFASR_CTO_T
#undef OFASR_CTO
    return check_launch("conv_thin_out");
}

}  // namespace ofasr
