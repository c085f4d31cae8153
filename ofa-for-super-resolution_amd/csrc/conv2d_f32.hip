// conv2d_f32.hip -- the static ConvLayers' dense KxK convolution (K in {3,5}, stride 1, zero padding K/2) with fp32
// activations, on the fp32-input matrix instruction v_mfma_f32_32x32x2_f32: forward, input gradient, weight gradient.
//
// Reference: nn.Conv2d in ConvLayer (ofa/layers.py:131-151; ofa_mbs4.py:65,105,120,123).  fp32 is the reference's own
// arithmetic: this path serves the fp32 parity nets, the `fp32` leg of bench.py and fp32
// evaluation (eval_ofa_net_sr.py) -- any H, W, no alignment requirement -- so that no layer of the SR nets leaves the
// library.  The MFMA is an exact fp32 fma chain (one rounding per product, k-ordered), at the fp32 vector rate
// (157 TFLOP/s peak = 1/16 of the bf16 matrix rate): the 16-bit kernels of conv2d.hip remain the fast path.
//
//   Y[n, m, h, w] = sum_{ty,tx} sum_k Wv(m, k, ty, tx) * X[n, k, h+ty-P, w+tx-P]
//   forward: (m, k) = (co, ci), Wv = w[m][k][ty][tx];   input gradient: (m, k) = (ci, co), Wv = w[k][m][K-1-ty][K-1-tx]
//
// forward / dgrad kernel: block = 4 waves = 4 output rows x 32 columns x 64 output channels; wave w owns row w:
// D[row = channel][col = pixel].  Per 16-channel input chunk the (4+K-1) x (32+K-1) window sits in LDS ([ci][row][col],
// fp32) and, per kernel row ty, the weight slab [tx][16 k][64 m] (copied with 16-byte loads from an image laid out in
// that order by conv_f32_prep_kernel): an A operand is one ds_read_b32 of 32 consecutive channels, a B operand one
// ds_read_b32 of 32 consecutive pixels (both conflict-free), K*8*2 MFMAs per wave and (chunk, ty).
// weight gradient: rows = co, columns = ci, reduction over pixels (2 per MFMA); a block of K waves owns a 32 x 32
// (co, ci) tile and a range of 4 x 32 pixel tiles, wave ty keeps its K accumulators (tx) in registers over the whole
// range; per-range partial slabs are summed in a fixed order (deterministic, no atomics).
#include "ofasr_common.h"

namespace ofasr {

typedef __attribute__((ext_vector_type(16))) float cf_f32x16;

constexpr int CF_TH = 4, CF_TW = 32, CF_KC = 16, CF_MB = 64;
constexpr int CF_WG_RB = 1;    // weight gradient: 32-row output-channel blocks per wave
constexpr int CF_WG_TH = 4;    // weight gradient: rows of the pixel tile staged per barrier round
// weight gradient: waves per kernel row.  A block of K = 5 waves puts two of them on one SIMD and one on each of the
// others, so the matrix pipes of three SIMDs idle half the time (5/8 of the fp32 matrix rate at best, whatever the
// number of blocks per CU -- every block maps its waves the same way).  With two waves per kernel row, each on half of
// the tile's pixel pairs, a block is 10 waves = 3 + 3 + 2 + 2 (10/12); the two halves write separate partial slabs.
constexpr int CF_WG_HALVES = 2;

// weight image [kc][ty][tx][kk = 16][m = Mpad] fp32 (zeros beyond the slice)
__global__ void __launch_bounds__(256) conv_f32_prep_kernel(const float* __restrict__ w, float* __restrict__ wimg, int Cin,
                                                            int KS, int dgrad, int M, int Kdim, int Mpad, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int m = (int)(idx % Mpad);
    long long t = idx / Mpad;
    const int kk = (int)(t % CF_KC);
    t /= CF_KC;
    const int tx = (int)(t % KS);
    t /= KS;
    const int ty = (int)(t % KS);
    const int kc = (int)(t / KS);
    const int k = kc * CF_KC + kk;
    float v = 0.f;
    if (m < M && k < Kdim) {
        const int taps = KS * KS, tap = ty * KS + tx;
        v = dgrad ? w[((long long)k * Cin + m) * taps + (taps - 1 - tap)] : w[((long long)m * Cin + k) * taps + tap];
    }
    wimg[idx] = v;
}

template <int KS>
__global__ void __launch_bounds__(256) conv_f32_kernel(const float* __restrict__ x, const float* __restrict__ wimg,
                                                       float* __restrict__ y, int Kdim, int M, int Mpad, int H, int W,
                                                       int tiles_x, int nkc) {
    constexpr int P = KS / 2;
    constexpr int RH = CF_TH + KS - 1, RW = CF_TW + KS - 1;
    constexpr int XPL = RH * RW + 1;                        // plane pitch (odd: the two k-lanes of a fragment differ by a bank)
    __shared__ float Xs[CF_KC * XPL];
    __shared__ __attribute__((aligned(16))) float Ws[KS * CF_KC * CF_MB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, kk = lane >> 5;
    const int tile = blockIdx.x, n = blockIdx.y, slab = blockIdx.z;
    const int h0 = (tile / tiles_x) * CF_TH, w0 = (tile % tiles_x) * CF_TW;
    const long long plane = (long long)H * W;
    const float* xn = x + (long long)n * Kdim * plane;
    cf_f32x16 acc[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[rb][i] = 0.f;

    for (int kc = 0; kc < nkc; ++kc) {
        __syncthreads();   // the previous chunk's readers are done with Xs / Ws
        for (int e = tid; e < CF_KC * RH * RW; e += 256) {
            const int ci = e / (RH * RW), r = (e - ci * (RH * RW)) / RW, col = e - ci * (RH * RW) - r * RW;
            const int k = kc * CF_KC + ci, gh = h0 - P + r, gw = w0 - P + col;
            float v = 0.f;
            if (k < Kdim && gh >= 0 && gh < H && gw >= 0 && gw < W) v = xn[(long long)k * plane + (long long)gh * W + gw];
            Xs[ci * XPL + r * RW + col] = v;
        }
        for (int ty = 0; ty < KS; ++ty) {
            if (ty) __syncthreads();   // the previous kernel row's readers are done with Ws
            {
                const float4* src = reinterpret_cast<const float4*>(wimg + ((long long)(kc * KS + ty) * KS) * CF_KC * Mpad);
                for (int q = tid; q < KS * CF_KC * (CF_MB / 4); q += 256) {
                    const int m4 = q % (CF_MB / 4), rest = q / (CF_MB / 4);   // rest = tx*16 + kk
                    reinterpret_cast<float4*>(Ws)[q] = src[(long long)rest * (Mpad / 4) + slab * (CF_MB / 4) + m4];
                }
            }
            __syncthreads();
#pragma unroll
            for (int tx = 0; tx < KS; ++tx) {
                const float* wt = Ws + tx * (CF_KC * CF_MB) + kk * CF_MB + c;
                const float* xt = Xs + kk * XPL + (wave + ty) * RW + c + tx;
#pragma unroll
                for (int s = 0; s < CF_KC / 2; ++s) {
                    const float b = xt[2 * s * XPL];
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wt[2 * s * CF_MB], b, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wt[2 * s * CF_MB + 32], b, acc[1], 0, 0, 0);
                }
            }
        }
    }
    const int oy = h0 + wave, ox = w0 + c;
    if (oy < H && ox < W) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int m = slab * CF_MB + 32 * rb + (reg & 3) + 8 * (reg >> 2) + 4 * kk;
                if (m < M) y[(((long long)n * M + m) * H + oy) * W + ox] = acc[rb][reg];
            }
    }
}

// ---- weight gradient: partial[split][co][ci][ty][tx] over the block's pixel tiles
template <int KS>
__global__ void __launch_bounds__(64 * KS * CF_WG_HALVES) conv_f32_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                 float* __restrict__ part, int Cin, int Cout, int H, int W,
                                                                 int tiles_x, int tiles_y, int ntiles, int nsplit) {
    constexpr int P = KS / 2, THREADS = 64 * KS * CF_WG_HALVES;
    constexpr int RH = CF_WG_TH + KS - 1, RW = CF_TW + KS - 1;
    constexpr int GPL = CF_WG_TH * CF_TW + 1;                  // dY plane pitch: bank = (co + pixel) % 32 -> conflict-free A reads
    constexpr int XPL = RH * RW + 1;
    // output channels per block = CF_WG_RB 32-row MFMA blocks per wave (2 measured slower: 57 against 67 TFLOP/s on
    // 64 -> 256 @128x128 -- 160 accumulator registers on top of the staging registers; 2-row pixel tiles, CF_WG_TH = 2,
    // for twice the blocks per CU: 63)
    constexpr int NRB = CF_WG_RB, NCO = 32 * NRB;
    __shared__ float Gs[NCO * GPL];
    __shared__ float Xs[32 * XPL];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ty = wv % KS, half = wv / KS;                    // this wave's kernel row and its share of the pixel pairs
    const int c = lane & 31, kk = lane >> 5;
    // Block order: the (co, ci) groups of one pixel split read the same dY / X tiles, so they should run at the same time
    // on the same XCD (each XCD has its own L2; consecutive workgroup ids go round the 8 XCDs).  id = 8 * (G * q + group)
    // + xcd with split = 8 q + xcd: the G blocks of a split follow each other on one XCD and the tiles are fetched from
    // HBM about once instead of once per group (measured 49 -> see DESIGN.md TFLOP/s on 64->256 @128x128).
    const int gco = (Cout + NCO - 1) / NCO, G = gco * ((Cin + 31) / 32);
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int group = j % G, split = (j / G) * 8 + xcd;
    if (split >= nsplit) return;     // workgroup-uniform: the grid is padded to whole rounds of 8 splits
    const int co0 = (group % gco) * NCO, ci0 = (group / gco) * 32;
    const long long plane = (long long)H * W;
    cf_f32x16 acc[NRB][KS];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int tx = 0; tx < KS; ++tx)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rb][tx][i] = 0.f;
    const int t0 = (int)((long long)split * ntiles / nsplit), t1 = (int)((long long)(split + 1) * ntiles / nsplit);
    auto compute = [&]() {
        // pixel pair s = (row r, columns 2q, 2q+1); lane (c, kk): A = dY[co = c][pixel 2s + kk], B = X[ci = c][same pixel + tap]
        constexpr int NS = CF_WG_TH * CF_TW / 2 / CF_WG_HALVES;
#pragma unroll 2
        for (int s = half * NS; s < (half + 1) * NS; ++s) {
            const int p = 2 * s + kk, r = p / CF_TW, col = p % CF_TW;
            float a[NRB];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) a[rb] = Gs[(32 * rb + c) * GPL + p];
            const float* xb = Xs + c * XPL + (r + ty) * RW + col;
#pragma unroll
            for (int tx = 0; tx < KS; ++tx) {
                const float b = xb[tx];
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) acc[rb][tx] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rb], b, acc[rb][tx], 0, 0, 0);
            }
        }
    };
    // Fast staging (even window start, W % 4 == 0, aligned bases): the tile's dY quads and X pairs are requested as
    // straight-line vector loads from clamped addresses into registers -- all in flight together, and for tile t+1 while
    // tile t is in the matrix cores -- and zeroed afterwards where they lie outside the tensor.  (Element-wise guarded
    // loads in a loop cost one memory round trip each: the kernel then spends its time staging, 49 TFLOP/s.)
    const bool fast = (P % 2 == 0) && (W % 4 == 0) &&
                      ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x)) & 15) == 0;
    if (fast) {
        constexpr int NQG = NCO * CF_WG_TH * CF_TW / 4, NG = (NQG + THREADS - 1) / THREADS;
        constexpr int XW2 = RW / 2, NQX = 32 * RH * XW2, NX = (NQX + THREADS - 1) / THREADS;
        float4 g[NG];
        float2 xv[NX];
        auto load_tile = [&](int t) {
            const int n = t / (tiles_x * tiles_y), rem = t - n * (tiles_x * tiles_y);
            const int h0 = (rem / tiles_x) * CF_WG_TH, w0 = (rem % tiles_x) * CF_TW;
#pragma unroll
            for (int it = 0; it < NG; ++it) {
                const int q0 = tid + it * THREADS, q = q0 < NQG ? q0 : NQG - 1;
                const int co = q / (CF_WG_TH * CF_TW / 4), pq = q - co * (CF_WG_TH * CF_TW / 4);
                const int gh = h0 + pq / (CF_TW / 4), gw = w0 + 4 * (pq % (CF_TW / 4));
                const int coc = co0 + co < Cout ? co0 + co : Cout - 1, ghc = gh < H ? gh : H - 1, gwc = gw + 4 <= W ? gw : W - 4;
                const float4 v = *reinterpret_cast<const float4*>(dy + ((long long)n * Cout + coc) * plane + (long long)ghc * W + gwc);
                const bool ok = co0 + co < Cout && gh < H && gw < W;
                g[it] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int it = 0; it < NX; ++it) {
                const int q0 = tid + it * THREADS, q = q0 < NQX ? q0 : NQX - 1;
                const int ci = q / (RH * XW2), r2 = q - ci * (RH * XW2);
                const int r = r2 / XW2, c2 = r2 - r * XW2;
                const int gh = h0 - P + r, gw = w0 - P + 2 * c2;
                const int cic = ci0 + ci < Cin ? ci0 + ci : Cin - 1;
                const int ghc = gh < 0 ? 0 : (gh < H ? gh : H - 1), gwc = gw < 0 ? 0 : (gw + 2 <= W ? gw : W - 2);
                const float2 v = *reinterpret_cast<const float2*>(x + ((long long)n * Cin + cic) * plane + (long long)ghc * W + gwc);
                const bool ok = ci0 + ci < Cin && gh >= 0 && gh < H && gw >= 0 && gw < W;
                xv[it] = ok ? v : make_float2(0.f, 0.f);
            }
        };
        auto store_tile = [&]() {
#pragma unroll
            for (int it = 0; it < NG; ++it) {
                const int q = tid + it * THREADS;
                if (q < NQG) {
                    const int co = q / (CF_WG_TH * CF_TW / 4), pq = q - co * (CF_WG_TH * CF_TW / 4);
                    float* d = Gs + co * GPL + 4 * pq;
                    d[0] = g[it].x; d[1] = g[it].y; d[2] = g[it].z; d[3] = g[it].w;
                }
            }
#pragma unroll
            for (int it = 0; it < NX; ++it) {
                const int q = tid + it * THREADS;
                if (q < NQX) {
                    const int ci = q / (RH * XW2), r2 = q - ci * (RH * XW2);
                    float* d = Xs + ci * XPL + 2 * r2;     // r * RW + 2 c2 = 2 (r * XW2 + c2)
                    d[0] = xv[it].x; d[1] = xv[it].y;
                }
            }
        };
        if (t0 < t1) load_tile(t0);
        for (int t = t0; t < t1; ++t) {
            __syncthreads();
            store_tile();
            __syncthreads();
            if (t + 1 < t1) load_tile(t + 1);
            compute();
        }
    } else {
    for (int t = t0; t < t1; ++t) {
        const int n = t / (tiles_x * tiles_y), rem = t - n * (tiles_x * tiles_y);
        const int h0 = (rem / tiles_x) * CF_WG_TH, w0 = (rem % tiles_x) * CF_TW;
        __syncthreads();
        for (int e = tid; e < NCO * CF_WG_TH * CF_TW; e += THREADS) {
            const int co = e / (CF_WG_TH * CF_TW), p = e - co * (CF_WG_TH * CF_TW);
            const int gh = h0 + p / CF_TW, gw = w0 + p % CF_TW;
            float v = 0.f;
            if (co0 + co < Cout && gh < H && gw < W) v = dy[((long long)n * Cout + co0 + co) * plane + (long long)gh * W + gw];
            Gs[co * GPL + p] = v;
        }
        for (int e = tid; e < 32 * RH * RW; e += THREADS) {
            const int ci = e / (RH * RW), r = (e - ci * (RH * RW)) / RW, col = e - ci * (RH * RW) - r * RW;
            const int gh = h0 - P + r, gw = w0 - P + col;
            float v = 0.f;
            if (ci0 + ci < Cin && gh >= 0 && gh < H && gw >= 0 && gw < W)
                v = x[((long long)n * Cin + ci0 + ci) * plane + (long long)gh * W + gw];
            Xs[ci * XPL + r * RW + col] = v;
        }
        __syncthreads();
        compute();
    }
    }
    // D[row = co][col = ci]
    float* dst = part + ((long long)split * CF_WG_HALVES + half) * Cout * Cin * KS * KS;
    const int ci = ci0 + c;
    if (ci < Cin) {
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int tx = 0; tx < KS; ++tx)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int co = co0 + 32 * rb + (reg & 3) + 8 * (reg >> 2) + 4 * kk;
                    if (co < Cout) dst[(((long long)co * Cin + ci) * KS + ty) * KS + tx] = acc[rb][tx][reg];
                }
    }
}

// 16 outputs x 16 slab lanes per block: lane z adds the slabs z, z + 16, ... in that order (8 requests in flight), the 16
// lanes are then added in lane order: a fixed order, so the result is deterministic
__global__ void __launch_bounds__(256) conv_f32_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                    long long total, int nslabs) {
    __shared__ float red[256];
    const int el = threadIdx.x & 15, zl = threadIdx.x >> 4;
    const long long idx = (long long)blockIdx.x * 16 + el;
    const long long idc = idx < total ? idx : total - 1;
    float s = 0.f;
    for (int z0 = zl; z0 < nslabs; z0 += 16 * 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int z = z0 + 16 * j < nslabs ? z0 + 16 * j : nslabs - 1;
            v[j] = part[(long long)z * total + idc];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) s += z0 + 16 * j < nslabs ? v[j] : 0.f;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (zl == 0 && idx < total) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += red[16 * j + el];
        dw[idx] = t;
    }
}

static int cf_mpad(int64_t M) { return (int)(cdiv(M, CF_MB) * CF_MB); }
static int cf_nsplit(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W) {
    const int64_t ntiles = N * cdiv(H, CF_WG_TH) * cdiv(W, CF_TW);
    const int64_t groups = cdiv(Cout, 32 * CF_WG_RB) * cdiv(Cin, 32);
    int64_t want = 1024 / (groups > 0 ? groups : 1);   // ~4 blocks per CU
    if (want > ntiles) want = ntiles;
    if (want < 1) want = 1;
    return (int)want;
}

}  // namespace ofasr

using namespace ofasr;

OFASR_EXPORT size_t ofasr_conv2d_f32_workspace(int64_t Cin, int64_t Cout, int K, int dgrad) {
    if (Cin <= 0 || Cout <= 0 || !(K == 3 || K == 5)) return 0;
    const int64_t M = dgrad ? Cin : Cout, Kd = dgrad ? Cout : Cin;
    return (size_t)cdiv(Kd, CF_KC) * K * K * CF_KC * cf_mpad(M) * sizeof(float);
}

static int conv2d_f32_entry(const char* name, const void* x, const float* w, void* y, int64_t N, int64_t Cin, int64_t Cout,
                            int64_t H, int64_t W, int K, int dgrad, void* ws, size_t ws_bytes, void* stream) {
    OFASR_REQUIRE(x && w && y, OFASR_ERR_INVALID_ARG, "%s: null pointer", name);
    OFASR_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, OFASR_ERR_INVALID_ARG, "%s: bad shape", name);
    OFASR_REQUIRE(K == 3 || K == 5, OFASR_ERR_UNSUPPORTED, "%s: K=%d not in {3,5}", name, K);
    OFASR_REQUIRE(N <= 65535 && H * W <= (1LL << 31), OFASR_ERR_UNSUPPORTED, "%s: too large", name);
    const size_t need = ofasr_conv2d_f32_workspace(Cin, Cout, K, dgrad);
    OFASR_REQUIRE(ws && ws_bytes >= need && (reinterpret_cast<uintptr_t>(ws) & 15) == 0, OFASR_ERR_WORKSPACE,
                  "%s: workspace %zu B < required %zu B (or not 16-byte aligned)", name, ws_bytes, need);
    {
        // a 3-channel result (the head's forward, the stem's input gradient): csrc/conv_thin.hip
        const int64_t Ct = dgrad ? Cin : Cout, Cw = dgrad ? Cout : Cin;
        if (conv_thin_out_supported(Ct, Cw, K, W, OFASR_F32, x, y))
            return conv_thin_out(x, w, y, N, Ct, Cw, H, W, K, OFASR_F32, dgrad, StatOut{nullptr, 0}, stream);
        if (conv_thin_in_supported(Cw, Ct, K, W, OFASR_F32, x, y))
            return conv_thin_in(x, w, y, N, Cw, Ct, H, W, K, OFASR_F32, dgrad, StatOut{nullptr, 0}, stream);
    }
    const int M = (int)(dgrad ? Cin : Cout), Kdim = (int)(dgrad ? Cout : Cin), Mpad = cf_mpad(M);
    const int nkc = (int)cdiv(Kdim, CF_KC);
    hipStream_t st = as_stream(stream);
    const long long total = (long long)(need / sizeof(float));
    OFASR_LAUNCH(conv_f32_prep_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, st, w, (float*)ws, (int)Cin, K, dgrad,
                 M, Kdim, Mpad, total);
    int rc = check_launch(name);
    if (rc) return rc;
    const int tiles_x = (int)cdiv(W, CF_TW), tiles_y = (int)cdiv(H, CF_TH);
    dim3 grid((unsigned)(tiles_x * tiles_y), (unsigned)N, (unsigned)(Mpad / CF_MB));
    prof_note(4.0 * (double)N * (double)H * (double)W * (double)(Cin + Cout),
              2.0 * (double)N * (double)H * (double)W * (double)Cin * (double)Cout * K * K);
    if (K == 5)
        OFASR_LAUNCH(conv_f32_kernel<5>, grid, dim3(256), 0, st, (const float*)x, (const float*)ws, (float*)y, Kdim, M, Mpad,
                     (int)H, (int)W, tiles_x, nkc);
    else
        OFASR_LAUNCH(conv_f32_kernel<3>, grid, dim3(256), 0, st, (const float*)x, (const float*)ws, (float*)y, Kdim, M, Mpad,
                     (int)H, (int)W, tiles_x, nkc);
    return check_launch(name);
}

OFASR_EXPORT int ofasr_conv2d_f32_fwd(const void* x, const float* w, void* y, int64_t N, int64_t Cin, int64_t Cout, int64_t H,
                                      int64_t W, int K, void* workspace, size_t workspace_bytes, void* stream) {
    return conv2d_f32_entry("ofasr_conv2d_f32_fwd", x, w, y, N, Cin, Cout, H, W, K, 0, workspace, workspace_bytes, stream);
}

OFASR_EXPORT int ofasr_conv2d_f32_dgrad(const void* dy, const float* w, void* dx, int64_t N, int64_t Cin, int64_t Cout,
                                        int64_t H, int64_t W, int K, void* workspace, size_t workspace_bytes, void* stream) {
    return conv2d_f32_entry("ofasr_conv2d_f32_dgrad", dy, w, dx, N, Cin, Cout, H, W, K, 1, workspace, workspace_bytes, stream);
}

OFASR_EXPORT size_t ofasr_conv2d_f32_wgrad_workspace(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W, int K) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || !(K == 3 || K == 5)) return 0;
    size_t need = (size_t)cf_nsplit(N, Cin, Cout, H, W) * CF_WG_HALVES * Cout * Cin * K * K * sizeof(float);
    const int64_t Ct = Cin < Cout ? Cin : Cout, Cw = Cin < Cout ? Cout : Cin;
    if (conv_thin_wgrad_supported(Ct, Cw, K, H, W, OFASR_F32, nullptr, nullptr)) {   // the head / stem: csrc/conv_thin.hip
        const size_t thin = conv_thin_wgrad_workspace(N, Ct, Cw, H, W, K, OFASR_F32);
        need = thin > need ? thin : need;
    }
    return need;
}

OFASR_EXPORT int ofasr_conv2d_f32_wgrad(const void* dy, const void* x, float* dw, int64_t N, int64_t Cin, int64_t Cout,
                                        int64_t H, int64_t W, int K, void* workspace, size_t workspace_bytes, void* stream) {
    const char* name = "ofasr_conv2d_f32_wgrad";
    OFASR_REQUIRE(dy && x && dw, OFASR_ERR_INVALID_ARG, "%s: null pointer", name);
    OFASR_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, OFASR_ERR_INVALID_ARG, "%s: bad shape", name);
    OFASR_REQUIRE(K == 3 || K == 5, OFASR_ERR_UNSUPPORTED, "%s: K=%d not in {3,5}", name, K);
    OFASR_REQUIRE(N * H * W <= (1LL << 30), OFASR_ERR_UNSUPPORTED, "%s: too large", name);
    const size_t need = ofasr_conv2d_f32_wgrad_workspace(N, Cin, Cout, H, W, K);
    OFASR_REQUIRE(workspace && workspace_bytes >= need, OFASR_ERR_WORKSPACE, "%s: workspace %zu B < required %zu B", name,
                  workspace_bytes, need);
    {
        const int64_t Ct = Cin < Cout ? Cin : Cout, Cw = Cin < Cout ? Cout : Cin;
        if (conv_thin_wgrad_supported(Ct, Cw, K, H, W, OFASR_F32, dy, x))
            return conv_thin_wgrad(dy, x, dw, N, Cin, Cout, H, W, K, OFASR_F32, workspace, workspace_bytes, stream);
    }
    const int nsplit = cf_nsplit(N, Cin, Cout, H, W);
    const int tiles_x = (int)cdiv(W, CF_TW), tiles_y = (int)cdiv(H, CF_WG_TH);
    const int ntiles = (int)(N * tiles_x * tiles_y);
    hipStream_t st = as_stream(stream);
    const long long nblocks = cdiv(nsplit, 8) * 8 * cdiv(Cout, 32 * CF_WG_RB) * cdiv(Cin, 32);
    OFASR_REQUIRE(nblocks <= INT32_MAX, OFASR_ERR_UNSUPPORTED, "%s: too many blocks", name);
    dim3 grid((unsigned)nblocks);
    prof_note(4.0 * (double)N * (double)H * (double)W * (double)(Cin + Cout),
              2.0 * (double)N * (double)H * (double)W * (double)Cin * (double)Cout * K * K);
    if (K == 5)
        OFASR_LAUNCH(conv_f32_wgrad_kernel<5>, grid, dim3(64 * 5 * CF_WG_HALVES), 0, st, (const float*)dy, (const float*)x, (float*)workspace,
                     (int)Cin, (int)Cout, (int)H, (int)W, tiles_x, tiles_y, ntiles, nsplit);
    else
        OFASR_LAUNCH(conv_f32_wgrad_kernel<3>, grid, dim3(64 * 3 * CF_WG_HALVES), 0, st, (const float*)dy, (const float*)x, (float*)workspace,
                     (int)Cin, (int)Cout, (int)H, (int)W, tiles_x, tiles_y, ntiles, nsplit);
    int rc = check_launch(name);
    if (rc) return rc;
    const long long total = (long long)Cout * Cin * K * K;
    OFASR_LAUNCH(conv_f32_wgrad_reduce_kernel, dim3((unsigned)cdiv(total, 16)), dim3(256), 0, st, (const float*)workspace, dw,
                 total, nsplit * CF_WG_HALVES);
    return check_launch(name);
}
