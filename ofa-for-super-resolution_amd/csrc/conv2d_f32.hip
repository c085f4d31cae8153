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
// weight gradient: rows = co, columns = ci, reduction over pixels (2 per MFMA); a block of 2 K compute waves + 2 loader
// waves owns a 32 x 32 (co, ci) tile and a range of 4 x 32 pixel tiles; a compute wave keeps the K accumulators (tx) of
// its kernel row in registers over the whole range, the loader waves fill the other LDS buffer meanwhile; per-range
// partial slabs are summed in a fixed order (deterministic, no atomics).
#include "ofasr_common.h"

namespace ofasr {

typedef __attribute__((ext_vector_type(16))) float cf_f32x16;

constexpr int CF_TH = 4, CF_TW = 32, CF_KC = 16, CF_MB = 64;
constexpr int CF_WG_RB = 1;    // weight gradient: 32-row output-channel blocks per wave
constexpr int CF_WG_TH = 4;    // weight gradient: rows of the pixel tile staged per barrier round
// weight gradient: compute waves per kernel row (each kernel row's pixel pairs are shared by two waves, which write
// separate partial slabs; see the wave shares at the kernel)
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
__global__ void __launch_bounds__(256, 3) conv_f32_kernel(const float* __restrict__ x, const float* __restrict__ wimg,
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

    // X window staging, fast form (K = 5: window start w0 - 2, so the aligned quads [w0 - 4, w0 + 36) cover it; W % 4 == 0,
    // 16-byte aligned base): 5 quad requests per thread and chunk, all in flight together, branch-free (a quad outside
    // the tensor reads the image's first quad and is zeroed by a mask).  The element loop below is one guarded scalar
    // load per iteration, each waited for before the next: 18 serial round trips per chunk.
    constexpr int XQ = (RW + 4) / 4, NQ = CF_KC * RH * XQ, NQT = (NQ + 255) / 256;
    const bool fastx = KS == 5 && (W % 4 == 0) && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    for (int kc = 0; kc < nkc; ++kc) {
        __syncthreads();   // the previous chunk's readers are done with Xs / Ws
        if (fastx) {
            float4 q4[NQT];
            uint32_t okm[NQT];
#pragma unroll
            for (int it = 0; it < NQT; ++it) {
                const int idx = tid + 256 * it, ci = idx / (RH * XQ), rem = idx - ci * (RH * XQ), r = rem / XQ, jq = rem - r * XQ;
                const int k = kc * CF_KC + ci, gh = h0 - P + r, gw = w0 - 4 + 4 * jq;
                const bool ok = idx < NQ && k < Kdim && gh >= 0 && gh < H && gw >= 0 && gw < W;
                okm[it] = ok ? 0xffffffffu : 0u;
                q4[it] = *reinterpret_cast<const float4*>(xn + (ok ? (long long)k * plane + (long long)gh * W + gw : 0));
            }
#pragma unroll
            for (int it = 0; it < NQT; ++it) {
                const int idx = tid + 256 * it, ci = idx / (RH * XQ), rem = idx - ci * (RH * XQ), r = rem / XQ, jq = rem - r * XQ;
                if (idx < NQ) {
                    float* d = Xs + ci * XPL + r * RW + 4 * jq - 2;      // window columns 4 jq - 2 .. 4 jq + 1
                    const uint32_t m = okm[it];
                    if (jq > 0) {
                        d[0] = __uint_as_float(__float_as_uint(q4[it].x) & m);
                        d[1] = __uint_as_float(__float_as_uint(q4[it].y) & m);
                    }
                    if (jq < XQ - 1) {
                        d[2] = __uint_as_float(__float_as_uint(q4[it].z) & m);
                        d[3] = __uint_as_float(__float_as_uint(q4[it].w) & m);
                    }
                }
            }
        } else
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
                // the slab's KS quads per thread are requested together, then written (as a rolled loop every request was
                // waited for before the next: KS serial L2 round trips per kernel row)
                const float4* src = reinterpret_cast<const float4*>(wimg + ((long long)(kc * KS + ty) * KS) * CF_KC * Mpad);
                constexpr int NWQ = KS * CF_KC * (CF_MB / 4) / 256;
                static_assert(KS * CF_KC * (CF_MB / 4) % 256 == 0, "whole rounds of the block");
                float4 wq[NWQ];
#pragma unroll
                for (int it = 0; it < NWQ; ++it) {
                    const int q = tid + 256 * it;
                    const int m4 = q % (CF_MB / 4), rest = q / (CF_MB / 4);   // rest = tx*16 + kk
                    wq[it] = src[(long long)rest * (Mpad / 4) + slab * (CF_MB / 4) + m4];
                }
#pragma unroll
                for (int it = 0; it < NWQ; ++it) reinterpret_cast<float4*>(Ws)[tid + 256 * it] = wq[it];
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

// ---- weight gradient: partial[split][half][co][ci][ty][tx] over the block's pixel tiles
// A wave owns one kernel row ty (its K accumulators tx stay in registers over the block's whole pixel range) and a range
// of the tile's 8 half rows (a half row = 8 pixel pairs = 8 x K MFMAs).  The hardware puts wave i of a block on SIMD
// i % 4, so 2 K waves are 3 + 3 + 2 + 2 (K = 5) or 2 + 2 + 1 + 1 (K = 3) per SIMD: the shares below give every SIMD the
// same number of half rows (10 of 40, 6 of 24) instead of every wave -- each kernel row is still covered by exactly two
// waves, which write the two partial slabs `half` of their split.  entry = ty | hb << 4 | he << 8 | half << 12
__constant__ const unsigned short cf_wg_shares5[10] = {0x1850, 0x1852, 0x0500, 0x0502, 0x1851, 0x1853, 0x0501, 0x0503, 0x0404, 0x1844};
__constant__ const unsigned short cf_wg_shares3[6] = {0x1860, 0x1861, 0x0600, 0x0601, 0x0402, 0x1842};

template <int KS>
__global__ void __launch_bounds__(128 * KS + 128) conv_f32_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                 float* __restrict__ part, int Cin, int Cout, int H, int W,
                                                                 int tiles_x, int tiles_y, int ntiles, int nsplit) {
    static_assert(CF_WG_RB == 1 && CF_WG_TH == 4 && CF_TW == 32 && CF_WG_HALVES == 2, "the wave shares assume 4 x 32 tiles");
    constexpr int P = KS / 2, THREADS = 128 * KS + 128;   // 2 K compute waves + 2 loader waves
    constexpr int RH = CF_WG_TH + KS - 1, RW = CF_TW + KS - 1;
    // plane pitches = 2 * odd: the 32 channel lanes x 2 pixel lanes of an operand read fall on 64 different banks, and
    // the staged pairs / quads stay 8-byte aligned (ds_write_b64)
    constexpr int GPL = CF_WG_TH * CF_TW + 2, XPL = RH * RW + 2;
    constexpr int GSZ = 32 * GPL, BUF = GSZ + 32 * XPL;
    static_assert((RH * RW) % 4 == 0, "pitch parity");
    extern __shared__ __attribute__((aligned(16))) char cf_wg_smem[];
    float* lds = reinterpret_cast<float*>(cf_wg_smem);       // [2][BUF]: dY planes, then X planes (one buffer on the slow path)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned share = KS == 5 ? cf_wg_shares5[wv < 10 ? wv : 0] : cf_wg_shares3[wv < 6 ? wv : 0];
    const int ty = share & 15, hb = (share >> 4) & 15, he = (share >> 8) & 15, half = share >> 12;
    const int c = lane & 31, kk = lane >> 5;
    // Block order: the (co, ci) groups of one pixel split read the same dY / X tiles, so they should run at the same time
    // on the same XCD (each XCD has its own L2; consecutive workgroup ids go round the 8 XCDs).  id = 8 * (G * q + group)
    // + xcd with split = 8 q + xcd: the G blocks of a split follow each other on one XCD and the tiles are fetched from
    // HBM about once instead of once per group.
    const int gco = (Cout + 31) / 32, G = gco * ((Cin + 31) / 32);
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int group = j % G, split = (j / G) * 8 + xcd;
    if (split >= nsplit) return;     // workgroup-uniform: the grid is padded to whole rounds of 8 splits
    const int co0 = (group % gco) * 32, ci0 = (group / gco) * 32;
    const long long plane = (long long)H * W;
    const int t0 = (int)((long long)split * ntiles / nsplit), t1 = (int)((long long)(split + 1) * ntiles / nsplit);
    // Fast staging (K = 5, W % 4 == 0, aligned bases): two extra LOADER waves (which the hardware puts on the two SIMDs
    // that hold only two compute waves) bring the next tile's dY and X quads through registers into the OTHER LDS buffer
    // while the compute waves have this tile in the matrix cores: the compute waves issue nothing but operand reads and
    // MFMAs, the staging's address arithmetic runs beside them on the vector ALUs, one barrier per tile.
    //   dY: thread = (plane co, tile row), its 8 quads along the row;   X: lane = (window row, channel % 8), its 10
    //   aligned quads [w0 - 4, w0 + 36) of the row (the window is [w0 - 2, w0 + 34)) for 2 channel groups per wave.
    // Every request is base (uniform: image, column quad) + one 32-bit lane offset (channel, clamped row): no per-request
    // address arithmetic and no per-request selects (a value select after a load is compiled into a branch that waits for
    // it, an address select into two loads in two branches).  Rows / channels outside the tensor are clamped and zeroed
    // by a lane mask, column quads outside the tensor (uniform) read the tile's first quad and are zeroed likewise.
    const bool fast = KS == 5 && (W % 4 == 0) && (long long)(Cin > Cout ? Cin : Cout) * plane < (1LL << 31) &&
                      ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x)) & 15) == 0;
    const bool loader = wv >= CF_WG_HALVES * KS;               // wave-uniform
    if constexpr (KS == 5) {
    if (fast && loader) {     // (no accumulators on this path: the staging registers have the file to themselves)
        static_assert(KS != 5 || (RH == 8 && RW == 36), "loader lane mapping");
        constexpr int XQ = 10, GQ = CF_TW / 4;
        const int ltid = tid - CF_WG_HALVES * KS * 64, lw = ltid >> 6;
        const int g_co = ltid >> 2, g_r = ltid & 3;
        const bool g_ch_ok = co0 + g_co < Cout;
        const unsigned g_ch_off = (unsigned)(g_ch_ok ? co0 + g_co : Cout - 1) * (unsigned)plane;
        const int g_dst = g_co * GPL + g_r * CF_TW;
        const int x_r = ltid & 7, x_cl = (ltid >> 3) & 7;
        bool x_ch_ok[2];
        unsigned x_ch_off[2];
        int x_dst[2];
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const int ci = (2 * lw + gi) * 8 + x_cl;
            x_ch_ok[gi] = ci0 + ci < Cin;
            x_ch_off[gi] = (unsigned)(x_ch_ok[gi] ? ci0 + ci : Cin - 1) * (unsigned)plane;
            x_dst[gi] = GSZ + ci * XPL + x_r * RW - 2;
        }
        auto stage = [&](int t, float* buf) {
            const int n = t / (tiles_x * tiles_y), rem = t - n * (tiles_x * tiles_y);
            const int h0 = (rem / tiles_x) * CF_WG_TH, w0 = (rem % tiles_x) * CF_TW;       // all uniform
            const float* gimg = dy + (long long)n * Cout * plane;
            const float* ximg = x + (long long)n * Cin * plane;
            const int gh = h0 + g_r;
            const unsigned g_off = g_ch_off + (unsigned)(gh < H ? gh : H - 1) * (unsigned)W;
            const unsigned g_mask = g_ch_ok && gh < H ? ~0u : 0u;
            const int xh = h0 - P + x_r, xhc = xh < 0 ? 0 : (xh < H ? xh : H - 1);
            const bool x_row_ok = xh >= 0 && xh < H;
            float4 gq[GQ], xq[2][XQ];
#pragma unroll
            for (int jq = 0; jq < GQ; ++jq) {
                const int gw = w0 + 4 * jq;
                const float* col = gimg + (gw < W ? gw : w0);                       // uniform
                gq[jq] = *reinterpret_cast<const float4*>(col + g_off);
            }
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                const unsigned x_off = x_ch_off[gi] + (unsigned)xhc * (unsigned)W;
#pragma unroll
                for (int jq = 0; jq < XQ; ++jq) {
                    const int gw = w0 - 4 + 4 * jq;
                    const float* col = ximg + (gw >= 0 && gw < W ? gw : w0);        // uniform
                    xq[gi][jq] = *reinterpret_cast<const float4*>(col + x_off);
                }
            }
            auto masked = [](float v, unsigned m) { return __uint_as_float(__float_as_uint(v) & m); };
#pragma unroll
            for (int jq = 0; jq < GQ; ++jq) {
                const unsigned m = w0 + 4 * jq < W ? g_mask : 0u;
                float2* d = reinterpret_cast<float2*>(buf + g_dst + 4 * jq);
                d[0] = make_float2(masked(gq[jq].x, m), masked(gq[jq].y, m));
                d[1] = make_float2(masked(gq[jq].z, m), masked(gq[jq].w, m));
            }
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                const unsigned x_mask = x_ch_ok[gi] && x_row_ok ? ~0u : 0u;
#pragma unroll
                for (int jq = 0; jq < XQ; ++jq) {
                    const int gw = w0 - 4 + 4 * jq;
                    const unsigned m = gw >= 0 && gw < W ? x_mask : 0u;
                    float* d = buf + x_dst[gi] + 4 * jq;                           // window columns 4 jq - 2 .. 4 jq + 1
                    if (jq > 0) *reinterpret_cast<float2*>(d) = make_float2(masked(xq[gi][jq].x, m), masked(xq[gi][jq].y, m));
                    if (jq < XQ - 1) *reinterpret_cast<float2*>(d + 2) = make_float2(masked(xq[gi][jq].z, m), masked(xq[gi][jq].w, m));
                }
            }
        };
        if (t0 < t1) stage(t0, lds);
        __syncthreads();
        for (int t = t0; t < t1; ++t) {
            if (t + 1 < t1) stage(t + 1, lds + (((t - t0) & 1) ^ 1) * BUF);
            __syncthreads();   // the compute waves are done with this tile's buffer, the other one is filled
        }
        return;
    }
    }
    cf_f32x16 acc[KS];
#pragma unroll
    for (int tx = 0; tx < KS; ++tx)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[tx][i] = 0.f;
    // Half rows [h0, h1) of the tile in `buf`: pixel pair (half row hr, q) = row hr / 2, columns 16 (hr & 1) + 2 q + kk;
    // lane (c, kk): A = dY[co = c][pixel], B = X[ci = c][pixel + tap].  The operands of pair q + 1 are requested before
    // the K MFMAs of pair q are issued (two register sets), so the LDS latency is covered by the wave's own matrix work.
    auto compute = [&](const float* buf, int h0, int h1) {
        if (h0 >= h1) return;
        const float* gl = buf + c * GPL + kk;
        const float* xl = buf + GSZ + c * XPL + ty * RW + kk;
        float a[2], b[2][KS];
        auto request = [&](int hr, int q, int slot) {
            a[slot] = gl[hr * 16 + 2 * q];
            const float* xp = xl + (hr >> 1) * RW + (hr & 1) * 16 + 2 * q;
#pragma unroll
            for (int tx = 0; tx < KS; ++tx) b[slot][tx] = xp[tx];
        };
        request(h0, 0, 0);
        for (int hr = h0; hr < h1; ++hr) {
            const int hn = hr + 1 < h1 ? hr + 1 : hr;          // the last request past the range re-reads (unused)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (q < 7) request(hr, q + 1, (q + 1) & 1);
                else request(hn, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tx = 0; tx < KS; ++tx)
                    acc[tx] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q & 1], b[q & 1][tx], acc[tx], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    if (fast) {
        __syncthreads();
        for (int t = t0; t < t1; ++t) {
            compute(lds + ((t - t0) & 1) * BUF, hb, he);
            __syncthreads();
        }
    } else {
    for (int t = t0; t < t1; ++t) {
        const int n = t / (tiles_x * tiles_y), rem = t - n * (tiles_x * tiles_y);
        const int h0 = (rem / tiles_x) * CF_WG_TH, w0 = (rem % tiles_x) * CF_TW;
        __syncthreads();
        for (int e = tid; e < 32 * CF_WG_TH * CF_TW; e += THREADS) {
            const int co = e / (CF_WG_TH * CF_TW), p = e - co * (CF_WG_TH * CF_TW);
            const int gh = h0 + p / CF_TW, gw = w0 + p % CF_TW;
            float v = 0.f;
            if (co0 + co < Cout && gh < H && gw < W) v = dy[((long long)n * Cout + co0 + co) * plane + (long long)gh * W + gw];
            lds[co * GPL + p] = v;
        }
        for (int e = tid; e < 32 * RH * RW; e += THREADS) {
            const int ci = e / (RH * RW), r = (e - ci * (RH * RW)) / RW, col = e - ci * (RH * RW) - r * RW;
            const int gh = h0 - P + r, gw = w0 - P + col;
            float v = 0.f;
            if (ci0 + ci < Cin && gh >= 0 && gh < H && gw >= 0 && gw < W)
                v = x[((long long)n * Cin + ci0 + ci) * plane + (long long)gh * W + gw];
            lds[GSZ + ci * XPL + r * RW + col] = v;
        }
        __syncthreads();
        if (!loader) compute(lds, hb, he);
    }
    }
    if (loader) return;
    // D[row = co][col = ci]: slab [tap][co][ci] -- the 32 ci lanes of a row are one 128-byte segment (the tensor's own
    // order [co][ci][tap] would make every store 64 scattered words); conv_f32_wgrad_reduce_kernel transposes
    float* dst = part + ((long long)split * CF_WG_HALVES + half) * Cout * Cin * KS * KS;
    const int ci = ci0 + c;
    if (ci < Cin) {
#pragma unroll
        for (int tx = 0; tx < KS; ++tx)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int co = co0 + (reg & 3) + 8 * (reg >> 2) + 4 * kk;
                if (co < Cout) dst[((long long)(ty * KS + tx) * Cout + co) * Cin + ci] = acc[tx][reg];
            }
    }
}

// LDS bytes of the weight-gradient kernel: two tile buffers where the fast staging can apply (even padding)
static size_t cf_wg_lds_bytes(int K) {
    const int RH = CF_WG_TH + K - 1, RW = CF_TW + K - 1;
    const size_t buf = (size_t)32 * (CF_WG_TH * CF_TW + 2) + (size_t)32 * (RH * RW + 2);
    return ((K / 2) % 2 == 0 ? 2 : 1) * buf * sizeof(float);
}

// 64 slab elements x 4 slab lanes per block: lane z adds the slabs z, z + 4, ... in that order (8 requests in flight, 256-byte
// segments), the 4 lanes are then added in lane order: a fixed order, so the result is deterministic.  Slab order
// [tap][co][ci] -> tensor order [co][ci][tap]
__global__ void __launch_bounds__(256) conv_f32_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                    long long total, int nslabs, int CC, int KK) {
    __shared__ float red[256];
    const int el = threadIdx.x & 63, zl = threadIdx.x >> 6;
    const long long idx = (long long)blockIdx.x * 64 + el;
    const long long idc = idx < total ? idx : total - 1;
    float s = 0.f;
    for (int z0 = zl; z0 < nslabs; z0 += 4 * 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int z = z0 + 4 * j < nslabs ? z0 + 4 * j : nslabs - 1;
            v[j] = part[(long long)z * total + idc];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) s += z0 + 4 * j < nslabs ? v[j] : 0.f;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (zl == 0 && idx < total) {
        const float t = ((red[el] + red[64 + el]) + red[128 + el]) + red[192 + el];
        const int tap = (int)(idx / CC), cc = (int)(idx - (long long)tap * CC);
        dw[(long long)cc * KK + tap] = t;
    }
}

static int cf_mpad(int64_t M) { return (int)(cdiv(M, CF_MB) * CF_MB); }
static int cf_nsplit(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W) {
    const int64_t ntiles = N * cdiv(H, CF_WG_TH) * cdiv(W, CF_TW);
    const int64_t groups = cdiv(Cout, 32 * CF_WG_RB) * cdiv(Cin, 32);
    // one block per CU, one round: each block pays its prologue (first tile's staging, exposed) and its partial slab
    // once (512 blocks: -1 %, 1024: -4 % on 64 -> 256 @128, -13 % @64)
    static const int blocks = [] { const char* e = getenv("OFASR_CONV_F32_WG_BLOCKS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 256; }();
    int64_t want = blocks / (groups > 0 ? groups : 1);
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
    const size_t lds = cf_wg_lds_bytes(K);
    if (K == 5) {
        static const bool attr5 = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f32_wgrad_kernel<5>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)cf_wg_lds_bytes(5)), true);
        (void)attr5;
        OFASR_LAUNCH(conv_f32_wgrad_kernel<5>, grid, dim3(64 * 5 * CF_WG_HALVES + 128), lds, st, (const float*)dy, (const float*)x, (float*)workspace,
                     (int)Cin, (int)Cout, (int)H, (int)W, tiles_x, tiles_y, ntiles, nsplit);
    } else {
        OFASR_LAUNCH(conv_f32_wgrad_kernel<3>, grid, dim3(64 * 3 * CF_WG_HALVES + 128), lds, st, (const float*)dy, (const float*)x, (float*)workspace,
                     (int)Cin, (int)Cout, (int)H, (int)W, tiles_x, tiles_y, ntiles, nsplit);
    }
    int rc = check_launch(name);
    if (rc) return rc;
    const long long total = (long long)Cout * Cin * K * K;
    OFASR_LAUNCH(conv_f32_wgrad_reduce_kernel, dim3((unsigned)cdiv(total, 64)), dim3(256), 0, st, (const float*)workspace, dw,
                 total, nsplit * CF_WG_HALVES, (int)(Cout * Cin), K * K);
    return check_launch(name);
}
