// 64 -> 64 channel 1-D convolutions on [B,64,T] fp32 frames as implicit GEMMs on the gfx950
// fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, frees the VALU for the fused
// prologue / epilogue work).
//
// Replaces, for the hot path of py/main16.py:
//   * ResBlock's two Conv1d(64,64,3,padding=1) (+ the BatchNorm1d / ReLU around them)   :115-121
//   * Generator.decoder[0] = ConvTranspose1d(64,64,7,padding=3) (+ the embedding add)      :144,156-159
//   * their data-gradients (same kernel, re-packed weights) and weight-gradients.
//
// One persistent 256-thread workgroup per CU streams (clip, time-tile) tiles:
//   global --(dwordx4, software-pipelined one tile ahead, through registers so the prologue
//   transform can be applied)--> LDS [64][NT+halo]  --ds_read_b32--> MFMA B operand
//   packed weights [tap][cin][cout] live in LDS for the whole kernel --> MFMA A operand
//   D tile (cout x time) leaves the accumulators as 128-B row segments (time is contiguous).
#include "wm_common.hpp"
using namespace wm;
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifdef WM_STAMP
// diagnostic build only (-DWM_STAMP): per-wave cycle totals of the kernel's phases
__device__ unsigned long long* g_wm_stamp = nullptr;
#define STAMP(var) unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define STAMP(var)
#endif

namespace {

enum { PRO_NONE = 0, PRO_BNRELU = 1, PRO_ADDVEC = 2, PRO_BNBWD = 3 };
enum { EPI_BIAS = 0, EPI_RELUMASK = 1, EPI_ADD = 2, EPI_NONE = 3, EPI_BNADDRELU = 4,    // 4: relu(e1 + (acc + bias) * ea + eb), conv64bf3 only
       EPI_BIASELU = 5, EPI_BIASADDELU = 6, EPI_MULDELU = 7,
       EPI_ADDSTATS = 8 };    // dwgrad64bf only: EPI_ADD, then the ReLU mask of the PREVIOUS block and its two BatchNorm sums   // main14b_2's 64-channel blocks (conv64bf_kernel only): elu(acc + bias),
                                                                 // elu(acc + bias + e1), acc * ELU'(e1) with e1 = the ELU output y

struct Conv64Args {
    const float* x;     // [B,64,T] primary input
    const float* x2;    // [B,64,T] second input (PRO_BNBWD)
    const float* wp;    // packed weights [KW][64 in][64 out]
    const float* pa;    // prologue per-channel a  (BNRELU: scale, BNBWD: A, ADDVEC: vec[B,64])
    const float* pb;    // prologue per-channel b  (BNRELU: shift, BNBWD: B)
    const float* pc;    // prologue per-channel c  (BNBWD: C)
    const float* bias;  // [64] (EPI_BIAS)
    const float* e1;    // [B,64,T] epilogue tensor (RELUMASK: pre-BN activation, ADD: addend)
    const float* ea;    // [64] epilogue scale (RELUMASK)
    const float* eb;    // [64] epilogue shift (RELUMASK)
    float* y;           // [B,64,T]
    float* stats;       // [gridDim.x][2][64] partial sums or nullptr
    int B, T;
};

template <int PRO>
__device__ __forceinline__ float pro_apply(float v, float v2, float ca, float cb, float cc, float cl = 0.f) {
    if (PRO == PRO_BNRELU) return fmaxf(fmaf(v, ca, cb), 0.f);
    if (PRO == PRO_ADDVEC) return v + ca;
    if (PRO == PRO_BNBWD) return fmaf(ca, v, fmaf(cc, v2, cb)) + cl;      // cb + cl: offset as hi + lo words (pb[0..63], pb[64..127])
    return v;
}

template <int KW, int NT, int PRO, int EPI, bool STATS>
__global__ __launch_bounds__(256) void conv64_kernel(Conv64Args a) {
    constexpr int PAD = KW / 2;
    constexpr int XS = NT + 8;            // LDS row stride; main part starts at column 4 (16-B aligned)
    constexpr int NTW = NT / 4;           // time columns per wave
    constexpr int NN = NTW / 32;          // 32-wide N tiles per wave
    constexpr int QR = NT / 4;            // float4 per row
    constexpr int NV = 64 * QR / 256;     // float4 per thread per staged tensor
    constexpr int HTOT = 64 * 2 * PAD;    // halo elements per tile
    constexpr int HN = (HTOT + 255) / 256;
    constexpr bool TWO = (PRO == PRO_BNBWD);
    constexpr bool E1 = (EPI == EPI_RELUMASK || EPI == EPI_ADD);

    extern __shared__ __align__(16) float smem[];
    float* Ws = smem;                     // [KW*64][64]
    float* Xs = smem + KW * 4096;         // [64][XS]
    float* Cs = Xs + 64 * XS;             // [6][64] per-channel constants: pa pb pc bias ea eb

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int T = a.T;
    const int tilesPerClip = (T + NT - 1) / NT;
    const int ntiles = a.B * tilesPerClip;

    float4 st[NV];
    float4 st2[TWO ? NV : 1];
    float hl[HN], hl2[TWO ? HN : 1];

    // Loads are UNCONDITIONAL (addresses clamped into the clip) and the out-of-range masking happens when the tile is
    // written to LDS: a per-element "load or zero" select makes hipcc branch around every load and drain vmcnt at
    // each join -- measured 15 K cycles of serialised round trips per tile in the two-tensor variants.
    auto load_tile = [&](int tile) {
        const int b = tile / tilesPerClip, t0 = (tile - b * tilesPerClip) * NT;
        const float* xb = a.x + (size_t)b * 64 * T;
        const float* xb2 = TWO ? a.x2 + (size_t)b * 64 * T : nullptr;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * 256, c = idx / QR, q = idx % QR;
            const int t = min(t0 + 4 * q, T - 4);
            st[i] = *reinterpret_cast<const float4*>(xb + (size_t)c * T + t);
            if (TWO) st2[i] = *reinterpret_cast<const float4*>(xb2 + (size_t)c * T + t);
        }
#pragma unroll
        for (int i = 0; i < HN; ++i) {
            const int idx = min(tid + i * 256, HTOT - 1);
            const int c = idx / (2 * PAD), h = idx % (2 * PAD);
            const int t = min(max((h < PAD) ? t0 - PAD + h : t0 + NT + (h - PAD), 0), T - 1);
            hl[i] = xb[(size_t)c * T + t];
            if (TWO) hl2[i] = xb2[(size_t)c * T + t];
        }
    };
    auto write_tile = [&](int tile) {
        const int b = tile / tilesPerClip, t0 = (tile - b * tilesPerClip) * NT;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * 256, c = idx / QR, q = idx % QR, t = t0 + 4 * q;
            float4 v = st[i];
            if (PRO != PRO_NONE) {
                const float ca = (PRO == PRO_ADDVEC) ? a.pa[b * 64 + c] : Cs[c];
                const float cb = Cs[64 + c], cc = Cs[128 + c], cl = TWO ? Cs[192 + c] : 0.f;
                const float4 w = TWO ? st2[i] : v;
                v.x = pro_apply<PRO>(v.x, w.x, ca, cb, cc, cl);
                v.y = pro_apply<PRO>(v.y, w.y, ca, cb, cc, cl);
                v.z = pro_apply<PRO>(v.z, w.z, ca, cb, cc, cl);
                v.w = pro_apply<PRO>(v.w, w.w, ca, cb, cc, cl);
            }
            if (t >= T) v = make_float4(0.f, 0.f, 0.f, 0.f);     // T % 4 == 0: a float4 is all in or all out
            *reinterpret_cast<float4*>(Xs + c * XS + 4 + 4 * q) = v;
        }
#pragma unroll
        for (int i = 0; i < HN; ++i) {
            const int idx = tid + i * 256;
            if (idx < HTOT) {
                const int c = idx / (2 * PAD), h = idx % (2 * PAD);
                const int t = (h < PAD) ? t0 - PAD + h : t0 + NT + (h - PAD);
                float v = hl[i];
                if (PRO != PRO_NONE) {
                    const float ca = (PRO == PRO_ADDVEC) ? a.pa[b * 64 + c] : Cs[c];
                    v = pro_apply<PRO>(v, TWO ? hl2[i] : v, ca, Cs[64 + c], Cs[128 + c], TWO ? Cs[192 + c] : 0.f);
                }
                if (t < 0 || t >= T) v = 0.f;
                Xs[c * XS + ((h < PAD) ? 4 - PAD + h : 4 + NT + (h - PAD))] = v;
            }
        }
    };

    int tile = blockIdx.x;
    if (tile < ntiles) load_tile(tile);
    // resident operands: packed weights + per-channel constants
    for (int i = tid; i < KW * 1024; i += 256)
        reinterpret_cast<float4*>(Ws)[i] = reinterpret_cast<const float4*>(a.wp)[i];
    if (tid < 64) {
        Cs[tid] = (PRO == PRO_BNRELU || PRO == PRO_BNBWD) ? a.pa[tid] : 0.f;
        Cs[64 + tid] = (PRO == PRO_BNRELU || PRO == PRO_BNBWD) ? a.pb[tid] : 0.f;
        Cs[128 + tid] = (PRO == PRO_BNBWD) ? a.pc[tid] : 0.f;
        Cs[192 + tid] = (EPI == EPI_BIAS && a.bias) ? a.bias[tid] : 0.f;
        Cs[256 + tid] = (EPI == EPI_RELUMASK) ? a.ea[tid] : 0.f;
        Cs[320 + tid] = (EPI == EPI_RELUMASK) ? a.eb[tid] : 0.f;
    }
    __syncthreads();
    if (tile < ntiles) write_tile(tile);
    __syncthreads();

    float s1[STATS ? 32 : 1], s2[STATS ? 32 : 1];
    if (STATS) {
#pragma unroll
        for (int j = 0; j < 32; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    }

#ifdef WM_STAMP
    unsigned long long tm[6] = {0, 0, 0, 0, 0, 0};
#endif
    while (tile < ntiles) {
        STAMP(ts0);
        const int next = tile + gridDim.x;
        if (next < ntiles) load_tile(next);           // in flight while the matrix cores work

        const int b = tile / tilesPerClip, t0 = (tile - b * tilesPerClip) * NT;
        float e1r[E1 ? 2 * NN * 16 : 1];
        if (E1) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NN; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = mt * 32 + mfma_row(r, half), t = min(t0 + wave * NTW + nt * 32 + l31, T - 1);
                        e1r[(mt * NN + nt) * 16 + r] = a.e1[((size_t)b * 64 + co) * T + t];      // masked at use (t < T)
                    }
        }

        STAMP(ts1);
        f32x16 acc[2][NN];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NN; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

        const float* xcol = Xs + half * XS + (4 - PAD) + wave * NTW + l31;
        const float* wrow = Ws + half * 64 + l31;
#pragma unroll 1
        for (int tap = 0; tap < KW; ++tap) {
#pragma unroll 8
            for (int cp = 0; cp < 32; ++cp) {
                const float* wk = wrow + (tap * 64 + 2 * cp) * 64;
                const float* xk = xcol + (2 * cp) * XS + tap;
                const float a0 = wk[0], a1 = wk[32];
                float bv[NN];
#pragma unroll
                for (int nt = 0; nt < NN; ++nt) bv[nt] = xk[nt * 32];
#pragma unroll
                for (int nt = 0; nt < NN; ++nt) {
                    acc[0][nt] = mfma32(a0, bv[nt], acc[0][nt]);
                    acc[1][nt] = mfma32(a1, bv[nt], acc[1][nt]);
                }
            }
        }

        STAMP(ts2);
        // epilogue straight from the accumulators
        float* yb = a.y + (size_t)b * 64 * T;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NN; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = mt * 32 + mfma_row(r, half), t = t0 + wave * NTW + nt * 32 + l31;
                    float v = acc[mt][nt][r];
                    float q = 0.f;
                    if (EPI == EPI_BIAS) v += Cs[192 + co];
                    if (EPI == EPI_RELUMASK) {
                        q = e1r[(mt * NN + nt) * 16 + r];
                        v = (fmaf(q, Cs[256 + co], Cs[320 + co]) > 0.f) ? v : 0.f;
                    }
                    if (EPI == EPI_ADD) v += e1r[(mt * NN + nt) * 16 + r];
                    if (t < T) {
                        yb[(size_t)co * T + t] = v;
                        if (STATS) {
                            s1[mt * 16 + r] += v;
                            s2[mt * 16 + r] += (EPI == EPI_RELUMASK) ? v * q : v * v;
                        }
                    }
                }

        STAMP(ts3);
        __syncthreads();                              // every wave is done with Xs
        STAMP(ts4);
        if (next < ntiles) write_tile(next);
        STAMP(ts5);
        __syncthreads();
        STAMP(ts6);
#ifdef WM_STAMP
        tm[0] += ts1 - ts0; tm[1] += ts2 - ts1; tm[2] += ts3 - ts2; tm[3] += ts4 - ts3; tm[4] += ts5 - ts4; tm[5] += ts6 - ts5;
#endif
        tile = next;
    }
#ifdef WM_STAMP
    if (g_wm_stamp && lane == 0) {
        unsigned long long* d = g_wm_stamp + ((size_t)blockIdx.x * 4 + wave) * 6;
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i] = tm[i];
    }
#endif

    if (STATS) {
        float* red = Xs;                              // [4 waves][2][64]
#pragma unroll
        for (int j = 0; j < 32; ++j) { s1[j] = half_wave_sum(s1[j]); s2[j] = half_wave_sum(s2[j]); }
        if (l31 == 0) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = mt * 32 + mfma_row(r, half);
                    red[wave * 128 + co] = s1[mt * 16 + r];
                    red[wave * 128 + 64 + co] = s2[mt * 16 + r];
                }
        }
        __syncthreads();
        if (tid < 128)
            a.stats[(size_t)blockIdx.x * 128 + tid] = red[tid] + red[128 + tid] + red[256 + tid] + red[384 + tid];
    }
}

template <int KW, int NT, int PRO, int EPI, bool STATS>
int launch_conv64(const Conv64Args& a, hipStream_t stream) {
    constexpr size_t lds = (size_t)(KW * 4096 + 64 * (NT + 8) + 6 * 64) * sizeof(float);
    static wm::DevOnce attr_done;
    auto kern = conv64_kernel<KW, NT, PRO, EPI, STATS>;
    if (!wm::dev_done(attr_done)) {
        WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        wm::dev_mark(attr_done);
    }
    const int ntiles = a.B * ((a.T + NT - 1) / NT);
    const int grid = ntiles < kNumCU ? ntiles : kNumCU;
    if (STATS && grid < kNumCU)
        WM_TRY(hipMemsetAsync(a.stats, 0, sizeof(float) * 128 * kNumCU, stream));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
    WM_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// weight packing: parameter layout -> [tap][in][out] GEMM-A image (runs every step: Adam updates
// the parameters in place, so the image is rebuilt from the live tensors; 12-28 K elements).
//   mode 0  conv   forward : wp[tap][ci][co] = w[co][ci][tap]
//   mode 1  conv   dgrad   : wp[tap][co][ci] = w[co][ci][KW-1-tap]
//   mode 2  convT  forward : wp[tap][ci][co] = w[ci][co][KW-1-tap]      (w is [in][out][k])
//   mode 3  convT  dgrad   : wp[tap][co][ci] = w[ci][co][tap]
// ---------------------------------------------------------------------------------------------
__global__ void pack_w64_kernel(const float* __restrict__ w, float* __restrict__ wp, int KW, int mode) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= KW * 4096) return;
    const int tap = i / 4096, in = (i / 64) % 64, out = i % 64;
    int src;
    if (mode == 0) src = (out * 64 + in) * KW + tap;
    else if (mode == 1) src = (in * 64 + out) * KW + (KW - 1 - tap);
    else if (mode == 2) src = (in * 64 + out) * KW + (KW - 1 - tap);
    else src = (out * 64 + in) * KW + tap;
    wp[i] = w[src];
}

// ---------------------------------------------------------------------------------------------
// weight gradient:  G[tap][out][in] = sum_{b,t} g[b,out,t] * xin[b,in,t+tap-PAD]   (+ bias grad)
// as a GEMM with the (b,t) axis as the contraction (K of the MFMA = 2 time steps).
// Each workgroup writes one partial slab; wgrad64_reduce sums the slabs in a fixed order
// (bitwise reproducible, no float atomics) into the parameter-gradient layout.
// ---------------------------------------------------------------------------------------------
struct Wgrad64Args {
    const float* g;  const float* g2;                   // gradient tensor(s) [B,64,T]
    const float* ga; const float* gb; const float* gc;  // BNBWD constants for g
    const float* x;                                     // layer input [B,64,T]
    const float* xa; const float* xb;                   // x prologue constants (BNRELU scale/shift, ADDVEC vec[B,64])
    float* partial;                                     // [grid][KW*4096 + 64]
    int B, T;
};

template <int KW, int GPRO, int XPRO>
__global__ __launch_bounds__(256) void wgrad64_kernel(Wgrad64Args a) {
    constexpr int NT = 128, PAD = KW / 2;
    constexpr int GS = NT + 4, XS = NT + 8;
    constexpr int QR = NT / 4, NV = 64 * QR / 256;      // 8 float4 per thread per tensor
    constexpr int HTOT = 64 * 2 * PAD, HN = (HTOT + 255) / 256;
    constexpr bool TIME_SPLIT = (KW == 3);              // waves split the tile's time range, else the taps
    constexpr int TL = TIME_SPLIT ? KW : 2;             // taps held per wave
    constexpr bool GTWO = (GPRO == PRO_BNBWD);

    extern __shared__ __align__(16) float smem[];
    float* Gs = smem;                 // [64][GS]
    float* Xs = Gs + 64 * GS;         // [64][XS]
    float* Cs = Xs + 64 * XS;         // [5][64]: ga gb gc xa xb

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int T = a.T;
    const int tilesPerClip = (T + NT - 1) / NT, ntiles = a.B * tilesPerClip;

    float4 sg[NV], sg2[GTWO ? NV : 1], sx[NV];
    float hx[HN];
    float bsum[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) bsum[i] = 0.f;

    auto load_tile = [&](int tile) {            // branch-free, see conv64_kernel
        const int b = tile / tilesPerClip, t0 = (tile - b * tilesPerClip) * NT;
        const size_t base = (size_t)b * 64 * T;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * 256, c = idx / QR, q = idx % QR;
            const int t = min(t0 + 4 * q, T - 4);
            sg[i] = *reinterpret_cast<const float4*>(a.g + base + (size_t)c * T + t);
            if (GTWO) sg2[i] = *reinterpret_cast<const float4*>(a.g2 + base + (size_t)c * T + t);
            sx[i] = *reinterpret_cast<const float4*>(a.x + base + (size_t)c * T + t);
        }
#pragma unroll
        for (int i = 0; i < HN; ++i) {
            const int idx = min(tid + i * 256, HTOT - 1);
            const int c = idx / (2 * PAD), h = idx % (2 * PAD);
            const int t = min(max((h < PAD) ? t0 - PAD + h : t0 + NT + (h - PAD), 0), T - 1);
            hx[i] = a.x[base + (size_t)c * T + t];
        }
    };
    auto write_tile = [&](int tile) {
        const int b = tile / tilesPerClip, t0 = (tile - b * tilesPerClip) * NT;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * 256, c = idx / QR, q = idx % QR, t = t0 + 4 * q;
            float4 v = sg[i], u = sx[i];
            if (GPRO == PRO_BNBWD) {
                const float ca = Cs[c], cb = Cs[64 + c], cc = Cs[128 + c], cl = Cs[320 + c];
                const float4 w = sg2[i];
                v.x = pro_apply<PRO_BNBWD>(v.x, w.x, ca, cb, cc, cl);
                v.y = pro_apply<PRO_BNBWD>(v.y, w.y, ca, cb, cc, cl);
                v.z = pro_apply<PRO_BNBWD>(v.z, w.z, ca, cb, cc, cl);
                v.w = pro_apply<PRO_BNBWD>(v.w, w.w, ca, cb, cc, cl);
            }
            if (XPRO != PRO_NONE) {
                const float ca = (XPRO == PRO_ADDVEC) ? a.xa[b * 64 + c] : Cs[192 + c];
                const float cb = Cs[256 + c];
                u.x = pro_apply<XPRO>(u.x, 0.f, ca, cb, 0.f);
                u.y = pro_apply<XPRO>(u.y, 0.f, ca, cb, 0.f);
                u.z = pro_apply<XPRO>(u.z, 0.f, ca, cb, 0.f);
                u.w = pro_apply<XPRO>(u.w, 0.f, ca, cb, 0.f);
            }
            if (t >= T) { v = make_float4(0.f, 0.f, 0.f, 0.f); u = v; }
            bsum[i] += (v.x + v.y) + (v.z + v.w);
            *reinterpret_cast<float4*>(Gs + c * GS + 4 * q) = v;
            *reinterpret_cast<float4*>(Xs + c * XS + 4 + 4 * q) = u;
        }
#pragma unroll
        for (int i = 0; i < HN; ++i) {
            const int idx = tid + i * 256;
            if (idx < HTOT) {
                const int c = idx / (2 * PAD), h = idx % (2 * PAD);
                const int t = (h < PAD) ? t0 - PAD + h : t0 + NT + (h - PAD);
                float v = hx[i];
                if (XPRO != PRO_NONE) {
                    const float ca = (XPRO == PRO_ADDVEC) ? a.xa[b * 64 + c] : Cs[192 + c];
                    v = pro_apply<XPRO>(v, 0.f, ca, Cs[256 + c], 0.f);
                }
                if (t < 0 || t >= T) v = 0.f;
                Xs[c * XS + ((h < PAD) ? 4 - PAD + h : 4 + NT + (h - PAD))] = v;
            }
        }
    };

    int tile = blockIdx.x;
    if (tile < ntiles) load_tile(tile);
    if (tid < 64) {
        Cs[tid] = GTWO ? a.ga[tid] : 0.f;
        Cs[64 + tid] = GTWO ? a.gb[tid] : 0.f;
        Cs[128 + tid] = GTWO ? a.gc[tid] : 0.f;
        Cs[192 + tid] = (XPRO == PRO_BNRELU) ? a.xa[tid] : 0.f;
        Cs[256 + tid] = (XPRO == PRO_BNRELU) ? a.xb[tid] : 0.f;
        Cs[320 + tid] = GTWO ? a.gb[64 + tid] : 0.f;        // low word of the BatchNorm-backward offset
    }
    __syncthreads();
    if (tile < ntiles) write_tile(tile);
    __syncthreads();

    f32x16 acc[TL][2][2];
#pragma unroll
    for (int tl = 0; tl < TL; ++tl)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tl][mt][nt][r] = 0.f;

    const int tbeg = TIME_SPLIT ? wave * (NT / 4) : 0;
    constexpr int NSTEP = (TIME_SPLIT ? NT / 4 : NT) / 2;

    while (tile < ntiles) {
        const int next = tile + gridDim.x;
        if (next < ntiles) load_tile(next);
        const float* gp = Gs + l31 * GS + tbeg + half;
        const float* xp = Xs + l31 * XS + (4 - PAD) + tbeg + half;
#pragma unroll 4
        for (int s = 0; s < NSTEP; ++s) {
            const float a0 = gp[2 * s], a1 = gp[32 * GS + 2 * s];
#pragma unroll
            for (int tl = 0; tl < TL; ++tl) {
                const int tap = TIME_SPLIT ? tl : wave + 4 * tl;
                if (TIME_SPLIT || tap < KW) {
                    const float b0 = xp[2 * s + tap], b1 = xp[32 * XS + 2 * s + tap];
                    acc[tl][0][0] = mfma32(a0, b0, acc[tl][0][0]);
                    acc[tl][0][1] = mfma32(a0, b1, acc[tl][0][1]);
                    acc[tl][1][0] = mfma32(a1, b0, acc[tl][1][0]);
                    acc[tl][1][1] = mfma32(a1, b1, acc[tl][1][1]);
                }
            }
        }
        __syncthreads();
        if (next < ntiles) write_tile(next);
        __syncthreads();
        tile = next;
    }

    float* out = a.partial + (size_t)blockIdx.x * (KW * 4096 + 64);
    if (TIME_SPLIT) {
        // fixed-order reduction of the four waves' slabs through LDS (reuses the tile buffers)
        float* red = smem;            // KW*4096 floats = 48 KB <= tile buffers (67 KB)
        for (int w = 0; w < 4; ++w) {
            if (wave == w) {
#pragma unroll
                for (int tl = 0; tl < TL; ++tl)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int o = (tl * 64 + mt * 32 + mfma_row(r, half)) * 64 + nt * 32 + l31;
                                red[o] = (w == 0) ? acc[tl][mt][nt][r] : red[o] + acc[tl][mt][nt][r];
                            }
            }
            __syncthreads();
        }
        for (int i = tid; i < KW * 4096; i += 256) out[i] = red[i];
    } else {
#pragma unroll
        for (int tl = 0; tl < TL; ++tl) {
            const int tap = wave + 4 * tl;
            if (tap < KW) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            out[(tap * 64 + mt * 32 + mfma_row(r, half)) * 64 + nt * 32 + l31] = acc[tl][mt][nt][r];
            }
        }
    }
    // bias gradient: rows of this thread are c = idx / QR with idx = tid + i*256 -> (tid>>5) + 8 i;
    // the 32 lanes of a half-wave share the row.
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float v = half_wave_sum(bsum[i]);
        if (l31 == 0) out[KW * 4096 + (tid >> 5) + 8 * i] = v;
    }
}

template <int KW, int GPRO, int XPRO>
int launch_wgrad64(const Wgrad64Args& a, int* grid_out, hipStream_t stream) {
    constexpr int NT = 128;
    constexpr size_t lds = (size_t)(64 * (NT + 4) + 64 * (NT + 8) + 6 * 64) * sizeof(float);
    static wm::DevOnce attr_done;
    auto kern = wgrad64_kernel<KW, GPRO, XPRO>;
    if (!wm::dev_done(attr_done)) {
        WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        wm::dev_mark(attr_done);
    }
    const int ntiles = a.B * ((a.T + NT - 1) / NT);
    const int grid = ntiles < kNumCU ? ntiles : kNumCU;
    *grid_out = grid;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
    WM_CHECK_LAUNCH();
    return 0;
}

// dst layout: mode 0 conv  dW[out][in][KW]   <- G[tap][out][in]
//             mode 1 convT dW[in][out][KW]   <- G[KW-1-k][out][in]
__global__ __launch_bounds__(256) void wgrad64_reduce_kernel(const float* __restrict__ partial, int nparts, int KW, int mode,
                                                             float* __restrict__ dw, float* __restrict__ dbias, int accumulate) {
    // block = 64 outputs x 4 quarters of the slab list; fixed order, fp64: bitwise reproducible and cancellation-safe
    __shared__ double sq[4][64];
    const int il = threadIdx.x & 63, grp = threadIdx.x >> 6, i = blockIdx.x * 64 + il;
    const int stride = KW * 4096 + 64;
    const int per = (nparts + 3) / 4, p0 = grp * per, p1 = min(p0 + per, nparts);
    double s0 = 0.0, s1 = 0.0;
    if (i < stride) {
        int p = p0;
        for (; p + 1 < p1; p += 2) {
            s0 += (double)partial[(size_t)p * stride + i];
            s1 += (double)partial[(size_t)(p + 1) * stride + i];
        }
        if (p < p1) s0 += (double)partial[(size_t)p * stride + i];
    }
    sq[grp][il] = s0 + s1;
    __syncthreads();
    if (grp != 0 || i >= stride) return;
    const float s = (float)((sq[0][il] + sq[1][il]) + (sq[2][il] + sq[3][il]));
    if (i < KW * 4096) {
        const int tap = i / 4096, out = (i / 64) % 64, in = i % 64;
        const int dst = (mode == 0) ? (out * 64 + in) * KW + tap : (in * 64 + out) * KW + (KW - 1 - tap);
        dw[dst] = accumulate ? dw[dst] + s : s;
    } else if (dbias) {
        const int c = i - KW * 4096;
        dbias[c] = accumulate ? dbias[c] + s : s;
    }
}


// ---------------------------------------------------------------------------------------------
// bf16x6 variant of the k3 convolution: fp32-grade results on the bf16 matrix cores.
// Every fp32 operand is split into three bf16 pieces (x = hi + mid + lo exactly to 24 bits) when it is staged
// into LDS; a product a*b is the sum of the six piece products of weight >= 2^-16 (hi*hi, hi*mid, mid*hi, hi*lo,
// lo*hi, mid*mid), accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The dropped terms are <= 2^-24 relative, i.e.
// the result carries the same error as a native fp32 FMA chain, at 6/16 of the matrix-pipe time of
// v_mfma_f32_32x32x2_f32 -- and, unlike the fp32 MFMA, the bf16 MFMA does not share the SIMD's VALU pipe.
// LDS images are channel-minor ([time][channel] and [tap][cout][cin], 144-B pitch) so that an MFMA fragment
// (8 consecutive k = 8 input channels) is one aligned ds_read_b128.
// ---------------------------------------------------------------------------------------------
#ifndef WM_LOADPOS
#define WM_LOADPOS 1
#endif
template <int PRO, int EPI, bool STATS>
__global__ __launch_bounds__(256) void conv64bf_kernel(Conv64Args a) {
    constexpr int KW = 3, PAD = 1, NT = 128, ROWS = NT + 2, PITCH = 72, NP = 3;
    constexpr int NC = 4;                         // (channel pair, time quad) combos per thread
    constexpr bool TWO = (PRO == PRO_BNBWD);
    constexpr bool E1 = (EPI == EPI_RELUMASK || EPI == EPI_ADD || EPI == EPI_BIASADDELU || EPI == EPI_MULDELU);
    constexpr bool HASBIAS = (EPI == EPI_BIAS || EPI == EPI_BIASELU || EPI == EPI_BIASADDELU);
    extern __shared__ __align__(16) unsigned char smem_raw[];
    unsigned short* Wb = reinterpret_cast<unsigned short*>(smem_raw);              // [NP][KW][64 out][PITCH]
    unsigned short* Xb = Wb + NP * KW * 64 * PITCH;                                // [NP][ROWS][PITCH]
    float* Cs = reinterpret_cast<float*>(Xb + NP * ROWS * PITCH);                  // [6][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int T = a.T;
    const int tilesPerClip = (T + NT - 1) / NT, ntiles = a.B * tilesPerClip;

    float4 sa[NC], sb[NC], sa2[TWO ? NC : 1], sb2[TWO ? NC : 1];
    float hl, hl2 = 0.f;
    // staging map: a wave covers 8 channel pairs x 8 time quads (128-B global segments, 2-way LDS write conflicts)
    auto combo = [&](int i, int& cp, int& q) {
        const int idx = tid + i * 256, widx = idx >> 6, l = idx & 63;
        cp = (widx & 3) * 8 + (l & 7);
        q = (widx >> 2) * 8 + (l >> 3);
    };
    auto load_tile = [&](int tile) {              // branch-free: clamped addresses, masked when written to LDS
        const int b = tile / tilesPerClip, t0 = (tile - b * tilesPerClip) * NT;
        const float* xb = a.x + (size_t)b * 64 * T;
        const float* xb2 = TWO ? a.x2 + (size_t)b * 64 * T : nullptr;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            int cp, q;
            combo(i, cp, q);
            const size_t o = (size_t)(2 * cp) * T + min(t0 + 4 * q, T - 4);
            sa[i] = *reinterpret_cast<const float4*>(xb + o);
            sb[i] = *reinterpret_cast<const float4*>(xb + o + T);
            if (TWO) { sa2[i] = *reinterpret_cast<const float4*>(xb2 + o); sb2[i] = *reinterpret_cast<const float4*>(xb2 + o + T); }
        }
        const int hc = (tid & 127) >> 1, hh = tid & 1;
        const int ht = min(max(hh ? t0 + NT : t0 - 1, 0), T - 1);
        hl = xb[(size_t)hc * T + ht];
        if (TWO) hl2 = xb2[(size_t)hc * T + ht];
    };
    auto write_tile = [&](int tile) {
        const int b = tile / tilesPerClip, t0 = (tile - b * tilesPerClip) * NT;
        unsigned* X32 = reinterpret_cast<unsigned*>(Xb);
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            int cp, q;
            combo(i, cp, q);
            const int c = 2 * cp, t = t0 + 4 * q;
            float va[4] = {sa[i].x, sa[i].y, sa[i].z, sa[i].w}, vb[4] = {sb[i].x, sb[i].y, sb[i].z, sb[i].w};
            if (PRO != PRO_NONE) {
                const float ca0 = (PRO == PRO_ADDVEC) ? a.pa[b * 64 + c] : Cs[c], ca1 = (PRO == PRO_ADDVEC) ? a.pa[b * 64 + c + 1] : Cs[c + 1];
                const float cb0 = Cs[64 + c], cb1 = Cs[64 + c + 1], cc0 = Cs[128 + c], cc1 = Cs[128 + c + 1];
                const float cl0 = TWO ? Cs[192 + c] : 0.f, cl1 = TWO ? Cs[192 + c + 1] : 0.f;
                float wa[4] = {0.f, 0.f, 0.f, 0.f}, wb[4] = {0.f, 0.f, 0.f, 0.f};
                if (TWO) { wa[0] = sa2[i].x; wa[1] = sa2[i].y; wa[2] = sa2[i].z; wa[3] = sa2[i].w; wb[0] = sb2[i].x; wb[1] = sb2[i].y; wb[2] = sb2[i].z; wb[3] = sb2[i].w; }
#pragma unroll
                for (int e = 0; e < 4; ++e) { va[e] = pro_apply<PRO>(va[e], wa[e], ca0, cb0, cc0, cl0); vb[e] = pro_apply<PRO>(vb[e], wb[e], ca1, cb1, cc1, cl1); }
            }
            const bool ok = t < T;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned p0, p1, p2;
                split3_pair(ok ? va[e] : 0.f, ok ? vb[e] : 0.f, p0, p1, p2);
                const int o = ((1 + 4 * q + e) * PITCH + c) >> 1;
                X32[o] = p0; X32[(ROWS * PITCH >> 1) + o] = p1; X32[2 * (ROWS * PITCH >> 1) + o] = p2;
            }
        }
        if (tid < 128) {
            const int hc = tid >> 1, hh = tid & 1, t = hh ? t0 + NT : t0 - 1;
            float v = hl;
            if (PRO != PRO_NONE) {
                const float ca = (PRO == PRO_ADDVEC) ? a.pa[b * 64 + hc] : Cs[hc];
                v = pro_apply<PRO>(v, hl2, ca, Cs[64 + hc], Cs[128 + hc], TWO ? Cs[192 + hc] : 0.f);
            }
            if (t < 0 || t >= T) v = 0.f;
            unsigned p0, p1, p2;
            split3_pair(v, 0.f, p0, p1, p2);
            const int o = (hh ? NT + 1 : 0) * PITCH + hc;
            Xb[o] = (unsigned short)p0; Xb[ROWS * PITCH + o] = (unsigned short)p1; Xb[2 * ROWS * PITCH + o] = (unsigned short)p2;
        }
    };

    int tile = blockIdx.x;
    if (tile < ntiles) load_tile(tile);
    // resident weights: global image [NP][KW][64][64] bf16 -> LDS pitch 72, 16 B at a time
    for (int i = tid; i < NP * KW * 64 * 8; i += 256) {
        const int row = i >> 3, seg = i & 7;
        *reinterpret_cast<uint4*>(Wb + row * PITCH + seg * 8) = reinterpret_cast<const uint4*>(a.wp)[i];
    }
    if (tid < 64) {
        Cs[tid] = (PRO == PRO_BNRELU || PRO == PRO_BNBWD) ? a.pa[tid] : 0.f;
        Cs[64 + tid] = (PRO == PRO_BNRELU || PRO == PRO_BNBWD) ? a.pb[tid] : 0.f;
        Cs[128 + tid] = (PRO == PRO_BNBWD) ? a.pc[tid] : 0.f;
        Cs[192 + tid] = (PRO == PRO_BNBWD) ? a.pb[64 + tid] : ((HASBIAS && a.bias) ? a.bias[tid] : 0.f);
        Cs[256 + tid] = (EPI == EPI_RELUMASK) ? a.ea[tid] : 0.f;
        Cs[320 + tid] = (EPI == EPI_RELUMASK) ? a.eb[tid] : 0.f;
    }
    __syncthreads();
    if (tile < ntiles) write_tile(tile);
    __syncthreads();

    float s1[STATS ? 32 : 1], s2[STATS ? 32 : 1];
    if (STATS) {
#pragma unroll
        for (int j = 0; j < 32; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    }
#ifdef WM_STAMP
    unsigned long long tm[6] = {0, 0, 0, 0, 0, 0};
#endif
    while (tile < ntiles) {
        STAMP(ts0);
        const int next = tile + gridDim.x;
        const int b = tile / tilesPerClip, t0 = (tile - b * tilesPerClip) * NT;
        const int tcol = t0 + wave * 32 + l31;
        float e1r[E1 ? 32 : 1];
        // The global loads of the next tile (and of this tile's epilogue operand) are issued LOADPOS taps into the
        // matrix phase: issued at the top they queue behind the previous tile's output stores and the wave stalls
        // at issue until those drain, with the matrix cores idle.
        auto issue_loads = [&]() {
            if (next < ntiles) load_tile(next);
            if (E1) {
                const float* eb1 = a.e1 + (size_t)b * 64 * T + min(tcol, T - 1);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) e1r[mt * 16 + r] = eb1[(size_t)(mt * 32 + mfma_row(r, half)) * T];
            }
        };
        if (WM_LOADPOS == 0) issue_loads();
        STAMP(ts1);
        f32x16 acc[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
        const unsigned short* wbase = Wb + l31 * PITCH + 8 * half;
        const unsigned short* xbase = Xb + (wave * 32 + l31) * PITCH + 8 * half;
#pragma unroll 1
        for (int tap = 0; tap < KW; ++tap) {
            if (WM_LOADPOS > 0 && tap == WM_LOADPOS) issue_loads();
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                bf16x8 A[2][NP], Bf[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    Bf[p] = *reinterpret_cast<const bf16x8*>(xbase + (p * ROWS + tap) * PITCH + 16 * ch);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        A[mt][p] = *reinterpret_cast<const bf16x8*>(wbase + ((p * KW + tap) * 64 + mt * 32) * PITCH + 16 * ch);
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt][1], Bf[1], acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt][0], Bf[2], acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt][2], Bf[0], acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt][0], Bf[1], acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt][1], Bf[0], acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt][0], Bf[0], acc[mt], 0, 0, 0);
                }
            }
        }
        STAMP(ts2);
        float* yb = a.y + (size_t)b * 64 * T + tcol;
        const bool ok = tcol < T;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = mt * 32 + mfma_row(r, half);
                float v = acc[mt][r];
                float q = 0.f;
                if (HASBIAS) v += Cs[192 + co];
                if (EPI == EPI_RELUMASK) { q = e1r[mt * 16 + r]; v = (fmaf(q, Cs[256 + co], Cs[320 + co]) > 0.f) ? v : 0.f; }
                if (EPI == EPI_ADD || EPI == EPI_BIASADDELU) v += e1r[mt * 16 + r];
                if (EPI == EPI_BIASELU || EPI == EPI_BIASADDELU) v = elu1(v);
                if (EPI == EPI_MULDELU) { const float yy = e1r[mt * 16 + r]; v *= (yy > 0.f ? 1.f : yy + 1.f); }
                if (ok) {
                    yb[(size_t)co * T] = v;
                    if (STATS) { s1[mt * 16 + r] += v; s2[mt * 16 + r] += (EPI == EPI_RELUMASK) ? v * q : v * v; }
                }
            }
        STAMP(ts3);
        __syncthreads();
        STAMP(ts4);
        if (next < ntiles) write_tile(next);
        STAMP(ts5);
        __syncthreads();
        STAMP(ts6);
#ifdef WM_STAMP
        tm[0] += ts1 - ts0; tm[1] += ts2 - ts1; tm[2] += ts3 - ts2; tm[3] += ts4 - ts3; tm[4] += ts5 - ts4; tm[5] += ts6 - ts5;
#endif
        tile = next;
    }
#ifdef WM_STAMP
    if (g_wm_stamp && lane == 0) {
        unsigned long long* d = g_wm_stamp + ((size_t)blockIdx.x * 4 + wave) * 6;
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i] = tm[i];
    }
#endif
    if (STATS) {
        float* red = reinterpret_cast<float*>(Xb);          // [4 waves][2][64]
#pragma unroll
        for (int j = 0; j < 32; ++j) { s1[j] = half_wave_sum(s1[j]); s2[j] = half_wave_sum(s2[j]); }
        if (l31 == 0) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = mt * 32 + mfma_row(r, half);
                    red[wave * 128 + co] = s1[mt * 16 + r];
                    red[wave * 128 + 64 + co] = s2[mt * 16 + r];
                }
        }
        __syncthreads();
        if (tid < 128)
            a.stats[(size_t)blockIdx.x * 128 + tid] = red[tid] + red[128 + tid] + red[256 + tid] + red[384 + tid];
    }
}

template <int PRO, int EPI, bool STATS>
int launch_conv64bf(const Conv64Args& a, hipStream_t stream) {
    constexpr size_t lds = (size_t)(3 * 3 * 64 * 72 + 3 * 130 * 72) * 2 + 6 * 64 * sizeof(float);
    static wm::DevOnce attr_done;
    auto kern = conv64bf_kernel<PRO, EPI, STATS>;
    if (!wm::dev_done(attr_done)) {
        WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        wm::dev_mark(attr_done);
    }
    const int ntiles = a.B * ((a.T + 127) / 128);
    const int grid = ntiles < kNumCU ? ntiles : kNumCU;
    if (STATS && grid < kNumCU) WM_TRY(hipMemsetAsync(a.stats, 0, sizeof(float) * 128 * kNumCU, stream));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
    WM_CHECK_LAUNCH();
    return 0;
}


// ---------------------------------------------------------------------------------------------
// bf16x6 convolution, register-resident weights and an interleaved pipeline ("bf3").
// conv64bf_kernel above runs its phases back to back on one wave per SIMD: matrix phase, epilogue, barrier, split +
// LDS write, barrier -- the matrix cores idle through every VALU phase (measured: 5.6 K of 14 K cycles per tile).
// Here wave (mt, nh) keeps the weight fragments of its 32 output rows in registers for the whole kernel
// (12 k-steps x 3 pieces x 4 VGPRs = 144 of the 512 a single resident wave may use), so LDS only carries the input
// image -- small enough (2 x 55 KB) to double-buffer.  While the matrix cores work on tile i out of image A, the
// same instruction stream, a few VALU/DS instructions after every MFMA, splits tile i+1 into image B and then
// issues the global loads of tile i+2: one LDS-only barrier per tile, output stores stay in flight across it.
// ---------------------------------------------------------------------------------------------
#ifndef WM_LOAD_H
#define WM_LOAD_H 17
#endif
#ifndef WM_XCD_MAP
#define WM_XCD_MAP 1
#endif
#ifndef WM_TILE_CONTIG
#define WM_TILE_CONTIG 0
#endif
#ifndef WM_NT_STORE
#define WM_NT_STORE 0
#endif
#ifndef WM_BF3_MANUAL
#define WM_BF3_MANUAL 1
#endif
#ifndef WM_BF3_PIN_W
#define WM_BF3_PIN_W 1
#endif
#ifndef WM_BF7_PIPE
#define WM_BF7_PIPE 1        // 7-tap convolution: 1 = register-resident weights, pipelined (T % 128 == 0), 0 = LDS weights, phase-serial
#endif
#ifndef WM_WGRAD_PIPE
#define WM_WGRAD_PIPE 1      // k3 weight gradient: 1 = pipelined 64-step tiles (wgrad64bfp_kernel), 0 = phase-serial 128-step tiles
#endif
// XCD-aware workgroup -> tile-slot map.  Workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8), each with its
// own L2.  Giving XCD x the 1/8 of the tile sequence [x*G/8, (x+1)*G/8) makes time-adjacent tiles share an L2, so the
// halo columns (a 128-B line per channel per side, +50 % fetched bytes otherwise) and the neighbours' lines hit in L2.
__device__ __forceinline__ int xcd_slot() {
    const int g = gridDim.x, b = blockIdx.x;
    return (g & 7) ? b : (b & 7) * (g >> 3) + (b >> 3);
}

__device__ __forceinline__ void lds_barrier() {          // s_barrier without draining vmcnt (global stores/prefetches stay in flight)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// H = true: the f16 two-piece build (pieces hi = f16(x), lo = f16(x - hi); three products lo hi, hi lo, hi hi on v_mfma_f32_32x32x16_f16;
// weight image from wm_pack_w64_h: w * ws with ws a power of two, the accumulators are multiplied by 1 / ws in the epilogue)
template <int PRO, int EPI, bool STATS, bool H = false>
__global__ __launch_bounds__(256) void conv64bf3_kernel(Conv64Args a) {
    constexpr int KW = 3, NT = 128, ROWS = NT + 2, PITCH = 72, NP = H ? 2 : 3, NC = 4;
    constexpr int XBUF = NP * ROWS * PITCH;               // bf16 elements per input image
    constexpr bool TWO = (PRO == PRO_BNBWD);
    constexpr bool E1 = (EPI == EPI_RELUMASK || EPI == EPI_ADD || EPI == EPI_BNADDRELU);
    extern __shared__ __align__(16) unsigned char smem_raw[];
    unsigned short* Xb0 = reinterpret_cast<unsigned short*>(smem_raw);             // 2 x [NP][ROWS][PITCH]
    float* Cs = reinterpret_cast<float*>(Xb0 + 2 * XBUF);                           // [6][64]
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform values stay in SGPRs (scalar address math)
    const int mt = wave & 1, nh = wave >> 1;
    const int T = a.T;                             // T % 128 == 0 (checked by the launcher): no partial tiles
    const int tilesPerClip = (T + NT - 1) / NT, ntiles = a.B * tilesPerClip;

    // ---- resident weight fragments: packed image [NP][KW][64 out][64 in] bf16
    u32x4 Wr[12][NP];
    {
        const uint4* wg = reinterpret_cast<const uint4*>(a.wp);
#pragma unroll
        for (int s = 0; s < 12; ++s)
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int e = ((p * KW + (s >> 2)) * 64 + 32 * mt + l31) * 64 + 16 * (s & 3) + 8 * half;
                u32x4 w_ = __builtin_bit_cast(u32x4, wg[e >> 3]);
#if WM_BF3_PIN_W
                // pinned to the AGPR half of the register file (the MFMA reads its A operand from there directly); left to the
                // allocator, the fragments beyond the VGPR budget are copied back (v_accvgpr_read) in front of their k-step
                asm volatile("" : "+a"(w_));
#endif
                Wr[s][p] = w_;
            }
    }
    const float winv = H ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(a.wp) + NP * KW * 4096)[1] : 1.f;
    const float wsc = H ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(a.wp) + NP * KW * 4096)[0] : 1.f;
    // ---- staging map (fixed per thread): channel pair cp = 8*wave + (lane & 7), time quads q = 8*i + (lane >> 3)
    const int cp = wave * 8 + (lane & 7), c0 = 2 * cp, q0 = lane >> 3;
    const int hc = (tid & 127) >> 1, hh = tid & 1;
    float4 sa[NC], sb[NC], sa2[TWO ? NC : 1], sb2[TWO ? NC : 1];
    float hl, hl2 = 0.f;
    // global -> register staging of one tile, in five pieces (4 combos + halo) so that the main loop can issue them
    // a few at a time: a 32-KB burst per CU backs up the memory pipeline and blocks the wave at issue for ~2.5 K cycles
    auto load_combo = [&](int tile, int i) {
        const int b = tile / tilesPerClip, t0 = (tile - b * tilesPerClip) * NT;
        const size_t o = ((size_t)b * 64 + c0) * T + t0 + 4 * (q0 + 8 * i);
        sa[i] = *reinterpret_cast<const float4*>(a.x + o);
        sb[i] = *reinterpret_cast<const float4*>(a.x + o + T);
        if (TWO) { sa2[i] = *reinterpret_cast<const float4*>(a.x2 + o); sb2[i] = *reinterpret_cast<const float4*>(a.x2 + o + T); }
    };
    auto load_halo = [&](int tile) {              // branch-free: clamped address, masked when written to LDS
        const int b = tile / tilesPerClip, t0 = (tile - b * tilesPerClip) * NT;
        const int ht = min(max(hh ? t0 + NT : t0 - 1, 0), T - 1);
        hl = a.x[((size_t)b * 64 + hc) * T + ht];
        if (TWO) hl2 = a.x2[((size_t)b * 64 + hc) * T + ht];
    };
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int i = 0; i < NC; ++i) load_combo(tile, i);
        load_halo(tile);
    };

#if WM_TILE_CONTIG
    const int chunk = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x, tstep = 1;
    int tile = blockIdx.x * chunk;
    const int tend = min(tile + chunk, ntiles);
#else
    const int tstep = gridDim.x, tend = ntiles;
    int tile = WM_XCD_MAP ? xcd_slot() : (int)blockIdx.x;        // grid <= ntiles
#endif
    load_tile(min(tile, ntiles - 1));
    if (tid < 64) {
        Cs[tid] = (PRO == PRO_BNRELU || PRO == PRO_BNBWD) ? a.pa[tid] : 0.f;
        Cs[64 + tid] = (PRO == PRO_BNRELU || PRO == PRO_BNBWD) ? a.pb[tid] : 0.f;
        Cs[128 + tid] = (PRO == PRO_BNBWD) ? a.pc[tid] : 0.f;
        Cs[192 + tid] = (PRO == PRO_BNBWD) ? a.pb[64 + tid] : (((EPI == EPI_BIAS || EPI == EPI_BNADDRELU) && a.bias) ? a.bias[tid] : 0.f);
        Cs[256 + tid] = (EPI == EPI_RELUMASK || EPI == EPI_BNADDRELU) ? a.ea[tid] : 0.f;
        Cs[320 + tid] = (EPI == EPI_RELUMASK || EPI == EPI_BNADDRELU) ? a.eb[tid] : 0.f;
    }
    __syncthreads();
    // per-thread prologue constants (the staging channels never change)
    const float ka0 = Cs[c0], ka1 = Cs[c0 + 1], kb0 = Cs[64 + c0], kb1 = Cs[64 + c0 + 1];
    const float kc0 = Cs[128 + c0], kc1 = Cs[128 + c0 + 1], kl0 = TWO ? Cs[192 + c0] : 0.f, kl1 = TWO ? Cs[192 + c0 + 1] : 0.f;
    const float ha = Cs[hc], hb = Cs[64 + hc], hcc = Cs[128 + hc], hlo = TWO ? Cs[192 + hc] : 0.f;

    // one (combo i, element e) unit of the split: two channels x one time step -> 3 dwords
    auto split_unit = [&](unsigned* X32, int t0, int i, int e) {
        const float4 fa = sa[i], fb = sb[i];
        float va = (e == 0) ? fa.x : (e == 1) ? fa.y : (e == 2) ? fa.z : fa.w;
        float vb = (e == 0) ? fb.x : (e == 1) ? fb.y : (e == 2) ? fb.z : fb.w;
        if (PRO != PRO_NONE) {
            float wa = 0.f, wb = 0.f;
            if (TWO) {
                const float4 ga = sa2[i], gb = sb2[i];
                wa = (e == 0) ? ga.x : (e == 1) ? ga.y : (e == 2) ? ga.z : ga.w;
                wb = (e == 0) ? gb.x : (e == 1) ? gb.y : (e == 2) ? gb.z : gb.w;
            }
            va = pro_apply<PRO>(va, wa, ka0, kb0, kc0, kl0);
            vb = pro_apply<PRO>(vb, wb, ka1, kb1, kc1, kl1);
        }
        const int o = (1 + 4 * (q0 + 8 * i) + e) * (PITCH / 2) + cp;
        if (H) {
            const h16x2 h_ = __builtin_convertvector(f32x2{va, vb}, h16x2);
            const h16x2 l_ = __builtin_convertvector(f32x2{va - (float)h_.x, vb - (float)h_.y}, h16x2);
            X32[o] = __builtin_bit_cast(unsigned, h_); X32[(ROWS * PITCH >> 1) + o] = __builtin_bit_cast(unsigned, l_);
        } else {
            unsigned p0, p1, p2;
            split3_pair(va, vb, p0, p1, p2);
            X32[o] = p0; X32[(ROWS * PITCH >> 1) + o] = p1; X32[2 * (ROWS * PITCH >> 1) + o] = p2;
        }
    };
    auto split_halo = [&](unsigned short* X, int t0) {
        {                                             // threads 128..255 repeat the writes of 0..127 (no branch)
            const int t = hh ? t0 + NT : t0 - 1;
            float v = hl;
            if (PRO != PRO_NONE) v = pro_apply<PRO>(v, hl2, ha, hb, hcc, hlo);
            if (t < 0 || t >= T) v = 0.f;
            const int o = (hh ? NT + 1 : 0) * PITCH + hc;
            if (H) {
                const _Float16 h_ = (_Float16)v;
                const _Float16 l_ = (_Float16)(v - (float)h_);
                X[o] = __builtin_bit_cast(unsigned short, h_); X[ROWS * PITCH + o] = __builtin_bit_cast(unsigned short, l_);
            } else {
                unsigned p0, p1, p2;
                split3_pair(v, 0.f, p0, p1, p2);
                X[o] = (unsigned short)p0; X[ROWS * PITCH + o] = (unsigned short)p1; X[2 * ROWS * PITCH + o] = (unsigned short)p2;
            }
        }
    };
    // the same split in stages, one per scheduling slice of the main loop (WM_BF3_MANUAL)
    float sva = 0.f, svb = 0.f;
    unsigned sp0 = 0, sp1 = 0;
    auto split_pick = [&](int i, int e) {
        const float4 fa = sa[i], fb = sb[i];
        float va = (e == 0) ? fa.x : (e == 1) ? fa.y : (e == 2) ? fa.z : fa.w;
        float vb = (e == 0) ? fb.x : (e == 1) ? fb.y : (e == 2) ? fb.z : fb.w;
        if (PRO != PRO_NONE) {
            float wa = 0.f, wb = 0.f;
            if (TWO) {
                const float4 ga = sa2[i], gb = sb2[i];
                wa = (e == 0) ? ga.x : (e == 1) ? ga.y : (e == 2) ? ga.z : ga.w;
                wb = (e == 0) ? gb.x : (e == 1) ? gb.y : (e == 2) ? gb.z : gb.w;
            }
            va = pro_apply<PRO>(va, wa, ka0, kb0, kc0, kl0);
            vb = pro_apply<PRO>(vb, wb, ka1, kb1, kc1, kl1);
        }
        sva = va; svb = vb;
        asm volatile("" : "+v"(sva), "+v"(svb));
    };
    auto split_st1 = [&]() {
        if (H) {
            const h16x2 hh_ = __builtin_convertvector(f32x2{sva, svb}, h16x2);
            sp0 = __builtin_bit_cast(unsigned, hh_);
            sva -= (float)hh_.x; svb -= (float)hh_.y;
        } else {
            const bf16x2 hh_ = {(__bf16)sva, (__bf16)svb};
            sp0 = __builtin_bit_cast(unsigned, hh_);
            sva -= __uint_as_float(sp0 << 16); svb -= __uint_as_float(sp0 & 0xffff0000u);
        }
        asm volatile("" : "+v"(sva), "+v"(svb), "+v"(sp0));
    };
    auto split_st2 = [&]() {
        if (H) return;                                   // two pieces: no middle one
        const bf16x2 mm_ = {(__bf16)sva, (__bf16)svb};
        sp1 = __builtin_bit_cast(unsigned, mm_);
        sva -= __uint_as_float(sp1 << 16); svb -= __uint_as_float(sp1 & 0xffff0000u);
        asm volatile("" : "+v"(sva), "+v"(svb), "+v"(sp1));
    };
    auto last_piece = [&]() -> unsigned {
        if (H) return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{sva, svb}, h16x2));
        const bf16x2 ll_ = {(__bf16)sva, (__bf16)svb};
        return __builtin_bit_cast(unsigned, ll_);
    };
    auto split_out = [&](unsigned* X32, int i, int e) {
        const unsigned sp2 = last_piece();
        const int o = (1 + 4 * (q0 + 8 * i) + e) * (PITCH / 2) + cp;
        X32[o] = sp0; if (!H) X32[(ROWS * PITCH >> 1) + o] = sp1; X32[(NP - 1) * (ROWS * PITCH >> 1) + o] = sp2;
    };
    auto halo_pick = [&](int t0) {
        const int t = hh ? t0 + NT : t0 - 1;
        float v = hl;
        if (PRO != PRO_NONE) v = pro_apply<PRO>(v, hl2, ha, hb, hcc, hlo);
        if (t < 0 || t >= T) v = 0.f;
        sva = v; svb = 0.f;
        asm volatile("" : "+v"(sva), "+v"(svb));
    };
    auto halo_out = [&](unsigned short* X) {
        const unsigned sp2 = last_piece();
        const int o = (hh ? NT + 1 : 0) * PITCH + hc;
        X[o] = (unsigned short)sp0; if (!H) X[ROWS * PITCH + o] = (unsigned short)sp1; X[(NP - 1) * ROWS * PITCH + o] = (unsigned short)sp2;
    };
    {
        const int t0 = (min(tile, ntiles - 1) % tilesPerClip) * NT;
#pragma unroll
        for (int u = 0; u < 16; ++u) split_unit(reinterpret_cast<unsigned*>(Xb0), t0, u >> 2, u & 3);
        split_halo(Xb0, t0);
    }
    load_tile(min(tile + tstep, ntiles - 1));
    __syncthreads();
    // nothing pending on entry: the loop's vmcnt waits are then derived from its own (steady-state) issue order only
    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0)

    float s1[STATS ? 16 : 1], s2[STATS ? 16 : 1];
    if (STATS) {
#pragma unroll
        for (int j = 0; j < 16; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    }
    int buf = 0;
#ifdef WM_STAMP
    unsigned long long tm[6] = {0, 0, 0, 0, 0, 0};
#endif
    // The epilogue of a tile is deferred into the matrix phase of the NEXT tile (accp = its accumulators): the
    // output stores trickle out between MFMAs instead of arriving as one 8-MB burst from all 256 CUs at once.
    f32x16 accp[2];
    // "previous tile" of the first iteration = the first tile itself with accumulators that make its epilogue values ZERO (minus
    // the bias where one is added): they are stored (and overwritten one iteration later by the same lanes, in program order) and
    // add nothing to the sums -- no flag, no extra multiply per value
    int pb_ = min(tile, ntiles - 1) / tilesPerClip, pt0 = (min(tile, ntiles - 1) % tilesPerClip) * NT;
    bool any_tile = false;
    float e1r[E1 ? 32 : 1];
    // per-row epilogue constants of this lane's 16 accumulator rows, in registers: an LDS read inside an epilogue slice stalls
    // the wave for the whole LDS latency in front of the next MFMA
    constexpr bool KB = (EPI == EPI_BIAS || EPI == EPI_BNADDRELU), KE = (EPI == EPI_RELUMASK || EPI == EPI_BNADDRELU);
    float kbias[KB ? 16 : 1], kea[KE ? 16 : 1], keb[KE ? 16 : 1];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = 32 * mt + mfma_row(r, half);
        if (KB) kbias[r] = Cs[192 + co];
        if (KE) { kea[r] = Cs[256 + co]; keb[r] = Cs[320 + co]; }
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) accp[nt][r] = (EPI == EPI_BIAS) ? -kbias[r] * wsc : 0.f;       // wsc: a power of two (exact)
    // output / epilogue-operand addressing through buffer descriptors over the wave's 32 channel rows of a clip: per-lane offset
    // = column, scalar offset = row (sixteen loop-invariant scalars), the column block nt as the immediate -- one instruction per
    // value instead of a 64-bit scalar add, a 64-bit vector add and the access
    const unsigned lcol = (unsigned)(4 * half * T + l31) * 4u;
    unsigned rowT[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rowT[r] = (unsigned)(((r & 3) + 8 * (r >> 2)) * T) * 4u;
    auto row_srd = [&](const float* base, int cb) { return make_srd(base + ((size_t)cb * 64 + 32 * mt) * T, (size_t)32 * T * sizeof(float)); };
    wm_srd_t syp = row_srd(a.y, pb_);                         // output rows of the PREVIOUS tile's clip
    unsigned vcolp = lcol + (unsigned)(pt0 + 64 * nh) * 4u;
    if (E1) {
#pragma unroll
        for (int j = 0; j < 32; ++j) e1r[j] = 0.f;
    }
    // epilogue of value idx = nt * 16 + r of the previous tile; se1c / vcolc = the tile being computed now, whose epilogue
    // operand replaces the consumed one in the same register (it is needed one full iteration from now)
    auto epi_value = [&](int idx, bool fetch, const wm_srd_t& se1c, unsigned vcolc) {
        const int nt = idx >> 4, r = idx & 15;
        float v = accp[nt][r];
        float q = 0.f;
        if (H) v *= winv;                               // exact (power of two); contracts with the bias add
        if (EPI == EPI_BIAS) v += kbias[r];
        if (EPI == EPI_RELUMASK) { q = e1r[idx]; v = (fmaf(q, kea[r], keb[r]) > 0.f) ? v : 0.f; }
        if (EPI == EPI_ADD) v += e1r[idx];
        if (EPI == EPI_BNADDRELU) v = fmaxf(e1r[idx] + fmaf(v + kbias[r], kea[r], keb[r]), 0.f);   // = wm_bn_add_relu of the biased conv
        buf_store(syp, v, vcolp + 128u * nt, rowT[r]);
        if (STATS) { s1[r] += v; s2[r] = fmaf(v, (EPI == EPI_RELUMASK) ? q : v, s2[r]); }
#if WM_BF3_MANUAL
        if (STATS) asm volatile("" : "+v"(s1[r]), "+v"(s2[r]));
#endif
        if (E1 && fetch) e1r[idx] = buf_load(se1c, vcolc + 128u * nt, rowT[r]);
    };
    auto mma = [&](const u32x4& A_, const u32x4& B_, f32x16 c) -> f32x16 {
        if (H) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, A_), __builtin_bit_cast(h16x8, B_), c, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A_), __builtin_bit_cast(bf16x8, B_), c, 0, 0, 0);
    };
    while (tile < tend) {
        STAMP(ts0);
        // tiles past the end are clamped: their (valid) data lands in the image nobody reads again
        const int next = min(tile + tstep, ntiles - 1), next2 = min(tile + 2 * tstep, ntiles - 1);
        const int b = tile / tilesPerClip, t0 = (tile - b * tilesPerClip) * NT;
        const int nt0 = (next % tilesPerClip) * NT;
        const wm_srd_t se1c = E1 ? row_srd(a.e1, b) : syp;
        const unsigned vcolc = lcol + (unsigned)(t0 + 64 * nh) * 4u;
        const unsigned short* xcur = Xb0 + buf * XBUF;
        unsigned short* xnxt = Xb0 + (buf ^ 1) * XBUF;
        f32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        const unsigned short* xrow = xcur + (64 * nh + l31) * PITCH + 8 * half;
        u32x4 Bq[2][NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) Bq[0][p] = *reinterpret_cast<const u32x4*>(xrow + p * ROWS * PITCH);
#if WM_BF3_MANUAL
        // Hand-pinned schedule: every MFMA is followed by one slice of the side work (sched_barrier fences; the arithmetic of a
        // slice is tied to it by an empty asm on its results).  The six MFMAs of a half-step chain on one accumulator, each
        // waiting 8 passes for its predecessor, so a slice of <= 8 VALU instructions between two of them costs nothing.  The
        // group-barrier pattern of the other branch leaves lumps of 30-40 instructions in front of back-to-back MFMAs.
#define FENCE __builtin_amdgcn_sched_barrier(0)
#define BF3_SLICE(k)                                                                                                        \
    {                                                                                                                           \
        if (h < 16) {                                                                                                           \
            if ((k) == 0) split_pick(h >> 2, h & 3);                                                                            \
            if ((k) == 1) split_st1();                                                                                          \
            if ((k) == 2) split_st2();                                                                                          \
            if ((k) == 3) { split_out(reinterpret_cast<unsigned*>(xnxt), h >> 2, h & 3); if ((h & 3) == 3) load_combo(next2, h >> 2); } \
        } else if (h == 16) {                                                                                                   \
            if ((k) == 0) halo_pick(nt0);                                                                                       \
            if ((k) == 1) split_st1();                                                                                          \
            if ((k) == 2) split_st2();                                                                                          \
            if ((k) == 3) halo_out(xnxt);                                                                                       \
        } else if (h == 17) {                                                                                                   \
            if ((k) == 0) load_halo(next2);                                                                                     \
        }                                                                                                                       \
        if (h < 8) {                                                                                                            \
            if ((k) == 4) epi_value(2 * h, true, se1c, vcolc);                                                                              \
            if ((k) == 5) epi_value(2 * h + 1, true, se1c, vcolc);                                                                          \
        } else if ((k) == 4) epi_value(8 + h, true, se1c, vcolc);                                                                           \
    }
#pragma unroll
        for (int h = 0; h < 24; ++h) {
            const int s = h >> 1, nt = h & 1;
            if (h + 1 < 24) {
                const int s1_ = (h + 1) >> 1, n1 = (h + 1) & 1;
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    Bq[(h + 1) & 1][p] = *reinterpret_cast<const u32x4*>(xrow + (p * ROWS + 32 * n1 + (s1_ >> 2)) * PITCH + 16 * (s1_ & 3));
            }
            const u32x4* Bf = Bq[h & 1];
            FENCE;
            if (H) {                                     // lo hi, hi lo, hi hi: two slices behind every MFMA
                acc[nt] = mma(Wr[s][1], Bf[0], acc[nt]); FENCE; BF3_SLICE(0) FENCE; BF3_SLICE(1) FENCE;
                acc[nt] = mma(Wr[s][0], Bf[1], acc[nt]); FENCE; BF3_SLICE(3) FENCE; BF3_SLICE(4) FENCE;
                acc[nt] = mma(Wr[s][0], Bf[0], acc[nt]); FENCE; BF3_SLICE(5) FENCE;
            } else {
                acc[nt] = mma(Wr[s][1], Bf[1], acc[nt]); FENCE; BF3_SLICE(0) FENCE;
                acc[nt] = mma(Wr[s][0], Bf[2], acc[nt]); FENCE; BF3_SLICE(1) FENCE;
                acc[nt] = mma(Wr[s][2], Bf[0], acc[nt]); FENCE; BF3_SLICE(2) FENCE;
                acc[nt] = mma(Wr[s][0], Bf[1], acc[nt]); FENCE; BF3_SLICE(3) FENCE;
                acc[nt] = mma(Wr[s][1], Bf[0], acc[nt]); FENCE; BF3_SLICE(4) FENCE;
                acc[nt] = mma(Wr[s][0], Bf[0], acc[nt]); FENCE; BF3_SLICE(5) FENCE;
            }
#ifdef WM_STAMP
#ifdef WM_STAMP_FINE
            if (h == 16) { STAMP(tsa); tm[0] += tsa - ts0; }
            if (h == 17) { STAMP(tsb); tm[1] += tsb - ts0; }
            if (h == 18) { STAMP(tsb); tm[3] += tsb - ts0; }
            if (h == 20) { STAMP(tsb); tm[5] += tsb - ts0; }
#else
            if (h == 3) { STAMP(tsa); tm[0] += tsa - ts0; }
            if (h == 15) { STAMP(tsb); tm[1] += tsb - ts0; }
#endif
#endif
        }
#undef FENCE
#undef BF3_SLICE
#else
#pragma unroll
        for (int h = 0; h < 24; ++h) {
            const int s = h >> 1, nt = h & 1;
            if (h + 1 < 24) {
                const int s1_ = (h + 1) >> 1, n1 = (h + 1) & 1;
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    Bq[(h + 1) & 1][p] = *reinterpret_cast<const u32x4*>(xrow + (p * ROWS + 32 * n1 + (s1_ >> 2)) * PITCH + 16 * (s1_ & 3));
            }
            const u32x4* Bf = Bq[h & 1];
            static_assert(!H, "the f16 build exists in the hand-pinned schedule only");
            acc[nt] = mma(Wr[s][1], Bf[1], acc[nt]);
            acc[nt] = mma(Wr[s][0], Bf[2], acc[nt]);
            acc[nt] = mma(Wr[s][2], Bf[0], acc[nt]);
            acc[nt] = mma(Wr[s][0], Bf[1], acc[nt]);
            acc[nt] = mma(Wr[s][1], Bf[0], acc[nt]);
            acc[nt] = mma(Wr[s][0], Bf[0], acc[nt]);
            // side work of this half-step
            if (h < 16) {
                split_unit(reinterpret_cast<unsigned*>(xnxt), nt0, h >> 2, h & 3);
                if ((h & 3) == 3) load_combo(next2, h >> 2);          // this combo's registers are free again
            } else if (h == 16) split_halo(xnxt, nt0);
            else if (h == 17) load_halo(next2);
            if (h < 8) { epi_value(2 * h, true, se1c, vcolc); epi_value(2 * h + 1, true, se1c, vcolc); }
            else epi_value(8 + h, true, se1c, vcolc);
            // interleave: after the fragment reads, one MFMA then a handful of the side instructions, six times
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
            for (int kk = 0; kk < 6; ++kk) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#ifdef WM_STAMP
#ifdef WM_STAMP_FINE
            if (h == 16) { STAMP(tsa); tm[0] += tsa - ts0; }
            if (h == 17) { STAMP(tsb); tm[1] += tsb - ts0; }
            if (h == 18) { STAMP(tsb); tm[3] += tsb - ts0; }
            if (h == 20) { STAMP(tsb); tm[5] += tsb - ts0; }
#else
            if (h == 3) { STAMP(tsa); tm[0] += tsa - ts0; }
            if (h == 15) { STAMP(tsb); tm[1] += tsb - ts0; }
#endif
#endif
        }
#endif
        STAMP(ts1);
        accp[0] = acc[0]; accp[1] = acc[1];
        pb_ = b; pt0 = t0; any_tile = true;
        syp = row_srd(a.y, b); vcolp = vcolc;
        STAMP(ts2);
        lds_barrier();
        STAMP(ts3);
#ifdef WM_STAMP
#ifdef WM_STAMP_FINE
        tm[2] += ts1 - ts0; tm[4] += ts3 - ts2;
#else
        tm[2] += ts1 - ts0; tm[3] += ts2 - ts1; tm[4] += ts3 - ts2;
#endif
#endif
        tile += tstep;
        buf ^= 1;
    }
#ifdef WM_STAMP
    if (g_wm_stamp && lane == 0) {
        unsigned long long* d = g_wm_stamp + ((size_t)blockIdx.x * 4 + wave) * 6;
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i] = tm[i];
    }
#endif
    if (any_tile) {                                  // flush: the last tile's epilogue (its operand is already in e1r)
#pragma unroll
        for (int idx = 0; idx < 32; ++idx) epi_value(idx, false, syp, 0u);
    }
    if (STATS) {
        float* red = reinterpret_cast<float*>(Xb0);          // [2 column halves][2][64]
#pragma unroll
        for (int j = 0; j < 16; ++j) { s1[j] = half_wave_sum(s1[j]); s2[j] = half_wave_sum(s2[j]); }
        __syncthreads();
        if (l31 == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = 32 * mt + mfma_row(r, half);
                red[nh * 128 + co] = s1[r];
                red[nh * 128 + 64 + co] = s2[r];
            }
        }
        __syncthreads();
        if (tid < 128) a.stats[(size_t)blockIdx.x * 128 + tid] = red[tid] + red[128 + tid];
    }
}

template <int PRO, int EPI, bool STATS, bool H = false>
int launch_conv64bf3(const Conv64Args& a, hipStream_t stream) {
    constexpr size_t lds = (size_t)(2 * (H ? 2 : 3) * 130 * 72) * 2 + 6 * 64 * sizeof(float);
    static wm::DevOnce attr_done;
    auto kern = conv64bf3_kernel<PRO, EPI, STATS, H>;
    if (!wm::dev_done(attr_done)) {
        WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        wm::dev_mark(attr_done);
    }
    const int ntiles = a.B * ((a.T + 127) / 128);
    const int grid = ntiles < kNumCU ? ntiles : kNumCU;
    if (STATS && grid < kNumCU) WM_TRY(hipMemsetAsync(a.stats, 0, sizeof(float) * 128 * kNumCU, stream));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
    WM_CHECK_LAUNCH();
    return 0;
}


// ---------------------------------------------------------------------------------------------
// Inference ResBlock in ONE launch:  out = relu(x + BN2(conv2(relu(BN1(conv1(x))))))  with both BatchNorms in eval
// mode (running statistics folded into per-channel scale / shift), py/main16.py:112-125.  Frame passes over HBM: x in,
// out out -- the intermediate never leaves the CU.
// A workgroup walks 124-column output tiles.  Its x window is the 128 columns [w0, w0+128), w0 = 124 k - 4: exactly 32
// aligned float4 per channel, no halo loads.  conv1 runs over MFMA columns j = 0..127 (a1 at time w0 + 1 + j, valid for
// j < 126), its epilogue applies BN1 + ReLU (and the zero padding of conv2: a1 = 0 outside the clip), splits the result
// into its three bf16 pieces and writes them to a second LDS image; conv2 reads that image (output column i = time
// w0 + 2 + i, valid for i < 124) and its epilogue adds BN2 and the residual x (re-read from L2 in accumulator layout) and
// stores.  Both convolutions keep the weight fragments of the wave's 32 output rows in registers (2 x 144 VGPRs of the
// 512 a lone wave may use), so LDS holds only the two activation images (2 x 55 KB).  While conv2 runs, the same
// instruction stream splits the NEXT tile's x window into the (by then free) x image; while conv1 runs, it fetches the
// residual operand.
// ---------------------------------------------------------------------------------------------
struct RbeArgs {
    const float* x;       // [B,64,T]
    const void* w1;       // packed bf16 image [3 pieces][3 taps][64 out][64 in] of w * sc[out] (wm_pack_w64_bf_scaled)
    const void* w2;
    const float* b1; const float* sc1; const float* sh1;     // conv bias, folded BN scale / shift
    const float* b2; const float* sc2; const float* sh2;
    float* y;             // [B,64,T]
    int B, T;
};

// H = true: f16 two-piece build (three products per product): w1 / w2 = wm_pack_w64_h_scaled images (w * sc[out] * ws, {ws, 1 / ws} behind
// the image), x and the intermediate a1 split into hi / lo f16 pieces unscaled, accumulators times 1 / ws in both epilogues
template <bool H>
__global__ __launch_bounds__(256) void resblock_eval_kernel(RbeArgs a) {
    constexpr int KW = 3, NTO = 124, ROWS = 130, PITCH = 72, NP = H ? 2 : 3, NC = 4;
    constexpr int XBUF = NP * ROWS * PITCH;               // bf16 elements per image
    extern __shared__ __align__(16) unsigned char smem_raw[];
    unsigned short* Xb = reinterpret_cast<unsigned short*>(smem_raw);              // x window: row r = time w0 + r
    unsigned short* Ab = Xb + XBUF;                                                // a1: row j = time w0 + 1 + j
    float* Cs = reinterpret_cast<float*>(Ab + XBUF);                               // [2][64]: k1 k2 (offsets; the scales ride in the weights)
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = wave & 1, nh = wave >> 1;
    const int T = a.T;
    const int tilesPerClip = (T + 2 + NTO - 1) / NTO, ntiles = a.B * tilesPerClip;

    u32x4 W1[12][NP], W2[12][NP];
    {
        const uint4* wg1 = reinterpret_cast<const uint4*>(a.w1);
        const uint4* wg2 = reinterpret_cast<const uint4*>(a.w2);
#pragma unroll
        for (int s = 0; s < 12; ++s)
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int e = ((p * KW + (s >> 2)) * 64 + 32 * mt + l31) * 64 + 16 * (s & 3) + 8 * half;
                // the second convolution's fragments are pinned into AGPRs (the MFMA reads its A operand from there in place):
                // 288 fragment registers do not fit the 256 VGPRs, and fragments the allocator spills are copied back per use
                u32x4 w2_ = __builtin_bit_cast(u32x4, wg2[e >> 3]);
                asm volatile("" : "+a"(w2_));
                W1[s][p] = __builtin_bit_cast(u32x4, wg1[e >> 3]);
                W2[s][p] = w2_;
            }
    }
    const float winv1 = H ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(a.w1) + NP * KW * 4096)[1] : 1.f;
    const float winv2 = H ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(a.w2) + NP * KW * 4096)[1] : 1.f;
    // staging map (fixed per thread): channel pair cp, time quads q0 + 8 i
    const int cp = wave * 8 + (lane & 7), c0 = 2 * cp, q0 = lane >> 3;
    float4 sa[NC], sb[NC];
    auto load_combo = [&](int tile, int i) {      // branch-free: clamped address, masked when written to LDS
        const int b = tile / tilesPerClip, w0 = (tile - b * tilesPerClip) * NTO - 4;
        const int t = min(max(w0 + 4 * (q0 + 8 * i), 0), T - 4);
        const size_t o = ((size_t)b * 64 + c0) * T + t;
        sa[i] = *reinterpret_cast<const float4*>(a.x + o);
        sb[i] = *reinterpret_cast<const float4*>(a.x + o + T);
    };
    auto split_unit = [&](int w0, int i, int e) {  // two channels x one time step of the window starting at w0 -> 3 dwords
        const int t = w0 + 4 * (q0 + 8 * i);
        const bool ok = (t >= 0) && (t < T);        // T % 4 == 0 and w0 % 4 == 0: a quad is inside or outside as a whole
        const float4 fa = sa[i], fb = sb[i];
        float va = (e == 0) ? fa.x : (e == 1) ? fa.y : (e == 2) ? fa.z : fa.w;
        float vb = (e == 0) ? fb.x : (e == 1) ? fb.y : (e == 2) ? fb.z : fb.w;
        va = ok ? va : 0.f; vb = ok ? vb : 0.f;
        unsigned* X32 = reinterpret_cast<unsigned*>(Xb);
        const int o = (4 * (q0 + 8 * i) + e) * (PITCH / 2) + cp;
        if (H) {
            const h16x2 h_ = __builtin_convertvector(f32x2{va, vb}, h16x2);
            const h16x2 l_ = __builtin_convertvector(f32x2{va - (float)h_.x, vb - (float)h_.y}, h16x2);
            X32[o] = __builtin_bit_cast(unsigned, h_); X32[(ROWS * PITCH >> 1) + o] = __builtin_bit_cast(unsigned, l_);
        } else {
            unsigned p0, p1, p2;
            split3_pair(va, vb, p0, p1, p2);
            X32[o] = p0; X32[(ROWS * PITCH >> 1) + o] = p1; X32[2 * (ROWS * PITCH >> 1) + o] = p2;
        }
    };

    const int tstep = gridDim.x;
    int tile = xcd_slot();                                         // grid <= ntiles
#pragma unroll
    for (int i = 0; i < NC; ++i) load_combo(tile, i);
    if (tid < 64) {
        Cs[tid] = fmaf(a.b1 ? a.b1[tid] : 0.f, a.sc1[tid], a.sh1[tid]);
        Cs[64 + tid] = fmaf(a.b2 ? a.b2[tid] : 0.f, a.sc2[tid], a.sh2[tid]);
    }
    // rows 128, 129 of both images are read by the discarded MFMA columns only: keep them finite (zero)
    for (int i = tid; i < NP * 2 * (PITCH / 2); i += 256) {
        const int p = i / (2 * (PITCH / 2)), rem = i - p * (2 * (PITCH / 2));
        const int o = (p * ROWS + 128) * (PITCH / 2) + rem;
        reinterpret_cast<unsigned*>(Xb)[o] = 0u;
        reinterpret_cast<unsigned*>(Ab)[o] = 0u;
    }
    {
        const int w0 = (tile % tilesPerClip) * NTO - 4;
#pragma unroll
        for (int u = 0; u < 16; ++u) split_unit(w0, u >> 2, u & 3);
    }
    __syncthreads();
    // per-lane epilogue offsets: the 16 channels of this lane's accumulator rows (the same for both column blocks)
    float k1[16], k2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = 32 * mt + mfma_row(r, half);
        k1[r] = Cs[co]; k2[r] = Cs[64 + co];
    }

    float e1r[32];
#ifdef WM_STAMP
    unsigned long long tm[6] = {0, 0, 0, 0, 0, 0};
#endif
    unsigned* const X32 = reinterpret_cast<unsigned*>(Xb);
    unsigned* const A32 = reinterpret_cast<unsigned*>(Ab);

    // One k-step of a matrix phase: the fragment reads of the NEXT step, then the six piece products with one slice of side
    // work (f0..f5: a handful of VALU / LDS / global instructions) pinned behind each of them.  The six MFMAs of a step
    // chain on one accumulator, so each waits 8 passes for its predecessor: the slice between two of them is free.  The
    // order is fixed by hand (sched_barrier fences): the group-barrier pattern conv64bf3 uses is not honoured here (the
    // weight fragments of two convolutions overflow into AGPRs and their copies break the pattern; the side work then lands
    // in one lump in front of six back-to-back MFMAs).
#define FENCE __builtin_amdgcn_sched_barrier(0)
    auto nop = []() {};
    auto mma = [&](const u32x4& A_, const u32x4& B_, f32x16 c) -> f32x16 {
        if (H) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, A_), __builtin_bit_cast(h16x8, B_), c, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A_), __builtin_bit_cast(bf16x8, B_), c, 0, 0, 0);
    };
    auto kstep = [&](const u32x4 (&W)[12][NP], const unsigned short* img, f32x16& ac, u32x4 (&Bq)[2][NP], int nt, int s,
                     auto&& f0, auto&& f1, auto&& f2, auto&& f3, auto&& f4, auto&& f5) __attribute__((always_inline)) {
        if (s + 1 < 12) {
#pragma unroll
            for (int p = 0; p < NP; ++p)
                Bq[(s + 1) & 1][p] = *reinterpret_cast<const u32x4*>(img + (p * ROWS + 32 * nt + ((s + 1) >> 2)) * PITCH + 16 * ((s + 1) & 3));
        }
        const u32x4* Bf = Bq[s & 1];
        FENCE;
        if (H) {                                         // lo hi, hi lo, hi hi: two slices behind every MFMA
            ac = mma(W[s][1], Bf[0], ac); FENCE; f0(); FENCE; f1(); FENCE;
            ac = mma(W[s][0], Bf[1], ac); FENCE; f2(); FENCE; f3(); FENCE;
            ac = mma(W[s][0], Bf[0], ac); FENCE; f4(); FENCE; f5(); FENCE;
        } else {
            ac = mma(W[s][1], Bf[1], ac); FENCE; f0(); FENCE;
            ac = mma(W[s][0], Bf[NP - 1], ac); FENCE; f1(); FENCE;
            ac = mma(W[s][NP - 1], Bf[0], ac); FENCE; f2(); FENCE;
            ac = mma(W[s][0], Bf[1], ac); FENCE; f3(); FENCE;
            ac = mma(W[s][1], Bf[0], ac); FENCE; f4(); FENCE;
            ac = mma(W[s][0], Bf[0], ac); FENCE; f5(); FENCE;
        }
    };
    // the bf16x3 split of a value pair in three stages (one slice each)
    float sva = 0.f, svb = 0.f;
    unsigned sp0 = 0, sp1 = 0, sp2 = 0;
    auto st1 = [&]() {
        if (H) {
            const h16x2 h = __builtin_convertvector(f32x2{sva, svb}, h16x2);
            sp0 = __builtin_bit_cast(unsigned, h);
            sva -= (float)h.x; svb -= (float)h.y;
        } else {
            const bf16x2 h = {(__bf16)sva, (__bf16)svb};
            sp0 = __builtin_bit_cast(unsigned, h);
            sva -= __uint_as_float(sp0 << 16); svb -= __uint_as_float(sp0 & 0xffff0000u);
        }
        asm volatile("" : "+v"(sva), "+v"(svb), "+v"(sp0));      // pins the stage into its slice (pure arithmetic sinks to its use otherwise)
    };
    auto st2 = [&]() {
        if (H) return;                                   // two pieces: no middle one
        const bf16x2 m = {(__bf16)sva, (__bf16)svb};
        sp1 = __builtin_bit_cast(unsigned, m);
        sva -= __uint_as_float(sp1 << 16); svb -= __uint_as_float(sp1 & 0xffff0000u);
        asm volatile("" : "+v"(sva), "+v"(svb), "+v"(sp1));
    };
    auto st3 = [&](unsigned* img32, int o) {
        if (H) {
            sp2 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{sva, svb}, h16x2));
            img32[o] = sp0; img32[(ROWS * PITCH >> 1) + o] = sp2;
            return;
        }
        const bf16x2 l = {(__bf16)sva, (__bf16)svb};
        sp2 = __builtin_bit_cast(unsigned, l);
        img32[o] = sp0; img32[(ROWS * PITCH >> 1) + o] = sp1; img32[2 * (ROWS * PITCH >> 1) + o] = sp2;
    };

    while (tile < ntiles) {
        const int next = min(tile + tstep, ntiles - 1);
        const int b = tile / tilesPerClip, w0 = (tile - b * tilesPerClip) * NTO - 4, o0 = w0 + 2;
        const int nw0 = (next % tilesPerClip) * NTO - 4;
        // output column of accumulator block nt of this lane: byte offset inside a channel row, 0xffffffff (dropped by the
        // buffer unit / read as 0) when the column is not an output of this tile; in1[nt]: the a1 column lies inside the clip
        unsigned tcl[2];
        bool in1[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int i = 64 * nh + 32 * nt + l31, t = o0 + i, u = w0 + 1 + i;
            const bool ok = (i < NTO) && (t >= 0) && (t < T);
            tcl[nt] = ok ? (unsigned)(t + 4 * half * T) * 4u : 0xffffffffu;
            in1[nt] = (u >= 0) && (u < T);
        }
        // this wave's 32 channel rows of clip b, as buffer descriptors (scalar); row r sits at the scalar offset roff(r)
        const size_t slab = ((size_t)b * 64 + 32 * mt) * T;
        const wm_srd_t sxr = make_srd(a.x + slab, (size_t)32 * T * sizeof(float));
        const wm_srd_t syr = make_srd(a.y + slab, (size_t)32 * T * sizeof(float));
        auto roff = [&](int r) { return (unsigned)(((r & 3) + 8 * (r >> 2)) * T) * 4u; };                  // wave-uniform
        auto e1_load = [&](int idx) { e1r[idx] = buf_load(sxr, tcl[idx >> 4], roff(idx & 15)); };
        // epilogue 1 of accumulator pair pi (rows 2 pi, 2 pi + 1 = two adjacent channels) of block nt, in slices:
        // BN1 offset + ReLU + zero padding -> split stages -> a1 image
        auto e1_act = [&](const f32x16& ac, int nt, int pi) {
            const float v0 = fmaxf(H ? fmaf(ac[2 * pi], winv1, k1[2 * pi]) : ac[2 * pi] + k1[2 * pi], 0.f),
                        v1 = fmaxf(H ? fmaf(ac[2 * pi + 1], winv1, k1[2 * pi + 1]) : ac[2 * pi + 1] + k1[2 * pi + 1], 0.f);
            sva = in1[nt] ? v0 : 0.f; svb = in1[nt] ? v1 : 0.f;
            asm volatile("" : "+v"(sva), "+v"(svb));
        };
        auto e1_out = [&](int nt, int pi) {
            const int j = 64 * nh + 32 * nt + l31, co = 32 * mt + mfma_row(2 * pi, half);
            st3(A32, (j * PITCH + co) >> 1);
        };
        // split of unit (combo i, element e) of the next tile's x window, in slices
        auto x_pick = [&](int i, int e) {
            const int t = nw0 + 4 * (q0 + 8 * i);
            const bool ok = (t >= 0) && (t < T);
            const float4 fa = sa[i], fb = sb[i];
            const float va = (e == 0) ? fa.x : (e == 1) ? fa.y : (e == 2) ? fa.z : fa.w;
            const float vb = (e == 0) ? fb.x : (e == 1) ? fb.y : (e == 2) ? fb.z : fb.w;
            sva = ok ? va : 0.f; svb = ok ? vb : 0.f;
            asm volatile("" : "+v"(sva), "+v"(svb));
        };
        auto x_out = [&](int i, int e) { st3(X32, (4 * (q0 + 8 * i) + e) * (PITCH / 2) + cp); };
        // epilogue 2 of accumulator row r of block nt: BN2 offset + residual + ReLU + store
        auto e2_value = [&](const f32x16& ac, int nt, int r) {
            const float v = fmaxf(e1r[nt * 16 + r] + (H ? fmaf(ac[r], winv2, k2[r]) : ac[r] + k2[r]), 0.f);
            buf_store(syr, v, tcl[nt], roff(r));
        };

        f32x16 acc[2];
        u32x4 Bq[2][NP];
        STAMP(ts0);
        // ---------------- conv1 from the x image
        const unsigned short* xrow = Xb + (64 * nh + l31) * PITCH + 8 * half;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
#pragma unroll
        for (int p = 0; p < NP; ++p) Bq[0][p] = *reinterpret_cast<const u32x4*>(xrow + p * ROWS * PITCH);
        // column block 0; side work: fetch the next tile's x window (split during conv2)
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            if (s < 4) kstep(W1, xrow, acc[0], Bq, 0, s, [&]() { load_combo(next, s); }, nop, nop, nop, nop, nop);
            else kstep(W1, xrow, acc[0], Bq, 0, s, nop, nop, nop, nop, nop, nop);
        }
        STAMP(ts1);
#pragma unroll
        for (int p = 0; p < NP; ++p) Bq[0][p] = *reinterpret_cast<const u32x4*>(xrow + (p * ROWS + 32) * PITCH);
        // column block 1; side work: epilogue 1 of block 0
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            if (s < 8)
                kstep(W1, xrow, acc[1], Bq, 1, s, [&]() { e1_act(acc[0], 0, s); }, st1, st2, [&]() { e1_out(0, s); }, nop, nop);
            else kstep(W1, xrow, acc[1], Bq, 1, s, nop, nop, nop, nop, nop, nop);
        }
        // epilogue 1 of block 1 (serial)
        STAMP(ts2);
#pragma unroll
        for (int pi = 0; pi < 8; ++pi) { e1_act(acc[1], 1, pi); st1(); st2(); e1_out(1, pi); }
        STAMP(ts3);
        lds_barrier();          // a1 image complete; every wave is done with the x image
        STAMP(ts4);
        // ---------------- conv2 from the a1 image
        const unsigned short* arow = Ab + (64 * nh + l31) * PITCH + 8 * half;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
#pragma unroll
        for (int p = 0; p < NP; ++p) Bq[0][p] = *reinterpret_cast<const u32x4*>(arow + p * ROWS * PITCH);
        // column block 0; side work: first half of the next tile's x window, residual operand of block 0
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            if (s < 8)
                kstep(W2, arow, acc[0], Bq, 0, s, [&]() { x_pick(s >> 2, s & 3); }, st1, st2, [&]() { x_out(s >> 2, s & 3); },
                      [&]() { e1_load(2 * s); }, [&]() { e1_load(2 * s + 1); });
            else kstep(W2, arow, acc[0], Bq, 0, s, nop, nop, nop, nop, nop, nop);
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) Bq[0][p] = *reinterpret_cast<const u32x4*>(arow + (p * ROWS + 32) * PITCH);
        // column block 1; side work: second half of the window, residual operand of block 1, epilogue 2 of block 0
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            if (s < 4)
                kstep(W2, arow, acc[1], Bq, 1, s, [&]() { x_pick(2 + (s >> 2), s & 3); }, st1, st2, [&]() { x_out(2 + (s >> 2), s & 3); },
                      [&]() { e1_load(16 + 4 * s); e1_load(17 + 4 * s); }, [&]() { e1_load(18 + 4 * s); e1_load(19 + 4 * s); });
            else if (s < 8)
                kstep(W2, arow, acc[1], Bq, 1, s, [&]() { x_pick(2 + (s >> 2), s & 3); }, st1, st2, [&]() { x_out(2 + (s >> 2), s & 3); },
                      [&]() { e2_value(acc[0], 0, 2 * (s - 4)); }, [&]() { e2_value(acc[0], 0, 2 * (s - 4) + 1); });
            else
                kstep(W2, arow, acc[1], Bq, 1, s, nop, nop, nop, nop,
                      [&]() { e2_value(acc[0], 0, 2 * (s - 4)); }, [&]() { e2_value(acc[0], 0, 2 * (s - 4) + 1); });
        }
        // epilogue 2 of block 1 (serial)
        STAMP(ts5);
#pragma unroll
        for (int r = 0; r < 16; ++r) e2_value(acc[1], 1, r);
        lds_barrier();          // next x image complete; a1 image free
        STAMP(ts6);
#ifdef WM_STAMP
        tm[0] += ts1 - ts0; tm[1] += ts2 - ts1; tm[2] += ts3 - ts2; tm[3] += ts4 - ts3; tm[4] += ts5 - ts4; tm[5] += ts6 - ts5;
#endif
        tile += tstep;
    }
#undef FENCE
#ifdef WM_STAMP
    if (g_wm_stamp && lane == 0) {
        unsigned long long* d = g_wm_stamp + ((size_t)blockIdx.x * 4 + wave) * 6;
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i] = tm[i];
    }
#endif
}

template <bool H>
static int launch_resblock_eval(const RbeArgs& a, hipStream_t stream) {
    constexpr size_t lds = (size_t)(2 * (H ? 2 : 3) * 130 * 72) * 2 + 2 * 64 * sizeof(float);
    static wm::DevOnce attr_done;
    if (!wm::dev_done(attr_done)) {
        WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(resblock_eval_kernel<H>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        wm::dev_mark(attr_done);
    }
    const int ntiles = a.B * ((a.T + 2 + 123) / 124);
    const int grid = ntiles < kNumCU ? ntiles : kNumCU;
    hipLaunchKernelGGL(resblock_eval_kernel<H>, dim3(grid), dim3(256), lds, stream, a);
    WM_CHECK_LAUNCH();
    return 0;
}


// ---------------------------------------------------------------------------------------------
// bf16x6 build of the 7-tap convolution (Generator.decoder[0] = ConvTranspose1d(64,64,7,padding=3), py/main16.py:144,
// forward and data gradient).  The 3-piece image of all 7 taps is 194 KB -- more than LDS -- so a workgroup keeps the
// taps of HALF the output channels (97 KB) and the (tile, half) pairs are dealt over the grid: workgroup g owns half
// g & 1 for its whole life.  The input tile is read by two workgroups (8 MB/clip extra on a kernel that moves 8 MB per
// 918 MFLOP: still far from HBM-bound); each of the four waves owns 32 columns of the 32 x 128 output tile.
// ---------------------------------------------------------------------------------------------
template <int PRO, int EPI>
__global__ __launch_bounds__(256) void conv64bf7_kernel(Conv64Args a) {
    constexpr int KW = 7, PAD = 3, NT = 128, ROWS = NT + 2 * PAD, PITCH = 72, NP = 3, NC = 4, MR = 32;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    unsigned short* Wb = reinterpret_cast<unsigned short*>(smem_raw);              // [NP][KW][MR][PITCH]
    unsigned short* Xb = Wb + NP * KW * MR * PITCH;                                // [NP][ROWS][PITCH]
    float* Cs = reinterpret_cast<float*>(Xb + NP * ROWS * PITCH);                  // [32] bias of this half
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int mh = blockIdx.x & 1, nslots = gridDim.x >> 1;                        // gridDim.x is even
    const int T = a.T;
    const int tilesPerClip = (T + NT - 1) / NT, ntiles = a.B * tilesPerClip;

    float4 sa[NC], sb[NC];
    float hl[2];
    float pv0[NC], pv1[NC], hv[2];                // ADDVEC: the clip's embedding values, prefetched with the tile (a load
                                                  // inside write_tile would put a global round trip into the serial phase)
    auto combo = [&](int i, int& cp, int& q) {
        const int idx = tid + i * 256, widx = idx >> 6, l = idx & 63;
        cp = (widx & 3) * 8 + (l & 7);
        q = (widx >> 2) * 8 + (l >> 3);
    };
    auto halo_of = [&](int k, int t0, int& c, int& r, int& t) {                    // k-th halo element of this thread
        const int idx = min(tid + k * 256, 64 * 2 * PAD - 1);
        c = idx / (2 * PAD);
        const int h = idx % (2 * PAD);
        r = (h < PAD) ? h : NT + h;                                                 // image row: time = t0 - PAD + r
        t = t0 - PAD + r;
    };
    // piece i < NC: one staging combo; piece NC: the halo.  Branch-free (clamped addresses, masked at the LDS write); the
    // main loop issues one piece per tap so that the tile's loads never arrive at the memory pipeline as one burst
    auto load_piece = [&](int tile, int i) {
        const int b = tile / tilesPerClip, t0 = (tile - b * tilesPerClip) * NT;
        const float* xb = a.x + (size_t)b * 64 * T;
        if (i < NC) {
            int cp, q;
            combo(i, cp, q);
            const size_t o = (size_t)(2 * cp) * T + min(t0 + 4 * q, T - 4);
            sa[i] = *reinterpret_cast<const float4*>(xb + o);
            sb[i] = *reinterpret_cast<const float4*>(xb + o + T);
            if (PRO == PRO_ADDVEC) { pv0[i] = a.pa[b * 64 + 2 * cp]; pv1[i] = a.pa[b * 64 + 2 * cp + 1]; }
        } else {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                int c, r, t;
                halo_of(k, t0, c, r, t);
                hl[k] = xb[(size_t)c * T + min(max(t, 0), T - 1)];
                if (PRO == PRO_ADDVEC) hv[k] = a.pa[b * 64 + c];
            }
        }
    };
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int i = 0; i <= NC; ++i) load_piece(tile, i);
    };
    auto write_tile = [&](int tile) {
        const int b = tile / tilesPerClip, t0 = (tile - b * tilesPerClip) * NT;
        unsigned* X32 = reinterpret_cast<unsigned*>(Xb);
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            int cp, q;
            combo(i, cp, q);
            const int c = 2 * cp, t = t0 + 4 * q;
            float va[4] = {sa[i].x, sa[i].y, sa[i].z, sa[i].w}, vb[4] = {sb[i].x, sb[i].y, sb[i].z, sb[i].w};
            if (PRO == PRO_ADDVEC) {
                const float v0 = pv0[i], v1 = pv1[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) { va[e] += v0; vb[e] += v1; }
            }
            const bool ok = t < T;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned p0, p1, p2;
                split3_pair(ok ? va[e] : 0.f, ok ? vb[e] : 0.f, p0, p1, p2);
                const int o = ((PAD + 4 * q + e) * PITCH + c) >> 1;
                X32[o] = p0; X32[(ROWS * PITCH >> 1) + o] = p1; X32[2 * (ROWS * PITCH >> 1) + o] = p2;
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (tid + k * 256 < 64 * 2 * PAD) {
                int c, r, t;
                halo_of(k, t0, c, r, t);
                float v = hl[k];
                if (PRO == PRO_ADDVEC) v += hv[k];
                if (t < 0 || t >= T) v = 0.f;
                unsigned p0, p1, p2;
                split3_pair(v, 0.f, p0, p1, p2);
                const int o = r * PITCH + c;
                Xb[o] = (unsigned short)p0; Xb[ROWS * PITCH + o] = (unsigned short)p1; Xb[2 * ROWS * PITCH + o] = (unsigned short)p2;
            }
        }
    };

    int tile = blockIdx.x >> 1;
    if (tile < ntiles) load_tile(tile);
    // resident weights of this half: global image [NP][KW][64 out][64 in] bf16 -> LDS [NP][KW][32][PITCH]
    for (int i = tid; i < NP * KW * MR * 8; i += 256) {
        const int row = i >> 3, seg = i & 7, pt = row / MR, o = row % MR;
        *reinterpret_cast<uint4*>(Wb + row * PITCH + seg * 8) = reinterpret_cast<const uint4*>(a.wp)[(pt * 64 + mh * MR + o) * 8 + seg];
    }
    if (tid < MR) Cs[tid] = (EPI == EPI_BIAS && a.bias) ? a.bias[mh * MR + tid] : 0.f;
    __syncthreads();
    if (tile < ntiles) write_tile(tile);
    __syncthreads();

    while (tile < ntiles) {
        const int next = tile + nslots, nextc = min(next, ntiles - 1);   // clamped: loaded (valid memory) but never written
        const int b = tile / tilesPerClip, t0 = (tile - b * tilesPerClip) * NT;
        const int tcol = t0 + wave * 32 + l31;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const unsigned short* wbase = Wb + l31 * PITCH + 8 * half;
        const unsigned short* xbase = Xb + (wave * 32 + l31) * PITCH + 8 * half;
#pragma unroll
        for (int tap = 0; tap < KW; ++tap) {
            if (tap >= 1 && tap <= NC + 1) load_piece(nextc, tap - 1);
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                bf16x8 A[NP], Bf[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    Bf[p] = *reinterpret_cast<const bf16x8*>(xbase + (p * ROWS + tap) * PITCH + 16 * ch);
                    A[p] = *reinterpret_cast<const bf16x8*>(wbase + ((p * KW + tap) * MR) * PITCH + 16 * ch);
                }
                acc = mfma_bf16x6(A, Bf, acc);
            }
        }
        float* yb = a.y + ((size_t)b * 64 + mh * MR) * T + tcol;
        if (tcol < T) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mfma_row(r, half);
                yb[(size_t)row * T] = acc[r] + Cs[row];
            }
        }
        __syncthreads();
        if (next < ntiles) write_tile(next);
        __syncthreads();
        tile = next;
    }
}

template <int PRO, int EPI>
int launch_conv64bf7(const Conv64Args& a, hipStream_t stream) {
    constexpr size_t lds = (size_t)(3 * 7 * 32 * 72 + 3 * 134 * 72) * 2 + 32 * sizeof(float);
    static wm::DevOnce attr_done;
    auto kern = conv64bf7_kernel<PRO, EPI>;
    if (!wm::dev_done(attr_done)) {
        WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        wm::dev_mark(attr_done);
    }
    const int ntiles = a.B * ((a.T + 127) / 128);
    const int grid = 2 * (ntiles < kNumCU / 2 ? ntiles : kNumCU / 2);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
    WM_CHECK_LAUNCH();
    return 0;
}

// bf16 three-piece image [piece][tap][out][in] of a 7-tap ConvTranspose1d weight w[in][out][7]: mode 2 forward, 3 dgrad
// ---------------------------------------------------------------------------------------------
// Pipelined build of the 7-tap convolution (conv64bf3_kernel's scheme with KW = 7).  conv64bf7_kernel above keeps the
// weight image in LDS (97 KB per output half), so two workgroups split every input tile twice and the matrix cores idle
// through each split phase.  Here wave (mt, nh) keeps the weight fragments of its 32 output rows in REGISTERS for the whole
// kernel (28 k-steps x 3 pieces x 4 = 336 of the 512 a lone resident wave may use), one workgroup computes all 64 rows of
// a 128-column tile (336 MFMAs per wave), and LDS only carries the input image, double-buffered (2 x 58 KB): the split of
// tile i+1 and the fetch of tile i+2 ride, one hand-pinned slice per MFMA, in the matrix phase of tile i.  T % 128 == 0.
// ---------------------------------------------------------------------------------------------
// H = true: f16 two-piece build (three products per product; weight image from wm_pack_w64_h7: w * ws; a.pb, when given, = {gs, 1 / gs}:
// the input -- a gradient -- is multiplied by gs before the split and clamped to the f16 range; the accumulators leave times 1 / (ws gs))
template <int PRO, int EPI, bool H = false>
__global__ __launch_bounds__(256) void conv64bf7p_kernel(Conv64Args a) {
    static_assert(PRO == PRO_NONE || PRO == PRO_ADDVEC, "prologue: none or + vec[b][c]");
    constexpr int KW = 7, PAD = 3, NT = 128, ROWS = NT + 2 * PAD, PITCH = 72, NP = H ? 2 : 3, NC = 4, NS = KW * 4;
    constexpr int XBUF = NP * ROWS * PITCH;               // bf16 elements per input image
    extern __shared__ __align__(16) unsigned char smem_raw[];
    unsigned short* Xb0 = reinterpret_cast<unsigned short*>(smem_raw);             // 2 x [NP][ROWS][PITCH]
    float* Cs = reinterpret_cast<float*>(Xb0 + 2 * XBUF);                           // [64] bias
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = wave & 1, nh = wave >> 1;
    const int T = a.T;                             // T % 128 == 0 (checked by the launcher): no partial tiles
    const int tilesPerClip = T / NT, ntiles = a.B * tilesPerClip;

    // ---- resident weight fragments: packed image [NP][KW][64 out][64 in] bf16; k-step s = tap * 4 + 16-channel block
    u32x4 Wr[NS][NP];
    {
        const uint4* wg = reinterpret_cast<const uint4*>(a.wp);
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int e = ((p * KW + (s >> 2)) * 64 + 32 * mt + l31) * 64 + 16 * (s & 3) + 8 * half;
                u32x4 w_ = __builtin_bit_cast(u32x4, wg[e >> 3]);
                if (s >= 8) asm volatile("" : "+a"(w_));      // 336 fragment registers: all but the first 8 k-steps pinned into AGPRs,
                Wr[s][p] = w_;                                 // where the MFMA reads them in place (no spill / copy-back traffic)
            }
    }
    float gs = 1.f, dinv = 1.f;                    // H: input scale, output scale
    if (H) {
        const float* tail = reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(a.wp) + NP * KW * 4096);
        dinv = tail[1];
        if (a.pb) { gs = a.pb[0]; dinv *= a.pb[1]; }
    }
    // ---- staging map (fixed per thread): channel pair cp, time quads q0 + 8 i; halo element k: channel hcn[k], image row hr[k]
    const int cp = wave * 8 + (lane & 7), c0 = 2 * cp, q0 = lane >> 3;
    int hcn[2], hr[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int idx = min(tid + k * 256, 64 * 2 * PAD - 1), h = idx % (2 * PAD);
        hcn[k] = idx / (2 * PAD);
        hr[k] = (h < PAD) ? h : NT + h;                                             // time = t0 - PAD + row
    }
    float4 sa[NC], sb[NC];
    float hl[2];
    float pv0 = 0.f, pv1 = 0.f, hv[2] = {0.f, 0.f};      // ADDVEC: the embedding values of the clip whose tile sits in the registers
    float nv0 = 0.f, nv1 = 0.f, nhv[2] = {0.f, 0.f};     // ... and of the tile being fetched
    auto load_combo = [&](int tile, int i) {
        const int b = tile / tilesPerClip, t0 = (tile - b * tilesPerClip) * NT;
        const size_t o = ((size_t)b * 64 + c0) * T + t0 + 4 * (q0 + 8 * i);
        sa[i] = *reinterpret_cast<const float4*>(a.x + o);
        sb[i] = *reinterpret_cast<const float4*>(a.x + o + T);
    };
    auto load_halo = [&](int tile) {              // branch-free: clamped addresses, masked when written to LDS
        const int b = tile / tilesPerClip, t0 = (tile - b * tilesPerClip) * NT;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            hl[k] = a.x[((size_t)b * 64 + hcn[k]) * T + min(max(t0 - PAD + hr[k], 0), T - 1)];
            if (PRO == PRO_ADDVEC) nhv[k] = a.pa[b * 64 + hcn[k]];
        }
        if (PRO == PRO_ADDVEC) { nv0 = a.pa[b * 64 + c0]; nv1 = a.pa[b * 64 + c0 + 1]; }
    };

    const int tstep = gridDim.x;
    int tile = WM_XCD_MAP ? xcd_slot() : (int)blockIdx.x;        // grid <= ntiles
#pragma unroll
    for (int i = 0; i < NC; ++i) load_combo(tile, i);
    load_halo(tile);
    if (tid < 64) Cs[tid] = (EPI == EPI_BIAS && a.bias) ? a.bias[tid] : 0.f;

    // ---- the split of one (combo i, element e) unit = two channels x one time step, in four stages (one slice each)
    float sva = 0.f, svb = 0.f;
    unsigned sp0 = 0, sp1 = 0;
    auto split_pick = [&](int i, int e) {
        const float4 fa = sa[i], fb = sb[i];
        float va = (e == 0) ? fa.x : (e == 1) ? fa.y : (e == 2) ? fa.z : fa.w;
        float vb = (e == 0) ? fb.x : (e == 1) ? fb.y : (e == 2) ? fb.z : fb.w;
        if (PRO == PRO_ADDVEC) { va += pv0; vb += pv1; }
        if (H) { va = __builtin_amdgcn_fmed3f(va * gs, -6.0e4f, 6.0e4f); vb = __builtin_amdgcn_fmed3f(vb * gs, -6.0e4f, 6.0e4f); }
        sva = va; svb = vb;
        asm volatile("" : "+v"(sva), "+v"(svb));
    };
    auto split_st1 = [&]() {
        if (H) {
            const h16x2 h_ = __builtin_convertvector(f32x2{sva, svb}, h16x2);
            sp0 = __builtin_bit_cast(unsigned, h_);
            sva -= (float)h_.x; svb -= (float)h_.y;
        } else {
            const bf16x2 h_ = {(__bf16)sva, (__bf16)svb};
            sp0 = __builtin_bit_cast(unsigned, h_);
            sva -= __uint_as_float(sp0 << 16); svb -= __uint_as_float(sp0 & 0xffff0000u);
        }
        asm volatile("" : "+v"(sva), "+v"(svb), "+v"(sp0));
    };
    auto last_piece = [&]() -> unsigned {
        if (H) return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{sva, svb}, h16x2));
        const bf16x2 l_ = {(__bf16)sva, (__bf16)svb};
        return __builtin_bit_cast(unsigned, l_);
    };
    auto split_st2 = [&]() {
        if (H) return;                                   // two pieces: no middle one
        const bf16x2 m_ = {(__bf16)sva, (__bf16)svb};
        sp1 = __builtin_bit_cast(unsigned, m_);
        sva -= __uint_as_float(sp1 << 16); svb -= __uint_as_float(sp1 & 0xffff0000u);
        asm volatile("" : "+v"(sva), "+v"(svb), "+v"(sp1));
    };
    auto split_out = [&](unsigned short* X, int i, int e) {
        const unsigned sp2 = last_piece();
        unsigned* X32 = reinterpret_cast<unsigned*>(X);
        const int o = (PAD + 4 * (q0 + 8 * i) + e) * (PITCH / 2) + cp;
        X32[o] = sp0; if (!H) X32[(ROWS * PITCH >> 1) + o] = sp1; X32[(NP - 1) * (ROWS * PITCH >> 1) + o] = sp2;
    };
    auto halo_pick = [&](int t0, int k) {
        const int t = t0 - PAD + hr[k];
        float v = hl[k];
        if (PRO == PRO_ADDVEC) v += hv[k];
        if (H) v = __builtin_amdgcn_fmed3f(v * gs, -6.0e4f, 6.0e4f);
        sva = (t < 0 || t >= T) ? 0.f : v; svb = 0.f;
        asm volatile("" : "+v"(sva), "+v"(svb));
    };
    auto halo_out = [&](unsigned short* X, int k) {
        const unsigned sp2 = last_piece();
        if (tid + k * 256 < 64 * 2 * PAD) {
            const int o = hr[k] * PITCH + hcn[k];
            X[o] = (unsigned short)sp0; if (!H) X[ROWS * PITCH + o] = (unsigned short)sp1; X[(NP - 1) * ROWS * PITCH + o] = (unsigned short)sp2;
        }
    };
    {   // first tile: split serially, then fetch the second
        const int t0 = (tile % tilesPerClip) * NT;
        pv0 = nv0; pv1 = nv1; hv[0] = nhv[0]; hv[1] = nhv[1];
#pragma unroll
        for (int u = 0; u < 16; ++u) { split_pick(u >> 2, u & 3); split_st1(); split_st2(); split_out(Xb0, u >> 2, u & 3); }
#pragma unroll
        for (int k = 0; k < 2; ++k) { halo_pick(t0, k); split_st1(); split_st2(); halo_out(Xb0, k); }
        const int nx = min(tile + tstep, ntiles - 1);
#pragma unroll
        for (int i = 0; i < NC; ++i) load_combo(nx, i);
        load_halo(nx);
    }
    __syncthreads();

    int buf = 0;
    auto mma = [&](const u32x4& A_, const u32x4& B_, f32x16 c) -> f32x16 {
        if (H) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, A_), __builtin_bit_cast(h16x8, B_), c, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A_), __builtin_bit_cast(bf16x8, B_), c, 0, 0, 0);
    };
#define FENCE __builtin_amdgcn_sched_barrier(0)
    while (tile < ntiles) {
        // registers: the operands of tile + tstep (clamped: a tile past the end lands in the image nobody reads again)
        const int next = min(tile + tstep, ntiles - 1), next2 = min(tile + 2 * tstep, ntiles - 1);
        const int b = tile / tilesPerClip, t0 = (tile - b * tilesPerClip) * NT;
        const int nt0 = (next % tilesPerClip) * NT;
        const unsigned short* xcur = Xb0 + buf * XBUF;
        unsigned short* xnxt = Xb0 + (buf ^ 1) * XBUF;
        f32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        const unsigned short* xrow = xcur + (64 * nh + l31) * PITCH + 8 * half;
        u32x4 Bq[2][NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) Bq[0][p] = *reinterpret_cast<const u32x4*>(xrow + p * ROWS * PITCH);
        // the embedding values of the tile in the registers become current; load_halo() below refills the "next" set
        pv0 = nv0; pv1 = nv1; hv[0] = nhv[0]; hv[1] = nhv[1];
#define BF7_SLICE(k)                                                                                                        \
    {                                                                                                                           \
        if (h < 16) {                                                                                                           \
            if ((k) == 0) split_pick(h >> 2, h & 3);                                                                            \
            if ((k) == 1) split_st1();                                                                                          \
            if ((k) == 2) split_st2();                                                                                          \
            if ((k) == 3) { split_out(xnxt, h >> 2, h & 3); if ((h & 3) == 3) load_combo(next2, h >> 2); }                      \
        } else if (h < 18) {                                                                                                    \
            if ((k) == 0) halo_pick(nt0, h - 16);                                                                               \
            if ((k) == 1) split_st1();                                                                                          \
            if ((k) == 2) split_st2();                                                                                          \
            if ((k) == 3) halo_out(xnxt, h - 16);                                                                               \
        } else if (h == 18) {                                                                                                   \
            if ((k) == 0) load_halo(next2);                                                                                     \
        }                                                                                                                       \
    }
#pragma unroll
        for (int h = 0; h < 2 * NS; ++h) {
            const int s = h >> 1, nt = h & 1;
            if (h + 1 < 2 * NS) {
                const int s1_ = (h + 1) >> 1, n1 = (h + 1) & 1;
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    Bq[(h + 1) & 1][p] = *reinterpret_cast<const u32x4*>(xrow + (p * ROWS + 32 * n1 + (s1_ >> 2)) * PITCH + 16 * (s1_ & 3));
            }
            const u32x4* Bf = Bq[h & 1];
            FENCE;
            if (H) {                                     // lo hi, hi lo, hi hi
                acc[nt] = mma(Wr[s][1], Bf[0], acc[nt]); FENCE; BF7_SLICE(0) FENCE; BF7_SLICE(1) FENCE;
                acc[nt] = mma(Wr[s][0], Bf[1], acc[nt]); FENCE; BF7_SLICE(3) FENCE;
                acc[nt] = mma(Wr[s][0], Bf[0], acc[nt]); FENCE;
            } else {
                acc[nt] = mma(Wr[s][1], Bf[1], acc[nt]); FENCE; BF7_SLICE(0) FENCE;
                acc[nt] = mma(Wr[s][0], Bf[2], acc[nt]); FENCE; BF7_SLICE(1) FENCE;
                acc[nt] = mma(Wr[s][2], Bf[0], acc[nt]); FENCE; BF7_SLICE(2) FENCE;
                acc[nt] = mma(Wr[s][0], Bf[1], acc[nt]); FENCE; BF7_SLICE(3) FENCE;
                acc[nt] = mma(Wr[s][1], Bf[0], acc[nt]); FENCE;
                acc[nt] = mma(Wr[s][0], Bf[0], acc[nt]); FENCE;
            }
        }
#undef BF7_SLICE
        // epilogue (serial: 32 values against 336 MFMAs)
        {
            float* yb = a.y + ((size_t)b * 64 + 32 * mt + 4 * half) * T + t0 + 64 * nh + l31;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2);
                    float v = acc[nt][r];
                    if (H) v *= dinv;
                    if (EPI == EPI_BIAS) v += Cs[32 * mt + 4 * half + row];
                    yb[(size_t)row * T + 32 * nt] = v;
                }
        }
        lds_barrier();
        tile += tstep;
        buf ^= 1;
    }
#undef FENCE
}

template <int PRO, int EPI, bool H = false>
int launch_conv64bf7p(const Conv64Args& a, hipStream_t stream) {
    constexpr size_t lds = (size_t)(2 * (H ? 2 : 3) * 134 * 72) * 2 + 64 * sizeof(float);
    static wm::DevOnce attr_done;
    auto kern = conv64bf7p_kernel<PRO, EPI, H>;
    if (!wm::dev_done(attr_done)) {
        WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        wm::dev_mark(attr_done);
    }
    const int ntiles = a.B * (a.T / 128);
    const int grid = ntiles < kNumCU ? ntiles : kNumCU;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
    WM_CHECK_LAUNCH();
    return 0;
}

__global__ void pack_w64_bf7_kernel(const float* __restrict__ w, unsigned short* __restrict__ wpb, int mode) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 7 * 4096) return;
    const int tap = i / 4096, out = (i / 64) % 64, in = i % 64;
    const float v = (mode == 2) ? w[(in * 64 + out) * 7 + (6 - tap)] : w[(out * 64 + in) * 7 + tap];
    unsigned p0, p1, p2;
    split3_pair(v, 0.f, p0, p1, p2);
    wpb[i] = (unsigned short)p0; wpb[7 * 4096 + i] = (unsigned short)p1; wpb[2 * 7 * 4096 + i] = (unsigned short)p2;
}

// bf16 three-piece weight image [piece][tap][out][in] (uint16) for the k3 convolutions; mode as wm_pack_w64 (0 / 1)
__global__ void pack_w64_bf_kernel(const float* __restrict__ w, const float* __restrict__ row_scale,
                                   unsigned short* __restrict__ wpb, int mode) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * 4096) return;
    const int tap = i / 4096, out = (i / 64) % 64, in = i % 64;
    float v = (mode == 0) ? w[(out * 64 + in) * 3 + tap] : w[(in * 64 + out) * 3 + (2 - tap)];
    if (row_scale) v *= row_scale[out];          // a per-output-channel factor (folded BatchNorm scale) rides in the weights
    unsigned p0, p1, p2;
    split3_pair(v, 0.f, p0, p1, p2);
    wpb[i] = (unsigned short)p0; wpb[3 * 4096 + i] = (unsigned short)p1; wpb[2 * 3 * 4096 + i] = (unsigned short)p2;
}

// ---------------------------------------------------------------------------------------------
// bf16x6 weight gradient of the k3 convolution.  The contraction runs over time, so both operands keep the natural
// [channel][time] layout in LDS (one aligned ds_read_b128 = 8 consecutive time steps of a channel = one MFMA fragment,
// no transposition).  The +-1 time shift of taps 0 and 2 is made in registers: the centre fragment plus the two
// neighbouring elements are funnel-shifted with v_alignbit (VALU work that runs beside the bf16 MFMA pipe).
// ---------------------------------------------------------------------------------------------
template <int GPRO, int XPRO>
__global__ __launch_bounds__(256) void wgrad64bf_kernel(Wgrad64Args a) {
    constexpr int KW = 3, NT = 128, NP = 3, PG = 136, PX = 152, XO = 8;   // X element index = (t - t0) + XO
    constexpr int QR = NT / 4, NV = 64 * QR / 256;
    constexpr bool GTWO = (GPRO == PRO_BNBWD);
    extern __shared__ __align__(16) unsigned char smem_raw[];
    unsigned short* Gb = reinterpret_cast<unsigned short*>(smem_raw);          // [NP][64][PG]
    unsigned short* Xb = Gb + NP * 64 * PG;                                    // [NP][64][PX]
    float* Cs = reinterpret_cast<float*>(Xb + NP * 64 * PX);                   // [5][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int T = a.T;
    const int tilesPerClip = (T + NT - 1) / NT, ntiles = a.B * tilesPerClip;

    float4 sg[NV], sg2[GTWO ? NV : 1], sx[NV];
    float hx;
    float bsum[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) bsum[i] = 0.f;

    // global -> register staging in NV + 1 pieces: the main loop issues one piece per MFMA block (a 64..96-KB burst per
    // CU backs up the memory pipeline and blocks the wave at issue for thousands of cycles; see conv64bf3_kernel)
    auto load_piece = [&](int tile, int i) {    // branch-free
        const int b = tile / tilesPerClip, t0 = (tile - b * tilesPerClip) * NT;
        const size_t base = (size_t)b * 64 * T;
        if (i < NV) {
            const int idx = tid + i * 256, c = idx / QR, q = idx % QR;
            const int t = min(t0 + 4 * q, T - 4);
            sg[i] = *reinterpret_cast<const float4*>(a.g + base + (size_t)c * T + t);
            if (GTWO) sg2[i] = *reinterpret_cast<const float4*>(a.g2 + base + (size_t)c * T + t);
            sx[i] = *reinterpret_cast<const float4*>(a.x + base + (size_t)c * T + t);
        } else {
            const int hc = (tid & 127) >> 1, hh = tid & 1;
            hx = a.x[base + (size_t)hc * T + min(max(hh ? t0 + NT : t0 - 1, 0), T - 1)];
        }
    };
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int i = 0; i <= NV; ++i) load_piece(tile, i);
    };
    auto put4 = [&](unsigned short* dst, int stride_p, float v0, float v1, float v2, float v3) {
        unsigned a0, a1, a2, b0, b1, b2;
        split3_pair(v0, v1, a0, a1, a2);
        split3_pair(v2, v3, b0, b1, b2);
        *reinterpret_cast<uint2*>(dst) = make_uint2(a0, b0);
        *reinterpret_cast<uint2*>(dst + stride_p) = make_uint2(a1, b1);
        *reinterpret_cast<uint2*>(dst + 2 * stride_p) = make_uint2(a2, b2);
    };
    auto write_tile = [&](int tile) {
        const int b = tile / tilesPerClip, t0 = (tile - b * tilesPerClip) * NT;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * 256, c = idx / QR, q = idx % QR, t = t0 + 4 * q;
            float4 v = sg[i], u = sx[i];
            if (GPRO == PRO_BNBWD) {
                const float ca = Cs[c], cb = Cs[64 + c], cc = Cs[128 + c], cl = Cs[320 + c];
                const float4 w = sg2[i];
                v.x = pro_apply<PRO_BNBWD>(v.x, w.x, ca, cb, cc, cl); v.y = pro_apply<PRO_BNBWD>(v.y, w.y, ca, cb, cc, cl);
                v.z = pro_apply<PRO_BNBWD>(v.z, w.z, ca, cb, cc, cl); v.w = pro_apply<PRO_BNBWD>(v.w, w.w, ca, cb, cc, cl);
            }
            if (XPRO != PRO_NONE) {
                const float ca = (XPRO == PRO_ADDVEC) ? a.xa[b * 64 + c] : Cs[192 + c];
                const float cb = Cs[256 + c];
                u.x = pro_apply<XPRO>(u.x, 0.f, ca, cb, 0.f); u.y = pro_apply<XPRO>(u.y, 0.f, ca, cb, 0.f);
                u.z = pro_apply<XPRO>(u.z, 0.f, ca, cb, 0.f); u.w = pro_apply<XPRO>(u.w, 0.f, ca, cb, 0.f);
            }
            if (t >= T) { v = make_float4(0.f, 0.f, 0.f, 0.f); u = v; }
            bsum[i] += (v.x + v.y) + (v.z + v.w);
            put4(Gb + c * PG + 4 * q, 64 * PG, v.x, v.y, v.z, v.w);
            put4(Xb + c * PX + XO + 4 * q, 64 * PX, u.x, u.y, u.z, u.w);
        }
        if (tid < 128) {
            const int hc = tid >> 1, hh = tid & 1, t = hh ? t0 + NT : t0 - 1;
            float v = hx;
            if (XPRO != PRO_NONE) {
                const float ca = (XPRO == PRO_ADDVEC) ? a.xa[b * 64 + hc] : Cs[192 + hc];
                v = pro_apply<XPRO>(v, 0.f, ca, Cs[256 + hc], 0.f);
            }
            if (t < 0 || t >= T) v = 0.f;
            unsigned p0, p1, p2;
            split3_pair(v, 0.f, p0, p1, p2);
            const int o = hc * PX + (hh ? XO + NT : XO - 1);
            Xb[o] = (unsigned short)p0; Xb[64 * PX + o] = (unsigned short)p1; Xb[2 * 64 * PX + o] = (unsigned short)p2;
        }
    };

    int tile = WM_XCD_MAP ? xcd_slot() : (int)blockIdx.x;
    if (tile < ntiles) load_tile(tile);
    if (tid < 64) {
        Cs[tid] = GTWO ? a.ga[tid] : 0.f;
        Cs[64 + tid] = GTWO ? a.gb[tid] : 0.f;
        Cs[128 + tid] = GTWO ? a.gc[tid] : 0.f;
        Cs[192 + tid] = (XPRO == PRO_BNRELU) ? a.xa[tid] : 0.f;
        Cs[256 + tid] = (XPRO == PRO_BNRELU) ? a.xb[tid] : 0.f;
        Cs[320 + tid] = GTWO ? a.gb[64 + tid] : 0.f;        // low word of the BatchNorm-backward offset
    }
    // the element right of the right halo is read by the funnel shift of the last fragment: keep it defined
    for (int i = tid; i < NP * 64; i += 256) Xb[i * PX + XO + NT + 1] = 0;
    __syncthreads();
    if (tile < ntiles) write_tile(tile);
    __syncthreads();

    f32x16 acc[KW][2][2];
#pragma unroll
    for (int k = 0; k < KW; ++k)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[k][mt][nt][r] = 0.f;

    while (tile < ntiles) {
        const int next = tile + gridDim.x;
        const int nextc = min(next, ntiles - 1);         // clamped: a tile past the end is loaded (valid memory) but never written
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {                 // this wave's 32 time steps = 2 k-blocks of 16
            const int e0 = wave * 32 + ss * 16 + 8 * half;
            bf16x8 A[2][NP];
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    A[mt][p] = *reinterpret_cast<const bf16x8*>(Gb + (p * 64 + mt * 32 + l31) * PG + e0);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                bf16x8 Bc[NP], Bl[NP], Br[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const unsigned short* xr = Xb + (p * 64 + nt * 32 + l31) * PX + XO + e0;
                    const uint4 f = *reinterpret_cast<const uint4*>(xr);
                    const unsigned L = *reinterpret_cast<const unsigned*>(xr - 2), R = *reinterpret_cast<const unsigned*>(xr + 8);
                    const uint4 fl = make_uint4(__builtin_amdgcn_alignbit(f.x, L, 16), __builtin_amdgcn_alignbit(f.y, f.x, 16),
                                                __builtin_amdgcn_alignbit(f.z, f.y, 16), __builtin_amdgcn_alignbit(f.w, f.z, 16));
                    const uint4 fr = make_uint4(__builtin_amdgcn_alignbit(f.y, f.x, 16), __builtin_amdgcn_alignbit(f.z, f.y, 16),
                                                __builtin_amdgcn_alignbit(f.w, f.z, 16), __builtin_amdgcn_alignbit(R, f.w, 16));
                    Bc[p] = __builtin_bit_cast(bf16x8, f);
                    Bl[p] = __builtin_bit_cast(bf16x8, fl);
                    Br[p] = __builtin_bit_cast(bf16x8, fr);
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
#define WM_MM6(ACC, BB)                                                                              \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt][1], BB[1], ACC, 0, 0, 0);                     \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt][0], BB[2], ACC, 0, 0, 0);                     \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt][2], BB[0], ACC, 0, 0, 0);                     \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt][0], BB[1], ACC, 0, 0, 0);                     \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt][1], BB[0], ACC, 0, 0, 0);                     \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt][0], BB[0], ACC, 0, 0, 0);
                    WM_MM6(acc[0][mt][nt], Bl)
                    WM_MM6(acc[1][mt][nt], Bc)
                    WM_MM6(acc[2][mt][nt], Br)
#undef WM_MM6
                    load_piece(nextc, (ss * 2 + nt) * 2 + mt);       // pieces 0..7 ride along the 8 MFMA blocks
                }
            }
        }
        load_piece(nextc, NV);
        __syncthreads();
        if (next < ntiles) write_tile(next);
        __syncthreads();
        tile = next;
    }

    float* out = a.partial + (size_t)blockIdx.x * (KW * 4096 + 64);
    float* red = reinterpret_cast<float*>(smem_raw);          // KW*4096 floats = 48 KB <= LDS images (104 KB)
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int k = 0; k < KW; ++k)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int o = (k * 64 + mt * 32 + mfma_row(r, half)) * 64 + nt * 32 + l31;
                            red[o] = (w == 0) ? acc[k][mt][nt][r] : red[o] + acc[k][mt][nt][r];
                        }
        }
        __syncthreads();
    }
    for (int i = tid; i < KW * 4096; i += 256) out[i] = red[i];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float v = half_wave_sum(bsum[i]);
        if (l31 == 0) out[KW * 4096 + (tid >> 5) + 8 * i] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Pipelined build of wgrad64bf_kernel.  The kernel above runs its phases back to back on one wave per SIMD -- matrix phase,
// barrier, prologue + split + LDS write (640 VALU instructions per lane), barrier -- and the matrix cores sit idle through
// every VALU phase (38 % busy).  Here a tile is 64 time steps, so both 3-piece images fit twice (2 x 60 KB): while the
// 72 MFMAs of tile i run out of one image pair, the same instruction stream prologues / splits tile i+1 into the other
// and fetches tile i+2, ONE slice of side work (<= 8 VALU / LDS / global instructions) pinned behind each MFMA by
// sched_barrier fences (see conv64bf3_kernel: the six MFMAs of a piece-product group chain on one accumulator and wait
// 8 passes for each other, so a slice costs nothing).  The +-1 shifted B fragments of a column block are built while
// its unshifted products run.  One LDS-only barrier per tile.
// ---------------------------------------------------------------------------------------------
template <int GPRO, int XPRO>
__global__ __launch_bounds__(256) void wgrad64bfp_kernel(Wgrad64Args a) {
    static_assert(XPRO == PRO_NONE || XPRO == PRO_BNRELU, "x prologue: none or BN+ReLU");
    constexpr int KW = 3, NT = 64, NP = 3, PG = 72, PX = 88, XO = 8;   // X element index = (t - t0) + XO
    constexpr int QR = NT / 4, NV = 64 * QR / 256;                     // 16 quads per channel row, 4 units per thread and tensor
    constexpr bool GTWO = (GPRO == PRO_BNBWD);
    constexpr int GIMG = NP * 64 * PG, XIMG = NP * 64 * PX;            // bf16 elements per image
    extern __shared__ __align__(16) unsigned char smem_raw[];
    unsigned short* Gb0 = reinterpret_cast<unsigned short*>(smem_raw);         // [2][NP][64][PG]
    unsigned short* Xb0 = Gb0 + 2 * GIMG;                                      // [2][NP][64][PX]
    float* Cs = reinterpret_cast<float*>(Xb0 + 2 * XIMG);                      // [6][64]
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = a.T;
    const int tilesPerClip = (T + NT - 1) / NT, ntiles = a.B * tilesPerClip;

    // staging coordinates (fixed per thread): unit i = channel cq + 16 i, time quad q
    const int cq = tid >> 4, q = tid & 15;
    const int hc = (tid & 127) >> 1, hh = tid & 1;
    float4 sg[NV], sg2[GTWO ? NV : 1], sx[NV];
    float hx = 0.f;
    float bsum[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) bsum[i] = 0.f;

    // operand fetch of one tile: per-tile scalar descriptors of the clip + ONE per-lane byte offset; unit i sits 16 i rows further
    wm_srd_t dg = make_srd(a.g, 0), dg2 = dg, dx = dg;
    unsigned voff = 0, hoff = 0;
    bool okq = true, okh = true;                    // the unit's quad / the halo column lies inside the clip
    auto set_tile = [&](int tile) {
        const int b = tile / tilesPerClip, t0 = (tile - b * tilesPerClip) * NT;
        const size_t clip = (size_t)b * 64 * T, bytes = (size_t)64 * T * sizeof(float);
        dg = make_srd(a.g + clip, bytes);
        if (GTWO) dg2 = make_srd(a.g2 + clip, bytes);
        dx = make_srd(a.x + clip, bytes);
        voff = (unsigned)(cq * T + min(t0 + 4 * q, T - 4)) * 4u;
        const int th = hh ? t0 + NT : t0 - 1;
        hoff = (unsigned)(hc * T + min(max(th, 0), T - 1)) * 4u;
        okq = t0 + 4 * q < T;
        okh = th >= 0 && th < T;
    };
    const unsigned rowstep = (unsigned)(16 * T) * 4u;
    auto load_g = [&](int i) {                     // branch-free: clamped address, masked at the split
        sg[i] = __builtin_bit_cast(float4, buf_load4(dg, voff, rowstep * i));
        if (GTWO) sg2[i] = __builtin_bit_cast(float4, buf_load4(dg2, voff, rowstep * i));
    };
    auto load_x = [&](int i) { sx[i] = __builtin_bit_cast(float4, buf_load4(dx, voff, rowstep * i)); };
    auto load_halo = [&]() { hx = buf_load(dx, hoff, 0u); };

    const int tstep = gridDim.x;
    int tile = WM_XCD_MAP ? xcd_slot() : (int)blockIdx.x;            // grid <= ntiles
    set_tile(tile);
    bool okq_cur = okq, okh_cur = okh;              // masks of the tile whose operands are in the registers
#pragma unroll
    for (int i = 0; i < NV; ++i) { load_g(i); load_x(i); }
    load_halo();
    if (tid < 64) {
        Cs[tid] = GTWO ? a.ga[tid] : 0.f;
        Cs[64 + tid] = GTWO ? a.gb[tid] : 0.f;
        Cs[128 + tid] = GTWO ? a.gc[tid] : 0.f;
        Cs[192 + tid] = (XPRO == PRO_BNRELU) ? a.xa[tid] : 0.f;
        Cs[256 + tid] = (XPRO == PRO_BNRELU) ? a.xb[tid] : 0.f;
        Cs[320 + tid] = GTWO ? a.gb[64 + tid] : 0.f;        // low word of the BatchNorm-backward offset
    }
    // the element right of the right halo only feeds bits that the funnel shift drops: keep it defined in both images
    for (int i = tid; i < 2 * NP * 64; i += 256) Xb0[(i / (NP * 64)) * XIMG + (i % (NP * 64)) * PX + XO + NT + 1] = 0;
    __syncthreads();
    // per-thread prologue constants: the unit channels never change
    float gca[GTWO ? NV : 1], gcb[GTWO ? NV : 1], gcc[GTWO ? NV : 1], gcl[GTWO ? NV : 1], xca[NV], xcb[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = cq + 16 * i;
        if (GTWO) { gca[i] = Cs[c]; gcb[i] = Cs[64 + c]; gcc[i] = Cs[128 + c]; gcl[i] = Cs[320 + c]; }
        xca[i] = Cs[192 + c]; xcb[i] = Cs[256 + c];
    }
    const float hxa = Cs[192 + hc], hxb = Cs[256 + hc];

    // ---- the split of one staged unit (four consecutive time steps of one channel = two value pairs) in six stages
    float va = 0.f, vb = 0.f, vc = 0.f, vd = 0.f;
    unsigned pa0 = 0, pa1 = 0, pb0 = 0, pb1 = 0;
    float bflag = 1.f;                  // 0 while a clamped duplicate of the last tile is being split (it must not reach the bias sums)
    float wc = 0.f, wd = 0.f;           // second half of a two-tensor unit, parked while its registers are refilled
    auto g_pro_a = [&](int i) {         // first value pair of unit i: prologue + mask + bias sum; the second pair is parked raw
        const float4 v = sg[i];
        float x0 = v.x, x1 = v.y;
        vc = v.z; vd = v.w;
        if (GTWO) {
            const float4 w = sg2[i];
            x0 = pro_apply<PRO_BNBWD>(x0, w.x, gca[i], gcb[i], gcc[i], gcl[i]); x1 = pro_apply<PRO_BNBWD>(x1, w.y, gca[i], gcb[i], gcc[i], gcl[i]);
            wc = w.z; wd = w.w;
        }
        va = okq_cur ? x0 : 0.f; vb = okq_cur ? x1 : 0.f;
        bsum[i] = fmaf(bflag, va + vb, bsum[i]);
        asm volatile("" : "+v"(va), "+v"(vb), "+v"(vc), "+v"(vd), "+v"(wc), "+v"(wd), "+v"(bsum[i]));
    };
    auto g_pro_b = [&](int i) {         // second value pair
        float x2 = vc, x3 = vd;
        if (GTWO) { x2 = pro_apply<PRO_BNBWD>(x2, wc, gca[i], gcb[i], gcc[i], gcl[i]); x3 = pro_apply<PRO_BNBWD>(x3, wd, gca[i], gcb[i], gcc[i], gcl[i]); }
        vc = okq_cur ? x2 : 0.f; vd = okq_cur ? x3 : 0.f;
        bsum[i] = fmaf(bflag, vc + vd, bsum[i]);
        asm volatile("" : "+v"(vc), "+v"(vd), "+v"(bsum[i]));
    };
    auto x_pro = [&](int i) {
        float4 u = sx[i];
        if (XPRO != PRO_NONE) {
            u.x = pro_apply<XPRO>(u.x, 0.f, xca[i], xcb[i], 0.f); u.y = pro_apply<XPRO>(u.y, 0.f, xca[i], xcb[i], 0.f);
            u.z = pro_apply<XPRO>(u.z, 0.f, xca[i], xcb[i], 0.f); u.w = pro_apply<XPRO>(u.w, 0.f, xca[i], xcb[i], 0.f);
        }
        const bool ok = okq_cur;
        va = ok ? u.x : 0.f; vb = ok ? u.y : 0.f; vc = ok ? u.z : 0.f; vd = ok ? u.w : 0.f;
        asm volatile("" : "+v"(va), "+v"(vb), "+v"(vc), "+v"(vd));
    };
    auto h_pro = [&]() {
        float v = hx;
        if (XPRO != PRO_NONE) v = pro_apply<XPRO>(v, 0.f, hxa, hxb, 0.f);
        va = okh_cur ? v : 0.f; vb = 0.f;
        asm volatile("" : "+v"(va), "+v"(vb));
    };
    auto s1a = [&]() {
        const bf16x2 h_ = {(__bf16)va, (__bf16)vb};
        pa0 = __builtin_bit_cast(unsigned, h_);
        va -= __uint_as_float(pa0 << 16); vb -= __uint_as_float(pa0 & 0xffff0000u);
        asm volatile("" : "+v"(va), "+v"(vb), "+v"(pa0));
    };
    auto s2a = [&]() {
        const bf16x2 m_ = {(__bf16)va, (__bf16)vb};
        pa1 = __builtin_bit_cast(unsigned, m_);
        va -= __uint_as_float(pa1 << 16); vb -= __uint_as_float(pa1 & 0xffff0000u);
        asm volatile("" : "+v"(va), "+v"(vb), "+v"(pa1));
    };
    auto s1b = [&]() {
        const bf16x2 h_ = {(__bf16)vc, (__bf16)vd};
        pb0 = __builtin_bit_cast(unsigned, h_);
        vc -= __uint_as_float(pb0 << 16); vd -= __uint_as_float(pb0 & 0xffff0000u);
        asm volatile("" : "+v"(vc), "+v"(vd), "+v"(pb0));
    };
    auto s2b = [&]() {
        const bf16x2 m_ = {(__bf16)vc, (__bf16)vd};
        pb1 = __builtin_bit_cast(unsigned, m_);
        vc -= __uint_as_float(pb1 << 16); vd -= __uint_as_float(pb1 & 0xffff0000u);
        asm volatile("" : "+v"(vc), "+v"(vd), "+v"(pb1));
    };
    auto out4 = [&](unsigned short* dst, int stride_p) {         // last stage + the three 8-byte LDS writes
        const bf16x2 la = {(__bf16)va, (__bf16)vb}, lb = {(__bf16)vc, (__bf16)vd};
        *reinterpret_cast<uint2*>(dst) = make_uint2(pa0, pb0);
        *reinterpret_cast<uint2*>(dst + stride_p) = make_uint2(pa1, pb1);
        *reinterpret_cast<uint2*>(dst + 2 * stride_p) = make_uint2(__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb));
    };
    auto h_out = [&](unsigned short* X) {
        const bf16x2 la = {(__bf16)va, (__bf16)vb};
        const int o = hc * PX + (hh ? XO + NT : XO - 1);
        X[o] = (unsigned short)pa0; X[64 * PX + o] = (unsigned short)pa1; X[2 * 64 * PX + o] = (unsigned short)__builtin_bit_cast(unsigned, la);
    };
    // Side work of one tile as a sequence of slices: the operands in the registers are split into images (G, X) and the
    // consumed registers are refilled from the tile set_tile() last pointed at.  Units (G0, X0, G1, X1, ...), then the halo.
    // A gradient unit takes SG slices (prologue a | hi a + refill | mid a | prologue b | hi b | mid b | lo + LDS writes),
    // an input unit 6 (prologue | hi a + refill | mid a | hi b | mid b | lo + LDS writes), the halo 4.
    constexpr int SG = 7, SX = 6, NSLICE = NV * (SG + SX) + 4;
    auto side = [&](int u, unsigned short* G, unsigned short* X) __attribute__((always_inline)) {
        if (u < NV * (SG + SX)) {
            const int i = u / (SG + SX), r = u % (SG + SX);
            if (r < SG) {
                if (r == 0) g_pro_a(i);
                if (r == 1) { s1a(); load_g(i); }
                if (r == 2) s2a();
                if (r == 3) g_pro_b(i);
                if (r == 4) s1b();
                if (r == 5) s2b();
                if (r == 6) out4(G + (cq + 16 * i) * PG + 4 * q, 64 * PG);
            } else {
                const int st = r - SG;
                if (st == 0) x_pro(i);
                if (st == 1) { s1a(); load_x(i); }
                if (st == 2) s2a();
                if (st == 3) s1b();
                if (st == 4) s2b();
                if (st == 5) out4(X + (cq + 16 * i) * PX + XO + 4 * q, 64 * PX);
            }
        } else {
            const int st = u - NV * (SG + SX);
            if (st == 0) { h_pro(); load_halo(); }
            if (st == 1) s1a();
            if (st == 2) s2a();
            if (st == 3) { if (tid < 128) h_out(X); }
        }
    };
    static_assert(NSLICE <= 60, "72 MFMAs per tile, 12 of their slices build the shifted fragments");

    {   // first tile: split serially while the second is fetched
        set_tile(min(tile + tstep, ntiles - 1));
        const bool okq_n = okq, okh_n = okh;
#pragma unroll
        for (int u = 0; u < NSLICE; ++u) side(u, Gb0, Xb0);
        okq_cur = okq_n; okh_cur = okh_n;
    }
    __syncthreads();

    f32x16 acc[KW][2][2];
#pragma unroll
    for (int k = 0; k < KW; ++k)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[k][mt][nt][r] = 0.f;

    const int e0 = wave * 16 + 8 * half;             // this lane's 8 time steps of the tile (k index of the MFMA)
    int buf = 0;
#define FENCE __builtin_amdgcn_sched_barrier(0)
    while (tile < ntiles) {
        // registers: tile + tstep (split below into the other image pair); fetched below: tile + 2 tstep
        bflag = (tile + tstep < ntiles) ? 1.f : 0.f;
        set_tile(min(tile + 2 * tstep, ntiles - 1));
        const bool okq_n = okq, okh_n = okh;
        const unsigned short* Gc = Gb0 + buf * GIMG;
        const unsigned short* Xc = Xb0 + buf * XIMG;
        unsigned short* Gn = Gb0 + (buf ^ 1) * GIMG;
        unsigned short* Xn = Xb0 + (buf ^ 1) * XIMG;
        bf16x8 A[2][NP];
        uint4 f[2][NP];
        unsigned Lw[2][NP], Rw[2][NP];
        bf16x8 Bl[NP], Br[NP];
        auto read_b = [&](int nt) {
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const unsigned short* xr = Xc + (p * 64 + nt * 32 + l31) * PX + XO + e0;
                f[nt][p] = *reinterpret_cast<const uint4*>(xr);
                Lw[nt][p] = *reinterpret_cast<const unsigned*>(xr - 2);
                Rw[nt][p] = *reinterpret_cast<const unsigned*>(xr + 8);
            }
        };
        auto shl = [&](int nt, int p) {      // fragment shifted to t - 1
            const uint4 g = f[nt][p];
            uint4 s_ = make_uint4(__builtin_amdgcn_alignbit(g.x, Lw[nt][p], 16), __builtin_amdgcn_alignbit(g.y, g.x, 16),
                                  __builtin_amdgcn_alignbit(g.z, g.y, 16), __builtin_amdgcn_alignbit(g.w, g.z, 16));
            asm volatile("" : "+v"(s_.x), "+v"(s_.y), "+v"(s_.z), "+v"(s_.w));      // built in this slice, not at its first use
            Bl[p] = __builtin_bit_cast(bf16x8, s_);
        };
        auto shr = [&](int nt, int p) {      // fragment shifted to t + 1
            const uint4 g = f[nt][p];
            uint4 s_ = make_uint4(__builtin_amdgcn_alignbit(g.y, g.x, 16), __builtin_amdgcn_alignbit(g.z, g.y, 16),
                                  __builtin_amdgcn_alignbit(g.w, g.z, 16), __builtin_amdgcn_alignbit(Rw[nt][p], g.w, 16));
            asm volatile("" : "+v"(s_.x), "+v"(s_.y), "+v"(s_.z), "+v"(s_.w));
            Br[p] = __builtin_bit_cast(bf16x8, s_);
        };
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                A[mt][p] = *reinterpret_cast<const bf16x8*>(Gc + (p * 64 + mt * 32 + l31) * PG + e0);
        read_b(0);
#pragma unroll
        for (int m = 0; m < 72; ++m) {
            const int nt = m / 36, tg = (m % 36) / 12, mt = (m % 12) / 6, j = m % 6;
            const int pa = (j == 0 || j == 4) ? 1 : (j == 2 ? 2 : 0), pb = (j == 0 || j == 3) ? 1 : (j == 1 ? 2 : 0);
            FENCE;
            if (tg == 0)
                acc[1][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt][pa], __builtin_bit_cast(bf16x8, f[nt][pb]), acc[1][mt][nt], 0, 0, 0);
            else if (tg == 1)
                acc[0][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt][pa], Bl[pb], acc[0][mt][nt], 0, 0, 0);
            else
                acc[2][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt][pa], Br[pb], acc[2][mt][nt], 0, 0, 0);
            FENCE;
            // ---- the slice behind MFMA m
            const int mm = m % 36;
            if (mm < 3) shl(nt, mm);
            else if (mm < 6) shr(nt, mm - 3);
            else {
                const int u = (nt == 0) ? mm - 6 : 30 + (mm - 6);          // 0..59: side-work slice index
                if (u < NSLICE) side(u, Gn, Xn);
            }
            if (m == 30) read_b(1);
            FENCE;
        }
        okq_cur = okq_n; okh_cur = okh_n;
        lds_barrier();
        tile += tstep;
        buf ^= 1;
    }
#undef FENCE

    float* out = a.partial + (size_t)blockIdx.x * (KW * 4096 + 64);
    float* red = reinterpret_cast<float*>(smem_raw);          // KW*4096 floats = 48 KB <= the images (120 KB)
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int k = 0; k < KW; ++k)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int o = (k * 64 + mt * 32 + mfma_row(r, half)) * 64 + nt * 32 + l31;
                            red[o] = (w == 0) ? acc[k][mt][nt][r] : red[o] + acc[k][mt][nt][r];
                        }
        }
        __syncthreads();
    }
    for (int i = tid; i < KW * 4096; i += 256) out[i] = red[i];
#pragma unroll
    for (int i = 0; i < NV; ++i) {       // the 16 lanes of a channel row sit side by side
        float v = bsum[i];
        v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
        if (q == 0) out[KW * 4096 + cq + 16 * i] = v;
    }
}

template <int GPRO, int XPRO>
int launch_wgrad64bfp(const Wgrad64Args& a, int* grid_out, hipStream_t stream) {
    constexpr size_t lds = (size_t)(2 * 3 * 64 * 72 + 2 * 3 * 64 * 88) * 2 + 6 * 64 * sizeof(float);
    static wm::DevOnce attr_done;
    auto kern = wgrad64bfp_kernel<GPRO, XPRO>;
    if (!wm::dev_done(attr_done)) {
        WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        wm::dev_mark(attr_done);
    }
    const int ntiles = a.B * ((a.T + 63) / 64);
    const int grid = ntiles < kNumCU ? ntiles : kNumCU;
    *grid_out = grid;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
    WM_CHECK_LAUNCH();
    return 0;
}

// Output-split build of the same kernel: a wave keeps one 32 x 32 block per tap instead of the whole 64 x 64 x 3 result, so the
// workgroup needs ~200 registers per lane and can be co-resident with the LSTM recurrence kernels (side-stream overlap).
template <int GPRO, int XPRO>
__global__ __launch_bounds__(256) void wgrad64bf_small_kernel(Wgrad64Args a) {
    constexpr int KW = 3, NT = 128, NP = 3, PG = 136, PX = 152, XO = 8;   // X element index = (t - t0) + XO
    constexpr int QR = NT / 4, NV = 64 * QR / 256;
    constexpr bool GTWO = (GPRO == PRO_BNBWD);
    extern __shared__ __align__(16) unsigned char smem_raw[];
    unsigned short* Gb = reinterpret_cast<unsigned short*>(smem_raw);          // [NP][64][PG]
    unsigned short* Xb = Gb + NP * 64 * PG;                                    // [NP][64][PX]
    float* Cs = reinterpret_cast<float*>(Xb + NP * 64 * PX);                   // [5][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int T = a.T;
    const int tilesPerClip = (T + NT - 1) / NT, ntiles = a.B * tilesPerClip;

    float4 sg[NV], sg2[GTWO ? NV : 1], sx[NV];
    float hx;
    float bsum[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) bsum[i] = 0.f;

    // global -> register staging in NV + 1 pieces: the main loop issues one piece per MFMA block (a 64..96-KB burst per
    // CU backs up the memory pipeline and blocks the wave at issue for thousands of cycles; see conv64bf3_kernel)
    auto load_piece = [&](int tile, int i) {    // branch-free
        const int b = tile / tilesPerClip, t0 = (tile - b * tilesPerClip) * NT;
        const size_t base = (size_t)b * 64 * T;
        if (i < NV) {
            const int idx = tid + i * 256, c = idx / QR, q = idx % QR;
            const int t = min(t0 + 4 * q, T - 4);
            sg[i] = *reinterpret_cast<const float4*>(a.g + base + (size_t)c * T + t);
            if (GTWO) sg2[i] = *reinterpret_cast<const float4*>(a.g2 + base + (size_t)c * T + t);
            sx[i] = *reinterpret_cast<const float4*>(a.x + base + (size_t)c * T + t);
        } else {
            const int hc = (tid & 127) >> 1, hh = tid & 1;
            hx = a.x[base + (size_t)hc * T + min(max(hh ? t0 + NT : t0 - 1, 0), T - 1)];
        }
    };
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int i = 0; i <= NV; ++i) load_piece(tile, i);
    };
    auto put4 = [&](unsigned short* dst, int stride_p, float v0, float v1, float v2, float v3) {
        unsigned a0, a1, a2, b0, b1, b2;
        split3_pair(v0, v1, a0, a1, a2);
        split3_pair(v2, v3, b0, b1, b2);
        *reinterpret_cast<uint2*>(dst) = make_uint2(a0, b0);
        *reinterpret_cast<uint2*>(dst + stride_p) = make_uint2(a1, b1);
        *reinterpret_cast<uint2*>(dst + 2 * stride_p) = make_uint2(a2, b2);
    };
    auto write_tile = [&](int tile) {
        const int b = tile / tilesPerClip, t0 = (tile - b * tilesPerClip) * NT;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * 256, c = idx / QR, q = idx % QR, t = t0 + 4 * q;
            float4 v = sg[i], u = sx[i];
            if (GPRO == PRO_BNBWD) {
                const float ca = Cs[c], cb = Cs[64 + c], cc = Cs[128 + c], cl = Cs[320 + c];
                const float4 w = sg2[i];
                v.x = pro_apply<PRO_BNBWD>(v.x, w.x, ca, cb, cc, cl); v.y = pro_apply<PRO_BNBWD>(v.y, w.y, ca, cb, cc, cl);
                v.z = pro_apply<PRO_BNBWD>(v.z, w.z, ca, cb, cc, cl); v.w = pro_apply<PRO_BNBWD>(v.w, w.w, ca, cb, cc, cl);
            }
            if (XPRO != PRO_NONE) {
                const float ca = (XPRO == PRO_ADDVEC) ? a.xa[b * 64 + c] : Cs[192 + c];
                const float cb = Cs[256 + c];
                u.x = pro_apply<XPRO>(u.x, 0.f, ca, cb, 0.f); u.y = pro_apply<XPRO>(u.y, 0.f, ca, cb, 0.f);
                u.z = pro_apply<XPRO>(u.z, 0.f, ca, cb, 0.f); u.w = pro_apply<XPRO>(u.w, 0.f, ca, cb, 0.f);
            }
            if (t >= T) { v = make_float4(0.f, 0.f, 0.f, 0.f); u = v; }
            bsum[i] += (v.x + v.y) + (v.z + v.w);
            put4(Gb + c * PG + 4 * q, 64 * PG, v.x, v.y, v.z, v.w);
            put4(Xb + c * PX + XO + 4 * q, 64 * PX, u.x, u.y, u.z, u.w);
        }
        if (tid < 128) {
            const int hc = tid >> 1, hh = tid & 1, t = hh ? t0 + NT : t0 - 1;
            float v = hx;
            if (XPRO != PRO_NONE) {
                const float ca = (XPRO == PRO_ADDVEC) ? a.xa[b * 64 + hc] : Cs[192 + hc];
                v = pro_apply<XPRO>(v, 0.f, ca, Cs[256 + hc], 0.f);
            }
            if (t < 0 || t >= T) v = 0.f;
            unsigned p0, p1, p2;
            split3_pair(v, 0.f, p0, p1, p2);
            const int o = hc * PX + (hh ? XO + NT : XO - 1);
            Xb[o] = (unsigned short)p0; Xb[64 * PX + o] = (unsigned short)p1; Xb[2 * 64 * PX + o] = (unsigned short)p2;
        }
    };

    int tile = WM_XCD_MAP ? xcd_slot() : (int)blockIdx.x;
    if (tile < ntiles) load_tile(tile);
    if (tid < 64) {
        Cs[tid] = GTWO ? a.ga[tid] : 0.f;
        Cs[64 + tid] = GTWO ? a.gb[tid] : 0.f;
        Cs[128 + tid] = GTWO ? a.gc[tid] : 0.f;
        Cs[192 + tid] = (XPRO == PRO_BNRELU) ? a.xa[tid] : 0.f;
        Cs[256 + tid] = (XPRO == PRO_BNRELU) ? a.xb[tid] : 0.f;
        Cs[320 + tid] = GTWO ? a.gb[64 + tid] : 0.f;        // low word of the BatchNorm-backward offset
    }
    // the element right of the right halo is read by the funnel shift of the last fragment: keep it defined
    for (int i = tid; i < NP * 64; i += 256) Xb[i * PX + XO + NT + 1] = 0;
    __syncthreads();
    if (tile < ntiles) write_tile(tile);
    __syncthreads();

    // wave (mt, nt) owns ONE 32 x 32 block of every tap's matrix (48 accumulator registers instead of 192) and walks the whole
    // 128-step tile: ~200 registers per lane in all, so a workgroup fits beside the 152-register LSTM recurrence waves
    const int mt = wave & 1, nt = wave >> 1;
    f32x16 acc[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

    while (tile < ntiles) {
        const int next = tile + gridDim.x;
        const int nextc = min(next, ntiles - 1);         // clamped: a tile past the end is loaded (valid memory) but never written
#pragma unroll
        for (int kb = 0; kb < NT / 16; ++kb) {           // the tile's 8 k-blocks of 16 time steps
            const int e0 = kb * 16 + 8 * half;
            bf16x8 A[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) A[p] = *reinterpret_cast<const bf16x8*>(Gb + (p * 64 + mt * 32 + l31) * PG + e0);
            bf16x8 Bc[NP], Bl[NP], Br[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const unsigned short* xr = Xb + (p * 64 + nt * 32 + l31) * PX + XO + e0;
                const uint4 f = *reinterpret_cast<const uint4*>(xr);
                const unsigned L = *reinterpret_cast<const unsigned*>(xr - 2), R = *reinterpret_cast<const unsigned*>(xr + 8);
                const uint4 fl = make_uint4(__builtin_amdgcn_alignbit(f.x, L, 16), __builtin_amdgcn_alignbit(f.y, f.x, 16),
                                            __builtin_amdgcn_alignbit(f.z, f.y, 16), __builtin_amdgcn_alignbit(f.w, f.z, 16));
                const uint4 fr = make_uint4(__builtin_amdgcn_alignbit(f.y, f.x, 16), __builtin_amdgcn_alignbit(f.z, f.y, 16),
                                            __builtin_amdgcn_alignbit(f.w, f.z, 16), __builtin_amdgcn_alignbit(R, f.w, 16));
                Bc[p] = __builtin_bit_cast(bf16x8, f);
                Bl[p] = __builtin_bit_cast(bf16x8, fl);
                Br[p] = __builtin_bit_cast(bf16x8, fr);
            }
#define WM_MM6(ACC, BB)                                                                              \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], BB[1], ACC, 0, 0, 0);                         \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], BB[2], ACC, 0, 0, 0);                         \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[2], BB[0], ACC, 0, 0, 0);                         \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], BB[1], ACC, 0, 0, 0);                         \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], BB[0], ACC, 0, 0, 0);                         \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], BB[0], ACC, 0, 0, 0);
            WM_MM6(acc[0], Bl)
            WM_MM6(acc[1], Bc)
            WM_MM6(acc[2], Br)
#undef WM_MM6
            load_piece(nextc, kb);                       // pieces 0..7 ride along the 8 k-blocks
        }
        load_piece(nextc, NV);
        __syncthreads();
        if (next < ntiles) write_tile(next);
        __syncthreads();
        tile = next;
    }

    float* out = a.partial + (size_t)blockIdx.x * (KW * 4096 + 64);
#pragma unroll
    for (int k = 0; k < KW; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[(k * 64 + mt * 32 + mfma_row(r, half)) * 64 + nt * 32 + l31] = acc[k][r];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float v = half_wave_sum(bsum[i]);
        if (l31 == 0) out[KW * 4096 + (tid >> 5) + 8 * i] = v;
    }
}

template <int GPRO, int XPRO>
int launch_wgrad64bf_small(const Wgrad64Args& a, int* grid_out, hipStream_t stream) {
    constexpr size_t lds = (size_t)(3 * 64 * 136 + 3 * 64 * 152) * 2 + 6 * 64 * sizeof(float);
    static wm::DevOnce attr_done;
    auto kern = wgrad64bf_small_kernel<GPRO, XPRO>;
    if (!wm::dev_done(attr_done)) {
        WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        wm::dev_mark(attr_done);
    }
    const int ntiles = a.B * ((a.T + 127) / 128);
    const int grid = ntiles < kNumCU ? ntiles : kNumCU;
    *grid_out = grid;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
    WM_CHECK_LAUNCH();
    return 0;
}

template <int GPRO, int XPRO>
int launch_wgrad64bf(const Wgrad64Args& a, int* grid_out, hipStream_t stream) {
#if WM_WGRAD_PIPE
    return launch_wgrad64bfp<GPRO, XPRO>(a, grid_out, stream);
#endif
    constexpr size_t lds = (size_t)(3 * 64 * 136 + 3 * 64 * 152) * 2 + 6 * 64 * sizeof(float);
    static wm::DevOnce attr_done;
    auto kern = wgrad64bf_kernel<GPRO, XPRO>;
    if (!wm::dev_done(attr_done)) {
        WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        wm::dev_mark(attr_done);
    }
    const int ntiles = a.B * ((a.T + 127) / 128);
    const int grid = ntiles < kNumCU ? ntiles : kNumCU;
    *grid_out = grid;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
    WM_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// bf16x6 weight gradient of the 7-tap ConvTranspose1d: G[tap][out][in] = sum_t g[out][t] x[in][t + tap - 3].
// Same LDS scheme as wgrad64bf_kernel ([channel][time] 3-piece images, fragments = aligned ds_read_b128 along time).
// Wave (nt, kh) owns the 32 input-channel columns nt and half kh of the tile's time range for ALL seven taps and both
// output halves (224 accumulator registers): per 16-step k-block it reads one 24-element window per piece and builds
// the seven shifted B fragments from it in registers (dword selects for even shifts, v_alignbit for odd ones), so the
// matrix phase needs 15 LDS reads per 84 MFMAs.  The two time halves meet in the final slab reduction.
// ---------------------------------------------------------------------------------------------
// H = true: f16 two-piece build (three products per product): a.ga = {gs, 1 / gs}, the gradient is multiplied by gs before the split and
// clamped to the f16 range, x is split unscaled, the sums leave times 1 / gs (the bias sums are formed from the unscaled gradient)
template <int XPRO, bool H = false>
__global__ __launch_bounds__(256) void wgrad64bf7_kernel(Wgrad64Args a) {
    constexpr int KW = 7, PAD = 3, NT = 128, NP = H ? 2 : 3, PG = 136, PX = 152, XO = 8;
    constexpr int QR = NT / 4, NV = 64 * QR / 256;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    unsigned short* Gb = reinterpret_cast<unsigned short*>(smem_raw);          // [NP][64][PG]
    unsigned short* Xb = Gb + NP * 64 * PG;                                    // [NP][64][PX]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int nt = wave & 1, mt = wave >> 1;           // this wave's 32 x 32 block of every tap's [out][in] matrix
    const int T = a.T;
    const int tilesPerClip = (T + NT - 1) / NT, ntiles = a.B * tilesPerClip;
    const int c0 = tid >> 5, q = tid & 31;             // staging: rows c0 + 8 i, float4 column q

    float4 sg[NV], sx[NV];
    float xv[NV], hx[2], hv[2];
    float bsum[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) bsum[i] = 0.f;
    const float gs = (H && a.ga) ? a.ga[0] : 1.f, ginv = (H && a.ga) ? a.ga[1] : 1.f;

    auto halo_of = [&](int k, int t0, int& c, int& e, int& t) {   // k-th halo element of this thread: channel, image element, time
        const int idx = min(tid + k * 256, 64 * 2 * PAD - 1);
        c = idx / (2 * PAD);
        const int h = idx % (2 * PAD);
        e = (h < PAD) ? XO - PAD + h : XO + NT + (h - PAD);
        t = t0 + e - XO;
    };
    auto load_piece = [&](int tile, int i) {        // branch-free (clamped addresses); piece NV = halo
        const int b = tile / tilesPerClip, t0 = (tile - b * tilesPerClip) * NT;
        const size_t base = (size_t)b * 64 * T;
        if (i < NV) {
            const int c = c0 + 8 * i, t = min(t0 + 4 * q, T - 4);
            sg[i] = *reinterpret_cast<const float4*>(a.g + base + (size_t)c * T + t);
            sx[i] = *reinterpret_cast<const float4*>(a.x + base + (size_t)c * T + t);
            if (XPRO == PRO_ADDVEC) xv[i] = a.xa[b * 64 + c];
        } else {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                int c, e, t;
                halo_of(k, t0, c, e, t);
                hx[k] = a.x[base + (size_t)c * T + min(max(t, 0), T - 1)];
                if (XPRO == PRO_ADDVEC) hv[k] = a.xa[b * 64 + c];
            }
        }
    };
    auto put4 = [&](unsigned short* dst, int stride_p, float v0, float v1, float v2, float v3) {
        if (H) {
            const h16x2 ha = __builtin_convertvector(f32x2{v0, v1}, h16x2), hb = __builtin_convertvector(f32x2{v2, v3}, h16x2);
            const h16x2 la = __builtin_convertvector(f32x2{v0 - (float)ha.x, v1 - (float)ha.y}, h16x2);
            const h16x2 lb = __builtin_convertvector(f32x2{v2 - (float)hb.x, v3 - (float)hb.y}, h16x2);
            *reinterpret_cast<uint2*>(dst) = make_uint2(__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb));
            *reinterpret_cast<uint2*>(dst + stride_p) = make_uint2(__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb));
            return;
        }
        unsigned a0, a1, a2, b0, b1, b2;
        split3_pair(v0, v1, a0, a1, a2);
        split3_pair(v2, v3, b0, b1, b2);
        *reinterpret_cast<uint2*>(dst) = make_uint2(a0, b0);
        *reinterpret_cast<uint2*>(dst + stride_p) = make_uint2(a1, b1);
        *reinterpret_cast<uint2*>(dst + 2 * stride_p) = make_uint2(a2, b2);
    };
    auto write_tile = [&](int tile) {
        const int t0 = (tile % tilesPerClip) * NT;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = c0 + 8 * i, t = t0 + 4 * q;
            float4 v = sg[i], u = sx[i];
            if (XPRO == PRO_ADDVEC) { u.x += xv[i]; u.y += xv[i]; u.z += xv[i]; u.w += xv[i]; }
            if (t >= T) { v = make_float4(0.f, 0.f, 0.f, 0.f); u = v; }
            bsum[i] += (v.x + v.y) + (v.z + v.w);
            if (H) {
                v.x = __builtin_amdgcn_fmed3f(v.x * gs, -6.0e4f, 6.0e4f); v.y = __builtin_amdgcn_fmed3f(v.y * gs, -6.0e4f, 6.0e4f);
                v.z = __builtin_amdgcn_fmed3f(v.z * gs, -6.0e4f, 6.0e4f); v.w = __builtin_amdgcn_fmed3f(v.w * gs, -6.0e4f, 6.0e4f);
            }
            put4(Gb + c * PG + 4 * q, 64 * PG, v.x, v.y, v.z, v.w);
            put4(Xb + c * PX + XO + 4 * q, 64 * PX, u.x, u.y, u.z, u.w);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (tid + k * 256 < 64 * 2 * PAD) {
                int c, e, t;
                halo_of(k, t0, c, e, t);
                float v = hx[k];
                if (XPRO == PRO_ADDVEC) v += hv[k];
                if (t < 0 || t >= T) v = 0.f;
                const int o = c * PX + e;
                if (H) {
                    const _Float16 h_ = (_Float16)v;
                    const _Float16 l_ = (_Float16)(v - (float)h_);
                    Xb[o] = __builtin_bit_cast(unsigned short, h_); Xb[64 * PX + o] = __builtin_bit_cast(unsigned short, l_);
                } else {
                    unsigned p0, p1, p2;
                    split3_pair(v, 0.f, p0, p1, p2);
                    Xb[o] = (unsigned short)p0; Xb[64 * PX + o] = (unsigned short)p1; Xb[2 * 64 * PX + o] = (unsigned short)p2;
                }
            }
        }
    };

    int tile = WM_XCD_MAP ? xcd_slot() : (int)blockIdx.x;          // grid <= ntiles
#pragma unroll
    for (int i = 0; i <= NV; ++i) load_piece(tile, i);
    // window elements outside [XO - 3, XO + NT + 3) are read but never reach an MFMA: give them a defined value once
    for (int i = tid; i < NP * 64 * PX; i += 256) Xb[i] = 0;
    __syncthreads();
    write_tile(tile);
    __syncthreads();

    f32x16 acc[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

    auto mma = [&](const u32x4& A_, const u32x4& B_, f32x16 c) -> f32x16 {
        if (H) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, A_), __builtin_bit_cast(h16x8, B_), c, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A_), __builtin_bit_cast(bf16x8, B_), c, 0, 0, 0);
    };
    while (tile < ntiles) {
        const int next = tile + gridDim.x;
        const int nextc = min(next, ntiles - 1);         // clamped: loaded (valid memory) but never written
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {                 // the tile's 128 time steps = 8 k-blocks of 16
            const int e0 = kb * 16 + 8 * half;
            u32x4 A[NP];
            unsigned W[NP][12];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                A[p] = *reinterpret_cast<const u32x4*>(Gb + (p * 64 + mt * 32 + l31) * PG + e0);
                const unsigned short* xr = Xb + (p * 64 + nt * 32 + l31) * PX + XO + e0;
                const uint4 w0 = *reinterpret_cast<const uint4*>(xr - 8), w1 = *reinterpret_cast<const uint4*>(xr),
                            w2 = *reinterpret_cast<const uint4*>(xr + 8);
                W[p][0] = w0.x; W[p][1] = w0.y; W[p][2] = w0.z; W[p][3] = w0.w;
                W[p][4] = w1.x; W[p][5] = w1.y; W[p][6] = w1.z; W[p][7] = w1.w;
                W[p][8] = w2.x; W[p][9] = w2.y; W[p][10] = w2.z; W[p][11] = w2.w;
            }
#pragma unroll
            for (int tap = 0; tap < KW; ++tap) {
                // fragment = 8 elements from window element 8 + (tap - PAD): dword d0, odd start -> funnel shift
                constexpr int dummy = 0; (void)dummy;
                const int st = 8 + tap - PAD, d0 = st >> 1;
                u32x4 Bt[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    uint4 f;
                    if (st & 1)
                        f = make_uint4(__builtin_amdgcn_alignbit(W[p][d0 + 1], W[p][d0], 16), __builtin_amdgcn_alignbit(W[p][d0 + 2], W[p][d0 + 1], 16),
                                       __builtin_amdgcn_alignbit(W[p][d0 + 3], W[p][d0 + 2], 16), __builtin_amdgcn_alignbit(W[p][d0 + 4], W[p][d0 + 3], 16));
                    else
                        f = make_uint4(W[p][d0], W[p][d0 + 1], W[p][d0 + 2], W[p][d0 + 3]);
                    Bt[p] = u32x4{f.x, f.y, f.z, f.w};
                }
                if (H) {                                 // lo hi, hi lo, hi hi
                    acc[tap] = mma(A[1], Bt[0], acc[tap]); acc[tap] = mma(A[0], Bt[1], acc[tap]); acc[tap] = mma(A[0], Bt[0], acc[tap]);
                } else {
                    acc[tap] = mma(A[1], Bt[1], acc[tap]); acc[tap] = mma(A[0], Bt[NP - 1], acc[tap]); acc[tap] = mma(A[NP - 1], Bt[0], acc[tap]);
                    acc[tap] = mma(A[0], Bt[1], acc[tap]); acc[tap] = mma(A[1], Bt[0], acc[tap]); acc[tap] = mma(A[0], Bt[0], acc[tap]);
                }
                if (tap == 0) load_piece(nextc, kb);                 // pieces 0..7 ride along the matrix phase
            }
        }
        load_piece(nextc, NV);
        __syncthreads();
        if (next < ntiles) write_tile(next);
        __syncthreads();
        tile = next;
    }

    float* out = a.partial + (size_t)blockIdx.x * (KW * 4096 + 64);
    // every wave owns its block of the slab: straight to global (128-B row segments)
#pragma unroll
    for (int k = 0; k < KW; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[(k * 64 + mt * 32 + mfma_row(r, half)) * 64 + nt * 32 + l31] = H ? acc[k][r] * ginv : acc[k][r];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float v = half_wave_sum(bsum[i]);
        if (l31 == 0) out[KW * 4096 + c0 + 8 * i] = v;
    }
}

template <int XPRO, bool H = false>
int launch_wgrad64bf7(const Wgrad64Args& a, int* grid_out, hipStream_t stream) {
    constexpr size_t lds = (size_t)((H ? 2 : 3) * 64 * 136 + (H ? 2 : 3) * 64 * 152) * 2;
    static wm::DevOnce attr_done;
    auto kern = wgrad64bf7_kernel<XPRO, H>;
    if (!wm::dev_done(attr_done)) {
        WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        wm::dev_mark(attr_done);
    }
    const int ntiles = a.B * ((a.T + 127) / 128);
    const int grid = ntiles < kNumCU ? ntiles : kNumCU;
    *grid_out = grid;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
    WM_CHECK_LAUNCH();
    return 0;
}

// f16 two-piece image of a k3 weight for dwgrad64bf_kernel<..., H = true>: [2 pieces][3 taps][64 out][64 in] f16 of w * ws, ws = the power
// of two that brings max |w| to [2^9, 2^10), followed by {ws, 1 / ws} as two floats.  One workgroup (12 288 values).
__global__ __launch_bounds__(1024) void pack_w64_h_kernel(const float* __restrict__ w, const float* __restrict__ row_scale,
                                                          unsigned short* __restrict__ wph, int mode) {
    __shared__ float red[16];
    float v[12];                                    // the thread's 12 of the 12 288 values: one trip to memory
    float mx = 0.f;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const int i = threadIdx.x + 1024 * j, tap = i / 4096, out = (i / 64) % 64, in = i % 64;
        v[j] = (mode == 0) ? w[(out * 64 + in) * 3 + tap] : w[(in * 64 + out) * 3 + (2 - tap)];
        if (row_scale) v[j] *= row_scale[out];      // a per-output-channel factor (folded BatchNorm scale) rides in the weights
        mx = fmaxf(mx, fabsf(v[j]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) m = fmaxf(m, red[i]);
    float ws = 1.f;
    if (m > 0.f && m < 3.0e38f) ws = exp2f(floorf(log2f(1023.f / m)));
    ws = fminf(fmaxf(ws, 1.0e-30f), 1.0e30f);
    if (threadIdx.x == 0) {
        float* tail = reinterpret_cast<float*>(wph + 2 * 3 * 4096);
        tail[0] = ws; tail[1] = 1.f / ws;
    }
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const int i = threadIdx.x + 1024 * j;
        const float x = v[j] * ws;
        const _Float16 hi = (_Float16)x;
        const _Float16 lo = (_Float16)(x - (float)hi);
        wph[i] = __builtin_bit_cast(unsigned short, hi);
        wph[3 * 4096 + i] = __builtin_bit_cast(unsigned short, lo);
    }
}

// the same for the 7-tap ConvTranspose1d weight w[in][out][7] (conv64bf7p_kernel<.., H = true>): [2 pieces][7 taps][64 out][64 in] f16 +
// {ws, 1 / ws}; mode 2 forward, 3 data gradient (as wm_pack_w64_bf7)
__global__ __launch_bounds__(1024) void pack_w64_h7_kernel(const float* __restrict__ w, unsigned short* __restrict__ wph, int mode) {
    __shared__ float red[16];
    float v[28];
    float mx = 0.f;
#pragma unroll
    for (int j = 0; j < 28; ++j) {
        const int i = threadIdx.x + 1024 * j, tap = i / 4096, out = (i / 64) % 64, in = i % 64;
        v[j] = (mode == 2) ? w[(in * 64 + out) * 7 + (6 - tap)] : w[(out * 64 + in) * 7 + tap];
        mx = fmaxf(mx, fabsf(v[j]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) m = fmaxf(m, red[i]);
    float ws = 1.f;
    if (m > 0.f && m < 3.0e38f) ws = exp2f(floorf(log2f(1023.f / m)));
    ws = fminf(fmaxf(ws, 1.0e-30f), 1.0e30f);
    if (threadIdx.x == 0) {
        float* tail = reinterpret_cast<float*>(wph + 2 * 7 * 4096);
        tail[0] = ws; tail[1] = 1.f / ws;
    }
#pragma unroll
    for (int j = 0; j < 28; ++j) {
        const int i = threadIdx.x + 1024 * j;
        const float x = v[j] * ws;
        const _Float16 hi = (_Float16)x;
        const _Float16 lo = (_Float16)(x - (float)hi);
        wph[i] = __builtin_bit_cast(unsigned short, hi);
        wph[7 * 4096 + i] = __builtin_bit_cast(unsigned short, lo);
    }
}

// ---------------------------------------------------------------------------------------------
// Data gradient AND weight gradient of a k3 convolution in ONE launch (ResBlock backward, py/main16.py:112-125 under
// autograd).  Both read the same gradient frames (dz, and y for the BatchNorm-backward rebuild g = A dz + B + C y): as two
// launches the pair moves 7 frames, fused 4 (conv2: dz2 y2 y1 in, dz1 out) or 5 (conv1: dz1 y1 x dz2 in, dx out).
// A workgroup walks 64-step tiles.  The staging thread holds two channels x four steps of the gradient, rebuilds g once and
// splits it once: time pairs of bf16 pieces go into the [channel][time] image of the weight gradient (phase A); the
// [time][channel] image of the data gradient takes channel pairs, which are the same pieces re-paired by v_perm (phase B).
// The data gradient is accumulated transposed (rows = time, columns = channel), so that a lane's accumulator quads are four
// consecutive steps of one channel: epilogue operand and result move as dwordx4.  Per tile and wave:
//   phase A: 72 MFMAs of the data gradient (weight fragments of its 32 output rows resident in registers, as conv64bf3)
//            out of image D(i); side work: tile i+1 -> images G'(i+1), X'(i+1) (double-buffered), the epilogue operand of tile i
//   barrier
//   phase B: 72 MFMAs of the weight gradient (one 32 x 32 block of each tap per wave, as wgrad64bf_small) out of G'(i), X'(i);
//            side work: tile i+1 -> image D (single-buffered), epilogue of the data gradient of tile i, refill for tile i+2
//   barrier
// One slice of side work pinned behind every MFMA.  LDS: 28 + 2 x 27 + 2 x 33 KB.  T % 64 == 0.
// ---------------------------------------------------------------------------------------------
struct DWArgs {
    const float* g;  const float* g2;                   // gradient dz and the BN input y [B,64,T]
    const float* ga; const float* gb; const float* gc;  // g = ga[c] dz + gb[c] (+ gb[64+c]) + gc[c] y
    const void* wp;                                     // data-gradient weight image (wm_pack_w64_bf mode 1)
    const float* x;  const float* xa; const float* xb;  // weight-gradient input operand (+ BN+ReLU constants when XPRO = BNRELU)
    const float* e1; const float* ea; const float* eb;  // data-gradient epilogue tensor (RELUMASK: pre-BN activation + its scale/shift; ADD: addend)
    float* y;                                           // data gradient out [B,64,T]
    float* stats;                                       // [grid][2][64] (RELUMASK) or null
    float* partial;                                     // weight-gradient slabs [grid][3*4096 + 64]
    int B, T;
    const unsigned* gmask;                              // GM: sign bits of the ReLU the incoming gradient passes (bit t % 32 of dword
                                                        // [row][t / 32]): applied to g (conv2 pair) / to e1 (conv1 pair) on load
    const unsigned* pmask; const float* py2;            // EPI_ADDSTATS: sign bits / pre-BatchNorm activation y2 of the block BEFORE this
                                                        // one: y = (data gradient + e1) masked, stats = (sum y, sum y py2)
    const float* gscale;                                // H (f16 two-piece split): {gs, 1 / gs}, the power-of-two scale of the rebuilt
                                                        // gradient (wm_bn_bwd_finalize); the weight image carries its own scale
    float* dzmax;                                       // H, epi 1 / 2 / 8: max |y| per workgroup [grid] (sizes the NEXT launch's scale)
};

// H = false: bf16 three-piece split, six piece products per product (bf16x6).  H = true: f16 TWO-piece split (hi = RNE_f16(x s),
// lo = RNE_f16(x s - hi): 22 bits), three products hi hi + hi lo + lo hi on v_mfma_f32_32x32x16_f16 -- half the matrix work, two
// thirds of the split / LDS work.  The f16 range is met by power-of-two scales: the weights' (chosen by the pack kernel from
// max |w|, stored behind the image), the gradient's (gs from wm_bn_bwd_finalize: max |A| max |dz| -> 2^9, applied through the
// BatchNorm-backward constants, i.e. for free; the rebuilt value is clamped to +-6e4 so that a pathological element saturates
// instead of becoming an infinity); activations need none.  Results are unscaled in the epilogue / at the slab write.
template <int EPI, int XPRO, bool GM, bool H>
__global__ __launch_bounds__(256) void dwgrad64bf_kernel(DWArgs a) {
    static_assert((EPI == EPI_RELUMASK && XPRO == PRO_BNRELU) || ((EPI == EPI_ADD || EPI == EPI_ADDSTATS) && XPRO == PRO_NONE), "conv2 pair or conv1 pair");
    constexpr int KW = 3, NT = 64, NP = H ? 2 : 3, ROWS = NT + 2, PITCH = 72, PG = 72, PX = 88, XO = 8;
    constexpr int NPR = H ? 3 : 6, SPM = 6 / NPR;          // piece products per product; side-work slices per MFMA
    constexpr bool STATS = (EPI == EPI_RELUMASK || EPI == EPI_ADDSTATS);
    constexpr bool FOLD = (EPI == EPI_ADDSTATS);          // conv1 pair that also does the previous block's ReLU backward + BN sums
    constexpr int DIMG = NP * ROWS * PITCH, GIMG = NP * 64 * PG, XIMG = NP * 64 * PX;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    unsigned short* Db = reinterpret_cast<unsigned short*>(smem_raw);          // [NP][ROWS][PITCH]   row r = time t0 - 1 + r
    unsigned short* Gb0 = Db + DIMG;                                           // [2][NP][64][PG]
    unsigned short* Xb0 = Gb0 + 2 * GIMG;                                      // [2][NP][64][PX]     element XO + (t - t0)
    float* Cs = reinterpret_cast<float*>(Xb0 + 2 * XIMG);                      // [8][64]
    // H: a wave-private [32 channels][32 steps (+4)] exchange tile.  In the accumulator layout a lane owns one channel, so a dwordx4
    // access touches 32 rows with 16 bytes each; such a store (load) holds the wave at issue for ~300 cycles (measured: the four
    // epilogue stores were 1 200 of a tile's 7 000 cycles).  Epilogue operands and results therefore cross HBM in ROW layout (a lane
    // = 4 steps, 8 lanes = one 128-byte row segment, 8 rows per instruction) and change layout through this tile.
    constexpr bool XL = H;
    constexpr int EXP = 36;
    float* Ex = Cs + 8 * 64 + (threadIdx.x >> 6) * (32 * EXP);
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = wave & 1, nh = wave >> 1;           // data gradient: 32 output rows x 32 columns; weight gradient: block (mt, nt = nh)
    const int T = a.T;
    const int tilesPerClip = T / NT, ntiles = a.B * tilesPerClip;

    // ---- resident data-gradient weight fragments
    u32x4 Wr[12][NP];
    // H: {ws, 1 / ws} behind the [NP][KW][64][64] image
    const float winv = H ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(a.wp) + NP * KW * 4096)[1] : 1.f;
    const float gs = H ? a.gscale[0] : 1.f, ginv = H ? a.gscale[1] : 1.f;
    const float dinv = ginv * winv;                        // data gradient: (g gs) (w ws) -> g w
    {
        const uint4* wg = reinterpret_cast<const uint4*>(a.wp);
#pragma unroll
        for (int s = 0; s < 12; ++s)
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int e = ((p * KW + (s >> 2)) * 64 + 32 * mt + l31) * 64 + 16 * (s & 3) + 8 * half;
                u32x4 w_ = __builtin_bit_cast(u32x4, wg[e >> 3]);
                // pinned to the AGPR half of the register file: the MFMA reads its A operand from there directly.  Left to the
                // allocator the fragments are SPILLED to AGPRs and copied back (v_accvgpr_read) in front of every k-step.
                asm volatile("" : "+a"(w_));
                Wr[s][p] = w_;
            }
    }
    // ---- staging map: channel pair cp (channels c0, c0 + 1), unit u = time quad q0 + 8 u; halo: channel hc, side hh
    const int cp = wave * 8 + (lane & 7), c0 = 2 * cp, q0 = lane >> 3;
    const int hc = (tid & 127) >> 1, hh = tid & 1;
    float4 ra_[2], rb_[2], ya_[2], yb_[2], xa_[2], xb_[2];     // raw, per unit: dz / y of channel c0 (a) and c0 + 1 (b); x of both
    float4 gz_;                                                // the rebuilt gradient of the (unit, channel) being split
    unsigned gp_[2][2][6];                                     // its bf16 pieces [unit][channel][piece * 2 + time pair] (phase A -> phase B)
    float hg = 0.f, hy = 0.f, hxv = 0.f;
    wm_srd_t dsg = make_srd(a.g, 0), dsy = dsg, dsx = dsg, dsm = dsg;
    unsigned voff = 0, hoff = 0;
    bool okh = true;
    constexpr bool GMG = GM && EPI == EPI_RELUMASK;      // the mask belongs to the gradient operand g (else, with GM, to e1)
    unsigned mk_[2][2] = {{0u, 0u}, {0u, 0u}}, hmk = 0u; // mask dwords of (unit, channel) / of the halo step, as loaded
    unsigned voffm = 0, hoffm = 0, hsh = 0, hsh_cur = 0;
    const unsigned nwm = (unsigned)T >> 5;               // mask dwords per row (T % 64 == 0)
    auto set_tile = [&](int tile) {
        const int b = tile / tilesPerClip, t0 = (tile - b * tilesPerClip) * NT;
        const size_t clip = (size_t)b * 64 * T, bytes = (size_t)64 * T * sizeof(float);
        dsg = make_srd(a.g + clip, bytes);
        dsy = make_srd(a.g2 + clip, bytes);
        dsx = make_srd(a.x + clip, bytes);
        voff = (unsigned)(c0 * T + t0 + 4 * q0) * 4u;
        const int th = hh ? t0 + NT : t0 - 1;
        hoff = (unsigned)(hc * T + min(max(th, 0), T - 1)) * 4u;
        okh = th >= 0 && th < T;
        if (GMG) {
            dsm = make_srd(reinterpret_cast<const float*>(a.gmask) + (size_t)b * 64 * nwm, (size_t)64 * nwm * sizeof(unsigned));
            voffm = ((unsigned)c0 * nwm + ((unsigned)t0 >> 5)) * 4u;
            const int thc = min(max(th, 0), T - 1);
            hoffm = ((unsigned)hc * nwm + ((unsigned)thc >> 5)) * 4u;
            hsh = (unsigned)thc & 31u;
        }
    };
    const unsigned rowT = (unsigned)T * 4u;
    auto load_g = [&](int u) {
        ra_[u] = __builtin_bit_cast(float4, buf_load4(dsg, voff + 128u * u, 0u));
        rb_[u] = __builtin_bit_cast(float4, buf_load4(dsg, voff + 128u * u, rowT));
        ya_[u] = __builtin_bit_cast(float4, buf_load4(dsy, voff + 128u * u, 0u));
        yb_[u] = __builtin_bit_cast(float4, buf_load4(dsy, voff + 128u * u, rowT));
    };
    auto load_x = [&](int u) {
        xa_[u] = __builtin_bit_cast(float4, buf_load4(dsx, voff + 128u * u, 0u));
        xb_[u] = __builtin_bit_cast(float4, buf_load4(dsx, voff + 128u * u, rowT));
    };
    // The refills of the raw staging registers are dealt ONE load at a time over both phases (a register set is free from the slice
    // that consumed it until the same slice of the next tile): a burst of four 1-KB loads from one wave stalls that wave at issue
    // whenever the CU's miss queue is full, a different wave every tile -- measured as 430 of 8 250 cycles per tile spent at the
    // phase-A barrier waiting for whichever wave was hit.
    auto refillA = [&](int m) {                     // phase-A slice m
        if (m == 14) ra_[0] = __builtin_bit_cast(float4, buf_load4(dsg, voff, 0u));
        if (m == 20) rb_[0] = __builtin_bit_cast(float4, buf_load4(dsg, voff, rowT));
        if (m == 26) ya_[0] = __builtin_bit_cast(float4, buf_load4(dsy, voff, 0u));
        if (m == 32) yb_[0] = __builtin_bit_cast(float4, buf_load4(dsy, voff, rowT));
        if (m == 38) ra_[1] = __builtin_bit_cast(float4, buf_load4(dsg, voff + 128u, 0u));
        if (m == 44) rb_[1] = __builtin_bit_cast(float4, buf_load4(dsg, voff + 128u, rowT));
        if (m == 50) ya_[1] = __builtin_bit_cast(float4, buf_load4(dsy, voff + 128u, 0u));
        if (m == 56) yb_[1] = __builtin_bit_cast(float4, buf_load4(dsy, voff + 128u, rowT));
        if (GMG) {
            if (m == 16) mk_[0][0] = __builtin_bit_cast(unsigned, buf_load(dsm, voffm, 0u));
            if (m == 22) mk_[0][1] = __builtin_bit_cast(unsigned, buf_load(dsm, voffm, nwm * 4u));
            if (m == 40) mk_[1][0] = __builtin_bit_cast(unsigned, buf_load(dsm, voffm + 4u, 0u));
            if (m == 46) mk_[1][1] = __builtin_bit_cast(unsigned, buf_load(dsm, voffm + 4u, nwm * 4u));
        }
    };
    auto refillB = [&](int v) {                     // phase-B free slice v
        if (v == 1) xa_[0] = __builtin_bit_cast(float4, buf_load4(dsx, voff, 0u));
        if (v == 7) xb_[0] = __builtin_bit_cast(float4, buf_load4(dsx, voff, rowT));
        if (v == 13) xa_[1] = __builtin_bit_cast(float4, buf_load4(dsx, voff + 128u, 0u));
        if (v == 19) xb_[1] = __builtin_bit_cast(float4, buf_load4(dsx, voff + 128u, rowT));
    };
    auto load_ghalo = [&]() {
        hg = buf_load(dsg, hoff, 0u); hy = buf_load(dsy, hoff, 0u);
        if (GMG) hmk = __builtin_bit_cast(unsigned, buf_load(dsm, hoffm, 0u));
    };
    auto load_gmask = [&](int u) {
        mk_[u][0] = __builtin_bit_cast(unsigned, buf_load(dsm, voffm + 4u * u, 0u));
        mk_[u][1] = __builtin_bit_cast(unsigned, buf_load(dsm, voffm + 4u * u, nwm * 4u));
    };
    auto load_xhalo = [&]() { hxv = buf_load(dsx, hoff, 0u); };

    const int tstep = gridDim.x;
    int tile = WM_XCD_MAP ? xcd_slot() : (int)blockIdx.x;          // grid <= ntiles
    set_tile(tile);
    bool okh_cur = okh;
    load_g(0); load_g(1); load_x(0); load_x(1); load_ghalo(); load_xhalo();
    if (GMG) { load_gmask(0); load_gmask(1); }
    hsh_cur = hsh;
    if (tid < 64) {
        Cs[tid] = a.ga[tid];
        Cs[64 + tid] = a.gb[tid];
        Cs[128 + tid] = a.gc[tid];
        Cs[192 + tid] = a.gb[64 + tid];                  // low word of the BatchNorm-backward offset
        Cs[256 + tid] = (XPRO == PRO_BNRELU) ? a.xa[tid] : 0.f;
        Cs[320 + tid] = (XPRO == PRO_BNRELU) ? a.xb[tid] : 0.f;
        Cs[384 + tid] = (EPI == EPI_RELUMASK) ? a.ea[tid] : 0.f;
        Cs[448 + tid] = (EPI == EPI_RELUMASK) ? a.eb[tid] : 0.f;
    }
    // the element right of the right halo only feeds bits that the funnel shift drops: keep it defined in both images
    for (int i = tid; i < 2 * NP * 64; i += 256) Xb0[(i / (NP * 64)) * XIMG + (i % (NP * 64)) * PX + XO + NT + 1] = 0;
    __syncthreads();
    // per-thread constants (the staging channels never change)
    float kga[2], kgb[2], kgc[2], kgl[2], kxa[2], kxb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        kga[j] = Cs[c0 + j] * gs; kgb[j] = Cs[64 + c0 + j] * gs; kgc[j] = Cs[128 + c0 + j] * gs; kgl[j] = Cs[192 + c0 + j] * gs;   // gs = 1 | 2^k: exact
        kxa[j] = Cs[256 + c0 + j]; kxb[j] = Cs[320 + c0 + j];
    }
    const float hga = Cs[hc] * gs, hgb = Cs[64 + hc] * gs, hgc = Cs[128 + hc] * gs, hgl = Cs[192 + hc] * gs, hxa = Cs[256 + hc], hxb = Cs[320 + hc];

    // ---- split stages on a value pair (va, vb) -> pieces (p0, p1, [lo]) and a second pair (vc, vd) -> (q0_, q1_, [lo])
    float va = 0.f, vb = 0.f, vc = 0.f, vd = 0.f;
    unsigned p0 = 0, p1 = 0, r0 = 0, r1 = 0;
    float bsum[2] = {0.f, 0.f};
    float bflag = 1.f;
    auto s1a = [&]() {
        if (H) {
            const h16x2 h_ = __builtin_convertvector(f32x2{va, vb}, h16x2);
            p0 = __builtin_bit_cast(unsigned, h_);
            va -= (float)h_.x; vb -= (float)h_.y;
        } else {
            const bf16x2 h_ = {(__bf16)va, (__bf16)vb};
            p0 = __builtin_bit_cast(unsigned, h_);
            va -= __uint_as_float(p0 << 16); vb -= __uint_as_float(p0 & 0xffff0000u);
        }
        asm volatile("" : "+v"(va), "+v"(vb), "+v"(p0));
    };
    auto s2a = [&]() {
        if (H) return;                                   // two pieces: no middle one
        const bf16x2 m_ = {(__bf16)va, (__bf16)vb};
        p1 = __builtin_bit_cast(unsigned, m_);
        va -= __uint_as_float(p1 << 16); vb -= __uint_as_float(p1 & 0xffff0000u);
        asm volatile("" : "+v"(va), "+v"(vb), "+v"(p1));
    };
    auto s1b = [&]() {
        if (H) {
            const h16x2 h_ = __builtin_convertvector(f32x2{vc, vd}, h16x2);
            r0 = __builtin_bit_cast(unsigned, h_);
            vc -= (float)h_.x; vd -= (float)h_.y;
        } else {
            const bf16x2 h_ = {(__bf16)vc, (__bf16)vd};
            r0 = __builtin_bit_cast(unsigned, h_);
            vc -= __uint_as_float(r0 << 16); vd -= __uint_as_float(r0 & 0xffff0000u);
        }
        asm volatile("" : "+v"(vc), "+v"(vd), "+v"(r0));
    };
    auto s2b = [&]() {
        if (H) return;
        const bf16x2 m_ = {(__bf16)vc, (__bf16)vd};
        r1 = __builtin_bit_cast(unsigned, m_);
        vc -= __uint_as_float(r1 << 16); vd -= __uint_as_float(r1 & 0xffff0000u);
        asm volatile("" : "+v"(vc), "+v"(vd), "+v"(r1));
    };
    unsigned l0 = 0, l1 = 0;
    auto lo_pair = [&](float x0, float x1) -> unsigned {         // the last piece of a value pair
        if (H) return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, h16x2));
        const bf16x2 l_ = {(__bf16)x0, (__bf16)x1};
        return __builtin_bit_cast(unsigned, l_);
    };
    auto out4 = [&](unsigned short* dst, int stride_p) {         // two time pairs of one channel row: 8-byte writes
        l0 = lo_pair(va, vb); l1 = lo_pair(vc, vd);
        *reinterpret_cast<uint2*>(dst) = make_uint2(p0, r0);
        if (!H) *reinterpret_cast<uint2*>(dst + stride_p) = make_uint2(p1, r1);
        *reinterpret_cast<uint2*>(dst + (NP - 1) * stride_p) = make_uint2(l0, l1);
    };
    // gradient rebuild of channel j of unit u from the raw staging registers (free for the refill afterwards): one value per slice
    // (a slice must stay within the shadow of one MFMA, about seven instructions: a twelve-instruction slice costs its excess in full)
    auto g_build = [&](int u, int j, int e) {
        const float4 dz = j ? rb_[u] : ra_[u], yy = j ? yb_[u] : ya_[u];
        float4& gz = gz_;
        float d = (e == 0) ? dz.x : (e == 1) ? dz.y : (e == 2) ? dz.z : dz.w;
        const float yv = (e == 0) ? yy.x : (e == 1) ? yy.y : (e == 2) ? yy.z : yy.w;
        if (GMG) {                          // dz = g_out where the block output was positive: its bit, sign-extended, ANDed in
            if (e == 0) mk_[u][j] >>= 4 * q0;
            d = __uint_as_float(__float_as_uint(d) & (unsigned)__builtin_amdgcn_sbfe((int)mk_[u][j], e, 1));
        }
        float v = pro_apply<PRO_BNBWD>(d, yv, kga[j], kgb[j], kgc[j], kgl[j]);
        if (H) v = __builtin_amdgcn_fmed3f(v, -6.0e4f, 6.0e4f);      // a spike saturates; it never becomes an f16 infinity
        bsum[j] = fmaf(bflag, v, bsum[j]);
        asm volatile("" : "+v"(v), "+v"(bsum[j]));
        if (e == 0) gz.x = v; else if (e == 1) gz.y = v; else if (e == 2) gz.z = v; else gz.w = v;
    };
    auto g_pick_t = [&](int u, int j) {                         // time pairs of channel j of unit u
        const float4 gz = gz_;
        va = gz.x; vb = gz.y; vc = gz.z; vd = gz.w;
        asm volatile("" : "+v"(va), "+v"(vb), "+v"(vc), "+v"(vd));
    };
    auto x_pick_t = [&](int u, int j) {
        float4 xx = j ? xb_[u] : xa_[u];
        if (XPRO == PRO_BNRELU) {
            xx.x = pro_apply<PRO_BNRELU>(xx.x, 0.f, kxa[j], kxb[j], 0.f); xx.y = pro_apply<PRO_BNRELU>(xx.y, 0.f, kxa[j], kxb[j], 0.f);
            xx.z = pro_apply<PRO_BNRELU>(xx.z, 0.f, kxa[j], kxb[j], 0.f); xx.w = pro_apply<PRO_BNRELU>(xx.w, 0.f, kxa[j], kxb[j], 0.f);
        }
        va = xx.x; vb = xx.y; vc = xx.z; vd = xx.w;
        asm volatile("" : "+v"(va), "+v"(vb), "+v"(vc), "+v"(vd));
    };
    // image D wants channel pairs (c0, c0 + 1) at ONE time step; the pieces of a value do not depend on what it is paired with, so
    // they are taken from the time-pair dwords of the two channels with one v_perm each instead of splitting the gradient again
    auto d_out = [&](int u, int e) {
        const unsigned sel = (e & 1) ? 0x07060302u : 0x05040100u;   // high | low halves of (channel c0 + 1 : channel c0)
        const int pr = e >> 1;
        const unsigned d0 = __builtin_amdgcn_perm(gp_[u][1][pr], gp_[u][0][pr], sel);
        const unsigned d1 = __builtin_amdgcn_perm(gp_[u][1][2 + pr], gp_[u][0][2 + pr], sel);
        const unsigned d2 = __builtin_amdgcn_perm(gp_[u][1][4 + pr], gp_[u][0][4 + pr], sel);
        unsigned* D32 = reinterpret_cast<unsigned*>(Db);
        const int o = (1 + 4 * (q0 + 8 * u) + e) * (PITCH / 2) + cp;
        D32[o] = d0;
        if (!H) D32[(ROWS * PITCH >> 1) + o] = d1;
        D32[(NP - 1) * (ROWS * PITCH >> 1) + o] = d2;
    };
    // halos: the gradient's for image D (rows 0 and 65), the input operand's for image X' (elements XO - 1, XO + 64)
    auto hx_pick = [&]() {
        float v = hxv;
        if (XPRO == PRO_BNRELU) v = pro_apply<PRO_BNRELU>(v, 0.f, hxa, hxb, 0.f);
        va = okh_cur ? v : 0.f; vb = 0.f;
        asm volatile("" : "+v"(va), "+v"(vb));
    };
    auto hx_out = [&](unsigned short* X) {
        const unsigned la = lo_pair(va, vb);
        if (tid < 128) {
            const int o = hc * PX + (hh ? XO + NT : XO - 1);
            X[o] = (unsigned short)p0;
            if (!H) X[64 * PX + o] = (unsigned short)p1;
            X[(NP - 1) * 64 * PX + o] = (unsigned short)la;
        }
    };
    auto hg_pick = [&]() {
        float hgm = hg;
        if (GMG) hgm = __uint_as_float(__float_as_uint(hg) & (unsigned)__builtin_amdgcn_sbfe((int)(hmk >> hsh_cur), 0, 1));
        float v = pro_apply<PRO_BNBWD>(hgm, hy, hga, hgb, hgc, hgl);
        if (H) v = __builtin_amdgcn_fmed3f(v, -6.0e4f, 6.0e4f);
        va = okh_cur ? v : 0.f; vb = 0.f;
        asm volatile("" : "+v"(va), "+v"(vb));
    };
    auto hg_out = [&]() {
        const unsigned la = lo_pair(va, vb);
        if (tid < 128) {
            const int o = (hh ? NT + 1 : 0) * PITCH + hc;
            Db[o] = (unsigned short)p0;
            if (!H) Db[ROWS * PITCH + o] = (unsigned short)p1;
            Db[(NP - 1) * ROWS * PITCH + o] = (unsigned short)la;
        }
    };
    // phase-A side work, slice v of 64 (the other 8 slices fetch the epilogue operand): the operands in the registers -> images G', X'
    //   v 0..35  gradient: per (unit, channel) 9 slices: rebuild x 4 | pick | hi a | mid a + hi b | mid b | lo + write
    //   v 36..59 input:    per (unit, channel) 6 slices: pick (+ BN + ReLU) | hi a | mid a | hi b | mid b | lo + write
    //   v 60..63 input halo
    constexpr int NSA = 64;
    auto sideA = [&](int v, unsigned short* G, unsigned short* X) __attribute__((always_inline)) {
        if (v < 36) {
            const int uj = v / 9, st = v % 9, u = uj >> 1, j = uj & 1;
            if (st < 4) g_build(u, j, st);
            if (st == 4) g_pick_t(u, j);
            if (st == 5) s1a();
            if (st == 6) { s2a(); s1b(); }
            if (st == 7) s2b();
            if (st == 8) {
                out4(G + (c0 + j) * PG + 4 * (q0 + 8 * u), 64 * PG);
                gp_[u][j][0] = p0; gp_[u][j][1] = r0; gp_[u][j][2] = p1; gp_[u][j][3] = r1; gp_[u][j][4] = l0; gp_[u][j][5] = l1;
            }
        } else if (v < 60) {
            const int w = v - 36, uj = w / 6, st = w % 6, u = uj >> 1, j = uj & 1;
            if (st == 0) x_pick_t(u, j);
            if (st == 1) s1a();
            if (st == 2) s2a();
            if (st == 3) s1b();
            if (st == 4) s2b();
            if (st == 5) out4(X + (c0 + j) * PX + XO + 4 * (q0 + 8 * u), 64 * PX);
        } else {
            const int st = v - 60;
            if (st == 0) { hx_pick(); load_xhalo(); }
            if (st == 1) s1a();
            if (st == 2) s2a();
            if (st == 3) hx_out(X);
        }
    };
    // phase-B side work, slice v of 12: the gradient's pieces -> image D (channel pairs by v_perm)
    //   v 0..7  one slice per (unit, time step): 3 perms + 3 writes;   v 8..11 gradient halo
    constexpr int NSB = 12;
    auto sideB = [&](int v) __attribute__((always_inline)) {
        if (v < 8) {
            d_out(v >> 2, v & 3);
        } else {
            const int st = v - 8;
            if (st == 0) { hg_pick(); load_ghalo(); }
            if (st == 1) s1a();
            if (st == 2) s2a();
            if (st == 3) hg_out();
        }
    };
    {   // first tile: all images serially; the refills inside the slices already fetch the second tile
        set_tile(min(tile + tstep, ntiles - 1));
        const bool okh_n = okh;
#pragma unroll
        for (int v = 0; v < NSA; ++v) { sideA(v, Gb0, Xb0); refillA(v); }
        const unsigned hsh_n = hsh;
#pragma unroll
        for (int v = 0; v < 2 * NSB; ++v) { if ((v & 1) == 0) sideB(v >> 1); else refillB(v); }
        okh_cur = okh_n; hsh_cur = hsh_n;
    }
    __syncthreads();

    f32x16 wacc[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) wacc[k][r] = 0.f;
    // The data gradient is accumulated TRANSPOSED (gradient image = A operand, weights = B operand): a lane then owns ONE output
    // channel (32 mt + l31) and, per accumulator quad, four consecutive time steps -- the epilogue operand comes in and the
    // result goes out as dwordx4 (8 memory instructions per tile instead of 32), the ReLU-mask constants and the two BatchNorm
    // sums are one register each instead of sixteen.
    float s1 = 0.f, s2 = 0.f;
    f32x4 e1q[4];
    // conv2 pair: the epilogue operand IS the input operand of the weight gradient (the pre-BatchNorm activation), read a second
    // time in the accumulator layout.  Fetched one tile AHEAD (e1n, for the next tile, while this one is in phase A) it follows
    // the staging loads of the same tile by less than one tile's worth of traffic and is served by the XCD's L2 instead of HBM
    // (fetched as late as the tile needs it, 32 % of the kernel's HBM traffic was this re-read).
    // (round 3, f16 build: with half the MFMAs per phase an operand fetched in phase A no longer has a phase's worth of time to arrive
    // before the epilogue in phase B needs it -- EVERY epilogue operand of every form is now fetched one tile ahead: e1, the mask words,
    // the previous block's y2)
    constexpr bool EAHEAD = true;
    f32x4 e1n[4];
    unsigned emk = 0u, pmk = 0u, emkn = 0u, pmkn = 0u;
    f32x4 pyq[FOLD ? 4 : 1], pyn[FOLD ? 4 : 1];
    f32x4 ec[XL ? 4 : 1], pyc[(XL && FOLD) ? 4 : 1];     // row-layout staging of the next tile's epilogue operands
    const int xrow = lane >> 3, xcol = 4 * (lane & 7);       // row layout: lane -> (row xrow + 8 r, steps xcol .. xcol + 3)
    const unsigned xlane = (unsigned)(xrow * T + xcol) * 4u;
    unsigned row8T[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) row8T[r] = (unsigned)(8 * r * T) * 4u;
    const float kea = (EPI == EPI_RELUMASK) ? Cs[384 + 32 * mt + l31] : 0.f, keb = (EPI == EPI_RELUMASK) ? Cs[448 + 32 * mt + l31] : 0.f;
    int buf = 0;
#ifdef WM_STAMP
    unsigned long long tm[6] = {0, 0, 0, 0, 0, 0};
#endif
    {                                                    // the first tile's epilogue operands
        const int b0 = tile / tilesPerClip, t00 = (tile - b0 * tilesPerClip) * NT;
        const size_t slab0 = ((size_t)b0 * 64 + 32 * mt) * T;
        const unsigned eo0 = (unsigned)(l31 * T + t00 + 32 * nh + 4 * half) * 4u, mo0 = ((unsigned)l31 * nwm + ((unsigned)t00 >> 5) + (unsigned)nh) * 4u;
        const wm_srd_t s0 = make_srd(a.e1 + slab0, (size_t)32 * T * sizeof(float));
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) e1q[q4] = buf_load4(s0, eo0 + 32u * q4, 0u);
        if (GM && (EPI == EPI_ADD || EPI == EPI_ADDSTATS))
            emk = __builtin_bit_cast(unsigned, buf_load(make_srd(reinterpret_cast<const float*>(a.gmask) + ((size_t)b0 * 64 + 32 * mt) * nwm,
                                                                 (size_t)32 * nwm * sizeof(unsigned)), mo0, 0u));
        if (FOLD) {
            pmk = __builtin_bit_cast(unsigned, buf_load(make_srd(reinterpret_cast<const float*>(a.pmask) + ((size_t)b0 * 64 + 32 * mt) * nwm,
                                                                 (size_t)32 * nwm * sizeof(unsigned)), mo0, 0u));
            const wm_srd_t sp0 = make_srd(a.py2 + slab0, (size_t)32 * T * sizeof(float));
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) pyq[q4] = buf_load4(sp0, eo0 + 32u * q4, 0u);
        }
    }
#define FENCE __builtin_amdgcn_sched_barrier(0)
    // one piece product on raw 128-bit fragments; the j-th product of a k-step pairs piece PA(j) of the first operand with piece PB(j)
    // of the second, small terms first (bf16x6: mid mid, hi lo, lo hi, hi mid, mid hi, hi hi; f16: lo hi, hi lo, hi hi)
    auto mma = [&](const u32x4& A_, const u32x4& B_, f32x16 c) -> f32x16 {
        if (H) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, A_), __builtin_bit_cast(h16x8, B_), c, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A_), __builtin_bit_cast(bf16x8, B_), c, 0, 0, 0);
    };
    auto PA = [](int j) { return H ? (j == 0 ? 1 : 0) : ((j == 0 || j == 4) ? 1 : (j == 2 ? 2 : 0)); };
    auto PB = [](int j) { return H ? (j == 1 ? 1 : 0) : ((j == 0 || j == 3) ? 1 : (j == 1 ? 2 : 0)); };
    float vmax = 0.f;                                    // H: running max |y| of the data-gradient values this lane stores
    while (tile < ntiles) {
        // registers: the raw operands of tile + tstep (clamped: the duplicate of the last tile is split but never used)
        const int b = tile / tilesPerClip, t0 = (tile - b * tilesPerClip) * NT;
        bflag = (tile + tstep < ntiles) ? 1.f : 0.f;
        set_tile(min(tile + 2 * tstep, ntiles - 1));       // what the refills inside the slices fetch
        const bool okh_n = okh;
        const unsigned hsh_n = hsh;
        const unsigned short* Gc = Gb0 + buf * GIMG;
        const unsigned short* Xc = Xb0 + buf * XIMG;
        unsigned short* Gn = Gb0 + (buf ^ 1) * GIMG;
        unsigned short* Xn = Xb0 + (buf ^ 1) * XIMG;
        // data-gradient output / epilogue operand of this lane: channel 32 mt + l31, steps t0 + 32 nh + 8 q + 4 half + (0..3) for quad q
        const size_t slab = ((size_t)b * 64 + 32 * mt) * T;
        const wm_srd_t sye = make_srd(a.y + slab, (size_t)32 * T * sizeof(float));
        const wm_srd_t se1 = make_srd(a.e1 + slab, (size_t)32 * T * sizeof(float));
        const unsigned eoff = (unsigned)(l31 * T + t0 + 32 * nh + 4 * half) * 4u;
        const int tnx = min(tile + tstep, ntiles - 1), bnx = tnx / tilesPerClip, t0nx = (tnx - bnx * tilesPerClip) * NT;
        const wm_srd_t se1n = EAHEAD ? make_srd(a.e1 + ((size_t)bnx * 64 + 32 * mt) * T, (size_t)32 * T * sizeof(float)) : se1;
        const unsigned eoffn = (unsigned)(l31 * T + t0nx + 32 * nh + 4 * half) * 4u;
        // GM, conv1 pair: e1 is the gradient that reaches the block output; its ReLU bits: dword t0 / 32 + nh of the lane's row
        constexpr bool GME = GM && (EPI == EPI_ADD || EPI == EPI_ADDSTATS);
        const wm_srd_t spm = FOLD ? make_srd(reinterpret_cast<const float*>(a.pmask) + ((size_t)b * 64 + 32 * mt) * nwm,
                                             (size_t)32 * nwm * sizeof(unsigned)) : se1;
        const wm_srd_t spy = FOLD ? make_srd(a.py2 + slab, (size_t)32 * T * sizeof(float)) : se1;
        const wm_srd_t sme = GME ? make_srd(reinterpret_cast<const float*>(a.gmask) + ((size_t)b * 64 + 32 * mt) * nwm,
                                            (size_t)32 * nwm * sizeof(unsigned)) : se1;
        const size_t slabn = ((size_t)bnx * 64 + 32 * mt) * T;
        const wm_srd_t spmn = FOLD ? make_srd(reinterpret_cast<const float*>(a.pmask) + ((size_t)bnx * 64 + 32 * mt) * nwm,
                                              (size_t)32 * nwm * sizeof(unsigned)) : se1;
        const wm_srd_t spyn = FOLD ? make_srd(a.py2 + slabn, (size_t)32 * T * sizeof(float)) : se1;
        const wm_srd_t smen = GME ? make_srd(reinterpret_cast<const float*>(a.gmask) + ((size_t)bnx * 64 + 32 * mt) * nwm,
                                             (size_t)32 * nwm * sizeof(unsigned)) : se1;
        const unsigned emoffn = ((unsigned)l31 * nwm + ((unsigned)t0nx >> 5) + (unsigned)nh) * 4u;
        const unsigned xoffn = xlane + (unsigned)(t0nx + 32 * nh) * 4u, xoff = xlane + (unsigned)(t0 + 32 * nh) * 4u;

        STAMP(ts0);
        // ---------------- phase A: data gradient out of image D
        f32x16 dacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) dacc[r] = 0.f;
        {
            const unsigned short* drow = Db + (32 * nh + l31) * PITCH + 8 * half;
            u32x4 Bq[2][NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) Bq[0][p] = *reinterpret_cast<const u32x4*>(drow + p * ROWS * PITCH);
#pragma unroll
            for (int s = 0; s < 12; ++s) {
                if (s + 1 < 12) {
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        Bq[(s + 1) & 1][p] = *reinterpret_cast<const u32x4*>(drow + (p * ROWS + ((s + 1) >> 2)) * PITCH + 16 * ((s + 1) & 3));
                }
                const u32x4* Bf = Bq[s & 1];
                // the weight-gradient accumulators stay where they are across this phase: left alone, the allocator lends a[0:15] of
                // one of them to dacc and copies it out and back around every phase A (32 v_accvgpr moves per tile)
                if ((s & 3) == 1) asm volatile("" : "+a"(wacc[0]), "+a"(wacc[1]), "+a"(wacc[2]));
#pragma unroll
                for (int j = 0; j < NPR; ++j) {
                    FENCE;
                    dacc = mma(Bf[PB(j)], Wr[s][PA(j)], dacc);      // D^T: rows = time, columns = channel (gradient piece PB, weight piece PA)
                    FENCE;
                    // SPM slices of side work behind this MFMA (72 slices per phase whatever the arithmetic)
#pragma unroll
                    for (int i_ = 0; i_ < SPM; ++i_) {
                        const int m = (s * NPR + j) * SPM + i_;          // 0..71
                        if (XL && m < 8 && (m & 1)) ec[m >> 1] = buf_load4(se1n, xoffn, row8T[m >> 1]);
                        if (XL && FOLD && m >= 8 && m < 16 && (m & 1)) pyc[(m - 8) >> 1] = buf_load4(spyn, xoffn, row8T[(m - 8) >> 1]);
                        if (m < NSA) {
                            sideA(m, Gn, Xn); refillA(m);
                            if (GME && m == 61) emkn = __builtin_bit_cast(unsigned, buf_load(smen, emoffn, 0u));
                            if (FOLD && m == 62) pmkn = __builtin_bit_cast(unsigned, buf_load(spmn, emoffn, 0u));
                        }
                        else if (XL) { }
                        else if (((m - NSA) & 1) == 0) e1n[(m - NSA) >> 1] = buf_load4(se1n, eoffn + 32u * ((m - NSA) >> 1), 0u);
                        else if (FOLD) pyn[(m - NSA) >> 1] = buf_load4(spyn, eoffn + 32u * ((m - NSA) >> 1), 0u);
                    }
                    FENCE;
                }
            }
        }
        STAMP(ts1);
#ifdef WM_STAMP
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        STAMP(ts1b);
        asm volatile("s_barrier" ::: "memory");
#else
        lds_barrier();          // every wave is done with image D; images G', X' of the next tile are complete
#endif
        STAMP(ts2);
        // ---------------- phase B: weight gradient out of images G', X'; image D of the next tile, epilogue of this one
        {
            const int e0 = 8 * half;
            u32x4 A[2][NP];
            uint4 f[2][NP];
            unsigned Lw[2][NP], Rw[2][NP];
            u32x4 Bl[NP], Br[NP];
            auto read_kb = [&](int kb, int set) {
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    A[set][p] = *reinterpret_cast<const u32x4*>(Gc + (p * 64 + mt * 32 + l31) * PG + kb * 16 + e0);
                    const unsigned short* xr = Xc + (p * 64 + nh * 32 + l31) * PX + XO + kb * 16 + e0;
                    f[set][p] = *reinterpret_cast<const uint4*>(xr);
                    Lw[set][p] = *reinterpret_cast<const unsigned*>(xr - 2);
                    Rw[set][p] = *reinterpret_cast<const unsigned*>(xr + 8);
                }
            };
            auto shl = [&](int set, int p) {
                const uint4 g = f[set][p];
                uint4 s_ = make_uint4(__builtin_amdgcn_alignbit(g.x, Lw[set][p], 16), __builtin_amdgcn_alignbit(g.y, g.x, 16),
                                      __builtin_amdgcn_alignbit(g.z, g.y, 16), __builtin_amdgcn_alignbit(g.w, g.z, 16));
                asm volatile("" : "+v"(s_.x), "+v"(s_.y), "+v"(s_.z), "+v"(s_.w));
                Bl[p] = __builtin_bit_cast(u32x4, s_);
            };
            auto shr = [&](int set, int p) {
                const uint4 g = f[set][p];
                uint4 s_ = make_uint4(__builtin_amdgcn_alignbit(g.y, g.x, 16), __builtin_amdgcn_alignbit(g.z, g.y, 16),
                                      __builtin_amdgcn_alignbit(g.w, g.z, 16), __builtin_amdgcn_alignbit(Rw[set][p], g.w, 16));
                asm volatile("" : "+v"(s_.x), "+v"(s_.y), "+v"(s_.z), "+v"(s_.w));
                Br[p] = __builtin_bit_cast(u32x4, s_);
            };
            // epilogue of accumulator registers 2 i, 2 i + 1 of the data gradient (two time steps of the lane's channel); every second
            // call completes a quad and stores it
            auto epi2 = [&](int i) {
                const int r0 = 2 * i, r1 = 2 * i + 1;
                float v0 = dacc[r0], v1 = dacc[r1];
                if (H) { v0 *= dinv; v1 *= dinv; }                // undo the gradient's and the weights' scales (a power of two: exact)
                const float q0_ = e1q[r0 >> 2][r0 & 3], q1_ = e1q[r1 >> 2][r1 & 3];
                if (EPI == EPI_RELUMASK) {
                    // both compares first, both selects after: a select right behind its compare costs two wait states
                    const bool k0 = fmaf(q0_, kea, keb) > 0.f, k1 = fmaf(q1_, kea, keb) > 0.f;
                    v0 = k0 ? v0 : 0.f; v1 = k1 ? v1 : 0.f;
                    s1 += v0; s2 = fmaf(v0, q0_, s2);
                    s1 += v1; s2 = fmaf(v1, q1_, s2);
                    asm volatile("" : "+v"(s1), "+v"(s2));
                } else if (GM) {
                    // bits 8 q + 4 half + e of the row's dword: shifted by 4 half once (first call), then constant field positions
                    if (i == 0) emk >>= 4 * half;
                    v0 += __uint_as_float(__float_as_uint(q0_) & (unsigned)__builtin_amdgcn_sbfe((int)emk, 8 * (r0 >> 2) + (r0 & 3), 1));
                    v1 += __uint_as_float(__float_as_uint(q1_) & (unsigned)__builtin_amdgcn_sbfe((int)emk, 8 * (r1 >> 2) + (r1 & 3), 1));
                } else {
                    v0 += q0_; v1 += q1_;
                }
                if (FOLD) {
                    // what leaves is dz2 of the PREVIOUS block: its output's ReLU mask on the gradient just formed, and the two sums its
                    // BatchNorm backward needs (sum dz2, sum dz2 y2) -- that block then has no reduction pass of its own
                    if (i == 0) pmk >>= 4 * half;
                    v0 = __uint_as_float(__float_as_uint(v0) & (unsigned)__builtin_amdgcn_sbfe((int)pmk, 8 * (r0 >> 2) + (r0 & 3), 1));
                    v1 = __uint_as_float(__float_as_uint(v1) & (unsigned)__builtin_amdgcn_sbfe((int)pmk, 8 * (r1 >> 2) + (r1 & 3), 1));
                    s1 += v0; s2 = fmaf(v0, pyq[r0 >> 2][r0 & 3], s2);
                    s1 += v1; s2 = fmaf(v1, pyq[r1 >> 2][r1 & 3], s2);
                    asm volatile("" : "+v"(s1), "+v"(s2));
                }
                dacc[r0] = v0; dacc[r1] = v1;
                if (H && (STATS || EPI == EPI_ADD)) vmax = fmaxf(vmax, fmaxf(fabsf(v0), fabsf(v1)));   // one v_max3 per pair
                if (i & 1) {
                    const int q4 = i >> 1;
                    const f32x4 quad = {dacc[4 * q4], dacc[4 * q4 + 1], dacc[4 * q4 + 2], dacc[4 * q4 + 3]};
                    if (XL) *reinterpret_cast<f32x4*>(Ex + l31 * EXP + 8 * q4 + 4 * half) = quad;
                    else buf_store4(sye, quad, eoff + 32u * q4, 0u);
                }
            };
            read_kb(0, 0);
            // the slice behind old-style MFMA index m (18 per k-block): shifts of the X fragments first, then 12 free slices per k-block
            // = 48: image D of the next tile in every second one of the first 24, the epilogue quads in the last 24
            auto sliceB = [&](int m) __attribute__((always_inline)) {
                const int kb = m / 18, mm = m % 18, set = kb & 1;
                if (mm < 3) { if (mm < NP) shl(set, mm); }
                else if (mm < 6) { if (mm - 3 < NP) shr(set, mm - 3); }
                else {
                    const int v = kb * 12 + (mm - 6);             // 0..47
                    if (v < 2 * NSB) {
                        if ((v & 1) == 0) sideB(v >> 1); else refillB(v);
                        if (XL) {       // the next tile's epilogue operands: row layout -> exchange tile -> accumulator layout
                            if (v == 3) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x4*>(Ex + (xrow + 8 * r) * EXP + xcol) = ec[r];
                            }
                            if (v == 5) {
#pragma unroll
                                for (int q4 = 0; q4 < 4; ++q4) e1n[q4] = *reinterpret_cast<const f32x4*>(Ex + l31 * EXP + 8 * q4 + 4 * half);
                            }
                            if (FOLD && v == 9) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x4*>(Ex + (xrow + 8 * r) * EXP + xcol) = pyc[r];
                            }
                            if (FOLD && v == 11) {
#pragma unroll
                                for (int q4 = 0; q4 < 4; ++q4) pyn[q4] = *reinterpret_cast<const f32x4*>(Ex + l31 * EXP + 8 * q4 + 4 * half);
                            }
                        }
                    }
                    else if (XL) {      // epilogue quads every second slice, then the result tile leaves in row layout
                        if (v < 40) { if ((v & 1) == 0) epi2((v - 24) >> 1); }
                        else if ((v & 1) == 0) {
                            const int r = (v - 40) >> 1;
                            buf_store4(sye, *reinterpret_cast<const f32x4*>(Ex + (xrow + 8 * r) * EXP + xcol), xoff, row8T[r]);
                        }
                    }
                    else if (v >= 24 && ((v - 24) % 3) == 0) epi2((v - 24) / 3);
                }
                // the next k-block's fragments (the other register set is free since this k-block began); with two slices per MFMA the
                // old place (slice 12 of 18) left them only three MFMAs to arrive
                if (mm == (H ? 4 : 12) && kb + 1 < 4) read_kb(kb + 1, set ^ 1);
            };
#pragma unroll
            for (int m = 0; m < 12 * NPR; ++m) {
                const int kb = m / (3 * NPR), tg = (m % (3 * NPR)) / NPR, j = m % NPR, set = kb & 1;
                FENCE;
                if (tg == 0)
                    wacc[1] = mma(A[set][PA(j)], __builtin_bit_cast(u32x4, f[set][PB(j)]), wacc[1]);
                else if (tg == 1)
                    wacc[0] = mma(A[set][PA(j)], Bl[PB(j)], wacc[0]);
                else
                    wacc[2] = mma(A[set][PA(j)], Br[PB(j)], wacc[2]);
                FENCE;
#pragma unroll
                for (int i_ = 0; i_ < SPM; ++i_) sliceB(m * SPM + i_);
                FENCE;
#ifdef WM_STAMP
                if (m == 6 * NPR - 1) { STAMP(tsh); tm[5] += tsh - ts2; }     // end of the second k-block: first half of phase B
#endif
            }
            okh_cur = okh_n; hsh_cur = hsh_n;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) { e1q[q4] = e1n[q4]; if (FOLD) pyq[q4] = pyn[q4]; }
            emk = emkn; pmk = pmkn;
        }
        STAMP(ts3);
        lds_barrier();          // image D of the next tile is complete; images G', X' of this tile are free
        STAMP(ts4);
#ifdef WM_STAMP
        tm[0] += ts1 - ts0; tm[1] += ts2 - ts1; tm[2] += ts3 - ts2; tm[3] += ts4 - ts3; tm[4] += ts1b - ts1;
#endif
        tile += tstep;
        buf ^= 1;
    }
#undef FENCE

#ifdef WM_STAMP
    if (g_wm_stamp && lane == 0) {
        unsigned long long* d = g_wm_stamp + ((size_t)blockIdx.x * 4 + wave) * 6;
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i] = tm[i];
    }
#endif
    // ---- outputs: weight-gradient slab (every wave owns its block), bias sums, BatchNorm sums of the data gradient
    float* out = a.partial + (size_t)blockIdx.x * (KW * 4096 + 64);
#pragma unroll
    for (int k = 0; k < KW; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[(k * 64 + mt * 32 + mfma_row(r, half)) * 64 + nh * 32 + l31] = wacc[k][r] * ginv;   // x is unscaled
#pragma unroll
    for (int j = 0; j < 2; ++j) {       // the eight lanes of a channel pair sit 8 apart
        float v = bsum[j] * ginv;
        v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
        if (lane < 8) out[KW * 4096 + c0 + j] = v;
    }
    if (STATS) {
        float* red = reinterpret_cast<float*>(smem_raw);          // [2 column halves][2][64]
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);       // the two time halves of the lane's channel
        __syncthreads();
        if (half == 0) {
            red[nh * 128 + 32 * mt + l31] = s1;
            red[nh * 128 + 64 + 32 * mt + l31] = s2;
        }
        __syncthreads();
        if (tid < 128) a.stats[(size_t)blockIdx.x * 128 + tid] = red[tid] + red[128 + tid];
    }
    if (H && (STATS || EPI == EPI_ADD) && a.dzmax) {          // max |y| of this workgroup's output: the next consumer's gradient scale
        float* red = reinterpret_cast<float*>(smem_raw);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
        __syncthreads();
        if (lane == 0) red[wave] = vmax;
        __syncthreads();
        if (tid == 0) a.dzmax[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    }
}

template <int EPI, int XPRO, bool GM, bool H>
int launch_dwgrad64bf(const DWArgs& a, int* grid_out, hipStream_t stream) {
    constexpr int NP = H ? 2 : 3;
    constexpr size_t lds = (size_t)(NP * 66 * 72 + 2 * NP * 64 * 72 + 2 * NP * 64 * 88) * 2 + 8 * 64 * sizeof(float) +
                           (H ? (size_t)4 * 32 * 36 * sizeof(float) : 0);
    static wm::DevOnce attr_done;
    auto kern = dwgrad64bf_kernel<EPI, XPRO, GM, H>;
    if (!wm::dev_done(attr_done)) {
        WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        wm::dev_mark(attr_done);
    }
    const int ntiles = a.B * (a.T / 64);
    const int grid = ntiles < kNumCU ? ntiles : kNumCU;
    *grid_out = grid;
    if (a.stats && grid < kNumCU) WM_TRY(hipMemsetAsync(a.stats, 0, sizeof(float) * 128 * kNumCU, stream));
    if (H && a.dzmax && grid < kNumCU) WM_TRY(hipMemsetAsync(a.dzmax, 0, sizeof(float) * kNumCU, stream));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
    WM_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

#ifdef WM_STAMP
int wm_debug_set_stamp_buffer(unsigned long long* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wm_stamp), &buf, sizeof(buf));
}
#endif

int wm_pack_w64(const float* w, float* wp, int KW, int mode, hipStream_t stream) {
    if ((KW != 3 && KW != 7) || mode < 0 || mode > 3) return (int)hipErrorInvalidValue;
    const int n = KW * 4096;
    hipLaunchKernelGGL(pack_w64_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, w, wp, KW, mode);
    WM_CHECK_LAUNCH();
    return 0;
}

// Generic 64->64 'same' convolution on [B,64,T] (T % 4 == 0).
//   pro: 0 none | 1 relu(x*pa[c]+pb[c]) | 2 x+pa[b*64+c] | 3 pa[c]*x + pb[c] + pc[c]*x2
//   epi: 0 +bias | 1 mask by (e1*ea[c]+eb[c] > 0), stats = (sum v, sum v*e1) | 2 +e1 | 3 none
//   stats (may be NULL; epi 0: (sum y, sum y^2)): [256][2][64] partial sums, reduce with wm_bn_finalize*.
int wm_conv64(const float* x, const float* x2, const float* wp, const float* pa, const float* pb, const float* pc,
              const float* bias, const float* e1, const float* ea, const float* eb, float* y, float* stats,
              int B, int T, int KW, int pro, int epi, hipStream_t stream) {
    if (B <= 0 || T <= 0 || (T & 3)) return (int)hipErrorInvalidValue;
    Conv64Args a{x, x2, wp, pa, pb, pc, bias, e1, ea, eb, y, stats, B, T};
    const bool st = stats != nullptr;
    if (KW == 3) {
        if (pro == PRO_NONE && epi == EPI_BIAS)
            return st ? launch_conv64<3, 256, PRO_NONE, EPI_BIAS, true>(a, stream) : launch_conv64<3, 256, PRO_NONE, EPI_BIAS, false>(a, stream);
        if (pro == PRO_BNRELU && epi == EPI_BIAS)
            return st ? launch_conv64<3, 256, PRO_BNRELU, EPI_BIAS, true>(a, stream) : launch_conv64<3, 256, PRO_BNRELU, EPI_BIAS, false>(a, stream);
        if (pro == PRO_BNBWD && epi == EPI_RELUMASK && st) return launch_conv64<3, 128, PRO_BNBWD, EPI_RELUMASK, true>(a, stream);
        if (pro == PRO_BNBWD && epi == EPI_ADD && !st) return launch_conv64<3, 128, PRO_BNBWD, EPI_ADD, false>(a, stream);
        if (pro == PRO_BNBWD && epi == EPI_NONE && !st) return launch_conv64<3, 128, PRO_BNBWD, EPI_NONE, false>(a, stream);
    } else if (KW == 7 && !st) {
        if (pro == PRO_ADDVEC && epi == EPI_BIAS) return launch_conv64<7, 128, PRO_ADDVEC, EPI_BIAS, false>(a, stream);
        if (pro == PRO_NONE && epi == EPI_BIAS) return launch_conv64<7, 128, PRO_NONE, EPI_BIAS, false>(a, stream);
        if (pro == PRO_NONE && epi == EPI_NONE) return launch_conv64<7, 128, PRO_NONE, EPI_NONE, false>(a, stream);
    }
    return (int)hipErrorInvalidValue;
}

// bf16x6 build of the k3 convolution (fp32-grade results on the bf16 matrix cores); wpb from wm_pack_w64_bf.
// Same pro / epi / stats contract as wm_conv64 with KW = 3.
int wm_pack_w64_bf(const float* w, void* wpb, int mode, hipStream_t stream) {
    if (mode < 0 || mode > 1) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(pack_w64_bf_kernel, dim3(48), dim3(256), 0, stream, w, (const float*)nullptr, reinterpret_cast<unsigned short*>(wpb), mode);
    WM_CHECK_LAUNCH();
    return 0;
}
// forward image (mode 0) of w[out][in][k] * row_scale[out]
int wm_pack_w64_bf_scaled(const float* w, const float* row_scale, void* wpb, hipStream_t stream) {
    if (!w || !row_scale || !wpb) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(pack_w64_bf_kernel, dim3(48), dim3(256), 0, stream, w, row_scale, reinterpret_cast<unsigned short*>(wpb), 0);
    WM_CHECK_LAUNCH();
    return 0;
}

static int g_bf_schedule = 2;   // 0: phase-serial kernel | 2: register-resident weights + interleaved pipeline (fastest, needs T % 128 == 0)
int wm_set_conv_bf_schedule(int schedule, hipStream_t) { g_bf_schedule = (schedule == 2) ? 2 : 0; return 0; }

int wm_conv64_bf(const float* x, const float* x2, const void* wpb, const float* pa, const float* pb, const float* pc,
                 const float* bias, const float* e1, const float* ea, const float* eb, float* y, float* stats,
                 int B, int T, int pro, int epi, int arith, hipStream_t stream) {
    if (B <= 0 || T <= 0 || (T & 3)) return (int)hipErrorInvalidValue;
    Conv64Args a{x, x2, reinterpret_cast<const float*>(wpb), pa, pb, pc, bias, e1, ea, eb, y, stats, B, T};
    const bool st = stats != nullptr;
    if (arith == 1) {                    // f16 two-piece split (wpb from wm_pack_w64_h): the forward convolutions of the pipelined kernel only
        if (g_bf_schedule != 2 || (T & 127)) return (int)hipErrorInvalidValue;
        if (pro == PRO_BNRELU && epi == EPI_BNADDRELU && !st) return launch_conv64bf3<PRO_BNRELU, EPI_BNADDRELU, false, true>(a, stream);
        if (epi != EPI_BIAS) return (int)hipErrorInvalidValue;
        if (pro == PRO_NONE) return st ? launch_conv64bf3<PRO_NONE, EPI_BIAS, true, true>(a, stream) : launch_conv64bf3<PRO_NONE, EPI_BIAS, false, true>(a, stream);
        if (pro == PRO_BNRELU) return st ? launch_conv64bf3<PRO_BNRELU, EPI_BIAS, true, true>(a, stream) : launch_conv64bf3<PRO_BNRELU, EPI_BIAS, false, true>(a, stream);
        return (int)hipErrorInvalidValue;
    }
    if (arith != 0) return (int)hipErrorInvalidValue;
    if (pro == PRO_NONE && !st) {        // main14b_2's 64-channel blocks (no BatchNorm): the phase-serial kernel, any T % 4 == 0
        if (epi == EPI_BIASELU) return launch_conv64bf<PRO_NONE, EPI_BIASELU, false>(a, stream);
        if (epi == EPI_BIASADDELU) return launch_conv64bf<PRO_NONE, EPI_BIASADDELU, false>(a, stream);
        if (epi == EPI_MULDELU) return launch_conv64bf<PRO_NONE, EPI_MULDELU, false>(a, stream);
        if (epi == EPI_ADD) return launch_conv64bf<PRO_NONE, EPI_ADD, false>(a, stream);
        if (epi == EPI_NONE) return launch_conv64bf<PRO_NONE, EPI_NONE, false>(a, stream);
    }
    if (g_bf_schedule == 2 && (T & 127) == 0) {
        if (pro == PRO_BNRELU && epi == EPI_BNADDRELU && !st) return launch_conv64bf3<PRO_BNRELU, EPI_BNADDRELU, false>(a, stream);
        if (pro == PRO_NONE && epi == EPI_BIAS) return st ? launch_conv64bf3<PRO_NONE, EPI_BIAS, true>(a, stream) : launch_conv64bf3<PRO_NONE, EPI_BIAS, false>(a, stream);
        if (pro == PRO_BNRELU && epi == EPI_BIAS) return st ? launch_conv64bf3<PRO_BNRELU, EPI_BIAS, true>(a, stream) : launch_conv64bf3<PRO_BNRELU, EPI_BIAS, false>(a, stream);
        if (pro == PRO_BNBWD && epi == EPI_RELUMASK && st) return launch_conv64bf3<PRO_BNBWD, EPI_RELUMASK, true>(a, stream);
        if (pro == PRO_BNBWD && epi == EPI_ADD && !st) return launch_conv64bf3<PRO_BNBWD, EPI_ADD, false>(a, stream);
        if (pro == PRO_BNBWD && epi == EPI_NONE && !st) return launch_conv64bf3<PRO_BNBWD, EPI_NONE, false>(a, stream);
        return (int)hipErrorInvalidValue;
    }
    if (pro == PRO_NONE && epi == EPI_BIAS) return st ? launch_conv64bf<PRO_NONE, EPI_BIAS, true>(a, stream) : launch_conv64bf<PRO_NONE, EPI_BIAS, false>(a, stream);
    if (pro == PRO_BNRELU && epi == EPI_BIAS) return st ? launch_conv64bf<PRO_BNRELU, EPI_BIAS, true>(a, stream) : launch_conv64bf<PRO_BNRELU, EPI_BIAS, false>(a, stream);
    if (pro == PRO_BNBWD && epi == EPI_RELUMASK && st) return launch_conv64bf<PRO_BNBWD, EPI_RELUMASK, true>(a, stream);
    if (pro == PRO_BNBWD && epi == EPI_ADD && !st) return launch_conv64bf<PRO_BNBWD, EPI_ADD, false>(a, stream);
    if (pro == PRO_BNBWD && epi == EPI_NONE && !st) return launch_conv64bf<PRO_BNBWD, EPI_NONE, false>(a, stream);
    return (int)hipErrorInvalidValue;
}

// inference ResBlock in one launch (both BatchNorms folded): see resblock_eval_kernel
int wm_resblock_eval_bf(const float* x, const void* w1pb, const void* w2pb, const float* b1, const float* sc1, const float* sh1,
                        const float* b2, const float* sc2, const float* sh2, float* y, int B, int T, int arith, hipStream_t stream) {
    if (B <= 0 || T <= 0 || (T & 3) || (arith != 0 && arith != 1)) return (int)hipErrorInvalidValue;
    if (!x || !w1pb || !w2pb || !sc1 || !sh1 || !sc2 || !sh2 || !y) return (int)hipErrorInvalidValue;
    RbeArgs a{x, w1pb, w2pb, b1, sc1, sh1, b2, sc2, sh2, y, B, T};
    return arith == 1 ? launch_resblock_eval<true>(a, stream) : launch_resblock_eval<false>(a, stream);
}

// bf16x6 build of the 7-tap ConvTranspose1d (forward: mode 2 image, pro 0|2, epi 0; data gradient: mode 3 image, pro 0, epi 3)
int wm_pack_w64_bf7(const float* w, void* wpb, int mode, hipStream_t stream) {
    if (mode != 2 && mode != 3) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(pack_w64_bf7_kernel, dim3((7 * 4096 + 255) / 256), dim3(256), 0, stream, w, reinterpret_cast<unsigned short*>(wpb), mode);
    WM_CHECK_LAUNCH();
    return 0;
}
int wm_conv64_bf7(const float* x, const void* wpb, const float* vec, const float* bias, float* y, int B, int T, int pro, int epi,
                  int arith, const float* gscale, hipStream_t stream) {
    if (B <= 0 || T <= 0 || (T & 3)) return (int)hipErrorInvalidValue;
    Conv64Args a{x, nullptr, reinterpret_cast<const float*>(wpb), vec, gscale, nullptr, bias, nullptr, nullptr, nullptr, y, nullptr, B, T};
    if (arith == 1) {                     // f16 two-piece split (wpb from wm_pack_w64_h7): the pipelined kernel only
        if (!WM_BF7_PIPE || (T & 127)) return (int)hipErrorInvalidValue;
        if (pro == PRO_NONE && epi == EPI_BIAS) return launch_conv64bf7p<PRO_NONE, EPI_BIAS, true>(a, stream);
        if (pro == PRO_ADDVEC && epi == EPI_BIAS) return launch_conv64bf7p<PRO_ADDVEC, EPI_BIAS, true>(a, stream);
        if (pro == PRO_NONE && epi == EPI_NONE) return launch_conv64bf7p<PRO_NONE, EPI_NONE, true>(a, stream);
        return (int)hipErrorInvalidValue;
    }
    if (arith != 0) return (int)hipErrorInvalidValue;
    a.pb = nullptr;
#if WM_BF7_PIPE
    if ((T & 127) == 0) {                 // register-resident weights + double-buffered input image (conv64bf7p_kernel)
        if (pro == PRO_NONE && epi == EPI_BIAS) return launch_conv64bf7p<PRO_NONE, EPI_BIAS>(a, stream);
        if (pro == PRO_ADDVEC && epi == EPI_BIAS) return launch_conv64bf7p<PRO_ADDVEC, EPI_BIAS>(a, stream);
        if (pro == PRO_NONE && epi == EPI_NONE) return launch_conv64bf7p<PRO_NONE, EPI_NONE>(a, stream);
    }
#endif
    if (pro == PRO_NONE && epi == EPI_BIAS) return launch_conv64bf7<PRO_NONE, EPI_BIAS>(a, stream);
    if (pro == PRO_ADDVEC && epi == EPI_BIAS) return launch_conv64bf7<PRO_ADDVEC, EPI_BIAS>(a, stream);
    if (pro == PRO_NONE && epi == EPI_NONE) return launch_conv64bf7<PRO_NONE, EPI_NONE>(a, stream);
    return (int)hipErrorInvalidValue;
}

int wm_pack_w64_h7(const float* w, void* wph, int mode, hipStream_t stream) {
    if (mode != 2 && mode != 3) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(pack_w64_h7_kernel, dim3(1), dim3(1024), 0, stream, w, reinterpret_cast<unsigned short*>(wph), mode);
    WM_CHECK_LAUNCH();
    return 0;
}

// bf16x6 build of the 7-tap ConvTranspose1d weight gradient (= wm_wgrad64 with KW 7, gpro 0, layout 1): dw [in][out][7]
int wm_wgrad64_bf7(const float* g, const float* x, const float* vec, float* partial, float* dw, float* dbias, int B, int T,
                   int xpro, int accumulate, int arith, const float* gscale, hipStream_t stream) {
    if (B <= 0 || T <= 0 || (T & 3) || (arith != 0 && arith != 1) || (arith == 1 && !gscale)) return (int)hipErrorInvalidValue;
    Wgrad64Args a{g, nullptr, arith == 1 ? gscale : nullptr, nullptr, nullptr, x, vec, nullptr, partial, B, T};
    int grid = 0, rc = (int)hipErrorInvalidValue;
    if (arith == 1) {
        if (xpro == PRO_ADDVEC) rc = launch_wgrad64bf7<PRO_ADDVEC, true>(a, &grid, stream);
        else if (xpro == PRO_NONE) rc = launch_wgrad64bf7<PRO_NONE, true>(a, &grid, stream);
    } else if (xpro == PRO_ADDVEC) rc = launch_wgrad64bf7<PRO_ADDVEC>(a, &grid, stream);
    else if (xpro == PRO_NONE) rc = launch_wgrad64bf7<PRO_NONE>(a, &grid, stream);
    if (rc) return rc;
    const int n = 7 * 4096 + 64;
    hipLaunchKernelGGL(wgrad64_reduce_kernel, dim3((n + 63) / 64), dim3(256), 0, stream, (const float*)partial, grid, 7, 1, dw,
                       dbias, accumulate);
    WM_CHECK_LAUNCH();
    return 0;
}

// bf16x6 build of the k3 weight gradient (same contract as wm_wgrad64 with KW = 3, layout 0)
int wm_wgrad64_bf(const float* g, const float* g2, const float* ga, const float* gb, const float* gc,
                  const float* x, const float* xa, const float* xb, float* partial, float* dw, float* dbias,
                  int B, int T, int gpro, int xpro, int accumulate, hipStream_t stream) {
    if (B <= 0 || T <= 0 || (T & 3)) return (int)hipErrorInvalidValue;
    Wgrad64Args a{g, g2, ga, gb, gc, x, xa, xb, partial, B, T};
    int grid = 0, rc = (int)hipErrorInvalidValue;
    const bool small = (accumulate & 2) != 0;            // output-split build: ~200 registers, co-resident with the LSTM recurrences
    if (gpro == PRO_BNBWD && xpro == PRO_BNRELU)
        rc = small ? launch_wgrad64bf_small<PRO_BNBWD, PRO_BNRELU>(a, &grid, stream) : launch_wgrad64bf<PRO_BNBWD, PRO_BNRELU>(a, &grid, stream);
    else if (gpro == PRO_BNBWD && xpro == PRO_NONE)
        rc = small ? launch_wgrad64bf_small<PRO_BNBWD, PRO_NONE>(a, &grid, stream) : launch_wgrad64bf<PRO_BNBWD, PRO_NONE>(a, &grid, stream);
    else if (gpro == PRO_NONE && xpro == PRO_NONE)            // main14b_2's 64-channel blocks: both operands as they are
        rc = launch_wgrad64bf<PRO_NONE, PRO_NONE>(a, &grid, stream);
    if (rc) return rc;
    const int n = 3 * 4096 + 64;
    hipLaunchKernelGGL(wgrad64_reduce_kernel, dim3((n + 63) / 64), dim3(256), 0, stream, (const float*)partial, grid, 3, 0, dw,
                       dbias, accumulate & 1);
    WM_CHECK_LAUNCH();
    return 0;
}

// data gradient + weight gradient of a k3 convolution in one launch (dwgrad64bf_kernel); T % 64 == 0
// f16 two-piece weight image for wm_dwgrad64_bf arith 1: 2 * 3 * 4096 f16 + 2 floats (mode 0 forward | 1 data gradient, as wm_pack_w64_bf)
int wm_pack_w64_h(const float* w, void* wph, int mode, hipStream_t stream) {
    if (!w || !wph || mode < 0 || mode > 1) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(pack_w64_h_kernel, dim3(1), dim3(1024), 0, stream, w, (const float*)nullptr, reinterpret_cast<unsigned short*>(wph), mode);
    WM_CHECK_LAUNCH();
    return 0;
}
// forward image (mode 0) of w[out][in][k] * row_scale[out] (wm_resblock_eval_bf arith 1)
int wm_pack_w64_h_scaled(const float* w, const float* row_scale, void* wph, hipStream_t stream) {
    if (!w || !row_scale || !wph) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(pack_w64_h_kernel, dim3(1), dim3(1024), 0, stream, w, row_scale, reinterpret_cast<unsigned short*>(wph), 0);
    WM_CHECK_LAUNCH();
    return 0;
}

int wm_dwgrad64_bf(const float* g, const float* g2, const float* ga, const float* gb, const float* gc, const void* wpb,
                   const float* x, const float* xa, const float* xb, const float* e1, const float* ea, const float* eb,
                   float* y, float* stats, float* partial, float* dw, float* dbias, int B, int T, int xpro, int epi, int accumulate,
                   const void* gmask, int arith, const float* gscale, float* dzmax, hipStream_t stream) {
    if (B <= 0 || T <= 0 || (T & 63)) return (int)hipErrorInvalidValue;
    if (!g || !g2 || !ga || !gb || !gc || !wpb || !x || !e1 || !y || !partial || !dw) return (int)hipErrorInvalidValue;
    if (arith != 0 && (arith != 1 || !gscale)) return (int)hipErrorInvalidValue;
    DWArgs a{g, g2, ga, gb, gc, wpb, x, xa, xb, e1, ea, eb, y, stats, partial, B, T, reinterpret_cast<const unsigned*>(gmask),
             epi == EPI_ADDSTATS ? reinterpret_cast<const unsigned*>(eb) : nullptr, epi == EPI_ADDSTATS ? ea : nullptr, gscale, dzmax};
    int grid = 0, rc = (int)hipErrorInvalidValue;
#define WM_DW(E, X, G) (arith ? launch_dwgrad64bf<E, X, G, true>(a, &grid, stream) : launch_dwgrad64bf<E, X, G, false>(a, &grid, stream))
    if (epi == EPI_RELUMASK && xpro == PRO_BNRELU && stats && xa && xb && ea && eb)
        rc = gmask ? WM_DW(EPI_RELUMASK, PRO_BNRELU, true) : WM_DW(EPI_RELUMASK, PRO_BNRELU, false);
    else if (epi == EPI_ADD && xpro == PRO_NONE && !stats)
        rc = gmask ? WM_DW(EPI_ADD, PRO_NONE, true) : WM_DW(EPI_ADD, PRO_NONE, false);
    else if (epi == EPI_ADDSTATS && xpro == PRO_NONE && stats && ea && eb && gmask)
        rc = WM_DW(EPI_ADDSTATS, PRO_NONE, true);
#undef WM_DW
    if (rc) return rc;
    const int n = 3 * 4096 + 64;
    hipLaunchKernelGGL(wgrad64_reduce_kernel, dim3((n + 63) / 64), dim3(256), 0, stream, (const float*)partial, grid, 3, 0, dw,
                       dbias, accumulate & 1);
    WM_CHECK_LAUNCH();
    return 0;
}

// Weight gradient of a 64->64 convolution.  partial: [256][KW*4096+64] scratch.
//   gpro: 0 g as is | 3 ga[c]*g + gb[c] + gc[c]*g2     xpro: 0 | 1 relu(x*xa+xb) | 2 x+xa[b*64+c]
//   layout: 0 Conv1d weight [out][in][KW] | 1 ConvTranspose1d weight [in][out][KW]
int wm_wgrad64(const float* g, const float* g2, const float* ga, const float* gb, const float* gc,
               const float* x, const float* xa, const float* xb, float* partial, float* dw, float* dbias,
               int B, int T, int KW, int gpro, int xpro, int layout, int accumulate, hipStream_t stream) {
    if (B <= 0 || T <= 0 || (T & 3)) return (int)hipErrorInvalidValue;
    Wgrad64Args a{g, g2, ga, gb, gc, x, xa, xb, partial, B, T};
    int grid = 0, rc = (int)hipErrorInvalidValue;
    if (KW == 3 && gpro == PRO_BNBWD && xpro == PRO_BNRELU) rc = launch_wgrad64<3, PRO_BNBWD, PRO_BNRELU>(a, &grid, stream);
    else if (KW == 3 && gpro == PRO_BNBWD && xpro == PRO_NONE) rc = launch_wgrad64<3, PRO_BNBWD, PRO_NONE>(a, &grid, stream);
    else if (KW == 7 && gpro == PRO_NONE && xpro == PRO_ADDVEC) rc = launch_wgrad64<7, PRO_NONE, PRO_ADDVEC>(a, &grid, stream);
    else if (KW == 7 && gpro == PRO_NONE && xpro == PRO_NONE) rc = launch_wgrad64<7, PRO_NONE, PRO_NONE>(a, &grid, stream);
    if (rc) return rc;
    const int n = KW * 4096 + 64;
    hipLaunchKernelGGL(wgrad64_reduce_kernel, dim3((n + 63) / 64), dim3(256), 0, stream, (const float*)partial, grid, KW,
                       layout, dw, dbias, accumulate);
    WM_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
