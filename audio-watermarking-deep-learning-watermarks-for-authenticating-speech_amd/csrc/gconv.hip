// Generic-shape convolution family for the main14b_2 deep-residual variant (py/main14b_2.py:83-224, BASELINE config 5):
// strided Conv1d(k3, stride 2/4/5/8), 1x1 strided skip convs, nn.Linear, Conv1d(k7), and ConvTranspose1d(k=2*st,
// stride st, padding st/2) for channel counts 1..512 -- one implicit-GEMM kernel on the fp32 matrix cores:
//     acc[m][n] = sum_{ci,k} wp[ci*K + k][m] * x[nb][ci][n*S + k - P]
// with M = output rows (channels, or channel*phase for the transposed convolutions), N = output positions, and a
// K-dimension of (input channel, tap) pairs streamed through LDS 8 channels at a time.
//   * A transposed convolution with stride st and kernel 2*st is a 2-tap stride-1 convolution onto Cout*st "phase
//     channels" followed by a pixel shuffle (t' = n*st + phase - padding); the shuffle is applied in the store.
//   * Data gradients reuse the kernel with re-packed weights (dgrad of a strided conv = transposed conv and vice
//     versa); the packing itself is pure data movement done by the host mirror.
// Epilogue: + bias[channel] + vec[nb][channel] (message embedding) + residual, then optional ELU (alpha = 1).
#include "wm_common.hpp"
#include <type_traits>
using namespace wm;

namespace {

struct GConvArgs {
    const float* x;      // [NB][Cin][Lin]
    const float* wp;     // [Cin*K][Mtot]
    const float* bias;   // [Cout] or null
    const float* vec;    // [NB][Cout] or null
    const float* res;    // same layout as y, or null
    float* y;            // [NB][Cout][Lout]
    int NB, Cin, Lin, K, S, P, Mtot, Nout;
    int st;              // 1: plain (Cout == Mtot, t' = n); >1: pixel shuffle, row m = co*st + phase
    int shp;             // padding of the transposed convolution (t' = n*st + phase - shp)
    int Cout, Lout;
    int act;             // 0 none | 1 ELU | 2 multiply by ELU'(aux): aux = `res` holds y = ELU(z) of the tensor the result is a gradient of
    int CI;              // input channels per LDS chunk (even; CI*K <= KR_MAX; CI*XW <= 256*xr_of(BN))
    int XW;              // input span of one N tile: (BN-1)*S + K
    const float* x2;     // optional second source: input channels >= Cin1 are rows of x2 [NB][Cin - Cin1][Lin] (two gradients
    int Cin1;            //   contracted by one launch: a strided conv and its 1x1 skip conv); Cin1 % CI == 0
    int nph;             // phases per output channel when st > 1: row m = co*nph + phase, phase < nph <= st
};

// (channel, tap) rows of the weight image per chunk: sized so that three workgroups' double buffers fit the 160 KB of LDS
constexpr int kr_max_of(int bm) { return bm >= 128 ? 36 : 48; }
constexpr int xr_of(int bn) { return bn >= 256 ? 16 : 10; }   // input-tile elements a thread stages per chunk, by tile width

// Implicit-GEMM tile kernel.  Workgroup = 4 waves as MW x (4/MW); a wave owns WM x WN blocks of 32 x 32 (fp32 MFMA
// 32x32x2, the two k of an instruction = the two channels of a pair at one tap).  Per chunk of CI input channels the
// workgroup stages the input rows [CI][XW] (dword loads, coalesced along time, any alignment) and the weight rows
// [(pair, tap, parity)][BM] (float4 loads when Mtot % 4 == 0) through REGISTERS into the other LDS buffer while the
// matrix cores work on the current one: one barrier per chunk.  Staging costs no address arithmetic inside the loop: every
// element's 32-bit offset is formed once, the chunk advances two wave-uniform base pointers, LDS stores are unconditional
// (the regions are padded to whole 256-thread passes) and zero padding is applied only by the tiles that touch a clip edge.
template <int MW, int WM, int WN, bool VECW>
__global__ __launch_bounds__(256, 3) void gconv2_kernel(GConvArgs a) {
    constexpr int NWN = 4 / MW, BM = MW * WM * 32, BN = NWN * WN * 32, XR = xr_of(BN), KR_MAX = kr_max_of(BM);
    constexpr int WPT = VECW ? (KR_MAX * BM / 4 + 255) / 256 : (KR_MAX * BM + 255) / 256;   // weight pieces per thread
    constexpr int WV = VECW ? 4 : 1;
    extern __shared__ __align__(16) float smem[];
    const int K = a.K, S = a.S, CI = a.CI, XW = a.XW, KR = CI * K;
    const int xsz = CI * XW;
    const int nx = (xsz + 255) >> 8, xreg = nx * 256;                       // padded regions: whole passes of 256 threads
    const int nwp = (KR * BM / WV + 255) >> 8, wreg = nwp * 256 * WV;
    const int bufsz = xreg + wreg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int wm_ = wave % MW, wn = wave / MW;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM, nb = blockIdx.z;
    const int u0 = n0 * S - a.P;
    const bool edge_tile = (u0 < 0) || (u0 + XW > a.Lin);                   // wave-uniform

    // ---- chunk-invariant staging maps
    unsigned xo[XR];            // ci * Lin + clamped column
    unsigned xz = 0;            // bit i: element i is zero padding (outside the clip)
#pragma unroll
    for (int i = 0; i < XR; ++i) {
        const int idx = tid + 256 * i;
        const int ci = min(idx / XW, CI - 1), j = idx - (idx / XW) * XW, u = u0 + j;
        xz |= ((u < 0 || u >= a.Lin) ? 1u : 0u) << i;
        xo[i] = (unsigned)(ci * a.Lin + min(max(u, 0), a.Lin - 1));
    }
    unsigned wo[WPT];           // row * Mtot + m0 + column of the chunk's weight slab (0 for pieces outside it)
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int idx = tid + 256 * i;
        int r, c;
        if (VECW) { r = idx / (BM / 4); c = (idx - r * (BM / 4)) * 4; } else { r = idx / BM; c = idx - r * BM; }
        const int pr = r >> 1, q = pr / K, tap = pr - q * K, ci = 2 * q + (r & 1);
        const bool ok = (r < KR) && (m0 + c < a.Mtot);
        wo[i] = ok ? (unsigned)((ci * K + tap) * a.Mtot + m0 + c) : 0u;
    }

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float xr[XR];
    f32x4 wr4[VECW ? WPT : 1];
    float wr1[VECW ? 1 : WPT];
    // buffer loads: per-lane 32-bit byte offsets (xo / wo, formed once) + a wave-uniform chunk offset; no 64-bit address
    // registers, no vector-ALU address arithmetic
    const int cin1 = a.x2 ? a.Cin1 : a.Cin;
    const wm_srd_t sx = make_srd(a.x + (size_t)nb * cin1 * a.Lin, (size_t)cin1 * a.Lin * sizeof(float));
    const wm_srd_t sx2 = make_srd(a.x2 ? a.x2 + (size_t)nb * (a.Cin - cin1) * a.Lin : a.x, (size_t)(a.Cin - cin1) * a.Lin * sizeof(float));
    const wm_srd_t sw = make_srd(a.wp, (size_t)a.Cin * K * a.Mtot * sizeof(float));
#pragma unroll
    for (int i = 0; i < XR; ++i) xo[i] *= 4u;
#pragma unroll
    for (int i = 0; i < WPT; ++i) wo[i] *= 4u;

    auto load_chunk = [&](int c0) {
        const int crem = a.Cin - c0;
        const bool second = c0 >= cin1;                      // wave-uniform: the chunk lies in the second source
        const wm_srd_t sxc = second ? sx2 : sx;
        const unsigned xs0 = (unsigned)((c0 - (second ? cin1 : 0)) * a.Lin) * 4u, ws0 = (unsigned)(c0 * K * a.Mtot) * 4u;
        if (crem >= CI) {                                    // full chunk: offsets only
#pragma unroll
            for (int i = 0; i < XR; ++i)
                if (i < nx) xr[i] = buf_load(sxc, xo[i], xs0);
#pragma unroll
            for (int i = 0; i < WPT; ++i) {
                if (i < nwp) {
                    if (VECW) wr4[i] = buf_load4(sw, wo[i], ws0);
                    else wr1[i] = buf_load(sw, wo[i], ws0);
                }
            }
        } else {                                             // last chunk of a channel count that CI does not divide
#pragma unroll
            for (int i = 0; i < XR; ++i) {
                if (i < nx) {
                    const int idx = tid + 256 * i;
                    const int ci = min(idx / XW, CI - 1);
                    const float v = buf_load(sxc, xo[i] - (unsigned)((ci - min(ci, crem - 1)) * a.Lin) * 4u, xs0);
                    xr[i] = ci < crem ? v : 0.f;
                }
            }
#pragma unroll
            for (int i = 0; i < WPT; ++i) {
                if (i < nwp) {
                    const int idx = tid + 256 * i;
                    const int r = VECW ? idx / (BM / 4) : idx / BM;
                    const int ci = 2 * ((r >> 1) / K) + (r & 1);
                    const bool ok = ci < crem;
                    if (VECW) {
                        const f32x4 v = buf_load4(sw, ok ? wo[i] : 0u, ws0);
                        wr4[i] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
                    } else {
                        const float v = buf_load(sw, ok ? wo[i] : 0u, ws0);
                        wr1[i] = ok ? v : 0.f;
                    }
                }
            }
        }
        if (edge_tile) {
#pragma unroll
            for (int i = 0; i < XR; ++i)
                if (i < nx) xr[i] = ((xz >> i) & 1u) ? 0.f : xr[i];
        }
    };
    auto store_chunk = [&](float* buf) {
#pragma unroll
        for (int i = 0; i < XR; ++i)
            if (i < nx) buf[tid + 256 * i] = xr[i];
        float* Ws = buf + xreg;
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            if (i < nwp) {
                if (VECW) *reinterpret_cast<f32x4*>(Ws + (tid + 256 * i) * 4) = wr4[i];
                else Ws[tid + 256 * i] = wr1[i];
            }
        }
    };

    const int nchunks = (a.Cin + CI - 1) / CI;
    load_chunk(0);
    store_chunk(smem);
    __syncthreads();
    const int npair = KR >> 1;
    for (int ch = 0; ch < nchunks; ++ch) {
        const float* cur = smem + (ch & 1) * bufsz;
        if (ch + 1 < nchunks) load_chunk((ch + 1) * CI);
        const float* wa = cur + xreg + half * BM + wm_ * (WM * 32) + l31;
        const float* xp = cur + half * XW + (wn * (WN * 32) + l31) * S;
        // software-pipelined pair loop, two pairs per trip with two operand sets: each set is read one MFMA group (4 x 64
        // cycles) before it is used, so no LDS latency is exposed
        float av[WM], bv[WN], an[WM], bn[WN];
        int tap = 0, xoff = 0;
        auto advance = [&]() {
            ++tap; ++xoff;
            if (tap == K) { tap = 0; xoff += 2 * XW - K; }
            wa += 2 * BM;
        };
        auto rd = [&](float (&A_)[WM], float (&B_)[WN]) {
#pragma unroll
            for (int i = 0; i < WM; ++i) A_[i] = wa[i * 32];
#pragma unroll
            for (int j = 0; j < WN; ++j) B_[j] = xp[xoff + j * 32 * S];
        };
        auto mm = [&](const float (&A_)[WM], const float (&B_)[WN]) {
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] = mfma32(A_[i], B_[j], acc[i][j]);
        };
        rd(av, bv);
        // (reads past the last pair stay inside the workgroup's LDS allocation: see the slack in launch_gconv2)
        for (int p = 0; p < npair; p += 2) {
            advance();
            rd(an, bn);
            __builtin_amdgcn_sched_barrier(0);
            mm(av, bv);
            __builtin_amdgcn_sched_barrier(0);
            advance();
            rd(av, bv);
            __builtin_amdgcn_sched_barrier(0);
            if (p + 1 < npair) mm(an, bn);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ch + 1 < nchunks) store_chunk(smem + ((ch + 1) & 1) * bufsz);
        __syncthreads();
    }

    // ---- epilogue: + bias + per-clip vector + residual, ELU, (pixel-shuffled) store.  Addresses are 32-bit offsets into
    // this clip's output rows (buffer stores: per-lane offset + a wave-uniform row offset), so a stored element costs its
    // arithmetic only; tiles that lie inside the output skip every bound check.
    const size_t clip = (size_t)nb * a.Cout * a.Lout;
    const wm_srd_t sy = make_srd(a.y + clip, (size_t)a.Cout * a.Lout * sizeof(float));
    const wm_srd_t sr = make_srd(a.res ? a.res + clip : a.y + clip, (size_t)a.Cout * a.Lout * sizeof(float));
    const bool has_res = a.res != nullptr && a.act != 2, act = a.act == 1, mul_dact = a.act == 2 && a.res != nullptr;
    const float* vecb = a.vec ? a.vec + (size_t)nb * a.Cout : nullptr;
    // per 32-row block: (1) + bias (+ per-clip vector), (2) all residual loads issued together, one wait, (3) ELU, (4) stores --
    // the optional stages are wave-uniform branches around straight-line code, never a branch per element
    const unsigned inv = (65536u + (unsigned)a.nph - 1u) / (unsigned)a.nph;     // exact m / nph for m < 8192, nph <= 8
    const bool shuffle = a.st > 1;
    int ncol[WN];                                                               // output position index n of this lane
#pragma unroll
    for (int j = 0; j < WN; ++j) ncol[j] = n0 + (wn * WN + j) * 32 + l31;
    const int tlo = n0 * a.st - a.shp, thi = (n0 + BN - 1) * a.st + a.st - 1 - a.shp;
    const bool inner = (m0 + BM <= a.Mtot) && (n0 + BN <= a.Nout) && (!shuffle || (tlo >= 0 && thi < a.Lout));
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
        for (int rh = 0; rh < 16; rh += 8) {                                    // 8 rows at a time: bounds the live offsets / loads
            unsigned off[8][WN];                                                // byte offset inside the clip, 0xffffffff = no store
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) {
                const int r = rh + r8;
                const int m = m0 + (wm_ * WM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int mc = min(m, a.Mtot - 1);
                int co = mc, ph = 0;
                if (shuffle) { co = (int)(((unsigned)mc * inv) >> 16); ph = mc - co * a.nph; }
                float add = a.bias ? a.bias[co] : 0.f;
                if (vecb) add += vecb[co];
                const int rowv = co * a.Lout + ph - (shuffle ? a.shp : 0);
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    const int t = shuffle ? ncol[j] * a.st + ph - a.shp : ncol[j];
                    const bool ok = inner || (m < a.Mtot && ncol[j] < a.Nout && t >= 0 && t < a.Lout);
                    off[r8][j] = ok ? (unsigned)(rowv + (shuffle ? ncol[j] * a.st : ncol[j])) * 4u : 0xffffffffu;
                    acc[i][j][r] += add;
                }
            }
            if (has_res) {
                float rv[8][WN];
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8)
#pragma unroll
                    for (int j = 0; j < WN; ++j) rv[r8][j] = buf_load(sr, off[r8][j], 0u);    // 0xffffffff is out of range: reads 0
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8)
#pragma unroll
                    for (int j = 0; j < WN; ++j) acc[i][j][rh + r8] += rv[r8][j];
            }
            if (act) {
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8)
#pragma unroll
                    for (int j = 0; j < WN; ++j) acc[i][j][rh + r8] = elu1(acc[i][j][rh + r8]);
            }
            if (mul_dact) {                                                     // dz = g * ELU'(z), ELU'(z) = 1 (y > 0) | y + 1
                float yv[8][WN];
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8)
#pragma unroll
                    for (int j = 0; j < WN; ++j) yv[r8][j] = buf_load(sr, off[r8][j], 0u);
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8)
#pragma unroll
                    for (int j = 0; j < WN; ++j) acc[i][j][rh + r8] *= (yv[r8][j] > 0.f ? 1.f : yv[r8][j] + 1.f);
            }
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8)
#pragma unroll
                for (int j = 0; j < WN; ++j) buf_store(sy, acc[i][j][rh + r8], off[r8][j], 0u);   // out-of-range offsets are dropped by the hardware
        }
    }
}

// largest even ci <= lim that divides Cin (so no partial tail chunk exists), unless that costs more than half the chunk
static int pick_ci(int Cin, int lim) {
    lim &= ~1;
    if (lim >= Cin + (Cin & 1)) return Cin + (Cin & 1);
    for (int c = lim; c >= 2 && 2 * c > lim; c -= 2)
        if (Cin % c == 0) return c;
    return lim;
}

template <int MW, int WM, int WN, bool VECW>
int launch_gconv2(GConvArgs a, hipStream_t stream) {
    constexpr int BM = MW * WM * 32, BN = (4 / MW) * WN * 32, WV = VECW ? 4 : 1, XR = xr_of(BN), KR_MAX = kr_max_of(BM);
    a.XW = (BN - 1) * a.S + a.K;
    int lim = KR_MAX / a.K;                                 // weight rows per chunk <= KR_MAX
    const int cx = (256 * XR) / a.XW;                       // staged input elements per thread <= XR
    if (cx < lim) lim = cx;
    if (lim < 2) return (int)hipErrorInvalidValue;
    a.CI = pick_ci(a.Cin, lim);
    if (a.x2) {                                             // chunks must not straddle the two sources
        int c = lim & ~1;
        while (c >= 2 && (a.Cin1 % c != 0 || a.Cin % c != 0)) c -= 2;
        if (c < 2) return (int)hipErrorInvalidValue;
        a.CI = c;
    }
    const int KR = a.CI * a.K;
    const size_t xreg = (((size_t)a.CI * a.XW + 255) >> 8) * 256, wreg = (((size_t)KR * BM / WV + 255) >> 8) * 256 * WV;
    // + slack: the pipelined pair loop reads up to two pairs past the chunk (four weight rows, two input rows further)
    const size_t lds = (2 * (xreg + wreg) + 4 * BM + 4 * (size_t)a.XW + 64) * sizeof(float);
    if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
    auto kern = gconv2_kernel<MW, WM, WN, VECW>;
    static wm::DevOnce once;
    if (!wm::dev_done(once)) {
        WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        wm::dev_mark(once);
    }
    dim3 grid((a.Nout + BN - 1) / BN, (a.Mtot + BM - 1) / BM, a.NB);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a);
    WM_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// The same implicit GEMM on the f16 matrix cores with the f16 two-piece split (hi = f16(v s), lo = f16(v s - hi); three products
// lo hi, hi lo, hi hi per product, fp32 accumulate: fp32-grade) for layers whose input channel count is a multiple
// of 16 -- v_mfma_f32_32x32x16_f16 contracts 16 (channel, tap) pairs per instruction where v_mfma_f32_32x32x2_f32 contracts 2 at half
// the rate, so the wide layers gain most.  The K dimension is walked as (16-channel chunk, tap):
//   * weights: a pre-packed f16 image [chunk][tap][Mtot][16 channels] x {hi, lo} (wm_gconv_pack_h: w * ws, ws a power of two from
//     max |w|, {ws, 1 / ws} behind the image).  An A fragment (one output row, 8 consecutive channels) is ONE 16-byte global load,
//     coalesced over the wave's 32 rows: the weights never pass through LDS (L2 / L1 serve the re-use between tiles and waves).
//   * input: a chunk's [16][XW] window is staged global -> registers -> LDS as a channel-minor image [position][16 channels] x {hi, lo}
//     (pitch 24 halves), one chunk ahead of the matrix phase; a B fragment (one output position, 8 consecutive channels at one tap) is
//     one aligned ds_read_b128 at row position * S + tap.  gscale (data gradients): the input is multiplied by gs before the split
//     and clamped to the f16 range; the accumulators leave times 1 / (ws gs).
// Epilogue as gconv2_kernel's.
// ---------------------------------------------------------------------------------------------------------------
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct GConvHArgs {
    GConvArgs g;
    const unsigned short* wph;      // [2 pieces][Cin / 16][K][Mtot][16] f16, then {ws, 1 / ws}
    const float* gscale;            // {gs, 1 / gs} or null
    float* ymax;                    // optional: max |y| over everything the launch stores (atomic max of non-negative floats as ints; zeroed by the caller)
};

template <int MW, int WM, int WN, int XRH>   // XRH: staged window elements (channel pairs) per thread, 8 XW <= 256 XRH
__global__ __launch_bounds__(256) void gconvh_kernel(GConvHArgs ha) {
    const GConvArgs& a = ha.g;
    constexpr int NWN = 4 / MW, BM = MW * WM * 32, BN = NWN * WN * 32, PX = 24;
    extern __shared__ __align__(16) unsigned char smem_h[];
    unsigned short* Xs = reinterpret_cast<unsigned short*>(smem_h);      // [2 buffers][2 pieces][XWP][PX]
    const int K = a.K, S = a.S, XW = a.XW;
    const int XWP = XW + 8;                                                 // slack rows: discarded columns read past the window
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5, l31 = lane & 31;
    const int wm_ = wave % MW, wn = wave / MW;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM, nb = blockIdx.z;
    const int u0 = n0 * S - a.P;
    const int nchunks = a.Cin >> 4;
    const size_t piece = (size_t)nchunks * K * a.Mtot * 16;                 // halves per weight piece
    const float* tail = reinterpret_cast<const float*>(ha.wph + 2 * piece);
    const float gs = ha.gscale ? ha.gscale[0] : 1.f;
    const float dinv = tail[1] * (ha.gscale ? ha.gscale[1] : 1.f);

    // ---- staging map: element i of a thread = (channel pair cp, window position j): idx = tid + 256 i = cp * XW + j
    const int nx = (8 * XW + 255) >> 8;
    unsigned xo[XRH];            // byte offset of channel 2 cp at the clamped position
    unsigned xl[XRH];            // LDS dword index (position * PX / 2 + cp), 0xffffffff: nothing to write
    unsigned xz = 0;             // bit i: zero padding
#pragma unroll
    for (int i = 0; i < XRH; ++i) {
        const int idx = tid + 256 * i, cp = idx / XW, j = idx - cp * XW, u = u0 + j;
        const bool live = (i < nx) && (cp < 8);
        xz |= ((u < 0 || u >= a.Lin) ? 1u : 0u) << i;
        xo[i] = (unsigned)((2 * min(cp, 7)) * a.Lin + min(max(u, 0), a.Lin - 1)) * 4u;
        xl[i] = live ? (unsigned)(j * (PX / 2) + cp) : 0xffffffffu;
    }
    const int cin1 = a.x2 ? a.Cin1 : a.Cin;
    const wm_srd_t sx = make_srd(a.x + (size_t)nb * cin1 * a.Lin, (size_t)cin1 * a.Lin * sizeof(float));
    const wm_srd_t sx2 = make_srd(a.x2 ? a.x2 + (size_t)nb * (a.Cin - cin1) * a.Lin : a.x, (size_t)(a.Cin - cin1) * a.Lin * sizeof(float));
    const wm_srd_t sw = make_srd(reinterpret_cast<const float*>(ha.wph), 2 * piece * sizeof(unsigned short));
    float xa[XRH], xb[XRH];
    auto load_chunk = [&](int c) {
        const int c0 = 16 * c;
        const bool second = c0 >= cin1;
        const wm_srd_t sxc = second ? sx2 : sx;
        const unsigned s0 = (unsigned)((c0 - (second ? cin1 : 0)) * a.Lin) * 4u;
#pragma unroll
        for (int i = 0; i < XRH; ++i)
            if (i < nx) { xa[i] = buf_load(sxc, xo[i], s0); xb[i] = buf_load(sxc, xo[i] + (unsigned)a.Lin * 4u, s0); }
    };
    auto store_chunk = [&](unsigned short* buf) {
        unsigned* hi32 = reinterpret_cast<unsigned*>(buf);
        unsigned* lo32 = reinterpret_cast<unsigned*>(buf + XWP * PX);
#pragma unroll
        for (int i = 0; i < XRH; ++i) {
            if (i < nx) {
                const bool z = (xz >> i) & 1u;
                const float va = z ? 0.f : __builtin_amdgcn_fmed3f(xa[i] * gs, -6.0e4f, 6.0e4f);
                const float vb = z ? 0.f : __builtin_amdgcn_fmed3f(xb[i] * gs, -6.0e4f, 6.0e4f);
                const h16x2 h_ = __builtin_convertvector(f32x2{va, vb}, h16x2);
                const h16x2 l_ = __builtin_convertvector(f32x2{va - (float)h_.x, vb - (float)h_.y}, h16x2);
                if (xl[i] != 0xffffffffu) { hi32[xl[i]] = __builtin_bit_cast(unsigned, h_); lo32[xl[i]] = __builtin_bit_cast(unsigned, l_); }
            }
        }
    };
    const int bufh = 2 * XWP * PX;                                          // halves per buffer (two pieces)
    // the slack rows are read by discarded columns only: keep them finite
    for (int i = tid; i < 2 * 2 * 8 * PX / 2; i += 256) {
        const int bq = i / (8 * PX / 2), r = i - bq * (8 * PX / 2);          // bq = buffer * 2 + piece
        reinterpret_cast<unsigned*>(Xs + (bq >> 1) * bufh + (bq & 1) * XWP * PX + XW * PX)[r] = 0u;
    }
    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // weight fragment of (chunk c, tap, row tile i, piece p): 16 bytes at ((c K + tap) Mtot + row) * 16 + 8 half halves; rows past Mtot read 0
    unsigned wrow[WM], wrow_lo[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int m = m0 + (wm_ * WM + i) * 32 + l31;
        wrow[i] = (m < a.Mtot) ? (unsigned)(m * 16 + 8 * half) * 2u : 0xfffffff0u;
        wrow_lo[i] = (m < a.Mtot) ? wrow[i] + (unsigned)(piece * 2) : 0xfffffff0u;
    }
    auto load_w = [&](int c, int tap, u32x4 (&A)[WM][2]) {
        const unsigned s0 = (unsigned)(((size_t)(c * K + tap) * a.Mtot) * 16) * 2u;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            A[i][0] = __builtin_bit_cast(u32x4, buf_load4(sw, wrow[i], s0));
            A[i][1] = __builtin_bit_cast(u32x4, buf_load4(sw, wrow_lo[i], s0));
        }
    };
    auto mma = [&](const u32x4& A_, const u32x4& B_, f32x16 c) -> f32x16 {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, A_), __builtin_bit_cast(h16x8, B_), c, 0, 0, 0);
    };
    load_chunk(0);
    store_chunk(Xs);
    __syncthreads();
    // weight fragments of this k-step and of the next two (in flight): an L2 hit takes longer than the 6-12 MFMAs of one k-step
    u32x4 Ac[WM][2], An[WM][2], An2[WM][2];
    const int nq = nchunks * K;                                             // k-steps q = c K + tap
    auto load_q = [&](int q, u32x4 (&A)[WM][2]) {
        if (q < nq) { const int c_ = q / K; load_w(c_, q - c_ * K, A); }
    };
    load_q(0, Ac); load_q(1, An);
    for (int c = 0; c < nchunks; ++c) {
        const unsigned short* cur = Xs + (c & 1) * bufh;
        if (c + 1 < nchunks) load_chunk(c + 1);
        const unsigned short* xrow = cur + ((wn * (WN * 32) + l31) * S) * PX + 8 * half;
        u32x4 Bh[WN], Bl[WN], Bhn[WN], Bln[WN];          // this tap's input fragments, the next tap's (LDS reads in flight)
        auto load_b = [&](int tap, u32x4 (&H_)[WN], u32x4 (&L_)[WN]) {
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const unsigned short* pB = xrow + (j * 32 * S + tap) * PX;
                H_[j] = *reinterpret_cast<const u32x4*>(pB);
                L_[j] = *reinterpret_cast<const u32x4*>(pB + XWP * PX);
            }
        };
        load_b(0, Bh, Bl);
        for (int tap = 0; tap < K; ++tap) {
            load_q(c * K + tap + 2, An2);
            if (tap + 1 < K) load_b(tap + 1, Bhn, Bln);
            // piece products outermost: the WM x WN accumulators form independent chains between two dependent MFMAs
#pragma unroll
            for (int pp = 0; pp < 3; ++pp)
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = mma(Ac[i][pp == 0 ? 1 : 0], pp == 1 ? Bl[j] : Bh[j], acc[i][j]);     // lo hi, hi lo, hi hi
#pragma unroll
            for (int i = 0; i < WM; ++i) { Ac[i][0] = An[i][0]; Ac[i][1] = An[i][1]; An[i][0] = An2[i][0]; An[i][1] = An2[i][1]; }
#pragma unroll
            for (int j = 0; j < WN; ++j) { Bh[j] = Bhn[j]; Bl[j] = Bln[j]; }
        }
        if (c + 1 < nchunks) store_chunk(Xs + ((c + 1) & 1) * bufh);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] *= dinv;

    // ---- epilogue (gconv2_kernel's): + bias + per-clip vector + residual, ELU / ELU', (pixel-shuffled) store
    float vmax = 0.f;
    const size_t clip = (size_t)nb * a.Cout * a.Lout;
    const wm_srd_t sy = make_srd(a.y + clip, (size_t)a.Cout * a.Lout * sizeof(float));
    const wm_srd_t sr = make_srd(a.res ? a.res + clip : a.y + clip, (size_t)a.Cout * a.Lout * sizeof(float));
    const bool has_res = a.res != nullptr && a.act != 2, act = a.act == 1, mul_dact = a.act == 2 && a.res != nullptr;
    const float* vecb = a.vec ? a.vec + (size_t)nb * a.Cout : nullptr;
    const unsigned inv = (65536u + (unsigned)a.nph - 1u) / (unsigned)a.nph;
    const bool shuffle = a.st > 1;
    int ncol[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) ncol[j] = n0 + (wn * WN + j) * 32 + l31;
    const int tlo = n0 * a.st - a.shp, thi = (n0 + BN - 1) * a.st + a.st - 1 - a.shp;
    const bool inner = (m0 + BM <= a.Mtot) && (n0 + BN <= a.Nout) && (!shuffle || (tlo >= 0 && thi < a.Lout));
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
        for (int rh = 0; rh < 16; rh += 8) {
            unsigned off[8][WN];
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) {
                const int r = rh + r8;
                const int m = m0 + (wm_ * WM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int mc = min(m, a.Mtot - 1);
                int co = mc, ph = 0;
                if (shuffle) { co = (int)(((unsigned)mc * inv) >> 16); ph = mc - co * a.nph; }
                float add = a.bias ? a.bias[co] : 0.f;
                if (vecb) add += vecb[co];
                const int rowv = co * a.Lout + ph - (shuffle ? a.shp : 0);
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    const int t = shuffle ? ncol[j] * a.st + ph - a.shp : ncol[j];
                    const bool ok = inner || (m < a.Mtot && ncol[j] < a.Nout && t >= 0 && t < a.Lout);
                    off[r8][j] = ok ? (unsigned)(rowv + (shuffle ? ncol[j] * a.st : ncol[j])) * 4u : 0xffffffffu;
                    acc[i][j][r] += add;
                }
            }
            if (has_res) {
                float rv[8][WN];
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8)
#pragma unroll
                    for (int j = 0; j < WN; ++j) rv[r8][j] = buf_load(sr, off[r8][j], 0u);
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8)
#pragma unroll
                    for (int j = 0; j < WN; ++j) acc[i][j][rh + r8] += rv[r8][j];
            }
            if (act) {
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8)
#pragma unroll
                    for (int j = 0; j < WN; ++j) acc[i][j][rh + r8] = elu1(acc[i][j][rh + r8]);
            }
            if (mul_dact) {
                float yv[8][WN];
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8)
#pragma unroll
                    for (int j = 0; j < WN; ++j) yv[r8][j] = buf_load(sr, off[r8][j], 0u);
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8)
#pragma unroll
                    for (int j = 0; j < WN; ++j) acc[i][j][rh + r8] *= (yv[r8][j] > 0.f ? 1.f : yv[r8][j] + 1.f);
            }
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    buf_store(sy, acc[i][j][rh + r8], off[r8][j], 0u);
                    vmax = fmaxf(vmax, off[r8][j] != 0xffffffffu ? fabsf(acc[i][j][rh + r8]) : 0.f);
                }
        }
    }
    if (ha.ymax) {                                   // the next consumer's gradient scale without a pass over y
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
        if (lane == 0 && vmax == vmax && vmax < 3.0e38f) atomicMax(reinterpret_cast<int*>(ha.ymax), __float_as_int(vmax));
    }
}

template <int MW, int WM, int WN, int XRH>
int launch_gconvh_x(GConvHArgs ha, hipStream_t stream) {
    GConvArgs& a = ha.g;
    constexpr int BM = MW * WM * 32, BN = (4 / MW) * WN * 32;
    a.XW = (BN - 1) * a.S + a.K;
    if (8 * a.XW > 256 * XRH) return (int)hipErrorInvalidValue;         // staged window elements per thread <= XRH
    const size_t lds = (size_t)2 * 2 * (a.XW + 8) * 24 * sizeof(unsigned short);
    if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
    auto kern = gconvh_kernel<MW, WM, WN, XRH>;
    static wm::DevOnce once;
    if (!wm::dev_done(once)) {
        WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        wm::dev_mark(once);
    }
    dim3 grid((a.Nout + BN - 1) / BN, (a.Mtot + BM - 1) / BM, a.NB);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, ha);
    WM_CHECK_LAUNCH();
    return 0;
}
template <int MW, int WM, int WN>
int launch_gconvh(GConvHArgs ha, hipStream_t stream) {
    constexpr int BN = (4 / MW) * WN * 32;
    const int XW = (BN - 1) * ha.g.S + ha.g.K;
    if (8 * XW <= 256 * 6) return launch_gconvh_x<MW, WM, WN, 6>(ha, stream);     // stride-1 layers: a short window, 56 registers less
    return launch_gconvh_x<MW, WM, WN, 20>(ha, stream);
}

// f16 two-piece image of a wm_gconv weight matrix wp [Cin * K][Mtot] (row ci * K + tap): out[piece][ci / 16][tap][m][ci % 16];
// the scale {ws, 1 / ws} sits behind the image (written by gscale_from_max beforehand)
// (ws_fixed > 0: that scale is used and written behind the image by workgroup 0 -- no pass for max |w|)
// mode -1: wp is the wm_gconv matrix [Cin * K][Mtot].  mode 0 / 1: wp is a Conv1d weight w [Cout][Cin_w][K] itself and the re-indexing of
// the host mirror happens here -- 0 forward (GEMM channel = ci, row = co), 1 stride-1 data gradient (GEMM channel = co, row = ci, taps
// flipped): the permute / flip / contiguous launches of the mirror disappear
__global__ __launch_bounds__(256) void gconv_pack_h_kernel(const float* __restrict__ wp, unsigned short* __restrict__ out, int Cin, int K, int Mtot,
                                                           float ws_fixed, int mode) {
    const size_t n = (size_t)Cin * K * Mtot, piece = n;
    float* tail = reinterpret_cast<float*>(out + 2 * piece);
    const float ws = ws_fixed > 0.f ? ws_fixed : tail[0];
    if (ws_fixed > 0.f && blockIdx.x == 0 && threadIdx.x == 0) { tail[0] = ws_fixed; tail[1] = 1.f / ws_fixed; }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        // i walks the OUTPUT image: ((c K + tap) Mtot + m) * 16 + e
        const int e = (int)(i & 15);
        const size_t q = i >> 4;
        const int m = (int)(q % Mtot);
        const size_t ct = q / Mtot;
        const int tap = (int)(ct % K), c = (int)(ct / K);
        const int qch = 16 * c + e;                  // GEMM input channel
        const size_t src = mode < 0 ? ((size_t)qch * K + tap) * Mtot + m
                         : mode == 0 ? ((size_t)m * Cin + qch) * K + tap                    // w[co = m][ci = q][tap]
                                     : ((size_t)qch * Mtot + m) * K + (K - 1 - tap);        // w[co = q][ci = m][K - 1 - tap]
        const float v = __builtin_amdgcn_fmed3f(wp[src] * ws, -6.0e4f, 6.0e4f);
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        out[i] = __builtin_bit_cast(unsigned short, hi);
        out[piece + i] = __builtin_bit_cast(unsigned short, lo);
    }
}

// [A][C][L] -> [L][C][A]   (batch-major <-> time-major sequence layouts around the LSTM)
__global__ void permute_acl_kernel(const float* __restrict__ x, float* __restrict__ y, int A, int C, int L) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)A * C * L) return;
    const int l = (int)(i % L), c = (int)((i / L) % C), aa = (int)(i / ((size_t)L * C));
    y[((size_t)l * C + c) * A + aa] = x[i];
}

// ---------------------------------------------------------------------------------------------------------------
// Generic weight gradient as ONE plain GEMM with the taps folded into the column index:
//     G[a][j] = sum_{nb,t} A[nb][a][t] * Bx[nb][b][t + k - P],      j = b*K + k   (= the memory order of dW[a][b][k])
//   Conv1d (stride 1): A = dL/dy (rows = out channels),  Bx = layer input  -> dW[out][in][k]
//   Linear / LSTM:     K = 1 (the "time" axis is whatever axis is contiguous: positions or batch columns)
//   strided Conv1d and ConvTranspose1d: the strided operand is first re-laid by wm_gather_taps (tap planes / stride
//   phases become channels), which turns them into stride-1 problems with K = 1 / K = 2; `remap` puts the columns back
//   into the weight's own order in the reduce.
// M = Ca rows, N = Cb*K columns, contraction over (clip, position).  A wave owns a 32 x (32*NS) block (NS <= 4
// accumulators, whatever K is: a 8 x 8 x 3 layer is ONE MFMA per position pair, not three); the workgroup's four waves
// split rows (WA), columns (WJW) and -- when the matrix is smaller than that -- the positions of each chunk (WT, summed in
// fixed order at the end).  Per chunk of TC positions the A rows and the Bx rows the tile's columns touch are staged
// through registers (buffer loads: wave-uniform row offset + one per-lane constant, no vector-ALU address arithmetic)
// into LDS while the matrix cores work on the previous chunk; lanes run along the position axis (coalesced), LDS pitches
// are chosen so both operand reads are conflict-free (A: odd pitch, read down a column; Bx: pitch == K mod 32, so
// column j of the tile reads bank j).
// Split-K over gridDim.y: every workgroup writes its partial tile to slab[z] and a fixed-order reduce kernel forms the
// result -- no float atomics, bit-reproducible.  dbias[a] = sum A[a][t] rides along (column tile 0, VALU sums of the
// staged rows).
struct GWArgs {
    const float* A;
    const float* Bx;
    float* slab;     // [gz][Ca][NJ]
    float* slabb;    // [gz][Ca] or null
    int NB, Ca, Cb, La, Lb, K, P;
    long long bcs;   // floats between consecutive clips of Bx (>= Cb*Lb: Bx may be a channel slice of a wider tensor)
    int NJ;          // Cb*K
    int TC;          // positions per chunk: multiple of 8
    int nchunks;     // chunks per clip
    int nsub;        // 32-column blocks per wave (1..4)
    int WJW, WT;     // waves along the columns / along the positions of a chunk (WA*WJW*WT == 4)
    int ntj;         // column tiles
    int NBCH;        // Bx rows (channels) staged per workgroup
    int ncb;         // 64-position blocks per staged row (A and Bx alike)
    int AP, BP;      // LDS pitches
    int nwork;       // NB * nchunks
};

constexpr int GW_RA = 16;   // A staging units per wave and chunk (unit = one wave-wide dword load)
constexpr int GW_RB = 24;   // Bx staging units per wave and chunk

template <int WA, int NS, bool GEO1>   // GEO1: one 64-position block per staged row (ncb == 1, rows wider than 32): no unit decode
__global__ __launch_bounds__(256, 3) void gwgrad2_kernel(GWArgs g) {
    constexpr int TA = 32 * WA;
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave % WA, rest = wave / WA, wj = rest % g.WJW, wt = rest / g.WJW;
    const int ta = blockIdx.x / g.ntj, tj = blockIdx.x - ta * g.ntj;
    const int TJ = g.WJW * 32 * g.nsub;
    const int a0 = ta * TA, j0 = tj * TJ, b0 = j0 / g.K;
    const int TC = g.TC, BW = TC + g.K - 1, ncb = GEO1 ? 1 : g.ncb;
    // staging geometry (wave-uniform).  A unit is one wave-wide dword load: 64 positions of one row, or (rows <= 32 wide)
    // 32 positions of two rows.  Only rows that exist are staged; LDS rows are padded to whole units (no store predicate).
    const int rpu = GEO1 ? 1 : (BW <= 32 ? 2 : 1);          // rows per unit, A and Bx alike (BW >= TC)
    const int a_rows = min(TA, g.Ca - a0), b_rows = min(g.NBCH, g.Cb - b0);
    const int a_units = ((a_rows + rpu - 1) / rpu) * ncb, b_units = ((b_rows + rpu - 1) / rpu) * ncb;
    const int row_l = rpu == 2 ? half : 0, col_l = rpu == 2 ? l31 : lane;
    const int asz = TA * g.AP, zrow = asz + g.NBCH * g.BP;         // + one all-zero row: what columns past NJ read
    const unsigned inv_ncb = (65536u + (unsigned)ncb - 1u) / (unsigned)ncb;
    const unsigned vA = (unsigned)(row_l * g.La + col_l), vB = (unsigned)(row_l * g.Lb + col_l);
    const unsigned lA = (unsigned)(row_l * g.AP + col_l), lB = (unsigned)(asz + row_l * g.BP + col_l);
    const bool even_rows = ((a_rows % rpu) == 0) && ((b_rows % rpu) == 0);

    // per-lane column maps of the MFMA B operand (LDS float index of position 0 of the column)
    int boff[NS];
#pragma unroll
    for (int jt = 0; jt < NS; ++jt) {
        const int j = j0 + (wj * g.nsub + jt) * 32 + l31;
        const int b = j / g.K, k = j - b * g.K;
        const bool ok = (jt < g.nsub) && (j < g.NJ);
        boff[jt] = (ok ? asz + (b - b0) * g.BP + k : zrow) + half;
    }
    f32x16 acc[NS];
#pragma unroll
    for (int jt = 0; jt < NS; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[jt][r] = 0.f;
    float bsum = 0.f;
    const bool do_bias = (g.slabb != nullptr) && (tj == 0);
    const int bparts = 256 / TA, brow = tid % TA, bpart = tid / TA, bcpp = TC / bparts;

    float ra[GW_RA], rb[GW_RB];
    auto load_work = [&](int w) {
        int wv = wave;
        asm volatile("" : "+s"(wv));                         // opaque: the per-unit scalars are not hoisted out of the chunk loop
        const int nb = w / g.nchunks, t0 = (w - nb * g.nchunks) * TC, u0 = t0 - g.P;
        const float* Ab = g.A + (size_t)nb * g.Ca * g.La;
        const float* Bb = g.Bx + (size_t)nb * g.bcs;
        // fast path: every lane of every unit reads inside its own row of the clip -- buffer loads with a wave-uniform
        // offset (row, chunk start) plus one per-lane constant, no address arithmetic on the vector ALU (which the fp32
        // matrix instructions share)
        const bool fast = even_rows && (t0 + ncb * (rpu == 2 ? 32 : 64) <= g.La) && (u0 >= 0) && (u0 + ncb * (rpu == 2 ? 32 : 64) <= g.Lb);
        if (fast) {
            const wm_srd_t sa = make_srd(Ab, ((size_t)(g.NB - nb) * g.Ca * g.La) * sizeof(float));
            const wm_srd_t sb = make_srd(Bb, ((size_t)(g.NB - 1 - nb) * g.bcs + (size_t)g.Cb * g.Lb) * sizeof(float));
#pragma unroll
            for (int i = 0; i < GW_RA; ++i) {
                const int u = wv + 4 * i;
                if (u < a_units) {
                    const int rg = GEO1 ? u : (int)(((unsigned)u * inv_ncb) >> 16), cb = GEO1 ? 0 : u - rg * ncb;
                    ra[i] = buf_load(sa, vA * 4u, (unsigned)((a0 + rg * rpu) * g.La + t0 + cb * 64) * 4u);
                }
            }
#pragma unroll
            for (int i = 0; i < GW_RB; ++i) {
                const int u = wv + 4 * i;
                if (u < b_units) {
                    const int rg = GEO1 ? u : (int)(((unsigned)u * inv_ncb) >> 16), cb = GEO1 ? 0 : u - rg * ncb;
                    rb[i] = buf_load(sb, vB * 4u, (unsigned)((b0 + rg * rpu) * g.Lb + u0 + cb * 64) * 4u);
                }
            }
        } else {                                             // clip edges / odd row counts: clamp every address, zero the padding
#pragma unroll
            for (int i = 0; i < GW_RA; ++i) {
                const int u = wv + 4 * i;
                if (u < a_units) {
                    const int rg = GEO1 ? u : (int)(((unsigned)u * inv_ncb) >> 16), cb = GEO1 ? 0 : u - rg * ncb;
                    const int a = a0 + rg * rpu + row_l, col = cb * 64 + col_l, t = t0 + col;
                    const bool ok = (a < g.Ca) && (col < TC) && (t < g.La);
                    const float v = Ab[(unsigned)(min(a, g.Ca - 1) * g.La + min(t, g.La - 1))];
                    ra[i] = ok ? v : 0.f;
                }
            }
#pragma unroll
            for (int i = 0; i < GW_RB; ++i) {
                const int u = wv + 4 * i;
                if (u < b_units) {
                    const int rg = GEO1 ? u : (int)(((unsigned)u * inv_ncb) >> 16), cb = GEO1 ? 0 : u - rg * ncb;
                    const int b = b0 + rg * rpu + row_l, col = cb * 64 + col_l, uu = u0 + col;
                    const bool ok = (b < g.Cb) && (col < BW) && (uu >= 0) && (uu < g.Lb);
                    const float v = Bb[(unsigned)(min(b, g.Cb - 1) * g.Lb + min(max(uu, 0), g.Lb - 1))];
                    rb[i] = ok ? v : 0.f;
                }
            }
        }
    };
    auto store_work = [&]() {
        int wv = wave;
        asm volatile("" : "+s"(wv));
#pragma unroll
        for (int i = 0; i < GW_RA; ++i) {
            const int u = wv + 4 * i;
            if (u < a_units) {
                const int rg = GEO1 ? u : (int)(((unsigned)u * inv_ncb) >> 16), cb = GEO1 ? 0 : u - rg * ncb;
                (smem + rg * rpu * g.AP + cb * 64)[lA] = ra[i];
            }
        }
#pragma unroll
        for (int i = 0; i < GW_RB; ++i) {
            const int u = wv + 4 * i;
            if (u < b_units) {
                const int rg = GEO1 ? u : (int)(((unsigned)u * inv_ncb) >> 16), cb = GEO1 ? 0 : u - rg * ncb;
                (smem + rg * rpu * g.BP + cb * 64)[lB] = rb[i];
            }
        }
    };

    for (int i = tid; i < g.BP; i += 256) smem[zrow + i] = 0.f;
    const int z = blockIdx.y, nz = gridDim.y;
    const int tbeg = wt * (TC / g.WT), npair = TC / g.WT / 2;
    load_work(z);
    for (int w = z; w < g.nwork; w += nz) {
        __syncthreads();                                     // the previous chunk's operand reads are done
        store_work();
        __syncthreads();
        if (w + nz < g.nwork) load_work(w + nz);             // in flight while the matrix cores run this chunk
        // software-pipelined position-pair loop, two pairs per trip with two operand sets (each read one MFMA group before
        // its use); the read-ahead index is clamped (scalar) so it stays inside the row: AP, BP carry 2 positions of slack
        const float* ap = smem + (wa * 32 + l31) * g.AP + half + tbeg;
        const float* bp[NS];
        float av = ap[0], bv[NS], an, bn[NS];
#pragma unroll
        for (int jt = 0; jt < NS; ++jt) { bp[jt] = smem + boff[jt] + tbeg; bv[jt] = bp[jt][0]; }
        for (int p = 0; p < npair; p += 2) {
            an = ap[2 * p + 2];
#pragma unroll
            for (int jt = 0; jt < NS; ++jt) bn[jt] = bp[jt][2 * p + 2];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jt = 0; jt < NS; ++jt) acc[jt] = mfma32(av, bv[jt], acc[jt]);
            __builtin_amdgcn_sched_barrier(0);
            const int t2 = min(2 * p + 4, 2 * npair);
            av = ap[t2];
#pragma unroll
            for (int jt = 0; jt < NS; ++jt) bv[jt] = bp[jt][t2];
            __builtin_amdgcn_sched_barrier(0);
            if (p + 1 < npair) {
#pragma unroll
                for (int jt = 0; jt < NS; ++jt) acc[jt] = mfma32(an, bn[jt], acc[jt]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (do_bias) {
            const float* rp = smem + brow * g.AP + bpart * bcpp;
            for (int c = 0; c < bcpp; ++c) bsum += rp[c];
        }
    }
    __syncthreads();

    // ---- waves that split the positions: sum in wave order (fixed), then the wt == 0 wave holds the tile
    if (g.WT > 1) {
        float* red = smem;                                   // [(WT-1)][waves with wt==0][NS][16][64]
        const int owner = wa + WA * wj;                      // index among the wt == 0 waves
        const int nown = WA * g.WJW;
        if (wt > 0) {
#pragma unroll
            for (int jt = 0; jt < NS; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((((wt - 1) * nown + owner) * NS + jt) * 16 + r) * 64 + lane] = acc[jt][r];
        }
        __syncthreads();
        if (wt == 0) {
            for (int o = 0; o < g.WT - 1; ++o)
#pragma unroll
                for (int jt = 0; jt < NS; ++jt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[jt][r] += red[(((o * nown + owner) * NS + jt) * 16 + r) * 64 + lane];
        }
        __syncthreads();
    }
    if (wt == 0) {
#pragma unroll
        for (int jt = 0; jt < NS; ++jt) {
            if (jt >= g.nsub) continue;
            const int j = j0 + (wj * g.nsub + jt) * 32 + l31;
            if (j >= g.NJ) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int a = a0 + wa * 32 + mfma_row(r, half);
                if (a < g.Ca) g.slab[((size_t)z * g.Ca + a) * g.NJ + j] = acc[jt][r];
            }
        }
    }
    if (do_bias) {
        float* bred = smem;                                  // [bparts][TA]
        bred[bpart * TA + brow] = bsum;
        __syncthreads();
        if (bpart == 0 && a0 + brow < g.Ca) {
            float sacc = 0.f;
            for (int p = 0; p < bparts; ++p) sacc += bred[p * TA + brow];
            g.slabb[(size_t)z * g.Ca + a0 + brow] = sacc;
        }
    }
}

// out[perm(i)] (+)= sum_z slab[z][i]: 8 z-lanes per output, each a fixed-order chain, combined in fixed order (fp64).  Outputs
// i >= n are the bias gradient: bout[i - n] (+)= sum_z bslab[z][i - n] (nb of them; bslab / bout may be null with nb = 0).
//   remap 0: identity | 1: column j' = k*r1 + b -> b*r2 + k (tap planes -> dW[a][b][k]; r1 = Cb, r2 = K)
//        | 2: column j' = (co*r1 + ph)*2 + q -> co*2*r1 + q*r1 + ph (stride phases -> ConvTranspose taps; r1 = stride)
__global__ __launch_bounds__(256) void gwgrad2_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, size_t n, int nz,
                                                             int NJ, int accumulate, int remap, int r1, int r2,
                                                             const float* __restrict__ bslab, float* __restrict__ bout, int nb) {
    __shared__ double part[8][32];
    const int o = threadIdx.x & 31, zl = threadIdx.x >> 5;
    const size_t i = (size_t)blockIdx.x * 32 + o;
    const bool is_w = i < n, is_b = !is_w && i < n + (size_t)nb;
    double s = 0.0;
    if (is_w)
        for (int z = zl; z < nz; z += 8) s += (double)slab[(size_t)z * n + i];
    else if (is_b)
        for (int z = zl; z < nz; z += 8) s += (double)bslab[(size_t)z * nb + (i - n)];
    part[zl][o] = s;
    __syncthreads();
    if (zl == 0 && (is_w || is_b)) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += part[q][o];
        if (is_b) {
            float* d = bout + (i - n);
            *d = (float)((accumulate ? (double)*d : 0.0) + t);
            return;
        }
        size_t d = i;
        if (remap) {
            const size_t a = i / (size_t)NJ;
            const int jp = (int)(i - a * (size_t)NJ);
            int j;
            if (remap == 1) { const int k = jp / r1, b = jp - k * r1; j = b * r2 + k; }
            else { const int q2 = jp & 1, cp = jp >> 1, co = cp / r1, ph = cp - co * r1; j = co * 2 * r1 + q2 * r1 + ph; }
            d = a * (size_t)NJ + j;
        }
        out[d] = (float)((accumulate ? (double)out[d] : 0.0) + t);
    }
}

// y[nb][row][t] = x[nb][c][t*S + k - P] (0 outside the clip), t in [0, Lout), for c < C, k < K:
//   order 0: row = k*C + c (tap planes: a strided Conv1d's weight gradient becomes a K = 1 GEMM, and its 1x1 strided
//            skip convolution reads plane P);  order 1: row = c*K + k (stride phases of a ConvTranspose1d's output gradient)
__global__ __launch_bounds__(256) void gather_taps_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int Lin, int K,
                                                          int S, int P, int Lout, int order) {
    // one workgroup = 256 output positions of one input row: the span of x they touch is read ONCE, coalesced, into LDS;
    // every tap plane is then a strided LDS read and a coalesced store
    __shared__ float xs[255 * 8 + 16 + 8];
    const int t0 = blockIdx.x * 256, c = blockIdx.y, nb = blockIdx.z, tid = threadIdx.x;
    const float* xr = x + ((size_t)nb * C + c) * Lin;
    const int u0 = t0 * S - P, span = 255 * S + K;
    for (int i = tid; i < span; i += 256) {
        const int u = u0 + i;
        xs[i] = (u >= 0 && u < Lin) ? xr[u] : 0.f;
    }
    __syncthreads();
    const int t = t0 + tid;
    if (t >= Lout) return;
    float* yb = y + (size_t)nb * C * K * Lout;
    for (int k = 0; k < K; ++k) {
        const int row = order ? c * K + k : k * C + c;
        yb[(size_t)row * Lout + t] = xs[tid * S + k];
    }
}

template <int WA, int NS, bool GEO1>
int launch_gwgrad2(const GWArgs& g, dim3 grid, size_t lds, hipStream_t stream) {
    auto kern = gwgrad2_kernel<WA, NS, GEO1>;
    static wm::DevOnce once;
    if (!wm::dev_done(once)) {
        WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        wm::dev_mark(once);
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, g);
    WM_CHECK_LAUNCH();
    return 0;
}

struct GWPlan {
    bool ok;
    int WA, TC, nchunks, nsub, WJW, WT, ntj, nta, NBCH, ncb, AP, BP, gz;
    size_t lds, slab_floats;
};

GWPlan gw_plan(int NB, int Ca, int Cb, int La, int K) {
    GWPlan p{};
    p.WA = Ca > 32 ? 2 : 1;
    const int TA = 32 * p.WA, R = 4 / p.WA, NJ = Cb * K;
    p.nta = (Ca + TA - 1) / TA;
    const int nsub_total = (NJ + 31) / 32;
    // wave split of the columns (WJW of the R waves that do not split rows) x 32-column blocks per wave (ns <= 4): the tile must
    // not touch more Bx rows than the staging registers hold (K = 1: every column is its own row); among those, waste the fewest
    // matrix-core columns over all column tiles, then take the widest tile (fewest re-reads of A), then the most column waves
    int bw = 1, bn = 1, best_waste = 1 << 30, best_tj = 0;
    for (int wjw = R; wjw >= 1; wjw >>= 1) {
        for (int ns = 4; ns >= 1; --ns) {
            const int tjb = wjw * ns;                                    // blocks per column tile
            if (tjb > nsub_total && !(wjw == 1 && ns == 1) && (tjb - nsub_total) >= ns) continue;   // a whole wave would idle
            int rows = (tjb * 32 - 1) / K + 2;
            if (rows > Cb) rows = Cb;
            if (rows > 4 * GW_RB && !(wjw == 1 && ns == 1)) continue;
            const int waste = ((nsub_total + tjb - 1) / tjb) * tjb - nsub_total;
            if (waste < best_waste || (waste == best_waste && tjb > best_tj)) { best_waste = waste; best_tj = tjb; bw = wjw; bn = ns; }
        }
    }
    p.WJW = bw;
    p.nsub = bn;
    p.WT = R / p.WJW;
    {
        const int TJ0 = p.WJW * 32 * p.nsub;
        p.NBCH = (TJ0 - 1) / K + 2;
        if (p.NBCH > Cb) p.NBCH = Cb;
    }
    const int TJ = p.WJW * 32 * p.nsub;
    p.ntj = (NJ + TJ - 1) / TJ;
    const int ra = Ca < TA ? Ca : TA, rbx = p.NBCH;          // rows staged per chunk
    // positions per chunk: ncb blocks of 64 (one block of 32 for short sequences); few rows -> long chunks (up to 256)
    for (int ncb = 4; ncb >= 1 && !p.ok; --ncb) {
        for (int narrow = 0; narrow < 2 && !p.ok; ++narrow) {
            if (narrow && ncb > 1) continue;
            const int width = narrow ? 32 : 64 * ncb;            // staged span of a row
            int tcmax = (width - (K - 1)) & ~7;                  // so that BW = TC + K - 1 <= width
            if (tcmax < 8) continue;
            if (!narrow && ncb > 1 && La < 64 * (ncb - 1)) continue;
            if (!narrow && ncb == 1 && La <= 24) continue;       // short sequences: the 32-wide form wastes fewer lanes
            p.nchunks = (La + tcmax - 1) / tcmax;
            p.TC = (((La + p.nchunks - 1) / p.nchunks) + 7) & ~7;
            if (p.TC > tcmax) p.TC = tcmax;
            p.nchunks = (La + p.TC - 1) / p.TC;
            const int BW = p.TC + K - 1, rpu = BW <= 32 ? 2 : 1;
            if ((rpu == 2) != (narrow == 1)) continue;
            const int a_units = ((ra + rpu - 1) / rpu) * ncb, b_units = ((rbx + rpu - 1) / rpu) * ncb;
            // pitches: whole staging units wide (unconditional stores) + slack for the pipelined loop's read-ahead (2 positions);
            // A: odd (operand read down a column); Bx: == K (mod 32), so tile column j reads bank j
            p.AP = (width + 2) | 1;
            const int bpw = width + 2;
            p.BP = bpw + (((K - bpw) % 32) + 32) % 32;
            size_t lds = ((size_t)TA * p.AP + (size_t)(p.NBCH + 1) * p.BP) * sizeof(float);
            const size_t red = (size_t)(p.WT - 1) * p.WA * p.WJW * 4 * 16 * 64 * sizeof(float);
            if (red > lds) lds = red;
            if (lds < 1024 * sizeof(float)) lds = 1024 * sizeof(float);
            if (a_units <= 4 * GW_RA && b_units <= 4 * GW_RB && lds <= 52 * 1024) { p.lds = lds; p.ncb = ncb; p.ok = true; }   // three workgroups per CU
        }
    }
    if (!p.ok) return p;
    const long long nwork = (long long)NB * p.nchunks;
    const int tiles = p.nta * p.ntj;
    long long gz = tiles >= 768 ? 1 : 768 / tiles;         // <= 768 workgroups = three per CU (registers and LDS admit three): one round
    if (gz > nwork) gz = nwork;
    if (gz < 1) gz = 1;
    p.gz = (int)gz;
    p.slab_floats = (size_t)p.gz * Ca * (NJ + 1);           // [gz][Ca][NJ] then [gz][Ca] bias partials
    return p;
}

// dz = g * elu'(y) with y = ELU(z):  elu'(z) = 1 for y > 0 else y + 1   (alpha = 1)
__global__ void elu_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y, float* __restrict__ dz, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float yy = y[i];
    dz[i] = g[i] * (yy > 0.f ? 1.f : yy + 1.f);
}

// partial[c][y] = sum over clips nb == y (mod gridDim.y) and all t of x[nb][c][t]; channel_sum_final adds the partials of a
// channel in fixed order (bias gradient of the transposed convolutions; no atomics: bit-reproducible)
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ x, float* __restrict__ partial, int NB, int C, int L) {
    __shared__ float scratch[4];
    const int c = blockIdx.x;
    float s = 0.f;
    for (int nb = blockIdx.y; nb < NB; nb += gridDim.y) {
        const float* r = x + ((size_t)nb * C + c) * L;
        for (int t = threadIdx.x; t < L; t += 256) s += r[t];
    }
    s = block_sum<4>(s, scratch);
    if (threadIdx.x == 0) partial[(size_t)c * gridDim.y + blockIdx.y] = s;
}
__global__ void channel_sum_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int C, int ny, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int y = 0; y < ny; ++y) s += (double)partial[(size_t)c * ny + y];
    out[c] = (float)((accumulate ? (double)out[c] : 0.0) + s);
}

// out[row] = sum_t x[row][t] for any row length
__global__ __launch_bounds__(256) void rowsum_any_kernel(const float* __restrict__ x, float* __restrict__ out, int L) {
    __shared__ float scratch[4];
    const float* r = x + (size_t)blockIdx.x * L;
    float s = 0.f;
    for (int t = threadIdx.x; t < L; t += 256) s += r[t];
    s = block_sum<4>(s, scratch);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// dtable[idx[b]][:] += dvec[b][:]   (dense embedding gradient for any embedding width).  One thread per column walks the
// batch in order, so duplicate ids add in a fixed order (bit-reproducible; B is ~128, the cost is nil)
__global__ void rows_scatter_add_kernel(float* __restrict__ dtable, const long long* __restrict__ idx,
                                        const float* __restrict__ dvec, int Bn, int dim, int nrows) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= dim) return;
    for (int b = 0; b < Bn; ++b) {
        const long long m = idx[b];
        if (m < 0 || m >= nrows) continue;
        dtable[(size_t)m * dim + d] += dvec[(size_t)b * dim + d];
    }
}

}  // namespace

namespace {

// ---------------------------------------------------------------------------------------------------------------
// nn.LSTM(hd, hd, num_layers=2) over the T = 50 latent steps (py/main14b_2.py:137, :165): the recurrence of ONE layer as
// a chain of per-step launches issued back to back by the C launcher (no host work between them; the kernel boundary is
// the step barrier, so no grid-wide spin and nothing that can hang).  Layouts: time-major, batch contiguous.
//
// forward step:  gates[q*H + u][b] = xp[q*H + u][b] + sum_k W_hh[q*H + u][k] * h_prev[k][b];  cell update.
//   workgroup = 8 units x 4 gates (32 gate rows) x 32 batch columns; its 4 waves split the contraction (K = H): lane
//   (i, half) of wave w owns k = w*H/4 + half*H/8 + s, s < H/8 -- a contiguous run of ITS weight row, so the A operand is
//   H/32 float4 loads straight from the PyTorch weight (no transpose, no LDS); the B operand rows are 128-byte coalesced.
//   All loads are issued up front, then H/8 MFMAs per wave; partial tiles are summed through LDS in wave order.
__global__ __launch_bounds__(256) void lstm_seq_fwd_kernel(const float* xp, const float* __restrict__ whh,
                                                           const float* __restrict__ hprev, const float* __restrict__ cprev,
                                                           float* __restrict__ hout, float* __restrict__ cout, float* gates_out,
                                                           int H, int Bn) {
    constexpr int MAXS = 32;                     // H/8 <= 32  (H <= 256)
    __shared__ float red[4][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int u0 = blockIdx.x * 8, b0 = blockIdx.y * 32;
    const int SL = H >> 3;                       // contraction values per lane
    const int grow = (l31 >> 3) * H + u0 + (l31 & 7);                  // this lane's gate row (A operand row)
    const int kbase = wave * (H >> 2) + half * SL;
    const int bcol = min(b0 + l31, Bn - 1);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (hprev) {
        f32x4 a4[MAXS / 4];
        float bv[MAXS];
        const float* wr = whh + (size_t)grow * H + kbase;
#pragma unroll
        for (int i = 0; i < MAXS / 4; ++i)
            if (4 * i < SL) a4[i] = *reinterpret_cast<const f32x4*>(wr + 4 * i);
#pragma unroll
        for (int s = 0; s < MAXS; ++s)
            if (s < SL) bv[s] = hprev[(size_t)(kbase + s) * Bn + bcol];
#pragma unroll
        for (int s = 0; s < MAXS; ++s)
            if (s < SL) acc = mfma32(a4[s >> 2][s & 3], bv[s], acc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    // cell update: thread = (unit, batch column); gate q of unit u is tile row q*8 + u = register (row&3) + 4*((row>>3)&3)... of lane half
    const int u = tid >> 5, bl = tid & 31, b = b0 + bl;
    if (b >= Bn) return;
    float pre[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = q * 8 + u;               // D row = (r&3) + 8*(r>>2) + 4*half
        const int hf = (row >> 2) & 1, r = (row & 3) + 4 * (row >> 3);
        const int ln = hf * 32 + bl;
        pre[q] = ((red[0][r][ln] + red[1][r][ln]) + red[2][r][ln]) + red[3][r][ln] + xp[(size_t)(q * H + u0 + u) * Bn + b];
    }
    const float gi = 1.f / (1.f + expf(-pre[0])), gf = 1.f / (1.f + expf(-pre[1])), gg = tanhf(pre[2]), go = 1.f / (1.f + expf(-pre[3]));
    const size_t o = (size_t)(u0 + u) * Bn + b;
    const float cp = cprev ? cprev[o] : 0.f;
    const float c = gf * cp + gi * gg;
    hout[o] = go * tanhf(c);
    cout[o] = c;
    if (gates_out) {
        gates_out[(size_t)(0 * H + u0 + u) * Bn + b] = gi; gates_out[(size_t)(1 * H + u0 + u) * Bn + b] = gf;
        gates_out[(size_t)(2 * H + u0 + u) * Bn + b] = gg; gates_out[(size_t)(3 * H + u0 + u) * Bn + b] = go;
    }
}

// backward step:  dh[u][b] = dout[u][b] + sum_g W_hh[g][u] * da_next[g][b]  (K = 4H), then the pointwise gate backward:
//   gates (activations) -> da (in place), dc (in/out).  Workgroup = 16 units x 16 batch columns (fp32 MFMA 16x16x4), its 4
//   waves split K; lane (i, kq) of wave w owns g = w*H + kq*H/4 + s, s < H/4 -- contiguous in the TRANSPOSED weight whhT
//   [H][4H] (float4 loads); the B operand rows are 64-byte segments of da_next.
__global__ __launch_bounds__(256) void lstm_seq_bwd_kernel(float* gates, const float* __restrict__ danext, const float* __restrict__ whhT,
                                                           const float* __restrict__ c, const float* __restrict__ cprev,
                                                           const float* __restrict__ dout, float* __restrict__ dc, int H, int Bn) {
    constexpr int MAXS = 64;                     // H/4 <= 64
    __shared__ float red[4][4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, l15 = lane & 15;
    const int u0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    const int SL = H >> 2;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (danext) {
        const int gbase = wave * H + kq * SL;
        const int bcol = min(b0 + l15, Bn - 1);
        const float* wr = whhT + (size_t)(u0 + l15) * 4 * H + gbase;
        f32x4 a4[MAXS / 4];
        float bv[MAXS];
#pragma unroll
        for (int i = 0; i < MAXS / 4; ++i)
            if (4 * i < SL) a4[i] = *reinterpret_cast<const f32x4*>(wr + 4 * i);
#pragma unroll
        for (int s = 0; s < MAXS; ++s)
            if (s < SL) bv[s] = danext[(size_t)(gbase + s) * Bn + bcol];
#pragma unroll
        for (int s = 0; s < MAXS; ++s)
            if (s < SL) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s >> 2][s & 3], bv[s], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    const int ul = tid >> 4, bl = tid & 15, b = b0 + bl;       // D: col = lane&15, row = (lane>>4)*4 + reg
    if (b >= Bn) return;
    const int r = ul & 3, ln = (ul >> 2) * 16 + bl;
    const size_t o = (size_t)(u0 + ul) * Bn + b, hb = (size_t)H * Bn;
    const float dht = ((red[0][r][ln] + red[1][r][ln]) + red[2][r][ln]) + red[3][r][ln] + dout[o];
    const float gi = gates[o], gf = gates[hb + o], gg = gates[2 * hb + o], go = gates[3 * hb + o];
    const float tc = tanhf(c[o]);
    const float dct = dc[o] + dht * go * (1.f - tc * tc);
    const float cp = cprev ? cprev[o] : 0.f;
    gates[o] = dct * gg * gi * (1.f - gi);
    gates[hb + o] = dct * cp * gf * (1.f - gf);
    gates[2 * hb + o] = dct * gi * (1.f - gg * gg);
    gates[3 * hb + o] = dht * tc * go * (1.f - go);
    dc[o] = dct * gf;
}

}  // namespace

extern "C" {

int wm_gscale_absmax(const float* x, long long n, float* scratch, float log2_target, float* gscale, hipStream_t stream);   // bn.hip

// y[nb][co][t'] = act( bias[co] + vec[nb][co] + res + sum_{ci,k} wp[ci*K+k][m] * x[nb][ci][n*S + k - P] )
//   st == 1: m = co, t' = n (Cout == Mtot, Lout == Nout);  st > 1: m = co*st + phase, t' = n*st + phase - shp.
int wm_gconv(const float* x, const float* wp, const float* bias, const float* vec, const float* res, float* y, int NB,
             int Cin, int Lin, int K, int S, int P, int Mtot, int Nout, int st, int shp, int Cout, int Lout, int act,
             const float* x2, int Cin1, int nph, hipStream_t stream) {
    if (NB <= 0 || Cin <= 0 || K <= 0 || K > 16 || S <= 0 || S > 8 || Mtot <= 0 || Nout <= 0 || st < 1 || st > 8 || Lin <= 0 ||
        Lin >= (1 << 24) || NB > 65535 || Mtot >= 8192)
        return (int)hipErrorInvalidValue;
    if (nph <= 0) nph = st;
    if (nph > st || (x2 && (Cin1 <= 0 || Cin1 >= Cin))) return (int)hipErrorInvalidValue;
    GConvArgs a{x, wp, bias, vec, res, y, NB, Cin, Lin, K, S, P, Mtot, Nout, st, shp, Cout, Lout, act, 0, 0, x2, x2 ? Cin1 : Cin, nph};
    const bool vec4 = (Mtot % 4 == 0) && ((reinterpret_cast<uintptr_t>(wp) & 15) == 0);
    // tile shape by problem shape: (rows, columns) = (128,128) | (64,128) | (32,256); short sequences (<= 64): (128,64) | (64,64)
    if (!vec4) {
        if (Mtot > 32) return (int)hipErrorInvalidValue;     // unaligned weight rows: only the narrow heads (1 / 17 channels)
        return launch_gconv2<1, 1, 2, false>(a, stream);
    }
    int rc;
    if (Nout <= 64) rc = Mtot > 64 ? launch_gconv2<2, 2, 1, true>(a, stream) : launch_gconv2<2, 1, 1, true>(a, stream);
    else if (Mtot > 64) rc = launch_gconv2<2, 2, 2, true>(a, stream);
    else if (Mtot > 32) rc = launch_gconv2<2, 1, 2, true>(a, stream);
    else rc = launch_gconv2<1, 1, 2, true>(a, stream);
    if (rc == (int)hipErrorInvalidValue) rc = launch_gconv2<2, 1, 1, true>(a, stream);   // widest-stride shapes: the narrowest input tile
    return rc;
}

// wm_gconv on the f16 two-piece split: wph = wm_gconv_pack_h's image of the SAME wp; Cin % 16 == 0 (and Cin1 % 16 == 0 with two sources);
// gscale = {gs, 1 / gs} for a gradient input (wm_gscale_absmax), NULL for activations.  hipErrorInvalidValue when the shape is outside
// the kernel's window limits (the caller then uses wm_gconv).
int wm_gconv_h(const float* x, const void* wph, const float* bias, const float* vec, const float* res, float* y, int NB,
               int Cin, int Lin, int K, int S, int P, int Mtot, int Nout, int st, int shp, int Cout, int Lout, int act,
               const float* x2, int Cin1, int nph, const float* gscale, float* ymax, hipStream_t stream) {
    if (NB <= 0 || Cin <= 0 || (Cin & 15) || K <= 0 || K > 16 || S <= 0 || S > 8 || Mtot <= 0 || Nout <= 0 || st < 1 || st > 8 || Lin <= 0 ||
        Lin >= (1 << 24) || NB > 65535 || Mtot >= 8192 || !wph)
        return (int)hipErrorInvalidValue;
    if (nph <= 0) nph = st;
    if (nph > st || (x2 && (Cin1 <= 0 || Cin1 >= Cin || (Cin1 & 15)))) return (int)hipErrorInvalidValue;
    GConvHArgs ha{GConvArgs{x, nullptr, bias, vec, res, y, NB, Cin, Lin, K, S, P, Mtot, Nout, st, shp, Cout, Lout, act, 0, 0, x2,
                            x2 ? Cin1 : Cin, nph},
                  reinterpret_cast<const unsigned short*>(wph), gscale, ymax};
    int rc;
    if (Nout <= 64) rc = Mtot > 64 ? launch_gconvh<2, 2, 1>(ha, stream) : launch_gconvh<2, 1, 1>(ha, stream);
    else if (Mtot > 64) rc = launch_gconvh<2, 2, 2>(ha, stream);
    else if (Mtot > 32) rc = launch_gconvh<2, 1, 2>(ha, stream);
    else rc = launch_gconvh<1, 1, 2>(ha, stream);
    if (rc == (int)hipErrorInvalidValue && Nout > 64) rc = Mtot > 64 ? launch_gconvh<2, 2, 1>(ha, stream) : launch_gconvh<2, 1, 1>(ha, stream);
    return rc;
}

// f16 image of wp [Cin * K][Mtot] for wm_gconv_h: wph holds 2 * Cin * K * Mtot f16 + 2 floats; scratch >= 1024 floats
int wm_gconv_pack_h(const float* wp, void* wph, float* scratch, int Cin, int K, int Mtot, hipStream_t stream) {
    if (!wp || !wph || Cin <= 0 || (Cin & 15) || K <= 0 || Mtot <= 0) return (int)hipErrorInvalidValue;
    const long long n = (long long)Cin * K * Mtot;
    unsigned short* out = reinterpret_cast<unsigned short*>(wph);
    float* tail = reinterpret_cast<float*>(out + 2 * n);
    // scratch == NULL: the fixed scale 2^8 (one launch).  Weights of magnitude 4e-6 ... 250 keep a normal hi piece and lose nothing that an
    // fp32 product would keep: the scale only has to hold the pieces inside f16's 30 binades; larger weights saturate at +-6e4 / 2^8
    if (scratch) {
        int rc = wm_gscale_absmax(wp, n, scratch, 10.0f, tail, stream);   // max |w| ws in (2^9, 2^10]
        if (rc) return rc;
    }
    const int grid = (int)((n + 255) / 256 < 512 ? (n + 255) / 256 : 512);
    hipLaunchKernelGGL(gconv_pack_h_kernel, dim3(grid), dim3(256), 0, stream, wp, out, Cin, K, Mtot, scratch ? 0.f : 256.f, -1);
    WM_CHECK_LAUNCH();
    return 0;
}

// the same image straight from a Conv1d weight w [Cout][Cin][K] (fixed scale 2^8): mode 0 = the forward matrix (GEMM channels Cin, rows
// Cout), mode 1 = the stride-1 data-gradient matrix (GEMM channels Cout, rows Cin, taps flipped).  wph: 2 * Cout * Cin * K f16 + 2 floats
int wm_gconv_pack_h_conv(const float* w, void* wph, int Cout, int Cin, int K, int mode, hipStream_t stream) {
    const int gch = mode == 0 ? Cin : Cout, rows = mode == 0 ? Cout : Cin;
    if (!w || !wph || (mode != 0 && mode != 1) || Cout <= 0 || Cin <= 0 || K <= 0 || (gch & 15)) return (int)hipErrorInvalidValue;
    const long long n = (long long)Cout * Cin * K;
    const int grid = (int)((n + 255) / 256 < 512 ? (n + 255) / 256 : 512);
    hipLaunchKernelGGL(gconv_pack_h_kernel, dim3(grid), dim3(256), 0, stream, w, reinterpret_cast<unsigned short*>(wph), gch, K, rows, 256.f, mode);
    WM_CHECK_LAUNCH();
    return 0;
}

int wm_permute_acl(const float* x, float* y, int A, int C, int L, hipStream_t stream) {
    const size_t n = (size_t)A * C * L;
    hipLaunchKernelGGL(permute_acl_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, y, A, C, L);
    WM_CHECK_LAUNCH();
    return 0;
}

// workspace of wm_gwgrad for a problem shape: *slab_floats = number of fp32 elements the caller must provide
int wm_gwgrad_plan(int NB, int Ca, int Cb, int La, int K, long long* slab_floats, hipStream_t) {
    if (NB <= 0 || Ca <= 0 || Cb <= 0 || La <= 0 || K <= 0 || K > 16 || !slab_floats) return (int)hipErrorInvalidValue;
    const GWPlan p = gw_plan(NB, Ca, Cb, La, K);
    if (!p.ok) return (int)hipErrorInvalidValue;
    *slab_floats = (long long)p.slab_floats;
    return 0;
}

// G[a][b][k] (+)= sum_{nb,t} A[nb][a][t] * Bx[nb][b][t + k - P]; dbias[a] (+)= sum A (dbias may be NULL).  Deterministic
// (split-K partial tiles in `slab`, >= wm_gwgrad_plan floats, then a fixed-order reduce).  accumulate: 0 overwrite | 1 add.
// b_clip_stride: floats between clips of Bx (0 = dense, Cb*Lb); remap / r1 / r2: column order of the result (see the reduce).
int wm_gwgrad(const float* A, const float* Bx, float* G, float* dbias, float* slab, int NB, int Ca, int Cb, int La, int Lb,
              int K, int P, long long b_clip_stride, int remap, int r1, int r2, int accumulate, hipStream_t stream) {
    if (NB <= 0 || Ca <= 0 || Cb <= 0 || La <= 0 || Lb <= 0 || K <= 0 || K > 16 || !slab || remap < 0 || remap > 2 ||
        (size_t)Ca * La >= (1u << 30) || (size_t)Cb * Lb >= (1u << 30))
        return (int)hipErrorInvalidValue;
    if (b_clip_stride == 0) b_clip_stride = (long long)Cb * Lb;
    if (b_clip_stride < (long long)Cb * Lb) return (int)hipErrorInvalidValue;
    const int NJ = Cb * K;
    if ((remap == 1 && (r1 <= 0 || r2 <= 0 || r1 * r2 != NJ)) || (remap == 2 && (r1 <= 0 || K != 2 || Cb % r1 != 0)))
        return (int)hipErrorInvalidValue;
    const GWPlan p = gw_plan(NB, Ca, Cb, La, K);
    if (!p.ok) return (int)hipErrorInvalidValue;
    float* slabb = dbias ? slab + (size_t)p.gz * Ca * NJ : nullptr;
    GWArgs g{A, Bx, slab, slabb, NB, Ca, Cb, La, Lb, K, P, b_clip_stride, NJ, p.TC, p.nchunks, p.nsub, p.WJW, p.WT, p.ntj, p.NBCH,
             p.ncb, p.AP, p.BP, NB * p.nchunks};
    dim3 grid(p.nta * p.ntj, p.gz);
    const int ns = p.nsub == 1 ? 1 : (p.nsub == 2 ? 2 : 4);
    int rc = 0;
    const bool geo1 = (p.ncb == 1) && (p.TC + K - 1 > 32);
#define WM_GW(WA_, NS_) (geo1 ? launch_gwgrad2<WA_, NS_, true>(g, grid, p.lds, stream) : launch_gwgrad2<WA_, NS_, false>(g, grid, p.lds, stream))
    if (p.WA == 1) rc = ns == 1 ? WM_GW(1, 1) : ns == 2 ? WM_GW(1, 2) : WM_GW(1, 4);
    else rc = ns == 1 ? WM_GW(2, 1) : ns == 2 ? WM_GW(2, 2) : WM_GW(2, 4);
#undef WM_GW
    if (rc) return rc;
    const size_t n = (size_t)Ca * NJ;
    const int nbias = dbias ? Ca : 0;
    hipLaunchKernelGGL(gwgrad2_reduce_kernel, dim3((unsigned)((n + nbias + 31) / 32)), dim3(256), 0, stream, slab, G, n, p.gz, NJ, accumulate,
                       remap, r1, r2, slabb, dbias, nbias);
    WM_CHECK_LAUNCH();
    return 0;
}

// y[nb][row][t] = x[nb][c][t*S + k - P] (0 outside), t < Lout; order 0: row = k*C + c | 1: row = c*K + k   (see gather_taps_kernel)
int wm_gather_taps(const float* x, float* y, int NB, int C, int Lin, int K, int S, int P, int Lout, int order, hipStream_t stream) {
    if (NB <= 0 || NB > 65535 || C <= 0 || C > 65535 || Lin <= 0 || K <= 0 || K > 16 || S <= 0 || S > 8 || Lout <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(gather_taps_kernel, dim3((Lout + 255) / 256, C, NB), dim3(256), 0, stream, x, y, C, Lin, K, S, P, Lout, order);
    WM_CHECK_LAUNCH();
    return 0;
}

int wm_elu_bwd(const float* g, const float* y, float* dz, long long n, hipStream_t stream) {
    hipLaunchKernelGGL(elu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, g, y, dz, (size_t)n);
    WM_CHECK_LAUNCH();
    return 0;
}

// out[c] (+)= sum_{nb,t} x[nb][c][t]; partial: >= 64*C floats of scratch.  Fixed summation order (no atomics).
int wm_channel_sum(const float* x, float* out, float* partial, int NB, int C, int L, int accumulate, hipStream_t stream) {
    if (NB <= 0 || C <= 0 || L <= 0 || !partial) return (int)hipErrorInvalidValue;
    const int ny = NB < 64 ? NB : 64;
    hipLaunchKernelGGL(channel_sum_kernel, dim3(C, ny), dim3(256), 0, stream, x, partial, NB, C, L);
    WM_CHECK_LAUNCH();
    hipLaunchKernelGGL(channel_sum_final_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, partial, out, C, ny, accumulate);
    WM_CHECK_LAUNCH();
    return 0;
}

int wm_rowsum_any(const float* x, float* out, int rows, int L, hipStream_t stream) {
    hipLaunchKernelGGL(rowsum_any_kernel, dim3(rows), dim3(256), 0, stream, x, out, L);
    WM_CHECK_LAUNCH();
    return 0;
}

int wm_rows_scatter_add(float* dtable, const long long* idx, const float* dvec, int Bn, int dim, int nrows, hipStream_t stream) {
    hipLaunchKernelGGL(rows_scatter_add_kernel, dim3((dim + 255) / 256), dim3(256), 0, stream, dtable, idx, dvec, Bn, dim, nrows);
    WM_CHECK_LAUNCH();
    return 0;
}

// One layer of nn.LSTM(H, H) over T steps, zero initial state (py/main14b_2.py:165).  xp [T][4H][B] = W_ih x_t + b_ih + b_hh
// (overwritten in place by the gate activations when save != 0 -- what wm_lstm_seq_bwd consumes); whh = weight_hh [4H][H];
// hs, cs [T+1][H][B] with hs[0] / cs[0] = 0 provided by the caller: h_t = hs[t+1].  T launches, issued back to back.
int wm_lstm_seq_fwd(float* xp, const float* whh, float* hs, float* cs, int T, int H, int B, int save, hipStream_t stream) {
    if (T <= 0 || H <= 0 || (H & 31) || H > 256 || B <= 0) return (int)hipErrorInvalidValue;
    const size_t hb = (size_t)H * B;
    dim3 grid(H / 8, (B + 31) / 32);
    for (int t = 0; t < T; ++t) {
        float* g = xp + (size_t)t * 4 * hb;
        hipLaunchKernelGGL(lstm_seq_fwd_kernel, grid, dim3(256), 0, stream, g, whh, t ? hs + (size_t)t * hb : nullptr,
                           t ? cs + (size_t)t * hb : nullptr, hs + (size_t)(t + 1) * hb, cs + (size_t)(t + 1) * hb, save ? g : nullptr, H, B);
    }
    WM_CHECK_LAUNCH();
    return 0;
}

// BPTT of that layer: gates [T][4H][B] (activations in, pre-activation gradients da out, in place), cs as written by the
// forward, dout [T][H][B] = dL/dh_t from above, whhT = weight_hh^T [H][4H], dc [H][B] scratch (zeroed here).  T launches.
int wm_lstm_seq_bwd(float* gates, const float* cs, const float* dout, const float* whhT, float* dc, int T, int H, int B,
                    hipStream_t stream) {
    if (T <= 0 || H <= 0 || (H & 31) || H > 256 || B <= 0) return (int)hipErrorInvalidValue;
    const size_t hb = (size_t)H * B;
    WM_TRY(hipMemsetAsync(dc, 0, hb * sizeof(float), stream));
    dim3 grid(H / 16, (B + 15) / 16);
    for (int t = T - 1; t >= 0; --t) {
        hipLaunchKernelGGL(lstm_seq_bwd_kernel, grid, dim3(256), 0, stream, gates + (size_t)t * 4 * hb,
                           t + 1 < T ? gates + (size_t)(t + 1) * 4 * hb : nullptr, whhT, cs + (size_t)(t + 1) * hb,
                           t ? cs + (size_t)t * hb : nullptr, dout + (size_t)t * hb, dc, H, B);
    }
    WM_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
