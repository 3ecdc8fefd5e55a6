// nn.LSTM(64, 64, batch_first=True), one layer, zero initial state, gate order i,f,g,o
// (py/main16.py:138,152-154) over T = 16000 dependent steps per clip.
//
// Split the MI355X way:
//   1. wm_lstm_xproj     xp[b,t,:] = W_ih x[b,:,t] + b_ih + b_hh for every t at once -- a GEMM with
//                        time as M, on the fp32 matrix cores, reading the channel-first encoder
//                        frame directly (no permute pass).
//   2. wm_lstm_fwd       the recurrence: ONE 256-thread workgroup per clip, persistent over all T
//                        steps.  W_hh (256x64 fp32) lives in registers (one gate row per lane), h is
//                        broadcast through a double-buffered 256-B LDS line, the four gates of a
//                        hidden unit sit in one wave (lane = gate*16 + unit) and are exchanged with
//                        bpermutes, so there is exactly one s_barrier per time step.  H = 64 makes
//                        the recurrent product a 256x64 GEMV per clip: an LDS-broadcast dot product,
//                        not an MFMA shape.  h is written channel-first [B,64,T] for the decoder.
//   3. wm_lstm_bwd       BPTT with the same geometry (dh = W_hh^T da as per-wave partial GEMVs, one
//                        barrier per step); it overwrites the saved gate activations with the
//                        pre-activation gradients da[b,t,256].
//   4. wm_lstm_dx / wm_lstm_wgrad   everything that is NOT sequential leaves the recurrence:
//                        dx = da W_ih, dW_ih = da^T x, dW_hh = da^T h_{t-1}, db = sum da are MFMA GEMMs.
#include "wm_common.hpp"
#include <type_traits>
using namespace wm;

namespace {

// -DWM_STAMP (diagnostic build, tests/diag_stamp_lstm.py): s_memtime stamps inside a recurrence step, summed over the steps of a
// clip per segment and written to a buffer of their own.  LSTAMP(var, val) stamps once `val` exists (and holds later uses of it back).
#ifdef WM_STAMP
__device__ unsigned long long* g_lstm_stamp = nullptr;
#define LSTAMP(var, val) unsigned long long var; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var), "+v"(val))
#define LSTAMP_ACC(i, a, b) tm[i] += (b) - (a)
#else
#define LSTAMP(var, val)
#define LSTAMP_ACC(i, a, b)
#endif

// Per-step tensors (xp / gates / da) are stored [B][T][256] with the column order  n' = unit*4 + gate  so that the
// four gates of a hidden unit sit in four adjacent lanes (one DPP quad) of the recurrence kernels and every per-step
// access of a wave is one contiguous 256-B segment.  gate row of the PyTorch parameters: n = gate*64 + unit.
__device__ __forceinline__ int gate_row(int np) { return (np & 3) * 64 + (np >> 2); }

typedef float v2f __attribute__((ext_vector_type(2)));
// broadcast lane k of every quad (DPP quad_perm: no LDS round trip, unlike ds_bpermute)
template <int K>
__device__ __forceinline__ float quad_bcast(float v) {
    constexpr int ctrl = K | (K << 2) | (K << 4) | (K << 6);
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, false));
}

// ------------------------------------------------------------------------------------ xproj
// block = (clip, 128-step tile); wave w owns gate columns [64w, 64w+64); D rows = time, cols = gate.
__global__ __launch_bounds__(256) void lstm_xproj_kernel(const float* __restrict__ x, const float* __restrict__ w_ih,
                                                         const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                                         float* __restrict__ xp, int T) {
    constexpr int NT = 128, XS = NT + 4;
    __shared__ __align__(16) float xs[64 * XS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int tilesPerClip = (T + NT - 1) / NT;
    const int b = blockIdx.x / tilesPerClip, t0 = (blockIdx.x % tilesPerClip) * NT;
    const float* xb = x + (size_t)b * 64 * T;
    for (int i = tid; i < 64 * (NT / 4); i += 256) {
        const int c = i / (NT / 4), q = i % (NT / 4), t = t0 + 4 * q;
        // unconditional load from a clamped address (rows past T are never stored): no branch, no vmcnt drain
        *reinterpret_cast<float4*>(xs + c * XS + 4 * q) = *reinterpret_cast<const float4*>(xb + (size_t)c * T + min(t, T - 4));
    }
    // B operand (W_ih^T) for this wave's two 32-column tiles: B[k = c][j = n] = w_ih[n][c]
    float breg[2][32];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int s = 0; s < 32; ++s) breg[nt][s] = w_ih[gate_row(wave * 64 + nt * 32 + l31) * 64 + 2 * s + half];
    float bias[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = gate_row(wave * 64 + nt * 32 + l31);
        bias[nt] = b_ih[n] + b_hh[n];
    }
    __syncthreads();
#pragma unroll 1
    for (int mt = 0; mt < 4; ++mt) {
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        const float* ap = xs + half * XS + mt * 32 + l31;     // A[i = t][k = c] = xs[c][t]
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const float a = ap[2 * s * XS];
            acc0 = mfma32(a, breg[0][s], acc0);
            acc1 = mfma32(a, breg[1][s], acc1);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int t = t0 + mt * 32 + mfma_row(r, half);
            if (t < T) {
                float* o = xp + ((size_t)b * T + t) * 256 + wave * 64 + l31;
                o[0] = acc0[r] + bias[0];
                o[32] = acc1[r] + bias[1];
            }
        }
    }
}

// ------------------------------------------------------------------------------- recurrence fwd
// sigmoid via v_exp_f32 + v_rcp_f32.  The absolute error of sigma is sigma*(1-sigma)*|z|*O(1e-7)
// <= 3e-8 for every z (the derivative vanishes where |z| is large), well inside fp32 round-off.
__device__ __forceinline__ float sigm(float z) { return __builtin_amdgcn_rcpf(1.0f + __expf(-z)); }
__device__ __forceinline__ float tanh_s(float z) { return fmaf(2.0f, sigm(2.0f * z), -1.0f); }
// one transcendental pair per lane whatever the gate: tanh(x) = 2*sigmoid(2x) - 1 for the g lanes
__device__ __forceinline__ float gate_act(float x, bool is_g) {
    const float s = sigm(is_g ? 2.0f * x : x);
    return is_g ? fmaf(2.0f, s, -1.0f) : s;
}

// acc += w * {v[lane k], v[lane k+1]}: the two broadcast values travel through an SGPR pair (v_readlane), which
// v_pk_fma_f32 takes directly as an operand -- no LDS return traffic, no VGPR copies
template <typename T2>
__device__ __forceinline__ T2 pk_fma_lanes(T2 w, float v, int k, T2 acc) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane(__float_as_int(v), k);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane(__float_as_int(v), k + 1);
    const unsigned long long pr = ((unsigned long long)hi << 32) | lo;
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "s"(pr));
    return acc;
}

// The same products with the lane reads issued AHEAD of the products that consume them: a v_readlane writes its SGPR late, and a
// product issued right behind the two reads of its own operand pair stalls on that write (the stamped step showed 610 cycles
// for 32 reads + 32 products).  N values (lanes k0 .. k0 + N - 1 of v) go into N scalar registers first, behind a scheduling
// fence; the products then find their operands complete.
template <int N>
__device__ __forceinline__ void lanes_to_sgprs(float v, int k0, unsigned long long (&pr)[N / 2]) {
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane(__float_as_int(v), k0 + 2 * i);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane(__float_as_int(v), k0 + 2 * i + 1);
        pr[i] = ((unsigned long long)hi << 32) | lo;
    }
    __builtin_amdgcn_sched_barrier(0);
}
template <typename T2>
__device__ __forceinline__ T2 pk_fma_s(T2 w, unsigned long long pr, T2 acc) {
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "s"(pr));
    return acc;
}

// ---- the recurrent 64-term dot product without lane reads: acc += v[lane i of my 16-lane row] * w  (DPP row_newbcast)
// Four registers hold the 64 broadcast values row-replicated (H[j][lane] = value[16 j + lane % 16]); product (j, i) is ONE
// v_fmac_f32_dpp.  Measured on one wave per SIMD (tests/diag/dpp_gemv.hip): 390 cycles per 64 products against 494 for
// 32 v_readlane + 32 v_pk_fma_f32 with scalar operand pairs (a lone wave issues a packed fp32 FMA every 8 cycles, not 4).
// sixteen products with the sixteen lanes of the row as ONE asm statement (hipcc pads every statement boundary with a wait state
// whose cost is an issue slot of the lone wave: 64 single-instruction statements carried 14 s_nop per step)
template <int J>
__device__ __forceinline__ void fmac_row16(float (&acc)[4], float H, const float (&w)[64]) {
    asm volatile(
        "v_fmac_f32_dpp %0, %4, %5 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %4, %6 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %4, %7 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, %4, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %4, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %4, %10 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %4, %11 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, %4, %12 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %4, %13 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %4, %14 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %4, %15 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, %4, %16 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %4, %17 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %4, %18 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %4, %19 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, %4, %20 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
        : "v"(H), "v"(w[16 * J]), "v"(w[16 * J + 1]), "v"(w[16 * J + 2]), "v"(w[16 * J + 3]), "v"(w[16 * J + 4]), "v"(w[16 * J + 5]),
          "v"(w[16 * J + 6]), "v"(w[16 * J + 7]), "v"(w[16 * J + 8]), "v"(w[16 * J + 9]), "v"(w[16 * J + 10]), "v"(w[16 * J + 11]),
          "v"(w[16 * J + 12]), "v"(w[16 * J + 13]), "v"(w[16 * J + 14]), "v"(w[16 * J + 15]));
}
__device__ __forceinline__ float dot64_rowbcast(const float (&H)[4], const float (&w)[64], float init = 0.f) {
    float acc[4] = {init, 0.f, 0.f, 0.f};
    // a DPP read needs two wait states behind a VALU write of its source (the swaps of rows_replicate); the compiler pads only
    // what it can see, not the inside of an asm statement
    asm volatile("s_nop 1" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    fmac_row16<0>(acc, H[0], w); fmac_row16<1>(acc, H[1], w); fmac_row16<2>(acc, H[2], w); fmac_row16<3>(acc, H[3], w);
    return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}
// where value k of the broadcast vector sits in its 64-float LDS line: lane l reads ONE float4 at 4 (l % 16) and has H[0..3]
__device__ __forceinline__ int rowbcast_slot(int k) { return (k & 15) * 4 + (k >> 4); }
// the same four registers from a value held lane-wise (lane l has value[l]): three half / row swaps, no LDS
__device__ __forceinline__ void rows_replicate(float v, float (&H)[4]) {
    const unsigned x = __float_as_uint(v);
    const auto r32 = __builtin_amdgcn_permlane32_swap(x, x, false, false);    // [r0 r1 r0 r1], [r2 r3 r2 r3]
    const auto a = __builtin_amdgcn_permlane16_swap(r32[0], r32[0], false, false);   // [r0 x4], [r1 x4]
    const auto b = __builtin_amdgcn_permlane16_swap(r32[1], r32[1], false, false);   // [r2 x4], [r3 x4]
    H[0] = __uint_as_float(a[0]); H[1] = __uint_as_float(a[1]); H[2] = __uint_as_float(b[0]); H[3] = __uint_as_float(b[1]);
}

#ifndef WM_LSTM_HS_BWD
#define WM_LSTM_HS_BWD 64       // measured: 32 -> 7.84 ms, 48 -> 7.65 ms, 64 (no LDS round trip at all) -> 7.53 ms
#endif
#ifndef WM_LSTM_HS
#define WM_LSTM_HS 32
#endif
template <bool SAVE>
__global__ __launch_bounds__(256) void lstm_fwd_kernel(const float* xp, const float* __restrict__ w_hh,
                                                       float* __restrict__ hout, float* gates,   // gates may alias xp
                                                       float* __restrict__ cst, int T) {
    constexpr int HS = WM_LSTM_HS;                  // h values taken through SGPRs (multiple of 8, power of two), rest via LDS broadcast
    __shared__ __align__(16) float hs[2][64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane & 3, ul = lane >> 2, u = wave * 16 + ul, n = q * 64 + u, np = wave * 64 + lane;
    // one W_hh row per lane, as 32 register pairs: the recurrent dot product issues v_pk_fma_f32, which keeps the
    // SIMD's fp32 pipe full from a single wave (a lone wave issues one VALU op per 4 cycles; a plain v_fma only
    // occupies the pipe for 2 of them)
    v2f wr[32];
#pragma unroll
    for (int k = 0; k < 64; k += 4) {
        const float4 v = *reinterpret_cast<const float4*>(w_hh + n * 64 + k);
        wr[k / 2] = v2f{v.x, v.y};
        wr[k / 2 + 1] = v2f{v.z, v.w};
    }
    if (tid < 128) (&hs[0][0])[tid] = 0.f;
    float c = 0.f;
    const float* xpb = xp + (size_t)b * T * 256 + np;
    float* gb = SAVE ? gates + (size_t)b * T * 256 + np : nullptr;
    float* cb = SAVE ? cst + (size_t)b * T * 64 + u : nullptr;
    float* hb = hout + ((size_t)b * 64 + u) * T;
    const bool is_g = (q == 2);

    float xa[16], xb_[16];
    auto prefetch = [&](float (&buf)[16], int t0) {
#pragma unroll
        for (int s = 0; s < 16; ++s) buf[s] = xpb[(size_t)min(t0 + s, T - 1) * 256];      // unconditional (clamped)
    };
    auto run_chunk = [&](const float (&xin)[16], int t0) {
        float hk[4];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int t = t0 + s;
            if (t < T) {                                   // uniform across the workgroup
                // h reaches the lanes two ways at once.  Broadcasting all 64 values to all 256 lanes through LDS reads
                // is 64 KB per step = 512 cycles of the CU's LDS return path -- the step's bottleneck.  So the first
                // HS values come as ONE ds_read_b32 per wave + v_readlane into SGPRs (VALU work, scalar FMA operands)
                // and only the rest as broadcast ds_read_b128: LDS and VALU each carry about half.
                const float hv = hs[s & 1][lane & (HS - 1)];
                const float4* hp = reinterpret_cast<const float4*>(hs[s & 1]);
                float4 hq[(64 - HS) / 4];
#pragma unroll
                for (int k = 0; k < (64 - HS) / 4; ++k) hq[k] = hp[HS / 4 + k];
                v2f a01 = v2f{xin[s], 0.f}, a23 = v2f{0.f, 0.f}, b01 = a23, b23 = a23;
                {
                    unsigned long long hp[HS / 2];
                    lanes_to_sgprs<HS>(hv, 0, hp);
#pragma unroll
                    for (int k = 0; k < HS; k += 8) {
                        a01 = pk_fma_s(wr[k / 2], hp[k / 2], a01);
                        a23 = pk_fma_s(wr[k / 2 + 1], hp[k / 2 + 1], a23);
                        b01 = pk_fma_s(wr[k / 2 + 2], hp[k / 2 + 2], b01);
                        b23 = pk_fma_s(wr[k / 2 + 3], hp[k / 2 + 3], b23);
                    }
                }
#pragma unroll
                for (int k = 0; k < (64 - HS) / 4; k += 2) {
                    const float4 h0 = hq[k], h1 = hq[k + 1];
                    a01 = __builtin_elementwise_fma(wr[HS / 2 + 2 * k], v2f{h0.x, h0.y}, a01);
                    a23 = __builtin_elementwise_fma(wr[HS / 2 + 2 * k + 1], v2f{h0.z, h0.w}, a23);
                    b01 = __builtin_elementwise_fma(wr[HS / 2 + 2 * k + 2], v2f{h1.x, h1.y}, b01);
                    b23 = __builtin_elementwise_fma(wr[HS / 2 + 2 * k + 3], v2f{h1.z, h1.w}, b23);
                }
                const v2f sm = (a01 + a23) + (b01 + b23);
                const float act = gate_act(sm.x + sm.y, is_g);
                const float gi = quad_bcast<0>(act), gf = quad_bcast<1>(act), gg = quad_bcast<2>(act), go = quad_bcast<3>(act);
                c = fmaf(gf, c, gi * gg);
                const float h = go * tanh_s(c);
                if (q == 0) hs[(s + 1) & 1][u] = h;
                if (SAVE) {
                    gb[(size_t)t * 256] = act;             // one contiguous 256-B segment per wave
                    if (q == 1) cb[(size_t)t * 64] = c;
                }
                if (((s >> 2) & 3) == q) hk[s & 3] = h;
                if (s == 15) {
                    // every lane now holds h for steps t0+4q .. t0+4q+3 of its unit
                    *reinterpret_cast<float4*>(hb + t0 + 4 * q) = make_float4(hk[0], hk[1], hk[2], hk[3]);
                }
                __syncthreads();
            }
        }
        // ragged tail (T not a multiple of 16): flush what the last partial chunk produced
        if (t0 + 16 > T) {
            const int tq = t0 + 4 * q;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (tq + j < T) hb[tq + j] = hk[j];
        }
    };

    prefetch(xa, 0);
    __syncthreads();
    for (int t0 = 0; t0 < T; t0 += 32) {
        prefetch(xb_, t0 + 16);
        run_chunk(xa, t0);
        prefetch(xa, t0 + 32);
        if (t0 + 16 < T) run_chunk(xb_, t0 + 16);
    }
}

// ------------------------------------------------------ recurrence fwd with the input projection inside
// Same recurrence as lstm_fwd_kernel, but xp = W_ih x + b is never materialised in HBM (4.2 GB written by
// lstm_xproj and read back here at B = 256).  While the workgroup steps through 32-step chunk c it also computes chunk
// c+1's projection [32 steps x 256 gate rows] on the bf16 matrix cores (bf16x6 split, fp32-grade; a separate pipe from
// the VALU the recurrence runs on): 48 MFMAs per wave dealt two per step between the recurrence's own instructions.
// W_ih lives in registers as 3-piece A fragments (96 VGPRs: this wave's 64 gate rows), the x tile goes global ->
// registers (two chunks ahead) -> split -> LDS [step][channel] pieces (B fragments), results land in a double-buffered
// LDS table xps[2][32][256(+1)] from which every lane picks its own gate row, one ds_read_b32 per step.
template <bool SAVE>
__global__ __launch_bounds__(256) void lstm_fwd_fused_kernel(const float* __restrict__ x, const float* __restrict__ w_ih,
                                                             const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                                             const float* __restrict__ w_hh, float* __restrict__ hout,
                                                             float* __restrict__ gates, float* __restrict__ cst, int T) {
    constexpr int HS = WM_LSTM_HS, CH = 32, XPP = 257, PITCH = 72, NP = 3;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    float* xps = reinterpret_cast<float*>(smem_raw);                                  // [2][CH][XPP]
    unsigned short* Xb = reinterpret_cast<unsigned short*>(xps + 2 * CH * XPP);       // [NP][CH][PITCH]
    float* hsm = reinterpret_cast<float*>(Xb + NP * CH * PITCH);                      // [2][64]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int q = lane & 3, ul = lane >> 2, u = wave * 16 + ul, n = q * 64 + u, np = wave * 64 + lane;
    float wr[64];                                  // W_hh row n of this lane
#pragma unroll
    for (int k = 0; k < 64; k += 4) {
        const float4 v = *reinterpret_cast<const float4*>(w_hh + n * 64 + k);
        wr[k] = v.x; wr[k + 1] = v.y; wr[k + 2] = v.z; wr[k + 3] = v.w;
    }
    // W_ih A fragments: row i = l31 of m-tile mt <-> gate column n' = wave*64 + mt*32 + l31, k = 16 ks + 8 half + j
    bf16x8 Wi[2][4][NP];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float* wp = w_ih + gate_row(wave * 64 + mt * 32 + l31) * 64 + 16 * ks + 8 * half;
            const float4 v0 = *reinterpret_cast<const float4*>(wp), v1 = *reinterpret_cast<const float4*>(wp + 4);
            unsigned a[4], m[4], l[4];
            split3_pair(v0.x, v0.y, a[0], m[0], l[0]); split3_pair(v0.z, v0.w, a[1], m[1], l[1]);
            split3_pair(v1.x, v1.y, a[2], m[2], l[2]); split3_pair(v1.z, v1.w, a[3], m[3], l[3]);
            Wi[mt][ks][0] = __builtin_bit_cast(bf16x8, make_uint4(a[0], a[1], a[2], a[3]));
            Wi[mt][ks][1] = __builtin_bit_cast(bf16x8, make_uint4(m[0], m[1], m[2], m[3]));
            Wi[mt][ks][2] = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
        }
    const float bias = b_ih[n] + b_hh[n];
    if (tid < 128) hsm[tid] = 0.f;
    float c = 0.f;
    // per-step outputs through buffer descriptors: the step index travels as a scalar offset, the lane's part is fixed, lanes that
    // have nothing to store point past the end of the descriptor (the hardware drops the access): no 64-bit address arithmetic and
    // no lane mask on the recurrence's critical path
    const wm_srd_t sgb = make_srd(SAVE ? gates + (size_t)b * T * 256 : hout, SAVE ? (size_t)T * 256 * sizeof(float) : 0);
    const wm_srd_t scb = make_srd(SAVE ? cst + (size_t)b * T * 64 : hout, SAVE ? (size_t)T * 64 * sizeof(float) : 0);
    const wm_srd_t shb = make_srd(hout + (size_t)b * 64 * T, (size_t)64 * T * sizeof(float));
    const unsigned vgb = (unsigned)np * 4u, vcb = (q == 1) ? (unsigned)u * 4u : 0xFFFFFF00u;
    const unsigned vhb = (unsigned)(u * T + 4 * q) * 4u;
    float* hb = hout + ((size_t)b * 64 + u) * T;
    const bool is_g = (q == 2);

    // x tile staging (one combo per thread): channel pair cp, time quad tq of the chunk
    const int cp = wave * 8 + (lane & 7), tq = lane >> 3;
    const float* xc = x + ((size_t)b * 64 + 2 * cp) * T + 4 * tq;
    float4 sa, sb;
    auto load_x = [&](int t0) {                    // clamped: chunks past the end re-read the last valid quad
        const int t = min(t0, T - 4 - 4 * tq);
        sa = *reinterpret_cast<const float4*>(xc + max(t, -4 * tq));
        sb = *reinterpret_cast<const float4*>(xc + T + max(t, -4 * tq));
    };
    auto split_x = [&]() {
        unsigned* X32 = reinterpret_cast<unsigned*>(Xb);
        const float va[4] = {sa.x, sa.y, sa.z, sa.w}, vb[4] = {sb.x, sb.y, sb.z, sb.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned p0, p1, p2;
            split3_pair(va[e], vb[e], p0, p1, p2);
            const int o = (4 * tq + e) * (PITCH / 2) + cp;
            X32[o] = p0; X32[CH * PITCH / 2 + o] = p1; X32[2 * (CH * PITCH / 2) + o] = p2;
        }
    };
    f32x16 acc[2];
    bf16x8 Bf[NP];
    // idx-th of the 48 MFMAs of a chunk's projection: block m = idx / 6 = ks * 2 + mt, piece product idx % 6.  They are
    // issued ONE at a time, far apart in the instruction stream: back-to-back dependent MFMAs would stall the wave's
    // in-order issue (and with it the latency-bound recurrence) for the 32 cycles each one occupies the pipe.
    auto mfma_one = [&](int idx) {
        const int m = idx / 6, pr = idx % 6, ks = m >> 1, mt = m & 1;
        const int pa = (pr == 0 || pr == 4) ? 1 : (pr == 2 ? 2 : 0), pb = (pr == 0 || pr == 3) ? 1 : (pr == 1 ? 2 : 0);
        if (mt == 0 && pr == 0) {
#pragma unroll
            for (int p = 0; p < NP; ++p) Bf[p] = *reinterpret_cast<const bf16x8*>(Xb + (p * CH + l31) * PITCH + 16 * ks + 8 * half);
        }
        if (ks == 0 && pr == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
        }
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wi[mt][ks][pa], Bf[pb], acc[mt], 0, 0, 0);
    };
    auto store_xp = [&](int buf) {                 // D row = gate column (this wave's 64), D column = step
        float* dst = xps + (buf * CH + l31) * XPP + wave * 64 + 4 * half;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[mt * 32 + (r & 3) + 8 * (r >> 2)] = acc[mt][r];
    };

    // ---- prologue: projection of chunk 0, x tile of chunk 1 in flight
    load_x(0);
    __syncthreads();
    split_x();
    load_x(CH);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 48; ++i) mfma_one(i);
    store_xp(0);
    __syncthreads();

    float hk[4];
    int t0 = 0, cbuf = 0;
#ifdef WM_STAMP
    unsigned long long tm[6] = {0, 0, 0, 0, 0, 0};
#endif
    // eight steps of the chunk with compile-time positions s = 8*SUB + j (register-array indices must be constants)
    auto run8 = [&](auto sub_c) {
        constexpr int SUB = decltype(sub_c)::value;
#pragma unroll
        for (int j8 = 0; j8 < 8; ++j8) {
            const int s = SUB * 8 + j8;
            const int t = t0 + s;
            // ---- side work: projection of the next chunk (independent of the recurrence; no effect past the end)
            if (s == 0) { split_x(); load_x(t0 + 2 * CH); }
            if (s >= 2 && s < 26) mfma_one(2 * (s - 2));
            if (s == 28) store_xp(cbuf ^ 1);
            if (t < T) {                                   // uniform across the workgroup
                LSTAMP(ls0, c);
                const float xin = xps[(cbuf * CH + s) * XPP + np] + bias;
                const float* hcur = hsm + (s & 1) * 64;
                const float4 hq4 = *reinterpret_cast<const float4*>(hcur + 4 * (lane & 15));   // h(t-1), row-replicated: one read per lane
                float Hh[4] = {hq4.x, hq4.y, hq4.z, hq4.w};
                float hv = Hh[0];
                LSTAMP(ls1, hv);                           // h of the previous step is in registers (LDS read latency)
                Hh[0] = hv;
                if (s >= 2 && s < 26) mfma_one(2 * (s - 2) + 1);
                float pre = dot64_rowbcast(Hh, wr, xin);     // the projection term rides in as the first accumulator's start value
                LSTAMP(ls2, pre);                          // 64-term dot product of the lane's gate row done
                float act = gate_act(pre, is_g);
                LSTAMP(ls3, act);                          // gate activation (exp + rcp)
                const float gi = quad_bcast<0>(act), gf = quad_bcast<1>(act), gg = quad_bcast<2>(act), go = quad_bcast<3>(act);
                c = fmaf(gf, c, gi * gg);
                float h = go * tanh_s(c);
                LSTAMP(ls4, h);                            // quad exchange, cell update, tanh(c)
                hsm[((s + 1) & 1) * 64 + rowbcast_slot(u)] = h;            // the four lanes of a quad hold the same h: no lane mask
                if (SAVE) {
                    buf_store(sgb, act, vgb, (unsigned)t * 1024u);      // one contiguous 256-B segment per wave
                    buf_store(scb, c, vcb, (unsigned)t * 256u);         // the q == 1 lane of every quad
                }
                if (((s >> 2) & 3) == q) hk[s & 3] = h;
                if ((s & 15) == 15)                        // every lane holds h for steps 4q .. 4q+3 of this 16-step group
                    buf_store4(shb, f32x4{hk[0], hk[1], hk[2], hk[3]}, vhb, (unsigned)(t0 + (s - 15)) * 4u);
                LSTAMP(ls5, h);                            // h to LDS, saved activations / outputs issued
                __syncthreads();
                LSTAMP(ls6, c);                            // barrier
                LSTAMP_ACC(0, ls0, ls1); LSTAMP_ACC(1, ls1, ls2); LSTAMP_ACC(2, ls2, ls3); LSTAMP_ACC(3, ls3, ls4);
                LSTAMP_ACC(4, ls4, ls5); LSTAMP_ACC(5, ls5, ls6);
            }
        }
    };
    for (; t0 < T; t0 += CH, cbuf ^= 1) {
        run8(std::integral_constant<int, 0>{});
        run8(std::integral_constant<int, 1>{});
        run8(std::integral_constant<int, 2>{});
        run8(std::integral_constant<int, 3>{});
        // ragged tail (T % 16 != 0): flush what the last partial 16-step group produced
        if (t0 + CH > T && (T & 15)) {
            const int g0 = (T >> 4) << 4, tq0 = g0 + 4 * q;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (tq0 + j < T) hb[tq0 + j] = hk[j];
        }
    }
#ifdef WM_STAMP
    if (g_lstm_stamp && lane == 0) {
        unsigned long long* d = g_lstm_stamp + ((size_t)blockIdx.x * 4 + wave) * 6;
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i] = tm[i];
    }
#endif
}

// ------------------------------------------ recurrence fwd, wave-specialised: the input projection beside it
// lstm_fwd_fused_kernel with the roles split over eight waves (as lstm_bwd_ws_kernel): waves 0..3 run the recurrence and nothing else;
// waves 4..7 -- the second wave of each SIMD -- compute chunk c+1's projection [32 steps x 256 gate rows] on the matrix cores (bf16x6,
// W_ih fragments in THEIR registers: the recurrence waves lose 96 registers, 48 MFMAs, the x split and the table stores per chunk)
// into the other half of the double-buffered LDS table.  The helpers execute the recurrence's one barrier per step and use those
// barriers as their own phase separators: slot 0 of a chunk splits the x tile into the LDS image, slots 1..24 issue two MFMAs each,
// slot 26 stores the table.  Same arithmetic and summation order as the fused kernel: bit-identical results.
template <bool SAVE>
__global__ __launch_bounds__(512) void lstm_fwd_ws_kernel(const float* __restrict__ x, const float* __restrict__ w_ih,
                                                          const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                                          const float* __restrict__ w_hh, float* __restrict__ hout,
                                                          float* __restrict__ gates, float* __restrict__ cst, int T) {
    constexpr int CH = 32, XPP = 257, PITCH = 72, NP = 3;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    float* xps = reinterpret_cast<float*>(smem_raw);                                  // [2][CH][XPP]
    unsigned short* Xb = reinterpret_cast<unsigned short*>(xps + 2 * CH * XPP);       // [NP][CH][PITCH]
    float* hsm = reinterpret_cast<float*>(Xb + NP * CH * PITCH);                      // [2][64]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (wave < 4) {
        // ------------------------------------------------------------------ the recurrence
        __builtin_amdgcn_s_setprio(3);
        const int q = lane & 3, ul = lane >> 2, u = wave * 16 + ul, n = q * 64 + u, np = wave * 64 + lane;
        float wr[64];                                  // W_hh row n of this lane
#pragma unroll
        for (int k = 0; k < 64; k += 4) {
            const float4 v = *reinterpret_cast<const float4*>(w_hh + n * 64 + k);
            wr[k] = v.x; wr[k + 1] = v.y; wr[k + 2] = v.z; wr[k + 3] = v.w;
        }
        const float bias = b_ih[n] + b_hh[n];
        if (tid < 128) hsm[tid] = 0.f;
        float c = 0.f;
        const wm_srd_t sgb = make_srd(SAVE ? gates + (size_t)b * T * 256 : hout, SAVE ? (size_t)T * 256 * sizeof(float) : 0);
        const wm_srd_t scb = make_srd(SAVE ? cst + (size_t)b * T * 64 : hout, SAVE ? (size_t)T * 64 * sizeof(float) : 0);
        const wm_srd_t shb = make_srd(hout + (size_t)b * 64 * T, (size_t)64 * T * sizeof(float));
        const unsigned vgb = (unsigned)np * 4u, vcb = (q == 1) ? (unsigned)u * 4u : 0xFFFFFF00u;
        const unsigned vhb = (unsigned)(u * T + 4 * q) * 4u;
        float* hb = hout + ((size_t)b * 64 + u) * T;
        const bool is_g = (q == 2);
        __syncthreads(); __syncthreads(); __syncthreads();          // the helpers' prologue: projection of chunk 0
        float hk[4];
        int t0 = 0, cbuf = 0;
        auto run8 = [&](auto sub_c) {
            constexpr int SUB = decltype(sub_c)::value;
#pragma unroll
            for (int j8 = 0; j8 < 8; ++j8) {
                const int s = SUB * 8 + j8;
                const int t = t0 + s;
                if (t < T) {                                   // uniform across the workgroup
                    const float xin = xps[(cbuf * CH + s) * XPP + np] + bias;
                    const float* hcur = hsm + (s & 1) * 64;
                    const float4 hq4 = *reinterpret_cast<const float4*>(hcur + 4 * (lane & 15));   // h(t-1), row-replicated
                    const float Hh[4] = {hq4.x, hq4.y, hq4.z, hq4.w};
                    const float pre = dot64_rowbcast(Hh, wr, xin);
                    const float act = gate_act(pre, is_g);
                    const float gi = quad_bcast<0>(act), gf = quad_bcast<1>(act), gg = quad_bcast<2>(act), go = quad_bcast<3>(act);
                    c = fmaf(gf, c, gi * gg);
                    const float h = go * tanh_s(c);
                    hsm[((s + 1) & 1) * 64 + rowbcast_slot(u)] = h;
                    if (SAVE) {
                        buf_store(sgb, act, vgb, (unsigned)t * 1024u);
                        buf_store(scb, c, vcb, (unsigned)t * 256u);
                    }
                    if (((s >> 2) & 3) == q) hk[s & 3] = h;
                    if ((s & 15) == 15)
                        buf_store4(shb, f32x4{hk[0], hk[1], hk[2], hk[3]}, vhb, (unsigned)(t0 + (s - 15)) * 4u);
                    __syncthreads();
                }
            }
        };
        for (; t0 < T; t0 += CH, cbuf ^= 1) {
            run8(std::integral_constant<int, 0>{});
            run8(std::integral_constant<int, 1>{});
            run8(std::integral_constant<int, 2>{});
            run8(std::integral_constant<int, 3>{});
            if (t0 + CH > T && (T & 15)) {                   // ragged tail: flush what the last partial 16-step group produced
                const int g0 = (T >> 4) << 4, tq0 = g0 + 4 * q;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (tq0 + j < T) hb[tq0 + j] = hk[j];
            }
        }
        return;
    }
    // ---------------------------------------------------------------------- the helpers: projection of the next chunk
    const int hw = wave - 4;
    bf16x8 Wi[2][4][NP];                                   // W_ih A fragments: row l31 of m-tile mt <-> gate column hw*64 + mt*32 + l31
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float* wp = w_ih + gate_row(hw * 64 + mt * 32 + l31) * 64 + 16 * ks + 8 * half;
            const float4 v0 = *reinterpret_cast<const float4*>(wp), v1 = *reinterpret_cast<const float4*>(wp + 4);
            unsigned a[4], m[4], l[4];
            split3_pair(v0.x, v0.y, a[0], m[0], l[0]); split3_pair(v0.z, v0.w, a[1], m[1], l[1]);
            split3_pair(v1.x, v1.y, a[2], m[2], l[2]); split3_pair(v1.z, v1.w, a[3], m[3], l[3]);
            Wi[mt][ks][0] = __builtin_bit_cast(bf16x8, make_uint4(a[0], a[1], a[2], a[3]));
            Wi[mt][ks][1] = __builtin_bit_cast(bf16x8, make_uint4(m[0], m[1], m[2], m[3]));
            Wi[mt][ks][2] = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
        }
    const int cp = hw * 8 + (lane & 7), tq = lane >> 3;    // x tile staging: channel pair cp, time quad tq of the chunk
    const float* xc = x + ((size_t)b * 64 + 2 * cp) * T + 4 * tq;
    float4 sa, sb;
    auto load_x = [&](int t0) {                    // clamped: chunks past the end re-read the last valid quad
        const int t = min(t0, T - 4 - 4 * tq);
        sa = *reinterpret_cast<const float4*>(xc + max(t, -4 * tq));
        sb = *reinterpret_cast<const float4*>(xc + T + max(t, -4 * tq));
    };
    auto split_x = [&]() {
        unsigned* X32 = reinterpret_cast<unsigned*>(Xb);
        const float va[4] = {sa.x, sa.y, sa.z, sa.w}, vb[4] = {sb.x, sb.y, sb.z, sb.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned p0, p1, p2;
            split3_pair(va[e], vb[e], p0, p1, p2);
            const int o = (4 * tq + e) * (PITCH / 2) + cp;
            X32[o] = p0; X32[CH * PITCH / 2 + o] = p1; X32[2 * (CH * PITCH / 2) + o] = p2;
        }
    };
    f32x16 acc[2];
    bf16x8 Bf[NP];
    auto mfma_one = [&](int idx) {                 // idx-th of the 48 MFMAs of a chunk: block m = idx / 6 = ks * 2 + mt, piece product idx % 6
        const int m = idx / 6, pr = idx % 6, ks = m >> 1, mt = m & 1;
        const int pa = (pr == 0 || pr == 4) ? 1 : (pr == 2 ? 2 : 0), pb = (pr == 0 || pr == 3) ? 1 : (pr == 1 ? 2 : 0);
        if (mt == 0 && pr == 0) {
#pragma unroll
            for (int p = 0; p < NP; ++p) Bf[p] = *reinterpret_cast<const bf16x8*>(Xb + (p * CH + l31) * PITCH + 16 * ks + 8 * half);
        }
        if (ks == 0 && pr == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
        }
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wi[mt][ks][pa], Bf[pb], acc[mt], 0, 0, 0);
    };
    auto store_xp = [&](int buf) {                 // D row = gate column (this wave's 64), D column = step
        float* dst = xps + (buf * CH + l31) * XPP + hw * 64 + 4 * half;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[mt * 32 + (r & 3) + 8 * (r >> 2)] = acc[mt][r];
    };
    // ---- prologue: projection of chunk 0, x tile of chunk 1 in flight (three barriers, matched by the recurrence waves)
    load_x(0);
    __syncthreads();
    split_x();
    load_x(CH);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 48; ++i) mfma_one(i);
    store_xp(0);
    __syncthreads();
    int cbuf = 0;
    for (int t0 = 0; t0 < T; t0 += CH, cbuf ^= 1) {
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            if (s == 0) { split_x(); load_x(t0 + 2 * CH); }
            if (s >= 1 && s <= 24) { mfma_one(2 * (s - 1)); mfma_one(2 * (s - 1) + 1); }
            if (s == 26) store_xp(cbuf ^ 1);
            if (t0 + s < T) __syncthreads();           // = the barrier of one recurrence step
        }
    }
}

// ------------------------------------------------------------------------------- recurrence bwd
// gates: in = saved activations, out = pre-activation gradients da  [B,T,256] (column order n' = unit*4 + gate)
__global__ __launch_bounds__(256) void lstm_bwd_kernel(float* __restrict__ gates, const float* __restrict__ cst,
                                                       const float* __restrict__ dh_out, const float* __restrict__ w_hh,
                                                       int T) {
    __builtin_amdgcn_s_setprio(3);   // latency-bound recurrence: win issue arbitration against co-resident weight-gradient waves
    __shared__ __align__(16) float part[2][64][4];  // [buffer][k][wave] partial dh
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane & 3, ul = lane >> 2, u = wave * 16 + ul, np = wave * 64 + lane;
    // transposed-product operand: lane = output k, register j = the gate row held by lane j of this wave
    float wt[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) wt[j] = w_hh[((j & 3) * 64 + wave * 16 + (j >> 2)) * 64 + lane];
    if (tid < 512 / 4) reinterpret_cast<float4*>(&part[0][0][0])[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    float dc = 0.f;
    // per-step operands through buffer descriptors (scalar step offset + a fixed lane part): no 64-bit address arithmetic
    const wm_srd_t sgb = make_srd(gates + (size_t)b * T * 256, (size_t)T * 256 * sizeof(float));
    const wm_srd_t scb = make_srd(cst + (size_t)b * T * 64, (size_t)T * 64 * sizeof(float));
    const wm_srd_t sdh = make_srd(dh_out + (size_t)b * 64 * T, (size_t)64 * T * sizeof(float));
    const unsigned vgb = (unsigned)np * 4u, vcb = (unsigned)u * 4u, vdh = (unsigned)(u * T) * 4u;

    constexpr int CH = 8;
    struct Buf { float ga[CH], cc[CH + 1], dh[CH]; };
    Buf A, Bf;
    // chunk covers steps t1-CH+1 .. t1 (descending); cc[j] = c_{t1-CH+j} so cc[CH] = c_{t1}.  Loads are unconditional
    // (clamped indices); steps with t < 0 are never executed and c_{-1} is selected to 0 at use.
    auto prefetch = [&](Buf& f, int t1) {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int t = max(t1 - CH + 1 + j, 0);
            f.ga[j] = buf_load(sgb, vgb, (unsigned)t * 1024u);
            f.dh[j] = buf_load(sdh, vdh, (unsigned)t * 4u);
        }
#pragma unroll
        for (int j = 0; j <= CH; ++j) f.cc[j] = buf_load(scb, vcb, (unsigned)max(t1 - CH + j, 0) * 256u);
    };
    const bool is0 = q == 0, is1 = q == 1, is2 = q == 2, is3 = q == 3;
    int pb = 0;   // partial buffer parity
#ifdef WM_STAMP
    unsigned long long tm[6] = {0, 0, 0, 0, 0, 0};
#endif
    auto run_chunk = [&](const Buf& f, int t1) {
#pragma unroll
        for (int jj = 0; jj < CH; ++jj) {
            const int j = CH - 1 - jj, t = t1 - jj;
            if (t >= 0) {
                LSTAMP(ls0, dc);
                const float4 p = *reinterpret_cast<const float4*>(&part[pb][u][0]);
                // everything that does not need dh of this step first (it overlaps the LDS read above), branch-free: a lane's role
                // (its gate q) only selects operands.  da = base * S * D with  base = dc_t (gates i, f, g) | dh_t (gate o),
                // S = g | c_{t-1} | i | tanh(c_t),  D = a (1 - a) for the sigmoid gates, 1 - a^2 for g  (a = the lane's own activation)
                const float act = f.ga[j];
                const float gi = quad_bcast<0>(act), gf = quad_bcast<1>(act), gg = quad_bcast<2>(act), go = quad_bcast<3>(act);
                const float tc = tanh_s(f.cc[j + 1]);
                const float cprev = (t >= 1) ? f.cc[j] : 0.f;
                const float S = is0 ? gg : (is1 ? cprev : (is2 ? gi : tc));
                const float D = is2 ? fmaf(-act, act, 1.f) : act * (1.f - act);
                const float M = S * D;
                const float K = go * fmaf(-tc, tc, 1.f);
                float dht = f.dh[j] + ((p.x + p.y) + (p.z + p.w));
                LSTAMP(ls1, dht);                          // the four waves' partial dh read back and summed
                const float dct = fmaf(dht, K, dc);
                float da = (is3 ? dht : dct) * M;
                dc = dct * gf;
                LSTAMP(ls2, da);                           // quad exchange, tanh(c), gate derivatives
                buf_store(sgb, da, vgb, (unsigned)t * 1024u);
                // lane k's partial dh[k] over the wave's own 64 gate rows: da row-replicated by three half / row swaps, then 64 products
                // with the DPP row broadcast (no lane reads, no LDS round trip)
                float Dd[4];
                rows_replicate(da, Dd);
                float psum = dot64_rowbcast(Dd, wt);
                LSTAMP(ls3, psum);                         // W_hh^T da over the wave's 64 gate rows (readlane + pk_fma)
                part[pb ^ 1][lane][wave] = psum;
                pb ^= 1;
                LSTAMP(ls4, psum);
                __syncthreads();
                LSTAMP(ls5, dc);
                LSTAMP_ACC(0, ls0, ls1); LSTAMP_ACC(1, ls1, ls2); LSTAMP_ACC(2, ls2, ls3); LSTAMP_ACC(3, ls3, ls4); LSTAMP_ACC(4, ls4, ls5);
            }
        }
    };
    prefetch(A, T - 1);
    __syncthreads();
    for (int t1 = T - 1; t1 >= 0; t1 -= 2 * CH) {
        prefetch(Bf, t1 - CH);
        run_chunk(A, t1);
        prefetch(A, t1 - 2 * CH);
        run_chunk(Bf, t1 - CH);
    }
#ifdef WM_STAMP
    if (g_lstm_stamp && lane == 0) {
        unsigned long long* d = g_lstm_stamp + ((size_t)blockIdx.x * 4 + wave) * 6;
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i] = tm[i];
    }
#endif
}

// ------------------------------------------------ recurrence bwd with the weight gradients beside it
// Wave-specialised BPTT.  Waves 0..3 run lstm_bwd_kernel's recurrence unchanged (one clip per workgroup, one barrier per step); waves
// 4..7 -- a second wave on each SIMD, whose matrix cores the recurrence never touches -- form the clip's weight gradients
//   dW_ih[n'][c] = sum_t da[t][n'] x[t][c],   dW_hh[n'][k] = sum_t da[t][n'] h[t-1][k]
// out of the 32-step chunk of da the recurrence finished LAST, while it walks the next one: da never has to be read back from HBM
// for them (lstm_wgrad_bf_kernel: 4.2 GB at B = 256, and its A fragments -- 8 consecutive steps of one gate column -- were gathered
// with ds_read_u16 from a [step][n'] image).  Here the recurrence lane that owns gate column n' drops its eight da values of an
// 8-step block as two ds_write_b128 into a [n'][step] fp32 image (double-buffered by chunk); a helper lane reads 8 consecutive steps
// of one column = one A fragment, splits it in registers (bf16x6) and multiplies it with B fragments that come straight from global
// memory (x and h are [B,64,T]: 8 consecutive steps of one channel are contiguous).  Helper wave w owns gate columns [64 w, 64 w + 64)
// for all 128 z rows (128 accumulator registers), 96 MFMAs per chunk = 3 per step.  Helpers execute the recurrence's barrier once
// per step (s_barrier counts every wave of the workgroup) with a 1/32 slice of a chunk's work in between, so they can never run
// behind; the recurrence keeps issue priority (s_setprio 3).  Per-clip slab: [256 n'][128 z] + [256] bias sums (the recurrence lanes
// add their own da), reduced by lstm_wgrad_reduce_kernel.  T % 32 == 0.
__global__ __launch_bounds__(512) void lstm_bwd_ws_kernel(float* __restrict__ gates, const float* __restrict__ cst,
                                                          const float* __restrict__ dh_out, const float* __restrict__ w_hh,
                                                          const float* __restrict__ x, const float* __restrict__ hseq,
                                                          float* __restrict__ partial, int T) {
    constexpr int CT = 32, PT = 36;                 // chunk length, fp32 pitch of one gate column's chunk (16-byte multiple)
    extern __shared__ __align__(16) unsigned char smem_raw[];
    float* daT = reinterpret_cast<float*>(smem_raw);                      // [2][256][PT]
    float (*part)[64][4] = reinterpret_cast<float (*)[64][4]>(daT + 2 * 256 * PT);     // [2][64][4] partial dh
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* out = partial + (size_t)b * (256 * 128 + 256);
    if (wave < 4) {
        // ------------------------------------------------------------------ the recurrence (lstm_bwd_kernel)
        __builtin_amdgcn_s_setprio(3);
        const int q = lane & 3, ul = lane >> 2, u = wave * 16 + ul, np = wave * 64 + lane;
        float wt[64];
#pragma unroll
        for (int j = 0; j < 64; ++j) wt[j] = w_hh[((j & 3) * 64 + wave * 16 + (j >> 2)) * 64 + lane];
        if (tid < 512 / 4) reinterpret_cast<float4*>(&part[0][0][0])[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
        float dc = 0.f, dbs = 0.f;
        const wm_srd_t sgb = make_srd(gates + (size_t)b * T * 256, (size_t)T * 256 * sizeof(float));
        const wm_srd_t scb = make_srd(cst + (size_t)b * T * 64, (size_t)T * 64 * sizeof(float));
        const wm_srd_t sdh = make_srd(dh_out + (size_t)b * 64 * T, (size_t)64 * T * sizeof(float));
        const unsigned vgb = (unsigned)np * 4u, vcb = (unsigned)u * 4u, vdh = (unsigned)(u * T) * 4u;
        constexpr int CH = 8;
        struct Buf { float ga[CH], cc[CH + 1], dh[CH]; };
        Buf A, Bf;
        auto prefetch = [&](Buf& f, int t1) {
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int t = max(t1 - CH + 1 + j, 0);
                f.ga[j] = buf_load(sgb, vgb, (unsigned)t * 1024u);
                f.dh[j] = buf_load(sdh, vdh, (unsigned)t * 4u);
            }
#pragma unroll
            for (int j = 0; j <= CH; ++j) f.cc[j] = buf_load(scb, vcb, (unsigned)max(t1 - CH + j, 0) * 256u);
        };
        const bool is0 = q == 0, is1 = q == 1, is2 = q == 2, is3 = q == 3;
        int pb = 0;
        auto run_chunk = [&](const Buf& f, int t1) {            // t1 % 8 == 7, t1 >= 7 (T % 16 == 0)
            float dav[CH];
#pragma unroll
            for (int jj = 0; jj < CH; ++jj) {
                const int j = CH - 1 - jj, t = t1 - jj;
                const float4 p = *reinterpret_cast<const float4*>(&part[pb][u][0]);
                const float act = f.ga[j];
                const float gi = quad_bcast<0>(act), gf = quad_bcast<1>(act), gg = quad_bcast<2>(act), go = quad_bcast<3>(act);
                const float tc = tanh_s(f.cc[j + 1]);
                const float cprev = (t >= 1) ? f.cc[j] : 0.f;
                const float S = is0 ? gg : (is1 ? cprev : (is2 ? gi : tc));
                const float D = is2 ? fmaf(-act, act, 1.f) : act * (1.f - act);
                const float M = S * D;
                const float K = go * fmaf(-tc, tc, 1.f);
                const float dht = f.dh[j] + ((p.x + p.y) + (p.z + p.w));
                const float dct = fmaf(dht, K, dc);
                const float da = (is3 ? dht : dct) * M;
                dc = dct * gf;
                dav[j] = da;
                buf_store(sgb, da, vgb, (unsigned)t * 1024u);
                float Dd[4];
                rows_replicate(da, Dd);
                const float psum = dot64_rowbcast(Dd, wt);
                part[pb ^ 1][lane][wave] = psum;
                pb ^= 1;
                if (jj == CH - 1) {                              // the block's eight values, oldest step first: [n'][t % 32 ...]
                    const int t0 = t1 - (CH - 1);
                    float* d = daT + (((t0 >> 5) & 1) * 256 + np) * PT + (t0 & (CT - 1));
                    *reinterpret_cast<float4*>(d) = make_float4(dav[0], dav[1], dav[2], dav[3]);
                    *reinterpret_cast<float4*>(d + 4) = make_float4(dav[4], dav[5], dav[6], dav[7]);
                }
                dbs += da;
                __syncthreads();
            }
        };
        prefetch(A, T - 1);
        __syncthreads();
        for (int t1 = T - 1; t1 >= 0; t1 -= 2 * CH) {
            prefetch(Bf, t1 - CH);
            run_chunk(A, t1);
            prefetch(A, t1 - 2 * CH);
            run_chunk(Bf, t1 - CH);
        }
        out[256 * 128 + np] = dbs;
        return;
    }
    // ---------------------------------------------------------------------- the helpers: weight gradients of the finished chunk
    const int hw = wave - 4, half = lane >> 5, l31 = lane & 31;
    f32x16 acc[2][2][2];                           // [z: x | h][row tile of the wave's 64 gate columns][column tile of 64 channels]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i >> 2][(i >> 1) & 1][i & 1][r] = 0.f;
    const wm_srd_t sx = make_srd(x + (size_t)b * 64 * T, (size_t)64 * T * sizeof(float));
    const wm_srd_t sh = make_srd(hseq + (size_t)b * 64 * T, (size_t)64 * T * sizeof(float));
    // unit u of a chunk = (ks = u >> 2, z = (u >> 1) & 1, ct = u & 1): the B fragment (z, ct, ks) against both row tiles' A fragments
    // of k-step ks.  Raw operands of unit u + 1 are fetched while unit u multiplies.
    float raw[8];
    auto load_raw = [&](int chunk, int u_) {       // chunk, u_ wave-uniform: the step rides in the scalar offset
        const int ks = u_ >> 2, z = (u_ >> 1) & 1, ct = u_ & 1;
        const unsigned v = (unsigned)((32 * ct + l31) * T + 8 * half) * 4u;
        const int ts = chunk * CT + 16 * ks;
        if (z == 0) {
            const f32x4 a0 = buf_load4(sx, v, (unsigned)ts * 4u), a1 = buf_load4(sx, v, (unsigned)(ts + 4) * 4u);
            raw[0] = a0[0]; raw[1] = a0[1]; raw[2] = a0[2]; raw[3] = a0[3]; raw[4] = a1[0]; raw[5] = a1[1]; raw[6] = a1[2]; raw[7] = a1[3];
        } else {                                    // h[t - 1]: one step earlier (not 16-byte aligned), zero initial state
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int te = ts + e - 1;
                // te = -1 (first step of the clip, wave-uniform): the upper half's lanes want step 7 = voffset - 4, the lower half's h[-1] = 0
                const float r_ = buf_load(sh, te < 0 ? v - 4u : v, (unsigned)max(te, 0) * 4u);
                raw[e] = (te < 0 && half == 0) ? 0.f : r_;
            }
        }
    };
    auto split8 = [&](const float (&v)[8], bf16x8 (&P)[3]) {
        unsigned w_[3][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) split3_pair(v[2 * i], v[2 * i + 1], w_[0][i], w_[1][i], w_[2][i]);
#pragma unroll
        for (int p = 0; p < 3; ++p) P[p] = __builtin_bit_cast(bf16x8, make_uint4(w_[p][0], w_[p][1], w_[p][2], w_[p][3]));
    };
    bf16x8 Af[2][3], Bp[3];
    auto slot = [&](int chunk, int sl) {            // slot sl = 0..31 of the chunk's work; four slots per unit
        const int u_ = sl >> 2, k = sl & 3, ks = u_ >> 2, z = (u_ >> 1) & 1, ct = u_ & 1;
        if (k == 0) {
            if ((u_ & 3) == 0) {                    // new k-step: the two row tiles' A fragments from the [n'][t] image
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const float* src = daT + ((chunk & 1) * 256 + 64 * hw + 32 * rt + l31) * PT + 16 * ks + 8 * half;
                    const float4 a0 = *reinterpret_cast<const float4*>(src), a1 = *reinterpret_cast<const float4*>(src + 4);
                    const float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                    split8(v, Af[rt]);
                }
            }
            split8(raw, Bp);
            if (u_ < 7) load_raw(chunk, u_ + 1); else load_raw(max(chunk - 1, 0), 0);
            return;
        }
        // k = 1..3: four of the unit's twelve piece products each (bf16x6 order: small terms first)
        constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = 4 * (k - 1) + i, rt = m / 6, j = m % 6;
            acc[z][rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Af[rt][PA[j]], Bp[PB[j]], acc[z][rt][ct], 0, 0, 0);
        }
    };
    const int nch = T / CT;
    load_raw(nch - 1, 0);
    __syncthreads();                                 // the recurrence's start-up barrier
    for (int c = nch - 1; c >= 0; --c) {             // the recurrence walks chunk c; chunk c + 1 is complete
        const bool work = c + 1 < nch;
#pragma unroll
        for (int sl = 0; sl < CT; ++sl) {
            if (work) slot(c + 1, sl);
            __syncthreads();                         // = the barrier of one recurrence step
        }
    }
#pragma unroll
    for (int sl = 0; sl < CT; ++sl) slot(0, sl);     // chunk 0, after the recurrence's last step
#pragma unroll
    for (int z = 0; z < 2; ++z)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    out[(64 * hw + 32 * rt + mfma_row(r, half)) * 128 + 64 * z + 32 * ct + l31] = acc[z][rt][ct][r];
}

// --------------------------------------------------- recurrence bwd with the input gradient inside
// Same BPTT as lstm_bwd_kernel; in addition dx[b,c,t] = sum_n' w_ih[gate_row(n')][c] da[t][n'] is formed here instead of
// in a separate GEMM over the 4.2-GB da tensor.  Every lane drops the three bf16 pieces of its da into a wave-private
// LDS image [piece][step][gate column] as it produces it; one 32-step chunk later the wave multiplies that image with its
// 64 rows of W_ih^T (A fragments resident in registers) on the bf16 matrix cores -- 48 MFMAs dealt two per step beside
// the VALU recurrence -- and the four waves' partial [64 x 32] tiles are summed through LDS and stored as coalesced rows.
__global__ __launch_bounds__(256) void lstm_bwd_fused_kernel(float* __restrict__ gates, const float* __restrict__ cst,
                                                             const float* __restrict__ dh_out, const float* __restrict__ w_hh,
                                                             const float* __restrict__ w_ih, float* __restrict__ dx, int T) {
    constexpr int HS = WM_LSTM_HS, CHK = 32, PITCH = 72, NP = 3, PDP = 33;
    constexpr int DIMG = NP * CHK * PITCH;                       // bf16 elements of one wave's image
    extern __shared__ __align__(16) unsigned char smem_raw[];
    unsigned short* Dab = reinterpret_cast<unsigned short*>(smem_raw);             // [2][4 waves][NP][CHK][PITCH]
    float* Pd = reinterpret_cast<float*>(Dab + 2 * 4 * DIMG);                       // [4 waves][64][PDP]
    float* das = Pd + 4 * 64 * PDP;                                                 // [4][64] wave-private da vectors
    float* part = das + 4 * 64;                                                     // [2][64][4] partial dh
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int q = lane & 3, ul = lane >> 2, u = wave * 16 + ul, np = wave * 64 + lane;
    v2f wt[32];                                    // transposed W_hh operand: lane = output k, register j = gate row of lane j
#pragma unroll
    for (int j = 0; j < 64; j += 2)
        wt[j / 2] = v2f{w_hh[((j & 3) * 64 + wave * 16 + (j >> 2)) * 64 + lane],
                        w_hh[(((j + 1) & 3) * 64 + wave * 16 + ((j + 1) >> 2)) * 64 + lane]};
    // W_ih^T A fragments: A[i = channel mt*32 + l31][k = this wave's gate column 16 ks + 8 half + j]
    bf16x8 Wt[2][4][NP];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = w_ih[gate_row(wave * 64 + 16 * ks + 8 * half + j) * 64 + mt * 32 + l31];
            unsigned a[4], m[4], l[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) split3_pair(v[2 * j], v[2 * j + 1], a[j], m[j], l[j]);
            Wt[mt][ks][0] = __builtin_bit_cast(bf16x8, make_uint4(a[0], a[1], a[2], a[3]));
            Wt[mt][ks][1] = __builtin_bit_cast(bf16x8, make_uint4(m[0], m[1], m[2], m[3]));
            Wt[mt][ks][2] = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
        }
    if (tid < 128) reinterpret_cast<float4*>(part)[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    float dc = 0.f;
    float* gb = gates + (size_t)b * T * 256 + np;
    const float* cb = cst + (size_t)b * T * 64 + u;
    const float* dhb = dh_out + ((size_t)b * 64 + u) * T;

    constexpr int CH = 8;
    struct Buf { float ga[CH], cc[CH + 1], dh[CH]; };
    Buf A, Bf;
    auto prefetch = [&](Buf& f, int t1) {          // steps t1-CH+1 .. t1; unconditional loads from clamped indices
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int t = max(t1 - CH + 1 + j, 0);
            f.ga[j] = gb[(size_t)t * 256];
            f.dh[j] = dhb[t];
        }
#pragma unroll
        for (int j = 0; j <= CH; ++j) f.cc[j] = cb[(size_t)max(t1 - CH + j, 0) * 64];
    };
    f32x16 acc[2];
    bf16x8 Bq[NP];
    int ibuf = 0;                                  // image being filled by the current 32-step chunk
    // idx-th of the 48 MFMAs of the previous chunk's product (block idx / 6 = ks*2 + mt, piece product idx % 6), issued one
    // at a time and far apart: see lstm_fwd_fused_kernel
    auto mfma_one = [&](int idx, int img) {
        const int m = idx / 6, pr = idx % 6, ks = m >> 1, mt = m & 1;
        const int pa = (pr == 0 || pr == 4) ? 1 : (pr == 2 ? 2 : 0), pbi = (pr == 0 || pr == 3) ? 1 : (pr == 1 ? 2 : 0);
        const unsigned short* im = Dab + (img * 4 + wave) * DIMG;
        if (mt == 0 && pr == 0) {
#pragma unroll
            for (int p = 0; p < NP; ++p) Bq[p] = *reinterpret_cast<const bf16x8*>(im + (p * CHK + l31) * PITCH + 16 * ks + 8 * half);
        }
        if (ks == 0 && pr == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
        }
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wt[mt][ks][pa], Bq[pbi], acc[mt], 0, 0, 0);
    };
    auto store_partial = [&]() {                   // D row = channel, D column = step of the chunk
        float* dst = Pd + (wave * 64 + 4 * half) * PDP + l31;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[(mt * 32 + (r & 3) + 8 * (r >> 2)) * PDP] = acc[mt][r];
    };
    auto reduce_store = [&](int tlo) {             // thread -> channel tid>>2, 8 steps from 8*(tid&3)
        const int ch = tid >> 2, j0 = 8 * (tid & 3);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = ch * PDP + j0 + j;
            o[j] = (Pd[i] + Pd[64 * PDP + i]) + (Pd[2 * 64 * PDP + i] + Pd[3 * 64 * PDP + i]);
        }
        float* dst = dx + ((size_t)b * 64 + ch) * T + tlo + j0;
        if (tlo + j0 >= 0) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        if (tlo + j0 + 4 >= 0) *reinterpret_cast<float4*>(dst + 4) = make_float4(o[4], o[5], o[6], o[7]);
    };
    int pb = 0;   // partial buffer parity
    int prev_tlo = 0;
    bool have_prev = false;
    // SUB = 8-step quarter of the 32-step chunk whose top step is t1 + 8*SUB
    auto run_chunk = [&](const Buf& f, int t1, auto sub_c) {
        constexpr int SUB = decltype(sub_c)::value;
        unsigned short* myimg = Dab + (ibuf * 4 + wave) * DIMG;
#pragma unroll
        for (int jj = 0; jj < CH; ++jj) {
            const int j = CH - 1 - jj, t = t1 - jj;
            constexpr int dummy = 0; (void)dummy;
            const int s32 = SUB * 8 + jj, tt = 31 - s32;   // position in the chunk: time = tlo + tt
            // ---- side work for the previous chunk (its image is complete)
            if (have_prev) {
                if (s32 >= 2 && s32 < 26) mfma_one(2 * (s32 - 2), ibuf ^ 1);
                if (s32 == 26) store_partial();
                if (s32 == 27) { __syncthreads(); reduce_store(prev_tlo); }
            }
            if (t >= 0) {
                const float4 p = *reinterpret_cast<const float4*>(part + (pb * 64 + u) * 4);
                const float dht = f.dh[j] + ((p.x + p.y) + (p.z + p.w));
                const float act = f.ga[j];
                const float gi = quad_bcast<0>(act), gf = quad_bcast<1>(act), gg = quad_bcast<2>(act), go = quad_bcast<3>(act);
                const float tc = tanh_s(f.cc[j + 1]);
                const float cprev = (t >= 1) ? f.cc[j] : 0.f;
                const float dct = fmaf(dht * go, 1.f - tc * tc, dc);
                float da;
                if (q == 0) da = dct * gg * gi * (1.f - gi);
                else if (q == 1) da = dct * cprev * gf * (1.f - gf);
                else if (q == 2) da = dct * gi * (1.f - gg * gg);
                else da = dht * tc * go * (1.f - go);
                dc = dct * gf;
                gb[(size_t)t * 256] = da;
                das[wave * 64 + lane] = da;                 // same wave reads it back: no barrier needed
                __builtin_amdgcn_wave_barrier();
                const float4* dp = reinterpret_cast<const float4*>(das + wave * 64);
                float4 dq[(64 - HS) / 4];
#pragma unroll
                for (int k = 0; k < (64 - HS) / 4; ++k) dq[k] = dp[HS / 4 + k];
                {   // under the latency of those reads: bf16 pieces of da for the dx product, [piece][step tt][gate column]
                    unsigned p0, p1, p2;
                    split3_pair(da, 0.f, p0, p1, p2);
                    unsigned short* d = myimg + tt * PITCH + lane;
                    d[0] = (unsigned short)p0; d[CHK * PITCH] = (unsigned short)p1; d[2 * CHK * PITCH] = (unsigned short)p2;
                }
                v2f a01 = v2f{0.f, 0.f}, a23 = a01;
                // two batches of 32 lane reads, each ahead of its 16 products (64 scalar registers at once would not fit beside the rest)
#pragma unroll
                for (int k0 = 0; k0 < HS; k0 += 32) {
                    constexpr int NB = HS < 32 ? HS : 32;
                    unsigned long long dp[NB / 2];
                    lanes_to_sgprs<NB>(da, k0, dp);
#pragma unroll
                    for (int k = 0; k < NB; k += 4) {
                        a01 = pk_fma_s(wt[(k0 + k) / 2], dp[k / 2], a01);
                        a23 = pk_fma_s(wt[(k0 + k) / 2 + 1], dp[k / 2 + 1], a23);
                    }
                }
#pragma unroll
                for (int k = 0; k < (64 - HS) / 4; ++k) {
                    const float4 d = dq[k];
                    a01 = __builtin_elementwise_fma(wt[HS / 2 + 2 * k], v2f{d.x, d.y}, a01);
                    a23 = __builtin_elementwise_fma(wt[HS / 2 + 2 * k + 1], v2f{d.z, d.w}, a23);
                }
                const v2f sm = a01 + a23;
                part[((pb ^ 1) * 64 + lane) * 4 + wave] = sm.x + sm.y;
                pb ^= 1;
                if (have_prev && s32 >= 2 && s32 < 26) mfma_one(2 * (s32 - 2) + 1, ibuf ^ 1);
                __syncthreads();
            } else {
                // steps before the clip: zero pieces so that the partial last chunk multiplies zeros
                unsigned short* d = myimg + tt * PITCH + lane;
                d[0] = 0; d[CHK * PITCH] = 0; d[2 * CHK * PITCH] = 0;
                if (have_prev && s32 >= 2 && s32 < 26) mfma_one(2 * (s32 - 2) + 1, ibuf ^ 1);
            }
        }
    };
    prefetch(A, T - 1);
    __syncthreads();
    for (int t1 = T - 1; t1 >= 0; t1 -= CHK) {
        prefetch(Bf, t1 - CH);
        run_chunk(A, t1, std::integral_constant<int, 0>{});
        prefetch(A, t1 - 2 * CH);
        run_chunk(Bf, t1 - CH, std::integral_constant<int, 1>{});
        prefetch(Bf, t1 - 3 * CH);
        run_chunk(A, t1 - 2 * CH, std::integral_constant<int, 2>{});
        prefetch(A, t1 - 4 * CH);
        run_chunk(Bf, t1 - 3 * CH, std::integral_constant<int, 3>{});
        prev_tlo = t1 - (CHK - 1);
        have_prev = true;
        ibuf ^= 1;
    }
    // flush: the last chunk's product
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 48; ++i) mfma_one(i, ibuf ^ 1);
    store_partial();
    __syncthreads();
    reduce_store(prev_tlo);
}

// ------------------------------------------------------------------------------------ dx GEMM
// dx[b,c,t] = sum_n da[b,t,n'] w_ih[gate_row(n')][c] ;  persistent blocks over (clip, 64-step) tiles, the next da
// tile is fetched into registers while the matrix cores work on the current one; wave = (c half, t half)
__global__ __launch_bounds__(256) void lstm_dx_kernel(const float* __restrict__ da, const float* __restrict__ w_ih,
                                                      float* __restrict__ dx, int B, int T) {
    constexpr int NT = 64, DS = 257;
    extern __shared__ __align__(16) float smem[];
    float* ws = smem;               // [256][64]  A[i = c][k = n'] = w_ih[gate_row(n')][c]
    float* ds = smem + 256 * 64;    // [NT][DS]   B[k = n'][j = t] = da[t][n']
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int tilesPerClip = (T + NT - 1) / NT, ntiles = B * tilesPerClip;
    for (int i = tid; i < 256 * 16; i += 256)      // row n' of the LDS image = parameter row gate_row(n')
        reinterpret_cast<float4*>(ws)[i] = reinterpret_cast<const float4*>(w_ih)[gate_row(i >> 4) * 16 + (i & 15)];
    float4 st[16];
    auto load_piece = [&](int tile, int k) {       // branch-free: rows past T read a clamped row, never stored
        const int b = tile / tilesPerClip, t0 = (tile % tilesPerClip) * NT;
        const int i = tid + k * 256;
        st[k] = reinterpret_cast<const float4*>(da + (size_t)b * T * 256)[(size_t)min(t0 + (i >> 6), T - 1) * 64 + (i & 63)];
    };
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int k = 0; k < 16; ++k) load_piece(tile, k);
    };
    auto write_tile = [&]() {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = tid + k * 256;
            float* d = ds + (i >> 6) * DS + 4 * (i & 63);
            d[0] = st[k].x; d[1] = st[k].y; d[2] = st[k].z; d[3] = st[k].w;
        }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) load_tile(tile);
    __syncthreads();
    if (tile < ntiles) write_tile();
    __syncthreads();
    const int mt = wave & 1, nt = wave >> 1;
    const float* ap = ws + half * 64 + mt * 32 + l31;
    const float* bp = ds + (nt * 32 + l31) * DS + half;
    while (tile < ntiles) {
        const int next = tile + gridDim.x, nextc = min(next, ntiles - 1);     // clamped: loaded (valid memory), never written
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int g8 = 0; g8 < 16; ++g8) {              // the next tile's 64 KB arrive one piece per 8 MFMAs, not as one burst
            load_piece(nextc, g8);
#pragma unroll
            for (int s = 8 * g8; s < 8 * g8 + 8; ++s) acc = mfma32(ap[2 * s * 64], bp[2 * s], acc);
        }
        const int b = tile / tilesPerClip, t0 = (tile % tilesPerClip) * NT;
        const int t = t0 + nt * 32 + l31;
        if (t < T) {
#pragma unroll
            for (int r = 0; r < 16; ++r) dx[((size_t)b * 64 + mt * 32 + mfma_row(r, half)) * T + t] = acc[r];
        }
        __syncthreads();
        if (next < ntiles) write_tile();
        __syncthreads();
        tile = next;
    }
}

// ------------------------------------------------------------------------------------ dx GEMM, bf16x6
// Same product as lstm_dx_kernel on the bf16 matrix cores (bf16x6 split, fp32-grade).  The contraction runs over the gate
// column n', which is the contiguous axis of da [t][n']: a B fragment (one time step, 8 consecutive n') is one aligned
// ds_read_b128 of the 3-piece LDS image of the tile, no transposition.  W_ih^T lives in registers as A fragments
// (16 k-steps x 3 pieces = 192 VGPRs for this wave's 32 channels).  Wave = (channel half mt, time half nt) of a 64-step tile.
__global__ __launch_bounds__(256) void lstm_dx_bf_kernel(const float* __restrict__ da, const float* __restrict__ w_ih,
                                                         float* __restrict__ dx, int B, int T) {
    constexpr int NT = 64, NP = 3, PD = 264;         // image row pitch in bf16 (528 B = 33 x 16 B)
    extern __shared__ __align__(16) unsigned char smem_raw[];
    unsigned short* Db = reinterpret_cast<unsigned short*>(smem_raw);          // [NP][NT][PD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int mt = wave & 1, nt = wave >> 1;
    const int tilesPerClip = (T + NT - 1) / NT, ntiles = B * tilesPerClip;
    // A[i = channel mt*32 + l31][k = n' = 16 ks + 8 half + j] = w_ih[gate_row(n')][channel]
    bf16x8 Wt[16][NP];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = w_ih[gate_row(16 * ks + 8 * half + j) * 64 + mt * 32 + l31];
        unsigned a[4], m[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) split3_pair(v[2 * j], v[2 * j + 1], a[j], m[j], l[j]);
        Wt[ks][0] = __builtin_bit_cast(bf16x8, make_uint4(a[0], a[1], a[2], a[3]));
        Wt[ks][1] = __builtin_bit_cast(bf16x8, make_uint4(m[0], m[1], m[2], m[3]));
        Wt[ks][2] = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
    }
    float4 st[16];
    auto load_piece = [&](int tile, int k) {       // branch-free: rows past T read a clamped row, never stored
        const int b = tile / tilesPerClip, t0 = (tile % tilesPerClip) * NT;
        const int i = tid + k * 256;
        st[k] = reinterpret_cast<const float4*>(da + (size_t)b * T * 256)[(size_t)min(t0 + (i >> 6), T - 1) * 64 + (i & 63)];
    };
    auto write_tile = [&]() {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = tid + k * 256;
            unsigned a0, a1, a2, b0, b1, b2;
            split3_pair(st[k].x, st[k].y, a0, a1, a2);
            split3_pair(st[k].z, st[k].w, b0, b1, b2);
            unsigned short* d = Db + (i >> 6) * PD + 4 * (i & 63);
            *reinterpret_cast<uint2*>(d) = make_uint2(a0, b0);
            *reinterpret_cast<uint2*>(d + NT * PD) = make_uint2(a1, b1);
            *reinterpret_cast<uint2*>(d + 2 * NT * PD) = make_uint2(a2, b2);
        }
    };
    int tile = blockIdx.x;
#pragma unroll
    for (int k = 0; k < 16; ++k) load_piece(min(tile, ntiles - 1), k);
    __syncthreads();
    write_tile();
    __syncthreads();
    const unsigned short* brow = Db + (nt * 32 + l31) * PD + 8 * half;
    while (tile < ntiles) {
        const int next = tile + gridDim.x, nextc = min(next, ntiles - 1);     // clamped: loaded (valid memory), never written
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            bf16x8 Bf[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) Bf[p] = *reinterpret_cast<const bf16x8*>(brow + p * NT * PD + 16 * ks);
            acc = mfma_bf16x6(Wt[ks], Bf, acc);
            load_piece(nextc, ks);                 // the next tile's 64 KB arrive one piece per k-step
        }
        const int b = tile / tilesPerClip, t0 = (tile % tilesPerClip) * NT;
        const int t = t0 + nt * 32 + l31;
        if (t < T) {
#pragma unroll
            for (int r = 0; r < 16; ++r) dx[((size_t)b * 64 + mt * 32 + mfma_row(r, half)) * T + t] = acc[r];
        }
        __syncthreads();
        if (next < ntiles) write_tile();
        __syncthreads();
        tile = next;
    }
}

// ---------------------------------------------------------------------------------- wgrad GEMM
// G[n'][j] = sum_{b,t} da[b,t,n'] * z[b,j,t],  z = [x (64 rows) ; h shifted by one step (64 rows)]
// persistent blocks over (clip, 64-step) tiles, next tile prefetched into registers during the MFMA phase;
// partial[block][256*128 + 256 (bias)]
__global__ __launch_bounds__(256) void lstm_wgrad_kernel(const float* __restrict__ da, const float* __restrict__ x,
                                                         const float* __restrict__ h, float* __restrict__ partial,
                                                         int B, int T) {
    constexpr int NT = 64, ZS = 67;  // odd stride: the B operand is read down a column (32 lanes = 32 rows)
    extern __shared__ __align__(16) float smem[];
    float* ds = smem;                // [NT][256]
    float* zs = smem + NT * 256;     // [128][ZS]; h rows are stored shifted: column tt holds h[t0 + tt - 1]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int tilesPerClip = (T + NT - 1) / NT, ntiles = B * tilesPerClip;
    f32x16 acc[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    float bsum = 0.f;
    float4 sd[16], sx[4], sh[4];
    float hh;
    // piece p of a tile's staging: 0..15 da rows, 16..19 x, 20..23 h, 24 the halo column h[t0 - 1].  Branch-free (clamped
    // addresses), masked when written to LDS.  The main loop issues one piece per k-step: the 96-KB tile would otherwise
    // hit the memory pipeline as one burst and block the wave at issue (see conv64bf3_kernel).
    auto load_piece = [&](int tile, int p) {
        const int b = tile / tilesPerClip, t0 = (tile % tilesPerClip) * NT;
        if (p < 16) {
            const int i = tid + p * 256;
            sd[p] = reinterpret_cast<const float4*>(da + (size_t)b * T * 256)[(size_t)min(t0 + (i >> 6), T - 1) * 64 + (i & 63)];
        } else if (p < 24) {
            const int k = p & 3, i = tid + k * 256, j = i >> 4, q = i & 15;
            const size_t off = ((size_t)b * 64 + j) * T + min(t0 + 4 * q, T - 4);
            if (p < 20) sx[k] = *reinterpret_cast<const float4*>(x + off);
            else sh[k] = *reinterpret_cast<const float4*>(h + off);
        } else {
            hh = h[((size_t)b * 64 + (tid & 63)) * T + max(t0 - 1, 0)];
        }
    };
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int p = 0; p < 25; ++p) load_piece(tile, p);
    };
    auto write_tile = [&](int tile) {
        const int t0 = (tile % tilesPerClip) * NT;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = tid + k * 256;
            float4 v = sd[k];
            if (t0 + (i >> 6) >= T) v = make_float4(0.f, 0.f, 0.f, 0.f);      // zero rows kill every product past T
            reinterpret_cast<float4*>(ds)[i] = v;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = tid + k * 256, j = i >> 4, q = i & 15;
            float* zx = zs + j * ZS + 4 * q;
            zx[0] = sx[k].x; zx[1] = sx[k].y; zx[2] = sx[k].z; zx[3] = sx[k].w;
            float* zh = zs + (64 + j) * ZS + 4 * q + 1;
            zh[0] = sh[k].x; zh[1] = sh[k].y; zh[2] = sh[k].z; zh[3] = sh[k].w;
        }
        if (tid < 64) zs[(64 + tid) * ZS] = (t0 >= 1) ? hh : 0.f;              // zero initial state
    };
    int tile = blockIdx.x;
    if (tile < ntiles) load_tile(tile);
    __syncthreads();
    if (tile < ntiles) write_tile(tile);
    __syncthreads();
    // A[i = n'][k = t] = ds[t][n'] ; B[k = t][j] = zs[j][t]
    const float* ap = ds + half * 256 + wave * 64 + l31;
    const float* bp = zs + l31 * ZS + half;
    while (tile < ntiles) {
        const int next = tile + gridDim.x, nextc = min(next, ntiles - 1);     // clamped: loaded (valid memory), never written
#pragma unroll
        for (int s = 0; s < NT / 2; ++s) {
            if (s < 25) load_piece(nextc, s);
            const float a0 = ap[2 * s * 256], a1 = ap[2 * s * 256 + 32];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const float bv = bp[nt * 32 * ZS + 2 * s];
                acc[0][nt] = mfma32(a0, bv, acc[0][nt]);
                acc[1][nt] = mfma32(a1, bv, acc[1][nt]);
            }
        }
#pragma unroll 8
        for (int tt = 0; tt < NT; ++tt) bsum += ds[tt * 256 + tid];
        __syncthreads();
        if (next < ntiles) write_tile(next);
        __syncthreads();
        tile = next;
    }
    float* out = partial + (size_t)blockIdx.x * (256 * 128 + 256);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                out[(wave * 64 + mt * 32 + mfma_row(r, half)) * 128 + nt * 32 + l31] = acc[mt][nt][r];
    out[256 * 128 + tid] = bsum;
}

// ---------------------------------------------------------------------------------- wgrad GEMM, bf16x6
// Same sums as lstm_wgrad_kernel on the bf16 matrix cores (bf16x6 split).  The contraction axis is time, the STRIDED
// axis of da [t][n']: its A fragments (one gate column, 8 consecutive steps) are gathered from the [piece][step][n']
// LDS image with eight ds_read_u16 per piece (lanes run along n': conflict-free) and packed with v_perm; z = [x ; h
// shifted by one step] is time-contiguous, so its B fragments are aligned ds_read_b128.  32-step tiles; wave w owns the
// gate columns [64 w, 64 w + 64) for all 128 z rows (128 accumulator registers).
__global__ __launch_bounds__(256) void lstm_wgrad_bf_kernel(const float* __restrict__ da, const float* __restrict__ x,
                                                            const float* __restrict__ h, float* __restrict__ partial,
                                                            int B, int T) {
    constexpr int NT = 32, NP = 3, PDD = 264, PZ = 40;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    unsigned short* Dd = reinterpret_cast<unsigned short*>(smem_raw);          // [NP][NT][PDD]   da pieces, [step][n']
    unsigned short* Zb = Dd + NP * NT * PDD;                                   // [NP][128][PZ]   z pieces, [row][step]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int tilesPerClip = (T + NT - 1) / NT, ntiles = B * tilesPerClip;
    f32x16 acc[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    float bs[4] = {0.f, 0.f, 0.f, 0.f};            // bias sums of gate columns 4 (tid & 63) .. + 3 over this thread's steps
    float4 sd[8], sx[2];
    float sh[2][4];
    // piece p: 0..7 da rows (step (tid >> 6) + 4 p, columns 4 (tid & 63)..), 8..9 x, 10..11 h (shifted: column tt = h[t0 + tt - 1])
    auto load_piece = [&](int tile, int p) {
        const int b = tile / tilesPerClip, t0 = (tile % tilesPerClip) * NT;
        if (p < 8) {
            const int t = min(t0 + (tid >> 6) + 4 * p, T - 1);
            sd[p] = reinterpret_cast<const float4*>(da + ((size_t)b * T + t) * 256)[tid & 63];
        } else {
            const int k = p & 1, i = tid + k * 256, j = i >> 3, q = i & 7;
            const size_t row = ((size_t)b * 64 + j) * T;
            if (p < 10) sx[k] = *reinterpret_cast<const float4*>(x + row + min(t0 + 4 * q, T - 4));
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) sh[k][e] = h[row + min(max(t0 + 4 * q + e - 1, 0), T - 1)];
            }
        }
    };
    auto put4 = [&](unsigned short* d, int stride_p, float v0, float v1, float v2, float v3) {
        unsigned a0, a1, a2, b0, b1, b2;
        split3_pair(v0, v1, a0, a1, a2);
        split3_pair(v2, v3, b0, b1, b2);
        *reinterpret_cast<uint2*>(d) = make_uint2(a0, b0);
        *reinterpret_cast<uint2*>(d + stride_p) = make_uint2(a1, b1);
        *reinterpret_cast<uint2*>(d + 2 * stride_p) = make_uint2(a2, b2);
    };
    auto write_tile = [&](int tile) {
        const int t0 = (tile % tilesPerClip) * NT;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int tt = (tid >> 6) + 4 * p;
            float4 v = sd[p];
            if (t0 + tt >= T) v = make_float4(0.f, 0.f, 0.f, 0.f);            // zero rows kill every product past T
            bs[0] += v.x; bs[1] += v.y; bs[2] += v.z; bs[3] += v.w;
            put4(Dd + tt * PDD + 4 * (tid & 63), NT * PDD, v.x, v.y, v.z, v.w);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = tid + k * 256, j = i >> 3, q = i & 7;
            put4(Zb + j * PZ + 4 * q, 128 * PZ, sx[k].x, sx[k].y, sx[k].z, sx[k].w);
            float hv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) hv[e] = (t0 + 4 * q + e - 1 >= 0) ? sh[k][e] : 0.f;      // zero initial state
            put4(Zb + (64 + j) * PZ + 4 * q, 128 * PZ, hv[0], hv[1], hv[2], hv[3]);
        }
    };
    int tile = blockIdx.x;
#pragma unroll
    for (int p = 0; p < 12; ++p) load_piece(min(tile, ntiles - 1), p);
    __syncthreads();
    write_tile(min(tile, ntiles - 1));
    __syncthreads();
    while (tile < ntiles) {
        const int next = tile + gridDim.x, nextc = min(next, ntiles - 1);     // clamped: loaded (valid memory), never written
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int e0 = kb * 16 + 8 * half;
            bf16x8 Bf[4][NP];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    Bf[nt][p] = *reinterpret_cast<const bf16x8*>(Zb + (p * 128 + nt * 32 + l31) * PZ + e0);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                bf16x8 A[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const unsigned short* col = Dd + (p * NT + e0) * PDD + wave * 64 + mt * 32 + l31;
                    unsigned w[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = (unsigned)col[(2 * j) * PDD] | ((unsigned)col[(2 * j + 1) * PDD] << 16);
                    A[p] = __builtin_bit_cast(bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
                }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    acc[mt][nt] = mfma_bf16x6(A, Bf[nt], acc[mt][nt]);
                    if ((kb * 2 + mt) * 4 + nt < 12) load_piece(nextc, (kb * 2 + mt) * 4 + nt);     // 12 pieces ride along the first 12 of 16 blocks
                }
            }
        }
        __syncthreads();
        if (next < ntiles) write_tile(next);
        __syncthreads();
        tile = next;
    }
    float* out = partial + (size_t)blockIdx.x * (256 * 128 + 256);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                out[(wave * 64 + mt * 32 + mfma_row(r, half)) * 128 + nt * 32 + l31] = acc[mt][nt][r];
    // bias: the four threads tid, tid + 64, .. share the gate columns 4 (tid & 63) .. + 3
    float* red = reinterpret_cast<float*>(smem_raw);          // [4][256]
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) red[(tid >> 6) * 256 + 4 * (tid & 63) + e] = bs[e];
    __syncthreads();
    out[256 * 128 + tid] = (red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid]);
}

__global__ void lstm_wgrad_reduce_kernel(const float* __restrict__ partial, int nparts, float* dw_ih, float* dw_hh,
                                         float* db_ih, float* db_hh, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    constexpr int stride = 256 * 128 + 256;
    if (i >= stride) return;
    double sd = 0.0;
    for (int p = 0; p < nparts; ++p) sd += (double)partial[(size_t)p * stride + i];
    const float s = (float)sd;
    if (i < 256 * 128) {
        const int n = gate_row(i >> 7), j = i & 127;
        float* dst = (j < 64) ? dw_ih + n * 64 + j : dw_hh + n * 64 + (j - 64);
        *dst = accumulate ? *dst + s : s;
    } else {
        const int n = gate_row(i - 256 * 128);
        db_ih[n] = accumulate ? db_ih[n] + s : s;
        db_hh[n] = accumulate ? db_hh[n] + s : s;
    }
}

}  // namespace

extern "C" {

#ifdef WM_STAMP
int wm_debug_set_lstm_stamp_buffer(unsigned long long* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_lstm_stamp), &buf, sizeof(buf));
}
#endif

// xp [B,T,256] = x[B,64,T]^T W_ih^T + b_ih + b_hh
int wm_lstm_xproj(const float* x, const float* w_ih, const float* b_ih, const float* b_hh, float* xp, int B, int T,
                  hipStream_t stream) {
    if (T & 3) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(lstm_xproj_kernel, dim3(B * ((T + 127) / 128)), dim3(256), 0, stream, x, w_ih, b_ih, b_hh, xp, T);
    WM_CHECK_LAUNCH();
    return 0;
}

// hout [B,64,T]; gates [B,T,256] and cst [B,T,64] are written only when both are non-NULL (training).
int wm_lstm_fwd(const float* xp, const float* w_hh, float* hout, float* gates, float* cst, int B, int T, hipStream_t stream) {
    if (T & 3) return (int)hipErrorInvalidValue;
    if (gates && cst) hipLaunchKernelGGL(lstm_fwd_kernel<true>, dim3(B), dim3(256), 0, stream, xp, w_hh, hout, gates, cst, T);
    else hipLaunchKernelGGL(lstm_fwd_kernel<false>, dim3(B), dim3(256), 0, stream, xp, w_hh, hout, gates, cst, T);
    WM_CHECK_LAUNCH();
    return 0;
}

static int g_lstm_fwd_ws = 1;       // 1: lstm_fwd_ws_kernel (projection on helper waves) | 0: lstm_fwd_fused_kernel (inside the recurrence's waves)
int wm_set_lstm_fwd_wave_specialised(int on, hipStream_t) { g_lstm_fwd_ws = on ? 1 : 0; return 0; }

// Recurrence with the input projection inside (no xp tensor): x [B,64,T] -> hout; gates / cst as wm_lstm_fwd.
int wm_lstm_fwd_fused(const float* x, const float* w_ih, const float* b_ih, const float* b_hh, const float* w_hh, float* hout,
                      float* gates, float* cst, int B, int T, hipStream_t stream) {
    if ((T & 3) || T < 8) return (int)hipErrorInvalidValue;
    constexpr size_t lds = (size_t)2 * 32 * 257 * sizeof(float) + (size_t)3 * 32 * 72 * 2 + 128 * sizeof(float);
    static wm::DevOnce done;
    if (!wm::dev_done(done)) {
        WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_fwd_fused_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_fwd_fused_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        wm::dev_mark(done);
    }
    if (g_lstm_fwd_ws) {                            // wave-specialised build: the projection on four helper waves (default)
        static wm::DevOnce donew;
        if (!wm::dev_done(donew)) {
            WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_fwd_ws_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_fwd_ws_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            wm::dev_mark(donew);
        }
        if (gates && cst) hipLaunchKernelGGL(lstm_fwd_ws_kernel<true>, dim3(B), dim3(512), lds, stream, x, w_ih, b_ih, b_hh, w_hh, hout, gates, cst, T);
        else hipLaunchKernelGGL(lstm_fwd_ws_kernel<false>, dim3(B), dim3(512), lds, stream, x, w_ih, b_ih, b_hh, w_hh, hout, gates, cst, T);
        WM_CHECK_LAUNCH();
        return 0;
    }
    if (gates && cst) hipLaunchKernelGGL(lstm_fwd_fused_kernel<true>, dim3(B), dim3(256), lds, stream, x, w_ih, b_ih, b_hh, w_hh, hout, gates, cst, T);
    else hipLaunchKernelGGL(lstm_fwd_fused_kernel<false>, dim3(B), dim3(256), lds, stream, x, w_ih, b_ih, b_hh, w_hh, hout, gates, cst, T);
    WM_CHECK_LAUNCH();
    return 0;
}

// gates: saved activations in, da out.  dh_out [B,64,T] = gradient w.r.t. hout.
int wm_lstm_bwd(float* gates, const float* cst, const float* dh_out, const float* w_hh, int B, int T, hipStream_t stream) {
    hipLaunchKernelGGL(lstm_bwd_kernel, dim3(B), dim3(256), 0, stream, gates, cst, dh_out, w_hh, T);
    WM_CHECK_LAUNCH();
    return 0;
}

// BPTT + input gradient in one launch (no separate pass over da for dx): gates in = saved activations, out = da.
int wm_lstm_bwd_fused(float* gates, const float* cst, const float* dh_out, const float* w_hh, const float* w_ih, float* dx,
                      int B, int T, hipStream_t stream) {
    if ((T & 3) || T < 4) return (int)hipErrorInvalidValue;
    constexpr size_t lds = (size_t)2 * 4 * 3 * 32 * 72 * 2 + (size_t)(4 * 64 * 33 + 4 * 64 + 2 * 64 * 4) * sizeof(float);
    static wm::DevOnce done;
    if (!wm::dev_done(done)) {
        WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_bwd_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        wm::dev_mark(done);
    }
    hipLaunchKernelGGL(lstm_bwd_fused_kernel, dim3(B), dim3(256), lds, stream, gates, cst, dh_out, w_hh, w_ih, dx, T);
    WM_CHECK_LAUNCH();
    return 0;
}

// BPTT + the clip's weight gradients in one launch (wave-specialised: see lstm_bwd_ws_kernel), then the fixed-order reduction
// over clips.  gates: saved activations in, da out (for wm_lstm_dx).  partial: >= B * (256*128 + 256) floats.  T % 32 == 0, T >= 64.
int wm_lstm_bwd_wgrad(float* gates, const float* cst, const float* dh_out, const float* w_hh, const float* x, const float* h,
                      float* partial, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh, int B, int T, int accumulate,
                      hipStream_t stream) {
    if (B <= 0 || (T & 31) || T < 64) return (int)hipErrorInvalidValue;
    constexpr size_t lds = (size_t)(2 * 256 * 36 + 2 * 64 * 4) * sizeof(float);
    static wm::DevOnce done;
    if (!wm::dev_done(done)) {
        WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_bwd_ws_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        wm::dev_mark(done);
    }
    hipLaunchKernelGGL(lstm_bwd_ws_kernel, dim3(B), dim3(512), lds, stream, gates, cst, dh_out, w_hh, x, h, partial, T);
    WM_CHECK_LAUNCH();
    constexpr int n = 256 * 128 + 256;
    hipLaunchKernelGGL(lstm_wgrad_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, (const float*)partial, B,
                       dw_ih, dw_hh, db_ih, db_hh, accumulate);
    WM_CHECK_LAUNCH();
    return 0;
}

static int g_lstm_dx_bf = 1;        // 1: bf16x6 split on the bf16 matrix cores (default) | 0: native fp32 MFMA
int wm_set_lstm_dx_bf16x6(int on, hipStream_t) { g_lstm_dx_bf = on ? 1 : 0; return 0; }

int wm_lstm_dx(const float* da, const float* w_ih, float* dx, int B, int T, hipStream_t stream) {
    const int ntiles = B * ((T + 63) / 64);
    if (g_lstm_dx_bf) {
        constexpr size_t ldsb = (size_t)3 * 64 * 264 * 2;
        static wm::DevOnce doneb;
        if (!wm::dev_done(doneb)) { WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_dx_bf_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb)); wm::dev_mark(doneb); }
        hipLaunchKernelGGL(lstm_dx_bf_kernel, dim3(ntiles < kNumCU ? ntiles : kNumCU), dim3(256), ldsb, stream, da, w_ih, dx, B, T);
        WM_CHECK_LAUNCH();
        return 0;
    }
    constexpr size_t lds = (size_t)(256 * 64 + 64 * 257) * sizeof(float);
    static wm::DevOnce done;
    if (!wm::dev_done(done)) { WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_dx_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); wm::dev_mark(done); }
    hipLaunchKernelGGL(lstm_dx_kernel, dim3(ntiles < kNumCU ? ntiles : kNumCU), dim3(256), lds, stream, da, w_ih, dx, B, T);
    WM_CHECK_LAUNCH();
    return 0;
}

// partial: >= 256 * (256*128 + 256) floats
int wm_lstm_wgrad(const float* da, const float* x, const float* h, float* partial, float* dw_ih, float* dw_hh,
                  float* db_ih, float* db_hh, int B, int T, int accumulate, hipStream_t stream) {
    int grid;
    if (g_lstm_dx_bf) {                           // same switch as wm_lstm_dx: bf16x6 (default) | native fp32 MFMA
        constexpr size_t ldsb = (size_t)(3 * 32 * 264 + 3 * 128 * 40) * 2;
        static wm::DevOnce doneb;
        if (!wm::dev_done(doneb)) { WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_wgrad_bf_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb)); wm::dev_mark(doneb); }
        const int ntiles = B * ((T + 31) / 32);
        grid = ntiles < kNumCU ? ntiles : kNumCU;
        hipLaunchKernelGGL(lstm_wgrad_bf_kernel, dim3(grid), dim3(256), ldsb, stream, da, x, h, partial, B, T);
    } else {
        constexpr size_t lds = (size_t)(64 * 256 + 128 * 67) * sizeof(float);
        static wm::DevOnce done;
        if (!wm::dev_done(done)) { WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); wm::dev_mark(done); }
        const int ntiles = B * ((T + 63) / 64);
        grid = ntiles < kNumCU ? ntiles : kNumCU;
        hipLaunchKernelGGL(lstm_wgrad_kernel, dim3(grid), dim3(256), lds, stream, da, x, h, partial, B, T);
    }
    WM_CHECK_LAUNCH();
    constexpr int n = 256 * 128 + 256;
    hipLaunchKernelGGL(lstm_wgrad_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, (const float*)partial, grid,
                       dw_ih, dw_hh, db_ih, db_hh, accumulate);
    WM_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
