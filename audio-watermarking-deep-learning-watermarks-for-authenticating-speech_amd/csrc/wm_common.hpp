// Shared device helpers for the gfx950 (MI355X / CDNA4) watermark kernels.
// Wave = 64 lanes everywhere; no portability layer on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

#define WM_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)
#define WM_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)

namespace wm {

// hipFuncSetAttribute (the > 64 KB dynamic-LDS opt-in) is a PER-DEVICE setting: remember it per device ordinal, so one
// process may drive several GPUs (and several host threads: the attribute call is idempotent, a race only repeats it).
// The only mutable state the launchers keep is this cache.
struct DevOnce {
    std::atomic<unsigned long long> mask{0};
};
inline bool dev_done(const DevOnce& d) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    return (d.mask.load(std::memory_order_acquire) >> (dev & 63)) & 1ull;
}
inline void dev_mark(DevOnce& d) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    d.mask.fetch_or(1ull << (dev & 63), std::memory_order_release);
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;
constexpr int kNumCU = 256;          // MI355X: 8 XCD x 32 CU

// D(32x32) += A(32x2) * B(2x32), exact fp32 (v_mfma_f32_32x32x2_f32).
// lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31];
// D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5) for register r in [0,16).
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int mfma_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// sum over the 32 lanes that share (lane>>5)
__device__ __forceinline__ float half_wave_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// Transposing butterfly: every lane brings 32 values x[0..31]; on return lane l of each 32-lane half-wave holds
// the sum over that half-wave of x[l & 31].  31 shuffles instead of the 160 of 32 independent tree reductions,
// and the 5 stages are 16/8/4/2/1 independent shuffles wide, so they pipeline.
__device__ __forceinline__ float half_wave_transpose_sum32(float (&x)[32], int lane) {
#pragma unroll
    for (int s = 4; s >= 0; --s) {
        const int h = 1 << s;
        const bool up = (lane >> s) & 1;
#pragma unroll
        for (int i = 0; i < h; ++i) {
            const float keep = up ? x[i + h] : x[i];
            const float send = up ? x[i] : x[i + h];
            x[i] = keep + __shfl_xor(send, h);
        }
    }
    return x[0];
}
// 16-value variant: on return every lane l holds the half-wave total of x[l & 15]
__device__ __forceinline__ float half_wave_transpose_sum16(float (&x)[16], int lane) {
#pragma unroll
    for (int s = 3; s >= 0; --s) {
        const int h = 1 << s;
        const bool up = (lane >> s) & 1;
#pragma unroll
        for (int i = 0; i < h; ++i) {
            const float keep = up ? x[i + h] : x[i];
            const float send = up ? x[i] : x[i + h];
            x[i] = keep + __shfl_xor(send, h);
        }
    }
    return x[0] + __shfl_xor(x[0], 16);
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// block-wide sum for blocks of NW waves; result valid in every thread. scratch: >= NW floats of LDS.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) r += scratch[i];
    return r;
}
template <int NW>
__device__ __forceinline__ double block_sum_d(double v, double* scratch) {
    v = wave_sum_d(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    double r = 0.0;
#pragma unroll
    for (int i = 0; i < NW; ++i) r += scratch[i];
    return r;
}

__device__ __forceinline__ float sigmoidf_acc(float x) { return 1.0f / (1.0f + expf(-x)); }

// ELU (alpha = 1) with a cheap expm1: Taylor to the 6th order on (-0.35, 0] (error < 4e-7 of the value), exp(v) - 1 below
__device__ __forceinline__ float elu1(float v) {
    const float p = v * (1.f + v * (0.5f + v * (0.16666667f + v * (0.041666668f + v * (0.0083333338f + v * 0.0013888889f)))));
    const float e = __expf(v) - 1.f;
    const float n = v > -0.35f ? p : e;
    return v > 0.f ? v : n;
}



// ---- bf16x6 split arithmetic (see conv64.hip): x = hi + mid + lo as three bf16 pieces, exact to 24 bits
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// two values -> three dwords, each holding the (a: low half, b: high half) pair of one piece
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    bf16x2 h = {(__bf16)a, (__bf16)b};
    p0 = __builtin_bit_cast(unsigned, h);
    a -= __uint_as_float(p0 << 16); b -= __uint_as_float(p0 & 0xffff0000u);
    bf16x2 m = {(__bf16)a, (__bf16)b};
    p1 = __builtin_bit_cast(unsigned, m);
    a -= __uint_as_float(p1 << 16); b -= __uint_as_float(p1 & 0xffff0000u);
    bf16x2 l = {(__bf16)a, (__bf16)b};
    p2 = __builtin_bit_cast(unsigned, l);
}

// acc += A*B to fp32 grade from the 3-piece fragments: the six piece products of weight >= 2^-16, small terms first
__device__ __forceinline__ f32x16 mfma_bf16x6(const bf16x8 (&A)[3], const bf16x8 (&B)[3], f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[2], B[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[0], acc, 0, 0, 0);
    return acc;
}

// ---- buffer addressing (scalar descriptor + 32-bit per-lane byte offset + scalar offset; out-of-range lanes read 0 / are dropped)
typedef __amdgpu_buffer_rsrc_t wm_srd_t;   // 128-bit buffer resource descriptor (kept in scalar registers)

// buffer descriptor over [p, p + bytes): 32-bit per-lane byte offsets + a scalar offset, reads past the end return 0
__device__ __forceinline__ wm_srd_t make_srd(const float* p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), (short)0, (int)(bytes > 0xfffff000ull ? 0xfffff000ull : bytes),
                                             0x00020000);
}
__device__ __forceinline__ float buf_load(wm_srd_t srd, unsigned voff_bytes, unsigned soff_bytes) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd, (int)voff_bytes, (int)soff_bytes, 0));
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 buf_load4(wm_srd_t srd, unsigned voff_bytes, unsigned soff_bytes) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, (int)voff_bytes, (int)soff_bytes, 0));
}
__device__ __forceinline__ void buf_store4(wm_srd_t srd, f32x4 v, unsigned voff_bytes, unsigned soff_bytes) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), srd, (int)voff_bytes, (int)soff_bytes, 0);
}
__device__ __forceinline__ void buf_store(wm_srd_t srd, float v, unsigned voff_bytes, unsigned soff_bytes) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), srd, (int)voff_bytes, (int)soff_bytes, 0);
}

}  // namespace wm
