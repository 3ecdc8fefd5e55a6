// BatchNorm1d(64) pieces that sit between the MFMA convolutions of a ResBlock
// (py/main16.py:112-125).  The convolution kernels fuse "normalise + ReLU" into their
// loads and the per-channel sum / sum-of-squares into their epilogues; what is left is
//   * finalize: partial sums -> (scale, shift), saved (mean, invstd), running statistics
//   * the block tail  out = relu(x + bn2(y2))
//   * backward: dz2 = g * (out > 0) with the two per-channel reductions BN backward needs,
//     and the per-channel constants that let the next kernel rebuild dy on load.
#include "wm_common.hpp"
using namespace wm;

namespace {

// sum partials[p][which][c] over p in double: 1024 threads = 64 channels x 16 slices, LDS tree at the end
__device__ __forceinline__ void reduce_partials_1024(const float* __restrict__ partials, int nparts, double& s1, double& s2,
                                                     double (*red)[2][64]) {
    const int c = threadIdx.x & 63, j = threadIdx.x >> 6;
    double a = 0.0, b = 0.0;
    for (int p = j; p < nparts; p += 16) {
        a += (double)partials[(size_t)p * 128 + c];
        b += (double)partials[(size_t)p * 128 + 64 + c];
    }
    red[j][0][c] = a;
    red[j][1][c] = b;
    __syncthreads();
    s1 = 0.0; s2 = 0.0;
    if (j == 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) { s1 += red[k][0][c]; s2 += red[k][1][c]; }
    }
}

// partials: [nparts][2][64] (sum, sum of squares) ; one 1024-thread block, threads 0..63 finish
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ partials, int nparts, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* running_mean, float* running_var, long long* num_batches_tracked,
                                   float momentum, float eps, float* scale, float* shift, float* save_mean,
                                   float* save_invstd) {
    __shared__ double red[16][2][64];
    double s1, s2;
    reduce_partials_1024(partials, nparts, s1, s2, red);
    if (threadIdx.x >= 64) return;
    const int c = threadIdx.x;
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const float sc = (float)((double)gamma[c] * invstd);
    scale[c] = sc;
    shift[c] = (float)((double)beta[c] - mean * (double)gamma[c] * invstd);
    save_mean[c] = (float)mean;
    save_invstd[c] = (float)invstd;
    if (running_mean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
        if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
    }
}

__global__ void bn_eval_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                               float* scale, float* shift) {
    const int c = threadIdx.x;
    const float inv = 1.0f / sqrtf(rv[c] + eps);
    scale[c] = gamma[c] * inv;
    shift[c] = beta[c] - rm[c] * gamma[c] * inv;
}

#ifndef WM_BNAR_UNROLL
#define WM_BNAR_UNROLL 4
#endif
#ifndef WM_STREAM_NT
#define WM_STREAM_NT 1      // nontemporal loads / stores on the frame streams: 0.59 -> 0.57 ms (bn_add_relu), 0.34 -> 0.32 ms (sums pass) at B = 256
#endif
// frame-sized operands that are read / written once per launch (1 GB per frame at B = 256: far beyond L2 + Infinity Cache)
__device__ __forceinline__ float4 stream_load(const float4* p) {
#if WM_STREAM_NT
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    const f32x4_ v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ void stream_store(float4* p, const float4& v) {
#if WM_STREAM_NT
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(f32x4_{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4_*>(p));
#else
    *p = v;
#endif
}
// out = relu(x + y*scale[c] + shift[c]);  grid (rows = B*64), float4 over T
// MASK: also record sign(out) as one bit per element -- the only thing the backward needs from `out`.  Layout: natural bit
// order, mask[row][t / 32] bit (t % 32), ceil(T / 32) dwords per (clip, channel) row -- any consumer finds the bits of a run of
// time steps of one row in one dword (the fused convolution-backward kernels apply the mask while they load the gradient, in
// their own thread -> element maps).  The lane's four bits are merged with its seven neighbours' by three lane exchanges.
template <bool MASK>
__global__ __launch_bounds__(256) void bn_add_relu_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          float* __restrict__ out, unsigned* __restrict__ mask, int T4) {
    const int row = blockIdx.x, c = row & 63;
    const float sc = scale[c], sh = shift[c];
    const float4* xr = reinterpret_cast<const float4*>(x) + (size_t)row * T4;
    const float4* yr = reinterpret_cast<const float4*>(y) + (size_t)row * T4;
    float4* orow = reinterpret_cast<float4*>(out) + (size_t)row * T4;
    const int lane = threadIdx.x & 63, nw = (T4 + 7) >> 3;
    unsigned* mrow = MASK ? mask + (size_t)row * nw : nullptr;
    constexpr int U = WM_BNAR_UNROLL;               // float4 pairs in flight per thread: all loads of an iteration are issued first
    for (int base = 0; base < T4; base += 256 * U) { // every thread runs every iteration: the lane exchanges need all 64 lanes
        float4 a[U], b[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int i = base + k * 256 + (int)threadIdx.x;
            if (i < T4) { a[k] = stream_load(xr + i); b[k] = stream_load(yr + i); }
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int i = base + k * 256 + (int)threadIdx.x;
            const bool ok = i < T4;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) {
                o.x = fmaxf(a[k].x + fmaf(b[k].x, sc, sh), 0.f);
                o.y = fmaxf(a[k].y + fmaf(b[k].y, sc, sh), 0.f);
                o.z = fmaxf(a[k].z + fmaf(b[k].z, sc, sh), 0.f);
                o.w = fmaxf(a[k].w + fmaf(b[k].w, sc, sh), 0.f);
                stream_store(orow + i, o);
            }
            if (MASK) {
                unsigned m = ((o.x > 0.f) ? 1u : 0u) | ((o.y > 0.f) ? 2u : 0u) | ((o.z > 0.f) ? 4u : 0u) | ((o.w > 0.f) ? 8u : 0u);
                m <<= 4 * (lane & 7);
                m |= __shfl_xor(m, 1); m |= __shfl_xor(m, 2); m |= __shfl_xor(m, 4);
                if ((lane & 7) == 0 && ok) mrow[i >> 3] = m;
            }
        }
    }
}

// dz = g * (out > 0);  partial[b][0][c] = sum dz, partial[b][1][c] = sum dz*y   (row = b*64+c).  MASK: the sign bits written by
// bn_add_relu_kernel<true> stand in for `out` (a 16.4 MB frame pass per clip less).  WRITE = false: the sums only -- the fused
// convolution-backward kernels re-apply the mask to g while loading it, so dz is never materialised (another frame pass less).
template <bool MASK, bool WRITE>
__global__ __launch_bounds__(256) void relu_bwd_reduce_kernel(const float* __restrict__ g, const float* __restrict__ out,
                                                              const unsigned* __restrict__ mask,
                                                              const float* __restrict__ y, float* __restrict__ dz,
                                                              float* __restrict__ partial, float* __restrict__ dzmax, int T4) {
    __shared__ float scratch[8];
    const int row = blockIdx.x, b = row >> 6, c = row & 63;
    const float4* gr = reinterpret_cast<const float4*>(g) + (size_t)row * T4;
    const float4* orow = MASK ? nullptr : reinterpret_cast<const float4*>(out) + (size_t)row * T4;
    const float4* yr = reinterpret_cast<const float4*>(y) + (size_t)row * T4;
    float4* dr = WRITE ? reinterpret_cast<float4*>(dz) + (size_t)row * T4 : nullptr;
    const unsigned* mrow = MASK ? mask + (size_t)row * ((T4 + 7) >> 3) : nullptr;
    float s1 = 0.f, s2 = 0.f, mx = 0.f;
    constexpr int U = WM_BNAR_UNROLL;
    for (int base = 0; base < T4; base += 256 * U) {
        float4 gq[U], yq[U], oq[MASK ? 1 : U];
        unsigned mq[MASK ? U : 1];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int i = base + k * 256 + (int)threadIdx.x;
            if (i < T4) {
                gq[k] = stream_load(gr + i); yq[k] = stream_load(yr + i);
                if (MASK) mq[k] = mrow[i >> 3]; else oq[k] = stream_load(orow + i);
            }
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int i = base + k * 256 + (int)threadIdx.x;
            if (i >= T4) continue;
            const float4 gg = gq[k], yy = yq[k];
            bool p0, p1, p2, p3;
            if (MASK) {
                const unsigned m = mq[k] >> (4 * (i & 7));
                p0 = m & 1u; p1 = m & 2u; p2 = m & 4u; p3 = m & 8u;
            } else {
                const float4 oo = oq[k];
                p0 = oo.x > 0.f; p1 = oo.y > 0.f; p2 = oo.z > 0.f; p3 = oo.w > 0.f;
            }
            float4 d;
            d.x = p0 ? gg.x : 0.f;
            d.y = p1 ? gg.y : 0.f;
            d.z = p2 ? gg.z : 0.f;
            d.w = p3 ? gg.w : 0.f;
            if (WRITE) stream_store(dr + i, d);
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w))));
            s1 += (d.x + d.y) + (d.z + d.w);
            s2 += fmaf(d.x, yy.x, d.y * yy.y) + fmaf(d.z, yy.z, d.w * yy.w);
        }
    }
    s1 = block_sum<4>(s1, scratch);
    s2 = block_sum<4>(s2, scratch + 4);
    if (dzmax) {                                     // max |dz| of the row: sizes the gradient's scale for the f16-split kernels
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        __syncthreads();
        if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0) dzmax[row] = fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
    }
    if (threadIdx.x == 0) {
        partial[(size_t)b * 128 + c] = s1;
        partial[(size_t)b * 128 + 64 + c] = s2;
    }
}

// partials [nparts][2][64] = (sum dz, sum dz*y_raw)  ->  dy = A*dz + Bc + Cc*y ; dgamma ; dbeta
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ partials, int nparts, double count,
                                       const float* __restrict__ gamma, const float* __restrict__ save_mean,
                                       const float* __restrict__ save_invstd, float* A, float* Bc, float* Cc,
                                       float* dgamma, float* dbeta, int accumulate, int eval_mode,
                                       const float* __restrict__ dzmax, int nmax, float* gscale) {
    __shared__ double red[16][2][64];
    __shared__ float mred[1024 / 64 + 64];
    double s1, s2;
    reduce_partials_1024(partials, nparts, s1, s2, red);
    if (gscale) {
        // max |dz| over the tensor (one entry per producer block): with max |A| it sizes the power-of-two scale that maps the rebuilt
        // gradient g = A dz + B + C y into the f16 range for the two-piece split (wm_dwgrad64_bf, arith 1)
        float mx = 0.f;
        for (int i = threadIdx.x; i < nmax; i += 1024) mx = fmaxf(mx, dzmax[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if ((threadIdx.x & 63) == 0) mred[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const int c = threadIdx.x;
    const double mu = save_mean[c], is = save_invstd[c], ga = gamma[c];
    const double sxh = (s2 - mu * s1) * is;            // sum dz * yhat
    // dy = A*dz + Cc*y + B must sum to zero over the batch (that is what BatchNorm backward guarantees); with fp32
    // constants the rounding of A, Cc and B would leave a *systematic* residue N*eps that lands in every upstream
    // "sum" gradient (biases, betas).  So B is derived from the ROUNDED A and Cc and handed over as hi + lo words.
    const float Af = (float)(ga * is);
    const float Cf = eval_mode ? 0.f : (float)(-ga * is * is * sxh / count);     // eval: statistics are constants
    const double Bd = eval_mode ? 0.0 : -((double)Af * s1 + (double)Cf * (mu * count)) / count;
    const float Bh = (float)Bd;
    A[c] = Af;
    Cc[c] = Cf;
    Bc[c] = Bh;
    Bc[64 + c] = (float)(Bd - (double)Bh);
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)sxh : (float)sxh;
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)s1 : (float)s1;
    if (gscale) {                                    // threads 0..63 = one wave: max |A| over channels, max |dz| over the blocks
        float amax = fabsf(Af), mx = 0.f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, mred[i]);
        if (c == 0) {
            // A dz lands at <= 2^9 = 512: 128 x below the f16 maximum for the B + C y part (the kernel clamps at +-6e4 on top, so a
            // pathological spike saturates one element instead of producing an infinity), tiny elements keep 2^-33 of the maximum
            const float big = amax * mx;
            float gs = 1.f;
            if (big > 0.f && big < 3.0e38f) gs = exp2f(floorf(log2f(512.f / big)));
            gs = fminf(fmaxf(gs, 1.0e-30f), 1.0e30f);
            gscale[0] = gs;
            gscale[1] = 1.f / gs;
        }
    }
}


// max |x| over a tensor -> the power-of-two scale of the f16 two-piece split kernels: {gs, 1 / gs} with max |x| gs in (2^(L-1), 2^L].
// Pass 1: one partial per workgroup (streaming, 1024 workgroups); pass 2: one workgroup.
__device__ unsigned g_absmax_ticket = 0;          // arrivals of the current wm_gscale_absmax launch (one launch at a time per device)
__global__ __launch_bounds__(256) void absmax_kernel(const float4* __restrict__ x, size_t n4, float* __restrict__ partial,
                                                     float log2_target, float* __restrict__ gscale) {
    __shared__ float red[4];
    __shared__ unsigned last;
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = stream_load(x + i);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        __threadfence();
        last = (atomicAdd(&g_absmax_ticket, 1u) == gridDim.x - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (!last) return;
    // the workgroup that arrives last reduces the partials and writes the scale (no second launch)
    __threadfence();
    float mm = 0.f;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) mm = fmaxf(mm, __builtin_nontemporal_load(partial + i));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mm = fmaxf(mm, __shfl_xor(mm, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mm;
    __syncthreads();
    if (threadIdx.x == 0) {
        mm = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        float gs = 1.f;
        if (mm > 0.f && mm < 3.0e38f) gs = exp2f(floorf(log2_target - log2f(mm)));      // NaN / inf / all-zero input: scale 1
        gs = fminf(fmaxf(gs, 1.0e-30f), 1.0e30f);
        gscale[0] = gs; gscale[1] = 1.f / gs;
        g_absmax_ticket = 0;
    }
}
__global__ __launch_bounds__(256) void gscale_from_max_kernel(const float* __restrict__ partial, int n, float log2_target,
                                                              float* __restrict__ gscale) {
    __shared__ float red[4];
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, partial[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        float gs = 1.f;
        if (m > 0.f && m < 3.0e38f) gs = exp2f(floorf(log2_target - log2f(m)));      // NaN / inf / all-zero input: scale 1
        gs = fminf(fmaxf(gs, 1.0e-30f), 1.0e30f);
        gscale[0] = gs; gscale[1] = 1.f / gs;
    }
}

}  // namespace

extern "C" {

// Train-mode BatchNorm finalize (py/main16.py:117,120 in .train()): batch mean / biased var for
// normalisation, running stats with momentum and unbiased var, num_batches_tracked += 1.
int wm_bn_finalize(const float* partials, int nparts, double count, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps,
                   float* scale, float* shift, float* save_mean, float* save_invstd, hipStream_t stream) {
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(1024), 0, stream, partials, nparts, count, gamma, beta,
                       running_mean, running_var, num_batches_tracked, momentum, eps, scale, shift, save_mean, save_invstd);
    WM_CHECK_LAUNCH();
    return 0;
}

// Eval-mode BatchNorm folded to per-channel (scale, shift) from the running statistics.
int wm_bn_eval_scale_shift(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                           float eps, float* scale, float* shift, hipStream_t stream) {
    hipLaunchKernelGGL(bn_eval_kernel, dim3(1), dim3(64), 0, stream, gamma, beta, running_mean, running_var, eps, scale, shift);
    WM_CHECK_LAUNCH();
    return 0;
}

// ResBlock tail: out = relu(x + y2*scale + shift)   (py/main16.py:125)
int wm_bn_add_relu(const float* x, const float* y2, const float* scale, const float* shift, float* out, int B, int T,
                   hipStream_t stream) {
    if (T & 3) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(bn_add_relu_kernel<false>, dim3(B * 64), dim3(256), 0, stream, x, y2, scale, shift, out, (unsigned*)nullptr, T / 4);
    WM_CHECK_LAUNCH();
    return 0;
}

// The same tail + the sign bits of `out`: mask holds ceil(T / 32) dwords per (clip, channel) row, bit (t % 32) of dword t / 32.
int wm_bn_add_relu_mask(const float* x, const float* y2, const float* scale, const float* shift, float* out, void* mask, int B, int T,
                        hipStream_t stream) {
    if ((T & 3) || !mask) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(bn_add_relu_kernel<true>, dim3(B * 64), dim3(256), 0, stream, x, y2, scale, shift, out,
                       reinterpret_cast<unsigned*>(mask), T / 4);
    WM_CHECK_LAUNCH();
    return 0;
}

// Backward of the tail: dz = g*(out>0) (also the residual-path gradient), partial[B][2][64].
int wm_relu_bwd_reduce(const float* g, const float* out, const float* y2, float* dz, float* partial, int B, int T,
                       hipStream_t stream) {
    if (T & 3) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL((relu_bwd_reduce_kernel<false, true>), dim3(B * 64), dim3(256), 0, stream, g, out, (const unsigned*)nullptr, y2, dz, partial,
                       (float*)nullptr, T / 4);
    WM_CHECK_LAUNCH();
    return 0;
}

// The same from the sign bits; dz == NULL: only the two sums (the consumer masks g itself: wm_dwgrad64_bf's gmask).
// dzmax (optional, B * 64 floats): max |dz| of every row, for wm_bn_bwd_finalize's gradient scale.
int wm_relu_bwd_reduce_mask(const float* g, const void* mask, const float* y2, float* dz, float* partial, float* dzmax, int B, int T,
                            hipStream_t stream) {
    if ((T & 3) || !mask) return (int)hipErrorInvalidValue;
    if (dz)
        hipLaunchKernelGGL((relu_bwd_reduce_kernel<true, true>), dim3(B * 64), dim3(256), 0, stream, g, (const float*)nullptr,
                           reinterpret_cast<const unsigned*>(mask), y2, dz, partial, dzmax, T / 4);
    else
        hipLaunchKernelGGL((relu_bwd_reduce_kernel<true, false>), dim3(B * 64), dim3(256), 0, stream, g, (const float*)nullptr,
                           reinterpret_cast<const unsigned*>(mask), y2, (float*)nullptr, partial, dzmax, T / 4);
    WM_CHECK_LAUNCH();
    return 0;
}

int wm_bn_bwd_finalize(const float* partials, int nparts, double count, const float* gamma, const float* save_mean,
                       const float* save_invstd, float* A, float* Bc, float* Cc, float* dgamma, float* dbeta,
                       int accumulate, int eval_mode, const float* dzmax, int nmax, float* gscale, hipStream_t stream) {
    if (gscale && (!dzmax || nmax <= 0)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(1), dim3(1024), 0, stream, partials, nparts, count, gamma, save_mean,
                       save_invstd, A, Bc, Cc, dgamma, dbeta, accumulate, eval_mode, dzmax, nmax, gscale);
    WM_CHECK_LAUNCH();
    return 0;
}


// the second half alone: maxes [n] = max |x| per producer workgroup (wm_dwgrad64_bf's dzmax output)
int wm_gscale_from_max(const float* maxes, int n, float log2_target, float* gscale, hipStream_t stream) {
    if (n <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(gscale_from_max_kernel, dim3(1), dim3(256), 0, stream, maxes, n, log2_target, gscale);
    WM_CHECK_LAUNCH();
    return 0;
}

// {gs, 1 / gs} for a gradient tensor that an f16 two-piece split kernel is about to read: gs = the power of two that puts max |x|
// into (2^(L-1), 2^L], L = log2_target.  scratch: >= 1024 floats.  n % 4 == 0.
int wm_gscale_absmax(const float* x, long long n, float* scratch, float log2_target, float* gscale, hipStream_t stream) {
    if (n <= 0 || (n & 3)) return (int)hipErrorInvalidValue;
    const size_t n4 = (size_t)n / 4;
    // eight float4 per thread and workgroup at least: a small tensor (a weight matrix) is a handful of workgroups, not a thousand
    const size_t want = (n4 + 2047) / 2048;
    const int grid = (int)(want < 1 ? 1 : (want < 1024 ? want : 1024));
    hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const float4*>(x), n4, scratch, log2_target, gscale);
    WM_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
