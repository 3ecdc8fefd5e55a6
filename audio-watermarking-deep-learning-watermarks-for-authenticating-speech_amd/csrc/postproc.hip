// delta post-processing between Generator and Detector (py/main16.py:53-72, applied at :245-247):
//   fir_lowpass (101-tap FIR, 'same')  ->  clamp_peak (+-0.02)  ->  limit_rms (per-clip RMS cap)
// One 256-thread workgroup per clip keeps the whole 64 KB clip (plus the FIR halo) in LDS, so the
// three stages and the per-clip reduction cost one HBM read and one write.  `stages` is a bit mask
// (1 fir, 2 clamp, 4 rms) so the three reference functions are also available one by one.
// Also here: the embedding row gather / gradient scatter and per-row sums used around decoder[0].
#include "wm_common.hpp"
using namespace wm;

namespace {

constexpr int kMaxTaps = 128;

// f_out: the FIR output (input of the clamp), saved for backward (may be nullptr)
// stats[b] = {cur_rms, gain}
__global__ __launch_bounds__(256) void postproc_fwd_kernel(const float* __restrict__ d_in, const float* __restrict__ taps,
                                                           int ntaps, float thr, float max_rms, float eps, int stages,
                                                           float* __restrict__ f_out, float* __restrict__ d_out,
                                                           float* __restrict__ stats, int T) {
    extern __shared__ __align__(16) float smem[];
    float* xs = smem;                       // [T + ntaps - 1]
    float* ks = smem + T + kMaxTaps;        // [ntaps]
    __shared__ float scratch[8];
    const int b = blockIdx.x, tid = threadIdx.x, pad = (ntaps - 1) / 2;
    const float* x = d_in + (size_t)b * T;
    const bool do_fir = stages & 1;
    for (int i = tid; i < T + 2 * pad; i += 256) {
        const int t = i - pad;
        xs[i] = (t >= 0 && t < T) ? x[t] : 0.f;
    }
    if (tid < ntaps) ks[tid] = taps[tid];
    __syncthreads();
    float ss = 0.f;
    // pass 1: FIR + clamp, keep the clamped value in registers-free form by rewriting LDS after a barrier
    // (outputs of pass 1 go to global f_out; the clamped values are re-derived in pass 2)
    for (int t = tid; t < T; t += 256) {
        float f;
        if (do_fir) {
            f = 0.f;
            for (int j = 0; j < ntaps; ++j) f = fmaf(ks[j], xs[t + j], f);
        } else {
            f = xs[t + pad];
        }
        if (f_out) f_out[(size_t)b * T + t] = f;
        const float cl = (stages & 2) ? fminf(fmaxf(f, -thr), thr) : f;
        d_out[(size_t)b * T + t] = cl;       // provisional (un-scaled); rescaled below when the RMS cap bites
        ss = fmaf(cl, cl, ss);
    }
    if (stages & 4) {
        const float tot = block_sum<4>(ss, scratch);
        const float cur = sqrtf(tot / (float)T + eps);
        const float gain = fminf(max_rms / cur, 1.0f);
        if (tid == 0 && stats) { stats[2 * b] = cur; stats[2 * b + 1] = gain; }
        // each thread rescales exactly the elements it wrote itself -> no fence needed
        for (int t = tid; t < T; t += 256) d_out[(size_t)b * T + t] *= gain;
    } else if (tid == 0 && stats) {
        stats[2 * b] = 0.f; stats[2 * b + 1] = 1.f;
    }
}

// g: gradient w.r.t. the post-processed delta.  f: saved FIR output.  Returns gradient w.r.t. the raw delta.
__global__ __launch_bounds__(256) void postproc_bwd_kernel(const float* __restrict__ g, const float* __restrict__ f_in,
                                                           const float* __restrict__ stats, const float* __restrict__ taps,
                                                           int ntaps, float thr, float max_rms, int stages,
                                                           float* __restrict__ d_raw, int T) {
    extern __shared__ __align__(16) float smem[];
    float* gs = smem;                       // [T + ntaps - 1] gradient at the FIR output, zero halo
    float* ks = smem + T + kMaxTaps;
    __shared__ float scratch[8];
    const int b = blockIdx.x, tid = threadIdx.x, pad = (ntaps - 1) / 2;
    const float* gb = g + (size_t)b * T;
    const float* fb = f_in + (size_t)b * T;
    if (tid < ntaps) ks[tid] = taps[tid];
    for (int i = tid; i < pad; i += 256) { gs[i] = 0.f; gs[pad + T + i] = 0.f; }
    float cur = 0.f, gain = 1.f;
    if (stages & 4) { cur = stats[2 * b]; gain = stats[2 * b + 1]; }
    // sum g*cl for the RMS-cap branch
    float dot = 0.f;
    if ((stages & 4) && gain < 1.0f) {
        float acc = 0.f;
        for (int t = tid; t < T; t += 256) {
            const float f = fb[t];
            const float cl = (stages & 2) ? fminf(fmaxf(f, -thr), thr) : f;
            acc = fmaf(gb[t], cl, acc);
        }
        dot = block_sum<4>(acc, scratch);
    }
    const float r = max_rms / cur;
    const float kk = dot / ((float)T * cur * cur);
    for (int t = tid; t < T; t += 256) {
        const float f = fb[t];
        const float cl = (stages & 2) ? fminf(fmaxf(f, -thr), thr) : f;
        float dd = gb[t];
        if ((stages & 4) && gain < 1.0f) dd = r * (dd - cl * kk);
        if ((stages & 2) && !(f >= -thr && f <= thr)) dd = 0.f;
        gs[pad + t] = dd;
    }
    __syncthreads();
    for (int t = tid; t < T; t += 256) {
        float acc;
        if (stages & 1) {
            acc = 0.f;                      // d_raw[t] = sum_j k[j] * dd[t - j + pad]
            for (int j = 0; j < ntaps; ++j) acc = fmaf(ks[j], gs[t + 2 * pad - j], acc);
        } else {
            acc = gs[pad + t];
        }
        d_raw[(size_t)b * T + t] = acc;
    }
}

__global__ void embed_gather_kernel(const float* __restrict__ table, const long long* __restrict__ message,
                                    float* __restrict__ vec, int B, int nrows, int* __restrict__ err) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 64) return;
    const long long m = message[i >> 6];
    if (m < 0 || m >= nrows) { if (err) *err = 1; vec[i] = 0.f; return; }
    vec[i] = table[(size_t)m * 64 + (i & 63)];
}

__global__ void embed_scatter_add_kernel(float* __restrict__ dtable, const long long* __restrict__ message,
                                         const float* __restrict__ dvec, int B, int nrows) {
    // one thread per embedding column walks the batch in order: duplicate message ids add in a fixed order
    // (bit-reproducible, no float atomics; B x 64 values, the cost is nil)
    const int d = threadIdx.x;
    if (d >= 64) return;
    for (int b = 0; b < B; ++b) {
        const long long m = message[b];
        if (m < 0 || m >= nrows) continue;
        dtable[(size_t)m * 64 + d] += dvec[(size_t)b * 64 + d];
    }
}

// out[row] = sum_t x[row, t]
__global__ __launch_bounds__(256) void rowsum_kernel(const float* __restrict__ x, float* __restrict__ out, int T4) {
    __shared__ float scratch[4];
    const float4* xr = reinterpret_cast<const float4*>(x) + (size_t)blockIdx.x * T4;
    float s = 0.f;
    for (int i = threadIdx.x; i < T4; i += 256) { const float4 v = xr[i]; s += (v.x + v.y) + (v.z + v.w); }
    s = block_sum<4>(s, scratch);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}

}  // namespace

extern "C" {

// stages: bit0 fir_lowpass (py/main16.py:53-64), bit1 clamp_peak (:66-67), bit2 limit_rms (:69-72).
// d_in/d_out [B,1,T]; f_out [B,1,T] or NULL; stats [B,2] = (rms before the cap, gain) or NULL.
int wm_postproc_fwd(const float* d_in, const float* taps, int ntaps, float thr, float max_rms, float eps, int stages,
                    float* f_out, float* d_out, float* stats, int B, int T, hipStream_t stream) {
    if (ntaps < 1 || ntaps > kMaxTaps || !(ntaps & 1) || T > 36000) return (int)hipErrorInvalidValue;
    const size_t lds = (size_t)(T + 2 * kMaxTaps) * sizeof(float);
    static size_t lds_set = 0;
    if (lds > lds_set) {
        WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(postproc_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set = lds;
    }
    hipLaunchKernelGGL(postproc_fwd_kernel, dim3(B), dim3(256), lds, stream, d_in, taps, ntaps, thr, max_rms, eps, stages,
                       f_out, d_out, stats, T);
    WM_CHECK_LAUNCH();
    return 0;
}

int wm_postproc_bwd(const float* g, const float* f_in, const float* stats, const float* taps, int ntaps, float thr,
                    float max_rms, int stages, float* d_raw, int B, int T, hipStream_t stream) {
    if (ntaps < 1 || ntaps > kMaxTaps || !(ntaps & 1) || T > 36000) return (int)hipErrorInvalidValue;
    const size_t lds = (size_t)(T + 2 * kMaxTaps) * sizeof(float);
    static size_t lds_set = 0;
    if (lds > lds_set) {
        WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(postproc_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set = lds;
    }
    hipLaunchKernelGGL(postproc_bwd_kernel, dim3(B), dim3(256), lds, stream, g, f_in, stats, taps, ntaps, thr, max_rms,
                       stages, d_raw, T);
    WM_CHECK_LAUNCH();
    return 0;
}

// vec[b,:] = table[message[b],:]   (nn.Embedding(2**bits, 64) lookup, py/main16.py:158); err: device int flag
int wm_embed_gather(const float* table, const long long* message, float* vec, int B, int nrows, int* err, hipStream_t stream) {
    hipLaunchKernelGGL(embed_gather_kernel, dim3((B * 64 + 255) / 256), dim3(256), 0, stream, table, message, vec, B, nrows, err);
    WM_CHECK_LAUNCH();
    return 0;
}

// dtable[message[b],:] += dvec[b,:]   (dense embedding gradient; dtable must be pre-zeroed by the caller)
int wm_embed_scatter_add(float* dtable, const long long* message, const float* dvec, int B, int nrows, hipStream_t stream) {
    hipLaunchKernelGGL(embed_scatter_add_kernel, dim3(1), dim3(64), 0, stream, dtable, message, dvec, B, nrows);
    WM_CHECK_LAUNCH();
    return 0;
}

int wm_rowsum(const float* x, float* out, int rows, int T, hipStream_t stream) {
    if (T & 3) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(rowsum_kernel, dim3(rows), dim3(256), 0, stream, x, out, T / 4);
    WM_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
