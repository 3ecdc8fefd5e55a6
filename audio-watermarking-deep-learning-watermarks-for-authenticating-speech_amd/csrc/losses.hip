// Point-wise loss terms of the training step (py/main16.py:252-266):
//   loc  = BCEWithLogits(logits[:, :, 0], [1 ... 1, 0 ... 0])            over 2B clips
//   bce  = BCEWithLogits(logits[:B, :, 1:], bits(message) broadcast over T)
//   l1   = mean |delta|
// Both BCE terms are produced by ONE pass over the (2B,T,1+bits) logits, and their gradient by one more.
#include "wm_common.hpp"
using namespace wm;

namespace {

// softplus(-|x|) = log1p(e^-|x|) on the hardware transcendentals: u = e^-|x| in (0, 1]; a cubic series below 0.01 (error
// < 3e-9 absolute), log(1 + u) above (the rounding of 1 + u costs <= 6e-6 relative there).  libm's log1pf/expf pair made
// the 139 M-element pass VALU-bound (0.7 ms at B = 256).
__device__ __forceinline__ float softplus_neg_abs(float x) {
    const float u = __expf(-fabsf(x));
    return (u < 0.01f) ? u * fmaf(u, fmaf(u, 0.33333334f, -0.5f), 1.0f) : __logf(1.0f + u);
}
__device__ __forceinline__ float bce_logits(float x, float y) { return fmaxf(x, 0.f) - x * y + softplus_neg_abs(x); }

// Both BCE kernels walk one clip's [T][NO] logits in flat, coalesced order: grid = (chunks of 4096 elements, clips).
// The channel o = idx % NO comes from a float reciprocal with a fix-up (idx < 2^23): the 64-bit div/mod per element of
// the first version made these kernels 6x slower than the memory system allows.
__device__ __forceinline__ int mod_small(int idx, int n, float inv_n) {
    int q = (int)((float)idx * inv_n);
    int r = idx - q * n;
    if (r < 0) r += n;
    if (r >= n) r -= n;
    return r;
}

// partial[0][blk] = sum of detection terms, partial[1][blk] = sum of bit terms (blk = flat block index)
__global__ __launch_bounds__(256) void bce_fwd_kernel(const float* __restrict__ logits, const long long* __restrict__ message,
                                                      int B, int R, int T, int NO, float* __restrict__ partial) {
    __shared__ float scratch[8];
    const int r = blockIdx.y, per_clip = T * NO;
    const float inv = 1.0f / (float)NO;
    const long long msg = (r < B) ? message[r] : 0;
    const float yl = r < B ? 1.f : 0.f;
    const float* lp = logits + (size_t)r * per_clip;
    float sl = 0.f, sb = 0.f;
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
        const int idx = blockIdx.x * 4096 + k * 256 + threadIdx.x;
        if (idx < per_clip) {
            const int o = mod_small(idx, NO, inv);
            const float x = lp[idx];
            if (o == 0) sl += bce_logits(x, yl);
            else if (r < B) sb += bce_logits(x, (float)((msg >> (o - 1)) & 1));
        }
    }
    sl = block_sum<4>(sl, scratch);
    sb = block_sum<4>(sb, scratch + 4);
    const int blk = blockIdx.y * gridDim.x + blockIdx.x, nblk = gridDim.x * gridDim.y;
    if (threadIdx.x == 0) { partial[blk] = sl; partial[nblk + blk] = sb; }
}

__global__ __launch_bounds__(256) void bce_bwd_kernel(const float* __restrict__ logits, const long long* __restrict__ message,
                                                      const float* __restrict__ g_loc, const float* __restrict__ g_bce,
                                                      int B, int R, int T, int NO, float* __restrict__ dlogits) {
    const int r = blockIdx.y, per_clip = T * NO;
    const float inv = 1.0f / (float)NO;
    const float kl = g_loc[0] / (float)((double)R * T);
    const float kb = (NO > 1) ? g_bce[0] / (float)((double)B * T * (NO - 1)) : 0.f;
    const long long msg = (r < B) ? message[r] : 0;
    const float yl = r < B ? 1.f : 0.f;
    const float* lp = logits + (size_t)r * per_clip;
    float* dp = dlogits + (size_t)r * per_clip;
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
        const int idx = blockIdx.x * 4096 + k * 256 + threadIdx.x;
        if (idx < per_clip) {
            const int o = mod_small(idx, NO, inv);
            const float x = lp[idx];
            const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-x));     // as in the LSTM gates: error <= 3e-8
            float d;
            if (o == 0) d = kl * (sg - yl);
            else d = (r < B) ? kb * (sg - (float)((msg >> (o - 1)) & 1)) : 0.f;
            dp[idx] = d;
        }
    }
}

__global__ __launch_bounds__(256) void abs_sum_kernel(const float* __restrict__ x, size_t n, float* __restrict__ partial) {
    __shared__ float scratch[4];
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += fabsf(x[i]);
    s = block_sum<4>(s, scratch);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ void l1_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, size_t n, float* __restrict__ dx) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float k = g[0] / (float)n, v = x[i];
    dx[i] = v > 0.f ? k : (v < 0.f ? -k : 0.f);
}

__global__ __launch_bounds__(256) void sum_scale2_kernel(const float* __restrict__ partial, int n, double scale, float* out) {
    __shared__ double scratch[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)partial[i];
    s = block_sum_d<4>(s, scratch);
    if (threadIdx.x == 0) out[0] = (float)(s * scale);
}

// Adam over one flat fp32 span (torch.optim.Adam defaults semantics, py/main16.py:504): no amsgrad,
// weight_decay 0, bias correction from the step count.
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            size_t n, float step_size, float b1, float b2, float eps, float bc2_sqrt) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = fmaf(b1, m[i], (1.f - b1) * gi);
    const float vi = fmaf(b2, v[i], (1.f - b2) * gi * gi);
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= step_size * (mi / denom);
}

}  // namespace

extern "C" {

// logits [R=2B,T,NO] (NO = 1 + bits); message [B] int64; partial: >= 2 * R * ceil(T*NO/4096) floats scratch.
int wm_bce_fwd(const float* logits, const long long* message, float* partial, float* loc_out, float* bce_out, int B, int R,
               int T, int NO, hipStream_t stream) {
    if ((long long)T * NO >= (1 << 23) || R <= 0 || R > 65535) return (int)hipErrorInvalidValue;
    const int chunks = (T * NO + 4095) / 4096, grid = chunks * R;
    hipLaunchKernelGGL(bce_fwd_kernel, dim3(chunks, R), dim3(256), 0, stream, logits, message, B, R, T, NO, partial);
    WM_CHECK_LAUNCH();
    hipLaunchKernelGGL(sum_scale2_kernel, dim3(1), dim3(256), 0, stream, (const float*)partial, grid, 1.0 / ((double)R * T), loc_out);
    WM_CHECK_LAUNCH();
    if (NO > 1) {
        hipLaunchKernelGGL(sum_scale2_kernel, dim3(1), dim3(256), 0, stream, (const float*)partial + grid, grid,
                           1.0 / ((double)B * T * (NO - 1)), bce_out);
        WM_CHECK_LAUNCH();
    }
    return 0;
}

// g_loc / g_bce: device scalars holding d(total)/d(loc), d(total)/d(bce)
int wm_bce_bwd(const float* logits, const long long* message, const float* g_loc, const float* g_bce, float* dlogits, int B,
               int R, int T, int NO, hipStream_t stream) {
    if ((long long)T * NO >= (1 << 23) || R <= 0 || R > 65535) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(bce_bwd_kernel, dim3((T * NO + 4095) / 4096, R), dim3(256), 0, stream, logits, message, g_loc, g_bce, B, R, T,
                       NO, dlogits);
    WM_CHECK_LAUNCH();
    return 0;
}

int wm_l1_fwd(const float* x, float* partial, float* out, long long n, hipStream_t stream) {
    const int grid = 256;
    hipLaunchKernelGGL(abs_sum_kernel, dim3(grid), dim3(256), 0, stream, x, (size_t)n, partial);
    WM_CHECK_LAUNCH();
    hipLaunchKernelGGL(sum_scale2_kernel, dim3(1), dim3(256), 0, stream, (const float*)partial, grid, 1.0 / (double)n, out);
    WM_CHECK_LAUNCH();
    return 0;
}

int wm_l1_bwd(const float* x, const float* g, float* dx, long long n, hipStream_t stream) {
    hipLaunchKernelGGL(l1_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, g, (size_t)n, dx);
    WM_CHECK_LAUNCH();
    return 0;
}

// one Adam update over a flat span; step >= 1 is the 1-based update count
// (the bias corrections are formed in double on the host, as torch.optim.Adam does, and handed over rounded once)
int wm_adam_step(float* p, const float* g, float* m, float* v, long long n, double lr, double beta1, double beta2, double eps,
                 int step, hipStream_t stream) {
    if (step < 1 || n < 0) return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2s = sqrt(1.0 - pow(beta2, (double)step));
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p, g, m, v, (size_t)n,
                       (float)(lr / bc1), (float)beta1, (float)beta2, (float)eps, (float)bc2s);
    WM_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
