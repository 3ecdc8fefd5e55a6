// The STFT loss stack of py/main16.py as fused framing + window + rFFT + spectral-loss kernels:
//   MEL   MultiScaleMelLoss :192-202   n_fft 1024 hop 256, |.|^2 -> 64 HTK mel bands -> L1 of log(.+1e-5)
//   LOUD  TFLoudnessLoss    :204-217   n_fft 2048 hop 512, masked squared magnitude difference
//   HF    high_freq_penalty :74-81     n_fft  512 hop 128, mean magnitude above 3.5 kHz
// (torch.stft defaults: periodic Hann, center=True with reflect padding, onesided, unnormalised.)
//
// One 256-thread workgroup owns one complex FFT in LDS (radix-2 DIF, twiddles from sincospi) and gets TWO
// real spectra out of it: clean + watermarked frame (MEL, LOUD) or two consecutive frames (HF).  The same
// workgroup then evaluates the loss term AND its gradient w.r.t. the time-domain frame (inverse transform of
// the Hermitian-completed spectral gradient, again two-for-one), so forward + backward of a loss is one
// kernel plus a deterministic overlap-add gather -- no spectrogram ever reaches HBM.
#include "wm_common.hpp"
using namespace wm;

namespace {

enum { MODE_MEL = 0, MODE_LOUD = 1, MODE_HF = 2 };

struct StftArgs {
    const float* a;      // [B,T]  MEL/LOUD: clean signal; HF: delta
    const float* b;      // [B,T]  MEL/LOUD: watermarked signal; HF: unused
    float* gframes;      // [B,F,N] windowed time-domain gradient per frame (w.r.t. b, or a for HF); may be NULL
    float* partial;      // [B*units] un-normalised loss sums
    const float* fb;     // MEL: [N/2+1][64] filterbank
    const int* klo; const int* khi;   // MEL: per-mel inclusive bin range
    const int* mlo;      // MEL: per-bin first mel with non-zero weight
    float gscale;        // d(loss)/d(sum) = 1/count
    float thresh;        // LOUD: mask threshold 0.01
    int kcut;            // HF: first penalised bin
    int T, F, hop;
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

template <int LOGN>
__device__ __forceinline__ void fft_dif(float2* z, const float2* tw, int tid) {
    constexpr int N = 1 << LOGN;
#pragma unroll 1
    for (int st = 0; st < LOGN; ++st) {
        const int lm = LOGN - 1 - st, m = 1 << lm;           // half size of this stage's butterflies
        for (int j = tid; j < N / 2; j += 256) {
            const int pos = j & (m - 1), i0 = ((j >> lm) << (lm + 1)) + pos, i1 = i0 + m;
            const float2 a = z[i0], b = z[i1], w = tw[pos << st];
            z[i0] = make_float2(a.x + b.x, a.y + b.y);
            z[i1] = cmul(make_float2(a.x - b.x, a.y - b.y), w);
        }
        __syncthreads();
    }
}
template <int LOGN>
__device__ __forceinline__ int brev(int k) { return (int)(__brev((unsigned)k) >> (32 - LOGN)); }

__device__ __forceinline__ int reflect_idx(int m, int T) {
    if (m < 0) m = -m;
    if (m >= T) m = 2 * (T - 1) - m;
    return m;
}

template <int LOGN, int MODE>
__global__ __launch_bounds__(256) void stft_loss_kernel(StftArgs p) {
    constexpr int N = 1 << LOGN, NB = N / 2 + 1;
    extern __shared__ __align__(16) float smem[];
    float2* z = reinterpret_cast<float2*>(smem);                 // [N]
    float2* tw = z + N;                                          // [N/2]
    float2* GA = tw + N / 2;                                     // [NB]
    float2* GB = GA + NB;                                        // [NB]
    float* pw = reinterpret_cast<float*>(GB + NB);               // MEL: [2][NB] power, then [2][64] mel, [64] dmel
    __shared__ float scratch[8];
    const int tid = threadIdx.x;
    const int units = (MODE == MODE_HF) ? (p.F + 1) / 2 : p.F;
    const int bclip = blockIdx.x / units, unit = blockIdx.x % units;
    const int fa = (MODE == MODE_HF) ? 2 * unit : unit;          // frame index of the 'a' lane
    const int fbi = (MODE == MODE_HF) ? 2 * unit + 1 : unit;     // frame index of the 'b' lane
    const bool b_valid = (MODE != MODE_HF) || (fbi < p.F);
    const float* sa = p.a + (size_t)bclip * p.T;
    const float* sb = (MODE == MODE_HF) ? sa : p.b + (size_t)bclip * p.T;

    for (int k = tid; k < N / 2; k += 256) {
        float s, c;
        sincospif(-2.0f * (float)k / (float)N, &s, &c);
        tw[k] = make_float2(c, s);
    }
    for (int n = tid; n < N; n += 256) {
        const float w = 0.5f - 0.5f * cospif(2.0f * (float)n / (float)N);
        const float va = sa[reflect_idx(fa * p.hop + n - N / 2, p.T)];
        const float vb = b_valid ? sb[reflect_idx(fbi * p.hop + n - N / 2, p.T)] : 0.f;
        z[n] = make_float2(va * w, vb * w);
    }
    __syncthreads();
    fft_dif<LOGN>(z, tw, tid);

    float loss = 0.f;
    for (int k = tid; k < NB; k += 256) {
        const float2 zk = z[brev<LOGN>(k)], zn = z[brev<LOGN>((N - k) & (N - 1))];
        const float2 A = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
        const float2 Bv = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
        if (MODE == MODE_MEL) {
            pw[k] = A.x * A.x + A.y * A.y;
            pw[NB + k] = Bv.x * Bv.x + Bv.y * Bv.y;
            GB[k] = Bv;
            GA[k] = make_float2(0.f, 0.f);
        } else if (MODE == MODE_LOUD) {
            const float ma = sqrtf(A.x * A.x + A.y * A.y), mb = sqrtf(Bv.x * Bv.x + Bv.y * Bv.y);
            const bool mask = ma > p.thresh;
            const float diff = mb - ma;
            if (mask) loss = fmaf(diff, diff, loss);
            const float coef = (mask && mb > 0.f) ? 2.0f * diff * p.gscale / mb : 0.f;
            GB[k] = make_float2(coef * Bv.x, coef * Bv.y);
            GA[k] = make_float2(0.f, 0.f);
        } else {
            float2 ga = make_float2(0.f, 0.f), gb = ga;
            if (k >= p.kcut) {
                const float ma = sqrtf(A.x * A.x + A.y * A.y), mb = sqrtf(Bv.x * Bv.x + Bv.y * Bv.y);
                loss += ma + mb;
                if (ma > 0.f) ga = make_float2(p.gscale * A.x / ma, p.gscale * A.y / ma);
                if (mb > 0.f) gb = make_float2(p.gscale * Bv.x / mb, p.gscale * Bv.y / mb);
            }
            GA[k] = ga;
            GB[k] = gb;
        }
    }
    if (MODE == MODE_MEL) {
        float* mel = pw + 2 * NB;       // [2][64]
        float* dm = mel + 128;          // [64]
        __syncthreads();
        if (tid < 128) {
            const int sig = tid >> 6, m = tid & 63;
            const float* pp = pw + sig * NB;
            float acc = 0.f;
            for (int k = p.klo[m]; k <= p.khi[m]; ++k) acc = fmaf(p.fb[k * 64 + m], pp[k], acc);
            mel[tid] = acc;
        }
        __syncthreads();
        if (tid < 64) {
            const float la = logf(mel[tid] + 1e-5f), lb = logf(mel[64 + tid] + 1e-5f);
            const float d = la - lb;
            loss = fabsf(d);
            const float sg = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
            dm[tid] = -sg * p.gscale / (mel[64 + tid] + 1e-5f);
        }
        __syncthreads();
        for (int k = tid; k < NB; k += 256) {
            const int m0 = p.mlo[k];
            float dp = 0.f;
            for (int m = m0; m < m0 + 3 && m < 64; ++m) dp = fmaf(p.fb[k * 64 + m], dm[m], dp);
            const float2 Bv = GB[k];
            GB[k] = make_float2(2.0f * dp * Bv.x, 2.0f * dp * Bv.y);
        }
    }
    loss = block_sum<4>(loss, scratch);                  // contains the barriers that publish GA / GB
    if (tid == 0) p.partial[blockIdx.x] = loss;
    if (!p.gframes) return;

    // spectral gradient -> time domain:  g_a + i g_b = IFFT(H_a + i H_b) = conj(FFT(conj(W)))
    for (int k = tid; k < N; k += 256) {
        float2 ha, hb;
        if (k == 0 || k == N / 2) {
            ha = make_float2(GA[k].x, 0.f);
            hb = make_float2(GB[k].x, 0.f);
        } else if (k < N / 2) {
            ha = make_float2(0.5f * GA[k].x, 0.5f * GA[k].y);
            hb = make_float2(0.5f * GB[k].x, 0.5f * GB[k].y);
        } else {
            ha = make_float2(0.5f * GA[N - k].x, -0.5f * GA[N - k].y);
            hb = make_float2(0.5f * GB[N - k].x, -0.5f * GB[N - k].y);
        }
        z[k] = make_float2(ha.x - hb.y, -(ha.y + hb.x));
    }
    __syncthreads();
    fft_dif<LOGN>(z, tw, tid);
    float* ga_out = (MODE == MODE_HF) ? p.gframes + ((size_t)bclip * p.F + fa) * N : nullptr;
    float* gb_out = b_valid ? p.gframes + ((size_t)bclip * p.F + fbi) * N : nullptr;
    for (int n = tid; n < N; n += 256) {
        const float w = 0.5f - 0.5f * cospif(2.0f * (float)n / (float)N);
        const float2 y = z[brev<LOGN>(n)];
        if (ga_out) ga_out[n] = y.x * w;
        if (gb_out) gb_out[n] = -y.y * w;
    }
}

// dsig[b,t] = sum over every (frame, position) whose reflect-padded sample is t
__global__ void ola_gather_kernel(const float* __restrict__ gframes, float* __restrict__ dsig, int N, int hop, int F,
                                  int T, int total, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int b = i / T, t = i % T, half = N / 2;
    const float* gf = gframes + (size_t)b * F * N;
    // padded positions that map to t:  direct, left reflection, right reflection
    int pos[3];
    int np = 0;
    pos[np++] = t + half;
    if (t >= 1 && t <= half) pos[np++] = half - t;
    if (t <= T - 2 && 2 * (T - 1) - t + half < T + N) { const int q = 2 * (T - 1) - t + half; if (q >= T + half) pos[np++] = q; }
    float acc = 0.f;
    for (int j = 0; j < np; ++j) {
        const int pp = pos[j];
        int f_hi = pp / hop;
        if (f_hi > F - 1) f_hi = F - 1;
        int f_lo = (pp - N + hop) / hop;               // ceil((pp - N + 1) / hop) for pp-N+1 > 0
        if (pp - N + 1 <= 0) f_lo = 0;
        for (int f = f_lo; f <= f_hi; ++f) acc += gf[(size_t)f * N + (pp - f * hop)];
    }
    dsig[i] = accumulate ? dsig[i] + acc : acc;
}

// out[0] = scale * sum(partial[0..n))   (double accumulation, one block)
__global__ __launch_bounds__(256) void sum_scale_kernel(const float* __restrict__ partial, int n, double scale, float* out) {
    __shared__ double scratch[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)partial[i];
    s = block_sum_d<4>(s, scratch);
    if (threadIdx.x == 0) out[0] = (float)(s * scale);
}

template <int LOGN, int MODE>
int launch_stft(const StftArgs& p, int B, hipStream_t stream) {
    constexpr int N = 1 << LOGN, NB = N / 2 + 1;
    constexpr size_t lds = (size_t)(2 * N + N + 4 * NB + (MODE == MODE_MEL ? 2 * NB + 192 : 0)) * sizeof(float);
    const int units = (MODE == MODE_HF) ? (p.F + 1) / 2 : p.F;
    hipLaunchKernelGGL((stft_loss_kernel<LOGN, MODE>), dim3(B * units), dim3(256), lds, stream, p);
    WM_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" {

// Common contract: signals [B,T] fp32; loss_out: device scalar (mean as the reference defines it);
// dsig [B,T] (may be NULL): d loss / d (watermarked signal | delta); gframes: scratch [B,F,N]; partial: scratch [B*F].
int wm_mel_loss(const float* clean, const float* wm, const float* fb, const int* klo, const int* khi, const int* mlo,
                float* gframes, float* partial, float* loss_out, float* dsig, int B, int T, hipStream_t stream) {
    const int N = 1024, hop = 256, F = 1 + T / hop;
    if (T <= N / 2) return (int)hipErrorInvalidValue;
    const double count = (double)B * 64.0 * F;
    StftArgs p{clean, wm, dsig ? gframes : nullptr, partial, fb, klo, khi, mlo, (float)(1.0 / count), 0.f, 0, T, F, hop};
    int rc = launch_stft<10, MODE_MEL>(p, B, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, stream, (const float*)partial, B * F, 1.0 / count, loss_out);
    WM_CHECK_LAUNCH();
    if (dsig) {
        hipLaunchKernelGGL(ola_gather_kernel, dim3((B * T + 255) / 256), dim3(256), 0, stream, (const float*)gframes, dsig, N, hop, F, T, B * T, 0);
        WM_CHECK_LAUNCH();
    }
    return 0;
}

int wm_loud_loss(const float* clean, const float* wm, float thresh, float* gframes, float* partial, float* loss_out,
                 float* dsig, int B, int T, hipStream_t stream) {
    const int N = 2048, hop = 512, F = 1 + T / hop;
    if (T <= N / 2) return (int)hipErrorInvalidValue;
    const double count = (double)B * (N / 2 + 1) * F;
    StftArgs p{clean, wm, dsig ? gframes : nullptr, partial, nullptr, nullptr, nullptr, nullptr, (float)(1.0 / count), thresh, 0, T, F, hop};
    int rc = launch_stft<11, MODE_LOUD>(p, B, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, stream, (const float*)partial, B * F, 1.0 / count, loss_out);
    WM_CHECK_LAUNCH();
    if (dsig) {
        hipLaunchKernelGGL(ola_gather_kernel, dim3((B * T + 255) / 256), dim3(256), 0, stream, (const float*)gframes, dsig, N, hop, F, T, B * T, 0);
        WM_CHECK_LAUNCH();
    }
    return 0;
}

int wm_hf_penalty(const float* delta, int kcut, float* gframes, float* partial, float* loss_out, float* dsig, int B, int T,
                  hipStream_t stream) {
    const int N = 512, hop = 128, F = 1 + T / hop;
    if (T <= N / 2) return (int)hipErrorInvalidValue;
    const double count = (double)B * (N / 2 + 1) * F;
    const int units = (F + 1) / 2;
    StftArgs p{delta, nullptr, dsig ? gframes : nullptr, partial, nullptr, nullptr, nullptr, nullptr, (float)(1.0 / count), 0.f, kcut, T, F, hop};
    int rc = launch_stft<9, MODE_HF>(p, B, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, stream, (const float*)partial, B * units, 1.0 / count, loss_out);
    WM_CHECK_LAUNCH();
    if (dsig) {
        hipLaunchKernelGGL(ola_gather_kernel, dim3((B * T + 255) / 256), dim3(256), 0, stream, (const float*)gframes, dsig, N, hop, F, T, B * T, 0);
        WM_CHECK_LAUNCH();
    }
    return 0;
}

}  // extern "C"
