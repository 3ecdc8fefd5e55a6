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
// This family is correctness-first (first build of config 5): 64x64 tiles, synchronous LDS staging, scalar loads.
#include "wm_common.hpp"
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
    int act;             // 0 none, 1 ELU
};

constexpr int GC = 8;    // input channels per LDS chunk

__global__ __launch_bounds__(256) void gconv_kernel(GConvArgs a) {
    extern __shared__ __align__(16) float smem[];
    const int XW = 63 * a.S + a.K;          // input span of a 64-position tile
    const int KK = GC * a.K;                // (channel, tap) rows per chunk -- always even
    float* Xs = smem;                       // [GC][XW]
    float* Ws = Xs + GC * XW;               // [KK][64]
    int* xoff = reinterpret_cast<int*>(Ws + KK * 64);   // [KK]: ci_local*XW + tap
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int wm_ = wave & 1, wn = wave >> 1;
    const int n0 = blockIdx.x * 64, m0 = blockIdx.y * 64, nb = blockIdx.z;
    const float* xb = a.x + (size_t)nb * a.Cin * a.Lin;
    for (int i = tid; i < KK; i += 256) xoff[i] = (i / a.K) * XW + (i % a.K);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int u0 = n0 * a.S - a.P;
    for (int c0 = 0; c0 < a.Cin; c0 += GC) {
        __syncthreads();
        for (int i = tid; i < GC * XW; i += 256) {
            const int ci = i / XW, j = i - ci * XW, u = u0 + j, c = c0 + ci;
            const bool ok = (c < a.Cin) && (u >= 0) && (u < a.Lin);
            const float v = xb[(size_t)min(c, a.Cin - 1) * a.Lin + min(max(u, 0), a.Lin - 1)];   // branch-free load
            Xs[i] = ok ? v : 0.f;
        }
        for (int i = tid; i < KK * 64; i += 256) {
            const int kk = i >> 6, m = i & 63, gk = c0 * a.K + kk, gm = m0 + m;
            const bool ok = (gk < a.Cin * a.K) && (gm < a.Mtot);
            const float v = a.wp[(size_t)min(gk, a.Cin * a.K - 1) * a.Mtot + min(gm, a.Mtot - 1)];
            Ws[i] = ok ? v : 0.f;
        }
        __syncthreads();
        const float* ap = Ws + wm_ * 32 + l31;
        const float* bp = Xs + (wn * 32 + l31) * a.S;
        for (int ks = 0; ks < KK / 2; ++ks) {
            const int kk = 2 * ks + half;
            acc = mfma32(ap[kk * 64], bp[xoff[kk]], acc);
        }
    }
    const int n = n0 + wn * 32 + l31;
    if (n >= a.Nout) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm_ * 32 + mfma_row(r, half);
        if (m >= a.Mtot) continue;
        int co = m, t = n;
        if (a.st > 1) { co = m / a.st; t = n * a.st + (m - co * a.st) - a.shp; }
        if (t < 0 || t >= a.Lout) continue;
        float v = acc[r];
        if (a.bias) v += a.bias[co];
        if (a.vec) v += a.vec[(size_t)nb * a.Cout + co];
        const size_t o = ((size_t)nb * a.Cout + co) * a.Lout + t;
        if (a.res) v += a.res[o];
        if (a.act == 1) v = v > 0.f ? v : expm1f(v);
        a.y[o] = v;
    }
}

// [A][C][L] -> [L][C][A]   (batch-major <-> time-major sequence layouts around the LSTM)
__global__ void permute_acl_kernel(const float* __restrict__ x, float* __restrict__ y, int A, int C, int L) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)A * C * L) return;
    const int l = (int)(i % L), c = (int)((i / L) % C), aa = (int)(i / ((size_t)L * C));
    y[((size_t)l * C + c) * A + aa] = x[i];
}

// One LSTM time step for hidden size H (multiple of 32) and batch Bn, state kept time-major / unit-major:
//   gates[q*H + u][b] = xp[q*H + u][b] + sum_k whhT[k][q*H + u] * hprev[k][b]          (MFMA, K = H)
//   c = sig(f) c + sig(i) tanh(g);  h = sig(o) tanh(c)
// wave = 32 units x 32 batch columns x 4 gates.  hprev == nullptr means zero initial state.
__global__ __launch_bounds__(64) void lstm_h_step_fwd_kernel(const float* xp, const float* __restrict__ whhT,
                                                            const float* __restrict__ hprev, const float* __restrict__ cprev,
                                                            float* __restrict__ hout, float* __restrict__ cout,
                                                            float* gates_out /* may alias xp */, int H, int Bn) {
    const int lane = threadIdx.x, half = lane >> 5, l31 = lane & 31;
    const int u0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
    const int bcol = min(b0 + l31, Bn - 1);
    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    if (hprev) {
        for (int ks = 0; ks < H / 2; ++ks) {
            const int k = 2 * ks + half;
            const float bv = hprev[(size_t)k * Bn + bcol];
            const float* wr = whhT + (size_t)k * 4 * H + u0 + l31;
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = mfma32(wr[q * H], bv, acc[q]);
        }
    }
    if (b0 + l31 >= Bn) return;
    const int b = b0 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int u = u0 + mfma_row(r, half);
        const float ai = acc[0][r] + xp[(size_t)(0 * H + u) * Bn + b];
        const float af = acc[1][r] + xp[(size_t)(1 * H + u) * Bn + b];
        const float ag = acc[2][r] + xp[(size_t)(2 * H + u) * Bn + b];
        const float ao = acc[3][r] + xp[(size_t)(3 * H + u) * Bn + b];
        const float gi = 1.f / (1.f + expf(-ai)), gf = 1.f / (1.f + expf(-af)), gg = tanhf(ag), go = 1.f / (1.f + expf(-ao));
        const float cp = cprev ? cprev[(size_t)u * Bn + b] : 0.f;
        const float c = gf * cp + gi * gg;
        const float h = go * tanhf(c);
        hout[(size_t)u * Bn + b] = h;
        cout[(size_t)u * Bn + b] = c;
        if (gates_out) {
            gates_out[(size_t)(0 * H + u) * Bn + b] = gi; gates_out[(size_t)(1 * H + u) * Bn + b] = gf;
            gates_out[(size_t)(2 * H + u) * Bn + b] = gg; gates_out[(size_t)(3 * H + u) * Bn + b] = go;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Generic weight gradient:  G[a][b][k] += sum_{nb,t} A[nb][a][t] * Bx[nb][b][t*S + k - P]      (k < K <= 16)
//   Conv1d:           A = dL/dy (rows = out channels),  Bx = layer input  -> dW[out][in][k]
//   ConvTranspose1d:  A = layer input (rows = in channels), Bx = dL/dy    -> dW[in][out][k]   (S = stride, P = padding)
//   Linear / LSTM:    K = 1.
// One wave owns a 32x32 (a,b) tile for every tap (K accumulators); the workgroup's 4 waves split each 64-position
// chunk; a workgroup walks a strided list of (nb, chunk) pairs and flushes once with float atomics (the slabs of the
// largest layer would be 8 MB per workgroup, so the fixed-order slab reduce of the main16 kernels does not scale here;
// consequence: the summation order over workgroups is not fixed, results are reproducible to fp32 round-off only).
// dbias[a] += sum A[a][t] is produced by the b-tile-0 workgroups.
template <int KMAX>
__global__ __launch_bounds__(256) void gwgrad_kernel(const float* __restrict__ A, const float* __restrict__ Bx,
                                                     float* __restrict__ G, float* __restrict__ dbias, int NB, int Ca, int Cb,
                                                     int La, int Lb, int K, int S, int P, int TC) {
    // Workgroup tile = 64 (a) x 64 (b); wave (ma, mb) owns one 32 x 32 block of it for every tap and walks the WHOLE
    // chunk, so a staged element feeds two waves (the first version staged a 32 x 32 tile and split the chunk over the
    // waves: twice the LDS traffic and four times the global traffic per MFMA).  TC = positions per LDS chunk (multiple of
    // 64, chosen by the launcher so that two workgroups fit a CU: one stages while the other multiplies).
    extern __shared__ __align__(16) float smem[];
    const int BW = (TC - 1) * S + K;            // Bx span of a chunk
    const int AS = TC + 1, BS = BW | 1;         // odd strides: operands are read down a column
    float* As = smem;                           // [64][AS]
    float* Bs = smem + 64 * AS;                 // [64][BS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int ma = wave & 1, mb = wave >> 1;
    const int a0 = blockIdx.x * 64, b0 = blockIdx.y * 64;
    const int nchunks = (La + TC - 1) / TC, nwork = NB * nchunks;
    f32x16 acc[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    float bsum = 0.f;
    for (int w = blockIdx.z; w < nwork; w += gridDim.z) {
        const int nb = w / nchunks, t0 = (w - nb * nchunks) * TC;
        const float* Ab = A + (size_t)nb * Ca * La;
        const float* Bb = Bx + (size_t)nb * Cb * Lb;
        __syncthreads();
        // staging: wave w copies rows w, w + 4, ...; lanes run along the position axis (coalesced, branch-free)
        for (int r = wave; r < 64; r += 4) {
            const int a = a0 + r;
            const float* src = Ab + (size_t)min(a, Ca - 1) * La;
            for (int j = lane; j < TC; j += 64) {
                const int t = t0 + j;
                const float v = src[min(t, La - 1)];
                As[r * AS + j] = (a < Ca && t < La) ? v : 0.f;
            }
        }
        const int u0 = t0 * S - P;
        for (int r = wave; r < 64; r += 4) {
            const int b = b0 + r;
            const float* src = Bb + (size_t)min(b, Cb - 1) * Lb;
            for (int j = lane; j < BW; j += 64) {
                const int u = u0 + j;
                const float v = src[min(max(u, 0), Lb - 1)];
                Bs[r * BS + j] = (b < Cb && u >= 0 && u < Lb) ? v : 0.f;
            }
        }
        __syncthreads();
        const float* ap = As + (ma * 32 + l31) * AS + half;
        const float* bp = Bs + (mb * 32 + l31) * BS + half * S;
#pragma unroll 4
        for (int s = 0; s < TC / 2; ++s) {
            const float av = ap[2 * s];
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < K) acc[k] = mfma32(av, bp[2 * s * S + k], acc[k]);
        }
        if (dbias && blockIdx.y == 0) {          // row tid & 63, quarter tid >> 6 of the chunk
            const float* rp = As + (tid & 63) * AS + (tid >> 6) * (TC / 4);
            for (int j = 0; j < TC / 4; ++j) bsum += rp[j];
        }
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int a = a0 + ma * 32 + mfma_row(r, half), b = b0 + mb * 32 + l31;
                if (a < Ca && b < Cb) atomicAdd(G + ((size_t)a * Cb + b) * K + k, acc[k][r]);
            }
        }
    }
    if (dbias && blockIdx.y == 0 && a0 + (tid & 63) < Ca) atomicAdd(dbias + a0 + (tid & 63), bsum);
}

// dz = g * elu'(y) with y = ELU(z):  elu'(z) = 1 for y > 0 else y + 1   (alpha = 1)
__global__ void elu_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y, float* __restrict__ dz, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float yy = y[i];
    dz[i] = g[i] * (yy > 0.f ? 1.f : yy + 1.f);
}

// backward of one LSTM time step (pointwise part): gates (activated i,f,g,o) -> pre-activation gradients, in place
//   dh = total gradient w.r.t. h_t, dc (in/out) = gradient w.r.t. c_t coming from step t+1 -> w.r.t. c_{t-1}
__global__ void lstm_h_step_bwd_kernel(float* __restrict__ gates, const float* __restrict__ c, const float* __restrict__ cprev,
                                       const float* __restrict__ dh, float* __restrict__ dc, int H, int Bn) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * Bn) return;
    const size_t hb = (size_t)H * Bn;
    const float gi = gates[i], gf = gates[hb + i], gg = gates[2 * hb + i], go = gates[3 * hb + i];
    const float tc = tanhf(c[i]);
    const float dht = dh[i];
    const float dct = dc[i] + dht * go * (1.f - tc * tc);
    const float cp = cprev ? cprev[i] : 0.f;
    gates[i] = dct * gg * gi * (1.f - gi);
    gates[hb + i] = dct * cp * gf * (1.f - gf);
    gates[2 * hb + i] = dct * gi * (1.f - gg * gg);
    gates[3 * hb + i] = dht * tc * go * (1.f - go);
    dc[i] = dct * gf;
}

// out[c] += sum_{nb,t} x[nb][c][t]      (bias gradient of the transposed convolutions); one block per channel
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int NB, int C, int L) {
    __shared__ float scratch[4];
    const int c = blockIdx.x;
    float s = 0.f;
    for (int nb = blockIdx.y; nb < NB; nb += gridDim.y) {
        const float* r = x + ((size_t)nb * C + c) * L;
        for (int t = threadIdx.x; t < L; t += 256) s += r[t];
    }
    s = block_sum<4>(s, scratch);
    if (threadIdx.x == 0) atomicAdd(out + c, s);
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

// dtable[idx[b]][:] += dvec[b][:]   (dense embedding gradient for any embedding width)
__global__ void rows_scatter_add_kernel(float* __restrict__ dtable, const long long* __restrict__ idx,
                                        const float* __restrict__ dvec, int Bn, int dim, int nrows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Bn * dim) return;
    const long long m = idx[i / dim];
    if (m < 0 || m >= nrows) return;
    atomicAdd(dtable + (size_t)m * dim + (i % dim), dvec[i]);
}

}  // namespace

extern "C" {

// y[nb][co][t'] = act( bias[co] + vec[nb][co] + res + sum_{ci,k} wp[ci*K+k][m] * x[nb][ci][n*S + k - P] )
//   st == 1: m = co, t' = n (Cout == Mtot, Lout == Nout);  st > 1: m = co*st + phase, t' = n*st + phase - shp.
int wm_gconv(const float* x, const float* wp, const float* bias, const float* vec, const float* res, float* y, int NB,
             int Cin, int Lin, int K, int S, int P, int Mtot, int Nout, int st, int shp, int Cout, int Lout, int act,
             hipStream_t stream) {
    if (NB <= 0 || Cin <= 0 || K <= 0 || K > 16 || S <= 0 || S > 8 || Mtot <= 0 || Nout <= 0 || st < 1) return (int)hipErrorInvalidValue;
    GConvArgs a{x, wp, bias, vec, res, y, NB, Cin, Lin, K, S, P, Mtot, Nout, st, shp, Cout, Lout, act};
    const size_t lds = (size_t)(GC * (63 * S + K) + GC * K * 64 + GC * K) * sizeof(float);
    dim3 grid((Nout + 63) / 64, (Mtot + 63) / 64, NB);
    hipLaunchKernelGGL(gconv_kernel, grid, dim3(256), lds, stream, a);
    WM_CHECK_LAUNCH();
    return 0;
}

int wm_permute_acl(const float* x, float* y, int A, int C, int L, hipStream_t stream) {
    const size_t n = (size_t)A * C * L;
    hipLaunchKernelGGL(permute_acl_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, y, A, C, L);
    WM_CHECK_LAUNCH();
    return 0;
}

// G[a][b][k] += sum_{nb,t} A[nb][a][t] * Bx[nb][b][t*S + k - P]; dbias[a] += sum A (dbias may be NULL).  ACCUMULATES
// (float atomics): the caller zeroes G / dbias.  K <= 16.
int wm_gwgrad(const float* A, const float* Bx, float* G, float* dbias, int NB, int Ca, int Cb, int La, int Lb, int K, int S,
              int P, hipStream_t stream) {
    if (NB <= 0 || Ca <= 0 || Cb <= 0 || La <= 0 || K <= 0 || K > 16 || S <= 0 || S > 8) return (int)hipErrorInvalidValue;
    int TC = 256;                                  // positions per LDS chunk: two workgroups of <= 78 KB per CU
    auto lds_of = [&](int tc) { return (size_t)(64 * (tc + 1) + 64 * (((tc - 1) * S + K) | 1)) * sizeof(float); };
    while (TC > 64 && lds_of(TC) > 78 * 1024) TC >>= 1;
    if (La <= 64) TC = 64;
    const int nchunks = (La + TC - 1) / TC, nwork = NB * nchunks;
    const int tiles = ((Ca + 63) / 64) * ((Cb + 63) / 64);
    int gz = (1024 + tiles - 1) / tiles;          // ~1024 workgroups in flight overall (2 per CU, twice over)
    if (gz > nwork) gz = nwork;
    if (gz < 1) gz = 1;
    const size_t lds = lds_of(TC);
    if (lds > 150 * 1024) return (int)hipErrorInvalidValue;
    dim3 grid((Ca + 63) / 64, (Cb + 63) / 64, gz);
    static wm::DevOnce done;
    if (!wm::dev_done(done)) {
        WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gwgrad_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gwgrad_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gwgrad_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        wm::dev_mark(done);
    }
    if (K <= 4) hipLaunchKernelGGL(gwgrad_kernel<4>, grid, dim3(256), lds, stream, A, Bx, G, dbias, NB, Ca, Cb, La, Lb, K, S, P, TC);
    else if (K <= 8) hipLaunchKernelGGL(gwgrad_kernel<8>, grid, dim3(256), lds, stream, A, Bx, G, dbias, NB, Ca, Cb, La, Lb, K, S, P, TC);
    else hipLaunchKernelGGL(gwgrad_kernel<16>, grid, dim3(256), lds, stream, A, Bx, G, dbias, NB, Ca, Cb, La, Lb, K, S, P, TC);
    WM_CHECK_LAUNCH();
    return 0;
}

int wm_elu_bwd(const float* g, const float* y, float* dz, long long n, hipStream_t stream) {
    hipLaunchKernelGGL(elu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, g, y, dz, (size_t)n);
    WM_CHECK_LAUNCH();
    return 0;
}

int wm_lstm_h_step_bwd(float* gates, const float* c, const float* cprev, const float* dh, float* dc, int H, int Bn,
                       hipStream_t stream) {
    hipLaunchKernelGGL(lstm_h_step_bwd_kernel, dim3((H * Bn + 255) / 256), dim3(256), 0, stream, gates, c, cprev, dh, dc, H, Bn);
    WM_CHECK_LAUNCH();
    return 0;
}

int wm_channel_sum(const float* x, float* out, int NB, int C, int L, hipStream_t stream) {
    hipLaunchKernelGGL(channel_sum_kernel, dim3(C, NB < 64 ? NB : 64), dim3(256), 0, stream, x, out, NB, C, L);
    WM_CHECK_LAUNCH();
    return 0;
}

int wm_rowsum_any(const float* x, float* out, int rows, int L, hipStream_t stream) {
    hipLaunchKernelGGL(rowsum_any_kernel, dim3(rows), dim3(256), 0, stream, x, out, L);
    WM_CHECK_LAUNCH();
    return 0;
}

int wm_rows_scatter_add(float* dtable, const long long* idx, const float* dvec, int Bn, int dim, int nrows, hipStream_t stream) {
    hipLaunchKernelGGL(rows_scatter_add_kernel, dim3((Bn * dim + 255) / 256), dim3(256), 0, stream, dtable, idx, dvec, Bn, dim, nrows);
    WM_CHECK_LAUNCH();
    return 0;
}

// one time step of an LSTM layer with hidden size H (H % 32 == 0); all tensors [rows][Bn] with Bn contiguous
int wm_lstm_h_step_fwd(const float* xp, const float* whhT, const float* hprev, const float* cprev, float* hout, float* cout,
                       float* gates_out, int H, int Bn, hipStream_t stream) {
    if (H <= 0 || (H & 31) || Bn <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(lstm_h_step_fwd_kernel, dim3(H / 32, (Bn + 31) / 32), dim3(64), 0, stream, xp, whhT, hprev, cprev, hout,
                       cout, gates_out, H, Bn);
    WM_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
