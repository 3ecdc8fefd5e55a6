// The thin ends of the two networks (py/main16.py):
//   stem   Conv1d(1,64,7,padding=3)   Generator.encoder[0] :134 / Detector.model[0] :177
//   head1  Conv1d(64,1,1)             Generator.decoder[2] :146
//   head17 Conv1d(64,1+bits,1)        Detector.model[3]    :180, written directly in the
//                                     (B,T,1+bits) layout Detector.forward returns (:186)
// These are HBM-bound (one 64-channel frame in or out per sample, a handful of FLOPs per
// byte) so they are plain VALU kernels with 16-B coalesced frame accesses; only the tiny
// one-channel side goes through LDS.
#include "wm_common.hpp"
using namespace wm;

namespace {

// ------------------------------------------------------------------------------- stem forward
// block = (clip, 1024-sample tile); thread = 4 consecutive samples; loop over the 64 filters.
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ s, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ y, int T) {
    __shared__ float ss[1024 + 8];
    __shared__ float ws[64 * 8];
    const int tilesPerClip = (T + 1023) / 1024;
    const int b = blockIdx.x / tilesPerClip, t0 = (blockIdx.x % tilesPerClip) * 1024;
    const float* sb = s + (size_t)b * T;
    for (int i = threadIdx.x; i < 1024 + 6; i += 256) {
        const int t = t0 - 3 + i;
        ss[i] = (t >= 0 && t < T) ? sb[t] : 0.f;
    }
    for (int i = threadIdx.x; i < 512; i += 256) {
        const int co = i >> 3, j = i & 7;
        ws[i] = (j < 7) ? w[co * 7 + j] : bias[co];
    }
    __syncthreads();
    const int t = t0 + 4 * threadIdx.x;
    if (t >= T) return;
    float v[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) v[i] = ss[4 * threadIdx.x + i];
    float* yb = y + (size_t)b * 64 * T + t;
#pragma unroll 4
    for (int co = 0; co < 64; ++co) {
        const float* wc = ws + co * 8;
        float4 o = make_float4(wc[7], wc[7], wc[7], wc[7]);
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const float wj = wc[j];
            o.x = fmaf(wj, v[j], o.x);
            o.y = fmaf(wj, v[j + 1], o.y);
            o.z = fmaf(wj, v[j + 2], o.z);
            o.w = fmaf(wj, v[j + 3], o.w);
        }
        *reinterpret_cast<float4*>(yb + (size_t)co * T) = o;
    }
}

// ------------------------------------------------------------------------------ stem backward
// g [B,64,T]; persistent blocks over (clip, 256-sample) tiles, next tile prefetched into registers.
//   dw[co][j] = sum_{b,t} g[co,t] s[t+j-3] ; db[co] = sum g[co,t]  -> on the fp32 matrix cores: M = co, K = time,
//               N = 8 columns (7 taps + a column of ones for the bias) padded to 32     -> partial[block][64*8]
//   ds[b,t]   = sum_{co,j} g[co,t-j+3] w[co][j]   (only if ds != nullptr; clips >= nds are skipped): VALU, register
//               blocked 4 samples x 16 channels per thread, the four channel groups are summed through LDS.
__global__ __launch_bounds__(256) void stem_bwd_kernel(const float* __restrict__ g, const float* __restrict__ s,
                                                       const float* __restrict__ w, float* __restrict__ ds,
                                                       float* __restrict__ partial, int B, int T, int nds) {
    constexpr int NT = 256, GS = NT + 8;             // g tile row: 3 halo | 256 | 3 halo starting at column 1; main part at 4
    extern __shared__ __align__(16) float smem[];
    float* gs = smem;                                // [64][GS]
    float* ss = gs + 64 * GS;                        // [NT + 40]: s[t0 - 3 + i]; zero padded so that B-operand reads stay in range
    float* ws = ss + NT + 40;                        // [64][8]
    float* dsp = ws + 512;                           // [4][NT] partial ds of the four channel groups
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    for (int i = tid; i < 512; i += 256) ws[i] = ((i & 7) < 7) ? w[(i >> 3) * 7 + (i & 7)] : 0.f;
    const int tilesPerClip = (T + NT - 1) / NT, ntiles = B * tilesPerClip;
    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    float4 sg[16];
    float hg[2], sv[2];
    // piece p < 16: one row group of the g tile; piece 16: halo columns + the clip samples.  Branch-free (clamped addresses).
    // The main loop issues one piece per two k-steps of the dw product instead of the whole 64-KB tile at once.
    auto load_piece = [&](int tile, int p) {
        const int b = tile / tilesPerClip, t0 = (tile % tilesPerClip) * NT;
        const float* gb = g + (size_t)b * 64 * T;
        if (p < 16) {
            const int i = tid + p * 256, c = i >> 6, q = i & 63;
            sg[p] = *reinterpret_cast<const float4*>(gb + (size_t)c * T + min(t0 + 4 * q, T - 4));
            return;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = min(tid + k * 256, 64 * 6 - 1), c = i / 6, h = i % 6;
            const int t = (h < 3) ? t0 - 3 + h : t0 + NT + (h - 3);
            hg[k] = gb[(size_t)c * T + min(max(t, 0), T - 1)];
            const int ts_ = t0 - 3 + tid + k * 256;
            sv[k] = s[(size_t)b * T + min(max(ts_, 0), T - 1)];
        }
    };
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int p = 0; p <= 16; ++p) load_piece(tile, p);
    };
    auto write_tile = [&](int tile) {
        const int t0 = (tile % tilesPerClip) * NT;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = tid + k * 256, c = i >> 6, q = i & 63;
            float4 v = sg[k];
            if (t0 + 4 * q >= T) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(gs + c * GS + 4 + 4 * q) = v;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = tid + k * 256;
            if (i < 64 * 6) {
                const int c = i / 6, h = i % 6;
                const int t = (h < 3) ? t0 - 3 + h : t0 + NT + (h - 3);
                gs[c * GS + ((h < 3) ? 1 + h : 4 + NT + (h - 3))] = (t >= 0 && t < T) ? hg[k] : 0.f;
            }
            if (i < NT + 40) {
                const int ts_ = t0 - 3 + i;
                ss[i] = (i < NT + 6 && ts_ >= 0 && ts_ < T) ? sv[k] : 0.f;
            }
        }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) load_tile(tile);
    __syncthreads();
    if (tile < ntiles) write_tile(tile);
    __syncthreads();
    while (tile < ntiles) {
        const int next = tile + gridDim.x, nextc = min(next, ntiles - 1);     // clamped: loaded (valid memory), never written
        const int b = tile / tilesPerClip, t0 = (tile % tilesPerClip) * NT;
        // ---- dw / db: this wave's 64 time steps.  A[i = co][k = t] = gs[co][t];  B[k = t][j] = s[t + j - 3] (j < 7), 1 (j == 7), 0
        {
            const float* ap = gs + l31 * GS + 4 + wave * 64 + half;
            const float* bp = ss + wave * 64 + half + l31;        // ss[t - t0 + j] = s[t + j - 3]
            const bool tapcol = l31 < 7, onecol = l31 == 7;
#pragma unroll
            for (int k2 = 0; k2 < 32; ++k2) {
                if ((k2 & 1) == 0 || k2 == 31) load_piece(nextc, k2 == 31 ? 16 : k2 >> 1);
                const float sval = bp[2 * k2];
                const float bv = tapcol ? sval : (onecol ? 1.f : 0.f);
                acc[0] = mfma32(ap[2 * k2], bv, acc[0]);
                acc[1] = mfma32(ap[32 * GS + 2 * k2], bv, acc[1]);
            }
        }
        // ---- ds (Detector stem: gradient w.r.t. the watermarked half of the batch)
        if (ds && b < nds) {
            const int cg = wave, tq = lane;                       // 16 channels x 4 samples per thread
            float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll 4
            for (int cc = 0; cc < 16; ++cc) {
                const int co = cg * 16 + cc;
                const float4 ga = *reinterpret_cast<const float4*>(gs + co * GS + 4 * tq);         // t0+4tq-4 .. -1
                const float4 gb4 = *reinterpret_cast<const float4*>(gs + co * GS + 4 * tq + 4);    // t0+4tq   .. +3
                const float4 gc = *reinterpret_cast<const float4*>(gs + co * GS + 4 * tq + 8);     // t0+4tq+4 .. +7
                const float4 w0 = *reinterpret_cast<const float4*>(ws + co * 8), w1 = *reinterpret_cast<const float4*>(ws + co * 8 + 4);
                // gv[i] = g[co][t0 + 4tq - 3 + i], i = 0..9 ;  ds[t] = sum_j g[t + 3 - j] w[j]
                const float gv[10] = {ga.y, ga.z, ga.w, gb4.x, gb4.y, gb4.z, gb4.w, gc.x, gc.y, gc.z};
                const float wj[7] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z};
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    o0 = fmaf(gv[6 - j], wj[j], o0); o1 = fmaf(gv[7 - j], wj[j], o1);
                    o2 = fmaf(gv[8 - j], wj[j], o2); o3 = fmaf(gv[9 - j], wj[j], o3);
                }
            }
            *reinterpret_cast<float4*>(dsp + cg * NT + 4 * tq) = make_float4(o0, o1, o2, o3);
        }
        __syncthreads();
        if (ds && b < nds) {
            const int t = t0 + tid;
            if (t < T) ds[(size_t)b * T + t] = (dsp[tid] + dsp[NT + tid]) + (dsp[2 * NT + tid] + dsp[3 * NT + tid]);
        }
        if (next < ntiles) write_tile(next);
        __syncthreads();
        tile = next;
    }
    // fixed-order reduction of the four waves' [64 x 8] tiles
    float* red = gs;
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv && l31 < 8) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = (mt * 32 + mfma_row(r, half)) * 8 + l31;
                    red[o] = (wv == 0) ? acc[mt][r] : red[o] + acc[mt][r];
                }
        }
        __syncthreads();
    }
    float* out = partial + (size_t)blockIdx.x * 512;
    for (int i = tid; i < 512; i += 256) out[i] = red[i];
}

// out[i] (+)= sum_p partial[p*stride + i],  i < count.  Block = 64 outputs x 4 quarters of the slab list (fp64, fixed
// order: bit-reproducible); launch with 256 threads and ceil(count / 64) blocks.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, int nparts, int stride, int count,
                                                              float* __restrict__ out, int accumulate) {
    __shared__ double sq[4][64];
    const int il = threadIdx.x & 63, grp = threadIdx.x >> 6, i = blockIdx.x * 64 + il;
    const int per = (nparts + 3) / 4, p0 = grp * per, p1 = min(p0 + per, nparts);
    double s0 = 0.0, s1 = 0.0;
    if (i < count) {
        int p = p0;
        for (; p + 1 < p1; p += 2) {
            s0 += (double)partial[(size_t)p * stride + i];
            s1 += (double)partial[(size_t)(p + 1) * stride + i];
        }
        if (p < p1) s0 += (double)partial[(size_t)p * stride + i];
    }
    sq[grp][il] = s0 + s1;
    __syncthreads();
    if (grp == 0 && i < count) {
        const float s = (float)((sq[0][il] + sq[1][il]) + (sq[2][il] + sq[3][il]));
        out[i] = accumulate ? out[i] + s : s;
    }
}
// stem partial [nparts][64][8] -> dw[64][7], db[64]
__global__ void stem_reduce_kernel(const float* __restrict__ partial, int nparts, float* dw, float* db, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 512) return;
    double sd = 0.0;
    for (int p = 0; p < nparts; ++p) sd += (double)partial[(size_t)p * 512 + i];
    const float s = (float)sd;
    const int co = i >> 3, j = i & 7;
    float* dst = (j < 7) ? dw + co * 7 + j : db + co;
    *dst = accumulate ? *dst + s : s;
}

// ------------------------------------------------------------------------------------ head1
__global__ __launch_bounds__(256) void head1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y, int T4,
                                                        int total4) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int b = i / T4, q = i % T4;
    const float4* xb = reinterpret_cast<const float4*>(x) + (size_t)b * 64 * T4 + q;
    const float b0 = bias[0];
    float4 o = make_float4(b0, b0, b0, b0);
#pragma unroll 8
    for (int c = 0; c < 64; ++c) {
        const float4 v = xb[(size_t)c * T4];
        const float wc = w[c];
        o.x = fmaf(wc, v.x, o.x); o.y = fmaf(wc, v.y, o.y); o.z = fmaf(wc, v.z, o.z); o.w = fmaf(wc, v.w, o.w);
    }
    reinterpret_cast<float4*>(y)[i] = o;
}

// dx[c,t] = w[c] g[t];  dw[c] = sum g[t] x[c,t];  db = sum g   -> partial[block][65]
__global__ __launch_bounds__(256) void head1_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                        const float* __restrict__ w, float* __restrict__ dx,
                                                        float* __restrict__ partial, int T4, int total4) {
    __shared__ float red[4][65];
    float acc[65];
#pragma unroll
    for (int c = 0; c < 65; ++c) acc[c] = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total4; i += gridDim.x * 256) {
        const int b = i / T4, q = i % T4;
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        const size_t base = (size_t)b * 64 * T4 + q;
        acc[64] += (gv.x + gv.y) + (gv.z + gv.w);
#pragma unroll
        for (int c = 0; c < 64; ++c) {
            const float4 v = reinterpret_cast<const float4*>(x)[base + (size_t)c * T4];
            acc[c] += fmaf(gv.x, v.x, gv.y * v.y) + fmaf(gv.z, v.z, gv.w * v.w);
            const float wc = w[c];
            reinterpret_cast<float4*>(dx)[base + (size_t)c * T4] = make_float4(wc * gv.x, wc * gv.y, wc * gv.z, wc * gv.w);
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 65; ++c) {
        const float v = wave_sum(acc[c]);
        if (lane == 0) red[wv][c] = v;
    }
    __syncthreads();
    if (threadIdx.x < 65)
        partial[(size_t)blockIdx.x * 65 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// ------------------------------------------------------------------------------------ head17
// logits[b,t,o] = bias[o] + sum_c w[o][c] x[b,c,t]     NO = 1 + message_bits (<= 17)
template <int NO>
__global__ __launch_bounds__(256) void headN_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y, int T) {
    // One thread = one time step: its 64 channel values sit in registers and every weight is a wave-uniform scalar
    // operand (s_load through the scalar cache) -- the first version broadcast the weights through LDS reads, whose
    // return path (64 lanes x 16 B per read) bounded the kernel at half the HBM rate.
    __shared__ float os[256 * NO];
    const int tilesPerClip = (T + 255) / 256;
    const int b = blockIdx.x / tilesPerClip, t0 = (blockIdx.x % tilesPerClip) * 256;
    const int t = min(t0 + (int)threadIdx.x, T - 1);           // clamped: lanes past T compute a copy that is never stored
    const float* xb = x + (size_t)b * 64 * T + t;
    float v[64];
#pragma unroll
    for (int c = 0; c < 64; ++c) v[c] = xb[(size_t)c * T];
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        const float* wo = w + o * 64;                            // uniform address: scalar loads
        float a0 = bias[o], a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int c = 0; c < 64; c += 4) {
            a0 = fmaf(wo[c], v[c], a0); a1 = fmaf(wo[c + 1], v[c + 1], a1);
            a2 = fmaf(wo[c + 2], v[c + 2], a2); a3 = fmaf(wo[c + 3], v[c + 3], a3);
        }
        os[threadIdx.x * NO + o] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    const int nvalid = min(256, T - t0) * NO;
    float* yb = y + ((size_t)b * T + t0) * NO;
    for (int i = threadIdx.x; i < nvalid; i += 256) yb[i] = os[i];
}

// g [B,T,NO] -> dx[b,c,t] = sum_o w[o][c] g[b,t,o];  dw[o][c] = sum g[b,t,o] x[b,c,t];  db[o] = sum g
// Both products run on the fp32 matrix cores (the VALU/LDS version spent 11.5 ms per step at B=256):
//   dx: M = channel (2 tiles), N = time (the wave's 64 steps), K = output index padded to 18
//   dw: M = output index padded to 32, N = channel (2 tiles), K = time; accumulators persist over tiles
// partial[block][NO*64 + NO]
template <int NO>
__global__ __launch_bounds__(256) void headN_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                        const float* __restrict__ w, float* __restrict__ dx,
                                                        float* __restrict__ partial, int B, int T) {
    constexpr int XS = 257, GS = 33, KS = (NO + 1) / 2;
    extern __shared__ __align__(16) float smem[];
    float* xs = smem;                  // [64][XS]
    float* gsm = xs + 64 * XS;         // [256][GS], columns >= NO are zero
    float* ws = gsm + 256 * GS;        // [32][64],  rows >= NO are zero
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    for (int i = tid; i < 32 * 64; i += 256) ws[i] = (i < NO * 64) ? w[i] : 0.f;
    for (int i = tid; i < 256 * GS; i += 256) gsm[i] = 0.f;
    const int tilesPerClip = (T + 255) / 256, ntiles = B * tilesPerClip;
    f32x16 accw[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { accw[0][r] = 0.f; accw[1][r] = 0.f; }
    float accb = 0.f;
    // next tile is fetched into registers (16-B loads, clamped addresses, no branches) while this one is processed
    constexpr int NG4 = (256 * NO + 3) / 4, NGV = (NG4 + 255) / 256;
    float4 sx[16], sgv[NGV];
    // piece p of the staging: 0..15 x rows, 16.. the g tile.  The main loop issues one piece per k-step of the dw product
    // (an 81-KB burst would block the wave at issue, see conv64bf3_kernel)
    auto load_piece = [&](int tile, int p) {
        const int b = tile / tilesPerClip, t0 = (tile % tilesPerClip) * 256;
        if (p < 16) {
            const int i = tid + p * 256, c = i >> 6, q = i & 63;
            sx[p] = *reinterpret_cast<const float4*>(x + ((size_t)b * 64 + c) * T + min(t0 + 4 * q, T - 4));
        } else {
            // the g tile is 256*NO contiguous floats starting at a 16-B aligned address (t0 % 256 == 0, T % 4 == 0)
            const float4* gb4 = reinterpret_cast<const float4*>(g + ((size_t)b * T + t0) * NO);
            const int lim4 = ((T - t0 < 256 ? T - t0 : 256) * NO) / 4;        // whole float4s available in this clip
            sgv[p - 16] = gb4[min(tid + (p - 16) * 256, lim4 - 1)];
        }
    };
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int p = 0; p < 16 + NGV; ++p) load_piece(tile, p);
    };
    auto write_tile = [&](int tile) {
        const int t0 = (tile % tilesPerClip) * 256;
        const int nt = min(256, T - t0);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = tid + k * 256, c = i >> 6, q = i & 63;
            float4 v = sx[k];
            if (4 * q >= nt) v = make_float4(0.f, 0.f, 0.f, 0.f);
            float* d = xs + c * XS + 4 * q;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
#pragma unroll
        for (int k = 0; k < NGV; ++k) {
            const int f = tid + k * 256;
            if (f < NG4) {
                const float e[4] = {sgv[k].x, sgv[k].y, sgv[k].z, sgv[k].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = 4 * f + j;
                    if (i < 256 * NO) gsm[(i / NO) * GS + (i % NO)] = (i < nt * NO) ? e[j] : 0.f;
                }
            }
        }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) load_tile(tile);
    __syncthreads();                       // ws / gsm zero fill visible
    if (tile < ntiles) write_tile(tile);
    __syncthreads();
    while (tile < ntiles) {
        const int next = tile + gridDim.x, nextc = min(next, ntiles - 1);     // clamped: loaded (valid memory), never written
        const int b = tile / tilesPerClip, t0 = (tile % tilesPerClip) * 256;
        const int nt = min(256, T - t0);
        // ---- dx tile: D[c][t]
        {
            f32x16 acc[2][2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][n2][r] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const float a0 = ws[(2 * s + half) * 64 + l31], a1 = ws[(2 * s + half) * 64 + 32 + l31];
                const float b0 = gsm[(wave * 64 + l31) * GS + 2 * s + half], b1 = gsm[(wave * 64 + 32 + l31) * GS + 2 * s + half];
                acc[0][0] = mfma32(a0, b0, acc[0][0]); acc[0][1] = mfma32(a0, b1, acc[0][1]);
                acc[1][0] = mfma32(a1, b0, acc[1][0]); acc[1][1] = mfma32(a1, b1, acc[1][1]);
            }
            float* dxb = dx + (size_t)b * 64 * T + t0;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int c = mt * 32 + mfma_row(r, half), tt = wave * 64 + n2 * 32 + l31;
                        if (tt < nt) dxb[(size_t)c * T + tt] = acc[mt][n2][r];
                    }
        }
        // ---- dw: D[o][c] += sum over this wave's 64 steps
        {
            const float* ap = gsm + (wave * 64 + half) * GS + l31;
            const float* bp = xs + l31 * XS + wave * 64 + half;
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                if (s < 16 + NGV) load_piece(nextc, s);
                const float a = ap[2 * s * GS];
                accw[0] = mfma32(a, bp[2 * s], accw[0]);
                accw[1] = mfma32(a, bp[32 * XS + 2 * s], accw[1]);
            }
        }
        if (tid < 8 * 32 && (tid & 31) < NO) {          // bias sums: output tid & 31, 32-step segment tid >> 5 (8 short loops, not one long one)
            const float* gp = gsm + (tid >> 5) * 32 * GS + (tid & 31);
#pragma unroll 8
            for (int tt = 0; tt < 32; ++tt) accb += gp[tt * GS];
        }
        __syncthreads();
        if (next < ntiles) write_tile(next);
        __syncthreads();
        tile = next;
    }
    // fixed-order reduction of the four waves' dw tiles through LDS
    __syncthreads();
    float* red = xs;     // [32][64]
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = mfma_row(r, half), c = n2 * 32 + l31;
                    red[o * 64 + c] = (wv == 0) ? accw[n2][r] : red[o * 64 + c] + accw[n2][r];
                }
        }
        __syncthreads();
    }
    float* out = partial + (size_t)blockIdx.x * (NO * 64 + NO);
    for (int i = tid; i < NO * 64; i += 256) out[i] = red[i];
    __syncthreads();
    red[tid] = ((tid & 31) < NO) ? accb : 0.f;          // [8 segments][32]
    __syncthreads();
    if (tid < NO)
        out[NO * 64 + tid] = ((red[tid] + red[32 + tid]) + (red[64 + tid] + red[96 + tid])) +
                             ((red[128 + tid] + red[160 + tid]) + (red[192 + tid] + red[224 + tid]));
}

}  // namespace

extern "C" {

int wm_stem_fwd(const float* s, const float* w, const float* bias, float* y, int B, int T, hipStream_t stream) {
    if (T & 3) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(stem_fwd_kernel, dim3(B * ((T + 1023) / 1024)), dim3(256), 0, stream, s, w, bias, y, T);
    WM_CHECK_LAUNCH();
    return 0;
}

#ifndef WM_STEM_WGS_PER_CU
#define WM_STEM_WGS_PER_CU 2      // 75 KB of LDS per workgroup: two fit a CU, one's matrix phase runs under the other's loads / ds phase
#endif
// partial: >= 512*512 floats of scratch.  ds may be NULL (Generator stem: the clip is data); only clips [0, nds) get a ds row
// (Detector stem: the clean half of [watermarked; clean] needs no input gradient).
int wm_stem_bwd(const float* g, const float* s, const float* w, float* ds, float* partial, float* dw, float* db, int B,
                int T, int nds, int accumulate, hipStream_t stream) {
    if (T & 3) return (int)hipErrorInvalidValue;
    constexpr size_t lds = (size_t)(64 * 264 + 296 + 512 + 4 * 256) * sizeof(float);
    static wm::DevOnce attr_done;
    if (!wm::dev_done(attr_done)) {
        WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(stem_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        wm::dev_mark(attr_done);
    }
    const int ntiles = B * ((T + 255) / 256);
    const int grid = ntiles < WM_STEM_WGS_PER_CU * kNumCU ? ntiles : WM_STEM_WGS_PER_CU * kNumCU;
    hipLaunchKernelGGL(stem_bwd_kernel, dim3(grid), dim3(256), lds, stream, g, s, w, ds, partial, B, T, nds);
    WM_CHECK_LAUNCH();
    hipLaunchKernelGGL(stem_reduce_kernel, dim3(2), dim3(256), 0, stream, (const float*)partial, grid, dw, db, accumulate);
    WM_CHECK_LAUNCH();
    return 0;
}

int wm_head1_fwd(const float* x, const float* w, const float* bias, float* y, int B, int T, hipStream_t stream) {
    if (T & 3) return (int)hipErrorInvalidValue;
    const int total4 = B * (T / 4);
    hipLaunchKernelGGL(head1_fwd_kernel, dim3((total4 + 255) / 256), dim3(256), 0, stream, x, w, bias, y, T / 4, total4);
    WM_CHECK_LAUNCH();
    return 0;
}

// partial: >= 1024*65 floats.  dwb: [65] = dw[64] followed by db[1] (two separate tensors on the host side).
int wm_head1_bwd(const float* g, const float* x, const float* w, float* dx, float* partial, float* dw, float* db, int B,
                 int T, int accumulate, hipStream_t stream) {
    if (T & 3) return (int)hipErrorInvalidValue;
    const int total4 = B * (T / 4);
    int grid = (total4 + 255) / 256;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(head1_bwd_kernel, dim3(grid), dim3(256), 0, stream, g, x, w, dx, partial, T / 4, total4);
    WM_CHECK_LAUNCH();
    // partial [grid][65]: first 64 -> dw, last -> db
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, stream, (const float*)partial, grid, 65, 64, dw, accumulate);
    WM_CHECK_LAUNCH();
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, stream, (const float*)partial + 64, grid, 65, 1, db, accumulate);
    WM_CHECK_LAUNCH();
    return 0;
}

int wm_headN_fwd(const float* x, const float* w, const float* bias, float* y, int B, int T, int NO, hipStream_t stream) {
    const int grid = B * ((T + 255) / 256);
    if (NO == 17) hipLaunchKernelGGL(headN_fwd_kernel<17>, dim3(grid), dim3(256), 0, stream, x, w, bias, y, T);
    else if (NO == 1) hipLaunchKernelGGL(headN_fwd_kernel<1>, dim3(grid), dim3(256), 0, stream, x, w, bias, y, T);
    else return (int)hipErrorInvalidValue;
    WM_CHECK_LAUNCH();
    return 0;
}

// partial: >= 256*(NO*64+NO) floats
int wm_headN_bwd(const float* g, const float* x, const float* w, float* dx, float* partial, float* dw, float* db, int B,
                 int T, int NO, int accumulate, hipStream_t stream) {
    const int ntiles = B * ((T + 255) / 256);
    const int grid = ntiles < kNumCU ? ntiles : kNumCU;
    const size_t lds = (size_t)(64 * 257 + 256 * 33 + 32 * 64) * sizeof(float);
    if (NO == 17) {
        static wm::DevOnce done;
        if (!wm::dev_done(done)) { WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(headN_bwd_kernel<17>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); wm::dev_mark(done); }
        hipLaunchKernelGGL(headN_bwd_kernel<17>, dim3(grid), dim3(256), lds, stream, g, x, w, dx, partial, B, T);
    } else if (NO == 1) {
        static wm::DevOnce done;
        if (!wm::dev_done(done)) { WM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(headN_bwd_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); wm::dev_mark(done); }
        hipLaunchKernelGGL(headN_bwd_kernel<1>, dim3(grid), dim3(256), lds, stream, g, x, w, dx, partial, B, T);
    } else return (int)hipErrorInvalidValue;
    WM_CHECK_LAUNCH();
    const int n = NO * 64 + NO;
    // partial rows are [NO*64 weights | NO biases]; reduce the two pieces with matching row stride
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((NO * 64 + 63) / 64), dim3(256), 0, stream, (const float*)partial, grid, n, NO * 64, dw, accumulate);
    WM_CHECK_LAUNCH();
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, stream, (const float*)partial + NO * 64, grid, n, NO, db, accumulate);
    WM_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
