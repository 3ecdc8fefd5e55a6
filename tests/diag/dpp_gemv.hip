// Micro-benchmark for the recurrent GEMV of the 16 000-step LSTM (one gate row per lane, 64 products per lane and step):
//   A  32 v_readlane -> SGPR pairs + 32 v_pk_fma_f32            (what lstm_fwd / lstm_bwd do)
//   B  64 v_fmac_f32_dpp row_newbcast:i on four row-replicated h registers (no lane reads, no scalar operands)
//   C  64 v_fmac_f32 with SGPR operands (64 v_readlane)
// One wave per SIMD, 4 waves per workgroup, 256 workgroups; cycles per "step" by s_memtime; results checked against the host.
// build: hipcc --offload-arch=gfx950 -O3 tests/diag/dpp_gemv.hip -o tests/diag/dpp_gemv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int I>
__device__ __forceinline__ void fmac_bcast(float& acc, float H, float w) {
    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(H), "v"(w), "n"(I));
}
template <int J, int I>
__device__ __forceinline__ void fmac16(float (&acc)[4], const float (&H)[4], const float (&w)[64]) {
    if constexpr (I < 16) {
        fmac_bcast<I>(acc[I & 3], H[J], w[16 * J + I]);
        fmac16<J, I + 1>(acc, H, w);
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void gemv(const float* __restrict__ W, const float* __restrict__ h0, float* __restrict__ out,
                                            unsigned long long* cyc, int steps) {
    const int lane = threadIdx.x & 63, row = threadIdx.x;       // gate row of this lane
    float w[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) w[k] = W[row * 64 + k];
    float hv = h0[lane];                                          // lane k holds h[k]
    float H[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) H[j] = h0[16 * j + (lane & 15)];  // row-replicated: H[j][lane] = h[16 j + lane % 16]
    float res = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; ++s) {
        float sum;
        if (MODE == 0) {
            v2f a0 = {0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
#pragma unroll
            for (int k = 0; k < 64; k += 8) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned lo = (unsigned)__builtin_amdgcn_readlane(__float_as_int(hv), k + 2 * q);
                    const unsigned hi = (unsigned)__builtin_amdgcn_readlane(__float_as_int(hv), k + 2 * q + 1);
                    const unsigned long long pr = ((unsigned long long)hi << 32) | lo;
                    v2f& a = q == 0 ? a0 : q == 1 ? a1 : q == 2 ? a2 : a3;
                    const v2f ww = {w[k + 2 * q], w[k + 2 * q + 1]};
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(ww), "s"(pr));
                }
            }
            const v2f t = (a0 + a1) + (a2 + a3);
            sum = t.x + t.y;
        } else if (MODE == 1) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            fmac16<0, 0>(acc, H, w); fmac16<1, 0>(acc, H, w); fmac16<2, 0>(acc, H, w); fmac16<3, 0>(acc, H, w);
            sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        } else {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 64; ++k) {
                const float hk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hv), k));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[k & 3]) : "s"(hk), "v"(w[k]));
            }
            sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        }
        res += sum;
        // keep the chain dependent on the result without changing the operands' values
        const float z = res * 0.f;
        hv += z;
#pragma unroll
        for (int j = 0; j < 4; ++j) H[j] += z;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = res / steps;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const int steps = 2000, NB = 256;
    std::vector<float> W(256 * 64), h(64), ref(256);
    for (int i = 0; i < 256 * 64; ++i) W[i] = (float)((i * 2654435761u >> 8) % 2001) / 1000.f - 1.f;
    for (int i = 0; i < 64; ++i) h[i] = (float)((i * 40503u) % 997) / 500.f - 1.f;
    for (int r = 0; r < 256; ++r) { double s = 0; for (int k = 0; k < 64; ++k) s += (double)W[r * 64 + k] * h[k]; ref[r] = (float)s; }
    float *dW, *dh, *dout; unsigned long long* dc;
    hipMalloc(&dW, W.size() * 4); hipMalloc(&dh, 256); hipMalloc(&dout, NB * 256 * 4); hipMalloc(&dc, NB * 8);
    hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dh, h.data(), 256, hipMemcpyHostToDevice);
    const char* names[3] = {"A 32 readlane + 32 pk_fma (SGPR pair operands)", "B 64 v_fmac_f32_dpp row_newbcast", "C 64 readlane + 64 v_fmac (SGPR operand)"};
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(gemv<0>, dim3(NB), dim3(256), 0, 0, dW, dh, dout, dc, steps);
            if (mode == 1) hipLaunchKernelGGL(gemv<1>, dim3(NB), dim3(256), 0, 0, dW, dh, dout, dc, steps);
            if (mode == 2) hipLaunchKernelGGL(gemv<2>, dim3(NB), dim3(256), 0, 0, dW, dh, dout, dc, steps);
            hipDeviceSynchronize();
        }
        std::vector<float> out(NB * 256); std::vector<unsigned long long> c(NB);
        hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, NB * 8, hipMemcpyDeviceToHost);
        double err = 0, cyc = 0;
        for (int r = 0; r < 256; ++r) err = fmax(err, fabs(out[r] - ref[r]));
        for (int b = 0; b < NB; ++b) cyc += (double)c[b] / NB;
        printf("%-52s  %.1f cycles per 64-product step, max |err| %.2e\n", names[mode], cyc / steps, err);
    }
    return 0;
}
