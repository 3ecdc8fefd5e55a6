// diagnostic (not a test): achieved HBM write bandwidth of the store patterns a conv64 epilogue can use.
// build: hipcc --offload-arch=gfx950 -O3 tests/diag/store_bw.hip -o tests/diag/store_bw ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int T = 16000, B = 256, TPC = 125, NTILES = B * TPC;

template <int MODE>
__global__ __launch_bounds__(256) void store_kernel(float* __restrict__ y, const float* __restrict__ x, float* sink) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int mt = wave & 1, nh = wave >> 1;
    float acc = 0.f;
    for (int tile = blockIdx.x; tile < NTILES; tile += gridDim.x) {
        const int b = tile / TPC, t0 = (tile % TPC) * 128;
        float* yb = y + (size_t)b * 64 * T;
        const float v = (float)tile;
        if (MODE == 0) {            // MFMA C layout, dword per lane: 2 x 128-B segments per instruction
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    yb[(size_t)(32 * mt + (r & 3) + 8 * (r >> 2) + 4 * half) * T + t0 + 64 * nh + 32 * nt + l31] = v;
        } else if (MODE == 1) {     // float4 per lane, 8 rows x 128 B per instruction
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    *reinterpret_cast<float4*>(yb + (size_t)(32 * mt + (l31 & 3) + 8 * k + 4 * half) * T + t0 + 64 * nh + 32 * nt + 4 * (l31 >> 2)) = make_float4(v, v, v, v);
        } else if (MODE == 2) {     // float4 per lane, 2 rows x 512 B per instruction
#pragma unroll
            for (int k = 0; k < 8; ++k)
                *reinterpret_cast<float4*>(yb + (size_t)((tid >> 5) + 8 * k) * T + t0 + 4 * (tid & 31)) = make_float4(v, v, v, v);
        } else if (MODE == 3) {     // read test: the load_tile pattern (8 rows x 128 B per instruction, float4)
            const float* xb = x + (size_t)b * 64 * T;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const size_t o = (size_t)(2 * (wave * 8 + (lane & 7))) * T + t0 + 4 * ((lane >> 3) + 8 * i);
                const float4 a4 = *reinterpret_cast<const float4*>(xb + o), b4 = *reinterpret_cast<const float4*>(xb + o + T);
                acc += a4.x + a4.y + a4.z + a4.w + b4.x + b4.y + b4.z + b4.w;
            }
        } else if (MODE == 4) {     // read test: 2 rows x 512 B per instruction
            const float* xb = x + (size_t)b * 64 * T;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float4 a4 = *reinterpret_cast<const float4*>(xb + (size_t)((tid >> 5) + 8 * k) * T + t0 + 4 * (tid & 31));
                acc += a4.x + a4.y + a4.z + a4.w;
            }
        } else if (MODE == 5) {     // read (pattern 3) + write (pattern 0) together
            const float* xb = x + (size_t)b * 64 * T;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const size_t o = (size_t)(2 * (wave * 8 + (lane & 7))) * T + t0 + 4 * ((lane >> 3) + 8 * i);
                const float4 a4 = *reinterpret_cast<const float4*>(xb + o), b4 = *reinterpret_cast<const float4*>(xb + o + T);
                acc += a4.x + a4.y + a4.z + a4.w + b4.x + b4.y + b4.z + b4.w;
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    yb[(size_t)(32 * mt + (r & 3) + 8 * (r >> 2) + 4 * half) * T + t0 + 64 * nh + 32 * nt + l31] = v;
        } else if (MODE == 6) {     // read (pattern 4) + write (pattern 2)
            const float* xb = x + (size_t)b * 64 * T;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float4 a4 = *reinterpret_cast<const float4*>(xb + (size_t)((tid >> 5) + 8 * k) * T + t0 + 4 * (tid & 31));
                acc += a4.x + a4.y + a4.z + a4.w;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                *reinterpret_cast<float4*>(yb + (size_t)((tid >> 5) + 8 * k) * T + t0 + 4 * (tid & 31)) = make_float4(v, v, v, v);
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

template <int MODE>
int run(const char* name, float* y, float* x, float* sink, double bytes, int grid) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(store_kernel<MODE>, dim3(grid), dim3(256), 0, 0, y, x, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(store_kernel<MODE>, dim3(grid), dim3(256), 0, 0, y, x, sink);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("%-58s grid %5d  %.3f ms  %.2f TB/s\n", name, grid, ms, bytes / ms / 1e9);
    return 0;
}

int main() {
    float *x, *y, *sink;
    const size_t n = (size_t)B * 64 * T;
    CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(x, 0, n * 4)); CK(hipMemset(y, 0, n * 4));
    const double one = (double)n * 4;
    for (int grid : {256, 512, 1024}) {
        run<0>("write dword, MFMA C layout (2 x 128 B / instr)", y, x, sink, one, grid);
        run<1>("write float4, 8 rows x 128 B / instr", y, x, sink, one, grid);
        run<2>("write float4, 2 rows x 512 B / instr", y, x, sink, one, grid);
        run<3>("read  float4, 8 rows x 128 B / instr (load_tile)", y, x, sink, one, grid);
        run<4>("read  float4, 2 rows x 512 B / instr", y, x, sink, one, grid);
        run<5>("read load_tile + write dword C layout", y, x, sink, 2 * one, grid);
        run<6>("read + write, 2 rows x 512 B float4", y, x, sink, 2 * one, grid);
    }
    return 0;
}
