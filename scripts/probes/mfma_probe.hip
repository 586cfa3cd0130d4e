// Micro-probe: issue rate of v_mfma_f32_16x16x4_f32 alone and mixed with VALU work, at 1..3 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 mfma_probe.hip -o mfma_probe && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

// MODE 0: pure MFMA, NACC independent accumulators.  MODE 1: layer-like: chain of 24 MFMA (3 acc), then NV VALU ops
// that depend on the accumulators, feeding the next chain (like relu between layers).
template <int MODE, int NACC, int NV>
__global__ __launch_bounds__(256, 2) void probe(float* out, int iters, unsigned long long* cyc) {
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 24 / NACC; ++k)
#pragma unroll
                for (int i = 0; i < NACC; ++i) acc[i] = MF(a, b, acc[i]);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < 3; ++i) acc[i] = MF(a, b, acc[i]);
            // NV dependent VALU ops spread over the 12 accumulator registers (integer max = relu)
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                int bits = __builtin_bit_cast(int, acc[(v / 4) % 3][v % 4]);
                bits = bits > 0 ? bits : (v + 1);
                acc[(v / 4) % 3][v % 4] = __builtin_bit_cast(float, bits);
            }
            b = acc[0][0] * 0.0f + b;  // next chain's B operand depends on the VALU results
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int NACC, int NV>
void run(const char* name, int blocks_per_cu, int threads) {
    int nblk = 256 * blocks_per_cu, iters = 20000;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, nblk * threads * sizeof(float));
    hipMalloc(&cyc, nblk * sizeof(unsigned long long));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MODE, NACC, NV>), dim3(nblk), dim3(threads), 0, 0, out, 100, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, NACC, NV>), dim3(nblk), dim3(threads), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(nblk); hipMemcpy(h.data(), cyc, nblk * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= nblk;
    int waves_per_simd = blocks_per_cu * threads / 256;
    double mfma_per_wave = 24.0 * iters;
    // s_memtime ticks at 100 MHz constant clock on gfx9? report both wall-derived numbers
    double ns_per_mfma_simd = ms * 1e6 / (mfma_per_wave * waves_per_simd);
    printf("%-28s waves/SIMD=%d  wall %.3f ms  -> %.2f ns per MFMA per SIMD (32 cyc @2.4GHz = 13.33 ns); memtime ticks/iter %.1f\n", name,
           waves_per_simd, ms, ns_per_mfma_simd, avg / iters);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0, 3, 0>("pure mfma 3acc", 1, 256);
    run<0, 3, 0>("pure mfma 3acc", 2, 256);
    run<0, 2, 0>("pure mfma 2acc", 1, 256);
    run<0, 2, 0>("pure mfma 2acc", 2, 256);
    run<0, 1, 0>("pure mfma 1acc", 1, 256);
    run<0, 1, 0>("pure mfma 1acc", 2, 256);
    run<1, 3, 0>("chain24 + 0 valu", 1, 256);
    run<1, 3, 0>("chain24 + 0 valu", 2, 256);
    run<1, 3, 12>("chain24 + 12 valu", 1, 256);
    run<1, 3, 12>("chain24 + 12 valu", 2, 256);
    run<1, 3, 12>("chain24 + 12 valu", 3, 256);
    run<1, 3, 24>("chain24 + 24 valu", 1, 256);
    run<1, 3, 24>("chain24 + 24 valu", 2, 256);
    run<1, 3, 48>("chain24 + 48 valu", 1, 256);
    run<1, 3, 48>("chain24 + 48 valu", 2, 256);
    run<1, 3, 48>("chain24 + 48 valu", 3, 256);
    return 0;
}
