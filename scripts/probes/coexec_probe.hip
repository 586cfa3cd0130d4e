// Does v_mfma_f32_16x16x4_f32 co-execute with VALU work of ANOTHER wave on the same SIMD?
// 512-thread workgroups (2 waves per SIMD): waves 0-3 run role A, waves 4-7 role B.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

// role: 0 idle, 1 = MFMA f32 stream, 2 = VALU fma stream, 3 = VALU integer stream (v_max_i32), 4 = bf16 MFMA stream
template <int RA, int RB>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
    const int role = (threadIdx.x < 256) ? RA : RB;
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    f32x4 acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float v[8];
    int iv[8];
    for (int i = 0; i < 8; ++i) { v[i] = a + i; iv[i] = threadIdx.x + i; }
    if (role == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < 3; ++i) acc[i] = MF(a, b, acc[i]);
        }
    } else if (role == 2) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 24; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], b, a);   // 192 VALU per iter (8 cycles per MFMA slot)
        }
    } else if (role == 3) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 24; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) iv[i] = max(iv[i] + 1, k);
        }
    }
    float s = 0;
    for (int i = 0; i < 3; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i] + iv[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int RA, int RB>
void run(const char* name) {
    int nblk = 256, iters = 20000;
    float* out;
    (void)hipMalloc(&out, nblk * 512 * sizeof(float));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<RA, RB>), dim3(nblk), dim3(512), 0, 0, out, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<RA, RB>), dim3(nblk), dim3(512), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %.3f ms  (%.1f cycles per iteration @2.4GHz; 24 MFMA = 768)\n", name, ms, ms * 1e-3 * 2.4e9 / iters);
    (void)hipFree(out);
}

int main() {
    run<1, 0>("A: mfma f32 alone");
    run<2, 0>("A: valu fma x192 alone");
    run<3, 0>("A: valu int x384 alone");
    run<1, 1>("A: mfma | B: mfma");
    run<1, 2>("A: mfma | B: valu fma");
    run<1, 3>("A: mfma | B: valu int");
    run<2, 2>("A: valu fma | B: valu fma");
    return 0;
}
