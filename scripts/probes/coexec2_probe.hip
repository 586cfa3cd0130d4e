// Which vector instruction classes co-execute with which matrix-pipe instructions when they come from the OTHER wave
// of the same SIMD?  512-thread workgroups (2 waves per SIMD): waves 0-3 run role RA, waves 4-7 role RB.
// Build: hipcc --offload-arch=gfx950 -O3 -o coexec2_probe coexec2_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

enum Role { IDLE = 0, MFMA_F32_16 = 1, VALU_FMA = 2, VALU_INT = 3, MFMA_BF16_16 = 4, MFMA_F32_4 = 5, TRANS = 6, MAD64 = 7, CVT_BF16 = 8, CVT_U32 = 9, VALU_MAXF = 10 };

template <int ROLE>
__device__ __forceinline__ float body(int iters, int tid) {
    float a = tid * 0.001f, b = 1.0f + tid * 0.002f;
    float s = 0;
    if constexpr (ROLE == MFMA_F32_16) {
        f32x4 acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < 3; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);  // 24 x 32 cycles
        for (int i = 0; i < 3; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else if constexpr (ROLE == MFMA_F32_4) {
        f32x4 acc[12];
        for (int i = 0; i < 12; ++i) acc[i] = (f32x4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < 12; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);  // 96 x 8 cycles
        for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else if constexpr (ROLE == MFMA_BF16_16) {
        f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        bf16x8 av, bv;
        for (int i = 0; i < 8; ++i) { av[i] = (short)(0x3f80 + tid + i); bv[i] = (short)(0x3f00 + tid * 3 + i); }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 12; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[i], 0, 0, 0);  // 48 x 16 cycles
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else if constexpr (ROLE == VALU_FMA) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = a + i;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 24; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(a));  // 192 per iteration
        for (int i = 0; i < 8; ++i) s += v[i];
    } else if constexpr (ROLE == VALU_MAXF) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = a + i;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 24; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
        for (int i = 0; i < 8; ++i) s += v[i];
    } else if constexpr (ROLE == VALU_INT) {
        int iv[8];
        for (int i = 0; i < 8; ++i) iv[i] = tid + i;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 24; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(iv[i]) : "v"(tid + k));
        for (int i = 0; i < 8; ++i) s += iv[i];
    } else if constexpr (ROLE == TRANS) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = 1.5f + a + i;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 6; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_log_f32 %0, %0" : "+v"(v[i]));  // 48 per iteration
        for (int i = 0; i < 8; ++i) s += v[i];
    } else if constexpr (ROLE == MAD64) {
        unsigned long long v[8];
        unsigned m = 0xD2511F53u + tid;
        for (int i = 0; i < 8; ++i) v[i] = tid + i;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 6; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = (unsigned long long)(unsigned)v[i] * m + (v[i] >> 32);  // 48 v_mad_u64_u32 per iteration
        for (int i = 0; i < 8; ++i) s += (float)(unsigned)v[i];
    } else if constexpr (ROLE == CVT_BF16) {
        float v[8];
        unsigned o[4] = {0, 0, 0, 0};
        for (int i = 0; i < 8; ++i) v[i] = a + i;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 48; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(o[i]) : "v"(v[2 * i]), "v"(v[2 * i + 1]));  // 192
        for (int i = 0; i < 4; ++i) s += o[i];
    } else if constexpr (ROLE == CVT_U32) {
        unsigned u[8];
        float v[8];
        for (int i = 0; i < 8; ++i) { u[i] = tid * 77 + i; v[i] = 0; }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 24; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(v[i]) : "v"(u[i] + k));
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    return s;
}

template <int RA, int RB>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
    float s = (threadIdx.x < 256) ? body<RA>(iters, threadIdx.x) : body<RB>(iters, threadIdx.x);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int RA, int RB>
float run(const char* name) {
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
    printf("%-44s %8.3f ms  (%.1f cycles per iteration @2.4GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / iters);
    (void)hipFree(out);
    return ms;
}

#define PAIR(M, V) run<M, V>(#M " | " #V)

int main() {
    printf("alone (other half of the workgroup idle):\n");
    run<MFMA_F32_16, IDLE>("mfma f32 16x16x4 x24");
    run<MFMA_F32_4, IDLE>("mfma f32 4x4x1 x96");
    run<MFMA_BF16_16, IDLE>("mfma bf16 16x16x32 x48");
    run<VALU_FMA, IDLE>("v_fma_f32 x192");
    run<VALU_MAXF, IDLE>("v_max_f32 x192");
    run<VALU_INT, IDLE>("v_xor_b32 x192");
    run<TRANS, IDLE>("v_log_f32 x48");
    run<MAD64, IDLE>("v_mad_u64_u32 x48");
    run<CVT_BF16, IDLE>("v_cvt_pk_bf16_f32 x192");
    run<CVT_U32, IDLE>("v_cvt_f32_u32 x192");
    printf("pairs (time = max of the two if they co-execute, sum if they exclude each other):\n");
    PAIR(MFMA_F32_4, VALU_FMA); PAIR(MFMA_F32_4, VALU_MAXF); PAIR(MFMA_F32_4, VALU_INT); PAIR(MFMA_F32_4, TRANS); PAIR(MFMA_F32_4, MAD64);
    PAIR(MFMA_F32_4, CVT_BF16); PAIR(MFMA_F32_4, CVT_U32);
    PAIR(MFMA_F32_16, VALU_FMA); PAIR(MFMA_F32_16, TRANS); PAIR(MFMA_F32_16, MAD64); PAIR(MFMA_F32_16, CVT_U32);
    PAIR(MFMA_BF16_16, VALU_FMA); PAIR(MFMA_BF16_16, VALU_MAXF); PAIR(MFMA_BF16_16, VALU_INT); PAIR(MFMA_BF16_16, TRANS); PAIR(MFMA_BF16_16, CVT_BF16);
    PAIR(MFMA_BF16_16, MFMA_BF16_16); PAIR(MFMA_F32_4, MFMA_F32_4);
    return 0;
}
