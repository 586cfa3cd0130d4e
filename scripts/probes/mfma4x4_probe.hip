// Decode the operand layout of v_mfma_f32_4x4x1_16b_f32 and measure its issue rate, alone and fed from LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF4(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0)

__global__ void layout(float* out) {  // out[p][lane][4]: a = (lane == p), b = lane + 1
    int lane = threadIdx.x;
    for (int p = 0; p < 64; ++p) {
        f32x4 d = MF4(lane == p ? 1.0f : 0.0f, (float)(lane + 1), ((f32x4){0, 0, 0, 0}));
        for (int r = 0; r < 4; ++r) out[(p * 64 + lane) * 4 + r] = d[r];
    }
}

template <int NACC, bool LDSFEED>
__global__ __launch_bounds__(256) void rate(float* out, int iters, const float* wsrc) {
    __shared__ __attribute__((aligned(16))) float w[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) w[i] = wsrc[i];
    __syncthreads();
    int lane = threadIdx.x & 63;
    float b = 1.0f + lane * 0.001f;
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float a0 = lane * 0.01f;
    for (int it = 0; it < iters; ++it) {
        if constexpr (LDSFEED) {
            // per k: 3 x ds_read_b128 (broadcast: 4 distinct 16-B addresses per wave) feed 10 (of 12) MFMAs
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const f32x4* src = reinterpret_cast<const f32x4*>(w) + ((k * 3) * 4 + (lane & 3));
                f32x4 q0 = src[0], q1 = src[4], q2 = src[8];
                float A[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
#pragma unroll
                for (int n = 0; n < NACC; ++n) acc[n] = MF4(A[n % 12], b, acc[n]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int n = 0; n < NACC; ++n) acc[n] = MF4(a0, b, acc[n]);
        }
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, bool LDSFEED>
void run_rate(const char* name, int blocks_per_cu, const float* wsrc) {
    int nblk = 256 * blocks_per_cu, iters = 4000;
    float* out; (void)hipMalloc(&out, nblk * 256 * sizeof(float));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((rate<NACC, LDSFEED>), dim3(nblk), dim3(256), 0, 0, out, 50, wsrc);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((rate<NACC, LDSFEED>), dim3(nblk), dim3(256), 0, 0, out, iters, wsrc);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double n_mfma = 16.0 * NACC * iters * blocks_per_cu;  // per SIMD
    printf("%-30s waves/SIMD=%d  %.3f ms -> %.2f cycles per MFMA per SIMD @2.4GHz (ideal 8)\n", name, blocks_per_cu, ms, ms * 1e-3 * 2.4e9 / n_mfma);
    (void)hipFree(out);
}

int main() {
    float* d; (void)hipMalloc(&d, 64 * 64 * 4 * sizeof(float));
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, d);
    std::vector<float> h(64 * 64 * 4); (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // for A lane p: list (lane, reg, contributing B lane)
    for (int p : {0, 1, 2, 3, 4, 5, 17, 63}) {
        printf("A lane %2d ->", p);
        int cnt = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) { float v = h[(p * 64 + l) * 4 + r]; if (v != 0 && cnt++ < 8) printf(" D[lane %d][reg %d]=Blane%d", l, r, (int)v - 1); }
        printf("  (%d outputs)\n", cnt);
    }
    float* w; (void)hipMalloc(&w, 4096 * 4); (void)hipMemset(w, 0, 4096 * 4);
    run_rate<10, false>("4x4x1 regs, 10 acc", 1, w);
    run_rate<10, false>("4x4x1 regs, 10 acc", 2, w);
    run_rate<5, false>("4x4x1 regs, 5 acc", 1, w);
    run_rate<5, false>("4x4x1 regs, 5 acc", 2, w);
    run_rate<2, false>("4x4x1 regs, 2 acc", 1, w);
    run_rate<1, false>("4x4x1 regs, 1 acc", 1, w);
    run_rate<10, true>("4x4x1 LDS-fed, 10 acc", 1, w);
    run_rate<10, true>("4x4x1 LDS-fed, 10 acc", 2, w);
    run_rate<10, true>("4x4x1 LDS-fed, 10 acc", 3, w);
    run_rate<5, true>("4x4x1 LDS-fed, 5 acc", 2, w);
    return 0;
}
