// bnn_ops_draw.hip -- SWAGModel.sample_weights for J draws (spock_reg_model.py:815-838) and the Philox fills (the normals the kernels generate,
// written out for inspection).
// One of the translation units of libbnn_chaos_hip.so (bnn_internal.h lists them); entry points declared in include/bnn_chaos_hip.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#include "bnn_abi_common.h"
#include "bnn_common.hip.h"
#include "bnn_stats.hip.h"

using namespace bnn;

__global__ void bnn_philox_fill_kernel(int kind, uint64_t seed, int64_t id0, int64_t n_rows, int64_t B, int64_t sys0, int width, int aux,
                                       float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (kind == 0 || kind == 1) {
        int64_t total = n_rows * width;
        if (i >= total) return;
        int64_t row = i / width;
        int el = (int)(i % width);
        out[i] = philox_z(kind == 0 ? TAG_Z1 : TAG_Z2, id0 + row, el, seed);
    } else if (kind == 2 || kind == 4) {   // eps [n_rows,B,2,L] with L = width (0: 20); eps_sum [n_rows,B,SM] with SM = width (0: 40)
        const int SM = kind == 4 ? (width > 0 ? width : S2) : 2 * (width > 0 ? width : L);
        int64_t total = n_rows * B * SM;
        if (i >= total) return;
        int el = (int)(i % SM);
        int64_t sys = (i / SM) % B, row = i / ((int64_t)SM * B);
        out[i] = philox_sys4(kind == 2 ? TAG_EPS : TAG_SUM, id0 + row, sys0 + sys, el >> 2, seed)[el & 3];
    } else if (kind == 5) {  // candidates of the truncated-normal draw [n_rows, B, nsamp = width] (bnn_stats.hip.h)
        int64_t total = n_rows * B * width;
        if (i >= total) return;
        int k = (int)(i % width);
        int64_t sys = (i / width) % B, row = i / ((int64_t)width * B);
        out[i] = philox_sys4(TAG_TNS, id0 + row, sys0 + sys, k >> 2, seed)[k & 3];
    } else if (kind == 6) {  // survival level of the prior draw [n_rows, B], uniform on (0, 1]
        int64_t total = n_rows * B;
        if (i >= total) return;
        int64_t sys = i % B, row = i / B;
        const uint4 q = philox4x32_10(philox_sys_ctr(TAG_US, id0 + row, sys0 + sys, 0), make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
        out[i] = ((float)(q.x >> 8) + 1.0f) * 5.9604644775390625e-8f;
    } else {  // kind 3: eps_in [n_rows, B, T = width, NF = aux (0: 41)]: block t * ceil(NF/6) + col/6, normal col%6 (bnn_common.hip.h)
        const int T = width, NF = aux > 0 ? aux : F, nblk = (NF + NIN_PER_BLOCK - 1) / NIN_PER_BLOCK;
        int64_t per = (int64_t)T * NF, total = n_rows * B * per;
        if (i >= total) return;
        int col = (int)(i % NF), t = (int)((i / NF) % T);
        int64_t sys = (i / per) % B, row = i / (per * B);
        float n6[6];
        philox_in6(id0 + row, sys0 + sys, t * nblk + col / NIN_PER_BLOCK, seed, n6);
        const int j = col % NIN_PER_BLOCK;
        out[i] = j == 0 ? n6[0] : j == 1 ? n6[1] : j == 2 ? n6[2] : j == 3 ? n6[3] : j == 4 ? n6[4] : n6[5];
    }
}

__global__ void bnn_philox_raw_kernel(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, int64_t n,
                                      uint32_t* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint4 r = philox4x32_10(make_uint4(c0 + (uint32_t)i, c1, c2, c3), make_uint2(k0, k1));
    out[i * 4] = r.x; out[i * 4 + 1] = r.y; out[i * 4 + 2] = r.z; out[i * 4 + 3] = r.w;
}

// SWAGModel.sample_weights for J draws: grid.x = (draw, 256-row slice of the parameter vector), so J is bounded only by 2^31 / 30.
// D = length of the flat parameter vector (7583; 7665 with fix_megno).
__global__ __launch_bounds__(256) void bnn_swag_draw_kernel(const float* __restrict__ w_avg, const float* __restrict__ w2_avg,
                                                            const float* __restrict__ pre_D, int D, int S, int K,
                                                            const int32_t* __restrict__ seed_idx, const float* __restrict__ z1,
                                                            const float* __restrict__ z2, float c1, float c2, float scale,
                                                            uint64_t seed, int64_t draw_id0, float* __restrict__ W_out) {
    __shared__ float slabs[4 * SLAB];
    __shared__ float zsh[MAXK_DRAW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int DRAW_SLICES = (D + 255) / 256;
    const int64_t e = blockIdx.x / DRAW_SLICES;
    const int slice = blockIdx.x % DRAW_SLICES;
    int s = seed_idx[e];
    const bool bad = (s < 0 || s >= S);
    if (bad) s = 0;
    for (int k = threadIdx.x; k < K; k += 256) zsh[k] = z2 ? z2[e * K + k] : philox_z(TAG_Z2, draw_id0 + e, k, seed);
    const int i0 = (slice * 4 + wave) * 64;
    const float* pd = pre_D + (int64_t)s * D * K;
    const int i = i0 + lane;
    const bool live = i < D;
    const float wa = live ? w_avg[(int64_t)s * D + i] : 0.0f;
    float w = 0.0f, dot = 0.0f;
    if (live) {
        const float z1v = z1 ? z1[e * (int64_t)D + i] : philox_z(TAG_Z1, draw_id0 + e, i, seed);
        w = draw_head(wa, w2_avg[(int64_t)s * D + i], z1v, c1);
    }
    for (int kc = 0; kc < K; kc += MAXK) {   // the deviation columns, 32 at a time through the slab; the dot product runs on in k order
        const int Kc = K - kc < MAXK ? K - kc : MAXK;
        if (kc) __syncthreads();
        if (i0 < D) draw_stage(pd, i0, D, K, kc, Kc, lane, slabs + wave * SLAB);
        __syncthreads();
        if (live) dot = draw_dot(slabs + wave * SLAB + lane * Kc, wa, zsh + kc, Kc, dot);
    }
    if (live) W_out[e * (int64_t)D + i] = bad ? __builtin_nanf("") : draw_finish(w, dot, c2, scale);
}

int bnn::draw_consts(int K, float scale, float* c1, float* c2, int kmax) {
    if (K < 2 || K > kmax) return fail(BNN_ERR_RANGE, kmax == MAXK ? "SWAG rank K above 32 needs the draw-once form (W_workspace): the in-prologue draw takes K in [2, 32]"
                                                                  : "SWAG rank K must be in [2, 256]");
    *c1 = (float)((double)scale * (1.0 / std::sqrt(2.0)));  // scale * (1.0/np.sqrt(2.0)), :834
    *c2 = (float)std::sqrt(2.0 * (K - 1));                   // np.sqrt(2*(K-1)), :835
    return 0;
}

extern "C" {

int bnn_swag_draw_f32(const bnn_plan* plan, const float* w_avg, const float* w2_avg, const float* pre_D, int32_t S, int32_t K,
                      const int32_t* seed_idx, int32_t J, const float* z1, const float* z2, float scale, uint64_t philox_seed,
                      int64_t draw_id0, float* W_out, void* stream) {
    if (J == 0) return 0;
    if (!plan || !w_avg || !w2_avg || !pre_D || !seed_idx || !W_out) return fail(BNN_ERR_INVALID, "NULL argument");
    if ((z1 == nullptr) != (z2 == nullptr)) return fail(BNN_ERR_INVALID, "z1 and z2 must both be given or both be NULL");
    if (S < 1 || J < 0) return fail(BNN_ERR_INVALID, "bad S/J");
    float c1, c2;
    int rc = draw_consts(K, scale, &c1, &c2, MAXK_DRAW);
    if (rc) return rc;
    if (J == 0) return 0;
    const int d = plan->d, slices = (d + 255) / 256;
    if ((int64_t)J * slices > 0x7fffffffLL) return fail(BNN_ERR_RANGE, "too many draws for one launch");
    dim3 grid((unsigned)((int64_t)J * slices)), block(256);
    hipLaunchKernelGGL(bnn_swag_draw_kernel, grid, block, 0, (hipStream_t)stream, w_avg, w2_avg, pre_D, d, S, K, seed_idx, z1, z2, c1,
                       c2, scale, philox_seed, draw_id0, W_out);
    HIP_TRY(hipGetLastError());
    return 0;
}

int bnn_philox_normal_f32(int32_t kind, uint64_t philox_seed, int64_t id0, int64_t n_rows, int64_t B, int64_t system_id0, int32_t width,
                          int32_t n_features, float* out, void* stream) {
    if (!out || kind < 0 || kind > 6 || n_rows < 0 || width < 0) return fail(BNN_ERR_INVALID, "bad argument");
    if (kind == 3 && n_features != 0 && n_features != F && n_features != 2 * F) return fail(BNN_ERR_INVALID, "n_features must be 41 or 82");
    const int NF = n_features > 0 ? n_features : F;
    int64_t total = kind == 2 ? n_rows * B * 2 * (int64_t)(width > 0 ? width : L) : kind == 4 ? n_rows * B * (int64_t)(width > 0 ? width : S2)
                    : kind == 3 ? n_rows * B * (int64_t)width * NF : kind == 5 ? n_rows * B * (int64_t)width : kind == 6 ? n_rows * B : n_rows * (int64_t)width;
    if (total == 0) return 0;
    if ((total + 255) / 256 > 0x7fffffffLL) return fail(BNN_ERR_RANGE, "too many normals for one launch");
    hipLaunchKernelGGL(bnn_philox_fill_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, kind, philox_seed,
                       id0, n_rows, B, system_id0, width, NF, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

int bnn_philox_raw_u32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, int64_t n, uint32_t* out,
                       void* stream) {
    if (!out || n < 0) return fail(BNN_ERR_INVALID, "bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(bnn_philox_raw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, c0, c1, c2, c3, k0, k1,
                       n, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
