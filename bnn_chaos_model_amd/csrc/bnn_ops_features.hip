// bnn_ops_features.hip -- pre-path feature packing + standardisation (figures/spock/regression.py:183-213, :144-145).
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

// data_setup_kernel + StandardScaler.transform + .float() (figures/spock/regression.py:183-213, :144-145).
// HBM-bound by design: 208 B read and 164 B (+ 328 B with X64) written per row.  A workgroup takes PACK_ROWS rows:
//   A. the rows' 26 raw doubles come in with fully coalesced 8-byte loads (one contiguous run per workgroup) into LDS;
//   B. the packed float64 row is built in LDS by DENSE task lists: (row, angle) pairs -- one float64 sincos each, every lane of a wave
//      busy (a thread per raw column left 9 of 32 lanes in the sincos code) -- then (row, plain column) pairs;
//   C. the 41-column rows go out as one contiguous run per workgroup: standardise in float64 (the scaler's own (v - mean) / scale),
//      round to float, coalesced 4-byte stores (8-byte stores for X64).
constexpr int PACK_ROWS = 64;
__constant__ int8_t PACK_ANGLE[9] = {11, 12, 13, 17, 18, 19, 23, 24, 25};   // raw columns expanded to (cos, sin) (regression.py:197-206)
// output column of raw column j (j < 29: the 26 series + 3 masses): every angle before it adds one column; flags follow at 38..40
__host__ __device__ inline int pack_out_col(int j) {
    return j + (j > 11 ? (j < 14 ? j - 11 : 3) : 0) + (j > 17 ? (j < 20 ? j - 17 : 3) : 0) + (j > 23 ? (j < 26 ? j - 23 : 3) : 0);
}
__global__ __launch_bounds__(256) void bnn_feature_pack_kernel(const double* __restrict__ ts, const double* __restrict__ mass, int64_t N, int T,
                                                               const double* __restrict__ mean, const double* __restrict__ scale,
                                                               double* __restrict__ X64, float* __restrict__ x32) {
    __shared__ double raw[PACK_ROWS * 26];
    __shared__ double pk[PACK_ROWS * F];
    __shared__ double ms[2 * F];                                            // the scaler's mean | scale
    const int tid = threadIdx.x;
    const int64_t rows = N * T, row0 = (int64_t)blockIdx.x * PACK_ROWS;
    const int nr = (int)(rows - row0 < PACK_ROWS ? rows - row0 : PACK_ROWS);
    const int64_t n0 = row0 / T;                                            // system of the block's first row (one 64-bit division per block)
    const int t0 = (int)(row0 - n0 * T);
    if (x32 && tid < 2 * F) ms[tid] = tid < F ? mean[tid] : scale[tid - F];
    const double* src = ts + row0 * 26;
    for (int i = tid; i < nr * 26; i += 256) {
        const double v = src[i];
        raw[i] = isfinite(v) ? v : 0.0;                                    // nan_to_num(posinf=0, neginf=0) (:195); the flags below read src again
    }
    __syncthreads();
    for (int t = tid; t < nr * 9; t += 256) {                               // B1: angles
        const int r = t / 9, a = t - 9 * r, j = PACK_ANGLE[a];
        double sn, cs;
        sincos(raw[r * 26 + j], &sn, &cs);
        const int o = pack_out_col(j);
        pk[r * F + o] = cs;
        pk[r * F + o + 1] = sn;
    }
    for (int t = tid; t < nr * 23; t += 256) {                              // B2: 17 plain series columns, 3 masses, 3 flags
        const int r = t / 23, c = t - 23 * r;
        double v;
        int o;
        if (c < 17) {
            const int j = c < 11 ? c : (c < 14 ? c + 3 : c + 6);            // raw columns 0..10, 14..16, 20..22
            v = raw[r * 26 + j];
            o = pack_out_col(j);
        } else if (c < 20) {
            const double m = mass[(n0 + (t0 + r) / T) * 3 + (c - 17)];
            v = isfinite(m) ? m : 0.0;
            o = pack_out_col(26 + (c - 17));
        } else {                                                            // isnotfinite flags of raw columns 3, 6, 7 (:191-193)
            const int j = c == 20 ? 3 : c == 21 ? 6 : 7;
            v = (double)!isfinite(src[r * 26 + j]);
            o = 38 + (c - 20);
        }
        pk[r * F + o] = v;
    }
    __syncthreads();
    for (int i = tid; i < nr * F; i += 256) {                               // C: one contiguous run per workgroup
        const int col = i % F;
        const double v = pk[i];
        if (X64) X64[row0 * F + i] = v;
        if (x32) x32[row0 * F + i] = (float)((v - ms[col]) / ms[F + col]);
    }
}

// Already packed X [N,T,41] float64: standardise only, one thread per element.
__global__ void bnn_standardise_kernel(const double* __restrict__ Xin, int64_t n, const double* __restrict__ mean, const double* __restrict__ scale,
                                       double* __restrict__ X64, float* __restrict__ x32) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int col = (int)(i % F);
    const double v = Xin[i];
    if (X64) X64[i] = v;
    if (x32) x32[i] = (float)((v - mean[col]) / scale[col]);
}

extern "C" {

int bnn_feature_pack_f64(const double* tseries, const double* mass, const double* X64_in, int64_t N, int32_t T, const double* mean,
                         const double* scale, double* X64_out, float* x32_out, void* stream) {
    if (N < 0 || T < 1) return fail(BNN_ERR_INVALID, "bad N/T");
    if (N == 0) return 0;
    if (!tseries && !X64_in) return fail(BNN_ERR_INVALID, "need tseries (+mass) or X64_in");
    if (tseries && !mass) return fail(BNN_ERR_INVALID, "tseries needs mass");
    if (!X64_out && !x32_out) return fail(BNN_ERR_INVALID, "no output requested");
    if (x32_out && (!mean || !scale)) return fail(BNN_ERR_INVALID, "x32_out needs mean and scale");
    const int64_t rows = N * T;
    if (tseries) {
        const int64_t nblk = (rows + PACK_ROWS - 1) / PACK_ROWS;
        if (nblk > 0x7fffffffLL) return fail(BNN_ERR_RANGE, "too many rows for one launch");
        hipLaunchKernelGGL(bnn_feature_pack_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, tseries, mass, N, (int)T, mean, scale,
                           X64_out, x32_out);
    } else {
        const int64_t total = rows * F;
        if ((total + 255) / 256 > 0x7fffffffLL) return fail(BNN_ERR_RANGE, "too many rows for one launch");
        hipLaunchKernelGGL(bnn_standardise_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X64_in, total, mean,
                           scale, X64_out, x32_out);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
