// bnn_internal.h -- what the translation units of libbnn_chaos_hip.so share on the HOST side: kernel parameter
// blocks and the launch functions each kernel TU exports to the C ABI (bnn_abi.hip).  No device code here.
//
//   bnn_fwd_k31.hip     forward kernel, 31 live input columns (the v50 mask), quiet      (workspace + in-prologue draw)
//   bnn_fwd_k41.hip     forward kernel, all 41 columns (any other mask), quiet           (workspace + in-prologue draw)
//   bnn_fwd_noisy.hip   forward kernel, forward(noisy_val=True)
//   bnn_fwd_megno.hip   forward kernel forms for hparams['fix_megno'] = True (42-wide summary, d = 7665)
//   bnn_fwd_lowp.hip    reduced-precision forward kernels (bf16 / half matrix pipe; opt-in, configs[4])
//   bnn_fwd_generic.hip, bnn_fwd_generic82.hip   generic forward engine: the network built from hparams (any hidden / latent / depth; 41 resp. 82 features), any T
//   bnn_generic.cpp     host: descriptor of that engine (layers, LDS image, register bucket)
//   bnn_nonfinite.hip   non-finite inputs the reference's way: the once-per-call scan of x and the exact re-evaluation of the listed systems
//   bnn_abi.hip         extern "C" entry points that launch a forward kernel, plans, specialised forms (bnn_abi_common.h lists the C-ABI units)
//   bnn_ops_draw.hip, bnn_ops_reduce.hip, bnn_ops_stats.hip, bnn_ops_features.hip   the small kernels with their entry points: SWAG draw +
//                       Philox fills; moments + regress_nn on a summary; statistics epilogue + quantile sketch; feature packing
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "bnn_generic.h"
#include "bnn_layout.h"

namespace bnn {

struct StatsParams {
    int32_t tn_nsamp;         // candidates per truncated-normal draw (the scripts use 40)
    float tn_left;            // left truncation point (4)
    float prior_thr;          // values >= this are redrawn from the prior (9); +inf: never
    const float* prior_surv;  // [prior_m] survival function of the prior at t_i = prior_thr + i * prior_step (decreasing from 1)
    int32_t prior_m;
    float prior_step;
};

struct FwdParams {
    const float* x;
    int64_t B;
    int32_t T, ntiles;
    int32_t J, nch;
    int64_t csz;  // chunk size = ceil(cB / nch): torch.chunk over the whole batch
    int64_t cB;   // systems of the whole batch the chunks partition (= B unless the batch is sharded over devices)
    int64_t coff; // index of this call's row 0 in that batch
    int32_t spc;  // systems per workgroup (multiple of 64; 16 in the tile-split form)
    int32_t xcd_order;  // 1: XCD k takes the k-th contiguous eighth of the work order (work_item, bnn_common.hip.h)
    int32_t K, S;
    const float* W;  // [J,d] materialised draws (unfused) or nullptr
    const float* w_avg;
    const float* w2_avg;
    const float* pre_D;
    const int32_t* seed_idx;
    const float* z1;
    const float* z2;
    float c1, c2, scale;
    const float* eps;
    const float* eps_in;
    const float* eps_sum;
    uint64_t seed;
    int64_t draw_id0, row_id0, sys_id0;
    float* out;
    float* pre_clamp;
    float* summary;
    float* latents;   // [R,B,T,latent] feature_nn's output per timestep (compute_summary_stats' side effect self.latents, :433), or null;
                      // written by the generic engine only (bnn_feature_nn_f32)
    const int16_t* tab_f2;
    const int16_t* tab_wr;  // feature_nn weight-register gather table [WR<KIN>::NR][64] (bnn_layout.h)
    const float* rcp_tab;   // [i] = 1/(i+1), correctly rounded
    uint64_t zero_mask;
    float std_lo, std_span;
    // fused statistics epilogue (bnn_multiswag_stats_f32): when `sink` is set the (mu, std) pair of every evaluation goes
    // through truncated-normal draw -> prior resampling (bnn_stats.hip.h) and lands in sink[R, B] instead of out[R, B, 2].
    float* sink;
    StatsParams st;
};

// generic forward engine (bnn_generic.hip.h): the common block + the descriptor (device copy owned by the plan) + pool merges
struct GenParams {
    FwdParams f;
    const GenArch* g;
    GenMerge m01, m23, m0123;   // partitions (0,1), (2,3), then the halves; counts from T
    int32_t noisy;              // forward(noisy_val=True): input + summary noise, masked columns keep their weights
};

// exact re-evaluation of the systems a non-finite scan listed (bnn_nonfinite.hip): the forward's own parameter block (inputs, noise,
// outputs, chunk geometry) + the plan's descriptor + the scan record
struct NfxParams {
    FwdParams f;
    const GenArch* g;       // the plan's ahead-of-time descriptor (every column, every layer at its natural shape)
    const int32_t* rec;     // [0] = listed systems, [1] = of which certainly NaN, [4 + i] = (system << 1) | certain
    int32_t noisy;          // forward(noisy_val=True)
    int32_t shortcut;       // 1: "certainly NaN" systems are answered without the evaluation (no summary / latents output was asked for)
    int32_t maxw;           // widest activation vector of feature_nn (incl. the input row): LDS rows of the kernel
};

constexpr int RCP_N = 4096;  // supports T up to 16384 timesteps

// Each returns hipGetLastError() after the launch.  grid = (draw, block-of-systems) pairs, 256 threads.
hipError_t launch_fwd_k31(bool fused, unsigned nblk, hipStream_t st, const FwdParams& p);
hipError_t launch_fwd_k41(bool fused, unsigned nblk, hipStream_t st, const FwdParams& p);
hipError_t launch_fwd_small(bool fused, unsigned nblk, hipStream_t st, const FwdParams& p);   // v50 mask, tile-split form: 16 systems per workgroup (small grids)
hipError_t launch_fwd_noisy(unsigned nblk, hipStream_t st, const FwdParams& p);
hipError_t launch_fwd_stats(bool k31, unsigned nblk, hipStream_t st, const FwdParams& p);  // quiet forward + fused statistics tail
hipError_t launch_fwd_lowp(int precision, unsigned nblk, hipStream_t st, const FwdParams& p);  // bf16 / half matrix pipe (bnn_precision)
hipError_t launch_fwd_megno(bool k31, bool fused, bool noisy, unsigned nblk, hipStream_t st, const FwdParams& p);  // hparams['fix_megno'] layout

hipError_t launch_fwd_generic(const GenArch& g, unsigned nblk, hipStream_t st, const GenParams& P);  // any hparams network, any T >= 2
hipError_t launch_fwd_v50spec(bool noisy, unsigned nblk, hipStream_t st, const GenParams& P);          // the pretrained network's specialised forms (generated unit)

// bnn_nonfinite.hip: rec = the caller's record (int32 [4 + B]); per = T * n_features floats per system
hipError_t launch_nonfinite_scan(const float* x, int64_t B, int64_t per, int F, uint64_t zero_mask, int32_t* rec, hipStream_t st);
hipError_t launch_nonfinite_fixup(const GenArch& g, NfxParams& q, hipStream_t st);

constexpr int MAX_DEVICES = 64;
inline int current_device_slot() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return (dev >= 0 && dev < MAX_DEVICES) ? dev : 0;
}

// Kernels that take more than 64 KB of dynamic LDS need the attribute set once per (function, device): call this in front of the launch.
// (std::call_once: concurrent first launches from several host threads are safe.)
template <auto Kernel>
inline void allow_big_lds() {
    static std::once_flag once[MAX_DEVICES];
    std::call_once(once[current_device_slot()], [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
}

}  // namespace bnn
