/*
 * bnn_chaos_hip.h -- C ABI of the MI355X (gfx950) MultiSWAG inference library.
 *
 * The reference (MilesCranmer/bnn_chaos_model) is pure Python: it has no FFI.  The boundary it
 * offers for this path is the Python object surface of spock_reg_model.py.
 * The entry points below are the native layer under that surface: one per reference method on the
 * path, each cited.  bnn_chaos_model_amd/_native.py binds them with ctypes.
 *
 * Conventions
 *   - plain pointers and sizes; every data pointer is a DEVICE pointer unless named host_*;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); every call only enqueues
 *     work on it: no allocation, no host synchronisation (plans excepted, see below);
 *   - return value 0 = success, negative = bnn_status; bnn_last_error() returns a message for the
 *     calling thread's most recent failure;
 *   - all arithmetic is IEEE fp32, round-to-nearest, no fast-math.
 *
 * Flat parameter vector (d = 7583 floats), the reference's state_dict order
 * (spock_reg_model.py:734-761): input_noise_logvar[41] | summary_noise_logvar[40] |
 * feature_nn.{0,2,4}.{weight,bias} | regress_nn.{0,2,4}.{weight,bias}.
 */
#ifndef BNN_CHAOS_HIP_H
#define BNN_CHAOS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* v3: bnn_arch carries the network depth (any hparams-built network, not only the pretrained ensemble's); bnn_philox_normal_f32 takes
 * the feature count; bnn_sketch_bins counts the NaN bin (the last hist row); bnn_fragment_table(which = 1) returns the weight-register
 * table; bnn_arch.reserved became fix_megno; bnn_build_flags() names the build; bnn_feature_nn_f32 (the latents side effect);
 * bnn_spec_source / bnn_plan_attach_spec / bnn_plan_spec_attached / bnn_spec_embedded_source (specialised forms of the generic engine);
 * bnn_grid.engine = 2. */
/* v4: non-finite inputs the reference's way -- bnn_nonfinite_record_bytes / bnn_nonfinite_scan_f32 and bnn_grid.nonfinite (the struct grew
 * by one pointer); bnn_gen_params_bytes (layout check for code objects compiled at run time). */
#define BNN_ABI_VERSION 4

enum bnn_status {
    BNN_OK = 0,
    BNN_ERR_INVALID = -1,     /* NULL pointer / negative size / inconsistent arguments            */
    BNN_ERR_UNSUPPORTED = -2, /* a shape no kernel is built for (widths above 128, a form limited to the v50 network, ...) */
    BNN_ERR_HIP = -3,         /* a HIP runtime call failed (message has the hipError string)      */
    BNN_ERR_NO_DEVICE = -4,   /* no gfx950 device visible                                         */
    BNN_ERR_RANGE = -5        /* seed index outside the ensemble, K > 256, ...                    */
};

/* Network description, from the checkpoint's hparams (spock_reg_model.py:343-397).
 * feature_nn = mlp(n_features, latent, hidden, depth_in), regress_nn = mlp(2 latent (+2), 2, hidden, depth_out) with
 * mlp(in_n, out_n, hidden, layers) (:301-321) = ONE Linear(in_n, out_n) for layers == 0, else Linear(in_n, hidden), ReLU,
 * layers x [Linear(hidden, hidden), ReLU], Linear(hidden, out_n).
 * Two engines sit behind every entry point: the pretrained ensemble's network (41, 40, 20, depth 1/1) at
 * T % 4 == 0, T >= 8 runs on kernels whose weights are register-resident; everything else -- hidden, latent and summary width up to
 * 128, any depth with at most 16 Linear modules, 41 or 82 features, any T >= 2 -- on the generic engine (weights streamed from LDS). */
typedef struct bnn_arch {
    int32_t n_features; /* hparams['time_series_features'] x (1 + include_derivatives): 41 or 82 (:346-358) */
    int32_t hidden;     /* hparams['hidden'] (40 in every pretrained checkpoint)                  */
    int32_t latent;     /* hparams['latent'] (20)                                                */
    int32_t fix_megno;  /* hparams['fix_megno'] (spock_reg_model.py:360-362): 1 = the summary is 42 wide ([.. | mean_t, std_t of the
                           raw MEGNO column 7], :480-491, :509-510), summary_noise_logvar [42], regress_nn.0 [40,42], d = 7665;
                           the reference zeroes column 7 as well then (:488-491: set bit 7 of zero_mask); the statistics are taken
                           from the raw column whatever the mask.  0 for every pretrained checkpoint.  (Was `reserved`.) */
    uint64_t zero_mask; /* bit f set: input column f is zeroed before feature_nn
                           (zero_megno/zero_mmr/zero_nan/zero_eplusminus, spock_reg_model.py:452-500) */
    float lowest_std;   /* soft_clamp floor of std: 0.5, or 0.1 with lower_std (:363-365)         */
    float pad;
    int32_t depth_in;   /* hparams['in']  (1): the `layers` argument of feature_nn's mlp() (:359)  */
    int32_t depth_out;  /* hparams['out'] (1): the same for regress_nn (:360)                     */
} bnn_arch;

typedef struct bnn_plan bnn_plan; /* opaque: device-resident operand tables for one bnn_arch */

int bnn_abi_version(void);
const char* bnn_last_error(void);
int bnn_device_count(void); /* number of visible HIP devices, or negative bnn_status */
int bnn_param_count(const bnn_arch* arch); /* d (7583 for the pretrained network; 7665 with fix_megno), or negative */
const char* bnn_build_flags(void); /* "" for the default build; otherwise the extra compile-time switches (e.g. "BNN_ABLATE=4"): a
                                      library that reports anything here is a profiling / A-B variant, not the product */

/* Plans own a few KB of device memory; create/destroy allocate and synchronise, nothing else does. */
int bnn_plan_create(const bnn_arch* arch, bnn_plan** out);
int bnn_plan_destroy(bnn_plan* plan);

/* Specialised forms of the generic engine: the same kernel source compiled for ONE network, every shape a
 * compile-time constant.  bnn_spec_source writes the HIP source text (needs no device; returns its length, or a negative status;
 * compile it with hipcc --genco --offload-arch=gfx950 -I<csrc> and the library's own flags: bnn_chaos_model_amd/specialize.py does);
 * bnn_plan_attach_spec loads the resulting code object into the plan (current device = the plan's); from then on the generic route
 * of every entry point launches it for that `noisy` form.  w8: 1 = eight waves at 256 registers, 0 = at most four at 512,
 * -1 = the builder's choice -- the same w8 and flags must be given to both calls.  The quiet forms are compiled for the plan's column
 * mask as well (layer 0 multiplies the unmasked columns only).  Results are bit-identical to the ahead-of-time form's
 * (same accumulation order); what changes is the schedule.  Replacing an attached form synchronises the device first (launches of
 * the old form may still be queued); callers that tune over several candidates should do so on a plan of their own (specialize.py does)
 * and attach only the winner to a plan other threads are launching on. */
#define BNN_SPEC_POOL_REGS 1 /* flags: the pool's Welford state in registers across the tiles (else in LDS); a tuning choice -- the caller
                                compiles, looks at the code object's scratch size and falls back (specialize.py does) */
#define BNN_SPEC_BLOCK_MAJOR 2 /* flags: the ahead-of-time form's block-major layer routine (fewest registers: the widest networks) instead of
                                  the input-quad-major one with its one-step-ahead weight reads */
#define BNN_SPEC_RESIDENT 4 /* flags: feature_nn's weight registers stay in VGPRs across the tiles (networks of at most 112 such registers,
                               e.g. the pretrained shapes' 74); LDS layout unchanged */
int bnn_spec_source(const bnn_arch* arch, int32_t w8, int32_t noisy, int32_t flags, char* buf, size_t cap);
int bnn_plan_attach_spec(bnn_plan* plan, int32_t noisy, int32_t w8, int32_t flags, const void* image, size_t bytes);
int bnn_plan_spec_attached(const bnn_plan* plan, int32_t noisy); /* 1 / 0 */
/* The pretrained network's own two forms are compiled into the library (bnn_fwd_v50spec.hip, generated: this call returns its text):
 * its ragged series lengths (T % 4 != 0, T < 8 -- everything its register-resident kernels do not take) run on them without any
 * compiler at run time; quiet form under the pretrained column mask, noisy form under any mask. */
int bnn_spec_embedded_source(char* buf, size_t cap);
/* sizeof the parameter block a specialised kernel takes by value: a code object compiled against other headers than the library's own
 * (a stale cache, an A/B build) would read it with another layout -- specialize.py mixes this and bnn_abi_version() into its cache key. */
size_t bnn_gen_params_bytes(void);
/* Accumulation order used by the kernels for Linear layer `layer` (0 .. number of Linear modules - 1, feature_nn's first): `order`
 * receives up to `cap` entries (input indices; the accumulator starts at the bias).  The generic engine's order is the natural one
 * (0, 1, 2, ...) for every layer; the v50 kernels permute regress_nn's.
 * `noisy` selects the 41-column variant used by bnn_forward_f32 with eps_in != NULL.
 * Returns the number of entries.  Lets a test pin a CPU model to the same order. */
int bnn_plan_layer_order(const bnn_plan* plan, int layer, int noisy, int32_t* host_order, int cap);

/* Host-only views of the operand layout (no device needed; used by the CPU test-suite):
 * bnn_layer_order = bnn_plan_layer_order without a plan;
 * bnn_fragment_table: which = 2: regress_nn gather table, entry [f*64 + lane] = index into the flat parameter vector (or
 * d = 7583 for "zero") that lane `lane` loads into v_mfma_f32_16x16x4 operand register f; which = 1: the feature_nn weight
 * registers of the v_mfma_f32_4x4x1 path (CBSZ broadcast), entry [R*64 + lane] = index of the parameter lane `lane` holds in weight
 * register R (layout: bnn_layout.h, WR<KIN>).
 * Both return the number of entries (the required capacity) or a negative bnn_status. */
int bnn_layer_order(const bnn_arch* arch, int layer, int noisy, int32_t* host_order, int cap);
int bnn_fragment_table(const bnn_arch* arch, int noisy, int which, int16_t* host_table, int cap);

/* Everything one MultiSWAG evaluation needs besides the ensemble and the noise. */
typedef struct bnn_grid {
    int64_t B;       /* systems (rows of x)                                                        */
    int32_t T;       /* timesteps per system (100); any T in [2, 16384]                            */
    int32_t J;       /* weight draws                                                               */
    int32_t nchunks; /* draw e covers chunk e % nchunks of the systems (torch.chunk semantics,
                        chunk size ceil(B/nchunks)) and writes output row e / nchunks.
                        1 = every draw covers all systems (dense systems x draws grid).
                        figures/multiswag_5_planet.py:295-298 is nchunks = 10.                     */
    int32_t systems_per_block; /* 0 = choose; else a multiple of 64.  With 0 the pretrained network's quiet forward takes a
                        TILE-SPLIT launch form on small grids (at most 256 blocks of 16 systems, or draws that cover at most 16
                        systems each: the evaluation scripts' per-chunk calls, figures/multiswag_5_planet.py:295-298): 16 systems per
                        workgroup, its four waves sharing a batch's tiles -- bit-identical outputs; an explicit value keeps the plain form */
    int32_t noisy;   /* bnn_forward_f32 only: 1 = forward(noisy_val=True) with ALL noise generated in-kernel
                        (eps, eps_in, eps_sum all NULL); explicit eps_in/eps_sum imply noisy regardless          */
    int32_t engine;  /* 0 = choose (the pretrained network at T % 4 == 0, T >= 8: its register-resident kernels; else the generic engine,
                            in the network's specialised form when one is attached to the plan);
                        1 = the generic engine's ahead-of-time form whatever the shape (cross-checks, measurements);
                        2 = the specialised form (error when none is attached)                                   */
    int64_t chunk_B;   /* nchunks > 1 with the batch sharded over devices: the chunks partition the WHOLE batch of chunk_B systems  */
    int64_t chunk_off; /* (torch.chunk semantics, chunk size ceil(chunk_B / nchunks)), of which this call holds rows
                          [chunk_off, chunk_off + B); a draw covers the part of its chunk that lies in the shard.  0, 0 = the call
                          holds the whole batch.  Results are then bit-identical to the unsharded call with system_id0 = chunk_off. */
    const void* nonfinite; /* DEVICE record written by bnn_nonfinite_scan_f32 for THIS call's x, or NULL = x is assumed finite.
                          With a record every forward entry point returns, for the systems it lists, what the reference returns
                          (spock_reg_model.py:452-478 x - mask, :301-321 nn.ReLU): see bnn_nonfinite_scan_f32. */
} bnn_grid;

/* Non-finite inputs.  The reference masks by subtraction (`x = x - mask`, spock_reg_model.py:452-478), so NaN / +-inf in a MASKED column
 * becomes NaN -- not 0 --, nn.Linear carries it into every neuron and nn.ReLU (:301-321) propagates NaN: the system's (mu, std) is NaN.
 * In a live column NaN does the same; +-inf becomes +-inf x weight, the ReLU keeps +inf and turns -inf into 0, and the system's outputs
 * are NaN unless the infinity only ever meets weights of one sign.  The forward kernels are written for finite data (they never read the
 * columns the pretrained mask drops; their ReLU is an integer max on the bit pattern).  bnn_nonfinite_scan_f32 reads x ONCE (not once per
 * draw) and writes the list of systems that hold a non-finite value into `record` (int32 [4 + B]: [0] = listed systems, [1] = of which
 * certainly NaN whatever the weights -- a NaN anywhere or +-inf in a masked column --, [4 + i] = (system << 1) | certain); a forward entry
 * point whose grid.nonfinite points at the record re-evaluates the listed systems with plain IEEE arithmetic after its kernel (x - x on
 * the masked columns, NaN-propagating ReLU, natural accumulation order) and overwrites their outputs -- (mu, std), pre_clamp, summary,
 * latents, or the statistics tail's value.  Enqueue only; the record may be reused for any number of calls on the same x (B, T and the
 * plan's mask must match).  Finite inputs whose intermediate values overflow fp32 are outside this mechanism: they follow IEEE
 * arithmetic in the kernels (the reference returns NaN for those as well). */
size_t bnn_nonfinite_record_bytes(int64_t B);
int bnn_nonfinite_scan_f32(const bnn_plan* plan, const float* x, int64_t B, int32_t T, void* record, void* stream);

/* SWAGModel.sample_weights (spock_reg_model.py:815-838), J draws at once.
 *   w_avg, w2_avg [S,d]; pre_D [S,d,K]; seed_idx [J] int32 (which ensemble member each draw uses:
 *   figures/spock/regression.py:78); z1 [J,d], z2 [J,K] = the reference's randn((1,d)), randn((K,1))
 *   per draw, or both NULL to generate them in-kernel from Philox4x32-10 keyed by philox_seed
 *   (counter = draw_id0 + e, element) -- see bnn_philox_normal_f32.  W_out [J,d]. */
int bnn_swag_draw_f32(const bnn_plan* plan, const float* w_avg, const float* w2_avg, const float* pre_D,
                      int32_t S, int32_t K, const int32_t* seed_idx, int32_t J, const float* z1,
                      const float* z2, float scale, uint64_t philox_seed, int64_t draw_id0, float* W_out,
                      void* stream);

/* VarModel.forward (spock_reg_model.py:486-528) for J already-materialised weight vectors.
 *   x [B,T,n_features] fp32 contiguous; W [J,d]; out [J/nchunks, B, 2] = cat(mu, std).
 *   eps [J/nchunks, B, 2, latent]: the two randn_like draws of compute_summary_stats (:426-427), or NULL
 *   for in-kernel Philox.  eps_in [J/nchunks,B,T,n_features] (:445) and eps_sum [J/nchunks,B,2 latent (+2)] (:449):
 *   both non-NULL = forward(noisy_val=True); both NULL = noisy_val=False / forward_swag_fast.
 *   Optional debug outputs (may be NULL): pre_clamp [J/nchunks,B,2] = regress_nn output,
 *   summary [J/nchunks,B,40] = compute_summary_stats output. */
int bnn_forward_f32(const bnn_plan* plan, const bnn_grid* grid, const float* x, const float* W, const float* eps,
                    const float* eps_in, const float* eps_sum, uint64_t philox_seed, int64_t draw_id0,
                    int64_t system_id0, float* out, float* pre_clamp, float* summary, void* stream);

/* feature_nn alone: the per-timestep latents that compute_summary_stats leaves in self.latents (spock_reg_model.py:417, 433), for
 * inspection (the out-of-scope figures/feature_importance.py reads them).  Same arguments as bnn_forward_f32 (masks, optional input
 * noise: eps_in explicit, or grid.noisy with in-kernel Philox); latents [J/nchunks, B, T, latent].  Runs on the generic engine for every
 * network, so its numbers are the natural-order ones (bit-identical to the oracle's `latents`). */
int bnn_feature_nn_f32(const bnn_plan* plan, const bnn_grid* grid, const float* x, const float* W, const float* eps_in,
                       uint64_t philox_seed, int64_t draw_id0, int64_t system_id0, float* latents, void* stream);

/* OPT-IN reduced-precision forward for the precision sweep of BASELINE.json configs[4] (the reference's 5-planet script casts to
 * fp32 at figures/multiswag_5_planet.py:287; a bf16 cast there is what this measures).  feature_nn runs on the bf16 matrix pipe
 * with fp32 accumulation; pool, sampled moments and regress_nn stay exact fp32; same arguments and Philox streams as
 * bnn_forward_f32, so outputs are comparable element by element.  v50 column mask only, no noisy form.
 *   BNN_PREC_BF16    x, weights, activations rounded to bf16, 1 product per layer    (|d mu| up to ~1e-1: NOT within the 1e-5 bar)
 *   BNN_PREC_BF16X3  operands split into hi + lo bf16 (16 significant bits), 3 products
 *   BNN_PREC_BF16X6  three bf16 parts (24 bits), the 6 products of order <= 2: fp32-level error, not bit-reproducible vs fp32
 *   BNN_PREC_F16     IEEE half operands (11 significant bits), 1 product; operands must stay below 65 504 in magnitude
 *                    (larger values saturate: finite but wrong; e.g. the scripts' constant-4 fill of unstable systems)
 *   BNN_PREC_F16X3   half operands split hi + lo (22 significant bits), 3 products: close to fp32 at the cost of bf16x3 */
enum bnn_precision { BNN_PREC_F32 = 0, BNN_PREC_BF16 = 1, BNN_PREC_BF16X3 = 2, BNN_PREC_BF16X6 = 3, BNN_PREC_F16 = 4, BNN_PREC_F16X3 = 5 };
int bnn_forward_lowp_f32(const bnn_plan* plan, const bnn_grid* grid, const float* x, const float* W, const float* eps,
                         uint64_t philox_seed, int64_t draw_id0, int64_t system_id0, int32_t precision, float* out,
                         float* pre_clamp, float* summary, void* stream);

/* Fused SWAGModel.forward_swag_fast (spock_reg_model.py:878-908) over the MC driver loop
 * (figures/spock/regression.py:74-92 inside figures/multiswag_5_planet.py:295-298 /
 * figures/main_figures.py:154-156): per draw, sample the weights in the kernel prologue (same
 * arithmetic as bnn_swag_draw_f32, bit for bit), keep them on chip, stream x once, write (mu, std).
 * system_id0 = global index of x row 0 (Philox counters use global ids so results do not depend on
 * how systems are sharded over GPUs).
 * W_workspace: NULL = every workgroup samples its draw in its own prologue (one launch, no scratch memory); only the pretrained
 * network's kernels have this form (T % 4 == 0, K <= 32): other shapes return BNN_ERR_UNSUPPORTED without a workspace.
 * Non-NULL [J,d] scratch = each draw is sampled ONCE by a draw kernel into the workspace and the forward
 * kernel of the same call picks it up (two launches on `stream`, identical results bit for bit; faster
 * whenever a draw is shared by many workgroups, i.e. B/nchunks above a few hundred systems). */
int bnn_multiswag_f32(const bnn_plan* plan, const bnn_grid* grid, const float* x, const float* w_avg,
                      const float* w2_avg, const float* pre_D, int32_t S, int32_t K, const int32_t* seed_idx,
                      const float* z1, const float* z2, const float* eps, float scale, uint64_t philox_seed,
                      int64_t draw_id0, int64_t system_id0, float* W_workspace, float* out, float* pre_clamp,
                      float* summary, void* stream);

/* Pre-path feature packing + standardisation:
 * data_setup_kernel (figures/spock/regression.py:183-213): tseries [N,T,26] + mass [N,3] (float64) -> 32 raw columns
 * (26 series + 3 masses + isnotfinite flags of columns 3, 6, 7), nan_to_num(nan, +-inf -> 0), the nine angle columns
 * {11,12,13,17,18,19,23,24,25} expanded to (cos, sin) -> X [N,T,41] float64;
 * then StandardScaler.transform in float64 and .float() (figures/multiswag_5_planet.py:280-287, regression.py:144-145).
 *   X64_out [N,T,41] float64 (may be NULL) = data_setup_kernel's return value;
 *   x32_out [N,T,41] float32 (may be NULL; needs mean/scale [41] float64) = the tensor the network consumes.
 * tseries == NULL: X64_in [N,T,41] is taken as already packed and only standardised (mass ignored). */
int bnn_feature_pack_f64(const double* tseries, const double* mass, const double* X64_in, int64_t N, int32_t T,
                         const double* mean, const double* scale, double* X64_out, float* x32_out, void* stream);

/* Predictive moments over draws: samples [R,B,2] -> moments [B,4] (float64):
 * sum mu, sum mu^2, sum std, sum std^2 over r (float64; a fixed 16-way partition of r, so deterministic).  accumulate != 0 adds to
 * the existing contents (for processing draws in slabs). */
int bnn_moments_f64(const float* samples, int64_t R, int64_t B, double* moments, int32_t accumulate, void* stream);

/* Per-system order statistics over the draws (the deterministic part):
 * np.median(_preds[..., 0], 0) / np.median(_preds[..., 1], 0) of figures/main_figures.py:277-278 and the
 * np.percentile calls of figures/multiswag_5_planet.py:484-489, numpy's default linear interpolation.
 *   samples [R,B,2] float32; q [nq] percentiles in [0,100] (host doubles, nq <= 16); out [B,2,nq] float32.
 * R <= 16384 (one workgroup sorts one system's column in LDS). */
int bnn_quantiles_f32(const float* samples, int64_t R, int64_t B, const double* host_q, int32_t nq, float* out, void* stream);

/* fast_truncnorm (figures/multiswag_5_planet.py:306-370, figures/main_figures.py:167-227); the scripts call it with left = 4,
 * right = inf; the other two forms of its acceptance test (:352-358: left = inf -> v < right; both finite -> left < v < right)
 * are built as well (pass INFINITY for an open side):
 *   musd [n,2] float32 = (loc, scale) pairs -- the [R,B,2] output of the forward as it stands;
 *   normals [nsamp, n] float64 (the reference's np.random.normal draws, element order) or NULL = in-kernel Philox;
 *   out [n] float32 = first of the nsamp candidates loc + scale*z (float64) inside the interval, else the first one. */
int bnn_truncnorm_f32(const float* musd, int64_t n, const double* normals, int32_t nsamp, double left, double right, uint64_t philox_seed,
                      int64_t id0, float* out, void* stream);

/* Prior resampling (figures/multiswag_5_planet.py:396-422): vals[i] >= threshold is replaced by inv_cdf(u[rank[i]]).
 *   rank [n] int64 = number of earlier elements >= threshold (exclusive prefix count, C order);
 *   cum, edge [m] float64 = the interp1d table, SORTED by cum (the host builds it: its size depends on the count);
 *   u [count] float64 (np.random.rand) or NULL = in-kernel Philox uniforms.  vals is updated in place. */
int bnn_prior_resample_f32(float* vals, int64_t n, const int64_t* rank, const double* cum, const double* edge, int64_t m,
                           const double* u, double threshold, uint64_t philox_seed, int64_t id0, void* stream);

/* predict_instability (spock_reg_model.py:437-442) on an explicit summary: regress_nn + soft_clamp.
 *   summary [J,B,40], W [J,d] (flat parameter vectors; only regress_nn.* is read) -> out [J,B,2] = (mu, std),
 *   pre_clamp [J,B,2] or NULL.  Same accumulation order as the tail of bnn_forward_f32 (bit-identical on its summary). */
int bnn_regress_f32(const bnn_plan* plan, const float* summary, const float* W, int64_t J, int64_t B, float* out, float* pre_clamp,
                    void* stream);

/* np.min over the last axis of size `group` (min over trios, figures/multiswag_5_planet.py:428): vals [n,group] -> out [n]. */
int bnn_group_min_f32(const float* vals, int64_t n, int32_t group, float* out, void* stream);

/* The normals the kernels generate when a noise pointer is NULL, written out for inspection:
 *   kind 0: z1 [n_draws, d]   kind 1: z2 [n_draws, K]   kind 2: eps [rows, B, 2, latent = width (0: 20)]
 *   kind 3: eps_in [rows, B, T = width, n_features (0: 41)] (six normals per Philox block; Philox4x32 with SEVEN rounds for this one
 *           stream, ten everywhere else)   kind 4: eps_sum [rows, B, width (0: 40)]
 *   kind 5: candidates of the statistics epilogue's truncated-normal draw [rows, B, nsamp = width]
 *   kind 6: survival level of its prior draw [rows, B], uniform on (0, 1]
 *   id0 = draw_id0 (kinds 0,1) or output-row id0 (kinds 2-6); system_id0 only for kinds 2-6. */
int bnn_philox_normal_f32(int32_t kind, uint64_t philox_seed, int64_t id0, int64_t n_rows, int64_t B,
                          int64_t system_id0, int32_t width, int32_t n_features, float* out, void* stream);

/* ---- streaming statistics epilogue: what the evaluation scripts consume, without [J,B,2] in memory -------
 * Per evaluation: (mu, std) -> fast_truncnorm(left, nsamp) (figures/multiswag_5_planet.py:306-370, 388-392; main_figures.py:167-227)
 * -> values >= prior_thr redrawn from the prior (:396-422), all noise Philox keyed by (global output row, global system), so the
 * numbers do not depend on sharding, draw slabs or launch mode.  The prior is inverted from its exact survival function on m
 * equally spaced knots (bnn_prior_table_f32; the reference inverts a Riemann-sum table whose size depends on the data). */
typedef struct bnn_stats {
    int32_t tn_nsamp;        /* candidates per truncated-normal draw: 40 in the scripts                  */
    float tn_left;           /* left truncation point: 4                                                 */
    float prior_thr;         /* values >= this are redrawn from the prior: 9 (INFINITY: never)           */
    int32_t prior_m;         /* knots of the survival table                                              */
    float prior_step;        /* knot spacing, from bnn_prior_table_f32                                   */
    int32_t reserved;
    const float* prior_surv; /* DEVICE [prior_m] fp32 survival function at prior_thr + i * prior_step    */
} bnn_stats;

/* Host: S(t_i) = P(T > t_i | T >= thr) of the scripts' prior (:400-404) at t_i = thr + i (top - thr)/(m - 1), float64 closed form
 * (exp, erfc) rounded to fp32; *host_step = the knot spacing.  Copy host_surv to the device for bnn_stats.prior_surv. */
int bnn_prior_table_f32(double thr, double top, int32_t m, float* host_surv, double* host_step);

/* The epilogue on materialised pairs: musd [R,B,2] -> out [R,B] (log10 instability time per evaluation). */
int bnn_stats_draw_f32(const float* musd, int64_t R, int64_t B, const bnn_stats* st, uint64_t philox_seed, int64_t row_id0,
                       int64_t system_id0, float* out, void* stream);

/* bnn_multiswag_f32 (draw-once workspace form; W_workspace [J,d] required) with the epilogue fused into the regress_nn tail:
 * t_out [J/nchunks, B] float32 and nothing else leaves the chip; bit-identical to bnn_multiswag_f32 + bnn_stats_draw_f32. */
int bnn_multiswag_stats_f32(const bnn_plan* plan, const bnn_grid* grid, const float* x, const float* w_avg, const float* w2_avg,
                            const float* pre_D, int32_t S, int32_t K, const int32_t* seed_idx, const float* z1, const float* z2,
                            const float* eps, float scale, uint64_t philox_seed, int64_t draw_id0, int64_t system_id0,
                            float* W_workspace, const bnn_stats* st, float* t_out, void* stream);

/* Streaming per-simulation quantile sketch: min over `group` consecutive systems (min over trios, :428), then a histogram over
 * 1..4 contiguous uniform segments (bin 0 = below lo[0], reported as lo[0] -- the one place where the error is not bounded by a
 * bin width; values at or above the last hi fall in the last value bin; the very last bin counts NaN draws) and float64
 * sum / sum of squares.  hist [nbins, n_sims] uint32 (bin-major), mom [n_sims, 2] float64, both zeroed by the caller and
 * accumulated over any number of slabs t [R,B].  Memory is O(n_sims * nbins) whatever the number of draws.
 * bnn_sketch_quantiles_f32: numpy 'linear' percentiles (np.median / np.percentile of :484-489, main_figures.py:277-278)
 * read off the histogram; every estimate lies within one bin width of the exact order statistic (draws below lo[0] excepted);
 * a simulation with any NaN draw gets NaN percentiles, like np.percentile.  out [n_sims, nq]. */
typedef struct bnn_sketch {
    int32_t nseg;
    int32_t reserved;
    float lo[4], hi[4];
    int32_t n[4];
} bnn_sketch;
int bnn_sketch_bins(const bnn_sketch* sk); /* total bins (1 + sum n + 1: underflow, values, NaN counter), or negative */
int bnn_sketch_update_u32(const float* t, int64_t R, int64_t B, int32_t group, const bnn_sketch* sk, uint32_t* hist, double* mom,
                          void* stream);
int bnn_sketch_quantiles_f32(const uint32_t* hist, int64_t n_sims, const bnn_sketch* sk, const double* host_q, int32_t nq,
                             float* out, void* stream);

/* Slab drivers (host loops over the kernels above; enqueue only, no synchronisation): the (systems x draws) grid of `grid`
 * evaluated `draws_per_launch` draws at a time (a positive multiple of nchunks) through caller-provided scratch
 *   W_workspace [draws_per_launch, d];  out_workspace [draws_per_launch/nchunks, B, 2]  resp.  t_workspace [draws_per_launch/nchunks, B]
 * and reduced on the fly, so nothing of size J x B exists: the multi-GPU payloads of SURVEY.md section 8e.
 *   bnn_multiswag_moments_f64: moments [B,4] float64 (overwritten) = sum mu, sum mu^2, sum std, sum std^2 over the output rows;
 *   bnn_multiswag_bands_f32:   the quantile sketch (hist, mom; accumulated, zeroed by the caller) of the post-epilogue times
 *                              (statistics fused in the forward tail, min over `group` consecutive systems); read it with
 *                              bnn_sketch_quantiles_f32.  Philox noise only (the replay forms need explicit noise per call). */
int bnn_multiswag_moments_f64(const bnn_plan* plan, const bnn_grid* grid, const float* x, const float* w_avg, const float* w2_avg,
                              const float* pre_D, int32_t S, int32_t K, const int32_t* seed_idx, float scale, uint64_t philox_seed,
                              int64_t draw_id0, int64_t system_id0, int32_t draws_per_launch, float* W_workspace,
                              float* out_workspace, double* moments, void* stream);
int bnn_multiswag_bands_f32(const bnn_plan* plan, const bnn_grid* grid, const float* x, const float* w_avg, const float* w2_avg,
                            const float* pre_D, int32_t S, int32_t K, const int32_t* seed_idx, float scale, uint64_t philox_seed,
                            int64_t draw_id0, int64_t system_id0, int32_t draws_per_launch, float* W_workspace, float* t_workspace,
                            const bnn_stats* st, int32_t group, const bnn_sketch* sk, uint32_t* hist, double* mom, void* stream);

/* Raw Philox4x32-10 blocks for known-answer tests: out[n][4] = philox(ctr = {c0+i, c1, c2, c3}, key). */
int bnn_philox_raw_u32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, int64_t n,
                       uint32_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BNN_CHAOS_HIP_H */
