/*
 * bnn_oracle.c -- CPU restatement of the MultiSWAG inference hot path of
 * MilesCranmer/bnn_chaos_model (reference file spock_reg_model.py).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (bnn_chaos_model_amd/)
 * never calls into this file and has no CPU fallback.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function below against
 * fixtures produced by the unmodified reference (tests/golden/make_golden.py, run in the
 * build container where /root/reference is importable).
 *
 * The file is compiled twice: -DREAL=float -DPFX=orc32_ (the fp32 restatement the HIP path
 * is diffed against) and -DREAL=double -DPFX=orc64_ (error-budget "truth").
 * Build flags must include -ffp-contract=off: every fused multiply-add below is an explicit
 * FMA() call, every unfused product/sum is meant to round separately, exactly as the
 * reference's eager torch ops do.
 *
 * Flat parameter vector layout (reference state_dict order, spock_reg_model.py:734-761;
 * own parameters precede submodules), F=n_features, H=hidden, L=latent:
 *   input_noise_logvar[F] | summary_noise_logvar[S] |
 *   feature_nn.0.{weight[H,F],bias[H]} | feature_nn.2.{weight[H,H],bias[H]} | feature_nn.4.{weight[L,H],bias[L]} |
 *   regress_nn.0.{weight[H,S],bias[H]} | regress_nn.2.{weight[H,H],bias[H]} | regress_nn.4.{weight[2,H],bias[2]}
 * with S = 2L summary entries, or 2L + 2 with fix_megno (:360-362).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL float
#define PFX orc32_
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(PFX, name)

#define IS_F32 (sizeof(REAL) == 4)
static inline REAL FMA(REAL a, REAL b, REAL c) { return IS_F32 ? (REAL)fmaf((float)a, (float)b, (float)c) : (REAL)fma(a, b, c); }
static inline REAL SQRT(REAL a) { return IS_F32 ? (REAL)sqrtf((float)a) : (REAL)sqrt(a); }
static inline REAL FABS(REAL a) { return IS_F32 ? (REAL)fabsf((float)a) : (REAL)fabs(a); }
static inline REAL TANH(REAL a) { return IS_F32 ? (REAL)tanhf((float)a) : (REAL)tanh(a); }
static inline REAL EXP(REAL a) { return IS_F32 ? (REAL)expf((float)a) : (REAL)exp(a); }

typedef struct {
    int32_t n_features; /* 41  (hparams['time_series_features'])        */
    int32_t hidden;     /* 40  (hparams['hidden'])                       */
    int32_t latent;     /* 20  (hparams['latent'])                       */
    int32_t T;          /* timesteps per system (100)                    */
    uint64_t zero_mask; /* bit f set => column f zeroed (spock_reg_model.py:452-478, applied :884-897) */
    double lowest;      /* soft_clamp floor for std: 0.5, or 0.1 with lower_std (:363-365)            */
    int32_t fix_megno;  /* hparams['fix_megno'] (:360-362): the summary gains [mean_t, std_t] of the RAW MEGNO column (7),
                           regress_nn.0 and summary_noise_logvar are 2 wider (:488-491, :509-510, summarize_megno :480-484) */
    int32_t reserved;
} orc_arch;
#define MEGNO_COL 7 /* self.megno_location (:371) */

/* Optional accumulation schedule.  The reference's summation order inside nn.Linear /
 * torch.std is whatever MKL and ATen do; the algorithm does not define one.  With sched==NULL the
 * oracle uses the natural order (bias first, k ascending; one Welford pass over t).  A schedule pins
 * a different -- equally valid -- order so that an implementation using that order can be compared
 * bit-for-bit instead of within a tolerance.
 *   order[l][i]: i-th term accumulated into every output of Linear layer l (0..5): an input index,
 *                or -1 for the bias term.  If no -1 is listed the accumulator STARTS at the bias.
 *   pool_parts:  1, or 4 = Welford over the four strided partitions t = p + 4i, merged pairwise
 *                (p^1 then p^2) with the equal-count form of Chan's update.                        */
typedef struct {
    const int32_t* order[6];
    int32_t order_len[6];
    int32_t pool_parts;
} orc_schedule;

int FN(param_count)(const orc_arch* a) {
    int F = a->n_features, H = a->hidden, L = a->latent, S = 2 * L + (a->fix_megno ? 2 : 0);
    return F + S + (H * F + H) + (H * H + H) + (L * H + L) + (H * S + H) + (H * H + H) + (2 * H + 2);
}

/* ---- SWAG weight draw: SWAGModel.sample_weights (spock_reg_model.py:815-838) ------------------
 *   D     = pre_D - w_avg[:,None]                                   (:826)
 *   sigma = abs(diag(w2_avg - w_avg**2))                            (:832)  -> elementwise, only the diagonal matters
 *   w     = w_avg + scale*(1/sqrt 2) * z1 @ sigma**0.5              (:834)
 *   w    += scale * (D @ z2).T / sqrt(2(K-1))                       (:835)
 * Scalars are float64 on the host and enter the tensor ops rounded to the tensor dtype. */
int FN(swag_draw)(const REAL* w_avg, const REAL* w2_avg, const REAL* pre_D, int d, int K,
                  const REAL* z1, const REAL* z2, double scale, REAL* w_out) {
    if (!w_avg || !w2_avg || !pre_D || !z1 || !z2 || !w_out || d <= 0 || K < 2) return -1;
    const REAL c1 = (REAL)(scale * (1.0 / sqrt(2.0)));
    const REAL c2 = (REAL)sqrt(2.0 * (K - 1));
    const REAL sc = (REAL)scale;
    for (int i = 0; i < d; ++i) {
        REAL sq = w_avg[i] * w_avg[i];
        REAL var = w2_avg[i] - sq;
        REAL sd = SQRT(FABS(var)); /* abs matters: some seeds have a negative element */
        REAL t1 = (c1 * z1[i]) * sd;
        REAL w = w_avg[i] + t1;
        REAL dot = 0;
        for (int k = 0; k < K; ++k) {
            REAL D = pre_D[(size_t)i * K + k] - w_avg[i];
            dot = FMA(D, z2[k], dot);
        }
        REAL t2 = (sc * dot) / c2;
        w_out[i] = w + t2;
    }
    return 0;
}

/* One Linear layer for one input row: y[j] = b[j] + sum_k W[j,k] x[k]  (nn.Linear, :301-321) */
static void linear_row(const REAL* W, const REAL* b, int n_out, int n_in, const REAL* x, REAL* y, int relu,
                       const int32_t* order, int order_len) {
    for (int j = 0; j < n_out; ++j) {
        const REAL* wr = W + (size_t)j * n_in;
        REAL acc;
        if (!order) {
            acc = b[j];
            for (int k = 0; k < n_in; ++k) acc = FMA(wr[k], x[k], acc);
        } else {
            int has_bias = 0;
            for (int i = 0; i < order_len; ++i) has_bias |= (order[i] < 0);
            acc = has_bias ? (REAL)0 : b[j];
            for (int i = 0; i < order_len; ++i) {
                int k = order[i];
                acc = (k < 0) ? FMA(b[j], (REAL)1, acc) : FMA(wr[k], x[k], acc);
            }
        }
        y[j] = (relu && !(acc > 0)) ? (REAL)0 : acc; /* nn.ReLU */
    }
}

/* soft_clamp (spock_reg_model.py:295-296): 0.5*(tanh(x)+1)*(high-lo) + lo */
static inline REAL soft_clamp(REAL x, double lo, double hi) {
    REAL t = TANH(x) + (REAL)1;
    REAL h = (REAL)0.5 * t;
    REAL s = h * (REAL)(hi - lo);
    return s + (REAL)lo;
}

/* ---- forward: VarModel.forward (:486-528) / the post-draw half of forward_swag_fast (:884-907) --
 *   x[B,T,F]; w = flat parameter vector; eps1, eps2 [B,L] = the two randn_like draws of
 *   compute_summary_stats (:426-427, ALWAYS consumed); eps_in [B,T,F] (:445) and eps_sum [B,2L] (:449)
 *   are consumed only by forward(noisy_val=True) -- pass NULL for forward_swag_fast / noisy_val=False.
 *   With fix_megno the summary is S = 2L + 2 wide (eps_sum [B,S], summary [B,S]).
 *   Optional outputs: pre_clamp[B,2] (regress_nn output), summary[B,S] (before summary noise),
 *   latents[B,T,L] (feature_nn output). */
int FN(forward)(const orc_arch* a, const REAL* x, int64_t B, const REAL* w, const REAL* eps_in, const REAL* eps1,
                const REAL* eps2, const REAL* eps_sum, const orc_schedule* sched, REAL* out, REAL* pre_clamp,
                REAL* summary, REAL* latents) {
    if (!a || !x || !w || !eps1 || !eps2 || !out || B < 0) return -1;
    const int F = a->n_features, H = a->hidden, L = a->latent, T = a->T;
    if (F <= 0 || F > 64 || H <= 0 || L <= 0 || T < 2) return -2;
    const int P = sched ? sched->pool_parts : 1;
    if (P != 1 && P != 4) return -3;
    if (P == 4 && T % 4) return -3;
    const int FM = a->fix_megno ? 1 : 0, S = 2 * L + 2 * FM;
    if (FM && F <= MEGNO_COL) return -2;
    const REAL* in_logvar = w;
    const REAL* sum_logvar = w + F;
    const REAL* W1 = sum_logvar + S;
    const REAL* b1 = W1 + H * F;
    const REAL* W2 = b1 + H;
    const REAL* b2 = W2 + H * H;
    const REAL* W3 = b2 + H;
    const REAL* b3 = W3 + L * H;
    const REAL* W4 = b3 + L;
    const REAL* b4 = W4 + H * S;
    const REAL* W5 = b4 + H;
    const REAL* b5 = W5 + H * H;
    const REAL* W6 = b5 + H;
    const REAL* b6 = W6 + 2 * H;
    const int32_t* ord[6] = {0, 0, 0, 0, 0, 0};
    int ordn[6] = {0, 0, 0, 0, 0, 0};
    if (sched)
        for (int l = 0; l < 6; ++l) { ord[l] = sched->order[l]; ordn[l] = sched->order_len[l]; }

    REAL in_scale[64];
    if (eps_in)
        for (int f = 0; f < F; ++f) in_scale[f] = EXP(in_logvar[f] / (REAL)2); /* exp(logvar/2), :445 */

    int rc = 0;
#pragma omp parallel
    {
        REAL* xr = (REAL*)malloc(sizeof(REAL) * (size_t)(F + 3 * H + 2 * L + 2 + 2 * L * 4 + 8));
        REAL* h1 = xr + F;
        REAL* h2 = h1 + H;
        REAL* y = h2 + H;        /* [L] */
        REAL* s = y + L;         /* [S] summary */
        REAL* mean = s + 2 * L + 2; /* [4][L] partition means */
        REAL* m2 = mean + 4 * L; /* [4][L] partition M2    */
        REAL gmean[4], gm2[4];   /* the same pool over the raw MEGNO column (fix_megno) */
#pragma omp for schedule(static)
        for (int64_t b = 0; b < B; ++b) {
            for (int i = 0; i < 4 * L; ++i) { mean[i] = 0; m2[i] = 0; }
            for (int i = 0; i < 4; ++i) { gmean[i] = 0; gm2[i] = 0; }
            for (int t = 0; t < T; ++t) {
                const REAL* xi = x + ((size_t)b * T + t) * F;
                if (FM) { /* summarize_megno (:480-484) sees x BEFORE zero_megno and before any noise (:488-491) */
                    int p = (P == 4) ? (t & 3) : 0;
                    int cnt = (P == 4) ? (t >> 2) + 1 : t + 1;
                    REAL rc_n = (REAL)1 / (REAL)cnt;
                    REAL delta = xi[MEGNO_COL] - gmean[p];
                    REAL mnew = FMA(delta, rc_n, gmean[p]);
                    gm2[p] = FMA(delta, xi[MEGNO_COL] - mnew, gm2[p]);
                    gmean[p] = mnew;
                }
                for (int f = 0; f < F; ++f) {
                    /* zero_megno/mmr/nan/eplusminus: x - mask == 0 on masked columns (:452-478) */
                    REAL v = ((a->zero_mask >> f) & 1) ? (REAL)0 : xi[f];
                    if (eps_in) v = v + eps_in[((size_t)b * T + t) * F + f] * in_scale[f]; /* add_input_noise :444-446 */
                    xr[f] = v;
                }
                linear_row(W1, b1, H, F, xr, h1, 1, ord[0], ordn[0]); /* feature_nn (:359, :417) */
                linear_row(W2, b2, H, H, h1, h2, 1, ord[1], ordn[1]);
                linear_row(W3, b3, L, H, h2, y, 0, ord[2], ordn[2]);
                if (latents) memcpy(latents + ((size_t)b * T + t) * L, y, sizeof(REAL) * L);
                /* mean / unbiased variance over t (torch.mean, torch.std: :418-419): Welford with fused updates */
                int p = (P == 4) ? (t & 3) : 0;
                int cnt = (P == 4) ? (t >> 2) + 1 : t + 1;
                REAL rc_n = (REAL)1 / (REAL)cnt;
                for (int n = 0; n < L; ++n) {
                    REAL delta = y[n] - mean[p * L + n];
                    REAL mnew = FMA(delta, rc_n, mean[p * L + n]);
                    m2[p * L + n] = FMA(delta, y[n] - mnew, m2[p * L + n]);
                    mean[p * L + n] = mnew;
                }
            }
            if (P == 4) { /* merge partitions: (0,1),(2,3) then the two halves; equal counts */
                REAL half_n = (REAL)(T / 4) * (REAL)0.5;
                for (int stage = 0; stage < 2; ++stage) {
                    int pa = 0, pb = stage == 0 ? 1 : 2;
                    for (int rep = 0; rep < (stage == 0 ? 2 : 1); ++rep, pa += 2, pb += 2)
                        for (int n = 0; n < L; ++n) {
                            REAL dl = mean[pb * L + n] - mean[pa * L + n];
                            REAL mm = (mean[pa * L + n] + mean[pb * L + n]) * (REAL)0.5;
                            REAL q = (m2[pa * L + n] + m2[pb * L + n]) + (dl * dl) * half_n;
                            mean[pa * L + n] = mm;
                            m2[pa * L + n] = q;
                        }
                    half_n = half_n * (REAL)2;
                }
                if (FM) {
                    REAL hn = (REAL)(T / 4) * (REAL)0.5;
                    for (int stage = 0; stage < 2; ++stage) {
                        int pa = 0, pb = stage == 0 ? 1 : 2;
                        for (int rep = 0; rep < (stage == 0 ? 2 : 1); ++rep, pa += 2, pb += 2) {
                            REAL dl = gmean[pb] - gmean[pa];
                            REAL mm = (gmean[pa] + gmean[pb]) * (REAL)0.5;
                            REAL q = (gm2[pa] + gm2[pb]) + (dl * dl) * hn;
                            gmean[pa] = mm;
                            gm2[pa] = q;
                        }
                        hn = hn * (REAL)2;
                    }
                }
            }
            for (int n = 0; n < L; ++n) { /* compute_summary_stats, :418-431 */
                REAL sample_mu = mean[n];
                REAL sd = SQRT(m2[n] / (REAL)(T - 1)); /* torch.std (unbiased) */
                REAL sample_var = sd * sd;              /* **2                  */
                REAL std_in_mu = SQRT(sample_var / (REAL)T);
                REAL std_in_var = SQRT(((REAL)2 * (sample_var * sample_var)) / (REAL)(T - 1));
                REAL mu_s = eps1[(size_t)b * L + n] * std_in_mu + sample_mu;
                REAL var_s = eps2[(size_t)b * L + n] * std_in_var + sample_var;
                s[n] = mu_s;
                s[L + n] = SQRT(FABS(var_s) + (REAL)1e-5); /* EPSILON, :337 */
            }
            if (FM) { /* torch.cat([summary_stats, megno_avg_std], dim=1) (:509-510): mean, then torch.std (unbiased) */
                s[2 * L] = gmean[0];
                s[2 * L + 1] = SQRT(gm2[0] / (REAL)(T - 1));
            }
            if (summary) memcpy(summary + (size_t)b * S, s, sizeof(REAL) * S);
            if (eps_sum) /* add_summary_noise :448-450 */
                for (int n = 0; n < S; ++n) s[n] = s[n] + eps_sum[(size_t)b * S + n] * EXP(sum_logvar[n] / (REAL)2);
            REAL r[2];
            linear_row(W4, b4, H, S, s, h1, 1, ord[3], ordn[3]); /* regress_nn (:360, :438) */
            linear_row(W5, b5, H, H, h1, h2, 1, ord[4], ordn[4]);
            linear_row(W6, b6, 2, H, h2, r, 0, ord[5], ordn[5]);
            if (pre_clamp) { pre_clamp[b * 2] = r[0]; pre_clamp[b * 2 + 1] = r[1]; }
            out[b * 2 + 0] = soft_clamp(r[0], 4.0, 12.0);      /* :440 */
            out[b * 2 + 1] = soft_clamp(r[1], a->lowest, 6.0); /* :441 */
        }
        free(xr);
    }
    return rc;
}

/* ---- MultiSWAG MC driver: figures/spock/regression.py:74-92 inside the loops of
 *      figures/multiswag_5_planet.py:295-298 / figures/main_figures.py:154-156 -----------------
 * Draw e (0..J-1) uses ensemble member seed_idx[e], covers chunk c = e % nchunks of the systems
 * (torch.chunk semantics: chunk size csz = ceil(B/nchunks)) and writes out[e / nchunks, rows of that chunk].
 * The dense systems x draws grid is nchunks = 1.  z1[J,d], z2[J,K], eps[J/nchunks, B, 2, L]. */
int FN(multiswag)(const orc_arch* a, const REAL* x, int64_t B, const REAL* w_avg, const REAL* w2_avg, const REAL* pre_D,
                  int S, int K, const int32_t* seed_idx, int64_t J, int64_t nchunks, const REAL* z1, const REAL* z2,
                  const REAL* eps, double scale, const orc_schedule* sched, REAL* out) {
    if (!a || !seed_idx || J < 0 || nchunks < 1 || (J % nchunks)) return -1;
    const int d = FN(param_count)(a), L = a->latent, T = a->T, F = a->n_features;
    const int64_t csz = (B + nchunks - 1) / nchunks;
    REAL* w = (REAL*)malloc(sizeof(REAL) * (size_t)d);
    REAL* e1 = (REAL*)malloc(sizeof(REAL) * (size_t)(csz > 0 ? csz : 1) * L * 2);
    REAL* e2 = e1 + (size_t)(csz > 0 ? csz : 1) * L;
    int rc = 0;
    for (int64_t e = 0; e < J && !rc; ++e) {
        int si = seed_idx[e];
        if (si < 0 || si >= S) { rc = -4; break; }
        int64_t c = e % nchunks, r = e / nchunks;
        int64_t b0 = c * csz, b1 = b0 + csz < B ? b0 + csz : B;
        if (b0 >= b1) continue; /* torch.chunk returns fewer chunks when B is small */
        rc = FN(swag_draw)(w_avg + (size_t)si * d, w2_avg + (size_t)si * d, pre_D + (size_t)si * d * K, d, K,
                           z1 + (size_t)e * d, z2 + (size_t)e * K, scale, w);
        if (rc) break;
        for (int64_t b = b0; b < b1; ++b) {
            memcpy(e1 + (size_t)(b - b0) * L, eps + (((size_t)r * B + b) * 2 + 0) * L, sizeof(REAL) * L);
            memcpy(e2 + (size_t)(b - b0) * L, eps + (((size_t)r * B + b) * 2 + 1) * L, sizeof(REAL) * L);
        }
        rc = FN(forward)(a, x + (size_t)b0 * T * F, b1 - b0, w, 0, e1, e2, 0, sched, out + ((size_t)r * B + b0) * 2, 0, 0, 0);
    }
    free(w);
    free(e1);
    return rc;
}
