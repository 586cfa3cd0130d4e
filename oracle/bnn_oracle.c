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
    int32_t depth_in;   /* hparams['in']:  `layers` of feature_nn = mlp(F, L, H, in)  (:301-321, :359)                        */
    int32_t depth_out;  /* hparams['out']: `layers` of regress_nn = mlp(S, 2, H, out) (:360)                                  */
    int32_t reserved;
} orc_arch;
/* mlp(in_n, out_n, hidden, layers) (:301-321): layers == 0 is ONE Linear(in_n, out_n); otherwise Linear(in_n, hidden), ReLU,
 * `layers` x [Linear(hidden, hidden), ReLU], Linear(hidden, out_n): layers + 2 Linear modules, ReLU after all but the last. */
#define ORC_MAX_LIN 18
static int mlp_nlin(int layers) { return layers == 0 ? 1 : layers + 2; }
static int mlp_in(int l, int nlin, int in_n, int hidden) { (void)nlin; return l == 0 ? in_n : hidden; }
static int mlp_out(int l, int nlin, int out_n, int hidden) { return l == nlin - 1 ? out_n : hidden; }
static int mlp_params(int in_n, int out_n, int hidden, int layers) {
    int n = 0, nlin = mlp_nlin(layers);
    for (int l = 0; l < nlin; ++l) n += mlp_out(l, nlin, out_n, hidden) * mlp_in(l, nlin, in_n, hidden) + mlp_out(l, nlin, out_n, hidden);
    return n;
}
#define MEGNO_COL 7 /* self.megno_location (:371) */

/* Optional accumulation schedule.  The reference's summation order inside nn.Linear /
 * torch.std is whatever MKL and ATen do; the algorithm does not define one.  With sched==NULL the
 * oracle uses the natural order (bias first, k ascending; one Welford pass over t).  A schedule pins
 * a different -- equally valid -- order so that an implementation using that order can be compared
 * bit-for-bit instead of within a tolerance.
 *   order[l][i]: i-th term accumulated into every output of Linear layer l (0..5): an input index,
 *                or -1 for the bias term.  If no -1 is listed the accumulator STARTS at the bias.
 *   pool_parts:  1, or 4 = Welford over the four strided partitions t = p + 4i, merged pairwise
 *                ((0,1), (2,3), then the halves) with Chan's update: its symmetric equal-count form where the two
 *                counts agree (always, when T % 4 == 0), the general form otherwise (merge_consts below); any T >= 2.
 *   Explicit orders name the six Linear layers of the depth (in, out) = (1, 1) network; other depths take the natural order. */
typedef struct {
    const int32_t* order[6];
    int32_t order_len[6];
    int32_t pool_parts;
} orc_schedule;

int FN(param_count)(const orc_arch* a) {
    int F = a->n_features, H = a->hidden, L = a->latent, S = 2 * L + (a->fix_megno ? 2 : 0);
    if (a->depth_in < 0 || a->depth_out < 0 || mlp_nlin(a->depth_in) > ORC_MAX_LIN || mlp_nlin(a->depth_out) > ORC_MAX_LIN) return -1;
    return F + S + mlp_params(F, L, H, a->depth_in) + mlp_params(S, 2, H, a->depth_out);
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
        y[j] = (relu && acc <= 0) ? (REAL)0 : acc; /* nn.ReLU: NaN is not <= 0 and stays NaN, -inf becomes 0, +inf stays (torch.relu) */
    }
}

/* soft_clamp (spock_reg_model.py:295-296): 0.5*(tanh(x)+1)*(high-lo) + lo */
static inline REAL soft_clamp(REAL x, double lo, double hi) {
    REAL t = TANH(x) + (REAL)1;
    REAL h = (REAL)0.5 * t;
    REAL s = h * (REAL)(hi - lo);
    return s + (REAL)lo;
}

/* Merge of two Welford partitions a (count na) and b (count nb), Chan et al.  Equal counts use the symmetric form (both inputs
 * enter alike, so a SIMD implementation whose lanes hold either side ends with the same bits); unequal counts the general one, with
 * its two weights formed in float64 and rounded once: wb = nb/n, wab = na*nb/n.  An empty side leaves the other untouched. */
typedef struct { int mode; REAL w1, w2; } pool_merge; /* mode 0: equal (w1 = na/2), 1: general (w1 = wb, w2 = wab), 2: keep a, 3: keep b */
static pool_merge merge_consts(int na, int nb) {
    pool_merge m;
    m.w1 = 0; m.w2 = 0;
    if (nb == 0) m.mode = 2;
    else if (na == 0) m.mode = 3;
    else if (na == nb) { m.mode = 0; m.w1 = (REAL)na * (REAL)0.5; }
    else {
        m.mode = 1;
        m.w1 = (REAL)((double)nb / (double)(na + nb));
        m.w2 = (REAL)((double)na * (double)nb / (double)(na + nb));
    }
    return m;
}
static inline void merge_apply(const pool_merge* m, REAL* ma, REAL* qa, REAL mb, REAL qb) {
    if (m->mode == 2) return;
    if (m->mode == 3) { *ma = mb; *qa = qb; return; }
    REAL dl = mb - *ma;
    if (m->mode == 0) {
        REAL mm = (*ma + mb) * (REAL)0.5;
        *qa = (*qa + qb) + (dl * dl) * m->w1;
        *ma = mm;
    } else {
        REAL mm = FMA(dl, m->w1, *ma);
        *qa = (*qa + qb) + (dl * dl) * m->w2;
        *ma = mm;
    }
}

/* ---- forward: VarModel.forward (:486-528) / the post-draw half of forward_swag_fast (:884-907) --
 *   x[B,T,F]; w = flat parameter vector; eps1, eps2 [B,L] = the two randn_like draws of
 *   compute_summary_stats (:426-427, ALWAYS consumed); eps_in [B,T,F] (:445) and eps_sum [B,2L] (:449)
 *   are consumed only by forward(noisy_val=True) -- pass NULL for forward_swag_fast / noisy_val=False.
 *   With fix_megno the summary is S = 2L + 2 wide (eps_sum [B,S], summary [B,S]).
 *   feature_nn = mlp(F, L, H, depth_in), regress_nn = mlp(S, 2, H, depth_out) (:301-321, :359-360).
 *   Optional outputs: pre_clamp[B,2] (regress_nn output), summary[B,S] (before summary noise),
 *   latents[B,T,L] (feature_nn output). */
int FN(forward)(const orc_arch* a, const REAL* x, int64_t B, const REAL* w, const REAL* eps_in, const REAL* eps1,
                const REAL* eps2, const REAL* eps_sum, const orc_schedule* sched, REAL* out, REAL* pre_clamp,
                REAL* summary, REAL* latents) {
    if (!a || !x || !w || !eps1 || !eps2 || !out || B < 0) return -1;
    const int F = a->n_features, H = a->hidden, L = a->latent, T = a->T;
    if (F <= 0 || F > 128 || H <= 0 || L <= 0 || T < 2) return -2;
    const int P = sched ? sched->pool_parts : 1;
    if (P != 1 && P != 4) return -3;
    const int FM = a->fix_megno ? 1 : 0, S = 2 * L + 2 * FM;
    if (FM && F <= MEGNO_COL) return -2;
    if (FN(param_count)(a) < 0) return -2;
    const int n1 = mlp_nlin(a->depth_in), n2 = mlp_nlin(a->depth_out), NL = n1 + n2;
    const REAL* in_logvar = w;
    const REAL* sum_logvar = w + F;
    /* Linear modules in state_dict order: feature_nn's, then regress_nn's (:734-761) */
    const REAL* Wl[2 * ORC_MAX_LIN];
    const REAL* bl[2 * ORC_MAX_LIN];
    int nin[2 * ORC_MAX_LIN], nout[2 * ORC_MAX_LIN], wmax = F > S ? F : S;
    {
        const REAL* q = sum_logvar + S;
        for (int l = 0; l < NL; ++l) {
            const int feat = l < n1, ll = feat ? l : l - n1, nn = feat ? n1 : n2;
            nin[l] = mlp_in(ll, nn, feat ? F : S, H);
            nout[l] = mlp_out(ll, nn, feat ? L : 2, H);
            Wl[l] = q; q += (size_t)nout[l] * nin[l];
            bl[l] = q; q += nout[l];
            if (nout[l] > wmax) wmax = nout[l];
        }
    }
    const int32_t* ord[2 * ORC_MAX_LIN];
    int ordn[2 * ORC_MAX_LIN];
    for (int l = 0; l < NL; ++l) { ord[l] = 0; ordn[l] = 0; }
    if (sched) {
        int any = 0;
        for (int l = 0; l < 6; ++l) any |= (sched->order[l] != 0);
        if (any && !(a->depth_in == 1 && a->depth_out == 1)) return -3; /* explicit orders name the six layers of the (1, 1) network */
        if (any)
            for (int l = 0; l < 6; ++l) { ord[l] = sched->order[l]; ordn[l] = sched->order_len[l]; }
    }
    /* partition counts and merge constants (P = 4: partition p holds t = p, p + 4, ...) */
    int cnt[4] = {T, 0, 0, 0};
    if (P == 4)
        for (int p = 0; p < 4; ++p) cnt[p] = p < T ? (T - p + 3) / 4 : 0;
    const pool_merge m01 = merge_consts(cnt[0], cnt[1]), m23 = merge_consts(cnt[2], cnt[3]),
                     m0123 = merge_consts(cnt[0] + cnt[1], cnt[2] + cnt[3]);

    REAL in_scale[128];
    if (eps_in)
        for (int f = 0; f < F; ++f) in_scale[f] = EXP(in_logvar[f] / (REAL)2); /* exp(logvar/2), :445 */

    int rc = 0;
#pragma omp parallel
    {
        REAL* buf0 = (REAL*)malloc(sizeof(REAL) * (size_t)(2 * wmax + S + 8 * L + 8));
        REAL* buf1 = buf0 + wmax;
        REAL* s = buf1 + wmax;    /* [S] summary */
        REAL* mean = s + S;       /* [4][L] partition means */
        REAL* m2 = mean + 4 * L;  /* [4][L] partition M2    */
        REAL gmean[4], gm2[4];    /* the same pool over the raw MEGNO column (fix_megno) */
#pragma omp for schedule(static)
        for (int64_t b = 0; b < B; ++b) {
            for (int i = 0; i < 4 * L; ++i) { mean[i] = 0; m2[i] = 0; }
            for (int i = 0; i < 4; ++i) { gmean[i] = 0; gm2[i] = 0; }
            for (int t = 0; t < T; ++t) {
                const REAL* xi = x + ((size_t)b * T + t) * F;
                const int p = (P == 4) ? (t & 3) : 0;
                const int c = (P == 4) ? (t >> 2) + 1 : t + 1;
                const REAL rc_n = (REAL)1 / (REAL)c;
                if (FM) { /* summarize_megno (:480-484) sees x BEFORE zero_megno and before any noise (:488-491) */
                    REAL delta = xi[MEGNO_COL] - gmean[p];
                    REAL mnew = FMA(delta, rc_n, gmean[p]);
                    gm2[p] = FMA(delta, xi[MEGNO_COL] - mnew, gm2[p]);
                    gmean[p] = mnew;
                }
                REAL* cur = buf0;
                REAL* nxt = buf1;
                for (int f = 0; f < F; ++f) {
                    /* zero_megno/mmr/nan/eplusminus: `x = x - mask` with mask = x on the masked columns (:452-478): 0 for a finite
                     * value, NaN for NaN and +-inf (inf - inf) -- evaluated as the reference does; bits past 63 do not exist */
                    REAL v = (f < 64 && ((a->zero_mask >> f) & 1)) ? (REAL)(xi[f] - xi[f]) : xi[f];
                    if (eps_in) v = v + eps_in[((size_t)b * T + t) * F + f] * in_scale[f]; /* add_input_noise :444-446 */
                    cur[f] = v;
                }
                for (int l = 0; l < n1; ++l) { /* feature_nn (:359, :417) */
                    linear_row(Wl[l], bl[l], nout[l], nin[l], cur, nxt, l < n1 - 1, ord[l], ordn[l]);
                    REAL* tmp = cur; cur = nxt; nxt = tmp;
                }
                const REAL* y = cur;
                if (latents) memcpy(latents + ((size_t)b * T + t) * L, y, sizeof(REAL) * L);
                /* mean / unbiased variance over t (torch.mean, torch.std: :418-419): Welford with fused updates */
                for (int n = 0; n < L; ++n) {
                    REAL delta = y[n] - mean[p * L + n];
                    REAL mnew = FMA(delta, rc_n, mean[p * L + n]);
                    m2[p * L + n] = FMA(delta, y[n] - mnew, m2[p * L + n]);
                    mean[p * L + n] = mnew;
                }
            }
            if (P == 4) { /* merge partitions: (0,1), (2,3), then the two halves */
                for (int n = 0; n < L; ++n) {
                    merge_apply(&m01, &mean[n], &m2[n], mean[L + n], m2[L + n]);
                    merge_apply(&m23, &mean[2 * L + n], &m2[2 * L + n], mean[3 * L + n], m2[3 * L + n]);
                    merge_apply(&m0123, &mean[n], &m2[n], mean[2 * L + n], m2[2 * L + n]);
                }
                if (FM) {
                    merge_apply(&m01, &gmean[0], &gm2[0], gmean[1], gm2[1]);
                    merge_apply(&m23, &gmean[2], &gm2[2], gmean[3], gm2[3]);
                    merge_apply(&m0123, &gmean[0], &gm2[0], gmean[2], gm2[2]);
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
            REAL* cur = buf0;
            REAL* nxt = buf1;
            memcpy(cur, s, sizeof(REAL) * S);
            for (int l = n1; l < NL; ++l) { /* regress_nn (:360, :438) */
                linear_row(Wl[l], bl[l], nout[l], nin[l], cur, nxt, l < NL - 1, ord[l], ordn[l]);
                REAL* tmp = cur; cur = nxt; nxt = tmp;
            }
            const REAL r0 = cur[0], r1 = cur[1];
            if (pre_clamp) { pre_clamp[b * 2] = r0; pre_clamp[b * 2 + 1] = r1; }
            out[b * 2 + 0] = soft_clamp(r0, 4.0, 12.0);      /* :440 */
            out[b * 2 + 1] = soft_clamp(r1, a->lowest, 6.0); /* :441 */
        }
        free(buf0);
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
