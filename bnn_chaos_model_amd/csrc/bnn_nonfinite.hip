// bnn_nonfinite.hip -- the reference's behaviour on NON-FINITE inputs.
//
// The reference masks by subtraction, `x = x - mask` (spock_reg_model.py:452-478): NaN / +-inf in a MASKED column becomes NaN, not 0;
// nn.Linear multiplies it into every neuron and nn.ReLU (:301-321) propagates NaN, keeps +inf and turns -inf into 0.  The forward
// kernels never read the masked columns of the pretrained mask, and their ReLU is an integer max on the bit pattern (a NaN with the
// sign bit set becomes 0): fast, and identical to the reference on finite data only.  So, once per call:
//   1. bnn_nonfinite_scan_kernel streams x ONCE (not once per draw; HBM-bound: 16.4 GB in ~3 ms at configs[2] next to a 560 ms step)
//      and lists the systems that hold a non-finite value anywhere in their [T, F] block, with one bit: "certainly NaN" = a NaN
//      anywhere, or +-inf in a masked column -- for those the reference's (mu, std) is NaN whatever the weights;
//   2. the forward kernels run unchanged;
//   3. bnn_nonfinite_fixup_kernel re-evaluates the listed systems for every output row with plain IEEE arithmetic -- x - x on the masked
//      columns, fmaf chains in the natural order (bias, inputs ascending: the generic engine's), a ReLU that propagates NaN -- and
//      overwrites their outputs.  "Certainly NaN" systems are answered without the evaluation unless a side-effect output (summary,
//      latents) was asked for.  An infinity in a live column is NOT certain: where it meets weights of one sign only it dies in the
//      ReLU and the reference's outputs stay finite (tests/golden/make_golden_nonfinite.py, the `dead` network).
// One four-wave workgroup per (output row, listed system); the network is read from the plan's descriptor (any hparams-built network, either engine).
#include "bnn_common.hip.h"
#include "bnn_stats.hip.h"

namespace bnn {

// ---- 1. the scan --------------------------------------------------------------------------------------------------------------
// record (int32): [0] = listed systems, [1] = of which certainly NaN, [2], [3] reserved (zero); [4 + i] = (system << 1) | certain.
DEVINL bool nf_bits(uint32_t b) { return (b & 0x7f800000u) == 0x7f800000u; }

// One wave looks through one system's [T, F] block: any = it holds NaN / +-inf, certain = a NaN anywhere or +-inf in a masked column.
DEVINL void nf_scan_system(const float* __restrict__ xs, int64_t per, int F, uint64_t zero_mask, int lane, bool& wany, bool& wcert) {
    bool any = false, certain = false;
    auto look = [&](float v, int64_t i) {   // the rare path: which kind, which column
        const uint32_t bits = __float_as_uint(v);
        if (!nf_bits(bits)) return;
        any = true;
        const int col = (int)(i % F);
        if ((bits & 0x007fffffu) != 0u || (col < 64 && ((zero_mask >> col) & 1ull))) certain = true;
    };
    const int64_t nvec = per >> 2;
    for (int64_t q0 = lane; q0 < nvec; q0 += 256) {   // four 16-byte loads in flight per lane (1 KB per wave each), then the tests
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t qq = q0 + 64 * u;
            v[u] = *reinterpret_cast<const f32x4u*>(xs + 4 * (qq < nvec ? qq : q0));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t qq = q0 + 64 * u;
            // exponent all ones <=> NaN or +-inf
            const uint32_t m0 = ~__float_as_uint(v[u].x) & 0x7f800000u, m1 = ~__float_as_uint(v[u].y) & 0x7f800000u,
                           m2 = ~__float_as_uint(v[u].z) & 0x7f800000u, m3 = ~__float_as_uint(v[u].w) & 0x7f800000u;
            if (qq < nvec && (m0 == 0u || m1 == 0u || m2 == 0u || m3 == 0u)) {
                look(v[u].x, 4 * qq); look(v[u].y, 4 * qq + 1); look(v[u].z, 4 * qq + 2); look(v[u].w, 4 * qq + 3);
            }
        }
    }
    for (int64_t i = 4 * nvec + lane; i < per; i += 64) look(xs[i], i);
    wany = __ballot(any) != 0ull;
    wcert = __ballot(certain) != 0ull;
}

__global__ __launch_bounds__(256) void bnn_nonfinite_scan_kernel(const float* __restrict__ x, int64_t B, int64_t per, int F, uint64_t zero_mask,
                                                                 int32_t* __restrict__ rec) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    bool wany, wcert;
    nf_scan_system(x + b * per, per, F, zero_mask, lane, wany, wcert);
    if (lane == 0 && wany) {
        const int slot = atomicAdd(&rec[0], 1);
        if (wcert) atomicAdd(&rec[1], 1);
        rec[4 + slot] = (int32_t)((b << 1) | (wcert ? 1 : 0));
    }
}

// A batch of at most NF_SMALL_MAX systems (the evaluation scripts' 15-row chunks, figures/multiswag_5_planet.py:295-298): ONE workgroup of
// sixteen waves scans it, keeps the list in LDS and writes header AND entries itself -- no header to clear beforehand, one launch instead of two.
constexpr int NF_SMALL_MAX = 64;
__global__ __launch_bounds__(1024) void bnn_nonfinite_scan_small_kernel(const float* __restrict__ x, int B, int64_t per, int F, uint64_t zero_mask,
                                                                        int32_t* __restrict__ rec) {
    __shared__ int cnt[2];
    __shared__ int list[NF_SMALL_MAX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
    __syncthreads();
    for (int b = wave; b < B; b += 16) {
        bool wany, wcert;
        nf_scan_system(x + (int64_t)b * per, per, F, zero_mask, lane, wany, wcert);
        if (lane == 0 && wany) {
            const int slot = atomicAdd(&cnt[0], 1);
            if (wcert) atomicAdd(&cnt[1], 1);
            list[slot] = (b << 1) | (wcert ? 1 : 0);
        }
    }
    __syncthreads();
    if (threadIdx.x < 4) rec[threadIdx.x] = threadIdx.x < 2 ? cnt[threadIdx.x] : 0;
    if ((int)threadIdx.x < cnt[0]) rec[4 + threadIdx.x] = list[threadIdx.x];
}

// The record's header is cleared by a KERNEL, never by hipMemsetAsync.  Round 5's first form used hipMemsetAsync(rec, 0, 16) and failed
// test_hip_graph_capture_and_replay on the SECOND replay: with x finite in both replays, system 0 came back re-evaluated by the fix-up in
// IEEE order (14 of 40 rows differed in the last bits).  That symptom means the fix-up saw count >= 1 with all-zero entries -- a NON-ZERO
// HEADER at fix-up time -- which a skipped memset cannot produce (replay 1 left the header 0).  Round 6 established the cause with a probe
// that holds no code of this library (scripts/dev/graph_nf_probe3.py micro -> profiles/r06_graph_memset_probe.txt): a hipMemsetAsync(ptr,
// 0, n) captured by torch.cuda.graph writes zeros on the FIRST launch of the graph exec and, on every later launch, a pointer-like 64-bit
// pattern (0x78a3_d2e0_0000 ...: [-757063680, 30883, ...] as int32) -- for n = 16, 64 and 4096, into ordinary (not graph-pool) memory;
// with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 every launch writes zeros.  So the memset node is neither skipped nor reordered: the HIP
// runtime's graph PACKET CAPTURE (torch 2.10.0+rocm7.0's libamdhip64) replays it with a wrong fill VALUE.  The record's storage plays no
// part (caller-owned record and graph-pool record behave alike, both libraries).  A garbage header whose first word is positive makes
// the fix-up walk that many "listed systems", whose entries read as zeros = "system 0, not certain": round 5's symptom.  (And with a
// damaged x the scan's atomic append lands wherever the garbage count points: the probe's one GPU memory fault.)  Kernel nodes replay correctly,
// and the captured graph of the current route is one dependency chain reset -> scan -> (draw ->) forward -> fix-up (probe `nodes`).
// Consequence for this library: NO hipMemsetAsync on any stream-ordered path (the moments slab driver likewise).
__global__ void bnn_nonfinite_reset_kernel(int32_t* __restrict__ rec) {
    if (threadIdx.x < 4) rec[threadIdx.x] = 0;
}

hipError_t launch_nonfinite_scan(const float* x, int64_t B, int64_t per, int F, uint64_t zero_mask, int32_t* rec, hipStream_t st) {
    if (B > 0 && B <= NF_SMALL_MAX) {
        hipLaunchKernelGGL(bnn_nonfinite_scan_small_kernel, dim3(1), dim3(1024), 0, st, x, (int)B, per, F, zero_mask, rec);
        return hipGetLastError();
    }
#if defined(BNN_NF_HEADER_MEMSET)   // PROBE BUILDS ONLY (scripts/dev/graph_nf_probe3.py): round 5's first form, a memset node in front of the kernels
    hipError_t e = hipMemsetAsync(rec, 0, 4 * sizeof(int32_t), st);
#else
    hipLaunchKernelGGL(bnn_nonfinite_reset_kernel, dim3(1), dim3(64), 0, st, rec);
    hipError_t e = hipGetLastError();
#endif
    if (e != hipSuccess || B == 0) return e;
    hipLaunchKernelGGL(bnn_nonfinite_scan_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, x, B, per, F, zero_mask, rec);
    return hipGetLastError();
}

// ---- 3. the fix-up --------------------------------------------------------------------------------------------------------------
// Chan merge of (na, ma, qa) with (nb, mb, qb), general form, counts as floats; an empty side leaves the other untouched
DEVINL void nf_merge(float& na, float& ma, float& qa, float nb, float mb, float qb) {
    if (nb == 0.0f) return;
    if (na == 0.0f) { na = nb; ma = mb; qa = qb; return; }
    const float n = na + nb, dl = mb - ma;
    ma = fmaf(dl, nb / n, ma);
    qa = (qa + qb) + (dl * dl) * (na * nb / n);
    na = n;
}

// Loads through the constant address space: for a wave-uniform address the compiler emits SCALAR loads (s_load_dwordx8 / x16 into SGPRs,
// which v_fmac reads directly) instead of one vector load per weight and lane.  W[e] was written by an earlier kernel (the draw) or by
// the caller, never by this one, so the scalar cache's lack of coherence with this kernel's own stores does not matter.
typedef const float __attribute__((address_space(4))) cfloat4;
DEVINL const cfloat4* as_constant(const float* p) { return (const cfloat4*)(uintptr_t)p; }
// the plan's descriptor likewise: every shape, offset and flag it holds is wave-uniform and lands in SGPRs
typedef const GenArch __attribute__((address_space(4))) GenArch4;
struct NfLayer { int K, N, off_w, off_b, relu; };
DEVINL NfLayer nf_layer(const GenArch4* g, int l) {
    NfLayer y;
    y.K = g->layer[l].K; y.N = g->layer[l].N; y.off_w = g->layer[l].off_w; y.off_b = g->layer[l].off_b; y.relu = g->layer[l].relu;
    return y;
}

// One WORKGROUP of four waves per (output row, listed system): lane = timestep as before, the four waves split every layer's neurons
// (and the input columns, the latents of the pool) between them and meet at a barrier per layer -- the LDS rows (34 KB for the pretrained
// shapes) are then shared by four waves instead of one: sixteen waves per CU instead of four, and a quarter of the dependent fmaf
// chains per wave.
// WLDS: the draw's flat parameter vector sits in LDS (the in-prologue draw has no other place for it: 30 KB more); otherwise it is read
// where it is, W[e] in global memory -- wave-uniform addresses through the constant address space, i.e. SCALAR loads.
template <bool WLDS>
__global__ __launch_bounds__(256) void bnn_nonfinite_fixup_kernel(const NfxParams q) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const FwdParams& p = q.f;
    // an empty list (the normal case: this launch follows EVERY forward launch) is left before anything else is read
    if (__builtin_amdgcn_readfirstlane(q.rec[0]) <= 0) return;
    const GenArch4* Gc = (const GenArch4*)(uintptr_t)q.g;
    struct { int F, L, SM, d, megno, n_feat, n_reg, nin_blocks, off_inlv, off_sumlv; } G = {Gc->F, Gc->L, Gc->SM, Gc->d, Gc->megno, Gc->n_feat, Gc->n_reg,
                                                                                            Gc->nin_blocks, Gc->off_inlv, Gc->off_sumlv};
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int F = G.F, L = G.L, SM = G.SM, d = G.d, T = p.T;
    int count = __builtin_amdgcn_readfirstlane(q.rec[0]);
    if (count < 0) count = 0;
    if ((int64_t)count > p.B) count = (int)p.B;   // (never walk further than a record of this call's batch can be long)
    const int R = p.J / p.nch;
    const int64_t total = (int64_t)R * count;
    if (total == 0) return;
    float* actA = lds;                           // [maxw][64] activations of the 64 timesteps of a pass, lane-minor
    float* actB = actA + (size_t)q.maxw * 64;
    float* pm = actB + (size_t)q.maxw * 64;      // [L + 1][64] per-lane Welford means (row L: the raw MEGNO column)
    float* pq = pm + (size_t)(L + 1) * 64;       // [L + 1][64] ... M2
    float* sum = pq + (size_t)(L + 1) * 64;      // [128] summary
    float* ra = sum + 128;                       // [128], [128] regress_nn activations
    float* rb = ra + 128;
    float* zsh = rb + 128;                       // [256] z2 of an in-prologue draw
    float* flat = zsh + 256;                     // [d] the draw's flat parameter vector (in-prologue draw only)
    const float nm1 = (float)(T - 1), nT = (float)T;
    const float qnan = __builtin_nanf("");

    for (int64_t w = blockIdx.x; w < total; w += gridDim.x) {
        const int32_t ent = __builtin_amdgcn_readfirstlane(q.rec[4 + (int)(w % count)]);   // (wave-uniform: keeps the draw's address scalar)
        const int64_t r = w / count, b = ent >> 1;
        if (b < 0 || b >= p.B) continue;   // (a record made for another batch: never write outside this call's rows; uniform over the workgroup)
        const int ch = (int)((p.coff + b) / p.csz);
        const int e = (int)r * p.nch + ch;
        const int64_t grow = p.row_id0 + r, gsys = p.sys_id0 + b;
        const int64_t ob = r * p.B + b;
        auto emit = [&](float r0, float r1, bool nan_out) {
            if (tid != 0) return;
            f32x2 ms = soft_clamp2(r0, r1, p.std_lo, p.std_span);
            if (nan_out) ms.x = ms.y = qnan;
            if (p.sink) p.sink[ob] = stats_draw(p.st, ms.x, ms.y, grow, gsys, p.seed);
            else if (p.out) {
                *reinterpret_cast<f32x2*>(p.out + ob * 2) = ms;
                if (p.pre_clamp) *reinterpret_cast<f32x2*>(p.pre_clamp + ob * 2) = (f32x2){nan_out ? qnan : r0, nan_out ? qnan : r1};
            }
        };
        if ((ent & 1) && q.shortcut) {   // NaN whatever the weights (uniform over the workgroup: no barrier is skipped by part of it)
            emit(qnan, qnan, true);
            continue;
        }
        // the draw's flat parameter vector: materialised (workspace / VarModel.forward), or sampled here like the forward kernel's prologue
        const float* __restrict__ wv;
        bool bad_seed = false;
        if constexpr (!WLDS) {
            wv = p.W + (int64_t)e * d;
        } else if (p.W) {
            for (int i = tid; i < d; i += 256) flat[i] = p.W[(int64_t)e * d + i];
            wv = flat;
        } else {
            int s = p.seed_idx[e];
            bad_seed = (s < 0 || s >= p.S);
            if (bad_seed) s = 0;
            const int K = p.K;
            for (int k = tid; k < K; k += 256) zsh[k] = p.z2 ? p.z2[(int64_t)e * K + k] : philox_z(TAG_Z2, p.draw_id0 + e, k, p.seed);
            __syncthreads();
            const float* wa = p.w_avg + (int64_t)s * d;
            const float* w2 = p.w2_avg + (int64_t)s * d;
            const float* pd = p.pre_D + (int64_t)s * d * K;
            for (int i = tid; i < d; i += 256) {
                const float z1v = p.z1 ? p.z1[(int64_t)e * d + i] : philox_z(TAG_Z1, p.draw_id0 + e, i, p.seed);
                flat[i] = draw_row_direct(wa, w2, pd, i, K, zsh, z1v, p.c1, p.c2, p.scale);
            }
            wv = flat;
        }
        for (int n = wave; n <= L; n += 4) { pm[n * 64 + lane] = 0.0f; pq[n * 64 + lane] = 0.0f; }
        __syncthreads();
        const float* xs = p.x + b * (int64_t)T * F;
        int npass = 0;
        for (int t0 = 0; t0 < T; t0 += 64, ++npass) {
            const int t = t0 + lane;
            const bool tv = t < T;
            const int tc = tv ? t : T - 1;
            const float* xr = xs + (int64_t)tc * F;
            const float rcn = 1.0f / (float)(npass + 1);
            if (G.megno && tv && wave == 0) {   // summarize_megno (:480-484): the RAW column, before the masks and before any noise
                const float xm = xr[MEGNO_COL];
                const float gm = pm[L * 64 + lane];
                const float dl = xm - gm, mn = fmaf(dl, rcn, gm);
                pq[L * 64 + lane] = fmaf(dl, xm - mn, pq[L * 64 + lane]);
                pm[L * 64 + lane] = mn;
            }
            // zero_megno / zero_mmr / zero_nan / zero_eplusminus: x - mask (:452-478); then add_input_noise (:444-446); a wave takes every
            // fourth block of six columns (one Philox block of the input-noise stream)
            const float* er = (q.noisy && p.eps_in) ? p.eps_in + (ob * T + tc) * (int64_t)F : nullptr;
            for (int k0 = NIN_PER_BLOCK * wave; k0 < F; k0 += 4 * NIN_PER_BLOCK) {
                float n6[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                if (q.noisy && !er) philox_in6(grow, gsys, tc * G.nin_blocks + k0 / NIN_PER_BLOCK, p.seed, n6);
                for (int j = 0; j < NIN_PER_BLOCK && k0 + j < F; ++j) {
                    const int k = k0 + j;
                    const float xv = xr[k];
                    float v = (k < 64 && ((p.zero_mask >> k) & 1ull)) ? xv - xv : xv;
                    if (q.noisy) {
                        const float nz = (er ? er[k] : (j == 0 ? n6[0] : j == 1 ? n6[1] : j == 2 ? n6[2] : j == 3 ? n6[3] : j == 4 ? n6[4] : n6[5])) *
                                         expf(wv[G.off_inlv + k] / 2.0f);
                        v = v + nz;
                    }
                    actA[k * 64 + lane] = v;
                }
            }
            __syncthreads();
            float* cur = actA;
            float* nxt = actB;
            for (int l = 0; l < G.n_feat; ++l) {   // feature_nn (:359, :417): bias, then the inputs ascending -- four neurons at a time: four
                const NfLayer ly = nf_layer(Gc, l);   // independent fmaf chains share one activation read (a lone chain waits out every latency)
                const int K = ly.K, N = ly.N;
                auto four = [&](auto wbase, int n0) {   // wbase: the flat vector through the address space it lives in
                    const auto w0 = wbase + ly.off_w + (int64_t)n0 * K;
                    const auto w1 = wbase + ly.off_w + (int64_t)(n0 + 1 < N ? n0 + 1 : N - 1) * K;
                    const auto w2 = wbase + ly.off_w + (int64_t)(n0 + 2 < N ? n0 + 2 : N - 1) * K;
                    const auto w3 = wbase + ly.off_w + (int64_t)(n0 + 3 < N ? n0 + 3 : N - 1) * K;
                    float a0 = wbase[ly.off_b + n0], a1 = wbase[ly.off_b + (n0 + 1 < N ? n0 + 1 : N - 1)],
                          a2 = wbase[ly.off_b + (n0 + 2 < N ? n0 + 2 : N - 1)], a3 = wbase[ly.off_b + (n0 + 3 < N ? n0 + 3 : N - 1)];
#pragma unroll 8
                    for (int k = 0; k < K; ++k) {
                        const float xk = cur[k * 64 + lane];
                        a0 = fmaf(w0[k], xk, a0); a1 = fmaf(w1[k], xk, a1); a2 = fmaf(w2[k], xk, a2); a3 = fmaf(w3[k], xk, a3);
                    }
                    if (ly.relu) { a0 = relu_ieee(a0); a1 = relu_ieee(a1); a2 = relu_ieee(a2); a3 = relu_ieee(a3); }
                    nxt[n0 * 64 + lane] = a0;
                    if (n0 + 1 < N) nxt[(n0 + 1) * 64 + lane] = a1;
                    if (n0 + 2 < N) nxt[(n0 + 2) * 64 + lane] = a2;
                    if (n0 + 3 < N) nxt[(n0 + 3) * 64 + lane] = a3;
                };
                for (int n0 = 4 * wave; n0 < N; n0 += 16) {   // this wave's neurons
                    if constexpr (WLDS) four(wv, n0);
                    else four(as_constant(wv), n0);
                }
                __syncthreads();
                float* tmp = cur; cur = nxt; nxt = tmp;
            }
            if (tv) {   // torch.mean / torch.std over time (:418-419): Welford over this lane's timesteps t0 + lane; a wave takes every fourth latent
                float* lat = p.latents ? p.latents + (ob * T + t) * (int64_t)L : nullptr;
                for (int n = wave; n < L; n += 4) {
                    const float y = cur[n * 64 + lane];
                    if (lat) lat[n] = y;
                    const float m = pm[n * 64 + lane];
                    const float dl = y - m, mn = fmaf(dl, rcn, m);
                    pq[n * 64 + lane] = fmaf(dl, y - mn, pq[n * 64 + lane]);
                    pm[n * 64 + lane] = mn;
                }
            }
            __syncthreads();   // the next pass writes its inputs over rows that may still be read here
        }
        // merge the 64 partitions (t mod 64) of a latent in partition order; thread n finishes latent n (compute_summary_stats :420-431)
        for (int n = tid; n <= L; n += 256) {
            if (n == L && !G.megno) break;
            float na = 0.0f, ma = 0.0f, qa = 0.0f;
            for (int pp = 0; pp < 64; ++pp) {
                const float nb = pp < T ? (float)((T - pp + 63) / 64) : 0.0f;
                nf_merge(na, ma, qa, nb, pm[n * 64 + pp], pq[n * 64 + pp]);
            }
            if (n < L) {
                float e1, e2;
                if (p.eps) {
                    const float* ep = p.eps + ob * 2 * L;
                    e1 = ep[n]; e2 = ep[L + n];
                } else {
                    e1 = philox_eps4(grow, gsys, n >> 2, p.seed)[n & 3];
                    e2 = philox_eps4(grow, gsys, (L + n) >> 2, p.seed)[(L + n) & 3];
                }
                float mu_s, sd_s;
                sampled_moments(ma, qa, e1, e2, nm1, nT, mu_s, sd_s);
                sum[n] = mu_s;
                sum[L + n] = sd_s;
            } else {   // torch.cat([summary_stats, megno_avg_std]) (:509-510): mean and unbiased std of the raw column
                sum[2 * L] = ma;
                sum[2 * L + 1] = sqrtf(qa / nm1);
            }
        }
        __syncthreads();
        for (int n = tid; n < SM; n += 256) {
            float s = sum[n];
            if (p.summary) p.summary[ob * SM + n] = s;   // _cur_summary (:512): before the summary noise
            if (q.noisy) {                               // add_summary_noise (:448-450)
                const float nz = p.eps_sum ? p.eps_sum[ob * SM + n] : philox_sys4(TAG_SUM, grow, gsys, n >> 2, p.seed)[n & 3];
                s = s + nz * expf(wv[G.off_sumlv + n] / 2.0f);
            }
            ra[n] = s;
        }
        __syncthreads();
        float* cur = ra;
        float* nxt = rb;
        for (int l = G.n_feat; l < G.n_feat + G.n_reg; ++l) {   // regress_nn (:360, :438): thread = neuron
            const NfLayer ly = nf_layer(Gc, l);
            for (int n = tid; n < ly.N; n += 256) {
                float acc = wv[ly.off_b + n];
                const float* __restrict__ wr = wv + ly.off_w + (int64_t)n * ly.K;
                int k = 0;
#pragma unroll 4
                for (; k + 4 <= ly.K; k += 4) {   // the thread's weight row, 16 bytes at a time (several loads in flight: a load per fmaf would serialise the chain)
                    const f32x4 w4 = *reinterpret_cast<const f32x4u*>(wr + k);
                    acc = fmaf(w4.x, cur[k], acc); acc = fmaf(w4.y, cur[k + 1], acc); acc = fmaf(w4.z, cur[k + 2], acc); acc = fmaf(w4.w, cur[k + 3], acc);
                }
                for (; k < ly.K; ++k) acc = fmaf(wr[k], cur[k], acc);
                nxt[n] = ly.relu ? relu_ieee(acc) : acc;
            }
            __syncthreads();
            float* tmp = cur; cur = nxt; nxt = tmp;
        }
        emit(cur[0], cur[1], bad_seed);   // predict_instability + soft_clamp (:295-296, :437-442)
        __syncthreads();  // the LDS areas are reused by the next item
    }
}

size_t nonfinite_fixup_lds_bytes(const GenArch& g, bool wlds) {
    int maxw = g.F;
    for (int l = 0; l < g.n_feat; ++l) maxw = g.layer[l].N > maxw ? g.layer[l].N : maxw;
    return sizeof(float) * ((size_t)2 * maxw * 64 + (size_t)2 * (g.L + 1) * 64 + 3 * 128 + 256 + (wlds ? (size_t)g.d : 0));
}

hipError_t launch_nonfinite_fixup(const GenArch& g, NfxParams& q, hipStream_t st) {
    int maxw = g.F;
    for (int l = 0; l < g.n_feat; ++l) maxw = g.layer[l].N > maxw ? g.layer[l].N : maxw;
    q.maxw = maxw;
    // the flat vector in LDS only where there is no other place for it (the in-prologue draw): with W[e] in global memory the 34 KB
    // workgroup fits four times on a CU and measured 6.9e6 (row, system) evaluations/s against 4.6e6 with the weights in LDS (two
    // workgroups per CU, LDS-issue bound) -- and 1.7e6 for the first version (one fmaf chain at a time, generic-pointer loads)
    const bool fused = q.f.W == nullptr;
    const size_t lds = nonfinite_fixup_lds_bytes(g, fused);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    // the list's length is known on the device only: a fixed grid walks it (every workgroup leaves at once when it is empty)
    const int64_t items = (int64_t)(q.f.J / q.f.nch) * q.f.B;
    const unsigned nblk = (unsigned)(items < 4096 ? (items > 0 ? items : 1) : 4096);
    if (fused) {
        allow_big_lds<bnn_nonfinite_fixup_kernel<true>>();
        hipLaunchKernelGGL(bnn_nonfinite_fixup_kernel<true>, dim3(nblk), dim3(256), lds, st, q);
    } else {
        allow_big_lds<bnn_nonfinite_fixup_kernel<false>>();
        hipLaunchKernelGGL(bnn_nonfinite_fixup_kernel<false>, dim3(nblk), dim3(256), lds, st, q);
    }
    return hipGetLastError();
}

}  // namespace bnn
