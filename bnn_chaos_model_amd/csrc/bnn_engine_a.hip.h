// bnn_engine_a.hip.h -- feature_nn engine A: v_mfma_f32_16x16x4_f32, weights resident in registers, any column
// mask, quiet or noisy forward.  Included by bnn_kernels.hip after bnn_common.hip.h.
#pragma once
// ------------------------------------------------------------------------------------------------
// x tile -> B operands.  Lane (g, c) owns row `rowp` (41 floats) and k slots kmap_input(NK1, s, g).
// ------------------------------------------------------------------------------------------------
template <int NK1>
struct XTile {
    float v[NK1];
};
// Raw loaded registers.  The selects that build the B operands are applied at USE time (xtile()), never at
// load time: a select right after the load makes the compiler wait for the data there, which turns the
// one-tile-ahead prefetch into a stall of a full memory latency per tile.
template <int NK1>
struct XRaw {
    f32x4 a, b;
    f32x3 c;  // NK1 == 8: c.x = column 0
};

template <int NK1>
DEVINL XRaw<NK1> load_x(const float* __restrict__ rowp, int g) {
    XRaw<NK1> t;
    if constexpr (NK1 == 8) {
        const float* p = rowp + 8 + 8 * g;
        t.a = *reinterpret_cast<const f32x4u*>(p);
        t.b = *reinterpret_cast<const f32x4u*>(p + 4);  // group 3: columns 36..39, 38/39 replaced in xtile()
        t.c.x = rowp[0];
    } else {
        const float* p = rowp + 11 * g;
        t.a = *reinterpret_cast<const f32x4u*>(p);
        t.b = *reinterpret_cast<const f32x4u*>(p + 4);
        const float* pc = (g == 3) ? rowp + 38 : p + 8;  // group 3 has no columns 41..43: stay inside the row
        t.c = *reinterpret_cast<const f32x3u*>(pc);
    }
    return t;
}

template <int NK1>
DEVINL XTile<NK1> xtile(const XRaw<NK1>& r, int g) {
    XTile<NK1> t;
    t.v[0] = r.a.x; t.v[1] = r.a.y; t.v[2] = r.a.z; t.v[3] = r.a.w;
    t.v[4] = r.b.x; t.v[5] = r.b.y;
    if constexpr (NK1 == 8) {
        t.v[6] = (g == 3) ? r.c.x : r.b.z;   // slot (6, group 3) = column 0
        t.v[7] = (g == 3) ? 1.0f : r.b.w;    // slot (7, group 3) = bias
    } else {
        t.v[6] = r.b.z; t.v[7] = r.b.w;
        t.v[8] = (g == 3) ? 1.0f : r.c.x;    // slot 41 = bias
        t.v[9] = (g == 3) ? 0.0f : r.c.y;
        t.v[10] = (g == 3) ? 0.0f : r.c.z;
    }
    return t;
}

// ------------------------------------------------------------------------------------------------
// the fused kernel
// ------------------------------------------------------------------------------------------------
template <int NK1, bool NOISY, bool FUSED>
__global__ __launch_bounds__(256, (NOISY || NK1 != 8) ? 2 : BNN_WAVES_PER_SIMD) void bnn_multiswag_kernel(const FwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* flat = lds;                 // [FLAT_LDS] flat parameter vector + zero slot
    float* zsh = lds + FLAT_LDS;       // [MAXK]
    float* f2frag = lds;               // [NF2][64] regress_nn operands in fragment order: OVERWRITES flat (below)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;

    // ---- work item: draw e, block `sub` of its chunk of systems (torch.chunk semantics)
    const int64_t id = blockIdx.x;
    const int e = (int)(id % p.J);
    const int64_t sub = id / p.J;
    const int ch = e % p.nch;
    const int64_t r = e / p.nch;  // output row
    const int64_t seg0 = (int64_t)ch * p.csz;
    const int64_t seg1 = (seg0 + p.csz < p.B) ? seg0 + p.csz : p.B;
    const int64_t b0 = seg0 + sub * p.spc;
    const int64_t b1 = (b0 + p.spc < seg1) ? b0 + p.spc : seg1;
    if (b0 >= b1) return;

    // ---- prologue: flat parameter vector of draw e -> LDS
    bool bad_seed = false;
    if constexpr (FUSED) {
        int s = p.seed_idx[e];
        bad_seed = (s < 0 || s >= p.S);
        if (bad_seed) s = 0;
        const int K = p.K;
        if (tid < K) zsh[tid] = p.z2 ? p.z2[(int64_t)e * K + tid] : philox_z(TAG_Z2, p.draw_id0 + e, tid, p.seed);
        const float* wa = p.w_avg + (int64_t)s * D;
        const float* w2 = p.w2_avg + (int64_t)s * D;
        const float* pd = p.pre_D + (int64_t)s * D * K;
        __syncthreads();  // zsh
        for (int i = tid; i < D; i += 256) {
            float z1v = p.z1 ? p.z1[(int64_t)e * D + i] : philox_z(TAG_Z1, p.draw_id0 + e, i, p.seed);
            flat[i] = draw_row_direct(wa, w2, pd, i, K, zsh, z1v, p.c1, p.c2, p.scale);
        }
    } else {
        const float* We = p.W + (int64_t)e * D;
        for (int i = tid; i < D; i += 256) flat[i] = We[i];
    }
    if (tid == 0) flat[ZERO_IDX] = 0.0f;
    __syncthreads();

    // ---- feature_nn operands: registers for the whole workgroup lifetime
    constexpr int NW1 = 3 * NK1, IW2 = NW1, IW3 = IW2 + 30, IB2 = IW3 + 20, IB3 = IB2 + 12, NF1 = IB3 + 8;
    float wf[NF1];
#pragma unroll
    for (int f = 0; f < NF1; ++f) wf[f] = flat[p.tab_f1[f * 64 + lane]];

    float in_scale[NOISY ? NK1 : 1], sum_scale[NOISY ? 10 : 1];
    if constexpr (NOISY) {
#pragma unroll
        for (int s = 0; s < NK1; ++s) {
            int col = 11 * g + s;
            in_scale[s] = col < F ? expf(flat[OFF_INLV + col] / 2.0f) : 0.0f;  // exp(logvar/2), :445
        }
#pragma unroll
        for (int k = 0; k < 10; ++k) sum_scale[k] = expf(flat[OFF_SUMLV + kmap_summary(k, g)] / 2.0f);  // :449
    }
    // regress_nn operands -> LDS in fragment order (read back with immediate offsets, once per 16 systems).
    // They replace the flat vector in place: gather to registers, barrier, write.
    {
        constexpr int PER = (NF2 + 3) / 4;
        float tmp[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            int f = wave + 4 * i;
            tmp[i] = f < NF2 ? flat[p.tab_f2[f * 64 + lane]] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            int f = wave + 4 * i;
            if (f < NF2) f2frag[f * 64 + lane] = tmp[i];
        }
        __syncthreads();
    }

    const int T = p.T, ntiles = p.ntiles;
    const float nm1 = (float)(T - 1), nT = (float)T;
    const float half_n0 = (float)ntiles * 0.5f;
    const int64_t rowstride = (int64_t)T * F;

#if BNN_PRIO_STAGGER
    // Waves that share a SIMD run the same program and fall into lockstep (both in their VALU phase, then both
    // wanting the matrix pipe).  Give odd hardware wave slots priority so the partner fills the gaps instead.
    if (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1) __builtin_amdgcn_s_setprio(BNN_PRIO_STAGGER);
#endif
    // ---- wave-batches of 16 systems
    for (int64_t wb0 = b0 + (int64_t)wave * 16; wb0 < b1; wb0 += 64) {
        float skeep[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) skeep[k] = 0.0f;

        for (int q = 0; q < 4; ++q) {
            if (wb0 + 4 * q >= b1) break;  // wave-uniform
            const int64_t sys = wb0 + 4 * q + (c >> 2);
            const bool valid = sys < b1;
            const int64_t sysc = valid ? sys : b1 - 1;
            const float* rowp = p.x + sysc * rowstride + (int64_t)(c & 3) * F;
            const float* epin = nullptr;
            if constexpr (NOISY) {
                if (p.eps_in) epin = p.eps_in + (r * p.B + sysc) * rowstride + (int64_t)(c & 3) * F;
            }

            f32x4 mean0 = {0, 0, 0, 0}, m20 = {0, 0, 0, 0};
            float mean1 = 0.0f, m21 = 0.0f;

            // ---- building blocks of one 16-row tile ----------------------------------------------------------
            auto make_cur = [&](const XRaw<NK1>& raw, const XRaw<NK1>& nraw, const int it_cur) {
                XTile<NK1> cur = xtile<NK1>(raw, g);
                if constexpr (NOISY) {
                    // masks then add_input_noise (:486-504): masked columns become pure noise
                    XTile<NK1> ncur;
                    if (p.eps_in) {
                        ncur = xtile<NK1>(nraw, g);
                    } else {
                        // this lane's 11 columns 11g..11g+10 sit in Philox quads q0..q0+3 of row t (q0 = 11g/4)
                        const int t = 4 * it_cur + (c & 3);
                        const int q0 = (11 * g) >> 2, off = 11 * g - 4 * q0;
                        float f16[16];
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            f32x4 n4 = philox_sys4(TAG_IN, p.row_id0 + r, p.sys_id0 + sysc, t * 11 + q0 + b, p.seed);
                            f16[4 * b] = n4.x; f16[4 * b + 1] = n4.y; f16[4 * b + 2] = n4.z; f16[4 * b + 3] = n4.w;
                        }
#pragma unroll
                        for (int s = 0; s < NK1; ++s) {
                            float v0 = f16[s], v1 = f16[s + 1], v2 = f16[s + 2], v3 = f16[s + 3];
                            ncur.v[s] = off == 0 ? v0 : off == 1 ? v1 : off == 2 ? v2 : v3;
                        }
                    }
#pragma unroll
                    for (int s = 0; s < NK1; ++s) {
                        int col = 11 * g + s;
                        if (col < F) {
                            float xv = ((p.zero_mask >> col) & 1ull) ? 0.0f : cur.v[s];
                            cur.v[s] = xv + ncur.v[s] * in_scale[s];
                        }
                    }
                }
                return cur;
            };
            struct H3 { f32x4 m[3]; };
            struct H2 { f32x4 m[2]; };
            auto layer1 = [&](const XTile<NK1>& cur) {  // feature_nn.0 (bias rides in a k slot)
                H3 h = {{{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}};
#pragma unroll
                for (int s = 0; s < NK1; ++s)
#pragma unroll
                    for (int mt = 0; mt < 3; ++mt) h.m[mt] = mfma(wf[s * 3 + mt], cur.v[s], h.m[mt]);
                return h;
            };
            auto relu3 = [&](H3 h) {
                h.m[0] = relu4(h.m[0]); h.m[1] = relu4(h.m[1]); h.m[2] = relu4<2>(h.m[2]);
                return h;
            };
            auto layer2 = [&](const H3& h) {  // feature_nn.2
                H3 o;
#pragma unroll
                for (int mt = 0; mt < 3; ++mt) o.m[mt] = (f32x4){wf[IB2 + mt * 4], wf[IB2 + mt * 4 + 1], wf[IB2 + mt * 4 + 2], wf[IB2 + mt * 4 + 3]};
#pragma unroll
                for (int ks = 0; ks < NKH; ++ks)
#pragma unroll
                    for (int mt = 0; mt < 3; ++mt) o.m[mt] = mfma(wf[IW2 + ks * 3 + mt], h.m[ks >> 2][ks & 3], o.m[mt]);
                return o;
            };
            auto layer3 = [&](const H3& h2) {  // feature_nn.4
                H2 y;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) y.m[mt] = (f32x4){wf[IB3 + mt * 4], wf[IB3 + mt * 4 + 1], wf[IB3 + mt * 4 + 2], wf[IB3 + mt * 4 + 3]};
#pragma unroll
                for (int ks = 0; ks < NKH; ++ks)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) y.m[mt] = mfma(wf[IW3 + ks * 2 + mt], h2.m[ks >> 2][ks & 3], y.m[mt]);
                return y;
            };
            // torch.mean / torch.std over time (:418-419): Welford (fused updates) over this lane's timesteps, in tile order;
            // 1/(it+1) comes correctly rounded from a table
            auto pool = [&](const H2& y, const int it) {
                const float rcn = p.rcp_tab[it];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float dl = y.m[0][i] - mean0[i];
                    float mn = fmaf(dl, rcn, mean0[i]);
                    m20[i] = fmaf(dl, y.m[0][i] - mn, m20[i]);
                    mean0[i] = mn;
                }
                {
                    float dl = y.m[1][0] - mean1;
                    float mn = fmaf(dl, rcn, mean1);
                    m21 = fmaf(dl, y.m[1][0] - mn, m21);
                    mean1 = mn;
                }
            };
            // Loads stay where they are written: without the may-write barrier InstCombine folds phi(load, load) into a
            // load of phi(addresses) in front of the first use, and the machine scheduler sinks it further.
            auto prefetch = [&](XRaw<NK1>& raw, XRaw<NK1>& nraw, int it) {
                const int itc = it < ntiles ? it : ntiles - 1;  // past the end: re-read the last tile (no overrun)
                raw = load_x<NK1>(rowp + (int64_t)itc * 4 * F, g);
                if constexpr (NOISY) {
                    if (p.eps_in) nraw = load_x<NK1>(epin + (int64_t)itc * 4 * F, g);
                }
            };
            auto pin_loads = [&]() {
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            };

            XRaw<NK1> rawA, rawB, nrawA, nrawB;
            int it = 0;
            // one tile at a time, next tile prefetched into a ping-pong pair of register sets
            auto do_tile = [&](const XRaw<NK1>& raw, const XRaw<NK1>& nraw, const int t) {
                H2 y = layer3(relu3(layer2(relu3(layer1(make_cur(raw, nraw, t))))));
                pool(y, t);
            };
            prefetch(rawA, nrawA, 0);
            pin_loads();
            for (; it + 1 < ntiles; it += 2) {
                prefetch(rawB, nrawB, it + 1);
                pin_loads();
                do_tile(rawA, nrawA, it);
                prefetch(rawA, nrawA, it + 2);
                pin_loads();
                do_tile(rawB, nrawB, it + 1);
            }
            if (it < ntiles) do_tile(rawA, nrawA, it);

            // merge the 4 lanes of a quad (timesteps t = 4*it + (c&3)): equal-count Chan update, symmetric
            float mean[5] = {mean0[0], mean0[1], mean0[2], mean0[3], mean1};
            float m2[5] = {m20[0], m20[1], m20[2], m20[3], m21};
            float half_n = half_n0;
#pragma unroll
            for (int stage = 1; stage <= 2; stage <<= 1) {
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    float om = __shfl_xor(mean[k], stage), o2 = __shfl_xor(m2[k], stage);
                    float dl = om - mean[k];
                    float mm = (mean[k] + om) * 0.5f;
                    float qq = (m2[k] + o2) + (dl * dl) * half_n;
                    mean[k] = mm;
                    m2[k] = qq;
                }
                half_n = half_n * 2.0f;
            }

            // compute_summary_stats (:420-431) with the two randn_like draws
            f32x4 e1a, e2a;
            float e1b, e2b;
            if (p.eps) {
                const float* ep = p.eps + (r * p.B + sysc) * (2 * L);
                e1a = *reinterpret_cast<const f32x4*>(ep + 4 * g);
                e1b = ep[16 + g];
                e2a = *reinterpret_cast<const f32x4*>(ep + L + 4 * g);
                e2b = ep[L + 16 + g];
            } else {
                // The four lanes of a quad serve the same system and need the same four Philox blocks (quads g, 4, 5+g, 9
                // of that system's 40 normals): lane p of the quad generates block p, then the quad exchanges them
                // through the LDS crossbar (fp32 VALU time is matrix-pipe time on this chip; shuffles are not).
                const int64_t grow = p.row_id0 + r, gsys = p.sys_id0 + sysc;
                const int pq = c & 3;
                const int quad = pq == 0 ? g : pq == 1 ? 4 : pq == 2 ? 5 + g : 9;
                const f32x4 mine = philox_eps4(grow, gsys, quad, p.seed);
                const int qb = lane & ~3;
                const float pick = mine[g];  // lanes 1 and 3 of the quad only contribute component g
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    e1a[k] = __shfl(mine[k], qb + 0);
                    e2a[k] = __shfl(mine[k], qb + 2);
                }
                e1b = __shfl(pick, qb + 1);
                e2b = __shfl(pick, qb + 3);
            }
            float snew[10];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                float e1 = k < 4 ? e1a[k] : e1b, e2 = k < 4 ? e2a[k] : e2b;
                float sample_mu = mean[k];
                float sd = sqrtf(m2[k] / nm1);   // torch.std (unbiased)
                float sample_var = sd * sd;      // **2
                float std_in_mu = sqrtf(sample_var / nT);
                float std_in_var = sqrtf((2.0f * (sample_var * sample_var)) / nm1);
                float mu_s = e1 * std_in_mu + sample_mu;
                float var_s = e2 * std_in_var + sample_var;
                snew[k] = mu_s;
                snew[5 + k] = sqrtf(fabsf(var_s) + 1e-5f);  // EPSILON (:337)
            }
            if (p.summary && valid && (c & 3) == 0) {
                float* sp = p.summary + (r * p.B + sys) * S2;
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    int n = k < 4 ? 4 * g + k : 16 + g;
                    sp[n] = snew[k];
                    sp[L + n] = snew[5 + k];
                }
            }
            if ((c & 3) == q) {
#pragma unroll
                for (int k = 0; k < 10; ++k) skeep[k] = snew[k];
            }
        }

        // ---- regress_nn on 16 systems: column c <-> system wb0 + 4*(c&3) + (c>>2)
        const int64_t sysb = wb0 + 4 * (c & 3) + (c >> 2);
        const bool validb = sysb < b1;
        if constexpr (NOISY) {
            // add_summary_noise (:448-450)
            const int64_t sc = validb ? sysb : b1 - 1;
            if (p.eps_sum) {
                const float* es = p.eps_sum + (r * p.B + sc) * S2;
#pragma unroll
                for (int k = 0; k < 10; ++k) skeep[k] = skeep[k] + es[kmap_summary(k, g)] * sum_scale[k];
            } else {
                const int64_t grow = p.row_id0 + r, gsys = p.sys_id0 + sc;
#pragma unroll
                for (int kind = 0; kind < 2; ++kind) {
                    f32x4 a4n = philox_sys4(TAG_SUM, grow, gsys, kind * 5 + g, p.seed);
                    float bn = philox_sys4(TAG_SUM, grow, gsys, kind * 5 + 4, p.seed)[g];
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) skeep[kind * 5 + rr] = skeep[kind * 5 + rr] + a4n[rr] * sum_scale[kind * 5 + rr];
                    skeep[kind * 5 + 4] = skeep[kind * 5 + 4] + bn * sum_scale[kind * 5 + 4];
                }
            }
        }
        const float* f2l = f2frag + lane;
        auto W2f = [&](int f) { return f2l[f * 64]; };
        f32x4 a4[3], a5[3], a6;
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) a4[mt] = (f32x4){W2f(70 + mt * 4), W2f(71 + mt * 4), W2f(72 + mt * 4), W2f(73 + mt * 4)};
#pragma unroll
        for (int ks = 0; ks < 10; ++ks)
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) a4[mt] = mfma(W2f(ks * 3 + mt), skeep[ks], a4[mt]);
        a4[0] = relu4(a4[0]); a4[1] = relu4(a4[1]); a4[2] = relu4<2>(a4[2]);
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) a5[mt] = (f32x4){W2f(82 + mt * 4), W2f(83 + mt * 4), W2f(84 + mt * 4), W2f(85 + mt * 4)};
#pragma unroll
        for (int ks = 0; ks < NKH; ++ks)
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) a5[mt] = mfma(W2f(30 + ks * 3 + mt), a4[ks >> 2][ks & 3], a5[mt]);
        a5[0] = relu4(a5[0]); a5[1] = relu4(a5[1]); a5[2] = relu4<2>(a5[2]);
        a6 = (f32x4){W2f(94), W2f(95), W2f(96), W2f(97)};
#pragma unroll
        for (int ks = 0; ks < NKH; ++ks) a6 = mfma(W2f(60 + ks), a5[ks >> 2][ks & 3], a6);

        if (g == 0 && validb) {
            // predict_instability + soft_clamp (:295-296, :437-442)
            float r0 = a6[0], r1 = a6[1];
            float mu = (0.5f * (tanhf(r0) + 1.0f)) * 8.0f + 4.0f;
            float sd = (0.5f * (tanhf(r1) + 1.0f)) * p.std_span + p.std_lo;
            if (bad_seed) mu = sd = __builtin_nanf("");
            const int64_t o = (r * p.B + sysb) * 2;
            *reinterpret_cast<f32x2*>(p.out + o) = (f32x2){mu, sd};
            if (p.pre_clamp) *reinterpret_cast<f32x2*>(p.pre_clamp + o) = (f32x2){r0, r1};
        }
    }
}


