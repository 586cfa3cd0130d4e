// bnn_abi.hip -- the C ABI of include/bnn_chaos_hip.h (libbnn_chaos_hip.so, gfx950 / MI355X): plans and their specialised forms, and every
// entry point that launches a forward kernel (forward, multiswag, statistics tail, slab drivers, latents, reduced precision) with the
// non-finite scan.  The small kernels and their entry points live in the bnn_ops_*.hip units (bnn_abi_common.h lists them); the forward
// kernels in bnn_forward.hip.h / bnn_generic.hip.h / bnn_lowp.hip.h, instantiated by the bnn_fwd_*.hip units (bnn_internal.h).
//
// Reference path (MilesCranmer/bnn_chaos_model): SWAGModel.sample_weights + forward_swag_fast / VarModel.forward in
// spock_reg_model.py:415-450, 486-528, 815-908, driven by figures/spock/regression.py:74-92 and
// figures/multiswag_5_planet.py:295-298.  DESIGN.md section 4 has the long form.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (explicit fmaf/MFMA are the only fusions).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "bnn_abi_common.h"
#include "bnn_common.hip.h"

using namespace bnn;

static thread_local std::string g_err;
int bnn_fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

static bool is_v50net(const bnn_arch* a) {
    return a->n_features == F && a->hidden == H && a->latent == L && a->depth_in == 1 && a->depth_out == 1;
}

static int check_arch(const bnn_arch* a, GenArch* gen_out = nullptr) {
    if (!a) return fail(BNN_ERR_INVALID, "arch is NULL");
    if (a->fix_megno != 0 && a->fix_megno != 1) return fail(BNN_ERR_INVALID, "fix_megno must be 0 or 1");
    GenArch g;
    const char* why = "";
    if (gen_build(a->n_features, a->hidden, a->latent, a->depth_in, a->depth_out, a->fix_megno != 0, &g, &why))
        return fail(BNN_ERR_UNSUPPORTED, why);
    if (a->n_features < 64 && (a->zero_mask >> a->n_features)) return fail(BNN_ERR_INVALID, "zero_mask has bits beyond the last column");
    if (gen_out) *gen_out = g;
    return 0;
}

extern "C" {

int bnn_abi_version(void) { return BNN_ABI_VERSION; }
const char* bnn_last_error(void) { return g_err.c_str(); }

int bnn_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(BNN_ERR_NO_DEVICE, hipGetErrorString(e));
    return n;
}

#ifndef BNN_BUILD_FLAGS
#define BNN_BUILD_FLAGS ""
#endif
const char* bnn_build_flags(void) { return BNN_BUILD_FLAGS; }

int bnn_param_count(const bnn_arch* arch) {
    GenArch g;
    int rc = check_arch(arch, &g);
    return rc ? rc : g.d;
}

static bool upload(const void* host, size_t bytes, void** dev) {
    return hipMalloc(dev, bytes) == hipSuccess && hipMemcpy(*dev, host, bytes, hipMemcpyHostToDevice) == hipSuccess;
}

int bnn_plan_create(const bnn_arch* arch, bnn_plan** out) {
    GenArch g;
    int rc = check_arch(arch, &g);
    if (rc) return rc;
    if (!out) return fail(BNN_ERR_INVALID, "out is NULL");
    bnn_plan* pl = new bnn_plan();
    pl->arch = *arch;
    pl->megno = arch->fix_megno != 0;
    pl->v50net = is_v50net(arch);
    pl->gen = g;
    pl->d = g.d;
    if (pl->v50net) {
        pl->tab[0] = build_tables(arch->zero_mask, false, pl->megno);
        pl->tab[1] = build_tables(arch->zero_mask, true, pl->megno);
        if (pl->d != layout_of(pl->megno).D) { delete pl; return fail(BNN_ERR_INVALID, "internal: the two engines disagree on the parameter count"); }
    }
    if (pl->v50net && !pl->megno) {   // the forms of bnn_fwd_v50spec.hip: noisy under any mask, quiet under the pretrained one
        const char* why = "";
        pl->emb[1] = gen_build_spec(F, H, L, 1, 1, false, 1, 0, 1, &pl->emb_gen[1], &why) == 0;
        pl->emb[0] = arch->zero_mask == V50_ZERO_MASK && gen_build_spec(F, H, L, 1, 1, false, 1, V50_ZERO_MASK, 1, &pl->emb_gen[0], &why) == 0;
    }
    if (hipGetDevice(&pl->device) != hipSuccess) {
        delete pl;
        return fail(BNN_ERR_NO_DEVICE, "no HIP device");
    }
    std::vector<float> rcp(RCP_N);
    for (int i = 0; i < RCP_N; ++i) rcp[i] = 1.0f / (float)(i + 1);
    bool ok = upload(rcp.data(), RCP_N * sizeof(float), (void**)&pl->d_rcp) && upload(&pl->gen, sizeof(GenArch), (void**)&pl->d_gen);
    if (ok && pl->v50net)
        ok = upload(pl->tab[0].f2.data(), pl->tab[0].f2.size() * sizeof(int16_t), (void**)&pl->d_f2) &&
             upload(pl->tab[0].f4.data(), pl->tab[0].f4.size() * sizeof(int16_t), (void**)&pl->d_f4) &&
             upload(pl->tab[1].f4.data(), pl->tab[1].f4.size() * sizeof(int16_t), (void**)&pl->d_f4n);
    if (!ok) {
        bnn_plan_destroy(pl);
        return fail(BNN_ERR_HIP, "plan table upload failed");
    }
    *out = pl;
    return 0;
}

int bnn_plan_destroy(bnn_plan* pl) {
    if (!pl) return 0;
    if (pl->d_f2) (void)hipFree(pl->d_f2);
    if (pl->d_f4) (void)hipFree(pl->d_f4);
    if (pl->d_f4n) (void)hipFree(pl->d_f4n);
    if (pl->d_rcp) (void)hipFree(pl->d_rcp);
    if (pl->d_gen) (void)hipFree(pl->d_gen);
    for (int i = 0; i < 2; ++i)
        if (pl->spec_mod[i]) (void)hipModuleUnload(pl->spec_mod[i]);
    delete pl;
    return 0;
}

// the descriptor of one specialised form: the quiet kq-major forms drop the plan's masked input columns from layer 0
static int spec_arch(const bnn_arch* a, int32_t w8, int32_t noisy, int32_t flags, GenArch* g) {
    if (flags & ~(BNN_SPEC_POOL_REGS | BNN_SPEC_BLOCK_MAJOR | BNN_SPEC_RESIDENT)) return fail(BNN_ERR_INVALID, "unknown specialisation flag");
    if ((flags & BNN_SPEC_RESIDENT) && (flags & BNN_SPEC_BLOCK_MAJOR)) return fail(BNN_ERR_INVALID, "BNN_SPEC_RESIDENT belongs to the input-quad-major forms");
    if (noisy != 0 && noisy != 1) return fail(BNN_ERR_INVALID, "noisy must be 0 or 1");
    const char* why = "";
    const uint64_t drop = (noisy || (flags & BNN_SPEC_BLOCK_MAJOR)) ? 0 : a->zero_mask;
    if (gen_build_spec(a->n_features, a->hidden, a->latent, a->depth_in, a->depth_out, a->fix_megno != 0, w8, drop, (flags & BNN_SPEC_POOL_REGS) ? 1 : 0, g, &why))
        return fail(BNN_ERR_UNSUPPORTED, why);
    return 0;
}

int bnn_spec_embedded_source(char* buf, size_t cap) {
    const int n = gen_spec_embedded_source(buf, cap);
    return n < 0 ? fail(BNN_ERR_UNSUPPORTED, "internal: the pretrained network's specialised descriptor") : n;
}

int bnn_spec_source(const bnn_arch* arch, int32_t w8, int32_t noisy, int32_t flags, char* buf, size_t cap) {
    int rc = check_arch(arch);
    if (rc) return rc;
    GenArch g;
    rc = spec_arch(arch, w8, noisy, flags, &g);
    if (rc) return rc;
    if (flags & BNN_SPEC_RESIDENT) {   // a few dozen registers at most: beyond that nothing is left for the activations
        int nw = 0;
        for (int l = 0; l < g.n_feat; ++l) nw += g.layer[l].nkq * g.layer[l].nblk;
        if (nw > 112) return fail(BNN_ERR_UNSUPPORTED, "feature_nn's weight registers do not fit next to the activations (more than 112)");
    }
    return gen_spec_source(g, noisy, (flags & BNN_SPEC_POOL_REGS) ? 1 : 0, (flags & BNN_SPEC_BLOCK_MAJOR) ? 1 : 0,
                           g.in_live ? arch->zero_mask : 0, buf, cap, (flags & BNN_SPEC_RESIDENT) ? 1 : 0);
}

int bnn_plan_attach_spec(bnn_plan* pl, int32_t noisy, int32_t w8, int32_t flags, const void* image, size_t bytes) {
    if (!pl || !image || !bytes) return fail(BNN_ERR_INVALID, "plan/image is NULL");
    GenArch g;
    int rc = spec_arch(&pl->arch, w8, noisy, flags, &g);
    if (rc) return rc;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != pl->device) return fail(BNN_ERR_INVALID, "the plan lives on another device than the current one");
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    hipError_t e = hipModuleLoadData(&mod, image);
    if (e != hipSuccess) return fail(BNN_ERR_HIP, std::string("hipModuleLoadData: ") + hipGetErrorString(e));
    e = hipModuleGetFunction(&fn, mod, "bnn_spec_forward");
    if (e != hipSuccess) {
        (void)hipModuleUnload(mod);
        return fail(BNN_ERR_HIP, std::string("the code object has no kernel bnn_spec_forward: ") + hipGetErrorString(e));
    }
    if (pl->spec_mod[noisy]) {   // launches of the form being replaced may still be queued on any stream: drain the device before its
        const hipError_t es = hipDeviceSynchronize();   // code leaves (a setup call, like plan creation: the only kind that synchronises)
        if (es != hipSuccess) {
            (void)hipModuleUnload(mod);   // (the freshly loaded module must not outlive a failed attach)
            return fail(BNN_ERR_HIP, std::string("hipDeviceSynchronize before replacing a specialised form: ") + hipGetErrorString(es));
        }
        (void)hipModuleUnload(pl->spec_mod[noisy]);
    }
    pl->spec_mod[noisy] = mod;
    pl->spec_fn[noisy] = fn;
    pl->spec_gen[noisy] = g;
    return 0;
}

int bnn_plan_spec_attached(const bnn_plan* pl, int32_t noisy) {
    if (!pl || (noisy != 0 && noisy != 1)) return fail(BNN_ERR_INVALID, "plan is NULL / noisy must be 0 or 1");
    return pl->spec_fn[noisy] ? 1 : 0;
}

size_t bnn_gen_params_bytes(void) { return sizeof(GenParams); }

size_t bnn_nonfinite_record_bytes(int64_t B) { return sizeof(int32_t) * (size_t)(4 + (B > 0 ? B : 0)); }

int bnn_nonfinite_scan_f32(const bnn_plan* pl, const float* x, int64_t B, int32_t T, void* record, void* stream) {
    if (!pl) return fail(BNN_ERR_INVALID, "plan is NULL");
    if (B < 0 || T < 1) return fail(BNN_ERR_INVALID, "bad B/T");
    if (!record) return fail(BNN_ERR_INVALID, "record is NULL");
    if (B >= (1LL << 30)) return fail(BNN_ERR_RANGE, "the scan record indexes at most 2^30 systems per call: shard the batch");
    if (B > 0 && !x) return fail(BNN_ERR_INVALID, "x is NULL");
    // a masked column holds NaN after `x - mask` whichever non-finite value it had; fix_megno zeroes the MEGNO column whatever the
    // mask says (:488-491) and summarises the raw one (:480-484): non-finite there is NaN in the summary either way
    const uint64_t mask = pl->arch.zero_mask | (pl->megno ? (1ull << MEGNO_COL) : 0ull);
    hipError_t e = launch_nonfinite_scan(x, B, (int64_t)T * pl->arch.n_features, pl->arch.n_features, mask, static_cast<int32_t*>(record), (hipStream_t)stream);
    if (e != hipSuccess) return fail(BNN_ERR_HIP, std::string("non-finite scan: ") + hipGetErrorString(e));
    return 0;
}

// natural order of the generic engine: the live inputs ascending (layer 0 drops the masked columns unless `noisy`)
static int natural_order(const GenArch& g, uint64_t zero_mask, int layer, int noisy, int32_t* host_order, int cap) {
    if (layer < 0 || layer >= g.n_feat + g.n_reg) return fail(BNN_ERR_INVALID, "layer index beyond the network's Linear modules");
    int n = 0;
    for (int k = 0; k < g.layer[layer].K; ++k) {
        if (layer == 0 && !noisy && k < 64 && ((zero_mask >> k) & 1ull)) continue;
        if (host_order && n < cap) host_order[n] = k;
        ++n;
    }
    return n;
}

int bnn_plan_layer_order(const bnn_plan* pl, int layer, int noisy, int32_t* host_order, int cap) {
    if (!pl) return fail(BNN_ERR_INVALID, "plan is NULL");
    if (!pl->v50net) return natural_order(pl->gen, pl->arch.zero_mask, layer, noisy, host_order, cap);
    if (layer < 0 || layer > 5) return fail(BNN_ERR_INVALID, "bad plan/layer");
    const std::vector<int32_t>& o = pl->tab[noisy ? 1 : 0].order[layer];
    if (host_order)
        for (int i = 0; i < (int)o.size() && i < cap; ++i) host_order[i] = o[i];
    return (int)o.size();
}

int bnn_layer_order(const bnn_arch* arch, int layer, int noisy, int32_t* host_order, int cap) {
    GenArch g;
    int rc = check_arch(arch, &g);
    if (rc) return rc;
    if (!is_v50net(arch)) return natural_order(g, arch->zero_mask, layer, noisy, host_order, cap);
    if (layer < 0 || layer > 5) return fail(BNN_ERR_INVALID, "layer must be 0..5");
    Tables t = build_tables(arch->zero_mask, noisy != 0, arch->fix_megno != 0);
    const std::vector<int32_t>& o = t.order[layer];
    if (host_order)
        for (int i = 0; i < (int)o.size() && i < cap; ++i) host_order[i] = o[i];
    return (int)o.size();
}

int bnn_fragment_table(const bnn_arch* arch, int noisy, int which, int16_t* host_table, int cap) {
    int rc = check_arch(arch);
    if (rc) return rc;
    if (!is_v50net(arch)) return fail(BNN_ERR_UNSUPPORTED, "fragment tables belong to the pretrained network's kernels; the generic engine builds its LDS image in the kernel prologue");
    if (which != 1 && which != 2) return fail(BNN_ERR_INVALID, "which must be 1 (feature_nn images) or 2 (regress_nn fragments)");
    Tables t = build_tables(arch->zero_mask, noisy != 0, arch->fix_megno != 0);
    const std::vector<int16_t>& v = which == 1 ? t.f4 : t.f2;
    if (host_table)
        for (int i = 0; i < (int)v.size() && i < cap; ++i) host_table[i] = v[i];
    return (int)v.size();
}

constexpr int64_t TSPLIT_MAX_BLOCKS = 256;   // one tile-split workgroup per CU

static int pick_spc(const bnn_grid* g, int64_t csz, bool xcd_order) {
    if (g->systems_per_block > 0) return g->systems_per_block;
    // The per-workgroup prologue (flat vector -> weight registers, regress_nn fragments) is amortised over the block: prefer big
    // blocks (512 systems: +1 % over 128 at configs[1]) as long as the grid still fills 256 CUs x 2 several times over.  Under the
    // XCD work order the block is also the L2's working set across draws: 256-system blocks (4.2 MB) reach HBM 79 GB per configs[2]
    // launch against 168 GB for 512 (8.4 MB) -- but HBM is at 3.5 % of its roof either way, the kernel time is the same within
    // 0.3 %, and 256 costs 1.2 % more cycles (twice the prologues; profiles/r03_spb_traffic.txt): 512 stays.
    (void)xcd_order;
    for (int spc : {512, 256, 128}) {
        int64_t nsub = (csz + spc - 1) / spc;
        if (nsub * (int64_t)g->J >= 4096) return spc;
    }
    return 64;
}

static GenMerge gen_merge_consts(int na, int nb) {   // oracle/bnn_oracle.c merge_consts, constant for constant
    GenMerge m{2, 0.0f, 0.0f};
    if (nb == 0) m.mode = 2;
    else if (na == 0) m.mode = 3;
    else if (na == nb) { m.mode = 0; m.w1 = (float)na * 0.5f; }
    else {
        m.mode = 1;
        m.w1 = (float)((double)nb / (double)(na + nb));
        m.w2 = (float)((double)na * (double)nb / (double)(na + nb));
    }
    return m;
}

static int launch_forward(const bnn_plan* pl, const bnn_grid* g, FwdParams& p, bool fused, bool noisy, void* stream, int lowp = 0) {
    if (!pl || !g) return fail(BNN_ERR_INVALID, "plan/grid is NULL");
    if (g->B < 0 || g->J < 0 || g->nchunks < 1) return fail(BNN_ERR_INVALID, "negative size");
    if (g->J % g->nchunks) return fail(BNN_ERR_INVALID, "J must be a multiple of nchunks");
    if (g->T < 2 || (g->T + 3) / 4 > RCP_N) return fail(BNN_ERR_UNSUPPORTED, "T must be in [2, 16384] (torch.std of a single timestep is NaN)");
    if (g->systems_per_block < 0 || (g->systems_per_block % 64)) return fail(BNN_ERR_INVALID, "systems_per_block must be a multiple of 64");
    // the pretrained network's kernels take whole tiles of 4 timesteps and at least two of them; everything else is the generic engine's
    if (g->engine < 0 || g->engine > 2) return fail(BNN_ERR_INVALID, "grid.engine must be 0 (choose), 1 (generic) or 2 (specialised)");
    const bool generic = !pl->v50net || (g->T % 4) != 0 || g->T < 8 || g->engine != 0;
    // the network's own compiled form of the generic engine, when one is attached (engine 2 insists on it)
    const int nz = noisy ? 1 : 0;
    const bool spec_mod = generic && g->engine != 1 && pl->spec_fn[nz] != nullptr;
    const bool spec_emb = generic && g->engine != 1 && !spec_mod && pl->emb[nz];   // (an attached form wins over the embedded one)
    const bool spec = spec_mod || spec_emb;
    if (g->engine == 2 && !spec) return fail(BNN_ERR_UNSUPPORTED, "grid.engine = 2: no specialised form is attached to this plan (bnn_plan_attach_spec)");
    if (generic && lowp) return fail(BNN_ERR_UNSUPPORTED, "the reduced-precision kernels are built for the pretrained network at T % 4 == 0 only");
    if (generic && fused) return fail(BNN_ERR_UNSUPPORTED, "the in-prologue draw (W_workspace = NULL) exists for the pretrained network at T % 4 == 0 only: pass a [J, d] workspace");
    if (g->B == 0 || g->J == 0) return 0;
    if (!p.x || !(p.out || p.sink || p.latents)) return fail(BNN_ERR_INVALID, "x/out is NULL");
    if (p.draw_id0 % g->nchunks) return fail(BNN_ERR_INVALID, "draw_id0 must be a multiple of nchunks");
    const int NF = pl->arch.n_features;
    p.B = g->B; p.T = g->T; p.ntiles = (g->T + 3) / 4; p.J = g->J; p.nch = g->nchunks;
    p.cB = g->chunk_B > 0 ? g->chunk_B : g->B;
    p.coff = g->chunk_B > 0 ? g->chunk_off : 0;
    if (p.coff < 0 || p.coff + g->B > p.cB) return fail(BNN_ERR_INVALID, "chunk_off / chunk_B: the shard [chunk_off, chunk_off + B) must lie inside the batch");
    p.csz = (p.cB + g->nchunks - 1) / g->nchunks;
    p.xcd_order = (g->nchunks > 1 || (double)g->B * g->T * NF * sizeof(float) > 256.0 * 1024 * 1024) ? 1 : 0;
    const int64_t cseg = p.csz < g->B ? p.csz : g->B;   // the longest stretch of one chunk inside this call's rows
    p.spc = pick_spc(g, cseg, p.xcd_order != 0);
    // Small grids (the evaluation scripts' per-chunk calls: 15 .. 3 000 rows under one draw) take the TILE-SPLIT form of the pretrained
    // network's kernel: 16 systems per workgroup, the four waves sharing a batch's tiles (bnn_forward.hip.h, TSPLIT) -- same bits, a
    // quarter of the time per batch -- as long as every workgroup is resident at once (one per CU: 93 KB of LDS), ...  An explicit
    // systems_per_block keeps the plain form (that is also how the tests compare the two).
    // ... or a draw never covers more than 16 systems (the 5-planet loop as ONE call: 15-row chunks under thousands of draws -- in the
    // plain form three of a workgroup's four waves would have no system at all).
    const bool tsplit = !generic && !lowp && !noisy && !p.sink && !pl->megno && pl->tab[0].kin4 == 31 && g->systems_per_block == 0 &&
                        (((cseg + 15) / 16) * (int64_t)g->J <= TSPLIT_MAX_BLOCKS || cseg <= 16);
    if (tsplit) p.spc = 16;
    p.row_id0 = p.draw_id0 / g->nchunks;
    p.tab_f2 = pl->d_f2; p.tab_wr = noisy ? pl->d_f4n : pl->d_f4; p.rcp_tab = pl->d_rcp;
    p.zero_mask = pl->arch.zero_mask;
    p.std_lo = pl->arch.lowest_std; p.std_span = (float)(6.0 - (double)pl->arch.lowest_std);
    const int64_t nsub = (cseg + p.spc - 1) / p.spc;
    const int64_t nblk = nsub * g->J;
    if (nblk > 0x7fffffffLL) return fail(BNN_ERR_RANGE, "grid too large; split the draws");
    static_assert(Lay<true>::NF2 * 64 <= FLAT_LDS, "regress_nn fragments overwrite the flat vector in place");
    hipStream_t st = (hipStream_t)stream;
    auto launch = [&]() -> int {
        hipError_t e;
        if (generic) {
            GenParams P{};
            P.f = p;
            P.g = pl->d_gen;
            int cnt[4];
            for (int q = 0; q < 4; ++q) cnt[q] = q < g->T ? (g->T - q + 3) / 4 : 0;   // timesteps t = q, q + 4, ... below T
            P.m01 = gen_merge_consts(cnt[0], cnt[1]);
            P.m23 = gen_merge_consts(cnt[2], cnt[3]);
            P.m0123 = gen_merge_consts(cnt[0] + cnt[1], cnt[2] + cnt[3]);
            P.noisy = noisy ? 1 : 0;
            if (spec_emb) {
                e = launch_fwd_v50spec(noisy, (unsigned)nblk, st, P);
                if (e != hipSuccess) return fail(BNN_ERR_HIP, std::string("embedded specialised forward kernel launch: ") + hipGetErrorString(e));
                return 0;
            }
            if (spec_mod) {
                size_t psz = sizeof(P);
                void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &P, HIP_LAUNCH_PARAM_BUFFER_SIZE, &psz, HIP_LAUNCH_PARAM_END};
                e = hipModuleLaunchKernel(pl->spec_fn[noisy ? 1 : 0], (unsigned)nblk, 1, 1, 64u * pl->spec_gen[noisy ? 1 : 0].nwaves, 1, 1, 0, st, nullptr, cfg);
                if (e != hipSuccess) return fail(BNN_ERR_HIP, std::string("specialised forward kernel launch: ") + hipGetErrorString(e));
                return 0;
            }
            e = launch_fwd_generic(pl->gen, (unsigned)nblk, st, P);
            if (e != hipSuccess) return fail(BNN_ERR_HIP, std::string("generic forward kernel launch: ") + hipGetErrorString(e));
            return 0;
        }
        const bool k31 = pl->tab[0].kin4 == 31;
        if (pl->megno && (lowp || p.sink)) return fail(BNN_ERR_UNSUPPORTED, "fix_megno: the reduced-precision and fused-statistics forms are not built");
        if (pl->megno) e = launch_fwd_megno(k31, fused, noisy, (unsigned)nblk, st, p);
        else if (lowp) e = launch_fwd_lowp(lowp, (unsigned)nblk, st, p);
        else if (p.sink) e = launch_fwd_stats(k31, (unsigned)nblk, st, p);
        else if (noisy) e = launch_fwd_noisy((unsigned)nblk, st, p);
        else if (tsplit) e = launch_fwd_small(fused, (unsigned)nblk, st, p);
        else if (k31) e = launch_fwd_k31(fused, (unsigned)nblk, st, p);
        else e = launch_fwd_k41(fused, (unsigned)nblk, st, p);
        if (e != hipSuccess) return fail(BNN_ERR_HIP, std::string("forward kernel launch: ") + hipGetErrorString(e));
        return 0;
    };
    int rc = launch();
    if (rc || !g->nonfinite) return rc;
    // the systems the scan listed (non-finite inputs) are re-evaluated the reference's way and overwrite what the kernel above wrote for
    // them (bnn_nonfinite.hip); with an empty list the launch leaves at once
    NfxParams q{};
    q.f = p;
    q.g = pl->d_gen;
    q.rec = static_cast<const int32_t*>(g->nonfinite);
    q.noisy = noisy ? 1 : 0;
    q.shortcut = (p.summary || p.latents) ? 0 : 1;
    hipError_t e = launch_nonfinite_fixup(pl->gen, q, st);
    if (e != hipSuccess) return fail(BNN_ERR_HIP, std::string("non-finite fix-up kernel launch: ") + hipGetErrorString(e));
    return 0;
}

int bnn_forward_f32(const bnn_plan* plan, const bnn_grid* grid, const float* x, const float* W, const float* eps, const float* eps_in,
                    const float* eps_sum, uint64_t philox_seed, int64_t draw_id0, int64_t system_id0, float* out, float* pre_clamp,
                    float* summary, void* stream) {
    if (grid && (grid->B == 0 || grid->J == 0)) return 0;
    if (!W) return fail(BNN_ERR_INVALID, "W is NULL");
    if ((eps_in == nullptr) != (eps_sum == nullptr)) return fail(BNN_ERR_INVALID, "eps_in and eps_sum must both be given or both be NULL");
    if (eps_in && !eps) return fail(BNN_ERR_INVALID, "explicit noise is all-or-nothing: eps_in/eps_sum need eps as well");
    const bool noisy = eps_in != nullptr || (grid && grid->noisy);
    if (noisy && !eps_in && eps) return fail(BNN_ERR_INVALID, "explicit noise is all-or-nothing: grid.noisy with eps but no eps_in/eps_sum");
    FwdParams p{};
    p.x = x; p.W = W; p.eps = eps; p.eps_in = eps_in; p.eps_sum = eps_sum;
    p.seed = philox_seed; p.draw_id0 = draw_id0; p.sys_id0 = system_id0;
    p.out = out; p.pre_clamp = pre_clamp; p.summary = summary;
    return launch_forward(plan, grid, p, false, noisy, stream);
}

int bnn_feature_nn_f32(const bnn_plan* plan, const bnn_grid* grid, const float* x, const float* W, const float* eps_in, uint64_t philox_seed,
                       int64_t draw_id0, int64_t system_id0, float* latents, void* stream) {
    if (!plan || !grid) return fail(BNN_ERR_INVALID, "plan/grid is NULL");
    if (grid->B == 0 || grid->J == 0) return 0;
    if (!W || !latents) return fail(BNN_ERR_INVALID, "W/latents is NULL");
    // the generic engine with its latents output switched on and no (mu, std) output: the pool and the tail run on in-kernel normals and
    // their results are dropped (the latents do not depend on them)
    bnn_grid g = *grid;
    g.engine = 1;
    FwdParams p{};
    p.x = x; p.W = W; p.eps_in = eps_in;
    p.seed = philox_seed; p.draw_id0 = draw_id0; p.sys_id0 = system_id0;
    p.latents = latents;
    return launch_forward(plan, &g, p, false, eps_in != nullptr || grid->noisy, stream);
}

int bnn_forward_lowp_f32(const bnn_plan* plan, const bnn_grid* grid, const float* x, const float* W, const float* eps, uint64_t philox_seed,
                         int64_t draw_id0, int64_t system_id0, int32_t precision, float* out, float* pre_clamp, float* summary, void* stream) {
    if (!plan || !grid) return fail(BNN_ERR_INVALID, "plan/grid is NULL");
    if (precision < BNN_PREC_BF16 || precision > BNN_PREC_F16X3) return fail(BNN_ERR_INVALID, "precision must be one of BNN_PREC_BF16 .. BNN_PREC_F16X3");
    if (!plan->v50net || plan->tab[0].kin4 != 31) return fail(BNN_ERR_UNSUPPORTED, "the reduced-precision kernels are built for the pretrained network with the v50 column mask only");
    if (grid->noisy) return fail(BNN_ERR_UNSUPPORTED, "the reduced-precision kernels have no noisy form");
    if (grid->B == 0 || grid->J == 0) return 0;
    if (!W) return fail(BNN_ERR_INVALID, "W is NULL");
    FwdParams p{};
    p.x = x; p.W = W; p.eps = eps;
    p.seed = philox_seed; p.draw_id0 = draw_id0; p.sys_id0 = system_id0;
    p.out = out; p.pre_clamp = pre_clamp; p.summary = summary;
    return launch_forward(plan, grid, p, false, false, stream, precision);
}

int bnn_multiswag_f32(const bnn_plan* plan, const bnn_grid* grid, const float* x, const float* w_avg, const float* w2_avg,
                      const float* pre_D, int32_t S, int32_t K, const int32_t* seed_idx, const float* z1, const float* z2,
                      const float* eps, float scale, uint64_t philox_seed, int64_t draw_id0, int64_t system_id0,
                      float* W_workspace, float* out, float* pre_clamp, float* summary, void* stream) {
    if (!plan || !grid) return fail(BNN_ERR_INVALID, "plan/grid is NULL");
    if (grid->B == 0 || grid->J == 0) return 0;  // nothing to do (empty tensors have NULL data pointers)
    if (!w_avg || !w2_avg || !pre_D || !seed_idx) return fail(BNN_ERR_INVALID, "NULL ensemble argument");
    if (W_workspace) {  // sample every draw once, then the forward kernel reads the materialised vectors
        if (!grid) return fail(BNN_ERR_INVALID, "plan/grid is NULL");
        int rc = bnn_swag_draw_f32(plan, w_avg, w2_avg, pre_D, S, K, seed_idx, grid->J, z1, z2, scale, philox_seed, draw_id0,
                                   W_workspace, stream);
        if (rc) return rc;
        return bnn_forward_f32(plan, grid, x, W_workspace, eps, nullptr, nullptr, philox_seed, draw_id0, system_id0, out, pre_clamp,
                               summary, stream);
    }
    if ((z1 == nullptr) != (z2 == nullptr)) return fail(BNN_ERR_INVALID, "z1 and z2 must both be given or both be NULL");
    if (S < 1) return fail(BNN_ERR_INVALID, "bad S");
    FwdParams p{};
    int rc = draw_consts(K, scale, &p.c1, &p.c2, MAXK);
    if (rc) return rc;
    p.scale = scale; p.K = K; p.S = S;
    p.x = x; p.w_avg = w_avg; p.w2_avg = w2_avg; p.pre_D = pre_D; p.seed_idx = seed_idx; p.z1 = z1; p.z2 = z2; p.eps = eps;
    p.seed = philox_seed; p.draw_id0 = draw_id0; p.sys_id0 = system_id0;
    p.out = out; p.pre_clamp = pre_clamp; p.summary = summary;
    return launch_forward(plan, grid, p, true, false, stream);
}

int bnn_multiswag_stats_f32(const bnn_plan* plan, const bnn_grid* grid, const float* x, const float* w_avg, const float* w2_avg,
                            const float* pre_D, int32_t S, int32_t K, const int32_t* seed_idx, const float* z1, const float* z2,
                            const float* eps, float scale, uint64_t philox_seed, int64_t draw_id0, int64_t system_id0,
                            float* W_workspace, const bnn_stats* st, float* t_out, void* stream) {
    if (!plan || !grid) return fail(BNN_ERR_INVALID, "plan/grid is NULL");
    StatsParams sp;
    int rc = stats_params(st, &sp);
    if (rc) return rc;
    if (grid->B == 0 || grid->J == 0) return 0;
    if (!w_avg || !w2_avg || !pre_D || !seed_idx || !W_workspace || !t_out) return fail(BNN_ERR_INVALID, "NULL argument (the statistics form needs a draw workspace)");
    rc = bnn_swag_draw_f32(plan, w_avg, w2_avg, pre_D, S, K, seed_idx, grid->J, z1, z2, scale, philox_seed, draw_id0, W_workspace, stream);
    if (rc) return rc;
    FwdParams p{};
    p.x = x; p.W = W_workspace; p.eps = eps;
    p.seed = philox_seed; p.draw_id0 = draw_id0; p.sys_id0 = system_id0;
    p.sink = t_out; p.st = sp;
    return launch_forward(plan, grid, p, false, false, stream);
}

// ---- slab drivers: the whole (systems x draws) grid reduced on the fly, nothing of size J x B ever in memory --------------------
// Both evaluate the draws in slabs of `draws_per_launch` (a multiple of nchunks) through caller-provided scratch, on `stream`,
// without synchronising: moments -> [B,4] float64; bands -> the quantile sketch (hist, mom) of the post-epilogue times.
static int slab_args(const bnn_plan* plan, const bnn_grid* grid, int32_t draws_per_launch) {
    if (!plan || !grid) return fail(BNN_ERR_INVALID, "plan/grid is NULL");
    if (grid->nchunks < 1 || grid->J < 0 || grid->J % grid->nchunks) return fail(BNN_ERR_INVALID, "J must be a multiple of nchunks");
    if (draws_per_launch < grid->nchunks || draws_per_launch % grid->nchunks) return fail(BNN_ERR_INVALID, "draws_per_launch must be a positive multiple of nchunks");
    return 0;
}

int bnn_multiswag_moments_f64(const bnn_plan* plan, const bnn_grid* grid, const float* x, const float* w_avg, const float* w2_avg,
                              const float* pre_D, int32_t S, int32_t K, const int32_t* seed_idx, float scale, uint64_t philox_seed,
                              int64_t draw_id0, int64_t system_id0, int32_t draws_per_launch, float* W_workspace, float* out_workspace,
                              double* moments, void* stream) {
    int rc = slab_args(plan, grid, draws_per_launch);
    if (rc) return rc;
    if (grid->B == 0) return 0;
    if (!moments || !out_workspace || !W_workspace) return fail(BNN_ERR_INVALID, "NULL workspace / moments");
    // (the first slab overwrites, the others accumulate; no draws at all = the moments kernel over zero rows writes the zeros: NO memset node
    //  anywhere in this library -- a captured hipMemsetAsync replays with a garbage fill value, see bnn_nonfinite.hip)
    if (grid->J == 0) return bnn_moments_f64(out_workspace, 0, grid->B, moments, 0, stream);
    for (int32_t j0 = 0; j0 < grid->J; j0 += draws_per_launch) {
        bnn_grid g = *grid;
        g.J = grid->J - j0 < draws_per_launch ? grid->J - j0 : draws_per_launch;
        rc = bnn_multiswag_f32(plan, &g, x, w_avg, w2_avg, pre_D, S, K, seed_idx + j0, nullptr, nullptr, nullptr, scale, philox_seed,
                               draw_id0 + j0, system_id0, W_workspace, out_workspace, nullptr, nullptr, stream);
        if (rc) return rc;
        rc = bnn_moments_f64(out_workspace, g.J / g.nchunks, g.B, moments, j0 > 0 ? 1 : 0, stream);
        if (rc) return rc;
    }
    return 0;
}

int bnn_multiswag_bands_f32(const bnn_plan* plan, const bnn_grid* grid, const float* x, const float* w_avg, const float* w2_avg,
                            const float* pre_D, int32_t S, int32_t K, const int32_t* seed_idx, float scale, uint64_t philox_seed,
                            int64_t draw_id0, int64_t system_id0, int32_t draws_per_launch, float* W_workspace, float* t_workspace,
                            const bnn_stats* st, int32_t group, const bnn_sketch* sk, uint32_t* hist, double* mom, void* stream) {
    int rc = slab_args(plan, grid, draws_per_launch);
    if (rc) return rc;
    if (grid->B == 0) return 0;
    if (!t_workspace || !W_workspace || !hist || !mom) return fail(BNN_ERR_INVALID, "NULL workspace / sketch");
    for (int32_t j0 = 0; j0 < grid->J; j0 += draws_per_launch) {
        bnn_grid g = *grid;
        g.J = grid->J - j0 < draws_per_launch ? grid->J - j0 : draws_per_launch;
        rc = bnn_multiswag_stats_f32(plan, &g, x, w_avg, w2_avg, pre_D, S, K, seed_idx + j0, nullptr, nullptr, nullptr, scale,
                                     philox_seed, draw_id0 + j0, system_id0, W_workspace, st, t_workspace, stream);
        if (rc) return rc;
        rc = bnn_sketch_update_u32(t_workspace, g.J / g.nchunks, g.B, group, sk, hist, mom, stream);
        if (rc) return rc;
    }
    return 0;
}

}  // extern "C"

