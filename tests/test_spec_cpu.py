"""Run-time specialisation of the generic forward engine, the parts that need no GPU (specialize.py, bnn_spec_source): the generated
source, the compile + cache round trip (hipcc cross-compiles gfx950 here), the variant search's resource report."""
import re

import pytest

hipcc = pytest.importorskip("bnn_chaos_model_amd.csrc.build").hipcc


@pytest.fixture(scope="module")
def N():
    from bnn_chaos_model_amd import _native
    return _native


def arch(N, H=30, L=12, din=0, dout=0, NF=41, mask=None):
    from bnn_chaos_model_amd.ops import V50_ZERO_MASK
    return N.BnnArch(NF, H, L, 0, V50_ZERO_MASK if mask is None else mask, 0.5, 0.0, din, dout)


def test_source_is_deterministic_and_carries_the_shapes(N):
    a = arch(N, 40, 20, 1, 1)
    src = N.spec_source(a, False, True, N.SPEC_POOL_REGS)
    assert src == N.spec_source(a, False, True, N.SPEC_POOL_REGS)
    assert 'extern "C" __global__ __launch_bounds__(512, 1) void bnn_spec_forward' in src
    assert "generic_body<11, 12, true, bnn::SpecArch, 0>" in src and "pool_lq = 5" in src
    # the quiet form multiplies the 31 unmasked columns only (8 input quads; padding repeats a live column), in ascending order
    assert "in_q = 8" in src
    live = [int(v) for v in re.search(r"constexpr int t\[\] = \{([^}]*)\}", src).group(1).split(",") if v.strip()]
    assert live[:31] == [c for c in range(41) if c not in (1, 2, 3, 4, 5, 6, 7, 38, 39, 40)] and live[31] == live[0] and len(live) == 32
    assert "{31, 40, 8, 3, 2," in src                      # layer 0: K = 31 live inputs, 8 quads, 3 blocks, 2 groups in the last
    noisy = N.spec_source(a, True, True, 0)                # the noisy form keeps every column (masked ones carry noise) and the LDS pool
    assert "in_q" not in noisy and "pool_lq = 0" in noisy and "SpecArch, 1>" in noisy and "{41, 40, 11, 3, 2," in noisy
    assert "in_q" not in N.spec_source(arch(N, 40, 20, 1, 1, mask=0), False, True, 0)          # nothing masked: nothing to drop
    assert "kq_major = false" in N.spec_source(a, False, False, N.SPEC_BLOCK_MAJOR)
    four = N.spec_source(a, False, False, 0)
    assert "__launch_bounds__(256, 1)" in four and "12, false," in four


def test_limits_are_errors(N):
    with pytest.raises(N.NativeError):
        N.spec_source(arch(N), False, True, 64)                                    # unknown flag
    with pytest.raises(N.NativeError):
        N.spec_source(arch(N, 128, 32, 1, 1), False, True, 0)                      # eight waves do not fit next to that image
    with pytest.raises(N.NativeError):
        N.spec_source(arch(N, 200, 20, 1, 1), False, None, 0)                      # the engine's own width limit
    assert "generic_body<21, " in N.spec_source(arch(N, 40, 20, 1, 1, NF=82), False, None, 0)


def test_abi_argument_errors(N):
    import ctypes as C
    L = N.lib()
    a = arch(N)
    assert L.bnn_spec_source(None, -1, 0, 0, None, 0) < 0                                    # NULL arch
    assert L.bnn_spec_source(C.byref(a), -1, 2, 0, None, 0) < 0                              # noisy is 0 / 1
    n = L.bnn_spec_source(C.byref(a), -1, 0, N.SPEC_POOL_REGS, None, 0)
    small = C.create_string_buffer(16)
    assert L.bnn_spec_source(C.byref(a), -1, 0, N.SPEC_POOL_REGS, small, 16) == n and len(small.value) == 15   # truncated, terminated, full length returned
    assert L.bnn_plan_attach_spec(None, 0, -1, 0, b"x", 1) < 0 and L.bnn_plan_spec_attached(None, 0) < 0
    bad = N.BnnArch(41, 40, 20, 2, 0, 0.5, 0.0, 1, 1)                                         # fix_megno must be 0 / 1
    assert L.bnn_spec_source(C.byref(bad), -1, 0, 0, None, 0) < 0
    masked_all = N.BnnArch(41, 40, 20, 0, (1 << 41) - 1, 0.5, 0.0, 1, 1)                      # every column masked: nothing to multiply
    assert L.bnn_spec_source(C.byref(masked_all), -1, 0, 0, None, 0) < 0 and b"masked" in L.bnn_last_error()


def test_compile_cache_and_resource_report(N, tmp_path, monkeypatch):
    from bnn_chaos_model_amd import specialize as S
    hipcc()   # RuntimeError when ROCm's compiler is missing: nothing to test then
    monkeypatch.setenv("BNN_SPEC_CACHE", str(tmp_path))
    assert S.cache_dir() == str(tmp_path)
    a = arch(N)
    cands = S.candidates(a, False)                         # every (waves, variant) form the builder accepts, compiled side by side
    assert {(i["w8"], i["flags"]) for _, i in cands} == {(w, f) for w in (True, False) for f in S.VARIANTS}
    assert "n_wres = " in N.spec_source(a, False, True, N.SPEC_POOL_REGS | N.SPEC_RESIDENT)   # (measured, not searched: weights resident in VGPRs)
    with pytest.raises(N.NativeError):
        N.spec_source(arch(N, 80, 20, 1, 1), False, True, N.SPEC_POOL_REGS | N.SPEC_RESIDENT)   # 160 weight registers: not offered
    with pytest.raises(N.NativeError):
        N.spec_source(a, False, True, N.SPEC_BLOCK_MAJOR | N.SPEC_RESIDENT)
    image, info = S.best_variant(a, False)                 # no GPU here: the static ranking -- eight waves, pool in registers, no scratch
    assert b"bnn_spec_forward" in image and info["scratch"] == 0 and info["w8"] is True and info["flags"] == N.SPEC_POOL_REGS
    assert 0 < info["vgpr"] <= 256 and 0 < info["lds"] <= 160 * 1024
    files = sorted(p.name for p in tmp_path.iterdir())
    assert len(files) == 2 * len(cands) and sum(f.endswith(".hsaco") for f in files) == len(cands)
    image2, info2 = S.best_variant(a, False)               # second time: from the cache, same bytes, same report
    assert image2 == image and info2 == info and sorted(p.name for p in tmp_path.iterdir()) == files
    monkeypatch.setenv("BNN_SPEC_DEFINES", "BNN_GEN_ABLATE=2")   # measurement builds key differently
    S.best_variant(a, False)
    assert len(list(tmp_path.iterdir())) == 4 * len(cands)
    monkeypatch.delenv("BNN_SPEC_DEFINES")
    # the ranking without a GPU (read off profiles/r04_spec_tuning.jsonl): a wave on every SIMD first, then no scratch, then more waves
    assert all(i["nwaves"] == (8 if i["w8"] else 4) for _, i in cands)
    fake = lambda w8, flags, scratch, lds=100000, vgpr=200, agpr=0, nwaves=None: (b"", dict(w8=w8, flags=flags, scratch=scratch, lds=lds, vgpr=vgpr, agpr=agpr,
                                                                                               nwaves=nwaves or (8 if w8 else 4)))
    pick = lambda *c: (lambda i: (i["w8"], i["flags"]))(S.rank_static(list(c))[0][1])
    assert pick(fake(False, 2, 0, lds=137000, vgpr=256, agpr=144, nwaves=1), fake(False, 1, 892, lds=152880, vgpr=256, agpr=256)) == (False, 1)   # hidden 128
    assert pick(fake(True, 1, 652, lds=150000, vgpr=256), fake(False, 1, 0, lds=61744, vgpr=256, agpr=196), fake(False, 2, 0, lds=70000, vgpr=256, agpr=190)) == (False, 1)   # 82 features, noisy
    assert pick(fake(False, 1, 0), fake(True, 1, 0), fake(True, 2, 0), fake(False, 2, 0)) == (True, 1)                                          # all clean: eight waves
    assert pick(fake(False, 1, 0, lds=120000, vgpr=250, agpr=200), fake(False, 1, 0, lds=60000, vgpr=170, agpr=50))[0] is False
    assert S.rank_static([fake(False, 1, 0, lds=120000, vgpr=250, agpr=200), fake(False, 1, 0, lds=60000, vgpr=170, agpr=50)])[0][1]["lds"] == 60000   # two four-wave workgroups per CU beat one


def test_cache_is_private_and_keyed_by_the_loaded_library(N, tmp_path, monkeypatch):
    """Code objects are loaded from the cache unchecked, and a specialised kernel takes the library's GenParams block by value: the
    cache directory must be the user's own (not writable by group / others), and the key must name the library the code object was
    compiled against (ABI version, sizeof(GenParams), the .so's source hash, its build flags)."""
    import os
    from bnn_chaos_model_amd import specialize as S
    shared = tmp_path / "shared"
    shared.mkdir()
    os.chmod(shared, 0o777)
    monkeypatch.setenv("BNN_SPEC_CACHE", str(shared))
    with pytest.raises(RuntimeError, match="private"):
        S.cache_dir()
    mine = tmp_path / "mine" / "nested"
    monkeypatch.setenv("BNN_SPEC_CACHE", str(mine))
    assert S.cache_dir() == str(mine) and (os.stat(mine).st_mode & 0o077) == 0          # created 0700
    lid = S._library_id()
    L = N.lib()
    assert f"abi{L.bnn_abi_version()}" in lid and f"genparams{L.bnn_gen_params_bytes()}" in lid and L.bnn_gen_params_bytes() > 300
    src = N.spec_source(arch(N), False, True, N.SPEC_POOL_REGS)
    k0 = S._key(src)
    monkeypatch.setattr(S, "_library_id", lambda: lid + ":another-build")
    assert S._key(src) != k0 and S._choice_path(arch(N), False, None) != ""            # another library: another cache entry
    # a tuning run in which no candidate can be timed raises instead of handing back an untested form
    monkeypatch.undo()
    monkeypatch.setenv("BNN_SPEC_CACHE", str(mine))
    monkeypatch.setattr(S, "candidates", lambda *a, **k: [(b"x", dict(w8=True, flags=1, scratch=0, lds=1, vgpr=1, agpr=0, nwaves=8)),
                                                           (b"y", dict(w8=False, flags=1, scratch=0, lds=1, vgpr=1, agpr=0, nwaves=4))])
    def boom(image, info):
        raise RuntimeError("does not load")
    with pytest.raises(RuntimeError, match="could be loaded and timed"):
        S.best_variant(arch(N), False, measure=boom)


def test_embedded_unit_matches_its_generator(N):
    """csrc/bnn_fwd_v50spec.hip (the pretrained network's two specialised forms, compiled into the library) is generated text: the
    committed file must be what the library's generator writes today (scripts/regen_embedded.py rewrites it)."""
    import os
    from bnn_chaos_model_amd.csrc import build
    with open(os.path.join(os.path.dirname(build.__file__), "bnn_fwd_v50spec.hip")) as f:
        assert f.read() == N.spec_embedded_source()
    src = N.spec_embedded_source()
    assert "bnn_spec_forward_v50q" in src and "bnn_spec_forward_v50n" in src and "in_q = 8" in src and "launch_fwd_v50spec" in src
