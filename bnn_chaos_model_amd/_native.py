"""ctypes binding of the C ABI in include/bnn_chaos_hip.h (libbnn_chaos_hip.so, gfx950 only).

There is NO CPU fallback: if the shared library is missing or a call fails this module raises.
torch is imported first so that the library binds to the HIP runtime torch already loaded
(both export soname libamdhip64.so.7) and device pointers / streams are interchangeable.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede loading the HIP library)

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("BNN_CHAOS_SO") or os.path.join(_HERE, "csrc", "libbnn_chaos_hip.so")  # override: A/B builds

BNN_OK = 0
ABI_VERSION = 4  # include/bnn_chaos_hip.h: BNN_ABI_VERSION
ERR_INVALID, ERR_UNSUPPORTED, ERR_HIP, ERR_NO_DEVICE, ERR_RANGE = -1, -2, -3, -4, -5


class BnnArch(C.Structure):
    _fields_ = [("n_features", C.c_int32), ("hidden", C.c_int32), ("latent", C.c_int32), ("fix_megno", C.c_int32),
                ("zero_mask", C.c_uint64), ("lowest_std", C.c_float), ("pad", C.c_float), ("depth_in", C.c_int32), ("depth_out", C.c_int32)]


class BnnGrid(C.Structure):
    _fields_ = [("B", C.c_int64), ("T", C.c_int32), ("J", C.c_int32), ("nchunks", C.c_int32),
                ("systems_per_block", C.c_int32), ("noisy", C.c_int32), ("engine", C.c_int32), ("chunk_B", C.c_int64), ("chunk_off", C.c_int64),
                ("nonfinite", C.c_void_p)]   # device record of bnn_nonfinite_scan_f32 for this call's x, or NULL = x is assumed finite


class BnnStats(C.Structure):
    _fields_ = [("tn_nsamp", C.c_int32), ("tn_left", C.c_float), ("prior_thr", C.c_float), ("prior_m", C.c_int32),
                ("prior_step", C.c_float), ("reserved", C.c_int32), ("prior_surv", C.c_void_p)]


class BnnSketch(C.Structure):
    _fields_ = [("nseg", C.c_int32), ("reserved", C.c_int32), ("lo", C.c_float * 4), ("hi", C.c_float * 4), ("n", C.c_int32 * 4)]


class NativeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"bnn_chaos_hip error {code}: {msg}")
        self.code = code


_lib = None
_vp = C.c_void_p


def lib():
    """Load libbnn_chaos_hip.so; (re)build it with hipcc first when it is missing or older than its sources (needs ROCm;
    build() is a no-op when the library is up to date; an explicit BNN_CHAOS_SO is taken as it is)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.environ.get("BNN_CHAOS_SO"):
        from .csrc import build as _b
        if _b.stale():
            # The library on disk was not built from the sources next to it.  Rebuild; a COMPILE error always propagates (a stale
            # binary must never stand in for sources that do not build).  Only a missing compiler is survivable, and only when
            # BNN_CHAOS_ALLOW_STALE=1 says so explicitly -- then the mismatch is announced, never silent.
            try:
                _b.hipcc()
                have_cc = True
            except RuntimeError:
                have_cc = False
            if have_cc:
                _b.build()
            elif os.path.exists(SO_PATH) and os.environ.get("BNN_CHAOS_ALLOW_STALE") == "1":
                import warnings
                warnings.warn(f"{SO_PATH} does not match the sources next to it (source hash differs) and hipcc is not available to "
                              "rebuild it; loading it because BNN_CHAOS_ALLOW_STALE=1", RuntimeWarning, stacklevel=2)
            else:
                raise RuntimeError(f"{SO_PATH} is missing or was built from other sources than the ones next to it, and hipcc was not "
                                   "found to rebuild it (there is no CPU fallback; set BNN_CHAOS_ALLOW_STALE=1 to load a stale library)")
    L = C.CDLL(SO_PATH)
    L.bnn_last_error.restype = C.c_char_p
    L.bnn_plan_create.argtypes = [C.POINTER(BnnArch), C.POINTER(_vp)]
    L.bnn_plan_destroy.argtypes = [_vp]
    L.bnn_plan_layer_order.argtypes = [_vp, C.c_int, C.c_int, _vp, C.c_int]
    L.bnn_param_count.argtypes = [C.POINTER(BnnArch)]
    L.bnn_layer_order.argtypes = [C.POINTER(BnnArch), C.c_int, C.c_int, _vp, C.c_int]
    L.bnn_fragment_table.argtypes = [C.POINTER(BnnArch), C.c_int, C.c_int, _vp, C.c_int]
    L.bnn_swag_draw_f32.argtypes = [_vp, _vp, _vp, _vp, C.c_int32, C.c_int32, _vp, C.c_int32, _vp, _vp, C.c_float,
                                    C.c_uint64, C.c_int64, _vp, _vp]
    L.bnn_forward_f32.argtypes = [_vp, C.POINTER(BnnGrid), _vp, _vp, _vp, _vp, _vp, C.c_uint64, C.c_int64, C.c_int64,
                                  _vp, _vp, _vp, _vp]
    L.bnn_spec_source.argtypes = [C.POINTER(BnnArch), C.c_int32, C.c_int32, C.c_int32, C.c_char_p, C.c_size_t]
    L.bnn_plan_attach_spec.argtypes = [_vp, C.c_int32, C.c_int32, C.c_int32, _vp, C.c_size_t]
    L.bnn_plan_spec_attached.argtypes = [_vp, C.c_int32]
    L.bnn_spec_embedded_source.argtypes = [C.c_char_p, C.c_size_t]
    L.bnn_gen_params_bytes.restype = C.c_size_t
    L.bnn_nonfinite_record_bytes.argtypes = [C.c_int64]
    L.bnn_nonfinite_record_bytes.restype = C.c_size_t
    L.bnn_nonfinite_scan_f32.argtypes = [_vp, _vp, C.c_int64, C.c_int32, _vp, _vp]
    L.bnn_feature_nn_f32.argtypes = [_vp, C.POINTER(BnnGrid), _vp, _vp, _vp, C.c_uint64, C.c_int64, C.c_int64, _vp, _vp]
    L.bnn_forward_lowp_f32.argtypes = [_vp, C.POINTER(BnnGrid), _vp, _vp, _vp, C.c_uint64, C.c_int64, C.c_int64, C.c_int32,
                                       _vp, _vp, _vp, _vp]
    L.bnn_multiswag_f32.argtypes = [_vp, C.POINTER(BnnGrid), _vp, _vp, _vp, _vp, C.c_int32, C.c_int32, _vp, _vp, _vp,
                                    _vp, C.c_float, C.c_uint64, C.c_int64, C.c_int64, _vp, _vp, _vp, _vp, _vp]
    L.bnn_moments_f64.argtypes = [_vp, C.c_int64, C.c_int64, _vp, C.c_int32, _vp]
    L.bnn_truncnorm_f32.argtypes = [_vp, C.c_int64, _vp, C.c_int32, C.c_double, C.c_double, C.c_uint64, C.c_int64, _vp, _vp]
    L.bnn_prior_resample_f32.argtypes = [_vp, C.c_int64, _vp, _vp, _vp, C.c_int64, _vp, C.c_double, C.c_uint64, C.c_int64, _vp]
    L.bnn_regress_f32.argtypes = [_vp, _vp, _vp, C.c_int64, C.c_int64, _vp, _vp, _vp]
    L.bnn_group_min_f32.argtypes = [_vp, C.c_int64, C.c_int32, _vp, _vp]
    L.bnn_quantiles_f32.argtypes = [_vp, C.c_int64, C.c_int64, _vp, C.c_int32, _vp, _vp]
    L.bnn_feature_pack_f64.argtypes = [_vp, _vp, _vp, C.c_int64, C.c_int32, _vp, _vp, _vp, _vp, _vp]
    L.bnn_philox_normal_f32.argtypes = [C.c_int32, C.c_uint64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                        _vp, _vp]
    L.bnn_build_flags.restype = C.c_char_p
    L.bnn_philox_raw_u32.argtypes = [C.c_uint32] * 6 + [C.c_int64, _vp, _vp]
    L.bnn_prior_table_f32.argtypes = [C.c_double, C.c_double, C.c_int32, _vp, C.POINTER(C.c_double)]
    L.bnn_stats_draw_f32.argtypes = [_vp, C.c_int64, C.c_int64, C.POINTER(BnnStats), C.c_uint64, C.c_int64, C.c_int64, _vp, _vp]
    L.bnn_multiswag_stats_f32.argtypes = [_vp, C.POINTER(BnnGrid), _vp, _vp, _vp, _vp, C.c_int32, C.c_int32, _vp, _vp, _vp,
                                          _vp, C.c_float, C.c_uint64, C.c_int64, C.c_int64, _vp, C.POINTER(BnnStats), _vp, _vp]
    L.bnn_sketch_bins.argtypes = [C.POINTER(BnnSketch)]
    L.bnn_sketch_update_u32.argtypes = [_vp, C.c_int64, C.c_int64, C.c_int32, C.POINTER(BnnSketch), _vp, _vp, _vp]
    L.bnn_sketch_quantiles_f32.argtypes = [_vp, C.c_int64, C.POINTER(BnnSketch), _vp, C.c_int32, _vp, _vp]
    L.bnn_multiswag_moments_f64.argtypes = [_vp, C.POINTER(BnnGrid), _vp, _vp, _vp, _vp, C.c_int32, C.c_int32, _vp, C.c_float, C.c_uint64,
                                            C.c_int64, C.c_int64, C.c_int32, _vp, _vp, _vp, _vp]
    L.bnn_multiswag_bands_f32.argtypes = [_vp, C.POINTER(BnnGrid), _vp, _vp, _vp, _vp, C.c_int32, C.c_int32, _vp, C.c_float, C.c_uint64,
                                          C.c_int64, C.c_int64, C.c_int32, _vp, _vp, C.POINTER(BnnStats), C.c_int32, C.POINTER(BnnSketch),
                                          _vp, _vp, _vp]
    if L.bnn_abi_version() != ABI_VERSION:
        raise NativeError(-1, f"ABI version mismatch: {SO_PATH} reports {L.bnn_abi_version()}, this binding is written for {ABI_VERSION}")
    _lib = L
    return L


EXPORTS = ("bnn_abi_version", "bnn_last_error", "bnn_device_count", "bnn_param_count", "bnn_build_flags", "bnn_plan_create",
           "bnn_plan_destroy", "bnn_plan_layer_order", "bnn_layer_order", "bnn_fragment_table", "bnn_swag_draw_f32", "bnn_forward_f32", "bnn_multiswag_f32",
           "bnn_moments_f64", "bnn_truncnorm_f32", "bnn_prior_resample_f32", "bnn_regress_f32", "bnn_group_min_f32", "bnn_quantiles_f32", "bnn_feature_pack_f64", "bnn_philox_normal_f32", "bnn_philox_raw_u32",
           "bnn_prior_table_f32", "bnn_stats_draw_f32", "bnn_multiswag_stats_f32", "bnn_sketch_bins", "bnn_sketch_update_u32",
           "bnn_sketch_quantiles_f32", "bnn_forward_lowp_f32", "bnn_multiswag_moments_f64", "bnn_multiswag_bands_f32", "bnn_feature_nn_f32", "bnn_spec_source", "bnn_plan_attach_spec",
           "bnn_plan_spec_attached", "bnn_spec_embedded_source", "bnn_gen_params_bytes", "bnn_nonfinite_record_bytes", "bnn_nonfinite_scan_f32")


def check(rc):
    if rc < 0:
        raise NativeError(rc, lib().bnn_last_error().decode())
    return rc


def ptr(t):
    """Device pointer of a contiguous CUDA(HIP) tensor, or NULL for None."""
    if t is None:
        return None
    if not t.is_cuda or not t.is_contiguous():
        raise ValueError("expected a contiguous tensor on the GPU")
    return t.data_ptr()


# torch's public accessors build a Stream object (or run a lazy-init check) per call: 2-5 us each, five to six of them per op on the scripts'
# per-chunk route.  The private C entry points return the same numbers directly; fall back to the public API where a torch build lacks them.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def current_device():
    """Index of the current GPU (torch.cuda.current_device())."""
    if _raw_device is not None and torch.cuda.is_initialized():
        return _raw_device()
    return torch.cuda.current_device()


def stream_ptr(device_index=None):
    """Raw handle of torch's current stream on the given (default: the current) device."""
    if _raw_stream is not None and torch.cuda.is_initialized():
        return _raw_stream(current_device() if device_index is None else device_index)
    return torch.cuda.current_stream(device_index).cuda_stream


SPEC_POOL_REGS = 1
SPEC_BLOCK_MAJOR = 2
SPEC_RESIDENT = 4


def _waves_code(w8):
    """None -> -1 (builder's choice), False -> 0 (at most four waves), True -> 1 (eight), 2 -> sixteen (four per SIMD at 128 registers)."""
    return -1 if w8 is None else (2 if w8 == 2 and w8 is not True else int(bool(w8)))


def spec_source(arch, noisy=False, w8=None, flags=0):
    w = _waves_code(w8)
    n = check(lib().bnn_spec_source(C.byref(arch), w, int(bool(noisy)), int(flags), None, 0))
    buf = C.create_string_buffer(n + 1)
    check(lib().bnn_spec_source(C.byref(arch), w, int(bool(noisy)), int(flags), buf, n + 1))
    return buf.value.decode()


def spec_embedded_source():
    """Text of csrc/bnn_fwd_v50spec.hip as the library would generate it now (scripts/regen_embedded.py writes it)."""
    n = check(lib().bnn_spec_embedded_source(None, 0))
    buf = C.create_string_buffer(n + 1)
    check(lib().bnn_spec_embedded_source(buf, n + 1))
    return buf.value.decode()


class Plan:
    """Owns a bnn_plan (device operand tables) for one architecture / column mask.
    depth_in / depth_out = hparams['in'] / hparams['out'] (the `layers` argument of the reference's mlp(), spock_reg_model.py:301-321)."""

    def __init__(self, zero_mask, lowest_std=0.5, n_features=41, hidden=40, latent=20, fix_megno=False, depth_in=1, depth_out=1):
        self.arch = BnnArch(n_features, hidden, latent, int(bool(fix_megno)), zero_mask, lowest_std, 0.0, int(depth_in), int(depth_out))
        self.d = check(lib().bnn_param_count(C.byref(self.arch)))
        self.fix_megno = bool(fix_megno)
        self.n_features, self.hidden, self.latent = int(n_features), int(hidden), int(latent)
        self.depth_in, self.depth_out = int(depth_in), int(depth_out)
        self.summary_width = 2 * latent + (2 if fix_megno else 0)   # [mu_sample | std_sample | megno mean, megno std]
        # the pretrained ensemble's network runs on the register-resident kernels (at T % 4 == 0), everything else on the generic engine
        self.v50net = (self.n_features, self.hidden, self.latent, self.depth_in, self.depth_out) == (41, 40, 20, 1, 1)
        self.n_linear = (1 if depth_in == 0 else depth_in + 2) + (1 if depth_out == 0 else depth_out + 2)
        h = _vp()
        check(lib().bnn_plan_create(C.byref(self.arch), C.byref(h)))
        self.handle = h

    def spec_source(self, noisy=False, w8=None, flags=0):
        """HIP source of this network's specialised form of the generic engine (bnn_spec_source; needs no device)."""
        return spec_source(self.arch, noisy, w8, flags)

    def attach_spec(self, image, noisy=False, w8=None, flags=0):
        """Load a compiled specialised form (code object bytes) into the plan; the current device must be the plan's."""
        check(lib().bnn_plan_attach_spec(self.handle, int(bool(noisy)), _waves_code(w8), int(flags), image, len(image)))

    def spec_attached(self, noisy=False):
        return bool(check(lib().bnn_plan_spec_attached(self.handle, int(bool(noisy)))))

    def layer_order(self, layer, noisy=False):
        import numpy as np
        buf = np.zeros(256, np.int32)
        n = check(lib().bnn_plan_layer_order(self.handle, layer, int(noisy), buf.ctypes.data, 256))
        return buf[:n].copy()

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                lib().bnn_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass
