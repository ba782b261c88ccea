"""ctypes binding of libicicle_hip.so (the C ABI declared in include/icicle_hip.h).

The product path has NO CPU fallback: if the HIP library is missing this module raises at import
time; if it is present but there is no GPU every compute entry point returns INVALID_DEVICE.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ICICLE_HIP_LIB", os.path.join(_HERE, "lib", "libicicle_hip.so"))


class IcicleError(RuntimeError):
    NAMES = [
        "SUCCESS", "INVALID_DEVICE", "OUT_OF_MEMORY", "INVALID_POINTER", "ALLOCATION_FAILED",
        "DEALLOCATION_FAILED", "COPY_FAILED", "SYNCHRONIZATION_FAILED", "STREAM_CREATION_FAILED",
        "STREAM_DESTRUCTION_FAILED", "API_NOT_IMPLEMENTED", "INVALID_ARGUMENT",
    ]

    def __init__(self, code, what=""):
        self.code = int(code)
        name = self.NAMES[self.code] if 0 <= self.code < len(self.NAMES) else str(self.code)
        super().__init__(f"eIcicleError::{name} {what}".strip())


def check(code, what=""):
    if code != 0:
        raise IcicleError(code, what)


class Device(ctypes.Structure):
    """icicle::Device (include/icicle/device.h:14-48): 68 bytes, id at offset 64."""
    _fields_ = [("type", ctypes.c_char * 64), ("id", ctypes.c_int)]


class DeviceProperties(ctypes.Structure):
    _fields_ = [("using_host_memory", ctypes.c_bool), ("num_memory_regions", ctypes.c_int),
                ("supports_pinned_memory", ctypes.c_bool)]


class MSMConfig(ctypes.Structure):
    """icicle::MSMConfig (include/icicle/msm.h:21-53), 40 bytes; mirrors the Rust #[repr(C)] struct
    (wrappers/rust/icicle-core/src/msm/mod.rs:13-49)."""
    _fields_ = [
        ("stream", ctypes.c_void_p),
        ("precompute_factor", ctypes.c_int),
        ("c", ctypes.c_int),
        ("bitsize", ctypes.c_int),
        ("batch_size", ctypes.c_int),
        ("are_points_shared_in_batch", ctypes.c_bool),
        ("are_scalars_on_device", ctypes.c_bool),
        ("are_scalars_montgomery_form", ctypes.c_bool),
        ("are_points_on_device", ctypes.c_bool),
        ("are_points_montgomery_form", ctypes.c_bool),
        ("are_results_on_device", ctypes.c_bool),
        ("is_async", ctypes.c_bool),
        ("ext", ctypes.c_void_p),
    ]

    @classmethod
    def default(cls):
        # default_msm_config() (msm.h:60-78)
        return cls(None, 1, 0, 0, 1, True, False, False, False, False, False, False, None)


class NTTConfigU32(ctypes.Structure):
    """icicle::NTTConfig<S> for a 4-byte S (include/icicle/ntt.h:53-64), 40 bytes."""
    _fields_ = [
        ("stream", ctypes.c_void_p),
        ("coset_gen", ctypes.c_uint32),
        ("batch_size", ctypes.c_int),
        ("columns_batch", ctypes.c_bool),
        ("ordering", ctypes.c_int),
        ("are_inputs_on_device", ctypes.c_bool),
        ("are_outputs_on_device", ctypes.c_bool),
        ("is_async", ctypes.c_bool),
        ("ext", ctypes.c_void_p),
    ]

    @classmethod
    def default(cls):
        # default_ntt_config() (ntt.h:72-85)
        return cls(None, 1, 1, False, 0, False, False, False, None)


class NTTConfigU256(ctypes.Structure):
    """icicle::NTTConfig<S> for the curves' 32-byte scalar_t (include/icicle/ntt.h:53-64), 64 bytes."""
    _fields_ = [
        ("stream", ctypes.c_void_p),
        ("coset_gen", ctypes.c_uint32 * 8),
        ("batch_size", ctypes.c_int),
        ("columns_batch", ctypes.c_bool),
        ("ordering", ctypes.c_int),
        ("are_inputs_on_device", ctypes.c_bool),
        ("are_outputs_on_device", ctypes.c_bool),
        ("is_async", ctypes.c_bool),
        ("ext", ctypes.c_void_p),
    ]

    @classmethod
    def default(cls):
        one = (ctypes.c_uint32 * 8)(1, 0, 0, 0, 0, 0, 0, 0)
        return cls(None, one, 1, False, 0, False, False, False, None)

    def set_coset_gen(self, g: int):
        for i in range(8):
            self.coset_gen[i] = (g >> (32 * i)) & 0xFFFFFFFF


class NTTConfigU64(ctypes.Structure):
    """icicle::NTTConfig<S> for goldilocks' 8-byte scalar_t, 40 bytes (include/icicle_hip.h icicle_ntt_config_u64_t)."""
    _fields_ = [
        ("stream", ctypes.c_void_p),
        ("coset_gen", ctypes.c_uint32 * 2),
        ("batch_size", ctypes.c_int),
        ("columns_batch", ctypes.c_bool),
        ("ordering", ctypes.c_int),
        ("are_inputs_on_device", ctypes.c_bool),
        ("are_outputs_on_device", ctypes.c_bool),
        ("is_async", ctypes.c_bool),
        ("ext", ctypes.c_void_p),
    ]

    @classmethod
    def default(cls):
        return cls(None, (ctypes.c_uint32 * 2)(1, 0), 1, False, 0, False, False, False, None)

    def set_coset_gen(self, g: int):
        self.coset_gen[0], self.coset_gen[1] = g & 0xFFFFFFFF, (g >> 32) & 0xFFFFFFFF


class VecOpsConfig(ctypes.Structure):
    """icicle::VecOpsConfig (include/icicle/vec_ops.h:19-37), 32 bytes."""
    _fields_ = [
        ("stream", ctypes.c_void_p),
        ("is_a_on_device", ctypes.c_bool),
        ("is_b_on_device", ctypes.c_bool),
        ("is_result_on_device", ctypes.c_bool),
        ("is_async", ctypes.c_bool),
        ("batch_size", ctypes.c_int),
        ("columns_batch", ctypes.c_bool),
        ("ext", ctypes.c_void_p),
    ]

    @classmethod
    def default(cls):
        return cls(None, False, False, False, False, 1, False, None)


class NTTInitDomainConfig(ctypes.Structure):
    _fields_ = [("stream", ctypes.c_void_p), ("is_async", ctypes.c_bool), ("ext", ctypes.c_void_p)]

    @classmethod
    def default(cls):
        return cls(None, False, None)


assert ctypes.sizeof(Device) == 68 and Device.id.offset == 64
assert ctypes.sizeof(MSMConfig) == 40 and MSMConfig.ext.offset == 32
assert ctypes.sizeof(NTTConfigU32) == 40 and NTTConfigU32.ordering.offset == 20
assert ctypes.sizeof(NTTInitDomainConfig) == 24
assert ctypes.sizeof(NTTConfigU256) == 64 and NTTConfigU256.batch_size.offset == 40 and NTTConfigU256.ordering.offset == 48
assert NTTConfigU256.ext.offset == 56
assert ctypes.sizeof(VecOpsConfig) == 32 and VecOpsConfig.batch_size.offset == 12 and VecOpsConfig.ext.offset == 24

# every symbol include/icicle_hip.h declares (tests/test_abi.py checks the header against this)
RUNTIME_SYMBOLS = [
    "icicle_load_backend", "icicle_load_backend_from_env_or_default", "icicle_set_device",
    "icicle_set_default_device", "icicle_get_active_device", "icicle_is_host_memory",
    "icicle_is_active_device_memory", "icicle_get_device_count", "icicle_malloc", "icicle_malloc_async",
    "icicle_free", "icicle_free_async", "icicle_get_available_memory", "icicle_memset",
    "icicle_memset_async", "icicle_copy", "icicle_copy_async", "icicle_copy_to_host",
    "icicle_copy_to_host_async", "icicle_copy_to_device", "icicle_copy_to_device_async",
    "icicle_create_stream", "icicle_destroy_stream", "icicle_stream_synchronize",
    "icicle_device_synchronize", "icicle_get_device_properties", "icicle_is_device_available",
    "icicle_get_registered_devices",
    "create_config_extension", "destroy_config_extension", "config_extension_set_int",
    "config_extension_set_bool", "config_extension_get_int", "config_extension_get_bool",
    "clone_config_extension",
]
CURVES = ["bn254", "bls12_381", "bls12_377", "grumpkin"]  # G1 MSM
G2_CURVES = ["bn254", "bls12_381", "bls12_377"]
ECNTT_CURVES = ["bn254", "bls12_381", "bls12_377"]
NTT_FIELDS = ["babybear", "koalabear"]
SCALAR_NTT_FIELDS = ["bn254", "bls12_381", "bls12_377", "stark252"]  # NTT over a 256-bit field, 8-word elements
BIG_VEC_FIELDS = SCALAR_NTT_FIELDS + ["grumpkin"]  # element-wise ops / Montgomery conversion over 8-word scalars
GOLD = "goldilocks"  # 2-word elements, NTTConfigU64; extension field = 2 components
API_SYMBOLS = (
    [f"{c}_{s}" for c in CURVES for s in ("msm", "msm_precompute_bases", "hip_generate_affine_points", "hip_projective_sum")]
    + [f"{c}_g2_{s}" for c in G2_CURVES for s in ("msm", "msm_precompute_bases", "hip_generate_affine_points", "hip_projective_sum")]
    + [f"icicle_hip_{c}_g2_{s}" for c in G2_CURVES for s in ("msm", "msm_precompute_bases")]
    + [f"{f}_{s}" for f in NTT_FIELDS for s in ("ntt", "ntt_init_domain", "ntt_release_domain", "get_root_of_unity",
                                               "get_root_of_unity_from_domain", "extension_ntt", "hip_twiddle_rows")]
    + [f"{f}_{s}" for f in SCALAR_NTT_FIELDS for s in ("ntt", "ntt_init_domain", "ntt_release_domain", "get_root_of_unity",
                                                      "get_root_of_unity_from_domain")]
    + [f"icicle_hip_{f}_{s}" for f in SCALAR_NTT_FIELDS for s in ("ntt", "ntt_init_domain", "ntt_release_domain",
                                                                 "get_root_of_unity_from_domain")]
    + [f"{pre}{c}_ecntt" for pre in ("", "icicle_hip_") for c in ECNTT_CURVES]
    + [f"{GOLD}_{s}" for s in ("ntt", "extension_ntt", "ntt_init_domain", "ntt_release_domain", "get_root_of_unity", "get_root_of_unity_from_domain")]
    + [f"icicle_hip_{GOLD}_{s}" for s in ("ntt", "extension_ntt", "ntt_init_domain", "ntt_release_domain", "get_root_of_unity_from_domain")]
    + [f"{pre}{f}_{op}" for pre in ("", "icicle_hip_") for f in NTT_FIELDS + BIG_VEC_FIELDS + [GOLD]
       for op in ("vector_add", "vector_sub", "vector_mul", "scalar_mul_vec", "scalar_add_vec", "scalar_sub_vec", "bit_reverse")]
    + [f"{pre}{f}_matrix_transpose" for pre in ("", "icicle_hip_") for f in NTT_FIELDS + BIG_VEC_FIELDS + [GOLD]]
    + [f"{pre}{f}_extension_matrix_transpose" for pre in ("", "icicle_hip_") for f in NTT_FIELDS + [GOLD]]
    + ["icicle_hip_msm_plan", "icicle_hip_msm_plan_info", "icicle_hip_version", "icicle_hip_kernel_timing", "icicle_hip_enable_kernel_timing", "icicle_hip_set_device",
       "icicle_hip_ubench_mixed_add", "icicle_hip_ubench_gather", "icicle_hip_ubench_ntt_pass", "icicle_hip_selftest_inplace_products", "icicle_hip_selftest_quad_group_ops", "icicle_hip_release_workspace", "icicle_hip_workspace_bytes",
       "icicle_hip_msm_release_resident_bases", "icicle_hip_multi_stats", "icicle_hip_multi_stats2", "icicle_hip_collectives_info", "icicle_hip_test_set_virtual_devices", "icicle_hip_test_set_no_peer_access",
       "icicle_hip_set_collectives_library", "icicle_hip_test_inject_failure",
       "icicle_hip_create_config_extension", "icicle_hip_destroy_config_extension", "icicle_hip_config_extension_set_int",
       "icicle_hip_config_extension_set_bool"]
    + [f"icicle_hip_{c}_{s}" for c in CURVES for s in ("msm", "msm_precompute_bases")]
    + [f"icicle_hip_{f}_{s}" for f in NTT_FIELDS for s in ("ntt", "extension_ntt", "ntt_init_domain", "ntt_release_domain",
                                                          "get_root_of_unity_from_domain")]
    + [f"{pre}{f}_scalar_convert_montgomery" for pre in ("", "icicle_hip_") for f in BIG_VEC_FIELDS + NTT_FIELDS + [GOLD]]
    + [f"{pre}{f}_extension_scalar_convert_montgomery" for pre in ("", "icicle_hip_") for f in NTT_FIELDS + [GOLD]]
    + [f"{pre}{c}_{k}_convert_montgomery" for pre in ("", "icicle_hip_") for c in CURVES for k in ("affine", "projective")]
    + [f"{pre}{c}_g2_{k}_convert_montgomery" for pre in ("", "icicle_hip_") for c in G2_CURVES for k in ("affine", "projective")]
)

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(hipcc --offload-arch=gfx950). icicle_amd has no CPU fallback."
    )
lib = ctypes.CDLL(LIB_PATH)
for _s in RUNTIME_SYMBOLS + API_SYMBOLS:
    getattr(lib, _s)  # AttributeError if the library does not export a declared symbol
lib.icicle_hip_version.restype = ctypes.c_char_p
lib.create_config_extension.restype = ctypes.c_void_p
lib.clone_config_extension.restype = ctypes.c_void_p
lib.clone_config_extension.argtypes = [ctypes.c_void_p]
lib.destroy_config_extension.argtypes = [ctypes.c_void_p]
lib.config_extension_set_int.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
lib.config_extension_set_bool.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_bool]
lib.config_extension_get_int.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
lib.config_extension_get_bool.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
lib.config_extension_get_bool.restype = ctypes.c_bool
for _n in ("icicle_malloc",):
    getattr(lib, _n).argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
lib.icicle_malloc_async.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_void_p]
lib.icicle_free.argtypes = [ctypes.c_void_p]
lib.icicle_free_async.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
lib.icicle_memset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
lib.icicle_memset_async.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
for _n in ("icicle_copy", "icicle_copy_to_host", "icicle_copy_to_device"):
    getattr(lib, _n).argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
for _n in ("icicle_copy_async", "icicle_copy_to_host_async", "icicle_copy_to_device_async"):
    getattr(lib, _n).argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
lib.icicle_is_host_memory.argtypes = [ctypes.c_void_p]
lib.icicle_is_active_device_memory.argtypes = [ctypes.c_void_p]
lib.icicle_create_stream.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
lib.icicle_destroy_stream.argtypes = [ctypes.c_void_p]
lib.icicle_stream_synchronize.argtypes = [ctypes.c_void_p]
lib.icicle_get_available_memory.argtypes = [ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
lib.icicle_get_registered_devices.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
for _c in CURVES:
    getattr(lib, f"{_c}_msm").argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(MSMConfig), ctypes.c_void_p]
    getattr(lib, f"{_c}_msm_precompute_bases").argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(MSMConfig), ctypes.c_void_p]
    getattr(lib, f"{_c}_hip_projective_sum").argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    getattr(lib, f"{_c}_hip_generate_affine_points").argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_bool, ctypes.c_void_p]
for _c in G2_CURVES:
    getattr(lib, f"{_c}_g2_msm").argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(MSMConfig), ctypes.c_void_p]
    getattr(lib, f"{_c}_g2_msm_precompute_bases").argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(MSMConfig), ctypes.c_void_p]
    getattr(lib, f"{_c}_g2_hip_projective_sum").argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    getattr(lib, f"{_c}_g2_hip_generate_affine_points").argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_bool, ctypes.c_void_p]
for _f in NTT_FIELDS:
    getattr(lib, f"{_f}_ntt").argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(NTTConfigU32), ctypes.c_void_p]
    getattr(lib, f"{_f}_extension_ntt").argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(NTTConfigU32), ctypes.c_void_p]
    getattr(lib, f"{_f}_ntt_init_domain").argtypes = [ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(NTTInitDomainConfig)]
    getattr(lib, f"{_f}_hip_twiddle_rows").argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_bool, ctypes.c_void_p]
    getattr(lib, f"{_f}_get_root_of_unity").argtypes = [ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint32)]
    getattr(lib, f"{_f}_get_root_of_unity_from_domain").argtypes = [ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint32)]
for _f in ECNTT_CURVES:
    getattr(lib, f"{_f}_ecntt").argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(NTTConfigU256), ctypes.c_void_p]
for _f in SCALAR_NTT_FIELDS:
    getattr(lib, f"{_f}_ntt").argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(NTTConfigU256), ctypes.c_void_p]
    getattr(lib, f"{_f}_ntt_init_domain").argtypes = [ctypes.c_void_p, ctypes.POINTER(NTTInitDomainConfig)]
    getattr(lib, f"{_f}_get_root_of_unity").argtypes = [ctypes.c_uint64, ctypes.c_void_p]
    getattr(lib, f"{_f}_get_root_of_unity_from_domain").argtypes = [ctypes.c_uint64, ctypes.c_void_p]
for _s in ("ntt", "extension_ntt"):
    getattr(lib, f"{GOLD}_{_s}").argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(NTTConfigU64), ctypes.c_void_p]
getattr(lib, f"{GOLD}_ntt_init_domain").argtypes = [ctypes.c_void_p, ctypes.POINTER(NTTInitDomainConfig)]
getattr(lib, f"{GOLD}_get_root_of_unity").argtypes = [ctypes.c_uint64, ctypes.c_void_p]
getattr(lib, f"{GOLD}_get_root_of_unity_from_domain").argtypes = [ctypes.c_uint64, ctypes.c_void_p]
for _f in NTT_FIELDS + BIG_VEC_FIELDS + [GOLD]:
    for _op in ("vector_add", "vector_sub", "vector_mul", "scalar_mul_vec", "scalar_add_vec", "scalar_sub_vec"):
        getattr(lib, f"{_f}_{_op}").argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(VecOpsConfig), ctypes.c_void_p]
    getattr(lib, f"{_f}_bit_reverse").argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(VecOpsConfig), ctypes.c_void_p]
for _n in API_SYMBOLS:
    if _n.endswith("matrix_transpose"):
        getattr(lib, _n).argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(VecOpsConfig), ctypes.c_void_p]
    if _n.endswith("convert_montgomery"):
        getattr(lib, _n).argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_bool, ctypes.POINTER(VecOpsConfig), ctypes.c_void_p]
lib.icicle_hip_kernel_timing.argtypes = [ctypes.c_int, ctypes.c_bool, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]
lib.icicle_hip_enable_kernel_timing.argtypes = [ctypes.c_bool]
lib.icicle_hip_msm_plan.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(MSMConfig), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
lib.icicle_hip_msm_plan_info.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(MSMConfig), ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
lib.icicle_hip_ubench_mixed_add.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
lib.icicle_hip_ubench_ntt_pass.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
lib.icicle_hip_ubench_gather.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(ctypes.c_double)]
lib.icicle_hip_workspace_bytes.argtypes = [ctypes.POINTER(ctypes.c_size_t)]
lib.icicle_hip_selftest_inplace_products.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
lib.icicle_hip_selftest_quad_group_ops.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
lib.icicle_hip_msm_release_resident_bases.argtypes = [ctypes.c_void_p]
lib.icicle_hip_multi_stats.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_bool]
lib.icicle_hip_multi_stats2.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int, ctypes.c_bool]
lib.icicle_hip_test_set_virtual_devices.argtypes = [ctypes.c_int]
lib.icicle_hip_set_collectives_library.argtypes = [ctypes.c_char_p]
lib.icicle_hip_test_inject_failure.argtypes = [ctypes.c_int, ctypes.c_int]


def multi_stats(reset=False):
    """dict of the multi-device / pipelined-path counters (icicle_hip_multi_stats)"""
    out = (ctypes.c_uint64 * 8)()
    check(lib.icicle_hip_multi_stats2(out, 8, reset), "multi_stats")
    return dict(zip(("staged_base_bytes", "staged_scalar_bytes", "exchanged_bucket_bytes", "resident_base_hits", "threaded_calls", "exchange_messages", "peer_staged_copies",
                     "plan_fallbacks"),
                    [int(v) for v in out]))
