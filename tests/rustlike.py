"""TEST INFRASTRUCTURE: the Rust wrapper's call surface, restated over ctypes.

No Rust (or Go) toolchain exists in this image, so the reference's wrapper tests cannot be compiled. What a wrapper does on
the MSM / NTT path is thin and mechanical -- fill a #[repr(C)] config, derive the `are_*_on_device` flags from the slice
types, make ONE extern "C" call into the reference's frontend library -- so this module restates exactly that layer
(same names, same derivations, same FFI symbol per call) on the REAL reference runtime + frontend libraries (oracle/_ref):

  wrappers/rust/icicle-runtime/src/{runtime.rs,stream.rs,memory.rs,config.rs,test_utilities.rs}
  wrappers/rust/icicle-core/src/{msm/mod.rs,ntt/mod.rs,ecntt/mod.rs,matrix_ops/mod.rs,traits.rs}

tests/rust_suite_driver.py replays the wrapper's test functions call for call on top of it, with the HIP plugin as the main
device and the reference's "CPU" device as the ref device (test_utilities.rs:11-64).
"""
import ctypes
import os

import numpy as np

from oracle import ref

C = ctypes
_dev = None


def dev():
    global _dev
    if _dev is None:
        _dev = ref.RefRuntime().lib
        _dev.icicle_malloc_async.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p]
        _dev.icicle_free_async.argtypes = [C.c_void_p, C.c_void_p]
        for n in ("icicle_copy_to_host_async", "icicle_copy_to_device_async", "icicle_copy_async"):
            getattr(_dev, n).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        _dev.create_config_extension.restype = C.c_void_p
        _dev.clone_config_extension.restype = C.c_void_p
        _dev.clone_config_extension.argtypes = [C.c_void_p]
        _dev.destroy_config_extension.argtypes = [C.c_void_p]
        _dev.config_extension_set_int.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        _dev.config_extension_set_bool.argtypes = [C.c_void_p, C.c_char_p, C.c_bool]
    return _dev


def wrap(rc, what=""):
    """eIcicleError::wrap() + .unwrap(): every call in the Rust tests panics on a non-zero code"""
    assert rc == 0, f"IcicleError {rc} {what}"


# ---- icicle-runtime: test_utilities.rs, runtime.rs, stream.rs, config.rs ----
MAIN_DEVICE = ("HIP", 0)
REF_DEVICE = ("CPU", 0)


def _set_device(t, i):
    d = ref.Device(t.encode(), i)
    wrap(dev().icicle_set_device(C.byref(d)), f"set_device {t} {i}")


def test_load_and_init_devices(backend_dir):
    """test_utilities.rs:11-32 (load_backend_from_env_or_default; main = the first registered non-CPU device)"""
    rt = ref.RefRuntime()
    if "HIP" not in rt.registered_devices():
        wrap(rt.load_backend(backend_dir), "load_backend")
    regs = rt.registered_devices().split(",")
    assert len(regs) >= 2, regs
    global MAIN_DEVICE
    MAIN_DEVICE = ([r for r in regs if r != "CPU"][0], 0)


def test_set_main_device():
    _set_device(*MAIN_DEVICE)


def test_set_main_device_with_id(i):
    _set_device(MAIN_DEVICE[0], i)


def test_set_ref_device():
    _set_device(*REF_DEVICE)


def get_device_count():
    n = C.c_int()
    wrap(dev().icicle_get_device_count(C.byref(n)))
    return n.value


class IcicleStream:  # stream.rs
    def __init__(self, handle=None):
        self.handle = handle

    @staticmethod
    def create():
        h = C.c_void_p()
        wrap(dev().icicle_create_stream(C.byref(h)), "create_stream")
        return IcicleStream(h.value)

    def is_null(self):
        return not self.handle

    def synchronize(self):
        wrap(dev().icicle_stream_synchronize(self.handle), "stream_synchronize")

    def destroy(self):
        wrap(dev().icicle_destroy_stream(self.handle), "destroy_stream")
        self.handle = None


def warmup(stream):  # runtime.rs:142-149
    p = C.c_void_p()
    wrap(dev().icicle_malloc_async(C.byref(p), (1 << 28) >> 1, stream.handle), "warmup malloc_async")
    wrap(dev().icicle_free_async(p, stream.handle), "warmup free_async")


class ConfigExtension:  # config.rs: owns a handle of the reference's own ConfigExtension object
    def __init__(self, handle=None):
        self.handle = handle if handle is not None else dev().create_config_extension()

    def set_int(self, key, v):
        dev().config_extension_set_int(self.handle, key.encode(), int(v))

    def set_bool(self, key, v):
        dev().config_extension_set_bool(self.handle, key.encode(), bool(v))

    def clone(self):
        return ConfigExtension(dev().clone_config_extension(self.handle))

    def drop(self):
        if self.handle:
            dev().destroy_config_extension(self.handle)
            self.handle = None


CUDA_MSM_LARGE_BUCKET_FACTOR = "large_bucket_factor"  # msm/mod.rs:52
CUDA_NTT_FAST_TWIDDLES_MODE = "fast_twiddles"  # ntt/mod.rs
CUDA_NTT_ALGORITHM = "ntt_algorithm"
NTT_RADIX2, NTT_MIXED_RADIX = 1, 2  # NttAlgorithm
kNN, kNR, kRN, kRR, kNM, kMN = range(6)
kForward, kInverse = 0, 1


# ---- icicle-runtime: memory.rs ----
class HostSlice:
    """&[T] / HostSlice<T>: a view of a C-contiguous uint32 array of shape (len, words)"""

    def __init__(self, arr):
        assert arr.dtype == np.uint32 and arr.ndim == 2 and arr.flags["C_CONTIGUOUS"]
        self.arr = arr

    def is_on_device(self):
        return False

    def is_on_active_device(self):
        return False

    def ptr(self):
        return self.arr.ctypes.data

    def __len__(self):
        return self.arr.shape[0]

    @property
    def words(self):
        return self.arr.shape[1]

    def __getitem__(self, sl):  # &v[a..b]
        return HostSlice(self.arr[sl])


class DeviceVec:
    """DeviceVec<T> (memory.rs:420-526); `words` = size_of::<T>() / 4. drop() = impl Drop (icicle_free)."""

    def __init__(self, p, count, words):
        self.p, self.count, self.words = p, count, words

    @staticmethod
    def device_malloc_async(count, words, stream):
        p = C.c_void_p()
        assert count * words > 0
        wrap(dev().icicle_malloc_async(C.byref(p), count * words * 4, stream.handle), "malloc_async")
        return DeviceVec(p.value, count, words)

    @staticmethod
    def malloc(count, words):
        p = C.c_void_p()
        wrap(dev().icicle_malloc(C.byref(p), count * words * 4), "device allocation failed")
        return DeviceVec(p.value, count, words)

    @staticmethod
    def from_host_slice(arr):
        v = DeviceVec.malloc(arr.shape[0], arr.shape[1])
        v.copy_from_host(HostSlice(arr))
        return v

    def is_on_device(self):
        return True

    def is_on_active_device(self):
        return dev().icicle_is_active_device_memory(C.c_void_p(self.p)) == 0

    def ptr(self):
        return self.p

    def __len__(self):
        return self.count

    def _check(self, val):
        assert len(self) == len(val), "destination and source slices have different lengths"
        assert self.is_on_active_device(), "not allocated on an active device"

    def copy_from_host(self, val):
        self._check(val)
        wrap(dev().icicle_copy_to_device(self.p, val.ptr(), self.count * self.words * 4), "copy_to_device")

    def copy_from_host_async(self, val, stream):
        self._check(val)
        wrap(dev().icicle_copy_to_device_async(self.p, val.ptr(), self.count * self.words * 4, stream.handle), "copy_to_device_async")

    def copy_to_host(self, val):
        self._check(val)
        wrap(dev().icicle_copy_to_host(val.ptr(), self.p, self.count * self.words * 4), "copy_to_host")

    def copy_to_host_async(self, val, stream):
        self._check(val)
        wrap(dev().icicle_copy_to_host_async(val.ptr(), self.p, self.count * self.words * 4, stream.handle), "copy_to_host_async")

    def to_host_vec(self):
        out = np.zeros((self.count, self.words), dtype=np.uint32)
        self.copy_to_host(HostSlice(out))
        return out

    def drop(self):
        if self.count and self.p:
            rc = dev().icicle_free(C.c_void_p(self.p))
            assert rc == 0, "releasing memory failed due to invalid active device"
            self.p = None


# ---- icicle-core: msm/mod.rs ----
class MSMConfig:
    """msm/mod.rs:15-73. to_c() is the #[repr(C)] image handed to the FFI (== icicle::MSMConfig, msm.h:21-53)."""

    def __init__(self):
        self.stream_handle = None
        self.precompute_factor = 1
        self.c = 0
        self.bitsize = 0
        self.batch_size = 1
        self.are_points_shared_in_batch = True
        self.are_scalars_on_device = False
        self.are_scalars_montgomery_form = False
        self.are_bases_on_device = False
        self.are_bases_montgomery_form = False
        self.are_results_on_device = False
        self.is_async = False
        self.ext = ConfigExtension()

    def clone(self):
        o = MSMConfig.__new__(MSMConfig)
        o.__dict__.update(self.__dict__)
        o.ext = self.ext.clone()
        return o

    def to_c(self):
        return ref.MSMConfig(self.stream_handle, self.precompute_factor, self.c, self.bitsize, self.batch_size,
                             self.are_points_shared_in_batch, self.are_scalars_on_device, self.are_scalars_montgomery_form,
                             self.are_bases_on_device, self.are_bases_montgomery_form, self.are_results_on_device,
                             self.is_async, self.ext.handle)


class Curve:
    """a Projective type of the wrapper (icicle-curves/icicle-<curve>): ScalarField = 8 words, Affine = 2L, Projective = 3L"""

    def __init__(self, name):
        self.name = name
        self.L = ref.CURVE_LIMBS[name]
        self.SW, self.AW, self.PW = 8, 2 * self.L, 3 * self.L
        self.lib = ref._load(name)
        self.flib = C.CDLL(os.path.join(ref.REF_DIR, f"libicicle_field_{name}.so"))
        self.scalar = ScalarField(name, 8, self.flib)

    def _fn(self, lib, sym, argtypes):
        fn = getattr(lib, sym)
        fn.argtypes = argtypes
        return fn

    # GenerateRandom
    def generate_random_affine(self, n):
        out = np.zeros((n, self.AW), dtype=np.uint32)
        self._fn(self.lib, f"{self.name}_generate_affine_points", [C.c_void_p, C.c_size_t])(out.ctypes.data, n)
        return out

    def generate_random_projective(self, n):
        out = np.zeros((n, self.PW), dtype=np.uint32)
        self._fn(self.lib, f"{self.name}_generate_projective_points", [C.c_void_p, C.c_size_t])(out.ctypes.data, n)
        return out

    def zero(self, n):  # P::zero(): (0, 1, 0)
        out = np.zeros((n, self.PW), dtype=np.uint32)
        out[:, self.L] = 1
        return out

    def eq(self, a, b):
        """impl PartialEq for Projective: <curve>_projective_eq element by element"""
        fn = self._fn(self.lib, f"{self.name}_projective_eq", [C.c_void_p, C.c_void_p])
        fn.restype = C.c_bool
        a = np.ascontiguousarray(a)
        b = np.ascontiguousarray(b)
        return a.shape == b.shape and all(fn(a[i].ctypes.data, b[i].ctypes.data) for i in range(a.shape[0]))

    def msm(self, scalars, bases, cfg, results):
        """MSM::msm (msm/mod.rs:90-156) + msm_unchecked (:270-286)"""
        assert len(bases) % cfg.precompute_factor == 0
        bases_size = len(bases) // cfg.precompute_factor
        assert len(scalars) % bases_size == 0
        assert len(scalars) % len(results) == 0
        for s in (scalars, bases, results):
            assert not s.is_on_device() or s.is_on_active_device(), "not allocated on the active device"
        local_cfg = cfg.clone()
        local_cfg.are_points_shared_in_batch = bases_size < len(scalars)
        local_cfg.batch_size = len(results)
        local_cfg.are_scalars_on_device = scalars.is_on_device()
        local_cfg.are_bases_on_device = bases.is_on_device()
        local_cfg.are_results_on_device = results.is_on_device()
        c = local_cfg.to_c()
        fn = self._fn(self.lib, f"{self.name}_msm", [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p])
        rc = fn(scalars.ptr(), bases.ptr(), len(scalars) // len(results), C.byref(c), results.ptr())
        local_cfg.ext.drop()
        wrap(rc, "msm")

    def precompute_bases(self, points, config, output_bases):
        """MSM::precompute_bases (msm/mod.rs:158-182) + precompute_bases_unchecked (:288-302)"""
        assert len(output_bases) == len(points) * config.precompute_factor
        assert output_bases.is_on_device()
        c = config.to_c()
        fn = self._fn(self.lib, f"{self.name}_msm_precompute_bases", [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p])
        wrap(fn(points.ptr(), len(points) // config.batch_size, C.byref(c), output_bases.ptr()), "precompute_bases")

    def ecntt(self, inp, direction, cfg, out):
        """ecntt/mod.rs:27-34: ECNTT::ntt hands cfg to ntt_unchecked as is (no are_*_on_device derivation) -> <curve>_ecntt
        (impl_ntt_without_domain, ntt/mod.rs:302-318)"""
        c = cfg.to_c()
        fn = self._fn(self.lib, f"{self.name}_ecntt", [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p])
        wrap(fn(inp.ptr(), len(inp) // cfg.batch_size, direction, C.byref(c), out.ptr()), "ecntt")

    def ecntt_inplace(self, inout, direction, cfg):
        self.ecntt(inout, direction, cfg, inout)


class NTTConfig:
    """ntt/mod.rs:77-111 for a field of `words` u32 words per element"""

    def __init__(self, field):
        self.field = field
        self.stream_handle = None
        self.coset_gen = field.one()
        self.batch_size = 1
        self.columns_batch = False
        self.ordering = kNN
        self.are_inputs_on_device = False
        self.are_outputs_on_device = False
        self.is_async = False
        self.ext = ConfigExtension()

    def clone(self):
        o = NTTConfig.__new__(NTTConfig)
        o.__dict__.update(self.__dict__)
        o.ext = self.ext.clone()
        return o

    def to_c(self):
        W = self.field.W
        cg = np.asarray(self.coset_gen, dtype=np.uint32).reshape(-1)
        if W == 1:
            return ref.NTTConfigU32(self.stream_handle, int(cg[0]), self.batch_size, self.columns_batch, self.ordering,
                                    self.are_inputs_on_device, self.are_outputs_on_device, self.is_async, self.ext.handle)
        assert W == 8
        return ref.NTTConfigU256(self.stream_handle, (C.c_uint32 * 8)(*[int(x) for x in cg]), self.batch_size, self.columns_batch,
                                 self.ordering, self.are_inputs_on_device, self.are_outputs_on_device, self.is_async, self.ext.handle)


class NTTInitDomainConfig:
    def __init__(self):
        self.stream_handle = None
        self.is_async = False
        self.ext = ConfigExtension()

    def to_c(self):
        return ref.NTTInitDomainConfig(self.stream_handle, self.is_async, self.ext.handle)


class ScalarField:
    """an IntegerRing + NTTDomain + NTT + GenerateRandom + MontgomeryConvertible + MatrixOps type of the wrapper"""

    def __init__(self, name, words, lib=None):
        self.name, self.W = name, words
        self.lib = lib if lib is not None else ref._load(name)

    def _fn(self, sym, argtypes):
        fn = getattr(self.lib, sym)
        fn.argtypes = argtypes
        return fn

    def one(self):
        o = np.zeros(self.W, dtype=np.uint32)
        o[0] = 1
        return o

    def zero(self, n):
        return np.zeros((n, self.W), dtype=np.uint32)

    def ones(self, n):
        o = self.zero(n)
        o[:, 0] = 1
        return o

    def generate_random(self, n):
        out = np.zeros((n, self.W), dtype=np.uint32)
        self._fn(f"{self.name}_generate_random", [C.c_void_p, C.c_size_t])(out.ctypes.data, n)
        return out

    def get_root_of_unity(self, max_size):
        r = np.zeros(self.W, dtype=np.uint32)
        wrap(self._fn(f"{self.name}_get_root_of_unity", [C.c_uint64, C.c_void_p])(max_size, r.ctypes.data), "get_root_of_unity")
        return r

    def initialize_domain(self, root, config):
        c = config.to_c()
        r = np.ascontiguousarray(root, dtype=np.uint32)
        wrap(self._fn(f"{self.name}_ntt_init_domain", [C.c_void_p, C.c_void_p])(r.ctypes.data, C.byref(c)), "initialize_domain")

    def release_domain(self):
        wrap(getattr(self.lib, f"{self.name}_ntt_release_domain")(), "release_domain")

    def _convert_montgomery(self, values, stream, is_into):  # traits.rs:48-85
        assert not values.is_on_device() or values.is_on_active_device(), "input not allocated on the active device"
        cfg = ref.VecOpsConfig(stream.handle, values.is_on_device(), False, values.is_on_device(), not stream.is_null(), 1, False, None)
        e = ConfigExtension()
        cfg.ext = e.handle
        fn = self._fn(f"{self.name}_scalar_convert_montgomery", [C.c_void_p, C.c_uint64, C.c_bool, C.c_void_p, C.c_void_p])
        rc = fn(values.ptr(), len(values), is_into, C.byref(cfg), values.ptr())
        e.drop()
        wrap(rc, "convert_montgomery")

    def to_mont(self, values, stream):
        self._convert_montgomery(values, stream, True)

    def from_mont(self, values, stream):
        self._convert_montgomery(values, stream, False)

    def ntt(self, inp, direction, cfg, out):
        """NTT::ntt (ntt/mod.rs:153-189) + ntt_unchecked (:302-318)"""
        assert len(inp) == len(out), "input and output lengths do not match"
        assert not inp.is_on_device() or inp.is_on_active_device()
        assert not out.is_on_device() or out.is_on_active_device()
        local = cfg.clone()
        local.are_inputs_on_device = inp.is_on_device()
        local.are_outputs_on_device = out.is_on_device()
        self._ntt_ffi(inp, direction, local, out)

    def ntt_inplace(self, inout, direction, cfg):
        """NTT::ntt_inplace (ntt/mod.rs:191-201) + ntt_inplace_unchecked (:320-336)"""
        local = cfg.clone()
        local.are_inputs_on_device = inout.is_on_device()
        local.are_outputs_on_device = inout.is_on_device()
        self._ntt_ffi(inout, direction, local, inout)

    def _ntt_ffi(self, inp, direction, local, out):
        c = local.to_c()
        fn = self._fn(f"{self.name}_ntt", [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p])
        rc = fn(inp.ptr(), len(inp) // local.batch_size, direction, C.byref(c), out.ptr())
        local.ext.drop()
        wrap(rc, "ntt")

    def matrix_transpose(self, inp, nof_rows, nof_cols, out):
        """matrix_ops/mod.rs matrix_transpose with VecOpsConfig::default()"""
        cfg = ref.VecOpsConfig(None, inp.is_on_device(), False, out.is_on_device(), False, 1, False, None)
        e = ConfigExtension()
        cfg.ext = e.handle
        fn = self._fn(f"{self.name}_matrix_transpose", [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p])
        rc = fn(inp.ptr(), nof_rows, nof_cols, C.byref(cfg), out.ptr())
        e.drop()
        wrap(rc, "matrix_transpose")
