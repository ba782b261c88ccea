"""TEST INFRASTRUCTURE ONLY (never imported by icicle_amd/).

ctypes access to the REAL reference: the unmodified ICICLE CPU backend compiled by
oracle/build_ref.sh into oracle/_ref/ (libicicle_device.so, libicicle_curve_<c>.so,
libicicle_field_<f>.so). Used as the parity oracle by tests/, by __graft_entry__.smoke() and as
the `cpu_baseline` ("kind": "reference") leg of bench.py. The libraries travel to the GPU box
with the snapshot; /root/reference itself is only needed to (re)build them.
"""
import ctypes
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(_HERE, "_ref")


CURVE_LIMBS = {"bn254": 8, "bls12_381": 12, "bls12_377": 12, "grumpkin": 8}  # u32 words per base-field element


def available(name: str = "device") -> bool:
    return os.path.exists(os.path.join(REF_DIR, _libname(name)))


def _libname(name):
    if name == "device":
        return "libicicle_device.so"
    if name in CURVE_LIMBS:
        return f"libicicle_curve_{name}.so"
    return f"libicicle_field_{name}.so"


_loaded = {}


def _load(name):
    if name not in _loaded:
        path = os.path.join(REF_DIR, _libname(name))
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path}: run oracle/build_ref.sh where /root/reference exists")
        _loaded[name] = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL if name == "device" else ctypes.RTLD_LOCAL)
    return _loaded[name]


class MSMConfig(ctypes.Structure):  # icicle/include/icicle/msm.h:21-53
    _fields_ = [
        ("stream", ctypes.c_void_p), ("precompute_factor", ctypes.c_int), ("c", ctypes.c_int),
        ("bitsize", ctypes.c_int), ("batch_size", ctypes.c_int),
        ("are_points_shared_in_batch", ctypes.c_bool), ("are_scalars_on_device", ctypes.c_bool),
        ("are_scalars_montgomery_form", ctypes.c_bool), ("are_points_on_device", ctypes.c_bool),
        ("are_points_montgomery_form", ctypes.c_bool), ("are_results_on_device", ctypes.c_bool),
        ("is_async", ctypes.c_bool), ("ext", ctypes.c_void_p),
    ]


class NTTConfigU32(ctypes.Structure):  # icicle/include/icicle/ntt.h:53-64 for a 4-byte scalar_t
    _fields_ = [
        ("stream", ctypes.c_void_p), ("coset_gen", ctypes.c_uint32), ("batch_size", ctypes.c_int),
        ("columns_batch", ctypes.c_bool), ("ordering", ctypes.c_int), ("are_inputs_on_device", ctypes.c_bool),
        ("are_outputs_on_device", ctypes.c_bool), ("is_async", ctypes.c_bool), ("ext", ctypes.c_void_p),
    ]


class NTTConfigU256(ctypes.Structure):  # the same struct for the curves' 32-byte scalar_t (storage<8>, 8-byte aligned)
    _fields_ = [
        ("stream", ctypes.c_void_p), ("coset_gen", ctypes.c_uint32 * 8), ("batch_size", ctypes.c_int),
        ("columns_batch", ctypes.c_bool), ("ordering", ctypes.c_int), ("are_inputs_on_device", ctypes.c_bool),
        ("are_outputs_on_device", ctypes.c_bool), ("is_async", ctypes.c_bool), ("ext", ctypes.c_void_p),
    ]


class NTTConfigU64(ctypes.Structure):  # goldilocks: scalar_t = storage<2>
    _fields_ = [
        ("stream", ctypes.c_void_p), ("coset_gen", ctypes.c_uint32 * 2), ("batch_size", ctypes.c_int),
        ("columns_batch", ctypes.c_bool), ("ordering", ctypes.c_int), ("are_inputs_on_device", ctypes.c_bool),
        ("are_outputs_on_device", ctypes.c_bool), ("is_async", ctypes.c_bool), ("ext", ctypes.c_void_p),
    ]


class NTTInitDomainConfig(ctypes.Structure):
    _fields_ = [("stream", ctypes.c_void_p), ("is_async", ctypes.c_bool), ("ext", ctypes.c_void_p)]


def _p(a: np.ndarray):
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)


class Device(ctypes.Structure):  # icicle/include/icicle/device.h:14-48
    _fields_ = [("type", ctypes.c_char * 64), ("id", ctypes.c_int)]


class RefRuntime:
    """The reference's own runtime C ABI (libicicle_device.so): backend loading, device selection,
    memory. Used by tests/test_gpu_plugin.py to drive the HIP backend the way a reference user would."""

    def __init__(self):
        self.lib = _load("device")
        self.lib.icicle_malloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        self.lib.icicle_free.argtypes = [ctypes.c_void_p]
        for n in ("icicle_copy", "icicle_copy_to_host", "icicle_copy_to_device"):
            getattr(self.lib, n).argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        self.lib.icicle_memset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
        self.lib.icicle_is_host_memory.argtypes = [ctypes.c_void_p]
        self.lib.icicle_is_active_device_memory.argtypes = [ctypes.c_void_p]
        self.lib.icicle_load_backend.argtypes = [ctypes.c_char_p, ctypes.c_bool]
        self.lib.icicle_get_registered_devices.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        self.lib.icicle_create_stream.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.lib.icicle_destroy_stream.argtypes = [ctypes.c_void_p]
        self.lib.icicle_stream_synchronize.argtypes = [ctypes.c_void_p]

    def load_backend(self, path: str) -> int:
        return self.lib.icicle_load_backend(path.encode(), True)

    def registered_devices(self) -> str:
        buf = ctypes.create_string_buffer(256)
        self.lib.icicle_get_registered_devices(buf, 256)
        return buf.value.decode()

    def set_device(self, dtype: str, did: int = 0) -> int:
        d = Device(dtype.encode(), did)
        return self.lib.icicle_set_device(ctypes.byref(d))

    def get_device_count(self) -> int:
        n = ctypes.c_int()
        self.lib.icicle_get_device_count(ctypes.byref(n))
        return n.value

    def malloc(self, nbytes: int):
        p = ctypes.c_void_p()
        rc = self.lib.icicle_malloc(ctypes.byref(p), nbytes)
        return rc, p.value

    def free(self, ptr) -> int:
        return self.lib.icicle_free(ptr)

    def to_device(self, ptr, arr: np.ndarray) -> int:
        return self.lib.icicle_copy_to_device(ptr, arr.ctypes.data, arr.nbytes)

    def to_host(self, arr: np.ndarray, ptr) -> int:
        return self.lib.icicle_copy_to_host(arr.ctypes.data, ptr, arr.nbytes)

    def copy(self, dst, src, nbytes) -> int:
        return self.lib.icicle_copy(dst, src, nbytes)


class VecOpsConfig(ctypes.Structure):  # icicle/include/icicle/vec_ops.h:19-37
    _fields_ = [("stream", ctypes.c_void_p), ("is_a_on_device", ctypes.c_bool), ("is_b_on_device", ctypes.c_bool),
                ("is_result_on_device", ctypes.c_bool), ("is_async", ctypes.c_bool), ("batch_size", ctypes.c_int),
                ("columns_batch", ctypes.c_bool), ("ext", ctypes.c_void_p)]


def ref_convert_montgomery(libname: str, symbol: str, arr: np.ndarray, count: int, to_mont: bool) -> np.ndarray:
    """<prefix>_scalar_convert_montgomery / _affine_ / _projective_convert_montgomery of the reference, on
    the calling thread's active device (host buffers)."""
    lib = _load(libname) if libname in ("bn254", "bls12_381", "babybear", "koalabear") and symbol.find("scalar") < 0 else None
    if lib is None:
        lib = ctypes.CDLL(os.path.join(REF_DIR, f"libicicle_field_{libname}.so")) if "scalar" in symbol else _load(libname)
    cfg = VecOpsConfig(None, False, False, False, False, 1, False, None)
    out = np.zeros_like(arr)
    fn = getattr(lib, symbol)
    fn.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_bool, ctypes.c_void_p, ctypes.c_void_p]
    rc = fn(arr.ctypes.data, count, to_mont, ctypes.byref(cfg), out.ctypes.data)
    assert rc == 0, f"{symbol} rc={rc}"
    return out


class RefCurve:
    """a curve of CURVE_LIMBS through the reference's own C ABI, on its "CPU" device.
    g2=True selects the <curve>_g2_* entry points (G2_ENABLED build): coordinates are Fq2 = 2 base-field elements."""

    def __init__(self, name: str, g2: bool = False):
        _load("device")
        _load(name)  # field lib (dependency of the curve lib, resolved through RPATH=$ORIGIN)
        self.name = name
        self.g2 = g2
        self.sym = f"{name}_g2" if g2 else name
        self.lib = _load(name)
        self.L = CURVE_LIMBS[name] * (2 if g2 else 1)

    def msm(self, scalars: np.ndarray, bases: np.ndarray, batch=1, shared=True, precompute_factor=1, c=0, bitsize=0,
            scalars_mont=False, points_mont=False, n_threads=0):
        """runs on whatever device is active for the calling thread ("CPU" unless RefRuntime.set_device was used)"""
        n = scalars.size // 8 // batch
        cfg = MSMConfig(None, precompute_factor, c, bitsize, batch, shared, False, scalars_mont, False, points_mont,
                        False, False, None)
        if not n_threads:  # (a session-wide default worker count, e.g. tests/conftest.py on many-core hosts)
            n_threads = int(os.environ.get("ICICLE_REF_MSM_THREADS", "0") or 0)
        ext = None
        if n_threads:
            dev = _load("device")
            dev.create_config_extension.restype = ctypes.c_void_p
            ext = dev.create_config_extension()
            dev.config_extension_set_int.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
            dev.config_extension_set_int(ext, b"n_threads", n_threads)
            cfg.ext = ext
        out = np.zeros((batch, 3 * self.L), dtype=np.uint32)
        rc = getattr(self.lib, f"{self.sym}_msm")(_p(scalars), _p(bases), n, ctypes.byref(cfg), _p(out))
        if ext:
            dev.destroy_config_extension.argtypes = [ctypes.c_void_p]
            dev.destroy_config_extension(ext)
        assert rc == 0, f"reference msm failed rc={rc}"
        return out

    def precompute_bases(self, bases: np.ndarray, precompute_factor: int, c=0):
        n = bases.size // (2 * self.L)
        cfg = MSMConfig(None, precompute_factor, c, 0, 1, True, False, False, False, False, False, False, None)
        out = np.zeros((n * precompute_factor, 2 * self.L), dtype=np.uint32)
        rc = getattr(self.lib, f"{self.sym}_msm_precompute_bases")(_p(bases), n, ctypes.byref(cfg), _p(out))
        assert rc == 0
        return out

    def to_affine(self, proj: np.ndarray) -> np.ndarray:
        """projective_t[...] -> affine_t[...] with Projective::to_affine (projective.h:55-59)."""
        proj = np.ascontiguousarray(proj.reshape(-1, 3 * self.L))
        out = np.zeros((proj.shape[0], 2 * self.L), dtype=np.uint32)
        fn = getattr(self.lib, f"{self.sym}_to_affine")
        for i in range(proj.shape[0]):
            fn(ctypes.c_void_p(proj[i].ctypes.data), ctypes.c_void_p(out[i].ctypes.data))
        return out

    def projective_eq(self, a: np.ndarray, b: np.ndarray) -> bool:
        fn = getattr(self.lib, f"{self.sym}_projective_eq")
        fn.restype = ctypes.c_bool
        a = np.ascontiguousarray(a)
        b = np.ascontiguousarray(b)
        return bool(fn(_p(a), _p(b)))

    def is_on_curve(self, a: np.ndarray) -> bool:
        fn = getattr(self.lib, f"{self.sym}_is_on_curve")
        fn.restype = ctypes.c_bool
        a = np.ascontiguousarray(a)
        return bool(fn(_p(a)))

    def ecntt(self, points: np.ndarray, size: int, direction: int, batch=1, columns_batch=False, ordering=0,
              coset_gen=1) -> np.ndarray:
        """<curve>_ecntt (src/ecntt.cpp:7-11) on projective_t[size*batch]; needs RefScalarNttField(name).init_domain"""
        cfg = NTTConfigU256(None, _w8(coset_gen), batch, columns_batch, ordering, False, False, False, None)
        out = np.zeros_like(points)
        fn = getattr(self.lib, f"{self.name}_ecntt")
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        rc = fn(_p(points), size, direction, ctypes.byref(cfg), _p(out))
        assert rc == 0, f"reference ecntt failed rc={rc}"
        return out

    def generate_affine_points(self, n: int) -> np.ndarray:
        """projective_t::rand_host_many(affine_t*, n) (projective.h:43-53): period-100 repetition."""
        out = np.zeros((n, 2 * self.L), dtype=np.uint32)
        getattr(self.lib, f"{self.sym}_generate_affine_points")(_p(out), n)
        return out

    def generate_scalars(self, n: int) -> np.ndarray:
        fl = _load(self.name)
        out = np.zeros((n, 8), dtype=np.uint32)
        # <prefix>_generate_random lives in the field library; the curve lib links it
        flib = ctypes.CDLL(os.path.join(REF_DIR, f"libicicle_field_{self.name}.so"))
        getattr(flib, f"{self.name}_generate_random")(_p(out), n)
        return out

    def scalars_to_montgomery(self, scalars: np.ndarray) -> np.ndarray:
        """host-side x -> x*R mod r using the reference field mul (R = 2^256)."""
        flib = ctypes.CDLL(os.path.join(REF_DIR, f"libicicle_field_{self.name}.so"))
        from . import pyref
        r = pyref.CURVES[self.name].r
        out = np.zeros_like(scalars)
        for i in range(scalars.shape[0]):
            v = sum(int(x) << (32 * k) for k, x in enumerate(scalars[i])) * (1 << 256) % r
            out[i] = [(v >> (32 * k)) & 0xFFFFFFFF for k in range(8)]
        return out


class RefNttField:
    """babybear / koalabear through the reference's C ABI on its "CPU" device."""

    def __init__(self, name: str):
        _load("device")
        self.name = name
        self.lib = _load(name)
        self._domain_log = None

    def get_root_of_unity(self, max_size: int) -> int:
        r = ctypes.c_uint32()
        fn = getattr(self.lib, f"{self.name}_get_root_of_unity")
        fn.argtypes = [ctypes.c_uint64, ctypes.c_void_p]
        rc = fn(max_size, ctypes.byref(r))
        assert rc == 0
        return r.value

    def init_domain(self, root: int):
        cfg = NTTInitDomainConfig(None, False, None)
        r = ctypes.c_uint32(root)
        rc = getattr(self.lib, f"{self.name}_ntt_init_domain")(ctypes.byref(r), ctypes.byref(cfg))
        assert rc == 0

    def release_domain(self):
        rc = getattr(self.lib, f"{self.name}_ntt_release_domain")()
        assert rc == 0

    def get_root_of_unity_from_domain(self, logn: int) -> int:
        r = ctypes.c_uint32()
        fn = getattr(self.lib, f"{self.name}_get_root_of_unity_from_domain")
        fn.argtypes = [ctypes.c_uint64, ctypes.c_void_p]
        rc = fn(logn, ctypes.byref(r))
        assert rc == 0
        return r.value

    def ntt(self, inp: np.ndarray, size: int, direction: int, batch=1, columns_batch=False, ordering=0, coset_gen=1,
            extension=False) -> np.ndarray:
        cfg = NTTConfigU32(None, coset_gen, batch, columns_batch, ordering, False, False, False, None)
        out = np.zeros_like(inp)
        fn = getattr(self.lib, f"{self.name}_extension_ntt" if extension else f"{self.name}_ntt")
        rc = fn(_p(inp), size, direction, ctypes.byref(cfg), _p(out))
        assert rc == 0, f"reference ntt failed rc={rc}"
        return out

    def matrix_transpose(self, inp: np.ndarray, nof_rows: int, nof_cols: int, batch=1, extension=False, inplace=False) -> np.ndarray:
        """<field>_matrix_transpose / _extension_matrix_transpose (icicle/src/matrix_ops.cpp:75-102) on the active device"""
        cfg = VecOpsConfig(None, False, False, False, False, batch, False, None)
        out = inp if inplace else np.zeros_like(inp)
        fn = getattr(self.lib, f"{self.name}_extension_matrix_transpose" if extension else f"{self.name}_matrix_transpose")
        fn.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
        rc = fn(_p(inp), nof_rows, nof_cols, ctypes.byref(cfg), _p(out))
        assert rc == 0, f"reference matrix_transpose failed rc={rc}"
        return out

    def ntt_device(self, d_in, d_out, size: int, direction: int, batch=1, ordering=0, coset_gen=1) -> int:
        """device-resident operands allocated through the reference runtime (icicle_malloc)"""
        cfg = NTTConfigU32(None, coset_gen, batch, False, ordering, True, True, False, None)
        fn = getattr(self.lib, f"{self.name}_ntt")
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        return fn(d_in, size, direction, ctypes.byref(cfg), d_out)


def _w8(x: int):
    return (ctypes.c_uint32 * 8)(*[(x >> (32 * i)) & 0xFFFFFFFF for i in range(8)])


def _i8(w) -> int:
    return sum(int(w[i]) << (32 * i) for i in range(8))


class RefScalarNttField:
    """NTT over a 256-bit field (a curve's scalar field, or stark252) through the reference's C ABI on "CPU"
    (src/ntt.cpp:11-84 compiled with FIELD = the curve's scalar field). Elements: 8 u32 words."""

    def __init__(self, name: str):
        _load("device")
        self.name = name
        self.lib = _load(name)

    def get_root_of_unity(self, max_size: int) -> int:
        w = (ctypes.c_uint32 * 8)()
        fn = getattr(self.lib, f"{self.name}_get_root_of_unity")
        fn.argtypes = [ctypes.c_uint64, ctypes.c_void_p]
        assert fn(max_size, w) == 0
        return _i8(w)

    def init_domain(self, root: int):
        cfg = NTTInitDomainConfig(None, False, None)
        fn = getattr(self.lib, f"{self.name}_ntt_init_domain")
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        assert fn(_w8(root), ctypes.byref(cfg)) == 0

    def release_domain(self):
        assert getattr(self.lib, f"{self.name}_ntt_release_domain")() == 0

    def get_root_of_unity_from_domain(self, logn: int) -> int:
        w = (ctypes.c_uint32 * 8)()
        fn = getattr(self.lib, f"{self.name}_get_root_of_unity_from_domain")
        fn.argtypes = [ctypes.c_uint64, ctypes.c_void_p]
        assert fn(logn, w) == 0
        return _i8(w)

    def ntt(self, inp: np.ndarray, size: int, direction: int, batch=1, columns_batch=False, ordering=0,
            coset_gen=1) -> np.ndarray:
        cfg = NTTConfigU256(None, _w8(coset_gen), batch, columns_batch, ordering, False, False, False, None)
        out = np.zeros_like(inp)
        fn = getattr(self.lib, f"{self.name}_ntt")
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        rc = fn(_p(inp), size, direction, ctypes.byref(cfg), _p(out))
        assert rc == 0, f"reference ntt failed rc={rc}"
        return out


class RefGoldField:
    """goldilocks through the reference's C ABI on "CPU" (src/ntt.cpp compiled with FIELD_ID 1005, EXT_FIELD): elements
    are 2 u32 words, extension elements (quadratic, u^2 = 7) 4; roots and coset generators are Python ints."""

    name = "goldilocks"

    def __init__(self):
        _load("device")
        self.lib = _load(self.name)

    @staticmethod
    def _w2(x: int):
        return (ctypes.c_uint32 * 2)(x & 0xFFFFFFFF, (x >> 32) & 0xFFFFFFFF)

    def get_root_of_unity(self, max_size: int) -> int:
        w = (ctypes.c_uint32 * 2)()
        fn = self.lib.goldilocks_get_root_of_unity
        fn.argtypes = [ctypes.c_uint64, ctypes.c_void_p]
        assert fn(max_size, w) == 0
        return int(w[0]) | (int(w[1]) << 32)

    def init_domain(self, root: int):
        cfg = NTTInitDomainConfig(None, False, None)
        fn = self.lib.goldilocks_ntt_init_domain
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        assert fn(self._w2(root), ctypes.byref(cfg)) == 0

    def release_domain(self):
        assert self.lib.goldilocks_ntt_release_domain() == 0

    def get_root_of_unity_from_domain(self, logn: int) -> int:
        w = (ctypes.c_uint32 * 2)()
        fn = self.lib.goldilocks_get_root_of_unity_from_domain
        fn.argtypes = [ctypes.c_uint64, ctypes.c_void_p]
        assert fn(logn, w) == 0
        return int(w[0]) | (int(w[1]) << 32)

    def ntt(self, inp: np.ndarray, size: int, direction: int, batch=1, columns_batch=False, ordering=0, coset_gen=1,
            extension=False) -> np.ndarray:
        cfg = NTTConfigU64(None, self._w2(coset_gen), batch, columns_batch, ordering, False, False, False, None)
        out = np.zeros_like(inp)
        fn = self.lib.goldilocks_extension_ntt if extension else self.lib.goldilocks_ntt
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        rc = fn(_p(inp), size, direction, ctypes.byref(cfg), _p(out))
        assert rc == 0, f"reference ntt failed rc={rc}"
        return out
