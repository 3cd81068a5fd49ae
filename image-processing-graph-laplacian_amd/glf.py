"""ctypes binding of libglf.so -- the C-ABI declared in include/glf.h.

Python is plumbing only: torch tensors provide device memory and the current
HIP stream, torch.distributed (backend "nccl" = RCCL) provides the collectives
plugged into glf_comm. All compute happens in the HIP library; if libglf.so is
missing this module raises at import time (there is no CPU fallback).

Function names mirror the reference's stage functions (hpc/*.h) exactly as the
C-ABI does: ComputeAffinityMatrices, ComputeLaplacianMatrix,
InversePowerIteration, OrthonormaliseVecs, Nystroem, Permutation,
ComputeResultFromLaplacian, plus image_processing for the whole path.
"""
import ctypes as C
import os

import numpy as np
# torch bundles its own libamdhip64.so.7; it must be loaded BEFORE libglf.so so that the
# process ends up with one HIP runtime (the dynamic linker de-duplicates by SONAME and the
# first one wins). Loading libglf.so first pairs torch with /opt/rocm's runtime and
# hipGetDeviceCount then fails inside this library.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GLF_LIBRARY", os.path.join(_HERE, "libglf.so"))  # override: profiling builds only
if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libglf.so not found at %s: build it with `make` (or __graft_entry__.build()); "
        "there is no CPU fallback for the HIP path" % LIB_PATH)
_lib = C.CDLL(LIB_PATH)

OK = 0
ERR_INVALID, ERR_NOMEM, ERR_HIP, ERR_NODEVICE, ERR_COMM, ERR_NOCONV, ERR_IO, ERR_UNSUPPORTED = range(-1, -9, -1)
MAT_DENSE, MAT_DIAG, MAT_KERNEL_B = 0, 1, 2
ROWS_NA, ROWS_SAMPLE_FIRST, ROWS_RASTER = 0, 1, 2
KERNEL_BILATERAL, KERNEL_PHOTOMETRIC, KERNEL_SPATIAL, KERNEL_NLM = 0, 1, 2, 3
CONTRACT_F32_MFMA, CONTRACT_F16_SPLIT = 1, 2
FILTER_REFERENCE, FILTER_POC, FILTER_SMOOTH, FILTER_SHARPEN = 0, 1, 2, 3
SAMPLING_UNIFORM, SAMPLING_RANDOM = 0, 1
MULTI_RCCL, MULTI_LOOPBACK = 0, 1
RCCL_ID_BYTES = 128

# every symbol include/glf.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "glf_strerror", "glf_ctx_create", "glf_ctx_destroy", "glf_ctx_synchronize", "glf_ctx_last_error",
    "glf_ctx_device_info", "glf_ctx_set_tuning", "glf_ctx_set_comm", "glf_rccl_unique_id", "glf_ctx_set_comm_rccl", "glf_ctx_comm_info", "glf_ctx_comm_counters", "glf_multi_create", "glf_multi_destroy",
    "glf_multi_size", "glf_multi_ctx", "glf_multi_last_error", "glf_multi_image_processing", "glf_shard_rows", "glf_ctx_set_contraction", "glf_malloc", "glf_free", "glf_memcpy_h2d", "glf_memcpy_d2h",
    "glf_memset", "glf_mat_create_dense", "glf_mat_create_diag", "glf_mat_destroy", "glf_mat_get_column", "glf_Sampling",
    "glf_host_free", "glf_random_vectors", "glf_synth_image", "glf_RandomSampling", "glf_ComputeAffinityMatrices",
    "glf_ComputeLaplacianMatrix", "glf_InversePowerIteration", "glf_OrthonormaliseVecs", "glf_NormaliseVecs",
    "glf_InverseDiagMat", "glf_Nystroem", "glf_Permutation", "glf_ComputeResultFromLaplacian", "glf_Sinkhorn", "glf_SinkhornRows", "glf_Orthogonalisation",
    "glf_options_default", "glf_image_processing", "glf_image_processing_capture", "glf_ctx_debug_violations", "glf_ctx_cached_bytes", "glf_image_processing_batch", "glf_EntireComputation", "glf_read_png", "glf_write_png", "glf_read_png_rgb", "glf_write_png_rgb",
]


class Mat(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("row_order", C.c_int32), ("rows", C.c_int64), ("cols", C.c_int64),
        ("ld", C.c_int64), ("data", C.c_void_p), ("owns_data", C.c_int32),
        ("img", C.c_void_p), ("samples", C.c_void_p), ("mask", C.c_void_p), ("idx", C.c_void_p),
        ("width", C.c_int32), ("height", C.c_int32), ("p", C.c_uint32), ("scale", C.c_float),
        ("h_loc", C.c_float), ("h_val", C.c_float), ("kernel", C.c_int32), ("degree", C.c_void_p),
        ("owns_desc", C.c_int32),
    ]


class EigStats(C.Structure):
    _fields_ = [("outer_its", C.c_int32), ("inner_its_total", C.c_int32), ("residual", C.c_double),
                ("matvecs", C.c_int32), ("matvec_ms", C.c_float), ("matvec_bytes", C.c_double),
                ("narrow_sweeps", C.c_int32), ("reserved", C.c_int32)]


class Options(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("num_samples", C.c_uint32), ("sample_frac", C.c_double),
        ("num_eigvals", C.c_uint32), ("opti_gs", C.c_int32), ("epsilon", C.c_double),
        ("inner_rtol", C.c_double), ("max_outer", C.c_int32), ("seed", C.c_uint64), ("gain", C.c_float),
        ("h_loc", C.c_float), ("h_val", C.c_float), ("kernel", C.c_int32), ("filter_pow", C.c_int32),
        ("filter_mode", C.c_int32), ("skip_exact_zeros", C.c_int32), ("filter_beta", C.c_float),
        ("sampling", C.c_int32), ("reserved_", C.c_uint32), ("sampling_seed", C.c_uint64),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("p", C.c_uint32), ("m", C.c_uint32), ("alpha", C.c_double), ("eig", EigStats),
        ("ms_affinity", C.c_float), ("ms_laplacian", C.c_float), ("ms_eigen", C.c_float),
        ("ms_nystroem", C.c_float), ("ms_filter", C.c_float), ("ms_total", C.c_float),
        ("nystroem_launches", C.c_int32), ("nystroem_kernel_ms", C.c_float),
        ("row0", C.c_int32), ("row1", C.c_int32), ("contraction", C.c_int32), ("skip_exact_zeros", C.c_int32),
        ("nystroem_evaluated", C.c_double), ("degree_evaluated", C.c_double),
        ("nystroem_mfma_flops", C.c_double), ("nystroem_path", C.c_int32), ("matvec_path", C.c_int32),
        ("nystroem_rowpass_launches", C.c_int32), ("nystroem_rowpass_ms", C.c_float), ("nystroem_rowpass_flops", C.c_double),
        ("nystroem_colpass_launches", C.c_int32), ("nystroem_colpass_ms", C.c_float), ("nystroem_colpass_flops", C.c_double),
        ("rank_terms", C.c_int32), ("filter_fused", C.c_int32), ("eigen_sharded", C.c_int32), ("reserved_", C.c_int32),
    ]


class Capture(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("ld", C.c_uint32), ("d_phi_A", C.c_void_p), ("phi_A_floats", C.c_size_t),
                ("d_phi", C.c_void_p), ("phi_floats", C.c_size_t), ("h_c", C.c_void_p), ("h_degree", C.c_void_p),
                ("d_corr", C.c_void_p), ("corr_floats", C.c_size_t)]


ALLREDUCE_F32 = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)
ALLREDUCE_F64 = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)
ALLGATHER_F32 = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)


class Comm(C.Structure):
    _fields_ = [("rank", C.c_int), ("size", C.c_int), ("allreduce_sum_f32", ALLREDUCE_F32),
                ("allreduce_sum_f64", ALLREDUCE_F64), ("allgather_f32", ALLGATHER_F32), ("user", C.c_void_p)]


_lib.glf_strerror.restype = C.c_char_p
_lib.glf_ctx_last_error.restype = C.c_char_p
_lib.glf_ctx_last_error.argtypes = [C.c_void_p]
_lib.glf_host_free.restype = None
_lib.glf_multi_ctx.restype = C.c_void_p
_lib.glf_multi_last_error.restype = C.c_char_p
_lib.glf_multi_last_error.argtypes = [C.c_void_p]
_lib.glf_options_default.restype = None
_lib.glf_ctx_cached_bytes.restype = C.c_size_t
_lib.glf_ctx_cached_bytes.argtypes = [C.c_void_p]


class GlfError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        msg = _lib.glf_strerror(status).decode()
        super().__init__("glf status %d (%s)%s" % (status, msg, (": " + detail) if detail else ""))


def default_options(**kw):
    opt = Options()
    _lib.glf_options_default(C.byref(opt))
    for k, v in kw.items():
        if not hasattr(opt, k):
            raise AttributeError("glf_options has no field %r" % k)
        setattr(opt, k, v)
    return opt


# ---- host-side stages ---------------------------------------------------------------

def Sampling(width, height, sample_size):
    """hpc/sampling.c:6-33 -> (realised count, uint32 indices)."""
    n = C.c_uint(sample_size)
    ptr = C.POINTER(C.c_uint)()
    rc = _lib.glf_Sampling(C.c_int(width), C.c_int(height), C.byref(n), C.byref(ptr))
    if rc != OK:
        raise GlfError(rc, "Sampling(%d, %d, %d)" % (width, height, sample_size))
    idx = np.ctypeslib.as_array(ptr, shape=(max(n.value, 1),))[:n.value].astype(np.uint32).copy()
    _lib.glf_host_free(ptr)
    return idx


def RandomSampling(width, height, sample_size, seed=1):
    """python/sampling/random.py:8-16 -> sample_size distinct pixel indices, ascending (the library's own generator)."""
    n = C.c_uint(sample_size)
    ptr = C.POINTER(C.c_uint)()
    rc = _lib.glf_RandomSampling(C.c_int(width), C.c_int(height), C.byref(n), C.byref(ptr), C.c_uint64(seed))
    if rc != OK:
        raise GlfError(rc, "RandomSampling(%d, %d, %d)" % (width, height, sample_size))
    idx = np.ctypeslib.as_array(ptr, shape=(max(n.value, 1),))[:n.value].astype(np.uint32).copy()
    _lib.glf_host_free(ptr)
    return idx


def random_vectors(p, m, seed=1):
    X0 = np.empty((m, p), dtype=np.float64)
    rc = _lib.glf_random_vectors(X0.ctypes.data_as(C.c_void_p), C.c_uint(p), C.c_uint(m), C.c_uint64(seed))
    if rc != OK:
        raise GlfError(rc)
    return X0


def synth_image(width, height, seed=0):
    out = np.empty((height, width), dtype=np.uint8)
    rc = _lib.glf_synth_image(out.ctypes.data_as(C.c_void_p), C.c_int(width), C.c_int(height), C.c_uint64(seed))
    if rc != OK:
        raise GlfError(rc)
    return out


def read_png(path):
    rows = C.POINTER(C.POINTER(C.c_uint8))()
    w, h = C.c_int(), C.c_int()
    rc = _lib.glf_read_png(path.encode(), C.byref(rows), C.byref(w), C.byref(h))
    if rc != 0:
        raise GlfError(ERR_IO, path)
    img = np.empty((h.value, w.value), dtype=np.uint8)
    for r in range(h.value):
        img[r] = np.ctypeslib.as_array(rows[r], shape=(w.value,))
        _lib.glf_host_free(rows[r])
    _lib.glf_host_free(rows)
    return img


def read_png_rgb(path):
    rows = C.POINTER(C.POINTER(C.c_uint8))()
    w, h = C.c_int(), C.c_int()
    rc = _lib.glf_read_png_rgb(path.encode(), C.byref(rows), C.byref(w), C.byref(h))
    if rc != 0:
        raise GlfError(ERR_IO, path)
    img = np.empty((h.value, w.value, 3), dtype=np.uint8)
    for r in range(h.value):
        img[r] = np.ctypeslib.as_array(rows[r], shape=(w.value * 3,)).reshape(w.value, 3)
        _lib.glf_host_free(rows[r])
    _lib.glf_host_free(rows)
    return img


def write_png(path, img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    rowptr = (C.POINTER(C.c_uint8) * h)()
    for r in range(h):
        rowptr[r] = C.cast(img.ctypes.data + r * w, C.POINTER(C.c_uint8))
    rc = _lib.glf_write_png(path.encode(), rowptr, C.c_uint(w), C.c_uint(h))
    if rc != 0:
        raise GlfError(ERR_IO, path)


def shard_rows(height, rank, size):
    """Pixel rows [row0, row1) owned by `rank` (glf_shard_rows; used by glf_image_processing)."""
    r0, r1 = C.c_int(), C.c_int()
    rc = _lib.glf_shard_rows(C.c_int(height), C.c_int(rank), C.c_int(size), C.byref(r0), C.byref(r1))
    if rc != OK:
        raise GlfError(rc, "shard_rows(%d, %d, %d)" % (height, rank, size))
    return r0.value, r1.value


def device_tensor_from_ptr(ptr, count, dtype, device):
    """Zero-copy torch view of `count` elements at device address `ptr`."""
    iface = {"shape": (count,), "typestr": "<f8" if dtype == torch.float64 else "<f4",
             "data": (ptr, False), "version": 2, "strides": None}

    class _Holder:
        __cuda_array_interface__ = iface
    return torch.as_tensor(_Holder(), device=device)


def make_comm(rank, size, allreduce, allgather=None):
    """glf_comm whose callbacks call allreduce(ptr, count, is_f64) and, if given,
    allgather(ptr, count_per_rank) (both in place). Exceptions become a non-zero status (they cannot
    cross the C boundary). Keep the returned struct alive for as long as the context uses it."""
    def wrap(fn, *extra):
        def cb(user, ptr, count):
            try:
                fn(ptr, count, *extra)
                return 0
            except Exception as exc:  # noqa: BLE001
                print("glf collective callback failed:", repr(exc))
                return 1
        return cb
    ag = ALLGATHER_F32(wrap(allgather)) if allgather is not None else ALLGATHER_F32()
    return Comm(rank, size, ALLREDUCE_F32(wrap(allreduce, False)), ALLREDUCE_F64(wrap(allreduce, True)), ag, None)


def rccl_unique_id():
    """ncclGetUniqueId through the library (one rank calls this, every rank passes the bytes to Context.set_comm_rccl)."""
    buf = C.create_string_buffer(RCCL_ID_BYTES)
    rc = _lib.glf_rccl_unique_id(buf, C.c_size_t(RCCL_ID_BYTES))
    if rc != OK:
        raise GlfError(rc, "glf_rccl_unique_id")
    return buf.raw


class Multi:
    """glf_multi: ONE process driving n GPU ranks (one context + one host thread each), the C host's -ngpu N.
    backend MULTI_RCCL: ncclCommInitAll over distinct devices; MULTI_LOOPBACK: host-staged collectives, ranks may share a
    device (tests on a one-GPU box)."""

    def __init__(self, n, devices=None, backend=MULTI_LOOPBACK):
        self._w = C.c_void_p()
        devs = (C.c_int * n)(*devices) if devices is not None else None
        rc = _lib.glf_multi_create(C.byref(self._w), C.c_int(n), devs, C.c_int(backend))
        if rc != OK:
            raise GlfError(rc, "glf_multi_create(%d, backend %d)" % (n, backend))
        self.n = n

    def close(self):
        if self._w:
            _lib.glf_multi_destroy(self._w)
            self._w = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_contraction(self, mode):
        for r in range(self.n):
            rc = _lib.glf_ctx_set_contraction(C.c_void_p(_lib.glf_multi_ctx(self._w, C.c_int(r))), C.c_int(mode))
            if rc != OK:
                raise GlfError(rc)

    def image_processing(self, img, opt=None, want_float=False):
        """Host image in, host image out (glf_multi_image_processing): (out u8 [H, W], zf f32 [H, W] or None, per-rank infos)."""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        opt = opt or default_options()
        out = np.zeros((h, w), dtype=np.uint8)
        zf = np.zeros((h, w), dtype=np.float32) if want_float else None
        lam = np.zeros(max(1, int(Sampling(w, h, int(opt.num_samples) if opt.num_samples else int(h * w * opt.sample_frac)).size)),
                       dtype=np.float64)
        stats = (Stats * self.n)()
        rc = _lib.glf_multi_image_processing(self._w, C.byref(opt), img.ctypes.data_as(C.c_void_p), C.c_int(w), C.c_int(h),
                                             out.ctypes.data_as(C.c_void_p), zf.ctypes.data_as(C.c_void_p) if want_float else None,
                                             lam.ctypes.data_as(C.c_void_p), stats)
        if rc != OK:
            raise GlfError(rc, "glf_multi_image_processing: " + _lib.glf_multi_last_error(self._w).decode())
        infos = [dict(p=s.p, m=s.m, alpha=s.alpha, outer_its=s.eig.outer_its, inner_its_total=s.eig.inner_its_total,
                      residual=s.eig.residual, row0=s.row0, row1=s.row1, ms_total=s.ms_total, ms_eigen=s.ms_eigen,
                      ms_nystroem=s.ms_nystroem, ms_affinity=s.ms_affinity, ms_laplacian=s.ms_laplacian, ms_filter=s.ms_filter,
                      matvecs=s.eig.matvecs, nystroem_path=s.nystroem_path,
                      matvec_path=s.matvec_path, filter_fused=s.filter_fused, eigen_sharded=s.eigen_sharded, eigvals=lam[:s.m].copy()) for s in stats]
        return out, zf, infos

    def comm_counters(self, rank=0, reset=True):
        """Collectives rank `rank` issued since the last reset: dict(allreduce_calls, allreduce_bytes, allgather_calls, allgather_bytes)."""
        return _comm_counters(C.c_void_p(_lib.glf_multi_ctx(self._w, C.c_int(rank))), reset)

    def comm_info(self, rank=0):
        return _comm_info(C.c_void_p(_lib.glf_multi_ctx(self._w, C.c_int(rank))))

    def set_tuning(self, **kw):
        for r in range(self.n):
            for k, v in kw.items():
                rc = _lib.glf_ctx_set_tuning(C.c_void_p(_lib.glf_multi_ctx(self._w, C.c_int(r))), k.encode(), None if v is None else str(v).encode())
                if rc != OK:
                    raise GlfError(rc, "set_tuning(%s=%r)" % (k, v))


def _comm_counters(ctx_ptr, reset=True):
    out = (C.c_ulonglong * 4)()
    rc = _lib.glf_ctx_comm_counters(ctx_ptr, out, C.c_int(1 if reset else 0))
    if rc != OK:
        raise GlfError(rc, "glf_ctx_comm_counters")
    return dict(allreduce_calls=int(out[0]), allreduce_bytes=int(out[1]), allgather_calls=int(out[2]), allgather_bytes=int(out[3]))


def _comm_info(ctx_ptr):
    info = (C.c_int * 4)()
    rc = _lib.glf_ctx_comm_info(ctx_ptr, info)
    if rc != OK:
        raise GlfError(rc, "glf_ctx_comm_info")
    return dict(rank=int(info[0]), size=int(info[1]), backend={0: "none", 1: "rccl", 2: "loopback"}.get(int(info[2]), "?"),
                rccl_ranks=int(info[3]))


# ---- device context --------------------------------------------------------------------

class Context:
    """One glf_ctx on one GPU. The library launches on a dedicated torch stream (self.stream); the
    collective callbacks run under that stream too, so RCCL / copies are ordered with the kernels.
    (torch's default stream has handle 0, which the C-ABI reads as "create your own stream" -- a
    private stream torch knows nothing about would race with the callbacks.)"""

    def __init__(self, device=0):
        import torch
        self.torch = torch
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.stream = torch.cuda.Stream(device=self.device)
        assert self.stream.cuda_stream != 0
        self._ctx = C.c_void_p()
        rc = _lib.glf_ctx_create(C.byref(self._ctx), C.c_int(device), C.c_void_p(self.stream.cuda_stream))
        if rc != OK:
            raise GlfError(rc, "glf_ctx_create(device=%d)" % device)
        self._comm_keepalive = None

    def close(self):
        if self._ctx:
            _lib.glf_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc, what=""):
        if rc != OK:
            raise GlfError(rc, (what + " " if what else "") + _lib.glf_ctx_last_error(self._ctx).decode())

    def set_contraction(self, mode):
        """CONTRACT_F32_MFMA or CONTRACT_F16_SPLIT (see include/glf.h)."""
        self._check(_lib.glf_ctx_set_contraction(self._ctx, C.c_int(mode)))

    def synchronize(self):
        self._check(_lib.glf_ctx_synchronize(self._ctx))

    TUNING_KEYS = ("NYS_PATH", "DEG_PATH", "MV_PATH", "ROWPASS", "ROWPASS_OP", "SWEEP_COLPASS", "COLPASS", "NYS_NO_LUT", "NO_ECR", "NO_NARROW", "NO_FUSED_FILTER", "EIG_SHARD", "ZMFMA_GROUPS", "GS", "RESIDUAL", "VERBOSE")

    def set_tuning(self, **kw):
        """glf_ctx_set_tuning: e.g. set_tuning(NYS_PATH="grid", MV_PATH="dense"); None / "" / "auto" = the default choice."""
        for k, v in kw.items():
            rc = _lib.glf_ctx_set_tuning(self._ctx, k.encode(), None if v is None else str(v).encode())
            if rc != OK:
                raise GlfError(rc, "set_tuning(%s=%r)" % (k, v))

    def reset_tuning(self):
        self.set_tuning(**{k: None for k in self.TUNING_KEYS})

    def debug_violations(self):
        """Guard zones of work buffers found overwritten (debug pool, GLF_POOL_DEBUG=1 at creation); -1 otherwise."""
        return int(_lib.glf_ctx_debug_violations(self._ctx))

    def cached_bytes(self):
        """Bytes of work buffers the context keeps cached and unused (glf_ctx_cached_bytes)."""
        return int(_lib.glf_ctx_cached_bytes(self._ctx))

    def device_info(self):
        name = C.create_string_buffer(256)
        cus, mem = C.c_int(), C.c_size_t()
        self._check(_lib.glf_ctx_device_info(self._ctx, name, C.c_size_t(256), C.byref(cus), C.byref(mem)))
        return dict(name=name.value.decode(), num_cus=cus.value, total_mem=mem.value)

    # -- collectives ---------------------------------------------------------------------------
    def set_comm_rccl(self, rank, size, unique_id, force=False):
        """The library's own RCCL collectives on this context's stream (glf_ctx_set_comm_rccl: ncclCommInitRank, collective
        over all `size` ranks). unique_id: the bytes of rccl_unique_id() from one rank."""
        buf = C.create_string_buffer(bytes(unique_id), RCCL_ID_BYTES)
        self._comm_keepalive = None
        self._native_rank = (rank, size)
        self._check(_lib.glf_ctx_set_comm_rccl(self._ctx, C.c_int(rank), C.c_int(size), buf, C.c_size_t(RCCL_ID_BYTES),
                                               C.c_int(1 if force else 0)), "set_comm_rccl")

    # -- the PoC's balancing steps ------------------------------------------------
    def Sinkhorn(self, phi, Pi, iterations=100, rows=None):
        """sinkhorn(phi, Pi), python/image_processing.py:90-107. phi: dense N x m Mat, Pi: diagonal Mat. Returns (r, c) as numpy
        f64 [N]; with rows = n also W_AB[:n] = diag(r) K diag(c) rows as numpy [n, N] (the PoC's [W_A W_B] for n = m)."""
        torch = self.torch
        N = int(phi.rows)
        r = torch.empty(N, dtype=torch.float64, device=self.device)
        c = torch.empty(N, dtype=torch.float64, device=self.device)
        self._check(_lib.glf_Sinkhorn(self._ctx, C.byref(phi), C.byref(Pi), C.c_int(iterations), C.c_void_p(r.data_ptr()),
                                      C.c_void_p(c.data_ptr())), "Sinkhorn")
        if rows is None:
            return r.cpu().numpy(), c.cpu().numpy()
        out = torch.empty((rows, N), dtype=torch.float64, device=self.device)
        self._check(_lib.glf_SinkhornRows(self._ctx, C.byref(phi), C.byref(Pi), C.c_void_p(r.data_ptr()), C.c_void_p(c.data_ptr()),
                                          C.c_int64(0), C.c_int(rows), C.c_void_p(out.data_ptr())), "SinkhornRows")
        return r.cpu().numpy(), c.cpu().numpy(), out.cpu().numpy()

    def Orthogonalisation(self, A, B):
        """orthogonalisation(A, B), python/image_processing.py:110-127. A: numpy n x n, B: numpy n x q -> (V [(n + q), n], Pi [n])."""
        torch = self.torch
        A = np.ascontiguousarray(A, dtype=np.float64)
        B = np.ascontiguousarray(B, dtype=np.float64)
        n, q = A.shape[0], B.shape[1]
        dA, dB = torch.from_numpy(A).to(self.device), torch.from_numpy(B).to(self.device)
        dV = torch.empty((n + q, n), dtype=torch.float64, device=self.device)
        Pi = np.zeros(n, dtype=np.float64)
        self._check(_lib.glf_Orthogonalisation(self._ctx, C.c_void_p(dA.data_ptr()), C.c_int(n), C.c_void_p(dB.data_ptr()), C.c_int(q),
                                               C.c_void_p(dV.data_ptr()), Pi.ctypes.data_as(C.c_void_p)), "Orthogonalisation")
        return dV.cpu().numpy(), Pi

    def comm_info(self):
        """What the context's communicator reports: dict(rank, size, backend, rccl_ranks = ncclCommCount)."""
        return _comm_info(self._ctx)

    def comm_counters(self, reset=True):
        """Collectives this rank issued through the library's own communicator since the last reset."""
        return _comm_counters(self._ctx, reset)

    def set_comm_torch(self, group=None, shard_eigensolve=True, force=False):
        """Plug torch.distributed all-reduces into glf_comm: RCCL on the device buffers in place
        (backend "nccl"), or staged through host memory for a gloo group (CPU rehearsal of the
        N > 1 path, several ranks sharing one GPU in tests)."""
        import torch.distributed as dist
        torch = self.torch
        size, rank = dist.get_world_size(group), dist.get_rank(group)
        if size == 1 and not force:   # force: keep the callbacks on a one-rank group (tests the device collectives)
            self._check(_lib.glf_ctx_set_comm(self._ctx, None))
            return
        dev = self.device
        on_device = dist.get_backend(group) == "nccl"

        stream = self.stream

        def allreduce(ptr, count, is_f64):
            with torch.cuda.stream(stream):   # ordered after the library's kernels, before its next ones
                t = device_tensor_from_ptr(ptr, count, torch.float64 if is_f64 else torch.float32, dev)
                if on_device:
                    dist.all_reduce(t, group=group)
                else:
                    h = t.cpu()
                    dist.all_reduce(h, group=group)
                    t.copy_(h)
                    stream.synchronize()      # h is pageable host memory

        def allgather(ptr, count_per_rank):
            with torch.cuda.stream(stream):
                full = device_tensor_from_ptr(ptr, count_per_rank * size, torch.float32, dev)
                mine = full[rank * count_per_rank:(rank + 1) * count_per_rank]
                if on_device:
                    dist.all_gather_into_tensor(full, mine.clone(), group=group)
                else:
                    parts = [torch.empty(count_per_rank, dtype=torch.float32) for _ in range(size)]
                    dist.all_gather(parts, mine.cpu(), group=group)
                    full.copy_(torch.cat(parts))
                    stream.synchronize()

        self._comm_keepalive = make_comm(rank, size, allreduce, allgather if shard_eigensolve else None)
        self._check(_lib.glf_ctx_set_comm(self._ctx, C.byref(self._comm_keepalive)))

    # -- helpers ---------------------------------------------------------------------------
    def to_device(self, img):
        with self.torch.cuda.stream(self.stream):
            t = self.torch.from_numpy(np.ascontiguousarray(img, dtype=np.uint8)).to(self.device)
        self.stream.synchronize()
        return t

    def mat_to_numpy(self, mat):
        """Dense / diagonal glf_mat -> numpy (rows x cols), padding stripped."""
        if mat.kind == MAT_DIAG:
            out = np.empty(mat.rows, dtype=np.float32)
            self._check(_lib.glf_memcpy_d2h(self._ctx, out.ctypes.data_as(C.c_void_p), C.c_void_p(mat.data),
                                            C.c_size_t(out.nbytes)))
            return out
        assert mat.kind == MAT_DENSE
        full = np.empty((mat.rows, mat.ld), dtype=np.float32)
        self._check(_lib.glf_memcpy_d2h(self._ctx, full.ctypes.data_as(C.c_void_p), C.c_void_p(mat.data),
                                        C.c_size_t(full.nbytes)))
        return full[:, :mat.cols].copy()

    def degree_of(self, K_B):
        out = np.empty(K_B.p, dtype=np.float64)
        self._check(_lib.glf_memcpy_d2h(self._ctx, out.ctypes.data_as(C.c_void_p), C.c_void_p(K_B.degree),
                                        C.c_size_t(out.nbytes)))
        return out

    def dense_from_numpy(self, arr, ld=None, row_order=ROWS_NA):
        """numpy (rows x cols) -> dense glf_mat with ld rounded up to a power of two >= 32."""
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        rows, cols = arr.shape
        if ld is None:
            ld = 32
            while ld < min(cols, 256):
                ld *= 2
            if cols > 256:
                ld = (cols + 255) // 256 * 256     # more than 256 vectors: a multiple of 256 (panels)
        mat = Mat()
        self._check(_lib.glf_mat_create_dense(self._ctx, C.byref(mat), C.c_int64(rows), C.c_int64(cols), C.c_int64(ld)))
        full = np.zeros((rows, ld), dtype=np.float32)
        full[:, :cols] = arr
        self._check(_lib.glf_memcpy_h2d(self._ctx, C.c_void_p(mat.data), full.ctypes.data_as(C.c_void_p),
                                        C.c_size_t(full.nbytes)))
        mat.row_order = row_order
        return mat

    def diag_from_numpy(self, vec):
        vec = np.ascontiguousarray(vec, dtype=np.float32)
        mat = Mat()
        self._check(_lib.glf_mat_create_diag(self._ctx, C.byref(mat), C.c_int64(vec.size)))
        self._check(_lib.glf_memcpy_h2d(self._ctx, C.c_void_p(mat.data), vec.ctypes.data_as(C.c_void_p),
                                        C.c_size_t(vec.nbytes)))
        return mat

    def destroy(self, *mats):
        for m in mats:
            _lib.glf_mat_destroy(self._ctx, C.byref(m))

    # -- stages (names as in hpc/*.h) -------------------------------------------------------
    def ComputeAffinityMatrices(self, d_img, sample_indices, want_KA=True, kernel=KERNEL_BILATERAL,
                                h_loc=40.0, h_val=30.0):
        assert d_img.dtype == self.torch.uint8 and d_img.is_cuda and d_img.dim() == 2 and d_img.is_contiguous()
        h, w = d_img.shape
        idx = np.ascontiguousarray(sample_indices, dtype=np.uint32)
        K_A, K_B = Mat(), Mat()
        self._check(_lib.glf_ComputeAffinityMatrices(
            self._ctx, C.byref(K_A) if want_KA else None, C.byref(K_B), C.c_void_p(d_img.data_ptr()), C.c_int(w),
            C.c_int(h), C.c_uint(idx.size), idx.ctypes.data_as(C.c_void_p), C.c_int(kernel), C.c_float(h_loc),
            C.c_float(h_val)), "ComputeAffinityMatrices")
        K_B._img_keepalive = d_img
        return (K_A if want_KA else None), K_B

    def ComputeLaplacianMatrix(self, K_A, K_B):
        L_A, L_B = Mat(), Mat()
        alpha = C.c_double()
        self._check(_lib.glf_ComputeLaplacianMatrix(self._ctx, C.byref(L_A), C.byref(L_B),
                                                    C.byref(K_A) if K_A is not None else None, C.byref(K_B),
                                                    C.byref(alpha)), "ComputeLaplacianMatrix")
        return L_A, L_B, alpha.value

    def InversePowerIteration(self, A, m, optiGramSchmidt=1, epsilon=0.1, inner_rtol=1e-5, max_outer=100000,
                              X0=None, allow_noconv=False):
        vecs, vals = Mat(), Mat()
        st = EigStats()
        x0p = None
        if X0 is not None:
            X0 = np.ascontiguousarray(X0, dtype=np.float64)
            assert X0.shape == (m, A.rows)
            x0p = X0.ctypes.data_as(C.c_void_p)
        rc = _lib.glf_InversePowerIteration(self._ctx, C.byref(A), C.c_uint(m), C.byref(vecs), C.byref(vals),
                                            C.c_int(optiGramSchmidt), C.c_double(epsilon), C.c_double(inner_rtol),
                                            C.c_int(max_outer), x0p, C.byref(st))
        if not (allow_noconv and rc == ERR_NOCONV):
            self._check(rc, "InversePowerIteration")
        return vecs, vals, dict(outer_its=st.outer_its, inner_its_total=st.inner_its_total, residual=st.residual)

    def OrthonormaliseVecs(self, X):
        norms = np.empty(X.cols, dtype=np.float64)
        self._check(_lib.glf_OrthonormaliseVecs(self._ctx, C.byref(X), norms.ctypes.data_as(C.c_void_p)))
        return norms

    def NormaliseVecs(self, X):
        norms = np.empty(X.cols, dtype=np.float64)
        self._check(_lib.glf_NormaliseVecs(self._ctx, C.byref(X), norms.ctypes.data_as(C.c_void_p)))
        return norms

    def InverseDiagMat(self, x):
        inv = Mat()
        self._check(_lib.glf_InverseDiagMat(self._ctx, C.byref(x), C.byref(inv)))
        return inv

    def Nystroem(self, B, phi_A, Pi_A_Inv):
        phi = Mat()
        self._check(_lib.glf_Nystroem(self._ctx, C.byref(B), C.byref(phi_A), C.byref(Pi_A_Inv), C.byref(phi)), "Nystroem")
        return phi

    def Permutation(self, mat, sample_indices):
        idx = np.ascontiguousarray(sample_indices, dtype=np.uint32)
        out = Mat()
        self._check(_lib.glf_Permutation(self._ctx, C.byref(mat), idx.ctypes.data_as(C.c_void_p), C.c_uint(idx.size),
                                         C.byref(out)), "Permutation")
        return out

    def ComputeResultFromLaplacian(self, d_img, phi, Pi, gain=3.0, want_float=True):
        torch = self.torch
        h, w = d_img.shape
        with torch.cuda.stream(self.stream):
            out = torch.empty((h, w), dtype=torch.uint8, device=self.device)
            zf = torch.empty((h, w), dtype=torch.float32, device=self.device) if want_float else None
        self._check(_lib.glf_ComputeResultFromLaplacian(
            self._ctx, C.c_void_p(d_img.data_ptr()), C.byref(phi), C.byref(Pi), C.c_uint(w), C.c_uint(h),
            C.c_float(gain), C.c_void_p(out.data_ptr()), C.c_void_p(zf.data_ptr()) if want_float else None),
            "ComputeResultFromLaplacian")
        self.stream.synchronize()
        return out, zf

    def EntireComputation(self, d_img, kernel=KERNEL_BILATERAL, h_loc=40.0, h_val=30.0):
        """-no_approx mode (hpc/image_processing.c:155-181): z = clamp(y - L y), full N x N Laplacian."""
        torch = self.torch
        h, w = d_img.shape
        with torch.cuda.stream(self.stream):
            out = torch.empty((h, w), dtype=torch.uint8, device=self.device)
            zf = torch.empty((h, w), dtype=torch.float32, device=self.device)
        alpha = C.c_double()
        self._check(_lib.glf_EntireComputation(self._ctx, C.c_void_p(d_img.data_ptr()), C.c_int(w), C.c_int(h),
                                               C.c_int(kernel), C.c_float(h_loc), C.c_float(h_val),
                                               C.c_void_p(out.data_ptr()), C.c_void_p(zf.data_ptr()), C.byref(alpha)),
                    "EntireComputation")
        self.stream.synchronize()
        return out, zf, alpha.value

    def image_processing(self, d_img, opt=None, want_float=False, out=None, capture=False):
        """Whole approximate path (hpc/image_processing.c:183-277) on a device image tensor.
        capture=True additionally returns the run's by-products in info["capture"] (glf_capture): phi_A [p, ld] and
        phi [rows of this rank * width, ld] as device tensors, c = Phi^T y and the degree vector as numpy arrays."""
        torch = self.torch
        assert d_img.dtype == torch.uint8 and d_img.is_cuda and d_img.dim() == 2 and d_img.is_contiguous()
        h, w = d_img.shape
        opt = opt or default_options()
        with torch.cuda.stream(self.stream):   # the zero fills must be ordered before the library's writes
            if out is None:
                out = torch.zeros((h, w), dtype=torch.uint8, device=self.device)
            zf = torch.zeros((h, w), dtype=torch.float32, device=self.device) if want_float else None
        st = Stats()
        # (the realised sample count sizes the eigenvalue array; hpc/sampling.c's rule, cached per image size and request)
        req = int(opt.num_samples) if opt.num_samples else int(h * w * opt.sample_frac)
        key = (w, h, req, int(getattr(opt, "sampling", 0)), int(getattr(opt, "sampling_seed", 0)))
        cache = self.__dict__.setdefault("_p_real_cache", {})
        if key not in cache:
            cache[key] = int(Sampling(w, h, req).size) if not getattr(opt, "sampling", 0) else max(req, 1) * 2 + 64
        p_real = cache[key]
        lam = np.zeros(max(p_real, 1), dtype=np.float64)       # m <= p - 1 eigenvalues come back
        cap, keep = None, None
        if capture:
            # the realised sample count (hpc/sampling.c rewrites the request) and the row stride, known before the call
            p_max = p_real
            m_req = int(opt.num_eigvals) if 0 < opt.num_eigvals < p_max else max(1, p_max - 1)
            ld = 32
            while ld < min(m_req, 256):
                ld *= 2
            if self._comm_keepalive:
                rows = shard_rows(h, self._comm_keepalive.rank, self._comm_keepalive.size)
            elif getattr(self, "_native_rank", None):
                rows = shard_rows(h, *self._native_rank)
            else:
                rows = (0, h)
            npix = (rows[1] - rows[0]) * w
            with torch.cuda.stream(self.stream):
                phi_A = torch.zeros(((p_max + 63) // 64 * 64, ld), dtype=torch.float32, device=self.device)
                phi = torch.zeros((npix, ld), dtype=torch.float32, device=self.device)
                corr = torch.zeros(npix, dtype=torch.float32, device=self.device)
            c_host, deg_host = np.zeros(ld, dtype=np.float64), np.zeros(p_max, dtype=np.float64)
            cap = Capture(C.sizeof(Capture), 0, phi_A.data_ptr(), phi_A.numel(), phi.data_ptr(), phi.numel(),
                          c_host.ctypes.data, deg_host.ctypes.data, corr.data_ptr(), corr.numel())
            keep = (phi_A, phi, c_host, deg_host, corr)
        rc = _lib.glf_image_processing_capture(self._ctx, C.byref(opt), C.c_void_p(d_img.data_ptr()), C.c_int(w), C.c_int(h),
                                               C.c_void_p(out.data_ptr()), C.c_void_p(zf.data_ptr()) if want_float else None,
                                               lam.ctypes.data_as(C.c_void_p), C.byref(st), C.byref(cap) if cap else None)
        self._check(rc, "image_processing")
        self.stream.synchronize()
        info = dict(p=st.p, m=st.m, alpha=st.alpha, outer_its=st.eig.outer_its,
                    inner_its_total=st.eig.inner_its_total, residual=st.eig.residual,
                    ms_affinity=st.ms_affinity, ms_laplacian=st.ms_laplacian, ms_eigen=st.ms_eigen,
                    ms_nystroem=st.ms_nystroem, ms_filter=st.ms_filter, ms_total=st.ms_total,
                    nystroem_kernel_ms=st.nystroem_kernel_ms, nystroem_launches=st.nystroem_launches,
                    row0=st.row0, row1=st.row1, contraction=st.contraction, skip_exact_zeros=st.skip_exact_zeros,
                    nystroem_evaluated=st.nystroem_evaluated, degree_evaluated=st.degree_evaluated,
                    nystroem_mfma_flops=st.nystroem_mfma_flops, nystroem_path=st.nystroem_path, matvec_path=st.matvec_path,
                    nystroem_rowpass_launches=st.nystroem_rowpass_launches, nystroem_rowpass_ms=st.nystroem_rowpass_ms,
                    nystroem_rowpass_flops=st.nystroem_rowpass_flops, nystroem_colpass_launches=st.nystroem_colpass_launches,
                    nystroem_colpass_ms=st.nystroem_colpass_ms, nystroem_colpass_flops=st.nystroem_colpass_flops, rank_terms=st.rank_terms, filter_fused=st.filter_fused, eigen_sharded=st.eigen_sharded,
                    matvecs=st.eig.matvecs, matvec_ms=st.eig.matvec_ms, matvec_bytes=st.eig.matvec_bytes,
                    narrow_sweeps=st.eig.narrow_sweeps,
                    eigvals=lam[:st.m].copy())
        if capture:
            assert cap.ld == keep[0].shape[1], (cap.ld, keep[0].shape)
            info["capture"] = dict(phi_A=keep[0][:st.p], phi=keep[1], c=keep[2][:st.m].copy(), degree=keep[3][:st.p].copy(),
                                   corr=keep[4], ld=int(cap.ld))
        return out, zf, info


def image_processing_batch(contexts, d_imgs, opt=None):
    """Throughput mode (glf_image_processing_batch; BASELINE config 5): d_imgs is a device uint8 tensor
    [tiles, H, W]; the contexts (each on its own stream, none with a comm) filter the tiles concurrently.
    Returns (outputs [tiles, H, W] uint8, list of per-tile dicts p / m / outer_its / ms_total)."""
    ctx0 = contexts[0]
    torch = ctx0.torch
    assert d_imgs.dtype == torch.uint8 and d_imgs.is_cuda and d_imgs.dim() == 3 and d_imgs.is_contiguous()
    t, h, w = d_imgs.shape
    opt = opt or default_options()
    outs = torch.zeros((t, h, w), dtype=torch.uint8, device=ctx0.device)
    torch.cuda.synchronize(ctx0.device)      # inputs and the zero fill are complete before any context's stream starts
    stats = (Stats * max(1, t))()
    handles = (C.c_void_p * len(contexts))(*[c._ctx for c in contexts])
    rc = _lib.glf_image_processing_batch(handles, C.c_int(len(contexts)), C.byref(opt), C.c_void_p(d_imgs.data_ptr()),
                                         C.c_int(w), C.c_int(h), C.c_int(t), C.c_void_p(outs.data_ptr()), None, stats)
    ctx0._check(rc, "image_processing_batch")
    for c in contexts:
        c.stream.synchronize()
    infos = [dict(p=stats[i].p, m=stats[i].m, alpha=stats[i].alpha, outer_its=stats[i].eig.outer_its,
                  ms_total=stats[i].ms_total) for i in range(t)]
    return outs, infos

