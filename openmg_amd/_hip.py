"""ctypes binding of libopenmg_hip.so (C ABI: include/openmg_hip.h).

There is deliberately NO fallback: if the shared library is missing, or no MI355X is
visible when a compute entry point is called, an exception is raised.
"""
import ctypes
import os

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OMG_LIB_PATH") or os.path.join(_HERE, "lib", "libopenmg_hip.so")   # override: kernel A/B builds

OMG_OK = 0
ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_SINGULAR, ERR_NO_DIAGONAL, ERR_ALLOC, ERR_UNSUPPORTED = range(1, 8)
SMOOTH_GS_LEX, SMOOTH_GS_COLOUR, SMOOTH_JACOBI = 0, 1, 2
PROFILE_CLASSES = 7
PROFILE_NAMES = ("smoother_set_sweep", "residual", "restrict", "prolong_add", "residual_norm", "plane_down", "plane_up")

SMOOTHERS = {
    "gs": SMOOTH_GS_LEX, "lex": SMOOTH_GS_LEX, "gauss-seidel": SMOOTH_GS_LEX, "gaussSeidel": SMOOTH_GS_LEX,
    "colour": SMOOTH_GS_COLOUR, "color": SMOOTH_GS_COLOUR, "rbgs": SMOOTH_GS_COLOUR,
    "red-black": SMOOTH_GS_COLOUR, "multicolour": SMOOTH_GS_COLOUR,
    "jacobi": SMOOTH_JACOBI, "weighted-jacobi": SMOOTH_JACOBI,
}


class HipError(RuntimeError):
    """A libopenmg_hip.so call failed; .code holds the OMG_ERR_* value."""

    def __init__(self, code, message):
        super().__init__("libopenmg_hip: %s (code %d)" % (message, code))
        self.code = code


class CsrView(ctypes.Structure):
    _fields_ = [("n_rows", ctypes.c_int64), ("n_cols", ctypes.c_int64), ("nnz", ctypes.c_int64),
                ("indptr", ctypes.c_void_p), ("indices", ctypes.c_void_p), ("data", ctypes.c_void_p)]


_P = ctypes.c_void_p
_I = ctypes.c_int
_D = ctypes.c_double
_PP = ctypes.POINTER(ctypes.c_void_p)
_CSR = ctypes.POINTER(CsrView)
_I64P = ctypes.POINTER(ctypes.c_int64)
_IP = ctypes.POINTER(ctypes.c_int)
_DP = ctypes.POINTER(ctypes.c_double)

# name -> (restype, argtypes).  Must list EVERY symbol include/openmg_hip.h declares
# (tests/test_cabi_symbols.py checks that against the header).
SIGNATURES = {
    "omg_last_error": (ctypes.c_char_p, []),
    "omg_version": (ctypes.c_char_p, []),
    "omg_device_count": (_I, [_IP]),
    "omg_set_device": (_I, [_I]),
    "omg_hierarchy_create": (_I, [_I, _CSR, _CSR, _I, _D, _PP]),
    "omg_hierarchy_create_ex": (_I, [_I, _CSR, _CSR, _I, _D, _I, _PP]),
    "omg_hierarchy_create_from_fine": (_I, [_CSR, _I, _I64P, _I, _I, _D, _I, _PP]),
    "omg_hierarchy_dtype": (_I, [_P, _IP]),
    "omg_hierarchy_update_fine": (_I, [_P, _P, ctypes.c_int64, _I]),
    "omg_hierarchy_destroy": (_I, [_P]),
    "omg_hierarchy_set_stream": (_I, [_P, _P]),
    "omg_hierarchy_sync": (_I, [_P]),
    "omg_hierarchy_level_rows": (_I, [_P, _I, _I64P]),
    "omg_hierarchy_level_sets": (_I, [_P, _I, _I64P]),
    "omg_hierarchy_set_info": (_I, [_P, _I, _I, _I64P, _I64P]),
    "omg_hierarchy_level_fused": (_I, [_P, _I, _IP]),
    "omg_hierarchy_level_flags": (_I, [_P, _I, _IP]),
    "omg_hierarchy_use_plane": (_I, [_P, _I]),
    "omg_hierarchy_plane_info": (_I, [_P, _I, _I64P]),
    "omg_hierarchy_format_info": (_I, [_P, _I, _I, _I, _I64P]),
    "omg_format_selftest": (_I, [_CSR, _I, _I64P]),
    "omg_vcycle": (_I, [_P, _I, _P, _P, _I, _I, _DP]),
    "omg_vcycle_ex": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _DP]),
    "omg_vcycle_dev": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _DP]),
    "omg_resident_load_dev": (_I, [_P, _P, _P]),
    "omg_resident_fetch_dev": (_I, [_P, _P]),
    "omg_device_synchronize": (_I, []),
    "omg_device_mem_info": (_I, [_I64P, _I64P]),
    "omg_solve": (_I, [_P, _P, _P, _I, _I, _I, _D, _IP, _DP]),
    "omg_resident_load": (_I, [_P, _P, _P]),
    "omg_resident_cycle": (_I, [_P, _I, _I, _DP]),
    "omg_resident_cycles": (_I, [_P, _I, _I, _I, _DP]),
    "omg_resident_fetch": (_I, [_P, _P]),
    "omg_resident_spmv_time": (_I, [_P, _I, _DP]),
    "omg_resident_use_graph": (_I, [_P, _I]),
    "omg_profile_enable": (_I, [_P, _I]),
    "omg_profile_read": (_I, [_P, _I64P, _DP]),
    "omg_level_smooth": (_I, [_P, _I, _P, _P, _I]),
    "omg_level_residual": (_I, [_P, _I, _P, _P, _P, _DP]),
    "omg_level_spmv": (_I, [_P, _I, _P, _P]),
    "omg_level_restrict": (_I, [_P, _I, _P, _P]),
    "omg_level_prolong_add": (_I, [_P, _I, _P, _P]),
    "omg_coarse_solve": (_I, [_P, _P, _P]),
    "omg_hierarchy_coarse_info": (_I, [_P, _I64P]),
    "omg_spmv": (_I, [_CSR, _P, _P]),
    "omg_residual": (_I, [_CSR, _P, _P, _P, _DP]),
    "omg_gauss_seidel": (_I, [_CSR, _P, _P, _I, _D, _I, _D, _IP]),
    "omg_direct_solve": (_I, [_CSR, _P, _P]),
    "omg_rap": (_I, [_CSR, _CSR, _PP, _I64P, _I64P, _I64P]),
    "omg_spgemm": (_I, [_CSR, _CSR, _PP, _I64P, _I64P, _I64P]),
    "omg_csr_result_fetch": (_I, [_P, _P, _P, _P]),
    "omg_csr_result_free": (_I, [_P]),
    "omg_restriction": (_I, [_I, _I64P, _P, _P, _P, _I64P, _I64P]),
    "omg_host_checksum": (_I, [_P, ctypes.c_int64, ctypes.POINTER(ctypes.c_uint64)]),
    "omg_dist_create": (_I, [_I, _I, _I, _P, _CSR, _I64P, _I, _D, _PP]),
    "omg_dist_create_ex": (_I, [_I, _I, _I, _P, _CSR, _I64P, _I, _D, _I, _PP]),
    "omg_dist_destroy": (_I, [_P]),
    "omg_dist_set_tail": (_I, [_P, _P]),
    "omg_hierarchy_cycle_dev": (_I, [_P, _P, _P, _I, _I, _P]),
    "omg_dist_set_stream": (_I, [_P, _P]),
    "omg_dist_sync": (_I, [_P]),
    "omg_rccl_unique_id": (_I, [_P]),
    "omg_rccl_self_exchange_time": (_I, [ctypes.c_int64, _I, _DP]),
    "omg_pdist_create": (_I, [_I, _I, _I, _I, _I, _I, _P, _D, _PP]),
    "omg_pdist_destroy": (_I, [_P]),
    "omg_pdist_set_tail": (_I, [_P, _P]),
    "omg_pdist_connect": (_I, [_P, _P, _P]),
    "omg_peer_access": (_I, [_I, _I, _P]),
    "omg_pdist_p2p_handle_count": (_I, [_P, _P]),
    "omg_pdist_p2p_handles": (_I, [_P, _P, _I]),
    "omg_pdist_p2p_open": (_I, [_P, _I, _P, _I]),
    "omg_pdist_p2p_local": (_I, [_P, _P]),
    "omg_pdist_p2p_enable": (_I, [_P, _I]),
    "omg_pdist_p2p_status": (_I, [_P, _P]),
    "omg_pdist_cycles_squares": (_I, [_P, _I, _P]),
    "omg_pdist_rccl_ranks": (_I, [_P, _IP]),
    "omg_pdist_load": (_I, [_P, _P, _P]),
    "omg_pdist_fetch": (_I, [_P, _P]),
    "omg_pdist_sync": (_I, [_P]),
    "omg_pdist_info": (_I, [_P, _I64P]),
    "omg_pdist_set_gate": (_I, [_P, _I]),
    "omg_pdist_trace": (_I, [_P, _I]),
    "omg_pdist_progress": (_I, [_P, ctypes.POINTER(ctypes.c_uint)]),
    "omg_pdist_cycles": (_I, [_P, _I, _P]),
    "omg_pdist_cycles_ex": (_I, [_P, _I, _I, _I, _P]),
    "omg_pdist_group_cycles_ex": (_I, [_P, _I, _I, _I, _P]),
    "omg_sdist_p2p_handle_count": (_I, [_P, _P]),
    "omg_sdist_p2p_handles": (_I, [_P, _P, _I]),
    "omg_sdist_p2p_open": (_I, [_P, _I, _P, _I]),
    "omg_sdist_p2p_local": (_I, [_P, _P]),
    "omg_sdist_p2p_enable": (_I, [_P, _I]),
    "omg_sdist_p2p_status": (_I, [_P, _P]),
    "omg_pdist_group_create": (_I, [_I, _P, _PP]),
    "omg_pdist_group_destroy": (_I, [_P]),
    "omg_pdist_group_cycles": (_I, [_P, _I, _P]),
    "omg_sdist_create": (_I, [_I, _I, _I, _I, _I, _I, _CSR, _D, _I, _PP]),
    "omg_sdist_destroy": (_I, [_P]),
    "omg_sdist_coarse_size": (_I, [_P, _I64P, _I64P, _I64P]),
    "omg_sdist_coarse_fetch": (_I, [_P, _P, _P, _P]),
    "omg_sdist_set_tail": (_I, [_P, _P]),
    "omg_sdist_connect": (_I, [_P, _P]),
    "omg_sdist_rccl_ranks": (_I, [_P, _IP]),
    "omg_sdist_info": (_I, [_P, _I, _I64P]),
    "omg_sdist_load": (_I, [_P, _P, _P]),
    "omg_sdist_fetch": (_I, [_P, _P]),
    "omg_sdist_sync": (_I, [_P]),
    "omg_sdist_cycles": (_I, [_P, _I, _I, _I, _P]),
    "omg_sdist_group_create": (_I, [_I, _P, _PP]),
    "omg_sdist_group_destroy": (_I, [_P]),
    "omg_sdist_group_cycles": (_I, [_P, _I, _I, _I, _P]),
    "omg_dist_connect": (_I, [_P, _P]),
    "omg_dist_rccl_ranks": (_I, [_P, _IP]),
    "omg_dist_load": (_I, [_P, _P, _P]),
    "omg_dist_fetch": (_I, [_P, _P]),
    "omg_dist_cycle": (_I, [_P, _I, _I, _DP]),
    "omg_dist_cycles": (_I, [_P, _I, _I, _I, _DP]),
    "omg_dist_spmv_time": (_I, [_P, _I, _DP]),
    "omg_dist_format_info": (_I, [_P, _I, _I, _I, _I64P]),
    "omg_dist_level_flags": (_I, [_P, _I, _IP]),
    "omg_dist_group_create": (_I, [_I, _PP, _PP]),
    "omg_dist_group_destroy": (_I, [_P]),
    "omg_dist_group_cycle": (_I, [_P, _I, _I, _DP]),
    "omg_dist_group_cycles": (_I, [_P, _I, _I, _I, _DP]),
}


class DistLevelView(ctypes.Structure):
    """omg_dist_level in include/openmg_hip.h."""
    _fields_ = [("A", CsrView), ("R", CsrView), ("n_halo", ctypes.c_int64),
                ("keys", ctypes.c_void_p), ("n_sets", ctypes.c_int32), ("n_peers", ctypes.c_int32),
                ("peers", ctypes.c_void_p), ("send_off", ctypes.c_void_p),
                ("send_idx", ctypes.c_void_p), ("recv_off", ctypes.c_void_p),
                ("set_group", ctypes.c_int32), ("entry_group", ctypes.c_void_p)]

_lib = None


def lib():
    """Load the shared library once.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with `make -C openmg_amd/csrc` (or __graft_entry__.build()). "
                "openmg_amd has no CPU fallback." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(code):
    if code != OMG_OK:
        raise HipError(code, lib().omg_last_error().decode("utf-8", "replace"))


def device_mem_info():
    """(free, total) bytes of the current device."""
    f, t = ctypes.c_int64(0), ctypes.c_int64(0)
    check(lib().omg_device_mem_info(ctypes.byref(f), ctypes.byref(t)))
    return f.value, t.value


def device_count():
    n = ctypes.c_int(0)
    check(lib().omg_device_count(ctypes.byref(n)))
    return n.value


def require_gpu():
    if device_count() <= 0:
        raise HipError(ERR_NO_DEVICE, "no MI355X / HIP device visible; openmg_amd has no CPU fallback")


def smoother_code(kind):
    if isinstance(kind, int):
        return kind
    try:
        return SMOOTHERS[kind]
    except KeyError:
        raise ValueError("unknown smoother %r (choose from %s)" % (kind, sorted(set(SMOOTHERS))))


DTYPE_F64, DTYPE_F32 = 0, 1


def dtype_code(dtype):
    """OMG_DTYPE_* for a numpy dtype / name."""
    if isinstance(dtype, int) and not isinstance(dtype, bool) and dtype in (DTYPE_F64, DTYPE_F32):
        return dtype
    dt = np.dtype(dtype)
    if dt == np.float64:
        return DTYPE_F64
    if dt == np.float32:
        return DTYPE_F32
    raise ValueError("dtype must be float64 or float32, not %r" % (dtype,))


def host_checksum(a):
    """64-bit digest of every byte of a NumPy array (omg_host_checksum: all host threads, no device)."""
    a = np.ascontiguousarray(a)
    out = ctypes.c_uint64(0)
    check(lib().omg_host_checksum(a.ctypes.data, a.nbytes, ctypes.byref(out)))
    return out.value


def as_csr(A):
    """CSR with int32 index arrays / float64 data; stored column order is kept as is."""
    if not sp.isspmatrix_csr(A):
        A = sp.csr_matrix(A)
    if A.nnz >= 2 ** 31 - 1 or max(A.shape) >= 2 ** 31 - 1:
        raise ValueError("operator too large for int32 indices; shard it first")
    if A.indptr.dtype != np.int32 or A.indices.dtype != np.int32 or A.data.dtype != np.float64 \
            or not (A.indptr.flags.c_contiguous and A.indices.flags.c_contiguous and A.data.flags.c_contiguous):
        A = sp.csr_matrix((np.ascontiguousarray(A.data, dtype=np.float64),
                           np.ascontiguousarray(A.indices, dtype=np.int32),
                           np.ascontiguousarray(A.indptr, dtype=np.int32)), shape=A.shape)
    return A


def csr_view(A):
    """(CsrView, keepalive) for a CSR matrix prepared by as_csr()."""
    v = CsrView(A.shape[0], A.shape[1], A.nnz, A.indptr.ctypes.data, A.indices.ctypes.data, A.data.ctypes.data)
    return v


def vec(x, n=None, copy=False):
    """Contiguous float64 1-D array (flattening (n,1) columns like the reference accepts)."""
    a = np.asarray(x, dtype=np.float64)
    a = a.reshape(-1)
    if copy or not a.flags.c_contiguous:
        a = np.array(a, dtype=np.float64, order="C", copy=True)
    if n is not None and a.size != n:
        raise ValueError("vector has %d entries, expected %d" % (a.size, n))
    return a


class Hierarchy:
    """Device-resident A/R hierarchy (omg_hierarchy)."""

    def __init__(self, A_list, R_list, smoother="gs", omega=1.0, dtype="float64"):
        """dtype: precision the levels are stored and computed in on the device ("float64", the
        reference's, or "float32"); host vectors are float64 either way."""
        if len(R_list) != len(A_list) - 1:
            raise ValueError("need len(R) == len(A) - 1")
        self._A = [as_csr(M) for M in A_list]
        self._R = [as_csr(M) for M in R_list]
        arrA = (CsrView * len(self._A))(*[csr_view(M) for M in self._A])
        arrR = (CsrView * max(len(self._R), 1))(*[csr_view(M) for M in self._R])
        self.smoother = smoother_code(smoother)
        self.omega = float(omega)
        self.n_levels = len(self._A)
        self.sizes = [M.shape[0] for M in self._A]
        h = ctypes.c_void_p()
        self.dtype = dtype_code(dtype)
        check(lib().omg_hierarchy_create_ex(self.n_levels, arrA, arrR, self.smoother, self.omega, self.dtype,
                                            ctypes.byref(h)))
        self._h = h
        # the device copy is complete; the host copies are only kept for .sizes
        self._A = self._R = None

    @classmethod
    def from_fine(cls, A_in, shape, n_restrictions, smoother="gs", omega=1.0, dtype="float64"):
        """mgSolve's setup on the device (omg_hierarchy_create_from_fine): restrictions, Galerkin products and the
        qualification of the levels all in HBM; len(sizes) = n_restrictions + 1."""
        self = cls.__new__(cls)
        A0 = as_csr(A_in)
        shape = tuple(int(s) for s in shape)
        arr = (ctypes.c_int64 * len(shape))(*shape)
        self.smoother = smoother_code(smoother)
        self.omega = float(omega)
        self.dtype = dtype_code(dtype)
        self.n_levels = int(n_restrictions) + 1
        self.sizes = [A0.shape[0] // (2 ** len(shape)) ** l for l in range(self.n_levels)]
        h = ctypes.c_void_p()
        v = csr_view(A0)
        check(lib().omg_hierarchy_create_from_fine(ctypes.byref(v), len(shape), arr, int(n_restrictions), self.smoother, self.omega,
                                                   self.dtype, ctypes.byref(h)))
        self._h = h
        self._A = self._R = None
        return self

    def update_fine(self, data, on_device=False):
        """New values for the fine operator, same pattern (omg_hierarchy_update_fine).  data: the CSR's value array as a
        float64 NumPy array, or — on_device — (device pointer, number of entries)."""
        if on_device:
            ptr, nnz = data
            check(lib().omg_hierarchy_update_fine(self._h, ctypes.c_void_p(int(ptr)), int(nnz), 1))
        else:
            d = np.ascontiguousarray(data, dtype=np.float64)
            check(lib().omg_hierarchy_update_fine(self._h, d.ctypes.data, d.size, 0))

    def close(self):
        if getattr(self, "_h", None):
            lib().omg_hierarchy_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- whole cycles -------------------------------------------------------------------
    def vcycle(self, b, x, pre, post, level=0):
        """x is updated in place; returns the residual norm."""
        n = self.sizes[level]
        b = vec(b, n)
        assert x.dtype == np.float64 and x.flags.c_contiguous and x.size == n
        norm = ctypes.c_double(0.0)
        check(lib().omg_vcycle(self._h, level, b.ctypes.data, x.ctypes.data, int(pre), int(post), ctypes.byref(norm)))
        return norm.value

    def vcycle_ex(self, b, x_in, x_out, x_pre, pre, post, level=0):
        """omg_vcycle_ex: x_in (None: zeros) -> x_out; x_pre (None: not wanted) receives the iterate after the
        pre-smoothing sweeps (the reference's in-place smoother overwrites `initial` with it).  Returns the norm."""
        n = self.sizes[level]
        b = vec(b, n)
        for a in (x_in, x_out, x_pre):
            assert a is None or (a.dtype == np.float64 and a.flags.c_contiguous and a.size == n)
        norm = ctypes.c_double(0.0)
        check(lib().omg_vcycle_ex(self._h, level, b.ctypes.data, None if x_in is None else x_in.ctypes.data, x_out.ctypes.data,
                                  None if x_pre is None else x_pre.ctypes.data, int(pre), int(post), ctypes.byref(norm)))
        return norm.value

    def vcycle_dev(self, b_ptr, x_in_ptr, x_out_ptr, x_pre_ptr, pre, post, level=0):
        """omg_vcycle_dev: vcycle_ex on DEVICE arrays given by address (double, natural numbering; 0 / None: absent).
        Returns the norm; the outputs are complete on return."""
        norm = ctypes.c_double(0.0)
        check(lib().omg_vcycle_dev(self._h, level, ctypes.c_void_p(int(b_ptr)), ctypes.c_void_p(int(x_in_ptr or 0)),
                                   ctypes.c_void_p(int(x_out_ptr)), ctypes.c_void_p(int(x_pre_ptr or 0)), int(pre), int(post),
                                   ctypes.byref(norm)))
        return norm.value

    def resident_load_dev(self, b_ptr, x0_ptr=None):
        check(lib().omg_resident_load_dev(self._h, ctypes.c_void_p(int(b_ptr)), ctypes.c_void_p(int(x0_ptr or 0))))

    def resident_fetch_dev(self, x_ptr):
        check(lib().omg_resident_fetch_dev(self._h, ctypes.c_void_p(int(x_ptr))))

    def solve(self, b, x, pre, post, max_cycles, threshold):
        n = self.sizes[0]
        b = vec(b, n)
        assert x.dtype == np.float64 and x.flags.c_contiguous and x.size == n
        cycles = ctypes.c_int(0)
        norm = ctypes.c_double(0.0)
        check(lib().omg_solve(self._h, b.ctypes.data, x.ctypes.data, int(pre), int(post), int(max_cycles),
                              float(threshold), ctypes.byref(cycles), ctypes.byref(norm)))
        return cycles.value, norm.value

    def cycle_dev(self, b_dev, x_dev, pre, post, hip_stream=None):
        """One cycle from a zero iterate on DEVICE vectors (addresses of level-0 doubles in natural numbering):
        omg_hierarchy_cycle_dev, enqueued on `hip_stream` (None: the hierarchy's own), no host synchronisation."""
        check(lib().omg_hierarchy_cycle_dev(self._h, ctypes.c_void_p(int(b_dev)), ctypes.c_void_p(int(x_dev)), int(pre), int(post),
                                            ctypes.c_void_p(hip_stream or 0)))

    # -- resident ------------------------------------------------------------------------
    def resident_load(self, b, x0=None):
        b = vec(b, self.sizes[0])
        x0 = None if x0 is None else vec(x0, self.sizes[0])        # kept alive until the call returns
        check(lib().omg_resident_load(self._h, b.ctypes.data, None if x0 is None else x0.ctypes.data))

    def resident_cycle(self, pre, post, want_norm=True):
        if want_norm:
            norm = ctypes.c_double(0.0)
            check(lib().omg_resident_cycle(self._h, int(pre), int(post), ctypes.byref(norm)))
            return norm.value
        check(lib().omg_resident_cycle(self._h, int(pre), int(post), None))
        return None

    def resident_cycles(self, pre, post, n_cycles):
        """n_cycles V-cycles back to back; returns every cycle's residual norm (omg_resident_cycles)."""
        norms = (ctypes.c_double * max(int(n_cycles), 1))()
        check(lib().omg_resident_cycles(self._h, int(pre), int(post), int(n_cycles), norms))
        return [float(norms[k]) for k in range(int(n_cycles))]

    def resident_fetch(self):
        x = np.empty(self.sizes[0], dtype=np.float64)
        check(lib().omg_resident_fetch(self._h, x.ctypes.data))
        return x

    def spmv_time(self, reps=20):
        """Average milliseconds of one fine-grid y = A[0] x launch on the resident operator."""
        ms = ctypes.c_double(0.0)
        check(lib().omg_resident_spmv_time(self._h, int(reps), ctypes.byref(ms)))
        return ms.value

    def use_graph(self, enable=True):
        check(lib().omg_resident_use_graph(self._h, 1 if enable else 0))

    def set_stream(self, hip_stream):
        check(lib().omg_hierarchy_set_stream(self._h, ctypes.c_void_p(hip_stream or 0)))

    def sync(self):
        check(lib().omg_hierarchy_sync(self._h))

    def device_dtype(self):
        """np.dtype the levels are held in on the device (omg_hierarchy_dtype)."""
        code = ctypes.c_int(-1)
        check(lib().omg_hierarchy_dtype(self._h, ctypes.byref(code)))
        return np.dtype(np.float32 if code.value == DTYPE_F32 else np.float64)

    def level_sets(self, level):
        v = ctypes.c_int64(0)
        check(lib().omg_hierarchy_level_sets(self._h, level, ctypes.byref(v)))
        return v.value

    def level_fused(self, level):
        v = ctypes.c_int(0)
        check(lib().omg_hierarchy_level_fused(self._h, level, ctypes.byref(v)))
        return bool(v.value)

    def level_flags(self, level):
        """dict(fused_last_set=bool, scatter_prolong=bool, union_walk=bool, march=bool, plane=bool, stencil27=bool, var7=bool,
        march_scan=bool) of a smoothed level."""
        f = ctypes.c_int(0)
        check(lib().omg_hierarchy_level_flags(self._h, int(level), ctypes.byref(f)))
        return {"fused_last_set": bool(f.value & 1), "scatter_prolong": bool(f.value & 2), "union_walk": bool(f.value & 16),
                "march": bool(f.value & 32), "plane": bool(f.value & 64), "stencil27": bool(f.value & 128), "var7": bool(f.value & 256),
                "march_scan": bool(f.value & 512)}

    def use_plane(self, enable=True):
        """Plane-pipelined passes on / off (omg_hierarchy_use_plane; same iterate either way)."""
        check(lib().omg_hierarchy_use_plane(self._h, 1 if enable else 0))

    PLANE_FIELDS = ("nx", "ny", "nz", "tile_x", "tile_y", "tile_z", "workgroups", "threads")

    def plane_info(self, level):
        out = (ctypes.c_int64 * len(self.PLANE_FIELDS))()
        check(lib().omg_hierarchy_plane_info(self._h, int(level), out))
        return dict(zip(self.PLANE_FIELDS, [int(v) for v in out]))

    def set_info(self, level, s):
        """(rows, stored entries) of smoother set `s` of a level."""
        rows, nnz = ctypes.c_int64(0), ctypes.c_int64(0)
        check(lib().omg_hierarchy_set_info(self._h, level, s, ctypes.byref(rows), ctypes.byref(nnz)))
        return rows.value, nnz.value

    FORMAT_FIELDS = ("rows", "nnz", "blocks", "pattern_blocks", "pattern_rows", "pattern_nnz",
                     "coldict_nnz", "valdict_nnz", "format_bytes", "csr_bytes", "ell_blocks", "ell_nnz")

    def format_info(self, level, op="A", set=-1):
        """How A / R / P (= R^T) of `level` is held in HBM (omg_hierarchy_format_info)."""
        out = (ctypes.c_int64 * len(self.FORMAT_FIELDS))()
        check(lib().omg_hierarchy_format_info(self._h, int(level), {"A": 0, "R": 1, "P": 2}[op], int(set), out))
        return dict(zip(self.FORMAT_FIELDS, [int(v) for v in out]))

    def profile_enable(self, classes=True):
        """True = every class, False = off, or an iterable of class names (PROFILE_NAMES)."""
        if classes is True:
            mask = -1
        elif not classes:
            mask = 0
        else:
            mask = 0
            for name in classes:
                mask |= 1 << PROFILE_NAMES.index(name)
        check(lib().omg_profile_enable(self._h, mask))

    def profile_read(self):
        n = (ctypes.c_int64 * PROFILE_CLASSES)()
        ms = (ctypes.c_double * PROFILE_CLASSES)()
        check(lib().omg_profile_read(self._h, n, ms))
        return {PROFILE_NAMES[i]: (int(n[i]), float(ms[i])) for i in range(PROFILE_CLASSES)}

    # -- single operations -----------------------------------------------------------------
    def smooth(self, level, b, x, iterations):
        n = self.sizes[level]
        b = vec(b, n)
        assert x.dtype == np.float64 and x.flags.c_contiguous and x.size == n
        check(lib().omg_level_smooth(self._h, level, b.ctypes.data, x.ctypes.data, int(iterations)))
        return x

    def residual(self, level, b, x, want_norm=False):
        n = self.sizes[level]
        b, x = vec(b, n), vec(x, n)
        r = np.empty(n)
        if want_norm:
            norm = ctypes.c_double(0.0)
            check(lib().omg_level_residual(self._h, level, b.ctypes.data, x.ctypes.data, r.ctypes.data, ctypes.byref(norm)))
            return r, norm.value
        check(lib().omg_level_residual(self._h, level, b.ctypes.data, x.ctypes.data, r.ctypes.data, None))
        return r

    def spmv(self, level, x):
        """A[level] x on the operator as the hierarchy holds it (omg_level_spmv)."""
        n = self.sizes[level]
        x = vec(x, n)
        y = np.empty(n)
        check(lib().omg_level_spmv(self._h, level, x.ctypes.data, y.ctypes.data))
        return y

    def restrict(self, level, fine):
        fine = vec(fine, self.sizes[level])
        out = np.empty(self.sizes[level + 1])
        check(lib().omg_level_restrict(self._h, level, fine.ctypes.data, out.ctypes.data))
        return out

    def prolong_add(self, level, coarse, fine):
        coarse = vec(coarse, self.sizes[level + 1])
        out = vec(fine, self.sizes[level], copy=True)
        check(lib().omg_level_prolong_add(self._h, level, coarse.ctypes.data, out.ctypes.data))
        return out

    def coarse_info(self):
        """dict(blocks, n, half_bandwidth, bytes_per_solve) of the coarsest level's direct solver:
        blocks == 1 is the explicit inverse, > 1 substructuring along the band."""
        out = (ctypes.c_int64 * 4)()
        check(lib().omg_hierarchy_coarse_info(self._h, out))
        return {"blocks": int(out[0]), "n": int(out[1]), "half_bandwidth": int(out[2]), "bytes_per_solve": int(out[3])}

    def coarse_solve(self, b):
        b = vec(b, self.sizes[-1])
        x = np.empty(self.sizes[-1])
        check(lib().omg_coarse_solve(self._h, b.ctypes.data, x.ctypes.data))
        return x


# ---- standalone ------------------------------------------------------------------------
def format_selftest(A, dtype="float64"):
    """Code A into the device format on the host, decode, compare bit for bit (no GPU needed);
    returns the format statistics (Hierarchy.FORMAT_FIELDS).  Raises HipError on a mismatch."""
    A = as_csr(A)
    out = (ctypes.c_int64 * len(Hierarchy.FORMAT_FIELDS))()
    v = csr_view(A)
    check(lib().omg_format_selftest(ctypes.byref(v), dtype_code(dtype), out))
    return dict(zip(Hierarchy.FORMAT_FIELDS, [int(x) for x in out]))


def spmv(A, x):
    A = as_csr(A)
    x = vec(x, A.shape[1])
    y = np.empty(A.shape[0])
    v = csr_view(A)
    check(lib().omg_spmv(ctypes.byref(v), x.ctypes.data, y.ctypes.data))
    return y


def residual(A, b, x, want_norm=False):
    A = as_csr(A)
    b, x = vec(b, A.shape[0]), vec(x, A.shape[1])
    r = np.empty(A.shape[0])
    v = csr_view(A)
    if want_norm:
        norm = ctypes.c_double(0.0)
        check(lib().omg_residual(ctypes.byref(v), b.ctypes.data, x.ctypes.data, r.ctypes.data, ctypes.byref(norm)))
        return r, norm.value
    check(lib().omg_residual(ctypes.byref(v), b.ctypes.data, x.ctypes.data, r.ctypes.data, None))
    return r


def gauss_seidel(A, b, x, smoother="gs", omega=1.0, iterations=None, threshold=None):
    """x (contiguous float64) is updated in place; returns the number of sweeps done."""
    A = as_csr(A)
    b = vec(b, A.shape[0])
    assert x.dtype == np.float64 and x.flags.c_contiguous and x.size == A.shape[0]
    v = csr_view(A)
    done = ctypes.c_int(0)
    check(lib().omg_gauss_seidel(ctypes.byref(v), b.ctypes.data, x.ctypes.data, smoother_code(smoother),
                                 float(omega), -1 if iterations is None else int(iterations),
                                 -1.0 if threshold is None else float(threshold), ctypes.byref(done)))
    return done.value


def direct_solve(A, b):
    A = as_csr(A)
    b = vec(b, A.shape[0])
    x = np.empty(A.shape[0])
    v = csr_view(A)
    check(lib().omg_direct_solve(ctypes.byref(v), b.ctypes.data, x.ctypes.data))
    return x


def _product(fn, X, Y):
    X, Y = as_csr(X), as_csr(Y)
    vx, vy = csr_view(X), csr_view(Y)
    res = ctypes.c_void_p()
    nr, nc, nnz = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
    check(fn(ctypes.byref(vx), ctypes.byref(vy), ctypes.byref(res), ctypes.byref(nr),
             ctypes.byref(nc), ctypes.byref(nnz)))
    indptr = np.empty(nr.value + 1, dtype=np.int32)
    indices = np.empty(max(nnz.value, 0), dtype=np.int32)
    data = np.empty(max(nnz.value, 0), dtype=np.float64)
    check(lib().omg_csr_result_fetch(res, indptr.ctypes.data, indices.ctypes.data, data.ctypes.data))
    return sp.csr_matrix((data, indices, indptr), shape=(nr.value, nc.value))


def rap(R, A):
    """Galerkin product (R A) R^T on the device -> scipy CSR (sorted columns)."""
    return _product(lib().omg_rap, R, A)


def spgemm(X, Y):
    """Sparse product X Y on the device -> scipy CSR (sorted columns)."""
    return _product(lib().omg_spgemm, X, Y)


def restriction(shape):
    """operators.restriction(shape) built on the device -> scipy CSR."""
    shape = tuple(int(s) for s in shape)
    dim = len(shape)
    N = int(np.prod(shape))
    n = N // (2 ** dim)
    arr = (ctypes.c_int64 * dim)(*shape)
    indptr = np.empty(n + 1, dtype=np.int32)
    indices = np.empty(n * (2 ** dim), dtype=np.int32)
    data = np.empty(n * (2 ** dim), dtype=np.float64)
    nr, nnz = ctypes.c_int64(0), ctypes.c_int64(0)
    check(lib().omg_restriction(dim, arr, indptr.ctypes.data, indices.ctypes.data, data.ctypes.data,
                                ctypes.byref(nr), ctypes.byref(nnz)))
    k = nnz.value
    if k and int(indices[:k].max()) >= N:
        # the reference's LIL assignment raises the same for shapes its offsets overrun
        raise IndexError("column index (%d) out of range" % int(indices[:k].max()))
    return sp.csr_matrix((data[:k], indices[:k], indptr[:nr.value + 1]), shape=(nr.value, N))
