"""zkp_hip -- ctypes binding of libzkp_hip.so (include/zkp_hip.h), the MI355X backend for the MSM / NTT hot path.

This module is plumbing: it loads the in-tree HIP library and exposes its C ABI over numpy arrays (host entry
points) and torch CUDA tensors (``*_dev`` entry points).  There is NO Python or CPU implementation of any
operation here: if the library is missing or no gfx950 device is usable, calls raise ``ZkpError``.

Data conventions (arkworks in-memory forms, see include/zkp_hip.h): uint64 little-endian Montgomery limbs.
    Fr (n,4)   Goldilocks (n,)   G1 affine (n,12) + uint8 infinity flags
"""
import ctypes as C
import os
import sys

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
LIB_PATH = os.path.join(_ROOT, "libzkp_hip.so")
if os.environ.get("ZKP_HIP_LIB"):  # development: try another build of the same library
    LIB_PATH = os.environ["ZKP_HIP_LIB"]

ZKP_OK, ZKP_E_ARG, ZKP_E_NOMEM, ZKP_E_DEVICE, ZKP_E_SIZE = 0, -1, -2, -3, -4


class ZkpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"zkp_hip error {code}: {msg}")
        self.code = code


_lib = None

_VP, _SZ, _U8P = C.c_void_p, C.c_size_t, C.c_void_p
_SIGS = {
    "zkp_init": ([C.c_int], C.c_int),
    "zkp_init_devices": ([_VP, C.c_int], C.c_int),
    "zkp_device_count": ([], C.c_int),
    "zkp_set_device": ([C.c_int], C.c_int),
    "zkp_shutdown": ([], None),
    "zkp_last_error": ([], C.c_char_p),
    "zkp_abi_version": ([], C.c_int),
    "zkp_profile_enable": ([C.c_int], None),
    "zkp_profile_reset": ([], None),
    "zkp_profile_read": ([C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)], C.c_int),
    "zkp_profile_clock_read": ([C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)], C.c_int),
    "zkp_probe_mad_rate": ([C.c_uint, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)], C.c_int),
    "zkp_g1_bases_create": ([_VP, _U8P, _SZ, C.POINTER(_VP)], C.c_int),
    "zkp_g1_bases_create_dev": ([_VP, _U8P, _SZ, _VP, C.POINTER(_VP)], C.c_int),
    "zkp_g1_bases_precompute": ([_VP, C.c_uint], C.c_int),
    "zkp_g1_bases_len": ([_VP], _SZ),
    "zkp_g1_bases_info": ([_VP, C.POINTER(C.c_uint), C.POINTER(C.c_uint)], C.c_int),
    "zkp_g1_bases_destroy": ([_VP], None),
    "zkp_msm_g1": ([_VP, _VP, _SZ, _VP, _VP], C.c_int),
    "zkp_msm_g1_dev": ([_VP, _VP, _SZ, _VP, _VP, _VP], C.c_int),
    "zkp_msm_g1_batch_dev": ([_VP, _VP, _SZ, _SZ, _VP, _VP, _VP], C.c_int),
    "zkp_msm_g1_partial_dev": ([_VP, _VP, _SZ, _VP, _VP], C.c_int),
    "zkp_msm_g1_partial": ([_VP, _VP, _SZ, _VP], C.c_int),
    "zkp_g1_bases_shard_count": ([_VP], C.c_int),
    "zkp_g1_bases_shard": ([_VP, _SZ, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(_SZ), C.POINTER(_SZ)], C.c_int),
    "zkp_msm_g1_sharded_dev": ([_VP, _VP, _SZ, _VP, _VP], C.c_int),
    "zkp_msm_g1_sharded_dev_after": ([_VP, _VP, _VP, _SZ, _VP, _VP], C.c_int),
    "zkp_g1_xyzz_sum": ([_VP, _SZ, _VP, _VP], C.c_int),
    "zkp_g1_mul": ([_VP, C.c_uint8, _VP, _VP, _VP], C.c_int),
    "zkp_g1_fixed_base_mul_dev": ([_VP, _SZ, _VP, _U8P, _VP], C.c_int),
    "zkp_selftest_fq_inverse_dev": ([_VP, _SZ, C.c_int, _VP, _VP], C.c_int),
    "zkp_srs_g1": ([_VP, _SZ, _VP], C.c_int),
    "zkp_ntt_fr": ([_VP, C.c_uint, C.c_int, _VP], C.c_int),
    "zkp_ntt_fr_dev": ([_VP, C.c_uint, _SZ, C.c_int, _VP, _VP], C.c_int),
    "zkp_ntt_fr_twiddle_dev": ([_VP, _SZ, _SZ, _SZ, C.c_uint, C.c_int, _VP], C.c_int),
    "zkp_ntt_fr_axis0_dev": ([_VP, _VP, C.c_uint, _SZ, C.c_int, C.c_uint, _SZ, _VP], C.c_int),
    "zkp_ntt_fr_layout_dev": ([_VP, _VP, C.c_uint, _SZ, C.c_int, _VP, _VP, C.c_uint, _SZ, _VP], C.c_int),
    "zkp_ntt_fr_sharded_geometry": ([C.c_uint, C.c_uint, C.c_uint, _VP], C.c_int),
    "zkp_ntt_fr_sharded_dev": ([_VP, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_uint, _VP], C.c_int),
    "zkp_ntt_fr_sharded": ([_VP, C.c_uint, C.c_int, _VP], C.c_int),
    "zkp_ntt_goldilocks": ([_VP, C.c_uint, C.c_int, _VP], C.c_int),
    "zkp_ntt_goldilocks_dev": ([_VP, C.c_uint, _SZ, C.c_int, _VP, _VP], C.c_int),
    "zkp_fri_layer_eval": ([_VP, _SZ, C.c_uint64, C.c_uint, _VP], C.c_int),
    "zkp_fri_fold": ([_VP, _SZ, C.c_uint64, _VP], C.c_int),
    "zkp_fri_merkle_node_count": ([_SZ], _SZ),
    "zkp_fri_merkle_tree": ([_VP, _SZ, _VP], C.c_int),
    "zkp_fri_merkle_tree_dev": ([_VP, _SZ, _VP, _VP], C.c_int),
    "zkp_fri_challenges": ([_VP, _SZ, C.c_uint64, _SZ, _VP, _VP], C.c_int),
    "zkp_fri_prove": ([_VP, _SZ, _SZ, _SZ, C.POINTER(_VP), C.POINTER(_SZ)], C.c_int),
    "zkp_fri_verify": ([_VP, _SZ], C.c_int),
    "zkp_free": ([_VP], None),
    "zkp_plonk_prove": ([_VP, _VP, _VP], C.c_int),
    "zkp_g2_generator": ([_VP], C.c_int),
    "zkp_g2_mul": ([_VP, C.c_uint8, _VP, _VP, _U8P], C.c_int),
    "zkp_pairing": ([_VP, C.c_uint8, _VP, C.c_uint8, _VP], C.c_int),
    "zkp_kzg_verify": ([_VP, _VP, C.c_uint8, _VP, C.c_uint8, _VP, _VP, C.POINTER(C.c_int)], C.c_int),
    "zkp_kzg_batch_verify": ([_VP, _SZ, _VP, _U8P, _VP, _VP, _U8P, _VP, _VP, C.POINTER(C.c_int)], C.c_int),
    "zkp_kzg_aggregate_commitments": ([_VP, _U8P, _SZ, _VP, _VP, _VP], C.c_int),
    "zkp_plonk_verify": ([_VP, _VP, _VP, C.POINTER(C.c_int)], C.c_int),
    "zkp_plonk_transcript_create": ([C.POINTER(_VP)], C.c_int),
    "zkp_plonk_transcript_destroy": ([_VP], None),
    "zkp_plonk_transcript_feed": ([_VP, _VP, C.c_uint8], C.c_int),
    "zkp_plonk_transcript_challenges": ([_VP, _SZ, _VP], C.c_int),
    "zkp_poly_mul_fr": ([_VP, _SZ, _VP, _SZ, _VP], C.c_int),
    "zkp_plonk_prover_create": ([_VP, C.c_uint, _VP, _VP, _VP, _VP, C.POINTER(_VP)], C.c_int),
    "zkp_plonk_prover_destroy": ([_VP], None),
    "zkp_plonk_round1": ([_VP, _VP, _VP, _VP], C.c_int),
    "zkp_plonk_round2": ([_VP, _VP, _VP, _VP, _VP, _VP], C.c_int),
    "zkp_plonk_round3": ([_VP, _VP, _VP, _VP, C.POINTER(C.c_size_t)], C.c_int),
    "zkp_plonk_round4": ([_VP, _VP, _VP], C.c_int),
    "zkp_plonk_round5": ([_VP, _VP, _VP, _VP], C.c_int),
    "zkp_plonk_get_poly": ([_VP, C.c_int, _VP, _SZ, C.POINTER(C.c_size_t)], C.c_int),
    "zkp_kzg_commit": ([_VP, _VP, _SZ, _VP, _VP], C.c_int),
    "zkp_kzg_open": ([_VP, _VP, _SZ, _VP, _VP, _VP, _VP], C.c_int),
}


def exported_symbols():
    """Names every include/zkp_hip.h declaration must be exported under (checked by the CPU test-suite)."""
    return sorted(_SIGS)


def lib():
    """Load libzkp_hip.so (no GPU needed to load it).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ZkpError(ZKP_E_DEVICE, f"{LIB_PATH} not found: run `python zkp-implementation_amd/build.py` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        # One HIP runtime per process: PyTorch's wheel bundles its own libamdhip64, and libzkp_hip.so resolves the same soname.  Whichever
        # is loaded first serves both -- unless this library came first and pulled in /opt/rocm's copy: the runtime torch loads afterwards
        # owns the devices and this one sees none ("no HIP device visible" from zkp_init in a process that had loaded the library
        # before importing torch, e.g. build() followed by smoke()).  So torch, where it exists, goes first; plain C callers never load it.
        if "torch" not in sys.modules:
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        l = C.CDLL(LIB_PATH)
        for name, (args, res) in _SIGS.items():
            fn = getattr(l, name)
            fn.argtypes = args
            fn.restype = res
        _lib = l
    return _lib


def _chk(code):
    if code != ZKP_OK:
        raise ZkpError(code, lib().zkp_last_error().decode())


def init(device=-1):
    _chk(lib().zkp_init(device))


def init_devices(devices=None, n=0):
    """One slot per listed HIP device (a device may repeat); devices=None: devices 0..n-1, n == 0: all visible."""
    if devices is None:
        _chk(lib().zkp_init_devices(None, int(n)))
    else:
        arr = (C.c_int * len(devices))(*[int(d) for d in devices])
        _chk(lib().zkp_init_devices(C.cast(arr, C.c_void_p), len(devices)))


def device_count():
    return int(lib().zkp_device_count())


def set_device(slot):
    """Slot of this thread's handle-less entries; -1 = default (slot 0; zkp_g1_bases_create shards over all slots)."""
    _chk(lib().zkp_set_device(int(slot)))


def shutdown():
    lib().zkp_shutdown()


def profile_enable(on=True):
    """True / 1: every phase; 2: the dominant kernel only (msm_accumulate: events and clock stamps -- each recorded phase boundary is a
    ~5 us bubble on the stream); False / 0: off."""
    lib().zkp_profile_enable(2 if on == 2 and on is not True else int(bool(on)))


def profile_reset():
    lib().zkp_profile_reset()


def profile_read(name):
    """-> (total milliseconds, number of records) of one phase since the last reset (see include/zkp_hip.h)."""
    ms, cnt = C.c_double(0), C.c_uint64(0)
    _chk(lib().zkp_profile_read(name.encode(), C.byref(ms), C.byref(cnt)))
    return ms.value, int(cnt.value)


def profile_clock_read(name):
    """-> (shader cycles, 100 MHz reference ticks, stamped workgroups) of an instrumented kernel family since the last reset;
    cycles / ticks * 100 = the shader clock in MHz it held under its own load (include/zkp_hip.h)."""
    cyc, ref, waves = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    _chk(lib().zkp_profile_clock_read(name.encode(), C.byref(cyc), C.byref(ref), C.byref(waves)))
    return int(cyc.value), int(ref.value), int(waves.value)


def probe_mad_rate(launches=20):
    """-> (v_mad_u64_u32 lane-ops per second, shader clock in MHz during the probe, ms per launch) measured now on this device."""
    rate, mhz, ms = C.c_double(0), C.c_double(0), C.c_double(0)
    _chk(lib().zkp_probe_mad_rate(launches, C.byref(rate), C.byref(mhz), C.byref(ms)))
    return rate.value, mhz.value, ms.value


def _np(a, dtype, shape=None):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a.reshape(shape) if shape is not None else a


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _stream_ptr(stream):
    if stream is None:
        try:
            import torch
            if torch.cuda.is_available():
                return C.c_void_p(torch.cuda.current_stream().cuda_stream)
        except ImportError:
            pass
        return None
    return C.c_void_p(int(stream))


def _dev_ptr(t, min_bytes):
    """torch CUDA tensor -> device pointer (contiguous, big enough)."""
    if not t.is_cuda or not t.is_contiguous():
        raise ZkpError(ZKP_E_ARG, "expected a contiguous CUDA tensor")
    if t.numel() * t.element_size() < min_bytes:
        raise ZkpError(ZKP_E_ARG, "tensor smaller than the operation needs")
    return C.c_void_p(t.data_ptr())


# ----------------------------------------------------------------------------- bases / MSM
class G1Bases:
    """Base points resident in HBM (the SRS of kzg/src/srs.rs:14-21, uploaded once)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def from_host(cls, xy, is_inf=None):
        xy = _np(xy, np.uint64, (-1, 12))
        inf = _np(is_inf, np.uint8) if is_inf is not None else None
        h = C.c_void_p()
        _chk(lib().zkp_g1_bases_create(_ptr(xy), _ptr(inf), xy.shape[0], C.byref(h)))
        return cls(h)

    @classmethod
    def from_device(cls, xy_tensor, n, is_inf_tensor=None, stream=None):
        h = C.c_void_p()
        inf = C.c_void_p(is_inf_tensor.data_ptr()) if is_inf_tensor is not None else None
        _chk(lib().zkp_g1_bases_create_dev(_dev_ptr(xy_tensor, 96 * n), inf, n, _stream_ptr(stream), C.byref(h)))
        return cls(h)

    def precompute(self, window_bits=20):
        """Expand to the multiples 2^(window_bits s) P (shared-bucket MSM, see include/zkp_hip.h)."""
        _chk(lib().zkp_g1_bases_precompute(self._h, window_bits))
        return self

    def shards(self):
        """[(slot, hip_device, offset, length)] of the chunks of this handle (one entry for a single-slot handle)."""
        out = []
        for i in range(lib().zkp_g1_bases_shard_count(self._h)):
            slot, dev, off, ln = C.c_int(0), C.c_int(0), C.c_size_t(0), C.c_size_t(0)
            _chk(lib().zkp_g1_bases_shard(self._h, i, C.byref(slot), C.byref(dev), C.byref(off), C.byref(ln)))
            out.append((int(slot.value), int(dev.value), int(off.value), int(ln.value)))
        return out

    def info(self):
        """-> (window_bits asked for, slices = insertions per scalar); (0, 0) when not expanded."""
        w, sl = C.c_uint(0), C.c_uint(0)
        _chk(lib().zkp_g1_bases_info(self._h, C.byref(w), C.byref(sl)))
        return int(w.value), int(sl.value)

    def __len__(self):
        return int(lib().zkp_g1_bases_len(self._h))

    def close(self):
        if self._h:
            lib().zkp_g1_bases_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def msm_g1(bases, scalars):
    """sum_i scalars[i] * bases[i]  ->  ((12,) uint64 affine Montgomery coords, is_infinity).  Host scalars."""
    scalars = _np(scalars, np.uint64, (-1, 4))
    out = np.zeros(12, dtype=np.uint64)
    inf = C.c_uint8(0)
    _chk(lib().zkp_msm_g1(bases._h, _ptr(scalars), scalars.shape[0], _ptr(out), C.byref(inf)))
    return out, int(inf.value)


def msm_g1_partial(bases, scalars):
    """Host scalars -> (24,) uint64 extended-Jacobian partial (summed over this process's devices for sharded bases)."""
    scalars = _np(scalars, np.uint64, (-1, 4))
    out = np.zeros(24, dtype=np.uint64)
    _chk(lib().zkp_msm_g1_partial(bases._h, _ptr(scalars), scalars.shape[0], _ptr(out)))
    return out


def msm_g1_sharded_dev(bases, scalar_tensors, n, events=None):
    """Sharded bases, one resident scalar tensor per chunk (on that chunk's device; None for a chunk beyond n) -> (affine, is_inf).

    Without `events` the library launches on each slot's own stream after a hipDeviceSynchronize() of that chunk's device, so tensors
    whose producing copy / kernel is still in flight on any torch stream are safe to pass.  With `events` (one recorded
    torch.cuda.Event, or None, per chunk) the chunk's launch waits for its event on the device instead and the host does not stall
    (include/zkp_hip.h, zkp_msm_g1_sharded_dev / zkp_msm_g1_sharded_dev_after)."""
    k = len(scalar_tensors)
    ptrs = (C.c_void_p * k)(*[(t.data_ptr() if t is not None else None) for t in scalar_tensors])
    out = np.zeros(12, dtype=np.uint64)
    inf = C.c_uint8(0)
    if events is None:
        _chk(lib().zkp_msm_g1_sharded_dev(bases._h, ptrs, n, _ptr(out), C.byref(inf)))
    else:
        if len(events) != k:
            raise ValueError("one event (or None) per chunk")
        evs = (C.c_void_p * k)(*[(e.cuda_event if e is not None else None) for e in events])
        _chk(lib().zkp_msm_g1_sharded_dev_after(bases._h, ptrs, evs, n, _ptr(out), C.byref(inf)))
    return out, int(inf.value)


def selftest_fq_inverse_dev(in_tensor, n, form, out_tensor, stream=None):
    """zkp_selftest_fq_inverse_dev: n raw base-field elements (form 0: 12 x u32, form 1: 16 x u32 per element) inverted on the device."""
    words = 12 if form == 0 else 16
    _chk(lib().zkp_selftest_fq_inverse_dev(_dev_ptr(in_tensor, 4 * words * n), n, form, _dev_ptr(out_tensor, 4 * words * n),
                                           _stream_ptr(stream)))


def msm_g1_dev(bases, scalars_tensor, n, stream=None):
    out = np.zeros(12, dtype=np.uint64)
    inf = C.c_uint8(0)
    _chk(lib().zkp_msm_g1_dev(bases._h, _dev_ptr(scalars_tensor, 32 * n), n, _stream_ptr(stream), _ptr(out),
                              C.byref(inf)))
    return out, int(inf.value)


def msm_g1_batch_dev(bases, scalar_tensors, n, stream=None):
    """Several MSMs over the same bases in one pass (equal length n): list of (affine (12,), is_inf)."""
    k = len(scalar_tensors)
    ptrs = (C.c_void_p * k)(*[_dev_ptr(t, 32 * n).value for t in scalar_tensors])
    xy, inf = np.zeros((k, 12), dtype=np.uint64), np.zeros(k, dtype=np.uint8)
    _chk(lib().zkp_msm_g1_batch_dev(bases._h, ptrs, k, n, _stream_ptr(stream), _ptr(xy), _ptr(inf)))
    return [(xy[i].copy(), int(inf[i])) for i in range(k)]


def msm_g1_partial_dev(bases, scalars_tensor, n, stream=None):
    """Unnormalised partial sum (X, Y, ZZ, ZZZ) as (24,) uint64 -- the multi-GPU exchange unit."""
    out = np.zeros(24, dtype=np.uint64)
    _chk(lib().zkp_msm_g1_partial_dev(bases._h, _dev_ptr(scalars_tensor, 32 * n), n, _stream_ptr(stream), _ptr(out)))
    return out


def g1_xyzz_sum(partials):
    partials = _np(partials, np.uint64, (-1, 24))
    out = np.zeros(12, dtype=np.uint64)
    inf = C.c_uint8(0)
    _chk(lib().zkp_g1_xyzz_sum(_ptr(partials), partials.shape[0], _ptr(out), C.byref(inf)))
    return out, int(inf.value)


def g1_mul(base_xy, base_inf, scalar):
    base_xy, scalar = _np(base_xy, np.uint64, (12,)), _np(scalar, np.uint64, (4,))
    out = np.zeros(12, dtype=np.uint64)
    inf = C.c_uint8(0)
    _chk(lib().zkp_g1_mul(_ptr(base_xy), int(base_inf), _ptr(scalar), _ptr(out), C.byref(inf)))
    return out, int(inf.value)


def g1_fixed_base_mul_dev(scalars_tensor, n, out_xy_tensor, out_inf_tensor=None, stream=None):
    inf = C.c_void_p(out_inf_tensor.data_ptr()) if out_inf_tensor is not None else None
    _chk(lib().zkp_g1_fixed_base_mul_dev(_dev_ptr(scalars_tensor, 32 * n), n, _dev_ptr(out_xy_tensor, 96 * n), inf,
                                         _stream_ptr(stream)))


def srs_g1(secret, n):
    """[s^i]G, i < n (kzg/src/srs.rs:48-63) -> (n,12) uint64."""
    secret = _np(secret, np.uint64, (4,))
    out = np.zeros((n, 12), dtype=np.uint64)
    _chk(lib().zkp_srs_g1(_ptr(secret), n, _ptr(out)))
    return out


# ----------------------------------------------------------------------------- NTT
def _log2(n):
    lg = int(n).bit_length() - 1
    if n <= 0 or (1 << lg) != n:
        raise ZkpError(ZKP_E_ARG, "size must be a power of two")
    return lg


def ntt_fr(data, inverse=False, coset=None):
    a = _np(data, np.uint64, (-1, 4)).copy()
    cs = _np(coset, np.uint64, (4,)) if coset is not None else None
    _chk(lib().zkp_ntt_fr(_ptr(a), _log2(a.shape[0]), int(bool(inverse)), _ptr(cs)))
    return a


def ntt_goldilocks(data, inverse=False, coset=None):
    a = _np(data, np.uint64).reshape(-1).copy()
    cs = _np(coset, np.uint64, (1,)) if coset is not None else None
    _chk(lib().zkp_ntt_goldilocks(_ptr(a), _log2(a.shape[0]), int(bool(inverse)), _ptr(cs)))
    return a


def ntt_fr_dev(tensor, log_n, batch=1, inverse=False, coset=None, stream=None):
    cs = _np(coset, np.uint64, (4,)) if coset is not None else None
    _chk(lib().zkp_ntt_fr_dev(_dev_ptr(tensor, (32 << log_n) * batch), log_n, batch, int(bool(inverse)), _ptr(cs),
                              _stream_ptr(stream)))


def ntt_fr_twiddle_dev(tensor, rows, cols, row0, log_n, inverse=False, stream=None):
    """tensor[r][c] *= omega_n^((row0 + r) * c) for a rows x cols row-major block (four-step transform, dist.py)."""
    _chk(lib().zkp_ntt_fr_twiddle_dev(_dev_ptr(tensor, 32 * rows * cols), rows, cols, row0, log_n, int(bool(inverse)),
                                      _stream_ptr(stream)))


class NttLayout(C.Structure):  # zkp_ntt_layout in include/zkp_hip.h (strides in elements)
    _fields_ = [("lo_bits", C.c_uint), ("mid_bits", C.c_uint), ("mid_stride", C.c_size_t), ("hi_stride", C.c_size_t),
                ("batch_stride", C.c_size_t)]


def ntt_fr_axis0_dev(t_in, t_out, log_len, cols, inverse=False, tw_log_n=0, tw_col0=0, stream=None):
    """Length-2^log_len transforms along axis 0 of a row-major [2^log_len][cols] matrix, optional four-step twiddle."""
    nbytes = 32 * (cols << log_len)
    _chk(lib().zkp_ntt_fr_axis0_dev(_dev_ptr(t_in, nbytes), _dev_ptr(t_out, nbytes), log_len, cols, int(inverse), tw_log_n,
                                    tw_col0, _stream_ptr(stream)))


def ntt_fr_layout_dev(t_in, t_out, log_n, batch, inverse=False, in_layout=None, out_layout=None, tw_log_n=0, tw_row0=0,
                      stream=None):
    """`batch` transforms with a gathered input / scattered output layout (NttLayout or None), optional four-step twiddle."""
    nbytes = 32 * (batch << log_n)
    li = C.byref(in_layout) if in_layout is not None else None
    lo = C.byref(out_layout) if out_layout is not None else None
    _chk(lib().zkp_ntt_fr_layout_dev(_dev_ptr(t_in, nbytes), _dev_ptr(t_out, nbytes), log_n, batch, int(inverse),
                                     C.cast(li, C.c_void_p) if li is not None else None,
                                     C.cast(lo, C.c_void_p) if lo is not None else None, tw_log_n, tw_row0, _stream_ptr(stream)))


NTT_NATURAL, NTT_K1SLAB, NTT_COLUMNS = 0, 1, 2


class NttShardGeometry(C.Structure):  # zkp_ntt_shard_geometry in include/zkp_hip.h
    _fields_ = [("slots", C.c_uint), ("log_n1", C.c_uint), ("log_n2", C.c_uint), ("chunks", C.c_uint), ("r1", C.c_size_t),
                ("r2", C.c_size_t), ("cw", C.c_size_t), ("slab", C.c_size_t)]


def ntt_fr_sharded_geometry(log_n, slots=0, chunks=0):
    """Split of the in-process multi-GPU transform: dict(slots, log_n1, log_n2, chunks, r1, r2, cw, slab)."""
    g = NttShardGeometry()
    _chk(lib().zkp_ntt_fr_sharded_geometry(log_n, slots, chunks, C.cast(C.byref(g), C.c_void_p)))
    return {k: int(getattr(g, k)) for k, _ in NttShardGeometry._fields_}


def ntt_fr_sharded_dev(slab_tensors, log_n, inverse=False, layout_in=NTT_NATURAL, layout_out=NTT_K1SLAB, chunks=0, streams=None):
    """zkp_ntt_fr_sharded_dev: one resident slab tensor per device slot (on that slot's device), transformed in place.
    streams: None (synchronous: returns when every device is done) or one torch stream / raw handle per slot (enqueue only)."""
    k = len(slab_tensors)
    nbytes = (32 << log_n) // k
    ptrs = (C.c_void_p * k)(*[_dev_ptr(t, nbytes).value for t in slab_tensors])
    sts = None
    if streams is not None:
        if len(streams) != k:
            raise ValueError("one stream per slot")
        sts = (C.c_void_p * k)(*[int(getattr(s, "cuda_stream", s)) for s in streams])
    _chk(lib().zkp_ntt_fr_sharded_dev(ptrs, log_n, int(bool(inverse)), layout_in, layout_out, chunks, sts))


def ntt_fr_sharded(data, inverse=False, coset=None, inplace=False):
    """zkp_ntt_fr_sharded: host vector, natural order in and out, over all device slots of the process."""
    a = _np(data, np.uint64, (-1, 4))
    if not inplace:
        a = a.copy()
    cs = _np(coset, np.uint64, (4,)) if coset is not None else None
    _chk(lib().zkp_ntt_fr_sharded(_ptr(a), _log2(a.shape[0]), int(bool(inverse)), _ptr(cs)))
    return a


def ntt_goldilocks_dev(tensor, log_n, batch=1, inverse=False, coset=None, stream=None):
    cs = _np(coset, np.uint64, (1,)) if coset is not None else None
    _chk(lib().zkp_ntt_goldilocks_dev(_dev_ptr(tensor, (8 << log_n) * batch), log_n, batch, int(bool(inverse)),
                                      _ptr(cs), _stream_ptr(stream)))


def fri_layer_eval(coeffs, coset, log_d):
    """FriLayer::from_poly evaluations (fri/src/fri_layer.rs:40-46); coset is a Montgomery-form u64."""
    coeffs = _np(coeffs, np.uint64).reshape(-1)
    out = np.zeros(1 << log_d, dtype=np.uint64)
    _chk(lib().zkp_fri_layer_eval(_ptr(coeffs), coeffs.size, int(coset), log_d, _ptr(out)))
    return out


def fri_fold(coeffs, r):
    coeffs = _np(coeffs, np.uint64).reshape(-1)
    out = np.zeros((coeffs.size + 1) // 2, dtype=np.uint64)
    _chk(lib().zkp_fri_fold(_ptr(coeffs), coeffs.size, int(r), _ptr(out)))
    return out


def fri_merkle_node_count(n):
    return int(lib().zkp_fri_merkle_node_count(n))


def fri_merkle_tree(leaves):
    """MerkleTree::new (fri/src/merkle_tree.rs:42-63): every level, concatenated; the root is the last element."""
    leaves = _np(leaves, np.uint64).reshape(-1)
    out = np.zeros(fri_merkle_node_count(leaves.size), dtype=np.uint64)
    _chk(lib().zkp_fri_merkle_tree(_ptr(leaves), leaves.size, _ptr(out)))
    return out


def fri_merkle_tree_dev(leaves_tensor, n, nodes_tensor, stream=None):
    _chk(lib().zkp_fri_merkle_tree_dev(_dev_ptr(leaves_tensor, 8 * n), n, _dev_ptr(nodes_tensor, 8 * fri_merkle_node_count(n)),
                                       _stream_ptr(stream)))


def fri_challenges(roots, const_val, num_queries):
    roots = _np(roots, np.uint64).reshape(-1)
    r_out = np.zeros(roots.size, dtype=np.uint64)
    q_out = np.zeros(num_queries, dtype=np.uint64)
    _chk(lib().zkp_fri_challenges(_ptr(roots), roots.size, int(const_val), num_queries, _ptr(r_out), _ptr(q_out)))
    return r_out, q_out


def fri_prove(coeffs, blowup_factor, num_queries):
    """generate_proof (fri/src/prover.rs:141-168) -> flat proof (layout in include/zkp_hip.h)."""
    coeffs = _np(coeffs, np.uint64).reshape(-1)
    p, words = C.c_void_p(), C.c_size_t()
    _chk(lib().zkp_fri_prove(_ptr(coeffs), coeffs.size, blowup_factor, num_queries, C.byref(p), C.byref(words)))
    try:
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(words.value,)).copy()
    finally:
        lib().zkp_free(p)


def fri_verify(proof):
    """verify (fri/src/verifier.rs:10-127): True, or raises ZkpError carrying the reference's error string."""
    proof = _np(proof, np.uint64).reshape(-1)
    _chk(lib().zkp_fri_verify(_ptr(proof), proof.size))
    return True


# ----------------------------------------------------------------------------- pairings / verifiers (host code)
def g2_generator():
    out = np.zeros(24, dtype=np.uint64)
    _chk(lib().zkp_g2_generator(_ptr(out)))
    return out


def g2_mul(q_xy, scalar, q_is_inf=0):
    q = _np(q_xy, np.uint64).reshape(24)
    s = _np(scalar, np.uint64).reshape(4)
    out = np.zeros(24, dtype=np.uint64)
    inf = C.c_uint8(0)
    _chk(lib().zkp_g2_mul(_ptr(q), int(q_is_inf), _ptr(s), _ptr(out), C.byref(inf)))
    return out, int(inf.value)


def pairing(p_xy, q_xy, p_is_inf=0, q_is_inf=0):
    """Reduced optimal ate pairing as 12 x 6 Montgomery limbs (tower order)."""
    p = _np(p_xy, np.uint64).reshape(12)
    q = _np(q_xy, np.uint64).reshape(24)
    out = np.zeros((12, 6), dtype=np.uint64)
    _chk(lib().zkp_pairing(_ptr(p), int(p_is_inf), _ptr(q), int(q_is_inf), _ptr(out)))
    return out


def kzg_verify(g2s_xy, commitment, opening, y, z):
    """KzgScheme::verify (kzg/src/scheme.rs:143-160); commitment / opening = (xy, is_inf)."""
    acc = C.c_int(0)
    g2s = _np(g2s_xy, np.uint64).reshape(24)
    c, w = _np(commitment[0], np.uint64).reshape(12), _np(opening[0], np.uint64).reshape(12)
    yy, zz = _np(y, np.uint64).reshape(4), _np(z, np.uint64).reshape(4)
    _chk(lib().zkp_kzg_verify(_ptr(g2s), _ptr(c), int(commitment[1]), _ptr(w), int(opening[1]), _ptr(yy), _ptr(zz), C.byref(acc)))
    return bool(acc.value)


def kzg_batch_verify(g2s_xy, commitments, points, openings, evals, r_primes):
    """KzgScheme::batch_verify (kzg/src/scheme.rs:215-245); commitments / openings: (n, 12) finite points."""
    g2s = _np(g2s_xy, np.uint64).reshape(24)
    cm, op = _np(commitments, np.uint64, (-1, 12)), _np(openings, np.uint64, (-1, 12))
    pts, ev, rp = _np(points, np.uint64, (-1, 4)), _np(evals, np.uint64, (-1, 4)), _np(r_primes, np.uint64, (-1, 4))
    acc = C.c_int(0)
    _chk(lib().zkp_kzg_batch_verify(_ptr(g2s), cm.shape[0], _ptr(cm), None, _ptr(pts), _ptr(op), None, _ptr(ev), _ptr(rp), C.byref(acc)))
    return bool(acc.value)


def kzg_aggregate_commitments(commitments, challenge, is_inf=None):
    """KzgScheme::aggregate_commitments (kzg/src/scheme.rs:187-202): sum_i challenge^i * C_i -> (affine, is_inf)."""
    cm = _np(commitments, np.uint64, (-1, 12))
    ch = _np(challenge, np.uint64).reshape(4)
    inf = _np(is_inf, np.uint8) if is_inf is not None else None
    out = np.zeros(12, dtype=np.uint64)
    oinf = C.c_uint8(0)
    _chk(lib().zkp_kzg_aggregate_commitments(_ptr(cm), _ptr(inf) if inf is not None else None, cm.shape[0], _ptr(ch), _ptr(out), C.byref(oinf)))
    return out, int(oinf.value)


class PlonkTranscript:
    """ChallengeGenerator<Sha256> (plonk/src/challenge.rs:22-77)."""

    def __init__(self):
        self._h = C.c_void_p()
        _chk(lib().zkp_plonk_transcript_create(C.byref(self._h)))

    def feed(self, xy, is_inf=0):
        xy = _np(xy, np.uint64).reshape(12)
        _chk(lib().zkp_plonk_transcript_feed(self._h, _ptr(xy), int(is_inf)))

    def challenges(self, n):
        out = np.zeros((n, 4), dtype=np.uint64)
        _chk(lib().zkp_plonk_transcript_challenges(self._h, n, _ptr(out)))
        return out

    def close(self):
        if self._h:
            lib().zkp_plonk_transcript_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close


def poly_mul_fr(a, b):
    a, b = _np(a, np.uint64, (-1, 4)), _np(b, np.uint64, (-1, 4))
    if a.shape[0] == 0 or b.shape[0] == 0:
        return np.zeros((0, 4), dtype=np.uint64)
    out = np.zeros((a.shape[0] + b.shape[0] - 1, 4), dtype=np.uint64)
    _chk(lib().zkp_poly_mul_fr(_ptr(a), a.shape[0], _ptr(b), b.shape[0], _ptr(out)))
    return out


# ----------------------------------------------------------------------------- KzgScheme mirror (kzg/src/scheme.rs)
def kzg_commit(srs, coeffs):
    coeffs = _np(coeffs, np.uint64, (-1, 4))
    out = np.zeros(12, dtype=np.uint64)
    inf = C.c_uint8(0)
    _chk(lib().zkp_kzg_commit(srs._h, _ptr(coeffs), coeffs.shape[0], _ptr(out), C.byref(inf)))
    return out, int(inf.value)


def kzg_open(srs, coeffs, z):
    coeffs, z = _np(coeffs, np.uint64, (-1, 4)), _np(z, np.uint64, (4,))
    out = np.zeros(12, dtype=np.uint64)
    ev = np.zeros(4, dtype=np.uint64)
    inf = C.c_uint8(0)
    _chk(lib().zkp_kzg_open(srs._h, _ptr(coeffs), coeffs.shape[0], _ptr(z), _ptr(out), C.byref(inf), _ptr(ev)))
    return (out, int(inf.value)), ev


class Srs:
    """kzg/src/srs.rs: g1_points = [s^i]G for i < circuit_size + 3, generated on the GPU and kept resident."""

    def __init__(self, g1_points_xy):
        self.g1_points_xy = _np(g1_points_xy, np.uint64, (-1, 12))
        self.bases = G1Bases.from_host(self.g1_points_xy)

    @classmethod
    def new_from_secret(cls, secret, circuit_size):  # srs.rs:48
        return cls(srs_g1(secret, circuit_size + 3))

    def g1_points(self):  # srs.rs:78 (the reference clones; here a view)
        return self.g1_points_xy


class KzgScheme:
    """Python face of csrc/kzg_host.hpp (same surface as kzg/src/scheme.rs:22-142)."""

    def __init__(self, srs, expand_bases=True):  # scheme.rs:34
        self.srs = srs
        if expand_bases:  # the SRS is fixed for the life of the scheme: pay the one-off expansion here
            srs.bases.precompute(0)

    def commit(self, coeffs):  # scheme.rs:49 / 63
        return kzg_commit(self.srs.bases, coeffs)

    commit_vector = commit

    def commit_para(self, para):  # scheme.rs:78
        return g1_mul(self.srs.g1_points_xy[0], 0, para)

    def open(self, coeffs, z):  # scheme.rs:108 / 132
        return kzg_open(self.srs.bases, coeffs, z)

    open_vector = open


# ----------------------------------------------------------------------------- PLONK prover rounds (plonk/src/prover.rs)
CIRCUIT_POLYS = ("q_m", "q_l", "q_r", "q_o", "q_c", "pi", "f_a", "f_b", "f_c", "s_sigma_1", "s_sigma_2", "s_sigma_3")
POLY_IDS = {"ax": 0, "bx": 1, "cx": 2, "z": 3, "r": 4, "w_zeta": 5, "w_zeta_omega": 6, "tx_compact": 7, "t": 8}


class _PlonkProof(C.Structure):  # zkp_plonk_proof in include/zkp_hip.h
    _fields_ = [("commit_xy", (C.c_uint64 * 12) * 9), ("commit_is_inf", C.c_uint8 * 9), ("bars", (C.c_uint64 * 4) * 6),
                ("u", C.c_uint64 * 4), ("degree", C.c_uint64)]


class PlonkProver:
    """Round-by-round face of generate_proof (plonk/src/prover.rs:61-293); blinders and challenges are inputs."""

    def __init__(self, srs_bases, log_n, circuit_polys, k1, k2):
        """circuit_polys: dict name -> (len,4) uint64 coefficient array (names in CIRCUIT_POLYS)."""
        arrs = [_np(circuit_polys[k], np.uint64, (-1, 4)) for k in CIRCUIT_POLYS]
        ptrs = (C.c_void_p * 12)(*[a.ctypes.data for a in arrs])
        lens = (C.c_size_t * 12)(*[a.shape[0] for a in arrs])
        k1, k2 = _np(k1, np.uint64, (4,)), _np(k2, np.uint64, (4,))
        self._bases = srs_bases  # keep alive
        self._h = C.c_void_p()
        self.n = 1 << log_n
        _chk(lib().zkp_plonk_prover_create(srs_bases._h, log_n, ptrs, lens, _ptr(k1), _ptr(k2), C.byref(self._h)))

    def prove(self, blinders):
        """generate_proof (plonk/src/prover.rs:61-293) with the reference's transcript; blinders = b1..b9 (9, 4)."""
        b = _np(blinders, np.uint64, (9, 4))
        out = _PlonkProof()
        _chk(lib().zkp_plonk_prove(self._h, _ptr(b), C.byref(out)))
        names = ("a", "b", "c", "z", "t_lo", "t_mid", "t_hi", "w_ev_x", "w_ev_wx")
        commits = {k: (np.array(out.commit_xy[i][:], dtype=np.uint64), int(out.commit_is_inf[i])) for i, k in enumerate(names)}
        bars = np.array([list(out.bars[i]) for i in range(6)], dtype=np.uint64)
        return {"commits": commits, "bars": bars, "u": np.array(out.u[:], dtype=np.uint64), "degree": int(out.degree)}

    def verify(self, g2s_xy, proof):
        """verify (plonk/src/verifier.rs:19-157) of a dict returned by prove(): 1 accepted, 0 pairing failed, -1 challenge mismatch."""
        out = _PlonkProof()
        names = ("a", "b", "c", "z", "t_lo", "t_mid", "t_hi", "w_ev_x", "w_ev_wx")
        for i, k in enumerate(names):
            xy, inf = proof["commits"][k]
            for q in range(12):
                out.commit_xy[i][q] = int(xy[q])
            out.commit_is_inf[i] = int(inf)
        for i in range(6):
            for q in range(4):
                out.bars[i][q] = int(proof["bars"][i][q])
        for q in range(4):
            out.u[q] = int(proof["u"][q])
        out.degree = int(proof["degree"])
        g2s = _np(g2s_xy, np.uint64).reshape(24)
        acc = C.c_int(0)
        _chk(lib().zkp_plonk_verify(self._h, _ptr(g2s), C.byref(out), C.byref(acc)))
        return int(acc.value)

    def close(self):
        if self._h:
            lib().zkp_plonk_prover_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _pts(xy, inf):
        return [(xy[i].copy(), int(inf[i])) for i in range(xy.shape[0])]

    def round1(self, blinders):  # b1..b6
        b = _np(blinders, np.uint64, (6, 4))
        xy, inf = np.zeros((3, 12), dtype=np.uint64), np.zeros(3, dtype=np.uint8)
        _chk(lib().zkp_plonk_round1(self._h, _ptr(b), _ptr(xy), _ptr(inf)))
        return self._pts(xy, inf)

    def round2(self, beta, gamma, blinders):  # b7..b9
        b = _np(blinders, np.uint64, (3, 4))
        beta, gamma = _np(beta, np.uint64, (4,)), _np(gamma, np.uint64, (4,))
        xy, inf = np.zeros((1, 12), dtype=np.uint64), np.zeros(1, dtype=np.uint8)
        _chk(lib().zkp_plonk_round2(self._h, _ptr(beta), _ptr(gamma), _ptr(b), _ptr(xy), _ptr(inf)))
        return self._pts(xy, inf)[0]

    def round3(self, alpha):
        alpha = _np(alpha, np.uint64, (4,))
        xy, inf = np.zeros((3, 12), dtype=np.uint64), np.zeros(3, dtype=np.uint8)
        deg = C.c_size_t(0)
        _chk(lib().zkp_plonk_round3(self._h, _ptr(alpha), _ptr(xy), _ptr(inf), C.byref(deg)))
        return self._pts(xy, inf), int(deg.value)

    def round4(self, zeta):
        zeta = _np(zeta, np.uint64, (4,))
        bars = np.zeros((6, 4), dtype=np.uint64)
        _chk(lib().zkp_plonk_round4(self._h, _ptr(zeta), _ptr(bars)))
        return bars

    def round5(self, v):
        v = _np(v, np.uint64, (4,))
        xy, inf = np.zeros((2, 12), dtype=np.uint64), np.zeros(2, dtype=np.uint8)
        _chk(lib().zkp_plonk_round5(self._h, _ptr(v), _ptr(xy), _ptr(inf)))
        return self._pts(xy, inf)

    def get_poly(self, name):
        ln = C.c_size_t(0)
        _chk(lib().zkp_plonk_get_poly(self._h, POLY_IDS[name], None, 0, C.byref(ln)))
        out = np.zeros((ln.value, 4), dtype=np.uint64)
        if ln.value:
            _chk(lib().zkp_plonk_get_poly(self._h, POLY_IDS[name], _ptr(out), ln.value, C.byref(ln)))
        return out
