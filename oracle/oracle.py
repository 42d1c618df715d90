"""ctypes view of oracle/libzkp_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (zkp-implementation_amd/) must never import it.

Data conventions (same as the C ABI in include/zkp_hip.h): numpy uint64 arrays holding
Montgomery residues, little-endian limbs.  Fr: (n,4)  Fq: (n,6)  Goldilocks: (n,)  G1 affine: (n,12)
plus a uint8 infinity array.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("ZKP_ORACLE_LIB", os.path.join(_HERE, "libzkp_oracle.so"))  # override: sanitizer builds

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
P_MOD = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
GL_MOD = 2**64 - 2**32 + 1


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("zkp_oracle.c", "fri_oracle.c", "zkp_oracle.h")]
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libzkp_oracle.so"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _u64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if shape is not None:
        a = a.reshape(shape)
    return a


# ----------------------------------------------------------------------------- int <-> limb helpers
def ints_to_limbs(vals, nlimbs):
    out = np.zeros((len(vals), nlimbs), dtype=np.uint64)
    for i, v in enumerate(vals):
        for k in range(nlimbs):
            out[i, k] = (v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    return out


def limbs_to_ints(arr):
    arr = np.asarray(arr, dtype=np.uint64)
    if arr.ndim == 1:
        arr = arr.reshape(1, -1)
    return [sum(int(x) << (64 * k) for k, x in enumerate(row)) for row in arr]


def _batch1(name, a, nl):
    a = _u64(a, (-1, nl))
    out = np.empty_like(a)
    getattr(lib(), name)(_p(a), _p(out), C.c_size_t(a.shape[0]))
    return out


def fr_from_ints(vals):
    """canonical python ints -> (n,4) Montgomery"""
    return _batch1("oracle_fr_to_mont", ints_to_limbs([v % R_MOD for v in vals], 4), 4)


def fr_to_ints(a):
    return limbs_to_ints(_batch1("oracle_fr_from_mont", a, 4))


def fq_from_ints(vals):
    return _batch1("oracle_fq_to_mont", ints_to_limbs([v % P_MOD for v in vals], 6), 6)


def fq_to_ints(a):
    return limbs_to_ints(_batch1("oracle_fq_from_mont", a, 6))


def gl_from_ints(vals):
    a = np.array([v % GL_MOD for v in vals], dtype=np.uint64)
    out = np.empty_like(a)
    lib().oracle_gl_to_mont(_p(a), _p(out), C.c_size_t(a.size))
    return out


def gl_to_ints(a):
    a = _u64(a).reshape(-1)
    out = np.empty_like(a)
    lib().oracle_gl_from_mont(_p(a), _p(out), C.c_size_t(a.size))
    return [int(x) for x in out]


def points_from_ints(pts):
    """list of (x,y) or None -> ((n,12) Montgomery, (n,) uint8 infinity)"""
    n = len(pts)
    xy = np.zeros((n, 12), dtype=np.uint64)
    inf = np.zeros(n, dtype=np.uint8)
    coords = []
    for p in pts:
        if p is None:
            coords += [0, 0]
        else:
            coords += [p[0], p[1]]
    m = fq_from_ints(coords).reshape(n, 12) if n else xy
    for i, p in enumerate(pts):
        if p is None:
            inf[i] = 1
        else:
            xy[i] = m[i]
    return xy, inf


def points_to_ints(xy, inf=None):
    xy = _u64(xy, (-1, 12))
    vals = fq_to_ints(xy.reshape(-1, 6))
    out = []
    for i in range(xy.shape[0]):
        if inf is not None and np.asarray(inf).reshape(-1)[i]:
            out.append(None)
        else:
            out.append((vals[2 * i], vals[2 * i + 1]))
    return out


# ----------------------------------------------------------------------------- field ops
def fr_mul(a, b):
    a, b = _u64(a, (-1, 4)), _u64(b, (-1, 4))
    out = np.empty_like(a)
    lib().oracle_fr_mul(_p(a), _p(b), _p(out), C.c_size_t(a.shape[0]))
    return out


def fr_add(a, b):
    a, b = _u64(a, (-1, 4)), _u64(b, (-1, 4))
    out = np.empty_like(a)
    lib().oracle_fr_add(_p(a), _p(b), _p(out), C.c_size_t(a.shape[0]))
    return out


def fr_sub(a, b):
    a, b = _u64(a, (-1, 4)), _u64(b, (-1, 4))
    out = np.empty_like(a)
    lib().oracle_fr_sub(_p(a), _p(b), _p(out), C.c_size_t(a.shape[0]))
    return out


def fr_inv(a):
    return _batch1("oracle_fr_inv", a, 4)


def fq_mul(a, b):
    a, b = _u64(a, (-1, 6)), _u64(b, (-1, 6))
    out = np.empty_like(a)
    lib().oracle_fq_mul(_p(a), _p(b), _p(out), C.c_size_t(a.shape[0]))
    return out


def fr_inner_product(a, b):
    a, b = _u64(a, (-1, 4)), _u64(b, (-1, 4))
    out = np.zeros(4, dtype=np.uint64)
    lib().oracle_fr_inner_product(_p(a), _p(b), C.c_size_t(a.shape[0]), _p(out))
    return out


def rand_fr(seed, n):
    out = np.empty((n, 4), dtype=np.uint64)
    lib().oracle_rand_fr(C.c_uint64(seed), C.c_size_t(n), _p(out))
    return out


def rand_gl(seed, n):
    out = np.empty(n, dtype=np.uint64)
    lib().oracle_rand_gl(C.c_uint64(seed), C.c_size_t(n), _p(out))
    return out


# ----------------------------------------------------------------------------- G1
def g1_generator():
    out = np.zeros(12, dtype=np.uint64)
    lib().oracle_g1_generator(_p(out))
    return out


def g1_mul(base_xy, base_inf, scalar):
    base_xy, scalar = _u64(base_xy, (12,)), _u64(scalar, (4,))
    out = np.zeros(12, dtype=np.uint64)
    inf = C.c_uint8(0)
    lib().oracle_g1_mul(_p(base_xy), C.c_uint8(int(base_inf)), _p(scalar), _p(out), C.byref(inf))
    return out, inf.value


def g1_add(a_xy, a_inf, b_xy, b_inf):
    a_xy, b_xy = _u64(a_xy, (12,)), _u64(b_xy, (12,))
    out = np.zeros(12, dtype=np.uint64)
    inf = C.c_uint8(0)
    lib().oracle_g1_add(_p(a_xy), C.c_uint8(int(a_inf)), _p(b_xy), C.c_uint8(int(b_inf)), _p(out), C.byref(inf))
    return out, inf.value


def g1_on_curve(xy, inf=0):
    xy = _u64(xy, (12,))
    return bool(lib().oracle_g1_on_curve(_p(xy), C.c_uint8(int(inf))))


def srs(secret, n):
    secret = _u64(secret, (4,))
    out = np.zeros((n, 12), dtype=np.uint64)
    lib().oracle_srs(_p(secret), C.c_size_t(n), _p(out))
    return out


def g1_fixed_base_mul(scalars):
    scalars = _u64(scalars, (-1, 4))
    n = scalars.shape[0]
    out = np.zeros((n, 12), dtype=np.uint64)
    inf = np.zeros(n, dtype=np.uint8)
    lib().oracle_g1_fixed_base_mul(_p(scalars), C.c_size_t(n), _p(out), _p(inf))
    return out, inf


def _msm(fn, points_xy, points_inf, scalars):
    points_xy, scalars = _u64(points_xy, (-1, 12)), _u64(scalars, (-1, 4))
    n = min(points_xy.shape[0], scalars.shape[0])  # zip truncation, kzg/src/scheme.rs:90-91
    if points_inf is not None:
        points_inf = np.ascontiguousarray(points_inf, dtype=np.uint8)
    out = np.zeros(12, dtype=np.uint64)
    inf = C.c_uint8(0)
    fn(_p(points_xy), _p(points_inf), _p(scalars), C.c_size_t(n), _p(out), C.byref(inf))
    return out, inf.value


def msm_naive(points_xy, points_inf, scalars):
    return _msm(lib().oracle_msm_naive, points_xy, points_inf, scalars)


def msm_pippenger(points_xy, points_inf, scalars):
    return _msm(lib().oracle_msm_pippenger, points_xy, points_inf, scalars)


def max_threads():
    return int(lib().oracle_max_threads())


def msm_pippenger_mt(points_xy, points_inf, scalars, threads):
    """CPU-best context baseline: the bucket method over `threads` cores (OpenMP).  Same result as msm_pippenger."""
    pts = _u64(points_xy, (-1, 12))
    sc = _u64(scalars, (-1, 4))
    inf = np.ascontiguousarray(points_inf, dtype=np.uint8) if points_inf is not None else None
    out, oinf = np.zeros(12, dtype=np.uint64), C.c_uint8(0)
    lib().oracle_msm_pippenger_mt(_p(pts), _p(inf), _p(sc), C.c_size_t(sc.shape[0]), C.c_int(int(threads)), _p(out), C.byref(oinf))
    return out, int(oinf.value)


def ntt_fr_mt(data, threads, inverse=False, coset=None, inplace=False):
    """CPU-best context baseline: oracle_ntt_fr's stages over `threads` cores.  Same outputs as ntt_fr."""
    a = _u64(data, (-1, 4))
    if not inplace:
        a = a.copy()
    log_n = a.shape[0].bit_length() - 1
    assert a.shape[0] == 1 << log_n
    cs = _u64(coset, (4,)) if coset is not None else None
    lib().oracle_ntt_fr_mt(_p(a), C.c_uint(log_n), C.c_int(int(inverse)), _p(cs), C.c_int(int(threads)))
    return a


# ----------------------------------------------------------------------------- NTT / poly / FRI
def ntt_fr(data, inverse=False, coset=None):
    a = _u64(data, (-1, 4)).copy()
    log_n = a.shape[0].bit_length() - 1
    assert a.shape[0] == 1 << log_n
    cs = _u64(coset, (4,)) if coset is not None else None
    lib().oracle_ntt_fr(_p(a), C.c_uint(log_n), C.c_int(int(inverse)), _p(cs))
    return a


def ntt_gl(data, inverse=False, coset=None):
    a = _u64(data).reshape(-1).copy()
    log_n = a.shape[0].bit_length() - 1
    assert a.shape[0] == 1 << log_n
    cs = _u64(coset, (1,)) if coset is not None else None
    lib().oracle_ntt_gl(_p(a), C.c_uint(log_n), C.c_int(int(inverse)), _p(cs))
    return a


def fr_root_of_unity(log_n):
    out = np.zeros(4, dtype=np.uint64)
    lib().oracle_fr_root_of_unity(C.c_uint(log_n), _p(out))
    return out


def poly_mul_fr(a, b):
    a, b = _u64(a, (-1, 4)), _u64(b, (-1, 4))
    if a.shape[0] == 0 or b.shape[0] == 0:
        return np.zeros((0, 4), dtype=np.uint64)
    out = np.zeros((a.shape[0] + b.shape[0] - 1, 4), dtype=np.uint64)
    lib().oracle_poly_mul_fr(_p(a), C.c_size_t(a.shape[0]), _p(b), C.c_size_t(b.shape[0]), _p(out))
    return out


def divide_by_vanishing_fr(c, n):
    c = _u64(c, (-1, 4))
    ln = c.shape[0]
    quot = np.zeros((max(ln - n, 0), 4), dtype=np.uint64)
    rem = np.zeros((n, 4), dtype=np.uint64)
    lib().oracle_divide_by_vanishing_fr(_p(c), C.c_size_t(ln), C.c_size_t(n), _p(quot), _p(rem))
    return quot, rem


def poly_eval_fr(c, z):
    c, z = _u64(c, (-1, 4)), _u64(z, (4,))
    out = np.zeros(4, dtype=np.uint64)
    lib().oracle_poly_eval_fr(_p(c), C.c_size_t(c.shape[0]), _p(z), _p(out))
    return out


def poly_div_linear_fr(c, z):
    c, z = _u64(c, (-1, 4)), _u64(z, (4,))
    quot = np.zeros((max(c.shape[0] - 1, 0), 4), dtype=np.uint64)
    lib().oracle_poly_div_linear_fr(_p(c), C.c_size_t(c.shape[0]), _p(z), _p(quot))
    return quot


def fri_layer_eval(coeffs, coset, log_d):
    coeffs = _u64(coeffs).reshape(-1)
    out = np.zeros(1 << log_d, dtype=np.uint64)
    lib().oracle_fri_layer_eval(_p(coeffs), C.c_size_t(coeffs.size), C.c_uint64(int(coset)), C.c_uint(log_d), _p(out))
    return out


def fri_fold(coeffs, r):
    coeffs = _u64(coeffs).reshape(-1)
    out = np.zeros((coeffs.size + 1) // 2, dtype=np.uint64)
    lib().oracle_fri_fold(_p(coeffs), C.c_size_t(coeffs.size), C.c_uint64(int(r)), _p(out))
    return out


# ----------------------------------------------------------------------------- FRI commitment path (fri_oracle.c)
def sha256(msg: bytes) -> bytes:
    out = (C.c_uint8 * 32)()
    lib().oracle_sha256(msg, C.c_size_t(len(msg)), out)
    return bytes(out)


def gl_hash(elems):
    a = _u64(elems, (-1,))
    out = np.zeros_like(a)
    lib().oracle_gl_hash(_p(a), C.c_size_t(a.shape[0]), _p(out))
    return out


def gl_hash_slice(elems):
    a = _u64(elems, (-1,))
    out = np.zeros(1, dtype=np.uint64)
    lib().oracle_gl_hash_slice(_p(a), C.c_size_t(a.shape[0]), _p(out))
    return out[0]


def merkle_node_count(n):
    lib().oracle_merkle_node_count.restype = C.c_size_t
    return int(lib().oracle_merkle_node_count(C.c_size_t(n)))


def merkle_tree(leaves):
    a = _u64(leaves, (-1,))
    out = np.zeros(merkle_node_count(a.shape[0]), dtype=np.uint64)
    lib().oracle_merkle_tree(_p(a), C.c_size_t(a.shape[0]), _p(out))
    return out


def chacha_block(key_words, counter, stream, rounds):
    k = np.ascontiguousarray(key_words, dtype=np.uint32)
    out = np.zeros(16, dtype=np.uint32)
    lib().oracle_chacha_block(_p(k), C.c_uint64(counter), C.c_uint64(stream), C.c_int(rounds), _p(out))
    return out


def stdrng_u64(seed, n):
    out = np.zeros(n, dtype=np.uint64)
    lib().oracle_stdrng_u64(C.c_uint64(seed), C.c_size_t(n), _p(out))
    return out


def fr_rand_from_seed(seed, n):
    out = np.zeros((n, 4), dtype=np.uint64)
    lib().oracle_fr_rand_from_seed(C.c_uint64(seed), C.c_size_t(n), _p(out))
    return out


def fri_challenges(roots, const_val, nq):
    r = _u64(roots, (-1,))
    r_out = np.zeros(r.shape[0], dtype=np.uint64)
    q_out = np.zeros(nq, dtype=np.uint64)
    lib().oracle_fri_challenges(_p(r), C.c_size_t(r.shape[0]), C.c_uint64(int(const_val)), C.c_size_t(nq), _p(r_out), _p(q_out))
    return r_out, q_out


def fri_proof_words(domain_size, nq):
    lib().oracle_fri_proof_words.restype = C.c_size_t
    return int(lib().oracle_fri_proof_words(C.c_size_t(domain_size), C.c_size_t(nq)))


def fri_prove(coeffs, blowup, nq):
    """Flat proof (layout in fri_oracle.c) or None where the reference panics (zero polynomial)."""
    c = _u64(coeffs, (-1,))
    d = c.shape[0]
    dom = 1
    while dom < d * blowup:
        dom <<= 1
    out = np.zeros(fri_proof_words(dom, nq), dtype=np.uint64)
    lib().oracle_fri_prove.restype = C.c_size_t
    words = int(lib().oracle_fri_prove(_p(c), C.c_size_t(d), C.c_size_t(blowup), C.c_size_t(nq), _p(out)))
    return out[:words] if words else None


def fri_verify(proof):
    p = _u64(proof, (-1,))
    return int(lib().oracle_fri_verify(_p(p), C.c_size_t(p.shape[0])))
