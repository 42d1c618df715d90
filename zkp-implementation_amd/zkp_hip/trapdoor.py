"""Known-answer check for full-size MSMs without a CPU oracle: when the bases are P_i = k_i G with known k_i (the benchmark's
and the tests' synthetic SRS), the exact result of sum s_i P_i is (sum s_i k_i) G -- the identity the reference's own
`commit` test uses with its known secret (kzg/src/commitment.rs:46-51).  The inner product over Fr is taken with exact
integer arithmetic on torch tensors (16-bit limbs, float64 matmuls whose partial sums stay below 2^53), the single scalar
multiplication by the library's host routine zkp_g1_mul (scheme.rs:78-82); nothing here touches the MSM kernels.
"""
import numpy as np

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
P_MOD = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
G1_X = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
G1_Y = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1
_SPLIT = 256
_CHUNK = 1 << 20  # rows per matmul: (2^16)^2 * 2^20 = 2^52 < 2^53, every partial sum is an exactly represented integer


def g1_generator_mont():
    """(12,) uint64: the generator's affine coordinates as arkworks Montgomery limbs (x * 2^384 mod p)."""
    out = np.empty(12, dtype=np.uint64)
    for j, v in enumerate((G1_X, G1_Y)):
        m = (v << 384) % P_MOD
        for i in range(6):
            out[6 * j + i] = (m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF
    return out


def fr_mont_limbs(v):
    """python int -> (4,) uint64 Montgomery residue."""
    x = (v << 256) % R_MOD
    return np.array([(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def _limbs16(t):
    """(n,4) int64 tensor of 64-bit limbs -> (n,16) float64 tensor of 16-bit limbs, least significant first."""
    import torch
    parts = [((t >> (16 * j)) & 0xFFFF) for j in range(4)]  # the mask discards the sign extension of the arithmetic shift
    return torch.stack(parts, dim=2).reshape(t.shape[0], 16).to(torch.float64)


def limb_products(a, b):
    """(16,16) array of python ints: C[x][y] = sum_i a_i[x] * b_i[y] over 16-bit limbs of the raw 256-bit memory words of two
    (n,4) int64 CUDA tensors -- the additive piece several ranks can sum before fr_inner_product_from_limbs."""
    import torch
    n = a.shape[0]
    acc = [[0] * 16 for _ in range(16)]
    for lo in range(0, n, _CHUNK):
        hi = min(n, lo + _CHUNK)
        la, lb = _limbs16(a[lo:hi]), _limbs16(b[lo:hi])
        rows = hi - lo
        if rows % _SPLIT == 0:  # a 16 x K x 16 product is one workgroup for the BLAS: cut K into many small products
            pa = la.reshape(_SPLIT, rows // _SPLIT, 16).transpose(1, 2)
            c = torch.bmm(pa, lb.reshape(_SPLIT, rows // _SPLIT, 16)).sum(dim=0)  # partial sums < 2^52: still exact
        else:
            c = la.T @ lb
        c = c.to(torch.int64).cpu().numpy()
        for x in range(16):
            for y in range(16):
                acc[x][y] += int(c[x, y])
    return acc


def fr_inner_product_from_limbs(acc):
    """sum of the true field values: the memory words are Montgomery residues s R and k R, so the limb sum is R^2 sum s k."""
    total = 0
    for x in range(16):
        for y in range(16):
            total += acc[x][y] << (16 * (x + y))
    r_inv = pow(1 << 256, -1, R_MOD)
    return total * r_inv % R_MOD * r_inv % R_MOD


def fr_inner_product(a, b):
    return fr_inner_product_from_limbs(limb_products(a, b))


def expected_msm(zkp, e):
    """e * G through the library's host scalar multiplication -> ((12,) uint64 affine, is_inf)."""
    return zkp.g1_mul(g1_generator_mont(), 0, fr_mont_limbs(e))
