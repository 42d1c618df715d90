"""Independent big-integer model of the MSM / NTT hot path (TEST INFRASTRUCTURE ONLY).

This is the *mathematical definition* of every object the hot path produces, written with
Python integers and no shared code with either the C oracle (`oracle/`) or the HIP product
(`zkp-implementation_amd/`).  It exists to (1) reproduce the reference's own known-answer tests
 and (2) generate the small golden fixtures under `tests/golden/`.

Reference anchors (file:line under /root/reference):
  * kzg/src/scheme.rs:84-96      evaluate_in_s  -> `msm_naive`
  * kzg/src/srs.rs:48-69         Srs::new_from_secret -> `srs`
  * kzg/src/scheme.rs:108-120    open -> `kzg_open`
  * fri/src/fri_layer.rs:36-56   FriLayer::from_poly -> `fri_layer_eval`
  * fri/src/prover.rs:34-42      fold_polynomial -> `fri_fold`
  * plonk/src/slice_polynomial.rs:22-70 -> `slice_poly`, `slice_compact`
  * ark-poly 0.4.2 (not vendored) radix-2 domain semantics: omega_n = ROOT^(2^(32-log n)),
    natural order in/out, ifft scales by n^-1 -> `ntt`, `intt`, `coset_ntt`, `coset_intt`.
"""

# ----------------------------------------------------------------------------- constants
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
GX = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
GY = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1
G1 = (GX, GY)
INF = None  # point at infinity

FR_GENERATOR = 7
FR_TWO_ADICITY = 32
FR_ROOT = pow(FR_GENERATOR, (R - 1) >> FR_TWO_ADICITY, R)

GL = 2**64 - 2**32 + 1  # Goldilocks, fri/src/fields/goldilocks.rs:5
GL_GENERATOR = 7        # fri/src/fields/goldilocks.rs:6
GL_TWO_ADICITY = 32
GL_ROOT = pow(GL_GENERATOR, (GL - 1) >> GL_TWO_ADICITY, GL)

FR_MONT_R = (1 << 256) % R
FQ_MONT_R = (1 << 384) % P
GL_MONT_R = (1 << 64) % GL


# ----------------------------------------------------------------------------- limbs / Montgomery
def to_limbs(x, n):
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def from_limbs(limbs):
    return sum(int(l) << (64 * i) for i, l in enumerate(limbs))


def fr_to_mont(x):
    return (x * FR_MONT_R) % R


def fr_from_mont(x):
    return (x * pow(FR_MONT_R, -1, R)) % R


def fq_to_mont(x):
    return (x * FQ_MONT_R) % P


def fq_from_mont(x):
    return (x * pow(FQ_MONT_R, -1, P)) % P


def gl_to_mont(x):
    return (x * GL_MONT_R) % GL


def gl_from_mont(x):
    return (x * pow(GL_MONT_R, -1, GL)) % GL


# ----------------------------------------------------------------------------- G1 (affine, y^2 = x^3 + 4)
def g1_on_curve(pt):
    if pt is INF:
        return True
    x, y = pt
    return (y * y - x * x * x - 4) % P == 0


def g1_neg(pt):
    if pt is INF:
        return INF
    return (pt[0], (-pt[1]) % P)


def g1_add(a, b):
    if a is INF:
        return b
    if b is INF:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return INF
        lam = (3 * x1 * x1) * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    y3 = (lam * (x1 - x3) - y1) % P
    return (x3, y3)


def g1_mul(pt, k):
    k %= R
    acc = INF
    for bit in bin(k)[2:] if k else "":
        acc = g1_add(acc, acc)
        if bit == "1":
            acc = g1_add(acc, pt)
    return acc


def msm_naive(scalars, points):
    """kzg/src/scheme.rs:88-94: zip (truncating), per-term scalar-mul, left fold, empty -> identity."""
    acc = None
    first = True
    for c, s in zip(scalars, points):
        term = g1_mul(s, c)
        if first:
            acc, first = term, False
        else:
            acc = g1_add(acc, term)
    return INF if first else acc


def srs(secret, circuit_size):
    """kzg/src/srs.rs:48-63: [s^i]G for i < circuit_size + 3."""
    out, cur = [], 1
    for _ in range(circuit_size + 3):
        out.append(g1_mul(G1, cur))
        cur = cur * secret % R
    return out


# ----------------------------------------------------------------------------- polynomials over a prime field
def poly_trim(c):
    c = list(c)
    while c and c[-1] == 0:
        c.pop()
    return c


def poly_eval(c, x, mod):
    acc = 0
    for a in reversed(c):
        acc = (acc * x + a) % mod
    return acc


def poly_mul(a, b, mod):
    if not a or not b:
        return []
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                out[i + j] = (out[i + j] + x * y) % mod
    return poly_trim(out)


def poly_div_linear(c, z, mod):
    """(p(X) - p(z)) / (X - z) by synthetic division; kzg/src/scheme.rs:110-118."""
    n = len(c)
    if n <= 1:
        return []
    q = [0] * (n - 1)
    acc = 0
    for i in range(n - 1, 0, -1):
        acc = (c[i] + acc * z) % mod
        q[i - 1] = acc
    return poly_trim(q)


def kzg_open(coeffs, z, points):
    """kzg/src/scheme.rs:108-120 -> (opening point, evaluation)."""
    assert len(coeffs) >= 1, "at least 1"
    y = poly_eval(coeffs, z, R)
    q = poly_div_linear(coeffs, z, R)
    return msm_naive(q, points), y


def divide_by_vanishing(c, n, mod):
    """ark-poly divide_by_vanishing_poly for Z_H = X^n - 1 -> (quotient, remainder)."""
    c = list(c)
    if len(c) < n + 1:
        return [], poly_trim(c)
    q = [0] * (len(c) - n)
    for i in range(len(c) - 1, n - 1, -1):
        q[i - n] = c[i]
        c[i - n] = (c[i - n] + c[i]) % mod
    return poly_trim(q), poly_trim(c[:n])


def mul_by_vanishing(c, n, mod):
    out = [0] * (len(c) + n)
    for i, x in enumerate(c):
        out[i + n] = (out[i + n] + x) % mod
        out[i] = (out[i] - x) % mod
    return poly_trim(out)


# ----------------------------------------------------------------------------- NTT (ark-poly radix-2 domain semantics)
def root_of_unity(log_n, mod=R, root=None, two_adicity=32):
    if root is None:
        root = FR_ROOT if mod == R else GL_ROOT
    assert log_n <= two_adicity
    return pow(root, 1 << (two_adicity - log_n), mod)


def dft_naive(a, mod, omega):
    n = len(a)
    return [sum(a[j] * pow(omega, i * j, mod) for j in range(n)) % mod for i in range(n)]


def ntt(a, mod=R, inverse=False):
    """Natural-order in, natural-order out; inverse scales by n^-1."""
    n = len(a)
    log_n = n.bit_length() - 1
    assert 1 << log_n == n
    w = root_of_unity(log_n, mod)
    if inverse:
        w = pow(w, -1, mod)
    a = list(a)
    # bit reversal + iterative Cooley-Tukey
    j = 0
    for i in range(1, n):
        bit = n >> 1
        while j & bit:
            j ^= bit
            bit >>= 1
        j |= bit
        if i < j:
            a[i], a[j] = a[j], a[i]
    length = 2
    while length <= n:
        wl = pow(w, n // length, mod)
        for s in range(0, n, length):
            t = 1
            for k in range(length // 2):
                u, v = a[s + k], a[s + k + length // 2] * t % mod
                a[s + k], a[s + k + length // 2] = (u + v) % mod, (u - v) % mod
                t = t * wl % mod
        length <<= 1
    if inverse:
        ninv = pow(n, -1, mod)
        a = [x * ninv % mod for x in a]
    return a


def coset_ntt(a, g, mod=R):
    """Evaluate on g*<omega>: scale coefficient j by g^j, then NTT."""
    t, out = 1, []
    for x in a:
        out.append(x * t % mod)
        t = t * g % mod
    return ntt(out, mod)


def coset_intt(a, g, mod=R):
    c = ntt(a, mod, inverse=True)
    gi, t, out = pow(g, -1, mod), 1, []
    for x in c:
        out.append(x * t % mod)
        t = t * gi % mod
    return out


# ----------------------------------------------------------------------------- FRI (Goldilocks)
def fri_layer_eval(coeffs, coset, domain_size, mod=GL):
    """fri/src/fri_layer.rs:40-46: evals[i] = poly(omega_D^i * coset), natural order, by Horner."""
    log_d = domain_size.bit_length() - 1
    w = root_of_unity(log_d, mod)
    out, root = [], 1
    for _ in range(domain_size):
        out.append(poly_eval(coeffs, root * coset % mod, mod))
        root = root * w % mod
    return out


def fri_fold(coeffs, r, mod=GL):
    """fri/src/prover.rs:34-42: even + r*odd."""
    even = coeffs[0::2]
    odd = coeffs[1::2]
    out = [0] * max(len(even), len(odd))
    for i, x in enumerate(even):
        out[i] = x
    for i, x in enumerate(odd):
        out[i] = (out[i] + r * x) % mod
    return poly_trim(out)


# ----------------------------------------------------------------------------- plonk slicing
def slice_poly(coeffs, n_slices=3):
    """plonk/src/slice_polynomial.rs:22-43: chunk = ceil(len/3) consecutive coefficients."""
    ln = len(coeffs)
    chunk = (ln + n_slices - 1) // n_slices
    return [list(coeffs[i * chunk:(i + 1) * chunk]) for i in range(n_slices)], chunk - 1


def slice_compact(slices, degree, zeta, mod=R):
    """slice_polynomial.rs:56-70: sum_i zeta^{(degree+1) i} * slice_i."""
    ln = max(len(s) for s in slices)
    out = [0] * ln
    for i, s in enumerate(slices):
        f = pow(zeta, (degree + 1) * i, mod)
        for j, x in enumerate(s):
            out[j] = (out[j] + f * x) % mod
    return poly_trim(out)


# ----------------------------------------------------------------------------- deterministic PRNG shared with C/HIP
MASK64 = (1 << 64) - 1


def splitmix64(state):
    state = (state + 0x9E3779B97F4A7C15) & MASK64
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return state, z ^ (z >> 31)


def rand_fr_list(seed, n):
    """Uniform-ish canonical Fr values: 4 splitmix64 words, top limb masked to 255 bits, rejection >= r."""
    out, st = [], seed & MASK64
    while len(out) < n:
        limbs = []
        for _ in range(4):
            st, w = splitmix64(st)
            limbs.append(w)
        limbs[3] &= (1 << 63) - 1
        v = from_limbs(limbs)
        if v < R:
            out.append(v)
    return out


def rand_gl_list(seed, n):
    out, st = [], seed & MASK64
    while len(out) < n:
        st, w = splitmix64(st)
        if w < GL:
            out.append(w)
    return out


# ----------------------------------------------------------------------------- FRI commitment path (independent model)
# hashlib SHA-256, Python big ints; follows fri/src/hasher.rs, merkle_tree.rs, fiat_shamir/transcript.rs, prover.rs and
# verifier.rs.  Values are canonical integers.  Third-party behaviour (ark-ff Display / rand, rand_chacha) as documented
# in oracle/fri_oracle.c.
import hashlib
import struct

GL = 2 ** 64 - 2 ** 32 + 1


def gl_display(x, zero_as_0=False):
    return (str(x) if x else ("0" if zero_as_0 else "")).encode()


def gl_hash_slice(vals):
    h = hashlib.sha256(b"".join(gl_display(v) for v in vals)).digest()
    return int.from_bytes(h, "little") % GL


def merkle_levels(leaves):
    n = len(leaves)
    depth = (n - 1).bit_length() if n > 1 else 0
    levels = [[gl_hash_slice([v]) for v in leaves]]
    for _ in range(depth):
        prev = levels[-1]
        levels.append([gl_hash_slice(prev[i:i + 2]) for i in range(0, len(prev), 2)])
    return levels


def merkle_path(levels, index):
    path, cur = [], index
    for i in range(len(levels) - 1):
        path.append(levels[i][cur ^ 1])
        cur //= 2
    return path


def _rotl32(x, n):
    return ((x << n) | (x >> (32 - n))) & 0xFFFFFFFF


def chacha_block(key_words, counter, stream, rounds):
    s = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key_words) + [counter & 0xFFFFFFFF, counter >> 32,
                                                                                  stream & 0xFFFFFFFF, stream >> 32]
    x = list(s)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & 0xFFFFFFFF; x[d] = _rotl32(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & 0xFFFFFFFF; x[b] = _rotl32(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & 0xFFFFFFFF; x[d] = _rotl32(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & 0xFFFFFFFF; x[b] = _rotl32(x[b] ^ x[c], 7)

    for _ in range(rounds // 2):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(a + b) & 0xFFFFFFFF for a, b in zip(x, s)]


class StdRng:
    """rand 0.8 StdRng::seed_from_u64: PCG32-expanded key, ChaCha12, 64-bit block counter, stream 0."""

    def __init__(self, seed):
        state, key = seed, []
        for _ in range(8):
            state = (state * 6364136223846793005 + 11634580027462260723) & (2 ** 64 - 1)
            xs = (((state >> 18) ^ state) >> 27) & 0xFFFFFFFF
            rot = state >> 59
            key.append(((xs >> rot) | (xs << ((32 - rot) & 31))) & 0xFFFFFFFF)
        self.key, self.counter, self.buf = key, 0, []

    def next_u32(self):
        if not self.buf:
            self.buf = chacha_block(self.key, self.counter, 0, 12)
            self.counter += 1
        return self.buf.pop(0)

    def next_u64(self):
        lo = self.next_u32()
        return lo | self.next_u32() << 32

    def rand_field(self, modulus, nlimbs):
        """ark-ff UniformRand for Fp: returns the MONTGOMERY RESIDUE (the sampled integer itself)."""
        shave = 64 * nlimbs - modulus.bit_length()
        while True:
            limbs = [self.next_u64() for _ in range(nlimbs)]
            limbs[-1] &= (2 ** 64 - 1) >> shave
            v = sum(l << (64 * i) for i, l in enumerate(limbs))
            if v < modulus:
                return v


class FriTranscript:
    def __init__(self):
        self.data, self.index = b"", 0
        self.digest(0)

    def digest(self, canon):
        self.data = hashlib.sha256(self.data + struct.pack("<Q", self.index) + gl_display(canon)).digest()
        self.index += 1

    def rng(self):
        return StdRng(int.from_bytes(self.data[:8], "little"))

    def challenge(self):
        return self.rng().rand_field(GL, 1) * pow(2 ** 64, -1, GL) % GL  # canonical value of the sampled residue

    def challenge_list_usize(self, n):
        r = self.rng()
        return [r.rand_field(GL, 1) * pow(2 ** 64, -1, GL) % GL for _ in range(n)]


def fri_prove(coeffs, blowup, nq):
    """generate_proof (fri/src/prover.rs:141-168) on canonical integers; returns a dict."""
    poly = poly_trim(list(coeffs))
    dom = 1
    while dom < len(poly) * blowup:
        dom <<= 1
    layers_n = dom.bit_length() - 1
    t, coset, size = FriTranscript(), 7, dom
    layers = []
    for _ in range(layers_n):
        evals = fri_layer_eval(poly, coset, size)
        levels = merkle_levels(evals)
        t.digest(levels[-1][0])
        layers.append((evals, levels, size))
        poly = fri_fold(poly, t.challenge())
        coset, size = coset * coset % GL, size // 2
    assert len(poly) == 1
    t.digest(poly[0])
    queries = []
    for ch in ([c % dom for c in t.challenge_list_usize(nq)] if layers else []):
        rec = []
        for evals, levels, size in layers:
            idx = ch % size
            sym = (idx + size // 2) % size
            rec.append((idx, evals[idx], evals[sym], merkle_path(levels, idx), merkle_path(levels, sym)))
        queries.append(rec)
    return {"domain_size": dom, "coset": 7, "number_of_queries": nq, "roots": [l[1][-1][0] for l in layers],
            "const": poly[0], "queries": queries}


def fri_flatten(proof, to_mont):
    out = [proof["domain_size"], len(proof["roots"]), proof["number_of_queries"], to_mont(proof["coset"])]
    out += [to_mont(r) for r in proof["roots"]] + [to_mont(proof["const"])]
    for rec in proof["queries"]:
        for idx, ev, sv, path, spath in rec:
            out += [idx, to_mont(ev), to_mont(sv)] + [to_mont(x) for x in path] + [to_mont(x) for x in spath]
    return out
