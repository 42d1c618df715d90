"""GPU parity of the FRI commitment path (Merkle trees, generate_proof) against the oracle, through the C ABI."""
import numpy as np
import pytest

import bigmodel as M
from test_fri_oracle import canon, mont

pytestmark = pytest.mark.gpu
GL = M.GL


@pytest.fixture(scope="module")
def zkp():
    import torch
    assert torch.cuda.is_available(), "no GPU"
    import zkp_hip
    zkp_hip.init()
    return zkp_hip


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 255, 256, 257, 1000, 4096, 70001])
def test_merkle_tree_vs_oracle(zkp, orc, n):
    leaves = orc.rand_gl(0x3E2C0000 + n, n)
    if n >= 5:  # zero (prints as the empty string), one, p - 1 and a 19-digit value among the leaves
        leaves[:4] = mont([0, 1, GL - 1, 10 ** 18])
    got = zkp.fri_merkle_tree(leaves)
    assert got.shape[0] == orc.merkle_node_count(n)
    assert np.array_equal(got, orc.merkle_tree(leaves))


def test_merkle_tree_golden_small(zkp):
    # independent of the oracle: hashlib model on the reference's own test leaves (fri/src/merkle_tree.rs:141-151)
    levels = M.merkle_levels([1, 2, 3, 4])
    assert canon(zkp.fri_merkle_tree(mont([1, 2, 3, 4]))) == [v for l in levels for v in l]


def test_merkle_tree_dev_large_root_property(zkp, orc):
    """2^21 leaves (a 2^20-coefficient polynomial at blowup 2): the tree built on the GPU verifies along random paths
    with the oracle's hash, and its upper levels equal the oracle's tree over the GPU's level 10."""
    import torch
    n = 1 << 21
    leaves = orc.rand_gl(77, n)
    d_leaves = torch.from_numpy(leaves.view(np.int64)).cuda()
    d_nodes = torch.zeros(zkp.fri_merkle_node_count(n), dtype=torch.int64, device="cuda")
    zkp.fri_merkle_tree_dev(d_leaves, n, d_nodes)
    torch.cuda.synchronize()
    nodes = d_nodes.cpu().numpy().view(np.uint64)
    off = [0]
    for l in range(22):
        off.append(off[-1] + (n >> l))
    rnd = np.random.default_rng(3)
    for idx in rnd.integers(0, n, 8):
        cur, h = int(idx), orc.gl_hash(leaves[idx:idx + 1])[0]
        for l in range(21):
            assert nodes[off[l] + cur] == h
            sib = nodes[off[l] + (cur ^ 1)]
            h = orc.gl_hash_slice(np.array([h, sib] if cur % 2 == 0 else [sib, h], dtype=np.uint64))
            cur //= 2
        assert nodes[off[21]] == h and off[22] == nodes.shape[0]
    lvl10 = nodes[off[10]:off[11]]  # 2048 nodes: a whole level rebuilt with the oracle's hash_slice
    pairs = np.array([orc.gl_hash_slice(lvl10[2 * j:2 * j + 2]) for j in range(1024)], dtype=np.uint64)
    assert np.array_equal(pairs, nodes[off[11]:off[12]])


@pytest.mark.parametrize("d,blowup,nq", [(4, 2, 2), (6, 2, 2), (1, 1, 3), (1, 2, 1), (39, 4, 5), (300, 2, 8), (1024, 4, 4)])
def test_fri_prove_vs_oracle(zkp, orc, d, blowup, nq):
    coeffs = mont(list(range(1, d + 1))) if d <= 6 else orc.rand_gl(0xF21 + d, d)
    got = zkp.fri_prove(coeffs, blowup, nq)
    want = orc.fri_prove(coeffs, blowup, nq)
    assert np.array_equal(got, want)
    assert zkp.fri_verify(got) and orc.fri_verify(got) == 0


def test_fri_prove_trailing_zeros_and_zero_polynomial(zkp, orc):
    c = np.concatenate([orc.rand_gl(5, 10), np.zeros(6, dtype=np.uint64)])  # from_coefficients_vec trims
    assert np.array_equal(zkp.fri_prove(c, 2, 3), orc.fri_prove(c[:10], 2, 3))
    with pytest.raises(zkp.ZkpError) as ei:
        zkp.fri_prove(np.zeros(4, dtype=np.uint64), 2, 1)
    assert ei.value.code == zkp.ZKP_E_ARG


@pytest.mark.parametrize("log_d", [16, 19])
def test_fri_prove_large_verifies(zkp, orc, log_d):
    """2^16 / 2^19 coefficients, blowup 2 (domains 2^17 / 2^20; from 2^18 leaves on the Merkle kernel hashes four inputs per
    lane): too large for the oracle's Horner prover; the proof must verify under the oracle's verifier and the library's
    own, and break when a folding evaluation is altered."""
    L = log_d + 1
    coeffs = orc.rand_gl(0xB16 + log_d, 1 << log_d)
    proof = zkp.fri_prove(coeffs, 2, 10)
    assert int(proof[0]) == 1 << L and int(proof[1]) == L
    assert orc.fri_verify(proof) == 0 and zkp.fri_verify(proof)
    bad = proof.copy()
    bad[4 + L + 1 + 1] ^= np.uint64(1)
    assert orc.fri_verify(bad) != 0
    # layer 0 evaluations inside the proof equal direct evaluation of the polynomial at coset * w^index
    idx = int(proof[4 + L + 1])
    x = pow(M.root_of_unity(L, GL), idx, GL) * 7 % GL
    acc = 0
    for c in reversed(canon(coeffs)):
        acc = (acc * x + c) % GL
    assert canon([proof[4 + L + 2]])[0] == acc


def test_gpu_matches_committed_fri_golden(zkp, golden):
    """Committed vectors from the hashlib / big-int model (tests/golden/gen_golden.py), no oracle involved."""
    g = golden["fri_commit"]
    hx = lambda s: int(s, 16)
    for ent in g["merkle"]:
        assert canon(zkp.fri_merkle_tree(mont([hx(v) for v in ent["leaves"]]))) == [hx(v) for v in ent["nodes"]]
    for ent in g["proofs"]:
        flat = zkp.fri_prove(mont([hx(v) for v in ent["coeffs"]]), ent["blowup"], ent["queries"])
        L, nq = int(flat[1]), int(flat[2])
        is_field = [False, False, False, True] + [True] * (L + 1)
        for _ in range(nq if L else 0):
            for l in range(L):
                is_field += [False] + [True] * (2 + 2 * (L - l))
        plain = [canon([v])[0] if f else int(v) for v, f in zip(flat, is_field)]
        assert plain == [hx(v) for v in ent["flat_canonical"]]


def test_zero_display_switch(zkp, orc, monkeypatch):
    """ark-ff's Display prints zero as the empty string (the default here); ZKP_FRI_ZERO_AS_0=1 switches product and
    oracle to "0" together, should a run of the Rust reference ever show the other behaviour."""
    leaves = mont([0, 1, 2, 0, 5])
    default = zkp.fri_merkle_tree(leaves)
    assert np.array_equal(default, orc.merkle_tree(leaves))
    coeffs = mont([0, 0, 3, 0, 7])  # layers with zero evaluations / a transcript that digests F::ZERO first
    p_default = zkp.fri_prove(coeffs, 2, 2)
    monkeypatch.setenv("ZKP_FRI_ZERO_AS_0", "1")
    switched = zkp.fri_merkle_tree(leaves)
    assert np.array_equal(switched, orc.merkle_tree(leaves)) and not np.array_equal(switched, default)
    p_switched = zkp.fri_prove(coeffs, 2, 2)
    assert np.array_equal(p_switched, orc.fri_prove(coeffs, 2, 2)) and not np.array_equal(p_switched, p_default)
    assert zkp.fri_verify(p_switched)
    with pytest.raises(zkp.ZkpError):
        zkp.fri_verify(p_default)  # made under the other convention
