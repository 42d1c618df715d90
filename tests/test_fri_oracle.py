"""CPU tests of the FRI commitment-path oracle (oracle/fri_oracle.c) against the independent Python model
(tests/model/bigmodel.py), public known-answer vectors (NIST SHA-256, RFC 8439 ChaCha) and the reference's own tests
(fri/src/merkle_tree.rs:141-151, fri/src/verifier.rs:128-170, fri/src/prover.rs:195-221)."""
import hashlib

import numpy as np
import pytest

import bigmodel as M

GL = M.GL
R64 = 2 ** 64 % GL


def mont(vals):
    return np.array([v * R64 % GL for v in vals], dtype=np.uint64)


def canon(arr):
    rinv = pow(R64, -1, GL)
    return [int(v) * rinv % GL for v in np.asarray(arr, dtype=np.uint64).reshape(-1)]


def test_sha256_vectors(orc):
    assert orc.sha256(b"abc").hex() == "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad"
    assert orc.sha256(b"").hex() == "e3b0c44298fc1c149afbf4c8996fb92427ae41e4649b934ca495991b7852b855"
    rnd = np.random.default_rng(1)
    for n in (1, 55, 56, 63, 64, 65, 119, 120, 200, 1000):
        msg = rnd.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert orc.sha256(msg) == hashlib.sha256(msg).digest()


def test_chacha_rfc8439_block_and_model(orc):
    # RFC 8439 section 2.3.2: key 00..1f, counter 1, nonce 00:00:00:09:00:00:00:4a:00:00:00:00, 20 rounds
    key = [int.from_bytes(bytes(range(4 * i, 4 * i + 4)), "little") for i in range(8)]
    counter, stream = 1 | 0x09000000 << 32, 0x4A000000
    want = [0xE4E7F110, 0x15593BD1, 0x1FDD0F50, 0xC47120A3, 0xC7F4D1C7, 0x0368C033, 0x9AAA2204, 0x4E6CD4C3,
            0x466482D2, 0x09AA9F07, 0x05D7C214, 0xA2028BD9, 0xD19C12B5, 0xB94E16DE, 0xE883D0CB, 0x4E3C50A2]
    assert [int(x) for x in orc.chacha_block(key, counter, stream, 20)] == want
    assert M.chacha_block(key, counter, stream, 20) == want
    assert [int(x) for x in orc.chacha_block(key, 7, 0, 12)] == M.chacha_block(key, 7, 0, 12)


def test_stdrng_and_field_sampling_match_model(orc):
    for seed in (0, 1, 0xDEADBEEFCAFEF00D, 2 ** 64 - 1):
        rng = M.StdRng(seed)
        assert [int(x) for x in orc.stdrng_u64(seed, 40)] == [rng.next_u64() for _ in range(40)]  # crosses a block refill
        rng = M.StdRng(seed)
        got = orc.fr_rand_from_seed(seed, 5)
        for row in got:
            assert sum(int(l) << (64 * i) for i, l in enumerate(row)) == rng.rand_field(M.R, 4)


def test_hash_and_merkle_match_model(orc):
    vals = [0, 1, 9, 10, GL - 1, 12345678901234567890 % GL, 2 ** 63, 7]
    assert canon(orc.gl_hash(mont(vals))) == [M.gl_hash_slice([v]) for v in vals]
    assert canon([orc.gl_hash_slice(mont(vals[:3]))]) == [M.gl_hash_slice(vals[:3])]
    # "12" + "3" and "1" + "23" hash alike: the strings are concatenated without a separator (hasher.rs:32)
    assert orc.gl_hash_slice(mont([12, 3])) == orc.gl_hash_slice(mont([1, 23]))
    rnd = np.random.default_rng(2)
    for n in (1, 2, 3, 4, 5, 8, 13, 64, 100):
        leaves = [int(x) % GL for x in rnd.integers(0, 2 ** 63, n)]
        levels = M.merkle_levels(leaves)
        assert orc.merkle_node_count(n) == sum(len(l) for l in levels)
        assert canon(orc.merkle_tree(mont(leaves))) == [v for l in levels for v in l]


def test_reference_merkle_test(orc):
    # fri/src/merkle_tree.rs:141-151: leaves 1..4, the proof for index 1 verifies
    levels = M.merkle_levels([1, 2, 3, 4])
    path, cur, h = M.merkle_path(levels, 1), 1, M.gl_hash_slice([2])
    for nb in path:
        h = M.gl_hash_slice([h, nb] if cur % 2 == 0 else [nb, h])
        cur //= 2
    assert h == levels[-1][0] == canon(orc.merkle_tree(mont([1, 2, 3, 4])))[-1]


@pytest.mark.parametrize("coeffs,blowup,nq", [([1, 2, 3, 4], 2, 2), ([1, 2, 3, 4, 5, 6], 2, 2), ([5], 1, 3), ([0, 0, 9], 4, 1),
                                              (list(range(1, 40)), 2, 5)])
def test_fri_prove_matches_model_and_verifies(orc, coeffs, blowup, nq):
    # first two cases = fri/src/verifier.rs:128-156
    proof = orc.fri_prove(mont(coeffs), blowup, nq)
    model = M.fri_prove(coeffs, blowup, nq)
    assert [int(x) for x in proof] == M.fri_flatten(model, lambda v: v * R64 % GL)
    assert orc.fri_verify(proof) == 0
    r, q = orc.fri_challenges(proof[4:4 + int(proof[1])], proof[4 + int(proof[1])], nq)
    t = M.FriTranscript()
    want_r = []
    for root in model["roots"]:
        t.digest(root)
        want_r.append(t.challenge())
    assert canon(r) == want_r


def test_fri_reference_commit_phase_values(orc):
    # fri/src/prover.rs:195-205: layer 1 has coset 49 and domain size 2 -- visible in the proof as index ranges/paths
    proof = M.fri_prove([1, 2, 3, 4], 1, 1)  # domain 4 as in the reference test (folding_phase called with 4)
    assert proof["domain_size"] == 4 and len(proof["roots"]) == 2
    evals1 = M.fri_layer_eval(M.fri_fold([1, 2, 3, 4], M_first_challenge([1, 2, 3, 4])), 49, 2)
    assert M.merkle_levels(evals1)[-1][0] == proof["roots"][1]
    idx0, sym0 = proof["queries"][0][0][0], (proof["queries"][0][0][0] + 2) % 4
    assert (idx0 + 2) % 4 == sym0  # prover.rs:208-221


def M_first_challenge(coeffs):
    t = M.FriTranscript()
    t.digest(M.merkle_levels(M.fri_layer_eval(coeffs, 7, 4))[-1][0])
    return t.challenge()


def test_fri_verify_rejects_tampering(orc):
    proof = orc.fri_prove(mont([1, 2, 3, 4]), 2, 2)
    L = int(proof[1])
    bad = proof.copy()
    bad[4 + L] = mont([(canon([bad[4 + L]])[0] - 1) % GL])[0]  # verifier.rs:159-170: const_val -= 1
    assert orc.fri_verify(bad) != 0
    bad = proof.copy()
    bad[4] ^= np.uint64(1)  # a root
    assert orc.fri_verify(bad) != 0
    bad = proof.copy()
    bad[4 + L + 1 + 1] ^= np.uint64(2)  # first query's first evaluation
    assert orc.fri_verify(bad) != 0
    bad = proof.copy()
    bad[-1] ^= np.uint64(4)  # last sibling hash
    assert orc.fri_verify(bad) != 0
    assert orc.fri_verify(proof[:-1]) != 0


def test_fri_zero_polynomial_is_refused(orc):
    assert orc.fri_prove(mont([0, 0]), 2, 1) is None  # assert_eq!(poly.len(), 1) at fri/src/prover.rs:72


def test_oracle_matches_committed_fri_golden(orc, golden):
    g = golden["fri_commit"]
    hx = lambda s: int(s, 16)
    for ent in g["hash"]:
        assert canon(orc.gl_hash(mont([hx(ent["in"])]))) == [hx(ent["out"])]
    assert canon([orc.gl_hash_slice(mont([hx(v) for v in g["hash_pair"]["in"]]))]) == [hx(g["hash_pair"]["out"])]
    for ent in g["merkle"]:
        assert canon(orc.merkle_tree(mont([hx(v) for v in ent["leaves"]]))) == [hx(v) for v in ent["nodes"]]
    assert [int(x) for x in orc.stdrng_u64(hx(g["stdrng_seed"]), 40)] == [hx(v) for v in g["stdrng_u64"]]
    for ent in g["proofs"]:
        flat = orc.fri_prove(mont([hx(v) for v in ent["coeffs"]]), ent["blowup"], ent["queries"])
        L, nq = int(flat[1]), int(flat[2])
        want = [hx(v) for v in ent["flat_canonical"]]
        # header words and indices are plain integers; everything else is a field element in memory form
        got = [int(v) for v in flat]
        is_field = [False, False, False, True] + [True] * (L + 1)
        p = 4 + L + 1
        for _ in range(nq if L else 0):
            for l in range(L):
                is_field += [False] + [True] * (2 + 2 * (L - l))
        assert len(is_field) == len(got) == len(want)
        plain = [canon([v])[0] if f else v for v, f in zip(got, is_field)]
        assert plain == want
