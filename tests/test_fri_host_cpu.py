"""Host-only pieces of the FRI / PLONK transcript ABI (no GPU needed): challenges, verifier, PLONK challenge generator,
checked against the oracle and the Python model."""
import hashlib

import numpy as np
import pytest

import bigmodel as M
from test_fri_oracle import canon, mont

GL = M.GL


@pytest.fixture(scope="module")
def zkp():
    import zkp_hip
    zkp_hip.lib()
    return zkp_hip


def test_fri_challenges_match_oracle(zkp, orc):
    roots = mont([5, 0, GL - 1, 123456789])
    r1, q1 = zkp.fri_challenges(roots, mont([77])[0], 6)
    r2, q2 = orc.fri_challenges(roots, mont([77])[0], 6)
    assert np.array_equal(r1, r2) and np.array_equal(q1, q2)


@pytest.mark.parametrize("coeffs,blowup,nq", [([1, 2, 3, 4], 2, 2), ([1, 2, 3, 4, 5, 6], 2, 2), ([5], 1, 3),
                                              (list(range(1, 40)), 4, 5)])
def test_fri_verify_accepts_oracle_proofs_and_rejects_tampering(zkp, orc, coeffs, blowup, nq):
    proof = orc.fri_prove(mont(coeffs), blowup, nq)
    assert zkp.fri_verify(proof)
    L = int(proof[1])
    # const_val feeds the transcript, so changing it (verifier.rs:159-170) already moves the query indices
    for pos, msg in ((4 + L, ""), (len(proof) - 1, "verify Merkle path failed!")):
        if L == 0:
            continue
        bad = proof.copy()
        bad[pos] = mont([(canon([bad[pos]])[0] + 1) % GL])[0]
        with pytest.raises(zkp.ZkpError) as ei:
            zkp.fri_verify(bad)
        assert msg in str(ei.value)
    if L:
        bad = proof.copy()
        bad[4 + L + 1] += np.uint64(1)  # index of the first record
        with pytest.raises(zkp.ZkpError) as ei:
            zkp.fri_verify(bad)
        assert "wrong index!" in str(ei.value)
    with pytest.raises(zkp.ZkpError):
        zkp.fri_verify(proof[:-1])


def test_plonk_transcript_matches_restatement(zkp, orc):
    """plonk/src/challenge.rs: SHA-256 chain over serialize_uncompressed points, seed = first 8 bytes LE."""
    pts = [M.g1_mul(M.G1, k) for k in (1, 17, 123456789)]
    xy, inf = orc.points_from_ints(pts)
    t = zkp.PlonkTranscript()
    with pytest.raises(zkp.ZkpError):  # nothing fed yet: the reference's expect("No data ...") panics
        t.challenges(1)
    data = b""
    for i, p in enumerate(pts):
        t.feed(xy[i], 0)
        data = hashlib.sha256(data + p[0].to_bytes(48, "big") + p[1].to_bytes(48, "big")).digest()
    got = t.challenges(3)
    seed = int.from_bytes(data[:8], "little")
    assert np.array_equal(got, orc.fr_rand_from_seed(seed, 3))
    rng = M.StdRng(seed)
    for row in got:
        assert sum(int(l) << (64 * k) for k, l in enumerate(row)) == rng.rand_field(M.R, 4)
    with pytest.raises(zkp.ZkpError) as ei:  # "I'm hungry! Feed me something first"
        t.challenges(1)
    assert "hungry" in str(ei.value)
    t.feed(np.zeros(12, dtype=np.uint64), 1)  # infinity: 96 zero bytes with bit 6 of byte 0
    data = hashlib.sha256(data + bytes([0x40]) + bytes(95)).digest()
    assert np.array_equal(t.challenges(2), orc.fr_rand_from_seed(int.from_bytes(data[:8], "little"), 2))
    t.close()
