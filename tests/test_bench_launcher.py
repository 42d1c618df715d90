"""bench.py must never report a one-rank run as an N-GPU one (VERDICT r1 #2 / ADVICE): the launcher guard is a pure function of
(--gpus, environment) that runs before anything touches a GPU, and the known-answer helper bench.py uses for `bit_exact_full`
(zkp_hip/trapdoor.py) agrees with the oracle and the big-int model.  CPU only."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import bigmodel as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launcher_action_table():
    import bench
    assert bench.launcher_action(1, {}) == "run"
    assert bench.launcher_action(1, {"WORLD_SIZE": "1"}) == "run"
    assert bench.launcher_action(8, {"WORLD_SIZE": "8"}) == "run"
    assert bench.launcher_action(8, {}) == "spawn"           # plain `python bench.py --gpus 8`: start the ranks ourselves
    assert bench.launcher_action(2, {}) == "spawn"
    for gpus, ws in ((8, "1"), (1, "8"), (4, "2"), (2, "x")):
        assert bench.launcher_action(gpus, {"WORLD_SIZE": ws}).startswith("error")
    assert bench.launcher_action(0, {}).startswith("error")


def test_world_size_mismatch_exits_nonzero_before_any_gpu_call():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=120)
    assert p.returncode == 2 and "WORLD_SIZE=3" in p.stderr
    assert not any(line.startswith("{") for line in p.stdout.splitlines())


@pytest.mark.timeout(600)
def test_gpus_2_without_launcher_spawns_two_ranks_and_never_prints_a_one_gpu_line():
    """No GPU here: the two ranks it starts fail ("needs a GPU"), the exit status is non-zero and no JSON line appears."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=580)
    assert p.returncode != 0
    for line in p.stdout.splitlines():
        if line.startswith("{"):
            assert json.loads(line).get("n_gpus") == 2  # (only possible on a box that really has two GPUs)
    assert "needs a GPU" in p.stderr or "GPU" in p.stderr


def test_trapdoor_inner_product_and_expected_point(orc):
    import torch
    import zkp_hip as zkp
    from zkp_hip import trapdoor
    for n in (4096, 3000):  # a multiple of the batched-product split, and not
        s, k = orc.rand_fr(11, n), orc.rand_fr(12, n)
        e = trapdoor.fr_inner_product(torch.from_numpy(s.view(np.int64)), torch.from_numpy(k.view(np.int64)))
        si, ki = orc.fr_to_ints(s), orc.fr_to_ints(k)
        assert e == sum(a * b for a, b in zip(si, ki)) % M.R
    assert [e] == orc.fr_to_ints(orc.fr_inner_product(s, k).reshape(1, 4))
    assert np.array_equal(trapdoor.g1_generator_mont(), orc.g1_generator())
    exp, einf = orc.g1_mul(orc.g1_generator(), 0, orc.fr_from_ints([e])[0])
    got, ginf = trapdoor.expected_msm(zkp, e)  # zkp_g1_mul is host code: no device needed
    assert ginf == einf and np.array_equal(got, exp)
    # edge: limbs at their maxima (the float64 partial sums must stay exact)
    top = np.tile(orc.fr_from_ints([M.R - 1]), (5120, 1))
    e2 = trapdoor.fr_inner_product(torch.from_numpy(top.view(np.int64)), torch.from_numpy(top.view(np.int64)))
    assert e2 == 5120 * (M.R - 1) * (M.R - 1) % M.R
