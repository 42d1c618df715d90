"""world_size-2 gloo tests (CPU) of the multi-GPU MSM path: chunk sharding, all-gather of the 192-byte partials and the
EC-add combine of the product library.  The per-rank MSM itself is a GPU kernel, so here the oracle stands in for it
(tests may use the oracle as a checker/stand-in; the product's combine code is what is under test)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    try:
        for p in (ROOT, os.path.join(ROOT, "zkp-implementation_amd"), os.path.join(ROOT, "tests", "model")):
            if p not in sys.path:
                sys.path.insert(0, p)
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import zkp_hip
        from zkp_hip import dist as zd
        from oracle import oracle as orc
        ks = orc.rand_fr(0xBA5E, n)
        sc = orc.rand_fr(0x5EED, n)
        pts, _ = orc.g1_fixed_base_mul(ks)
        lo, hi = zd.shard_range(n, rank, world)
        # stand-in for zkp.msm_g1_partial_dev on this rank's chunk: affine result -> XYZZ with ZZ = ZZZ = 1
        xy, inf = orc.msm_pippenger(pts[lo:hi], None, sc[lo:hi])
        one = orc.fq_from_ints([1])[0]
        part = np.zeros(24, dtype=np.uint64)
        if not inf:
            part[:12] = xy
            part[12:18] = one
            part[18:24] = one
        parts = zd.allgather_partials(part)
        assert parts.shape == (world, 24)
        got, ginf = zkp_hip.g1_xyzz_sum(parts)
        exp, einf = orc.msm_pippenger(pts, None, sc)
        ok = bool(ginf == einf and np.array_equal(got, exp))
        # also: identity partials are neutral
        z = np.zeros((1, 24), dtype=np.uint64)
        got2, _ = zkp_hip.g1_xyzz_sum(np.concatenate([parts, z]))
        ok = ok and bool(np.array_equal(got2, exp))
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, ok, None))
    except Exception as e:  # pragma: no cover
        q.put((rank, False, repr(e)))


def test_shard_range_covers_everything():
    sys.path.insert(0, os.path.join(ROOT, "zkp-implementation_amd"))
    from zkp_hip import dist as zd
    for n in (0, 1, 7, 8, 1000, 1 << 20):
        for world in (1, 2, 3, 8):
            spans = [zd.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("n", [64, 301])
def test_msm_shard_allgather_combine_world2(n):
    import torch.multiprocessing as mp
    from oracle import oracle as orc
    orc.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, err in res:
        assert ok, f"rank {rank}: {err}"
