"""Worker of tests/test_gpu_dist_rccl.py (own process: it creates a process group).  One rank, backend nccl (= RCCL): the four-step
transform is forced through the process group, so the asynchronous all_to_all_single handles, their waits on the compute stream and
the chunk pipeline run exactly as they do with 8 ranks; the MSM exchange goes through all_gather_into_tensor.  Prints OK."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "zkp-implementation_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
import zkp_hip as zkp  # noqa: E402
from zkp_hip import dist as zd  # noqa: E402
from zkp_hip import trapdoor  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
zkp.init(0)
ops = zd.TorchOps(zkp)
for log_n, chunks in ((16, 1), (20, 4), (22, 4)):
    n = 1 << log_n
    x = bench.rand_fr_tensor(torch, n, 0x2CC1 + log_n, dev)
    exp = x.clone().reshape(-1)
    zkp.ntt_fr_dev(exp, log_n)
    ph = {}
    y = zd.ntt_fr_distributed(x, log_n, False, ops=ops, chunks=chunks, force_collective=True, timings=ph)
    l1 = zd.four_step_split(log_n, 1)
    n1, n2 = 1 << l1, 1 << (log_n - l1)
    want = exp.reshape(n2, n1, 4).permute(1, 0, 2).contiguous().reshape(n, 4)      # k1-slab layout of one rank = [k1][k2]
    assert torch.equal(y, want), f"forward 2^{log_n}"
    back = zd.ntt_fr_distributed(y, log_n, True, ops=ops, chunks=chunks, force_collective=True, input_layout="k1slab")
    assert torch.equal(back, x), f"inverse 2^{log_n}"
    nat = zd.ntt_fr_distributed(x, log_n, False, ops=ops, chunks=chunks, force_collective=True, natural_output=True)
    assert torch.equal(nat.reshape(-1), exp), f"natural 2^{log_n}"
    assert set(zd.resolve_timings(ph)) >= {"4_row_ntt"}
    # one exchange per transform (columns layout in / out) through the same asynchronous all_to_all_single path
    C = zd.columns_chunks(log_n, 1, chunks)
    share = zd.columns_shard(x, log_n, 0, 1, C)
    ph1 = {}
    y1 = zd.ntt_fr_distributed(share, log_n, False, ops=ops, chunks=C, force_collective=True, input_layout="columns", timings=ph1)
    assert torch.equal(y1, want), f"one-exchange forward 2^{log_n}"
    assert "0_pack+1_all_to_all_columns(issue)" not in zd.resolve_timings(ph1)
    b1 = zd.ntt_fr_distributed(y1, log_n, True, ops=ops, chunks=C, force_collective=True, input_layout="k1slab", output_layout="columns")
    assert torch.equal(b1, share), f"one-exchange inverse 2^{log_n}"
# MSM exchange through RCCL: all-gather of the 192-byte partial of this (only) rank + EC-add combine
n = 1 << 14
wl = bench.MsmWorkload(zkp, torch, dev, 14, chunk=0, expand="auto")
res = zd.msm_g1_sharded(zkp, wl.bases, wl.scalars, n, device=dev)
assert bench.check_against_trapdoor(zkp, wl.limb_sums(), res)
dist.barrier()
dist.destroy_process_group()
print("OK rccl one-rank: four-step forward / mirrored inverse / natural output / one-exchange columns layout through all_to_all_single(async), msm all-gather")
