"""Multi-GPU layer: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm); gloo on CPU for tests.

MSM shards by contiguous point/scalar chunk.  The only exchange is the final one: every rank's
partial sum (one extended-Jacobian point, 24 x u64 = 192 B) is all-gathered and the ranks add the partials locally
-- EC addition is not an RCCL reduction operator, so the "all-reduce of partial sums" is gather + local add.
"""
import numpy as np


def shard_range(n, rank, world):
    """Contiguous chunk [lo, hi) of n terms owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allgather_partials(partial, group=None, device=None):
    """partial: (24,) uint64 numpy (X, Y, ZZ, ZZZ).  Returns (world, 24) uint64 numpy on every rank."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return np.ascontiguousarray(partial, dtype=np.uint64).reshape(1, 24)
    world = dist.get_world_size(group)
    mine = torch.from_numpy(np.ascontiguousarray(partial, dtype=np.uint64).view(np.int64).copy())
    if device is not None:
        mine = mine.to(device)
    out = torch.empty(world * 24, dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(out, mine, group=group)
    return out.cpu().numpy().view(np.uint64).reshape(world, 24)


def msm_g1_sharded(zkp, bases_local, scalars_local, n_local, group=None, device=None, stream=None):
    """Every rank passes ITS chunk (bases resident on its GPU); every rank returns the full MSM (affine, is_inf)."""
    part = zkp.msm_g1_partial_dev(bases_local, scalars_local, n_local, stream=stream)
    parts = allgather_partials(part, group=group, device=device)
    return zkp.g1_xyzz_sum(parts)


# ----------------------------------------------------------------------------- four-step NTT across GPUs
# N = 2^log_n = N1 * N2 (N1 = 2^ceil(log_n/2)), global index n = n1 * N2 + n2, output index k = k1 + N1 * k2.
# Rank g owns the contiguous slab of rows n1 in [g N1/G, (g+1) N1/G) (i.e. its N/G consecutive elements).
#   1. all-to-all transpose: every rank gets complete columns (all n1 for its N2/G values of n2)
#   2. N2/G local column transforms of length N1                       (batched LDS-tiled kernel)
#   3. twiddle by omega_N^(n2 k1)                                        (zkp_ntt_fr_twiddle_dev)
#   4. all-to-all transpose back: rank g owns k1 in its slab, all n2
#   5. N1/G local row transforms of length N2
# Result on rank g: local[k1 - g N1/G][k2] = X[k1 + N1 k2]  ("k1-slab" layout); `natural_output=True` adds a third
# all-to-all so that rank g ends with X[g N/G .. (g+1) N/G).  The inverse transform runs the same steps with inverse
# kernels (the two local scalings 1/N1 and 1/N2 multiply to 1/N) and expects / produces the same layouts.
# xGMI is point-to-point: an all-to-all uses all 7 links of every GPU at once, which is why the exchange is an
# all-to-all of large contiguous blocks and not a ring.

class TorchOps:
    """Local kernels of the distributed transform on this rank's GPU (torch int64 tensors shaped [..., 4])."""

    def __init__(self, zkp):
        self.zkp = zkp

    def ntt_batch(self, t, log_len, batch, inverse):
        flat = t.reshape(-1)
        self.zkp.ntt_fr_dev(flat, log_len, batch=batch, inverse=inverse)
        return t

    def twiddle(self, t, rows, cols, row0, log_n, inverse):
        self.zkp.ntt_fr_twiddle_dev(t.reshape(-1), rows, cols, row0, log_n, inverse=inverse)
        return t


def _all_to_all(blocks, group):
    """blocks: list (len world) of equal-shape tensors to send; returns the list received (rank order)."""
    import torch
    import torch.distributed as dist
    world = len(blocks)
    if world == 1:
        return blocks
    send = torch.stack([b.contiguous() for b in blocks])
    recv = torch.empty_like(send)
    if dist.get_backend(group) != "gloo":
        dist.all_to_all_single(recv, send, group=group)  # RCCL: a failure here is a real one and must surface on this rank
        return [recv[r] for r in range(world)]
    try:
        dist.all_to_all_single(recv, send, group=group)
    except (RuntimeError, NotImplementedError):  # gloo builds without all_to_all (CPU tests only): gather everything, keep my column
        gathered = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(gathered, send, group=group)
        me = dist.get_rank(group)
        recv = torch.stack([gathered[r][me] for r in range(world)])
    return [recv[r] for r in range(world)]


class _Phase:
    """Times one phase of ntt_fr_distributed with a pair of events on the current stream (torch's RCCL collectives join
    the current stream before returning, so an event recorded after one marks its completion)."""

    def __init__(self, timings, name, t):
        self.rec = None
        if timings is not None and t.is_cuda:
            import torch
            self.rec = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            timings.setdefault(name, []).append(self.rec)

    def __enter__(self):
        if self.rec:
            self.rec[0].record()
        return self

    def __exit__(self, *exc):
        if self.rec:
            self.rec[1].record()
        return False


def resolve_timings(timings):
    """{phase: [(start, end) events]} as filled by ntt_fr_distributed(timings=...) -> {phase: milliseconds}; synchronises."""
    out = {}
    for name, recs in timings.items():
        ms = 0.0
        for a, b in recs:
            b.synchronize()
            ms += a.elapsed_time(b)
        out[name] = ms
    return out


def ntt_fr_distributed(local, log_n, inverse=False, group=None, ops=None, rank=None, world=None, exchange=None,
                       natural_output=False, timings=None):
    """local: torch tensor [N/G, 4] (this rank's contiguous slab).  Returns a tensor of the same shape (layout above).
    `exchange(list_of_blocks) -> list_of_blocks` overrides the collective (used by the single-process loopback tests).
    `timings`: optional dict that receives CUDA event pairs per phase (see resolve_timings)."""
    import torch
    import torch.distributed as dist
    if world is None:
        world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        rank = dist.get_rank(group) if world > 1 else 0
    if exchange is None:
        exchange = lambda blocks: _all_to_all(blocks, group)
    l1 = (log_n + 1) // 2
    l2 = log_n - l1
    n1, n2 = 1 << l1, 1 << l2
    assert n1 % world == 0 and n2 % world == 0, "world size must divide both matrix dimensions"
    r1, r2 = n1 // world, n2 // world
    x = local.reshape(r1, n2, 4)
    # 1. transpose: block for rank h = my rows, h's columns
    with _Phase(timings, "1_all_to_all_columns", x):
        recv = exchange([x[:, h * r2:(h + 1) * r2, :] for h in range(world)])   # each [r1, r2, 4]
    with _Phase(timings, "1b_local_transpose", x):
        cols = torch.cat(recv, dim=0)                                            # [n1, r2, 4]  (all n1, my n2)
        cols = cols.permute(1, 0, 2).contiguous()                                # [r2 (n2), n1, 4]
    # 2. column transforms, 3. twiddle
    with _Phase(timings, "2_column_ntt", x):
        cols = ops.ntt_batch(cols, l1, r2, inverse)
    with _Phase(timings, "3_twiddle", x):
        cols = ops.twiddle(cols, r2, n1, rank * r2, log_n, inverse)
    # 4. transpose back: block for rank h = my n2 rows, h's k1 slab
    with _Phase(timings, "4_all_to_all_rows", x):
        recv = exchange([cols[:, h * r1:(h + 1) * r1, :] for h in range(world)])  # each [r2, r1, 4]
    with _Phase(timings, "4b_local_transpose", x):
        rows = torch.cat(recv, dim=0)                                            # [n2, r1 (my k1), 4]
        rows = rows.permute(1, 0, 2).contiguous()                                # [r1 (k1), n2, 4]
    # 5. row transforms
    with _Phase(timings, "5_row_ntt", x):
        rows = ops.ntt_batch(rows, l2, r1, inverse)                              # [k1][k2] = X[k1 + N1 k2]
    if not natural_output:
        return rows.reshape(-1, 4)
    # optional: natural order slabs.  X index k = k1 + N1 k2; rank h owns k in [h N/G, (h+1) N/G) <=> k2 in h's r2 range
    with _Phase(timings, "6_all_to_all_natural", x):
        recv = exchange([rows[:, h * r2:(h + 1) * r2, :] for h in range(world)])  # each [r1 (k1 of sender), r2 (my k2), 4]
        full = torch.cat(recv, dim=0)                                            # [n1 (k1), r2 (k2), 4]
        out = full.permute(1, 0, 2).contiguous().reshape(-1, 4)                  # [k2][k1] -> k = k1 + N1 k2 ascending
    return out


class LoopbackExchange:
    """`world` logical ranks as threads of ONE process exchanging blocks through shared memory: lets a 1-GPU box (or a
    CPU test) exercise the multi-GPU data flow of ntt_fr_distributed without any process group."""

    def __init__(self, world):
        import threading
        self.world = world
        self.slots = [None] * world
        self.barrier = threading.Barrier(world)

    def exchange_fn(self, rank):
        def exchange(blocks):
            self.slots[rank] = blocks
            self.barrier.wait()
            got = [self.slots[src][rank] for src in range(self.world)]
            self.barrier.wait()
            return got
        return exchange

    def run(self, fn):
        """fn(rank, exchange) -> result, one thread per logical rank; returns the results in rank order."""
        import threading
        out, err = [None] * self.world, [None] * self.world

        def work(r):
            try:
                out[r] = fn(r, self.exchange_fn(r))
            except BaseException as e:  # noqa: BLE001 - re-raised below
                err[r] = e
                self.barrier.abort()
        threads = [threading.Thread(target=work, args=(r,)) for r in range(self.world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for e in err:
            if e is not None:
                raise e
        return out
