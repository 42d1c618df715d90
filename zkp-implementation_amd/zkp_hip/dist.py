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
# Rank g owns the contiguous slab of rows n1 in [g N1/G, (g+1) N1/G) (i.e. its N/G consecutive elements); r1 = N1/G, r2 = N2/G.
#
# Forward (natural slabs in, "k1-slab" layout out: local[k1 - g r1][k2] = X[k1 + N1 k2]):
#   0. pack         S[h][j][c] = x[j][h r2 + c]                     the one copy: an all-to-all needs contiguous per-peer blocks
#   1. all-to-all   R[g][j][c] -- which IS the row-major matrix [N1][r2] (all n1, my n2): no transpose
#   2. column transforms along axis 0 of that matrix, natural order, the twiddle omega_N^(n2 k1) fused into the store
#      (zkp_ntt_fr_axis0_dev)                                                                      -> Y[k1][c]
#   3. all-to-all   rank h gets the rows of its k1 slab, which are contiguous in Y: no pack          -> R2[g][j'][c]
#   4. row transforms READ that gathered layout (zkp_ntt_fr_layout_dev, n2 = g r2 + c) and write contiguous rows
# i.e. four passes over the data (two per transform) + one pack copy, against nine in round 1 (three permute copies, two
# stack copies, a separate twiddle pass).  The columns are cut into `chunks` groups that go through steps 0-3 as a pipeline:
# the all-to-all of chunk q+1 (RCCL's stream, all 7 xGMI links of the GPU busy: the exchange is point-to-point, which is why it is
# an all-to-all of large contiguous blocks and not a ring) overlaps the column transforms of chunk q; the row transforms then
# read all chunks through the layout's `mid` field.
#
# Inverse from the k1-slab layout back to natural slabs = the mirror image: row transforms write the twiddled, scattered
# send blocks directly, all-to-all, axis-0 column transforms, all-to-all, one unpack copy.  (1/N2 and 1/N1 multiply to 1/N.)
# `natural_output=True` (forward) / natural-order input (inverse) cost a third all-to-all + a transposing copy and exist
# for callers that need ascending k; a prover that multiplies pointwise between the two transforms does not.
# xGMI budget at 2^26 on 8 GPUs: every rank sends 7 x 32 MiB per all-to-all.

class TorchOps:
    """Local kernels of the distributed transform on this rank's GPU (torch int64 tensors shaped [..., 4])."""

    def __init__(self, zkp):
        self.zkp = zkp

    def ntt_batch(self, t, log_len, batch, inverse):
        flat = t.reshape(-1)
        self.zkp.ntt_fr_dev(flat, log_len, batch=batch, inverse=inverse)
        return t

    def axis0(self, src, dst, log_len, cols, inverse, tw_log_n, col0):
        self.zkp.ntt_fr_axis0_dev(src.reshape(-1), dst.reshape(-1), log_len, cols, inverse=inverse, tw_log_n=tw_log_n, tw_col0=col0)

    def layout(self, src, dst, log_n, batch, inverse, in_layout=None, out_layout=None, tw_log_n=0, tw_row0=0):
        mk = lambda l: None if l is None else self.zkp.NttLayout(*l)
        self.zkp.ntt_fr_layout_dev(src.reshape(-1), dst.reshape(-1), log_n, batch, inverse=inverse, in_layout=mk(in_layout),
                                   out_layout=mk(out_layout), tw_log_n=tw_log_n, tw_row0=tw_row0)


class _Exchanger:
    """all_to_all_single on [G, ...] buffers; `override(list_of_blocks) -> list_of_blocks` replaces the collective in the
    single-process loopback tests.  start() returns a handle to wait() on (RCCL: asynchronous on its own stream)."""

    def __init__(self, group, world, override, force_collective=False):
        self.group, self.world, self.override = group, world, override
        self.force = force_collective  # tests: go through the process group even with one rank (RCCL's asynchronous path)

    def start(self, recv, send):
        import torch
        import torch.distributed as dist
        if self.world == 1 and not self.force:
            recv.copy_(send)
            return None
        if self.override is not None:
            got = self.override([send[h] for h in range(self.world)])
            recv.copy_(torch.stack(list(got)))
            return None
        if dist.get_backend(self.group) != "gloo":
            return dist.all_to_all_single(recv, send, group=self.group, async_op=True)  # a failure here must surface on this rank
        try:
            dist.all_to_all_single(recv, send, group=self.group)
        except (RuntimeError, NotImplementedError):  # gloo builds without all_to_all (CPU tests only): gather, keep my column
            gathered = [torch.empty_like(send) for _ in range(self.world)]
            dist.all_gather(gathered, send.contiguous(), group=self.group)
            me = dist.get_rank(self.group)
            recv.copy_(torch.stack([gathered[r][me] for r in range(self.world)]))
        return None

    @staticmethod
    def wait(handle):
        if handle is not None:
            handle.wait()  # the current stream waits for the collective; the host does not


def _all_to_all(blocks, group):
    """blocks: list (len world) of equal-shape tensors to send; returns the list received (rank order)."""
    import torch
    world = len(blocks)
    if world == 1:
        return blocks
    send = torch.stack([b.contiguous() for b in blocks])
    recv = torch.empty_like(send)
    _Exchanger.wait(_Exchanger(group, world, None).start(recv, send))
    return [recv[r] for r in range(world)]


class _Phase:
    """Times one phase of ntt_fr_distributed with a pair of events on the current stream (a waited-for collective has joined
    the current stream, so an event recorded after the wait marks its completion)."""

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


def four_step_split(log_n, world=1):
    """log2 of N1, the length of the column transforms (the axis that is spread over the ranks), for N = N1 * N2 = 2^log_n.

    The column transform runs along axis 0 of a row-major matrix and is ONE kernel pass up to 2^8 rows (two up to 2^16); the row
    transforms of length N2 are ordinary batched transforms (two passes up to 2^18 with the wide radix).  A balanced split of 2^26
    (2^13 x 2^13) therefore costs 2 + 2 passes per rank where the single-GPU transform makes 3 for the same 26 stages; 2^8 x 2^18
    costs 1 + 2.  Both directions, and anything that reads the k1-slab layout, must use the same split: it is a function of
    (log_n, world) only."""
    lw = _log2(world)
    l1 = min(8, (log_n + 1) // 2)
    l1 = max(l1, lw)                       # world divides N1
    assert log_n - l1 >= lw + 2, "transform too small for this many ranks (at least four columns per rank)"
    return l1


def _log2(v):
    assert v > 0 and v & (v - 1) == 0, "power of two expected"
    return v.bit_length() - 1


def ntt_fr_distributed(local, log_n, inverse=False, group=None, ops=None, rank=None, world=None, exchange=None,
                       natural_output=False, timings=None, input_layout="natural", chunks=None, force_collective=False,
                       output_layout=None):
    """local: torch tensor [N/G, 4], this rank's slab.  Returns a tensor of the same shape.

    forward / inverse with input_layout="natural": natural slabs in -> k1-slab layout out (natural slabs with natural_output=True).
    inverse with input_layout="k1slab": the k1-slab layout a forward call produced -> natural slabs (the exact mirror).
    ONE all-to-all instead of two: forward with input_layout="columns" (columns layout in -> k1-slab layout out) and inverse with
    input_layout="k1slab", output_layout="columns" (k1-slab in -> columns layout out); `chunks` is then part of the layout
    (columns_shard / columns_gather / columns_chunks).
    `exchange(list_of_blocks) -> list_of_blocks` overrides the collective (single-process loopback tests).
    `timings`: optional dict that receives CUDA event pairs per phase (see resolve_timings).
    `chunks`: column groups pipelined through pack / all-to-all / column transforms (default: 4 when the group is RCCL)."""
    import torch
    import torch.distributed as dist
    if world is None:
        world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        rank = dist.get_rank(group) if world > 1 else 0
    l1 = four_step_split(log_n, world)
    l2 = log_n - l1
    n1, n2 = 1 << l1, 1 << l2
    assert n1 % world == 0 and n2 % world == 0, "world size must divide both matrix dimensions"
    r1, r2 = n1 // world, n2 // world
    assert r2 >= 4, "at least four columns per rank (16-byte... 128-byte runs of the tile kernels)"
    if input_layout == "columns" or output_layout == "columns":
        # `chunks` is part of this layout: a count the loop below would have to adjust describes another arrangement of the same
        # memory, and the transform would silently return wrong values
        assert chunks is not None, "the columns layout is chunk-major: pass the `chunks` it was built with (columns_chunks)"
        assert chunks == columns_chunks(log_n, world, chunks), (
            f"chunks={chunks} is not a valid chunk count of the columns layout at log_n={log_n}, world={world} "
            f"(power of two, at least four columns per chunk): use columns_chunks() -> {columns_chunks(log_n, world, chunks)}")
    if chunks is None:
        nccl = exchange is None and (world > 1 or force_collective) and dist.get_backend(group) != "gloo"
        chunks = 4 if nccl else 1
    while chunks > 1 and (r2 // chunks < 4 or r2 % chunks):
        chunks //= 2
    C, cw = chunks, r2 // chunks
    ex = _Exchanger(group, world, exchange, force_collective)
    G = world
    assert input_layout in ("natural", "k1slab", "columns") and output_layout in (None, "columns")
    if output_layout == "columns":
        assert inverse and input_layout == "k1slab", "the columns layout leaves the mirrored inverse (k1-slab in) only"
    if input_layout == "k1slab":
        assert inverse and not natural_output, "the k1-slab layout is what a forward transform leaves: only the inverse reads it"
        return _inverse_from_k1slab(local, log_n, ops, ex, rank, G, r1, r2, l1, l2, C, cw, timings, output_layout == "columns")
    h1, h2 = [None] * C, [None] * C
    if input_layout == "columns":
        assert not inverse, "the columns layout enters the forward transform only (its mirror is output_layout='columns')"
        recv = local.reshape(C, G, r1, cw, 4)        # already what the first all-to-all would have delivered: [C][N1][cw]
        y = torch.empty_like(recv)
        recv2 = torch.empty_like(recv)
    else:
        x = local.reshape(r1, G, C, cw, 4)
        send = torch.empty((C, G, r1, cw, 4), dtype=local.dtype, device=local.device)
        recv = torch.empty_like(send)
        y = torch.empty_like(send)                       # per chunk [N1][cw]
        recv2 = torch.empty_like(send)                   # [C][G][r1][cw]
        with _Phase(timings, "0_pack+1_all_to_all_columns(issue)", local):
            for q in range(C):
                send[q].copy_(x[:, :, q].permute(1, 0, 2, 3))            # S[h][j][c]
                h1[q] = ex.start(recv[q], send[q])
    with _Phase(timings, "2_column_ntt+twiddle(+waits)", local):
        for q in range(C):
            ex.wait(h1[q])
            ops.axis0(recv[q], y[q], l1, cw, inverse, log_n, rank * r2 + q * cw)   # [N1][cw], rows k1, twiddled
            h2[q] = ex.start(recv2[q], y[q])                         # k1 slabs are contiguous rows of y[q]
    with _Phase(timings, "3_all_to_all_rows(wait)", local):
        for q in range(C):
            ex.wait(h2[q])
    rows = torch.empty((r1, n2, 4), dtype=local.dtype, device=local.device)
    with _Phase(timings, "4_row_ntt", local):
        # logical n2 = (g, q, c_lo) at recv2[q][g][j'][c_lo]
        ops.layout(recv2, rows, l2, r1, inverse, in_layout=(_log2(cw), _log2(C), G * r1 * cw, r1 * cw, cw))
    if not natural_output:
        return rows.reshape(-1, 4)
    # optional: natural order slabs.  X index k = k1 + N1 k2; rank h owns k in [h N/G, (h+1) N/G) <=> k2 in h's r2 range
    with _Phase(timings, "5_all_to_all_natural", local):
        got = _exchange_blocks(ex, [rows[:, h * r2:(h + 1) * r2, :] for h in range(G)])   # each [r1 (k1 of sender), r2 (my k2), 4]
        full = torch.cat(got, dim=0)                                                      # [n1 (k1), r2 (k2), 4]
        out = full.permute(1, 0, 2).contiguous().reshape(-1, 4)                           # [k2][k1] -> k ascending
    return out


def _exchange_blocks(ex, blocks):
    import torch
    if ex.world == 1:
        return blocks
    send = torch.stack([b.contiguous() for b in blocks])
    recv = torch.empty_like(send)
    ex.wait(ex.start(recv, send))
    return [recv[r] for r in range(ex.world)]


def _inverse_from_k1slab(local, log_n, ops, ex, rank, G, r1, r2, l1, l2, C, cw, timings, to_columns=False):
    """Mirror of the forward flow: rows [k1 local][k2] -> natural slab [n1 local][n2] (or, to_columns, the columns layout: the
    column transforms' own output, without the second all-to-all and the unpack copy)."""
    import torch
    n2 = G * r2
    z = local.reshape(r1, n2, 4)
    send = torch.empty((C, G, r1, cw, 4), dtype=local.dtype, device=local.device)
    recv = torch.empty_like(send)
    xcol = torch.empty_like(send)
    recv2 = torch.empty_like(send)
    h1, h2 = [None] * C, [None] * C
    with _Phase(timings, "4'_row_intt+twiddle", local):
        # output n2 = (h, q, c_lo) of row j goes to send[q][h][j][c_lo], multiplied by omega_N^-((rank r1 + j) n2)
        ops.layout(z, send, l2, r1, True, out_layout=(_log2(cw), _log2(C), G * r1 * cw, r1 * cw, cw), tw_log_n=log_n,
                   tw_row0=rank * r1)
    with _Phase(timings, "3'_all_to_all_rows(issue)", local):
        for q in range(C):
            h1[q] = ex.start(recv[q], send[q])                       # [g][j][c] = rows k1 = g r1 + j: the matrix [N1][cw]
    with _Phase(timings, "2'_column_intt(+waits)", local):
        for q in range(C):
            ex.wait(h1[q])
            ops.axis0(recv[q], xcol[q], l1, cw, True, 0, 0)          # [N1 (n1)][cw]
            if not to_columns:
                h2[q] = ex.start(recv2[q], xcol[q])                  # n1 slabs are contiguous rows
    if to_columns:
        return xcol.reshape(-1, 4)                                   # [C][N1][cw]: x[n1 N2 + rank r2 + q cw + c]
    with _Phase(timings, "1'_all_to_all_columns(wait)", local):
        for q in range(C):
            ex.wait(h2[q])
    with _Phase(timings, "0'_unpack", local):
        out = recv2.permute(2, 1, 0, 3, 4).contiguous()              # [j''][g][q][c_lo] = x[n1 local][n2]
    return out.reshape(-1, 4)


# ----------------------------------------------------------------------------- one exchange per transform: the "columns" layout
# The flow above pays TWO all-to-alls per transform because it starts from contiguous slabs: the butterflies that join the most
# distant elements come first, and those live on different ranks.  A distributed prover is free to keep its coefficient vectors
# block-cyclically instead: view x as the row-major N1 x N2 matrix of the four-step split and give rank g the COLUMNS
# [g r2, (g + 1) r2) of every row (runs of r2 = N2 / G consecutive elements dealt out round-robin; 1 MiB runs at 2^26 on 8 GPUs).
# That is exactly what the first all-to-all would have delivered, so the transform starts at step 2 and one exchange is left:
#   forward   columns layout -> [2. column transforms + twiddle] -> [3. all-to-all] -> [4. row transforms] -> k1-slab layout
#   inverse   k1-slab layout -> [4'. row transforms + twiddle] -> [3'. all-to-all] -> [2'. column transforms] -> columns layout
# (input_layout="columns" / output_layout="columns" of ntt_fr_distributed).  Per rank and transform at 2^26 on 8 GPUs: one exchange
# of 7 x 32 MiB instead of two, three kernel passes, no pack copy.  Pointwise products between a forward and an inverse transform
# do not care about the layout, and commitments do not either (the MSM is a sum: shard the SRS the same way).
# The local tensor is chunk-major, [C][N1][cw] with cw = r2 / C: element (q, n1, c) = x[n1 N2 + g r2 + q cw + c]; chunk q's
# all-to-all runs under chunk q + 1's column transforms.  C = 1 is the plain [N1][r2] matrix.

def columns_shard(full, log_n, rank, world, chunks=1):
    """Rank `rank`'s share of a full-length vector [N, 4] in the columns layout with `chunks` column groups: [C * N1 * cw, 4]."""
    l1 = four_step_split(log_n, world)
    n1, n2 = 1 << l1, 1 << (log_n - l1)
    r2 = n2 // world
    assert chunks == columns_chunks(log_n, world, chunks), f"chunks={chunks}: not a chunk count of this layout (columns_chunks)"
    cw = r2 // chunks
    m = full.reshape(n1, n2, 4)[:, rank * r2:(rank + 1) * r2]                    # [N1][r2]
    return m.reshape(n1, chunks, cw, 4).permute(1, 0, 2, 3).contiguous().reshape(-1, 4)


def columns_gather(parts, log_n, chunks=1):
    """The full natural-order vector [N, 4] from every rank's columns-layout share (rank order)."""
    import torch
    world = len(parts)
    l1 = four_step_split(log_n, world)
    n1, n2 = 1 << l1, 1 << (log_n - l1)
    assert chunks == columns_chunks(log_n, world, chunks), f"chunks={chunks}: not a chunk count of this layout (columns_chunks)"
    cw = n2 // world // chunks
    ms = [p.reshape(chunks, n1, cw, 4).permute(1, 0, 2, 3).reshape(n1, chunks * cw, 4) for p in parts]
    return torch.cat(ms, dim=1).contiguous().reshape(-1, 4)


def columns_chunks(log_n, world, chunks):
    """The number of column groups ntt_fr_distributed will actually use for a requested `chunks` (a layout parameter of the
    columns layout: both directions and every reader must agree on it)."""
    l1 = four_step_split(log_n, world)
    r2 = (1 << (log_n - l1)) // world
    chunks = 1 << (max(int(chunks), 1).bit_length() - 1)   # a power of two (the library's zkp_ntt_fr_sharded_geometry accepts no other)
    while chunks > 1 and (r2 // chunks < 4 or r2 % chunks):
        chunks //= 2
    return max(chunks, 1)


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
