"""Worker of tests/test_gpu_sharded_ntt.py (own process: it initialises the library with several device slots).
python tests/sharded_ntt_worker.py NSLOTS LOG_N[,LOG_N...] [big]  -- slots share GPU 0 when the box has fewer GPUs.  Prints OK.

The in-process multi-GPU transform behind the C ABI (zkp_ntt_fr_sharded_dev / zkp_ntt_fr_sharded, include/zkp_hip.h) against the
single-device transform zkp_ntt_fr_dev -- itself checked against the oracle up to 2^25 by tests/test_gpu_parity.py, and against the
oracle directly here for the small sizes -- for every supported pair of layouts, both directions, several chunk counts, with
library-owned and caller-owned streams.  "big": one pass over the layouts with fewer variants (2^24 .. 2^26)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "zkp-implementation_amd")):
    sys.path.insert(0, p)
os.environ["ZKP_NTT_SHARD_MIN_LOG"] = "12"   # zkp_ntt_fr routes to the sharded transform from 2^12 on in this process
import torch  # noqa: E402

import bench  # noqa: E402
import zkp_hip as zkp  # noqa: E402

nslots = int(sys.argv[1])
logs = [int(v) for v in sys.argv[2].split(",")]
big = len(sys.argv) > 3 and sys.argv[3] == "big"
ngpu = torch.cuda.device_count()
devs = [i % ngpu for i in range(nslots)]
zkp.init_devices(devs)
assert zkp.device_count() == nslots
NAT, K1, COLS = zkp.NTT_NATURAL, zkp.NTT_K1SLAB, zkp.NTT_COLUMNS
dev0 = torch.device("cuda", 0)


def to_layout(full, layout, geo):
    """full: [N, 4] on device 0 (index = the vector's own index) -> list of per-slot slabs [N/G, 4] on the slots' devices."""
    n1, n2, G, C, cw, r2 = 1 << geo["log_n1"], 1 << geo["log_n2"], geo["slots"], geo["chunks"], geo["cw"], geo["r2"]
    if layout == NAT:
        parts = list(full.reshape(G, -1, 4))
    elif layout == K1:   # slab g [j][i2] = v[(g r1 + j) + N1 i2]
        parts = list(full.reshape(n2, n1, 4).permute(1, 0, 2).contiguous().reshape(G, -1, 4))
    else:                # slab g [q][n1][c] = v[n1 N2 + g r2 + q cw + c]
        m = full.reshape(n1, G, C, cw, 4).permute(1, 2, 0, 3, 4).contiguous()
        parts = list(m.reshape(G, -1, 4))
    return [p.contiguous().to(torch.device("cuda", devs[g])).clone() for g, p in enumerate(parts)]


def from_layout(slabs, layout, geo):
    n1, n2, G, C, cw = 1 << geo["log_n1"], 1 << geo["log_n2"], geo["slots"], geo["chunks"], geo["cw"]
    st = torch.stack([s.to(dev0) for s in slabs])
    if layout == NAT:
        return st.reshape(-1, 4)
    if layout == K1:
        return st.reshape(n1, n2, 4).permute(1, 0, 2).contiguous().reshape(-1, 4)
    return st.reshape(G, C, n1, cw, 4).permute(2, 0, 1, 3, 4).contiguous().reshape(-1, 4)


PAIRS = [(NAT, K1), (COLS, K1), (K1, NAT), (K1, COLS), (NAT, NAT)]
checked = 0
for log_n in logs:
    n = 1 << log_n
    full = bench.rand_fr_tensor(torch, n, 0x5A4D + log_n, dev0).reshape(n, 4)
    exp = {}
    for inv in (False, True):
        t = full.clone().reshape(-1)
        zkp.ntt_fr_dev(t, log_n, inverse=inv)        # handle-less entry: slot 0, one device
        exp[inv] = t.reshape(n, 4)
    torch.cuda.synchronize()
    if log_n <= 16:  # and the oracle itself for the small sizes
        from oracle import oracle as orc
        orc.build()
        h = full.cpu().numpy().view(np.uint64).reshape(n, 4)
        assert np.array_equal(exp[False].cpu().numpy().view(np.uint64).reshape(n, 4), orc.ntt_fr(h))
    auto = zkp.ntt_fr_sharded_geometry(log_n)
    chunk_list = [0] if big else sorted({0, 1, 2, min(8, auto["r2"] // 4)} - ({2} if auto["r2"] < 8 else set()))
    for chunks in chunk_list:
        geo = zkp.ntt_fr_sharded_geometry(log_n, 0, chunks)
        assert geo["slots"] == nslots and geo["slab"] * nslots == n and geo["cw"] * geo["chunks"] == geo["r2"]
        variants = ([(NAT, K1, False), (K1, NAT, True), (COLS, K1, False), (K1, COLS, True), (NAT, NAT, False)] if big else
                    [(a, b, inv) for (a, b) in PAIRS for inv in (False, True)])
        for (lin, lout, inv) in variants:
            slabs = to_layout(full, lin, geo)
            torch.cuda.synchronize()
            zkp.ntt_fr_sharded_dev(slabs, log_n, inverse=inv, layout_in=lin, layout_out=lout, chunks=chunks)
            got = from_layout(slabs, lout, geo)
            assert torch.equal(got, exp[inv]), ("sharded != single device", log_n, chunks, lin, lout, inv)
            checked += 1
            del slabs, got
    # caller-owned streams: enqueue only, producers and consumers on the same streams, no host synchronisation in between;
    # a forward transform into the K1SLAB layout, a pointwise operation there, the mirrored inverse back (what a prover does)
    geo = zkp.ntt_fr_sharded_geometry(log_n)
    streams = [torch.cuda.Stream(device=torch.device("cuda", d)) for d in devs]
    slabs = []
    for g in range(nslots):
        with torch.cuda.device(devs[g]), torch.cuda.stream(streams[g]):
            torch.cuda._sleep(20_000_000)             # the producer is still running when the library is called
            slabs.append((full.reshape(nslots, -1, 4)[g].to(torch.device("cuda", devs[g])) ^ 0).contiguous())
    zkp.ntt_fr_sharded_dev(slabs, log_n, layout_in=NAT, layout_out=K1, streams=streams)
    zkp.ntt_fr_sharded_dev(slabs, log_n, inverse=True, layout_in=K1, layout_out=NAT, streams=streams)
    outs = []
    for g in range(nslots):
        with torch.cuda.device(devs[g]), torch.cuda.stream(streams[g]):
            outs.append(slabs[g].clone())
    for s in streams:
        s.synchronize()
    assert torch.equal(from_layout(outs, NAT, geo), full), ("round trip on caller streams", log_n)
    checked += 1
    # the host-pointer form, natural order in and out, with and without a coset; zkp_ntt_fr takes the same route in this process
    if log_n <= 22:
        h = full.cpu().numpy().view(np.uint64).reshape(n, 4)
        coset = bench.rand_fr_tensor(torch, 1, 0xC05E, dev0).cpu().numpy().view(np.uint64).reshape(4)
        for inv in (False, True):
            for cs in (None, coset):
                t = full.clone().reshape(-1)
                zkp.ntt_fr_dev(t, log_n, inverse=inv, coset=cs)
                want = t.cpu().numpy().view(np.uint64).reshape(n, 4)
                assert np.array_equal(zkp.ntt_fr_sharded(h, inverse=inv, coset=cs), want), ("host form", log_n, inv, cs is not None)
                assert np.array_equal(zkp.ntt_fr(h, inverse=inv, coset=cs), want), ("zkp_ntt_fr routed", log_n, inv)
                checked += 2
    del full, exp, outs
    torch.cuda.empty_cache()

# refusals: loud, never adjusted
lg = logs[0]
geo = zkp.ntt_fr_sharded_geometry(lg)
x = [torch.zeros((geo["slab"], 4), dtype=torch.int64, device=torch.device("cuda", d)) for d in devs]
for bad in (3, geo["r2"] // 2 if geo["r2"] >= 4 else 64, 128):
    try:
        zkp.ntt_fr_sharded_dev(x, lg, chunks=bad)
        raise SystemExit(f"chunks={bad} was accepted")
    except zkp.ZkpError as e:
        assert e.code == zkp.ZKP_E_ARG, e
for (lin, lout) in ((COLS, NAT), (COLS, COLS), (K1, K1), (NAT, COLS), (7, 0)):
    try:
        zkp.ntt_fr_sharded_dev(x, lg, layout_in=lin, layout_out=lout)
        raise SystemExit(f"layout pair {lin}->{lout} was accepted")
    except zkp.ZkpError as e:
        assert e.code == zkp.ZKP_E_ARG, e
try:
    zkp.ntt_fr_sharded_geometry(3 + 2 * (nslots.bit_length() - 1) - 1 if nslots > 1 else 3)
    raise SystemExit("a transform with fewer than four columns per slot was accepted")
except zkp.ZkpError as e:
    assert e.code == zkp.ZKP_E_ARG and "too small" in str(e), e
zkp.shutdown()
print(f"OK sharded ntt: {nslots} slots on {ngpu} GPU(s), sizes {logs}, {checked} transforms bit-identical to the single-device one")
