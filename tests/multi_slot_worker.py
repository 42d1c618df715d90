"""Worker of tests/test_gpu_multi_slot.py (own process: it initialises the library with several device slots).
python tests/multi_slot_worker.py NSLOTS [big]  -- slots share GPU 0 when the box has fewer GPUs.  Prints OK."""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "zkp-implementation_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
import zkp_hip as zkp  # noqa: E402
from zkp_hip import trapdoor  # noqa: E402

nslots = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ngpu = torch.cuda.device_count()
devs = [i % ngpu for i in range(nslots)]
zkp.init_devices(devs)
assert zkp.device_count() == nslots
n = (1 << 18) + 77
dev0 = torch.device("cuda", 0)
ks = bench.rand_fr_tensor(torch, n, 0x51A7, dev0)
sc = bench.rand_fr_tensor(torch, n, 0x51A8, dev0)
pts = torch.zeros(n * 12, dtype=torch.int64, device=dev0)
zkp.g1_fixed_base_mul_dev(ks, n, pts)                      # handle-less entry: slot 0
torch.cuda.synchronize()
exp = trapdoor.expected_msm(zkp, trapdoor.fr_inner_product(sc, ks))
h_pts = pts.cpu().numpy().view(np.uint64).reshape(n, 12)
h_sc = sc.cpu().numpy().view(np.uint64).reshape(n, 4)
sharded = zkp.G1Bases.from_host(h_pts)                     # default thread slot: sharded over all slots
chunks = sharded.shards()
assert len(chunks) == nslots and [c[0] for c in chunks] == list(range(nslots)) and sum(c[3] for c in chunks) == n
assert all(chunks[i][2] + chunks[i][3] == chunks[i + 1][2] for i in range(nslots - 1))
for rnd in range(2):
    got = zkp.msm_g1(sharded, h_sc)                        # host scalars, one Pippenger per slot
    assert got[1] == exp[1] and np.array_equal(got[0], exp[0]), ("host scalars", rnd)
    # made on side streams and NOT synchronised: the entry itself waits for each chunk's device (on a multi-GPU box the
    # copies to devices 1..k would otherwise still be in flight when the digits kernel reads them)
    resident = []
    for (_, d, off, ln) in chunks:
        with torch.cuda.stream(torch.cuda.Stream(device=torch.device("cuda", d))):
            resident.append((sc[off:off + ln].to(torch.device("cuda", d)) ^ 0).contiguous())
    got = zkp.msm_g1_sharded_dev(sharded, resident, n)     # scalars resident per chunk
    assert got[1] == exp[1] and np.array_equal(got[0], exp[0]), ("resident scalars", rnd)
    # the non-blocking contract (zkp_msm_g1_sharded_dev_after): producers on side streams behind a long spin, one recorded event
    # per chunk, no host synchronisation anywhere -- each chunk's launch waits for its event on the device
    resident2, events = [], []
    for (_, d, off, ln) in chunks:
        st = torch.cuda.Stream(device=torch.device("cuda", d))
        with torch.cuda.device(d), torch.cuda.stream(st):
            torch.cuda._sleep(100_000_000)
            resident2.append((sc[off:off + ln].to(torch.device("cuda", d)) ^ 0).contiguous())
            ev = torch.cuda.Event()
            ev.record(st)
            events.append(ev)
    got = zkp.msm_g1_sharded_dev(sharded, resident2, n, events=events)
    assert got[1] == exp[1] and np.array_equal(got[0], exp[0]), ("resident scalars behind events", rnd)
    m = chunks[1][2] + 5                                   # a prefix that ends inside chunk 1: later chunks contribute nothing
    got = zkp.msm_g1_sharded_dev(sharded, resident[:2] + [None] * (nslots - 2), m)
    expm = trapdoor.expected_msm(zkp, trapdoor.fr_inner_product(sc[:m], ks[:m]))
    assert got[1] == expm[1] and np.array_equal(got[0], expm[0]), ("prefix", rnd)
    if rnd == 0:
        sharded.precompute(0)                              # second round: every chunk expanded on its own slot
        assert sharded.info()[1] > 0
# device-pointer entries refuse a sharded handle, loudly
try:
    zkp.msm_g1_dev(sharded, sc, n)
    raise SystemExit("msm_g1_dev accepted a sharded handle")
except zkp.ZkpError as e:
    assert e.code == zkp.ZKP_E_ARG
# one independent worker per slot (the PLONK / replica pattern): threads choose their slot, run concurrently, results exact
ref = sc.clone().reshape(-1)[: 4 << 18].clone()
zkp.ntt_fr_dev(ref, 18)
errs = []


def replica(slot):
    try:
        zkp.set_device(slot)
        d = torch.device("cuda", devs[slot])
        with torch.cuda.device(d):
            x = sc.reshape(-1)[: 4 << 18].to(d).clone()
            for _ in range(3):
                y = x.clone()
                zkp.ntt_fr_dev(y, 18)
                torch.cuda.synchronize()
                assert torch.equal(y.cpu(), ref.cpu())
            b = zkp.G1Bases.from_host(h_pts[:5000])        # this thread's slot only
            assert len(b.shards()) == 1 and b.shards()[0][0] == slot
            g = zkp.msm_g1(b, h_sc[:5000])
            e = trapdoor.expected_msm(zkp, trapdoor.fr_inner_product(sc[:5000], ks[:5000]))
            assert np.array_equal(g[0], e[0])
    except BaseException as ex:  # noqa: BLE001
        errs.append((slot, repr(ex)))


ts = [threading.Thread(target=replica, args=(s,)) for s in range(nslots)]
for t in ts:
    t.start()
for t in ts:
    t.join()
assert not errs, errs
sharded.close()
if len(sys.argv) > 2 and sys.argv[2] == "big":
    # chunks of 2^19 scalars and more over expanded bases: every slot's host-scalar MSM walks its chunk in ranges, the later ones
    # uploaded by that slot's uploader thread -- all slots at once (one caller thread per slot inside zkp_msm_g1)
    nb = nslots * (1 << 19) + 4097
    ksb = bench.rand_fr_tensor(torch, nb, 0x51B7, dev0)
    scb = bench.rand_fr_tensor(torch, nb, 0x51B8, dev0)
    ptsb = torch.zeros(nb * 12, dtype=torch.int64, device=dev0)
    zkp.set_device(0)
    zkp.g1_fixed_base_mul_dev(ksb, nb, ptsb)
    torch.cuda.synchronize()
    zkp.set_device(-1)
    expb = trapdoor.expected_msm(zkp, trapdoor.fr_inner_product(scb, ksb))
    big = zkp.G1Bases.from_host(ptsb.cpu().numpy().view(np.uint64).reshape(nb, 12))
    assert len(big.shards()) == nslots and min(c[3] for c in big.shards()) >= 1 << 19
    big.precompute(0)
    h_scb = scb.cpu().numpy().view(np.uint64).reshape(nb, 4)
    for rnd in range(4):
        got = zkp.msm_g1(big, h_scb)
        assert got[1] == expb[1] and np.array_equal(got[0], expb[0]), ("big chunks, host scalars", rnd)
    os.environ["ZKP_MSM_FEED_FIRST_PCT"], os.environ["ZKP_MSM_FEED_SECOND_PCT"] = "10", "30"   # three ranges per chunk
    got = zkp.msm_g1(big, h_scb)
    assert got[1] == expb[1] and np.array_equal(got[0], expb[0]), "big chunks, three ranges"
    big.close()
zkp.shutdown()
print(f"OK multi-slot: {nslots} slots on {ngpu} GPU(s)")
