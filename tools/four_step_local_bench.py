#!/usr/bin/env python3
"""Local (per-rank) cost of the multi-GPU four-step NTT on ONE GPU: the kernels one rank of `world` runs for a transform of
2^LOG_N elements, on buffers of the real per-rank shapes, without any exchange.  python tools/four_step_local_bench.py [26] [8]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "zkp-implementation_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
import zkp_hip as zkp  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 26
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
zkp.init()
from zkp_hip import dist as zd  # noqa: E402
l1 = int(os.environ["ZKP_FOUR_STEP_L1"]) if os.environ.get("ZKP_FOUR_STEP_L1") else zd.four_step_split(log_n, G)
l2 = log_n - l1
n1, n2 = 1 << l1, 1 << l2
r1, r2 = n1 // G, n2 // G
slab = r1 * n2
x = bench.rand_fr_tensor(torch, slab, 7, "cuda")
buf = torch.empty_like(x)
out = torch.empty_like(x)


def timed(name, fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"{name:58s} {ms:8.3f} ms   {2 * 32 * slab / ms / 1e6:8.1f} GB/s (read + write of the slab)", flush=True)
    return ms


print(f"four-step 2^{log_n} over {G} ranks: per rank {slab} elements = {32 * slab / 2 ** 20:.0f} MiB; N1 = 2^{l1}, N2 = 2^{l2}, r1 = {r1}, r2 = {r2}")
tot = 0.0
tot += timed("0 pack  x[j][h r2 + c] -> S[h][j][c]", lambda: buf.view(G, r1, r2, 4).copy_(x.view(r1, G, r2, 4).permute(1, 0, 2, 3)))
for C in (1, 4):
    cw = r2 // C
    t = 0.0
    t = timed(f"2 column transforms, axis 0 of [2^{l1}][{cw}] x {C} chunk(s), twiddle fused",
              lambda: [zkp.ntt_fr_axis0_dev(buf.view(C, -1)[q].reshape(-1), out.view(C, -1)[q].reshape(-1), l1, cw, tw_log_n=log_n,
                                            tw_col0=q * cw) for q in range(C)])
    lay = zkp.NttLayout(cw.bit_length() - 1, C.bit_length() - 1, G * r1 * cw, r1 * cw, cw)
    t += timed(f"4 row transforms, {r1} x 2^{l2}, gathered input ({C} chunk(s))",
               lambda: zkp.ntt_fr_layout_dev(out.reshape(-1), buf.reshape(-1), l2, r1, in_layout=lay))
    print(f"   -> local kernels with {C} chunk(s): {tot + t:.3f} ms per forward transform per rank; one-exchange form (columns layout in, no pack): {t:.3f} ms")
timed("(reference) plain batched transforms of the same shape", lambda: zkp.ntt_fr_dev(buf.reshape(-1), l2, batch=r1))
timed("(reference) round-1 style permute().contiguous() of the slab", lambda: out.view(r2 * G, r1, 4).copy_(x.view(r1, r2 * G, 4).permute(1, 0, 2)))
