#!/usr/bin/env python3
"""Reporting grid of SURVEY.md §8d on ONE GPU: MSM and Fr NTT at n = 2^20, 2^22, 2^24, 2^26 (inputs resident in HBM).
Prints a markdown table; the multi-GPU columns are produced by the driver's bench.py --gpus N runs."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "zkp-implementation_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
import zkp_hip as zkp  # noqa: E402

zkp.init()
dev = torch.device("cuda", 0)
print("| workload | n | ms | throughput | algorithmic GB/s | % of 8 TB/s |")
print("|---|---|---|---|---|---|")
for ln in (20, 22, 24, 26):
    n = 1 << ln
    ks = bench.rand_fr_tensor(torch, n, 1000 + ln, dev)
    sc = bench.rand_fr_tensor(torch, n, 2000 + ln, dev)
    pts = torch.zeros(n * 12, dtype=torch.int64, device=dev)
    zkp.g1_fixed_base_mul_dev(ks, n, pts)
    torch.cuda.synchronize()
    bases = zkp.G1Bases.from_device(pts, n)
    del pts, ks
    for mode in ("plain c=16", "expanded c=20"):
        if mode.startswith("expanded"):
            t0 = time.perf_counter()
            bases.precompute(20)
            torch.cuda.synchronize()
            print(f"| (one-off SRS expansion) | 2^{ln} | {(time.perf_counter() - t0) * 1e3:.1f} | | | |", flush=True)
        zkp.msm_g1_dev(bases, sc, n)
        reps = 5 if ln <= 22 else 2
        t0 = time.perf_counter()
        for _ in range(reps):
            zkp.msm_g1_dev(bases, sc, n)
        dt = (time.perf_counter() - t0) / reps
        print(f"| G1 MSM, {mode} | 2^{ln} | {dt * 1e3:.2f} | {n / dt:.3e} scalar-muls/s | {128 * n / dt / 1e9:.1f} | "
              f"{128 * n / dt / 8e12 * 100:.2f} |", flush=True)
    bases.close()
    del sc
    torch.cuda.empty_cache()
for ln in (20, 22, 24, 26):
    n = 1 << ln
    data = bench.rand_fr_tensor(torch, n, 3000 + ln, dev).reshape(-1)
    ref = data.clone()
    zkp.ntt_fr_dev(data, ln)
    zkp.ntt_fr_dev(data, ln, inverse=True)
    torch.cuda.synchronize()
    assert torch.equal(data, ref)
    reps = 10 if ln <= 22 else 3
    t0 = time.perf_counter()
    for _ in range(reps):
        zkp.ntt_fr_dev(data, ln)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"| Fr NTT (one transform) | 2^{ln} | {dt * 1e3:.3f} | {n / dt:.3e} elem/s | {64 * n / dt / 1e9:.1f} | {64 * n / dt / 8e12 * 100:.2f} |", flush=True)
    del data, ref
    torch.cuda.empty_cache()
for ln in (20, 24, 26):
    n = 1 << ln
    g = torch.Generator(device=dev)
    g.manual_seed(ln)
    t = torch.randint(0, 2 ** 62, (n,), dtype=torch.int64, device=dev, generator=g)
    zkp.ntt_goldilocks_dev(t, ln)
    torch.cuda.synchronize()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        zkp.ntt_goldilocks_dev(t, ln)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"| Goldilocks NTT | 2^{ln} | {dt * 1e3:.3f} | {n / dt:.3e} elem/s | {16 * n / dt / 1e9:.1f} | {16 * n / dt / 8e12 * 100:.2f} |", flush=True)
