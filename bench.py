#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X MSM / NTT backend.

Contract: `python bench.py --gpus N --steps K --warmup W`.  For N > 1 the driver launches it under
torch.distributed.run (one rank per GPU over RCCL); started WITHOUT a launcher, `--gpus N` (N > 1) re-launches itself as N
ranks through torch.distributed.run before anything touches a GPU and exits with that job's status -- it never reports a
one-rank run as an N-GPU one, and any WORLD_SIZE that disagrees with --gpus is an error.

One "step" = one pass of the hot path over one batch of synthetic input already resident in HBM: a Pippenger G1 MSM over
2^log_n (default 2^20 = BASELINE.json configs[1]) BLS12-381 points PER GPU.  With N > 1 the MSM of N * 2^log_n terms is
sharded by contiguous point/scalar chunk (weak scaling); each step ends with the real exchange step of the path: an RCCL
all-gather of the per-GPU partial sums (192 B each) followed by the EC-add combine (EC addition is not an RCCL reduction
operator, so the "all-reduce" is gather + local add).  `--total-log-n T` fixes the TOTAL instead (strong scaling, 2^T / N
terms per GPU): `--gpus 8 --total-log-n 26` is BASELINE.json configs[4] as written.

Rank 0 prints ONE JSON line.  `value` = scalar-muls/s over all GPUs.  `roofline` prices the dominant kernel
(msm_accumulate) in algorithmic bytes (128 B per scalar-mul) against the 8 TB/s HBM peak, with the kernel's
duration measured live by HIP events on the launch stream; `cpu_baseline` times the oracle's reference-faithful naive MSM
(kzg/src/scheme.rs:88-94 restated in C) on a bounded sample of the same inputs on this box's host cores, and checks the
GPU result bit-exactly on that sample.  The timed full-size result itself is checked against the known answer
(sum s_i k_i) G of the synthetic SRS P_i = k_i G (`bit_exact_full`; zkp_hip/trapdoor.py, no oracle involved).

`extra` (N = 1): the north-star sizes 2^22 / 2^24 / 2^26 (`msm_grid`), the PCIe-inclusive rate, the one-off SRS expansion,
the unexpanded-bases mode, the Fr NTT round trip (configs[2]), FRI, and the PLONK prover (configs[3]).
`extra` (N > 1): `sharded_grid` -- total sizes 2^22 / 2^24 / 2^26 sharded over the N GPUs (MSM by chunk, Fr NTT as the four-step
transform with its RCCL all-to-alls, per-phase milliseconds); `config4` = its 2^26 entry = BASELINE.json configs[4].
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "zkp-implementation_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
MSM_BYTES_PER_UNIT = 128  # 32 B scalar + 96 B affine point
NTT_BYTES_PER_ELEM = 64   # read 32 B + write 32 B per element per transform
MAD_PEAK = 3.33e13        # measured v_mad_u64_u32 lane-ops/s of one MI355X (profiles/r01_issue_rate.txt)

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def rand_fr_tensor(torch, n, seed, device):
    """n pseudo-random Fr residues (any 4 limbs < 2^254 < r is a valid Montgomery residue)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    t = torch.randint(-(2 ** 63), 2 ** 63 - 1, (n, 4), dtype=torch.int64, device=device, generator=g)
    t[:, 3] &= 0x3FFFFFFFFFFFFFFF
    return t.contiguous()


def fr_mont(vals):
    """python ints -> (n,4) uint64 Montgomery residues (x * 2^256 mod r), no oracle involved."""
    out = np.empty((len(vals), 4), dtype=np.uint64)
    m64 = (1 << 64) - 1
    for i, v in enumerate(vals):
        x = (v << 256) % R_MOD
        out[i, 0], out[i, 1], out[i, 2], out[i, 3] = x & m64, (x >> 64) & m64, (x >> 128) & m64, x >> 192
    return out


# ----------------------------------------------------------------------------------------------------------- launcher
def launcher_action(gpus, env):
    """What to do before any GPU call: "run" (this process is the right rank of the right world), "spawn" (no launcher was
    used for N > 1: start N ranks ourselves), or an error string.  Pure function of (--gpus, environment): unit-tested."""
    ws = env.get("WORLD_SIZE")
    if gpus < 1:
        return "error: --gpus must be >= 1"
    if ws is None:
        return "run" if gpus == 1 else "spawn"
    try:
        world = int(ws)
    except ValueError:
        return f"error: WORLD_SIZE={ws!r} is not a number"
    if world != gpus:
        return (f"error: --gpus {gpus} but WORLD_SIZE={world}: launch with `python -m torch.distributed.run --nnodes=1 "
                f"--nproc-per-node {gpus} bench.py --gpus {gpus} ...` (or run `python bench.py --gpus {gpus}` without a "
                "launcher and it starts its own ranks)")
    return "run"


def spawn_ranks(gpus, argv):
    """Re-launch this script as `gpus` ranks under torch.distributed.run (a child process: nothing here has touched a GPU)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd)


# ----------------------------------------------------------------------------------------------------------- workloads
class MsmWorkload:
    """2^log_n synthetic (scalar, point) pairs resident on `device`: P_i = k_i G (valid curve points with a known discrete log,
    so that the exact MSM is (sum s_i k_i) G), scalars uniform.  Seeds follow SURVEY 8d; `chunk` separates the ranks' chunks."""

    def __init__(self, zkp, torch, device, log_n, chunk=0, expand=0, keep_points=False):
        self.zkp, self.torch, self.device, self.log_n, self.n = zkp, torch, device, log_n, 1 << log_n
        n = self.n
        self.ks = rand_fr_tensor(torch, n, 0xBA5E0000 + log_n * 64 + chunk, device)
        self.scalars = rand_fr_tensor(torch, n, 0x5EED0000 + log_n * 64 + chunk, device)
        pts = torch.zeros(n * 12, dtype=torch.int64, device=device)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        zkp.g1_fixed_base_mul_dev(self.ks, n, pts)  # Srs::new_from_secret's kernel (kzg/src/srs.rs:48-63)
        torch.cuda.synchronize()
        self.gen_ms = (time.perf_counter() - t0) * 1e3
        self.bases = zkp.G1Bases.from_device(pts, n)
        self.pts = pts if keep_points else None
        self.expand_ms, self.expand_bytes, self.planes, self.window_bits = None, None, None, 0
        if expand == "auto":
            self.expand(0)   # the library's automatic width
        elif expand:
            self.expand(expand)

    def expand(self, window_bits):
        self.torch.cuda.synchronize()
        t0 = time.perf_counter()
        self.bases.precompute(window_bits)  # one-off, as KzgScheme::new would do for a fixed SRS
        self.torch.cuda.synchronize()
        self.expand_ms = (time.perf_counter() - t0) * 1e3
        self.window_bits, self.planes = self.bases.info()
        self.expand_bytes = self.planes * self.n * 128

    def limb_sums(self):
        from zkp_hip import trapdoor
        return trapdoor.limb_products(self.scalars, self.ks)

    def close(self):
        self.bases.close()
        self.ks = self.scalars = self.pts = None
        self.torch.cuda.empty_cache()


def check_against_trapdoor(zkp, limb_sums, result):
    """result == (sum s_i k_i) G ?  limb_sums: (16,16) python ints already summed over every chunk that went into `result`."""
    from zkp_hip import trapdoor
    e = trapdoor.fr_inner_product_from_limbs(limb_sums)
    exp, einf = trapdoor.expected_msm(zkp, e)
    got, ginf = result
    return bool(int(ginf) == int(einf) and np.array_equal(np.asarray(got, dtype=np.uint64), exp))


def allreduce_limb_sums(torch, dist, sums, device):
    t = torch.tensor(sums, dtype=torch.int64, device=device)  # entries < 2^58 per rank: world * 2^58 < 2^63 up to 32 ranks
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [[int(v) for v in row] for row in t.cpu().tolist()]


def time_msm(zkp, torch, step, steps, warmup, fence):
    """-> (seconds for `steps` steps, last result, per-phase ms per step)"""
    result = None
    for _ in range(warmup):
        result = step()
    # The timed region records the dominant kernel only (HIP events around msm_accumulate + its in-kernel clock stamps: what
    # `roofline` is made of): every recorded phase boundary is a marker on the stream, a bubble of ~5 us, and with all five phases
    # recorded they were ~1 % of the step.  The other phases are measured right after it, in a short pass of their own.
    zkp.profile_reset()
    zkp.profile_enable(2)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        result = step()
    fence()
    elapsed = time.perf_counter() - t0
    zkp.profile_enable(False)
    phases = {}
    ms, cnt = zkp.profile_read("msm_accumulate")
    phases["msm_accumulate"] = ms / steps if cnt else None  # a step may run a phase more than once (scalar ranges)
    clk_acc = None
    try:
        clk_acc = zkp.profile_clock_read("msm_accumulate")
    except Exception as e:  # noqa: BLE001
        clk_acc = e
    zkp.profile_reset()
    extra_steps = max(2, min(5, steps))
    zkp.profile_enable(True)
    for _ in range(extra_steps):
        step()
    fence()
    zkp.profile_enable(False)
    for name in ("msm_digits", "msm_sort", "msm_bucket_reduce", "msm_tail_host"):
        ms, cnt = zkp.profile_read(name)
        phases[name] = ms / extra_steps if cnt else None
    phases = {k: phases.get(k) for k in ("msm_digits", "msm_sort", "msm_accumulate", "msm_bucket_reduce", "msm_tail_host")}
    # the shader clock msm_accumulate held under its own load during the timed region (in-kernel s_memtime / s_memrealtime stamps,
    # include/zkp_hip.h: zkp_profile_clock_read), and the v_mad_u64_u32 issue peak of THIS box measured right after it, chip warm
    try:
        if isinstance(clk_acc, Exception):
            raise clk_acc
        cyc, ref, waves = clk_acc
        phases["clock"] = {"msm_accumulate_mhz": 100.0 * cyc / ref if ref else None, "stamped_workgroups": waves}
        rate, mhz, ms = zkp.probe_mad_rate(20)
        phases["clock"]["mad_probe"] = {"lane_mads_per_s": rate, "clock_mhz": mhz, "ms_per_launch": ms}
    except Exception as e:  # noqa: BLE001 -- the headline must not depend on the diagnostics
        phases["clock"] = {"error": repr(e)}
    zkp.profile_reset()
    return elapsed, result, phases


def bench_plonk(zkp, torch, device, log_n, expand=0):
    """Five prover rounds (plonk/src/prover.rs:61-293) on a synthetic mul / add / constant-gate chain with copy constraints and public inputs."""
    n = 1 << log_n
    rnd = np.random.default_rng(0xC16C)
    rb = [int(x) for x in rnd.integers(1, 2 ** 62, n)]
    # chain mul / add / mul / constant with non-zero public inputs on all three kinds (gate.rs:38-111: the stored pi is negated;
    # a constant gate has q_l = 1, q_o = 0, q_c = -constant and passes its input on): same family as tests/test_gpu_plonk.py
    a_v, c_v, a = [0] * n, [0] * n, 5
    q_m, q_l, q_r, q_o, q_c, pi_v = ([0] * n for _ in range(6))
    for i in range(n):
        kind, pi = i % 4, (7 * i + 1 if i % 8 in (0, 1, 3) else 0)
        a_v[i] = a
        if kind == 3:
            q_l[i], q_c[i], c_v[i] = 1, (pi - a) % R_MOD, a
        elif kind == 1:
            q_l[i], q_r[i], q_o[i], c_v[i] = 1, 1, R_MOD - 1, (a + rb[i] - pi) % R_MOD
        else:
            q_m[i], q_o[i], c_v[i] = 1, R_MOD - 1, (a * rb[i] - pi) % R_MOD
        pi_v[i] = (-pi) % R_MOD
        a = c_v[i]
    w = pow(pow(7, (R_MOD - 1) >> 32, R_MOD), 1 << (32 - log_n), R_MOD)
    roots = [1] * n
    for i in range(1, n):
        roots[i] = roots[i - 1] * w % R_MOD
    cols = {"f_a": a_v, "f_b": rb, "f_c": c_v, "q_m": q_m, "q_l": q_l, "q_r": q_r, "q_o": q_o, "q_c": q_c, "pi": pi_v,
            "s_sigma_1": [(roots[i - 1] * 3) % R_MOD if i else roots[0] for i in range(n)],
            "s_sigma_2": [roots[i] * 2 % R_MOD for i in range(n)],
            "s_sigma_3": [roots[i + 1] if i < n - 1 else roots[i] * 3 % R_MOD for i in range(n)]}
    # Circuit::compile's 12 interpolations (circuit.rs:173-176, 230-232) on the GPU
    stack = torch.from_numpy(np.concatenate([fr_mont(cols[k]) for k in zkp.CIRCUIT_POLYS]).view(np.int64)).to(device)
    warm = stack.clone()
    zkp.ntt_fr_dev(warm.reshape(-1), log_n, batch=12, inverse=True)  # first use of this size builds the plan's tables: not timed
    torch.cuda.synchronize()
    del warm
    t0 = time.perf_counter()
    zkp.ntt_fr_dev(stack.reshape(-1), log_n, batch=12, inverse=True)
    torch.cuda.synchronize()
    t_compile = time.perf_counter() - t0
    host = stack.cpu().numpy().view(np.uint64).reshape(12, n, 4)
    polys = {k: host[i] for i, k in enumerate(zkp.CIRCUIT_POLYS)}
    f = lambda v: fr_mont([v])[0]
    srs = zkp.Srs.new_from_secret(f(0x5EC12E7), n)
    if expand:
        srs.bases.precompute(0 if expand == "auto" else expand)  # one-off, as KzgScheme::new would do for a fixed SRS; 0 = the library's width
        expand = srs.bases.info()[0]
    vals = [int(x) for x in rnd.integers(1, 2 ** 62, 14)]
    times, rounds, phases = [], None, {}
    for rep in range(3):
        pr = zkp.PlonkProver(srs.bases, log_n, polys, f(2), f(3))
        torch.cuda.synchronize()
        if rep == 2:
            zkp.profile_reset()
            zkp.profile_enable(True)
        ts = [time.perf_counter()]
        pr.round1(fr_mont(vals[:6]))
        ts.append(time.perf_counter())
        pr.round2(f(vals[9]), f(vals[10]), fr_mont(vals[6:9]))
        ts.append(time.perf_counter())
        _, degree = pr.round3(f(vals[11]))
        ts.append(time.perf_counter())
        pr.round4(f(vals[12]))
        ts.append(time.perf_counter())
        pr.round5(f(vals[13]))
        torch.cuda.synchronize()
        ts.append(time.perf_counter())
        times.append(ts[-1] - ts[0])
        if times[-1] == min(times):
            rounds = [round((b - a) * 1e3, 3) for a, b in zip(ts, ts[1:])]
        if rep == 2:
            zkp.profile_enable(False)
            for name in ("msm_digits", "msm_sort", "msm_accumulate", "msm_bucket_reduce", "msm_tail_host", "ntt_fr_pass"):
                ms, cnt = zkp.profile_read(name)
                phases[name] = {"ms": round(ms, 3), "count": cnt}
            zkp.profile_reset()
        pr.close()
    # generate_proof in one call, transcript included (zkp_plonk_prove)
    pr = zkp.PlonkProver(srs.bases, log_n, polys, f(2), f(3))
    bl = fr_mont(vals[:9])
    pr.prove(bl)
    torch.cuda.synchronize()
    fulls = []  # best of five, as prove_ms above (a single call is at the mercy of one host hiccup: 6.1 ms was seen once for a 3.5 ms proof)
    for _ in range(5):
        t0 = time.perf_counter()
        full = pr.prove(bl)
        fulls.append(time.perf_counter() - t0)
    t_full = min(fulls)
    g2s, _ = zkp.g2_mul(zkp.g2_generator(), f(0x5EC12E7))  # [s]_2 of the same SRS
    t0 = time.perf_counter()
    verdict = pr.verify(g2s, full)  # plonk/src/verifier.rs with real pairings (host) + 8 circuit commitments (GPU)
    t_verify = time.perf_counter() - t0
    pr.close()
    return {"workload": f"PLONK prover rounds 1-5, 2^{log_n}-gate synthetic circuit, 1 GPU (BASELINE configs[3]); "
                        "9 MSMs of n+2..n+3 terms, 6+1+15+1 NTTs", "prove_ms": min(times) * 1e3,
            "gates_per_s": n / min(times), "compile_12_interpolations_ms": t_compile * 1e3, "slice_degree": degree,
            "round_ms": rounds, "expanded_srs_window_bits": expand, "phase_ms_one_proof": phases,
            "generate_proof_ms_with_transcript": t_full * 1e3, "generate_proof_ms_with_transcript_median": sorted(fulls)[len(fulls) // 2] * 1e3,
            "proof_degree": full["degree"],
            "verified_with_pairings": verdict == 1, "verify_ms": t_verify * 1e3}


def ntt_fr_products_per_element(log_n):
    """Field products per element of one Fr transform of 2^log_n elements as the library plans it (csrc/api.hip: get_plan; csrc/ntt.hpp):
    a radix-2^r pass runs r DIT stages of n/2 butterflies, stage 0 has twiddle 1 everywhere (no product) and half of the stage-1
    butterflies of a tile's first round have it too: r/2 - 3/4 products per element; every pass but the last multiplies each element
    by its inter-pass twiddle on the way out (one product; pass 0 reads it from a matrix up to 2^24 and forms it as a product of two
    table entries above: one more)."""
    if log_n <= 11:
        radices = [log_n]
    else:
        passes = -(-log_n // 8)
        for maxr in (9, 10):   # the narrowest wide radix that saves a pass (radix 2^10 only while cache-resident)
            if maxr == 10 and log_n > 20:
                break
            passes = min(passes, -(-log_n // maxr))
        base, rem = divmod(log_n, passes)
        radices = [base + (1 if p < rem else 0) for p in range(passes)]
    products = sum(max(0.0, r / 2 - 0.75) if r >= 2 else 0.0 for r in radices) + (len(radices) - 1) + (1 if (len(radices) > 1 and log_n > 24) else 0)
    return {"radices": radices, "products": products}


TRAFFIC_KERNEL_SOURCES = {"msm_accumulate": ["msm.hpp", "g1_28.hpp", "fq28.hpp", "fq28_mul_asm.inc", "fq28_mul2x_asm.inc", "fq28_sqr_asm.inc", "fq28_mul2_asm.inc", "ff.hpp"], "ntt_fr": ["ntt.hpp", "fr29.hpp", "fr29_mul2_asm.inc", "ff.hpp"]}  # = tools/update_traffic.py


def traffic_record(key):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/traffic.json), with the
    profile file and commit the figure came from -- a citation, not a counter taken with this run.  An entry measured before the
    kernel's source files last changed is still returned, with " [STALE: ...]" appended to its source: a warning, not a failure."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(tpath))
    except Exception:
        return None, None
    src = t.get("_source", {}).get(key)
    if t.get(key) is not None and src is not None:
        commit = t.get("_commit", {}).get(key)
        if commit:
            src += f" @ {commit}"
        want = t.get("_kernel_src_hash", {}).get(key)
        family = "ntt_fr" if key.startswith("ntt_fr") else "msm_accumulate"
        try:
            import hashlib
            h = hashlib.sha256()
            for f in TRAFFIC_KERNEL_SOURCES[family]:
                h.update(open(os.path.join(ROOT, "zkp-implementation_amd", "csrc", f), "rb").read())
            if want is None:
                src += " [STALE?: measured before entries carried a kernel source hash (rounds 1-4)]"
            elif want != h.hexdigest()[:16]:
                src += " [STALE: the kernel's source files changed after this counter pass]"
        except Exception:  # noqa: BLE001
            pass
    return t.get(key), src


def in_process_leg(args):
    """BASELINE configs[4] through the C ABI alone: ONE process drives every GPU (zkp_init_devices: one slot per device, one resident
    host thread per slot) -- the MSM over an SRS sharded at zkp_g1_bases_create with resident and with host scalars (the latter is
    what the Rust seam `evaluate_in_s` gets), and the four-step Fr NTT with its exchanges as peer copies inside the library
    (zkp_ntt_fr_sharded_dev / zkp_ntt_fr_sharded).  No torch.distributed, no RCCL: the second way a node can be used, and the one a
    node run still yields if the RCCL path misbehaves.  With more slots than GPUs the slots share the devices round-robin (a 1-GPU
    box rehearses the code path; the times then mean nothing for scaling and the object says so)."""
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: there is no CPU fallback for the hot path")
    import zkp_hip as zkp
    ngpu = torch.cuda.device_count()
    slots = args.in_process_slots or ngpu
    if slots & (slots - 1):
        slots = 1 << (slots.bit_length() - 1)  # the four-step transform wants a power of two; the MSM does not care
    devs = [i % ngpu for i in range(slots)]
    zkp.init_devices(devs)
    T = args.config4_log_n or 26
    res = {"workload": f"2^{T} terms / elements over {slots} device slots of ONE process, everything behind the C ABI "
                       "(zkp_init_devices, sharded zkp_bases, zkp_msm_g1_sharded_dev / zkp_msm_g1, zkp_ntt_fr_sharded_dev / zkp_ntt_fr_sharded)",
           "slots": slots, "visible_gpus": ngpu, "one_gpu_per_slot": ngpu >= slots, "total_log_n": T,
           "note": None if ngpu >= slots else "slots share devices: a rehearsal of the code path, not a scaling measurement"}
    for key, fn in (("ntt_fr_sharded", _in_process_ntt), ("msm", _in_process_msm)):
        try:
            res[key] = fn(zkp, torch, devs, T, args)
        except Exception as e:  # noqa: BLE001 -- one leg must not take the other with it
            res[key] = {"error": repr(e)}
        for d in set(devs):
            with torch.cuda.device(d):
                torch.cuda.empty_cache()
    zkp.shutdown()
    return res


def _in_process_ntt(zkp, torch, devs, T, args):
    G = len(devs)
    NAT, K1, COLS = zkp.NTT_NATURAL, zkp.NTT_K1SLAB, zkp.NTT_COLUMNS
    geo = zkp.ntt_fr_sharded_geometry(T)
    slabs = [rand_fr_tensor(torch, geo["slab"], 0x01770000 + T * 64 + g, torch.device("cuda", devs[g])) for g in range(G)]
    refs = [t.clone() for t in slabs]

    def run(inv, lin, lout):
        zkp.ntt_fr_sharded_dev(slabs, T, inverse=inv, layout_in=lin, layout_out=lout)   # synchronous: returns when every device is done

    def same():
        return all(bool(torch.equal(a, b)) for a, b in zip(slabs, refs))

    out = {"geometry": geo, "forms": {}}
    reps = 3 if T >= 24 else 10
    forms = (("two_exchanges", (False, NAT, K1), (True, K1, NAT), "natural slabs -> k1-slab layout -> natural slabs (mirrored inverse)"),
             ("one_exchange", (False, COLS, K1), (True, K1, COLS), "columns layout -> k1-slab layout -> columns layout"),
             ("natural_order", (False, NAT, NAT), (True, NAT, NAT), "natural order in and out both ways (what ark-poly's fft / ifft return): three exchanges"))
    names = ("ntt_sharded_pack", "ntt_sharded_columns", "ntt_sharded_exchange_wait", "ntt_sharded_rows", "ntt_sharded_unpack")
    for key, fwd, inv, what in forms:
        run(*fwd)
        run(*inv)
        ok = same()
        times, phases = {}, {}
        for tag, v in (("forward", fwd), ("inverse", inv)):
            best = None
            for _trial in range(2):
                t0 = time.perf_counter()
                for _ in range(reps):
                    run(*v)
                d1 = (time.perf_counter() - t0) / reps
                best = d1 if best is None or d1 < best else best
            times[tag + "_ms"] = best * 1e3
            zkp.profile_reset()
            zkp.profile_enable(True)
            run(*v)
            zkp.profile_enable(False)
            ph = {}
            for nm in names + ("ntt_fr_pass",):
                ms, cnt = zkp.profile_read(nm)
                if cnt:
                    ph[nm] = round(ms / G, 4)   # summed over the slots by the library: the average slot
            zkp.profile_reset()
            phases["phase_ms_per_slot_" + tag] = ph
        out["forms"][key] = {"what": what, **times, **phases, "roundtrip_identity": ok,
                             "elems_per_s_forward": (1 << T) / (times["forward_ms"] * 1e-3)}
        for a, b in zip(slabs, refs):
            a.copy_(b)
    del slabs, refs
    # the same total on ONE device through the single-device entry (slot 0): the denominator of the speedup
    with torch.cuda.device(devs[0]):
        data = rand_fr_tensor(torch, 1 << T, 0x01770000 + T * 64, torch.device("cuda", devs[0])).reshape(-1)
        zkp.ntt_fr_dev(data, T)
        torch.cuda.synchronize()
        one = {}
        for inv, key in ((False, "forward_ms"), (True, "inverse_ms")):
            best = None
            for _trial in range(2):
                t0 = time.perf_counter()
                for _ in range(reps):
                    zkp.ntt_fr_dev(data, T, inverse=inv)
                torch.cuda.synchronize()
                d1 = (time.perf_counter() - t0) / reps
                best = d1 if best is None or d1 < best else best
            one[key] = best * 1e3
        del data
        torch.cuda.empty_cache()
    out["one_device"] = one
    for key in out["forms"]:
        f = out["forms"][key]
        f["speedup_vs_one_device"] = one["forward_ms"] / f["forward_ms"]
        f["speedup_vs_one_device_inverse"] = one["inverse_ms"] / f["inverse_ms"]
    # the host-pointer form (zkp_ntt_fr_sharded, which zkp_ntt_fr itself takes from 2^24 on): every slot moves its slab over its own
    # PCIe link -- PCIe-inclusive, never a headline number
    if T <= 26:
        g0 = torch.Generator()
        g0.manual_seed(0x01770000 + T)
        h = torch.randint(0, 2 ** 62, ((1 << T), 4), dtype=torch.int64, generator=g0).numpy().view(np.uint64)
        y = h.copy()
        t0 = time.perf_counter()
        zkp.ntt_fr_sharded(y, inplace=True)
        t_f = time.perf_counter() - t0
        t0 = time.perf_counter()
        zkp.ntt_fr_sharded(y, inverse=True, inplace=True)
        t_i = time.perf_counter() - t0
        out["host_form"] = {"forward_ms": t_f * 1e3, "inverse_ms": t_i * 1e3, "roundtrip_identity": bool(np.array_equal(y, h)),
                            "note": "pageable host memory in and out (2 x 32 B per element over PCIe), natural order both ways"}
    return out


def _in_process_msm(zkp, torch, devs, T, args):
    from zkp_hip import trapdoor
    G, n = len(devs), 1 << T
    per = n // G
    h_pts = np.empty((n, 12), dtype=np.uint64)
    ks, sc = [], []
    t0 = time.perf_counter()
    for g in range(G):   # P_i = k_i G on each slot's own device (Srs::new_from_secret's kernel), then home: a caller's SRS is a host Vec
        d = torch.device("cuda", devs[g])
        zkp.set_device(g)
        with torch.cuda.device(d):
            k = rand_fr_tensor(torch, per, 0xBA5E0000 + T * 64 + g, d)
            pts = torch.zeros(per * 12, dtype=torch.int64, device=d)
            zkp.g1_fixed_base_mul_dev(k, per, pts)
            torch.cuda.synchronize()
            h_pts[g * per:(g + 1) * per] = pts.cpu().numpy().view(np.uint64).reshape(per, 12)
            del pts
            ks.append(k)
            sc.append(rand_fr_tensor(torch, per, 0x5EED0000 + T * 64 + g, d))
    zkp.set_device(-1)
    gen_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    bases = zkp.G1Bases.from_host(h_pts)     # sharded by contiguous chunk, one chunk resident per slot
    create_s = time.perf_counter() - t0
    del h_pts
    chunks = bases.shards()
    assert [c[3] for c in chunks] == [per] * G
    expand_ms = None
    if args.expand_bases:
        t0 = time.perf_counter()
        bases.precompute(0 if args.expand_bases < 0 else args.expand_bases)   # every chunk on its own device
        expand_ms = (time.perf_counter() - t0) * 1e3
    sums = [[0] * 16 for _ in range(16)]
    for g in range(G):
        with torch.cuda.device(devs[g]):
            part = trapdoor.limb_products(sc[g], ks[g])
        sums = [[a + b for a, b in zip(ra, rb)] for ra, rb in zip(sums, part)]
    res = zkp.msm_g1_sharded_dev(bases, sc, n)
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        res = zkp.msm_g1_sharded_dev(bases, sc, n)
    dt = (time.perf_counter() - t0) / reps
    out = {"ms_per_msm": dt * 1e3, "scalar_muls_per_s": n / dt, "bit_exact_full": check_against_trapdoor(zkp, sums, res),
           "window_bits": bases.info()[0], "insertions_per_scalar": bases.info()[1], "srs_expansion_ms": expand_ms,
           "base_point_generation_s": gen_s, "bases_create_from_host_s": create_s,
           "entry": "zkp_msm_g1_sharded_dev: one resident scalar array per chunk, one Pippenger per device, host-side add of the partial sums"}
    h_sc = np.concatenate([t.cpu().numpy().view(np.uint64).reshape(per, 4) for t in sc])
    got = zkp.msm_g1(bases, h_sc)
    t0 = time.perf_counter()
    for _ in range(2):
        got = zkp.msm_g1(bases, h_sc)
    dth = (time.perf_counter() - t0) / 2
    out["host_scalars"] = {"ms_per_msm": dth * 1e3, "scalar_muls_per_s": n / dth,
                           "same_result": bool(int(got[1]) == int(res[1]) and np.array_equal(got[0], res[0])),
                           "entry": "zkp_msm_g1 (what kzg/src/scheme.rs:84-96 binds): every slot's thread uploads its chunk's scalars over its own PCIe link"}
    bases.close()
    return out


def one_gpu_reference_bytes(log_n, planes=12):
    """Device memory rank 0 needs for `one_gpu_reference` at 2^log_n terms (the largest single-GPU footprint of a node run): the
    expanded SRS, its unexpanded copy during the expansion, scalars and discrete logs, and the MSM workspaces (digits, sort entries,
    double-buffered sorted indices for scalar ranges of at most 2^24, 2^21 buckets of 256 B x 4 arrays); then the NTT (data, a
    reference copy, the scratch slab).  Printed into extra.config4 and checked against the 288 GB of an MI355X by the CPU tests."""
    n = 1 << log_n
    rng = min(n, 1 << 24)
    msm = planes * n * 128 + n * 128 + 2 * n * 32 + n * 96 + planes * rng * (4 + 8 + 2 * 4) + 4 * (1 << 21) * 256
    ntt = 3 * n * 32
    return max(msm, ntt)


_EMIT = None  # rank 0: prints the JSON line exactly once (set by main() when the headline is complete)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 50 timed steps after 20 warm-up steps (~0.18 s of MSMs).  Three warm-up steps (rounds 1-3) left the first process on a
    # fresh box timing while the shader clock was still settling (2.03-2.09 GHz in the timed region against 2.15-2.26 later on the same box)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--log-n", type=int, default=20, help="log2 of MSM terms per GPU (default 20: BASELINE configs[1])")
    ap.add_argument("--total-log-n", type=int, default=0,
                    help="strong scaling: log2 of the TOTAL number of MSM terms, split evenly over the GPUs "
                         "(--gpus 8 --total-log-n 26 = BASELINE configs[4]); overrides --log-n")
    ap.add_argument("--ntt-log-n", type=int, default=24, help="log2 size of the secondary Fr NTT+iNTT measurement")
    ap.add_argument("--config4-log-n", type=int, default=int(os.environ.get("ZKP_BENCH_CONFIG4_LOG_N", "26")),
                    help="N > 1: log2 of the total size of the configs[4] extras (sharded MSM, four-step NTT); 0 = skip")
    ap.add_argument("--grid-max-log-n", type=int, default=int(os.environ.get("ZKP_BENCH_GRID_MAX", "26")),
                    help="N = 1: largest size of extra.msm_grid (2^22, 2^24, 2^26 up to this); 0 = skip")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary measurements")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-one-gpu-reference", action="store_true",
                    help="strong scaling: do not also time the whole problem on rank 0's GPU alone")
    ap.add_argument("--expand-bases", type=int, default=int(os.environ.get("ZKP_BENCH_EXPAND", "-1")),
                    help="window bits for zkp_g1_bases_precompute, the one-off SRS expansion that lets all windows share "
                         "one bucket set (default -1: the library's automatic width, 20 bits at 2^20 points and 22 from 2^22; "
                         "0 = plain per-window buckets over the unexpanded bases)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU work budget of the cpu_baseline sample")
    ap.add_argument("--in-process-leg", action="store_true",
                    help="run ONLY the in-process multi-device leg (one process, zkp_init_devices over --in-process-slots slots: sharded "
                         "MSM and the four-step Fr NTT through the C ABI, no torch.distributed) and print its JSON object; the N > 1 "
                         "run starts this as a child of rank 0 after its RCCL legs (extra.config4_in_process)")
    ap.add_argument("--in-process-slots", type=int, default=0, help="device slots of the in-process leg (0 = every visible GPU); more "
                                                                     "slots than GPUs share the devices round-robin (1-GPU rehearsal)")
    ap.add_argument("--no-in-process-leg", action="store_true", help="N > 1: skip extra.config4_in_process")
    ap.add_argument("--extras-deadline-s", type=float, default=float(os.environ.get("ZKP_BENCH_EXTRAS_DEADLINE_S", "420")),
                    help="N > 1: seconds after the headline is known at which rank 0 prints the line with the extras it has and "
                         "exits (a peer that died inside a collective must not take the headline with it)")
    args = ap.parse_args()
    if args.in_process_leg:
        print(json.dumps(in_process_leg(args)), flush=True)
        return

    # ---- launcher guard: BEFORE torch.cuda or the library are touched
    action = launcher_action(args.gpus, os.environ)
    if action == "spawn":
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    if action != "run":
        print(action, file=sys.stderr)
        sys.exit(2)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # ZKP_BENCH_DRYRUN=<tests/bench_dryrun_backend.py>: tests only -- THIS file's rank logic at any world size without a GPU, the
    # kernels stubbed on the CPU by the named test module (tests/test_bench_dryrun.py runs eight ranks that way; no box of the pool can
    # put eight processes on its card).  The line it prints says "dry_run": true, its metric string says so, and `value` is null.
    dry = os.environ.get("ZKP_BENCH_DRYRUN")
    stub = None
    if dry:
        import importlib.util
        spec = importlib.util.spec_from_file_location("zkp_bench_dryrun_backend", dry)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        stub = mod.install(torch)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: there is no CPU fallback for the hot path")
    # ZKP_BENCH_REHEARSAL=1: every rank on GPU 0 with the gloo backend -- lets a 1-GPU box exercise the N > 1 code path
    # (RCCL refuses two ranks on one device); never used by the driver's scaling runs.
    rehearsal = os.environ.get("ZKP_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    elif world > torch.cuda.device_count():
        raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} GPU(s) visible (ZKP_BENCH_REHEARSAL=1 puts every "
                         "rank on GPU 0 over gloo to rehearse the code path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cpu") if dry else torch.device("cuda", local_rank)
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if rehearsal else "nccl"
        import datetime
        tmo = datetime.timedelta(minutes=8)  # a rank that died must fail the job, not hang it
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=tmo)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")
    coll_device = "cpu" if rehearsal else device

    import zkp_hip as zkp
    from zkp_hip import dist as zdist
    if stub is not None:
        zkp = stub
    zkp.init(local_rank)

    strong = args.total_log_n > 0
    if strong:
        if world & (world - 1) or (1 << args.total_log_n) % world or (1 << args.total_log_n) // world < 1024:
            raise SystemExit("--total-log-n needs a power-of-two number of GPUs and at least 1024 terms per GPU")
        args.log_n = args.total_log_n - (world.bit_length() - 1)
    n = 1 << args.log_n

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sharded_step(wl):
        # per-GPU Pippenger on the local chunk, RCCL all-gather of the 192-byte partials, EC-add combine
        return lambda: zdist.msm_g1_sharded(zkp, wl.bases, wl.scalars, wl.n, device=device if (world > 1 and not rehearsal) else None)

    # ---- synthetic inputs, resident in HBM: this rank's contiguous chunk of the world * n term MSM
    wl = MsmWorkload(zkp, torch, device, args.log_n, chunk=rank, expand="auto" if args.expand_bases < 0 else args.expand_bases,
                     keep_points=True)
    if args.expand_bases < 0:
        args.expand_bases = wl.window_bits
    elapsed, result, phases = time_msm(zkp, torch, sharded_step(wl), args.steps, args.warmup, fence)
    elapsed = reduce_max(elapsed)
    sums = wl.limb_sums()
    if world > 1:
        sums = allreduce_limb_sums(torch, dist, sums, coll_device)
    bit_exact_full = check_against_trapdoor(zkp, sums, result)

    ms_per_step = 1e3 * elapsed / args.steps
    value = world * n * args.steps / elapsed
    acc_ms = phases["msm_accumulate"]
    clock = phases.pop("clock", {})
    slices = wl.planes if args.expand_bases else 16  # bucket insertions per scalar
    # v_mad_u64_u32 per mixed addition.  Round 5: every product of an insertion is a hand-written multiply-add chain (fq28.hpp,
    # tools/gen_fq28_mul_asm.py), so the count is by construction: six plain products of 392, two squarings of 301 (each off-diagonal
    # pair once against a doubled operand + 196 for the reduction), and Y3's two products with one reduction, 588.  (Rounds 1-3 priced
    # 3 724, round 4 counted 3 550 in the compiler's code.)  The first two insertions of a run are cheaper still (a copy, then six
    # products): not modelled, the figure is an upper bound by ~1.5 % at 2^20.
    MADS_PER_INSERTION = 6 * 392 + 2 * 301 + 588
    mads = n * slices * MADS_PER_INSERTION
    # box-proof form of the same numbers: cycles instead of milliseconds (the chip lowers its clock under load and boxes differ by
    # ~10 %: the same binary takes the same cycles and a different time), and the issue peak measured in this run on this box
    acc_mhz = clock.get("msm_accumulate_mhz")
    probe = clock.get("mad_probe") or {}
    n_simd = 4 * torch.cuda.get_device_properties(device).multi_processor_count
    acc_cycles = acc_ms * 1e-3 * acc_mhz * 1e6 if (acc_ms and acc_mhz) else None
    peak_s = probe.get("lane_mads_per_s") or MAD_PEAK
    peak_per_cycle = probe["lane_mads_per_s"] / (probe["clock_mhz"] * 1e6) if probe.get("clock_mhz") else None
    achieved = MSM_BYTES_PER_UNIT * n / (acc_ms * 1e-3) / 1e9 if acc_ms else None
    traffic, traffic_src = traffic_record(f"msm_accumulate_log{args.log_n}_c{args.expand_bases}")
    roofline = {"bound": "hbm", "kernel": "msm_accumulate_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                "traffic_source": traffic_src,
                "avg_kernel_ms": acc_ms, "algorithmic_bytes_per_launch": MSM_BYTES_PER_UNIT * n,
                "shader_clock_mhz": acc_mhz,  # held by msm_accumulate during the timed region (in-kernel stamps)
                "accumulate_cycles_per_launch": acc_cycles,
                "accumulate_simd_cycles_per_insertion": (acc_cycles * n_simd / (n * slices)) if acc_cycles else None,
                "clock_source": "s_memtime / s_memrealtime deltas of wave 0 of every msm_accumulate workgroup in the timed region "
                                "(zkp_profile_clock_read)" if acc_mhz else clock.get("error"),
                "phase_ms": phases,
                "phase_ms_note": "msm_accumulate: HIP events inside the timed region (the only phase recorded there); the other phases: a "
                                 "short fully recorded pass right after it",
                # the expanded SRS trades HBM bytes for arithmetic: every insertion gathers one 128 B record
                "traffic_by_design_bytes": (n * slices * 128 + n * slices * 4 + (1 << max(args.expand_bases - 1, 0)) * 256)
                if args.expand_bases else None,
                # informative: the bound that actually limits 381-bit arithmetic on 32-bit multipliers:
                # insertions per scalar x 3542 v_mad_u64_u32 per mixed add (six products, two squarings, one two-product reduction: asm chains), against the measured issue peak
                "integer_issue": {"insertions_per_scalar": slices, "lane_mads_per_insertion": MADS_PER_INSERTION, "lane_mads_per_launch": mads,
                                  "achieved_lane_mads_per_s": (mads / (acc_ms * 1e-3)) if acc_ms else None,
                                  "measured_peak_lane_mads_per_s": peak_s,
                                  "peak_source": "zkp_probe_mad_rate in this run, right after the timed region" if probe else
                                                 "round-1 constant (profiles/r01_issue_rate.txt): the probe failed",
                                  "probe_clock_mhz": probe.get("clock_mhz"), "probe_ms_per_launch": probe.get("ms_per_launch"),
                                  "frac": (mads / (acc_ms * 1e-3) / peak_s) if acc_ms else None,
                                  # the same ratio with both sides in shader cycles: independent of the clock either kernel held
                                  "peak_lane_mads_per_cycle": peak_per_cycle,
                                  "achieved_lane_mads_per_cycle": (mads / acc_cycles) if acc_cycles else None,
                                  "frac_in_cycles": (mads / acc_cycles / peak_per_cycle) if (acc_cycles and peak_per_cycle) else None,
                                  "round1_constant_lane_mads_per_s": MAD_PEAK}}

    out = {"metric": "G1 MSM scalar-muls/sec", "value": value, "unit": "scalar-muls/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
           "scaling": "strong" if strong else "weak", "vs_baseline": None,
           "dtype": "u32 (unsaturated 28-bit-limb Montgomery, 381-bit Fq; 64-bit accumulate)",
           "data": "synthetic", "bit_exact_full": bit_exact_full,
           "config": {"workload": f"Pippenger MSM, 2^{args.log_n} BLS12-381 G1 points per GPU" +
                                  (f" (2^{args.total_log_n} in total, BASELINE.json configs[4])" if strong else "") +
                                  ", scalars and bases resident in HBM, result bit-exact vs CPU (BASELINE.json configs[1]); " +
                                  (f"SRS expanded once outside the timed region to {slices} multiples 2^({args.expand_bases}s) P "
                                   "per point (zkp_g1_bases_precompute), one shared bucket set" if args.expand_bases else
                                   "unexpanded SRS, 16 bucket sets of 16-bit windows"),
                      "window_bits": args.expand_bases or 16, "expanded_bases": bool(args.expand_bases),
                      "log_n_per_gpu": args.log_n, "total_terms": world * n,
                      "world_size": dist.get_world_size() if world > 1 else 1, "collective_backend": backend,
                      "srs_expansion": {"ms": wl.expand_ms, "bytes": wl.expand_bytes, "planes": wl.planes,
                                        "base_point_generation_ms": wl.gen_ms} if args.expand_bases else None,
                      "parallelism": f"point/scalar chunk shard x{world} + RCCL all-gather of 192 B partial sums + EC add"},
           "roofline": roofline}
    if dry:
        out.update(dry_run=True, value=None, metric="DRY RUN of bench.py's rank logic, kernels stubbed on the CPU by " + os.path.basename(dry) +
                                                    " -- not a measurement (G1 MSM scalar-muls/sec in a real run)")
    extra = out.setdefault("extra", {})

    # ---- the line goes out exactly once.  The headline above is complete; everything below is secondary and, for N > 1, full of
    #      collectives: a peer that dies inside one leaves this rank blocked in RCCL (no Python exception to catch) until the launcher
    #      terminates the job.  Rank 0 therefore keeps a watchdog thread that prints the line with the extras gathered so far and
    #      exits when (a) the extras overrun --extras-deadline-s or (b) the launcher's SIGTERM arrives (signal.set_wakeup_fd: the
    #      C-level handler writes to a pipe at once, whatever the main thread is blocked in).
    import threading
    done, emit_lock, emitted = threading.Event(), threading.Lock(), []

    def emit(note=None):
        with emit_lock:
            if emitted:
                return
            emitted.append(1)
            line = None
            for _ in range(5):  # the main thread may be adding an extra right now
                try:
                    if note:
                        extra["extras_cut_short"] = note
                    line = json.dumps(out)
                    break
                except RuntimeError:
                    time.sleep(0.01)
            if line is None:
                line = json.dumps({k: v for k, v in out.items() if k != "extra"} | {"extra": {"extras_cut_short": note or "unserialisable"}})
            print(line, flush=True)

    global _EMIT
    if rank == 0:
        _EMIT = emit
    if rank == 0 and world > 1:
        import select
        import signal
        rfd, wfd = os.pipe()
        os.set_blocking(wfd, False)
        signal.signal(signal.SIGTERM, lambda *_: None)  # a Python-level handler must exist for the wake-up byte to be written
        signal.set_wakeup_fd(wfd, warn_on_full_buffer=False)
        t_deadline = time.monotonic() + args.extras_deadline_s

        def watchdog():
            while not done.is_set():
                left = t_deadline - time.monotonic()
                r, _, _ = select.select([rfd], [], [], max(0.0, min(left, 1.0)))
                if done.is_set():
                    return
                if r:
                    emit("terminated by the launcher (a peer rank failed?) after the headline: the extras are incomplete")
                    os._exit(143)
                if left <= 0:
                    emit(f"the extras overran --extras-deadline-s {args.extras_deadline_s:.0f}: the line carries what was finished by then")
                    os._exit(0)

        threading.Thread(target=watchdog, daemon=True).start()
    if os.environ.get("ZKP_BENCH_TEST_FAIL_RANK") == str(rank) and world > 1:  # test hook: a peer dies right after the headline
        os._exit(3)

    # ---- strong scaling: the same total on rank 0's GPU alone (the other ranks wait), for the speedup in the same line
    if strong and world > 1 and not args.no_one_gpu_reference and not args.no_extra:
        fence()
        if rank == 0:
            try:
                t1 = time.perf_counter()
                ks_all = torch.cat([rand_fr_tensor(torch, n, 0xBA5E0000 + args.log_n * 64 + r, device) for r in range(world)])
                sc_all = torch.cat([rand_fr_tensor(torch, n, 0x5EED0000 + args.log_n * 64 + r, device) for r in range(world)])
                nt = world * n
                pts_all = torch.zeros(nt * 12, dtype=torch.int64, device=device)
                zkp.g1_fixed_base_mul_dev(ks_all, nt, pts_all)
                torch.cuda.synchronize()
                b_all = zkp.G1Bases.from_device(pts_all, nt)
                del pts_all
                if args.expand_bases:
                    b_all.precompute(0)
                torch.cuda.synchronize()
                setup_s = time.perf_counter() - t1
                reps = 2
                r1 = zkp.msm_g1_dev(b_all, sc_all, nt)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(reps):
                    r1 = zkp.msm_g1_dev(b_all, sc_all, nt)
                one_ms = (time.perf_counter() - t1) / reps * 1e3
                extra["one_gpu_same_total"] = {"total_log_n": args.total_log_n, "ms_per_msm": one_ms,
                                               "speedup_of_this_run": one_ms / ms_per_step, "setup_s": setup_s,
                                               "same_result": bool(np.array_equal(r1[0], result[0]))}
                b_all.close()
                del ks_all, sc_all
                torch.cuda.empty_cache()
            except Exception as e:  # noqa: BLE001 -- the headline number must not depend on the secondary measurement
                extra["one_gpu_same_total"] = {"error": repr(e)}
        fence()

    single = world == 1 and rank == 0 and not args.no_extra

    # ---- PCIe-inclusive rate: the same MSM with the scalars in pageable host memory (zkp_msm_g1, what KzgScheme::commit binds)
    if single:
        try:
            h_sc = wl.scalars.cpu().numpy().view(np.uint64).reshape(-1, 4)
            got_h = zkp.msm_g1(wl.bases, h_sc)
            trials_h = []   # three loops of five calls, the best loop (this path is host-latency-sensitive: an uploader thread, polling)
            for _trial in range(3):
                t1 = time.perf_counter()
                for _ in range(5):
                    got_h = zkp.msm_g1(wl.bases, h_sc)
                trials_h.append((time.perf_counter() - t1) / 5)
            dt = min(trials_h)
            # the PCIe share as numbers of their own: the raw pageable-host -> device rate of the same 32 B x n, and the upload of the
            # first range (the first 25 % of the scalars), which is the part no kernel can run under
            h_t = torch.from_numpy(h_sc.view(np.int64).reshape(-1))
            d_t = torch.empty_like(wl.scalars.reshape(-1))
            first = max(1024, (n * (10 if n >= 1 << 21 else 25) // 100) & ~1023) * 4   # = msm_host_scalars' first range (api.hip)
            ups = {}
            for key, cnt in (("whole", 4 * n), ("first_range", first)):
                best = None
                for _ in range(5):
                    torch.cuda.synchronize()
                    t2 = time.perf_counter()
                    d_t[:cnt].copy_(h_t[:cnt])
                    torch.cuda.synchronize()
                    d2 = time.perf_counter() - t2
                    best = d2 if best is None or d2 < best else best
                ups[key] = best
            extra["msm_h2d_inclusive"] = {"workload": f"same 2^{args.log_n} MSM, scalars uploaded from pageable host memory inside "
                                                      "the timed call (32 B per scalar over PCIe, in two ranges below 2^21 terms -- the last 75 % uploaded, by a second host thread, under the kernels of the first 25 % -- and three from there, 10 % + 30 % + 60 %)",
                                          "ms_per_step": dt * 1e3, "scalar_muls_per_s": n / dt, "ms_per_step_trials": [round(t * 1e3, 4) for t in trials_h],
                                          "timing": "best of 3 loops x 5 calls",
                                          "same_result": bool(np.array_equal(got_h[0], result[0])),
                                          "h2d_pageable_GBs": 32 * n / ups["whole"] / 1e9, "h2d_whole_upload_ms": ups["whole"] * 1e3,
                                          "exposed_upload_ms": ups["first_range"] * 1e3,
                                          "exposed_upload_note": "the upload of the first range (25 % of the scalars; 10 % from 2^21 terms) measured on its own: the part of "
                                                                 "the PCIe transfer no kernel runs under; the rest overlaps the first range's kernels",
                                          "ms_over_resident_scalars": dt * 1e3 - ms_per_step}
            del h_t, d_t
            del h_sc
        except Exception as e:  # noqa: BLE001
            extra["msm_h2d_inclusive"] = {"error": repr(e)}

    # ---- throughput of a batch: four commitments over the same SRS in ONE pass through the kernels (zkp_msm_g1_batch_dev, what the
    #      PLONK prover uses for its groups of commitments): four bucket sets side by side, one sort / reduction launch sequence
    if single and args.expand_bases:
        try:
            others = [rand_fr_tensor(torch, n, 0x5EED1000 + args.log_n * 64 + j, device) for j in range(3)]
            vecs = [wl.scalars] + others
            got_b = zkp.msm_g1_batch_dev(wl.bases, vecs, n)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                got_b = zkp.msm_g1_batch_dev(wl.bases, vecs, n)
            dt = (time.perf_counter() - t1) / 5
            from zkp_hip import trapdoor
            ok_b = bool(np.array_equal(got_b[0][0], result[0]))
            for j in range(3):
                ok_b = ok_b and check_against_trapdoor(zkp, trapdoor.limb_products(others[j], wl.ks), got_b[j + 1])
            extra["msm_batch_of_4"] = {"workload": f"four 2^{args.log_n}-term MSMs over the same expanded SRS in one pass "
                                                   "(zkp_msm_g1_batch_dev: commit_round1 / SlicePoly::commit style groups)",
                                       "ms_per_batch": dt * 1e3, "ms_per_msm": dt * 1e3 / 4, "scalar_muls_per_s": 4 * n / dt,
                                       "bit_exact_all_four": ok_b}
            del others, vecs
        except Exception as e:  # noqa: BLE001
            extra["msm_batch_of_4"] = {"error": repr(e)}

    # ---- the same MSM over the UNEXPANDED bases (per-window buckets, no SRS preprocessing at all), for comparison
    if single and args.expand_bases:
        plain = zkp.G1Bases.from_device(wl.pts, n)
        for _ in range(2):
            zkp.msm_g1_dev(plain, wl.scalars, n)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(10):
            got_plain = zkp.msm_g1_dev(plain, wl.scalars, n)
        dt = (time.perf_counter() - t1) / 10
        extra["msm_unexpanded_bases"] = {"workload": f"same 2^{args.log_n} MSM, 16-bit windows over the original points",
                                         "ms_per_step": dt * 1e3, "scalar_muls_per_s": n / dt,
                                         "same_result_as_expanded": bool(np.array_equal(got_plain[0], result[0]))}
        plain.close()
        del plain

    # ---- CPU baseline: the oracle's reference-faithful naive MSM on a bounded sample (rank 0, N = 1 only).  Taken here, while the
    #      2^20 workload is still resident; the larger sizes below free it.
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        from oracle import oracle as orc
        orc.build()
        h_pts = wl.pts[: 12 * min(n, 1 << 16)].cpu().numpy().view(np.uint64).reshape(-1, 12)
        h_sc = wl.scalars[: min(n, 1 << 16)].cpu().numpy().view(np.uint64).reshape(-1, 4)
        probe = min(64, n)
        t2 = time.perf_counter()
        orc.msm_naive(h_pts[:probe], None, h_sc[:probe])
        per = (time.perf_counter() - t2) / probe
        m = int(max(probe, min(len(h_sc), args.cpu_seconds / per)))
        t3 = time.perf_counter()
        exp, einf = orc.msm_naive(h_pts[:m], None, h_sc[:m])
        cpu_dt = time.perf_counter() - t3
        sub = zkp.G1Bases.from_device(wl.pts[: 12 * m].contiguous(), m)
        got, ginf = zkp.msm_g1_dev(sub, wl.scalars[:m].contiguous(), m)
        out["cpu_baseline"] = {"value": m / cpu_dt, "unit": "scalar-muls/s", "cores": 1, "kind": "port",
                               "sample": f"first {m} (scalar, point) pairs of the same workload through the oracle's "
                                         "restatement of kzg/src/scheme.rs:88-94 (n scalar-muls + 2n inversions), "
                                         f"{cpu_dt:.1f} s single-thread",
                               "host_cores_available": os.cpu_count(),
                               "gpu_bit_exact_on_sample": bool(ginf == einf and np.array_equal(got, exp))}
        # context (CPU (ii)): what THIS host could do with a CPU algorithm of the same family -- the
        # bucket method and the ark-poly style radix-2 NTT, on one core and on all the cores this process may use (OpenMP, oracle side:
        # test infrastructure, never the product).  Not the reference's algorithm; `value` above stays the reference-faithful number.
        try:
            mp = min(len(h_sc), 1 << 15)
            t4 = time.perf_counter()
            orc.msm_pippenger(h_pts[:mp], None, h_sc[:mp])
            pip_dt = time.perf_counter() - t4
            ln_c = 18
            hv = orc.rand_fr(0x01770000 + ln_c, 1 << ln_c)
            t5 = time.perf_counter()
            orc.ntt_fr(hv)
            ntt_dt = time.perf_counter() - t5
            try:
                usable = len(os.sched_getaffinity(0))
            except AttributeError:
                usable = os.cpu_count() or 1
            quota = None  # a container may grant fewer CPU-seconds per second than the affinity mask has CPUs (cgroup v2 cpu.max)
            try:
                q, per = open("/sys/fs/cgroup/cpu.max").read().split()
                quota = None if q == "max" else float(q) / float(per)
            except Exception:  # noqa: BLE001
                pass
            threads = max(1, min(usable, orc.max_threads()))
            if quota:  # threads beyond the quota only get throttled (the GPU boxes of this pool: 128 CPUs in the mask, a quota of 16)
                threads = max(1, min(threads, int(quota + 0.999)))
            h_pts_all = wl.pts.cpu().numpy().view(np.uint64).reshape(-1, 12)
            h_sc_all = wl.scalars.cpu().numpy().view(np.uint64).reshape(-1, 4)
            t6 = time.perf_counter()
            mt_res = orc.msm_pippenger_mt(h_pts_all, None, h_sc_all, threads)
            mt_dt = time.perf_counter() - t6
            ln_m = min(args.ntt_log_n, 24)
            hv2 = orc.rand_fr(0x01770000 + ln_m, 1 << ln_m)
            t7 = time.perf_counter()
            orc.ntt_fr_mt(hv2, threads, inplace=True)
            ntt_mt_dt = time.perf_counter() - t7
            del hv2
            out["cpu_baseline"]["context"] = {
                "cores": threads, "nproc": os.cpu_count(), "threads_note": "OpenMP threads = min(CPUs in this process's affinity mask, cgroup CPU quota)", "affinity_cpus": usable,
                "cgroup_cpu_quota_cores": quota,
                "pippenger_all_cores_scalar_muls_per_s": n / mt_dt,
                "pippenger_all_cores_sample": f"all 2^{args.log_n} pairs of the workload, {mt_dt:.2f} s on {threads} threads",
                "pippenger_all_cores_same_result_as_gpu": bool(int(mt_res[1]) == int(result[1]) and np.array_equal(mt_res[0], np.asarray(result[0], dtype=np.uint64))),
                "ntt_fr_all_cores_elems_per_s": (1 << ln_m) / ntt_mt_dt,
                "ntt_all_cores_sample": f"2^{ln_m} elements, {ntt_mt_dt:.2f} s on {threads} threads",
                "pippenger_1core_scalar_muls_per_s": mp / pip_dt, "pippenger_sample": f"first {mp} pairs, {pip_dt:.2f} s",
                "ntt_fr_1core_elems_per_s": (1 << ln_c) / ntt_dt, "ntt_sample": f"2^{ln_c} elements, {ntt_dt:.2f} s"}
            del h_pts_all, h_sc_all
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"]["context"] = {"error": repr(e)}
        sub.close()
    wl.close()
    del wl

    # ---- the north-star sizes on ONE GPU (SURVEY 8d reporting grid, 1-GPU column): same code path, inputs resident
    if single and args.grid_max_log_n:
        grid = {}
        for ln in (22, 24, 26):
            if ln > args.grid_max_log_n or ln == args.log_n:
                continue
            try:
                t_setup = time.perf_counter()
                g = MsmWorkload(zkp, torch, device, ln, chunk=0, expand="auto" if args.expand_bases else 0)
                setup_s = time.perf_counter() - t_setup
                reps = 4 if ln <= 22 else 2
                el, res, ph = time_msm(zkp, torch, lambda: zkp.msm_g1_dev(g.bases, g.scalars, g.n), reps, 1,
                                       torch.cuda.synchronize)
                ok = check_against_trapdoor(zkp, g.limb_sums(), res)
                gacc = ph["msm_accumulate"]
                gtraffic, gtraffic_src = traffic_record(f"msm_accumulate_log{ln}_c{g.window_bits}")
                launches = max(1, g.n >> 24) if (g.planes and g.planes <= 12) else max(1, g.n >> 23)  # scalar ranges per MSM (api.hip: msm_partial_batch, pre_planes <= 12)
                gclk = (ph.get("clock") or {}).get("msm_accumulate_mhz")
                gsimd = 4 * torch.cuda.get_device_properties(device).multi_processor_count
                grid[f"2^{ln}"] = {"ms_per_msm": el / reps * 1e3, "scalar_muls_per_s": g.n * reps / el,
                                   "traffic": gtraffic, "traffic_source": gtraffic_src, "traffic_launches_per_msm": launches,
                                   "traffic_bytes_per_insertion": (gtraffic * launches / (g.n * g.planes)) if gtraffic and g.planes else None,
                                   "msm_accumulate_ms": gacc, "phase_ms": ph, "shader_clock_mhz": gclk,
                                   "accumulate_simd_cycles_per_insertion": (gacc * 1e-3 * gclk * 1e6 * gsimd / (g.n * g.planes))
                                   if (gacc and gclk and g.planes) else None,
                                   "phase_note": "above 2^24 the scalars are walked in ranges of 2^24: digits + sort of the next range run on "
                                                 "a second stream underneath the accumulate, their wall time overlaps it" if ln > 24 else None,
                                   "roofline_frac": (MSM_BYTES_PER_UNIT * g.n / (gacc * 1e-3) / 1e9 / HBM_PEAK_GBS) if gacc else None,
                                   "whole_msm_hbm_algorithmic_frac": MSM_BYTES_PER_UNIT * g.n * reps / el / 1e9 / HBM_PEAK_GBS,
                                   "bit_exact_full": ok, "window_bits": g.window_bits, "insertions_per_scalar": g.planes,
                                   "srs_expansion_ms": g.expand_ms, "srs_expansion_bytes": g.expand_bytes,
                                   "base_point_generation_ms": g.gen_ms, "setup_s": setup_s}
                if ln <= 24:  # the caller's number at this size: zkp_msm_g1 with the scalars in pageable host memory (as extra.msm_h2d_inclusive)
                    try:
                        h_sc = g.scalars.cpu().numpy().view(np.uint64).reshape(-1, 4).copy()
                        got_h = zkp.msm_g1(g.bases, h_sc)
                        t_h = time.perf_counter()
                        for _ in range(reps):
                            got_h = zkp.msm_g1(g.bases, h_sc)
                        dt_h = (time.perf_counter() - t_h) / reps
                        grid[f"2^{ln}"]["host_scalars_ms_per_msm"] = dt_h * 1e3
                        grid[f"2^{ln}"]["host_scalars_same_result"] = bool(np.array_equal(got_h[0], res[0]))
                        del h_sc
                    except Exception as e:  # noqa: BLE001
                        grid[f"2^{ln}"]["host_scalars_error"] = repr(e)
                g.close()
                del g
            except Exception as e:  # noqa: BLE001
                grid[f"2^{ln}"] = {"error": repr(e)}
                torch.cuda.empty_cache()
        extra["msm_grid"] = {"workload": "the same MSM at the north-star sizes on this one GPU (expanded SRS, scalars and bases "
                                         "resident; algorithmic bytes 128 B per scalar-mul against 8 TB/s)", **grid}

    # ---- secondary metric of BASELINE.json: Fr NTT + iNTT round trip (configs[2]), rank 0's GPU only
    if single:
        ln = args.ntt_log_n
        m = 1 << ln
        data = rand_fr_tensor(torch, m, 0x01770000 + ln, device).reshape(-1)
        ref = data.clone()
        # untimed: the GPU has been idle since the grid above released its 100 GB of SRS planes, and its clocks take tens of
        # milliseconds to come back (tools/clock_probe.py: the first 20 ms after 3 s of idle run 15 % slow)
        for _ in range(12):
            zkp.ntt_fr_dev(data, ln)
            zkp.ntt_fr_dev(data, ln, inverse=True)
        torch.cuda.synchronize()
        ok = bool(torch.equal(data, ref))
        reps, dt, trials = 10, None, []
        for _trial in range(3):  # best of three: the loop that follows a host synchronisation runs ~7 % slow (GPU clocks dip at once)
            t1 = time.perf_counter()
            for _ in range(reps):
                zkp.ntt_fr_dev(data, ln)
                zkp.ntt_fr_dev(data, ln, inverse=True)
            torch.cuda.synchronize()
            d1 = (time.perf_counter() - t1) / reps
            trials.append(d1)
            dt = d1 if dt is None or d1 < dt else dt
        # the per-pass kernel time from a second, marked loop: the event markers between the passes stay out of the number above
        zkp.profile_reset()
        zkp.profile_enable(True)
        for _ in range(reps):
            zkp.ntt_fr_dev(data, ln)
            zkp.ntt_fr_dev(data, ln, inverse=True)
        torch.cuda.synchronize()
        zkp.profile_enable(False)
        pms, pcnt = zkp.profile_read("ntt_fr_pass")
        # the same box-proof form the MSM has: the shader clock the passes held (in-kernel stamps of one workgroup in sixteen), the
        # issue peak probed right after them, and the multiply-adds of the transform's field products against it, in cycles
        nclk = {}
        try:
            cyc, ref_t, stamped = zkp.profile_clock_read("ntt_fr_pass")
            mhz = 100.0 * cyc / ref_t if ref_t else None
            rate, pmhz, _pms = zkp.probe_mad_rate(20)
            prods = ntt_fr_products_per_element(ln)
            n_simd = 4 * torch.cuda.get_device_properties(device).multi_processor_count
            t_transform = (pms / (2 * reps)) * 1e-3 if pcnt else None  # kernel time of one transform (forward and inverse alike)
            cycles = t_transform * mhz * 1e6 if (t_transform and mhz) else None
            peak_per_cycle = rate / (pmhz * 1e6) if pmhz else None
            nclk = {"shader_clock_mhz": mhz, "stamped_workgroups": stamped, "kernel_ms_per_transform": t_transform * 1e3 if t_transform else None,
                    "cycles_per_transform": cycles,
                    "simd_cycles_per_element_per_pass": (cycles * n_simd / m / (pcnt // (2 * reps))) if (cycles and pcnt) else None,
                    "field_products_per_element": prods["products"], "passes": prods["radices"],
                    "lane_mads_per_field_product": 162, "lane_mads_per_element": prods["products"] * 162,
                    "integer_issue": {"measured_peak_lane_mads_per_s": rate, "probe_clock_mhz": pmhz,
                                      "peak_lane_mads_per_cycle": peak_per_cycle,
                                      "achieved_lane_mads_per_cycle": (prods["products"] * 162 * m / cycles) if cycles else None,
                                      "frac_in_cycles": (prods["products"] * 162 * m / cycles / peak_per_cycle) if (cycles and peak_per_cycle) else None,
                                      "note": "multiply-adds of the field products only (162 v_mad_u64_u32 each); the product's own carry / mask / "
                                              "shift instructions and everything between products (LDS exchange, lazy-reduction fix-ups, limb "
                                              "conversion at load and store) are the rest: profiles/r05_ntt_isa_histogram.md"}}
        except Exception as e:  # noqa: BLE001
            nclk = {"clock_error": repr(e)}
        zkp.profile_reset()
        ntraffic, ntraffic_src = traffic_record(f"ntt_fr_transform_log{ln}")  # bytes per transform from the committed PMC passes
        extra["ntt_fr"] = {"workload": f"Fr NTT + iNTT round trip, 2^{ln} elements, 1 GPU (BASELINE configs[2])",
                           "hbm_traffic_bytes_per_transform": ntraffic, "traffic_source": ntraffic_src,
                           "hbm_GBs_from_counters": (2 * ntraffic / dt / 1e9) if ntraffic else None,
                           "hbm_frac_from_counters": (2 * ntraffic / dt / 1e9 / HBM_PEAK_GBS) if ntraffic else None,
                           "elems_per_s_per_transform": 2 * m / dt, "roundtrip_ms": dt * 1e3, "timing": f"best of 3 x {reps} round trips",
                           "roundtrip_ms_median": sorted(trials)[1] * 1e3, "roundtrip_ms_trials": [round(t * 1e3, 4) for t in trials],
                           "kernel_ms_per_roundtrip": (pms / reps) if pcnt else None,
                           "roundtrip_identity": ok,
                           "hbm_algorithmic_GBs": 2 * NTT_BYTES_PER_ELEM * m / dt / 1e9,
                           "hbm_frac": 2 * NTT_BYTES_PER_ELEM * m / dt / 1e9 / HBM_PEAK_GBS,
                           "avg_pass_kernel_ms": pms / pcnt if pcnt else None, "passes_per_transform":
                           (pcnt // (2 * reps)) if pcnt else None, **nclk}
        del data, ref
        torch.cuda.empty_cache()

    # ---- the NTT half of the SURVEY 8d grid on one GPU: forward Fr transforms at 2^20 .. 2^26, round trip checked
    if single:
        ngrid = {}
        for ln in (20, 22, 24, 26):
            if ln > max(args.grid_max_log_n, args.ntt_log_n):
                continue
            try:
                m = 1 << ln
                data = rand_fr_tensor(torch, m, 0x01770000 + ln, device).reshape(-1)
                ref = data.clone()
                zkp.ntt_fr_dev(data, ln)
                zkp.ntt_fr_dev(data, ln, inverse=True)
                torch.cuda.synchronize()
                ok = bool(torch.equal(data, ref))
                del ref
                reps = 40 if ln <= 22 else 10 if ln <= 24 else 4
                for _ in range(3 if ln <= 24 else 1):  # untimed
                    zkp.ntt_fr_dev(data, ln)
                torch.cuda.synchronize()
                dt, trials = None, []
                for _trial in range(3):  # best of three, median beside it (see above)
                    t1 = time.perf_counter()
                    for _ in range(reps):
                        zkp.ntt_fr_dev(data, ln)
                    torch.cuda.synchronize()
                    d1 = (time.perf_counter() - t1) / reps
                    trials.append(d1)
                    dt = d1 if dt is None or d1 < dt else dt
                ngrid[f"2^{ln}"] = {"forward_ms": dt * 1e3, "forward_ms_median": sorted(trials)[1] * 1e3, "elems_per_s": m / dt, "hbm_algorithmic_GBs": NTT_BYTES_PER_ELEM * m / dt / 1e9,
                                    "hbm_frac": NTT_BYTES_PER_ELEM * m / dt / 1e9 / HBM_PEAK_GBS, "roundtrip_identity": ok}
                del data
                torch.cuda.empty_cache()
            except Exception as e:  # noqa: BLE001
                ngrid[f"2^{ln}"] = {"error": repr(e)}
        extra["ntt_grid"] = {"workload": "forward Fr NTT (natural order in and out) on this one GPU, data resident; 64 B per element "
                                         "algorithmic against 8 TB/s", **ngrid}

    # ---- FRI commitment path (Goldilocks polynomial of 2^20 coefficients, blowup 2), rank 0's GPU only
    if single:
        try:
            rnd = np.random.default_rng(0x0F21)
            fc = rnd.integers(1, 2 ** 63, 1 << 20, dtype=np.uint64)
            zkp.fri_prove(fc, 2, 32)
            zkp.profile_reset()
            zkp.profile_enable(True)
            t1 = time.perf_counter()
            proof = zkp.fri_prove(fc, 2, 32)
            dt = time.perf_counter() - t1
            zkp.profile_enable(False)
            mk_ms, mk_cnt = zkp.profile_read("fri_merkle")
            nt_ms, nt_cnt = zkp.profile_read("ntt_gl_pass")
            zkp.profile_reset()
            extra["fri"] = {"workload": "FRI generate_proof, Goldilocks, 2^20 coefficients, blowup 2, 32 queries, 1 GPU "
                                        "(coset NTT + SHA-256 Merkle tree + fold per layer, host transcript)",
                            "prove_ms": dt * 1e3, "coeffs_per_s": (1 << 20) / dt, "merkle_ms": mk_ms, "merkle_trees": mk_cnt,
                            "ntt_ms": nt_ms, "ntt_passes": nt_cnt, "proof_bytes": int(proof.size) * 8,
                            "verified": bool(zkp.fri_verify(proof))}
        except Exception as e:  # noqa: BLE001
            extra["fri"] = {"error": repr(e)}

    # ---- BASELINE configs[3]: PLONK prover, 2^16-gate synthetic circuit, 1 GPU (MSM + NTT combined, KZG opens)
    if single:
        try:
            extra["plonk"] = bench_plonk(zkp, torch, device, 16, expand="auto" if args.expand_bases else 0)
        except Exception as e:  # noqa: BLE001 -- the headline number must not depend on the secondary measurement
            extra["plonk"] = {"error": repr(e)}

    # ---- N > 1 GPUs: the SURVEY 8d grid sharded over the ranks -- total sizes 2^22, 2^24 and BASELINE configs[4]'s 2^26: the MSM by
    #      point/scalar chunk with the all-gather of the partial sums, the Fr NTT as the four-step transform with its all-to-alls
    if world > 1 and not args.no_extra and args.config4_log_n and not (world & (world - 1)):
        grid = {}
        for T in sorted({t for t in (22, 24, args.config4_log_n) if t <= args.config4_log_n}):
            per = T - (world.bit_length() - 1)
            if per < 12:
                continue
            c4 = {"total_log_n": T, "log_n_per_gpu": per}
            try:
                if strong and args.total_log_n == T:  # the headline IS this measurement
                    c4["msm"] = {"see": "the headline fields of this line"}
                else:
                    g = MsmWorkload(zkp, torch, device, per, chunk=rank, expand="auto" if args.expand_bases else 0)
                    reps = 3
                    el, res, ph = time_msm(zkp, torch, sharded_step(g), reps, 1, fence)
                    el = reduce_max(el)
                    ok = check_against_trapdoor(zkp, allreduce_limb_sums(torch, dist, g.limb_sums(), coll_device), res)
                    c4["msm"] = {"ms_per_msm": el / reps * 1e3, "scalar_muls_per_s": (1 << T) * reps / el, "phase_ms_rank0": ph,
                                 "bit_exact_full": ok, "window_bits": g.window_bits, "insertions_per_scalar": g.planes,
                                 "srs_expansion_ms_rank0": g.expand_ms}
                    g.close()
                    del g
            except Exception as e:  # noqa: BLE001
                c4["msm"] = {"error": repr(e)}
            fence()
            try:
                c4["ntt_fr_four_step"] = bench_four_step(zkp, zdist, torch, dist, device, T, rank, world, fence, reduce_max)
            except Exception as e:  # noqa: BLE001
                c4["ntt_fr_four_step"] = {"error": repr(e)}
            torch.cuda.empty_cache()
            # BASELINE configs[4] / ">= 6x whole-node speedup": the same total on rank 0's GPU ALONE (the other ranks wait at the
            # barrier), so that the plain `--gpus N` line carries the strong-scaling verdict without --total-log-n
            if T == args.config4_log_n and not args.no_one_gpu_reference:
                fence()
                if rank == 0:
                    one = one_gpu_reference(zkp, torch, device, T, bool(args.expand_bases))
                    c4["one_gpu_same_total"] = one
                    m_ms = ms_per_step if (strong and args.total_log_n == T) else c4.get("msm", {}).get("ms_per_msm")
                    if m_ms and one.get("msm_ms"):
                        c4.setdefault("msm", {})["one_gpu_ms"] = one["msm_ms"]
                        c4["msm"]["speedup_vs_one_gpu"] = one["msm_ms"] / m_ms
                    f4 = c4.get("ntt_fr_four_step", {})
                    if f4.get("forward_ms") and one.get("ntt_forward_ms"):
                        f4["one_gpu_ms"] = one["ntt_forward_ms"]
                        f4["speedup_vs_one_gpu"] = one["ntt_forward_ms"] / f4["forward_ms"]
                        f4["one_gpu_inverse_ms"] = one["ntt_inverse_ms"]
                        f4["speedup_vs_one_gpu_inverse"] = one["ntt_inverse_ms"] / f4["inverse_ms"]
                        ox = f4.get("one_exchange") or {}
                        if ox.get("forward_ms"):
                            ox["speedup_vs_one_gpu"] = one["ntt_forward_ms"] / ox["forward_ms"]
                            ox["speedup_vs_one_gpu_inverse"] = one["ntt_inverse_ms"] / ox["inverse_ms"]
                fence()
            c4["rccl_world_size"] = dist.get_world_size()
            c4["collective_backend"] = backend
            grid[f"2^{T}"] = c4
        extra["sharded_grid"] = {"workload": f"total sizes sharded over {world} GPUs (MSM: chunk per GPU + all-gather of 192 B partials; "
                                             "NTT: four-step with all-to-all exchanges)", **grid}
        if f"2^{args.config4_log_n}" in grid:
            extra["config4"] = grid[f"2^{args.config4_log_n}"]  # BASELINE.json configs[4] (2^26 unless overridden)
            extra["config4"]["one_gpu_memory_estimate_gb"] = one_gpu_reference_bytes(args.config4_log_n) / 1e9

    # ---- the same configs[4] a second way: ONE process over every GPU, through the C ABI alone (in_process_leg).  A child of rank 0,
    #      started after the RCCL legs while the other ranks wait at the barrier with their device memory released: a failing child
    #      is an error string in the line, nothing else (no retry, no re-exec of this process).
    if world > 1 and not args.no_extra and not args.no_in_process_leg and args.config4_log_n:
        torch.cuda.empty_cache()
        fence()
        if rank == 0:
            cmd = [sys.executable, os.path.abspath(__file__), "--in-process-leg", "--config4-log-n", str(args.config4_log_n),
                   "--in-process-slots", str(world), "--expand-bases", str(-1 if args.expand_bases else 0)]
            env = {k: v for k, v in os.environ.items()
                   if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT")
                   and not k.startswith("TORCHELASTIC")}
            t_leg = time.perf_counter()
            try:
                p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
                lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
                leg = json.loads(lines[-1]) if (p.returncode == 0 and lines) else {"error": f"exit code {p.returncode}", "stderr_tail": p.stderr[-1500:]}
            except Exception as e:  # noqa: BLE001 (a time-out included)
                leg = {"error": repr(e)}
            leg["wall_s"] = time.perf_counter() - t_leg
            c4 = extra.get("config4") or {}
            one = c4.get("one_gpu_same_total") or {}
            if one.get("msm_ms") and isinstance(leg.get("msm"), dict) and leg["msm"].get("ms_per_msm"):
                leg["msm"]["speedup_vs_one_gpu"] = one["msm_ms"] / leg["msm"]["ms_per_msm"]
            extra["config4_in_process"] = leg
        fence()

    done.set()
    if rank == 0:
        emit()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def one_gpu_reference(zkp, torch, device, log_n, expand):
    """The MSM of 2^log_n terms and the Fr NTT of 2^log_n elements on THIS GPU alone: the denominators of the strong-scaling
    speedups of extra.config4.  Same library entries as the N = 1 grid (msm_g1_dev over the SRS expanded at the automatic width,
    ntt_fr_dev), inputs resident; the MSM result is checked against the trapdoor answer."""
    res = {"total_log_n": log_n}
    try:
        t0 = time.perf_counter()
        g = MsmWorkload(zkp, torch, device, log_n, chunk=0, expand="auto" if expand else 0)
        res["msm_setup_s"] = time.perf_counter() - t0
        reps = 2 if log_n >= 24 else 4
        el, r1, _ = time_msm(zkp, torch, lambda: zkp.msm_g1_dev(g.bases, g.scalars, g.n), reps, 1, torch.cuda.synchronize)
        res["msm_ms"] = el / reps * 1e3
        res["msm_bit_exact_full"] = check_against_trapdoor(zkp, g.limb_sums(), r1)
        res["msm_window_bits"] = g.window_bits
        g.close()
        del g
    except Exception as e:  # noqa: BLE001
        res["msm_error"] = repr(e)
    torch.cuda.empty_cache()
    try:
        m = 1 << log_n
        data = rand_fr_tensor(torch, m, 0x01770000 + log_n * 64, device).reshape(-1)
        ref = data.clone()
        zkp.ntt_fr_dev(data, log_n)
        zkp.ntt_fr_dev(data, log_n, inverse=True)
        torch.cuda.synchronize()
        res["ntt_roundtrip_identity"] = bool(torch.equal(data, ref))
        del ref
        reps = 4 if log_n >= 25 else 10
        for inverse, key in ((False, "ntt_forward_ms"), (True, "ntt_inverse_ms")):
            best = None
            for _trial in range(2):
                t1 = time.perf_counter()
                for _ in range(reps):
                    zkp.ntt_fr_dev(data, log_n, inverse=inverse)
                torch.cuda.synchronize()
                d1 = (time.perf_counter() - t1) / reps
                best = d1 if best is None or d1 < best else best
            res[key] = best * 1e3
        del data
    except Exception as e:  # noqa: BLE001
        res["ntt_error"] = repr(e)
    torch.cuda.empty_cache()
    return res


def bench_four_step(zkp, zdist, torch, dist, device, log_n, rank, world, fence, reduce_max):
    """Four-step Fr NTT of 2^log_n elements over `world` ranks (rank g owns the g-th contiguous slab), forward then inverse;
    per-phase milliseconds of the forward transform (max over ranks), round trip checked against the input."""
    slab = (1 << log_n) // world
    local = rand_fr_tensor(torch, slab, 0x01770000 + log_n * 64 + rank, device)
    ref = local.clone()
    ops = zdist.TorchOps(zkp)
    y = zdist.ntt_fr_distributed(local, log_n, False, ops=ops)          # warm-up (tables, RCCL channels)
    back = zdist.ntt_fr_distributed(y, log_n, True, ops=ops, input_layout="k1slab")   # the mirrored inverse reads that layout
    torch.cuda.synchronize()
    ok = bool(torch.equal(back.reshape(-1), ref.reshape(-1)))
    reps = 3
    phase_tot, phase_inv = {}, {}
    fence()
    t0 = time.perf_counter()
    for _ in range(reps):
        ph = {}
        y = zdist.ntt_fr_distributed(local, log_n, False, ops=ops, timings=ph)
        torch.cuda.synchronize()
        for k, v in zdist.resolve_timings(ph).items():
            phase_tot[k] = phase_tot.get(k, 0.0) + v
    fence()
    fwd = reduce_max((time.perf_counter() - t0) / reps)
    t0 = time.perf_counter()
    for _ in range(reps):
        ph = {}
        back = zdist.ntt_fr_distributed(y, log_n, True, ops=ops, input_layout="k1slab", timings=ph)
        torch.cuda.synchronize()
        for k, v in zdist.resolve_timings(ph).items():
            phase_inv[k] = phase_inv.get(k, 0.0) + v
    fence()
    inv = reduce_max((time.perf_counter() - t0) / reps)
    # the one-exchange form: a prover that keeps its vectors in the columns layout (zkp_hip/dist.py) skips the pack copy and the first
    # all-to-all; same kernels, same k1-slab layout in the middle.  The share is this rank's own random data (any data is a valid share).
    # (No try / except around this block: it is made of collectives, and a rank that left it early would pair its next collective
    # with a peer's all-to-all.  A failure propagates to the caller, which records it for the whole four-step entry -- ADVICE r4.)
    C1 = zdist.columns_chunks(log_n, world, 4)
    one = {}
    if True:
        share = local  # [N/G, 4]: read as [C][N1][cw]
        y1 = zdist.ntt_fr_distributed(share, log_n, False, ops=ops, chunks=C1, input_layout="columns")
        b1 = zdist.ntt_fr_distributed(y1, log_n, True, ops=ops, chunks=C1, input_layout="k1slab", output_layout="columns")
        torch.cuda.synchronize()
        ok1 = bool(torch.equal(b1.reshape(-1), ref.reshape(-1)))
        ph_f, ph_i = {}, {}
        fence()
        t0 = time.perf_counter()
        for _ in range(reps):
            ph = {}
            y1 = zdist.ntt_fr_distributed(share, log_n, False, ops=ops, chunks=C1, input_layout="columns", timings=ph)
            torch.cuda.synchronize()
            for k, v in zdist.resolve_timings(ph).items():
                ph_f[k] = ph_f.get(k, 0.0) + v
        fence()
        fwd1 = reduce_max((time.perf_counter() - t0) / reps)
        t0 = time.perf_counter()
        for _ in range(reps):
            ph = {}
            b1 = zdist.ntt_fr_distributed(y1, log_n, True, ops=ops, chunks=C1, input_layout="k1slab", output_layout="columns", timings=ph)
            torch.cuda.synchronize()
            for k, v in zdist.resolve_timings(ph).items():
                ph_i[k] = ph_i.get(k, 0.0) + v
        fence()
        inv1 = reduce_max((time.perf_counter() - t0) / reps)
        f1 = torch.tensor([1 if ok1 else 0], dtype=torch.int64, device="cpu" if dist.get_backend() == "gloo" else device)
        dist.all_reduce(f1, op=dist.ReduceOp.MIN)
        one = {"workload": "the same transform from / to the columns layout (rank g holds columns [g r2, (g + 1) r2) of the N1 x N2 matrix, "
                           f"{C1} column chunks): ONE all-to-all per direction, three kernel passes, no pack copy",
               "forward_ms": fwd1 * 1e3, "inverse_ms": inv1 * 1e3, "roundtrip_identity_all_ranks": bool(f1.item() == 1),
               "phase_ms_forward": {k: reduce_max(v / reps) for k, v in sorted(ph_f.items())},
               "phase_ms_inverse": {k: reduce_max(v / reps) for k, v in sorted(ph_i.items())}}
        del y1, b1
    flags = torch.tensor([1 if ok else 0], dtype=torch.int64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(flags, op=dist.ReduceOp.MIN)
    phases = {k: reduce_max(v / reps) for k, v in sorted(phase_tot.items())}
    phases_inv = {k: reduce_max(v / reps) for k, v in sorted(phase_inv.items())}
    n = 1 << log_n
    return {"workload": f"four-step Fr NTT, 2^{log_n} elements over {world} GPUs: natural slabs -> k1-slab layout (forward) and back "
                        "(mirrored inverse); per direction two RCCL all-to-alls (pipelined in column chunks against the column "
                        "transforms), four kernel passes, one pack copy",
            "forward_ms": fwd * 1e3, "inverse_ms": inv * 1e3, "elems_per_s_forward": n / fwd,
            "roundtrip_identity_all_ranks": bool(flags.item() == 1), "phase_ms_forward": phases, "phase_ms_inverse": phases_inv,
            "hbm_algorithmic_frac_per_gpu": NTT_BYTES_PER_ELEM * n / world / fwd / 1e9 / HBM_PEAK_GBS,
            "one_exchange": one}


if __name__ == "__main__":
    try:
        main()
    except BaseException as e:  # noqa: BLE001
        # after the headline is known rank 0 still owes the driver its ONE line: print it with the extras finished so far and say
        # what happened (a failure before that point has no line to print and surfaces as it is)
        if _EMIT is None or (isinstance(e, SystemExit) and not e.code):
            raise
        import traceback
        traceback.print_exc()
        _EMIT(f"failed after the headline: {e!r}")
        sys.stdout.flush()
        os._exit(1)  # not through destroy_process_group: peers may be inside a collective
