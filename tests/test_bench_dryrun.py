"""bench.py's OWN rank logic at world 8, on the CPU: eight real processes under torch.distributed.run over gloo, the kernels stubbed
by tests/bench_dryrun_backend.py (oracle + the CPU statements of the axis-0 / layout transforms).  The driver's 8-GPU run is blind --
no box of this pool holds eight processes on its card, the rehearsal on the GPU stops at four ranks -- so everything around the
kernels is executed here: shard ranges and per-chunk seeds, the all-gather of the 192-byte partials and the EC-add combine, the
trapdoor check over every rank's limb sums, the four-step transform with both exchange forms, `one_gpu_reference` next to rank 0's
shard while seven ranks wait, the strong-scaling arithmetic of --total-log-n, the JSON merge and the emit-once logic."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dry(extra_args, world=8):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(ZKP_BENCH_DRYRUN=os.path.join(ROOT, "tests", "bench_dryrun_backend.py"), ZKP_BENCH_REHEARSAL="1", OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
                        "--no-in-process-leg"] + extra_args, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_rank_logic_world8_weak_scaling_line():
    d = _dry(["--log-n", "8", "--config4-log-n", "15"])
    assert d["dry_run"] is True and d["value"] is None and d["metric"].startswith("DRY RUN")   # never mistaken for a measurement
    assert d["n_gpus"] == 8 and d["config"]["world_size"] == 8 and d["scaling"] == "weak" and d["steps"] == 2 and d["warmup"] == 1
    assert d["config"]["total_terms"] == 8 << 8 and d["config"]["log_n_per_gpu"] == 8
    assert d["bit_exact_full"] is True          # eight chunk MSMs + all-gather + EC add == (sum over all ranks' s_i k_i) G
    assert "cpu_baseline" not in d and "extras_cut_short" not in d["extra"]
    c4 = d["extra"]["config4"]
    assert c4["total_log_n"] == 15 and c4["log_n_per_gpu"] == 12 and c4["rccl_world_size"] == 8
    assert c4["msm"]["bit_exact_full"] is True and c4["msm"]["one_gpu_ms"] > 0 and c4["msm"]["speedup_vs_one_gpu"] > 0
    f4 = c4["ntt_fr_four_step"]
    assert f4["roundtrip_identity_all_ranks"] is True and f4["speedup_vs_one_gpu"] > 0 and f4["speedup_vs_one_gpu_inverse"] > 0
    assert f4["one_exchange"]["roundtrip_identity_all_ranks"] is True and f4["one_exchange"]["speedup_vs_one_gpu"] > 0
    one = c4["one_gpu_same_total"]
    assert one["total_log_n"] == 15 and one["msm_bit_exact_full"] is True and one["ntt_roundtrip_identity"] is True
    assert 0 < c4["one_gpu_memory_estimate_gb"] < 288
    assert d["extra"]["sharded_grid"]["2^15"] == c4


def test_bench_rank_logic_world8_strong_scaling_total_log_n():
    """--gpus 8 --total-log-n T (BASELINE configs[4] as written, T = 26 there): 2^T / 8 terms per rank, the headline IS the sharded
    MSM of the whole problem, and the one-GPU time of the same total sits in the same line."""
    d = _dry(["--total-log-n", "15", "--config4-log-n", "15"])
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["bit_exact_full"] is True
    assert d["config"]["log_n_per_gpu"] == 12 and d["config"]["total_terms"] == 1 << 15 and "configs[4]" in d["config"]["workload"]
    same = d["extra"]["one_gpu_same_total"]
    assert same["total_log_n"] == 15 and same["same_result"] is True and same["speedup_of_this_run"] > 0
    c4 = d["extra"]["config4"]
    assert c4["msm"]["see"] and c4["msm"]["one_gpu_ms"] > 0 and c4["msm"]["speedup_vs_one_gpu"] > 0
    assert c4["ntt_fr_four_step"]["roundtrip_identity_all_ranks"] is True


def test_one_gpu_reference_fits_one_mi355x_at_configs4_size():
    """Rank 0 of the real 8-GPU run holds the whole 2^26 problem on its GPU for `one_gpu_reference` while its own shard's library
    workspaces are still allocated: the estimate bench.py prints into the line must stay well inside 288 GB."""
    sys.path.insert(0, ROOT)
    import bench
    gb = bench.one_gpu_reference_bytes(26) / 1e9
    print(f"one_gpu_reference at 2^26: {gb:.1f} GB of 288 GB")
    assert 100 < gb < 200       # the expanded SRS alone is 12 x 2^26 x 128 B = 103 GB
    assert bench.one_gpu_reference_bytes(27) / 1e9 > 200   # ... and 2^27 is where a single GPU stops (ZKP_E_NOMEM, tested on the GPU)
