#!/usr/bin/env python3
"""Render the SURVEY 8d reporting grid (1-GPU column) from a bench.py JSON line:  python tools/render_grid.py bench.json > grid.md"""
import json
import sys

d = json.load(open(sys.argv[1]))
ex = d["extra"]
cpu = d.get("cpu_baseline", {})
cpu_naive = cpu.get("value")
cpu_pip = cpu.get("context", {}).get("pippenger_1core_scalar_muls_per_s")
cpu_ntt = cpu.get("context", {}).get("ntt_fr_1core_elems_per_s")
print(f"# 1-GPU reporting grid (SURVEY 8d) rendered from `{sys.argv[1]}`\n")
print("MSM: expanded SRS (automatic width), scalars and bases resident; algorithmic bytes 128 B per scalar-mul against 8 TB/s; "
      "`accumulate frac` prices the dominant kernel alone, `whole` the complete MSM.  CPU (i) = the oracle's reference-faithful naive "
      f"MSM ({cpu_naive:.0f} scalar-muls/s on 1 core, `{cpu.get('kind')}`), CPU (ii) = the oracle's bucket method on 1 core "
      f"({cpu_pip:.0f}/s).\n" if cpu_naive and cpu_pip else "")
print("| n | window bits | ms | scalar-muls/s | algorithmic GB/s (whole) | whole frac | accumulate frac | x CPU (i) | x CPU (ii) | bit-exact |")
print("|---|---|---|---|---|---|---|---|---|---|")
rows = [("2^%d" % d["config"]["log_n_per_gpu"], {"ms_per_msm": d["ms_per_step"], "scalar_muls_per_s": d["value"], "window_bits": d["config"]["window_bits"],
                                                 "roofline_frac": d["roofline"]["frac"], "bit_exact_full": d["bit_exact_full"]})]
rows += [(k, v) for k, v in ex.get("msm_grid", {}).items() if k != "workload" and "error" not in v]
for k, v in rows:
    rate = v["scalar_muls_per_s"]
    gbs = 128 * rate / 1e9
    print(f"| {k} | {v.get('window_bits')} | {v['ms_per_msm']:.2f} | {rate:.3e} | {gbs:.1f} | {gbs / 8000 * 100:.2f} % | "
          f"{v['roofline_frac'] * 100:.2f} % | {rate / cpu_naive:.2e} | {rate / cpu_pip:.1e} | {v['bit_exact_full']} |" if cpu_naive and cpu_pip else
          f"| {k} | {v.get('window_bits')} | {v['ms_per_msm']:.2f} | {rate:.3e} | {gbs:.1f} | {gbs / 8000 * 100:.2f} % | {v['roofline_frac'] * 100:.2f} % | | | {v['bit_exact_full']} |")
if "ntt_grid" in ex:
    print("\nFr NTT (one forward transform, natural order in and out, data resident; 64 B per element algorithmic). "
          + (f"CPU = the oracle's ark-poly style radix-2 transform on 1 core ({cpu_ntt:.3e} elem/s at 2^18).\n" if cpu_ntt else "\n"))
    print("| n | ms | elem/s | algorithmic GB/s | frac of 8 TB/s | x CPU | round trip exact |")
    print("|---|---|---|---|---|---|---|")
    for k, v in ex["ntt_grid"].items():
        if k == "workload" or "error" in v:
            continue
        print(f"| {k} | {v['forward_ms']:.3f} | {v['elems_per_s']:.3e} | {v['hbm_algorithmic_GBs']:.0f} | {v['hbm_frac'] * 100:.2f} % | "
              f"{(v['elems_per_s'] / cpu_ntt if cpu_ntt else 0):.0f} | {v['roundtrip_identity']} |")
print("\nThe 2 / 4 / 8-GPU columns come from `bench.py --gpus N` (`extra.sharded_grid`: total sizes 2^22 / 2^24 / 2^26 sharded over the ranks) "
      "on a multi-GPU node; the boxes of this pool have one GPU.")
