"""CPU-only checks of the gfx950 code the build produced (hipcc cross-compiles here; llvm-objdump reads the code object back).

1. No kernel may hold the scalar-memory / long-branch register race of profiles/r05_l_profile_mode_fault.md: LLVM's branch relaxation reusing,
   for a long jump, an SGPR pair that a scalar-memory instruction (s_load, s_memtime, s_memrealtime) is still writing.  Round 5 found one in
   msm_accumulate_kernel's clock-stamp epilogue (profiling on only): an intermittent GPU memory fault.  tools/check_smem_long_branch.py.
2. The checker itself recognises the pattern (a synthetic listing), so that a silent pass means something.
3. The hot kernels neither spill nor use scratch (profiles/r05_regs.md)."""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "zkp-implementation_amd", "libzkp_hip.so")


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def built():
    load(os.path.join(ROOT, "zkp-implementation_amd", "build.py"), "zkp_build").build()
    assert os.path.exists(LIB)
    return LIB


def test_no_scalar_memory_result_in_flight_into_a_long_branch(built):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_smem_long_branch.py"), built], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "long-branch sequences, 0 with" in r.stdout
    # the library does contain long branches (msm_accumulate is ~250 KB of code): the scan looked at something
    n = int(r.stdout.split(" long-branch sequences")[0].split()[-1])
    assert n >= 10, r.stdout


def test_checker_recognises_the_pattern():
    chk = load(os.path.join(ROOT, "tools", "check_smem_long_branch.py"), "chk")
    bad = """
0000000000001000 <kernel_a>:
\ts_memtime s[4:5]                                           // 000000001000: C0900100 00000000
\ts_memrealtime s[0:1]                                       // 000000001008: C0940000 00000000
\tv_cmp_eq_u32_e32 vcc, 0, v0                                // 000000001010: 7D940080
\ts_and_saveexec_b64 s[2:3], vcc                             // 000000001014: BE82206A
\ts_cbranch_execnz 6                                         // 000000001018: BF890006
\ts_getpc_b64 s[0:1]                                         // 00000000101C: BE801C00
\ts_add_u32 s0, s0, 0xfffd40b8                               // 000000001020: 8000FF00 FFFD40B8
\ts_addc_u32 s1, s1, -1                                      // 000000001028: 8201FF01 FFFFFFFF
\ts_setpc_b64 s[0:1]                                         // 000000001030: BE801D00
"""
    hits, kernels, lb = chk.scan(bad)
    assert kernels == 1 and lb == 1 and len(hits) == 1 and "s_memrealtime s[0:1]" in hits[0]
    good = bad.replace("\tv_cmp_eq_u32_e32 vcc, 0, v0 ", "\ts_waitcnt lgkmcnt(0)\n\tv_cmp_eq_u32_e32 vcc, 0, v0 ")
    hits, _, lb = chk.scan(good)
    assert lb == 1 and not hits
    other_pair = bad.replace("s_getpc_b64 s[0:1]", "s_getpc_b64 s[6:7]")
    assert not chk.scan(other_pair)[0]
    wide = bad.replace("s_memrealtime s[0:1]", "s_load_dwordx4 s[0:3], s[8:9], 0x10")
    assert len(chk.scan(wide)[0]) == 1


def test_hot_kernels_do_not_spill(built):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_regs.py"), built], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    seen = 0
    for line in r.stdout.splitlines():
        if any(k in line for k in ("msm_accumulate_kernel", "msm_pyramid_kernel", "ntt_pass_strided", "ntt_pass_last", "msm_accumulate_quad_kernel")):
            f = line.split()
            spill, scratch = int(f[f.index("spill") + 1]), int(f[f.index("scratch") + 1])
            assert spill == 0 and scratch == 0, line
            seen += 1
    assert seen >= 6
