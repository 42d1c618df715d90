"""CPU-only checks of the drop-in boundary: the library builds, loads and exports every symbol of include/zkp_hip.h.
No compute call is made here (there is no GPU in the build container)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "zkp_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(zkp_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def zkp():
    import importlib.util
    spec = importlib.util.spec_from_file_location("zkp_build", os.path.join(ROOT, "zkp-implementation_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build()
    import zkp_hip
    return zkp_hip


def test_library_exports_every_declared_symbol(zkp):
    lib = zkp.lib()
    names = header_functions()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/zkp_hip.h but not exported"
    # the binding covers exactly the header
    assert set(zkp.exported_symbols()) == set(names)
    assert lib.zkp_abi_version() == 1


def test_no_cpu_fallback_without_gpu(zkp):
    """Without a gfx950 device every compute entry must fail loudly, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np
    with pytest.raises(zkp.ZkpError) as ei:
        zkp.ntt_fr(np.zeros((4, 4), dtype=np.uint64))
    assert ei.value.code == zkp.ZKP_E_DEVICE
    with pytest.raises(zkp.ZkpError):
        zkp.G1Bases.from_host(np.zeros((1, 12), dtype=np.uint64))


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under zkp-implementation_amd/ may reference it."""
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "zkp-implementation_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".inc", ".cpp", ".h")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"zkp_oracle|libzkp_oracle|from oracle|import oracle|oracle/", txt):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_host_side_g1_mul_and_sum_without_gpu(zkp, golden):
    """zkp_g1_mul (KzgScheme::commit_para, kzg/src/scheme.rs:78-82) and zkp_g1_xyzz_sum are host code: exercised on the CPU
    against the golden scalar multiples of G."""
    import numpy as np
    from oracle import oracle as orc
    orc.build()
    g = orc.g1_generator()
    for ent in golden["g1_mul"]:
        k = orc.fr_from_ints([int(ent["k"], 16)])[0]
        out, inf = zkp.g1_mul(g, 0, k)
        if ent["out"] is None:
            assert inf
        else:
            assert not inf and orc.points_to_ints(out)[0] == (int(ent["out"][0], 16), int(ent["out"][1], 16))
    # infinity base, and summing G + 2G + identity = 3G
    out, inf = zkp.g1_mul(g, 1, orc.fr_from_ints([5])[0])
    assert inf
    one = orc.fq_from_ints([1])[0]
    two_g, _ = zkp.g1_mul(g, 0, orc.fr_from_ints([2])[0])
    parts = np.zeros((3, 24), dtype=np.uint64)
    parts[0, :12], parts[0, 12:18], parts[0, 18:] = g, one, one
    parts[1, :12], parts[1, 12:18], parts[1, 18:] = two_g, one, one
    s, sinf = zkp.g1_xyzz_sum(parts)
    three_g, _ = zkp.g1_mul(g, 0, orc.fr_from_ints([3])[0])
    assert not sinf and np.array_equal(s, three_g)


def test_device_slot_entries_without_gpu(zkp):
    """The multi-device entries validate their arguments and fail loudly without a gfx950 device (no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert zkp.device_count() == 0
    with pytest.raises(zkp.ZkpError) as ei:
        zkp.init_devices(None, 2)
    assert ei.value.code == zkp.ZKP_E_DEVICE
    with pytest.raises(zkp.ZkpError) as ei:
        zkp.init_devices([0, 0])
    assert ei.value.code in (zkp.ZKP_E_DEVICE, zkp.ZKP_E_ARG)
    with pytest.raises(zkp.ZkpError) as ei:
        zkp.set_device(-2)
    assert ei.value.code == zkp.ZKP_E_ARG
    zkp.set_device(-1)
    assert zkp.device_count() == 0


def test_loading_the_library_puts_torch_and_its_hip_runtime_first():
    """PyTorch's wheel bundles its own HIP runtime.  A process that loaded libzkp_hip.so (and with it /opt/rocm's runtime) BEFORE
    importing torch ended with two runtimes, and zkp_init saw no device (`build()` followed by `smoke()` in one process on the GPU
    box).  zkp_hip.lib() therefore imports torch first, wherever torch exists."""
    import subprocess
    import sys
    code = ("import sys; sys.path[:0] = [%r, %r]; import zkp_hip; assert 'torch' not in sys.modules; zkp_hip.lib(); "
            "assert 'torch' in sys.modules; print('ok')") % (ROOT, os.path.join(ROOT, "zkp-implementation_amd"))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "ok" in p.stdout, p.stderr[-2000:]
