"""rust/zkp-hip-sys/src/lib.rs (source-only, never compiled here) must declare exactly the functions of include/zkp_hip.h,
with the same number of parameters each -- the reference-side binding cannot drift from the C ABI unnoticed."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header():
    src = open(os.path.join(ROOT, "include", "zkp_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for name, args in re.findall(r"\b(zkp_[a-z0-9_]+)\s*\(([^;{]*)\)\s*;", src):
        args = args.strip()
        out[name] = 0 if args in ("", "void") else len(args.split(","))
    return out


def _rust():
    src = open(os.path.join(ROOT, "rust", "zkp-hip-sys", "src", "lib.rs")).read()
    out = {}
    for name, args in re.findall(r"pub fn (zkp_[a-z0-9_]+)\(([^)]*)\)", src):
        out[name] = 0 if not args.strip() else len(args.split(","))
    return out


def test_rust_sys_declares_the_header():
    h, r = _header(), _rust()
    assert len(h) >= 50
    assert set(h) == set(r), (sorted(set(h) - set(r)), sorted(set(r) - set(h)))
    assert h == r
    text = open(os.path.join(ROOT, "rust", "README.md")).read()
    assert "NOT compiled" in text  # the label stays
