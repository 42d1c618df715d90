#!/usr/bin/env python3
"""Regenerate rust/zkp-hip-sys/src/lib.rs (source-only crate, never compiled here) from include/zkp_hip.h, so that the
reference-side binding cannot drift from the C ABI.  tests/test_rust_sys_matches_header.py checks the result."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRIM = {"uint64_t": "u64", "uint8_t": "u8", "int": "i32", "unsigned": "u32", "size_t": "usize", "void": "c_void", "char": "c_char",
        "double": "f64", "zkp_bases": "zkp_bases", "zkp_plonk_prover": "zkp_plonk_prover",
        "zkp_plonk_transcript": "zkp_plonk_transcript", "zkp_plonk_proof": "zkp_plonk_proof", "zkp_ntt_layout": "zkp_ntt_layout",
        "zkp_ntt_shard_geometry": "zkp_ntt_shard_geometry"}
RET = {"int": "i32", "void": "()", "size_t": "usize", "const char *": "*const c_char", "const char*": "*const c_char"}


def ptype(t, arr):
    t = t.strip().replace("*const", "*")
    const = t.startswith("const ")
    base = t.replace("const ", "").strip()
    stars = base.count("*") + (1 if "[" in arr else 0)
    prim = PRIM[base.replace("*", "").strip()]
    for _ in range(stars):
        prim = ("*const " if const else "*mut ") + prim
    return prim


def main():
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "zkp_hip.h")).read(), flags=re.S)
    protos = re.findall(r"\n((?:const\s+)?[a-z_0-9]+\s*\*?\s*\*?\s*zkp_[a-z0-9_]+\s*\([^;{]*\))\s*;", src)
    out = []
    for p in protos:
        m = re.match(r"(.*?)(zkp_[a-z0-9_]+)\s*\((.*)\)$", " ".join(p.split()))
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        al = []
        if args != "void":
            for a in args.split(","):
                mm = re.match(r"(.*?)([A-Za-z_][A-Za-z0-9_]*)(\s*(?:\[[0-9]*\])*)$", a.strip())
                nm = {"type": "ty", "in": "input", "ref": "r"}.get(mm.group(2), mm.group(2))
                al.append(f"{nm}: {ptype(mm.group(1), mm.group(3))}")
        r = RET[ret]
        out.append(f"    pub fn {name}({', '.join(al)})" + ("" if r == "()" else f" -> {r}") + ";")
    lib = '''//! zkp-hip-sys -- `extern "C"` declarations for libzkp_hip.so, one to one with include/zkp_hip.h.
//!
//! SOURCE-ONLY: this crate has not been compiled (no Rust toolchain in the repository's build environment); the
//! declarations below are generated from the header by tools/gen_rust_sys.py and checked against it by
//! tests/test_rust_sys_matches_header.py, and every symbol is checked against the built library by tests/test_abi_cpu.py.
//! Each function's contract, and the reference file:line it replaces, is documented in include/zkp_hip.h.
#![allow(non_camel_case_types)]
use core::ffi::{c_char, c_void};

pub const ZKP_OK: i32 = 0;
pub const ZKP_E_ARG: i32 = -1;
pub const ZKP_E_NOMEM: i32 = -2;
pub const ZKP_E_DEVICE: i32 = -3;
pub const ZKP_E_SIZE: i32 = -4;

#[repr(C)] pub struct zkp_bases { _private: [u8; 0] }
#[repr(C)] pub struct zkp_plonk_prover { _private: [u8; 0] }
#[repr(C)] pub struct zkp_plonk_transcript { _private: [u8; 0] }

/// struct Proof of plonk/src/prover.rs:23-41 in ABI form
#[repr(C)]
pub struct zkp_plonk_proof {
    pub commit_xy: [[u64; 12]; 9], // a, b, c, z, t_lo, t_mid, t_hi, w_ev_x, w_ev_wx
    pub commit_is_inf: [u8; 9],
    pub bars: [[u64; 4]; 6],       // bar_a, bar_b, bar_c, bar_s_sigma_1, bar_s_sigma_2, bar_z_w
    pub u: [u64; 4],
    pub degree: u64,
}

/// gathered / scattered transform layout of zkp_ntt_fr_layout_dev (strides in elements)
#[repr(C)]
pub struct zkp_ntt_layout {
    pub lo_bits: u32,
    pub mid_bits: u32,
    pub mid_stride: usize,
    pub hi_stride: usize,
    pub batch_stride: usize,
}

/// split of the in-process multi-GPU transform (zkp_ntt_fr_sharded_geometry)
#[repr(C)]
pub struct zkp_ntt_shard_geometry {
    pub slots: u32,
    pub log_n1: u32,
    pub log_n2: u32,
    pub chunks: u32,
    pub r1: usize,
    pub r2: usize,
    pub cw: usize,
    pub slab: usize,
}
pub const ZKP_NTT_NATURAL: i32 = 0;
pub const ZKP_NTT_K1SLAB: i32 = 1;
pub const ZKP_NTT_COLUMNS: i32 = 2;

extern "C" {
''' + "\n".join(out) + '''
}

/// The thread-local message of the last failed call.
pub fn last_error() -> String {
    unsafe { std::ffi::CStr::from_ptr(zkp_last_error()).to_string_lossy().into_owned() }
}
'''
    open(os.path.join(ROOT, "rust", "zkp-hip-sys", "src", "lib.rs"), "w").write(lib)
    print(len(out), "functions")


if __name__ == "__main__":
    main()
