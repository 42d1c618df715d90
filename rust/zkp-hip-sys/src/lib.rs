//! zkp-hip-sys -- `extern "C"` declarations for libzkp_hip.so, one to one with include/zkp_hip.h.
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
    pub fn zkp_init(device: i32) -> i32;
    pub fn zkp_init_devices(devices: *const i32, n_devices: i32) -> i32;
    pub fn zkp_device_count() -> i32;
    pub fn zkp_set_device(slot: i32) -> i32;
    pub fn zkp_shutdown();
    pub fn zkp_last_error() -> *const c_char;
    pub fn zkp_abi_version() -> i32;
    pub fn zkp_profile_enable(on: i32);
    pub fn zkp_profile_reset();
    pub fn zkp_profile_read(name: *const c_char, total_ms: *mut f64, count: *mut u64) -> i32;
    pub fn zkp_profile_clock_read(name: *const c_char, cycles: *mut u64, ref_ticks: *mut u64, waves: *mut u64) -> i32;
    pub fn zkp_probe_mad_rate(launches: u32, lane_mads_per_s: *mut f64, clock_mhz: *mut f64, ms_per_launch: *mut f64) -> i32;
    pub fn zkp_g1_bases_create(xy: *const u64, is_inf: *const u8, n: usize, out: *mut *mut zkp_bases) -> i32;
    pub fn zkp_g1_bases_create_dev(d_xy: *const c_void, d_is_inf: *const u8, n: usize, stream: *mut c_void, out: *mut *mut zkp_bases) -> i32;
    pub fn zkp_g1_bases_precompute(b: *mut zkp_bases, window_bits: u32) -> i32;
    pub fn zkp_g1_bases_len(b: *const zkp_bases) -> usize;
    pub fn zkp_g1_bases_info(b: *const zkp_bases, window_bits: *mut u32, slices: *mut u32) -> i32;
    pub fn zkp_g1_bases_destroy(b: *mut zkp_bases);
    pub fn zkp_msm_g1(bases: *const zkp_bases, scalars: *const u64, n: usize, out_xy: *mut u64, out_is_inf: *mut u8) -> i32;
    pub fn zkp_msm_g1_dev(bases: *const zkp_bases, d_scalars: *const c_void, n: usize, stream: *mut c_void, out_xy: *mut u64, out_is_inf: *mut u8) -> i32;
    pub fn zkp_msm_g1_batch_dev(bases: *const zkp_bases, d_scalars: *const *const c_void, count: usize, n: usize, stream: *mut c_void, out_xy: *mut u64, out_is_inf: *mut u8) -> i32;
    pub fn zkp_msm_g1_partial_dev(bases: *const zkp_bases, d_scalars: *const c_void, n: usize, stream: *mut c_void, out_xyzz: *mut u64) -> i32;
    pub fn zkp_msm_g1_partial(bases: *const zkp_bases, scalars: *const u64, n: usize, out_xyzz: *mut u64) -> i32;
    pub fn zkp_g1_bases_shard_count(b: *const zkp_bases) -> i32;
    pub fn zkp_g1_bases_shard(b: *const zkp_bases, i: usize, slot: *mut i32, device: *mut i32, offset: *mut usize, len: *mut usize) -> i32;
    pub fn zkp_msm_g1_sharded_dev(bases: *const zkp_bases, d_scalars: *const *const c_void, n: usize, out_xy: *mut u64, out_is_inf: *mut u8) -> i32;
    pub fn zkp_msm_g1_sharded_dev_after(bases: *const zkp_bases, d_scalars: *const *const c_void, ready_events: *mut *mut c_void, n: usize, out_xy: *mut u64, out_is_inf: *mut u8) -> i32;
    pub fn zkp_g1_xyzz_sum(partials: *const u64, count: usize, out_xy: *mut u64, out_is_inf: *mut u8) -> i32;
    pub fn zkp_kzg_commit(srs: *const zkp_bases, coeffs: *const u64, len: usize, out_xy: *mut u64, out_is_inf: *mut u8) -> i32;
    pub fn zkp_kzg_open(srs: *const zkp_bases, coeffs: *const u64, len: usize, z: *const u64, out_xy: *mut u64, out_is_inf: *mut u8, out_eval: *mut u64) -> i32;
    pub fn zkp_g1_mul(base_xy: *const u64, base_is_inf: u8, scalar: *const u64, out_xy: *mut u64, out_is_inf: *mut u8) -> i32;
    pub fn zkp_g1_fixed_base_mul_dev(d_scalars: *const c_void, n: usize, d_out_xy: *mut c_void, d_out_is_inf: *mut u8, stream: *mut c_void) -> i32;
    pub fn zkp_selftest_fq_inverse_dev(d_in: *const c_void, n: usize, form: i32, d_out: *mut c_void, stream: *mut c_void) -> i32;
    pub fn zkp_srs_g1(secret: *const u64, n: usize, out_xy: *mut u64) -> i32;
    pub fn zkp_ntt_fr(data: *mut u64, log_n: u32, inverse: i32, coset: *const u64) -> i32;
    pub fn zkp_ntt_fr_dev(d_data: *mut c_void, log_n: u32, batch: usize, inverse: i32, coset: *const u64, stream: *mut c_void) -> i32;
    pub fn zkp_ntt_fr_twiddle_dev(d_data: *mut c_void, rows: usize, cols: usize, row0: usize, log_n: u32, inverse: i32, stream: *mut c_void) -> i32;
    pub fn zkp_ntt_fr_axis0_dev(d_in: *const c_void, d_out: *mut c_void, log_len: u32, cols: usize, inverse: i32, tw_log_n: u32, tw_col0: usize, stream: *mut c_void) -> i32;
    pub fn zkp_ntt_fr_layout_dev(d_in: *const c_void, d_out: *mut c_void, log_n: u32, batch: usize, inverse: i32, in_layout: *const zkp_ntt_layout, out_layout: *const zkp_ntt_layout, tw_log_n: u32, tw_row0: usize, stream: *mut c_void) -> i32;
    pub fn zkp_ntt_fr_sharded_geometry(log_n: u32, slots: u32, chunks: u32, out: *mut zkp_ntt_shard_geometry) -> i32;
    pub fn zkp_ntt_fr_sharded_dev(d_slabs: *mut *mut c_void, log_n: u32, inverse: i32, layout_in: i32, layout_out: i32, chunks: u32, streams: *mut *mut c_void) -> i32;
    pub fn zkp_ntt_fr_sharded(data: *mut u64, log_n: u32, inverse: i32, coset: *const u64) -> i32;
    pub fn zkp_ntt_goldilocks(data: *mut u64, log_n: u32, inverse: i32, coset: *const u64) -> i32;
    pub fn zkp_ntt_goldilocks_dev(d_data: *mut c_void, log_n: u32, batch: usize, inverse: i32, coset: *const u64, stream: *mut c_void) -> i32;
    pub fn zkp_fri_layer_eval(coeffs: *const u64, d: usize, coset: u64, log_D: u32, out: *mut u64) -> i32;
    pub fn zkp_fri_fold(coeffs: *const u64, d: usize, r: u64, out: *mut u64) -> i32;
    pub fn zkp_fri_merkle_node_count(n: usize) -> usize;
    pub fn zkp_fri_merkle_tree(leaves: *const u64, n: usize, nodes_out: *mut u64) -> i32;
    pub fn zkp_fri_merkle_tree_dev(d_leaves: *const c_void, n: usize, d_nodes: *mut c_void, stream: *mut c_void) -> i32;
    pub fn zkp_fri_challenges(roots: *const u64, layers: usize, const_val: u64, num_queries: usize, r_out: *mut u64, q_out: *mut u64) -> i32;
    pub fn zkp_fri_prove(coeffs: *const u64, d: usize, blowup_factor: usize, num_queries: usize, out_proof: *mut *mut u64, out_words: *mut usize) -> i32;
    pub fn zkp_fri_verify(proof: *const u64, words: usize) -> i32;
    pub fn zkp_free(p: *mut c_void);
    pub fn zkp_plonk_transcript_create(out: *mut *mut zkp_plonk_transcript) -> i32;
    pub fn zkp_plonk_transcript_destroy(t: *mut zkp_plonk_transcript);
    pub fn zkp_plonk_transcript_feed(t: *mut zkp_plonk_transcript, xy: *const u64, is_inf: u8) -> i32;
    pub fn zkp_plonk_transcript_challenges(t: *mut zkp_plonk_transcript, n: usize, out: *mut u64) -> i32;
    pub fn zkp_poly_mul_fr(a: *const u64, la: usize, b: *const u64, lb: usize, out: *mut u64) -> i32;
    pub fn zkp_plonk_prover_create(srs: *const zkp_bases, log_n: u32, polys: *const *const u64, lens: *const usize, k1: *const u64, k2: *const u64, out: *mut *mut zkp_plonk_prover) -> i32;
    pub fn zkp_plonk_prover_destroy(p: *mut zkp_plonk_prover);
    pub fn zkp_plonk_round1(p: *mut zkp_plonk_prover, blinders: *const u64, out_xy: *mut u64, out_is_inf: *mut u8) -> i32;
    pub fn zkp_plonk_round2(p: *mut zkp_plonk_prover, beta: *const u64, gamma: *const u64, blinders: *const u64, out_xy: *mut u64, out_is_inf: *mut u8) -> i32;
    pub fn zkp_plonk_round3(p: *mut zkp_plonk_prover, alpha: *const u64, out_xy: *mut u64, out_is_inf: *mut u8, out_degree: *mut usize) -> i32;
    pub fn zkp_plonk_round4(p: *mut zkp_plonk_prover, zeta: *const u64, out_bars: *mut u64) -> i32;
    pub fn zkp_plonk_round5(p: *mut zkp_plonk_prover, v: *const u64, out_xy: *mut u64, out_is_inf: *mut u8) -> i32;
    pub fn zkp_plonk_prove(p: *mut zkp_plonk_prover, blinders: *const u64, out: *mut zkp_plonk_proof) -> i32;
    pub fn zkp_g2_generator(out_xy: *mut u64) -> i32;
    pub fn zkp_g2_mul(q_xy: *const u64, q_is_inf: u8, scalar: *const u64, out_xy: *mut u64, out_is_inf: *mut u8) -> i32;
    pub fn zkp_pairing(p_xy: *const u64, p_is_inf: u8, q_xy: *const u64, q_is_inf: u8, out_fq12: *mut u64) -> i32;
    pub fn zkp_kzg_verify(g2s_xy: *const u64, commit_xy: *const u64, commit_is_inf: u8, w_xy: *const u64, w_is_inf: u8, y: *const u64, z: *const u64, accepted: *mut i32) -> i32;
    pub fn zkp_kzg_batch_verify(g2s_xy: *const u64, n: usize, commits_xy: *const u64, commits_is_inf: *const u8, points: *const u64, openings_xy: *const u64, openings_is_inf: *const u8, evals: *const u64, r_primes: *const u64, accepted: *mut i32) -> i32;
    pub fn zkp_kzg_aggregate_commitments(commits_xy: *const u64, commits_is_inf: *const u8, n: usize, challenge: *const u64, out_xy: *mut u64, out_is_inf: *mut u8) -> i32;
    pub fn zkp_plonk_verify(p: *mut zkp_plonk_prover, g2s_xy: *const u64, proof: *const zkp_plonk_proof, accepted: *mut i32) -> i32;
    pub fn zkp_plonk_get_poly(p: *mut zkp_plonk_prover, which: i32, out: *mut u64, cap_elems: usize, len: *mut usize) -> i32;
}

/// The thread-local message of the last failed call.
pub fn last_error() -> String {
    unsafe { std::ffi::CStr::from_ptr(zkp_last_error()).to_string_lossy().into_owned() }
}
