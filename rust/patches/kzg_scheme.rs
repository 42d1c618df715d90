// SOURCE-ONLY (not compiled here, see ../README.md): the replacement for kzg/src/scheme.rs:22-47 and 84-96.
// Everything else in scheme.rs (commit, commit_vector, open, open_vector, commit_para, verify, batch_verify) is unchanged: they
// all funnel into evaluate_in_s.
use ark_ec::AffineRepr;
use ark_ff::{BigInt, Fp};
use ark_poly::Polynomial;
use zkp_hip_sys as sys;

use crate::srs::Srs;
use crate::types::{G1Point, Poly};

/// The SRS `Vec<G1Affine>` resident in HBM, uploaded ONCE (the reference clones the vector on every commit, srs.rs:78-80).
/// With `zkp_init_devices` and more than one device the upload shards the points over the GPUs by contiguous chunk and every
/// later MSM runs on all of them (include/zkp_hip.h, "Device slots").
pub struct GpuBases {
    ptr: *mut sys::zkp_bases,
}
unsafe impl Send for GpuBases {} // handles are immutable after creation and may be shared across threads (zkp_hip.h)
unsafe impl Sync for GpuBases {}

impl GpuBases {
    pub fn upload(points: &[G1Point]) -> Self {
        // G1Affine { x, y, infinity } is repr(Rust): repack, never transmute
        let mut xy = Vec::with_capacity(points.len() * 12);
        let mut inf = Vec::with_capacity(points.len());
        for p in points {
            xy.extend_from_slice(&p.x.0 .0); // Montgomery residue limbs, as in memory
            xy.extend_from_slice(&p.y.0 .0);
            inf.push(p.infinity as u8);
        }
        let mut ptr = core::ptr::null_mut();
        let rc = unsafe { sys::zkp_g1_bases_create(xy.as_ptr(), inf.as_ptr(), points.len(), &mut ptr) };
        assert_eq!(rc, sys::ZKP_OK, "{}", sys::last_error());
        // the SRS is fixed for the life of the scheme: pay the one-off expansion (shared bucket set, ~1.4x faster commits)
        let rc = unsafe { sys::zkp_g1_bases_precompute(ptr, 0) };
        assert_eq!(rc, sys::ZKP_OK, "{}", sys::last_error());
        Self { ptr }
    }
    pub fn len(&self) -> usize {
        unsafe { sys::zkp_g1_bases_len(self.ptr) }
    }
}
impl Drop for GpuBases {
    fn drop(&mut self) {
        unsafe { sys::zkp_g1_bases_destroy(self.ptr) }
    }
}

pub struct KzgScheme(pub(crate) Srs, pub(crate) GpuBases);

impl KzgScheme {
    /// scheme.rs:34
    pub fn new(srs: Srs) -> Self {
        let bases = GpuBases::upload(&srs.g1_points());
        Self(srs, bases)
    }

    /// scheme.rs:84-96 -- the one seam: sum_i c_i [s^i]G_1 as ONE Pippenger MSM on the GPU(s) instead of n double-and-add
    /// scalar multiplications and 2n inversions.
    pub(crate) fn evaluate_in_s(&self, polynomial: &Poly) -> G1Point {
        assert!(self.1.len() > polynomial.degree()); // unchanged (scheme.rs:86)
        let c = &polynomial.coeffs; // &[Fr]: BigInt<4> Montgomery residues, contiguous
        let (mut xy, mut inf) = ([0u64; 12], 0u8);
        let rc = unsafe { sys::zkp_msm_g1(self.1.ptr, c.as_ptr() as *const u64, c.len(), xy.as_mut_ptr(), &mut inf) };
        assert_eq!(rc, sys::ZKP_OK, "{}", sys::last_error());
        if inf != 0 {
            return G1Point::zero(); // empty polynomial -> identity (scheme.rs:94)
        }
        let limb = |o: usize| BigInt::<6>([xy[o], xy[o + 1], xy[o + 2], xy[o + 3], xy[o + 4], xy[o + 5]]);
        // the limbs ARE the Montgomery residues: construct the field elements without another conversion
        G1Point::new_unchecked(Fp::new_unchecked(limb(0)), Fp::new_unchecked(limb(6)))
    }
}
