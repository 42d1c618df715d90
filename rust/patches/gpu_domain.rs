// SOURCE-ONLY (not compiled here, see ../README.md): the NTT call sites of the reference routed through the C ABI.
use ark_bls12_381::Fr;
use ark_ff::Zero;
use ark_poly::univariate::DensePolynomial;
use ark_poly::DenseUVPolynomial;
use zkp_hip_sys as sys;

/// Stand-in for `GeneralEvaluationDomain::<Fr>::new(n)` where the reference only interpolates or multiplies
/// (plonk/src/prover.rs:70,374-375,396-426,463; plonk/src/circuit.rs:170-176,202,230-232).
pub struct GpuDomain {
    pub log_n: u32,
}

impl GpuDomain {
    pub fn new(n: usize) -> Self {
        Self { log_n: n.next_power_of_two().trailing_zeros() }
    }
    pub fn size(&self) -> usize {
        1 << self.log_n
    }

    /// `Evaluations::from_vec_and_domain(v, domain).interpolate()`: inverse transform, natural order, scaled by 1/n; shorter
    /// vectors are zero-padded to the domain (ark-poly) and trailing zero coefficients trimmed by from_coefficients_vec.
    pub fn interpolate(&self, mut evals: Vec<Fr>) -> DensePolynomial<Fr> {
        evals.resize(self.size(), Fr::zero());
        let rc = unsafe { sys::zkp_ntt_fr(evals.as_mut_ptr() as *mut u64, self.log_n, 1, core::ptr::null()) };
        assert_eq!(rc, sys::ZKP_OK, "{}", sys::last_error());
        DensePolynomial::from_coefficients_vec(evals)
    }

    /// `fft` / `coset_fft` (coset = Some(g)): coefficients -> evaluations on g * <omega>
    pub fn evaluate(&self, mut coeffs: Vec<Fr>, coset: Option<Fr>) -> Vec<Fr> {
        coeffs.resize(self.size(), Fr::zero());
        let c = coset.as_ref().map_or(core::ptr::null(), |g| g as *const Fr as *const u64);
        let rc = unsafe { sys::zkp_ntt_fr(coeffs.as_mut_ptr() as *mut u64, self.log_n, 0, c) };
        assert_eq!(rc, sys::ZKP_OK, "{}", sys::last_error());
        coeffs
    }
}

/// A vector of 2^log_n field elements resident on the devices of the node, one slab per device slot (`zkp_init_devices`), for a
/// prover that keeps its polynomials in HBM between `fft`, pointwise work and `ifft` (plonk/src/prover.rs:396-443 at sizes where one
/// GPU is not enough).  `slabs[g]` is device memory of slot g holding `geometry().slab` elements in the layout the last call left.
pub struct ShardedVector {
    pub log_n: u32,
    pub slabs: Vec<*mut core::ffi::c_void>,
    pub chunks: u32,
}

impl ShardedVector {
    pub fn geometry(&self) -> sys::zkp_ntt_shard_geometry {
        let mut g = core::mem::MaybeUninit::<sys::zkp_ntt_shard_geometry>::zeroed();
        let rc = unsafe { sys::zkp_ntt_fr_sharded_geometry(self.log_n, 0, self.chunks, g.as_mut_ptr()) };
        assert_eq!(rc, sys::ZKP_OK, "{}", sys::last_error());
        unsafe { g.assume_init() }
    }
    fn run(&mut self, inverse: bool, from: i32, to: i32) {
        let rc = unsafe {
            sys::zkp_ntt_fr_sharded_dev(self.slabs.as_mut_ptr(), self.log_n, inverse as i32, from, to, self.chunks, core::ptr::null_mut())
        };
        assert_eq!(rc, sys::ZKP_OK, "{}", sys::last_error());
    }
    /// coefficients in natural slabs -> evaluations in the transposed (k1-slab) order: pointwise products do not care
    pub fn fft_to_k1slab(&mut self) { self.run(false, sys::ZKP_NTT_NATURAL, sys::ZKP_NTT_K1SLAB) }
    /// the mirror image: evaluations in k1-slab order -> coefficients in natural slabs
    pub fn ifft_from_k1slab(&mut self) { self.run(true, sys::ZKP_NTT_K1SLAB, sys::ZKP_NTT_NATURAL) }
    /// the same pair with ONE exchange each, for vectors kept in the columns layout
    pub fn fft_columns_to_k1slab(&mut self) { self.run(false, sys::ZKP_NTT_COLUMNS, sys::ZKP_NTT_K1SLAB) }
    pub fn ifft_k1slab_to_columns(&mut self) { self.run(true, sys::ZKP_NTT_K1SLAB, sys::ZKP_NTT_COLUMNS) }
}
// (GpuDomain::interpolate / evaluate above need no multi-GPU variant: with several device slots zkp_ntt_fr takes the sharded route by
// itself from 2^24 elements on -- include/zkp_hip.h, ZKP_NTT_SHARD_MIN_LOG.)

/// `&a * &b` for DensePolynomial<Fr> (plonk/src/prover.rs:396-426,437,516-548): FFT product on the radix-2 domain of size
/// next_pow2(len_a + len_b - 1); a zero operand gives the zero polynomial.
pub fn mul_gpu(a: &DensePolynomial<Fr>, b: &DensePolynomial<Fr>) -> DensePolynomial<Fr> {
    if a.coeffs.is_empty() || b.coeffs.is_empty() {
        return DensePolynomial::from_coefficients_vec(vec![]);
    }
    let mut out = vec![Fr::zero(); a.coeffs.len() + b.coeffs.len() - 1];
    let rc = unsafe {
        sys::zkp_poly_mul_fr(a.coeffs.as_ptr() as *const u64, a.coeffs.len(), b.coeffs.as_ptr() as *const u64, b.coeffs.len(),
                             out.as_mut_ptr() as *mut u64)
    };
    assert_eq!(rc, sys::ZKP_OK, "{}", sys::last_error());
    DensePolynomial::from_coefficients_vec(out)
}

/// The evaluation loop of `FriLayer::from_poly` (fri/src/fri_layer.rs:40-46) for F = Goldilocks (Fp64<MontBackend<_, 1>>):
/// evaluations[i] = poly(coset * omega_D^i), natural order.  `coeffs` / the result are the field's memory words.
pub fn fri_layer_evaluations(coeffs: &[u64], coset_word: u64, domain_size: usize) -> Vec<u64> {
    let mut out = vec![0u64; domain_size];
    let rc = unsafe { sys::zkp_fri_layer_eval(coeffs.as_ptr(), coeffs.len(), coset_word, domain_size.trailing_zeros(), out.as_mut_ptr()) };
    assert_eq!(rc, sys::ZKP_OK, "{}", sys::last_error());
    out
}
