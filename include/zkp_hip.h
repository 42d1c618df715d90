/*
 * zkp_hip.h -- C ABI of libzkp_hip.so, the MI355X (gfx950) backend for the MSM / NTT hot path of
 * sota-zk-labs/zkp-implementation.
 *
 * The reference has no FFI: the surfaces below are what a Rust `-sys` binding would call from inside the
 * reference's own functions (INTEGRATION.md shows the stubs).  Each entry cites the reference code it replaces
 * (file:line under the reference repository root).
 *
 * Data formats = arkworks 0.4 in-memory forms, so a Rust caller passes `&[Fr]` / coordinates untouched:
 *   Fr          uint64_t[4]   Montgomery residue (R = 2^256), little-endian limbs        kzg/src/types.rs:7
 *   Fq (base)   uint64_t[6]   Montgomery residue (R = 2^384)                             kzg/src/types.rs:6
 *   Goldilocks  uint64_t[1]   Montgomery residue (R = 2^64)                              fri/src/fields/goldilocks.rs:4-8
 *   G1 affine   uint64_t[12]  x || y; the point at infinity is carried in a separate byte (1 = infinity),
 *                             never as magic coordinates                                 kzg/src/types.rs:6
 *
 * Conventions: every function returns ZKP_OK (0) or a negative error code and never aborts or unwinds
 * across the ABI; zkp_last_error() gives a thread-local message.  Host buffers are owned by the caller for the
 * duration of the call.  `*_dev` variants take DEVICE pointers (hipMalloc'd or torch CUDA tensors) and a
 * hipStream_t passed as void* (NULL = the default stream); they enqueue work and return without synchronising
 * unless documented otherwise.  Calls may use different streams: workspaces and cached tables are per slot, and an entry
 * that arrives on another stream than the previous one first waits (on the device, hipStreamWaitEvent) for the work the
 * previous entry enqueued.  There is no CPU implementation of the hot path behind this ABI: if no gfx950 device is
 * usable, zkp_init() fails with ZKP_E_DEVICE and every MSM / NTT / Merkle / prover entry fails the same way.  The entries
 * marked "host" (transcripts, verifiers, pairings) are host code by nature and need no device.
 *
 * Environment (read by the library; none of them changes a result): ZKP_MSM_C (window bits of the per-window MSM over
 * unexpanded bases, 8..16), ZKP_MSM_RANGE_LOG (log2 of the scalar range of one pass of the shared-bucket MSM; default 24 over an SRS expanded into at most 12 planes -- the automatic 22-bit windows -- and 23 over more planes),
 * ZKP_MSM_NCHUNK (chunks of the counting sort), ZKP_SORT_LO_BITS (bins of its second pass, log2), ZKP_MSM_FEED_FIRST_PCT (zkp_msm_g1 uploads host scalars in two ranges, the first one this share of them, default 20; 0 = equal ranges), ZKP_MSM_FEED_RANGES (that many EQUAL ranges instead),
 * ZKP_MSM_SPLIT_LOG (0..2: log2 of the lanes that share a bucket's run in a small single-pass MSM; default: chosen per launch),
 * ZKP_MSM_NO_OVERLAP=1 (digits + sort of the next scalar range on the launch stream instead of a second one), ZKP_NTT_NO_WIDE_PASS=1 (Fr
 * transforms with radix <= 2^8 passes only), ZKP_NTT_TW_MATRIX_MAX_LOG (largest Fr transform whose first-pass twiddles are kept as a
 * 32-byte-per-element matrix, default 24, 0 = never) -- tuning and test aids; ZKP_FRI_ZERO_AS_0=1 prints the field element zero as "0"
 * instead of the empty string in the FRI hash input (the one third-party formatting detail that could not be confirmed offline);
 * ZKP_SRS_EXPAND_MAX_BYTES (zkp_g1_bases_precompute refuses, with ZKP_E_NOMEM and the sizes in zkp_last_error(), an expansion larger
 * than this many bytes -- it is also refused when it exceeds the device's free memory; the handle then stays usable unexpanded),
 * ZKP_MSM_BALANCE_FROM (overshoot in bits from which the slices of an expansion are balanced, default 1 = always; tuning aid),
 * ZKP_PYR_TAIL_THREADS / ZKP_PYR_TAIL_BLOCKS / ZKP_PYR_TAIL_HALF (geometry of the launch that runs the last levels of the bucket reduction:
 * workgroup size 64..512, workgroups per bucket set 1..256, pairs per array from which it takes over; defaults 256 / 16 / 64; tuning aids).
 */
#ifndef ZKP_HIP_H
#define ZKP_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZKP_OK 0
#define ZKP_E_ARG (-1)    /* bad argument (null pointer, log_n out of range, ...) */
#define ZKP_E_NOMEM (-2)  /* host or device allocation failed */
#define ZKP_E_DEVICE (-3) /* no usable gfx950 device / HIP runtime error */
#define ZKP_E_SIZE (-4)   /* more scalars than bases: the reference's assert at kzg/src/scheme.rs:86 */

typedef struct zkp_bases zkp_bases; /* opaque: G1 base points resident in HBM */

/* ---- library ----
 * Device slots.  The library keeps one context (workspaces, cached twiddle tables, a launch stream, one mutex) per device
 * SLOT; entries on different slots run concurrently, entries on one slot are serialised.
 *   zkp_init(device)             one slot on HIP device `device` (-1 = the current HIP device).  Idempotent.  This is also what
 *                                the first compute entry does by itself if nothing was initialised.
 *   zkp_init_devices(devs, n)    n slots, slot i on HIP device devs[i]; devs == NULL: devices 0..n-1; n == 0: every visible
 *                                device.  A device may be listed more than once (separate slots on one GPU: how a 1-GPU box
 *                                rehearses the multi-device entries).  With more than one slot:
 *                                  - zkp_g1_bases_create SHARDS the points by contiguous chunk, one chunk resident per slot,
 *                                    unless the calling thread chose a slot with zkp_set_device;
 *                                  - zkp_msm_g1 / zkp_msm_g1_partial / zkp_kzg_commit / zkp_kzg_open over sharded bases run
 *                                    the whole Pippenger on every device's chunk concurrently (one resident host thread per
 *                                    slot uploads that chunk's scalars over its own PCIe link) and add the per-device
 *                                    partial sums (192 B each) on the host -- EC addition is not a collective's reduction
 *                                    operator, so "all-reduce of partial sums" is gather + add, here without leaving the
 *                                    process;
 *                                  - zkp_g1_bases_precompute expands every chunk on its own device;
 *                                  - `*_dev` entries and the PLONK prover take single-slot handles (device memory belongs
 *                                    to one device; a 2^16-gate proof does not shard: one prover per device).
 *   zkp_set_device(slot)         slot used by THIS thread's entries that carry no handle (NTT, FRI, fixed-base, *_create);
 *                                -1 = default: slot 0, and zkp_g1_bases_create shards over all slots.
 *   zkp_device_count()           number of slots (0 before initialisation). */
int zkp_init(int device);
int zkp_init_devices(const int *devices, int n_devices);
int zkp_device_count(void);
int zkp_set_device(int slot);
void zkp_shutdown(void);
const char *zkp_last_error(void);
/* ABI version of this header (bumped on incompatible change). */
int zkp_abi_version(void);

/* ---- optional per-phase timing (used by bench.py for the roofline object).  When enabled the library brackets
 *      each kernel phase with HIP events on the launch stream.  Phase names: "msm_digits", "msm_sort",
 *      "msm_accumulate", "msm_bucket_reduce", "msm_tail_host" (host, wall clock), "ntt_fr_pass", "ntt_gl_pass",
 *      "fri_merkle".
 *      zkp_profile_read waits for the recorded events and returns the summed milliseconds and the number of
 *      records with that name since the last reset.  zkp_profile_enable(2) records the dominant kernel only ("msm_accumulate",
 *      events and clock stamps): every recorded phase boundary is a marker on the stream, a bubble of ~5 us inside the region a
 *      benchmark times; 1 records every phase, 0 none. ---- */
void zkp_profile_enable(int on);
void zkp_profile_reset(void);
int zkp_profile_read(const char *name, double *total_ms, uint64_t *count);
/* The shader clock a kernel family ran at.  The chip lowers its clock under load and boxes differ, so the same cycle count takes a
 * different time from box to box; while profiling is enabled, wave 0 of every workgroup of the instrumented kernels ("msm_accumulate")
 * stamps s_memtime (shader cycles) and s_memrealtime (constant 100 MHz) at its start and end.  *cycles / *ref_ticks x 100 MHz = the
 * clock held under that kernel's own load, weighted by wave lifetime, since the last zkp_profile_reset; *waves = stamped workgroups. */
int zkp_profile_clock_read(const char *name, uint64_t *cycles, uint64_t *ref_ticks, uint64_t *waves);
/* Issue rate of v_mad_u64_u32 -- the instruction a 381-bit Montgomery product is made of (392 per product) -- on this device, now:
 * `launches` back-to-back launches (~1 ms each) of a probe kernel at full occupancy.  *lane_mads_per_s is the measured peak the
 * multiply-add rate of msm_accumulate is priced against (bench.py: roofline.integer_issue), *clock_mhz the shader clock the probe
 * held.  A measurement aid, not part of the hot path. */
int zkp_probe_mad_rate(unsigned launches, double *lane_mads_per_s, double *clock_mhz, double *ms_per_launch);

/* ---- G1 bases: the SRS `Vec<G1Affine>` of kzg/src/srs.rs:14-21 uploaded ONCE (the reference clones it per
 *      commit, srs.rs:78-80).  `xy` is n x 12 limbs; `is_inf` may be NULL (no infinity points). ---- */
int zkp_g1_bases_create(const uint64_t *xy, const uint8_t *is_inf, size_t n, zkp_bases **out);
/* Same, from n x 12 limbs already in device memory (copied; the caller keeps its buffer). */
int zkp_g1_bases_create_dev(const void *d_xy, const uint8_t *d_is_inf, size_t n, void *stream, zkp_bases **out);
/* Optional one-off expansion of resident bases for the shared-bucket MSM: stores the ceil(256 / window_bits) multiples
 * 2^(window_bits * s) * P_i of every point (128 B each), so that all windows of a scalar fall into ONE bucket set: a wider
 * window (fewer bucket insertions per scalar), one bucket reduction instead of one per window, and no window combination.
 * Costs ceil(256/window_bits) x the memory (13 x at 20 bits: 1.7 GB for 2^20 points; 12 x at 22 bits: 103 GB for 2^26 of the
 * 288 GB) and ~650 field products per stored point, once per SRS.  Results of zkp_msm_g1* are unchanged (same group element).
 * window_bits: 9..24, or 0 = automatic (22 from 2^22 points, 20 from 2^19, 16 above 2^13, 14 above 2^11, 12 from 64, otherwise
 * left as is; below 2^19 points the MSM is latency-bound and the narrow windows go with bucket runs split over several lanes).  A
 * scalar is cut into ceil(256 / window_bits) slices; when that many windows overshoot the 256 bits (every width but 16) the
 * slices are balanced to floor/ceil(256 / slices) bits instead (19 -> 14 slices of 18/19 bits, 2^18 buckets; 20 -> 13 slices of
 * 19/20 bits, 2^19 buckets; 22 -> 12 slices of 21/22 bits, 2^21 buckets), so that no slice is short and no group of buckets
 * collects a multiple of the others' points.  Once expanded, every MSM over these bases
 * uses the shared bucket set (2^10 terms: 0.28 ms against 0.85 ms per-window, whose host-side window combination alone is
 * 0.4 ms).  For a sharded handle every chunk is expanded on its own device, all chunks alike (automatic width: that of the largest
 * chunk) and all-or-nothing: every device is asked for room before any chunk allocates, so a ZKP_E_NOMEM refusal leaves the whole
 * handle unexpanded and a later call with another width is accepted. */
int zkp_g1_bases_precompute(zkp_bases *b, unsigned window_bits);
size_t zkp_g1_bases_len(const zkp_bases *b);
/* How the bases are expanded: *window_bits = the width asked for (0 = not expanded), *slices = insertions per scalar. */
int zkp_g1_bases_info(const zkp_bases *b, unsigned *window_bits, unsigned *slices);
void zkp_g1_bases_destroy(zkp_bases *b);

/* ---- MSM: replaces the body of KzgScheme::evaluate_in_s, kzg/src/scheme.rs:84-96 (reached from commit :49,
 *      commit_vector :63, open :108, open_vector :132 and the 9 commit sites of plonk/src/prover.rs:92,123,150,
 *      267-268).  out = sum_{i<n} scalars[i] * bases[i]; n == 0 gives the identity (scheme.rs:94).
 *      n > len(bases) returns ZKP_E_SIZE (the reference asserts, scheme.rs:86).
 *      `scalars` is host memory (pageable is fine) and is read only while the call runs: from 2^19 terms over expanded bases it is
 *      uploaded in two or three ranges, the later ones by a resident uploader thread of the library underneath the kernels of the
 *      earlier ones. ---- */
int zkp_msm_g1(const zkp_bases *bases, const uint64_t *scalars, size_t n, uint64_t out_xy[12], uint8_t *out_is_inf);
/* Scalars already in device memory (n x 4 limbs).  Synchronises `stream` before returning the host result. */
int zkp_msm_g1_dev(const zkp_bases *bases, const void *d_scalars, size_t n, void *stream, uint64_t out_xy[12],
                   uint8_t *out_is_inf);
/* `count` MSMs over the same bases with scalar vectors of the same length n (device pointers), processed as one pass
 * through the kernels: the commit_round1 / SlicePoly::commit / W_zeta, W_zeta_omega groups of plonk/src/prover.rs:92,150,
 * 267-268.  out_xy: count x 12 limbs; out_is_inf: count bytes. */
int zkp_msm_g1_batch_dev(const zkp_bases *bases, const void *const *d_scalars, size_t count, size_t n, void *stream,
                         uint64_t *out_xy, uint8_t *out_is_inf);
/* Multi-GPU building block: the same sum left UNNORMALISED as an extended-Jacobian point
 * (X, Y, ZZ, ZZZ = 24 limbs, ZZ == 0 for the identity) so that per-GPU partial sums can be exchanged
 * (RCCL all-gather of 192 bytes per rank) and combined with zkp_g1_xyzz_sum. */
int zkp_msm_g1_partial_dev(const zkp_bases *bases, const void *d_scalars, size_t n, void *stream, uint64_t out_xyzz[24]);
/* The same from host scalars (sharded bases allowed: the partial is then already the sum over this process's devices; a
 * multi-node caller exchanges these between processes). */
int zkp_msm_g1_partial(const zkp_bases *bases, const uint64_t *scalars, size_t n, uint64_t out_xyzz[24]);
/* Sharded bases with the scalars already RESIDENT on the devices (a prover that keeps its polynomials in HBM): chunk i of the handle
 * (zkp_g1_bases_shard: its slot, HIP device, first point and length) multiplies the scalars at d_scalars[i], which must be memory of that
 * chunk's device; n is the TOTAL number of scalars (chunk i uses those of its range that are below n).  Every device runs its chunk
 * on its own stream concurrently; the call returns the affine sum.  A single-slot handle has one chunk (d_scalars[0]).
 * Ordering: the entry takes no stream argument and launches on each slot's own non-blocking stream (the legacy null stream for a
 * runtime with a single slot), neither of which is ordered after work on the caller's non-blocking streams.  zkp_msm_g1_sharded_dev
 * therefore waits for ALL work previously enqueued on each chunk's device before reading d_scalars[i] (hipDeviceSynchronize, single-
 * and multi-slot handles alike): a copy or kernel that produces the scalars on any stream of that device may still be in flight
 * when the call is made.  zkp_msm_g1_sharded_dev_after is the non-blocking form for resident pipelines: ready_events[i] is a
 * hipEvent_t the producer of d_scalars[i] recorded after enqueuing that work; the chunk's launch waits for it ON THE DEVICE
 * (hipStreamWaitEvent) and the host does not stall.  A null array or a null entry falls back to the device-wide wait for that chunk.
 * Work enqueued on the device concurrently with the call from another thread is not ordered against it. */
int zkp_g1_bases_shard_count(const zkp_bases *b);
int zkp_g1_bases_shard(const zkp_bases *b, size_t i, int *slot, int *device, size_t *offset, size_t *len);
int zkp_msm_g1_sharded_dev(const zkp_bases *bases, const void *const *d_scalars, size_t n, uint64_t out_xy[12],
                           uint8_t *out_is_inf);
int zkp_msm_g1_sharded_dev_after(const zkp_bases *bases, const void *const *d_scalars, void *const *ready_events, size_t n,
                                 uint64_t out_xy[12], uint8_t *out_is_inf);
/* Sum `count` extended-Jacobian partials (host memory, count x 24 limbs) and normalise to affine. */
int zkp_g1_xyzz_sum(const uint64_t *partials, size_t count, uint64_t out_xy[12], uint8_t *out_is_inf);

/* ---- KzgScheme mirror (host logic in csrc/kzg_host.hpp): commit / commit_vector, kzg/src/scheme.rs:49-67 --
 *      trailing zero coefficients are trimmed first, as DensePolynomial::from_coefficients_vec does; an SRS that
 *      is empty or shorter than the trimmed polynomial gives ZKP_E_SIZE (assert at scheme.rs:86). ---- */
int zkp_kzg_commit(const zkp_bases *srs, const uint64_t *coeffs, size_t len, uint64_t out_xy[12], uint8_t *out_is_inf);
/* open / open_vector, kzg/src/scheme.rs:108-142: out_eval = p(z), out = commit((p - p(z)) / (X - z)).
 * len == 0 returns ZKP_E_ARG (the reference panics with "at least 1", scheme.rs:112). */
int zkp_kzg_open(const zkp_bases *srs, const uint64_t *coeffs, size_t len, const uint64_t z[4], uint64_t out_xy[12],
                 uint8_t *out_is_inf, uint64_t out_eval[4]);

/* ---- single scalar multiplication: KzgScheme::commit_para, kzg/src/scheme.rs:78-82 (`g1_0.mul(para)`),
 *      6x per proof at plonk/src/prover.rs:183-188.  Serial by nature: computed on the host. ---- */
int zkp_g1_mul(const uint64_t base_xy[12], uint8_t base_is_inf, const uint64_t scalar[4], uint64_t out_xy[12],
               uint8_t *out_is_inf);
/* P_i = k_i * G for n scalars (fixed-base, on the GPU): Srs::new_from_secret, kzg/src/srs.rs:48-63, with
 * k_i = s^i, and the benchmark's base-point generator.  Output n x 12 limbs to device memory; d_out_is_inf
 * (nullable, n bytes) receives 1 where k_i == 0 (coordinates are then written as zeros). */
int zkp_g1_fixed_base_mul_dev(const void *d_scalars, size_t n, void *d_out_xy, uint8_t *d_out_is_inf, void *stream);
/* Self-test hook for the device field inversion used by the two entries above and by zkp_g1_bases_precompute (`into_affine` of
 * kzg/src/scheme.rs:92-93 on the GPU: Bernstein-Yang division steps, csrc/fq28_inv.hpp): inverts n raw base-field elements in device
 * memory, one lane each.  form 0: 12 x u32 limbs per element, Montgomery radix 2^384 (the arkworks form), any value below 2p in,
 * canonical a^-1 R out; form 1: the library's internal 14 x 28-bit limbs (+ 2 pad words = 64 B per element), Montgomery radix 2^392,
 * any value below 2p in, a^-1 R below 2p out.  0 maps to 0.  Not part of the hot path. */
int zkp_selftest_fq_inverse_dev(const void *d_in, size_t n, int form, void *d_out, void *stream);
/* [s^i]G for i < n into host memory (kzg/src/srs.rs:48-63: n = circuit_size + 3). */
int zkp_srs_g1(const uint64_t secret[4], size_t n, uint64_t *out_xy);

/* ---- NTT over Fr: ark-poly Radix2EvaluationDomain semantics as used by the reference --
 *      `Evaluations::interpolate` (plonk/src/prover.rs:374-375,463; plonk/src/circuit.rs:175,230-232) = inverse;
 *      the FFTs inside `&DensePolynomial * &DensePolynomial` (prover.rs:396-426,437) = forward + inverse.
 *      In place, natural order in and out, size 2^log_n (log_n <= 32 and fits memory).
 *      inverse != 0 scales by n^-1.  coset != NULL: forward scales coefficient j by coset^j first (coset_fft);
 *      inverse scales output j by coset^-j (coset_ifft). ---- */
int zkp_ntt_fr(uint64_t *data, unsigned log_n, int inverse, const uint64_t *coset /* nullable Fr */);
/* `batch` independent transforms of size 2^log_n stored back to back in device memory. */
int zkp_ntt_fr_dev(void *d_data, unsigned log_n, size_t batch, int inverse, const uint64_t *coset, void *stream);

/* Twiddle step of a four-step (multi-GPU) transform of total size 2^log_n: the rows x cols row-major block is multiplied
 * element-wise by omega_n^((row0 + r) * c) (omega_n^-1 when inverse != 0).  Used by zkp_hip/dist.py between the column and
 * row transforms, around the RCCL all-to-all transposes. */
int zkp_ntt_fr_twiddle_dev(void *d_data, size_t rows, size_t cols, size_t row0, unsigned log_n, int inverse, void *stream);

/* The two local halves of the four-step transform in the layouts the all-to-all exchanges produce, so that no transpose or
 * twiddle pass over the data is needed (zkp_hip/dist.py; BASELINE.json configs[4]):
 * zkp_ntt_fr_axis0_dev -- transforms of length 2^log_len (<= 2^16) along axis 0 of a row-major [2^log_len][cols] matrix (cols a
 *   power of two >= 4, the contiguous direction), natural order in and out, out-of-place or in place; when tw_log_n != 0
 *   output (k, b) is also multiplied by omega_{2^tw_log_n}^((tw_col0 + b) k) (omega^-1 and the 1/2^log_len factor when
 *   inverse).  The column transforms: the first all-to-all delivers [all rows n1][my columns].
 * zkp_ntt_fr_layout_dev -- `batch` transforms of size 2^log_n, natural order, out-of-place, reading a gathered input and/or
 *   writing a scattered output: with a layout, logical element e of transform b lives at element
 *       b * batch_stride + (e mod 2^lo_bits) + ((e >> lo_bits) mod 2^mid_bits) * mid_stride + (e >> (lo_bits + mid_bits)) * hi_stride
 *   (NULL = contiguous transforms, b * 2^log_n + e); when tw_log_n != 0 output k of transform b is multiplied by
 *   omega_{2^tw_log_n}^((tw_row0 + b) k).  The row transforms: the second all-to-all delivers [source rank][my rows][that
 *   rank's columns], possibly in several column chunks (mid field). */
typedef struct {
    unsigned lo_bits, mid_bits;
    size_t mid_stride, hi_stride, batch_stride; /* in elements */
} zkp_ntt_layout;
int zkp_ntt_fr_axis0_dev(const void *d_in, void *d_out, unsigned log_len, size_t cols, int inverse, unsigned tw_log_n,
                         size_t tw_col0, void *stream);
int zkp_ntt_fr_layout_dev(const void *d_in, void *d_out, unsigned log_n, size_t batch, int inverse,
                          const zkp_ntt_layout *in_layout, const zkp_ntt_layout *out_layout, unsigned tw_log_n, size_t tw_row0,
                          void *stream);

/* ---- The same transform spread over the device slots of ONE process (BASELINE configs[4]; the multi-GPU face of the
 *      GeneralEvaluationDomain call sites above: plonk/src/prover.rs:70,374-375,396-443, plonk/src/circuit.rs:170-176).
 *      Four-step decomposition N = N1 x N2 over G = zkp_device_count() slots (G a power of two; slots created by zkp_init_devices, which
 *      may list one device several times -- how a 1-GPU box runs this code): column transforms (zkp_ntt_fr_axis0_dev's kernels), an
 *      all-to-all, row transforms (zkp_ntt_fr_layout_dev's kernels), with the exchange done INSIDE the library as peer-to-peer block
 *      copies between the slots' devices (hipMemcpyPeerAsync, every slot pulling its G blocks on its own copy stream: all xGMI links of a
 *      device busy at once; plain device-to-device copies between slots that share a device).  The columns are cut into `chunks`
 *      groups that go through pack / exchange / column transform as a pipeline (the copies of group q + 1 run under the transforms of
 *      group q).  No collective library, no other process: one resident host thread per slot drives its device.
 *
 *      Geometry (zkp_ntt_fr_sharded_geometry): N1 = 2^log_n1 with log_n1 = max(min(8, ceil(log_n / 2)), log2 G), N2 = N / N1, r1 = N1 / G,
 *      r2 = N2 / G (>= 4, otherwise ZKP_E_ARG: the transform is too small for this many slots), cw = r2 / chunks (>= 4).
 *      Slot g holds N / G elements at d_slabs[g] (memory of slot g's device), in one of three layouts:
 *        ZKP_NTT_NATURAL   elements [g N/G, (g+1) N/G) of the vector, i.e. rows [g r1, (g+1) r1) of the row-major N1 x N2 matrix
 *        ZKP_NTT_K1SLAB    slab g [j][k2] = X[(g r1 + j) + N1 k2]: the transposed order a four-step transform leaves; pointwise
 *                          products between a forward and an inverse transform do not care, and an MSM does not either
 *        ZKP_NTT_COLUMNS   slab g [q][n1][c] = x[n1 N2 + g r2 + q cw + c], q < chunks, c < cw: the COLUMNS [g r2, (g+1) r2) of every row
 *                          (chunk-major; `chunks` is part of this layout)
 *      Supported (layout_in -> layout_out), each for inverse == 0 and != 0 (in place: the result overwrites d_slabs[g]):
 *        NATURAL -> K1SLAB   two exchanges        K1SLAB -> NATURAL   two exchanges (the mirror image)
 *        COLUMNS -> K1SLAB   ONE exchange         K1SLAB -> COLUMNS   ONE exchange
 *        NATURAL -> NATURAL  three exchanges (the order ark-poly's fft / ifft return)
 *      Reading "input in K1SLAB order" as index i = i1 + N1 i2 stored at [i1][i2], every pair is the full transform of that vector.
 *      chunks: 0 = automatic (4, fewer when r2 < 16); otherwise a power of two with r2 / chunks >= 4 (anything else is ZKP_E_ARG,
 *      never adjusted silently).
 *      streams: NULL -- the library launches on each slot's own stream after waiting for all work enqueued on that slot's device
 *      (hipDeviceSynchronize) and returns when every device has finished; or G hipStream_t (streams[g] on slot g's device): the work
 *      of slot g is enqueued on streams[g], ordered after what that stream already holds, and the call returns without waiting (the
 *      copy streams are joined back into streams[g] by events before the call returns).
 *      Workspace per slot: two slabs of exchange buffers + the transform's scratch slab.
 *      zkp_ntt_fr_sharded is the host-pointer form (natural order in and out, optional coset, like zkp_ntt_fr): every slot uploads and
 *      downloads its own slab over its own PCIe link.  zkp_ntt_fr itself takes this route when the library runs on more than one slot
 *      and log_n >= ZKP_NTT_SHARD_MIN_LOG (environment, default 24), so the Rust seam needs no change to use a whole node. ---- */
#define ZKP_NTT_NATURAL 0
#define ZKP_NTT_K1SLAB 1
#define ZKP_NTT_COLUMNS 2
typedef struct {
    unsigned slots, log_n1, log_n2, chunks;
    size_t r1, r2, cw, slab; /* rows / columns per slot, columns per chunk, elements per slot */
} zkp_ntt_shard_geometry;
int zkp_ntt_fr_sharded_geometry(unsigned log_n, unsigned slots /* 0 = zkp_device_count() */, unsigned chunks,
                                zkp_ntt_shard_geometry *out);
int zkp_ntt_fr_sharded_dev(void *const *d_slabs, unsigned log_n, int inverse, int layout_in, int layout_out, unsigned chunks,
                           void *const *streams);
int zkp_ntt_fr_sharded(uint64_t *data, unsigned log_n, int inverse, const uint64_t *coset /* nullable Fr */);

/* ---- NTT over Goldilocks: the evaluation loop of FriLayer::from_poly, fri/src/fri_layer.rs:40-46 ---- */
int zkp_ntt_goldilocks(uint64_t *data, unsigned log_n, int inverse, const uint64_t *coset /* nullable, 1 limb */);
int zkp_ntt_goldilocks_dev(void *d_data, unsigned log_n, size_t batch, int inverse, const uint64_t *coset, void *stream);
/* evals[i] = poly(coset * omega_D^i), i < D = 2^log_D, natural order (fri_layer.rs:40-46); d <= D coefficients. */
int zkp_fri_layer_eval(const uint64_t *coeffs, size_t d, uint64_t coset, unsigned log_D, uint64_t *out);
/* fold_polynomial, fri/src/prover.rs:34-42: out[j] = c[2j] + r*c[2j+1]; out has ceil(d/2) entries. */
int zkp_fri_fold(const uint64_t *coeffs, size_t d, uint64_t r, uint64_t *out);

/* ---- FRI commitment path around the NTT.  Field elements are Goldilocks memory-form limbs. ----
 * MerkleTree::new, fri/src/merkle_tree.rs:42-63, with hash / hash_slice of fri/src/hasher.rs:14-36 (SHA-256 over the
 * decimal strings, digest taken mod p as a little-endian integer).  `nodes` receives every level, concatenated: the n leaf
 * hashes, then ceil(n/2) parents, ... up to the root (zkp_fri_merkle_node_count(n) elements; the root is the last one). */
size_t zkp_fri_merkle_node_count(size_t n);
int zkp_fri_merkle_tree(const uint64_t *leaves, size_t n, uint64_t *nodes_out);
int zkp_fri_merkle_tree_dev(const void *d_leaves, size_t n, void *d_nodes, void *stream);
/* Transcript replay, fri/src/fiat_shamir/transcript.rs:30-139 as used by fri/src/verifier.rs:13-29: r_out[l] = the folding
 * challenge drawn after digesting root l (memory form); q_out[i] = the i-th query challenge as usize (before % domain),
 * drawn after digesting const_val.  Host only. */
int zkp_fri_challenges(const uint64_t *roots, size_t layers, uint64_t const_val, size_t num_queries, uint64_t *r_out,
                       uint64_t *q_out);
/* generate_proof, fri/src/prover.rs:141-168 (folding phase on the GPU: coset NTT + Merkle tree + fold per layer; query
 * decommitments gathered on the GPU).  *out_proof is a malloc'ed flat proof of *out_words words, layout:
 *   [0] domain_size [1] layers = log2(domain_size) [2] number_of_queries [3] coset (= 7)
 *   layers_root[layers], const_val, then per query, per layer l: index, evaluation, sym_evaluation,
 *   auth path (log2(domain_size >> l) sibling hashes from the leaf level up), sym auth path (same length).
 * Release with zkp_free.  A zero polynomial is ZKP_E_ARG (the reference's assert at prover.rs:72 panics). */
int zkp_fri_prove(const uint64_t *coeffs, size_t d, size_t blowup_factor, size_t num_queries, uint64_t **out_proof,
                  size_t *out_words);
/* verify, fri/src/verifier.rs:10-127, on the flat proof (host only).  ZKP_OK = accepted; ZKP_E_ARG otherwise, with the
 * reference's error string ("wrong index!", "verify Merkle path failed!", "folding wrong!") in zkp_last_error(). */
int zkp_fri_verify(const uint64_t *proof, size_t words);
void zkp_free(void *p);

/* ---- ChallengeGenerator<Sha256>, plonk/src/challenge.rs:22-77 (host): feed commitments (serialize_uncompressed, 96
 *      bytes), then draw Fr challenges (memory form, n x 4 limbs) through StdRng::seed_from_u64 + Fr::rand.  Drawing twice
 *      without feeding is ZKP_E_ARG (the reference panics "I'm hungry! Feed me something first"). ---- */
typedef struct zkp_plonk_transcript zkp_plonk_transcript;
int zkp_plonk_transcript_create(zkp_plonk_transcript **out);
void zkp_plonk_transcript_destroy(zkp_plonk_transcript *t);
int zkp_plonk_transcript_feed(zkp_plonk_transcript *t, const uint64_t xy[12], uint8_t is_inf);
int zkp_plonk_transcript_challenges(zkp_plonk_transcript *t, size_t n, uint64_t *out);

/* ---- polynomial product with ark-poly `Mul` semantics (plonk/src/prover.rs:396-426): FFT-based on the radix-2
 *      domain of size next_pow2(la+lb-1); out has la+lb-1 entries; either operand empty gives an empty product ---- */
int zkp_poly_mul_fr(const uint64_t *a, size_t la, const uint64_t *b, size_t lb, uint64_t *out);

/* ---- PLONK prover rounds: plonk/src/prover.rs::generate_proof (61-293), one entry per round.  Blinders (the
 *      reference draws them from StdRng::from_entropy(), prover.rs:68-77,104-106) and Fiat-Shamir challenges (the
 *      reference's SHA-256 ChallengeGenerator, plonk/src/challenge.rs) are INPUTS: the caller keeps its transcript and
 *      feeds the returned commitments to it between rounds.  Polynomials stay in HBM between rounds.
 *      create: the CompiledCircuit (plonk/src/compiled_circuit.rs, constraint.rs) as 12 coefficient vectors of at most
 *      n = 2^log_n entries, in the order q_m q_l q_r q_o q_c pi f_a f_b f_c s_sigma_1 s_sigma_2 s_sigma_3, plus k1, k2
 *      (circuit.rs:238-245); the SRS must hold n + 3 points (kzg/src/srs.rs:51). ---- */
typedef struct zkp_plonk_prover zkp_plonk_prover;
int zkp_plonk_prover_create(const zkp_bases *srs, unsigned log_n, const uint64_t *const polys[12], const size_t lens[12],
                            const uint64_t k1[4], const uint64_t k2[4], zkp_plonk_prover **out);
void zkp_plonk_prover_destroy(zkp_plonk_prover *p);
/* Round 1 (prover.rs:68-92): b1..b6 -> commitments to a(X), b(X), c(X) (3 x 12 limbs, 3 infinity bytes). */
int zkp_plonk_round1(zkp_plonk_prover *p, const uint64_t *blinders, uint64_t *out_xy, uint8_t *out_is_inf);
/* Round 2 (prover.rs:98-123, compute_acc 302-377): beta, gamma, b7..b9 -> commitment to z(X). */
int zkp_plonk_round2(zkp_plonk_prover *p, const uint64_t beta[4], const uint64_t gamma[4], const uint64_t *blinders,
                     uint64_t out_xy[12], uint8_t *out_is_inf);
/* Round 3 (prover.rs:136-150, compute_quotient_polynomial 381-444, SlicePoly): alpha -> t_lo, t_mid, t_hi commitments and
 * the slice degree.  An unsatisfied circuit returns ZKP_E_ARG (the reference panics with "No remainder"). */
int zkp_plonk_round3(zkp_plonk_prover *p, const uint64_t alpha[4], uint64_t *out_xy, uint8_t *out_is_inf, size_t *out_degree);
/* Round 4 (prover.rs:156-178): zeta -> bar_a bar_b bar_c bar_s_sigma_1 bar_s_sigma_2 bar_z_w (6 x 4 limbs). */
int zkp_plonk_round4(zkp_plonk_prover *p, const uint64_t zeta[4], uint64_t *out_bars);
/* Round 5 (prover.rs:183-268, compute_linearisation_polynomial 469-568): v -> commitments to W_zeta, W_zeta_omega. */
int zkp_plonk_round5(zkp_plonk_prover *p, const uint64_t v[4], uint64_t *out_xy, uint8_t *out_is_inf);
/* generate_proof in ONE call, plonk/src/prover.rs:61-293: rounds 1-5 above driven by the reference's transcript
 * (ChallengeGenerator<Sha256>, plonk/src/challenge.rs; the six evaluations are fed as commit_para(bar), scheme.rs:78-82).
 * blinders = b1..b9 (the reference draws them from StdRng::from_entropy, prover.rs:66-75,103-105).  Field order of the
 * proof = struct Proof, prover.rs:23-41. */
typedef struct {
    uint64_t commit_xy[9][12]; /* a, b, c, z, t_lo, t_mid, t_hi, w_ev_x, w_ev_wx */
    uint8_t commit_is_inf[9];
    uint64_t bars[6][4];       /* bar_a, bar_b, bar_c, bar_s_sigma_1, bar_s_sigma_2, bar_z_w */
    uint64_t u[4];
    uint64_t degree;
} zkp_plonk_proof;
int zkp_plonk_prove(zkp_plonk_prover *p, const uint64_t *blinders, zkp_plonk_proof *out);

/* ---- Verifiers with real pairings (host code, never a throughput path: ~20 ms per pairing).
 *      G2 points in arkworks memory form: x.c0 x.c1 y.c0 y.c1, 4 x 6 Montgomery limbs; Fq12 = 12 x 6 limbs in tower order.
 *      The pairing value is the reduced optimal ate pairing with the plain exponent (p^12 - 1) / r; the reference only ever
 *      compares two values (kzg/src/scheme.rs:157-159,244; plonk/src/verifier.rs:151).
 *      The ABI is the deserialisation boundary (the reference only ever holds typed, validated arkworks points): every finite G1
 *      and G2 input of the entries below must have canonical limbs (< p), lie on its curve and in the prime-order subgroup
 *      ([r]P = O; both cofactors are large), otherwise the call fails with ZKP_E_ARG before any pairing runs. ---- */
int zkp_g2_generator(uint64_t out_xy[24]);
int zkp_g2_mul(const uint64_t q_xy[24], uint8_t q_is_inf, const uint64_t scalar[4], uint64_t out_xy[24], uint8_t *out_is_inf);
int zkp_pairing(const uint64_t p_xy[12], uint8_t p_is_inf, const uint64_t q_xy[24], uint8_t q_is_inf, uint64_t out_fq12[72]);
/* KzgScheme::verify, kzg/src/scheme.rs:143-160; g2s = [s]_2 (Srs::g2s, kzg/src/srs.rs:66).  *accepted = 1 / 0. */
int zkp_kzg_verify(const uint64_t g2s_xy[24], const uint64_t commit_xy[12], uint8_t commit_is_inf, const uint64_t w_xy[12],
                   uint8_t w_is_inf, const uint64_t y[4], const uint64_t z[4], int *accepted);
/* KzgScheme::batch_verify, kzg/src/scheme.rs:215-245; r_primes (n x 4 limbs) stand for Fr::from(rng.gen::<u128>()). */
int zkp_kzg_batch_verify(const uint64_t g2s_xy[24], size_t n, const uint64_t *commits_xy, const uint8_t *commits_is_inf,
                         const uint64_t *points, const uint64_t *openings_xy, const uint8_t *openings_is_inf,
                         const uint64_t *evals, const uint64_t *r_primes, int *accepted);
/* KzgScheme::aggregate_commitments, kzg/src/scheme.rs:187-202 (test: kzg/src/commitment.rs:78-89): sum_i challenge^i * C_i over n
 * commitments (n x 12 limbs; commits_is_inf nullable); host code, the points are validated like the verifiers' inputs. */
int zkp_kzg_aggregate_commitments(const uint64_t *commits_xy, const uint8_t *commits_is_inf, size_t n, const uint64_t challenge[4],
                                  uint64_t out_xy[12], uint8_t *out_is_inf);
/* verify, plonk/src/verifier.rs:19-157, against the circuit and SRS held by `p`: *accepted = 1 accepted, 0 "Pairing failed,
 * rejected", -1 "Challenge verification failed".  The eight circuit commitments and pi(zeta) are computed on the GPU. */
int zkp_plonk_verify(zkp_plonk_prover *p, const uint64_t g2s_xy[24], const zkp_plonk_proof *proof, int *accepted);

/* Parity accessor: copy a working polynomial to the host.  which: 0 ax 1 bx 2 cx 3 z 4 r 5 W_zeta 6 W_zeta_omega
 * 7 tx_compact 8 t (before round 5). */
int zkp_plonk_get_poly(zkp_plonk_prover *p, int which, uint64_t *out, size_t cap_elems, size_t *len);

#ifdef __cplusplus
}
#endif
#endif /* ZKP_HIP_H */
