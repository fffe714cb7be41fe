/* mzk.h -- C ABI of libmi355zk: the MI355X (gfx950) backend for the jf-plonk prover's
 * arithmetic hot path (radix-2 NTT over Fr, Pippenger MSM on G1).
 *
 * The reference (renegade-fi/mpc-jellyfish) has no FFI of its own: the path sits behind two
 * third-party Rust trait surfaces.  Each entry point below names the reference call it replaces
 * (paths relative to the reference tree); INTEGRATION.md shows the Rust-side binding.
 *
 * Conventions
 *   curve_id        0 = BLS12-381, 1 = BN254.
 *   field elements  little-endian 64-bit limbs; Fr = 4 limbs; Fq = 6 limbs (BLS12-381) / 4 (BN254).
 *                   "mont" = Montgomery form a*R mod p with R = 2^(64*limbs), fully reduced --
 *                   the in-memory image of ark-ff's Fp<MontBackend<_,N>,N>.
 *   affine points   packed x||y (mont), no flag word; (0,0) encodes the point at infinity.
 *   Jacobian out    X||Y||Z (mont); Z = 0 encodes infinity (ark-ec `Projective::new_unchecked(X,Y,Z)`).
 *   return value    0 on success, < 0 on error (mzk_strerror); never unwinds, never aborts.
 *   threading       every entry point may be called concurrently (the reference calls commit and
 *                   fft from Rayon workers: univariate_kzg/mod.rs:125-127, prover.rs:552-562);
 *                   the kernels of all calls ON ONE DEVICE are enqueued under that device's lock (its
 *                   plan cache and workspace), but the host-pointer entry points move their operands on
 *                   per-call I/O streams outside it, so one caller's transfers overlap another's
 *                   kernels (see mzk_host_alloc).  Calls on different devices never contend (mzk_init).
 *   memory          host buffers are borrowed for the duration of the call; the library owns all
 *                   device memory it allocates, including the registered SRS copy.
 *   "_dev" variants take device pointers (hipMalloc / torch tensors) and a hipStream_t passed as
 *                   void* (NULL = the null stream); they do not synchronise unless stated.
 */
#ifndef MZK_H
#define MZK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define MZK_API __attribute__((visibility("default")))
#else
#define MZK_API
#endif

#define MZK_OK 0
#define MZK_ERR_INVALID_ARG (-1)
#define MZK_ERR_HIP (-2)          /* a HIP runtime call failed; text via mzk_last_error() */
#define MZK_ERR_NO_DEVICE (-3)
#define MZK_ERR_BAD_HANDLE (-4)
#define MZK_ERR_UNSUPPORTED (-5)
#define MZK_ERR_OOM (-6)
#define MZK_ERR_NOT_INIT (-7)
#define MZK_ERR_LOOKUP (-8)       /* Plookup: a lookup value is not in the table (constraint_system.rs:1410-1412) */

#define MZK_CURVE_BLS12_381 0
#define MZK_CURVE_BN254 1

/* Create the library's context on HIP device `device` if it has none yet and bind the CALLING THREAD to it (idempotent;
 * -1 = the thread's / the process's current binding, else the current HIP device).  Fails with MZK_ERR_NO_DEVICE when no GPU
 * is visible: there is no CPU fallback.
 *
 * Several devices from one process (`mzk_init(n_devices)`, "copies to device(s) once"; 8(e)): call
 * mzk_init(g) for every device g, one host thread per device.  Each device has its own context -- lock, workspace, NTT plan
 * cache, I/O slots, SRS and proving-key registries -- so calls on different devices run concurrently and share nothing.
 * SRS and proving-key handles carry their device: a call that takes one runs on that device whatever thread makes it.  Calls
 * that take bare device pointers (mzk_ntt_dev, mzk_poly_*, mzk_dev_*, the register functions) run on the calling thread's
 * device: the last mzk_init / mzk_set_device on that thread, or -- for threads that never bound one, e.g. the Rayon workers of a
 * single-GPU prover -- the first device initialised.  Device memory belongs to the device it was allocated on; between devices
 * it moves with mzk_dev_copy_peer.
 * MZK_VIRTUAL_DEVICES=G (environment, read at mzk_init): devices 0..G-1 exist whatever the machine has, mapped round-robin onto
 * the physical GPUs -- G full contexts on one card, to rehearse a multi-GPU host on a one-GPU box (results are identical). */
MZK_API int32_t mzk_init(int32_t device);
/* Rebind the calling thread to an initialised device / report its binding / number of devices mzk_init accepts. */
MZK_API int32_t mzk_set_device(int32_t device);
MZK_API int32_t mzk_get_device(int32_t* out_device);
MZK_API int32_t mzk_device_count(int32_t* out_count);
/* Releases every context (all devices). */
MZK_API int32_t mzk_shutdown(void);
MZK_API const char* mzk_strerror(int32_t code);
MZK_API const char* mzk_last_error(void);
MZK_API const char* mzk_version(void);

/* ---- SRS (CommitKey = UnivariateProverParam{powers_of_g}; primitives/src/pcs/univariate_kzg/srs.rs:36-40) ---- */

/* Copy n_points affine bases to the device once; returns an opaque handle. */
MZK_API int32_t mzk_srs_register(int32_t curve_id, const uint64_t* xy_mont, uint64_t n_points, uint64_t* out_handle);
MZK_API int32_t mzk_srs_register_dev(int32_t curve_id, const void* d_xy_mont, uint64_t n_points, uint64_t* out_handle, void* stream);
MZK_API int32_t mzk_srs_release(uint64_t handle);
/* Testing SRS on the device: point i = beta^i * G (G = standard generator), beta canonical 4 limbs.
 * Mirrors gen_srs_for_testing (srs.rs:118-153) with g fixed to the generator. */
MZK_API int32_t mzk_srs_generate_for_testing(int32_t curve_id, const uint64_t* beta_canonical, uint64_t n_points, uint64_t* out_handle);
/* The same with a base point of the caller's: point i = beta^i * g, g affine x||y (mont), on the curve and in the subgroup (not
 * checked).  `universal_setup_for_testing` (plonk/src/proof_system/snark.rs:495-517) draws beta = Fr::rand, then g = G1::rand and
 * h = G2::rand from the same rng: a host that mirrors those draws passes its g here.  g_xy_mont = NULL: the standard generator. */
MZK_API int32_t mzk_srs_generate_for_testing_g(int32_t curve_id, const uint64_t* beta_canonical, const uint64_t* g_xy_mont, uint64_t n_points,
                                               uint64_t* out_handle);
/* The testing SRS over the LAGRANGE basis of the gate domain H (|H| = 2^log_n, generated by the primitive root w): point i = L_i(beta) g
 * for i < 2^log_n, then n_extra points beta^j (beta^n - 1) g = the commitments of X^j Z_H(X).  For a polynomial p with values v_i = p(w^i)
 * and its masked form p + (b_0 + b_1 X + ..) Z_H (mask_polynomial, prover.rs:463-486), an MSM of (v_0 .. v_(n-1), b_0, b_1, ..) over this
 * SRS is the SAME group element as the commitment of the masked coefficients over beta^i g (univariate_kzg/mod.rs:90-116) -- but the
 * scalars are the witness VALUES, which are mostly small numbers (flags, counters, 64-bit amounts): their high digits are zero and the MSM
 * touches a fraction of the table rows (mzk_msm_*: zero digits cost nothing, heavy buckets have their own kernels).  Round 1 of both
 * hosts commits the wires this way when given such a key.  (From an SRS without trapdoor the same points are the inverse group-NTT of
 * its first 2^log_n points: mzk_srs_lagrange_from_srs.) */
MZK_API int32_t mzk_srs_generate_lagrange_for_testing(int32_t curve_id, const uint64_t* beta_canonical, const uint64_t* g_xy_mont, uint32_t log_n,
                                                      uint32_t n_extra, uint64_t* out_handle);
/* The same key from the points of a registered SRS alone (no trapdoor): [L_i(beta)]g = (1/n) sum_j w^(-ij) [beta^j]g, the inverse NTT over the
 * group of its first 2^log_n points, then [beta^(n+j)]g - [beta^j]g for the n_extra tail (the SRS must hold 2^log_n + n_extra points).
 * (n / 2) log2 n scalar multiplications: 0.51 s at 2^20 on BLS12-381, 0.22 s on BN254 -- once per SRS and domain size.  Set-up peak:
 * (5n + 8) internal points of scratch (1.17 GB at 2^20 on BLS12-381), handed back before the call returns.  Synchronises. */
MZK_API int32_t mzk_srs_lagrange_from_srs(uint64_t srs_handle, uint32_t log_n, uint32_t n_extra, uint64_t* out_handle);
/* A new SRS handle holding the points [first, first + n_points) of a registered one (a device copy; `trim` generalised, srs.rs:77-93): a rank
 * of a multi-GPU prover keeps the range it commits over -- 1 / G of the points and of the fixed-base table, built with the window that suits
 * the slice's size -- and hands it to mzk_prover_create as its commit key (see there).  The source stays registered. */
MZK_API int32_t mzk_srs_slice(uint64_t handle, uint64_t first, uint64_t n_points, uint64_t* out_handle);
MZK_API int32_t mzk_srs_download(uint64_t handle, uint64_t first, uint64_t n_points, uint64_t* out_xy_mont);
MZK_API int32_t mzk_srs_len(uint64_t handle, uint64_t* out_n_points);

/* ---- MSM: replaces <E::G1 as VariableBaseMSM>::msm_bigint(bases, bigints)
 *      at univariate_kzg/mod.rs:109-111 (commit) and :151-155 (open).
 * Computes sum_{i<n} scalars[i] * srs[base_offset + i]; base_offset = num_leading_zeros of
 * skip_leading_zeros_and_convert_to_bigints (mod.rs:379-388).  scalars: n x 4 limbs, canonical
 * integers (msm_bigint semantics) or, with scalars_are_mont != 0, Montgomery Fr as stored in a
 * DensePolynomial (saves the CPU-side into_bigint pass, mod.rs:390-395).
 * n = 0 yields infinity.  Fails with MZK_ERR_INVALID_ARG if base_offset + n exceeds the SRS
 * (the reference's degree guard, mod.rs:98-104).
 * The result POINT is unique; its Jacobian representative (X, Y, Z) is not (bucket entries are ordered by atomics),
 * exactly as ark-ec's Projective result is only defined up to scaling: normalise before comparing or hashing, as the
 * reference does with `.into_affine()` (mod.rs:111) -- mzk_msm_affine / mzk_g1_jacobian_to_affine. */
MZK_API int32_t mzk_msm(uint64_t srs_handle, uint64_t base_offset, const uint64_t* scalars, uint64_t n,
                int32_t scalars_are_mont, uint64_t* out_xyz_mont);
/* Device-resident scalars; the result lands in host memory (the call synchronises the stream). */
MZK_API int32_t mzk_msm_dev(uint64_t srs_handle, uint64_t base_offset, const void* d_scalars, uint64_t n,
                    int32_t scalars_are_mont, uint64_t* out_xyz_mont, void* stream);
/* batch_commit (mod.rs:119-131) in one call: n_polys independent MSMs over one SRS. */
MZK_API int32_t mzk_msm_batch(uint64_t srs_handle, uint32_t n_polys, const uint64_t* const* scalars, const uint64_t* lens,
                      const uint64_t* base_offsets, int32_t scalars_are_mont, uint64_t* out_xyz_mont);
/* Device-resident scalars: d_scalars[i] is a device pointer to lens[i] x 4 limbs.  The MSMs run back to back
 * and share one bucket-reduction phase and one host synchronisation; base_offsets may be NULL (all 0). */
MZK_API int32_t mzk_msm_batch_dev(uint64_t srs_handle, uint32_t n_polys, const void* const* d_scalars, const uint64_t* lens,
                          const uint64_t* base_offsets, int32_t scalars_are_mont, uint64_t* out_xyz_mont, void* stream);
/* Same point as mzk_msm but normalised on the host: x||y (mont), (0,0) for infinity
 * (the `.into_affine()` of mod.rs:111 folded in). */
MZK_API int32_t mzk_msm_affine(uint64_t srs_handle, uint64_t base_offset, const uint64_t* scalars, uint64_t n,
                       int32_t scalars_are_mont, uint64_t* out_xy_mont);

/* Host-only (no GPU needed): out = sum of n Jacobian points (X||Y||Z mont each).  Used to combine the
 * per-GPU partial sums of a point-range-sharded MSM after an all-gather; RCCL has
 * no EC-add reduction, so the "all-reduce of partial EC sums" is all-gather + this local sum. */
MZK_API int32_t mzk_g1_sum_jacobian(int32_t curve_id, const uint64_t* xyz_mont, uint64_t n, uint64_t* out_xyz_mont);

/* Host-only: n Jacobian points -> affine x||y ((0,0) for infinity): `.into_affine()` (mod.rs:111) / normalize_batch. */
MZK_API int32_t mzk_g1_jacobian_to_affine(int32_t curve_id, const uint64_t* xyz_mont, uint64_t n, uint64_t* out_xy_mont);

/* Host-only: the Keccak-f[1600] permutation on 200 bytes (lane (x, y) at byte offset 8 (x + 5 y), little-endian): the
 * primitive under merlin's STROBE-128, for hosts that mirror the transcript of plonk/src/transcript/standard.rs:16-46
 * outside Rust (the Python prover mirror; a Rust caller keeps using merlin). */
MZK_API int32_t mzk_keccak_f1600(uint8_t* state200);

/* Host-only: n_blocks consecutive 64-byte ChaCha blocks (djb variant: 64-bit block counter in state words 12-13, stream id 0) as
 * 16 little-endian u32 words each -- rand_chacha's ChaCha{8,12,20}Rng core, for hosts that mirror `jf_utils::test_rng()`
 * (utilities/src/lib.rs:62-70: the blinders of mask_polynomial / split_quotient_polynomial) and
 * `compute_coset_representatives` (relation/src/constants.rs:30-80) outside Rust. */
MZK_API int32_t mzk_chacha_blocks(const uint32_t key[8], uint64_t counter, uint32_t rounds, uint32_t n_blocks, uint32_t* out_words);

/* ---- NTT: replaces EvaluationDomain::{fft,ifft}_in_place on Radix2EvaluationDomain
 *      forward coset: plonk/src/proof_system/prover.rs:554,557,561,566,567,579-591
 *      inverse coset: plonk/src/proof_system/prover.rs:672
 *      inverse plain: relation/src/constraint_system.rs:1172,1189,1221,1240,1257,1266-1287,1366,1414-1415
 * data: 2^log_n x 4 limbs (mont), transformed in place, natural order in and out.  Elements at
 * index >= in_len are treated as zero on input (the reference zero-pads short coefficient
 * vectors).  coset_offset_mont = NULL means offset 1; otherwise one Fr (mont), e.g. Fr::GENERATOR
 * for `quot_domain.get_coset(Fr::GENERATOR)` (prover.rs:545).
 *   forward: out[i] = sum_j c[j] (h w^i)^j      inverse: c[j] = h^-j N^-1 sum_i e[i] w^(-ij) */
MZK_API int32_t mzk_ntt(int32_t curve_id, uint64_t* data_mont, uint64_t in_len, uint32_t log_n, int32_t inverse,
                const uint64_t* coset_offset_mont);
MZK_API int32_t mzk_ntt_batch(int32_t curve_id, uint32_t n_polys, uint64_t* const* data_mont, const uint64_t* in_lens,
                      uint32_t log_n, int32_t inverse, const uint64_t* coset_offset_mont);
/* Device-resident: `batch` polynomials of 2^log_n elements each, `batch_stride` elements apart
 * (>= 2^log_n), transformed in place.  in_len applies to every polynomial.  coset_offset_mont is
 * a HOST pointer.  Asynchronous on `stream`. */
MZK_API int32_t mzk_ntt_dev(int32_t curve_id, void* d_data_mont, uint64_t in_len, uint32_t log_n, int32_t inverse,
                    const uint64_t* coset_offset_mont, uint32_t batch, uint64_t batch_stride, void* stream);

/* ---- TurboPlonk quotient round on the device: replaces the body of
 *      Prover::compute_quotient_polynomial, plonk/src/proof_system/prover.rs:512-673 (one instance, no Plookup).
 * mzk_plonk_pk_register keeps, per proving key, the coset evaluations of the 13 selector and 5 sigma
 * polynomials (`ProvingKey{selectors, sigmas}`, plonk/src/proof_system/structs.rs:575-590) on the
 * 8n-point quotient domain -- the reference recomputes those 18 FFTs in every proof (prover.rs:552-558).
 * selector_coeffs: 13 x poly_len, sigma_coeffs: 5 x poly_len Montgomery coefficients (low order first,
 * selector order q_lc[4], q_mul[2], q_hash[4], q_o, q_c, q_ecc); k_mont: the 5 coset representatives
 * `vk.k` (relation/src/constants.rs:30-80). */
MZK_API int32_t mzk_plonk_pk_register(int32_t curve_id, uint32_t log_n, uint32_t num_wire_types, const uint64_t* selector_coeffs,
                                      const uint64_t* sigma_coeffs, uint64_t poly_len, const uint64_t* k_mont, uint64_t* out_handle);
MZK_API int32_t mzk_plonk_pk_release(uint64_t pk_handle);
/* d_polys: device buffer of (5 + 2) x 8n elements: wire polynomials, then the permutation product z,
 * then the public-input polynomial, each as coefficients in its first in_len slots (the rest is
 * ignored).  The buffer is overwritten with the coset evaluations.  alpha/beta/gamma: host, Montgomery.
 * d_out: 8n elements, the coefficients of the quotient polynomial (what `coset.ifft` returns at
 * prover.rs:672; the caller strips trailing zeros as DensePolynomial::from_coefficients_vec does).
 * Asynchronous on `stream`. */
MZK_API int32_t mzk_plonk_quotient_dev(uint64_t pk_handle, void* d_polys, uint64_t in_len, const uint64_t* alpha_mont,
                                       const uint64_t* beta_mont, const uint64_t* gamma_mont, void* d_out, void* stream);
/* Host-pointer form: polys = 7 x in_len coefficients (wires, z, pi), out = 8n coefficients. */
MZK_API int32_t mzk_plonk_quotient(uint64_t pk_handle, const uint64_t* polys, uint64_t in_len, const uint64_t* alpha_mont,
                                   const uint64_t* beta_mont, const uint64_t* gamma_mont, uint64_t* out);

/* ---- UltraPlonk (Plookup).  The proving key additionally holds q_lookup as the 14th
 * selector, a 6th wire type (sigma, k) and the four table polynomials `PlookupProvingKey{range_table_poly,
 * key_table_poly, table_dom_sep_poly, q_dom_sep_poly}` (plonk/src/proof_system/structs.rs:592-640).
 * selector_coeffs: 14 x poly_len, sigma_coeffs: 6 x poly_len, table_coeffs: 4 x poly_len in the order range, key,
 * table_dom_sep, q_dom_sep; k_mont: 6 coset representatives. */
MZK_API int32_t mzk_plonk_pk_register_ultra(int32_t curve_id, uint32_t log_n, const uint64_t* selector_coeffs, const uint64_t* sigma_coeffs,
                                            const uint64_t* table_coeffs, uint64_t poly_len, const uint64_t* k_mont, uint64_t* out_handle);
/* Quotient with the Plookup terms (prover.rs:512-673 with compute_quotient_plookup_contribution, :773-888).
 * d_polys: (6 + 2 + 3) x 8n elements: wires, z, public input, h_1, h_2, Plookup product polynomial. */
MZK_API int32_t mzk_plonk_quotient_ultra_dev(uint64_t pk_handle, void* d_polys, uint64_t in_len, const uint64_t* tau_mont,
                                             const uint64_t* alpha_mont, const uint64_t* beta_mont, const uint64_t* gamma_mont, void* d_out,
                                             void* stream);
/* Round 1.5: replaces compute_merged_lookup_table (relation/src/constraint_system.rs:1290-1309) and the merge of
 * compute_lookup_sorted_vec_polynomials (:1370-1417, before its two iFFTs).  d_wire_values: 6 x n wire evaluations.
 * Outputs (device): merged table (n), merged lookup witness (n), sorted vector (2n - 1): h_1 = ifft(sorted[..n]),
 * h_2 = ifft(sorted[n-1..]).  Returns MZK_ERR_LOOKUP when a looked-up value is not in the table.  Synchronises. */
MZK_API int32_t mzk_plookup_sorted_vec_dev(uint64_t pk_handle, const void* d_wire_values, const uint64_t* tau_mont, void* d_merged_table,
                                           void* d_merged_lookup, void* d_sorted, void* stream);
/* Round 2.5: replaces compute_lookup_prod_polynomial (constraint_system.rs:1311-1368; one division per row there).
 * d_out: the n coefficients of the (unmasked) Plookup product polynomial.  Asynchronous. */
MZK_API int32_t mzk_plookup_product_dev(uint64_t pk_handle, const void* d_merged_table, const void* d_merged_lookup, const void* d_sorted,
                                        const uint64_t* beta_mont, const uint64_t* gamma_mont, void* d_out, void* stream);

/* ---- Coset-chunked quotient for multi-GPU proving.  The 8n-point coset g*H_8n is the union of the 8
 * cosets h_k*H_n, h_k = g*w_8n^k (the residue classes of the point index mod 8); the closure of prover.rs:605-659 reads
 * index i and (i + 8) mod 8n only, so each class is self-contained.  A chunked key holds the fixed polynomials on the listed
 * classes only (strictly increasing `classes`, each < 8); num_wire_types 5 (table_coeffs NULL) or 6.
 * mzk_plonk_quotient_chunked_dev: d_polys = (W + 2 [+ 3]) rows of in_stride elements, coefficients in the first in_len
 * (<= 2n) slots, NOT overwritten; d_out = n_classes x n elements: for each resident class k the n coefficients of
 * t mod (X^n - h_k^n) (local inverse coset NTT done).  After the ranks exchange these (one all-gather / all-to-all),
 * mzk_plonk_quotient_combine_dev turns the 8 class remainders (class-major, 8 x n) into the 8n coefficients that
 * `coset.ifft` returns at prover.rs:672 (an 8-point inverse DFT per coefficient index). */
MZK_API int32_t mzk_plonk_pk_register_chunked(int32_t curve_id, uint32_t log_n, uint32_t num_wire_types, const uint64_t* selector_coeffs,
                                              const uint64_t* sigma_coeffs, const uint64_t* table_coeffs, uint64_t poly_len,
                                              const uint64_t* k_mont, const uint32_t* classes, uint32_t n_classes, uint64_t* out_handle);
MZK_API int32_t mzk_plonk_quotient_chunked_dev(uint64_t pk_handle, const void* d_polys, uint64_t in_stride, uint64_t in_len,
                                               const uint64_t* tau_mont, const uint64_t* alpha_mont, const uint64_t* beta_mont,
                                               const uint64_t* gamma_mont, void* d_out, void* stream);
/* The same with flags.  MZK_QUOTIENT_PI_ZERO: the caller's public-input polynomial (row W + 1 of d_polys) is the zero polynomial -- a circuit
 * without public inputs, or all of them zero: its row is then neither transformed nor read (one coset NTT in W + 2 less per class). */
#define MZK_QUOTIENT_PI_ZERO 1u
MZK_API int32_t mzk_plonk_quotient_chunked_flags_dev(uint64_t pk_handle, const void* d_polys, uint64_t in_stride, uint64_t in_len, uint32_t flags,
                                                     const uint64_t* tau_mont, const uint64_t* alpha_mont, const uint64_t* beta_mont,
                                                     const uint64_t* gamma_mont, void* d_out, void* stream);
MZK_API int32_t mzk_plonk_quotient_combine_dev(int32_t curve_id, uint32_t log_n, const void* d_class_remainders, void* d_out, void* stream);
/* The same recovery from FEWER classes.  The quotient has degree W (n + 1) + 2 (`quotient_polynomial_degree`,
 * plonk/src/proof_system/prover.rs:916-919, 1126-1128), below (W + 1) n once n > W + 2: the remainders modulo X^n - h_k^n on any
 * W + 1 classes -- 6 of the 8 for TurboPlonk, 7 for UltraPlonk -- determine it (Chinese remainder theorem), so the other classes
 * need neither be evaluated nor be resident in the proving key.  d_class_remainders: n_classes x n elements, class-major in the
 * order of `classes` (strictly increasing, each < 8); per coefficient index an n_classes x n_classes inverse Vandermonde system in
 * h_k^n is applied (for all 8 classes it is the 8-point inverse DFT above).  d_out: 8n coefficients, the slabs above n_classes
 * zero: the polynomial `coset.ifft` returns at prover.rs:672, provided its degree is below n_classes * n. */
MZK_API int32_t mzk_plonk_quotient_combine_classes_dev(int32_t curve_id, uint32_t log_n, const uint32_t* classes, uint32_t n_classes,
                                                       const void* d_class_remainders, void* d_out, void* stream);
/* ONE class fewer: W classes (5 of the 8 for TurboPlonk, 6 for UltraPlonk).  The W + 3 coefficients of the quotient t from X^(Wn) on
 * are the top coefficients of its numerator (t (X^n - 1) = numerator and deg t = W (n + 1) + 2, so numerator[i] = t[i - n] above
 * that degree; needs n > W + 2), and those are products of the TOP coefficients of the numerator's factors: the two permutation
 * products of prover.rs:741-752 and, for TurboPlonk, the q_ecc / q_hash terms of :696-708 -- everything else lies lower.
 * mzk_plonk_quotient_top_dev computes them on the device (one small kernel; d_polys as for mzk_plonk_quotient_chunked_dev: rows of
 * in_stride elements, the W wire polynomials, then the permutation product, in_len >= n + 3 coefficient slots; key registered by
 * mzk_plonk_pk_register_chunked) into d_top[0 .. W + 3), *out_n_top = W + 3 (nullable).  mzk_plonk_quotient_combine_top_dev subtracts
 * their share h_k^(n n_classes) d_top[j] from the class remainders, solves the n_classes x n_classes system and places d_top in slab
 * n_classes of d_out.  NOTE: the polynomial so recovered has degree W (n + 1) + 2 whatever the witness -- the reference's guard
 * (`WrongQuotientPolyDegree`, prover.rs:915-918) can no longer fire; a host using this path must check the quotient identity itself
 * (both hosts here do, at the evaluation challenge: r(zeta) against the verifier's linearisation constant).  Asynchronous. */
MZK_API int32_t mzk_plonk_quotient_top_dev(uint64_t pk_handle, const void* d_polys, uint64_t in_stride, uint64_t in_len, const uint64_t* alpha_mont,
                                           const uint64_t* beta_mont, const uint64_t* gamma_mont, void* d_top, uint32_t* out_n_top, void* stream);
MZK_API int32_t mzk_plonk_quotient_combine_top_dev(int32_t curve_id, uint32_t log_n, const uint32_t* classes, uint32_t n_classes,
                                                   const void* d_class_remainders, const void* d_top, uint32_t n_top, void* d_out, void* stream);

/* Round 2: replaces Arithmetization::compute_prod_permutation_polynomial
 * (relation/src/constraint_system.rs:1197-1223), whose loop performs one field division per gate.
 * wire_values: 5 x n (UltraPlonk key: 6 x n) wire evaluations witness[wire_variable(i, j)]; the sigma evaluations
 * come from the registered proving key.  out: the n coefficients of the permutation product polynomial (unmasked). */
MZK_API int32_t mzk_plonk_perm_product_dev(uint64_t pk_handle, const void* d_wire_values, const uint64_t* beta_mont,
                                           const uint64_t* gamma_mont, void* d_out, void* stream);
MZK_API int32_t mzk_plonk_perm_product(uint64_t pk_handle, const uint64_t* wire_values, const uint64_t* beta_mont,
                                       const uint64_t* gamma_mont, uint64_t* out);

/* The gather of `Arithmetization::compute_wire_polynomials` (relation/src/constraint_system.rs:1225-1247: `self.witness[var]` for
 * every cell of `self.wire_variables[i]`) on the device: d_out[t] = d_witness[d_wire_variables[t]], t < count = W x n, 32-byte field
 * elements of either curve, d_wire_variables = u32 variable indices (wire-major).  The index table is circuit structure -- resident
 * once per circuit -- so a host-resident witness crosses PCIe as n_vars x 32 B instead of W x n x 32 B (the bench circuit: 1 / 5).
 * An index >= n_vars yields zero and is NOT reported here (the call is asynchronous): the reference panics on such an index, so validate
 * the table once where it is registered -- mzk_prover_set_wire_variables does (MZK_ERR_INVALID_ARG).  Asynchronous; the wire iNTTs (mzk_ntt_dev) follow on the same stream. */
MZK_API int32_t mzk_plonk_gather_witness_dev(const void* d_witness, uint64_t n_vars, const void* d_wire_variables, uint64_t count, void* d_out,
                                             void* stream);

/* ---- dense-polynomial primitives of prover rounds 4 and 5 on device-resident coefficient vectors ----
 * evaluate: `DensePolynomial::evaluate` (plonk/src/proof_system/prover.rs:216-235): `batch` polynomials of
 * `len` coefficients, `batch_stride` elements apart, all at the point x; out_mont: batch x 4 limbs on the
 * HOST (the call synchronises). */
MZK_API int32_t mzk_poly_eval_dev(int32_t curve_id, const void* d_coeffs, uint64_t len, uint32_t batch, uint64_t batch_stride,
                                  const uint64_t* x_mont, uint64_t* out_mont, void* stream);
/* Evaluations of several (batches of) polynomials at up to two points in ONE call and one synchronisation: job j evaluates batches[j]
 * polynomials of lens[j] coefficients, strides[j] elements apart from d_coeffs[j], at x_mont[which_x[j]] (which_x[j] is 0 or 1; x_mont
 * holds two elements, the second may repeat the first); out_mont receives the values job after job (sum of batches[] elements).  What
 * round 4 of the prover needs (prover.rs:216-299: everything at zeta and at zeta * omega).  At most 64 jobs. */
MZK_API int32_t mzk_poly_eval_many_dev(int32_t curve_id, uint32_t n_jobs, const void* const* d_coeffs, const uint64_t* lens, const uint32_t* batches,
                                       const uint64_t* strides, const uint32_t* which_x, const uint64_t* x_mont, uint64_t* out_mont, void* stream);
/* out[j] = sum_k scalars[k] * polys[k][j], j < out_len (a polynomial shorter than out_len counts as zero beyond its
 * length): `mul_poly` and the polynomial additions of prover.rs:302-358, 497-501, 1115-1122.  At most 32 terms;
 * d_out may be one of the inputs.  scalars_mont: n_terms x 4 limbs, host.  Asynchronous. */
MZK_API int32_t mzk_poly_lincomb_dev(int32_t curve_id, uint32_t n_terms, const void* const* d_polys, const uint64_t* lens,
                                     const uint64_t* scalars_mont, void* d_out, uint64_t out_len, void* stream);
/* `Prover::mask_polynomial` (prover.rs:463-486) for up to 8 polynomials in one launch: d_polys[i] holds n coefficients in a row
 * of at least n + n_blinders slots; p += (b_0 + b_1 X + ..)(X^n - 1), i.e. p[j] -= b_j and p[n + j] = b_j.  blinders_mont:
 * n_polys x n_blinders x 4 limbs, host (the `DensePolynomial::rand` draws, 1 <= n_blinders <= 4).  Asynchronous. */
MZK_API int32_t mzk_poly_mask_dev(int32_t curve_id, uint32_t n_polys, void* const* d_polys, uint64_t n, uint32_t n_blinders,
                                  const uint64_t* blinders_mont, void* stream);
/* `split_quotient_polynomial` (prover.rs:902-960) in one launch: the W (n + 1) + 3 coefficients at d_quot (the quotient's expected degree is
 * W (n + 1) + 2, W = n_parts, 2..8) into n_parts rows of out_stride (>= n + 3) slots at d_out: row i holds the coefficients
 * [i (n + 2), (i + 1)(n + 2)) -- the last row what is left -- plus the blinder b_i as coefficient n + 2 (rows before the last) minus
 * b_{i-1} in the constant term (rows after the first); every other slot of a row is zeroed.  blinders_mont: (n_parts - 1) x 4 limbs, host.
 * d_out must not overlap d_quot.  Asynchronous. */
MZK_API int32_t mzk_poly_split_quotient_dev(int32_t curve_id, const void* d_quot, uint64_t n, uint32_t n_parts, const uint64_t* blinders_mont,
                                            void* d_out, uint64_t out_stride, void* stream);
/* quotient of p(X) / (X - z), len - 1 coefficients into d_out (remainder dropped, as ark-poly's `/` does at
 * prover.rs:504-506).  d_out must not alias d_poly.  Asynchronous. */
MZK_API int32_t mzk_poly_div_linear_dev(int32_t curve_id, const void* d_poly, uint64_t len, const uint64_t* z_mont, void* d_out, void* stream);
/* The same division, and the remainder p(z) -- which its suffix sums hold anyway -- into *d_rem (one element, device).  The opening
 * proof's batch polynomial divided at zeta leaves (linearisation polynomial)(zeta) + sum_i v^i eval_i there: what a prover needs to
 * check the quotient identity at zeta itself (mzk_plonk_quotient_top_dev). */
MZK_API int32_t mzk_poly_div_linear_rem_dev(int32_t curve_id, const void* d_poly, uint64_t len, const uint64_t* z_mont, void* d_out, void* d_rem,
                                            void* stream);
/* *d_out_len (a u64 in DEVICE memory) = number of coefficients up to and including the highest non-zero one, 0 for the zero
 * polynomial: `DensePolynomial::degree` + 1 after `from_coefficients_vec` has stripped the trailing zeros.  The prover's only
 * guard against an unsatisfied witness is `quot_poly.degree() != expected_degree => WrongQuotientPolyDegree`
 * (plonk/src/proof_system/prover.rs:915-918).  Field-independent (an element is zero iff its 32 bytes are).  Asynchronous:
 * read the word back (mzk_dev_download) once the stream has been synchronised anyway. */
MZK_API int32_t mzk_poly_degree_dev(const void* d_poly, uint64_t len, uint64_t* d_out_len, void* stream);
/* floor quotient of p(X) by the vanishing polynomial of a proof-linking domain, Z_D(X) = prod_{i < count} (X - w^(first + i))
 * with w the primitive 2^log_order-th root of unity (GroupLayout{alignment = log_order, offset = first, size = count},
 * relation/src/proof_linking/mod.rs:16-54): len - count coefficients into d_out, remainder dropped.  Replaces
 * compute_vanishing_polynomial + `&diff / &vanishing_poly` of compute_linking_quotient
 * (plonk/src/proof_system/proof_linking.rs:119-158).  When p vanishes on the whole domain (a valid link) the quotient comes
 * from two coset NTTs and a pointwise 1 / Z_D(x); otherwise the linear factors are divided out one by one -- same
 * coefficients either way.  d_out must not alias d_poly.  The call synchronises the stream (it reads p at the roots). */
MZK_API int32_t mzk_poly_div_roots_dev(int32_t curve_id, const void* d_poly, uint64_t len, uint32_t log_order, uint64_t first, uint64_t count,
                                       void* d_out, void* stream);

/* ---- the prover's rounds on device-resident state: replaces the BODIES of Prover::run_1st_round .. compute_opening_proofs
 *      (plonk/src/proof_system/prover.rs:72-419) as PlonkKzgSnark::batch_prove_internal calls them (snark.rs:263-431).
 * One mzk_prover per (proving key, instance slot): it owns, in HBM, the coefficient forms of the key's fixed polynomials, their
 * evaluations on the residue classes of the quotient domain that the quotient needs (mzk_plonk_pk_register_chunked), and the
 * workspace of one proof in flight (`Oracles`, structs.rs:875-887).  The CALLER keeps what the reference's snark.rs keeps: the
 * transcript (challenges in, commitments / evaluations out), the rng (blinders in, in the reference's draw order: SURVEY.md
 * Appendix C) and the `Proof` struct.  Nothing but challenges, blinders, public inputs, commitments (affine x||y, mont; (0,0) =
 * infinity: `Commitment(G1Affine)`) and evaluations crosses the boundary per round; the witness crosses once (or not at all:
 * MZK_WITNESS_DEV_*).  All field elements Montgomery, 4 limbs.
 *
 *   reference                                          here
 *   Prover::new + ProvingKey (structs.rs:575-590)       mzk_prover_create
 *   run_1st_round          prover.rs:72-87             mzk_prover_round1        -> W wire commitments
 *   run_plookup_1st_round  prover.rs:89-118            mzk_prover_round1_5      -> h_1, h_2 commitments          (UltraPlonk)
 *   run_2nd_round          prover.rs:125-141           mzk_prover_round2        -> permutation-product commitment
 *   run_plookup_2nd_round  prover.rs:143-183           mzk_prover_round2_5      -> Plookup-product commitment    (UltraPlonk)
 *   run_3rd_round          prover.rs:192-209           mzk_prover_round3        -> W split-quotient commitments  (all instances)
 *   compute_evaluations (+ plookup)  prover.rs:216-299 mzk_prover_round4        -> ProofEvaluations (+ PlookupEvaluations)
 *   compute_(non_)quotient_component_for_lin_poly + compute_opening_proofs  prover.rs:302-460
 *                                                      mzk_prover_round5        -> opening proof, shifted opening proof (all instances)
 *
 * Rounds must be called in this order (MZK_ERR_STATE otherwise); a new mzk_prover_round1 starts the next proof on the same handle.
 * `batch_prove` over K instances: K handles (one per instance, same domain size and curve), rounds 1 - 2.5 and 4 per handle,
 * rounds 3 and 5 once with all handles (alpha_base_k = alpha^(3k), alpha^(7k) with Plookup: prover.rs:661-669, snark.rs:408-428).
 *
 * Unsatisfied witness: the reference's only guard is `WrongQuotientPolyDegree` (prover.rs:915-918), raised in round 3.  Here the
 * quotient is recovered from W residue classes and the top coefficients of its numerator (mzk_plonk_quotient_top_dev), which has
 * the expected degree whatever the witness; the guard is the quotient identity at zeta, checked at the end of round 5 on a value
 * the opening's division leaves anyway: mzk_prover_round5 returns MZK_ERR_WRONG_QUOTIENT_DEGREE (the proof must be discarded).
 * Tiny domains (n <= W + 2) keep the reference's guard and round 3 returns that code.
 *
 * Streams (round 5).  A handle runs its rounds on a non-blocking stream of its device context (two per context, handed to its handles in
 * turn; MZK_PROVER_NULL_STREAM=1: the null stream, as before).  mzk_prover_round1 makes that stream wait for everything the caller has
 * enqueued on the NULL stream (a device-resident witness written there is complete first; work on other streams of the caller's must be
 * synchronised by the caller); every round returns when its outputs are in host memory; mzk_prover_poly_dev's pointers are valid to read
 * on any stream once round 5 has returned.  Two proofs on one card -- two host threads, each bound (mzk_init) to its own device context
 * (MZK_VIRTUAL_DEVICES maps several contexts onto one GPU), or two handles of one context -- run concurrently where the hardware allows.
 * Domains whose quotient does not fit the 8n-point domain (num_wire_types * (n + 1) + 2 >= 8 n: n = 2; n = 4 with six wire types) are
 * refused by mzk_prover_create with MZK_ERR_UNSUPPORTED. */
#define MZK_ERR_WRONG_QUOTIENT_DEGREE (-9) /* PlonkError::WrongQuotientPolyDegree: the witness does not satisfy the circuit */
#define MZK_ERR_STATE (-10)                /* prover rounds called out of order */

/* Several devices / processes: this prover is rank `rank` of `world` -- it commits over the SRS points
 * [rank (n+3) / world ..) of every polynomial, evaluates ceil(W / world) residue classes of the quotient, runs rounds 4-5 on its
 * coefficient range.  The callbacks move a few hundred bytes through host memory (Jacobian partial sums of 144 / 96 B, field elements
 * of 32 B); every rank ends each round with the SAME commitments and evaluations.  all_gather: every rank's `bytes` bytes,
 * concatenated in rank order, into recv (world x bytes).  exchange_classes (nullable): make the class remainders of all ranks
 * resident in d_rem (n_classes x class_bytes, device; this rank has filled [first_own, first_own + n_own)); NULL = the prover pushes
 * its own classes into the peers' buffers itself (mzk_prover_set_peer_buffers, hipMemcpyPeerAsync over xGMI) and calls barrier.
 * Callbacks return 0 on success.  FAILURE: a rank whose round fails (bad argument, MZK_ERR_WRONG_QUOTIENT_DEGREE, a HIP error) returns
 * without entering the round's remaining collectives, so its peers would wait in all_gather / barrier for ever: the CALLER's
 * collectives must be abortable or time out (the in-tree LocalComm has an abort flag every waiter polls; torch.distributed callers set
 * a process-group timeout), and a non-zero return of a callback ends the round on that rank with MZK_ERR_INVALID_ARG. */
typedef struct mzk_comm {
    void* ctx;
    int32_t rank, world;
    int32_t (*all_gather)(void* ctx, const void* send, uint64_t bytes, void* recv);
    int32_t (*barrier)(void* ctx);
    int32_t (*exchange_classes)(void* ctx, void* d_rem, uint64_t class_bytes, uint32_t first_own, uint32_t n_own, uint32_t n_classes);
} mzk_comm;

/* selector_coeffs: nsel x poly_len (nsel = 13, or 14 with q_lookup last: num_wire_types 6), sigma_coeffs: W x poly_len, table_coeffs:
 * 4 x poly_len (range, key, table_dom_sep, q_dom_sep; NULL for TurboPlonk): `ProvingKey{selectors, sigmas, plookup_pk}` as
 * DensePolynomial coefficient vectors, low order first, poly_len <= n = 2^log_n (host).  k_mont: the W coset representatives `vk.k`.
 * commit_key: SRS handle with >= n + 3 points (`pk.commit_key`, trim(n + 2): snark.rs:535, 561).  lagrange_key: 0, or a handle from
 * mzk_srs_lagrange_from_srs(commit_key, log_n, 3): round 1 (and 1.5) then commit the wire VALUES over the Lagrange basis -- same
 * group elements, mostly small scalars.  comm: NULL = one device.  The prover lives on the calling thread's device.
 * With a comm, commit_key (and lagrange_key) may instead hold ONLY this rank's point range [rank (n+3) / world ..) -- exactly that many points
 * (mzk_srs_slice): the rank then keeps 1 / world of the SRS and of its fixed-base table. */
MZK_API int32_t mzk_prover_create(int32_t curve_id, uint32_t log_n, uint32_t num_wire_types, const uint64_t* selector_coeffs,
                                  const uint64_t* sigma_coeffs, const uint64_t* table_coeffs, uint64_t poly_len, const uint64_t* k_mont,
                                  uint64_t commit_key, uint64_t lagrange_key, const mzk_comm* comm, uint64_t* out_prover);
MZK_API int32_t mzk_prover_destroy(uint64_t prover);
/* `VerifyingKey{selector_comms, sigma_comms}` (preprocess, snark.rs:562-594) from the resident coefficient forms: nsel + W affine points,
 * then -- UltraPlonk, out_plookup_xy non-NULL -- range_table_comm, key_table_comm, table_dom_sep_comm, q_dom_sep_comm (:575-590). */
MZK_API int32_t mzk_prover_vk_commitments(uint64_t prover, uint64_t* out_xy_mont, uint64_t* out_plookup_xy_mont);      /* (with a comm: a collective, every rank calls it) */
/* `wire_variables` of the finalised circuit (relation/src/constraint_system.rs:1225-1247), W x n u32 (host), every entry < n_vars (checked:
 * MZK_ERR_INVALID_ARG -- the reference panics on such an index): resident circuit structure for MZK_WITNESS_*_VECTOR. */
MZK_API int32_t mzk_prover_set_wire_variables(uint64_t prover, const uint32_t* wire_variables, uint64_t n_vars);
#define MZK_WITNESS_DEV_WIRES 0   /* device, W x n: witness[wire_variable(i, j)] already gathered.  The prover keeps the POINTER, not a copy:
                                   * the buffer must stay valid and unmodified until round 2 (round 2.5 for UltraPlonk) has returned --
                                   * rounds 1.5 and 2 read the wire values again (sorted vector, permutation product) */
#define MZK_WITNESS_HOST_WIRES 1  /* host, W x n (page-locked memory: column k + 1 crosses PCIe under the iNTT of column k) */
#define MZK_WITNESS_HOST_VECTOR 2 /* host, n_vars witness values; gathered on the device (mzk_prover_set_wire_variables) */
#define MZK_WITNESS_DEV_VECTOR 3  /* device, n_vars witness values */
/* Round 1: compute_wire_polynomials (gather + W iNTTs), compute_pub_input_polynomial, mask_polynomial, batch_commit.
 * witness_len: W x n or n_vars elements.  Public input: n_pub values; pub_input_rows = NULL places them on rows 0 .. n_pub - 1 (where
 * finalize_for_arithmetization puts the IO gates), else row pub_input_rows[i] < n.  blinders_mont: W x 2 (`DensePolynomial::rand(1)` per
 * wire, in wire order).  out_comms_xy: W affine points. */
MZK_API int32_t mzk_prover_round1(uint64_t prover, int32_t witness_kind, const void* witness, uint64_t witness_len, const uint64_t* pub_input_rows,
                                  const uint64_t* pub_input_mont, uint64_t n_pub, const uint64_t* blinders_mont, uint64_t* out_comms_xy);
/* UltraPlonk only.  blinders: 2 x 3 (h_1 then h_2); out: 2 points.  MZK_ERR_LOOKUP when a looked-up value is not in the table. */
MZK_API int32_t mzk_prover_round1_5(uint64_t prover, const uint64_t* tau_mont, const uint64_t* blinders_mont, uint64_t* out_comms_xy);
/* blinders: 3; out: 1 point. */
MZK_API int32_t mzk_prover_round2(uint64_t prover, const uint64_t* beta_mont, const uint64_t* gamma_mont, const uint64_t* blinders_mont,
                                  uint64_t* out_comm_xy);
/* UltraPlonk only.  blinders: 3; out: 1 point. */
MZK_API int32_t mzk_prover_round2_5(uint64_t prover, const uint64_t* blinders_mont, uint64_t* out_comm_xy);
/* provers: the n_instances handles in instance order (all past round 2 / 2.5).  blinders: W - 1 (split_quotient_polynomial's draws,
 * prover.rs:947-955).  out: W points. */
MZK_API int32_t mzk_prover_round3(const uint64_t* provers, uint32_t n_instances, const uint64_t* alpha_mont, const uint64_t* blinders_mont,
                                  uint64_t* out_comms_xy);
/* out_evals_mont: wires_evals (W), wire_sigma_evals (W - 1), perm_next_eval (1), then for UltraPlonk the 15 PlookupEvaluations in
 * their declaration order (structs.rs:496-541): range_table, key_table, table_dom_sep, q_dom_sep, h_1, q_lookup, prod_next,
 * range_table_next, key_table_next, table_dom_sep_next, h_1_next, h_2_next, q_lookup_next, w_3_next, w_4_next. */
MZK_API int32_t mzk_prover_round4(uint64_t prover, const uint64_t* zeta_mont, uint64_t* out_evals_mont);
/* out: opening_proof, shifted_opening_proof (2 points).  MZK_ERR_WRONG_QUOTIENT_DEGREE: see above. */
MZK_API int32_t mzk_prover_round5(const uint64_t* provers, uint32_t n_instances, const uint64_t* v_mont, uint64_t* out_comms_xy);
/* Multi-device hosts in one process: this rank's class-remainder buffer (device pointer, bytes), and the same buffers of all `world`
 * ranks with their logical devices, for the one device-to-device exchange of round 3 (mzk_comm.exchange_classes == NULL). */
MZK_API int32_t mzk_prover_exchange_buffer(uint64_t prover, void** out_dptr, uint64_t* out_bytes);
MZK_API int32_t mzk_prover_set_peer_buffers(uint64_t prover, void* const* peer_dptrs, const int32_t* peer_devices);
/* A device-resident polynomial of the proof in flight (valid until the next mzk_prover_round1 on this handle): which = 0 .. W - 1 the
 * masked wire polynomials (n + 2 coefficients; wire 0 is the `linking_wire_poly` of prove_with_link_hint, snark.rs:81-119), W the
 * permutation product (n + 3). */
MZK_API int32_t mzk_prover_poly_dev(uint64_t prover, uint32_t which, const void** out_dptr, uint64_t* out_len);
/* on != 0: every round synchronises the device at its internal stage boundaries and records wall milliseconds; mzk_prover_timings writes
 * them as one JSON object ("r1_ntt_mask", "r1_commit", .. "r5_commit") into buf (NUL-terminated, truncated to cap). */
MZK_API int32_t mzk_prover_profile(uint64_t prover, int32_t on);
MZK_API int32_t mzk_prover_timings(uint64_t prover, char* buf, uint64_t cap);
/* Bytes of HBM this prover holds: coefficient forms, resident class evaluations (proving key), per-proof workspace. */
MZK_API int32_t mzk_prover_hbm_bytes(uint64_t prover, uint64_t* out_fixed, uint64_t* out_proving_key, uint64_t* out_workspace);

/* ---- page-locked host memory for the host-pointer entry points (mzk_ntt, mzk_ntt_batch, mzk_msm, mzk_msm_batch) ----
 * A shim that swaps only the two third-party call sites moves every operand over PCIe.  From memory
 * obtained here (hipHostMalloc), or registered in place (hipHostRegister: worth it for long-lived buffers only), the transfers
 * are DMA at link rate and asynchronous, so that inside a batch call -- and between concurrent callers, the reference's Rayon
 * `par_iter`s (plonk/src/proof_system/prover.rs:552-562, primitives/src/pcs/univariate_kzg/mod.rs:125-127) -- the upload of
 * polynomial k+1 and the download of k-1 overlap the transform of k.  Pageable memory works too, staged by the runtime. */
MZK_API int32_t mzk_host_alloc(uint64_t bytes, void** out_ptr);
MZK_API int32_t mzk_host_free(void* ptr);
MZK_API int32_t mzk_host_register(void* ptr, uint64_t bytes);
MZK_API int32_t mzk_host_unregister(void* ptr);

/* ---- device memory helpers for bindings without HIP of their own ---- */
MZK_API int32_t mzk_dev_alloc(uint64_t bytes, void** out_dptr);
MZK_API int32_t mzk_dev_free(void* dptr);
MZK_API int32_t mzk_dev_upload(void* dptr, const void* host, uint64_t bytes);
MZK_API int32_t mzk_dev_download(void* host, const void* dptr, uint64_t bytes);
MZK_API int32_t mzk_dev_sync(void);
/* Non-blocking streams of the calling thread's device and asynchronous transfers on them (page-locked host memory: mzk_host_alloc),
 * for a host that overlaps the upload of witness column k + 1 with the iNTT of column k (the reference gathers the witness on the
 * host, relation/src/constraint_system.rs:1225-1247).  mzk_stream_wait_stream(waiter, signaller): work enqueued on `waiter` after
 * this call starts once everything enqueued on `signaller` BEFORE this call has completed (NULL = the null stream). */
MZK_API int32_t mzk_stream_create(void** out_stream);
MZK_API int32_t mzk_stream_destroy(void* stream);
MZK_API int32_t mzk_stream_sync(void* stream);
MZK_API int32_t mzk_stream_wait_stream(void* waiter, void* signaller);
MZK_API int32_t mzk_dev_upload_async(void* dptr, const void* host, uint64_t bytes, void* stream);
MZK_API int32_t mzk_dev_download_async(void* host, const void* dptr, uint64_t bytes, void* stream);
/* device-to-device copy, 2-D copy (pitches and width in bytes) and byte fill, asynchronous on `stream`: what a host
 * orchestrating device-resident rounds needs between kernels (mpc-jellyfish_amd/host/mzk_host.hpp). */
MZK_API int32_t mzk_dev_copy(void* dst, const void* src, uint64_t bytes, void* stream);
/* bytes from device `src_device` to device `dst_device` (logical indices, both initialised), asynchronous on `stream`, a stream of
 * the SOURCE device (NULL = its null stream): hipMemcpyPeerAsync -- over xGMI when the devices can access each other, which
 * mzk_init enables -- or a plain device-to-device copy when both are virtual devices of one card.  The one exchange of a
 * multi-GPU proof (class remainders of the quotient) goes through here. */
MZK_API int32_t mzk_dev_copy_peer(void* dst, int32_t dst_device, const void* src, int32_t src_device, uint64_t bytes, void* stream);
MZK_API int32_t mzk_dev_copy2d(void* dst, uint64_t dst_pitch, const void* src, uint64_t src_pitch, uint64_t width, uint64_t height, void* stream);
MZK_API int32_t mzk_dev_memset(void* dptr, int32_t value, uint64_t bytes, void* stream);
/* `height` runs of `width` bytes, `pitch` bytes apart, in one launch (the tails of the rows of a coefficient slab) */
MZK_API int32_t mzk_dev_memset2d(void* dptr, uint64_t pitch, int32_t value, uint64_t width, uint64_t height, void* stream);

/* ---- measurement hooks (bench.py): HIP-event timing of the library's own kernels ---- */
/* on != 0: every subsequent call brackets its dominant kernels with hipEvents on the launch stream. */
MZK_API int32_t mzk_profile_enable(int32_t on);
/* name: "msm_accumulate", "msm_total", "ntt_pass", "ntt_total".  Returns accumulated device
 * milliseconds and launch count since the last reset (synchronises the recorded events). */
MZK_API int32_t mzk_profile_get(const char* name, double* out_ms, uint64_t* out_count);
MZK_API int32_t mzk_profile_reset(void);
/* MSMs of >= 2^17 pairs over a BLS12-381 SRS use a table of precomputed multiples 2^(c*w) * P_i built in HBM on
 * the first such call (13 x the SRS size for 2^20 points); on != 0 (default) enables it.  Results are identical. */
MZK_API int32_t mzk_msm_set_precompute(int32_t on);
/* Builds the table of precomputed multiples of an SRS now (instead of lazily on its first MSM of >= 1024 pairs) and reports it:
 * window bits c, levels W = ceil(256 / c), bytes of HBM it occupies (W x n x 112 B for BLS12-381, x 72 B for BN254: 2 x 9 words of 29-bit limbs) and the wall
 * time the build took (W - 1 launches of c doublings per point, synchronised).  A fixed-base table is legitimate for KZG -- the
 * commit key never changes (primitives/src/pcs/univariate_kzg/srs.rs:77-93) -- but it is a set-up cost ark-ec's
 * VariableBaseMSM does not pay: bench.py prints it beside the headline.  All zeros when the table is disabled or did not fit:
 * a table is only built while it takes at most half of the free HBM, and never beyond MZK_MSM_TABLE_BUDGET bytes (environment; unset = no
 * budget) -- such an SRS commits on the plain (variable-base) path, same points. */
MZK_API int32_t mzk_srs_precompute(uint64_t srs_handle, uint32_t* out_window_bits, uint32_t* out_levels, uint64_t* out_table_bytes,
                                   double* out_build_ms);
/* HBM accounting (bench.py reports it per leg).  An SRS: its points (boundary form + the MSM's internal form: 96 + 112 B per point on
 * BLS12-381, 64 + 72 B on BN254) and its fixed-base table (0 until built; the reference keeps the points only: srs.rs:36-40).  A proving
 * key: the resident evaluations of the fixed polynomials and the per-point tables.  The device context: the shared scratch of the NTT /
 * MSM / quotient kernels (grow-only) and the buffers of the host-pointer I/O slots. */
MZK_API int32_t mzk_srs_hbm_bytes(uint64_t srs_handle, uint64_t* out_points_bytes, uint64_t* out_table_bytes);
MZK_API int32_t mzk_plonk_pk_hbm_bytes(uint64_t pk_handle, uint64_t* out_bytes);
MZK_API int32_t mzk_workspace_hbm_bytes(uint64_t* out_bytes);
/* Frees that scratch (synchronises the device; it grows again on demand). */
MZK_API int32_t mzk_workspace_release(void);
/* Kernel launches the library has made in this process so far (all device contexts; copies and fills are not kernels): bench.py
 * reports launches per proof and per MSM from differences of it.  out may not be null. */
MZK_API int32_t mzk_launch_count(uint64_t* out_launches);
/* Last MSM's shape: window bits, windows, buckets per window (for DESIGN.md's op counts). */
MZK_API int32_t mzk_msm_last_shape(uint32_t* out_window_bits, uint32_t* out_windows, uint32_t* out_buckets);

#ifdef __cplusplus
}
#endif
#endif /* MZK_H */
