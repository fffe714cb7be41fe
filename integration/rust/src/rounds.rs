//! Round-level swap: the BODIES of `Prover::run_1st_round .. compute_opening_proofs`
//! (plonk/src/proof_system/prover.rs:72-419) replaced by `mzk_prover_round1 .. round5` (include/mzk.h; csrc/prover.hip).
//! `batch_prove_internal` (plonk/src/proof_system/snark.rs:201-469) keeps its shape: the same loops over the instances, the same
//! transcript calls between the rounds, the same `prng` draws in the same order -- so `PlonkKzgSnark::prove`'s signature and the
//! proof bytes are unchanged.  What changes is where the polynomials live: `Oracles` (structs.rs:875-887) is replaced by one
//! `Mi355Prover` per instance whose vectors stay in HBM from the wire iNTTs to the opening MSMs.
//!
//! This file is meant to live in the reference's `plonk` crate as `src/proof_system/mi355_rounds.rs` behind
//! `#[cfg(feature = "mi355")]`; it needs one accessor added to `relation` (the fields are `pub(crate)` there,
//! relation/src/constraint_system.rs:127, 131):
//!     impl<F: FftField> PlonkCircuit<F> { pub fn witness_and_wire_variables(&self) -> (&[F], &[Vec<Variable>; GATE_WIDTH + 2]) }
//! NOT COMPILED in this repository's image (no Rust toolchain); the C++ host (mpc-jellyfish_amd/host/mzk_prover.hpp) and the ctypes
//! driver (mpc-jellyfish_amd/prover.py) are the compiled / executed twins of exactly this call sequence.
use ark_ec::{pairing::Pairing, short_weierstrass::Affine, AffineRepr};
use ark_ff::{PrimeField, UniformRand};
use ark_std::rand::{CryptoRng, RngCore};
use core::ffi::c_void;

use crate::{check, mzk_srs_release, Mi355Error, SrsHandle};

pub const MZK_WITNESS_HOST_VECTOR: i32 = 2;
pub const MZK_ERR_WRONG_QUOTIENT_DEGREE: i32 = -9;

#[cfg_attr(feature = "link", link(name = "mi355zk"))]
extern "C" {
    pub fn mzk_srs_lagrange_from_srs(srs: u64, log_n: u32, n_extra: u32, out_handle: *mut u64) -> i32;
    pub fn mzk_prover_create(curve_id: i32, log_n: u32, num_wire_types: u32, selector_coeffs: *const u64, sigma_coeffs: *const u64,
                             table_coeffs: *const u64, poly_len: u64, k_mont: *const u64, commit_key: u64, lagrange_key: u64,
                             comm: *const c_void, out_prover: *mut u64) -> i32;
    pub fn mzk_prover_destroy(prover: u64) -> i32;
    pub fn mzk_prover_set_wire_variables(prover: u64, wire_variables: *const u32, n_vars: u64) -> i32;
    pub fn mzk_prover_round1(prover: u64, witness_kind: i32, witness: *const c_void, witness_len: u64, pub_input_rows: *const u64,
                             pub_input_mont: *const u64, n_pub: u64, blinders_mont: *const u64, out_comms_xy: *mut u64) -> i32;
    pub fn mzk_prover_round1_5(prover: u64, tau: *const u64, blinders: *const u64, out_comms_xy: *mut u64) -> i32;
    pub fn mzk_prover_round2(prover: u64, beta: *const u64, gamma: *const u64, blinders: *const u64, out_comm_xy: *mut u64) -> i32;
    pub fn mzk_prover_round2_5(prover: u64, blinders: *const u64, out_comm_xy: *mut u64) -> i32;
    pub fn mzk_prover_round3(provers: *const u64, n_instances: u32, alpha: *const u64, blinders: *const u64, out_comms_xy: *mut u64) -> i32;
    pub fn mzk_prover_round4(prover: u64, zeta: *const u64, out_evals: *mut u64) -> i32;
    pub fn mzk_prover_round5(provers: *const u64, n_instances: u32, v: *const u64, out_comms_xy: *mut u64) -> i32;
}

/// One instance's device-resident state: `ProvingKey{selectors, sigmas, plookup_pk}` (structs.rs:575-590) uploaded once, the
/// circuit's wire-variable table, and the workspace of the proof in flight.  Created once per (proving key, instance slot).
pub struct Mi355Prover {
    pub handle: u64,
    pub num_wire_types: usize,
    pub ultra: bool,
    fq_limbs: usize,
    /// the Lagrange-basis key derived for this prover (0 = none): its points and fixed-base table live in HBM until released
    lagrange_key: u64,
}

impl Drop for Mi355Prover {
    fn drop(&mut self) {
        unsafe {
            mzk_prover_destroy(self.handle);
            if self.lagrange_key != 0 {
                mzk_srs_release(self.lagrange_key);
            }
        }
    }
}

fn flat<F: PrimeField>(v: &[F]) -> *const u64 {
    v.as_ptr() as *const u64 // Fp<MontBackend<_, 4>, 4> is a [u64; 4] newtype in Montgomery form: the byte image the ABI takes
}

impl Mi355Prover {
    /// `selectors` / `sigmas` / `tables`: the coefficient vectors of the proving key's polynomials, each padded to n = 2^log_n and
    /// concatenated (what `pk.selectors.iter().flat_map(|p| padded(p.coeffs()))` yields); `commit_key`: the registered
    /// `pk.commit_key.powers_of_g` (n + 3 points); `wire_variables`: W x n variable indices, wire-major.
    #[allow(clippy::too_many_arguments)]
    pub fn new<F: PrimeField>(curve_id: i32, log_n: u32, selectors: &[F], sigmas: &[F], tables: Option<&[F]>, k: &[F], commit_key: &SrsHandle,
                              lagrange_round1: bool, wire_variables: &[u32], n_vars: usize, fq_limbs: usize) -> Result<Self, Mi355Error> {
        let num_wire_types = k.len();
        let (mut lagrange, mut handle) = (0u64, 0u64);
        unsafe {
            if lagrange_round1 {
                // [L_i(beta)]g from the SRS's own points (inverse NTT over the group, once per SRS and domain): round 1 then commits the
                // wire VALUES -- same group elements, mostly small scalars
                check(mzk_srs_lagrange_from_srs(commit_key.raw(), log_n, 3, &mut lagrange))?;
            }
            let made = check(mzk_prover_create(curve_id, log_n, num_wire_types as u32, flat(selectors), flat(sigmas),
                                               tables.map_or(core::ptr::null(), |t| flat(t)), 1u64 << log_n, flat(k), commit_key.raw(), lagrange,
                                               core::ptr::null(), &mut handle));
            if made.is_err() && lagrange != 0 {
                mzk_srs_release(lagrange);                         // nothing of a failed construction stays in HBM
            }
            made?;
        }
        // from here on Drop releases both handles
        let me = Self { handle, num_wire_types, ultra: tables.is_some(), fq_limbs, lagrange_key: lagrange };
        unsafe { check(mzk_prover_set_wire_variables(handle, wire_variables.as_ptr(), n_vars as u64))? };
        Ok(me)
    }

    fn points<P: ark_ec::short_weierstrass::SWCurveConfig>(&self, xy: &[u64]) -> Vec<Affine<P>>
    where P::BaseField: PrimeField {
        xy.chunks(2 * self.fq_limbs).map(|c| crate::affine_from_limbs::<P>(c)).collect() // (0, 0) = infinity
    }
}

/// `DensePolynomial::rand(hiding_bound, prng)` draws hiding_bound + 1 coefficients, low order first (ark-poly): these ARE the
/// blinders b_0 .. b_h of `mask_polynomial` (prover.rs:463-486): p + (b_0 + b_1 X + ..)(X^n - 1).
fn draw<F: PrimeField, R: RngCore + CryptoRng>(prng: &mut R, count: usize) -> Vec<F> {
    (0..count).map(|_| F::rand(prng)).collect()
}

/// The round calls of `batch_prove_internal` (snark.rs:263-431) with the device prover.  `T`, `E`, `circuits`, `prove_keys`,
/// `transcript`, `challenges` as in the reference; `provers[i]` belongs to `(prove_keys[i], circuits[i])`.
/// Returns what the reference assembles into `BatchProof` (snark.rs:453-462).
#[cfg(feature = "reference-types")]
pub fn prove_rounds<E, F, P, C, R, T>(prng: &mut R, circuits: &[&C], provers: &[&Mi355Prover], transcript: &mut T)
    -> Result<BatchProof<E>, PlonkError>
where
    E: Pairing<BaseField = F, G1Affine = Affine<P>>, F: RescueParameter + SWToTEConParam, P: SWCurveConfig<BaseField = F>,
    C: Arithmetization<E::ScalarField>, R: CryptoRng + RngCore, T: PlonkTranscript<F>,
{
    let w = provers[0].num_wire_types;
    let ql = provers[0].fq_limbs;
    let handles: Vec<u64> = provers.iter().map(|p| p.handle).collect();
    let fail = |rc: i32| -> PlonkError {
        if rc == MZK_ERR_WRONG_QUOTIENT_DEGREE { SnarkError::WrongQuotientPolyDegree(0, 0).into() } else { SnarkError::ParameterError(crate::last_error()).into() }
    };
    // Round 1 (prover.rs:72-87): the witness vector crosses PCIe once; gather, iNTTs, masking, batch_commit on the device
    let mut wires_poly_comms_vec = vec![];
    for (cs, p) in circuits.iter().zip(provers) {
        let (witness, _) = cs.witness_and_wire_variables();
        let pub_input = cs.public_input()?;                       // rows 0 .. num_inputs - 1 after finalize_for_arithmetization
        let blinders: Vec<E::ScalarField> = draw(prng, 2 * w);    // W x DensePolynomial::rand(1), in wire order
        let mut xy = vec![0u64; w * 2 * ql];
        let rc = unsafe { mzk_prover_round1(p.handle, MZK_WITNESS_HOST_VECTOR, witness.as_ptr() as *const c_void, witness.len() as u64,
                                            core::ptr::null(), flat(&pub_input), pub_input.len() as u64, flat(&blinders), xy.as_mut_ptr()) };
        if rc != 0 { return Err(fail(rc)); }
        let comms: Vec<Commitment<E>> = p.points::<P>(&xy).into_iter().map(Commitment).collect();
        transcript.append_commitments(b"witness_poly_comms", &comms)?;
        wires_poly_comms_vec.push(comms);
    }
    // Round 1.5 (prover.rs:89-118)
    let tau = transcript.get_and_append_challenge::<E>(b"tau")?;
    let mut h_poly_comms_vec = vec![];
    for p in provers {
        h_poly_comms_vec.push(if p.ultra {
            let blinders: Vec<E::ScalarField> = draw(prng, 6);    // h_1, h_2: DensePolynomial::rand(2) each
            let mut xy = vec![0u64; 2 * 2 * ql];
            let rc = unsafe { mzk_prover_round1_5(p.handle, flat(&[tau]), flat(&blinders), xy.as_mut_ptr()) };
            if rc != 0 { return Err(fail(rc)); }
            let comms: Vec<Commitment<E>> = p.points::<P>(&xy).into_iter().map(Commitment).collect();
            transcript.append_commitments(b"h_poly_comms", &comms)?;
            Some(comms)
        } else { None });
    }
    // Round 2 (prover.rs:125-141)
    let beta = transcript.get_and_append_challenge::<E>(b"beta")?;
    let gamma = transcript.get_and_append_challenge::<E>(b"gamma")?;
    let mut prod_perm_poly_comms_vec = vec![];
    for p in provers {
        let blinders: Vec<E::ScalarField> = draw(prng, 3);
        let mut xy = vec![0u64; 2 * ql];
        let rc = unsafe { mzk_prover_round2(p.handle, flat(&[beta]), flat(&[gamma]), flat(&blinders), xy.as_mut_ptr()) };
        if rc != 0 { return Err(fail(rc)); }
        let comm = Commitment(p.points::<P>(&xy)[0]);
        transcript.append_commitment(b"perm_poly_comms", &comm)?;
        prod_perm_poly_comms_vec.push(comm);
    }
    // Round 2.5 (prover.rs:143-183)
    let mut prod_lookup_poly_comms_vec = vec![];
    for p in provers {
        prod_lookup_poly_comms_vec.push(if p.ultra {
            let blinders: Vec<E::ScalarField> = draw(prng, 3);
            let mut xy = vec![0u64; 2 * ql];
            let rc = unsafe { mzk_prover_round2_5(p.handle, flat(&blinders), xy.as_mut_ptr()) };
            if rc != 0 { return Err(fail(rc)); }
            let comm = Commitment(p.points::<P>(&xy)[0]);
            transcript.append_commitment(b"plookup_poly_comms", &comm)?;
            Some(comm)
        } else { None });
    }
    // Round 3 (prover.rs:192-209): ONE call over all instances -- quotient, split (W - 1 draws), W commitments
    let alpha = transcript.get_and_append_challenge::<E>(b"alpha")?;
    let blinders: Vec<E::ScalarField> = draw(prng, w - 1);
    let mut xy = vec![0u64; w * 2 * ql];
    let rc = unsafe { mzk_prover_round3(handles.as_ptr(), handles.len() as u32, flat(&[alpha]), flat(&blinders), xy.as_mut_ptr()) };
    if rc != 0 { return Err(fail(rc)); }
    let split_quot_poly_comms: Vec<Commitment<E>> = provers[0].points::<P>(&xy).into_iter().map(Commitment).collect();
    transcript.append_commitments(b"quot_poly_comms", &split_quot_poly_comms)?;
    // Rounds 4 / 4.5 (prover.rs:216-299): ProofEvaluations of every instance first, then the PlookupEvaluations (snark.rs:365-399)
    let zeta = transcript.get_and_append_challenge::<E>(b"zeta")?;
    let (mut poly_evals_vec, mut plookup_evals_vec) = (vec![], vec![]);
    for p in provers {
        let mut ev = vec![E::ScalarField::zero(); 2 * w + if p.ultra { 15 } else { 0 }];
        let rc = unsafe { mzk_prover_round4(p.handle, flat(&[zeta]), ev.as_mut_ptr() as *mut u64) };
        if rc != 0 { return Err(fail(rc)); }
        let poly_evals = ProofEvaluations { wires_evals: ev[..w].to_vec(), wire_sigma_evals: ev[w..2 * w - 1].to_vec(), perm_next_eval: ev[2 * w - 1] };
        transcript.append_proof_evaluations::<E>(&poly_evals)?;
        poly_evals_vec.push(poly_evals);
        plookup_evals_vec.push(if p.ultra { Some(plookup_evaluations_from_slice(&ev[2 * w..])) } else { None });   // declaration order, structs.rs:496-541
    }
    for evals in plookup_evals_vec.iter().flatten() {
        transcript.append_plookup_evaluations::<E>(evals)?;
    }
    // Round 5 (prover.rs:302-460): linearisation polynomial and both opening proofs, ONE call.  MZK_ERR_WRONG_QUOTIENT_DEGREE here
    // is the quotient identity failing at zeta: the witness does not satisfy the circuit (the reference raises it in round 3)
    let v = transcript.get_and_append_challenge::<E>(b"v")?;
    let mut xy = vec![0u64; 2 * 2 * ql];
    let rc = unsafe { mzk_prover_round5(handles.as_ptr(), handles.len() as u32, flat(&[v]), xy.as_mut_ptr()) };
    if rc != 0 { return Err(fail(rc)); }
    let open = provers[0].points::<P>(&xy);
    let plookup_proofs_vec = (0..provers.len()).map(|i| plookup_evals_vec[i].clone().map(|poly_evals| PlookupProof {
        h_poly_comms: h_poly_comms_vec[i].clone().unwrap(), prod_lookup_poly_comm: prod_lookup_poly_comms_vec[i].unwrap(), poly_evals })).collect();
    Ok(BatchProof { wires_poly_comms_vec, prod_perm_poly_comms_vec, poly_evals_vec, plookup_proofs_vec, split_quot_poly_comms,
                    opening_proof: Commitment(open[0]), shifted_opening_proof: Commitment(open[1]) })
}
