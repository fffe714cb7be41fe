//! `mi355` -- Rust binding of libmi355zk (C ABI: `include/mzk.h`) for the jf-plonk prover.
//!
//! The reference has no FFI of its own: its hot path sits behind two third-party trait surfaces,
//! `ark_ec::VariableBaseMSM::msm_bigint` (primitives/src/pcs/univariate_kzg/mod.rs:109-111, 151-155) and
//! `ark_poly::EvaluationDomain::{fft, ifft}` (plonk/src/proof_system/prover.rs:545-567, 672;
//! relation/src/constraint_system.rs:1162-1259).  The wrappers below are what those call sites call instead
//! under `#[cfg(feature = "mi355")]`: same inputs, same outputs, proof bytes unchanged.
//!
//! Memory images: `Fp<MontBackend<_, N>, N>` is a `BigInt<N>` = `[u64; N]` newtype holding a*R mod p, so a slice of
//! field elements IS the byte image the ABI takes; affine points are repacked x||y once per SRS because rustc does not
//! fix the field order of `Affine { x, y, infinity }`.
//!
//! NOT COMPILED in the build image of this repository (no Rust toolchain there): kept as source so that a maintainer
//! with cargo can build it; the same ABI is exercised end to end by the C++ host (`mpc-jellyfish_amd/host/`) and by the
//! Python mirror's ctypes bindings, which is what the parity tests drive.
#![allow(clippy::missing_safety_doc)]

use ark_ec::{pairing::Pairing, short_weierstrass::{Affine, Projective, SWCurveConfig}, AffineRepr};
use ark_ff::{BigInteger, Field, PrimeField, Zero};
use ark_std::vec::Vec;
use core::ffi::{c_char, c_void, CStr};

pub const CURVE_BLS12_381: i32 = 0;
pub const CURVE_BN254: i32 = 1;

#[cfg_attr(feature = "link", link(name = "mi355zk"))]
extern "C" {
    pub fn mzk_init(device: i32) -> i32;
    pub fn mzk_last_error() -> *const c_char;
    pub fn mzk_srs_register(curve_id: i32, xy_mont: *const u64, n_points: u64, out_handle: *mut u64) -> i32;
    pub fn mzk_srs_release(handle: u64) -> i32;
    pub fn mzk_srs_precompute(handle: u64, out_window_bits: *mut u32, out_levels: *mut u32, out_table_bytes: *mut u64, out_build_ms: *mut f64) -> i32;
    pub fn mzk_msm(srs: u64, base_offset: u64, scalars: *const u64, n: u64, scalars_are_mont: i32, out_xyz_mont: *mut u64) -> i32;
    pub fn mzk_msm_batch(srs: u64, n_polys: u32, scalars: *const *const u64, lens: *const u64, base_offsets: *const u64,
                         scalars_are_mont: i32, out_xyz_mont: *mut u64) -> i32;
    pub fn mzk_ntt(curve_id: i32, data_mont: *mut u64, in_len: u64, log_n: u32, inverse: i32, coset_offset_mont: *const u64) -> i32;
    pub fn mzk_ntt_batch(curve_id: i32, n_polys: u32, data_mont: *const *mut u64, in_lens: *const u64, log_n: u32, inverse: i32,
                         coset_offset_mont: *const u64) -> i32;
    pub fn mzk_host_alloc(bytes: u64, out_ptr: *mut *mut c_void) -> i32;
    pub fn mzk_host_free(ptr: *mut c_void) -> i32;
}

#[derive(Debug)]
pub struct Mi355Error(pub i32, pub String);

fn check(rc: i32) -> Result<(), Mi355Error> {
    if rc == 0 {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(mzk_last_error()) }.to_string_lossy().into_owned();
    Err(Mi355Error(rc, msg))
}

/// The registered copy of `UnivariateProverParam::powers_of_g` (primitives/src/pcs/univariate_kzg/srs.rs:36-40), kept
/// beside it for the life of the proving key.
pub struct SrsHandle {
    handle: u64,
    pub len: usize,
}

impl SrsHandle {
    /// the library's handle (for the round-level entry points: rounds.rs)
    pub fn raw(&self) -> u64 {
        self.handle
    }
}

/// text of the calling thread's last library error
pub fn last_error() -> String {
    unsafe { CStr::from_ptr(mzk_last_error()) }.to_string_lossy().into_owned()
}

/// x||y Montgomery limbs -> affine point; (0, 0) encodes infinity (`Commitment(G1Affine)`, pcs/structs.rs:16-19)
pub fn affine_from_limbs<P: SWCurveConfig>(xy: &[u64]) -> Affine<P>
where
    P::BaseField: PrimeField,
{
    let limbs = xy.len() / 2;
    if xy.iter().all(|w| *w == 0) {
        return Affine::<P>::identity();
    }
    let f = |w: &[u64]| {
        let mut v = P::BaseField::zero();
        unsafe { core::ptr::copy_nonoverlapping(w.as_ptr(), &mut v as *mut P::BaseField as *mut u64, limbs) };
        v
    };
    Affine::<P>::new_unchecked(f(&xy[..limbs]), f(&xy[limbs..]))
}

/// The round-level swap: the bodies of Prover::run_1st_round .. compute_opening_proofs behind mzk_prover_* (INTEGRATION.md section 2b).
pub mod rounds;

impl Drop for SrsHandle {
    fn drop(&mut self) {
        unsafe { mzk_srs_release(self.handle) };
    }
}

/// Repack `powers_of_g` as x||y Montgomery limbs and register it on the device; also builds the fixed-base table now rather
/// than inside the first commitment.
pub fn register_srs<P: SWCurveConfig>(powers_of_g: &[Affine<P>], curve_id: i32) -> Result<SrsHandle, Mi355Error>
where
    P::BaseField: PrimeField,
{
    let limbs = <<P::BaseField as PrimeField>::BigInt as BigInteger>::NUM_LIMBS;        // 6 (BLS12-381) / 4 (BN254)
    let mut packed = vec![0u64; powers_of_g.len() * 2 * limbs];
    for (i, p) in powers_of_g.iter().enumerate() {
        if let Some((x, y)) = p.xy() {
            // the Montgomery image, NOT into_bigint(): Fp's inner BigInt is what sits in memory
            packed[2 * i * limbs..(2 * i + 1) * limbs].copy_from_slice(mont_limbs(x));
            packed[(2 * i + 1) * limbs..(2 * i + 2) * limbs].copy_from_slice(mont_limbs(y));
        }                                                                                // infinity stays (0, 0)
    }
    let mut handle = 0u64;
    unsafe {
        check(mzk_init(-1))?;
        check(mzk_srs_register(curve_id, packed.as_ptr(), powers_of_g.len() as u64, &mut handle))?;
        check(mzk_srs_precompute(handle, core::ptr::null_mut(), core::ptr::null_mut(), core::ptr::null_mut(), core::ptr::null_mut()))?;
    }
    Ok(SrsHandle { handle, len: powers_of_g.len() })
}

/// The in-memory limbs of a prime-field element (a*R mod p).  `Fp<MontBackend<_, N>, N>` is `#[repr(transparent)]`-like
/// over `BigInt<N>` = `[u64; N]`; the slice view below is that image.
fn mont_limbs<F: PrimeField>(x: &F) -> &[u64] {
    let n = <F::BigInt as BigInteger>::NUM_LIMBS;
    unsafe { core::slice::from_raw_parts(x as *const F as *const u64, n) }
}

/// `E::G1::msm_bigint(&powers_of_g[skip..], &coeffs.into_bigint())` (univariate_kzg/mod.rs:106-112): the Montgomery
/// coefficients go across as they sit in the `DensePolynomial`; the device converts.  Returns the Jacobian point.
pub fn msm<P: SWCurveConfig, F: PrimeField>(srs: &SrsHandle, skip: usize, coeffs: &[F]) -> Result<Projective<P>, Mi355Error>
where
    P::BaseField: PrimeField,
{
    let limbs = <<P::BaseField as PrimeField>::BigInt as BigInteger>::NUM_LIMBS;
    let mut xyz = vec![0u64; 3 * limbs];
    check(unsafe { mzk_msm(srs.handle, skip as u64, coeffs.as_ptr() as *const u64, coeffs.len() as u64, 1, xyz.as_mut_ptr()) })?;
    Ok(jacobian_from_limbs::<P>(&xyz, limbs))
}

/// `batch_commit` (univariate_kzg/mod.rs:119-131) in one call: sort, accumulate and bucket reduction fused over the polynomials.
pub fn msm_batch<P: SWCurveConfig, F: PrimeField>(srs: &SrsHandle, polys: &[&[F]]) -> Result<Vec<Projective<P>>, Mi355Error>
where
    P::BaseField: PrimeField,
{
    let limbs = <<P::BaseField as PrimeField>::BigInt as BigInteger>::NUM_LIMBS;
    // skip_leading_zeros_and_convert_to_bigints (mod.rs:379-395): leading zero coefficients shift the bases instead
    let skips: Vec<u64> = polys.iter().map(|p| p.iter().take_while(|c| c.is_zero()).count() as u64).collect();
    let ptrs: Vec<*const u64> = polys.iter().zip(&skips).map(|(p, s)| p[*s as usize..].as_ptr() as *const u64).collect();
    let lens: Vec<u64> = polys.iter().zip(&skips).map(|(p, s)| (p.len() as u64) - s).collect();
    let mut xyz = vec![0u64; polys.len() * 3 * limbs];
    check(unsafe { mzk_msm_batch(srs.handle, polys.len() as u32, ptrs.as_ptr(), lens.as_ptr(), skips.as_ptr(), 1, xyz.as_mut_ptr()) })?;
    Ok(xyz.chunks(3 * limbs).map(|c| jacobian_from_limbs::<P>(c, limbs)).collect())
}

fn jacobian_from_limbs<P: SWCurveConfig>(xyz: &[u64], limbs: usize) -> Projective<P>
where
    P::BaseField: PrimeField,
{
    let f = |w: &[u64]| {
        // the library returns Montgomery limbs: rebuild the element from its in-memory image
        let mut v = P::BaseField::zero();
        unsafe { core::ptr::copy_nonoverlapping(w.as_ptr(), &mut v as *mut P::BaseField as *mut u64, limbs) };
        v
    };
    Projective::<P>::new_unchecked(f(&xyz[..limbs]), f(&xyz[limbs..2 * limbs]), f(&xyz[2 * limbs..]))     // Z = 0 encodes infinity
}

/// `domain.fft_in_place(&mut buf)` / `coset.fft_in_place` (prover.rs:552-567): `buf.len()` = domain size, the first
/// `in_len` entries hold the coefficients; offset = `domain.coset_offset()` (None when it is one).
pub fn fft_in_place<F: PrimeField>(curve_id: i32, buf: &mut [F], in_len: usize, offset: Option<&F>) -> Result<(), Mi355Error> {
    ntt(curve_id, buf, in_len, false, offset)
}

/// `domain.ifft_in_place(&mut buf)` (constraint_system.rs:1172 and on; prover.rs:672 with the coset offset)
pub fn ifft_in_place<F: PrimeField>(curve_id: i32, buf: &mut [F], offset: Option<&F>) -> Result<(), Mi355Error> {
    let n = buf.len();
    ntt(curve_id, buf, n, true, offset)
}

fn ntt<F: PrimeField>(curve_id: i32, buf: &mut [F], in_len: usize, inverse: bool, offset: Option<&F>) -> Result<(), Mi355Error> {
    assert!(buf.len().is_power_of_two() && in_len <= buf.len());
    let off = offset.filter(|o| !o.is_one()).map(|o| o as *const F as *const u64).unwrap_or(core::ptr::null());
    check(unsafe { mzk_ntt(curve_id, buf.as_mut_ptr() as *mut u64, in_len as u64, buf.len().trailing_zeros(), inverse as i32, off) })
}

/// The 18 + 7 forward coset FFTs of round 3 (prover.rs:552-567) in one pipelined call: upload k+1 | transform k | download k-1.
pub fn fft_batch_in_place<F: PrimeField>(curve_id: i32, bufs: &mut [&mut [F]], in_lens: &[usize], offset: Option<&F>) -> Result<(), Mi355Error> {
    let n = bufs[0].len();
    assert!(bufs.iter().all(|b| b.len() == n) && n.is_power_of_two());
    let ptrs: Vec<*mut u64> = bufs.iter_mut().map(|b| b.as_mut_ptr() as *mut u64).collect();
    let lens: Vec<u64> = in_lens.iter().map(|&l| l as u64).collect();
    let off = offset.filter(|o| !o.is_one()).map(|o| o as *const F as *const u64).unwrap_or(core::ptr::null());
    check(unsafe { mzk_ntt_batch(curve_id, bufs.len() as u32, ptrs.as_ptr(), lens.as_ptr(), n.trailing_zeros(), 0, off) })
}

/// A `Vec`-like buffer of field elements in page-locked host memory (mzk_host_alloc): transfers from it are DMA at link
/// rate and asynchronous, which is what lets the batch calls overlap them with the transforms.
pub struct PinnedBuf<F: PrimeField> {
    ptr: *mut F,
    len: usize,
}

impl<F: PrimeField> PinnedBuf<F> {
    pub fn zeroed(len: usize) -> Result<Self, Mi355Error> {
        let mut p: *mut c_void = core::ptr::null_mut();
        check(unsafe { mzk_host_alloc((len * core::mem::size_of::<F>()) as u64, &mut p) })?;
        unsafe { core::ptr::write_bytes(p as *mut u8, 0, len * core::mem::size_of::<F>()) };     // the all-zero image is F::zero()
        Ok(Self { ptr: p as *mut F, len })
    }
    pub fn as_mut_slice(&mut self) -> &mut [F] {
        unsafe { core::slice::from_raw_parts_mut(self.ptr, self.len) }
    }
}

impl<F: PrimeField> Drop for PinnedBuf<F> {
    fn drop(&mut self) {
        unsafe { mzk_host_free(self.ptr as *mut c_void) };
    }
}

/// Which `curve_id` a pairing engine maps to (the library supports the two curves of BASELINE.json's configs).
pub fn curve_id_of<E: Pairing>() -> Option<i32> {
    match <E::ScalarField as PrimeField>::MODULUS_BIT_SIZE {
        255 => Some(CURVE_BLS12_381),
        254 => Some(CURVE_BN254),
        _ => None,
    }
}
