//! gen_fixtures -- reference-side fixtures for the parity tests of mi355-zk.
//!
//! Reads the INPUTS of the committed vectors `tests/golden/{ntt,msm,kzg,proof}_vectors.json` (made by
//! `tests/golden/make_golden.py` / `make_proof_golden.py` from the big-int restatements under `oracle/`), runs the
//! REFERENCE's own code on them -- ark-poly `Radix2EvaluationDomain::{fft, ifft}`, ark-ec `VariableBaseMSM::msm_bigint`,
//! `UnivariateKzgPCS::commit`, `PlonkKzgSnark::{preprocess, prove}` with `jf_utils::test_rng` and `StandardTranscript` --
//! and writes `tests/golden/ref_{ntt,msm,kzg,proof}_vectors.json` in the same schema.  When those files exist,
//! `tests/test_oracle.py::test_reference_fixtures_*` and `tests/test_golden_proofs.py` compare the restatements' vectors with
//! them field by field; that comparison is what pins the oracle to the reference's outputs.
//!
//!     cd integration/rust && cargo run --release --features fixtures --bin gen_fixtures -- ../../tests/golden
//!
//! SRS of the proof cases, two families.  `proof_vectors.json`: `powers_of_g[i] = beta^i * G` with G the curve's standard generator
//! and beta = the FIRST draw of `test_rng`; the SRS is rebuilt here from (beta, G) and the same rng continues into `prove` -- the
//! sequence of `make_proof_golden.py`.  `proof_vectors_refsetup.json` ("setup": "universal_setup_for_testing"): the reference's OWN
//! `universal_setup_for_testing` (plonk/src/proof_system/snark.rs:485-526), which draws beta the same way and then g = G1::rand(rng),
//! h = G2::rand(rng) before the blinders -- what plonk/benches/bench.rs and the reference's tests prove over; those vectors also pin
//! the restated `G1::rand` (oracle/pyref_rng.py) and the product's mirror of it (mpc-jellyfish_amd/rng.py).
//!
//! Round 5: three more families.  `batch_vectors.json` -> `ref_batch_vectors.json` (`PlonkKzgSnark::batch_prove`, snark.rs:64-78),
//! `link_vectors.json` -> `ref_link_vectors.json` (`prove_with_link_hint` twice on one rng + `link_proofs`, snark.rs:81-114,
//! proof_linking.rs:80-111), and `general_ref_cases.json` -> `ref_general_circuits.json`: GENERAL circuits built here through the
//! reference's own gadgets (public inputs, add / mul / pow5 / linear-combination gates, shared variables = copy constraints, range and
//! key lookups), exported in the circuit-file format of `mpc-jellyfish_amd/circuit_io.py` ("MZKCIRC1": the arrays `Arithmetization`
//! exposes, Montgomery limbs) and proved by the reference.  The GPU tests feed that very file to `mzk_prove <curve> file`, to the
//! round-level C ABI and to the Python mirror, the CPU tests to the schoolbook oracle: every host must emit the reference's bytes.
//!
//! NOT COMPILED in this repository's build image (no Rust toolchain); see README.md.
use ark_ec::{pairing::Pairing, AffineRepr, CurveGroup, VariableBaseMSM};
use ark_ff::{BigInt, BigInteger, PrimeField, UniformRand, Zero};
use ark_poly::{univariate::DensePolynomial, DenseUVPolynomial, EvaluationDomain, Radix2EvaluationDomain};
use ark_serialize::CanonicalSerialize;
use jf_primitives::pcs::{
    prelude::{UnivariateKzgPCS, UnivariateProverParam, UnivariateUniversalParams},
    PolynomialCommitmentScheme,
};
use mpc_plonk::{
    proof_system::{structs::ProvingKey, PlonkKzgSnark, UniversalSNARK},
    transcript::StandardTranscript,
};
use mpc_relation::{proof_linking::GroupLayout, traits::*, PlonkCircuit, Variable};
use num_bigint::BigUint;
use serde_json::{json, Value};
use std::{fs, path::Path};

fn big(hex: &str) -> BigUint {
    BigUint::parse_bytes(hex.as_bytes(), 16).expect("hex integer")
}
fn fe<F: PrimeField>(hex: &str) -> F {
    F::from(big(hex))
}
fn hx<F: PrimeField>(x: &F) -> String {
    let b: BigUint = x.into_bigint().into();
    format!("{:x}", b)
}
/// a plain 256-bit integer (msm_bigint takes integers, not field elements: a scalar may exceed r)
fn bigint4(hex: &str) -> BigInt<4> {
    let digits = big(hex).to_u64_digits();
    let mut l = [0u64; 4];
    l[..digits.len()].copy_from_slice(&digits);
    BigInt::new(l)
}
fn strs(v: &Value) -> Vec<&str> {
    v.as_array().unwrap().iter().map(|s| s.as_str().unwrap()).collect()
}

macro_rules! curve_fixtures {
    ($modname:ident, $cid:expr, $engine:ty, $fr:ty, $fq:ty, $g1a:ty, $g1:ty, $g2:ty) => {
        mod $modname {
            use super::*;
            type E = $engine;
            type Fr = $fr;
            type Fq = $fq;
            type G1Affine = $g1a;
            type G1 = $g1;
            type G2 = $g2;
            const C_ID: u32 = $cid;

            fn point(v: &Value) -> G1Affine {
                if v.is_null() {
                    return G1Affine::identity();
                }
                let xy = strs(v);
                G1Affine::new_unchecked(fe::<Fq>(xy[0]), fe::<Fq>(xy[1]))
            }
            fn point_json(p: &G1Affine) -> Value {
                match p.xy() {
                    Some((x, y)) => json!([hx(x), hx(y)]),
                    None => Value::Null,
                }
            }
            fn g1_hex(p: &G1Affine) -> String {
                let mut b = Vec::new();
                p.serialize_compressed(&mut b).unwrap();
                hex::encode(b)
            }

            /// `Radix2EvaluationDomain::{fft, ifft}` on the generator coset or on H itself
            pub fn ntt(case: &Value) -> Value {
                let log_n = case["log_n"].as_u64().unwrap() as usize;
                let offset: Fr = fe(case["offset"].as_str().unwrap());
                let input: Vec<Fr> = strs(&case["input"]).into_iter().map(fe::<Fr>).collect();
                let h = Radix2EvaluationDomain::<Fr>::new(1 << log_n).unwrap();
                let d = if offset == Fr::from(1u64) { h } else { h.get_coset(offset).unwrap() };
                let fwd = d.fft(&input);
                let inv = d.ifft(&input);
                let mut out = case.clone();
                out["forward"] = json!(fwd.iter().map(hx).collect::<Vec<_>>());
                out["inverse"] = json!(inv.iter().map(hx).collect::<Vec<_>>());
                out
            }

            /// `VariableBaseMSM::msm_bigint`, then `into_affine` as the call sites do (univariate_kzg/mod.rs:109-111)
            pub fn msm(case: &Value) -> Value {
                let bases: Vec<G1Affine> = case["bases"].as_array().unwrap().iter().map(point).collect();
                let scalars: Vec<BigInt<4>> = strs(&case["scalars"]).into_iter().map(bigint4).collect();
                let res = <G1 as VariableBaseMSM>::msm_bigint(&bases, &scalars).into_affine();
                let mut out = case.clone();
                out["result"] = point_json(&res);
                out
            }

            /// `UnivariateKzgPCS::commit` over the vector's SRS
            pub fn kzg(case: &Value) -> Value {
                let srs: Vec<G1Affine> = case["srs"].as_array().unwrap().iter().map(point).collect();
                let coeffs: Vec<Fr> = strs(&case["coeffs"]).into_iter().map(fe::<Fr>).collect();
                let pp = UnivariateProverParam::<E> { powers_of_g: srs };
                let poly = DensePolynomial::from_coefficients_vec(coeffs);
                let com = UnivariateKzgPCS::<E>::commit(&pp, &poly).unwrap();
                let mut out = case.clone();
                out["commitment"] = point_json(&com.0);
                out
            }

            /// bench.rs:29-46 with the range_bit_len of the vector; `test_rng`: SRS trapdoor first, then `prove`'s own draws
            pub fn proof(case: &Value) -> Value {
                let num_gates = case["num_gates"].as_u64().unwrap() as usize;
                let ultra = case["plonk_type"].as_str().unwrap() == "UltraPlonk";
                let range_bits = case["range_bit_len"].as_u64().unwrap() as usize;
                let cs = bench_circuit(num_gates, ultra, range_bits);
                let n = cs.eval_domain_size().unwrap();
                assert_eq!(n as u64, case["domain_size"].as_u64().unwrap(), "domain size");

                let rng = &mut jf_utils::test_rng();
                let srs = if case["setup"].as_str() == Some("universal_setup_for_testing") {
                    // the reference's OWN testing setup (snark.rs:485-526): beta = Fr::rand, g = G1::rand, h = G2::rand from this rng,
                    // which then continues into `prove`; the vector records beta and g as oracle/pyref_rng.py restates them
                    let srs = PlonkKzgSnark::<E>::universal_setup_for_testing(n + 2, rng).unwrap();
                    assert_eq!(point_json(&srs.powers_of_g[0]), case["srs_g"], "g = G1::rand(rng): the restated sampler disagrees");
                    srs
                } else {
                    let beta = Fr::rand(rng);
                    assert_eq!(hx(&beta), case["srs_beta"].as_str().unwrap(), "first draw of test_rng");
                    let (g, h) = (G1::generator(), G2::generator());
                    let mut powers = Vec::with_capacity(n + 3);
                    let mut cur = g;
                    for _ in 0..n + 3 {
                        powers.push(cur);
                        cur *= beta;
                    }
                    UnivariateUniversalParams::<E> {
                        powers_of_g: G1::normalize_batch(&powers),
                        h: h.into_affine(),
                        beta_h: (h * beta).into_affine(),
                    }
                };
                let (pk, vk) = PlonkKzgSnark::<E>::preprocess(&srs, &cs).unwrap();
                let proof = PlonkKzgSnark::<E>::prove::<_, _, StandardTranscript>(rng, &cs, &pk, None).unwrap();
                let mut bytes = Vec::new();
                proof.serialize_compressed(&mut bytes).unwrap();
                let mut vk_bytes = Vec::new();
                vk.serialize_compressed(&mut vk_bytes).unwrap();
                let mut out = case.clone();
                out["k"] = json!(vk.k.iter().map(hx).collect::<Vec<_>>());
                out["selector_comms"] = json!(vk.selector_comms.iter().map(|c| g1_hex(&c.0)).collect::<Vec<_>>());
                out["sigma_comms"] = json!(vk.sigma_comms.iter().map(|c| g1_hex(&c.0)).collect::<Vec<_>>());
                out["proof"] = json!(hex::encode(bytes));
                out["vk_serialized"] = json!(hex::encode(vk_bytes));
                out.as_object_mut().unwrap().remove("challenges");       // not observable through the reference's public API
                out.as_object_mut().unwrap().remove("plookup_comms");    // PlookupVerifyingKey's fields are pub(crate): see vk_serialized
                out
            }

            /// plonk/benches/bench.rs:29-46, finalised
            fn bench_circuit(num_gates: usize, ultra: bool, range_bits: usize) -> PlonkCircuit<Fr> {
                let mut cs: PlonkCircuit<Fr> = if ultra { PlonkCircuit::new_ultra_plonk(range_bits) } else { PlonkCircuit::new_turbo_plonk() };
                let mut a = cs.zero();
                for _ in 0..num_gates - 10 {
                    a = cs.add(a, cs.one()).unwrap();
                }
                cs.finalize_for_arithmetization().unwrap();
                cs
            }

            /// powers_of_g[i] = beta^i * G with beta the FIRST draw of `rng` (the sequence of make_proof_golden.py)
            fn srs_from_first_draw<R: ark_std::rand::RngCore + ark_std::rand::CryptoRng>(rng: &mut R, n: usize, case: &Value) -> UnivariateUniversalParams<E> {
                let beta = Fr::rand(rng);
                assert_eq!(hx(&beta), case["srs_beta"].as_str().unwrap(), "first draw of test_rng");
                let (g, h) = (G1::generator(), G2::generator());
                let mut powers = Vec::with_capacity(n + 3);
                let mut cur = g;
                for _ in 0..n + 3 {
                    powers.push(cur);
                    cur *= beta;
                }
                UnivariateUniversalParams::<E> { powers_of_g: G1::normalize_batch(&powers), h: h.into_affine(), beta_h: (h * beta).into_affine() }
            }

            /// `PlonkKzgSnark::batch_prove` (snark.rs:64-78) over bench circuits of one domain size, one `test_rng`
            pub fn batch(case: &Value) -> Value {
                let ultra = case["plonk_type"].as_str().unwrap() == "UltraPlonk";
                let range_bits = case["range_bit_len"].as_u64().unwrap() as usize;
                let circuits: Vec<PlonkCircuit<Fr>> =
                    case["gates"].as_array().unwrap().iter().map(|g| bench_circuit(g.as_u64().unwrap() as usize, ultra, range_bits)).collect();
                let n = circuits[0].eval_domain_size().unwrap();
                assert_eq!(n as u64, case["domain_size"].as_u64().unwrap(), "domain size");
                let rng = &mut jf_utils::test_rng();
                let srs = srs_from_first_draw(rng, n, case);
                let keys: Vec<_> = circuits.iter().map(|cs| PlonkKzgSnark::<E>::preprocess(&srs, cs).unwrap()).collect();
                let cs_refs: Vec<&PlonkCircuit<Fr>> = circuits.iter().collect();
                let pk_refs: Vec<&ProvingKey<E>> = keys.iter().map(|(pk, _)| pk).collect();
                let proof = PlonkKzgSnark::<E>::batch_prove::<_, _, StandardTranscript>(rng, &cs_refs, &pk_refs).unwrap();
                let mut bytes = Vec::new();
                proof.serialize_compressed(&mut bytes).unwrap();
                let mut out = case.clone();
                out["batch_proof"] = json!(hex::encode(bytes));
                out.as_object_mut().unwrap().remove("challenges");
                out
            }

            /// `prove_with_link_hint` on two bench circuits (consecutive draws of one `test_rng`), then `link_proofs` (proof_linking.rs:80-111)
            pub fn link(case: &Value) -> Value {
                let gates: Vec<usize> = case["gates"].as_array().unwrap().iter().map(|g| g.as_u64().unwrap() as usize).collect();
                let lay: Vec<usize> = case["layout"].as_array().unwrap().iter().map(|g| g.as_u64().unwrap() as usize).collect();
                let (cs1, cs2) = (bench_circuit(gates[0], false, 8), bench_circuit(gates[1], false, 8));
                let n = cs1.eval_domain_size().unwrap();
                assert_eq!(n, cs2.eval_domain_size().unwrap(), "the two circuits share one domain");
                let rng = &mut jf_utils::test_rng();
                let srs = srs_from_first_draw(rng, n, case);
                let (pk1, _) = PlonkKzgSnark::<E>::preprocess(&srs, &cs1).unwrap();
                let (pk2, _) = PlonkKzgSnark::<E>::preprocess(&srs, &cs2).unwrap();
                let (proof1, hint1) = PlonkKzgSnark::<E>::prove_with_link_hint::<_, _, StandardTranscript>(rng, &cs1, &pk1).unwrap();
                let (proof2, hint2) = PlonkKzgSnark::<E>::prove_with_link_hint::<_, _, StandardTranscript>(rng, &cs2, &pk2).unwrap();
                let layout = GroupLayout::new(lay[0], lay[1], lay[2]);
                let lp = PlonkKzgSnark::<E>::link_proofs::<StandardTranscript>(&hint1, &hint2, &layout, &pk1.commit_key).unwrap();
                let ser = |p: &mpc_plonk::proof_system::structs::Proof<E>| {
                    let mut b = Vec::new();
                    p.serialize_compressed(&mut b).unwrap();
                    hex::encode(b)
                };
                let mut link_bytes = Vec::new();
                lp.serialize_compressed(&mut link_bytes).unwrap();                // quotient_commitment || opening_proof
                let mut out = case.clone();
                out["proofs"] = json!([ser(&proof1), ser(&proof2)]);
                out["link_proof"] = json!(hex::encode(link_bytes));
                out.as_object_mut().unwrap().remove("eta");                      // the challenge is not observable through the public API
                out
            }

            /// the in-memory image of an Fr element: 4 x u64 Montgomery limbs, little-endian (what the C ABI takes)
            fn mont_bytes(x: &Fr, out: &mut Vec<u8>) {
                for limb in (x.0).0.iter() {
                    out.extend_from_slice(&limb.to_le_bytes());
                }
            }

            /// A GENERAL circuit through the reference's own gadgets, exported ("MZKCIRC1", mpc-jellyfish_amd/circuit_io.py) and proved.
            /// case: {"curve", "plonk_type", "rounds", "seed", "range_bit_len"}
            pub fn general(case: &Value) -> Value {
                let ultra = case["plonk_type"].as_str().unwrap() == "UltraPlonk";
                let rounds = case["rounds"].as_u64().unwrap() as usize;
                let seed = case["seed"].as_u64().unwrap();
                let range_bits = case["range_bit_len"].as_u64().unwrap() as usize;
                let mut cs: PlonkCircuit<Fr> = if ultra { PlonkCircuit::new_ultra_plonk(range_bits) } else { PlonkCircuit::new_turbo_plonk() };
                // two public inputs, then `rounds` of:  s = x + y;  m = s * x;  p = m^5;  t = 2 s + 3 m + 5 p + 7 y;  (x, y) <- (t, s)
                // -- every variable is used by several gates (copy constraints over all five wires)
                let mut x: Variable = cs.create_public_variable(Fr::from(seed)).unwrap();
                let mut y: Variable = cs.create_public_variable(Fr::from(seed + 1)).unwrap();
                let coeffs = [Fr::from(2u64), Fr::from(3u64), Fr::from(5u64), Fr::from(7u64)];
                let mut small: Vec<Variable> = Vec::new();
                for i in 0..rounds {
                    let s = cs.add(x, y).unwrap();
                    let m = cs.mul(s, x).unwrap();
                    let p = cs.pow5(m).unwrap();
                    let t = cs.lc(&[s, m, p, y], &coeffs).unwrap();
                    let c = cs.add_constant(t, &Fr::from(11u64 + i as u64)).unwrap();
                    small.push(cs.create_variable(Fr::from((seed + 3 * i as u64) % (1u64 << range_bits))).unwrap());
                    x = c;
                    y = s;
                }
                if ultra {
                    for v in small.iter() {
                        cs.enforce_in_range(*v, range_bits).unwrap();                       // range lookups
                    }
                    // a key table of `rounds` entries (values = pairs of circuit variables) and as many lookups into it
                    let table: Vec<(Variable, Variable)> = (0..rounds).map(|i| (small[i], small[(i + 1) % rounds])).collect();
                    let lookups: Vec<(Variable, Variable, Variable)> = (0..rounds)
                        .map(|i| {
                            let j = (i * 5 + 1) % rounds;
                            let key = cs.create_variable(Fr::from(j as u64)).unwrap();
                            (key, small[j], small[(j + 1) % rounds])
                        })
                        .collect();
                    cs.create_table_and_lookup_variables(&lookups, &table).unwrap();
                }
                cs.finalize_for_arithmetization().unwrap();
                let n = cs.eval_domain_size().unwrap();
                let pub_input = cs.public_input().unwrap();
                cs.check_circuit_satisfiability(&pub_input).unwrap();

                let rng = &mut jf_utils::test_rng();
                let beta = Fr::rand(rng);
                let mut with_beta = case.clone();
                with_beta["srs_beta"] = json!(hx(&beta));
                let srs = srs_from_first_draw(&mut jf_utils::test_rng(), n, &with_beta);
                let (pk, vk) = PlonkKzgSnark::<E>::preprocess(&srs, &cs).unwrap();
                let proof = PlonkKzgSnark::<E>::prove::<_, _, StandardTranscript>(rng, &cs, &pk, None).unwrap();
                let mut bytes = Vec::new();
                proof.serialize_compressed(&mut bytes).unwrap();
                let mut vk_bytes = Vec::new();
                vk.serialize_compressed(&mut vk_bytes).unwrap();

                // the circuit file: values on H of everything `Arithmetization` exposes as polynomials
                let dom = Radix2EvaluationDomain::<Fr>::new(n).unwrap();
                let w = cs.num_wire_types();
                let mut file: Vec<u8> = b"MZKCIRC1".to_vec();
                for v in [C_ID, w as u32, n.trailing_zeros(), pub_input.len() as u32] {
                    file.extend_from_slice(&v.to_le_bytes());
                }
                for k in vk.k.iter() {
                    mont_bytes(k, &mut file);
                }
                let mut push_polys = |polys: Vec<DensePolynomial<Fr>>, file: &mut Vec<u8>| {
                    for p in polys.iter() {
                        for v in dom.fft(&p.coeffs).iter() {
                            mont_bytes(v, file);
                        }
                    }
                };
                push_polys(cs.compute_selector_polynomials().unwrap(), &mut file);
                push_polys(cs.compute_extended_permutation_polynomials().unwrap(), &mut file);
                if ultra {
                    push_polys(
                        vec![cs.compute_range_table_polynomial().unwrap(), cs.compute_key_table_polynomial().unwrap(),
                             cs.compute_table_dom_sep_polynomial().unwrap(), cs.compute_q_dom_sep_polynomial().unwrap()],
                        &mut file,
                    );
                }
                push_polys(cs.compute_wire_polynomials().unwrap(), &mut file);
                for row in 0..pub_input.len() as u64 {
                    file.extend_from_slice(&row.to_le_bytes());                            // the IO gates sit on rows 0 .. num_inputs - 1
                }
                for v in pub_input.iter() {
                    mont_bytes(v, &mut file);
                }

                let mut out = with_beta;
                out["domain_size"] = json!(n);
                out["num_gates"] = json!(cs.num_gates());
                out["public_input"] = json!(pub_input.iter().map(hx).collect::<Vec<_>>());
                out["k"] = json!(vk.k.iter().map(hx).collect::<Vec<_>>());
                out["selector_comms"] = json!(vk.selector_comms.iter().map(|c| g1_hex(&c.0)).collect::<Vec<_>>());
                out["sigma_comms"] = json!(vk.sigma_comms.iter().map(|c| g1_hex(&c.0)).collect::<Vec<_>>());
                out["vk_serialized"] = json!(hex::encode(vk_bytes));
                out["circuit_file"] = json!(hex::encode(file));
                out["proof"] = json!(hex::encode(bytes));
                out
            }
        }
    };
}

curve_fixtures!(bls, 0u32, ark_bls12_381::Bls12_381, ark_bls12_381::Fr, ark_bls12_381::Fq, ark_bls12_381::G1Affine, ark_bls12_381::G1Projective,
                ark_bls12_381::G2Projective);
curve_fixtures!(bn, 1u32, ark_bn254::Bn254, ark_bn254::Fr, ark_bn254::Fq, ark_bn254::G1Affine, ark_bn254::G1Projective, ark_bn254::G2Projective);

fn run_to(dir: &Path, name: &str, out_name: &str, f0: fn(&Value) -> Value, f1: fn(&Value) -> Value) {
    let text = fs::read_to_string(dir.join(format!("{name}.json"))).expect("golden vector file");
    let cases: Vec<Value> = serde_json::from_str(&text).unwrap();
    let out: Vec<Value> = cases.iter().map(|c| if c["curve"].as_u64().unwrap() == 0 { f0(c) } else { f1(c) }).collect();
    fs::write(dir.join(format!("{out_name}.json")), serde_json::to_string(&out).unwrap()).unwrap();
    println!("{out_name}.json: {} cases", out.len());
}
fn run(dir: &Path, name: &str, f0: fn(&Value) -> Value, f1: fn(&Value) -> Value) {
    run_to(dir, name, &format!("ref_{name}"), f0, f1)
}

fn main() {
    let dir = std::env::args().nth(1).unwrap_or_else(|| "../../tests/golden".to_string());
    let dir = Path::new(&dir);
    run(dir, "ntt_vectors", bls::ntt, bn::ntt);
    run(dir, "msm_vectors", bls::msm, bn::msm);
    run(dir, "kzg_vectors", bls::kzg, bn::kzg);
    run(dir, "proof_vectors", bls::proof, bn::proof);
    run(dir, "proof_vectors_refsetup", bls::proof, bn::proof);
    run(dir, "batch_vectors", bls::batch, bn::batch);
    run(dir, "link_vectors", bls::link, bn::link);
    run_to(dir, "general_ref_cases", "ref_general_circuits", bls::general, bn::general);
    let _ = BigInt::<4>::zero().is_zero();
}
