"""GPU: the prover's rounds behind the C ABI (mzk_prover_create / round1 .. round5, include/mzk.h; csrc/prover.hip) driven through
ctypes (mpc-jellyfish_amd/prover.py) the way a Rust caller would drive them -- transcript, rng and Proof assembly on the caller's
side -- on GENERAL circuits: non-zero public input, add / mul / x^5 gates, copy constraints, key and range lookups.  The bytes must
equal the Python mirror's (prover.TurboPlonkProver, which sequences the library's primitives itself) and the golden vectors the
oracle produced on the CPU, and the restated reference verifier (oracle/pyref_verifier.py) must accept them."""
import ctypes as C
import random
from importlib import import_module

import numpy as np
import pytest

import mirror_prover as MP          # the primitive-level sequencing of the rounds: test code since round 5

from conftest import build_circuit, build_ultra_circuit, fr_mont_limbs, load_golden, verifying_key
import pyref_fs as FS

pytestmark = pytest.mark.gpu
TABLES = ("range", "key", "table_dom_sep", "q_dom_sep")


def _native(mj):
    """the product's client of the round-level C ABI under the names these tests have used since round 4"""
    import types
    return types.SimpleNamespace(NativeProver=mj.prover.TurboPlonkProver, preprocess=mj.snark.preprocess, prove=mj.snark.prove,
                                 batch_prove=mj.batch.batch_prove, round3=mj.prover.round3, round5=mj.prover.round5)


def _general_instance(mj, pc, c, log_n, ultra, rng):
    dom = mj.Radix2EvaluationDomain(c, log_n)
    tabs = None
    if ultra:
        sel, sig, k, w, pi, tabs = build_ultra_circuit(pc, log_n, rng)
    else:
        sel, sig, k, w, pi = build_circuit(pc, log_n, rng)
    kw = {"plookup": {name: dom.ifft(fr_mont_limbs(c, tabs[key])) for name, key in zip(mj.plonk.PLOOKUP_TABLE_POLYS, TABLES)}} if ultra else {}
    sel_p, sig_p = [dom.ifft(fr_mont_limbs(c, s)) for s in sel], [dom.ifft(fr_mont_limbs(c, s)) for s in sig]
    return sel_p, sig_p, k, np.stack([fr_mont_limbs(c, col) for col in w]), pi, kw


@pytest.mark.parametrize("curve_id,ultra,log_n", [(0, False, 6), (1, False, 9), (1, True, 6), (0, True, 8), (0, False, 3), (1, True, 4), (0, False, 12)])
def test_round_level_abi_on_general_circuits(gpu, mj, pyref, curve_id, ultra, log_n):
    import pyref_verifier as V
    N = _native(mj)
    c, pc = mj.params.CURVES[curve_id], pyref.CURVES[curve_id]
    n, r, W = 1 << log_n, c.r, 6 if ultra else 5
    rng = random.Random(9100 + curve_id + 2 * ultra + log_n)
    sel_p, sig_p, k, wires, pi, kw = _general_instance(mj, pc, c, log_n, ultra, rng)
    srs_beta = rng.randrange(1, r)
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, srs_beta, n + 2)
    mirror = MP.TurboPlonkProver(c, n, sel_p, sig_p, k, ck, **kw)
    native = N.NativeProver(c, n, sel_p, sig_p, k, ck, **kw)
    pub = pi[:4]
    assert pub[3] != 0 and not any(pi[4:])
    # the verifying key the library derives from its resident coefficient forms is the mirror's
    for a, b in zip(native.vk_commitments(), mirror.vk_commitments()):
        assert a == b
    if ultra:
        assert native.plookup_vk_commitments() == mirror.plookup_vk_commitments()
    blind = mj.snark.draw_blinders(c, mj.rng.test_rng(), W, ultra)
    want = mj.snark.serialize_proof(c, mirror.prove(wires, fr_mont_limbs(c, pi), mj.prover.TranscriptChallenges(mirror, pub), blind))
    # public input as the list of the first rows (the reference's layout), as (rows, values), and as the mirror's n-vector
    for pub_arg in (pub, ([3], [pub[3]]), fr_mont_limbs(c, pi)):
        src = mj.prover.TranscriptChallenges(native, pub)
        core = native.prove(wires, pub_arg, src, blind)
        got = mj.snark.serialize_proof(c, core)
        assert got == want
    vk = verifying_key(mj, pc, native, len(pub))
    assert V.verify(pc, FS.StandardTranscript(pc, b"PlonkProof"), vk, pub, got, pyref.g1_gen(pc), srs_beta)
    # device-resident wire values, and a profiled proof: same bytes, per-stage timings on the handle
    import torch
    dev_w = torch.from_numpy(wires.view(np.int64)).cuda()
    core = native.prove(dev_w, pub, mj.prover.TranscriptChallenges(native, pub), blind, profile=True)
    assert mj.snark.serialize_proof(c, core) == want
    assert {"r1_ntt_mask", "r1_commit", "r2_product", "r3_quotient", "r3_commit", "r4_evals", "r5_polys", "r5_commit"} <= set(core.timings_ms)
    hb = native.hbm_bytes()
    assert hb["fixed_coefficient_forms"] == (len(sel_p) + W + (4 if ultra else 0)) * n * 32 and hb["proving_key_evaluations"] > 0 and hb["prover_workspace"] > 0
    mirror.release()
    native.release()
    ck.release()


@pytest.mark.parametrize("curve_id,ultra,log_n", [(0, False, 6), (1, True, 6), (1, False, 3)])
def test_unsatisfied_witness_is_rejected_under_the_reference_error_name(gpu, mj, pyref, curve_id, ultra, log_n):
    """prover.rs:915-918: WrongQuotientPolyDegree.  With the quotient's top coefficients taken from its numerator the guard is the
    identity at zeta, at the end of round 5 (the tiny-domain path that keeps the reference's own degree guard in round 3: next test)."""
    N = _native(mj)
    c, pc = mj.params.CURVES[curve_id], pyref.CURVES[curve_id]
    n, r, W = 1 << log_n, c.r, 6 if ultra else 5
    rng = random.Random(77 + curve_id)
    sel_p, sig_p, k, wires, pi, kw = _general_instance(mj, pc, c, log_n, ultra, rng)
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, rng.randrange(1, r), n + 2)
    native = N.NativeProver(c, n, sel_p, sig_p, k, ck, **kw)
    blind = mj.snark.draw_blinders(c, mj.rng.test_rng(), W, ultra)
    pub = pi[:4]
    ok = native.prove(wires, pub, mj.prover.TranscriptChallenges(native, pub), blind)
    bad = wires.copy()
    bad[4, 0] = fr_mont_limbs(c, [rng.randrange(r)])[0]                  # row 0 is an addition gate: its output no longer matches
    with pytest.raises(mj.plonk.PlonkError) as e:
        native.prove(bad, pub, mj.prover.TranscriptChallenges(native, pub), blind)
    assert e.value.kind == "WrongQuotientPolyDegree"
    # a public input the circuit does not hold (asserted wrongly as zero included) trips it too
    for wrong in ([], pub[:3] + [(pub[3] + 1) % r]):
        with pytest.raises(mj.plonk.PlonkError):
            native.prove(wires, wrong, mj.prover.TranscriptChallenges(native, pub), blind)
    # the handle is usable afterwards
    again = native.prove(wires, pub, mj.prover.TranscriptChallenges(native, pub), blind)
    assert mj.snark.serialize_proof(c, again) == mj.snark.serialize_proof(c, ok)
    native.release()
    ck.release()


def test_tiny_domain_keeps_the_degree_guard_of_round_3(gpu, mj):
    """n = 8 with six wire types: n <= W + 2, the quotient comes from all classes and `WrongQuotientPolyDegree` fires where the
    reference raises it (prover.rs:915-918), in round 3."""
    import ctypes as C
    N = _native(mj)
    c = mj.params.CURVES[1]
    cs = mj.snark.gen_circuit_for_bench(c, 16, "UltraPlonk", range_bit_len=2)
    assert cs.n == 8
    rng = mj.rng.test_rng()
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, mj.rng.fr_rand(c, rng), cs.n + 2)
    pk, npk = MP.preprocess(ck, cs), N.preprocess(ck, cs)
    g1, g2 = mj.rng.test_rng(), mj.rng.test_rng()
    assert N.prove(g1, cs, npk)[1] == MP.prove(g2, cs, pk)[1]
    bad = cs.wire_values.clone()
    bad[4, 3] = bad[4, 4]                                                # the output of the addition gate on row 3 takes row 4's value
    blind = mj.snark.draw_blinders(c, mj.rng.test_rng(), 6, True)
    src = mj.prover.TranscriptChallenges(npk, [])
    wires_comms = npk.round1(bad, [], blind.wires)
    tau = src.after_round1(wires_comms)
    beta, gamma = src.after_round1_5(npk.round1_5(tau, blind.h))
    alpha = src.after_round2(npk.round2(beta, gamma, blind.z), npk.round2_5(blind.prod_lookup))
    with pytest.raises(mj.plonk.PlonkError) as e:
        N.round3([npk], alpha, blind.quot)
    assert e.value.kind == "WrongQuotientPolyDegree" and "degree" in str(e.value)
    pk.release()
    npk.release()
    ck.release()


def test_rounds_out_of_order_and_bad_arguments(gpu, mj, pyref):
    N = _native(mj)
    L = gpu.load()
    c, pc = mj.params.CURVES[0], pyref.CURVES[0]
    log_n, n = 5, 32
    rng = random.Random(5)
    sel_p, sig_p, k, wires, pi, kw = _general_instance(mj, pc, c, log_n, False, rng)
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, 12345, n + 2)
    native = N.NativeProver(c, n, sel_p, sig_p, k, ck)
    one = mj.params.fr_to_mont(c, [1, 2, 3])
    out = np.zeros((8, 2, 6), dtype=np.uint64)
    p = lambda a: C.c_void_p(a.ctypes.data)
    STATE, UNSUPPORTED, INVALID, BAD_HANDLE = -10, -5, -1, -4
    assert L.mzk_prover_round2(native.handle, p(one), p(one), p(one), p(out)) == STATE           # before round 1
    assert L.mzk_prover_round4(native.handle, p(one), p(out)) == STATE
    hs = (C.c_uint64 * 1)(native.handle)
    assert L.mzk_prover_round3(hs, 1, p(one), p(one), p(out)) == STATE
    assert b"out of order" in L.mzk_last_error()
    assert L.mzk_prover_round1_5(native.handle, p(one), p(one), p(out)) == UNSUPPORTED           # TurboPlonk has no round 1.5
    bl = mj.params.fr_to_mont(c, list(range(1, 11)))
    assert L.mzk_prover_round1(native.handle, 7, p(wires), 5 * n, None, None, 0, p(bl), p(out)) == INVALID      # unknown witness kind
    assert L.mzk_prover_round1(native.handle, 1, p(wires), 5 * n - 1, None, None, 0, p(bl), p(out)) == INVALID  # wrong length
    assert L.mzk_prover_round1(native.handle, 2, p(wires), 5, None, None, 0, p(bl), p(out)) == STATE            # no wire variables yet
    assert L.mzk_prover_round1(native.handle + 99, 1, p(wires), 5 * n, None, None, 0, p(bl), p(out)) == BAD_HANDLE
    with pytest.raises(mj.MzkError):
        native.set_wire_variables(np.full((5, n), 9, dtype=np.uint32), 9)                        # index 9 >= 9 variables: the reference panics
    hs2 = (C.c_uint64 * 2)(native.handle, native.handle)
    assert L.mzk_prover_round3(hs2, 2, p(one), p(one), p(out)) == INVALID                         # one handle per instance
    # a commit key that is too small
    small = mj.UnivariateProverParam.gen_srs_for_testing(c, 12345, n)
    with pytest.raises(mj.MzkError):
        N.NativeProver(c, n, sel_p, sig_p, k, small)
    small.release()
    # domains whose quotient does not fit the 8n-point domain are refused with a message (the reference would take a 16n domain there):
    # n = 2 with five wire types, n = 4 with six
    z = np.zeros((14 * 4, 4), dtype=np.uint64)
    kk = mj.params.fr_to_mont(c, [1, 2, 3, 4, 5, 6])
    hh = C.c_uint64()
    for lg, W in ((1, 5), (1, 6), (2, 6)):
        assert L.mzk_prover_create(0, lg, W, p(z), p(z), p(z) if W == 6 else None, 1 << lg, p(kk), ck.handle, 0, None, C.byref(hh)) == UNSUPPORTED
        assert b"domain too small" in L.mzk_last_error()
    native.release()
    assert L.mzk_prover_destroy(native.handle or 12345) == BAD_HANDLE
    ck.release()


@pytest.mark.parametrize("curve_id,plonk_type,num_gates", [(0, "TurboPlonk", 1 << 13), (1, "UltraPlonk", 1 << 13), (0, "TurboPlonk", 100)])
def test_bench_circuit_every_witness_kind_and_the_lagrange_key(gpu, mj, pyref, curve_id, plonk_type, num_gates):
    """The reference's bench circuit (plonk/benches/bench.rs:29-46) through snark.prove (Python mirror) and native.prove: device wires,
    host wires, the witness VECTOR from host and from device memory gathered through the resident wire_variables; round 1 (and 1.5)
    over the Lagrange-basis key when asked for (the default asks from 2^18 gates on: snark.LAGRANGE_MIN_DOMAIN)."""
    import torch
    N = _native(mj)
    c = mj.params.CURVES[curve_id]
    cs = mj.snark.gen_circuit_for_bench(c, num_gates, plonk_type)
    rng0 = mj.rng.test_rng()
    srs_beta = mj.rng.fr_rand(c, rng0)
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, srs_beta, cs.n + 2)
    pk = MP.preprocess(ck, cs)
    npk = N.preprocess(ck, cs, lagrange=True if cs.n >= 1 << 13 else None)
    assert (npk.lagrange_ck is not None) == (cs.n >= 1 << 13) and mj.snark.LAGRANGE_MIN_DOMAIN == 1 << 18

    def fresh_rng():
        g = mj.rng.test_rng()
        mj.rng.fr_rand(c, g)
        return g

    _, want = MP.prove(fresh_rng(), cs, pk)
    npk.set_wire_variables(cs.wire_variables.cpu().numpy().astype(np.uint32), int(cs.witness.shape[0]))
    host_vec = cs.witness.cpu().pin_memory()
    kinds = {"device wires": None, "host wires": cs.wire_values.cpu(), "host vector": mj.snark.HostWitness(host_vec, cs.wire_variables),
             "device vector": mj.snark.HostWitness(cs.witness, cs.wire_variables)}
    for name, wit in kinds.items():
        _, got = N.prove(fresh_rng(), cs, npk, witness=wit)
        assert got == want, name
    # consecutive proofs from one rng stream agree too (the bench's usage: bench.rs:56-60)
    ga, gb = fresh_rng(), fresh_rng()
    for _ in range(3):
        assert N.prove(ga, cs, npk)[1] == MP.prove(gb, cs, pk)[1]
    torch.cuda.synchronize()
    pk.release()
    if npk.lagrange_ck is not None:
        npk.lagrange_ck.release()
    npk.release()
    ck.release()


@pytest.mark.parametrize("index,lagrange", [(0, False), (1, False), (2, True), (3, True)])
def test_golden_proofs_through_the_round_level_abi(gpu, mj, index, lagrange):
    """tests/golden/proof_vectors.json: whole proofs produced on the CPU by the oracle alone (oracle/pyref_snark.py) -- the library's
    own rounds must emit exactly these bytes and verifying-key commitments (round 1 from coefficient forms / over the Lagrange key)."""
    N = _native(mj)
    vec = load_golden("proof_vectors")[index]
    c = mj.params.CURVES[vec["curve"]]
    cs = mj.snark.gen_circuit_for_bench(c, vec["num_gates"], vec["plonk_type"], range_bit_len=vec["range_bit_len"])
    rng = mj.rng.test_rng()
    srs_beta = mj.rng.fr_rand(c, rng)
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, srs_beta, cs.n + 2)
    npk = N.preprocess(ck, cs, lagrange=lagrange)
    sel, sig = npk.vk_commitments()
    assert [mj.snark._g1(c, x).hex() for x in sel] == vec["selector_comms"] and [mj.snark._g1(c, x).hex() for x in sig] == vec["sigma_comms"]
    _, got = N.prove(rng, cs, npk)
    assert got.hex() == vec["proof"]
    assert {name: "%x" % v for name, v in npk.last_challenges.items()} == vec["challenges"]
    if npk.lagrange_ck is not None:
        npk.lagrange_ck.release()
    npk.release()
    ck.release()


@pytest.mark.parametrize("curve_id,ultra,log_n", [(0, False, 5), (1, True, 5)])
def test_batch_prove_over_native_handles_matches_the_mirror(gpu, mj, pyref, curve_id, ultra, log_n):
    """PlonkKzgSnark::batch_prove (snark.rs:64-78): K handles, rounds 3 and 5 once over all of them (alpha_base_k = alpha^(3k) / alpha^(7k))."""
    N = _native(mj)
    c, pc = mj.params.CURVES[curve_id], pyref.CURVES[curve_id]
    n, r, W = 1 << log_n, c.r, 6 if ultra else 5
    rng = random.Random(600 + curve_id)
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, rng.randrange(1, r), n + 2)
    mirrors, natives, wires_l, pis, pubs, blinds = [], [], [], [], [], []
    for _ in range(3):
        sel_p, sig_p, k, wires, pi, kw = _general_instance(mj, pc, c, log_n, ultra, rng)
        mirrors.append(MP.TurboPlonkProver(c, n, sel_p, sig_p, k, ck, **kw))
        natives.append(N.NativeProver(c, n, sel_p, sig_p, k, ck, **kw))
        wires_l.append(wires); pis.append(fr_mont_limbs(c, pi)); pubs.append(pi[:4])
        rnd = lambda cnt: [rng.randrange(r) for _ in range(cnt)]
        blinds.append(mj.prover.Blinders([rnd(2) for _ in range(W)], rnd(3), [], [rnd(3), rnd(3)] if ultra else None, rnd(3) if ultra else None))
    quot_blind = [rng.randrange(r) for _ in range(W - 1)]
    want = MP.batch_prove(mirrors, wires_l, pis, pubs, blinds, quot_blind, extra_transcript_init_msg=b"batch")
    got = N.batch_prove(natives, wires_l, pubs, blinds, quot_blind, extra_transcript_init_msg=b"batch")
    assert got.challenges == want.challenges
    assert mj.batch.serialize_batch_proof(c, got) == mj.batch.serialize_batch_proof(c, want)
    # an aggregate of one instance is that instance's plain proof
    one = N.batch_prove(natives[:1], wires_l[:1], pubs[:1], blinds[:1], quot_blind)
    b0 = mj.prover.Blinders(blinds[0].wires, blinds[0].z, quot_blind, blinds[0].h, blinds[0].prod_lookup)
    single = natives[0].prove(wires_l[0], pubs[0], mj.prover.TranscriptChallenges(natives[0], pubs[0]), b0)
    assert one.split_quot_poly_comms == single.split_quot_poly_comms and one.opening_proof == single.opening_proof
    for p in mirrors + natives:
        p.release()
    ck.release()


def test_a_wrongly_asserted_pi_zero_trips_the_identity_at_zeta(gpu, mj, pyref):
    """MZK_QUOTIENT_PI_ZERO / `pi_zero=True` lets round 3 skip the public-input polynomial; asserted for a circuit that HAS a non-zero
    public input, the quotient no longer matches the numerator and the check at zeta (the only guard on the W-class path) must fire --
    in the Python mirror, where the caller asserts it, as in the library's rounds, which decide it from the data."""
    c, pc = mj.params.CURVES[0], pyref.CURVES[0]
    log_n, n, W = 6, 64, 5
    rng = random.Random(4242)
    sel_p, sig_p, k, wires, pi, kw = _general_instance(mj, pc, c, log_n, False, rng)
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, rng.randrange(1, c.r), n + 2)
    mirror = MP.TurboPlonkProver(c, n, sel_p, sig_p, k, ck)
    blind = mj.snark.draw_blinders(c, mj.rng.test_rng(), W, False)
    pub = pi[:4]
    mirror.prove(wires, fr_mont_limbs(c, pi), mj.prover.TranscriptChallenges(mirror, pub), blind)                     # fine
    with pytest.raises(mj.plonk.PlonkError) as e:
        mirror.prove(wires, fr_mont_limbs(c, pi), mj.prover.TranscriptChallenges(mirror, pub), blind, pi_zero=True)
    assert e.value.kind == "WrongQuotientPolyDegree"
    # switching the guard off on this path is refused: it would leave nothing
    mirror.identity_check = False
    with pytest.raises(mj.plonk.PlonkError):
        mirror.prove(wires, fr_mont_limbs(c, pi), mj.prover.TranscriptChallenges(mirror, pub), blind)
    mirror.release()
    ck.release()


@pytest.mark.parametrize("args", [["--log-n", "12"], ["--log-n", "13", "--ultra"]])
def test_proofs_in_flight_on_one_card_are_the_same_bytes(gpu, args):
    """Round 5: every handle runs on a stream of its context.  tools/prove_in_flight.py --check: three host threads, three device contexts
    on the ONE card (MZK_VIRTUAL_DEVICES), proving concurrently from identical rng streams -- every proof made while the others are in
    flight must be the same bytes, and the same as each context's proof made alone."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "prove_in_flight.py"), "--in-flight", "1,3", "--reps", "6", "--check"] + args,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["contexts_agree_on_proof"] is True and d["proofs_made_concurrently_identical"] is True
    assert d["in_flight"]["3"]["proofs_per_s"] > 0
