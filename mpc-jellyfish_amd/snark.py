"""`PlonkKzgSnark::{preprocess, prove}` around the device prover core, with the reference's own deterministic inputs
(SURVEY.md 8(d), 8(f) N3/N4):

    gen_circuit_for_bench            plonk/benches/bench.rs:29-46        a = 0; repeat (gates - 10) times a = a + 1
    PlonkCircuit::new / finalize     relation/src/constraint_system.rs:195-225, 966-999 (padding gates, wire permutation)
    PlonkKzgSnark::preprocess        plonk/src/proof_system/snark.rs:529-617 (selector / sigma polynomials and commitments)
    PlonkKzgSnark::prove             snark.rs:624-651 -> batch_prove_internal :201-469 (one instance)
    Proof / PlookupProof layout      plonk/src/proof_system/structs.rs:59-84, 208-222, 440-450, 496-541 (ark-serialize, compressed)

The circuit builder here is not the reference's constraint-system DSL (out of scope): it produces, vectorised, exactly
the finalised arithmetisation that DSL yields for the bench circuit -- gate order, padding rows, the variable -> wire
cycles of compute_wire_permutation (:743-778) and the ChaCha-derived coset representatives.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass

import numpy as np

from . import kzg, poly, prover as _prover, rng as _rng, transcript as _transcript
from .domain import Radix2EvaluationDomain
from .params import CurveParams, curve as _curve, fr_from_mont, fr_to_mont, fq_from_mont

TURBO, ULTRA = "TurboPlonk", "UltraPlonk"


@dataclass
class BenchCircuit:
    """What `Arithmetization` exposes of a finalised PlonkCircuit (relation/src/traits.rs), as device tensors."""
    curve: CurveParams
    plonk_type: str
    n: int
    k: list
    wire_values: object          # (W, n, 4) int64 CUDA, Montgomery: witness[wire_variable(i, j)]
    selector_values: object      # (13 | 14, n, 4)
    sigma_values: object         # (W, n, 4): extended permutation k[pi] * w^pj
    pub_input_values: object     # (n, 4) -- the bench circuit has no public input
    public_input: list
    table_values: object | None  # (4, n, 4): range, key, table_dom_sep, q_dom_sep (UltraPlonk)
    witness: object = None       # (n_vars, 4): the witness vector `wire_values` was gathered from ...
    wire_variables: object = None  # ... (W, n) int32: by these variable indices (constraint_system.rs:1225-1247)

    @property
    def num_wire_types(self):
        return 6 if self.plonk_type == ULTRA else 5


def _to_mont_dev(c: CurveParams, canon):
    """(len, 4) int64 CUDA tensor of canonical limbs -> Montgomery form: x * (R mod r) with the Montgomery product."""
    return poly.lincomb(c, [((1 << 256) % c.r, canon)])


def gen_circuit_for_bench(curve, num_gates: int, plonk_type: str = TURBO, range_bit_len: int = 8, dense_seed: int | None = None) -> BenchCircuit:
    """dense_seed: not the reference's circuit but one with the SAME gates (the same selector polynomials: two constant gates, then
    num_gates - 10 addition gates, padding) whose witness is dense -- the bench circuit's wires are 0, 1, 2, .. / all ones / all
    zero / all zero / 1, 2, 3, ..: two of its five wire polynomials are zero and two are sparse, which flatters round 1.  Here row
    t adds a + b = c with b, the unconstrained wires 2 and 3 and every second a drawn at random in [0, r) (a of an odd row is the
    previous row's c, so the copy constraints still chain rows together): every wire polynomial has n random coefficients."""
    import torch
    c = _curve(curve)
    assert num_gates >= 16 and plonk_type in (TURBO, ULTRA)
    ultra = plonk_type == ULTRA
    W = 6 if ultra else 5
    n_add = num_gates - 10                                   # bench.rs:38-41
    used = 2 + n_add                                         # two constant gates for the variables 0 and 1 (:221-223)
    need = max(used, (1 << range_bit_len) + 1) if ultra else used          # constraint_system.rs:982-987 / proof_linking/mod.rs:78-88
    n = 1 << (need - 1).bit_length()
    log_n = n.bit_length() - 1
    dev = torch.device("cuda")
    var = torch.zeros((W, n), dtype=torch.int64, device=dev)
    rows = torch.arange(2, 2 + n_add, device=dev)
    var[4, 1] = 1                                                                          # constant gate of `one`
    if dense_seed is None:
        # variable index on every (wire, row): 0 = zero, 1 = one, 2 + t = output of the t-th addition
        var[0, 2:2 + n_add] = torch.where(rows == 2, torch.zeros_like(rows), rows - 1)      # a: previous sum (the first is `zero`)
        var[1, 2:2 + n_add] = 1                                                            # b = one
        var[4, 2:2 + n_add] = rows                                                         # c = 2 + t  (row 2 + t)
        # witness: zero, one, then 1, 2, 3, ...
        wit = torch.zeros((2 + n_add, 4), dtype=torch.int64, device=dev)
        wit[1, 0] = 1
        wit[2:, 0] = torch.arange(1, n_add + 1, device=dev)
        wit = _to_mont_dev(c, wit)
    else:
        assert not ultra, "dense witness: TurboPlonk only"
        from .params import random_fr_mont
        t = torch.arange(n_add, device=dev)
        C0, A0, B0, U0, V0 = 2, 2 + n_add, 2 + 2 * n_add, 2 + 3 * n_add, 2 + 4 * n_add     # variables c_t, a_t (even t), b_t, and the free wires 2, 3
        odd = (t & 1) == 1
        var[0, 2:2 + n_add] = torch.where(odd, C0 + t - 1, A0 + t)
        var[1, 2:2 + n_add] = B0 + t
        var[2, 2:2 + n_add] = U0 + t
        var[3, 2:2 + n_add] = V0 + t
        var[4, 2:2 + n_add] = C0 + t
        rnd = torch.from_numpy(random_fr_mont(c, 4 * n_add, seed=dense_seed).view(np.int64)).to(dev).reshape(4, n_add, 4)
        a_even, b = rnd[0], rnd[1]
        c_even = poly.lincomb(c, [(1, a_even.contiguous()), (1, b.contiguous())])            # c_t = a_t + b_t where a_t is free ...
        prev = torch.roll(c_even, 1, 0)
        c_odd = poly.lincomb(c, [(1, prev.contiguous()), (1, b.contiguous())])              # ... and c_t = c_(t-1) + b_t on odd rows (t - 1 is even)
        cc = torch.where(odd[:, None], c_odd, c_even)
        wit = torch.zeros((2 + 5 * n_add, 4), dtype=torch.int64, device=dev)
        wit[1:2] = torch.from_numpy(fr_to_mont(c, [1]).view(np.int64)).to(dev)
        wit[C0:C0 + n_add], wit[A0:A0 + n_add], wit[B0:B0 + n_add] = cc, a_even, b
        wit[U0:U0 + n_add], wit[V0:V0 + n_add] = rnd[2], rnd[3]
    wire_values = wit[var.reshape(-1)].reshape(W, n, 4).contiguous()
    one = torch.from_numpy(fr_to_mont(c, [1]).view(np.int64)).to(dev)[0]
    nsel = 14 if ultra else 13
    sel = torch.zeros((nsel, n, 4), dtype=torch.int64, device=dev)
    sel[0, 2:2 + n_add] = one                                # AdditionGate: q_lc = [1, 1, 0, 0], q_o = 1 (gates/arithmetic.rs:38-51)
    sel[1, 2:2 + n_add] = one
    sel[10, 0:2 + n_add] = one                               # q_o of the constant and addition gates
    sel[11, 1] = one                                         # ConstantGate(1): q_c = 1
    # wire permutation (constraint_system.rs:743-778): occurrences of a variable in (wire, row) order form a cycle
    flat = var.reshape(-1)
    order = torch.argsort(flat, stable=True)
    sv = flat[order]
    first = torch.ones_like(sv, dtype=torch.bool)
    first[1:] = sv[1:] != sv[:-1]
    last = torch.ones_like(first)
    last[:-1] = first[1:]
    idx = torch.arange(flat.numel(), device=dev)
    group_start = torch.cummax(torch.where(first, idx, torch.zeros_like(idx)), 0).values
    nxt = torch.where(last, order[group_start], torch.roll(order, -1))
    perm = torch.empty_like(order)
    perm[order] = nxt
    k = _rng.compute_coset_representatives(c, W, n)
    dom = Radix2EvaluationDomain(c, log_n)
    ext = torch.zeros((W, n, 4), dtype=torch.int64, device=dev)                            # id[i*n + j] = k_i * w^j  (:913-931)
    ext[:, 1] = torch.from_numpy(fr_to_mont(c, k).view(np.int64)).to(dev)                  # the polynomial k_i * X ...
    dom.fft_in_place(ext)                                                                  # ... evaluated on H
    sigma = ext.reshape(-1, 4)[perm].reshape(W, n, 4).contiguous()
    tables = None
    if ultra:
        tables = torch.zeros((4, n, 4), dtype=torch.int64, device=dev)
        rt = torch.zeros((1 << range_bit_len, 4), dtype=torch.int64, device=dev)
        rt[:, 0] = torch.arange(1 << range_bit_len, device=dev)
        tables[0, :1 << range_bit_len] = _to_mont_dev(c, rt)                               # compute_range_table (:1423-1438)
    return BenchCircuit(c, plonk_type, n, k, wire_values, sel, sigma, torch.zeros((n, 4), dtype=torch.int64, device=dev), [], tables,
                        witness=wit, wire_variables=var.to(torch.int32).contiguous())


@dataclass
class HostWitness:
    """A witness that starts every proof in HOST memory, the way the reference holds it (`self.witness`, gathered per wire by
    compute_wire_polynomials, relation/src/constraint_system.rs:1225-1247): `witness` = (n_vars, 4) int64 CPU tensor (page-locked for
    asynchronous DMA), `wire_variables` = (W, n) int32 CUDA tensor, resident circuit structure.  Passed to `prove` in place of
    `wire_values`: n_vars x 32 B cross PCIe and the gather runs on the device (mzk_plonk_gather_witness_dev)."""
    witness: object
    wire_variables: object


LAGRANGE_MIN_DOMAIN = 1 << 18          # preprocess(lagrange=None): round 1 over the Lagrange basis from this domain size on (re-measured in round 5 with the
                                       # fused small batches split over more threads: at 2^13 / 2^15 gates the coefficient-form commits are 0.28 / 0.2 ms
                                       # FASTER than the heavy-bucket paths small values take, equal at 2^17; 2^20: 2.6 against 8.0 ms)
LAGRANGE_SAMPLE = 2048                 # ... when a sample of the witness shows small values (below)


def witness_is_small(curve, wire_values) -> bool:
    """Does round 1 gain from the Lagrange-basis key?  It does when the wire VALUES are mostly small numbers (flags, counters, 64-bit
    amounts: their high digits cost the MSM nothing); on a dense witness the key and its fixed-base table are 1.75 GB of HBM (2^20 gates,
    BLS12-381) and 0.5 s of set-up for nothing.  Decided from LAGRANGE_SAMPLE strided values: small = below 2^64 for at
    least half of them.  wire_values: (W, n, 4) Montgomery limbs (tensor on either side) or a HostWitness (its witness vector)."""
    c = _curve(curve)
    t = wire_values.witness if isinstance(wire_values, HostWitness) else wire_values
    flat = t.reshape(-1, 4)
    total = int(flat.shape[0])
    if total == 0:
        return False
    step = max(1, total // LAGRANGE_SAMPLE)
    sample = flat[::step][:LAGRANGE_SAMPLE].cpu().numpy().view(np.uint64)
    small = sum(1 for v in fr_from_mont(c, sample) if v < (1 << 64))
    return 2 * small >= sample.shape[0]


def preprocess(commit_key: kzg.UnivariateProverParam, circuit: BenchCircuit, lagrange: bool | None = None, lagrange_ck=None,
               comm=None) -> _prover.TurboPlonkProver:
    """snark.rs:529-617: interpolate selectors, sigmas (and Plookup tables) and hand the coefficient forms to the library
    (mzk_prover_create keeps them, their evaluations on the needed residue classes and the workspace of one proof in HBM).  The
    verifying-key commitments come on demand from `TurboPlonkProver.vk_commitments()`.
    lagrange: also derive the commit key over the Lagrange basis of the gate domain from the SRS's points
    (kzg.UnivariateProverParam.lagrange_key) -- round 1 then commits the wires from their VALUES (same commitments).  None: from 2^18
    gates on (below, an MSM is a chain of latencies and small scalars only add over-long buckets to it) AND only when a sample of the
    circuit's witness shows small values (witness_is_small: a dense witness gains nothing from the key).  lagrange_ck: an existing key.
    comm: a sharding.TorchComm -- this process is one rank of a sharded proof and keeps only its point range of the SRS (and of the
    Lagrange key): mzk_srs_slice."""
    c, n = circuit.curve, circuit.n
    if commit_key.length < n + 3:
        raise ValueError("SRS too small: need domain size + 3 powers (srs.rs:88)")          # snark.rs:535-541
    if commit_key.length > n + 3:       # snark.rs:535, 561: the proving key keeps trim(srs_size) = n + 3 powers -- a view of the same registration
        commit_key = commit_key.trim(n + 2)
    dom = Radix2EvaluationDomain(c, n.bit_length() - 1)
    sel = circuit.selector_values.clone()
    sig = circuit.sigma_values.clone()
    dom.ifft_in_place(sel)
    dom.ifft_in_place(sig)
    host = lambda t: t.cpu().numpy().view(np.uint64)
    plookup = None
    if circuit.table_values is not None:
        tab = circuit.table_values.clone()
        dom.ifft_in_place(tab)
        tab_h = host(tab)
        plookup = {name: tab_h[i] for i, name in enumerate(("range_table_poly", "key_table_poly", "table_dom_sep_poly", "q_dom_sep_poly"))}
    if lagrange is None:
        lagrange = n >= LAGRANGE_MIN_DOMAIN and witness_is_small(c, circuit.wire_values)
    lck = lagrange_ck if lagrange_ck is not None else (commit_key.lagrange_key(n) if lagrange else None)
    owned = []                                                   # keys this prover releases with itself
    if comm is not None and comm.world > 1:
        from .sharding import shard_range
        lo, hi = shard_range(n + 3, comm.rank, comm.world)       # ONE partition of the n + 3 powers for every commitment of the proof
        commit_key = commit_key.slice(lo, hi - lo)
        owned.append(commit_key)
        if lck is not None:
            full, lck = lck, lck.slice(lo, hi - lo)
            owned.append(lck)
            if lagrange_ck is None:
                full.release()                                   # derived here: only this rank's range of it stays
    elif lagrange_ck is None and lck is not None:
        owned.append(lck)
    pk = _prover.TurboPlonkProver(c, n, list(host(sel)), list(host(sig)), circuit.k, commit_key, plookup=plookup, lagrange_ck=lck, comm=comm)
    pk.owned_keys = owned
    return pk


def draw_blinders(curve, rng: _rng.ChaChaRng, num_wire_types: int, ultra: bool) -> _prover.Blinders:
    """Every `DensePolynomial::rand` / `F::rand` of one proof, in the order the rounds draw them (prover.rs:79-83, 113-114,
    133-138, 169-180, 947-955): the draws depend on nothing the rounds compute, so they can be taken up front."""
    c = _curve(curve)
    wires = [_rng.dense_poly_rand(c, 1, rng) for _ in range(num_wire_types)]
    h = [_rng.dense_poly_rand(c, 2, rng) for _ in range(2)] if ultra else None
    z = _rng.dense_poly_rand(c, 2, rng)
    pl = _rng.dense_poly_rand(c, 2, rng) if ultra else None
    quot = [_rng.fr_rand(c, rng) for _ in range(num_wire_types - 1)]
    return _prover.Blinders(wires, z, quot, h, pl)


def _g1(c: CurveParams, comm: kzg.Commitment) -> bytes:
    if comm.is_infinity():
        return _transcript.g1_bytes(c, None)
    x, y = fq_from_mont(c, comm.xy)
    return _transcript.g1_bytes(c, (x, y))


def serialize_proof(curve, proof: _prover.ProofCore) -> bytes:
    """`Proof::serialize_compressed` (derive(CanonicalSerialize), structs.rs:59-84): fields in declaration order,
    Vec = u64 LE length + items, Option = one byte + payload, G1 compressed, Fr 32 bytes LE."""
    c = _curve(curve)
    fr = lambda v: _transcript.fr_bytes(c, v)
    vec = lambda items, enc: struct.pack("<Q", len(items)) + b"".join(enc(x) for x in items)
    g1 = lambda cm: _g1(c, cm)
    out = vec(proof.wires_poly_comms, g1) + g1(proof.prod_perm_poly_comm) + vec(proof.split_quot_poly_comms, g1)
    out += g1(proof.opening_proof) + g1(proof.shifted_opening_proof)
    out += vec(proof.wires_evals, fr) + vec(proof.wire_sigma_evals, fr) + fr(proof.perm_next_eval)              # ProofEvaluations :440-450
    if proof.plookup_evals is None:
        out += b"\x00"
    else:                                                                                                        # PlookupProof :208-222
        out += b"\x01" + vec(proof.h_poly_comms, g1) + g1(proof.prod_lookup_poly_comm)
        out += b"".join(fr(proof.plookup_evals[name]) for name in _prover.PLOOKUP_EVALS)                          # PlookupEvaluations :496-541
    return out


def prove(rng: _rng.ChaChaRng, circuit: BenchCircuit, pk: _prover.TurboPlonkProver, extra_transcript_init_msg: bytes | None = None,
          profile: bool = False, witness=None):
    """PlonkKzgSnark::prove (snark.rs:624-651) through the round-level C ABI: returns (ProofCore, compressed proof bytes).
    witness: what to prove from instead of circuit.wire_values (a HostWitness, a host tensor ..)."""
    if (circuit.plonk_type == ULTRA) != pk.ultra:
        raise ValueError("Mismatched Plonk types between the proving key and the circuit")                       # snark.rs:249-254
    if circuit.n != pk.n:
        raise ValueError("proving key domain size %d != expected domain size %d" % (pk.n, circuit.n))           # snark.rs:233-240
    blind = draw_blinders(circuit.curve, rng, circuit.num_wire_types, pk.ultra)
    src = _prover.TranscriptChallenges(pk, circuit.public_input, extra_transcript_init_msg)
    core = pk.prove(circuit.wire_values if witness is None else witness, list(circuit.public_input), src, blind, profile=profile)
    return core, serialize_proof(circuit.curve, core)


def prove_with_link_hint(rng: _rng.ChaChaRng, circuit: BenchCircuit, pk: _prover.TurboPlonkProver, extra_transcript_init_msg: bytes | None = None):
    """PlonkKzgSnark::prove_with_link_hint (snark.rs:81-119): the proof plus the LinkingHint -- the masked wire polynomial that
    carries the proof-linking gates (wire PROOF_LINK_WIRE_IDX, device resident: mzk_prover_poly_dev) and its commitment from round 1."""
    from . import linking
    core, proof_bytes = prove(rng, circuit, pk, extra_transcript_init_msg)
    hint = linking.LinkingHint(pk.poly_dev(linking.PROOF_LINK_WIRE_IDX), core.wires_poly_comms[linking.PROOF_LINK_WIRE_IDX])
    return core, proof_bytes, hint


def draw_batch_blinders(curve, rng: _rng.ChaChaRng, num_wire_types: int, ultras: list):
    """The masking draws of batch_prove_internal in the reference's order (snark.rs:277-360): round 1 of every instance, round
    1.5 of every (UltraPlonk) instance, round 2, round 2.5, then the W - 1 scalars of the one quotient split.
    Returns ([Blinders per instance], quot_blinders)."""
    c = _curve(curve)
    wires = [[_rng.dense_poly_rand(c, 1, rng) for _ in range(num_wire_types)] for _ in ultras]
    h = [[_rng.dense_poly_rand(c, 2, rng) for _ in range(2)] if u else None for u in ultras]
    z = [_rng.dense_poly_rand(c, 2, rng) for _ in ultras]
    pl = [_rng.dense_poly_rand(c, 2, rng) if u else None for u in ultras]
    quot = [_rng.fr_rand(c, rng) for _ in range(num_wire_types - 1)]
    return [_prover.Blinders(wires[i], z[i], quot, h[i], pl[i]) for i in range(len(ultras))], quot


def batch_prove(rng: _rng.ChaChaRng, circuits: list, pks: list, extra_transcript_init_msg: bytes | None = None):
    """PlonkKzgSnark::batch_prove (snark.rs:64-78): one aggregated proof for several instances of one domain size.
    Returns (BatchProofCore, compressed BatchProof bytes)."""
    from . import batch as _batch
    if not circuits:
        raise ValueError("zero number of circuits/proving keys")
    if len(circuits) != len(pks):
        raise ValueError("the number of circuits %d != the number of proving keys %d" % (len(circuits), len(pks)))
    n, W = circuits[0].n, circuits[0].num_wire_types
    for cs, pk in zip(circuits, pks):
        if cs.n != n:
            raise ValueError("circuit domain size %d != expected domain size %d" % (cs.n, n))
        if pk.n != n:
            raise ValueError("proving key domain size %d != expected domain size %d" % (pk.n, n))
        if (cs.plonk_type == ULTRA) != pk.ultra:
            raise ValueError("Mismatched Plonk types between the proving key and the circuit")
        if cs.num_wire_types != W:
            raise ValueError("inconsistent plonk circuit types")
    blinds, quot = draw_batch_blinders(circuits[0].curve, rng, W, [pk.ultra for pk in pks])
    core = _batch.batch_prove(pks, [cs.wire_values for cs in circuits], [list(cs.public_input) for cs in circuits], blinds, quot, extra_transcript_init_msg)
    return core, _batch.serialize_batch_proof(circuits[0].curve, core)
