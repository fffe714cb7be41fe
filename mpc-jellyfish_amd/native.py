"""The prover's rounds behind the C ABI (`mzk_prover_*`, include/mzk.h), driven through ctypes the way a Rust caller would drive
them: this module keeps what `PlonkKzgSnark::batch_prove_internal` keeps (plonk/src/proof_system/snark.rs:263-431) -- the
transcript, the blinding draws, the `Proof` -- and hands challenges / blinders in and commitments / evaluations out, round by round:

    Prover::run_1st_round .. compute_opening_proofs      prover.rs:72-419      NativeProver.prove
    the same over several instances (batch_prove)        snark.rs:64-78        batch_prove

`prover.TurboPlonkProver` is the Python mirror that sequences the library's primitives itself; this class sequences nothing: the
round logic lives in the library (csrc/prover.hip).  Both must emit the same bytes (tests/test_native_prover_gpu.py)."""
from __future__ import annotations

import ctypes as C
import json

import numpy as np

from . import kzg, lib as _lib, prover as _prover
from .params import curve as _curve, fr_from_mont, fr_to_mont

WITNESS_DEV_WIRES, WITNESS_HOST_WIRES, WITNESS_HOST_VECTOR, WITNESS_DEV_VECTOR = 0, 1, 2, 3
ERR_WRONG_QUOTIENT_DEGREE = -9


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


def _check(rc: int, where: str):
    if rc == ERR_WRONG_QUOTIENT_DEGREE:
        L = _lib.load()
        raise _prover.PlonkError(L.mzk_last_error().decode(), kind="WrongQuotientPolyDegree")
    _lib.check(rc, where)


class NativeProver:
    """One `mzk_prover` handle: proving key + the workspace of one proof in flight, on the device.  Same constructor arguments as
    prover.TurboPlonkProver (coefficient forms of ProvingKey{selectors, sigmas, plookup_pk}, structs.rs:575-590)."""

    def __init__(self, curve, domain_size: int, selector_polys, sigma_polys, k, commit_key: kzg.UnivariateProverParam, plookup=None,
                 lagrange_ck: kzg.UnivariateProverParam | None = None):
        from . import plonk
        self.curve = c = _curve(curve)
        self.n = domain_size
        self.log_n = domain_size.bit_length() - 1
        self.k = list(k)
        self.ck = commit_key
        self.lagrange_ck = lagrange_ck
        self.ultra = plookup is not None
        self.W = len(sigma_polys)
        self.nsel = len(selector_polys)
        assert commit_key.offset == 0, "the prover takes the SRS handle itself: a trimmed view must start at power 0"
        pad = lambda p: np.concatenate([np.asarray(p, dtype=np.uint64).reshape(-1, 4),
                                        np.zeros((domain_size - np.asarray(p).reshape(-1, 4).shape[0], 4), dtype=np.uint64)])
        sel = np.ascontiguousarray(np.stack([pad(p) for p in selector_polys]))
        sig = np.ascontiguousarray(np.stack([pad(p) for p in sigma_polys]))
        tab = np.ascontiguousarray(np.stack([pad(plookup[x]) for x in plonk.PLOOKUP_TABLE_POLYS])) if self.ultra else None
        kk = fr_to_mont(c, self.k)
        L = _lib.ensure_init()
        h = C.c_uint64()
        _check(L.mzk_prover_create(c.curve_id, self.log_n, self.W, _ptr(sel), _ptr(sig), _ptr(tab) if self.ultra else None, domain_size, _ptr(kk),
                                   commit_key.handle, lagrange_ck.handle if lagrange_ck is not None else 0, None, C.byref(h)), "mzk_prover_create")
        self.handle = h.value
        self._vk = self._pvk = None
        self.timings_ms = {}

    def release(self):
        if self.handle:
            _check(_lib.load().mzk_prover_destroy(self.handle), "mzk_prover_destroy")
            self.handle = 0

    def _commitments(self):
        if self._vk is None:
            c, L = self.curve, _lib.load()
            xy = np.zeros((self.nsel + self.W, 2, c.fq_limbs), dtype=np.uint64)
            pxy = np.zeros((4, 2, c.fq_limbs), dtype=np.uint64)
            _check(L.mzk_prover_vk_commitments(self.handle, _ptr(xy), _ptr(pxy) if self.ultra else None), "mzk_prover_vk_commitments")
            self._vk = ([kzg.Commitment(c, xy[i]) for i in range(self.nsel)], [kzg.Commitment(c, xy[self.nsel + i]) for i in range(self.W)])
            self._pvk = [kzg.Commitment(c, p) for p in pxy]
        return self._vk, self._pvk

    def vk_commitments(self):
        """selector_comms, sigma_comms of the verifying key (preprocess, snark.rs:562-594)"""
        return self._commitments()[0]

    def plookup_vk_commitments(self):
        assert self.ultra
        return self._commitments()[1]

    def set_wire_variables(self, wire_variables, n_vars: int):
        """wire_variables: (W, n) uint32 (host); witness kinds HOST_VECTOR / DEV_VECTOR gather through it on the device"""
        v = np.ascontiguousarray(wire_variables, dtype=np.uint32).reshape(self.W, self.n)
        _check(_lib.load().mzk_prover_set_wire_variables(self.handle, _ptr(v), n_vars), "mzk_prover_set_wire_variables")

    def hbm_bytes(self):
        a, b, w = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(_lib.load().mzk_prover_hbm_bytes(self.handle, C.byref(a), C.byref(b), C.byref(w)), "mzk_prover_hbm_bytes")
        return {"fixed_coefficient_forms": a.value, "proving_key_evaluations": b.value, "prover_workspace": w.value}

    # ---- argument plumbing ---------------------------------------------------------------------------------------------------
    def _mont(self, ints):
        return fr_to_mont(self.curve, [int(x) % self.curve.r for x in ints])

    def _witness_args(self, wire_values):
        """-> (kind, pointer, length, keep-alive)"""
        import torch
        if hasattr(wire_values, "wire_variables"):                       # snark.HostWitness: the witness vector, gathered on the device
            w = wire_values.witness
            if torch.is_tensor(w) and w.is_cuda:
                return WITNESS_DEV_VECTOR, C.c_void_p(w.data_ptr()), int(w.shape[0]), w
            w = w if torch.is_tensor(w) else torch.from_numpy(np.ascontiguousarray(w).view(np.int64))
            return WITNESS_HOST_VECTOR, C.c_void_p(w.data_ptr()), int(w.shape[0]), w
        if torch.is_tensor(wire_values):
            w = wire_values.contiguous()
            kind = WITNESS_DEV_WIRES if w.is_cuda else WITNESS_HOST_WIRES
            return kind, C.c_void_p(w.data_ptr()), self.W * self.n, w
        w = np.ascontiguousarray(wire_values, dtype=np.uint64).reshape(self.W, self.n, 4)
        return WITNESS_HOST_WIRES, _ptr(w), self.W * self.n, w

    def _pub_args(self, pub_input):
        """pub_input: None / [] (no public input), a list of values for rows 0.. (where finalisation puts the IO gates), a
        (rows, values) pair, or the n-vector of the Python mirror (numpy Montgomery limbs or a tensor): -> (rows | None, values, count)"""
        import torch
        if pub_input is None:
            return None, None, 0
        if isinstance(pub_input, tuple):
            rows, vals = pub_input
            return np.ascontiguousarray(rows, dtype=np.uint64), self._mont(vals), len(vals)
        if torch.is_tensor(pub_input):
            pub_input = pub_input.cpu().numpy().view(np.uint64)
        if isinstance(pub_input, np.ndarray):
            v = np.ascontiguousarray(pub_input, dtype=np.uint64).reshape(-1, 4)
            rows = np.flatnonzero(v.any(axis=1)).astype(np.uint64)
            return (rows, np.ascontiguousarray(v[rows]), int(rows.shape[0])) if rows.size else (None, None, 0)
        vals = list(pub_input)
        return (None, self._mont(vals), len(vals)) if vals else (None, None, 0)

    def _points(self, count):
        return np.zeros((count, 2, self.curve.fq_limbs), dtype=np.uint64)

    def _comms(self, xy):
        return [kzg.Commitment(self.curve, p) for p in xy]

    # ---- the rounds ----------------------------------------------------------------------------------------------------------
    def round1(self, wire_values, pub_input, blind_wires):
        L = _lib.load()
        kind, wptr, wlen, keep = self._witness_args(wire_values)
        rows, vals, n_pub = self._pub_args(pub_input)
        bl = self._mont([b for row in blind_wires for b in row])
        out = self._points(self.W)
        _check(L.mzk_prover_round1(self.handle, kind, wptr, wlen, _ptr(rows) if rows is not None else None, _ptr(vals) if n_pub else None, n_pub,
                                   _ptr(bl), _ptr(out)), "mzk_prover_round1")
        del keep
        return self._comms(out)

    def round1_5(self, tau, blind_h):
        bl, t = self._mont([b for row in blind_h for b in row]), self._mont([tau])      # (named: the arrays must outlive the call)
        out = self._points(2)
        _check(_lib.load().mzk_prover_round1_5(self.handle, _ptr(t), _ptr(bl), _ptr(out)), "mzk_prover_round1_5")
        return self._comms(out)

    def round2(self, beta, gamma, blind_z):
        out, b, g, bl = self._points(1), self._mont([beta]), self._mont([gamma]), self._mont(blind_z)
        _check(_lib.load().mzk_prover_round2(self.handle, _ptr(b), _ptr(g), _ptr(bl), _ptr(out)), "mzk_prover_round2")
        return self._comms(out)[0]

    def round2_5(self, blind_pl):
        out, bl = self._points(1), self._mont(blind_pl)
        _check(_lib.load().mzk_prover_round2_5(self.handle, _ptr(bl), _ptr(out)), "mzk_prover_round2_5")
        return self._comms(out)[0]

    def round4(self, zeta):
        c, W = self.curve, self.W
        cnt = 2 * W + (15 if self.ultra else 0)
        out, z = np.zeros((cnt, 4), dtype=np.uint64), self._mont([zeta])
        _check(_lib.load().mzk_prover_round4(self.handle, _ptr(z), _ptr(out)), "mzk_prover_round4")
        ev = fr_from_mont(c, out)
        pe = dict(zip(_prover.PLOOKUP_EVALS, ev[2 * W:])) if self.ultra else None
        return ev[:W], ev[W:2 * W - 1], ev[2 * W - 1], pe

    def timings(self):
        buf = C.create_string_buffer(2048)
        _check(_lib.load().mzk_prover_timings(self.handle, buf, 2048), "mzk_prover_timings")
        return json.loads(buf.value.decode())

    def prove(self, wire_values, pub_input, ch, blind: _prover.Blinders, profile: bool = False) -> _prover.ProofCore:
        """One instance: the calls of batch_prove_internal (snark.rs:263-431) with the challenge source of the Python mirror
        (prover.TranscriptChallenges / FixedChallenges)."""
        src = _prover.FixedChallenges(ch) if isinstance(ch, _prover.ProverChallenges) else ch
        L = _lib.load()
        _check(L.mzk_prover_profile(self.handle, 1 if profile else 0), "mzk_prover_profile")
        wires_comms = self.round1(wire_values, pub_input, blind.wires)
        tau = src.after_round1(wires_comms)
        h_comms = self.round1_5(tau, blind.h) if self.ultra else None
        beta, gamma = src.after_round1_5(h_comms)
        z_comm = self.round2(beta, gamma, blind.z)
        pl_comm = self.round2_5(blind.prod_lookup) if self.ultra else None
        alpha = src.after_round2(z_comm, pl_comm)
        split_comms = round3([self], alpha, blind.quot)
        zeta = src.after_round3(split_comms)
        wires_evals, wire_sigma_evals, perm_next_eval, pe = self.round4(zeta)
        v_ch = src.after_round4(wires_evals, wire_sigma_evals, perm_next_eval, pe)
        open_comms = round5([self], v_ch)
        self.last_challenges = {"tau": tau, "beta": beta, "gamma": gamma, "alpha": alpha, "zeta": zeta, "v": v_ch}
        tm = self.timings() if profile else {}
        return _prover.ProofCore(wires_comms, z_comm, split_comms, open_comms[0], open_comms[1], wires_evals, wire_sigma_evals, perm_next_eval, tm,
                                 h_comms, pl_comm, pe)


def round3(provers, alpha, blind_quot):
    """run_3rd_round over all instances (prover.rs:192-209): one quotient, split, W commitments"""
    p0 = provers[0]
    hs = (C.c_uint64 * len(provers))(*[p.handle for p in provers])
    out, a, bl = p0._points(p0.W), p0._mont([alpha]), p0._mont(blind_quot)
    _check(_lib.load().mzk_prover_round3(hs, len(provers), _ptr(a), _ptr(bl), _ptr(out)), "mzk_prover_round3")
    return p0._comms(out)


def round5(provers, v_ch):
    """linearisation polynomial + compute_opening_proofs over all instances (prover.rs:302-460): two commitments"""
    p0 = provers[0]
    hs = (C.c_uint64 * len(provers))(*[p.handle for p in provers])
    out, v = p0._points(2), p0._mont([v_ch])
    _check(_lib.load().mzk_prover_round5(hs, len(provers), _ptr(v), _ptr(out)), "mzk_prover_round5")
    return p0._comms(out)


def preprocess(commit_key: kzg.UnivariateProverParam, circuit, lagrange: bool | None = None, lagrange_ck=None) -> NativeProver:
    """snark.preprocess for the native prover: interpolate selectors, sigmas (and tables) and hand the coefficient forms over.
    lagrange_ck: an existing Lagrange-basis key of this SRS and domain (else derived when `lagrange`: None = from 2^13 gates on when a
    sample of the circuit's witness shows small values, snark.witness_is_small)."""
    from . import snark
    from .domain import Radix2EvaluationDomain
    c, n = circuit.curve, circuit.n
    if commit_key.length < n + 3:
        raise ValueError("SRS too small: need domain size + 3 powers (srs.rs:88)")
    if commit_key.length > n + 3:
        commit_key = commit_key.trim(n + 2)
    dom = Radix2EvaluationDomain(c, n.bit_length() - 1)
    host = lambda t: t.cpu().numpy().view(np.uint64)
    sel, sig = circuit.selector_values.clone(), circuit.sigma_values.clone()
    dom.ifft_in_place(sel)
    dom.ifft_in_place(sig)
    plookup = None
    if circuit.table_values is not None:
        tab = circuit.table_values.clone()
        dom.ifft_in_place(tab)
        tab_h = host(tab)
        plookup = {name: tab_h[i] for i, name in enumerate(("range_table_poly", "key_table_poly", "table_dom_sep_poly", "q_dom_sep_poly"))}
    if lagrange is None:
        lagrange = n >= snark.LAGRANGE_MIN_DOMAIN and snark.witness_is_small(c, circuit.wire_values)
    lck = lagrange_ck if lagrange_ck is not None else (commit_key.lagrange_key(n) if lagrange else None)
    return NativeProver(c, n, list(host(sel)), list(host(sig)), circuit.k, commit_key, plookup=plookup, lagrange_ck=lck)


def prove(rng, circuit, pk: NativeProver, extra_transcript_init_msg: bytes | None = None, profile: bool = False, witness=None):
    """PlonkKzgSnark::prove (snark.rs:624-651) through the round-level ABI: returns (ProofCore, compressed proof bytes)."""
    from . import snark
    if (circuit.plonk_type == snark.ULTRA) != pk.ultra:
        raise ValueError("Mismatched Plonk types between the proving key and the circuit")
    if circuit.n != pk.n:
        raise ValueError("proving key domain size %d != expected domain size %d" % (pk.n, circuit.n))
    blind = snark.draw_blinders(circuit.curve, rng, circuit.num_wire_types, pk.ultra)
    src = _prover.TranscriptChallenges(pk, circuit.public_input, extra_transcript_init_msg)
    core = pk.prove(circuit.wire_values if witness is None else witness, list(circuit.public_input), src, blind, profile=profile)
    return core, snark.serialize_proof(circuit.curve, core)


def batch_prove(provers, wire_values: list, pub_inputs: list, blinds: list, quot_blinders: list, extra_transcript_init_msg: bytes | None = None):
    """PlonkKzgSnark::batch_prove (snark.rs:64-78, 201-469) over K native handles: rounds 1 - 2.5 and 4 per instance, rounds 3 and 5
    once over all handles.  pub_inputs[k]: instance k's public input as ints, on rows 0.. of its circuit.  Returns batch.BatchProofCore."""
    from . import batch as _batch, transcript as _transcript
    if not provers:
        raise ValueError("zero number of circuits/proving keys")
    if not (len(provers) == len(wire_values) == len(pub_inputs) == len(blinds)):
        raise ValueError("the number of circuits != the number of proving keys")
    p0 = provers[0]
    c = p0.curve
    pt = lambda cm: _batch._pt(c, cm)
    t = _transcript.StandardTranscript(c, b"PlonkProof")
    if extra_transcript_init_msg is not None:
        t.append_message(b"extra info", extra_transcript_init_msg)
    for p, pub in zip(provers, pub_inputs):
        sel, sig = p.vk_commitments()
        t.append_vk_and_pub_input(p.n, len(pub), p.k, [pt(x) for x in sel], [pt(x) for x in sig], pub)
    wires_vec = []
    for k, p in enumerate(provers):
        wires_vec.append(p.round1(wire_values[k], list(pub_inputs[k]), blinds[k].wires))
        t.append_commitments(b"witness_poly_comms", [pt(x) for x in wires_vec[-1]])
    tau = t.get_and_append_challenge(b"tau")
    h_vec = []
    for k, p in enumerate(provers):
        h_vec.append(p.round1_5(tau, blinds[k].h) if p.ultra else None)
        if h_vec[-1] is not None:
            t.append_commitments(b"h_poly_comms", [pt(x) for x in h_vec[-1]])
    beta = t.get_and_append_challenge(b"beta")
    gamma = t.get_and_append_challenge(b"gamma")
    z_vec = []
    for k, p in enumerate(provers):
        z_vec.append(p.round2(beta, gamma, blinds[k].z))
        t.append_commitment(b"perm_poly_comms", pt(z_vec[-1]))
    pl_vec = []
    for k, p in enumerate(provers):
        pl_vec.append(p.round2_5(blinds[k].prod_lookup) if p.ultra else None)
        if pl_vec[-1] is not None:
            t.append_commitment(b"plookup_poly_comms", pt(pl_vec[-1]))
    alpha = t.get_and_append_challenge(b"alpha")
    split_comms = round3(provers, alpha, quot_blinders)
    t.append_commitments(b"quot_poly_comms", [pt(x) for x in split_comms])
    zeta = t.get_and_append_challenge(b"zeta")
    evals_vec, pes = [], []
    for p in provers:
        we, se, zn, pe = p.round4(zeta)
        for e in we:
            t.append_field_elem(b"wire_evals", e)
        for e in se:
            t.append_field_elem(b"wire_sigma_evals", e)
        t.append_field_elem(b"perm_next_eval", zn)
        evals_vec.append((we, se, zn))
        pes.append(pe)
    for pe in pes:
        if pe is not None:
            t.append_plookup_evaluations(pe)
    v = t.get_and_append_challenge(b"v")
    open_comms = round5(provers, v)
    plookup_vec = [None if pe is None else (h, pl, pe) for pe, h, pl in zip(pes, h_vec, pl_vec)]
    return _batch.BatchProofCore(wires_vec, z_vec, evals_vec, plookup_vec, split_comms, open_comms[0], open_comms[1],
                                 {"tau": tau, "beta": beta, "gamma": gamma, "alpha": alpha, "zeta": zeta, "v": v})
