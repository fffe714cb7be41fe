"""The prover's rounds behind the C ABI (`mzk_prover_*`, include/mzk.h; csrc/prover.hip), driven through ctypes the way a Rust caller
would drive them.  This module keeps what `PlonkKzgSnark::batch_prove_internal` keeps (plonk/src/proof_system/snark.rs:263-431) -- the
challenge sources over the Merlin transcript, the blinding draws' container, the `Proof` -- and hands challenges / blinders in and
commitments / evaluations out, round by round:

    round 1  run_1st_round   prover.rs:72-87     mzk_prover_round1     wire iNTTs, masking, batch_commit, public-input iNTT
    round 1.5 run_plookup_1st_round prover.rs:89-118   round1_5        merged table, sorted vector, h_1 / h_2, batch_commit   (UltraPlonk)
    round 2  run_2nd_round   prover.rs:125-141   round2                permutation grand product, masking, commit
    round 2.5 run_plookup_2nd_round prover.rs:143-183  round2_5        Plookup grand product, masking, commit                    (UltraPlonk)
    round 3  run_3rd_round   prover.rs:192-209   round3                quotient, split, batch_commit (all instances of a batch)
    round 4  compute_evaluations (+ Plookup)     prover.rs:216-299     round4
    round 5  linearisation + opening proofs      prover.rs:302-460     round5

ONE implementation of the rounds (round 5 of this build): the sequencing of NTTs, MSMs, scans and polynomial passes lives in the
library; `TurboPlonkProver` here sequences nothing.  The Python sequencing of the primitives that used to live here is test code now
(tests/mirror_prover.py); both must emit the same bytes (tests/test_native_prover_gpu.py).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

import ctypes as C
import json

from . import kzg, lib as _lib, plonk
from . import transcript as _transcript
from .params import curve as _curve, fr_from_mont, fr_to_mont


PlonkError = plonk.PlonkError


@dataclass
class Blinders:
    """DensePolynomial::rand draws, in the reference's order: W x 2 for the wires,
    (2 x 3 for h_1, h_2,) 3 for z, (3 for the Plookup product,) W - 1 for the split quotient."""
    wires: list
    z: list
    quot: list
    h: list | None = None
    prod_lookup: list | None = None


@dataclass
class ProverChallenges:
    beta: int
    gamma: int
    alpha: int
    zeta: int
    v: int
    tau: int = 0


PLOOKUP_EVALS = ("range_table_eval", "key_table_eval", "table_dom_sep_eval", "q_dom_sep_eval", "h_1_eval", "q_lookup_eval", "prod_next_eval",
                 "range_table_next_eval", "key_table_next_eval", "table_dom_sep_next_eval", "h_1_next_eval", "h_2_next_eval",
                 "q_lookup_next_eval", "w_3_next_eval", "w_4_next_eval")      # field order of PlookupEvaluations, structs.rs:496-541


class FixedChallenges:
    """Challenge source with externally fixed values."""

    def __init__(self, ch: ProverChallenges):
        self.ch = ch

    def after_round1(self, wires_comms):
        return self.ch.tau

    def after_round1_5(self, h_comms):
        return self.ch.beta, self.ch.gamma

    def after_round2(self, z_comm, prod_lookup_comm):
        return self.ch.alpha

    def after_round3(self, split_comms):
        return self.ch.zeta

    def after_round4(self, wires_evals, wire_sigma_evals, perm_next_eval, plookup_evals):
        return self.ch.v


class TranscriptChallenges:
    """Challenge source following batch_prove_internal (snark.rs:263-431) on a StandardTranscript."""

    def __init__(self, prover: "TurboPlonkProver", pub_input, extra_msg: bytes | None = None):
        import copy
        c = prover.curve
        self.c = c
        # the transcript after the label, the verifying key and the public input is the same for every proof of one instance:
        # absorbed once per (extra message, public input), its 200-byte STROBE state copied afterwards (0.25 ms of Python per proof)
        cache = prover.__dict__.setdefault("_transcript_heads", {})
        key = (extra_msg, tuple(int(x) for x in pub_input))
        if key not in cache:
            t = _transcript.StandardTranscript(c, b"PlonkProof")
            if extra_msg is not None:
                t.append_message(b"extra info", extra_msg)
            sel, sig = prover.vk_commitments()
            t.append_vk_and_pub_input(prover.n, len(pub_input), prover.k, [self._pt(x) for x in sel], [self._pt(x) for x in sig], pub_input)
            if len(cache) < 16:
                cache[key] = t
            else:
                cache = None
        self.t = copy.deepcopy(cache[key]) if cache is not None else t
        self.challenges = {}

    def _pt(self, comm: kzg.Commitment):
        if comm.is_infinity():
            return None
        from .params import fq_from_mont
        x, y = fq_from_mont(self.c, comm.xy)
        return (x, y)

    def _squeeze(self, name: str) -> int:
        v = self.t.get_and_append_challenge(name.encode())
        self.challenges[name] = v
        return v

    def after_round1(self, wires_comms):
        self.t.append_commitments(b"witness_poly_comms", [self._pt(cm) for cm in wires_comms])
        return self._squeeze("tau")                            # squeezed even without Plookup (snark.rs:293)

    def after_round1_5(self, h_comms):
        if h_comms is not None:
            self.t.append_commitments(b"h_poly_comms", [self._pt(cm) for cm in h_comms])
        return self._squeeze("beta"), self._squeeze("gamma")

    def after_round2(self, z_comm, prod_lookup_comm):
        self.t.append_commitment(b"perm_poly_comms", self._pt(z_comm))
        if prod_lookup_comm is not None:
            self.t.append_commitment(b"plookup_poly_comms", self._pt(prod_lookup_comm))
        return self._squeeze("alpha")

    def after_round3(self, split_comms):
        self.t.append_commitments(b"quot_poly_comms", [self._pt(cm) for cm in split_comms])
        return self._squeeze("zeta")

    def after_round4(self, wires_evals, wire_sigma_evals, perm_next_eval, plookup_evals):
        for e in wires_evals:
            self.t.append_field_elem(b"wire_evals", e)
        for e in wire_sigma_evals:
            self.t.append_field_elem(b"wire_sigma_evals", e)
        self.t.append_field_elem(b"perm_next_eval", perm_next_eval)
        if plookup_evals is not None:
            self.t.append_plookup_evaluations(plookup_evals)
        return self._squeeze("v")


@dataclass
class ProofCore:
    """Proof (structs.rs:62-84) minus serialisation: commitments as affine x||y limbs, evaluations as ints."""
    wires_poly_comms: list
    prod_perm_poly_comm: kzg.Commitment
    split_quot_poly_comms: list
    opening_proof: kzg.Commitment
    shifted_opening_proof: kzg.Commitment
    wires_evals: list
    wire_sigma_evals: list
    perm_next_eval: int
    timings_ms: dict = field(default_factory=dict)
    h_poly_comms: list | None = None                 # PlookupProof (structs.rs:208-222)
    prod_lookup_poly_comm: kzg.Commitment | None = None
    plookup_evals: dict | None = None


WITNESS_DEV_WIRES, WITNESS_HOST_WIRES, WITNESS_HOST_VECTOR, WITNESS_DEV_VECTOR = 0, 1, 2, 3
ERR_WRONG_QUOTIENT_DEGREE = -9


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


def _check(rc: int, where: str):
    if rc == ERR_WRONG_QUOTIENT_DEGREE:
        L = _lib.load()
        raise PlonkError(L.mzk_last_error().decode(), kind="WrongQuotientPolyDegree")
    _lib.check(rc, where)


class TurboPlonkProver:
    """One `mzk_prover` handle: proving key + the workspace of one proof in flight, on the device (coefficient forms of
    ProvingKey{selectors, sigmas, plookup_pk}, structs.rs:575-590).  lagrange_ck: a Lagrange-basis key of the same SRS and domain
    (kzg.UnivariateProverParam.lagrange_key): round 1 (and 1.5) then commit the wire VALUES.  comm: a sharding.TorchComm -- this prover is
    one rank of a multi-process proof; commit_key (and lagrange_ck) may then be this rank's slice of the SRS."""

    def __init__(self, curve, domain_size: int, selector_polys, sigma_polys, k, commit_key: kzg.UnivariateProverParam, plookup=None,
                 lagrange_ck: kzg.UnivariateProverParam | None = None, comm=None):
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
                                   commit_key.handle, lagrange_ck.handle if lagrange_ck is not None else 0, comm.struct_ptr() if comm is not None else None,
                                   C.byref(h)), "mzk_prover_create")
        self.handle = h.value
        self.comm = comm                                               # (keeps the callbacks alive)
        self._vk = self._pvk = None
        self.timings_ms = {}

    def release(self):
        if self.handle:
            _check(_lib.load().mzk_prover_destroy(self.handle), "mzk_prover_destroy")
            self.handle = 0
        for key in getattr(self, "owned_keys", []):                       # keys snark.preprocess derived for this prover (Lagrange key, SRS slices)
            key.release()
        self.owned_keys = []
        if self.lagrange_ck is not None and self.lagrange_ck.handle == 0:
            self.lagrange_ck = None

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
            if getattr(self, "_vars_of", None) is not wire_values.wire_variables:       # the circuit's variable table goes to the device once
                wv = wire_values.wire_variables
                self.set_wire_variables(wv.cpu().numpy() if torch.is_tensor(wv) else wv, int(w.shape[0]))
                self._vars_of = wire_values.wire_variables
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
        pe = dict(zip(PLOOKUP_EVALS, ev[2 * W:])) if self.ultra else None
        return ev[:W], ev[W:2 * W - 1], ev[2 * W - 1], pe

    def timings(self):
        buf = C.create_string_buffer(2048)
        _check(_lib.load().mzk_prover_timings(self.handle, buf, 2048), "mzk_prover_timings")
        return json.loads(buf.value.decode())

    def poly_dev(self, which: int):
        """A polynomial of the proof in flight as a CUDA tensor (a copy): which = 0 .. W - 1 the masked wire polynomials (wire 0 is the
        `linking_wire_poly` of prove_with_link_hint, snark.rs:81-119), W the permutation product."""
        import torch
        L = _lib.load()
        p, ln = C.c_void_p(), C.c_uint64()
        _check(L.mzk_prover_poly_dev(self.handle, which, C.byref(p), C.byref(ln)), "mzk_prover_poly_dev")
        t = torch.empty((ln.value, 4), dtype=torch.int64, device="cuda")
        _check(L.mzk_dev_copy(C.c_void_p(t.data_ptr()), p, ln.value * 32, None), "mzk_dev_copy")
        torch.cuda.synchronize()
        return t

    def prove(self, wire_values, pub_input, ch, blind: Blinders, profile: bool = False, pi_zero: bool = False) -> ProofCore:
        """One instance: the calls of batch_prove_internal (snark.rs:263-431) with a challenge source (TranscriptChallenges /
        FixedChallenges).  pub_input: values for rows 0.., a (rows, values) pair, or the n-vector of evaluations; pi_zero is accepted for
        the callers of the former Python sequencing and ignored (the library decides from the data)."""
        src = FixedChallenges(ch) if isinstance(ch, ProverChallenges) else ch
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
        return ProofCore(wires_comms, z_comm, split_comms, open_comms[0], open_comms[1], wires_evals, wire_sigma_evals, perm_next_eval, tm,
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
