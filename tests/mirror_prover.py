"""tests/mirror_prover.py -- TEST CODE: the primitive-level sequencing of the prover's rounds in Python.

Until round 4 this was the product's `prover.TurboPlonkProver`: every NTT, MSM, pointwise pass, scan and polynomial operation of
`PlonkKzgSnark::batch_prove_internal` (plonk/src/proof_system/snark.rs:201-469) called one by one through the library's primitive
entry points (mzk_*_dev).  The product now has ONE implementation of the rounds -- csrc/prover.hip behind `mzk_prover_*`, of which
`mpc-jellyfish_amd/prover.py` is a thin ctypes client -- and this file stays as a second, independent sequencing of the same
primitives for the tests: it exposes every intermediate polynomial of a proof (`prover.last[...]`: what tests/test_prover_gpu.py and
test_ultra_gpu.py compare with the schoolbook oracle), the torch.distributed sharding of the commit path (`ShardedCommitter`, with an
injectable MSM for the CPU tests) and the per-stage hooks tools/scale_model.py times.  Not imported by the product.

    round 1  run_1st_round   prover.rs:72-87     wire iNTTs, masking, batch_commit, public-input iNTT
    round 1.5 run_plookup_1st_round prover.rs:89-118   merged table, sorted vector, h_1 / h_2, batch_commit      (UltraPlonk)
    round 2  run_2nd_round   prover.rs:125-141   permutation grand product, masking, commit
    round 2.5 run_plookup_2nd_round prover.rs:143-183  Plookup grand product, masking, commit                       (UltraPlonk)
    round 3  run_3rd_round   prover.rs:192-209   quotient (coset NTTs + fused kernel + coset iNTT), split, batch_commit
    round 4  compute_evaluations (+ compute_plookup_evaluations)   prover.rs:216-299
    round 5  linearisation + opening proofs      prover.rs:302-358, 362-460, 490-509, 963-1112
"""
from __future__ import annotations

import numpy as np

import mpc_jellyfish_amd as mj
from importlib import import_module

kzg, plonk, poly = import_module("mpc-jellyfish_amd.kzg"), mj.plonk, mj.poly
_transcript = mj.transcript
Radix2EvaluationDomain = mj.Radix2EvaluationDomain
_P = mj.prover
Blinders, ProverChallenges, FixedChallenges, TranscriptChallenges, ProofCore, PLOOKUP_EVALS = (
    _P.Blinders, _P.ProverChallenges, _P.FixedChallenges, _P.TranscriptChallenges, _P.ProofCore, _P.PLOOKUP_EVALS)
PlonkError = plonk.PlonkError
CurveParams = mj.params.CurveParams
_curve, fr_from_mont, fr_to_mont = mj.params.curve, mj.params.fr_from_mont, mj.params.fr_to_mont
_lib = import_module("mpc-jellyfish_amd.lib")
_sharding = mj.sharding


class _Evals:
    """Evaluations of device polynomials at up to two points, collected and finished together: ONE library call and one wait per round
    (mzk_poly_eval_many_dev)."""

    def __init__(self, prover):
        self.p, self.vals, self.jobs, self.count, self.xs = prover, [], [], 0, []

    def _which(self, x: int) -> int:
        if x not in self.xs:
            assert len(self.xs) < 2, "two evaluation points per round"
            self.xs.append(x)
        return self.xs.index(x)

    def _push(self, polys, length, x, offset=0):
        batch = 1 if polys.dim() == 2 else polys.shape[0]
        self.jobs.append((polys, length, self._which(x), offset))       # (keeps gathered copies alive until finish)
        h = (self.count, batch)
        self.count += batch
        return h

    def add(self, polys, x: int, length: int | None = None):
        return self._push(polys, length, x)

    def _run(self):
        return [v for job in poly.evaluate_many(self.p.curve, self.jobs, self.xs) for v in job] if self.jobs else []

    def finish(self):
        self.vals = self._run()

    def get(self, h):
        return self.vals[h[0]:h[0] + h[1]]


class _RangeEvals(_Evals):
    """The same over several ranks: every rank evaluates its coefficient range [lo, hi) of each polynomial --
    sum_{j in range} c_j x^j = x^lo * (the range read as a polynomial of its own) -- and ONE all-gather of the partial values
    (32 bytes each) at the end of the round gives every rank all the sums."""

    def __init__(self, prover):
        super().__init__(prover)
        self.lo, self.hi = prover.committer.point_range()
        self.scale = []

    def add(self, polys, x: int, length: int | None = None):
        r = self.p.curve.r
        stride = polys.shape[0] if polys.dim() == 2 else polys.shape[1]
        L = stride if length is None else length
        a, b = min(self.lo, L), min(self.hi, L)
        h = self._push(polys, max(b - a, 0), x, a if b > a else 0)
        self.scale += [pow(x, a, r)] * h[1]
        return h

    def finish(self):
        r = self.p.curve.r
        mine = [v * s % r for v, s in zip(self._run(), self.scale)]
        every = self.p.committer.all_gather_fr(mine)
        self.vals = [sum(col) % r for col in zip(*every)]


class TurboPlonkProver:
    """Holds a proving key on the device: coefficient forms (for rounds 4-5), the resident coset
    evaluations (round 3) and the commit key.  With `plookup` (the four table polynomials of
    PlookupProvingKey) it is the UltraPlonk prover: 14 selectors, 6 wire types."""

    def __init__(self, curve, domain_size: int, selector_polys, sigma_polys, k, commit_key: kzg.UnivariateProverParam, plookup=None,
                 quotient_classes=None, quotient_gather=None, quotient_shard=None):
        """quotient_classes / quotient_gather: the coset-chunked quotient of SURVEY.md 8(e).3.  The quotient is evaluated on the
        residue classes plonk.quotient_classes_needed(W, n) only (6 of 8 for TurboPlonk, 7 for UltraPlonk: its degree needs no more).
        quotient_classes = None (default): this GPU evaluates all of them; "whole": the un-chunked path over all 8n points (kept
        for comparison).  quotient_shard = (rank, world): this rank keeps sharding.class_range(rank, world, needed) -- possibly no
        class at all (8 GPUs, 6 classes) -- and `quotient_gather(local, n_classes) -> (n_classes, n, 4)` performs the one exchange
        (sharding.gather_quotient_classes)."""
        import torch
        self.curve: CurveParams = _curve(curve)
        self.n = domain_size
        self.log_n = domain_size.bit_length() - 1
        self.k = list(k)
        self.ck = commit_key
        self.ultra = plookup is not None
        self.committer = None                        # set to a sharding.ShardedCommitter for multi-GPU commits
        self.lagrange_ck = None                      # kzg.UnivariateProverParam.gen_lagrange_srs_for_testing(...): round 1 commits from the wire VALUES
        self.identity_check = True                   # check_quotient_identity at the end of a proof (tools/scale_model.py times rank 0's share of a
                                                     # multi-rank proof with stand-in exchanges and turns it off)
        self.range_mode = True                       # several ranks: rounds 4 and 5 work on this rank's coefficient range only
        self.W = len(sigma_polys)
        self.nsel = len(selector_polys)
        self.quotient_gather = quotient_gather
        self.W = len(sigma_polys)
        self.classes_needed = plonk.quotient_classes_needed(self.W, domain_size)
        if quotient_shard is not None:
            quotient_classes = _sharding.class_range(quotient_shard[0], quotient_shard[1], len(self.classes_needed))
        elif quotient_classes is None:
            quotient_classes = self.classes_needed
        elif isinstance(quotient_classes, str):
            assert quotient_classes == "whole"
            quotient_classes = None
        self.own_classes = None if quotient_classes is None else list(quotient_classes)
        # a rank that owns no class still registers one (the key cannot be empty); what it computes there is dropped
        resident = quotient_classes if quotient_classes is None or len(quotient_classes) else [self.classes_needed[-1]]
        self.pk = plonk.ProvingKeyDevice.register(self.curve, domain_size, selector_polys, sigma_polys, k, plookup, classes=resident)
        pad = lambda p: np.concatenate([np.asarray(p, dtype=np.uint64).reshape(-1, 4),
                                        np.zeros((domain_size - np.asarray(p).reshape(-1, 4).shape[0], 4), dtype=np.uint64)])
        tabs = [plookup[x] for x in plonk.PLOOKUP_TABLE_POLYS] if self.ultra else []
        self.fixed = torch.from_numpy(np.stack([pad(p) for p in list(selector_polys) + list(sigma_polys) + tabs]).view(np.int64)).cuda()
        self.sigma0 = self.nsel                      # row of sigma_0 in self.fixed
        self.tab0 = self.nsel + self.W               # rows of range, key, table_dom_sep, q_dom_sep
        self.domain = Radix2EvaluationDomain(self.curve, self.log_n)
        self.w_n = pow(self.curve.fr_generator, (self.curve.r - 1) >> self.log_n, self.curve.r)
        # per-proof workspace, allocated once (the round-3 slab alone is (W + 2 [+ 3]) x 8n x 32 B: 1.9 GB at n = 2^20):
        # the caching allocator would otherwise re-acquire gigabytes per proof
        rows = self.W + 2 + (3 if self.ultra else 0)
        dev = self.fixed.device
        # the class-wise quotient reads the coefficient rows without overwriting them: n + 3 columns do, and no second copy is kept
        chunked = self.pk.classes is not None
        self._slab = torch.empty((rows, domain_size + 3 if chunked else 8 * domain_size, 4), dtype=torch.int64, device=dev)
        self._quot = torch.empty((8 * domain_size, 4), dtype=torch.int64, device=dev)
        self._keep = self._slab if chunked else torch.empty((rows, domain_size + 3, 4), dtype=torch.int64, device=dev)
        self._coeff = torch.empty((self.W + 1, domain_size, 4), dtype=torch.int64, device=dev)

    def vk_commitments(self):
        """selector_comms, sigma_comms of the verifying key (preprocess, snark.rs:562-594), cached."""
        if getattr(self, "_vk", None) is None:
            nf = self.nsel + self.W
            jac = kzg.msm_bigint_batch(self.ck, [self.fixed[i] for i in range(nf)], scalars_are_mont=True)
            xy = kzg.jacobian_to_affine(self.curve, jac)
            self._vk = ([kzg.Commitment(self.curve, xy[i]) for i in range(self.nsel)],
                        [kzg.Commitment(self.curve, xy[self.nsel + i]) for i in range(self.W)])
        return self._vk

    def plookup_vk_commitments(self):
        """PlookupVerifyingKey{range_table_comm, key_table_comm, table_dom_sep_comm, q_dom_sep_comm} (snark.rs:575-590)."""
        assert self.ultra
        if getattr(self, "_pvk", None) is None:
            jac = kzg.msm_bigint_batch(self.ck, [self.fixed[self.tab0 + i] for i in range(4)], scalars_are_mont=True)
            self._pvk = [kzg.Commitment(self.curve, xy) for xy in kzg.jacobian_to_affine(self.curve, jac)]
        return self._pvk

    def release(self):
        self.pk.release()
        if self.lagrange_ck is not None:
            self.lagrange_ck.release()
            self.lagrange_ck = None

    def _mask(self, t, rows, blinders):
        """poly + (b_0 + b_1 X + ..)(X^n - 1) on the device rows (prover.rs:463-486), one launch for all of them."""
        poly.mask(self.curve, [t[r] for r in rows], self.n, [list(b) for b in blinders])

    def _ranged(self) -> bool:
        return self.range_mode and self.committer is not None and hasattr(self.committer, "point_range") and self.committer.world() > 1

    def _commit(self, polys):
        """batch_commit (mod.rs:119-131); with `self.committer` (sharding.ShardedCommitter) the MSMs are split by point
        range over the ranks of a process group and every rank obtains the same commitments."""
        if self.committer is not None:
            jac = self.committer.commit_jacobian([p.contiguous() for p in polys])
        else:
            jac = kzg.msm_bigint_batch(self.ck, [p.contiguous() for p in polys], scalars_are_mont=True)
        return [kzg.Commitment(self.curve, xy) for xy in kzg.jacobian_to_affine(self.curve, jac)]

    # ---- the rounds of one instance, as separate stages so that batch_prove (batch.py) can interleave several instances the way
    # ---- batch_prove_internal does (snark.rs:263-431); `st` carries what Oracles (structs.rs:875-887) carries, on the device
    def _stage_round1(self, wire_values, pub_input_values, blind: Blinders, tick, pi_zero: bool = False):
        """prover.rs:72-87: wire and public-input iNTTs, masking, W commitments.  pi_zero: the caller knows that pub_input_values is all
        zero (no public input): round 3 then skips the public-input polynomial."""
        import time
        import types
        import torch
        n, W, ultra = self.n, self.W, self.ultra
        st = types.SimpleNamespace(blind=blind, pi_zero=bool(pi_zero))
        dev = self.fixed.device
        on_dev = lambda x: torch.is_tensor(x) and x.is_cuda
        as_host = lambda x: x if torch.is_tensor(x) else torch.from_numpy(np.ascontiguousarray(x).view(np.int64))
        pv = pub_input_values if on_dev(pub_input_values) else as_host(pub_input_values).to(dev)
        # one slab for round 3: rows 0..W-1 wires, W z, W+1 public input (, h_1, h_2, Plookup product); coefficients in the first n+3 columns
        st.Z, st.PI, st.H1, st.PL = W, W + 1, W + 2, W + 4
        t0 = time.perf_counter()
        slab = self._slab                                               # only the first n + 3 columns are read (in_len of the coset NTT)
        slab[:, n:n + 3] = 0
        coeff = self._coeff
        if hasattr(wire_values, "wire_variables"):
            # snark.HostWitness: the witness VECTOR crosses PCIe (n_vars x 32 B), the per-wire gather of compute_wire_polynomials
            # (constraint_system.rs:1225-1247) runs on the device over the resident variable-index table
            hw = wire_values
            n_vars = int(hw.witness.shape[0])
            # the reference panics on a variable index outside the witness (`self.witness[var]`, constraint_system.rs:1239); the device gather
            # would read zero instead: checked once per index table (one reduction + one 8-byte read)
            if getattr(self, "_vars_checked", None) != (hw.wire_variables.data_ptr(), n_vars):
                top = int(hw.wire_variables.max().item()) if hw.wire_variables.numel() else -1
                if top >= n_vars or int(hw.wire_variables.min().item()) < 0:
                    raise PlonkError("wire_variables: variable index %d outside the witness vector of %d variables" % (top, n_vars))
                self._vars_checked = (hw.wire_variables.data_ptr(), n_vars)
            if getattr(self, "_wit", None) is None or self._wit.shape[0] < n_vars:
                self._wit = torch.empty((n_vars, 4), dtype=torch.int64, device=dev)
                self._wv = torch.empty((W, n, 4), dtype=torch.int64, device=dev)
            self._wit[:n_vars].copy_(hw.witness, non_blocking=True)
            st.wv = self._wv
            poly.gather_witness(self._wit[:n_vars], hw.wire_variables, out=st.wv)
            coeff[:W] = st.wv
            coeff[W] = pv
            self.domain.ifft_in_place(coeff[:W] if st.pi_zero else coeff)          # (iNTT of the zero vector is the zero vector)
        elif on_dev(wire_values):
            st.wv = wire_values
            coeff[:W] = st.wv
            coeff[W] = pv
            self.domain.ifft_in_place(coeff[:W] if st.pi_zero else coeff)          # (iNTT of the zero vector is the zero vector)
        else:
            # HOST-resident witness: the reference gathers witness[wire_variable(i, j)] on the host and starts from there
            # (constraint_system.rs:1225-1247).  Wire k + 1 crosses PCIe on a copy stream while wire k is transformed; from
            # page-locked memory (torch pin_memory / mzk_host_alloc) the copies are asynchronous DMA.
            hv = as_host(wire_values)
            if getattr(self, "_wv", None) is None:
                self._wv = torch.empty((W, n, 4), dtype=torch.int64, device=dev)
            if getattr(self, "_copy_stream", None) is None:             # (the witness-vector path allocates _wv too, without the stream)
                self._copy_stream = torch.cuda.Stream(device=dev)
                self._wv_ev = [torch.cuda.Event() for _ in range(W)]
            st.wv = self._wv
            main = torch.cuda.current_stream(dev)
            self._copy_stream.wait_stream(main)                          # the previous proof has finished with the buffer
            with torch.cuda.stream(self._copy_stream):
                for i in range(W):
                    st.wv[i].copy_(hv[i], non_blocking=True)
                    self._wv_ev[i].record(self._copy_stream)
            coeff[W] = pv
            if not st.pi_zero:
                self.domain.ifft_in_place(coeff[W:W + 1])
            for i in range(W):
                main.wait_event(self._wv_ev[i])
                coeff[i] = st.wv[i]
                self.domain.ifft_in_place(coeff[i:i + 1])
        slab[:W, :n] = coeff[:W]
        slab[st.PI, :n] = coeff[W]
        self._mask(slab, list(range(W)), blind.wires)
        tick("r1_ntt_mask", t0)
        t0 = time.perf_counter()
        if self.lagrange_ck is not None and self.committer is None:
            # commit from the VALUES over the Lagrange-basis key: sum_i v_i [L_i(beta)]g + b_0 [Z_H(beta)]g + b_1 [beta Z_H(beta)]g is the
            # same group element as the commitment of the masked coefficients (include/mzk.h, mzk_srs_generate_lagrange_for_testing) --
            # with scalars that are mostly small numbers
            if getattr(self, "_vals_ext", None) is None:
                self._vals_ext = torch.zeros((W, n + 3, 4), dtype=torch.int64, device=dev)
            ext = self._vals_ext
            ext[:, :n] = st.wv
            bl = fr_to_mont(self.curve, [b for row in blind.wires for b in row]).view(np.int64).reshape(W, 2, 4)
            ext[:, n:n + 2] = torch.from_numpy(bl).to(dev)
            jac = kzg.msm_bigint_batch(self.lagrange_ck, [ext[i, :n + 2] for i in range(W)], scalars_are_mont=True)
            wires_comms = [kzg.Commitment(self.curve, xy) for xy in kzg.jacobian_to_affine(self.curve, jac)]
        else:
            wires_comms = self._commit([slab[i, :n + 2] for i in range(W)])
        tick("r1_commit", t0)
        return st, wires_comms

    def _stage_round1_5(self, st, tau, tick):
        """prover.rs:89-118; constraint_system.rs:1290-1309, 1370-1417 (UltraPlonk only; None otherwise)."""
        import time
        import torch
        st.tau = tau
        if not self.ultra:
            return None
        n, slab = self.n, self._slab
        t0 = time.perf_counter()
        st.table, st.lookup, st.sorted_vec = plonk.compute_lookup_sorted_vec(self.pk, tau, st.wv)
        hh = torch.empty((2, n, 4), dtype=torch.int64, device=self.fixed.device)
        hh[0] = st.sorted_vec[:n]
        hh[1] = st.sorted_vec[n - 1:]
        lagrange = self.lagrange_ck is not None and self.committer is None
        if lagrange:                                                     # h_1, h_2 from the sorted vector's VALUES + three blinders each, as the wires in round 1
            ext = self._vals_ext
            ext[:2, :n] = hh
            bl = fr_to_mont(self.curve, [b for row in st.blind.h for b in row]).view(np.int64).reshape(2, 3, 4)
            ext[:2, n:n + 3] = torch.from_numpy(bl).to(hh.device)
        self.domain.ifft_in_place(hh)
        slab[st.H1:st.H1 + 2, :n] = hh
        self._mask(slab, [st.H1, st.H1 + 1], st.blind.h)
        tick("r1_5_sorted_vec", t0)
        t0 = time.perf_counter()
        if lagrange:
            jac = kzg.msm_bigint_batch(self.lagrange_ck, [ext[0, :n + 3], ext[1, :n + 3]], scalars_are_mont=True)
            h_comms = [kzg.Commitment(self.curve, xy) for xy in kzg.jacobian_to_affine(self.curve, jac)]
        else:
            h_comms = self._commit([slab[st.H1, :n + 3], slab[st.H1 + 1, :n + 3]])
        tick("r1_5_commit", t0)
        return h_comms

    def _stage_round2(self, st, beta, gamma, tick):
        """prover.rs:125-141; constraint_system.rs:1197-1223"""
        import time
        n, slab, coeff = self.n, self._slab, self._coeff
        st.beta, st.gamma = beta, gamma
        t0 = time.perf_counter()
        plonk.compute_prod_permutation_polynomial_dev(self.pk, beta, gamma, st.wv.contiguous(), out_dev=coeff[0])
        slab[st.Z, :n] = coeff[0]
        self._mask(slab, [st.Z], [st.blind.z])
        tick("r2_product", t0)
        t0 = time.perf_counter()
        z_comm = self._commit([slab[st.Z, :n + 3]])[0]
        tick("r2_commit", t0)
        return z_comm

    def _stage_round2_5(self, st, tick):
        """prover.rs:143-183; constraint_system.rs:1311-1368 (UltraPlonk only)"""
        import time
        if not self.ultra:
            return None
        n, slab, coeff = self.n, self._slab, self._coeff
        t0 = time.perf_counter()
        plonk.compute_lookup_prod_polynomial(self.pk, st.beta, st.gamma, st.table, st.lookup, st.sorted_vec, out_dev=coeff[0])
        slab[st.PL, :n] = coeff[0]
        self._mask(slab, [st.PL], [st.blind.prod_lookup])
        tick("r2_5_product", t0)
        t0 = time.perf_counter()
        pl_comm = self._commit([slab[st.PL, :n + 3]])[0]
        tick("r2_5_commit", t0)
        return pl_comm

    def _stage_quotient(self, st, alpha, tick):
        """prover.rs:512-673 for this instance: the quotient's 8n coefficients into self._quot (the sum over instances and the
        split are the caller's: prover.rs:661-669, 902-960)."""
        import time
        c, n = self.curve, self.n
        st.alpha = alpha
        t0 = time.perf_counter()
        slab, keep, quot = self._slab, self._keep, self._quot
        if keep is not slab:
            keep.copy_(slab[:, :n + 3])                                  # whole-domain path: coefficient forms survive the in-place coset NTT
        ch = plonk.Challenges(alpha, st.beta, st.gamma, st.tau)
        if self.pk.classes is None:
            plonk.compute_quotient_polynomial_dev(self.pk, ch, slab, n + 3, quot)
        else:                                                            # local classes, one exchange, 8-point iDFT per coefficient
            local = plonk.compute_quotient_chunked_dev(self.pk, ch, slab, n + 3, pi_zero=st.pi_zero) if self.own_classes else None
            if local is None:                                            # this rank owns no class: it only takes part in the exchange
                import torch
                local = torch.empty((0, n, 4), dtype=torch.int64, device=slab.device)
            every = self.quotient_gather(local, len(self.classes_needed)) if self.quotient_gather is not None else local
            resident = self.classes_needed if self.quotient_gather is not None else self.own_classes
            if len(resident) == self.W and plonk.quotient_top_supported(self.W, n):
                # W classes + the W + 3 top coefficients of the numerator (every rank computes its own copy of those)
                top = plonk.compute_quotient_top_dev(self.pk, ch, slab, n + 3)
                plonk.combine_quotient_classes(c, n, every.contiguous(), classes=resident, out_dev=quot, top=top, n_top=self.W + 3)
            else:
                plonk.combine_quotient_classes(c, n, every.contiguous(), classes=resident, out_dev=quot)
        # quot_poly.degree() != expected_degree => WrongQuotientPolyDegree (prover.rs:915-918): the reference's only guard against an
        # unsatisfied witness (batch_prove_internal never runs check_circuit_satisfiability).  The length is computed on the
        # device now and read in check_quotient_degree, after the round's commitments have synchronised the stream anyway.
        expected = self.W * (n + 1) + 2
        st.quot_len = poly.degree_len_async(quot[expected:])             # only what lies at and above the expected degree is scanned
        tick("r3_quotient", t0)
        st.wire_polys = [keep[i, :n + 2] for i in range(self.W)]
        st.z_poly = keep[st.Z]
        if self.ultra:
            st.h1, st.h2, st.pl_poly = keep[st.H1], keep[st.H1 + 1], keep[st.PL]
        return quot

    def check_quotient_degree(self, quot_len, num_instances: int = 1):
        """prover.rs:915-918 on the length produced by poly.degree_len_async (one 8-byte read; call it after a synchronising step)."""
        expected = self.W * (self.n + 1) + 2
        tail_len = int(quot_len.item())                                  # of quot[expected:]: 1 <=> degree exactly `expected`
        got = expected + tail_len - 1 if tail_len else expected - 1      # (below `expected`: reported as expected - 1)
        if tail_len != 1:
            raise PlonkError("quotient polynomial of degree %d, expected %d (the witness does not satisfy the circuit)" % (got, expected),
                             kind="WrongQuotientPolyDegree")

    def _split_quotient(self, quot, blind_quot):
        """split_quotient_polynomial (prover.rs:902-960): W slices of n + 2 coefficients, masked by W - 1 scalars.  The scalars
        travel as kernel arguments of mzk_poly_lincomb_dev (times a resident one): no host-to-device copy, hence no stream
        synchronisation between the quotient kernels and the commitments."""
        import torch
        c, n, r, W = self.curve, self.n, self.curve.r, self.W
        dev = self.fixed.device
        if getattr(self, "_one", None) is None:
            self._one = torch.from_numpy(fr_to_mont(c, [1]).view(np.int64)).to(dev)
        expected = W * (n + 1) + 2                                       # quotient_polynomial_degree, prover.rs:1125-1128
        split = []
        last = 0
        for i in range(W):
            lo = i * (n + 2)
            hi = (i + 1) * (n + 2) if i < W - 1 else expected + 1
            p = torch.zeros((n + 3, 4), dtype=torch.int64, device=dev)
            p[:hi - lo] = quot[lo:hi]
            if i < W - 1:
                poly.lincomb(c, [(blind_quot[i] % r, self._one)], out=p[n + 2:n + 3])
            if last:
                poly.lincomb(c, [(1, p[:1].clone()), ((-last) % r, self._one)], out=p[:1])
            last = blind_quot[i] if i < W - 1 else 0
            split.append(p if i < W - 1 else p[:hi - lo])
        return split

    def _stage_round4(self, st, zeta, tick):
        """compute_evaluations / compute_plookup_evaluations (prover.rs:216-299).  Over several ranks every evaluation is the sum of
        the ranks' coefficient-range contributions (one exchange of 32-byte partial values per call of this stage)."""
        import time
        c, n, r, W = self.curve, self.n, self.curve.r, self.W
        keep = self._keep
        st.zeta = zeta
        t0 = time.perf_counter()
        zeta_w = zeta * self.w_n % r
        ev = _RangeEvals(self) if self._ranged() else _Evals(self)
        # wires, and in the same launch z and the public-input polynomial (rows W, W + 1; every row is zero above its own length): pi(zeta)
        # is not part of the proof, check_quotient_identity needs it
        h_w = ev.add(keep[:W + 2], zeta, length=n + 3)
        h_s = ev.add(self.fixed[self.sigma0:self.sigma0 + W - 1], zeta)
        h_z = ev.add(st.z_poly, zeta_w)
        if self.ultra:
            tabs = self.fixed[self.tab0:self.tab0 + 4]                    # range, key, table_dom_sep, q_dom_sep
            q_lookup = self.fixed[13]
            h_tz, h_tn = ev.add(tabs, zeta), ev.add(tabs[:3], zeta_w)
            h_h1, h_ql, h_qln = ev.add(st.h1, zeta), ev.add(q_lookup, zeta), ev.add(q_lookup, zeta_w)
            h_nx = ev.add(keep[[st.PL, st.H1, st.H1 + 1, 3, 4]], zeta_w)
        ev.finish()
        st.wires_evals, st.wire_sigma_evals, st.perm_next_eval = ev.get(h_w)[:W], ev.get(h_s), ev.get(h_z)[0]
        st.pi_eval = ev.get(h_w)[W + 1]
        st.pe = None
        if self.ultra:
            at_zeta, at_next, nx = ev.get(h_tz), ev.get(h_tn), ev.get(h_nx)
            st.pe = {"range_table_eval": at_zeta[0], "key_table_eval": at_zeta[1], "table_dom_sep_eval": at_zeta[2], "q_dom_sep_eval": at_zeta[3],
                     "range_table_next_eval": at_next[0], "key_table_next_eval": at_next[1], "table_dom_sep_next_eval": at_next[2],
                     "h_1_eval": ev.get(h_h1)[0], "q_lookup_eval": ev.get(h_ql)[0], "q_lookup_next_eval": ev.get(h_qln)[0],
                     "prod_next_eval": nx[0], "h_1_next_eval": nx[1], "h_2_next_eval": nx[2], "w_3_next_eval": nx[3], "w_4_next_eval": nx[4]}
        tick("r4_evals", t0)
        return st.wires_evals, st.wire_sigma_evals, st.perm_next_eval, st.pe

    def _lin_poly_terms(self, st, alpha_base: int = 1):
        """compute_non_quotient_component_for_lin_poly (prover.rs:302-337, 963-1112) as (scalar, polynomial) terms, every scalar
        times alpha_base (the combiner over instances, snark.rs:408-428)."""
        r, n, W = self.curve.r, self.n, self.W
        alpha, beta, gamma, tau, zeta = st.alpha, st.beta, st.gamma, st.tau, st.zeta
        we, wire_sigma_evals, perm_next_eval, pe = st.wires_evals, st.wire_sigma_evals, st.perm_next_eval, st.pe
        sel = self.fixed
        terms = [(we[j], sel[j]) for j in range(4)]
        terms += [(we[0] * we[1] % r, sel[4]), (we[2] * we[3] % r, sel[5])]
        terms += [(pow(we[j], 5, r), sel[6 + j]) for j in range(4)]
        terms += [(we[0] * we[1] % r * we[2] % r * we[3] % r * we[4] % r, sel[12]), ((-we[4]) % r, sel[10]), (1, sel[11])]
        vanish = (pow(zeta, n, r) - 1) % r
        lagrange_1 = vanish * pow(n * (zeta - 1) % r, -1, r) % r
        cf = alpha
        for j in range(W):
            cf = cf * (we[j] + beta * self.k[j] % r * zeta + gamma) % r
        terms.append(((cf + alpha * alpha % r * lagrange_1) % r, st.z_poly))
        cf = alpha * beta % r * perm_next_eval % r
        for j in range(W - 1):
            cf = cf * (we[j] + beta * wire_sigma_evals[j] + gamma) % r
        terms.append(((-cf) % r, self.fixed[self.sigma0 + W - 1]))
        if self.ultra:                                                   # compute_lin_poly_plookup_contribution, prover.rs:1037-1112
            em = lambda first, ql, ds, a0, a1, a2: (first + ql * tau % r * (ds + tau * (a0 + tau * (a1 + tau * a2))) % r) % r
            mt = em(pe["range_table_eval"], pe["q_lookup_eval"], pe["table_dom_sep_eval"], pe["key_table_eval"], we[3], we[4])
            mt_next = em(pe["range_table_next_eval"], pe["q_lookup_next_eval"], pe["table_dom_sep_next_eval"], pe["key_table_next_eval"],
                         pe["w_3_next_eval"], pe["w_4_next_eval"])
            ml = em(we[5], pe["q_lookup_eval"], pe["q_dom_sep_eval"], we[0], we[1], we[2])
            w_inv = pow(self.w_n, -1, r)
            lagrange_n = vanish * w_inv % r * pow(n * (zeta - w_inv) % r, -1, r) % r
            a4, a5, a6 = (pow(alpha, e, r) for e in (4, 5, 6))
            b1 = (1 + beta) % r
            g1 = gamma * b1 % r
            zmg = (zeta - w_inv) % r
            cf = (a4 * lagrange_1 + a5 * lagrange_n + a6 * zmg % r * b1 % r * ((gamma + ml) % r) % r * ((g1 + mt + beta * mt_next) % r)) % r
            terms.append((cf, st.pl_poly))
            cf = a6 * zmg % r * pe["prod_next_eval"] % r * ((g1 + pe["h_1_eval"] + beta * pe["h_1_next_eval"]) % r) % r
            terms.append(((-cf) % r, st.h2))
        if alpha_base != 1:
            terms = [(s * alpha_base % r, p) for s, p in terms]
        return terms

    def _lin_poly_constant(self, st, alpha_base: int = 1) -> int:
        """What the verifier takes for -(linearisation polynomial)(zeta): Verifier::compute_lin_poly_constant_term (verifier.rs:340-414) for
        this instance, times alpha_base.  The prover knows every input: its own evaluations and pi(zeta)."""
        r, n, W = self.curve.r, self.n, self.W
        alpha, beta, gamma, zeta = st.alpha, st.beta, st.gamma, st.zeta
        we, se, zn, pe = st.wires_evals, st.wire_sigma_evals, st.perm_next_eval, st.pe
        a2 = alpha * alpha % r
        vanish = (pow(zeta, n, r) - 1) % r
        lagrange_1 = vanish * pow(n * (zeta - 1) % r, -1, r) % r
        tmp = (st.pi_eval - a2 * lagrange_1) % r
        acc = alpha * zn % r * ((gamma + we[W - 1]) % r) % r
        for j in range(W - 1):
            acc = acc * ((gamma + we[j] + beta * se[j]) % r) % r
        tmp = (tmp - acc) % r
        if self.ultra:
            a3 = a2 * alpha % r
            w_inv = pow(self.w_n, -1, r)
            lagrange_n = vanish * w_inv % r * pow(n * (zeta - w_inv) % r, -1, r) % r
            g1 = gamma * ((1 + beta) % r) % r
            pc = (lagrange_n * ((pe["h_1_eval"] - pe["h_2_next_eval"] - a2) % r) - alpha * lagrange_1
                  - a3 * ((zeta - w_inv) % r) % r * pe["prod_next_eval"] % r * ((g1 + pe["h_1_eval"] + beta * pe["h_1_next_eval"]) % r) % r
                  * ((g1 + beta * pe["h_2_next_eval"]) % r)) % r
            tmp = (tmp + a3 * pc) % r
        return tmp * alpha_base % r

    def _opened_evals(self, st):
        """the evaluations at zeta in the order of _open_lists' first list (after the linearisation polynomial)"""
        out = list(st.wires_evals) + list(st.wire_sigma_evals)
        if self.ultra:
            pe = st.pe
            out += [pe["range_table_eval"], pe["key_table_eval"], pe["h_1_eval"], pe["q_lookup_eval"], pe["table_dom_sep_eval"], pe["q_dom_sep_eval"]]
        return out

    def check_quotient_identity(self, batch_at_zeta: int, lin_constant: int, opened_evals, v_ch: int):
        """t(X) Z_H(X) = numerator(X), checked at the evaluation challenge the way the verifier will check it (verifier.rs:186-231, 340-414):
        the opening proof's batch polynomial lin + sum_i v^i p_i must take the value -r_0 + sum_i v^i p_i(zeta) at zeta -- and its value
        there is the remainder its division by (X - zeta) leaves (mzk_poly_div_linear_rem_dev), so the check costs one 32-byte read.
        This is the guard against an unsatisfied witness on the path that takes the top coefficients of the quotient from its
        numerator (plonk.compute_quotient_top_dev): the reference's `WrongQuotientPolyDegree` (prover.rs:915-918) cannot fire there,
        the recovered polynomial having the expected degree by construction."""
        r = self.curve.r
        want, cf = (-lin_constant) % r, 1
        for e in opened_evals:
            cf = cf * v_ch % r
            want = (want + cf * e) % r
        if batch_at_zeta % r != want:
            raise PlonkError("the quotient identity t(X) Z_H(X) = numerator(X) does not hold at the evaluation challenge "
                             "(the witness does not satisfy the circuit)", kind="WrongQuotientPolyDegree")

    def _quotient_lin_terms(self, zeta, split):
        """compute_quotient_component_for_lin_poly (prover.rs:343-358)"""
        r, n = self.curve.r, self.n
        vanish = (pow(zeta, n, r) - 1) % r
        zeta_n2 = (vanish + 1) * zeta % r * zeta % r
        terms, cf = [], 1
        for p in split:
            terms.append(((-vanish) * cf % r, p))
            cf = cf * zeta_n2 % r
        return terms

    def _open_lists(self, st):
        """the polynomials opened at zeta (after the linearisation polynomial) and at zeta * w for this instance
        (compute_opening_proofs, prover.rs:362-419; plookup lists :421-460)"""
        W = self.W
        sig = [self.fixed[self.sigma0 + j] for j in range(W)]
        open_polys = list(st.wire_polys) + sig[:W - 1]
        shifted_polys = [st.z_poly]
        if self.ultra:
            tabs = self.fixed[self.tab0:self.tab0 + 4]
            q_lookup = self.fixed[13]
            open_polys += [tabs[0], tabs[1], st.h1, q_lookup, tabs[2], tabs[3]]
            shifted_polys += [st.pl_poly, tabs[0], tabs[1], st.h1, st.h2, q_lookup, st.wire_polys[3], st.wire_polys[4], tabs[2]]
        return open_polys, shifted_polys

    def _batched_witness(self, polys, v_ch, point, rem_out=None):
        """compute_batched_witness_polynomial_commitment (prover.rs:490-509) up to the commitment: sum_i v^i p_i, divided by (X - point);
        rem_out (a (1, 4) device tensor) receives the remainder = the batch polynomial's value at the point"""
        c, r, n = self.curve, self.curve.r, self.n
        if len(polys) == 1:
            return poly.div_by_linear(c, polys[0].contiguous(), point, rem_out=rem_out)
        bterms, cf = [], 1
        for p in polys:
            bterms.append((cf, p))
            cf = cf * v_ch % r
        if len(bterms) <= poly.MAX_TERMS:
            return poly.div_by_linear(c, poly.lincomb(c, bterms, out_len=n + 3), point, rem_out=rem_out)
        acc = poly.lincomb(c, bterms[:poly.MAX_TERMS], out_len=n + 3)      # more terms than one launch takes: accumulate
        for i in range(poly.MAX_TERMS, len(bterms), poly.MAX_TERMS - 1):
            acc = poly.lincomb(c, [(1, acc)] + bterms[i:i + poly.MAX_TERMS - 1], out_len=n + 3)
        return poly.div_by_linear(c, acc, point, rem_out=rem_out)

    def _lincomb_many(self, terms, out_len):
        """sum of (scalar, polynomial) terms, more than one launch's worth if need be"""
        c = self.curve
        if len(terms) <= poly.MAX_TERMS:
            return poly.lincomb(c, terms, out_len=out_len)
        acc = poly.lincomb(c, terms[:poly.MAX_TERMS], out_len=out_len)
        for i in range(poly.MAX_TERMS, len(terms), poly.MAX_TERMS - 1):
            acc = poly.lincomb(c, [(1, acc)] + terms[i:i + poly.MAX_TERMS - 1], out_len=out_len)
        return acc

    def _openings_ranged(self, lin_terms, open_polys, shifted_polys, v_ch, zeta, tick, t0):
        """Round 5 over several ranks.  The opening witness of a batch polynomial b at a point z is
        w_j = sum_{i > j} b_i z^(i-j-1).  A rank needs w on its own coefficient range [lo, hi) only -- that is its MSM shard -- and
        w_j = (the same sum over i < hi) + z^(hi-1-j) S_hi with S_hi = sum_{i >= hi} b_i z^(i-hi): the higher ranks' contribution enters
        as ONE field element.  So: linear combinations on the range only (they are pointwise); e = the range read as a polynomial,
        evaluated at z; one all-gather of the e's; S_hi appended as an extra top coefficient, after which the ordinary division by
        (X - z) of the extended range returns exactly w on the range; commit over the range.  Returns (lin range, opening range,
        shifted range, commitments)."""
        import time
        import torch
        c, r, n = self.curve, self.curve.r, self.n
        com = self.committer
        lo, hi = com.point_range()
        hi = min(hi, n + 3)
        width = max(hi - lo, 0)
        dev = self.fixed.device

        def cut(terms):
            out = []
            for s_, p_ in terms:
                a, b = min(lo, int(p_.shape[0])), min(hi, int(p_.shape[0]))
                if b > a:
                    out.append((s_, p_[a:b]))
            return out

        zw = zeta * self.w_n % r
        vs = [pow(v_ch, i, r) for i in range(max(len(open_polys) + 1, len(shifted_polys)))]
        open_terms = cut(lin_terms + [(vs[i + 1], p_) for i, p_ in enumerate(open_polys)])       # 1 * lin + sum_i v^(i+1) p_i
        shift_terms = cut([(vs[i], p_) for i, p_ in enumerate(shifted_polys)])
        zero = torch.zeros((max(width, 1), 4), dtype=torch.int64, device=dev)
        lin = self._lincomb_many(cut(lin_terms), width) if width and cut(lin_terms) else zero[:width]
        b_open = self._lincomb_many(open_terms, width) if width and open_terms else zero[:width]
        b_shift = self._lincomb_many(shift_terms, width) if width and shift_terms else zero[:width]
        e_open = poly.evaluate(c, b_open, zeta)[0] if width else 0
        e_shift = poly.evaluate(c, b_shift, zw)[0] if width else 0
        every = com.all_gather_fr([e_open, e_shift])
        shard_range = _sharding.shard_range
        rank, world = com.rank(), com.world()
        carry = [0, 0]
        for q in range(rank + 1, world):                                 # S_hi: the ranges above, shifted down to start at hi
            lo_q = min(shard_range(com.ck.length, q, world)[0], n + 3)
            carry[0] = (carry[0] + pow(zeta, lo_q - hi, r) * every[q][0]) % r
            carry[1] = (carry[1] + pow(zw, lo_q - hi, r) * every[q][1]) % r
        # the batch polynomial's value at zeta (check_quotient_identity): every rank's range value times zeta^lo
        self._batch_at_zeta = sum(pow(zeta, min(shard_range(com.ck.length, q, world)[0], n + 3), r) * every[q][0] for q in range(world)) % r
        wit = []
        for b_, cy, z_ in ((b_open, carry[0], zeta), (b_shift, carry[1], zw)):
            if not width:
                wit.append(zero[:0])
                continue
            ext = torch.empty((width + 1, 4), dtype=torch.int64, device=dev)
            ext[:width] = b_
            ext[width:] = torch.from_numpy(fr_to_mont(c, [cy]).view(np.int64)).to(dev)
            wit.append(poly.div_by_linear(c, ext, z_))                   # width coefficients: w on [lo, hi)
        tick("r5_polys", t0)
        t0 = time.perf_counter()
        jac = com.commit_jacobian_slices(wit)
        open_comms = [kzg.Commitment(c, xy) for xy in kzg.jacobian_to_affine(c, jac)]
        tick("r5_commit", t0)
        return lin, wit[0], wit[1], open_comms

    def prove(self, wire_values, pub_input_values, ch, blind: Blinders, profile: bool = False, pi_zero: bool = False) -> ProofCore:
        """ch: ProverChallenges (fixed) or a challenge source (FixedChallenges / TranscriptChallenges).  pi_zero: pub_input_values is all
        zero (a circuit without public inputs)."""
        src = FixedChallenges(ch) if isinstance(ch, ProverChallenges) else ch
        import time
        import torch
        c, n, r, W, ultra = self.curve, self.n, self.curve.r, self.W, self.ultra
        tm = {}

        def tick(name, t0):
            if profile:
                torch.cuda.synchronize()
                tm[name] = round((time.perf_counter() - t0) * 1e3, 3)

        st, wires_comms = self._stage_round1(wire_values, pub_input_values, blind, tick, pi_zero=pi_zero)
        tau = src.after_round1(wires_comms)
        h_comms = self._stage_round1_5(st, tau, tick)
        beta, gamma = src.after_round1_5(h_comms)
        z_comm = self._stage_round2(st, beta, gamma, tick)
        pl_comm = self._stage_round2_5(st, tick)
        # ---- round 3 (prover.rs:192-209, 512-673, 902-960)
        alpha = src.after_round2(z_comm, pl_comm)
        quot = self._stage_quotient(st, alpha, tick)
        t0 = time.perf_counter()
        split = self._split_quotient(quot, blind.quot)
        tick("r3_split", t0)
        t0 = time.perf_counter()
        split_comms = self._commit(split)
        tick("r3_commit", t0)
        self.check_quotient_degree(st.quot_len)
        # ---- round 4 (prover.rs:216-299)
        t0 = time.perf_counter()
        zeta = src.after_round3(split_comms)
        tick("r4_transcript", t0)
        wires_evals, wire_sigma_evals, perm_next_eval, pe = self._stage_round4(st, zeta, tick)
        # ---- round 5: linearisation polynomial (prover.rs:963-1112, 343-358) and openings (362-460, 490-509)
        t0 = time.perf_counter()
        v_ch = src.after_round4(wires_evals, wire_sigma_evals, perm_next_eval, pe)
        open_polys, shifted_polys = self._open_lists(st)
        if self._ranged():                                               # this rank's coefficient range of both witness polynomials only
            lin, opening, shifted, open_comms = self._openings_ranged(self._lin_poly_terms(st) + self._quotient_lin_terms(zeta, split),
                                                                      open_polys, shifted_polys, v_ch, zeta, tick, t0)
        else:
            lin = poly.lincomb(c, self._lin_poly_terms(st) + self._quotient_lin_terms(zeta, split), out_len=n + 3)
            rem = torch.zeros((1, 4), dtype=torch.int64, device=lin.device)
            opening = self._batched_witness([lin] + open_polys, v_ch, zeta, rem_out=rem)
            shifted = self._batched_witness(shifted_polys, v_ch, zeta * self.w_n % r)
            tick("r5_polys", t0)
            t0 = time.perf_counter()
            open_comms = self._commit([opening, shifted])
            tick("r5_commit", t0)
            self._batch_at_zeta = fr_from_mont(c, rem.cpu().numpy().view(np.uint64))[0]      # (the commitments have synchronised the stream)
        if not self.identity_check and self.pk.classes is not None and len(self.classes_needed) == self.W:
            # the W-class path recovers a polynomial of the expected degree whatever the witness: the identity at zeta is its ONLY guard
            if not getattr(self, "_allow_unchecked", False):
                raise PlonkError("identity_check = False on the W-class quotient path leaves an unsatisfied witness undetected; measurement "
                                 "harnesses set _allow_unchecked as well")
        if self.identity_check:
            self.check_quotient_identity(self._batch_at_zeta, self._lin_poly_constant(st), self._opened_evals(st), v_ch)
        self.last_challenges = {"tau": tau, "beta": beta, "gamma": gamma, "alpha": alpha, "zeta": zeta, "v": v_ch}
        self.last = {"wire_polys": st.wire_polys, "z_poly": st.z_poly, "quot": quot, "split": split, "lin": lin, "opening": opening, "shifted": shifted}
        if ultra:
            self.last.update({"h_polys": [st.h1, st.h2], "prod_lookup_poly": st.pl_poly, "sorted_vec": st.sorted_vec, "merged_table": st.table,
                              "merged_lookup": st.lookup})
        return ProofCore(wires_comms, z_comm, split_comms, open_comms[0], open_comms[1], wires_evals, wire_sigma_evals, perm_next_eval, tm,
                         h_comms, pl_comm, pe)


# ---- PlonkKzgSnark::batch_prove on the mirror's stages (until round 4: batch.batch_prove) -----------------------------------------------
def _bpt(c, cm):
    return mj.batch._pt(c, cm)


def batch_prove(provers: list, wire_values: list, pub_input_values: list, pub_inputs: list, blinds: list[Blinders],
                quot_blinders: list[int], extra_transcript_init_msg: bytes | None = None) -> BatchProofCore:
    """provers[k]: the proving key of instance k on the device; wire_values[k] (W, n, 4) / pub_input_values[k] (n, 4): its witness
    and public-input evaluations on H (Montgomery); pub_inputs[k]: its public input as ints (for the transcript);
    blinds[k]: its masking scalars (the `quot` field is ignored); quot_blinders: the W - 1 scalars of the one split (round 3)."""
    if not provers:
        raise ValueError("zero number of circuits/proving keys")                                  # snark.rs:213-215
    if not (len(provers) == len(wire_values) == len(pub_input_values) == len(pub_inputs) == len(blinds)):
        raise ValueError("the number of circuits != the number of proving keys")                  # snark.rs:216-223
    if len({id(p) for p in provers}) != len(provers):
        # the device workspace (slab, coefficient forms, quotient) belongs to the TurboPlonkProver: two instances of one circuit
        # need two provers (preprocess twice) -- the reference's `prove_keys` may repeat because its Oracles live on the host
        raise ValueError("one TurboPlonkProver per instance: the same prover object was passed twice")
    p0 = provers[0]
    c, n, r, W = p0.curve, p0.n, p0.curve.r, p0.W
    for p in provers:
        if p.n != n:
            raise ValueError("proving key domain size %d != expected domain size %d" % (p.n, n))  # snark.rs:236-243
        if p.W != W:
            raise ValueError("inconsistent plonk circuit types")                                  # snark.rs:258-260
        if p.curve.curve_id != c.curve_id:
            raise ValueError("instances over different curves")
    K = len(provers)
    tick = lambda name, t0: None
    # transcript init (snark.rs:263-270)
    t = _transcript.StandardTranscript(c, b"PlonkProof")
    if extra_transcript_init_msg is not None:
        t.append_message(b"extra info", extra_transcript_init_msg)
    for p, pub in zip(provers, pub_inputs):
        sel, sig = p.vk_commitments()
        t.append_vk_and_pub_input(p.n, len(pub), p.k, [_bpt(c, x) for x in sel], [_bpt(c, x) for x in sig], pub)
    # round 1
    states, wires_vec = [], []
    for k, p in enumerate(provers):
        st, wires_comms = p._stage_round1(wire_values[k], pub_input_values[k], blinds[k], tick, pi_zero=not any(pub_inputs[k]))
        t.append_commitments(b"witness_poly_comms", [_bpt(c, x) for x in wires_comms])
        states.append(st)
        wires_vec.append(wires_comms)
    tau = t.get_and_append_challenge(b"tau")
    # round 1.5
    h_vec = []
    for p, st in zip(provers, states):
        h_comms = p._stage_round1_5(st, tau, tick)
        if h_comms is not None:
            t.append_commitments(b"h_poly_comms", [_bpt(c, x) for x in h_comms])
        h_vec.append(h_comms)
    beta = t.get_and_append_challenge(b"beta")
    gamma = t.get_and_append_challenge(b"gamma")
    # round 2
    z_vec = []
    for p, st in zip(provers, states):
        z_comm = p._stage_round2(st, beta, gamma, tick)
        t.append_commitment(b"perm_poly_comms", _bpt(c, z_comm))
        z_vec.append(z_comm)
    # round 2.5
    pl_vec = []
    for p, st in zip(provers, states):
        pl_comm = p._stage_round2_5(st, tick)
        if pl_comm is not None:
            t.append_commitment(b"plookup_poly_comms", _bpt(c, pl_comm))
        pl_vec.append(pl_comm)
    # round 3: per-instance quotients, one weighted sum, one split (prover.rs:661-673, 902-960)
    alpha = t.get_and_append_challenge(b"alpha")
    a3 = pow(alpha, 3, r)
    a7 = pow(alpha, 7, r)
    bases, base, terms = [], 1, []
    for p, st in zip(provers, states):
        terms.append((base, p._stage_quotient(st, alpha, tick)))
        bases.append(base)
        base = base * (a7 if p.ultra else a3) % r
    quot = terms[0][1] if K == 1 else poly.lincomb(c, terms)
    quot_len = poly.degree_len_async(quot[p0.W * (n + 1) + 2:])          # of the aggregated quotient, from the expected degree up (prover.rs:915-918)
    split = p0._split_quotient(quot, quot_blinders)
    split_comms = p0._commit(split)
    p0.check_quotient_degree(quot_len)
    t.append_commitments(b"quot_poly_comms", [_bpt(c, x) for x in split_comms])
    # round 4 / 4.5: all ProofEvaluations first, then all PlookupEvaluations (snark.rs:365-399)
    zeta = t.get_and_append_challenge(b"zeta")
    evals_vec = []
    for p, st in zip(provers, states):
        we, se, zn, _ = p._stage_round4(st, zeta, tick)
        for e in we:
            t.append_field_elem(b"wire_evals", e)
        for e in se:
            t.append_field_elem(b"wire_sigma_evals", e)
        t.append_field_elem(b"perm_next_eval", zn)
        evals_vec.append((we, se, zn))
    for st in states:
        if st.pe is not None:
            t.append_plookup_evaluations(st.pe)
    # linearisation polynomial (snark.rs:403-428)
    lin_terms = p0._quotient_lin_terms(zeta, split)
    for p, st, b in zip(provers, states, bases):
        lin_terms += p._lin_poly_terms(st, b)
    lin = None
    for i in range(0, len(lin_terms), poly.MAX_TERMS - 1):
        chunk = lin_terms[i:i + poly.MAX_TERMS - 1]
        lin = poly.lincomb(c, chunk if lin is None else [(1, lin)] + chunk, out_len=n + 3)
    # round 5 (prover.rs:362-419)
    v = t.get_and_append_challenge(b"v")
    open_polys, shifted_polys = [lin], []
    for p, st in zip(provers, states):
        o, s = p._open_lists(st)
        open_polys += o
        shifted_polys += s
    import torch
    rem = torch.zeros((1, 4), dtype=torch.int64, device=lin.device)
    opening = p0._batched_witness(open_polys, v, zeta, rem_out=rem)
    shifted = p0._batched_witness(shifted_polys, v, zeta * p0.w_n % r)
    open_comms = p0._commit([opening, shifted])
    # the quotient identity at zeta over all instances (prover.check_quotient_identity): the guard against an unsatisfied witness
    lin_constant = sum(p._lin_poly_constant(st, b) for p, st, b in zip(provers, states, bases)) % r
    opened = [e for p, st in zip(provers, states) for e in p._opened_evals(st)]
    p0.check_quotient_identity(fr_from_mont(c, rem.cpu().numpy().view(np.uint64))[0], lin_constant, opened, v)
    plookup_vec = [None if st.pe is None else (h, pl, st.pe) for st, h, pl in zip(states, h_vec, pl_vec)]
    return mj.batch.BatchProofCore(wires_vec, z_vec, evals_vec, plookup_vec, split_comms, open_comms[0], open_comms[1],
                          {"tau": tau, "beta": beta, "gamma": gamma, "alpha": alpha, "zeta": zeta, "v": v})


# ---- torch.distributed sharding of the commit path on the mirror (until round 4: sharding.ShardedCommitter) -------------------------------
class ShardedCommitter:
    """`UnivariateKzgPCS::{commit, batch_commit}` (mod.rs:90-131) across the ranks of a process group.

    Every polynomial is sharded by point range: rank g runs ONE fused batch of k MSMs over its slice
    [g*len/G, (g+1)*len/G) of every polynomial (base_offset = slice start into the replicated SRS and its
    precomputed table), then the k x G Jacobian partials (144 B each) are all-gathered and summed on every rank, so
    all ranks hold identical commitments and their transcripts stay in step.  Compared with "polynomial i on rank
    i % G"  the load is balanced for any k and G (5 wire commitments on 8 GPUs), at the price of
    the same one small collective.
    `msm_batch(ck, scalars_list, base_offsets) -> (k, 3, fq_limbs)` defaults to the device path; the CPU tests inject
    the oracle there."""

    def __init__(self, curve, ck, group=None, device=None, msm_batch=None, slice_srs: bool = False):
        """slice_srs: register this rank's point range of the SRS as an SRS of its own.  The fixed-base table's window is chosen
        by SRS size (csrc/msm.hip srs_build_pre_t): a 2^17-point slice gets a small window and 2^15 buckets per MSM instead of the
        full SRS's 2^19 -- at 8 GPUs the bucket reduction, not the accumulation, is what a shard's MSM costs -- and the rank
        holds 1 / G of the table."""
        self.c = _curve(curve)
        self.ck, self.group, self.device = ck, group, device
        self.msm_batch = msm_batch
        self.slice_srs = slice_srs and ck is not None and msm_batch is None
        self._slice = None

    def _local(self, slices, offsets):
        if self.msm_batch is not None:
            return self.msm_batch(self.ck, slices, offsets)
        if self._slice is not None:
            lo = self._slice_lo
            return kzg.msm_bigint_batch(self._slice, slices, [o - lo for o in offsets], scalars_are_mont=True)
        return kzg.msm_bigint_batch(self.ck, slices, offsets, scalars_are_mont=True)

    def commit_jacobian(self, polys) -> np.ndarray:
        """polys: list of (len, 4) Montgomery coefficient arrays / CUDA tensors, identical on every rank.
        Returns (k, 3, fq_limbs) Jacobian commitments, identical on every rank."""
        import torch
        import torch.distributed as dist
        world = dist.get_world_size(self.group)
        rank = dist.get_rank(self.group)
        k, L = len(polys), self.c.fq_limbs
        # ONE partition of the point indices for all polynomials -- by the SRS length when there is an SRS (so that a rank always
        # works on the same points and can keep just those), else by the longest polynomial of the call
        total = self.ck.length if self.ck is not None else max([int(p.shape[0]) for p in polys] + [1])
        lo_r, hi_r = _sharding.shard_range(total, rank, world)
        if self.slice_srs and self._slice is None and hi_r > lo_r:
            self._slice = self.ck.slice(lo_r, hi_r - lo_r)               # mzk_srs_slice: a device copy of this rank's range
            self._slice_lo = lo_r
        slices, offsets = [], []
        for p in polys:
            lo, hi = min(lo_r, int(p.shape[0])), min(hi_r, int(p.shape[0]))
            s = p[lo:hi]
            slices.append(s.contiguous() if hasattr(s, "contiguous") else np.ascontiguousarray(s))
            offsets.append(lo if hi > lo else lo_r)
        part = np.ascontiguousarray(self._local(slices, offsets), dtype=np.uint64).reshape(k, 3, L)
        stacked = _sharding.gather_partials(part, self.group, self.device)
        return np.stack([_sharding.sum_jacobian(self.c, stacked[:, i]) for i in range(k)])

    # ---- coefficient-range mode: the pointwise stages of rounds 4 and 5 run on this rank's
    # ---- coefficient range only -- the very range its MSM shard needs -- with small exchanges of field elements
    def world(self) -> int:
        import torch.distributed as dist
        return dist.get_world_size(self.group)

    def rank(self) -> int:
        import torch.distributed as dist
        return dist.get_rank(self.group)

    def point_range(self):
        """[lo, hi) of the SRS indices this rank commits over (the fixed partition commit_jacobian uses)."""
        import torch.distributed as dist
        return _sharding.shard_range(self.ck.length, dist.get_rank(self.group), dist.get_world_size(self.group))

    def all_gather_fr(self, values) -> list:
        """Every rank's list of field elements (canonical ints, same count on every rank) -> [rank][i]; 32 bytes per element."""
        import torch
        import torch.distributed as dist
        world = dist.get_world_size(self.group)
        k = len(values)
        t = torch.tensor([(int(v) >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for v in values for j in range(4)], dtype=torch.uint64).view(torch.int64)
        if self.device is not None:
            t = t.to(self.device)
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t, group=self.group)
        out = []
        for p in parts:
            w = p.cpu().numpy().view(np.uint64).reshape(k, 4)
            out.append([sum(int(w[i, j]) << (64 * j) for j in range(4)) for i in range(k)])
        return out

    def commit_jacobian_slices(self, slices) -> np.ndarray:
        """Like commit_jacobian, for polynomials of which this rank holds ONLY its coefficient range: slices[i] = coefficients
        [lo, lo + len(slices[i])) of polynomial i, lo = point_range()[0] (an empty slice: nothing of it falls into the range)."""
        import torch
        import torch.distributed as dist
        world = dist.get_world_size(self.group)
        k, L = len(slices), self.c.fq_limbs
        lo_r, hi_r = self.point_range()
        if self.slice_srs and self._slice is None and hi_r > lo_r:
            self._slice = self.ck.slice(lo_r, hi_r - lo_r)               # mzk_srs_slice: a device copy of this rank's range
            self._slice_lo = lo_r
        sl = [s.contiguous() if hasattr(s, "contiguous") else np.ascontiguousarray(s) for s in slices]
        assert all(int(s.shape[0]) <= hi_r - lo_r for s in sl)
        part = np.ascontiguousarray(self._local(sl, [lo_r] * k), dtype=np.uint64).reshape(k, 3, L)
        stacked = _sharding.gather_partials(part, self.group, self.device)
        return np.stack([_sharding.sum_jacobian(self.c, stacked[:, i]) for i in range(k)])

    def release(self):
        if self._slice is not None:
            self._slice.release()
            self._slice = None


# ---- snark-level entry points on the mirror ---------------------------------------------------------------------------------------------
def preprocess(commit_key, circuit, quotient_classes=None, quotient_gather=None, quotient_shard=None, lagrange: bool | None = None):
    """snark.preprocess for the mirror: interpolate selectors, sigmas (and Plookup tables), keep them with the commit key."""
    snark = mj.snark
    c, n = circuit.curve, circuit.n
    if commit_key.length < n + 3:
        raise ValueError("SRS too small: need domain size + 3 powers (srs.rs:88)")
    if commit_key.length > n + 3:
        commit_key = commit_key.trim(n + 2)
    dom = Radix2EvaluationDomain(c, n.bit_length() - 1)
    sel, sig = circuit.selector_values.clone(), circuit.sigma_values.clone()
    dom.ifft_in_place(sel)
    dom.ifft_in_place(sig)
    host = lambda t: t.cpu().numpy().view(np.uint64)
    plookup = None
    if circuit.table_values is not None:
        tab = circuit.table_values.clone()
        dom.ifft_in_place(tab)
        tab_h = host(tab)
        plookup = {name: tab_h[i] for i, name in enumerate(("range_table_poly", "key_table_poly", "table_dom_sep_poly", "q_dom_sep_poly"))}
    pk = TurboPlonkProver(c, n, list(host(sel)), list(host(sig)), circuit.k, commit_key, plookup=plookup,
                          quotient_classes=quotient_classes, quotient_gather=quotient_gather, quotient_shard=quotient_shard)
    if lagrange is None:
        lagrange = n >= snark.LAGRANGE_MIN_DOMAIN and snark.witness_is_small(c, circuit.wire_values)
    if lagrange and quotient_shard is None and quotient_gather is None and commit_key.offset == 0:
        pk.lagrange_ck = commit_key.lagrange_key(n)
    return pk


def prove(rng, circuit, pk, extra_transcript_init_msg: bytes | None = None, profile: bool = False):
    """PlonkKzgSnark::prove on the mirror: (ProofCore, compressed proof bytes)."""
    snark = mj.snark
    blind = snark.draw_blinders(circuit.curve, rng, circuit.num_wire_types, pk.ultra)
    src = TranscriptChallenges(pk, circuit.public_input, extra_transcript_init_msg)
    core = pk.prove(circuit.wire_values, circuit.pub_input_values, src, blind, profile=profile, pi_zero=not any(circuit.public_input))
    return core, snark.serialize_proof(circuit.curve, core)
