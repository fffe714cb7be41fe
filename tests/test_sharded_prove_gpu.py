"""GPU: SURVEY.md 8(e) end to end -- 2, 4 and 5 ranks (all on the box's one GPU, gloo for the collectives; the pool allows at most
six processes on a card and the test runner is one of them, so worlds of 6 and 8 are rehearsed on the CPU side only:
tests/test_sharding.py) run PlonkKzgSnark::prove with
every commitment's MSM split by point range (8(e).1) and the needed residue classes of the quotient domain -- 6 of 8 for
TurboPlonk, 7 for UltraPlonk -- split over the ranks with one exchange (8(e).3; with 4 or 5 ranks some own fewer classes, or none);
every rank must emit exactly the proof bytes of the single-process run."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, curve_id, plonk_type, num_gates, out_dir, device_payloads=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import mpc_jellyfish_amd as mj
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c = mj.params.CURVES[curve_id]
        cs = mj.snark.gen_circuit_for_bench(c, num_gates, plonk_type)
        rng = mj.rng.test_rng()
        ck = mj.UnivariateProverParam.gen_srs_for_testing(c, mj.rng.fr_rand(c, rng), cs.n + 2)
        # (i) the product: this process is one rank of the library's own rounds (mzk_comm over torch.distributed: sharding.TorchComm),
        # keeping only its point range of the SRS
        # device_payloads: the collectives' payloads in CUDA tensors, as under RCCL (staging tensors filled / drained by device-to-device
        # copies, the class exchange device to device) -- gloo carries CUDA tensors too, so the one-GPU box rehearses that code path
        import torch
        comm_dev = torch.device("cuda", 0) if device_payloads else None
        pk = mj.snark.preprocess(ck, cs, comm=mj.sharding.TorchComm(device=comm_dev) if world > 1 else None)
        g1 = mj.rng.test_rng()
        mj.rng.fr_rand(c, g1)
        _, proof_bytes = mj.snark.prove(g1, cs, pk)
        with open(os.path.join(out_dir, f"proof_{world}_{rank}.bin"), "wb") as f:
            f.write(proof_bytes)
        pk.release()
        if device_payloads:
            return
        # (ii) the test-side sequencing of the primitives with the torch.distributed committer (tests/mirror_prover.py)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import mirror_prover as MP
        if world > 1:
            gather = lambda local, n_classes: mj.sharding.gather_quotient_classes(local, via_host=True, n_classes=n_classes)
            mk = MP.preprocess(ck, cs, quotient_shard=(rank, world), quotient_gather=gather)
            mk.committer = MP.ShardedCommitter(c, ck)
        else:
            mk = MP.preprocess(ck, cs)
        _, proof_bytes = MP.prove(rng, cs, mk)
        with open(os.path.join(out_dir, f"mirror_{world}_{rank}.bin"), "wb") as f:
            f.write(proof_bytes)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("curve_id,plonk_type,num_gates,worlds", [(0, "TurboPlonk", 1 << 12, (2, 4)), (1, "UltraPlonk", 1 << 11, (2, 5)),
                                                                 (0, "TurboPlonk", 1 << 10, (5,))])
def test_sharded_prove_matches_single_process(gpu, tmp_path, curve_id, plonk_type, num_gates, worlds):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() + num_gates) % 2000
    for world in (1,) + tuple(worlds):
        mp.spawn(_worker, args=(world, port + world, curve_id, plonk_type, num_gates, str(tmp_path)), nprocs=world, join=True)
    single = (tmp_path / "proof_1_0.bin").read_bytes()
    assert len(single) > 500
    assert (tmp_path / "mirror_1_0.bin").read_bytes() == single
    for world in worlds:
        for rank in range(world):
            assert (tmp_path / f"proof_{world}_{rank}.bin").read_bytes() == single, (world, rank)
            assert (tmp_path / f"mirror_{world}_{rank}.bin").read_bytes() == single, ("mirror", world, rank)


@pytest.mark.parametrize("curve_id,plonk_type,num_gates,world", [(0, "TurboPlonk", 1 << 12, 2), (1, "UltraPlonk", 1 << 11, 3), (0, "TurboPlonk", 1 << 10, 5)])
def test_sharded_prove_with_device_payloads(gpu, tmp_path, curve_id, plonk_type, num_gates, world):
    """The library's rounds as one rank of a multi-process proof with the collectives' payloads in CUDA tensors -- the form
    sharding.TorchComm / sharding.gather_partials take under RCCL (bench.py --gpus N): same proof bytes as the single-process run."""
    import torch.multiprocessing as mp
    port = 31500 + (os.getpid() + num_gates) % 2000
    mp.spawn(_worker, args=(1, port, curve_id, plonk_type, num_gates, str(tmp_path), True), nprocs=1, join=True)
    mp.spawn(_worker, args=(world, port + world, curve_id, plonk_type, num_gates, str(tmp_path), True), nprocs=world, join=True)
    single = (tmp_path / "proof_1_0.bin").read_bytes()
    assert len(single) > 500
    for rank in range(world):
        assert (tmp_path / f"proof_{world}_{rank}.bin").read_bytes() == single, (world, rank)


@pytest.mark.parametrize("curve_id,plonk_type,num_gates,classes", [(0, "TurboPlonk", 1 << 10, list(range(8))), (1, "UltraPlonk", 1 << 9, list(range(8))),
                                                                  (0, "TurboPlonk", 1 << 10, None), (1, "UltraPlonk", 1 << 9, None),
                                                                  (1, "TurboPlonk", 1 << 7, [0, 2, 3, 5, 6, 7]), (0, "UltraPlonk", 1 << 8, [1, 2, 3, 4, 5, 6, 7]),
                                                                  (1, "TurboPlonk", 1 << 7, [0, 2, 3, 5, 7]), (0, "UltraPlonk", 1 << 8, [1, 2, 4, 5, 6, 7]),
                                                                  (0, "TurboPlonk", 64, [1, 4, 6])])
def test_chunked_quotient_single_process(gpu, mj, curve_id, plonk_type, num_gates, classes):
    """Residue classes on one GPU: class-wise size-n coset NTTs + fused kernel + local inverse + the inverse Vandermonde per
    coefficient must give the very coefficients of the whole-domain quotient (coset FFT(8n) path, `quotient_classes="whole"`) --
    from all 8 classes (an 8-point iDFT), from W + 1 of them (6 of 8 for TurboPlonk, 7 for UltraPlonk: deg t = W (n + 1) + 2 needs no
    more, prover.rs:916-919), and from the default W classes together with the W + 3 top coefficients that mzk_plonk_quotient_top_dev
    takes from the numerator (contiguous and scattered class sets).  A key holding 3 classes must produce
    those classes' remainders t mod (X^n - h_k^n) of the same quotient."""
    import torch
    c = mj.params.CURVES[curve_id]
    cs = mj.snark.gen_circuit_for_bench(c, num_gates, plonk_type)
    n = cs.n
    rng = mj.rng.test_rng()
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, mj.rng.fr_rand(c, rng), n + 2)
    import mirror_prover as MP
    pk0 = MP.preprocess(ck, cs, quotient_classes="whole")
    core0, bytes0 = MP.prove(mj.rng.test_rng(), cs, pk0)
    quot0 = pk0.last["quot"].clone()
    pk1 = MP.preprocess(ck, cs, quotient_classes=classes)
    W = 6 if plonk_type == "UltraPlonk" else 5
    if classes is None:
        assert pk1.own_classes == list(range(W))                     # W classes + the top W + 3 coefficients from the numerator (round 3)
    if classes is None or len(classes) >= W:
        core1, bytes1 = MP.prove(mj.rng.test_rng(), cs, pk1)
        assert torch.equal(pk1.last["quot"], quot0)
        assert bytes1 == bytes0
    else:
        # remainders of the known quotient modulo X^n - c_k, computed with big ints
        r = c.r
        q_int = mj.params.fr_from_mont(c, quot0.cpu().numpy().view(np.uint64))
        w_m = pow(c.fr_generator, (r - 1) // (8 * n), r)
        # replay round 3 inputs: the prover object keeps the coefficient slab of its last proof
        ch = mj.plonk.Challenges(*(pk0.last_challenges[x] for x in ("alpha", "beta", "gamma", "tau")))
        local = mj.plonk.compute_quotient_chunked_dev(pk1.pk, ch, pk0._keep.contiguous(), n + 3)
        got = mj.params.fr_from_mont(c, local.cpu().numpy().view(np.uint64).reshape(-1, 4))
        for lc, k in enumerate(classes):
            ck_ = pow(c.fr_generator * pow(w_m, k, r) % r, n, r)
            want = [sum(q_int[j + q * n] * pow(ck_, q, r) for q in range(8)) % r for j in range(n)]
            assert got[lc * n:(lc + 1) * n] == want, k
    pk0.release(); pk1.release(); ck.release()
