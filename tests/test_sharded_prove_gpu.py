"""GPU: SURVEY.md 8(e).1 end to end -- two ranks (both on the box's one GPU, gloo for the 144-byte collectives) run
PlonkKzgSnark::prove with every commitment's MSM split by point range (8(e).1) and the quotient domain split into residue
classes with one exchange (8(e).3); both must emit exactly the proof bytes of the single-process run."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, curve_id, plonk_type, num_gates, out_dir):
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
        if world > 1:
            gather = lambda local: mj.sharding.gather_quotient_classes(local, via_host=True)
            pk = mj.snark.preprocess(ck, cs, quotient_classes=mj.sharding.class_range(rank, world), quotient_gather=gather)
            pk.committer = mj.sharding.ShardedCommitter(c, ck)
        else:
            pk = mj.snark.preprocess(ck, cs)
        _, proof_bytes = mj.snark.prove(rng, cs, pk)
        with open(os.path.join(out_dir, f"proof_{world}_{rank}.bin"), "wb") as f:
            f.write(proof_bytes)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("curve_id,plonk_type,num_gates", [(0, "TurboPlonk", 1 << 12), (1, "UltraPlonk", 1 << 11)])
def test_sharded_prove_matches_single_process(gpu, tmp_path, curve_id, plonk_type, num_gates):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() + num_gates) % 2000
    ctx = mp.get_context("spawn")
    for world in (1, 2):
        mp.spawn(_worker, args=(world, port + world, curve_id, plonk_type, num_gates, str(tmp_path)), nprocs=world, join=True)
    single = (tmp_path / "proof_1_0.bin").read_bytes()
    assert len(single) > 500
    for rank in range(2):
        assert (tmp_path / f"proof_2_{rank}.bin").read_bytes() == single, rank


@pytest.mark.parametrize("curve_id,plonk_type,num_gates,classes", [(0, "TurboPlonk", 1 << 10, list(range(8))), (1, "UltraPlonk", 1 << 9, list(range(8))),
                                                                  (0, "TurboPlonk", 64, [1, 4, 6])])
def test_chunked_quotient_single_process(gpu, mj, curve_id, plonk_type, num_gates, classes):
    """All 8 residue classes on one GPU: class-wise size-n coset NTTs + fused kernel + local inverse + 8-point iDFT must give
    the very coefficients of the whole-domain quotient (coset FFT(8n) path).  A key holding 3 classes must produce those
    classes' remainders t mod (X^n - h_k^n) of the same quotient."""
    import torch
    c = mj.params.CURVES[curve_id]
    cs = mj.snark.gen_circuit_for_bench(c, num_gates, plonk_type)
    n = cs.n
    rng = mj.rng.test_rng()
    ck = mj.UnivariateProverParam.gen_srs_for_testing(c, mj.rng.fr_rand(c, rng), n + 2)
    pk0 = mj.snark.preprocess(ck, cs)
    core0, bytes0 = mj.snark.prove(mj.rng.test_rng(), cs, pk0)
    quot0 = pk0.last["quot"].clone()
    pk1 = mj.snark.preprocess(ck, cs, quotient_classes=classes)
    if len(classes) == 8:
        core1, bytes1 = mj.snark.prove(mj.rng.test_rng(), cs, pk1)
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
