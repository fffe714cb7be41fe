"""GPU: SURVEY.md 8(e).1 end to end -- two ranks (both on the box's one GPU, gloo for the 144-byte collectives) run
PlonkKzgSnark::prove with every commitment's MSM split by point range; both must emit exactly the proof bytes of the
single-process run."""
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
        pk = mj.snark.preprocess(ck, cs)
        if world > 1:
            pk.committer = mj.sharding.ShardedCommitter(c, ck)
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
