# tools/rehearse_two_ranks.sh [tag] -- `python bench.py --gpus 2` at the driver's default sizes, both ranks on the ONE card of the box, gloo carrying
# CUDA tensors (MZK_BENCH_GLOO_CUDA=1: the payload handling of the RCCL path) -> gpurun_out/<tag>_bench_2ranks_rehearsal.json.  Figures of this run
# are NOT scaling numbers (two ranks share one GPU); what it shows is that the N > 1 line is complete and the sharded proof is the single-GPU proof.
cd $GRAFT_REPO_ROOT
T=${1:-r05_e}
export MZK_BENCH_BACKEND=gloo MZK_BENCH_SINGLE_DEVICE=1 MZK_BENCH_GLOO_CUDA=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 700 python3 bench.py --gpus 2 --steps 10 --warmup 3 > gpurun_out/${T}_bench_2ranks_rehearsal.json 2> gpurun_out/${T}_bench_2ranks_rehearsal.err
echo "rc=$?"
python3 - <<P
import json
d = json.loads(open("gpurun_out/${T}_bench_2ranks_rehearsal.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("n_gpus", "value", "ms_per_step", "prove_replicas_proofs_per_s", "prove_replicas_ranks_agree", "prove_sharded_ms", "prove_sharded_ranks_agree",
                             "prove_sharded_same_bytes_as_single_gpu", "prove_cpp_host_multi_gpu_ms", "prove_cpp_host_multi_gpu_speedup", "prove_cpp_host_multi_gpu_same_bytes")})
print(d.get("prove_sharded", {}).get("error"), d.get("prove_replicas", {}).get("error"))
P
