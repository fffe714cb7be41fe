"""GPU: bench.py's one-line JSON contract at toy sizes (the driver depends on it)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys(gpu):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--log-n", "14", "--plonk-log-n", "10", "--ultra-log-n", "10", "--cpu-prove-log-n", "11",
                          "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["value"] > 0
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert d["roofline"]["bound"] in ("hbm", "mfma") and "workload" in d["config"]
    cb = d["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(cb) and cb["kind"] in ("port", "reference") and cb["matches_gpu"] is True
    assert cb["prove_2p11"]["matches_gpu"] is True and cb["prove_c1"]["matches_gpu"] is True
    # the evidence as FLAT scalars where the driver's record keeps them -- the host's CPU, every CPU figure at one and at
    # all threads, the proof-level figures (general case = dense witness) -- under cpu_baseline / roofline and at the top level
    for key in ("cpu_model", "cpu_threads_host", "cpu_threads_used", "value_1_thread", "cpu_msm_2p20_1_thread_ms", "cpu_ntt_2p22_ms", "cpu_ntt_2p22_1_thread_ms",
                "cpu_prove_c1_ms", "cpu_prove_2p11_ms", "gpu_prove_ms", "gpu_prove_dense_witness_ms", "gpu_ntt_2p22_ms"):
        assert cb.get(key) not in (None, ""), key
    assert cb["cores"] == cb["cpu_threads_used"] >= 1 and cb["cpu_ntt_2p22_matches_gpu_whole_vector"] is True and cb["cpu_prove_2p11_matches_gpu"] is True
    for key in ("prove_ms", "prove_dense_witness_ms", "prove_coefficient_commit_ms", "ntt_2p22_ms", "kernel_launches_per_proof", "hbm_total_bytes"):
        assert d[key] == d["roofline"][key] and d[key] > 0, key
    assert not any(isinstance(v, (dict, list)) for k, v in cb.items() if k.startswith(("cpu_", "gpu_", "value")))
    # round 5: proofs in flight on the one card, and what a dense-witness prover holds in HBM, as flat scalars too
    assert d["prove_proofs_per_s_1_in_flight"] > 0 and d["prove_proofs_per_s_2_in_flight"] > 0 and d["prove_in_flight_contexts_agree_on_proof"] is True
    assert 0 < d["hbm_total_bytes_dense_witness_prover"] and d["hbm_total_bytes"] == d["roofline"]["hbm_total_bytes"] > 0
    assert d["roofline"]["launches_per_msm"] > 0 and d["prove"]["round_level_abi_same_proof_bytes"] is True
    assert big_keys(d), "precompute cost, variable-base leg, shim-only leg"
    assert d["prove_cpp_host"]["turbo_bls12_381"]["proof_bytes"] == d["prove"]["proof_bytes"]
    assert d["prove_cpp_host"]["ultra_bn254"]["proof_bytes"] == d["prove_ultra_bn254"]["proof_bytes"]
    assert all(v.get("proof_matches_python_mirror") is True for v in d["prove_cpp_host"].values()), d["prove_cpp_host"]


def big_keys(d):
    """what VERDICT r1 / r2 asked the line to carry: the headline on the variable-base path, the fixed-base table path with its
    cost as a named secondary, the shim-only figure, reps, and a proof from a host-resident and from a dense witness"""
    assert "VARIABLE-BASE" in d["config"]["workload"] and "precompute" not in d["config"]
    fb = d["fixed_base"]
    pc = fb["precompute"]
    assert pc["levels"] >= 1 and pc["table_bytes"] > 0 and pc["build_ms"] > 0
    assert fb["value"] > 0 and fb["same_point_as_variable_base"] is True and fb["roofline"]["bound"] == "hbm"
    assert d["cpu_baseline"]["gpu_over_cpu"] > 0
    assert d["prove"]["from_host_witness_ms"] > 0 and d["prove"]["from_host_witness_same_proof_bytes"] is True and d["prove"]["dense_witness_ms"] > 0
    assert set(d["prove_dropin"]["ms"]) == {"pageable", "pinned", "batch", "four_site"} and all(v > 0 for v in d["prove_dropin"]["ms"].values())
    assert d["prove"]["reps"] == 10 and d["prove"]["min_ms"] <= d["prove"]["prove_ms"] <= d["prove"]["max_ms"] * 1.5
    return True


def _two_ranks(extra_args, port, env_extra=None):
    """port = None: `python bench.py --gpus 2` with NO launcher around it, as the driver's command line reads -- bench.py starts its
    own two workers (bench.self_launch); otherwise wrapped in torch.distributed.run, the contract's other form."""
    env = dict(os.environ, MZK_BENCH_BACKEND="gloo", MZK_BENCH_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.update(env_extra or {})
    launcher = [] if port is None else ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                        "--master-port", str(port)]
    cmd = [sys.executable] + launcher + [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--log-n", "14", "--plonk-log-n", "10", "--ultra-sharded-log-n", "10",
                                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + extra_args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    return out, lines


def test_bench_two_ranks_rehearsal_and_watchdog(gpu):
    """The N > 1 launch of the contract, rehearsed with two ranks on the one GPU over gloo: one JSON line from rank 0 with the
    whole-job value and the sharded proofs; and with a secondary section that cannot finish in time, the watchdog still prints
    the headline line (the driver's SCALE run must never end without one)."""
    # no launcher: bench.py --gpus 2 launches its own workers; MZK_BENCH_GLOO_CUDA=1: the collectives' payloads in CUDA tensors, as under RCCL
    out, lines = _two_ranks([], None, {"MZK_BENCH_GLOO_CUDA": "1"})
    assert out.returncode == 0, out.stderr[-3000:]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    sh = d["prove_sharded"]
    assert "error" not in sh and sh["turbo_bls12_381"]["ranks_agree_on_proof"] and sh["ultra_bn254"]["ranks_agree_on_proof"]
    assert sh["same_proof_bytes_as_single_gpu_replicas"] is True and d["prove_sharded_same_bytes_as_single_gpu"] is True and d["prove_sharded_ms"] > 0
    rp = d["prove_replicas"]                              # N independent provers, one per rank: proofs/s, the same proof bytes everywhere
    assert "error" not in rp and rp["ranks_agree_on_proof"] is True and rp["proofs_per_s"] > 0 and d["prove_replicas_proofs_per_s"] == rp["proofs_per_s"]
    cm = d["prove_cpp_host_multi_gpu"]                    # the compiled host driving both (virtual) devices from one process
    assert cm["same_proof_bytes"] is True and cm["gpus_2"]["prove_ms"] > 0, cm
    out, lines = _two_ranks(["--secondary-timeout", "1", "--plonk-log-n", "16", "--ultra-sharded-log-n", "16"], 29633)
    assert len(lines) == 1, (lines, out.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "watchdog" in d["prove_sharded"]["error"]


def test_self_launch_reports_a_failed_worker(gpu):
    """`bench.py --gpus 2` without a launcher must end non-zero when a worker does (here: an unknown rehearsal backend)."""
    env = dict(os.environ, MZK_BENCH_BACKEND="no-such-backend", MZK_BENCH_SINGLE_DEVICE="1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--log-n", "12", "--steps", "1", "--warmup", "0", "--no-plonk",
                          "--no-ntt", "--no-cpu-baseline", "--no-fixed-base", "--no-batch"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0
