#!/usr/bin/env python3
"""bench.py -- headline benchmark of the jf-plonk hot path on MI355X.

Workload (BASELINE.json configs[1], "Standalone MSM"): 2^20 random G1 points x Fr scalars on BLS12-381, bases and scalars resident
in HBM when the timed region starts; one step = one MSM of 2^20 pairs per GPU through the C ABI (mzk_msm_dev).  With N > 1 ranks
the N*2^20 pairs are sharded by point range (one shard per GPU, weak scaling) and each step ends with the all-gather + local EC
sum of the N partial points (mpc-jellyfish_amd/sharding.py).

`value`, `ms_per_step` and `roofline` are the VARIABLE-BASE path: plain Pippenger on the registered bases with nothing
precomputed per base -- what configs[1] states, what ark-ec's VariableBaseMSM does and what `cpu_baseline` runs.  The library's
default for a registered SRS (KZG commit keys never change) is a FIXED-BASE table of precomputed multiples; the same K steps on
that path are the `fixed_base` object, with what the table costs (levels, bytes of HBM, build time -- paid once per SRS, outside
its timed region) in `fixed_base.precompute`.

Alongside, untimed by `value`: NTT 2^22 (configs[2]), round 3, batch commit, PlonkKzgSnark::prove on the reference's bench
circuit (configs[3]; 10 timed repetitions as plonk/benches/bench.rs:25), the same proof in shim-only mode (host pointers,
INTEGRATION.md section 2), UltraPlonk/BN254, the C++ host, and the CPU restatement on the host cores -- including ONE CPU proof
of the same 2^20-gate circuit, so that the line itself carries the GPU/CPU ratio the north star is quoted on.

    python bench.py [--gpus N --steps K --warmup W] [--log-n 20] [--no-cpu-baseline]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset) starts its own N workers: this process touches no GPU, runs
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child, relays rank 0's line and exits with the child's code.

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import srchash  # noqa: E402  (tools/srchash.py: hash of the kernel sources a PMC pass was collected on)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "r05_traffic.json")
VOP3_NS = 1.9                  # measured cost of one VOP3 wave-instruction per SIMD (profiles/r01_valu_ubench2.txt); 1024 SIMDs


def _pmc(key, sources):
    """(value, note) of one figure of the committed PMC passes (profiles/r05_traffic.json, written by tools/pmc_passes.sh +
    tools/pmc_aggregate.py: PMC counters cannot be read inside the timed run).  The file carries the hash of the kernel sources
    it was collected on; when those sources have changed since, the stale number is NOT quoted: None and a note saying so."""
    try:
        d = json.load(open(TRAFFIC_JSON))
    except Exception:
        return None, "no committed PMC pass (profiles/r05_traffic.json missing)"
    name = "msm" if sources is srchash.MSM_SOURCES else "ntt"
    want = (d.get("source_sha16") or {}).get(name)
    have = srchash.sha16(sources)
    if want != have:
        return None, "PMC pass is stale: collected on %s sources %s, the tree has %s (re-run tools/pmc_passes.sh)" % (name, want, have)
    return d.get(key), "%s, collected on %s sources %s" % (os.path.basename(TRAFFIC_JSON), name, have)


def host_cpu():
    """CPU model and the hardware threads this process may use -- what Rayon's default pool takes (utilities/src/par_utils.rs:20-29,
    scripts/run_benchmarks.sh:88: RAYON_NUM_THREADS = all): the logical CPUs of the host, cut by the affinity mask and by a cgroup quota."""
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    logical = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = logical
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            usable = max(1, min(usable, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return model, logical, usable


def self_launch(n_gpus):
    """`python bench.py --gpus N` with no launcher around it: this process stays off the GPU (nothing below imports torch or the
    library) and starts N FRESH worker processes -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py <same
    arguments>`, one rank per GPU over RCCL -- as a child, relays their output (rank 0 prints the one JSON line) and returns the
    child's exit code, non-zero if any worker failed.  A process that has initialised the GPU is never replaced by another."""
    import socket
    import subprocess
    with socket.socket() as sk:                                       # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")                 # dmabuf IPC: what RCCL needs between the ranks of one node
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = 0
    for line in proc.stdout:                                          # stderr goes straight through; stdout is relayed line by line
        if line.lstrip().startswith("{"):
            lines += 1
        sys.stdout.write(line)
        sys.stdout.flush()
    rc = proc.wait()
    if rc == 0 and lines != 1:
        print("bench.py: the workers printed %d JSON lines, expected 1" % lines, file=sys.stderr)
        return 4
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=20, help="log2 of the MSM size per GPU (default 2^20 = config C2)")
    ap.add_argument("--ntt-log-n", type=int, default=22, help="log2 of the NTT size for the secondary measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-prove-log-n", type=int, default=20, help="log2 gates of the CPU-vs-device whole-proof comparison (0 disables; "
                    "~60-70 s of CPU on 16 threads at 20 = the north star's configuration, ~4 s at 16)")
    ap.add_argument("--no-ntt", action="store_true")
    ap.add_argument("--no-plonk", action="store_true", help="skip the proof-level legs (round 3, prove, drop-in, UltraPlonk, C++ host)")
    ap.add_argument("--no-fixed-base", action="store_true", help="skip the fixed-base (precomputed table) repetition of the headline steps")
    ap.add_argument("--no-batch", action="store_true", help="skip the batch_commit5 leg (with the other --no-* switches the run is the headline's "
                    "launches alone: what the rocprofv3 --stats average of msm_accumulate_kernel is compared with)")
    ap.add_argument("--plonk-log-n", type=int, default=20)
    ap.add_argument("--prove-reps", type=int, default=10, help="timed repetitions of PlonkKzgSnark::prove (plonk/benches/bench.rs:25 uses 10)")
    ap.add_argument("--ultra-log-n", type=int, default=20, help="UltraPlonk/BN254 prove leg (0 disables; config C5 is 22)")
    ap.add_argument("--no-dropin", action="store_true", help="skip the shim-only (host-pointer) leg")
    ap.add_argument("--link-batch", action="store_true", help="also time link_proofs / batch_prove (outside SURVEY.md section 8; off by default)")
    ap.add_argument("--secondary-timeout", type=int, default=420, help="N > 1: seconds after which a stalled secondary (sharded prove) section "
                    "is abandoned, the headline line printed as it stands and the process ended with a non-zero code (0 disables)")
    ap.add_argument("--ultra-sharded-log-n", type=int, default=22, help="N > 1: UltraPlonk/BN254 sharded prove leg (config C5: 22; 0 disables)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))                     # before torch or the library are even imported: no GPU call in this process

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libmi355zk has no CPU fallback)")
    # rehearsal switches (not used by the driver): all ranks on GPU 0 with gloo collectives, to exercise the N > 1 code on a 1-GPU box
    backend = os.environ.get("MZK_BENCH_BACKEND", "nccl")
    if os.environ.get("MZK_BENCH_SINGLE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # MZK_BENCH_GLOO_CUDA=1 (rehearsal): gloo carrying CUDA tensors -- the payloads live where they live under RCCL, so the one-GPU box runs
    # the very tensor handling of the nccl path (staging copies, device-to-device class exchange)
    dev_payloads = backend == "nccl" or os.environ.get("MZK_BENCH_GLOO_CUDA") == "1"
    cdev = dev if dev_payloads else torch.device("cpu")                # where collective payloads live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import mpc_jellyfish_amd as mj
    from importlib import import_module
    import ctypes as C
    mlib = import_module("mpc-jellyfish_amd.lib")
    L = mlib.init(local_rank)
    coll_dev = dev if dev_payloads else None

    curve = mj.params.BLS12_381
    n = 1 << args.log_n
    # ---- synthetic inputs, generated on the device side of the boundary (no oracle involved) --------
    # bases: a testing SRS [beta^i]G (srs.rs:118-153) -- distinct, on-curve, in-subgroup points.
    # Each rank builds the n bases of its own point-range shard from its own trapdoor.
    beta = 0x1f3a5c7e9b2d4f6081a3c5e7092b4d6f8091a2b3c4d5e6f708192a3b4c5d6e7f % curve.r
    shard_beta = beta * pow(3, rank, curve.r) % curve.r
    t0 = time.time()
    pp = mj.UnivariateProverParam.gen_srs_for_testing(curve, shard_beta, n - 1)
    t_srs = time.time() - t0
    scalars = mj.params.random_fr_mont(curve, n, seed=0x6d7a6b5f + rank)      # uniform in [0, r)
    d_scalars = torch.from_numpy(scalars.view(np.int64)).to(dev)
    torch.cuda.synchronize()

    def step():
        jac = mj.msm_bigint(pp, d_scalars, scalars_are_mont=True)            # one Pippenger MSM, result on host
        if world > 1:
            jac = mj.sharding.all_gather_sum(curve, jac, device=coll_dev)
        return jac

    def launch_count():
        v = C.c_uint64()
        mlib.check(L.mzk_launch_count(C.byref(v)), "mzk_launch_count")
        return int(v.value)

    def timed_steps():
        """W warm-ups, then exactly K steps between barrier + synchronize on both sides; max over ranks."""
        for _ in range(args.warmup):
            step()
        L.mzk_profile_reset()
        L.mzk_profile_enable(1)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        l0 = launch_count()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        L.mzk_profile_enable(0)
        launches = (launch_count() - l0) // max(args.steps, 1)
        if world > 1:
            tmax = torch.tensor([el], dtype=torch.float64, device=cdev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        acc_ms, acc_cnt = mlib.profile_get("msm_accumulate")
        tot_ms, tot_cnt = mlib.profile_get("msm_total")
        sort_ms, _ = mlib.profile_get("msm_sort")
        red_ms, _ = mlib.profile_get("msm_reduce")
        comb_ms, _ = mlib.profile_get("msm_split_combine")
        phases = {"sort": round(sort_ms / max(tot_cnt, 1), 4), "accumulate": round(acc_ms / max(acc_cnt, 1), 4),
                  "split_combine": round(comb_ms / max(tot_cnt, 1), 4),
                  "reduce": round(red_ms / max(tot_cnt, 1), 4), "device_total": round(tot_ms / max(tot_cnt, 1), 4),
                  "kernel_launches": launches}
        return el, res, acc_ms / max(acc_cnt, 1), phases, mlib.msm_last_shape()

    # With several ranks every leg holds collectives (barriers around the timed steps, the sharded proofs): if a rank fails or stalls,
    # the others would wait for ever and no line would be printed -- a watchdog prints the headline as it stands (once it exists) and ends
    # the process with a NON-ZERO code (a stalled section is not a success).  Started before the first leg.
    out = None
    emit_lock = threading.Lock()
    state = {"printed": False}

    def emit():
        with emit_lock:
            if rank == 0 and not state["printed"]:
                state["printed"] = True
                print(json.dumps(out), flush=True)

    def bail():
        if rank == 0:
            with emit_lock:
                if not state["printed"] and out is not None:
                    out["prove_sharded"] = {"error": "watchdog: the secondary multi-rank section did not finish in %d s" % args.secondary_timeout}
                    state["printed"] = True
                    print(json.dumps(out), flush=True)
        os._exit(3)

    watchdog = None
    if world > 1 and args.secondary_timeout > 0:
        watchdog = threading.Timer(args.secondary_timeout, bail)
        watchdog.daemon = True
        watchdog.start()

    table_off_only = os.environ.get("MZK_BENCH_TABLE") == "0"   # tools/pmc_passes.sh: PMC passes of the headline kernels alone
    # ---- secondary leg, run FIRST: the same W + K steps on the library's default path for a registered SRS (a fixed-base table of
    #      precomputed multiples, built here explicitly and timed apart).  The two legs are independent (the switch below decides per MSM
    #      whether the table is looked at); this one goes first because a card coming out of idle runs its first ~10 steps 5-7 % slower
    #      (clock ramp: profiles/r05_f_warmup_ab.txt -- headline 3.48 ms after 2 warm-ups, 3.36 after 5, 3.31-3.32 after 50 or 200), and
    #      the headline should be read at the sustained clocks a prover sees, not at the ramp.  --no-fixed-base gives the cold figure.
    fb_raw = None
    if not args.no_fixed_base and not table_off_only:
        L.mzk_msm_set_precompute(1)
        pc_bits, pc_levels, pc_bytes, pc_ms = C.c_uint32(), C.c_uint32(), C.c_uint64(), C.c_double()
        mlib.check(L.mzk_srs_precompute(pp.handle, C.byref(pc_bits), C.byref(pc_levels), C.byref(pc_bytes), C.byref(pc_ms)), "mzk_srs_precompute")
        fb_raw = (pc_bits.value, pc_levels.value, pc_bytes.value, pc_ms.value) + tuple(timed_steps())
    # the headline steps run with the fixed-base table OFF: nothing is precomputed per base (BASELINE configs[1], ark-ec's VariableBaseMSM)
    L.mzk_msm_set_precompute(0)
    elapsed, result, acc_avg_ms, phases, (c_bits, n_win, n_buckets) = timed_steps()
    alg_bytes = 128.0 * n                          # N * (2*|Fq| + 32) bytes per BLS12-381 MSM

    def msm_roofline(acc_ms, windows, pmc_prefix):
        achieved = alg_bytes / (acc_ms * 1e-3) / 1e9
        traffic, note = _pmc(pmc_prefix + "_hbm_bytes_per_launch", srchash.MSM_SOURCES)
        insts, _ = _pmc(pmc_prefix + "_SQ_INSTS_VALU", srchash.MSM_SOURCES)
        valu = None
        if insts and args.log_n == 20:
            bound_ms = insts / 1024 * VOP3_NS * 1e-6
            # 32-bit integer multiply-adds (v_mad_u64_u32) the launch retires: windows x n mixed adds x (8 products of 2 * 14^2 MADs
            # + 2 squarings of 14 * 15 / 2 + 14^2), SURVEY.md 8(d)'s second figure; peak = 64 lanes / 1.9 ns on 1024 SIMDs
            mads = windows * n * (8 * 2 * 196 + 2 * (105 + 196))
            valu = {"wave_insts_per_launch": insts, "issue_bound_ms": round(bound_ms, 3), "util": round(bound_ms / acc_ms, 3),
                    "int_mad_per_s": round(mads / (acc_ms * 1e-3), -9), "int_mad_peak_per_s": round(1024 * 64 / (VOP3_NS * 1e-9), -9),
                    "note": "SQ_INSTS_VALU of the committed PMC pass x 1.9 ns / 1024 SIMDs vs the live launch time"}
        kname, _ = _pmc(pmc_prefix + "_kernel", srchash.MSM_SOURCES)
        return {"bound": "hbm", "kernel": kname or "msm_accumulate_kernel<EcFx<BlsFqX>>", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": note, "algorithmic_bytes_per_launch": alg_bytes,
                "avg_launch_ms": round(acc_ms, 4), "valu_issue": valu,
                "note": "integer-ALU bound (381-bit Montgomery mixed adds), not HBM bound: see DESIGN.md"}

    if rank == 0:
        out = {
            "metric": "msm_g1_scalar_pairs_per_s", "value": world * n * args.steps / elapsed, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 (381-bit Fq as 14 x 29-bit limbs, Montgomery)",
            "data": "synthetic",
            "config": {"workload": f"standalone MSM, 2^{args.log_n} G1 x Fr pairs per GPU, BLS12-381 (BASELINE configs[1]); VARIABLE-BASE path: plain "
                                   "Pippenger over the registered bases, nothing precomputed per base (mzk_msm_set_precompute(0)) -- what "
                                   "`VariableBaseMSM::msm_bigint` (univariate_kzg/mod.rs:109-111) is; `fixed_base` = the same steps on the "
                                   "library's default path for a registered SRS",
                       "curve": "bls12-381", "pairs_per_gpu": n, "window_bits": c_bits, "windows": n_win,
                       "buckets_per_window": n_buckets, "sharding": "point-range" if world > 1 else "none",
                       "srs_gen_s": round(t_srs, 3),
                       "leg_order": ("fixed_base leg (table build + W + K steps) ran before these W + K steps: sustained clocks"
                                     if fb_raw is not None else "headline first (card out of idle: clock ramp inside the timed steps)")},
            "roofline": msm_roofline(acc_avg_ms, n_win, "msm_accumulate_plain"),
            "phases_ms": phases,
            "cpu_baseline": None, "fixed_base": None, "ntt": None, "plonk_round3": None, "batch_commit5": None, "prove": None,
            "prove_dropin": None, "prove_cpp_host": None, "prove_sharded": None, "prove_replicas": None, "prove_cpp_host_multi_gpu": None, "prove_ultra_bn254": None, "link_and_batch": None,
        }

    # (the watchdog of the multi-rank sections was started before the MSM legs: see `bail` above)

    # ---- secondary: the fixed-base leg measured above, before the headline -----
    fixed_base = None
    L.mzk_msm_set_precompute(1)
    if fb_raw is not None:
        pcb, pcl, pcy, pcm, fb_el, fb_res, fb_acc, fb_phases, (fc, fw, fm) = fb_raw
        precompute = {"window_bits": pcb, "levels": pcl, "table_bytes": pcy, "build_ms": round(pcm, 2),
                      "note": "fixed-base table table[w][i] = 2^(c*w) * P_i of the registered SRS, built once per SRS by pre_next_level_kernel "
                              "OUTSIDE the timed region of this leg"}
        if rank == 0:
            fixed_base = {"what": "the headline's steps with mzk_msm_set_precompute(1), the library default: the digits of all windows index rows of "
                                  "the table and share ONE bucket set.  Legitimate for KZG (the commit key never changes, srs.rs:77-93) and what "
                                  "`prove` below runs on, but a set-up ark-ec's VariableBaseMSM does not have: not `value`",
                          "value": world * n * args.steps / fb_el, "unit": "pairs/s", "ms_per_step": fb_el / args.steps * 1e3,
                          "window_bits": fc, "windows": fw, "buckets_per_window": fm, "phases_ms": fb_phases, "precompute": precompute,
                          "same_point_as_variable_base": bool(np.array_equal(mj.kzg.jacobian_to_affine(curve, np.asarray(fb_res).reshape(1, -1)),
                                                                             mj.kzg.jacobian_to_affine(curve, np.asarray(result).reshape(1, -1)))),
                          "roofline": msm_roofline(fb_acc, fw, "msm_accumulate")}
    if table_off_only:
        L.mzk_msm_set_precompute(0)

    # ---- secondary: batch_commit of 5 polynomials (round 1 / round 3 of a proof) in one fused call -----
    batch = None
    if rank == 0 and world == 1 and not args.no_batch:
        sets = [d_scalars] * 5
        for _ in range(2):
            mj.msm_bigint_batch(pp, sets, scalars_are_mont=True)
        torch.cuda.synchronize()
        times = []
        for _ in range(3):
            t1 = time.perf_counter()
            mj.msm_bigint_batch(pp, sets, scalars_are_mont=True)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t1)
        bt = sorted(times)[1]
        batch = {"what": "5 MSMs of 2^%d pairs in one mzk_msm_batch_dev call (shared bucket reduction, one sync); median of 3" % args.log_n,
                 "ms_per_batch": round(bt * 1e3, 3), "pairs_per_s": 5 * n / bt}

    # ---- secondary: NTT 2^22 forward + inverse on the Fr::GENERATOR coset (config C3) ---------------
    ntt = None
    if not args.no_ntt and rank == 0:
        nl = args.ntt_log_n
        N = 1 << nl
        x = torch.from_numpy(mj.params.random_fr_mont(curve, N, seed=22).view(np.int64)).to(dev)
        d = mj.Radix2EvaluationDomain(curve, nl).get_coset(curve.fr_generator)
        for _ in range(2):
            d.fft_in_place(x)
            d.ifft_in_place(x)
        torch.cuda.synchronize()
        L.mzk_profile_reset()
        L.mzk_profile_enable(1)
        reps = 10
        t1 = time.perf_counter()
        for _ in range(reps):
            d.fft_in_place(x)
            d.ifft_in_place(x)
        torch.cuda.synchronize()
        ntt_wall = (time.perf_counter() - t1) / reps
        L.mzk_profile_enable(0)
        pass_ms, pass_cnt = mlib.profile_get("ntt_pass")
        nt_ms, nt_cnt = mlib.profile_get("ntt_total")
        L.mzk_profile_reset()
        per_transform_ms = nt_ms / max(nt_cnt, 1)
        ntt_traffic, ntt_note = _pmc("ntt_pass_hbm_bytes_per_launch", srchash.NTT_SOURCES)
        ntt = {"log_n": nl, "fwd_plus_inv_ms": round(ntt_wall * 1e3, 4), "transform_ms": round(per_transform_ms, 4),
               "passes_per_transform": pass_cnt // max(nt_cnt, 1), "pass_ms": round(pass_ms / max(pass_cnt, 1), 4),
               "butterflies_per_s": (N // 2 * nl) / (per_transform_ms * 1e-3),
               "roofline": {"bound": "hbm", "kernel": "nttx_pass_kernel<BlsFrX>, per transform (all passes)",
                            "achieved": round(64.0 * N / (per_transform_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(64.0 * N / (per_transform_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                            "traffic": ntt_traffic, "traffic_source": ntt_note,
                            "traffic_is": "HBM bytes of ONE pass launch (a transform is %d)" % (pass_cnt // max(nt_cnt, 1))}}
        del x

    # ---- secondary: TurboPlonk round 3 on the device at 2^20 gates (config C4's heaviest round) -------
    plonk = None
    if not args.no_plonk and rank == 0:
        pl, pn = args.plonk_log_n, 1 << args.plonk_log_n
        pm = 8 * pn
        fixed = mj.params.random_fr_mont(curve, 18 * pn, seed=31).reshape(18, pn, 4)
        pk = mj.plonk.ProvingKeyDevice.register(curve, pn, list(fixed[:13]), list(fixed[13:]), [1, 2, 3, 4, 5])       # whole-domain key
        wit = mj.params.random_fr_mont(curve, 7 * (pn + 3), seed=32).reshape(7, pn + 3, 4)
        d_coeffs = torch.from_numpy(wit.view(np.int64)).to(dev)
        d_polys = torch.zeros((7, pm, 4), dtype=torch.int64, device=dev)
        d_out = torch.empty((pm, 4), dtype=torch.int64, device=dev)
        ch = mj.plonk.Challenges(0x1234567, 0x89abcde, 0xf012345)

        def round3():
            d_polys[:, :pn + 3] = d_coeffs            # fresh coefficients (the call overwrites them)
            mj.plonk.compute_quotient_polynomial_dev(pk, ch, d_polys, pn + 3, d_out)
        def median_ms(fn, reps=5):                     # (a release of gigabytes just before can stall one call by tens of ms: median, not mean)
            ts = []
            for _ in range(reps):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t1) * 1e3)
            return sorted(ts)[len(ts) // 2]
        round3()
        whole_ms = median_ms(round3)
        pk.release()
        # the default path of both hosts: W = 5 residue classes, class by class, the W + 3 top coefficients from the numerator's factors, then
        # the inverse Vandermonde (random polynomials here: what is timed is the work, the result is not a quotient)
        needed = mj.plonk.quotient_classes_needed(5, pn)
        t1 = time.perf_counter()
        pk = mj.plonk.ProvingKeyDevice.register(curve, pn, list(fixed[:13]), list(fixed[13:]), [1, 2, 3, 4, 5], classes=needed)
        t_pk = time.perf_counter() - t1
        d_rows = d_coeffs.contiguous()
        d_rem = torch.empty((len(needed), pn, 4), dtype=torch.int64, device=dev)

        def round3_classes():
            mj.plonk.compute_quotient_chunked_dev(pk, ch, d_rows, pn + 3, out_dev=d_rem)
            top = mj.plonk.compute_quotient_top_dev(pk, ch, d_rows, pn + 3)
            mj.plonk.combine_quotient_classes(curve, pn, d_rem, classes=needed, out_dev=d_out, top=top, n_top=8)
        round3_classes()
        r3_wall = median_ms(round3_classes) * 1e-3
        L.mzk_profile_reset()                          # one more pass with the library's event timers on (they cost time: not in round3_ms)
        L.mzk_profile_enable(1)
        round3_classes()
        torch.cuda.synchronize()
        L.mzk_profile_enable(0)
        qk_ms, qk_cnt = mlib.profile_get("plonk_quotient_kernel")
        L.mzk_profile_reset()
        plonk = {"what": "TurboPlonk round 3 without commitments on RANDOM polynomials (every selector non-zero, a public-input polynomial), the default "
                         "path: per residue class (5 of 8) 7 coset NTT(n) read in place + fused quotient kernel + inverse coset NTT(n), the top 8 "
                         "coefficients from the numerator's factors, then the inverse Vandermonde per coefficient; `whole_domain_ms` = 7 coset NTT(8n) + "
                         "kernel on 8n points + coset iNTT(8n), the round-1 path; selector/sigma evaluations resident per proving key.  (In a proof of "
                         "the bench circuit -- additions only, no public input -- the kernel skips the zero selectors and the round takes `prove."
                         "rounds_ms.r3_quotient`.)",
                 "log_n": pl, "classes": len(needed), "round3_ms": round(r3_wall * 1e3, 3), "whole_domain_ms": round(whole_ms, 3),
                 "quotient_kernel_ms_per_class": round(qk_ms / max(qk_cnt, 1), 3), "pk_register_s": round(t_pk, 3)}
        pk.release()
        del d_polys, d_out, d_coeffs, d_rem, d_rows, fixed

    # ---- secondary: PlonkKzgSnark::prove on the reference's bench circuit at 2^20 gates (config C4, bench.rs:29-74) -----
    prove = None
    if not args.no_plonk and rank == 0 and world == 1:
        pl, pn = args.plonk_log_n, 1 << args.plonk_log_n
        ck = pp if pp.length >= pn + 3 else mj.UnivariateProverParam.gen_srs_for_testing(curve, beta, pn + 2)
        t1 = time.perf_counter()
        cs = mj.snark.gen_circuit_for_bench(curve, pn, "TurboPlonk")
        torch.cuda.synchronize()
        t_circ = time.perf_counter() - t1
        t1 = time.perf_counter()
        prover = mj.snark.preprocess(ck, cs)
        prover.vk_commitments()
        t_pre = time.perf_counter() - t1
        rng = mj.rng.test_rng()
        for _ in range(3):                                  # warm-up: plans, precomputed SRS table, allocator steady state
            mj.snark.prove(rng, cs, prover)
        torch.cuda.synchronize()
        import gc
        gc.collect()                                        # the interpreter's first full collection costs ~40 ms and would otherwise
        gc.freeze()                                         # land in one of the timed proofs (tools/prove_time.py shows it)
        reps = args.prove_reps                              # plonk/benches/bench.rs:25: `let rep = 10`
        each = []
        t1 = time.perf_counter()
        for _ in range(reps):
            t2 = time.perf_counter()
            core, proof_bytes = mj.snark.prove(rng, cs, prover)
            each.append((time.perf_counter() - t2) * 1e3)
        torch.cuda.synchronize()
        prove_ms = (time.perf_counter() - t1) / reps * 1e3
        core, proof_bytes = mj.snark.prove(rng, cs, prover, profile=True)
        proof_bytes_first = mj.snark.prove(mj.rng.test_rng(), cs, prover)[1]      # the proof of a fresh `test_rng` stream: what the other legs must reproduce

        def timed_proofs(circuit, k):
            for _ in range(2):
                mj.snark.prove(rng, circuit, prover)
            torch.cuda.synchronize()
            ta = time.perf_counter()
            for _ in range(k):
                mj.snark.prove(rng, circuit, prover)
            torch.cuda.synchronize()
            return (time.perf_counter() - ta) / k * 1e3
        # round 1 from the masked COEFFICIENT forms, as the reference commits (univariate_kzg/mod.rs:90-116): a second handle of the same
        # circuit without the Lagrange-basis key
        coeff_prover = mj.snark.preprocess(ck, cs, lagrange=False)
        for _ in range(2):
            mj.snark.prove(rng, cs, coeff_prover)
        torch.cuda.synchronize()
        ta = time.perf_counter()
        for _ in range(reps):
            mj.snark.prove(rng, cs, coeff_prover)
        torch.cuda.synchronize()
        coeff_ms = (time.perf_counter() - ta) / reps * 1e3
        coeff_core, coeff_bytes = mj.snark.prove(mj.rng.test_rng(), cs, coeff_prover, profile=True)
        coeff_same = bool(coeff_bytes == mj.snark.prove(mj.rng.test_rng(), cs, prover)[1])
        coeff_prover.release()
        tl = time.perf_counter()
        tmp_key = ck.lagrange_key(pn)
        lagrange_key_s = time.perf_counter() - tl
        tmp_key.release()
        # `prove_ms` above IS the round-level C ABI (mzk_prover_create / round1 .. round5, include/mzk.h) driven from ctypes: since round 5
        # the product has one implementation of the rounds (csrc/prover.hip) and snark.prove is its thin client.  Here: launches per
        # proof, the same proof from a host-resident witness VECTOR, and what the instance holds in HBM with the library's grow-only
        # scratch started afresh (`hbm_bytes.library_scratch` = what PROOFS of this size need, not the earlier legs' maxima)
        torch.cuda.synchronize()
        mlib.check(L.mzk_workspace_release(), "mzk_workspace_release")
        npk = prover
        for _ in range(2):
            mj.snark.prove(rng, cs, npk)
        torch.cuda.synchronize()
        l0 = launch_count()
        ta = time.perf_counter()
        for _ in range(reps):
            mj.snark.prove(rng, cs, npk)
        torch.cuda.synchronize()
        abi_ms = (time.perf_counter() - ta) / reps * 1e3
        abi_launches = (launch_count() - l0) // reps
        abi_core, abi_bytes = mj.snark.prove(mj.rng.test_rng(), cs, npk, profile=True)
        abi_same = bool(abi_bytes == proof_bytes_first)
        npk.set_wire_variables(cs.wire_variables.cpu().numpy(), int(cs.witness.shape[0]))
        abi_vec = mj.snark.HostWitness(cs.witness.cpu().pin_memory(), cs.wire_variables)
        for _ in range(2):
            mj.snark.prove(rng, cs, npk, witness=abi_vec)
        torch.cuda.synchronize()
        ta = time.perf_counter()
        for _ in range(reps):
            mj.snark.prove(rng, cs, npk, witness=abi_vec)
        torch.cuda.synchronize()
        abi_vec_ms = (time.perf_counter() - ta) / reps * 1e3
        # HBM held for this proof system instance (the reference keeps one commit key: srs.rs:36-40)
        hbm = dict(npk.hbm_bytes())
        for name, key in (("commit_key", prover.ck), ("lagrange_key", prover.lagrange_ck)):
            if key is not None:
                a_, b_ = C.c_uint64(), C.c_uint64()
                mlib.check(L.mzk_srs_hbm_bytes(key.handle, C.byref(a_), C.byref(b_)), "mzk_srs_hbm_bytes")
                hbm[name + "_points"], hbm[name + "_fixed_base_table"] = a_.value, b_.value
        a_ = C.c_uint64()
        mlib.check(L.mzk_workspace_hbm_bytes(C.byref(a_)), "mzk_workspace_hbm_bytes")
        hbm["library_scratch"] = a_.value
        hbm["total"] = sum(hbm.values())
        del abi_vec
        # (i) the witness starts in page-locked HOST memory, as the reference holds it (constraint_system.rs:1225-1247 gathers it on the
        # host): every proof uploads its 5 x n x 32 B, wire k + 1 under the iNTT of wire k (prover.py _stage_round1)
        import dataclasses
        host_cs = dataclasses.replace(cs, wire_values=cs.wire_values.cpu().pin_memory())
        host_ms = timed_proofs(host_cs, reps)
        ref_bytes = mj.snark.prove(mj.rng.test_rng(), cs, prover)[1]
        host_bytes_same = bool(mj.snark.prove(mj.rng.test_rng(), host_cs, prover)[1] == ref_bytes)
        del host_cs
        # ... or only the witness VECTOR crosses (n_vars x 32 B) and the gather of compute_wire_polynomials runs on the device
        vec_cs = dataclasses.replace(cs, wire_values=mj.snark.HostWitness(cs.witness.cpu().pin_memory(), cs.wire_variables))
        vec_ms = timed_proofs(vec_cs, reps)
        host_bytes_same = host_bytes_same and bool(mj.snark.prove(mj.rng.test_rng(), vec_cs, prover)[1] == ref_bytes)
        n_vars = int(cs.witness.shape[0])
        del vec_cs
        # (ii) the same gates with a DENSE witness: the bench circuit's wires are 0, 1, 2.. / ones / zeros / zeros / 1, 2, 3..
        dense_cs = mj.snark.gen_circuit_for_bench(curve, pn, "TurboPlonk", dense_seed=77)
        prover.release()
        prover = mj.snark.preprocess(ck, dense_cs)                  # same selectors, another wire permutation
        prover.vk_commitments()
        dense_ms = timed_proofs(dense_cs, reps)
        dense_core, _ = mj.snark.prove(rng, dense_cs, prover, profile=True)
        del dense_cs
        prove = {"what": "PlonkKzgSnark::prove of one TurboPlonk proof on the reference's bench circuit (bench.rs:29-46: a = a + 1, "
                         "gates - 10 times): 7 iNTT(n), grand product, coset NTTs, quotient, 13 MSM, evaluations, linearisation, "
                         "openings, ChaCha test_rng blinders, Merlin transcript, compressed proof bytes; proving key resident; "
                         "an unsatisfied witness raises WrongQuotientPolyDegree -- from the quotient identity at zeta at the end of round 5 "
                         "(the quotient comes from W residue classes + the numerator's top coefficients, so its degree cannot be wrong: DESIGN.md 4.3)",
                 "log_n": pl, "prove_ms": round(prove_ms, 2), "reps": reps, "min_ms": round(min(each), 2), "median_ms": round(sorted(each)[len(each) // 2], 2),
                 "max_ms": round(max(each), 2),
                 "ns_per_gate": round(prove_ms * 1e6 / pn, 1), "rounds_ms": core.timings_ms, "proof_bytes": len(proof_bytes),
                 "round1_commit": "Lagrange basis: the wire commitments are MSMs of the wire VALUES (plus two blinders) over [L_i(beta)]g, the key "
                                  "derived from the SRS's own points by an inverse NTT over the group at preprocess (`lagrange_key_s`, once per SRS "
                                  "and domain; its fixed-base table is a second %.1f GB).  Same group elements, same proof bytes.  The bench circuit's "
                                  "values are 0, 1, 2, .. / all ones / zeros: small scalars whose high digits cost nothing -- a DENSE witness gains "
                                  "nothing (`dense_witness_ms`).  `coefficient_commit_ms` is the same proof with round 1 committed from the masked "
                                  "coefficient forms, as the reference does" % (13 * (pn + 3) * 112 / 1e9),
                 "lagrange_key_s": round(lagrange_key_s, 3),
                 "round_level_abi_ms": round(abi_ms, 2), "round_level_abi_rounds_ms": abi_core.timings_ms, "round_level_abi_same_proof_bytes": abi_same,
                 "round_level_abi_from_host_witness_vector_ms": round(abi_vec_ms, 2),
                 "kernel_launches_per_proof": abi_launches,
                 "round_level_abi_note": "the same proofs through mzk_prover_create / round1 .. round5 (include/mzk.h): the rounds of prover.rs:72-419 "
                                         "run inside the library, the caller (here ctypes + the Python transcript) keeps transcript, rng and Proof",
                 "hbm_bytes": hbm,
                 "coefficient_commit_ms": round(coeff_ms, 2), "coefficient_commit_rounds_ms": coeff_core.timings_ms, "coefficient_commit_same_proof_bytes": coeff_same,
                 "from_host_witness_ms": round(host_ms, 2), "from_host_witness_same_proof_bytes": host_bytes_same,
                 "from_host_witness_vector_ms": round(vec_ms, 2),
                 "from_host_witness_note": "`prove_ms` has the 5 x n wire values already in HBM.  from_host_witness_ms: they start in page-locked host "
                                           "memory (%.0f MB per proof over PCIe, wire k + 1 uploaded under the iNTT of wire k).  "
                                           "from_host_witness_vector_ms: only the witness vector does (%.0f MB); witness[wire_variable(i, j)] "
                                           "(constraint_system.rs:1225-1247) is gathered on the device over the resident index table"
                                           % (5 * pn * 32 / 1e6, n_vars * 32 / 1e6),
                 "dense_witness_ms": round(dense_ms, 2), "dense_witness_rounds_ms": dense_core.timings_ms,
                 "dense_witness_note": "same gates (selectors), random satisfying witness: all five wire polynomials dense and all wire values "
                                       "random field elements (the bench circuit's values are small numbers; in coefficient form two of its wire "
                                       "polynomials are zero and two sparse)",
                 "circuit_build_s": round(t_circ, 3), "preprocess_s": round(t_pre, 3),
                 "reference_published": "29591 ns/gate at 2^15 gates, 24 threads of a 5900X (bench.md:16); not comparable hardware"}
        prover.release()
        del cs
        if ck is not pp:
            ck.release()
        # HBM held by a prover of a DENSE witness (no Lagrange-basis key: snark.witness_is_small decides from a sample) and by the C5-shaped
        # proof (UltraPlonk / BN254, 2^22 gates) -- child processes, so that the grow-only scratch is what THOSE proofs need
        try:
            import subprocess
            hb = {}
            for name, extra in (("dense_witness_prover", ["--log-n", str(pl), "--dense"]), ("c5_shape_ultra_bn254_2p22", ["--ultra", "--log-n", "22"])):
                if name.startswith("c5") and pl < 20:
                    continue                                    # (toy sizes of the contract test: skip the 2^22 proof)
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "hbm_report.py")] + extra, capture_output=True, text=True, timeout=600)
                d_ = json.loads(r.stdout.strip().splitlines()[-1])
                hb[name] = {"hbm_total_bytes": d_["hbm_bytes"]["total"], "library_scratch": d_["hbm_bytes"]["library_scratch"], "prove_ms": d_["prove_ms"]}
            prove["hbm_other_provers"] = hb
        except Exception as e:                                  # noqa: BLE001  (secondary)
            prove["hbm_other_provers"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        # K proofs IN FLIGHT on this one card: K host threads, each with its own device context and prover handle (every handle runs on a
        # stream of its own, csrc/prover.hip) -- a child process, because the K contexts are configured before the library is loaded
        try:
            import subprocess
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prove_in_flight.py"), "--log-n", str(pl), "--in-flight", "1,2,3", "--reps", str(reps)],
                               capture_output=True, text=True, timeout=600)
            fl = json.loads(r.stdout.strip().splitlines()[-1])
            prove["in_flight"] = {"proofs_per_s_1_in_flight": fl["in_flight"]["1"]["proofs_per_s"], "proofs_per_s_2_in_flight": fl["in_flight"]["2"]["proofs_per_s"],
                                  "proofs_per_s_3_in_flight": fl["in_flight"]["3"]["proofs_per_s"], "in_flight_contexts_agree_on_proof": fl["contexts_agree_on_proof"],
                                  "what": "tools/prove_in_flight.py: K threads x (device context + mzk_prover handle on its own stream) on ONE card, round-level C ABI "
                                          "driven from ctypes; the host Horner tails, transcripts and launch gaps of one proof run under the kernels of the others"}
        except Exception as e:                                  # noqa: BLE001  (secondary)
            prove["in_flight"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}

    # ---- secondary: the same proof's NTTs and MSMs in SHIM-ONLY mode: host pointers through the two call-site swaps of
    #      INTEGRATION.md section 2, nothing else of the Rust prover changed (tools/dropin_time.py) --------------------------------
    dropin = None
    if not args.no_plonk and not args.no_dropin and rank == 0 and world == 1:
        import dropin_time
        try:
            modes = {}
            for mode in ("pageable", "pinned", "batch", "four_site"):
                modes[mode] = dropin_time.measure(mj, L, curve, args.plonk_log_n, mode, reps=2,
                                                  srs=pp if pp.length >= (1 << args.plonk_log_n) + 3 else None)
            dropin = {"what": "library time of ONE TurboPlonk proof in shim-only mode: 7 ifft(n) + 25 coset fft(8n) + 1 coset ifft(8n) through "
                              "host-pointer mzk_ntt, 13 commits through host-pointer mzk_msm; the quotient closure (prover.rs:605-659) stays on the "
                              "CPU in this mode and is NOT in the figure.  pageable: ordinary host memory, one call per polynomial; pinned: "
                              "buffers from mzk_host_alloc; batch: pinned + mzk_ntt_batch / mzk_msm_batch (upload k+1 | transform k | download k-1); "
                              "four_site: the grand product and the quotient swapped too (mzk_plonk_perm_product, mzk_plonk_quotient over a registered "
                              "proving key, host pointers): the 25 coset FFTs never reach the host and the CPU loops are gone",
                      "log_n": args.plonk_log_n, "pcie_gb_per_proof": {k: v["pcie_gb"] for k, v in modes.items()},
                      "ms": {k: v["ms"] for k, v in modes.items()}, "pcie_gb_per_s": {k: v["pcie_gb_per_s"] for k, v in modes.items()},
                      "round_level": prove["round_level_abi_ms"] if prove else None,
                      "round_level_note": "ms per proof when the caller swaps the BODIES of Prover::run_1st_round .. compute_opening_proofs for "
                                          "mzk_prover_round1 .. round5 (INTEGRATION.md section 2b): witness vector in, commitments and evaluations out",
                      "device_resident_prove_ms": prove["prove_ms"] if prove else None}
        except Exception as e:                              # noqa: BLE001  (secondary: the headline must still be printed)
            dropin = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}

    # ---- secondary, N > 1: N independent provers, one per rank, each proving the SAME circuit on its own GPU with no collective on the
    #      data path -- the reference's own parallelism is Rayon over whole polynomials (univariate_kzg/mod.rs:119-131), and whole proofs
    #      are what N GPUs run best: proofs/s, weak scaling
    prove_replicas = None
    if not args.no_plonk and world > 1 and not os.environ.get("MZK_BENCH_NO_REPLICAS"):
        try:
            native = mj.snark
            pn = 1 << args.plonk_log_n
            ckr = mj.UnivariateProverParam.gen_srs_for_testing(curve, beta, pn + 2)            # the same SRS, circuit and rng seed on every rank
            csr = mj.snark.gen_circuit_for_bench(curve, pn, "TurboPlonk")
            dense = os.environ.get("MZK_BENCH_REPLICAS_DENSE") == "1"
            npk = native.preprocess(ckr, csr, lagrange=not dense)
            rngr = mj.rng.test_rng()
            for _ in range(3):
                native.prove(rngr, csr, npk)
            torch.cuda.synchronize()
            dist.barrier()
            reps = args.prove_reps
            t1 = time.perf_counter()
            for _ in range(reps):
                native.prove(rngr, csr, npk)
            torch.cuda.synchronize()
            mine = time.perf_counter() - t1
            dist.barrier()
            tmax = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=cdev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            _, rb = native.prove(mj.rng.test_rng(), csr, npk)
            digest = torch.tensor([int.from_bytes(hashlib.sha256(rb).digest()[:7], "little")], dtype=torch.int64, device=cdev)
            lo, hi = digest.clone(), digest.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            prove_replicas = {"what": "N independent TurboPlonk provers of the 2^%d-gate bench circuit, one per rank, through the round-level C ABI "
                                      "(mzk_prover_*); no collective inside the timed proofs; weak scaling: proofs/s = N * reps / max-over-ranks time"
                                      % args.plonk_log_n,
                              "log_n": args.plonk_log_n, "reps_per_rank": reps, "proofs_per_s": round(world * reps / float(tmax.item()), 2),
                              "ms_per_proof_rank0": round(mine / reps * 1e3, 2), "ranks_agree_on_proof": bool(lo.item() == hi.item()),
                              "proof_sha16": hashlib.sha256(rb).hexdigest()[:16], "proof_bytes": len(rb)}
            npk.release()
            ckr.release()
            del csr
        except Exception as e:                                             # noqa: BLE001  (secondary: the headline must still be printed)
            prove_replicas = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}

    # ---- secondary, N > 1: the same proof sharded over the ranks
    prove_sharded = None
    if not args.no_plonk and world > 1 and not os.environ.get("MZK_BENCH_NO_SHARDED_PROVE"):
        def sharded_prove(crv, log_gates, plonk_type):
            pn = 1 << log_gates
            ck = mj.UnivariateProverParam.gen_srs_for_testing(crv, beta, pn + 2)          # the same SRS on every rank
            cs = mj.snark.gen_circuit_for_bench(crv, pn, plonk_type)
            chunked = True                                                               # 8(e).3: the needed classes (6 / 7 of 8) split over the ranks
            # this process is ONE RANK of the library's own rounds (mzk_comm over torch.distributed: sharding.TorchComm) and keeps only its
            # point range of the SRS (1 / N of the fixed-base table): commitments by point range, the quotient by residue class with one
            # exchange, rounds 4-5 by coefficient range (csrc/prover.hip)
            prover = mj.snark.preprocess(ck, cs, comm=mj.sharding.TorchComm(device=coll_dev))
            prover.vk_commitments()
            rng = mj.rng.test_rng()
            for _ in range(3):
                mj.snark.prove(rng, cs, prover)
            torch.cuda.synchronize()
            dist.barrier()
            reps = 5
            t1 = time.perf_counter()
            for _ in range(reps):
                core, proof_bytes = mj.snark.prove(rng, cs, prover)
            torch.cuda.synchronize()
            dist.barrier()
            tmax = torch.tensor([(time.perf_counter() - t1) / reps * 1e3], dtype=torch.float64, device=cdev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            # every rank must hold the SAME proof: a digest of all its bytes (round 1 to the openings), min == max over the ranks
            digest = torch.tensor([int.from_bytes(hashlib.sha256(proof_bytes).digest()[:7], "little")], dtype=torch.int64, device=cdev)
            lo, hi = digest.clone(), digest.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            core, _ = mj.snark.prove(rng, cs, prover, profile=True)
            _, fresh = mj.snark.prove(mj.rng.test_rng(), cs, prover)        # from a fresh test_rng: the bytes `prove_replicas` reports (same SRS, circuit, seed)
            res = {"plonk_type": plonk_type, "curve": crv.name, "log_n": cs.n.bit_length() - 1, "chunked_quotient": chunked,
                   "prove_ms": round(float(tmax.item()), 2), "ranks_agree_on_proof": bool(lo.item() == hi.item()),
                   "proof_bytes": len(proof_bytes), "proof_sha16": hashlib.sha256(fresh).hexdigest()[:16], "rounds_ms_rank0": core.timings_ms}
            prover.release()
            ck.release()
            return res

        prove_sharded = {"what": "PlonkKzgSnark::prove on the bench circuit, strong scaling: commitments sharded by point range over the ranks "
                                 "(all-gather of Jacobian partials + local EC sum, 8(e).1), quotient domain split into residue classes with one "
                                 "exchange (8(e).3), rounds 4-5 by coefficient range -- the library's own rounds on every rank, mzk_comm over torch.distributed "
                                 "(sharding.TorchComm); compare with `prove` of the 1-GPU line; "
                                 "profiles/r04_scale_model.json (tools/scale_model.py) holds the model this is to be checked against",
                         }
        try:
            prove_sharded["turbo_bls12_381"] = sharded_prove(curve, args.plonk_log_n, "TurboPlonk")
            if prove_replicas and "proof_sha16" in prove_replicas:            # the sharded proof IS the single-GPU proof, byte for byte
                prove_sharded["same_proof_bytes_as_single_gpu_replicas"] = prove_sharded["turbo_bls12_381"]["proof_sha16"] == prove_replicas["proof_sha16"]
            if args.ultra_sharded_log_n:                               # config C5: UltraPlonk over BN254, 2^22 constraints
                prove_sharded["ultra_bn254"] = sharded_prove(mj.params.BN254, args.ultra_sharded_log_n, "UltraPlonk")
        except Exception as e:                                         # noqa: BLE001  (secondary: the headline must still be printed)
            prove_sharded["error"] = "%s: %s" % (type(e).__name__, str(e)[:300])

    # ---- secondary, N > 1: the same proof from ONE process driving all N devices -- the compiled host, one host thread per device
    #      context of libmi355zk (host/mzk_prover.hpp ShardedProver), no torch.distributed, no Python.  The ranks of this run keep
    #      their GPUs but sit in the barrier below while rank 0's child process uses them.
    prove_cpp_multi = None
    if not args.no_plonk and world > 1 and not os.environ.get("MZK_BENCH_NO_CPP_MULTI"):
        torch.cuda.synchronize()
        dist.barrier()
        if rank == 0:
            import subprocess
            binp = os.path.join(ROOT, "mpc-jellyfish_amd", "mzk_prove")
            env = dict(os.environ)
            if os.environ.get("MZK_BENCH_SINGLE_DEVICE"):                         # rehearsal on a one-GPU box: N device contexts on the one card
                env["MZK_VIRTUAL_DEVICES"] = str(world)
            prove_cpp_multi = {"what": "PlonkKzgSnark::prove from ONE process driving all N devices: mzk_prove --gpus N (C++ host, one host thread per "
                                       "device, commitments sharded by point range and summed on the host, the quotient by residue class with one "
                                       "device-to-device exchange, rounds 4-5 by coefficient range); strong scaling against `gpus_1`"}
            for name, argv in (("gpus_%d" % world, ["0", "turbo", str(1 << args.plonk_log_n), "10", "--gpus", str(world), "--check-agree"]),
                               ("gpus_1", ["0", "turbo", str(1 << args.plonk_log_n), "10"])):
                try:
                    r = subprocess.run([binp] + argv, capture_output=True, text=True, timeout=240, env=env)
                    d = json.loads(r.stdout.strip().splitlines()[-1])
                    prove_cpp_multi[name] = {"prove_ms": d["prove_ms"], "rounds_ms": d["rounds_ms"], "preprocess_s": d["preprocess_s"],
                                             "proof_sha16": hashlib.sha256(d["proof_hex"].encode()).hexdigest()[:16]}
                except Exception as e:                                            # noqa: BLE001
                    prove_cpp_multi[name] = {"error": repr(e)[:200]}
            a, b = prove_cpp_multi.get("gpus_%d" % world, {}), prove_cpp_multi.get("gpus_1", {})
            if "prove_ms" in a and "prove_ms" in b:
                prove_cpp_multi["same_proof_bytes"] = a["proof_sha16"] == b["proof_sha16"]
                prove_cpp_multi["speedup"] = round(b["prove_ms"] / a["prove_ms"], 2)
        dist.barrier()

    # ---- secondary: UltraPlonk (Plookup) on BN254, the shape of config C5 at --ultra-log-n gates, one GPU ---------------
    ultra = None
    if not args.no_plonk and rank == 0 and world == 1 and args.ultra_log_n:
        ul, un = args.ultra_log_n, 1 << args.ultra_log_n
        bn = mj.params.BN254
        ck2 = mj.UnivariateProverParam.gen_srs_for_testing(bn, beta, un + 2)
        cs = mj.snark.gen_circuit_for_bench(bn, un, "UltraPlonk")
        t1 = time.perf_counter()
        prover = mj.snark.preprocess(ck2, cs)
        prover.vk_commitments()
        t_pre = time.perf_counter() - t1
        rng = mj.rng.test_rng()
        for _ in range(3):
            mj.snark.prove(rng, cs, prover)
        torch.cuda.synchronize()
        reps = args.prove_reps
        t1 = time.perf_counter()
        for _ in range(reps):
            core, proof_bytes = mj.snark.prove(rng, cs, prover)
        torch.cuda.synchronize()
        prove_ms = (time.perf_counter() - t1) / reps * 1e3
        core, proof_bytes = mj.snark.prove(rng, cs, prover, profile=True)
        ultra = {"what": "PlonkKzgSnark::prove, UltraPlonk (Plookup, range_bit_len 8) bench circuit over BN254 on ONE GPU",
                 "log_n": ul, "prove_ms": round(prove_ms, 2), "reps": reps, "ns_per_gate": round(prove_ms * 1e6 / un, 1), "rounds_ms": core.timings_ms,
                 "proof_bytes": len(proof_bytes), "preprocess_s": round(t_pre, 3)}
        prover.release()
        ck2.release()
        del cs

    # ---- optional (outside SURVEY.md section 8): proof linking and an aggregated proof --------------------------------------------
    link_batch = None
    if args.link_batch and not args.no_plonk and rank == 0 and world == 1:
        ln = args.plonk_log_n
        g1, g2 = (1 << ln) - 576, (1 << ln) - 76                          # two circuits of one domain size, sharing wire-0 rows
        size = min(256, max(1, (1 << ln) // 8))
        layout = mj.linking.GroupLayout(max(ln - 2, 1), 5, size)
        cs_a, cs_b = (mj.snark.gen_circuit_for_bench(curve, g, "TurboPlonk") for g in (g1, g2))
        if cs_a.n == cs_b.n == (1 << ln):
            rngl = mj.rng.test_rng()
            ckl = mj.UnivariateProverParam.gen_srs_for_testing(curve, beta, cs_a.n + 2)
            pa, pb = mj.snark.preprocess(ckl, cs_a), mj.snark.preprocess(ckl, cs_b)
            _, _, ha = mj.snark.prove_with_link_hint(rngl, cs_a, pa)
            _, _, hb = mj.snark.prove_with_link_hint(rngl, cs_b, pb)
            mj.linking.link_proofs(ha, hb, layout, ckl)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                lp = mj.linking.link_proofs(ha, hb, layout, ckl)
            torch.cuda.synchronize()
            link_ms = (time.perf_counter() - t1) / 3 * 1e3
            for _ in range(2):
                mj.snark.batch_prove(rngl, [cs_a, cs_b], [pa, pb])
            times = []
            for _ in range(3):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                _, blob = mj.snark.batch_prove(rngl, [cs_a, cs_b], [pa, pb])
                torch.cuda.synchronize()
                times.append((time.perf_counter() - t1) * 1e3)
            link_batch = {"link_proofs_ms": round(link_ms, 2), "link_proof_bytes": len(lp.serialize_compressed()),
                          "batch_prove_2_instances_ms": round(sorted(times)[1], 2), "batch_proof_bytes": len(blob)}
            pa.release()
            pb.release()
            ckl.release()
        del cs_a, cs_b

    # ---- secondary: the same proofs driven by the C++ host layer (mpc-jellyfish_amd/host/, g++, C ABI only; no Python in the loop) ----
    prove_cpp = None
    if not args.no_plonk and rank == 0 and world == 1:
        import subprocess
        binp = os.path.join(ROOT, "mpc-jellyfish_amd", "mzk_prove")
        prove_cpp = {}
        # ... and at the reference's own bench size (NUM_GATES_LARGE = 32768, plonk/benches/bench.rs:26), whose published CPU figures
        # are 29 591 (TurboPlonk, BLS12-381) and 33 701 (UltraPlonk, BN254) ns per constraint on 24 threads of a 5900X (bench.md)
        for name, argv in (("turbo_bls12_381", ["0", "turbo", str(1 << args.plonk_log_n), "10"]),
                           ("turbo_bls12_381_coefficient_commit", ["0", "turbo", str(1 << args.plonk_log_n), "10", "--no-lagrange"]),
                           ("turbo_bls12_381_host_witness", ["0", "turbo", str(1 << args.plonk_log_n), "10", "--host-witness"]),
                           ("turbo_bls12_381_host_witness_vector", ["0", "turbo", str(1 << args.plonk_log_n), "10", "--host-witness-vars"]),
                           ("ultra_bn254", ["1", "ultra", str(1 << (args.ultra_log_n or args.plonk_log_n)), "10"]),
                           # (100 proofs each: one proof in a few dozen takes milliseconds longer on the host, and a mean over 20 moved by 0.5 ms with it;
                           # the JSON of mzk_prove carries the median, minimum and maximum too)
                           ("turbo_bls12_381_1024_gates", ["0", "turbo", "1024", "100"]),
                           ("turbo_bls12_381_32768_gates", ["0", "turbo", "32768", "100"]),
                           ("ultra_bn254_32768_gates", ["1", "ultra", "32768", "100"])):
            try:
                r = subprocess.run([binp] + argv, capture_output=True, text=True, timeout=600)
                d = json.loads(r.stdout.strip().splitlines()[-1])
                proof_hex = d.pop("proof_hex", None)
                d.pop("vk_hex", None)
                d["ns_per_gate"] = round(d["prove_ms"] * 1e6 / d["num_gates"], 1)
                # the printed proof is the first one after preprocess on a fresh `test_rng`: the Python mirror must emit the same bytes
                crv = mj.params.CURVES[int(argv[0])]
                csx = mj.snark.gen_circuit_for_bench(crv, int(argv[2]), "UltraPlonk" if argv[1] == "ultra" else "TurboPlonk")
                rngx = mj.rng.test_rng()
                ckx = mj.UnivariateProverParam.gen_srs_for_testing(crv, mj.rng.fr_rand(crv, rngx), csx.n + 2)
                pkx = mj.snark.preprocess(ckx, csx)
                d["proof_matches_python_mirror"] = bool(mj.snark.prove(rngx, csx, pkx)[1].hex() == proof_hex)
                pkx.release()
                ckx.release()
                del csx
                prove_cpp[name] = d
            except Exception as e:                      # noqa: BLE001  (the binary is optional for the headline)
                prove_cpp[name] = {"error": repr(e)[:200]}

    # ---- CPU baseline: the C oracle ("port" of the ark-ec / ark-poly algorithms) on this host, rank 0 only -------
    cpu = None
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import cref
        import cref_prover
        cpu_model, threads_host, threads = host_cpu()                 # all hardware threads this process may use, as Rayon would
        bases = pp.powers_of_g()
        canon = cref.fr_convert(0, scalars, False)
        t1 = time.perf_counter()
        want = cref.msm(0, bases, canon, threads=threads)
        cpu_s = time.perf_counter() - t1
        same = np.array_equal(cref.jac_to_affine(0, want)[0], cref.jac_to_affine(0, result)[0])
        t1 = time.perf_counter()
        want1 = cref.msm(0, bases, canon, threads=1)                  # ... and on ONE thread
        cpu_s1 = time.perf_counter() - t1
        same = same and np.array_equal(cref.jac_to_affine(0, want1)[0], cref.jac_to_affine(0, result)[0])
        del bases, canon
        # the same host's figure for config C3 (NTT 2^22), beside the MSM one: the WHOLE vector is compared
        xs = mj.params.random_fr_mont(curve, 1 << 22, seed=11)
        t1 = time.perf_counter()
        ev_cpu = cref.ntt(0, xs, 22, False, None, threads=threads)
        ntt_cpu_s = time.perf_counter() - t1
        ntt_same = bool(np.array_equal(ev_cpu, mj.Radix2EvaluationDomain(curve, 22).fft(xs)))
        t1 = time.perf_counter()
        ev_cpu = cref.ntt(0, xs, 22, False, None, threads=1)
        ntt_cpu_s1 = time.perf_counter() - t1
        del xs, ev_cpu
        hostv = lambda t: t.cpu().numpy().view(np.uint64)

        def cpu_vs_device(lg, cpu_threads, gpu_reps):
            """One TurboPlonk proof of the 2^lg-gate bench circuit by the C restatement (oracle/cref_prover.py) and by the device
            prover, on the same circuit, SRS, blinders and transcript challenges; compared commitment by commitment."""
            csm = mj.snark.gen_circuit_for_bench(curve, 1 << lg, "TurboPlonk")
            rngm = mj.rng.test_rng()
            ckm = mj.UnivariateProverParam.gen_srs_for_testing(curve, mj.rng.fr_rand(curve, rngm), csm.n + 2)
            pkm = mj.snark.preprocess(ckm, csm)
            blm = mj.snark.draw_blinders(curve, rngm, 5, False)
            for _ in range(2):
                pkm.prove(csm.wire_values, [], mj.prover.TranscriptChallenges(pkm, []), blm)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(gpu_reps):
                srcm = mj.prover.TranscriptChallenges(pkm, [])
                corem = pkm.prove(csm.wire_values, [], srcm, blm)
            torch.cuda.synchronize()
            gpu_ms = (time.perf_counter() - t1) / gpu_reps * 1e3
            srs_xy = ckm.powers_of_g()
            sel, sig, wv, pv = hostv(csm.selector_values), hostv(csm.sigma_values), hostv(csm.wire_values), hostv(csm.pub_input_values)
            pkm.release()                                                 # the CPU proof needs the host's memory and cores, not the device
            cm_ = cref_prover.prove_turbo(0, curve.r, curve.fr_generator, lg, sel, sig, csm.k, wv, pv, {"wires": blm.wires, "z": blm.z, "quot": blm.quot},
                                          dict(srcm.challenges), srs_xy, threads=cpu_threads)
            ok = bool(np.array_equal(corem.opening_proof.xy, cm_["opening"]) and np.array_equal(corem.shifted_opening_proof.xy, cm_["shifted"])
                      and all(np.array_equal(a.xy, b) for a, b in zip(corem.split_quot_poly_comms, cm_["split_comms"]))
                      and corem.wires_evals == cm_["wires_evals"] and corem.perm_next_eval == cm_["perm_next_eval"])
            ckm.release()
            return {"log_n": lg, "ms": round(cm_["prove_seconds"] * 1e3, 1), "cores": cpu_threads, "gpu_ms": round(gpu_ms, 2),
                    "gpu_over_cpu": round(cm_["prove_seconds"] * 1e3 / gpu_ms, 1), "matches_gpu": ok,
                    "cpu_spent_s": cm_.get("spent_seconds"),
                    "sample": "ONE TurboPlonk proof of the 2^%d-gate bench circuit over BLS12-381 by the C restatement (oracle/cref_prover.py: ark-poly "
                              "style FFTs, ark-ec style Pippenger with its window rule, the reference's serial grand product and per-point quotient "
                              "closure; `preprocess` work excluded) on %d threads, vs the device prover (round-level C ABI from ctypes, proving key resident) on the "
                              "same circuit, SRS, blinders and transcript; restatement of the ark-* algorithms, not the Rust binary" % (lg, cpu_threads)}

        c1 = cpu_vs_device(10, 1, 3)                                      # config C1 (BASELINE.json configs[0]): 2^10 gates, ONE CPU thread
        one_thread = cpu_vs_device(14, 1, 3) if args.cpu_prove_log_n >= 14 else None      # a bounded one-thread sample (2^20 gates would take ~15 min)
        big = cpu_vs_device(args.cpu_prove_log_n, threads, 3) if args.cpu_prove_log_n else None
        cpu = {"value": n / cpu_s, "unit": "pairs/s", "cores": threads, "kind": "port",
               "sample": f"one full 2^{args.log_n}-pair MSM (same bases and scalars as the GPU step), oracle/cpu_ref.c "
                         f"Pippenger with the ark-ec window rule, {threads} threads; restatement of ark-ec's VariableBaseMSM, not the Rust binary "
                         "(like for like with `value`: neither has a fixed-base table)",
               "seconds": round(cpu_s, 3), "matches_gpu": bool(same),
               "gpu_over_cpu": round(out["value"] / (n / cpu_s), 1),
               # flat scalars (the driver's record keeps only those): the host, and every CPU figure at ONE thread and at ALL hardware threads
               "cpu_model": cpu_model, "cpu_threads_host": threads_host, "cpu_threads_used": threads,
               "value_1_thread": n / cpu_s1, "cpu_msm_2p20_ms": round(cpu_s * 1e3, 1), "cpu_msm_2p20_1_thread_ms": round(cpu_s1 * 1e3, 1),
               "cpu_ntt_2p22_ms": round(ntt_cpu_s * 1e3, 1), "cpu_ntt_2p22_1_thread_ms": round(ntt_cpu_s1 * 1e3, 1), "cpu_ntt_2p22_matches_gpu_whole_vector": ntt_same,
               "cpu_prove_c1_ms": c1["ms"], "cpu_prove_c1_threads": 1, "cpu_prove_c1_matches_gpu": c1["matches_gpu"], "gpu_prove_c1_ms": c1["gpu_ms"],
               "cpu_prove_2p14_1_thread_ms": one_thread["ms"] if one_thread else None,
               "cpu_prove_2p14_matches_gpu": one_thread["matches_gpu"] if one_thread else None,
               ("cpu_prove_2p%d_ms" % args.cpu_prove_log_n): big["ms"] if big else None,
               ("cpu_prove_2p%d_threads" % args.cpu_prove_log_n): threads if big else None,
               ("cpu_prove_2p%d_matches_gpu" % args.cpu_prove_log_n): big["matches_gpu"] if big else None,
               "published_anchor": "29591 ns/constraint, TurboPlonk BLS12-381 at 2^15 gates, 24 threads of a 5900X (bench.md:16); other hardware",
               "parity_note": "matches_gpu compares with this repo's own C restatement: parity with the Rust code is unpinned (DESIGN.md section 2)",
               "prove_c1": c1, "prove_2p14_1_thread": one_thread,
               ("prove_2p%d" % args.cpu_prove_log_n): big,
               "ntt_2^22": {"ms": round(ntt_cpu_s * 1e3, 1), "ms_1_thread": round(ntt_cpu_s1 * 1e3, 1), "cores": threads, "matches_gpu": ntt_same,
                            "sample": "one forward 2^22-point NTT, oracle/cpu_ref.c in-order radix-2 (ark-poly's algorithm restated); all "
                                      "2^22 outputs compared with the device's"}}

    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        out.update({"cpu_baseline": cpu, "fixed_base": fixed_base, "ntt": ntt, "plonk_round3": plonk, "batch_commit5": batch, "prove": prove,
                    "prove_dropin": dropin, "prove_cpp_host": prove_cpp, "prove_sharded": prove_sharded, "prove_replicas": prove_replicas, "prove_cpp_host_multi_gpu": prove_cpp_multi, "prove_ultra_bn254": ultra,
                    "link_and_batch": link_batch})
        # Flat scalars: the driver's record keeps the top level's standard keys and the scalars directly under `roofline` / `cpu_baseline`;
        # the nested objects above are for readers of the raw line.  `prove_dense_witness_ms` is the GENERAL-CASE proof time (a real
        # circuit's witness is dense field elements); `prove_ms` rides on the bench circuit's small witness values through the Lagrange basis.
        flat = {}
        if prove:
            flat.update({"prove_ms": prove["prove_ms"], "prove_dense_witness_ms": prove["dense_witness_ms"],
                         "prove_coefficient_commit_ms": prove["coefficient_commit_ms"], "prove_round_level_abi_ms": prove["round_level_abi_ms"],
                         "prove_from_host_witness_vector_ms": prove["from_host_witness_vector_ms"],
                         "kernel_launches_per_proof": prove["kernel_launches_per_proof"], "hbm_total_bytes": prove["hbm_bytes"]["total"],
                         "prove_log_n": prove["log_n"]})
            for name_, v_ in (prove.get("hbm_other_provers") or {}).items():
                if isinstance(v_, dict) and "hbm_total_bytes" in v_:
                    flat["hbm_total_bytes_" + name_] = v_["hbm_total_bytes"]
            for k_, v_ in (prove.get("in_flight") or {}).items():
                if isinstance(v_, (int, float, bool)):
                    flat["prove_" + k_] = v_
        if ntt:
            flat.update({"ntt_2p%d_ms" % ntt["log_n"]: ntt["transform_ms"], "ntt_hbm_frac": ntt["roofline"]["frac"]})
        if fixed_base:
            flat["fixed_base_ms_per_step"] = round(fixed_base["ms_per_step"], 4)
        if batch:
            flat["batch_commit5_ms"] = batch["ms_per_batch"]
        if ultra:
            flat.update({"prove_ultra_bn254_ms": ultra["prove_ms"], "prove_ultra_bn254_log_n": ultra["log_n"]})
        if prove_cpp:
            for name, key in (("turbo_bls12_381", "prove_cpp_host_ms"), ("turbo_bls12_381_32768_gates", "prove_cpp_host_2p15_ms"),
                              ("turbo_bls12_381_1024_gates", "prove_cpp_host_2p10_ms"), ("ultra_bn254_32768_gates", "prove_cpp_host_ultra_2p15_ms")):
                if "prove_ms" in prove_cpp.get(name, {}):
                    flat[key] = prove_cpp[name]["prove_ms"]                       # mean over the repetitions, like every figure since round 1
                if "prove_median_ms" in prove_cpp.get(name, {}):
                    flat[key.replace("_ms", "_median_ms")] = prove_cpp[name]["prove_median_ms"]
        if prove_replicas:
            flat.update({"prove_replicas_proofs_per_s": prove_replicas.get("proofs_per_s"), "prove_replicas_ranks_agree": prove_replicas.get("ranks_agree_on_proof")})
        if prove_sharded and isinstance(prove_sharded.get("turbo_bls12_381"), dict):
            flat.update({"prove_sharded_ms": prove_sharded["turbo_bls12_381"].get("prove_ms"),
                         "prove_sharded_ranks_agree": prove_sharded["turbo_bls12_381"].get("ranks_agree_on_proof"),
                         "prove_sharded_same_bytes_as_single_gpu": prove_sharded.get("same_proof_bytes_as_single_gpu_replicas")})
        if prove_cpp_multi and "speedup" in prove_cpp_multi:
            flat.update({"prove_cpp_host_multi_gpu_ms": prove_cpp_multi["gpus_%d" % world]["prove_ms"], "prove_cpp_host_multi_gpu_speedup": prove_cpp_multi["speedup"],
                         "prove_cpp_host_multi_gpu_same_bytes": prove_cpp_multi.get("same_proof_bytes")})
        vi = out["roofline"].get("valu_issue") or {}
        out["roofline"].update({"valu_issue_util": vi.get("util"), "valu_issue_bound_ms": vi.get("issue_bound_ms"),
                                "valu_wave_insts_per_launch": vi.get("wave_insts_per_launch"), "launches_per_msm": phases["kernel_launches"],
                                "step_sort_ms": phases["sort"], "step_reduce_ms": phases["reduce"], "step_split_combine_ms": phases["split_combine"]})
        out["roofline"].update(flat)
        if cpu:
            cpu.update({"gpu_" + k_: v_ for k_, v_ in flat.items()})
        out.update(flat)
        emit()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
