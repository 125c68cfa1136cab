#!/usr/bin/env python
"""bench.py - env-steps/s of the fused Avoiding step() on MI355X (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--envs 4096] [--no-cpu-baseline]

A "step" is one pass of the hot path over one batch: for every one of the 4096 environments per GPU the
device-side random policy writes the action (Philox, seed 42, counter = global env index x t), d3il_step runs
the 35 fused physics sub-steps, and finished environments are auto-reset (d3il_auto_reset), everything enqueued on one HIP stream with the state resident in HBM.  N > 1: one process
per GPU (torch.distributed / RCCL), env shards are independent (weak scaling, 4096 envs per GPU), the only
collective is the final int64 count all-reduce, outside the per-step path but inside the timed region.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6   # vector FP64, 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz
# algorithmic HBM bytes per env step: state read + written once (42 f64 + flags + step
# counter = 344 B each way), action 56 B read, obs 8 B + done/success/mode 4 B written
ALG_BYTES_PER_ENV_STEP = 2 * (42 * 8 + 4 + 4) + 56 + 8 + 4


def cpu_baseline(blob, init_qpos, budget_s=12.0):
    """Oracle (scalar C port of the reference path) timed on one host core on a bounded sample of the same
    workload: one environment, random policy, as many env steps as fit in ~budget_s."""
    import numpy as np
    from oracle.oracle import Oracle
    o = Oracle(blob)
    o.env_start(init_qpos)
    o.env_reset()
    s, _ = o.env_state()
    des = s[25:28].copy()
    rng = np.random.default_rng(42)
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        des[:2] += rng.uniform(-0.01, 0.01, 2)
        _, done, _, _ = o.env_step(np.array([des[0], des[1], des[2], 0, 1, 0, 0]))
        n += 1
        if done:
            o.env_reset()
            s, _ = o.env_state()
            des = s[25:28].copy()
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": "1 env, random policy, %d env steps (35 sub-steps each) in %.1f s on one host core of %d; "
                      "scalar C oracle (oracle/d3il_oracle.c); the Python reference is bounded above by 146 "
                      "env-steps/s/core (BASELINE.md section 2)" % (n, dt, os.cpu_count() or 0)}


def _local_device() -> int:
    """GPU of this rank: LOCAL_RANK (one process per GPU).  D3IL_BENCH_FORCE_DEVICE pins every rank to one GPU - only for
    exercising the multi-process code path on a single-GPU box (with D3IL_DIST_BACKEND=gloo; RCCL needs distinct GPUs)."""
    if os.environ.get("D3IL_BENCH_FORCE_DEVICE") is not None:
        return int(os.environ["D3IL_BENCH_FORCE_DEVICE"])
    return int(os.environ.get("LOCAL_RANK", "0"))


# Pushing: cube state adds 26 f64 (+ 2 flag/counter words as above); action 56 B; obs 32 B; done/success/mode 4 B; info 16 B
PUSH_ALG_BYTES_PER_ENV_STEP = 2 * (68 * 8 + 4 + 4) + 56 + 32 + 4 + 16


def cpu_baseline_pushing(blob, init_qpos, contexts, budget_s=12.0):
    """Scalar C oracle on one host core, one environment, same stand-in policy (run on the CPU), bounded sample."""
    import numpy as np
    import torch
    from d3il_amd.agents import RandomResidualMLPPolicy
    from oracle.oracle import Oracle
    o = Oracle(blob)
    o.env_start(init_qpos)
    pol = RandomResidualMLPPolicy(device="cpu")
    n, ep = 0, 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        obs = o.push_reset(contexts[ep % len(contexts)])
        s, _ = o.push_state()
        des, z = s[25:27].copy(), s[27]
        for t in range(400):
            x = torch.as_tensor(np.concatenate([des, obs.astype(np.float64)])[None], dtype=torch.float64)
            des = des + pol.predict_batch(x)[0].numpy().astype(np.float64)
            obs, _, done, _ = o.push_step(np.array([des[0], des[1], z, 0, 1, 0, 0]))
            n += 1
            if done or time.perf_counter() - t0 >= budget_s:
                break
        ep += 1
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": "1 env, ResidualMLP stand-in policy on the CPU, %d env steps (35 sub-steps each) in %.1f s on one host core of %d; "
                      "scalar C oracle (oracle/d3il_oracle.c)" % (n, dt, os.cpu_count() or 0)}


def cpu_baseline_sorting(blob, init_qpos, contexts, budget_s=15.0):
    """Scalar C oracle on one host core, one environment, same stand-in policy (run on the CPU), bounded sample."""
    import numpy as np
    import torch
    from d3il_amd.agents import RandomResidualMLPPolicy
    from oracle.oracle import Oracle
    o = Oracle(blob)
    o.env_start(init_qpos)
    pol = RandomResidualMLPPolicy(input_dim=16, device="cpu")
    n, ep = 0, 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        obs = o.sort_reset(contexts[ep % len(contexts)].reshape(-1, 7))
        des, z = obs[:2].astype(np.float64), float(o.body(blob.tcp_body)[0][2])
        for t in range(500):
            x = torch.as_tensor(np.concatenate([des, obs.astype(np.float64)])[None], dtype=torch.float64)
            des = des + pol.predict_batch(x)[0].numpy().astype(np.float64)
            obs, done, _ = o.sort_step(np.array([des[0], des[1], z, 0, 1, 0, 0]))
            n += 1
            if done or time.perf_counter() - t0 >= budget_s:
                break
        ep += 1
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": "1 env, ResidualMLP stand-in policy on the CPU, %d env steps (35 sub-steps each) in %.1f s on one host core of %d; "
                      "scalar C oracle (oracle/d3il_oracle.c)" % (n, dt, os.cpu_count() or 0)}


# algorithmic HBM bytes per env step, Sorting-4: state column read + written once (129 f64 rows + flags + step counter), action 56 B,
# obs 14 x 4 B, done/success/mode 4 B
SORT_ALG_BYTES_PER_ENV_STEP = 2 * (129 * 8 + 4 + 4) + 56 + 56 + 4


def bench_pushing(args):
    """BASELINE config 3: Pushing, 4096 envs per GPU, the 60 reference test contexts tiled, ResidualMLP 10 -> 128 x 6 -> 2 (Mish)
    stand-in policy with fixed random weights, 400-step episode cap; a step = policy forward + d3il_step."""
    import numpy as np
    import torch
    from d3il_amd import distributed as D
    from d3il_amd.agents import RandomResidualMLPPolicy
    from d3il_amd.envs.pushing import BlockPushVecEnv
    from d3il_amd.simulation.pushing_sim import load_test_contexts
    sorting = args.task == "sorting"
    if sorting:
        from d3il_amd.envs.sorting import SortingVecEnv, sample_contexts

    local_rank = _local_device()
    torch.cuda.set_device(local_rank)
    rank, world = D.init_from_env(os.environ.get("D3IL_DIST_BACKEND", "nccl"))
    dev = torch.device("cuda:%d" % local_rank)
    n = args.envs
    env = SortingVecEnv(n, device=dev) if sorting else BlockPushVecEnv(n, device=dev)
    q, iters, err = env.start()
    # Sorting: the reference's 4_test_contexts.pkl is not part of its tree; contexts are drawn like BlockContextManager.sample
    ctx60 = sample_contexts(60, 4, seed=0) if sorting else load_test_contexts()
    ctx = torch.as_tensor(ctx60[(rank * n + np.arange(n)) % len(ctx60)], dtype=torch.float64, device=dev)
    pol = RandomResidualMLPPolicy(input_dim=2 + env.obs.shape[1], device=dev)
    quat = torch.tensor([0.0, 1.0, 0.0, 0.0], dtype=torch.float64, device=dev).expand(n, 4)
    state = {}

    def begin_episode():
        env.reset(context=ctx)
        rs = env.robot_state()
        state["des"], state["z"] = rs[:, :2].clone(), rs[:, 2:3].clone()

    def one_step(t):
        if t % env.max_steps_per_episode == 0:
            begin_episode()
        obs10 = torch.cat((state["des"], env.obs.to(torch.float64)), dim=1)
        state["des"] = state["des"] + pol.predict_batch(obs10).to(torch.float64)
        act = torch.cat((state["des"], state["z"], quat), dim=1).contiguous()
        if state.get("ev") is not None:      # events on the stream the kernel is launched on (torch's current stream), one pair per step
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); env.step(act); e1.record()
            state["ev"].append((e0, e1))
        else:
            env.step(act)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for t in range(args.warmup):
        one_step(t)
    env.set_timing(True)
    kernel_ms_lib = []
    state["ev"] = []
    barrier()
    t0 = time.perf_counter()
    for t in range(args.steps):
        one_step(args.warmup + t)
        if t % 16 == 15:      # cross-check: the library's own HIP event pair around the launch, read every 16th step
            kernel_ms_lib.append(env.last_step_ms())
    barrier()
    dt = time.perf_counter() - t0
    env.set_timing(False)
    kernel_ms = [a.elapsed_time(b) for a, b in state["ev"]]      # every launch of the timed region
    t_max = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t_max, op=torch.distributed.ReduceOp.MAX)
    dt = float(t_max.item())
    st, fl, sc = env.get_state()
    bad = int(((fl >> 16) & 1).sum()), int(((fl >> 18) & 1).sum()), int(((fl >> 19) & 1).sum())
    if rank == 0:
        k_ms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
        alg_bytes = SORT_ALG_BYTES_PER_ENV_STEP if sorting else PUSH_ALG_BYTES_PER_ENV_STEP
        achieved = alg_bytes * n / (k_ms * 1e-3) / 1e9
        traffic = None
        try:  # HBM bytes per launch from the committed PMC passes of this same command (separate rocprofv3 --pmc runs)
            with open(os.path.join(ROOT, "profiles", "r01", "pmc_summary_%s.json" % args.task)) as f:
                pm = json.load(f)
            if n == 4096:
                traffic = (2 * pm["FETCH_SIZE"]["mean_per_dispatch"] + pm["WRITE_SIZE"]["mean_per_dispatch"]) * 1024.0
        except Exception:
            pass
        line = {
            "metric": "env-steps/s", "value": world * n * args.steps / dt, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("Sorting-4 task, %d envs per GPU, 60 contexts sampled like BlockContextManager.sample tiled, ResidualMLP 16->128x6->2 (Mish) "
                                    "stand-in policy with fixed random weights (torch, f32), 35 fused physics sub-steps per env step, 500-step episodes" % n) if sorting else
                                   ("Pushing task, %d envs per GPU, the 60 reference test contexts tiled, ResidualMLP 10->128x6->2 (Mish) "
                                    "stand-in policy with fixed random weights (torch, f32), 35 fused physics sub-steps per env step, "
                                    "400-step episodes" % n),
                       "envs_per_gpu": n, "n_substeps": 35, "parallelism": "env-shard x%d" % world,
                       "finite": bool(np.isfinite(st[:env.state_rows - 2]).all()), "flagged_envs_solver_overflow_offtable": bad},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_sorting_step<true>" if sorting else "k_pushing_step_split<true>", "kernel_ms": k_ms,
                         "kernel_ms_min": float(np.min(kernel_ms)) if kernel_ms else None, "kernel_ms_max": float(np.max(kernel_ms)) if kernel_ms else None,
                         "kernel_ms_library_events_every_16th": float(np.mean(kernel_ms_lib)) if kernel_ms_lib else None,
                         "algorithmic_bytes_per_launch": alg_bytes * n,
                         "note": ("FP64 instruction-issue bound (DESIGN.md section 13): one wave per SIMD, solver loops over LDS-resident systems; HBM traffic beyond the "
                                  "state column is the contact records of the constraint solver") if sorting else
                                 ("FP64 latency bound like the Avoiding step (DESIGN.md sections 4, 12.3).  Measured HBM traffic is ~25x the algorithmic bytes: "
                                  "register spills of the solver functions (private scratch) and the solver warm start / scratch rows, not state traffic")},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline_sorting(env.blob, q, ctx60) if sorting else cpu_baseline_pushing(env.blob, q, ctx60)
        print(json.dumps(line))
    env.close()
    if world > 1:
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="avoiding", choices=["avoiding", "pushing", "sorting"], help="avoiding = the headline configuration (BASELINE configs[1])")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-auto-reset", action="store_true")
    ap.add_argument("--lanes", type=int, default=None, help="environments per wave (default 64)")
    ap.add_argument("--split", type=int, default=None, help="1: two-wave controller||physics kernel, 0: fused kernel, default auto")
    ap.add_argument("--lds-pad", type=int, default=None, help="override the LDS bytes requested per workgroup (placement control)")
    args = ap.parse_args()

    import numpy as np
    import torch
    from d3il_amd import distributed as D
    from d3il_amd.envs.avoiding import ObstacleAvoidanceVecEnv

    if not torch.cuda.is_available():
        print("bench.py needs a HIP device (there is no CPU fallback for the rollout path)", file=sys.stderr)
        sys.exit(2)
    if args.task in ("pushing", "sorting"):
        return bench_pushing(args)
    local_rank = _local_device()
    torch.cuda.set_device(local_rank)
    rank, world = D.init_from_env(os.environ.get("D3IL_DIST_BACKEND", "nccl"))
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    dev = torch.device("cuda:%d" % local_rank)
    n = args.envs
    env = ObstacleAvoidanceVecEnv(n, device=dev)
    q, iters, err = env.start()
    if args.lanes is not None:
        env.set_option("lanes_per_wave", args.lanes)
    if args.split is not None:
        env.set_option("split_waves", args.split)
    if args.lds_pad is not None:
        env.set_option("lds_pad_bytes", args.lds_pad)
    env_offset = rank * n
    actions = torch.zeros(n, 7, dtype=torch.float64, device=dev)
    counts = torch.zeros(514, dtype=torch.int64, device=dev)
    episodes = torch.zeros(2, dtype=torch.int64, device=dev)   # finished, successful

    evs = None

    def one_step(t):
        env.policy_action(42, env_offset, t, actions)
        if evs is not None:      # events on the stream the kernel is launched on (torch's current stream), one pair per step
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); env.step(actions); e1.record()
            evs.append((e0, e1))
        else:
            env.step(actions)
        if not args.no_auto_reset:
            env.auto_reset(episodes)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    env.reset(); env.policy_begin()
    for t in range(args.warmup):
        one_step(t)
    episodes.zero_()
    env.set_timing(True)
    kernel_ms_lib = []
    evs = []
    barrier()
    t0 = time.perf_counter()
    for t in range(args.steps):
        one_step(args.warmup + t)
        # cross-check: HIP events recorded by the library around the step kernel on the launch stream; reading the
        # previous pair costs one event sync on an already finished kernel every 16 steps
        if t % 16 == 15:
            kernel_ms_lib.append(env.last_step_ms())
    env.count_metrics(counts)
    D.reduce_counts(counts)
    barrier()
    dt = time.perf_counter() - t0
    env.set_timing(False)
    kernel_ms = [a.elapsed_time(b) for a, b in evs]      # every launch of the timed region
    t_max = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t_max, op=torch.distributed.ReduceOp.MAX)
    dt = float(t_max.item())
    st, fl, sc = env.get_state()
    ok = bool(np.isfinite(st).all()) and not bool((fl & (1 << 16)).any())
    if rank == 0:
        value = world * n * args.steps / dt
        k_ms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
        traffic, valu = None, None
        try:  # HBM bytes per launch from the committed PMC passes of this same command (separate rocprofv3 --pmc runs)
            with open(os.path.join(ROOT, "profiles", "r01", "pmc_summary_bench300.json")) as f:
                pm = json.load(f)
            if n == 4096:
                traffic = (2 * pm["FETCH_SIZE"]["mean_per_dispatch"] + pm["WRITE_SIZE"]["mean_per_dispatch"]) * 1024.0
                # binding resource: FP64 VALU issue.  Instruction count from the PMC pass, ~2/3 of the VALU stream is FP64
                # arithmetic (static mix), an FMA counts 2 flop; peak = 78.6 TFLOP/s vector FP64 (whole chip, 1024 SIMDs)
                insts = pm["SQ_INSTS_VALU"]["mean_per_dispatch"]
                tflops = insts * 0.66 * 1.6 * 64 / (k_ms * 1e-3) / 1e12
                valu = {"bound": "fp64_valu", "valu_insts_per_launch": insts, "achieved_tflops_est": tflops, "peak_tflops": FP64_VALU_PEAK_TFLOPS,
                        "frac_est": tflops / FP64_VALU_PEAK_TFLOPS, "simds_used": 128, "simds_total": 1024,
                        "valu_active_frac_of_wave_cycles": pm["SQ_ACTIVE_INST_VALU"]["mean_per_dispatch"] / pm["SQ_WAVE_CYCLES"]["mean_per_dispatch"]}
        except Exception:
            pass
        achieved = ALG_BYTES_PER_ENV_STEP * n / (k_ms * 1e-3) / 1e9
        line = {
            "metric": "env-steps/s", "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "Avoiding task, %d envs per GPU, random policy (Philox seed 42), state obs, "
                                   "35 fused physics sub-steps per env step, auto-reset" % n,
                       "envs_per_gpu": n, "n_substeps": 35, "parallelism": "env-shard x%d" % world,
                       "auto_reset": not args.no_auto_reset, "finite_and_solver_ok": ok,
                       "episodes_finished_rank0": int(episodes[0].item()), "episodes_success_rank0": int(episodes[1].item())},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "k_avoiding_step_split<true>", "kernel_ms": k_ms,
                         "kernel_ms_library_events_every_16th": float(np.mean(kernel_ms_lib)) if kernel_ms_lib else None,
                         "kernel_ms_min": float(np.min(kernel_ms)) if kernel_ms else None, "kernel_ms_max": float(np.max(kernel_ms)) if kernel_ms else None,
                         "algorithmic_bytes_per_launch": ALG_BYTES_PER_ENV_STEP * n,
                         "note": "path is FP64-VALU issue/latency bound, not HBM bound: 756 B of HBM per env step with all 35 "
                                 "sub-steps fused in registers (DESIGN.md section 4); see the valu object and profiles/r01",
                         "valu": valu},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(env.blob, q)
        print(json.dumps(line))
    env.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
