#!/usr/bin/env python
"""bench.py - env-steps/s of the fused D3IL step() on MI355X (headline: BASELINE.json configs[1], Avoiding, 4096 envs).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--task avoiding|pushing|sorting] [--envs 4096]
                    [--policy random|mlp|scripted_push] [--no-cpu-baseline] [--no-preroll]

A "step" is one pass of the hot path over one batch: for every one of the 4096 environments per GPU the policy writes the
action (Avoiding: device-side Philox random policy, seed 42, counter = global env index x t; Pushing / Sorting: torch policy on
the resident observations), d3il_step runs the 35 fused physics sub-steps and finished environments start their next
trajectory (d3il_auto_reset) - everything enqueued on one HIP stream with the state resident in HBM, no host synchronisation
inside the loop.

Steady state.  Before the warm-up the batch is brought to the steady-state mix of episode phases (untimed "pre-roll"): the
episode counters are staggered over [0, max_steps) and max_steps env steps are run with auto-reset, so every lane has been
re-started at its own time and a K-step window measures the phase mix of a long evaluation run - not the first, contact-free
steps of a freshly reset batch.

N > 1: one process per GPU (torch.distributed / RCCL); env shards are independent (weak scaling, 4096 envs per GPU), the only
collective is the final int64 tally all-reduce, outside the per-step path but inside the timed region.  When started as plain
`python bench.py --gpus N` (no WORLD_SIZE in the environment) the script re-launches itself under torch.distributed.run with
N ranks.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6   # vector FP64, 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz
PROFILE_DIRS = ("r06",)  # committed rocprofv3 --pmc summaries of this command (separate passes), newest first

# Algorithmic HBM bytes per env step: SURVEY.md section 8(d)'s per-unit figures B_alg = 2 S + A + O + F (state read once + written once per
# fused step, f64 state, f32 action / observation) - these define `roofline.achieved`.  IMPL_BYTES is what THIS implementation's state
# column moves per env step (it also carries the flag / counter words, f64 actions, the stale-TCP / bias rows and - contact tasks - the
# solver's warm start); reported next to it as `implementation_bytes_per_launch`, not used for `frac`.
ALG_BYTES = {"avoiding": 704, "pushing": 1152, "sorting": 1620, "stacking": 1228,
             "aligning": 2 * (8 * (16 + 15) + 168 + 56 + 8) + 28 + 68 + 12,      # SURVEY 8(d)'s formula for the 16 / 15 model of the Aligning task (S_task: the 7-double target + flags): 1068
             "inserting": 2 * (8 * (30 + 27) + 168 + 16) + 28 + 44 + 8}          # the same formula for the 30 / 27 model of the Inserting task (three cubes; obs 11 f32): 1360
IMPL_BYTES = {
    "avoiding": 2 * (42 * 8 + 4 + 4) + 56 + 8 + 4,
    "pushing": 2 * (89 * 8 + 4 + 4) + 56 + 32 + 4 + 16,
    "sorting": 2 * (129 * 8 + 4 + 4) + 56 + 56 + 4,
    "stacking": 2 * (94 * 8 + 4 + 4) + 64 + 48 + 4 + 8,
    "aligning": 2 * (77 * 8 + 4 + 4) + 56 + 68 + 4 + 16,
    "inserting": 2 * (110 * 8 + 4 + 4) + 56 + 44 + 4,
}
KERNEL = {"avoiding": "k_avoiding_step_split<true, true>", "pushing": "k_sorting_step<true, false>",      # Pushing runs on the generic engine
          "sorting": "k_sorting_step<true, false>", "stacking": "k_stacking_step", "aligning": "k_aligning_step", "inserting": "k_sorting_step<true, true>"}


# ---------------------------------------------------------------------------------------------------- CPU baseline (oracle)
def _cpu_worker(task, blob_bytes, init_qpos, contexts, budget_s, seed):
    """One oracle environment on one host core for ~budget_s seconds; returns (env steps, seconds).  Test infrastructure timed as
    the CPU baseline only (oracle/d3il_oracle.c is the scalar C restatement of the reference path)."""
    import numpy as np
    from d3il_amd.model import blob as blob_mod
    from oracle.oracle import Oracle
    blob = blob_mod.ModelBlob.from_buffer_copy(blob_bytes)
    o = Oracle(blob)
    o.env_start(init_qpos)
    rng = np.random.default_rng(seed)
    n, ep = 0, seed
    t0 = time.perf_counter()
    if task == "avoiding":
        o.env_reset()
        s, _ = o.env_state()
        des = s[25:28].copy()
        while time.perf_counter() - t0 < budget_s:
            des[:2] += rng.uniform(-0.01, 0.01, 2)
            _, done, _, _ = o.env_step(np.array([des[0], des[1], des[2], 0, 1, 0, 0]))
            n += 1
            if done:
                o.env_reset()
                s, _ = o.env_state()
                des = s[25:28].copy()
        return n, time.perf_counter() - t0
    if task == "stacking":      # scripted pick-and-place of one context (the GPU workload's policy), repeated
        from d3il_amd.controllers.scripted_stacking import build_trajectory
        js = blob_mod.load_json("stacking")
        ctx = contexts[seed % len(contexts)]
        traj = build_trajectory(js, init_qpos, ctx, speed=0.5)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < budget_s:
            o.stack_reset(ctx)
            for a in traj:
                _, done, _ = o.stack_step(a)
                n += 1
                if done or time.perf_counter() - t0 >= budget_s:
                    break
        return n, time.perf_counter() - t0
    import torch
    torch.set_num_threads(1)
    if task == "aligning":      # the GPU workload's scripted policy (inside / outside pushes alternate over the contexts), episode after episode
        from d3il_amd.agents import ScriptedAlignPolicy
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < budget_s:
            c = ep % len(contexts)
            obs = o.align_reset(contexts[c])
            des = np.array(o.body(blob.tcp_body)[0], dtype=np.float64)
            pol = ScriptedAlignPolicy(inside=[c % 2 == 0], device="cpu")
            for t in range(400):
                x = torch.as_tensor(np.concatenate([des, obs.astype(np.float64)])[None], dtype=torch.float64)
                des = des + pol.predict_batch(x)[0].numpy()
                obs, _, done, _ = o.align_step(np.concatenate([des, [0, 1, 0, 0]]))
                n += 1
                if done or time.perf_counter() - t0 >= budget_s:
                    break
            ep += 1
        return n, time.perf_counter() - t0
    from d3il_amd.agents import RandomResidualMLPPolicy
    pol = RandomResidualMLPPolicy(input_dim={"pushing": 10, "inserting": 13}.get(task, 16), device="cpu")
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        if task == "pushing":
            obs = o.push_reset(contexts[ep % len(contexts)])
            s, _ = o.push_state()
            des, z = s[25:27].copy(), s[27]
        elif task == "inserting":
            obs = o.ins_reset(contexts[ep % len(contexts)])
            des, z = obs[:2].astype(np.float64), float(o.body(blob.tcp_body)[0][2])
        else:
            obs = o.sort_reset(contexts[ep % len(contexts)].reshape(-1, 7))
            des, z = obs[:2].astype(np.float64), float(o.body(blob.tcp_body)[0][2])
        for t in range({"pushing": 400, "inserting": 2000}.get(task, 500)):
            x = torch.as_tensor(np.concatenate([des, obs.astype(np.float64)])[None], dtype=torch.float64)
            des = des + pol.predict_batch(x)[0].numpy().astype(np.float64)
            a = np.array([des[0], des[1], z, 0, 1, 0, 0])
            if task == "pushing":
                obs, _, done, _ = o.push_step(a)
            elif task == "inserting":
                obs, done, _ = o.ins_step(a)
            else:
                obs, done, _ = o.sort_step(a)
            n += 1
            if done or time.perf_counter() - t0 >= budget_s:
                break
        ep += 1
    return n, time.perf_counter() - t0


def _available_cores():
    """(cores this process may use, logical CPUs of the host): the scheduler affinity capped by the cgroup CPU quota (cpu.max of cgroup
    v2, cfs_quota_us / cfs_period_us of v1) - a container with a 16-CPU quota on a 256-thread host runs 16 workers, not 256."""
    logical = os.cpu_count() or 1
    cores = logical
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = float(f.read()), float(g.read())
                if q > 0:
                    quota = q / per
        except Exception:
            pass
    if quota is not None:
        cores = max(1, min(cores, int(quota + 0.5)))
    return cores, logical


def cpu_baseline(task, blob, init_qpos, contexts, budget_s=10.0):
    """The oracle timed on the GPU box's host cores on a bounded sample of the same workload: first one environment on one core,
    then one environment per core on all cores (independent OS processes, like the reference's n_cores workers; plain
    subprocesses of this script - the parent holds an initialised HIP runtime, so nothing is forked)."""
    import tempfile
    import numpy as np
    n1, t1 = _cpu_worker(task, bytes(blob), init_qpos, contexts, budget_s * 0.5, 0)
    cores, logical = _available_cores()
    with tempfile.TemporaryDirectory() as td:
        arg = os.path.join(td, "args.npz")
        np.savez(arg, blob=np.frombuffer(bytes(blob), dtype=np.uint8), init_qpos=np.asarray(init_qpos, dtype=np.float64),
                 contexts=np.zeros((0, 0)) if contexts is None else np.asarray(contexts, dtype=np.float64))
        env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
        t0 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", task, arg, str(budget_s), str(1 + i)],
                                  stdout=subprocess.PIPE, env=env) for i in range(cores)]
        outs = [p.communicate()[0] for p in procs]
        wall = time.perf_counter() - t0
    res = [tuple(float(x) for x in o.decode().split()[-2:]) for o in outs if o.strip()]
    total = sum(r[0] for r in res)
    busy = max(r[1] for r in res) if res else float("nan")
    pol = {"avoiding": "random policy", "stacking": "scripted pick-and-place", "aligning": "scripted inside / outside pushes"}.get(task, "ResidualMLP stand-in policy on the CPU")
    return {"value": total / busy, "unit": "env-steps/s", "cores": len(res), "kind": "port",
            "single_core_value": n1 / t1, "host_logical_cpus": logical,
            "sample": "one oracle environment per core on all %d cores this container may use (affinity capped by the cgroup CPU quota) (%s, %d env steps of 35 (Stacking: 30) sub-steps in %.1f s of stepping per worker, %.1f s wall "
                      "including interpreter start-up), after one environment on one core (%d env steps in %.1f s); scalar C oracle (oracle/d3il_oracle.c, the CPU "
                      "restatement of the reference path - the Python reference itself is bounded above by 146 env-steps/s/core, BASELINE.md section 2)"
                      % (len(res), pol, total, busy, wall, n1, t1)}


def _cpu_worker_main(argv):
    import numpy as np
    task, arg, budget, seed = argv[0], argv[1], float(argv[2]), int(argv[3])
    z = np.load(arg)
    ctx = z["contexts"] if z["contexts"].size else None
    n, t = _cpu_worker(task, z["blob"].tobytes(), z["init_qpos"], ctx, budget, seed)
    print(n, t)


# ---------------------------------------------------------------------------------------------------- helpers
def _local_device() -> int:
    """GPU of this rank: LOCAL_RANK (one process per GPU).  D3IL_BENCH_FORCE_DEVICE pins every rank to one GPU - only for
    exercising the multi-process code path on a single-GPU box (with D3IL_DIST_BACKEND=gloo; RCCL needs distinct GPUs)."""
    if os.environ.get("D3IL_BENCH_FORCE_DEVICE") is not None:
        return int(os.environ["D3IL_BENCH_FORCE_DEVICE"])
    return int(os.environ.get("LOCAL_RANK", "0"))


def _free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _self_spawn(n_gpus: int) -> int:
    """`python bench.py --gpus N` without a launcher: run N ranks (one per GPU) under torch.distributed.run and pass its output on."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def _pmc(task, policy, n, S):
    """Counters per launch of the step kernel from the committed rocprofv3 --pmc passes of THIS command line - same task, policy (regime), environments
    per GPU and sub-batch count (tools/profile_r06.sh writes profiles/<round>/pmc/<task>_<policy>_sb<S>.json; the passes are separate runs by construction:
    counters cannot be collected inside the timed run, and a counter pass serialises dispatches).  No borrowing across regimes: a line whose command was
    not profiled carries no traffic figure."""
    if n != 4096:
        return None, None
    for d in PROFILE_DIRS:
        path = os.path.join(ROOT, "profiles", d, "pmc", "%s_%s_sb%d.json" % (task, policy, S))
        try:
            with open(path) as f:
                pm = json.load(f)
            if "FETCH_SIZE" not in pm or "WRITE_SIZE" not in pm:      # an incomplete counter summary (a pass that did not finish) is no evidence
                continue
            return pm, os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def _hbm_bytes(pm):
    return (2 * pm["FETCH_SIZE"]["mean_per_dispatch"] + pm["WRITE_SIZE"]["mean_per_dispatch"]) * 1024.0      # MI355X_MICROARCH.md: KiB units, FETCH_SIZE counts half on gfx950


def _isa_mix(task):
    """Counted FP64 instruction mix of the task's step kernel (tools/isa_fp64_mix.py), newest committed profile round first."""
    for d in PROFILE_DIRS:
        path = os.path.join(ROOT, "profiles", d, "isa_fp64_mix.json")
        try:
            with open(path) as f:
                return json.load(f)[task], os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


POLICY_TEXT = {
    "random": "random policy",
    "mlp": "ResidualMLP %d->128x6->2 (Mish) stand-in policy with fixed random weights (torch, f32)",
    "ddpm": "DDPM policy of BASELINE config 4 (DiffusionMLP %d->256x8->2, t_dim 8, 4 denoising steps, fixed random weights, f32: the sampling chain as one matrix-core kernel of the library, d3il_ddpm_mlp_f32; D3IL_POLICY_FUSED_DDPM=0 = the torch chain)",
    "beso": "BESO policy of BASELINE config 5 (DiffusionGPT 6 layers x 6 heads x 120, window 5, 16 Euler-ancestral steps, fixed random weights, f32; GEMMs on the matrix cores with split-f16 operands, f32 accumulate)",
    "scripted_push": "scripted pushing policy (every rod drives a cube to its target / bin: the contact regime)",
    "scripted_align": "scripted pushes from inside / outside the box walls (the two behaviour modes of the Aligning task)",
    "scripted_stack": "scripted pick-and-place policy (joint-space table from host IK: grasp, carry, stack - the contact regime of the task)",
}


def _random_ddpm(obs_dim, dev, graph=False):
    """BASELINE config 4's policy: the reference's DDPM agent as scripts/sorting_4/ddpm_benchmark.sh configures it (DiffusionMLP obs
    16 -> act 2, hidden 256 x 8 layers, t_dim 8, n_timesteps 4, window 1) with fixed random weights (torch seed 0; no checkpoints
    offline), scaled actions clamped to +-1 = +-0.01 m (the env's action box, pushing.py:203-205)."""
    import torch
    from d3il_amd.policies import DDPMPolicy, DiffusionMLP, Scaler
    torch.manual_seed(0)
    net = DiffusionMLP(action_dim=2, obs_dim=obs_dim, t_dim=8, hidden_dim=256, num_hidden_layers=8).to(dev)
    sc = Scaler([0.0] * obs_dim, [1.0] * obs_dim, [0.0, 0.0], [0.01, 0.01], y_bounds=[[-1.0, -1.0], [1.0, 1.0]], device=dev)
    pol = DDPMPolicy(net, sc, n_timesteps=4, window_size=1)
    return pol.captured() if graph else pol


def _random_beso(dev):
    """BASELINE config 5's policy: the reference's BESO agent as scripts/stacking/beso_benchmark.sh configures it (DiffusionGPT state 20,
    action 8, n_embd 120, 6 layers, 6 heads, window 5; 16 Euler-ancestral steps, sigma 0.01 .. 1) with fixed random weights (torch
    seed 0).  Action scaling: joint deltas +-0.01 rad, gripper command 0.04 +- 0.04 (open iff > 0.075, stacking.py:337-346)."""
    import torch
    from d3il_amd.policies import BESOPolicy, DiffusionGPT, Scaler
    torch.manual_seed(0)
    net = DiffusionGPT(state_dim=20, action_dim=8, embed_dim=120, n_layers=6, n_heads=6, obs_seq_len=5).to(dev)
    sc = Scaler([0.0] * 20, [1.0] * 20, [0.0] * 7 + [0.04], [0.01] * 7 + [0.04], y_bounds=[[-1.0] * 8, [1.0] * 8], device=dev)
    return BESOPolicy(net, sc, window_size=5, num_sampling_steps=16, sigma_min=0.01, sigma_max=1.0, use_graph=os.environ.get("D3IL_POLICY_GRAPH", "1") == "1")


def _ddpm_policy_roofline(pol, n_rows, dev):
    """Config 4's policy kernel (d3il_ddpm_mlp_f32): one predict call of a sub-batch's rows, event-timed on the current stream after the
    benchmark loop (alone on the GPU); algorithmic flops of the denoiser (4 steps x (26 x 256 + 8 x 256 x 256 + 256 x 2) multiply-adds per row) against the
    dense f32 MFMA peak."""
    import torch
    F32_MFMA_PEAK_TFLOPS = 157.3
    inner = getattr(pol, "inner", pol)
    if not (hasattr(inner, "fused_ok") and inner.fused_ok()):
        return None
    sd = inner.model.layers.layers[0].in_features - 10
    s = torch.randn(n_rows, sd, device=dev)
    with torch.no_grad():
        for _ in range(5):
            inner._sample_fused(s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            inner._sample_fused(s)
        e1.record()
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / 50
    nblk = len(inner.model.layers.layers) - 2
    flops = 2.0 * n_rows * inner.T * ((10 + sd) * 256 + 2 * nblk * 256 * 256 + 256 * 2)
    return {"bound": "mfma", "kernel": "k_ddpm_mlp_f32", "achieved": flops / (ms * 1e-3) / 1e12, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": flops / (ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, "kernel_ms": ms, "rows": n_rows, "launches_per_policy_step": 1,
            "note": "one launch per predict call and sub-batch (+ one torch.randn); a call of <= 4096 rows is one workgroup's latency chain (16 rows per workgroup, at most "
                    "256 workgroups): the fraction rises with the row count up to 0.5 at 16384 rows (tools/gpu_ddpm_kernel_time.py)"}


def _beso_policy_roofline(pol, n, dev):
    """Config 5 is policy-bound: the dominant kernel of a policy step (the fused transformer MLP of a DiffusionGPT block), event-timed on the current stream after the
    benchmark loop.  Default GEMM mode (policies.policy_gemm_mode() == "f16x3"): the split-f16 kernel - `achieved` counts the f16 matrix flops it ISSUES (three f16
    products per f32 product) against the dense f16 MFMA peak; `f32_equivalent_tflops` is the algorithmic rate (what an f32 GEMM of the same shape would be credited
    with).  D3IL_POLICY_GEMM=f32: the f32-input MFMA kernel of rounds 3 - 5 against the dense f32 MFMA peak."""
    import torch
    from d3il_amd import capi
    from d3il_amd.policies import pack_mlp_weights, pack_mlp_weights_f16x3, policy_gemm_mode
    F32_MFMA_PEAK_TFLOPS, F16_MFMA_PEAK_TFLOPS = 157.3, 2500.0      # MI355X_MICROARCH.md: dense peaks
    f16x3 = policy_gemm_mode() == "f16x3"
    blk = pol.inner.blocks[0]
    fc1, fc2 = blk.mlp[0], blk.mlp[2]
    T = 2 * pol.W + 1
    M = n * T
    x = torch.randn(M, 120, device=dev)
    out = torch.empty_like(x)
    wp = pack_mlp_weights_f16x3(fc1.weight, fc2.weight) if f16x3 else pack_mlp_weights(fc1, fc2)
    L = capi.load()
    fn = L.d3il_mlp_ln_gelu_residual_f16x3 if f16x3 else L.d3il_mlp_ln_gelu_residual_f32
    st = torch.cuda.current_stream(dev).cuda_stream

    def run():
        capi.check(fn(x.data_ptr(), blk.ln2.weight.data_ptr(), blk.ln2.bias.data_ptr(), float(blk.ln2.eps), x.data_ptr(), wp.data_ptr(),
                      fc1.bias.data_ptr(), fc2.bias.data_ptr(), out.data_ptr(), M, 120, 480, st))
    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / 50
    flops = 2.0 * M * (120 * 480) * 2
    calls = len(pol.inner.blocks) * (len(pol.sigmas) - 1)
    issued = flops * (3.0 * 128 / 120 if f16x3 else 1.0)      # three f16 products per f32 product, K padded 120 -> 128 in the first product (counted for both, slightly high)
    peak = F16_MFMA_PEAK_TFLOPS if f16x3 else F32_MFMA_PEAK_TFLOPS
    return {"bound": "mfma", "kernel": "k_mlp_gelu_residual_f16x3<4>" if f16x3 else "k_mlp_gelu_residual_f32", "gemm_mode": "f16x3 (split-f16 operands, f32 accumulate)" if f16x3 else "f32",
            "achieved": issued / (ms * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s", "frac": issued / (ms * 1e-3) / 1e12 / peak,
            "f32_equivalent_tflops": flops / (ms * 1e-3) / 1e12, "kernel_ms": ms, "rows": M, "launches_per_policy_step": calls,
            "note": "dense MFMA peak of the operand type (MI355X_MICROARCH.md); the kernel runs %d times per policy step (6 blocks x 16 sampling steps), next to two linear "
                    "kernels and the attention kernel per block; counters (profiles/r06/f16x3_mlp_pmc.log): the wave is issue / dependency bound at two waves per SIMD, the "
                    "matrix pipe is busy a fifth of the kernel" % calls}


# ---------------------------------------------------------------------------------------------------- the benchmark
def _make_env(args, task, dev, n):
    if task == "avoiding":
        from d3il_amd.envs.avoiding import ObstacleAvoidanceVecEnv
        return ObstacleAvoidanceVecEnv(n, device=dev)
    if task == "pushing":
        from d3il_amd.envs.pushing import BlockPushVecEnv
        return BlockPushVecEnv(n, device=dev)
    if task == "sorting":
        from d3il_amd.envs.sorting import SortingVecEnv
        return SortingVecEnv(n, device=dev, max_steps_per_episode=args.max_steps or 700)     # configs/sorting_4_config.yaml:80
    if task == "aligning":
        from d3il_amd.envs.aligning import RobotPushVecEnv
        return RobotPushVecEnv(n, device=dev)
    if task == "inserting":
        from d3il_amd.envs.inserting import GateInsertionVecEnv
        return GateInsertionVecEnv(n, device=dev, max_steps_per_episode=args.max_steps or 2000)     # gate_insertion.py:158 (the reference has no config for this task)
    from d3il_amd.envs.stacking import CubeStackingVecEnv
    return CubeStackingVecEnv(n, device=dev)


class _Shard:
    """One sub-batch of this rank's environments: its own environment handle, policy state and HIP stream.  The rank's environments are stepped as
    `--sub-batches` independent sub-batches: a launch lasts as long as its slowest workgroup (an environment in a rare path - rod contact,
    clipped-eigenvalue IK, a hard contact island), and sub-batches on different streams do not wait for each other's tails."""

    def __init__(self, args, task, dev, sb, ctx60, q, stack_tables):
        import numpy as np
        import torch
        # the handle, its stream and the serve-wave threshold are the product's (d3il_amd/envs/sub_batch.py SubBatchSet - what the Sim classes use)
        n, env_offset = sb.n, args.rank_offset + sb.offset
        self.task, self.n, self.env_offset, self.dev = task, n, env_offset, dev
        self.own_stream, self.stream = sb.own_stream, sb.stream
        with torch.cuda.stream(self.stream):
            env = self.env = sb.env
            env.set_init_qpos(q)
            if args.lanes is not None:
                env.set_option("lanes_per_wave", args.lanes)
            if args.split is not None:
                env.set_option("split_waves", args.split)
            if args.lds_pad is not None:
                env.set_option("lds_pad_bytes", args.lds_pad)
            if args.serve_max_wg is not None:      # otherwise SubBatchSet's rule: 256 CUs / number of sub-batches (the library sees one sub-batch, the set all of them)
                env.set_option("serve_wave_max_workgroups", args.serve_max_wg)
            if args.solver_strict:
                env.set_option("solver_strict", 1)
            self.graph = bool(args.graph_rollout) and task == "avoiding" and self.own_stream and (args.policy or "random") == "random" and not args.no_auto_reset
            ctx_id = None
            if ctx60 is not None:
                ids = (env_offset + np.arange(n)) % len(ctx60)
                ctx_id = torch.as_tensor(ids, dtype=torch.int32, device=dev)
                env.reset(context=ctx60[ids])
            else:
                env.reset()
            env.policy_begin()
            self.table = env.set_tally(len(ctx60) if ctx60 is not None else 1, ctx_id)
            self.episodes = torch.zeros(2, dtype=torch.int64, device=dev)   # finished, successful
            self.actions = actions = torch.zeros(n, env.action_dim, dtype=torch.float64, device=dev)
            policy = self.policy = args.policy or {"avoiding": "random", "stacking": "scripted_stack", "aligning": "scripted_align"}.get(task, "mlp")
            pol = None
            self.last_cmd = None
            if task == "stacking":
                from d3il_amd.agents import RandomResidualMLPPolicy, ScriptedStackPolicy
                if policy == "scripted_stack":
                    pol = ScriptedStackPolicy(stack_tables, ctx_id.to(torch.int64), device=dev)
                elif policy == "mlp":
                    pol = RandomResidualMLPPolicy(input_dim=20, output_dim=8, device=dev, bound=0.01)
                elif policy == "beso":
                    pol = _random_beso(dev)
                else:
                    raise SystemExit("--policy %s is not available for task %s" % (policy, task))
                self.last_cmd = env.robot_state().to(torch.float32).clone()                      # stacking_sim.py:90-91
            elif task == "aligning":
                from d3il_amd.agents import RandomResidualMLPPolicy, ScriptedAlignPolicy
                if policy == "scripted_align":
                    pol = ScriptedAlignPolicy(inside=(ctx_id % 2 == 0), device=dev)
                elif policy == "mlp":
                    pol = RandomResidualMLPPolicy(input_dim=20, output_dim=3, device=dev, bound=0.01)
                else:
                    raise SystemExit("--policy %s is not available for task %s" % (policy, task))
                actions[:, 3:] = torch.tensor([0.0, 1.0, 0.0, 0.0], dtype=torch.float64, device=dev)
            elif task != "avoiding" or policy != "random":
                from d3il_amd.agents import RandomResidualMLPPolicy, ScriptedPushPolicy
                if policy == "mlp":
                    pol = RandomResidualMLPPolicy(input_dim=2 + env.obs.shape[1], device=dev)
                    if args.policy_graph:
                        from d3il_amd.policies import CapturedPolicy
                        pol = CapturedPolicy(pol)
                elif policy == "scripted_push":
                    pol = ScriptedPushPolicy(task, device=dev)
                elif policy == "ddpm":
                    pol = _random_ddpm(2 + env.obs.shape[1], dev, graph=bool(args.policy_graph))
                else:
                    raise SystemExit("--policy %s is not available for task %s" % (policy, task))
                actions[:, 3:] = torch.tensor([0.0, 1.0, 0.0, 0.0], dtype=torch.float64, device=dev)
            self.pol = pol
            if task == "avoiding" and args.fuse_tail:
                env.set_option("fuse_rollout_tail", 1)      # two launches per rollout step instead of five and a copy
            if self.graph:
                env.set_option("graph_rollout", 1)      # after set_tally / every other option: the first step captures what a step launches NOW
            self.des_xy = env.policy_des[:2, :n]                              # [2, n] view: the harness set-point the library re-latches on auto-reset
            self.des_z = env.policy_des[2, :n]
        self.auto_reset = not args.no_auto_reset

    def one_step(self, t):
        import torch
        env, pol, n, task, actions = self.env, self.pol, self.n, self.task, self.actions
        if pol is None and task == "avoiding":      # BASELINE config 2: the device random policy - policy, step and auto-reset in one library call
            if self.auto_reset:
                env.random_rollout_step(42, self.env_offset, t, actions, self.episodes)
            else:
                env.policy_action(42, self.env_offset, t, actions); env.step(actions)
            return
        if self.own_stream:
            torch.cuda.set_stream(self.stream)       # the policy's torch kernels go to this sub-batch's stream
        if task == "stacking":
            if hasattr(pol, "begin_episodes"):
                pol.begin_episodes(env.last_reset)
            self.last_cmd = torch.where(env.last_reset.bool().unsqueeze(1), env.robot_state().to(torch.float32), self.last_cmd)
            obs20 = torch.cat((self.last_cmd, env.obs), dim=1)                           # np.concatenate((pred_action, obs)), stacking_sim.py:99
            out = pol.predict_batch(obs20).to(torch.float32)
            self.last_cmd = torch.cat((out[:, :7] + obs20[:, :7], out[:, 7:8]), dim=1)   # stacking_sim.py:104
            actions.copy_(self.last_cmd)
        elif task == "aligning":
            if hasattr(pol, "begin_episodes"):
                pol.begin_episodes(env.last_reset)
            des3 = env.policy_des[:, :n]                                             # [3, n]: the library re-latches it to the TCP on auto-reset
            obs_in = torch.cat((des3.t(), env.obs.to(torch.float64)), dim=1)        # np.concatenate((pred_action[:3], obs)), aligning_sim.py:99
            des3.add_(pol.predict_batch(obs_in).to(torch.float64).t())              # aligning_sim.py:101-102: x, y and z are commanded
            actions[:, 0:3] = des3.t()
        else:
            if hasattr(pol, "begin_episodes"):
                pol.begin_episodes(env.last_reset)
            obs_in = torch.cat((self.des_xy.t(), env.obs.to(torch.float64)), dim=1)      # np.concatenate((pred_action[:2], obs)), pushing_sim.py:75
            self.des_xy.add_(pol.predict_batch(obs_in).to(torch.float64).t())            # pushing_sim.py:78
            actions[:, 0:2] = self.des_xy.t()
            actions[:, 2] = self.des_z
        if self.auto_reset:
            env.step_auto_reset(actions, self.episodes)
        else:
            env.step(actions)


DEFAULT_SUB_BATCHES = {"avoiding": 4, "pushing": 4, "sorting": 4, "inserting": 4}      # measured: tools/gpu_subbatch.py, DESIGN section 18.10


def run(args):
    import numpy as np
    import torch
    from d3il_amd import distributed as D

    task = args.task
    local_rank = _local_device()
    torch.cuda.set_device(local_rank)
    rank, world = D.init_from_env(os.environ.get("D3IL_DIST_BACKEND", "nccl"))
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d (the line reports n_gpus = WORLD_SIZE)" % (args.gpus, world), file=sys.stderr)
    dev = torch.device("cuda:%d" % local_rank)
    if world > 1 and torch.distributed.get_backend() == "nccl":
        # The first collective creates torch.distributed's own RCCL communicator: ranks that were given the SAME GPU (a launcher without LOCAL_RANK pinning, a
        # one-GPU box) fail here ("Duplicate GPU detected") - on every rank alike, RCCL compares the bus ids of all ranks.  That is a set-up error of the launch,
        # not of the path: exit code 3 (the documented multi-GPU self-check code) with a message instead of a stack trace per rank.
        try:
            probe = torch.ones(1, dtype=torch.int32, device=dev)
            torch.distributed.all_reduce(probe)
            torch.cuda.synchronize()
            ok = int(probe.item()) == world
        except Exception as exc:      # noqa: BLE001 - whatever RCCL / c10d raises
            ok = False
            if rank == 0:
                print("bench.py: multi-GPU self-check failed: the RCCL communicator of %d ranks could not be created (%s: %s); one process per DISTINCT GPU is "
                      "required under the nccl backend" % (world, type(exc).__name__, str(exc).splitlines()[0][:300]), file=sys.stderr)
        if not ok:
            sys.exit(3)
    n = args.envs
    env_offset = rank * n
    ctx60 = None
    if task == "pushing":
        from d3il_amd.simulation.pushing_sim import load_test_contexts
        ctx60 = load_test_contexts()
    elif task == "sorting":
        from d3il_amd.envs.sorting import sample_contexts
        ctx60 = sample_contexts(60, 4, seed=0)     # the reference's 4_test_contexts.pkl is not part of its tree
    elif task == "aligning":
        from d3il_amd.envs.aligning import load_test_contexts as load_align_contexts
        ctx60 = load_align_contexts()
    elif task == "inserting":
        from d3il_amd.envs.inserting import sample_contexts as sample_insert_contexts
        ctx60 = sample_insert_contexts(60, seed=0)
    elif task == "stacking":
        from d3il_amd.envs.stacking import load_test_contexts as load_stack_contexts
        ctx60 = load_stack_contexts()[:args.stack_contexts]     # the first contexts of the reference's 100 test contexts, tiled
    # sub-batches: the rank's environments as S independent sub-batches on S streams (1 = one launch per step over the whole batch)
    S = args.sub_batches if args.sub_batches is not None else DEFAULT_SUB_BATCHES.get(task, 1)
    if args.policy_graph is None:
        args.policy_graph = 1 if args.policy == "ddpm" else 0      # (the stand-in MLP is ~25 kernels per step: not host bound, no gain from the capture)
    if args.sub_batches is None and (args.policy == "beso" or (args.policy == "ddpm" and not args.policy_graph)):
        S = 1      # the diffusion policies launch hundreds of torch kernels per step and sub-batch: issued one by one, four sub-batches are host bound (DDPM 0.53 M
                   # against 0.67 M as one batch; captured as one graph per sub-batch - policies.CapturedPolicy - 0.81 M: profiles/r05/policy_graph/).  BESO
                   # (Stacking: the cooperative engine fills the chip with one launch) stays one batch with its own captured sampling loop
    if S < 1 or n % S != 0 or n // S < 64:
        S = 1
    args.sub_batches = S
    # env.start(): the offline IK to the task's start pose (host, once)
    from d3il_amd.controllers.offline_ik import offline_ik
    from d3il_amd.kinematics import UrdfChain
    from d3il_amd.model import blob as blob_mod
    js = blob_mod.load_json(task)
    c_, tc_ = js["controller"], js["task_const"]
    q, iters, err = offline_ik(UrdfChain(js["urdf_chain"]), c_["default_qpos"], list(tc_["init_end_eff_pos"]) + list(tc_["init_end_eff_quat"]),
                               np.array(c_["joint_pos_min"]), np.array(c_["joint_pos_max"]))
    stack_tables = None
    if task == "stacking" and (args.policy or "scripted_stack") == "scripted_stack":
        from d3il_amd.controllers.scripted_stacking import build_trajectory
        stack_tables = [build_trajectory(js, q, c, speed=0.5) for c in ctx60]     # host IK once per context (untimed set-up)
    from d3il_amd.envs.sub_batch import SubBatchSet
    args.rank_offset = env_offset
    batches = SubBatchSet(n, S, dev, lambda cnt, off: _make_env(args, task, dev, cnt))      # handles, streams, serve-wave threshold, hardware queues
    shards = [_Shard(args, task, dev, sb, ctx60, q, stack_tables) for sb in batches]
    env = shards[0].env
    policy, pol = shards[0].policy, shards[0].pol
    max_steps = env.max_steps_per_episode

    default_stream = torch.cuda.current_stream(dev)

    def one_step(t):
        for sh in shards:
            sh.one_step(t)
        if S > 1:
            torch.cuda.set_stream(default_stream)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # the metric reduction is the library's own RCCL all-reduce (d3il_reduce_metrics); the communicator is set up once, outside the timed
    # region.  If RCCL cannot be resolved / initialised (e.g. the gloo test mode with every rank on one GPU) the line says so and the
    # reduction goes through torch.distributed instead - the numbers are integer sums either way.
    lib_comm, reduction, rccl_ranks = None, "none (single process)", None
    if world > 1:
        lib_comm = D.auto_comm(dev)          # collective; every failure mode is agreed on by all ranks inside LibraryComm (no rank is left behind)
        if lib_comm is not None:
            rccl_ranks = lib_comm.ranks()    # ncclCommCount: what RCCL itself sees - the driver's SCALE run can check it against n_gpus
            reduction = "libd3il_rollout d3il_reduce_metrics: one RCCL ncclAllReduce(sum, int64) of the tally table"
        else:
            reduction = "torch.distributed all_reduce (%s)" % torch.distributed.get_backend()
        # self-check of the multi-GPU run: under the nccl backend the reduction MUST be the library's RCCL call over exactly
        # WORLD_SIZE ranks - anything else (fallback to torch.distributed, a communicator of another size) makes every rank fail loudly instead of
        # printing a line that looks like a scaling result.  (auto_comm agrees on success / failure across ranks, so all ranks take the same branch.)
        if torch.distributed.get_backend() == "nccl" and os.environ.get("D3IL_ALLOW_REDUCTION_FALLBACK") != "1":
            if lib_comm is None or rccl_ranks != world:
                if rank == 0:
                    print("bench.py: multi-GPU self-check failed: metric reduction = %s, rccl_ranks = %s, WORLD_SIZE = %d (set D3IL_ALLOW_REDUCTION_FALLBACK=1 to run anyway)"
                          % (reduction, rccl_ranks, world), file=sys.stderr)
                torch.distributed.destroy_process_group()
                sys.exit(3)
    t_run = 0
    preroll = 0
    if not args.no_preroll and not args.no_auto_reset:
        # steady-state phase mix: lane i pretends to be (i * 977) % max_steps steps into its episode, then one full episode length of
        # untimed steps: every lane is re-started at its own time
        for sh in shards:
            with torch.cuda.stream(sh.stream):
                stagger = (torch.arange(sh.n, device=dev, dtype=torch.int64) + sh.env_offset) * 977 % max_steps
                sh.env.step_count[:sh.n] = stagger.to(torch.int32)
        preroll = max_steps if args.preroll is None else args.preroll
        for t in range(preroll):
            one_step(t_run); t_run += 1
    for t in range(args.warmup):
        one_step(t_run); t_run += 1
    barrier()
    for sh in shards:
        sh.episodes.zero_(); sh.table.zero_()
        sh.env.set_timing(True)      # HIP events around EVERY step-kernel launch, on the stream it is launched on, kept in a ring inside the library
        if sh.graph:                 # the captured form of the step (with its event pairs) is built here, outside the timed region
            sh.env.random_rollout_prepare(42, sh.env_offset, t_run, sh.actions, sh.episodes)
    barrier()
    stats_lib = os.environ.get("D3IL_STATS_LIB") == "1"      # diagnostics library (device-side phase timers): its counters over the timed region go to stderr
    if stats_lib:
        import ctypes as _C
        from d3il_amd import capi
        _st = (_C.c_uint64 * 32)()
        capi.check(capi.load().d3il_debug_stats(_st, 1))

        def _diag0():      # per-environment phase timers of environment 0 (cooperative engine: SG_DIAG + 4 ..; accumulated since creation)
            buf = np.zeros(32 * 36 + 24)
            if capi.load().d3il_debug_scratch(env.h, 0, buf.ctypes.data_as(_C.c_void_p), len(buf)) != 0:
                return None
            return buf[32 * 36 + 4:32 * 36 + 17].copy()
        _d0 = _diag0() if task in ("stacking", "aligning") else None
    t0 = time.perf_counter()
    for t in range(args.steps):
        one_step(t_run); t_run += 1
    torch.cuda.synchronize()                         # all sub-batch streams
    if stats_lib:
        capi.check(capi.load().d3il_debug_stats(_st, 0))
        print("device stats (g_dev_stats[0..31], timed region of %d steps): %s" % (args.steps, json.dumps([int(v) for v in _st])), file=sys.stderr)
        if _d0 is not None:
            _d1 = _diag0()
            if _d1 is not None:
                print("environment 0 phase ticks (slots 0 .. 12: 0 arm dynamics + tables, 1 Cartesian controller (Aligning), 3 limit rows / start point, 5 integration): %s" % json.dumps([float(v) for v in (_d1 - _d0)]), file=sys.stderr)
    table = shards[0].table
    for sh in shards[1:]:
        table += sh.table
    episodes = sum(sh.episodes for sh in shards)
    D.reduce_counts(table, lib_comm, env.h)          # the one collective of the path: int64 episode tally, RCCL inside the library
    barrier()
    dt = time.perf_counter() - t0
    tstats = [sh.env.timing_stats() for sh in shards]      # (sum, min, max, launches) over every launch of the timed region
    for sh in shards:
        sh.env.set_timing(False)
    n_launches = sum(x[3] for x in tstats)
    t_max = torch.tensor([dt], dtype=torch.float64, device=dev)
    dt_by_rank = [dt]
    if world > 1:
        gathered = [torch.zeros_like(t_max) for _ in range(world)]
        torch.distributed.all_gather(gathered, t_max)
        dt_by_rank = [float(x.item()) for x in gathered]
        torch.distributed.all_reduce(t_max, op=torch.distributed.ReduceOp.MAX)
    dt = float(t_max.item())
    states = [sh.env.get_state() for sh in shards]
    st = np.concatenate([x[0] for x in states], axis=1)
    fl = np.concatenate([x[1] for x in states])
    n_state = env.state_rows - (2 if task in ("sorting", "inserting") else 0)
    n_sub = env.n_substeps
    n_launch = n // S                                 # environments per launch of the step kernel
    finite = bool(np.isfinite(st[:n_state]).all())      # (the Aligning mean distance / reward may be NaN like the reference's: they are info rows, not state)
    flagged = {"solver_fail": int(((fl >> 16) & 1).sum())}
    if task != "avoiding":
        flagged.update(contact_overflow=int(((fl >> 18) & 1).sum()), off_table=int(((fl >> 19) & 1).sum()))
    if task in ("stacking", "aligning"):
        flagged.update(hand_near=int(((fl >> 20) & 1).sum()))
    if rank == 0:
        value = world * n * args.steps / dt
        k_ms = sum(x[0] for x in tstats) / n_launches if n_launches else float("nan")
        alg = ALG_BYTES[task]
        # `achieved` / `frac` are CHIP-level and per env step: the algorithmic bytes of all S launches of a step (alg x envs per GPU) over the
        # wall time of a step on this GPU - comparable between --sub-batches 1 and 4 (the per-launch figure of a sub-batch over ITS duration fell with S although the
        # step got faster).  The per-launch figure stays next to it (achieved_per_launch / frac_per_launch: one launch of n_launch environments over the mean launch duration).
        ms_step = dt / args.steps * 1e3
        achieved = alg * n / (ms_step * 1e-3) / 1e9
        achieved_launch = alg * n_launch / (k_ms * 1e-3) / 1e9
        pm, pm_path = _pmc(task, policy, n, S)
        traffic, traffic_isolated, traffic_note, valu = None, None, None, None
        if pm is not None:
            traffic_isolated = _hbm_bytes(pm)
            if S == 1:
                traffic = traffic_isolated
                traffic_note = "%s: separate rocprofv3 --pmc passes of this command (not this run); one launch per step, alone on the chip in the timed run too" % pm_path
            else:
                # A counter pass serialises dispatches: each launch of a sub-batch was measured ALONE on the chip, its working set having the whole L2 to
                # itself.  In the timed run the S launches of a step are in flight together and share the L2: their combined working set is that of ONE launch
                # over all environments, which the --sub-batches 1 passes of the same command measured without serialisation artefacts (that launch is alone
                # in the timed run as well).  traffic = that figure / S (per launch of a sub-batch, under the concurrency of this run); traffic_isolated = the
                # serialised per-launch figure.
                pm1, pm1_path = _pmc(task, policy, n, 1)
                if pm1 is not None:
                    traffic = _hbm_bytes(pm1) / S
                    traffic_note = ("%s / %d: the %d launches of a step run concurrently and share the L2 - their combined working set is that of one launch over all %d "
                                    "environments, measured by the --sub-batches 1 counter passes of this command; traffic_isolated (%s): the serialised counter passes of "
                                    "this command, every launch of %d environments alone on the chip" % (pm1_path, S, S, n, pm_path, n_launch))
                else:
                    traffic_note = "%s holds the serialised per-launch figure only (traffic_isolated); no --sub-batches 1 pass of this command is committed" % pm_path
            if "SQ_INSTS_VALU" in pm and "SQ_WAVE_CYCLES" in pm:
                # binding resource: FP64 VALU issue.  Dynamic VALU instruction count from the PMC pass x the flop per VALU instruction COUNTED in the
                # kernel's ISA (tools/isa_fp64_mix.py -> profiles/<round>/isa_fp64_mix.json: FMA-class FP64 = 2 flop, other FP64 arithmetic = 1, the
                # rest 0; a static mix - loops and branches weight it differently at run time); lanes idle inside partially filled waves are counted
                # as if they worked (an upper estimate); peak = 78.6 TFLOP/s vector FP64 (1024 SIMDs)
                insts = pm["SQ_INSTS_VALU"]["mean_per_dispatch"]
                mix, mix_path = _isa_mix(task)
                if mix is not None:
                    tflops = insts * mix["flop_per_valu"] * 64 / (k_ms * 1e-3) / 1e12
                    valu = {"bound": "fp64_valu", "source": pm_path, "valu_insts_per_launch": insts, "flop_per_valu_inst": mix["flop_per_valu"],
                            "fp64_share_of_valu_insts": mix["fp64_share_of_valu"], "mix_source": mix_path + " (static ISA count)",
                            "achieved_tflops_est": tflops, "peak_tflops": FP64_VALU_PEAK_TFLOPS, "frac_est": tflops / FP64_VALU_PEAK_TFLOPS,
                            "valu_active_frac_of_wave_cycles": pm["SQ_ACTIVE_INST_VALU"]["mean_per_dispatch"] / pm["SQ_WAVE_CYCLES"]["mean_per_dispatch"]}
        tb = table.cpu().numpy()
        workload = {
            "avoiding": "Avoiding task, %d envs per GPU, random policy (Philox seed 42), state obs, 35 fused physics sub-steps per env step, "
                        "250-step episodes with auto-reset" % n,
            "pushing": "Pushing task, %d envs per GPU, the 60 reference test contexts tiled, %s, 35 fused physics sub-steps per env step, "
                       "400-step episodes with auto-reset" % (n, POLICY_TEXT[policy] % 10 if policy in ("mlp", "ddpm") else POLICY_TEXT[policy]),
            "sorting": "Sorting-4 task, %d envs per GPU, 60 contexts sampled like BlockContextManager.sample tiled, %s, 35 fused physics sub-steps per "
                       "env step, %d-step episodes with auto-reset" % (n, POLICY_TEXT[policy] % 16 if policy in ("mlp", "ddpm") else POLICY_TEXT[policy], max_steps),
            "inserting": "Inserting task (Gate_Insertion_Env; the reference defines no evaluation configuration for it), %d envs per GPU, 60 contexts sampled like "
                         "BlockContextManager.sample tiled, %s, 35 fused physics sub-steps per env step, %d-step episodes with auto-reset"
                         % (n, POLICY_TEXT[policy] % 13 if policy in ("mlp", "ddpm") else POLICY_TEXT[policy], max_steps),
            "aligning": "Aligning task, %d envs per GPU, the 60 reference test contexts tiled, %s, 35 fused physics sub-steps per env step, 400-step episodes with "
                        "auto-reset" % (n, POLICY_TEXT["scripted_align"] if policy == "scripted_align" else "ResidualMLP 20->128x6->3 (Mish) stand-in policy with fixed random weights (torch, f32)"),
            "stacking": "Stacking task, %d envs per GPU, the first %d of the reference's 100 test contexts tiled, %s, 30 fused physics sub-steps per env step, "
                        "1000-step episodes with auto-reset" % (n, len(ctx60) if ctx60 is not None else 0,
                                                                POLICY_TEXT[policy] if policy != "mlp" else "ResidualMLP 20->128x6->8 (Mish) stand-in policy with fixed random weights (torch, f32)"),
        }[task]
        line = {
            "metric": "env-steps/s", "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "envs_per_gpu": n, "sub_batches": S, "envs_per_launch": n_launch,
                       "sub_batch_note": ("the %d environments of a GPU are stepped as %d independent sub-batches of %d on %d HIP streams (own handle, policy state, "
                                          "tally; Philox counters and context ids by global environment index, so the work is that of one batch): a launch lasts as "
                                          "long as its slowest workgroup and sub-batches do not wait for each other's rare-path tails; --sub-batches 1 = one launch "
                                          "per step" % (n, S, n_launch, S)) if S > 1 else "one launch per step over the whole batch",
                       "n_substeps": n_sub, "parallelism": "env-shard x%d" % world, "policy": policy, "policy_captured_as_graph": bool(args.policy_graph),
                       "preroll_steps_untimed": preroll, "phase_mix": "steady state (staggered episode phases)" if preroll else "fresh reset",
                       "auto_reset": not args.no_auto_reset, "finite": finite, "flagged_envs": flagged,
                       "episodes_finished_all_ranks": int(tb[:, 0].sum()), "episodes_success_all_ranks": int(tb[:, 1].sum()),
                       "episodes_finished_rank0": int(episodes[0].item()),
                       "metric_reduction": reduction, "rccl_ranks": rccl_ranks,
                       "ms_per_step_by_rank": {"min": min(dt_by_rank) / args.steps * 1e3, "max": max(dt_by_rank) / args.steps * 1e3,
                                               "all": [x / args.steps * 1e3 for x in dt_by_rank]}},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "achieved_basis": "per env step, chip level: algorithmic_bytes_per_env_step x envs_per_gpu / ms_per_step (all sub-batch launches of a step, policy time included)",
                         "achieved_per_launch": achieved_launch, "frac_per_launch": achieved_launch / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_step": alg * n,
                         "traffic": traffic, "traffic_isolated": traffic_isolated, "traffic_source": traffic_note,
                         "kernel": ("k_avoiding_step_split<true, false>" if task == "avoiding" and (n_launch + 63) // 64 > (args.serve_max_wg if args.serve_max_wg is not None else 256 // S) else KERNEL[task]),
                         "kernel_ms": k_ms,
                         "kernel_launches_timed": n_launches,
                         "kernel_ms_min": min(x[1] for x in tstats) if n_launches else None, "kernel_ms_max": max(x[2] for x in tstats) if n_launches else None,
                         "algorithmic_bytes_per_launch": alg * n_launch, "algorithmic_bytes_per_env_step": alg,
                         "implementation_bytes_per_launch": IMPL_BYTES[task] * n_launch, "envs_per_launch": n_launch,
                         "note": "the path is FP64-VALU issue/latency bound, not HBM bound: < 2.2 KB of HBM per env step with all 35 sub-steps "
                                 "fused in registers / LDS (DESIGN.md sections 4, 12.3, 13.3); `frac` is the (structurally tiny) HBM fraction the contract asks for, "
                                 "`valu` the binding resource",
                         "valu": valu},
        }
        if policy == "beso":
            line["policy_roofline"] = _beso_policy_roofline(pol, n, dev)
        if policy == "ddpm":
            pr = _ddpm_policy_roofline(pol, n // S, dev)
            if pr is not None:
                line["policy_roofline"] = pr
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(task, env.blob, q, ctx60)
        print(json.dumps(line))
    for sh in shards:
        sh.env.close()          # (the library communicator is process-wide: distributed.auto_comm)
    if world > 1:
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="avoiding", choices=["avoiding", "pushing", "sorting", "stacking", "aligning", "inserting"], help="avoiding = the headline configuration (BASELINE configs[1])")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--policy", default=None, choices=["random", "mlp", "scripted_push", "scripted_stack", "scripted_align", "ddpm", "beso"],
                    help="default: random (Avoiding), mlp (Pushing / Sorting), scripted_stack (Stacking: pick-and-place, the contact regime of the task)")
    ap.add_argument("--max-steps", type=int, default=None, help="Sorting: episode cap (default 700 = configs/sorting_4_config.yaml:80; the Sim class default is 500)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stack-contexts", type=int, default=16, help="Stacking: number of reference test contexts in the tile (host IK of the scripted policy: ~1.3 s each)")
    ap.add_argument("--no-auto-reset", action="store_true")
    ap.add_argument("--no-preroll", action="store_true", help="measure from a freshly reset batch (round-1 behaviour)")
    ap.add_argument("--preroll", type=int, default=None, help="untimed steady-state pre-roll steps (default: one episode length)")
    ap.add_argument("--solver-strict", action="store_true", help="contact solvers iterate to round-off like the oracle (parity A/B)")
    ap.add_argument("--lanes", type=int, default=None, help="environments per wave (default 64)")
    ap.add_argument("--split", type=int, default=None, help="1: two-wave controller||physics kernel, 0: fused kernel, default auto")
    ap.add_argument("--serve-max-wg", type=int, default=None, help="Avoiding: workgroup count up to which the split kernel runs with its third wave (rare constraint paths); 0 = the two-wave kernel (A/B)")
    ap.add_argument("--sub-batches", type=int, default=None, help="step the GPU's environments as this many independent sub-batches on as many HIP streams "
                    "(default: 4 for avoiding / pushing / sorting / inserting, 1 otherwise; 1 = one launch per step over the whole batch)")
    ap.add_argument("--fuse-tail", type=int, default=1, help="Avoiding, random policy: everything between two step launches (mask, tally, auto-reset, the next action) in one kernel")
    ap.add_argument("--graph-rollout", type=int, default=0, help="Avoiding, random policy, sub-batches on their own streams: the rollout step as one captured HIP graph launch (0: eight runtime calls per step)")
    ap.add_argument("--policy-graph", type=int, default=None, help="mlp / ddpm policy of the push tasks: the whole predict chain of a sub-batch as one captured HIP graph (policies.CapturedPolicy; default: on for ddpm)")
    ap.add_argument("--lds-pad", type=int, default=None, help="override the LDS bytes requested per workgroup (placement control)")
    args = ap.parse_args()
    # sub-batches run on their own HIP streams; with the runtime's default of four hardware queues two of four streams share a queue (the null stream
    # holds one) and their launches serialise: measured 3.6 M instead of 7.2 M env-steps/s.  Must be set before the HIP runtime starts.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_self_spawn(args.gpus))

    import torch
    if not torch.cuda.is_available():
        print("bench.py needs a HIP device (there is no CPU fallback for the rollout path)", file=sys.stderr)
        sys.exit(2)
    run(args)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        _cpu_worker_main(sys.argv[2:])
    else:
        main()
