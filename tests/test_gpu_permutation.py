"""Results may not depend on where an environment sits in the batch (workgroup, lane, workgroup mates): every environment gets its own context and
is driven either open loop (Avoiding: its own random set-point walk, biased towards the obstacles) or by a closed-loop scripted policy that reads
only its own observation (Pushing / Sorting: rod pushes, cube <-> cube, cube <-> wall contacts); the batch runs twice, the second time PERMUTED,
and each environment's state must be bit-identical in both runs at every step.  This is the criterion that exposed the code-generation defect of
the Stacking kernel in round 3 (the Stacking test is test_gpu_parity_stacking.py::test_permuted_batch_*); long forms:
tools/gpu_perm_push_sort.py, tools/gpu_stack_perm.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
SOLVER_FAIL = 1 << 16


def _compare(runs, perm, steps):
    for t in range(steps):
        d = np.nonzero((runs[0][t] != runs[1][t]).any(axis=0))[0]
        assert d.size == 0, "step %d: environments %s differ between the two arrangements (slots %s / %s)" % (t, d[:6].tolist(), d[:6].tolist(), np.argsort(perm)[d[:6]].tolist())


def test_avoiding_permuted_batch():
    from d3il_amd.envs.avoiding import ObstacleAvoidanceVecEnv
    n, steps = 2048, 150
    rng = np.random.default_rng(3)
    delta = rng.uniform(-0.01, 0.01, size=(steps, n, 2)) + np.array([0.0, 0.004])      # drift towards the obstacle rows: rod contacts, mode bits
    perm = rng.permutation(n)
    runs = []
    for order in (np.arange(n), perm):
        env = ObstacleAvoidanceVecEnv(n, device=0)
        env.start()
        env.reset()
        rs = env.robot_state()
        des, z = rs[:, :2].clone(), rs[:, 2:3].clone()
        quat = torch.tensor([0.0, 1.0, 0.0, 0.0], dtype=torch.float64, device=des.device).expand(n, 4)
        inv = np.argsort(order)
        out = []
        for t in range(steps):
            des = des + torch.as_tensor(delta[t][order], dtype=torch.float64, device=des.device)
            env.step(torch.cat((des, z, quat), dim=1).contiguous())
            torch.cuda.synchronize()
            st, fl, _ = env.get_state()
            assert not (fl & SOLVER_FAIL).any()
            out.append(np.concatenate([st[:, inv], fl[None, inv].astype(np.float64)], axis=0))
        runs.append(out)
        env.close()
    _compare(runs, perm, steps)
    assert (runs[0][-1][-1].astype(np.int64) & (1 << 14)).any() or (runs[0][-1][-1].astype(np.int64) & (1 << 12)).any(), "no environment reached an obstacle"


@pytest.mark.parametrize("task,steps", [("pushing", 220), ("sorting", 300), ("inserting", 260)])
def test_contact_tasks_permuted_batch(task, steps):
    from d3il_amd.agents import ScriptedGoalPushPolicy, ScriptedPushPolicy
    n = 1024
    rng = np.random.default_rng(5)
    if task == "inserting":      # cube <-> wall, rod <-> cube and rod <-> wall contacts (the gates)
        from d3il_amd.envs.inserting import GateInsertionVecEnv as Env, sample_contexts
        ctx, plan = sample_contexts(n, seed=5), None
    elif task == "pushing":
        from d3il_amd.envs.pushing import BlockPushVecEnv as Env, sample_contexts
        ctx, plan = sample_contexts(n, seed=5), rng.integers(0, 4, size=n)
    else:
        from d3il_amd.envs.sorting import SortingVecEnv as Env, sample_contexts
        ctx, plan = sample_contexts(n, 4, seed=5).reshape(n, -1), None
    perm = rng.permutation(n)
    runs = []
    for order in (np.arange(n), perm):
        env = Env(n, device=0)
        env.start()
        obs = env.reset(random=False, context=ctx[order])
        dev = obs.device
        pol = ScriptedPushPolicy(task, device=dev) if task == "inserting" else ScriptedGoalPushPolicy(task, plan=None if plan is None else plan[order], device=dev)
        rs = env.robot_state()
        des, z = rs[:, :2].clone(), rs[:, 2:3].clone()
        quat = torch.tensor([0.0, 1.0, 0.0, 0.0], dtype=torch.float64, device=dev).expand(n, 4)
        inv = np.argsort(order)
        out = []
        for t in range(steps):
            des = des + pol.predict_batch(torch.cat((des, obs.to(torch.float64)), dim=1))
            obs, _, done, info = env.step(torch.cat((des, z, quat), dim=1).contiguous())
            torch.cuda.synchronize()
            st, fl, _ = env.get_state()
            assert not (fl & SOLVER_FAIL).any()
            out.append(st[:, inv].copy())
        runs.append(out)
        env.close()
    _compare(runs, perm, steps)
    moved = (np.abs(runs[0][-1][42:44] - runs[0][0][42:44]).max(axis=0) > 1e-3).sum()
    assert moved > n // 2, "the scripted policy has to reach the cubes (%d of %d moved)" % (moved, n)
