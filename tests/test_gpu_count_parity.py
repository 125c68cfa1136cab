"""Integer-count parity of the contact tasks over WHOLE episodes (VERDICT r2, next #1a).

north_star: "success flags, behaviour-mode histograms bit-exact as integer counts".  Trajectory equality cannot be the long-horizon
criterion for the contact tasks (two correct f64 implementations separate on a chaotic contact system), so this
file compares what the evaluation actually reports: for every BASELINE context the closed-loop scripted policy runs a full episode on
the device (through the batched Sim class = the C ABI) AND on the CPU oracle (tests/oracle_episodes.py: the same policy object on a
batch of one), and the per-context (success, mode code) tables - and the metric count tables built from them - are compared.

What is asserted (round 4).  `tests/golden/oracle_outcome_sets.json` (made by tools/oracle_sensitivity.py on the CPU oracle
alone) holds, per context, the set of (success, mode) outcomes the ORACLE ITSELF reaches when the context's cube positions are moved by
multiples of 1e-12 m (Sorting: 1e-10 m) - far below the f32 observations the policy sees and below the one-step agreement of any two f64
implementations of the soft-contact step.  A context with ONE outcome is *decided* at f64 resolution: there the device's (success, mode)
must be IDENTICAL to the oracle's - no bound, no tolerance.  A context with several outcomes is one on which the oracle does not agree with
itself; there no second implementation can be asked to agree with it (all of them are episodes of the scripted plan in which the cubes
cross paths): the device's outcome must then be ONE OF the oracle's own outcomes (`device_outside_oracle_set == []`), and the success counts
may differ by at most the number of such contexts.  Stacking: every context identical.
tools/gpu_count_onset.py / gpu_count_onset_sorting.py show, for the differing contexts, that the oracle restarted from the device's state
reproduces the device's next state at the conditioning level of the solve at EVERY step of the episode (profiles/r04/onset_*.json).
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
ORDERS = ((0, 1, 2), (1, 0, 2))          # stacking orders of the scripted pick-and-place: red, green, blue / green, red, blue


def _dump(name, obj):
    try:
        os.makedirs(OUT, exist_ok=True)
        with open(os.path.join(OUT, name), "w") as f:
            json.dump(obj, f, indent=1)
    except OSError:
        pass


def _outcome_sets(task):
    with open(os.path.join(ROOT, "tests", "golden", "oracle_outcome_sets.json")) as f:
        fx = json.load(f)[task]
    return {int(i): [tuple(o) for o in outs] for i, outs in fx["outcomes"].items()}, fx


def _compare(task, dev_rows, orc_rows, sets=None, max_undecided=None):
    """rows: list of (success, mode) per context.  `sets` = the oracle's own outcome sets (None: every context must be identical).
    Asserts identity on every decided context; prints the others; returns the summary that is also written to gpurun_out/count_parity_<task>.json."""
    n = len(dev_rows)
    diff = [i for i, (a, b) in enumerate(zip(dev_rows, orc_rows)) if tuple(a) != tuple(b)]
    undecided = [] if sets is None else [i for i in range(n) if len(sets[i]) > 1]
    decided = [i for i in range(n) if i not in undecided]
    bad = [i for i in diff if i in decided]
    summary = dict(task=task, contexts=n, identical=n - len(diff), differing=diff, decided=len(decided), undecided=undecided, differing_decided=bad,
                   device=[list(map(str, r)) for r in dev_rows], oracle=[list(map(str, r)) for r in orc_rows],
                   device_outside_oracle_set=[i for i in undecided if (bool(dev_rows[i][0]), int(dev_rows[i][1])) not in sets[i]],
                   device_successes=int(sum(bool(r[0]) for r in dev_rows)), oracle_successes=int(sum(bool(r[0]) for r in orc_rows)))
    _dump("count_parity_%s.json" % task, summary)
    print("\n%s: %d of %d contexts with identical (success, mode) - all %d decided contexts must be; successes device %d / oracle %d" %
          (task, summary["identical"], n, len(decided), summary["device_successes"], summary["oracle_successes"]))
    for i in undecided:
        print("  context %3d (the oracle's own outcomes under 1e-12 m perturbations: %s): device %s   oracle %s" % (i, sets[i], dev_rows[i], orc_rows[i]))
    for i in bad:
        print("  DECIDED context %3d differs: device %s   oracle %s" % (i, dev_rows[i], orc_rows[i]))
    assert not bad, "%s: the device differs from the oracle on decided contexts %s" % (task, bad)
    # the undecided contexts are not free: the device's outcome must be one the oracle itself reaches under the perturbations, and
    # the number of successes may differ from the oracle's by no more than the undecided contexts on which the two differ
    assert not summary["device_outside_oracle_set"], "%s: device outcome outside the oracle's own outcome set on contexts %s" % (task, summary["device_outside_oracle_set"])
    assert abs(summary["device_successes"] - summary["oracle_successes"]) <= len([i for i in diff if i in undecided])
    if sets is not None:
        # the fixture must stay meaningful: the live oracle episode of a decided context is the fixture's outcome, and most contexts are decided
        stale = [i for i in decided if (bool(orc_rows[i][0]), int(orc_rows[i][1])) != sets[i][0]]
        assert not stale, "%s: tests/golden/oracle_outcome_sets.json is stale for contexts %s (regenerate with tools/oracle_sensitivity.py)" % (task, stale)
        assert len(undecided) <= max_undecided, (len(undecided), max_undecided)
    return summary


def test_pushing_success_and_mode_tables_over_full_episodes():
    """Pushing, the 60 reference test contexts, 400-step episodes, scripted two-cube pushing policy that finishes the task in the behaviour mode `context % 4` (pushing_sim.py:61-83, 140-167)."""
    from d3il_amd.agents import ScriptedGoalPushPolicy
    from d3il_amd.simulation.pushing_sim import Pushing_Sim, load_test_contexts
    from tests import oracle_episodes as oe
    ctx = load_test_contexts()
    sim = Pushing_Sim(seed=0, device="cuda:0", render=False, n_cores=1, n_contexts=60, n_trajectories_per_context=1, max_steps_per_episode=400)
    sim.test_agent(ScriptedGoalPushPolicy("pushing", plan=np.arange(60) % 4, device="cuda:0"))
    r = sim.last_rollout
    assert not (r["flags"].cpu().numpy() & ((1 << 16) | (1 << 18) | (1 << 19))).any()
    dev_rows = list(zip(r["success"].cpu().numpy().astype(bool).tolist(), r["mode"].cpu().numpy().tolist()))
    from d3il_amd.envs.pushing import BlockPushVecEnv
    env = BlockPushVecEnv(1, device=0)
    q0 = env.start()[0]
    env.close()
    res = oe.run_many(oe.pushing_episode, [(i, ctx[i], q0, 400, i % 4) for i in range(60)])
    orc_rows = [(s, m) for _, s, m, _, _ in res]
    sets, _ = _outcome_sets("pushing")
    s = _compare("pushing", dev_rows, orc_rows, sets, max_undecided=7)      # exactly the fixture's undecided contexts (6, 10, 22, 26, 30, 34, 54): the set may not grow unnoticed
    # the metric's integer table (mode counts of the successful rollouts per context): the rows of the decided contexts are identical
    tab = np.zeros((60, 4), dtype=np.int64)
    for i, (ok, m) in enumerate(orc_rows):
        if ok and m >= 0:
            tab[i, m] += 1
    dec = [i for i in range(60) if i not in s["undecided"]]
    assert np.array_equal(tab[dec], r["counts"][:-1].reshape(60, 4)[dec])
    if not s["differing"]:
        assert int(r["counts"][-1]) == s["oracle_successes"]


def test_pushing_tables_on_sampled_contexts():
    """The same comparison on 120 contexts drawn like BlockContextManager.sample (seed 3) instead of the reference's 60 test contexts."""
    from d3il_amd.agents import ScriptedGoalPushPolicy
    from d3il_amd.envs.pushing import BlockPushVecEnv, sample_contexts
    from d3il_amd.simulation.pushing_sim import Pushing_Sim
    from tests import oracle_episodes as oe
    n = 120
    ctx = sample_contexts(n, seed=3)
    sim = Pushing_Sim(seed=0, device="cuda:0", render=False, n_cores=1, n_contexts=n, n_trajectories_per_context=1, max_steps_per_episode=400, contexts=ctx)
    sim.test_agent(ScriptedGoalPushPolicy("pushing", plan=np.arange(n) % 4, device="cuda:0"))
    r = sim.last_rollout
    assert not (r["flags"].cpu().numpy() & ((1 << 16) | (1 << 18) | (1 << 19))).any()
    dev_rows = list(zip(r["success"].cpu().numpy().astype(bool).tolist(), r["mode"].cpu().numpy().tolist()))
    env = BlockPushVecEnv(1, device=0)
    q0 = env.start()[0]
    env.close()
    res = oe.run_many(oe.pushing_episode, [(i, ctx[i], q0, 400, i % 4) for i in range(n)])
    orc_rows = [(s, m) for _, s, m, _, _ in res]
    sets, _ = _outcome_sets("pushing_sampled")
    s = _compare("pushing_sampled", dev_rows, orc_rows, sets, max_undecided=12)      # exactly the fixture's undecided contexts
    assert s["oracle_successes"] >= n // 2


def test_sorting_success_and_mode_tables_over_full_episodes():
    """Sorting-4, 60 contexts sampled like BlockContextManager.sample (seed 0: bench.py's tile), 700-step episodes
    (configs/sorting_4_config.yaml:80), scripted push-over-the-edge policy (sorting_sim.py:118-133, 191-208)."""
    from d3il_amd.agents import ScriptedGoalPushPolicy
    from d3il_amd.envs.sorting import SortingVecEnv, sample_contexts
    from d3il_amd.simulation.sorting_sim import Sorting_Sim
    from tests import oracle_episodes as oe
    ctx = sample_contexts(60, 4, seed=0)
    sim = Sorting_Sim(seed=0, device="cuda:0", render=False, n_cores=1, n_contexts=60, n_trajectories_per_context=1, max_steps_per_episode=700, contexts=ctx)
    sim.test_agent(ScriptedGoalPushPolicy("sorting", device="cuda:0"))
    r = sim.last_rollout
    fl = r["flags"].cpu().numpy()
    assert not (fl & ((1 << 16) | (1 << 18))).any()
    dev_rows = list(zip(r["success"].cpu().numpy().astype(bool).tolist(), r["mode"].cpu().numpy().tolist()))
    env = SortingVecEnv(1, device=0)
    q0 = env.start()[0]
    env.close()
    res = oe.run_many(oe.sorting_episode, [(i, ctx[i], q0, 700) for i in range(60)])
    orc_rows = [(s, m) for _, s, m, _ in res]
    sets, _ = _outcome_sets("sorting")
    s = _compare("sorting", dev_rows, orc_rows, sets, max_undecided=3)      # exactly the fixture's undecided contexts (34, 45, 52)
    hist_d = np.bincount(np.array([m for _, m in dev_rows]), minlength=256)
    hist_o = np.bincount(np.array([m for _, m in orc_rows]), minlength=256)
    print("  mode-code histogram L1 distance: %d of %d rollouts" % (int(np.abs(hist_d - hist_o).sum()) // 2, 60))


def test_stacking_success_and_mode_tables_over_full_episodes():
    """Stacking, all 100 of the reference's test contexts (bench.py tiles the first 16), 1000-step episodes (configs/stacking_config.yaml:84),
    scripted three-box pick-and-place (stacking_sim.py:88-136: order string, success, 1- / 2-box successes)."""
    from d3il_amd.agents import ScriptedStackPolicy
    from d3il_amd.controllers.scripted_stacking import build_trajectory
    from d3il_amd.envs.stacking import CubeStackingVecEnv, load_test_contexts, mode_string
    from d3il_amd.model import blob
    from d3il_amd.simulation.stacking_sim import Stacking_Sim
    from tests import oracle_episodes as oe
    nctx = 100
    ctx = load_test_contexts()[:nctx]
    js = blob.load_json("stacking")
    env = CubeStackingVecEnv(1, device=0)
    q0 = env.start()[0]
    env.close()
    tables = [build_trajectory(js, q0, c, order=ORDERS[i % len(ORDERS)], speed=0.5) for i, c in enumerate(ctx)]
    sim = Stacking_Sim(seed=0, device="cuda:0", render=False, n_cores=1, n_contexts=nctx, n_trajectories_per_context=1, max_steps_per_episode=1000, contexts=ctx)
    sim.test_agent(ScriptedStackPolicy(tables, torch.arange(nctx), device="cuda:0"))
    r = sim.last_rollout
    fl = r["flags"].cpu().numpy()
    assert not (fl & ((1 << 16) | (1 << 18))).any()
    dev_rows = [(bool(s), mode_string(int(m))) for s, m in zip(r["success"].cpu().numpy(), r["mode"].cpu().numpy())]
    res = oe.run_many(oe.stacking_episode, [(i, ctx[i], q0, 1000, tables[i]) for i in range(nctx)])
    orc_rows = [(s, m) for _, s, m, _ in res]
    s = _compare("stacking", dev_rows, orc_rows)          # every context identical: (success, order string) tables bit-exact
    assert s["oracle_successes"] >= nctx // 2, "the scripted pick-and-place should stack three boxes on most contexts"
    assert s["device_successes"] == s["oracle_successes"]


def test_inserting_letter_tables_over_scripted_gate_pushes():
    """Inserting (gate_insertion.py:386-432: the `modes` letters, the mode code, success) over whole scripted episodes, every context compared with the oracle
    - the task has no Sim class in the reference, so the env protocol is the boundary (VERDICT r4 missing #7).  32 contexts: the red cube in the mouth of
    its gate with small position / yaw variations, pushed west into the goal by an open-loop waypoint walk (the scenario of
    test_gpu_parity_inserting.test_physically_produced_insertions_match_oracle); two contexts push the green cube north instead.
    The table (letters, code, success) must be IDENTICAL on every context on which the oracle agrees with itself under K = 4 perturbations of 1e-12 m
    (fixed beforehand; the rule of the other tasks), and inside the oracle's own outcome set on the others."""
    from d3il_amd.envs.inserting import GateInsertionVecEnv
    from tests import oracle_episodes as oe
    n, steps, K = 32, 170, 4
    rng = np.random.default_rng(11)
    park = np.array([[0.40, -0.17, 0, 1, 0, 0, 0], [0.62, -0.08, 0, 1, 0, 0, 0], [0.45, 0.02, 0, 1, 0, 0, 0]], float)
    ctx = np.tile(park[None], (n, 1, 1))
    way = np.zeros((n, 3, 2))
    way[:] = np.array([(0.49, 0.10), (0.49, 0.276), (0.388, 0.276)])
    for e in range(n):
        if e % 16 == 15:      # the green cube from the south into its gate
            ctx[e, 1, :2] = [0.525 + rng.uniform(-0.001, 0.001), 0.36]
            ctx[e, 2, :2] = [0.40, 0.02]
            way[e] = np.array([(0.525, 0.30), (0.525, 0.425), (0.525, 0.425)])
        else:
            yaw = rng.uniform(-0.03, 0.03)
            ctx[e, 0, :2] = [0.41 + 0.0005 * e, 0.276 + rng.uniform(-0.001, 0.001)]
            ctx[e, 0, 3:] = [np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)]
    env = GateInsertionVecEnv(n, device=0)
    q0 = env.start()[0]
    env.reset(context=ctx.reshape(n, 21))
    z = env.robot_state()[:, 2:3].clone()
    des = env.obs[:, :2].to(torch.float64).clone()
    wayt = torch.as_tensor(way, dtype=torch.float64, device=des.device)
    wi = torch.zeros(n, dtype=torch.long, device=des.device)
    ar = torch.arange(n, device=des.device)
    quat = torch.tensor([0.0, 1, 0, 0], dtype=torch.float64, device=des.device).expand(n, 4)
    first_dev = np.full(n, -1)
    for t in range(steps):
        d = wayt[ar, wi] - des
        nn = d.norm(dim=1, keepdim=True)
        wi = torch.where((nn[:, 0] < 1e-9) & (wi < 2), wi + 1, wi)
        d = wayt[ar, wi] - des
        nn = d.norm(dim=1, keepdim=True)
        des = des + d / nn.clamp_min(1e-12) * torch.minimum(nn, torch.full_like(nn, 0.006))
        obs, rew, done, info = env.step(torch.cat([des, z, quat], 1).contiguous())
        nm = (env.mode.to(torch.int32) >> 3).cpu().numpy()
        first_dev = np.where((first_dev < 0) & (nm > 0), t, first_dev)
    torch.cuda.synchronize()
    assert not (env.flags[:n].cpu().numpy() & ((1 << 16) | (1 << 18) | (1 << 19))).any()
    code = env.mode.to(torch.int32).cpu().numpy()
    dev_rows = [(int(code[e] >> 3), int(code[e] & 7), bool(env.success[e])) for e in range(n)]
    env.close()
    jobs = []
    for e in range(n):
        for k in range(K + 1):
            c = ctx[e].copy()
            c[0 if e % 16 != 15 else 1, 0] += k * 1e-12
            jobs.append((e * 100 + k, c, q0, way[e], steps))
    res = {i: (nl, cd, su, fs) for i, nl, cd, su, fs in oe.run_many(oe.inserting_episode, jobs)}
    orc_rows = [res[e * 100][:3] for e in range(n)]
    sets = {e: sorted({res[e * 100 + k][:3] for k in range(K + 1)}) for e in range(n)}
    undecided = [e for e in range(n) if len(sets[e]) > 1]
    diff = [e for e in range(n) if dev_rows[e] != orc_rows[e]]
    summary = dict(task="inserting", contexts=n, identical=n - len(diff), differing=diff, undecided=undecided, K=K,
                   device=[list(map(int, r)) for r in dev_rows], oracle=[list(map(int, r)) for r in orc_rows],
                   first_letter_step_device=first_dev.tolist(), first_letter_step_oracle=[res[e * 100][3] for e in range(n)],
                   device_outside_oracle_set=[e for e in undecided if dev_rows[e] not in sets[e]])
    _dump("count_parity_inserting.json", summary)
    print("\ninserting: %d of %d contexts with identical (letters, code, success); undecided under K = %d perturbations: %s; differing: %s" % (n - len(diff), n, K, undecided, diff))
    assert not [e for e in diff if e not in undecided], summary
    assert not summary["device_outside_oracle_set"], summary
    assert len(undecided) <= 4
    # the letters appear at the oracle's step on the decided contexts, and most of the batch does insert
    dec = [e for e in range(n) if e not in undecided]
    assert [int(first_dev[e]) for e in dec] == [res[e * 100][3] for e in dec]
    assert sum(1 for r in dev_rows if r[0] >= 1) >= 20
