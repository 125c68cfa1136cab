"""Full-size, full-episode health check (VERDICT r1 next #1b): 4096 environments on every BASELINE context run whole episodes with
the stand-in MLP policy and with the scripted pushing policy (contact regime); the divergence flags - solver failure, contact-table
overflow, a cube leaving the modelled part of the table (the pairs the kernel does not evaluate) - must never
fire, and the state must stay finite.  This is what makes "the kernel's pair set is a subset of the oracle's" harmless on the
evaluation contexts: the excluded pairs (cube <-> frame beams / finger tips) are never approached."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "ref_offline_ik.npz"))
BAD = (1 << 16) | (1 << 18) | (1 << 19)


def _run(env, pol, steps):
    n = env.n_envs
    env.policy_begin()
    actions = torch.zeros(n, 7, dtype=torch.float64, device=env.device)
    actions[:, 3:] = torch.tensor([0.0, 1.0, 0.0, 0.0], dtype=torch.float64, device=env.device)
    des_xy, des_z = env.policy_des[:2, :n], env.policy_des[2, :n]
    seen = torch.zeros(n, dtype=torch.int32, device=env.device)
    counts = torch.zeros(2, dtype=torch.int64, device=env.device)
    for t in range(steps):
        if hasattr(pol, "begin_episodes"):
            pol.begin_episodes(env.last_reset)
        obs_in = torch.cat((des_xy.t(), env.obs.to(torch.float64)), dim=1)
        des_xy.add_(pol.predict_batch(obs_in).to(torch.float64).t())
        actions[:, 0:2] = des_xy.t()
        actions[:, 2] = des_z
        env.step(actions)
        seen |= env.flags[:n] & BAD             # flags are per episode: collect them before the auto-reset clears them
        env.auto_reset(counts)
    torch.cuda.synchronize()
    st, fl, sc = env.get_state()
    return seen.cpu().numpy(), st, counts.cpu().numpy()


@pytest.mark.parametrize("policy", ["mlp", "scripted_push"])
def test_pushing_4096_envs_all_60_contexts_full_episode(policy):
    from d3il_amd.agents import RandomResidualMLPPolicy, ScriptedPushPolicy
    from d3il_amd.envs.pushing import BlockPushVecEnv
    ctx60 = np.load(os.path.join(ROOT, "d3il_amd", "data", "pushing_test_contexts.npy"))
    n = 4096
    env = BlockPushVecEnv(n, device=0)
    env.set_init_qpos(G["avoiding__traj_last"].copy())
    env.reset(context=ctx60[np.arange(n) % 60])
    pol = RandomResidualMLPPolicy(input_dim=10, device=env.device) if policy == "mlp" else ScriptedPushPolicy("pushing", device=env.device)
    seen, st, counts = _run(env, pol, 401)
    assert not seen.any(), "flagged envs: %d (bits %s)" % (int((seen != 0).sum()), hex(int(np.bitwise_or.reduce(seen))))
    assert np.isfinite(st[:68]).all() and counts[0] >= n
    env.close()


@pytest.mark.parametrize("policy", ["mlp", "scripted_push"])
def test_sorting_4096_envs_full_episode(policy):
    from d3il_amd.agents import RandomResidualMLPPolicy, ScriptedPushPolicy
    from d3il_amd.envs.sorting import SortingVecEnv, sample_contexts
    n = 4096
    env = SortingVecEnv(n, device=0, max_steps_per_episode=300)
    env.set_init_qpos(G["sorting__traj_last"].copy())
    env.reset(context=sample_contexts(60, 4, seed=0)[np.arange(n) % 60])
    pol = RandomResidualMLPPolicy(input_dim=16, device=env.device) if policy == "mlp" else ScriptedPushPolicy("sorting", device=env.device)
    seen, st, counts = _run(env, pol, 301)
    assert not seen.any(), "flagged envs: %d (bits %s)" % (int((seen != 0).sum()), hex(int(np.bitwise_or.reduce(seen))))
    assert np.isfinite(st[:94]).all() and counts[0] >= n
    env.close()


@pytest.mark.parametrize("policy", ["mlp", "scripted_push"])
def test_inserting_4096_envs_full_episode(policy):
    """Inserting: 4096 environments, 60 sampled contexts, 300-step episodes with device auto-reset and tally; with the stand-in MLP the rods wander into
    cubes and walls, the scripted policy pushes the cubes towards their gates.  No solver failure, contact overflow (incl. the three rod <-> wall
    slots) or off-table flag; the tally row of every context adds up to its finished episodes."""
    from d3il_amd.agents import RandomResidualMLPPolicy, ScriptedPushPolicy
    from d3il_amd.envs.inserting import GateInsertionVecEnv, sample_contexts
    n = 4096
    env = GateInsertionVecEnv(n, device=0, max_steps_per_episode=300)
    env.set_init_qpos(G["avoiding__traj_last"].copy())
    ids = np.arange(n) % 60
    env.reset(context=sample_contexts(60, seed=0)[ids])
    table = env.set_tally(60, torch.as_tensor(ids, dtype=torch.int32, device=env.device))
    pol = RandomResidualMLPPolicy(input_dim=13, device=env.device) if policy == "mlp" else ScriptedPushPolicy("inserting", device=env.device)
    seen, st, counts = _run(env, pol, 301)
    assert not seen.any(), "flagged envs: %d (bits %s)" % (int((seen != 0).sum()), hex(int(np.bitwise_or.reduce(seen))))
    assert np.isfinite(st[:81]).all() and counts[0] >= n
    tb = table.cpu().numpy()
    assert tb[:, 0].sum() == counts[0] and tb[:, 1].sum() == counts[1]
    # every finished episode is also counted by its code (mode_dict code | letters << 3) in the upper half of the row
    from d3il_amd import capi
    assert tb[:, 2 + capi.TALLY_ALL:2 + capi.TALLY_ALL + 256].sum() == counts[0]
    env.close()
