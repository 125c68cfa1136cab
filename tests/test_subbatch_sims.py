"""Sub-batches in the product (envs/sub_batch.py, the Sim classes' ``n_sub_batches``): the reference's n_cores process fan-out
(simulation/avoiding_sim.py:87-124, sorting_sim.py:160-189) as independent environment batches on their own HIP streams.

CPU: the partition and the agent fork.  GPU: ``Avoiding_Sim`` / ``Sorting_Sim`` / ``Pushing_Sim`` / ``Stacking_Sim`` / ``Aligning_Sim`` with four sub-batches return the integer tables of one
batch, bit for bit (rollouts are independent; the policies used here compute every row from that row alone, so nothing depends on which rows share a
batch)."""
import collections

import numpy as np
import pytest

torch = pytest.importorskip("torch")


def test_plan_partitions_contiguously():
    from d3il_amd.envs.sub_batch import plan
    assert plan(4096, 4) == [(0, 1024), (1024, 1024), (2048, 1024), (3072, 1024)]
    assert plan(300, 4) == [(0, 75), (75, 75), (150, 75), (225, 75)]
    assert plan(301, 4) == [(0, 76), (76, 75), (151, 75), (226, 75)]
    assert plan(130, 4) == [(0, 65), (65, 65)]           # every sub-batch at least one wavefront of environments
    assert plan(63, 4) == [(0, 63)] and plan(1, 1) == [(0, 1)] and plan(1000, 1) == [(0, 1000)]
    for n in (64, 257, 1000, 4097):
        for s in (1, 2, 3, 4, 8):
            p = plan(n, s)
            assert p[0][0] == 0 and sum(c for _, c in p) == n and all(p[i][0] + p[i][1] == p[i + 1][0] for i in range(len(p) - 1))


def test_fork_agent_shares_weights_and_copies_episode_state():
    from d3il_amd.envs.sub_batch import fork_agent

    class Agent:
        def __init__(self):
            self.model = torch.nn.Linear(4, 2)
            self.window = collections.deque(maxlen=3)
            self.hist = torch.zeros(5)
            self.scale = 2.0

        def reset(self):
            self.window.clear()

        def predict_batch(self, o):
            return o[:, :2]

    a = Agent()
    a.window.append(1)
    b = fork_agent(a)
    assert b.model is a.model and b.scale == a.scale
    assert b.window is not a.window and list(b.window) == [1] and b.hist is not a.hist
    b.window.append(2); b.hist += 1
    assert list(a.window) == [1] and float(a.hist.sum()) == 0.0

    class Forking(Agent):
        def fork(self):
            return "mine"
    assert fork_agent(Forking()) == "mine"


class RowPolicy:
    """delta = 6 mm towards a goal that depends on the row's own observation and on WHICH rollout the row is (set_rollout_range)."""

    def reset(self):
        pass

    def set_rollout_range(self, offset, count):
        self.offset = offset

    @torch.no_grad()
    def predict_batch(self, obs_in):
        o = obs_in.to(torch.float64)
        des, tcp = o[:, :2], o[:, 2:4]
        k = self.offset + torch.arange(o.shape[0], device=o.device, dtype=torch.float64)
        goal = torch.stack((0.5 + 0.15 * torch.sin(0.37 * k + 40.0 * tcp[:, 1]), tcp[:, 1] + 0.05), dim=1)
        d = goal - des
        return d / d.norm(dim=1, keepdim=True).clamp_min(1e-9) * 0.006


@pytest.mark.gpu
def test_avoiding_sim_sub_batches_identical_tables():
    from d3il_amd.simulation.avoiding_sim import Avoiding_Sim
    out = []
    for S in (1, 4):
        sim = Avoiding_Sim(seed=0, device="cuda:0", render=False, n_cores=1, n_trajectories=320, max_steps_per_episode=120, n_sub_batches=S)
        succ, ent = sim.test_agent(RowPolicy())
        r = sim.last_rollout
        out.append((succ.cpu().numpy(), ent, r["counts"].copy(), r["mode_code"].cpu().numpy(), r["n_pos"].cpu().numpy(), r["c_pos"].cpu().numpy()))
    a, b = out
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[0], b[0]) and a[1] == b[1]
    assert np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5])          # the logged TCP paths too: the sub-batches computed the same rollouts
    assert len(np.unique(a[3])) > 1                                           # and the rollouts are not all alike


@pytest.mark.gpu
def test_sorting_and_pushing_sim_sub_batches_identical_tables():
    from d3il_amd.agents import ScriptedPushPolicy
    from d3il_amd.simulation.pushing_sim import Pushing_Sim
    from d3il_amd.simulation.sorting_sim import Sorting_Sim
    res = {}
    for S in (1, 4):
        sim = Sorting_Sim(seed=0, device="cuda:0", render=False, n_cores=1, n_contexts=60, n_trajectories_per_context=5, max_steps_per_episode=150, n_sub_batches=S)
        m = sim.test_agent(ScriptedPushPolicy("sorting", device="cuda:0"))
        r = sim.last_rollout
        res["sorting", S] = (r["counts"].copy(), r["mode"].cpu().numpy(), r["success"].cpu().numpy(), r["mode_hist"].copy(), m["Metrics/entropy"])
        assert not bool((r["flags"] & ((1 << 16) | (1 << 18))).any())
        sim = Pushing_Sim(seed=0, device="cuda:0", render=False, n_cores=1, n_contexts=60, n_trajectories_per_context=5, max_steps_per_episode=200, n_sub_batches=S)
        sim.test_agent(ScriptedPushPolicy("pushing", device="cuda:0"))
        r = sim.last_rollout
        res["pushing", S] = (r["counts"].copy(), r["mode"].cpu().numpy(), r["success"].cpu().numpy(), r["mean_distance"].cpu().numpy())
    for task in ("sorting", "pushing"):
        for x, y in zip(res[task, 1], res[task, 4]):
            assert np.array_equal(np.asarray(x), np.asarray(y), equal_nan=True), task      # (the entropy of a table without successes is nan in both)
    assert len(np.unique(res["sorting", 1][1])) > 1          # some cubes were delivered: the tables are not trivially equal


@pytest.mark.gpu
def test_stacking_and_aligning_sim_sub_batches_identical_tables():
    """The counterpart of the reference's n_cores fan-out in stacking_sim.py:182-216 / aligning_sim.py:125-160: ``n_sub_batches=4`` steps the rank's rollouts as
    four environment batches on four streams and returns the tables of one batch, bit for bit (scripted policies: every row from that row alone; they learn which
    rollouts their rows are through set_rollout_range)."""
    from d3il_amd.agents import ScriptedAlignPolicy, ScriptedStackPolicy
    from d3il_amd.controllers.scripted_stacking import build_trajectory
    from d3il_amd.envs.stacking import CubeStackingVecEnv, load_test_contexts
    from d3il_amd.model import blob
    from d3il_amd.simulation.aligning_sim import Aligning_Sim
    from d3il_amd.simulation.stacking_sim import Stacking_Sim
    nctx, ntraj = 8, 40                                   # 320 rollouts: four sub-batches of 80
    ctx = load_test_contexts()[:nctx]
    env = CubeStackingVecEnv(1, device=0)
    q0 = env.start()[0]
    env.close()
    js = blob.load_json("stacking")
    orders = [(0, 1, 2), (1, 0, 2), (2, 1, 0), (0, 2, 1)]
    tables = [build_trajectory(js, q0, c, order=orders[i % 4], speed=0.5) for i, c in enumerate(ctx)]
    res = {}
    for S in (1, 4):
        sim = Stacking_Sim(seed=0, device="cuda:0", render=False, n_cores=1, n_contexts=nctx, n_trajectories_per_context=ntraj, max_steps_per_episode=260, contexts=ctx,
                           n_sub_batches=S)
        ret = sim.test_agent(ScriptedStackPolicy(tables, torch.arange(nctx * ntraj) // ntraj, device="cuda:0"))
        r = sim.last_rollout
        assert not bool((r["flags"] & ((1 << 16) | (1 << 18))).any())
        res["stacking", S] = (r["counts"].copy(), r["mode"].cpu().numpy(), r["success"].cpu().numpy(), r["mean_distance"].cpu().numpy(), ret[0].cpu().numpy(), ret[1].cpu().numpy())
        sim = Aligning_Sim(seed=0, device="cuda:0", render=False, n_cores=1, n_contexts=60, n_trajectories_per_context=5, max_steps_per_episode=180, n_sub_batches=S)
        ret = sim.test_agent(ScriptedAlignPolicy(inside=(np.arange(300) // 5) % 2 == 0, device="cuda:0"))
        r = sim.last_rollout
        res["aligning", S] = (r["counts"].copy(), r["mode"].cpu().numpy(), r["success"].cpu().numpy(), r["mean_distance"].cpu().numpy(), np.float64(ret[0]), ret[1].cpu().numpy())
    for task in ("stacking", "aligning"):
        for x, y in zip(res[task, 1], res[task, 4]):
            assert np.array_equal(np.asarray(x), np.asarray(y), equal_nan=True), task
    assert len(np.unique(res["stacking", 1][1])) > 1 and len(np.unique(res["aligning", 1][3])) > 30      # boxes were placed / plates were moved: not trivially equal


@pytest.mark.gpu
def test_captured_policy_equals_the_eager_chain_and_forks_its_buffers():
    """policies.CapturedPolicy: the predict chain of a batch as one HIP graph - same kernels, same numbers; a fork owns its graph and static buffers."""
    from d3il_amd.agents import RandomResidualMLPPolicy
    from d3il_amd.policies import CapturedPolicy, DDPMPolicy, DiffusionMLP, Scaler
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = DiffusionMLP(action_dim=2, obs_dim=22, t_dim=8, hidden_dim=256, num_hidden_layers=8).to(dev)
    sc = Scaler([0.0] * 22, [1.0] * 22, [0.0, 0.0], [0.01, 0.01], y_bounds=[[-1.0, -1.0], [1.0, 1.0]], device=dev)
    fixed = torch.randn(512, 2, device=dev)
    ddpm = DDPMPolicy(net, sc, n_timesteps=4, window_size=1, noise_fn=lambda shape: fixed[:shape[0]])
    mlp = RandomResidualMLPPolicy(input_dim=22, device=dev)
    obs = [torch.randn(512, 22, device=dev, dtype=torch.float64) for _ in range(3)]
    for pol in (ddpm, mlp):
        cap = CapturedPolicy(pol)
        for o in obs:
            want = pol.predict_batch(o).clone()
            got = cap.predict_batch(o)
            assert torch.equal(want, got)
        twin = cap.fork()
        a = cap.predict_batch(obs[0]).clone()
        b = twin.predict_batch(obs[1])
        assert twin._g is not cap._g and twin._g_in.data_ptr() != cap._g_in.data_ptr() and b.data_ptr() != cap._g_out.data_ptr()
        assert torch.equal(cap._g_out, a)                      # the twin's call did not touch the first one's output
        assert torch.equal(b, pol.predict_batch(obs[1]))
    with pytest.raises(AssertionError):
        DDPMPolicy(net, sc, n_timesteps=4, window_size=4).captured()
    # a captured policy follows its parameters: the packed buffers of the fused chain are refreshed in place (the EMA swap of a rollout, ddpm_agent.py:213-222)
    cap = CapturedPolicy(ddpm)
    before = cap.predict_batch(obs[0]).clone()
    ddpm.use_ema([p.detach() * 1.25 for p in net.parameters()])
    after = cap.predict_batch(obs[0]).clone()
    assert not torch.equal(before, after) and torch.equal(after, ddpm.predict_batch(obs[0]))
    # default noise: drawn inside the graph, fresh at every replay
    cap = DDPMPolicy(net, sc, n_timesteps=4, window_size=1).captured()
    x, y = cap.predict_batch(obs[0]).clone(), cap.predict_batch(obs[0]).clone()
    assert not torch.equal(x, y) and bool(torch.isfinite(x).all())


@pytest.mark.gpu
def test_sorting_sim_with_a_captured_policy_in_four_sub_batches():
    from d3il_amd.agents import RandomResidualMLPPolicy
    from d3il_amd.policies import CapturedPolicy
    from d3il_amd.simulation.sorting_sim import Sorting_Sim
    res = []
    for S, wrap in ((1, False), (4, True)):
        pol = RandomResidualMLPPolicy(input_dim=2 + 2 + 3 * 4, device="cuda:0")      # desired xy || obs of Sorting-4 (robot xy + 3 per box)
        sim = Sorting_Sim(seed=0, device="cuda:0", render=False, n_cores=1, n_contexts=60, n_trajectories_per_context=5, max_steps_per_episode=80, n_sub_batches=S)
        sim.test_agent(CapturedPolicy(pol) if wrap else pol)
        r = sim.last_rollout
        res.append((r["counts"].copy(), r["mode"].cpu().numpy(), r["success"].cpu().numpy()))
    for x, y in zip(*res):
        assert np.array_equal(np.asarray(x), np.asarray(y))


@pytest.mark.gpu
@pytest.mark.parametrize("n,obs_dim,layers", [(1000, 16, 8), (64, 10, 8), (37, 16, 2)])
def test_fused_ddpm_chain_matches_the_torch_chain(n, obs_dim, layers, monkeypatch):
    """csrc/rollout.hip k_ddpm_mlp_f32 (the DDPM policy's whole sampling chain in one launch, f32 matrix cores) against the torch chain of the same policy on the
    same noise: f32 products in a different summation order and Mish through one exponential - 5e-5 of the scaled action range (measured 2e-5 at worst, 1e-7 typically) [-1, 1]."""
    from d3il_amd.policies import DDPMPolicy, DiffusionMLP, Scaler
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    net = DiffusionMLP(action_dim=2, obs_dim=obs_dim, t_dim=8, hidden_dim=256, num_hidden_layers=layers).to(dev)
    sc = Scaler([0.1] * obs_dim, [0.7] * obs_dim, [0.0, 0.0], [1.0, 1.0], y_bounds=[[-1.0, -1.0], [1.0, 1.0]], device=dev)
    draws = torch.randn(5, n, 2, device=dev)

    class Noise:
        def __init__(self):
            self.k = 0

        def __call__(self, shape):
            self.k += 1
            return draws[(self.k - 1) % 5][:shape[0]]

    pol = DDPMPolicy(net, sc, n_timesteps=4, window_size=1, noise_fn=Noise())
    assert pol.fused_ok()
    obs = torch.randn(n, obs_dim, device=dev, dtype=torch.float64)
    s = sc.scale_input(obs.to(torch.float32))
    with torch.no_grad():
        monkeypatch.setenv("D3IL_POLICY_FUSED_RESMLP", "0")      # the reference side of this comparison is torch's layers all the way down
        want = pol._sample(s)
        monkeypatch.delenv("D3IL_POLICY_FUSED_RESMLP")
        pol.noise_fn.k = 0
        got = pol._sample_fused(s)
    assert got.shape == want.shape and bool(torch.isfinite(got).all())
    assert float((got - want).abs().max()) < 5e-5, float((got - want).abs().max())
    assert float(want.abs().max()) > 0.05 and float((want.abs() < 1.0).float().mean()) > 0.3      # the comparison is not one of clamped values only
    pol.noise_fn.k = 0
    assert torch.equal(pol.predict_batch(obs), sc.inverse_scale_output(got))                     # predict_batch takes the fused path


@pytest.mark.gpu
def test_sorting_sim_with_the_fused_ddpm_policy_in_four_sub_batches():
    """BASELINE config 4 through the product classes: Sorting_Sim, four sub-batches, the DDPM policy (fused sampling chain, captured per sub-batch; fresh noise at
    every call, so the tables are not comparable between runs - what is asserted: every rollout ran to its end, no solver flag, the forks own their buffers)."""
    from d3il_amd.policies import DDPMPolicy, DiffusionMLP, Scaler
    from d3il_amd.simulation.sorting_sim import Sorting_Sim
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = DiffusionMLP(action_dim=2, obs_dim=16, t_dim=8, hidden_dim=256, num_hidden_layers=8).to(dev)
    sc = Scaler([0.0] * 16, [1.0] * 16, [0.0, 0.0], [0.01, 0.01], y_bounds=[[-1.0, -1.0], [1.0, 1.0]], device=dev)
    pol = DDPMPolicy(net, sc, n_timesteps=4, window_size=1)
    assert pol.fused_ok()
    twin = pol.fork()
    o = torch.zeros(64, 16, device=dev, dtype=torch.float64)
    pol.predict_batch(o); twin.predict_batch(o)
    assert twin._fw["w_blk"].data_ptr() != pol._fw["w_blk"].data_ptr() and torch.equal(twin._fw["w_blk"], pol._fw["w_blk"])
    sim = Sorting_Sim(seed=0, device="cuda:0", render=False, n_cores=1, n_contexts=60, n_trajectories_per_context=5, max_steps_per_episode=60, n_sub_batches=4)
    sim.test_agent(pol.captured())
    r = sim.last_rollout
    assert int(r["mode_hist"].sum()) == 300 and r["mode"].shape[0] == 300 and not bool((r["flags"] & ((1 << 16) | (1 << 18))).any())


@pytest.mark.gpu
@pytest.mark.parametrize("n,inp,hid,layers,out", [(1000, 10, 128, 6, 2), (37, 20, 128, 6, 8), (4096, 16, 128, 6, 2), (300, 26, 256, 8, 2), (64, 28, 256, 2, 16)])
def test_fused_residual_mlp_matches_torch_layers(n, inp, hid, layers, out, monkeypatch):
    """csrc/rollout.hip k_resmlp_f32 (ResidualMLPNetwork, common/mlp.py:114-182, in one launch on the f32 matrix cores) against torch's layers: f32 sums in another
    order, Mish through one exponential."""
    from d3il_amd.policies import ResidualMLP
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    net = ResidualMLP(inp, hid, layers, out).to(dev).eval()
    x = torch.randn(n, inp, device=dev)
    with torch.no_grad():
        monkeypatch.setenv("D3IL_POLICY_FUSED_RESMLP", "0")
        want = net(x)
        monkeypatch.delenv("D3IL_POLICY_FUSED_RESMLP")
        got = net(x)
        assert net._fused is not None and net._fused._key is not None            # the device path ran
        assert float((got - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max())), float((got - want).abs().max())
        net.layers[0].weight.mul_(1.5)                                            # the packed copy follows the parameters
        assert float((net(x) - got).abs().max()) > 1e-3
        import copy
        twin = copy.deepcopy(net)                                                 # a deep copy packs ITS weights into ITS buffers
        twin.layers[-1].bias.add_(1.0)
        assert float((twin(x) - net(x) - 1.0).abs().max()) < 1e-5 and twin._fused._fw["w_blk"].data_ptr() != net._fused._fw["w_blk"].data_ptr()
    x.requires_grad_(True)
    assert net(x).requires_grad                                                   # a caller that wants gradients gets torch's layers


@pytest.mark.gpu
def test_stand_in_mlp_policy_takes_the_fused_path(monkeypatch):
    from d3il_amd.agents import RandomResidualMLPPolicy
    pol = RandomResidualMLPPolicy(input_dim=16, device="cuda:0")
    obs = torch.randn(777, 16, device="cuda:0", dtype=torch.float64) * 0.3
    monkeypatch.setenv("D3IL_POLICY_FUSED_RESMLP", "0")
    want = pol.predict_batch(obs).clone()
    monkeypatch.delenv("D3IL_POLICY_FUSED_RESMLP")
    got = pol.predict_batch(obs)
    assert pol._fused._key is not None and float((got - want).abs().max()) < 1e-6 and float(want.abs().max()) > 1e-3
