"""Batched adapters for the reference's policy protocol (the consumer side of the rollout boundary).

The reference agents are batch-1 and numpy-in/numpy-out: ``agent.predict(np.ndarray[obs]) -> np.ndarray[1, act]``
(agents/base_agent.py:110-122), with a host<->device round trip inside every call (agents/bc_agent.py:259-271).  The
adapters below run the *same* computation once per step on the whole environment batch, on device-resident
observations, and are what ``Avoiding_Sim.test_agent`` looks for (``predict_batch``).  Networks and scalers are the
reference's own objects - nothing is re-implemented here.
"""
from __future__ import annotations

import torch


class BatchedBCAgent:
    """Wraps a reference ``BC_Agent`` (agents/bc_agent.py): attributes used are ``model``, ``scaler``
    (``scale_input`` / ``inverse_scale_output``, agents/utils/scaler.py:72-113), ``min_action``, ``max_action``."""

    def __init__(self, agent):
        self.agent = agent

    def reset(self):
        if hasattr(self.agent, "reset"):
            self.agent.reset()

    @torch.no_grad()
    def predict_batch(self, obs: torch.Tensor) -> torch.Tensor:
        """obs [N, obs_dim] on the policy's device -> actions [N, act_dim]; row i equals ``agent.predict(obs[i])[0]``
        (bc_agent.py:240-271 with the batch dimension N instead of 1)."""
        a = self.agent
        a.model.eval()
        state = obs.to(torch.float32).unsqueeze(1)            # [N, 1, obs]; the reference builds [1, 1, obs]
        state = a.scaler.scale_input(state)
        out = a.model(state)
        out = out.clamp_(a.min_action, a.max_action)
        pred = a.scaler.inverse_scale_output(out)
        return pred[:, 0]

    def predict(self, state, *args, **kwargs):               # the reference protocol still works
        return self.agent.predict(state, *args, **kwargs)


def _is_lane_state(v) -> bool:
    """Attributes of a reference agent that hold per-episode state: history deques (ddpm_agent.py:61, beso_agent.py:86-87,
    bet_agent.py:131), action-chunk buffers, plain containers and non-parameter tensors / arrays."""
    import collections
    import numpy as np
    if isinstance(v, (collections.deque, list, dict, set, bytearray, np.ndarray)):
        return True
    return isinstance(v, torch.Tensor) and not isinstance(v, torch.nn.Parameter)


class RowwiseAgent:
    """Fallback adapter for any reference agent without ``predict_batch``: calls ``predict`` row by row (host round trip per
    environment, slow), with ONE AGENT STATE PER LANE.  Most reference agents keep per-episode state inside ``predict`` - BeT,
    BESO, DDPM and ACT hold an observation deque and action-chunk counters (e.g. ddpm_agent.py:223-227, beso_agent.py:354-355) -
    so a single instance stepped through N lock-stepped environments would mix the histories of different environments.  Every
    lane therefore gets a shallow clone of the agent that shares the network, scaler and everything immutable, and owns copies
    of the per-episode containers."""

    def __init__(self, agent, n_envs: int | None = None):
        self.agent = agent
        self.lanes = []
        if n_envs:
            self._grow(n_envs)

    def _clone(self):
        import copy
        c = copy.copy(self.agent)
        for k, v in vars(self.agent).items():
            if _is_lane_state(v):
                setattr(c, k, copy.deepcopy(v))
        return c

    def _grow(self, n):
        while len(self.lanes) < n:
            self.lanes.append(self._clone())

    def reset(self):
        if hasattr(self.agent, "reset"):
            self.agent.reset()
            for a in self.lanes:
                a.reset()

    def begin_episodes(self, mask: torch.Tensor):
        """Lanes that start their next trajectory (env.last_reset) get a fresh history."""
        for i in torch.nonzero(mask.reshape(-1)).reshape(-1).tolist():
            if i < len(self.lanes) and hasattr(self.lanes[i], "reset"):
                self.lanes[i].reset()

    def predict_batch(self, obs: torch.Tensor) -> torch.Tensor:
        import numpy as np
        rows = obs.detach().cpu().numpy()
        self._grow(len(rows))
        acts = np.stack([np.asarray(self.lanes[i].predict(r)).reshape(-1) for i, r in enumerate(rows)])
        return torch.as_tensor(acts, dtype=torch.float64, device=obs.device)


def as_batched(agent, n_envs: int | None = None):
    """What the batched sims call: agents that provide ``predict_batch`` (the adapters of this module, or any policy written
    for the batch) are used as they are; everything else is wrapped row by row with per-lane state."""
    if hasattr(agent, "predict_batch"):
        if not hasattr(agent, "reset"):
            agent.reset = lambda: None
        return agent
    return RowwiseAgent(agent, n_envs)


class RandomResidualMLPPolicy(torch.nn.Module):
    """Stand-in policy of BASELINE config 3: the architecture of the reference's BC policy for Pushing
    (agents/models/common/mlp.py:114-182 as configured by configs/agents/bc_agent.yaml:11-22 and
    configs/pushing_config.yaml: input 10 = desired xy + obs 8, hidden 128, 6 hidden layers = 3 pre-activation residual
    blocks, Mish, output 2) with fixed random weights - there are no checkpoints offline.  Outputs are clamped to the env's
    action box +-0.01 (pushing.py:203-205), the role the data-derived action bounds play in BC_Agent.predict
    (bc_agent.py:262)."""

    def __init__(self, input_dim=10, hidden_dim=128, num_hidden_layers=6, output_dim=2, seed=0, device="cuda", bound=0.01):
        super().__init__()
        assert num_hidden_layers % 2 == 0
        g = torch.Generator().manual_seed(seed)
        self.bound = bound

        def lin(i, o):
            l = torch.nn.Linear(i, o)
            with torch.no_grad():
                k = 1.0 / i ** 0.5
                l.weight.copy_((torch.rand(o, i, generator=g) * 2 - 1) * k)
                l.bias.copy_((torch.rand(o, generator=g) * 2 - 1) * k)
            return l

        self.inp = lin(input_dim, hidden_dim)
        self.blocks = torch.nn.ModuleList([torch.nn.ModuleList([lin(hidden_dim, hidden_dim), lin(hidden_dim, hidden_dim)])
                                           for _ in range(num_hidden_layers // 2)])
        self.out = lin(hidden_dim, output_dim)
        self.act = torch.nn.Mish()
        self.to(device)

    def reset(self):
        pass

    def _parts(self):
        return (self.inp, [(b[0], b[1]) for b in self.blocks], self.out)

    def ensure_packed(self):
        if getattr(self, "_fused", None) is not None and self._fused._fw is not None:
            self._fused.ensure_packed(self._parts())

    def invalidate_packed(self):
        """After ``param.data`` writes (invisible to the tensors' version counters): the next call repacks the device path's weights."""
        if getattr(self, "_fused", None) is not None:
            self._fused.invalidate()

    @torch.no_grad()
    def predict_batch(self, obs: torch.Tensor) -> torch.Tensor:
        if obs.is_cuda and obs.dim() == 2:
            if getattr(self, "_fused", None) is None:
                from .policies import FusedResMLP
                object.__setattr__(self, "_fused", FusedResMLP())
            x32 = obs.to(torch.float32)
            parts = self._parts()
            if self._fused.ok(x32, parts):
                return self._fused(x32, parts).clamp_(-self.bound, self.bound)      # one launch on the f32 matrix cores (policies.FusedResMLP)
        x = self.inp(obs.to(torch.float32))
        for l1, l2 in self.blocks:
            x = x + l2(self.act(l1(self.act(x))))
        return self.out(x).clamp_(-self.bound, self.bound)


class ScriptedPushPolicy:
    """Scripted contact-regime policy for the measurement harness (bench.py --policy scripted_push, tools/): every rod walks
    behind a cube and pushes it - Pushing: the red cube to the red target, then the green cube to the green target (two-phase
    script); Sorting: the first cube still on the platform over the platform edge into the bins; Inserting: the red, green and blue cube one after the
    other to the mouth of its gate and then into it (six-phase script; the walls are not planned around: cube and rod run into them, which is the
    contact regime the measurement wants).  Input / output like the
    rollout loops of the sims: ``predict_batch([des_xy, obs]) -> delta_xy`` with |delta| <= 6 mm (the env's action box is +-1 cm).
    Per-lane phase state is re-latched by ``begin_episodes(mask)`` when a lane starts its next trajectory."""

    STEP = 0.006

    def __init__(self, task: str, device="cuda"):
        assert task in ("pushing", "sorting", "inserting")
        self.task, self.device = task, torch.device(device)
        self.goals = torch.tensor([[0.42, 0.3], [0.63, 0.3]], dtype=torch.float64, device=self.device)   # pushing_objects.py:11-15 (x, y of the two targets)
        if task == "inserting":      # per cube: the mouth of its gate, then its target (gate_insertion_objects.py:17-24; the gates open towards +x, -y, -x)
            self.goals = torch.tensor([[0.45, 0.276], [0.3575, 0.276], [0.525, 0.33], [0.525, 0.4535], [0.60, 0.276], [0.6925, 0.276], [0.6925, 0.276]],
                                      dtype=torch.float64, device=self.device)
        self.phase = None

    def reset(self):
        self.phase = None

    def begin_episodes(self, mask: torch.Tensor):
        if self.phase is not None:
            self.phase = torch.where(mask.bool(), torch.zeros_like(self.phase), self.phase)

    @torch.no_grad()
    def predict_batch(self, obs_in: torch.Tensor) -> torch.Tensor:
        o = obs_in.to(torch.float64)
        n = o.shape[0]
        des, obs = o[:, :2], o[:, 2:]
        if self.phase is None:
            self.phase = torch.zeros(n, dtype=torch.long, device=o.device)
        if self.task == "pushing":
            box = torch.where(self.phase.unsqueeze(1) == 0, obs[:, 2:4], obs[:, 5:7])
            goal = self.goals[self.phase]
            togo = goal - box
            dist = togo.norm(dim=1, keepdim=True)
            self.phase = torch.where((dist.squeeze(1) < 0.03) & (self.phase == 0), torch.ones_like(self.phase), self.phase)
            dirn = togo / dist.clamp_min(1e-9)
        elif self.task == "inserting":
            cube = torch.clamp(self.phase // 2, max=2)
            box = obs[:, 2:].reshape(n, 3, 3)[torch.arange(n, device=o.device), cube, :2]
            togo = self.goals[self.phase] - box
            dist = togo.norm(dim=1, keepdim=True)
            reached = dist.squeeze(1) < torch.where(self.phase % 2 == 0, 0.012, 0.006)
            self.phase = torch.where(reached & (self.phase < 6), self.phase + 1, self.phase)
            dirn = togo / dist.clamp_min(1e-9)
            dist = (self.phase < 6).to(torch.float64).unsqueeze(1)       # all three in: hold
        else:
            nb = (obs.shape[1] - 2) // 3
            xy = obs[:, 2:].reshape(n, nb, 3)[:, :, :2]
            on_platform = xy[:, :, 1] < 0.19
            first = torch.argmax(on_platform.to(torch.int64), dim=1)                 # first cube still on the platform (0 when none is)
            box = xy[torch.arange(n, device=o.device), first]
            dirn = torch.tensor([0.0, 1.0], dtype=torch.float64, device=o.device).expand(n, 2)
            dist = torch.where(on_platform.any(dim=1, keepdim=True), torch.ones(n, 1, dtype=torch.float64, device=o.device),
                               torch.zeros(n, 1, dtype=torch.float64, device=o.device))
        behind = box - dirn * 0.055                                # stand-off point behind the cube, on the line to its goal
        off = des - behind
        along = (off * dirn).sum(1, keepdim=True)
        lateral = off - along * dirn
        aligned = (lateral.norm(dim=1, keepdim=True) < 0.012) & (along < 0.02)
        target = torch.where(aligned, box, behind)
        d = target - des
        dn = d.norm(dim=1, keepdim=True)
        step = d / dn.clamp_min(1e-9) * torch.minimum(dn, torch.full_like(dn, self.STEP))
        # go around the cube when the straight line to the stand-off point crosses it
        rel = des - box
        rn = rel.norm(dim=1, keepdim=True)
        near = (rn < 0.06) & ~aligned
        away = rel / rn.clamp_min(1e-9)
        tang = torch.stack((-away[:, 1], away[:, 0]), dim=1)
        tang = tang * torch.sign((tang * (behind - des)).sum(1, keepdim=True) + 1e-12)
        step = torch.where(near, self.STEP * (0.6 * tang + 0.4 * away), step)
        if self.task in ("sorting", "inserting"):
            step = step * dist                                        # nothing left on the platform / all cubes in their gates: hold
        return step


class ScriptedStackPolicy:
    """Scripted pick-and-place policy of the Stacking measurement harness: a per-context table of joint-space actions
    (controllers/scripted_stacking.py) replayed by episode step.  ``predict_batch([last command 8, obs 12]) -> [delta joints 7,
    gripper command]`` - the interface of the Stacking rollout loop (stacking_sim.py:99-106: the policy output is added to the last
    commanded joints).  ``ctx_id``: context index of every lane; the episode step of a lane restarts with ``begin_episodes(mask)``."""

    def __init__(self, tables, ctx_id, device="cuda"):
        self.device = torch.device(device)
        T = max(len(t) for t in tables)
        tab = torch.zeros(len(tables), T, 8, dtype=torch.float64)
        for i, t in enumerate(tables):
            tt = torch.as_tensor(t, dtype=torch.float64)
            tab[i, :len(t)] = tt
            tab[i, len(t):] = tt[-1]
        self.table = tab.to(self.device)
        self.ctx_id = torch.as_tensor(ctx_id, dtype=torch.int64, device=self.device)
        self._ctx_all = self.ctx_id
        self.t = torch.zeros(len(self.ctx_id), dtype=torch.int64, device=self.device)

    def set_rollout_range(self, offset, count):
        """Rows 0 .. count-1 of this clone's batch are rollouts offset .. offset+count-1 (a sub-batch of the rank's batch, envs/sub_batch.py)."""
        self.ctx_id = self._ctx_all[offset:offset + count]
        self.t = torch.zeros(count, dtype=torch.int64, device=self.device)

    def reset(self):
        self.t.zero_()

    def begin_episodes(self, mask: torch.Tensor):
        self.t = torch.where(mask.bool(), torch.zeros_like(self.t), self.t)

    @torch.no_grad()
    def predict_batch(self, obs20: torch.Tensor) -> torch.Tensor:
        a = self.table[self.ctx_id, self.t.clamp_max(self.table.shape[1] - 1)]
        self.t += 1
        out = a.clone()
        out[:, :7] -= obs20[:, :7].to(torch.float64)
        return out


class ScriptedGoalPushPolicy:
    """Closed-loop scripted policy that FINISHES the pushing tasks (evaluation harness for the integer-count parity tests and the
    contact-regime bench lines; not part of the reference).  Same interface as the rollout loops of the sims:
    ``predict_batch([des_xy, obs]) -> delta_xy`` with |delta| <= ``STEP`` (the env's action box is +-1 cm, pushing.py:203-205).

    * ``task='pushing'``: the two cubes are pushed to the two targets in one of the FOUR behaviour modes of pushing.py:341-377, chosen per
      lane by ``plan`` (0: red->red target then green->green target, 1: green->green then red->red, 2: red->green target then
      green->red target, 3: green->red then red->green) - so a batch produces a non-trivial mode table.
    * ``task='sorting'``: every cube still on the platform is pushed over the platform edge into the bin of its colour (red: x 0.4, blue:
      x 0.625; sorting.py:300-301), front-most cube first; the completion order (the mode code, sorting.py:460-507) follows from the context.

    A push: walk (around the cube if necessary) to a stand-off point behind the cube on the line cube -> goal, then advance along that
    line while steering the lateral offset to zero; the line is re-aimed every step from the observed cube position."""

    STEP = 0.006
    GOALS = ((0.42, 0.3), (0.63, 0.3))                                   # pushing_objects.py:11-15: red target, green target
    PLANS = (((0, 0), (1, 1)), ((1, 1), (0, 0)), ((0, 1), (1, 0)), ((1, 0), (0, 1)))   # ((cube, target), (cube, target)) per mode

    def __init__(self, task: str, plan=None, device="cuda"):
        assert task in ("pushing", "sorting")
        self.task, self.device = task, torch.device(device)
        self.plan = None if plan is None else torch.as_tensor(plan, dtype=torch.long, device=self.device)
        self.stage = None

    def reset(self):
        self.stage = None

    def begin_episodes(self, mask: torch.Tensor):
        if self.stage is not None:
            self.stage = torch.where(mask.bool(), torch.zeros_like(self.stage), self.stage)

    def _current(self, obs, n, dev):
        """(cube xy, goal xy, active) of the push every lane is working on."""
        f64 = dict(dtype=torch.float64, device=dev)
        if self.task == "pushing":
            cubes = torch.stack((obs[:, 2:4], obs[:, 5:7]), dim=1)                      # [n, 2, 2]
            plan = self.plan if self.plan is not None else torch.zeros(n, dtype=torch.long, device=dev)
            table = torch.tensor(self.PLANS, dtype=torch.long, device=dev)[plan]        # [n, 2 stages, (cube, target)]
            goals = torch.tensor(self.GOALS, **f64)
            ar = torch.arange(n, device=dev)
            for _ in range(2):                                                           # a finished stage hands over to the next one
                st = self.stage.clamp_max(1)
                cube_i, goal_i = table[ar, st, 0], table[ar, st, 1]
                box, goal = cubes[ar, cube_i], goals[goal_i]
                arrived = ((goal - box).norm(dim=1) < 0.02) & (self.stage < 2)
                self.stage = self.stage + arrived.long()
            return box, goal, self.stage < 2
        nb = (obs.shape[1] - 2) // 3
        xy = obs[:, 2:].reshape(n, nb, 3)[:, :, :2]
        goal_x = torch.tensor([0.4] * (nb // 2) + [0.625] * (nb // 2), **f64)
        on_platform = xy[:, :, 1] < 0.232                                               # not yet over the bin wall (y = 0.22 +- 0.005)
        key = torch.where(on_platform, xy[:, :, 1], torch.full_like(xy[:, :, 1], -1e9))
        k = torch.argmax(key, dim=1)                                                    # front-most cube still on the platform
        ar = torch.arange(n, device=dev)
        box = xy[ar, k]
        gx = goal_x[k]
        # first sideways on the platform to the x of the cube's bin (at its own y, at most 0.10: away from the edge), then straight over it
        lined_up = (box[:, 0] - gx).abs() < 0.025
        goal = torch.stack((gx, torch.where(lined_up, torch.full((n,), 0.34, **f64), box[:, 1].clamp_max(0.10))), dim=1)
        return box, goal, on_platform.any(dim=1)

    @torch.no_grad()
    def predict_batch(self, obs_in: torch.Tensor) -> torch.Tensor:
        o = obs_in.to(torch.float64)
        n, dev = o.shape[0], o.device
        des, obs = o[:, :2], o[:, 2:]
        if self.stage is None:
            self.stage = torch.zeros(n, dtype=torch.long, device=dev)
        box, goal, active = self._current(obs, n, dev)
        togo = goal - box
        dirn = togo / togo.norm(dim=1, keepdim=True).clamp_min(1e-9)
        behind = box - dirn * 0.06
        off = des - behind
        along = (off * dirn).sum(1, keepdim=True)
        lateral = off - along * dirn
        latn = lateral.norm(dim=1, keepdim=True)
        aligned = (latn < 0.015) & (along > -0.03) & (along < 0.05)
        push = self.STEP * dirn - lateral * torch.clamp(0.003 / latn.clamp_min(1e-9), max=0.5)
        d = behind - des
        dn = d.norm(dim=1, keepdim=True)
        walk = d / dn.clamp_min(1e-9) * torch.minimum(dn, torch.full_like(dn, self.STEP))
        rel = des - box                                                                   # around the cube, not through it
        rn = rel.norm(dim=1, keepdim=True)
        away = rel / rn.clamp_min(1e-9)
        tang = torch.stack((-away[:, 1], away[:, 0]), dim=1)
        tang = tang * torch.sign((tang * (behind - des)).sum(1, keepdim=True) + 1e-12)
        around = self.STEP * (0.8 * tang + 0.6 * (0.075 - rn) / 0.01 * away).clamp(-1.0, 1.0)
        step = torch.where(aligned, push, torch.where((rn < 0.08) & (dn > 0.02), around, walk))
        return step * active.unsqueeze(1).to(torch.float64)


class ScriptedAlignPolicy:
    """Scripted policy for the Aligning task (evaluation harness for the parity tests and the bench line; not part of the reference).  Interface of
    the rollout loop of ``Aligning_Sim``: ``predict_batch([des_xyz, obs17]) -> delta_xyz``.  Two behaviours, chosen per lane by ``inside`` - they are
    the two behaviour modes the task counts (aligning.py:288-312):

    * inside  (mode 0): above the box centre, down between the walls (rod tip 2 cm above the plate), then the box is dragged towards the target;
    * outside (mode 1): to a stand-off point 12 cm behind the box on the line box -> target (3 cm off-centre, so that the box also turns), down, then
      a push along that line.

    The way-points are fixed from the FIRST observation of an episode and the phases by the step count (40 steps travel, 40 steps descent, then 4 mm per
    step), so the commanded path depends on the context only - the episode's outcome then measures the physics, not a feedback loop."""

    HIGH, LOW = 0.25, 0.14

    def __init__(self, inside=None, device="cuda"):
        self.device = torch.device(device)
        self.inside = None if inside is None else torch.as_tensor(inside, dtype=torch.bool, device=self.device)
        self._inside_all = self.inside
        self.t = None
        self.way = None

    def set_rollout_range(self, offset, count):
        """Rows 0 .. count-1 of this clone's batch are rollouts offset .. offset+count-1 (a sub-batch of the rank's batch, envs/sub_batch.py)."""
        if self._inside_all is not None:
            self.inside = self._inside_all[offset:offset + count]
        self.t, self.way = None, None

    def reset(self):
        self.t, self.way = None, None

    def begin_episodes(self, mask: torch.Tensor):
        if self.t is not None:
            self.t = torch.where(mask.bool(), torch.zeros_like(self.t), self.t)
            self._fresh = mask.bool() if getattr(self, "_fresh", None) is None else (self._fresh | mask.bool())

    @torch.no_grad()
    def predict_batch(self, obs20: torch.Tensor) -> torch.Tensor:
        x = obs20.to(torch.float64)
        n, dev = x.shape[0], x.device
        des, box, tgt = x[:, 0:3], x[:, 6:8], x[:, 13:15]
        if self.t is None:
            self.t = torch.zeros(n, dtype=torch.long, device=dev)
            self._fresh = torch.ones(n, dtype=torch.bool, device=dev)
        inside = self.inside if self.inside is not None else torch.ones(n, dtype=torch.bool, device=dev)
        fresh = self._fresh | (self.t == 0)
        d = tgt - box
        u = d / d.norm(dim=1, keepdim=True).clamp_min(1e-9)
        perp = torch.stack((-u[:, 1], u[:, 0]), dim=1)
        start = torch.where(inside.unsqueeze(1), box, box - 0.12 * u + 0.03 * perp)
        goal = torch.where(inside.unsqueeze(1), tgt, box + 0.8 * d + 0.03 * perp)
        way = torch.cat((start, goal), dim=1)
        self.way = way if self.way is None else torch.where(fresh.unsqueeze(1), way, self.way)
        self._fresh = torch.zeros(n, dtype=torch.bool, device=dev)
        t = self.t.unsqueeze(1)
        z = torch.where(t < 40, torch.full((n, 1), self.HIGH, dtype=torch.float64, device=dev), torch.full((n, 1), self.LOW, dtype=torch.float64, device=dev))
        xy = torch.where(t < 80, self.way[:, 0:2], self.way[:, 2:4])
        step = torch.where(t < 80, torch.full((n, 1), 0.008, dtype=torch.float64, device=dev), torch.full((n, 1), 0.004, dtype=torch.float64, device=dev))
        dd = torch.cat((xy, z), dim=1) - des
        nn = dd.norm(dim=1, keepdim=True).clamp_min(1e-12)
        self.t = self.t + 1
        return dd / nn * torch.minimum(nn, step)
