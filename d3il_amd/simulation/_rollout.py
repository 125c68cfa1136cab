"""The rollout loops of the Sim classes: the one the xy-commanded tasks share (pushing_sim.py:69-84, sorting_sim.py:118-133: obs := desired xy || env obs, action := policy
delta + desired xy, frozen z and quaternion; what is recorded is ``info[...]`` of the step that returned ``done``), the joint-space loop of Stacking and the xyz loop of Aligning, run over the sub-batches of a
rank (envs/sub_batch.py): every sub-batch steps on its own stream with its own agent clone; the host looks at the ``finished`` flags every 16th
step only."""
from __future__ import annotations

import torch

from ..envs.sub_batch import SubBatchSet


class _Lanes:
    pass


def xy_rollout(batches: SubBatchSet, max_steps: int, record: dict, predict=None):
    """record: {info key: (dtype, initial value)}.  Returns {key: tensor [n]} (sub-batch order = rollout order) plus 'flags' (the environments'
    flag words after the rollout).  predict(agent, obs_in) -> [n, 2] f64 (the Sim class's ``_predict`` hook)."""
    dev = batches.device
    if predict is None:
        def predict(agent, obs_in):
            return agent.predict_batch(obs_in).to(device=dev, dtype=torch.float64).reshape(obs_in.shape[0], 2)

    def begin(b):
        st = b.state = _Lanes()
        n = b.n
        st.quat = torch.tensor([0.0, 1.0, 0.0, 0.0], dtype=torch.float64, device=dev).expand(n, 4)
        st.finished = torch.zeros(n, dtype=torch.bool, device=dev)
        st.rec = {k: torch.full((n,), init, dtype=dt, device=dev) for k, (dt, init) in record.items()}
        pred_action = b.env.robot_state().clone()
        st.fixed_z = pred_action[:, 2:3].clone()
        st.des_xy = pred_action[:, :2].clone()
        st.obs = b.env.obs

    def step(b):
        st = b.state
        obs_in = torch.cat((st.des_xy, st.obs.to(torch.float64)), dim=1)
        delta = predict(b.agent, obs_in)
        des_new = delta + obs_in[:, :2]
        st.des_xy = torch.where(st.finished.unsqueeze(1), st.des_xy, des_new)
        action = torch.cat((st.des_xy, st.fixed_z, st.quat), dim=1).contiguous()
        st.obs, _, done, info = b.env.step(action)
        newly = ~st.finished & done.bool()
        for k in st.rec:
            st.rec[k] = torch.where(newly, info[k].to(st.rec[k].dtype), st.rec[k])
        st.finished |= done.bool()

    return _run(batches, max_steps, begin, step, record)


def _run(batches: SubBatchSet, max_steps: int, begin, step, record: dict):
    """The driver the rollout loops share: begin / step per sub-batch on its own stream, the host looks at the ``finished`` flags every 16th step."""
    dev = batches.device
    batches.each(begin)
    for t in range(max_steps):
        batches.each(step)
        if t % 16 == 15:                                   # the only host synchronisation of the loop
            batches.join()
            if bool(torch.stack([b.state.finished.all() for b in batches]).all()):
                break
    batches.join()
    torch.cuda.synchronize(dev)      # the tables below are read on the caller's stream; the sub-batch streams have nothing left in flight
    out = {k: torch.cat([b.state.rec[k] for b in batches]) for k in record}
    out["flags"] = torch.cat([b.env.flags[:b.n].clone() for b in batches])
    return out


def _record(st, done, info):
    newly = ~st.finished & done.bool()
    for k in st.rec:
        st.rec[k] = torch.where(newly, info[k].to(st.rec[k].dtype), st.rec[k])
    st.finished |= done.bool()


def _begin_lanes(b, record, dev):
    st = b.state = _Lanes()
    st.finished = torch.zeros(b.n, dtype=torch.bool, device=dev)
    st.rec = {k: torch.full((b.n,), init, dtype=dt, device=dev) for k, (dt, init) in record.items()}
    return st


def joint_rollout(batches: SubBatchSet, max_steps: int, record: dict):
    """Stacking (stacking_sim.py:88-109): the policy input is the LAST COMMAND (7 desired joint positions + gripper command, initially env.robot_state())
    concatenated with the env observation; its output is a joint-position delta plus the gripper command."""
    dev = batches.device

    def begin(b):
        st = _begin_lanes(b, record, dev)
        st.pred_action = b.env.robot_state().to(torch.float32)              # stacking_sim.py:90-91
        st.obs = b.env.obs

    def step(b):
        st = b.state
        obs20 = torch.cat((st.pred_action, st.obs), dim=1)                  # np.concatenate((pred_action, obs)), stacking_sim.py:99
        out = b.agent.predict_batch(obs20).to(device=dev, dtype=torch.float32).reshape(b.n, 8)
        new_action = torch.cat((out[:, :7] + obs20[:, :7], out[:, 7:8]), dim=1)   # pred_action[:7] += obs[:7], stacking_sim.py:104
        st.pred_action = torch.where(st.finished.unsqueeze(1), st.pred_action, new_action)
        st.obs, _, done, info = b.env.step(st.pred_action.to(torch.float64).contiguous())
        _record(st, done, info)

    return _run(batches, max_steps, begin, step, record)


def xyz_rollout(batches: SubBatchSet, max_steps: int, record: dict, predict):
    """Aligning (aligning_sim.py:96-108): obs := desired xyz || env obs, action := policy delta + desired xyz (the policy commands x, y AND z), frozen
    quaternion [0, 1, 0, 0].  predict(agent, obs20) -> [n, 3] f64."""
    dev = batches.device

    def begin(b):
        st = _begin_lanes(b, record, dev)
        st.quat = torch.tensor([0.0, 1.0, 0.0, 0.0], dtype=torch.float64, device=dev).expand(b.n, 4)
        st.des = b.env.robot_state().clone()                                  # pred_action = env.robot_state(), aligning_sim.py:96
        st.obs = b.env.obs

    def step(b):
        st = b.state
        obs20 = torch.cat((st.des, st.obs.to(torch.float64)), dim=1)         # np.concatenate((pred_action[:3], obs)), aligning_sim.py:99
        des_new = predict(b.agent, obs20) + obs20[:, :3]                       # aligning_sim.py:101-102
        st.des = torch.where(st.finished.unsqueeze(1), st.des, des_new)
        st.obs, _, done, info = b.env.step(torch.cat((st.des, st.quat), dim=1).contiguous())
        _record(st, done, info)

    return _run(batches, max_steps, begin, step, record)
