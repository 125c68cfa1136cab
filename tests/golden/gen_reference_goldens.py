"""Generate golden vectors by executing the reference's own Python code (shim-imported).

Run in the build container only (needs /root/reference):
    python tests/golden/gen_reference_goldens.py
Outputs (committed, small): tests/golden/*.npz.  These pin, against the *actual reference code*:
  * CartPosQuatImpedenceController.getControl   (IKControllers.py:163-323)
  * JointPDController.getControl                (Controller.py:164-185)
  * RobotBase.fing_ctrl_step / preprocessCommand (Robots.py:441-476, 530-572)
  * OfflineIKTrajectoryGenerator                (TrajectoryTracking.py:331-447)  -> init_qpos
  * ObstacleAvoidanceEnv.check_mode / check_success (avoiding.py:173-224)
  * Avoiding_Sim.test_agent metric tail          (avoiding_sim.py:126-144)
The forward kinematics / Jacobian handed to the controller is the build's own URDF chain
(d3il_amd/kinematics.py) because pinocchio is not installed; the KAT in
tests/test_kinematics.py pins that chain independently.
"""
import io
import os
import sys
import contextlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_shims  # noqa: E402

ref_shims.install()

import environments.d3il.d3il_sim.controllers as ctrl  # noqa: E402
import environments.d3il.d3il_sim.core.Robots as Robots  # noqa: E402
from environments.d3il.d3il_sim.core.time_keeper import TimeKeeper  # noqa: E402

from d3il_amd.kinematics import UrdfChain  # noqa: E402
from d3il_amd.model import blob  # noqa: E402

JS = blob.load_json("avoiding")
CHAIN = UrdfChain(JS["urdf_chain"])
QMIN = np.array(JS["controller"]["joint_pos_min"])
QMAX = np.array(JS["controller"]["joint_pos_max"])
QDEF = np.array(JS["controller"]["default_qpos"])
Q_AVOID = np.array([-0.362426, 0.423735, -0.132901, -2.047119, 0.121720, 2.456769, 0.207342])


class FakeRobot:
    def __init__(self, q, dt=0.001):
        self.dt = dt
        self.time_stamp = 0.0
        self.step_count = 0
        self.current_j_pos = np.array(q, float)
        self.current_j_vel = np.zeros(7)
        self.joint_pos_min, self.joint_pos_max = QMIN, QMAX
        self.jointTrackingController = ctrl.JointPDController()
        self.des_joint_pos = np.full(7, np.nan)  # RobotBase.reset state (Robots.py:116)
        self.smooth_spline = True

    def getForwardKinematics(self, q=None):
        return CHAIN.fk(self.current_j_pos if q is None else q)

    def getJacobian(self, q=None):
        return CHAIN.jacobian(self.current_j_pos if q is None else q)


def run_ik_case(rng, q0, setpoint_fn, K, noise):
    robot = FakeRobot(q0)
    c = ctrl.CartPosQuatImpedenceController()
    jpos, jvel, sps, out_u, out_q, out_qd = [], [], [], [], [], []
    virt = np.array(q0, float)
    for k in range(K):
        sp = setpoint_fn(k)
        c.setSetPoint(sp)
        robot.current_j_pos = virt + noise * rng.standard_normal(7)
        robot.current_j_vel = 0.5 * noise / 1e-3 * rng.standard_normal(7) * 0.01
        u = c.getControl(robot)
        virt = c.old_q.copy()
        jpos.append(robot.current_j_pos.copy())
        jvel.append(robot.current_j_vel.copy())
        sps.append(np.asarray(sp, float))
        out_u.append(np.asarray(u, float).copy())
        out_q.append(c.old_q.copy())
        out_qd.append(np.asarray(c.old_des_joint_vel, float).copy())
    return dict(jpos=np.array(jpos), jvel=np.array(jvel), setpoint=np.array(sps), control=np.array(out_u),
                old_q=np.array(out_q), old_qd=np.array(out_qd))


def gen_ik():
    rng = np.random.default_rng(1234)
    cases = {}
    p0, _ = CHAIN.fk(Q_AVOID)
    # 0: Avoiding start pose, slow random walk of the set-point (what the harness does)
    walk = np.cumsum(rng.uniform(-0.01, 0.01, size=(300, 2)), axis=0) / 35.0

    def sp0(k):
        return np.array([p0[0] + walk[k, 0], p0[1] + walk[k, 1], p0[2], 0, 1, 0, 0])
    cases["walk"] = run_ik_case(rng, Q_AVOID, sp0, 300, 1e-4)
    # 1: default posture (min eigenvalue of J J^T < min_svd -> clipping active), far target
    pd, _ = CHAIN.fk(QDEF)

    def sp1(k):
        return np.array([pd[0] - 0.2, pd[1] + 0.3, pd[2] - 0.3, 0.1, 0.9, -0.2, 0.1])
    cases["far"] = run_ik_case(rng, QDEF, sp1, 200, 1e-3)
    # 2: start near joint limits (np.clip to limits active), negative desired quaternion
    qlim = np.array([2.89, 1.76, 1.99, -0.08, 2.89, 3.74, 2.89])
    pl, ql = CHAIN.fk(qlim)

    def sp2(k):
        return np.concatenate([pl + np.array([0.05, 0.05, 0.05]), -ql])
    cases["limits"] = run_ik_case(rng, qlim, sp2, 100, 1e-3)
    # 3: un-normalised quaternion set-point, zero noise
    def sp3(k):
        return np.array([0.45, 0.1 * np.sin(k / 50.0), 0.2, 0.0, 2.0, 0.1, 0.0])
    cases["unnorm"] = run_ik_case(rng, Q_AVOID, sp3, 150, 0.0)
    flat = {}
    for name, d in cases.items():
        for k, v in d.items():
            flat["%s__%s" % (name, k)] = v
    np.savez_compressed(os.path.join(HERE, "ref_ik_controller.npz"), **flat)
    print("ik:", {k: v["control"].shape for k, v in cases.items()})


def gen_pd_finger():
    rng = np.random.default_rng(7)
    pd = ctrl.JointPDController()
    N = 64
    q = rng.uniform(QMIN, QMAX, size=(N, 7))
    v = rng.normal(size=(N, 7))
    qd = rng.uniform(QMIN, QMAX, size=(N, 7))
    vd = rng.normal(size=(N, 7))
    u = np.zeros((N, 7))
    for i in range(N):
        r = FakeRobot(q[i])
        r.current_j_vel = v[i]
        pd.setSetPoint(qd[i], vd[i], np.zeros(7))
        u[i] = pd.getControl(r)
    # finger controller + preprocessCommand
    M = 96
    fpos = rng.uniform(0.0, 0.04, size=(M, 2))
    fpos[:8] = 0.0
    fpos[8:16] = 0.04
    fvel = rng.normal(scale=0.1, size=(M, 2))
    setw = rng.choice([0.001, 0.04, 0.0, 0.02], size=M)
    grasp = rng.integers(0, 2, size=M).astype(bool)
    bias = rng.normal(scale=5.0, size=(M, 9))
    tau = rng.normal(scale=20.0, size=(M, 7))
    force = np.zeros((M, 2))
    uff = np.zeros((M, 9))
    for i in range(M):
        rb = object.__new__(Robots.RobotBase)
        rb.num_DoF = 7
        rb.current_fing_pos, rb.current_fing_vel = fpos[i].copy(), fvel[i].copy()
        rb.set_gripper_width, rb.grasp_flag = float(setw[i]), bool(grasp[i])
        rb.use_inv_dyn, rb.gravity_comp, rb.clip_rate, rb.clip_actions = False, True, False, False
        rb.time_keeper = TimeKeeper(0.001)
        rb.time_keeper.step_count = 5
        rb.uff_last = np.zeros(9)
        b = bias[i].copy()
        rb.get_command_from_inverse_dynamics = lambda target_j_acc, mj_calc_inv=False, b=b: b.copy()
        force[i] = rb.fing_ctrl_step()
        rb.preprocessCommand(tau[i].copy())
        uff[i] = rb.uff
        assert np.allclose(rb.finger_commands, force[i])
    np.savez_compressed(os.path.join(HERE, "ref_pd_finger.npz"), pd_q=q, pd_v=v, pd_qd=qd, pd_vd=vd, pd_u=u,
                        f_pos=fpos, f_vel=fvel, f_setw=setw, f_grasp=grasp, f_bias=bias, f_tau=tau,
                        f_force=force, f_uff=uff)
    print("pd/finger done")


def gen_offline_ik():
    targets = {"avoiding": [0.525, -0.28, 0.12], "sorting": [0.525, -0.3, 0.25], "stacking": [0.525, 0, 0.3]}
    out = {}
    for name, pos in targets.items():
        robot = FakeRobot(QDEF)
        gen = ctrl.OfflineIKTrajectoryGenerator()
        gen.setDesiredPos(np.array(pos + [0, 1, 0, 0], float))
        tracker = ctrl.JointPDController()
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            traj = gen.generate_trajectory(tracker, robot, 1.0)
        msg = buf.getvalue().strip()
        n_it = int(msg.split("(")[1].split(" ")[0])
        out[name + "__target"] = np.array(pos + [0, 1, 0, 0], float)
        out[name + "__qstar"] = np.asarray(gen.interpolation_trajectory.desiredPosition, float)
        out[name + "__traj_last"] = traj[-1].copy()
        out[name + "__iters"] = np.array(n_it)
        p, qt = CHAIN.fk(traj[-1])
        out[name + "__tcp"] = np.concatenate([p, qt])
        print(name, msg, traj[-1], p)
    np.savez_compressed(os.path.join(HERE, "ref_offline_ik.npz"), **out)


def gen_avoiding_task():
    from envs.gym_avoiding_env.gym_avoiding.envs.avoiding import ObstacleAvoidanceEnv
    import simulation.avoiding_sim as S
    import torch

    rng = np.random.default_rng(99)
    env = object.__new__(ObstacleAvoidanceEnv)
    # the geometry attributes are set in __init__ (avoiding.py:94-113); re-run that block by
    # executing __init__'s arithmetic through a light subclass is not possible (it builds a scene),
    # so set them from the same literals and let check_mode (the code under test) do the logic.
    level_distance, obstacle_offset = 0.18, 0.075
    env.l1_ypos = -0.1
    env.l2_ypos = -0.1 + level_distance
    env.l3_ypos = -0.1 + 2 * level_distance
    env.goal_ypos = -0.1 + 2.5 * level_distance
    env.l1_xpos = 0.5
    env.l2_top_xpos = 0.5 - obstacle_offset
    env.l2_bottom_xpos = 0.5 + obstacle_offset
    env.l3_top_xpos = 0.5 - 2 * obstacle_offset
    env.l3_mid_xpos = 0.5
    env.l3_bottom_xpos = 0.5 + 2 * obstacle_offset
    E, T = 200, 60
    paths = np.zeros((E, T, 2))
    modes = np.zeros((E, T, 9))
    succ = np.zeros((E, T), dtype=bool)
    for e in range(E):
        env.l1_passed = env.l2_passed = env.l3_passed = False
        env.mode_encoding = np.zeros(9)
        x = 0.5 + rng.uniform(-0.25, 0.25)
        xs = x + np.cumsum(rng.normal(scale=0.02, size=T))
        ys = np.linspace(-0.28, 0.45, T) + rng.normal(scale=0.01, size=T)
        if e < 20:  # exact-threshold cases
            xs[:] = rng.choice([0.35, 0.425, 0.5, 0.575, 0.65])
        robot = type("R", (), {})()
        env.robot = robot
        for t in range(T):
            robot.current_c_pos = np.array([xs[t], ys[t], 0.12])
            env.check_mode()
            paths[e, t] = xs[t], ys[t]
            modes[e, t] = env.mode_encoding
            succ[e, t] = env.check_success()
    # metric tail of Avoiding_Sim.test_agent with a stubbed rollout
    n = 480
    enc = np.zeros((n, 9))
    for i in range(n):
        enc[i, rng.integers(0, 2)] = 1
        enc[i, 2 + rng.integers(0, 3)] = 1
        enc[i, 5 + rng.integers(0, 4)] = 1
    su = (rng.uniform(size=n) < 0.7).astype(np.float32)

    def fake_eval(self, agent, n_trajectories, mode_encoding, successes, robot_c_pos, pid, cpu_set):
        mode_encoding[:] = torch.tensor(enc, dtype=torch.float32)
        successes[:] = torch.tensor(su)

    S.Avoiding_Sim.eval_agent = fake_eval
    S.wandb.log = lambda *a, **k: None
    sim = S.Avoiding_Sim(seed=0, device="cpu", render=False, n_cores=1, n_trajectories=n)
    with contextlib.redirect_stdout(io.StringIO()):
        successes, entropy = sim.test_agent(agent=None)
    np.savez_compressed(os.path.join(HERE, "ref_avoiding_task.npz"), paths=paths, modes=modes, succ=succ,
                        metric_enc=enc, metric_succ=su, metric_success_rate=float(successes.mean()),
                        metric_entropy=float(entropy))
    print("avoiding task: success_rate %.6f entropy %.9f" % (float(successes.mean()), float(entropy)))


def gen_pushing_task():
    """Block_Push_Env task logic (pushing.py:255-280,335-459) and the metric tail of Pushing_Sim.test_agent
    (pushing_sim.py:140-178), driven with synthetic box poses through a fake scene."""
    from envs.gym_pushing_env.gym_pushing.envs.pushing import Block_Push_Env
    import simulation.pushing_sim as S
    import torch

    rng = np.random.default_rng(7)
    env = object.__new__(Block_Push_Env)
    env.push_box1, env.push_box2, env.target_box_1, env.target_box_2 = "b1", "b2", "t1", "t2"
    env.target_min_dist = 0.05
    poses = {"t1": (np.array([0.42, 0.3, 0.0]), np.array([0.0, 1, 0, 0])),
             "t2": (np.array([0.63, 0.3, 0.0]), np.array([0.0, 1, 0, 0]))}
    scene = type("S", (), {})()
    scene.get_obj_pos = lambda o: poses[o][0].copy()
    scene.get_obj_quat = lambda o: poses[o][1].copy()
    env.scene = scene
    tcp = np.zeros(3)
    env.robot_state = lambda: tcp.copy()
    E, T = 70, 64
    box = np.zeros((E, T, 2, 7))
    rob = np.zeros((E, T, 3))
    obs = np.zeros((E, T, 8), dtype=np.float32)
    mode = np.zeros((E, T), dtype=np.int64)
    first = np.zeros((E, T), dtype=np.int64)
    meand = np.zeros((E, T))
    succ = np.zeros((E, T), dtype=bool)
    rew = np.zeros((E, T))
    tg = [poses["t1"][0], poses["t2"][0]]
    for e in range(E):
        env.first_visit = -1
        env.terminated = False
        # each box is dragged towards a target (a random pairing and visiting order) with noise, then on to the other
        pairing = rng.integers(0, 2)           # 0: b1->t1,b2->t2 ; 1: b1->t2,b2->t1
        order = rng.integers(0, 2)             # which box moves first
        p = [np.array([rng.uniform(0.4, 0.5), rng.uniform(-0.15, 0.0), rng.uniform(0.0, 0.02)]),
             np.array([rng.uniform(0.55, 0.65), rng.uniform(-0.15, 0.0), rng.uniform(0.0, 0.02)])]
        yaw = [rng.uniform(-np.pi, np.pi), rng.uniform(-np.pi, np.pi)]
        tilt = [rng.normal(scale=0.02, size=2) if e % 3 == 0 else np.zeros(2) for _ in range(2)]
        for t in range(T):
            mover = order if t < T // 2 else 1 - order
            goal = tg[mover ^ pairing] + np.array([0, 0, 0.011])
            d = goal - p[mover]
            p[mover] = p[mover] + 0.09 * d + rng.normal(scale=0.004, size=3) * (e % 4 != 1)
            if e % 5 == 4 and t > T // 3:   # wander off again
                p[mover] += rng.normal(scale=0.03, size=3)
            for k in range(2):
                yaw[k] += rng.normal(scale=0.05)
                q = np.array([np.cos(yaw[k] / 2), tilt[k][0], tilt[k][1], np.sin(yaw[k] / 2)])
                if e % 7 == 0:
                    q *= 1 + 1e-3 * rng.normal()      # un-normalised quaternion (qpos is read raw)
                poses["b%d" % (k + 1)] = (p[k].copy(), q)
                box[e, t, k, :3], box[e, t, k, 3:] = p[k], q
            tcp[:] = rng.uniform([0.3, -0.4, 0.1], [0.8, 0.45, 0.14])
            rob[e, t] = tcp
            obs[e, t] = env.get_observation()
            succ[e, t] = env._check_early_termination()
            mode[e, t], meand[e, t] = env.check_mode()
            first[e, t] = env.first_visit
            rew[e, t] = env.get_reward()
    # metric tail with a stubbed rollout
    nc, nt = 30, 16
    me = rng.integers(-1, 4, size=(nc, nt)).astype(np.float32)
    su = (rng.uniform(size=(nc, nt)) < 0.6).astype(np.float32)
    su[3] = 0
    md = rng.uniform(0, 0.3, size=(nc, nt)).astype(np.float32)

    def fake_eval(self, agent, contexts, n_trajectories, mode_encoding, successes, mean_distance, pid, cpu_set):
        mode_encoding[:] = torch.tensor(me)
        successes[:] = torch.tensor(su)
        mean_distance[:] = torch.tensor(md)

    logged = {}
    S.Pushing_Sim.eval_agent = fake_eval
    S.wandb.log = lambda d, *a, **k: logged.update(d)
    sim = S.Pushing_Sim(seed=0, device="cpu", render=False, n_cores=1, n_contexts=nc, n_trajectories_per_context=nt)
    with contextlib.redirect_stdout(io.StringIO()):
        sim.test_agent(agent=None)
    ctx = np.load(os.path.join(ref_shims.REF, "environments/dataset/data/pushing/test_contexts.pkl"), allow_pickle=True)
    ctx_arr = np.array([np.concatenate([np.asarray(a, dtype=np.float64) for a in c]) for c in ctx])   # [60][3+4+3+4]
    np.savez_compressed(os.path.join(HERE, "ref_pushing_task.npz"), box=box, rob=rob, obs=obs, mode=mode, first=first,
                        mean_distance=meand, succ=succ, reward=rew, metric_mode=me, metric_succ=su, metric_dist=md,
                        metric_success_rate=float(logged["Metrics/successes"]), metric_entropy=float(logged["Metrics/entropy"]),
                        metric_distance=float(logged["Metrics/distance"]), metric_score=float(logged["score"]),
                        test_contexts=ctx_arr)
    print("pushing task: success %.4f entropy %.6f; modes seen %s; successes %d" % (
        logged["Metrics/successes"], logged["Metrics/entropy"], np.unique(mode), succ.sum()))


def gen_sorting_stacking_metrics():
    """Metric tails of Sorting_Sim.test_agent (sorting_sim.py:191-221) and Stacking_Sim.test_agent / cal_KL
    (stacking_sim.py:143-167,226-257) with stubbed rollouts.  Sorting's context / mode-prior files are not part of the
    reference checkout, so its sim object is built without __init__ and given a synthetic prior."""
    import simulation.sorting_sim as SS
    import simulation.stacking_sim as ST
    import torch

    rng = np.random.default_rng(11)
    out = {}
    # ---- sorting: 4 boxes -> modes are np.packbits of 4 bits (decode_mode), keys arbitrary ints
    nc, nt = 20, 12
    keys = np.array([48, 80, 96, 144, 160, 192])            # the six orders of two red / two blue boxes, packed
    prior = rng.dirichlet(np.ones(len(keys))).astype(np.float64)
    me = rng.choice(np.concatenate([keys, [112, 240]]), size=(nc, nt)).astype(np.float32)
    su = (rng.uniform(size=(nc, nt)) < 0.65).astype(np.float32)
    su[5] = 0
    sim = object.__new__(SS.Sorting_Sim)
    sim.n_contexts, sim.n_trajectories_per_context, sim.n_cores = nc, nt, 1
    sim.mode_keys, sim.n_mode, sim.mode_encoding = keys, len(keys), torch.tensor(prior)

    def fake_eval(self, agent, contexts, n_trajectories, mode_encoding, successes, mean_distance, pid, cpu_set):
        mode_encoding[:] = torch.tensor(me)
        successes[:] = torch.tensor(su)

    logged = {}
    SS.Sorting_Sim.eval_agent = fake_eval
    SS.wandb.log = lambda d, *a, **k: logged.update(d)
    with contextlib.redirect_stdout(io.StringIO()):
        sim.test_agent(agent=None)
    out.update(sort_keys=keys, sort_prior=prior, sort_mode=me, sort_succ=su, sort_success_rate=float(logged["Metrics/successes"]),
               sort_entropy=float(logged["Metrics/entropy"]), sort_KL=float(logged["Metrics/KL"]), sort_score=float(logged["score"]))
    # ---- stacking: real constructor (test_contexts.pkl and mode_prob.pkl are in the checkout)
    nc, nt = 15, 10
    m3 = rng.integers(0, 6, size=(nc, nt)).astype(np.float32)
    m2 = rng.integers(0, 6, size=(nc, nt)).astype(np.float32)
    m1 = rng.integers(0, 3, size=(nc, nt)).astype(np.float32)
    s1 = (rng.uniform(size=(nc, nt)) < 0.8).astype(np.float32)
    s2 = s1 * (rng.uniform(size=(nc, nt)) < 0.7)
    s3 = s2 * (rng.uniform(size=(nc, nt)) < 0.5)
    s3[2] = 0

    def fake_eval2(self, agent, contexts, n_trajectories, mode_encoding, mode_encoding_1_box, mode_encoding_2_box, successes, successes_1,
                   successes_2, pid, cpu_set):
        mode_encoding[:] = torch.tensor(m3); mode_encoding_2_box[:] = torch.tensor(m2); mode_encoding_1_box[:] = torch.tensor(m1)
        successes[:] = torch.tensor(s3.astype(np.float32)); successes_1[:] = torch.tensor(s1); successes_2[:] = torch.tensor(s2.astype(np.float32))

    logged2 = {}
    ST.Stacking_Sim.eval_agent = fake_eval2
    ST.wandb.log = lambda d, *a, **k: logged2.update(d)
    sim2 = ST.Stacking_Sim(seed=0, device="cpu", render=False, n_cores=1, n_contexts=nc, n_trajectories_per_context=nt)
    with contextlib.redirect_stdout(io.StringIO()):
        sim2.test_agent(agent=None)
    out.update(stack_m1=m1, stack_m2=m2, stack_m3=m3, stack_s1=s1, stack_s2=s2.astype(np.float32), stack_s3=s3.astype(np.float32),
               stack_prior3=sim2.mode_encoding_3.numpy(), stack_prior2=sim2.mode_encoding_2.numpy(), stack_prior1=sim2.mode_encoding_1.numpy(),
               **{"stack_" + k.split("/")[-1]: float(v) for k, v in logged2.items()})
    np.savez_compressed(os.path.join(HERE, "ref_sorting_stacking_metrics.npz"), **out)
    print("sorting: success %.4f entropy %.6f KL %.6f | stacking:" % (out["sort_success_rate"], out["sort_entropy"], out["sort_KL"]),
          {k: round(v, 5) for k, v in out.items() if k.startswith("stack_") and np.ndim(v) == 0})


def gen_sort_stack_task():
    """Sorting_Env / Stacking env task logic (sorting.py:308-390,444-543; stacking.py:228-277,395-447) driven with synthetic box
    poses through a fake scene.  Boxes that are not in the model get one constant pose, as MjScene does for body id -1."""
    from envs.gym_sorting_env.gym_sorting.envs.sorting import Sorting_Env
    from envs.gym_stacking_env.gym_stacking.envs.stacking import CubeStacking_Env

    rng = np.random.default_rng(21)
    out = {}
    names = ["r1", "r2", "r3", "b1", "b2", "b3"]
    bogus = (np.array([0.0088, -0.0001, 0.0584]), np.array([1.0, 0, 0, 0]))
    for nb in (2, 4, 6):
        env = object.__new__(Sorting_Env)
        env.num_boxes, env.if_vision = nb, False
        env.red_box_1, env.red_box_2, env.red_box_3, env.blue_box_1, env.blue_box_2, env.blue_box_3 = names
        env.red_target_pos, env.blue_target_pos = np.array([0.4, 0.32]), np.array([0.625, 0.32])
        poses = {}
        scene = type("S", (), {})()
        scene.get_obj_pos = lambda o: poses[o][0].copy()
        scene.get_obj_quat = lambda o: poses[o][1].copy()
        env.scene = scene
        tcp = np.zeros(3)
        env.robot_state = lambda: tcp.copy()
        E, T = 18, 56
        present = names[:nb // 2] + names[3:3 + nb // 2]
        box = np.zeros((E, T, 6, 7)); rob = np.zeros((E, T, 3))
        obs = np.zeros((E, T, 2 + 3 * nb), dtype=np.float32); succ = np.zeros((E, T), dtype=bool); code = np.zeros((E, T), dtype=np.int64)
        for e in range(E):
            env.mode = np.array([-1, -1, -1, -1, -1, -1]); env.mode_step = 0; env.min_inds = []; env.terminated = False
            p = {k: np.array([rng.uniform(0.4, 0.65), rng.uniform(-0.15, 0.1), 0.13]) for k in present}
            yaw = {k: rng.uniform(-np.pi, np.pi) for k in present}
            order = list(rng.permutation(present))
            wrong = e % 5 == 0                      # sometimes a cube goes into the other colour's bin
            for t in range(T):
                mover = order[min(len(order) - 1, t * len(order) // T)]
                red = mover.startswith("r")
                goal = np.array([0.4 if (red != wrong) else 0.625, 0.315, -0.008]) + 0.0
                p[mover] = p[mover] + 0.12 * (goal - p[mover]) + rng.normal(scale=0.004, size=3) * (e % 3 != 1)
                for k in names:
                    if k in present:
                        yaw[k] += rng.normal(scale=0.05)
                        poses[k] = (p[k].copy(), np.array([np.cos(yaw[k] / 2), 0.01 * (e % 2), 0, np.sin(yaw[k] / 2)]))
                    else:
                        poses[k] = (bogus[0].copy(), bogus[1].copy())
                    box[e, t, names.index(k), :3], box[e, t, names.index(k), 3:] = poses[k]
                tcp[:] = rng.uniform([0.3, -0.4, 0.2], [0.8, 0.45, 0.3]); rob[e, t] = tcp
                obs[e, t] = env.get_observation()
                succ[e, t] = env._check_early_termination()
                mode, min_inds = env.check_mode()
                code[e, t] = env.decode_mode(mode[:nb])
        out.update({"sort%d_box" % nb: box, "sort%d_rob" % nb: rob, "sort%d_obs" % nb: obs, "sort%d_succ" % nb: succ, "sort%d_code" % nb: code})
        print("sorting-%d: codes %s successes %d" % (nb, np.unique(code)[:12], succ.sum()))
    # ---- stacking
    env = object.__new__(CubeStacking_Env)
    env.if_vision = False
    env.red_box, env.green_box, env.blue_box, env.target_box = "r", "g", "b", "t"
    env.pos_min_dist = 0.06
    poses = {"t": (np.array([0.55, 0.2, 0.0]), np.array([1.0, 0, 0, 0]))}
    scene = type("S", (), {})()
    scene.get_obj_pos = lambda o: poses[o][0].copy()
    scene.get_obj_quat = lambda o: poses[o][1].copy()
    env.scene = scene
    env.robot_state = lambda: (np.zeros(8), np.zeros(7), np.array([0.0, 1, 0, 0]))
    E, T = 24, 60
    box = np.zeros((E, T, 3, 7)); obs = np.zeros((E, T, 12), dtype=np.float32); succ = np.zeros((E, T), dtype=bool)
    md = np.zeros((E, T)); modes = np.zeros((E, T), dtype="U3")
    for e in range(E):
        env.min_inds = []; env.mode_encoding = []; env.terminated = False
        p = {k: np.array([rng.uniform(0.35, 0.65), rng.uniform(-0.2, 0.0), 0.03]) for k in "rgb"}
        yaw = {k: rng.uniform(-np.pi, np.pi) for k in "rgb"}
        order = list(rng.permutation(list("rgb")))
        for t in range(T):
            lvl = min(2, t * 3 // T)
            mover = order[lvl]
            goal = np.array([0.55, 0.2, 0.03 + 0.06 * lvl * (e % 4 != 3)]) + rng.normal(scale=0.01, size=3) * (e % 6 == 5)
            p[mover] = p[mover] + 0.15 * (goal - p[mover]) + rng.normal(scale=0.003, size=3)
            for i, k in enumerate("rgb"):
                yaw[k] += rng.normal(scale=0.04)
                poses[k] = (p[k].copy(), np.array([np.cos(yaw[k] / 2), 0, 0.01 * (e % 2), np.sin(yaw[k] / 2)]))
                box[e, t, i, :3], box[e, t, i, 3:] = poses[k]
            obs[e, t] = env.get_observation()
            succ[e, t] = env._check_early_termination()
            m, md[e, t] = env.check_mode()
            modes[e, t] = "".join(m)
    out.update(stack_box=box, stack_obs=obs, stack_succ=succ, stack_md=md, stack_mode=modes, stack_target=poses["t"][0])
    print("stacking: modes %s successes %d" % (np.unique(modes[:, -1]), succ.sum()))
    np.savez_compressed(os.path.join(HERE, "ref_sort_stack_task.npz"), **out)


def gen_aligning_task():
    """Robot_Push_Env task logic of the Aligning task (aligning.py:223-340: observation, check_mode, reward, early termination) and the metric
    tail of Aligning_Sim.test_agent (aligning_sim.py:160-204, two behaviour modes: pushing from inside / from outside), driven with synthetic
    box / target poses through a fake scene; plus the reference's test contexts as data."""
    from envs.gym_aligning_env.gym_aligning.envs.aligning import Robot_Push_Env, rotation_distance
    import simulation.aligning_sim as S
    import torch

    rng = np.random.default_rng(11)
    env = object.__new__(Robot_Push_Env)
    env.push_box, env.target_box = "box", "target"
    env.if_vision = False
    env.pos_min_dist, env.rot_min_dist, env.robot_box_dist = 0.018, 0.048, 0.051          # aligning.py:215-218
    poses = {}
    scene = type("S", (), {})()
    scene.get_obj_pos = lambda o: poses[o][0].copy()
    scene.get_obj_quat = lambda o: poses[o][1].copy()
    env.scene = scene
    tcp = np.zeros(3)
    env.robot_state = lambda: tcp.copy()
    E, T = 60, 48
    box = np.zeros((E, T, 7)); target = np.zeros((E, 7)); rob = np.zeros((E, T, 3))
    obs = np.zeros((E, T, 17), dtype=np.float32)
    mode = np.zeros((E, T), dtype=np.int64); meand = np.zeros((E, T)); succ = np.zeros((E, T), dtype=bool); rew = np.zeros((E, T))

    def yawq(a):
        return np.array([np.cos(a / 2), 0.0, 0.0, np.sin(a / 2)])

    for e in range(E):
        env.terminated = False
        tp = np.array([rng.uniform(0.4, 0.6), rng.uniform(0.2, 0.35), 0.0]); ta = rng.uniform(-np.pi / 2, np.pi / 2)
        poses["target"] = (tp, yawq(ta)); target[e, :3], target[e, 3:] = tp, yawq(ta)
        p = np.array([rng.uniform(0.4, 0.6), rng.uniform(-0.25, -0.1), rng.uniform(0.0, 0.012)]); a = rng.uniform(-np.pi / 2, np.pi / 2)
        for t in range(T):
            # the box drifts to the target pose (position and yaw), some episodes with noise, a sign-flipped or un-normalised quaternion, a tilt
            p = p + 0.12 * (tp + np.array([0, 0, 0.0105]) - p) + rng.normal(scale=0.002, size=3) * (e % 3 != 0)
            a = a + 0.12 * (ta - a) + rng.normal(scale=0.01) * (e % 3 != 0)
            q = yawq(a)
            if e % 4 == 1:
                q = -q
            if e % 5 == 2:
                q = q * (1 + 1e-3 * rng.normal())
            if e % 6 == 3:
                q = q + np.array([0, 0.01, -0.01, 0]) * rng.normal()
            poses["box"] = (p.copy(), q)
            box[e, t, :3], box[e, t, 3:] = p, q
            # the rod: inside the box walls, just outside them, or far away
            r = rng.uniform(0, 0.1) if t % 3 else rng.uniform(0.045, 0.057)
            ang = rng.uniform(0, 2 * np.pi)
            tcp[:] = [p[0] + r * np.cos(ang), p[1] + r * np.sin(ang), rng.uniform(0.02, 0.25)]
            rob[e, t] = tcp
            obs[e, t] = env.get_observation()
            succ[e, t] = env._check_early_termination()
            mode[e, t], meand[e, t] = env.check_mode()
            rew[e, t] = env.get_reward()
    rd = np.array([rotation_distance(box[e, t, 3:], target[e, 3:]) for e in range(E) for t in range(T)]).reshape(E, T)
    # metric tail with a stubbed rollout
    nc, nt = 30, 8
    me = rng.integers(0, 2, size=(nc, nt)).astype(np.float32)
    su = (rng.uniform(size=(nc, nt)) < 0.6).astype(np.float32)
    su[5] = 0
    md = rng.uniform(0, 0.3, size=(nc, nt)).astype(np.float32)

    def fake_eval(self, agent, contexts, n_trajectories, mode_encoding, successes, mean_distance, pid, cpu_set):
        mode_encoding[:] = torch.tensor(me)
        successes[:] = torch.tensor(su)
        mean_distance[:] = torch.tensor(md)

    logged = {}
    S.Aligning_Sim.eval_agent = fake_eval
    S.wandb.log = lambda d, *a, **k: logged.update(d)
    sim = S.Aligning_Sim(seed=0, device="cpu", render=False, n_cores=1, n_contexts=nc, n_trajectories_per_context=nt)
    with contextlib.redirect_stdout(io.StringIO()):
        sim.test_agent(agent=None)
    ctx = np.load(os.path.join(ref_shims.REF, "environments/dataset/data/aligning/test_contexts.pkl"), allow_pickle=True)
    ctx_arr = np.array([np.concatenate([np.asarray(a_, dtype=np.float64).reshape(-1) for a_ in c]) for c in ctx])   # [n][3 + 4 + 3 + 4]: (x, y, yaw deg), quat, target ...
    np.savez_compressed(os.path.join(HERE, "ref_aligning_task.npz"), box=box, target=target, rob=rob, obs=obs, mode=mode, mean_distance=meand, succ=succ,
                        reward=rew, rot_dist=rd, metric_mode=me, metric_succ=su, metric_dist=md,
                        metric_success_rate=float(logged["Metrics/successes"]), metric_entropy=float(logged["Metrics/entropy"]),
                        metric_distance=float(logged["Metrics/distance"]), metric_score=float(logged["score"]), test_contexts=ctx_arr)
    print("aligning task: success %.4f entropy %.6f; modes seen %s; successes %d; %d test contexts" % (
        logged["Metrics/successes"], logged["Metrics/entropy"], np.unique(mode), succ.sum(), len(ctx_arr)))


def gen_inserting_task():
    """Gate_Insertion_Env task logic (gate_insertion.py:278-309 get_observation, :386-409 step's info, :411-446 check_mode / check_mean_dist,
    :475-498 _check_early_termination), driven with synthetic box poses through a fake scene.  The reference has no Sim class for this task."""
    from envs.gym_inserting_env.gym_inserting.envs.gate_insertion import Gate_Insertion_Env

    rng = np.random.default_rng(33)
    env = object.__new__(Gate_Insertion_Env)
    env.push_box1, env.push_box2, env.push_box3 = "b1", "b2", "b3"
    env.target_box1, env.target_box2, env.target_box3 = "t1", "t2", "t3"
    env.target_min_dist = 0.01
    env.mode_dict = {'rgb': 1, 'rbg': 2, 'grb': 3, 'gbr': 4, 'brg': 5, 'bgr': 6}
    tg = [np.array([0.3575, 0.276, 0.0]), np.array([0.525, 0.4535, 0.0]), np.array([0.6925, 0.276, 0.0])]
    poses = {"t%d" % (i + 1): (tg[i].copy(), np.array([0.0, 1, 0, 0])) for i in range(3)}
    scene = type("S", (), {})()
    scene.get_obj_pos = lambda o: poses[o][0].copy()
    scene.get_obj_quat = lambda o: poses[o][1].copy()
    env.scene = scene
    tcp = np.zeros(3)
    env.robot_state = lambda: tcp.copy()
    E, T = 48, 72
    box = np.zeros((E, T, 3, 7)); rob = np.zeros((E, T, 3)); obs = np.zeros((E, T, 11), dtype=np.float32)
    succ = np.zeros((E, T), dtype=bool); meand = np.zeros((E, T)); nmode = np.zeros((E, T), dtype=np.int64); code = np.zeros((E, T), dtype=np.int64)
    rew = np.zeros((E, T))
    for e in range(E):
        env.modes = []; env.terminated = False
        p = [np.array([rng.uniform(0.35, 0.5), rng.uniform(-0.2, -0.15), -0.0072]), np.array([rng.uniform(0.55, 0.7), rng.uniform(-0.1, -0.05), -0.0072]),
             np.array([rng.uniform(0.35, 0.5), rng.uniform(0.0, 0.05), -0.0072])]
        yaw = [rng.uniform(-np.pi / 2, np.pi / 2) for _ in range(3)]
        order = list(rng.permutation(3))
        stop_short = e % 6 == 5                                  # the last box never arrives
        for t in range(T):
            k = order[min(2, 3 * t // T)]
            goal = tg[k] + np.array([0, 0, -0.0072 if e % 4 else 0.004])       # the z offset enters the 3-D distance (obj_distance on full positions)
            if stop_short and k == order[2]:
                goal = goal + np.array([0.03, 0, 0])
            p[k] = p[k] + 0.2 * (goal - p[k]) + rng.normal(scale=0.0006, size=3) * (e % 3 != 1)
            if e % 7 == 6 and t % 24 == 23:                      # a box is knocked out of its goal again: the mode list keeps it (sticky)
                p[k] += np.array([0.05, 0, 0])
            for j in range(3):
                yaw[j] += rng.normal(scale=0.03)
                q = np.array([np.cos(yaw[j] / 2), 0.003 * (e % 2), 0, np.sin(yaw[j] / 2)])
                poses["b%d" % (j + 1)] = (p[j].copy(), q)
                box[e, t, j, :3], box[e, t, j, 3:] = p[j], q
            tcp[:] = rng.uniform([0.3, -0.4, 0.1], [0.8, 0.5, 0.14]); rob[e, t] = tcp
            obs[e, t] = env.get_observation()
            succ[e, t] = env._check_early_termination()
            meand[e, t] = env.check_mean_dist()
            mode = ''.join(env.check_mode())
            nmode[e, t] = len(mode)
            code[e, t] = env.mode_dict[mode] if len(mode) == 3 else 0
            rew[e, t] = env.get_reward()
    np.savez_compressed(os.path.join(HERE, "ref_inserting_task.npz"), box=box, rob=rob, obs=obs, succ=succ, mean_distance=meand, n_mode=nmode, code=code, reward=rew)
    print("inserting task: successes %d, full mode codes %s, mode lengths %s" % (succ.sum(), np.unique(code), np.unique(nmode)))


if __name__ == "__main__":
    gen_ik()
    gen_pd_finger()
    gen_offline_ik()
    gen_avoiding_task()
    gen_pushing_task()
    gen_sorting_stacking_metrics()
    gen_sort_stack_task()
    gen_aligning_task()
    gen_inserting_task()
