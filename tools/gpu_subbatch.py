"""Does stepping the 4096 environments as S independent sub-batches on S streams raise the throughput?  A launch lasts as long as its slowest workgroup
(an environment in a rare path: rod contact, clipped-eigenvalue IK, a hard island); sub-batches do not wait for each other's tails.
Avoiding: device random policy + auto-reset (the bench loop); Sorting / Pushing: stand-in MLP + auto-reset.
    python tools/gpu_subbatch.py avoiding|sorting|pushing [--envs 4096] [--steps 300]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

task = sys.argv[1] if len(sys.argv) > 1 else "avoiding"
N = int(sys.argv[sys.argv.index("--envs") + 1]) if "--envs" in sys.argv else 4096
K = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 300
FUSED, TALLY, GRAPH = "--fused" in sys.argv, "--tally" in sys.argv, "--graph" in sys.argv
SLIST = [int(x) for x in sys.argv[sys.argv.index("--s") + 1].split(",")] if "--s" in sys.argv else [1, 2, 4, 8]
dev = torch.device("cuda:0")


class Shard:
    def __init__(self, n, off):
        self.n, self.off = n, off
        self.stream = torch.cuda.Stream(dev)
        with torch.cuda.stream(self.stream):
            if task == "avoiding":
                from d3il_amd.envs.avoiding import ObstacleAvoidanceVecEnv
                self.env, ctx = ObstacleAvoidanceVecEnv(n, device=dev), None
            elif task == "sorting":
                from d3il_amd.envs.sorting import SortingVecEnv, sample_contexts
                self.env, ctx = SortingVecEnv(n, device=dev, max_steps_per_episode=700), sample_contexts(60, 4, seed=0)
            else:
                from d3il_amd.envs.pushing import BlockPushVecEnv
                from d3il_amd.simulation.pushing_sim import load_test_contexts
                self.env, ctx = BlockPushVecEnv(n, device=dev), load_test_contexts()
            env = self.env
            env.start()
            if ctx is not None:
                ids = (off + np.arange(n)) % len(ctx)
                env.reset(context=ctx[ids])
            else:
                env.reset()
            env.policy_begin()
            if TALLY:
                self.table = env.set_tally(len(ctx) if ctx is not None else 1, None if ctx is None else torch.as_tensor(ids, dtype=torch.int32, device=dev))
            if FUSED:
                env.bind_stream(self.stream)
            self.episodes = torch.zeros(2, dtype=torch.int64, device=dev)
            self.actions = torch.zeros(n, 7, dtype=torch.float64, device=dev)
            self.actions[:, 3:] = torch.tensor([0.0, 1.0, 0.0, 0.0], dtype=torch.float64, device=dev)
            self.pol = None
            if task != "avoiding":
                from d3il_amd.agents import RandomResidualMLPPolicy
                self.pol = RandomResidualMLPPolicy(input_dim=2 + env.obs.shape[1], device=dev)
            ms = env.max_steps_per_episode
            env.step_count[:n] = ((torch.arange(n, device=dev, dtype=torch.int64) + off) * 977 % ms).to(torch.int32)
        self.t = 0
        self.graph = None
        if GRAPH and self.pol is not None:      # the policy's ~25 small torch kernels of one step as ONE graph replay on this sub-batch's stream
            env.bind_stream(self.stream)
            torch.cuda.synchronize()
            with torch.cuda.stream(self.stream):
                for _ in range(3):
                    self._policy()
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=self.stream):
                self._policy()
            torch.cuda.synchronize()

    def _policy(self):
        env, n = self.env, self.n
        des_xy = env.policy_des[:2, :n]
        obs_in = torch.cat((des_xy.t(), env.obs.to(torch.float64)), dim=1)
        des_xy.add_(self.pol.predict_batch(obs_in).to(torch.float64).t())
        self.actions[:, 0:2] = des_xy.t()
        self.actions[:, 2] = env.policy_des[2, :n]

    def step(self):
        if self.graph is not None:
            torch.cuda.set_stream(self.stream)
            self.graph.replay()
            self.env.step_auto_reset(self.actions, self.episodes)
            self.t += 1
            return
        return self._step_eager()

    def _step_eager(self):
        env, n = self.env, self.n
        if FUSED and self.pol is None:
            env.random_rollout_step(42, self.off, self.t, self.actions, self.episodes)
            self.t += 1
            return
        with torch.cuda.stream(self.stream):
            if self.pol is None:
                env.policy_action(42, self.off, self.t, self.actions)
            else:
                des_xy = env.policy_des[:2, :n]
                obs_in = torch.cat((des_xy.t(), env.obs.to(torch.float64)), dim=1)
                des_xy.add_(self.pol.predict_batch(obs_in).to(torch.float64).t())
                self.actions[:, 0:2] = des_xy.t()
                self.actions[:, 2] = env.policy_des[2, :n]
            env.step(self.actions)
            env.auto_reset(self.episodes)
        self.t += 1


for S in SLIST:
    shards = [Shard(N // S, i * (N // S)) for i in range(S)]
    pre = {"avoiding": 250, "sorting": 300, "pushing": 400}[task]
    for t in range(pre):
        for sh in shards: sh.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(K):
        for sh in shards: sh.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ep = sum(int(sh.episodes[0]) for sh in shards)
    print("%s, %d envs as %d sub-batch(es) on %d stream(s): %.3f ms per global step, %.3f M env-steps/s, %d episodes finished" % (task, N, S, S, dt / K * 1e3, N * K / dt / 1e6, ep))
    for sh in shards: sh.env.close()
