"""Golden vectors for the batched policies: the reference's own agents (shim-imported, fixed-seed random weights) are rolled out
batch-1, environment by environment, and their inputs, weights (tensors = data), noise and outputs are stored.

Run in the build container only (needs /root/reference):  python tests/golden/gen_agent_goldens.py
Output (committed): tests/golden/ref_agents.npz.  Pins, against the actual reference code:
  * BC_Agent.predict            (agents/bc_agent.py:240-271)     with ResidualMLPNetwork (agents/models/common/mlp.py:114-182)
  * DiffusionAgent.predict      (agents/ddpm_agent.py:213-274)   with Diffusion / DiffusionMLPNetwork (gc_diffusion.py, diffusion_models.py)
  * BesoAgent.predict           (agents/beso_agent.py:316-443)   with GCDenoiser / DiffusionGPT and sample_euler_ancestral
  * Scaler                      (agents/utils/scaler.py:10-113)
The agents are built without Hydra: ``hydra.utils.instantiate`` is replaced by a ten-line resolver of ``_target_`` dictionaries, the
constructors that need datasets are bypassed with ``object.__new__`` and the attributes ``predict`` reads are set by hand.
The Gaussian noise of the samplers comes from a bank (torch.randn / randn_like are patched while the reference runs), so that the
batched implementation can be fed the identical noise: call j of environment i draws bank[j][i].
"""
import importlib
import os
import sys
from collections import deque

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims._StubFinder.ROOTS = ref_shims._StubFinder.ROOTS + ("hydra", "omegaconf", "torchsde", "torchdiffeq")
ref_shims.install()
import hydra  # noqa: E402  (stub)


def instantiate(cfg, *args, **kwargs):
    cfg = dict(cfg)
    target = cfg.pop("_target_")
    cfg.pop("_recursive_", None)
    mod, name = target.rsplit(".", 1)
    cfg.update(kwargs)
    return getattr(importlib.import_module(mod), name)(*args, **cfg)


hydra.utils.instantiate = instantiate

import agents.bc_agent as bc_mod  # noqa: E402
import agents.beso_agent as beso_mod  # noqa: E402
import agents.ddpm_agent as ddpm_mod  # noqa: E402
from agents.models.beso.agents.diffusion_agents.k_diffusion.score_wrappers import GCDenoiser  # noqa: E402
from agents.models.common.mlp import ResidualMLPNetwork  # noqa: E402
from agents.models.diffusion.ema import ExponentialMovingAverage  # noqa: E402
from agents.models.diffusion.gc_diffusion import Diffusion  # noqa: E402
from agents.utils.scaler import Scaler  # noqa: E402

N_ENV, T_STEPS = 5, 7


class NoiseBank:
    """Patches torch.randn / torch.randn_like: call number j (per environment rollout) returns bank[j] rows [env : env + batch]."""

    def __init__(self, seed, n_calls, width):
        g = torch.Generator().manual_seed(seed)
        self.bank = torch.randn(n_calls, N_ENV, 8, width, generator=g)     # [call, env, seq, dim]
        self.env, self.call = 0, 0

    def take(self, shape):
        shape = tuple(shape)
        b = self.bank[self.call, self.env]
        self.call += 1
        if len(shape) == 2:
            return b[0, :shape[1]].reshape(1, shape[1]).clone()
        return b[:shape[1], :shape[2]].reshape(1, shape[1], shape[2]).clone()

    def __enter__(self):
        self._r, self._rl = torch.randn, torch.randn_like
        torch.randn = lambda *s, **k: self.take(s[0] if len(s) == 1 and not isinstance(s[0], int) else s)
        torch.randn_like = lambda x, **k: self.take(x.shape)
        return self

    def __exit__(self, *a):
        torch.randn, torch.randn_like = self._r, self._rl


def make_scaler(obs_dim, act_dim, seed):
    rng = np.random.default_rng(seed)
    x = rng.normal(size=(400, obs_dim)) * rng.uniform(0.05, 0.5, obs_dim) + rng.normal(size=obs_dim) * 0.3
    y = rng.normal(size=(400, act_dim)) * 0.004
    return Scaler(x.astype(np.float64), y.astype(np.float64), True, "cpu")


def sd_arrays(prefix, sd):
    return {prefix + k.replace(".", "__"): v.detach().cpu().numpy() for k, v in sd.items()}


def main():
    out = {}
    # ---------------------------------------------------------------- BC (Pushing: obs 10 -> act 2)
    torch.manual_seed(1)
    bc = object.__new__(bc_mod.BC_Agent)
    bc.device = "cpu"
    bc.model = ResidualMLPNetwork(input_dim=10, hidden_dim=32, num_hidden_layers=4, output_dim=2, dropout=0, activation="Mish", device="cpu")
    bc.scaler = make_scaler(10, 2, 1)
    bc.min_action = torch.from_numpy(bc.scaler.y_bounds[0, :]).to("cpu")
    bc.max_action = torch.from_numpy(bc.scaler.y_bounds[1, :]).to("cpu")
    obs = np.random.default_rng(2).normal(size=(N_ENV, T_STEPS, 10)) * 0.3
    ref = np.zeros((N_ENV, T_STEPS, 2))
    for e in range(N_ENV):
        for t in range(T_STEPS):
            ref[e, t] = np.asarray(bc.predict(obs[e, t])).reshape(-1)
    out.update(sd_arrays("bc_sd__", bc.model.state_dict()))
    out.update(bc_obs=obs, bc_ref=ref, bc_x_mean=bc.scaler.x_mean.numpy(), bc_x_std=bc.scaler.x_std.numpy(), bc_y_mean=bc.scaler.y_mean.numpy(),
               bc_y_std=bc.scaler.y_std.numpy(), bc_y_bounds=bc.scaler.y_bounds)
    # ---------------------------------------------------------------- DDPM (Sorting-4: obs 16 -> act 2, n_timesteps 4, t_dim 8; scripts/sorting_4/ddpm_benchmark.sh)
    torch.manual_seed(3)
    dd = object.__new__(ddpm_mod.DiffusionAgent)
    dd.device = "cpu"
    dd.model = Diffusion(state_dim=16, action_dim=2, beta_schedule="cosine", n_timesteps=4, loss_type="l2", clip_denoised=True, predict_epsilon=True, device="cpu",
                         model=dict(_target_="agents.models.diffusion.diffusion_models.DiffusionMLPNetwork", action_dim=2, obs_dim=16, t_dim=8, residual_style=True,
                                    hidden_dim=32, num_hidden_layers=4, dropout=0, activation="Mish", device="cpu", goal_conditioned=False))
    dd.scaler = make_scaler(16, 2, 3)
    dd.model.min_action = torch.from_numpy(dd.scaler.y_bounds[0, :]).to("cpu")
    dd.model.max_action = torch.from_numpy(dd.scaler.y_bounds[1, :]).to("cpu")
    dd.window_size, dd.obs_context, dd.diffusion_kde = 1, deque(maxlen=1), False
    dd.use_ema = True
    dd.ema_helper = ExponentialMovingAverage(dd.model.get_params(), 0.999, "cpu")
    with torch.no_grad():      # the EMA shadow differs from the raw weights, as after training
        g = torch.Generator().manual_seed(4)
        for s in dd.ema_helper.shadow_params:
            s.add_(0.02 * torch.randn(s.shape, generator=g))
    obs = np.random.default_rng(5).normal(size=(N_ENV, T_STEPS, 16)) * 0.3
    ref = np.zeros((N_ENV, T_STEPS, 2))
    bank = NoiseBank(6, T_STEPS * 5, 2)
    with bank:
        for e in range(N_ENV):
            dd.reset(); bank.env, bank.call = e, 0
            for t in range(T_STEPS):
                ref[e, t] = np.asarray(dd.predict(obs[e, t])).reshape(-1)
    out.update(sd_arrays("ddpm_sd__", dd.model.state_dict()))
    out.update({"ddpm_ema__%03d" % i: s.numpy() for i, s in enumerate(dd.ema_helper.shadow_params)})
    out.update(ddpm_obs=obs, ddpm_ref=ref, ddpm_noise=bank.bank.numpy(), ddpm_x_mean=dd.scaler.x_mean.numpy(), ddpm_x_std=dd.scaler.x_std.numpy(),
               ddpm_y_mean=dd.scaler.y_mean.numpy(), ddpm_y_std=dd.scaler.y_std.numpy(), ddpm_y_bounds=dd.scaler.y_bounds)
    # ---------------------------------------------------------------- BESO (Stacking: obs 20 -> act 8, window 5, 16 sampling steps, sigma 0.01 .. 1; scripts/stacking/beso_benchmark.sh)
    torch.manual_seed(7)
    be = object.__new__(beso_mod.BesoAgent)
    be.device = "cpu"
    be.model = GCDenoiser(sigma_data=0.5, inner_model=dict(
        _target_="agents.models.beso.agents.diffusion_agents.k_diffusion.score_gpts.DiffusionGPT", state_dim=20, action_dim=8, goal_conditioned=False, embed_dim=32,
        embed_pdrob=0, attn_pdrop=0.2, resid_pdrop=0.1, n_layers=2, n_heads=4, goal_seq_len=1, obs_seq_len=5, sigma_vocab_size=8, device="cpu", linear_output=True,
        time_embedding_fn=None))
    with torch.no_grad():      # the positional embedding is zero-initialised: give it values
        be.model.inner_model.pos_emb.normal_(0, 0.1)
    be.scaler = make_scaler(20, 8, 7)
    be.model.min_action = torch.from_numpy(be.scaler.y_bounds[0, :]).to("cpu")
    be.model.max_action = torch.from_numpy(be.scaler.y_bounds[1, :]).to("cpu")
    be.window_size, be.obs_context, be.action_context = 5, deque(maxlen=5), deque(maxlen=4)
    be.sampler_type, be.num_sampling_steps, be.sigma_min, be.sigma_max, be.sigma_data, be.rho = "euler_ancestral", 16, 0.01, 1.0, 0.5, 5
    be.noise_scheduler, be.use_ema = "linear", False
    obs = np.random.default_rng(8).normal(size=(N_ENV, T_STEPS, 20)) * 0.3
    ref = np.zeros((N_ENV, T_STEPS, 8))
    bank = NoiseBank(9, T_STEPS * 17, 8)
    with bank:
        for e in range(N_ENV):
            be.reset(); bank.env, bank.call = e, 0
            for t in range(T_STEPS):
                ref[e, t] = np.asarray(be.predict(obs[e, t])).reshape(-1)
    out.update(sd_arrays("beso_sd__", be.model.state_dict()))
    out.update(beso_obs=obs, beso_ref=ref, beso_noise=bank.bank.numpy(), beso_x_mean=be.scaler.x_mean.numpy(), beso_x_std=be.scaler.x_std.numpy(),
               beso_y_mean=be.scaler.y_mean.numpy(), beso_y_std=be.scaler.y_std.numpy(), beso_y_bounds=be.scaler.y_bounds)
    dst = os.path.join(HERE, "ref_agents.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, "%.0f KB" % (os.path.getsize(dst) / 1024))


if __name__ == "__main__":
    main()
