"""Native batched policies (d3il_amd/policies.py) against the reference's own agents: tests/golden/ref_agents.npz holds fixed-seed
weights, scaler statistics, observations, the sampler noise and the outputs of BC_Agent / DiffusionAgent / BesoAgent.predict rolled
out batch-1 per environment (tests/golden/gen_agent_goldens.py, run where the reference is).  Row i of predict_batch must equal the
reference's predict for environment i at every step - including the history windows (BESO: 5 observations, 4 previous actions) and
the single EMA swap.  Runs on the CPU; the gpu-marked variant runs the same replay on cuda:0."""
import os

import numpy as np
import pytest
import torch

from d3il_amd import policies as P

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_agents.npz"))


def _sd(prefix):
    return {k[len(prefix):].replace("__", "."): torch.as_tensor(G[k]) for k in G.files if k.startswith(prefix)}


def _scaler(tag, dev):
    return P.Scaler(G[tag + "_x_mean"], G[tag + "_x_std"], G[tag + "_y_mean"], G[tag + "_y_std"], G[tag + "_y_bounds"], device=dev)


class _Bank:
    def __init__(self, bank, dev):
        self.bank, self.call, self.dev = torch.as_tensor(bank), 0, dev

    def __call__(self, shape):
        b = self.bank[self.call]
        self.call += 1
        out = b[:, 0, :shape[1]] if len(shape) == 2 else b[:, :shape[1], :shape[2]]
        return out.to(self.dev).contiguous()


def _bc(dev):
    model = P.ResidualMLP(10, 32, 4, 2).to(dev)
    model.load_state_dict(_sd("bc_sd__"))
    sc = _scaler("bc", dev)
    pol = P.BCPolicy(model, sc, sc.y_bounds[0], sc.y_bounds[1])
    obs, ref = G["bc_obs"], G["bc_ref"]
    worst = 0.0
    for t in range(obs.shape[1]):
        a = pol.predict_batch(torch.as_tensor(obs[:, t], device=dev))
        worst = max(worst, float(np.abs(a.cpu().numpy() - ref[:, t]).max()))
    return worst


def _ddpm(dev):
    model = P.DiffusionMLP(2, 16, 8, 32, 4).to(dev)
    bank = _Bank(G["ddpm_noise"], dev)
    pol = P.DDPMPolicy(model, _scaler("ddpm", dev), n_timesteps=4, noise_fn=bank)
    pol.load_reference_state_dict(_sd("ddpm_sd__"))
    pol.use_ema([G[k] for k in sorted(k for k in G.files if k.startswith("ddpm_ema__"))])
    obs, ref = G["ddpm_obs"], G["ddpm_ref"]
    worst = 0.0
    for t in range(obs.shape[1]):
        a = pol.predict_batch(torch.as_tensor(obs[:, t], device=dev))
        worst = max(worst, float(np.abs(a.cpu().numpy() - ref[:, t]).max()))
    return worst


def _beso(dev, staggered=False):
    inner = P.DiffusionGPT(20, 8, 32, 2, 4, 5, linear_output=True).to(dev)
    bank = _Bank(G["beso_noise"], dev)
    pol = P.BESOPolicy(inner, _scaler("beso", dev), window_size=5, num_sampling_steps=16, sigma_min=0.01, sigma_max=1.0, sigma_data=0.5, noise_fn=bank)
    pol.load_reference_state_dict(_sd("beso_sd__"))
    obs, ref = G["beso_obs"], G["beso_ref"]
    worst = 0.0
    for t in range(obs.shape[1]):
        a = pol.predict_batch(torch.as_tensor(obs[:, t], device=dev))
        worst = max(worst, float(np.abs(a.cpu().numpy() - ref[:, t]).max()))
    return worst


def test_bc_policy_rows_equal_reference_predict():
    assert _bc("cpu") < 2e-6


def test_ddpm_policy_rows_equal_reference_predict():
    assert _ddpm("cpu") < 2e-6


def test_beso_policy_rows_equal_reference_predict():
    assert _beso("cpu") < 5e-6


def test_history_restart_of_single_lanes():
    """begin_episodes(mask): a lane that restarts its trajectory gets a fresh window while the others keep theirs - the restarted lane
    must reproduce the reference's first steps again (the reference agent.reset() clears its deques)."""
    dev = "cpu"
    inner = P.DiffusionGPT(20, 8, 32, 2, 4, 5, linear_output=True)
    pol = P.BESOPolicy(inner, _scaler("beso", dev), window_size=5, num_sampling_steps=16, sigma_min=0.01, sigma_max=1.0, sigma_data=0.5,
                       noise_fn=lambda shape: torch.zeros(shape))
    pol.load_reference_state_dict(_sd("beso_sd__"))
    obs = torch.as_tensor(G["beso_obs"])
    first = [pol.predict_batch(obs[:, t]).clone() for t in range(3)]
    mask = torch.zeros(obs.shape[0], dtype=torch.uint8); mask[2] = 1
    pol.begin_episodes(mask)
    again = [pol.predict_batch(torch.where(mask.bool().unsqueeze(1), obs[:, t], obs[:, 3 + t])) for t in range(3)]
    for t in range(3):      # lane 2 sees its first observations again with an empty history: same outputs as at the start
        np.testing.assert_allclose(again[t][2].numpy(), first[t][2].numpy(), atol=1e-6)
    assert pol.obs_hist.len.tolist() == [5, 5, 3, 5, 5]


def test_beso_padded_batch_equals_per_lane_histories():
    """Lanes that restart at different times run in ONE right-padded batch (no grouping by history length, no host synchronisation):
    every lane must get what a single-environment policy with the same history and the same noise computes."""
    dev, n, T = "cpu", 6, 10
    gen = torch.Generator().manual_seed(7)
    inner = P.DiffusionGPT(20, 8, 32, 2, 4, 5, linear_output=True)
    sd = _sd("beso_sd__")
    rec = []

    def noise_rec(shape):
        rec.append(torch.randn(shape, generator=gen)); return rec[-1]

    pol = P.BESOPolicy(inner, _scaler("beso", dev), window_size=5, num_sampling_steps=16, sigma_min=0.01, sigma_max=1.0, sigma_data=0.5, noise_fn=noise_rec)
    pol.load_reference_state_dict(sd)
    obs = torch.randn(n, T, 20, generator=gen)
    restarts = {2: [1, 4], 5: [1], 6: [0, 3]}
    out = []
    for t in range(T):
        if t in restarts:
            m = torch.zeros(n, dtype=torch.uint8); m[restarts[t]] = 1
            pol.begin_episodes(m)
        out.append(pol.predict_batch(obs[:, t]).clone())
    calls_per_predict = len(rec) // T
    assert calls_per_predict * T == len(rec) and pol.obs_hist.lockstep < 0
    for i in range(n):
        k = [0]

        def noise_replay(shape, i=i, k=k):
            z = rec[k[0]][i:i + 1, :shape[1]]; k[0] += 1
            assert tuple(z.shape) == tuple(shape)
            return z

        one = P.BESOPolicy(P.DiffusionGPT(20, 8, 32, 2, 4, 5, linear_output=True), _scaler("beso", dev), window_size=5, num_sampling_steps=16, sigma_min=0.01,
                           sigma_max=1.0, sigma_data=0.5, noise_fn=noise_replay)
        one.load_reference_state_dict(sd)
        for t in range(T):
            if t in restarts and i in restarts[t]:
                one.reset()
            a = one.predict_batch(obs[i:i + 1, t])
            np.testing.assert_allclose(a[0].numpy(), out[t][i].numpy(), atol=2e-6, err_msg="lane %d step %d" % (i, t))


@pytest.mark.gpu
def test_fused_attention_kernel_equals_torch_math():
    """d3il_attention_causal_f32 (one lane per sequence / head / query, online softmax) against the masked-softmax formulation of
    score_gpts.py:59-76 on random inputs, incl. a sequence length and head size that are not those of BASELINE config 5."""
    from d3il_amd import capi
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    for B, T, H, D in ((257, 11, 6, 20), (33, 7, 4, 32), (5, 1, 2, 8), (64, 32, 3, 16)):
        C_ = H * D
        qkv = torch.randn(B, T, 3 * C_, generator=g).to(dev)
        out = torch.empty(B, T, C_, device=dev)
        capi.check(capi.load().d3il_attention_causal_f32(qkv.data_ptr(), out.data_ptr(), B, T, H, D, torch.cuda.current_stream(dev).cuda_stream))
        q, k, v = (qkv[..., i * C_:(i + 1) * C_].view(B, T, H, D).transpose(1, 2).double() for i in range(3))
        att = (q @ k.transpose(-2, -1)) / D ** 0.5
        att = att.masked_fill(torch.tril(torch.ones(T, T, device=dev)) == 0, float("-inf"))
        ref = (torch.softmax(att, dim=-1) @ v).transpose(1, 2).reshape(B, T, C_)
        assert float((out.double() - ref).abs().max()) < 2e-5
    assert capi.load().d3il_attention_causal_f32(qkv.data_ptr(), out.data_ptr(), 4, 33, 2, 8, 0) != 0      # T > 32 is refused


@pytest.mark.gpu
def test_fused_layernorm_kernel_equals_torch():
    from d3il_amd import capi
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    for rows, C_ in ((45056, 120), (7, 128), (1, 4), (333, 64)):
        x = (torch.randn(rows, C_, generator=g) * 3 + 0.5).to(dev)
        ln = torch.nn.LayerNorm(C_).to(dev)
        with torch.no_grad():
            ln.weight.copy_(torch.randn(C_, generator=g)); ln.bias.copy_(torch.randn(C_, generator=g))
        y = torch.empty_like(x)
        capi.check(capi.load().d3il_layernorm_f32(x.data_ptr(), ln.weight.data_ptr(), ln.bias.data_ptr(), y.data_ptr(), rows, C_, float(ln.eps), torch.cuda.current_stream(dev).cuda_stream))
        with torch.no_grad():
            ref = torch.nn.functional.layer_norm(x.double(), (C_,), ln.weight.double(), ln.bias.double(), ln.eps)
        assert float((y.double() - ref).abs().max()) < 2e-5
    assert capi.load().d3il_layernorm_f32(x.data_ptr(), ln.weight.data_ptr(), ln.bias.data_ptr(), y.data_ptr(), 4, 130, 1e-5, 0) != 0      # C > 128 is refused


@pytest.mark.gpu
def test_beso_graph_replay_equals_eager():
    """use_graph: the captured sampling loop gives the eager result for the same generator state."""
    import bench
    dev = torch.device("cuda:0")
    obs = torch.randn(64, 9, 20, generator=torch.Generator().manual_seed(1)).to(dev)
    outs = []
    for graph in (False, True):
        torch.manual_seed(5); torch.cuda.manual_seed(5)
        pol = bench._random_beso(dev)
        pol.use_graph = graph
        torch.cuda.manual_seed(11)
        acts = [pol.predict_batch(obs[:, t]).clone() for t in range(9)]
        outs.append(torch.stack(acts))
    # the first W - 1 steps run eagerly in both with the same generator state; afterwards the draws differ (the warm-up runs before the
    # capture advance the generator): the replayed actions must be finite and inside the policy's action bounds
    np.testing.assert_allclose(outs[0][:4].cpu().numpy(), outs[1][:4].cpu().numpy(), atol=1e-6)
    lo = torch.tensor([-0.01] * 7 + [0.0], device=dev) - 1e-6
    hi = torch.tensor([0.01] * 7 + [0.08], device=dev) + 1e-6
    assert torch.isfinite(outs[1]).all() and bool(((outs[1] >= lo) & (outs[1] <= hi)).all())
    assert float((outs[1][4:] - outs[1][3:-1]).abs().max()) > 0      # and they are not a stale buffer


@pytest.mark.gpu
def test_policies_on_the_gpu():
    assert torch.cuda.is_available()
    assert _bc("cuda:0") < 1e-5 and _ddpm("cuda:0") < 1e-5 and _beso("cuda:0") < 2e-5
