"""BASELINE configs 4 / 5 through the Sim mirrors with the native batched policies (d3il_amd/policies.py; fixed random weights - there
are no checkpoints offline): the plumbing a D3IL user gets when the Hydra `simulation._target_` points at this package and the agent
offers `predict_batch` (INTEGRATION.md sections 1 and 7).  Physics parity lives in the test_gpu_parity_* files."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_sorting_sim_with_ddpm_policy():
    import bench
    from d3il_amd.simulation.sorting_sim import Sorting_Sim
    sim = Sorting_Sim(seed=0, device="cuda:0", render=False, n_cores=1, n_contexts=6, n_trajectories_per_context=4, max_steps_per_episode=25)
    pol = bench._random_ddpm(16, torch.device("cuda:0"))           # obs 16 = desired xy + the 14-d observation (configs/sorting_4_config.yaml:47)
    res = sim.test_agent(pol)
    assert set(res) == {"score", "Metrics/successes", "Metrics/KL", "Metrics/entropy"}     # KL / entropy of an empty success table follow the reference
    r = sim.last_rollout
    assert r["mode"].shape[0] == 24 and not bool((r["flags"] & ((1 << 16) | (1 << 18))).any())     # no solver failure, no contact overflow
    assert res["Metrics/successes"] == 0.0                           # nothing is sorted within 25 steps


def test_stacking_sim_with_beso_policy():
    import bench
    from d3il_amd.simulation.stacking_sim import Stacking_Sim
    sim = Stacking_Sim(seed=0, device="cuda:0", render=False, n_cores=1, n_contexts=5, n_trajectories_per_context=3, max_steps_per_episode=12)
    pol = bench._random_beso(torch.device("cuda:0"))
    pol.use_graph = False
    res = sim.test_agent(pol)
    successes, mode_encoding = res                                    # the reference's return value: two [n_contexts, n_trajectories] tables
    assert tuple(successes.shape) == (5, 3) == tuple(mode_encoding.shape) and float(successes.sum()) == 0.0
    assert float(sim.last_rollout["metrics"]["successes"]) == 0.0
    r = sim.last_rollout
    assert r["mode"].shape[0] == 15 and not bool((r["flags"] & ((1 << 16) | (1 << 18))).any())
    assert not bool(r["success"].any()) and pol.obs_hist.len.tolist() == [5] * 15      # window 5 filled, per-lane histories in lock step


def test_fused_mlp_kernel_matches_the_torch_block():
    """d3il_mlp_gelu_residual_f32 (fc1 + GELU + fc2 + residual of the DiffusionGPT block on the f32 matrix cores, hidden activations kept in registers)
    against torch's two Linear layers: same result to the round-off of an f32 GEMM (both are within 2e-6 of the f64 evaluation), for ragged row counts
    (the kernel works on 16-row tiles, 64 rows per workgroup)."""
    import torch
    from d3il_amd import capi, policies as P
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    fc1, fc2 = torch.nn.Linear(120, 480).to(dev), torch.nn.Linear(480, 120).to(dev)
    L = capi.load()
    for M in (1, 15, 16, 17, 63, 65, 1000, 4096 * 11):
        h, x = torch.randn(M, 120, device=dev), torch.randn(M, 120, device=dev)
        with torch.no_grad():
            ref64 = x.double() + torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(h.double(), fc1.weight.double(), fc1.bias.double())),
                                                            fc2.weight.double(), fc2.bias.double())
            ref32 = x + fc2(torch.nn.functional.gelu(fc1(h)))
            wp = P.pack_mlp_weights(fc1, fc2)
            out = torch.full_like(x, float("nan"))
            capi.check(L.d3il_mlp_gelu_residual_f32(h.data_ptr(), x.data_ptr(), wp.data_ptr(), fc1.bias.data_ptr(), fc2.bias.data_ptr(), out.data_ptr(), M, 120, 480,
                                                    torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
        e_fused, e_torch = float((out.double() - ref64).abs().max()), float((ref32.double() - ref64).abs().max())
        assert e_fused < 3e-6 and e_fused < 2.5 * e_torch + 1e-7, (M, e_fused, e_torch)
    with pytest.raises(capi.D3ilError):
        capi.check(L.d3il_mlp_gelu_residual_f32(h.data_ptr(), x.data_ptr(), wp.data_ptr(), fc1.bias.data_ptr(), fc2.bias.data_ptr(), out.data_ptr(), 4, 128, 512, None))


def test_beso_policy_with_and_without_the_fused_mlp():
    """The batched BESO policy (DiffusionGPT 120 / 6 layers / 6 heads, 16 sampling steps) gives the same actions with the fused MLP kernel and with torch's
    layers (D3IL_POLICY_FUSED_MLP=0): the difference stays at f32 round-off through the 16 x 6 blocks."""
    import os
    import torch
    import bench
    dev = torch.device("cuda:0")
    outs = {}
    for fused in ("1", "0"):
        os.environ["D3IL_POLICY_FUSED_MLP"] = fused
        os.environ["D3IL_POLICY_GRAPH"] = "0"
        try:
            pol = bench._random_beso(dev)
            torch.manual_seed(3)
            obs = torch.randn(512, 20, device=dev)
            acts = []
            for t in range(3):
                torch.manual_seed(100 + t)
                acts.append(pol.predict_batch(obs + 0.01 * t).clone())
            outs[fused] = torch.stack(acts)
        finally:
            os.environ.pop("D3IL_POLICY_FUSED_MLP", None)
            os.environ.pop("D3IL_POLICY_GRAPH", None)
    d = float((outs["1"] - outs["0"]).abs().max())
    assert torch.isfinite(outs["1"]).all() and d < 2e-5, d


def test_fused_linear120_and_layernorm_kernels_match_torch():
    """d3il_linear120_f32 (LayerNorm + linear layer with 120 inputs + bias + residual on the f32 matrix cores) and the LayerNorm-fused MLP kernel against
    torch, for the two shapes of the DiffusionGPT block (N = 360 after ln1, N = 120 with residual) and ragged row counts."""
    import torch
    from d3il_amd import capi, policies as P
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    L = capi.load()
    ln = torch.nn.LayerNorm(120).to(dev)
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5); ln.bias.uniform_(-0.2, 0.2)
    st = torch.cuda.current_stream().cuda_stream
    for N, use_ln, use_res in ((360, True, False), (120, False, True), (120, True, True), (24, False, False)):
        lin = torch.nn.Linear(120, N).to(dev)
        for M in (1, 17, 64, 1000, 4096 * 11):
            x = torch.randn(M, 120, device=dev) * 2 + 0.3
            res = torch.randn(M, N, device=dev)
            with torch.no_grad():
                h64 = torch.nn.functional.layer_norm(x.double(), (120,), ln.weight.double(), ln.bias.double(), ln.eps) if use_ln else x.double()
                ref64 = torch.nn.functional.linear(h64, lin.weight.double(), lin.bias.double()) + (res.double() if use_res else 0)
                ref32 = lin(ln(x) if use_ln else x) + (res if use_res else 0)
                out = torch.full((M, N), float("nan"), device=dev)
                wp = P.pack_linear120_weights(lin.weight)
                capi.check(L.d3il_linear120_f32(x.data_ptr(), ln.weight.data_ptr() if use_ln else None, ln.bias.data_ptr() if use_ln else None, float(ln.eps), wp.data_ptr(),
                                                lin.bias.data_ptr(), res.data_ptr() if use_res else None, out.data_ptr(), M, N, st))
                torch.cuda.synchronize()
            e_f, e_t = float((out.double() - ref64).abs().max()), float((ref32.double() - ref64).abs().max())
            assert e_f < 5e-6 and e_f < 3 * e_t + 2e-7, (N, use_ln, use_res, M, e_f, e_t)
    fc1, fc2 = torch.nn.Linear(120, 480).to(dev), torch.nn.Linear(480, 120).to(dev)
    for M in (5, 4096):
        x = torch.randn(M, 120, device=dev)
        with torch.no_grad():
            h64 = torch.nn.functional.layer_norm(x.double(), (120,), ln.weight.double(), ln.bias.double(), ln.eps)
            ref64 = x.double() + torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(h64, fc1.weight.double(), fc1.bias.double())), fc2.weight.double(), fc2.bias.double())
            out = torch.empty_like(x)
            wp = P.pack_mlp_weights(fc1, fc2)
            capi.check(L.d3il_mlp_ln_gelu_residual_f32(x.data_ptr(), ln.weight.data_ptr(), ln.bias.data_ptr(), float(ln.eps), x.data_ptr(), wp.data_ptr(), fc1.bias.data_ptr(),
                                                       fc2.bias.data_ptr(), out.data_ptr(), M, 120, 480, st))
            torch.cuda.synchronize()
        assert float((out.double() - ref64).abs().max()) < 4e-6

