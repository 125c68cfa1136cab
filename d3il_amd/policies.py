"""Native batched policies for the rollout boundary: the reference's evaluation-time policies restated for a
BATCH of environments on device-resident observations.

The reference agents are batch-1 and numpy-in / numpy-out (``agent.predict(np.ndarray[obs]) -> np.ndarray[1, act]``,
agents/base_agent.py:110-122) with a host <-> device round trip, an EMA parameter swap (store / copy_to / restore over every
parameter) and a Python deque of the observation history inside EVERY call (ddpm_agent.py:213-274, beso_agent.py:316-443).  With the
simulator at millions of env-steps/s the policy is the bottleneck, so the classes below run the same computation once per step on
the whole batch: ``predict_batch(obs[N, obs_dim]) -> act[N, act_dim]`` with
  * the scaler (agents/utils/scaler.py:72-113) folded into two affine maps on the device,
  * the observation / action history in device ring buffers ``[N, W, dim]`` with a per-lane length (lanes that start a new
    trajectory - ``begin_episodes(mask)`` - restart their history),
  * the EMA weights swapped in ONCE (``use_ema(shadow_params)``) instead of per call,
  * the denoising loops (DDPM ancestral sampling, gc_diffusion.py:144-200; BESO Euler-ancestral on the Karras-preconditioned
    denoiser, gc_sampling.py:217-256, score_wrappers.py:20-99) running on the batch.
Networks are re-stated with the reference's parameter names, so a reference checkpoint (``model_state_dict``) loads unchanged
(``load_reference_state_dict``); tests/golden/gen_agent_goldens.py builds the reference agents with fixed-seed weights in the build
container and stores weights, inputs, noise and the reference's outputs, and tests/test_policies.py replays them here (row i of the
batch == the reference's batch-1 ``predict`` of environment i).
"""
from __future__ import annotations

import math
import os

import torch
from torch import nn
from torch.nn import functional as F


# ------------------------------------------------------------------------------------------------ scaler
class Scaler:
    """agents/utils/scaler.py:72-113 for ``scale_data=True``: x -> (x - mean) / (std + 1e-12), y -> y (std + 1e-12) + mean."""

    def __init__(self, x_mean, x_std, y_mean, y_std, y_bounds=None, device="cuda"):
        f = lambda a: torch.as_tensor(a, dtype=torch.float32, device=device)
        self.x_mean, self.x_std, self.y_mean, self.y_std = f(x_mean), f(x_std), f(y_mean), f(y_std)
        self.y_bounds = None if y_bounds is None else f(y_bounds)

    def scale_input(self, x):
        return ((x - self.x_mean) / (self.x_std + 1e-12)).to(torch.float32)

    def inverse_scale_output(self, y):
        return y * (self.y_std + 1e-12) + self.y_mean


# ------------------------------------------------------------------------------------------------ networks
class _ResBlock(nn.Module):          # TwoLayerPreActivationResNetLinear, agents/models/common/mlp.py:9-46 (no norm, no dropout at eval)
    def __init__(self, hidden_dim):
        super().__init__()
        self.l1, self.l2 = nn.Linear(hidden_dim, hidden_dim), nn.Linear(hidden_dim, hidden_dim)

    def forward(self, x):
        return x + self.l2(F.mish(self.l1(F.mish(x))))


class ResidualMLP(nn.Module):
    """ResidualMLPNetwork (agents/models/common/mlp.py:114-182), Mish: Linear, num_hidden_layers / 2 pre-activation residual blocks, Linear."""

    def __init__(self, input_dim, hidden_dim, num_hidden_layers, output_dim):
        super().__init__()
        assert num_hidden_layers % 2 == 0
        self.layers = nn.ModuleList([nn.Linear(input_dim, hidden_dim)] + [_ResBlock(hidden_dim) for _ in range(1, num_hidden_layers, 2)] + [nn.Linear(hidden_dim, output_dim)])

    def _parts(self):
        return (self.layers[0], [(b.l1, b.l2) for b in self.layers[1:-1]], self.layers[-1])

    def ensure_packed(self):
        """Refresh the packed weight buffers of the device path (in place) if a parameter has changed - what a captured graph's owner calls before a replay."""
        if getattr(self, "_fused", None) is not None and self._fused._fw is not None:
            self._fused.ensure_packed(self._parts())

    def invalidate_packed(self):
        """After ``param.data`` writes (invisible to the version counters): the next call / ensure_packed() repacks."""
        if getattr(self, "_fused", None) is not None:
            self._fused.invalidate()

    def forward(self, x):
        x = x.to(torch.float32)
        if x.dim() == 2 and x.is_cuda:
            if getattr(self, "_fused", None) is None:
                object.__setattr__(self, "_fused", FusedResMLP())
            parts = self._parts()
            if self._fused.ok(x, parts):
                return self._fused(x, parts)      # one launch on the f32 matrix cores (D3IL_POLICY_FUSED_RESMLP=0: torch's layers)
        for layer in self.layers:
            x = layer(x)
        return x


class _SinusoidalPosEmb(nn.Module):   # agents/models/diffusion/utils.py:9-22
    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, x):
        half = self.dim // 2
        emb = torch.exp(torch.arange(half, device=x.device) * -(math.log(10000) / (half - 1)))
        emb = x[:, None] * emb[None, :]
        return torch.cat((emb.sin(), emb.cos()), dim=-1)


class DiffusionMLP(nn.Module):
    """DiffusionMLPNetwork (agents/models/diffusion/diffusion_models.py:20-118), residual style, not goal conditioned:
    eps(x, t, state) = ResidualMLP(cat(x, time_mlp(t), state))."""

    def __init__(self, action_dim, obs_dim, t_dim, hidden_dim, num_hidden_layers):
        super().__init__()
        self.temp_layers = nn.Sequential(_SinusoidalPosEmb(t_dim), nn.Linear(t_dim, t_dim * 2), nn.Mish(), nn.Linear(t_dim * 2, t_dim))
        self.layers = ResidualMLP(obs_dim + action_dim + t_dim, hidden_dim, num_hidden_layers, action_dim)

    def forward(self, x, t, state):
        t = self.temp_layers(t)
        if state.dim() == 3:
            return self.layers(torch.cat([x, t[:, None, :].expand(-1, state.shape[1], -1), state], dim=2))
        return self.layers(torch.cat([x, t, state], dim=1))


class _CausalSelfAttention(nn.Module):     # score_gpts.py:15-80
    def __init__(self, n_embd, n_heads, block_size):
        super().__init__()
        self.key, self.query, self.value, self.proj = (nn.Linear(n_embd, n_embd) for _ in range(4))
        self.register_buffer("mask", torch.tril(torch.ones(block_size, block_size)).view(1, 1, block_size, block_size))
        self.n_head = n_heads

    def forward(self, x):
        B, T, C = x.size()
        hd = C // self.n_head
        if x.is_cuda and x.dtype == torch.float32 and T <= 32 and hd <= 32:
            # device path: ONE linear layer for query | key | value, then the fused short-sequence attention kernel of the rollout
            # library (d3il_attention_causal_f32) which writes token-major output - no [B, H, T, T] tensors, no batched 11 x 20 GEMMs
            from . import capi
            w = torch.cat((self.query.weight, self.key.weight, self.value.weight), dim=0)
            bqkv = torch.cat((self.query.bias, self.key.bias, self.value.bias), dim=0)
            qkv = F.linear(x, w, bqkv).contiguous()
            y = torch.empty(B, T, C, dtype=torch.float32, device=x.device)
            capi.check(capi.load().d3il_attention_causal_f32(qkv.data_ptr(), y.data_ptr(), B, T, self.n_head, hd, torch.cuda.current_stream(x.device).cuda_stream))
            return self.proj(y)
        k = self.key(x).view(B, T, self.n_head, hd).transpose(1, 2)
        q = self.query(x).view(B, T, self.n_head, hd).transpose(1, 2)
        v = self.value(x).view(B, T, self.n_head, hd).transpose(1, 2)
        att = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(k.size(-1)))
        att = att.masked_fill(self.mask[:, :, :T, :T] == 0, float("-inf"))
        y = F.softmax(att, dim=-1) @ v
        return self.proj(y.transpose(1, 2).contiguous().view(B, T, C))


def _layer_norm(ln: nn.LayerNorm, x):
    """nn.LayerNorm; on the device the narrow-row kernel of the rollout library (the activations are [B * T][120]: torch's kernel
    reaches an eighth of the memory bandwidth on rows this short)."""
    C = x.shape[-1]
    if x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and 4 <= C <= 128 and C % 4 == 0 and x.data_ptr() % 16 == 0:
        from . import capi
        y = torch.empty_like(x)
        capi.check(capi.load().d3il_layernorm_f32(x.data_ptr(), ln.weight.data_ptr(), ln.bias.data_ptr(), y.data_ptr(), x.numel() // C, C, float(ln.eps),
                                                 torch.cuda.current_stream(x.device).cuda_stream))
        return y
    return ln(x)


_MLP_PACK_IDX = {}


def mlp_pack_index(C: int, H: int, device) -> torch.Tensor:
    """Gather index that puts the two weight matrices of a transformer MLP (fc1.weight [H, C], fc2.weight [C, H], flattened and concatenated, plus one
    trailing zero) into the per-chunk order d3il_mlp_gelu_residual_f32 copies to LDS: chunk c (16 hidden units) = 16 blocks of [4 g][16 i][4 e] floats (lane 16 g + i reads its float4 at [block][lane]);
    blocks q < 8: fc1.weight[16 c + i][4 (4 q + e) + g] (the A operand of step s = 4 q + e of the first product, zero for s >= C / 4); blocks 8 + t:
    fc2.weight[16 t + i][16 c + 4 g + e] (the A operand of step e of output tile t of the second product, zero for rows >= C)."""
    key = (C, H, str(device))
    if key not in _MLP_PACK_IDX:
        c = torch.arange(H // 16).view(-1, 1, 1, 1, 1)
        blk = torch.arange(8).view(1, -1, 1, 1, 1)
        g = torch.arange(4).view(1, 1, -1, 1, 1)          # lane = 16 g + i: the float4 of a lane sits at [block][lane]
        i = torch.arange(16).view(1, 1, 1, -1, 1)
        e = torch.arange(4).view(1, 1, 1, 1, -1)
        zero = C * H * 2
        s = 4 * blk + e
        i1 = torch.where(s < C // 4, (16 * c + i) * C + 4 * s + g, torch.full_like(s + c + i + g, zero))
        row = 16 * blk + i
        i2 = torch.where(row < C, C * H + row * H + 16 * c + 4 * g + e, torch.full_like(row + c + g + e, zero))
        _MLP_PACK_IDX[key] = torch.cat((i1.expand(H // 16, 8, 4, 16, 4), i2.expand(H // 16, 8, 4, 16, 4)), dim=1).reshape(-1).to(device)
    return _MLP_PACK_IDX[key]


def linear120_pack_index(N: int, device) -> torch.Tensor:
    """Gather index for d3il_linear120_f32: weight [N, 120] (flattened, plus one trailing zero) -> ceil(N / 16) tiles of 8 blocks of [4 g][16 i][4 e] floats,
    block q of tile t: W[16 t + i][4 (4 q + e) + g] (zero beyond N rows / 30 steps)."""
    key = ("lin", N, str(device))
    if key not in _MLP_PACK_IDX:
        nt = (N + 15) // 16
        t = torch.arange(nt).view(-1, 1, 1, 1, 1)
        q = torch.arange(8).view(1, -1, 1, 1, 1)
        g = torch.arange(4).view(1, 1, -1, 1, 1)
        i = torch.arange(16).view(1, 1, 1, -1, 1)
        e = torch.arange(4).view(1, 1, 1, 1, -1)
        s_, row = 4 * q + e, 16 * t + i
        idx = torch.where((s_ < 30) & (row < N), row * 120 + 4 * s_ + g, torch.full_like(row + s_ + g, N * 120))
        _MLP_PACK_IDX[key] = idx.reshape(-1).to(device)
    return _MLP_PACK_IDX[key]


def pack_linear120_weights(weight: torch.Tensor) -> torch.Tensor:
    flat = torch.cat((weight.reshape(-1), weight.new_zeros(1)))
    return flat[linear120_pack_index(weight.shape[0], weight.device)]


def pack_mlp_weights(fc1: nn.Linear, fc2: nn.Linear) -> torch.Tensor:
    H, C = fc1.weight.shape
    flat = torch.cat((fc1.weight.reshape(-1), fc2.weight.reshape(-1), fc1.weight.new_zeros(1)))
    return flat[mlp_pack_index(C, H, fc1.weight.device)]


def split_f16(w: torch.Tensor):
    """(hi, lo) f16 halves of an f32 tensor as csrc/policy_f16x3.h uses them: hi = f16(w) (saturating), lo = f16((w - hi) * 2^11)."""
    w = w.detach().to(torch.float32).clamp(-65504.0, 65504.0)
    hi = w.to(torch.float16)
    lo = ((w - hi.to(torch.float32)) * 2048.0).to(torch.float16)
    return hi, lo


def mlp_f16x3_pack_index(C: int, H: int, device) -> torch.Tensor:
    """Gather index [H / 32 pairs][2048 vectors][8] into cat(fc1.weight [H, C], fc2.weight [C, H], one zero) for d3il_mlp_ln_gelu_residual_f16x3 - WITHOUT the half
    dimension (the packer interleaves hi / lo): entry (c, v, e) with v < 512: tile = v // 256, s = (v // 64) % 4, lane = v % 64 -> fc1.weight[32 c + 16 tile + i][32 s + 8 g + e];
    v >= 512: t = (v - 512) // 64 -> fc2.weight[16 t + i][32 c + 16 (e >> 2) + 4 g + (e & 3)]; lane = 16 g + i; zero beyond the matrices."""
    key = ("mlp16", C, H, str(device))
    if key not in _MLP_PACK_IDX:
        c = torch.arange(H // 32).view(-1, 1, 1, 1, 1, 1)
        tile = torch.arange(2).view(1, -1, 1, 1, 1, 1)
        s_ = torch.arange(4).view(1, 1, -1, 1, 1, 1)
        g = torch.arange(4).view(1, 1, 1, -1, 1, 1)
        i = torch.arange(16).view(1, 1, 1, 1, -1, 1)
        e = torch.arange(8).view(1, 1, 1, 1, 1, -1)
        zero = 2 * C * H
        k = 32 * s_ + 8 * g + e
        i1 = torch.where(k < C, (32 * c + 16 * tile + i) * C + k, torch.full_like(k + c + tile + i, zero))            # [P, 2, 4, 4, 16, 8]
        t = torch.arange(8).view(1, -1, 1, 1, 1)
        c2, g2, i2_, e2 = c.view(-1, 1, 1, 1, 1), g.view(1, 1, -1, 1, 1), i.view(1, 1, 1, -1, 1), e.view(1, 1, 1, 1, -1)
        row = 16 * t + i2_
        hid = 32 * c2 + 16 * (e2 >> 2) + 4 * g2 + (e2 & 3)
        i2 = torch.where(row < C, C * H + row * H + hid, torch.full_like(row + hid, zero))                              # [P, 8, 4, 16, 8]
        P = H // 32
        _MLP_PACK_IDX[key] = (i1.reshape(P, 512, 8).to(device), i2.reshape(P, 512, 8).to(device))
    return _MLP_PACK_IDX[key]


def pack_mlp_weights_f16x3(fc1_weight: torch.Tensor, fc2_weight: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """f16 [H / 32 + 1 stages][2048][8]: stage k = first-product vectors ((tile * 4 + s) * 2 + p) * 64 + lane of hidden pair k (zero for the last stage), then the
    second-product vectors 1024 + (t * 2 + p) * 64 + lane of pair k - 1 (zero for stage 0); p: 0 hi, 1 lo - the software pipeline of k_mlp_gelu_residual_f16x3."""
    H, C = fc1_weight.shape
    i1, i2 = mlp_f16x3_pack_index(C, H, fc1_weight.device)
    flat = torch.cat((fc1_weight.reshape(-1), fc2_weight.reshape(-1), fc1_weight.new_zeros(1)))
    hi, lo = split_f16(flat)
    P = H // 32
    a = torch.stack((hi[i1].view(P, 8, 64, 8), lo[i1].view(P, 8, 64, 8)), dim=2).reshape(P, 1024, 8)      # [(tile, s)][p][lane]
    b = torch.stack((hi[i2].view(P, 8, 64, 8), lo[i2].view(P, 8, 64, 8)), dim=2).reshape(P, 1024, 8)      # [t][p][lane]
    z = torch.zeros_like(a[:1])
    res = torch.cat((torch.cat((a, z), dim=0), torch.cat((z, b), dim=0)), dim=1).contiguous()      # [P + 1, 2048, 8]
    if out is not None:
        out.copy_(res)
        return out
    return res


def pack_linear120_weights_f16x3(weight: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """f16 [2 ceil(N / 32) tiles][512][8] for d3il_linear120_f16x3: vector (s * 2 + p) * 64 + lane of tile t = W_p[16 t + i][32 s + 8 g + e] (zero beyond N x 120)."""
    N, C = weight.shape
    key = ("lin16", N, C, str(weight.device))
    if key not in _MLP_PACK_IDX:
        nt = 2 * ((N + 31) // 32)
        t = torch.arange(nt).view(-1, 1, 1, 1, 1)
        s_ = torch.arange(4).view(1, -1, 1, 1, 1)
        g = torch.arange(4).view(1, 1, -1, 1, 1)
        i = torch.arange(16).view(1, 1, 1, -1, 1)
        e = torch.arange(8).view(1, 1, 1, 1, -1)
        row, k = 16 * t + i, 32 * s_ + 8 * g + e
        _MLP_PACK_IDX[key] = torch.where((row < N) & (k < C), row * C + k, torch.full_like(row + k, N * C)).reshape(nt, 4, 64, 8).to(weight.device)
    idx = _MLP_PACK_IDX[key]
    hi, lo = split_f16(torch.cat((weight.reshape(-1), weight.new_zeros(1))))
    res = torch.stack((hi[idx], lo[idx]), dim=2).reshape(idx.shape[0], 512, 8).contiguous()
    if out is not None:
        out.copy_(res)
        return out
    return res


def policy_gemm_mode() -> str:
    """Which matrix-core path the DiffusionGPT blocks take: "f16x3" (default: split-f16 products, csrc/policy_f16x3.h) or "f32" (D3IL_POLICY_GEMM=f32: the f32-input MFMA
    kernels of rounds 3 - 5)."""
    return os.environ.get("D3IL_POLICY_GEMM", "f16x3")


class _Block(nn.Module):                   # score_gpts.py:83-115
    def __init__(self, n_embd, n_heads, block_size):
        super().__init__()
        self.ln1, self.ln2 = nn.LayerNorm(n_embd), nn.LayerNorm(n_embd)
        self.attn = _CausalSelfAttention(n_embd, n_heads, block_size)
        self.mlp = nn.Sequential(nn.Linear(n_embd, 4 * n_embd), nn.GELU(), nn.Linear(4 * n_embd, n_embd), nn.Dropout(0.0))

    def _fused_static_ok(self):
        return (self.mlp[0].weight.shape == (480, 120) and self.mlp[0].weight.is_cuda and self.mlp[0].weight.dtype == torch.float32 and 120 // self.attn.n_head <= 32
                and os.environ.get("D3IL_POLICY_FUSED_MLP", "1") == "1")

    def _fused_ok(self, x):
        # the matrix-core kernels are inference only (no autograd node): any caller that wants gradients - fine-tuning, a gradient check - takes the torch path
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return False
        return self._fused_static_ok() and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.shape[-1] == 120 and x.data_ptr() % 16 == 0 and x.shape[1] <= 32

    def invalidate_packed(self):
        """Force a repack at the next ensure_packed() (after ``param.data`` writes, which the version counters do not see)."""
        self._pack_key = None

    def ensure_packed(self):
        """Packed copies of the block's weights in the tile order of the matrix-core kernels, in PERSISTENT device buffers refreshed in place whenever a
        parameter has changed (tensor version counters: no device synchronisation).  The addresses never change, so a captured HIP graph keeps reading the
        current weights as long as this runs before every replay (BESOPolicy.predict_batch does) - e.g. after the EMA swap of a rollout."""
        a, fc1, fc2 = self.attn, self.mlp[0], self.mlp[2]
        params = (a.query.weight, a.key.weight, a.value.weight, a.query.bias, a.key.bias, a.value.bias, a.proj.weight, fc1.weight, fc2.weight)
        key = tuple((p.data_ptr(), p._version) for p in params)
        if getattr(self, "_pack_key", None) == key:
            return
        dev = fc1.weight.device
        if getattr(self, "_wp_qkv", None) is None or self._wp_qkv.device != dev:
            self._wp_qkv = torch.empty(linear120_pack_index(360, dev).numel(), dtype=torch.float32, device=dev)
            self._wp_proj = torch.empty(linear120_pack_index(120, dev).numel(), dtype=torch.float32, device=dev)
            self._wp_mlp = torch.empty(mlp_pack_index(120, 480, dev).numel(), dtype=torch.float32, device=dev)
            self._b_qkv = torch.empty(360, dtype=torch.float32, device=dev)
        z = fc1.weight.new_zeros(1)
        with torch.no_grad():
            torch.index_select(torch.cat((a.query.weight.reshape(-1), a.key.weight.reshape(-1), a.value.weight.reshape(-1), z)), 0, linear120_pack_index(360, dev), out=self._wp_qkv)
            torch.index_select(torch.cat((a.proj.weight.reshape(-1), z)), 0, linear120_pack_index(120, dev), out=self._wp_proj)
            torch.index_select(torch.cat((fc1.weight.reshape(-1), fc2.weight.reshape(-1), z)), 0, mlp_pack_index(120, 480, dev), out=self._wp_mlp)
            torch.cat((a.query.bias, a.key.bias, a.value.bias), dim=0, out=self._b_qkv)
            # the split-f16 forms of the same three matrices (csrc/policy_f16x3.h), refreshed in place like the f32 ones
            wq = torch.cat((a.query.weight, a.key.weight, a.value.weight), dim=0)
            self._hp_qkv = pack_linear120_weights_f16x3(wq, getattr(self, "_hp_qkv", None))
            self._hp_proj = pack_linear120_weights_f16x3(a.proj.weight, getattr(self, "_hp_proj", None))
            self._hp_mlp = pack_mlp_weights_f16x3(fc1.weight, fc2.weight, getattr(self, "_hp_mlp", None))
            # (query | key | value) tiles followed by the projection's: the weight stream of the one-kernel attention half (d3il_attn_half_f16x3)
            if getattr(self, "_hp_attn", None) is None or self._hp_attn.device != dev:
                self._hp_attn = torch.empty(self._hp_qkv.shape[0] + self._hp_proj.shape[0], 512, 8, dtype=self._hp_qkv.dtype, device=dev)
            torch.cat((self._hp_qkv, self._hp_proj), dim=0, out=self._hp_attn)
        self._pack_key = key

    def forward(self, x, keep=None):
        """keep: token positions (LongTensor) whose outputs are needed; the block then returns [B, len(keep), C] - attention still sees every token, the output
        projection and the MLP run on the kept rows only (the last block of DiffusionGPT: only the action positions are decoded)."""
        if self._fused_ok(x):
            # device path, four kernels of the rollout library per block, all GEMMs on the matrix cores (split-f16 products by default, policy_gemm_mode()) with the
            # LayerNorms, biases, GELU and residuals fused in: ln1 + (query | key | value) product -> causal attention -> output projection + residual -> ln2 + fc1 + GELU + fc2 + residual
            # (the [B T][480] hidden activations stay in registers)
            from . import capi
            L = capi.load()
            st = torch.cuda.current_stream(x.device).cuda_stream
            B, T, C = x.shape
            M = B * T
            a = self.attn
            if not torch.cuda.is_current_stream_capturing():
                self.ensure_packed()
            else:
                assert getattr(self, "_pack_key", None) is not None, "a captured graph replays the packed weight buffers: call ensure_packed() before capturing"
            f16x3 = policy_gemm_mode() == "f16x3"
            linear = L.d3il_linear120_f16x3 if f16x3 else L.d3il_linear120_f32
            mlp = L.d3il_mlp_ln_gelu_residual_f16x3 if f16x3 else L.d3il_mlp_ln_gelu_residual_f32
            w_qkv, w_proj, w_mlp = (self._hp_qkv, self._hp_proj, self._hp_mlp) if f16x3 else (self._wp_qkv, self._wp_proj, self._wp_mlp)
            if f16x3 and T <= 16 and a.n_head == 6 and self._hp_attn.shape[0] == 32 and os.environ.get("D3IL_POLICY_FUSED_ATTN", "1") == "1":
                # the attention half in ONE launch, one wave per sequence: q | k | v and the attention output never leave the CU (csrc/policy_f16x3.h k_attn_half_f16x3)
                x1 = torch.empty_like(x)
                capi.check(L.d3il_attn_half_f16x3(x.data_ptr(), self.ln1.weight.data_ptr(), self.ln1.bias.data_ptr(), float(self.ln1.eps), self._hp_attn.data_ptr(),
                                                  self._b_qkv.data_ptr(), a.proj.bias.data_ptr(), x1.data_ptr(), B, T, a.n_head, C, st))
                if keep is not None:
                    x1 = x1.index_select(1, keep)
                    M = x1.shape[0] * x1.shape[1]
            else:
                qkv = torch.empty(B, T, 3 * C, dtype=torch.float32, device=x.device)
                capi.check(linear(x.data_ptr(), self.ln1.weight.data_ptr(), self.ln1.bias.data_ptr(), float(self.ln1.eps), w_qkv.data_ptr(), self._b_qkv.data_ptr(), None,
                                  qkv.data_ptr(), M, 3 * C, st))
                y = torch.empty_like(x)
                capi.check(L.d3il_attention_causal_f32(qkv.data_ptr(), y.data_ptr(), B, T, a.n_head, C // a.n_head, st))
                if keep is not None:
                    y, x = y.index_select(1, keep), x.index_select(1, keep)
                    M = y.shape[0] * y.shape[1]
                x1 = torch.empty_like(x)
                capi.check(linear(y.data_ptr(), None, None, 0.0, w_proj.data_ptr(), a.proj.bias.data_ptr(), x.data_ptr(), x1.data_ptr(), M, C, st))
            fc1, fc2 = self.mlp[0], self.mlp[2]
            out = torch.empty_like(x1)
            capi.check(mlp(x1.data_ptr(), self.ln2.weight.data_ptr(), self.ln2.bias.data_ptr(), float(self.ln2.eps), x1.data_ptr(), w_mlp.data_ptr(),
                           fc1.bias.data_ptr(), fc2.bias.data_ptr(), out.data_ptr(), M, 120, 480, st))
            return out
        x = x + self.attn(_layer_norm(self.ln1, x))
        if keep is not None:
            x = x.index_select(1, keep)
        return x + self.mlp(_layer_norm(self.ln2, x))


class DiffusionGPT(nn.Module):
    """DiffusionGPT (score_gpts.py:118-361), not goal conditioned: tokens = [sigma, s_1, a_1, ..., s_t, a_t], causal transformer,
    the action positions are decoded."""

    def __init__(self, state_dim, action_dim, embed_dim, n_layers, n_heads, obs_seq_len, linear_output=True):
        super().__init__()
        block_size = 2 * obs_seq_len + 1
        self.tok_emb = nn.Linear(state_dim, embed_dim)
        self.pos_emb = nn.Parameter(torch.zeros(1, obs_seq_len + 1, embed_dim))
        self.blocks = nn.Sequential(*[_Block(embed_dim, n_heads, block_size) for _ in range(n_layers)])
        self.ln_f = nn.LayerNorm(embed_dim)
        self.sigma_emb = nn.Linear(1, embed_dim)
        self.action_emb = nn.Linear(action_dim, embed_dim)
        self.action_pred = nn.Linear(embed_dim, action_dim) if linear_output else nn.Sequential(nn.Linear(embed_dim, 100), nn.SiLU(), nn.Linear(100, action_dim))
        self.obs_seq_len, self.embed_dim = obs_seq_len, embed_dim

    # ---- the sampling loop's form (BESOPolicy._sample on the device): everything that does not change between the sampling steps of one predict call is computed
    # once (state tokens, position rows, the sigma embeddings of the whole schedule), the token buffer [b, 2 t + 1, C] is filled in place - 3 small kernels per
    # step in front of the blocks instead of ~12 (mul, 2 linear, 2 add, log, div, full, stack, permute copy, cat)
    def begin_sampling(self, states, sigmas):
        b, t, _ = states.size()
        C = self.embed_dim
        pos = self.pos_emb[0, :t, :]
        xbuf = torch.empty(b, 2 * t + 1, C, dtype=torch.float32, device=states.device)
        torch.add(self.tok_emb(states), pos, out=xbuf[:, 1::2])                              # state tokens: the same in every sampling step
        emb_all = self.sigma_emb(sigmas.reshape(-1, 1).log() / 4)                                # [steps, C]; sigmas: a DEVICE tensor (no host copy inside a captured loop)
        bias_pos = (self.action_emb.bias + pos).repeat(b, 1)                                  # [b t, C]: action_emb's bias + position rows
        return dict(xbuf=xbuf, emb_all=emb_all, bias_pos=bias_pos, keep=torch.arange(2, 2 * t + 1, 2, device=states.device), t=t, b=b)

    def forward_step(self, ctx, actions, i: int, c_in: float):
        """The network on (states of begin_sampling, c_in * actions, sigma_i): same numbers as forward() up to f32 rounding of c_in (W a) against W (c_in a)."""
        b, t, xbuf = ctx["b"], ctx["t"], ctx["xbuf"]
        xbuf[:, 0] = ctx["emb_all"][i]
        xbuf[:, 2::2] = torch.addmm(ctx["bias_pos"], actions.reshape(b * t, -1), self.action_emb.weight.t(), alpha=c_in).view(b, t, -1)
        x = xbuf
        for blk in self.blocks[:-1]:
            x = blk(x)
        x = _layer_norm(self.ln_f, self.blocks[-1](x, keep=ctx["keep"]).contiguous())
        return self.action_pred(x)

    def forward(self, states, actions, sigma):
        b, t, _ = states.size()
        emb_t = self.sigma_emb((sigma.log() / 4).reshape(b, 1).to(torch.float32)).unsqueeze(1)
        pos = self.pos_emb[:, :t, :]
        state_x, action_x = self.tok_emb(states) + pos, self.action_emb(actions) + pos
        sa = torch.stack([state_x, action_x], dim=1).permute(0, 2, 1, 3).reshape(b, 2 * t, self.embed_dim)
        x = torch.cat([emb_t, sa], dim=1)
        for blk in self.blocks[:-1]:
            x = blk(x)
        # only the action positions (tokens 2, 4, ..., 2 t) are decoded: the last block's output projection and MLP, the final LayerNorm and the head run
        # on those rows only (row-wise operations: the same numbers as decoding everything and slicing, score_gpts.py:340-361)
        keep = torch.arange(2, 2 * t + 1, 2, device=x.device)
        x = _layer_norm(self.ln_f, self.blocks[-1](x, keep=keep).contiguous())
        return self.action_pred(x)


# ------------------------------------------------------------------------------------------------ history ring buffer
class _History:
    """Device ring buffer [N, W, dim], newest entry last, with a per-lane length: the deque(maxlen=W) of the reference agents for a
    batch.  Lanes are grouped by length for the network call (all lanes run in lock step unless some restart their trajectory)."""

    def __init__(self, n, w, dim, device):
        self.buf = torch.zeros(n, w, dim, dtype=torch.float32, device=device)
        self.len = torch.zeros(n, dtype=torch.int64, device=device)
        self.w = w
        self.lockstep = 0           # host copy of the common length, or -1 when lanes differ

    def reset(self, mask=None):
        if mask is None:
            self.len.zero_(); self.lockstep = 0
        else:
            self.len = torch.where(mask.bool(), torch.zeros_like(self.len), self.len)
            self.lockstep = -1

    def append(self, x):
        self.buf = torch.cat((self.buf[:, 1:], x.unsqueeze(1)), dim=1)
        self.len = (self.len + 1).clamp_max(self.w)
        if self.lockstep >= 0:
            self.lockstep = min(self.lockstep + 1, self.w)

    def groups(self):
        """[(L, lane index tensor or None for all lanes)] - one entry when the lanes are in lock step."""
        if self.lockstep >= 0:
            return [(self.lockstep, None)]
        lens = torch.unique(self.len).tolist()          # host sync, only while lanes differ
        if len(lens) == 1:
            self.lockstep = int(lens[0])
            return [(self.lockstep, None)]
        return [(int(L), torch.nonzero(self.len == L).reshape(-1)) for L in lens]


# ------------------------------------------------------------------------------------------------ policies
class BCPolicy:
    """BC_Agent.predict (agents/bc_agent.py:240-271) on a batch: scale, MLP, clamp to the data bounds, inverse scale."""

    def __init__(self, model: ResidualMLP, scaler: Scaler, min_action, max_action):
        self.model, self.scaler = model.eval(), scaler
        dev = scaler.x_mean.device
        self.min_action, self.max_action = torch.as_tensor(min_action, device=dev), torch.as_tensor(max_action, device=dev)

    def reset(self):
        pass

    def ensure_packed(self):
        self.model.ensure_packed()

    @torch.no_grad()
    def predict_batch(self, obs):
        out = self.model(self.scaler.scale_input(obs.to(torch.float32)))
        return self.scaler.inverse_scale_output(torch.clamp(out, self.min_action, self.max_action))


def pack_resmlp_weights(lin_in, blocks, lin_out) -> dict:
    """Linear, residual blocks [(l1, l2), ..], Linear of a ResidualMLPNetwork in the operand order of k_resmlp_f32 (the order of pack_ddpm_weights for any hidden
    width that is a multiple of 16): [T_out][t][lane (g, i)][r] = W[16 T_out + i][16 t + 4 g + r]."""
    dev = lin_in.weight.device
    H = lin_in.out_features
    NT = H // 16
    ar = lambda k: torch.arange(k, device=dev)
    To, t, g, i, r = ar(NT)[:, None, None, None, None], ar(NT)[None, :, None, None, None], ar(4)[None, None, :, None, None], ar(16)[None, None, None, :, None], ar(4)[None, None, None, None, :]
    pack = lambda W: W[16 * To + i, 16 * t + 4 * g + r].reshape(NT, NT, 64, 4)
    wi = torch.zeros(H, 32, device=dev)
    wi[:, :lin_in.in_features] = lin_in.weight
    w_in = wi[16 * ar(NT)[:, None, None, None] + ar(16)[None, None, :, None], 4 * ar(8)[None, None, None, :] + ar(4)[None, :, None, None]].reshape(NT, 64, 8)
    wo = torch.zeros(16, H, device=dev)
    wo[:lin_out.out_features] = lin_out.weight
    w_out = wo[i[0], 16 * t[0] + 4 * g[0] + r[0]].reshape(NT, 64, 4)
    bo = torch.zeros(16, device=dev)
    bo[:lin_out.out_features] = lin_out.bias
    if blocks:
        w_blk = torch.stack([pack(l.weight) for b in blocks for l in b])
        b_blk = torch.stack([l.bias for b in blocks for l in b])
    else:
        w_blk, b_blk = torch.zeros(1, NT, NT, 64, 4, device=dev), torch.zeros(1, H, device=dev)
    f = lambda x: x.detach().to(torch.float32).contiguous()
    return {"w_in": f(w_in), "b_in": f(lin_in.bias), "w_blk": f(w_blk), "b_blk": f(b_blk), "w_out": f(w_out), "b_out": f(bo), "n_blocks": len(blocks)}


class FusedResMLP:
    """The device path of a ResidualMLPNetwork (csrc/rollout.hip k_resmlp_f32 through d3il_resmlp_f32): packed weights in persistent buffers, refreshed in place
    when a parameter's version counter has changed (a captured graph keeps reading current weights).  Every call takes ``parts`` = (lin_in, [(l1, l2), ..],
    lin_out) of the module it serves (no reference to the module is kept: a deep copy of the module gets its own buffers and packs ITS weights)."""

    def __init__(self):
        self._fw, self._key = None, None

    def invalidate(self):
        """Force a repack at the next ensure_packed().  Needed after writes that bypass the tensors' version counters - ``param.data.copy_(...)`` as the
        reference's EMA helper does in copy_to / restore (agents/models/.../ema.py) - which the (data_ptr, _version) key cannot see; ``use_ema()``,
        ``load_state_dict`` and in-place ops on the parameters themselves are detected without it."""
        self._key = None

    def ok(self, x, parts):
        lin_in, blocks, lin_out = parts
        w = lin_in.weight
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for l in [lin_in, lin_out] + [m for b in blocks for m in b] for p in (l.weight, l.bias))):
            return False      # inference only: no autograd node (ANY trainable layer sends the call to torch's layers, not only a trainable first one)
        return (x.is_cuda and x.dim() == 2 and w.is_cuda and w.dtype == torch.float32 and lin_in.out_features in (128, 256) and lin_in.in_features <= 28
                and lin_out.out_features <= 16 and os.environ.get("D3IL_POLICY_FUSED_RESMLP", "1") == "1")

    def ensure_packed(self, parts):
        lin_in, blocks, lin_out = parts
        params = [lin_in.weight, lin_in.bias, lin_out.weight, lin_out.bias] + [p for b in blocks for l in b for p in (l.weight, l.bias)]
        key = tuple((p.data_ptr(), p._version) for p in params)
        if self._key == key:
            return
        with torch.no_grad():
            fw = pack_resmlp_weights(lin_in, blocks, lin_out)
            if self._fw is not None and all(not torch.is_tensor(v) or v.shape == self._fw[k].shape for k, v in fw.items()):
                for k, v in fw.items():
                    if torch.is_tensor(v):
                        self._fw[k].copy_(v)
            else:
                self._fw = fw
        self._key = key

    def __call__(self, x, parts):
        from . import capi
        lib = capi.load()
        lin_in, blocks, lin_out = parts
        if not torch.cuda.is_current_stream_capturing():
            self.ensure_packed(parts)
        else:
            assert self._key is not None, "a captured graph replays the packed weight buffers: call ensure_packed() before capturing"
        x = x.to(torch.float32).contiguous()
        out = torch.empty(x.shape[0], lin_out.out_features, dtype=torch.float32, device=x.device)
        w = self._fw
        capi.check(lib.d3il_resmlp_f32(x.data_ptr(), w["w_in"].data_ptr(), w["b_in"].data_ptr(), w["w_blk"].data_ptr(), w["b_blk"].data_ptr(), w["w_out"].data_ptr(), w["b_out"].data_ptr(),
                                       out.data_ptr(), x.shape[0], lin_in.in_features, lin_in.out_features, w["n_blocks"], lin_out.out_features, torch.cuda.current_stream(x.device).cuda_stream))
        return out


def pack_ddpm_weights(model: "DiffusionMLP") -> dict:
    """DiffusionMLP (hidden 256, action 2, t_dim 8) in the operand order of k_ddpm_mlp_f32: lane (g, i) = 16 g + i of a wave holds, for output tile T_out and
    input group t, the four weights W[16 T_out + i][16 t + 4 g + r] (r = 0..3) - the D registers of one layer are the B operands of the next."""
    L = model.layers.layers
    lin_in, blocks, lin_out = L[0], list(L[1:-1]), L[-1]
    dev = lin_in.weight.device
    ar = lambda k: torch.arange(k, device=dev)
    To, t, g, i, r = ar(16)[:, None, None, None, None], ar(16)[None, :, None, None, None], ar(4)[None, None, :, None, None], ar(16)[None, None, None, :, None], ar(4)[None, None, None, None, :]
    pack = lambda W: W[16 * To + i, 16 * t + 4 * g + r].reshape(16, 16, 64, 4)
    wi = torch.zeros(256, 32, device=dev)
    wi[:, :lin_in.in_features] = lin_in.weight
    s8 = ar(8)[None, None, None, :]
    w_in = wi[16 * ar(16)[:, None, None, None] + ar(16)[None, None, :, None], 4 * s8 + ar(4)[None, :, None, None]].reshape(16, 64, 8)
    wo = torch.zeros(16, 256, device=dev)
    wo[:2] = lin_out.weight
    w_out = wo[i[0], 16 * t[0] + 4 * g[0] + r[0]].reshape(16, 64, 4)
    if blocks:
        w_blk = torch.stack([pack(l.weight) for b in blocks for l in (b.l1, b.l2)])
        b_blk = torch.stack([l.bias for b in blocks for l in (b.l1, b.l2)])
    else:
        w_blk, b_blk = torch.zeros(1, 16, 16, 64, 4, device=dev), torch.zeros(1, 256, device=dev)
    f = lambda x: x.detach().to(torch.float32).contiguous()
    return {"w_in": f(w_in), "b_in": f(lin_in.bias), "w_blk": f(w_blk), "b_blk": f(b_blk), "w_out": f(w_out), "b_out": f(lin_out.bias), "n_blocks": len(blocks)}


class CapturedPolicy:
    """Any policy whose ``predict_batch`` is a fixed chain of device kernels on a fixed batch shape (no host round trip, no data-dependent shapes: BCPolicy,
    the stand-in MLP of agents.py, DDPMPolicy with window_size 1) as ONE captured HIP graph: the first call of a batch shape warms the chain up on a side
    stream and captures it, later calls copy the observation into the graph's static input and replay it on the caller's current stream.  Same kernels, same
    results; the host issues one launch instead of dozens - which is what bounds several sub-batches on several streams.  Random
    draws inside the chain (torch.randn on the device) come from the device generator at every replay.  The returned tensor is the graph's static output:
    consume it before the next call (the rollout loops do)."""

    def __init__(self, inner):
        self.inner = inner
        self._g, self._g_in, self._g_out = None, None, None

    def reset(self):
        if hasattr(self.inner, "reset"):
            self.inner.reset()

    def begin_episodes(self, mask):
        if hasattr(self.inner, "begin_episodes"):
            self.inner.begin_episodes(mask)

    def set_rollout_range(self, offset, count):
        if hasattr(self.inner, "set_rollout_range"):
            self.inner.set_rollout_range(offset, count)

    def fork(self):
        """A clone for another sub-batch: the inner policy forked by its own rule, graph and static buffers its own."""
        from .envs.sub_batch import fork_agent
        return CapturedPolicy(fork_agent(self.inner))

    @torch.no_grad()
    def predict_batch(self, obs):
        if not obs.is_cuda:
            return self.inner.predict_batch(obs)
        if self._g is None or self._g_in.shape != obs.shape or self._g_in.dtype != obs.dtype:
            dev = obs.device
            self._g_in = obs.clone()
            cur = torch.cuda.current_stream(dev)
            side = torch.cuda.Stream(dev)      # warm-up outside the capture: library workspaces and lazy initialisation must not happen inside it
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(2):
                    self.inner.predict_batch(self._g_in)
            cur.wait_stream(side)
            self._g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g):
                self._g_out = self.inner.predict_batch(self._g_in)
        if hasattr(self.inner, "ensure_packed"):
            self.inner.ensure_packed()      # packed weight buffers of a fused policy follow the parameters (in place) - e.g. after the EMA swap of a rollout
        self._g_in.copy_(obs)
        self._g.replay()
        return self._g_out


def cosine_beta_schedule(timesteps, s=0.008):     # agents/models/diffusion/utils.py:31-42 (float64 numpy -> float32)
    import numpy as np
    steps = timesteps + 1
    x = np.linspace(0, steps, steps)
    ac = np.cos(((x / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
    ac = ac / ac[0]
    return torch.tensor(np.clip(1 - (ac[1:] / ac[:-1]), a_min=0, a_max=0.999), dtype=torch.float32)


class DDPMPolicy:
    """DiffusionAgent.predict (ddpm_agent.py:213-274) with the Diffusion sampler (gc_diffusion.py:101-216: epsilon prediction, clipped
    x0, posterior mean / variance, n_timesteps ancestral steps, final clamp) on a batch.  ``noise_fn(shape)`` supplies the Gaussian
    noise (default torch.randn on the policy's device); window_size > 1 keeps the observation history per lane."""

    def __init__(self, model: DiffusionMLP, scaler: Scaler, n_timesteps: int, window_size: int = 1, n_envs: int | None = None, noise_fn=None):
        self.model, self.scaler, self.T, self.W = model.eval(), scaler, int(n_timesteps), int(window_size)
        dev = scaler.x_mean.device
        self.device = dev
        betas = cosine_beta_schedule(self.T).to(dev)
        alphas = 1.0 - betas
        ac = torch.cumprod(alphas, dim=0)
        ac_prev = torch.cat([torch.ones(1, device=dev), ac[:-1]])
        self.sqrt_recip_ac, self.sqrt_recipm1_ac = torch.sqrt(1.0 / ac), torch.sqrt(1.0 / ac - 1)
        post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
        self.post_logvar = torch.log(torch.clamp(post_var, min=1e-20))
        self.coef1, self.coef2 = betas * torch.sqrt(ac_prev) / (1.0 - ac), (1.0 - ac_prev) * torch.sqrt(alphas) / (1.0 - ac)
        self.min_action, self.max_action = scaler.y_bounds[0], scaler.y_bounds[1]
        self._custom_noise = noise_fn is not None
        self.noise_fn = noise_fn or (lambda shape: torch.randn(shape, device=dev))
        self.hist = None
        self.n_envs = n_envs

    # ---- the whole chain in one kernel of the rollout library (csrc/rollout.hip k_ddpm_mlp_f32)
    def fused_ok(self):
        m = self.model
        L = m.layers.layers
        return (self.W <= 1 and isinstance(m, DiffusionMLP) and L[0].weight.is_cuda and L[0].weight.dtype == torch.float32 and L[0].out_features == 256 and L[-1].out_features == 2
                and m.temp_layers[-1].out_features == 8 and 1 <= L[0].in_features - 10 <= 18 and all(isinstance(b, _ResBlock) for b in L[1:-1])
                and os.environ.get("D3IL_POLICY_FUSED_DDPM", "1") == "1")

    def invalidate_packed(self):
        """Force a repack at the next call.  Needed after writes through ``param.data`` (e.g. the reference EMA helper's copy_to / restore), which do not bump the
        version counters ensure_packed() keys on; use_ema() and load_state_dict are seen without it."""
        self._pack_key = None
        self.model.layers.invalidate_packed()

    def ensure_packed(self):
        """The denoiser's weights in the tile order of the kernel, the time embeddings of the T steps and the schedule table, in persistent device buffers
        refreshed whenever a parameter has changed (tensor version counters) - e.g. after the EMA swap of a rollout."""
        params = list(self.model.parameters())
        key = tuple((p.data_ptr(), p._version) for p in params)
        if getattr(self, "_pack_key", None) == key:
            return
        self.model.layers.ensure_packed()      # (the torch chain's inner network, if its device path has been used)
        L = self.model.layers.layers
        dev = L[0].weight.device
        with torch.no_grad():
            fw = pack_ddpm_weights(self.model)
            fw["temb"] = self.model.temp_layers(torch.arange(self.T, device=dev)).to(torch.float32).contiguous()
            sig = (0.5 * self.post_logvar).exp() * torch.cat((torch.zeros(1, device=dev), torch.ones(self.T - 1, device=dev)))
            fw["sched"] = torch.stack((self.sqrt_recip_ac, self.sqrt_recipm1_ac, self.coef1, self.coef2, sig), dim=1).to(torch.float32).contiguous()
            fw["bounds"] = torch.cat((self.min_action.reshape(-1), self.max_action.reshape(-1))).to(torch.float32).contiguous()
            old = getattr(self, "_fw", None)
            if old is not None and all(torch.is_tensor(v) == torch.is_tensor(old.get(k)) and (not torch.is_tensor(v) or v.shape == old[k].shape) for k, v in fw.items()):
                for k, v in fw.items():      # in place: the addresses never change, a captured graph keeps reading the current weights
                    if torch.is_tensor(v):
                        old[k].copy_(v)
            else:
                self._fw = fw
        self._temb, self._sched, self._bounds = self._fw["temb"], self._fw["sched"], self._fw["bounds"]
        self._pack_key = key

    def _sample_fused(self, state):
        from . import capi
        lib = capi.load()
        n, sd = state.shape
        if not torch.cuda.is_current_stream_capturing():
            self.ensure_packed()
        else:
            assert getattr(self, "_pack_key", None) is not None, "a captured graph replays the packed weight buffers: call ensure_packed() before capturing"
        shape = (n, 2)
        noise = torch.stack([self.noise_fn(shape) for _ in range(self.T + 1)]).to(torch.float32).contiguous() if self._custom_noise else torch.randn((self.T + 1, n, 2), device=self.device)
        out = torch.empty(n, 2, dtype=torch.float32, device=self.device)
        w = self._fw
        st = torch.cuda.current_stream(self.device).cuda_stream
        state = state.contiguous()
        capi.check(lib.d3il_ddpm_mlp_f32(state.data_ptr(), noise.data_ptr(), self._temb.data_ptr(), w["w_in"].data_ptr(), w["b_in"].data_ptr(), w["w_blk"].data_ptr(), w["b_blk"].data_ptr(),
                                         w["w_out"].data_ptr(), w["b_out"].data_ptr(), self._sched.data_ptr(), self._bounds.data_ptr(), out.data_ptr(), n, sd, self.T, 256, w["n_blocks"], st))
        return out

    def fork(self):
        """A clone for another sub-batch (envs/sub_batch.fork_agent): network, scaler and schedule shared, observation history and the packed buffers of the
        fused chain its own (packed again at its first call)."""
        import copy
        c = copy.copy(self)
        c.hist = copy.deepcopy(self.hist)
        c._fw, c._pack_key = None, None
        return c

    def captured(self):
        """window_size 1: the whole predict chain (input scaling, the T denoising steps with their noise draws - ~60 torch kernels each -, clamp, output scaling)
        as one captured graph (CapturedPolicy)."""
        assert self.W <= 1, "a history window regroups the lanes by history length at every call: not a fixed chain"
        return CapturedPolicy(self)

    def load_reference_state_dict(self, sd):
        """``Diffusion.state_dict()`` of the reference: the denoiser sits under ``model.``."""
        self.model.load_state_dict({k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")})

    def use_ema(self, shadow_params):
        """One EMA swap per rollout (the reference swaps per predict call): shadow parameters in ``model.parameters()`` order."""
        with torch.no_grad():
            for p, s in zip(self.model.parameters(), shadow_params):
                p.copy_(torch.as_tensor(s, dtype=p.dtype, device=p.device))

    def reset(self):
        if self.hist is not None:
            self.hist.reset()

    def begin_episodes(self, mask):
        if self.hist is not None:
            self.hist.reset(mask)

    def _sample(self, state):
        shape = (state.shape[0], state.shape[1], self.min_action.shape[0]) if state.dim() == 3 else (state.shape[0], self.min_action.shape[0])
        x = self.noise_fn(shape)
        for i in reversed(range(self.T)):
            t = torch.full((shape[0],), i, device=self.device, dtype=torch.long)
            eps = self.model(x, t, state)
            x0 = (self.sqrt_recip_ac[i] * x - self.sqrt_recipm1_ac[i] * eps).clamp(self.min_action, self.max_action)
            mean = self.coef1[i] * x0 + self.coef2[i] * x
            noise = self.noise_fn(shape)
            x = mean + (0.0 if i == 0 else 1.0) * (0.5 * self.post_logvar[i]).exp() * noise
        return x.clamp(self.min_action, self.max_action)

    @torch.no_grad()
    def predict_batch(self, obs):
        s = self.scaler.scale_input(obs.to(device=self.device, dtype=torch.float32))
        if self.W <= 1:
            if s.is_cuda and s.dim() == 2 and self.fused_ok():
                return self.scaler.inverse_scale_output(self._sample_fused(s))
            return self.scaler.inverse_scale_output(self._sample(s))
        if self.hist is None:
            self.hist = _History(s.shape[0], self.W, s.shape[1], self.device)
        self.hist.append(s)
        out = torch.empty(s.shape[0], self.min_action.shape[0], device=self.device)
        for L, idx in self.hist.groups():
            st = self.hist.buf[:, self.W - L:] if idx is None else self.hist.buf[idx, self.W - L:]
            a = self._sample(st)[:, -1, :]
            if idx is None:
                out = a
            else:
                out[idx] = a
        return self.scaler.inverse_scale_output(out)


class BESOPolicy:
    """BesoAgent.predict (beso_agent.py:316-443) on a batch: observation history (deque maxlen W) and action history (deque maxlen
    W - 1 of the clamped scaled actions), x_T = N(0, sigma_max^2) for the newest action, Euler-ancestral sampling over the linear noise
    schedule (beso_agent.py:122, gc_sampling.py:41-44, 217-256) of the Karras-preconditioned DiffusionGPT (score_wrappers.py:33-99),
    last action of the sequence, clamp, inverse scale."""

    def __init__(self, inner: DiffusionGPT, scaler: Scaler, window_size: int, num_sampling_steps: int, sigma_min: float, sigma_max: float,
                 sigma_data: float = 0.5, noise_fn=None, use_graph: bool = False):
        self.inner, self.scaler, self.W = inner.eval(), scaler, int(window_size)
        # use_graph: capture the sampling loop for full windows ([N, W] sequences, ~1500 small kernels) in a HIP graph and replay it
        # (default noise only: the generator state is part of the capture)
        self.use_graph = bool(use_graph) and noise_fn is None
        self._graph = None
        dev = scaler.x_mean.device
        self.device = dev
        self.n_steps, self.sigma_min, self.sigma_max, self.sigma_data = int(num_sampling_steps), float(sigma_min), float(sigma_max), float(sigma_data)
        self.min_action, self.max_action = scaler.y_bounds[0], scaler.y_bounds[1]
        self.noise_fn = noise_fn or (lambda shape: torch.randn(shape, device=dev))
        self.obs_hist = self.act_hist = None
        # the noise schedule as host floats (one transfer here instead of a device synchronisation per sampling step)
        self.sigmas = torch.cat([torch.linspace(self.sigma_max, self.sigma_min, self.n_steps, device=dev), torch.zeros(1, device=dev)]).tolist()

    def load_reference_state_dict(self, sd):
        """``GCDenoiser.state_dict()`` of the reference: the transformer sits under ``inner_model.``."""
        own = self.inner.state_dict()
        self.inner.load_state_dict({k[len("inner_model."):]: v for k, v in sd.items() if k.startswith("inner_model.") and k[len("inner_model."):] in own})

    def use_ema(self, shadow_params):
        with torch.no_grad():
            for p, s in zip(self.inner.parameters(), shadow_params):
                p.copy_(torch.as_tensor(s, dtype=p.dtype, device=p.device))

    def reset(self):
        for h in (self.obs_hist, self.act_hist):
            if h is not None:
                h.reset()

    def begin_episodes(self, mask):
        for h in (self.obs_hist, self.act_hist):
            if h is not None:
                h.reset(mask)

    def _denoise(self, states, actions, sigma: float):
        sd2 = self.sigma_data ** 2
        c_skip, c_out, c_in = sd2 / (sigma ** 2 + sd2), sigma * self.sigma_data / (sigma ** 2 + sd2) ** 0.5, 1 / (sigma ** 2 + sd2) ** 0.5
        s_in = torch.full((actions.shape[0],), sigma, device=self.device)
        return self.inner(states, actions * c_in, s_in) * c_out + actions * c_skip

    def _sample_device(self, states, x):
        """_sample for the device: step-invariant work hoisted (DiffusionGPT.begin_sampling), the Karras combination and the Euler-ancestral update of a step
        merged algebraically - den = c_out net + c_skip x;  x' = x + (x - den) (s_down - s_from) / s_from  =  (1 + k (1 - c_skip)) x - k c_out net  with
        k = (s_down - s_from) / s_from - two element-wise kernels instead of seven (f32 rounding differs from the step-by-step form at the 1e-7 level)."""
        sigmas, sd2 = self.sigmas, self.sigma_data ** 2
        if getattr(self, "_sig_dev", None) is None or self._sig_dev.device != states.device:
            self._sig_dev = torch.tensor(sigmas[:-1], dtype=torch.float32, device=states.device)
        ctx = self.inner.begin_sampling(states, self._sig_dev)
        for i in range(len(sigmas) - 1):
            s_from, s_to = sigmas[i], sigmas[i + 1]
            c_skip, c_out, c_in = sd2 / (s_from ** 2 + sd2), s_from * self.sigma_data / (s_from ** 2 + sd2) ** 0.5, 1 / (s_from ** 2 + sd2) ** 0.5
            net = self.inner.forward_step(ctx, x, i, c_in)
            s_up = min(s_to, (s_to ** 2 * (s_from ** 2 - s_to ** 2) / s_from ** 2) ** 0.5)
            s_down = (s_to ** 2 - s_up ** 2) ** 0.5
            k = (s_down - s_from) / s_from
            x = torch.add(x * (1.0 + k * (1.0 - c_skip)), net, alpha=-k * c_out)
            if s_down > 0:
                x = torch.add(x, self.noise_fn(tuple(x.shape)), alpha=s_up)
        return x

    def _sample(self, states, x):
        if states.is_cuda and os.environ.get("D3IL_POLICY_BESO_FUSED_GLUE", "1") == "1" and not (torch.is_grad_enabled() and any(p.requires_grad for p in self.inner.parameters())):
            return self._sample_device(states, x)
        sigmas = self.sigmas
        for i in range(len(sigmas) - 1):
            s_from, s_to = sigmas[i], sigmas[i + 1]
            den = self._denoise(states, x, s_from)
            s_up = min(s_to, (s_to ** 2 * (s_from ** 2 - s_to ** 2) / s_from ** 2) ** 0.5)
            s_down = (s_to ** 2 - s_up ** 2) ** 0.5
            x = x + (x - den) / s_from * (s_down - s_from)
            if s_down > 0:
                x = x + self.noise_fn(tuple(x.shape)) * s_up
        return x

    def _sample_full(self, states, x):
        """``_sample`` for full windows; with ``use_graph`` one HIP-graph replay instead of the eager kernel sequence."""
        if not (self.use_graph and states.is_cuda and states.shape[1] == self.W):
            return self._sample(states, x)
        if self._graph is None or self._g_st.shape != states.shape:
            self._g_st, self._g_x = states.clone(), x.clone()
            side = torch.cuda.Stream(device=states.device)
            side.wait_stream(torch.cuda.current_stream(states.device))
            with torch.cuda.stream(side):                      # warm-up outside the capture (library handles, workspaces)
                for _ in range(2):
                    self._sample(self._g_st, self._g_x)
            torch.cuda.current_stream(states.device).wait_stream(side)
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._g_out = self._sample(self._g_st, self._g_x)
        self._g_st.copy_(states); self._g_x.copy_(x)
        self._graph.replay()
        return self._g_out

    def _padded_inputs(self, noise):
        """Lanes with different history lengths in ONE batch: the sequences are left-aligned and padded on the right to the window
        size.  The transformer is causal and its position embedding counts from the first token, so the tokens of a lane never see
        the padding behind them; the newest action of lane i sits at position len_i - 1."""
        W, L = self.W, self.obs_hist.len
        j = torch.arange(W, device=self.device)
        src = ((W - L).unsqueeze(1) + j).clamp_max(W - 1)
        st = torch.gather(self.obs_hist.buf, 1, src.unsqueeze(2).expand(-1, -1, self.obs_hist.buf.shape[2])) * (j < L.unsqueeze(1)).unsqueeze(2)
        newest = (j == (L - 1).unsqueeze(1)).unsqueeze(2)
        x = newest * noise
        if self.act_hist is not None:
            srca = ((W - L).unsqueeze(1) + j).clamp(0, W - 2)
            xa = torch.gather(self.act_hist.buf, 1, srca.unsqueeze(2).expand(-1, -1, self.act_hist.buf.shape[2]))
            x = x + xa * (j < (L - 1).unsqueeze(1)).unsqueeze(2)
        return st, x

    @torch.no_grad()
    def predict_batch(self, obs):
        s = self.scaler.scale_input(obs.to(device=self.device, dtype=torch.float32))
        n, act_dim = s.shape[0], self.min_action.shape[0]
        if s.is_cuda and getattr(self, "_sig_dev", None) is None:
            self._sig_dev = torch.tensor(self.sigmas[:-1], dtype=torch.float32, device=s.device)      # (made here, outside a captured sampling loop)
        if s.is_cuda:                          # packed weight copies of the fused blocks: refreshed here, OUTSIDE a captured sampling loop (an EMA swap changes them)
            for blk in self.inner.blocks:
                if blk._fused_static_ok():
                    blk.ensure_packed()
        if self.obs_hist is None:
            self.obs_hist = _History(n, self.W, s.shape[1], self.device)
            self.act_hist = _History(n, self.W - 1, act_dim, self.device) if self.W > 1 else None
        self.obs_hist.append(s)
        noise = self.noise_fn((n, 1, act_dim)) * self.sigma_max
        L = self.obs_hist.lockstep
        if L >= 0:                             # all lanes have the same history length: the reference's shapes
            st = self.obs_hist.buf[:, self.W - L:]
            x = noise
            if L > 1:                          # previous actions: the action deque holds min(L - 1, W - 1) entries
                x = torch.cat([self.act_hist.buf[:, (self.W - 1) - (L - 1):], x], dim=1)
            x0_all = self._sample_full(st, x)[:, -1, :].clamp(self.min_action, self.max_action)
        else:                                  # lanes restarted at different times: one padded batch, no host synchronisation
            st, x = self._padded_inputs(noise)
            out = self._sample_full(st, x)
            x0_all = out.gather(1, (self.obs_hist.len - 1).view(n, 1, 1).expand(-1, 1, act_dim)).squeeze(1).clamp(self.min_action, self.max_action)
        if self.act_hist is not None:
            self.act_hist.append(x0_all)
            self.act_hist.len = (self.obs_hist.len - 1).clamp_min(0).clamp_max(self.W - 1)
            self.act_hist.lockstep = -1 if self.obs_hist.lockstep < 0 else min(max(self.obs_hist.lockstep - 1, 0), self.W - 1)
        return self.scaler.inverse_scale_output(x0_all)
