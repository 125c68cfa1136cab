"""The split-f16 GEMM kernels of the batched DiffusionGPT block (csrc/policy_f16x3.h; SURVEY 8(f)-1, score_gpts.py:83-115).

CPU: the operand split and the packed tile order.  GPU: both kernels against an f64 evaluation of the same layer - the deviation of the split-f16 products is
that of an f32 FMA chain (compared with the f32-input MFMA kernels of rounds 3 - 5 on the same inputs) -, and the whole BESO policy in both modes."""
import random

import numpy as np
import pytest

torch = pytest.importorskip("torch")


def test_split_keeps_22_bits():
    from d3il_amd.policies import split_f16
    torch.manual_seed(0)
    w = torch.cat((torch.randn(4000) * 0.05, torch.randn(4000) * 3.0, torch.tensor([0.0, 1.0, -1.0, 65504.0, 1e5, -1e5, 1e-7])))
    hi, lo = split_f16(w)
    rec = hi.double() + lo.double() / 2048.0
    ref = w.double().clamp(-65504.0, 65504.0)
    err = (rec - ref).abs()
    assert float((err / ref.abs().clamp_min(6.2e-5)).max()) < 2.0 ** -21           # relative, down to the smallest normal f16
    assert float(err.max()) < 65504.0 * 2.0 ** -21


def test_packed_tile_order():
    from d3il_amd.policies import pack_linear120_weights_f16x3, pack_mlp_weights_f16x3, split_f16
    torch.manual_seed(1)
    random.seed(1)
    W1, W2 = torch.randn(480, 120) * 0.05, torch.randn(120, 480) * 0.05
    p = pack_mlp_weights_f16x3(W1, W2)
    assert tuple(p.shape) == (16, 2048, 8) and p.dtype == torch.float16      # 15 hidden pairs, staged for the software pipeline: stage k = W1 of pair k | W2 of pair k - 1
    assert float(p[15, :1024].abs().max()) == 0.0 and float(p[0, 1024:].abs().max()) == 0.0
    h1, h2 = split_f16(W1), split_f16(W2)
    for _ in range(3000):
        c, tile, s, pp, lane, e, t = (random.randrange(k) for k in (15, 2, 4, 2, 64, 8, 8))
        g, i = lane // 16, lane % 16
        k = 32 * s + 8 * g + e
        assert float(p[c, ((tile * 4 + s) * 2 + pp) * 64 + lane, e]) == (0.0 if k >= 120 else float(h1[pp][32 * c + 16 * tile + i, k]))
        row, hid = 16 * t + i, 32 * c + 16 * (e >> 2) + 4 * g + (e & 3)
        assert float(p[c + 1, 1024 + (t * 2 + pp) * 64 + lane, e]) == (0.0 if row >= 120 else float(h2[pp][row, hid]))
    for N in (120, 360):
        W = torch.randn(N, 120) * 0.05
        q = pack_linear120_weights_f16x3(W)
        assert q.shape[0] % 2 == 0 and q.shape[0] * 16 >= N and tuple(q.shape[1:]) == (512, 8)
        hw = split_f16(W)
        for _ in range(2000):
            t, s, pp, lane, e = (random.randrange(k) for k in (q.shape[0], 4, 2, 64, 8))
            g, i = lane // 16, lane % 16
            k, row = 32 * s + 8 * g + e, 16 * t + i
            assert float(q[t, (s * 2 + pp) * 64 + lane, e]) == (0.0 if (k >= 120 or row >= N) else float(hw[pp][row, k]))


def _ln64(x, w, b, eps):
    x = x.double()
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w.double() + b.double()


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [1, 63, 64, 1000, 45056])
def test_split_f16_kernels_match_an_f64_reference_like_the_f32_kernels_do(rows):
    import ctypes as C
    from d3il_amd import capi
    from d3il_amd.policies import pack_linear120_weights, pack_linear120_weights_f16x3, pack_mlp_weights_f16x3, _Block
    L = capi.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(rows)
    blk = _Block(120, 6, 11).to(dev)
    with torch.no_grad():
        for p in blk.parameters():
            p.mul_(1.7)                                   # away from the init scale
        blk.ln1.weight.add_(torch.randn(120, device=dev) * 0.3); blk.ln1.bias.add_(torch.randn(120, device=dev) * 0.2)
        blk.ln2.weight.add_(torch.randn(120, device=dev) * 0.3); blk.ln2.bias.add_(torch.randn(120, device=dev) * 0.2)
    x = (torch.randn(rows, 120, device=dev) * 2.0 + 0.3).contiguous()
    st = torch.cuda.current_stream(dev).cuda_stream
    a = blk.attn
    # ---- linear: ln1 + (query | key | value)
    W = torch.cat((a.query.weight, a.key.weight, a.value.weight), 0).detach()
    bq = torch.cat((a.query.bias, a.key.bias, a.value.bias), 0).detach().contiguous()
    ref = _ln64(x, blk.ln1.weight, blk.ln1.bias, blk.ln1.eps) @ W.double().t() + bq.double()
    out32, out16 = torch.empty(rows, 360, device=dev), torch.empty(rows, 360, device=dev)
    w32, w16 = pack_linear120_weights(W), pack_linear120_weights_f16x3(W)
    capi.check(L.d3il_linear120_f32(x.data_ptr(), blk.ln1.weight.data_ptr(), blk.ln1.bias.data_ptr(), float(blk.ln1.eps), w32.data_ptr(), bq.data_ptr(), None, out32.data_ptr(), rows, 360, st))
    capi.check(L.d3il_linear120_f16x3(x.data_ptr(), blk.ln1.weight.data_ptr(), blk.ln1.bias.data_ptr(), float(blk.ln1.eps), w16.data_ptr(), bq.data_ptr(), None, out16.data_ptr(), rows, 360, st))
    scale = float(ref.abs().max())
    e32, e16 = float((out32.double() - ref).abs().max()) / scale, float((out16.double() - ref).abs().max()) / scale
    print("linear rows %d: max |err| / max |ref|: f32 MFMA %.2e, split f16 %.2e" % (rows, e32, e16))
    assert e16 < 3e-6 and e16 < 4 * e32 + 5e-7
    # ---- projection with residual (no LayerNorm), N = 120
    Wp, bp = a.proj.weight.detach(), a.proj.bias.detach()
    y = (torch.randn(rows, 120, device=dev) * 1.5).contiguous()
    refp = y.double() @ Wp.double().t() + bp.double() + x.double()
    o16 = torch.empty(rows, 120, device=dev)
    capi.check(L.d3il_linear120_f16x3(y.data_ptr(), None, None, 0.0, pack_linear120_weights_f16x3(Wp).data_ptr(), bp.data_ptr(), x.data_ptr(), o16.data_ptr(), rows, 120, st))
    assert float((o16.double() - refp).abs().max()) / float(refp.abs().max()) < 3e-6
    # ---- MLP: x + fc2(GELU(fc1(ln2(x))))
    fc1, fc2 = blk.mlp[0], blk.mlp[2]
    hid = _ln64(x, blk.ln2.weight, blk.ln2.bias, blk.ln2.eps) @ fc1.weight.double().t() + fc1.bias.double()
    refm = x.double() + torch.nn.functional.gelu(hid) @ fc2.weight.double().t() + fc2.bias.double()
    blk.ensure_packed()
    m32, m16 = torch.empty_like(x), torch.empty_like(x)
    args = (x.data_ptr(), blk.ln2.weight.data_ptr(), blk.ln2.bias.data_ptr(), float(blk.ln2.eps), x.data_ptr())
    capi.check(L.d3il_mlp_ln_gelu_residual_f32(*args, blk._wp_mlp.data_ptr(), fc1.bias.data_ptr(), fc2.bias.data_ptr(), m32.data_ptr(), rows, 120, 480, st))
    capi.check(L.d3il_mlp_ln_gelu_residual_f16x3(*args, pack_mlp_weights_f16x3(fc1.weight, fc2.weight).data_ptr(), fc1.bias.data_ptr(), fc2.bias.data_ptr(), m16.data_ptr(), rows, 120, 480, st))
    torch.cuda.synchronize()
    scale = float(refm.abs().max())
    e32, e16 = float((m32.double() - refm).abs().max()) / scale, float((m16.double() - refm).abs().max()) / scale
    print("mlp rows %d: max |err| / max |ref|: f32 MFMA %.2e, split f16 %.2e" % (rows, e32, e16))
    assert e16 < 3e-6 and e16 < 4 * e32 + 5e-7
    assert torch.isfinite(m16).all() and torch.isfinite(out16).all()


@pytest.mark.gpu
def test_block_in_both_gemm_modes_and_against_torch(monkeypatch):
    """_Block.forward on the device: the split-f16 path (default), the f32 MFMA path (D3IL_POLICY_GEMM=f32) and torch's own layers agree to f32 accuracy; the last
    block's keep= form too."""
    from d3il_amd.policies import _Block
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    blk = _Block(120, 6, 11).to(dev).eval()
    x = torch.randn(700, 11, 120, device=dev).contiguous()
    keep = torch.arange(2, 11, 2, device=dev)
    outs = {}
    with torch.no_grad():
        for mode in ("f16x3", "f32"):
            monkeypatch.setenv("D3IL_POLICY_GEMM", mode)
            outs[mode] = (blk(x), blk(x, keep=keep))
        monkeypatch.setenv("D3IL_POLICY_FUSED_MLP", "0")
        outs["torch"] = (blk(x), blk(x, keep=keep))
    for k in (0, 1):
        ref = outs["torch"][k]
        scale = float(ref.abs().max())
        for mode in ("f16x3", "f32"):
            err = float((outs[mode][k] - ref).abs().max()) / scale
            assert err < 2e-5, (mode, k, err)
    assert tuple(outs["f16x3"][1].shape) == (700, 5, 120)


@pytest.mark.gpu
@pytest.mark.parametrize("B,T", [(1, 11), (3, 11), (4, 16), (257, 11), (4096, 11), (5, 1), (6, 7)])
def test_one_kernel_attention_half_equals_the_three_kernel_path(monkeypatch, B, T):
    """d3il_attn_half_f16x3 (one wave per sequence, q | k | v in LDS) runs the same matrix-core products and the same attention arithmetic in the same order as
    ln1 + qkv product -> d3il_attention_causal_f32 -> projection + residual: the block's output is bit-identical, with and without `keep`; and it stays within the
    f32-level distance of torch's layers."""
    from d3il_amd import policies as P
    torch.manual_seed(B * 100 + T)
    blk = P._Block(120, 6, 16).cuda().eval()
    for p_ in blk.parameters():
        p_.data.normal_(0.0, 0.08)
    x = torch.randn(B, T, 120, device="cuda")
    keep = torch.arange(T // 2, T, device="cuda")
    with torch.no_grad():
        monkeypatch.setenv("D3IL_POLICY_FUSED_ATTN", "1")
        y1, k1 = blk(x), blk(x, keep=keep)
        monkeypatch.setenv("D3IL_POLICY_FUSED_ATTN", "0")
        y0, k0 = blk(x), blk(x, keep=keep)
        monkeypatch.setenv("D3IL_POLICY_FUSED_MLP", "0")
        yt = blk(x)
    torch.cuda.synchronize()
    assert torch.equal(y1, y0) and torch.equal(k1, k0)
    assert torch.equal(k1, y1.index_select(1, keep))
    assert float((y1 - yt).abs().max()) < 2e-4 * float(yt.abs().max().clamp_min(1.0))
