"""GPU: the HIP denoiser (through the C ABI behind paella_amd.Paella) vs the golden outputs of the reference itself
(tests/golden) and, at full size, vs the CPU oracle on the same seeded weights/inputs.

Tolerance (fp32 path, different summation order than the reference's CPU BLAS): |logit diff| <= 2e-4 on logits of
std ~1 for the small fixtures, <= 1e-3 at the 570M size; argmax equality is asserted with the near-tie policy of
SURVEY section 4 (positions whose reference top1-top2 margin < 1e-4 are counted and reported, never dropped)."""
import ctypes

import numpy as np
import pytest
import torch

import paella_amd
from oracle import golden_configs as G
from oracle import paella_oracle as O
from tests.helpers import argmax_report, cond_for, to_dev, weights_for

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(cfg, golden_npz=None):
    m = paella_amd.Paella(**cfg)
    sd = weights_for(m, sum(cfg["blocks"]), golden_npz)
    return m.to(DEV), sd


@pytest.fixture(scope="module")
def tiny(golden, built_lib):
    return _model(G.UNET_TINY, golden("unet_tiny_forward"))


def _cmp(got, ref_np, atol):
    ref = torch.from_numpy(ref_np)
    got = got.float().cpu()
    assert got.shape == ref.shape
    diff = (got - ref).abs().max().item()
    assert diff <= atol, "max |logit diff| %.3e > %.1e" % (diff, atol)
    clear, near, n_near = argmax_report(ref, got)
    assert clear == 0, "%d argmax mismatches with a clear reference margin (near-tie: %d of %d)" % (clear, near, n_near)
    return diff


def test_tiny_forward_vs_reference(golden, tiny):
    m, _ = tiny
    g = golden("unet_tiny_forward")
    x, r = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["r"]).to(DEV)
    c = to_dev(cond_for(G.UNET_TINY, 2, 5, 1, G.COND_SEED), DEV)
    out = m(x, r, **c)
    assert out.shape == (2, 64, 16, 16)  # the reference's [B, num_labels, H, W]
    _cmp(out, g["logits"], 2e-4)
    np.testing.assert_allclose(m.gen_r_embedding(r).cpu().numpy(), g["r_embed"], atol=2e-6)
    np.testing.assert_allclose(m.gen_c_embeddings(**c).cpu().numpy(), g["c_embed"], atol=2e-5)
    # bit-reproducible run to run (no atomics anywhere on the path)
    assert torch.equal(out, m(x, r, **c))


def test_tiny_conditioning_variants(golden, tiny):
    m, _ = tiny
    g = golden("unet_tiny_forward")
    x, r = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["r"]).to(DEV)
    c2 = to_dev(cond_for(G.UNET_TINY, 2, 3, 0, G.COND_SEED + 1), DEV)
    _cmp(m(x, r, **c2), golden("unet_tiny_forward_textonly")["logits"], 2e-4)
    c3 = dict(c2, byt5=c2["byt5"][:, :0])  # CLIP-only: S_byt5 = 0
    _cmp(m(x, r, **c3), golden("unet_tiny_forward_cliponly")["logits"], 2e-4)


def test_attn_weights_and_clip_image_list(golden, tiny):
    m, _ = tiny
    g = golden("unet_tiny_attnw")
    paella_amd.replace_attention_layers(m)  # call-site compatibility: no-op
    x, r = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["r"]).to(DEV)
    c = to_dev(cond_for(G.UNET_TINY, 2, 5, 2, G.COND_SEED), DEV)
    _cmp(m(x, r, **c, attn_weights=torch.from_numpy(g["attn_weights"]).to(DEV)), g["logits"], 2e-4)
    _cmp(m(x, r, **c), g["logits_noaw"], 2e-4)


def test_mid_forward_head_dim_80(golden, built_lib):
    g = golden("unet_mid_forward")
    m, _ = _model(G.UNET_MID, g)
    c = to_dev(cond_for(G.UNET_MID, 1, 0, 0, G.COND_SEED), DEV)
    out = m(torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["r"]).to(DEV), **c).detach().cpu()
    np.testing.assert_allclose(out[:, :, ::2, ::2].numpy(), g["logits_sub"], atol=3e-4)
    mism = out.argmax(1).numpy() != g["argmax"]
    assert not (mism & (g["top2_margin"] > 1e-4)).any()


def test_variant_blocks(golden, built_lib):
    """F blocks, TimestepBlocks that cannot be fused, cross-attention only, patch_size 1, two levels."""
    g = golden("unet_variant_forward")
    m, _ = _model(G.UNET_VARIANT, g)
    c = to_dev(cond_for(G.UNET_VARIANT, 2, 3, 1, G.COND_SEED), DEV)
    _cmp(m(torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["r"]).to(DEV), **c), g["logits"], 2e-4)


def test_batching_is_row_independent(tiny):
    """cond+uncond batched as 2B rows equals two separate evaluations bit-for-bit per row (no cross-sample coupling)."""
    m, _ = tiny
    g = torch.Generator().manual_seed(1)
    x = torch.randint(0, 64, (3, 16, 16), generator=g).to(DEV)
    r = torch.rand(3, generator=g).to(DEV)
    c = to_dev(cond_for(G.UNET_TINY, 3, 4, 1, 11), DEV)
    full = m(x, r, **c)
    for b in range(3):
        cb = {k: (v[b:b + 1] if v is not None else None) for k, v in c.items()}
        one = m(x[b:b + 1], r[b:b + 1], **cb)
        assert torch.allclose(full[b:b + 1], one, atol=1e-5)


def test_x_cat_and_errors(tiny):
    m, _ = tiny
    c = to_dev(cond_for(G.UNET_TINY, 1, 2, 0, 3), DEV)
    x = torch.zeros(1, 8, 16, dtype=torch.long, device=DEV)
    out = m(x, torch.zeros(1, device=DEV), **c, x_cat=x)  # token rows concatenated on dim 1 (src/modules.py:264-265)
    assert out.shape == (1, 64, 16, 16)
    with pytest.raises(RuntimeError, match="multiple of"):
        m(torch.zeros(1, 12, 12, dtype=torch.long, device=DEV), torch.zeros(1, device=DEV), **c)
    with pytest.raises(TypeError):
        m(x, torch.zeros(1, device=DEV), **c, bogus=1)


def test_state_dict_roundtrip_reloads_engine(tiny):
    m, sd = tiny
    c = to_dev(cond_for(G.UNET_TINY, 1, 2, 0, 3), DEV)
    x = torch.zeros(1, 16, 16, dtype=torch.long, device=DEV)
    r = torch.full((1,), 0.5, device=DEV)
    a = m(x, r, **c).clone()
    sd2 = {k: v * 1.01 for k, v in m.state_dict().items()}
    m.load_state_dict(sd2)
    b = m(x, r, **c).clone()
    assert not torch.equal(a, b)  # the native copy was refreshed
    m.load_state_dict({k: v.to(DEV) for k, v in sd.items()})
    assert torch.equal(a, m(x, r, **c))


def test_570m_forward_vs_oracle(built_lib):
    """BASELINE config 2 shape: 570M-class stand-in (blocks=[4,8,4]), 32x32 tokens, CLIP-text only, B=1."""
    cfg = G.UNET_570M
    m = paella_amd.Paella(**cfg)
    sd = weights_for(m, sum(cfg["blocks"]))
    m = m.to(DEV)
    g = torch.Generator().manual_seed(5)
    x = torch.randint(0, 8192, (1, 32, 32), generator=g)
    r = torch.tensor([0.625])
    c = cond_for(cfg, 1, 0, 0, G.COND_SEED)
    with torch.no_grad():
        ref = O.unet_forward(sd, cfg, x, r, **c)
    got = m(x.to(DEV), r.to(DEV), **to_dev(c, DEV)).float().cpu()
    diff = (got - ref).abs().max().item()
    std = ref.std().item()
    clear, near, n_near = argmax_report(ref, got)
    print("570M forward: logit std %.3f, max|diff| %.3e, argmax mismatches clear=%d near-tie=%d (of %d near-tie positions / 1024)"
          % (std, diff, clear, near, n_near))
    assert std > 0.05, "degenerate logits"
    assert diff <= 1e-3 * max(1.0, std)
    assert clear == 0


def test_1b_forward_ragged_conditioning_vs_oracle(built_lib):
    """Released-size model (default ctor, 1.007B params), ByT5 + CLIP text + CLIP image conditioning (S = 16), B = 2."""
    cfg = G.UNET_1B
    m = paella_amd.Paella(**cfg)
    assert sum(p.numel() for p in m.parameters()) == 1007302016
    sd = weights_for(m, sum(cfg["blocks"]))
    m = m.to(DEV)
    g = torch.Generator().manual_seed(6)
    x = torch.randint(0, 8192, (2, 16, 16), generator=g)
    r = torch.tensor([0.9, 0.1])
    c = cond_for(cfg, 2, 8, 1, G.COND_SEED)
    with torch.no_grad():
        ref = O.unet_forward(sd, cfg, x, r, **c)
    got = m(x.to(DEV), r.to(DEV), **to_dev(c, DEV)).float().cpu()
    diff = (got - ref).abs().max().item()
    clear, near, n_near = argmax_report(ref, got)
    print("1B forward: logit std %.3f, max|diff| %.3e, argmax mismatches clear=%d near-tie=%d" % (ref.std().item(), diff, clear, near))
    assert diff <= 1e-3 * max(1.0, ref.std().item()) and clear == 0


def test_570m_64x64_grid_vs_oracle(built_lib):
    """BASELINE config 3 geometry (512 px = 64x64 tokens) at batch 1: level-1 attention has 256 queries (query-parallel kernel path)."""
    cfg = G.UNET_570M
    m = paella_amd.Paella(**cfg)
    sd = weights_for(m, sum(cfg["blocks"]))
    m = m.to(DEV)
    g = torch.Generator().manual_seed(8)
    x = torch.randint(0, 8192, (1, 64, 64), generator=g)
    r = torch.tensor([0.35])
    c = cond_for(cfg, 1, 3, 0, G.COND_SEED)
    with torch.no_grad():
        ref = O.unet_forward(sd, cfg, x, r, **c)
    got = m(x.to(DEV), r.to(DEV), **to_dev(c, DEV)).float().cpu()
    diff = (got - ref).abs().max().item()
    clear, near, n_near = argmax_report(ref, got)
    print("570M 64x64: logit std %.3f, max|diff| %.3e, argmax mismatches clear=%d near-tie=%d of %d" % (ref.std().item(), diff, clear, near, 64 * 64))
    assert diff <= 1e-3 * max(1.0, ref.std().item()) and clear == 0


def _real_geometry_vs_oracle(cfg, grid, S_byt5, n_img, seed, what):
    """One B = 1 forward at a BASELINE configuration's REAL per-sample geometry against the CPU oracle (src/modules.py:263-275,
    attention src/modules.py:7-19 at the real query / key counts): logits within 1e-3 * std, argmax identical except at reference
    near-ties (counted, printed)."""
    m = paella_amd.Paella(**cfg)
    sd = weights_for(m, sum(cfg["blocks"]))
    m = m.to(DEV)
    g = torch.Generator().manual_seed(seed)
    x = torch.randint(0, cfg["num_labels"], (1, grid, grid), generator=g)
    r = torch.tensor([0.55])
    c = cond_for(cfg, 1, S_byt5, n_img, G.COND_SEED + seed)
    with torch.no_grad():
        ref = O.unet_forward(sd, cfg, x, r, **c)
    got = m(x.to(DEV), r.to(DEV), **to_dev(c, DEV)).float().cpu()
    diff = (got - ref).abs().max().item()
    std = ref.std().item()
    clear, near, n_near = argmax_report(ref, got)
    print("%s: logit std %.3f, max|diff| %.3e, argmax mismatches clear=%d near-tie=%d (of %d near-tie positions / %d)"
          % (what, std, diff, clear, near, n_near, grid * grid))
    assert std > 0.05, "degenerate logits"
    assert diff <= 1e-3 * max(1.0, std)
    assert clear == 0
    del m
    torch.cuda.empty_cache()


def test_1b_config4_geometry_64x64_s776_vs_oracle(built_lib):
    """BASELINE configs[3] per-sample geometry: released-size 1B model, 512 px = 64x64 tokens, ByT5 at the tokenizer's max_length
    (768 rows, src/train.py:56) + CLIP text + CLIP image = 776 conditioning rows: level-1 attention 256 queries x 1032 keys,
    level-2 64 x 840."""
    _real_geometry_vs_oracle(G.UNET_1B, 64, 768, 1, 13, "1B 64x64 S=776 (configs[3] geometry)")


def test_1b_config5_geometry_128x128_s264_vs_oracle(built_lib):
    """BASELINE configs[4] per-sample geometry: 1B model, 1024 px = 128x128 tokens, S = 256 + 4 + 4 = 264: level-1 attention
    1024 queries x 1288 keys (attention_lds_kernel inside the real network), level-2 256 x 520; 2.2 TFLOP on the CPU oracle."""
    _real_geometry_vs_oracle(G.UNET_1B, 128, 256, 1, 14, "1B 128x128 S=264 (configs[4] geometry)")


def test_1b_largest_key_count_128x128_s776_vs_oracle(built_lib):
    """The corner VERDICT r05 named as untested: configs[4]'s grid (128x128 tokens) with configs[3]'s conditioning length (ByT5 768 + CLIP text + CLIP image =
    776 rows) -- the largest key count of any BASELINE geometry: level-1 attention 1024 queries x 1800 keys, level-2 256 x 1032."""
    _real_geometry_vs_oracle(G.UNET_1B, 128, 768, 1, 15, "1B 128x128 S=776 (largest key count)")


def test_large_grid_properties(built_lib):
    """BASELINE config 5 geometry (128x128 tokens) on a narrow model: finite, deterministic, batch rows independent."""
    cfg = dict(G.UNET_MID)
    m = paella_amd.Paella(**cfg)
    weights_for(m, sum(cfg["blocks"]))
    m = m.to(DEV)
    g = torch.Generator().manual_seed(2)
    x = torch.randint(0, cfg["num_labels"], (2, 128, 128), generator=g).to(DEV)
    r = torch.tensor([0.7, 0.2], device=DEV)
    c = to_dev(cond_for(cfg, 2, 40, 1, 5), DEV)
    a = m(x, r, **c)
    assert torch.isfinite(a).all() and a.shape == (2, cfg["num_labels"], 128, 128)
    assert torch.equal(a, m(x, r, **c))
    c0 = {k: (v[:1] if v is not None else None) for k, v in c.items()}
    assert torch.allclose(a[:1], m(x[:1], r[:1], **c0), atol=1e-5)


@pytest.mark.parametrize("cfg_name", ["UNET_TINY", "UNET_MID", "UNET_570M"])
def test_shared_cfg_prefix_matches_full_evaluation(built_lib, cfg_name):
    """Classifier-free guidance batches cond + uncond rows with identical tokens / r; computing the conditioning-free prefix
    once (n_unique = B/2) must give the logits of the plain 2B-row evaluation."""
    cfg = dict(getattr(G, cfg_name))
    m = paella_amd.Paella(**cfg)
    weights_for(m, sum(cfg["blocks"]))
    m = m.to(DEV)
    B = 2 if cfg_name != "UNET_570M" else 1
    g = torch.Generator().manual_seed(11)
    x = torch.randint(0, cfg["num_labels"], (B, 16, 16), generator=g).to(DEV)
    r = torch.rand(B, generator=g).to(DEV)
    c = to_dev(cond_for(cfg, B, 3, 0, 21), DEV)
    u = to_dev(cond_for(cfg, B, 3, 0, 22), DEV)
    both = {k: (torch.cat([c[k], u[k]]) if c[k] is not None else None) for k in c}
    cache = m.prepare_cond(**both)
    x2, r2 = torch.cat([x, x]), torch.cat([r, r])
    full = m.forward_prepared(x2, r2, cache).clone()
    shared = m.forward_prepared(x, r, cache)
    assert torch.isfinite(shared).all()
    scale = max(1.0, full.std().item())
    assert (full - shared).abs().max().item() <= 2e-5 * scale
    assert not torch.equal(shared[:B], shared[B:])  # the two halves really saw different conditioning
    if B > 1:
        with pytest.raises(ValueError):
            m.forward_prepared(torch.cat([x, x[:1]]), torch.cat([r, r[:1]]), cache)  # 3 token rows against 4 conditioning rows


def test_guidance_mix_through_linear_head(built_lib):
    """cfg_mix folds l_c*a + l_u*b (src/utils.py:47) through the bias-free head: equal to mixing the two logits tensors."""
    cfg = dict(G.UNET_MID)
    m = paella_amd.Paella(**cfg)
    weights_for(m, sum(cfg["blocks"]))
    m = m.to(DEV)
    B = 2
    g = torch.Generator().manual_seed(5)
    x = torch.randint(0, cfg["num_labels"], (B, 16, 16), generator=g).to(DEV)
    r = torch.rand(B, generator=g).to(DEV)
    c = to_dev(cond_for(cfg, B, 3, 0, 31), DEV)
    u = to_dev(cond_for(cfg, B, 3, 0, 32), DEV)
    both = {k: (torch.cat([c[k], u[k]]) if c[k] is not None else None) for k in c}
    cache = m.prepare_cond(**both)
    x2, r2 = torch.cat([x, x]), torch.cat([r, r])
    full = m.forward_prepared(x2, r2, cache).clone()
    a, b = 8.0, -7.0
    ref = full[:B] * a + full[B:] * b
    mixed = m.forward_prepared(x, r, cache, cfg_mix=(a, b))
    assert mixed.shape == ref.shape
    assert (mixed - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    with pytest.raises(ValueError):
        m.forward_prepared(x2, r2, cache, cfg_mix=(a, b))  # needs the distinct rows only


@pytest.mark.parametrize("B,grid,regime", [(1, 16, "batch-1 regime: row statistics derived inside the GEMM"), (2, 128, ">= 2048 rows: row statistics finished by the pre-pass")])
@pytest.mark.parametrize("shift", [100.0, 1000.0])
def test_layernorm_guard_inside_the_network(built_lib, B, grid, regime, shift):
    """The LayerNorm folded into the consuming GEMM (reference src/modules.py:22-27 ahead of the attention in-projection, the up-sampler and clf) has an
    operand-side guard for rows with |mean| >> std.  Ordinary activations never trip it (|mean| / std <= 0.07), so here the residual stream is PUSHED there:
    every TimestepBlock (the last op before each LayerNorm consumer, `x * (1 + a) + b`, src/modules.py:99-106) gets a nearly constant `b` = shift plus three
    outlier channels (+60, -45, +80) and a small `a`: the rows the consumers normalise have |mean| / std ~ 16 (shift 100) and ~ 160 (shift 1000) at all three levels.
    The whole network -- producer epilogue's centred partials -> pre-pass / in-kernel derivation -> guarded consumer -- must still match the oracle:
    logits within 1e-3 * std, no argmax mismatch with a clear reference margin, and the guard must really have run (device counter hook)."""
    lib = built_lib
    cfg = G.UNET_MID
    m = paella_amd.Paella(**cfg)
    sd = weights_for(m, sum(cfg["blocks"]))
    for k in list(sd):
        if k.endswith(".mapper.weight") and sd[k].dim() == 2 and sd[k].shape[1] == cfg["c_r"]:  # TimestepBlock mapper [2c, c_r] -> (a | b)
            c = sd[k].shape[0] // 2
            sd[k] = sd[k] * 0.05
            b = sd[k[:-6] + "bias"].clone()
            b[:c] *= 0.1
            b[c:] += shift
            b[c + 3] += 60.0
            b[c + c // 2 + 1] -= 45.0
            b[2 * c - 5] += 80.0
            sd[k[:-6] + "bias"] = b
    m.load_state_dict(sd)
    m = m.to(DEV)
    g = torch.Generator().manual_seed(17)
    x = torch.randint(0, cfg["num_labels"], (B, grid, grid), generator=g)
    r = torch.rand(B, generator=g)
    c = cond_for(cfg, B, 3, 0, G.COND_SEED + 3)
    with torch.no_grad():
        ref = O.unet_forward(sd, cfg, x, r, **c)
    counter = torch.zeros(1, dtype=torch.int32, device=DEV)
    lib.paella_test_ln_guard_counter(ctypes.c_void_p(counter.data_ptr()))
    try:
        got = m(x.to(DEV), r.to(DEV), **to_dev(c, DEV)).float().cpu()
        torch.cuda.synchronize()
    finally:
        lib.paella_test_ln_guard_counter(None)
    n_guard = int(counter.item())
    diff, std = (got - ref).abs().max().item(), ref.std().item()
    clear, near, n_near = argmax_report(ref, got)
    print("LayerNorm guard in the network (%s, TimestepBlock shift %g): %d waves took the operand-side path; logit std %.3f, max|diff| %.3e, argmax mismatches clear=%d near-tie=%d (of %d)"
          % (regime, shift, n_guard, std, diff, clear, near, n_near))
    assert n_guard > 0, "the guard never tripped: the test does not exercise the operand-side path"
    assert std > 0.05 and diff <= 1e-3 * max(1.0, std) and clear == 0
