"""GPU: captured-graph hygiene.

  * paella_select_tokens (the only integer elementwise work of the eval path that used to run through ATen) against torch;
  * GraphSampler refuses / recaptures a stale capture: after load_state_dict, an optimizer-style in-place update or set_gemm_precision the replay either equals
    a fresh eager call bit for bit or raises a RuntimeError naming the cause (the call site that reloads weights between sampling calls: src/train.py:40,64-69,76);
  * GraphInpainter (encode -> masked renoise -> sample(init_x) -> re-impose -> decode in ONE graph) == the eager `inpaint(noise="philox")`, sharded too;
  * a NaN-poisoned conditioning (what a receiver of a mismatched broadcast samples on, paella_amd/dist.py) still yields tokens inside [0, num_labels)."""
import pytest
import torch

import paella_amd
from oracle import golden_configs as G
from paella_amd import synth
from tests.helpers import cond_for, to_dev, weights_for

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _tiny():
    m = paella_amd.Paella(**G.UNET_TINY)
    sd = weights_for(m, sum(G.UNET_TINY["blocks"]))
    return m.to(DEV), sd


def _tiny_vq():
    vc = dict(G.VQ_TINY_F8, codebook_size=G.UNET_TINY["num_labels"])
    vq = paella_amd.VQModel(**vc)
    weights_for(vq, vc["bottleneck_blocks"])
    return vq.to(DEV)


def test_select_tokens_kernel(built_lib):
    g = torch.Generator().manual_seed(1)
    a = torch.randint(0, 8192, (3, 17, 23), generator=g).to(DEV)
    b = torch.randint(0, 8192, (3, 17, 23), generator=g).to(DEV)
    mask = torch.randint(0, 2, (3, 17, 23), generator=g).to(DEV)
    one, zero = torch.ones(1, device=DEV), torch.zeros(1, device=DEV)
    assert torch.equal(paella_amd.select_tokens(a, b, mask), a * mask + b * (1 - mask))
    assert torch.equal(paella_amd.select_tokens(a, b, mask, flag=one), a * mask + b * (1 - mask))
    assert torch.equal(paella_amd.select_tokens(a, b, mask, flag=zero), b)
    assert torch.equal(paella_amd.select_tokens(a, flag=one, fill=-1), a)
    assert torch.equal(paella_amd.select_tokens(a, flag=zero, fill=-1), torch.full_like(a, -1))
    assert torch.equal(paella_amd.select_tokens(a, mask=mask, fill=-7), torch.where(mask != 0, a, torch.full_like(a, -7)))
    big = torch.randint(0, 8192, (5, 512, 512), generator=g).to(DEV)  # more elements than one grid-stride pass of 4096 x 256 threads
    mb = (big % 3 == 0).long()
    assert torch.equal(paella_amd.select_tokens(big, big + 1, mb), torch.where(mb != 0, big, big + 1))
    with pytest.raises(ValueError):
        paella_amd.select_tokens(a, b[:2])


def test_graph_sampler_recaptures_after_weight_updates(built_lib):
    m, sd = _tiny()
    vq = _tiny_vq()
    cfg = G.UNET_TINY
    shape = (2, 16, 16)
    cs, us = to_dev(cond_for(cfg, 2, 3, 0, 1), DEV), to_dev(cond_for(cfg, 2, 3, 0, 2), DEV)
    kw = dict(steps=4, renoise_steps=3, device=DEV)
    eager = lambda: (lambda t: (t, vq.decode_indices(t)))(paella_amd.sample(m, cs, shape, unconditional_inputs=us, noise="philox", seed=9, **kw))
    gs = paella_amd.GraphSampler(m, cs, us, shape, vqgan=vq, **kw)
    t0, i0 = eager()
    a = gs(cs, us, seed=9)
    assert torch.equal(a[0], t0) and torch.equal(a[1], i0) and gs.captures == 1
    assert torch.equal(gs(cs, us, seed=9)[0], t0) and gs.captures == 1  # a fresh graph is replayed, not recaptured
    # 1. load_state_dict (src/train.py:40)
    m.load_state_dict({k: v * 1.05 for k, v in sd.items()})
    t1, i1 = eager()
    assert not torch.equal(t1, t0)
    a = gs(cs, us, seed=9)
    assert torch.equal(a[0], t1) and torch.equal(a[1], i1) and gs.captures == 2
    # 2. an optimizer-style in-place step on ONE parameter (src/train.py:64-69)
    with torch.no_grad():
        m.clf._modules["1"].weight.add_(0.01)
    t2, i2 = eager()
    a = gs(cs, us, seed=9)
    assert torch.equal(a[0], t2) and torch.equal(a[1], i2) and gs.captures == 3
    # 3. the VQGAN's weights (its ResBlock gammas are HOST constants baked into the captured launches)
    with torch.no_grad():
        for p in vq.parameters():
            if p.dim() == 1 and p.numel() == 6:
                p.mul_(1.5)
    t3, i3 = eager()
    assert torch.equal(t3, t2) and not torch.equal(i3, i2)
    a = gs(cs, us, seed=9)
    assert torch.equal(a[0], t3) and torch.equal(a[1], i3) and gs.captures == 4
    # 4. the precision mode (another kernel family, a bigger workspace): recaptured in the new mode, then back, bit for bit
    m.set_gemm_precision("bf16")
    tb = paella_amd.sample(m, cs, shape, unconditional_inputs=us, noise="philox", seed=9, **kw)
    assert torch.equal(gs(cs, us, seed=9)[0], tb) and gs.captures == 5
    m.set_gemm_precision("fp32")
    a = gs(cs, us, seed=9)
    assert torch.equal(a[0], t3) and torch.equal(a[1], i3) and gs.captures == 6


def test_graph_sampler_raises_when_asked_to(built_lib):
    m, sd = _tiny()
    cfg = G.UNET_TINY
    shape = (1, 16, 16)
    cs, us = to_dev(cond_for(cfg, 1, 3, 0, 1), DEV), to_dev(cond_for(cfg, 1, 3, 0, 2), DEV)
    gs = paella_amd.GraphSampler(m, cs, us, shape, steps=2, renoise_steps=1, device=DEV, on_stale="raise")
    gs(cs, us, seed=1)
    m.load_state_dict({k: v * 0.99 for k, v in sd.items()})
    with pytest.raises(RuntimeError, match="denoiser weights"):
        gs(cs, us, seed=1)
    m.load_state_dict(sd)  # same values, but new versions: still stale (the check is on identity + version, never on contents)
    with pytest.raises(RuntimeError, match="stale"):
        gs(cs, us, seed=1)
    gs2 = paella_amd.GraphSampler(m, cs, us, shape, steps=2, renoise_steps=1, device=DEV, on_stale="raise")
    m.set_gemm_precision("bf16")
    with pytest.raises(RuntimeError, match="precision"):
        gs2(cs, us, seed=1)
    m.set_gemm_precision("fp32")
    gs2(cs, us, seed=1)  # back in the captured mode with untouched weights: fresh again


@pytest.mark.parametrize("keep_known", [True, False])
def test_graph_inpainter_equals_eager_inpaint(built_lib, keep_known):
    m, _ = _tiny()
    vq = _tiny_vq()
    cfg = G.UNET_TINY
    g = torch.Generator().manual_seed(4)
    B, H, W = 3, 8, 16
    img = torch.rand(B, 3, H * 8, W * 8, generator=g).to(DEV)
    mask = torch.zeros(B, H, W, dtype=torch.int64)
    mask[:, 2:6, 3:13] = 1
    cs, us = to_dev(cond_for(cfg, B, 3, 0, 1), DEV), to_dev(cond_for(cfg, B, 3, 0, 2), DEV)
    toks, out = paella_amd.inpaint(m, vq, img, mask, cs, us, steps=4, t_start=0.6, noise="philox", seed=21, keep_known=keep_known)
    gi = paella_amd.GraphInpainter(m, vq, img, mask, cs, us, steps=4, t_start=0.6, keep_known=keep_known, device=DEV)
    gt, go = gi(img, mask, cs, us, seed=21)
    assert torch.equal(gt, toks) and torch.equal(go, out)
    orig = vq.encode(img)[2]
    mk = mask.to(DEV).bool()
    if keep_known:
        assert torch.equal(toks[~mk], orig[~mk])
    assert (toks[mk] != orig[mk]).float().mean() > 0.2
    # other inputs through the same graph: a new image, mask, seed
    img2 = torch.rand(B, 3, H * 8, W * 8, generator=g).to(DEV)
    mask2 = torch.zeros_like(mask)
    mask2[:, :, :5] = 1
    t2, o2 = paella_amd.inpaint(m, vq, img2, mask2, cs, us, steps=4, t_start=0.6, noise="philox", seed=22, keep_known=keep_known)
    gt, go = gi(img2, mask2, cs, us, seed=22)
    assert torch.equal(gt, t2) and torch.equal(go, o2)
    # rows [1, 3) as a batch shard of the same request: the graph for 2 rows with a row offset == those rows of the unsharded call
    sl = lambda d: {k: (v[1:] if v is not None else None) for k, v in d.items()}
    gi2 = paella_amd.GraphInpainter(m, vq, img2[1:], mask2[1:], sl(cs), sl(us), steps=4, t_start=0.6, keep_known=keep_known, device=DEV)
    gt, go = gi2(img2[1:], mask2[1:], sl(cs), sl(us), seed=22, shard=(1, B))
    assert torch.equal(gt, t2[1:])
    pt, _ = paella_amd.inpaint(m, vq, img2[1:], mask2[1:], sl(cs), sl(us), steps=4, t_start=0.6, noise="philox", seed=22, keep_known=keep_known, shard=(1, B))
    assert torch.equal(pt, t2[1:])


def test_nan_conditioning_keeps_tokens_in_range(built_lib):
    """dist.sample_sharded: a receiver of a poisoned (NaN) conditioning buffer still runs its sampling pass; the tokens it feeds back into the next step's
    embedding gather must stay inside [0, num_labels) (they are replaced by -1 afterwards, on the device)."""
    m, _ = _tiny()
    cfg = G.UNET_TINY
    cs, us = to_dev(cond_for(cfg, 2, 3, 0, 1), DEV), to_dev(cond_for(cfg, 2, 3, 0, 2), DEV)
    nan = lambda d: {k: (None if v is None else torch.full_like(v, float("nan"))) for k, v in d.items()}
    for fused in (True, False):
        t = paella_amd.sample(m, nan(cs), (2, 16, 16), unconditional_inputs=nan(us), steps=3, renoise_steps=2, device=DEV, noise="philox", seed=3, fused_tail=fused)
        assert int(t.min()) >= 0 and int(t.max()) < cfg["num_labels"]
    flag = torch.zeros(1, device=DEV)
    assert bool((paella_amd.select_tokens(t, flag=flag, fill=-1) == -1).all())
