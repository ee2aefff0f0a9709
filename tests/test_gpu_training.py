"""GPU: the training-mode path ON THE DEVICE and its hand-over to the HIP inference engine.

The reference's loops (src/train.py:59-66, src_distributed/train.py:104-114) run `model.train(); pred = model(...); loss.backward();
optimizer.step()` on cuda tensors and later sample from the SAME module in eval mode.  Checked here, on `cuda`:
  * one train-mode step reproduces the REFERENCE's loss / logits / gradients (tests/golden/train_tiny_step.npz, generated from
    /root/reference by oracle/make_golden.py) -- the CPU test's assertions, on the device;
  * after `optimizer.step()` and `model.eval()` the hand-written HIP engine serves the UPDATED weights: its logits equal the CPU
    oracle evaluated on the module's new state dict (and differ from the pre-step logits);
  * sampling straight after training works (tokens in range) -- the train -> sample round trip of the reference's loops."""
import numpy as np
import pytest
import torch
from torch import nn

import paella_amd
from oracle import golden_configs as G
from oracle import paella_oracle as O
from tests.helpers import argmax_report, cond_for, to_dev, weights_for

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _train_step(m, cfg, p_drop=0.0):
    latents, t, mask, random_x, c = G.train_step_inputs(cfg)
    latents, t, mask, random_x = latents.to(DEV), t.to(DEV), mask.to(DEV), random_x.to(DEV)
    c = to_dev(c, DEV)
    m.train()
    m.dropout = p_drop
    m.zero_grad(set_to_none=True)
    noised = latents * (1 - mask) + random_x * mask   # Paella.add_noise with explicit mask / random_x (src/modules.py:277-283)
    lw = m.get_loss_weight(t, mask)
    pred = m(noised, t, **c)
    assert pred.is_cuda and pred.requires_grad
    loss = nn.CrossEntropyLoss(label_smoothing=0.1, reduction='none')(pred, latents)
    loss = ((loss * lw).sum(dim=[1, 2]) / lw.sum(dim=[1, 2])).mean()
    loss.backward()
    return pred.detach(), loss.detach()


@pytest.mark.parametrize("which", ["tiny", "variant"])
def test_train_step_on_device_then_hip_engine_serves_updated_weights(golden, built_lib, which):
    cfg = G.UNET_TINY if which == "tiny" else G.UNET_VARIANT
    g = golden("train_%s_step" % which)
    m = paella_amd.Paella(**cfg)
    weights_for(m, sum(cfg["blocks"]), golden("unet_%s_forward" % which))
    m = m.to(DEV)
    # eval-mode logits before the step (HIP engine)
    gen = torch.Generator().manual_seed(17)
    x = torch.randint(0, cfg["num_labels"], (2, 16, 16), generator=gen)
    r = torch.tensor([0.8, 0.3])
    c = cond_for(cfg, 2, 3, 1, G.COND_SEED + 3)
    before = m(x.to(DEV), r.to(DEV), **to_dev(c, DEV)).float().cpu().clone()

    # ---- one training step on the device vs the reference's step ----
    pred, loss = _train_step(m, cfg)
    np.testing.assert_allclose(float(loss), float(g["nodrop_loss"]), rtol=2e-5)
    np.testing.assert_allclose(pred[:, ::4, ::2, ::2].cpu().numpy(), g["nodrop_pred_sub"], atol=5e-5, rtol=1e-4)
    params = dict(m.named_parameters())
    names = g["names"].tolist()
    norms = np.array([float(params[k].grad.norm()) for k in names])
    np.testing.assert_allclose(norms, g["nodrop_grad_norms"], rtol=5e-4, atol=1e-6)
    for k in [k for k in g.files if k.startswith("nodrop_grad:")]:
        ref = g[k]
        np.testing.assert_allclose(params[k.split(":", 1)[1]].grad.cpu().numpy(), ref, rtol=5e-4, atol=5e-4 * max(float(np.abs(ref).max()), 1e-6), err_msg=k)

    # ---- optimizer step, back to eval: the HIP engine must pick the new weights up ----
    opt = torch.optim.AdamW(m.parameters(), lr=2e-3)
    nn.utils.clip_grad_norm_(m.parameters(), 1.0)
    opt.step()
    m.eval()
    after = m(x.to(DEV), r.to(DEV), **to_dev(c, DEV)).float().cpu()
    sd_new = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        ref = O.unet_forward(sd_new, cfg, x, r, **c)
    diff = (after - ref).abs().max().item()
    moved = (after - before).abs().max().item()
    clear, near, n_near = argmax_report(ref, after)
    print("%s: train step on device ok (loss %.5f); after optimizer.step + eval: HIP vs oracle(updated state dict) max|diff| %.2e, logits moved by %.2e, "
          "argmax mismatches clear=%d near-tie=%d" % (which, float(loss), diff, moved, clear, near))
    assert diff <= 2e-4 * max(1.0, ref.std().item())
    assert moved > 50 * diff, "the engine still serves the pre-step weights"
    assert clear == 0

    # ---- and the train -> sample round trip ----
    u = cond_for(cfg, 2, 3, 1, G.COND_SEED + 4)
    toks = paella_amd.sample(m, to_dev(c, DEV), (2, 16, 16), unconditional_inputs=to_dev(u, DEV), steps=2, renoise_steps=1, device=DEV, noise="philox", seed=3)
    assert toks.shape == (2, 16, 16) and int(toks.min()) >= 0 and int(toks.max()) < cfg["num_labels"]


def test_eval_mode_forward_refuses_autograd(built_lib):
    """A freshly constructed paella_amd.Paella is in eval mode: asking the HIP engine for gradients must fail with a
    message that names the fix, not with a bare 'does not require grad' from loss.backward()."""
    m = paella_amd.Paella(**G.UNET_TINY)
    weights_for(m, sum(G.UNET_TINY["blocks"]))
    m = m.to(DEV)
    c = to_dev(cond_for(G.UNET_TINY, 1, 2, 0, 3), DEV)
    x = torch.zeros(1, 16, 16, dtype=torch.long, device=DEV)
    with torch.enable_grad():
        out = m(x, torch.zeros(1, device=DEV), **c)
        assert out.requires_grad  # like the reference module's eval-mode output
        with pytest.raises(RuntimeError, match=r"model\.train\(\)"):
            out.sum().backward()
        out2 = m(x, torch.zeros(1, device=DEV), **c)
        ref = out2.detach().clone()
        out2.mul_(2.0)  # in-place edits of the logits work as on the reference's output: the result is not a view created inside a custom Function
        assert torch.equal(out2.detach(), ref * 2.0)
    with torch.no_grad():
        assert not m(x, torch.zeros(1, device=DEV), **c).requires_grad
