"""CPU: training-mode forward / backward of paella_amd.Paella against ONE TRAINING STEP OF THE REFERENCE
(tests/golden/train_tiny_step.npz, produced by oracle/make_golden.py from /root/reference/src_distributed: add_noise ->
get_loss_weight -> model(...) in train mode -> label-smoothed cross entropy weighted by loss_weight -> backward).
Checked: loss, logits, and the gradient of EVERY parameter (norm and sum to 1e-4, a set of tensors elementwise) -- without dropout
and with the reference's dropout 0.1 under the same torch seed (same mask stream)."""
import numpy as np
import pytest
import torch
from torch import nn

import paella_amd
from oracle import golden_configs as G
from tests.helpers import weights_for


@pytest.fixture(scope="module")
def model(golden):
    m = paella_amd.Paella(**G.UNET_TINY)
    weights_for(m, sum(G.UNET_TINY["blocks"]), golden("unet_tiny_forward"))
    return m


@pytest.fixture(scope="module")
def variant_model(golden):
    """FeedForwardBlock ('F'), cross-attention only (self_attn=False), patch_size 1, two levels (level_config ['CFT', 'TAC'])."""
    m = paella_amd.Paella(**G.UNET_VARIANT)
    weights_for(m, sum(G.UNET_VARIANT["blocks"]), golden("unet_variant_forward"))
    return m


def _step(m, p_drop, seed, cfg=G.UNET_TINY):
    latents, t, mask, random_x, c = G.train_step_inputs(cfg)
    m.train()
    m.dropout = p_drop
    m.zero_grad(set_to_none=True)
    noised = latents * (1 - mask) + random_x * mask          # Paella.add_noise with explicit mask / random_x (src/modules.py:277-283)
    lw = m.get_loss_weight(t, mask)
    if seed is not None:
        torch.manual_seed(seed)
    pred = m(noised, t, **c)
    assert pred.requires_grad and pred.grad_fn is not None
    loss = nn.CrossEntropyLoss(label_smoothing=0.1, reduction='none')(pred, latents)
    loss = ((loss * lw).sum(dim=[1, 2]) / lw.sum(dim=[1, 2])).mean()
    loss.backward()
    return pred.detach(), loss.detach()


@pytest.mark.parametrize("tag,p_drop,seed", [("nodrop", 0.0, None), ("drop", 0.1, 1234)])
@pytest.mark.parametrize("which", ["tiny", "variant"])
def test_training_step_matches_reference(golden, model, variant_model, which, tag, p_drop, seed):
    g = golden("train_%s_step" % which)
    model, cfg = (model, G.UNET_TINY) if which == "tiny" else (variant_model, G.UNET_VARIANT)
    pred, loss = _step(model, p_drop, seed, cfg)
    np.testing.assert_allclose(float(loss), float(g[tag + "_loss"]), rtol=1e-5)
    np.testing.assert_allclose(pred[:, ::4, ::2, ::2].numpy(), g[tag + "_pred_sub"], atol=2e-5, rtol=1e-5)
    params = dict(model.named_parameters())
    names = g["names"].tolist()
    assert names == [k for k, _ in model.named_parameters()], "parameter order / names differ from the reference module"
    norms = np.array([float(params[k].grad.norm()) for k in names])
    sums = np.array([float(params[k].grad.double().sum()) for k in names])
    np.testing.assert_allclose(norms, g[tag + "_grad_norms"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(sums, g[tag + "_grad_sums"], rtol=1e-3, atol=1e-4 * float(np.abs(g[tag + "_grad_norms"]).max()))
    full = [k for k in g.files if k.startswith(tag + "_grad:")]
    assert len(full) >= 8
    for k in full:
        ref = g[k]
        got = params[k.split(":", 1)[1]].grad.numpy()
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4 * max(float(np.abs(ref).max()), 1e-6), err_msg=k)
    model.eval()


def test_mode_switch_and_optimizer_step(model):
    """train() / eval() select the path; an optimizer step changes what the next train-mode forward returns; eval mode still
    refuses to run off a HIP device (no CPU fallback for the inference engine)."""
    assert paella_amd.Paella(**G.UNET_TINY).training is False  # constructed in eval mode
    _, l0 = _step(model, 0.0, None)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    opt.step()
    _, l1 = _step(model, 0.0, None)
    assert float(l1) < float(l0)
    model.eval()
    with pytest.raises(RuntimeError, match="HIP device"):
        model(torch.zeros(1, 8, 8, dtype=torch.long), torch.zeros(1), torch.zeros(1, 2, 40))
