"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np
import torch

from oracle import golden_configs as G
from paella_amd import synth


def weights_for(module, cfg_blocks, golden_npz=None, seed=G.WEIGHT_SEED):
    """Seeded synthetic weights for `module`; verifies key/shape compatibility and the checksum stored with the fixture."""
    sd = synth.synth_state_dict(module.state_dict(), seed=seed, n_blocks=cfg_blocks)
    if golden_npz is not None:
        keys = sorted(k + ":" + ",".join(str(d) for d in v.shape) for k, v in module.state_dict().items())
        assert keys == sorted(golden_npz["keys"].tolist()), "state-dict keys/shapes differ from the reference's"
        c = synth.checksum(sd)
        np.testing.assert_allclose(np.array(c), golden_npz["checksum"], rtol=1e-9, err_msg="synthetic weight generator drifted")
    module.load_state_dict(sd)
    return sd


def cond_for(cfg, B, S_byt5, n_img, seed, device="cpu"):
    return synth.synth_conditioning(B, S_byt5, cfg["byt5_embd"], cfg["clip_embd"], seed=seed, with_clip=True, n_clip_image=n_img,
                                    device=device)


def to_dev(inputs, device):
    out = {}
    for k, v in inputs.items():
        if v is None:
            out[k] = None
        elif isinstance(v, (list, tuple)):
            out[k] = [t.to(device) for t in v]
        else:
            out[k] = v.to(device)
    return out


def argmax_report(ref_logits, got_logits, eps=1e-4):
    """Near-tie policy: positions whose reference top1-top2 margin is below eps are counted
    separately, never silently dropped.  logits [B, L, H, W].  Returns (mismatches_clear, mismatches_near_tie, n_near_tie)."""
    top2 = ref_logits.topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    mism = ref_logits.argmax(1) != got_logits.argmax(1)
    near = margin < eps
    return int((mism & ~near).sum()), int((mism & near).sum()), int(near.sum())


# ---------------------------------------------------------------------------------------------------------------------
# Token parity to the contract: integer outputs are compared EXACTLY; a differing token is tolerated only where the
# reference's own decision was a near-tie (top-1 / top-2 of the quantity it takes the argmax of within `eps`), and every
# such position is counted and printed -- never a blanket agreement fraction.
# ---------------------------------------------------------------------------------------------------------------------
def oracle_step(fwd, model_inputs, unconditional_inputs, tokens, t, cfg_pair, temperature, noise_q, argmax):
    """One reference step (src/utils.py:43-50) on the oracle: returns (drawn tokens [B,H,W], relative top1-top2 margin [B,H,W])
    of the score the categorical draw maximises: softmax(l/T)/q (multinomial == argmax(p/q)), or the mixed logits at T = 0."""
    from oracle import paella_oracle as O  # noqa: F401  (test infrastructure)
    B = tokens.size(0)
    r = torch.ones(B) * t
    lc = fwd(tokens, r, **model_inputs)
    lu = fwd(tokens, r, **unconditional_inputs) if cfg_pair is not None else None
    l = lc if lu is None else lc * cfg_pair[0] + lu * cfg_pair[1]
    Bq, L, H, W = l.shape
    if argmax:
        score = l.permute(0, 2, 3, 1).reshape(-1, L)
        top = score.topk(2, dim=1).values
        margin = (top[:, 0] - top[:, 1]) / top[:, 0].abs().clamp_min(1.0)
    else:
        score = l.div(temperature).softmax(dim=1).permute(0, 2, 3, 1).reshape(-1, L) / noise_q
        top = score.topk(2, dim=1).values
        margin = (top[:, 0] - top[:, 1]) / top[:, 0]
    return score.argmax(dim=1).view(Bq, H, W), margin.view(Bq, H, W)


def stepwise_token_parity(model, fwd, num_labels, cond, uncond, cond_dev, uncond_dev, noise, steps, renoise_steps, temperature, cfg,
                          t_start=1.0, t_end=0.0, argmax=False, eps=1e-3, device="cuda"):
    """Teacher-forced per-step parity of the HIP sampler against the oracle: every step starts from the ORACLE's tokens, both
    sides use the same explicit noise, and each differing token is classified by the oracle's decision margin.
    Returns dict(clear=.., near_tie=.., positions=.., per_step=[..]); `clear` must be 0."""
    import paella_amd
    from oracle import paella_oracle as O
    t_list = [float(v) for v in torch.linspace(t_start, t_end, steps + 1)]
    temps = [float(v) for v in torch.linspace(temperature[0], temperature[1], steps)]
    pair = None if not cfg else (float(torch.tensor(float(cfg), dtype=torch.float32)), float(torch.tensor(1.0 - float(cfg), dtype=torch.float32)))
    init_noise = noise["init_noise"]
    tokens = init_noise.clone()
    rep = {"clear": 0, "near_tie": 0, "positions": 0, "per_step": []}
    for i in range(steps):
        drawn, margin = oracle_step(fwd, cond, uncond, tokens, t_list[i], pair, 1.0 if argmax else temps[i], None if argmax else noise["q"][i], argmax)
        renoise = i < renoise_steps
        ref_next = drawn
        if renoise:
            ref_next, _ = O.add_noise(drawn, torch.ones(tokens.size(0)) * t_list[i + 1], num_labels, random_x=init_noise, rand_u=noise["u"][i])
        # the same single step through the public HIP API: a 1-step schedule from t_i to t_{i+1}
        step_noise = {"init_noise": init_noise, "q": [None if argmax else noise["q"][i]], "u": [noise["u"][i] if renoise else None]}
        got = paella_amd.sample_distributed(model, cond_dev, uncond_dev, tuple(tokens.shape), init_x=tokens.to(device), steps=1,
                                            renoise_steps=1 if renoise else 0, temperature=((0.0, 0.0) if argmax else (temps[i], temps[i])),
                                            cfg=(None if not cfg else (float(cfg), float(cfg))), t_start=t_list[i], t_end=t_list[i + 1],
                                            noise=step_noise).cpu()
        mism = got != ref_next
        near = margin < eps
        rep["clear"] += int((mism & ~near).sum())
        rep["near_tie"] += int((mism & near).sum())
        rep["positions"] += mism.numel()
        rep["per_step"].append((int(mism.sum()), int(near.sum())))
        tokens = ref_next
    return rep


def assert_token_parity(rep, what):
    print("%s: %d positions x steps, %d mismatches at reference near-ties (margin < eps), %d clear mismatches; per step (mismatches, near-ties present): %s"
          % (what, rep["positions"], rep["near_tie"], rep["clear"], rep["per_step"]))
    assert rep["clear"] == 0, "%s: %d token(s) differ where the reference decision was NOT a near-tie" % (what, rep["clear"])
