"""Generate tests/golden/*.npz by running the REFERENCE itself (imported from /root/reference) in this container.

Run from the repo root:  python oracle/make_golden.py
The reference tree does not exist on the GPU box, so tests never import it; they regenerate the same seeded
weights / inputs (paella_amd.synth, CPU generator) and compare against the arrays stored here.  Every fixture
stores the weight checksum so that any drift of the generator is detected instead of silently mis-compared.

What is run:
  * Paella.forward            /root/reference/src/modules.py        (tiny, mid, variant configs)
  * attn_weights + list clip_image   /root/reference/utils/modules.py + utils/alter_attention.py
  * sample()                  /root/reference/src/utils.py:35 (imported with torchvision/torchtools stubs) and
                              /root/reference/src_distributed/utils.py:97 (function source exec'd: its module-level
                              imports need packages that are not installed)
  * Paella.add_noise          /root/reference/src/modules.py:277
  * VQModel.encode/decode/decode_indices  /root/reference/src/vqgan.py with torchtools.nn.VectorQuantize replaced by the
                              stand-in below (third-party, not in the snapshot: parity unpinned for that piece)
"""
import importlib.util
import os
import sys
import types
from unittest.mock import MagicMock

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
from torch import nn

from oracle import golden_configs as G
from oracle import paella_oracle as O
from paella_amd import synth

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)


def load_module(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class VQStandIn(nn.Module):
    """Call-site contract of torchtools.nn.VectorQuantize; algorithm = oracle.vector_quantize."""

    def __init__(self, embedding_size, k):
        super().__init__()
        self.codebook = nn.Embedding(k, embedding_size)

    def forward(self, x, get_losses=True, dim=-1):
        if dim != -1:
            x = x.movedim(dim, -1)
        rows = x.contiguous().view(-1, x.size(-1))
        q, idx = O.vector_quantize(rows, self.codebook.weight.detach())
        mse = (q - rows).pow(2).mean()
        q = q.view(x.shape)
        if dim != -1:
            q = q.movedim(-1, dim)
        return q, (mse, mse), idx.view(x.shape[:-1])

    def idx2vq(self, idx, dim=-1):
        q = self.codebook(idx)
        return q.movedim(-1, dim) if dim != -1 else q


def import_reference():
    import transformers  # noqa: F401  (must be imported before torchvision is mocked -- SURVEY D9)
    sys.modules["torchvision"] = MagicMock()
    tt = types.ModuleType("torchtools")
    ttn = types.ModuleType("torchtools.nn")
    ttn.VectorQuantize = VQStandIn
    tt.nn = ttn
    sys.modules["torchtools"] = tt
    sys.modules["torchtools.nn"] = ttn
    sys.path.insert(0, os.path.join(REF, "src"))
    ref = {}
    ref["modules"] = load_module(os.path.join(REF, "src", "modules.py"), "ref_src_modules")
    ref["vqgan"] = load_module(os.path.join(REF, "src", "vqgan.py"), "vqgan")
    sys.modules["vqgan"] = ref["vqgan"]
    ref["utils"] = load_module(os.path.join(REF, "src", "utils.py"), "ref_src_utils")
    ref["utils_modules"] = load_module(os.path.join(REF, "utils", "modules.py"), "ref_utils_modules")
    ref["alter"] = load_module(os.path.join(REF, "utils", "alter_attention.py"), "ref_alter_attention")
    # src_distributed/utils.py imports webdataset/open_clip at module level: exec only its sample() source
    src = open(os.path.join(REF, "src_distributed", "utils.py")).read()
    start = src.index("def sample(")
    ns = {"torch": torch}
    exec(compile(src[start:], "ref_src_distributed_sample", "exec"), ns)
    ref["sample_distributed"] = ns["sample"]
    return ref


def make_ref_unet(ref_mod, cfg, seed):
    torch.manual_seed(0)
    m = ref_mod.Paella(**cfg).eval()
    sd = synth.synth_state_dict(m.state_dict(), seed=seed, n_blocks=sum(cfg["blocks"]))
    m.load_state_dict(sd)
    return m, sd


def cond_for(cfg, B, S_byt5, n_img, seed):
    return synth.synth_conditioning(B, S_byt5, cfg["byt5_embd"], cfg["clip_embd"], seed=seed, with_clip=True, n_clip_image=n_img)


def keyshapes(sd):
    return np.array([k + ":" + ",".join(str(d) for d in v.shape) for k, v in sorted(sd.items())])


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.numpy() if torch.is_tensor(v) else v) for k, v in arrays.items()})
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


TRAIN_GOLDENS = {  # fixture name -> (config, parameters whose gradient is stored in full)
    "train_tiny_step": ("UNET_TINY", ("in_mapper.0.weight", "byt5_mapper.weight", "down_blocks.1.3.attention.attn.in_proj_weight", "down_blocks.1.1.channelwise.2.gamma",
                                      "down_blocks.0.1.mapper.weight", "up_blocks.0.0.depthwise.weight", "clf.1.weight", "out_mapper.1.weight", "up_blocks.1.6.1.weight")),
    # FeedForwardBlock ('F'), cross-attention only (self_attn=False), patch_size 1, two levels, level_config ['CFT', 'TAC']
    "train_variant_step": ("UNET_VARIANT", None),
}


def make_train_golden(ref, name="train_tiny_step"):
    """One training step of the REFERENCE (src_distributed/train.py:98-114 call sequence: add_noise -> get_loss_weight -> model(...)
    in train mode with dropout 0.1 -> label-smoothed CE weighted by loss_weight -> backward); stores the loss, the logits and the
    gradient of every parameter as (norm, sum) plus a few small tensors in full."""
    cfg_name, full_keys = TRAIN_GOLDENS[name]
    cfg = getattr(G, cfg_name)
    # src_distributed/modules.py carries get_loss_weight; same network as src/modules.py
    mod = load_module(os.path.join(REF, "src_distributed", "modules.py"), "ref_dist_modules")
    torch.manual_seed(0)
    m = mod.Paella(**cfg)
    sd = synth.synth_state_dict(m.state_dict(), seed=G.WEIGHT_SEED, n_blocks=sum(cfg["blocks"]))
    m.load_state_dict(sd)
    latents, t, mask, random_x, c = G.train_step_inputs(cfg)
    out = {}
    for tag, p_drop_seed in (("nodrop", None), ("drop", 1234)):
        m.train()
        for mm in m.modules():  # "nodrop": dropout off but still train mode (pure gradient check); "drop": the reference's 0.1
            if isinstance(mm, nn.Dropout):
                mm.p = 0.0 if p_drop_seed is None else 0.1
            if isinstance(mm, nn.MultiheadAttention):
                mm.dropout = 0.0 if p_drop_seed is None else 0.1
        m.zero_grad(set_to_none=True)
        noised, mk = m.add_noise(latents, t, mask=mask, random_x=random_x)
        lw = m.get_loss_weight(t, mk)
        if p_drop_seed is not None:
            torch.manual_seed(p_drop_seed)
        pred = m(noised, t, **c)
        loss = nn.CrossEntropyLoss(label_smoothing=0.1, reduction='none')(pred, latents)
        loss = ((loss * lw).sum(dim=[1, 2]) / lw.sum(dim=[1, 2])).mean()
        loss.backward()
        names = [k for k, _ in m.named_parameters()]
        out[tag + "_loss"] = loss.detach()
        out[tag + "_pred_sub"] = pred.detach()[:, ::4, ::2, ::2].contiguous()
        out[tag + "_grad_norms"] = torch.stack([p.grad.norm() for _, p in m.named_parameters()])
        out[tag + "_grad_sums"] = torch.stack([p.grad.double().sum().float() for _, p in m.named_parameters()])
        keys = full_keys
        if keys is None:  # the first parameter of every kind, small tensors only
            keys, seen = [], set()
            for k, p in m.named_parameters():
                kind = ".".join(s for s in k.split(".") if not s.isdigit())
                if kind not in seen and p.numel() <= 6000 and len(keys) < 16:
                    seen.add(kind)
                    keys.append(k)
        for k in keys:
            out[tag + "_grad:" + k] = dict(m.named_parameters())[k].grad.detach().clone()
    save(name, names=np.array(names), checksum=np.array(synth.checksum(sd)), **out)


def make_nonsquare_goldens(ref):
    """H != W fixtures.  The reference is fully convolutional (src/modules.py:130-134,153-156,172-183;
    src/vqgan.py:54-89): any token grid divisible by the down-sampling factor runs.  These pin the space-to-depth / depth-to-space /
    pixel-shuffle / k2s2 / 4-phase convT index maps of the HIP path for H != W: UNET_TINY forward at (2,16,32) and (1,24,8),
    the sample() closed loop at (1,16,32), VQModel(levels=2 / 3) encode / decode / decode_indices at 64x128 / 128x256 px."""
    with torch.no_grad():
        cfg = G.UNET_TINY
        m, sd = make_ref_unet(ref["modules"], cfg, G.WEIGHT_SEED)
        arrays = {}
        for tag, (B, H, W), seed in (("wide", (2, 16, 32), 17), ("tall", (1, 24, 8), 18)):
            g = torch.Generator().manual_seed(seed)
            x = torch.randint(0, cfg["num_labels"], (B, H, W), generator=g)
            r = torch.rand(B, generator=g)
            c = cond_for(cfg, B, 5, 1, G.COND_SEED + seed)
            logits = m(x, r, **c)
            lo = O.unet_forward(sd, cfg, x, r, **c)
            assert torch.allclose(lo, logits, atol=2e-5, rtol=1e-5), (lo - logits).abs().max()
            arrays.update({tag + "_x": x, tag + "_r": r, tag + "_logits": logits})
        save("unet_tiny_forward_nonsquare", checksum=np.array(synth.checksum(sd)), keys=keyshapes(sd), **arrays)

        # sample(), src/utils.py signature, on a 16x32 grid
        B1, H1, W1 = 1, 16, 32
        cs = cond_for(cfg, B1, 4, 0, G.COND_SEED)
        us = cond_for(cfg, B1, 4, 0, G.COND_SEED + 5)
        torch.manual_seed(G.SAMPLER_SEED + 7)
        toks = ref["utils"].sample(m, cs, (B1, H1, W1), unconditional_inputs=us, steps=8, renoise_steps=7, temperature=(1.0, 0.2),
                                   cfg=8.0, device="cpu")
        noise = O.replay_torch_noise(G.SAMPLER_SEED + 7, (B1, H1, W1), cfg["num_labels"], 8, 7)
        t_list = [float(v) for v in torch.linspace(1.0, 0.0, 9)]
        temps = [float(v) for v in torch.linspace(1.0, 0.2, 8)]
        cf = (float(torch.tensor(8.0)), float(torch.tensor(1.0 - 8.0)))
        fwd = lambda tk, rr, **inp: O.unet_forward(sd, cfg, tk, rr, **inp)
        otoks, traj = O.sample(fwd, cfg["num_labels"], cs, us, (B1, H1, W1), steps=8, renoise_steps=7, temperatures=temps,
                               cfgs=[cf] * 8, t_list=t_list, noise=noise)
        assert torch.equal(otoks, toks), "oracle sample loop does not reproduce the reference on the 16x32 grid"
        save("sample_tiny_nonsquare", tokens=toks, traj=torch.stack(traj))

        for name, vc, (hp, wp) in (("vq_tiny_f4_nonsquare", G.VQ_TINY_F4, (64, 128)), ("vq_tiny_f8_nonsquare", G.VQ_TINY_F8, (128, 256))):
            torch.manual_seed(0)
            vq = ref["vqgan"].VQModel(**vc).eval()
            vsd = synth.synth_state_dict(vq.state_dict(), seed=G.WEIGHT_SEED, n_blocks=vc["bottleneck_blocks"])
            vq.load_state_dict(vsd)
            gq = torch.Generator().manual_seed(6)
            img = torch.rand(1, 3, hp, wp, generator=gq)
            qe, lat, idx, loss = vq.encode(img)
            dec = vq.decode(qe)
            dec_i = vq.decode_indices(idx)
            oq, olat, oidx, oloss = O.vq_encode(vsd, vc, img)
            assert torch.equal(oidx, idx) and torch.allclose(olat, lat, atol=1e-5)
            assert torch.allclose(O.vq_decode_indices(vsd, vc, idx), dec_i, atol=2e-5)
            assert torch.allclose(O.vq_decode(vsd, vc, qe), dec, atol=2e-5)
            # the image is a function of the seed (torch.rand on a seeded CPU generator): tests regenerate it and check img_sum
            save(name, img_sum=np.array(float(img.double().sum())), qe=qe, lat=lat, idx=idx, loss=loss, dec=dec, dec_idx=dec_i,
                 checksum=np.array(synth.checksum(vsd)), keys=keyshapes(vsd))


def main():
    ref = import_reference()
    if "--only-nonsquare" in sys.argv:
        make_nonsquare_goldens(ref)
        return
    if "--only-train" in sys.argv:
        for name in TRAIN_GOLDENS:
            make_train_golden(ref, name)
        return
    for name in TRAIN_GOLDENS:
        make_train_golden(ref, name)
    with torch.no_grad():
        # ---- 1. forward, tiny ----
        cfg = G.UNET_TINY
        m, sd = make_ref_unet(ref["modules"], cfg, G.WEIGHT_SEED)
        B, H, W = 2, 16, 16
        g = torch.Generator().manual_seed(7)
        x = torch.randint(0, cfg["num_labels"], (B, H, W), generator=g)
        r = torch.rand(B, generator=g)
        c = cond_for(cfg, B, 5, 1, G.COND_SEED)
        logits = m(x, r, **c)
        taps = {}
        lo = O.unet_forward(sd, cfg, x, r, **c, taps=taps)
        assert torch.allclose(lo, logits, atol=2e-5, rtol=1e-5), (lo - logits).abs().max()
        save("unet_tiny_forward", logits=logits, x=x, r=r, r_embed=m.gen_r_embedding(r), c_embed=m.gen_c_embeddings(**c),
             checksum=np.array(synth.checksum(sd)), keys=keyshapes(sd))
        # no-clip / no-image variants of the conditioning (S changes)
        c2 = cond_for(cfg, B, 3, 0, G.COND_SEED + 1)
        save("unet_tiny_forward_textonly", logits=m(x, r, **c2), x=x, r=r)
        c3 = dict(c2, byt5=c2["byt5"][:, :0])  # CLIP-only: byt5 of length 0
        save("unet_tiny_forward_cliponly", logits=m(x, r, **c3), x=x, r=r)

        # ---- 2. attn_weights + list clip_image through utils/modules.py + alter_attention ----
        torch.manual_seed(0)
        mu = ref["utils_modules"].Paella(**cfg).eval()
        mu.load_state_dict(sd)
        ref["alter"].replace_attention_layers(mu)
        cl = cond_for(cfg, B, 5, 2, G.COND_SEED)
        aw = torch.tensor([2.0, 2.0, 0.5, 0.5, 1.5])
        save("unet_tiny_attnw", logits=mu(x, r, **cl, attn_weights=aw), logits_noaw=mu(x, r, **cl), attn_weights=aw, x=x, r=r)

        # ---- 3. add_noise ----
        t = torch.tensor([0.3, 0.8])
        torch.manual_seed(11)
        xn, mask = m.add_noise(x, t)
        gm = torch.Generator().manual_seed(12)
        um = torch.randint(0, 2, (B, H, W), generator=gm)
        rx = torch.randint(0, cfg["num_labels"], (B, H, W), generator=gm)
        xn2, mask2 = m.add_noise(x, t, mask=um, random_x=rx)
        save("add_noise", x=x, t=t, x_noised=xn, mask=mask, user_mask=um, random_x=rx, x_noised_user=xn2, mask_user=mask2)

        # ---- 4. sample(), src/utils.py signature: tiny model, 32x32 grid, 8 steps, batch 1 (BASELINE config 1) ----
        B1, H1, W1 = 1, 32, 32
        cs = cond_for(cfg, B1, 4, 0, G.COND_SEED)
        us = cond_for(cfg, B1, 4, 0, G.COND_SEED + 5)
        torch.manual_seed(G.SAMPLER_SEED)
        toks = ref["utils"].sample(m, cs, (B1, H1, W1), unconditional_inputs=us, steps=8, renoise_steps=7, temperature=(1.0, 0.2),
                                   cfg=8.0, device="cpu")
        noise = O.replay_torch_noise(G.SAMPLER_SEED, (B1, H1, W1), cfg["num_labels"], 8, 7)
        t_list = [float(v) for v in torch.linspace(1.0, 0.0, 9)]
        temps = [float(v) for v in torch.linspace(1.0, 0.2, 8)]
        cf = (float(torch.tensor(8.0)), float(torch.tensor(1.0 - 8.0)))
        fwd = lambda tk, rr, **inp: O.unet_forward(sd, cfg, tk, rr, **inp)
        otoks, traj = O.sample(fwd, cfg["num_labels"], cs, us, (B1, H1, W1), steps=8, renoise_steps=7, temperatures=temps,
                               cfgs=[cf] * 8, t_list=t_list, noise=noise)
        agree = (otoks == toks).float().mean().item()
        print("sample(): oracle vs reference token agreement %.4f" % agree)
        assert agree == 1.0, "oracle sample loop does not reproduce the reference"
        atoks, atraj = O.sample(fwd, cfg["num_labels"], cs, us, (B1, H1, W1), steps=8, renoise_steps=7, temperatures=temps,
                                cfgs=[cf] * 8, t_list=t_list, noise=noise, argmax=True)
        save("sample_tiny", tokens=toks, traj=torch.stack(traj), tokens_argmax=atoks, traj_argmax=torch.stack(atraj))

        # ---- 5. sample(), src_distributed signature: init_x, cfg schedule, conditional-step cutoff ----
        B2, H2, W2 = 2, 16, 16
        cd = cond_for(cfg, B2, 5, 1, G.COND_SEED)
        ud = cond_for(cfg, B2, 2, 0, G.COND_SEED + 5)  # different S for the unconditional set (as train.py:159-160)
        gi = torch.Generator().manual_seed(3)
        init_x = torch.randint(0, cfg["num_labels"], (B2, H2, W2), generator=gi)
        torch.manual_seed(G.SAMPLER_SEED + 1)
        toks_d = ref["sample_distributed"](m, cd, ud, (B2, H2, W2), init_x=init_x, steps=6, temperature=(0.7, 0.3), cfg=(8.0, 4.0),
                                           t_start=0.8, sampling_conditional_steps=4)
        save("sample_tiny_distributed", tokens=toks_d, init_x=init_x)

        # ---- 6. forward, mid config (head_dim 80), CLIP-only ----
        cfgm = G.UNET_MID
        mm, sdm = make_ref_unet(ref["modules"], cfgm, G.WEIGHT_SEED)
        gmid = torch.Generator().manual_seed(8)
        xm = torch.randint(0, cfgm["num_labels"], (1, 16, 16), generator=gmid)
        rm = torch.rand(1, generator=gmid)
        cmid = cond_for(cfgm, 1, 0, 0, G.COND_SEED)
        lm = mm(xm, rm, **cmid)
        lom = O.unet_forward(sdm, cfgm, xm, rm, **cmid)
        assert torch.allclose(lom, lm, atol=5e-5, rtol=1e-5), (lom - lm).abs().max()
        save("unet_mid_forward", logits_sub=lm[:, :, ::2, ::2].contiguous(), argmax=lm.argmax(1), x=xm, r=rm,
             top2_margin=(lm.topk(2, dim=1).values[:, 0] - lm.topk(2, dim=1).values[:, 1]),
             checksum=np.array(synth.checksum(sdm)), keys=keyshapes(sdm))

        # ---- 7. forward, variant config (F blocks, cross-attention only, patch 1) ----
        cfgv = G.UNET_VARIANT
        mv, sdv = make_ref_unet(ref["modules"], cfgv, G.WEIGHT_SEED)
        gv = torch.Generator().manual_seed(9)
        xv = torch.randint(0, cfgv["num_labels"], (2, 8, 8), generator=gv)
        rv = torch.rand(2, generator=gv)
        cv = cond_for(cfgv, 2, 3, 1, G.COND_SEED)
        lv = mv(xv, rv, **cv)
        assert torch.allclose(O.unet_forward(sdv, cfgv, xv, rv, **cv), lv, atol=2e-5, rtol=1e-5)
        save("unet_variant_forward", logits=lv, x=xv, r=rv, checksum=np.array(synth.checksum(sdv)), keys=keyshapes(sdv))

        # ---- 8. VQGAN f4 / f8 tiny ----
        for name, vc in (("vq_tiny_f4", G.VQ_TINY_F4), ("vq_tiny_f8", G.VQ_TINY_F8)):
            torch.manual_seed(0)
            vq = ref["vqgan"].VQModel(**vc).eval()
            vsd = synth.synth_state_dict(vq.state_dict(), seed=G.WEIGHT_SEED, n_blocks=vc["bottleneck_blocks"])
            vq.load_state_dict(vsd)
            gq = torch.Generator().manual_seed(5)
            img = torch.rand(2, 3, 32, 32, generator=gq)
            qe, lat, idx, loss = vq.encode(img)
            dec = vq.decode(qe)
            dec_i = vq.decode_indices(idx)
            oq, olat, oidx, oloss = O.vq_encode(vsd, vc, img)
            assert torch.equal(oidx, idx) and torch.allclose(olat, lat, atol=1e-5)
            assert torch.allclose(O.vq_decode_indices(vsd, vc, idx), dec_i, atol=2e-5)
            assert torch.allclose(O.vq_decode(vsd, vc, qe), dec, atol=2e-5)
            save(name, img=img, qe=qe, lat=lat, idx=idx, loss=loss, dec=dec, dec_idx=dec_i, checksum=np.array(synth.checksum(vsd)),
                 keys=keyshapes(vsd))
    make_nonsquare_goldens(ref)
    print("done")


if __name__ == "__main__":
    main()
