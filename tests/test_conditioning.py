"""CPU: the conditioning front-end and checkpoint loading (reference src_distributed/utils.py:65-82, train.py:143-152,
src/utils.py:23-31).  No released weights exist offline, so the encoders are built from (small) transformers configs with random weights:
what is checked is the reference's call structure, the tensor shapes / dtypes `sample()` needs, the exact ByT5 tokenisation (byte-level,
needs no vocabulary file), checkpoint files in the reference's `{'state_dict': ...}` layout, and the preprocessing arithmetic."""
import os

import pytest
import torch

import paella_amd
from oracle import golden_configs as G
from paella_amd import conditioning as C
from paella_amd import synth

transformers = pytest.importorskip("transformers")


def small_configs():
    byt5 = transformers.T5Config(vocab_size=384, d_model=G.UNET_TINY["byt5_embd"], d_kv=8, d_ff=64, num_layers=2, num_heads=4,
                                 feed_forward_proj="gated-gelu", tie_word_embeddings=False)
    clip = transformers.CLIPConfig(
        text_config=dict(vocab_size=300, hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=4, max_position_embeddings=77,
                         projection_dim=G.UNET_TINY["clip_embd"], eos_token_id=299, bos_token_id=298, pad_token_id=0),
        vision_config=dict(hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=4, image_size=224, patch_size=32,
                           projection_dim=G.UNET_TINY["clip_embd"]),
        projection_dim=G.UNET_TINY["clip_embd"])
    return byt5, clip


def test_load_conditional_models_structure_and_embeddings(tmp_path):
    vq = paella_amd.VQModel(**G.VQ_TINY_F8)
    vq_sd = synth.randomize_(vq, seed=3)
    path = os.path.join(tmp_path, "vqgan.pt")
    torch.save({"state_dict": vq.state_dict(), "iter": 7}, path)   # the reference's checkpoint layout (src/train.py:40)
    byt5_cfg, clip_cfg = small_configs()
    torch.manual_seed(0)
    vqgan, (clip_tok, clip_model, preprocess), (byt5_tok, byt5) = C.load_conditional_models(
        ("ViT-H-14", "laion2b_s32b_b79k"), "google/byt5-xl", path, "cpu", encoders="config", vqgan_kwargs=G.VQ_TINY_F8,
        byt5_config=byt5_cfg, clip_config=clip_cfg)
    # frozen, eval, weights from the file
    assert not vqgan.training and not any(p.requires_grad for p in vqgan.parameters())
    for k, v in vqgan.state_dict().items():
        assert torch.equal(v, vq_sd[k].to(v.dtype)), k
    assert not byt5.training and not any(p.requires_grad for p in byt5.parameters())
    assert not clip_model.training and not any(p.requires_grad for p in clip_model.parameters())

    # ByT5 tokenisation is byte-level: utf-8 bytes + 3, EOS = 1, pad = 0 (identical to the released tokenizer; no vocabulary file exists)
    ids = byt5_tok(["hi", ""], padding="longest", return_tensors="pt", max_length=768, truncation=True).input_ids
    assert ids.tolist() == [[ord("h") + 3, ord("i") + 3, 1], [1, 0, 0]]

    captions = ["a photograph of a paella", "xyz"]
    images = torch.rand(2, 3, 256, 256)
    cond, uncond = C.embed_prompts(captions, (byt5_tok, byt5), (clip_tok, clip_model, preprocess), images=images)
    S = len(captions[0].encode()) + 1
    assert cond["byt5"].shape == (2, S, G.UNET_TINY["byt5_embd"]) and cond["byt5"].dtype == torch.float32
    assert uncond["byt5"].shape == (2, 1, G.UNET_TINY["byt5_embd"])           # '' -> just EOS (train.py:145)
    assert cond["clip"].shape == uncond["clip"].shape == (2, G.UNET_TINY["clip_embd"])
    assert cond["clip_image"].shape == (2, G.UNET_TINY["clip_embd"]) and uncond["clip_image"] is None   # train.py:159-160
    assert torch.equal(uncond["clip"][0], uncond["clip"][1]) and not torch.equal(cond["clip"][0], cond["clip"][1])
    # ByT5-only form of src/utils.py:23
    v2, (t2, b2) = C.load_conditional_models(None, "google/byt5-xl", None, "cpu", encoders="config", vqgan_kwargs=G.VQ_TINY_F8, byt5_config=byt5_cfg)
    c2, u2 = C.embed_prompts(["q"], (t2, b2))
    assert c2["clip"] is None and c2["byt5"].shape == (1, 2, G.UNET_TINY["byt5_embd"])


def test_pretrained_mode_fails_loudly_offline(tmp_path):
    with pytest.raises(Exception):  # no network, no local files: the underlying transformers error surfaces, nothing is silently random-initialised
        C.load_conditional_models(("ViT-H-14", "laion2b_s32b_b79k"), os.path.join(tmp_path, "no-such-byt5"), None, "cpu", vqgan_kwargs=G.VQ_TINY_F8)
    with pytest.raises(ValueError):
        C.load_conditional_models(None, "x", None, "cpu", encoders="nope", vqgan_kwargs=G.VQ_TINY_F8)


def test_checkpoint_layouts(tmp_path):
    m = paella_amd.Paella(**G.UNET_TINY)
    sd = synth.randomize_(m, seed=5)
    p1, p2, p3 = (os.path.join(tmp_path, n) for n in ("a.pt", "b.pt", "c.pt"))
    torch.save({"state_dict": m.state_dict(), "iter": 123, "optimizer_state_dict": {}}, p1)   # src_distributed/train.py:131-137
    torch.save({("module." + k): v for k, v in m.state_dict().items()}, p2)                     # a DDP wrapper saved whole
    torch.save({"state_dict": {"bogus": torch.zeros(1)}}, p3)
    for p in (p1, p2):
        m2 = paella_amd.Paella(**G.UNET_TINY)
        extra = C.load_checkpoint(m2, p)
        for k, v in m2.state_dict().items():
            assert torch.equal(v, sd[k]), k
    assert extra == {} and C.load_checkpoint(paella_amd.Paella(**G.UNET_TINY), p1)["iter"] == 123
    with pytest.raises(RuntimeError):
        C.load_checkpoint(paella_amd.Paella(**G.UNET_TINY), p3)
    assert next(C.build_paella(p1, device="cpu", **G.UNET_TINY).parameters()).device.type == "cpu"


def test_clip_preprocess_matches_resize_normalize():
    x = torch.rand(2, 3, 256, 320)
    y = C.clip_preprocess(x)
    assert y.shape == (2, 3, 224, 280)     # smaller edge -> 224, aspect kept (torchvision Resize(224))
    ref = torch.nn.functional.interpolate(x, size=(224, 280), mode="bicubic", align_corners=False, antialias=True)
    for c in range(3):
        torch.testing.assert_close(y[:, c], (ref[:, c] - C.CLIP_MEAN[c]) / C.CLIP_STD[c])
    assert C.clip_preprocess(torch.rand(1, 3, 224, 224)).shape == (1, 3, 224, 224)
    with pytest.raises(ValueError):
        C.clip_preprocess(torch.rand(3, 224, 224))


def test_signature_mirrors_the_reference():
    """Positional parameters of load_conditional_models = the reference's (src_distributed/utils.py:65).  The reference tree exists in the authoring
    container only; elsewhere the expected names are the ones recorded here from it."""
    import inspect
    import re
    expected = ["clip_model_name", "byt5_model_name", "vqgan_path", "device"]
    ref = "/root/reference/src_distributed/utils.py"
    if os.path.exists(ref):
        m = re.search(r"def load_conditional_models\(([^)]*)\)", open(ref).read())
        assert [a.strip() for a in m.group(1).split(",")] == expected
    sig = inspect.signature(C.load_conditional_models)
    positional = [p.name for p in sig.parameters.values() if p.kind == p.POSITIONAL_OR_KEYWORD]
    assert positional == expected
    assert all(p.kind == p.KEYWORD_ONLY for n, p in sig.parameters.items() if n not in expected)


def test_spec_layout_equals_layout_from_tensors():
    """cond_spec_layout (what a rank that never sees the tensors computes) == conditioning_layout (from the tensors): the one-collective path."""
    from paella_amd.dist import cond_spec_layout, conditioning_layout

    class M:
        _cfg = dict(byt5_embd=24, clip_embd=10)
    B = 3
    for S, Su, clip, n_img in [(5, 1, True, 0), (0, None, True, 1), (7, 2, False, 2)]:
        mk = lambda s, img: {"byt5": torch.zeros(B, s, 24), "clip": torch.zeros(B, 10) if clip else None,
                             "clip_image": None if img == 0 else (torch.zeros(B, 10) if img == 1 else [torch.zeros(B, 10)] * img)}
        want = conditioning_layout([mk(S, n_img), mk(S if Su is None else Su, n_img)])
        got = cond_spec_layout(M(), B, S_byt5=S, S_byt5_uncond=Su, clip=clip, n_clip_image=n_img)
        assert got[1] == want[1]
        assert [[tuple(e) if not isinstance(e, tuple) else e for e in d] for d in got[0]] == [list(d) for d in want[0]]
