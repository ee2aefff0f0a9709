"""Conditioning front-end and checkpoint loading: what the reference does AROUND the hot path to turn prompts / images
into the `model_inputs` dicts `sample()` consumes, and to load released weights.

Mirrors
  * `load_conditional_models(clip_model_name, byt5_model_name, vqgan_path, device)` -- src_distributed/utils.py:65-82 (superset of
    src/utils.py:23-31): VQGAN from `torch.load(path)['state_dict']`, frozen ByT5 encoder + tokenizer, frozen CLIP + tokenizer + preprocess;
    same return structure `vqgan, (clip_tokenizer, clip_model, clip_preprocess), (byt5_tokenizer, byt5)`;
  * the embedding calls of src_distributed/train.py:143-152 (`embed_prompts`): ByT5 `last_hidden_state` of the tokenised captions
    (padding "longest", max_length 768), CLIP `encode_text` / `encode_image`, and the same for the empty prompt (unconditional set);
  * checkpoint files `{'state_dict': ...}` (src/train.py:40, src_distributed/train.py:131-137): `load_checkpoint`.

The encoders are EXTERNAL frozen models, not part of the hot path: they run through `transformers` as plain torch modules
on whatever device they are put on; only what they feed -- the denoiser and the VQGAN -- is this package's HIP code.  Differences from the
reference, all forced by what is installable here and stated where they bite:
  * CLIP comes from `open_clip` when that package is importable (the reference's own call sequence) and otherwise -- as in this image, where it is
    absent and there is no network -- from `transformers` (`CLIPModel`): `('ViT-H-14', 'laion2b_s32b_b79k')` maps to the Hugging Face export of the SAME checkpoint, `laion/CLIP-ViT-H-14-laion2B-s32B-b79K`; the wrapper exposes open_clip's `encode_text` /
    `encode_image` (projected, un-normalised features, as open_clip returns by default);
  * `torchvision` is absent: `clip_preprocess` is the same Resize(224, bicubic, antialias) + Normalize written with torch ops;
  * offline, `from_pretrained` can only succeed from a local directory / cache.  `encoders="config"` builds the encoders from configs with
    random weights (shape-faithful plumbing for tests and benchmarks); real-weight fidelity cannot be verified in this environment.
"""
import torch
import torch.nn.functional as F

from .modules import Paella
from .vqgan import VQModel

# open_clip (architecture, pretrained tag) -> Hugging Face repository holding the same weights in transformers layout
OPEN_CLIP_TO_HF = {("ViT-H-14", "laion2b_s32b_b79k"): "laion/CLIP-ViT-H-14-laion2B-s32B-b79K"}
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def load_checkpoint(module, path, map_location=None, key="state_dict", strict=True, weights_only=True):
    """`module.load_state_dict(torch.load(path)['state_dict'])` (src/utils.py:26, src/train.py:40, src_distributed/utils.py:67).  Accepts the
    reference's checkpoint files unchanged: a dict with the weights under `key`; a bare state dict is accepted too, and a leading
    'module.' (a DistributedDataParallel wrapper saved whole) is stripped.  Returns the checkpoint's other entries (iter, optimizer state, ...).
    weights_only=True (default) restricts unpickling to tensors / plain containers -- what the reference's checkpoints hold (state dicts, optimizer and
    scaler state, counters, strings); pass False only for a trusted file that pickles other objects."""
    ckpt = torch.load(path, map_location=map_location if map_location is not None else "cpu", weights_only=weights_only)
    sd = ckpt[key] if isinstance(ckpt, dict) and key in ckpt else ckpt
    if not isinstance(sd, dict) or not sd:
        raise ValueError("%s holds no state dict under %r" % (path, key))
    if all(k.startswith("module.") for k in sd):
        sd = {k[len("module."):]: v for k, v in sd.items()}
    module.load_state_dict(sd, strict=strict)
    return {k: v for k, v in ckpt.items() if k != key} if isinstance(ckpt, dict) and key in ckpt else {}


def clip_preprocess(images, size=224):
    """torchvision Compose([Resize(224, BICUBIC), Normalize(CLIP mean / std)]) on a float [B,3,H,W] tensor in [0,1]
    (src_distributed/utils.py:77-80): the smaller edge is resized to `size` (aspect kept, antialiased bicubic), then normalised."""
    if images.dim() != 4 or images.size(1) != 3:
        raise ValueError("clip_preprocess expects [B, 3, H, W]")
    h, w = images.shape[-2:]
    if h <= w:
        nh, nw = size, max(size, int(size * w / h))
    else:
        nh, nw = max(size, int(size * h / w)), size
    x = F.interpolate(images.float(), size=(nh, nw), mode="bicubic", align_corners=False, antialias=True)
    mean = torch.tensor(CLIP_MEAN, device=x.device, dtype=x.dtype)[None, :, None, None]
    std = torch.tensor(CLIP_STD, device=x.device, dtype=x.dtype)[None, :, None, None]
    return (x - mean) / std


class ClipEncoders(torch.nn.Module):
    """open_clip's `encode_text(tokens)` / `encode_image(pixels)` over a `transformers.CLIPModel`: projected, un-normalised features
    [B, projection_dim] (what src_distributed/train.py:92,97 feed the denoiser after `.float()`).  Non-square inputs are centre-cropped
    to the vision tower's square input (the reference's RandomCrop(256) images are square already)."""

    def __init__(self, hf_clip):
        super().__init__()
        self.model = hf_clip

    @property
    def embed_dim(self):
        return self.model.config.projection_dim

    def encode_text(self, tokens):
        out = self.model.text_model(input_ids=tokens)
        return self.model.text_projection(out.pooler_output)

    def encode_image(self, pixels):
        s = self.model.config.vision_config.image_size
        h, w = pixels.shape[-2:]
        if (h, w) != (s, s):
            top, left = max(0, (h - s) // 2), max(0, (w - s) // 2)
            pixels = pixels[..., top:top + s, left:left + s]
        out = self.model.vision_model(pixel_values=pixels)
        return self.model.visual_projection(out.pooler_output)


class ClipTokenizer:
    """open_clip.get_tokenizer(...) call shape: `tokenizer(list_of_str) -> LongTensor [B, context_length]` (src_distributed/train.py:91),
    over a `transformers` CLIP tokenizer (padding to the context length, truncation)."""

    def __init__(self, hf_tokenizer, context_length=77):
        self.tok, self.context_length = hf_tokenizer, context_length

    def __call__(self, texts):
        if isinstance(texts, str):
            texts = [texts]
        return self.tok(list(texts), padding="max_length", truncation=True, max_length=self.context_length, return_tensors="pt").input_ids


class ByteClipTokenizer:
    """Stand-in CLIP tokenizer for config-initialised encoders (no vocabulary files offline): UTF-8 bytes + 2, BOS = vocab-2, EOS = vocab-1,
    zero padding (a custom `clip_config` must set text_config.eos_token_id = vocab_size - 1: CLIP pools the hidden state at the EOS position).
    NOT the BPE of the released checkpoints -- only for shape-faithful plumbing with random-weight encoders."""

    def __init__(self, vocab_size, context_length=77):
        self.vocab_size, self.context_length = vocab_size, context_length

    def __call__(self, texts):
        if isinstance(texts, str):
            texts = [texts]
        out = torch.zeros(len(texts), self.context_length, dtype=torch.long)
        for i, t in enumerate(texts):
            ids = [self.vocab_size - 2] + [2 + (b % (self.vocab_size - 4)) for b in t.encode("utf-8")][: self.context_length - 2] + [self.vocab_size - 1]
            out[i, :len(ids)] = torch.tensor(ids)
        return out


def _freeze(m, device):
    return m.to(device).eval().requires_grad_(False)


def load_conditional_models(clip_model_name, byt5_model_name, vqgan_path, device, *, encoders="pretrained", vqgan_kwargs=None,
                            byt5_config=None, clip_config=None):
    """Drop-in for src_distributed/utils.py:65 (pass clip_model_name=None for the ByT5-only form of src/utils.py:23).
    Returns `vqgan, (clip_tokenizer, clip_model, clip_preprocess), (byt5_tokenizer, byt5)`.

    vqgan_path: a `{'state_dict': ...}` checkpoint (None = randomly initialised VQGAN); vqgan_kwargs: VQModel constructor arguments
    (the reference builds `VQModel()`, i.e. f4; BASELINE's 256 px <-> 32x32 tokens needs levels=3).
    encoders="pretrained": `from_pretrained(name)` -- offline this needs local files and raises the underlying error otherwise;
    encoders="config": build ByT5 / CLIP from `byt5_config` / `clip_config` (transformers config objects; defaults = the released sizes,
    ByT5-XL d_model 2560 and CLIP ViT-H/14 projection 1024) with RANDOM weights and byte-level stand-in tokenizers."""
    import transformers

    vqgan = VQModel(**(vqgan_kwargs or {}))
    if vqgan_path is not None:
        load_checkpoint(vqgan, vqgan_path, map_location="cpu")
    vqgan = _freeze(vqgan, device)

    if encoders not in ("pretrained", "config"):
        raise ValueError("encoders must be 'pretrained' or 'config'")
    if encoders == "pretrained":
        byt5 = transformers.T5EncoderModel.from_pretrained(byt5_model_name)
        byt5_tokenizer = transformers.AutoTokenizer.from_pretrained(byt5_model_name)
    else:
        cfg = byt5_config or transformers.T5Config(vocab_size=384, d_model=2560, d_kv=64, d_ff=6720, num_layers=36, num_heads=32,
                                                   feed_forward_proj="gated-gelu", tie_word_embeddings=False)  # google/byt5-xl encoder
        byt5 = transformers.T5EncoderModel(cfg)
        byt5_tokenizer = transformers.ByT5Tokenizer()  # byte-level: needs no vocabulary file, identical to the released tokenizer
    byt5 = _freeze(byt5, device)
    if clip_model_name is None:
        return vqgan, (byt5_tokenizer, byt5)

    if encoders == "pretrained" and not isinstance(clip_model_name, str):
        try:  # the reference's own loader when it is installed (it is not in this image): src_distributed/utils.py:73-75, verbatim call sequence
            import open_clip
        except ImportError:
            open_clip = None
        if open_clip is not None:
            clip_model, _, _ = open_clip.create_model_and_transforms(clip_model_name[0], pretrained=clip_model_name[1], device=device)
            clip_model = _freeze(clip_model, device)
            return vqgan, (open_clip.get_tokenizer(clip_model_name[0]), clip_model, clip_preprocess), (byt5_tokenizer, byt5)
    if encoders == "pretrained":
        repo = OPEN_CLIP_TO_HF.get(tuple(clip_model_name), None) if not isinstance(clip_model_name, str) else clip_model_name
        if repo is None:
            raise ValueError("no transformers export known for open_clip model %r; pass a Hugging Face repository / local path instead" % (clip_model_name,))
        hf_clip = transformers.CLIPModel.from_pretrained(repo)
        clip_tokenizer = ClipTokenizer(transformers.CLIPTokenizer.from_pretrained(repo), hf_clip.config.text_config.max_position_embeddings)
    else:
        ccfg = clip_config or transformers.CLIPConfig(
            text_config=dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, max_position_embeddings=77,
                             vocab_size=49408, projection_dim=1024, hidden_act="gelu"),
            vision_config=dict(hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=16, image_size=224, patch_size=14,
                               projection_dim=1024, hidden_act="gelu"),
            projection_dim=1024)  # ViT-H/14
        hf_clip = transformers.CLIPModel(ccfg)
        clip_tokenizer = ByteClipTokenizer(ccfg.text_config.vocab_size, ccfg.text_config.max_position_embeddings)
    clip_model = _freeze(ClipEncoders(hf_clip), device)
    return vqgan, (clip_tokenizer, clip_model, clip_preprocess), (byt5_tokenizer, byt5)


@torch.no_grad()
def embed_prompts(captions, byt5_pair, clip_triple=None, images=None, device=None, max_length=768):
    """The embedding calls of src_distributed/train.py:143-152: returns `(model_inputs, unconditional_inputs)` for `sample()`.
    model_inputs = {'byt5': ByT5 last_hidden_state of the captions, 'clip': CLIP text features, 'clip_image': CLIP image features of
    `images` (None without images)}; unconditional_inputs = the same for the EMPTY caption, clip_image None (train.py:159-160)."""
    byt5_tokenizer, byt5 = byt5_pair
    device = device if device is not None else next(byt5.parameters()).device
    captions = list(captions)

    def byt5_embed(texts):
        ids = byt5_tokenizer(texts, padding="longest", return_tensors="pt", max_length=max_length, truncation=True).input_ids.to(device)
        return byt5(input_ids=ids).last_hidden_state.float()

    cond = {"byt5": byt5_embed(captions), "clip": None, "clip_image": None}
    uncond = {"byt5": byt5_embed([""] * len(captions)), "clip": None, "clip_image": None}
    if clip_triple is not None:
        clip_tokenizer, clip_model, preprocess = clip_triple
        cond["clip"] = clip_model.encode_text(clip_tokenizer(captions).to(device)).float()
        uncond["clip"] = clip_model.encode_text(clip_tokenizer([""] * len(captions)).to(device)).float()
        if images is not None:
            cond["clip_image"] = clip_model.encode_image(preprocess(images.to(device))).float()
    return cond, uncond


def build_paella(checkpoint_path=None, device="cuda", **ctor):
    """`Paella(**ctor)` (src_distributed/train.py:48 builds `Paella(byt5_embd=2560)`) with an optional reference checkpoint, moved to `device`."""
    model = Paella(**ctor)
    if checkpoint_path is not None:
        load_checkpoint(model, checkpoint_path, map_location="cpu")
    return model.to(device)
