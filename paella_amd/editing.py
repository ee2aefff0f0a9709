"""Inpainting / structural editing on the token grid (BASELINE config 5).

The reference snapshot has no dedicated function for this (it lived in the missing notebook); it is the composition of
on-disk pieces, reproduced here with the same calls:
    VQModel.encode(img)[2]                                   -> tokens              (src/vqgan.py:91-95)
    Paella.add_noise(tokens, t, mask=user_mask, random_x=..) -> masked renoise      (src/modules.py:277-283)
    sample(..., init_x=noised, t_start<1)                    -> denoise             (src_distributed/utils.py:97-109)
    VQModel.decode_indices(tokens)                           -> image               (src/vqgan.py:103-107)
EXTENSION (not reference behaviour, labelled as such): `keep_known=True` re-imposes the known tokens on the result.

Every op is a HIP kernel (the token select included: paella_select_tokens), so the whole recipe can be captured in ONE HIP graph: `GraphInpainter`.
"""
import torch

from .sampling import GraphSampler, fresh_seed, linspace_schedule, sample_distributed, select_tokens, start_tokens

# the tokens add_noise writes into the masked region in the counter-based noise mode: Philox start tokens under a salted seed (so they differ from the
# sampler's own start tokens, which are keyed by the plain seed), a function of (seed, GLOBAL position) like every other random number of that mode
RANDOM_X_SALT = 0x5851F42D4C957F2D
_MASK64 = (1 << 64) - 1


def _philox_random_x(model, shape, seed, device, shard=None, out=None, seed_dev=None, row_offset_dev=None):
    return start_tokens(model.num_labels, shape, (int(seed) + RANDOM_X_SALT) & _MASK64, device, shard, out=out, seed_dev=seed_dev, row_offset_dev=row_offset_dev)


def inpaint(model, vqgan, images, mask, model_inputs, unconditional_inputs, steps=12, t_start=1.0, temperature=(0.7, 0.3),
            cfg=(8.0, 8.0), keep_known=True, decode=True, random_x=None, **kwargs):
    """images fp32 [B,3,Hp,Wp] in [0,1]; mask int/bool [B,h,w] on the TOKEN grid (1 = regenerate); random_x (optional) the tokens
    add_noise writes into the masked region -- default: torch.randint_like, as Paella.add_noise draws them; with noise="philox" (kwargs) a
    function of (seed, global position), so that a shard / a captured graph (GraphInpainter) reproduces the unsharded eager call bit for bit.
    Returns (tokens, image or None)."""
    tokens = vqgan.encode(images)[2]
    mask = mask.to(device=tokens.device, dtype=torch.int64)
    if mask.shape != tokens.shape:
        raise ValueError("mask must be given on the token grid %s" % (tuple(tokens.shape),))
    B = tokens.size(0)
    if random_x is None and kwargs.get("noise") == "philox":
        if kwargs.get("seed") is None:
            kwargs["seed"] = fresh_seed()
        random_x = _philox_random_x(model, tuple(tokens.shape), kwargs["seed"], tokens.device, kwargs.get("shard"))
    t = torch.full((B,), float(t_start), device=tokens.device)
    noised, _ = model.add_noise(tokens, t, mask=mask, random_x=random_x)
    out = sample_distributed(model, model_inputs, unconditional_inputs, tuple(tokens.shape), init_x=noised, steps=steps,
                             temperature=temperature, cfg=cfg, t_start=t_start, **kwargs)
    if keep_known:  # extension: the reference's sample() may also rewrite known positions
        out = select_tokens(out, tokens, mask)  # == out * mask + tokens * (1 - mask) for a 0/1 mask
    return out, (vqgan.decode_indices(out) if decode else None)


class GraphInpainter(GraphSampler):
    """`inpaint(..., noise="philox")` -- VQGAN encode -> masked renoise -> sample(init_x, t_start) -> re-impose known tokens -> VQGAN decode -- captured ONCE
    into a HIP graph for fixed shapes (BASELINE configs[4]) and replayed per request with fresh images / masks / conditioning / seed / shard offset.
    Bit-identical to the eager call with the same seed (tests/test_gpu_sample.py).  Staleness handling as GraphSampler."""

    def __init__(self, model, vqgan, images, mask, model_inputs, unconditional_inputs, steps=12, t_start=1.0, temperature=(0.7, 0.3), cfg=(8.0, 8.0),
                 keep_known=True, device="cuda", attn_weights=None, on_stale="recapture"):
        dev = torch.device(device)
        f = 2 ** vqgan.levels
        B, _, Hp, Wp = images.shape
        shape = (B, Hp // f, Wp // f)
        if tuple(mask.shape) != shape:
            raise ValueError("mask must be given on the token grid %s" % (shape,))
        self.images = images.detach().to(device=dev, dtype=torch.float32).clone()
        self.mask = mask.to(device=dev, dtype=torch.int64).clone()
        self.random_x = torch.zeros(shape, dtype=torch.int64, device=dev)
        self.keep_known = bool(keep_known)
        self._t0 = torch.full((B,), float(t_start), device=dev)
        self._cfg_schedule = cfg
        self._known = None
        super().__init__(model, model_inputs, unconditional_inputs, shape, steps=steps, renoise_steps=steps - 1, temperature=temperature, cfg=cfg,
                         t_start=t_start, t_end=0.0, device=dev, vqgan=vqgan, attn_weights=attn_weights, on_stale=on_stale)

    def _schedule(self):
        """the src_distributed/utils.py:97-109 form (linear cfg schedule, `1 - cfg` in fp32), as sample_distributed"""
        k = self.kw
        t_list = linspace_schedule(k["t_start"], k["t_end"], k["steps"] + 1)
        temps = linspace_schedule(k["temperature"][0], k["temperature"][1], k["steps"])
        cfgs = [None] * k["steps"]
        if self._cfg_schedule is not None:
            sched = torch.linspace(self._cfg_schedule[0], self._cfg_schedule[1], k["steps"])
            cfgs = [(float(sched[i]), float(1 - sched[i])) for i in range(k["steps"])]
        return t_list, temps, cfgs

    def _init_x(self):
        self._known = self.vqgan.encode(self.images, ws=self.vq_ws)[2]
        rx = _philox_random_x(self.model, self.shape, 0, self.device, out=self.random_x, seed_dev=self.seed_dev, row_offset_dev=self.row_offset_dev)
        return self.model.add_noise(self._known, self._t0, mask=self.mask, random_x=rx)[0]

    def _finish(self, toks):
        return select_tokens(toks, self._known, self.mask) if self.keep_known else toks

    def __call__(self, images=None, mask=None, model_inputs=None, unconditional_inputs=None, seed=None, shard=None):
        """Replay; returns (tokens, image) in graph-owned buffers."""
        self._check_fresh()
        if images is not None:
            if tuple(images.shape) != tuple(self.images.shape):
                raise ValueError("image shape differs from the captured one")
            self.images.copy_(images)
        if mask is not None:
            if tuple(mask.shape) != tuple(self.mask.shape):
                raise ValueError("mask shape differs from the captured one")
            self.mask.copy_(mask)
        if model_inputs is not None:
            self._copy_inputs(self.cond, model_inputs)
        if unconditional_inputs is not None:
            self._copy_inputs(self.uncond, unconditional_inputs)
        self._set_words(seed, shard)
        self.graph.replay()
        return self.out
