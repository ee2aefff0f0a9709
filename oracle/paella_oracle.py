"""CPU oracle for the Paella sampling hot path -- TEST INFRASTRUCTURE, never part of the product path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product
(paella_amd/) never does and has no CPU fallback.

A functional restatement (torch CPU ops over a plain state dict, no nn.Module) of the reference algorithm:
  reference src/modules.py:7-106 (blocks), :212-283 (Paella.gen_r_embedding / gen_c_embeddings / _down_encode /
  _up_decode / forward / add_noise); utils/alter_attention.py:15-36 (attention written out, attn_weights);
  src/utils.py:35-55 and src_distributed/utils.py:97-126 (sample); src/vqgan.py:34-42, 91-107 (VQGAN).
Each function cites the lines it follows.

Pinning: the reference ships no tests or golden vectors, so this oracle is pinned against the
reference ITSELF: oracle/make_golden.py imports /root/reference/src/modules.py (and utils/modules.py,
utils/alter_attention.py, src/utils.py with stubs, src/vqgan.py with the VectorQuantize stand-in below) in the
authoring container, runs it on seeded synthetic weights/inputs and stores the outputs under tests/golden/;
tests/test_oracle.py checks this file against those fixtures (bit-level for token tensors, 1e-5-level for fp32).
One part is NOT pinned: `vector_quantize` restates the third-party torchtools.nn.VectorQuantize
(pabloppp/pytorch-tools, pulled by reference requirements.txt:12 with no version pin and absent from the
snapshot) from its published algorithm -- "parity unpinned" for the nearest-code search and its tie-breaking.
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------------------------
# building blocks (NCHW tensors, like the reference)
# --------------------------------------------------------------------------------------------------------------
def ln_channels(x, eps=1e-6):
    """LayerNorm2d without affine (reference src/modules.py:22-27)."""
    return F.layer_norm(x.permute(0, 2, 3, 1), (x.size(1),), None, None, eps).permute(0, 3, 1, 2)


def grn(x, gamma, beta):
    """GlobalResponseNorm on NHWC (reference src/modules.py:37-40)."""
    gx = torch.norm(x, p=2, dim=(1, 2), keepdim=True)
    nx = gx / (gx.mean(dim=-1, keepdim=True) + 1e-6)
    return gamma * (x * nx) + beta + x


def mlp(sd, p, x_nhwc):
    """channelwise Sequential: Linear, GELU, GRN, Dropout(eval), Linear (reference src/modules.py:48-54)."""
    h = F.linear(x_nhwc, sd[p + ".channelwise.0.weight"], sd[p + ".channelwise.0.bias"])
    h = F.gelu(h)
    h = grn(h, sd[p + ".channelwise.2.gamma"], sd[p + ".channelwise.2.beta"])
    return F.linear(h, sd[p + ".channelwise.4.weight"], sd[p + ".channelwise.4.bias"])


def res_block(sd, p, x, skip=None):
    """ResBlock (reference src/modules.py:55-62): depthwise (grouped over cat([x, skip])) -> LN -> MLP -> + x."""
    w = sd[p + ".depthwise.weight"]
    inp = x if skip is None else torch.cat([x, skip], dim=1)
    h = F.conv2d(inp, w, sd[p + ".depthwise.bias"], padding=w.size(-1) // 2, groups=x.size(1))
    h = ln_channels(h).permute(0, 2, 3, 1)
    return x + mlp(sd, p, h).permute(0, 3, 1, 2)


def ff_block(sd, p, x):
    """FeedForwardBlock (reference src/modules.py:94-96)."""
    return x + mlp(sd, p, ln_channels(x).permute(0, 2, 3, 1)).permute(0, 3, 1, 2)


def timestep_block(sd, p, x, r_embed):
    """TimestepBlock (reference src/modules.py:104-106)."""
    ab = F.linear(r_embed, sd[p + ".mapper.weight"], sd[p + ".mapper.bias"])
    a, b = ab[:, :, None, None].chunk(2, dim=1)
    return x * (1 + a) + b


def mha(sd, p, q_in, kv_in, nhead, attn_weights=None):
    """nn.MultiheadAttention(batch_first, bias) written out as utils/alter_attention.py:15-36 does."""
    w = sd[p + ".in_proj_weight"].chunk(3, dim=0)
    b = sd[p + ".in_proj_bias"].chunk(3, dim=0)
    B, Lq, C = q_in.shape
    Lk = kv_in.size(1)
    q = F.linear(q_in, w[0], b[0]).view(B, Lq, nhead, -1).permute(0, 2, 1, 3)
    k = F.linear(kv_in, w[1], b[1]).view(B, Lk, nhead, -1).permute(0, 2, 1, 3)
    v = F.linear(kv_in, w[2], b[2]).view(B, Lk, nhead, -1).permute(0, 2, 1, 3)
    att = ((q @ k.transpose(-2, -1)) / (q.size(-1) ** 0.5)).softmax(dim=-1)
    if attn_weights is not None:  # post-softmax re-weighting of the last n key columns, no renormalisation
        wts = torch.ones(Lq, Lk, dtype=att.dtype)
        wts[:, -attn_weights.numel():] = attn_weights.to(att.dtype)
        att = att * wts
    o = (att @ v).permute(0, 2, 1, 3).reshape(B, Lq, C)
    return F.linear(o, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def attn_block(sd, p, x, c_embed, nhead, self_attn=True, attn_weights=None):
    """AttnBlock + Attention2D (reference src/modules.py:12-19, 76-79)."""
    kv = F.linear(F.silu(c_embed), sd[p + ".kv_mapper.1.weight"], sd[p + ".kv_mapper.1.bias"])
    xn = ln_channels(x)
    B, C, H, W = x.shape
    q = xn.reshape(B, C, H * W).permute(0, 2, 1)
    keys = torch.cat([q, kv], dim=1) if self_attn else kv
    o = mha(sd, p + ".attention.attn", q, keys, nhead, attn_weights)
    return x + o.permute(0, 2, 1).reshape(B, C, H, W)


def r_embedding(r, c_r, max_positions=10000):
    """Paella.gen_r_embedding (reference src/modules.py:212-221)."""
    r = r * max_positions
    half = c_r // 2
    f = math.log(max_positions) / (half - 1)
    f = torch.arange(half).float().mul(-f).exp().to(r.dtype)
    e = r[:, None] * f[None, :]
    e = torch.cat([e.sin(), e.cos()], dim=1)
    if c_r % 2 == 1:
        e = F.pad(e, (0, 1))
    return e


def c_embeddings(sd, cfg, byt5, clip=None, clip_image=None):
    """Paella.gen_c_embeddings (reference src/modules.py:223-232; list clip_image: utils/modules.py:229-235)."""
    cc = cfg["c_cond"]
    seq = F.linear(byt5, sd["byt5_mapper.weight"], sd["byt5_mapper.bias"])
    if clip is not None:
        seq = torch.cat([seq, F.linear(clip, sd["clip_mapper.weight"], sd["clip_mapper.bias"]).view(clip.size(0), -1, cc)], dim=1)
    if clip_image is not None:
        for ci in (clip_image if isinstance(clip_image, (list, tuple)) else [clip_image]):
            seq = torch.cat([seq, F.linear(ci, sd["clip_image_mapper.weight"], sd["clip_image_mapper.bias"]).view(ci.size(0), -1, cc)], dim=1)
    return F.layer_norm(seq, (cc,), None, None, 1e-6)


def _level_blocks(cfg, prefix, i, start):
    j = start
    for rep in range(cfg["blocks"][i]):
        for k, t in enumerate(cfg["level_config"][i]):
            yield t, f"{prefix}.{j}", rep, k
            j += 1


def unet_forward(sd, cfg, x, r, byt5, clip=None, clip_image=None, x_cat=None, attn_weights=None, dtype=torch.float32, taps=None):
    """Paella.forward (reference src/modules.py:263-275 with _down_encode :234-247 and _up_decode :249-261).
    `sd` uses the reference's state-dict keys; `taps` (dict) optionally collects intermediate activations."""
    sd = {k: v.to(dtype) if v.is_floating_point() else v for k, v in sd.items()}
    byt5 = byt5.to(dtype)
    clip = None if clip is None else clip.to(dtype)
    if clip_image is not None:
        clip_image = [c.to(dtype) for c in clip_image] if isinstance(clip_image, (list, tuple)) else clip_image.to(dtype)
    if x_cat is not None:
        x = torch.cat([x, x_cat], dim=1)
    n = len(cfg["c_hidden"])
    p = cfg["patch_size"]
    r_embed = r_embedding(r.to(dtype), cfg["c_r"])
    c_embed = c_embeddings(sd, cfg, byt5, clip, clip_image)
    if taps is not None:
        taps["r_embed"], taps["c_embed"] = r_embed, c_embed
    h = F.layer_norm(F.embedding(x, sd["in_mapper.0.weight"]), (cfg["c_in"],), None, None, 1e-6).permute(0, 3, 1, 2)
    h = F.pixel_unshuffle(h, p)
    h = ln_channels(F.conv2d(h, sd["embedding.1.weight"], sd["embedding.1.bias"]))
    if taps is not None:
        taps["embedding"] = h

    def run(t, pfx, h, i, skip):
        if t == 'C':
            return res_block(sd, pfx, h, skip)
        if t == 'A':
            return attn_block(sd, pfx, h, c_embed, cfg["nhead"][i], cfg.get("self_attn", True), attn_weights)
        if t == 'T':
            return timestep_block(sd, pfx, h, r_embed)
        if t == 'F':
            return ff_block(sd, pfx, h)
        raise ValueError(t)

    outs = []
    for i in range(n):
        start = 0
        if i > 0:
            q = f"down_blocks.{i}.0.1"
            h = F.conv2d(ln_channels(h), sd[q + ".weight"], sd[q + ".bias"], stride=2)
            start = 1
        for t, pfx, _, _ in _level_blocks(cfg, f"down_blocks.{i}", i, start):
            h = run(t, pfx, h, i, None)
        outs.insert(0, h)
        if taps is not None:
            taps[f"down{i}"] = h
    h = outs[0]
    for u in range(n):
        i = n - 1 - u
        j = 0
        for t, pfx, rep, k in _level_blocks(cfg, f"up_blocks.{u}", i, 0):
            skip = outs[u] if (t == 'C' and rep == 0 and k == 0 and u > 0) else None
            h = run(t, pfx, h, i, skip)
            j += 1
        if i > 0:
            q = f"up_blocks.{u}.{j}.1"
            h = F.conv_transpose2d(ln_channels(h), sd[q + ".weight"], sd[q + ".bias"], stride=2)
        if taps is not None:
            taps[f"up{i}"] = h
    h = F.pixel_shuffle(F.conv2d(ln_channels(h), sd["clf.1.weight"], sd["clf.1.bias"]), p)
    return F.conv2d(ln_channels(h), sd["out_mapper.1.weight"])


def add_noise(x, t, num_labels, mask=None, random_x=None, rand_u=None):
    """Paella.add_noise (reference src/modules.py:277-283); rand_u / random_x may be supplied for determinism."""
    if mask is None:
        if rand_u is None:
            rand_u = torch.rand_like(x.float())
        mask = (rand_u <= t[:, None, None]).long()
    if random_x is None:
        random_x = torch.randint_like(x, 0, num_labels)
    return x * (1 - mask) + random_x * mask, mask


def sample_tail(logits_c, logits_u, cfg, omc, temperature, noise_q=None, mode=0):
    """reference src/utils.py:45-50.  logits [B, L, H, W]; noise_q [B*H*W, L] ~ Exp(1) makes the categorical draw
    explicit: torch.multinomial(p, 1) == argmax(p / q) (ATen multinomial, n_sample == 1 path)."""
    l = logits_c if logits_u is None else logits_c * cfg + logits_u * omc
    if mode == 1:
        return l.argmax(dim=1)
    scores = l.div(temperature).softmax(dim=1)
    flat = scores.permute(0, 2, 3, 1).reshape(-1, l.size(1))
    tok = torch.multinomial(flat, 1)[:, 0] if noise_q is None else (flat / noise_q).argmax(dim=-1)
    return tok.view(l.size(0), *l.shape[2:])


def sample(forward_fn, num_labels, model_inputs, unconditional_inputs, latent_shape, init_x=None, steps=12, renoise_steps=11,
           temperatures=None, cfgs=None, t_list=None, noise=None, argmax=False):
    """The sampling loop of reference src/utils.py:35-55 / src_distributed/utils.py:97-126 with explicit noise.
    forward_fn(tokens, r, **inputs) -> logits [B, L, H, W].  noise = dict(init_noise, q=[per step], u=[per step])."""
    B = latent_shape[0]
    init_noise = noise["init_noise"]
    sampled = init_noise.clone() if init_x is None else init_x
    traj = []
    for i in range(steps):
        r = torch.ones(B) * t_list[i]
        lc = forward_fn(sampled, r, **model_inputs)
        lu = None
        cfg = omc = None
        if cfgs[i] is not None:
            lu = forward_fn(sampled, r, **unconditional_inputs)
            cfg, omc = cfgs[i]
        if argmax:
            sampled = sample_tail(lc, lu, cfg, omc, 1.0, mode=1)
        else:
            sampled = sample_tail(lc, lu, cfg, omc, temperatures[i], noise_q=noise["q"][i])
        if i < renoise_steps:
            t_next = torch.ones(B) * t_list[i + 1]
            sampled, _ = add_noise(sampled, t_next, num_labels, random_x=init_noise, rand_u=noise["u"][i])
        traj.append(sampled)
    return sampled, traj


# --------------------------------------------------------------------------------------------------------------
# VQGAN (reference src/vqgan.py)
# --------------------------------------------------------------------------------------------------------------
def vector_quantize(x_rows, codebook):
    """torchtools.nn.VectorQuantize nearest-code search (third party, UNPINNED -- see module docstring):
    dist = (|e|^2 + |x|^2) - 2 x e^T via addmm; indices = dist.min(dim=1); returns (codebook rows, indices)."""
    cb_sqr = torch.sum(codebook ** 2, dim=1)
    x_sqr = torch.sum(x_rows ** 2, dim=1, keepdim=True)
    dist = torch.addmm(cb_sqr + x_sqr, x_rows, codebook.t(), alpha=-2.0, beta=1.0)
    idx = dist.min(dim=1)[1]
    return codebook[idx], idx


def vq_res_block(sd, p, x):
    """vqgan.ResBlock.forward (reference src/vqgan.py:34-42)."""
    g = sd[p + ".gammas"]
    c = x.size(1)
    xt = F.layer_norm(x.permute(0, 2, 3, 1), (c,), None, None, 1e-6).permute(0, 3, 1, 2) * (1 + g[0]) + g[1]
    x = x + F.conv2d(F.pad(xt, (1, 1, 1, 1), mode="replicate"), sd[p + ".depthwise.1.weight"], sd[p + ".depthwise.1.bias"], groups=c) * g[2]
    xt = F.layer_norm(x.permute(0, 2, 3, 1), (c,), None, None, 1e-6) * (1 + g[3]) + g[4]
    h = F.linear(F.gelu(F.linear(xt, sd[p + ".channelwise.0.weight"], sd[p + ".channelwise.0.bias"])),
                 sd[p + ".channelwise.2.weight"], sd[p + ".channelwise.2.bias"])
    return x + h.permute(0, 3, 1, 2) * g[5]


def _vq_layout(cfg):
    L = cfg["levels"]
    down, j = [], 0
    for i in range(L):
        if i > 0:
            down.append(("conv", f"down_blocks.{j}")); j += 1
        down.append(("res", f"down_blocks.{j}")); j += 1
    down.append(("latent", f"down_blocks.{j}"))
    up, j = [("conv1", "up_blocks.0.0")], 1
    for i in range(L):
        for _ in range(cfg["bottleneck_blocks"] if i == 0 else 1):
            up.append(("res", f"up_blocks.{j}")); j += 1
        if i < L - 1:
            up.append(("convT", f"up_blocks.{j}")); j += 1
    return down, up


def vq_encode(sd, cfg, img):
    """VQModel.encode (reference src/vqgan.py:91-95) with eval-mode BatchNorm."""
    down, _ = _vq_layout(cfg)
    x = F.conv2d(F.pixel_unshuffle(img, 2), sd["in_block.1.weight"], sd["in_block.1.bias"])
    for kind, p in down:
        if kind == "conv":
            x = F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=2, padding=1)
        elif kind == "res":
            x = vq_res_block(sd, p, x)
        else:
            x = F.conv2d(x, sd[p + ".0.weight"])
            x = F.batch_norm(x, sd[p + ".1.running_mean"], sd[p + ".1.running_var"], sd[p + ".1.weight"], sd[p + ".1.bias"], False, 0.1, 1e-5)
    rows = x.permute(0, 2, 3, 1).reshape(-1, x.size(1))
    q, idx = vector_quantize(rows, sd["vquantizer.codebook.weight"])
    qe = q.view(x.size(0), x.size(2), x.size(3), -1).permute(0, 3, 1, 2)
    mse = (q - rows).pow(2).mean()
    sf = cfg["scale_factor"]
    return qe / sf, x / sf, idx.view(x.size(0), x.size(2), x.size(3)), mse + mse * 0.25


def _vq_decoder(sd, cfg, x):
    _, up = _vq_layout(cfg)
    for kind, p in up:
        if kind == "conv1":
            x = F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"])
        elif kind == "res":
            x = vq_res_block(sd, p, x)
        else:
            x = F.conv_transpose2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=2, padding=1)
    return F.pixel_shuffle(F.conv2d(x, sd["out_block.0.weight"], sd["out_block.0.bias"]), 2)


def vq_decode(sd, cfg, latents):
    """VQModel.decode (reference src/vqgan.py:97-101)."""
    return _vq_decoder(sd, cfg, latents * cfg["scale_factor"])


def vq_decode_indices(sd, cfg, idx):
    """VQModel.decode_indices (reference src/vqgan.py:103-107); idx2vq = embedding lookup moved to dim 1."""
    return _vq_decoder(sd, cfg, sd["vquantizer.codebook.weight"][idx].permute(0, 3, 1, 2))


def replay_torch_noise(seed, latent_shape, num_labels, steps, renoise_steps, categorical_steps=None):
    """Re-draw, from a seeded CPU generator, exactly the random numbers the reference's sample() consumes, in its
    order: randint for the start tokens (src/utils.py:37), then per step the Exp(1) tensor torch.multinomial draws
    internally (:50) and the U[0,1) tensor of add_noise's rand_like (:54 -> src/modules.py:279)."""
    B, H, W = latent_shape
    g = torch.Generator(device="cpu").manual_seed(seed)
    noise = {"init_noise": torch.randint(0, num_labels, size=(B, H, W), generator=g), "q": [], "u": []}
    def draw_q():
        # multinomial draws q = empty_like(p).exponential_(1) in p's MEMORY order.  p is
        # scores.permute(0,2,3,1).reshape(-1, L) (src/utils.py:49): for B > 1 the reshape copies (row-major), for
        # B == 1 it is a column-major VIEW of the NCHW softmax output, so q[row, col] sits at memory col*rows+row.
        if B == 1:
            return torch.empty(num_labels, H * W).exponential_(1, generator=g).t()
        return torch.empty(B * H * W, num_labels).exponential_(1, generator=g)

    for i in range(steps):
        cat = categorical_steps is None or categorical_steps[i]
        noise["q"].append(draw_q() if cat else None)
        noise["u"].append(torch.rand(B, H, W, generator=g) if i < renoise_steps else None)
    return noise
