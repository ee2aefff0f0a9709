"""Batch-sharded sampling across the GPUs of one node (one process per GPU, torch.distributed over RCCL/xGMI).

Every op of the sampling path is per-sample (LayerNorm per position, GRN and attention within a sample, eval-mode
BatchNorm), so the batch axis shards with no data-path collective.  The only exchange is ONE broadcast of the
frozen conditioning (byt5 / clip / clip_image of the conditional and unconditional sets) from the rank that ran
the text/image encoders, issued before step 0 -- packed into a single flat buffer so it is one RCCL call whose
root fans out over all xGMI links.  The reference samples on a single device (sample() has no collective,
src_distributed/utils.py:97-126); its only collectives are DDP's, for training (src_distributed/train.py:54).

The functions here are plain host logic over torch.distributed and work with the gloo backend on CPU tensors
(that is how tests/test_dist.py covers world_size 2); compute always goes through the HIP path.
"""
import torch
import torch.distributed as dist


def shard_bounds(B, rank, world_size):
    """Rows [lo, hi) of a batch of B owned by `rank` (contiguous, sizes differ by at most one)."""
    base, rem = divmod(B, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _flatten_inputs(inputs):
    """dict(byt5, clip, clip_image[list]) -> (list of tensors, structure descriptor)"""
    tensors, desc = [], []
    for k in ("byt5", "clip", "clip_image"):
        v = inputs.get(k) if inputs is not None else None
        if v is None:
            desc.append((k, None))
        elif isinstance(v, (list, tuple)):
            desc.append((k, [tuple(t.shape) for t in v]))
            tensors.extend(v)
        else:
            desc.append((k, tuple(v.shape)))
            tensors.append(v)
    return tensors, desc


def _unflatten(flat, desc, device):
    out, off = {}, 0

    def take(shape):
        nonlocal off
        n = 1
        for d in shape:
            n *= d
        t = flat[off:off + n].view(shape)
        off += n
        return t

    for k, d in desc:
        if d is None:
            out[k] = None
        elif isinstance(d, list):
            out[k] = [take(s) for s in d]
        else:
            out[k] = take(d)
    return out


def conditioning_layout(input_sets):
    """(structure descriptors, total element count) of a list of conditioning dicts.  Every rank that knows the shapes
    (fixed-shape serving: same batch, sequence lengths and widths per request) can compute it locally from same-shaped
    tensors and pass it as `layout=` to broadcast_conditioning, which then needs no shape exchange."""
    descs, numel = [], 0
    for s in input_sets:
        ts, d = _flatten_inputs(s)
        descs.append(d)
        numel += sum(t.numel() for t in ts)
    return descs, numel


def cond_spec_layout(model, B, S_byt5=0, S_byt5_uncond=None, clip=True, n_clip_image=0, sets=2):
    """The layout of [model_inputs, unconditional_inputs] derived from what every rank knows WITHOUT seeing the tensors: the model's
    embedding widths, the global batch, the ByT5 sequence lengths and which CLIP inputs are present (fixed-shape serving).  Pass the
    result as `layout=` to sample_sharded / broadcast_conditioning: the conditioning then travels in exactly one collective."""
    cfg = model._cfg
    out, numel = [], 0
    for i in range(sets):
        S = S_byt5 if (i == 0 or S_byt5_uncond is None) else S_byt5_uncond
        d = [("byt5", (B, S, cfg["byt5_embd"]))]
        numel += B * S * cfg["byt5_embd"]
        d.append(("clip", (B, cfg["clip_embd"]) if clip else None))
        numel += B * cfg["clip_embd"] if clip else 0
        if n_clip_image == 0:
            d.append(("clip_image", None))
        elif n_clip_image == 1:
            d.append(("clip_image", (B, cfg["clip_embd"])))
        else:
            d.append(("clip_image", [(B, cfg["clip_embd"])] * n_clip_image))
        numel += n_clip_image * B * cfg["clip_embd"]
        out.append(d)
    return out, numel


def _pack_seed(seed, device):
    """An int64 seed as two fp32 words (bit pattern preserved): it rides at the end of the flat conditioning buffer."""
    return torch.tensor([int(seed)], dtype=torch.int64).view(torch.float32).to(device)


def broadcast_conditioning(input_sets, src=0, device=None, group=None, layout=None, seed=None, with_seed=False, seed_on_device=False, validate=None,
                           return_flag=False):
    """Broadcast a list of conditioning dicts (e.g. [model_inputs, unconditional_inputs]) from `src` with ONE
    tensor collective.  Non-source ranks pass None.  Without `layout` the shapes travel first in one small object
    broadcast (which synchronises host and device; a receiver cannot size its buffer otherwise); with `layout` =
    conditioning_layout(...) / cond_spec_layout(...) known on every rank the call is a single asynchronous RCCL broadcast on
    the current stream.  with_seed=True appends the source's Philox seed (`seed`, or a fresh one) to the SAME buffer and
    returns (sets, seed): seed agreement costs no collective of its own.  seed_on_device=True returns the seed as a 1-element
    int64 DEVICE tensor (pass it as sample(..., seed=0, seed_dev=t)): no host synchronisation at all on the layout path;
    otherwise the seed is read back with .item() (one host-device sync per call).
    A conditioning that does not match the agreed `layout` can only be detected on the source.  It never enters the collective
    with a wrong-sized buffer (the other ranks would block forever): the source sends a buffer of the AGREED size whose trailing
    flag word is 0 and whose payload is NaN, then raises ValueError; receivers raise too when validate=True (default on the
    handshake path, which synchronises anyway; on the layout path it costs one host sync, so it defaults to False there and a
    receiver of a poisoned buffer computes NaN-conditioned output instead of hanging).
    return_flag=True is the form for callers that issue FURTHER collectives (sample_sharded): nothing is raised here, not even on the
    source, and the result carries (ok_dev, src_bad) at its end -- ok_dev = the flag word as a 1-element fp32 DEVICE tensor (1 = valid,
    0 = poisoned; consume it on the device, e.g. to poison the outputs with -1, without a host sync), src_bad = the source's host-side
    knowledge of the mismatch (always False on receivers) -- so that every rank can keep its collective sequence and fail together."""
    rank = dist.get_rank(group)
    meta = [None]
    flat = None
    bad = False
    if validate is None:
        validate = layout is None
    if rank == src:
        all_tensors, descs = [], []
        for s in input_sets:
            ts, d = _flatten_inputs(s)
            all_tensors.extend(ts)
            descs.append(d)
        device = all_tensors[0].device if device is None else torch.device(device)
        flat = torch.cat([t.reshape(-1).float() for t in all_tensors]).to(device) if all_tensors else torch.zeros(0, device=device)
        meta = [(descs, flat.numel())]
        if layout is not None and (layout[1] != flat.numel() or [list(map(tuple, d)) for d in layout[0]] != [list(map(tuple, d)) for d in descs]):
            bad = True  # keep the collective well-formed: agreed size, poisoned payload, flag 0
            flat = torch.full((layout[1],), float("nan"), dtype=torch.float32, device=device)
        if with_seed:
            if seed is None:
                from .sampling import fresh_seed
                seed = fresh_seed()
            flat = torch.cat([flat, _pack_seed(seed, flat.device)])
        flat = torch.cat([flat, torch.full((1,), 0.0 if bad else 1.0, dtype=torch.float32, device=flat.device)])  # (a fill kernel, not a host-to-device copy)
    if layout is None:
        dist.broadcast_object_list(meta, src=src, group=group)
        descs, numel = meta[0]
    else:
        descs, numel = layout
    extra = (2 if with_seed else 0) + 1
    if rank != src:
        if device is None:
            raise ValueError("non-source ranks must pass device")
        flat = torch.empty(numel + extra, dtype=torch.float32, device=device)
    dist.broadcast(flat, src=src, group=group)
    ok_dev = flat[numel + extra - 1:numel + extra]
    if not return_flag:
        if bad:
            raise ValueError("conditioning does not match the agreed layout (a poisoned buffer of the agreed size was broadcast so that no rank blocks)")
        if validate and float(ok_dev.item()) != 1.0:
            raise ValueError("the source rank's conditioning did not match the agreed layout")
    if with_seed:
        seed_t = flat[numel:numel + 2].clone().view(torch.int64)  # 1-element int64 tensor on the buffer's device
        seed = seed_t if seed_on_device else int(seed_t.item())
    flat = flat[:numel]
    out, off = [], 0
    for d in descs:
        n = 0
        for _, shape in d:
            for s in ([shape] if isinstance(shape, tuple) else (shape or [])):
                m = 1
                for x in s:
                    m *= x
                n += m
        out.append(_unflatten(flat[off:off + n], d, flat.device))
        off += n
    res = (out, seed) if with_seed else out
    if return_flag:
        return (res + (ok_dev, bad)) if with_seed else (res, ok_dev, bad)
    return res


def shard_inputs(inputs, lo, hi):
    """Slice every conditioning tensor to this rank's rows."""
    if inputs is None:
        return None
    out = {}
    for k, v in inputs.items():
        if v is None:
            out[k] = None
        elif isinstance(v, (list, tuple)):
            out[k] = [t[lo:hi].contiguous() for t in v]
        else:
            out[k] = v[lo:hi].contiguous()
    return out


def agree_on_seed(seed, src=0, device=None, group=None):
    """Every rank must key its counter-based noise with the SAME seed: `seed` if given (then all ranks must pass the same
    value), otherwise a fresh one drawn on `src` and broadcast (one 8-byte tensor)."""
    if seed is not None:
        return int(seed)
    from .sampling import fresh_seed
    backend_cpu = dist.get_backend(group) == "gloo"
    t = torch.zeros(1, dtype=torch.int64, device="cpu" if backend_cpu or device is None else device)
    if dist.get_rank(group) == src:
        t[0] = fresh_seed()
    dist.broadcast(t, src=src, group=group)
    return int(t.item())


def sample_sharded(model, model_inputs, unconditional_inputs, latent_shape, src=0, group=None, gather=False, noise="philox", seed=None,
                   layout=None, **kwargs):
    """Batch-sharded `sample`: broadcast the conditioning once, sample this rank's rows, optionally all_gather.
    `model_inputs` / `unconditional_inputs` are needed on rank `src` only.  kwargs go to paella_amd.sample.
    Collectives: with `layout=` (conditioning_layout / cond_spec_layout, computable on every rank) exactly ONE -- the packed
    broadcast, which also carries the Philox seed; without it one small shape handshake precedes that broadcast (a receiver
    cannot size its buffer otherwise).  Never a separate seed collective.
    With the counter-based noise (default here) every random number -- start tokens, categorical draws, renoise mask -- is
    keyed by (seed, GLOBAL row, step), so the concatenation of the shards equals the unsharded `sample(..., noise="philox",
    seed=seed)` bit for bit, whatever the world size.  noise="torch" consumes each rank's own torch generator
    (no cross-rank equivalence; the reference has none either).
    FAILURE CONTRACT (a source-side conditioning that does not match the agreed `layout`; detectable on the source's host only): the broadcast stays
    well-formed (agreed size, NaN payload, flag word 0) and every rank stays in its collective sequence.  gather=True: every rank raises ValueError after
    the all_gather.  gather=False: the SOURCE raises ValueError; every RECEIVER returns a token grid filled with -1 (set on the device from the flag word,
    no host synchronisation) -- callers on that path must treat negative tokens as "request failed" before decoding them.  Receivers still run their
    sampling pass on the NaN conditioning (skipping it would need a host read of the flag on every healthy call); its tokens stay inside [0, num_labels)
    -- the tail's argmax keeps the first label on NaN scores -- so the next step's embedding gather never indexes out of range."""
    from . import sampling as S  # (looked up at call time: tests/test_dist.py substitutes CPU stand-ins for the two HIP entry points)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    device = next(model.parameters()).device
    philox = noise == "philox"
    res = broadcast_conditioning([model_inputs, unconditional_inputs] if rank == src else None, src=src, device=device, group=group,
                                 layout=layout, seed=seed, with_seed=philox, seed_on_device=philox, return_flag=True)
    # the seed and the validity flag stay on the device: no host sync between the broadcast and the sampler
    ((cond, uncond), seed_dev, ok_dev, src_bad) = res if philox else (res[0], None, res[1], res[2])
    B, H, W = latent_shape
    lo, hi = shard_bounds(B, rank, world)
    shard = (lo, B) if philox else None
    local = None
    # A conditioning that does not match the agreed layout is known on the source's host and reaches the other ranks as the flag word of the
    # (well-formed, NaN-filled) broadcast.  The failure is COLLECTIVE: nobody raises before the gather, the source contributes -1 tokens without
    # sampling, every receiver's tokens are replaced by -1 ON THE DEVICE (one select on the flag word, no host sync), and every rank raises after
    # the gather -- no rank is left waiting in a collective the others never enter.
    if hi > lo:
        if src_bad:
            local = torch.full((hi - lo, H, W), -1, dtype=torch.int64, device=device)
        else:
            local = S.sample(model, shard_inputs(cond, lo, hi), (hi - lo, H, W), unconditional_inputs=shard_inputs(uncond, lo, hi),
                             device=device, noise=noise, seed=0 if philox else seed, seed_dev=seed_dev, shard=shard, **kwargs)
            local = S.select_tokens(local, flag=ok_dev, fill=-1)  # one HIP kernel on the device flag word: tokens, or -1 everywhere when the flag is 0
    if not gather:
        if src_bad:
            raise ValueError("conditioning does not match the agreed layout (a poisoned buffer of the agreed size was broadcast: the other ranks return -1 tokens)")
        return local
    sizes = [shard_bounds(B, r, world) for r in range(world)]
    mx = max(h - l for l, h in sizes)
    pad = torch.zeros(mx, H, W, dtype=torch.int64, device=device)
    if local is not None:
        pad[:hi - lo] = local
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad, group=group)
    out = torch.cat([o[:h - l] for o, (l, h) in zip(outs, sizes)], dim=0)
    if src_bad or (layout is not None and out.numel() and int(out.min()) < 0):  # (the gathered tokens are about to be consumed by the host anyway)
        raise ValueError("the source rank's conditioning did not match the agreed layout: every rank received poisoned (-1) tokens")
    return out
