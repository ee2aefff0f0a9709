"""Paella's iterative token-denoising sampler on MI355X.

Host mirror of the reference's two `sample()` functions:
  * `sample`             -- src/utils.py:35-55          (model, model_inputs, latent_shape, unconditional_inputs=None, ...)
  * `sample_distributed` -- src_distributed/utils.py:97-126 (model, model_inputs, unconditional_inputs, latent_shape, init_x=None, ...)
Both drive the same core loop.  The loop itself is host code (a dozen lines, as in the reference); everything
it calls is a HIP kernel: `Paella.prepare_cond` once per call (step-invariant conditioning work hoisted out of
the loop), `Paella.forward_prepared` per step with the conditional and unconditional rows batched into one
2B-row evaluation when their conditioning shapes agree (the reference runs two forwards, src/utils.py:44,46),
and one fused tail kernel per step for CFG mix -> temperature -> softmax -> categorical draw -> renoise.

Noise: with `noise="torch"` (default) the random numbers are drawn from torch's generator with the same calls in
the same order as the reference (randint for the start tokens; per step `exponential_` for the multinomial draw
-- torch.multinomial(p, 1) is argmax(p / Exp(1)) -- and `rand` for the renoise mask), so under the same seed the
result equals the reference's whenever the logits agree.  `noise="philox"` generates all per-step noise inside
the tail kernel (Philox4x32-10 keyed by `seed`), which removes the [B*H*W, num_labels] noise tensor.

Extension (not reference behaviour): a step temperature of 0 selects argmax of the mixed logits.
"""
import torch

from . import _lib
from .modules import Paella


def linspace_schedule(start, end, n):
    """torch.linspace(start, end, n) as python floats of the fp32 values (what the reference multiplies by)."""
    return [float(v) for v in torch.linspace(start, end, n)]


def _same_cond_layout(a, b):
    if a is None or b is None:
        return False
    for k in ("byt5", "clip", "clip_image"):
        va, vb = a.get(k), b.get(k)
        if (va is None) != (vb is None):
            return False
        if va is None:
            continue
        if isinstance(va, (list, tuple)) != isinstance(vb, (list, tuple)):
            return False
        if isinstance(va, (list, tuple)):
            if len(va) != len(vb) or any(x.shape != y.shape for x, y in zip(va, vb)):
                return False
        elif va.shape != vb.shape:
            return False
    return True


def _cond_seq_len(model, inputs):
    """Conditioning rows per sample: S_byt5 + clip_seq_len * [clip] + clip_seq_len * #clip_image (src/modules.py:223-232)."""
    if inputs is None:
        return 0
    n = model._cfg["clip_seq_len"]
    S = inputs["byt5"].size(1) if inputs.get("byt5") is not None else 0
    if inputs.get("clip") is not None:
        S += n
    ci = inputs.get("clip_image")
    if ci is not None:
        S += n * (len(ci) if isinstance(ci, (list, tuple)) else 1)
    return S


def _cat_inputs(a, b):
    out = {}
    for k in ("byt5", "clip", "clip_image"):
        va, vb = a.get(k), b.get(k)
        if va is None:
            out[k] = None
        elif isinstance(va, (list, tuple)):
            out[k] = [torch.cat([x, y], dim=0) for x, y in zip(va, vb)]
        else:
            out[k] = torch.cat([va, vb], dim=0)
    return out


def _tail(logits_c, logits_u, rows, L, cfg, omc, temperature, mode, noise_q, seed, offset, init_noise, mask_u, t_next, out,
          seed_dev=None, row_offset=0, row_offset_dev=None):
    lib = _lib.load()
    dev = logits_c.device
    with torch.cuda.device(dev):
        _lib.check(lib.paella_sample_tail_ex(_lib.ptr(logits_c), _lib.ptr(logits_u), rows, L, cfg, omc, temperature, mode,
                                             _lib.ptr(noise_q), seed, _lib.ptr(seed_dev), offset, row_offset, _lib.ptr(row_offset_dev),
                                             _lib.ptr(init_noise), _lib.ptr(mask_u), t_next, _lib.ptr(out), None, _lib.stream_ptr(dev)))


def fresh_seed():
    """A 62-bit seed drawn from torch's default (CPU) generator: reproducible under torch.manual_seed, different on every
    call otherwise -- what `seed=None` means in the counter-based (Philox) noise mode.  NOTE: this consumes one draw of torch's
    global CPU generator per call (the reference's sample() consumes the global DEVICE generator instead, src/utils.py:37)."""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


def start_tokens(num_labels, shape, seed, device, shard=None, out=None, seed_dev=None, row_offset_dev=None):
    """The start tokens of the counter-based noise mode (the reference draws torch.randint on the global generator,
    src/utils.py:37 -- a stream that cannot be sharded): token i of the GLOBAL [total, H, W] grid is Philox(seed, i) mod
    num_labels (paella_start_tokens), so a batch shard draws exactly its rows of the unsharded draw at O(shard) cost.
    seed_dev / row_offset_dev: optional device-resident words added to seed / the row offset (a captured graph draws fresh
    tokens for any seed and shard without host work)."""
    B, H, W = shape
    lo = 0 if shard is None else int(shard[0])
    device = torch.device(device)
    if out is None:
        out = torch.empty(B, H, W, dtype=torch.int64, device=device)
    with torch.cuda.device(device):
        _lib.check(_lib.load().paella_start_tokens(int(seed), _lib.ptr(seed_dev), lo * H * W, _lib.ptr(row_offset_dev), int(num_labels),
                                                   B * H * W, _lib.ptr(out), _lib.stream_ptr(device)))
    return out


def timestep_table(t_list, steps, B, device):
    """[steps, B] fp32: row i = the timestep of step i for every sample (one upload instead of a fill kernel per step)."""
    return torch.tensor([float(v) for v in t_list[:steps]], dtype=torch.float32).to(device)[:, None].repeat(1, B).contiguous()


def select_tokens(a, b=None, mask=None, flag=None, fill=-1, out=None):
    """out[i] = keep(i) ? a[i] : (b[i] if b is given else fill), keep(i) = (mask is None or mask[i] != 0) and (flag is None or flag[0] == 1.0) -- int64 token
    grids, `flag` a 1-element fp32 DEVICE tensor (paella_select_tokens: one HIP kernel, no host synchronisation, graph-capturable).  Used by the inpainting
    wrapper (re-impose the known tokens) and by the batch-sharded sampler (-1 tokens when the conditioning broadcast was flagged invalid)."""
    dev = a.device
    if dev.type != "cuda" or a.dtype != torch.int64:
        raise ValueError("select_tokens needs int64 HIP tensors")
    a = a.contiguous()
    for name, t in (("b", b), ("mask", mask)):
        if t is not None and (t.dtype != torch.int64 or t.shape != a.shape or t.device != dev):
            raise ValueError("select_tokens: %s must be an int64 tensor of a's shape on a's device" % name)
    if flag is not None and (flag.dtype != torch.float32 or flag.numel() != 1 or flag.device != dev):
        raise ValueError("select_tokens: flag must be a 1-element fp32 tensor on a's device")
    if out is None:
        out = torch.empty_like(a)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().paella_select_tokens(_lib.ptr(a), _lib.ptr(None if b is None else b.contiguous()), _lib.ptr(None if mask is None else mask.contiguous()),
                                                    _lib.ptr(flag), int(fill), a.numel(), _lib.ptr(out), _lib.stream_ptr(dev)))
    return out


def _sample_core(model, model_inputs, unconditional_inputs, latent_shape, init_x, steps, renoise_steps, t_list, temperatures,
                 cfgs, device, noise="torch", seed=None, attn_weights=None, seed_dev=None, init_noise_buf=None, r_all=None, shard=None,
                 ws=None, fused_tail=True, row_offset_dev=None):
    """cfgs: per-step list of (cfg_fp32, one_minus_cfg_fp32) or None (no guidance at that step).
    seed_dev / row_offset_dev / init_noise_buf / r_all: device-resident seed and row-offset words, a buffer for the start tokens
    and the [steps, B] timestep table (HIP-graph capture cannot upload from the host, see GraphSampler); ws: caller-owned
    workspace for every library call.
    shard = (lo, total): this call samples rows [lo, lo + B) of a global batch of `total`; with noise="philox" every random
    number is keyed by the GLOBAL row, so the shard reproduces those rows of the unsharded call bit for bit."""
    explicit = isinstance(noise, dict)  # parity tests: {"init_noise": [B,H,W], "q": [rows,L] per step, "u": [B,H,W] per step}
    if not explicit and noise not in ("torch", "philox"):
        raise ValueError("noise must be 'torch', 'philox' or a dict of explicit noise tensors")
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("paella_amd.sample runs on a HIP device only (got device=%s); there is no CPU path" % device)
    _lib.load()
    B, H, W = (int(v) for v in latent_shape)
    L = model.num_labels
    rows = B * H * W
    native = isinstance(model, Paella)
    philox = (not explicit) and noise == "philox"
    if philox and seed is None:
        seed = 0 if seed_dev is not None else fresh_seed()  # GraphSampler keeps the live seed in device memory
    if seed is None:
        seed = 0  # unused: every random number comes from torch's generator or from explicit tensors
    if shard is not None and not (philox or explicit):
        raise ValueError("shard=(lo, total) needs noise='philox' (or explicit noise tensors): torch's generator stream cannot be sharded")
    row_offset = 0 if shard is None else int(shard[0]) * H * W
    with torch.inference_mode():
        if init_noise_buf is not None and philox:
            init_noise = start_tokens(L, (B, H, W), seed, device, shard, out=init_noise_buf, seed_dev=seed_dev, row_offset_dev=row_offset_dev)
        elif init_noise_buf is not None:
            init_noise = init_noise_buf
        elif explicit:
            init_noise = noise["init_noise"].to(device=device, dtype=torch.int64).contiguous()
        elif philox:
            init_noise = start_tokens(L, (B, H, W), seed, device, shard, seed_dev=seed_dev, row_offset_dev=row_offset_dev)
        else:
            init_noise = torch.randint(0, L, size=(B, H, W), device=device)
        sampled = init_noise.clone() if init_x is None else init_x.to(device=device, dtype=torch.int64).contiguous()
        any_cfg = any(c is not None for c in cfgs)
        batched = False
        if native:
            if any_cfg and _same_cond_layout(model_inputs, unconditional_inputs):
                cond_both = model.prepare_cond(**_cat_inputs(model_inputs, unconditional_inputs), ws=ws)
                batched = True
            if not batched or not all(c is not None for c in cfgs):
                cond_c = model.prepare_cond(**model_inputs, ws=ws)
                cond_u = model.prepare_cond(**unconditional_inputs, ws=ws) if any_cfg and not batched else None
        # logits buffers exist only for steps that take the unfused path (allocated on first use: 8.6 GB each at BASELINE configs[2])
        bufs = {}
        def logits_buf(name, rows_b):
            if name not in bufs:
                bufs[name] = torch.empty(rows_b, H, W, L, dtype=torch.float32, device=device)
            return bufs[name]
        out = torch.empty(B, H, W, dtype=torch.int64, device=device)
        if r_all is None:
            r_all = timestep_table(t_list, steps, B, device)
        fuse = philox and native and fused_tail  # head GEMM + tail in one launch: no logits tensor at all
        for i in range(steps):
            r = r_all[i]
            use_cfg = cfgs[i] is not None
            temp = temperatures[i]
            mode = 1 if temp == 0 else 0
            renoise = i < renoise_steps
            if fuse and ((use_cfg and batched and mode == 0) or not use_cfg):
                # in place is safe: the token gather at the head of the forward and the token store at its tail are different kernels
                # of one stream
                model.forward_sample(sampled, r, cond_both if use_cfg else cond_c, out, temperature=temp if mode == 0 else 1.0, argmax=mode == 1,
                                     seed=seed, seed_dev=seed_dev, offset=i, row_offset=row_offset, row_offset_dev=row_offset_dev,
                                     init_noise=init_noise if renoise else None,
                                     t_next=t_list[i + 1] if renoise else 0.0, cfg_mix=cfgs[i] if use_cfg else None, attn_weights=attn_weights, ws=ws)
                sampled = out
                continue
            if native:
                if use_cfg and batched:
                    # one evaluation against the 2B-row conditioning cache: tokens / r are passed once, the library computes
                    # the conditioning-free prefix for them and replicates it where the two passes diverge.  With the
                    # counter-based generator (no bit-parity promise towards torch's RNG stream) a categorical step also lets
                    # the guidance mix ride through the linear head: one mixed logits tensor comes back.
                    fold = philox and mode == 0
                    logits2 = logits_buf("both", 2 * B)
                    model.forward_prepared(sampled, r, cond_both, attn_weights=attn_weights, out=logits2[:B] if fold else logits2,
                                           cfg_mix=cfgs[i] if fold else None, ws=ws)
                    lc, lu = logits2[:B], (None if fold else logits2[B:])
                else:
                    lc, lu = logits_buf("c", B), None
                    model.forward_prepared(sampled, r, cond_c, attn_weights=attn_weights, out=lc, ws=ws)
                    if use_cfg:
                        lu = logits_buf("u", B)
                        model.forward_prepared(sampled, r, cond_u, attn_weights=attn_weights, out=lu, ws=ws)
            else:  # any other callable with the reference's signature; logits come back [B, L, H, W]
                lc = model(sampled, r, **model_inputs).permute(0, 2, 3, 1).float().contiguous()
                lu = model(sampled, r, **unconditional_inputs).permute(0, 2, 3, 1).float().contiguous() if use_cfg else None
            noise_q = None
            if explicit and mode == 0:
                noise_q = noise["q"][i].to(device=device, dtype=torch.float32).contiguous()
            elif noise == "torch" and mode == 0:
                # what torch.multinomial(scores, 1) draws internally: q = empty_like(scores).exponential_(1)
                # in the memory order of the reference's `scores.permute(0,2,3,1).reshape(-1, L)`: row-major copy for
                # B > 1, but a column-major VIEW of the NCHW softmax output for B == 1
                if B == 1:
                    noise_q = torch.empty(L, rows, dtype=torch.float32, device=device).exponential_(1).t().contiguous()
                else:
                    noise_q = torch.empty(rows, L, dtype=torch.float32, device=device).exponential_(1)
            mask_u = None
            if renoise and explicit:
                mask_u = noise["u"][i].to(device=device, dtype=torch.float32).contiguous()
            elif renoise and noise == "torch":
                mask_u = torch.rand(B, H, W, dtype=torch.float32, device=device)  # == torch.rand_like(x.float())
            cfg, omc = cfgs[i] if use_cfg else (1.0, 0.0)
            _tail(lc, lu, rows, L, cfg, omc, temp if mode == 0 else 1.0, mode, noise_q, seed, i,
                  init_noise if renoise else None, mask_u, t_list[i + 1] if renoise else 0.0, out, seed_dev=seed_dev,
                  row_offset=row_offset, row_offset_dev=row_offset_dev)
            sampled = out  # the tail never reads `sampled`, so one output buffer is enough (stream-ordered reuse)
    return sampled


def sample(model, model_inputs, latent_shape, unconditional_inputs=None, steps=12, renoise_steps=11, temperature=(1.0, 0.2),
           cfg=8.0, t_start=1.0, t_end=0.0, device="cuda", *, noise="torch", seed=None, seed_dev=None, attn_weights=None, shard=None, fused_tail=True):
    """Drop-in for reference src/utils.py:35 `sample` (same positional order and defaults).
    Keyword-only extensions: noise ("torch" = consume torch's generator exactly like the reference; "philox" = counter-based
    noise generated in the kernels, keyed by `seed` -- None draws a fresh seed from torch's generator), attn_weights
    (utils/modules.py:268), shard=(lo, total) for batch-sharded sampling (paella_amd.dist.sample_sharded), fused_tail (counter-based
    mode only: take the categorical decision inside the head GEMM so the logits are never written; False keeps the two-kernel
    path -- same tokens bit for bit, for A/B measurements), seed_dev (counter-based mode: a 1-element int64 DEVICE tensor added to `seed` inside
    the kernels -- a seed that arrived in a collective is used without a host round trip; pass seed=0 with it)."""
    if cfg and unconditional_inputs is None:
        # the reference raises TypeError at src/utils.py:46 (`**None`); keep the failure, make it readable
        raise TypeError("cfg=%r requires unconditional_inputs" % (cfg,))
    t_list = linspace_schedule(t_start, t_end, steps + 1)
    temperatures = linspace_schedule(temperature[0], temperature[1], steps)
    if cfg:
        # `logits * cfg + logits_u * (1 - cfg)` with python-float cfg: both scalars are rounded to fp32 by torch
        pair = (float(torch.tensor(float(cfg), dtype=torch.float32)), float(torch.tensor(1.0 - float(cfg), dtype=torch.float32)))
        cfgs = [pair] * steps
    else:
        cfgs = [None] * steps
    return _sample_core(model, model_inputs, unconditional_inputs, latent_shape, None, steps, renoise_steps, t_list, temperatures,
                        cfgs, device, noise=noise, seed=seed, seed_dev=seed_dev, attn_weights=attn_weights, shard=shard, fused_tail=fused_tail)


def sample_distributed(model, model_inputs, unconditional_inputs, latent_shape, init_x=None, steps=12, renoise_steps=None,
                       temperature=(0.7, 0.3), cfg=(8.0, 8.0), t_start=1.0, t_end=0.0, sampling_conditional_steps=None, *,
                       noise="torch", seed=None, attn_weights=None, shard=None, fused_tail=True):
    """Drop-in for reference src_distributed/utils.py:97 `sample` (init_x, cfg schedule, conditional-step cutoff)."""
    device = unconditional_inputs["byt5"].device
    if sampling_conditional_steps is None:
        sampling_conditional_steps = steps
    if renoise_steps is None:
        renoise_steps = steps - 1
    t_list = linspace_schedule(t_start, t_end, steps + 1)
    temperatures = linspace_schedule(temperature[0], temperature[1], steps)
    cfgs = [None] * steps
    if cfg is not None:
        sched = torch.linspace(cfg[0], cfg[1], steps)
        for i in range(min(steps, sampling_conditional_steps)):
            # `logits * cfgs[i] + logits_u * (1 - cfgs[i])` with a 0-dim fp32 tensor: (1 - cfg) is computed in fp32
            cfgs[i] = (float(sched[i]), float(1 - sched[i]))
    return _sample_core(model, model_inputs, unconditional_inputs, latent_shape, init_x, steps, renoise_steps, t_list, temperatures,
                        cfgs, device, noise=noise, seed=seed, attn_weights=attn_weights, shard=shard, fused_tail=fused_tail)


class GraphSampler:
    """The whole `sample()` call (conditioning prep, all denoising steps, sampling tails; optionally the VQGAN
    decode) captured ONCE into a HIP graph for fixed shapes and replayed per request.

    Why: at batch 1 the path is ~3000 short kernels per image and eager launches are host-bound at ~2.8 us per kernel
    on this platform (tools/launch_floor.py), a graph replays the same kernels at ~1.6-1.8 us.  Arithmetic, kernels and
    results are identical to `sample(..., noise="philox")`; only the submission mechanism changes.  Per request the
    conditioning tensors are copied into the graph's static input buffers and two device words are rewritten: the Philox seed
    and the GLOBAL row offset of this replay's batch shard.  Start tokens, categorical draws and renoise masks are all drawn
    inside the graph as functions of (seed, global row, step): `sampler(seed=s, shard=(lo, total))` returns exactly rows
    [lo, lo + B) of `sample(..., noise="philox", seed=s)` over the global batch (tests/test_gpu_sample.py).
    seed=None draws a fresh seed from torch's global CPU generator (one draw per call; see fresh_seed).

    Staleness (the call site that reloads weights between sampling calls is src/train.py:40,64-69,76): a capture bakes raw pointers,
    host-side constants read from the weights (the VQGAN's ResBlock gammas) and the precision mode's kernel choice.  Every replay first compares the models'
    parameter signature (data_ptr + version of every tensor) and precision with what the capture saw; on a difference it RECAPTURES (on_stale="recapture",
    default: the replay then equals a fresh eager call bit for bit) or raises a RuntimeError naming the cause (on_stale="raise")."""

    def __init__(self, model, model_inputs, unconditional_inputs, latent_shape, steps=12, renoise_steps=11, temperature=(1.0, 0.2),
                 cfg=8.0, t_start=1.0, t_end=0.0, device="cuda", vqgan=None, attn_weights=None, on_stale="recapture"):
        if on_stale not in ("recapture", "raise"):
            raise ValueError("on_stale must be 'recapture' or 'raise'")
        self.model, self.vqgan, self.device, self.on_stale = model, vqgan, torch.device(device), on_stale
        self.shape = tuple(int(v) for v in latent_shape)
        self.kw = dict(steps=steps, renoise_steps=renoise_steps, temperature=temperature, cfg=cfg, t_start=t_start, t_end=t_end)
        self.attn_weights = attn_weights
        clone = lambda d: None if d is None else {k: (None if v is None else ([t.clone() for t in v] if isinstance(v, (list, tuple)) else v.clone()))
                                                  for k, v in d.items()}
        self.cond, self.uncond = clone(model_inputs), clone(unconditional_inputs)
        self.seed_dev = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.row_offset_dev = torch.zeros(1, dtype=torch.int64, device=self.device)  # lo * H * W of the shard being sampled
        self.init_noise = torch.zeros(self.shape, dtype=torch.int64, device=self.device)
        self.r_all = timestep_table(linspace_schedule(t_start, t_end, steps + 1), steps, self.shape[0], self.device)
        self.captures = 0
        self._capture()

    # ---- what a capture depends on besides the shapes
    def _state(self):
        st = [("denoiser weights", self.model._signature()), ("denoiser gemm precision", self.model._precision)]
        if self.vqgan is not None:
            st += [("VQGAN weights", self.vqgan._signature()), ("VQGAN gemm precision", self.vqgan._precision)]
        return st

    def _capture(self):
        # The graph bakes raw pointers: it owns its workspaces (the modules' own scratch is dropped and reallocated whenever a
        # later eager call needs a bigger one) and it is the only user of their split-K tickets.  Sized here: the fast mode needs more room.
        B, H, W = self.shape
        S = max(_cond_seq_len(self.model, self.cond), _cond_seq_len(self.model, self.uncond), 1)
        self.graph = self.out = None
        self.ws = _lib.new_workspace(self.model.workspace_bytes(2 * B, H, W, S), self.device)
        self.vq_ws = None if self.vqgan is None else _lib.new_workspace(self.vqgan.workspace_bytes(B, H, W), self.device)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(2):  # warm-up: weights (re)loaded, workspaces sized, allocator pools populated
                self._run()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        state = self._state()  # AFTER the warm-up: the engines are loaded with exactly these tensors
        graph = torch.cuda.CUDAGraph()
        # thread_local: a process-group watchdog thread polling events must not invalidate the capture
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            out = self._run()
        torch.cuda.synchronize(self.device)
        self.graph, self.out, self._captured_state = graph, out, state
        self.captures += 1

    def _schedule(self):
        """(t_list, temperatures, per-step guidance pairs) of the src/utils.py:35 signature"""
        k = self.kw
        t_list = linspace_schedule(k["t_start"], k["t_end"], k["steps"] + 1)
        temps = linspace_schedule(k["temperature"][0], k["temperature"][1], k["steps"])
        if k["cfg"]:
            pair = (float(torch.tensor(float(k["cfg"]), dtype=torch.float32)), float(torch.tensor(1.0 - float(k["cfg"]), dtype=torch.float32)))
            cfgs = [pair] * k["steps"]
        else:
            cfgs = [None] * k["steps"]
        return t_list, temps, cfgs

    def _init_x(self):
        """tokens the loop starts from (None = the Philox start tokens); GraphInpainter encodes + renoises an image here"""
        return None

    def _finish(self, toks):
        return toks

    def _run(self):
        k = self.kw
        t_list, temps, cfgs = self._schedule()
        toks = _sample_core(self.model, self.cond, self.uncond, self.shape, self._init_x(), k["steps"], k["renoise_steps"], t_list, temps, cfgs,
                            self.device, noise="philox", seed=0, attn_weights=self.attn_weights, seed_dev=self.seed_dev,
                            init_noise_buf=self.init_noise, r_all=self.r_all, ws=self.ws, row_offset_dev=self.row_offset_dev)
        toks = self._finish(toks)
        return toks if self.vqgan is None else (toks, self.vqgan.decode_indices(toks, ws=self.vq_ws))

    @staticmethod
    def _copy_inputs(dst, src):
        if dst is None:
            if src is not None:
                raise ValueError("the graph was captured without this conditioning set")
            return
        for key, d in dst.items():
            s = src.get(key) if src is not None else None
            if d is None:
                if s is not None:
                    raise ValueError("conditioning layout differs from the captured one (%s)" % key)
                continue
            if isinstance(d, (list, tuple)):
                if not isinstance(s, (list, tuple)) or len(s) != len(d) or any(a.shape != b.shape for a, b in zip(d, s)):
                    raise ValueError("conditioning layout differs from the captured one (%s: list length / shapes)" % key)
                for a, b in zip(d, s):
                    a.copy_(b)
            else:
                if s is None or isinstance(s, (list, tuple)) or s.shape != d.shape:
                    raise ValueError("conditioning shape differs from the captured one (%s)" % key)
                d.copy_(s)

    def _check_fresh(self):
        now = self._state()
        if now == self._captured_state:
            return
        causes = [a[0] for a, b in zip(now, self._captured_state) if a != b]
        if self.on_stale == "raise":
            raise RuntimeError("GraphSampler: the captured graph is stale -- changed since the capture: %s (load_state_dict / optimizer step / .to() / set_gemm_precision). "
                               "Build a new GraphSampler or construct it with on_stale='recapture'." % ", ".join(causes))
        self._capture()

    def _set_words(self, seed, shard):
        if seed is None:
            seed = fresh_seed()
        lo = 0
        if shard is not None:
            lo, total = int(shard[0]), int(shard[1])
            if lo < 0 or lo + self.shape[0] > total:
                raise ValueError("shard=(lo, total): rows [lo, lo + %d) must lie inside the global batch" % self.shape[0])
        self.row_offset_dev.fill_(lo * self.shape[1] * self.shape[2])
        self.seed_dev.fill_(int(seed))

    def __call__(self, model_inputs=None, unconditional_inputs=None, seed=None, shard=None):
        """Replay. Returns tokens (and the decoded image if a VQGAN was given); outputs live in graph-owned buffers that
        the next replay overwrites.  seed=None draws a fresh seed from torch's CPU generator; the start tokens and all per-step
        noise are functions of the seed and of the GLOBAL row: shard=(lo, total) makes this replay rows [lo, lo + B) of a
        global batch of `total` (bit-identical to those rows of the unsharded call with the same seed)."""
        self._check_fresh()
        if model_inputs is not None:
            self._copy_inputs(self.cond, model_inputs)
        if unconditional_inputs is not None:
            self._copy_inputs(self.uncond, unconditional_inputs)
        self._set_words(seed, shard)
        self.graph.replay()
        return self.out
