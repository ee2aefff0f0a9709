"""`VQModel` -- Paella's VQGAN tokenizer / detokenizer on MI355X (host mirror of reference `src/vqgan.py:45-107`).

Same constructor, `encode` / `decode` / `decode_indices` signatures and state-dict keys as the reference; the
arithmetic runs in libpaella_hip.so (conv-as-GEMM on the fp32 matrix cores, NHWC depthwise / LayerNorm kernels).
`vquantizer` mirrors the call-site contract of `torchtools.nn.VectorQuantize` (third-party, un-vendored and
unpinned in the reference's requirements.txt:12): `.codebook.weight`, `.forward(x, dim)`, `.idx2vq(idx, dim)`.
The `Discriminator` (src/vqgan.py:115) and the broken `VQModel.forward` (:109) are out of scope.
"""
import ctypes

import torch
from torch import nn

from . import _lib
from .modules import _Holder, _p, _wb


def _res_block(c):
    return _Holder(depthwise=_Holder(**{"1": _wb((c, 1, 3, 3))}),
                   channelwise=_Holder(**{"0": _wb((4 * c, c)), "2": _wb((c, 4 * c))}),
                   gammas=nn.Parameter(torch.zeros(6)))


class _BatchNormParams(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class VectorQuantize(nn.Module):
    """Call-site compatible stand-in for torchtools.nn.VectorQuantize, backed by the HIP nearest-code kernel."""

    def __init__(self, embedding_size, k, owner=None):
        super().__init__()
        self.codebook = nn.Embedding(k, embedding_size)
        self.codebook.weight.data.uniform_(-1. / k, 1. / k)
        self._owner = [owner]  # list: keep the parent out of nn.Module registration

    def forward(self, x, get_losses=True, dim=-1):
        owner = self._owner[0]
        if dim != -1:
            x = x.movedim(dim, -1)
        flat = x.contiguous().view(-1, x.size(-1)).float().contiguous()
        idx, qe, mse = owner._quantize_rows(flat, get_losses)
        vq_loss = commit_loss = None
        if get_losses:
            vq_loss, commit_loss = mse[0], mse[0].clone()
        z_q = qe.view(x.shape)
        if dim != -1:
            z_q = z_q.movedim(-1, dim)
        return z_q, (vq_loss, commit_loss), idx.view(x.shape[:-1])

    def idx2vq(self, idx, dim=-1):
        owner = self._owner[0]
        q = owner._gather_rows(idx.contiguous().view(-1)).view(*idx.shape, -1)
        if dim != -1:
            q = q.movedim(-1, dim)
        return q


class VQModel(nn.Module):
    """Drop-in for reference `VQModel` (src/vqgan.py:45). levels=2 is f4 (default), levels=3 is f8."""

    def __init__(self, levels=2, bottleneck_blocks=12, c_hidden=384, c_latent=4, codebook_size=8192, scale_factor=0.3764):
        super().__init__()
        self.c_latent = c_latent
        self.scale_factor = scale_factor
        self.codebook_size = codebook_size
        self.levels = levels
        self._cfg = dict(levels=levels, bottleneck_blocks=bottleneck_blocks, c_hidden=c_hidden, c_latent=c_latent,
                         codebook_size=codebook_size, scale_factor=scale_factor)
        c_levels = [c_hidden // (2 ** i) for i in reversed(range(levels))]
        self.in_block = _Holder(**{"1": _wb((c_levels[0], 12, 1, 1))})
        down = []
        for i in range(levels):
            if i > 0:
                down.append(_wb((c_levels[i], c_levels[i - 1], 4, 4)))
            down.append(_res_block(c_levels[i]))
        down.append(_Holder(**{"0": _Holder(weight=_p(c_latent, c_levels[-1], 1, 1)), "1": _BatchNormParams(c_latent)}))
        self.down_blocks = nn.ModuleList(down)
        self.vquantizer = VectorQuantize(c_latent, k=codebook_size, owner=self)
        up = [_Holder(**{"0": _wb((c_levels[-1], c_latent, 1, 1))})]
        for i in range(levels):
            for _ in range(bottleneck_blocks if i == 0 else 1):
                up.append(_res_block(c_levels[levels - 1 - i]))
            if i < levels - 1:
                cin, cout = c_levels[levels - 1 - i], c_levels[levels - 2 - i]
                up.append(_Holder(weight=_p(cin, cout, 4, 4), bias=_p(cout)))
        self.up_blocks = nn.ModuleList(up)
        self.out_block = _Holder(**{"0": _wb((12, c_levels[0], 1, 1))})
        self.reset_parameters()
        self._handle = None
        self._loaded_sig = None
        self._ws = None
        self._precision = 0

    def set_gemm_precision(self, mode):
        """OPT-IN fast mode of THIS model, outside the fp32 parity contract (include/paella_hip.h: paella_vqgan_set_precision): "bf16" runs the ResBlock
        MLPs on bf16-operand MFMA with fp32 accumulation; "fp32" (default) is the exact path.  Rebuild objects that sized a workspace / captured
        launches before the switch (`GraphSampler`)."""
        modes = {"fp32": 0, "f32": 0, "bf16": 1}
        if mode not in modes:
            raise ValueError("gemm precision must be 'fp32' or 'bf16'")
        self._precision = modes[mode]
        if self._handle is not None:
            dev = self._device()
            with torch.cuda.device(dev):
                _lib.check(_lib.load().paella_vqgan_set_precision(self._handle, self._precision, _lib.stream_ptr(dev)))
        self._ws = None
        return self

    def get_gemm_precision(self):
        return "bf16" if self._precision == 1 else "fp32"

    @torch.no_grad()
    def reset_parameters(self):
        # same distributions as the reference: xavier_uniform weights, zero biases, zero gammas (src/vqgan.py:23-31)
        for name, p in self.named_parameters():
            if name.startswith("vquantizer.") or name.endswith("gammas") or ".1.weight" in name and p.dim() == 1:
                continue
            if p.dim() >= 2:
                nn.init.xavier_uniform_(p)
            else:
                p.zero_()
        self.down_blocks[-1]._modules["1"].weight.fill_(1.0)

    # ------------------------------------------------------------------ engine plumbing
    def _device(self):
        return self.vquantizer.codebook.weight.device

    def _signature(self):
        """(data_ptr, version) of every parameter and buffer (the tensor list is cached; `_apply` and `refresh()` drop it) -- see Paella._signature."""
        ts = self.__dict__.get("_sig_tensors")
        if ts is None:
            ts = list(self.parameters()) + list(self.buffers())
            self.__dict__["_sig_tensors"] = ts
        return tuple((t.data_ptr(), t._version) for t in ts)

    def _apply(self, fn, *args, **kwargs):
        self.__dict__["_sig_tensors"] = None
        return super()._apply(fn, *args, **kwargs)

    def refresh(self):
        """Force a reload of every tensor on the next call (needed only after edits torch does not version, e.g. `p.data = ...` with a new storage of a NEW
        Parameter object)."""
        self._loaded_sig = None
        self.__dict__["_sig_tensors"] = None

    def _engine(self):
        dev = self._device()
        if dev.type != "cuda":
            raise RuntimeError("paella_amd.VQModel executes only on a HIP device (module is on '%s'); there is no CPU fallback." % dev)
        lib = _lib.load()
        sig = self._signature()
        if self._handle is not None and sig == self._loaded_sig:
            return self._handle
        with torch.cuda.device(dev):
            if self._handle is None:
                c = _lib.VqganConfig()
                for k in ("levels", "bottleneck_blocks", "c_hidden", "c_latent", "codebook_size"):
                    setattr(c, k, int(self._cfg[k]))
                c.scale_factor = float(self._cfg["scale_factor"])
                h = ctypes.c_void_p()
                _lib.check(lib.paella_vqgan_create(ctypes.byref(c), ctypes.byref(h)))
                self._handle = h
                if self._precision:
                    _lib.check(lib.paella_vqgan_set_precision(self._handle, self._precision, _lib.stream_ptr(dev)))
            st = _lib.stream_ptr(dev)
            for key, t in self.state_dict().items():
                if key.endswith("num_batches_tracked"):
                    continue
                t = t.detach().float().contiguous()
                shape = (ctypes.c_int64 * t.dim())(*t.shape)
                _lib.check(lib.paella_vqgan_load_tensor(self._handle, key.encode(), _lib.ptr(t), shape, t.dim(), st))
            _lib.check(lib.paella_vqgan_finalize(self._handle, st))
        self._loaded_sig = sig
        return self._handle

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None:
            try:
                _lib.load().paella_vqgan_destroy(h)
            except Exception:
                pass

    def _workspace(self, nbytes, ws=None):
        dev = self._device()
        if ws is not None:  # caller-owned (a captured HIP graph must not depend on the module's growable scratch)
            if ws.device != dev or ws.dtype != torch.uint8 or ws.numel() < nbytes:
                raise ValueError("workspace too small or on the wrong device (%d bytes needed)" % nbytes)
            return ws
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != dev:
            self._ws = None
            self._ws = _lib.new_workspace(nbytes, dev)
        return self._ws

    def _quantize_rows(self, flat, with_mse=False):
        h = self._engine()
        dev = self._device()
        idx = torch.empty(flat.size(0), dtype=torch.int64, device=dev)
        qe = torch.empty_like(flat)
        mse = torch.empty(1, dtype=torch.float32, device=dev) if with_mse else None
        with torch.cuda.device(dev):
            _lib.check(_lib.load().paella_vqgan_quantize_rows(h, _lib.ptr(flat), flat.size(0), _lib.ptr(idx), _lib.ptr(qe), _lib.ptr(mse),
                                                              _lib.stream_ptr(dev)))
        return idx, qe, mse

    def _gather_rows(self, idx_flat):
        h = self._engine()
        dev = self._device()
        if not idx_flat.is_cuda or idx_flat.dtype != torch.int64:
            raise ValueError("indices must be an int64 HIP tensor")
        idx_flat = idx_flat.contiguous()
        out = torch.empty(idx_flat.numel(), self.c_latent, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().paella_vqgan_lookup_rows(h, _lib.ptr(idx_flat), idx_flat.numel(), _lib.ptr(out), _lib.stream_ptr(dev)))
        return out

    # ------------------------------------------------------------------ reference surface
    def encode(self, x, ws=None):
        """reference src/vqgan.py:91-95 -> (qe / sf, x / sf, indices, vq_loss + 0.25 * commit_loss).  ws: optional caller-owned workspace (a captured graph
        must not depend on the module's growable scratch)."""
        h = self._engine()
        lib = _lib.load()
        dev = self._device()
        if not x.is_cuda or x.dim() != 4 or x.size(1) != 3:
            raise ValueError("x must be a HIP tensor [B, 3, H, W]")
        x = x.detach().float().contiguous()
        B, _, Hp, Wp = x.shape
        f = 2 ** self.levels
        if Hp % f or Wp % f:
            raise ValueError("image size must be a multiple of %d" % f)
        hh, ww = Hp // f, Wp // f
        qe = torch.empty(B, self.c_latent, hh, ww, dtype=torch.float32, device=dev)
        lat = torch.empty_like(qe)
        idx = torch.empty(B, hh, ww, dtype=torch.int64, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            ws = self._workspace(lib.paella_vqgan_workspace_bytes(h, B, hh, ww), ws)
            _lib.check(lib.paella_vqgan_encode(h, _lib.ptr(x), B, Hp, Wp, _lib.ptr(qe), _lib.ptr(lat), _lib.ptr(idx), _lib.ptr(loss),
                                               _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)))
        return qe, lat, idx, loss[0]

    def workspace_bytes(self, B, hh, ww):
        return int(_lib.load().paella_vqgan_workspace_bytes(self._engine(), B, hh, ww))

    def _decode_common(self, fn, src, B, hh, ww, ws=None):
        h = self._engine()
        lib = _lib.load()
        dev = self._device()
        f = 2 ** self.levels
        img = torch.empty(B, 3, hh * f, ww * f, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            ws = self._workspace(lib.paella_vqgan_workspace_bytes(h, B, hh, ww), ws)
            _lib.check(getattr(lib, fn)(h, _lib.ptr(src), B, hh, ww, _lib.ptr(img), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)))
        return img

    def decode(self, x):
        """reference src/vqgan.py:97-101: x = latents as returned by encode()[0] ([B, c_latent, h, w])"""
        self._engine()
        if not x.is_cuda or x.dim() != 4 or x.size(1) != self.c_latent:
            raise ValueError("x must be a HIP tensor [B, c_latent, h, w]")
        x = x.detach().float().contiguous()
        return self._decode_common("paella_vqgan_decode", x, x.size(0), x.size(2), x.size(3))

    def decode_indices(self, x, ws=None):
        """reference src/vqgan.py:103-107: x = int64 token grid [B, h, w]"""
        self._engine()
        if not x.is_cuda or x.dtype != torch.int64 or x.dim() != 3:
            raise ValueError("x must be an int64 HIP tensor [B, h, w]")
        x = x.contiguous()
        return self._decode_common("paella_vqgan_decode_indices", x, x.size(0), x.size(1), x.size(2), ws)
