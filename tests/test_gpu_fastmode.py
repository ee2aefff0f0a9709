"""GPU: the OPT-IN bf16-operand fast mode (paella_set_gemm_precision(1)).  It is outside the fp32 parity
contract, so these tests pin (a) the kernel against exact arithmetic on bf16-rounded operands, (b) the size of the deviation
from the fp32 path (logit error, argmax-flip rate -- printed), and (c) that switching back restores the exact path."""
import ctypes

import numpy as np
import pytest
import torch

import paella_amd
from paella_amd import _lib
from oracle import golden_configs as G
from tests.helpers import cond_for, to_dev, weights_for

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.fixture(autouse=True)
def _restore_precision():
    yield
    paella_amd.set_gemm_precision("fp32")


@pytest.mark.parametrize("tile", [96, 97, 98])
@pytest.mark.parametrize("splitk", [1, 3])
@pytest.mark.parametrize("M,N,K", [(200, 328, 416), (33, 1280, 1288), (512, 640, 2560)])
def test_bf16_gemm_is_exact_on_rounded_operands(built_lib, tile, splitk, M, N, K):
    lib = built_lib
    g = torch.Generator().manual_seed(tile + splitk + M)
    A = torch.randn(M, K, generator=g) + torch.arange(K)[None, :] * 0.01   # asymmetric operands catch transposed fragments
    W = (torch.randn(N, K, generator=g) + torch.arange(N)[:, None] * 0.02) / 8
    bias = torch.randn(N, generator=g)
    ref = (A.bfloat16().double() @ W.bfloat16().double().t() + bias.double()).float()
    Ad, Wd, bd = A.to(DEV), W.to(DEV), bias.to(DEV)
    C = torch.full((M, N), float("nan"), device=DEV)
    ws = _lib.new_workspace(64 << 20, DEV)
    assert lib.paella_test_register_weight(_p(Wd), Wd.numel(), 1) == 0
    try:
        paella_amd.set_gemm_precision("bf16")   # converts the registered matrix
        rc = lib.paella_op_gemm(_p(Ad), _p(Wd), _p(bd), None, _p(C), M, N, K, 0, tile, splitk, _p(ws), ws.numel(), _st())
        assert rc == 0, lib.paella_last_error()
        torch.cuda.synchronize()
    finally:
        lib.paella_test_register_weight(_p(Wd), Wd.numel(), 0)
    np.testing.assert_allclose(C.cpu().numpy(), ref.numpy(), atol=2e-3, rtol=2e-5)


def _flip_report(ref, got):
    top2 = ref.topk(2, dim=1).values
    flips = (ref.argmax(1) != got.argmax(1)).float().mean().item()
    return flips, (got - ref).abs().max().item(), ref.std().item()


@pytest.mark.parametrize("cfg_name,B,grid", [("UNET_MID", 2, 16), ("UNET_570M", 1, 32)])
def test_bf16_forward_deviation_and_restore(built_lib, cfg_name, B, grid):
    cfg = dict(getattr(G, cfg_name))
    m = paella_amd.Paella(**cfg)
    weights_for(m, sum(cfg["blocks"]))
    m = m.to(DEV)
    g = torch.Generator().manual_seed(3)
    x = torch.randint(0, cfg["num_labels"], (B, grid, grid), generator=g).to(DEV)
    r = torch.rand(B, generator=g).to(DEV)
    c = to_dev(cond_for(cfg, B, 3, 0, 7), DEV)
    exact = m(x, r, **c).clone()
    paella_amd.set_gemm_precision("bf16")
    assert paella_amd.get_gemm_precision() == "bf16"
    fast = m(x, r, **c).clone()
    paella_amd.set_gemm_precision("fp32")
    again = m(x, r, **c)
    flips, diff, std = _flip_report(exact, fast)
    print("%s bf16 fast mode: argmax-flip rate %.4f, max|logit diff| %.3e on logits of std %.3f" % (cfg_name, flips, diff, std))
    assert torch.isfinite(fast).all()
    assert not torch.equal(fast, exact)          # the fast path really ran
    assert diff <= 0.25 * max(1.0, std) and flips <= 0.15
    assert torch.equal(again, exact)             # and the exact path is back, bit for bit


def test_bf16_sampling_runs_end_to_end(built_lib):
    cfg = G.UNET_TINY
    m = paella_amd.Paella(**cfg)
    weights_for(m, sum(cfg["blocks"]))
    m = m.to(DEV)
    vq = paella_amd.VQModel(**dict(G.VQ_TINY_F8, codebook_size=cfg["num_labels"]))
    weights_for(vq, 2)
    vq = vq.to(DEV)
    cs, us = to_dev(cond_for(cfg, 2, 3, 0, 1), DEV), to_dev(cond_for(cfg, 2, 3, 0, 2), DEV)
    paella_amd.set_gemm_precision("bf16")
    toks = paella_amd.sample(m, cs, (2, 16, 16), unconditional_inputs=us, steps=4, renoise_steps=3, device=DEV, noise="philox", seed=3)
    img = vq.decode_indices(toks)
    assert int(toks.min()) >= 0 and int(toks.max()) < cfg["num_labels"] and torch.isfinite(img).all()
