"""GPU: VQGAN encode / decode through the C ABI vs the reference's golden outputs and, at full size, the CPU oracle."""
import numpy as np
import pytest
import torch

import paella_amd
from oracle import golden_configs as G
from oracle import paella_oracle as O
from tests.helpers import weights_for

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("name,vc", [("vq_tiny_f4", G.VQ_TINY_F4), ("vq_tiny_f8", G.VQ_TINY_F8)])
def test_tiny_vs_reference(golden, built_lib, name, vc):
    g = golden(name)
    v = paella_amd.VQModel(**vc)
    sd = weights_for(v, vc["bottleneck_blocks"], g)
    v = v.to(DEV)
    img = torch.from_numpy(g["img"]).to(DEV)
    qe, lat, idx, loss = v.encode(img)
    np.testing.assert_allclose(lat.cpu().numpy(), g["lat"], atol=2e-5)
    mism = idx.cpu().numpy() != g["idx"]
    # integer output: exact, except where the reference-side latent is equidistant (within eps) from its two nearest codes --
    # the only place the unpinned third-party quantiser's tie-break / distance formulation can matter.  Counted and printed.
    rows = (torch.from_numpy(g["lat"]) * vc["scale_factor"]).permute(0, 2, 3, 1).reshape(-1, vc["c_latent"]).double()
    d = torch.cdist(rows, sd["vquantizer.codebook.weight"].double()).pow(2)
    top = d.topk(2, dim=1, largest=False).values
    near = ((top[:, 1] - top[:, 0]) < 1e-5).numpy().reshape(mism.shape)
    print(name, "token mismatches: %d of %d, all at nearest-code near-ties: %s (%d near-ties present)"
          % (int(mism.sum()), mism.size, not (mism & ~near).any(), int(near.sum())))
    assert not (mism & ~near).any(), "%d token(s) differ where the nearest code was unambiguous" % int((mism & ~near).sum())
    if not mism.any():
        np.testing.assert_allclose(qe.cpu().numpy(), g["qe"], atol=1e-6)
        np.testing.assert_allclose(float(loss), float(g["loss"]), rtol=1e-4)
    dec_i = v.decode_indices(torch.from_numpy(g["idx"]).to(DEV))
    np.testing.assert_allclose(dec_i.cpu().numpy(), g["dec_idx"], atol=5e-5)
    dec = v.decode(torch.from_numpy(g["qe"]).to(DEV))
    np.testing.assert_allclose(dec.cpu().numpy(), g["dec"], atol=5e-5)
    # round-trip property: decode(encode(x)[0]) == decode_indices(encode(x)[2])
    np.testing.assert_allclose(v.decode(qe).cpu().numpy(), v.decode_indices(idx).cpu().numpy(), atol=1e-5)
    # vquantizer call-site contract (src_distributed/train.py:155-156)
    rows = lat.permute(0, 2, 3, 1) * vc["scale_factor"]
    assert torch.equal(v.vquantizer.forward(rows, dim=-1)[-1], idx)
    assert v.vquantizer.idx2vq(idx, dim=1).shape == qe.shape


def test_full_size_f8_decode_vs_oracle(built_lib):
    """BASELINE decode: VQModel(levels=3) on a 32x32 token grid -> 256x256 px (38.8 GFLOP)."""
    vc = G.VQ_F8
    v = paella_amd.VQModel(**vc)
    sd = weights_for(v, vc["bottleneck_blocks"])
    v = v.to(DEV)
    g = torch.Generator().manual_seed(1)
    idx = torch.randint(0, vc["codebook_size"], (1, 32, 32), generator=g)
    with torch.no_grad():
        ref = O.vq_decode_indices(sd, vc, idx)
    got = v.decode_indices(idx.to(DEV)).cpu()
    assert got.shape == (1, 3, 256, 256)
    diff = (got - ref).abs().max().item()
    print("f8 decode: output std %.3f max|diff| %.3e" % (ref.std().item(), diff))
    assert diff <= 2e-4 * max(1.0, ref.abs().max().item())
