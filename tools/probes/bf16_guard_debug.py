import ctypes, sys, torch
sys.path.insert(0, '.')
from paella_amd import _lib, build
lib = _lib.load()
DEV = "cuda"
_p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
_st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def parts(blk):
    blk = blk.float(); s = blk.sum(-1)
    return torch.stack([s, ((blk - (s / 16)[..., None]) ** 2).sum(-1)], dim=-1).contiguous()
for tile, splitk in [(10, 1), (30, 1), (36, 1)]:
    M, N, K = 216, 168, 448
    g = torch.Generator().manual_seed(tile * 13 + splitk)
    A = torch.randn(M, K, generator=g) * 1.5 + 0.3 + torch.arange(K)[None, :] * 0.004
    A[:80] += 240.0 + torch.arange(80)[:, None] * 0.5
    W = torch.randn(N, K, generator=g) / K ** 0.5 + torch.arange(N)[:, None] * 0.001
    stats = parts(A.view(M, K // 16, 16)).to(DEV)
    A16, W16 = A.bfloat16(), W.bfloat16()
    mu = A.double().mean(1, keepdim=True)
    rstd = 1.0 / torch.sqrt(A.double().var(1, unbiased=False, keepdim=True) + 1e-6)
    ln = (A.double() - mu) * rstd
    ref_g = (ln.float().bfloat16().double() @ W16.double().t()).float()
    ref_f = (((A16.double() - mu) * rstd) @ W16.double().t()).float()
    C = torch.full((M, N), float("nan"), device=DEV)
    ws = _lib.new_workspace(128 << 20, DEV)
    a16d, a32d, w16d = A16.to(DEV), A.to(DEV), W16.to(DEV)
    rc = lib.paella_test_gemm_bf16_ln(_p(a16d), _p(a32d), _p(w16d), _p(C), M, N, K, _p(stats), tile, splitk, _p(ws), ws.numel(), _st())
    torch.cuda.synchronize()
    c = C.cpu()
    eg, ef = (c - ref_g).abs(), (c - ref_f).abs()
    print("tile", tile, "rc", rc)
    for lo in range(0, M, 16):
        print("  rows %3d..%3d  vs guard-ref max %.2e (n > 2e-3: %d)  vs fold-ref max %.2e" % (lo, min(lo + 16, M) - 1, eg[lo:lo + 16].max(), int((eg[lo:lo+16] > 2e-3).sum()), ef[lo:lo + 16].max()))
    if tile == 10:
        bad = (eg[:80] > 2e-3).nonzero()
        print("  violations (row, col):", bad[:20].tolist())
