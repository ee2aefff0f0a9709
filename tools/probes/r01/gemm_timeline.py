"""Per-workgroup timeline of one fp32 GEMM launch (probe build, tile config 128 + tile): when do workgroups start, when
has their first K tile landed, when does the K loop end, how long do slab publish / ticket / combine + epilogue take?
Usage (GPU box): python tools/gemm_timeline.py M N K tile splitk"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from paella_amd import _lib

M, N, K, tile, sk = (int(v) for v in sys.argv[1:6])
spread = int(sys.argv[6]) if len(sys.argv) > 6 else 0
lib = _lib.load()
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
A = torch.randn(M, K, device="cuda")
Ws = [torch.randn(N, K, device="cuda") for _ in range(max(2, int(600e6 // (N * K * 4)) + 1))]
C = torch.empty(M, N, device="cuda")
BM = 32 if tile == 5 else 64
nwg = ((M + BM - 1) // BM) * ((N + BM - 1) // BM) * sk
trace = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
run = lambda W, cfg: lib.paella_op_gemm(A.data_ptr(), W.data_ptr(), None, None, C.data_ptr(), M, N, K, 0, cfg, sk, ws.data_ptr(), ws.numel(), st())
for W in Ws:
    assert run(W, tile) == 0
assert lib.paella_debug_set_trace(ctypes.c_void_p(trace.data_ptr())) == 0
lib.paella_debug_set_spread(spread)
for W in Ws:  # back-to-back launches like inside the model; the LAST launch's stamps are the ones kept
    trace.zero_()
    assert run(W, 128 + tile) == 0, lib.paella_last_error()
torch.cuda.synchronize()
lib.paella_debug_set_trace(None)
t = trace.cpu().numpy().reshape(nwg, 8).astype(np.float64) * 0.01  # us
t0 = t[:, 0].min()
names = ["start", "first tile in LDS", "K loop done", "slab published", "ticket taken", "epilogue done"]
print("M=%d N=%d K=%d tile %d split-K %d: %d workgroups; times in us from the first workgroup's start" % (M, N, K, tile, sk, nwg))
for i, nm in enumerate(names):
    col = t[:, i]
    col = col[col > 0] - t0
    if len(col) == 0:
        continue
    print("  %-20s n=%5d  min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f" % (nm, len(col), col.min(), np.percentile(col, 10), np.median(col), np.percentile(col, 90), col.max()))
d = lambda a, b: (t[:, b] - t[:, a])[(t[:, a] > 0) & (t[:, b] > 0)]
print("  per-workgroup spans: first-tile latency median %.2f | K loop median %.2f (%.3f per tile) | publish %.2f | ticket %.2f | combine+epilogue (last arrivers) %.2f"
      % (np.median(d(0, 1)), np.median(d(1, 2)), np.median(d(1, 2)) / max(1, (K // sk) // 32 - 1), np.median(d(2, 3)) if sk > 1 else 0, np.median(d(3, 4)) if sk > 1 else 0, np.median(d(4, 5)) if sk > 1 else np.median(d(2, 5))))

# placement: HW_ID bits [11:8] cu, [12] sh, [15:13] se; XCC_ID bits [3:0]
raw = trace.cpu().numpy().reshape(nwg, 8)
hw, xcc = raw[:, 6], raw[:, 7] & 0xF
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
uniq, cnt = np.unique(key, return_counts=True)
print("  placement: %d distinct CUs used; workgroups per CU: min %d median %d max %d; histogram %s" % (len(uniq), cnt.min(), int(np.median(cnt)), cnt.max(), dict(zip(*np.unique(cnt, return_counts=True)))))
end = t[:, 2] - t0
per_cu_end = {k: end[key == k].max() for k in uniq}
by_cnt = {}
for k, c in zip(uniq, cnt):
    by_cnt.setdefault(int(c), []).append(per_cu_end[k])
print("  K-loop end of a CU's slowest workgroup, by workgroups on that CU: " + ", ".join("%d wg: %.1f us (n=%d)" % (c, np.mean(v), len(v)) for c, v in sorted(by_cnt.items())))
per_xcc = [end[xcc == x].mean() for x in sorted(set(xcc.tolist()))]
print("  mean K-loop end per XCC: " + " ".join("%.1f" % v for v in per_xcc))
