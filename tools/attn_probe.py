"""A/B of the attention kernels at the BASELINE configs[4] per-GPU shapes (1B model, 128x128 tokens, batch 16 -> 32 guidance rows,
16 heads x head_dim 80, ByT5 768 + CLIP text + CLIP image conditioning = 776 rows): the LDS-staged kernel with its three stagings
(direct-to-LDS = the product path; register-staged and the unpadded 4-workgroups-per-CU layout behind the test hook) vs the register-fed
kernel.  Prints time, TFLOP/s (4*Lq*Lk*D flop per head and sample) and checks the outputs are bit-identical.
Usage (GPU box): python tools/attn_probe.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from paella_amd import _lib

lib = _lib.load()
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for (B, Lq, Lcond) in [(32, 256, 776), (32, 1024, 776), (8, 4096, 776), (128, 256, 4), (128, 1024, 4)]:
    nh, D = 16, 80
    c = nh * D
    g = torch.Generator(device="cuda").manual_seed(Lq)
    qkv = torch.randn(B * Lq, 3 * c, device="cuda", generator=g)
    kvc = torch.randn(B * Lcond, 2 * c, device="cuda", generator=g)
    outs = []
    for variant, name in ((1, "register-fed"), (10, "LDS, register-staged"), (11, "LDS, direct-to-LDS padded"), (0, "LDS, DMA unpadded (product)")):
        lib.paella_test_attention_variant(variant)
        out = torch.empty(B * Lq, c, device="cuda")
        # q / k / v are column blocks of the packed projection output, exactly as the model calls it (ld = 3c / 2c)
        def run():
            # paella_op_attention takes contiguous [rows, nhead*D] operands: split once outside the timed region
            return lib.paella_op_attention(q.data_ptr(), ks.data_ptr(), vs.data_ptr(), kc.data_ptr(), vc.data_ptr(), out.data_ptr(), B, nh, D, Lq, Lq, Lcond,
                                           None, 0, st())
        q, ks, vs = (t.contiguous() for t in qkv.split(c, dim=1))
        kc, vc = (t.contiguous() for t in kvc.split(c, dim=1))
        assert run() == 0, lib.paella_last_error()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                run()
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) / 4)
        ms = sorted(ts)[2]
        flop = 4.0 * Lq * (Lq + Lcond) * D * nh * B
        print("B=%3d Lq=%4d Lk=%4d %-26s %8.3f ms  %6.1f TFLOP/s (%.3f of the 157.3 fp32-MFMA peak)" % (B, Lq, Lq + Lcond, name, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3), flush=True)
        outs.append(out)
    lib.paella_test_attention_variant(0)
    print("    outputs bit-identical:", all(bool(torch.equal(outs[0], o)) for o in outs[1:]))
