import os, sys, torch
sys.path.insert(0, os.getcwd())
import paella_amd
from oracle import golden_configs as G
from paella_amd import synth
v = paella_amd.VQModel(**G.VQ_F8); synth.randomize_(v, seed=0); v = v.to("cuda")
x = torch.randn(262144, 4, device="cuda")
for n in (262144, 4000):
    xs = x[:n].contiguous()
    v.vquantizer.forward(xs, get_losses=False); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): v.vquantizer.forward(xs, get_losses=False)
    e1.record(); e1.synchronize()
    print("codebook search, %d rows: %.3f ms per call (incl. the gather of the quantised rows)" % (n, e0.elapsed_time(e1) / 5))
