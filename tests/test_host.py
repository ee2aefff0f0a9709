"""CPU: host-side logic of the drop-in surface (state-dict layout, schedules, fail-loud behaviour off-GPU)."""
import numpy as np
import pytest
import torch

import paella_amd
from oracle import golden_configs as G
from paella_amd import sampling, synth
from paella_amd.dist import shard_bounds


def _keys(module):
    return sorted(k + ":" + ",".join(str(d) for d in v.shape) for k, v in module.state_dict().items())


@pytest.mark.parametrize("name,cfg", [("unet_tiny_forward", G.UNET_TINY), ("unet_mid_forward", G.UNET_MID), ("unet_variant_forward", G.UNET_VARIANT)])
def test_unet_state_dict_matches_reference(golden, name, cfg):
    """Key names and shapes equal those enumerated from the live reference instance (stored with the fixture)."""
    m = paella_amd.Paella(**cfg)
    assert _keys(m) == sorted(golden(name)["keys"].tolist())


@pytest.mark.parametrize("name,cfg", [("vq_tiny_f4", G.VQ_TINY_F4), ("vq_tiny_f8", G.VQ_TINY_F8)])
def test_vqgan_state_dict_matches_reference(golden, name, cfg):
    assert _keys(paella_amd.VQModel(**cfg)) == sorted(golden(name)["keys"].tolist())


def test_default_block_layout():
    # default blocks=[6,16,6] / level_config=['CT','CTA','CTA'], checked on a narrow model
    cfg = G.UNET_1B
    tiny = paella_amd.Paella(**dict(cfg, c_hidden=[8, 16, 16], nhead=[-1, 1, 1], c_in=8, c_out=8, num_labels=16, c_cond=16,
                                    clip_embd=8, byt5_embd=8))
    assert len(tiny.down_blocks[1]) == 1 + 16 * 3 and len(tiny.up_blocks[0]) == 6 * 3 + 1
    assert paella_amd.DenoiseUNet is paella_amd.Paella


def test_reference_init_semantics():
    m = paella_amd.Paella(**G.UNET_TINY)
    sd = m.state_dict()
    assert sd["clf.1.weight"].abs().max() == 0  # logits identically zero at init, as in the reference
    assert all(v.abs().max() == 0 for k, v in sd.items() if k.endswith(".mapper.weight") and "byt5" not in k and "clip" not in k)
    assert torch.equal(sd["out_mapper.1.weight"][:, :, 0, 0], sd["in_mapper.0.weight"])
    synth.randomize_(m, seed=1)
    assert m.state_dict()["clf.1.weight"].abs().max() > 0


def test_fails_loudly_off_gpu():
    m = paella_amd.Paella(**G.UNET_TINY)
    x = torch.zeros(1, 8, 8, dtype=torch.long)
    with pytest.raises(RuntimeError, match="HIP device"):
        m(x, torch.zeros(1), torch.zeros(1, 2, 40))
    with pytest.raises(RuntimeError, match="HIP device"):
        m.add_noise(x, torch.zeros(1))
    with pytest.raises(RuntimeError, match="HIP device"):
        paella_amd.sample(m, {"byt5": torch.zeros(1, 2, 40)}, (1, 8, 8), unconditional_inputs={"byt5": torch.zeros(1, 2, 40)}, device="cpu")
    v = paella_amd.VQModel(**G.VQ_TINY_F4)
    with pytest.raises(RuntimeError, match="HIP device"):
        v.decode_indices(x)


def test_schedules_match_torch_linspace():
    assert sampling.linspace_schedule(1.0, 0.0, 9) == [float(v) for v in torch.linspace(1.0, 0.0, 9)]
    assert sampling.linspace_schedule(1.0, 0.2, 8)[-1] == float(torch.tensor(0.2))


def test_sample_signatures_match_reference():
    import inspect
    s = inspect.signature(paella_amd.sample)
    assert list(s.parameters)[:11] == ["model", "model_inputs", "latent_shape", "unconditional_inputs", "steps", "renoise_steps",
                                       "temperature", "cfg", "t_start", "t_end", "device"]
    assert s.parameters["steps"].default == 12 and s.parameters["renoise_steps"].default == 11
    assert s.parameters["temperature"].default == (1.0, 0.2) and s.parameters["cfg"].default == 8.0
    d = inspect.signature(paella_amd.sample_distributed)
    assert list(d.parameters)[:12] == ["model", "model_inputs", "unconditional_inputs", "latent_shape", "init_x", "steps",
                                       "renoise_steps", "temperature", "cfg", "t_start", "t_end", "sampling_conditional_steps"]
    assert d.parameters["temperature"].default == (0.7, 0.3) and d.parameters["cfg"].default == (8.0, 8.0)
    f = inspect.signature(paella_amd.Paella.forward)
    assert list(f.parameters)[:7] == ["self", "x", "r", "byt5", "clip", "clip_image", "x_cat"]
    with pytest.raises(TypeError):  # reference: cfg=8.0 with unconditional_inputs=None raises TypeError (src/utils.py:46)
        paella_amd.sample(None, {}, (1, 8, 8))


def test_shard_bounds_partition():
    for B in (1, 7, 8, 64, 255):
        for w in (1, 2, 4, 8):
            spans = [shard_bounds(B, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_synth_weights_are_deterministic():
    m = paella_amd.Paella(**G.UNET_TINY)
    a = synth.synth_state_dict(m.state_dict(), seed=0, n_blocks=4)
    b = synth.synth_state_dict(m.state_dict(), seed=0, n_blocks=4)
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert synth.checksum(a) == synth.checksum(b)


def test_bench_traffic_files_are_stamped_and_stale_ones_refused(monkeypatch):
    """bench.py's `roofline.traffic` comes from committed rocprofv3 PMC passes; each file carries the hash of the kernel sources it was
    measured on.  A file measured on other sources must be refused (traffic null + the reason), never silently reused."""
    import bench
    for wl in [(1, 32, 8), (32, 32, 8), (64, 64, 12)]:
        val, note = bench.load_traffic("570m", *wl)
        if val is None:  # kernels edited since the last collection (tools/collect_profiles.sh): reported, not hidden
            assert "no PMC traffic file for the current kernel sources" in note
        else:
            assert val > 0 and bench.source_stamp() in note
    monkeypatch.setattr(bench, "source_stamp", lambda: "0" * 16)
    val, note = bench.load_traffic("570m", 1, 32, 8)
    assert val is None and "stale" in note and "refused" in note
    val, note = bench.load_traffic("570m", 7, 32, 8)  # a workload nobody profiled
    assert val is None


def test_gemm_precision_is_a_per_model_switch_with_no_process_wide_state():
    """The opt-in bf16 fast mode is selected per model object and validated on the host; the package exposes no process-wide setter, and
    the library exports none (include/paella_hip.h: paella_unet_set_precision / paella_vqgan_set_precision take a model handle)."""
    import pytest
    import paella_amd
    from paella_amd import _lib
    a, b = paella_amd.Paella(**G.UNET_TINY), paella_amd.Paella(**G.UNET_TINY)
    assert a.get_gemm_precision() == b.get_gemm_precision() == "fp32"
    assert a.set_gemm_precision("bf16") is a and a.get_gemm_precision() == "bf16" and b.get_gemm_precision() == "fp32"
    a.set_gemm_precision("fp32")
    with pytest.raises(ValueError):
        a.set_gemm_precision("fp16")
    v = paella_amd.VQModel(**G.VQ_TINY_F8)
    assert v.set_gemm_precision("bf16").get_gemm_precision() == "bf16"
    assert not hasattr(paella_amd, "set_gemm_precision")
    assert not any("gemm_precision" in n for n in list(_lib.SIGNATURES) + list(_lib.TEST_HOOKS))
    assert "paella_unet_set_precision" in _lib.SIGNATURES and "paella_vqgan_set_precision" in _lib.SIGNATURES


def test_bench_workload_tables_and_broadcast_layouts():
    """bench.py's N > 1 path: the distributed table names BASELINE's two 8-GPU configurations at their per-GPU share (batch 256 / 8 = 32 and 128 / 8 = 16,
    the second one the inpainting path), every workload has its SURVEY 8(d) algorithmic FLOPs, and the broadcast layout every rank derives from the request SHAPES alone
    (cond_spec_layout) equals the layout of the tensors rank 0 actually packs (conditioning_layout) -- a mismatch would poison every step's broadcast."""
    import bench
    from paella_amd.dist import cond_spec_layout, conditioning_layout
    dist_tab = [bench.EXTRA_WORKLOADS[i] for i in bench.EXTRA_DISTRIBUTED]
    assert ("1b", 256 // 8, 64, 12, 256, 1, False) in [w[:7] for w in dist_tab]
    assert ("1b", 128 // 8, 128, 12, 256, 1, True) in [w[:7] for w in dist_tab]
    assert all(w[9] for w in bench.EXTRA_WORKLOADS), "every throughput workload goes through a captured graph"
    assert all(w[7] >= 2 for w in bench.EXTRA_WORKLOADS), ">= 2 timed steps per workload"
    for (name, batch, grid, steps, s_byt5, n_ci, inpaint, k, w, gr) in bench.EXTRA_WORKLOADS + bench.REHEARSAL_WORKLOADS:
        if name != "tiny":
            assert (name, grid, steps) in bench.ALGO_GFLOP_PER_IMAGE
        cfg = bench.MODELS[name]
        m = paella_amd.Paella(**cfg) if name == "tiny" else None

        class _Shape:  # cond_spec_layout needs the embedding widths only
            _cfg = cfg
        for world in (1, 2, 8):
            total = batch * world
            mk = lambda seed: synth.synth_conditioning(min(total, 4), s_byt5, cfg["byt5_embd"], cfg["clip_embd"], seed=seed, n_clip_image=n_ci)
            lay = cond_spec_layout(_Shape, min(total, 4), S_byt5=s_byt5, clip=True, n_clip_image=n_ci)
            ref = conditioning_layout([mk(2), mk(3)])
            assert lay[1] == ref[1] and [list(map(tuple, d)) for d in lay[0]] == [list(map(tuple, d)) for d in ref[0]]
        del m
