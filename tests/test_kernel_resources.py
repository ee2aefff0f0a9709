"""CPU: what hipcc made of the kernels (the -Rpass-analysis=kernel-resource-usage remarks build.py keeps next to every object).
No kernel of the product library may spill to scratch, and the host's hand-kept table of resident ring-tile workgroups (gemm.hip: ring_resident, which
sizes the batch-1 launches) must not promise more workgroups per CU than registers, LDS and the SGPR admission rule of MI355X_MICROARCH.md allow."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "paella_amd", "csrc", "build")


def _kernels(built_lib):
    files = [f for f in os.listdir(BUILD)] if os.path.isdir(BUILD) else []
    files = [f for f in files if f.endswith(".resources.txt")]
    if not files:
        pytest.skip("no compiler remarks next to the objects (library reused from a snapshot without its build directory)")
    out = []
    for f in files:
        txt = open(os.path.join(BUILD, f)).read()
        for blk in re.split(r"remark: Function Name: ", txt)[1:]:
            name = blk.split()[0]

            def g(key):
                m = re.search(re.escape(key) + r": (\d+)", blk)
                return int(m.group(1)) if m else None
            out.append(dict(unit=f, name=name, vgpr=g("VGPRs"), agpr=g("AGPRs"), sgpr=g("SGPRs"), scratch=g("ScratchSize [bytes/lane]"),
                            occupancy=g("Occupancy [waves/SIMD]"), lds=g("LDS Size [bytes/block]")))
    return out


def test_no_kernel_uses_scratch(built_lib):
    ks = _kernels(built_lib)
    assert len(ks) > 150, "expected the remarks of every instantiation (%d found)" % len(ks)
    spilled = [(k["unit"], k["name"], k["scratch"]) for k in ks if k["scratch"]]
    assert not spilled, "kernels spill to scratch: %s" % spilled


GEMM = re.compile(r"_Z14gemm_nt_kernelILi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELb(\d)ELi(\d+)ELb(\d)ELi(\d)ELb(\d)E")
RING_ID = {(1, 1, 3): 30, (1, 1, 4): 31, (1, 2, 3): 32, (2, 1, 3): 33, (2, 2, 3): 34, (1, 2, 4): 35}


def test_ring_tile_residency_table_matches_the_compiler(built_lib):
    """256-thread workgroups: one wave per SIMD each, so workgroups per CU = min(occupancy in waves per SIMD -- registers and LDS --, 8,
    floor(800 / (ceil(sgpr / 16) * 16 + 16))) (MI355X_MICROARCH.md, residency and cooperative launch); x 256 CUs."""
    seen = 0
    for k in _kernels(built_lib):
        m = GEMM.match(k["name"])
        if not m:
            continue
        wm, wn, tm, tn, pd, apro, tail, bk, dma, ring, bf = (int(v) for v in m.groups())
        if ring == 0 or wm * wn != 4 or bf:
            continue
        cfg = RING_ID[(tm, tn, ring)]
        per_cu = min(k["occupancy"], 8, 800 // (-(-k["sgpr"] // 16) * 16 + 16), (160 * 1024) // max(k["lds"], 1))
        table = built_lib.paella_test_ring_resident(cfg, {0: 0, 1: 1, 4: 1, 2: 2}[apro])
        assert table > 0 and table % 256 == 0
        assert per_cu * 256 >= table, "ring tile %d (prologue %d): the host launches up to %d workgroups as resident, the kernel admits %d per CU (%s)" % (cfg, apro, table, per_cu, k)
        seen += 1
    assert seen >= 20


def test_attention_unpadded_layout_really_fits_four_workgroups_per_cu(built_lib):
    """attention_lds_kernel<5, 2> (head_dim 80, the 1B model): the unpadded direct-to-LDS layout exists to run FOUR workgroups per CU -- that needs <= 128 VGPRs, exactly
    40 960 B of LDS (no other static LDS in the kernel) and the SGPR admission rule to allow 4 waves per SIMD; the padded layout of the same head_dim stops at three."""
    ks = {k["name"]: k for k in _kernels(built_lib) if k["name"].startswith("_Z20attention_lds_kernel")}
    unp, pad = ks["_Z20attention_lds_kernelILi5ELi2EEv8AttnArgs"], ks["_Z20attention_lds_kernelILi5ELi1EEv8AttnArgs"]
    per_cu = lambda k: min(k["occupancy"], 8, 800 // (-(-k["sgpr"] // 16) * 16 + 16), (160 * 1024) // k["lds"])
    assert unp["lds"] == 40960 and unp["vgpr"] + (unp["agpr"] or 0) <= 128 and per_cu(unp) == 4, unp
    assert per_cu(pad) == 3, pad
