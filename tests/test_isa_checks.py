"""Static checks of the compiled gfx950 code of the two kernels whose hand-off / hot loops rely on what the compiler emitted
(ADVICE round 3): the headline kernel's LDS hand-off is ordered only if data and flag are both LDS instructions of the wave --
a FLAT load or store there would be a silent race -- and the one-chain-per-wave orbit kernel must not use scratch (scratch-using
waves are admitted 64 per XCD on this stack).  hipcc cross-compiles without a GPU: this runs in the CPU suite."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "walnuts_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fno-strict-aliasing", "--offload-arch=gfx950", "-Wno-unused-result", "-Wno-unused-value",
         "--cuda-device-only", "-S"]


def kernels_of(unit, tmp_path, extra=()):
    out = tmp_path / (unit + ".s")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + list(extra) + ["-o", str(out), os.path.join(CSRC, unit + ".hip")],
                          stderr=subprocess.DEVNULL)
    txt = out.read_text()
    ks = {}
    for m in re.finditer(r"^(_ZN2wn\w+_kernel\w*):", txt, re.M):  # each kernel: from its label to the end of the function
        ks[m.group(1)] = txt[m.start():txt.index(".Lfunc_end", m.start())]  # (a kernel may hold more than one s_endpgm)
    meta = {}
    for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size: *\d+", txt, re.S):
        blk = m.group(0)
        g = lambda k: int((re.search(r"\." + k + r": *(\d+)", blk) or [None, "0"])[1])
        name = re.search(r"\.name: *(\S+)", blk).group(1)
        meta[name] = dict(scratch=g("private_segment_fixed_size"), vgpr_spill=g("vgpr_spill_count"), sgpr_spill=g("sgpr_spill_count"))
    return ks, meta


@pytest.mark.timeout(900)
def test_headline_kernel_has_no_flat_or_scratch_instructions(tmp_path):
    ks, meta = kernels_of("wn_k_fast4", tmp_path)
    hot = [k for k in ks if "wn_fast4_kernel" in k]
    assert hot
    for k in hot:
        body = ks[k]
        assert not re.search(r"\bflat_(load|store|atomic)", body), k
    # the plain (untraced) instantiations: no scratch, no vector-register spills
    for name, m in meta.items():
        if "wn_fast4_kernel" in name and re.search(r"wn_fast4_kernelILi\dELb[01]ELb0E", name):
            assert m["scratch"] == 0 and m["vgpr_spill"] == 0, (name, m)


    # halves mode (a group on two T waves, each alone on its SIMD): the fragments live in registers -- the tick bodies hold no 16-byte
    # fragment reads (the whole-group form has one per two MFMAs), and the code object still has no spills
    halves = "_ZN2wn15wn_fast4_kernelILi7ELb1ELb0ELb1ELb1ELb0EEEvNS_6ParamsEPii"
    whole = "_ZN2wn15wn_fast4_kernelILi7ELb1ELb0ELb1ELb0ELb0EEEvNS_6ParamsEPii"
    assert halves in ks and whole in ks
    n128 = lambda k: len(re.findall(r"\bds_read_b128", ks[k]))
    assert n128(whole) >= 150 and n128(halves) <= 4, (n128(whole), n128(halves))
    assert min(len(re.findall(r"v_mfma_f64_4x4x4", ks[k])) for k in (halves, whole)) >= 314  # 157 per tick body, two bodies


@pytest.mark.timeout(900)
def test_headline_kernel_warmup_instantiations_have_no_scratch(tmp_path):
    """Round 5: the instantiations that also run warm-up adaptation (wn_k_fast4w.o: -DWN4_WARM_TU; RetireBookT<true> with the P2 markers
    in registers) -- no FLAT instruction in the hand-off, no scratch, no vector-register spills (the first form of wn_p2_push stored
    through a run-time choice of two elements' addresses and kept the markers in scratch: select form now)."""
    ks, meta = kernels_of("wn_k_fast4", tmp_path, extra=["-DWN4_WARM_TU=1"])
    hot = [k for k in ks if "wn_fast4_kernel" in k]
    assert hot and all(k.endswith("ELb1EEEvNS_6ParamsEPii") for k in hot), hot[:3]  # WARM = true, every one of them
    for k in hot:
        assert not re.search(r"\bflat_(load|store|atomic)", ks[k]), k
    for name, m in meta.items():
        if "wn_fast4_kernel" in name:
            assert m["scratch"] == 0 and m["vgpr_spill"] == 0, (name, m)


@pytest.mark.timeout(900)
def test_orbit_kernel_has_no_flat_or_scratch_instructions(tmp_path):
    """wn_orbit_kernel<Tgt, INTEG, FUSED, NE, K, WARM> (mangled: ...ELi<INTEG>ELb<FUSED>ELi<NE>ELi<K>ELb<WARM>EEEv...): no FLAT access anywhere
    (ring, slots and proposals are LDS- or global-address-space accesses); the one-register instantiations (NE = 1: funnel10 and the 2-D
    targets, BASELINE config 3) keep everything in registers, with or without the warm-up code; so do the plain wide ones of the Gaussians
    up to 4 registers per lane.  Known and measured: StockWatson's adaptive one-wave instantiations (three states of 36 doubles around a
    backward search: not routed by default), the dense Gaussian's WARM ones (256-register cap at two waves per SIMD: 2-4 spilled registers
    outside the micro-step loop)."""
    ks, meta = kernels_of("wn_k_orbit", tmp_path)
    hot = [k for k in ks if "wn_orbit_kernel" in k]
    assert len(hot) >= 40, len(hot)
    one_reg = lambda k: re.search(r"ELb[01]ELi1ELi1ELb[01]EEEv", k) is not None
    assert sum(one_reg(k) for k in hot) >= 30
    for k in hot:
        assert not re.search(r"\bflat_(load|store|atomic)", ks[k]), k
        plain_small_wide = re.search(r"(WTgtDiagILi[24]E|WTgtDenseLDS).*ELb0EEEv", k) is not None
        sw_fixed = "WTgtStockWatson" in k and "EELi0ELb0ELi12E" in k
        if one_reg(k) or plain_small_wide or sw_fixed:
            assert not re.search(r"\bscratch_", ks[k]), k
    for name, m in meta.items():
        if "wn_orbit_kernel" in name and one_reg(name):
            assert m["scratch"] == 0 and m["vgpr_spill"] == 0, (name, m)
    # the tolerated instantiations have explicit ceilings (an exemption hides a future spill inside a micro-step loop):
    # dense-LDS WARM forms (the default route of adaptive runs with warm-up on dense Gaussians up to 128-D; 256-register cap at two waves per
    # SIMD) <= 4 spilled registers / 32 B; the wide diagonal forms <= 4 / 0 B; StockWatson's adaptive one-wave forms (not routed by default)
    # <= 56 / 224 B; every plain dense-LDS form (the chain-queue launch of round 6) none at all
    for name, m in meta.items():
        if "wn_orbit_kernel" not in name or one_reg(name):
            continue
        if "WTgtDenseLDS" in name:
            cap = (4, 32) if name.endswith("ELb1EEEvNS_6ParamsE") else (0, 0)
        elif "WTgtStockWatson" in name:
            cap = (4, 0) if "EELi0ELb0ELi12E" in name else (56, 224)
        else:
            cap = (4, 0)
        assert m["vgpr_spill"] <= cap[0] and m["scratch"] <= cap[1], (name, m, cap)


@pytest.mark.timeout(900)
def test_two_wave_orbit_kernel_hand_off_is_lds_and_scratch_free(tmp_path):
    """wn_k_orbit2.hip (round 5: StockWatson's fixedLeapFrog run, trajectory and bookkeeping on a wave each): the hand-off words and the leaf
    records are LDS-address-space accesses (a FLAT access there would wait on both counters and could be reordered against the fences),
    and the default instantiation -- fixedLeapFrog -- keeps everything in registers."""
    ks, meta = kernels_of("wn_k_orbit2", tmp_path)
    hot = [k for k in ks if "wn_orbit2_kernel" in k]
    assert len(hot) == 3, hot
    for k in hot:
        assert not re.search(r"\bflat_(load|store|atomic)", ks[k]), k
        assert re.search(r"\bds_(read|write|load|store)", ks[k]), k
    fixed = [k for k in hot if k.endswith("ELi12ELi0EEEvNS_6ParamsEPi")]
    assert len(fixed) == 1, hot
    assert not re.search(r"\bscratch_", ks[fixed[0]])
    # the adaptive instantiations (config 5's adaptLeapFrogD / R2P runs: four waves of 512 registers) spill: bounded, and not inside the
    # innermost loops' bodies beyond what round 6 found there (profiles/r06_stockwatson_summary.txt: 43-54 registers, 160-208 B)
    for name, m in meta.items():
        if "wn_orbit2_kernel" in name and not name.endswith("ELi12ELi0EEEvNS_6ParamsEPi"):
            assert m["vgpr_spill"] <= 60 and m["scratch"] <= 224, (name, m)
    for name, m in meta.items():
        if name.endswith("ELi12ELi0EEEvNS_6ParamsEPi"):
            assert m["scratch"] == 0, (name, m)
