"""Register / scratch budgets the design relies on, checked at COMPILE time (hipcc cross-compiles for
gfx950 without a GPU; `-Rpass-analysis=kernel-resource-usage`):

  * no production kernel spills to scratch (a spill in the MFMA sweep of predict_ranks or in the
    tile kernel's gather would put memory round trips into their inner loops);
  * the LDS-DMA tile kernel fits three workgroups per CU (<= 168 VGPRs: 12 wavefronts per CU is
    where its throughput comes from), the register-staged variant two;
  * the MFMA ranks kernel runs two wavefronts per SIMD for d <= 64.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lightfm_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
         "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c"]


def _hipcc():
    for c in ("/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    return None


def _usage(source, tmp_path):
    hipcc = _hipcc()
    if hipcc is None:
        pytest.skip("hipcc not available")
    out = subprocess.run([hipcc] + FLAGS + [os.path.join(CSRC, source), "-o", str(tmp_path / "x.o")],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    kernels, name = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"remark: +Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            kernels[name] = {}
            continue
        m = re.search(r"remark: +(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m and name:
            kernels[name][m.group(1).split(" ")[0]] = int(m.group(2))
    assert kernels, "no resource remarks in the compiler output"
    return kernels


def _one(kernels, fragment):
    hits = [v for k, v in kernels.items() if fragment in k]
    assert len(hits) == 1, (fragment, sorted(kernels))
    return hits[0]


@pytest.mark.timeout(1200)
def test_tile_kernel_budgets(tmp_path):
    k = _usage("warp_tile_lpr16.hip", tmp_path)
    for name, u in k.items():
        assert u["ScratchSize"] == 0, (name, u)
    # fit_warp_tile_kernel<16, 4, TIMED = false, ADADELTA = false, DMA4 = true / false, REG = false / true>
    dma = _one(k, "fit_warp_tile_kernelILi16ELi4ELb0ELb0ELb1ELb0ELi1EEE")
    regs = _one(k, "fit_warp_tile_kernelILi16ELi4ELb0ELb0ELb0ELb0ELi1EEE")
    assert dma["VGPRs"] + dma["AGPRs"] <= 148 and dma["Occupancy"] == 3, dma  # round 2: 163, round 3: 145
    assert regs["Occupancy"] >= 2 and dma["VGPRs"] < regs["VGPRs"], (dma, regs)
    # the L2-regularised variant (item_alpha / user_alpha != 0) keeps the third workgroup per CU
    reg = _one(k, "fit_warp_tile_kernelILi16ELi4ELb0ELb0ELb1ELb1ELi1EEE")
    assert reg["VGPRs"] + reg["AGPRs"] <= 168 and reg["Occupancy"] == 3, reg


@pytest.mark.timeout(1200)
def test_bpr_and_logistic_tile_kernel_budgets(tmp_path):
    """warp_tile_bpr.hip: <lanes per row, 4, TIMED, ADADELTA, DMA4, REG, loss id (2 = BPR, 0 = logistic)>.  The LDS-DMA variants keep the
    third workgroup per CU; the BPR ones pay for it with a few spilled pointers (<= 48 / 80 bytes per lane without / with the lazy
    regularisation, reloaded in the update); every other instantiation runs without scratch at two or three workgroups per CU."""
    k = _usage("warp_tile_bpr.hip", tmp_path)
    assert len(k) == 16, sorted(k)
    for name, u in k.items():
        dma, reg, bpr = "ELb0ELb0ELb1ELb" in name, "ELb1ELi" in name.split("ELb0ELb0ELb", 1)[1], "ELi2EEE" in name
        assert u["ScratchSize"] <= ((80 if reg else 48) if (dma and bpr) else 0), (name, u)
        assert u["Occupancy"] >= (3 if dma else 2), (name, u)


@pytest.mark.timeout(1200)
def test_gather_ahead_kernel_budget_and_waits(tmp_path):
    """The steady-state tile kernel (warp_tile_ahead.hpp): no scratch, three workgroups per CU, and -- what the
    variant is about -- no wait for outstanding memory operations between the top of a pass and the gather of the
    next one (a compiler-placed vmcnt(0) there would wait for the previous pass's atomics)."""
    k = _usage("warp_tile_ahead.hip", tmp_path)
    # <candidates, owner-sharded items, user rows by plain stores, floats of a row per lane>
    FRAGS = ("fit_warp_tile_ahead_kernelILi10ELb0ELb0ELi4EEE", "fit_warp_tile_ahead_kernelILi10ELb1ELb0ELi4EEE",
             "fit_warp_tile_ahead_kernelILi10ELb0ELb1ELi4EEE")
    for frag in ["fit_warp_tile_narrow_kernelILi10ELb%dELb%dELb%dEEE" % f for f in  # <NBF, USTORE, RP, BIN>
                 ((0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0), (0, 1, 1), (1, 1, 1))]:
        u = _one(k, frag)  # the narrow-model kernel (warp_tile_narrow.hpp): four workgroups per CU, no scratch
        assert u["ScratchSize"] == 0 and u["VGPRs"] + u["AGPRs"] <= 128 and u["Occupancy"] >= 4, (frag, u)
    for frag in FRAGS:
        u = _one(k, frag)
        # (three workgroups per CU -- the LDS tile's bound -- need <= 168 registers)
        assert u["ScratchSize"] == 0 and u["VGPRs"] + u["AGPRs"] <= 168 and u["Occupancy"] >= 3, (frag, u)
    bodies = _asm("warp_tile_ahead.hip", tmp_path)
    for frag in FRAGS:
        lines = [b for n, b in bodies.items() if frag in n][0].splitlines()
        dma = [i for i, l in enumerate(lines) if "global_load_lds_dword " in l]   # the bias DMAs end a gather
        assert len(dma) == 4, (frag, dma)                                         # prologue + in-loop gather, two each
        waits = [i for i, l in enumerate(lines) if "s_waitcnt" in l and "vmcnt" in l]
        prologue_wait = min(i for i in waits if i > dma[1])
        between = [i for i in waits if prologue_wait < i < dma[2]]
        assert not between, (frag, [lines[i] for i in between])
        atomics = [i for i, l in enumerate(lines) if "global_atomic_add_f32" in l]
        assert atomics and min(atomics) > dma[3], frag                            # publication follows the next pass's gather


@pytest.mark.timeout(1200)
def test_scoring_kernel_budgets(tmp_path):
    k = _usage("predict_kernels.hip", tmp_path)
    for name, u in k.items():
        assert u["ScratchSize"] == 0, (name, u)
    for ks in (16, 32):  # d <= 32, d <= 64: two wavefronts per SIMD (MFMA of one overlaps the compares of the other)
        u = _one(k, "ranks_mfma2_kernelILi%dEEE" % ks)
        assert u["Occupancy"] >= 2 and u["VGPRs"] + u["AGPRs"] <= 256, (ks, u)
    assert _one(k, "ranks_mfma2_kernelILi64EEE")["Occupancy"] >= 1
    # the bucket-search sweep (the default): three wavefronts per SIMD for d <= 64 (12.3 KB of LDS each), two for d <= 128
    # (Lb1: the products on the bf16 matrix pipe, the default; Lb0: fp32 products)
    for bf in (1, 0):
        for ks, occ in ((16, 3), (32, 3), (64, 2)):
            u = _one(k, "ranks_mfma3_kernelILi%dELb%dEEE" % (ks, bf))
            assert u["Occupancy"] >= occ and u["VGPRs"] + u["AGPRs"] <= 512 // occ and u["LDS"] <= 160 * 1024 // (4 * occ), (ks, bf, u)


@pytest.mark.timeout(1200)
def test_row_stream_kernel_budgets(tmp_path):
    k = _usage("feat_kernels.hip", tmp_path)
    for name, u in k.items():
        assert u["ScratchSize"] == 0, (name, u)
        assert u["Occupancy"] >= 2, (name, u)  # 8 wavefronts per CU are launched (csrc/session.hip)


def _asm(source, tmp_path):
    hipcc = _hipcc()
    if hipcc is None:
        pytest.skip("hipcc not available")
    flags = [f for f in FLAGS if f not in ("-c", "-Rpass-analysis=kernel-resource-usage")]
    out = subprocess.run([hipcc] + flags + ["-S", os.path.join(CSRC, source), "-o", str(tmp_path / "x.s")],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    text = open(str(tmp_path / "x.s")).read()
    bodies = {}
    for m in re.finditer(r"^(_ZN3lfm\w+):.*?s_endpgm", text, re.S | re.M):
        bodies[m.group(1)] = m.group(0)
    return bodies


@pytest.mark.timeout(1200)
def test_instruction_selection_of_the_hot_kernels(tmp_path):
    """What the kernels are DESIGNED around is what the compiler emitted: LDS-DMA gathers with no
    ds_write staging and hardware float atomics in the tile kernel; the matrix cores in predict_ranks."""
    tile = _asm("warp_tile_lpr16.hip", tmp_path)
    dma = [b for n, b in tile.items() if "fit_warp_tile_kernelILi16ELi4ELb0ELb0ELb1ELb0ELi1EEE" in n][0]
    regs = [b for n, b in tile.items() if "fit_warp_tile_kernelILi16ELi4ELb0ELb0ELb0ELb0ELi1EEE" in n][0]
    reg = [b for n, b in tile.items() if "fit_warp_tile_kernelILi16ELi4ELb0ELb0ELb1ELb1ELi1EEE" in n][0]
    # the regularised variant: the same LDS-DMA gathers, no scratch, no float64 exp / log library code
    # (v_exp_f32 / v_log_f32 only on the rare large-step paths), its scale state in ONE line of memory
    assert reg.count("global_load_lds_dwordx4") >= 12 and "scratch_" not in reg and "ds_write_b128" not in reg
    assert dma.count("global_load_lds_dwordx4") >= 12  # user rows (loop), positive row, up to 15 candidates
    assert "ds_write_b128" not in dma and regs.count("ds_write_b128") > 10
    assert dma.count("global_atomic_add_f32") >= 8 and "scratch_" not in dma
    assert "row_newbcast" in dma  # candidate ids reach the loading lanes by DPP
    ranks = _asm("predict_kernels.hip", tmp_path)
    sweep = [b for n, b in ranks.items() if "ranks_mfma2_kernelILi32EEE" in n][0]
    assert sweep.count("v_mfma_f32_32x32x2_f32") == 32  # d = 64: 32 steps of k = 2
    assert "scratch_" not in sweep and "v_pk_add_f32" in sweep
    # the bucket-search sweep: biases as a 33rd step, the item table through buffer loads with scalar row offsets (no
    # per-load address arithmetic), bucket counts by LDS atomics, every rank published by a hardware float atomic
    search = [b for n, b in ranks.items() if "ranks_mfma3_kernelILi32ELb0EEE" in n][0]  # fp32 products
    assert search.count("v_mfma_f32_32x32x2_f32") == 33 and "scratch_" not in search
    assert search.count("buffer_load_dword") >= 2 * 35 and search.count("ds_add_u32") >= 16
    assert "global_atomic_add_f32" in search
    # the default: the products on the bf16 matrix pipe -- (hi, lo) split operands, three products per 16 components (hi hi,
    # hi lo, lo hi: 4 steps of k = 16 at d = 64), the biases one fp32 step
    split = [b for n, b in ranks.items() if "ranks_mfma3_kernelILi32ELb1EEE" in n][0]
    assert split.count("v_mfma_f32_32x32x16_bf16") == 12 and split.count("v_mfma_f32_32x32x2_f32") == 1 and "scratch_" not in split
    assert split.count("ds_add_u32") >= 16 and "global_atomic_add_f32" in split
