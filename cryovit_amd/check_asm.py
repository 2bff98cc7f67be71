"""Static check of the persistent GEMM tile kernels (csrc/gemm256p.h) -- and, for the M0 / DMA rules, of every other kernel
that uses the LDS-DMA helpers -- in the emitted gfx950 assembly.

The kernel counts its own vector-memory operations in hand-written ``s_waitcnt vmcnt(N)`` waits, so three properties of
the COMPILED code are part of its correctness and are asserted here (``python -m cryovit_amd.check_asm``, run by the CPU
test suite -- hipcc cross-compiles without a GPU):

  * no scratch: a register spill is a scratch store / load, i.e. an uncounted entry in the in-order vmcnt queue;
  * the epilogue of an interior-tile ("FULL") kernel issues exactly ``epi_stores_per_wave`` global stores (8 SwiGLU, 16 bf16 /
    fp16, 32 fp32, 33 bf16 hi / lo pair + row statistics) -- hipcc splits and merges the stores it generates itself, the kernel's come from inline asm;
  * every LDS-DMA is the inline-asm saddr form (``global_load_lds_dwordx4 vOFF, s[BASE]``): the builtin form would let hipcc
    cache what it believes M0 holds across the asm statements that rewrite it.
"""

from __future__ import annotations

import re
import subprocess
import sys
import tempfile
from pathlib import Path

from cryovit_amd.build import ARCH, CSRC, FILE_FLAGS, FLAGS, INCLUDE, hipcc_path

EXPECTED_STORES = {"EpiSwiGLU": 8, "EpiBF16": 16, "EpiResidT": 32, "EpiResidHL": 33, "EpiVT": 16}


def compile_asm(src: Path) -> str:
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / (src.stem + ".s")
        flags = [f for f in FLAGS if f != "-fPIC"]
        cmd = [hipcc_path(), *flags, *FILE_FLAGS.get(src.name, []), "-I", str(INCLUDE), "-S", "--cuda-device-only", "-o", str(out), str(src)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc -S failed on {src.name}:\n{r.stderr}")
        return out.read_text()


def check_gemm256p(asm: str) -> list[str]:
    """One line per persistent kernel: name, VGPRs, stores; raises AssertionError on a violated property."""
    bodies = {m.group(1): m.group(2) for m in re.finditer(r"^(_ZN3cvx15k_gemm256p_[nm]reg\w+):.*?\n(.*?)s_endpgm", asm, re.S | re.M)}
    assert bodies, "no k_gemm256p kernels in the assembly"
    report = []
    for m in re.finditer(r"\.amdhsa_kernel (_ZN3cvx15k_gemm256p_[nm]reg\w+)(.*?)\.end_amdhsa_kernel", asm, re.S):
        name, desc = m.group(1), m.group(2)
        body = bodies[name]
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", desc).group(1))
        vgpr = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", desc).group(1))
        assert scratch == 0 and "scratch_" not in body, f"{name}: spills to scratch ({scratch} B): uncounted vmcnt entries"
        assert vgpr <= 256, f"{name}: {vgpr} VGPRs"
        dma = re.findall(r"global_load_lds_dwordx4 (.*)", body)
        assert dma and all(re.fullmatch(r"v\d+, s\[\d+:\d+\]", d.strip()) for d in dma), f"{name}: LDS-DMA not in saddr form: {set(dma)}"
        stores = len(re.findall(r"global_store_", body))
        full = re.search(r"ELb1E(Lb[01]E)?EEv", name) is not None  # template args <Epi, FULL = true[, DBG]>
        kind = next(k for k in EXPECTED_STORES if k in name)
        if full:
            assert stores == EXPECTED_STORES[kind], f"{name}: {stores} store instructions, the waits assume {EXPECTED_STORES[kind]}"
            assert all(s.startswith("global_store_dwordx4") for s in re.findall(r"global_store_\w+", body)), name
        report.append(f"{name[24:70]:48s} vgpr {vgpr:3d} stores {stores:2d} dma {len(dma):2d} {'FULL' if full else 'edge'}")
    return report


def check_dma_users(asm: str, fname: str) -> list[str]:
    """Every other kernel that uses the LDS-DMA helpers of common.h (attention, the marching / tiled convolutions, the one-shot
    GEMM tiles): the DMA instructions and every access to M0 sit inside inline-asm statements (hipcc brackets those with
    ``;;#ASMSTART`` / ``;;#ASMEND``), i.e. the compiler never emits a DMA of its own (the builtin form) and never writes or
    caches M0 itself; and no kernel of the file spills to scratch."""
    report, inside, kernel, n_dma = [], False, None, {}
    for ln in asm.splitlines():
        t = ln.strip()
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            kernel = m.group(1)
        if "#ASMSTART" in t:
            inside = True
        elif "#ASMEND" in t:
            inside = False
        elif t and not t.startswith((";", ".", "//")):
            if "global_load_lds" in t:
                n_dma[kernel] = n_dma.get(kernel, 0) + 1
            if not inside:
                assert "global_load_lds" not in t, f"{fname}:{kernel}: compiler-generated LDS-DMA: {t}"
                assert not re.search(r"\bm0\b", t), f"{fname}:{kernel}: M0 touched outside inline asm: {t}"
    for m in re.finditer(r"\.amdhsa_kernel (\w+)(.*?)\.end_amdhsa_kernel", asm, re.S):
        name, desc = m.group(1), m.group(2)
        if name not in n_dma:
            continue
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", desc).group(1))
        vgpr = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", desc).group(1))
        assert scratch == 0, f"{fname}:{name}: {scratch} B of scratch in a kernel with counted vector-memory waits"
        report.append(f"{fname:16s} {name[:60]:60s} vgpr {vgpr:3d} dma {n_dma[name]:3d}")
    return report


DMA_SOURCES = ("attention.hip", "conv_halo.hip")  # (conv_halo: kernels only in the ablation build)


def main() -> None:
    gemm_asm = compile_asm(CSRC / "gemm.hip")
    for line in check_gemm256p(gemm_asm):
        print(line)
    for line in check_dma_users(gemm_asm, "gemm.hip"):
        print(line)
    for f in DMA_SOURCES:
        for line in check_dma_users(compile_asm(CSRC / f), f):
            print(line)
    print("ok")


if __name__ == "__main__":
    assert ARCH == "gfx950"
    main()
