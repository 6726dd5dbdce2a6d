"""Register budgets the overlapped optimiser depends on, read from the built library's gfx950 code objects (no GPU).

The background AdamW (coral_amd/csrc/misc.hip adamw_kernel, one workgroup per CU) runs UNDER the next step's forward
GEMMs only while its waves fit into the registers a persistent kernel-X workgroup (two waves per SIMD) leaves free:
2 x alloc(X, NT form) + alloc(AdamW) <= 512 per lane and SIMD, alloc = the count rounded up to the granule of 8.  Two
registers too many in kernel X (225 -> 232 allocated) cost 19 ms per XLS-R-2B step in round 5 with every parity test
green (NOTEBOOK.md R5.6), so the budget is a test."""
import os
import re
import struct
import subprocess
import tempfile
from pathlib import Path

import pytest

LLVM = Path("/opt/rocm/lib/llvm/bin")
LIB = Path(__file__).resolve().parents[1] / "coral_amd" / "libcoral_amd.so"


def kernel_vgprs(lib):
    """{kernel symbol: .vgpr_count} over every gfx950 code object bundled in the library's .hip_fatbin section."""
    out = {}
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run([str(LLVM / "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", str(lib), fat], check=True)
        blob = Path(fat).read_bytes()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        pos, n = blob.find(magic), 0
        while pos >= 0:
            (entries,) = struct.unpack_from("<Q", blob, pos + 24)
            p = pos + 32
            for _ in range(entries):
                off, size, tl = struct.unpack_from("<QQQ", blob, p)
                triple = blob[p + 24:p + 24 + tl].decode()
                p += 24 + tl
                if "gfx950" in triple and size:
                    co = os.path.join(td, f"co{n}.o")
                    n += 1
                    Path(co).write_bytes(blob[pos + off:pos + off + size])
                    notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", co], capture_output=True, text=True,
                                           check=True).stdout
                    for m in re.finditer(r"\.name:\s+(\S+).*?\.vgpr_count:\s+(\d+)", notes, re.S):
                        out[m.group(1)] = int(m.group(2))
            pos = blob.find(magic, pos + 24)
    return out


def alloc(n):
    return -(-n // 8) * 8


@pytest.mark.skipif(not (LLVM / "llvm-readelf").exists() or not LIB.exists(), reason="needs the built library and llvm-readelf")
def test_background_adamw_fits_beside_the_forward_gemm_kernel():
    regs = kernel_vgprs(LIB)
    assert len(regs) > 100, "code objects not found in the library"
    x_nt = [v for k, v in regs.items() if "ca_gemm_kernel_xILi0ELi0ELb0E" in k]
    adamw = [v for k, v in regs.items() if "adamw_kernel" in k]
    assert len(x_nt) == 1 and len(adamw) == 4, (x_nt, adamw)  # (non-temporal or not) x (fp32 or bf16 gradient)
    assert 2 * alloc(x_nt[0]) + alloc(max(adamw)) <= 512, (x_nt, adamw)


@pytest.mark.gpu
def test_the_loaded_code_object_reports_the_same_budget():
    """ca_background_update_fits (what the trainer asks at start-up, falling back to the full-grid update if the answer
    is no) reads the registers of the LOADED kernels: the same numbers as the code objects' notes above."""
    from coral_amd import ops

    fits, rx, ru = ops.background_update_fits()
    regs = kernel_vgprs(LIB)
    x_nt = [v for k, v in regs.items() if "ca_gemm_kernel_xILi0ELi0ELb0E" in k][0]
    upd = [v for k, v in regs.items() if "adamw_kernelILb1ELb1E" in k][0]
    print(f"\nforward GEMM kernel {rx} registers (notes {x_nt}), background update {ru} (notes {upd}), fits: {fits}")
    assert (rx, ru) == (x_nt, upd) and fits
