"""Properties of the SHIPPED code object that the parity results rely on (no GPU needed: the library's device code is
disassembled with llvm-objdump).

Packed fp32 (`v_pk_mul_f32` / `v_pk_fma_f32` / `v_pk_add_f32`): a compiler-formed pair of them in the RoPE epilogue of
the dense q/k/v GEMM stored a wrong product term while a second process shared the GPU (profiles/README.md, "A wrong
product term under GPU sharing").  The library is therefore built with the SLP vectoriser off and with the target feature
packed-fp32-ops switched off for every translation unit but attn_flash.hip (csrc/Makefile: NO_PK); this test pins that
no such instruction exists outside `attn_flash_bf16_kernel`, where it is written out by hand, so that dropping a flag
or a compiler bump cannot bring it back unnoticed.  The same check runs at build time (`make check-pk`).
"""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
LIB = ROOT / "vall-e_amd" / "libvalle_hip.so"
OBJDUMP = Path("/opt/rocm/lib/llvm/bin/llvm-objdump")


def _device_disassembly(tmp_path):
    lib = tmp_path / "lib.so"
    shutil.copy(LIB, lib)
    subprocess.run([str(OBJDUMP), "--offloading", lib.name], cwd=tmp_path, check=True, capture_output=True)
    objs = sorted(tmp_path.glob("lib.so.*gfx950"))
    assert objs, "no gfx950 code object inside libvalle_hip.so"
    return [(o.name, subprocess.run([str(OBJDUMP), "-d", str(o)], check=True, capture_output=True, text=True).stdout) for o in objs]


@pytest.mark.skipif(not LIB.exists() or not OBJDUMP.exists(), reason="needs the built library and llvm-objdump")
def test_no_packed_fp32_outside_the_flash_attention(tmp_path):
    fn_re = re.compile(r"^[0-9a-f]+ <(.*)>:$")
    bad, seen_flash, n_kernels = {}, 0, 0
    for name, text in _device_disassembly(tmp_path):
        fn = None
        for line in text.splitlines():
            m = fn_re.match(line)
            if m:
                fn = m.group(1)
                n_kernels += 1
                continue
            if re.search(r"\bv_pk_(mul|fma|add)_f32\b", line):
                if fn and "attn_flash_bf16_kernel" in fn:
                    seen_flash += 1
                else:
                    bad[fn] = bad.get(fn, 0) + 1
    assert n_kernels > 50, n_kernels                 # the disassembly really covers the library's kernels
    assert seen_flash > 0                            # the pattern does match what it is meant to match
    assert not bad, f"packed fp32 outside attn_flash_bf16_kernel: {bad}"
