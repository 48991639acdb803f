"""csrc/rm_peephole.py (the ISA peephole between the compiler and the assembler): VOP2 v_cndmask_b32 with a VGPR or an
inline constant as src0 is re-encoded as VOP3, everything else -- literals (VOP3 cannot hold them on gfx9-class targets),
SGPR operands, other instructions, comments -- is left alone; and the assembler accepts what it writes."""
import importlib.util
import os
import shutil
import subprocess
import tempfile

import pytest

from conftest import ROOT


def _mod():
    spec = importlib.util.spec_from_file_location("rm_peephole", os.path.join(ROOT, "raymarch_algo_compare_amd", "csrc", "rm_peephole.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_rewrites_only_what_is_legal():
    m = _mod()
    src = ["\tv_cndmask_b32_e32 v1, v2, v3, vcc\n",
           "\tv_cndmask_b32_e32 v10, 0, v11, vcc                         ; a comment\n",
           "\tv_cndmask_b32_e32 v4, 2.0, v5, vcc\n",
           "\tv_cndmask_b32_e32 v4, -1, v5, vcc\n",
           "\tv_cndmask_b32_e32 v6, 0x3ff00000, v7, vcc\n",          # literal: stays
           "\tv_cndmask_b32_e32 v6, 1.5, v7, vcc\n",                 # not an inline constant: stays
           "\tv_cndmask_b32_e32 v6, s4, v7, vcc\n",                  # SGPR + VCC: two scalar reads, stays
           "\tv_cndmask_b32_e64 v8, v9, v10, s[4:5]\n",
           "\tv_add_f64 v[0:1], v[2:3], v[4:5]\n"]
    out, changed, kept = m.rewrite(src)
    assert changed == 4 and kept == 3
    assert out[0] == "\tv_cndmask_b32_e64 v1, v2, v3, vcc\n"
    assert out[1].startswith("\tv_cndmask_b32_e64 v10, 0, v11, vcc") and "; a comment" in out[1]
    assert out[2] == "\tv_cndmask_b32_e64 v4, 2.0, v5, vcc\n" and out[3] == "\tv_cndmask_b32_e64 v4, -1, v5, vcc\n"
    assert out[4:] == src[4:]


def test_the_assembler_accepts_the_rewritten_forms():
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if not os.path.exists(clang) or shutil.which("python3") is None:
        pytest.skip("no ROCm LLVM here")
    m = _mod()
    src = ["\t.text\n", "\t.globl k\n", "k:\n", "\tv_cndmask_b32_e32 v1, v2, v3, vcc\n", "\tv_cndmask_b32_e32 v1, 1.0, v3, vcc\n",
           "\tv_cndmask_b32_e32 v1, 64, v3, vcc\n", "\tv_cndmask_b32_e32 v1, -16, v3, vcc\n", "\ts_endpgm\n"]
    out, changed, _ = m.rewrite(src)
    assert changed == 4
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "k.s")
        open(p, "w").writelines(out)
        subprocess.run([clang, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", p, "-o", os.path.join(td, "k.o")], check=True)
        dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", os.path.join(td, "k.o")], capture_output=True, text=True, check=True).stdout
        assert dis.count("v_cndmask_b32_e64") == 4
