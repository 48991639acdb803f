#!/usr/bin/env python3
"""ISA peephole over the gfx950 assembly hipcc emits for the kernels (csrc/Makefile runs it between -S and the assembler).

v_cndmask_b32 in its short VOP2 encoding (`v_cndmask_b32_e32 vD, src0, vS1, vcc`) is SERIALIZED ACROSS THE WAVEFRONTS OF A
COMPUTE UNIT on gfx950 when such instructions follow one another: measured (tools/ubench/simd_share.hip) 6 cycles per
instruction with one wave, 8 / 12 / 16 with two / three / four waves on DIFFERENT SIMDs, while the same select in the VOP3
encoding (`v_cndmask_b32_e64 ..., vcc` or with an SGPR-pair mask) costs 5.8 at any wave count -- as do v_bfi, v_cmp, fp64
arithmetic and everything else tried.  The exact libm restatements (rm_math_*.h) are branch-free by design: a trip of the
Mandelbulb SDF holds ~170 selects, half of them shrunk to the VOP2 form by the compiler, in runs (a double is two
selects).  This pass re-encodes them as VOP3 wherever that is legal: src0 a VGPR or an inline constant (VOP3 cannot hold a
32-bit literal on gfx9-class targets; those few stay).  Same operation, same operands, same results -- four more bytes.
"""
import re
import sys

INLINE_INT = set(str(i) for i in range(-16, 65))
INLINE_FP = {"0.5", "-0.5", "1.0", "-1.0", "2.0", "-2.0", "4.0", "-4.0", "0.15915494", "0"}
PAT = re.compile(r"^(\s*)v_cndmask_b32_e32(\s+)(v\d+),\s*([^,]+),\s*(v\d+),\s*vcc(\s*(?:;.*)?)$")


def legal_src0(tok: str) -> bool:
    tok = tok.strip()
    return bool(re.fullmatch(r"v\d+", tok)) or tok in INLINE_INT or tok in INLINE_FP


def rewrite(lines):
    changed = kept = 0
    out = []
    for line in lines:
        m = PAT.match(line.rstrip("\n"))
        if m and legal_src0(m.group(4)):
            out.append(f"{m.group(1)}v_cndmask_b32_e64{m.group(2)}{m.group(3)}, {m.group(4).strip()}, {m.group(5)}, vcc{m.group(6)}\n")
            changed += 1
        else:
            if "v_cndmask_b32_e32" in line:
                kept += 1
            out.append(line)
    return out, changed, kept


def main():
    src, dst = sys.argv[1], sys.argv[2]
    with open(src) as f:
        lines = f.readlines()
    out, changed, kept = rewrite(lines)
    with open(dst, "w") as f:
        f.writelines(out)
    print(f"rm_peephole: {changed} v_cndmask_b32 re-encoded as VOP3, {kept} left in VOP2 (literal operand)", file=sys.stderr)


if __name__ == "__main__":
    main()
