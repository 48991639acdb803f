#!/usr/bin/env python3
"""Extract the lookup tables of glibc 2.35's fp64 libm routines into a C header.

The reference computes `x ** y`, math.sin/cos/acos/atan2/log through CPython ->
glibc libm (Ubuntu GLIBC 2.35-0ubuntu3.11, x86-64, FMA ifunc variants).  libm is a
third-party dependency that is not part of /root/reference; to restate its
published algorithms bit-for-bit on the GPU the kernels need the same constant
tables.  This script locates each table inside the system's libm.so.6 by anchor
constants (never by address) and writes raymarch_algo_compare_amd/csrc/rm_libm_tables.h.
The generated header is committed; this script documents where it came from.
"""
import struct
import sys

LIBM = "/lib/x86_64-linux-gnu/libm.so.6"
OUT = "raymarch_algo_compare_amd/csrc/rm_libm_tables.h"


def d2b(x: float) -> bytes:
    return struct.pack("<d", x)


def find(blob: bytes, anchor: bytes, what: str, nbytes: int = 0) -> int:
    """Offset of the table starting with `anchor`.  Static tables are emitted once per multiarch
    variant (sse2 / fma / fma4); several hits are accepted when all copies are byte-identical
    over the `nbytes` that will be read."""
    i = blob.find(anchor)
    if i < 0:
        sys.exit(f"anchor for {what} not found")
    j = blob.find(anchor, i + 1)
    while j >= 0:
        if not nbytes or blob[j:j + nbytes] != blob[i:i + nbytes]:
            sys.exit(f"anchor for {what} is ambiguous")
        j = blob.find(anchor, j + 1)
    return i


def doubles(blob, off, n):
    return struct.unpack_from(f"<{n}d", blob, off)


def u64s(blob, off, n):
    return struct.unpack_from(f"<{n}Q", blob, off)


def fmt_d(vals, per=2):
    out = []
    for i in range(0, len(vals), per):
        out.append("    " + ", ".join(float.hex(v) for v in vals[i:i + per]) + ",")
    return "\n".join(out)


def fmt_u(vals, per=2):
    out = []
    for i in range(0, len(vals), per):
        out.append("    " + ", ".join(f"0x{v:016x}ull" for v in vals[i:i + per]) + ",")
    return "\n".join(out)


def main():
    blob = open(LIBM, "rb").read()
    parts = []

    # __pow_log_data: ln2hi, ln2lo, poly[7], tab[128]{invc, pad, logc, logctail}
    off = find(blob, d2b(float.fromhex("0x1.62e42fefa3800p-1")) + d2b(float.fromhex("0x1.ef35793c76730p-45"))
               + d2b(-0.5), "pow_log_data")
    head = doubles(blob, off, 9)
    tab = doubles(blob, off + 72, 128 * 4)
    assert tab[64 * 4 + 0] != 0.0
    parts.append("// __pow_log_data (sysdeps/ieee754/dbl-64/e_pow_log_data.c): ln2hi, ln2lo, poly[7]\n"
                 "RM_TAB double rm_pow_log_head[9] = {\n" + fmt_d(head, 3) + "\n};\n"
                 "// tab[128] = {invc, pad, logc, logctail}\n"
                 "RM_TAB double rm_pow_log_tab[512] = {\n" + fmt_d(tab, 4) + "\n};\n")

    # __exp_data: invln2N, shift, negln2hiN, negln2loN, poly[4], exp2_shift, exp2_poly[5], tab[256]
    off = find(blob, d2b(float.fromhex("0x1.71547652b82fep0") * 128) + d2b(float.fromhex("0x1.8p52")), "exp_data")
    head = doubles(blob, off, 14)
    tab = u64s(blob, off + 14 * 8, 256)
    assert tab[1] == 0x3ff0000000000000 and tab[0] == 0
    parts.append("// __exp_data (sysdeps/ieee754/dbl-64/e_exp_data.c): invln2N, shift, negln2hiN, negln2loN,\n"
                 "// poly[4], exp2_shift, exp2_poly[5]\n"
                 "RM_TAB double rm_exp_head[14] = {\n" + fmt_d(head, 2) + "\n};\n"
                 "// tab[2*128]: {tail bits, scale bits - (i << 45)}\n"
                 "RM_TAB uint64_t rm_exp_tab[256] = {\n" + fmt_u(tab, 2) + "\n};\n")

    # __log_data (e_log_data.c): ln2hi, ln2lo, poly[5], poly1[11], tab[128]{invc, logc}
    off = find(blob, d2b(float.fromhex("0x1.62e42fefa3800p-1")) + d2b(float.fromhex("0x1.ef35793c76730p-45"))
               + d2b(float.fromhex("-0x1.0000000000001p-1")), "log_data")
    head = doubles(blob, off, 18)
    tab = doubles(blob, off + 18 * 8, 256)
    assert abs(tab[0] * 0.7 - 1.0) < 0.1
    parts.append("// __log_data (sysdeps/ieee754/dbl-64/e_log_data.c): ln2hi, ln2lo, poly[5], poly1[11]\n"
                 "RM_TAB double rm_log_head[18] = {\n" + fmt_d(head, 3) + "\n};\n"
                 "// tab[128] = {invc, logc}\n"
                 "RM_TAB double rm_log_tab[256] = {\n" + fmt_d(tab, 2) + "\n};\n")

    # __sincostab (sysdeps/ieee754/dbl-64/sincostab.c): 110 x {sin hi, sin lo, cos hi, cos lo} at k/128
    import math
    off = find(blob, d2b(0.0) + d2b(0.0) + d2b(1.0) + d2b(0.0) + d2b(math.sin(1.0 / 128.0)), "sincostab")
    tab = doubles(blob, off, 440)
    assert abs(tab[4 * 64] - math.sin(0.5)) < 1e-15 and abs(tab[4 * 64 + 2] - math.cos(0.5)) < 1e-15
    parts.append("// __sincostab (sysdeps/ieee754/dbl-64/sincostab.c): {sn, ssn, cs, ccs} for x = k/128, k = 0..109\n"
                 "RM_TAB double rm_sincostab[440] = {\n" + fmt_d(tab, 4) + "\n};\n")

    # asncs (sysdeps/ieee754/dbl-64/asincos.tbl): 2568 doubles of per-interval centre points,
    # polynomial coefficients and asin values (rows of 11 / 12 / 13 / 14 / 15 doubles)
    def anchor_hex(*hx):
        return b"".join(d2b(float.fromhex(h)) for h in hx)
    off = find(blob, anchor_hex("0x1.0400000000000p-3", "0x1.0216988994424p+0", "0x1.0a6a2b799b115p-4"), "asncs", 2568 * 8)
    tab = doubles(blob, off, 2568)
    # Re-laid out as 216 uniform rows of 13 doubles so the device routine needs no per-band selects:
    #   [x0, a1 .. a6, a7 .. a10 (zero where the band's polynomial is shorter), c0, asin(x0)]
    # e_asin.c's bands: |x| < 0.25 and < 0.5: degree 5, rows of 11; then degree 6 / 7 / 8 / 9 with rows of
    # 12 / 13 / 14 / 15.  A zero leading coefficient leaves the Horner value unchanged (fma(t, +0, c) == c).
    rows = []
    def add_row(src, deg):
        a = tab[src:src + deg + 4]
        rows.extend(list(a[0:7]) + [a[7 + j] if deg >= 6 + j else 0.0 for j in range(4)] + [a[deg + 2], a[deg + 3]])
    for i in range(32):
        add_row(11 * i, 5)
    for i in range(64):
        add_row(352 + 11 * i, 5)
    for i in range(120):
        if i < 64: add_row(1056 + 12 * i, 6)
        elif i < 108: add_row(992 + 13 * i, 7)
        elif i < 116: add_row(884 + 14 * i, 8)
        else: add_row(768 + 15 * i, 9)
    assert len(rows) == 216 * 13
    parts.append("// asncs (sysdeps/ieee754/dbl-64/asincos.tbl), uniform rows: x0, a1..a10 (zero-padded), c0, asin(x0)\n"
                 "RM_TAB double rm_asncs[2808] = {\n" + fmt_d(rows, 13) + "\n};\n")
    # inroot (sysdeps/ieee754/dbl-64/root.tbl): 1/sqrt seeds
    off = find(blob, anchor_hex("0x1.68a1f80d71820p+0", "0x1.65de82af9631fp+0", "0x1.632b1201d39e5p+0"), "inroot", 128 * 8)
    tab = doubles(blob, off, 128)
    parts.append("// inroot (sysdeps/ieee754/dbl-64/root.tbl)\n"
                 "RM_TAB double rm_inroot[128] = {\n" + fmt_d(tab, 4) + "\n};\n")
    # cij (sysdeps/ieee754/dbl-64/uatan.tbl): 241 rows {x_i, atan(x_i), c1..c5}
    off = find(blob, anchor_hex("0x1.0400665e0244ep-4", "0x1.03a737b53dd20p-4", "0x1.fdf1fcf5cfb72p-1"), "cij", 241 * 7 * 8)
    tab = doubles(blob, off, 241 * 7)
    parts.append("// cij (sysdeps/ieee754/dbl-64/uatan.tbl): 241 x 7\n"
                 "RM_TAB double rm_cij[1687] = {\n" + fmt_d(tab, 7) + "\n};\n")

    with open(OUT, "w") as f:
        f.write("// GENERATED by tools/extract_libm_tables.py from the system libm.so.6\n"
                "// (Ubuntu GLIBC 2.35-0ubuntu3.11).  Constant tables of glibc's published fp64\n"
                "// algorithms; see rm_math_*.h for the restated routines.  Do not edit.\n"
                "#pragma once\n#include <stdint.h>\n\n"
                "#if defined(__HIPCC__)\n#define RM_TAB static __device__ __constant__ const\n"
                "#else\n#define RM_TAB static const\n#endif\n\n")
        body = "\n".join(parts)
        import re as _re
        names = _re.findall(r"RM_TAB (double|uint64_t) (rm_\w+)\[(\d+)\]", body)
        body = _re.sub(r"RM_TAB (double|uint64_t) rm_(\w+)\[", r"RM_TAB \1 rm_g_\2[", body)
        f.write(body)
        # table access (constant memory or the padded LDS mirrors of the kernels): csrc/rm_tables.h
        f.write("\n")
    print("wrote", OUT)


if __name__ == "__main__":
    main()
