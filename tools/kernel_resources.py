#!/usr/bin/env python3
"""Registers, spills, scratch and LDS of the built gfx950 kernels, read from the code objects' metadata notes.

  python tools/kernel_resources.py [object or .so ...] [--filter REGEX]

Default input: raymarch_algo_compare_amd/_build/scene_*.o (host objects carrying the device code as an offload
bundle).  tests/test_code_objects.py asserts on the same numbers.
"""
import argparse
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def code_objects(path, workdir):
    """Extract the gfx950 code objects bundled in a host object / shared library; yields file paths."""
    base = os.path.join(workdir, os.path.basename(path))
    if os.path.lexists(base):
        os.remove(base)
    os.symlink(os.path.abspath(path), base)               # llvm-objdump writes next to its input
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", base], check=True, capture_output=True)
    for f in sorted(glob.glob(base + ".*gfx950*")):
        yield f


def kernels_of(code_object):
    out = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", code_object], check=True, capture_output=True, text=True).stdout
    res, cur = [], None
    for line in out.splitlines():
        m = re.match(r"\s+- \.agpr_count:\s+(\d+)", line)
        if m:
            cur = {"agpr_count": int(m.group(1))}
            res.append(cur)
            continue
        m = re.match(r"\s+\.(\w+):\s+(\S+)", line)
        if m and cur is not None and m.group(1) in ("name", "vgpr_count", "sgpr_count", "sgpr_spill_count", "vgpr_spill_count",
                                                   "private_segment_fixed_size", "group_segment_fixed_size"):
            v = m.group(2)
            cur[m.group(1)] = v if m.group(1) == "name" else int(v)
    return [k for k in res if "name" in k]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r"rm::", "", n) for n in out]


def collect(paths):
    rows = []
    with tempfile.TemporaryDirectory() as td:
        for p in paths:
            for co in code_objects(p, td):
                ks = kernels_of(co)
                for k, d in zip(ks, demangle([k["name"] for k in ks])):
                    k["demangled"] = d
                    k["object"] = os.path.basename(p)
                    rows.append(k)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("paths", nargs="*")
    ap.add_argument("--filter", default="")
    args = ap.parse_args()
    paths = args.paths or sorted(glob.glob(os.path.join(ROOT, "raymarch_algo_compare_amd", "_build", "scene_*.o")))
    flt = re.compile(args.filter) if args.filter else None
    for k in collect(paths):
        if flt and not flt.search(k["demangled"]):
            continue
        print("%-110s vgpr %3d sgpr %3d sgpr_spill %3d vgpr_spill %2d scratch %3d lds %6d" % (
            k["demangled"][:110], k.get("vgpr_count", -1), k.get("sgpr_count", -1), k.get("sgpr_spill_count", -1),
            k.get("vgpr_spill_count", -1), k.get("private_segment_fixed_size", -1), k.get("group_segment_fixed_size", -1)))


if __name__ == "__main__":
    sys.exit(main())
