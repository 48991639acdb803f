#!/usr/bin/env python3
"""One-off diagnostic behind DESIGN.md section 0 "b threading": what round 2's two-thread test did to the process.

librm_hip.so is loaded FIRST (it binds to /opt/rocm's libamdhip64.so.7), PyTorch afterwards (its wheel bundles a
runtime under the file name libamdhip64.so, which the loader does not match with the loaded SONAME): the process then
holds two HIP runtimes, and ctypes.CDLL("libamdhip64.so") -- what the aborted test used for its streams -- is the
OTHER one.  Single-threaded on purpose: the only variable is whose stream it is.

  python tools/diag_two_runtimes.py maps      # which copies are mapped, who owns hipStreamCreate (CPU-only is enough)
  python tools/diag_two_runtimes.py stream    # GPU: a stream of the other runtime is handed to rm_render_device
"""
import ctypes
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def mapped():
    found = set()
    for line in open("/proc/self/maps"):
        m = re.search(r"(/\S*(libamdhip64|libhsa-runtime64)\S*)", line)
        if m:
            found.add(m.group(1))
    return sorted(found)


def owner(fn):
    class DlInfo(ctypes.Structure):
        _fields_ = [("fname", ctypes.c_char_p), ("fbase", ctypes.c_void_p), ("sname", ctypes.c_char_p), ("saddr", ctypes.c_void_p)]
    libc = ctypes.CDLL(None)
    libc.dladdr.argtypes = [ctypes.c_void_p, ctypes.POINTER(DlInfo)]
    info = DlInfo()
    libc.dladdr(ctypes.cast(fn, ctypes.c_void_p).value, ctypes.byref(info))
    return info.fname.decode() if info.fname else "?"


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "maps"
    from raymarch_algo_compare_amd import _native
    L = _native.load()
    print("after librm_hip.so :", mapped(), flush=True)
    print("library bound to   :", _native.runtime_info(), flush=True)
    import torch
    print("torch", torch.__version__, "device_count", torch.cuda.device_count(), flush=True)
    print("after import torch :", mapped(), flush=True)
    rt = ctypes.CDLL("libamdhip64.so")
    print('CDLL("libamdhip64.so").hipStreamCreate lives in', owner(rt.hipStreamCreate), flush=True)
    print("librm_hip.so's hipStreamCreate lives in        ", owner(L.hipStreamCreate), flush=True)
    if what != "stream":
        return
    import numpy as np
    from raymarch_algo_compare_amd.camera import Camera
    _native.init(0)
    vp = ctypes.c_void_p
    foreign = vp()
    rt.hipStreamCreate.argtypes = [ctypes.POINTER(vp)]
    rc = rt.hipStreamCreate(ctypes.byref(foreign))           # initialises the second runtime (second HSA runtime too)
    print("other runtime: hipStreamCreate ->", rc, hex(foreign.value or 0), flush=True)
    cam = Camera((0.0, 0.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 60.0, 160, 120).params14()
    desc = _native.make_desc(10, 0, cam, 160, 120, pipeline=2, suspend_after=(6, 30))
    p = [vp(), vp(), vp()]
    _native.check(L.rm_alloc_frame(160, 120, *[ctypes.byref(q) for q in p]))
    rc = L.rm_render_device(ctypes.byref(desc), p[0], p[1], p[2], None, foreign)
    print("rm_render_device(stream of the other runtime) ->", rc, L.rm_last_error().decode(), flush=True)
    # what round 2's library did with it (no validation then): the first call of launch() on the caller's stream
    buf = vp()
    L.hipMalloc.argtypes = [ctypes.POINTER(vp), ctypes.c_size_t]
    L.hipMemsetAsync.argtypes = [vp, ctypes.c_int, ctypes.c_size_t, vp]
    L.hipStreamSynchronize.argtypes = [vp]
    print("hipMalloc ->", L.hipMalloc(ctypes.byref(buf), 4096), flush=True)
    if len(sys.argv) > 2 and sys.argv[2] == "unchecked":
        print("library runtime: hipMemsetAsync on the foreign stream ...", flush=True)
        rc = L.hipMemsetAsync(buf, 0, 4096, foreign)
        print("   ->", rc, flush=True)
        rc = L.hipStreamSynchronize(foreign)
        print("library runtime: hipStreamSynchronize(foreign) ->", rc, flush=True)
    iters = np.empty((120, 160), np.int32)
    _native.check(L.rm_render_device(ctypes.byref(desc), p[0], p[1], p[2], None, None))
    _native.check(L.rm_copy_frame_to_host(160, 120, p[0], p[1], p[2], None, iters.ctypes.data_as(vp), None))
    print("library stream still fine: mean iterations", float(iters.mean()), flush=True)


if __name__ == "__main__":
    main()
