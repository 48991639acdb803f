#!/usr/bin/env python3
"""Store-path probe (rm_bench_store_path): bandwidth of the render kernel's flush code alone."""
import sys, os, ctypes, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raymarch_algo_compare_amd import _native
L = _native.init()
for W, H in ((1920, 1080), (7680, 4320), (15360, 8640)):
    d, i, h = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    _native.check(L.rm_alloc_frame(W, H, ctypes.byref(d), ctypes.byref(i), ctypes.byref(h)))
    t = _native.RmTiming(); t.warmup, t.repeats = 3, 20
    _native.check(L.rm_bench_store_path(W, H, d, i, h, ctypes.byref(t)))
    gbs = 9.0 * W * H / (t.ms_median * 1e-3) / 1e9
    print(json.dumps({"WxH": f"{W}x{H}", "ms_median": round(t.ms_median, 4), "GB_per_s": round(gbs, 1), "frac_of_8TBps": round(gbs / 8000, 3)}))
    _native.check(L.rm_free_frame(d, i, h))
