#!/usr/bin/env python3
"""Frame time of (scene, strategy) cells at 1920x1080 over the values of ONE RmFrameDesc schedule knob.
  python tools/knob_sweep.py <knob> "<v0,v1,...>" "<sid,kid;sid,kid;...>" [WxH] [--lib path] [--set knob=value ...]
e.g.  python tools/knob_sweep.py hold_after "0,-1,48,64,96,128" "0,0;1,0;3,0;12,0;13,0"
(timing only; every knob value yields the same frames -- tests/ and tests/fuzz_parity.py check that)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    args = sys.argv[1:]
    if "--lib" in args:
        os.environ["RM_HIP_LIB"] = os.path.abspath(args.pop(args.index("--lib") + 1))
        args.remove("--lib")
    from raymarch_algo_compare_amd import _native, registry
    from raymarch_algo_compare_amd.camera import Camera
    def val(v):                      # "16:48" is a pair (suspend_after)
        return tuple(int(x) for x in v.split(":")) if ":" in v else int(v)
    knob, values, cells = args[0], [val(v) for v in args[1].split(",")], [tuple(int(v) for v in c.split(",")) for c in args[2].split(";")]
    fixed = {}
    while "--set" in args:           # --set name=value: another knob held fixed
        k, v = args.pop(args.index("--set") + 1).split("=")
        fixed[k] = val(v)
        args.remove("--set")
    W, H = (int(v) for v in args[3].split("x")) if len(args) > 3 else (1920, 1080)
    _native.init()
    for sid, kid in cells:
        sc = registry.SCENES[sid]
        cam = Camera(sc.camera_position or (0.0, 0.0, 5.0), sc.camera_target or (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 60.0, W, H).params14()
        row = {"scene": sc.name, "strategy": kid, "knob": knob, "ms": {}}
        for rnd in range(2):
            for v in values:
                out = _native.render(_native.make_desc(sid, kid, cam, W, H, **{knob: v}, **fixed), warmup=2, repeats=7)
                ms = out["timing"]["ms_median"]
                row["ms"][str(v)] = round(min(ms, row["ms"].get(str(v), 1e9)), 4)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
