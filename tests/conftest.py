import hashlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# Scenes exempt from bit-exactness.  Empty: pow / sin / cos / acos / atan2 / log are all exact
# restatements of glibc 2.35 (DESIGN.md "math parity"), so Mandelbulb and Gyroid are held to the
# same bar as the algebraic scenes.
TRANSCENDENTAL_SCENES = set()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class GoldenFrames:
    """One frames_*.npz written by oracle/gen_golden.py (outputs of the reference itself)."""

    def __init__(self, tag):
        self.tag = tag
        self.z = np.load(os.path.join(GOLDEN, f"frames_{tag}.npz"))
        self.pairs = sorted({tuple(int(p[1:]) for p in k.split("_")[:2]) for k in self.z.files})
        sp = os.path.join(GOLDEN, f"stats_{tag}.json")
        self.stats = json.load(open(sp, encoding="utf-8")) if os.path.exists(sp) else {}

    def get(self, sid, kid):
        p = f"s{sid}_k{kid}_"
        meta = self.z[p + "meta"]
        W, H, row0, rows = (int(meta[i]) for i in range(4))
        n = rows * W
        hit = np.unpackbits(self.z[p + "hitbits"])[:n].reshape(rows, W).astype(np.uint8)
        depth = np.zeros(n, dtype=np.float64)
        depth[hit.reshape(-1) > 0] = self.z[p + "t_hit"]
        return {
            "W": W, "H": H, "row0": row0, "rows": rows, "max_iterations": int(meta[4]),
            "hit_threshold": float(meta[5]), "max_distance": float(meta[6]), "lipschitz": float(meta[7]),
            "cam": self.z[p + "cam"].copy(), "iters": self.z[p + "iters"].astype(np.int32), "hit": hit,
            "depth": depth.reshape(rows, W), "sha_t": self.z[p + "sha_t"].tobytes(),
            "sha_fs": self.z[p + "sha_fs"].tobytes(),
        }


PARAM_ORDER = ["omega", "ar_omega_min", "ar_omega_max", "ar_smoothing", "ar_growth_rate", "ar_decay_rate", "beta",
               "overstep_min_step", "hybrid_stuck_step_ratio", "hybrid_min_step", "margin", "ar_omega_init",
               "overstep_bisection_steps", "hybrid_stuck_threshold", "segment_bisection_steps", "revaa_bisection_steps"]


def golden_param_cases(tag="48x36"):
    """frames_params_*.npz: frames the reference's strategy classes marched with NON-default constructor arguments
    (oracle/gen_golden.py --only params).  Yields (scene id, strategy id, params dict, frame record)."""
    z = np.load(os.path.join(GOLDEN, f"frames_params_{tag}.npz"))
    for n in range(int(z["ncases"][0])):
        p = f"c{n}_"
        meta = z[p + "meta"]
        W, H, row0, rows = (int(meta[i]) for i in range(4))
        hit = np.unpackbits(z[p + "hitbits"])[:rows * W].reshape(rows, W).astype(np.uint8)
        depth = np.zeros(rows * W, dtype=np.float64)
        depth[hit.reshape(-1) > 0] = z[p + "t_hit"]
        prm = {k: (int(v) if k.endswith(("_steps", "_threshold")) else float(v)) for k, v in zip(PARAM_ORDER, z[p + "prm"])}
        yield int(z[p + "ids"][0]), int(z[p + "ids"][1]), prm, {
            "W": W, "H": H, "row0": row0, "rows": rows, "max_iterations": int(meta[4]), "hit_threshold": float(meta[5]),
            "max_distance": float(meta[6]), "lipschitz": float(meta[7]), "cam": z[p + "cam"].copy(),
            "iters": z[p + "iters"].astype(np.int32), "hit": hit, "depth": depth.reshape(rows, W),
            "sha_t": z[p + "sha_t"].tobytes(), "sha_fs": z[p + "sha_fs"].tobytes()}


def sha_f64(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<f8").tobytes()).digest()


_cache = {}


def golden_frames(tag):
    if tag not in _cache:
        _cache[tag] = GoldenFrames(tag)
    return _cache[tag]


@pytest.fixture(scope="session")
def hip():
    """The ctypes binding, initialised on cuda:0 -- fails (does not skip) without the HIP library."""
    from raymarch_algo_compare_amd import _native
    _native.init()
    return _native


def build_native(name: str) -> str:
    """g++ build of a tests/native/*.cpp check harness (host compile of the product's kernel
    headers, tests only).  Returns the path of the shared object."""
    import subprocess
    src = os.path.join(ROOT, "tests", "native", f"{name}.cpp")
    out = os.path.join(ROOT, "tests", "native", f"_build_{name}.so")
    hdr_dir = os.path.join(ROOT, "raymarch_algo_compare_amd", "csrc")
    newest = max([os.path.getmtime(src)] + [os.path.getmtime(os.path.join(hdr_dir, f))
                                            for f in os.listdir(hdr_dir) if f.endswith(".h")])
    if not os.path.exists(out) or os.path.getmtime(out) < newest:
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-mfma", "-msse4.1",
                        "-fno-builtin", "-o", out, src], check=True)
    return out
