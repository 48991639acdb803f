"""Register allocation of the built gfx950 kernels, read from the code objects' metadata (llvm-readelf --notes of the
device code bundled in raymarch_algo_compare_amd/_build/scene_*.o; tools/kernel_resources.py).  No GPU needed.

Two waves per SIMD need <= 256 unified registers per lane; the Mandelbulb kernels sit at that limit, and crossing it
once cost the 7680x4320 frame 33 % with every parity test green ("same-box A/B").  Spilled vector
registers mean scratch-memory traffic; a handful of loop-invariant values parked in scratch at kernel entry is what
the Mandelbulb pipeline kernel has today and what this test allows -- not more."""
import glob
import importlib.util
import os

import pytest

from conftest import ROOT

BUILD = os.path.join(ROOT, "raymarch_algo_compare_amd", "_build")


def _tool():
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def kernels():
    objs = [os.path.join(BUILD, f"scene_{sid}.o") for sid in (0, 2, 10)]
    if not all(os.path.exists(o) for o in objs):
        pytest.skip("scene objects are not in the tree (make -C raymarch_algo_compare_amd/csrc)")
    return _tool().collect(objs)


def _pick(kernels, text):
    ks = [k for k in kernels if text in k["demangled"]]
    assert ks, text
    return ks


def test_mandelbulb_kernels_fit_two_waves_per_simd(kernels):
    for name in ("render_kernel<SceneMandelbulb, StratStandard, 4, true, false>", "render_kernel<SceneMandelbulb, StratStandard, 4, true, true>",
                 "pipeline_kernel<SceneMandelbulb, StratStandard, 4, true, false>", "pipeline_kernel<SceneMandelbulb, StratStandard, 1, true, false>",
                 "resume_kernel<SceneMandelbulb, StratStandard, true, false>", "resume_team_kernel<SceneMandelbulb, StratStandard, false>"):
        for k in _pick(kernels, name):
            assert k["vgpr_count"] + k.get("agpr_count", 0) <= 256, (name, k)
            assert k["vgpr_spill_count"] <= 4 and k["private_segment_fixed_size"] <= 40, (name, k)      # entry-time parking only
    # every strategy's render kernel of the throughput regime stays within the limit
    for k in _pick(kernels, "render_kernel<SceneMandelbulb, "):
        assert k["vgpr_count"] + k.get("agpr_count", 0) <= 256, k["demangled"]


def test_cheap_scene_kernels_do_not_spill_vector_registers(kernels):
    for scene in ("SceneSphere", "SceneCube"):
        for k in _pick(kernels, f"render_kernel<{scene}, "):
            assert k["vgpr_spill_count"] == 0, k["demangled"]
            assert k["vgpr_count"] <= 168, k["demangled"]            # three waves per SIMD
            assert k["group_segment_fixed_size"] <= 40 * 1024, k["demangled"]
