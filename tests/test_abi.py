"""The C-ABI library loads without a GPU, exports every symbol include/rm_hip.h declares, its
structs have the layout the ctypes binding assumes, and it fails loudly (no CPU fallback)."""
import ctypes
import os
import re
import subprocess
import tempfile

import pytest

from conftest import ROOT
from raymarch_algo_compare_amd import _native


def _header():
    return open(os.path.join(ROOT, "include", "rm_hip.h"), encoding="utf-8").read()


def test_every_declared_symbol_is_exported():
    declared = set(re.findall(r"^(?:int|int32_t|void|size_t|const char\*)\s+(rm_\w+)\(", _header(), flags=re.M))
    assert declared and declared == set(_native.EXPORTS)
    lib = _native.load()
    for name in declared:
        assert hasattr(lib, name), name


def test_struct_layout_matches_header():
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "rm_hip.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(RmMarchConfig),' \
          'sizeof(RmFrameDesc), sizeof(RmStats), sizeof(RmTiming), sizeof(RmDeviceInfo), sizeof(RmStrategyParams),' \
          'offsetof(RmMarchConfig, params), offsetof(RmStrategyParams, overstep_bisection_steps)); return 0;}\n'
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "s")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        sizes = [int(v) for v in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    assert sizes == [ctypes.sizeof(_native.RmMarchConfig), ctypes.sizeof(_native.RmFrameDesc),
                     ctypes.sizeof(_native.RmStats), ctypes.sizeof(_native.RmTiming), ctypes.sizeof(_native.RmDeviceInfo),
                     ctypes.sizeof(_native.RmStrategyParams), _native.RmMarchConfig.params.offset,
                     _native.RmStrategyParams.overstep_bisection_steps.offset]


def test_default_strategy_params_are_the_reference_defaults():
    """rm_default_strategy_params (a host-only call) == the defaults of the reference's constructors, and the
    header documents every field of the binding in the same order."""
    lib = _native.load()
    p = _native.RmStrategyParams()
    lib.rm_default_strategy_params(ctypes.byref(p))
    assert {n: getattr(p, n) for n, _, _ in _native.STRATEGY_PARAM_FIELDS} == _native.DEFAULT_STRATEGY_PARAMS
    body = re.search(r"typedef struct RmStrategyParams \{(.*?)\} RmStrategyParams;", _header(), flags=re.S).group(1)
    fields = re.findall(r"^\s*(?:double|int32_t)\s+(\w+);", body, flags=re.M)
    assert fields == [n for n, _, _ in _native.STRATEGY_PARAM_FIELDS]
    c = _native.march_config(params={"omega": 1.6, "overstep_bisection_steps": 8})
    assert c.use_params == 1 and c.params.omega == 1.6 and c.params.overstep_bisection_steps == 8 and c.params.beta == 0.3
    assert _native.march_config().use_params == 0
    with pytest.raises(KeyError):
        _native.march_config(params={"gain": 2.0})


def test_registry_sizes():
    lib = _native.load()
    assert lib.rm_num_scenes() == 20 and lib.rm_num_strategies() == 11       # the registry; kernels exist for 13 (rm_hip.h)
    assert lib.rm_stats_device_bytes() == 8 * (24 + _native.RM_HIST_BINS) * 65   # canonical block + 64 partial blocks


def test_no_cpu_fallback_without_device():
    """Without rm_init on a gfx950 device every compute entry point reports RM_E_NO_DEVICE."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the no-device behaviour is checked on CPU-only hosts")
    lib = _native.load()
    assert lib.rm_init(0) == -4
    d = _native.make_desc(0, 0, [0.0] * 14, 8, 8)
    assert lib.rm_render(ctypes.byref(d), None, None, None, None, None, None, None, None) == -4
    assert b"rm_init" in lib.rm_last_error()
    with pytest.raises(_native.RmError):
        _native.init(0)
    from raymarch_algo_compare_amd import run_once
    with pytest.raises(_native.RmError):
        run_once()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "raymarch_algo_compare_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dp, f), encoding="utf-8", errors="replace").read()
                for needle in ("import oracle", "from oracle", "oracle/", "oracle.render", "oracle.lib", "librm_oracle", "rmo_", "_build_host_check", "_build_math_check"):
                    assert needle not in text, (os.path.join(dp, f), needle)


def test_tools_do_not_use_the_oracle_either():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load oracle/; the developer tools time and
    trace the product, they do not check it."""
    tools = os.path.join(ROOT, "tools")
    for dp, _, files in os.walk(tools):
        for f in files:
            if f.endswith((".py", ".sh", ".hip")):
                text = open(os.path.join(dp, f), encoding="utf-8", errors="replace").read()
                for needle in ("import oracle", "from oracle", "librm_oracle", "rmo_"):
                    assert needle not in text, (os.path.join(dp, f), needle)


def test_shard_plan_in_c_equals_the_python_plan():
    """rm_shard_rows / the row plan rm_gather_frame checks == sharding.plan_rows (a host-only call)."""
    from raymarch_algo_compare_amd import sharding
    lib = _native.load()
    for H in (4, 37, 48, 50, 1080, 2160, 4320, 4321):
        for N in (1, 2, 3, 4, 8):
            per = lib.rm_shard_rows(H, N)
            plans = [sharding.plan_rows(H, N, r) for r in range(N)]
            if plans[0].cyclic:
                assert H % (4 * N) == 0 and all(p.rows == H // N for p in plans)
            elif N > 1:
                assert [(p.row0, p.rows) for p in plans] == [(min(r * per, H), min((r + 1) * per, H) - min(r * per, H)) for r in range(N)]
            assert sum(p.rows for p in plans) == H


def test_rccl_entry_points_fail_loudly_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = _native.load()
    ident = ctypes.create_string_buffer(128)
    assert lib.rm_comm_init(ident, 1, 0) == -4            # RM_E_NO_DEVICE: rm_init has not succeeded
    d = _native.make_desc(0, 0, [0.0] * 14, 8, 8)
    assert lib.rm_gather_frame(ctypes.byref(d), None, None, None, None, None, None, None) == -4
    assert lib.rm_comm_destroy() == 0                      # nothing to destroy
