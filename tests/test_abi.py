"""CPU-side checks of the drop-in boundary: the built library loads and exports every symbol that
include/windgym_hip.h declares; the ctypes mirror of wg_config matches the C layout."""
import ctypes as C
import os
import re
import subprocess

import pytest

from windgym_amd import binding, build
from windgym_amd.config import CConfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    return build.build()


def test_header_symbols_are_exported(lib_path):
    hdr = open(os.path.join(ROOT, "include", "windgym_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(wg_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(binding.ABI_SYMBOLS), declared ^ set(binding.ABI_SYMBOLS)
    L = C.CDLL(lib_path)
    for name in declared:
        assert hasattr(L, name), f"{name} not exported"
    L.wg_abi_version.restype = C.c_int
    assert L.wg_abi_version() == 4


def test_config_struct_layout_matches_c(tmp_path):
    """sizeof/offsetof of wg_config as gcc sees it == the ctypes mirror."""
    src = tmp_path / "l.c"
    fields = [f[0] for f in CConfig._fields_]
    body = "\n".join(f'printf("{f} %zu\\n", offsetof(wg_config, {f}));' for f in fields)
    src.write_text(f'#include <stdio.h>\n#include <stddef.h>\n#include "{ROOT}/include/windgym_hip.h"\n'
                   f'int main(){{ printf("sizeof %zu\\n", sizeof(wg_config));\n{body}\nreturn 0; }}')
    exe = tmp_path / "l"
    subprocess.run(["gcc", str(src), "-o", str(exe)], check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    assert int(out["sizeof"]) == C.sizeof(CConfig)
    for f in fields:
        assert int(out[f]) == getattr(CConfig, f).offset, f


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from windgym_amd.config import EnvConfig
    from windgym_amd.turbine import V80
    from helpers import load_golden
    _, meta = load_golden("env1")
    cfg = EnvConfig(turbine=V80(), yaml_dict=meta["cfg"], turbtype="None")
    with pytest.raises(binding.WindGymHipError):
        binding.HipBatch(cfg)
