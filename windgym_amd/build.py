"""Build libwindgym_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libwindgym_hip.so")
SOURCES = ["wg_flow.hip", "wg_env.hip", "wg_envb.hip", "wg_kernels.hip", "wg_api.hip", "wg_mann.hip", "wg_steady.hip"]


def _headers():
    """Every header / include file of csrc/ (a glob: a header left out of a hand-kept list made needs_build() call a stale
    binary fresh) + the public ABI header."""
    hs = sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")))
    return hs + [os.path.join(HERE, "..", "include", "windgym_hip.h")]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in [os.path.join(CSRC, f) for f in SOURCES] + _headers())


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the translation units in parallel (wg_flow.hip and wg_kernels.hip take about a minute each: they
    hold all the kernel instantiations) and link them; objects go to a temporary directory, only the .so stays in-tree."""
    if not force and not needs_build():
        return LIB
    import tempfile
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("WG_HIPCC_FLAGS", "").split()
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + extra + ["-Wno-unused-result", "-Wno-unused-value"]
    if verbose:
        flags.insert(0, "-Rpass-analysis=kernel-resource-usage")
    with tempfile.TemporaryDirectory(prefix="wg_build_") as tmp:
        objs, procs = [], []
        for src in SOURCES:
            obj = os.path.join(tmp, src.replace(".hip", ".o"))
            cmd = [hipcc] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd)))
            objs.append(obj)
        for cmd, pr in procs:
            if pr.wait() != 0:
                for _, other in procs:
                    if other.poll() is None:
                        other.kill()
                raise subprocess.CalledProcessError(pr.returncode, cmd)
        # (hipFFT: the one library call on the box-generation path, wg_mann.hip)
        subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lhipfft"], check=True)
    return LIB


if __name__ == "__main__":
    build(force=True, verbose="-v" in sys.argv)
    print(LIB)
