"""In-tree build of libhpmn_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
OUT = os.path.join(_HERE, "lib", "libhpmn_hip.so")


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (expected under /opt/rocm/bin)")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def is_stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source into one shared object.  Returns the path."""
    if not force and not is_stale():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I" + INCLUDE, "-I" + CSRC, "-o", OUT + ".tmp"] + os.environ.get("HPMN_HIPCC_FLAGS", "").split() + sources()
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed:\n%s\n%s" % (proc.stdout, proc.stderr))
    os.replace(OUT + ".tmp", OUT)
    return OUT


HOST_OUT = os.path.join(_HERE, "lib", "libhpmn_host.so")
HOST_SRC = os.path.join(CSRC, "host", "crc32c.c")


def build_host_library(force: bool = False) -> str:
    """Host-only helpers (crc32c for the TF checkpoint files): plain C, gcc, no HIP."""
    if not force and os.path.exists(HOST_OUT) and os.path.getmtime(HOST_OUT) >= os.path.getmtime(HOST_SRC):
        return HOST_OUT
    os.makedirs(os.path.dirname(HOST_OUT), exist_ok=True)
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        raise RuntimeError("no C compiler for the host helper library")
    proc = subprocess.run([cc, "-O2", "-shared", "-fPIC", "-o", HOST_OUT + ".tmp", HOST_SRC], capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("host helper build failed:\n%s" % proc.stderr)
    os.replace(HOST_OUT + ".tmp", HOST_OUT)
    return HOST_OUT


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
