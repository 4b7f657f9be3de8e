"""In-tree build of libhpmn_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import concurrent.futures
import glob
import hashlib
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


def _headers():
    return sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h")))


BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def _extra_flags():
    return os.environ.get("HPMN_HIPCC_FLAGS", "").split()


def _sha(paths, extra=()):
    h = hashlib.sha1()
    for x in extra:
        h.update(x.encode())
        h.update(b"\0")
    for p in paths:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def source_sha(flags=None) -> str:
    """sha1 over every kernel source, header and compiler flag: what the shared object was built from."""
    return _sha(sources() + _headers(), BASE_FLAGS + (list(flags) if flags is not None else _extra_flags()))


def is_stale(out: str = OUT, flags=None) -> bool:
    """By CONTENT, not by mtime: a shipped .so newer than the sources of a fresh checkout (or an older one restored
    beside edited sources) is caught either way.  The stamp is written next to the library by build_library()."""
    stamp = out + ".sha1"
    if not (os.path.exists(out) and os.path.exists(stamp)):
        return True
    with open(stamp) as f:
        return f.read().strip() != source_sha(flags)


def build_library(force: bool = False, verbose: bool = False, out: str = OUT, flags=None) -> str:
    """Compile every HIP source (one object each, cached by content hash, compiled in parallel) and link them into one
    shared object.  Returns its path.  ``flags`` (default: $HPMN_HIPCC_FLAGS) are extra compiler flags -- developer
    variants (tools/) pass -D switches and their own ``out``."""
    extra = list(flags) if flags is not None else _extra_flags()
    if not force and not is_stale(out, extra):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    # one builder at a time: the ranks of a data-parallel launch import the package together, and a stale library would have
    # every one of them compile into the same object files (the others find it fresh once the first is done)
    import fcntl
    with open(out + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not is_stale(out, extra):
                return out
            return _build_locked(force, verbose, out, extra)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force: bool, verbose: bool, out: str, extra) -> str:
    objdir = os.path.join(_HERE, "lib", "obj")
    os.makedirs(objdir, exist_ok=True)
    hdrs = _headers()
    cc = _hipcc()
    jobs = []
    for src in sources():
        key = _sha([src] + hdrs, BASE_FLAGS + extra)[:16]
        obj = os.path.join(objdir, "%s-%s.o" % (os.path.splitext(os.path.basename(src))[0], key))
        jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        if os.path.exists(obj) and not force:
            return None
        cmd = [cc] + BASE_FLAGS + ["-I" + INCLUDE, "-I" + CSRC, "-c", "-o", obj + ".tmp"] + extra + [src]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            return "hipcc failed on %s:\n%s\n%s" % (os.path.basename(src), proc.stdout, proc.stderr)
        os.replace(obj + ".tmp", obj)
        return None

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        errors = [e for e in pool.map(compile_one, jobs) if e]
    if errors:
        raise RuntimeError("\n".join(errors))
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out + ".tmp"] + [obj for _, obj in jobs]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (proc.stdout, proc.stderr))
    os.replace(out + ".tmp", out)
    with open(out + ".sha1", "w") as f:
        f.write(source_sha(extra) + "\n")
    # drop the objects of older source versions (and of developer variants: they recompile when asked for again)
    live = {obj for _, obj in jobs}
    if out == OUT:
        for old in glob.glob(os.path.join(objdir, "*.o")):
            if old not in live:
                os.remove(old)
    return out


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
