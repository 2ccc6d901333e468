"""Build csrc/*.hip into csrc/libgicap.so with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import concurrent.futures
import glob
import os
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# GIC_LIB_VARIANT=stamps: the tools build (-DGIC_STAMPS: per-phase s_memtime stamps, tools/*_stamps.py) as csrc/libgicap_stamps.so with
# its own object directory, beside the product library; GIC_LIB_VARIANT=nt: the trunk's activation loads with the nt cache policy
# (common.h GIC_TRUNK_NT; a measurement build).  Tests, smoke() and the driver's bench.py never set it
VARIANT = os.environ.get("GIC_LIB_VARIANT", "")
LIB = os.path.join(CSRC, f"libgicap_{VARIANT}.so" if VARIANT else "libgicap.so")
OBJDIR = os.path.join(CSRC, f"build_{VARIANT}" if VARIANT else "build")
ARCH = "gfx950"
FLAGS = (["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"] + (["-DGIC_STAMPS"] if VARIANT == "stamps" else []) + (["-DGIC_TRUNK_NT=2"] if VARIANT == "nt" else [])
         + os.environ.get("GIC_EXTRA_FLAGS", "").split())


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; cannot build libgicap.so")
    return exe


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile what is stale.  An exclusive file lock serialises concurrent callers (one rank per GPU under torchrun)."""
    import fcntl
    os.makedirs(OBJDIR, exist_ok=True)
    with open(os.path.join(OBJDIR, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force: bool, verbose: bool) -> str:
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [
        os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "gicap.h")]
    objdir = OBJDIR
    hipcc = _hipcc()

    def compile_one(src: str) -> str:
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc, *FLAGS, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
