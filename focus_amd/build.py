"""Builds focus_amd/lib/libfocus_amd.so (hipcc, gfx950 only) in-tree.  `python -m focus_amd.build`."""
import concurrent.futures
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "lib", "obj")
LIB = os.path.join(HERE, "lib", "libfocus_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = os.environ.get("FOCUS_EXTRA_HIPFLAGS", "").split() + ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-pass-failed", "-Wall",
         "-Wno-unused-function"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, headers):
    obj = os.path.join(OBJ, os.path.basename(src).replace(".hip", ".o"))
    if _stale(obj, [src] + headers):
        subprocess.check_call([HIPCC] + FLAGS + ["-c", src, "-o", obj])
    return obj


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    headers = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "focus_amd.h")]
    if force:
        for f in glob.glob(os.path.join(OBJ, "*.o")):
            os.remove(f)
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, headers), srcs))
    if _stale(LIB, objs):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
