"""Builds focus_amd/lib/libfocus_amd.so (hipcc, gfx950 only) in-tree.  `python -m focus_amd.build`."""
import concurrent.futures
import glob
import os
import re
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


def _hand_loads(src):
    """True if the file issues global loads from inline asm (hand-counted s_waitcnt, cdna_hip_programming.md 5.7)."""
    with open(src) as f:
        return 'asm volatile("global_load_dword' in f.read()


def _device_asm(src):
    return os.path.join(OBJ, os.path.basename(src).replace(".hip", "-hip-amdgcn-amd-amdhsa-gfx950.s"))


def lint_hand_loads(asm_path):
    """[(kernel, problem)] for the kernels of a device .s that issue loads by hand (focus_amd/asm_lint.py)."""
    from focus_amd import asm_lint
    return asm_lint.lint_hand_loads(asm_path)


def _compile(src, headers):
    obj = os.path.join(OBJ, os.path.basename(src).replace(".hip", ".o"))
    if _stale(obj, [src] + headers):
        extra = ["-save-temps=obj"] if _hand_loads(src) else []       # keeps the device .s next to the object: linted below
        subprocess.check_call([HIPCC] + FLAGS + extra + ["-c", src, "-o", obj])
    if _hand_loads(src):
        if not os.path.exists(_device_asm(src)):                        # an object without its assembly was not linted
            os.remove(obj)
            raise RuntimeError("%s issues loads by hand but its device assembly %s is missing: rebuild" % (
                os.path.basename(src), _device_asm(src)))
        bad = lint_hand_loads(_device_asm(src))
        if bad:
            os.remove(obj)
            raise RuntimeError("%s: registers of hand-issued loads touched before their wait (%d reports):\n  %s" % (
                os.path.basename(src), len(bad), "\n  ".join("%s: %s" % b for b in bad[:40])))
    return obj


def source_hash():
    """sha1 (12 hex digits) over the kernel sources and the public header: identifies WHICH kernels a build or a
    profile belongs to (bench.py only quotes PMC traffic recorded for the sources it is running)."""
    import hashlib
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))) + [
            os.path.join(HERE, "..", "include", "focus_amd.h")]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def _stamp():
    """focus_amd/lib/BUILD_HEAD: `<git head>[-dirty] src:<source_hash>` of the tree the library was built from (.git does
    not travel to the GPU box; this sidecar does)."""
    root = os.path.join(HERE, "..")
    head = "nogit"
    try:
        head = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                              check=True).stdout.strip()
        if subprocess.run(["git", "-C", root, "status", "--porcelain", "--", "focus_amd/csrc", "include"],
                          capture_output=True, text=True).stdout.strip():
            head += "-dirty"
    except Exception:
        pass
    with open(os.path.join(HERE, "lib", "BUILD_HEAD"), "w") as f:
        f.write("%s src:%s\n" % (head, source_hash()))


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
    if os.path.isdir(os.path.join(HERE, "..", ".git")):
        _stamp()
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
