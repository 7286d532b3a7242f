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
    """Problems in kernels that issue loads by hand: the compiler does not know such a load is still in flight, so a
    register COPY of its destination (a spill to scratch, or a park in an AGPR: v_accvgpr_write_b32 aN, vM) made before
    the hand-counted wait copies garbage.  Returns [(kernel, what)] for every kernel with hand-issued loads that has
    scratch or any VGPR -> AGPR copy (strict: the copy may be of another value, but then the kernel is one register
    allocation away from the bad case)."""
    problems = []
    name, hand, copies, pending = None, False, 0, None
    in_asm = False
    with open(asm_path) as f:
        for line in f:
            t = line.strip()
            if name is None:
                if pending and t.startswith("; ScratchSize:"):
                    if int(t.split(":")[1]) != 0:
                        problems.append((pending, "scratch " + t.split(":")[1].strip() + " B/lane"))
                    pending = None
                m = re.match(r"^(_Z\w+):", line)
                if m:
                    name, hand, copies, pending = m.group(1), False, 0, None
                continue
            if t.startswith(";;#ASMSTART"):
                in_asm = True
            elif t.startswith(";;#ASMEND"):
                in_asm = False
            elif in_asm and t.startswith("global_load_"):
                hand = True
            elif re.match(r"v_accvgpr_write_b32 a\d+, v\d+", t):
                copies += 1
            elif t.startswith(".Lfunc_end"):
                if hand and copies:
                    problems.append((name, "%d VGPR->AGPR copies" % copies))
                pending = name if hand else None
                name = None
    return problems


def _compile(src, headers):
    obj = os.path.join(OBJ, os.path.basename(src).replace(".hip", ".o"))
    if _stale(obj, [src] + headers):
        extra = ["-save-temps=obj"] if _hand_loads(src) else []       # keeps the device .s next to the object: linted below
        subprocess.check_call([HIPCC] + FLAGS + extra + ["-c", src, "-o", obj])
    if _hand_loads(src) and os.path.exists(_device_asm(src)):
        bad = lint_hand_loads(_device_asm(src))
        if bad:
            os.remove(obj)
            raise RuntimeError("%s: kernels with hand-issued loads whose registers the compiler copies:\n  %s" % (
                os.path.basename(src), "\n  ".join("%s: %s" % b for b in bad)))
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
