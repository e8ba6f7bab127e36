"""Build openpcdet_amd/lib/libpcdet_mi355.so: hand-written HIP for gfx950 only, plain hipcc, in-tree.

    python -m openpcdet_amd.build            # incremental
    python -m openpcdet_amd.build --force

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so travels to the
GPU box with the tree (it is git-ignored, not gpurun-ignored).
"""
import os
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(LIBDIR, "libpcdet_mi355.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
          "-I", os.path.join(HERE, "..", "include")]
# per-TU extra flags
EXTRA = {
    # binary32 arithmetic in the reference's operation order: no FMA contraction on host or device
    "iou3d_nms.hip": ["-ffp-contract=off"],
    "vector_pool.hip": ["-ffp-contract=off"],
}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(HERE, "..", "include", "pcdet_mi355.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, force):
    obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
    spath = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(spath)
            and os.path.getmtime(obj) > _headers_mtime()):
        return obj
    cmd = [HIPCC] + COMMON + EXTRA.get(src, []) + os.environ.get("PVX_EXTRA_CFLAGS", "").split() + ["-c", spath, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if (not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) > os.path.getmtime(o) for o in objs)):
        if verbose:
            print("up to date:", LIB)
        return LIB
    # link in a scratch directory and move the result: hipcc leaves empty `<out>.N.host-*` temporaries beside its output
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, os.path.basename(LIB))
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        shutil.move(out, LIB)
    if verbose:
        print("built:", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
