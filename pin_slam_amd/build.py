"""Build libpinhip.so (hipcc, gfx950) in-tree.  `python -m pin_slam_amd.build [--force]`.

Each csrc/*.hip is compiled to an object of its own (in parallel, only when it or a header changed)
and the objects are linked into pin_slam_amd/libpinhip.so."""
from __future__ import annotations

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libpinhip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"] + os.environ.get("PIN_EXTRA_CFLAGS", "").split()
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-ldl"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "pin_abi.h")]


def _obj(src: str) -> str:
    return os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")


def _newer(path: str, deps) -> bool:
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    return any(os.path.getmtime(d) > t for d in deps)


def stale() -> bool:
    return _newer(LIB, sources() + headers())


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not stale():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    hdr = headers()
    todo = [s for s in sources() if force or _newer(_obj(s), [s] + hdr)]

    def cc(src):
        cmd = [HIPCC, *CFLAGS, "-c", src, "-o", _obj(src)]
        if verbose:
            print("[pin_slam_amd.build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(cc, todo))
    cmd = [HIPCC, *LDFLAGS, "-o", LIB, *[_obj(s) for s in sources()]]
    if verbose:
        print("[pin_slam_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
