"""Build libcontrastors_hip.so (gfx950) in-tree with hipcc.

The library has no torch / python dependency: it is a plain C-ABI shared object (include/contrastors_hip.h).
`python -m contrastors_amd.build` compiles every csrc/*.hip to an object (in parallel, only when stale) and links
contrastors_amd/lib/libcontrastors_hip.so.  hipcc cross-compiles for gfx950 without a GPU present.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIBDIR = PKG / "lib"
OBJDIR = PKG / "lib" / "obj"
LIB = LIBDIR / "libcontrastors_hip.so"
INCLUDE = PKG.parent / "include"

HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall",
    "-Wno-unused-function", "-Wno-unused-variable", f"-I{INCLUDE}",
]


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def _compile(src: Path) -> Path:
    obj = OBJDIR / (src.stem + ".o")
    headers = list(CSRC.glob("*.h")) + list(INCLUDE.glob("*.h"))
    if _stale(obj, [src, *headers, Path(__file__)]):
        cmd = [HIPCC, *FLAGS, "-c", str(src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(verbose: bool = False) -> Path:
    OBJDIR.mkdir(parents=True, exist_ok=True)
    srcs = sorted(CSRC.glob("*.hip"))
    if not srcs:
        raise RuntimeError("no HIP sources found")
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    if _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB} ({LIB.stat().st_size/1e6:.1f} MB) from {len(srcs)} sources")
    return LIB


if __name__ == "__main__":
    build(verbose=True)
