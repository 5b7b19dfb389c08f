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
] + os.environ.get("CX_EXTRA_HIPCC_FLAGS", "").split()   # experiments only (e.g. -DCX_V6_NT=1); the default build has none


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


DEV_ONLY = {"gemm_bf16.hip", "probe.hip"}   # earlier GEMM generations (v1 / v2 + the dev dispatch), probes
PRODUCT_ONLY = {"gemm_api.hip"}                                                      # the product's GEMM entry points
DEV_LIB = LIBDIR / "libcontrastors_hip_dev.so"


def _compile(src: Path, product: bool) -> Path:
    objdir = OBJDIR / ("product" if product else "dev")
    obj = objdir / (src.stem + ".o")
    headers = list(CSRC.glob("*.h")) + list(CSRC.glob("*.inc")) + list(INCLUDE.glob("*.h"))
    if _stale(obj, [src, *headers, Path(__file__)]):
        cmd = [HIPCC, *FLAGS, *(["-DCX_PRODUCT"] if product else []), "-c", str(src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def _audit_v6(product: bool) -> None:
    """The GEMM accumulators are physical AGPRs the compiler cannot see: refuse a build in which it generated its own AGPR
    traffic or scratch spills in those kernels (scripts/audit_agpr.py; found the hard way in round 2)."""
    cmd = [sys.executable, str(PKG.parent / "scripts" / "audit_agpr.py"), *(["-DCX_PRODUCT"] if product else [])]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"gemm_bf16_v6.hip register audit failed ({'product' if product else 'dev'} build):\n{r.stdout}\n{r.stderr}")


def _link(lib: Path, objs) -> None:
    if _stale(lib, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(lib), *map(str, objs)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")


def build(verbose: bool = False, dev: bool = True) -> Path:
    """libcontrastors_hip.so = the product (-DCX_PRODUCT: one kernel per op, no debug switches, none of the superseded
    GEMM generations); libcontrastors_hip_dev.so = the same sources without the define + csrc files in DEV_ONLY."""
    (OBJDIR / "product").mkdir(parents=True, exist_ok=True)
    (OBJDIR / "dev").mkdir(parents=True, exist_ok=True)
    srcs = sorted(CSRC.glob("*.hip"))
    if not srcs:
        raise RuntimeError("no HIP sources found")
    jobs = [(s, True) for s in srcs if s.name not in DEV_ONLY]
    if dev:
        jobs += [(s, False) for s in srcs if s.name not in PRODUCT_ONLY]
    v6 = CSRC / "gemm_bf16_v6.hip"
    v7 = CSRC / "gemm_bf16_v7.hip"
    audit_needed = [prod for prod in ((True, False) if dev else (True,))
                    if _stale(OBJDIR / ("product" if prod else "dev") / "gemm_bf16_v6.o", [v6, v7, CSRC / "gemm_v6_acc.inc", Path(__file__)])]
    with cf.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
        audits = [ex.submit(_audit_v6, prod) for prod in audit_needed]
        objs = list(ex.map(lambda j: _compile(*j), jobs))
        for a in audits:
            a.result()
    _link(LIB, [o for o, (_, prod) in zip(objs, jobs) if prod])
    if dev:
        _link(DEV_LIB, [o for o, (_, prod) in zip(objs, jobs) if not prod])
    if verbose:
        print(f"built {LIB} ({LIB.stat().st_size/1e6:.1f} MB)" + (f" and {DEV_LIB} ({DEV_LIB.stat().st_size/1e6:.1f} MB)" if dev else ""))
    return LIB


if __name__ == "__main__":
    build(verbose=True)
