"""Build experiment variants of both libraries: one or more csrc files recompiled with extra -D flags, everything else taken from
the regular object directories.  usage: python scripts/build_variant.py NAME FILE.hip -DFLAG[=V] ... [--- FILE2.hip -DFLAG ...]
-> contrastors_amd/lib/variants/libcontrastors_hip_NAME.so and libcontrastors_hip_dev_NAME.so   (scripts/, experiments)"""
import subprocess
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from contrastors_amd import build as B  # noqa: E402

name, rest = sys.argv[1], sys.argv[2:]
specs, cur = [], []
for tok in rest:
    if tok == "---":
        specs.append(cur)
        cur = []
    else:
        cur.append(tok)
specs.append(cur)
B.build()
vdir = B.LIBDIR / "variants"
vdir.mkdir(exist_ok=True)
for product in (True, False):
    kind = "product" if product else "dev"
    objs, stems = [], set()
    for spec in specs:
        fname, flags = spec[0], spec[1:]
        obj = vdir / f"{Path(fname).stem}_{name}_{kind}.o"
        cmd = [B.HIPCC, *B.FLAGS, *(["-DCX_PRODUCT"] if product else []), *flags, "-c", str(B.CSRC / fname), "-o", str(obj)]
        subprocess.check_call(cmd)
        objs.append(obj)
        stems.add(Path(fname).stem)
    others = [o for o in sorted((B.OBJDIR / kind).glob("*.o")) if o.stem not in stems]
    lib = vdir / f"libcontrastors_hip{'' if product else '_dev'}_{name}.so"
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(lib), *map(str, objs), *map(str, others)])
    print("built", lib)
