"""LayerNorm forward / backward at the metric's row count (T = 262144, d = 768): time and HBM rate.
usage: python scripts/ln_microbench.py [--lib dev]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from contrastors_amd import _C  # noqa: E402

lib = _C.lib()
s = torch.cuda.current_stream().cuda_stream
T, d = 262144, 768
x = torch.randn(T, d, device="cuda").bfloat16()
r = torch.randn(T, d, device="cuda").bfloat16()
g = torch.ones(d, device="cuda")
b = torch.zeros(d, device="cuda")
out, z = torch.empty_like(x), torch.empty_like(x)
mean, rstd = torch.empty(T, device="cuda"), torch.empty(T, device="cuda")
dout, dz = torch.randn_like(x), torch.empty_like(x)
dg, db = torch.zeros(d, device="cuda"), torch.zeros(d, device="cuda")
ws = torch.empty(768 * 2 * d, device="cuda")


def t(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


f1 = t(lambda: lib.cx_layernorm_fwd(x.data_ptr(), None, g.data_ptr(), b.data_ptr(), out.data_ptr(), None, mean.data_ptr(),
                                    rstd.data_ptr(), T, d, 1e-12, s))
f2 = t(lambda: lib.cx_layernorm_fwd(x.data_ptr(), r.data_ptr(), g.data_ptr(), b.data_ptr(), out.data_ptr(), z.data_ptr(),
                                    mean.data_ptr(), rstd.data_ptr(), T, d, 1e-12, s))
bw = t(lambda: lib.cx_layernorm_bwd(dout.data_ptr(), None, x.data_ptr(), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(), None,
                                    dz.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(), T, d, s))
B = T * d * 2
print(f"ln fwd (x -> out):              {f1:7.1f} us  {2 * B / f1 / 1e6:5.2f} TB/s")
print(f"ln fwd (x + res -> out, z):     {f2:7.1f} us  {4 * B / f2 / 1e6:5.2f} TB/s")
print(f"ln bwd (dout, z -> dz, dg, db): {bw:7.1f} us  {3 * B / bw / 1e6:5.2f} TB/s")
