"""Like-for-like: the shipped bf16 GEMM (gemm_bf16_v6, through the C-ABI) against the vendor BLAS behind torch.matmul on the seven dense
shapes of a nomic-bert-2048 block at T = 262144 token rows.  Measurement only: nothing under contrastors_amd/ calls
the vendor library.

  python scripts/v6_vs_vendor.py time [--seconds 1.0] [--rounds 2]
        per shape: a warm-up window, then alternating ~`seconds`-long windows of back-to-back launches of each kernel (vendor, v6, vendor,
        v6, ...), every launch event-timed in batches of 8, socket power / shader clock sampled over each window (librocm_smi64).
  python scripts/v6_vs_vendor.py workload [--reps 3]
        the launch sequence for a `rocprofv3 --pmc ... --kernel-trace` pass: per shape `reps` vendor launches, then `reps` v6 launches.
  python scripts/v6_vs_vendor.py parse <counter_collection.csv> [<more.csv> ...] [--reps 3]
        per shape and kernel: the mean of every counter over the group's launches + per-MFMA-FLOP ratios, vendor | v6 | v6 / vendor.
"""
from __future__ import annotations

import argparse
import csv
import sys
import time
from collections import defaultdict
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

T, d, I = 262144, 768, 3072
SHAPES = (("qkv_fwd", 3 * d, d), ("out_fwd", d, d), ("fc1_fwd", 2 * I, d), ("fc2_fwd", d, I), ("fc1_dgrad", d, 2 * I),
          ("fc2_dgrad", I, d), ("qkv_dgrad", d, 3 * d))


def _setup(N, K, dev="cuda"):
    import torch
    g = torch.Generator(device=dev).manual_seed(1234 + N + K)
    x = torch.randn(T, K, device=dev, generator=g).bfloat16()
    w = (torch.randn(N, K, device=dev, generator=g) * 0.05).bfloat16()
    y = torch.empty(T, N, device=dev, dtype=torch.bfloat16)
    return x, w, y


def _launchers(x, w, y, N, K):
    import torch

    from contrastors_amd import _C
    lib = _C.lib()
    s = torch.cuda.current_stream().cuda_stream
    wt = w.t()

    def vendor():
        torch.matmul(x, wt, out=y)

    def v6():
        _C.check(lib.cx_gemm_bf16_nt(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, T, N, K, K, K, N, 0, 1, 1.0, s), "cx_gemm_bf16_nt")

    return {"vendor": vendor, "v6": v6}


def cmd_time(a):
    import torch

    from scripts.box_calibration import SmiSampler
    print(f"# T = {T} rows, bf16 x bf16 -> bf16; windows of {a.seconds:.1f} s of back-to-back launches, alternating kernels, {a.rounds} rounds; "
          f"median us per launch over the window's event-timed batches of 8; power / clock: librocm_smi64 at 20 Hz over the window")
    print(f"{'shape':10s} {'N':>5s} {'K':>5s} | {'vendor us':>9s} {'TF':>7s} {'MHz':>5s} {'W':>5s} | {'v6 us':>9s} {'TF':>7s} {'MHz':>5s} {'W':>5s} | v6/vendor (TF)")
    geo = 1.0
    for name, N, K in SHAPES:
        x, w, y = _setup(N, K)
        fns = _launchers(x, w, y, N, K)
        fl = 2.0 * T * N * K

        def window(fn, seconds, sample):
            smp = SmiSampler(torch.cuda.current_device(), hz=20.0).start() if sample else None
            ts = []
            t_end = time.perf_counter() + seconds
            while time.perf_counter() < t_end:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(8):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / 8)
            smi = smp.stop() if smp else {}
            ts.sort()
            return ts[len(ts) // 2], smi.get("mean_sclk_mhz"), smi.get("mean_power_w")

        for fn in fns.values():   # both code paths warm, the package at its sustained clock
            window(fn, 0.4, False)
        acc = {k: [] for k in fns}
        for _ in range(a.rounds):
            for k, fn in fns.items():
                acc[k].append(window(fn, a.seconds, True))
        row = {}
        for k, v in acc.items():
            us = sum(r[0] for r in v) / len(v)
            mhz = [r[1] for r in v if r[1]]
            pw = [r[2] for r in v if r[2]]
            row[k] = (us, fl / us / 1e6, sum(mhz) / len(mhz) if mhz else float("nan"), sum(pw) / len(pw) if pw else float("nan"))
        r = row["v6"][1] / row["vendor"][1]
        geo *= r
        print(f"{name:10s} {N:5d} {K:5d} | {row['vendor'][0]:9.1f} {row['vendor'][1]:7.1f} {row['vendor'][2]:5.0f} {row['vendor'][3]:5.0f} | "
              f"{row['v6'][0]:9.1f} {row['v6'][1]:7.1f} {row['v6'][2]:5.0f} {row['v6'][3]:5.0f} | {r:6.3f}")
        del x, w, y
        torch.cuda.empty_cache()
    print(f"# geometric mean of v6 / vendor over the seven shapes: {geo ** (1 / len(SHAPES)):.3f}")


def cmd_workload(a):
    import torch
    for name, N, K in SHAPES:
        x, w, y = _setup(N, K)
        fns = _launchers(x, w, y, N, K)
        torch.cuda.synchronize()
        for k in ("vendor", "v6"):
            for _ in range(a.reps):
                fns[k]()
            torch.cuda.synchronize()
        del x, w, y
        torch.cuda.empty_cache()


def _is_gemm(name: str) -> str | None:
    if "gemm_bf16_v6" in name or "gemm_bf16_v7" in name:
        return "v6"
    if name.startswith("Cijk_") or "Cijk_" in name or "hipblaslt" in name.lower() or "rocblas" in name.lower():
        return "vendor"
    return None


def cmd_parse(a):
    per = defaultdict(dict)   # dispatch id -> {counter: value, "_k": kernel name}
    import gzip
    for path in a.csv:
        for r in csv.DictReader(gzip.open(path, "rt") if path.endswith(".gz") else open(path)):
            did = (path, int(r.get("Dispatch_Id") or r.get("Correlation_Id") or 0))
            per[did]["_k"] = r["Kernel_Name"]
            if r.get("End_Timestamp") and r.get("Start_Timestamp"):
                per[did]["duration_us (this pass)"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            per[did][r["Counter_Name"]] = per[did].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    # the workload's order inside each file: per shape `reps` vendor launches, then `reps` v6 launches
    table = defaultdict(lambda: defaultdict(lambda: defaultdict(list)))   # shape -> kernel -> counter -> values
    names = defaultdict(dict)
    for path in a.csv:
        seq = [(did[1], v) for did, v in per.items() if did[0] == path and _is_gemm(v["_k"])]
        seq.sort(key=lambda t: t[0])
        expect = len(SHAPES) * 2 * a.reps
        if len(seq) != expect:
            others = sorted({v["_k"][:60] for did, v in per.items() if did[0] == path and not _is_gemm(v["_k"])})
            print(f"# WARNING {path}: {len(seq)} GEMM dispatches, expected {expect}; other kernels: {others[:8]}")
        i = 0
        for name, N, K in SHAPES:
            for kern in ("vendor", "v6"):
                grp = seq[i:i + a.reps]
                i += a.reps
                for _, v in grp:
                    if _is_gemm(v["_k"]) != kern:
                        print(f"# WARNING {name}/{kern}: dispatch is {v['_k'][:50]}")
                    names[name][kern] = v["_k"][:110]
                    for c, val in v.items():
                        if c != "_k":
                            table[name][kern][c].append(val)
                    # effective shader clock of THIS dispatch: GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_BUSY_CYCLES over 32 SEs
                    if "GRBM_GUI_ACTIVE" in v and v.get("duration_us (this pass)"):
                        table[name][kern]["effective_clock_MHz (GRBM/8/duration)"].append(v["GRBM_GUI_ACTIVE"] / 8 / v["duration_us (this pass)"])
                    if "SQ_BUSY_CYCLES" in v and v.get("duration_us (this pass)"):
                        table[name][kern]["effective_clock_MHz (SQ_BUSY/32/duration)"].append(v["SQ_BUSY_CYCLES"] / 32 / v["duration_us (this pass)"])
    for name, N, K in SHAPES:
        if name not in table:
            continue
        fl = 2.0 * T * N * K
        print(f"== {name}  N={N} K={K}   ({fl / 1e12:.3f} TFLOP = {fl / 32768 / 1e6:.2f} M 32x32x16-MFMA equivalents)")
        print(f"   vendor kernel: {names[name].get('vendor', '?')}")
        print(f"   v6 kernel:     {names[name].get('v6', '?')}")
        ctrs = sorted(set(table[name]["vendor"]) | set(table[name]["v6"]))
        print(f"   {'counter':30s} {'vendor':>16s} {'v6':>16s} {'v6/vendor':>10s}   per MFMA-eq: vendor / v6")
        for c in ctrs:
            mv = lambda k: (sum(table[name][k][c]) / len(table[name][k][c])) if table[name][k].get(c) else float("nan")   # noqa: E731
            v, x = mv("vendor"), mv("v6")
            per_v, per_x = v / (fl / 32768), x / (fl / 32768)
            print(f"   {c:30s} {v:16.0f} {x:16.0f} {x / v if v else float('nan'):10.3f}   {per_v:9.3f} / {per_x:9.3f}")
        print()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    t = sub.add_parser("time"); t.add_argument("--seconds", type=float, default=1.0); t.add_argument("--rounds", type=int, default=2)
    wl = sub.add_parser("workload"); wl.add_argument("--reps", type=int, default=3)
    pr = sub.add_parser("parse"); pr.add_argument("csv", nargs="+"); pr.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    {"time": cmd_time, "workload": cmd_workload, "parse": cmd_parse}[a.cmd](a)
