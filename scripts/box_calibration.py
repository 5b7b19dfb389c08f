"""Per-box calibration for bench.py.

The same code measured 4209 / 4143 / 4127 pairs/s on three boxes of the pool and 3966 on a fourth: the matrix-core clock a
package sustains under this load is set by its power budget, and that differs by box.  Nothing in the round-4 bench line let a
reader tell box from code.  This module gives the line three things:

  calibrate(dev)    BEFORE the headline leg: ~2 s of a register-only bf16 MFMA loop on random fragments (cx_calib_mfma_bf16:
                    one wave per SIMD on every CU -- the ceiling any GEMM main loop has on THIS package at its power limit) and a 2 GB
                    16-byte-per-lane copy (cx_calib_copy: the HBM stream rate), each timed with events after a warm-up half so that
                    the package is at its sustained clock, not its boost clock; socket power / shader clock of the probe sampled.
  SmiSampler        DURING the timed region: a background thread reading socket power and the shader clock from
                    librocm_smi64 through ctypes (no subprocess, ~10 Hz); every value is optional -- a box that does not
                    expose a sensor yields null, never an exception.
  frac_of_box_ceiling = roofline.achieved / box.mfma_probe_tflops, next to roofline.frac (which stays priced against the
                    2.5 PFLOP/s data-sheet peak); vs_box_blas = roofline.achieved / box.blas_ref_tflops (the vendor BLAS behind
                    torch.matmul on two of the step's GEMM shapes, measured on the same box seconds earlier: measurement only).

Usage on its own:  python scripts/box_calibration.py   -> one JSON object.
"""
from __future__ import annotations

import ctypes as C
import json
import sys
import threading
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def calibrate(dev=None, mfma_seconds: float = 2.0, copy_gb: float = 2.0) -> dict:
    import torch

    from contrastors_amd import _C

    lib = _C.lib()
    dev = dev or torch.device("cuda", torch.cuda.current_device())
    out = {}
    with torch.cuda.device(dev):
        s = torch.cuda.current_stream().cuda_stream
        n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
        g = torch.Generator(device="cpu").manual_seed(20240925)
        seed = torch.randn(1024 * 8, generator=g).to(dev).to(torch.bfloat16)
        sink = torch.zeros(4, device=dev)
        cyc = torch.zeros(n_cu * 4, dtype=torch.int64, device=dev)
        iters = 20000
        flop = n_cu * 4 * iters * 8 * 2.0 * 32 * 32 * 16

        def probe(fn, what, seconds, prefix):
            """`seconds` of back-to-back launches: the first half untimed (the package reaches its sustained clock), the second half with
            events around every launch; socket power / shader clock sampled over the timed half."""
            def launch():
                _C.check(fn(seed.data_ptr(), iters, n_cu, cyc.data_ptr(), sink.data_ptr(), s), what)

            launch()
            torch.cuda.synchronize()
            t_end = time.perf_counter() + seconds / 2
            while time.perf_counter() < t_end:
                for _ in range(16):
                    launch()
                torch.cuda.synchronize()
            smp = SmiSampler(dev.index or 0, hz=20.0).start()
            evs = []
            t_end = time.perf_counter() + seconds / 2
            while time.perf_counter() < t_end:
                batch = [torch.cuda.Event(enable_timing=True) for _ in range(17)]
                batch[0].record()
                for i in range(16):
                    launch()
                    batch[i + 1].record()
                torch.cuda.synchronize()
                evs += [batch[i].elapsed_time(batch[i + 1]) for i in range(16)]
            smi = smp.stop()
            evs.sort()
            med = evs[len(evs) // 2]
            mean_cyc = float(cyc.double().mean())
            out[prefix + "_tflops"] = flop / (med * 1e-3) / 1e12
            out[prefix + "_ms"] = med
            out[prefix + "_launches"] = len(evs)
            out[prefix + "_clock_mhz"] = mean_cyc / (med * 1e-3) / 1e6 if med > 0 else None   # s_memtime ticks per wall second of the last launch
            out[prefix + "_cycles_per_mfma"] = mean_cyc / (iters * 8)   # per 32 KFLOP: two v_mfma_f32_16x16x32_bf16 since round 6 (one 32x32x16 before)
            out[prefix + "_power_w"] = smi["mean_power_w"]
            out[prefix + "_sclk_mhz"] = smi["mean_sclk_mhz"]

        # register-only loop: the matrix pipes' own limit on this package (does not reach the power cap on the boxes measured)
        probe(lib.cx_calib_mfma_bf16, "cx_calib_mfma_bf16", mfma_seconds, "mfma_probe")
        # (round 5 also tried the same loop with its fragments re-read from LDS at the GEMM's rate as a second ceiling: that naive
        # loop issues an MFMA every 42 cycles where the shipped GEMM main loop manages 37 -- a "ceiling" the product beats per cycle
        # is none; dropped.  gpurun_out/r5g: 1508 vs 1921 TFLOP/s on that box.)
        # Vendor-BLAS reference (measurement only -- nothing under contrastors_amd/ calls it): torch.matmul (hipBLASLt) on two of the step's
        # own GEMM shapes, ~0.6 s each after a warm-up third.  Unlike the register-only probe it loads LDS and HBM like the shipped GEMMs
        # and runs into the same power cap: a frozen library on a fixed shape is the steadier per-box yardstick (round 5 saw a box whose
        # MFMA probe read 9 % low while its step -- and its in-step clock -- were 1-2 % low).
        # Round 6: the PRODUCT kernel (cx_gemm_bf16_nt -> gemm_bf16_v6) on the same two shapes, in the same windows:
        # per shape a warm-up, then alternating ~0.3 s windows vendor / v6 / vendor / v6 -> `v6_same_shapes_tflops` and
        # `like_for_like` = v6 / vendor, a ratio of two plain GEMMs on one shape, one box and one power state.
        try:
            tf, tf6 = [], []
            for (M, N, K) in ((131072, 3072, 768), (131072, 768, 3072)):
                xa = torch.randn(M, K, generator=g).to(dev).to(torch.bfloat16)
                wb = (torch.randn(N, K, generator=g) * 0.05).to(dev).to(torch.bfloat16)
                yo = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
                wbt = wb.t()
                f_vendor = lambda: torch.matmul(xa, wbt, out=yo)   # noqa: E731
                f_v6 = lambda: _C.check(lib.cx_gemm_bf16_nt(xa.data_ptr(), wb.data_ptr(), yo.data_ptr(), None, M, N, K, K, K, N, 0, 1, 1.0, s),   # noqa: E731
                                        "cx_gemm_bf16_nt")

                def window(fn, seconds):
                    ts = []
                    t_end = time.perf_counter() + seconds
                    while time.perf_counter() < t_end:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(8):
                            fn()
                        e1.record()
                        torch.cuda.synchronize()
                        ts.append(e0.elapsed_time(e1) / 8)
                    ts.sort()
                    return ts[len(ts) // 2]

                for fn in (f_vendor, f_v6):
                    for _ in range(3):
                        fn()
                    torch.cuda.synchronize()
                window(f_vendor, 0.25)
                tv, t6 = [], []
                for _ in range(2):
                    tv.append(window(f_vendor, 0.3))
                    t6.append(window(f_v6, 0.3))
                tf.append(2.0 * M * N * K / (sum(tv) / len(tv) * 1e-3) / 1e12)
                tf6.append(2.0 * M * N * K / (sum(t6) / len(t6) * 1e-3) / 1e12)
                del xa, wb, yo, wbt
            out["blas_ref_tflops"] = sum(tf) / len(tf)
            out["blas_ref_tflops_short_k"], out["blas_ref_tflops_long_k"] = tf
            out["v6_same_shapes_tflops"] = sum(tf6) / len(tf6)
            out["v6_same_shapes_tflops_short_k"], out["v6_same_shapes_tflops_long_k"] = tf6
            out["like_for_like"] = out["v6_same_shapes_tflops"] / out["blas_ref_tflops"]
            out["like_for_like_short_k"], out["like_for_like_long_k"] = tf6[0] / tf[0], tf6[1] / tf[1]
        except Exception as e:  # noqa: BLE001
            out["blas_ref_error"] = f"{type(e).__name__}: {e}"[:160]
        # HBM stream: copy_gb read + copy_gb written per launch
        nbytes = int(copy_gb * (1 << 30)) & ~15
        src = torch.empty(nbytes, dtype=torch.uint8, device=dev).random_(0, 255)
        dst = torch.empty_like(src)
        for _ in range(3):
            _C.check(lib.cx_calib_copy(src.data_ptr(), dst.data_ptr(), nbytes, s), "cx_calib_copy")
        torch.cuda.synchronize()
        ce = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
        ce[0].record()
        for i in range(10):
            _C.check(lib.cx_calib_copy(src.data_ptr(), dst.data_ptr(), nbytes, s), "cx_calib_copy")
            ce[i + 1].record()
        torch.cuda.synchronize()
        cms = sorted(ce[i].elapsed_time(ce[i + 1]) for i in range(10))
        out["hbm_copy_tbs"] = 2.0 * nbytes / (cms[len(cms) // 2] * 1e-3) / 1e12
        out["hbm_copy_gb"] = nbytes / 2**30
        del src, dst
        torch.cuda.empty_cache()
    out["n_cu"] = n_cu
    return out


class _Freqs(C.Structure):   # rsmi_frequencies_t (rocm_smi.h: has_deep_sleep, num_supported, current, frequency[33])
    _fields_ = [("has_deep_sleep", C.c_bool), ("num_supported", C.c_uint32), ("current", C.c_uint32), ("frequency", C.c_uint64 * 33)]


class SmiSampler:
    """Background sampler of socket power (W) and shader clock (MHz) of device `index` through librocm_smi64.  `index` is taken as the SMI
    device index = torch's ordinal: true on the one-GPU boxes this runs on and for an un-remapped node; under HIP_VISIBLE_DEVICES /
    ROCR_VISIBLE_DEVICES remapping the box_* power / clock fields may describe another GPU of the node (the timing fields are
    unaffected).  start() / stop();
    stop() returns {mean_power_w, max_power_w, mean_sclk_mhz, min_sclk_mhz, power_cap_w, samples, source}; values a box does not
    expose are None.  Never raises: calibration data must not be able to take the benchmark down."""

    def __init__(self, index: int = 0, hz: float = 10.0):
        self.index, self.period = int(index), 1.0 / hz
        self._stop = threading.Event()
        self._thread = None
        self._power, self._sclk = [], []
        self._cap = None
        self._lib = None
        self._why = None
        try:
            lib = C.CDLL("librocm_smi64.so")
            lib.rsmi_init.argtypes = [C.c_uint64]
            if lib.rsmi_init(0) != 0:
                raise OSError("rsmi_init failed")
            self._lib = lib
            cap = C.c_uint64(0)
            if lib.rsmi_dev_power_cap_get(C.c_uint32(self.index), C.c_uint32(0), C.byref(cap)) == 0 and cap.value:
                self._cap = cap.value / 1e6
        except Exception as e:  # noqa: BLE001
            self._why = f"{type(e).__name__}: {e}"[:120]

    def _read(self):
        lib, dv = self._lib, C.c_uint32(self.index)
        p, ptype = C.c_uint64(0), C.c_int(0)
        try:
            if lib.rsmi_dev_power_get(dv, C.byref(p), C.byref(ptype)) == 0 and p.value:
                self._power.append(p.value / 1e6)
            elif lib.rsmi_dev_current_socket_power_get(dv, C.byref(p)) == 0 and p.value:
                self._power.append(p.value / 1e6)
        except Exception:  # noqa: BLE001
            pass
        try:
            f = _Freqs()
            if lib.rsmi_dev_gpu_clk_freq_get(dv, C.c_int(0), C.byref(f)) == 0 and f.num_supported and f.current < 33:
                self._sclk.append(f.frequency[f.current] / 1e6)
        except Exception:  # noqa: BLE001
            pass

    def _run(self):
        while not self._stop.is_set():
            self._read()
            self._stop.wait(self.period)

    def start(self):
        if self._lib is not None and self._thread is None:
            self._stop.clear()
            self._power, self._sclk = [], []
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()
        return self

    def stop(self) -> dict:
        if self._thread is not None:
            self._stop.set()
            self._thread.join(timeout=2.0)
            self._thread = None
        mean = lambda v: (sum(v) / len(v)) if v else None   # noqa: E731
        return {"mean_power_w": mean(self._power), "max_power_w": max(self._power) if self._power else None,
                "mean_sclk_mhz": mean(self._sclk), "min_sclk_mhz": min(self._sclk) if self._sclk else None,
                "power_cap_w": self._cap, "samples": max(len(self._power), len(self._sclk)),
                "source": "librocm_smi64 (ctypes)" if self._lib is not None else f"unavailable ({self._why})"}


if __name__ == "__main__":
    import torch

    smp = SmiSampler(torch.cuda.current_device()).start()
    rec = calibrate()
    rec["during_probe"] = smp.stop()
    print(json.dumps(rec))
