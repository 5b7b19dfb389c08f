"""Per-kernel averages of every counter in a rocprofv3 counter_collection.csv, plus derived MFMA utilisation.

usage: python scripts/pmc_multi.py <counter_collection.csv> [substring filter]
MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 4 SIMDs * CUs) -- SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed
over SIMDs (MI355X_MICROARCH.md: 32 x N_mfma for 32x32x16 bf16), GRBM_GUI_ACTIVE is the kernel's wall time in shader
clocks summed over XCDs (8), so the denominator is GRBM/8 * 256 CUs * 4 SIMDs."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"]
    if flt and flt not in k:
        continue
    a = agg[k[:90]][r["Counter_Name"]]
    a[0] += float(r["Counter_Value"])
    a[1] += 1
for k, ctrs in sorted(agg.items(), key=lambda kv: -sum(v[0] for v in kv[1].values())):
    n = max(v[1] for v in ctrs.values())
    print(f"== {k}  ({n} dispatches)")
    avg = {c: v[0] / max(1, v[1]) for c, v in ctrs.items()}
    for c in sorted(avg):
        print(f"   {c:34s} {avg[c]:18.1f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and "GRBM_GUI_ACTIVE" in avg and avg["GRBM_GUI_ACTIVE"] > 0:
        g = avg["GRBM_GUI_ACTIVE"]
        for xcds in (1, 8):
            print(f"   MfmaUtil (GRBM summed over {xcds} XCD{'s' if xcds > 1 else ''})   {100 * avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (g / xcds * 256 * 4):10.1f} %")
    if "SQ_WAVE_CYCLES" in avg:
        w = avg["SQ_WAVE_CYCLES"]
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_LDS"):
            if c in avg:
                print(f"   {c + ' / SQ_WAVE_CYCLES':34s} {100 * avg[c] / w:10.1f} %")
