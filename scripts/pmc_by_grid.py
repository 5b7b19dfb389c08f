"""Per (kernel, grid size) average of one PMC counter from rocprofv3's counter_collection.csv: tells the launches of one kernel
apart by their shapes (the four wgrad GEMMs of a layer have four different grids).  usage: pmc_by_grid.py CSV COUNTER [substr]"""
import csv
import sys
from collections import defaultdict

path, ctr = sys.argv[1], sys.argv[2]
sub = sys.argv[3] if len(sys.argv) > 3 else ""
agg = defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(path)):
    if r["Counter_Name"] != ctr or sub not in r["Kernel_Name"]:
        continue
    a = agg[(r["Kernel_Name"][:60], r.get("Grid_Size", "?"))]
    a[0] += float(r["Counter_Value"])
    a[1] += 1
print(f"# {ctr}: average per dispatch, by kernel and grid size (raw counter units: KB for FETCH_SIZE / WRITE_SIZE)")
for (k, g), (s, n) in sorted(agg.items()):
    print(f"{s / n:14.2f} avg {n:5d} calls  grid {g:>10s}  {k}")
