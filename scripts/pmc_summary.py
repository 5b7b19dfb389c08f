"""Per-kernel average of one PMC counter from rocprofv3's counter_collection.csv."""
import csv
import sys
from collections import defaultdict

path, ctr = sys.argv[1], sys.argv[2]
agg = defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(path)):
    if r["Counter_Name"] != ctr:
        continue
    a = agg[r["Kernel_Name"][:100]]
    a[0] += float(r["Counter_Value"])
    a[1] += 1
print(f"# {ctr}: sum over dispatches / dispatch count (raw counter units, KB for FETCH_SIZE/WRITE_SIZE)")
for k, (s, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:60]:
    print(f"{s:16.1f} total {n:7d} calls {s/n:14.2f} avg  {k}")
