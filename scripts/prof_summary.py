"""Summarise a rocprofv3 kernel trace (rocpd sqlite .db or *_kernel_trace.csv) into a per-kernel table."""
import csv
import sqlite3
import sys
from collections import defaultdict


def rows_from(path):
    if path.endswith(".db"):
        c = sqlite3.connect(path)
        for name, start, end in c.execute("select name, start, end from kernels"):
            yield name, (end - start)
    else:
        with open(path) as f:
            for r in csv.DictReader(f):
                yield r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])


def main(path, top=30):
    agg = defaultdict(list)
    for name, dur in rows_from(path):
        agg[name].append(dur)
    tot = sum(sum(v) for v in agg.values())
    print(f"# source: {path}\n# total kernel time {tot/1e6:.2f} ms over {sum(len(v) for v in agg.values())} dispatches")
    print(f"{'total_ms':>10} {'pct':>6} {'calls':>7} {'avg_us':>9} {'min_us':>9} {'max_us':>9}  kernel")
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:top]:
        s = sum(v)
        print(f"{s/1e6:10.2f} {100*s/tot:6.1f} {len(v):7d} {s/len(v)/1e3:9.1f} {min(v)/1e3:9.1f} {max(v)/1e3:9.1f}  {name[:110]}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30)
