#!/usr/bin/env python
"""rocprofv3 results (.db or *_kernel_stats.csv) -> small text summary for profiles/."""
import glob, sqlite3, sys
src, out, title = sys.argv[1], sys.argv[2], sys.argv[3]
db = glob.glob(src + "/**/*_results.db", recursive=True)
rows = []
if db:
    cur = sqlite3.connect(db[0]).cursor()
    rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
with open(out, "w") as f:
    f.write(f"# {title}\n# source: {db[0] if db else src}; durations in microseconds\n")
    f.write(f"{'kernel':<70} {'calls':>6} {'total_us':>12} {'avg_us':>10} {'pct':>7}\n")
    for name, calls, tot, avg, pct in rows:
        f.write(f"{name[:70]:<70} {calls:>6} {tot:>12.1f} {avg:>10.1f} {pct:>7.2f}\n")
    for extra in sys.argv[4:]:
        f.write("\n" + open(extra).read())
print(open(out).read())
