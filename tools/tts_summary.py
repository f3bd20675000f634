"""Kernel timeline of the LAST run of tools/tts_run.py from a rocprofv3 kernel trace csv: runs are
separated by the >= 50 ms sleeps; prints every launch with the idle gap before it and totals by kernel."""
import collections
import csv
import re
import sys

rows = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: r[1])
cuts = [0] + [i for i in range(1, len(rows)) if rows[i][1] - rows[i - 1][2] > 30e6] + [len(rows)]
run = rows[cuts[-2]:cuts[-1]]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z_0-9:]+(<[^(]*>)?)", n)
    return (m.group(1) if m else n)[:64]


t0 = run[0][1]
print(f"last run: {len(run)} launches, span {(run[-1][2] - t0) / 1e6:.3f} ms")
agg = collections.defaultdict(lambda: [0, 0.0])
prev = None
idle = 0.0
for n, s, e in run:
    gap = (s - prev) / 1e3 if prev else 0.0
    idle += gap
    agg[short(n)][0] += 1
    agg[short(n)][1] += (e - s) / 1e3
    if len(sys.argv) > 2:
        print(f"t {(s - t0) / 1e3:9.1f} us  gap {gap:7.1f}  dur {(e - s) / 1e3:8.1f}  {short(n)}")
    prev = e
print(f"idle (gaps) {idle / 1e3:.3f} ms")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{n:66s} {c:5d} x  {t / 1e3:8.3f} ms")
