"""Per-cycle kernel breakdown of a bench run traced with
rocprofv3 --kernel-trace --output-format csv (argv[1] = *_kernel_trace.csv): averages over ten
steady-state SCF cycles, then the launch-by-launch timeline of one cycle with the idle gaps."""
import collections
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort(key=lambda r: r[1])
idx = [i for i, r in enumerate(rows) if re.search(r"jk_(s4|sym|dense)_kernel", r[0])]
i0, i1 = idx[10], idx[20]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z_0-9:]+(<[^(]*>)?)", n)
    return (m.group(1) if m else n)[:70]


print("cycle period ms", (rows[i1][1] - rows[i0][1]) / 10 / 1e6)
agg = collections.defaultdict(lambda: [0, 0.0])
busy = 0
for r in rows[i0:i1]:
    n = short(r[0])
    agg[n][0] += 1
    agg[n][1] += (r[2] - r[1]) / 1e6
    busy += (r[2] - r[1]) / 1e6
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:72s} {c / 10:6.1f}/cyc {t / 10 * 1000:8.1f} us/cyc")
print("busy ms/cycle", busy / 10, "kernels/cycle", (i1 - i0) / 10)
prev = None
for r in rows[idx[19]:idx[20] + 1]:
    gap = (r[1] - prev) / 1e3 if prev else 0
    print(f"gap {gap:7.1f} us  dur {(r[2] - r[1]) / 1e3:7.1f} us  {short(r[0])}")
    prev = r[2]

# where the idle time of every cycle sits (gap before the named kernel, > 3 us)
for c in range(0, len(idx) - 1):
    prev, tot, big = None, 0.0, []
    for r in rows[idx[c]:idx[c + 1] + 1]:
        if prev is not None:
            gap = (r[1] - prev) / 1e3
            tot += gap
            if gap > 3.0:
                big.append(f"{gap:.1f} before {short(r[0])[:24]}")
        prev = r[2]
    print(f"cycle {c}: period {(rows[idx[c + 1]][1] - rows[idx[c]][1]) / 1e3:7.1f} us  idle {tot:6.1f} us  "
          f"kernels {idx[c + 1] - idx[c]}  " + "; ".join(big))
