"""HIP API calls longer than 25 us inside the SCF loop of a bench run traced with
rocprofv3 --hip-trace --kernel-trace --output-format csv (argv[1] = directory with the csv files):
which host call made the queue run dry."""
import csv
import glob
import re
import sys

d = sys.argv[1]
kt = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
at = sorted(glob.glob(d + "/**/*hip_api_trace.csv", recursive=True))[0]
jk = sorted(int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))
            if re.search(r"jk_(s4|sym|dense)_kernel", r["Kernel_Name"]))
calls = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]) for r in csv.DictReader(open(at))]
calls.sort()
t0, t1 = jk[2], jk[-1]
import bisect
for s, e, f in calls:
    if s < t0 or s > t1:
        continue
    if e - s > 25000:
        c = bisect.bisect_right(jk, s) - 1
        print(f"during GPU cycle {c:3d} (+{(s - jk[c]) / 1e3:7.1f} us)  {f:32s} {(e - s) / 1e3:8.1f} us")
