#!/usr/bin/env python
"""
Where the EM loop's wall time goes between its kernels: from a rocprofv3 --kernel-trace CSV of tools/run_pipeline.py,
the iteration kernels' own time, the finalize kernels', and the idle gaps between consecutive kernels of the loop
(by size class).
    python tools/loop_gaps.py <..._kernel_trace.csv> [iteration-kernel-name-prefix]
"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else "em_iter_quad_coded_kernel"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
it = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
lo, hi = it[0], it[-1]
span = rows[lo:hi + 2]
busy = {}
for r in span:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
    busy[name] = busy.get(name, [0, 0.0])
    busy[name][0] += 1
    busy[name][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
wall = (int(span[-1]["End_Timestamp"]) - int(span[0]["Start_Timestamp"])) / 1e6
print("loop: %d kernels over %.1f ms" % (len(span), wall))
for name, (n, ms) in sorted(busy.items(), key=lambda kv: -kv[1][1]):
    print("  %-50s %5d calls %9.2f ms (%.4f each)" % (name, n, ms, ms / n))
gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(span[:-1], span[1:])]
print("gaps between consecutive kernels: %d, total %.2f ms" % (len(gaps), sum(gaps) / 1e3))
for lo_us, hi_us in ((-1e9, 2), (2, 10), (10, 50), (50, 200), (200, 1e9)):
    sel = [g for g in gaps if lo_us <= g < hi_us]
    print("  %6s .. %-6s us: %5d gaps, %8.2f ms" % (lo_us if lo_us > -1e8 else "", hi_us if hi_us < 1e8 else "", len(sel), sum(sel) / 1e3))
durs = [(int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e6 for i in it]
n = len(durs)
print("iteration kernel: first 16 mean %.4f ms, last 16 mean %.4f ms, overall %.4f; slowest %.4f" %
      (sum(durs[:16]) / 16, sum(durs[-16:]) / 16, sum(durs) / n, max(durs)))
for a in range(0, n, max(1, n // 8)):
    part = durs[a:a + max(1, n // 8)]
    print("    iterations %4d..%4d: mean %.4f ms" % (a, a + len(part) - 1, sum(part) / len(part)))
