#!/usr/bin/env python
"""
Per-kernel summary of a rocprofv3 SQ counter pass (tools/profile_round.sh: SQ_WAVE_CYCLES,
SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY, SQ_ACTIVE_INST_VALU, SQ_INSTS_VALU,
SQ_LDS_BANK_CONFLICT, SQ_LDS_IDX_ACTIVE, GRBM_GUI_ACTIVE): where the waves' cycles go.

    python tools/sq_summary.py profiles/rNN/pmc_sq.csv

WAIT_ANY (parked in s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY add up to
the wave cycles (MI355X_MICROARCH.md, rocprofv3 PMC slots); GRBM_GUI_ACTIVE is summed over the 8
XCDs, so clock = GUI / 8 / duration.
"""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    name = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
    acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("%-40s %5s %8s %8s %8s %8s %12s %10s" % ("kernel", "n", "parked", "stalled", "issuing", "(VALU)", "VALU insts", "LDS confl"))
for name, counters in acc.items():
    if "_kernel" not in name or name.startswith("at::"):
        continue
    m = {c: sum(v) / len(v) for c, v in counters.items()}
    wc = m.get("SQ_WAVE_CYCLES") or 1.0
    idx = m.get("SQ_LDS_IDX_ACTIVE") or 0.0
    print("%-40s %5d %7.0f%% %7.0f%% %7.0f%% %7.0f%% %12.3e %9.1f%%"
          % (name[:40], len(next(iter(counters.values()))), 100 * m.get("SQ_WAIT_ANY", 0) / wc,
             100 * m.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc,
             100 * m.get("SQ_ACTIVE_INST_VALU", 0) / wc, m.get("SQ_INSTS_VALU", 0),
             (100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / idx) if idx else 0.0))
