#!/usr/bin/env python
"""
Summarise rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, collected in
separate runs as MI355X_MICROARCH.md prescribes) into per-kernel HBM traffic.

    python tools/pmc_summary.py FETCH.csv WRITE.csv [ROWS HAPS [STORAGE [CALIBRATION.json]]] > profiles/rNN/pmc_traffic_<shape>.json

ROWS / HAPS / STORAGE (default 1000000 / 5408 / f64, bench.py's default workload) are recorded as
"_workload"; bench.py only quotes a traffic figure whose workload AND storage match the run, from the
kernel instance the run launched (mxm_describe_stream_kernel).

Units and corrections (MI355X_MICROARCH.md, section HBM):
  * both counters count units of 1024 B;
  * on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced
    streaming read (16 B/lane), so it is doubled for kernels that read that way
    (listed in WIDE_READERS); other access widths are uncalibrated and reported
    raw (correction 1.0);
  * WRITE_SIZE is exact for 16-B-per-lane streaming stores.
"""
import collections
import csv
import json
import sys

# kernels whose reads are 16 B/lane coalesced streams (FETCH_SIZE x2 applies): the dense matrix's readers, and the
# encoder (two 16-byte loads per lane and chunk over whole dense rows; round 3's files reported it raw at exactly half
# the matrix: 22.4 GB of 43.26)
WIDE_READERS = ("em_iter_wide_kernel", "em_iter_wide_f32_kernel", "estep_wide_kernel", "linearize_wide_kernel",
                "encode_rows_kernel", "encode_wide_rows_kernel", "row_argmax_wide_kernel")
# kernels calibrated against a bare reader of exactly their bytes in the SAME counter pass (tools/pmc_calibrate_coded.py):
#   factor = known bytes / FETCH_SIZE of the bare reader
CALIBRATED = {"em_iter_coded_kernel": "diag_stream_coded_kernel", "em_fused_coded_kernel": "diag_stream_coded_kernel",
              "em_iter_quad_coded_kernel": "diag_stream_quads_kernel"}


def per_kernel(path):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        acc[name].append(float(row["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
    calib = None
    if len(sys.argv) > 6:                                  # the JSON line tools/pmc_calibrate_coded.py printed in that pass
        with open(sys.argv[6]) as fin:
            calib = json.loads([ln for ln in fin if ln.startswith("{")][-1])
    out = {}
    for name in sorted(set(fetch) | set(write)):
        if "_kernel" not in name or name.startswith("at::"):
            continue
        f, nf = fetch.get(name, (0.0, 0))
        w, nw = write.get(name, (0.0, 0))
        corr, why = (2.0, "16 B/lane stream (guide)") if name.startswith(WIDE_READERS) else (1.0, "uncalibrated: raw")
        base = next((b for k, b in CALIBRATED.items() if name.startswith(k)), None)
        bare = next((v for k, v in fetch.items() if base and k.startswith(base)), None)
        if calib is not None and bare is not None and bare[0] > 0:
            corr = calib["record_bytes_read_per_pass"] / (bare[0] * 1024.0)
            why = ("%s read the same records' %.4f GB in the same pass and showed FETCH_SIZE %.1f"
                   % (base, calib["record_bytes_read_per_pass"] / 1e9, bare[0]))
        out[name] = {"launches": max(nf, nw), "FETCH_SIZE_raw": f, "WRITE_SIZE_raw": w,
                     "fetch_correction": corr, "fetch_correction_source": why,
                     "hbm_read_bytes_per_launch": f * 1024.0 * corr,
                     "hbm_write_bytes_per_launch": w * 1024.0,
                     "hbm_bytes_per_launch": f * 1024.0 * corr + w * 1024.0}
    rows = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
    haps = int(sys.argv[4]) if len(sys.argv) > 4 else 5408
    storage = sys.argv[5] if len(sys.argv) > 5 else "f64"
    out["_workload"] = {"rows_per_gpu": rows, "haps": haps, "storage": storage,
                        **({"matrix": calib.get("matrix", "encoded")} if calib is not None else {}),
                        "command": ("rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 tools/pmc_calibrate_coded.py "
                                    "%d%s%s (tools/profile_round.sh)" % (rows, " --quads" if str(calib.get("kernel", "")).startswith("em_iter_quad") else "",
                                                                         " --records" if calib.get("matrix") == "records" else "")
                                    if calib is not None else
                                    "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py "
                                    "--steps 4 --warmup 1 --no-cpu-baseline (tools/profile_round.sh)")}
    json.dump(out, sys.stdout, indent=1)
    sys.stdout.write("\n")


if __name__ == "__main__":
    main()
