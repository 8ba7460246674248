"""Idle gaps on the GPU between consecutive kernels of a `rocprofv3 --kernel-trace` run: how much time the device spent
waiting (for the host, for a copy, for an event) rather than computing.
    python tools/gap_report.py <rocprof output dir> [threshold_us]
"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    thr = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
    path = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))))
    gaps = []
    end = rows[0][1]
    prev = rows[0][2]
    for s, e, name in rows[1:]:
        if s - end > thr * 1e3:
            gaps.append(((s - end) / 1e3, prev[:50], name[:50]))
        if e > end:
            end, prev = e, name
    busy = sum(e - s for s, e, _ in rows) / 1e6
    span = (max(e for _, e, _ in rows) - rows[0][0]) / 1e6
    print("kernels %d, span %.1f ms, summed kernel time %.1f ms, gaps > %.0f us: %d totalling %.1f ms" %
          (len(rows), span, busy, thr, len(gaps), sum(g[0] for g in gaps) / 1e3))
    for g in sorted(gaps, reverse=True)[:12]:
        print("  %8.1f us  after %-50s before %s" % g)
    tail = [g[0] for g in gaps[len(gaps) // 2:]]
    if tail:
        tail.sort()
        print("  second half of the run: %d gaps, median %.1f us" % (len(tail), tail[len(tail) // 2]))


if __name__ == "__main__":
    main()
