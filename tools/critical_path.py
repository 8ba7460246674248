"""Which kernels a multi-stream iteration is waiting for: from a `rocprofv3 --kernel-trace` output directory, the time
during which exactly one kernel is running (and so nothing hides it), the time no kernel is running, and the time with
two or more, each broken down by kernel name.
    python tools/critical_path.py <rocprof output dir> [window ms] [end offset ms]
The window (default 40 ms) ends `end offset` (default 2 ms) before the last kernel: the replayed iterations at the end of
a benchmark run made with --no-kernel-timing (with kernel timing the run ends in eager, instrumented iterations).
"""
import collections
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    window = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
    path = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
    rows.sort()
    t_hi = max(r[1] for r in rows) - (float(sys.argv[3]) if len(sys.argv) > 3 else 2.0) * 1e6
    t_lo = t_hi - window * 1e6
    rows = [r for r in rows if r[0] >= t_lo and r[1] <= t_hi]
    events = []
    for i, (s, e, n) in enumerate(rows):
        events.append((s, 1, i))
        events.append((e, -1, i))
    events.sort()
    running = set()
    alone = collections.Counter()
    shared = collections.Counter()
    idle = 0
    last = events[0][0]
    for t, kind, i in events:
        dt = t - last
        if dt > 0:
            if not running:
                idle += dt
            elif len(running) == 1:
                alone[rows[next(iter(running))][2]] += dt
            else:
                for j in running:
                    shared[rows[j][2]] += dt / len(running)
        last = t
        if kind == 1:
            running.add(i)
        else:
            running.discard(i)
    total = events[-1][0] - events[0][0]
    print("span %.2f ms: idle %.1f %%, one kernel %.1f %%, two or more %.1f %%" %
          (total / 1e6, 100.0 * idle / total, 100.0 * sum(alone.values()) / total, 100.0 * sum(shared.values()) / total))
    print("time with nothing beside it, by kernel (% of span):")
    for n, v in alone.most_common(25):
        print("  %5.1f  %s" % (100.0 * v / total, n[:110]))


if __name__ == "__main__":
    main()
