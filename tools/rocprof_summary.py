"""Turn a `rocprofv3 --kernel-trace --stats` output directory into the two files kept under profiles/:
  <out>.csv  the kernel_stats table as rocprofv3 wrote it
  <out>.md   the same table, readable, with the profiled command on top

usage: python tools/rocprof_summary.py <rocprof output dir> <out prefix> "<profiled command>" [iterations]
"""
import csv
import glob
import os
import shutil
import sys


def main():
    src, out, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
    iters = sys.argv[4] if len(sys.argv) > 4 else None
    found = sorted(glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True))
    if not found:
        sys.exit("no *kernel_stats.csv under %s" % src)
    shutil.copyfile(found[0], out + ".csv")
    rows = list(csv.DictReader(open(found[0])))
    total_ns = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(out + ".md", "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats : %s\n\n" % cmd)
        f.write("Total kernel time %.2f ms%s\n\n" % (total_ns / 1e6, " over %s" % iters if iters else ""))
        f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
        for r in rows:
            if float(r["Percentage"]) < 0.05:
                continue
            f.write("| `%s` | %s | %.2f | %.2f | %.1f |\n" % (
                r["Name"][:110], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                float(r["Percentage"])))


if __name__ == "__main__":
    main()
