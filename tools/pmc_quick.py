"""Per-kernel averages of one `rocprofv3 --kernel-trace --pmc ...` output directory (any counter set).
    python tools/pmc_quick.py <dir> [kernel-name substring]
"""
import collections
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    path = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if only not in name:
            continue
        per[name][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[name].add(r["Dispatch_Id"])
    for name, c in per.items():
        n = len(disp[name])
        print("%s  (%d launches)" % (name[:120], n))
        for k in sorted(c):
            print("    %-32s %16.0f per launch" % (k, c[k] / n))


if __name__ == "__main__":
    main()
