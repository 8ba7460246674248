"""Per-kernel summary of three `rocprofv3 --kernel-trace --pmc ...` passes over the same program:
    pass 1: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
    pass 2: FETCH_SIZE          pass 3: WRITE_SIZE
(each in its own run: the TCC block cannot hold FETCH_SIZE and WRITE_SIZE together).

usage: python tools/pmc_summary.py <dir pass1> <dir pass2> <dir pass3> <out.md> "<profiled command>"
Also writes <out>.json: {kernel name: {"hbm_bytes_per_launch", "launches", "avg_us", "read_bytes", "write_bytes", "gbps"}},
which bench.py reads to report the dominant kernel's measured traffic next to its algorithmic work and the achieved HBM
rate of the bandwidth-bound kernels.

HBM bytes follow the gfx950 reading of the counters: FETCH_SIZE and WRITE_SIZE are in KiB, and FETCH_SIZE counts a
128-byte request of a wide (16 B per lane) coalesced read as 64 bytes, so reads = 2 * FETCH_SIZE * 1024 (an upper
bound for kernels with narrow reads), writes = WRITE_SIZE * 1024. Infinity-cache hits are included in both.
MFMA flops = (SQ_INSTS_VALU_MFMA_MOPS_F32 + SQ_INSTS_VALU_MFMA_MOPS_BF16, whichever the pass collected) * 512.  Effective clock = GRBM_GUI_ACTIVE / 8 XCDs / duration.
"""
import collections
import csv
import glob
import json
import os
import sys


def counters(d):
    path = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    per = collections.defaultdict(lambda: collections.defaultdict(float))   # dispatch -> counter -> value
    names = {}
    for r in csv.DictReader(open(path)):
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        names[int(r["Dispatch_Id"])] = r["Kernel_Name"]
    return per, names


def durations(d):
    path = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    return {int(r["Dispatch_Id"]): int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(path))}


def main():
    d1, d2, d3, out, cmd = sys.argv[1:6]
    c1, names = counters(d1)
    c2, n2 = counters(d2)
    c3, n3 = counters(d3)
    dur = durations(d1)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for disp, name in names.items():
        a = agg[name]
        a["calls"] += 1
        a["ns"] += dur.get(disp, 0)
        for k, v in c1[disp].items():
            a[k] += v
    # the same program dispatches the same kernels in the same order in every pass
    for cx, nx, key in ((c2, n2, "FETCH_SIZE"), (c3, n3, "WRITE_SIZE")):
        for disp, name in nx.items():
            agg[name][key] += cx[disp].get(key, 0.0)
            agg[name][key + "_calls"] += 1
    traffic = {}
    for name, a in agg.items():
        if a["FETCH_SIZE_calls"] and a["WRITE_SIZE_calls"]:
            rd = 2.0 * a["FETCH_SIZE"] * 1024 / a["FETCH_SIZE_calls"]
            wr = a["WRITE_SIZE"] * 1024 / a["WRITE_SIZE_calls"]
            us = a["ns"] / max(a["calls"], 1) / 1e3
            traffic[name] = {"hbm_bytes_per_launch": round(rd + wr), "launches": int(a["calls"]),
                             "avg_us": round(us, 2), "read_bytes": round(rd), "write_bytes": round(wr),
                             "gbps": round((rd + wr) / (us * 1e-6) / 1e9, 1) if us > 0 else None}
    with open(os.path.splitext(out)[0] + ".json", "w") as f:
        json.dump({"command": cmd, "kernels": traffic}, f, indent=1, sort_keys=True)
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --pmc (3 passes) : %s\n\n" % cmd)
        f.write("Per launch averages. read = 2 x FETCH_SIZE KiB (gfx950 wide-read correction), write = WRITE_SIZE KiB, "
                "both include infinity-cache hits; MFMA TFLOP/s = MOPS_{F32,BF16} x 512 / duration; clock = GRBM_GUI_ACTIVE / 8 "
                "/ duration (reads high for launches well under 0.3 ms); durations are those of the profiled (serialised) "
                "pass 1.\n\n")
        f.write("| kernel | calls | avg us | read MB | write MB | GB/s | MFMA TFLOP/s | MFMA busy / CU busy | clock GHz |\n")
        f.write("|---|---|---|---|---|---|---|---|---|\n")
        for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["ns"]):
            if a["ns"] <= 0 or a["ns"] / sum(x["ns"] for x in agg.values()) < 0.002:
                continue
            n = a["calls"]
            us = a["ns"] / n / 1e3
            rd = 2.0 * a["FETCH_SIZE"] * 1024 / max(a["FETCH_SIZE_calls"], 1)
            wr = a["WRITE_SIZE"] * 1024 / max(a["WRITE_SIZE_calls"], 1)
            gbs = (rd + wr) / (us * 1e-6) / 1e9
            tf = (a["SQ_INSTS_VALU_MFMA_MOPS_F32"] + a["SQ_INSTS_VALU_MFMA_MOPS_BF16"]) * 512 / n / (us * 1e-6) / 1e12
            busy = a["SQ_VALU_MFMA_BUSY_CYCLES"] / a["SQ_BUSY_CU_CYCLES"] if a["SQ_BUSY_CU_CYCLES"] else 0.0
            clk = a["GRBM_GUI_ACTIVE"] / 8 / a["ns"]
            f.write("| `%s` | %d | %.1f | %.1f | %.1f | %.0f | %s | %s | %.2f |\n" % (
                name[:100], n, us, rd / 1e6, wr / 1e6, gbs, "%.1f" % tf if tf > 0.05 else "-",
                "%.2f" % busy if tf > 0.05 else "-", clk))


if __name__ == "__main__":
    main()
