"""Per-kernel memory-side traffic from two rocprofv3 --pmc runs of one bench.py command
(one with FETCH_SIZE, one with WRITE_SIZE; the counters do not fit one pass).

usage: tools/traffic_kernels.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>

Per kernel name: launches, mean / minimum / maximum bytes fetched and written per launch.
FETCH_SIZE and WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte requests
of wide streaming reads as 64 bytes, so both the raw figure and its double are given: the
double is right for coalesced 16-byte-per-lane reads and an upper bound for the narrow
gathers of a tree walk.  Both counters sit on the L2's memory side: reads that the
256 MiB on-die cache serves are counted as well."""
import csv
import json
import re
import sys


def per_kernel(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("icpx::", "")
        out.setdefault(name, []).append(1024.0 * float(r["Counter_Value"]))
    return out


def main():
    fetch, write, dst = sys.argv[1:4]
    f = per_kernel(fetch, "FETCH_SIZE")
    w = per_kernel(write, "WRITE_SIZE")
    res = {}
    for k in sorted(f, key=lambda k: -sum(f[k])):
        fv, wv = f[k], w.get(k, [0.0])
        res[k] = {
            "launches": len(fv),
            "fetch_bytes_mean_raw": sum(fv) / len(fv),
            "fetch_bytes_mean_x2": 2.0 * sum(fv) / len(fv),
            "fetch_bytes_min_x2": 2.0 * min(fv),
            "fetch_bytes_max_x2": 2.0 * max(fv),
            "write_bytes_mean": sum(wv) / len(wv),
            "write_bytes_max": max(wv),
        }
    json.dump(res, open(dst, "w"), indent=1)
    for k, v in list(res.items())[:12]:
        print("%-40s n=%4d fetch x2 mean %.3e (min %.3e max %.3e)  write mean %.3e" % (
            k[:40], v["launches"], v["fetch_bytes_mean_x2"], v["fetch_bytes_min_x2"],
            v["fetch_bytes_max_x2"], v["write_bytes_mean"]))


if __name__ == "__main__":
    main()
