"""Per-pass HBM traffic of the search kernels from two rocprofv3 --pmc runs of bench.py
(one with FETCH_SIZE, one with WRITE_SIZE; the counters do not fit one pass).

usage: tools/traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <warmup> <steps> <out.json>

A pass is the search kernels up to and including one k_finalize.  The timed passes are
those after the warm-up registration.  FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE tallies 128-byte requests as 64 bytes for wide reads, so it is doubled (an upper
bound for the narrow gathers of a tree walk, which are not calibrated)."""
import csv
import json
import sys

SEARCH = ("k_search", "k_contested", "k_fallback", "k_certify")


def per_pass(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    passes, cur = [], 0.0
    for r in rows:
        k = r["Kernel_Name"]
        if any(s in k for s in SEARCH):
            cur += float(r["Counter_Value"])
        elif "k_finalize" in k:
            passes.append(cur)
            cur = 0.0
    return passes


def main():
    fetch, write, warmup, steps, out = sys.argv[1:6]
    warmup, steps = int(warmup), int(steps)
    f = per_pass(fetch, "FETCH_SIZE")[warmup:warmup + steps]
    w = per_pass(write, "WRITE_SIZE")[warmup:warmup + steps]
    assert len(f) == steps and len(w) == steps, (len(f), len(w))
    fetch_b = 2.0 * 1024.0 * sum(f) / steps
    write_b = 1024.0 * sum(w) / steps
    res = {
        "k_search_hbm_bytes_per_launch": fetch_b + write_b,
        "fetch_bytes_per_launch_corrected_x2": fetch_b,
        "write_bytes_per_launch": write_b,
        "passes_averaged": steps,
        "first_pass_fetch_bytes_corrected_x2": 2.0 * 1024.0 * f[0],
        "last_pass_fetch_bytes_corrected_x2": 2.0 * 1024.0 * f[-1],
        "note": "sum over the search kernels of one pass (shortcut, tree walk, tie resolution), "
                "averaged over the timed passes of bench.py; FETCH_SIZE doubled as for wide reads",
    }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
