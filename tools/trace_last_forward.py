"""Per-kernel breakdown of the LAST forward in a rocprofv3 --kernel-trace CSV (eager run of bench.py).
Forwards are delimited by the MSDA launches (12 per forward); everything between the launch that follows the
previous forward's last MSDA call and this forward's end is attributed to it.

    python tools/trace_last_forward.py <..._kernel_trace.csv> [top_n]
"""
import collections
import csv
import re
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    msda = [i for i, r in enumerate(rows) if "msda_" in r["Kernel_Name"]]
    per_fwd = 12
    if len(msda) < 2 * per_fwd:
        sys.exit("need at least two forwards in the trace")
    # a forward = from the first kernel after the previous forward's final top-k ... approximate with the midpoint
    # between the previous forward's last MSDA and this forward's first MSDA, measured in kernels of the backbone:
    last_first, prev_last = msda[-per_fwd], msda[-per_fwd - 1]
    # the backbone (hundreds of kernels) precedes the first MSDA of a forward; the decoder tail (~150 kernels) follows
    # the last one.  Split at the largest launch gap between the two MSDA groups (host-side gap between forwards).
    gaps = [(int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"]), i) for i in range(prev_last, last_first)]
    split = max(gaps)[1] + 1
    sel = rows[split:]
    agg = collections.defaultdict(lambda: [0, 0])
    for r in sel:
        n = re.sub(r"\(anonymous namespace\)::|at::native::", "", r["Kernel_Name"])
        n = re.sub(r"^void ", "", n)[:110]
        agg[n][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        agg[n][1] += 1
    tot = sum(v[0] for v in agg.values())
    print(f"last forward: {len(sel)} launches, {tot / 1e6:.2f} ms of kernel time")
    for n, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
        print(f"{d / 1e6:8.3f} ms {c:5d}x {d / c / 1e3:8.1f} us  {n}")


if __name__ == "__main__":
    main()
