"""Per-phase kernel breakdown of one mid-run forward from a rocprofv3 --kernel-trace run of
`bench.py --no-graph` (rocpd .db output).  Forwards are delimited by the largest launch gap between the
MSDA launches of consecutive forwards (12 fused MSDA launches per forward).

    python tools/phase_breakdown.py gpurun_out/prof_x/x_results.db [forward_index] [top_n] [--csv out.csv]
"""
import collections
import csv
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::|at::native::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"elementwise_kernel_manual_unroll<128, \d, gpu_kernel_impl(_nocast)?<", "EW<", n)
    n = re.sub(r"vectorized_elementwise_kernel<\d+, ", "VEW<", n)
    return n[:100]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    db = sqlite3.connect(args[0])
    k = int(args[1]) if len(args) > 1 else 8
    top = int(args[2]) if len(args) > 2 else 12
    rows = db.execute("select name,start,end from kernels order by start").fetchall()
    if "--csv" in sys.argv:
        out = sys.argv[sys.argv.index("--csv") + 1]
        agg = collections.defaultdict(list)
        for n, s, e in rows:
            agg[n].append(e - s)
        tot = sum(sum(v) for v in agg.values())
        with open(out, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
                w.writerow([n, len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / tot, min(v), max(v)])
    ms = [i for i, r in enumerate(rows) if "msda_tiled_kernel" in r[0] and "true" in r[0]]

    def split(a, b):
        return max((rows[i + 1][1] - rows[i][2], i) for i in range(a, b))[1] + 1

    s = split(ms[12 * k - 1], ms[12 * k])
    e = split(ms[12 * k + 11], ms[12 * k + 12])
    sel = rows[s:e]
    names = [short(r[0]) for r in sel]
    m = [i for i, n in enumerate(names) if "msda_tiled_kernel" in n]
    gn = [i for i, n in enumerate(names) if "gn_partial" in n]
    ffn = [i for i, n in enumerate(names) if "ffn_fused" in n]
    enc_end = (ffn[5] + 2) if len(ffn) >= 6 else m[5] + 8
    phases = [("backbone", 0, gn[0] - 1), ("neck + masks + pos", gn[0] - 1, m[0] - 3), ("encoder", m[0] - 3, enc_end),
              ("two-stage head", enc_end, m[6] - 14), ("decoder + detections", m[6] - 14, len(sel))]
    print(f"forward {k}: {len(sel)} launches, {sum(r[2] - r[1] for r in sel) / 1e6:.3f} ms of kernel time")
    for nm, x, y in phases:
        seg = sel[x:y]
        print(f"{nm:22s} {len(seg):5d} launches {sum(r[2] - r[1] for r in seg) / 1e6:7.3f} ms")
        agg = collections.defaultdict(lambda: [0, 0])
        for r in seg:
            a = agg[short(r[0])]
            a[0] += r[2] - r[1]
            a[1] += 1
        for n, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
            print(f"      {d / 1e3:8.1f} us {c:4d}x {d / c / 1e3:7.1f}  {n}")


if __name__ == "__main__":
    main()
