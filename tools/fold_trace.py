"""Fold a rocprofv3 --kernel-trace run of `bench.py --no-graph` (rocpd .db) into the two small files kept under
profiles/: <prefix>_kernel_stats.csv (per-kernel totals over the whole run) and <prefix>_summary.txt (per-image kernel
time of the steady-state forwards, top kernels).

    python tools/fold_trace.py <results.db> <out_prefix> <images_per_forward> "<command line that was profiled>"
"""
import collections
import csv
import re
import sqlite3
import sys


def short(nm):
    nm = re.sub(r"\(anonymous namespace\)::|at::native::", "", nm)
    nm = re.sub(r"^void ", "", nm)
    nm = re.sub(r"elementwise_kernel_manual_unroll<128, \d, gpu_kernel_impl(_nocast)?<", "EW<", nm)
    nm = re.sub(r"vectorized_elementwise_kernel<\d+, ", "VEW<", nm)
    nm = re.sub(r"_ZN12_GLOBAL__N_1\d+(\w+?)(ILi\d+E)?E[vP].*", r"\1", nm)
    nm = re.sub(r"\(.*", "", nm)
    return nm[:70]


def main():
    db, prefix, imgs, cmd = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    rows = sqlite3.connect(db).execute("select name,start,end from kernels order by start").fetchall()
    agg = collections.defaultdict(list)
    for n, s, e in rows:
        agg[n].append(e - s)
    tot = sum(sum(v) for v in agg.values())
    with open(prefix + "_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([n, len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / tot, min(v), max(v)])
    # forwards are delimited by the 6 encoder launches of the packed MSDA kernel (exactly one per encoder layer at every
    # batch size; the fused FFN kernel -- the delimiter of rounds 1-5 -- runs TWICE per layer where a left-over partial round
    # goes to its 64-row form, which made the single-image summaries of rounds 5 and 6 count every forward as two); the fused
    # FFN's launches only where that kernel is absent; skip the first two forwards (warm-up)
    ffn = [i for i, r in enumerate(rows) if "msda_encoder_v4" in r[0]]
    if len(ffn) < 18:
        ffn = [i for i, r in enumerate(rows) if "ffn_fused" in r[0]]
    sel = rows[ffn[11] + 1:ffn[-1] + 1]
    n = (len(ffn) - 12) / 6
    t = sum(r[2] - r[1] for r in sel)
    out = [cmd, f"steady-state window of {n:.0f} forwards of {imgs} image(s): {t / n / imgs / 1e6:.3f} ms of kernel time per "
           f"image, {len(sel) / n:.0f} launches per forward", "", "   us/image  launches/fwd   avg us  kernel"]
    ag = collections.defaultdict(lambda: [0, 0])
    for r in sel:
        a = ag[short(r[0])]
        a[0] += r[2] - r[1]
        a[1] += 1
    for nm, (dd, c) in sorted(ag.items(), key=lambda kv: -kv[1][0])[:28]:
        out.append(f"  {dd / n / imgs / 1e3:9.1f}  {c / n:10.1f}  {dd / c / 1e3:8.1f}  {nm}")
    open(prefix + "_summary.txt", "w").write("\n".join(out) + "\n")
    print("\n".join(out[:12]))


if __name__ == "__main__":
    main()
