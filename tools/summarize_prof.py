"""Condense a rocprofv3 `*_kernel_stats.csv` into per-category totals (short names).

    python tools/summarize_prof.py gpurun_out/prof_x/.../NNN_kernel_stats.csv [steps]
"""
import csv
import re
import sys
from collections import defaultdict

CATS = [
    (r"^Cijk_|^Custom_Cijk", "GEMM (hipBLASLt/Tensile)"),
    (r"linear_kernel", "linear / GEMM (native MFMA)"),
    (r"msda_", "MSDA (native)"),
    (r"window_attention_kernel", "window attention (native)"),
    (r"layernorm_kernel", "layer norm (native)"),
    (r"layer_norm|layernorm|RowwiseMoments|GroupNorm|group_norm", "norm"),
    (r"softmax", "softmax"),
    (r"attn_fwd|flash|sdpa", "SDPA"),
    (r"direct_copy|copyBuffer|CatArray|roll_cuda|fill|FillFunctor|index|gather|scatter", "copy / cat / roll / gather"),
    (r"conv|Conv|igemm|naive_conv|miopen|MIOpen|Im2Col|im2col", "conv"),
    (r"elementwise|Functor|gelu|Gelu|clamp|sin_|cos_|sigmoid|exp|log", "elementwise"),
    (r"reduce|topk|sort|radix|scan|cumsum", "reduce / topk / scan"),
]


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"at::native::", "", name)
    m = re.match(r"(Cijk_\w+?_MT\d+x\d+x\d+)", name)
    if m:
        return m.group(1)
    return name[:90]


def main():
    path = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    rows = list(csv.DictReader(open(path)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    cat_tot = defaultdict(float)
    cat_calls = defaultdict(int)
    for r in rows:
        for pat, cat in CATS:
            if re.search(pat, r["Name"]):
                break
        else:
            cat = "other"
        cat_tot[cat] += float(r["TotalDurationNs"])
        cat_calls[cat] += int(r["Calls"])
    print(f"total kernel time {total / 1e6:.2f} ms over {steps:g} steps -> {total / 1e6 / steps:.2f} ms/step")
    for cat, t in sorted(cat_tot.items(), key=lambda kv: -kv[1]):
        print(f"  {cat:34s} {t / 1e6 / steps:8.3f} ms/step  {100 * t / total:5.1f}%  {cat_calls[cat] / steps:8.1f} calls/step")
    print("top kernels:")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:int(sys.argv[3]) if len(sys.argv) > 3 else 14]:
        print(f"  {float(r['TotalDurationNs']) / 1e6 / steps:8.3f} ms/step {int(r['Calls']) / steps:7.1f}x avg {float(r['AverageNs']) / 1e3:9.1f} us  {short(r['Name'])}")


if __name__ == "__main__":
    main()
