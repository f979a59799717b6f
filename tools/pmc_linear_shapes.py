"""Target + fold for per-SHAPE HBM traffic of the native linears (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes).

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/lf -- python tools/pmc_linear_shapes.py run
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/lw -- python tools/pmc_linear_shapes.py run
    python tools/pmc_linear_shapes.py fold <fetch.csv> <write.csv> [times.json] > profiles/r06_linear_pmc.txt

`run` launches every shape CALLS times behind a marker kernel (ATen tril of a 2 x 2 tensor), so the fold can cut the
dispatch list per shape without kernel arguments; `time` (no profiler) writes the same shapes' launch times to JSON.
Reads are 2 x FETCH_SIZE (gfx950: a 128-B request is tallied as 64 B, MI355X_MICROARCH.md HBM section)."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "co-detr-tensorrt_amd"))
CALLS = 2


def shapes(images=4, W_=1920, H_=1280):
    out = []
    C, h, w = 192, H_ // 4, W_ // 4
    for s in range(4):
        hp, wp = -(-h // 12) * 12, -(-w // 12) * 12
        tp, t = images * hp * wp, images * h * w
        out += [(f"swin{s}.qkv", tp, 3 * C, C, None, False), (f"swin{s}.proj", tp, C, C, None, True),
                (f"swin{s}.fc1", t, 4 * C, C, "gelu", False), (f"swin{s}.fc2", t, C, 4 * C, None, True)]
        C, h, w = 2 * C, -(-h // 2), -(-w // 2)
    S = images * 51150
    out += [("dec.value6", S, 1536, 256, None, False), ("enc.value", S, 256, 256, None, False)]
    return out


def algorithmic(M, N, K, res):
    return 2 * (M * K + N * K + M * N * (2 if res else 1)) + 2 * N


def run(timed):
    import torch
    from codetr import _cabi, hip_ops
    marker = torch.zeros(2, 2, device="cuda")
    times = {}
    for name, M, N, K, act, res in shapes():
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randn(M, K, device="cuda", generator=g).half()
        w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).half()
        b = torch.randn(N, device="cuda", generator=g).half()
        r = torch.randn(M, N, device="cuda", generator=g).half() if res else None
        hip_ops.linear(x, w, b, act=act, residual=r)           # warm (derived weights, attributes)
        torch.cuda.synchronize()
        torch.tril(marker)
        before = dict(_cabi.CALLS)
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                hip_ops.linear(x, w, b, act=act, residual=r)
            e1.record()
            torch.cuda.synchronize()
            kern = [k for k in ("linear_pp", "linear_sk", "linear_tile256", "linear_tile128", "linear_xs", "linear_splitk")
                    if _cabi.CALLS.get(k, 0) > before.get(k, 0)]
            times[name] = {"us": e0.elapsed_time(e1) / 20 * 1e3, "kernel": ",".join(kern)}
        else:
            for _ in range(CALLS):
                hip_ops.linear(x, w, b, act=act, residual=r)
        torch.tril(marker)                                      # closes the measured segment; what follows up to the next marker is set-up
        torch.cuda.synchronize()
        del x, w, b, r
    torch.cuda.synchronize()
    if timed:
        json.dump(times, sys.stdout)


def segments(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    segs, cur = [], None
    for r in rows:
        if "triu_tril" in r["Kernel_Name"]:
            if cur is not None:
                segs.append(cur)
            cur = []
        elif cur is not None:
            cur.append(float(r["Counter_Value"]) * 1024.0)
    return segs


def fold(fetch, write, times):
    sf, sw = segments(fetch, "FETCH_SIZE")[0::2], segments(write, "WRITE_SIZE")[0::2]
    sh = shapes()
    assert len(sf) == len(sh) == len(sw), (len(sf), len(sw), len(sh))
    tm = json.load(open(times)) if times else {}
    print("# per-shape HBM-side traffic of hip_ops.linear at the 4-image launch shapes (bytes per call; read = 2 x FETCH_SIZE)")
    print(f"# {'shape':11s} {'M':>7s} {'N':>5s} {'K':>5s}  {'algorithmic':>11s} {'read':>9s} {'write':>9s} {'traffic':>9s}  x alg   us    TB/s  kernel")
    for (name, M, N, K, act, res), f, w in zip(sh, sf, sw):
        rd, wr = 2 * sum(f) / CALLS, sum(w) / CALLS
        alg = algorithmic(M, N, K, res)
        t = tm.get(name, {})
        us = t.get("us", float("nan"))
        print(f"  {name:11s} {M:7d} {N:5d} {K:5d}  {alg / 1e6:9.1f}MB {rd / 1e6:7.1f}MB {wr / 1e6:7.1f}MB {(rd + wr) / 1e6:7.1f}MB  "
              f"{(rd + wr) / alg:5.2f} {us:6.1f} {(rd + wr) / us / 1e6 if us == us else 0:6.2f}  {t.get('kernel', '')}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(False)
    elif sys.argv[1] == "time":
        run(True)
    else:
        fold(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None)
