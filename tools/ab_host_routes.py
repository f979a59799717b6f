"""A/B runs of the host package's route switches.  Since round 5 the package reads no CODETR_* environment variable
(tests/test_cabi.py::test_host_package_reads_no_route_switch_from_the_environment): the switches are module attributes
at their measured-best defaults, and this tool patches them from the OUTSIDE before it starts a bench / a script.

    python tools/ab_host_routes.py --set MSDA_V4_REGION=16x8 --set DEC_FUSED=0 -- bench.py --batch 4 --steps 5
    python tools/ab_host_routes.py --list

Switch names: hip_ops.{LN_GEMM, XADD, XADD_MIN_ROWS, MERGE_LN, MSDA_ENCODER, LINEAR_PP,
MSDA_FP32_REF, MSDA_V4 (+ _THREADS, _REGION, _LDS_BUDGET, _MARGIN_CAP, _HEAD_MAJOR), FP8_MIN_TILES}, transformer.{DEC_FUSED, DEC_VPROJ}, multi_scale_deformable_attention.HEAD_MAJOR_VALUE."""
import argparse
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "co-detr-tensorrt_amd"))
sys.path.insert(0, ROOT)


def _parse(name, old, text):
    """NAME=VALUE by the type of the switch's default: bool / int through int(), float through float(), a tuple as
    AxB[xC...] of ints, a string as it stands; any other type is refused instead of being set to something else."""
    if isinstance(old, bool):
        return bool(int(text))
    if isinstance(old, int):
        return int(text)
    if isinstance(old, float):
        return float(text)
    if isinstance(old, tuple):
        val = tuple(int(t) for t in text.lower().split("x"))
        if len(val) != len(old):
            raise SystemExit(f"{name}: expected {len(old)} values joined by 'x', got {text!r}")
        return val
    if isinstance(old, str):
        return text
    raise SystemExit(f"{name}: cannot parse a {type(old).__name__} switch from the command line")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--set", action="append", default=[], metavar="NAME=VALUE")
    ap.add_argument("--list", action="store_true")
    ap.add_argument("script", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    from codetr import hip_ops, multi_scale_deformable_attention as msda_mod, transformer as tr_mod

    homes = {"DEC_FUSED": tr_mod, "DEC_VPROJ": tr_mod, "HEAD_MAJOR_VALUE": msda_mod}
    if a.list:
        for k in sorted(hip_ops._SWITCH_DEFAULTS):
            print(f"hip_ops.{k} = {getattr(hip_ops, k)!r}")
        for k, m in sorted(homes.items()):
            print(f"{m.__name__.split('.')[-1]}.{k} = {getattr(m, k)!r}")
        return
    for item in a.set:
        k, v = item.split("=", 1)
        mod = homes.get(k, hip_ops)
        if not hasattr(mod, k):
            raise SystemExit(f"unknown switch {k}")
        setattr(mod, k, _parse(k, getattr(mod, k), v))
    print("non-default switches:", hip_ops.nondefault_switches(), file=sys.stderr)
    script = [s for s in a.script if s != "--"]
    if not script:
        raise SystemExit("nothing to run")
    sys.argv = script
    runpy.run_path(script[0], run_name="__main__")


if __name__ == "__main__":
    main()
