#!/usr/bin/env python3
"""Root-cause probe for the round-1 flake of tests/test_model_gpu.py::test_full_codetr_fp32_vs_oracle (about one
fresh-box run in twenty failed the final detection-set assertion and passed on an in-process retry).

Two hypotheses are separated here, on the GPU:

 (A) a product kernel is unstable (race, uninitialised read of a torch.empty workspace, stream-order bug): then the same
     inputs give different intermediates run to run.  Probe: the SAME model / inputs N times in one process, the
     caching allocator poisoned with NaN bit patterns between runs (so every torch.empty hands out NaNs), every
     captured stage hashed and compared bit for bit with the first run.
 (B) the assertion is brittle: it compared detections as sets of tuples rounded to 6 (score) / 2 (box) decimals, the
     product side computed on the GPU, the expected side by the oracle's decode on the CPU from the product's own
     logits -- a sigmoid that differs by one ulp next to a rounding boundary lands in a different cell.  Probe: many
     different seeded inputs, both the old rounding-based check and the tolerance-based check
     (helpers_model.unmatched_detections) evaluated, every disagreement printed with the nearest product detection.

Usage (GPU box):  python tools/diag_fp32_flake.py [--same 40] [--seeds 300] [--out gpurun_out/diag_fp32_flake.json]
"""
import argparse
import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "co-detr-tensorrt_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

DEV = "cuda:0"


def digest(t):
    if isinstance(t, (list, tuple)):
        return [digest(x) for x in t]
    return hashlib.sha1(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:16]


def old_check(boxes, scores, labels, bx, sc, lb):
    """the round-1 assertion, verbatim in behaviour: rounded tuples, set inclusion (image 0 only, as it was)"""
    own = {(round(float(s_), 6), int(l_), tuple(np.round(b_.numpy(), 2))) for s_, l_, b_ in
           zip(scores[0].cpu(), labels[0].cpu(), boxes[0].cpu()) if torch.isfinite(b_).all() and torch.isfinite(s_)}
    exp = {(round(float(s_), 6), int(l_), tuple(np.round(b_.numpy(), 2))) for s_, l_, b_ in zip(sc[0], lb[0], bx[0])
           if torch.isfinite(b_).all() and torch.isfinite(s_)}
    untied = {t for t in exp if sum(1 for u in exp if u[0] == t[0]) == 1 and t[0] > float(sc[0].min())}
    return sorted(untied - own), own


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--same", type=int, default=40)
    ap.add_argument("--seeds", type=int, default=300)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "diag_fp32_flake.json"))
    a = ap.parse_args()

    import codetr
    import codetr_fp32 as M
    from helpers_model import poison_allocator, seeded_params, unmatched_detections, valid_topk
    from test_model_gpu import _tiny_codetr_cfg

    torch.manual_seed(0)
    model = codetr.CoDETR(**_tiny_codetr_cfg("swin"))
    model.init_weights()
    spec = [(k, tuple(v.shape)) for k, v in model.named_parameters()]
    full = dict(model.state_dict())
    full.update(seeded_params(spec, 77, scale=1.5))
    model.load_state_dict(full)
    model = model.to(DEV).eval()
    H, W = 76, 100

    def inputs(seed):
        g = torch.Generator().manual_seed(seed)
        img = torch.randn(2, 3, H, W, generator=g)
        mask = torch.zeros(2, H, W)
        mask[1, :, int(W * 0.8):] = 1
        mask[1, int(H * 0.9):, :] = 1
        return img.to(DEV), mask.to(DEV)

    def run(img, mask, picks=None):
        cap = {}
        with torch.no_grad():
            out = model(img, mask, forced_topk_indices=picks, capture=cap)
        torch.cuda.synchronize()
        return out, cap

    report = {"device": torch.cuda.get_device_name(0)}

    # ---- (A) same inputs, poisoned allocator, bitwise stability of every stage
    img, mask = inputs(1)
    _, cap = run(img, mask)
    picks = valid_topk(cap["enc_outputs_class"].float().cpu(), cap["enc_outputs_coord_unact"].float().cpu(), 50).to(DEV)
    stages = ("backbone_feats", "neck_feats", "memory", "enc_outputs_class", "topk_coords_unact", "final_state",
              "outputs_classes", "outputs_coords")
    def stability(tag):
        first, prev, vs_first, vs_prev = None, None, {}, {}
        for it in range(a.same):
            poison_allocator(DEV)
            (b, s, l), cap = run(img, mask, picks)
            d = {}
            for k in stages:   # lists (feature levels) are reported per level
                v = cap[k]
                if isinstance(v, (list, tuple)):
                    d.update({f"{k}[{i}]": digest(x) for i, x in enumerate(v)})
                else:
                    d[k] = digest(v)
            d.update(boxes=digest(b), scores=digest(s), labels=digest(l))
            if first is not None:
                for k in d:
                    if d[k] != first[k]:
                        vs_first[k] = vs_first.get(k, 0) + 1
                    if d[k] != prev[k]:
                        vs_prev[k] = vs_prev.get(k, 0) + 1
            first = first or d
            prev = d
        order = list(d)
        first_moving = next((k for k in order if k in vs_first), None)
        report[tag] = {"runs": a.same, "runs_differing_from_run0": vs_first, "runs_differing_from_previous_run": vs_prev,
                       "first_stage_that_moves": first_moving}
        print(f"(A/{tag}) {a.same} runs of the same input, NaN-poisoned allocator: "
              + ("ALL stages bit-identical" if not vs_first else
                 f"first stage that moves: {first_moving}; vs run 0: {json.dumps(vs_first)}; vs previous run: {json.dumps(vs_prev)}"))

    stability("library_defaults")
    torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = True, False
    stability("cudnn_deterministic")
    torch.backends.cudnn.deterministic = False

    # ---- (B) many inputs: the old rounded-set assertion vs tolerance matching
    old_fail, new_fail, sig_ulps, events = 0, 0, 0, []
    for seed in range(1000, 1000 + a.seeds):
        img, mask = inputs(seed)
        _, cap = run(img, mask)
        picks = valid_topk(cap["enc_outputs_class"].float().cpu(), cap["enc_outputs_coord_unact"].float().cpu(), 50).to(DEV)
        (boxes, scores, labels), cap = run(img, mask, picks)
        bx, sc, lb = M.decode_detections(cap["outputs_classes"].cpu(), cap["outputs_coords"].cpu(), H, W, 20, 80)
        # how often does the GPU sigmoid differ from the CPU sigmoid at all?
        sg, sc_cpu = cap["outputs_classes"].sigmoid().cpu(), cap["outputs_classes"].cpu().sigmoid()
        sig_ulps += int((sg != sc_cpu).sum())
        miss_old, own = old_check(boxes, scores, labels, bx, sc, lb)
        miss_new = [m for bi in range(2) for m in
                    unmatched_detections((boxes[bi], scores[bi], labels[bi]), (bx[bi], sc[bi], lb[bi]))]
        if miss_old:
            old_fail += 1
            for t in miss_old:
                near = min(own, key=lambda u: (u[1] != t[1], abs(u[0] - t[0]) + sum(abs(x - y) for x, y in zip(u[2], t[2]))))
                events.append({"seed": seed, "missing_rounded_tuple": [t[0], t[1], [float(v) for v in t[2]]],
                               "nearest_product_tuple": [near[0], near[1], [float(v) for v in near[2]]]})
        if miss_new:
            new_fail += 1
            events.append({"seed": seed, "tolerance_check_missing": miss_new})
    n_sig = a.seeds * 2 * 50 * 80
    report.update(seeds=a.seeds, old_rounded_assertion_failures=old_fail, tolerance_assertion_failures=new_fail,
                  gpu_vs_cpu_sigmoid_differing_elements=sig_ulps, sigmoid_elements_compared=n_sig, events=events[:40])
    print(f"(B) {a.seeds} seeded inputs: old rounded-set assertion failed {old_fail}x, tolerance matching failed "
          f"{new_fail}x; GPU sigmoid != CPU sigmoid on {sig_ulps} of {n_sig} logits")
    for e in events[:10]:
        print("    ", json.dumps(e))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(report, f, indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
