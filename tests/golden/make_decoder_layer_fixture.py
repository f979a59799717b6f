"""Generate tests/golden/decoder_layer_1920x1280.npz: the fp32 CPU ORACLE's inputs and outputs of ONE decoder layer
(tests/decoder_layer_case.py) -- run in the build container (CPU, ~1 minute); the GPU test only reads the .npz.

    python tests/golden/make_decoder_layer_fixture.py

Two oracle runs: (1) the 3-layer decoder from the seeded start, to obtain realistic inputs of layer 1 (state, un-activated
reference boxes); those are rounded to fp16 -- what the product's layer would be handed -- and (2) layer 1 alone on the
rounded inputs.  Stored: the rounded layer inputs and, from run (2), the oracle's query_pos of the layer, the state after
the layer's third LayerNorm, the refined reference boxes and the NEXT layer's query_pos (sine embedding of the refined
boxes through ref_point_head), all 900 rows, fp32."""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "co-detr-tensorrt_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import codetr_fp32 as M  # noqa: E402
import decoder_layer_case as D  # noqa: E402


def main():
    t0 = time.time()
    dec, reg = D.build_decoder()
    sd = D.state_dict(dec, reg)
    memory, pad, vr, ss, start = D.memory_and_masks()
    query, ref = D.first_inputs()
    with torch.no_grad():
        cap = []
        M.decoder(sd, "dec", query, memory, pad, ref, vr, ss, start, "reg", layer_capture=cap)
        x_in = cap[D.LAYER]["x_in"].half().float()
        ref_in = cap[D.LAYER]["ref_in_unact"].half().float()
        # layer D.LAYER alone: a state dict whose layer 0 / reg branch 0 are that layer's, fed with the rounded inputs
        one = {k: v for k, v in sd.items() if ".layers." not in k and not k.startswith("reg.")}
        one.update({k.replace(f"dec.layers.{D.LAYER}.", "dec.layers.0."): v for k, v in sd.items()
                    if k.startswith(f"dec.layers.{D.LAYER}.")})
        one.update({k.replace(f"reg.{D.LAYER}.", "reg.0."): v for k, v in sd.items() if k.startswith(f"reg.{D.LAYER}.")})
        cap1 = []
        M.decoder(one, "dec", x_in, memory, pad, ref_in, vr, ss, start, "reg", layer_capture=cap1)
        c = cap1[0]
        vr4 = torch.cat((vr, vr), -1)
        nxt = c["ref_out_unact"].sigmoid()[:, :, None, :] * vr4[:, None]
        qpos_next = M._lin(one, "dec.ref_point_head.2", torch.relu(M._lin(one, "dec.ref_point_head.0", M._sine_embed(nxt[:, :, 0, :]))))
    np.savez_compressed(D.FIXTURE, x_in=x_in.half().numpy(), ref_in_unact=ref_in.half().numpy(), qpos=c["qpos"].numpy(),
                        x_out=c["x_out"].numpy(), ref_out_unact=c["ref_out_unact"].numpy(), qpos_next=qpos_next.numpy(),
                        seed=np.array(D.SEED), layer=np.array(D.LAYER))
    print(f"oracle {time.time() - t0:.1f} s -> {D.FIXTURE} ({os.path.getsize(D.FIXTURE) / 1024:.0f} KiB); "
          f"|x_out| {float(c['x_out'].norm()):.3f}  ref_out[0] {c['ref_out_unact'][0, 0].tolist()}")


if __name__ == "__main__":
    main()
