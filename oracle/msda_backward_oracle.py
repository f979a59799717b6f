"""ORACLE -- TEST INFRASTRUCTURE ONLY (not part of the product path).

CPU restatement of the BACKWARD of multi-scale deformable attention, following the reference's col2im kernels
explicitly (closed-form gradients, no autograd):

  bilinear gradient of one (sample point, channel)   codetr/csrc/ms_deform_attn.cu:79-146
  loop structure / index decode / range gate          codetr/csrc/ms_deform_attn.cu:263-336 (and the 5 sibling
                                                      kernels up to :760, which differ only in how the per-block
                                                      reduction over channels is carried out)
  accumulate-into-pre-zeroed-outputs contract         codetr/csrc/ms_deform_attn.cu:975-1028, codetr/ops.py:94-96

numpy, float64 (or float32) arithmetic; python loops over (image, query, head, level, point), vectorised over the
channel axis -- small cases only.  Pinned: tests/test_oracle_msda_backward.py checks it against gradients that
torch.autograd produced through the IMPORTED reference's differentiable formulation (ops.py:129-186), stored in
tests/golden/msda_grad.npz, for the reference's own gradient-test geometry with channels 4/30/32/64/71/1025.

Only tests/ may import this."""
import numpy as np


def msda_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, dtype=np.float64):
    """value [B,S,M,D], spatial_shapes [L,2] (h,w), level_start_index [L], sampling_loc [B,Nq,M,L,P,2] (x,y in [0,1]),
    attn_weight [B,Nq,M,L,P], grad_output [B,Nq,M*D] -> (grad_value, grad_sampling_loc, grad_attn_weight)"""
    value = np.asarray(value, dtype=dtype)
    loc = np.asarray(sampling_loc, dtype=dtype)
    aw = np.asarray(attn_weight, dtype=dtype)
    B, S, M, D = value.shape
    Nq, L, P = loc.shape[1], loc.shape[3], loc.shape[4]
    go = np.asarray(grad_output, dtype=dtype).reshape(B, Nq, M, D)
    gv = np.zeros_like(value)
    gl = np.zeros_like(loc)
    gw = np.zeros_like(aw)
    for b in range(B):
        for q in range(Nq):
            for m in range(M):
                top = go[b, q, m]                                  # [D]
                for l in range(L):
                    H, W = int(spatial_shapes[l][0]), int(spatial_shapes[l][1])
                    s0 = int(level_start_index[l])
                    for p in range(P):
                        w_im = loc[b, q, m, l, p, 0] * W - dtype(0.5)   # cu:304-305
                        h_im = loc[b, q, m, l, p, 1] * H - dtype(0.5)
                        if not (h_im > -1 and w_im > -1 and h_im < H and w_im < W):   # cu:309
                            continue
                        h0, w0 = int(np.floor(h_im)), int(np.floor(w_im))
                        h1, w1 = h0 + 1, w0 + 1
                        lh, lw = h_im - h0, w_im - w0
                        hh, hw = 1 - lh, 1 - lw
                        c1, c2, c3, c4 = hh * hw, hh * lw, lh * hw, lh * lw
                        a = aw[b, q, m, l, p]
                        tgv = top * a                                    # [D]
                        gh = np.zeros(D, dtype=dtype)
                        gwv = np.zeros(D, dtype=dtype)
                        val = np.zeros(D, dtype=dtype)
                        if h0 >= 0 and w0 >= 0:
                            v = value[b, s0 + h0 * W + w0, m]
                            gh -= hw * v
                            gwv -= hh * v
                            gv[b, s0 + h0 * W + w0, m] += c1 * tgv
                            val += c1 * v
                        if h0 >= 0 and w1 <= W - 1:
                            v = value[b, s0 + h0 * W + w1, m]
                            gh -= lw * v
                            gwv += hh * v
                            gv[b, s0 + h0 * W + w1, m] += c2 * tgv
                            val += c2 * v
                        if h1 <= H - 1 and w0 >= 0:
                            v = value[b, s0 + h1 * W + w0, m]
                            gh += hw * v
                            gwv -= lh * v
                            gv[b, s0 + h1 * W + w0, m] += c3 * tgv
                            val += c3 * v
                        if h1 <= H - 1 and w1 <= W - 1:
                            v = value[b, s0 + h1 * W + w1, m]
                            gh += lw * v
                            gwv += lh * v
                            gv[b, s0 + h1 * W + w1, m] += c4 * tgv
                            val += c4 * v
                        # the kernels reduce the per-channel contributions over the D channel-threads (cu:320-336)
                        gw[b, q, m, l, p] = np.sum(top * val)
                        gl[b, q, m, l, p, 0] = np.sum(W * gwv * tgv)
                        gl[b, q, m, l, p, 1] = np.sum(H * gh * tgv)
    return gv, gl, gw
