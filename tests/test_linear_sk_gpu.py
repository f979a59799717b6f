"""GPU parity of the persistent 256-tile GEMM (C ABI codetr_linear_sk_{f16,bf16}, csrc/gemm_sk.hip) against a plain
PyTorch fp32 reference of the same op, y = act(x @ w.T + b) (+ r), at the tolerance of tests/test_linear_gpu.py
(1 ulp of the rounded result + fp32 accumulation noise; with a residual the linear output is rounded before the add).

Covers: the default form (whole tiles, the operand stream running across tile boundaries -- so a workgroup with
several tiles is the case that matters), the stream-K split of the left-over tiles (flag 0x40: partial sums through the
workspace, last-arriver epilogue, counters back at zero), one workgroup per tile (0x20), edge tiles in M and N, every
epilogue, both 16-bit types, and that `hip_ops.linear` routes the large short-K layers here."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(x, w, b, r, act):
    y = x.float() @ w.float().t()
    if b is not None:
        y = y + b.float()
    if act == "relu":
        y = torch.relu(y)
    elif act == "gelu":
        y = torch.nn.functional.gelu(y)
    return y, r.float() if r is not None else None


def _run(M, N, K, dtype, bias, act, res, flags, seed=0):
    from codetr import _cabi

    g = torch.Generator(device=DEV).manual_seed(seed)
    x = torch.randn(M, K, device=DEV, generator=g).to(dtype)
    w = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).to(dtype)
    b = torch.randn(N, device=DEV, generator=g).to(dtype) if bias else None
    r = torch.randn(M, N, device=DEV, generator=g).to(dtype) if res else None
    y = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    _cabi.linear_sk(x, w, b, r, act, y, flags=flags)
    torch.cuda.synchronize()
    lin, rf = _ref(x, w, b, r, act)
    ref = lin + rf if rf is not None else lin
    ulp = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    tol = ulp * ref.abs() + 1e-3 * ulp * 64 + K * 2.0 ** -22
    if rf is not None:
        tol = tol + ulp * lin.abs()
    err = (y.float() - ref).abs()
    bad = ~(err <= tol)   # NaN (an element never written) counts
    assert not bad.any(), f"{int(bad.sum())} / {bad.numel()} outside 1 ulp; max err {err.nan_to_num(1e9).max().item()}"
    return y


# (M, N, K): one tile; ragged M and N; more tiles than CUs (several tiles per workgroup); a left-over round
SHAPES = [(256, 256, 128), (300, 200, 128), (1000, 520, 192), (257, 1544, 1024), (33000, 768, 128), (40320, 2304, 768)]


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("flags", [0, 0x40, 0x20])
def test_linear_sk_shapes(M, N, K, flags):
    _run(M, N, K, torch.float16, True, None, False, flags)


@pytest.mark.parametrize("bias,act,res", [(False, None, False), (True, "relu", False), (True, "gelu", False),
                                          (True, None, True), (False, "relu", True), (True, "gelu", True)])
@pytest.mark.parametrize("flags", [0, 0x40])
def test_linear_sk_epilogues(bias, act, res, flags):
    _run(70000, 384, 384, torch.float16, bias, act, res, flags, seed=3)   # 274 x 2 tiles: a left-over round to split


@pytest.mark.parametrize("K", [192, 768])
def test_waves_without_an_epilogue_wait_for_their_dma_pieces(K):
    """with N = 192 (one column tile, Swin stage-0 proj / fc2) the wn = 3 waves of every tile store nothing,
    and with M % 256 in 1..128 neither do the wm = 1 waves of the last tile row; such a wave must not take the relaxed
    `vmcnt(VMN + 32)` wait of the phases behind an epilogue (it would run ahead of LDS-DMA pieces everybody reads).
    A race shows up as rare wrong elements: many tiles per workgroup, several repeats, no residual."""
    for rep, M in enumerate((256 * 1300 + 1, 256 * 1300 + 128, 256 * 700 + 64, 256 * 1024 + 100)):
        for again in range(2):
            _run(M, 192, K, torch.float16, True, None, False, 0, seed=20 + rep)


def test_linear_sk_bf16():
    _run(33000, 776, 256, torch.bfloat16, True, "gelu", True, 0, seed=4)
    _run(33000, 776, 256, torch.bfloat16, True, None, True, 0x40, seed=5)


def test_stream_k_leaves_the_ticket_counters_at_zero_and_repeats_bit_for_bit():
    from codetr import _cabi

    a = _run(40320, 768, 768, torch.float16, True, None, True, 0x40, seed=6)
    ws = _cabi.linear_sk_workspace(torch.device(DEV))
    n_counters = ws.numel() - (ws.numel() // (2 * 8 * 8192 * 4 + 8 * 4)) * (2 * 8 * 8192 * 4)
    assert int(ws[-n_counters:].view(torch.int32).abs().sum()) == 0
    b = _run(40320, 768, 768, torch.float16, True, None, True, 0x40, seed=6)
    # the parts of a tile are summed in the order of their workgroup ids, whichever arrives last
    assert torch.equal(a, b)


def test_contract():
    from codetr import _cabi

    lib = _cabi.load()
    assert lib.codetr_linear_sk_supported(1000, 256, 192) == 1
    assert lib.codetr_linear_sk_supported(1000, 256, 64) == 0       # K < 128
    assert lib.codetr_linear_sk_supported(1000, 260, 192) == 0      # N % 8
    assert lib.codetr_linear_sk_supported(1000, 256, 200) == 0      # K % 64
    x = torch.zeros(1000, 192, dtype=torch.float16, device=DEV)
    w = torch.zeros(256, 192, dtype=torch.float16, device=DEV)
    y = torch.zeros(1000, 256, dtype=torch.float16, device=DEV)
    st = _cabi.current_stream_ptr(x.device)
    # the stream-K split needs its workspace; the default form does not
    assert lib.codetr_linear_sk_f16(st, x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), 1000, 256, 192, 0, None, 0,
                                    0x40) == -1
    assert lib.codetr_linear_sk_f16(st, x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), 1000, 256, 192, 0, None, 0,
                                    0) == 0
    assert lib.codetr_linear_sk_f16(st, x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), 1000, 256, 192, 3, None, 0,
                                    0) == -4
    torch.cuda.synchronize()


def test_hip_ops_routes_the_large_short_k_layers_here():
    from codetr import _cabi, hip_ops

    g = torch.Generator(device=DEV).manual_seed(7)
    x = torch.randn(161280, 384, device=DEV, generator=g).half()          # Swin stage 1 at four 1920x1280 images
    w = (torch.randn(1152, 384, device=DEV, generator=g) / 384 ** 0.5).half()
    b = torch.randn(1152, device=DEV, generator=g).half()
    before = dict(_cabi.CALLS)
    y = hip_ops.linear(x, w, b)
    assert _cabi.CALLS["linear_sk"] == before["linear_sk"] + 1
    ref = (x[:4096].float() @ w.float().t() + b.float())
    assert torch.allclose(y[:4096].float(), ref, rtol=2e-3, atol=2e-3)
    # a row mask or a long K stays on codetr_linear_*
    mask = torch.zeros(161280, dtype=torch.bool, device=DEV)
    hip_ops.linear(x, w, b, row_mask=mask)
    x2 = torch.randn(38400, 3072, device=DEV, generator=g).half()
    w2 = (torch.randn(768, 3072, device=DEV, generator=g) / 3072 ** 0.5).half()
    hip_ops.linear(x2, w2, None)
    assert _cabi.CALLS["linear_sk"] == before["linear_sk"] + 1
