"""GPU parity of the ping-pong persistent GEMM (C ABI codetr_linear_pp_{f16,bf16}, csrc/gemm_pp.hip) against a plain
PyTorch fp32 reference of the same op, y = act(x @ w.T + b) (+ r), at the tolerance of tests/test_linear_gpu.py
(1 ulp of the rounded result + fp32 accumulation noise; with a residual the linear output is rounded before the add).

The kernel's correctness rests on its barrier / counted-wait protocol (two wave groups one barrier apart, LDS-DMA pieces
in flight across barriers), so beside shapes and epilogues there are: workgroups with several tiles (the operand stream runs
across tile boundaries), waves that store nothing, a left-over round, repeats under load compared bit for bit, and the routing
of `hip_ops.linear`."""
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


def _inputs(M, N, K, dtype, bias, res, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    x = torch.randn(M, K, device=DEV, generator=g).to(dtype)
    w = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).to(dtype)
    b = torch.randn(N, device=DEV, generator=g).to(dtype) if bias else None
    r = torch.randn(M, N, device=DEV, generator=g).to(dtype) if res else None
    return x, w, b, r


def _run(M, N, K, dtype, bias, act, res, seed=0):
    from codetr import _cabi

    x, w, b, r = _inputs(M, N, K, dtype, bias, res, seed)
    y = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    _cabi.linear_pp(x, w, b, r, act, y)
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


# (M, N, K): one tile; ragged M and N; few k-tiles (K = 128: a tile is four half-stages, the ring's depth); more tiles than
# CUs (several tiles per workgroup); a left-over round; the long-K layers this kernel serves (Swin stage 2 fc2, stage 3 fc2)
SHAPES = [(256, 256, 128), (300, 200, 128), (1000, 520, 192), (257, 1544, 1024), (33000, 768, 128), (40320, 2304, 768),
          (38400, 768, 3072), (9600, 1536, 6144)]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_linear_pp_shapes(M, N, K):
    _run(M, N, K, torch.float16, True, None, False)


@pytest.mark.parametrize("bias,act,res", [(False, None, False), (True, "relu", False), (True, "gelu", False),
                                          (True, None, True), (False, "relu", True), (True, "gelu", True)])
def test_linear_pp_epilogues(bias, act, res):
    _run(70000, 384, 384, torch.float16, bias, act, res, seed=3)   # 274 x 2 tiles: two rounds and a left-over round


def test_waves_without_an_epilogue_keep_the_protocol():
    """N = 192 (one column tile): the strips 3 of every tile store nothing, and with M % 256 in 1 .. 128 neither does group 1
    of the last tile row -- such a wave must not take the relaxed wait of the LOAD segments behind an epilogue."""
    for rep, M in enumerate((256 * 700 + 1, 256 * 700 + 128, 256 * 400 + 64)):
        for K in (192, 768):
            _run(M, 192, K, torch.float16, True, None, False, seed=20 + rep)


def test_linear_pp_bf16():
    _run(33000, 776, 256, torch.bfloat16, True, "gelu", True, seed=4)
    _run(9600, 1536, 1536, torch.bfloat16, True, None, True, seed=5)


def test_repeats_under_load_are_bit_identical():
    """a missed wait / an early refill of a ring slot shows up as a sporadic difference between launches of the same inputs"""
    from codetr import _cabi

    side = torch.cuda.Stream()
    noise_a = torch.randn(32 << 20, device=DEV).half()
    noise_b = torch.empty_like(noise_a)
    for (M, N, K, res) in ((40320, 768, 768, True), (11520, 4608, 1536, False), (38400, 768, 3072, True)):
        x, w, b, r = _inputs(M, N, K, torch.float16, True, res, seed=M % 97)
        first = None
        for it in range(8):
            with torch.cuda.stream(side):
                noise_b.copy_(noise_a)
            y = torch.empty((M, N), dtype=torch.float16, device=DEV)
            _cabi.linear_pp(x, w, b, r, None, y)
            torch.cuda.synchronize()
            if first is None:
                first = y
            else:
                assert torch.equal(y, first), f"launch {it} of {M}x{N}x{K} differs from the first"


def test_contract():
    from codetr import _cabi

    lib = _cabi.load()
    assert lib.codetr_linear_pp_supported(1000, 256, 192) == 1
    assert lib.codetr_linear_pp_supported(1000, 256, 64) == 0       # K < 128
    assert lib.codetr_linear_pp_supported(1000, 260, 192) == 0      # N % 8
    assert lib.codetr_linear_pp_supported(1000, 256, 200) == 0      # K % 64
    x = torch.zeros(1000, 192, dtype=torch.float16, device=DEV)
    w = torch.zeros(256, 192, dtype=torch.float16, device=DEV)
    y = torch.zeros(1000, 256, dtype=torch.float16, device=DEV)
    st = _cabi.current_stream_ptr(x.device)
    assert lib.codetr_linear_pp_f16(st, x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), 1000, 256, 192, 0, 0) == 0
    assert lib.codetr_linear_pp_f16(st, x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), 1000, 256, 192, 3, 0) == -4   # act
    assert lib.codetr_linear_pp_f16(st, x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), 1000, 256, 192, 0, 1) == -4   # flags
    assert lib.codetr_linear_pp_f16(st, None, w.data_ptr(), None, None, y.data_ptr(), 1000, 256, 192, 0, 0) == -1
    # the library's own rule: long K, a tile per CU, little of a 256-wide tile wasted
    assert lib.codetr_linear_pp_preferred(38400, 768, 3072, 0, 1) == 1
    assert lib.codetr_linear_pp_preferred(9600, 1536, 6144, 0, 1) == 1
    assert lib.codetr_linear_pp_preferred(161280, 1152, 384, 0, 0) == 0     # K = 384: the persistent kernel of round 4
    assert lib.codetr_linear_pp_preferred(614400, 192, 768, 0, 1) == 0      # N = 192
    assert lib.codetr_linear_pp_preferred(2400, 1536, 6144, 0, 1) == 0      # one image: 60 tiles
    torch.cuda.synchronize()


def test_hip_ops_routes_the_long_k_layers_here():
    from codetr import _cabi, hip_ops

    g = torch.Generator(device=DEV).manual_seed(7)
    x = torch.randn(38400, 3072, device=DEV, generator=g).half()            # Swin stage-2 fc2 at four 1920x1280 images
    w = (torch.randn(768, 3072, device=DEV, generator=g) / 3072 ** 0.5).half()
    b = torch.randn(768, device=DEV, generator=g).half()
    r = torch.randn(38400, 768, device=DEV, generator=g).half()
    before = dict(_cabi.CALLS)
    y = hip_ops.linear(x, w, b, residual=r)
    assert _cabi.CALLS["linear_pp"] == before["linear_pp"] + 1
    ref = (x[:4096].float() @ w.float().t() + b.float()).half().float() + r[:4096].float()
    assert torch.allclose(y[:4096].float(), ref, rtol=2e-3, atol=2e-3)
    # a row mask stays on codetr_linear_*, and the switch takes the route away
    hip_ops.linear(x, w, b, row_mask=torch.zeros(38400, dtype=torch.bool, device=DEV))
    hip_ops.LINEAR_PP = False
    try:
        hip_ops.linear(x, w, b, residual=r)
    finally:
        hip_ops.LINEAR_PP = True
    assert _cabi.CALLS["linear_pp"] == before["linear_pp"] + 1
