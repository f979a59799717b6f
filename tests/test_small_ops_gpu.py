"""The small fp16 kernels of the decoder / head (csrc/small_ops.hip) against the ATen formulation they replace, bit for
bit, and the property they exist for: a steady-state fp16 forward of the whole model issues NO library (ATen / rocBLAS /
MIOpen / rocPRIM) kernels -- every launch is one of libcodetr_hip.so's, which is what the launch-plan runner replays."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _inference_mode():
    """the native small kernels serve inference (no autograd graph): the model always runs them under no_grad"""
    with torch.no_grad():
        yield


def _g(seed=0):
    return torch.Generator(device=DEV).manual_seed(seed)


def test_add_and_broadcast_add():
    from codetr import _cabi, hip_ops

    g = _g()
    a = torch.randn(3, 900, 256, device=DEV, generator=g).half()
    b = torch.randn(3, 900, 256, device=DEV, generator=g).half()
    before = _cabi.CALLS["small_ops"]
    assert torch.equal(hip_ops.add(a, b), a + b)
    w = torch.randn(900, 256, device=DEV, generator=g).half()
    assert torch.equal(hip_ops.add(w[None].expand(3, -1, -1), b), w[None] + b)
    assert _cabi.CALLS["small_ops"] == before + 2
    # shapes the kernel does not take fall back to ATen (odd sizes, fp32)
    assert torch.equal(hip_ops.add(a[..., :7].contiguous(), b[..., :7].contiguous()), a[..., :7] + b[..., :7])
    assert torch.equal(hip_ops.add(a.float(), b.float()), a.float() + b.float())


def test_sigmoid_gather_valid_ratios():
    from codetr import hip_ops

    g = _g(1)
    x = (torch.randn(2, 900, 80, device=DEV, generator=g) * 4).half()
    x[0, 0, :4] = torch.tensor([0.0, -30.0, 30.0, float("nan")], device=DEV).half()
    got, ref = hip_ops.sigmoid(x), x.sigmoid()
    same = (got == ref) | (got.isnan() & ref.isnan())
    assert same.float().mean() > 0.999                 # fp32 exp may differ in its last bit: an fp16 boundary case
    assert (got.float() - ref.float()).nan_to_num().abs().max() <= 2.0 ** -11
    src = torch.randn(2, 5000, 256, device=DEV, generator=g).half()
    idx = torch.randint(0, 5000, (2, 900), device=DEV, generator=g)
    assert torch.equal(hip_ops.gather_rows(src, idx), torch.gather(src, 1, idx[..., None].expand(-1, -1, 256)))
    p4 = torch.randn(2, 5000, 4, device=DEV, generator=g).half()
    assert torch.equal(hip_ops.gather_rows(p4, idx), torch.gather(p4, 1, idx[..., None].expand(-1, -1, 4)))
    counts = torch.randint(1, 480, (3, 5, 2), device=DEV, generator=g).float()
    wh = torch.tensor([[480.0, 320], [240, 160], [120, 80], [60, 40], [30, 20]], device=DEV).half()
    assert torch.equal(hip_ops.valid_ratios(counts, wh), counts.half() / wh)


def test_decode_boxes_matches_the_aten_sequence():
    from codetr import hip_ops
    from codetr.co_dino_head import bbox_cxcywh_to_xyxy

    g = _g(2)
    B, Nq, C, K, W, H = 2, 900, 80, 300, 1920, 1280
    unact = (torch.randn(B, Nq, 4, device=DEV, generator=g) * 3).half()
    unact[0, 0] = torch.tensor([20.0, -20.0, 20.0, 20.0], device=DEV).half()       # saturated: clamps engage
    unact[1, 5, 2] = float("nan")
    idx = torch.randint(0, Nq * C, (B, K), device=DEV, generator=g)
    idx[0, 0], idx[1, 0] = 0 * C + 3, 5 * C + 7
    boxes, labels = hip_ops.decode_boxes(unact, idx, C, W, H)
    q = idx // C
    ref = bbox_cxcywh_to_xyxy(torch.gather(unact.sigmoid(), 1, q[..., None].expand(-1, -1, 4)))
    scale = ref.new_tensor([W, H, W, H])
    ref = torch.minimum((ref * scale).clamp(min=0), scale)
    assert torch.equal(labels, idx % C)
    both_nan = boxes.isnan() & ref.isnan()
    diff = ((boxes.float() - ref.float()).abs().nan_to_num() > 0) & ~both_nan
    assert diff.float().mean() < 2e-3 and torch.equal(boxes.isnan(), ref.isnan())   # (sigmoid's last fp32 bit, as above)
    assert (boxes.float() - ref.float()).nan_to_num().abs().max() <= 1.0            # <= one fp16 step at 1920 px


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_steady_state_forward_issues_only_native_kernels(dtype):
    import codetr
    from codetr.export import is_own_kernel
    from test_model_gpu import _tiny_codetr_cfg
    from helpers_model import seeded_params

    torch.manual_seed(0)
    cfg = _tiny_codetr_cfg("swin")
    cfg["backbone"].update(embed_dims=64, num_heads=[2, 4, 8, 16], window_size=12)
    cfg["neck"]["in_channels"] = [64, 128, 256, 512]
    model = codetr.CoDETR(**cfg)
    spec = [(k, tuple(v.shape)) for k, v in model.named_parameters()]
    full = dict(model.state_dict())
    full.update(seeded_params(spec, 5, scale=1.0))
    model.load_state_dict(full)
    model = model.to(DEV).to(dtype).eval()
    # (even sizes at every Swin stage, as at all three BASELINE resolutions: an odd map would take PatchMerging's F.pad)
    img = torch.randn(2, 3, 160, 192, device=DEV, generator=_g(3)).to(dtype)
    mask = torch.zeros(2, 160, 192, device=DEV, dtype=dtype)
    mask[1, :, 170:] = 1
    with torch.no_grad():
        for _ in range(2):          # warm-up: shape-keyed caches, derived weights
            model(img, mask)
        torch.cuda.synchronize()
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
            model(img, mask)
            torch.cuda.synchronize()
    names = [e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    kernels = [n for n in names if "Memcpy" not in n and "Memset" not in n]
    assert len(kernels) > 100, "the profiler saw no kernels"
    # whitelist, as the plan exporter applies it: every device activity is one of libcodetr_hip.so's own kernels
    foreign = sorted({n[:120] for n in kernels if not is_own_kernel(n)})
    assert not foreign, f"device work outside libcodetr_hip.so on the steady-state {dtype} forward: {foreign}"
    assert not [n for n in names if "Memcpy" in n], "a device copy (torch .contiguous() / .to()) on the steady-state forward"
