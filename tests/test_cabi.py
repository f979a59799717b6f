"""CPU: the C-ABI library loads and exports every symbol include/codetr_hip.h declares; pure-host
entry points behave.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "codetr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(codetr_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    syms = _declared_symbols()
    for s in ("codetr_msda_forward_f16", "codetr_msda_forward_bf16", "codetr_msda_forward_f32",
              "codetr_msda_forward_f64", "codetr_hip_abi_version", "codetr_hip_strerror"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    from codetr import _cabi

    lib = ctypes.CDLL(_cabi.LIB_PATH)
    for s in _declared_symbols():
        assert hasattr(lib, s), f"{s} declared in include/codetr_hip.h but not exported"
    # and the ctypes table binds exactly the declared set
    assert sorted(_cabi.SIGNATURES) == _declared_symbols()


def test_host_only_entry_points():
    from codetr import _cabi
    import torch

    lib = _cabi.load()
    assert lib.codetr_hip_abi_version() == _cabi.ABI_VERSION
    assert _cabi.strerror(0) == "success"
    assert "im2col_step" in _cabi.strerror(-2)
    # the model shape (M=8, D=32, L=5, P=4) must take the tiled kernel in fp16/bf16/fp32
    assert _cabi.msda_variant(torch.float16, 8, 32, 5, 4) == "tiled_x4"
    assert _cabi.msda_variant(torch.bfloat16, 8, 32, 5, 4) == "tiled_x4"
    assert _cabi.msda_variant(torch.float32, 8, 32, 5, 4) == "tiled_x8"
    assert _cabi.msda_variant(torch.float64, 8, 32, 5, 4) == "scalar"
    assert _cabi.msda_variant(torch.float16, 2, 2, 2, 2) == "scalar"  # reference tiny case D=2


def test_round5_host_only_tables_and_contract_errors():
    """codetr_ffn_oproj_w1_index: the column order of W1 for the FFN launch that starts with the attention output projection
    -- the epilogue's lane layout (a lane of group g owns channels 32 s + 16 (g & 1) + 8 (g >> 1) .. + 7) written as MFMA
    k-slots (32 s + 8 g ..): groups 1 and 2 exchanged inside every 32-channel block; argument errors of the two fused
    encoder entry points are host-side."""
    from codetr import _cabi

    idx = _cabi.ffn_oproj_w1_index(256)
    assert sorted(idx) == list(range(256))
    for s in range(8):
        blk = idx[32 * s:32 * s + 32]
        assert blk[0:8] == list(range(32 * s, 32 * s + 8)) and blk[8:16] == list(range(32 * s + 16, 32 * s + 24))
        assert blk[16:24] == list(range(32 * s + 8, 32 * s + 16)) and blk[24:32] == list(range(32 * s + 24, 32 * s + 32))
    lib = _cabi.load()
    null, one = ctypes.c_void_p(0), ctypes.c_void_p(16)
    bad = ctypes.c_void_p(24)   # not 16-byte aligned
    assert lib.codetr_ffn_oproj_w1_index(192, ctypes.cast((ctypes.c_int32 * 192)(), ctypes.c_void_p)) == -1
    # encoder projections: null operand, misaligned operand -> BADARG; K != 256 / too few rows / value width not 64 n -> UNSUPPORTED
    f = lib.codetr_encoder_projections_f16
    assert f(null, null, one, one, one, null, one, one, 40000, 256, 512, 256, 0, 0) == -1
    assert f(null, one, bad, one, one, null, one, one, 40000, 256, 512, 256, 0, 0) == -1
    assert f(null, one, one, one, one, null, one, one, 40000, 256, 512, 192, 0, 0) == _cabi.E_UNSUPPORTED
    assert f(null, one, one, one, one, null, one, one, 1000, 256, 512, 256, 0, 0) == _cabi.E_UNSUPPORTED
    assert f(null, one, one, one, one, null, one, one, 40000, 224, 512, 256, 0, 0) == _cabi.E_UNSUPPORTED
    g = lib.codetr_ffn_oproj_relu_ln2_f16
    assert g(null, one, null, one, one, one, one, one, one, one, 1000, 256, 2048, null, null, 0.0, null, null, 0.0, null, null) == -1
    assert g(null, one, one, one, one, one, one, one, one, one, 1000, 192, 2048, null, null, 0.0, null, null, 0.0, null, null) == _cabi.E_UNSUPPORTED


def test_argument_validation_without_gpu():
    """Contract errors are detected on the host before any launch, so they are testable here."""
    from codetr import _cabi

    lib = _cabi.load()
    null = ctypes.c_void_p(0)
    one = ctypes.c_void_p(16)  # never dereferenced: validation fails first
    assert lib.codetr_msda_forward_f16(null, null, one, one, one, one, 1, 1, 1, 8, 1, 1, 1, 64, one) == -1
    assert lib.codetr_msda_forward_f16(null, one, one, one, one, one, 0, 1, 1, 8, 1, 1, 1, 64, one) == -1
    assert lib.codetr_msda_forward_f32(null, one, one, one, one, one, 3, 1, 1, 8, 1, 1, 1, 2, one) == -2


def test_op_is_registered_with_reference_schema():
    import torch
    import codetr  # noqa: F401

    s = str(torch.ops.codetr.multi_scale_deformable_attention.default._schema)
    assert s == ("codetr::multi_scale_deformable_attention(Tensor value, Tensor spatial_shapes, Tensor "
                 "level_start_index, Tensor sampling_loc, Tensor attn_weight, int im2col_step) -> Tensor")


def test_fake_kernel_shapes_and_checks():
    import torch
    import codetr  # noqa: F401

    v = torch.empty(2, 21, 4, 16, device="meta", dtype=torch.float16)
    ss = torch.empty(2, 2, device="meta", dtype=torch.int64)
    ls = torch.empty(2, device="meta", dtype=torch.int64)
    loc = torch.empty(2, 7, 4, 2, 3, 2, device="meta", dtype=torch.float16)
    w = torch.empty(2, 7, 4, 2, 3, device="meta", dtype=torch.float16)
    out = torch.ops.codetr.multi_scale_deformable_attention(v, ss, ls, loc, w, 64)
    assert out.shape == (2, 7, 64) and out.dtype == torch.float16
    with pytest.raises(RuntimeError):
        torch.ops.codetr.multi_scale_deformable_attention(v, ss, ls, loc.float(), w, 64)


def test_cpu_tensors_are_rejected_not_emulated():
    """The product has no CPU formulation of the op: CPU tensors fail in the dispatcher."""
    import torch
    import codetr  # noqa: F401

    v = torch.zeros(1, 4, 1, 8)
    ss = torch.tensor([[2, 2]])
    ls = torch.tensor([0])
    loc = torch.zeros(1, 1, 1, 1, 1, 2)
    w = torch.zeros(1, 1, 1, 1, 1)
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.codetr.multi_scale_deformable_attention(v, ss, ls, loc, w, 64)


def test_runner_dispatch_table_is_in_sync_with_the_signature_table():
    """runner/dispatch_gen.inc (typed trampolines of the C++ plan runner) is generated from codetr/_cabi.py SIGNATURES"""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gen = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_runner_dispatch.py")], stdout=subprocess.PIPE,
                         text=True, check=True).stdout
    assert gen == open(os.path.join(root, "runner", "dispatch_gen.inc")).read(), \
        "regenerate: python tools/gen_runner_dispatch.py > runner/dispatch_gen.inc"
    assert os.access(os.path.join(root, "runner", "codetr_runner"), os.X_OK)   # built by __graft_entry__.build()


def test_kernel_name_whitelist_matches_sources():
    """codetr/_kernel_names.py (the plan exporter's whitelist of replayable device activity) lists exactly the
    __global__ functions of csrc/*.hip: regenerate with tools/gen_kernel_names.py after adding a kernel"""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_kernel_names", os.path.join(root, "tools", "gen_kernel_names.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    from codetr._kernel_names import KERNEL_NAMES
    from codetr.export import is_own_kernel

    assert list(KERNEL_NAMES) == gen.kernel_names()
    assert is_own_kernel("(anonymous namespace)::add_kernel(unsigned short const*, ...)")
    assert is_own_kernel("void (anonymous namespace)::linear_256_kernel<(anonymous namespace)::HalfT, 0, true, false, true>(")
    assert is_own_kernel("_ZN12_GLOBAL__N_114row_max_kernelEPKDF16_PDF16_li")
    for foreign in ("void (anonymous namespace)::elementwise_kernel_with_index<int, ...>", "void at::native::vectorized_elementwise_kernel<4",
                    "Cijk_Alik_Bljk_HHS_BH", "Memcpy DtoD (Device -> Device)", "__amd_rocclr_copyBuffer", "triton_poi_fused_add_0"):
        assert not is_own_kernel(foreign), foreign


def test_recorder_keeps_only_accepted_launches():
    """hip_ops probes the three-pass encoder kernel and retries another form when the library answers
    E_UNSUPPORTED; the rejected probe enqueued nothing and must not reach an exported launch plan (the in-process replay and
    the C++ runner would fail on it).  A rejected call returns before any HIP call, so this runs without a GPU."""
    from codetr import _cabi

    _cabi.RECORDER = []
    try:
        lib = _cabi.load()
        one = ctypes.c_void_p(16)
        assert lib.codetr_linear_sk_f16(None, one, one, None, None, one, 1000, 256, 192, 3, None, 0, 0) == -4   # act = 3
        assert lib.codetr_linear_sk_f16(None, None, one, None, None, one, 1000, 256, 192, 0, None, 0, 0) == -1  # null x
        assert _cabi.RECORDER == []
        # pure host queries are never recorded either
        assert lib.codetr_linear_sk_supported(1000, 256, 192) == 1
        assert _cabi.RECORDER == []
    finally:
        _cabi.RECORDER = None


def test_product_library_reads_no_environment_variable():
    """an exported plan or a deployment must not change kernels with the environment -- the A/B
    switches live in diagnostic builds (tools/micro/diag_env.h) only"""
    import glob

    for path in glob.glob(os.path.join(ROOT, "co-detr-tensorrt_amd", "csrc", "*.hip")) + \
            glob.glob(os.path.join(ROOT, "co-detr-tensorrt_amd", "csrc", "*.h")):
        assert "getenv" not in open(path).read(), path


def test_host_package_reads_no_route_switch_from_the_environment():
    """`build_CoDETR` in a stray environment must run the kernel set the headline test covers -- the
    host package's route switches are plain module attributes (hip_ops.nondefault_switches lists the ones that are off
    their default; tools/ab_host_routes.py patches them).  The one environment read left is the checkpoint loader's
    explicit pickle opt-in."""
    import glob
    import re

    hits = []
    for path in glob.glob(os.path.join(ROOT, "co-detr-tensorrt_amd", "codetr", "*.py")):
        for n, line in enumerate(open(path), 1):
            if re.search(r"os\.environ|getenv\(", line):
                hits.append((os.path.basename(path), n, line.strip()))
    assert len(hits) == 1 and hits[0][0] == "checkpoint.py" and "CODETR_ALLOW_PICKLE" in hits[0][2], hits
    from codetr import hip_ops, transformer

    assert hip_ops.nondefault_switches() == []
    transformer.DEC_FUSED = False
    hip_ops.MSDA_V4_THREADS = 256
    try:
        assert hip_ops.nondefault_switches() == ["DEC_FUSED", "MSDA_V4_THREADS"]
    finally:
        transformer.DEC_FUSED, hip_ops.MSDA_V4_THREADS = True, 512
