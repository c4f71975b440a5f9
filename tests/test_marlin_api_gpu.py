"""The reference's own int4 kernel tests, ported 1:1 onto this library's `_C.kernels`-shaped pybind
surface (scalellm_amd/csrc/shim/slm_shim_pybind.cpp <- scalellm/csrc/kernels.cu:24-54):

  tests/kernels/marlin_gemm_test.py:47-107     test_marlin_gemm    (same grid, same call, same metric)
  tests/kernels/marlin_repack_test.py:11-84    test_gptq_repack / test_awq_repack

`kernels.marlin_gemm / marlin_gptq_repack / marlin_awq_repack` are the reference's symbols
(marlin::gptq_gemm / gptq_repack / awq_repack, src/kernels/quantization/marlin.h:17-37) with the
reference's keyword arguments.  Two deliberate differences, both consequences of owning the packed
layout (the Marlin byte order is an NVIDIA mma.m16n8k16 artefact):
  * B is produced by kernels.marlin_gptq_repack (not by the python Marlin packer) and the scales
    are passed in plain column order (no permute_marlin_scales);
  * the repack tests cannot compare against the python Marlin packer; they check the repack
    BIT-EXACTLY through the GEMM instead: with A = identity and scales = 1 the GEMM returns
    (q - 8) for every weight, integers that fp16 holds exactly.
num_bits = 8 runs as two int4 planes (include/slm_hip.h section 3b): the full 4-bit grid plus the
reference's m x n x k x group x act_order axes at 8 bits (is_k_full / use_fp32_reduce do not select
different code here, so they are not multiplied into the 8-bit grid).
quantize_weights / sort_rows are torch ports of tests/kernels/quant_utils.py:22-98.
"""
import importlib.util
import os

import pytest
import torch

from tests import helpers

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def kernels():
    path = os.path.join(ROOT, "scalellm_amd", "csrc", "_slm_shim.so")
    assert os.path.exists(path), "build it: python -m scalellm_amd.build_shim"
    spec = importlib.util.spec_from_file_location("_slm_shim", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def quantize_weights(w, num_bits, group_size=-1, act_order=False, generator=None):
    """quant_utils.py:34-98: symmetric per-group quantisation, zero point 2^(bits-1)."""
    k, n = w.shape
    max_q = 2 ** num_bits - 1
    if group_size != -1:
        w = w.reshape((-1, group_size, n)).permute(1, 0, 2).reshape((group_size, -1))
    s = torch.max(torch.abs(w), dim=0, keepdim=True)[0]
    s *= 2 / max_q
    q_zero = (max_q + 1) // 2
    q_w = torch.clamp(torch.round(w / s).int() + q_zero, 0, max_q)
    w_ref = (q_w - q_zero).to(w.dtype) * s
    if group_size != -1:
        def reshape_w(x):
            return x.reshape((group_size, -1, n)).permute(1, 0, 2).reshape((k, n)).contiguous()
        q_w, w_ref = reshape_w(q_w), reshape_w(w_ref)
    s = s.reshape((-1, n)).contiguous()
    if act_order:
        gs = k if group_size == -1 else group_size
        g_idx = torch.arange(k, dtype=torch.int32, device=w.device) // gs
        perm = torch.randperm(k, generator=generator, device="cpu").to(w.device)
        w_ref, q_w, g_idx = w_ref[perm].contiguous(), q_w[perm].contiguous(), g_idx[perm].contiguous()
    else:
        g_idx = torch.empty(0, dtype=torch.int, device=w.device)
        perm = torch.empty(0, dtype=torch.int, device=w.device)
    return w_ref, q_w, s, g_idx, perm


def sort_rows(q_w, g_idx):
    """quant_utils.py:22-31."""
    perm = torch.argsort(g_idx).to(torch.int32)
    return q_w[perm.long()].contiguous(), g_idx[perm.long()].contiguous(), perm


def pack_gptq_weights(q_w, num_bits=4):
    import numpy as np
    f = helpers.pack_rows if num_bits == 4 else helpers.pack_rows8
    return torch.from_numpy(f(q_w.cpu().numpy().astype(np.int32))).to(q_w.device)


def pack_awq_weights(q_w, num_bits=4):
    import numpy as np
    f = helpers.pack_awq if num_bits == 4 else helpers.pack_awq8
    return torch.from_numpy(f(q_w.cpu().numpy().astype(np.int32))).to(q_w.device)


@pytest.mark.parametrize("m", [16, 32, 64])
@pytest.mark.parametrize("n", [64, 128, 256, 512])
@pytest.mark.parametrize("k", [128, 256])
@pytest.mark.parametrize("num_bits", [4])
@pytest.mark.parametrize("group_size", [-1, 32, 64, 128])
@pytest.mark.parametrize("act_order", [False, True])
@pytest.mark.parametrize("is_k_full", [False, True])
@pytest.mark.parametrize("use_fp32_reduce", [False, True])
def test_marlin_gemm(kernels, m, n, k, num_bits, group_size, act_order, is_k_full, use_fp32_reduce):
    # the reference's grid (marlin_gemm_test.py:47-56) has num_bits {4, 8}: the 8-bit half is
    # test_marlin_gemm_8bit below.  is_k_full = False on these full-K layers is the same computation
    # as True (every group is whole): the shim checks that from g_idx.
    _marlin_gemm_case(kernels, m, n, k, num_bits, group_size, act_order, is_k_full, use_fp32_reduce)


@pytest.mark.parametrize("m", [16, 32, 64])
@pytest.mark.parametrize("n", [64, 128, 256, 512])
@pytest.mark.parametrize("k", [128, 256])
@pytest.mark.parametrize("group_size", [-1, 32, 64, 128])
@pytest.mark.parametrize("act_order", [False, True])
def test_marlin_gemm_8bit(kernels, m, n, k, group_size, act_order):
    _marlin_gemm_case(kernels, m, n, k, 8, group_size, act_order, True, True)


@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
@pytest.mark.parametrize("num_bits", [4, 8])
@pytest.mark.parametrize("m", [16, 64, 256])
@pytest.mark.parametrize("k", [1024, 4096])
@pytest.mark.parametrize("group_size,act_order", [(-1, False), (128, False), (128, True)])
def test_marlin_gemm_deep_k_and_bf16(kernels, dtype, num_bits, m, k, group_size, act_order):
    """The reference's grid stops at k = 256 and fp16 (marlin_gemm_test.py:47-56); decode layers have
    k = 4096 ... 28672 and run in bf16: the same call and metric at depth, both dtypes (bf16 bound 8e-3:
    8 fewer mantissa bits), every M regime of the plan (GEMV-adjacent, small, general, wave-specialised)."""
    _marlin_gemm_case(kernels, m, 512, k, num_bits, group_size, act_order, True, True, dtype=dtype)


def _marlin_gemm_case(kernels, m, n, k, num_bits, group_size, act_order, is_k_full, use_fp32_reduce,
                      dtype=torch.half):
    if act_order and (group_size == -1 or group_size == k):
        pytest.skip("act_order=True requires group_size < k (marlin_gemm_test.py:64)")
    gen = torch.Generator(device="cuda").manual_seed(m * 7 + n * 3 + k + group_size)
    a = torch.randn((m, k), dtype=dtype, device="cuda", generator=gen)
    w = torch.randn((k, n), dtype=dtype, device="cuda", generator=gen)
    w_ref, q_w, s, g_idx, _ = quantize_weights(w, num_bits=num_bits, group_size=group_size,
                                               act_order=act_order,
                                               generator=torch.Generator().manual_seed(k + n))
    # checkpoint-format weights -> this library's layout (rows sorted by group when act_order)
    gptq_q_w = pack_gptq_weights(q_w, num_bits)
    if act_order:
        _, g_idx, perm = sort_rows(q_w, g_idx)
    else:
        perm = torch.empty(0, dtype=torch.int32, device="cuda")
    marlin_q_w = torch.empty(k // 16, n * 16 // (32 // num_bits), dtype=torch.int32, device="cuda")
    kernels.marlin_gptq_repack(q_weight=gptq_q_w, perm=perm, out=marlin_q_w, num_bits=num_bits)
    marlin_s = s                                           # plain order: no permute_marlin_scales
    marlin_zp = torch.empty(0, dtype=torch.int32, device="cuda")
    workspace = torch.zeros(n // 64 * 16, dtype=torch.int32, device="cuda")
    output = torch.empty((m, n), dtype=dtype, device="cuda")
    kernels.marlin_gemm(A=a, B=marlin_q_w, C=output, scales=marlin_s, zeros=marlin_zp, g_idx=g_idx,
                        perm=perm, workspace=workspace, num_bits=num_bits, is_k_full=is_k_full,
                        has_zp=False, use_fp32_reduce=use_fp32_reduce)
    torch.cuda.synchronize()
    # (deep k: reference product in fp32, so that the yardstick's own bf16 / fp16 rounding of a long sum
    # does not enter; at the reference's k <= 256 this is the same number to 4 digits)
    output_ref = torch.matmul(a.float(), w_ref.float())
    max_diff = torch.mean(torch.abs(output.float() - output_ref)) / torch.mean(torch.abs(output_ref))
    assert max_diff < (0.001 if dtype == torch.half else 0.008)


def test_marlin_other_bit_widths_are_refused(kernels):
    """The quantised linears take bits 4 and 8 (qlinear_awq_marlin_impl.cpp:25-26); anything else is
    refused loudly by every entry point."""
    k, n = 128, 64
    out = torch.empty(k // 16, n * 16 // 16, dtype=torch.int32, device="cuda")
    empty = torch.empty(0, dtype=torch.int32, device="cuda")
    with pytest.raises(RuntimeError, match="num_bits must be 4 or 8"):
        kernels.marlin_gptq_repack(q_weight=torch.zeros(k // 16, n, dtype=torch.int32, device="cuda"),
                                   perm=empty, out=out, num_bits=2)
    with pytest.raises(RuntimeError, match="num_bits must be 4 or 8"):
        kernels.marlin_awq_repack(q_weight=torch.zeros(k, n // 16, dtype=torch.int32, device="cuda"), out=out,
                                  num_bits=2)
    with pytest.raises(RuntimeError, match="num_bits must be 4 or 8"):
        kernels.marlin_gemm(A=torch.zeros(16, k, dtype=torch.half, device="cuda"), B=out,
                            C=torch.empty(16, n, dtype=torch.half, device="cuda"),
                            scales=torch.ones(1, n, dtype=torch.half, device="cuda"), zeros=empty, g_idx=empty,
                            perm=empty, workspace=empty, num_bits=2, is_k_full=True, has_zp=False,
                            use_fp32_reduce=True)


def test_marlin_gemm_refuses_an_uneven_act_order_shard_and_sweeps_its_table_cache(kernels):
    """is_k_full = False with a g_idx whose groups are NOT whole (a row-parallel act-order shard) cannot
    be expressed on weights packed at checkpoint size: refused, with a pointer to the layer class.
    And the fused {scale, zero} table cache behind gptq_gemm drops the tables of freed parameters."""
    m, k, n, gs = 16, 256, 64, 64
    gen = torch.Generator().manual_seed(1)
    q_w = torch.randint(0, 16, (k, n), generator=gen, dtype=torch.int32).cuda()
    packed = torch.empty(k // 16, n * 16 // 8, dtype=torch.int32, device="cuda")
    g_idx_sorted = (torch.arange(k) // gs).to(torch.int32)
    g_idx_sorted[gs - 1] = 1                                  # group 0 one row short, group 1 one too many
    perm = torch.arange(k, dtype=torch.int32).cuda()
    kernels.marlin_gptq_repack(q_weight=pack_gptq_weights(q_w), perm=perm, out=packed, num_bits=4)
    a = torch.randn(m, k, dtype=torch.half, device="cuda")
    c = torch.empty(m, n, dtype=torch.half, device="cuda")
    empty = torch.empty(0, dtype=torch.int32, device="cuda")
    with pytest.raises(RuntimeError, match="RowParallelQLinearHipImpl"):
        kernels.marlin_gemm(A=a, B=packed, C=c, scales=torch.ones(k // gs, n, dtype=torch.half, device="cuda"),
                            zeros=empty, g_idx=g_idx_sorted.cuda(), perm=perm, workspace=empty, num_bits=4,
                            is_k_full=False, has_zp=False, use_fp32_reduce=True)
    before = kernels.marlin_sz_cache_entries()
    for i in range(4):  # four generations of "reloaded" parameters: only the live one keeps its table
        scales = torch.full((k // gs, n), 1.0 + i, dtype=torch.half, device="cuda")
        kernels.marlin_gemm(A=a, B=packed, C=c, scales=scales, zeros=empty, g_idx=empty, perm=empty,
                            workspace=empty, num_bits=4, is_k_full=True, has_zp=False, use_fp32_reduce=True)
        torch.cuda.synchronize()
        del scales
    assert kernels.marlin_sz_cache_entries() <= before + 2


def _dequant_through_gemm(kernels, packed, k, n, perm, num_bits=4):
    """(q - 2^(bits-1)) for every weight, exactly: GEMM with A = identity, scales = 1, symmetric zero
    (8 bits: 16 (hi - 8) + lo, integers <= 127 in magnitude that fp16 holds and sums exactly)."""
    eye = torch.eye(k, dtype=torch.half, device="cuda")
    ones = torch.ones(1, n, dtype=torch.half, device="cuda")
    out = torch.empty(k, n, dtype=torch.half, device="cuda")
    empty = torch.empty(0, dtype=torch.int32, device="cuda")
    kernels.marlin_gemm(A=eye, B=packed, C=out, scales=ones, zeros=empty, g_idx=empty, perm=perm,
                        workspace=empty, num_bits=num_bits, is_k_full=True, has_zp=False, use_fp32_reduce=True)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("k", [128, 256])
@pytest.mark.parametrize("n", [64, 128, 256])
@pytest.mark.parametrize("group_size", [-1, 32, 64, 128])
@pytest.mark.parametrize("act_order", [False, True])
@pytest.mark.parametrize("num_bits", [4, 8])
def test_gptq_repack(kernels, k, n, group_size, act_order, num_bits):
    if act_order and group_size in (-1, k):
        return
    w = torch.randn((k, n), dtype=torch.half, device="cuda")
    _, q_w, _, g_idx, _ = quantize_weights(w, num_bits=num_bits, group_size=group_size, act_order=act_order,
                                           generator=torch.Generator().manual_seed(n))
    gptq_q_w = pack_gptq_weights(q_w, num_bits)
    if act_order:
        _, g_idx, perm = sort_rows(q_w, g_idx)
    else:
        perm = torch.empty(0, dtype=torch.int32, device="cuda")
    out = torch.empty(k // 16, n * 16 // (32 // num_bits), dtype=torch.int32, device="cuda")
    kernels.marlin_gptq_repack(q_weight=gptq_q_w, perm=perm, out=out, num_bits=num_bits)
    # identity x W in CHECKPOINT row order (the GEMM gathers A's columns by perm, so row i of the
    # result is checkpoint row i again)
    got = _dequant_through_gemm(kernels, out, k, n, perm, num_bits)
    assert torch.equal(got, (q_w - 2 ** (num_bits - 1)).to(torch.half))


@pytest.mark.parametrize("k", [128, 256])
@pytest.mark.parametrize("n", [64, 128, 256])
@pytest.mark.parametrize("group_size", [-1, 32, 64, 128])
@pytest.mark.parametrize("num_bits", [4, 8])
def test_awq_repack(kernels, k, n, group_size, num_bits):
    w = torch.randn((k, n), dtype=torch.half, device="cuda")
    _, q_w, _, _, _ = quantize_weights(w, num_bits=num_bits, group_size=group_size)
    awq_q_w = pack_awq_weights(q_w, num_bits)
    out = torch.empty(k // 16, n * 16 // (32 // num_bits), dtype=torch.int32, device="cuda")
    kernels.marlin_awq_repack(q_weight=awq_q_w, out=out, num_bits=num_bits)
    got = _dequant_through_gemm(kernels, out, k, n, torch.empty(0, dtype=torch.int32, device="cuda"), num_bits)
    assert torch.equal(got, (q_w - 2 ** (num_bits - 1)).to(torch.half))


def test_marlin_gemm_with_zero_points_awq_checkpoint(kernels):
    """has_zp = True (the reference left it as "TODO: test with zero point"): zeros are the AWQ
    checkpoint's qzeros as stored; also exercises the (scales, zeros) table cache against address
    reuse: fresh tensors of the same shapes, different contents, back to back."""
    from oracle import oracle
    import numpy as np
    for seed in range(4):
        case = helpers.make_quant_case(50 + seed, 256, 128, 64, "awq", "f16")
        qweight = torch.from_numpy(case["qweight"]).cuda()
        qzeros = torch.from_numpy(case["qzeros"]).cuda()
        scales = torch.from_numpy(case["scales_bits"].view(np.int16)).cuda().view(torch.half)
        b = torch.empty(256 // 16, 128 * 2, dtype=torch.int32, device="cuda")
        kernels.marlin_awq_repack(q_weight=qweight, out=b, num_bits=4)
        a = torch.randn(24, 256, dtype=torch.half, device="cuda")
        c = torch.empty(24, 128, dtype=torch.half, device="cuda")
        empty = torch.empty(0, dtype=torch.int32, device="cuda")
        kernels.marlin_gemm(A=a, B=b, C=c, scales=scales, zeros=qzeros, g_idx=empty, perm=empty,
                            workspace=empty, num_bits=4, is_k_full=True, has_zp=True, use_fp32_reduce=True)
        torch.cuda.synchronize()
        ref = oracle.gemm_f32(a.float().cpu().numpy(),
                              oracle.awq_dequant(case["qweight"], case["qzeros"], case["scales"], 64))
        err = np.abs(c.float().cpu().numpy() - ref).mean() / np.abs(ref).mean()
        assert err < 1e-3, (seed, err)
        del qweight, qzeros, scales, b, a, c


def test_silu_with_mul_signature(kernels):
    """llm::kernel::silu_with_mul(input) -> Tensor (activation_kernels.h:14): bit-identical to the
    in-place form the captured step uses."""
    x = torch.randn(7, 2 * 96, dtype=torch.bfloat16, device="cuda")
    out = kernels.silu_with_mul(x)
    out2 = torch.empty(7, 96, dtype=torch.bfloat16, device="cuda")
    kernels.silu_and_mul(out2, x)
    torch.cuda.synchronize()
    assert out.shape == (7, 96) and torch.equal(out, out2)
