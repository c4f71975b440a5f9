"""8-bit weights (num_bits = 8 of marlin::gptq_gemm / gptq_repack / awq_repack, bits = 8 of the
quantised linears: qlinear_awq_marlin_impl.cpp:25-26, CPU semantics qlinear_impl.cpp:21-100) through
the C ABI: the checkpoint is rewritten into two int4 planes over 2K packed rows
(csrc/w8_planes.hip, include/slm_hip.h section 3b) and runs on the int4 GEMM kernels.

Checks: the plane decomposition is EXACT on the integers (both planes dequantised separately and
summed in fp32 reproduce s (q - z) bit for bit where the products are exact), the GEMM against the
oracle's construct_weights restatement + fp32 GEMM over the reference's marlin test axes
(marlin_gemm_test.py:47-107, which parametrises num_bits over [4, 8]) at the reference's tolerance,
the committed 8-bit goldens produced by the reference's own quant_utils helpers, every kernel
regime (GEMV, K-sliced, general POST / PRE, wave-specialised), bias, the fused SiLU*mul pairing."""
import numpy as np
import pytest
import torch

from oracle import oracle
from tests import helpers

pytestmark = pytest.mark.gpu
DEV = "cuda"
GEMM_TOL = {"f16": 1e-3, "bf16": 8e-3}  # marlin_gemm_test.py:104-107; bf16 = 8x (8 fewer mantissa bits)
Q8 = helpers.load_npz_groups("quant8_cases.npz")


def _tdtype(bits):
    return torch.bfloat16 if bits == "bf16" else torch.float16


def _oracle_w(case):
    if case["fmt"] == "awq":
        return oracle.awq_dequant(case["qweight"], case["qzeros"], case["scales"], case["group_size"], bits=8)
    return oracle.gptq_dequant(case["qweight"], case["qzeros"], case["scales"], case["group_size"],
                               case["g_idx"], bits=8)


def _rel_err(c, ref):
    return float(np.abs(c - ref).mean() / np.abs(ref).mean())


@pytest.mark.parametrize("bits", ["bf16", "f16"])
@pytest.mark.parametrize("fmt,gs,act,sym", [("gptq", 128, False, False), ("gptq", 64, False, False),
                                            ("gptq", -1, False, True), ("gptq", 64, True, False),
                                            ("gptq", 32, False, True), ("awq", 128, False, False),
                                            ("awq", 64, False, False)])
def test_plane_decomposition_is_exact(bits, fmt, gs, act, sym):
    """dequant(high plane) + dequant(low plane) == s (q - z): each plane's value (<= 5 + 11 bits)
    is rounded to T once by w4_dequant, so compare against the same two roundings of the integer truth."""
    from scalellm_amd import kernels
    case = helpers.make_quant8_case(7, 256, 96, gs, fmt, bits, act_order=act, sym=sym)
    np.testing.assert_array_equal(_oracle_w(case), helpers.dense_weight8(case))  # oracle == integer truth
    packed = helpers.pack_case8(case, bits)
    K = case["K"]
    assert packed.K == 2 * K and packed.k_src == K and packed.perm.numel() == 2 * K
    w = kernels.w4_dequant(packed).float().cpu().numpy()  # [2K, N]: plane rows in packed order
    perm2 = packed.perm.cpu().numpy()
    assert np.array_equal(perm2[:K], perm2[K:])  # both planes gather the same activation columns
    gi = case["g_idx"] if case["g_idx"] is not None else np.arange(K) // case["group_size"]
    rows = perm2[:K]
    z, s, q = case["z_eff"][gi[rows]], case["scales"][gi[rows]], case["q"][rows]
    rnd = (lambda x: helpers.bf16_bits_to_f32(helpers.f32_to_bf16_bits(x))) if bits == "bf16" else \
        (lambda x: x.astype(np.float16).astype(np.float32))
    hi = rnd((16.0 * s) * ((q >> 4) - (z >> 4)).astype(np.float32))
    lo = rnd(s * ((q & 15) - (z & 15)).astype(np.float32))
    np.testing.assert_array_equal(w[:K], hi)
    np.testing.assert_array_equal(w[K:], lo)
    # and unrounded the two planes ARE the weight: 16 (qh - zh) + (ql - zl) == q - z on the integers
    assert np.array_equal(16 * ((q >> 4) - (z >> 4)) + ((q & 15) - (z & 15)), q - z)


def _run_gemm(case, bits, M, bias=False, seed=0):
    from scalellm_amd import kernels
    dt = _tdtype(bits)
    g = torch.Generator(device=DEV).manual_seed(seed)
    a = torch.randn(M, case["K"], device=DEV, dtype=dt, generator=g)
    b = torch.randn(case["N"], device=DEV, dtype=dt, generator=g) if bias else None
    packed = helpers.pack_case8(case, bits)
    c = torch.full((M, case["N"]), float("nan"), device=DEV, dtype=dt)
    kernels.gptq_gemm(a, packed, c, b)
    torch.cuda.synchronize()
    ref = oracle.gemm_f32(a.float().cpu().numpy(), _oracle_w(case))
    if bias:
        ref = ref + b.float().cpu().numpy()[None, :]
    out = c.float().cpu().numpy()
    assert not np.isnan(out).any()
    return out, ref


@pytest.mark.parametrize("bits", ["f16", "bf16"])
def test_reference_marlin_grid_8bit(bits):
    # marlin_gemm_test.py:47-56 axes with num_bits = 8 (+ ragged m, zero points, both formats)
    i = 0
    for M in (1, 16, 32, 33, 64, 100, 256):
        for N, K in ((64, 128), (128, 256), (256, 128), (512, 256)):
            for gs in (-1, 32, 64, 128):
                for fmt, act, sym in (("gptq", False, True), ("gptq", True, False), ("gptq", False, False),
                                      ("awq", False, False)):
                    i += 1
                    if i % 3 != (M % 3):
                        continue
                    if act and (gs == -1 or gs == K):
                        continue
                    case = helpers.make_quant8_case(300 + i, K, N, gs, fmt, bits, act_order=act, sym=sym)
                    out, ref = _run_gemm(case, bits, M, bias=(i % 2 == 0), seed=i)
                    err = _rel_err(out, ref)
                    assert err < GEMM_TOL[bits], (M, N, K, gs, fmt, act, sym, err)


@pytest.mark.parametrize("name", sorted(Q8))
def test_golden_8bit_cases_from_the_reference_helpers(name):
    """tests/golden/quant8_cases.npz: tensors packed by the reference's quant_utils (num_bits = 8)."""
    from scalellm_amd import kernels
    c = Q8[name]
    gs = int(c["group_size"][0])
    qweight = torch.from_numpy(c["qweight"]).to(DEV)
    qzeros = torch.from_numpy(c["qzeros"]).to(DEV)
    scales = torch.from_numpy(c["scales"].view(np.int16)).to(DEV).view(torch.float16)
    if name.startswith("awq"):
        packed = kernels.awq_repack(qweight, qzeros, scales, gs, bits=8)
    else:
        g_idx = torch.from_numpy(c["g_idx"]).to(DEV) if int(c["act_order"][0]) else None
        packed = kernels.gptq_repack(qweight, qzeros, scales, gs, g_idx, bits=8)
    K, N = c["w"].shape
    a = torch.randn(24, K, device=DEV, dtype=torch.float16, generator=torch.Generator(device=DEV).manual_seed(5))
    out = torch.empty(24, N, device=DEV, dtype=torch.float16)
    kernels.gptq_gemm(a, packed, out)
    ref = a.float().cpu().numpy() @ c["w"]
    assert _rel_err(out.float().cpu().numpy(), ref) < GEMM_TOL["f16"]


@pytest.mark.parametrize("M", [1, 4, 32, 64, 128, 256])
@pytest.mark.parametrize("K,N", [(4096, 4096), (4096, 6144)])
def test_llama_layer_shapes_8bit_every_kernel_regime(M, K, N):
    """Full-size rows through every kernel the plan picks for 2K = 8192 packed rows (GEMV, K-sliced,
    general POST / PRE, wave-specialised), checked against a dense fp32 GEMM on the integer truth."""
    case = helpers.make_quant8_case(900 + M, K, N, 128, "awq", "bf16")
    from scalellm_amd import kernels
    packed = helpers.pack_case8(case, "bf16")
    g = torch.Generator(device=DEV).manual_seed(M)
    a = torch.randn(M, K, device=DEV, dtype=torch.bfloat16, generator=g)
    c = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    kernels.gptq_gemm(a, packed, c)
    w = torch.from_numpy(helpers.dense_weight8(case)).to(DEV)
    ref = (a.float() @ w).cpu().numpy()
    assert _rel_err(c.float().cpu().numpy(), ref) < GEMM_TOL["bf16"]


@pytest.mark.parametrize("K", [384, 1792])
@pytest.mark.parametrize("fmt", ["gptq", "awq"])
def test_per_channel_scales_with_k_not_a_power_of_two(K, fmt):
    """group_size = -1 (one scale row for all of K) with K = 3 x 128 / 14 x 128: the int4 GEMM wants
    power-of-two groups, so the packed table is written out at 128-row granularity
    (slm_w8_packed_group_size) -- 2K / 128 identical-per-plane rows."""
    from scalellm_amd import kernels
    case = helpers.make_quant8_case(17 + K, K, 128, -1, fmt, "bf16", sym=(fmt == "gptq"))
    packed = helpers.pack_case8(case, "bf16")
    assert packed.group_size == 128 and packed.K == 2 * K and packed.sz.numel() == (2 * K // 128) * 128
    for M in (3, 40, 200):
        a = torch.randn(M, K, device=DEV, dtype=torch.bfloat16, generator=torch.Generator(device=DEV).manual_seed(M))
        c = torch.empty(M, 128, device=DEV, dtype=torch.bfloat16)
        kernels.gptq_gemm(a, packed, c)
        ref = a.float().cpu().numpy() @ helpers.dense_weight8(case)
        assert _rel_err(c.float().cpu().numpy(), ref) < GEMM_TOL["bf16"], (K, fmt, M)


@pytest.mark.parametrize("M", [8, 32, 256])
def test_8bit_weights_with_a_deferred_splitk_reduce(M):
    """The plane form runs through the column gather (perm2) -- also under SLM_W4_DEFER_REDUCE: a split-K
    call leaves its fp32 slabs for slm_rms_norm_splitk; bit-identical to GEMM -> reduce -> rms_norm."""
    from scalellm_amd import kernels
    K, N = 4096, 4096
    case = helpers.make_quant8_case(33 + M, K, N, 128, "awq", "bf16")
    packed = helpers.pack_case8(case, "bf16")
    g = torch.Generator(device=DEV).manual_seed(M)
    a = torch.randn(M, K, device=DEV, dtype=torch.bfloat16, generator=g)
    w = (1 + 0.1 * torch.randn(N, device=DEV, generator=g)).to(torch.bfloat16)
    res0 = torch.randn(M, N, device=DEV, dtype=torch.bfloat16, generator=g)
    c = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    assert not kernels.gptq_gemm(a, packed, c)
    out_ref, res_ref = torch.empty_like(c), res0.clone()
    kernels.rms_norm(out_ref, c, w, 1e-5, res_ref)
    c2 = torch.full_like(c, float("nan"))
    h = kernels.gptq_gemm(a, packed, c2, defer_reduce=True)
    out, res = torch.empty_like(c), res0.clone()
    kernels.rms_norm(out, c2, w, 1e-5, res, partials=h)
    torch.cuda.synchronize()
    if not h:
        assert torch.equal(c2, c)
    assert torch.equal(out, out_ref) and torch.equal(res, res_ref)


def test_8bit_paired_gate_up_fuses_silu_mul():
    """SLM_W4_PAIRED composes with the plane form: silu(gate) * up in the GEMM epilogue is
    bit-identical to the unfused GEMM + slm_silu_mul."""
    from scalellm_amd import kernels
    K, N = 512, 256
    case = helpers.make_quant8_case(41, K, N, 128, "awq", "bf16")
    plain = helpers.pack_case8(case, "bf16")
    paired = helpers.pack_case8(case, "bf16", paired=True)
    a = torch.randn(8, K, device=DEV, dtype=torch.bfloat16, generator=torch.Generator(device=DEV).manual_seed(2))
    full = torch.empty(8, N, device=DEV, dtype=torch.bfloat16)
    kernels.gptq_gemm(a, plain, full)
    want = torch.empty(8, N // 2, device=DEV, dtype=torch.bfloat16)
    kernels.silu_and_mul(want, full)
    got = torch.empty(8, N // 2, device=DEV, dtype=torch.bfloat16)
    kernels.gptq_gemm(a, paired, got, silu_mul=True)
    assert torch.equal(got, want)


def test_8bit_argument_errors():
    from scalellm_amd import kernels
    from scalellm_amd._lib import SlmError
    qweight = torch.zeros(32, 64, dtype=torch.int32, device=DEV)      # K = 128
    scales = torch.ones(1, 64, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(SlmError):
        kernels.gptq_repack(qweight, None, scales, 128, bits=2)
    with pytest.raises(SlmError):  # qzeros of the 4-bit shape
        kernels.gptq_repack(qweight, torch.zeros(1, 8, dtype=torch.int32, device=DEV), scales, 128, bits=8)
