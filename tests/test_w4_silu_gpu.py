"""SiLU*mul in the int4 GEMM epilogue.

The reference MLP runs the merged gate_up column-parallel linear
(layers/linear/multi_parallel_linear.cpp:14-41) and then kernel::act_and_mul
(kernels/activation_kernels.cu:84) as a separate launch over the [T, 2d] intermediate.  Here the
merged weight is packed with its gate / up halves interleaved by 32-column tile (SLM_W4_PAIRED) and
slm_w4a16_gemm(SLM_W4_SILU_MUL) writes silu(gate) * up directly.

Checked: the paired pack is exactly the documented column permutation; the fused result is
BIT-IDENTICAL to the unfused sequence (same launch plan) for every kernel the plan can pick --
GEMV, small-M (in-kernel pair exchange and split-K reduce), the general kernel (NTW = 1 / 2), the
wave-specialised 256 x 128 and the 256 x 256 kernels -- with and without bias, both dtypes; and it
agrees with the CPU oracle (oracle.gemm_f32 + oracle.silu_mul) within the GEMM tolerance.
"""
import numpy as np
import pytest
import torch

from oracle import oracle
from tests import helpers

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _dt(bits):
    return torch.bfloat16 if bits == "bf16" else torch.float16


def _src_cols(N):
    n = np.arange(N)
    return (n >> 6) * 32 + (n & 31) + np.where(n & 32, N // 2, 0)


@pytest.mark.parametrize("fmt,gs,act", [("awq", 128, False), ("awq", 32, False), ("gptq", 64, False),
                                        ("gptq", 128, True), ("gptq", -1, False)])
@pytest.mark.parametrize("bits", ["bf16", "f16"])
def test_paired_prepack_is_the_documented_column_permutation(fmt, gs, act, bits):
    from scalellm_amd import kernels
    K, N = 256, 384
    case = helpers.make_quant_case(11, K, N, gs, fmt, bits, act_order=act)
    plain = kernels.w4_dequant(helpers.pack_case(case, bits))
    paired = kernels.w4_dequant(helpers.pack_case(case, bits, paired=True))
    src = torch.from_numpy(_src_cols(N)).to(DEV)
    assert torch.equal(paired, plain[:, src])


# (M, K, N, group, knobs): every kernel / reduce path the plan can take
PLANS = [
    (1, 1024, 512, 128, dict(SLM_W4_GEMV_KS=4)),                 # dot2 GEMV, 2 tiles x 4 K slices
    (1, 4096, 1024, 128, dict(SLM_W4_GEMV_KS=2)),                # 4 tiles x 2 K slices
    (3, 512, 256, 32, dict(SLM_W4_GEMV=2, SLM_W4_GEMV_KS=4)),    # GEMV with MT = 4, group 32
    (8, 1024, 512, 128, dict(SLM_W4_SPLITK=1)),                  # small-M kernel, pair exchange in LDS
    (32, 1024, 512, 64, dict(SLM_W4_SPLITK=1)),
    (17, 2048, 256, 128, dict()),                                # small-M kernel, split-K -> fused reduce
    (32, 4096, 1024, 128, dict(SLM_W4_SPLITK=4)),
    (24, 1024, 512, 128, dict(SLM_W4_SMALL=0, SLM_W4_SPLITK=1)),  # general kernel MT = 1 (POST form)
    (24, 1024, 512, 128, dict(SLM_W4_SMALL=0, SLM_W4_SPLITK=1, SLM_W4_NTW=2)),  # pair inside one wave
    (48, 1024, 512, 128, dict(SLM_W4_SPLITK=1)),                 # general kernel, MT = 2
    (48, 1024, 512, 128, dict(SLM_W4_KS_MT2=1)),                 # K-sliced stream, two row tiles: in-kernel pair
    (64, 4096, 1024, 128, dict(SLM_W4_KS_MT2=1)),                # ... split over 4 workgroups: fused reduce
    (33, 2048, 448, 128, dict(SLM_W4_KS_MT2=1, SLM_W4_KS_TPW=4)),  # ... ragged tile runs, one row in tile 2
    (64, 1024, 512, 32, dict(SLM_W4_SPLITK=2)),
    (100, 1024, 512, 128, dict(SLM_W4_MT=4, SLM_W4_SPLITK=1)),   # MT = 4 (PRE form)
    (128, 2048, 1024, 128, dict(SLM_W4_MT=4, SLM_W4_SPLITK=2)),
    (256, 1024, 1024, 128, dict(SLM_W4_MT=8, SLM_W4_SPLITK=1)),  # wave-specialised 256 x 128
    (300, 1024, 1152, 64, dict(SLM_W4_MT=8, SLM_W4_SPLITK=1)),   # ragged M, N = 9 tiles of 128
    (256, 2048, 1024, 128, dict(SLM_W4_MT=8, SLM_W4_SPLITK=2)),
    (256, 1024, 1024, 128, dict(SLM_W4_MT=16, SLM_W4_SPLITK=1)),  # symmetric 256 x 256
    (384, 1024, 1280, 128, dict(SLM_W4_MT=16, SLM_W4_SPLITK=1)),
    (256, 4096, 2048, 128, dict()),                              # whatever the plan picks
    (128, 2048, 1024, 128, dict(SLM_W4_M128=1)),                 # round 5: w4_m128.hip (65 <= M <= 128), the plan's split
    (100, 1024, 512, 32, dict(SLM_W4_M128=1, SLM_W4_SPLITK=1)),  # ... in-kernel pair exchange, two scale groups per chunk
    (96, 4096, 1024, 64, dict(SLM_W4_M128=1, SLM_W4_SPLITK=4)),  # ... fp32 slabs + the fused reduce
    (128, 1024, 512, 128, dict(SLM_W4_M128=1, SLM_W4_M128_WD=4, SLM_W4_SPLITK=1)),  # ... four-chunk weight ring
    (65, 1024, 448, 128, dict(SLM_W4_M128=1, SLM_W4_SPLITK=1)),  # ... N = 7 tile pairs: a clamped wave pair
    (128, 8192, 1024, 128, dict()),                              # ... K >= 8192: the plan's own choice
    (128, 2048, 1024, 128, dict(SLM_W4_M128=1, SLM_W4_M128_KW=2, SLM_W4_SPLITK=1)),  # ... two waves per column tile
    (100, 1024, 448, 32, dict(SLM_W4_M128=1, SLM_W4_M128_KW=2, SLM_W4_SPLITK=2)),    # ... + slabs, group 32, clamped pair
    (96, 1024, 512, 64, dict(SLM_W4_M128=1, SLM_W4_M128_KW=1, SLM_W4_SPLITK=1)),
    (128, 2048, 1024, 128, dict(SLM_W4_M128=1, SLM_W4_M128_CT=8, SLM_W4_SPLITK=1)),  # ... 256-column workgroups: four (gate, up) wave pairs
    (100, 1024, 448, 32, dict(SLM_W4_M128=1, SLM_W4_M128_CT=8, SLM_W4_SPLITK=2)),    # ... + slabs, clamped pairs
    (65, 1024, 1216, 128, dict(SLM_W4_M128=1, SLM_W4_M128_CT=8, SLM_W4_M128_WD=4)),  # ... N = 19 tile pairs
    (128, 2048, 1024, 128, dict(SLM_W4_M128=1, SLM_W4_M128_CT=8, SLM_W4_M128_ADMA=1, SLM_W4_SPLITK=1)),  # ... activations by LDS-DMA
    (100, 1024, 448, 32, dict(SLM_W4_M128=1, SLM_W4_M128_CT=8, SLM_W4_M128_ADMA=1, SLM_W4_SPLITK=2)),
]


@pytest.mark.parametrize("bits", ["bf16", "f16"])
@pytest.mark.parametrize("bias", [False, True])
@pytest.mark.parametrize("M,K,N,gs,knobs", PLANS)
def test_fused_silu_mul_is_bit_identical_to_gemm_then_silu(M, K, N, gs, knobs, bias, bits, tune):
    from scalellm_amd import kernels
    tune(**knobs)
    dt = _dt(bits)
    fmt = "awq" if (M + K) % 3 else "gptq"
    case = helpers.make_quant_case(M * 7 + N, K, N, gs, fmt, bits)
    g = torch.Generator(device=DEV).manual_seed(M + K + N)
    a = torch.randn(M, K, device=DEV, dtype=dt, generator=g)
    b = (torch.randn(N, device=DEV, dtype=dt, generator=g) * 0.5) if bias else None
    plain, paired = helpers.pack_case(case, bits), helpers.pack_case(case, bits, paired=True)
    full = torch.full((M, N), float("nan"), device=DEV, dtype=dt)
    kernels.gptq_gemm(a, plain, full, b)
    want = torch.empty(M, N // 2, device=DEV, dtype=dt)
    kernels.silu_and_mul(want, full)
    b_packed = b[torch.from_numpy(_src_cols(N)).to(DEV)].contiguous() if bias else None
    got = torch.full((M, N // 2), float("nan"), device=DEV, dtype=dt)
    kernels.gptq_gemm(a, paired, got, b_packed, silu_mul=True)
    torch.cuda.synchronize()
    assert not torch.isnan(got.float()).any()
    assert torch.equal(got, want), f"max |diff| {(got.float() - want.float()).abs().max().item()}"
    # and against the CPU oracle (fp32 GEMM, fp32 silu * up), GEMM tolerance of marlin_gemm_test.py:104-107
    w = (oracle.awq_dequant if fmt == "awq" else oracle.gptq_dequant)(
        case["qweight"], case["qzeros"], case["scales"], case["group_size"])
    ref = oracle.gemm_f32(a.float().cpu().numpy(), w)
    if bias:
        ref = ref + b.float().cpu().numpy()[None, :]
    ref = oracle.silu_mul(ref)
    err = float(np.abs(got.float().cpu().numpy() - ref).mean() / np.abs(ref).mean())
    assert err < (2e-2 if bits == "bf16" else 2e-3), err


def test_fused_silu_mul_strided_output_rows_and_llama_shape():
    """gate_up of Llama-3-8B at decode batch sizes, output written into a wider buffer (ldc > N/2)."""
    from scalellm_amd import kernels
    K, N = 4096, 28672
    case = helpers.make_quant_case(5, K, N, 128, "awq", "bf16")
    plain, paired = helpers.pack_case(case), helpers.pack_case(case, paired=True)
    for M in (1, 8, 32, 256):
        a = torch.randn(M, K, device=DEV, dtype=torch.bfloat16)
        full = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        with kernels.tuning(**({"SLM_W4_GEMV_KS": 4} if M == 1 else {})):
            kernels.gptq_gemm(a, plain, full)
            want = torch.empty(M, N // 2, device=DEV, dtype=torch.bfloat16)
            kernels.silu_and_mul(want, full)
            wide = torch.zeros(M, N // 2 + 64, device=DEV, dtype=torch.bfloat16)
            kernels.gptq_gemm(a, paired, wide[:, :N // 2], silu_mul=True)
        assert torch.equal(wide[:, :N // 2], want), M
        assert not wide[:, N // 2:].any()


def test_fused_silu_mul_argument_checks():
    from scalellm_amd import kernels
    case = helpers.make_quant_case(3, 256, 128, 128, "gptq", "bf16")
    plain, paired = helpers.pack_case(case), helpers.pack_case(case, paired=True)
    a = torch.randn(4, 256, device=DEV, dtype=torch.bfloat16)
    half = torch.empty(4, 64, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(kernels.SlmError, match="paired"):
        kernels.gptq_gemm(a, plain, half, silu_mul=True)
    with pytest.raises(kernels.SlmError, match="shape"):
        kernels.gptq_gemm(a, paired, torch.empty(4, 128, device=DEV, dtype=torch.bfloat16), silu_mul=True)
    with pytest.raises(kernels.SlmError, match="defer_reduce"):
        kernels.gptq_gemm(a, paired, half, silu_mul=True, defer_reduce=True)
    odd = helpers.make_quant_case(3, 256, 96, 128, "gptq", "bf16")  # N % 64 != 0
    with pytest.raises(kernels.SlmError, match="N % 64"):
        helpers.pack_case(odd, paired=True)


@pytest.mark.parametrize("quant", ["awq", "gptq"])
def test_column_parallel_qlinear_act_mul_matches_linear_then_silu(quant):
    from scalellm_amd import kernels
    from scalellm_amd.layers import ColumnParallelQLinear, ParallelArgs, QuantArgs
    K, N = 512, 768
    case = helpers.make_quant_case(9, K, N, 128, quant, "bf16")
    sd = dict(qweight=torch.from_numpy(case["qweight"]), qzeros=torch.from_numpy(case["qzeros"]),
              scales=torch.from_numpy(case["scales_bits"].view(np.int16)).view(torch.bfloat16),
              bias=torch.randn(N).to(torch.bfloat16))
    qa, pa = QuantArgs(quant_method=quant, bits=4, group_size=128), ParallelArgs()
    plain = ColumnParallelQLinear(K, N, True, qa, False, pa, torch.bfloat16, DEV)
    fused = ColumnParallelQLinear(K, N, True, qa, False, pa, torch.bfloat16, DEV, act_mul="silu")
    for m in (plain, fused):
        m.load_state_dict(sd)
        m.verify_loaded_weights()
    x = torch.randn(40, K, device=DEV, dtype=torch.bfloat16)
    want = torch.empty(40, N // 2, device=DEV, dtype=torch.bfloat16)
    kernels.silu_and_mul(want, plain.forward(x))
    assert torch.equal(fused.forward(x), want)
