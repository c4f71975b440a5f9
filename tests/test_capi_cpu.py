"""CPU tests of the C-ABI boundary: the library loads without a GPU, exports every symbol that
include/slm_hip.h declares, validates arguments before touching the device, and its host-side
planning entry points (workspace sizes, split heuristics) behave."""
import ctypes as C
import os
import re
import subprocess

import pytest
import torch

from scalellm_amd import _lib
from scalellm_amd._lib import AttnArgs, W4GemmArgs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "slm_hip.h")


def _declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"SLM_API\s+[\w\s\*]+?\b(slm_\w+)\s*\(", src)))


def test_header_declares_the_expected_entry_points():
    syms = _declared_symbols()
    for must in ("slm_paged_kv_varlen_mha", "slm_set_kv_cache", "slm_w4_prepack", "slm_w4a16_gemm",
                 "slm_rms_norm", "slm_rope_kv_append", "slm_silu_mul", "slm_layer_norm", "slm_gelu", "slm_decode_advance",
                 "slm_allreduce", "slm_allreduce_simulate", "slm_shm_alloc", "slm_shm_export",
                 "slm_shm_import", "slm_ar_signal_bytes"):
        assert must in syms
    assert len(syms) >= 26


def test_library_exports_every_declared_symbol():
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = set(re.findall(r"\sT\s+(slm_\w+)", out))
    missing = [s for s in _declared_symbols() if s not in exported]
    assert not missing, f"declared in slm_hip.h but not exported: {missing}"
    # nothing else leaks out of the library (fvisibility=hidden)
    assert exported == set(_declared_symbols())


def test_library_loads_without_gpu_and_reports_version():
    L = _lib.lib()
    assert b"gfx950" in L.slm_version()
    assert L.slm_status_string(0) == b"ok"
    assert L.slm_status_string(-3) == b"workspace missing or too small"


def test_tuning_table_is_explicit_and_ignores_later_environment_changes():
    """kernel choice must not silently follow ambient SLM_* variables.  The
    environment is parsed once (first use); afterwards only slm_tuning_set / _clear change a knob,
    and the product sources contain exactly one getenv call (that one-time parse)."""
    L = _lib.lib()
    assert L.slm_tuning_clear(None) == 0
    one = _attn_args_for(1, 1)
    auto = L.slm_paged_kv_varlen_mha_auto_splits(C.byref(one))
    os.environ["SLM_ATTN_SPLITS"] = "3"        # too late: must have no effect
    try:
        assert L.slm_paged_kv_varlen_mha_auto_splits(C.byref(one)) == auto
    finally:
        del os.environ["SLM_ATTN_SPLITS"]
    assert L.slm_tuning_set(b"SLM_ATTN_SPLITS", 3) == 0
    v, s = C.c_int32(-1), C.c_int32(-1)
    assert L.slm_tuning_get(b"SLM_ATTN_SPLITS", C.byref(v), C.byref(s)) == 0 and (v.value, s.value) == (3, 1)
    assert L.slm_paged_kv_varlen_mha_auto_splits(C.byref(one)) == 3
    assert L.slm_tuning_clear(b"SLM_ATTN_SPLITS") == 0
    assert L.slm_tuning_get(b"SLM_ATTN_SPLITS", C.byref(v), C.byref(s)) == 0 and s.value == 0
    assert L.slm_paged_kv_varlen_mha_auto_splits(C.byref(one)) == auto
    assert L.slm_tuning_set(b"SLM_NO_SUCH_KNOB", 1) == -1
    assert L.slm_tuning_set(None, 1) == -1
    # the python context manager restores what it found
    from scalellm_amd import kernels
    with kernels.tuning(SLM_ATTN_SPLITS=5, SLM_W4_SPLITK=2):
        assert L.slm_paged_kv_varlen_mha_auto_splits(C.byref(one)) == 5
    assert L.slm_paged_kv_varlen_mha_auto_splits(C.byref(one)) == auto
    # a child process that exports the variable BEFORE the library loads does see it (parse-once)
    code = ("import ctypes as C; from scalellm_amd import _lib; from scalellm_amd._lib import AttnArgs;"
            "L=_lib.lib(); a=AttnArgs(); a.dtype=1; a.batch_size=1; a.n_tokens=1; a.n_heads=32;"
            "a.n_kv_heads=8; a.head_dim=128; a.block_size=16; a.max_q_len=1; a.max_kv_len=4096;"
            "a.k_stride[0]=1024; a.v_stride[0]=1024; print(L.slm_paged_kv_varlen_mha_auto_splits(C.byref(a)))")
    out = subprocess.check_output(["python", "-c", code], cwd=ROOT, text=True,
                                  env=dict(os.environ, SLM_ATTN_SPLITS="9"))
    assert out.strip().endswith("9")
    # source hygiene: the only getenv in the kernel library is the one-time table fill
    csrc = os.path.join(ROOT, "scalellm_amd", "csrc")
    hits = []
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            for i, line in enumerate(open(os.path.join(csrc, f)), 1):
                if "getenv" in line and not line.lstrip().startswith("//"):
                    hits.append((f, i))
    assert hits and all(f == "capi.hip" for f, _ in hits) and len(hits) == 1, hits


def _attn_args_for(bs, n_tokens, max_kv=4096):
    a = AttnArgs()
    a.dtype, a.batch_size, a.n_tokens = 1, bs, n_tokens
    a.n_heads, a.n_kv_heads, a.head_dim, a.block_size = 32, 8, 128, 16
    a.max_q_len, a.max_kv_len = 1, max_kv
    a.k_stride[0], a.v_stride[0] = 1024, 1024
    return a


def _attn(bs, n_tokens, heads=32, kv_heads=8, d=128, block=16, max_kv=4096, max_q=1, splits=0):
    a = AttnArgs()
    a.dtype, a.batch_size, a.n_tokens = 1, bs, n_tokens
    a.n_heads, a.n_kv_heads, a.head_dim, a.block_size = heads, kv_heads, d, block
    a.max_q_len, a.max_kv_len, a.num_splits = max_q, max_kv, splits
    a.k_stride[0], a.v_stride[0] = kv_heads * d, kv_heads * d
    return a


def test_split_kv_heuristic_host_side():
    L = _lib.lib()
    assert L.slm_tuning_clear(None) == 0  # heuristics only: no override knob set
    big = _attn(256, 256)
    assert L.slm_paged_kv_varlen_mha_auto_splits(C.byref(big)) == 1
    assert L.slm_paged_kv_varlen_mha_workspace_bytes(C.byref(big)) == 0
    mid = _attn(32, 32)
    s_mid = L.slm_paged_kv_varlen_mha_auto_splits(C.byref(mid))
    assert 4 <= s_mid <= 16
    one = _attn(1, 1)
    s_one = L.slm_paged_kv_varlen_mha_auto_splits(C.byref(one))
    assert 16 <= s_one <= 64  # >= 64 KV rows per split
    need = L.slm_paged_kv_varlen_mha_workspace_bytes(C.byref(one))
    assert need == 1 * 32 * s_one * (128 + 2) * 4
    short = _attn(1, 1, max_kv=100)
    assert L.slm_paged_kv_varlen_mha_auto_splits(C.byref(short)) == 1
    prefill = _attn(2, 4096, max_q=2048)
    assert L.slm_paged_kv_varlen_mha_auto_splits(C.byref(prefill)) == 1
    forced = _attn(256, 256, splits=7)
    assert L.slm_paged_kv_varlen_mha_auto_splits(C.byref(forced)) == 7
    # MFMA tile kernel (q_len > 1): split only when the history is long against the chunk AND the
    # tiles cannot put a wave on every SIMD; never for plain prefill (kv ~ q)
    one_chunk = _attn(1, 256, max_q=256, max_kv=8192)       # 8 tiles x 8 heads x 4 waves = 256 waves
    s_chunk = L.slm_paged_kv_varlen_mha_auto_splits(C.byref(one_chunk))
    assert s_chunk == 4
    assert L.slm_paged_kv_varlen_mha_workspace_bytes(C.byref(one_chunk)) == 256 * 32 * s_chunk * 130 * 4
    verify_small = _attn(8, 40, max_q=5, max_kv=4096)       # 64 one-wave tiles, 8 splits by length
    assert L.slm_paged_kv_varlen_mha_auto_splits(C.byref(verify_small)) == 8
    verify_big = _attn(120, 600, max_q=5, max_kv=4096)      # 960 waves: the chip is full already
    assert L.slm_paged_kv_varlen_mha_auto_splits(C.byref(verify_big)) == 1
    causal = _attn(1, 2048, max_q=2048, max_kv=2048)
    assert L.slm_paged_kv_varlen_mha_auto_splits(C.byref(causal)) == 1


@pytest.mark.parametrize("mut,code", [
    (dict(heads=30), -1),          # n_heads % n_kv_heads
    (dict(d=7), -2),               # head_dim % 8
    (dict(d=512), -2),             # head_dim > 256
    (dict(block=6), -1),           # not a power of two (mha_params.h:71-74)
])
def test_attention_argument_validation_precedes_any_launch(mut, code):
    L = _lib.lib()
    a = _attn(1, 1, **mut)
    assert L.slm_paged_kv_varlen_mha(C.byref(a), None) == code
    assert L.slm_paged_kv_varlen_mha_auto_splits(C.byref(a)) == 0


def test_attention_null_pointers_and_alignment_rejected():
    L = _lib.lib()
    a = _attn(1, 1)
    assert L.slm_paged_kv_varlen_mha(C.byref(a), None) == -1  # null tensors
    buf = (C.c_char * 4096)()
    base = C.addressof(buf)
    base = (base + 15) & ~15
    for f in ("out", "query", "key_cache", "value_cache", "q_cu_lens", "kv_cu_lens", "block_table",
              "block_cu_lens"):
        setattr(a, f, base)
    a.o_stride[0] = a.q_stride[0] = 32 * 128
    a.o_stride[1] = a.q_stride[1] = 128
    a.k_stride[1] = a.v_stride[1] = 128
    a.query = base + 2  # misaligned
    assert L.slm_paged_kv_varlen_mha(C.byref(a), None) == -5


def test_w4_host_side_planning_and_validation():
    L = _lib.lib()
    assert L.slm_w4_packed_weight_bytes(4096, 4096) == 4096 * 4096 // 2
    assert L.slm_w4_packed_weight_bytes(4096, 4100) == 0  # N % 32
    assert L.slm_w4_packed_sz_bytes(4096, 6144, 128) == 32 * 6144 * 4
    g = W4GemmArgs()
    g.M, g.K, g.N, g.lda, g.ldc, g.group_size, g.dtype = 32, 4096, 4096, 4096, 4096, 128, 1
    assert L.slm_tuning_clear(None) == 0
    ws = L.slm_w4a16_gemm_workspace_bytes(C.byref(g))
    assert ws % (32 * 4096 * 4) == 0  # split-K partials: whole [M, N] fp32 slabs (or none)
    g.group_size = 48
    assert L.slm_w4a16_gemm(C.byref(g), None) == -2
    g.group_size, g.K = 128, 4000
    assert L.slm_w4a16_gemm(C.byref(g), None) == -2
    assert L.slm_w4_prepack(0, None, None, None, None, 128, 64, 128, 1, None, None, None) == -1
    assert L.slm_set_kv_cache(None, None, None, 0, 0, None, None, 0, 8, 128, 1, None) == 0  # empty
    assert L.slm_set_kv_cache(None, None, None, 0, 0, None, None, 3, 8, 128, 1, None) == -1


def test_w8_plane_form_host_side():
    """8-bit weights (slm_hip.h section 3b): sizes and argument validation are pure host code that
    runs before any launch -- packed rows = 2K, the packed group size (the checkpoint's when the int4
    GEMM supports it, else 128-row granularity), format / shape / pointer checks."""
    L = _lib.lib()
    assert L.slm_w8_packed_rows(4096) == 8192 and L.slm_w8_packed_rows(0) == 0
    for K, gs, want in ((4096, 128, 128), (4096, 64, 64), (4096, 32, 32), (4096, 4096, 4096), (4096, 1024, 1024),
                        (14336, 14336, 128),   # per-channel, K not a power of two: written out per 128 rows
                        (384, 384, 128), (4096, 48, 0), (4096, 96, 0), (4096, 3000, 0), (0, 128, 0)):
        assert L.slm_w8_packed_group_size(K, gs) == want, (K, gs)
    W8G, W8A = _lib.SLM_W8_GPTQ, _lib.SLM_W8_AWQ
    # slm_w8_prepack_weights(format, qweight, perm, K, N, wq_out, perm2_out, stream)
    assert L.slm_w8_prepack_weights(W8G, None, None, 128, 64, 256, 512, None) == -1        # NULL qweight
    assert L.slm_w8_prepack_weights(W8G, 256, None, 128, 64, 256, None, None) == -1        # NULL perm2_out
    assert L.slm_w8_prepack_weights(_lib.SLM_W4_GPTQ, 256, None, 128, 64, 256, 512, None) == -2   # a 4-bit format
    assert L.slm_w8_prepack_weights(W8A, 256, None, 100, 64, 256, 512, None) == -2         # K % 64
    assert L.slm_w8_prepack_weights(W8A | _lib.SLM_W4_PAIRED, 256, None, 128, 96, 256, 512, None) == -2  # paired: N % 64
    # slm_w8_prepack_sz(format, qzeros, scales, K, N, group_size, dtype, sz_out, stream)
    assert L.slm_w8_prepack_sz(W8G, None, None, 128, 64, 128, 1, 256, None) == -1          # NULL scales
    assert L.slm_w8_prepack_sz(W8G, None, 256, 128, 64, 48, 1, 256, None) == -2            # group size
    assert L.slm_w8_prepack_sz(W8G, None, 256, 128, 64, 128, 7, 256, None) == -2           # dtype
    assert L.slm_w8_prepack_sz(_lib.SLM_W4_AWQ, None, 256, 128, 64, 128, 1, 256, None) == -2


def test_build_step_inputs_validates_before_any_launch():
    """slm_build_step_inputs(q_lens, kv_cached, block_table, block_cu_lens, n_seqs, block_size,
    n_tokens_padded, commit, positions, q_cu_lens, kv_cu_lens, new_cache_slots, overflow_flag, stream)."""
    L = _lib.lib()
    f = L.slm_build_step_inputs
    assert f(256, 512, 768, 1024, -1, 16, 8, 1, 2048, 4096, 8192, 16384, None, None) == -1     # n_seqs < 0
    assert f(256, 512, 768, 1024, 4, 16, -1, 1, 2048, 4096, 8192, 16384, None, None) == -1     # padded < 0
    assert f(None, 512, 768, 1024, 4, 16, 8, 1, 2048, 4096, 8192, 16384, None, None) == -1     # NULL q_lens
    assert f(256, 512, 768, 1024, 4, 16, 8, 1, None, 4096, 8192, 16384, None, None) == -1      # NULL positions
    assert f(256, 512, 768, 1024, 4, 16, 8, 1, 2048, None, 8192, 16384, None, None) == -1      # NULL q_cu_lens
    assert f(256, 512, 768, 1024, 4, 12, 8, 1, 2048, 4096, 8192, 16384, None, None) == -2      # block size 12
    assert f(256, 512, 768, 1024, 4, 0, 8, 1, 2048, 4096, 8192, 16384, None, None) == -2


def test_deferred_splitk_reduce_host_side():
    """SLM_W4_DEFER_REDUCE: whether a call defers is a pure function of its argument block, and
    slm_rms_norm_splitk validates before any launch."""
    L = _lib.lib()
    assert L.slm_tuning_clear(None) == 0
    g = W4GemmArgs()
    g.M, g.K, g.N, g.lda, g.ldc, g.group_size, g.dtype = 256, 14336, 4096, 14336, 4096, 128, 1
    assert L.slm_w4a16_gemm_deferred_splits(C.byref(g)) == 0          # flag not set
    g.flags = _lib.SLM_W4_DEFER_REDUCE
    n = L.slm_w4a16_gemm_deferred_splits(C.byref(g))
    assert n >= 2                                                      # the down projection is split over K
    assert L.slm_w4a16_gemm_workspace_bytes(C.byref(g)) == n * 256 * 4096 * 4
    g.bias = 4096                                                      # bias: reduced as usual
    assert L.slm_w4a16_gemm_deferred_splits(C.byref(g)) == 0
    g.bias = None
    g.N, g.ldc = 28672, 28672                                          # gate_up at M = 256: never split
    g.K, g.lda = 4096, 4096
    assert L.slm_w4a16_gemm_deferred_splits(C.byref(g)) == 0
    assert L.slm_rms_norm_splitk(256, None, 4, 256, None, 8, 4096, 1e-5, 1, None) == -1   # no partials
    assert L.slm_rms_norm_splitk(256, 256, 0, 256, None, 8, 4096, 1e-5, 1, None) == -1    # n_splits < 1
    assert L.slm_rms_norm_splitk(256, 256, 4, 256, None, 8, 4100, 1e-5, 1, None) == -2    # dim % 8
    assert L.slm_rms_norm_splitk(256, 264, 4, 256, None, 8, 4096, 1e-5, 1, None) == -5    # alignment
    assert L.slm_rms_norm_splitk(256, 256, 4, 256, None, 0, 4096, 1e-5, 1, None) == 0     # empty batch


def test_round2_entry_points_validate_before_any_launch():
    """SiLU*mul epilogue flag, paired prepack format and the split-K RoPE form: argument checks
    that must hold without a GPU (every one returns before the first HIP call)."""
    L = _lib.lib()
    assert L.slm_tuning_clear(None) == 0
    g = W4GemmArgs()
    g.M, g.K, g.N, g.lda, g.ldc, g.group_size, g.dtype = 32, 4096, 28672, 4096, 14336, 128, 1
    g.flags = _lib.SLM_W4_SILU_MUL
    assert L.slm_w4a16_gemm_workspace_bytes(C.byref(g)) >= 0             # a valid plan
    g.flags = _lib.SLM_W4_SILU_MUL | _lib.SLM_W4_DEFER_REDUCE            # cannot be combined
    assert L.slm_w4a16_gemm(C.byref(g), None) == -1
    g.flags = 64                                                          # unknown flag bit
    assert L.slm_w4a16_gemm(C.byref(g), None) == -1
    g.flags, g.N, g.ldc = _lib.SLM_W4_SILU_MUL, 96, 48                    # N % 64 != 0
    assert L.slm_w4a16_gemm(C.byref(g), None) == -2
    # prepack: PAIRED needs N % 64 == 0; unknown format bits are rejected
    assert L.slm_w4_prepack_weights(_lib.SLM_W4_AWQ | _lib.SLM_W4_PAIRED, 256, None, 128, 96, 256, None) == -2
    assert L.slm_w4_prepack_weights(_lib.SLM_W4_AWQ | 0x40, 256, None, 128, 128, 256, None) == -2
    assert L.slm_w4_prepack_sz(_lib.SLM_W4_GPTQ | _lib.SLM_W4_PAIRED, None, 256, 128, 96, 128, 1, 256, None) == -2
    # slm_rope_kv_append_splitk(partials, n_splits, q, q_ts, k, k_ts, v, v_ts, positions, cos_sin,
    #                           cos_sin_is_f32, rot_dim, interleaved, slot_ids, kc, vc, T, H, HKV, D, dtype, stream)
    ok = dict(partials=256, n_splits=4, q=512, q_ts=6144, k=1024, k_ts=6144, v=2048, v_ts=6144, pos=4096,
              cs=8192, f32=1, rot=128, inter=0, slots=None, kc=None, vc=None, T=8, H=32, HKV=8, D=128, dt=1)

    def call(**kw):
        a = dict(ok, **kw)
        return L.slm_rope_kv_append_splitk(a["partials"], a["n_splits"], a["q"], a["q_ts"], a["k"], a["k_ts"],
                                           a["v"], a["v_ts"], a["pos"], a["cs"], a["f32"], a["rot"], a["inter"],
                                           a["slots"], a["kc"], a["vc"], a["T"], a["H"], a["HKV"], a["D"],
                                           a["dt"], None)
    assert call(T=0) == 0                       # empty batch
    assert call(partials=None) == -1
    assert call(n_splits=0) == -1
    assert call(cs=None) == -1                  # the split-K form needs the rotary table
    assert call(slots=4096) == -1               # append without caches
    assert call(rot=20) == -2                   # rot_dim % 8 (16-byte partial loads)
    assert call(D=130, rot=128) == -2           # head_dim % 4
    assert call(q_ts=6146) == -2
    assert call(partials=264) == -5             # alignment
    assert call(dt=7) == -2                     # unknown dtype code


def test_python_mirror_fails_loudly_on_cpu_tensors():
    """No CPU / PyTorch fallback anywhere in the product path."""
    from scalellm_amd import kernels
    from scalellm_amd._lib import SlmError
    q = torch.zeros(1, 8, 64, dtype=torch.float16)
    kc = torch.zeros(16, 2, 64, dtype=torch.float16)
    cu = torch.tensor([0, 1], dtype=torch.int32)
    with pytest.raises(SlmError):
        kernels.paged_kv_varlen_mha(q.clone(), q, kc, kc, cu, cu, cu[:1], cu, None, 8, 1, 1, 1.0)
    with pytest.raises(SlmError):
        kernels.set_kv_cache(cu[:1], kc[:1], kc[:1], kc, kc.clone())
    with pytest.raises(SlmError):
        kernels.rms_norm(torch.zeros(2, 64, dtype=torch.float16), torch.zeros(2, 64, dtype=torch.float16),
                         torch.ones(64, dtype=torch.float16), 1e-5)


def test_product_path_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under scalellm_amd/ may reference it."""
    pkg = os.path.join(ROOT, "scalellm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and \
                    "libslm_oracle" not in txt, f"{f} references the oracle"


def test_gemv_norm_prologue_host_side():
    """slm_w4a16_gemv_norm: whether a call is eligible is a pure function of the argument block
    (and the tuning table), and every argument check returns before the first HIP call."""
    from scalellm_amd._lib import W4NormPrologue
    L = _lib.lib()
    assert L.slm_tuning_clear(None) == 0
    g = W4GemmArgs()
    g.M, g.K, g.N, g.lda, g.ldc, g.group_size, g.dtype = 1, 4096, 6144, 4096, 6144, 128, 1
    assert L.slm_w4a16_gemv_norm_supported(C.byref(g)) == 1          # the M = 1 GEMV
    assert L.slm_w4a16_gemv_norm_supported(None) == 0
    g.M = 2
    assert L.slm_w4a16_gemv_norm_supported(C.byref(g)) == 0          # M = 2..4: MFMA kernel by default
    assert L.slm_tuning_set(b"SLM_W4_GEMV", 2) == 0
    try:
        assert L.slm_w4a16_gemv_norm_supported(C.byref(g)) == 1      # ... on the GEMV with the knob
        g.M = 5
        assert L.slm_w4a16_gemv_norm_supported(C.byref(g)) == 0      # never above 4 rows
        g.M, g.K, g.lda = 4, 16384, 16384                            # A (128 KiB) + the fp32 row > 160 KiB LDS
        assert L.slm_w4a16_gemv_norm_supported(C.byref(g)) == 0
        g.M = 1
        assert L.slm_w4a16_gemv_norm_supported(C.byref(g)) == 1
    finally:
        assert L.slm_tuning_clear(b"SLM_W4_GEMV") == 0
    g.M, g.K, g.lda = 1, 4096, 4096
    g.perm = 4096                                                     # act-order: the gather is a separate launch
    assert L.slm_w4a16_gemv_norm_supported(C.byref(g)) == 0
    g.perm = None
    g.flags, g.N, g.ldc = _lib.SLM_W4_SILU_MUL, 28672, 14336         # gate_up with the SiLU epilogue
    assert L.slm_w4a16_gemv_norm_supported(C.byref(g)) == 1
    g.flags, g.N, g.ldc = 0, 6144, 6144

    g.wq, g.sz, g.c = 1 << 20, 2 << 20, 3 << 20                       # (never dereferenced: all calls fail first)
    n = W4NormPrologue()
    n.x, n.weight, n.eps = 4 << 20, 5 << 20, 1e-5
    assert L.slm_w4a16_gemv_norm(None, C.byref(n), None) == -1
    assert L.slm_w4a16_gemv_norm(C.byref(g), None, None) == -1
    g.M = 8
    assert L.slm_w4a16_gemv_norm(C.byref(g), C.byref(n), None) == -2  # not a GEMV shape
    g.M = 1
    n.weight = None
    assert L.slm_w4a16_gemv_norm(C.byref(g), C.byref(n), None) == -1
    n.weight, n.partials, n.n_splits = 5 << 20, 6 << 20, 2            # x AND partials
    assert L.slm_w4a16_gemv_norm(C.byref(g), C.byref(n), None) == -1
    n.x = None
    n.n_splits = 0                                                    # partials without a slab count
    assert L.slm_w4a16_gemv_norm(C.byref(g), C.byref(n), None) == -1
    n.n_splits, n.residual_in = 2, 7 << 20                            # residual without a second buffer
    assert L.slm_w4a16_gemv_norm(C.byref(g), C.byref(n), None) == -1
    n.residual_out = 7 << 20                                          # ... or with the same one
    assert L.slm_w4a16_gemv_norm(C.byref(g), C.byref(n), None) == -1
    n.residual_out = (7 << 20) + 4096                                 # overlapping rows (8 KiB each)
    assert L.slm_w4a16_gemv_norm(C.byref(g), C.byref(n), None) == -1
    n.residual_out, n.normed_out = 8 << 20, 8 << 20                   # normed_out on top of residual_out
    assert L.slm_w4a16_gemv_norm(C.byref(g), C.byref(n), None) == -1
    n.normed_out = None
    n.residual_out = (8 << 20) + 8                                    # alignment
    assert L.slm_w4a16_gemv_norm(C.byref(g), C.byref(n), None) == -5
    g.M = 0
    assert L.slm_w4a16_gemv_norm(C.byref(g), C.byref(n), None) == 0   # empty batch


def test_ctypes_structs_match_the_c_header(tmp_path):
    """ABI drift guard: size and every field offset of the ctypes mirrors (_lib.py) equal what a C
    compiler makes of include/slm_hip.h."""
    import shutil
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    from scalellm_amd._lib import ArArgs, W4GemmArgs, W4NormPrologue
    structs = {"slm_attn_args": AttnArgs, "slm_w4_gemm_args": W4GemmArgs, "slm_ar_args": ArArgs,
               "slm_w4_norm_prologue": W4NormPrologue}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "slm_hip.h"', 'int main(void) {']
    for cname, ct in structs.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in ct._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "abi.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-I", os.path.dirname(HEADER), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True)
    seen = 0
    for line in out.splitlines():
        cname, field, val = line.split()
        ct = structs[cname]
        want = C.sizeof(ct) if field == "size" else getattr(ct, field).offset
        assert int(val) == want, f"{cname}.{field}: C says {val}, ctypes says {want}"
        seen += 1
    assert seen == sum(len(ct._fields_) + 1 for ct in structs.values())


def test_tuning_knob_names_follow_the_enum_order():
    """capi.hip's name table is indexed by tuning.h's enum: a knob added in one place and not (or elsewhere) in
    the other would silently steer a different knob.  Every enumerator carries its SLM_* name in its comment;
    the two sequences must be identical."""
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scalellm_amd", "csrc")
    enum_names = []
    in_enum = False
    for line in open(os.path.join(root, "tuning.h")):
        if re.match(r"\s*enum\b", line):
            in_enum = True
            continue
        if in_enum:
            m = re.match(r"\s*(TUNE_\w+)\s*(=\s*0)?\s*,\s*//\s*(SLM_\w+)", line)
            if m:
                enum_names.append(m.group(3))
            elif "TUNE_COUNT" in line:
                break
    src = open(os.path.join(root, "capi.hip")).read()
    table = src[src.index("kTuneNames[TUNE_COUNT]"):]
    table = table[:table.index("};")]
    table_names = re.findall(r'"(SLM_\w+)"', table)
    assert len(enum_names) > 30 and enum_names == table_names, (enum_names, table_names)
