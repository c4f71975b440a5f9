"""End-to-end LOGITS parity of the HIP decode path (BASELINE north_star: "Outputs match the
reference CPU path in examples/cpu_offline_inference.py on the same inputs -- logits within
stated fp tolerance").

(a) Llama: scalellm_amd.decode.LlamaDecodeStep (RMSNorm -> int4 qkv -> RoPE+KV-append -> paged
    attention -> int4 o_proj -> RMSNorm -> int4 gate_up -> SiLU*mul -> int4 down -> ... -> lm_head,
    every op a HIP kernel of ours except the embedding gather and the lm_head GEMM), driven the way
    the engine drives the reference (src/models/meta/llama.h:123-265, Batch::prepare_model_input
    engine/batch.cpp:77-270): prefill (incl. a chunked prefill + decode MIXED batch), then decode
    steps, through one paged KV cache with shuffled block ids -- against an fp32 forward composed
    ONLY from oracle.* (tests/e2e_common.OracleLlama, itself pinned against HuggingFace
    LlamaForCausalLM by tests/test_e2e_oracle_cpu.py) on the same checkpoint-format int4 weights.
(b) BASELINE config 1: GPT-2-small-shaped model, fp16 / bf16, MHA D=64, learned positions (no
    RoPE), attention + KV append through scalellm_amd.kernels.*, against HuggingFace fp32 logits
    (the prompt shapes of examples/cpu_offline_inference.py:4-9).

Stated tolerance (relative L2 error of the logits, per step), Llama bf16 stacks, TWO references:
  * the pure-fp32 oracle forward (= the reference CPU path): <= 3e-2.  This bound is the price of
    bf16 STORAGE, not of the kernels: every activation, the residual stream and the KV cache are
    rounded to 8 mantissa bits between ops (measured on MI355X: 0.5-1.0e-2 on the tiny stacks,
    1.8-2.1e-2 on the 2-layer 4096-wide stack, where random weights make the attention logits
    ~3.6 sigma wide and amplify it; against the twin below 2.5-4.3e-3 and 8.7-9.8e-3);
  * the same oracle forward with exactly those storage roundings emulated (OracleLlama(storage=
    "bf16")): <= 1.5e-2.  Stage by stage (tools/diag_e2e.py, profiles/r02_e2e_stage_errors.txt) the
    HIP path reproduces the twin's q / k / v BIT FOR BIT and its attention output to 1e-4..9e-4;
    from there on a perturbation eps in front of a bf16 rounding comes out as ~sqrt(eps * ulp)
    (a fraction eps/ulp of the elements flips by one ulp), so 1e-4 becomes ~1e-3 after the next
    rounded op and 3e-3..9e-3 at the logits: chaotic rounding, not a kernel error -- which is why
    the twin is only 2-3x tighter than the fp32 reference and not 100x.
GPT-2 (12 layers, the whole residual stream in the 16-bit dtype) <= 3e-2 bf16 / 4e-3 fp16; greedy
token ids identical wherever the reference's top-2 margin exceeds 4x the largest absolute logit
error of that row (a smaller margin is a coin flip at any 16-bit precision), and in >= 97 % of
all rows overall (two rows of a 20-row set; observed in round 6: 19/20 ... 319/320, printed by every test).
"""
import numpy as np
import pytest
import torch

from oracle import oracle
from tests.e2e_common import OracleLlama, Sequences, check_logits

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0) if torch.cuda.is_available() else None


def _params(inp):
    from scalellm_amd.layers import InputParameters
    t = lambda a: torch.from_numpy(a).to(DEV)  # noqa: E731
    return (t(inp["tokens"]), t(inp["positions"]),
            InputParameters(q_cu_seq_lens=t(inp["q_cu"]), kv_cu_seq_lens=t(inp["kv_cu"]),
                            new_cache_slots=t(inp["slots"]), block_tables=t(inp["table"]),
                            cu_block_lens=t(inp["bcu"]), q_max_seq_len=inp["max_q"],
                            kv_max_seq_len=inp["max_kv"]))


def _oracle_twin(model, quant_method, group_size, storage=None, bits=4):
    """The same model as a LlamaDecodeStep(keep_checkpoint=True), rebuilt on the CPU from the
    CHECKPOINT-format int4 tensors: oracle.{awq,gptq}_dequant = construct_weights
    (qlinear_impl.cpp:21-100), then fp32 matmul (:171-183)."""
    s = model.shape
    f = lambda t: t.float().cpu().numpy()  # noqa: E731
    layers = []
    for li, ck in enumerate(model.ckpt):
        W = {}
        for name, t in ck.items():
            qw, qz = t["qweight"].cpu().numpy(), t["qzeros"].cpu().numpy()
            sc = f(t["scales"].to(model.dtype))   # scales as the layer rounds them to T
            W[name] = (oracle.awq_dequant(qw, qz, sc, group_size, bits=bits) if quant_method == "awq"
                       else oracle.gptq_dequant(qw, qz, sc, group_size, bits=bits))
        L = model.layers[li]
        W["in_norm"], W["post_norm"] = f(L["in_norm"]), f(L["post_norm"])
        layers.append(W)
    D = s.head_dim
    inv_freq = (1.0 / (s.rope_theta ** (np.arange(0, D, 2, dtype=np.float32) / D))).astype(np.float32)
    return OracleLlama(layers, f(model.final_norm), f(model.embed), f(model.lm_head), s.n_heads,
                       s.n_kv_heads, D, s.rms_eps, inv_freq, model.block_size,
                       model.layers[0]["kv"].key_cache.size(0), storage=storage)


def _llama_cases():
    from scalellm_amd.decode import LlamaShape
    shaped_8b = LlamaShape(hidden=4096, n_heads=32, n_kv_heads=8, head_dim=128, intermediate=14336,
                           n_layers=2, vocab=8192, max_position=1024)
    # 8 query heads per KV head (the Llama-3-70B ratio) and 33 sequences = 66 (sequence, KV head)
    # pairs: the decode steps of this case run their attention on the MFMA tile kernel
    gqa8 = LlamaShape(hidden=256, n_heads=16, n_kv_heads=2, head_dim=64, intermediate=512,
                      n_layers=2, vocab=1024, max_position=512)
    return {"tiny-awq": (LlamaShape.tiny(), "awq", 128, [37, 45, 5, 18]),
            # bits = 8 (two int4 planes): the whole step -- norm prologue off (gather path), deferred
            # reduces, paired gate|up -- on 8-bit checkpoints
            "tiny-awq-8bit": (LlamaShape.tiny(), "awq", 128, [37, 45, 5, 18]),
            "tiny-gptq-g64-8bit": (LlamaShape.tiny(), "gptq", 64, [37, 45, 5, 18]),
            "tiny-gptq-g64": (LlamaShape.tiny(), "gptq", 64, [37, 45, 5, 18]),
            "8b-shaped-2-layers-awq": (shaped_8b, "awq", 128, [23, 45, 12]),
            "tiny-gqa8-awq": (gqa8, "awq", 128, [9, 26] + [3 + (7 * i) % 11 for i in range(31)])}


@pytest.mark.parametrize("name", ["tiny-awq", "tiny-gptq-g64", "8b-shaped-2-layers-awq", "tiny-gqa8-awq",
                                  "tiny-awq-8bit", "tiny-gptq-g64-8bit"])
def test_llama_prefill_then_decode_logits_match_oracle(name):
    from scalellm_amd.decode import LlamaDecodeStep
    shape, quant, gs, prompt_lens = _llama_cases()[name]
    bits = 8 if name.endswith("8bit") else 4
    B, n_decode = 16, 3
    seqs = Sequences(prompt_lens, n_decode + 1, B, shape.vocab, seed=7)
    # schedule: step 0 prefills every prompt, except that sequence 1 is CHUNKED (first 20 tokens
    # now, the other 25 in step 1 -- beside the other sequences' first decode token: a mixed batch
    # of a prefill chunk over history and q_len = 1 rows); then plain decode steps
    first = list(prompt_lens)
    first[1] = 20
    steps = [first, [1 if i != 1 else prompt_lens[1] - 20 for i in range(len(prompt_lens))]]
    steps += [[1] * len(prompt_lens)] * n_decode
    model = LlamaDecodeStep(shape, sum(prompt_lens) + 8, seqs.n_blocks, B, quant_method=quant,
                            group_size=gs, dtype=torch.bfloat16, device=DEV, seed=3, keep_checkpoint=True,
                            bits=bits)
    ref_model = _oracle_twin(model, quant, gs, bits=bits)                   # the reference CPU path: fp32
    twin_model = _oracle_twin(model, quant, gs, storage="bf16", bits=bits)  # + the GPU path's storage roundings
    agree = total = 0
    rels = []
    for si, new_lens in enumerate(steps):
        inp = seqs.inputs(new_lens)
        tokens, positions, params = _params(inp)
        logits = model.forward(tokens, positions, params, return_logits=True)
        torch.cuda.synchronize()
        got = logits.float().cpu().numpy()
        ref = ref_model.forward(inp)
        a, n, rel = check_logits(got, ref, 3e-2, f"{name} step {si} (q_lens {new_lens}) vs fp32 oracle")
        agree, total = agree + a, total + n
        # (8 bits: the twin rounds s (q - z) to bf16 once, the large-M kernels round each int4 plane's
        # value -- two roundings: a slightly wider bound)
        _, _, rel_t = check_logits(got, twin_model.forward(inp), 1.5e-2 if bits == 4 else 2e-2,
                                   f"{name} step {si} (q_lens {new_lens}) vs bf16-storage twin")
        rels.append((round(rel, 5), round(rel_t, 5)))
        seqs.advance(new_lens)
        # teacher forcing with the GPU's greedy token: both paths see the same next input, so
        # every step is compared on its own (a near-tie flip must not cascade)
        seqs.feed(inp, got.argmax(-1))
    print(f"[e2e] {name}: per-step relative L2 error (vs fp32 oracle, vs bf16-storage twin): {rels}; "
          f"greedy ids equal on {agree}/{total} rows")
    # observed (round 6, one MI355X): 19/20, 20/20, 20/20, 162/165, 20/20, 19/20 -- every miss a near tie (a decisive
    # top-2 margin with a different id already failed in check_logits).  The bar: 97 %, or two near ties of a small set
    assert agree >= min(0.97 * total, total - 2), f"{name}: greedy ids agree on only {agree}/{total} rows"


@pytest.mark.parametrize("host", ["py", "cpp"])
def test_llama_two_lane_decode_step_and_cpp_host_match_oracle(host):
    """Round 5 (round-4 review, parity hole): the TWO-LANE decode step and the C++ host step
    (slm::LlamaForCausalLMHip) were pinned to the oracle only transitively (bit-identical to their half batches /
    to the Python mirror).  Here both are compared with the oracle-composed fp32 forward and its bf16-storage
    twin directly: 72 sequences (>= 64: two lanes of 64 + 8 rows on two streams), prefill in one step, then
    three pure-decode steps, same bounds as the one-lane cases above."""
    from scalellm_amd.decode import LlamaDecodeStep, LlamaShape
    shape, quant, gs = LlamaShape.tiny(), "awq", 128
    prompt_lens = [3 + (5 * i) % 17 for i in range(72)]
    B, n_decode = 16, 3
    seqs = Sequences(prompt_lens, n_decode + 1, B, shape.vocab, seed=11)
    model = LlamaDecodeStep(shape, sum(prompt_lens) + 8, seqs.n_blocks, B, quant_method=quant, group_size=gs,
                            dtype=torch.bfloat16, device=DEV, seed=5, keep_checkpoint=True)
    model.lanes_min = 64   # (the automatic rule wants 12 MiB of K + V per sequence: tests/test_host_logic_cpu.py)
    model.reserve_workspaces(sum(prompt_lens) + 8, 64)
    runner = model
    if host == "cpp":
        from scalellm_amd import cpp_host
        cpp_host.load_shim()
        runner = cpp_host.from_decode_step(model, B, sum(prompt_lens) + 8, fused=True, lanes=64)
    ref_model = _oracle_twin(model, quant, gs)
    twin_model = _oracle_twin(model, quant, gs, storage="bf16")
    steps = [list(prompt_lens)] + [[1] * len(prompt_lens)] * n_decode
    agree = total = 0
    for si, new_lens in enumerate(steps):
        inp = seqs.inputs(new_lens)
        tokens, positions, params = _params(inp)
        if host == "cpp":
            from scalellm_amd import cpp_host
            logits = runner.decode_step(tokens, positions, cpp_host.cpp_params(params), return_logits=True)
            lanes = runner.last_lanes()
        else:
            logits = model.forward(tokens, positions, params, return_logits=True)
            lanes = model.last_lanes
        torch.cuda.synchronize()
        assert lanes == (2 if si > 0 else 1), (host, si, lanes)
        got = logits.float().cpu().numpy()
        a, n, _ = check_logits(got, ref_model.forward(inp), 3e-2, f"two lanes ({host}) step {si} vs fp32 oracle")
        agree, total = agree + a, total + n
        check_logits(got, twin_model.forward(inp), 1.5e-2, f"two lanes ({host}) step {si} vs bf16-storage twin")
        seqs.advance(new_lens)
        seqs.feed(inp, got.argmax(-1))
    print(f"[e2e] greedy ids equal on {agree}/{total} rows")
    assert agree >= min(0.97 * total, total - 2), f"greedy ids agree on only {agree}/{total} rows"


def _all_row_check(h_got, lm_head_f32, ref_model, twin_model, inp, what):
    """EVERY row of a step (not only each sequence's last token: a speculative-verify step samples at all
    k + 1 positions, speculative_engine.cpp:162-185): final-norm hidden states and the logits they give, against
    the fp32 oracle forward (<= 3e-2) and its bf16-storage twin (<= 1.5e-2).  Returns (#rows whose greedy id
    equals the fp32 oracle's, #rows)."""
    ref_model.forward(inp)
    twin_model.forward(inp)
    got = h_got.astype(np.float32) @ lm_head_f32
    out = None
    for model, tol, tag in ((ref_model, 3e-2, "fp32 oracle"), (twin_model, 1.5e-2, "bf16-storage twin")):
        ref = model.trace["normed"].astype(np.float32) @ lm_head_f32
        res = check_logits(got, ref, tol, f"{what} vs {tag} (all {len(got)} rows)")
        out = out or res
    return out[0], out[1]


@pytest.mark.parametrize("host", ["py", "cpp"])
def test_llama_config5_mixed_step_matches_oracle_on_every_row(host):
    """Round 6 (round-5 review, parity hole 1): a BASELINE configs[4]-shaped step through the whole HIP stack
    against the oracle twin DIRECTLY, on every query row -- one batch holding a prefill chunk over history
    (q_len 19 on 16 cached tokens), speculative-verify rows (q_len = k + 1 = 5 over history: the pending token +
    4 draft tokens, speculative_engine.cpp:162-239), plain decode rows (q_len 1) and a fresh prefill; then the
    engine's bookkeeping after validation -- every verify sequence keeps a different number of accepted tokens
    (1..5: its cache position moves by that much, the rejected draft rows stay in the cache as garbage past the
    end, batch.cpp:304-350) -- and a second mixed step (verify again + decode) whose rows read that cache.
    host = py: decode.LlamaDecodeStep; cpp: slm::LlamaForCausalLMHip (csrc/shim/slm_llama_hip.cpp)."""
    from scalellm_amd.decode import LlamaDecodeStep, LlamaShape
    shape, quant, gs, B = LlamaShape.tiny(), "awq", 128, 16
    prompt_lens = [35, 21, 9, 40, 12, 30, 7, 26, 18, 5, 33, 14]
    n = len(prompt_lens)
    verify, K1 = [1, 2, 3, 4, 5, 6], 5          # sequences that run k + 1 = 5 rows per step
    seqs = Sequences(prompt_lens, 24, B, shape.vocab, seed=23)
    rng = np.random.default_rng(5)
    model = LlamaDecodeStep(shape, 256, seqs.n_blocks, B, quant_method=quant, group_size=gs, dtype=torch.bfloat16,
                            device=DEV, seed=9, keep_checkpoint=True)
    model.reserve_workspaces(256, 128)
    runner = None
    if host == "cpp":
        from scalellm_amd import cpp_host
        cpp_host.load_shim()
        runner = cpp_host.from_decode_step(model, B, 256, fused=True, lanes=64)
    ref_model = _oracle_twin(model, quant, gs)
    twin_model = _oracle_twin(model, quant, gs, storage="bf16")
    lm_head = model.lm_head.float().cpu().numpy()

    def run(new_lens, what):
        inp = seqs.inputs(new_lens)
        tokens, positions, params = _params(inp)
        if host == "cpp":
            from scalellm_amd import cpp_host
            h = runner.forward(tokens, positions, cpp_host.cpp_params(params))
        else:
            model.forward(tokens, positions, params, return_logits=True)
            h = model.buf["normed"][:tokens.numel()]
        torch.cuda.synchronize()
        h = h.float().cpu().numpy()
        a, t = _all_row_check(h, lm_head, ref_model, twin_model, inp, f"config-5 step ({host}) {what}")
        return inp, h @ lm_head, a, t

    agree = total = 0
    # step 0: prefill (sequence 0 only its first 16 tokens; the last sequence sits the step out)
    first = list(prompt_lens)
    first[0], first[-1] = 16, 0
    inp, lg, a, t = run(first, "prefill")
    agree, total = agree + a, total + t
    seqs.advance(first)
    last = inp["q_cu"][1:] - 1
    seqs.feed(inp, lg[last].argmax(-1))
    for rnd in range(2):
        # the draft model's proposals: 4 tokens behind the pending one
        for s in verify:
            seqs.tokens[s] = seqs.tokens[s][:seqs.cached[s] + 1] + rng.integers(0, shape.vocab, size=K1 - 1).tolist()
        new = [K1 if s in verify else 1 for s in range(n)]
        if rnd == 0:
            new[0], new[-1] = prompt_lens[0] - 16, prompt_lens[-1]    # chunk over history + a fresh prefill
        inp, lg, a, t = run(new, f"mixed step {rnd} (q_lens {new})")
        agree, total = agree + a, total + t
        # validation: verify sequence s keeps `acc` of its 5 rows (1 = only the pending token ... 5 = all drafts);
        # its next input token is the target model's choice at the last accepted row
        row0 = {s: int(inp["q_cu"][i]) for i, s in enumerate(inp["rows"])}
        adv = list(new)
        for j, s in enumerate(verify):
            acc = 1 + (j + rnd) % K1
            adv[s] = acc
            seqs.tokens[s] = seqs.tokens[s][:seqs.cached[s] + acc] + [int(lg[row0[s] + acc - 1].argmax())]
        seqs.advance(adv)
        for i, s in enumerate(inp["rows"]):
            if s not in verify and seqs.cached[s] == len(seqs.tokens[s]):
                seqs.tokens[s].append(int(lg[int(inp["q_cu"][i + 1]) - 1].argmax()))
    print(f"[e2e] config-5 mixed steps ({host}): greedy ids equal on {agree}/{total} rows")
    assert agree >= 0.97 * total, f"greedy ids agree on only {agree}/{total} rows"


@pytest.mark.parametrize("nd", [1, 5])
def test_llama_model_runner_replay_matches_oracle(nd):
    """Round 6 (parity hole 1): the REPLAYED graph of ModelRunner (model_runner.cpp:112-211) against the oracle
    twin directly -- until now a replay was only compared with the eager step.  nd = 5: every sequence brings
    k + 1 = 5 verify rows (`num_decoding_tokens`, the reference's speculative graph); nd = 1: plain decode, and
    with 72 sequences the replayed graph is the TWO-LANE variant."""
    from scalellm_amd.decode import LlamaDecodeStep, LlamaShape
    from scalellm_amd.model_runner import ModelRunner, ModelRunnerOptions
    shape, quant, gs, B = LlamaShape.tiny(), "awq", 128, 16
    bs = 72 if nd == 1 else 12
    prompt_lens = [4 + (7 * i) % 29 for i in range(bs)]
    seqs = Sequences(prompt_lens, 4 * nd + 2, B, shape.vocab, seed=31)
    rng = np.random.default_rng(6)
    model = LlamaDecodeStep(shape, max(sum(prompt_lens), bs * nd) + 8, seqs.n_blocks, B, quant_method=quant,
                            group_size=gs, dtype=torch.bfloat16, device=DEV, seed=13, keep_checkpoint=True)
    model.lanes_min = 64
    model.reserve_workspaces(max(sum(prompt_lens), bs * nd) + 8, 128)
    opts = ModelRunnerOptions(block_size=B, cuda_graph_max_seq_len=96, cuda_graph_batch_sizes=[bs],
                              num_decoding_tokens=nd)
    runner = ModelRunner(model, DEV, opts, return_logits=True)
    ref_model = _oracle_twin(model, quant, gs)
    twin_model = _oracle_twin(model, quant, gs, storage="bf16")
    lm_head = model.lm_head.float().cpu().numpy()
    # prefill eagerly (both sides), THEN capture (a capture runs the step: it must not touch live cache rows --
    # the runner's static inputs name slot 0 of the unused block 0)
    inp = seqs.inputs(prompt_lens)
    tokens, positions, params = _params(inp)
    lg = model.forward(tokens, positions, params, return_logits=True).float().cpu().numpy()
    ref_model.forward(inp)
    twin_model.forward(inp)
    seqs.advance(prompt_lens)
    seqs.feed(inp, lg.argmax(-1))
    runner.capture_cuda_graphs(bs)
    agree = total = 0
    for step in range(3):
        for s in range(bs):
            seqs.tokens[s] = seqs.tokens[s][:seqs.cached[s] + 1] + rng.integers(0, shape.vocab, size=nd - 1).tolist()
        inp = seqs.inputs([nd] * bs)
        tokens, positions, params = _params(inp)
        before = runner.num_graph_replayed
        runner.forward(tokens, positions, params)
        torch.cuda.synchronize()
        assert runner.num_graph_replayed == before + 1
        if nd == 1:
            assert runner.graphs[bs].last_variant[0] == 2, runner.graphs[bs].last_variant   # the two-lane graph
        h = model.buf["normed"][:bs * nd].float().cpu().numpy()
        a, t = _all_row_check(h, lm_head, ref_model, twin_model, inp, f"ModelRunner replay nd={nd} step {step}")
        agree, total = agree + a, total + t
        lg = h @ lm_head
        adv = []
        for s in range(bs):   # keep 1..nd rows of every sequence; next token = the target's choice at the last kept row
            acc = 1 + (s + step) % nd
            adv.append(acc)
            seqs.tokens[s] = seqs.tokens[s][:seqs.cached[s] + acc] + [int(lg[s * nd + acc - 1].argmax())]
        seqs.advance(adv)
    print(f"[e2e] ModelRunner replay nd={nd}: greedy ids equal on {agree}/{total} rows")
    assert agree >= 0.97 * total, f"greedy ids agree on only {agree}/{total} rows"


@pytest.mark.parametrize("wide", [False, True])
def test_llama_one_layer_stage_by_stage_against_storage_twin(wide):
    """The tight pin behind the logits bounds: ONE decoder layer, prefill step, every buffer of the
    HIP path against the storage twin's value at the same point.  Before the first chaotic
    rounding the HIP kernels are at rounding-flip level: q / k / v (RMSNorm -> int4 GEMM -> RoPE)
    <= 3e-4 (bit-exact on the narrow model), paged attention output <= 3e-3, and every later stage
    <= 1e-2."""
    from scalellm_amd.decode import LlamaDecodeStep, LlamaShape
    from tests.e2e_common import rel_l2
    if wide:
        shape = LlamaShape(hidden=4096, n_heads=32, n_kv_heads=8, head_dim=128, intermediate=14336,
                           n_layers=1, vocab=8192, max_position=1024)
        prompt_lens = [23, 20, 12]
    else:
        shape, prompt_lens = LlamaShape.tiny(), [37, 20, 5, 18]
        shape.n_layers = 1
    B = 16
    seqs = Sequences(prompt_lens, 2, B, shape.vocab, seed=7)
    model = LlamaDecodeStep(shape, sum(prompt_lens) + 8, seqs.n_blocks, B, quant_method="awq",
                            group_size=128, dtype=torch.bfloat16, device=DEV, seed=3, keep_checkpoint=True)
    twin = _oracle_twin(model, "awq", 128, storage="bf16")
    inp = seqs.inputs(prompt_lens)
    tokens, positions, params = _params(inp)
    model.forward(tokens, positions, params, return_logits=True)
    torch.cuda.synchronize()
    twin.forward(inp)
    T, s = len(inp["tokens"]), shape
    nq, nkv = s.n_heads * s.head_dim, s.n_kv_heads * s.head_dim
    f = lambda t: t.float().cpu().numpy()  # noqa: E731
    qkv = f(model.buf["qkv"][:T])
    err = {"q": rel_l2(qkv[:, :nq].reshape(T, s.n_heads, -1), twin.trace["q"]),
           "k": rel_l2(qkv[:, nq:nq + nkv].reshape(T, s.n_kv_heads, -1), twin.trace["k"]),
           "v": rel_l2(qkv[:, nq + nkv:].reshape(T, s.n_kv_heads, -1), twin.trace["v"]),
           "attn": rel_l2(f(model.buf["attn"][:T]), twin.trace["attn"]),
           "act": rel_l2(f(model.buf["act"][:T]), twin.trace["act"]),
           "resid": rel_l2(f(model.buf["resid"][:T]), twin.trace["resid"]),
           "hidden": rel_l2(f(model.last_hidden), twin.last_hidden)}
    print(f"[e2e] one layer ({'4096-wide' if wide else 'tiny'}) stage errors vs the storage twin: "
          f"{ {k: float(f'{v:.2e}') for k, v in err.items()} }")
    assert max(err["q"], err["k"], err["v"]) <= 3e-4, err
    if not wide:
        assert err["q"] == err["k"] == err["v"] == 0.0, err
    assert err["attn"] <= 3e-3, err
    assert max(err["act"], err["resid"], err["hidden"]) <= 1e-2, err


def test_llama_graph_replayed_decode_step_matches_eager_logits():
    """The captured step (what bench.py times) reproduces the eager step's logits bit for bit."""
    from scalellm_amd.decode import LlamaDecodeStep, LlamaShape, make_decode_inputs
    shape = LlamaShape.tiny()
    bs, kv_len, B = 8, 200, 16
    tokens, positions, params, n_blocks = make_decode_inputs(bs, kv_len, B, DEV, seed=2, vocab=shape.vocab)
    model = LlamaDecodeStep(shape, bs, n_blocks, B, dtype=torch.bfloat16, device=DEV, seed=1, kv_fill="randn")
    model.reserve_workspaces(bs, kv_len)
    eager = model.forward(tokens, positions, params, return_logits=True).clone()
    out = torch.empty_like(eager)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out.copy_(model.forward(tokens, positions, params, return_logits=True))
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)


# ---------------------------------------------------------------------------------------------
# (b) BASELINE config 1: GPT-2 small through kernels.* vs HuggingFace fp32
# ---------------------------------------------------------------------------------------------
class HipGPT2:
    """GPT-2 whose attention (paged KV append + paged_kv_varlen_mha) runs on the HIP kernels; layer
    norm, the dense projections and gelu_new are plain torch ops on the GPU in the test dtype (GPT-2
    has no int4 linears and no RMSNorm: they are not hot-path kernels of this library)."""

    def __init__(self, hf, dtype, block_size, n_blocks):
        from scalellm_amd.layers import KVCache
        c = hf.config
        self.c, self.dtype, self.B = c, dtype, block_size
        self.sd = {k: v.detach().to(DEV).to(dtype) for k, v in hf.state_dict().items()}
        self.H, self.D = c.n_head, c.n_embd // c.n_head
        self.kv = [KVCache(n_blocks, block_size, self.H, self.D, dtype, DEV) for _ in range(c.n_layer)]

    def forward(self, tokens, positions, params):
        from scalellm_amd import kernels
        F = torch.nn.functional
        sd, c = self.sd, self.c
        x = sd["transformer.wte.weight"][tokens.long()] + sd["transformer.wpe.weight"][positions.long()]
        T = tokens.numel()
        for i in range(c.n_layer):
            p = f"transformer.h.{i}."
            h = F.layer_norm(x, (c.n_embd,), sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], c.layer_norm_epsilon)
            qkv = torch.addmm(sd[p + "attn.c_attn.bias"], h, sd[p + "attn.c_attn.weight"])
            q, k, v = (t.reshape(T, self.H, self.D) for t in qkv.split(c.n_embd, dim=-1))
            kc, vc = self.kv[i].get_kv_cache()
            kernels.set_kv_cache(params.new_cache_slots, k, v, kc, vc)
            out = torch.empty(T, self.H, self.D, device=DEV, dtype=self.dtype)
            kernels.paged_kv_varlen_mha(out, q, kc, vc, params.q_cu_seq_lens, params.kv_cu_seq_lens,
                                        params.block_tables, params.cu_block_lens, None, self.B,
                                        params.q_max_seq_len, params.kv_max_seq_len, self.D ** -0.5)
            x = x + torch.addmm(sd[p + "attn.c_proj.bias"], out.view(T, -1), sd[p + "attn.c_proj.weight"])
            h = F.layer_norm(x, (c.n_embd,), sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], c.layer_norm_epsilon)
            h = F.gelu(torch.addmm(sd[p + "mlp.c_fc.bias"], h, sd[p + "mlp.c_fc.weight"]), approximate="tanh")
            x = x + torch.addmm(sd[p + "mlp.c_proj.bias"], h, sd[p + "mlp.c_proj.weight"])
        x = F.layer_norm(x, (c.n_embd,), sd["transformer.ln_f.weight"], sd["transformer.ln_f.bias"],
                         c.layer_norm_epsilon)
        last = (params.q_cu_seq_lens[1:] - 1).long()
        return x[last].float() @ sd["transformer.wte.weight"].float().t()  # tied lm_head


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_gpt2_small_config1_logits_match_hf_fp32(dtype):
    transformers = pytest.importorskip("transformers")
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
    tol = 3e-2 if dtype == "bf16" else 4e-3
    torch.manual_seed(0)
    cfg = transformers.GPT2Config()  # GPT-2 small: 12 layers, 12 heads (D = 64), 768, vocab 50257
    hf = transformers.GPT2LMHeadModel(cfg).eval()
    prompt_lens, n_new, B = [5, 7, 6, 6], 6, 8   # cpu_offline_inference.py:4-9: 4 short prompts
    seqs = Sequences(prompt_lens, n_new + 1, B, cfg.vocab_size, seed=0)
    model = HipGPT2(hf, tdt, B, seqs.n_blocks)
    new_lens = list(prompt_lens)
    agree = total = 0
    for step in range(n_new):
        inp = seqs.inputs(new_lens)
        tokens, positions, params = _params(inp)
        got = model.forward(tokens, positions, params).cpu().numpy()
        seqs.advance(new_lens)
        with torch.no_grad():
            ref = np.stack([hf(torch.tensor([seqs.tokens[s][:seqs.cached[s]]])).logits[0, -1].numpy()
                            for s in range(len(prompt_lens))])
        a, n, rel = check_logits(got, ref, tol, f"gpt2 {dtype} step {step}")
        agree, total = agree + a, total + n
        seqs.feed(inp, got.argmax(-1))
        new_lens = [1] * len(prompt_lens)
    print(f"[e2e] gpt2 {dtype}: greedy ids equal on {agree}/{total} rows")
    assert agree >= min(0.97 * total, total - 2), f"greedy ids agree on only {agree}/{total} rows"
