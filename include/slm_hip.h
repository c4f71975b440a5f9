/*
 * slm_hip.h -- C ABI of libslm_hip.so: the MI355X (gfx950 / CDNA4) kernel library
 * behind ScaleLLM's decode hot path.
 *
 * Drop-in boundary: every entry point below replaces one C++ kernel-level
 * symbol of the reference (vectorch-ai/ScaleLLM, paths relative to the
 * reference root).  The reference passes torch::Tensor; a thin libtorch shim
 * (scalellm_amd/csrc/shim/) adapts tensor -> (pointer,
 * strides, sizes) and calls these functions, so the reference src/layers tree compiles
 * unchanged.
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes only; no torch / C++ types;
 *   - device pointers unless stated "host";
 *   - asynchronous on `stream` (a hipStream_t passed as void*); never
 *     synchronises, never allocates, never reads device memory on the host --
 *     therefore safe under hipGraph stream capture (the only exceptions are the
 *     init-time slm_shm_* / slm_ar_read_error helpers of section 6, marked there)
 *     (reference: src/engine/model_runner.cpp:141-178);
 *   - returns 0 (SLM_OK) or a negative slm_status; never aborts;
 *   - thread-safe per stream (one Worker thread per GPU in the reference:
 *     src/engine/worker.cpp:202-213).
 */
#ifndef SLM_HIP_H_
#define SLM_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLM_API __attribute__((visibility("default")))

typedef enum slm_status {
  SLM_OK = 0,
  SLM_ERR_INVALID_ARG = -1,   /* null pointer, bad size, non power-of-two block_size ... */
  SLM_ERR_UNSUPPORTED = -2,   /* dtype / head_dim / bits / group not supported */
  SLM_ERR_WORKSPACE = -3,     /* workspace missing or too small */
  SLM_ERR_LAUNCH = -4,        /* hipLaunchKernel reported an error */
  SLM_ERR_ALIGNMENT = -5      /* pointer or stride not 16-byte aligned */
} slm_status;

typedef enum slm_dtype {
  SLM_F16 = 0,  /* IEEE half   (torch::kHalf)     */
  SLM_BF16 = 1  /* bfloat16    (torch::kBFloat16) */
} slm_dtype;

SLM_API const char* slm_status_string(int status);
SLM_API const char* slm_version(void);
/* hipGetErrorString of the last launch status seen by this thread (detail for SLM_ERR_LAUNCH). */
SLM_API const char* slm_last_hip_error(void);

/* ========================================================================== */
/* 0. Launch-shape overrides (sweeps, tests that force one kernel).           */
/*    The library never reads the environment per call: the SLM_* variables   */
/*    (SLM_ATTN_NW, SLM_ATTN_SPLITS, SLM_W4_MT, SLM_W4_SPLITK ... -- the full */
/*    list is scalellm_amd/csrc/tuning.h) are parsed ONCE, at first use, and   */
/*    only these calls change a knob afterwards.  Process-wide; not meant to  */
/*    be flipped while other threads launch.  name = the variable's name.     */
/*    slm_tuning_clear(NULL) clears every knob (environment values included). */
/* ========================================================================== */
SLM_API int slm_tuning_set(const char* name, int32_t value);
SLM_API int slm_tuning_clear(const char* name);
SLM_API int slm_tuning_get(const char* name, int32_t* value, int32_t* is_set);

/* ========================================================================== */
/* 1. Paged-KV varlen attention (prefill / chunked prefill / decode / verify) */
/*    replaces  llm::paged_kv_varlen_mha                                      */
/*              src/kernels/attention/attn_api.h:12-27 (impl attn_api.cpp:14) */
/*    caller    ScaleAttnHandler::batch_decode                                */
/*              src/layers/attention/scale_attn_handler.cpp:44-68             */
/*    params    mirror MHAPagedKVParams, src/kernels/attention/mha_params.h   */
/*              :11-117 (strides in ELEMENTS; last dim contiguous).           */
/*    Block table = flattened first-slot ids + CSR offsets                    */
/*    (src/engine/batch.cpp:206-209, src/models/parameters.h:50-55);          */
/*    slot(kv_idx) = block_table[block_cu_lens[b] + (kv_idx >> log2(bs))]     */
/*                   + (kv_idx & (bs-1))   (kernel/sm80_kernel_mha.cuh:146).  */
/* ========================================================================== */
typedef struct slm_attn_args {
  void* out;                /* [n_tokens, n_heads, head_dim]                  */
  const void* query;        /* [n_tokens, n_heads, head_dim]                  */
  const void* key_cache;    /* [n_slots, n_kv_heads, head_dim]                */
  const void* value_cache;  /* [n_slots, n_kv_heads, head_dim]                */
  int64_t o_stride[2];      /* {token stride, head stride} in elements        */
  int64_t q_stride[2];
  int64_t k_stride[2];      /* {slot stride, head stride}                     */
  int64_t v_stride[2];
  const int32_t* q_cu_lens;      /* [batch+1]                                 */
  const int32_t* kv_cu_lens;     /* [batch+1]                                 */
  const int32_t* block_table;    /* [sum_b ceil(kv_len_b / block_size)]       */
  const int32_t* block_cu_lens;  /* [batch+1]                                 */
  const float* alibi_slopes;     /* [n_heads] or NULL                         */
  int32_t dtype;            /* slm_dtype of out/query/caches                  */
  int32_t batch_size;
  int32_t n_tokens;         /* = query.size(0) = q_cu_lens[batch].  Only a pure-decode batch (max_q_len <= 1)
                             * may carry MORE rows than q_cu_lens[batch]: the graph padding of batch.cpp:219-244,
                             * left untouched.  (With max_q_len > 1 the plan relies on the equality: when
                             * n_tokens == batch_size * max_q_len every sequence has max_q_len tokens and only
                             * that row class is launched.) */
  int32_t n_heads;
  int32_t n_kv_heads;
  int32_t head_dim;         /* multiple of 8, <= 256                          */
  int32_t block_size;       /* power of two (mha_params.h:71-74)              */
  int32_t max_q_len;        /* scheduling hint, as in the reference           */
  int32_t max_kv_len;       /* scheduling hint (unused by the reference): sizes the KV splits /
                             * the balanced decode partition.  Results are correct for ANY value:
                             * a sequence longer than the hint is streamed by one workgroup
                             * (slower, never wrong or out of bounds)                          */
  float sm_scale;
  float logits_soft_cap;    /* 0 = off                                        */
  int32_t sliding_window;   /* -1 = off                                       */
  void* workspace;          /* split-KV scratch, may be NULL if bytes == 0    */
  size_t workspace_bytes;
  int32_t num_splits;       /* 0 = auto (heuristic); >0 forces the split count */
  int32_t total_kv_len;     /* scheduling hint, 0 = unknown: kv_cu_lens[batch] if the host has it
                             * (Batch::prepare_model_input builds cu_seq_lens on the host, batch.cpp:137).
                             * total_kv_len == batch_size * max_kv_len tells the plan that every sequence
                             * has max_kv_len tokens: a pure-decode call that needs no KV split then skips
                             * the balanced partition and with it the combine launch that would find
                             * nothing to merge.  Like max_kv_len it only shapes the launch: results are
                             * correct whatever the value (a wrong "uniform" claim costs balance, not bits) */
  int32_t phase;            /* 0 = the whole call (default).  A call that splits the KV range runs as stream
                             * kernel(s) + a combine pass; a host that overlaps calls on two streams (the two
                             * decode lanes) may issue the halves separately: 1 = everything BUT
                             * the combine pass, 2 = the combine pass only (a no-op when the plan needs none).
                             * Same arguments for both; phase 1 then phase 2 on one stream == phase 0. */
  int32_t reserved;
} slm_attn_args;

/* Scratch needed for `a` (depends only on host-side sizes; AttentionHandler::
 * get_estimate_workspace_size / set_workspace, src/layers/attention/handler.h
 * :40-47 is the natural home).  Pass num_splits to query a forced split count.*/
SLM_API size_t slm_paged_kv_varlen_mha_workspace_bytes(const slm_attn_args* a);
/* Split count the heuristic would pick for `a` (host-side sizes only).        */
SLM_API int32_t slm_paged_kv_varlen_mha_auto_splits(const slm_attn_args* a);
/* which kernel the plan gives the q_len = 1 (decode) rows of this call: 0 = attn_token_kernel (HBM
 * stream, dot2), 1 = attn_tile_kernel (MFMA tile form: wide GQA groups), -1 = invalid arguments.
 * A pure function of the argument block and the tuning table (what a bench labels its roofline with). */
SLM_API int32_t slm_paged_kv_varlen_mha_decode_kernel(const slm_attn_args* a);
SLM_API int slm_paged_kv_varlen_mha(const slm_attn_args* a, void* stream);

/* ========================================================================== */
/* 2. KV-cache append                                                         */
/*    replaces  llm::kernel::set_kv_cache                                     */
/*              src/kernels/kv_cache_kernels.h:6-11 (.cu:9-78)                */
/*    caller    KVCache::set_kv_cache_cuda  src/memory/kv_cache.cpp:53-57     */
/*    cache[slot_ids[t], h, d] = kv[t, h, d] for K and V (bit-exact copy).    */
/* ========================================================================== */
SLM_API int slm_set_kv_cache(const int32_t* slot_ids,  /* [n_tokens]           */
                             const void* keys,         /* [n_tokens, n_kv_heads, head_dim] */
                             const void* values,
                             int64_t k_token_stride,   /* elements             */
                             int64_t v_token_stride,
                             void* key_cache,          /* [n_slots, n_kv_heads, head_dim] contiguous */
                             void* value_cache,
                             int64_t n_tokens, int32_t n_kv_heads, int32_t head_dim,
                             int32_t dtype, void* stream);

/* ========================================================================== */
/* 3. int4 weight prepack (checkpoint format -> MFMA-native layout)           */
/*    replaces  marlin::gptq_repack / marlin::awq_repack                      */
/*              src/kernels/quantization/marlin.h:27-35                       */
/*              (gptq_repack.cu:252, awq_repack.cu:191) and the host-side     */
/*              scale / zero-point permutations of                            */
/*              qlinear_awq_marlin_impl.cpp:34-125,                           */
/*              qlinear_gptq_marlin_impl.cpp:41-71.                           */
/*    Inputs are the stable on-disk formats:                                  */
/*      GPTQ: qweight [K/8, N] int32 (nibble k%8 at bit 4*(k%8)),             */
/*            qzeros  [G, N/8] int32 (plain order, zero = stored + 1),        */
/*            scales  [G, N] T, optional g_idx [K] (act-order);               */
/*      AWQ : qweight [K, N/8] int32, qzeros [G, N/8] int32, both with the    */
/*            [0,2,4,6,1,3,5,7] nibble interleave, zero = stored;             */
/*            scales  [G, N] T.                                               */
/*    Output layout (owned by this library):                   */
/*      wq  [K/64][N/32][64 lanes][4] uint32  -- lane l, word j holds the 8   */
/*           nibbles n = 32*nt + (l&31), k = 64*kt + 16*j + 8*(l>>5) + p'     */
/*           in a pair-interleaved nibble order (MFMA 32x32x16 B-fragment);   */
/*      sz  [G][N] {scale, magic + zero} as 2 x T (magic = 128 for bf16,      */
/*           1024 for fp16: the sum is exact in T; fused scale/zero table);   */
/*      perm[K] int32 (act-order only): row k' of wq = checkpoint row perm[k']*/
/*           perm[k'] < 0 marks a PADDING row (weights 0, activation column   */
/*           gathered as +0.0): a row-parallel act-order shard               */
/*           (qlinear_gptq_marlin_impl.cpp:236-243: sharded g_idx, full       */
/*           scales) holds an uneven number of rows of every group, so the    */
/*           host pads each group's sorted rows to a multiple of 32 and packs */
/*           with K = the padded row count, group_size = 32 and one scale row */
/*           per 32-row block (tools: kernels.gptq_repack / slm::W4Linear).   */
/* ========================================================================== */
typedef enum slm_w4_format {
  SLM_W4_GPTQ = 0,
  SLM_W4_AWQ = 1,
  SLM_W8_GPTQ = 2, /* 8-bit checkpoints: slm_w8_prepack_* only (section 3b) */
  SLM_W8_AWQ = 3
} slm_w4_format;
#define SLM_W4_FORMAT_MASK 0xF
/* OR into `format`: the checkpoint tensors are a merged [gate | up] column-parallel weight (the
 * reference builds it by concatenating gate_proj and up_proj along N,
 * layers/linear/multi_parallel_linear.cpp:14-41; N = 2 * intermediate, N % 64 == 0).  The packed
 * form then interleaves the two halves by 32-column tile -- packed tile 2j = columns
 * [32j, 32j+32) of gate, packed tile 2j+1 = the same columns of up -- which is what lets
 * slm_w4a16_gemm apply SiLU*mul in its epilogue (SLM_W4_SILU_MUL).  Without that flag a GEMM on
 * paired weights returns the columns in the packed (interleaved) order. */
#define SLM_W4_PAIRED 0x10

SLM_API size_t slm_w4_packed_weight_bytes(int64_t K, int64_t N);
SLM_API size_t slm_w4_packed_sz_bytes(int64_t K, int64_t N, int64_t group_size);

SLM_API int slm_w4_prepack(int32_t format,            /* slm_w4_format          */
                           const int32_t* qweight, const int32_t* qzeros,
                           const void* scales,        /* [G, N] T               */
                           const int32_t* perm,       /* [K] sorted-row -> ckpt-row, or NULL */
                           int64_t K, int64_t N, int64_t group_size, int32_t dtype,
                           void* wq_out, void* sz_out, void* stream);
/* The two halves of slm_w4_prepack, for callers that hold the weights and the scale / zero-point
 * tensors at different times -- marlin::gptq_repack / awq_repack take q_weight alone
 * (marlin.h:27-35) and marlin::gptq_gemm takes scales / zeros per call (marlin.h:17-25):
 *   weights: qweight (+ perm) -> wq_out  [K*N/8 uint32, this library's layout];
 *   sz     : scales (+ qzeros) -> sz_out [G*N uint32].  qzeros == NULL means symmetric
 *            quantisation, zero = 8 (what Marlin's has_zp = false means for 4 bits). */
SLM_API int slm_w4_prepack_weights(int32_t format, const int32_t* qweight, const int32_t* perm,
                                   int64_t K, int64_t N, void* wq_out, void* stream);
SLM_API int slm_w4_prepack_sz(int32_t format, const int32_t* qzeros /* or NULL */, const void* scales,
                              int64_t K, int64_t N, int64_t group_size, int32_t dtype, void* sz_out,
                              void* stream);

/* ========================================================================== */
/* 3b. 8-bit weights (num_bits = 8 of marlin::gptq_gemm / gptq_repack /        */
/*     awq_repack, marlin.h:17-37; bits = 8 of the quantised linears,          */
/*     qlinear_awq_marlin_impl.cpp:25-26; CPU semantics construct_weights,     */
/*     qlinear_impl.cpp:21-100) as TWO int4 planes of the int4 GEMM:           */
/*       s (q - z) = (16 s)(hi - zh) + s (lo - zl),  q = 16 hi + lo            */
/*     The packed weight has 2K rows -- [0, K) high nibbles (scale 16 s,       */
/*     zero z >> 4), [K, 2K) low nibbles (scale s, zero z & 15) -- and the     */
/*     GEMM reads the activations twice through its column gather:            */
/*       slm_w4_gemm_args { K = 2K, lda = the real row stride of A,            */
/*                          perm = perm2 (written by slm_w8_prepack_weights),  */
/*                          group_size = slm_w8_packed_group_size(K, g) }      */
/*     Exact in exact arithmetic; same HBM bytes as a native 8-bit kernel,     */
/*     twice the MFMA work of an int4 GEMM (csrc/w8_planes.hip).               */
/*     Checkpoint layouts: GPTQ qweight [K/4, N] (byte k%4), qzeros [G, N/4]   */
/*     (byte n%4, zero = stored + 1), AWQ qweight [K, N/4] / qzeros [G, N/4]   */
/*     with the byte order [0,2,1,3] (tests/kernels/quant_utils.py:182-184).   */
/*     Sizes: slm_w4_packed_weight_bytes(2K, N),                               */
/*            slm_w4_packed_sz_bytes(2K, N, packed group size), perm2 [2K].    */
/* ========================================================================== */
SLM_API int64_t slm_w8_packed_rows(int64_t K);
SLM_API int64_t slm_w8_packed_group_size(int64_t K, int64_t group_size); /* 0 = unsupported */
SLM_API int slm_w8_prepack_weights(int32_t format, /* SLM_W8_* (| SLM_W4_PAIRED) */
                                   const int32_t* qweight,
                                   const int32_t* perm, /* [K] act-order sorted-row -> ckpt-row, or NULL */
                                   int64_t K, int64_t N, void* wq_out, int32_t* perm2_out /* [2K] */,
                                   void* stream);
SLM_API int slm_w8_prepack_sz(int32_t format, const int32_t* qzeros /* or NULL: symmetric, zero 128 */,
                              const void* scales /* [K / group_size, N] T */, int64_t K, int64_t N,
                              int64_t group_size, int32_t dtype, void* sz_out, void* stream);

/* ========================================================================== */
/* 4. int4-weight x fp16/bf16-activation GEMM  C[M,N] = A[M,K] . dequant(W)   */
/*    replaces  marlin::gptq_gemm  src/kernels/quantization/marlin.h:17-25    */
/*              (gptq_gemm.cu:585-710) incl. permute_cols_kernel              */
/*              (gptq_gemm.cu:69-118) for act-order.                          */
/*    callers   {Column,Row}ParallelQLinear{AWQ,GPTQ}MarlinImpl::forward      */
/*              qlinear_awq_marlin_impl.cpp:238,344;                          */
/*              qlinear_gptq_marlin_impl.cpp:188,310.                         */
/*    w = scale * (q - zero); fp32 accumulate; output rounded RN to T.        */
/* ========================================================================== */
typedef struct slm_w4_gemm_args {
  const void* a;        /* [M, K] T, row stride lda (elements)                */
  const void* wq;       /* packed by slm_w4_prepack                           */
  const void* sz;       /* packed scale/zero table                            */
  const int32_t* perm;  /* [K] act-order column gather for A (entries < lda; < 0 = zero column), or NULL */
  const void* bias;     /* [N] T or NULL (added after accumulation)           */
  void* c;              /* [M, N] T, row stride ldc                           */
  int64_t M, K, N;      /* K = packed rows; with perm, A may be narrower (padded act-order shard) */
  int64_t lda, ldc;
  int64_t group_size;   /* K for per-channel (-1 in the checkpoint)           */
  int32_t dtype;
  int32_t flags;        /* SLM_W4_* bits, 0 = none                            */
  void* workspace;      /* split-K partials + act-order A copy                */
  size_t workspace_bytes;
} slm_w4_gemm_args;

/* flags: leave a split-K GEMM's fp32 partial sums [splits][M][N] at the START of `workspace`
 * instead of reducing them into c (which is then NOT written): the consumer absorbs the sum --
 * slm_rms_norm_splitk does, removing one launch and one round trip of the activations per
 * row-parallel linear.  Ignored (normal reduce) when bias != NULL.  Whether a given call defers is
 * a pure function of the argument block: ask slm_w4a16_gemm_deferred_splits first. */
#define SLM_W4_DEFER_REDUCE 1
/* flags: wq / sz were packed with SLM_W4_PAIRED and c is [M, N/2] (ldc >= N/2):
 *   c[m, i] = T( silu(g) * u ),  g = T(acc[m, gate col i] + bias), u = T(acc[m, up col i] + bias)
 * -- bit-identical to the unfused sequence "GEMM into [M, N], then slm_silu_mul"
 * (kernel::act_and_mul, src/kernels/activation_kernels.cu:84, after the merged gate_up linear of
 * the MLP), minus one launch and the round trip of the [M, N] intermediate.  bias, if given, is in
 * packed column order.  Needs N % 64 == 0; cannot be combined with SLM_W4_DEFER_REDUCE. */
#define SLM_W4_SILU_MUL 2
/* flags: the call SHARES THE CHIP with kernels of another stream that must keep running next to it (the two
 * half-batch decode lanes of DESIGN.md 3.6: one lane's GEMMs run under the other lane's attention stream).  A
 * scheduling hint: the plan then avoids launch shapes whose workgroups cannot sit on a CU beside other waves --
 * the two-row-tile K-sliced stream for 33 <= M <= 64 (512 threads x 236 VGPRs: alone it is the faster kernel,
 * 89 vs 96 us per Llama-3-8B layer; beside an attention stream its workgroups wait for whole CUs, 14.2 -> 21.5 ms
 * at bs 128).  Results are correct either way; the SAME flag must be passed to the workspace / deferred-splits
 * queries (they describe the plan the call will take). */
#define SLM_W4_SHARES_CHIP 4

SLM_API size_t slm_w4a16_gemm_workspace_bytes(const slm_w4_gemm_args* a);
/* number of partial slabs the call will leave in the workspace (>= 2), or 0 when it writes c as usual */
SLM_API int32_t slm_w4a16_gemm_deferred_splits(const slm_w4_gemm_args* a);
SLM_API int slm_w4a16_gemm(const slm_w4_gemm_args* a, void* stream);

/* The M <= 4 GEMV with the RMSNorm that feeds it computed in its prologue: at batch 1 a decoder
 * layer is launch-bound, and the norm before the qkv and gate_up projections
 * (input_layernorm_ / post_attention_layernorm_, src/models/meta/llama.h:174-176) is 8 KiB of work per token.
 *   h      = T(x) + residual_in            (fp32; x given directly or as split-K slabs)
 *   a      = rms_norm(h) * weight          (what the GEMV consumes; never leaves the chip unless
 *                                           normed_out is given)
 *   residual_out = T(h)
 * followed by the GEMM of `a` exactly as slm_w4a16_gemm does it (a->a and a->lda are ignored;
 * flags as usual).  Bit-identical to slm_rms_norm / slm_rms_norm_splitk followed by
 * slm_w4a16_gemm.  Every workgroup recomputes the norm while workgroup 0 stores residual_out /
 * normed_out, hence residual_out may not overlap residual_in or x (double-buffer the residual);
 * SLM_ERR_INVALID_ARG otherwise.  SLM_ERR_UNSUPPORTED unless slm_w4a16_gemv_norm_supported(a). */
typedef struct slm_w4_norm_prologue {
  const void* x;           /* [M, K] T contiguous, or NULL when `partials` is given         */
  const float* partials;   /* [n_splits, M, K] fp32 slabs of a deferred GEMM, or NULL       */
  int32_t n_splits;
  float eps;
  const void* residual_in; /* [M, K] T or NULL (plain rms_norm)                             */
  void* residual_out;      /* [M, K] T, required with residual_in                           */
  const void* weight;      /* [K] T                                                         */
  void* normed_out;        /* optional [M, K] T copy of the normalised activations          */
} slm_w4_norm_prologue;
/* 1 when the argument block would take the GEMV path (M <= 4, supported group size, no act-order
 * permutation, the extra fp32 row fits the LDS): a pure function of the block and the tuning table */
SLM_API int32_t slm_w4a16_gemv_norm_supported(const slm_w4_gemm_args* a);
SLM_API int slm_w4a16_gemv_norm(const slm_w4_gemm_args* a, const slm_w4_norm_prologue* np,
                                void* stream);

/* Slow dequantise-to-dense helper (debug / parity): w_out [K, N] T.          */
SLM_API int slm_w4_dequant(const void* wq, const void* sz, int64_t K, int64_t N,
                           int64_t group_size, int32_t dtype, void* w_out, void* stream);

/* ========================================================================== */
/* 5. Glue ops of one decoder layer (SURVEY 8f "next" rows f1/f2).            */
/*    replaces  kernel::apply_rotary_pos_emb  src/kernels/pos_embedding_      */
/*              kernels.cu:35-121; kernel::rms_norm / rms_norm_residual       */
/*              src/kernels/layernorm_kernels.cu:15,125;                      */
/*              kernel::act_and_mul (silu) src/kernels/activation_kernels.cu  */
/*              :84.                                                          */
/* ========================================================================== */
/* residual != NULL: x := x + residual (fp32), residual := T(x), then normalise
 * (rms_norm_residual, src/layers/normalization.h:42-52). */
SLM_API int slm_rms_norm(void* out, const void* x, const void* weight, void* residual,
                         int64_t n_tokens, int64_t dim, float eps, int32_t dtype, void* stream);
/* The same with x given as the split-K partial sums a deferred GEMM left behind
 * (SLM_W4_DEFER_REDUCE): x := T(sum_s partials[s][t][:]) in the reduce kernel's own order, so the
 * result is bit-identical to "reduce, then slm_rms_norm". */
SLM_API int slm_rms_norm_splitk(void* out, const float* partials /* [n_splits, n_tokens, dim] */,
                                int32_t n_splits, const void* weight, void* residual,
                                int64_t n_tokens, int64_t dim, float eps, int32_t dtype, void* stream);
/* LayerNorm with weight and optional bias (bias == NULL: none), fp32 statistics, one rounding:
 * out = T((x - mean) * rsqrt(var + eps) * weight + bias).  replaces kernel::layer_norm
 * (src/kernels/layernorm_kernels.cu:185-256; LayerNormImpl::forward src/layers/normalization.h:86-95)
 * for the LayerNorm model families (GPT-2: BASELINE configs[0], GPT-NeoX, Bloom, MPT).  dim % 8 == 0,
 * dim <= 16384, contiguous rows. */
SLM_API int slm_layer_norm(void* out, const void* x, const void* weight, const void* bias /* or NULL */,
                           int64_t n_tokens, int64_t dim, float eps, int32_t dtype, void* stream);
/* tanh-form GELU: kind = SLM_GELU_NEW (kernel::gelu_new, GPT-2's "gelu_new") or SLM_GELU_FAST
 * (kernel::gelu_fast), src/kernels/activation_kernels.cu:20-40,111-120.  with_mul = 0: out [T, d] =
 * act(x [T, d]);  with_mul = 1: out [T, d] = T(act(x[:, :d])) * x[:, d:], x [T, 2 d]
 * (gelu_new_with_mul / gelu_fast_with_mul, activation_kernels.cu:128-145).  d % 8 == 0. */
#define SLM_GELU_NEW 0
#define SLM_GELU_FAST 1
SLM_API int slm_gelu(void* out, const void* x, int64_t n_tokens, int64_t d, int32_t kind, int32_t with_mul,
                     int32_t dtype, void* stream);
/* Rotary embedding applied in place to q and k, fused with the KV append that always follows it
 * (src/layers/attention/attention.cpp:36-42).  cos_sin row = [cos(rot/2) | sin(rot/2)] per
 * position (the reference cache layout, pos_embedding_kernels.cu:41: [max_pos, 2, rot/2]), either
 * in the activation dtype (as RotaryEmbeddingKernel builds it, pos_embedding.cpp:195-196) or
 * fp32.  cos_sin == NULL skips the rotation (alibi models); slot_ids == NULL skips the append. */
SLM_API int slm_rope_kv_append(void* q /* [T, n_heads, D] in place */, int64_t q_token_stride,
                               void* k /* [T, n_kv_heads, D] in place */, int64_t k_token_stride,
                               const void* v /* [T, n_kv_heads, D] */, int64_t v_token_stride,
                               const int32_t* positions /* [T] */,
                               const void* cos_sin /* [max_pos, rot_dim] */, int32_t cos_sin_is_f32,
                               int32_t rot_dim, int32_t interleaved,
                               const int32_t* slot_ids /* [T] or NULL */,
                               void* key_cache, void* value_cache,
                               int64_t n_tokens, int32_t n_heads, int32_t n_kv_heads,
                               int32_t head_dim, int32_t dtype, void* stream);
/* The same with q / k / v given as the split-K partial sums the fused qkv GEMM left behind
 * (SLM_W4_DEFER_REDUCE; GEMM row = [q | k | v], N = (n_heads + 2 n_kv_heads) * head_dim):
 * x := T(sum_s partials[s][t][col]) in the reduce kernel's own order, so the result is
 * bit-identical to "reduce, then slm_rope_kv_append" with one launch less.  q, k, v are OUTPUTS
 * here (q rotated, k rotated, v), k / v also go to their cache slot when slot_ids != NULL.
 * Needs cos_sin, rot_dim % 8 == 0, head_dim % 4 == 0 (SLM_ERR_UNSUPPORTED otherwise: reduce first). */
SLM_API int slm_rope_kv_append_splitk(const float* partials /* [n_splits, T, N] */, int32_t n_splits,
                                      void* q, int64_t q_token_stride, void* k, int64_t k_token_stride,
                                      void* v, int64_t v_token_stride, const int32_t* positions,
                                      const void* cos_sin, int32_t cos_sin_is_f32, int32_t rot_dim,
                                      int32_t interleaved, const int32_t* slot_ids /* or NULL */,
                                      void* key_cache, void* value_cache, int64_t n_tokens,
                                      int32_t n_heads, int32_t n_kv_heads, int32_t head_dim,
                                      int32_t dtype, void* stream);
SLM_API int slm_silu_mul(void* out /* [T, d] */, const void* x /* [T, 2d]: gate | up */,
                         int64_t n_tokens, int64_t d, int32_t dtype, void* stream);

/* ---- device-side input advance for the next decode step ----------------------
 * Replaces, for a steady decode batch (q_len = 1 per sequence), the per-step host rebuild + H2D
 * copies of Batch::prepare_model_input (engine/batch.cpp:97-255, model_runner.cpp:194-203):
 * updates the graph's static input buffers IN PLACE on the device so a captured step can be
 * replayed without touching the host.  For sequence b with len = positions[b] + 1 tokens cached:
 *     positions[b]       <- len                                   (batch.cpp:155)
 *     new_cache_slots[b] <- block_table[block_cu_lens[b] + len / B] + len % B
 *                                                 (Sequence::kv_cache_slots, sequence.cpp:303-317)
 *     kv_cu_lens[i]      <- kv_cu_lens[i] + i      for i = 0..n_seqs   (every length grows by 1)
 * The block table is persistent: the host appends a block's first-slot id only when a sequence
 * crosses a block boundary; a missing block (len / B >= blocks of b) sets *overflow_flag |= 1 and
 * clamps the slot to the sequence's last block (the step must then be discarded by the host).
 * q_cu_lens is unchanged.  block_size must be a power of two.  Bit-exact integer contract. */
SLM_API int slm_decode_advance(int32_t* positions /* [n_seqs] */, int32_t* kv_cu_lens /* [n_seqs+1] */,
                               int32_t* new_cache_slots /* [n_seqs] */,
                               const int32_t* block_table, const int32_t* block_cu_lens /* [n_seqs+1] */,
                               int32_t n_seqs, int32_t block_size, int32_t* overflow_flag /* [1], may be NULL */,
                               void* stream);

/* The same for ANY batch (prefill chunks, speculative-verify rows and decode rows mixed, sequences
 * joining or leaving between steps): the integer inputs of a step built on the device from two small
 * per-sequence arrays and the persistent block table, instead of the per-token host loops and the
 * H2D copy of the flattened table of Batch::prepare_model_input (engine/batch.cpp:97-255).
 * For sequence b with n_kv = kv_cached[b] tokens already in the cache and q = max(q_lens[b], 0) new ones:
 *     q_cu_lens[b+1]  = q_cu_lens[b]  + q                                   (batch.cpp:139-140)
 *     kv_cu_lens[b+1] = kv_cu_lens[b] + n_kv + q
 *     for j in [n_kv, n_kv + q), t = q_cu_lens[b] + (j - n_kv):
 *         positions[t]       = j                                            (batch.cpp:155)
 *         new_cache_slots[t] = block_table[block_cu_lens[b] + j / B] + j % B (sequence.cpp:303-317)
 *     commit != 0: kv_cached[b] += q afterwards         (Sequence::commit_kv_cache, batch.cpp:197)
 * Rows t in [q_cu_lens[n_seqs], n_tokens_padded) are the graph padding of batch.cpp:219-244:
 * position 0, slot 0.  A position without a block sets *overflow_flag |= 1 and is clamped to the
 * sequence's last block (slot 0 when the sequence has no block at all; the host then discards the
 * step).  More new tokens than n_tokens_padded rows sets *overflow_flag |= 2: the rows that fit are
 * written, the cache positions are NOT committed.  A sequence with q = 0 (no token budget
 * this step: the reference drops it from the batch, batch.cpp:113-117) stays in the arrays as an
 * empty row range; the attention kernels skip it.  The host keeps what only it can decide: which
 * sequences run, their token budgets (q_lens), block allocation (appending first-slot ids).
 * block_size must be a power of two.  Bit-exact integer contract; one launch, capture-safe. */
SLM_API int slm_build_step_inputs(const int32_t* q_lens /* [n_seqs] */, int32_t* kv_cached /* [n_seqs] */,
                                  const int32_t* block_table, const int32_t* block_cu_lens /* [n_seqs+1] */,
                                  int32_t n_seqs, int32_t block_size, int32_t n_tokens_padded, int32_t commit,
                                  int32_t* positions /* [n_tokens_padded] */, int32_t* q_cu_lens /* [n_seqs+1] */,
                                  int32_t* kv_cu_lens /* [n_seqs+1] */, int32_t* new_cache_slots /* [n_tokens_padded] */,
                                  int32_t* overflow_flag /* [1], may be NULL */, void* stream);

/* ========================================================================== */
/* 6. xGMI all-reduce fused with residual-add + RMSNorm    */
/*    replaces  ProcessGroupNCCL::allreduce  (ncclAllReduce SUM, in place)    */
/*              src/model_parallel/process_group.cpp:135-153                  */
/*    callers   reduce_from_model_parallel_region, model_parallel.cpp:33-44,  */
/*              from the row-parallel linears (qlinear_awq_marlin_impl.cpp    */
/*              :357-363), followed in the decoder layer by                   */
/*              kernel::rms_norm_residual (layernorm_kernels.cu:125).         */
/*                                                                            */
/* Two-shot all-reduce in ONE launch per rank over peer-mapped buffers: every */
/* rank owns ceil(M / world) consecutive ROWS of the [M, H] message; it reads */
/* its rows of every rank's partial sum over xGMI, adds them in fp32 in rank  */
/* order (so all ranks hold bit-identical results), rounds to T, optionally   */
/* applies   h = x + residual; residual = T(h); y = T(h * rsqrt(mean h^2 +    */
/* eps)) * weight   (normalization.h:42-52, same arithmetic as slm_rms_norm)  */
/* to those rows, publishes them in place in its own buffer, and after one    */
/* cross-rank flag barrier gathers every other rank's rows into `out`.  The   */
/* residual stream stays row-sharded: a rank only ever touches its own rows.  */
/* Synchronisation: per-workgroup monotonically increasing flags in           */
/* peer-writable UNCACHED signal blocks (no host involvement, no reset, graph */
/* replay safe); bounded spins (20 s) -- a peer that never arrives raises     */
/* SLM_AR_ERR_TIMEOUT in the signal block's error word instead of hanging.    */
/* ========================================================================== */
#define SLM_AR_MAX_RANKS 8
#define SLM_AR_MAX_BLOCKS 128
#define SLM_SHM_HANDLE_BYTES 64
#define SLM_AR_ERR_TIMEOUT 1

/* Host-side set-up helpers (init time only; these DO allocate / map and are not capture-safe):
 * peer-shareable device memory and its interprocess handles (hipIpcGetMemHandle /
 * hipIpcOpenMemHandle; one process per GPU).  uncached != 0 allocates fine-grained uncached
 * memory (required for signal blocks: peers write them while a local kernel polls).  The memory is
 * zero-filled.  The 64-byte handle is exchanged by the caller (torch.distributed, MPI, a pipe ...). */
SLM_API int slm_shm_alloc(void** ptr, size_t bytes, int32_t uncached);
SLM_API int slm_shm_free(void* ptr);
SLM_API int slm_shm_export(void* ptr, uint8_t handle[SLM_SHM_HANDLE_BYTES]);
SLM_API int slm_shm_import(const uint8_t handle[SLM_SHM_HANDLE_BYTES], void** ptr);
SLM_API int slm_shm_close(void* ptr);
/* Thread-per-GPU shape (all ranks in ONE process, the reference's: process_group.cpp:98-123): no
 * handles -- the raw pointers are valid everywhere once `device` may access `peer_device`.  No-op
 * (SLM_OK) when both are the same device or access is already enabled. */
SLM_API int slm_shm_enable_peer_access(int32_t device, int32_t peer_device);

/* bytes of one rank's signal block (allocate with slm_shm_alloc(..., uncached = 1)) */
SLM_API size_t slm_ar_signal_bytes(void);
/* host read of a signal block's sticky error word (synchronises the device; debugging / self-test) */
SLM_API int slm_ar_read_error(const void* own_signal, int32_t* err);

typedef struct slm_ar_args {
  int32_t rank, world;                 /* world in 2..SLM_AR_MAX_RANKS */
  void* signals[SLM_AR_MAX_RANKS];     /* signals[r]: rank r's signal block as mapped in THIS process */
  void* buffers[SLM_AR_MAX_RANKS];     /* buffers[r]: rank r's [M, H] T partial sums as mapped here;
                                          buffers[rank] is local and is overwritten (own rows) */
  void* out;                           /* [M, H] T, local: the reduced (fused: normalised) rows of
                                          every rank; may alias buffers[rank] (in-place all-reduce) */
  void* residual;                      /* fused: [M, H] T local residual stream, only this rank's
                                          rows are read and updated (ALL rows in one-shot mode:
                                          M <= world and out != buffers[rank], where every rank
                                          reduces every row itself behind a single barrier);
                                          NULL = plain all-reduce */
  const void* weight;                  /* fused: RMSNorm weight [H] T */
  float eps;
  int32_t dtype;                       /* slm_dtype */
  int64_t M, H;                        /* H % 8 == 0, H <= 16384 */
  int32_t end_barrier;                 /* != 0: also wait until every peer has finished reading this
                                          rank's buffer (needed when the SAME buffer is refilled by
                                          the next producer; alternate two buffers to avoid it) */
  int32_t reserved;
} slm_ar_args;

SLM_API int slm_allreduce(const slm_ar_args* args, void* stream);
/* Test hook: the work of ALL `world` ranks in one launch on one device (ranks[r] holds rank r's
 * argument block, every pointer local).  Runs exactly the device code of slm_allreduce; lets the
 * algorithm be verified on a single GPU. */
SLM_API int slm_allreduce_simulate(const slm_ar_args* ranks, int32_t world, void* stream);

/* ========================================================================== */
/* 7. Host policy: one lane or two for a pure-decode step (round 5)           */
/*    no reference counterpart -- the two half-batch lanes are this build's   */
/*    own schedule of the decoder stack (models/meta/llama.h:123-345 is one   */
/*    serial chain per step); DESIGN.md 3.6.  Pure host code, no device work: */
/*    ONE source of the rule for the Python mirror (decode.two_lane_split)    */
/*    and the C++ host step (slm::LlamaForCausalLMHip::lane_split).           */
/*    The rule: (a) hard conditions -- lanes_min != 0, a pure-decode batch in */
/*    the reference's graph-replay sense (q_max_seq_len == 1 and n_tokens ==  */
/*    n_seqs, model_runner.cpp:112-140), >= 64 tokens, one rank or a          */
/*    tensor-parallel rank whose reductions may run on two streams            */
/*    (tp_lanes_ok); (b) lanes_min > 0: every such batch of >= lanes_min      */
/*    tokens; (c) lanes_min < 0 (auto): a MEASUREMENT recorded for this model */
/*    geometry and batch size (slm_decode_lane_policy_record: the start-up    */
/*    probe's one-lane vs two-lane time at a context length; the nearest      */
/*    recorded length within a factor 1.5 decides), else the constants        */
/*    measured on one MI355X for Llama-3-8B shapes (96 <= T <= 256, >= 12 MiB */
/*    of K + V per sequence, KV bytes of the step >= 8 x the layer's weights).*/
/* ========================================================================== */
typedef struct slm_lane_query {
  int32_t n_tokens, n_seqs, q_max_seq_len, kv_max_seq_len;
  int32_t world_size;        /* tensor-parallel ranks                                        */
  int32_t tp_lanes_ok;       /* world_size > 1: != 0 if the reductions may run on two streams */
  int32_t lanes_min;         /* -1 auto, 0 never, N = every pure-decode batch of >= N tokens  */
  int32_t n_heads, n_kv_heads, head_dim;   /* per rank                                        */
  int64_t layer_weight_bytes;              /* packed weight bytes of one decoder layer, per rank */
  int32_t kv_elem_bytes;     /* bytes per KV element (2)                                      */
  int32_t reserved;
} slm_lane_query;
/* rows of lane 0 (a multiple of 32, about half the batch) when the step should run as two lanes, else 0 */
SLM_API int32_t slm_decode_lane_split(const slm_lane_query* q);
/* record a measurement for (geometry of q, n_tokens, kv_max_seq_len): time of the same layers as one lane
 * and as two.  Process-wide table (<= 256 entries, oldest replaced); thread-safe. */
SLM_API int slm_decode_lane_policy_record(const slm_lane_query* q, float one_lane_us, float two_lane_us);
SLM_API int slm_decode_lane_policy_clear(void);
/* 1 if a recorded measurement (not the constants) would decide q, else 0 */
SLM_API int32_t slm_decode_lane_policy_measured(const slm_lane_query* q);

#ifdef __cplusplus
}
#endif
#endif /* SLM_HIP_H_ */
