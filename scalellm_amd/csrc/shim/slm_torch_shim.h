// slm_torch_shim.h -- libtorch adapter: the reference's C++ operator signatures on top of the
// C-ABI HIP kernel library (include/slm_hip.h).  A ScaleLLM build links this instead of
// src/kernels/attention, src/kernels/kv_cache_kernels.cu, src/kernels/pos_embedding_kernels.cu and
// src/kernels/quantization/marlin; src/layers/** then compiles unchanged.
//
// Signatures below are the reference's, verbatim in types / order / meaning:
//   llm::paged_kv_varlen_mha           src/kernels/attention/attn_api.h:12-27
//   llm::kernel::set_kv_cache          src/kernels/kv_cache_kernels.h:6-11
//   llm::kernel::apply_rotary_pos_emb  src/kernels/pos_embedding_kernels.h
// The int4 path is offered at the layer boundary the survey recommends (ParallelLinearImpl,
// src/layers/linear/parallel_linear.h:17-37): slm::W4Linear owns the MFMA-native packed layout
// and is what a QLinearHipImpl subclass wraps (INTEGRATION.md shows the 40-line subclass).
// slm::ProcessGroupRCCL implements llm::ProcessGroup's contract (process_group.h:10-60) on RCCL.
#pragma once
#include <torch/torch.h>

#include <memory>
#include <optional>
#include <vector>

namespace llm {

void paged_kv_varlen_mha(torch::Tensor& out,                // [n_tokens, n_heads, head_dim]
                         const torch::Tensor& query,        // [n_tokens, n_heads, head_dim]
                         const torch::Tensor& key_cache,    // [n_slots, n_kv_heads, head_dim]
                         const torch::Tensor& value_cache,  // [n_slots, n_kv_heads, head_dim]
                         const torch::Tensor& q_cu_lens,    // [batch + 1]
                         const torch::Tensor& kv_cu_lens,   // [batch + 1]
                         const torch::Tensor& block_table,
                         const torch::Tensor& block_cu_lens,                // [batch + 1]
                         const std::optional<torch::Tensor>& alibi_slopes,  // [n_heads]
                         int block_size, int max_q_len, int max_kv_len, float sm_scale,
                         float logits_soft_cap, int sliding_window);

// AttentionHandler::get_estimate_workspace_size / set_workspace (handler.h:40-47) hooks: the
// split-KV scratch.  Without a workspace the shim keeps one growable buffer per device.
int64_t paged_kv_varlen_mha_workspace_size(int64_t n_tokens, int64_t n_heads, int64_t head_dim);
void paged_kv_varlen_mha_set_workspace(const torch::Tensor& workspace);

namespace kernel {

void set_kv_cache(const torch::Tensor& slot_ids,  // [n_tokens]
                  const torch::Tensor& keys,      // [n_tokens, n_kv_heads, head_dim]
                  const torch::Tensor& values,    // [n_tokens, n_kv_heads, head_dim]
                  torch::Tensor& key_cache,       // [n_slots, n_kv_heads, head_dim]
                  torch::Tensor& value_cache);

void apply_rotary_pos_emb(torch::Tensor& query,            // [n_tokens, n_heads, head_dim]
                          torch::Tensor& key,              // [n_tokens, n_kv_heads, head_dim]
                          const torch::Tensor& positions,  // [n_tokens]
                          const torch::Tensor& cos_sin,    // [max_positions, 2, rotary_dim/2]
                          int rotary_dim, bool interleaved);

// fused form used by a HipAttnHandler: rope + KV append in one launch (attention.cpp:36-39)
void apply_rotary_pos_emb_and_append(torch::Tensor& query, torch::Tensor& key,
                                     const torch::Tensor& value, const torch::Tensor& positions,
                                     const torch::Tensor& cos_sin, int rotary_dim, bool interleaved,
                                     const torch::Tensor& slot_ids, torch::Tensor& key_cache,
                                     torch::Tensor& value_cache);

void rms_norm(torch::Tensor& out, torch::Tensor input, torch::Tensor weight, float epsilon);
void rms_norm_residual(torch::Tensor& out, torch::Tensor& residual, torch::Tensor input,
                       torch::Tensor weight, float epsilon);
// src/kernels/activation_kernels.h:14: act(x) * y with x = input[..., :d], y = input[..., d:]
torch::Tensor silu_with_mul(torch::Tensor input);
// the same into a caller-owned buffer (graph-captured steps reuse static buffers)
void silu_and_mul(torch::Tensor& out, torch::Tensor input);
// src/kernels/layernorm_kernels.h:22-26 (the LayerNorm model families: GPT-2, GPT-NeoX, Bloom, MPT);
// bias undefined = no bias (layernorm_kernels.cu:250)
void layer_norm(torch::Tensor& out, torch::Tensor input, torch::Tensor weight, torch::Tensor bias, float epsilon);
// src/kernels/activation_kernels.h:6-7, 12-13
torch::Tensor gelu_new(torch::Tensor input);
torch::Tensor gelu_fast(torch::Tensor input);
torch::Tensor gelu_new_with_mul(torch::Tensor input);
torch::Tensor gelu_fast_with_mul(torch::Tensor input);

}  // namespace kernel
}  // namespace llm

// The reference's int4 kernel-level boundary, src/kernels/quantization/marlin.h:17-37 -- the
// signatures verbatim, so scalellm/csrc/kernels.cu:24-54 (`_C.kernels.marlin_gemm`, `..._repack`)
// and qlinear_{awq,gptq}_marlin_impl.cpp bind against this library unchanged.  What differs is
// what the opaque tensors CARRY (the Marlin byte layout is an NVIDIA mma.m16n8k16 artefact):
//   B / out  : same shape and dtype as Marlin's ([K/16, N*16/8] int32 = K*N/8 words), holding
//              THIS library's MFMA-native layout (include/slm_hip.h section 3); only ever
//              produced by gptq_repack / awq_repack and consumed by gptq_gemm;
//   scales   : [G, N] T in plain column order -- drop the host-side marlin_permute_scales step
//              (qlinear_awq_marlin_impl.cpp:34-60);
//   zeros    : has_zp = true: [G, N/8] int32 exactly as the AWQ checkpoint stores them (drop
//              marlin_awq_to_zero_points, :62-97); has_zp = false: ignored, zero = 8 (GPTQ sym);
//   g_idx    : ignored (rows were sorted by group at repack time through `perm`);
//   perm     : [K] int32 act-order permutation (empty = none): gptq_repack sorts the weight rows
//              by it, gptq_gemm gathers the activation columns by it (gptq_gemm.cu:69-118);
//   workspace: ignored (no locks: split-K partials live in the library's own scratch);
//   num_bits : 4 or 8 (8 = two int4 planes, include/slm_hip.h section 3b); is_k_full /
//              use_fp32_reduce: accepted, the reduction is always fp32.
namespace marlin {

void gptq_gemm(const torch::Tensor& A,  // (m, k)
               const torch::Tensor& B,  // (k/16, n*16/8): this library's layout
               torch::Tensor& C,        // (m, n)
               const torch::Tensor& scales, const torch::Tensor& zeros, const torch::Tensor& g_idx,
               const torch::Tensor& perm, torch::Tensor& workspace, int num_bits, bool is_k_full,
               bool has_zp, bool use_fp32_reduce);

void gptq_repack(const torch::Tensor& q_weight,  // (k/8, n)
                 const torch::Tensor& perm,      // (k) or empty
                 torch::Tensor& out,             // (k/16, n*16/8)
                 int64_t num_bits);

void awq_repack(const torch::Tensor& q_weight,  // (k, n/8)
                torch::Tensor& out,             // (k/16, n*16/8)
                int64_t num_bits);

}  // namespace marlin

namespace slm {

// The total_kv_len the attention plan is given (slm_attn_args::total_kv_len), from what the HOST knows -- same rule
// as layers.uniform_kv_hint: the caller's value when > 0; "not uniform" (< 0) -> 0; unknown (0): a pure-decode batch
// whose flattened block table (batch.cpp:206-209) has exactly n_seqs * ceil(kv_max_seq_len / block_size) entries
// holds more than kv_max_seq_len - block_size tokens in EVERY sequence -- as uniform as the plan needs -- and is
// reported as n_seqs * kv_max_seq_len.  Anything else: 0 (the balanced partition, right for every batch).
inline int64_t uniform_kv_hint(int64_t kv_total_len, int64_t n_seqs, int64_t q_max_seq_len, int64_t kv_max_seq_len,
                               int64_t block_table_len, int64_t block_size) {
  if (kv_total_len > 0) return kv_total_len;
  if (kv_total_len < 0 || n_seqs <= 0 || q_max_seq_len > 1 || kv_max_seq_len <= 0 || block_size <= 0) return 0;
  const int64_t blocks = (kv_max_seq_len + block_size - 1) / block_size;
  return block_table_len == n_seqs * blocks ? n_seqs * kv_max_seq_len : 0;
}


// int4 linear in the library's packed layout.  Built once from CHECKPOINT-format tensors (the
// same tensors the reference's load_state_dict collects), then forward() = one GEMM launch.
// entries of the fused {scale, zero} table cache behind marlin::gptq_gemm (tests: the cache is swept
// of tables whose parameters were freed)
size_t marlin_sz_cache_entries();

class W4Linear {
 public:
  // quant_method "awq": qweight [K, N/8], qzeros [G, N/8] (AWQ interleave), scales [G, N]
  // quant_method "gptq": qweight [K/8, N], qzeros [G, N/8], scales [G, N], optional g_idx [K]
  // bits = 8: 4 values per int32 instead of 8 (byte order [0,2,1,3] for AWQ); packed as two int4
  // planes over 2K rows (include/slm_hip.h section 3b), same forward()
  // paired: the tensors are a merged [gate | up] column-parallel weight (multi_parallel_linear.cpp
  // :14-41); packed with SLM_W4_PAIRED so that forward_into(.., SLM_W4_SILU_MUL, ..) can apply
  // act_and_mul in the GEMM epilogue (4-bit, evenly grouped weights only)
  W4Linear(const std::string& quant_method, const torch::Tensor& qweight,
           const torch::Tensor& qzeros, const torch::Tensor& scales,
           const std::optional<torch::Tensor>& g_idx, int64_t group_size, int64_t bits = 4,
           bool paired = false);

  // C[M, N] = A[M, K] . dequant(W) (+ bias); `out` may be pre-allocated
  torch::Tensor forward(const torch::Tensor& input, const std::optional<torch::Tensor>& bias,
                        std::optional<torch::Tensor> out = std::nullopt) const;
  torch::Tensor dequantize() const;  // dense [K, N] (debug / parity)

  // The fused forms a decoder layer uses (csrc/shim/slm_llama_hip.cpp), scratch owned by the CALLER
  // (a step that runs two streams keeps one scratch per stream): C = A . dequant(W) with
  // flags = SLM_W4_DEFER_REDUCE (leave split-K slabs at the start of `workspace` for the consumer)
  // or SLM_W4_SILU_MUL (paired weights: c is [M, N/2] = silu(gate) * up).  Returns the number of
  // fp32 slabs left in `workspace` (>= 2: c was NOT written) or 0 (c written as usual).
  int forward_into(const torch::Tensor& a, torch::Tensor& c, int flags, const torch::Tensor& workspace) const;
  // scratch bytes forward_into needs for M rows with these flags (0 = none)
  size_t workspace_bytes(int64_t M, int flags) const;
  bool paired() const { return paired_; }

  int64_t in_features() const { return k_src_; }
  int64_t out_features() const { return N_; }

 private:
  // a row-parallel shard of an act-order checkpoint: g_idx [K] sharded, scales / qzeros the FULL
  // tables (qlinear_gptq_marlin_impl.cpp:236-243) -> padded groups, packed K_ > k_src_
  void pack_uneven_groups(const torch::Tensor& qweight, const torch::Tensor& qzeros, const torch::Tensor& scales,
                          const torch::Tensor& gi, const torch::Tensor& perm);
  torch::Tensor wq_, sz_, perm_;
  int64_t K_ = 0, N_ = 0, group_size_ = 0;
  int64_t k_src_ = 0;  // width of the activations (== K_ unless the shard was padded)
  bool paired_ = false;
  torch::ScalarType dtype_;
};

// The reference's abstract process group, same five virtuals with the same meaning
// (src/model_parallel/process_group.h:10-60): whatever drives llm::ProcessGroup* -- the parallel
// layers, Worker::process_group_test -- drives this.
class ProcessGroup {
 public:
  ProcessGroup(int rank, int world_size, const torch::Device& device)
      : rank_(rank), world_size_(world_size), device_(device) {}
  virtual ~ProcessGroup() = default;
  int rank() const { return rank_; }
  int world_size() const { return world_size_; }
  const torch::Device& device() const { return device_; }
  // in-place SUM over all ranks
  virtual void allreduce(torch::Tensor& input) const = 0;
  virtual void allgather(const torch::Tensor& input, std::vector<torch::Tensor>& outputs) const = 0;
  virtual void allgather(const torch::Tensor& input, torch::Tensor& outputs) const = 0;
  // equal splits: rank r receives slice r of every rank's input
  virtual void alltoall(const torch::Tensor& input, torch::Tensor& output) const = 0;
  // splits along dim 0 (rows); an empty list means equal splits
  virtual void alltoall(const torch::Tensor& input, torch::Tensor& output,
                        const std::vector<int64_t>& input_split_sizes,
                        const std::vector<int64_t>& output_split_sizes) const = 0;
  // one group per device, created together (process_group.h:46-49)
  static std::vector<std::unique_ptr<ProcessGroup>> create_process_groups(
      const std::vector<torch::Device>& devices);

 private:
  int rank_ = 0, world_size_ = 0;
  torch::Device device_;
};

// llm::ProcessGroupNCCL's contract over RCCL: one communicator per local GPU created together
// (ncclCommInitAll, as process_group.cpp:98-123), collectives on the CURRENT stream of the
// tensor's device so they are captured into hipGraphs with the kernels around them; all-to-all as
// grouped ncclSend / ncclRecv pairs (process_group.cpp:206-292).
class ProcessGroupRCCL : public ProcessGroup {
 public:
  static std::vector<std::unique_ptr<ProcessGroupRCCL>> create_process_groups(
      const std::vector<torch::Device>& devices);
  ~ProcessGroupRCCL() override;
  void allreduce(torch::Tensor& input) const override;
  void allgather(const torch::Tensor& input, std::vector<torch::Tensor>& outputs) const override;
  void allgather(const torch::Tensor& input, torch::Tensor& outputs) const override;
  void alltoall(const torch::Tensor& input, torch::Tensor& output) const override;
  void alltoall(const torch::Tensor& input, torch::Tensor& output, const std::vector<int64_t>& input_split_sizes,
                const std::vector<int64_t>& output_split_sizes) const override;

 private:
  ProcessGroupRCCL(int rank, int world_size, torch::Device device, void* comm)
      : ProcessGroup(rank, world_size, device), comm_(comm) {}
  void* comm_;  // ncclComm_t
};

// Tuning / test aid (the C++ twin of model_parallel.LocalShardProcessGroup, bench.py --simulate-tp): behaves
// like rank `rank` of a `world_size`-way group on ONE GPU with the collectives replaced by local stand-ins of
// the same output shape (all-reduce = no-op, all-gather = world_size copies, all-to-all = copy).  The per-rank
// COMPUTE and the weight sharding are exactly those of a real TP run; nothing measured or checked through
// it is a multi-GPU result.
class LocalShardProcessGroup : public ProcessGroup {
 public:
  LocalShardProcessGroup(int rank, int world_size, const torch::Device& device)
      : ProcessGroup(rank, world_size, device) {}
  void allreduce(torch::Tensor&) const override {}
  void allgather(const torch::Tensor& input, std::vector<torch::Tensor>& outputs) const override {
    for (auto& o : outputs) o.copy_(input);
  }
  void allgather(const torch::Tensor& input, torch::Tensor& outputs) const override {
    outputs.view({world_size(), -1}).copy_(input.reshape({1, -1}).expand({world_size(), -1}));
  }
  void alltoall(const torch::Tensor& input, torch::Tensor& output) const override { output.copy_(input); }
  void alltoall(const torch::Tensor& input, torch::Tensor& output, const std::vector<int64_t>&,
                const std::vector<int64_t>&) const override { output.copy_(input); }
};

// The two row-parallel reductions of a decoder layer as ONE launch per rank each: two-shot
// all-reduce over peer-mapped buffers fused with the residual add + RMSNorm that follows
// (slm_allreduce, include/slm_hip.h section 6).  Replaces, at the call sites of
// reduce_from_model_parallel_region (model_parallel.cpp:33-44) in the row-parallel linears, the
// pair ProcessGroup::allreduce (process_group.cpp:135-153) + kernel::rms_norm_residual
// (layernorm_kernels.cu:125).  Thread-per-GPU shape, like create_process_groups: create() is
// called once by the engine thread, rank r's object is then used from worker thread r.
class FusedAllReduce {
 public:
  static std::vector<std::shared_ptr<FusedAllReduce>> create(const std::vector<torch::Device>& devices,
                                                             int64_t max_tokens, int64_t hidden,
                                                             torch::ScalarType dtype);
  ~FusedAllReduce();
  int rank() const { return rank_; }
  int world_size() const { return world_size_; }
  // [n_tokens, hidden] view of this rank's message buffer i (0 / 1, alternate per call site): the
  // row-parallel GEMM writes its partial sums here
  torch::Tensor buffer(int i, int64_t n_tokens) const;
  // in-place SUM over ranks of buffer i  (== ProcessGroup::allreduce on that tensor)
  torch::Tensor allreduce(int i, int64_t n_tokens) const;
  // out = RMSNorm(allreduce(buffer i) + residual) * weight for ALL rows; residual is updated in
  // place on THIS rank's rows only (the residual stream stays row-sharded across ranks)
  void allreduce_residual_rmsnorm(int i, int64_t n_tokens, torch::Tensor& out, torch::Tensor& residual,
                                  const torch::Tensor& weight, float eps) const;
  int error() const;  // sticky SLM_AR_ERR_* bits of this rank's signal block (synchronises)

 private:
  struct Shared;  // the allocations of every rank, freed with the last rank object
  FusedAllReduce(int rank, int world_size, torch::Device device, std::shared_ptr<Shared> sh)
      : rank_(rank), world_size_(world_size), device_(device), sh_(std::move(sh)) {}
  int rank_, world_size_;
  torch::Device device_;
  std::shared_ptr<Shared> sh_;
};

}  // namespace slm
