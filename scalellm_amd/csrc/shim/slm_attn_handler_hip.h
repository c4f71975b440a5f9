// slm_attn_handler_hip.h -- the attention-handler layer of the reference over the HIP kernels,
// compiled C++.
//
// Mirrors, member for member, what src/layers/attention/{handler.h, scale_attn_handler.{h,cpp},
// attention.{h,cpp}} expose to the model code: AttentionHandler (the virtual interface,
// handler.h:15-48), a concrete handler over llm::paged_kv_varlen_mha, and AttentionImpl::forward
// (attention.cpp:22-46: view -> apply_pos_emb -> append_kv_cache -> batch_decode).  The reference
// headers themselves pull in glog / gflags / boost and the whole ModelArgs tree, so the three
// value types the interface mentions are restated here with the reference's member names:
//   slm::KVCache          memory/kv_cache.h:10-60      (the two cache tensors + block size)
//   slm::InputParameters  models/parameters.h:11-56    (same members, same meaning)
// A maintainer who links against the reference types instead only changes the `using` lines at
// the bottom of this header.
//
// What the HIP handler does differently from ScaleAttnHandler (and why it is a separate class):
// RoPE and the KV append are ONE launch.  AttentionImpl::forward always calls apply_pos_emb and
// append_kv_cache back to back (attention.cpp:36-39), so apply_pos_emb only records its
// arguments and append_kv_cache issues slm_rope_kv_append (rotate q and k in place, write k / v to
// their slots); with an empty cache (the profiling run, scale_attn_handler.cpp:76) the rotation
// alone runs.  The split-KV scratch goes through the interface's own workspace hooks
// (handler.h:19-23) so nothing is allocated after graph capture.
#pragma once
#include <torch/torch.h>

#include "slm_torch_shim.h"  // slm::uniform_kv_hint

#include <memory>
#include <tuple>

namespace slm {

// memory/kv_cache.h:10-60
class KVCache final {
 public:
  KVCache() = default;
  KVCache(int64_t n_blocks, int64_t block_size, int64_t n_kv_heads, int64_t head_dim,
          const torch::TensorOptions& options);
  // over tensors somebody else allocated (kv_cache.h:14: KVCache(key_cache, value_cache)); the
  // block size is not recoverable from the flat [n_slots, n_kv_heads, head_dim] shape
  KVCache(torch::Tensor key_cache, torch::Tensor value_cache, int64_t block_size)
      : block_size_(block_size), key_cache_(std::move(key_cache)), value_cache_(std::move(value_cache)) {}
  bool empty() const { return block_size_ == 0; }
  int64_t block_size() const { return block_size_; }
  std::tuple<torch::Tensor, torch::Tensor> get_kv_cache() const { return {key_cache_, value_cache_}; }
  // slot_ids [n_tokens] int32, keys / values [n_tokens, n_kv_heads, head_dim] (kv_cache.cpp:59-73)
  void set_kv_cache(const torch::Tensor& slot_ids, const torch::Tensor& keys,
                    const torch::Tensor& values);

 private:
  int64_t block_size_ = 0;
  torch::Tensor key_cache_;    // [n_blocks * block_size, n_kv_heads, head_dim] (kv_cache.cpp:21-27)
  torch::Tensor value_cache_;
};

// models/parameters.h:11-56
struct InputParameters {
  int32_t num_sequences = 0;
  torch::Tensor q_cu_seq_lens;   // [n_seq + 1] int32
  torch::Tensor kv_cu_seq_lens;  // [n_seq + 1] int32
  int32_t kv_max_seq_len = 0;
  int32_t q_max_seq_len = 0;
  torch::Tensor new_cache_slots;  // [n_tokens] int32
  torch::Tensor block_tables;     // [n_blocks] int32, first-slot ids (batch.cpp:206-209)
  torch::Tensor cu_block_lens;    // [n_seq + 1] int32
  // extension (not in models/parameters.h): cu_seq_lens.back() as Batch::prepare_model_input has it on the
  // host (batch.cpp:137), 0 = unknown -- a scheduling hint like the two maxima (slm_attn_args::total_kv_len)
  // 0 = unknown: derived from host-known sizes where they settle it (slm::uniform_kv_hint, slm_torch_shim.h: what
  // an unchanged engine gets); < 0 = known NOT to be uniform (a graph captured over padded static buffers, the
  // half of a ragged batch)
  int64_t kv_total_len = 0;
};

// handler.h:15-48
class AttentionHandler {
 public:
  virtual ~AttentionHandler() = default;
  virtual int64_t get_estimate_workspace_size() { return -1; }
  virtual void set_workspace(const torch::Tensor& /*workspace*/) {}
  virtual std::tuple<torch::Tensor, torch::Tensor> apply_pos_emb(const torch::Tensor& query,
                                                                 const torch::Tensor& key,
                                                                 const torch::Tensor& positions) = 0;
  virtual void batch_decode(const torch::Tensor& query, const KVCache& kv_cache,
                            const InputParameters& input_params, int32_t sliding_window,
                            torch::Tensor& output) = 0;
  virtual void append_kv_cache(KVCache& kv_cache, const torch::Tensor& key,
                               const torch::Tensor& value, const InputParameters& input_params) = 0;
};

class HipAttnHandler : public AttentionHandler {
 public:
  // rotary models (ScaleAttnHandler's first constructor, scale_attn_handler.cpp:11-22)
  HipAttnHandler(float sm_scale, float logits_soft_cap, int64_t rotary_dim, int64_t max_position,
                 torch::Tensor inv_freq, bool interleaved, const torch::TensorOptions& options);
  // alibi / no positional embedding (scale_attn_handler.cpp:24-30)
  HipAttnHandler(float sm_scale, float logits_soft_cap, torch::optional<torch::Tensor> alibi_slopes);

  // worst-case split-KV scratch for `max_tokens` query tokens (call before graph capture)
  void reserve(int64_t max_tokens, int64_t n_heads, int64_t head_dim);
  int64_t get_estimate_workspace_size() override { return workspace_bytes_; }
  void set_workspace(const torch::Tensor& workspace) override;

  std::tuple<torch::Tensor, torch::Tensor> apply_pos_emb(const torch::Tensor& query,
                                                         const torch::Tensor& key,
                                                         const torch::Tensor& positions) override;
  void batch_decode(const torch::Tensor& query, const KVCache& kv_cache,
                    const InputParameters& input_params, int32_t sliding_window,
                    torch::Tensor& output) override;
  void append_kv_cache(KVCache& kv_cache, const torch::Tensor& key, const torch::Tensor& value,
                       const InputParameters& input_params) override;
  // what a fused caller needs (the RoPE + append kernel that also sums the qkv GEMM's split-K slabs,
  // per-stream attention scratch: csrc/shim/slm_llama_hip.cpp)
  const torch::Tensor& cos_sin_cache() const { return cos_sin_cache_; }
  // replace the table (fp32 [max_position, rotary_dim] = cos | sin): parity tests share ONE table
  // between this class and the Python mirror (cos / sin evaluated on different devices differ in ulps)
  void set_cos_sin_cache(const torch::Tensor& t) { cos_sin_cache_ = t; }
  int64_t rotary_dim() const { return rotary_dim_; }
  bool interleaved() const { return interleaved_; }
  float sm_scale() const { return sm_scale_; }
  float logits_soft_cap() const { return logits_soft_cap_; }

 private:
  float sm_scale_ = 0.f;
  float logits_soft_cap_ = 0.f;
  torch::optional<torch::Tensor> alibi_slopes_;
  // [max_position, rotary_dim] = cos | sin, fp32 (RotaryEmbeddingKernel builds the same table in
  // the activation dtype, pos_embedding.cpp:183-197; fp32 keeps the rotation exact)
  torch::Tensor cos_sin_cache_;
  int64_t rotary_dim_ = 0;
  bool interleaved_ = false;
  int64_t workspace_bytes_ = -1;
  // recorded by apply_pos_emb, consumed by the append_kv_cache that follows it
  torch::Tensor pending_query_, pending_positions_;
};

// attention.h / attention.cpp:8-46
class AttentionImpl {
 public:
  AttentionImpl(int64_t n_heads, int64_t n_kv_heads, int64_t head_dim, AttentionHandler* handler,
                int32_t sliding_window = -1);
  // query [n_tokens, n_heads * head_dim], key / value [n_tokens, n_kv_heads * head_dim]
  torch::Tensor forward(const torch::Tensor& query, const torch::Tensor& key,
                        const torch::Tensor& value, const torch::Tensor& positions,
                        KVCache& kv_cache, const InputParameters& input_params);

 private:
  int64_t n_heads_, n_kv_heads_, head_dim_;
  AttentionHandler* handler_;
  int32_t sliding_window_;
};

}  // namespace slm
