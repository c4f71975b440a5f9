// slm_attn_handler_hip.cpp -- see the header.  Host code only: every device operation goes through
// the reference-signature functions of slm_torch_shim.h (and from there through the C ABI).
#include "slm_attn_handler_hip.h"

#include "slm_torch_shim.h"

namespace slm {

KVCache::KVCache(int64_t n_blocks, int64_t block_size, int64_t n_kv_heads, int64_t head_dim,
                 const torch::TensorOptions& options)
    : block_size_(block_size) {
  // kv_cache.cpp:21-27: one [n_blocks * block_size, n_kv_heads, head_dim] tensor each
  key_cache_ = torch::empty({n_blocks * block_size, n_kv_heads, head_dim}, options);
  value_cache_ = torch::empty_like(key_cache_);
}

void KVCache::set_kv_cache(const torch::Tensor& slot_ids, const torch::Tensor& keys,
                           const torch::Tensor& values) {
  llm::kernel::set_kv_cache(slot_ids, keys, values, key_cache_, value_cache_);
}

HipAttnHandler::HipAttnHandler(float sm_scale, float logits_soft_cap, int64_t rotary_dim,
                               int64_t max_position, torch::Tensor inv_freq, bool interleaved,
                               const torch::TensorOptions& options)
    : sm_scale_(sm_scale), logits_soft_cap_(logits_soft_cap), rotary_dim_(rotary_dim),
      interleaved_(interleaved) {
  // pos_embedding.cpp:183-197
  const auto t = torch::arange(0, max_position, 1, torch::kFloat32);
  const auto freqs = torch::einsum("i,j->ij", {t, inv_freq.to(torch::kFloat32).cpu()});
  cos_sin_cache_ = torch::cat({freqs.cos(), freqs.sin()}, /*dim=*/-1)
                       .contiguous()
                       .to(options.device(), torch::kFloat32);
}

HipAttnHandler::HipAttnHandler(float sm_scale, float logits_soft_cap,
                               torch::optional<torch::Tensor> alibi_slopes)
    : sm_scale_(sm_scale), logits_soft_cap_(logits_soft_cap), alibi_slopes_(std::move(alibi_slopes)) {}

void HipAttnHandler::reserve(int64_t max_tokens, int64_t n_heads, int64_t head_dim) {
  workspace_bytes_ = llm::paged_kv_varlen_mha_workspace_size(max_tokens, n_heads, head_dim);
}

void HipAttnHandler::set_workspace(const torch::Tensor& workspace) {
  llm::paged_kv_varlen_mha_set_workspace(workspace);
}

std::tuple<torch::Tensor, torch::Tensor> HipAttnHandler::apply_pos_emb(const torch::Tensor& query,
                                                                       const torch::Tensor& key,
                                                                       const torch::Tensor& positions) {
  // for alibi models no table is registered (scale_attn_handler.cpp:36-41)
  if (positions.defined() && cos_sin_cache_.defined()) {
    pending_query_ = query;  // rotated in place by the append_kv_cache() that follows
    pending_positions_ = positions;
  }
  return {query, key};
}

void HipAttnHandler::append_kv_cache(KVCache& kv_cache, const torch::Tensor& key,
                                     const torch::Tensor& value, const InputParameters& input_params) {
  torch::Tensor q = std::move(pending_query_), pos = std::move(pending_positions_);
  pending_query_ = torch::Tensor();
  pending_positions_ = torch::Tensor();
  torch::Tensor k = key;
  if (q.defined()) {
    torch::Tensor none;
    if (kv_cache.empty()) {  // profiling run: rotate only
      llm::kernel::apply_rotary_pos_emb(q, k, pos, cos_sin_cache_, static_cast<int>(rotary_dim_),
                                        interleaved_);
      return;
    }
    auto [kc, vc] = kv_cache.get_kv_cache();
    llm::kernel::apply_rotary_pos_emb_and_append(q, k, value, pos, cos_sin_cache_,
                                                 static_cast<int>(rotary_dim_), interleaved_,
                                                 input_params.new_cache_slots, kc, vc);
    return;
  }
  if (!kv_cache.empty()) kv_cache.set_kv_cache(input_params.new_cache_slots, key, value);
}

void HipAttnHandler::batch_decode(const torch::Tensor& query, const KVCache& kv_cache,
                                  const InputParameters& input_params, int32_t sliding_window,
                                  torch::Tensor& output) {
  auto [key_cache, value_cache] = kv_cache.get_kv_cache();
  llm::paged_kv_varlen_mha(output, query, key_cache, value_cache, input_params.q_cu_seq_lens,
                           input_params.kv_cu_seq_lens, input_params.block_tables,
                           input_params.cu_block_lens, alibi_slopes_,
                           static_cast<int64_t>(kv_cache.block_size()), input_params.q_max_seq_len,
                           input_params.kv_max_seq_len, sm_scale_, logits_soft_cap_, sliding_window);
}

AttentionImpl::AttentionImpl(int64_t n_heads, int64_t n_kv_heads, int64_t head_dim,
                             AttentionHandler* handler, int32_t sliding_window)
    : n_heads_(n_heads), n_kv_heads_(n_kv_heads), head_dim_(head_dim), handler_(handler),
      sliding_window_(sliding_window) {
  TORCH_CHECK(handler_ != nullptr);
  TORCH_CHECK(n_heads % n_kv_heads == 0, "n_heads ", n_heads, " not divisible by n_kv_heads ", n_kv_heads);
}

torch::Tensor AttentionImpl::forward(const torch::Tensor& query, const torch::Tensor& key,
                                     const torch::Tensor& value, const torch::Tensor& positions,
                                     KVCache& kv_cache, const InputParameters& input_params) {
  const int64_t n_tokens = query.size(0);
  auto q = query.view({n_tokens, n_heads_, head_dim_});
  auto k = key.view({n_tokens, n_kv_heads_, head_dim_});
  auto v = value.view({n_tokens, n_kv_heads_, head_dim_});
  std::tie(q, k) = handler_->apply_pos_emb(q, k, positions);
  handler_->append_kv_cache(kv_cache, k, v, input_params);
  auto output = torch::empty({n_tokens, n_heads_, head_dim_}, query.options());
  handler_->batch_decode(q, kv_cache, input_params, sliding_window_, output);
  return output.view({n_tokens, -1});
}

}  // namespace slm
