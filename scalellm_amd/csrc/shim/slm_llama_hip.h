// slm_llama_hip.h -- the Llama decoder stack of the reference composed in C++ from the HIP layer
// classes of this directory: the HOST side of the decode hot path, compiled (north_star: "host code
// stays C++ calling HIP through a thin C-ABI kernel library").
//
// Mirrors src/models/meta/llama.h:123-345:
//   LlamaMLPImpl            gate_up (merged column-parallel) -> act_and_mul -> down (row-parallel)
//   LlamaAttentionImpl      qkv (fused column-parallel) -> Attention -> o_proj (row-parallel)
//   LlamaDecoderLayerImpl   h = x + attn(input_layernorm(x)); h + mlp(post_attention_layernorm(h))
//   LlamaModelImpl          embed_tokens -> layers -> norm
//   LlamaForCausalLMImpl    forward + logits(hidden, selected_idxes)
// built from slm::{Column,Row}ParallelQLinearHipImpl (parallel_linear.h:17-37 interface),
// slm::AttentionImpl / HipAttnHandler (attention.cpp:22-46, handler.h:15-48), llm::kernel::rms_norm /
// rms_norm_residual / silu_and_mul and slm::ProcessGroup / FusedAllReduce.
//
// Two compositions of the SAME operators, selected by Options::fused:
//   plain  the reference's call sequence, one interface call per module:
//            rms_norm[_residual] -> qkv->forward -> AttentionImpl::forward -> o->forward ->
//            rms_norm_residual -> gate_up->forward -> silu_and_mul -> down->forward
//   fused  what the decode step of an MI355X build runs (same bits as plain, fewer launches):
//            the split-K reductions of qkv / o / down are absorbed by their consumers (RoPE + append,
//            RMSNorm), SiLU*mul is the gate_up GEMM's epilogue, residual add + norm one launch, and
//            a large pure-decode batch runs as two half-batch LANES on two streams (the int4 GEMMs
//            of one half under the HBM-bound attention of the other; attention launches chained by
//            events) -- capturable as one hipGraph.
// scalellm_amd/decode.py (LlamaDecodeStep) is the Python mirror of exactly this file;
// tests/test_shim_gpu.py holds the two bit-identical.
#pragma once
#include <torch/torch.h>

#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <memory>
#include <optional>
#include <string>
#include <vector>

#include "slm_attn_handler_hip.h"
#include "slm_qlinear_hip.h"
#include "slm_torch_shim.h"

namespace slm {

struct LlamaArgs {  // the ModelArgs members llama.h reads (models/model_args.h)
  int64_t hidden_size = 4096;
  int64_t n_heads = 32;
  int64_t n_kv_heads = 8;
  int64_t head_dim = 128;
  int64_t intermediate_size = 14336;
  int64_t n_layers = 32;
  int64_t vocab_size = 128256;
  int64_t max_position_embeddings = 8192;
  float rope_theta = 500000.0f;
  float rms_norm_eps = 1e-5f;
};

class LlamaForCausalLMHip {
 public:
  struct Options {
    int64_t max_tokens = 256;    // rows of the static activation buffers
    bool fused = true;           // see the header comment
    // two-lane policy for pure-decode batches: -1 = auto (the measured batch sizes, as
    // decode.LlamaDecodeStep._lane_split), 0 = never, N = from N tokens on
    int64_t decode_lanes = -1;
    bool lanes_chain = true;     // serialise the lanes' attention launches by events
  };

  LlamaForCausalLMHip(const LlamaArgs& args, const QuantArgs& quant_args, const ParallelArgs& parallel_args,
                      const torch::TensorOptions& options, const Options& opt,
                      std::shared_ptr<FusedAllReduce> fused_allreduce = nullptr,
                      // round 5: a second, independent instance for lane 1 lets a tensor-parallel rank run two
                      // lanes (each lane's reductions meet the peers' same lane: own signal block, own buffers)
                      std::shared_ptr<FusedAllReduce> fused_allreduce_lane1 = nullptr);

  // HuggingFace names, as the reference's loaders feed them (llama.h:64-121 register_module names):
  //   model.embed_tokens.weight, model.norm.weight, lm_head.weight,
  //   model.layers.N.{input_layernorm,post_attention_layernorm}.weight,
  //   model.layers.N.self_attn.{q,k,v,o}_proj.{qweight,qzeros,scales[,g_idx]},
  //   model.layers.N.mlp.{gate,up,down}_proj.{qweight,qzeros,scales[,g_idx]}
  void load_state_dict(const StateDict& state_dict);
  void verify_loaded_weights() const;

  // scratch of every lane + the lanes' side stream: once, before graph capture
  void reserve(int64_t n_tokens);

  // LlamaForCausalLMImpl::forward: final-norm hidden states [n_tokens, hidden] (static buffer view)
  torch::Tensor forward(const torch::Tensor& tokens, const torch::Tensor& positions,
                        std::vector<KVCache>& kv_caches, const InputParameters& input_params);
  // LlamaForCausalLMImpl::logits
  torch::Tensor logits(const torch::Tensor& hidden_states, const torch::Tensor& selected_idxes);
  // one greedy decode step: forward -> logits of every sequence's last token -> argmax
  // (return_logits: the logits instead of the token ids)
  torch::Tensor decode_step(const torch::Tensor& tokens, const torch::Tensor& positions,
                            std::vector<KVCache>& kv_caches, const InputParameters& input_params,
                            bool return_logits);
  int last_lanes() const { return last_lanes_; }
  HipAttnHandler& handler() { return *handler_; }
  int64_t n_local_heads() const { return n_heads_; }
  int64_t n_local_kv_heads() const { return n_kv_heads_; }

 private:
  struct Layer {
    std::shared_ptr<ColumnParallelQLinearHipImpl> qkv, gate_up;
    std::shared_ptr<RowParallelQLinearHipImpl> o, down;
    torch::Tensor input_norm, post_norm;
  };
  struct Lane {  // a range of token rows walking the stack on one stream with its own scratch
    int idx = 0;
    FusedAllReduce* far = nullptr;   // this lane's fused all-reduce instance (TP), else nullptr
    int64_t r0 = 0, r1 = 0;
    torch::Tensor positions, resid, normed, qkv, attn, act, gate_up, o_buf, down_buf, q;
    InputParameters params;
    bool norm_pending = false;      // `normed` is not up to date: (x or slabs) + residual still to run
    torch::Tensor pend_x, pend_w;   // ... with this input (unwritten when pend_splits > 0) and weight
    int pend_splits = 0, pend_slot = 0;
    bool pend_residual = false;
    int qkv_splits = 0;
  };
  StateDict select_qkv(const StateDict& layer_sd) const;
  bool tp_lanes_ok() const;
  int64_t lane_split(int64_t T, const InputParameters& p) const;
  std::vector<Lane> make_lanes(int64_t T, const torch::Tensor& positions, const InputParameters& p);
  void run_norm(Lane& ln);
  void pre_attn(Lane& ln, size_t li, std::vector<KVCache>& kv);
  void attn(Lane& ln, size_t li, std::vector<KVCache>& kv, int phase = 0);
  void post_attn(Lane& ln, size_t li);
  void plain_layer(Lane& ln, size_t li, std::vector<KVCache>& kv);
  void reduce_add_norm(Lane& ln, int which, torch::Tensor& partial, const torch::Tensor& weight, int splits, int slot);
  void run_two_lanes(Lane& l0, Lane& l1, std::vector<KVCache>& kv);
  const torch::Tensor& scratch(int lane, size_t bytes);
  const torch::Tensor& deferred(int lane, int slot, size_t bytes);

  LlamaArgs args_;
  QuantArgs quant_args_;
  ParallelArgs parallel_args_;
  torch::TensorOptions options_;
  Options opt_;
  std::shared_ptr<FusedAllReduce> far_, far_lane1_;
  int64_t n_heads_ = 0, n_kv_heads_ = 0;
  int64_t kv_replication_ = 0;  // world_size / n_kv_heads when KV heads are replicated, else 0
  std::unique_ptr<HipAttnHandler> handler_;
  std::unique_ptr<AttentionImpl> atten_;
  std::vector<Layer> layers_;
  torch::Tensor embed_, final_norm_, lm_head_;  // embed [vocab, hidden / tp], lm_head [hidden, vocab / tp]
  bool embed_loaded_ = false, norm_loaded_ = false, lm_head_loaded_ = false;
  // static activation buffers [max_tokens, ...]
  torch::Tensor resid_, normed_, qkv_, attn_, act_, gate_up_, o_, down_;
  torch::Tensor lane_q_cu_[2], lane_kv_cu_[2];
  torch::Tensor scratch_[2], deferred_[2][2];
  std::vector<torch::Tensor> retired_;
  // lane 1's stream: per model instance, on the model's device (not per thread: two models on two GPUs
  // driven from one thread each need their own)
  std::optional<c10::hip::HIPStreamMasqueradingAsCUDA> side_stream_;
  int last_lanes_ = 1;
  int chip_flag_ = 0;          // SLM_W4_SHARES_CHIP while a two-lane step issues its GEMMs, else 0
  bool all_packed_ = false;   // every layer repacked before the first two-lane step (forward())
};

}  // namespace slm
