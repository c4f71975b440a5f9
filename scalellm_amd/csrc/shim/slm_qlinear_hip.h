// slm_qlinear_hip.h -- the int4 linear LAYERS of the reference on the HIP kernels:
//
//   llm::ParallelLinearImpl                          src/layers/linear/parallel_linear.h:17-37
//   {Column,Row}ParallelQLinear{AWQ,GPTQ}MarlinImpl   src/layers/quantization/
//                                                     qlinear_awq_marlin_impl.{h,cpp}:128-366,
//                                                     qlinear_gptq_marlin_impl.{h,cpp}:74-330
//   create_{column,row}_parallel_qlinear              src/layers/linear/parallel_linear.cpp:103-165
//
// ColumnParallelQLinearHipImpl / RowParallelQLinearHipImpl are what the factory returns for
// quant_method "awq" / "gptq" in an MI355X build: same constructor arguments, same virtual
// interface (forward, load_state_dict x2, verify_loaded_weights), same TP sharding dims
// (column: qweight / qzeros / scales on dim 1, bias on dim 0; row: on dim 0, bias whole), same
// lazy repack on the first forward (inside the engine's warm-up, before graph capture), same
// "GEMM, all-reduce, THEN bias" order for row-parallel (qlinear_awq_marlin_impl.cpp:357-363).
// They own this library's packed layout through slm::W4Linear -- no Marlin permutes on the host.
//
// The reference's Module / StateDict / ParallelArgs / QuantArgs headers pull in glog, folly and
// boost, which this image does not have, so the few members these classes touch are mirrored
// below under the SAME names and signatures (state_dict.h:28-40, parallel_args.h, quant_args.h);
// inside a ScaleLLM tree the four `slm::` types are replaced by `using llm::...` and
// ParallelLinearImpl by the reference's -- the class bodies compile
// unchanged.
#pragma once
#include <torch/torch.h>

#include <functional>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "slm_torch_shim.h"

namespace slm {

#define SLM_ARG(T, name)                                              \
 public:                                                              \
  const T& name() const noexcept { return name##_; }                  \
  T& name() noexcept { return name##_; }                              \
  auto name(const T& v) -> decltype(*this)& { name##_ = v; return *this; } \
                                                                      \
 private:                                                             \
  T name##_

struct QuantArgs {  // layers/quantization/quant_args.h:10-33
  SLM_ARG(std::string, quant_method);
  SLM_ARG(int64_t, bits) = 0;
  SLM_ARG(int64_t, group_size) = 0;
  SLM_ARG(bool, desc_act) = false;
  SLM_ARG(bool, is_sym) = false;
  SLM_ARG(bool, zero_point) = false;

 public:
  bool can_be_fused() const { return quant_method().empty() || !desc_act(); }
};

struct ParallelArgs {  // model_parallel/parallel_args.h
  ParallelArgs(int32_t rank, int32_t world_size, ProcessGroup* process_group)
      : rank_(rank), world_size_(world_size), process_group_(process_group) {}
  SLM_ARG(int32_t, rank) = 0;
  SLM_ARG(int32_t, world_size) = 0;

 public:
  ProcessGroup* process_group() const { return process_group_; }

 private:
  ProcessGroup* process_group_ = nullptr;
};
#undef SLM_ARG

class StateDict final {  // model_loader/state_dict.h:28-40 (the loading half is the reference's)
 public:
  explicit StateDict(std::unordered_map<std::string, torch::Tensor> dict, std::string prefix = "")
      : dict_(std::move(dict)), prefix_(std::move(prefix)) {}
  // undefined tensor if absent
  torch::Tensor get_tensor(const std::string& tensor_name) const;
  // the rank-th of world_size equal chunks along dim
  torch::Tensor get_sharded_tensor(const std::string& tensor_name, int64_t dim, int rank,
                                   int world_size) const;
  // tensors whose name starts with prefix, renamed to the suffix
  StateDict select(const std::string& prefix) const;
  // ... each passed through transform_func(suffix name, tensor) (state_dict.h:44-47)
  using TensorTransform = std::function<torch::Tensor(const std::string&, const torch::Tensor&)>;
  StateDict select_with_transform(const std::string& prefix, TensorTransform transform_func) const;
  size_t size() const { return dict_.size(); }
  const std::string& prefix() const { return prefix_; }

 private:
  std::unordered_map<std::string, torch::Tensor> dict_;
  std::string prefix_;
};

class ParallelLinearImpl {  // parallel_linear.h:17-37
 public:
  virtual ~ParallelLinearImpl() = default;
  virtual torch::Tensor forward(torch::Tensor input) = 0;
  virtual void load_state_dict(const StateDict& state_dict) = 0;
  virtual void verify_loaded_weights(const std::string& prefix = "") const = 0;
  // fused layers (qkv, gate_up): one checkpoint tensor set per prefix, concatenated on dim 1
  virtual void load_state_dict(const StateDict& state_dict, const std::vector<std::string>& prefixes) = 0;
};

class QLinearHipBase : public ParallelLinearImpl {
 public:
  void verify_loaded_weights(const std::string& prefix = "") const override;
  // checkpoint-format tensors of THIS rank's shard (testing)
  torch::Tensor qweight() const { return qweight_; }
  // The merged gate|up projection of the MLP: pack it paired so the SiLU*mul of
  // kernel::act_and_mul runs in the GEMM epilogue (before the first forward)
  void set_act_mul_silu() { TORCH_CHECK(!packed_, "set_act_mul_silu after the repack"); paired_ = true; }
  // the packed weight (repacked on first use, like forward) -- for the fused calls of a decoder
  // layer (W4Linear::forward_into: deferred split-K, SiLU*mul epilogue, caller-owned scratch)
  W4Linear& packed();
  bool has_bias() const { return has_bias_; }

 protected:
  QLinearHipBase(int64_t in_features, int64_t out_features, bool bias, const QuantArgs& quant_args,
                 const ParallelArgs& parallel_args, const torch::TensorOptions& options);
  // one tensor of one layer; dim < 0 = replicated
  void load_one(const StateDict& sd, const std::string& name, int64_t dim, torch::Tensor& dst, bool& loaded);
  void load_fused(const StateDict& sd, const std::vector<std::string>& prefixes, const std::string& name,
                  int64_t dim, std::vector<torch::Tensor>& parts, torch::Tensor& dst, bool& loaded);
  torch::Tensor gemm(const torch::Tensor& input, const std::optional<torch::Tensor>& bias);
  void ensure_packed();

  int64_t in_features_, out_features_;
  QuantArgs quant_args_;
  ParallelArgs parallel_args_;
  torch::TensorOptions options_;
  bool awq_ = false, has_bias_ = false, paired_ = false;
  // checkpoint-format shards (released once repacked)
  torch::Tensor qweight_, qzeros_, scales_, g_idx_, bias_;
  bool qweight_is_loaded_ = false, qzeros_is_loaded_ = false, scales_is_loaded_ = false,
       g_idx_is_loaded_ = false, bias_is_loaded_ = false;
  std::vector<torch::Tensor> qweight_list_, qzeros_list_, scales_list_, bias_list_;
  std::unique_ptr<W4Linear> packed_;  // built on the first forward (weight_repacked_)
  int64_t local_in_ = 0, local_out_ = 0;
};

class ColumnParallelQLinearHipImpl : public QLinearHipBase {
 public:
  ColumnParallelQLinearHipImpl(int64_t in_features, int64_t out_features, bool bias,
                               const QuantArgs& quant_args, bool gather_output,
                               const ParallelArgs& parallel_args, const torch::TensorOptions& options);
  torch::Tensor forward(torch::Tensor input) override;
  void load_state_dict(const StateDict& state_dict) override;
  void load_state_dict(const StateDict& state_dict, const std::vector<std::string>& prefixes) override;

 private:
  bool gather_output_;
};

class RowParallelQLinearHipImpl : public QLinearHipBase {
 public:
  RowParallelQLinearHipImpl(int64_t in_features, int64_t out_features, bool bias,
                            const QuantArgs& quant_args, bool input_is_parallelized,
                            const ParallelArgs& parallel_args, const torch::TensorOptions& options);
  torch::Tensor forward(torch::Tensor input) override;
  void load_state_dict(const StateDict& state_dict) override;
  void load_state_dict(const StateDict& state_dict, const std::vector<std::string>& prefixes) override;

 private:
  bool input_is_parallelized_;
  bool full_scales_ = false;  // act-order shard: scales / qzeros loaded whole (load_full_scales_)
};

// parallel_linear.cpp:103-165: "gptq" / "awq" / "GEMM" (case-insensitive) -> the HIP impls
std::shared_ptr<ParallelLinearImpl> create_column_parallel_qlinear(
    int64_t in_features, int64_t out_features, bool bias, bool gather_output, const QuantArgs& quant_args,
    const ParallelArgs& parallel_args, const torch::TensorOptions& options);
std::shared_ptr<ParallelLinearImpl> create_row_parallel_qlinear(
    int64_t in_features, int64_t out_features, bool bias, bool input_is_parallelized,
    const QuantArgs& quant_args, const ParallelArgs& parallel_args, const torch::TensorOptions& options);

}  // namespace slm
