// slm_qlinear_hip.cpp -- see slm_qlinear_hip.h.  Host code only.
#include "slm_qlinear_hip.h"

#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>

#include <algorithm>
#include <cctype>

namespace slm {

// ---------------------------------------------------------------------------------------------
// StateDict (model_loader/state_dict.cpp semantics: get_tensor, chunk-sharding, select)
// ---------------------------------------------------------------------------------------------
torch::Tensor StateDict::get_tensor(const std::string& tensor_name) const {
  const auto it = dict_.find(tensor_name);
  return it == dict_.end() ? torch::Tensor() : it->second;
}

torch::Tensor StateDict::get_sharded_tensor(const std::string& tensor_name, int64_t dim, int rank,
                                            int world_size) const {
  TORCH_CHECK(dim == 0 || dim == 1, "sharding dim must be 0 or 1");
  auto t = get_tensor(tensor_name);
  if (!t.defined() || world_size <= 1) return t;
  TORCH_CHECK(t.size(dim) % world_size == 0, "can't divide tensor ", tensor_name, " evenly on dim ", dim,
              " with world_size ", world_size);
  return t.chunk(world_size, dim)[rank];
}

StateDict StateDict::select(const std::string& prefix) const {
  std::unordered_map<std::string, torch::Tensor> sel;
  for (const auto& kv : dict_)
    if (kv.first.compare(0, prefix.size(), prefix) == 0) sel[kv.first.substr(prefix.size())] = kv.second;
  return StateDict(std::move(sel), prefix_ + prefix);
}

StateDict StateDict::select_with_transform(const std::string& prefix, TensorTransform transform_func) const {
  std::unordered_map<std::string, torch::Tensor> sel;
  for (const auto& kv : dict_)
    if (kv.first.compare(0, prefix.size(), prefix) == 0) {
      const std::string name = kv.first.substr(prefix.size());
      sel[name] = transform_func ? transform_func(name, kv.second) : kv.second;
    }
  return StateDict(std::move(sel), prefix_ + prefix);
}

// ---------------------------------------------------------------------------------------------
// common part
// ---------------------------------------------------------------------------------------------
namespace {
bool iequals(std::string a, const char* b) {
  std::transform(a.begin(), a.end(), a.begin(), [](unsigned char c) { return std::tolower(c); });
  return a == b;
}
bool is_awq(const QuantArgs& qa) { return iequals(qa.quant_method(), "awq") || iequals(qa.quant_method(), "gemm"); }

// check_awq_quant_args / check_gptq_quant_args (qlinear_awq_marlin_impl.cpp:21-32,
// qlinear_gptq_marlin_impl.cpp:18-29): 4 and 8 bits, as the reference.  GPTQ: the reference's
// Marlin impl requires is_sym; these kernels also take stored zero points, so asymmetric GPTQ
// checkpoints load as well.
void check_quant_args(const QuantArgs& qa) {
  const bool awq = is_awq(qa);
  TORCH_CHECK(awq || iequals(qa.quant_method(), "gptq"), "Unsupported quant method: ", qa.quant_method());
  if (awq) TORCH_CHECK(qa.zero_point() && !qa.is_sym(), "Only zero_point is supported for AWQ");
  TORCH_CHECK(qa.bits() == 4 || qa.bits() == 8, "Only 4 and 8 bits are supported, got bits = ", qa.bits());
  const auto gs = qa.group_size();
  TORCH_CHECK(gs == -1 || gs == 32 || gs == 64 || gs == 128,
              "Only group_size of -1, 32, 64, 128 are supported, got ", gs);
}
}  // namespace

QLinearHipBase::QLinearHipBase(int64_t in_features, int64_t out_features, bool bias,
                               const QuantArgs& quant_args, const ParallelArgs& parallel_args,
                               const torch::TensorOptions& options)
    : in_features_(in_features), out_features_(out_features), quant_args_(quant_args),
      parallel_args_(parallel_args), options_(options), has_bias_(bias) {
  check_quant_args(quant_args);
  awq_ = is_awq(quant_args);
  TORCH_CHECK(parallel_args.world_size() >= 1 && parallel_args.rank() >= 0 &&
              parallel_args.rank() < parallel_args.world_size(), "bad ParallelArgs");
}

void QLinearHipBase::load_one(const StateDict& sd, const std::string& name, int64_t dim, torch::Tensor& dst,
                              bool& loaded) {
  // WeightUtils::load_sharded_weight (weight_utils.h:60-66): absent tensors are skipped -- a
  // checkpoint is spread over several files and every file is offered to every layer
  const auto t = dim < 0 ? sd.get_tensor(name)
                         : sd.get_sharded_tensor(name, dim, parallel_args_.rank(), parallel_args_.world_size());
  if (!t.defined()) return;
  TORCH_CHECK(!loaded, "weight ", sd.prefix() + name, " is loaded twice");
  dst = t.to(options_.device()).contiguous();
  loaded = true;
  packed_.reset();
}

void QLinearHipBase::load_fused(const StateDict& sd, const std::vector<std::string>& prefixes,
                                const std::string& name, int64_t dim, std::vector<torch::Tensor>& parts,
                                torch::Tensor& dst, bool& loaded) {
  // WeightUtils::load_fused_weight (weight_utils.h:48-58): one shard per prefix, kept until all
  // prefixes have arrived, then concatenated along `dim`
  if (parts.size() < prefixes.size()) parts.resize(prefixes.size());
  for (size_t i = 0; i < prefixes.size(); ++i) {
    const auto t = sd.get_sharded_tensor(prefixes[i] + name, dim, parallel_args_.rank(),
                                         parallel_args_.world_size());
    if (!t.defined()) continue;
    TORCH_CHECK(!parts[i].defined(), "weight ", sd.prefix() + prefixes[i] + name, " is loaded twice");
    parts[i] = t.to(options_.device());
  }
  if (std::all_of(parts.begin(), parts.end(), [](const torch::Tensor& t) { return t.defined(); })) {
    dst = torch::cat(parts, dim).contiguous();
    parts.clear();
    loaded = true;
    packed_.reset();
  }
}

void QLinearHipBase::verify_loaded_weights(const std::string& prefix) const {
  TORCH_CHECK(qweight_is_loaded_ || packed_, "qweight is not loaded for ", prefix + "qweight");
  TORCH_CHECK(qzeros_is_loaded_ || packed_, "qzeros is not loaded for ", prefix + "qzeros");
  TORCH_CHECK(scales_is_loaded_ || packed_, "scales is not loaded for ", prefix + "scales");
  TORCH_CHECK(!has_bias_ || bias_is_loaded_, "bias is not loaded for ", prefix + "bias");
  if (!awq_ && quant_args_.desc_act())
    TORCH_CHECK(g_idx_is_loaded_ || packed_, "g_idx is not loaded for ", prefix + "g_idx");
}

void QLinearHipBase::ensure_packed() {
  if (packed_) return;  // repack at the first call, like the reference (weight_repacked_)
  verify_loaded_weights();
  const int64_t per = 32 / quant_args_.bits();  // values per int32
  const int64_t K = awq_ ? qweight_.size(0) : qweight_.size(0) * per;
  const int64_t N = awq_ ? qweight_.size(1) * per : qweight_.size(1);
  TORCH_CHECK(K == local_in_ && N == local_out_, "loaded qweight is [", K, ", ", N, "], expected [",
              local_in_, ", ", local_out_, "]");
  const int64_t gs = quant_args_.group_size() > 0 ? quant_args_.group_size() : K;
  std::optional<torch::Tensor> gi;
  if (g_idx_.defined() && g_idx_.numel() > 0) gi = g_idx_;
  packed_ = std::make_unique<W4Linear>(awq_ ? "awq" : "gptq", qweight_, qzeros_,
                                       scales_.to(options_.dtype()), gi, gs, quant_args_.bits(), paired_);
  // the checkpoint-format shards are no longer needed
  qweight_ = torch::Tensor(); qzeros_ = torch::Tensor(); scales_ = torch::Tensor(); g_idx_ = torch::Tensor();
  if (has_bias_) bias_ = bias_.to(options_.dtype()).contiguous();
}

W4Linear& QLinearHipBase::packed() {
  ensure_packed();
  return *packed_;
}

torch::Tensor QLinearHipBase::gemm(const torch::Tensor& input, const std::optional<torch::Tensor>& bias) {
  ensure_packed();
  // (a paired weight through the plain interface returns its columns in packed order: only the
  // fused decoder layer sets it, and it calls forward_into)
  return packed_->forward(input, bias);
}

// ---------------------------------------------------------------------------------------------
// column parallel: Y = X [A_1 .. A_p] + b, A sharded along its second dimension
// ---------------------------------------------------------------------------------------------
ColumnParallelQLinearHipImpl::ColumnParallelQLinearHipImpl(int64_t in_features, int64_t out_features,
                                                           bool bias, const QuantArgs& quant_args,
                                                           bool gather_output,
                                                           const ParallelArgs& parallel_args,
                                                           const torch::TensorOptions& options)
    : QLinearHipBase(in_features, out_features, bias, quant_args, parallel_args, options),
      gather_output_(gather_output) {
  const int64_t world = parallel_args.world_size();
  TORCH_CHECK(out_features % world == 0, "out_features ", out_features, " not divisible by world_size ", world);
  local_in_ = in_features;
  local_out_ = out_features / world;
  // qlinear_awq_marlin_impl.cpp:150-151 asks N % 64, K % 128; the MFMA layout needs N % 32, K % 128
  TORCH_CHECK(local_out_ % 32 == 0 && in_features % 128 == 0, "int4 shapes: N per rank % 32, K % 128");
  if (quant_args.group_size() > 0) TORCH_CHECK(in_features % quant_args.group_size() == 0);
}

void ColumnParallelQLinearHipImpl::load_state_dict(const StateDict& sd) {
  load_one(sd, "qweight", 1, qweight_, qweight_is_loaded_);
  load_one(sd, "qzeros", 1, qzeros_, qzeros_is_loaded_);
  load_one(sd, "scales", 1, scales_, scales_is_loaded_);
  if (!awq_ && quant_args_.desc_act()) load_one(sd, "g_idx", -1, g_idx_, g_idx_is_loaded_);
  if (has_bias_) load_one(sd, "bias", 0, bias_, bias_is_loaded_);
}

void ColumnParallelQLinearHipImpl::load_state_dict(const StateDict& sd, const std::vector<std::string>& prefixes) {
  TORCH_CHECK(quant_args_.can_be_fused(), "act-order (desc_act) weights can't be fused");
  load_fused(sd, prefixes, "qweight", 1, qweight_list_, qweight_, qweight_is_loaded_);
  load_fused(sd, prefixes, "qzeros", 1, qzeros_list_, qzeros_, qzeros_is_loaded_);
  load_fused(sd, prefixes, "scales", 1, scales_list_, scales_, scales_is_loaded_);
  if (has_bias_) load_fused(sd, prefixes, "bias", 0, bias_list_, bias_, bias_is_loaded_);
}

torch::Tensor ColumnParallelQLinearHipImpl::forward(torch::Tensor input) {
  auto out = gemm(input, has_bias_ ? std::optional<torch::Tensor>(bias_) : std::nullopt);
  auto* pg = parallel_args_.process_group();
  if (parallel_args_.world_size() > 1 && gather_output_ && pg != nullptr) {
    // gather_from_model_parallel_region (model_parallel.cpp:13-31): all-gather, concat on the last dim
    std::vector<torch::Tensor> parts;
    for (int r = 0; r < parallel_args_.world_size(); ++r) parts.push_back(torch::empty_like(out));
    pg->allgather(out.contiguous(), parts);
    out = torch::cat(parts, /*dim=*/-1).contiguous();
  }
  return out;
}

// ---------------------------------------------------------------------------------------------
// row parallel: Y = sum_r X_r A_r + b, A sharded along its first dimension
// ---------------------------------------------------------------------------------------------
RowParallelQLinearHipImpl::RowParallelQLinearHipImpl(int64_t in_features, int64_t out_features, bool bias,
                                                     const QuantArgs& quant_args, bool input_is_parallelized,
                                                     const ParallelArgs& parallel_args,
                                                     const torch::TensorOptions& options)
    : QLinearHipBase(in_features, out_features, bias, quant_args, parallel_args, options),
      input_is_parallelized_(input_is_parallelized) {
  const int64_t world = parallel_args.world_size();
  TORCH_CHECK(in_features % world == 0, "in_features ", in_features, " not divisible by world_size ", world);
  local_in_ = in_features / world;
  local_out_ = out_features;
  TORCH_CHECK(local_out_ % 32 == 0 && local_in_ % 128 == 0, "int4 shapes: N % 32, K per rank % 128");
  if (quant_args.group_size() > 0)
    TORCH_CHECK(local_in_ % quant_args.group_size() == 0, "K per rank must hold whole scale groups");
  // act-order rows of one K shard reference groups all over K: like the reference
  // (qlinear_gptq_marlin_impl.cpp:236-243 load_full_scales_) every rank then keeps the FULL scale /
  // zero tables next to its rows and its slice of g_idx, and W4Linear packs the shard with padded
  // groups (the HIP path's form of Marlin's is_k_full = false, :319)
  full_scales_ = !awq_ && quant_args.desc_act() && world > 1;
}

void RowParallelQLinearHipImpl::load_state_dict(const StateDict& sd) {
  const bool grouped = quant_args_.group_size() > 0;
  load_one(sd, "qweight", 0, qweight_, qweight_is_loaded_);
  // per-channel scales (group_size -1) are one row: every rank keeps it whole
  load_one(sd, "qzeros", grouped && !full_scales_ ? 0 : -1, qzeros_, qzeros_is_loaded_);
  load_one(sd, "scales", grouped && !full_scales_ ? 0 : -1, scales_, scales_is_loaded_);
  if (!awq_ && quant_args_.desc_act()) load_one(sd, "g_idx", 0, g_idx_, g_idx_is_loaded_);
  if (has_bias_) load_one(sd, "bias", -1, bias_, bias_is_loaded_);  // added once, after the reduction
}

void RowParallelQLinearHipImpl::load_state_dict(const StateDict&, const std::vector<std::string>&) {
  TORCH_CHECK(false, "row-parallel linears are never fused (parallel_linear.h:28-33)");
}

torch::Tensor RowParallelQLinearHipImpl::forward(torch::Tensor input) {
  const int world = parallel_args_.world_size();
  if (!input_is_parallelized_ && world > 1) {
    // scatter_to_model_parallel_region (model_parallel.cpp:46-65): local split of the last dim
    TORCH_CHECK(input.size(-1) % world == 0);
    input = input.chunk(world, /*dim=*/-1)[parallel_args_.rank()].contiguous();
  }
  auto* pg = parallel_args_.process_group();
  if (world > 1) {
    auto out = gemm(input, std::nullopt);
    // process_group == nullptr with world_size > 1: the caller owns the reduction (the fused xGMI
    // all-reduce + residual + RMSNorm, slm::FusedAllReduce) and gets this rank's PARTIAL sums
    if (pg == nullptr) {
      TORCH_CHECK(!has_bias_, "partial sums requested from a row-parallel linear with a bias");
      return out;
    }
    pg->allreduce(out);  // reduce_from_model_parallel_region (model_parallel.cpp:33-44)
    if (has_bias_) out.add_(bias_);  // bias AFTER the reduction (qlinear_awq_marlin_impl.cpp:357-363)
    return out;
  }
  return gemm(input, has_bias_ ? std::optional<torch::Tensor>(bias_) : std::nullopt);
}

// ---------------------------------------------------------------------------------------------
std::shared_ptr<ParallelLinearImpl> create_column_parallel_qlinear(
    int64_t in_features, int64_t out_features, bool bias, bool gather_output, const QuantArgs& quant_args,
    const ParallelArgs& parallel_args, const torch::TensorOptions& options) {
  check_quant_args(quant_args);  // "Unsupported quant method" for anything but gptq / awq / GEMM
  return std::make_shared<ColumnParallelQLinearHipImpl>(in_features, out_features, bias, quant_args,
                                                        gather_output, parallel_args, options);
}

std::shared_ptr<ParallelLinearImpl> create_row_parallel_qlinear(
    int64_t in_features, int64_t out_features, bool bias, bool input_is_parallelized,
    const QuantArgs& quant_args, const ParallelArgs& parallel_args, const torch::TensorOptions& options) {
  check_quant_args(quant_args);
  return std::make_shared<RowParallelQLinearHipImpl>(in_features, out_features, bias, quant_args,
                                                     input_is_parallelized, parallel_args, options);
}

}  // namespace slm
