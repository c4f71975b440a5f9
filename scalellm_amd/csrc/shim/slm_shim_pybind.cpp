// Python test surface of the C++ shim, modelled on the reference's scalellm/csrc/kernels.cu
// (`_C.kernels`): lets pytest drive the C++ operator API exactly as a ScaleLLM layer would.
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/extension.h>

#include <algorithm>
#include <thread>

#include "slm_attn_handler_hip.h"
#include "slm_llama_hip.h"
#include "slm_qlinear_hip.h"
#include "slm_torch_shim.h"

namespace py = pybind11;

PYBIND11_MODULE(_slm_shim, m) {
  m.doc() = "libtorch shim over libslm_hip (reference C++ operator signatures)";
  m.def("paged_kv_varlen_mha",
        [](torch::Tensor out, const torch::Tensor& query, const torch::Tensor& key_cache,
           const torch::Tensor& value_cache, const torch::Tensor& q_cu_lens,
           const torch::Tensor& kv_cu_lens, const torch::Tensor& block_table,
           const torch::Tensor& block_cu_lens, const std::optional<torch::Tensor>& alibi_slopes,
           int block_size, int max_q_len, int max_kv_len, float sm_scale, float logits_soft_cap,
           int sliding_window) {
          llm::paged_kv_varlen_mha(out, query, key_cache, value_cache, q_cu_lens, kv_cu_lens,
                                   block_table, block_cu_lens, alibi_slopes, block_size, max_q_len,
                                   max_kv_len, sm_scale, logits_soft_cap, sliding_window);
        });
  m.def("set_kv_cache", [](const torch::Tensor& slot_ids, const torch::Tensor& keys,
                           const torch::Tensor& values, torch::Tensor key_cache,
                           torch::Tensor value_cache) {
    llm::kernel::set_kv_cache(slot_ids, keys, values, key_cache, value_cache);
  });
  m.def("apply_rotary_pos_emb",
        [](torch::Tensor query, torch::Tensor key, const torch::Tensor& positions,
           const torch::Tensor& cos_sin, int rotary_dim, bool interleaved) {
          llm::kernel::apply_rotary_pos_emb(query, key, positions, cos_sin, rotary_dim, interleaved);
        });
  m.def("rms_norm", [](torch::Tensor out, torch::Tensor input, torch::Tensor weight, float eps) {
    llm::kernel::rms_norm(out, input, weight, eps);
  });
  m.def("silu_and_mul",
        [](torch::Tensor out, torch::Tensor input) { llm::kernel::silu_and_mul(out, input); });
  m.def("silu_with_mul", &llm::kernel::silu_with_mul, py::arg("input"));
  m.def("layer_norm", [](torch::Tensor out, torch::Tensor input, torch::Tensor weight,
                         std::optional<torch::Tensor> bias, float eps) {
    llm::kernel::layer_norm(out, input, weight, bias.has_value() ? *bias : torch::Tensor(), eps);
  });
  m.def("gelu_new", &llm::kernel::gelu_new, py::arg("input"));
  m.def("gelu_fast", &llm::kernel::gelu_fast, py::arg("input"));
  m.def("gelu_new_with_mul", &llm::kernel::gelu_new_with_mul, py::arg("input"));
  m.def("gelu_fast_with_mul", &llm::kernel::gelu_fast_with_mul, py::arg("input"));
  // scalellm/csrc/kernels.cu:24-54, verbatim names and keyword arguments: `_C.kernels`
  m.def("marlin_sz_cache_entries", &slm::marlin_sz_cache_entries);
  m.def("marlin_gemm",
        [](const torch::Tensor& A, const torch::Tensor& B, torch::Tensor C, const torch::Tensor& scales,
           const torch::Tensor& zeros, const torch::Tensor& g_idx, const torch::Tensor& perm,
           torch::Tensor workspace, int num_bits, bool is_k_full, bool has_zp, bool use_fp32_reduce) {
          marlin::gptq_gemm(A, B, C, scales, zeros, g_idx, perm, workspace, num_bits, is_k_full, has_zp,
                            use_fp32_reduce);
        },
        "Marlin GPTQ GEMM", py::arg("A"), py::arg("B"), py::arg("C"), py::arg("scales"), py::arg("zeros"),
        py::arg("g_idx"), py::arg("perm"), py::arg("workspace"), py::arg("num_bits"), py::arg("is_k_full"),
        py::arg("has_zp"), py::arg("use_fp32_reduce"));
  m.def("marlin_gptq_repack",
        [](const torch::Tensor& q_weight, const torch::Tensor& perm, torch::Tensor out, int64_t num_bits) {
          marlin::gptq_repack(q_weight, perm, out, num_bits);
        },
        "Marlin GPTQ repack", py::arg("q_weight"), py::arg("perm"), py::arg("out"), py::arg("num_bits"));
  m.def("marlin_awq_repack",
        [](const torch::Tensor& q_weight, torch::Tensor out, int64_t num_bits) {
          marlin::awq_repack(q_weight, out, num_bits);
        },
        "Marlin AWQ repack", py::arg("q_weight"), py::arg("out"), py::arg("num_bits"));
  // the layer boundary: ParallelLinearImpl (parallel_linear.h:17-37) through its factory
  py::class_<slm::ParallelLinearImpl, std::shared_ptr<slm::ParallelLinearImpl>>(m, "ParallelLinearImpl")
      .def("forward", &slm::ParallelLinearImpl::forward)
      .def("load_state_dict",
           [](slm::ParallelLinearImpl& self, std::unordered_map<std::string, torch::Tensor> sd) {
             self.load_state_dict(slm::StateDict(std::move(sd)));
           })
      .def("load_state_dict_fused",
           [](slm::ParallelLinearImpl& self, std::unordered_map<std::string, torch::Tensor> sd,
              const std::vector<std::string>& prefixes) {
             self.load_state_dict(slm::StateDict(std::move(sd)), prefixes);
           })
      .def("verify_loaded_weights", &slm::ParallelLinearImpl::verify_loaded_weights, py::arg("prefix") = "");
  auto make_args = [](const std::string& method, int64_t bits, int64_t group_size, bool desc_act, bool is_sym,
                      bool zero_point) {
    slm::QuantArgs qa;
    qa.quant_method(method).bits(bits).group_size(group_size).desc_act(desc_act).is_sym(is_sym).zero_point(zero_point);
    return qa;
  };
  m.def("create_column_parallel_qlinear",
        [make_args](int64_t in_features, int64_t out_features, bool bias, bool gather_output,
                    const std::string& quant_method, int64_t bits, int64_t group_size, bool desc_act, bool is_sym,
                    bool zero_point, int rank, int world_size, torch::ScalarType dtype, int device_index) {
          return slm::create_column_parallel_qlinear(
              in_features, out_features, bias, gather_output,
              make_args(quant_method, bits, group_size, desc_act, is_sym, zero_point),
              slm::ParallelArgs(rank, world_size, nullptr),
              torch::dtype(dtype).device(torch::Device(torch::kCUDA, device_index)));
        });
  m.def("create_row_parallel_qlinear",
        [make_args](int64_t in_features, int64_t out_features, bool bias, bool input_is_parallelized,
                    const std::string& quant_method, int64_t bits, int64_t group_size, bool desc_act, bool is_sym,
                    bool zero_point, int rank, int world_size, torch::ScalarType dtype, int device_index) {
          return slm::create_row_parallel_qlinear(
              in_features, out_features, bias, input_is_parallelized,
              make_args(quant_method, bits, group_size, desc_act, is_sym, zero_point),
              slm::ParallelArgs(rank, world_size, nullptr),
              torch::dtype(dtype).device(torch::Device(torch::kCUDA, device_index)));
        });
  // the attention layer boundary: KVCache / InputParameters / AttentionHandler / AttentionImpl
  // (memory/kv_cache.h, models/parameters.h, layers/attention/{handler,attention}.h)
  py::class_<slm::KVCache>(m, "KVCache")
      .def(py::init([](int64_t n_blocks, int64_t block_size, int64_t n_kv_heads, int64_t head_dim,
                       torch::ScalarType dtype, int device_index) {
        return slm::KVCache(n_blocks, block_size, n_kv_heads, head_dim,
                            torch::dtype(dtype).device(torch::Device(torch::kCUDA, device_index)));
      }))
      .def(py::init<>())
      .def("empty", &slm::KVCache::empty)
      .def("block_size", &slm::KVCache::block_size)
      .def("get_kv_cache", &slm::KVCache::get_kv_cache)
      .def("set_kv_cache", &slm::KVCache::set_kv_cache);
  py::class_<slm::InputParameters>(m, "InputParameters")
      .def(py::init<>())
      .def_readwrite("num_sequences", &slm::InputParameters::num_sequences)
      .def_readwrite("q_cu_seq_lens", &slm::InputParameters::q_cu_seq_lens)
      .def_readwrite("kv_cu_seq_lens", &slm::InputParameters::kv_cu_seq_lens)
      .def_readwrite("kv_max_seq_len", &slm::InputParameters::kv_max_seq_len)
      .def_readwrite("q_max_seq_len", &slm::InputParameters::q_max_seq_len)
      .def_readwrite("new_cache_slots", &slm::InputParameters::new_cache_slots)
      .def_readwrite("block_tables", &slm::InputParameters::block_tables)
      .def_readwrite("cu_block_lens", &slm::InputParameters::cu_block_lens)
      .def_readwrite("kv_total_len", &slm::InputParameters::kv_total_len);
  py::class_<slm::HipAttnHandler>(m, "HipAttnHandler")
      .def(py::init([](float sm_scale, float logits_soft_cap, int64_t rotary_dim, int64_t max_position,
                       torch::Tensor inv_freq, bool interleaved, int device_index) {
        return std::make_unique<slm::HipAttnHandler>(
            sm_scale, logits_soft_cap, rotary_dim, max_position, std::move(inv_freq), interleaved,
            torch::TensorOptions().device(torch::Device(torch::kCUDA, device_index)));
      }))
      .def(py::init([](float sm_scale, float logits_soft_cap, std::optional<torch::Tensor> alibi_slopes) {
        return std::make_unique<slm::HipAttnHandler>(sm_scale, logits_soft_cap, std::move(alibi_slopes));
      }))
      .def("reserve", &slm::HipAttnHandler::reserve)
      .def("get_estimate_workspace_size", &slm::HipAttnHandler::get_estimate_workspace_size)
      .def("set_workspace", &slm::HipAttnHandler::set_workspace);
  // AttentionImpl::forward through the virtual AttentionHandler interface
  m.def("attention_forward",
        [](slm::HipAttnHandler& handler, int64_t n_heads, int64_t n_kv_heads, int64_t head_dim,
           int32_t sliding_window, const torch::Tensor& query, const torch::Tensor& key,
           const torch::Tensor& value, const torch::Tensor& positions, slm::KVCache& kv_cache,
           const slm::InputParameters& params) {
          slm::AttentionImpl attn(n_heads, n_kv_heads, head_dim, &handler, sliding_window);
          return attn.forward(query, key, value, positions, kv_cache, params);
        });
  // the first two steps of AttentionImpl::forward alone (what a profiling run with an empty cache
  // exercises): apply_pos_emb + append_kv_cache through the virtual interface
  m.def("handler_pos_emb_and_append",
        [](slm::HipAttnHandler& handler, int64_t n_heads, int64_t n_kv_heads, int64_t head_dim,
           torch::Tensor query, torch::Tensor key, const torch::Tensor& value,
           const torch::Tensor& positions, slm::KVCache& kv_cache, const slm::InputParameters& params) {
          slm::AttentionHandler& h = handler;
          const int64_t T = query.size(0);
          auto q = query.view({T, n_heads, head_dim});
          auto k = key.view({T, n_kv_heads, head_dim});
          auto v = value.view({T, n_kv_heads, head_dim});
          std::tie(q, k) = h.apply_pos_emb(q, k, positions);
          h.append_kv_cache(kv_cache, k, v, params);
        });
  // the C++ HOST STEP: slm::LlamaForCausalLMHip (csrc/shim/slm_llama_hip.h = the decoder stack of
  // src/models/meta/llama.h:123-345 composed from the C++ layer classes above) with the KV caches a
  // Worker would own.  tests/test_shim_gpu.py holds it bit-identical to decode.LlamaDecodeStep;
  // bench.py --host cpp times it.
  struct PyLlama {
    std::unique_ptr<slm::ProcessGroup> pg;  // world_size > 1: a LocalShardProcessGroup (one GPU, stubbed collectives)
    std::unique_ptr<slm::LlamaForCausalLMHip> model;
    std::vector<slm::KVCache> kv;
  };
  py::class_<PyLlama>(m, "LlamaForCausalLMHip")
      .def(py::init([make_args](int64_t hidden, int64_t n_heads, int64_t n_kv_heads, int64_t head_dim,
                                int64_t intermediate, int64_t n_layers, int64_t vocab, int64_t max_position,
                                double rope_theta, double rms_eps, const std::string& quant_method, int64_t bits,
                                int64_t group_size, bool desc_act, int64_t max_tokens, bool fused,
                                int64_t decode_lanes, bool lanes_chain, torch::ScalarType dtype, int device_index,
                                int rank, int world_size) {
             slm::LlamaArgs a;
             a.hidden_size = hidden; a.n_heads = n_heads; a.n_kv_heads = n_kv_heads; a.head_dim = head_dim;
             a.intermediate_size = intermediate; a.n_layers = n_layers; a.vocab_size = vocab;
             a.max_position_embeddings = max_position; a.rope_theta = static_cast<float>(rope_theta);
             a.rms_norm_eps = static_cast<float>(rms_eps);
             slm::LlamaForCausalLMHip::Options o;
             o.max_tokens = max_tokens; o.fused = fused; o.decode_lanes = decode_lanes; o.lanes_chain = lanes_chain;
             const bool awq = quant_method == "awq";
             auto p = std::make_unique<PyLlama>();
             const torch::Device dev(torch::kCUDA, device_index);
             // world_size > 1: rank `rank`'s shard of a TP model on this one GPU, collectives stubbed
             if (world_size > 1) p->pg = std::make_unique<slm::LocalShardProcessGroup>(rank, world_size, dev);
             p->model = std::make_unique<slm::LlamaForCausalLMHip>(
                 a, make_args(quant_method, bits, group_size, desc_act, /*is_sym=*/false, /*zero_point=*/awq),
                 slm::ParallelArgs(rank, world_size, p->pg.get()), torch::dtype(dtype).device(dev), o);
             return p;
           }),
           py::arg("hidden"), py::arg("n_heads"), py::arg("n_kv_heads"), py::arg("head_dim"), py::arg("intermediate"),
           py::arg("n_layers"), py::arg("vocab"), py::arg("max_position"), py::arg("rope_theta"), py::arg("rms_eps"),
           py::arg("quant_method"), py::arg("bits"), py::arg("group_size"), py::arg("desc_act"), py::arg("max_tokens"),
           py::arg("fused") = true, py::arg("decode_lanes") = -1, py::arg("lanes_chain") = true,
           py::arg("dtype") = torch::kBFloat16, py::arg("device_index") = 0, py::arg("rank") = 0,
           py::arg("world_size") = 1)
      .def("n_local_heads", [](PyLlama& self) { return self.model->n_local_heads(); })
      .def("n_local_kv_heads", [](PyLlama& self) { return self.model->n_local_kv_heads(); })
      .def("load_state_dict",
           [](PyLlama& self, std::unordered_map<std::string, torch::Tensor> sd) {
             self.model->load_state_dict(slm::StateDict(std::move(sd)));
           })
      .def("verify_loaded_weights", [](PyLlama& self) { self.model->verify_loaded_weights(); })
      // the engine-owned caches: one (key_cache, value_cache) pair per layer, shared, not copied
      .def("set_kv_caches",
           [](PyLlama& self, const std::vector<std::pair<torch::Tensor, torch::Tensor>>& caches, int64_t block_size) {
             self.kv.clear();
             for (const auto& kvp : caches) self.kv.emplace_back(kvp.first, kvp.second, block_size);
           })
      .def("set_cos_sin_cache", [](PyLlama& self, const torch::Tensor& t) { self.model->handler().set_cos_sin_cache(t); })
      .def("reserve", [](PyLlama& self, int64_t n_tokens) { self.model->reserve(n_tokens); })
      .def("decode_step",
           [](PyLlama& self, const torch::Tensor& tokens, const torch::Tensor& positions,
              const slm::InputParameters& params, bool return_logits) {
             return self.model->decode_step(tokens, positions, self.kv, params, return_logits);
           },
           py::arg("tokens"), py::arg("positions"), py::arg("params"), py::arg("return_logits") = false)
      .def("forward",
           [](PyLlama& self, const torch::Tensor& tokens, const torch::Tensor& positions,
              const slm::InputParameters& params) { return self.model->forward(tokens, positions, self.kv, params); })
      .def("last_lanes", [](PyLlama& self) { return self.model->last_lanes(); });
  py::class_<slm::W4Linear>(m, "W4Linear")
      .def(py::init<const std::string&, const torch::Tensor&, const torch::Tensor&,
                    const torch::Tensor&, const std::optional<torch::Tensor>&, int64_t, int64_t>(),
           py::arg("quant_method"), py::arg("qweight"), py::arg("qzeros"), py::arg("scales"),
           py::arg("g_idx"), py::arg("group_size"), py::arg("bits") = 4)
      .def("forward", &slm::W4Linear::forward, py::arg("input"), py::arg("bias") = std::nullopt,
           py::arg("out") = std::nullopt)
      .def("dequantize", &slm::W4Linear::dequantize)
      .def("in_features", &slm::W4Linear::in_features)
      .def("out_features", &slm::W4Linear::out_features);
  m.def("process_group_test", [](int n_devices) {
    // The reference's ProcessGroupTest (src/model_parallel/process_group_test.cpp:48-171) in its
    // own shape: ONE process, one communicator per GPU created together (ncclCommInitAll), one
    // host thread + stream per rank.  n_devices <= 0 means every visible GPU, so the same test
    // widens from world 1 on a 1-GPU box to world 8 on a full node.  Returns, per rank, whether
    // all-reduce (SUM of rank+1 == n(n+1)/2, exact) and both all-gather forms are correct.
    py::gil_scoped_release nogil;
    int n = n_devices > 0 ? n_devices : static_cast<int>(torch::cuda::device_count());
    std::vector<torch::Device> devs;
    for (int i = 0; i < n; ++i) devs.emplace_back(torch::kCUDA, i);
    // through the ABSTRACT interface (process_group.h:46-49), as every caller in the reference does
    std::vector<std::unique_ptr<slm::ProcessGroup>> pgs = slm::ProcessGroup::create_process_groups(devs);
    std::vector<int> ok(n, 0);
    std::vector<std::thread> threads;
    for (int r = 0; r < n; ++r)
      threads.emplace_back([&, r]() {
        c10::hip::HIPGuardMasqueradingAsCUDA guard(devs[r]);
        auto opt = torch::dtype(torch::kHalf).device(devs[r]);
        for (int rep = 0; rep < 3; ++rep) {   // repeated: a communicator must survive reuse
          auto t = torch::full({10, 10}, static_cast<float>(r + 1), opt);
          pgs[r]->allreduce(t);
          const bool ar = torch::equal(t.cpu(), torch::full({10, 10}, n * (n + 1) / 2.0f, torch::kHalf));
          auto mine = torch::full({4, 8}, static_cast<float>(r + 1), opt);
          auto flat = torch::zeros({n * 4, 8}, opt);
          pgs[r]->allgather(mine, flat);
          std::vector<torch::Tensor> outs;
          for (int q = 0; q < n; ++q) outs.push_back(torch::zeros({4, 8}, opt));
          pgs[r]->allgather(mine, outs);
          bool ag = true;
          for (int q = 0; q < n; ++q) {
            const auto want = torch::full({4, 8}, static_cast<float>(q + 1), torch::kHalf);
            ag = ag && torch::equal(flat.narrow(0, 4 * q, 4).cpu(), want) && torch::equal(outs[q].cpu(), want);
          }
          // all-to-all, equal splits (process_group_test.cpp:110-140): rank r sends row q = 100 r + q to
          // rank q and must end up with rows 100 q + r
          auto a2a_in = (torch::arange(n, torch::dtype(torch::kFloat).device(devs[r])) + 100.f * r)
                            .view({n, 1}).expand({n, 3}).contiguous();
          auto a2a_out = torch::zeros({n, 3}, a2a_in.options());
          pgs[r]->alltoall(a2a_in, a2a_out);
          auto want_a2a = (torch::arange(n, torch::kFloat) * 100.f + static_cast<float>(r)).view({n, 1}).expand({n, 3});
          bool aa = torch::equal(a2a_out.cpu(), want_a2a.contiguous());
          // uneven splits (process_group_test.cpp:142-171): rank r sends q + 1 rows to rank q and
          // receives r + 1 rows from everybody
          std::vector<int64_t> in_split, out_split;
          for (int q = 0; q < n; ++q) { in_split.push_back(q + 1); out_split.push_back(r + 1); }
          auto u_in = torch::full({n * (n + 1) / 2, 2}, static_cast<float>(r), torch::dtype(torch::kFloat).device(devs[r]));
          auto u_out = torch::full({n * (r + 1), 2}, -1.f, u_in.options());
          pgs[r]->alltoall(u_in, u_out, in_split, out_split);
          auto want_u = torch::arange(n, torch::kFloat).repeat_interleave(r + 1).view({-1, 1}).expand({n * (r + 1), 2});
          aa = aa && torch::equal(u_out.cpu(), want_u.contiguous());
          ok[r] = (rep == 0 ? 1 : ok[r]) && ar && ag && aa && pgs[r]->rank() == r && pgs[r]->world_size() == n &&
                  pgs[r]->device() == devs[r];
        }
      });
    for (auto& t : threads) t.join();
    return std::make_pair(n, ok);
  });
  m.def("process_group_selftest", [](int device_index) {
    // Worker::process_group_test (engine/worker.cpp:111-123) on the GPUs visible here
    std::vector<torch::Device> devs{torch::Device(torch::kCUDA, device_index)};
    auto pgs = slm::ProcessGroupRCCL::create_process_groups(devs);
    auto t = torch::ones({10, 10}, torch::dtype(torch::kHalf).device(devs[0]));
    pgs[0]->allreduce(t);
    auto g = torch::empty({10, 10}, t.options());
    pgs[0]->allgather(t, g);
    return std::make_pair(t.sum().item<float>(), g.sum().item<float>());
  });
  m.def("fused_allreduce_selftest", [](int device_index, int world, int64_t n_tokens, int64_t hidden) {
    // every rank on the SAME device, one host thread + one stream per rank (the reference drives
    // one Worker thread per GPU, engine/worker.cpp:202-213): checks FusedAllReduce against the
    // sequential sum -> llm::kernel::rms_norm_residual path.  Returns (max |diff| fused, max |diff|
    // plain sum, error bits).
    py::gil_scoped_release nogil;
    torch::Device dev(torch::kCUDA, device_index);
    std::vector<torch::Device> devs(world, dev);
    auto ars = slm::FusedAllReduce::create(devs, n_tokens, hidden, torch::kBFloat16);
    auto opt = torch::dtype(torch::kBFloat16).device(dev);
    std::vector<torch::Tensor> parts, parts2;
    for (int r = 0; r < world; ++r) {
      parts.push_back(torch::randn({n_tokens, hidden}, opt));
      parts2.push_back(torch::randn({n_tokens, hidden}, opt));
    }
    auto weight = torch::randn({hidden}, opt) * 0.1 + 1.0;
    auto res0 = torch::randn({n_tokens, hidden}, opt);
    auto sum_rn = [&](const std::vector<torch::Tensor>& ps) {
      auto acc = ps[0].to(torch::kFloat);
      for (int r = 1; r < world; ++r) acc = acc + ps[r].to(torch::kFloat);
      return acc.to(torch::kBFloat16);
    };
    auto x1 = sum_rn(parts), x2 = sum_rn(parts2);
    auto res_want = res0.clone();
    auto out_want = torch::empty_like(x1);
    llm::kernel::rms_norm_residual(out_want, res_want, x1, weight, 1e-5f);
    torch::cuda::synchronize(device_index);
    std::vector<double> d_fused(world, -1.0), d_sum(world, -1.0);
    std::vector<int> errs(world, -1);
    std::vector<std::thread> threads;
    for (int r = 0; r < world; ++r)
      threads.emplace_back([&, r]() {
        c10::hip::HIPStreamGuardMasqueradingAsCUDA sg(c10::hip::getStreamFromPoolMasqueradingAsCUDA(false, device_index));
        auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(device_index);
        auto out = torch::empty({n_tokens, hidden}, opt);
        auto res = res0.clone();
        ars[r]->buffer(0, n_tokens).copy_(parts[r]);
        ars[r]->allreduce_residual_rmsnorm(0, n_tokens, out, res, weight, 1e-5f);
        ars[r]->buffer(1, n_tokens).copy_(parts2[r]);
        auto s2 = ars[r]->allreduce(1, n_tokens);
        stream.synchronize();
        d_fused[r] = (out.to(torch::kFloat) - out_want.to(torch::kFloat)).abs().max().item<double>();
        d_sum[r] = (s2.to(torch::kFloat) - x2.to(torch::kFloat)).abs().max().item<double>();
        errs[r] = ars[r]->error();
      });
    for (auto& t : threads) t.join();
    double mf = 0, ms = 0;
    int e = 0;
    for (int r = 0; r < world; ++r) {
      mf = std::max(mf, d_fused[r]);
      ms = std::max(ms, d_sum[r]);
      e |= errs[r];
    }
    return std::make_tuple(mf, ms, e);
  });
}
