// Python test surface of the C++ shim, modelled on the reference's scalellm/csrc/kernels.cu
// (`_C.kernels`): lets pytest drive the C++ operator API exactly as a ScaleLLM layer would.
#include <torch/extension.h>

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
  py::class_<slm::W4Linear>(m, "W4Linear")
      .def(py::init<const std::string&, const torch::Tensor&, const torch::Tensor&,
                    const torch::Tensor&, const std::optional<torch::Tensor>&, int64_t>())
      .def("forward", &slm::W4Linear::forward, py::arg("input"), py::arg("bias") = std::nullopt,
           py::arg("out") = std::nullopt)
      .def("dequantize", &slm::W4Linear::dequantize);
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
}
