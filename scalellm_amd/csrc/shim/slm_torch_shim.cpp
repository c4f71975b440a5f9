// slm_torch_shim.cpp -- see slm_torch_shim.h.  Pure host code (g++), no device code: tensors are
// unpacked to (pointer, strides, sizes) and handed to the C ABI on torch's current HIP stream.
#include "slm_torch_shim.h"

// torch-ROCm tensors carry DeviceType "cuda" (HIP masquerades as CUDA), so the guard / stream
// accessors are the *MasqueradingAsCUDA flavours -- what the reference's at::cuda::OptionalCUDAGuard
// and at::cuda::getCurrentCUDAStream() resolve to in a hipified torch build.
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/hip/HIPGraphsC10Utils.h>
#include <rccl/rccl.h>

#include <mutex>
#include <stdexcept>
#include <unordered_map>

#include "slm_hip.h"

namespace {

int dtype_code(const torch::Tensor& t) {
  if (t.scalar_type() == torch::kBFloat16) return SLM_BF16;
  if (t.scalar_type() == torch::kHalf) return SLM_F16;
  // same restriction as the reference's DISPATCH_TORCH_DTYPE (common/static_dispatch.h:16-27)
  TORCH_CHECK(false, "slm: only fp16 / bf16 tensors are supported, got ", t.scalar_type());
  return -1;
}

void* current_stream(const torch::Tensor& t) {
  return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream();
}

void check(int rc, const char* what) {
  TORCH_CHECK(rc == SLM_OK, what, " failed: ", slm_status_string(rc), " (", rc, ")",
              rc == SLM_ERR_LAUNCH ? slm_last_hip_error() : "");
}

// one growable scratch buffer per device (split-KV / split-K partials); grown outside capture
//
// Lifetime rule: a buffer that was ever handed to a kernel is NEVER released.  The reference
// captures its graphs in ascending batch size, each after a warm-up (llm_engine.cpp:79,223;
// model_runner.cpp:162-175), and eager prefills may grow the scratch again later; a graph
// captured earlier keeps replaying against the raw address it recorded.  Growth therefore RETIRES
// the old buffer (kept alive in g_retired) and at least doubles, so the retired total stays below
// the final size.  Sizing it once up front (paged_kv_varlen_mha_set_workspace, one tensor per
// device) avoids retirements altogether.  Growth during stream capture is refused.
std::mutex g_ws_mu;
std::unordered_map<int, torch::Tensor> g_ws;       // device index -> current scratch
std::unordered_map<int, torch::Tensor> g_user_ws;  // device index -> caller-supplied scratch
std::vector<torch::Tensor> g_retired;              // replaced buffers (earlier captures point here)

torch::Tensor workspace_for(const torch::Tensor& like, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_ws_mu);
  const int dev = like.device().index();
  auto user = g_user_ws.find(dev);
  if (user != g_user_ws.end() && static_cast<size_t>(user->second.nbytes()) >= bytes)
    return user->second;
  auto& ws = g_ws[dev];
  if (!ws.defined() || static_cast<size_t>(ws.nbytes()) < bytes) {
    const auto capturing = c10::hip::currentStreamCaptureStatusMayInitCtx();
    TORCH_CHECK(capturing == c10::hip::CaptureStatus::None,
                "slm: the kernel workspace must be sized before graph capture (need ", bytes,
                " bytes): warm the step up first or call paged_kv_varlen_mha_set_workspace()");
    size_t size = std::max<size_t>(bytes, 1 << 20);
    if (ws.defined()) {
      size = std::max<size_t>(size, 2 * static_cast<size_t>(ws.nbytes()));
      g_retired.push_back(ws);
    }
    ws = torch::empty({static_cast<int64_t>(size)}, torch::dtype(torch::kUInt8).device(like.device()));
  }
  return ws;
}

}  // namespace

namespace llm {

void paged_kv_varlen_mha(torch::Tensor& out, const torch::Tensor& query,
                         const torch::Tensor& key_cache, const torch::Tensor& value_cache,
                         const torch::Tensor& q_cu_lens, const torch::Tensor& kv_cu_lens,
                         const torch::Tensor& block_table, const torch::Tensor& block_cu_lens,
                         const std::optional<torch::Tensor>& alibi_slopes, int block_size,
                         int max_q_len, int max_kv_len, float sm_scale, float logits_soft_cap,
                         int sliding_window) {
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(query.device());
  slm_attn_args a{};
  a.out = out.mutable_data_ptr();
  a.query = query.const_data_ptr();
  a.key_cache = key_cache.const_data_ptr();
  a.value_cache = value_cache.const_data_ptr();
  // strides in elements, last dim contiguous (attn_api.cpp:38-45)
  a.o_stride[0] = out.stride(0); a.o_stride[1] = out.stride(1);
  a.q_stride[0] = query.stride(0); a.q_stride[1] = query.stride(1);
  a.k_stride[0] = key_cache.stride(0); a.k_stride[1] = key_cache.stride(1);
  a.v_stride[0] = value_cache.stride(0); a.v_stride[1] = value_cache.stride(1);
  a.q_cu_lens = q_cu_lens.const_data_ptr<int32_t>();
  a.kv_cu_lens = kv_cu_lens.const_data_ptr<int32_t>();
  a.block_table = block_table.const_data_ptr<int32_t>();
  a.block_cu_lens = block_cu_lens.const_data_ptr<int32_t>();
  a.alibi_slopes = alibi_slopes.has_value() ? alibi_slopes.value().const_data_ptr<float>() : nullptr;
  a.dtype = dtype_code(query);
  a.batch_size = static_cast<int32_t>(q_cu_lens.size(0) - 1);
  a.n_tokens = static_cast<int32_t>(query.size(0));
  a.n_heads = static_cast<int32_t>(query.size(-2));
  a.n_kv_heads = static_cast<int32_t>(key_cache.size(-2));
  a.head_dim = static_cast<int32_t>(query.size(-1));
  a.block_size = block_size;
  a.max_q_len = max_q_len;
  a.max_kv_len = max_kv_len;
  a.sm_scale = sm_scale;
  a.logits_soft_cap = logits_soft_cap;
  a.sliding_window = sliding_window;
  a.num_splits = 0;
  if (a.n_tokens == 0 || a.batch_size == 0) return;
  // The reference's signature (attn_api.h:12-27) has no total-length argument and its engine fills none: what the
  // SIZES it hands over settle is derived here -- a pure-decode batch whose flattened block table holds exactly
  // batch * ceil(max_kv_len / block_size) entries is uniform to within one block (slm::uniform_kv_hint), and the
  // plan then skips the balanced partition and its combine launch.  Not under stream capture: a captured call
  // sees padded static buffers and bounds (model_runner.cpp:88-90, 196-200), not a batch.
  if (c10::hip::currentStreamCaptureStatusMayInitCtx() == c10::hip::CaptureStatus::None) {
    const int64_t total = slm::uniform_kv_hint(0, a.batch_size, max_q_len, max_kv_len, block_table.numel(), block_size);
    if (total > 0 && total < (int64_t(1) << 31)) a.total_kv_len = static_cast<int32_t>(total);
  }
  const size_t need = slm_paged_kv_varlen_mha_workspace_bytes(&a);
  torch::Tensor ws;
  if (need > 0) {
    ws = workspace_for(query, need);
    a.workspace = ws.mutable_data_ptr();
    a.workspace_bytes = ws.nbytes();
  }
  check(slm_paged_kv_varlen_mha(&a, current_stream(query)), "slm_paged_kv_varlen_mha");
}

int64_t paged_kv_varlen_mha_workspace_size(int64_t n_tokens, int64_t n_heads, int64_t head_dim) {
  return n_tokens * n_heads * 256 * (head_dim + 2) * static_cast<int64_t>(sizeof(float));
}

void paged_kv_varlen_mha_set_workspace(const torch::Tensor& workspace) {
  TORCH_CHECK(workspace.defined() && workspace.is_cuda() && workspace.is_contiguous(),
              "set_workspace: a contiguous device tensor");
  std::lock_guard<std::mutex> lk(g_ws_mu);
  // one per device (thread-per-GPU engines call this once per worker); a replaced tensor stays
  // alive: graphs captured against it may still replay
  auto& slot = g_user_ws[workspace.device().index()];
  if (slot.defined()) g_retired.push_back(slot);
  slot = workspace;
}

namespace kernel {

void set_kv_cache(const torch::Tensor& slot_ids, const torch::Tensor& keys,
                  const torch::Tensor& values, torch::Tensor& key_cache,
                  torch::Tensor& value_cache) {
  // keys and values contiguous at n_kv_heads and head_dim dims (kv_cache_kernels.cu:50-51)
  TORCH_CHECK(keys.stride(-1) == 1 && keys.stride(-2) == keys.size(-1));
  TORCH_CHECK(values.stride(-1) == 1 && values.stride(-2) == values.size(-1));
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(keys.device());
  check(slm_set_kv_cache(slot_ids.const_data_ptr<int32_t>(), keys.const_data_ptr(),
                         values.const_data_ptr(), keys.stride(-3), values.stride(-3),
                         key_cache.mutable_data_ptr(), value_cache.mutable_data_ptr(),
                         keys.size(-3), static_cast<int32_t>(keys.size(-2)),
                         static_cast<int32_t>(keys.size(-1)), dtype_code(keys),
                         current_stream(keys)),
        "slm_set_kv_cache");
}

void apply_rotary_pos_emb_and_append(torch::Tensor& query, torch::Tensor& key,
                                     const torch::Tensor& value, const torch::Tensor& positions,
                                     const torch::Tensor& cos_sin, int rotary_dim, bool interleaved,
                                     const torch::Tensor& slot_ids, torch::Tensor& key_cache,
                                     torch::Tensor& value_cache) {
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(query.device());
  const bool append = slot_ids.defined() && slot_ids.numel() > 0;
  const bool f32 = cos_sin.scalar_type() == torch::kFloat;
  TORCH_CHECK(f32 || cos_sin.scalar_type() == query.scalar_type());
  check(slm_rope_kv_append(query.mutable_data_ptr(), query.stride(0), key.mutable_data_ptr(),
                           key.stride(0), append ? value.const_data_ptr() : nullptr,
                           append ? value.stride(0) : 0, positions.const_data_ptr<int32_t>(),
                           cos_sin.const_data_ptr(), f32 ? 1 : 0, rotary_dim, interleaved ? 1 : 0,
                           append ? slot_ids.const_data_ptr<int32_t>() : nullptr,
                           append ? key_cache.mutable_data_ptr() : nullptr,
                           append ? value_cache.mutable_data_ptr() : nullptr, query.size(0),
                           static_cast<int32_t>(query.size(1)), static_cast<int32_t>(key.size(1)),
                           static_cast<int32_t>(query.size(2)), dtype_code(query),
                           current_stream(query)),
        "slm_rope_kv_append");
}

void apply_rotary_pos_emb(torch::Tensor& query, torch::Tensor& key, const torch::Tensor& positions,
                          const torch::Tensor& cos_sin, int rotary_dim, bool interleaved) {
  torch::Tensor none;
  apply_rotary_pos_emb_and_append(query, key, none, positions, cos_sin, rotary_dim, interleaved,
                                  none, none, none);
}

void rms_norm(torch::Tensor& out, torch::Tensor input, torch::Tensor weight, float epsilon) {
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(input.device());
  const int64_t dim = input.size(-1);
  check(slm_rms_norm(out.mutable_data_ptr(), input.const_data_ptr(), weight.const_data_ptr(),
                     nullptr, input.numel() / dim, dim, epsilon, dtype_code(input),
                     current_stream(input)),
        "slm_rms_norm");
}

void rms_norm_residual(torch::Tensor& out, torch::Tensor& residual, torch::Tensor input,
                       torch::Tensor weight, float epsilon) {
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(input.device());
  const int64_t dim = input.size(-1);
  check(slm_rms_norm(out.mutable_data_ptr(), input.const_data_ptr(), weight.const_data_ptr(),
                     residual.mutable_data_ptr(), input.numel() / dim, dim, epsilon,
                     dtype_code(input), current_stream(input)),
        "slm_rms_norm");
}

void silu_and_mul(torch::Tensor& out, torch::Tensor input) {
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(input.device());
  const int64_t d = input.size(-1) / 2;
  check(slm_silu_mul(out.mutable_data_ptr(), input.const_data_ptr(), input.numel() / (2 * d), d,
                     dtype_code(input), current_stream(input)),
        "slm_silu_mul");
}

void layer_norm(torch::Tensor& out, torch::Tensor input, torch::Tensor weight, torch::Tensor bias, float epsilon) {
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(input.device());
  TORCH_CHECK(input.is_contiguous() && out.is_contiguous(), "layer_norm: contiguous tensors");  // (the reference DCHECKs)
  const int64_t dim = input.size(-1);
  check(slm_layer_norm(out.mutable_data_ptr(), input.const_data_ptr(), weight.const_data_ptr(),
                       bias.defined() ? bias.const_data_ptr() : nullptr, input.numel() / dim, dim, epsilon,
                       dtype_code(input), current_stream(input)),
        "slm_layer_norm");
}

namespace {
torch::Tensor gelu_impl(const torch::Tensor& input, int kind, bool with_mul) {
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(input.device());
  TORCH_CHECK(input.is_contiguous() && (!with_mul || input.size(-1) % 2 == 0));
  auto sizes = input.sizes().vec();
  if (with_mul) sizes.back() /= 2;
  auto out = torch::empty(sizes, input.options());
  const int64_t d = sizes.back();
  check(slm_gelu(out.mutable_data_ptr(), input.const_data_ptr(), out.numel() / d, d, kind, with_mul ? 1 : 0,
                 dtype_code(input), current_stream(input)),
        "slm_gelu");
  return out;
}
}  // namespace

torch::Tensor gelu_new(torch::Tensor input) { return gelu_impl(input, SLM_GELU_NEW, false); }
torch::Tensor gelu_fast(torch::Tensor input) { return gelu_impl(input, SLM_GELU_FAST, false); }
torch::Tensor gelu_new_with_mul(torch::Tensor input) { return gelu_impl(input, SLM_GELU_NEW, true); }
torch::Tensor gelu_fast_with_mul(torch::Tensor input) { return gelu_impl(input, SLM_GELU_FAST, true); }

torch::Tensor silu_with_mul(torch::Tensor input) {
  // activation_kernels.cu:84-ff: out [..., d] = silu(input[..., :d]) * input[..., d:]
  TORCH_CHECK(input.is_contiguous() && input.size(-1) % 2 == 0);
  auto sizes = input.sizes().vec();
  sizes.back() /= 2;
  auto out = torch::empty(sizes, input.options());
  silu_and_mul(out, input);
  return out;
}

}  // namespace kernel
}  // namespace llm

// ---------------------------------------------------------------------------------------------
// marlin:: -- the reference's int4 kernel boundary (marlin.h:17-37), see slm_torch_shim.h
// ---------------------------------------------------------------------------------------------
namespace marlin {
namespace {
struct SzKey {
  const void* scales;
  const void* zeros;
  int64_t version, K, N;
  bool operator==(const SzKey& o) const {
    return scales == o.scales && zeros == o.zeros && version == o.version && K == o.K && N == o.N;
  }
};
struct SzKeyHash {
  size_t operator()(const SzKey& k) const {
    return std::hash<const void*>()(k.scales) ^ (std::hash<const void*>()(k.zeros) << 1) ^
           std::hash<int64_t>()(k.version * 1315423911 + k.K * 31 + k.N);
  }
};
// gptq_gemm receives scales / zeros on EVERY call (they are layer parameters, constant after
// load); the kernels want them fused into one {scale, magic + zero} word per (group, column).
// Built once per (scales, zeros) pair on first use -- during the engine's warm-up, before graph
// capture -- and kept for the life of the process; steady state is a hash lookup.
// An entry is valid only while the storages it was built from are alive: a freed parameter's
// address can be handed to a different tensor by the caching allocator.
struct SzEntry {
  torch::Tensor sz;
  c10::weak_intrusive_ptr<c10::StorageImpl> scales_st, zeros_st;
};
std::mutex g_sz_mu;
std::unordered_map<SzKey, SzEntry, SzKeyHash> g_sz_cache;

// num_bits = 8: the doubled activation gather perm2[k'] = perm[k' mod K] (or k' mod K), one per
// (perm tensor, K, device), same lifetime rule as the sz tables
struct Perm2Key {
  const void* perm;
  uint32_t version;
  int64_t K;
  int device;
  bool operator==(const Perm2Key& o) const {
    return perm == o.perm && version == o.version && K == o.K && device == o.device;
  }
};
struct Perm2KeyHash {
  size_t operator()(const Perm2Key& k) const {
    return std::hash<const void*>()(k.perm) ^ std::hash<int64_t>()(k.K * 131 + k.version * 7 + k.device);
  }
};
struct Perm2Entry {
  torch::Tensor perm2;
  c10::weak_intrusive_ptr<c10::StorageImpl> perm_st;
};
std::unordered_map<Perm2Key, Perm2Entry, Perm2KeyHash> g_perm2_cache;

bool same_live_storage(const c10::weak_intrusive_ptr<c10::StorageImpl>& w, const torch::Tensor& t) {
  const auto alive = w.lock();
  return alive && alive.get() == t.storage().unsafeGetStorageImpl();
}

// is_k_full = false on a tensor that DOES hold whole groups (the reference's own test grid runs
// that combination on full layers, marlin_gemm_test.py:52) is the same computation as is_k_full =
// true; whether the SORTED g_idx the caller passes is the regular 0,0,..,1,1,.. pattern is decided
// once per g_idx tensor (a device -> host comparison: first call = warm-up, never under capture).
std::unordered_map<const void*, std::pair<c10::weak_intrusive_ptr<c10::StorageImpl>, bool>> g_gidx_cache;
bool sorted_groups_are_whole(const torch::Tensor& g_idx, int64_t K, int64_t G) {
  if (!g_idx.defined() || g_idx.numel() != K || G <= 0 || K % G) return false;
  std::lock_guard<std::mutex> lk(g_sz_mu);
  const auto it = g_gidx_cache.find(g_idx.const_data_ptr());
  if (it != g_gidx_cache.end() && same_live_storage(it->second.first, g_idx)) return it->second.second;
  for (auto e = g_gidx_cache.begin(); e != g_gidx_cache.end();)
    e = e->second.first.expired() ? g_gidx_cache.erase(e) : std::next(e);
  const auto trivial = torch::arange(K, torch::dtype(torch::kLong).device(g_idx.device())).floor_divide(K / G);
  const bool whole = torch::equal(g_idx.to(torch::kLong), trivial);
  g_gidx_cache.erase(g_idx.const_data_ptr());  // (weak pointers are not default-constructible: no operator[])
  g_gidx_cache.emplace(g_idx.const_data_ptr(),
                       std::make_pair(c10::weak_intrusive_ptr<c10::StorageImpl>(g_idx.storage().getWeakStorageImpl()), whole));
  return whole;
}

void check_repack_args(const torch::Tensor& q_weight, const torch::Tensor& out, int64_t num_bits, int64_t K,
                       int64_t N) {
  TORCH_CHECK(num_bits == 4 || num_bits == 8, "num_bits must be 4 or 8, got ", num_bits);
  TORCH_CHECK(q_weight.is_cuda() && q_weight.is_contiguous() && q_weight.scalar_type() == torch::kInt);
  // marlin repack output: [K/16, N*16/pack_factor] int32 (gptq_repack.cu / awq_repack.cu), pack_factor = 32/bits
  TORCH_CHECK(out.is_cuda() && out.is_contiguous() && out.scalar_type() == torch::kInt &&
              out.numel() == K * N / (32 / num_bits), "out must be int32 [K/16, N*16/pack_factor]");
  // 8 bits: two int4 planes over 2K packed rows (include/slm_hip.h section 3b) = the same byte count
  const int64_t Kp = num_bits == 8 ? 2 * K : K;
  TORCH_CHECK(slm_w4_packed_weight_bytes(Kp, N) == static_cast<size_t>(Kp * N / 2),
              "unsupported shape K=", K, " N=", N, " (need K % 64 == 0, N % 32 == 0)");
}
}  // namespace

void gptq_repack(const torch::Tensor& q_weight, const torch::Tensor& perm, torch::Tensor& out,
                 int64_t num_bits) {
  TORCH_CHECK(num_bits == 4 || num_bits == 8, "num_bits must be 4 or 8, got ", num_bits);
  const int64_t K = q_weight.size(0) * (32 / num_bits), N = q_weight.size(1);
  check_repack_args(q_weight, out, num_bits, K, N);
  const bool has_perm = perm.defined() && perm.numel() > 0;
  if (has_perm)
    TORCH_CHECK(perm.numel() == K && perm.scalar_type() == torch::kInt && perm.is_contiguous(),
                "perm must be contiguous int32 [K]");
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(q_weight.device());
  if (num_bits == 8) {
    // the doubled activation gather is rebuilt (and cached) by gptq_gemm from `perm`: this signature
    // has nowhere to return it
    auto perm2 = torch::empty({2 * K}, torch::dtype(torch::kInt).device(q_weight.device()));
    check(slm_w8_prepack_weights(SLM_W8_GPTQ, q_weight.const_data_ptr<int32_t>(),
                                 has_perm ? perm.const_data_ptr<int32_t>() : nullptr, K, N, out.mutable_data_ptr(),
                                 perm2.mutable_data_ptr<int32_t>(), current_stream(q_weight)),
          "slm_w8_prepack_weights");
    return;
  }
  check(slm_w4_prepack_weights(SLM_W4_GPTQ, q_weight.const_data_ptr<int32_t>(),
                               has_perm ? perm.const_data_ptr<int32_t>() : nullptr, K, N,
                               out.mutable_data_ptr(), current_stream(q_weight)),
        "slm_w4_prepack_weights");
}

void awq_repack(const torch::Tensor& q_weight, torch::Tensor& out, int64_t num_bits) {
  TORCH_CHECK(num_bits == 4 || num_bits == 8, "num_bits must be 4 or 8, got ", num_bits);
  const int64_t K = q_weight.size(0), N = q_weight.size(1) * (32 / num_bits);
  check_repack_args(q_weight, out, num_bits, K, N);
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(q_weight.device());
  if (num_bits == 8) {
    auto perm2 = torch::empty({2 * K}, torch::dtype(torch::kInt).device(q_weight.device()));
    check(slm_w8_prepack_weights(SLM_W8_AWQ, q_weight.const_data_ptr<int32_t>(), nullptr, K, N,
                                 out.mutable_data_ptr(), perm2.mutable_data_ptr<int32_t>(), current_stream(q_weight)),
          "slm_w8_prepack_weights");
    return;
  }
  check(slm_w4_prepack_weights(SLM_W4_AWQ, q_weight.const_data_ptr<int32_t>(), nullptr, K, N,
                               out.mutable_data_ptr(), current_stream(q_weight)),
        "slm_w4_prepack_weights");
}

void gptq_gemm(const torch::Tensor& A, const torch::Tensor& B, torch::Tensor& C,
               const torch::Tensor& scales, const torch::Tensor& zeros, const torch::Tensor& g_idx,
               const torch::Tensor& perm, torch::Tensor& /*workspace*/, int num_bits, bool is_k_full,
               bool has_zp, bool /*use_fp32_reduce*/) {
  TORCH_CHECK(num_bits == 4 || num_bits == 8, "num_bits must be 4 or 8, got ", num_bits);
  const bool w8 = num_bits == 8;
  // is_k_full = false is Marlin's mode for a row-parallel act-order shard: rows of one K shard belong
  // to groups all over the FULL scale table, looked up through g_idx.  This function takes the
  // packed weights at their checkpoint size, where that cannot be expressed; the layer classes
  // (slm::RowParallelQLinearHipImpl -> slm::W4Linear) pack such a shard with padded groups instead.
  if (!is_k_full && perm.defined() && perm.numel() > 0)
    TORCH_CHECK(sorted_groups_are_whole(g_idx, A.size(1), scales.size(0)),
                "marlin::gptq_gemm on the HIP path: is_k_full = false with uneven groups (a row-parallel "
                "act-order shard) is handled by slm::RowParallelQLinearHipImpl, not by the raw kernel entry point");
  TORCH_CHECK(A.dim() == 2 && C.dim() == 2 && A.stride(1) == 1 && C.stride(1) == 1);
  const int64_t M = A.size(0), K = A.size(1), N = C.size(1);
  TORCH_CHECK(C.size(0) == M && B.numel() == K * N / (32 / num_bits) && B.scalar_type() == torch::kInt &&
                  B.is_contiguous(),
              "B must be the int32 [K/16, N*16/pack_factor] tensor gptq_repack / awq_repack produced");
  TORCH_CHECK(scales.dim() == 2 && scales.size(1) == N && scales.is_contiguous() &&
              scales.scalar_type() == A.scalar_type() && K % scales.size(0) == 0,
              "scales must be [n_groups, N] of the activation dtype, plain column order");
  const int64_t G = scales.size(0), gs = K / G;
  const bool zp = has_zp && zeros.defined() && zeros.numel() > 0;
  if (zp)
    TORCH_CHECK(zeros.scalar_type() == torch::kInt && zeros.is_contiguous() &&
                    zeros.numel() == G * N / (32 / num_bits),
                "zeros must be the AWQ checkpoint tensor [n_groups, N/pack_factor] int32");
  // 8 bits: two int4 planes over 2K packed rows; the scale table is written at the packed group size
  const int64_t Kp = w8 ? slm_w8_packed_rows(K) : K;
  const int64_t gsp = w8 ? slm_w8_packed_group_size(K, gs) : gs;
  TORCH_CHECK(gsp > 0, "unsupported group size ", gs, " for ", num_bits, "-bit weights");
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(A.device());
  torch::Tensor sz;
  {
    // _version() throws on inference-mode tensors (they carry no version counter and cannot be
    // modified in place by autograd-visible ops): those count as version 0
    const auto ver = [](const torch::Tensor& t) -> int64_t { return t.is_inference() ? 0 : static_cast<int64_t>(t._version()); };
    const SzKey key{scales.const_data_ptr(), zp ? zeros.const_data_ptr() : nullptr,
                    ver(scales) * 65537 + (zp ? ver(zeros) : 0), w8 ? -K : K, N};
    std::lock_guard<std::mutex> lk(g_sz_mu);
    auto it = g_sz_cache.find(key);
    if (it != g_sz_cache.end() &&
        !(same_live_storage(it->second.scales_st, scales) && (!zp || same_live_storage(it->second.zeros_st, zeros)))) {
      g_sz_cache.erase(it);  // the address was recycled for another tensor
      it = g_sz_cache.end();
    }
    if (it == g_sz_cache.end()) {
      // a miss is rare (once per layer, during warm-up): sweep the entries whose parameters are gone
      // (a model reload puts the layers at new addresses; their tables would otherwise live forever)
      for (auto e = g_sz_cache.begin(); e != g_sz_cache.end();)
        e = (e->second.scales_st.expired() || e->second.zeros_st.expired()) ? g_sz_cache.erase(e) : std::next(e);
      sz = torch::empty({(Kp / gsp) * N}, torch::dtype(torch::kInt).device(A.device()));
      if (w8)
        check(slm_w8_prepack_sz(zp ? SLM_W8_AWQ : SLM_W8_GPTQ, zp ? zeros.const_data_ptr<int32_t>() : nullptr,
                                scales.const_data_ptr(), K, N, gs, dtype_code(scales), sz.mutable_data_ptr(),
                                current_stream(A)),
              "slm_w8_prepack_sz");
      else
        check(slm_w4_prepack_sz(zp ? SLM_W4_AWQ : SLM_W4_GPTQ, zp ? zeros.const_data_ptr<int32_t>() : nullptr,
                                scales.const_data_ptr(), K, N, gs, dtype_code(scales), sz.mutable_data_ptr(),
                                current_stream(A)),
              "slm_w4_prepack_sz");
      using WeakStorage = c10::weak_intrusive_ptr<c10::StorageImpl>;
      const auto& keep_alive_of_zeros = zp ? zeros : scales;
      g_sz_cache.emplace(key, SzEntry{sz, WeakStorage(scales.storage().getWeakStorageImpl()),
                                      WeakStorage(keep_alive_of_zeros.storage().getWeakStorageImpl())});
    } else {
      sz = it->second.sz;
    }
  }
  const bool has_perm = perm.defined() && perm.numel() > 0;
  torch::Tensor perm2;  // 8 bits: packed row k' reads activation column perm[k' mod K]
  if (w8) {
    // built once per (perm tensor, K) and kept next to the sz tables: no allocation and no extra
    // launches on the decode path (and none inside a stream capture after the warm-up call)
    const auto ver = [](const torch::Tensor& t) { return t.is_inference() ? 0u : t._version(); };
    const Perm2Key pk{has_perm ? perm.const_data_ptr() : nullptr, has_perm ? ver(perm) : 0u, K,
                      static_cast<int>(A.device().index())};
    std::lock_guard<std::mutex> lk(g_sz_mu);
    auto it = g_perm2_cache.find(pk);
    if (it != g_perm2_cache.end() && has_perm && !same_live_storage(it->second.perm_st, perm)) {
      g_perm2_cache.erase(it);
      it = g_perm2_cache.end();
    }
    if (it == g_perm2_cache.end()) {
      for (auto e = g_perm2_cache.begin(); e != g_perm2_cache.end();)
        e = (e->first.perm && e->second.perm_st.expired()) ? g_perm2_cache.erase(e) : std::next(e);
      const auto base = has_perm ? perm.to(torch::kInt)
                                 : torch::arange(K, torch::dtype(torch::kInt).device(A.device()));
      perm2 = torch::cat({base, base}).contiguous();
      using WeakStorage = c10::weak_intrusive_ptr<c10::StorageImpl>;
      g_perm2_cache.emplace(pk, Perm2Entry{perm2, has_perm ? WeakStorage(perm.storage().getWeakStorageImpl())
                                                            : WeakStorage(perm2.storage().getWeakStorageImpl())});
    } else {
      perm2 = it->second.perm2;
    }
  }
  slm_w4_gemm_args g{};
  g.a = A.const_data_ptr();
  g.wq = B.const_data_ptr();
  g.sz = sz.const_data_ptr();
  g.perm = w8 ? perm2.const_data_ptr<int32_t>() : (has_perm ? perm.const_data_ptr<int32_t>() : nullptr);
  g.c = C.mutable_data_ptr();
  g.M = M; g.K = Kp; g.N = N;
  g.lda = A.stride(0); g.ldc = C.stride(0);
  g.group_size = gsp;
  g.dtype = dtype_code(A);
  if (M == 0) return;
  const size_t need = slm_w4a16_gemm_workspace_bytes(&g);
  torch::Tensor ws;
  if (need > 0) {
    ws = workspace_for(A, need);
    g.workspace = ws.mutable_data_ptr();
    g.workspace_bytes = ws.nbytes();
  }
  check(slm_w4a16_gemm(&g, current_stream(A)), "slm_w4a16_gemm");
}

}  // namespace marlin

namespace slm {

size_t marlin_sz_cache_entries() {
  std::lock_guard<std::mutex> lk(marlin::g_sz_mu);
  return marlin::g_sz_cache.size();
}

W4Linear::W4Linear(const std::string& quant_method, const torch::Tensor& qweight,
                   const torch::Tensor& qzeros, const torch::Tensor& scales,
                   const std::optional<torch::Tensor>& g_idx, int64_t group_size, int64_t bits, bool paired) {
  const bool awq = quant_method == "awq";
  paired_ = paired;
  TORCH_CHECK(!paired || bits == 4, "paired (gate | up) packing is built for 4-bit weights");
  TORCH_CHECK(awq || quant_method == "gptq", "quant_method must be awq or gptq");
  TORCH_CHECK(bits == 4 || bits == 8, "bits must be 4 or 8, got ", bits);
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(qweight.device());
  const int64_t per = 32 / bits;
  K_ = awq ? qweight.size(0) : qweight.size(0) * per;
  N_ = awq ? qweight.size(1) * per : qweight.size(1);
  group_size_ = group_size > 0 ? group_size : K_;
  dtype_ = scales.scalar_type();
  if (!awq && g_idx.has_value() && g_idx->numel() > 0) {
    // act-order: rows sorted by group, like qlinear_gptq_marlin_impl.cpp:41-71
    const auto gi = g_idx->to(torch::kLong);
    const auto trivial =
        torch::arange(K_, torch::dtype(torch::kLong).device(gi.device())).floor_divide(group_size_);
    if (!torch::equal(gi, trivial)) {
      const auto perm = torch::argsort(gi, /*stable=*/true, /*dim=*/0, /*descending=*/false);
      if (!torch::equal(gi.index({perm}), trivial)) {
        // uneven groups after sorting: a row-parallel shard of an act-order checkpoint (sharded
        // qweight / g_idx, FULL scales: qlinear_gptq_marlin_impl.cpp:236-243,270-276; the reference
        // then runs Marlin with is_k_full = false, :319)
        TORCH_CHECK(bits == 4, "act-order shards with uneven groups are supported for 4-bit weights only");
        TORCH_CHECK(!paired, "paired (gate | up) packing of an act-order shard with uneven groups is not supported");
        pack_uneven_groups(qweight, qzeros, scales, gi, perm);
        return;
      }
      perm_ = perm.to(torch::kInt).contiguous();
    }
  }
  k_src_ = K_;
  if (bits == 8) {
    // two int4 planes over 2K packed rows + the doubled activation gather (slm_hip.h section 3b)
    const int64_t K = K_, gs = group_size_;
    const int64_t gsp = slm_w8_packed_group_size(K, gs);
    TORCH_CHECK(gsp > 0, "unsupported 8-bit group size ", gs);
    K_ = slm_w8_packed_rows(K);
    group_size_ = gsp;
    const size_t wb = slm_w4_packed_weight_bytes(K_, N_), sb = slm_w4_packed_sz_bytes(K_, N_, gsp);
    TORCH_CHECK(wb > 0 && sb > 0, "unsupported 8-bit shape K=", K, " N=", N_, " group=", gs);
    const auto iopt = torch::dtype(torch::kInt).device(qweight.device());
    wq_ = torch::empty({static_cast<int64_t>(wb / 4)}, iopt);
    sz_ = torch::empty({static_cast<int64_t>(sb / 4)}, iopt);
    auto perm2 = torch::empty({K_}, iopt);
    const auto qw = qweight.contiguous(), sc = scales.contiguous();
    const bool has_qz = qzeros.defined() && qzeros.numel() > 0;
    const auto qz = has_qz ? qzeros.contiguous() : torch::Tensor();
    const int fmt = awq ? SLM_W8_AWQ : SLM_W8_GPTQ;
    check(slm_w8_prepack_weights(fmt, qw.const_data_ptr<int32_t>(),
                                 perm_.defined() ? perm_.const_data_ptr<int32_t>() : nullptr, K, N_,
                                 wq_.mutable_data_ptr(), perm2.mutable_data_ptr<int32_t>(), current_stream(qw)),
          "slm_w8_prepack_weights");
    check(slm_w8_prepack_sz(fmt, has_qz ? qz.const_data_ptr<int32_t>() : nullptr, sc.const_data_ptr(), K, N_, gs,
                            dtype_code(sc), sz_.mutable_data_ptr(), current_stream(qw)),
          "slm_w8_prepack_sz");
    perm_ = perm2;
    return;
  }
  const size_t wb = slm_w4_packed_weight_bytes(K_, N_);
  const size_t sb = slm_w4_packed_sz_bytes(K_, N_, group_size_);
  TORCH_CHECK(wb > 0 && sb > 0, "unsupported int4 shape K=", K_, " N=", N_, " group=", group_size_);
  const auto iopt = torch::dtype(torch::kInt).device(qweight.device());
  wq_ = torch::empty({static_cast<int64_t>(wb / 4)}, iopt);
  sz_ = torch::empty({static_cast<int64_t>(sb / 4)}, iopt);
  const auto qw = qweight.contiguous(), qz = qzeros.contiguous(), sc = scales.contiguous();
  if (paired) TORCH_CHECK(N_ % 64 == 0, "paired (gate | up) prepack needs N % 64 == 0, got N = ", N_);
  check(slm_w4_prepack((awq ? SLM_W4_AWQ : SLM_W4_GPTQ) | (paired ? SLM_W4_PAIRED : 0), qw.const_data_ptr<int32_t>(),
                       qz.const_data_ptr<int32_t>(), sc.const_data_ptr(),
                       perm_.defined() ? perm_.const_data_ptr<int32_t>() : nullptr, K_, N_,
                       group_size_, dtype_code(sc), wq_.mutable_data_ptr(), sz_.mutable_data_ptr(),
                       current_stream(qw)),
        "slm_w4_prepack");
}

// Same plan as kernels.plan_uneven_groups (scalellm_amd/kernels.py): the sorted rows of every
// group are padded to a multiple of 32 rows (padding rows: perm = -1, weights 0, activation column
// gathered as +0.0), the total to a multiple of 128; every 32-row block then belongs to one group
// and the kernels run it as group_size = 32 with one {scale, zero} row per block.
void W4Linear::pack_uneven_groups(const torch::Tensor& qweight, const torch::Tensor& qzeros,
                                  const torch::Tensor& scales, const torch::Tensor& gi,
                                  const torch::Tensor& perm) {
  constexpr int64_t B = 32;
  const int64_t G = scales.size(0), gs = group_size_;
  TORCH_CHECK(scales.dim() == 2 && scales.size(1) == N_ && qzeros.size(0) == G && qzeros.size(1) == N_ / 8,
              "act-order shard: scales / qzeros must be the FULL tables [n_groups, N]");
  TORCH_CHECK(gi.min().item<int64_t>() >= 0 && gi.max().item<int64_t>() < G,
              "g_idx refers to a scale group the scales tensor does not have");
  const auto lopt = torch::dtype(torch::kLong).device(gi.device());
  const auto gs_sorted = gi.index({perm});
  const auto counts = torch::bincount(gs_sorted, /*weights=*/{}, /*minlength=*/G);
  const auto padded = (counts + (B - 1)).floor_divide(B) * B;
  const auto starts = torch::cumsum(padded, 0) - padded;
  const auto cstarts = torch::cumsum(counts, 0) - counts;
  const auto pos = starts.index({gs_sorted}) + (torch::arange(K_, lopt) - cstarts.index({gs_sorted}));
  const int64_t total = padded.sum().item<int64_t>();
  const int64_t kp = (total + 127) / 128 * 128;
  auto perm_p = torch::full({kp}, -1, torch::dtype(torch::kInt).device(gi.device()));
  perm_p.index_put_({pos}, perm.to(torch::kInt));
  auto block_group = torch::repeat_interleave(torch::arange(G, lopt), padded.floor_divide(B));
  if (kp > total) block_group = torch::cat({block_group, torch::zeros({(kp - total) / B}, lopt)});
  k_src_ = K_;
  K_ = kp;
  group_size_ = B;
  perm_ = perm_p.contiguous();
  const auto iopt = torch::dtype(torch::kInt).device(qweight.device());
  const size_t wb = slm_w4_packed_weight_bytes(K_, N_);
  TORCH_CHECK(wb > 0, "unsupported int4 shape K=", K_, " N=", N_);
  wq_ = torch::empty({static_cast<int64_t>(wb / 4)}, iopt);
  const auto qw = qweight.contiguous(), qz = qzeros.contiguous(), sc = scales.contiguous();
  check(slm_w4_prepack_weights(SLM_W4_GPTQ, qw.const_data_ptr<int32_t>(), perm_.const_data_ptr<int32_t>(), K_, N_,
                               wq_.mutable_data_ptr(), current_stream(qw)),
        "slm_w4_prepack_weights");
  auto sz_full = torch::empty({G, N_}, iopt);
  check(slm_w4_prepack_sz(SLM_W4_GPTQ, qz.const_data_ptr<int32_t>(), sc.const_data_ptr(), G * gs, N_, gs,
                          dtype_code(sc), sz_full.mutable_data_ptr(), current_stream(qw)),
        "slm_w4_prepack_sz");
  sz_ = sz_full.index({block_group}).contiguous().view({-1});
}

torch::Tensor W4Linear::forward(const torch::Tensor& input, const std::optional<torch::Tensor>& bias,
                                std::optional<torch::Tensor> out) const {
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(input.device());
  const auto a = input.reshape({-1, input.size(-1)});
  TORCH_CHECK(a.size(1) == k_src_ && a.stride(1) == 1 && a.scalar_type() == dtype_);
  torch::Tensor c = out.has_value() ? *out : torch::empty({a.size(0), N_}, a.options());
  slm_w4_gemm_args g{};
  g.a = a.const_data_ptr();
  g.wq = wq_.const_data_ptr();
  g.sz = sz_.const_data_ptr();
  g.perm = perm_.defined() ? perm_.const_data_ptr<int32_t>() : nullptr;
  g.bias = bias.has_value() && bias->defined() ? bias->const_data_ptr() : nullptr;
  g.c = c.mutable_data_ptr();
  g.M = a.size(0); g.K = K_; g.N = N_;
  g.lda = a.stride(0); g.ldc = c.stride(0);
  g.group_size = group_size_;
  g.dtype = dtype_code(a);
  if (g.M == 0) return c;
  const size_t need = slm_w4a16_gemm_workspace_bytes(&g);
  torch::Tensor ws;
  if (need > 0) {
    ws = workspace_for(a, need);
    g.workspace = ws.mutable_data_ptr();
    g.workspace_bytes = ws.nbytes();
  }
  check(slm_w4a16_gemm(&g, current_stream(a)), "slm_w4a16_gemm");
  return c;
}

namespace {
slm_w4_gemm_args w4_args(const torch::Tensor& a, const void* wq, const void* sz, const int32_t* perm, void* c,
                         int64_t ldc, int64_t K, int64_t N, int64_t gs, int flags) {
  slm_w4_gemm_args g{};
  g.a = a.defined() ? a.const_data_ptr() : nullptr;
  g.wq = wq; g.sz = sz; g.perm = perm; g.bias = nullptr; g.c = c;
  g.M = a.defined() ? a.size(0) : 0; g.K = K; g.N = N;
  g.lda = a.defined() ? a.stride(0) : K; g.ldc = ldc;
  g.group_size = gs;
  g.dtype = a.defined() ? dtype_code(a) : SLM_BF16;
  g.flags = flags;
  return g;
}
}  // namespace

size_t W4Linear::workspace_bytes(int64_t M, int flags) const {
  slm_w4_gemm_args g{};
  g.wq = wq_.const_data_ptr(); g.sz = sz_.const_data_ptr();
  g.perm = perm_.defined() ? perm_.const_data_ptr<int32_t>() : nullptr;
  g.M = M; g.K = K_; g.N = N_; g.lda = k_src_; g.ldc = (flags & SLM_W4_SILU_MUL) ? N_ / 2 : N_;
  g.group_size = group_size_;
  g.dtype = dtype_ == torch::kBFloat16 ? SLM_BF16 : SLM_F16;
  g.flags = flags;
  if ((flags & SLM_W4_DEFER_REDUCE) && !slm_w4a16_gemm_deferred_splits(&g)) g.flags &= ~SLM_W4_DEFER_REDUCE;
  return slm_w4a16_gemm_workspace_bytes(&g);
}

int W4Linear::forward_into(const torch::Tensor& a, torch::Tensor& c, int flags, const torch::Tensor& workspace) const {
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(a.device());
  TORCH_CHECK(a.dim() == 2 && a.size(1) == k_src_ && a.stride(1) == 1 && a.scalar_type() == dtype_);
  TORCH_CHECK(!(flags & SLM_W4_SILU_MUL) || paired_, "SLM_W4_SILU_MUL needs weights packed paired");
  const int64_t n_out = (flags & SLM_W4_SILU_MUL) ? N_ / 2 : N_;
  TORCH_CHECK(c.dim() == 2 && c.size(0) == a.size(0) && c.size(1) == n_out && c.stride(1) == 1);
  auto g = w4_args(a, wq_.const_data_ptr(), sz_.const_data_ptr(),
                   perm_.defined() ? perm_.const_data_ptr<int32_t>() : nullptr, c.mutable_data_ptr(), c.stride(0), K_,
                   N_, group_size_, flags);
  if (g.M == 0) return 0;
  int deferred = 0;
  if (flags & SLM_W4_DEFER_REDUCE) {
    deferred = slm_w4a16_gemm_deferred_splits(&g);
    if (!deferred) g.flags &= ~SLM_W4_DEFER_REDUCE;
  }
  const size_t need = slm_w4a16_gemm_workspace_bytes(&g);
  if (need > 0) {
    TORCH_CHECK(workspace.defined() && static_cast<size_t>(workspace.nbytes()) >= need,
                "W4Linear::forward_into: scratch of ", need, " bytes needed (reserve it before the step)");
    g.workspace = workspace.mutable_data_ptr();
    g.workspace_bytes = workspace.nbytes();
  }
  check(slm_w4a16_gemm(&g, current_stream(a)), "slm_w4a16_gemm");
  return deferred;
}

torch::Tensor W4Linear::dequantize() const {
  // The dense weight of the CHECKPOINT, [in_features, N] in its own row order, whatever the packing:
  // the packed rows k' (act-order: sorted by group; 8 bits: two int4 planes over 2K rows; padded
  // act-order shards: perm = -1 rows that hold nothing) are dequantised as the kernels see them and
  // scattered back, W[perm[k']] += Wp[k'] in fp32, rounded to T once.
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(wq_.device());
  auto wp = torch::empty({K_, N_}, torch::dtype(dtype_).device(wq_.device()));
  check(slm_w4_dequant(wq_.const_data_ptr(), sz_.const_data_ptr(), K_, N_, group_size_,
                       dtype_ == torch::kBFloat16 ? SLM_BF16 : SLM_F16, wp.mutable_data_ptr(),
                       current_stream(wq_)),
        "slm_w4_dequant");
  if (!perm_.defined()) return wp;
  const auto idx = perm_.to(torch::kLong);
  const auto live = idx.ge(0);
  auto w = torch::zeros({k_src_, N_}, torch::dtype(torch::kFloat).device(wq_.device()));
  w.index_add_(0, idx.masked_select(live), wp.to(torch::kFloat).index({live}));
  return w.to(dtype_);
}

// --------------------------------------------------------------------------------------------
// ProcessGroupRCCL
// --------------------------------------------------------------------------------------------
namespace {
ncclDataType_t to_nccl(const torch::Tensor& t) {
  switch (t.scalar_type()) {
    case torch::kFloat: return ncclFloat;
    case torch::kHalf: return ncclHalf;
    case torch::kBFloat16: return ncclBfloat16;
    case torch::kInt: return ncclInt32;
    case torch::kLong: return ncclInt64;
    case torch::kDouble: return ncclDouble;
    case torch::kByte: return ncclUint8;
    default: TORCH_CHECK(false, "unsupported dtype for RCCL: ", t.scalar_type());
  }
  return ncclFloat;
}
void nccl_check(ncclResult_t r, const char* what) {
  TORCH_CHECK(r == ncclSuccess, what, " failed: ", ncclGetErrorString(r));
}
void check_input(const torch::Tensor& t) {  // process_group.cpp:59-63
  TORCH_CHECK(t.is_cuda() && t.is_contiguous(), "tensor must be a contiguous device tensor");
}
}  // namespace

std::vector<std::unique_ptr<ProcessGroupRCCL>> ProcessGroupRCCL::create_process_groups(
    const std::vector<torch::Device>& devices) {
  std::vector<int> ids;
  for (const auto& d : devices) ids.push_back(d.index());
  std::vector<ncclComm_t> comms(devices.size());
  nccl_check(ncclCommInitAll(comms.data(), static_cast<int>(devices.size()), ids.data()),
             "ncclCommInitAll");
  std::vector<std::unique_ptr<ProcessGroupRCCL>> out;
  for (size_t i = 0; i < devices.size(); ++i)
    out.emplace_back(new ProcessGroupRCCL(static_cast<int>(i), static_cast<int>(devices.size()),
                                          devices[i], comms[i]));
  return out;
}

ProcessGroupRCCL::~ProcessGroupRCCL() {
  if (comm_) ncclCommDestroy(static_cast<ncclComm_t>(comm_));
}

void ProcessGroupRCCL::allreduce(torch::Tensor& input) const {
  check_input(input);
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device());
  nccl_check(ncclAllReduce(input.const_data_ptr(), input.mutable_data_ptr(), input.numel(),
                           to_nccl(input), ncclSum, static_cast<ncclComm_t>(comm_),
                           c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(device().index()).stream()),
             "ncclAllReduce");
}

void ProcessGroupRCCL::allgather(const torch::Tensor& input, torch::Tensor& outputs) const {
  check_input(input);
  check_input(outputs);
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device());
  nccl_check(ncclAllGather(input.const_data_ptr(), outputs.mutable_data_ptr(), input.numel(),
                           to_nccl(input), static_cast<ncclComm_t>(comm_),
                           c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(device().index()).stream()),
             "ncclAllGather");
}

void ProcessGroupRCCL::allgather(const torch::Tensor& input,
                                 std::vector<torch::Tensor>& outputs) const {
  TORCH_CHECK(static_cast<int>(outputs.size()) == world_size());
  auto flat = torch::empty({world_size() * input.numel()}, input.options());
  allgather(input, flat);
  for (int r = 0; r < world_size(); ++r)
    outputs[r].copy_(flat.narrow(0, r * input.numel(), input.numel()).view_as(outputs[r]));
}

void ProcessGroupRCCL::alltoall(const torch::Tensor& input, torch::Tensor& output) const {
  alltoall(input, output, {}, {});
}

// process_group.cpp:240-292 (check_split_sizes / compute_lengths_and_offsets :65-96): splits are
// counted in ROWS (dim 0); an empty list = equal splits, which must then divide dim 0
void ProcessGroupRCCL::alltoall(const torch::Tensor& input, torch::Tensor& output,
                                const std::vector<int64_t>& input_split_sizes,
                                const std::vector<int64_t>& output_split_sizes) const {
  check_input(input);
  check_input(output);
  TORCH_CHECK(input.scalar_type() == output.scalar_type(), "input and output should have the same dtype");
  TORCH_CHECK(input.device() == device() && output.device() == device(),
              "tensors should be on the same device as the process group");
  const int n = world_size();
  auto lengths = [&](const std::vector<int64_t>& splits, const torch::Tensor& t, std::vector<size_t>& len,
                     std::vector<size_t>& off) {
    const int64_t rows = t.dim() > 0 ? t.size(0) : 1;
    const int64_t row_numel = rows > 0 ? t.numel() / rows : 0;
    if (splits.empty()) {
      TORCH_CHECK(rows % n == 0, "tensor's dim 0 does not divide equally across the group size");
    } else {
      TORCH_CHECK(static_cast<int>(splits.size()) == n, "number of tensor splits not equal to group size");
      int64_t sum = 0;
      for (const auto v : splits) sum += v;
      TORCH_CHECK(sum == rows, "split sizes do not match the total dim 0 size");
    }
    size_t o = 0;
    for (int r = 0; r < n; ++r) {
      const size_t l = static_cast<size_t>((splits.empty() ? rows / n : splits[r]) * row_numel);
      len[r] = l;
      off[r] = o;
      o += l;
    }
  };
  std::vector<size_t> slen(n), soff(n), rlen(n), roff(n);
  lengths(input_split_sizes, input, slen, soff);
  lengths(output_split_sizes, output, rlen, roff);
  const size_t es = input.element_size();
  const char* sb = reinterpret_cast<const char*>(input.const_data_ptr());
  char* rb = reinterpret_cast<char*>(output.mutable_data_ptr());
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device());
  const auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(device().index()).stream();
  const auto comm = static_cast<ncclComm_t>(comm_);
  const auto type = to_nccl(input);
  nccl_check(ncclGroupStart(), "ncclGroupStart");
  for (int r = 0; r < n; ++r) {
    nccl_check(ncclSend(sb + soff[r] * es, slen[r], type, r, comm, stream), "ncclSend");
    nccl_check(ncclRecv(rb + roff[r] * es, rlen[r], type, r, comm, stream), "ncclRecv");
  }
  nccl_check(ncclGroupEnd(), "ncclGroupEnd");
}

std::vector<std::unique_ptr<ProcessGroup>> ProcessGroup::create_process_groups(
    const std::vector<torch::Device>& devices) {
  std::vector<std::unique_ptr<ProcessGroup>> out;
  for (auto& pg : ProcessGroupRCCL::create_process_groups(devices)) out.push_back(std::move(pg));
  return out;
}

// --------------------------------------------------------------------------------------------
// FusedAllReduce
// --------------------------------------------------------------------------------------------
struct FusedAllReduce::Shared {
  std::vector<torch::Device> devices;
  std::vector<void*> signals;
  std::vector<void*> buffers[2];
  int64_t max_tokens = 0, hidden = 0;
  torch::ScalarType dtype = torch::kBFloat16;
  ~Shared() {
    for (size_t r = 0; r < devices.size(); ++r) {
      c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(devices[r]);
      if (r < signals.size() && signals[r]) slm_shm_free(signals[r]);
      for (auto& b : buffers)
        if (r < b.size() && b[r]) slm_shm_free(b[r]);
    }
  }
};

std::vector<std::shared_ptr<FusedAllReduce>> FusedAllReduce::create(
    const std::vector<torch::Device>& devices, int64_t max_tokens, int64_t hidden,
    torch::ScalarType dtype) {
  const int world = static_cast<int>(devices.size());
  TORCH_CHECK(world >= 2 && world <= SLM_AR_MAX_RANKS, "FusedAllReduce: world size ", world);
  TORCH_CHECK(dtype == torch::kHalf || dtype == torch::kBFloat16, "FusedAllReduce: fp16 / bf16 only");
  auto sh = std::make_shared<Shared>();
  sh->devices = devices;
  sh->max_tokens = max_tokens;
  sh->hidden = hidden;
  sh->dtype = dtype;
  sh->signals.assign(world, nullptr);
  for (auto& b : sh->buffers) b.assign(world, nullptr);
  for (int r = 0; r < world; ++r) {
    c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(devices[r]);
    check(slm_shm_alloc(&sh->signals[r], slm_ar_signal_bytes(), /*uncached=*/1), "slm_shm_alloc");
    for (auto& b : sh->buffers)
      check(slm_shm_alloc(&b[r], static_cast<size_t>(max_tokens * hidden * 2), 0), "slm_shm_alloc");
    for (int q = 0; q < world; ++q)
      check(slm_shm_enable_peer_access(devices[r].index(), devices[q].index()),
                "slm_shm_enable_peer_access");
  }
  std::vector<std::shared_ptr<FusedAllReduce>> out;
  for (int r = 0; r < world; ++r)
    out.emplace_back(new FusedAllReduce(r, world, devices[r], sh));
  return out;
}

FusedAllReduce::~FusedAllReduce() = default;

torch::Tensor FusedAllReduce::buffer(int i, int64_t n_tokens) const {
  TORCH_CHECK((i == 0 || i == 1) && n_tokens >= 1 && n_tokens <= sh_->max_tokens);
  return torch::from_blob(sh_->buffers[i][rank_], {n_tokens, sh_->hidden},
                          torch::dtype(sh_->dtype).device(device_));
}

namespace {
void fill_ar_args(slm_ar_args& a, int rank, int world, const std::vector<void*>& signals,
                  const std::vector<void*>& buffers, int64_t n_tokens, int64_t hidden,
                  torch::ScalarType dtype) {
  a.rank = rank;
  a.world = world;
  for (int r = 0; r < world; ++r) {
    a.signals[r] = signals[r];
    a.buffers[r] = buffers[r];
  }
  a.M = n_tokens;
  a.H = hidden;
  a.dtype = dtype == torch::kBFloat16 ? SLM_BF16 : SLM_F16;
}
}  // namespace

torch::Tensor FusedAllReduce::allreduce(int i, int64_t n_tokens) const {
  auto t = buffer(i, n_tokens);
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_);
  slm_ar_args a{};
  fill_ar_args(a, rank_, world_size_, sh_->signals, sh_->buffers[i], n_tokens, sh_->hidden, sh_->dtype);
  a.out = t.mutable_data_ptr();
  check(slm_allreduce(&a, c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(device_.index()).stream()),
            "slm_allreduce");
  return t;
}

void FusedAllReduce::allreduce_residual_rmsnorm(int i, int64_t n_tokens, torch::Tensor& out,
                                                torch::Tensor& residual, const torch::Tensor& weight,
                                                float eps) const {
  TORCH_CHECK(i == 0 || i == 1);
  for (const torch::Tensor* t : {static_cast<const torch::Tensor*>(&out), static_cast<const torch::Tensor*>(&residual), &weight})
    TORCH_CHECK(t->is_cuda() && t->is_contiguous() && t->scalar_type() == sh_->dtype,
                "allreduce_residual_rmsnorm: contiguous device tensors of the group dtype");
  TORCH_CHECK(out.size(0) == n_tokens && out.size(1) == sh_->hidden && residual.sizes() == out.sizes() &&
              weight.numel() == sh_->hidden, "allreduce_residual_rmsnorm: shape mismatch");
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_);
  slm_ar_args a{};
  fill_ar_args(a, rank_, world_size_, sh_->signals, sh_->buffers[i], n_tokens, sh_->hidden, sh_->dtype);
  a.out = out.mutable_data_ptr();
  a.residual = residual.mutable_data_ptr();
  a.weight = weight.const_data_ptr();
  a.eps = eps;
  check(slm_allreduce(&a, c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(device_.index()).stream()),
            "slm_allreduce");
}

int FusedAllReduce::error() const {
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_);
  int32_t e = 0;
  check(slm_ar_read_error(sh_->signals[rank_], &e), "slm_ar_read_error");
  return e;
}

}  // namespace slm
