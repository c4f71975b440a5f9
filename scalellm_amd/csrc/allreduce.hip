// allreduce.hip -- two-shot all-reduce over peer-mapped (xGMI) buffers, fused with the residual add
// and RMSNorm that follow every row-parallel linear of a decoder layer.
//
// Replaces ProcessGroupNCCL::allreduce (reference src/model_parallel/process_group.cpp:135-153, as
// called by reduce_from_model_parallel_region, model_parallel.cpp:33-44) + kernel::rms_norm_residual
// (src/kernels/layernorm_kernels.cu:125; math src/layers/normalization.h:42-52) for the decode-size
// messages ([n_tokens, hidden], 2 MiB at bs = 256) where a ring collective is latency-bound.
//
// Shape of the thing (one launch per rank, 256-thread workgroups, one ROW of the message at a time):
//   start barrier   every rank's partial sums are complete (its producer ran earlier in its stream)
//   stage 1         rank r owns rows [r * rpr, (r + 1) * rpr): for each of them read the row of EVERY
//                   rank (7 of 8 over xGMI, all loads of a batch in flight together), add in fp32 in
//                   rank order, round to T; fused: h = x + residual, residual = T(h),
//                   y = T(h * rsqrt(mean(h^2) + eps)) * w -- the exact arithmetic (and reduction
//                   tree) of rms_norm_kernel in glue.hip; write y to `out` and, in place, to the own
//                   buffer where the peers will fetch it
//   mid barrier     release (L2 write-back) -> flags -> acquire (invalidate)
//   stage 2         gather the rows of the other ranks from THEIR buffers into `out`
//   (end barrier)   optional; callers alternate two buffers instead (see slm_hip.h)
//   one-shot        M <= world (and `out` is not the message buffer): start barrier, then every rank
//                   reduces ALL rows itself -- no publish, no mid barrier, no gather
// Workgroup b only ever depends on workgroup b of the peers (it produces rows b, b + nb, ... of its
// rank's share and consumes the same rows of the others), so the barriers are per workgroup:
// flags[b][rank] in the PEER's signal block, written with system-scope stores, polled locally.
// Flags carry a per-workgroup launch counter kept in the signal block, so nothing is ever reset and
// a captured graph replays correctly.  Every remote access is a READ of peer memory or a 4-byte flag
// store; all bulk writes are local.
//
// Memory-model notes (gfx942-family rules, which gfx950 shares): signal blocks are fine-grained
// UNCACHED memory (a local L2 would keep a stale copy of a word a peer writes).  Data buffers are
// ordinary device memory: a producer kernel's writes reach memory at its end-of-kernel release;
// everything this kernel reads from a message buffer (local or remote) is a system-scope load and
// the rows it publishes are system-scope (write-through) stores whose completion is awaited
// before the flag goes out -- so no cache level ever holds a stale or a private copy of shared data
// and the barriers need no L2 write-back / invalidate.
#include <hip/hip_runtime.h>

#include <cstring>
#include <type_traits>

#include "common.h"

namespace slm {

struct ArSignal {
  uint32_t start[SLM_AR_MAX_BLOCKS][SLM_AR_MAX_RANKS];
  uint32_t mid[SLM_AR_MAX_BLOCKS][SLM_AR_MAX_RANKS];
  uint32_t end[SLM_AR_MAX_BLOCKS][SLM_AR_MAX_RANKS];
  uint32_t counter[SLM_AR_MAX_BLOCKS];  // launches seen by workgroup b (local use only)
  uint32_t err;                         // sticky SLM_AR_ERR_* bits
  uint32_t pad[3];
};

struct ArParams {
  int rank, world;
  ArSignal* sig[SLM_AR_MAX_RANKS];
  uint16_t* buf[SLM_AR_MAX_RANKS];
  uint16_t* out;
  uint16_t* residual;
  const uint16_t* weight;
  float eps;
  int64_t M, H;
  int rpr;  // rows per rank = ceil(M / world)
  int end_barrier;
  int one_shot;  // M <= world and out is not the message buffer: every rank reduces EVERY row
};

struct ArSimParams {
  ArParams r[SLM_AR_MAX_RANKS];
};

// System-scope accesses (sc0 sc1: never served from, never left dirty in, a non-coherent cache
// level) for everything a peer reads or writes, as relaxed 8-byte atomics so that the compiler sees
// them (its wait counts stay exact; volatile would serialise every load).  With them no L2
// write-back / invalidate is needed around the flag barriers -- release/acquire fences cost a
// buffer_wbl2 / buffer_inv walk of the whole L2 per workgroup (measured: 2-3x the kernel time).
__device__ __forceinline__ u32x4 ar_ld_sys(const uint16_t* ptr) {
  const uint64_t* q = reinterpret_cast<const uint64_t*>(ptr);
  const uint64_t lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const uint64_t hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  u32x4 v;
  v.x = (uint32_t)lo; v.y = (uint32_t)(lo >> 32); v.z = (uint32_t)hi; v.w = (uint32_t)(hi >> 32);
  return v;
}
__device__ __forceinline__ void ar_st_sys(uint16_t* ptr, const u32x4 v) {
  uint64_t* q = reinterpret_cast<uint64_t*>(ptr);
  __hip_atomic_store(q, (uint64_t)v.x | ((uint64_t)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(q + 1, (uint64_t)v.z | ((uint64_t)v.w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// a peer that has not arrived after this long (100 MHz wall clock ticks: 20 s -- start-up skew
// between ranks can be seconds) is presumed dead: raise the error word, do not hang the GPU
constexpr uint64_t AR_TIMEOUT_TICKS = 20ull * 100000000ull;

// Cross-rank barrier of workgroup b.  which: 0 start, 1 mid, 2 end.  All threads call it.
// RELEASE: the caller's earlier system-scope stores must have completed (be visible to the peers)
// before its flag is: every wave drains its own stores, then the workgroup barrier, then the flag.
template <bool RELEASE>
__device__ __forceinline__ void ar_barrier(const ArParams& p, int which, int b, uint32_t flag) {
  if constexpr (RELEASE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int t = threadIdx.x;
  if (t < p.world) {
    ArSignal* peer = p.sig[t];
    ArSignal* self = p.sig[p.rank];
    uint32_t* dst = which == 0 ? &peer->start[b][p.rank] : which == 1 ? &peer->mid[b][p.rank] : &peer->end[b][p.rank];
    const uint32_t* src = which == 0 ? &self->start[b][t] : which == 1 ? &self->mid[b][t] : &self->end[b][t];
    __hip_atomic_store(dst, flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint64_t t0 = wall_clock64();
    bool arrived = false;
    for (;;) {
      const uint32_t v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if ((int32_t)(v - flag) >= 0) {
        arrived = true;
        break;
      }
      if (wall_clock64() - t0 > AR_TIMEOUT_TICKS) break;
      __builtin_amdgcn_s_sleep(4);
    }
    if (!arrived)
      __hip_atomic_fetch_or(&self->err, (uint32_t)SLM_AR_ERR_TIMEOUT, __ATOMIC_RELAXED,
                            __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __syncthreads();  // peer data is read with system-scope loads issued after this point
}

__device__ __forceinline__ int ar_rows_of(const ArParams& p, int r) {
  const int64_t lo = (int64_t)r * p.rpr;
  const int64_t hi = lo + p.rpr < p.M ? lo + p.rpr : p.M;
  return hi > lo ? (int)(hi - lo) : 0;
}

// WMAX: compile-time bound of world (2, 4, 8) so that the per-rank loads of a batch unroll and fly
// together.  The thread -> column mapping and the reduction tree are those of rms_norm_kernel.
template <typename T, bool FUSED, int WMAX>
__device__ __forceinline__ void ar_body(const ArParams& p, const int b, const int nb) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const int64_t H = p.H;
  const int nvec = (int)(H / 8);
  constexpr int MAXV = 8;  // up to 8 x 8 x 256 = 16384 columns
  ArSignal* self = p.sig[p.rank];
  const uint32_t flag =
      __hip_atomic_load(&self->counter[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1u;

  ar_barrier<false>(p, 0, b, flag);

  // ---- stage 1: reduce (and normalise) this rank's rows.  One-shot mode (at most one row per
  // rank): every rank reduces ALL rows itself straight into `out` -- redundant work on a few rows
  // instead of a second flag barrier and a gather; nothing is published, the peers' reads of this
  // rank's buffer go on undisturbed ----
  const int my_rows = p.one_shot ? (int)p.M : ar_rows_of(p, p.rank);
  const int64_t row_base = p.one_shot ? 0 : (int64_t)p.rank * p.rpr;
  for (int row = b; row < my_rows; row += nb) {
    const int64_t off = (row_base + row) * H;
    float v[MAXV][8];
    float ss = 0.f;
#pragma unroll
    for (int i0 = 0; i0 < MAXV; i0 += 2) {
      if (tid + 256 * i0 >= nvec) break;
      // one batch = this thread's two vectors of every rank's row, all in flight together
      u32x4 ld[WMAX * 2];
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int vi = tid + 256 * (i0 + ii);
#pragma unroll
        for (int r = 0; r < WMAX; ++r)
          if (r < p.world && vi < nvec) ld[r * 2 + ii] = ar_ld_sys(p.buf[r] + off + vi * 8);
      }
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int i = i0 + ii;
        const int vi = tid + 256 * i;
        if (vi >= nvec) break;
        float f[8];
#pragma unroll
        for (int r = 0; r < WMAX; ++r)
          if (r < p.world) {  // rank order, fp32: every rank computes the same bits
            const u32x4 l = ld[r * 2 + ii];
            if (r == 0) {
              f[0] = lo_f32<T>(l.x); f[1] = hi_f32<T>(l.x); f[2] = lo_f32<T>(l.y); f[3] = hi_f32<T>(l.y);
              f[4] = lo_f32<T>(l.z); f[5] = hi_f32<T>(l.z); f[6] = lo_f32<T>(l.w); f[7] = hi_f32<T>(l.w);
            } else {
              f[0] += lo_f32<T>(l.x); f[1] += hi_f32<T>(l.x); f[2] += lo_f32<T>(l.y); f[3] += hi_f32<T>(l.y);
              f[4] += lo_f32<T>(l.z); f[5] += hi_f32<T>(l.z); f[6] += lo_f32<T>(l.w); f[7] += hi_f32<T>(l.w);
            }
          }
        // the all-reduce result, rounded to T (what ncclAllReduce leaves in the tensor)
        u32x4 x;
        x.x = pack2<T>(f[0], f[1]); x.y = pack2<T>(f[2], f[3]);
        x.z = pack2<T>(f[4], f[5]); x.w = pack2<T>(f[6], f[7]);
        if constexpr (!FUSED) {
          if (!p.one_shot) ar_st_sys(p.buf[p.rank] + off + vi * 8, x);
          if (p.out != p.buf[p.rank]) *reinterpret_cast<u32x4*>(p.out + off + vi * 8) = x;
        } else {
          // h = x + residual (fp32), residual = T(h): normalization.h:42-52
          const u32x4 rr = *reinterpret_cast<const u32x4*>(p.residual + off + vi * 8);
          f[0] = lo_f32<T>(x.x) + lo_f32<T>(rr.x); f[1] = hi_f32<T>(x.x) + hi_f32<T>(rr.x);
          f[2] = lo_f32<T>(x.y) + lo_f32<T>(rr.y); f[3] = hi_f32<T>(x.y) + hi_f32<T>(rr.y);
          f[4] = lo_f32<T>(x.z) + lo_f32<T>(rr.z); f[5] = hi_f32<T>(x.z) + hi_f32<T>(rr.z);
          f[6] = lo_f32<T>(x.w) + lo_f32<T>(rr.w); f[7] = hi_f32<T>(x.w) + hi_f32<T>(rr.w);
          u32x4 w;
          w.x = pack2<T>(f[0], f[1]); w.y = pack2<T>(f[2], f[3]);
          w.z = pack2<T>(f[4], f[5]); w.w = pack2<T>(f[6], f[7]);
          *reinterpret_cast<u32x4*>(p.residual + off + vi * 8) = w;
#pragma unroll
          for (int j = 0; j < 8; ++j) v[i][j] = f[j];
          ss = rms_sumsq8(f, ss);
        }
      }
    }
    if constexpr (FUSED) {
      ss = group_sum<64>(ss);
      if ((tid & 63) == 0) red[tid >> 6] = ss;
      __syncthreads();
      const float tot = red[0] + red[1] + red[2] + red[3];
      const float rs = rsqrtf(tot / (float)H + p.eps);
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int vi = tid + 256 * i;
        if (vi < nvec) {
          const u32x4 wv = *reinterpret_cast<const u32x4*>(p.weight + vi * 8);
          const u32x4 y = rms_apply8<T>(v[i], rs, wv);
          if (!p.one_shot) ar_st_sys(p.buf[p.rank] + off + vi * 8, y);
          if (p.out != p.buf[p.rank]) *reinterpret_cast<u32x4*>(p.out + off + vi * 8) = y;
        }
      }
      __syncthreads();  // red[] is reused by the next row
    }
  }

  if (p.one_shot) {
    if (p.end_barrier) ar_barrier<false>(p, 2, b, flag);
    if (tid == 0)
      __hip_atomic_store(&self->counter[b], flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  ar_barrier<true>(p, 1, b, flag);

  // ---- stage 2: gather the other ranks' rows (peer order rotated so that the 7 links of a rank
  // are not all asked for the same peer first) ----
  for (int row = b; row < p.rpr; row += nb) {
#pragma unroll
    for (int i0 = 0; i0 < MAXV; i0 += 2) {
      if (tid + 256 * i0 >= nvec) break;
      u32x4 ld[(WMAX - 1) * 2];
#pragma unroll
      for (int k = 1; k < WMAX; ++k) {
        int q = p.rank + k;
        if (q >= p.world) q -= p.world;
        const bool live = k < p.world && row < ar_rows_of(p, q);
        const int64_t off = ((int64_t)q * p.rpr + row) * H;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const int vi = tid + 256 * (i0 + ii);
          if (live && vi < nvec) ld[(k - 1) * 2 + ii] = ar_ld_sys(p.buf[q] + off + vi * 8);
        }
      }
#pragma unroll
      for (int k = 1; k < WMAX; ++k) {
        int q = p.rank + k;
        if (q >= p.world) q -= p.world;
        const bool live = k < p.world && row < ar_rows_of(p, q);
        const int64_t off = ((int64_t)q * p.rpr + row) * H;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const int vi = tid + 256 * (i0 + ii);
          if (live && vi < nvec) *reinterpret_cast<u32x4*>(p.out + off + vi * 8) = ld[(k - 1) * 2 + ii];
        }
      }
    }
  }

  if (p.end_barrier) ar_barrier<false>(p, 2, b, flag);
  if (tid == 0)
    __hip_atomic_store(&self->counter[b], flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <typename T, bool FUSED, int WMAX>
__global__ void __launch_bounds__(256) xgmi_allreduce_kernel(const ArParams p) {
  ar_body<T, FUSED, WMAX>(p, (int)blockIdx.x, (int)gridDim.x);
}

// all ranks in one launch on one device (slm_allreduce_simulate): blockIdx.y is the rank
template <typename T, bool FUSED, int WMAX>
__global__ void __launch_bounds__(256) xgmi_allreduce_sim_kernel(const ArSimParams sp) {
  ar_body<T, FUSED, WMAX>(sp.r[blockIdx.y], (int)blockIdx.x, (int)gridDim.x);
}

static int ar_fill(const slm_ar_args* a, ArParams* p) {
  if (!a) return SLM_ERR_INVALID_ARG;
  if (a->world < 2 || a->world > SLM_AR_MAX_RANKS || a->rank < 0 || a->rank >= a->world)
    return SLM_ERR_INVALID_ARG;
  if (a->M < 1 || a->H < 8 || !a->out) return SLM_ERR_INVALID_ARG;
  if (a->H % 8 != 0 || a->H > 16384) return SLM_ERR_UNSUPPORTED;
  if (a->dtype != SLM_BF16 && a->dtype != SLM_F16) return SLM_ERR_UNSUPPORTED;
  if ((a->residual != nullptr) != (a->weight != nullptr)) return SLM_ERR_INVALID_ARG;
  if (!aligned16(a->out) || (a->residual && (!aligned16(a->residual) || !aligned16(a->weight))))
    return SLM_ERR_ALIGNMENT;
  for (int r = 0; r < a->world; ++r) {
    if (!a->signals[r] || !a->buffers[r]) return SLM_ERR_INVALID_ARG;
    if (!aligned16(a->signals[r]) || !aligned16(a->buffers[r])) return SLM_ERR_ALIGNMENT;
  }
  p->rank = a->rank;
  p->world = a->world;
  for (int r = 0; r < SLM_AR_MAX_RANKS; ++r) {
    p->sig[r] = r < a->world ? reinterpret_cast<ArSignal*>(a->signals[r]) : nullptr;
    p->buf[r] = r < a->world ? reinterpret_cast<uint16_t*>(a->buffers[r]) : nullptr;
  }
  p->out = reinterpret_cast<uint16_t*>(a->out);
  p->residual = reinterpret_cast<uint16_t*>(a->residual);
  p->weight = reinterpret_cast<const uint16_t*>(a->weight);
  p->eps = a->eps;
  p->M = a->M;
  p->H = a->H;
  p->rpr = (int)((a->M + a->world - 1) / a->world);
  p->end_barrier = a->end_barrier;
  p->one_shot = (a->M <= a->world && a->out != a->buffers[a->rank]) ? 1 : 0;
  return SLM_OK;
}

template <typename F>
static void ar_dispatch(int dtype, bool fused, int world, F&& launch) {
  const int wmax = world <= 2 ? 2 : world <= 4 ? 4 : 8;
#define SLM_AR_CASE(TT, FF, WW) launch(TT{}, std::integral_constant<bool, FF>{}, std::integral_constant<int, WW>{})
#define SLM_AR_W(TT, FF)                     \
  do {                                       \
    if (wmax == 2) SLM_AR_CASE(TT, FF, 2);   \
    else if (wmax == 4) SLM_AR_CASE(TT, FF, 4); \
    else SLM_AR_CASE(TT, FF, 8);             \
  } while (0)
  if (dtype == SLM_BF16) {
    if (fused) SLM_AR_W(bf16_tag, true);
    else SLM_AR_W(bf16_tag, false);
  } else {
    if (fused) SLM_AR_W(f16_tag, true);
    else SLM_AR_W(f16_tag, false);
  }
#undef SLM_AR_W
#undef SLM_AR_CASE
}

}  // namespace slm

using namespace slm;

extern "C" {

SLM_API size_t slm_ar_signal_bytes(void) { return sizeof(ArSignal); }

SLM_API int slm_shm_alloc(void** ptr, size_t bytes, int32_t uncached) {
  if (!ptr || bytes == 0) return SLM_ERR_INVALID_ARG;
  hip_clear_error();
  hipError_t e = uncached ? hipExtMallocWithFlags(ptr, bytes, hipDeviceMallocUncached)
                          : hipMalloc(ptr, bytes);
  if (e == hipSuccess) e = hipMemset(*ptr, 0, bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  hip_last_error_slot() = (int)e;
  return e == hipSuccess ? SLM_OK : SLM_ERR_LAUNCH;
}

SLM_API int slm_shm_free(void* ptr) {
  if (!ptr) return SLM_ERR_INVALID_ARG;
  const hipError_t e = hipFree(ptr);
  hip_last_error_slot() = (int)e;
  return e == hipSuccess ? SLM_OK : SLM_ERR_LAUNCH;
}

SLM_API int slm_shm_export(void* ptr, uint8_t handle[SLM_SHM_HANDLE_BYTES]) {
  static_assert(sizeof(hipIpcMemHandle_t) == SLM_SHM_HANDLE_BYTES, "IPC handle size");
  if (!ptr || !handle) return SLM_ERR_INVALID_ARG;
  hipIpcMemHandle_t h;
  const hipError_t e = hipIpcGetMemHandle(&h, ptr);
  hip_last_error_slot() = (int)e;
  if (e != hipSuccess) return SLM_ERR_LAUNCH;
  memcpy(handle, &h, sizeof(h));
  return SLM_OK;
}

SLM_API int slm_shm_import(const uint8_t handle[SLM_SHM_HANDLE_BYTES], void** ptr) {
  if (!ptr || !handle) return SLM_ERR_INVALID_ARG;
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  const hipError_t e = hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess);
  hip_last_error_slot() = (int)e;
  return e == hipSuccess ? SLM_OK : SLM_ERR_LAUNCH;
}

SLM_API int slm_shm_close(void* ptr) {
  if (!ptr) return SLM_ERR_INVALID_ARG;
  const hipError_t e = hipIpcCloseMemHandle(ptr);
  hip_last_error_slot() = (int)e;
  return e == hipSuccess ? SLM_OK : SLM_ERR_LAUNCH;
}

SLM_API int slm_shm_enable_peer_access(int32_t device, int32_t peer_device) {
  if (device < 0 || peer_device < 0) return SLM_ERR_INVALID_ARG;
  if (device == peer_device) return SLM_OK;
  int prev = 0;
  hipError_t e = hipGetDevice(&prev);
  if (e == hipSuccess) e = hipSetDevice(device);
  if (e == hipSuccess) {
    e = hipDeviceEnablePeerAccess(peer_device, 0);
    if (e == hipErrorPeerAccessAlreadyEnabled) {
      (void)hipGetLastError();
      e = hipSuccess;
    }
    (void)hipSetDevice(prev);
  }
  hip_last_error_slot() = (int)e;
  return e == hipSuccess ? SLM_OK : SLM_ERR_LAUNCH;
}

SLM_API int slm_ar_read_error(const void* own_signal, int32_t* err) {
  if (!own_signal || !err) return SLM_ERR_INVALID_ARG;
  uint32_t v = 0;
  const hipError_t e = hipMemcpy(&v, &reinterpret_cast<const ArSignal*>(own_signal)->err, sizeof(v),
                                 hipMemcpyDeviceToHost);
  hip_last_error_slot() = (int)e;
  if (e != hipSuccess) return SLM_ERR_LAUNCH;
  *err = (int32_t)v;
  return SLM_OK;
}

SLM_API int slm_allreduce(const slm_ar_args* a, void* stream) {
  ArParams p;
  const int rc = ar_fill(a, &p);
  if (rc != SLM_OK) return rc;
  hip_clear_error();
  const int rows = p.one_shot ? (int)p.M : p.rpr;  // rows a rank reduces: one workgroup each
  const int nb = rows < SLM_AR_MAX_BLOCKS ? rows : SLM_AR_MAX_BLOCKS;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  ar_dispatch(a->dtype, a->residual != nullptr, a->world, [&](auto t, auto f, auto w) {
    using TT = decltype(t);
    hipLaunchKernelGGL((xgmi_allreduce_kernel<TT, decltype(f)::value, decltype(w)::value>),
                       dim3((unsigned)nb), dim3(256), 0, st, p);
  });
  return hip_check_launch();
}

SLM_API int slm_allreduce_simulate(const slm_ar_args* ranks, int32_t world, void* stream) {
  if (!ranks || world < 2 || world > SLM_AR_MAX_RANKS) return SLM_ERR_INVALID_ARG;
  ArSimParams sp;
  for (int r = 0; r < world; ++r) {
    if (ranks[r].world != world || ranks[r].rank != r) return SLM_ERR_INVALID_ARG;
    if (ranks[r].M != ranks[0].M || ranks[r].H != ranks[0].H || ranks[r].dtype != ranks[0].dtype ||
        (ranks[r].residual != nullptr) != (ranks[0].residual != nullptr))
      return SLM_ERR_INVALID_ARG;
    const int rc = ar_fill(&ranks[r], &sp.r[r]);
    if (rc != SLM_OK) return rc;
  }
  hip_clear_error();
  // every workgroup of every rank must be resident at once (they wait for each other): at most one
  // workgroup per CU
  for (int r = 1; r < world; ++r)
    if (sp.r[r].one_shot != sp.r[0].one_shot) return SLM_ERR_INVALID_ARG;  // every rank, one mode
  int nb = sp.r[0].one_shot ? (int)sp.r[0].M : sp.r[0].rpr;
  const int cap = 256 / world < SLM_AR_MAX_BLOCKS ? 256 / world : SLM_AR_MAX_BLOCKS;
  if (nb > cap) nb = cap;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  ar_dispatch(ranks[0].dtype, ranks[0].residual != nullptr, world, [&](auto t, auto f, auto w) {
    using TT = decltype(t);
    hipLaunchKernelGGL((xgmi_allreduce_sim_kernel<TT, decltype(f)::value, decltype(w)::value>),
                       dim3((unsigned)nb, (unsigned)world), dim3(256), 0, st, sp);
  });
  return hip_check_launch();
}

}  // extern "C"
