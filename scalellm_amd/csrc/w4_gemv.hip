// w4_gemv.hip -- int4-weight x fp16/bf16-activation GEMV for M <= 4 tokens (decode at batch 1-4).
//
// Same operator as w4.hip (replaces marlin::gptq_gemm, reference gptq_gemm.cu:585-710, on the
// single-sequence decode shapes), same packed layout and scale/zero table (w4.hip header).
//
// With 1-4 tokens the matrix core has nothing to do (31 of 32 MFMA rows idle) and what is left is a
// pure HBM stream of the packed weights; the kernel is built around instructions per weight word
// and bytes in flight, not FLOPs:
//   * no MFMA, no per-chunk barrier: each wave owns one 32-column tile over its own K slice and
//     accumulates with v_dot2_f32_{bf16,f16} (two weights per instruction, fp32 accumulate) on the
//     raw magic-number values (magic + q, exact in T; unpack = shift + v_and_or_b32 per pair);
//     the affine part is applied per scale group: sum_k x_k s (q_k - z) = s (T - (magic + z) X_g);
//   * the activations (at most 4 x K values) are staged ONCE per workgroup in LDS together with
//     their per-group sums X_g; the main loop reads them as broadcast ds_read_b128 (all lanes of a
//     half-wave share the address) -- so the only VMEM in the loop is the weight stream, every
//     wave keeps an 8-chunk register ring (8 KiB) in flight with exact counted waits;
//   * K is split across the waves of a workgroup (8 waves = TW column tiles x KS K-slices) and
//     reduced through LDS at the end: no global split-K, no second launch.
// Numerics: fp32 accumulation of exact products, affine correction in fp32 (the post-scaled form of
// the other small-M kernels): within the reference tests' GEMM tolerance, not bit-identical to
// "dequantise to T, then multiply".
#include "w4_common.h"
#include "tuning.h"

namespace slm {

struct GemvParams {
  const void* a;
  const uint32_t* wq;
  const uint32_t* sz;
  const void* bias;
  void* c;
  int64_t M, K, N, lda, ldc;
  int gs_shift;  // log2(group size); 30 = per-channel
  int tw, ks;    // column tiles x K slices per workgroup (tw * ks = 8)
  int n64;       // K / 64
  int splits;    // K splits across workgroups (> 1: fp32 partial slabs to `part`, c not written)
  int n_wgs;     // workgroups per split
  float* part;   // [splits][M][N]
  // norm prologue (GemmKParams): norm_weight == NULL -> activations come from `a`
  const void* norm_x; const float* norm_part; int norm_splits; float norm_eps;
  const void* norm_res_in; void* norm_res_out; const void* norm_weight; void* norm_out;
  int silu;      // SLM_W4_SILU_MUL: tw is even, tiles (2j, 2j+1) = (gate, up), c is [M, N/2]
};

// weight ring depth (64-deep chunks per wave).  Measured alternatives for the norm-prologue kernel
// (Llama-3-8B bs=1 step, ring 8 issued first by every wave = 2.28 ms): ring 16 -> 2.36 ms (loads
// return in order, so the prologue's own loads wait behind a twice as long burst); the four norm
// waves issuing their ring only after row 0's loads landed -> 2.31 ms (their stream starts late).
constexpr int GV_RING = 8;

// NGC: scale groups per 64-deep chunk (2 for group 32, 1 otherwise); MT: token rows (1, 2, 4)
// REFILL: the K slice of a wave is longer than the ring (otherwise every chunk is preloaded)
// NORM: RMSNorm prologue (slm_w4a16_gemv_norm); its own instantiation so that the plain kernel keeps
// its register count
template <typename T, int NGC, int MT, bool REFILL, bool NORM>
__global__ void __launch_bounds__(512) w4a16_gemv_kernel(const GemvParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS: A [MT][K] T | X32 [MT][K/32] f32 (activation sums per 32 of K) | reduction [8][MT][32] f32
  // (| NORM: h [K] f32 | 4 f32)
  const int n32 = (int)(p.K / 32);
  uint16_t* a_lds = reinterpret_cast<uint16_t*>(smem);
  float* x_lds = reinterpret_cast<float*>(smem + (size_t)MT * p.K * 2);
  float* red = x_lds + (size_t)MT * n32;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t n_tiles = p.N / 32;

  // ---- this wave's column tile and K slice; its weight ring is issued FIRST so that the HBM
  // latency of the first chunks overlaps the activation staging below ----
  const int tw_i = wave % p.tw, ks_i = wave / p.tw;
  const int split_id = (int)blockIdx.x / p.n_wgs, wg = (int)blockIdx.x % p.n_wgs;
  int64_t nt = (int64_t)wg * p.tw + tw_i;
  const bool nvalid = nt < n_tiles;
  if (!nvalid) nt = n_tiles - 1;  // clamped duplicate work, never stored
  const int per = (p.n64 + p.ks * p.splits - 1) / (p.ks * p.splits);
  const int c0 = (split_id * p.ks + ks_i) * per;
  const int c1 = min(p.n64, c0 + per);
  const int nC = max(c1 - c0, 0);
  const int last = max(c1 - 1, 0);
  auto clampc = [&](int c) { return c < last ? c : last; };

  // per-lane base pointers once; per load only a wave-uniform 32-bit byte offset (the host checks
  // that the packed weights and the scale table are < 4 GiB)
  const char* wlane = reinterpret_cast<const char*>(p.wq + (nt * 64 + lane) * 4);
  const char* szlane = reinterpret_cast<const char*>(p.sz + nt * 32 + (lane & 31));
  const uint32_t wstride = (uint32_t)(n_tiles * 1024);  // bytes per 64-deep chunk
  const uint32_t szstride = (uint32_t)(p.N * 4);        // bytes per scale group
  const int cpg_shift = p.gs_shift >= 30 ? 30 : (p.gs_shift > 6 ? p.gs_shift - 6 : 0);  // log2(chunks per group)
  u32x4 wreg[GV_RING];
  uint32_t szreg[GV_RING][NGC];
  auto w_load = [&](int c, u32x4& w, uint32_t (&sz)[NGC]) {
    const uint32_t cc = (uint32_t)clampc(c);
    w = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wlane + cc * wstride));
#pragma unroll
    for (int g = 0; g < NGC; ++g) {
      const uint32_t grp = NGC == 2 ? cc * 2 + g : (cc >> cpg_shift);
      sz[g] = *reinterpret_cast<const uint32_t*>(szlane + grp * szstride);
    }
  };
#pragma unroll
  for (int d = 0; d < GV_RING; ++d) {
    w_load(c0 + d, wreg[d], szreg[d]);
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- phase 0: activations -> LDS (rows >= M replicate the last row; never stored) ----
  if constexpr (NORM) {
    // Norm prologue: a[m, :] = rms_norm(x[m, :] + residual[m, :]) * weight, with x possibly given
    // as the fp32 split-K slabs of the producing GEMM -- the arithmetic, the thread -> column
    // mapping and the reduction tree of rms_norm_kernel (glue.hip), executed by the first 256
    // threads of EVERY workgroup (a 4096-wide row is 8 KiB of L2 hits: cheaper to recompute per
    // workgroup than to launch a kernel for it).  Workgroup 0 also stores the updated residual
    // (and, if asked, the normalised row) -- into buffers nobody reads during this launch.
    // h = x + residual (fp32) waits in LDS between the two passes (a thread reads back only what
    // it wrote itself); two vectors per thread are in flight per trip.
    float* h_lds = red + 8 * MT * 32;  // [K] fp32, one row at a time
    float* nred = h_lds + p.K;         // [4] (dynamic too: static LDS on top of a 160 KiB opt-in fails the launch)
    const int64_t dim = p.K, nvec = dim / 8;
    const bool writer = blockIdx.x == 0;
    auto load_h = [&](int64_t mc, int64_t vi, float (&f)[8]) {
      u32x4 a;
      if (p.norm_part) {
        const f32x4 s0 = splitk_sum4(p.norm_part + mc * dim + vi * 8, p.M * dim, p.norm_splits);
        const f32x4 s1 = splitk_sum4(p.norm_part + mc * dim + vi * 8 + 4, p.M * dim, p.norm_splits);
        a.x = pack2<T>(s0.x, s0.y); a.y = pack2<T>(s0.z, s0.w);
        a.z = pack2<T>(s1.x, s1.y); a.w = pack2<T>(s1.z, s1.w);
      } else {
        a = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(p.norm_x) + mc * dim + vi * 8);
      }
      f[0] = lo_f32<T>(a.x); f[1] = hi_f32<T>(a.x); f[2] = lo_f32<T>(a.y); f[3] = hi_f32<T>(a.y);
      f[4] = lo_f32<T>(a.z); f[5] = hi_f32<T>(a.z); f[6] = lo_f32<T>(a.w); f[7] = hi_f32<T>(a.w);
      if (p.norm_res_in) {
        const u32x4 r = *reinterpret_cast<const u32x4*>(
            reinterpret_cast<const uint16_t*>(p.norm_res_in) + mc * dim + vi * 8);
        f[0] += lo_f32<T>(r.x); f[1] += hi_f32<T>(r.x); f[2] += lo_f32<T>(r.y); f[3] += hi_f32<T>(r.y);
        f[4] += lo_f32<T>(r.z); f[5] += hi_f32<T>(r.z); f[6] += lo_f32<T>(r.w); f[7] += hi_f32<T>(r.w);
      }
    };
    auto keep_h = [&](int m, int64_t mc, int64_t vi, const float (&f)[8], float ss) {
      if (p.norm_res_in && writer && m < p.M) {
        u32x4 w;
        w.x = pack2<T>(f[0], f[1]); w.y = pack2<T>(f[2], f[3]);
        w.z = pack2<T>(f[4], f[5]); w.w = pack2<T>(f[6], f[7]);
        *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(p.norm_res_out) + mc * dim + vi * 8) = w;
      }
      *reinterpret_cast<f32x4*>(h_lds + vi * 8) = f32x4{f[0], f[1], f[2], f[3]};
      *reinterpret_cast<f32x4*>(h_lds + vi * 8 + 4) = f32x4{f[4], f[5], f[6], f[7]};
      return rms_sumsq8(f, ss);
    };
    // the norm weights of the first trip depend on nothing: in flight before the activations
    const uint16_t* nw = reinterpret_cast<const uint16_t*>(p.norm_weight);
    const int64_t tv = tid & 255;
    const u32x4 wpre0 = *reinterpret_cast<const u32x4*>(nw + (tv < nvec ? tv : nvec - 1) * 8);
    const u32x4 wpre1 = *reinterpret_cast<const u32x4*>(nw + (tv + 256 < nvec ? tv + 256 : nvec - 1) * 8);
    for (int m = 0; m < MT; ++m) {
      const int64_t mc = m < p.M ? m : p.M - 1;
      if (tid < 256) {
        float ss = 0.f;
        for (int64_t v0 = tid; v0 < nvec; v0 += 512) {  // ascending vi per thread, as rms_norm_kernel
          const int64_t v1 = v0 + 256;
          float f0[8], f1[8];
          load_h(mc, v0, f0);
          load_h(mc, v1 < nvec ? v1 : v0, f1);  // unconditional: both trips in flight together
          ss = keep_h(m, mc, v0, f0, ss);
          if (v1 < nvec) ss = keep_h(m, mc, v1, f1, ss);
        }
        ss = group_sum<64>(ss);
        if ((tid & 63) == 0) nred[tid >> 6] = ss;
      }
      __syncthreads();
      if (tid < 256) {
        const float tot = nred[0] + nred[1] + nred[2] + nred[3];
        const float rs = rsqrtf(tot / (float)dim + p.norm_eps);
        for (int64_t vi = tid; vi < nvec; vi += 256) {
          const u32x4 wv = vi == tid ? wpre0 : vi == tid + 256 ? wpre1
                                             : *reinterpret_cast<const u32x4*>(nw + vi * 8);
          const f32x4 h0 = *reinterpret_cast<const f32x4*>(h_lds + vi * 8);
          const f32x4 h1 = *reinterpret_cast<const f32x4*>(h_lds + vi * 8 + 4);
          const float f[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
          const u32x4 o = rms_apply8<T>(f, rs, wv);
          *reinterpret_cast<u32x4*>(a_lds + (size_t)m * p.K + vi * 8) = o;
          if (writer && p.norm_out && m < p.M)
            *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(p.norm_out) + mc * dim + vi * 8) = o;
        }
      }
      __syncthreads();  // nred and h_lds are reused by the next row
    }
  } else {
    const int vec_per_row = (int)(p.K / 8);
    for (int v = tid; v < MT * vec_per_row; v += 512) {
      const int m = v / vec_per_row, kk = v - m * vec_per_row;
      const int64_t mc = m < p.M ? m : p.M - 1;
      const u32x4 val = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(p.a) +
                                                         mc * p.lda + (int64_t)kk * 8);
      *reinterpret_cast<u32x4*>(a_lds + (size_t)m * p.K + (size_t)kk * 8) = val;
    }
  }
  __syncthreads();
  // activation sums per 32 of K (fp32): the affine correction needs sum_k x_k over whatever part
  // of a scale group a wave's K slice covers, so the table is kept at the finest granularity
  for (int t = tid; t < MT * n32; t += 512) {
    const uint16_t* src = a_lds + (size_t)t * 32;  // [m][32 b .. 32 b + 31]: rows are K apart = n32 * 32
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(src + k);
      s += (lo_f32<T>(v.x) + hi_f32<T>(v.x)) + (lo_f32<T>(v.y) + hi_f32<T>(v.y)) +
           (lo_f32<T>(v.z) + hi_f32<T>(v.z)) + (lo_f32<T>(v.w) + hi_f32<T>(v.w));
    }
    x_lds[t] = s;
  }
  __syncthreads();

  // ---- phase 1: the weight stream ----
  uint32_t magic_v = W4Magic<T>::bits;
  asm volatile("" : "+v"(magic_v));  // keep it in a VGPR (VOP3 takes no literal on gfx9-family)
  const int kh = lane >> 5;
  float acc[MT], tsum[MT], tsum2[MT], xsum[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] = tsum[m] = tsum2[m] = xsum[m] = 0.f;

  // activation fragments of a chunk (4 k-steps x MT rows, broadcast LDS reads) are fetched one
  // chunk ahead: a read issued right before its dot2s exposes the LDS latency four times per chunk
  u32x4 afr[2][MT][4];
  auto a_fetch = [&](int cabs, u32x4 (&dst)[MT][4]) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        dst[m][j] = *reinterpret_cast<const u32x4*>(a_lds + (size_t)m * p.K + cabs * 64 + j * 16 + kh * 8);
  };
  a_fetch(clampc(c0), afr[0]);
  const int n_iter = (nC + GV_RING - 1) / GV_RING * GV_RING;
  for (int base = 0; base < n_iter; base += GV_RING) {
#pragma unroll
    for (int u = 0; u < GV_RING; ++u) {
      const int i = base + u;
      a_fetch(clampc(c0 + i + 1), afr[(u + 1) & 1]);
      if (i < nC) {
        const int cabs = c0 + i;
        const u32x4 wv = wreg[u];
        // group boundary at the end of this chunk (groups >= 64; groups of 32 end inside it)
        const bool chunk_ends_group = i == nC - 1 || (((cabs + 1) >> cpg_shift) != (cabs >> cpg_shift));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t word = j == 0 ? wv.x : j == 1 ? wv.y : j == 2 ? wv.z : wv.w;
          uint32_t o[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t x = q == 0 ? word : word >> (4 * q);
            asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(o[q]) : "v"(x), "s"(0x000F000Fu), "v"(magic_v));
          }
          // this lane's 8 k values of the k-step: k = 64 c + 16 j + 8 kh + e (the same for all
          // lanes of a half-wave)
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const u32x4 av = afr[u & 1][m][j];
            // two independent chains (dependent dot2 issue back to back would stall the wave)
            float t0 = tsum[m], t1 = tsum2[m];
            t0 = dot2<T>(o[0], av.x, t0);
            t1 = dot2<T>(o[1], av.y, t1);
            t0 = dot2<T>(o[2], av.z, t0);
            t1 = dot2<T>(o[3], av.w, t1);
            tsum[m] = t0;
            tsum2[m] = t1;
          }
          if (j & 1) {  // a 32-deep half chunk is complete: its activation sum
#pragma unroll
            for (int m = 0; m < MT; ++m) xsum[m] += x_lds[m * n32 + cabs * 2 + (j >> 1)];
          }
          const bool grp_end = NGC == 2 ? (j & 1) == 1 : (j == 3 && chunk_ends_group);
          if (grp_end) {
            // acc += s * (T - (magic + z) * X): each half-wave holds half of T (its k values),
            // X is the full sum, so the zero term is applied by the kh = 0 half only
            float sc, zm;
            W4Magic<T>::decode(szreg[u][NGC == 2 ? (j >> 1) : 0], sc, zm);
            const float nzs = kh == 0 ? -zm * sc : 0.f;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
              acc[m] = fmaf(sc, tsum[m] + tsum2[m], fmaf(nzs, xsum[m], acc[m]));
              tsum[m] = 0.f;
              tsum2[m] = 0.f;
              xsum[m] = 0.f;
            }
          }
        }
      }
      // refill AFTER the old value is consumed (pinned): each ring slot keeps its registers
      if constexpr (REFILL) {
        __builtin_amdgcn_sched_barrier(0);
        w_load(c0 + i + GV_RING, wreg[u], szreg[u]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // ---- phase 2: halves of the wave, then the K slices of the workgroup ----
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] += __shfl_xor(acc[m], 32);
  if (lane < 32) {
#pragma unroll
    for (int m = 0; m < MT; ++m) red[(wave * MT + m) * 32 + lane] = acc[m];
  }
  __syncthreads();
  if (p.silu) {
    if (ks_i == 0 && nvalid && lane < 32 && !(tw_i & 1)) {
      const int64_t gcol = nt * 32 + lane, ocol = (nt >> 1) * 32 + lane;
      const uint16_t* bias = reinterpret_cast<const uint16_t*>(p.bias);
      const float bg = bias ? lo_f32<T>((uint32_t)bias[gcol]) : 0.f;
      const float bu = bias ? lo_f32<T>((uint32_t)bias[gcol + 32]) : 0.f;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        float g = 0.f, u = 0.f;
        for (int k = 0; k < p.ks; ++k) {
          g += red[((k * p.tw + tw_i) * MT + m) * 32 + lane];
          u += red[((k * p.tw + tw_i + 1) * MT + m) * 32 + lane];
        }
        if (m < p.M)
          reinterpret_cast<uint16_t*>(p.c)[(int64_t)m * p.ldc + ocol] =
              pack1<T>(silu_mul_acc<T>(g + bg, u + bu));
      }
    }
    return;
  }
  if (p.splits > 1) {  // this split's share of the sum, fp32, for the deferred consumer
    if (ks_i == 0 && nvalid && lane < 32) {
      const int64_t ncol = nt * 32 + lane;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        float s = 0.f;
        for (int k = 0; k < p.ks; ++k) s += red[((k * p.tw + tw_i) * MT + m) * 32 + lane];
        if (m < p.M) p.part[((int64_t)split_id * p.M + m) * p.N + ncol] = s;
      }
    }
    return;
  }
  if (ks_i == 0 && nvalid && lane < 32) {
    const int64_t ncol = nt * 32 + lane;
    float bv = 0.f;
    if (p.bias) bv = lo_f32<T>((uint32_t) reinterpret_cast<const uint16_t*>(p.bias)[ncol]);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float s = 0.f;
      for (int k = 0; k < p.ks; ++k) s += red[((k * p.tw + tw_i) * MT + m) * 32 + lane];
      if (m < p.M) reinterpret_cast<uint16_t*>(p.c)[(int64_t)m * p.ldc + ncol] = pack1<T>(s + bv);
    }
  }
}

template <typename T, int NGC>
static void launch_gemv_m(const GemvParams& gp, int n_wgs, size_t lds, hipStream_t st) {
  // The ring is always refilled (REFILL = true), even when the wave's K slice fits it: measured, the
  // clamped (L2-hit) extra loads are FASTER on the wide layers (gate_up M=1 17.5-19 us vs 23.4 us
  // without them) -- they keep the issue pattern the compiler's counted waits were built for.  The
  // no-refill instantiations were only reachable through a tuning knob and were the ones at the
  // 256-VGPR cap with spills; they are no longer built.
#define SLM_GEMV(MTT)                                                                          \
  do {                                                                                         \
    auto kfn = gp.norm_weight ? w4a16_gemv_kernel<T, NGC, MTT, true, true>                     \
                              : w4a16_gemv_kernel<T, NGC, MTT, true, false>;                   \
    if (lds > 65536) {                                                                         \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),                            \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);       \
    }                                                                                          \
    hipLaunchKernelGGL(kfn, dim3((unsigned)(n_wgs * gp.splits)), dim3(512), lds, st, gp);      \
  } while (0)
  if (gp.M <= 1) SLM_GEMV(1);
  else if (gp.M <= 2) SLM_GEMV(2);
  else SLM_GEMV(4);
#undef SLM_GEMV
}

bool gemv_supported(int64_t M, int64_t K, int64_t group_size, bool norm) {
  if (M < 1 || M > 4) return false;
  const int mt = M <= 1 ? 1 : M <= 2 ? 2 : 4;
  const size_t lds = (size_t)mt * K * 2 + (size_t)mt * (K / 32) * 4 + 8 * mt * 32 * 4 +
                     (norm ? (size_t)K * 4 + 16 : 0);
  (void)group_size;
  return lds <= 160 * 1024 && K % 64 == 0;  // (the < 4 GiB checks are in launch_gemv's caller)
}

static void gemv_shape(int64_t K, int64_t N, bool silu, int& ks, int& tw, int& n_wgs) {
  const int n64 = (int)(K / 64);
  // 8 waves per workgroup = tw column tiles x ks K-slices: enough workgroups to cover the CUs a
  // few times over, but at least 4 chunks (256 of K) per slice
  const int64_t tiles = N / 32;
  ks = 8;
  while (ks > 1 && (n64 / ks < 4 || tiles * ks / 8 > 2048)) ks >>= 1;
  // wide layers: about one workgroup per CU (more column tiles per workgroup, fewer K slices):
  // every workgroup stages the activation vector, so 896 of them cost more than they hide --
  // gate_up 4096 x 28672 at M = 1: 8 slices 19.6 us, 4 slices 16.7, 2 slices 15.5, 1 slice 22.3
  while (ks > 2 && tiles * ks / 8 > 320) ks >>= 1;
  const int forced_ks = tune_get(TUNE_W4_GEMV_KS, 0);
  if (forced_ks == 1 || forced_ks == 2 || forced_ks == 4 || forced_ks == 8) ks = forced_ks;
  if (silu && ks == 8) ks = 4;  // a (gate, up) tile pair has to share the workgroup: tw >= 2
  tw = 8 / ks;
  n_wgs = (int)((tiles + tw - 1) / tw);
}

int gemv_global_splits(int64_t M, int64_t K, int64_t N, bool partials_ok) {
  (void)M;
  if (!partials_ok) return 1;
  const int forced = tune_get(TUNE_W4_SPLITK, 0);
  int ks, tw, n_wgs;
  gemv_shape(K, N, false, ks, tw, n_wgs);
  // measured at M = 1 (rotating weights): o 4096 x 4096 (128 workgroups) 6.6 -> 5.9 us and down
  // 14336 x 4096 (128) 13.4 -> 10.8 us with 2 splits; qkv 4096 x 6144 (192 workgroups) LOSES (7.2 -> 7.9)
  int sp = forced > 0 ? forced : (n_wgs <= 160 ? (n_wgs <= 80 ? 4 : 2) : 1);
  // every (workgroup, in-workgroup slice) keeps at least 4 chunks of K
  while (sp > 1 && (K / 64) / ((int64_t)ks * sp) < 4) sp >>= 1;
  return sp < 1 ? 1 : sp;
}

void launch_gemv(const GemmKParams& kp, int dtype, int ng, hipStream_t st) {
  GemvParams gp;
  gp.a = kp.a; gp.wq = kp.wq; gp.sz = kp.sz; gp.bias = kp.bias; gp.c = kp.c;
  gp.M = kp.M; gp.K = kp.K; gp.N = kp.N; gp.lda = kp.lda; gp.ldc = kp.ldc;
  gp.gs_shift = kp.gs_shift;
  gp.n64 = (int)(kp.K / 64);
  gp.silu = kp.silu;
  int n_wgs;
  gemv_shape(kp.K, kp.N, kp.silu != 0, gp.ks, gp.tw, n_wgs);
  gp.n_wgs = n_wgs;
  gp.splits = kp.split_k > 1 ? kp.split_k : 1;
  gp.part = kp.part;
  gp.norm_x = kp.norm_x; gp.norm_part = kp.norm_part; gp.norm_splits = kp.norm_splits;
  gp.norm_eps = kp.norm_eps; gp.norm_res_in = kp.norm_res_in; gp.norm_res_out = kp.norm_res_out;
  gp.norm_weight = kp.norm_weight; gp.norm_out = kp.norm_out;
  const int mt = kp.M <= 1 ? 1 : kp.M <= 2 ? 2 : 4;
  const size_t lds = (size_t)mt * kp.K * 2 + (size_t)mt * (kp.K / 32) * 4 + 8 * mt * 32 * 4 +
                     (kp.norm_weight ? (size_t)kp.K * 4 + 16 : 0);  // + one fp32 row of the norm prologue
  if (dtype == SLM_BF16) {
    if (ng == 4) launch_gemv_m<bf16_tag, 2>(gp, n_wgs, lds, st);
    else launch_gemv_m<bf16_tag, 1>(gp, n_wgs, lds, st);
  } else {
    if (ng == 4) launch_gemv_m<f16_tag, 2>(gp, n_wgs, lds, st);
    else launch_gemv_m<f16_tag, 1>(gp, n_wgs, lds, st);
  }
}

}  // namespace slm
