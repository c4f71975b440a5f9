// w4_ws.hip -- int4-weight x fp16/bf16-activation GEMM for M > 128: wave-specialised kernel.
//
// Same operator as w4.hip (replaces marlin::gptq_gemm, reference gptq_gemm.cu:585-710, for the
// large-batch decode / prefill shapes), same packed layout (w4.hip header), same dequant
// (W4Dq<T>: bit-identical to "magic-number int4 -> T, subtract zero, scale", reference
// marlin/numeric_conversion.h:19-62).
//
// Why a second kernel.  On gfx950 one wave issues ~1 instruction per 4 cycles and a
// v_mfma_f32_32x32x16 occupies the matrix pipe for 32 cycles, so a wave that dequantises its own
// weights (28 VALU per 8-weight word for bf16) AND feeds the matrix pipe can hide at most ~5
// other instructions per MFMA (MI355X_MICROARCH.md, "one wave per SIMD").  Measured with the
// single-role kernel (w4.hip, BM = 128): ~10 VALU + 1 ds_read + 1.3 s_waitcnt per MFMA -> 30 % of
// the MFMA peak.  Here each SIMD hosts TWO waves of one 512-thread workgroup with different jobs:
//
//   consumer waves 0-3 (2 x 2 over the 256 x 128 output tile, 128 x 64 each = 8 accumulator
//     tiles): per 16-deep k-step 4 A-fragment + 2 B-fragment ds_read_b128 and 8 MFMAs.  They
//     also own the HBM weight stream: one 16-B load per lane per chunk into an 8-deep register
//     ring (8 KiB per wave in flight), handed to the producers through a small raw-weight LDS
//     ring -- the consumers issue no other VMEM, so their vmcnt waits stay counted and deep;
//   producer waves 4-7: stage the activation chunk by LDS-DMA (global_load_lds_dwordx4, no VGPR
//     round trip: the VGPR -> LDS store path, ~80 B/clk per CU, was the measured bottleneck of
//     the first version) and dequantise the raw weights into ready-made MFMA B fragments in LDS.
//     Their only VMEM is the DMA, so "wait for my DMA" never waits behind an HBM weight load.
//
// LDS: 3 stages x (A 256 rows x 64 k = 32 KiB, swizzled 16-B slots; B 4 n-tiles x 4 k-steps x
// 1 KiB fragments = 16 KiB) = 144 KiB + raw ring 2 x 4 x (1 KiB weights + 512 B scale words).
// One s_barrier per 64-deep chunk.  In iteration i the consumers compute chunk i from stage i%3
// (prefetching the first fragments of stage (i+1)%3, published one barrier earlier) and write raw
// chunk i+3; the producers read raw chunk i+2 (published one barrier earlier) and fill stage
// (i+2)%3, which the consumers released at the end of iteration i-1.
#include "w4_common.h"

namespace slm {

constexpr int WS_KC = 64;
constexpr int WS_A_BYTES = 256 * 128;      // 256 rows x 64 k x 2 B
constexpr int WS_B_BYTES = 4 * 4 * 1024;   // [n-tile][k-step][lane][16 B]
constexpr int WS_STAGE_BYTES = WS_A_BYTES + WS_B_BYTES;
constexpr int WS_STAGES = 3;
constexpr int WS_RAW_TILE = 1024 + 512;    // per n-tile: 64 x 16 B weights, 2 x 64 x 4 B scale words
constexpr int WS_RAW_BYTES = 4 * WS_RAW_TILE;
constexpr int WS_RAW_BASE = WS_STAGES * WS_STAGE_BYTES;
constexpr int WS_WD = 8;                   // weight ring depth (chunks in flight per consumer wave)
static_assert(WS_RAW_BASE + 2 * WS_RAW_BYTES == W4_WS_LDS_BYTES, "LDS size");
static_assert(W4_WS_LDS_BYTES <= 160 * 1024, "LDS capacity");

// Bare barrier with an explicit wait: __syncthreads() would add s_waitcnt vmcnt(0) lgkmcnt(0) and
// drain the consumers' prefetched fragment reads and in-flight weight loads.  Every wave waits for
// exactly what it publishes (LDS operations of one wave complete in order).
#define WS_WAIT_BARRIER(waitstr) asm volatile("s_waitcnt " waitstr "\n\ts_barrier" ::: "memory")

// NGC: scale groups per 64-deep chunk (2 for group 32, else 1)
template <typename T, int NGC, int EXP = 0>
__global__ void __launch_bounds__(512, 2) w4a16_gemm_ws_kernel(const GemmKParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename Mfma<T>::frag frag_t;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  const int nb = bid % p.n_nblocks;
  bid /= p.n_nblocks;
  const int mb = bid % p.n_mblocks;
  const int ks = bid / p.n_mblocks;
  const int64_t m0 = (int64_t)mb * 256;
  const int64_t n_tiles = p.N / 32;

  // chunk range of this split, in 64-deep chunks (the plan counts 128-deep units)
  const int c0 = 2 * ks * p.chunks_per_split;
  const int c1 = 2 * min(p.n_chunks, (ks + 1) * p.chunks_per_split);
  const int n = c1 - c0;                               // >= 2
  const int n_iter = (n + WS_WD - 1) / WS_WD * WS_WD;  // main-loop iterations (both roles)
  const int last = c1 - 1;
  auto clampc = [&](int c) { return c < last ? c : last; };

  if (wave >= 4) {
    // =============================== producer ===============================
    const int pw = wave - 4;
    // A staging by LDS-DMA (1 KiB = 8 rows x 128 B per wave-instruction, lane l lands at LDS base
    // + 16*l): this wave owns rows 64*pw .. 64*pw+63 of every chunk.  The XOR swizzle is applied
    // on the GLOBAL side: the lane that lands in physical slot ps of row r fetches logical slot
    // ps ^ ((r >> 1) & 7), so the consumers' ds_read_b128 stay conflict-free.
    const char* abase = reinterpret_cast<const char*>(p.a);
    const char* a_ptr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = (pw * 8 + i) * 8 + (lane >> 3);
      const int slot = (lane & 7) ^ ((row >> 1) & 7);
      const int64_t m = m0 + row;
      const int64_t mc = m < p.M ? m : p.M - 1;  // rows >= M: clamped loads, never stored
      a_ptr[i] = abase + 2 * (mc * p.lda + slot * 8);
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    auto a_dma = [&](int c, int stage) {
      const int64_t coff = (int64_t)c * (WS_KC * 2);
      const uint32_t dst = lds0 + stage * WS_STAGE_BYTES + pw * 8192;
      asm volatile(
          "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\t"
          "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
          "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
          "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
          "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\t"
          "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, off\n\t"
          "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, off\n\t"
          "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %7, off\n\t"
          :
          : "v"(a_ptr[0] + coff), "v"(a_ptr[1] + coff), "v"(a_ptr[2] + coff), "v"(a_ptr[3] + coff),
            "v"(a_ptr[4] + coff), "v"(a_ptr[5] + coff), "v"(a_ptr[6] + coff), "v"(a_ptr[7] + coff),
            "s"(dst)
          : "memory", "scc");
    };

    WS_WAIT_BARRIER("lgkmcnt(0)");  // iteration -3: the consumers wrote raw chunk 0
    int stage = 0;
    for (int it = -2; it < n_iter; ++it) {
      const int cc = it + 2;  // chunk (relative to c0) produced in this slot
      if (cc < n) {
        char* sbase = smem + stage * WS_STAGE_BYTES;
        if constexpr (!(EXP & 2)) a_dma(c0 + cc, stage);  // first: the DMA has this whole slot to land
        const char* raw = smem + WS_RAW_BASE + (cc & 1) * WS_RAW_BYTES + pw * WS_RAW_TILE;
        const u32x4 wv = *reinterpret_cast<const u32x4*>(raw + lane * 16);
        uint32_t szc[NGC];
#pragma unroll
        for (int g = 0; g < NGC; ++g)
          szc[g] = *reinterpret_cast<const uint32_t*>(raw + 1024 + g * 256 + lane * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t word = j == 0 ? wv.x : j == 1 ? wv.y : j == 2 ? wv.z : wv.w;
          const W4Dq<T> dq(szc[j * NGC / 4]);
          uint32_t o[4];
          if constexpr (EXP & 1) { o[0] = word; o[1] = word >> 1; o[2] = word + 3; o[3] = ~word; }
          else dq.word(word, o);
          const u32x4 packed = {o[0], o[1], o[2], o[3]};
          *reinterpret_cast<u32x4*>(sbase + WS_A_BYTES + (((pw * 4 + j) * 64 + lane) << 4)) = packed;
        }
      } else if (cc < n_iter) {
        // tail slots (n is not a multiple of the ring depth): the consumers' loop body is
        // unconditional, so give it zero B fragments; the A stage keeps older data of the same rows
        char* sbase = smem + stage * WS_STAGE_BYTES;
        const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<u32x4*>(sbase + WS_A_BYTES + (((pw * 4 + j) * 64 + lane) << 4)) = zero;
      }
      stage = stage == WS_STAGES - 1 ? 0 : stage + 1;
      WS_WAIT_BARRIER("vmcnt(0) lgkmcnt(0)");  // DMA landed, B fragments written
    }
    return;
  }

  // ================================= consumer =================================
  const int mh = wave >> 1, nh = wave & 1;
  const int mrow = lane & 31, kh = lane >> 5;

  // ---- weight stream (this wave feeds n-tile `wave` of the block to the producers) ----
  int64_t nt = (int64_t)nb * 4 + wave;
  if (nt >= n_tiles) nt = n_tiles - 1;  // clamped duplicate work, never stored
  u32x4 wreg[WS_WD];
  uint32_t szreg[WS_WD][NGC];
  auto w_load = [&](int c, u32x4& w, uint32_t (&sz)[NGC]) {
    const uint32_t* wp = p.wq + (((int64_t)c * n_tiles + nt) * 64 + lane) * 4;
    w = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp));
#pragma unroll
    for (int g = 0; g < NGC; ++g) {
      const int64_t grp = ((int64_t)c * WS_KC + g * (WS_KC / NGC)) >> p.gs_shift;
      sz[g] = p.sz[grp * p.N + nt * 32 + (lane & 31)];
    }
  };
  // raw chunk r (relative) -> LDS ring slot r & 1, then refill the register slot WS_WD chunks ahead
  auto raw_put = [&](int r, u32x4& w, uint32_t (&sz)[NGC]) {
    char* raw = smem + WS_RAW_BASE + (r & 1) * WS_RAW_BYTES + wave * WS_RAW_TILE;
    *reinterpret_cast<u32x4*>(raw + lane * 16) = w;
#pragma unroll
    for (int g = 0; g < NGC; ++g) *reinterpret_cast<uint32_t*>(raw + 1024 + g * 256 + lane * 4) = sz[g];
    // refill AFTER the old value is consumed (pinned) so the slot keeps its physical registers
    __builtin_amdgcn_sched_barrier(0);
    w_load(clampc(c0 + r + WS_WD), w, sz);
    __builtin_amdgcn_sched_barrier(0);
  };
#pragma unroll
  for (int d = 0; d < WS_WD; ++d) {
    w_load(clampc(c0 + d), wreg[d], szreg[d]);
    __builtin_amdgcn_sched_barrier(0);
  }

  f32x16 acc[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

  // per-lane LDS offsets: A row (mh*4 + i)*32 + mrow, slot (2*kstep + kh) ^ swizzle;  B fragment
  const int swz = (mrow >> 1) & 7;
  const int a_row_off = (mh * 128 + mrow) * 128;
  const int b_off = WS_A_BYTES + ((nh * 2 * 4 * 64 + lane) << 4);
  frag_t afr[2][4], bfr[2][2];
  auto load_frags = [&](int stage, int kstep, frag_t (&af)[4], frag_t (&bf)[2]) {
    const char* sbase = smem + stage * WS_STAGE_BYTES;
    const int aoff = a_row_off + (((2 * kstep + kh) ^ swz) << 4);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      af[i] = __builtin_bit_cast(frag_t, *reinterpret_cast<const u32x4*>(sbase + aoff + i * 32 * 128));
#pragma unroll
    for (int j = 0; j < 2; ++j)
      bf[j] = __builtin_bit_cast(
          frag_t, *reinterpret_cast<const u32x4*>(sbase + b_off + ((j * 4 + kstep) << 10)));
  };

  // iterations -3, -2, -1: raw chunks 0, 1, 2 (the producers start two chunks ahead)
  raw_put(0, wreg[0], szreg[0]);
  WS_WAIT_BARRIER("lgkmcnt(0)");
  raw_put(1, wreg[1], szreg[1]);
  WS_WAIT_BARRIER("lgkmcnt(0)");
  raw_put(2, wreg[2], szreg[2]);
  WS_WAIT_BARRIER("lgkmcnt(0)");  // B0: chunks 0 and 1 are staged

  load_frags(0, 0, afr[0], bfr[0]);
  int stage = 0;
  for (int base = 0; base < n_iter; base += WS_WD) {
#pragma unroll
    for (int u = 0; u < WS_WD; ++u) {
      const int i = base + u;
      const int nstage = stage == WS_STAGES - 1 ? 0 : stage + 1;
      {
#pragma unroll
        for (int kstep = 0; kstep < 4; ++kstep) {
          const int cur = kstep & 1, nxt = cur ^ 1;
          // raw chunk i+3 goes out under the MFMAs of k-step 1 (not at the head of the iteration,
          // where its ds_write latency would sit between the barrier and the first MFMA)
          if (kstep == 1) raw_put(i + 3, wreg[(u + 3) % WS_WD], szreg[(u + 3) % WS_WD]);
          if (kstep < 3) load_frags(stage, kstep + 1, afr[nxt], bfr[nxt]);
          else load_frags(nstage, 0, afr[nxt], bfr[nxt]);  // published one barrier ago
#pragma unroll
          for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[j][i4] = Mfma<T>::run(bfr[cur][j], afr[cur][i4], acc[j][i4]);  // C^T tile
          // pin: the 6 LDS reads of the next k-step spread under the 8 MFMAs of this one
#pragma unroll
          for (int q = 0; q < 6; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
        // the raw-chunk ds_writes are older than the fragment reads of k-steps 2, 3 and the
        // prefetch: leaving the 6 prefetch reads in flight still guarantees the writes are done
        WS_WAIT_BARRIER("lgkmcnt(6)");
      }
      stage = nstage;
    }
  }

  // ---- epilogue.  The MFMAs ran with the operands swapped (weights as the A operand), so every
  // accumulator tile is C^T: this lane holds token m = tile row base + (lane & 31) and the 16
  // columns n = 32 t + (r & 3) + 8 (r >> 2) + 4 (lane >> 5) -- four consecutive columns per
  // r >> 2, i.e. one 8-byte (bf16/fp16) or 16-byte (fp32 partial) store instead of four 2-byte ones.
  const bool wide = ((p.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.c) & 7) == 0);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int64_t t = (int64_t)nb * 4 + nh * 2 + j;
    if (t >= n_tiles) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t ncol = t * 32 + 8 * q + 4 * (lane >> 5);
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (p.split_k == 1 && p.bias) {
        const u32x2 b = *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(p.bias) + ncol);
        bv[0] = lo_f32<T>(b.x); bv[1] = hi_f32<T>(b.x);
        bv[2] = lo_f32<T>(b.y); bv[3] = hi_f32<T>(b.y);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t row = m0 + (mh * 4 + i) * 32 + (lane & 31);
        if (row >= p.M) continue;
        const float v0 = acc[j][i][4 * q + 0], v1 = acc[j][i][4 * q + 1];
        const float v2 = acc[j][i][4 * q + 2], v3 = acc[j][i][4 * q + 3];
        if (p.split_k == 1) {
          uint16_t* dst = reinterpret_cast<uint16_t*>(p.c) + row * p.ldc + ncol;
          u32x2 o;
          o.x = pack2<T>(v0 + bv[0], v1 + bv[1]);
          o.y = pack2<T>(v2 + bv[2], v3 + bv[3]);
          if (wide) {
            *reinterpret_cast<u32x2*>(dst) = o;
          } else {
            dst[0] = (uint16_t)(o.x & 0xffffu); dst[1] = (uint16_t)(o.x >> 16);
            dst[2] = (uint16_t)(o.y & 0xffffu); dst[3] = (uint16_t)(o.y >> 16);
          }
        } else {
          const f32x4 o = {v0, v1, v2, v3};
          *reinterpret_cast<f32x4*>(p.part + ((int64_t)ks * p.M + row) * p.N + ncol) = o;
        }
      }
    }
  }
}

template <typename T, int NGC, int EXP = 0>
static void launch_ws(const GemmKParams& kp, int n_blocks, hipStream_t st) {
  auto kfn = w4a16_gemm_ws_kernel<T, NGC, EXP>;
  static bool opted = false;  // > 64 KiB of dynamic LDS has to be opted into once per kernel
  if (!opted) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)W4_WS_LDS_BYTES);
    opted = true;
  }
  hipLaunchKernelGGL(kfn, dim3((unsigned)n_blocks), dim3(512), W4_WS_LDS_BYTES, st, kp);
}

void launch_gemm_ws(const GemmKParams& kp, int dtype, int ng, int n_blocks, hipStream_t st) {
  // ng = scale groups per 128 of K (w4.hip plan): 4 for group 32 -> 2 per 64-deep chunk
  if (dtype == SLM_BF16) {
    if (ng == 4) launch_ws<bf16_tag, 2>(kp, n_blocks, st);
    else {
      const char* ev = getenv("SLM_W4_EXP");
      const int e = ev ? atoi(ev) : 0;
      switch (e) {
#define X(E) case E: launch_ws<bf16_tag, 1, E>(kp, n_blocks, st); break;
        X(1) X(2) X(3)
#undef X
        default: launch_ws<bf16_tag, 1>(kp, n_blocks, st);
      }
    }
  } else {
    if (ng == 4) launch_ws<f16_tag, 2>(kp, n_blocks, st);
    else launch_ws<f16_tag, 1>(kp, n_blocks, st);
  }
}

}  // namespace slm
