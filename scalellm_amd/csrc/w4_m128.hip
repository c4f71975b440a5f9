// w4_m128.hip -- int4-weight x fp16/bf16-activation GEMM for 65 <= M <= 128 (round 5).
//
// Same operator as w4.hip (replaces marlin::gptq_gemm, reference gptq_gemm.cu:585-710; Marlin's
// thread-block tile for this regime is gemm_kernel.cuh:49 with thread_m_blocks = 4), same packed
// layout and scale/zero table (w4.hip header), same bit-faithful dequant (W4Dq<T>,
// marlin/numeric_conversion.h:19-62).  This is the row count the two-lane decode step (decode.py,
// DESIGN 3.6) runs ALL of its GEMMs at -- 128 rows per lane -- next to the other lane's attention
// stream, and the row count of every 70B layer at bs = 128.
//
// What the general kernel does there, and why it is slow (round-4 review, weak 2): BM = 64 tiles --
// every weight word is fetched and dequantised by TWO workgroups (one per 64-row block), 19 VALU per
// word for 2 MFMAs; next to the attention stream (one wave per SIMD, ~65 % of the issue slots, high
// priority) the GEMM is bound by its instruction count: 193 VALU per 16 MFMAs.  Its BM = 128 form
// halves that but needs ~200 VGPRs and 64 KiB of LDS -- one workgroup per CU beside the attention
// workgroup (200 VGPRs, 80 KiB) instead of two -- and measured slower in the step.
//
// This kernel: all 128 rows in one workgroup (MT = 4: a weight word is dequantised ONCE, 15 VALU with
// the round-5 fp8-pair route, for 4 MFMAs = 3.75 VALU per MFMA), built to sit TWICE on a CU beside
// the attention workgroup:
//   * K chunks of 64 (one 16-B weight load per lane = exactly one kt block of the packed layout),
//     double-buffered in 2 x 16 KiB of LDS (128 rows x 128 B, XOR-swizzled: slot ^ ((row >> 1) & 7)
//     makes the 16-lane groups of ds_read_b128 hit 16 distinct bank quads);
//   * <= 152 VGPRs (64 accumulators + a WD-deep weight ring + 16 staging registers), so two of its
//     waves fit a SIMD next to a 200-VGPR attention wave (200 + 2 x 152 = 504 <= 512);
//   * buffer resources for every address (32-bit offsets, no 64-bit VALU address math, branch-free
//     predication of rows >= M), all VMEM visible to the compiler and issued in steady-state order:
//     activations of chunk c + 1 first, then the weight / scale words of chunk c + WD, so the counted
//     wait in front of the LDS stores leaves the whole weight ring in flight (VMEM returns in order).
// Epilogues as in w4.hip: 16-bit store (+ bias), fp32 split-K slabs (reduce kernel or the consumer,
// SLM_W4_DEFER_REDUCE), SiLU*mul on (gate, up) tile pairs held by adjacent waves (SLM_W4_SILU_MUL).
#include "w4_common.h"

namespace slm {

constexpr int M128_KC = 64;               // K chunk of the main loop
constexpr int M128_BUF = 128 * 128;       // one activation buffer: 128 rows x 64 k x 2 B
constexpr uint32_t M128_OOB = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t m128_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// NG: scale groups per 64-deep chunk (1: group size >= 64, 2: group size 32)
// WD: weight ring depth in chunks (= unroll of the chunk loop; the host guarantees chunks % WD == 0)
// AD: activation look-ahead in chunks.  1: chunk c + 1 is loaded at the top of chunk c and stored to LDS at its
//     bottom (16 staging VGPRs; the counted wait in front of the stores sees loads issued ~one chunk = 16 MFMAs
//     ago -- short of an L2 round trip under load).  2: chunk c + 2 is loaded at the top of chunk c and sits in
//     registers for a whole chunk before it is stored (two staging sets, 32 VGPRs): the wait at the bottom of
//     chunk c is for loads issued a full chunk earlier, and -- VMEM returns in order -- the weight loads issued
//     behind them get two chunks of slack instead of one.
// KW: waves per column tile (1 / 2).  2 (round 5): a 512-thread workgroup whose wave pairs (w, w + 4) share a column
//     tile and split every 64-deep chunk between them -- wave w takes k-steps 0, 1 (the first 8 bytes of the lane's
//     weight vector), wave w + 4 k-steps 2, 3 -- each with its own accumulators, summed once through LDS at the end.
//     Same tile, same bytes, same instructions in total; but a launch whose grid gives a CU ONE workgroup (gate_up at
//     M = 128: 224 tiles on 256 CUs) now has two waves per SIMD to overlap each other's LDS / VMEM waits and MFMA
//     phases instead of one.
// CT: column tiles (of 32) per workgroup: 4 (128 columns, 256 KW threads) or 8 (256 columns, 512 threads, KW = 1).
//     What a CU ingests per output tile is the 128-row activation panel ONCE PER WORKGROUP plus the weights; at
//     M = 128 the panel (128 x K x 2 B) is 4 x the weights of a 128-column workgroup, and the L2 -> CU path
//     (~21 B/clk per CU) -- not HBM, not the matrix pipe -- is what a 70B-sized layer waits for (gate_up:
//     448 workgroups x 2.5 MB = 1.17 GB = 91 us at 12.9 TB/s of aggregate ingest, 144 us measured).  Eight
//     column tiles share one panel: 0.70 GB.
// ADMA (round 5): the activation chunk goes L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: a wave instruction
//     writes 1 KiB = 8 rows of the 128-B-row image linearly, so the XOR swizzle is applied on the SOURCE side:
//     lane l of instruction i fetches row 8i + l/8, 16-B slot (l % 8) ^ ((row >> 1) & 7)) instead of through
//     staging VGPRs and ds_write_b128.  A probe build without any activation staging ran the 70B shapes 10...14 %
//     faster (profiles/r05_m128_nostage_bound.jsonl): that is what this goes after.  The DMA is inline asm
//     (through the builtin hipcc makes every ds_read of the activation image wait for the DMA into the OTHER
//     buffer: attn_tile.hip); the one wait it needs is the counted vmcnt in front of the chunk's barrier -- the
//     weight / scale loads issued behind it in the same iteration stay in flight, as with the register form.
template <typename T, int NG, int WD, int AD, int KW, int CT, bool ADMA = false>
__global__ void __launch_bounds__(64 * CT * KW, (CT == 8 || KW == 2) ? 2 : 3)
w4a16_gemm_m128_kernel(const GemmKParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename Mfma<T>::frag frag_t;
  constexpr int MT = 4;
  static_assert((CT == 4 || CT == 8) && CT * KW <= 8, "workgroups of 256 or 512 threads");
  constexpr int NT = 64 * CT * KW; // threads
  constexpr int AI = 1024 / NT;    // 16-B activation items per thread and chunk
  constexpr int JW = 4 / KW;       // k-steps (weight words) per wave and chunk

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = wave & (CT - 1);  // column tile of the workgroup
  const int kw = wave / CT;        // k part of the chunk (0 when KW == 1)
  const int nb = (int)(blockIdx.x % (unsigned)p.n_nblocks);
  const int ks = (int)(blockIdx.x / (unsigned)p.n_nblocks);
  // split-K ranges are planned in 128-deep units (W4_KC); this loop walks 64-deep chunks
  const int c0 = 2 * ks * p.chunks_per_split;
  const int c1 = 2 * min(p.n_chunks, (ks + 1) * p.chunks_per_split);
  const int nc = c1 - c0;
  const int clast = 2 * p.n_chunks - 1;

  const int n_tiles = (int)(p.N / 32);
  const int gt = nb * CT + ct;
  const bool nvalid = gt < n_tiles;
  const int ntile = nvalid ? gt : n_tiles - 1;  // (clamped: computes on the last valid tile, no store)

  // ---- resources -------------------------------------------------------------------------------
  const __amdgpu_buffer_rsrc_t a_rs = m128_rsrc(p.a, (uint32_t)(((p.M - 1) * p.lda + p.K) * 2));
  const __amdgpu_buffer_rsrc_t w_rs = m128_rsrc(p.wq, (uint32_t)((uint64_t)p.K * p.N / 2));
  const __amdgpu_buffer_rsrc_t sz_rs = m128_rsrc(p.sz, (uint32_t)((uint64_t)p.ks_groups * p.N * 4));

  // activations: thread -> AI x (row, 16-B slot) of the 128 x 64 tile
  uint32_t a_voff[AI], a_lds[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int idx = tid + NT * i;
    const int row = idx >> 3, slot = idx & 7;
    const int rc = row < p.M ? row : (int)p.M - 1;  // rows >= M: clamped duplicates, never stored
    a_voff[i] = (uint32_t)(2 * (rc * (int)p.lda + slot * 8));
    a_lds[i] = (uint32_t)(row * 128 + ((slot ^ ((row >> 1) & 7)) << 4));
  }
  // ADMA: this wave's DI instructions of a chunk (16 in all, 8 rows each)
  constexpr int DI = 16 / (CT * KW);
  static_assert(!ADMA || (AD == 1 && 16 % (CT * KW) == 0), "whole DMA instructions per wave");
  uint32_t d_voff[DI];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  u32x4 a_rs4 = {0u, 0u, 0u, 0u};
  if constexpr (ADMA) {
    const uint64_t ab = (uint64_t)(uintptr_t)p.a;
    a_rs4 = u32x4{(uint32_t)ab, (uint32_t)((ab >> 32) & 0xffffu), (uint32_t)(((p.M - 1) * p.lda + p.K) * 2), 0x00020000u};
#pragma unroll
    for (int i = 0; i < DI; ++i) {
      const int row = 8 * (wave * DI + i) + (lane >> 3);
      const int rc = row < p.M ? row : (int)p.M - 1;
      const int slot = (lane & 7) ^ ((row >> 1) & 7);
      d_voff[i] = (uint32_t)(2 * (rc * (int)p.lda + slot * 8));
    }
  }
  auto a_dma = [&](int c, int buf) {   // chunk c (absolute, 64-deep) -> LDS buffer `buf`
    if constexpr (ADMA) {
      const uint32_t soff = (uint32_t)min(c, clast) * 128u;
#pragma unroll
      for (int i = 0; i < DI; ++i) {
        const uint32_t dst = lds0 + (uint32_t)(buf * M128_BUF + (wave * DI + i) * 1024);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
                     :
                     : "v"(d_voff[i]), "s"(a_rs4), "s"(dst), "s"(soff)
                     : "memory", "m0");
      }
    }
  };
  // this wave's JW words of the lane's 16-B weight vector: bytes [8 kw, 8 kw + 4 JW)
  const uint32_t w_voff = (uint32_t)ntile * 1024u + (uint32_t)lane * 16u + (uint32_t)kw * (4u * JW);
  const uint32_t kt_stride = (uint32_t)n_tiles * 1024u;  // bytes per 64-deep kt block row
  const uint32_t sz_voff = (uint32_t)(ntile * 32 + (lane & 31)) * 4u;
  const uint32_t sz_stride = (uint32_t)p.N * 4u;

  u32x4 areg[ADMA ? 1 : AD][ADMA ? 1 : AI];
  auto a_load = [&](int set, int c) {  // chunk c (absolute, 64-deep) -> staging set
    const uint32_t soff = (uint32_t)min(c, clast) * 128u;
#pragma unroll
    for (int i = 0; i < (ADMA ? 0 : AI); ++i)
      areg[set][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, (int)a_voff[i], (int)soff, 0));
  };
  auto a_store = [&](int set, int buf) {
#pragma unroll
    for (int i = 0; i < (ADMA ? 0 : AI); ++i) *reinterpret_cast<u32x4*>(smem + buf * M128_BUF + a_lds[i]) = areg[set][i];
  };
  // weights: plain (cacheable) loads -- the other lane of the decode step re-reads the layer within
  // ~0.4 ms and finds it in the Infinity Cache (w4.hip, round 4)
  struct WVec { uint32_t w[JW]; };
  auto w_load = [&](int c) -> WVec {
    WVec v;
    const int soff = (int)((uint32_t)min(c, clast) * kt_stride);
    if constexpr (KW == 1) {
      const u32x4 t = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rs, (int)w_voff, soff, 0));
      v.w[0] = t.x; v.w[1] = t.y; v.w[2] = t.z; v.w[3] = t.w;
    } else {
      const u32x2 t = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(w_rs, (int)w_voff, soff, 0));
      v.w[0] = t.x; v.w[1] = t.y;
    }
    return v;
  };
  // scale / zero words this wave needs per chunk: one per scale group its k-steps touch
  constexpr int NGW = (NG == 2 && KW == 1) ? 2 : 1;
  auto sz_load = [&](int c, int g) -> uint32_t {
    const uint32_t k = (uint32_t)min(c, clast) * 64u + (uint32_t)(KW == 2 ? kw : g) * 32u;
    const uint32_t grp = p.gs_shift >= 30 ? 0u : (k >> p.gs_shift);
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(sz_rs, (int)sz_voff, (int)(grp * sz_stride), 0);
  };

  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

  WVec wring[WD];
  uint32_t szr[WD][NGW];
  frag_t bfrag[2];

  auto dequant = [&](const WVec& wv, const uint32_t (&sz)[NGW], int jj) -> frag_t {  // jj: word of THIS wave
    const W4Dq<T> dq(sz[NGW == 2 ? (jj >> 1) : 0]);
    uint32_t o[4];
    dq.word(wv.w[jj], o);
    const u32x4 packed = {o[0], o[1], o[2], o[3]};
    return __builtin_bit_cast(frag_t, packed);
  };

  // ---- prologue: ring filled in steady-state order, chunk c0 staged ------------------------------
  if (nc > 0) {
    if constexpr (ADMA) a_dma(c0, 0); else a_load(0, c0);
#pragma unroll
    for (int d = 0; d < WD; ++d) {
      wring[d] = w_load(c0 + d);
#pragma unroll
      for (int g = 0; g < NGW; ++g) szr[d][g] = sz_load(c0 + d, g);
    }
    a_store(0, 0);
    if constexpr (AD == 2) a_load(1, c0 + 1);   // (sub-iteration u finds chunk c + 1 in set (u + 1) & 1)
    bfrag[0] = dequant(wring[0], szr[0], 0);   // (the compiler's wait for ring slot 0 drains the DMA issued in front of it too)
  }
  if constexpr (ADMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int mrow = lane & 31, kh = lane >> 5;
  const uint32_t fr_base = (uint32_t)(mrow * 128);
  const uint32_t fr_x = (uint32_t)((mrow >> 1) & 7);
  const uint32_t slot0 = (uint32_t)(2 * JW * kw + kh);   // 16-B slot of this wave's first k-step

  for (int cb = 0; cb < nc; cb += WD) {
#pragma unroll
    for (int u = 0; u < WD; ++u) {
      const int c = c0 + cb + u;        // this chunk; its weights sit in ring slot u
      const int buf = u & 1;            // (WD is even: the buffer parity is static too)
      // (past the range: clamped reloads, stored but never read)
      if constexpr (ADMA) a_dma(c + 1, buf ^ 1);   // (the other buffer: last read one barrier ago)
      else if constexpr (AD == 2) a_load(u & 1, c + 2); else a_load(0, c + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int jj = 0; jj < JW; ++jj) {
        const int cur = jj & 1, nxt = cur ^ 1;
        const uint32_t slot = slot0 + 2u * jj;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const uint32_t off = (uint32_t)(buf * M128_BUF + m * 32 * 128) + fr_base + ((slot ^ fr_x) << 4);
          const frag_t af = __builtin_bit_cast(frag_t, *reinterpret_cast<const u32x4*>(smem + off));
          acc[m] = Mfma<T>::run(af, bfrag[cur], acc[m]);
        }
        // the next k-step's B fragment in the shadow of the MFMAs above
        if (jj < JW - 1) {
          bfrag[nxt] = dequant(wring[u], szr[u], jj + 1);
        } else {
          const int un = (u + 1) % WD;
          bfrag[nxt] = dequant(wring[un], szr[un], 0);
          // ring slot u is free (its last word was dequantised under the previous k-step): re-issue it WD
          // chunks ahead, AFTER this iteration's activation loads (older in the vmcnt queue)
          __builtin_amdgcn_sched_barrier(0);
          wring[u] = w_load(c + WD);
#pragma unroll
          for (int g = 0; g < NGW; ++g) szr[u][g] = sz_load(c + WD, g);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      a_store(AD == 2 ? ((u + 1) & 1) : 0, buf ^ 1);
      // ADMA: chunk c + 1 has landed; the 1 + NGW weight / scale loads issued behind it stay in flight
      if constexpr (ADMA) {
        if constexpr (NGW == 2) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      }
      __syncthreads();
    }
  }

  // ---- KW == 2: the k parts of a column tile meet in LDS (two passes of two row tiles: 4 x 2 x 4 KiB = the
  //      32 KiB of the activation buffers); waves 0-3 carry on with the sums
  if constexpr (KW == 2) {
    float* xs = reinterpret_cast<float*>(smem) + ct * (2 * 16 * 64);
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      if (kw == 1) {
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
          for (int r = 0; r < 16; ++r) xs[(m2 * 16 + r) * 64 + lane] = acc[2 * ps + m2][r];
      }
      __syncthreads();
      if (kw == 0) {
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[2 * ps + m2][r] += xs[(m2 * 16 + r) * 64 + lane];
      }
      __syncthreads();
    }
  }
  const bool owner = kw == 0;   // the wave that holds the tile's sums

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  if (p.silu && p.split_k == 1) {
    // SLM_W4_SILU_MUL: column tiles are (gate, up) pairs held by waves (0, 1) and (2, 3): the up wave
    // hands its T-rounded tile to the gate wave through the (now idle) activation buffers
    const uint16_t* bias = reinterpret_cast<const uint16_t*>(p.bias);
    uint16_t* ex = reinterpret_cast<uint16_t*>(smem) + (ct >> 1) * (MT * 1024);
    if (owner && (ct & 1)) {
      const float bu = bias ? lo_f32<T>((uint32_t)bias[ntile * 32 + (lane & 31)]) : 0.f;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) ex[(m * 16 + r) * 64 + lane] = pack1<T>(acc[m][r] + bu);
    }
    __syncthreads();
    if (!owner || (ct & 1) || !nvalid) return;
    const int64_t gcol = (int64_t)ntile * 32 + (lane & 31), ocol = (int64_t)(ntile >> 1) * 32 + (lane & 31);
    const float bg = bias ? lo_f32<T>((uint32_t)bias[gcol]) : 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const float g = lo_f32<T>((uint32_t)pack1<T>(acc[m][r] + bg));
        const float uu = lo_f32<T>((uint32_t)ex[(m * 16 + r) * 64 + lane]);
        if (row < p.M) reinterpret_cast<uint16_t*>(p.c)[row * p.ldc + ocol] = pack1<T>(silu_mul1(g, uu));
      }
    }
    return;
  }
  if (!owner || !nvalid) return;
  const int64_t n = (int64_t)ntile * 32 + (lane & 31);
  float bv = 0.f;
  if (p.split_k == 1 && p.bias) bv = lo_f32<T>((uint32_t)reinterpret_cast<const uint16_t*>(p.bias)[n]);
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row < p.M) {
        if (p.split_k == 1)
          reinterpret_cast<uint16_t*>(p.c)[row * p.ldc + n] = pack1<T>(acc[m][r] + bv);
        else
          p.part[((int64_t)ks * p.M + row) * p.N + n] = acc[m][r];
      }
    }
  }
}

template <typename T, int KW, int CT, bool ADMA>
static void launch_m128_t(const GemmKParams& kp, int ng, int wd, int n_blocks, hipStream_t st) {
  const dim3 grid((unsigned)n_blocks), blk(64 * CT * KW);
  const size_t lds = 2 * M128_BUF;
#define SLM_M128(NGG, WDD) \
  hipLaunchKernelGGL((w4a16_gemm_m128_kernel<T, NGG, WDD, 1, KW, CT, ADMA>), grid, blk, lds, st, kp)
  if (ng == 2) { if (wd == 4) SLM_M128(2, 4); else SLM_M128(2, 2); }
  else { if (wd == 4) SLM_M128(1, 4); else SLM_M128(1, 2); }
#undef SLM_M128
}

// group_size 32 -> two scale groups per 64-deep chunk; chunks (64-deep) per split must be a multiple of wd;
// kw = waves per column tile (1: 256-thread workgroups, 2: 512-thread workgroups with the chunk split in two)
// (The activation look-ahead of two chunks, AD = 2, measured the same as one on every shape and is not built:
//  profiles/r05_m128_70b_shapes.jsonl.)
// ct = column tiles per workgroup (4 / 8; 8 only with kw = 1); n_blocks counts workgroups of ct tiles;
// adma = activations by LDS-DMA (built for the 256-column form)
void launch_gemm_m128(const GemmKParams& kp, int dtype, int group_size, int wd, int kw, int ct, int adma, int n_blocks, hipStream_t st) {
  const int ng = group_size == 32 ? 2 : 1;
  if (dtype == SLM_BF16) {
    if (ct == 8 && adma) launch_m128_t<bf16_tag, 1, 8, true>(kp, ng, wd, n_blocks, st);
    else if (ct == 8) launch_m128_t<bf16_tag, 1, 8, false>(kp, ng, wd, n_blocks, st);
    else if (kw == 2) launch_m128_t<bf16_tag, 2, 4, false>(kp, ng, wd, n_blocks, st);
    else launch_m128_t<bf16_tag, 1, 4, false>(kp, ng, wd, n_blocks, st);
  } else {
    if (ct == 8 && adma) launch_m128_t<f16_tag, 1, 8, true>(kp, ng, wd, n_blocks, st);
    else if (ct == 8) launch_m128_t<f16_tag, 1, 8, false>(kp, ng, wd, n_blocks, st);
    else if (kw == 2) launch_m128_t<f16_tag, 2, 4, false>(kp, ng, wd, n_blocks, st);
    else launch_m128_t<f16_tag, 1, 4, false>(kp, ng, wd, n_blocks, st);
  }
}

}  // namespace slm
