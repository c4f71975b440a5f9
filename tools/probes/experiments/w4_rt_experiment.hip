// w4_rt.hip -- int4-weight x bf16-activation GEMM for 32 < M <= 512: the REGISTER-TILE kernel.
//
// Same operator as w4.hip (replaces marlin::gptq_gemm, reference gptq_gemm.cu:585-710, on the
// decode shapes of BASELINE configs[1] (bs = 256) and configs[3] (Llama-3-70B, bs = 128)), same
// packed layout and scale/zero table (w4.hip header), post-scaled numerics of the small-M kernels.
//
// Why another tile kernel.  What bounds the M = 128..256 GEMMs here is not the matrix pipe:
//  * w4.hip (BM = 128, one column tile per wave): 28 dequant VALU + 4 ds_read_b128 per 4 MFMAs --
//    7 VALU per MFMA is the whole issue budget of a SIMD (one VALU / MFMA issue per 4 cycles, an
//    MFMA holds the pipe 32), and 1 KiB of LDS read per MFMA is the LDS's whole bandwidth;
//  * w4_ws.hip (256 x 128, dequantised B fragments handed over through LDS): 0.75 fragment reads per
//    MFMA + the producers' B writes + the A DMA = ~144 B/clk of LDS traffic against 128.
// This kernel keeps the weights OUT of LDS and the dequant cheap:
//  * one wave per SIMD (4 waves, up to 512 VGPRs each), each owning a 128 x 64 output block = 4 row
//    blocks x 2 column tiles = 8 accumulator tiles: per 16-deep k-step 4 A-fragment reads feed 8
//    MFMAs (0.5 KiB of LDS per MFMA), the B fragments are unpacked in registers from the wave's own
//    weight stream (register ring, 2 chunks ahead);
//  * post-scaled form: the MFMA consumes magic + q (7 VALU per 8-weight word, 1.75 per MFMA); per
//    scale group the partial tile is folded into the accumulator with one fma per element (2 per
//    MFMA).  Together with the activation group sums (below) ~4.5 VALU per MFMA: MFMA-bound;
//  * the zero-point term sum_g X_g[m] * (-(magic + z) s)_g[n] rides the matrix pipe: ONE bf16 MFMA per
//    tile and group whose 16 k-slots hold the products of a 3-way bf16 split of X (24 bits) with a
//    2-way split of the constant (16 bits, exact) -- fp32-accurate, +12.5 % matrix-pipe time.  X_g of
//    the workgroup's 128 rows: wave w sums row block w from its own fragments (v_dot2 against ones)
//    and the four waves exchange them through 1 KiB of LDS one chunk later;
//  * activations: LDS-DMA (buffer_load_dwordx4 ... lds, whole cache lines, swizzled on the global
//    side), three 32-KiB chunk buffers, issued two chunks ahead; one s_barrier per 128-deep chunk.
//
// All VMEM of a chunk is issued in one burst at the end of the body two chunks earlier: weights x 4
// and scales x 2 (builtins, visible to hipcc), then DMA x 8 (asm, invisible).  The explicit wait for the
// DMA counts everything (vmcnt(14) = one burst may stay in flight); hipcc's own waits for the weight
// registers count only the visible loads and therefore over-wait by the DMA instructions issued
// after them -- which, two chunks deep, have long landed.
//
// Group size >= 128 (any multiple, per-channel); 32 / 64 stay on the older kernels.  bf16 only:
// fp16 has packed VALU and is served well enough by the pre-scaled kernels.
//
// Build flags: no SLP vectoriser (as w4_ks.hip), and VGPR-form MFMAs -- the accumulators are touched by
// the VALU at every chunk boundary (rescale), so they must live in arch VGPRs; hipcc's default puts
// them in AGPRs and pays v_accvgpr_read + write around every multiply (1158 copies in the first build).
// hipcc-flags: -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form
#include "w4_common.h"

namespace slm {

constexpr int RT_A_BUF = 128 * 256;  // one 128-deep chunk of 128 rows

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rt_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// truncating 3-way split of an fp32 into bf16 pieces (hi + mid + lo == x to 24 bits)
__device__ __forceinline__ void rt_split3(float x, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
  const uint32_t xb = __builtin_bit_cast(uint32_t, x);
  const float fh = __builtin_bit_cast(float, xb & 0xffff0000u);
  const float r1 = x - fh;  // exact
  const uint32_t r1b = __builtin_bit_cast(uint32_t, r1);
  const float fm = __builtin_bit_cast(float, r1b & 0xffff0000u);
  const float r2 = r1 - fm;  // exact
  hi = xb >> 16;
  mid = r1b >> 16;
  lo = __builtin_bit_cast(uint32_t, r2) >> 16;
}

// RBW: row blocks per wave.  4: four waves (one per SIMD), each 128 x 64.  2: eight waves (two per
// SIMD), each 64 x 64 -- the partner wave's MFMAs cover this wave's chunk-boundary work (rescale,
// splits, VMEM issue), which a single in-order wave can only serialise (measured: 48 % MFMA-busy
// with four waves even with every load disabled).
template <bool SILU, int RBW>
__global__ void __launch_bounds__(RBW == 4 ? 256 : 512, 2) w4a16_gemm_rt_kernel(const GemmKParams p) {
  constexpr int NW = 16 / RBW;              // waves per workgroup
  constexpr int DPW = 32 / NW;              // DMA instructions per wave and chunk
  constexpr int BURST = DPW + 4 + 2;        // VMEM operations per wave and chunk
  // THREE separate LDS objects for the chunk buffers: hipcc tracks LDS-DMA per LDS object (alias
  // scope), so a fragment read of buffer c waits for the DMA into buffer c only -- with one big
  // array every ds_read would wait for the DMA issued two chunks ahead.
  __shared__ __attribute__((aligned(16))) char abuf0[RT_A_BUF];
  __shared__ __attribute__((aligned(16))) char abuf1[RT_A_BUF];
  __shared__ __attribute__((aligned(16))) char abuf2[RT_A_BUF];
  __shared__ float xs[2 * 4 * 32];  // activation sums [chunk parity][row block][row]
  typedef bf16_tag T;
  typedef Mfma<T>::frag frag_t;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  const int nb = bid % p.n_nblocks;
  bid /= p.n_nblocks;
  const int mb = bid % p.n_mblocks;
  const int ks = bid / p.n_mblocks;
  const int m0 = mb * 128;
  const int n_tiles = (int)(p.N / 32);
  const bool kh = lane >= 32;
  const int m = lane & 31;

  const int c0 = ks * p.chunks_per_split;
  const int c1 = min(p.n_chunks, c0 + p.chunks_per_split);
  const int nC = c1 - c0;  // >= 1
  const int clast = c1 - 1;

  // column tiles of this wave (clamped duplicates past N: computed, never stored)
  const int cwv = wave & 3;   // column pair of this wave
  const int rh = wave >> 2;   // row half (RBW == 2), 0 otherwise
  const int rot = RBW == 4 ? cwv : (cwv & 1);  // rotation of the row-block order (own block first)
  const int t0 = nb * 8 + cwv * 2;
  int tcl[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) tcl[t] = min(t0 + t, n_tiles - 1);

  const __amdgpu_buffer_rsrc_t w_rs = rt_rsrc(p.wq, (uint32_t)((uint64_t)p.K * p.N / 2));
  const __amdgpu_buffer_rsrc_t sz_rs = rt_rsrc(p.sz, (uint32_t)((uint64_t)p.ks_groups * p.N * 4));
  const uint32_t w_voff = (uint32_t)lane * 16u;
  const uint32_t sz_voff = (uint32_t)m * 4u;
  const uint32_t kt_stride = (uint32_t)n_tiles * 1024u;
  const uint32_t sz_stride = (uint32_t)p.N * 4u;
  const int cpg_shift = p.gs_shift >= 30 ? 30 : p.gs_shift - 7;  // log2(chunks per scale group), gs >= 128

  // ---- A staging: DMA instruction i of a chunk moves rows 4i .. 4i+3 (wave w: i = 8w .. 8w+7, i.e.
  // row block w); lane l -> row 4i + (l >> 4), LDS position l & 15 holds octet (l & 15) ^ (row & 15).
  // The builtin (not inline asm): every VMEM operation of the kernel is then visible to hipcc and
  // its counted vmcnt waits for the weight registers are exact.
  const __amdgpu_buffer_rsrc_t a_rs = rt_rsrc(p.a, (uint32_t)(((p.M - 1) * p.lda + p.K) * 2));
  uint32_t dma_voff[DPW];
#pragma unroll
  for (int i = 0; i < DPW; ++i) {
    const int row = 4 * (DPW * wave + i) + (lane >> 4);
    const int64_t mr = (int64_t)m0 + row;
    const int64_t rc = mr < p.M ? mr : p.M - 1;  // rows >= M: clamped duplicates, never stored
    dma_voff[i] = (uint32_t)(2 * rc * p.lda + (((lane & 15) ^ (row & 15)) << 4));
  }
  typedef __attribute__((address_space(3))) void* lds_ptr;
  auto abuf = [&](int buf) -> char* { return buf == 0 ? abuf0 : buf == 1 ? abuf1 : abuf2; };
  auto dma_chunk = [&](int c, int buf) __attribute__((always_inline)) {
    const uint32_t soff = (p.ks_dbg & 1) ? 0x80000000u : (uint32_t)min(c, clast) * 256u;  // probe bit 0: no activation traffic
#pragma unroll
    for (int i = 0; i < DPW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (lds_ptr)(abuf(buf) + (DPW * wave + i) * 1024), 16,
                                               (int)dma_voff[i], (int)soff, 0, 0);
  };
  // fragment (row block rb, row m, octet 2j + h): row = 32 rb + m -> row group 8 rb + (m >> 2).
  // Wave w walks the row blocks in the rotated order w, w+1, w+2, w+3 (mod 4): its accumulator
  // index k stands for row block (k + w) & 3, so its OWN row block -- whose activation sums it owes
  // the workgroup -- is always fragment 0 (no selection, no branch in the k-step).
  const uint32_t fr_lane = (uint32_t)((m >> 2) * 1024 + (m & 3) * 256);
  const uint32_t fr_x = (uint32_t)(((m & 15) ^ (kh ? 1 : 0)) << 4);
  auto frag_read = [&](int buf, int k, int j) -> frag_t {
    const uint32_t off = (uint32_t)((rh * RBW + ((k + rot) & (RBW - 1))) * 8192) + fr_lane + ((uint32_t)(32 * j) ^ fr_x);
    return __builtin_bit_cast(frag_t, *reinterpret_cast<const u32x4*>(abuf(buf) + off));
  };

  // ---- weight / scale ring: 3 chunks x 2 tiles x 2 half chunks
  u32x4 wr[3][2][2];
  uint32_t szr[3][2];
  auto w_burst = [&](int c, int slot) {
    const uint32_t cc = (uint32_t)min(c, clast);
    const uint32_t wdbg = (p.ks_dbg & 2) ? 0x80000000u : 0u;  // probe bit 1: no weight traffic
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
        wr[slot][t][h] = __builtin_bit_cast(
            u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                       w_rs, (int)w_voff, (int)(((cc * 2 + h) * kt_stride + (uint32_t)tcl[t] * 1024u) | wdbg), 2));
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
      szr[slot][t] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(
          sz_rs, (int)sz_voff, (int)((cc >> cpg_shift) * sz_stride + (uint32_t)tcl[t] * 128u), 0);
  };

  if (tid < 256) xs[tid] = 0.f;  // (2 x 4 x 32 = 256 sums: chunk 0 reads a "previous chunk" of zeros; published by the first barrier)
  // prologue: bursts of chunks 0 and 1, in the steady-state order (weights + scales, then the DMA)
  w_burst(c0, 0);
  __builtin_amdgcn_sched_barrier(0);
  dma_chunk(c0, 0);
  w_burst(c0 + 1, 1);
  __builtin_amdgcn_sched_barrier(0);
  dma_chunk(c0 + 1, 1);

  // ONE accumulator set, kept in units of the CURRENT chunk's scale: true partial sum = s_cur * acc.
  // At a chunk boundary acc *= s_prev / s_cur (one VALU multiply per element; a separate per-group
  // partial tile would double the accumulator registers past the 256 arch VGPRs the VALU can reach
  // -- the 512-register form keeps half of them in AGPRs and pays three copies per element).
  f32x16 acc[RBW][2];
#pragma unroll
  for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][t][r] = 0.f;

  uint32_t magic_v = W4Magic<T>::bits;
  asm volatile("" : "+v"(magic_v));
  uint32_t mask_s = 0x000F000Fu;
  asm volatile("" : "+s"(mask_s));

  float sc_prev[2] = {1.f, 1.f}, zm_prev[2] = {0.f, 0.f};  // scale / (magic + zero) of the previous chunk

  auto chunk_body = [&](const int i, const int slot) __attribute__((always_inline)) {
    // everything of this chunk's burst (issued two bodies ago) has landed once at most ONE later
    // burst (14 VMEM operations) is still in flight; the barrier then makes every wave's DMA (and
    // its activation sums of the previous chunk: lgkmcnt) visible and retires the buffer the next
    // DMA overwrites
    if constexpr (BURST == 14) asm volatile("s_waitcnt vmcnt(14) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    static_assert(BURST == 14 || BURST == 10, "explicit vmcnt");
    float sc[2], zmc[2], ratio[2];
    frag_t cfr[2], xfr[RBW];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      W4Magic<T>::decode(szr[slot][t], sc[t], zmc[t]);
      // running sums -> this chunk's scale (ratio == 1 inside a wider group; chunk 0: acc == 0).  A
      // zero scale (an all-zero group) must not poison the ratio: such a column restarts from 0.
      ratio[t] = sc[t] != 0.f ? sc_prev[t] * __builtin_amdgcn_rcpf(sc[t]) : 0.f;
      // zero-point term of the PREVIOUS chunk in units of this scale: both factors of
      // X[m] * (-(magic + z)_prev * ratio) split three ways into bf16 pieces, the six significant
      // products in k-slots 0..5 of half 0 of ONE bf16 MFMA per tile (chunk 0: zm_prev == 0, X == 0)
      uint32_t ch, cm, cl3;
      rt_split3(-zm_prev[t] * ratio[t], ch, cm, cl3);
      u32x4 v = {ch | (cm << 16), cl3 | (ch << 16), cm | (ch << 16), 0u};  // c: hi mid lo hi mid hi
      if (kh) v = u32x4{0u, 0u, 0u, 0u};
      cfr[t] = __builtin_bit_cast(frag_t, v);
    }
#pragma unroll
    for (int k = 0; k < RBW; ++k) {
      const float x = xs[(((i + 1) & 1) * 4 + rh * RBW + ((k + rot) & (RBW - 1))) * 32 + m];
      uint32_t xh, xm, xl;
      rt_split3(x, xh, xm, xl);
      u32x4 v = {xh | (xh << 16), xh | (xm << 16), xm | (xl << 16), 0u};    // x: hi hi hi mid mid lo
      if (kh) v = u32x4{0u, 0u, 0u, 0u};
      xfr[k] = __builtin_bit_cast(frag_t, v);
    }
    auto unpack2 = [&](int j, frag_t (&bf)[2]) __attribute__((always_inline)) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const u32x4 wv = wr[slot][t][j >> 2];
        const uint32_t word = (j & 3) == 0 ? wv.x : (j & 3) == 1 ? wv.y : (j & 3) == 2 ? wv.z : wv.w;
        uint32_t o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t x = q == 0 ? word : word >> (4 * q);
          o[q] = (x & mask_s) | magic_v;
        }
        const u32x4 packed = {o[0], o[1], o[2], o[3]};
        bf[t] = __builtin_bit_cast(frag_t, packed);
      }
    };
    float xsum0 = 0.f, xsum1 = 0.f;
    // Manual software pipeline, pinned per k-step: the A fragments and the unpacked B fragments of
    // k-step j + 1 are produced in the region of k-step j, where hipcc can weave them between the 8
    // MFMAs (an in-order wave overlaps VALU with the matrix pipe only if they alternate in program
    // order).  Without the pins the scheduler hoists the fragment reads of the WHOLE chunk (spills).
    frag_t afn[RBW], bfn[2];
#pragma unroll
    for (int k = 0; k < RBW; ++k) afn[k] = frag_read(slot, k, 0);
    unpack2(0, bfn);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      frag_t af[RBW], bf[2];
#pragma unroll
      for (int k = 0; k < RBW; ++k) af[k] = afn[k];
      bf[0] = bfn[0]; bf[1] = bfn[1];
      if (j < 7) {
#pragma unroll
        for (int k = 0; k < RBW; ++k) afn[k] = frag_read(slot, k, j + 1);
        unpack2(j + 1, bfn);
      }
      // activation sums of row block `wave` = fragment 0 (the other three come from the other waves)
      {
        const u32x4 own = __builtin_bit_cast(u32x4, af[0]);
        xsum0 = dot2<T>(own.x, 0x3F803F80u, xsum0);
        xsum1 = dot2<T>(own.y, 0x3F803F80u, xsum1);
        xsum0 = dot2<T>(own.z, 0x3F803F80u, xsum0);
        xsum1 = dot2<T>(own.w, 0x3F803F80u, xsum1);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int k = 0; k < RBW; ++k) {
          if (j == 0) {
            // chunk boundary, tile by tile so that the rescale of tile n + 1 runs under the MFMAs of tile n
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[k][t][r] *= ratio[t];
            acc[k][t] = Mfma<T>::run(xfr[k], cfr[t], acc[k][t]);
          }
          acc[k][t] = Mfma<T>::run(af[k], bf[t], acc[k][t]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // burst of chunk i + 2, at the END of the body: hipcc's waits for the weight registers count only
    // the loads it can see, so the (invisible) DMA must never be the newest thing in flight when a
    // weight register is first used -- issued here it is a whole chunk old by then.  Its buffer held
    // chunk i - 1, which every wave finished before this body's barrier.
    __builtin_amdgcn_sched_barrier(0);
    w_burst(c0 + i + 2, (slot + 2) % 3);
    __builtin_amdgcn_sched_barrier(0);
    dma_chunk(c0 + i + 2, (slot + 2) % 3);
    // publish this wave's row-block sums (both k halves added) for everybody's zero-point term
    {
      const float xs_half = xsum0 + xsum1;
      const float xfull = xs_half + __shfl_xor(xs_half, 32, 64);
      // own row block = rh * RBW + rot; with eight waves the column pairs 2, 3 duplicate 0, 1 and stay silent
      if (!kh && (RBW == 4 || cwv < 2)) xs[((i & 1) * 4 + rh * RBW + rot) * 32 + m] = xfull;
    }
    zm_prev[0] = zmc[0]; zm_prev[1] = zmc[1];
    sc_prev[0] = sc[0]; sc_prev[1] = sc[1];
  };

  // chunks in rounds of three so that buffer / ring slot indices are compile-time
  const int n_round = (nC + 2) / 3;
  int i = 0;
  for (int rd = 0; rd < n_round; ++rd) {
    chunk_body(i, 0);
    ++i;
    if (i < nC) chunk_body(i, 1);  // (workgroup-uniform: the barriers inside stay matched)
    ++i;
    if (i < nC) chunk_body(i, 2);
    ++i;
  }
  // zero-point term of the last chunk (units of the last scale: the constant is exact)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    uint32_t ch, cm, cl3;
    rt_split3(-zm_prev[t], ch, cm, cl3);
    u32x4 cv = {ch | (cm << 16), cl3 | (ch << 16), cm | (ch << 16), 0u};
    if (kh) cv = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < RBW; ++k) {
      const float x = xs[(((nC - 1) & 1) * 4 + rh * RBW + ((k + rot) & (RBW - 1))) * 32 + m];
      uint32_t xh, xm, xl;
      rt_split3(x, xh, xm, xl);
      u32x4 xv = {xh | (xh << 16), xh | (xm << 16), xm | (xl << 16), 0u};
      if (kh) xv = u32x4{0u, 0u, 0u, 0u};
      acc[k][t] = Mfma<T>::run(__builtin_bit_cast(frag_t, xv), __builtin_bit_cast(frag_t, cv), acc[k][t]);
    }
  }

  // ---- epilogue: one row block at a time through LDS (the chunk buffers are dead: all DMA drained)
  // to become row-major [32][64 + pad] fp32 per wave, then 16-B stores.  C/D layout of the 32x32 MFMA:
  // col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  Waves 0, 1 stage in buffer 0, waves
  // 2, 3 in buffer 1; the staged block is wave-private (LDS operations of one wave are ordered).
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  constexpr int EP_LD = 68;  // floats per row (+4 pad: conflict-free column writes)
  float* const ep = reinterpret_cast<float*>(wave < 3 ? abuf0 : wave < 6 ? abuf1 : abuf2) + (wave % 3) * (32 * EP_LD);
  const bool final_out = p.split_k == 1;
  float bv[2] = {0.f, 0.f};
  if (p.bias && final_out) {
    const uint16_t* bp = reinterpret_cast<const uint16_t*>(p.bias);
#pragma unroll
    for (int t = 0; t < 2; ++t) bv[t] = lo_f32<T>((uint32_t)bp[(int64_t)tcl[t] * 32 + m]);
  }
  constexpr int OUT_COLS = SILU ? 32 : 64;   // columns of the wave's output block
  constexpr int LPR2 = OUT_COLS / 4;         // lanes per row at 4 columns per lane
  constexpr int RPP = 64 / LPR2;             // rows per pass
  const int cl = (lane % LPR2) * 4;
  const int64_t col0 = SILU ? (int64_t)(t0 >> 1) * 32 : (int64_t)t0 * 32;
  const int64_t n_out = SILU ? p.N / 2 : p.N;
#pragma unroll
  for (int k = 0; k < RBW; ++k) {
    const int rb = rh * RBW + ((k + rot) & (RBW - 1));  // the row block accumulator k stands for
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + (kh ? 4 : 0);
      const float v0 = fmaf(acc[k][0][r], sc_prev[0], bv[0]), v1 = fmaf(acc[k][1][r], sc_prev[1], bv[1]);
      if constexpr (SILU) {  // (gate, up) = this wave's two tiles: same lane, same register
        ep[row * EP_LD + m] = silu_mul_acc<T>(v0, v1);
      } else {
        ep[row * EP_LD + m] = v0;
        ep[row * EP_LD + 32 + m] = v1;
      }
    }
#pragma unroll
    for (int r0 = 0; r0 < 32; r0 += RPP) {
      const int row = r0 + lane / LPR2;
      const f32x4 v = *reinterpret_cast<const f32x4*>(ep + row * EP_LD + cl);
      const int64_t mr = (int64_t)m0 + rb * 32 + row;
      const int64_t col = col0 + cl;
      if (mr < p.M && col < n_out) {
        if (final_out) {
          u32x2 o2;
          o2.x = pack2<T>(v.x, v.y);
          o2.y = pack2<T>(v.z, v.w);
          *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(p.c) + mr * p.ldc + col) = o2;
        } else {
          *reinterpret_cast<f32x4*>(p.part + ((int64_t)ks * p.M + mr) * p.N + col) = v;
        }
      }
    }
  }
}

bool gemm_rt_supported(int64_t M, int64_t K, int64_t N, int64_t group_size, int dtype, int64_t lda,
                       int64_t ldc) {
  if (dtype != SLM_BF16) return false;
  if (group_size < 128 || N % 64 || ldc % 4) return false;
  if (K * N / 2 >= ((int64_t)1 << 32) || (K / group_size) * N * 4 >= ((int64_t)1 << 32)) return false;
  return ((M - 1) * lda + K) * 2 < ((int64_t)1 << 31);
}

void launch_gemm_rt(const GemmKParams& kp, int n_blocks, hipStream_t st) {
  const bool four = (kp.ks_dbg & 16) != 0;  // probe: the four-wave form
  if (kp.silu && kp.split_k == 1) {  // (with split-K the reduce kernel applies SiLU*mul to the summed slabs)
    if (four) hipLaunchKernelGGL((w4a16_gemm_rt_kernel<true, 4>), dim3((unsigned)n_blocks), dim3(256), 0, st, kp);
    else hipLaunchKernelGGL((w4a16_gemm_rt_kernel<true, 2>), dim3((unsigned)n_blocks), dim3(512), 0, st, kp);
  } else {
    if (four) hipLaunchKernelGGL((w4a16_gemm_rt_kernel<false, 4>), dim3((unsigned)n_blocks), dim3(256), 0, st, kp);
    else hipLaunchKernelGGL((w4a16_gemm_rt_kernel<false, 2>), dim3((unsigned)n_blocks), dim3(512), 0, st, kp);
  }
}

}  // namespace slm
