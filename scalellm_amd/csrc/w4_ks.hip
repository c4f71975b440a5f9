// w4_ks.hip -- int4-weight x fp16/bf16-activation GEMM for M <= 32: the K-SLICED weight stream.
//
// Same operator as w4.hip (replaces marlin::gptq_gemm, reference gptq_gemm.cu:585-710, on the
// small-batch decode shapes -- BASELINE configs[2]: Llama-3-8B AWQ, bs = 32), same packed layout
// and scale/zero table (w4.hip header), same post-scaled numerics as w4_small.hip.
//
// What rounds 2-3 measured (profiles/r03_ks_*): at M <= 32 the weight stream itself is
// VALU-issue-bound on the int4 unpack (a SIMD issues one VALU/MFMA instruction per 4 cycles: 7 VALU +
// 1 MFMA per 8-weight word = 32 cycles = the MFMA's own pipe time), and w4_small.hip loses another 2x
// around it: activation fragments re-read from LDS per MFMA (LDS-bandwidth-bound), a second MFMA per
// k-step for the activation sums, a workgroup barrier per 128-deep chunk.  This kernel keeps the
// stream at the issue limit and removes what surrounds it:
//
//  * K is split over the WAVES of a workgroup, and each wave keeps the activations of its K slice
//    IN REGISTERS for the whole launch (CW chunks of 128 = 32 CW VGPRs in MFMA A-fragment form).
//    The main loop has no activation staging, no LDS fragment reads and no per-chunk barrier: per
//    KiB of weights it issues one 16-B load, 28 unpack VALU, 4 MFMAs and 8 VALU of group epilogue.
//  * the activations reach the registers through LDS: LDS-DMA (buffer_load_dwordx4 ... lds) of
//    4 rows x 256 B per instruction -- whole cache lines, XOR-swizzled on the global side -- then
//    conflict-free ds_read_b128 fragments.  Loading the fragments straight from global memory
//    (32 rows x 32 B per instruction) costs one L1 tag lookup per 32-B sector: 4x the lookups, measured
//    7.4 k cycles of a 41 k-cycle kernel on gate_up.
//  * a wave walks the column tiles of its workgroup one after the other; the NW partial 32 x 32
//    tiles meet in LDS once per column tile and every wave sums and stores 1/NW of the tile in a
//    fixed order -- bit-reproducible, no global split-K and no reduce launch when NW * CW chunks
//    cover K.  The meeting is PIPELINED and has no s_barrier: a wave publishes its partial tile t
//    (LDS writes + one LDS atomic on an arrival counter), streams tile t + 1, and only then sums
//    tile t -- by which time the other waves have long arrived, so the skew between waves (measured
//    1.0 k of 5.0 k cycles per tile with a barrier) costs nothing.  Four partial slots make the
//    reuse safe without any further synchronisation (argument at the slot arithmetic below).
//  * the zero-point term of the post-scaled form, sum_g X_g[m] * (-(magic + z_g[n]) s_g[n]), is a
//    rank-(K/group) update: it runs on the matrix pipe as exact-fp32 MFMAs (v_mfma_f32_32x32x2_f32,
//    two scale groups per instruction), with the activation group sums X_g taken ONCE per launch
//    from the wave's own fragments (MFMA against a ones fragment: the result lands lane = token).
//  * weights: 8-slot register ring of 1-KiB half chunks (32 VGPRs), each slot refilled right after
//    its last use with the half chunk 8 positions ahead; every VMEM operation of the loop is visible
//    to the compiler and the loop body is straight-line code (disabled stores go out of range of
//    their buffer resource), so the waits are exact counted vmcnt.
//
// Launch shape (plan_gemm): grid = split_k x tile runs; workgroup = NW waves x CW chunks of K,
// `ks_tpw` consecutive column tiles.
//
// Built without the SLP vectoriser: it packs the group epilogue into v_pk_fma_f32 and gathers the
// packed tree at the END of a tile, which keeps every partial tile live (spills) and serialises the
// epilogue behind the MFMAs.
// hipcc-flags: -fno-slp-vectorize
#include "w4_common.h"

namespace slm {

template <typename T>
struct KsOnes;
template <>
struct KsOnes<bf16_tag> {
  static constexpr uint32_t bits = 0x3F803F80u;
};
template <>
struct KsOnes<f16_tag> {
  static constexpr uint32_t bits = 0x3C003C00u;
};

// raw buffer resources: SGPR base + 32-bit offsets (no 64-bit VALU address math per load), and
// out-of-range stores are dropped by the hardware -- rows >= M and tiles past the run need no branch,
// so the loop body is straight-line code and hipcc's vmcnt waits stay exact
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ks_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
constexpr uint32_t KS_OOB = 0x80000000u;  // beyond every resource here (all < 2 GiB or checked < 4 GiB)
constexpr int KS_AUX_NT = 2;              // gfx940+ cache policy bits: sc0 = 1, nt = 2, sc1 = 16

constexpr int KS_SLOTS = 4;  // partial-tile slots in LDS (see the slot arithmetic in the kernel)

// CW: 128-deep chunks of K per wave (1, 2, 4);  NG: scale groups per chunk (1: group >= 128,
// 2: 64, 4: 32);  NW: waves per workgroup (4, 8);  TL: timeline probe (tools/probe_ks_timeline.py)
// MT: 32-row tiles of tokens (1: M <= 32; 2: 33 <= M <= 64, round 4).  With MT = 2 every weight word is
// unpacked ONCE and feeds two MFMAs (one per row tile): the stream that was VALU-issue-bound on the
// unpack at M <= 32 (7 VALU + 1 MFMA per word) becomes 7 VALU + 2 MFMAs -- matrix-pipe-bound -- instead
// of a second 55 us pass or the general kernel's 94-100 us (LDS fragment reads per MFMA, a barrier per
// chunk).  The activations of both row tiles live in registers (64 VGPRs per chunk), which leaves room
// for ONE chunk per wave: K is split over the 8 waves x ceil(K / 1024) workgroups (fp32 slabs, summed
// by the consumer or the reduce kernel).  A 64 x 32 partial tile goes through the LDS meeting as two
// consecutive entries of the slot sequence (2t, 2t + 1).
// (A fragment-major activation layout was measured in round 5 -- 3.6 % -- and is kept only as
// tools/probes/experiments/w4_ks_fragment_major_activations.diff.)
template <typename T, int CW, int NG, int NW, bool TL = false, bool PK = false, int MT = 1>
__global__ void __launch_bounds__(NW * 64, 2) w4a16_gemm_ks_kernel(const GemmKParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename Mfma<T>::frag frag_t;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  constexpr int RD = 8 / (2 * CW);   // column tiles per loop body (one turn of the 8-slot weight ring)
  constexpr int VC = CW * MT;        // activation pieces staged per wave: (chunk, row tile), v = c * MT + mt
  constexpr int PP = 2;              // partial-sum tiles in flight (ping-pong under the epilogue)
  constexpr int NGW = CW * NG;       // scale segments per wave and tile
  constexpr int NP = (NGW + 1) / 2;  // segment pairs = fp32 MFMAs of the zero-point term
  constexpr int WPG = 8 / NG;        // k-steps (weight words) per scale segment
  constexpr int RPW = 16 / NW;       // accumulator registers a wave reduces and stores (2 or 4)
  constexpr int SLOT_FLOATS = NW * 1024;
  constexpr int CNT_OFF = KS_SLOTS * NW * 4096;  // arrival counters behind the partial slots

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_tiles = (int)(p.N / 32);
  const int run = (int)(blockIdx.x % (unsigned)p.n_nblocks);
  const int ks = (int)(blockIdx.x / (unsigned)p.n_nblocks);
  const int nt0 = run * p.ks_tpw;
  const int ntl = min(p.ks_tpw, n_tiles - nt0);  // >= 1
  const int cw0 = (ks * NW + wave) * CW;         // first chunk of this wave's K slice
  const int clast = p.n_chunks - 1;
  const bool kh = lane >= 32;
  // TL (probe instantiation, SLM_W4_KS_DBG & 4): s_memtime stamps of every phase go to `c` instead of
  // the result -- [workgroup][wave][32] u64, read by tools/probe_ks_timeline.py
  auto stamp = [&](int idx) {
    if constexpr (TL) {
      const uint64_t tm = __builtin_amdgcn_s_memtime();
      if (lane == 0 && idx < 32)
        reinterpret_cast<uint64_t*>(p.c)[((int64_t)blockIdx.x * NW + wave) * 32 + idx] = tm;
    }
  };
  stamp(0);

  // arrival counters start at zero; the one workgroup barrier that publishes this sits further down,
  // under the latency of the first loads
  if (tid < KS_SLOTS) reinterpret_cast<uint32_t*>(smem + CNT_OFF)[tid] = 0u;

  // ---- resources (the host checks the sizes: packed weights / scale table < 4 GiB, the rest < 2 GiB)
  const __amdgpu_buffer_rsrc_t w_rs = ks_rsrc(p.wq, (uint32_t)((uint64_t)p.K * p.N / 2));
  const __amdgpu_buffer_rsrc_t sz_rs = ks_rsrc(p.sz, (uint32_t)((uint64_t)p.ks_groups * p.N * 4));
  const bool has_bias = p.bias != nullptr;
  const __amdgpu_buffer_rsrc_t b_rs = ks_rsrc(has_bias ? p.bias : (const void*)p.sz, (uint32_t)(p.N * 2));
  const bool final_out = p.split_k == 1;
  const __amdgpu_buffer_rsrc_t c_rs =
      ks_rsrc(p.c, final_out ? (uint32_t)(((p.M - 1) * p.ldc + (p.silu ? p.N / 2 : p.N)) * 2) : 0u);
  const __amdgpu_buffer_rsrc_t part_rs =
      ks_rsrc(final_out ? nullptr : p.part + (int64_t)ks * p.M * p.N, final_out ? 0u : (uint32_t)(p.M * p.N * 4));

  const uint32_t w_voff = (uint32_t)lane * 16u;
  const uint32_t sz_voff = (uint32_t)(lane & 31) * 4u;
  const uint32_t b_voff = (uint32_t)(lane & 31) * 2u;
  const uint32_t kt_stride = (uint32_t)n_tiles * 1024u;  // bytes per 64-deep kt
  const uint32_t sz_stride = (uint32_t)p.N * 4u;         // bytes per scale group
  const int cpg_shift = p.gs_shift >= 30 ? 30 : (p.gs_shift > 7 ? p.gs_shift - 7 : 0);
  uint32_t woff[CW][2], soff[CW][NG];
#pragma unroll
  for (int c = 0; c < CW; ++c) {
    const uint32_t cc = (uint32_t)min(cw0 + c, clast);  // chunks past K: clamped loads x zero activations
#pragma unroll
    for (int h = 0; h < 2; ++h) woff[c][h] = (cc * 2 + h) * kt_stride + (uint32_t)nt0 * 1024u;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const uint32_t grp = NG > 1 ? cc * NG + g : (cc >> cpg_shift);
      soff[c][g] = grp * sz_stride + (uint32_t)nt0 * 128u;
    }
  }
  const uint32_t w_dbg = (p.ks_dbg & 2) ? KS_OOB : 0u;  // probes: bit 0 = no activation loads, bit 1 = no weight loads
  u32x4 ring[RD][CW][2];
  uint32_t szr[RD][CW][NG];
  uint32_t bsr[RD];
  auto w_load = [&](int t, int c, int h) -> u32x4 {
    const uint32_t tc = (uint32_t)min(t, ntl - 1);  // tiles past the run: clamped duplicates, never stored
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                         w_rs, (int)w_voff, (int)((woff[c][h] + tc * 1024u) | w_dbg), KS_AUX_NT));
  };
  auto sz_load = [&](int t, int c, int g) -> uint32_t {
    const uint32_t tc = (uint32_t)min(t, ntl - 1);
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(sz_rs, (int)sz_voff, (int)(soff[c][g] + tc * 128u), 0);
  };
  auto b_load = [&](int t) -> uint32_t {
    const uint32_t tc = (uint32_t)min(t, ntl - 1);
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(b_rs, (int)b_voff, (int)((uint32_t)(nt0 + tc) * 64u), 0);
  };

  // prologue, in the order the steady state issues (scales and bias of a tile, then its weights):
  // the loop-carried vmcnt waits are merged with this path, so a different order here would make
  // every wait in the loop conservative.  The ring goes first (HBM latency), the activations (L2)
  // queue behind it.
#pragma unroll
  for (int d = 0; d < RD; ++d) {
#pragma unroll
    for (int c = 0; c < CW; ++c) {
#pragma unroll
      for (int g = 0; g < NG; ++g) szr[d][c][g] = sz_load(d, c, g);
    }
    bsr[d] = b_load(d);
#pragma unroll
    for (int c = 0; c < CW; ++c) {
#pragma unroll
      for (int h = 0; h < 2; ++h) ring[d][c][h] = w_load(d, c, h);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  stamp(1);
  {  // probe (SLM_W4_KS_DBG bits 8..14): delay the second wave of every SIMD by 256-cycle steps
    const int nsl = (p.ks_dbg >> 8) & 127;
    if (wave >= NW / 2)
      for (int i = 0; i < nsl; ++i) __builtin_amdgcn_s_sleep(4);
  }

  // ---- activations of this wave's K slice -> A fragments (row = lane & 31, k = 16 j + 8 (lane >> 5) ..+7)
  // through LDS.  Staging buffer b of wave w = its OWN partial-tile slots 2b and 2b + 1 (2 x 4 KiB: one
  // chunk of 32 rows x 256 B), so no other wave ever writes there and the hand-over to the reduce
  // needs no barrier.  DMA instruction i of a chunk moves rows 4i .. 4i+3: lane l -> row 4i + (l >> 4),
  // LDS position l & 15 holds the 16-B octet (l & 15) ^ (row & 15) (swizzle on the global side; the
  // fragment reads below are then conflict-free ds_read_b128).
  frag_t act[MT][CW][8];
  float xa[MT][NP];
  {
    float xg[MT][2 * NP];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int s2 = 0; s2 < 2 * NP; ++s2) xg[mt][s2] = 0.f;
    }
    auto xsum_chunk = [&](int v) {
      const int c = v / MT, mt = v % MT;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        float xs0 = 0.f, xs1 = 0.f;
#pragma unroll
        for (int jj = 0; jj < WPG; ++jj) {
          const u32x4 f = __builtin_bit_cast(u32x4, act[mt][c][g * WPG + jj]);
          xs0 = dot2<T>(f.x, KsOnes<T>::bits, xs0);
          xs1 = dot2<T>(f.y, KsOnes<T>::bits, xs1);
          xs0 = dot2<T>(f.z, KsOnes<T>::bits, xs0);
          xs1 = dot2<T>(f.w, KsOnes<T>::bits, xs1);
        }
        const float xs = xs0 + xs1;
        xg[mt][c * NG + g] = xs + __shfl_xor(xs, 32, 64);
      }
    };
    {
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    // hand-made resource words for the asm DMA: base, base_hi (stride 0), bytes, flags
    const uint64_t abits = reinterpret_cast<uint64_t>(p.a);
    u32x4 a_rs4;
    a_rs4.x = (uint32_t)abits; a_rs4.y = (uint32_t)(abits >> 32) & 0xffffu;
    a_rs4.z = (uint32_t)(((p.M - 1) * p.lda + p.K) * 2); a_rs4.w = 0x00020000u;
    a_rs4.x = __builtin_amdgcn_readfirstlane(a_rs4.x); a_rs4.y = __builtin_amdgcn_readfirstlane(a_rs4.y);
    a_rs4.z = __builtin_amdgcn_readfirstlane(a_rs4.z); a_rs4.w = __builtin_amdgcn_readfirstlane(a_rs4.w);
    uint32_t dma_voff[MT][8];  // per-lane byte offset of DMA instruction i within a chunk column
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = 32 * mt + 4 * i + (lane >> 4);
        const int rc = row < p.M ? row : (int)p.M - 1;  // rows >= M: clamped duplicates, never stored
        dma_voff[mt][i] = (uint32_t)(2 * rc * (int)p.lda + (((lane & 15) ^ (row & 15)) << 4));
      }
    }
    auto stage_base = [&](int b, int i) -> uint32_t {  // LDS byte address of DMA instruction i, buffer b
      return lds0 + (uint32_t)(((2 * b + (i >> 2)) * NW + wave) * 4096 + (i & 3) * 1024);
    };
    auto dma_chunk = [&](int v) {  // piece v = (chunk v / MT, row tile v % MT) into staging buffer v & 1
      const int c = v / MT, mt = v % MT;
      const int cabs = cw0 + c;
      // chunks past K: out-of-range loads write zeros (zero activations x clamped weights = 0)
      const uint32_t a_soff = (cabs <= clast && !(p.ks_dbg & 1)) ? (uint32_t)cabs * 256u : KS_OOB;
      // rows >= M are never stored and MFMA rows are independent: their four-row DMA instructions are
      // skipped (the staging memory then holds whatever was there -- any bit pattern is fine).  At
      // M <= 4 that is 1/8 of the activation traffic and of the LDS fill, the bulk of the prologue.
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (32 * mt + 4 * i < p.M)
          asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds"
                       :
                       : "v"(dma_voff[mt][i]), "s"(stage_base(v & 1, i)), "s"(a_rs4), "s"(a_soff)
                       : "memory", "m0");
    };
    // fragment (row m = lane & 31, octet 2j + h) sits at row group m >> 2, row-in-group m & 3,
    // position (2j + h) ^ (m & 15) = 2j ^ ((m & 15) ^ h)
    const int m = lane & 31;
    const uint32_t fr_lane = (uint32_t)((m >> 4) * NW * 4096 + ((m >> 2) & 3) * 1024 + (m & 3) * 256);
    const uint32_t fr_x = (uint32_t)(((m & 15) ^ (kh ? 1 : 0)) << 4);
    auto frag_read = [&](int v, int j) -> frag_t {
      const uint32_t off = (uint32_t)(((v & 1) * 2 * NW + wave) * 4096) + fr_lane + ((uint32_t)(32 * j) ^ fr_x);
      return __builtin_bit_cast(frag_t, *reinterpret_cast<const u32x4*>(smem + off));
    };
    // activation sums per scale segment: v_dot2 of every packed pair against (1, 1) -- exact products,
    // fp32 accumulate -- then the two k halves of a row (lanes l and l + 32) are added; every lane ends
    // with X[m = lane & 31].  (32 VALU per chunk; eight MFMAs against a ones fragment cost twice the
    // issue time and serialise on the accumulator.)
#pragma unroll
    for (int c0 = 0; c0 < VC; c0 += 2) {
      if (c0 > 0)  // the fragment reads of the previous pair have returned before their buffers are refilled
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      dma_chunk(c0);
      if (c0 + 1 < VC) dma_chunk(c0 + 1);
      if (c0 > 0) {  // group sums of the previous pair run under this pair's DMA
        xsum_chunk(c0 - 2);
        xsum_chunk(c0 - 1);
      }
      stamp(c0 == 0 ? 2 : 31);
      if (c0 == 0)  // counters zeroed (top of the kernel) before anybody can publish a tile
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      // the DMA is invisible to the compiler: explicit wait (also lands the ring, issued before it)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (c0 == 0) {
        // ... and the compiler does not know the ring has landed: without these uses it merges
        // "prologue loads pending" into the loop's vmcnt bookkeeping and every wait of the steady
        // state comes out 4 loads too conservative (measured in the ISA: vmcnt(12) instead of 16)
#pragma unroll
        for (int d = 0; d < RD; ++d) {
#pragma unroll
          for (int c = 0; c < CW; ++c) {
            asm volatile("" : "+v"(ring[d][c][0]), "+v"(ring[d][c][1]));
#pragma unroll
            for (int g = 0; g < NG; ++g) asm volatile("" : "+v"(szr[d][c][g]));
          }
          asm volatile("" : "+v"(bsr[d]));
        }
      }
#pragma unroll
      for (int v = c0; v < c0 + 2 && v < VC; ++v) {
#pragma unroll
        for (int j = 0; j < 8; ++j) act[v % MT][v / MT][j] = frag_read(v, j);
      }
    }
    xsum_chunk(VC >= 2 ? VC - 2 : 0);
    if (VC >= 2) xsum_chunk(VC - 1);
    }
    // the partial-slot writes of tile 0 reuse the staging memory: the fragment reads are done (their
    // values fed the MFMAs above)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        float x0 = xg[mt][2 * q], x1 = xg[mt][2 * q + 1];
        asm volatile("" : "+v"(x0), "+v"(x1));  // values, not array slots: keeps the select off the stack
        xa[mt][q] = kh ? x1 : x0;  // A operand of the fp32 MFMA: k = lane >> 5 picks the segment of the pair
      }
    }
  }

  if constexpr (TL) asm volatile("" : "+v"(xa[0][0]));
  stamp(3);
  uint32_t magic_v = W4Magic<T>::bits;
  asm volatile("" : "+v"(magic_v));  // keep it in a VGPR (not re-materialised as a literal)
  uint32_t mask_s = 0x000F000Fu;
  asm volatile("" : "+s"(mask_s));   // ... and the nibble-pair mask in an SGPR (v_and_or_b32 takes no literal)

  float* const red = reinterpret_cast<float*>(smem);
  const uint32_t cnt_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + CNT_OFF;
  float hold[MT][RPW];  // SLM_W4_SILU_MUL: the gate tile's values wait here for the up tile
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int i = 0; i < RPW; ++i) hold[mt][i] = 0.f;
  }
  const bool silu = p.silu != 0;
  const uint32_t ldc2 = (uint32_t)p.ldc * 2u, n4 = (uint32_t)p.N * 4u;

  // Sum and store tile tp (all NW partials of it have been published or are about to be).
  // Slot arithmetic: tile t lives in slot t & 3.  When a wave A publishes tile t + 1 it has passed
  // the arrival check of tile t - 1, so every wave B has published t - 1; in program order B's
  // reads of slot (t + 1) & 3 = (t - 3) & 3 (its reduce of tile t - 3) precede its publication of
  // t - 1: they are complete.  Three slots would not do (B may still be reducing t - 2).
  // `publish` (tile t, or nothing when acc_pub == nullptr) is issued between the partial reads and
  // the sums of tile tp: the reads go ahead of the 4 KiB of LDS writes instead of queueing behind them
  // (MT = 2: column tile t is the entries t * MT and t * MT + 1 of the slot sequence -- its two row
  // tiles; the argument above holds for the sequence whatever its entries are)
  auto reduce_store = [&](const int tp, const uint32_t braw, const int t_pub, const f32x16* acc_pub) {
    const bool live = tp >= 0 && tp < ntl;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int sidx = tp * MT + mt;
      const uint32_t target = tp >= 0 ? (uint32_t)(NW * ((sidx >> 2) + 1)) : 0u;
      const uint32_t caddr = cnt_lds + (uint32_t)((sidx & 3) * 4);
      uint32_t seen;
      do {  // arrival check: by now (one tile later) it passes at the first look
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen) : "v"(caddr) : "memory");
      } while ((uint32_t)__builtin_amdgcn_readfirstlane(seen) < target);
    }
    stamp(6 + 4 * (tp < 0 ? 0 : tp));
    f32x4 pv4[MT][RPW == 4 ? NW : 1];
    f32x2 pv2[MT][RPW == 2 ? NW : 1];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const float* const theirs = red + ((tp * MT + mt) & 3) * SLOT_FLOATS + (wave * 64 + lane) * RPW;
#pragma unroll
      for (int src = 0; src < NW; ++src) {
        if constexpr (RPW == 4) pv4[mt][src] = *reinterpret_cast<const f32x4*>(theirs + src * 1024);
        else pv2[mt][src] = *reinterpret_cast<const f32x2*>(theirs + src * 1024);
      }
    }
    if (acc_pub) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        // publish the partial tile: [slot][source wave][reducing wave][lane][RPW], then arrive
        const f32x16& acc = acc_pub[mt];
        const int pidx = t_pub * MT + mt;
        float* const mine = red + (pidx & 3) * SLOT_FLOATS + wave * 1024 + lane * RPW;
#pragma unroll
        for (int rg = 0; rg < NW; ++rg) {
          if constexpr (RPW == 4) {
            const f32x4 v = {acc[rg * 4], acc[rg * 4 + 1], acc[rg * 4 + 2], acc[rg * 4 + 3]};
            *reinterpret_cast<f32x4*>(mine + rg * 256) = v;
          } else {
            const f32x2 v = {acc[rg * 2], acc[rg * 2 + 1]};
            *reinterpret_cast<f32x2*>(mine + rg * 128) = v;
          }
        }
        // one lane adds 1 to the tile's counter; DS operations of a wave execute in issue order, so
        // whoever sees the count also sees the partial tile
        const uint32_t paddr = cnt_lds + (uint32_t)((pidx & 3) * 4);
        const uint32_t one = 1u;
        uint64_t ex;
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_add_u32 %1, %2\n\ts_mov_b64 exec, %0"
                     : "=&s"(ex)
                     : "v"(paddr), "v"(one)
                     : "memory");
      }
      stamp(5 + 4 * t_pub);
    }
    // store (straight-line: disabled stores go out of range and are dropped).  C/D layout of the
    // 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int nt = nt0 + tp;
    const uint32_t col = (uint32_t)nt * 32u + (uint32_t)(lane & 31);
    const uint32_t ocol = silu ? (uint32_t)(nt >> 1) * 32u + (uint32_t)(lane & 31) : col;
    const bool c_on = !TL && live && final_out && (!silu || (tp & 1));
    const bool part_on = live && !final_out;
    const float bv = has_bias ? lo_f32<T>(braw) : 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float sum[RPW];
#pragma unroll
      for (int i = 0; i < RPW; ++i) sum[i] = 0.f;
#pragma unroll
      for (int src = 0; src < NW; ++src) {  // fixed order: bit-reproducible
        if constexpr (RPW == 4) {
          sum[0] += pv4[mt][src].x; sum[1] += pv4[mt][src].y; sum[2] += pv4[mt][src].z; sum[3] += pv4[mt][src].w;
        } else {
          sum[0] += pv2[mt][src].x; sum[1] += pv2[mt][src].y;
        }
      }
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        const int r = wave * RPW + i;
        const uint32_t row = (uint32_t)(32 * mt + (r & 3) + 8 * (r >> 2)) + (kh ? 4u : 0u);
        const float v = sum[i] + bv;
        float o = v;
        if (silu) o = silu_mul_acc<T>(hold[mt][i], v);  // wave-uniform branch, VALU only: vmcnt bookkeeping unaffected
        hold[mt][i] = v;
        const uint16_t o16 = pack1<T>(o);
        __builtin_amdgcn_raw_buffer_store_b16(o16, c_rs, (int)(c_on ? row * ldc2 + ocol * 2u : KS_OOB), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, sum[i]), part_rs,
                                              (int)(part_on ? row * n4 + col * 4u : KS_OOB), 0, 0);
      }
    }
    if constexpr (TL) {
      float keep = hold[0][0];
      asm volatile("" : "+v"(keep));
      stamp(7 + 4 * (tp < 0 ? 0 : tp));
    }
  };

  uint32_t bprev = 0u;
  const int n_iter = (ntl + RD - 1) / RD;  // >= 1
  int it = 0;
  do {  // do-while: with a guarded loop hipcc sinks the prologue loads behind the guard
#pragma unroll
    for (int d = 0; d < RD; ++d) {
      const int t = it * RD + d;
      stamp(4 + 4 * t);
      // ---- scales of this tile (loaded one ring turn ago), then their registers take the loads for
      // tile t + RD right away: issued BEFORE this tile's weight refills, so waiting for them at the
      // start of tile t + RD leaves a full ring of weight loads in flight
      float sc[NGW];
      f32x16 acc[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
      }
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        float cz2[2] = {0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int s2 = 2 * q + e;
          if (s2 < NGW) {
            float sv, zm;
            W4Magic<T>::decode(szr[d][s2 / NG][s2 % NG], sv, zm);
            sc[s2] = sv;
            cz2[e] = -zm * sv;  // <= 16 significant bits: exact
          }
        }
        float c0 = cz2[0], c1 = cz2[1];
        asm volatile("" : "+v"(c0), "+v"(c1));  // values, not array slots: keeps the select off the stack
        // zero-point term on the matrix pipe (exact fp32): acc = sum_seg X_seg[m] * (-(magic+z) s)[n]
        const float cz = kh ? c1 : c0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[mt][q], cz, acc[mt], 0, 0, 0);
      }
      const uint32_t braw = bsr[d];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < CW; ++c) {
#pragma unroll
        for (int g = 0; g < NG; ++g) szr[d][c][g] = sz_load(t + RD, c, g);
      }
      bsr[d] = b_load(t + RD);
      __builtin_amdgcn_sched_barrier(0);

      // ---- the weight stream of this tile.  Segment s accumulates in tmp[s & 1] while the scale
      // epilogue of segment s - 1 runs under its MFMAs: exactly two partial tiles are live
      f32x16 tmp[MT][PP];
#pragma unroll
      for (int s2 = 0; s2 < NGW; ++s2) {
        const int c = s2 / NG, g = s2 % NG;
#pragma unroll
        for (int jj = 0; jj < WPG; ++jj) {
          const int j = g * WPG + jj;
          const u32x4 wv = ring[d][c][j >> 2];
          const uint32_t word = (j & 3) == 0 ? wv.x : (j & 3) == 1 ? wv.y : (j & 3) == 2 ? wv.z : wv.w;
          uint32_t o[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            // plain expression on opaque registers, NOT inline asm (hipcc inserts no hazard wait
            // states behind an asm VALU feeding an MFMA: w4_small.hip)
            const uint32_t x = q == 0 ? word : word >> (4 * q);
            o[q] = (x & mask_s) | magic_v;
          }
          const u32x4 packed = {o[0], o[1], o[2], o[3]};
          const frag_t bf = __builtin_bit_cast(frag_t, packed);
          // ONE unpack, MT MFMAs: the same B fragment against each row tile's activations
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            if (jj == 0) {
              f32x16 z;
#pragma unroll
              for (int r = 0; r < 16; ++r) z[r] = 0.f;
              tmp[mt][s2 % PP] = Mfma<T>::run(act[mt][c][j], bf, z);
            } else {
              tmp[mt][s2 % PP] = Mfma<T>::run(act[mt][c][j], bf, tmp[mt][s2 % PP]);
            }
          }
          if (s2 > 0 && jj == (WPG > 1 ? 1 : 0)) {
            const float sv = sc[s2 - 1];
            if constexpr (PK) {
              const f32x2 sv2 = {sv, sv};
#pragma unroll
              for (int r = 0; r < 16; r += 2) {
                f32x2 a2 = {acc[0][r], acc[0][r + 1]};
                const f32x2 t2 = {tmp[0][(s2 - 1) % PP][r], tmp[0][(s2 - 1) % PP][r + 1]};
                a2 = __builtin_elementwise_fma(sv2, t2, a2);
                acc[0][r] = a2.x; acc[0][r + 1] = a2.y;
              }
            } else {
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][r] = fmaf(sv, tmp[mt][(s2 - 1) % PP][r], acc[mt][r]);
              }
            }
            // pin the epilogue HERE (under this segment's MFMAs): without an anchor the scheduler
            // sinks all of a tile's epilogues to the tile end and keeps every partial tile live.
            // The anchor ties acc to the next weight word, so it can neither sink nor hoist.
            if (jj + 1 < WPG || s2 + 1 < NGW) {
              const int jn = jj + 1 < WPG ? j + 1 : ((s2 + 1) % NG) * WPG;
              const int cn = jj + 1 < WPG ? c : (s2 + 1) / NG;
              if constexpr (MT == 2)
                asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(ring[d][cn][jn >> 2]));
              else
                asm volatile("" : "+v"(acc[0]), "+v"(ring[d][cn][jn >> 2]));
            }
          }
          if ((j & 3) == 3) {  // last word of ring slot (c, j >> 2): refill it for tile t + RD
            __builtin_amdgcn_sched_barrier(0);
            ring[d][c][j >> 2] = w_load(t + RD, c, j >> 2);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (s2 == NGW - 1) {
          const float sv = sc[s2];
          if constexpr (PK) {
            const f32x2 sv2 = {sv, sv};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              f32x2 a2 = {acc[0][r], acc[0][r + 1]};
              const f32x2 t2 = {tmp[0][s2 % PP][r], tmp[0][s2 % PP][r + 1]};
              a2 = __builtin_elementwise_fma(sv2, t2, a2);
              acc[0][r] = a2.x; acc[0][r + 1] = a2.y;
            }
          } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[mt][r] = fmaf(sv, tmp[mt][s2 % PP][r], acc[mt][r]);
            }
          }
        }
      }

      // ---- publish this tile; one tile later: sum and store the previous one
      reduce_store(t - 1, bprev, t, acc);
      bprev = braw;
    }
  } while (++it < n_iter);
  reduce_store(n_iter * RD - 1, bprev, 0, nullptr);
}

// 33 <= M <= 64: two row tiles, one chunk per wave, 8 waves (the register budget: see the kernel header)
template <typename T, int NG>
static void launch_ks_mt2(const GemmKParams& kp, int n_blocks, hipStream_t st) {
  hipLaunchKernelGGL((w4a16_gemm_ks_kernel<T, 1, NG, 8, false, false, 2>), dim3((unsigned)n_blocks), dim3(512),
                     w4_ks_lds_bytes(8), st, kp);
}

template <typename T, int CW, int NG, int NW>
static void launch_ks_t(const GemmKParams& kp, int n_blocks, hipStream_t st) {
  hipLaunchKernelGGL((w4a16_gemm_ks_kernel<T, CW, NG, NW>), dim3((unsigned)n_blocks), dim3(NW * 64),
                     w4_ks_lds_bytes(NW), st, kp);
}

template <typename T, int CW, int NG>
static void launch_ks_nw(const GemmKParams& kp, int nw, int n_blocks, hipStream_t st) {
  if (nw == 4) {
    // (CW = 4, NG = 2, NW = 4) spills at the 256-VGPR cap: refused by gemm_ks_config_ok, not built
    if constexpr (!(CW == 4 && NG == 2)) launch_ks_t<T, CW, NG, 4>(kp, n_blocks, st);
  } else {
    launch_ks_t<T, CW, NG, 8>(kp, n_blocks, st);
  }
}

template <typename T, int CW>
static void launch_ks_ng(const GemmKParams& kp, int ng, int nw, int n_blocks, hipStream_t st) {
  if (ng == 4) {
    if constexpr (CW <= 2) launch_ks_nw<T, CW, 4>(kp, nw, n_blocks, st);
  } else if (ng == 2) launch_ks_nw<T, CW, 2>(kp, nw, n_blocks, st);
  else launch_ks_nw<T, CW, 1>(kp, nw, n_blocks, st);
}

template <typename T>
static void launch_ks_cw(const GemmKParams& kp, int ng, int cw, int nw, int n_blocks, hipStream_t st) {
  if (cw == 4) launch_ks_ng<T, 4>(kp, ng, nw, n_blocks, st);
  else if (cw == 2) launch_ks_ng<T, 2>(kp, ng, nw, n_blocks, st);
  else launch_ks_ng<T, 1>(kp, ng, nw, n_blocks, st);
}

bool gemm_ks_config_ok(int ng, int cw, int nw, int mt) {
  // two row tiles: group >= 128 only -- the NG = 2 / 4 instantiations spill (1 / 15 VGPRs at the 256 cap)
  // and are neither planned nor built
  if (mt == 2) return cw == 1 && nw == 8 && ng == 1;
  if (mt != 1) return false;
  if (cw != 1 && cw != 2 && cw != 4) return false;
  if (nw != 4 && nw != 8) return false;
  if (ng == 4 && cw == 4) return false;  // 16 segments per wave: register budget
  if (ng == 2 && cw == 4 && nw == 4) return false;  // spills (4 accumulator rows per wave to reduce)
  return ng == 1 || ng == 2 || ng == 4;
}

void launch_gemm_ks(const GemmKParams& kp, int dtype, int ng, int cw, int nw, int n_blocks,
                    hipStream_t st, int mt) {
  if (mt == 2) {  // (ng == 1: gemm_ks_config_ok)
    if (dtype == SLM_BF16) launch_ks_mt2<bf16_tag, 1>(kp, n_blocks, st);
    else launch_ks_mt2<f16_tag, 1>(kp, n_blocks, st);
    return;
  }
  if ((kp.ks_dbg & 8) && !(kp.ks_dbg & 4) && dtype == SLM_BF16 && ng == 1 && nw == 8 && cw == 4) {  // packed-fma probe
    hipLaunchKernelGGL((w4a16_gemm_ks_kernel<bf16_tag, 4, 1, 8, false, true>), dim3((unsigned)n_blocks), dim3(512),
                       w4_ks_lds_bytes(8), st, kp);
    return;
  }
  if ((kp.ks_dbg & 4) && dtype == SLM_BF16 && ng == 1 && nw == 8 && (cw == 4 || cw == 2)) {  // timeline probe
    if (cw == 4)
      hipLaunchKernelGGL((w4a16_gemm_ks_kernel<bf16_tag, 4, 1, 8, true>), dim3((unsigned)n_blocks), dim3(512),
                         w4_ks_lds_bytes(8), st, kp);
    else
      hipLaunchKernelGGL((w4a16_gemm_ks_kernel<bf16_tag, 2, 1, 8, true>), dim3((unsigned)n_blocks), dim3(512),
                         w4_ks_lds_bytes(8), st, kp);
    return;
  }
  if (dtype == SLM_BF16) launch_ks_cw<bf16_tag>(kp, ng, cw, nw, n_blocks, st);
  else launch_ks_cw<f16_tag>(kp, ng, cw, nw, n_blocks, st);
}

}  // namespace slm
