// attn_tile.hip -- MFMA tile kernel for paged-KV attention with MANY query rows per KV head:
// prefill, chunked prefill and speculative verify (q_len = k+1), where attention is a real dense
// contraction (unlike q_len = 1 decode, which stays on the VALU stream kernel in attn.hip).
// Same operator, same semantics (reference src/kernels/attention/attn_api.cpp:14-73; masks
// common/mask.h:49-89; online softmax common/online_softmax.cuh:39-162).
//
// Work item: one workgroup = (sequence, kv head, tile of 32*NW query rows), a "query row" being a
// (token, q-head-in-group) pair with the head fastest -- the GQA group shares every K/V byte
// (the reference packs GQA into M the same way: sm80_kernel_mha.cuh:93-137).  Each wave owns 32
// query rows and walks the KV range in tiles of 32 rows:
//
//   S^T[32 kv x 32 q] = K . Q^T     8 (D=128) x v_mfma_f32_32x32x16: A = K rows from LDS (XOR-swizzled,
//                                   conflict-free ds_read_b128), B = Q rows, loaded once from global
//                                   straight into fragment registers (8 consecutive dims per lane).
//   The swapped product leaves every lane with 16 scores of ONE query row (C layout: col = q row):
//   row max / sum are 16 register ops + one lane-half exchange -- no LDS, no 32-lane butterflies.
//   P^T stays in registers: the C-fragment row order {4h+e, 8+4h+e} per k-step is used as the
//   contraction order of P.V as well, so cvt_pk of the softmax outputs IS the MFMA B operand.
//   O^T[D x 32 q] += V^T . P^T      (D/32 x 2) MFMAs; V stays ROW-major in LDS (16-B stores, as it
//                                   arrives from HBM) and the A = V^T fragments come out of
//                                   gfx950's transpose read: two ds_read_b64_tr_b16 per fragment,
//                                   each handing a lane 4 consecutive kv rows of ITS d column.
//   O accumulators have col = q row, so the online-softmax rescale is one scalar per lane.
//
// K/V rows are gathered through the block table (slot = table[i >> log2 bs] + (i & (bs-1)),
// bit-exact with sm80_kernel_mha.cuh:146-152) by all waves of the workgroup, 16 B per lane.
#include <type_traits>

#include "attn_common.h"
#include "tuning.h"

namespace slm {

template <typename T>
struct TileMfma;
template <>
struct TileMfma<bf16_tag> {
  typedef bf16x8_t frag;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <>
struct TileMfma<f16_tag> {
  typedef f16x8_t frag;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

// (s_setprio 1 around the two MFMA clusters was measured on this form: chunked 8 x 256 over 4 k +1 %,
// causal 1 x 2048 -13 % -- the long query tile's co-resident partner starves it; not used)
// KV rows per tile: 32 (one S^T block per wave and tile) or 64 (two: KVT template parameter)
// V tile in LDS: HD/16 sub-tiles of [32 kv][16 d] bf16 (32 B per kv row, 1 KiB per sub-tile), the
// image ds_read_b64_tr_b16 gathers from: a 16-lane group reads a [4 kv][16 d] block (lane t points
// at kv row t/4, 8-byte chunk t%4) and lane t receives the 4 kv values of column t.  Sub-tile s
// sits at s * 1280 + 32 * pi(s), pi = (s >> 1) + 4 * (s & 1): the two sub-tiles a 32-lane read
// phase touches (2m, 2m+1) land 32 banks apart and the 16 slots of a kv row a store phase writes
// cover all 64 banks once.
constexpr int V_SUB_STRIDE = 1280;
__device__ __forceinline__ int v_sub_base(int s) { return s * V_SUB_STRIDE + 32 * ((s >> 1) + 4 * (s & 1)); }
typedef short tr_v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) tr_v4s tr_lds_v4s;

// NW: waves per workgroup (compile-time so the staging registers are statically indexed).
// PF: prefetch the next K/V tile into registers while the current one is consumed.
// waves_per_eu(2): left alone the one-wave variant takes 312 VGPRs (one wave per SIMD, and then no
// amount of extra workgroups hides the block-table -> KV load latency chain); capped at 256 it
// fits two without spilling.
// PLAIN: no soft-cap, no alibi, no sliding window (the Llama case): KV tiles that lie entirely below
// the causal diagonal of every row of the wave then skip the whole mask / bias arithmetic
// (scale + max only) -- left generic, the compiler if-converts the feature tests into straight-line
// code (16 tanh + 16 int->float + 5 compare/select per score, ~2/3 of the loop's VALU work).
// KVT / DB (round 4, the prefill classes): KV tiles of 64 rows -- two S^T blocks per wave and tile, so
// half the barriers, table lookups and loop overhead per KV row -- in a DOUBLE-BUFFERED LDS image: the
// next tile's registers are stored into the other buffer right after this tile's MFMAs, ONE barrier per
// tile (the single-buffer form needs "everybody done reading" + "everybody done writing").
// DMA (round 5, the double-buffered classes): the next tile travels HBM -> LDS by LDS-DMA
// (buffer_load_dwordx4 ... lds) instead of through 32 staging VGPRs + ds_write.  Measured on the
// register-staged form (tools/probes/experiments/attn_tile_pipe_halfsteps_and_lds_ahead.md): with the
// staging taken out of the loop the kernel runs 19...29 % faster -- global-load issue, per-lane 64-bit
// address arithmetic, eight 16-B LDS stores per thread and their waits are what the tile costs beside its
// MFMAs.  Here:
//   * the cache is addressed as a STRUCTURED buffer: index = cache slot (the block-table gather, bit-exact
//     with sm80_kernel_mha.cuh:146-152), stride = the slot stride in the descriptor, per-lane offset = the
//     16-B column -- the address multiply-add happens in the buffer unit, no 64-bit VALU;
//   * a wave instruction writes 1 KiB of LDS linearly (lane l -> base + 16 l), so the LDS image is made
//     on the SOURCE side: for K, lane l of instruction i fetches row 4i + l/16 (head_dim 128), column
//     (l % 16) ^ (row % 16) -- the XOR swizzle of the image; for V, instruction (half h, sub-tile s)
//     fetches row 32h + l/2, column 2s + l%2 -- exactly one [32 kv][16 d] sub-tile of the transpose-read
//     image;
//   * the cache slots of a tile's 64 rows are looked up ONCE per workgroup (64 threads, one row each, two
//     tiles ahead) and handed round through LDS, instead of by every thread for its own rows.
// Needs slot strides < 16 KiB (the descriptor's 14-bit stride field); the launcher falls back otherwise.
// PIPE (round 5, on the DMA form): the wave is software-pipelined ACROSS tiles -- two score blocks live:
//     A(t): S' = K[t+1] . Q^T  (16 MFMAs at head_dim 128)   ||   P = exp2(S * c - m), l += sum P   (tile t)
//     B(t): O += V[t]^T . P    (16 MFMAs)                    ||   row max of S', lazy-rescale decision for t + 1
// Unpipelined, a wave's tile is QK^T -> softmax -> P.V strictly in turn (each needs the previous result) and the
// matrix pipe idles through the wave's whole softmax unless the co-resident wave happens to be in a matrix phase;
// here the exponentials sit in the shadow of the NEXT tile's QK^T inside the same wave.  K runs one tile ahead
// of V; with the copy issued at the START of an iteration (LDS-DMA, after the barrier) two K and two V buffers
// still suffice: iteration t reads K[t+1] (buffer (t+1)&1) and V[t] (t&1) and fills K[t+2] -> t&1 (last read
// by QK^T(t) in iteration t-1) and V[t+1] -> (t+1)&1 (last read by P.V(t-1)).  One barrier per 64 rows as before;
// 32 more VGPRs (the second score block), which the DMA form freed.  Cross-half exchanges by v_permlane32_swap.
// KV2 (round 5, on the PIPE form; plain causal prefill): TWO GROUPS of NW waves share a workgroup's 32 NW query
// rows and split its KV range -- group g takes the 64-row tiles g, g + 2, g + 4, ... with its own K / V buffers,
// slot ring and online-softmax state; the two (m, l, O) meet once, through LDS, at the end.  A causal prefill of
// one long sequence is the serial chain of its longest query tile (32 KV tiles at 2 k: ~55 of the kernel's 59 us
// whatever the rest of the chip does); splitting the KV range ACROSS workgroups answers that with an fp32 partial
// per (row, head) through HBM and a combine pass (measured: 60 -> 97 us); inside the workgroup the partial is 16 KiB
// of LDS per wave and the chain halves for every query tile.  Same waves per SIMD as two 4-wave workgroups.
template <typename T, int HD, int NW, bool PF, bool PLAIN, int KVT = 32, bool DB = false, bool DMA = false, bool PIPE = false,
          bool KV2 = false>
__global__ void __launch_bounds__(64 * NW * (KV2 ? 2 : 1)) __attribute__((amdgpu_waves_per_eu(2))) attn_tile_kernel(const AttnKParams p, int tiles_per_seq) {
  typedef typename TileMfma<T>::frag frag_t;
  static_assert(KVT == 32 || KVT == 64, "KV tile rows");
  static_assert(!DB || PF, "the double-buffered form prefetches through registers");
  static_assert(!DMA || DB, "LDS-DMA staging is built on the double-buffered form");
  static_assert(!PIPE || (DMA && KVT == 64), "the pipelined form is built on the LDS-DMA form");
  static_assert(!KV2 || PIPE, "the two-group form is built on the pipelined form");
  constexpr int GW = KV2 ? 2 : 1;     // wave groups sharing the query rows (each its own KV tiles and LDS buffers)
  constexpr int TILE_KV = KVT;
  constexpr int NH = KVT / 32;        // 32-row S^T blocks per tile
  constexpr int KSTEPS = HD / 16;     // MFMA k-steps of the QK product
  constexpr int DT = HD / 32;         // 32-row d-tiles of the output
  constexpr int NSLOT = HD / 8;       // 16-B slots per K row
  constexpr int K_BYTES = TILE_KV * HD * 2;
  constexpr int VH_BYTES = (HD / 16) * V_SUB_STRIDE;  // V image of one 32-row half tile
  constexpr int V_BYTES = NH * VH_BYTES;
  constexpr int NBUF = DB ? 2 : 1;
  __shared__ __attribute__((aligned(16))) char k_lds[GW * NBUF * K_BYTES];
  __shared__ __attribute__((aligned(16))) char v_lds[GW * NBUF * V_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = KV2 ? wave_all / NW : 0;            // wave group (KV2)
  const int wave = KV2 ? wave_all % NW : wave_all;    // wave inside its group = 32-row block of the query tile
  constexpr int nthreads = 64 * NW;                   // (threads of ONE group: the row / staging arithmetic below)
  constexpr int ITEMS = TILE_KV * (HD / 8) / nthreads;  // 16-B staging items per thread per tile
  const int hh = lane >> 5;
  const int l31 = lane & 31;

  int bid = blockIdx.x;
  const int split = bid % p.n_splits;  // split-KV: this workgroup's share of the KV range
  bid /= p.n_splits;
  // Causal prefill is triangular (query tile t reads (t + 1) / T of the history) and the whole grid
  // is usually resident at once, so nothing rebalances it at run time.  Order the items so that
  // the first half of the grid (first resident workgroup of every CU) takes the LONG query tiles
  // in descending order and the second half (its co-resident partner) the SHORT ones ascending:
  // every CU then holds about the same total work.
  {
    const int hb = p.n_kv_heads * p.batch;
    const int n_items = tiles_per_seq * hb;
    const int half = (n_items + 1) / 2;
    // (KV2: one workgroup per CU at a time, handed out in order: longest first)
    const int r = KV2 ? bid : (bid < half ? bid : n_items - 1 - (bid - half));  // rank by descending query tile
    bid = (tiles_per_seq - 1 - r / hb) + tiles_per_seq * (r % hb);
  }
  const int tile = bid % tiles_per_seq;
  bid /= tiles_per_seq;
  const int kvh = bid % p.n_kv_heads;
  const int b = bid / p.n_kv_heads;

  const int q_start = p.q_cu[b];
  const int q_len = p.q_cu[b + 1] - q_start;
  const int kv_len = p.kv_cu[b + 1] - p.kv_cu[b];
  const int G = p.group;
  const int rows_total = q_len * G;
  const int rows_per_wg = 32 * (nthreads >> 6);
  const int row0 = tile * rows_per_wg;
  if (row0 >= rows_total) return;  // workgroup-uniform
  if (rows_total < p.rows_lo || rows_total >= p.rows_hi) return;  // another launch owns this sequence

  // this lane's query row
  const int jrow = row0 + wave * 32 + l31;
  const bool jvalid = jrow < rows_total;
  const int tq = jvalid ? jrow / G : (rows_total - 1) / G;  // token index inside the sequence
  const int head = kvh * G + (jvalid ? jrow % G : 0);
  const int diag = kv_len - q_len + tq;  // last visible kv index of this row (causal)
  if (kv_len <= 0) {
    if (grp != 0) return;   // (workgroup-uniform condition: nobody reaches a barrier)
    // empty history (workgroup-uniform): this kernel owns the rows, so it publishes the empty
    // result -- a zero output row, or a zero-weight partial (l = 0, O = 0) for the combine kernel,
    // whose loads are unconditional (an unwritten partial would be combined as garbage)
    if (!jvalid) return;
    if (p.n_splits > 1) {
      const int64_t pi = ((int64_t)(q_start + tq) * p.n_heads + head) * p.part_slots + split;
      float* opp = p.o_part + pi * HD;
      for (int d = 4 * hh; d < HD; d += 8) *reinterpret_cast<f32x4*>(opp + d) = f32x4{0.f, 0.f, 0.f, 0.f};
      if (hh == 0) {
        p.ml_part[pi * 2 + 0] = ATTN_M_INIT;
        p.ml_part[pi * 2 + 1] = 0.f;
      }
    } else {
      char* op = reinterpret_cast<char*>(p.out) + 2 * ((int64_t)(q_start + tq) * p.o_ts + (int64_t)head * p.o_hs);
      for (int d = 4 * hh; d < HD; d += 8) *reinterpret_cast<u32x2*>(op + 2 * d) = u32x2{0u, 0u};
    }
    return;
  }

  // Q fragments (B operand of S^T = K . Q^T): 8 consecutive dims per lane per k-step
  frag_t qf[KSTEPS];
  {
    const char* qp = reinterpret_cast<const char*>(p.q) +
                     2 * ((int64_t)(q_start + tq) * p.q_ts + (int64_t)head * p.q_hs + 8 * hh);
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (jvalid) v = *reinterpret_cast<const u32x4*>(qp + 32 * s);
      qf[s] = __builtin_bit_cast(frag_t, v);
    }
  }
  const float slope2 = p.alibi ? p.alibi[head] * LOG2E : 0.f;

  // KV range of the workgroup: causal upper bound of its last row, window lower bound of its first
  const int last_row = min(row0 + rows_per_wg, rows_total) - 1;
  const int wg_hi = min(kv_len, kv_len - q_len + last_row / G + 1);
  int wg_lo = 0;
  if (p.window >= 0) wg_lo = max(0, kv_len - q_len + row0 / G - p.window);
  wg_lo = (wg_lo / TILE_KV) * TILE_KV;
  int wg_hi_s = wg_hi;
  if (p.n_splits > 1) {
    // equal shares in whole KV tiles; trailing splits may be empty (they publish l = 0)
    const int n_t = (wg_hi - wg_lo + TILE_KV - 1) / TILE_KV;
    const int per = (n_t > 0 ? (n_t + p.n_splits - 1) / p.n_splits : 0) * TILE_KV;
    wg_lo = min(wg_lo + split * per, max(wg_hi, wg_lo));
    wg_hi_s = min(wg_hi, wg_lo + per);
  }

  f32x16 oacc[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
  float m_run = ATTN_M_INIT, l_run = 0.f;

  const int bcu0 = p.bcu[b];
  const char* kbase = reinterpret_cast<const char*>(p.kc) + 2 * (int64_t)kvh * p.k_hs;
  const char* vbase = reinterpret_cast<const char*>(p.vc) + 2 * (int64_t)kvh * p.v_hs;
  const uint32_t k_sb = (uint32_t)(2 * p.k_ss), v_sb = (uint32_t)(2 * p.v_ss);

  u32x4 kreg[DMA ? 1 : ITEMS], vreg[DMA ? 1 : ITEMS];
  // AHEAD: the block-table lookups (a global load the K/V addresses depend on) run one tile ahead
  // of the K/V loads that use them: prefill 423 -> 447 TFLOP/s, chunked 678 -> 731.  The one-wave
  // (verify) form is at its 256-VGPR cap and spills with the 8 extra registers (335 -> 401 us), so
  // it looks the slots up right before its loads.
  constexpr bool AHEAD = NW > 1;
  // One-wave form: the block-table slice of the workgroup's KV range is staged in LDS once (one
  // coalesced read, as the decode kernel does), so the per-tile lookup is an LDS read (~100
  // cycles) instead of an L2 round trip in front of every tile's K/V loads -- with one tile in
  // flight per wave that round trip was ~20 % of the per-tile cycle.  Ranges longer than the
  // staged window (TILE_TBL blocks) fall back to the global lookup.
  constexpr int TILE_TBL = NW == 1 ? 2048 : 1;
  __shared__ int tbl_lds[TILE_TBL];
  bool tbl_staged = false;
  const int blk_lo = wg_lo >> p.block_shift;
  if constexpr (NW == 1) {
    const int n_ent = wg_lo < wg_hi_s ? ((min(wg_hi_s, kv_len) - 1) >> p.block_shift) - blk_lo + 1 : 0;
    tbl_staged = n_ent <= TILE_TBL;
    if (tbl_staged) {
      for (int i = tid; i < n_ent; i += nthreads) tbl_lds[i] = p.bt[bcu0 + blk_lo + i];
      __syncthreads();
    }
  }
  int sreg[DMA ? 1 : ITEMS];  // cache slots of the next tile_load
  auto slot_load = [&](int kt0) {
#pragma unroll
    for (int i = 0; i < (DMA ? 0 : ITEMS); ++i) {
      const int r = (tid + nthreads * i) / NSLOT;
      // clamp: masked below, must stay in bounds (and inside the staged window)
      const int row = min(kt0 + r, min(wg_hi_s, kv_len) - 1);
      const int blk = row >> p.block_shift;
      const int first = (NW == 1 && tbl_staged) ? tbl_lds[blk - blk_lo] : p.bt[bcu0 + blk];
      sreg[i] = first + (row & p.block_mask);
    }
  };
  // K/V rows of the tile whose slots slot_load() fetched; then the slots of tile `kt_next`, so the
  // table lookup (a dependent global load) is off the critical path of the NEXT tile's K/V loads
  auto tile_load = [&](int kt_next) {
#pragma unroll
    for (int i = 0; i < (DMA ? 0 : ITEMS); ++i) {
      const int sl = (tid + nthreads * i) % NSLOT;
      const int slot = sreg[i];
      const u32x4* kp_ = reinterpret_cast<const u32x4*>(kbase + (uint64_t)(uint32_t)slot * k_sb + 16 * sl);
      const u32x4* vp_ = reinterpret_cast<const u32x4*>(vbase + (uint64_t)(uint32_t)slot * v_sb + 16 * sl);
      if constexpr (NW == 1) {
        // verify class (<= 32 query rows per KV head): every KV byte is read by exactly one
        // workgroup -- stream it past the caches like the decode kernel does (measured on
        // specverify_120x5_kv4096: 360 -> 337 us, 5.6 -> 6.0 TB/s)
        kreg[i] = __builtin_nontemporal_load(kp_);
        vreg[i] = __builtin_nontemporal_load(vp_);
      } else {
        kreg[i] = *kp_;
        vreg[i] = *vp_;
      }
    }
    if constexpr (AHEAD) slot_load(kt_next);
  };
  auto tile_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < (DMA ? 0 : ITEMS); ++i) {
      const int idx = tid + nthreads * i;
      const int r = idx / NSLOT, sl = idx % NSLOT;
      *reinterpret_cast<u32x4*>(k_lds + buf * K_BYTES + r * (HD * 2) + ((sl ^ (r & (NSLOT - 1))) << 4)) = kreg[i];
      // V: row-major sub-tiles (slot sl = d / 8 -> sub-tile sl / 2, half sl % 2), one image per 32 rows
      *reinterpret_cast<u32x4*>(v_lds + buf * V_BYTES + (r >> 5) * VH_BYTES + v_sub_base(sl >> 1) + (r & 31) * 32 +
                                ((sl & 1) << 4)) = vreg[i];
    }
  };
  // transpose-read address of this lane inside a sub-tile pair: lanes 16..31 / 48..63 read the odd
  // sub-tile (d columns 16..31 of the 32-row d tile), lane halves read kv rows +4 (hh)
  const uint32_t v_lane = (uint32_t)(uintptr_t)v_lds +
                          (uint32_t)(((lane & 15) >> 2) * 32 + (lane & 3) * 8 + hh * 128 +
                                     ((lane >> 4) & 1) * (v_sub_base(1) - v_sub_base(0)));

  // ---- LDS-DMA staging (DMA) ----
  typedef __attribute__((address_space(3))) void lds_void;
  constexpr int ROWS_KI = 64 / NSLOT;                 // K rows one wave instruction (64 lanes x 16 B) covers
  constexpr int KI = DMA ? (TILE_KV / ROWS_KI) / NW : 1;   // K instructions per wave and tile
  constexpr int NSUB = HD / 16;                       // V sub-tiles per 32-row half
  constexpr int VI = DMA ? (NH * NSUB) / NW : 1;      // V instructions per wave and tile
  static_assert(!DMA || ((TILE_KV / ROWS_KI) % NW == 0 && (NH * NSUB) % NW == 0), "whole instructions per wave");
  __shared__ int slot_lds[PIPE ? GW * 3 * TILE_KV : DMA ? 2 * TILE_KV : 1];   // cache slots of the rows of two (PIPE: three) tiles
  // (the DMA is issued as inline asm, invisible to the compiler's waitcnt bookkeeping: through the builtin
  //  hipcc makes every ds_read of k_lds / v_lds wait for the wave's own outstanding DMA into the OTHER
  //  buffer -- it cannot tell the halves of one LDS array apart -- which serialises copy and compute; the
  //  one wait the data needs is the explicit vmcnt(0) in front of the tile's barrier)
  u32x4 k_rs = {0u, 0u, 0u, 0u}, v_rs = {0u, 0u, 0u, 0u};
  // per-lane parts of the DMA addressing (two VGPRs for K, two for V); everything that depends on the wave and
  // on the instruction index is wave-uniform and stays scalar:
  //   K instruction j of the wave covers rows R0 + ROWS_KI j + kl_row (R0 = ROWS_KI KI wave), 16-B slot
  //   (lane % NSLOT) ^ (row % NSLOT); R0 + ROWS_KI j is a multiple of ROWS_KI > kl_row, so the XOR splits into a lane
  //   part and a scalar part;  V instruction ii covers rows 32 (ii / NSUB) + lane / 2, columns 2 (ii % NSUB) + lane % 2
  const int kl_row = lane / NSLOT;
  const int kl_voff = 16 * ((lane % NSLOT) ^ kl_row);
  const int vl_row = lane >> 1;
  const int vl_voff = 16 * (lane & 1);
  const uint32_t k_lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)k_lds;
  const uint32_t v_lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)v_lds;
  if constexpr (DMA) {
    // structured buffer descriptors: base, stride (14 bits), records, raw 32-bit data format
    const uint64_t kb64 = (uint64_t)(uintptr_t)kbase, vb64 = (uint64_t)(uintptr_t)vbase;
    k_rs = u32x4{(uint32_t)kb64, (uint32_t)((kb64 >> 32) & 0xffffu) | (k_sb << 16), 0x7fffffffu, 0x00020000u};
    v_rs = u32x4{(uint32_t)vb64, (uint32_t)((vb64 >> 32) & 0xffffu) | (v_sb << 16), 0x7fffffffu, 0x00020000u};
  }
  auto dma16 = [&](const u32x4& rs, uint32_t dst, int vindex, int voff) __attribute__((always_inline)) {
    const u32x2 iv = {(uint32_t)vindex, (uint32_t)voff};
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 idxen offen lds"
                 :
                 : "v"(iv), "s"(rs), "s"(dst)
                 : "memory", "m0");
  };
  // K / V rows of the tile whose slots sit in slot_lds[sb] -> LDS buffer `buf`
  auto dma_k = [&](int buf, int sb) __attribute__((always_inline)) {
    if constexpr (DMA) {
      int ks[KI];
#pragma unroll
      for (int j = 0; j < KI; ++j) ks[j] = slot_lds[sb * TILE_KV + ROWS_KI * (wave * KI + j) + kl_row];
#pragma unroll
      for (int j = 0; j < KI; ++j) {
        const int r0 = ROWS_KI * (wave * KI + j);   // (wave-uniform)
        dma16(k_rs, k_lds0 + buf * K_BYTES + (wave * KI + j) * 1024, ks[j], kl_voff ^ (16 * (r0 & (NSLOT - 1))));
      }
    }
  };
  auto dma_v = [&](int buf, int sb) __attribute__((always_inline)) {
    if constexpr (DMA) {
      int vs[VI];
#pragma unroll
      for (int j = 0; j < VI; ++j) vs[j] = slot_lds[sb * TILE_KV + 32 * ((wave * VI + j) / NSUB) + vl_row];
#pragma unroll
      for (int j = 0; j < VI; ++j) {
        const int ii = wave * VI + j;               // (wave-uniform)
        dma16(v_rs, v_lds0 + buf * V_BYTES + (ii / NSUB) * VH_BYTES + v_sub_base(ii % NSUB), vs[j], vl_voff + 32 * (ii % NSUB));
      }
    }
  };
  auto dma_tile = [&](int buf, int sb) __attribute__((always_inline)) {
    dma_k(buf, sb);
    dma_v(buf, sb);
  };
  // one row per thread (threads 0..63): the slot of row kt0 + tid (clamped: masked below, in bounds)
  auto slot_lookup = [&](int kt0) __attribute__((always_inline)) -> int {
    const int row = min(kt0 + (tid & (TILE_KV - 1)), min(wg_hi_s, kv_len) - 1);
    return p.bt[bcu0 + (row >> p.block_shift)] + (row & p.block_mask);
  };

  if constexpr (PIPE) {
    constexpr float LAZY_TH = 6.0f;
    constexpr int SPS = 16 * NH / KSTEPS;   // scores exponentiated per k-step of the QK^T beside them (4 / 8)
    static_assert(SPS == 4 || SPS == 8, "head_dim 128 / 64");
    typedef std::integral_constant<int, 0> P0;
    typedef std::integral_constant<int, 1> P1;
    f32x16 sc[2][NH];             // two score blocks: the tile being exponentiated and the next tile's QK^T
    float cm[2] = {1.0f, 1.0f};   // multiplier of a block's scores inside the exponent's fma (1 once a masked block was rewritten)
    float l_lane = 0.f;           // this lane's share of the row sum (the lane halves meet once, at the end)
    const int dmin = kv_len - q_len + (row0 + wave * 32) / G;   // smallest diagonal of the wave's rows
    auto xhalf = [&](float x, bool take_max) __attribute__((always_inline)) {   // combine with the other lane half
      const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
      const float a = __uint_as_float(r[0]), b = __uint_as_float(r[1]);
      return take_max ? fmaxf(a, b) : a + b;
    };
    auto qk_step = [&](auto pc, int kbuf, int s) __attribute__((always_inline)) {   // k-step s of sc[P] = K[kbuf] . Q^T
      constexpr int P = decltype(pc)::value;
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const int kr = 32 * h + l31;
        const u32x4 kv4 = *reinterpret_cast<const u32x4*>(k_lds + kbuf * K_BYTES + kr * (HD * 2) +
                                                          (((2 * s + hh) ^ (kr & (NSLOT - 1))) << 4));
        sc[P][h] = TileMfma<T>::run(__builtin_bit_cast(frag_t, kv4), qf[s], sc[P][h]);
      }
    };
    auto zero = [&](auto pc) __attribute__((always_inline)) {
      constexpr int P = decltype(pc)::value;
#pragma unroll
      for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[P][h][r] = 0.f;
    };
    // scale / soft-cap / alibi / mask of block P (tile at kt0), its row max, lazy rescale of O and l
    auto decide = [&](auto pc, int kt0) __attribute__((always_inline)) {
      constexpr int P = decltype(pc)::value;
      const bool interior = PLAIN && p.scale_log2 > 0.f && kt0 + TILE_KV <= kv_len && kt0 + TILE_KV - 1 <= dmin;
      float mloc = -INFINITY;
      if (interior) {
#pragma unroll
        for (int r = 0; r < 16 * NH; r += 2) mloc = fmaxf(fmaxf(mloc, sc[P][r >> 4][r & 15]), sc[P][(r + 1) >> 4][(r + 1) & 15]);
        mloc *= p.scale_log2;
        cm[P] = p.scale_log2;
      } else {
#pragma unroll
        for (int r = 0; r < 16 * NH; ++r) {
          const int kv_idx = kt0 + 32 * (r >> 4) + (r & 3) + 8 * ((r & 15) >> 2) + 4 * hh;
          float a = sc[P][r >> 4][r & 15];
          if constexpr (!PLAIN) {
            if (p.softcap > 0.f) a = fast_tanh(a * p.pre_scale);
            a = a * p.scale_log2 + slope2 * (float)kv_idx;
          } else {
            a = a * p.scale_log2;
          }
          bool vis = jvalid && kv_idx <= diag && kv_idx < kv_len;
          if constexpr (!PLAIN) {
            if (p.window >= 0) vis = vis && (diag - kv_idx) <= p.window;
          }
          a = vis ? a : -INFINITY;
          sc[P][r >> 4][r & 15] = a;
          mloc = fmaxf(mloc, a);
        }
        cm[P] = 1.0f;
      }
      mloc = xhalf(mloc, true);
      // (every P.V issued so far has been issued in front of this point: O is complete up to the previous tile)
      if (__any(mloc > m_run + LAZY_TH)) {
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = fast_exp2(m_run - m_new);
        m_run = m_new;
        l_lane *= alpha;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
      }
    };
    // A: [QK^T of the next tile -> sc[1 - P]]  ||  exponentials of sc[P], packed as the B fragments of P.V
    auto step_a = [&](auto pc, auto with_qk, int kbuf_next, u32x4 (&pb)[2 * NH]) {
      constexpr int P = decltype(pc)::value;
      constexpr bool QK = decltype(with_qk)::value;
      if constexpr (QK) zero(std::integral_constant<int, 1 - P>{});
      const float c = cm[P];
      float sv[8];
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) {
        if constexpr (QK) qk_step(std::integral_constant<int, 1 - P>{}, kbuf_next, s);
#pragma unroll
        for (int e = 0; e < SPS; ++e) {
          const int r = SPS * s + e;
          const float v = fast_exp2(fmaf(sc[P][r >> 4][r & 15], c, -m_run));
          sv[r & 7] = v;
          l_lane += v;
        }
        if (((s + 1) * SPS) % 8 == 0) {
          const int g = (s + 1) * SPS / 8 - 1;
          pb[g].x = pack2<T>(sv[0], sv[1]);
          pb[g].y = pack2<T>(sv[2], sv[3]);
          pb[g].z = pack2<T>(sv[4], sv[5]);
          pb[g].w = pack2<T>(sv[6], sv[7]);
        }
      }
    };
    // B: O^T += V[vbuf]^T . P^T
    auto step_b = [&](int vbuf, const u32x4 (&pb)[2 * NH]) {
#pragma unroll
      for (int s2 = 0; s2 < 2 * NH; ++s2) {
        const frag_t pfrag = __builtin_bit_cast(frag_t, pb[s2]);
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          const uintptr_t va0 = v_lane + (uint32_t)(v_sub_base(2 * d) + (16 * (s2 & 1)) * 32 + (s2 >> 1) * VH_BYTES + vbuf * V_BYTES);
          const tr_v4s t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_lds_v4s*)va0);
          const tr_v4s t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_lds_v4s*)(va0 + 8 * 32));
          const u32x2 v0 = __builtin_bit_cast(u32x2, t0), v1 = __builtin_bit_cast(u32x2, t1);
          const u32x4 va = {v0.x, v0.y, v1.x, v1.y};
          oacc[d] = TileMfma<T>::run(__builtin_bit_cast(frag_t, va), pfrag, oacc[d]);
        }
      }
    };
    if (wg_lo < wg_hi_s) {
      const int nt = (wg_hi_s - wg_lo + TILE_KV - 1) / TILE_KV;
      // this group's tiles: grp, grp + GW, ...  (i-th one at kt_of(i)); every group runs nt_loop iterations of
      // the loop below (the barriers are the workgroup's), the last of which may be empty for group 1
      const int nt_g = (nt - grp + GW - 1) / GW;
      const int nt_loop = (nt + GW - 1) / GW;
      const int gb = 2 * grp, sbase = 3 * grp;     // first LDS buffer / slot-ring entry of the group
      auto kt_of = [&](int i) __attribute__((always_inline)) { return wg_lo + (i * GW + grp) * TILE_KV; };
      if (wave == 0) {
        slot_lds[(sbase + 0) * TILE_KV + lane] = slot_lookup(kt_of(0));
        slot_lds[(sbase + 1) * TILE_KV + lane] = slot_lookup(kt_of(1));
        slot_lds[(sbase + 2) * TILE_KV + lane] = slot_lookup(kt_of(2));
      }
      __syncthreads();
      dma_tile(gb + 0, sbase + 0);
      dma_k(gb + 1, sbase + 1);
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) asm volatile("" ::"v"(qf[s]));   // (Q is waited for here, not in the loop: see DMA)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (nt_g > 0) {
        zero(P0{});
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) qk_step(P0{}, gb + 0, s);
        decide(P0{}, kt_of(0));
      }
      __syncthreads();   // K buffer 0 is refilled at the top of the first iteration: every wave is done with it
      int m3 = 0;        // i mod 3
      // the group's i-th tile: its scores in sc[P]
      auto body = [&](auto pc, int i) __attribute__((always_inline)) {
        constexpr int P = decltype(pc)::value;
        const int kt0 = kt_of(i);
        const bool more = i + 1 < nt_g;
        // wave 0 of the group looks up the slots of its tile i + 3 (one row per lane), in FRONT of the DMA: see DMA
        int slot_next = 0;
        if (wave == 0) slot_next = slot_lookup(kt_of(i + 3));
        const int m3p1 = m3 == 2 ? 0 : m3 + 1, m3p2 = m3 == 0 ? 2 : m3 - 1;
        dma_k(gb + (i & 1), sbase + m3p2);          // K of tile i + 2
        dma_v(gb + ((i + 1) & 1), sbase + m3p1);    // V of tile i + 1
        u32x4 pb[2 * NH];
        if (i < nt_g) {
          if (more) step_a(pc, std::true_type{}, gb + ((i + 1) & 1), pb);
          else step_a(pc, std::false_type{}, 0, pb);
        }
        if constexpr (KV2) __syncthreads();   // (between the segments: the other group is half an iteration away)
        if (i < nt_g) {
          step_b(gb + (i & 1), pb);
          if (more) decide(std::integral_constant<int, 1 - P>{}, kt_of(i + 1));
        }
        if (wave == 0) slot_lds[(sbase + m3) * TILE_KV + lane] = slot_next;   // tile i + 3 takes tile i's place in the ring
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        m3 = m3p1;
      };
      // KV2, PING-PONG: the two groups run half an iteration apart -- group 1 passes one extra barrier in front of
      // the loop (group 0 one behind it), and the body has a second barrier between its A and B segments: while
      // one group is in segment A of its tile (QK^T of the next tile || this tile's exponentials: matrix + VALU)
      // the other is in segment B of ITS tile (P.V + the next row max: matrix, little VALU).  In step -- both in A,
      // then both in B -- the two waves of a SIMD want the VALU at the same time and the matrix pipe at the same
      // time.  Same code for both groups inside the loop; a group's DMA is issued at the top of its A segment and
      // waited for at the bottom of its B segment, two barriers in front of the A segment that reads it.
      if constexpr (KV2) {
        if (grp == 1) __syncthreads();
      }
      for (int i = 0; i < nt_loop; i += 2) {
        body(P0{}, i);
        if (i + 1 < nt_loop) body(P1{}, i + 1);
      }
      if constexpr (KV2) {
        if (grp == 0) __syncthreads();
      }
    }
    l_run = xhalf(l_lane, false);
    if constexpr (KV2) {
      // the two groups' (m, l, O) of a query row meet in LDS (the V buffers: every wave is behind the loop's last
      // barrier): group 1 publishes, group 0 merges and carries on to the epilogue
      float* mg = reinterpret_cast<float*>(v_lds) + wave * ((DT * 16 + 2) * 64) + lane;
      __syncthreads();
      if (grp == 1) {
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) mg[(d * 16 + r) * 64] = oacc[d][r];
        mg[(DT * 16) * 64] = m_run;
        mg[(DT * 16 + 1) * 64] = l_run;
      }
      __syncthreads();
      if (grp == 1) return;
      const float m1 = mg[(DT * 16) * 64], l1 = mg[(DT * 16 + 1) * 64];
      const float m_new = fmaxf(m_run, m1);
      const float a0 = fast_exp2(m_run - m_new), a1 = fast_exp2(m1 - m_new);
      m_run = m_new;
      l_run = l_run * a0 + l1 * a1;
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = oacc[d][r] * a0 + mg[(d * 16 + r) * 64] * a1;
    }
  } else {
  if constexpr (DMA) {
    if (wg_lo < wg_hi_s) {
      if (tid < TILE_KV) {
        slot_lds[tid] = slot_lookup(wg_lo);
        slot_lds[TILE_KV + tid] = slot_lookup(wg_lo + TILE_KV);
      }
      __syncthreads();
      dma_tile(0, 0);
    }
    // the Q fragments are consumed HERE as far as the compiler is concerned: otherwise its wait for them sits
    // in front of the first MFMA inside the loop, counted against its own loads only -- and drains the DMA
    // issued in front of it on every iteration
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) asm volatile("" ::"v"(qf[s]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  } else if constexpr (PF) {
    if (wg_lo < wg_hi_s) {
      slot_load(wg_lo);
      tile_load(min(wg_lo + TILE_KV, wg_hi_s - 1));
      tile_store(0);
    }
    __syncthreads();
  }
  int cur = 0;  // LDS buffer holding the tile being consumed (DB)
  int tpar = 0; // parity of the tile being consumed (DMA: slot_lds[tpar ^ 1] holds the next tile's slots)
  for (int kt0 = wg_lo; kt0 < wg_hi_s; kt0 += TILE_KV) {
    int slot_next = 0;
    if constexpr (DMA) {
      // next tile HBM -> the other LDS buffer (free since the last barrier), the tile after that: its slots
      // (the lookup -- wave 0, one row per lane, a wave-uniform branch -- is issued in FRONT of the DMA: a
      //  compiler-tracked load behind the untracked DMA would make the first wait on it drain the whole copy;
      //  its one consumer is the store at the bottom of the iteration, where everything is drained anyway)
      if (wave == 0) slot_next = slot_lookup(kt0 + 2 * TILE_KV);
      dma_tile(cur ^ 1, tpar ^ 1);
    } else if constexpr (PF) {
      // next tile's rows travel HBM -> registers while this tile is consumed from LDS
      if constexpr (!AHEAD) slot_load(min(kt0 + TILE_KV, wg_hi_s - 1));
      tile_load(min(kt0 + 2 * TILE_KV, wg_hi_s - 1));
    } else {
      __syncthreads();  // previous tile fully consumed
      slot_load(kt0);
      tile_load(kt0);
      tile_store(0);
      __syncthreads();
    }
    const char* const kb = k_lds + (DB ? cur * K_BYTES : 0);

    // ---- S^T = K . Q^T : NH blocks of 32 kv rows ----
    f32x16 sacc[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[h][r] = 0.f;
    }
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      const int sl = 2 * s + hh;
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const int kr = 32 * h + l31;
        const u32x4 kv4 = *reinterpret_cast<const u32x4*>(kb + kr * (HD * 2) + ((sl ^ (kr & (NSLOT - 1))) << 4));
        sacc[h] = TileMfma<T>::run(__builtin_bit_cast(frag_t, kv4), qf[s], sacc[h]);
      }
    }

    // ---- scale, soft-cap, alibi, mask; online softmax for this lane's query row ----
    // The scores stay in the accumulator registers: masked tiles rewrite them in place (scaled, biased,
    // -inf where invisible) and exponentiate with multiplier 1; interior tiles (no mask, no bias) leave
    // the RAW scores, take the row max on them (pairs -> v_max3_f32) and let the scale ride in the
    // exponent's fma (sm_scale > 0: checked by `interior`).
    float mloc = -INFINITY;
    // wave-uniform: the smallest diagonal of the wave's rows is that of its first row
    const bool interior = PLAIN && p.scale_log2 > 0.f && kt0 + TILE_KV <= kv_len &&
                          kt0 + TILE_KV - 1 <= kv_len - q_len + (row0 + wave * 32) / G;
    float cmul = 1.0f;
    if (interior) {
      float mraw = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16 * NH; r += 2)
        mraw = fmaxf(fmaxf(mraw, sacc[r >> 4][r & 15]), sacc[(r + 1) >> 4][(r + 1) & 15]);
      mloc = mraw * p.scale_log2;
      cmul = p.scale_log2;
    } else {
#pragma unroll
      for (int r = 0; r < 16 * NH; ++r) {
        const int kv_idx = kt0 + 32 * (r >> 4) + (r & 3) + 8 * ((r & 15) >> 2) + 4 * hh;
        float a = sacc[r >> 4][r & 15];
        if constexpr (!PLAIN) {
          if (p.softcap > 0.f) a = fast_tanh(a * p.pre_scale);
          a = a * p.scale_log2 + slope2 * (float)kv_idx;
        } else {
          a = a * p.scale_log2;
        }
        bool vis = jvalid && kv_idx <= diag && kv_idx < kv_len;
        if constexpr (!PLAIN) {
          if (p.window >= 0) vis = vis && (diag - kv_idx) <= p.window;
        }
        a = vis ? a : -INFINITY;
        sacc[r >> 4][r & 15] = a;
        mloc = fmaxf(mloc, a);
      }
    }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    // Lazy reference update: the running max only has to BOUND the exponents, not equal the true
    // max -- O and l are both relative to it and it cancels in O / l.  It is advanced (and O, l
    // rescaled: 4 * HD/32 * 16 multiplies per lane) only when some row of the wave would otherwise
    // exceed 2^LAZY_TH; with the true max every tile the rescale ran on nearly every tile of a
    // 2-4 k history.  P <= 2^6 keeps its full relative precision in fp16 / bf16.
    constexpr float LAZY_TH = 6.0f;
    if (__any(mloc > m_run + LAZY_TH)) {
      const float m_new = fmaxf(m_run, mloc);
      const float alpha = fast_exp2(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
    }
    // P as packed MFMA B fragments (the softmax registers ARE the operand order of P.V).  One-wave (verify)
    // class, round 5: exponentiate and pack per 8-score group -- 8 floats live instead of 16, which takes the
    // soft-cap / alibi / window instantiation from 3 spilled VGPRs to none at its 256-register cap; the
    // multi-wave classes keep all 16 * NH scores in flight and pack inside the MFMA loop (same arithmetic, same
    // summation order either way).
    float lsum = 0.f;
    u32x4 pbq[NW == 1 ? 2 * NH : 1];
    float sv[NW == 1 ? 8 : 16 * NH];
    if constexpr (NW == 1) {
#pragma unroll
      for (int s2 = 0; s2 < 2 * NH; ++s2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int r = 8 * s2 + e;
          sv[e] = fast_exp2(fmaf(sacc[r >> 4][r & 15], cmul, -m_run));
          lsum += sv[e];
        }
        pbq[s2].x = pack2<T>(sv[0], sv[1]);
        pbq[s2].y = pack2<T>(sv[2], sv[3]);
        pbq[s2].z = pack2<T>(sv[4], sv[5]);
        pbq[s2].w = pack2<T>(sv[6], sv[7]);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16 * NH; ++r) {
        sv[r] = fast_exp2(fmaf(sacc[r >> 4][r & 15], cmul, -m_run));
        lsum += sv[r];
      }
    }
    lsum += __shfl_xor(lsum, 32, 64);
    l_run += lsum;

    // ---- O^T += V^T . P^T : P fragments straight from the softmax registers ----
#pragma unroll
    for (int s2 = 0; s2 < 2 * NH; ++s2) {
      u32x4 pb;
      if constexpr (NW == 1) {
        pb = pbq[s2];
      } else {
        pb.x = pack2<T>(sv[8 * s2 + 0], sv[8 * s2 + 1]);
        pb.y = pack2<T>(sv[8 * s2 + 2], sv[8 * s2 + 3]);
        pb.z = pack2<T>(sv[8 * s2 + 4], sv[8 * s2 + 5]);
        pb.w = pack2<T>(sv[8 * s2 + 6], sv[8 * s2 + 7]);
      }
      const frag_t pfrag = __builtin_bit_cast(frag_t, pb);
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        // A[i = d row][k]: kv = 16*s2 + 4*hh + e (e < 4), 16*s2 + 8 + 4*hh + (e - 4): the order the
        // softmax registers hold P in
        // (sub-tile 2d+1 is always v_sub_base(1) - v_sub_base(0) past sub-tile 2d: folded into v_lane)
        const uintptr_t va0 = v_lane + (uint32_t)(v_sub_base(2 * d) + (16 * (s2 & 1)) * 32 + (s2 >> 1) * VH_BYTES +
                                                  (DB ? cur * V_BYTES : 0));
        const tr_v4s t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_lds_v4s*)va0);
        const tr_v4s t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_lds_v4s*)(va0 + 8 * 32));
        const u32x2 v0 = __builtin_bit_cast(u32x2, t0), v1 = __builtin_bit_cast(u32x2, t1);
        const u32x4 va = {v0.x, v0.y, v1.x, v1.y};
        oacc[d] = TileMfma<T>::run(__builtin_bit_cast(frag_t, va), pfrag, oacc[d]);
      }
    }
    if constexpr (DMA) {
      // slot_lds[tpar] held this tile's slots (read when its DMA was issued, one barrier ago)
      if (wave == 0) slot_lds[tpar * TILE_KV + lane] = slot_next;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the next tile has landed
      __syncthreads();
      cur ^= 1;
      tpar ^= 1;
    } else if constexpr (DB) {
      // the other buffer was last read one tile ago and every wave has passed the barrier since:
      // store the prefetched tile there right away; ONE barrier publishes it and retires this tile
      tile_store(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    } else if constexpr (PF) {
      __syncthreads();  // every wave is done reading this tile
      tile_store(0);
      __syncthreads();
    }
  }
  }  // !PIPE

  // ---- epilogue: O^T[d][q] / l -> out[token][head][d]; 4 consecutive d per register quad ----
  if (!jvalid) return;
  if (p.n_splits > 1) {
    // split-KV partials in the token-major kernel's format (combined by attn_combine_kernel):
    // un-normalised O (fp32), running max (log2 domain) and sum of this KV share
    const int64_t pi = ((int64_t)(q_start + tq) * p.n_heads + head) * p.part_slots + split;
    float* opp = p.o_part + pi * HD;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4)
        *reinterpret_cast<f32x4*>(opp + d * 32 + 8 * q4 + 4 * hh) =
            f32x4{oacc[d][4 * q4 + 0], oacc[d][4 * q4 + 1], oacc[d][4 * q4 + 2], oacc[d][4 * q4 + 3]};
    if (hh == 0) {
      p.ml_part[pi * 2 + 0] = m_run;
      p.ml_part[pi * 2 + 1] = l_run;
    }
    return;
  }
  const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
  char* op = reinterpret_cast<char*>(p.out) +
             2 * ((int64_t)(q_start + tq) * p.o_ts + (int64_t)head * p.o_hs);
#pragma unroll
  for (int d = 0; d < DT; ++d) {
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int dd = d * 32 + 8 * q4 + 4 * hh;  // rows (r&3) + 8*(r>>2) + 4*hh, r = 4*q4 .. 4*q4+3
      u32x2 o2;
      o2.x = pack2<T>(oacc[d][4 * q4 + 0] * inv, oacc[d][4 * q4 + 1] * inv);
      o2.y = pack2<T>(oacc[d][4 * q4 + 2] * inv, oacc[d][4 * q4 + 3] * inv);
      *reinterpret_cast<u32x2*>(op + 2 * dd) = o2;
    }
  }
}

bool attn_tile_supported(int head_dim) { return head_dim == 64 || head_dim == 128; }

// rows = largest q_len * group this launch has to cover (query rows per (sequence, kv head))
int launch_attn_tile(const AttnKParams& kp, int dtype, int64_t rows, int64_t max_kv_len, hipStream_t st) {
  if (!attn_tile_supported(kp.head_dim)) return SLM_ERR_UNSUPPORTED;
  if (rows < 1) return SLM_ERR_UNSUPPORTED;
  int nw = (int)((rows + 31) / 32);
  if (nw > 4) nw = 4;
  if (nw == 3) nw = 4;
  const int64_t tiles_per_seq = (rows + 32 * nw - 1) / (32 * nw);
  const int64_t grid = tiles_per_seq * kp.n_kv_heads * kp.batch * kp.n_splits;
  if (grid <= 0 || grid > 0x7fffffffLL) return SLM_ERR_UNSUPPORTED;
  const dim3 g((unsigned)grid), blk(64 * nw);
  const int pf_mode = tune_get(TUNE_ATTN_TILE_PF, 1);
  const bool pf = pf_mode != 0;
  const bool plain = kp.softcap <= 0.f && kp.alibi == nullptr && kp.window < 0;
  // LDS-DMA staging of the 64-row classes: the slot strides must fit the buffer descriptor's 14-bit stride
  // field (SLM_ATTN_TILE_PF = 4 keeps the register-staged form for A/B runs)
  // the cross-tile pipeline: the instantiations without soft-cap / alibi / window (with them the head_dim-128 form
  // spills 22...31 VGPRs at its 256-register cap); SLM_ATTN_TILE_PF = 5: LDS-DMA staging without it
  const bool pipe = pf_mode != 5;
  // two wave groups per query tile (KV2): measured (profiles/r05_prefill_tile_kv2.jsonl) it LOSES wherever the grid
  // fills the chip with 4-wave workgroups -- eight waves on one barrier, even half an iteration apart, run a tile
  // slower than two independent workgroups (chunked 8 x 256: 915 -> 793 TFLOP/s, causal 4 x 1024: 561 -> 445, 1 x 2048
  // the same) -- and wins where the 4-wave grid leaves CUs empty: one 256-token chunk over an 8 k history 65 -> 58 us.
  // Automatic only there (fewer workgroups than CUs); SLM_ATTN_TILE_KV2: 1 = always where the form exists, 0 = never.
  const int kv2_mode = tune_get(TUNE_ATTN_TILE_KV2, -1);
  (void)max_kv_len;
  const bool kv2 = kv2_mode > 0 || (kv2_mode < 0 && grid <= 256);
  const bool dma = pf_mode != 4 && 2 * kp.k_ss < 16384 && 2 * kp.v_ss < 16384 && 2 * kp.k_ss > 0 && 2 * kp.v_ss > 0;
#define SLM_TILE(TT, HDD, NWW)                                                                    \
  do {                                                                                            \
    if (pf && plain) hipLaunchKernelGGL((attn_tile_kernel<TT, HDD, NWW, true, true>), g, blk, 0, st, kp, (int)tiles_per_seq); \
    else if (pf) hipLaunchKernelGGL((attn_tile_kernel<TT, HDD, NWW, true, false>), g, blk, 0, st, kp, (int)tiles_per_seq); \
    else hipLaunchKernelGGL((attn_tile_kernel<TT, HDD, NWW, false, false>), g, blk, 0, st, kp, (int)tiles_per_seq); \
  } while (0)
  // the prefill classes (2 / 4 waves): 64-row KV tiles, double-buffered (SLM_ATTN_TILE_PF=2 keeps the
  // 32-row single-buffer form for A/B runs)
#define SLM_TILE64(TT, HDD, NWW)                                                                  \
  do {                                                                                            \
    if (plain && dma && pipe && kv2 && NWW == 4 && HDD == 128) hipLaunchKernelGGL((attn_tile_kernel<TT, HDD, NWW == 4 && HDD == 128 ? 4 : NWW, true, true, 64, true, true, true, NWW == 4 && HDD == 128>), g, dim3(blk.x * 2), 0, st, kp, (int)tiles_per_seq); \
    else if (plain && dma && pipe) hipLaunchKernelGGL((attn_tile_kernel<TT, HDD, NWW, true, true, 64, true, true, true>), g, blk, 0, st, kp, (int)tiles_per_seq); \
    else if (plain && dma) hipLaunchKernelGGL((attn_tile_kernel<TT, HDD, NWW, true, true, 64, true, true>), g, blk, 0, st, kp, (int)tiles_per_seq); \
    else if (dma) hipLaunchKernelGGL((attn_tile_kernel<TT, HDD, NWW, true, false, 64, true, true>), g, blk, 0, st, kp, (int)tiles_per_seq); \
    else if (plain) hipLaunchKernelGGL((attn_tile_kernel<TT, HDD, NWW, true, true, 64, true>), g, blk, 0, st, kp, (int)tiles_per_seq); \
    else hipLaunchKernelGGL((attn_tile_kernel<TT, HDD, NWW, true, false, 64, true>), g, blk, 0, st, kp, (int)tiles_per_seq); \
  } while (0)
  // 64-row tiles where the instantiation keeps two waves per SIMD (no AGPR overflow): head_dim 128 with 4
  // waves, head_dim 64 with 2 or 4 (head_dim 128 x 2 waves stages 8 + 8 rows of K / V per thread and
  // lands at 254 VGPRs + 104 AGPRs = one wave per SIMD: it keeps the 32-row form)
#define SLM_TILE_NW128(TT)                                                                        \
  do {                                                                                            \
    if (nw == 1) SLM_TILE(TT, 128, 1);                                                            \
    else if (nw == 2) SLM_TILE(TT, 128, 2);                                                       \
    else if (pf_mode == 1 || pf_mode == 4 || pf_mode == 5) SLM_TILE64(TT, 128, 4);                                                \
    else SLM_TILE(TT, 128, 4);                                                                    \
  } while (0)
#define SLM_TILE_NW64(TT)                                                                         \
  do {                                                                                            \
    if (nw == 1) SLM_TILE(TT, 64, 1);                                                             \
    else if ((pf_mode == 1 || pf_mode == 4 || pf_mode == 5) && nw == 2) SLM_TILE64(TT, 64, 2);                                      \
    else if (pf_mode == 1 || pf_mode == 4 || pf_mode == 5) SLM_TILE64(TT, 64, 4);                                                 \
    else if (nw == 2) SLM_TILE(TT, 64, 2); else SLM_TILE(TT, 64, 4);                              \
  } while (0)
  if (dtype == SLM_BF16) {
    if (kp.head_dim == 128) SLM_TILE_NW128(bf16_tag); else SLM_TILE_NW64(bf16_tag);
  } else {
    if (kp.head_dim == 128) SLM_TILE_NW128(f16_tag); else SLM_TILE_NW64(f16_tag);
  }
#undef SLM_TILE_NW128
#undef SLM_TILE_NW64
#undef SLM_TILE64
#undef SLM_TILE
  return hip_check_launch();
}

}  // namespace slm
