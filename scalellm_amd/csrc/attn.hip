// attn.hip -- paged-KV varlen attention for gfx950 (CDNA4), hand-written HIP.
//
// Replaces llm::paged_kv_varlen_mha (reference src/kernels/attention/attn_api.cpp:14-73,
// kernel src/kernels/attention/kernel/sm80_kernel_mha.cuh:174-335).  Designed from the
// byte stream, not from the reference's 64x64 MMA tile:
//
//  * token-major work items: one workgroup = (query token, group of KV heads, q-head chunk,
//    KV split).  Decode (q_len = 1), speculative verify (q_len = k+1) and prefill chunks are
//    the same kernel with a per-token visible range [lo, hi) (causal diagonal
//    kv_len - q_len + qi, sliding window) -- sm80_kernel_mha.cuh:255-262, common/mask.h:49-89.
//  * the KV row of a slot, [n_kv_heads][head_dim] T, is contiguous: a wave's 64 lanes x 16 B
//    load covers 1 KiB of it, i.e. 64/LPR (slot, kv-head) units with LPR = head_dim/8 lanes
//    per unit.  When n_kv_heads >= 64/LPR the units of one load are adjacent heads of ONE slot
//    (fully contiguous 1 KiB); otherwise they are heads x consecutive rows.
//  * every lane group owns its (kv head, row phase): it keeps running (m, l, O[GC][8]) in
//    registers for the GC query heads that share the kv head -- no cross-lane traffic in the
//    loop except the LPR-lane DPP reduction of the q.k partial sums, no LDS on the data path
//    (K/V are streamed once; nothing is reused across waves -- see DESIGN.md for why LDS
//    staging is not used on the decode stream).
//  * block table slice staged in LDS once per workgroup (one coalesced read), slot =
//    tbl[row >> log2(bs)] + (row & (bs-1)) -- bit-exact with sm80_kernel_mha.cuh:146-152.
//  * U-deep register ring of 16-B K/V loads per lane keeps >= 2*U KiB per wave in flight.
//  * online softmax in base 2 (common/online_softmax.cuh:39-162 semantics), fp32 accumulate,
//    exact conditional rescale (skipped when no running max moved in the wave).
//  * split-KV partials (m, l, O) + combine kernel (math of attn_combine_kernel.cuh:14-21).
#include "attn_common.h"
#include "tuning.h"

namespace slm {

// P.V as packed dot products over row pairs (P rounded to T) instead of fp32 FMAs: ~45 % fewer
// VALU issue slots in the P.V part of the loop (profiles/README.md)
#ifndef SLM_ATTN_PV_DOT2
#define SLM_ATTN_PV_DOT2 1
#endif
constexpr bool PV_DOT2 = SLM_ATTN_PV_DOT2 != 0;

template <bool NT>
__device__ __forceinline__ u32x4 ld16(const void* p) {
  if constexpr (NT)
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  else
    return *reinterpret_cast<const u32x4*>(p);
}

// FX: the rare score transforms (logits soft-cap and/or alibi) are compiled in; the FX = false fast
// path folds the softmax scale into one fma per score, like the reference's exp2(x*s - max*s)
// (common/online_softmax.cuh:39-162).
// W (round 4): 16-byte chunks of a K / V row per lane.  W = 1: LPR lanes x 8 dims cover the row.  W = 2: a
// lane owns TWO chunks (dims [8 sub, 8 sub + 8) and [8 LPR + 8 sub, ...)), so half as many lanes share a row:
// the 16-lane reduction of every score becomes an 8-lane one, and the softmax arithmetic every lane of a
// group repeats for its rows is done for half as many rows per lane -- the same bytes in flight (U rows x W
// chunks), ~37 % fewer VALU instructions per KV byte.  The stream kernel was at the HBM rate already; what
// this buys is ISSUE SLOTS for the int4 GEMMs that share the SIMDs in the two-lane decode step.
template <typename T, int LPR, int GC, int U, bool NT, bool FX, int W = 1>
__global__ void __launch_bounds__(512) attn_token_kernel(const AttnKParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* tbl = reinterpret_cast<int*>(smem);
  float* xch = reinterpret_cast<float*>(smem + ATTN_TBL_ENT * sizeof(int));
  // Wave priority (SLM_ATTN_PRIO, p.prio): in the two-lane decode step (decode.py) this kernel shares
  // the CUs with the other lane's int4 GEMMs.  The KV stream is the step's critical resource and the
  // GEMMs are filler: with a raised priority the stream's waves win the issue arbitration against
  // co-resident GEMM waves, which then run in the gaps the stream leaves (its waves wait on HBM most
  // of the time).  No effect when the kernel has the chip to itself.
  if (p.prio > 0) __builtin_amdgcn_s_setprio(3);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;

  int bid = blockIdx.x;
  const int split = bid % p.n_splits;
  bid /= p.n_splits;
  const int chunk = bid % p.n_chunks;
  bid /= p.n_chunks;
  const int hgb = bid % p.nhgb;
  const int tok = bid / p.nhgb;

  // token -> sequence: q_cu[b] <= tok < q_cu[b+1].  Pure decode batches (one query token per
  // sequence: q_cu[i] == i) are recognised with two independent loads instead of log2(batch)
  // dependent ones -- the binary search is ~0.4 us per step of start-up latency in front of every
  // workgroup's stream, which is what small batches are made of.
  int b;
  const int tq0 = p.q_cu[min(tok, p.batch)], tq1 = p.q_cu[min(tok + 1, p.batch)];
  if (tok < p.batch && tq0 == tok && tq1 == tok + 1) {
    b = tok;
  } else {
    int lo_b = 0, hi_b = p.batch;
    while (lo_b < hi_b) {
      const int mid = (lo_b + hi_b) >> 1;
      if (p.q_cu[mid + 1] <= tok)
        lo_b = mid + 1;
      else
        hi_b = mid;
    }
    b = lo_b;
  }
  // Balanced pure-decode partition (p.bal; plan_attn): this workgroup is number w of its head
  // group and owns the tokens [g, g1) of the concatenated histories; it walks the sequences that
  // range touches ("pieces"), each piece a complete pass of the body below with its own partial.
  // The plan enables it from host hints (max_q_len <= 1, n_tokens == batch_size); the kernel
  // re-checks the one thing it relies on -- token index == sequence index -- on the device: with
  // every q_len <= 1 (max_q_len is the grid contract, tile_scheduler.cuh:23-27) q_cu[batch] == batch
  // means q_cu[i] == i.  A batch that keeps q = 0 sequences next to graph-padding rows
  // (slm_build_step_inputs allows it) fails the check and runs the classic partition, which follows
  // q_cu (same grid; part_slots > n_splits, so the partial slots are there).
  const bool bal = p.bal && p.q_cu[p.batch] == p.batch;
  int bal_g = 0, bal_g1 = 0, bal_Q = 1;  // (all < 2^31: kv_cu is int32)
  if (bal) {
    const int Wtot = p.kv_cu[p.batch];
    bal_Q = attn_bal_q(p, Wtot);
    const int64_t g64 = (int64_t)(tok * p.n_splits + split) * bal_Q;
    if (g64 >= (int64_t)Wtot) return;
    bal_g = (int)g64;
    bal_g1 = Wtot - bal_g > bal_Q ? bal_g + bal_Q : Wtot;
    int lo_b = 0, hi_b = p.batch;  // sequence holding token g: kv_cu[b] <= g < kv_cu[b + 1]
    while (lo_b < hi_b) {
      const int mid = (lo_b + hi_b) >> 1;
      if (p.kv_cu[mid + 1] <= bal_g) lo_b = mid + 1; else hi_b = mid;
    }
    b = lo_b;
  }
  if (b >= p.batch) return;  // padding token past q_cu[batch]
  if (!bal) {
    const int rows = (p.q_cu[b + 1] - p.q_cu[b]) * p.group;
    if (rows < p.rows_lo || rows >= p.rows_hi) return;  // another launch owns this sequence
  }

  // lane / wave decomposition
  constexpr int UPW = 64 / LPR;
  const int g = lane / LPR;
  const int sub = lane % LPR;
  const int HPW = 1 << p.hpw_shift;
  const int hsub = g & (HPW - 1);
  const int rsub = g >> p.hpw_shift;
  const int RPW = UPW >> p.hpw_shift;
  const int HGW = 1 << p.hgw_shift;
  const int hgw = wave & (HGW - 1);
  const int rp = wave >> p.hgw_shift;
  const int RP = nw >> p.hgw_shift;
  const int RPI = RP * RPW;  // rows per workgroup iteration

  const int kvh = (((hgb << p.hgw_shift) + hgw) << p.hpw_shift) + hsub;
  const int qh0_lane = kvh * p.group + chunk * GC;
  constexpr int CH = 16 * LPR;              // bytes between the W chunks of a lane (W = 2 needs head_dim = 16 LPR)
  const bool act = (sub * 8) < p.head_dim;  // head_dim < 8*LPR leaves idle lanes (D = 40, 96)
  const int sub_ld = act ? sub : 0;  // idle lanes (q = 0) re-read dims 0..7: finite, never stored
  const char* kbase = reinterpret_cast<const char*>(p.kc) + 2 * ((int64_t)kvh * p.k_hs + sub_ld * 8);
  const char* vbase = reinterpret_cast<const char*>(p.vc) + 2 * ((int64_t)kvh * p.v_hs + sub_ld * 8);
  // slot stride in bytes (< 2^32: plan_attn checks); slot >= 0 -> one v_mad_u64_u32 per address
  const uint32_t k_sb = (uint32_t)(2 * p.k_ss), v_sb = (uint32_t)(2 * p.v_ss);
  float slope2[GC];
#pragma unroll
  for (int h = 0; h < GC; ++h) slope2[h] = p.alibi ? p.alibi[qh0_lane + h] * LOG2E : 0.f;

  for (;;) {  // one pass per piece (classic partition: exactly one)
  int tok_p = tok, s_lo, s_hi, part_idx = split;
  bool single = p.n_splits == 1;  // the pass covers its sequence alone: it writes the final output
  bool skip = false;
  if (bal) {
    const int kv0 = p.kv_cu[b], kv1 = p.kv_cu[b + 1];
    tok_p = b;  // pure decode: token index == sequence index
    s_lo = bal_g - kv0;
    s_hi = min(kv1, bal_g1) - kv0;
    const int w_first = kv0 / bal_Q;
    const int n_pc = (kv1 - 1) / bal_Q - w_first + 1;  // pieces of this sequence
    part_idx = bal_g / bal_Q - w_first;
    single = n_pc == 1;
    if (n_pc > p.part_slots) {
      // longer than the max_kv_len hint promised (bal_qmin is derived from it): more pieces than
      // partial slots.  max_kv_len stays a HINT: the workgroup that holds the first piece streams
      // the whole sequence and writes the final row, the others skip it (correct for any hint,
      // unbalanced for that sequence); the combine kernel takes the same decision.
      if (part_idx == 0) {
        s_hi = kv1 - kv0;
        single = true;
      } else {
        skip = true;
        s_hi = s_lo;
      }
    }
  } else {
    const int q_start = p.q_cu[b];
    const int q_len = p.q_cu[b + 1] - q_start;
    const int kv_len = p.kv_cu[b + 1] - p.kv_cu[b];
    const int diag = kv_len - q_len + (tok - q_start);  // last visible kv index (causal)
    const int hi = min(diag + 1, kv_len);
    const int lo = (p.window >= 0) ? max(0, diag - p.window) : 0;
    // split range, aligned to RPI rows
    const int len = max(hi - lo, 0);
    int per = (len + p.n_splits - 1) / p.n_splits;
    per = ((per + RPI - 1) / RPI) * RPI;
    s_lo = lo + split * per;
    s_hi = min(hi, s_lo + per);
  }

  // q fragment: GC heads x W chunks x 8 dims (packed pairs)
  uint32_t qv[GC][4 * W];
#pragma unroll
  for (int h = 0; h < GC; ++h) {
#pragma unroll
    for (int c = 0; c < W; ++c) {
      u32x4 t = {0u, 0u, 0u, 0u};
      if (act) {
        const char* ptr = reinterpret_cast<const char*>(p.q) + c * CH +
                          2 * ((int64_t)tok_p * p.q_ts + (int64_t)(qh0_lane + h) * p.q_hs + sub * 8);
        t = *reinterpret_cast<const u32x4*>(ptr);
      }
      qv[h][4 * c + 0] = t.x; qv[h][4 * c + 1] = t.y; qv[h][4 * c + 2] = t.z; qv[h][4 * c + 3] = t.w;
    }
  }

  constexpr int OD = 8 * W;  // output dims per lane and head
  float m[GC], l[GC], o[GC][OD];
#pragma unroll
  for (int h = 0; h < GC; ++h) {
    m[h] = ATTN_M_INIT;
    l[h] = 0.f;
#pragma unroll
    for (int j = 0; j < OD; ++j) o[h][j] = 0.f;
  }

  const int bcu0 = p.bcu[b];

  u32x4 kr[U][W], vr[U][W];

  for (int c_lo = s_lo; c_lo < s_hi;) {
    const int blk0 = c_lo >> p.block_shift;
    const int blk_end = min(((s_hi - 1) >> p.block_shift) + 1, blk0 + ATTN_TBL_ENT);
    const int c_hi = min(s_hi, blk_end << p.block_shift);
    __syncthreads();
    for (int e = tid; e < blk_end - blk0; e += blockDim.x) tbl[e] = p.bt[bcu0 + blk0 + e];
    __syncthreads();

    const int wrow0 = c_lo + rp * RPW;  // wave-uniform first row of this wave in iteration 0
    const int n_it = (c_hi - c_lo + RPI - 1) / RPI;

    // Batched online softmax: U rows per lane group per loop trip, branch-free.
    //  - loads are unconditional with clamped rows, so the loop body is straight-line code and
    //    hipcc emits COUNTED s_waitcnt vmcnt(N): the K loads of the next batch are issued as
    //    each K register is consumed, the V loads as each V register is consumed -> 2*U 16-B
    //    loads per lane stay in flight across the loop back-edge;
    //  - the running max is updated once per batch (one O rescale per U rows, no branch);
    //  - past-the-end rows re-read the last valid row (cache hit) and are masked to -inf.
    int slot_n[U];
    auto slots_for = [&](int it0) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int row = min(wrow0 + (it0 + u) * RPI + rsub, c_hi - 1);
        slot_n[u] = tbl[(row >> p.block_shift) - blk0] + (row & p.block_mask);
      }
    };
    slots_for(0);
    // (W == 1 keeps its statements exactly as they were: the loop's schedule -- counted vmcnt waits, the
    // position of every load -- is what the 6.9 TB/s rest on, and hipcc re-derives it from the source shape)
    if constexpr (W == 1) {
#pragma unroll
      for (int u = 0; u < U; ++u) kr[u][0] = ld16<NT>(kbase + (uint64_t)(uint32_t)slot_n[u] * k_sb);
#pragma unroll
      for (int u = 0; u < U; ++u) vr[u][0] = ld16<NT>(vbase + (uint64_t)(uint32_t)slot_n[u] * v_sb);
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int c = 0; c < W; ++c) kr[u][c] = ld16<NT>(kbase + (uint64_t)(uint32_t)slot_n[u] * k_sb + c * CH);
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int c = 0; c < W; ++c) vr[u][c] = ld16<NT>(vbase + (uint64_t)(uint32_t)slot_n[u] * v_sb + c * CH);
    }
    // compiler fence: keeps hipcc from sinking the prologue / next-batch loads into the
    // consuming iteration (which would shorten the prefetch distance to < 1 batch)
    asm volatile("" ::: "memory");

    for (int it0 = 0; it0 < n_it; it0 += U) {
      slots_for(it0 + U);  // next batch (clamped past the end)
      float s[U][GC];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if constexpr (W == 1) {
          const u32x4 kk = kr[u][0];
#pragma unroll
          for (int h = 0; h < GC; ++h) {
            float a = dot2<T>(kk.x, qv[h][0], 0.f);
            a = dot2<T>(kk.y, qv[h][1], a);
            a = dot2<T>(kk.z, qv[h][2], a);
            s[u][h] = dot2<T>(kk.w, qv[h][3], a);
          }
          kr[u][0] = ld16<NT>(kbase + (uint64_t)(uint32_t)slot_n[u] * k_sb);
        } else {
#pragma unroll
          for (int c = 0; c < W; ++c) {
            const u32x4 kk = kr[u][c];
#pragma unroll
            for (int h = 0; h < GC; ++h) {
              float a = dot2<T>(kk.x, qv[h][4 * c + 0], c == 0 ? 0.f : s[u][h]);
              a = dot2<T>(kk.y, qv[h][4 * c + 1], a);
              a = dot2<T>(kk.z, qv[h][4 * c + 2], a);
              s[u][h] = dot2<T>(kk.w, qv[h][4 * c + 3], a);
            }
            kr[u][c] = ld16<NT>(kbase + (uint64_t)(uint32_t)slot_n[u] * k_sb + c * CH);
          }
        }
        __builtin_amdgcn_sched_barrier(0);  // pin: next-batch K load(s) issue right here
      }
      // 16-lane all-reduce of the U*GC partial sums, STEP-major: the U*GC independent chains are
      // interleaved so the VALU-write -> DPP-read wait states are filled with useful work
      group_sum_many<LPR, U * GC>(&s[0][0]);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int row = wrow0 + (it0 + u) * RPI + rsub;
        const bool valid = row < c_hi;
#pragma unroll
        for (int h = 0; h < GC; ++h) {
          float a = s[u][h];
          if constexpr (FX) {
            if (p.softcap > 0.f) a = fast_tanh(a * p.pre_scale);
            a = a * p.scale_log2 + slope2[h] * (float)row;
          }
          s[u][h] = valid ? a : -INFINITY;
        }
      }
      // one running-max update per batch of U rows (m is kept in score units when !FX)
      float negm[GC];
#pragma unroll
      for (int h = 0; h < GC; ++h) {
        float mn = m[h];
#pragma unroll
        for (int u = 0; u < U; ++u) mn = fmaxf(mn, s[u][h]);
        const float alpha = FX ? fast_exp2(m[h] - mn) : fast_exp2((m[h] - mn) * p.scale_log2);
        negm[h] = FX ? -mn : -mn * p.scale_log2;
        m[h] = mn;
        l[h] *= alpha;
#pragma unroll
        for (int j = 0; j < OD; ++j) o[h][j] *= alpha;
      }
      float pr[U][GC];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int h = 0; h < GC; ++h) {
          pr[u][h] = FX ? fast_exp2(s[u][h] + negm[h]) : fast_exp2(fmaf(s[u][h], p.scale_log2, negm[h]));
          l[h] += pr[u][h];
        }
      if constexpr (PV_DOT2 && (U % 2 == 0)) {
        // P.V over row PAIRS with v_dot2c: V words of two rows are byte-permuted into
        // (row a, row b) pairs per dim, P is rounded to T (as every MFMA flash-attention does)
#pragma unroll
        for (int u = 0; u < U; u += 2) {
          if constexpr (W == 1) {
            const u32x4 va = vr[u][0], vb = vr[u + 1][0];
            uint32_t pp[GC];
#pragma unroll
            for (int h = 0; h < GC; ++h) pp[h] = pack2<T>(pr[u][h], pr[u + 1][h]);
            const uint32_t wa[4] = {va.x, va.y, va.z, va.w}, wb[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const uint32_t lo = __builtin_amdgcn_perm(wb[j], wa[j], 0x05040100u);
              const uint32_t hi = __builtin_amdgcn_perm(wb[j], wa[j], 0x07060302u);
#pragma unroll
              for (int h = 0; h < GC; ++h) {
                o[h][2 * j] = dot2<T>(lo, pp[h], o[h][2 * j]);
                o[h][2 * j + 1] = dot2<T>(hi, pp[h], o[h][2 * j + 1]);
              }
            }
            vr[u][0] = ld16<NT>(vbase + (uint64_t)(uint32_t)slot_n[u] * v_sb);
            vr[u + 1][0] = ld16<NT>(vbase + (uint64_t)(uint32_t)slot_n[u + 1] * v_sb);
            __builtin_amdgcn_sched_barrier(0);  // pin: next-batch V loads issue right here
            continue;
          }
          uint32_t pp[GC];
#pragma unroll
          for (int h = 0; h < GC; ++h) pp[h] = pack2<T>(pr[u][h], pr[u + 1][h]);
#pragma unroll
          for (int c = 0; c < W; ++c) {
            const u32x4 va = vr[u][c], vb = vr[u + 1][c];
            const uint32_t wa[4] = {va.x, va.y, va.z, va.w}, wb[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const uint32_t lo = __builtin_amdgcn_perm(wb[j], wa[j], 0x05040100u);
              const uint32_t hi = __builtin_amdgcn_perm(wb[j], wa[j], 0x07060302u);
#pragma unroll
              for (int h = 0; h < GC; ++h) {
                o[h][8 * c + 2 * j] = dot2<T>(lo, pp[h], o[h][8 * c + 2 * j]);
                o[h][8 * c + 2 * j + 1] = dot2<T>(hi, pp[h], o[h][8 * c + 2 * j + 1]);
              }
            }
            vr[u][c] = ld16<NT>(vbase + (uint64_t)(uint32_t)slot_n[u] * v_sb + c * CH);
            vr[u + 1][c] = ld16<NT>(vbase + (uint64_t)(uint32_t)slot_n[u + 1] * v_sb + c * CH);
          }
          __builtin_amdgcn_sched_barrier(0);  // pin: next-batch V loads issue right here
        }
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
          for (int c = 0; c < W; ++c) {
            const u32x4 vv = vr[u][c];
            float vf[8];
            vf[0] = lo_f32<T>(vv.x); vf[1] = hi_f32<T>(vv.x);
            vf[2] = lo_f32<T>(vv.y); vf[3] = hi_f32<T>(vv.y);
            vf[4] = lo_f32<T>(vv.z); vf[5] = hi_f32<T>(vv.z);
            vf[6] = lo_f32<T>(vv.w); vf[7] = hi_f32<T>(vv.w);
#pragma unroll
            for (int h = 0; h < GC; ++h) {
#pragma unroll
              for (int j = 0; j < 8; ++j) o[h][8 * c + j] = fmaf(pr[u][h], vf[j], o[h][8 * c + j]);
            }
            vr[u][c] = ld16<NT>(vbase + (uint64_t)(uint32_t)slot_n[u] * v_sb + c * CH);
          }
          __builtin_amdgcn_sched_barrier(0);  // pin: next-batch V load issues right here
        }
      }
      asm volatile("" ::: "memory");
    }
    c_lo = c_hi;
  }

  if constexpr (!FX) {  // running max was kept in raw score units: convert to log2 units for the merges
#pragma unroll
    for (int h = 0; h < GC; ++h) m[h] *= p.scale_log2;
  }

  // ---- merge the row-phase lane groups of this wave (same kv head, different rows) ----
  auto merge = [&](const float (&om)[GC], const float (&ol)[GC], const float (&oo)[GC][OD]) {
#pragma unroll
    for (int h = 0; h < GC; ++h) {
      const float mn = fmaxf(m[h], om[h]);
      const float fa = fast_exp2(m[h] - mn);
      const float fb = fast_exp2(om[h] - mn);
      m[h] = mn;
      l[h] = l[h] * fa + ol[h] * fb;
#pragma unroll
      for (int j = 0; j < OD; ++j) o[h][j] = o[h][j] * fa + oo[h][j] * fb;
    }
  };
  for (int d = LPR << p.hpw_shift; d < 64; d <<= 1) {
    float om[GC], ol[GC], oo[GC][OD];
#pragma unroll
    for (int h = 0; h < GC; ++h) {
      om[h] = __shfl_xor(m[h], d, 64);
      ol[h] = __shfl_xor(l[h], d, 64);
#pragma unroll
      for (int j = 0; j < OD; ++j) oo[h][j] = __shfl_xor(o[h][j], d, 64);
    }
    merge(om, ol, oo);
  }

  // ---- merge row-phase waves through LDS (sequential rounds; once per workgroup) ----
  constexpr int SF = 2 + OD;    // floats of state per lane and head
  constexpr int NF = SF * GC;   // floats of state per lane
  float* my = xch + (size_t)hgw * NF * 64;
  for (int r = 1; r < RP; ++r) {
    __syncthreads();
    if (rp == r) {
#pragma unroll
      for (int h = 0; h < GC; ++h) {
        my[(h * SF + 0) * 64 + lane] = m[h];
        my[(h * SF + 1) * 64 + lane] = l[h];
#pragma unroll
        for (int j = 0; j < OD; ++j) my[(h * SF + 2 + j) * 64 + lane] = o[h][j];
      }
    }
    __syncthreads();
    if (rp == 0) {
      float om[GC], ol[GC], oo[GC][OD];
#pragma unroll
      for (int h = 0; h < GC; ++h) {
        om[h] = my[(h * SF + 0) * 64 + lane];
        ol[h] = my[(h * SF + 1) * 64 + lane];
#pragma unroll
        for (int j = 0; j < OD; ++j) oo[h][j] = my[(h * SF + 2 + j) * 64 + lane];
      }
      merge(om, ol, oo);
    }
  }

  if (rp == 0 && rsub == 0 && act && !skip) {
    // (opaque copy: keeps hipcc from hoisting the GC output addresses out of the piece loop, where
    // they would sit in -- or spill from -- registers across the whole stream)
    int qh0 = qh0_lane;
    asm volatile("" : "+v"(qh0));
    if (single) {
#pragma unroll
      for (int h = 0; h < GC; ++h) {
        const float inv = l[h] > 0.f ? 1.0f / l[h] : 0.f;
#pragma unroll
        for (int c = 0; c < W; ++c) {
          u32x4 r;
          r.x = pack2<T>(o[h][8 * c + 0] * inv, o[h][8 * c + 1] * inv);
          r.y = pack2<T>(o[h][8 * c + 2] * inv, o[h][8 * c + 3] * inv);
          r.z = pack2<T>(o[h][8 * c + 4] * inv, o[h][8 * c + 5] * inv);
          r.w = pack2<T>(o[h][8 * c + 6] * inv, o[h][8 * c + 7] * inv);
          char* ptr = reinterpret_cast<char*>(p.out) + c * CH +
                      2 * ((int64_t)tok_p * p.o_ts + (int64_t)(qh0 + h) * p.o_hs + sub * 8);
          *reinterpret_cast<u32x4*>(ptr) = r;
        }
      }
    } else {
#pragma unroll
      for (int h = 0; h < GC; ++h) {
        const int64_t pi = ((int64_t)tok_p * p.n_heads + (qh0 + h)) * p.part_slots + part_idx;
#pragma unroll
        for (int c = 0; c < W; ++c) {
          float* op = p.o_part + pi * p.head_dim + c * (CH / 2) + sub * 8;
          *reinterpret_cast<f32x4*>(op) = f32x4{o[h][8 * c + 0], o[h][8 * c + 1], o[h][8 * c + 2], o[h][8 * c + 3]};
          *reinterpret_cast<f32x4*>(op + 4) = f32x4{o[h][8 * c + 4], o[h][8 * c + 5], o[h][8 * c + 6], o[h][8 * c + 7]};
        }
        if (sub == 0) {
          p.ml_part[pi * 2 + 0] = m[h];
          p.ml_part[pi * 2 + 1] = l[h];
        }
      }
    }
  }
  if (!bal) break;
  // next piece: the following non-empty sequence, if this workgroup's range reaches into it
  do {
    ++b;
  } while (b < p.batch && p.kv_cu[b + 1] == p.kv_cu[b]);
  if (b >= p.batch) break;
  bal_g = p.kv_cu[b];
  if (bal_g >= bal_g1) break;
  }  // piece loop
}

// out[tok, head, :] = sum_s 2^(m_s - M) O_s / sum_s 2^(m_s - M) l_s   (one wave per (tok, head)).
// Latency-oriented: lanes first own SPLITS (one (m, l) pair each, wave-reduced to M and L, the
// weights parked in LDS), then own DIMS (4 consecutive floats per lane, several split phases
// per wave) with 4 independent 16-B loads in flight per lane -- no per-split dependent chain.
constexpr int COMBINE_MAX_SPLITS = 256;

template <typename T>
__global__ void __launch_bounds__(256) attn_combine_kernel(const AttnKParams p, int lps_shift) {
  __shared__ float wsm[4][COMBINE_MAX_SPLITS];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  // No early exit in front of the loads: out-of-range waves and graph-padding rows (past
  // q_cu_lens[batch]: no kernel of the call wrote their partials) run on clamped / unwritten but
  // in-bounds data and only skip the final store, so the q_cu lookup overlaps the partial loads
  // instead of adding a round trip in front of them.
  const int64_t n_items = (int64_t)p.n_tokens * p.n_heads;
  const int64_t item_raw = (int64_t)blockIdx.x * 4 + wv;
  bool valid = item_raw < n_items;
  const int64_t item = valid ? item_raw : n_items - 1;
  const int tok = (int)(item / p.n_heads), head = (int)(item % p.n_heads);
  const int q_end = p.q_cu[p.batch];
  if (p.rows_hi != 0x7fffffff || p.rows_lo != 0) {
    // mixed call: only the rows the token-major kernel produced are combined
    int lo_b = 0, hi_b = p.batch;
    while (lo_b < hi_b) {
      const int mid = (lo_b + hi_b) >> 1;
      if (p.q_cu[mid + 1] <= tok) lo_b = mid + 1; else hi_b = mid;
    }
    if (lo_b >= p.batch) return;
    const int rows = (p.q_cu[lo_b + 1] - p.q_cu[lo_b]) * p.group;
    if (rows < p.rows_lo || rows >= p.rows_hi) return;
  }
  int n_used = p.n_splits;
  const bool bal = p.bal && q_end == p.batch;  // the stream kernel's device-side check (attn_token_kernel)
  if (p.bal && !bal && p.n_splits == 1) return;  // classic fallback, one split: final rows already written
  if (bal) {
    // balanced pure-decode partition: the pieces of sequence tok come from the consecutive
    // workgroups floor(kv_cu[tok] / Q) .. floor((kv_cu[tok + 1] - 1) / Q) (attn_common.h)
    if (!valid || tok >= p.batch) return;
    const int kv0 = p.kv_cu[tok], kv1 = p.kv_cu[tok + 1];
    if (kv1 <= kv0) {  // no history: nobody streamed this sequence -- the output row is zero
      const int d0z = lane * 4;
      if (d0z < p.head_dim)
        *reinterpret_cast<u32x2*>(reinterpret_cast<char*>(p.out) +
                                  2 * ((int64_t)tok * p.o_ts + (int64_t)head * p.o_hs + d0z)) = u32x2{0u, 0u};
      return;
    }
    const int Q = attn_bal_q(p, p.kv_cu[p.batch]);
    n_used = (kv1 - 1) / Q - kv0 / Q + 1;
    // one piece, or more pieces than slots (a sequence past the max_kv_len hint: streamed whole by
    // the workgroup of its first piece): the final output is already written
    if (n_used == 1 || n_used > p.part_slots) return;
  }
  const float* ml = p.ml_part + item * p.part_slots * 2;
  // every load below is UNCONDITIONAL (index clamped, result masked afterwards): a load inside a
  // branch makes hipcc wait for it at the join, which turns independent loads into a chain of HBM
  // round trips (tools/probes/experiments/attn_in_kernel_combine.md)
  const int s_last = n_used - 1;
  float mreg[4], lreg[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float2 v = *reinterpret_cast<const float2*>(ml + 2 * min(lane + 64 * i, s_last));
    mreg[i] = v.x;
    lreg[i] = v.y;
  }
  const int LPS = 1 << lps_shift;      // lanes per split (>= head_dim/4)
  const int phase = lane >> lps_shift;  // split phase of this lane
  const int n_phase = 64 >> lps_shift;
  const int d0 = (lane & (LPS - 1)) * 4;
  const bool actd = d0 < p.head_dim;  // idle dim lanes (head_dim < 4 LPS) re-read dims 0..3
  const float* op = p.o_part + item * p.part_slots * p.head_dim + (actd ? d0 : 0);
  // first batch of O loads: issued BEFORE the weights are known (they do not depend on them)
  constexpr int NFIRST = 8;
  f32x4 first[NFIRST];
#pragma unroll
  for (int i = 0; i < NFIRST; ++i)
    first[i] = *reinterpret_cast<const f32x4*>(op + (int64_t)min(phase + i * n_phase, s_last) * p.head_dim);
  float M = ATTN_M_INIT;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (lane + 64 * i > s_last) lreg[i] = 0.f;
    if (lreg[i] > 0.f) M = fmaxf(M, mreg[i]);
  }
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) M = fmaxf(M, __shfl_xor(M, d, 64));
  float L = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float w = lreg[i] > 0.f ? fast_exp2(mreg[i] - M) : 0.f;
    L += w * lreg[i];
    wsm[wv][lane + 64 * i] = w;
  }
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) L += __shfl_xor(L, d, 64);
  __builtin_amdgcn_wave_barrier();  // wsm row is wave-private; LDS ops of one wave are ordered

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NFIRST; ++i) {
    const int s = phase + i * n_phase;
    acc += (s <= s_last ? wsm[wv][min(s, COMBINE_MAX_SPLITS - 1)] : 0.f) * first[i];
  }
  for (int s = phase + NFIRST * n_phase; s <= s_last; s += 4 * n_phase) {
    const int s1 = s + n_phase, s2 = s + 2 * n_phase, s3 = s + 3 * n_phase;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(op + (int64_t)s * p.head_dim);
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(op + (int64_t)min(s1, s_last) * p.head_dim);
    const f32x4 a2 = *reinterpret_cast<const f32x4*>(op + (int64_t)min(s2, s_last) * p.head_dim);
    const f32x4 a3 = *reinterpret_cast<const f32x4*>(op + (int64_t)min(s3, s_last) * p.head_dim);
    acc += wsm[wv][s] * a0 + (s1 <= s_last ? wsm[wv][s1] : 0.f) * a1 +
           (s2 <= s_last ? wsm[wv][s2] : 0.f) * a2 + (s3 <= s_last ? wsm[wv][s3] : 0.f) * a3;
  }
  for (int d = LPS; d < 64; d <<= 1) {
    acc.x += __shfl_xor(acc.x, d, 64);
    acc.y += __shfl_xor(acc.y, d, 64);
    acc.z += __shfl_xor(acc.z, d, 64);
    acc.w += __shfl_xor(acc.w, d, 64);
  }
  if (valid && tok < q_end && phase == 0 && actd) {
    const float inv = L > 0.f ? 1.0f / L : 0.f;
    u32x2 r;
    r.x = pack2<T>(acc.x * inv, acc.y * inv);
    r.y = pack2<T>(acc.z * inv, acc.w * inv);
    char* optr = reinterpret_cast<char*>(p.out) +
                 2 * ((int64_t)tok * p.o_ts + (int64_t)head * p.o_hs + d0);
    *reinterpret_cast<u32x2*>(optr) = r;
  }
}

// ------------------------------- host side ------------------------------------------
struct AttnPlan {
  int lpr, gc, hpw_shift, hgw_shift, nhgb, n_chunks, nw, n_splits, u, w;
  bool nt;
  size_t lds_bytes;
  // balanced pure-decode partition (AttnKParams::bal): partial slots per (token, head) and the
  // piece-size floor that keeps a sequence's pieces within them
  int bal, part_slots, bal_qmin;
};
constexpr int ATTN_BAL_ALIGN = 64;

// q_len = 1 sequences on the MFMA tile kernel?  With a wide GQA group the token kernel's VALU work
// per KV byte (one dot-product / P.V chain per query head) is what bounds it -- G = 8: 4.4-5.3 TB/s
// -- while the tile kernel's cost per KV byte does not depend on how many of its 32 query rows are
// real: measured at G = 8, bs = 128, 4 k context: 1 KV head (the 70B TP = 8 rank) 61.2 -> 45.8 us,
// 8 KV heads (70B TP = 1) 487 -> 370 us, 4.4 -> 5.8 TB/s.  At G = 4 (Llama-3-8B) the token kernel's
// 6.9 TB/s stays ahead.  SLM_ATTN_TILE_DECODE: minimum group size (default 8), 0 = never.
static bool decode_on_tile(const slm_attn_args* a) {
  const int G = a->n_heads / a->n_kv_heads;
  const int g_min = tune_get(TUNE_ATTN_TILE_DECODE, 8);
  // small launches stay on the token kernel: with few (sequence, KV head) pairs the tile kernel's
  // per-tile round trip dominates (8 pairs: 22 vs 13 us; 128 pairs: 48 vs 62 us; 256: 86 vs 128 us)
  return g_min > 0 && G >= g_min && G <= 32 && a->num_splits <= 0 && attn_tile_supported(a->head_dim) &&
         tune_get(TUNE_ATTN_TILE, 1) != 0 && (int64_t)a->batch_size * a->n_kv_heads >= 64;
}

static int plan_attn(const slm_attn_args* a, AttnPlan* pl) {
  if (!a) return SLM_ERR_INVALID_ARG;
  if (a->n_heads <= 0 || a->n_kv_heads <= 0 || a->n_heads % a->n_kv_heads) return SLM_ERR_INVALID_ARG;
  if (a->head_dim <= 0 || a->head_dim > 256 || a->head_dim % 8) return SLM_ERR_UNSUPPORTED;
  if (!is_pow2(a->block_size)) return SLM_ERR_INVALID_ARG;
  if (a->dtype != SLM_F16 && a->dtype != SLM_BF16) return SLM_ERR_UNSUPPORTED;
  if (a->k_stride[0] <= 0 || a->v_stride[0] <= 0 || a->k_stride[0] >= (1ll << 30) ||
      a->v_stride[0] >= (1ll << 30))
    return SLM_ERR_INVALID_ARG;
  const int D = a->head_dim, G = a->n_heads / a->n_kv_heads;
  pl->lpr = D <= 32 ? 4 : D <= 64 ? 8 : D <= 128 ? 16 : 32;
  pl->gc = (G % 8 == 0) ? 8 : (G % 4 == 0) ? 4 : (G % 2 == 0) ? 2 : 1;
  // Two 16-byte chunks of a row per lane (attn_token_kernel W = 2): head_dim 128 on 8 lanes per row, plain
  // softmax only (the soft-cap / alibi instantiation keeps one chunk), <= 4 query heads per lane (8 x 16 output
  // dims do not fit the registers).  Measured (profiles/r04_attn_w2.jsonl): alone the two forms stream at the same
  // rate from 64 sequences up with >= 4 KV heads per rank (32 / 8 heads, bs 256: 638 vs 642 us; 16 / 4: 327 vs 338)
  // and W = 2 loses below that (bs 1: 15.4 vs 13.3 us; one KV head, bs 64: 32.1 vs 30.3) -- but it needs 19 % fewer
  // instructions per KV byte, and in the two-lane decode step those issue slots go to the other lane's GEMMs
  // (qkv / o / gate_up 82 -> 58 us next to the stream): bs 128 14.09 -> 13.83 ms, bs 256 24.09 -> 23.85.
  // SLM_ATTN_W: 1 / 2 force a form.
  pl->w = 1;
  if (D == 128 && pl->gc <= 4 && a->logits_soft_cap <= 0.f && a->alibi_slopes == nullptr) {
    const int w_knob = tune_get(TUNE_ATTN_W, 0);
    if (w_knob == 2 || (w_knob != 1 && a->n_tokens >= 64 && a->n_kv_heads >= 4)) {
      pl->w = 2;
      pl->lpr = 8;
    }
  }
  const int upw = 64 / pl->lpr;
  pl->n_chunks = G / pl->gc;
  int hpw = 1;
  while (hpw * 2 <= upw && a->n_kv_heads % (hpw * 2) == 0) hpw *= 2;
  // decode batches up to 128 tokens: one KV head per wave load (4 consecutive slots of it instead of 4
  // heads of one slot) gives 4x the head groups -- a finer grain for the workgroup count, so the grid
  // fills whole rounds of the CUs with fewer KV splits and the combine merges fewer partials.
  // Measured on one box, old -> new plan (32 q / 8 kv heads, 4 k context, uniform | ragged U[2048, 4096]):
  // bs 1 14.9 -> 13.5 us, 4 22.4 -> 20.9, 8 31.5 -> 29.2, 12 46.1 -> 42.9, 24 81.0 -> 76.1 | 78.7 -> 71.0,
  // 48 142.3 -> 139.8 | 128.6 -> 106.2, 64 176.2 -> 175.8 | 143.2 -> 136.4; equal within 2 % at 16, 32, 96,
  // 128; above 128 tokens the wide load wins or ties (160: 430 vs 449 us)
  const int hpw_max = hpw;
  if (a->max_q_len <= 1 && a->n_tokens <= 128) hpw = 1;
  // Launch shape tuned on MI355X (tools/sweep_attn.py, profiles/attn_sweep_r1.md): 4 waves per
  // workgroup, ~256 workgroups per launch, >= 64 KV rows per split.
  pl->nw = tune_get(TUNE_ATTN_NW, 4);
  if (pl->nw != 1 && pl->nw != 2 && pl->nw != 4 && pl->nw != 8) pl->nw = 4;
  const size_t state = (size_t)(2 + 8 * pl->w) * pl->gc * 64 * sizeof(float);
  const int64_t target_wgs = 256;
  const int64_t max_by_len = (a->max_kv_len > 64 ? a->max_kv_len : 64) / 64;
  int forced_splits = a->num_splits > 0 ? a->num_splits : tune_get(TUNE_ATTN_SPLITS, 0);
  struct Shape { int hgw, n_splits; double fill; };
  auto grid_fill = [&](int64_t w) {
    return w > 0 ? (double)w / (double)(((w + target_wgs - 1) / target_wgs) * target_wgs) : 0.0;
  };
  // head groups per workgroup and KV splits for `hpw_c` KV heads per wave load
  auto shape_for = [&](int hpw_c) -> Shape {
  const int nhg = a->n_kv_heads / hpw_c;
  int hgw_cap = tune_get(TUNE_ATTN_HGW, pl->nw);
  int hgw = 1, n_splits = 1;
  int64_t base = 1;
  for (int pass = 0; pass < 2; ++pass) {
    // LDS: table + HGW x per-wave exchange state, kept under the 64 KiB default dynamic limit
    hgw = 1;
    while (hgw * 2 <= pl->nw && hgw * 2 <= hgw_cap && nhg % (hgw * 2) == 0 &&
           (size_t)(hgw * 2) * state <= (pl->w == 2 ? 96 : 48) * 1024)  // (W = 2: 18 KiB per head group, opted in at launch)
      hgw *= 2;
    base = (int64_t)a->n_tokens * (nhg / hgw) * pl->n_chunks;
    int64_t want = (target_wgs + base - 1) / (base > 0 ? base : 1);
    if (want > max_by_len) want = max_by_len;
    if (want < 1) want = 1;
    if (want > COMBINE_MAX_SPLITS) want = COMBINE_MAX_SPLITS;
    if (a->max_q_len <= 1) {  // (pure decode: n_tokens IS the token kernel's row count)
      // workgroup-count quantisation: 288 workgroups on 256 CUs take as long as 512 (bs = 96:
      // 3 splits = 288 workgroups ran at 4.5 TB/s).  Look a few split counts further for one whose
      // grid fills whole rounds of the CUs (>= 90 %), and take the best seen otherwise.
      auto fill = [&](int64_t sp) { return grid_fill(base * sp); };
      int64_t s_max = want * 4 > want + 3 ? want * 4 : want + 3;
      if (s_max > max_by_len) s_max = max_by_len;
      if (s_max > COMBINE_MAX_SPLITS) s_max = COMBINE_MAX_SPLITS;
      int64_t best = want;
      double best_fill = fill(want);
      for (int64_t sp = want + 1; sp <= s_max && best_fill < 0.9; ++sp) {
        if (fill(sp) > best_fill + 0.02) {
          best = sp;
          best_fill = fill(sp);
        }
      }
      want = best;
    }
    n_splits = forced_splits > 0 ? forced_splits : (int)want;
    // tiny batches: if the grid is still short of one workgroup per CU, stop sharing a
    // workgroup between head groups (doubles / quadruples the workgroup count)
    if (pass == 0 && base * n_splits < target_wgs && hgw > 1 && !tune_is_set(TUNE_ATTN_HGW))
      hgw_cap = 1;
    else
      break;
  }
  return Shape{hgw, n_splits, grid_fill(base * n_splits)};
  };
  Shape sh = shape_for(hpw);
  if (a->max_q_len <= 1 && a->n_tokens > 128 && hpw_max > 1 && forced_splits <= 0) {
    // between 129 and ~255 tokens neither grain wins everywhere (160: wide 430 vs narrow 449 us;
    // 192: 516 vs 504): take the one whose grid fills its last round of CUs better, then the one
    // with fewer KV splits; a tie keeps the wide load (bs = 256: both 1 split, full rounds)
    const Shape alt = shape_for(1);
    if (alt.fill > sh.fill + 0.02 || (alt.fill >= sh.fill - 0.02 && alt.n_splits < sh.n_splits)) {
      sh = alt;
      hpw = 1;
    }
  }
  pl->hpw_shift = ilog2(hpw);
  const int nhg = a->n_kv_heads / hpw;
  const int hgw = sh.hgw;
  int n_splits = sh.n_splits;
  pl->hgw_shift = ilog2(hgw);
  pl->nhgb = nhg / hgw;
  pl->lds_bytes = ATTN_TBL_ENT * sizeof(int) + (size_t)hgw * state;
  // MFMA tile kernel (q_len > 1) with a long KV history and few query tiles -- speculative verify,
  // chunked prefill: its workgroups (one per 32 / 128 query rows per KV head) are too few to hide
  // HBM latency, so the KV range is split for it too (same partial format, same combine pass).
  // The count is shared by every kernel of the call.  Plain prefill (kv ~ q) is never split.
  if ((a->max_q_len > 1 || decode_on_tile(a)) && forced_splits <= 0 && attn_tile_supported(D) &&
      tune_get(TUNE_ATTN_TILE, 1) != 0 && a->max_kv_len >= 4 * (int64_t)a->max_q_len) {
    const int64_t rows_max = (int64_t)a->max_q_len * G;
    const int64_t nw_t = rows_max <= 32 ? 1 : rows_max <= 64 ? 2 : 4;
    int64_t tiles = ((int64_t)a->n_tokens * G + 32 * nw_t - 1) / (32 * nw_t);
    if (tiles < a->batch_size) tiles = a->batch_size;
    const int64_t waves = tiles * a->n_kv_heads * nw_t;
    int64_t want = tune_get(TUNE_ATTN_TILE_SPLITS, 0);
    if (want <= 0) {
      // just enough to put a wave on every SIMD: measured (tools/bench_prefill.py), splitting
      // beyond that never pays -- with the chip full the kernel sits at its memory-system rate
      // (~5 TB/s for 256-B rows at a 2-KiB stride) whatever the occupancy
      want = 1024 / (waves > 0 ? waves : 1);
      const int64_t by_len = a->max_kv_len / 512;            // >= 16 KV tiles per split
      if (want > by_len) want = by_len;
      if (want > 16) want = 16;
    }
    if (want > n_splits) n_splits = (int)want;
    // pure decode on the tile kernel: the token kernel does not run, its split count does not apply
    if (a->max_q_len <= 1) n_splits = (int)(want > 0 ? want : 1);
  }
  if (n_splits > COMBINE_MAX_SPLITS) n_splits = COMBINE_MAX_SPLITS;
  pl->n_splits = n_splits;
  // Pure decode on the token kernel: the SAME number of workgroups, but each takes an equal share
  // of the batch's concatenated KV tokens instead of 1 / n_splits of its own sequence.  With one
  // round of resident workgroups (bs >= 128) the classic partition finishes with the LONGEST
  // sequence -- kv_len ~ U[2048, 4096] ran at 5.75 TB/s against 6.73 uniform
  // (profiles/r03_attn_serving.jsonl); the lengths live on the device, so the partition is derived
  // there (kv_cu_lens is its own prefix sum).  A sequence is cut into at most
  // ceil(len / Q) + 1 pieces; Q >= max_kv_len / (slots - 1) bounds that by the partial slots.
  // Uniform batches come out as one piece per sequence: final output written directly, the
  // combine launch exits.  Not with a caller-forced split count (tests pin the classic form),
  // not with a sliding window (the visible range is no longer a prefix sum).
  pl->bal = 0;
  pl->part_slots = n_splits;
  pl->bal_qmin = 1;
  // Measured (profiles/r03_attn_serving.jsonl, classic -> balanced): bs = 256 ragged 5.80 -> 6.43 TB/s,
  // uniform 6.91 -> 6.85 (the combine launch that only exits); bs = 32 ragged 5.07 -> 5.37, uniform
  // 5.73 -> 5.65; bs = 8 loses both ways (few sequences: the binary search and the piece
  // bookkeeping are not amortised) -- hence the batch floor.  And only when a workgroup streams
  // >= 4 KV heads of its tokens: with one KV head per workgroup (a TP = 8 shard of Llama-3: 4 q / 1 kv
  // heads, 2 MiB per workgroup) the same bookkeeping is 5 % of the launch and a ragged batch
  // gains nothing (bs = 256: uniform 86.5 -> 91.0 us, ragged 77.8 -> 77.5 us).
  // SLM_ATTN_BAL: 0 = never, 2 = always.
  const int bal_mode = tune_get(TUNE_ATTN_BAL, 1);
  // ... and only while a workgroup's share stays long: with many KV splits (bs = 24: 16 splits, shares of
  // ~200 tokens) the per-piece bookkeeping and the extra partials cost more than the balance gains
  // (bs = 24 ragged 63.9 us classic vs 71.6 balanced; bs = 32, 4 splits: 80.3 vs 75.7)
  const bool bal_pays = ((1 << pl->hpw_shift) << pl->hgw_shift) >= 4 &&
                        a->max_kv_len / (pl->n_splits > 0 ? pl->n_splits : 1) >= 768;
  // the host says every sequence has max_kv_len tokens (slm_attn_args::total_kv_len) and the classic
  // plan needs no KV split: one workgroup per (token, head group) already IS the balanced partition
  // -- and writes the final rows itself, so the call is ONE launch (the balanced form would add a
  // combine launch that finds nothing to merge: ~5 us per layer, 64 times per two-lane step).  A wrong
  // claim is still computed correctly: the classic partition follows the device-side lengths.
  const bool uniform_hint = a->total_kv_len > 0 && bal_mode != 2 &&
                            (int64_t)a->total_kv_len == (int64_t)a->batch_size * a->max_kv_len;
  if (bal_mode != 0 && !(uniform_hint && n_splits == 1) &&
      ((a->n_tokens >= 16 && bal_pays) || bal_mode == 2) && a->max_q_len <= 1 &&
      a->n_tokens == a->batch_size &&
      a->sliding_window < 0 && forced_splits <= 0 && !decode_on_tile(a) && a->n_tokens > 0) {
    int slots = n_splits + 1 > 9 ? n_splits + 1 : 9;
    if (slots > COMBINE_MAX_SPLITS) slots = COMBINE_MAX_SPLITS;
    pl->bal = 1;
    pl->part_slots = slots;
    const int64_t qmin = (a->max_kv_len + slots - 2) / (slots - 1);
    pl->bal_qmin = (int)(qmin > 1 ? qmin : 1);
    if (pl->n_splits >= slots) pl->n_splits = slots - 1;  // P = n_tokens * n_splits workgroups per head group
  }
  pl->u = tune_get(TUNE_ATTN_U, pl->w == 2 ? 2 : 4);  // (two chunks: the same bytes in flight with half the rows)
  if (pl->u != 2 && pl->u != 4) pl->u = 4;
  pl->nt = tune_get(TUNE_ATTN_NT, a->max_q_len <= 1 ? 1 : 0) != 0;
  return SLM_OK;
}

template <typename T, int LPR, int GC>
static void launch_token_kernel(const AttnKParams& kp, const AttnPlan& pl, int64_t grid,
                                hipStream_t st) {
  const dim3 g((unsigned)grid), blk(pl.nw * 64);
#define SLM_LAUNCH(UU, NTT, SCC)                                                                \
  hipLaunchKernelGGL((attn_token_kernel<T, LPR, GC, UU, NTT, SCC>), g, blk, pl.lds_bytes, st, kp)
  if constexpr (LPR == 8 && GC <= 4) {
    if (pl.w == 2) {  // (plan_attn: plain softmax, head_dim 128, <= 4 query heads per lane)
#define SLM_LAUNCH_W2(UU, NTT)                                                                             \
  do {                                                                                                     \
    auto kfn = attn_token_kernel<T, LPR, GC, UU, NTT, false, 2>;                                           \
    static bool opted[64] = {}; /* > 64 KiB of dynamic LDS: opted into once per kernel AND per device */   \
    int devi = 0;                                                                                          \
    (void)hipGetDevice(&devi);                                                                             \
    if (devi < 0 || devi >= 64 || !opted[devi]) {                                                          \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                128 * 1024);                                                               \
      if (devi >= 0 && devi < 64) opted[devi] = true;                                                      \
    }                                                                                                      \
    hipLaunchKernelGGL(kfn, g, blk, pl.lds_bytes, st, kp);                                                 \
  } while (0)
      if (pl.u == 2) { if (pl.nt) SLM_LAUNCH_W2(2, true); else SLM_LAUNCH_W2(2, false); }
      else { if (pl.nt) SLM_LAUNCH_W2(4, true); else SLM_LAUNCH_W2(4, false); }
#undef SLM_LAUNCH_W2
      return;
    }
  }
  if (kp.softcap > 0.f || kp.alibi != nullptr) {
    SLM_LAUNCH(2, false, true);  // soft-cap / alibi models: one tuned shape
  } else if (pl.u == 2) {
    if (pl.nt) SLM_LAUNCH(2, true, false); else SLM_LAUNCH(2, false, false);
  } else {
    if (pl.nt) SLM_LAUNCH(4, true, false); else SLM_LAUNCH(4, false, false);
  }
#undef SLM_LAUNCH
}

template <typename T, int LPR>
static void dispatch_gc(const AttnKParams& kp, const AttnPlan& pl, int64_t grid, hipStream_t st) {
  switch (pl.gc) {
    case 8: launch_token_kernel<T, LPR, 8>(kp, pl, grid, st); break;
    case 4: launch_token_kernel<T, LPR, 4>(kp, pl, grid, st); break;
    case 2: launch_token_kernel<T, LPR, 2>(kp, pl, grid, st); break;
    default: launch_token_kernel<T, LPR, 1>(kp, pl, grid, st); break;
  }
}

template <typename T>
static void dispatch_lpr(const AttnKParams& kp, const AttnPlan& pl, int64_t grid, hipStream_t st) {
#ifdef SLM_ATTN_DEV_ONLY  // ISA-inspection builds: flagship instantiation only
  launch_token_kernel<bf16_tag, 16, 4>(kp, pl, grid, st);
  return;
#else
  switch (pl.lpr) {
    case 4: dispatch_gc<T, 4>(kp, pl, grid, st); break;
    case 8: dispatch_gc<T, 8>(kp, pl, grid, st); break;
    case 16: dispatch_gc<T, 16>(kp, pl, grid, st); break;
    default: dispatch_gc<T, 32>(kp, pl, grid, st); break;
  }
#endif
}

}  // namespace slm

using namespace slm;

extern "C" {

SLM_API int32_t slm_paged_kv_varlen_mha_auto_splits(const slm_attn_args* a) {
  AttnPlan pl;
  if (plan_attn(a, &pl) != SLM_OK) return 0;
  return pl.n_splits;
}

SLM_API int32_t slm_paged_kv_varlen_mha_decode_kernel(const slm_attn_args* a) {
  AttnPlan pl;
  if (plan_attn(a, &pl) != SLM_OK) return -1;
  return decode_on_tile(a) ? 1 : 0;
}

SLM_API size_t slm_paged_kv_varlen_mha_workspace_bytes(const slm_attn_args* a) {
  AttnPlan pl;
  if (plan_attn(a, &pl) != SLM_OK) return 0;
  if (pl.n_splits <= 1 && !pl.bal) return 0;
  return (size_t)a->n_tokens * a->n_heads * pl.part_slots * (a->head_dim + 2) * sizeof(float);
}

SLM_API int slm_paged_kv_varlen_mha(const slm_attn_args* a, void* stream) {
  AttnPlan pl;
  int rc = plan_attn(a, &pl);
  if (rc != SLM_OK) return rc;
  if (a->n_tokens == 0 || a->batch_size == 0) return SLM_OK;
  if (!a->out || !a->query || !a->key_cache || !a->value_cache || !a->q_cu_lens ||
      !a->kv_cu_lens || !a->block_table || !a->block_cu_lens)
    return SLM_ERR_INVALID_ARG;
  if (a->n_tokens < 0 || a->batch_size < 0) return SLM_ERR_INVALID_ARG;
  // 16-byte vector access on rows: base pointers and strides must keep 16-B alignment
  if (!aligned16(a->out) || !aligned16(a->query) || !aligned16(a->key_cache) ||
      !aligned16(a->value_cache))
    return SLM_ERR_ALIGNMENT;
  for (int i = 0; i < 2; ++i)
    if (a->o_stride[i] % 8 || a->q_stride[i] % 8 || a->k_stride[i] % 8 || a->v_stride[i] % 8)
      return SLM_ERR_ALIGNMENT;

  if (a->phase < 0 || a->phase > 2) return SLM_ERR_INVALID_ARG;
  const bool do_stream = a->phase != 2, do_combine = a->phase != 1;
  AttnKParams kp;
  kp.out = a->out; kp.q = a->query; kp.kc = a->key_cache; kp.vc = a->value_cache;
  kp.o_ts = a->o_stride[0]; kp.o_hs = a->o_stride[1];
  kp.q_ts = a->q_stride[0]; kp.q_hs = a->q_stride[1];
  kp.k_ss = a->k_stride[0]; kp.k_hs = a->k_stride[1];
  kp.v_ss = a->v_stride[0]; kp.v_hs = a->v_stride[1];
  kp.q_cu = a->q_cu_lens; kp.kv_cu = a->kv_cu_lens;
  kp.bt = a->block_table; kp.bcu = a->block_cu_lens;
  kp.alibi = a->alibi_slopes;
  kp.batch = a->batch_size; kp.n_tokens = a->n_tokens;
  kp.n_heads = a->n_heads; kp.n_kv_heads = a->n_kv_heads; kp.head_dim = a->head_dim;
  kp.block_shift = ilog2(a->block_size); kp.block_mask = a->block_size - 1;
  kp.group = a->n_heads / a->n_kv_heads;
  kp.n_chunks = pl.n_chunks; kp.hpw_shift = pl.hpw_shift; kp.hgw_shift = pl.hgw_shift;
  kp.nhgb = pl.nhgb; kp.n_splits = pl.n_splits; kp.window = a->sliding_window;
  if (a->logits_soft_cap > 0.f) {
    // softmax(tanh(x*sm_scale/cap)*cap): mha_params.h:56-67
    kp.softcap = a->logits_soft_cap;
    kp.pre_scale = a->sm_scale / a->logits_soft_cap;
    kp.scale_log2 = a->logits_soft_cap * LOG2E;
  } else {
    kp.softcap = 0.f; kp.pre_scale = 0.f; kp.scale_log2 = a->sm_scale * LOG2E;
  }
  kp.o_part = nullptr; kp.ml_part = nullptr;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // Many query rows per KV head (prefill, chunked prefill, speculative verify): MFMA tile kernel.
  // max_q_len is the same scheduling contract as the reference's grid (tile_scheduler.cuh:23-27).
  // A forced split count keeps the call on the token-major kernel.
  // Sequence classes by query rows per KV head (rows = q_len * group), decided on the DEVICE from
  // q_cu_lens, so one call handles mixed prefill / verify / decode batches (BASELINE config 5):
  //   rows <= group (q_len = 1)   token-major stream kernel (attn_token_kernel)
  //   group < rows <= 32          MFMA tile kernel, one-wave tiles   (speculative verify)
  //   rows > 32                   MFMA tile kernel, 2/4-wave tiles   (prefill, chunked prefill)
  // max_q_len is the same scheduling contract as the reference's grid (tile_scheduler.cuh:23-27).
  // A forced split count or an unsupported head_dim keeps everything on the token-major kernel.
  kp.rows_lo = 0;
  kp.rows_hi = 0x7fffffff;
  kp.bal = pl.bal; kp.part_slots = pl.part_slots; kp.bal_qmin = pl.bal_qmin; kp.bal_align = ATTN_BAL_ALIGN;
  kp.prio = tune_get(TUNE_ATTN_PRIO, 1);  // measured on the two-lane bs 256 step: 25.08 -> 24.85 ms (2 x 2 runs)
  if (pl.n_splits > 1 || pl.bal) {
    const size_t need =
        (size_t)a->n_tokens * a->n_heads * pl.part_slots * (a->head_dim + 2) * sizeof(float);
    if (!a->workspace || a->workspace_bytes < need) return SLM_ERR_WORKSPACE;
    kp.o_part = reinterpret_cast<float*>(a->workspace);
    kp.ml_part = kp.o_part + (size_t)a->n_tokens * a->n_heads * pl.part_slots * a->head_dim;
  }
  bool tile_used = false;
  const bool dec_tile = decode_on_tile(a);
  const int first_tile_rows = dec_tile ? kp.group : kp.group + 1;  // smallest row count on the tile kernel
  // Every sequence brings exactly max_q_len tokens (the q lengths are bounded by max_q_len -- the grid contract
  // -- and add up to batch_size * max_q_len): ONE row class is populated, the launches of the others would only
  // start and exit (2.5 us each: the token kernel and the one-wave tile class in front of a pure prefill were
  // 8 % of a 65 us causal 1 x 2048 call).  Anything else (mixed batches) launches every class as before.
  const bool uniform_q = a->max_q_len > 0 && (int64_t)a->batch_size * a->max_q_len == (int64_t)a->n_tokens;
  bool token_rows_possible = true;
  if ((a->max_q_len > 1 || dec_tile) && a->num_splits <= 0 && attn_tile_supported(kp.head_dim) &&
      tune_get(TUNE_ATTN_TILE, 1) != 0) {
    hip_clear_error();
    const int64_t max_rows = (int64_t)a->max_q_len * kp.group;
    AttnKParams tk = kp;
    token_rows_possible = !(uniform_q && max_rows >= first_tile_rows);
    if (do_stream && first_tile_rows < 33 && !(uniform_q && max_rows > 32)) {  // (group >= 32: every multi-row sequence already has > 32 rows)
      tk.rows_lo = first_tile_rows;
      tk.rows_hi = 33;
      rc = launch_attn_tile(tk, a->dtype, max_rows < 32 ? max_rows : 32, a->max_kv_len, st);
      if (rc != SLM_OK) return rc;
    }
    if (do_stream && max_rows > 32) {
      // rows <= group (q_len = 1) stay with the token-major kernel also when group > 32 (MQA)
      tk.rows_lo = first_tile_rows > 33 ? first_tile_rows : 33;
      tk.rows_hi = 0x7fffffff;
      rc = launch_attn_tile(tk, a->dtype, max_rows, a->max_kv_len, st);
      if (rc != SLM_OK) return rc;
    }
    // the token-major kernel keeps the q_len = 1 sequences
    kp.rows_hi = first_tile_rows;
    tile_used = true;
  }
  hip_clear_error();
  const int64_t grid = (int64_t)a->n_tokens * pl.nhgb * pl.n_chunks * pl.n_splits;
  if (grid <= 0 || grid > 0x7fffffffLL) return SLM_ERR_INVALID_ARG;
  if (do_stream && !(tile_used && (dec_tile || !token_rows_possible))) {  // (every sequence has >= group rows: nothing left for it then)
    if (a->dtype == SLM_BF16)
      dispatch_lpr<bf16_tag>(kp, pl, grid, st);
    else
      dispatch_lpr<f16_tag>(kp, pl, grid, st);
    rc = hip_check_launch();
    if (rc != SLM_OK) return rc;
  }
  if (do_combine && (pl.n_splits > 1 || pl.bal)) {
    const int64_t items = (int64_t)a->n_tokens * a->n_heads;
    const dim3 g((unsigned)((items + 3) / 4)), blk(256);
    int lps_shift = 3;  // lanes per split: power of two >= head_dim / 4
    while ((4 << lps_shift) < a->head_dim) ++lps_shift;
    AttnKParams ck = kp;
    if (tile_used) {  // every kernel of the call wrote partials: combine every row
      ck.rows_lo = 0;
      ck.rows_hi = 0x7fffffff;
    }
    if (a->dtype == SLM_BF16)
      hipLaunchKernelGGL(attn_combine_kernel<bf16_tag>, g, blk, 0, st, ck, lps_shift);
    else
      hipLaunchKernelGGL(attn_combine_kernel<f16_tag>, g, blk, 0, st, ck, lps_shift);
    rc = hip_check_launch();
  }
  return rc;
}

}  // extern "C"
