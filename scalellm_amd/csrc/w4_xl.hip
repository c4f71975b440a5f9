// w4_xl.hip -- int4-weight x fp16/bf16-activation GEMM, 256 x 256 output tiles (large M x N).
//
// Same operator as w4.hip (replaces marlin::gptq_gemm, reference gptq_gemm.cu:585-710), same packed
// layout (w4.hip header), same bit-faithful dequant (W4Dq<T>, marlin/numeric_conversion.h:19-62).
//
// Why 256 x 256: the measured global -> LDS rate of a CU (~23 B/clk) is what bounds
// the 256 x 128 wave-specialised kernel (36 B/clk of A + W needed to run the matrix pipe flat out);
// a 256 x 256 tile needs 20 B/clk.  256 accumulator tiles do not fit a producer/consumer split, so
// all eight waves are symmetric: wave (mh, nq) owns the 128 x 64 block (4 m-tiles x 2 n-tiles =
// 128 accumulator VGPRs), two waves per SIMD take turns on the matrix pipe, and every wave also
//   * dequantises ONE of the tile's 8 column tiles into ready-made MFMA B fragments in LDS
//     (2 words = 56 VALU per 32-deep step, under its own MFMAs),
//   * stages 32 of the 256 activation rows by LDS-DMA (2 x 1 KiB per step, XOR swizzle applied on
//     the global side, 7-unit ring, issued 5 steps ahead).
// A step = 32 of K = 16 MFMAs per wave = 1024 matrix-pipe cycles per SIMD; one bare s_barrier per
// step; every wave publishes with counted waits (its B-fragment writes are older than the 6
// prefetch reads of the next unit, so lgkmcnt(6) covers them; its DMA of unit g+2 is older than 6
// DMAs and 2 refills).  LDS: A 7 x 16 KiB + B 3 x 16 KiB = 160 KiB, one workgroup per CU.
#include "w4_common.h"

namespace slm {

constexpr int XL_A_UNIT = 256 * 64;         // 256 rows x 32 k x 2 B
constexpr int XL_A_UNITS = 7;
constexpr int XL_B_UNIT = 8 * 2 * 1024;     // [n-tile 8][k-step 2][lane][16 B]
constexpr int XL_B_UNITS = 3;
constexpr int XL_B_BASE = XL_A_UNITS * XL_A_UNIT;
constexpr int XL_RING = 4;                  // weight ring (64-deep chunks per wave)
constexpr int XL_AL = 5;                    // DMA for A unit g + XL_AL issued in step g
static_assert(XL_B_BASE + XL_B_UNITS * XL_B_UNIT == W4_XL_LDS_BYTES, "LDS size");
static_assert(W4_XL_LDS_BYTES <= 160 * 1024, "LDS capacity");
static_assert(XL_AL <= XL_A_UNITS - 2, "slot of unit g + XL_AL was released before step g");

// NGC: scale groups per 64-deep chunk (2 for group 32, else 1)
// WNT: non-temporal weight loads -- right when every weight is read ONCE per launch (one block of 256
// rows: the decode batch), wrong when several row blocks re-read the weights (prefill: cacheable lines
// are served from L2 / Infinity Cache the second time; round 4: M = 1024 layer 533 -> 508 us)
// One output tile (column block nb, row block mb) over the 64-deep chunks [c0, c1) of K: prologue + main loop, the
// sums left in `acc` (C^T accumulators).  MULTI: the caller runs several passes in one launch (the stream-K form
// below): the pass ends drained (no DMA / load in flight, every wave past its last LDS read).
template <typename T, int NGC, bool WNT, bool MULTI>
__device__ __forceinline__ void xl_pass(const GemmKParams& p, char* smem, const int nb, const int mb, const int c0,
                                        const int c1, f32x16 (&acc)[2][4]) {
  typedef typename Mfma<T>::frag frag_t;

  int tid_ = threadIdx.x;
  // (several passes per launch: keep hipcc from hoisting the per-lane address arithmetic out of the pass loop,
  // where it would stay live -- or spill -- across the whole body)
  if constexpr (MULTI) asm volatile("" : "+v"(tid_));
  const int tid = tid_;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t m0 = (int64_t)mb * 256;
  const int64_t n_tiles = p.N / 32;
  const int mh = wave >> 2, nq = wave & 3;
  const int mrow = lane & 31, kh = lane >> 5;
  (void)nq;

  const int n = c1 - c0;                                           // chunks, >= 2
  const int n_steps = 2 * ((n + XL_RING - 1) / XL_RING * XL_RING);  // main-loop steps
  const int last = c1 - 1;
  auto clampc = [&](int c) { return c < last ? c : last; };

  // ---- activations: LDS-DMA, this wave owns rows 32*wave .. 32*wave+31 of every unit ----
  const char* abase = reinterpret_cast<const char*>(p.a);
  uint32_t a_off[2];  // byte offsets from p.a (the host checks that A spans < 2 GiB)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave * 2 + i) * 16 + (lane >> 2);
    const int slot = (lane & 3) ^ ((row >> 2) & 3);
    const int64_t m = m0 + row;
    const int64_t mc = m < p.M ? m : p.M - 1;  // rows >= M: clamped loads, never stored
    a_off[i] = (uint32_t)(2 * (mc * p.lda + slot * 8));
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int u_last = 2 * c1 - 1;  // last valid 32-deep unit (absolute); later units re-fetch it
  auto a_dma_piece = [&](int unit_rel, int slot, int i) {
    int ua = 2 * c0 + unit_rel;
    ua = ua < u_last ? ua : u_last;
    const uint32_t dst = lds0 + slot * XL_A_UNIT + wave * 2048 + i * 1024;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2"
                 :
                 : "v"(a_off[i]), "s"(dst), "s"(abase + (int64_t)ua * 64)
                 : "memory", "m0");
  };

  // ---- weights: this wave dequantises column tile `wave` of the block's 8 ----
  int64_t nt = (int64_t)nb * 8 + wave;
  if (nt >= n_tiles) nt = n_tiles - 1;  // clamped duplicate work, never stored
  const uint32_t* wlane = p.wq + (nt * 64 + lane) * 4;
  const uint32_t* szlane = p.sz + nt * 32 + (lane & 31);
  const int64_t wstride = n_tiles * 256;  // u32 per 64-deep chunk
  u32x4 wreg[XL_RING];
  uint32_t szreg[XL_RING][NGC];
  auto w_load = [&](int c, u32x4& w, uint32_t (&sz)[NGC]) {
    const int cc = clampc(c);
    if constexpr (WNT) w = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wlane + (int64_t)cc * wstride));
    else w = *reinterpret_cast<const u32x4*>(wlane + (int64_t)cc * wstride);
#pragma unroll
    for (int g = 0; g < NGC; ++g) {
      const int64_t grp = ((int64_t)cc * 64 + g * (64 / NGC)) >> p.gs_shift;
      sz[g] = szlane[grp * p.N];
    }
  };
  // B unit `bu` (relative 32-deep unit) of this wave's column tile: dequant 2 words, 2 LDS writes
  auto b_produce = [&](int bu, int slot, const u32x4& wv, const uint32_t (&sz)[NGC]) {
    const int half = bu & 1;
    const uint32_t keep = ((bu >> 1) < n) ? 0xffffffffu : 0u;  // tail units: zero fragments
    char* bdst = smem + XL_B_BASE + slot * XL_B_UNIT + ((wave * 2 * 64 + lane) << 4);
    const W4Dq<T> dq(sz[NGC == 2 ? half : 0]);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t word = half == 0 ? (j == 0 ? wv.x : wv.y) : (j == 0 ? wv.z : wv.w);
      uint32_t o[4];
      dq.word(word, o);
      const u32x4 packed = {o[0] & keep, o[1] & keep, o[2] & keep, o[3] & keep};
      *reinterpret_cast<u32x4*>(bdst + j * 1024) = packed;
    }
  };

#pragma unroll
  for (int d = 0; d < XL_RING; ++d) {
    w_load(c0 + d, wreg[d], szreg[d]);
    __builtin_amdgcn_sched_barrier(0);
  }

#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

  // per-lane LDS offsets: A row (mh*4 + i)*32 + mrow, slot (2*kstep + kh) ^ swizzle;  B fragment
  const int swz = (mrow >> 2) & 3;
  const int a_row_off = (mh * 128 + mrow) * 64;
  const int b_off = XL_B_BASE + ((nq * 2 * 2 * 64 + lane) << 4);
  frag_t afr[2][4], bfr[2][2];
  auto load_frags = [&](int aslot, int bslot, int kstep, frag_t (&af)[4], frag_t (&bf)[2]) {
    const char* abase_l = smem + aslot * XL_A_UNIT + a_row_off + (((2 * kstep + kh) ^ swz) << 4);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      af[i] = __builtin_bit_cast(frag_t, *reinterpret_cast<const u32x4*>(abase_l + i * 32 * 64));
    const char* bbase_l = smem + b_off + bslot * XL_B_UNIT + (kstep << 10);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      bf[j] = __builtin_bit_cast(frag_t, *reinterpret_cast<const u32x4*>(bbase_l + j * 2048));
  };

  // ---- prologue: A units 0..4 in flight, B units 0 and 1 written, everything landed ----
#pragma unroll
  for (int u = 0; u < XL_AL; ++u) {
    a_dma_piece(u, u, 0);
    a_dma_piece(u, u, 1);
  }
  b_produce(0, 0, wreg[0], szreg[0]);
  b_produce(1, 1, wreg[0], szreg[0]);
  __builtin_amdgcn_sched_barrier(0);
  w_load(c0 + XL_RING, wreg[0], szreg[0]);
  __builtin_amdgcn_sched_barrier(0);
  // units 0..2 must have landed (their 6 DMAs are older than 4 DMAs + the refill)
  if constexpr (NGC == 2) asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");

  load_frags(0, 0, 0, afr[0], bfr[0]);
  int aslot = 0, bslot = 0;           // slots of the unit being computed
  int dslot = XL_AL % XL_A_UNITS;     // slot of the unit whose DMA is issued in this step
  for (int base = 0; base < n_steps; base += 2 * XL_RING) {
#pragma unroll
    for (int u = 0; u < 2 * XL_RING; ++u) {
      const int g = base + u;  // step; produces B unit g+2 = chunk (g+2) >> 1 (ring slot static)
      const int naslot = aslot == XL_A_UNITS - 1 ? 0 : aslot + 1;
      const int nbslot = bslot == XL_B_UNITS - 1 ? 0 : bslot + 1;
      const int wslot = nbslot == XL_B_UNITS - 1 ? 0 : nbslot + 1;  // slot of B unit g+2
      const int rs = ((u + 2) >> 1) % XL_RING;  // ring slot of chunk (g+2) >> 1

      a_dma_piece(g + XL_AL, dslot, 0);
      __builtin_amdgcn_sched_barrier(0);
      // first half: MFMAs of k-step 0 (fragments prefetched) over the 6 fragment reads of k-step 1
      // and the dequant of B unit g+2
      load_frags(aslot, bslot, 1, afr[1], bfr[1]);
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[j][i4] = Mfma<T>::run(bfr[0][j], afr[0][i4], acc[j][i4]);  // C^T tile
      b_produce(g + 2, wslot, wreg[rs], szreg[rs]);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (q < 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      a_dma_piece(g + XL_AL, dslot, 1);
      if (u & 1) {  // second half of the chunk consumed: refill its ring slot 4 chunks ahead
        __builtin_amdgcn_sched_barrier(0);
        w_load(c0 + ((g + 2) >> 1) + XL_RING, wreg[rs], szreg[rs]);
      }
      __builtin_amdgcn_sched_barrier(0);
      // second half: MFMAs of k-step 1 over the 6 prefetch reads of the next unit's k-step 0
      // (issued AFTER this step's B writes: lgkmcnt(6) below then covers the writes)
      load_frags(naslot, nbslot, 0, afr[0], bfr[0]);
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[j][i4] = Mfma<T>::run(bfr[1][j], afr[1][i4], acc[j][i4]);
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      // publish: A unit g+2 landed (6 younger DMAs + 2 refills stay in flight), B unit g+2 written
      if constexpr (NGC == 2) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(6)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(10) lgkmcnt(6)\n\ts_barrier" ::: "memory");
      aslot = naslot;
      bslot = nbslot;
      dslot = dslot == XL_A_UNITS - 1 ? 0 : dslot + 1;
    }
  }
  if constexpr (MULTI) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---- epilogue (C^T accumulators: lane = token, 4 consecutive columns per r >> 2) ----
// final = true: T(acc + bias) into c;  false: fp32 into the split-K slab `ks` of p.part
template <typename T>
__device__ __forceinline__ void xl_store(const GemmKParams& p, const f32x16 (&acc)[2][4], const int nb, const int mb,
                                         const int ks, const bool final) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int64_t m0 = (int64_t)mb * 256;
  const int64_t n_tiles = p.N / 32;
  const int mh = wave >> 2, nq = wave & 3;
  if (p.silu && final) {
    store_ct_silu_pair<T>(p, acc, (int64_t)nb * 8 + nq * 2, m0 + mh * 128 + (lane & 31), lane);
    return;
  }
  const bool wide = ((p.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.c) & 7) == 0);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int64_t t = (int64_t)nb * 8 + nq * 2 + j;
    if (t >= n_tiles) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t ncol = t * 32 + 8 * q + 4 * (lane >> 5);
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (final && p.bias) {
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
        if (final) {
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

template <typename T, int NGC, bool WNT>
__global__ void __launch_bounds__(512, 2) w4a16_gemm_xl_kernel(const GemmKParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int bid = blockIdx.x;
  const int nb = bid % p.n_nblocks;
  bid /= p.n_nblocks;
  const int mb = bid % p.n_mblocks;
  const int ks = bid / p.n_mblocks;
  // chunk range of this split in 64-deep chunks (the plan counts 128-deep units); a step is half a chunk
  const int c0 = 2 * ks * p.chunks_per_split;
  const int c1 = 2 * min(p.n_chunks, (ks + 1) * p.chunks_per_split);
  f32x16 acc[2][4];
  xl_pass<T, NGC, WNT, false>(p, smem, nb, mb, c0, c1, acc);
  xl_store<T>(p, acc, nb, mb, ks, p.split_k == 1);
}

// ---- stream-K form (round 6) ---------------------------------------------------------------------------------
// Tiles that do not fill whole rounds of the 256 CUs (M = 2648, N = 4096: 176 tiles = one 69 %-full round) leave
// the rest of the chip idle for a whole tile time.  Here the WORK -- tiles x 128-deep chunks of K, tile-major --
// is cut into 256 equal ranges, one per workgroup (the reference's own answer: Marlin's striped partition,
// gemm_kernel.cuh:66-150).  A range is split at tile boundaries into pieces; the piece that reaches its tile's
// last chunk OWNS the tile, an earlier piece is a PARTIAL: its fp32 sums go to the workgroup's slot (write-through
// stores, fragment-major) and a flag.  A workgroup runs its partial piece FIRST (at most one: the head of its last
// tile), then its owner pieces; the owner adds the partial slots of the workgroups in front of it -- which
// wrote them first thing -- in workgroup order and stores the 16-bit tile: a fixed summation order, the same bits
// launch after launch.  Workgroup indices are TICKETS drawn at start, so "the workgroups in front" are running or
// done whatever the dispatch order; the waits are bounded all the same (~2 s, then a trap).  Hand-off as the guide's R1 recipe:
// sc1 payload stores, every wave drains, barrier, one relaxed agent-scope flag store; the owner polls relaxed and
// reads the slots past its L2 (sc1 loads).
// Measured (profiles/r06_gemm_streamk.jsonl, M = 2648): qkv 186 -> 155 us, o 110 -> 105, down 351 -> 300; a range costs
// ~1.4x its share of a tile's time (a second pipeline fill per workgroup, the partial tile's round trip -- ~9 us of
// o's 105 --, and 256 busy CUs clock lower than 176), which is what the plan's model charges.  Giving each XCD
// a run of consecutive tiles (tickets 8 j + x as group x, ranges inside the group) changed nothing (104.8 / 155.6 /
// 300.2 us): the activations' L2 reuse is not what the ranges lose.
__device__ __forceinline__ uint32_t xl_sk_frag_off(int slot, int wave, int f, int lane) {
  return (uint32_t)slot * (uint32_t)W4_XL_SK_SLOT_BYTES + (uint32_t)(((wave * 32 + f) * 64 + lane) << 4);
}

template <typename T, int NGC, bool WNT>
__global__ void __launch_bounds__(512, 2) w4a16_gemm_xl_sk_kernel(const GemmKParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // (the ticket is broadcast through the first word of the tile's own LDS -- all 160 KiB belong to the passes --
  // before any pass touches it)
  int* bc = reinterpret_cast<int*>(smem);
  if (threadIdx.x == 0)
    bc[0] = (int)__hip_atomic_fetch_add(p.sk_sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int g = __builtin_amdgcn_readfirstlane(bc[0]);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int nc = p.n_chunks;
  const int per = p.sk_per;
  const int64_t W = (int64_t)p.n_mblocks * p.n_nblocks * nc;     // the work list, tile-major, in 128-deep chunks
  const int64_t w0 = (int64_t)g * per;
  if (w0 >= W) return;
  const int64_t w1 = w0 + per < W ? w0 + per : W;
  const int t_first = (int)(w0 / nc), t_last = (int)((w1 - 1) / nc);
  const bool has_partial = w1 < (int64_t)(t_last + 1) * nc;
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(p.sk_part, 0, (int)(W4_XL_SK_WGS * W4_XL_SK_SLOT_BYTES), 0x00020000);
  unsigned* flags = p.sk_sync + 1;
  f32x16 acc[2][4];
  if (has_partial) {
    const int64_t tb = (int64_t)t_last * nc;
    const int k0 = (int)((w0 > tb ? w0 : tb) - tb), k1 = (int)(w1 - tb);
    xl_pass<T, NGC, WNT, true>(p, smem, t_last % p.n_nblocks, t_last / p.n_nblocks, 2 * k0, 2 * k1, acc);
    if (!(p.ks_dbg & 1))
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = {acc[j][i][4 * q], acc[j][i][4 * q + 1], acc[j][i][4 * q + 2], acc[j][i][4 * q + 3]};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs,
                                                 (int)xl_sk_frag_off(g, wave, (j * 4 + i) * 4 + q, lane), 0, /*sc1*/ 16);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains ...
    __syncthreads();                                    // ... then ONE lane publishes
    if (threadIdx.x == 0) __hip_atomic_store(flags + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // owner pieces: whole tiles first, the tile that STARTS in the workgroups in front (k0 > 0: the tail of the
  // first tile) LAST -- its partials were the first thing those workgroups did, and this workgroup has done all
  // its other work by the time it asks for them (asked first, a two-chunk tail would wait for a 30-chunk head)
  const int t_own_last = has_partial ? t_last - 1 : t_last;
  const bool first_dep = w0 > (int64_t)t_first * nc;
  const int n_own = t_own_last - t_first + 1;
  for (int it = 0; it < n_own; ++it) {
    const int t = first_dep ? (it + 1 < n_own ? t_first + 1 + it : t_first) : t_first + it;
    const int64_t tb = (int64_t)t * nc;
    const int k0 = (int)((w0 > tb ? w0 : tb) - tb);
    xl_pass<T, NGC, WNT, true>(p, smem, t % p.n_nblocks, t / p.n_nblocks, 2 * k0, 2 * nc, acc);
    if (k0 > 0 && !(p.ks_dbg & 2)) {   // the partial slots of the workgroups in front, in workgroup order
      const int g_first = (int)(tb / per);
      for (int gp = g_first; gp < g; ++gp) {
        if (threadIdx.x == 0) {
          const uint64_t t0 = wall_clock64();
          while (__hip_atomic_load(flags + gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            __builtin_amdgcn_s_sleep(8);
            // ~2 s of the 100 MHz clock: a partial that never comes is a broken launch -- fail loudly (the
            // process aborts on the trap) rather than hang the GPU or store a tile that misses a piece
            if (wall_clock64() - t0 > 200000000ull) __builtin_trap();
          }
        }
        __syncthreads();
        // (16 loads = 8 KiB per wave in flight per trip: the reads go past the L2, ~2 us each way)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          u32x4 pv[16];
#pragma unroll
          for (int f = 0; f < 16; ++f)
            pv[f] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)xl_sk_frag_off(gp, wave, h * 16 + f, lane), 0, /*sc1*/ 16);
#pragma unroll
          for (int f = 0; f < 16; ++f) {
            const f32x4 v = __builtin_bit_cast(f32x4, pv[f]);
            const int i = f >> 2, q = f & 3;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[h][i][4 * q + e] += v[e];
          }
        }
      }
    }
    xl_store<T>(p, acc, t % p.n_nblocks, t / p.n_nblocks, 0, true);
  }
}

template <typename T, int NGC, bool WNT>
static void launch_xl_sk(const GemmKParams& kp, int n_wgs, hipStream_t st) {
  auto kfn = w4a16_gemm_xl_sk_kernel<T, NGC, WNT>;
  static bool opted[64] = {};  // > 64 KiB of dynamic LDS: opted into once per kernel AND per device
  int devi = 0;
  (void)hipGetDevice(&devi);
  if (devi < 0 || devi >= 64 || !opted[devi]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)W4_XL_LDS_BYTES);
    if (devi >= 0 && devi < 64) opted[devi] = true;
  }
  hipLaunchKernelGGL(kfn, dim3((unsigned)n_wgs), dim3(512), W4_XL_LDS_BYTES, st, kp);
}

void launch_gemm_xl_sk(const GemmKParams& kp, int dtype, int ng, int n_wgs, hipStream_t st) {
  // (several row blocks always re-read the weights here: cacheable loads)
  if (dtype == SLM_BF16) {
    if (ng == 4) launch_xl_sk<bf16_tag, 2, false>(kp, n_wgs, st);
    else launch_xl_sk<bf16_tag, 1, false>(kp, n_wgs, st);
  } else {
    if (ng == 4) launch_xl_sk<f16_tag, 2, false>(kp, n_wgs, st);
    else launch_xl_sk<f16_tag, 1, false>(kp, n_wgs, st);
  }
}

template <typename T, int NGC, bool WNT>
static void launch_xl(const GemmKParams& kp, int n_blocks, hipStream_t st) {
  auto kfn = w4a16_gemm_xl_kernel<T, NGC, WNT>;
  static bool opted[64] = {};  // > 64 KiB of dynamic LDS: opted into once per kernel AND per device
  int devi = 0;
  (void)hipGetDevice(&devi);
  if (devi < 0 || devi >= 64 || !opted[devi]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)W4_XL_LDS_BYTES);
    if (devi >= 0 && devi < 64) opted[devi] = true;
  }
  hipLaunchKernelGGL(kfn, dim3((unsigned)n_blocks), dim3(512), W4_XL_LDS_BYTES, st, kp);
}

void launch_gemm_xl(const GemmKParams& kp, int dtype, int ng, int n_blocks, hipStream_t st) {
  // ng = scale groups per 128 of K (w4.hip plan): 4 for group 32 -> 2 per 64-deep chunk
  const bool once = kp.n_mblocks <= 1;  // every weight read by one row block only: stream it past the caches
  if (dtype == SLM_BF16) {
    if (ng == 4) (once ? launch_xl<bf16_tag, 2, true>(kp, n_blocks, st) : launch_xl<bf16_tag, 2, false>(kp, n_blocks, st));
    else (once ? launch_xl<bf16_tag, 1, true>(kp, n_blocks, st) : launch_xl<bf16_tag, 1, false>(kp, n_blocks, st));
  } else {
    if (ng == 4) (once ? launch_xl<f16_tag, 2, true>(kp, n_blocks, st) : launch_xl<f16_tag, 2, false>(kp, n_blocks, st));
    else (once ? launch_xl<f16_tag, 1, true>(kp, n_blocks, st) : launch_xl<f16_tag, 1, false>(kp, n_blocks, st));
  }
}

}  // namespace slm
