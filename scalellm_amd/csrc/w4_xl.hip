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
template <typename T, int NGC, bool WNT>
__global__ void __launch_bounds__(512, 2) w4a16_gemm_xl_kernel(const GemmKParams p) {
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
  const int mh = wave >> 2, nq = wave & 3;
  const int mrow = lane & 31, kh = lane >> 5;

  // chunk range of this split in 64-deep chunks (the plan counts 128-deep units); a step is half a chunk
  const int c0 = 2 * ks * p.chunks_per_split;
  const int c1 = 2 * min(p.n_chunks, (ks + 1) * p.chunks_per_split);
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

  f32x16 acc[2][4];
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

  // ---- epilogue (C^T accumulators: lane = token, 4 consecutive columns per r >> 2) ----
  if (p.silu && p.split_k == 1) {
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

template <typename T, int NGC, bool WNT>
static void launch_xl(const GemmKParams& kp, int n_blocks, hipStream_t st) {
  auto kfn = w4a16_gemm_xl_kernel<T, NGC, WNT>;
  static bool opted = false;  // > 64 KiB of dynamic LDS has to be opted into once per kernel
  if (!opted) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)W4_XL_LDS_BYTES);
    opted = true;
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
