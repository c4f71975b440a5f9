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
//     also own the HBM weight stream: one 16-B load per lane per 64-deep chunk into an 8-deep
//     register ring (8 KiB per wave in flight), handed to the producers through a small raw-weight
//     LDS ring -- the consumers issue no other VMEM, so their vmcnt waits stay counted and deep;
//   producer waves 4-7: stage the activations by LDS-DMA (global_load_lds_dwordx4, no VGPR round
//     trip: the VGPR -> LDS store path, ~80 B/clk per CU, was the measured bottleneck of the first
//     version) and dequantise the raw weights into ready-made MFMA B fragments in LDS.  Their only
//     VMEM is the DMA, so "wait for my DMA" is an exact counted vmcnt and never waits behind an
//     HBM weight load.
//
// Pipeline unit = one STEP = 32 of K = 2 MFMA k-steps = 16 MFMAs per consumer wave (512 cycles),
// one bare s_barrier per step.  LDS (156 KiB, one workgroup per CU):
//   A ring  7 units x 16 KiB (256 rows x 64 B, 16-B slots XOR-swizzled by (row >> 2) & 3)
//   B ring  4 units x  8 KiB ([n-tile 4][k-step 2][lane][16 B] ready-made MFMA fragments)
//   raw ring 2 chunks x 4 x (1 KiB packed weights + 512 B scale words)
// In step g the consumers compute unit g (and prefetch the first fragments of unit g+1, published
// one barrier earlier); the producers issue the DMA for A unit g+6 (its slot held unit g-1), write
// B unit g+3 and publish (counted waits) A unit g+2 and B unit g+2.  A global -> LDS transfer therefore has 4 steps
// (~2000 cycles) to land: with one 64-deep chunk of lead (the previous version) the measured
// issue -> landed time of a 32 KiB chunk (~1100 cycles from L2, more from HBM) was fully exposed.
#include "w4_common.h"

namespace slm {

constexpr int WS_A_UNIT = 256 * 64;        // 256 rows x 32 k x 2 B
constexpr int WS_A_UNITS = 7;
constexpr int WS_B_UNIT = 4 * 2 * 1024;    // [n-tile][k-step][lane][16 B]
constexpr int WS_B_UNITS = 4;
constexpr int WS_B_BASE = WS_A_UNITS * WS_A_UNIT;
constexpr int WS_RAW_TILE = 1024 + 512;    // per n-tile: 64 x 16 B weights, 2 x 64 x 4 B scale words
constexpr int WS_RAW_BYTES = 4 * WS_RAW_TILE;
constexpr int WS_RAW_BASE = WS_B_BASE + WS_B_UNITS * WS_B_UNIT;
constexpr int WS_WD = 8;                   // weight ring depth (64-deep chunks in flight per consumer wave)
constexpr int WS_AL = 6;                   // A units in flight: DMA for unit g + WS_AL issued in step g
static_assert(WS_RAW_BASE + 2 * WS_RAW_BYTES == W4_WS_LDS_BYTES, "LDS size");
static_assert(W4_WS_LDS_BYTES <= 160 * 1024, "LDS capacity");
static_assert(WS_AL <= WS_A_UNITS - 1, "the slot of unit g + WS_AL must have been released before step g");

// Bare barrier with an explicit wait: __syncthreads() would add s_waitcnt vmcnt(0) lgkmcnt(0) and
// drain the consumers' prefetched fragment reads / in-flight weight loads and the producers'
// in-flight DMA.  Every wave waits for exactly what it publishes (LDS operations of one wave
// complete in order; so do its VMEM operations).
#define WS_WAIT_BARRIER(waitstr) asm volatile("s_waitcnt " waitstr "\n\ts_barrier" ::: "memory")
#define WS_STR2(x) #x
#define WS_STR(x) WS_STR2(x)

// NGC: scale groups per 64-deep chunk (2 for group 32, else 1)
// WNT: non-temporal weight loads -- right when every weight is read ONCE per launch (one block of 256
// rows: the decode batch), wrong when several row blocks re-read the weights (prefill: cacheable lines
// are served from L2 / Infinity Cache the second time; round 4: M = 1024 layer 533 -> 508 us)
template <typename T, int NGC, bool WNT>
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

  // chunk range of this split in 64-deep chunks (the plan counts 128-deep units); a step is half a chunk
  const int c0 = 2 * ks * p.chunks_per_split;
  const int c1 = 2 * min(p.n_chunks, (ks + 1) * p.chunks_per_split);
  const int n = c1 - c0;                                        // chunks, >= 2
  const int n_steps = 2 * ((n + WS_WD - 1) / WS_WD * WS_WD);    // main-loop steps (both roles)
  const int last = c1 - 1;
  auto clampc = [&](int c) { return c < last ? c : last; };

  if (wave >= 4) {
    // =============================== producer ===============================
    const int pw = wave - 4;
    // A staging by LDS-DMA (1 KiB = 16 rows x 64 B per wave-instruction, lane l lands at LDS base
    // + 16*l): this wave owns rows 64*pw .. 64*pw+63 of every unit.  The XOR swizzle is applied on
    // the GLOBAL side: the lane that lands in physical slot ps of row r fetches logical slot
    // ps ^ ((r >> 2) & 3), so the consumers' ds_read_b128 stay conflict-free.
    const char* abase = reinterpret_cast<const char*>(p.a);
    uint32_t a_off[4];  // byte offsets from p.a (the host checks that A spans < 2 GiB)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (pw * 4 + i) * 16 + (lane >> 2);
      const int slot = (lane & 3) ^ ((row >> 2) & 3);
      const int64_t m = m0 + row;
      const int64_t mc = m < p.M ? m : p.M - 1;  // rows >= M: clamped loads, never stored
      a_off[i] = (uint32_t)(2 * (mc * p.lda + slot * 8));
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int u_last = 2 * c1 - 1;  // last valid 32-deep unit (absolute); later units re-fetch it
    // one 1-KiB piece (16 rows) of a unit; the four pieces of a step are spread between the dequant
    // VALU so that the wave is never parked behind a full VMEM queue with nothing else to issue
    auto a_dma_piece = [&](int unit_rel, int slot, int i) {
      int ua = 2 * c0 + unit_rel;
      ua = ua < u_last ? ua : u_last;
      const uint32_t dst = lds0 + slot * WS_A_UNIT + pw * 4096 + i * 1024;
      // SGPR base + 32-bit VGPR offset (half the address traffic of the 64-bit VGPR form)
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2"
                   :
                   : "v"(a_off[i]), "s"(dst), "s"(abase + (int64_t)ua * 64)
                   : "memory", "m0");
    };
    // Step g (from -6): DMA for A unit g+6, dequant + write of B unit g+3, publish A unit g+2 and
    // B unit g+2.  Nothing on the critical path waits for a latency: the B writes of a step are
    // published one barrier later (counted lgkmcnt), the raw chunk is read one step before it is
    // needed, the DMA wait is counted.  Every step issues the same operations (clamped / masked
    // at both ends) so that the counted waits are exact.
    int aslot = 0;  // ring slot of the unit issued in this step
    u32x4 wv = {0u, 0u, 0u, 0u}, wv_n = {0u, 0u, 0u, 0u};
    uint32_t szc[NGC], szc_n[NGC];
#pragma unroll
    for (int q = 0; q < NGC; ++q) szc[q] = szc_n[q] = 0u;
    auto step = [&](int g, int half) {
      const int bu = g + 3;    // B unit written in this step
      const int cc = bu >> 1;  // its chunk (relative)
      const uint32_t keep = (bu >= 0 && cc < n) ? 0xffffffffu : 0u;  // tail / head units: zero fragments
      char* bdst = smem + WS_B_BASE + (bu & (WS_B_UNITS - 1)) * WS_B_UNIT + ((pw * 2 * 64 + lane) << 4);
      const W4DqMagic<T> dq(szc[NGC == 2 ? half : 0]);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uint32_t word = half == 0 ? (j == 0 ? wv.x : wv.y) : (j == 0 ? wv.z : wv.w);
        uint32_t o[4];
        // DMA piece, 2 pairs of dequant, DMA piece, 2 pairs: VALU issues while the TA digests
        a_dma_piece(g + WS_AL, aslot, 2 * j);
        __builtin_amdgcn_sched_barrier(0);
        o[0] = dq.pair(word, 0);
        o[1] = dq.pair(word, 1);
        __builtin_amdgcn_sched_barrier(0);
        a_dma_piece(g + WS_AL, aslot, 2 * j + 1);
        __builtin_amdgcn_sched_barrier(0);
        o[2] = dq.pair(word, 2);
        o[3] = dq.pair(word, 3);
        const u32x4 packed = {o[0] & keep, o[1] & keep, o[2] & keep, o[3] & keep};
        *reinterpret_cast<u32x4*>(bdst + j * 1024) = packed;
      }
      aslot = aslot == WS_A_UNITS - 1 ? 0 : aslot + 1;
    };
    for (int g = -6; g < n_steps; g += 2) {
      // even step: second half of the current chunk; then fetch the next raw chunk (published at
      // least one barrier ago) for the following step
      step(g, 1);
      {
        const int cn = (g + 4) >> 1;  // chunk of B unit g+4
        const char* raw = smem + WS_RAW_BASE + (cn & 1) * WS_RAW_BYTES + pw * WS_RAW_TILE;
        wv_n = *reinterpret_cast<const u32x4*>(raw + lane * 16);
#pragma unroll
        for (int q = 0; q < NGC; ++q)
          szc_n[q] = *reinterpret_cast<const uint32_t*>(raw + 1024 + q * 256 + lane * 4);
      }
      // A unit g+2 landed (16 younger DMAs in flight); B unit g+2 (previous step) written: the
      // 1 + NGC raw reads and this step's 2 fragment writes may still be in flight
      if constexpr (NGC == 2) WS_WAIT_BARRIER("vmcnt(16) lgkmcnt(5)");
      else WS_WAIT_BARRIER("vmcnt(16) lgkmcnt(4)");
      // odd step: first half of the next chunk
      wv = wv_n;
#pragma unroll
      for (int q = 0; q < NGC; ++q) szc[q] = szc_n[q];
      step(g + 1, 0);
      WS_WAIT_BARRIER("vmcnt(16) lgkmcnt(2)");
    }
    return;
  }
  static_assert(4 * (WS_AL - 2) == 16, "update the producers' vmcnt wait");

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
    if constexpr (WNT) w = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp));
    else w = *reinterpret_cast<const u32x4*>(wp);
#pragma unroll
    for (int g = 0; g < NGC; ++g) {
      const int64_t grp = ((int64_t)c * 64 + g * (64 / NGC)) >> p.gs_shift;
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
  const int swz = (mrow >> 2) & 3;
  const int a_row_off = (mh * 128 + mrow) * 64;
  const int b_off = WS_B_BASE + ((nh * 2 * 2 * 64 + lane) << 4);
  frag_t afr[2][4], bfr[2][2];
  auto load_frags = [&](int aslot, int bslot, int kstep, frag_t (&af)[4], frag_t (&bf)[2]) {
    const char* abase_l = smem + aslot * WS_A_UNIT + a_row_off + (((2 * kstep + kh) ^ swz) << 4);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      af[i] = __builtin_bit_cast(frag_t, *reinterpret_cast<const u32x4*>(abase_l + i * 32 * 64));
    const char* bbase_l = smem + b_off + bslot * WS_B_UNIT + (kstep << 10);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      bf[j] = __builtin_bit_cast(frag_t, *reinterpret_cast<const u32x4*>(bbase_l + j * 2048));
  };

  // steps -6 .. -1: raw chunks 0, 1, 2 at the even steps (the producers read a raw chunk one step
  // before they dequantise it, three B units ahead of the consumers)
  raw_put(0, wreg[0], szreg[0]);
  WS_WAIT_BARRIER("lgkmcnt(0)");
  WS_WAIT_BARRIER("lgkmcnt(0)");
  raw_put(1, wreg[1], szreg[1]);
  WS_WAIT_BARRIER("lgkmcnt(0)");
  WS_WAIT_BARRIER("lgkmcnt(0)");
  raw_put(2, wreg[2], szreg[2]);
  WS_WAIT_BARRIER("lgkmcnt(0)");
  WS_WAIT_BARRIER("lgkmcnt(0)");  // units 0 and 1 are staged

  load_frags(0, 0, 0, afr[0], bfr[0]);
  int aslot = 0, bslot = 0;
  for (int base = 0; base < n_steps; base += 2 * WS_WD) {
#pragma unroll
    for (int u = 0; u < 2 * WS_WD; ++u) {
      const int g = base + u;  // step: chunk g >> 1, half g & 1
      const int naslot = aslot == WS_A_UNITS - 1 ? 0 : aslot + 1;
      const int nbslot = (bslot + 1) & (WS_B_UNITS - 1);
      // k-step 0 (its fragments were prefetched), prefetch k-step 1
      load_frags(aslot, bslot, 1, afr[1], bfr[1]);
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[j][i4] = Mfma<T>::run(bfr[0][j], afr[0][i4], acc[j][i4]);  // C^T tile
#pragma unroll
      for (int q = 0; q < 6; ++q) {  // pin: the 6 LDS reads spread under the 8 MFMAs
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      // even steps: raw chunk (g >> 1) + 3 goes out between the two k-steps (its ds_write latency
      // hides under MFMAs; the producers read this slot's previous chunk two steps ago)
      if ((u & 1) == 0) {
        constexpr int dummy = 0; (void)dummy;
        raw_put((g >> 1) + 3, wreg[((u >> 1) + 3) % WS_WD], szreg[((u >> 1) + 3) % WS_WD]);
      }
      // k-step 1, prefetch k-step 0 of the next unit (published one barrier ago)
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
      // the raw-chunk ds_writes are older than the 6 prefetch reads: leaving those in flight still
      // guarantees the writes are done
      WS_WAIT_BARRIER("lgkmcnt(6)");
      aslot = naslot;
      bslot = nbslot;
    }
  }

  // ---- epilogue.  The MFMAs ran with the operands swapped (weights as the A operand), so every
  // accumulator tile is C^T: this lane holds token m = tile row base + (lane & 31) and the 16
  // columns n = 32 t + (r & 3) + 8 (r >> 2) + 4 (lane >> 5) -- four consecutive columns per
  // r >> 2, i.e. one 8-byte (bf16/fp16) or 16-byte (fp32 partial) store instead of four 2-byte ones.
  if (p.silu && p.split_k == 1) {
    store_ct_silu_pair<T>(p, acc, (int64_t)nb * 4 + nh * 2, m0 + mh * 128 + (lane & 31), lane);
    return;
  }
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

template <typename T, int NGC, bool WNT>
static void launch_ws(const GemmKParams& kp, int n_blocks, hipStream_t st) {
  auto kfn = w4a16_gemm_ws_kernel<T, NGC, WNT>;
  static bool opted[64] = {};  // > 64 KiB of dynamic LDS: opted into once per kernel AND per device
  int devi = 0;
  (void)hipGetDevice(&devi);
  if (devi < 0 || devi >= 64 || !opted[devi]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)W4_WS_LDS_BYTES);
    if (devi >= 0 && devi < 64) opted[devi] = true;
  }
  hipLaunchKernelGGL(kfn, dim3((unsigned)n_blocks), dim3(512), W4_WS_LDS_BYTES, st, kp);
}

void launch_gemm_ws(const GemmKParams& kp, int dtype, int ng, int n_blocks, hipStream_t st) {
  // ng = scale groups per 128 of K (w4.hip plan): 4 for group 32 -> 2 per 64-deep chunk
  const bool once = kp.n_mblocks <= 1;  // every weight read by one row block only: stream it past the caches
  if (dtype == SLM_BF16) {
    if (ng == 4) (once ? launch_ws<bf16_tag, 2, true>(kp, n_blocks, st) : launch_ws<bf16_tag, 2, false>(kp, n_blocks, st));
    else (once ? launch_ws<bf16_tag, 1, true>(kp, n_blocks, st) : launch_ws<bf16_tag, 1, false>(kp, n_blocks, st));
  } else {
    if (ng == 4) (once ? launch_ws<f16_tag, 2, true>(kp, n_blocks, st) : launch_ws<f16_tag, 2, false>(kp, n_blocks, st));
    else (once ? launch_ws<f16_tag, 1, true>(kp, n_blocks, st) : launch_ws<f16_tag, 1, false>(kp, n_blocks, st));
  }
}

}  // namespace slm
