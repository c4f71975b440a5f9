// w4_small.hip -- int4-weight x fp16/bf16-activation GEMM for M <= 32 (decode at small batch):
// a lean weight-streaming kernel.
//
// Same operator as w4.hip (replaces marlin::gptq_gemm, reference gptq_gemm.cu:585-710, on the
// small-batch decode shapes), same packed layout and scale/zero table (w4.hip header).
//
// At M <= 32 the GEMM is a pure HBM stream of the packed weights (58.7 MB for the Llama-3-8B
// gate_up layer = 8.4 us at 7 TB/s) and the cost that matters is INSTRUCTIONS PER WEIGHT WORD: a
// gfx950 SIMD issues about one instruction per 4 cycles in total, and the general kernel's small-M
// instantiation spent ~35 instructions per 8-weight word (measured, SQ_INSTS_*: 24 VALU + 7 SALU +
// LDS/waits), which alone is 16 us on this layer whatever the occupancy or split-K.  This kernel
// spends ~15:
//   * post-scaled form: the MFMA consumes the raw magic-number values (magic + q, exact in T);
//     unpack = shift + and_or per nibble pair.  The affine part is applied per scale group:
//         sum_k x_k s (q_k - z) = s * ( T - (magic + z) * X ),  T = sum_k x_k (magic + q_k),
//     X = sum_k x_k.  X comes out of the matrix pipe as well -- a second MFMA per k-step against a
//     constant all-ones B fragment, in exactly the accumulator layout the epilogue needs -- so no
//     VALU / DPP / LDS work is spent on activation sums (the matrix pipe is idle anyway).
//   * activations: two 16-B loads per thread per 128-deep chunk, two chunks ahead, through a
//     double-buffered XOR-swizzled LDS tile (conflict-free ds_read_b128 fragments).  Deliberately
//     NOT LDS-DMA here: hipcc's s_waitcnt insertion cannot count DMA issued from inline asm, and an
//     over-estimated vmcnt wait on the weight ring is exactly what bounded the previous kernels
//     (weights effectively prefetched ~1 chunk ahead = an exposed HBM round trip per chunk, the
//     same 25-30 us on gate_up whatever the split-K or the instruction count).  With every VMEM
//     operation visible to the compiler all waits are exact counted vmcnt.
//   * weights: 16-B loads, 4-chunk register ring, refilled right after use (8 KiB per wave in
//     flight; HBM latency x 7 TB/s needs ~14 MB in flight chip-wide).
// Numerics are those of the general kernel's post-scaled path (fp32 accumulate of exact products,
// affine correction in fp32): within the GEMM tolerance of the reference tests
// (marlin_gemm_test.py:104-107), not bit-identical to "dequantise to T, then multiply".
#include "w4_common.h"

namespace slm {

constexpr int SM_STAGES = 2;          // A tile buffers
constexpr int SM_STAGE_BYTES = 32 * 256;
constexpr int SM_RING = 4;            // weight ring (chunks)

template <typename T>
struct SmOnes;
template <>
struct SmOnes<bf16_tag> {
  static constexpr uint32_t bits = 0x3F803F80u;
};
template <>
struct SmOnes<f16_tag> {
  static constexpr uint32_t bits = 0x3C003C00u;
};

// NG: scale groups per 128-deep chunk (1 for group >= 128, 2 for 64, 4 for 32)
// SPAN: scale groups wider than a chunk (group 256.., per-channel): group boundaries are tested at
//       run time; the common group sizes keep every accumulate/epilogue decision static
template <typename T, int NG, bool SPAN>
__global__ void __launch_bounds__(256, 2) w4a16_gemm_small_kernel(const GemmKParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename Mfma<T>::frag frag_t;
  constexpr int WPG = 8 / NG;  // k-steps (words) per scale group within a chunk

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  const int nb = bid % p.n_nblocks;
  bid /= p.n_nblocks;
  const int mb = bid % p.n_mblocks;
  const int ks = bid / p.n_mblocks;
  const int64_t m0 = (int64_t)mb * 32;
  const int64_t n_tiles = p.N / 32;
  int64_t nt = (int64_t)nb * 4 + wave;
  const bool nvalid = nt < n_tiles;
  if (!nvalid) nt = n_tiles - 1;  // clamped duplicate work, never stored

  const int c0 = ks * p.chunks_per_split;
  const int c1 = min(p.n_chunks, c0 + p.chunks_per_split);
  const int nC = c1 - c0;  // >= 1
  const int last = c1 - 1;
  auto clampc = [&](int c) { return c < last ? c : last; };

  // ---- A staging: thread -> (row, 16-B slot) x 2 per chunk; global loads, swizzled LDS writes ----
  const char* abase = reinterpret_cast<const char*>(p.a);
  const char* a_src[2];
  int a_dst[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + 256 * i;
    const int row = idx >> 4, slot = idx & 15;
    const int64_t m = m0 + row;
    const int64_t mc = m < p.M ? m : p.M - 1;  // rows >= M: clamped loads, never stored
    a_src[i] = abase + 2 * (mc * p.lda + slot * 8);
    a_dst[i] = row * 256 + ((slot ^ (row & 15)) << 4);
  }
  // chunk c lives in areg[c % SM_RING] from its load (iteration c-4) to its LDS store (iteration
  // c-1).  The long residence is deliberate: VMEM completes in order, so waiting for an A load
  // also waits for every weight load issued before it -- an A load only one iteration old would
  // cap the weight ring at two chunks in flight; a three-iterations-old one costs nothing.
  u32x4 areg[SM_RING][2];
  auto a_load = [&](int c, u32x4 (&dst)[2]) {
    const uint32_t off = (uint32_t)clampc(c) * 256u;  // < 2 GiB: checked on the host
#pragma unroll
    for (int i = 0; i < 2; ++i) dst[i] = *reinterpret_cast<const u32x4*>(a_src[i] + off);
  };
  auto a_store = [&](int stage, const u32x4 (&src)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      *reinterpret_cast<u32x4*>(smem + stage * SM_STAGE_BYTES + a_dst[i]) = src[i];
  };

  // ---- weight / scale rings ----
  u32x4 wreg[SM_RING][2];
  uint32_t szreg[SM_RING][NG];
  // per-lane bases once; per load only a wave-uniform 32-bit byte offset is added (the host checks
  // that the packed weights and the scale table are < 4 GiB): scalar address math is issue slots too
  const char* wlane = reinterpret_cast<const char*>(p.wq + (nt * 64 + lane) * 4);
  const char* szlane = reinterpret_cast<const char*>(p.sz + nt * 32 + (lane & 31));
  const uint32_t wstride = (uint32_t)(n_tiles * 1024);  // bytes per 64-deep half chunk
  const uint32_t szstride = (uint32_t)(p.N * 4);        // bytes per scale group
  const int cpg_shift = p.gs_shift >= 30 ? 30 : (p.gs_shift > 7 ? p.gs_shift - 7 : 0);  // log2(chunks per group)
  auto w_load = [&](int c, u32x4 (&w)[2], uint32_t (&sz)[NG]) {
    const uint32_t cc = (uint32_t)clampc(c);
#pragma unroll
    for (int h = 0; h < 2; ++h)
      w[h] = __builtin_nontemporal_load(
          reinterpret_cast<const u32x4*>(wlane + (cc * 2 + h) * wstride));
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const uint32_t grp = NG > 1 ? cc * NG + g : (cc >> cpg_shift);
      sz[g] = *reinterpret_cast<const uint32_t*>(szlane + grp * szstride);
    }
  };

  // prologue in the ORDER the steady-state iterations issue (iteration k: A for chunk k+4, then
  // the refill = weights for chunk k+4), so the compiler's counted waits hold from iteration 0
  a_load(c0, areg[0]);
  w_load(c0, wreg[0], szreg[0]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int d = 1; d < SM_RING; ++d) {
    a_load(c0 + d, areg[d]);
    __builtin_amdgcn_sched_barrier(0);
    w_load(c0 + d, wreg[d], szreg[d]);
    __builtin_amdgcn_sched_barrier(0);
  }
  a_store(0, areg[0]);

  f32x16 acc, tmp, tmpx;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = tmp[r] = tmpx[r] = 0.f;
  const u32x4 ones4 = {SmOnes<T>::bits, SmOnes<T>::bits, SmOnes<T>::bits, SmOnes<T>::bits};
  const frag_t ones = __builtin_bit_cast(frag_t, ones4);
  uint32_t magic_v = W4Magic<T>::bits;
  asm volatile("" : "+v"(magic_v));  // keep it in a VGPR (not re-materialised as a literal)
  uint32_t mask_s = 0x000F000Fu;
  asm volatile("" : "+s"(mask_s));   // ... and the nibble-pair mask in an SGPR
  const int mrow = lane & 31, kh = lane >> 5;
  const int a_row = mrow * 256;
  const int a_swz = mrow & 15;

  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  bool group_open = false;  // tmp / tmpx hold a partial group (groups wider than a chunk)
  int stage = 0;
  const int n_iter = (nC + SM_RING - 1) / SM_RING * SM_RING;
  for (int base = 0; base < n_iter; base += SM_RING) {
#pragma unroll
    for (int u = 0; u < SM_RING; ++u) {
      const int i = base + u;  // chunk (relative); ring slot u
      // A for chunk i+4 into the registers chunk i left (stored one iteration ago)
      a_load(c0 + i + SM_RING, areg[u]);
      __builtin_amdgcn_sched_barrier(0);
      if (i < nC) {
        const char* sbase = smem + stage * SM_STAGE_BYTES + a_row;
        // does the scale group that ends this chunk end HERE (groups >= 128 may span chunks)
        const int cabs = c0 + i;
        const bool grp_ends = !SPAN || i == nC - 1 || ((cabs + 1) >> cpg_shift) != (cabs >> cpg_shift);
        frag_t af = __builtin_bit_cast(
            frag_t, *reinterpret_cast<const u32x4*>(sbase + (((0 * 2 + kh) ^ a_swz) << 4)));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          frag_t af_n = af;
          if (j < 7)
            af_n = __builtin_bit_cast(
                frag_t, *reinterpret_cast<const u32x4*>(sbase + ((((j + 1) * 2 + kh) ^ a_swz) << 4)));
          const u32x4 wv = wreg[u][j >> 2];
          const uint32_t word = (j & 3) == 0 ? wv.x : (j & 3) == 1 ? wv.y : (j & 3) == 2 ? wv.z : wv.w;
          uint32_t o[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            // (x & mask) | magic in ONE VALU op, v_and_or_b32: VOP3 takes no literals on gfx9-family,
            // so the mask rides in an SGPR and the magic in a VGPR, both opaque to the optimiser (with
            // literals hipcc emits v_and + v_or).  A plain expression, NOT inline asm: hipcc inserts
            // no hazard wait states behind an asm statement, and an MFMA issued right behind an asm
            // v_and_or_b32 reads stale B operands (seen in round 2 with independent MFMA chains).
            const uint32_t x = q == 0 ? word : word >> (4 * q);
            o[q] = (x & mask_s) | magic_v;
          }
          const u32x4 packed = {o[0], o[1], o[2], o[3]};
          const frag_t bf = __builtin_bit_cast(frag_t, packed);
          const bool g_first = (j % WPG) == 0 && !(SPAN && group_open);
          if (g_first) {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            tmp = Mfma<T>::run(af, bf, z);
            tmpx = Mfma<T>::run(af, ones, z);
          } else {
            tmp = Mfma<T>::run(af, bf, tmp);
            tmpx = Mfma<T>::run(af, ones, tmpx);
          }
          const bool g_last = (j % WPG) == WPG - 1;
          if (g_last && (!SPAN || grp_ends)) {
            // acc += s * (tmp - (magic + z) * X) for this lane's column
            float sc, zm;
            W4Magic<T>::decode(szreg[u][j / WPG], sc, zm);
            const float nzs = -zm * sc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = fmaf(sc, tmp[r], fmaf(nzs, tmpx[r], acc[r]));
          }
          af = af_n;
        }
        if constexpr (SPAN) group_open = !grp_ends;
      }
      // refills AFTER the old values are consumed (pinned): each ring slot keeps its registers
      __builtin_amdgcn_sched_barrier(0);
      w_load(c0 + i + SM_RING, wreg[u], szreg[u]);
      __builtin_amdgcn_sched_barrier(0);
      // chunk i+1 (loaded three iterations ago; counted wait: three and a half iterations of
      // loads stay in flight) -> the buffer everybody finished reading one barrier ago
      a_store(stage ^ 1, areg[(u + 1) % SM_RING]);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      stage ^= 1;
    }
  }

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int64_t ncol = nt * 32 + (lane & 31);
  float bv = 0.f;
  if (p.split_k == 1 && p.bias) {
    const uint16_t braw = reinterpret_cast<const uint16_t*>(p.bias)[ncol];
    bv = lo_f32<T>((uint32_t)braw);
  }
  if (p.silu && p.split_k == 1) {
    // SLM_W4_SILU_MUL: waves (0, 1) and (2, 3) hold a (gate, up) tile pair.  The up wave hands its
    // T-rounded tile to the gate wave through the (now idle) A buffers; same lane, same r.
    uint16_t* ex = reinterpret_cast<uint16_t*>(smem) + (wave >> 1) * 1024;
    if (wave & 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ex[r * 64 + lane] = pack1<T>(acc[r] + bv);
    }
    __syncthreads();
    if ((wave & 1) || !nvalid) return;
    const int64_t ocol = (nt >> 1) * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = m0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const float g = lo_f32<T>((uint32_t)pack1<T>(acc[r] + bv));
      const float u = lo_f32<T>((uint32_t)ex[r * 64 + lane]);
      if (row < p.M) reinterpret_cast<uint16_t*>(p.c)[row * p.ldc + ocol] = pack1<T>(silu_mul1(g, u));
    }
    return;
  }
  if (!nvalid) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t row = m0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (row < p.M) {
      if (p.split_k == 1)
        reinterpret_cast<uint16_t*>(p.c)[row * p.ldc + ncol] = pack1<T>(acc[r] + bv);
      else
        p.part[((int64_t)ks * p.M + row) * p.N + ncol] = acc[r];
    }
  }
}

template <typename T, int NG, bool SPAN>
static void launch_small_t(const GemmKParams& kp, int n_blocks, hipStream_t st) {
  hipLaunchKernelGGL((w4a16_gemm_small_kernel<T, NG, SPAN>), dim3((unsigned)n_blocks), dim3(256),
                     SM_STAGES * SM_STAGE_BYTES, st, kp);
}

template <typename T>
static void launch_small_ng(const GemmKParams& kp, int ng, int n_blocks, hipStream_t st) {
  if (ng == 4) launch_small_t<T, 4, false>(kp, n_blocks, st);
  else if (ng == 2) launch_small_t<T, 2, false>(kp, n_blocks, st);
  else if (kp.gs_shift == 7) launch_small_t<T, 1, false>(kp, n_blocks, st);  // group 128
  else launch_small_t<T, 1, true>(kp, n_blocks, st);
}

void launch_gemm_small(const GemmKParams& kp, int dtype, int ng, int n_blocks, hipStream_t st) {
  if (dtype == SLM_BF16) launch_small_ng<bf16_tag>(kp, ng, n_blocks, st);
  else launch_small_ng<f16_tag>(kp, ng, n_blocks, st);
}

}  // namespace slm
