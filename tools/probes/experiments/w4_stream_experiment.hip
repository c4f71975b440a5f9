// w4_stream.hip -- int4-weight x fp16/bf16-activation GEMM for M <= 32, second generation
// ("stream" kernel): replaces marlin::gptq_gemm (reference gptq_gemm.cu:585-710, small-M tile
// configs gptq_gemm.cu:132-149) on the decode shapes at small batch.  Same packed layout, same
// scale / zero table, same post-scaled numerics as w4_small.hip (see its header): the MFMA
// consumes magic + q (exact in T), the affine part s * (T - (magic + z) * X) is applied per scale
// group in fp32.
//
// What the round-2 measurements said about w4_small.hip (profiles/r02_small_gemm_ablation.jsonl,
// tools/probes/): at M = 32 a wave needs ~1450 cycles per 128-deep chunk of its column tile and
// the memory side would deliver one in ~700 (pure streaming of the same layout: gate_up 10.4 us
// vs 22-24 us for the GEMM); compiling pieces out showed where the gap is:
//     the activation-sum MFMA (against an all-ones fragment)   -4.1 us
//     activation staging + the per-chunk workgroup barrier     -4.3 us
//     the nibble unpack (56 of the 164 instructions per chunk)  0.0 us
// i.e. dependency / synchronisation structure, not instruction count.  Hence:
//   * TWO column tiles per wave (NTW = 2, workgroup = 4 waves x 64 columns): one activation
//     fragment read, one staging pass and one barrier now serve twice the weight bytes, and the
//     two tiles' MFMA chains are independent (one tile's group epilogue runs under the other's
//     MFMAs).  Narrow layers keep NTW = 1 (more workgroups).
//   * the activation sums X[row][group] come from the STAGING registers (v_dot2 against ones +
//     a 16-lane DPP reduction while the tile is being written to LDS) and are shared through LDS
//     by all tiles of the workgroup -- the second MFMA per k-step is gone.
//   * everything else as before: weights HBM -> registers (nt loads, one KiB per wave instruction,
//     4-chunk ring refilled right after use, exact counted vmcnt waits because every VMEM
//     operation is visible to the compiler), activations through a double-buffered XOR-swizzled
//     LDS tile loaded 4 chunks ahead (VMEM completes in order: a younger activation load would
//     cap the weight ring).
#include "w4_common.h"

namespace slm {

constexpr int ST_RING = 4;                 // activation stage = 4 chunks
constexpr int ST_WRING = 8;                // weight ring depth (chunks): two stages
constexpr int ST_TILE_BYTES = 32 * 256;    // one 128-deep activation chunk: 32 rows x 256 B

template <typename T>
struct StOnes;
template <>
struct StOnes<bf16_tag> { static constexpr uint32_t bits = 0x3F803F80u; };
template <>
struct StOnes<f16_tag> { static constexpr uint32_t bits = 0x3C003C00u; };

// NG: scale groups per 128-deep chunk (1 for group >= 128, 2 for 64, 4 for 32)
// SPAN: scale groups wider than a chunk (256.., per-channel): group ends are tested at run time
// NTW: 32-column tiles per wave
template <typename T, int NG, bool SPAN, int NTW>
__global__ void __launch_bounds__(256, 2) w4a16_gemm_stream_kernel(const GemmKParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename Mfma<T>::frag frag_t;
  constexpr int WPG = 8 / NG;  // k-steps (words) per scale group within a chunk
  float* xs_base = reinterpret_cast<float*>(smem + 2 * ST_RING * ST_TILE_BYTES);  // [2 buffers][ST_RING chunks][NG][32 rows]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  const int nb = bid % p.n_nblocks;
  const int ks = bid / p.n_nblocks;  // one M block (M <= 32)
  const int64_t n_tiles = p.N / 32;
  int64_t nt[NTW];
  bool nvalid[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    // tile t of wave w: the workgroup's 4 * NTW tiles are dealt tile-major, so that at one k
    // position the four waves read 4 adjacent KiB per t
    const int64_t g = (int64_t)nb * 4 * NTW + t * 4 + wave;
    nvalid[t] = g < n_tiles;
    nt[t] = nvalid[t] ? g : n_tiles - 1;  // clamped duplicate work, never stored
  }

  const int c0 = ks * p.chunks_per_split;
  const int c1 = min(p.n_chunks, c0 + p.chunks_per_split);
  const int nC = c1 - c0;  // >= 1
  const int last = c1 - 1;
  auto clampc = [&](int c) { return c < last ? c : last; };

  // ---- A staging: thread -> (row, 16-B slot) x 2 per chunk ----
  const char* const a_u = reinterpret_cast<const char*>(p.a);
  uint32_t a_off[2];
  int a_dst[2], x_dst[2];
  bool x_wr[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + 256 * i;
    const int row = idx >> 4, slot = idx & 15;
    const int64_t mc = row < p.M ? row : p.M - 1;  // rows >= M: clamped loads, never stored
    a_off[i] = (uint32_t)(2 * (mc * p.lda + slot * 8));  // < 2 GiB: checked on the host
    a_dst[i] = row * 256 + ((slot ^ (row & 15)) << 4);
    x_dst[i] = (slot / (16 / NG)) * 32 + row;       // xs[group][row]
    x_wr[i] = (slot & (16 / NG - 1)) == 0;
  }
  // A of one STAGE = ST_RING chunks (32 rows x 512 k = 32 KiB), TWO stages ahead in registers:
  // the loads of stage s+2 are issued at the start of stage s and written to LDS at the end of
  // stage s+1.  The distance is the point: VMEM completes in order, so an activation load (an L2
  // hit) comes back only after every weight load issued BEFORE it -- including refills issued
  // moments earlier for chunks a whole weight ring ahead, which take a loaded-HBM latency.  With
  // the activations only ONE stage ahead every stage lasted at least that latency, whatever the
  // rest of the kernel did (measured: ~0.75 us per chunk in every variant, profiles/r02_*); two
  // stages ahead, the loads the wait falls behind have all been consumed anyway.
  u32x4 areg[2][ST_RING][2];
  auto a_load_stage = [&](int cfirst, u32x4 (&dst)[ST_RING][2]) {
#pragma unroll
    for (int ch = 0; ch < ST_RING; ++ch) {
      const char* ab = a_u + (size_t)clampc(cfirst + ch) * 256u;  // uniform
#pragma unroll
      for (int i = 0; i < 2; ++i) dst[ch][i] = *reinterpret_cast<const u32x4*>(ab + a_off[i]);
    }
  };
  uint32_t ones_v = StOnes<T>::bits;
  asm volatile("" : "+v"(ones_v));
  auto a_store_stage = [&](int buf, const u32x4 (&src)[ST_RING][2]) {
#pragma unroll
    for (int ch = 0; ch < ST_RING; ++ch) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const u32x4 v = src[ch][i];
        *reinterpret_cast<u32x4*>(smem + (buf * ST_RING + ch) * ST_TILE_BYTES + a_dst[i]) = v;
        // X[row][group] = sum of the group's activations: 8 elements here (dot2 against ones: the
        // products are exact, fp32 accumulate), then the 16 / NG lanes holding the group's slots
        float s8 = dot2<T>(v.x, ones_v, 0.f);
        s8 = dot2<T>(v.y, ones_v, s8);
        s8 = dot2<T>(v.z, ones_v, s8);
        s8 = dot2<T>(v.w, ones_v, s8);
        s8 = group_sum<16 / NG>(s8);
        if (x_wr[i]) xs_base[(buf * ST_RING + ch) * (NG * 32) + x_dst[i]] = s8;
      }
    }
  };

  // ---- weight / scale rings ----
  u32x4 wreg[ST_WRING][NTW][2];
  uint32_t szreg[ST_WRING][NTW][NG];
  // addressing: UNIFORM 64-bit base (SALU) + per-lane 32-bit offset that never changes -> the
  // global_load saddr + voffset form, no per-load VALU address arithmetic (the host checks that
  // the packed weights and the scale table are < 4 GiB)
  const char* const wq_u = reinterpret_cast<const char*>(p.wq);
  const char* const sz_u = reinterpret_cast<const char*>(p.sz);
  uint32_t woff[NTW], szoff[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    woff[t] = (uint32_t)((nt[t] * 64 + lane) * 16);
    szoff[t] = (uint32_t)((nt[t] * 32 + (lane & 31)) * 4);
  }
  const uint32_t wstride = (uint32_t)(n_tiles * 1024);  // bytes per 64-deep half chunk
  const uint32_t szstride = (uint32_t)(p.N * 4);        // bytes per scale group
  const int cpg_shift = p.gs_shift >= 30 ? 30 : (p.gs_shift > 7 ? p.gs_shift - 7 : 0);  // log2(chunks per group)
  auto w_load = [&](int c, u32x4 (&w)[NTW][2], uint32_t (&sz)[NTW][NG]) {
    const uint32_t cc = (uint32_t)clampc(c);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const char* wb = wq_u + (size_t)(cc * 2 + h) * wstride;  // uniform
#pragma unroll
      for (int t = 0; t < NTW; ++t)
        w[t][h] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wb + woff[t]));
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const uint32_t grp = NG > 1 ? cc * NG + g : (cc >> cpg_shift);
      const char* sb = sz_u + (size_t)grp * szstride;  // uniform
#pragma unroll
      for (int t = 0; t < NTW; ++t) sz[t][g] = *reinterpret_cast<const uint32_t*>(sb + szoff[t]);
    }
  };

  // prologue: A of stages 0 and 1, then the weight ring, then A of stage 0 -> LDS
  a_load_stage(c0, areg[0]);
  a_load_stage(c0 + ST_RING, areg[1]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int d = 0; d < ST_WRING; ++d) {
    w_load(c0 + d, wreg[d], szreg[d]);
    __builtin_amdgcn_sched_barrier(0);
  }
  a_store_stage(0, areg[0]);

  // NCH independent MFMA accumulation chains per tile (k-step j feeds chain j % NCH): an MFMA
  // whose accumulator operand was written by the PREVIOUS MFMA must wait for that result to be
  // written back unless the two issue back to back -- and here 8-16 unpack instructions sit
  // between them.  Measured (profiles/r02_*): one chain costs ~1250 cycles per 128-deep chunk
  // whatever else the kernel does; two chains (dependent distance 2 MFMAs + their unpack) are enough
  // and cost one add per accumulator element in the group epilogue (four chains: three).
  constexpr int NCH = 2;
  f32x16 acc[NTW], tmp[NTW][NCH];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[t][r] = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) tmp[t][c][r] = 0.f;
    }
  float xacc[SPAN ? 16 : 1];  // X of a group wider than a chunk, summed over its chunks
  if constexpr (SPAN) {
#pragma unroll
    for (int r = 0; r < 16; ++r) xacc[r] = 0.f;
  }
  (void)xacc;
  uint32_t magic_v = W4Magic<T>::bits;
  asm volatile("" : "+v"(magic_v));  // keep it in a VGPR (not re-materialised as a literal)
  uint32_t mask_s = 0x000F000Fu;
  asm volatile("" : "+s"(mask_s));   // ... and the nibble-pair mask in an SGPR
  const int mrow = lane & 31, kh = lane >> 5;
  const int a_row = mrow * 256;
  const int a_swz = mrow & 15;

  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  bool group_open = false;  // tmp / xacc hold a partial group (groups wider than a chunk)
  const int n_stage = (nC + ST_RING - 1) / ST_RING;
  for (int stg2 = 0; stg2 < n_stage; stg2 += 2) {
#pragma unroll
   for (int buf = 0; buf < 2; ++buf) {  // two stages per trip: LDS buffer and ring slots are static
    const int stg = stg2 + buf;
    a_load_stage(c0 + (stg + 2) * ST_RING, areg[buf]);  // activations two stages ahead (clamped past the end)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u4 = 0; u4 < ST_RING; ++u4) {
      const int u = buf * ST_RING + u4;  // weight ring slot
      const int i = stg * ST_RING + u4;  // chunk (relative)
      if (i < nC) {
        const char* sbase = smem + (buf * ST_RING + u4) * ST_TILE_BYTES + a_row;
        const float* xsb = xs_base + (buf * ST_RING + u4) * (NG * 32) + 4 * kh;
        const int cabs = c0 + i;
        const bool grp_ends = !SPAN || i == nC - 1 || ((cabs + 1) >> cpg_shift) != (cabs >> cpg_shift);
        // ALL eight activation fragments of the chunk are requested up front (8 x ds_read_b128, one
        // base register + immediates), pinned ahead of the unpack / MFMA stream: left to itself hipcc
        // issues each read right in front of its MFMA and the wave eats the LDS latency eight times
        // per chunk (measured: the dominant stall of the M <= 32 kernels, profiles/r02_*)
        frag_t afr[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          afr[j] = __builtin_bit_cast(
              frag_t, *reinterpret_cast<const u32x4*>(sbase + (((j * 2 + kh) ^ a_swz) << 4)));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const frag_t af = afr[j];
          const bool g_first = (j % WPG) < NCH && !(SPAN && group_open);  // first k-step of its chain
          const bool g_last = (j % WPG) == WPG - 1;
#pragma unroll
          for (int t = 0; t < NTW; ++t) {
            const u32x4 wv = wreg[u][t][j >> 2];
            const uint32_t word = (j & 3) == 0 ? wv.x : (j & 3) == 1 ? wv.y : (j & 3) == 2 ? wv.z : wv.w;
            uint32_t o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              // (x & mask) | magic: ONE v_and_or_b32 -- as a plain expression on two opaque registers
              // (mask in an SGPR, magic in a VGPR; VOP3 takes no literals on gfx9-family), NOT inline
              // asm: hipcc adds no hazard wait states behind an asm statement, and with independent
              // MFMA chains the consuming MFMA issues right behind the unpack (wrong B operands).
              const uint32_t x = q == 0 ? word : word >> (4 * q);
              o[q] = (x & mask_s) | magic_v;
            }
            const u32x4 packed = {o[0], o[1], o[2], o[3]};
            const frag_t bf = __builtin_bit_cast(frag_t, packed);
            if (g_first) {
              f32x16 z;
#pragma unroll
              for (int r = 0; r < 16; ++r) z[r] = 0.f;
              tmp[t][j % NCH] = Mfma<T>::run(af, bf, z);
            } else {
              tmp[t][j % NCH] = Mfma<T>::run(af, bf, tmp[t][j % NCH]);
            }
          }
          if (g_last) {
            // this lane's 16 rows of X for the group: rows (r & 3) + 8 (r >> 2) + 4 kh
            float xr[16];
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
              const f32x4 xv = *reinterpret_cast<const f32x4*>(xsb + (j / WPG) * 32 + 8 * q4);
              xr[4 * q4 + 0] = xv.x; xr[4 * q4 + 1] = xv.y; xr[4 * q4 + 2] = xv.z; xr[4 * q4 + 3] = xv.w;
            }
            if constexpr (SPAN) {
#pragma unroll
              for (int r = 0; r < 16; ++r) xr[r] = (xacc[r] += xr[r]);
            }
            if (!SPAN || grp_ends) {
#pragma unroll
              for (int t = 0; t < NTW; ++t) {
                // acc += s * (tmp - (magic + z) * X) for this lane's column of tile t
                float sc, zm;
                W4Magic<T>::decode(szreg[u][t][j / WPG], sc, zm);
                const float nzs = -zm * sc;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                  float ts = tmp[t][0][r];
#pragma unroll
                  for (int c = 1; c < NCH; ++c) ts += tmp[t][c][r];
                  acc[t][r] = fmaf(sc, ts, fmaf(nzs, xr[r], acc[t][r]));
                }
              }
              if constexpr (SPAN) {
#pragma unroll
                for (int r = 0; r < 16; ++r) xacc[r] = 0.f;
              }
            }
          }
        }
        if constexpr (SPAN) group_open = !grp_ends;
      }
      // refills AFTER the old values are consumed (pinned): each ring slot keeps its registers
      __builtin_amdgcn_sched_barrier(0);
      w_load(c0 + i + ST_WRING, wreg[u], szreg[u]);
      __builtin_amdgcn_sched_barrier(0);
    }
    // next stage -> the buffer everybody finished reading one barrier ago
    a_store_stage(buf ^ 1, areg[buf ^ 1]);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
   }
  }

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    if (!nvalid[t]) continue;
    const int64_t ncol = nt[t] * 32 + (lane & 31);
    float bv = 0.f;
    if (p.split_k == 1 && p.bias) {
      const uint16_t braw = reinterpret_cast<const uint16_t*>(p.bias)[ncol];
      bv = lo_f32<T>((uint32_t)braw);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row < p.M) {
        if (p.split_k == 1)
          reinterpret_cast<uint16_t*>(p.c)[row * p.ldc + ncol] = pack1<T>(acc[t][r] + bv);
        else
          p.part[((int64_t)ks * p.M + row) * p.N + ncol] = acc[t][r];
      }
    }
  }
}

template <typename T, int NG, bool SPAN, int NTW>
static void launch_stream_k(const GemmKParams& kp, int n_blocks, hipStream_t st) {
  constexpr size_t lds = 2 * ST_RING * ST_TILE_BYTES + 2 * ST_RING * NG * 32 * sizeof(float);
  auto kfn = w4a16_gemm_stream_kernel<T, NG, SPAN, NTW>;
  static bool attr_set = false;  // 65-69 KiB of dynamic LDS: above the 64 KiB default
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kfn, dim3((unsigned)n_blocks), dim3(256), lds, st, kp);
}

template <typename T, int NG, bool SPAN>
static void launch_stream_t(const GemmKParams& kp, int ntw, int n_blocks, hipStream_t st) {
  if constexpr (NG == 1) {  // two tiles per wave only for one group per chunk (registers: 210 VGPRs)
    if (ntw == 2) {
      launch_stream_k<T, NG, SPAN, 2>(kp, n_blocks, st);
      return;
    }
  }
  launch_stream_k<T, NG, SPAN, 1>(kp, n_blocks, st);
}

template <typename T>
static void launch_stream_ng(const GemmKParams& kp, int ng, int ntw, int n_blocks, hipStream_t st) {
  if (ng == 4) launch_stream_t<T, 4, false>(kp, ntw, n_blocks, st);
  else if (ng == 2) launch_stream_t<T, 2, false>(kp, ntw, n_blocks, st);
  else if (kp.gs_shift == 7) launch_stream_t<T, 1, false>(kp, ntw, n_blocks, st);  // group 128
  else launch_stream_t<T, 1, true>(kp, ntw, n_blocks, st);
}

void launch_gemm_stream(const GemmKParams& kp, int dtype, int ng, int ntw, int n_blocks, hipStream_t st) {
  if (dtype == SLM_BF16) launch_stream_ng<bf16_tag>(kp, ng, ntw, n_blocks, st);
  else launch_stream_ng<f16_tag>(kp, ng, ntw, n_blocks, st);
}

}  // namespace slm
