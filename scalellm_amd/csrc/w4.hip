// w4.hip -- int4-weight x fp16/bf16-activation GEMM for gfx950 (CDNA4): prepack, dequant, GEMM.
//
// Replaces the reference's vendored Marlin path (src/kernels/quantization/marlin/*:
// marlin::gptq_gemm gptq_gemm.cu:585-710, marlin::gptq_repack gptq_repack.cu:252,
// marlin::awq_repack awq_repack.cu:191, permute_cols_kernel gptq_gemm.cu:69-118) with a layout
// and a kernel designed for the CDNA4 matrix core (v_mfma_f32_32x32x16_{bf16,f16}, wave64):
//
//  packed weights  wq[K/64][N/32][64 lanes][4] u32 (kt-major): lane l, word j = the 8 weights
//      n = 32*nt + (l & 31),  k = 64*kt + 16*j + 8*(l >> 5) + e,  e = 0..7
//    i.e. exactly the B-operand fragment of one 32x32x16 MFMA, so a wave's 16-B/lane load is one
//    contiguous KiB that feeds 4 MFMA k-steps with NO shuffle and NO LDS round trip.  kt-major
//    order: at one k position the column tiles of all concurrently running waves are adjacent in
//    memory, so the chip reads one dense N/32-KiB stripe at a time (an nt-major order makes every
//    wave stream at a power-of-two stride from its neighbours: measured HBM-channel pile-up,
//    ~2.2 TB/s ceiling).  Nibbles are
//    pair-interleaved (bit 4i = e 2i, bit 16+4i = e 2i+1) so `(w >> 4i) & 0x000F000F | magic`
//    yields a packed 16-bit pair directly (v_and_or_b32).
//  scale/zero table sz[G][N] u32 = { scale : T, magic+zero : T }  (magic = 128 bf16 / 1024 fp16;
//    the sum is exact in T), one 4-byte load per (group, column).
//  dequant: w = (q - z) * s, bit-identical to "dequantise in T then multiply" (reference
//    marlin/numeric_conversion.h:19-62,121-166,232-240: magic-number int4 -> T, sub zp, scale):
//      fp16: v_pk_add_f16 (exact) + v_pk_mul_f16 (RN);  bf16 (no packed bf16 VALU on gfx950):
//      fp32 fma(128+q, s, -(128+z)s) is exact, then v_cvt_pk_bf16_f32 (RN).
//  GEMM: C[M,N] = A[M,K] . W[K,N]; activations are the MFMA A operand (rows = tokens), staged
//    per 128-deep K chunk through XOR-swizzled LDS (conflict-free ds_read_b128); weights go
//    HBM -> registers -> dequant -> MFMA B operand; fp32 accumulate; optional split-K with fp32
//    partials + reduce (bias added after the reduction, as qlinear_awq_marlin_impl.cpp:357-363).
//
// This file holds the prepack / dequant kernels, the GENERAL GEMM kernel (32..128-row tiles) and the
// one entry point, slm_w4a16_gemm, whose plan picks among five kernels by (M, N, K):
//    M == 1                          w4_gemv.hip   dot2 GEMV, K split inside the workgroup
//    M <= 32                         w4_small.hip  lean weight stream (MFMA, post-scaled)
//    32 < M <= 128, narrow layers    this file
//    M > 128, >= 112 tiles of 256x128 w4_ws.hip     producer / consumer waves, LDS-DMA activations
//    prefill-sized M x N             w4_xl.hip     symmetric 256 x 256 tiles
// (DESIGN.md 3.3 has the measurements behind each boundary.)
#include "w4_common.h"
#include "tuning.h"

namespace slm {

__device__ __forceinline__ int awq_pos(int col_in_word) {  // [0,2,4,6,1,3,5,7] interleave
  return (col_in_word >> 1) + 4 * (col_in_word & 1);
}

// SLM_W4_PAIRED: the checkpoint tensor is a merged [gate | up] column-parallel weight
// (layers/linear/multi_parallel_linear.cpp:14-41 concatenates the two along N); its packed form
// interleaves the halves by 32-column tile -- packed tile 2j = gate tile j, packed tile 2j+1 = up
// tile j -- so the wave (pair) that owns gate columns also owns the matching up columns and can
// apply SiLU*mul in the GEMM epilogue (SLM_W4_SILU_MUL).
__device__ __forceinline__ int64_t paired_src_col(int format, int64_t n_packed, int64_t N) {
  if (!(format & SLM_W4_PAIRED)) return n_packed;
  return (n_packed >> 6) * 32 + (n_packed & 31) + ((n_packed & 32) ? N / 2 : 0);
}

// ------------------------------------------------------------------------------------------
// prepack: checkpoint formats -> wq / sz   (bit-exact integer work)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) w4_prepack_weight_kernel(
    int format, const uint32_t* __restrict__ qweight, const int* __restrict__ perm, int64_t K,
    int64_t N, uint32_t* __restrict__ wq) {
  const int64_t widx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (widx >= N * K / 8) return;
  const int j = (int)(widx & 3);
  const int lane = (int)((widx >> 2) & 63);
  const int64_t tile = widx >> 8;
  const int64_t nt = tile % (N / 32), kt = tile / (N / 32);  // kt-major: see layout note
  const int64_t n = paired_src_col(format, nt * 32 + (lane & 31), N);
  const int64_t kb = kt * 64 + j * 16 + (lane >> 5) * 8;
  uint32_t out = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int64_t k = perm ? (int64_t)perm[kb + e] : kb + e;
    uint32_t q;
    if (k < 0)  // padding row of an uneven act-order shard: its activation column is gathered as 0
      q = 0u;
    else if ((format & SLM_W4_FORMAT_MASK) == SLM_W4_GPTQ)
      q = (qweight[(k / 8) * N + n] >> (4 * (k % 8))) & 0xFu;
    else
      q = (qweight[k * (N / 8) + n / 8] >> (4 * awq_pos((int)(n % 8)))) & 0xFu;
    const int pos = (e >> 1) + 4 * (e & 1);
    out |= q << (4 * pos);
  }
  wq[widx] = out;
}

__global__ void __launch_bounds__(256) w4_prepack_sz_kernel(
    int format, const uint32_t* __restrict__ qzeros, const uint16_t* __restrict__ scales, int64_t G,
    int64_t N, int dtype, uint32_t* __restrict__ sz) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= G * N) return;
  const int64_t g = idx / N, n = paired_src_col(format, idx % N, N);
  uint32_t z = 8u;  // no zero-point tensor: symmetric quantisation, zero = 2^(bits-1) (Marlin has_zp = false)
  if (qzeros) {
    const uint32_t zw = qzeros[g * (N / 8) + n / 8];
    if ((format & SLM_W4_FORMAT_MASK) == SLM_W4_GPTQ)
      z = ((zw >> (4 * (n % 8))) & 0xFu) + 1u;  // qlinear_impl.cpp:45 (zeros.add_(1))
    else
      z = (zw >> (4 * awq_pos((int)(n % 8)))) & 0xFu;
  }
  const uint32_t zm = (dtype == SLM_BF16 ? 0x4300u : 0x6400u) + z;  // 128 + z  /  1024 + z
  sz[idx] = (uint32_t)scales[g * N + n] | (zm << 16);
}

// debug / parity: dense W[K, N] in T from the packed form (uses W4Dq = the GEMM's dequant)
template <typename T>
__global__ void __launch_bounds__(256) w4_dequant_kernel(const uint32_t* __restrict__ wq,
                                                         const uint32_t* __restrict__ sz, int64_t K,
                                                         int64_t N, int64_t gs,
                                                         uint16_t* __restrict__ w_out) {
  const int64_t widx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (widx >= N * K / 8) return;
  const int j = (int)(widx & 3);
  const int lane = (int)((widx >> 2) & 63);
  const int64_t tile = widx >> 8;
  const int64_t nt = tile % (N / 32), kt = tile / (N / 32);  // kt-major: see layout note
  const int64_t n = nt * 32 + (lane & 31);
  const int64_t kb = kt * 64 + j * 16 + (lane >> 5) * 8;
  const W4Dq<T> dq(sz[(kb / gs) * N + n]);
  uint32_t o[4];
  dq.word(wq[widx], o);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    w_out[(kb + 2 * i) * N + n] = (uint16_t)(o[i] & 0xffffu);
    w_out[(kb + 2 * i + 1) * N + n] = (uint16_t)(o[i] >> 16);
  }
}

// A'[m, k] = A[m, perm[k]]  (act-order; reference permute_cols_kernel gptq_gemm.cu:69-118)
__global__ void __launch_bounds__(256) w4_permute_cols_kernel(const uint16_t* __restrict__ a,
                                                              const int* __restrict__ perm,
                                                              int64_t M, int64_t K, int64_t lda,
                                                              uint16_t* __restrict__ out) {
  const int64_t m = blockIdx.y;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < K;
       k += (int64_t)gridDim.x * blockDim.x)
    out[m * K + k] = perm[k] >= 0 ? a[m * lda + perm[k]] : (uint16_t)0;  // < 0: padding column (+0.0)
}

// ------------------------------------------------------------------------------------------
// GEMM
// ------------------------------------------------------------------------------------------
// W4_KC = 128: K chunk (LDS row = 256 B = 16 x 16-B slots, XOR-swizzled by row&15)

// MT : 32-token tiles per workgroup (BM = 32*MT)
// NTW: 32-column tiles per wave      (BN = 128*NTW, 4 waves split N: weights stay wave-private)
// NG : scale groups per 128-deep chunk (1 for group >= 128, 2 for 64, 4 for 32)
// PC : K chunks staged per pass; PC*MT*8 KiB per LDS buffer (32 KiB when PC*MT = 4), 2 buffers.
//
// Pipeline per pass (PC chunks = 2*PC weight loads per n-tile per lane):
//   top   : issue the NEXT pass's scale loads and A-tile loads (global -> registers)
//   body  : for every half-chunk (one 16-B weight load = 4 MFMA k-steps):
//             dequantise its 4 words -> 4 B fragments, re-issue that ring slot with the next pass's
//             load (pinned by sched_barrier so hipcc keeps COUNTED vmcnt waits), then the MFMAs,
//             A fragments coming from the swizzled LDS tile
//   bottom: the A registers (older in the vmcnt queue than the re-issued weight loads, so a counted
//           wait leaves a full pass of weight loads in flight) -> other LDS buffer, one barrier.
template <typename T, int MT, int NTW, int NG, int PC, bool POST>
__global__ void __launch_bounds__(256, (POST && PC * MT == 4 && MT < 4) ? 1 : 2) w4a16_gemm_kernel(const GemmKParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename Mfma<T>::frag frag_t;
  constexpr int BM = 32 * MT;
  constexpr int A_LD = PC * MT * 2;  // 16-B slots staged per thread per pass
  constexpr int CHUNK_BYTES = BM * 256;
  constexpr int BUF_BYTES = PC * CHUNK_BYTES;
  constexpr int HC = 2 * PC;  // half-chunks (16-B weight loads per lane) per pass
  // POST: per-(chunk, group, row) activation sums X, fp32, after the two A buffers
  constexpr int XS_FLOATS = PC * NG * BM;
  float* xs_base = reinterpret_cast<float*>(smem + 2 * BUF_BYTES);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  const int nb = bid % p.n_nblocks;
  bid /= p.n_nblocks;
  const int mb = bid % p.n_mblocks;
  const int ks = bid / p.n_mblocks;

  const int64_t m0 = (int64_t)mb * BM;
  const int c0 = ks * p.chunks_per_split;
  const int c1 = min(p.n_chunks, c0 + p.chunks_per_split);
  const int n_pass = (c1 - c0) / PC;  // host guarantees (c1 - c0) % PC == 0

  // this wave's column tiles (clamped: out-of-range tiles compute on the last valid tile, no store)
  const int64_t n_tiles = p.N / 32;
  int64_t ntile[NTW];
  bool nvalid[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int64_t g = ((int64_t)nb * 4 + wave) * NTW + t;
    nvalid[t] = g < n_tiles;
    ntile[t] = nvalid[t] ? g : n_tiles - 1;
  }

  f32x16 acc[NTW][MT];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][m][r] = 0.f;

  f32x16 tmp[POST ? NTW : 1][POST ? MT : 1];  // per-group partial sums (POST form only)
  (void)tmp;

  // ---- A staging: thread -> (chunk, row, slot), global 16-B loads, swizzled LDS writes ----
  const char* abase = reinterpret_cast<const char*>(p.a);
  u32x4 areg[A_LD];
  auto a_load = [&](int cfirst) {  // chunks cfirst .. cfirst+PC-1
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      const int idx = tid + 256 * i;
      const int ch = idx / (BM * 16), rem = idx % (BM * 16);
      const int row = rem >> 4, slot = rem & 15;
      const int64_t m = m0 + row;
      const int64_t mc = m < p.M ? m : p.M - 1;  // clamp (rows >= M are never stored)
      areg[i] = *reinterpret_cast<const u32x4*>(
          abase + 2 * (mc * p.lda + (int64_t)(cfirst + ch) * W4_KC + slot * 8));
    }
  };
  auto a_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      const int idx = tid + 256 * i;
      const int ch = idx / (BM * 16), rem = idx % (BM * 16);
      const int row = rem >> 4, slot = rem & 15;
      *reinterpret_cast<u32x4*>(smem + buf * BUF_BYTES + ch * CHUNK_BYTES + row * 256 +
                                ((slot ^ (row & 15)) << 4)) = areg[i];
      if constexpr (POST) {
        // the 16 slots of one (chunk, row) sit in 16 consecutive lanes (one DPP row): reduce the
        // 8-element partial sums over the 16/NG lanes of each scale group
        const u32x4 a = areg[i];
        float sum = lo_f32<T>(a.x) + hi_f32<T>(a.x) + lo_f32<T>(a.y) + hi_f32<T>(a.y) +
                    lo_f32<T>(a.z) + hi_f32<T>(a.z) + lo_f32<T>(a.w) + hi_f32<T>(a.w);
        sum = group_sum<16 / NG>(sum);
        if ((slot & (16 / NG - 1)) == 0)
          xs_base[buf * XS_FLOATS + (ch * NG + slot / (16 / NG)) * BM + row] = sum;
      }
    }
  };

  // ---- weight ring (one 16-B load per half-chunk per n-tile) and per-chunk scale/zero words ----
  u32x4 wreg[NTW][HC];
  uint32_t szcur[NTW][PC][NG], sznext[NTW][PC][NG];
  auto w_issue = [&](int t, int h, int cfirst) {  // half-chunk h of the pass starting at cfirst
    const uint32_t* wp = p.wq + ((((int64_t)cfirst * 2 + h) * n_tiles + ntile[t]) * 64 + lane) * 4;
    // plain (cacheable) loads, not non-temporal ones (round 4): a layer's weights are re-read within
    // ~0.4 ms -- by the second BM = 64 row block at M = 65...128 and by the second lane of the two-lane
    // decode step -- and a cacheable line is still in the Infinity Cache then.  Two-lane bs 256 step
    // 24.85 -> 23.98 ms, one lane 26.1 -> 25.9; stand-alone with rotating weights 3-6 % slower (124 ->
    // 132 us per layer at M = 128): the step is what counts.  The kernels that read every weight once
    // per launch (w4_ks / w4_gemv: M <= 32; w4_ws at M = 256) keep their nt loads.
    wreg[t][h] = *reinterpret_cast<const u32x4*>(wp);
  };
  auto sz_load = [&](uint32_t (&dst)[NTW][PC][NG], int cfirst) {
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
      for (int c = 0; c < PC; ++c)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const int64_t grp = ((int64_t)(cfirst + c) * W4_KC + g * (W4_KC / NG)) >> p.gs_shift;
          dst[t][c][g] = p.sz[grp * p.N + ntile[t] * 32 + (lane & 31)];
        }
  };

  // dequantise (PRE) / unpack (POST) one 16-B weight vector into 4 MFMA B fragments
  auto make_frags = [&](const u32x4 wv, const uint32_t (&szc)[NG], int half, frag_t (&out)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int w8 = half * 4 + j;  // word index inside the 128-deep chunk
      uint32_t o[4];
      const uint32_t word = j == 0 ? wv.x : j == 1 ? wv.y : j == 2 ? wv.z : wv.w;
      if constexpr (POST) {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = ((word >> (4 * i)) & 0x000F000Fu) | W4Magic<T>::bits;
      } else {
        const W4Dq<T> dq(szc[w8 * NG / 8]);
        dq.word(word, o);
      }
      const u32x4 packed = {o[0], o[1], o[2], o[3]};
      out[j] = __builtin_bit_cast(frag_t, packed);
    }
  };

  // Software pipeline inside the wave: while the MFMAs of half-chunk h run (matrix pipe), the VALU
  // dequantises half-chunk h+1 into the other fragment buffer; the ring slot of h+1 is then
  // re-issued one pass ahead (pinned by sched_barrier so the vmcnt waits stay counted).
  frag_t bfrag[2][NTW][4];
  if (n_pass > 0) {
    sz_load(szcur, c0);
    a_load(c0);
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
      for (int h = 0; h < HC; ++h) w_issue(t, h, c0);
    a_store(0);
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      make_frags(wreg[t][0], szcur[t][0], 0, bfrag[0][t]);
      w_issue(t, 0, c0 + min(1, n_pass - 1) * PC);
    }
  }
  __syncthreads();

  const int mrow = lane & 31, kh = lane >> 5;
  for (int ps = 0; ps < n_pass; ++ps) {
    const int buf = ps & 1;
    const int cnext = c0 + min(ps + 1, n_pass - 1) * PC;   // clamped: last pass reloads itself
    const int cnext2 = c0 + min(ps + 2, n_pass - 1) * PC;
    sz_load(sznext, cnext);
    a_load(cnext);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < HC; ++h) {
      const int cl = h >> 1;  // chunk within the pass
      const int cur = h & 1, nxt = cur ^ 1;
      const int hf = (h + 1) % HC;          // following half-chunk (first of the next pass at the end)
      const bool wrap = (h + 1) == HC;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int w8 = (h & 1) * 4 + j;
        const int slot = w8 * 2 + kh;
        constexpr int WPG = 8 / NG;  // k-steps (words) per scale group
        const bool g_first = (w8 % WPG) == 0, g_last = (w8 % WPG) == WPG - 1;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int row = m * 32 + mrow;
          const u32x4 av = *reinterpret_cast<const u32x4*>(
              smem + buf * BUF_BYTES + cl * CHUNK_BYTES + row * 256 + ((slot ^ (row & 15)) << 4));
          const frag_t af = __builtin_bit_cast(frag_t, av);
#pragma unroll
          for (int t = 0; t < NTW; ++t) {
            if constexpr (POST) {
              if (g_first) {
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.f;
                tmp[t][m] = Mfma<T>::run(af, bfrag[cur][t][j], z);
              } else {
                tmp[t][m] = Mfma<T>::run(af, bfrag[cur][t][j], tmp[t][m]);
              }
              if (g_last) {
                // acc += s * (tmp - (magic + z) * X[row]) for this lane's column
                float sc, zm;
                W4Magic<T>::decode(szcur[t][cl][w8 * NG / 8], sc, zm);
                const float nzs = -zm * sc;
                const float* xs = xs_base + buf * XS_FLOATS + (cl * NG + w8 * NG / 8) * BM + m * 32 + 4 * kh;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                  const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + 8 * q4);
#pragma unroll
                  for (int e = 0; e < 4; ++e) {
                    const int r = q4 * 4 + e;
                    acc[t][m][r] = fmaf(sc, tmp[t][m][r], fmaf(nzs, xv[e], acc[t][m][r]));
                  }
                }
              }
            } else {
              acc[t][m] = Mfma<T>::run(af, bfrag[cur][t][j], acc[t][m]);
            }
          }
        }
        // one word of the following half-chunk per k-step, in the shadow of the MFMAs above
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          const u32x4 wv = wreg[t][hf];
          const int w8n = (hf & 1) * 4 + j;
          uint32_t o[4];
          const uint32_t word = j == 0 ? wv.x : j == 1 ? wv.y : j == 2 ? wv.z : wv.w;
          if constexpr (POST) {
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = ((word >> (4 * i)) & 0x000F000Fu) | W4Magic<T>::bits;
          } else {
            const uint32_t szw = wrap ? sznext[t][0][w8n * NG / 8] : szcur[t][hf >> 1][w8n * NG / 8];
            const W4Dq<T> dq(szw);
            dq.word(word, o);
          }
          const u32x4 packed = {o[0], o[1], o[2], o[3]};
          bfrag[nxt][t][j] = __builtin_bit_cast(frag_t, packed);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < NTW; ++t) w_issue(t, hf, wrap ? cnext2 : cnext);  // slot hf is free again
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
      for (int c = 0; c < PC; ++c)
#pragma unroll
        for (int g = 0; g < NG; ++g) szcur[t][c][g] = sznext[t][c][g];
    a_store(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C/D layout of 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  if (p.silu && p.split_k == 1) {
    // SLM_W4_SILU_MUL: column tiles are (gate, up) pairs.  NTW == 2: both tiles of a pair are this
    // wave's own; NTW == 1: waves (0, 1) and (2, 3) hold a pair -- the up wave hands its T-rounded
    // tile to the gate wave through the (now idle) A buffers, same lane, same (m, r).
    const uint16_t* bias = reinterpret_cast<const uint16_t*>(p.bias);
    uint16_t* ex = reinterpret_cast<uint16_t*>(smem) + (wave >> 1) * (MT * 1024);
    if constexpr (NTW == 1) {
      if (wave & 1) {
        const float bu = bias ? lo_f32<T>((uint32_t)bias[ntile[0] * 32 + (lane & 31)]) : 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) ex[(m * 16 + r) * 64 + lane] = pack1<T>(acc[0][m][r] + bu);
      }
      __syncthreads();
      if (wave & 1) return;
    }
    if (!nvalid[0]) return;
    const int64_t gcol = ntile[0] * 32 + (lane & 31), ocol = (ntile[0] >> 1) * 32 + (lane & 31);
    const float bg = bias ? lo_f32<T>((uint32_t)bias[gcol]) : 0.f;
    const float bu2 = (NTW == 2 && bias) ? lo_f32<T>((uint32_t)bias[gcol + 32]) : 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const float g = lo_f32<T>((uint32_t)pack1<T>(acc[0][m][r] + bg));
        float u;
        if constexpr (NTW == 2) u = lo_f32<T>((uint32_t)pack1<T>(acc[NTW - 1][m][r] + bu2));
        else u = lo_f32<T>((uint32_t)ex[(m * 16 + r) * 64 + lane]);
        if (row < p.M)
          reinterpret_cast<uint16_t*>(p.c)[row * p.ldc + ocol] = pack1<T>(silu_mul1(g, u));
      }
    }
    return;
  }
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    if (!nvalid[t]) continue;
    const int64_t n = ntile[t] * 32 + (lane & 31);
    float bv = 0.f;
    if (p.split_k == 1 && p.bias) {
      const uint16_t braw = reinterpret_cast<const uint16_t*>(p.bias)[n];
      bv = lo_f32<T>((uint32_t)braw);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < p.M) {
          if (p.split_k == 1)
            reinterpret_cast<uint16_t*>(p.c)[row * p.ldc + n] = pack1<T>(acc[t][m][r] + bv);
          else if (p.ks_dbg & 8) {  // (probe bit 8: tile-contiguous slab layout -- consumers not adapted, timing only)
            const int64_t tile = (int64_t)mb * p.n_nblocks + nb;
            const int rit = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int cit = (wave * NTW + t) * 32 + (lane & 31);
            p.part[(int64_t)ks * p.M * p.N + tile * (BM * 128 * NTW) + rit * (128 * NTW) + cit] = acc[t][m][r];
          } else if ((p.ks_dbg & 16) && ks != 0) {  // (probe bit 16: only slice 0 stores its slab: the traffic of an in-place reduce)
          } else if (!(p.ks_dbg & 4))  // (probe bit 4 of SLM_W4_KS_DBG: no slab stores -- WRONG results, timing only)
            p.part[((int64_t)ks * p.M + row) * p.N + n] = acc[t][m][r];
        }
      }
    }
  }
}

// C[m, n] = T( sum_s part[s][m][n] + bias[n] )
template <typename T>
__global__ void __launch_bounds__(256) w4_splitk_reduce_kernel(const float* __restrict__ part,
                                                               const void* __restrict__ bias,
                                                               void* __restrict__ c, int64_t M,
                                                               int64_t N, int64_t ldc, int split_k) {
  const int64_t idx4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // 4 columns per thread
  if (idx4 * 4 >= M * N) return;
  const int64_t m = (idx4 * 4) / N, n = (idx4 * 4) % N;
  f32x4 s = splitk_sum4(part + m * N + n, M * N, split_k);
  if (bias) {
    const u32x2 b = *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(bias) + n);
    s.x += lo_f32<T>(b.x); s.y += hi_f32<T>(b.x);
    s.z += lo_f32<T>(b.y); s.w += hi_f32<T>(b.y);
  }
  u32x2 r;
  r.x = pack2<T>(s.x, s.y);
  r.y = pack2<T>(s.z, s.w);
  *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(c) + m * ldc + n) = r;
}

// SLM_W4_SILU_MUL with split-K: C[m, i] = T( silu(g) * u ), g / u = T( sum_s part[s][m][col] + bias )
// at the gate / up columns of output column i (packed tile pair 2j, 2j+1; N = packed width)
template <typename T>
__global__ void __launch_bounds__(256) w4_splitk_reduce_silu_kernel(
    const float* __restrict__ part, const void* __restrict__ bias, void* __restrict__ c, int64_t M,
    int64_t N, int64_t ldc, int split_k) {
  const int64_t idx4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // 4 output columns per thread
  const int64_t half = N / 2;
  if (idx4 * 4 >= M * half) return;
  const int64_t m = (idx4 * 4) / half, oc = (idx4 * 4) % half;
  const int64_t gc = (oc >> 5) * 64 + (oc & 31);
  f32x4 g = splitk_sum4(part + m * N + gc, M * N, split_k);
  f32x4 u = splitk_sum4(part + m * N + gc + 32, M * N, split_k);
  if (bias) {
    const uint16_t* bp = reinterpret_cast<const uint16_t*>(bias) + gc;
    const u32x2 bg = *reinterpret_cast<const u32x2*>(bp);
    const u32x2 bu = *reinterpret_cast<const u32x2*>(bp + 32);
    g.x += lo_f32<T>(bg.x); g.y += hi_f32<T>(bg.x); g.z += lo_f32<T>(bg.y); g.w += hi_f32<T>(bg.y);
    u.x += lo_f32<T>(bu.x); u.y += hi_f32<T>(bu.x); u.z += lo_f32<T>(bu.y); u.w += hi_f32<T>(bu.y);
  }
  u32x2 r;
  r.x = pack2<T>(silu_mul_acc<T>(g.x, u.x), silu_mul_acc<T>(g.y, u.y));
  r.y = pack2<T>(silu_mul_acc<T>(g.z, u.z), silu_mul_acc<T>(g.w, u.w));
  *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(c) + m * ldc + oc) = r;
}

// ------------------------------- host side ------------------------------------------
// Instantiations of the general kernel that do not fit 256 VGPRs (hipcc spills: measured 4-7x
// SLOWER than the PRE form of the same tile -- group 32 at 32 < M <= 64, qkv 83 vs 19 us) are neither
// planned nor built: the post-scaled form needs one more accumulator set per scale group and tile.
constexpr bool w4_post_fits(int mt, int ntw, int ng, int pc) {
  return mt <= 2 && ntw == 1 && !(ng == 4 && (mt == 2 || pc >= 2));
}
constexpr bool w4_pre_fits(int /*mt*/, int ntw, int ng, int pc) { return !(ntw == 2 && ng == 4 && pc >= 2); }

struct GemmPlan {
  int xl_sk, sk_per;      // stream-K form of the 256 x 256 kernel (w4_xl.hip): 128-deep chunks per workgroup
  int mt, ntw, ng, pc, post, small, gemv, split_k, chunks_per_split, n_mblocks, n_nblocks;
  int ks, ks_cw, ks_nw, ks_tpw, ks_mt;  // K-sliced small-M kernel (w4_ks.hip)
  int m128, m128_wd, m128_kw, m128_ct, m128_adma;  // 65 <= M <= 128 kernel (w4_m128.hip)
  size_t lds_bytes, part_bytes, aperm_bytes;
};

static int plan_gemm(const slm_w4_gemm_args* a, GemmPlan* pl) {
  if (!a) return SLM_ERR_INVALID_ARG;
  if (a->M < 0 || a->K <= 0 || a->N <= 0) return SLM_ERR_INVALID_ARG;
  if (a->dtype != SLM_F16 && a->dtype != SLM_BF16) return SLM_ERR_UNSUPPORTED;
  if (a->K % W4_KC || a->N % 32) return SLM_ERR_UNSUPPORTED;  // reference: K%128, N%64
  if (a->flags & ~(SLM_W4_DEFER_REDUCE | SLM_W4_SILU_MUL | SLM_W4_SHARES_CHIP)) return SLM_ERR_INVALID_ARG;
  if (a->flags & SLM_W4_SILU_MUL) {
    if (a->flags & SLM_W4_DEFER_REDUCE) return SLM_ERR_INVALID_ARG;
    if (a->N % 64) return SLM_ERR_UNSUPPORTED;
  }
  const int64_t gs = a->group_size;
  if (!(gs == 32 || gs == 64 || (gs >= 128 && gs % 128 == 0 && is_pow2(gs)) || gs == a->K))
    return SLM_ERR_UNSUPPORTED;
  if (a->K % gs) return SLM_ERR_UNSUPPORTED;
  pl->ng = gs == 32 ? 4 : gs == 64 ? 2 : 1;
  // Launch shape from tools/sweep_gemm.py on MI355X (profiles/r01_gemm_sweep.jsonl):
  //  M <= 64 : one M tile (MT = 1/2), post-scaled dequant, ~256 workgroups (split-K fills the chip)
  //  M  > 64 : BM = 128 when the N x M tiling alone gives >= 256 tiles or K is deep, else BM = 64;
  //            ~512 workgroups (2 per CU), split-K <= 8
  const int n_chunks = (int)(a->K / W4_KC);
  int mt;
  int xl_sk = 0;
  if (a->M <= 32) mt = 1;
  else if (a->M <= 64) mt = 2;
  else if (a->M <= 128) {
    // BM = 128 when its tiles alone fill the chip, or when K is deep AND there are enough column
    // tiles to spread (a deep, very narrow shard -- 70B TP=8 qkv: 8192 x 1280 -- runs 20 % faster
    // on twice as many BM = 64 tiles: 22.0 -> 17.7 us)
    // (round 4: the deep-K rule starts at 64 tiles, not 32.  Llama-3-8B's down_proj -- 14336 x 4096, 32 tiles --
    // is 3 % faster alone on BM = 128 (33.0 vs 34.1 us at M = 128), but in the two-lane decode step its 64 KiB,
    // ~200-VGPR workgroups share a CU badly with the attention stream's: 79 -> 135 us per call once the stream
    // kernel keeps two row chunks per lane, 58 us on BM = 64 tiles like the other three layers)
    const int64_t tiles4 = ((a->M + 127) / 128) * ((a->N + 127) / 128);
    mt = (tiles4 >= 256 || (a->K >= 8192 && tiles4 >= 64)) ? 4 : 2;
  } else {
    // M > 128: the wave-specialised 256 x 128 kernel (w4_ws.hip) when its tiles alone keep about
    // half of the 256 CUs busy (measured: 0.89-1.06 PFLOP/s vs 0.74-0.86 for the single-role
    // kernel on gate_up/down at M = 256..2048); narrow layers stay on the 128 x 128 kernel, whose
    // 2 workgroups per CU need less split-K.  Its A addressing uses 32-bit offsets.
    const int64_t tiles8 = ((a->M + 255) / 256) * ((a->N + 127) / 128);
    const bool a_fits = ((a->M - 1) * a->lda + a->K) * 2 < ((int64_t)1 << 31);
    const int64_t tiles4 = ((a->M + 127) / 128) * ((a->N + 127) / 128);
    mt = (tiles8 >= 112 && a_fits) ? 8 : ((tiles4 >= 256 || a->K >= 8192) ? 4 : 2);
    // prefill-sized problems: the symmetric 256 x 256 kernel (w4_xl.hip) when its tiles fill whole
    // rounds of the 256 CUs (measured +5..7 % over the 256 x 128 kernel there, -36 % when they don't)
    const int64_t tiles16 = ((a->M + 255) / 256) * ((a->N + 255) / 256);
    const int64_t rounds = (tiles16 + 255) / 256;
    if (a_fits && tiles16 >= 224 && tiles16 * 100 >= rounds * 256 * 87) mt = 16;
    // ... and when the 256 x 128 tiles need MORE rounds than they save (round 5, M = 2648 -- the mixed step's row
    // count: o 4096 x 4096 = 176 / 352 tiles: one 69 %-full round of 256 x 256 tiles 102.6 us against two rounds of
    // 256 x 128 tiles 117.4; down 340.8 against 382.8): a 256 x 128 tile takes 0.58 of a 256 x 256 tile's time
    // (profiles/r05_gemm_large_m.jsonl; layer chain 1282 -> 1192 us at M = 2648, 1400 -> 1322 at 3072,
    // the mixed step 55.1 -> 52.6 ms)
    const int64_t rounds8 = (tiles8 + 255) / 256;
    // (and only with its rounds >= 65 % full: at 56 % -- qkv at M = 1536, 144 tiles -- the layer chain LOSES 5.6 %)
    if (a_fits && mt == 8 && rounds8 >= 2 && rounds * 100 < rounds8 * 58 && tiles16 * 100 >= rounds * 256 * 65 &&
        tune_get(TUNE_W4_XL_MODEL, 1) != 0)
      mt = 16;
    // Round 6: the STREAM-K form of the 256 x 256 kernel (w4_xl.hip): the tile x K work cut into 256 equal ranges
    // -- no round of tiles is left part-filled.  Cost model in units of one 256 x 256 tile's time: the chosen
    // kernel's rounds (a 256 x 128 tile: 0.58) against 1.4 x tiles16 / 256 (measured 1.28...1.41 over M = 1536...3072:
    // a second pipeline fill per workgroup, the partial tiles' round trip, a fuller chip's lower clock;
    // profiles/r06_gemm_streamk.jsonl).  At least half a tile per
    // workgroup (a tile is then cut into at most three pieces); not with the SiLU pair epilogue (gate_up fills
    // its rounds), not next to another stream (its workgroups wait for each other: SLM_W4_SHARES_CHIP).
    const int sk_mode = tune_get(TUNE_W4_XL_SK, 1);
    if (a_fits && sk_mode != 0 && tiles16 >= 128 && a->N % 256 == 0 && !(a->flags & (SLM_W4_SILU_MUL | SLM_W4_SHARES_CHIP)) &&
        !tune_is_set(TUNE_W4_MT) && !tune_is_set(TUNE_W4_SPLITK)) {
      const double cur = mt == 16 ? (double)rounds : mt == 8 ? 0.58 * (double)rounds8 : 1e30;
      const double sk = 1.40 * (double)tiles16 / 256.0;
      if (sk_mode >= 2 || sk < 0.97 * cur) {
        mt = 16;
        xl_sk = 1;
      }
    }
  }
  // M <= 4: dot2 GEMV (w4_gemv.hip); M <= 32: the lean weight-streaming kernel (w4_small.hip)
  // (measured: the GEMV wins on every layer shape at M = 1 and loses on some at M = 2..4, so the
  // default is M = 1 only; SLM_W4_GEMV=2 forces it for M <= 4; its 32-bit offsets need < 4 GiB tables)
  const int gemv_mode = tune_get(TUNE_W4_GEMV, 1);
  pl->gemv = (gemv_mode != 0 && (a->M == 1 || gemv_mode == 2) && gemv_supported(a->M, a->K, gs) &&
              a->K * a->N / 2 < ((int64_t)1 << 32) && (a->K / gs) * a->N * 4 < ((int64_t)1 << 32)) ? 1 : 0;
  pl->small = (a->M <= 32 && tune_get(TUNE_W4_SMALL, 1) != 0 &&
               a->K * a->N / 2 < ((int64_t)1 << 32) && (a->K / gs) * a->N * 4 < ((int64_t)1 << 32) &&
               ((a->M - 1) * a->lda + a->K) * 2 < ((int64_t)1 << 31)) ? 1 : 0;
  mt = tune_get(TUNE_W4_MT, mt);
  if (a->M > 64 && a->M <= 128 && a->N >= 16384) mt = tune_get(TUNE_W4_MT_WIDE, mt);  // (wide layers: gate_up)
  if (pl->small) mt = 1;
  // 8 = wave-specialised 256 x 128 kernel (w4_ws.hip), 16 = symmetric 256 x 256 kernel (w4_xl.hip)
  if (mt != 1 && mt != 2 && mt != 4 && mt != 8 && mt != 16) mt = 4;
  if (mt >= 8 && ((a->M - 1) * a->lda + a->K) * 2 >= ((int64_t)1 << 31)) mt = 4;
  if (mt != 16 || pl->small || pl->gemv) xl_sk = 0;
  int ntw = tune_get(TUNE_W4_NTW, 1);
  if (ntw != 1 && ntw != 2) ntw = 1;
  if (mt >= 4) ntw = 1;
  if (pl->small) ntw = 1;
  pl->mt = mt;
  pl->ntw = ntw;
  const int bm = mt == 16 ? 256 : 32 * mt, bn = mt == 16 ? 256 : 128 * ntw;
  pl->n_mblocks = (int)((a->M + bm - 1) / bm);
  pl->n_nblocks = (int)((a->N + bn - 1) / bn);
  const int64_t tiles = (int64_t)pl->n_mblocks * pl->n_nblocks;
  // pass = PC chunks per LDS buffer (PC*MT <= 4); one chunk per pass measured best or equal
  int pc = tune_get(TUNE_W4_PC, 1);
  if (pc != 1 && pc != 2 && pc != 4) pc = 1;
  if (pc * mt > 4) pc = mt >= 4 ? 1 : 4 / mt;
  while (pc > 1 && n_chunks % pc) pc >>= 1;
  if (!w4_pre_fits(mt, ntw, pl->ng, pc)) pc = 1;  // (a knob combination that does not fit the registers)
  const int n_units = n_chunks / pc;  // split-K granularity = whole passes
  int split_k = tune_get(TUNE_W4_SPLITK, 0);
  if (split_k <= 0) {
    const int64_t target = (a->M <= 64 || mt >= 8) ? 256 : tune_get(TUNE_W4_SPLIT_TARGET, 512);
    int64_t want = (target + tiles / 2) / (tiles > 0 ? tiles : 1);
    // M > 64: keep >= 8 chunks (1024 of K) per split -- short K (row-parallel TP shards) does not
    // amortise the fp32 partial round trip
    const int64_t cap = a->M <= 64 ? 8 : (n_chunks / 8 > 0 ? n_chunks / 8 : 1);
    if (want > cap) want = cap;
    if (want > 8) want = 8;
    if (want < 1) want = 1;
    if (want > n_units) want = n_units;
    split_k = (int)want;
  }
  if (split_k > n_units) split_k = n_units;
  const int units_per_split = (n_units + split_k - 1) / split_k;
  pl->pc = pc;
  pl->chunks_per_split = units_per_split * pc;
  pl->split_k = (n_units + units_per_split - 1) / units_per_split;
  // small-M tiles use the post-scaled form (7 VALU per 8 weights instead of ~27)
  pl->post = tune_get(TUNE_W4_POST, a->M <= 64 ? 1 : 0) != 0 && w4_post_fits(mt, ntw, pl->ng, pc);
  pl->lds_bytes = (size_t)2 * pc * bm * 256 + (pl->post ? (size_t)2 * pc * pl->ng * bm * sizeof(float) : 0);
  if (pl->gemv) {
    // K is split inside the workgroup: no partials, no reduce launch.  Exception: when the caller
    // defers the reduction to the consumer anyway (SLM_W4_DEFER_REDUCE: RMSNorm, RoPE + append)
    // a narrow layer is ALSO split across workgroups so that its launch covers all the CUs --
    // o_proj at M = 1 has 128 column tiles = 128 workgroups on 256 CUs, and a CU's load path
    // (~14 B/clk) caps 128 of them at ~3.3 TB/s.
    pl->split_k = gemv_global_splits(a->M, a->K, a->N,
                                     (a->flags & SLM_W4_DEFER_REDUCE) && !a->bias && !a->perm);
    pl->chunks_per_split = n_chunks;
  }
  // M <= 32 (M == 1 stays on the GEMV): the K-sliced weight stream (w4_ks.hip) -- K split over the
  // waves of a workgroup (activations in registers), partial tiles reduced through LDS.  Launch
  // shape from tools/bench_small_gemm.py sweeps on MI355X (profiles/r03_ks_sweep_m32.jsonl): the
  // widest K slice per wave wins on every layer shape (fewest workgroups re-reading the
  // activations), tiles per workgroup = about one workgroup per CU.  The slice width depends on K
  // only -- NOT on the epilogue flags -- so that a fused SiLU*mul call and the plain call sum in the
  // same order (bit-identical results, tests/test_w4_silu_gpu.py).
  pl->ks = 0;
  if (tune_get(TUNE_W4_KS, 1) != 0 && a->M >= 1 && a->M <= 32 && !pl->gemv && pl->small &&
      ((a->M - 1) * a->ldc + a->N) * 2 < ((int64_t)1 << 31) && a->M * a->N * 4 < ((int64_t)1 << 31)) {
    const bool silu = (a->flags & SLM_W4_SILU_MUL) != 0;
    const int n_tiles = (int)(a->N / 32);
    const int cw_max = pl->ng == 4 ? 2 : 4;
    int nw = tune_get(TUNE_W4_KS_NW, 0), cw = tune_get(TUNE_W4_KS_CW, 0), tpw = tune_get(TUNE_W4_KS_TPW, 0);
    const int forced_split = tune_get(TUNE_W4_SPLITK, 0);
    if (nw == 0 && cw == 0 && forced_split > 0) {  // tests / sweeps that pin the split: honour it or step aside
      for (int tw = 8; tw >= 4 && !nw; tw -= 4)
        for (int tc = cw_max; tc >= 1 && !nw; tc >>= 1)
          if ((n_chunks + tw * tc - 1) / (tw * tc) == forced_split && gemm_ks_config_ok(pl->ng, tc, tw)) {
            nw = tw;
            cw = tc;
          }
    } else {
      if (nw == 0) nw = n_chunks <= 4 ? 4 : 8;
      if (cw == 0) {
        cw = 1;
        while (cw < cw_max && nw * cw < n_chunks) cw *= 2;
      }
    }
    if (nw && cw && gemm_ks_config_ok(pl->ng, cw, nw)) {
      const int ksplit = (n_chunks + nw * cw - 1) / (nw * cw);
      if (tpw <= 0) {
        tpw = (int)(((int64_t)ksplit * n_tiles + 128) / 256);
        if (tpw < 1) tpw = 1;
      }
      if (silu) tpw = (tpw + 1) & ~1;  // (gate, up) tile pairs stay in one workgroup
      if (tpw > n_tiles) tpw = n_tiles;
      if (ksplit <= 16 && (forced_split <= 0 || ksplit == forced_split)) {
        pl->ks = 1;
        pl->ks_cw = cw; pl->ks_nw = nw; pl->ks_tpw = tpw;
        pl->split_k = ksplit;
        pl->chunks_per_split = nw * cw;
        pl->n_mblocks = 1;
        pl->n_nblocks = (n_tiles + tpw - 1) / tpw;
      }
    }
  }
  pl->ks_mt = 1;
  // 33 <= M <= 64 (round 4; the default outside the two-lane steps since round 6): the K-sliced stream with TWO row tiles -- every
  // weight word unpacked once for two MFMAs.  One chunk of K per wave (the activations of both row
  // tiles fill the registers), 8 waves: a workgroup covers 1024 of K, the rest is split across
  // workgroups (fp32 slabs, summed by the consumer under SLM_W4_DEFER_REDUCE or by the reduce kernel).
  // Measured (profiles/r04_ks_mt2.jsonl, M = 64 stand-alone): qkv 19.5 -> 17.6 us, gate_up 40.4 -> 36.6,
  // o 15.8 -> 15.6, down 26.3 -> 36.1 (14 slabs: excluded below); the bs = 64 decode step 9.07 -> 8.93 ms.
  // NOT under the two-lane decode step (decode.py; SLM_W4_SHARES_CHIP): there its 512-thread, 236-VGPR workgroups
  // cannot share a CU with the other lane's attention waves and wait for them instead -- bs = 128
  // (two lanes of 64 rows) 14.2 -> 21.5 ms -- and the stand-alone gain is small because A (512 KB at
  // M = 64, K = 4096) cannot stay on one CU: either K is split over CUs (slab traffic, this kernel) or
  // A is re-streamed per column tile (the general kernel); the step from M = 32 stays.
  // Round 6: ON by default where the caller does not say the call shares the chip (SLM_W4_SHARES_CHIP, set by
  // the two-lane decode steps): M = 33 / 48 / 64 layer chain 96 / 97 / 101 -> 89 / 90 / 93 us, bs = 64 step
  // 9.17 -> 8.95 ms (profiles/r06_ks_mt2_default.jsonl).  SLM_W4_KS_MT2 = 0 never, 1 / 2 always (2: any split).
  const int mt2_knob = tune_get(TUNE_W4_KS_MT2, -1);
  const bool mt2_on = mt2_knob > 0 || (mt2_knob < 0 && !(a->flags & SLM_W4_SHARES_CHIP));
  if (tune_get(TUNE_W4_KS, 1) != 0 && mt2_on && a->M > 32 && a->M <= 64 && !pl->gemv &&
      a->K * a->N / 2 < ((int64_t)1 << 32) && (a->K / gs) * a->N * 4 < ((int64_t)1 << 32) &&
      ((a->M - 1) * a->lda + a->K) * 2 < ((int64_t)1 << 31) &&
      ((a->M - 1) * a->ldc + a->N) * 2 < ((int64_t)1 << 31) && a->M * a->N * 4 < ((int64_t)1 << 31) &&
      gemm_ks_config_ok(pl->ng, 1, 8, 2)) {
    const bool silu = (a->flags & SLM_W4_SILU_MUL) != 0;
    const int n_tiles = (int)(a->N / 32);
    const int ksplit = (n_chunks + 7) / 8;
    const int forced_split = tune_get(TUNE_W4_SPLITK, 0);
    int tpw = tune_get(TUNE_W4_KS_TPW, 0);
    if (tpw <= 0) {
      tpw = (int)(((int64_t)ksplit * n_tiles + 128) / 256);
      if (tpw < 1) tpw = 1;
    }
    if (silu) tpw = (tpw + 1) & ~1;  // (gate, up) tile pairs stay in one workgroup
    if (tpw > n_tiles) tpw = n_tiles;
    // deep K (down_proj: 14 slabs of fp32 partials) loses to the general kernel's 8-way split with
    // wider tiles (M = 64: 36.1 vs 26.3 us); up to 4 slabs it wins or ties (qkv 17.6 vs 19.5, o 15.6 vs
    // 15.8, gate_up 36.6 vs 40.4 us; profiles/r04_ks_mt2.jsonl).  SLM_W4_KS_MT2=2 lifts the bound (tests).
    const int max_split = mt2_knob >= 2 ? 16 : 4;
    if (ksplit <= max_split && (forced_split <= 0 || ksplit == forced_split)) {
      pl->ks = 1;
      pl->ks_mt = 2;
      pl->ks_cw = 1; pl->ks_nw = 8; pl->ks_tpw = tpw;
      pl->split_k = ksplit;
      pl->chunks_per_split = 8;
      pl->n_mblocks = 1;
      pl->n_nblocks = (n_tiles + tpw - 1) / tpw;
    }
  }
  // 65 <= M <= 128 (round 5): all rows in ONE workgroup (w4_m128.hip) -- every weight word fetched and
  // dequantised once for 4 MFMAs instead of once per 64-row block for 2 -- in 132-VGPR / 32-KiB workgroups
  // that sit twice on a CU next to the decode attention stream of the other lane.  Split-K aims at two
  // workgroups per CU with >= 512 of K each (the consumers take up to 16 slabs).
  pl->m128 = 0;
  pl->m128_wd = 2;
  pl->m128_kw = 1;
  pl->m128_ct = 4;
  pl->m128_adma = 0;
  // Where (measured, profiles/r05_m128_*.jsonl): deep-K layers (K >= 8192: the Llama-3-70B shapes, where the
  // general kernel already took its ~200-VGPR BM = 128 tiles) -- the 70B step 50.6 -> 49.4 ms.  On the
  // Llama-3-8B shapes (K = 4096, and 14336 x 4096) it ties the BM = 64 general kernel alone and in the two-lane
  // step: there the GEMMs are starved of HBM bandwidth by the attention stream, not bound by their
  // instruction count (tools/probe_corun.py), and the plan with fewer, longer workgroups leaves the chain
  // longer.  SLM_W4_M128 = 1 forces it everywhere (tests), 0 disables it.
  const int m128_mode = tune_get(TUNE_W4_M128, -1);
  // ... and wide enough to fill the chip with 128-column tiles (>= 64 of them): the TP = 8 shards of the 70B
  // layers (8192 x 1280, 8192 x 7168) stay on twice as many BM = 64 tiles (rank-0 shard step 11.97 vs 12.11 ms)
  if (m128_mode != 0 && (m128_mode > 0 || (a->K >= 8192 && a->N >= 8192)) && a->M > 64 && a->M <= 128 && !pl->gemv && !pl->ks &&
      !tune_is_set(TUNE_W4_MT) &&
      a->K * a->N / 2 < ((int64_t)1 << 32) && (a->K / gs) * a->N * 4 < ((int64_t)1 << 32) &&
      ((a->M - 1) * a->lda + a->K) * 2 < ((int64_t)1 << 31)) {
    // 256-column workgroups (8 column tiles share the activation panel a CU ingests, w4_m128.hip "CT"): one
    // 512-thread workgroup per CU is the fill they aim at.  Default where the plan picks this kernel itself (the
    // deep, wide 70B shapes: layer at M = 128 306.6 -> 285.0 us, every GEMM of it faster,
    // profiles/r05_m128_ct8.jsonl); forced onto the 8B shapes (SLM_W4_M128 = 1) the two forms tie.
    const int ct = tune_get(TUNE_W4_M128_CT, a->K >= 8192 && a->N >= 8192 ? 8 : 4) == 8 ? 8 : 4;
    const int64_t tiles1 = (a->N + 32 * ct - 1) / (32 * ct);
    const int64_t target = tune_get(TUNE_W4_M128_SPLITS, ct == 8 ? 256 : 512);
    int64_t want = (target + tiles1 / 2) / tiles1;
    const int64_t cap = n_chunks / 4 > 0 ? n_chunks / 4 : 1;
    if (want > cap) want = cap;
    if (want > 16) want = 16;
    if (want < 1) want = 1;
    const int forced = tune_get(TUNE_W4_SPLITK, 0);
    if (forced > 0) want = forced < n_chunks ? forced : n_chunks;
    const int per = (int)((n_chunks + want - 1) / want);
    pl->m128 = 1;
    pl->mt = 4; pl->ntw = 1; pl->pc = 1; pl->post = 0;
    pl->n_mblocks = 1;
    pl->n_nblocks = (int)tiles1;
    pl->chunks_per_split = per;
    pl->split_k = (n_chunks + per - 1) / per;
    // ring depth 4 where it was measured (Llama-3-70B shapes, profiles/r05_m128_70b_shapes.jsonl: layer 318 ->
    // 310 us, gate_up 155 -> 148, down 83 -> 82); the general kernel's BM = 64 tiles: 336 us
    int wd = tune_get(TUNE_W4_M128_WD, a->K >= 8192 ? 4 : 2);
    if (wd != 4 || (2 * per) % 4 != 0 || n_chunks % per != 0) wd = 2;
    pl->m128_wd = wd;
    // two waves per column tile (512-thread workgroups) up to two workgroups per CU: measured on the 70B shapes
    // (profiles/r05_m128_kw.jsonl, one box): layer 333 -> 313 us, gate_up 155 -> 148 (448 workgroups), the others
    // within 1 us (480 / 512 workgroups).  SLM_W4_M128_KW: 1 / 2 force a form.
    const int kw_knob = tune_get(TUNE_W4_M128_KW, 0);
    pl->m128_kw = kw_knob == 2 || (kw_knob != 1 && (int64_t)pl->n_nblocks * pl->split_k <= 512) ? 2 : 1;
    pl->m128_ct = ct;
    if (ct == 8) pl->m128_kw = 1;
    // activations by LDS-DMA (256-column form): 70B layer at M = 128 284.6 -> 271.1 us (profiles/r05_m128_adma.jsonl)
    pl->m128_adma = ct == 8 && tune_get(TUNE_W4_M128_ADMA, 1) != 0;
    pl->lds_bytes = W4_M128_LDS_BYTES;
  }
  pl->xl_sk = 0;
  pl->sk_per = 0;
  if (xl_sk && !pl->ks && !pl->m128 && pl->mt == 16) {
    // equal ranges of the work list, in whole 64-deep chunk QUADS (the kernel's ring granularity: 2 chunks of 128)
    const int64_t work = (int64_t)pl->n_mblocks * pl->n_nblocks * n_chunks;
    int64_t per = (work + W4_XL_SK_WGS - 1) / W4_XL_SK_WGS;
    per = (per + 1) & ~(int64_t)1;
    if (per >= 2 && (n_chunks & 1) == 0 && per < ((int64_t)1 << 30)) {
      pl->xl_sk = 1;
      pl->sk_per = (int)per;
      pl->split_k = 1;
      pl->chunks_per_split = n_chunks;
    }
  }
  pl->part_bytes = pl->split_k > 1 ? (size_t)pl->split_k * a->M * a->N * sizeof(float) : 0;
  if (pl->xl_sk) pl->part_bytes = W4_XL_SK_WGS * W4_XL_SK_SLOT_BYTES + W4_XL_SK_SYNC_BYTES;
  pl->aperm_bytes = a->perm ? (((size_t)a->M * a->K * 2 + 255) & ~(size_t)255) : 0;
  return SLM_OK;
}

template <typename T, int MT, int NTW, int PC>
static void launch_gemm_ng(const GemmKParams& kp, const GemmPlan& pl, hipStream_t st) {
  const dim3 grid((unsigned)((int64_t)pl.n_nblocks * pl.n_mblocks * pl.split_k)), blk(256);
#define SLM_GEMM(NGG, POSTT)                                                                        \
  hipLaunchKernelGGL((w4a16_gemm_kernel<T, MT, NTW, NGG, PC, POSTT>), grid, blk, pl.lds_bytes, st, kp)
  if (pl.post) {  // (plan_gemm only sets it where w4_post_fits: the other instantiations are not built)
    switch (pl.ng) {
      case 4: if constexpr (w4_post_fits(MT, NTW, 4, PC)) SLM_GEMM(4, true); break;
      case 2: if constexpr (w4_post_fits(MT, NTW, 2, PC)) SLM_GEMM(2, true); break;
      default: if constexpr (w4_post_fits(MT, NTW, 1, PC)) SLM_GEMM(1, true); break;
    }
    return;
  }
  switch (pl.ng) {
    case 4: if constexpr (w4_pre_fits(MT, NTW, 4, PC)) SLM_GEMM(4, false); break;
    case 2: SLM_GEMM(2, false); break;
    default: SLM_GEMM(1, false); break;
  }
#undef SLM_GEMM
}

template <typename T, int MT, int NTW>
static void launch_gemm_pc(const GemmKParams& kp, const GemmPlan& pl, hipStream_t st) {
  constexpr int PCMAX = 4 / MT;
  if (pl.pc == PCMAX) launch_gemm_ng<T, MT, NTW, PCMAX>(kp, pl, st);
  else if constexpr (PCMAX >= 4) {
    if (pl.pc == 2) launch_gemm_ng<T, MT, NTW, 2>(kp, pl, st);
    else launch_gemm_ng<T, MT, NTW, 1>(kp, pl, st);
  } else launch_gemm_ng<T, MT, NTW, 1>(kp, pl, st);
}

template <typename T>
static void launch_gemm(const GemmKParams& kp, const GemmPlan& pl, hipStream_t st) {
  if (pl.mt == 4) launch_gemm_pc<T, 4, 1>(kp, pl, st);
  else if (pl.mt == 2 && pl.ntw == 2) launch_gemm_pc<T, 2, 2>(kp, pl, st);
  else if (pl.mt == 2) launch_gemm_pc<T, 2, 1>(kp, pl, st);
  else if (pl.ntw == 2) launch_gemm_pc<T, 1, 2>(kp, pl, st);
  else launch_gemm_pc<T, 1, 1>(kp, pl, st);
}

}  // namespace slm

using namespace slm;

static bool w4_format_ok(int32_t format, int64_t N) {
  const int32_t base = format & SLM_W4_FORMAT_MASK;
  if (format & ~(SLM_W4_FORMAT_MASK | SLM_W4_PAIRED)) return false;
  if (base != SLM_W4_GPTQ && base != SLM_W4_AWQ) return false;
  return !(format & SLM_W4_PAIRED) || N % 64 == 0;
}

extern "C" {

SLM_API size_t slm_w4_packed_weight_bytes(int64_t K, int64_t N) {
  if (K <= 0 || N <= 0 || K % 64 || N % 32) return 0;
  return (size_t)K * N / 2;
}

SLM_API size_t slm_w4_packed_sz_bytes(int64_t K, int64_t N, int64_t group_size) {
  if (K <= 0 || N <= 0 || group_size <= 0 || K % group_size) return 0;
  return (size_t)(K / group_size) * N * sizeof(uint32_t);
}

SLM_API int slm_w4_prepack_weights(int32_t format, const int32_t* qweight, const int32_t* perm,
                                   int64_t K, int64_t N, void* wq_out, void* stream) {
  if (!qweight || !wq_out) return SLM_ERR_INVALID_ARG;
  if (!w4_format_ok(format, N)) return SLM_ERR_UNSUPPORTED;
  if (K <= 0 || N <= 0 || K % 64 || N % 32) return SLM_ERR_UNSUPPORTED;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hip_clear_error();
  const int64_t words = K * N / 8;
  hipLaunchKernelGGL(w4_prepack_weight_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0,
                     st, format, reinterpret_cast<const uint32_t*>(qweight), perm, K, N,
                     reinterpret_cast<uint32_t*>(wq_out));
  return hip_check_launch();
}

SLM_API int slm_w4_prepack_sz(int32_t format, const int32_t* qzeros, const void* scales, int64_t K,
                              int64_t N, int64_t group_size, int32_t dtype, void* sz_out,
                              void* stream) {
  if (!scales || !sz_out) return SLM_ERR_INVALID_ARG;
  if (!w4_format_ok(format, N)) return SLM_ERR_UNSUPPORTED;
  if (dtype != SLM_F16 && dtype != SLM_BF16) return SLM_ERR_UNSUPPORTED;
  if (K <= 0 || N <= 0 || N % 32 || group_size <= 0 || K % group_size) return SLM_ERR_UNSUPPORTED;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hip_clear_error();
  const int64_t G = K / group_size;
  hipLaunchKernelGGL(w4_prepack_sz_kernel, dim3((unsigned)((G * N + 255) / 256)), dim3(256), 0, st,
                     format, reinterpret_cast<const uint32_t*>(qzeros),
                     reinterpret_cast<const uint16_t*>(scales), G, N, dtype,
                     reinterpret_cast<uint32_t*>(sz_out));
  return hip_check_launch();
}

SLM_API int slm_w4_prepack(int32_t format, const int32_t* qweight, const int32_t* qzeros,
                           const void* scales, const int32_t* perm, int64_t K, int64_t N,
                           int64_t group_size, int32_t dtype, void* wq_out, void* sz_out,
                           void* stream) {
  if (!qweight || !qzeros || !scales || !wq_out || !sz_out) return SLM_ERR_INVALID_ARG;
  if (K <= 0 || N <= 0 || K % 64 || N % 32 || group_size <= 0 || K % group_size)
    return SLM_ERR_UNSUPPORTED;
  if (dtype != SLM_F16 && dtype != SLM_BF16) return SLM_ERR_UNSUPPORTED;
  const int rc = slm_w4_prepack_weights(format, qweight, perm, K, N, wq_out, stream);
  if (rc != SLM_OK) return rc;
  return slm_w4_prepack_sz(format, qzeros, scales, K, N, group_size, dtype, sz_out, stream);
}

SLM_API int slm_w4_dequant(const void* wq, const void* sz, int64_t K, int64_t N,
                           int64_t group_size, int32_t dtype, void* w_out, void* stream) {
  if (!wq || !sz || !w_out) return SLM_ERR_INVALID_ARG;
  if (K <= 0 || N <= 0 || K % 64 || N % 32 || group_size < 16 || K % group_size)
    return SLM_ERR_UNSUPPORTED;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hip_clear_error();
  const int64_t words = K * N / 8;
  const dim3 grid((unsigned)((words + 255) / 256)), blk(256);
  if (dtype == SLM_BF16)
    hipLaunchKernelGGL(w4_dequant_kernel<bf16_tag>, grid, blk, 0, st, (const uint32_t*)wq,
                       (const uint32_t*)sz, K, N, group_size, (uint16_t*)w_out);
  else if (dtype == SLM_F16)
    hipLaunchKernelGGL(w4_dequant_kernel<f16_tag>, grid, blk, 0, st, (const uint32_t*)wq,
                       (const uint32_t*)sz, K, N, group_size, (uint16_t*)w_out);
  else
    return SLM_ERR_UNSUPPORTED;
  return hip_check_launch();
}

SLM_API size_t slm_w4a16_gemm_workspace_bytes(const slm_w4_gemm_args* a) {
  GemmPlan pl;
  if (plan_gemm(a, &pl) != SLM_OK) return 0;
  return pl.part_bytes + pl.aperm_bytes;
}

SLM_API int32_t slm_w4a16_gemm_deferred_splits(const slm_w4_gemm_args* a) {
  GemmPlan pl;
  if (plan_gemm(a, &pl) != SLM_OK) return 0;
  return (a->flags & SLM_W4_DEFER_REDUCE) && !a->bias && pl.split_k > 1 ? pl.split_k : 0;
}

}  // extern "C"

// np != NULL: the activations are produced by the GEMV's norm prologue (slm_w4a16_gemv_norm)
static int gemm_impl(const slm_w4_gemm_args* a, const slm_w4_norm_prologue* np, void* stream) {
  GemmPlan pl;
  int rc = plan_gemm(a, &pl);
  if (rc != SLM_OK) return rc;
  if (a->M == 0) return SLM_OK;
  if (np && (!pl.gemv || a->perm || !gemv_supported(a->M, a->K, a->group_size, true)))
    return SLM_ERR_UNSUPPORTED;
  if ((!np && !a->a) || !a->wq || !a->sz || !a->c) return SLM_ERR_INVALID_ARG;
  const bool silu = (a->flags & SLM_W4_SILU_MUL) != 0;
  if (np) {
    // exactly one activation source; the residual is double-buffered: every workgroup recomputes
    // x + residual_in while workgroup 0 stores residual_out, so the two must not share memory
    if (!np->weight || (np->x != nullptr) == (np->partials != nullptr) ||
        (np->partials && np->n_splits < 1) || (np->residual_in && !np->residual_out))
      return SLM_ERR_INVALID_ARG;
    const size_t row_bytes = (size_t)a->M * a->K * 2;
    auto overlaps = [&](const void* w, const void* r) {
      const char* wp = reinterpret_cast<const char*>(w);
      const char* rp = reinterpret_cast<const char*>(r);
      return w && r && wp < rp + row_bytes && rp < wp + row_bytes;
    };
    if (overlaps(np->residual_out, np->residual_in) || overlaps(np->residual_out, np->x) ||
        overlaps(np->normed_out, np->residual_in) || overlaps(np->normed_out, np->x) ||
        overlaps(np->normed_out, np->residual_out))
      return SLM_ERR_INVALID_ARG;
    if (!aligned16(np->x) || !aligned16(np->partials) || !aligned16(np->residual_in) ||
        !aligned16(np->residual_out) || !aligned16(np->weight) || !aligned16(np->normed_out))
      return SLM_ERR_ALIGNMENT;
  }
  // with perm the packed K may exceed the source width (padded act-order shards): perm[k] < lda is
  // the caller's contract then
  if ((!np && (!aligned16(a->a) || a->lda % 8 || (!a->perm && a->lda < a->K))) || !aligned16(a->wq) ||
      a->ldc < (silu ? a->N / 2 : a->N))
    return SLM_ERR_ALIGNMENT;
  if ((pl.part_bytes + pl.aperm_bytes) > 0 &&
      (!a->workspace || a->workspace_bytes < pl.part_bytes + pl.aperm_bytes))
    return SLM_ERR_WORKSPACE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hip_clear_error();

  GemmKParams kp;
  kp.a = a->a; kp.lda = a->lda;
  kp.norm_weight = nullptr;
  if (np) {
    kp.a = nullptr; kp.lda = a->K;
    kp.norm_x = np->x; kp.norm_part = np->partials; kp.norm_splits = np->n_splits;
    kp.norm_eps = np->eps; kp.norm_res_in = np->residual_in; kp.norm_res_out = np->residual_out;
    kp.norm_weight = np->weight; kp.norm_out = np->normed_out;
  }
  if (a->perm) {  // act-order: gather the activation columns once (gptq_gemm.cu:69-118)
    uint16_t* ap = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(a->workspace) + pl.part_bytes);
    const unsigned gx = (unsigned)((a->K + 255) / 256);
    hipLaunchKernelGGL(w4_permute_cols_kernel, dim3(gx > 64 ? 64 : gx, (unsigned)a->M), dim3(256), 0,
                       st, reinterpret_cast<const uint16_t*>(a->a), a->perm, a->M, a->K, a->lda, ap);
    rc = hip_check_launch();
    if (rc != SLM_OK) return rc;
    kp.a = ap; kp.lda = a->K;
  }
  kp.wq = reinterpret_cast<const uint32_t*>(a->wq);
  kp.sz = reinterpret_cast<const uint32_t*>(a->sz);
  kp.bias = a->bias; kp.c = a->c;
  kp.part = pl.split_k > 1 ? reinterpret_cast<float*>(a->workspace) : nullptr;
  kp.M = a->M; kp.K = a->K; kp.N = a->N; kp.ldc = a->ldc;
  // per-channel scales (group_size == K, not necessarily a power of two): every k maps to group 0
  kp.gs_shift = (a->group_size == a->K) ? 30 : ilog2(a->group_size);
  kp.n_chunks = (int)(a->K / W4_KC);
  kp.split_k = pl.split_k; kp.chunks_per_split = pl.chunks_per_split;
  kp.n_mblocks = pl.n_mblocks; kp.n_nblocks = pl.n_nblocks;
  kp.silu = silu ? 1 : 0;
  kp.ks_tpw = pl.ks ? pl.ks_tpw : 0;
  kp.ks_groups = (int)(a->K / a->group_size);
  kp.ks_dbg = tune_get(TUNE_W4_KS_DBG, 0);
  kp.sk_per = pl.sk_per; kp.sk_sync = nullptr; kp.sk_part = nullptr;
  if (pl.xl_sk) {
    kp.sk_part = reinterpret_cast<float*>(a->workspace);
    kp.sk_sync = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(a->workspace) + W4_XL_SK_WGS * W4_XL_SK_SLOT_BYTES);
    // the ticket and the flags start at zero on every call (a memset node under capture)
    if (hipMemsetAsync(kp.sk_sync, 0, W4_XL_SK_SYNC_BYTES, st) != hipSuccess) return hip_check_launch();
    launch_gemm_xl_sk(kp, a->dtype, pl.ng, W4_XL_SK_WGS, st);
  } else
  if (pl.ks)
    launch_gemm_ks(kp, a->dtype, pl.ng, pl.ks_cw, pl.ks_nw, pl.n_nblocks * pl.split_k, st, pl.ks_mt);
  else if (pl.gemv)
    launch_gemv(kp, a->dtype, pl.ng, st);
  else if (pl.small)
    launch_gemm_small(kp, a->dtype, pl.ng, pl.n_nblocks * pl.n_mblocks * pl.split_k, st);
  else if (pl.m128)
    launch_gemm_m128(kp, a->dtype, (int)a->group_size, pl.m128_wd, pl.m128_kw, pl.m128_ct, pl.m128_adma, pl.n_nblocks * pl.split_k, st);
  else if (pl.mt == 16)
    launch_gemm_xl(kp, a->dtype, pl.ng, pl.n_nblocks * pl.n_mblocks * pl.split_k, st);
  else if (pl.mt == 8)
    launch_gemm_ws(kp, a->dtype, pl.ng, pl.n_nblocks * pl.n_mblocks * pl.split_k, st);
  else if (a->dtype == SLM_BF16) launch_gemm<bf16_tag>(kp, pl, st);
  else launch_gemm<f16_tag>(kp, pl, st);
  rc = hip_check_launch();
  if (rc != SLM_OK) return rc;
  if (pl.split_k > 1 && !((a->flags & SLM_W4_DEFER_REDUCE) && !a->bias)) {
    const int64_t n4 = a->M * (silu ? a->N / 2 : a->N) / 4;
    const dim3 grid((unsigned)((n4 + 255) / 256)), blk(256);
    if (silu && a->dtype == SLM_BF16)
      hipLaunchKernelGGL(w4_splitk_reduce_silu_kernel<bf16_tag>, grid, blk, 0, st, kp.part, a->bias,
                         a->c, a->M, a->N, a->ldc, pl.split_k);
    else if (silu)
      hipLaunchKernelGGL(w4_splitk_reduce_silu_kernel<f16_tag>, grid, blk, 0, st, kp.part, a->bias,
                         a->c, a->M, a->N, a->ldc, pl.split_k);
    else if (a->dtype == SLM_BF16)
      hipLaunchKernelGGL(w4_splitk_reduce_kernel<bf16_tag>, grid, blk, 0, st, kp.part, a->bias, a->c,
                         a->M, a->N, a->ldc, pl.split_k);
    else
      hipLaunchKernelGGL(w4_splitk_reduce_kernel<f16_tag>, grid, blk, 0, st, kp.part, a->bias, a->c,
                         a->M, a->N, a->ldc, pl.split_k);
    rc = hip_check_launch();
  }
  return rc;
}

extern "C" {

SLM_API int slm_w4a16_gemm(const slm_w4_gemm_args* a, void* stream) {
  return gemm_impl(a, nullptr, stream);
}

SLM_API int32_t slm_w4a16_gemv_norm_supported(const slm_w4_gemm_args* a) {
  GemmPlan pl;
  if (!a || plan_gemm(a, &pl) != SLM_OK) return 0;
  return pl.gemv && !a->perm && gemv_supported(a->M, a->K, a->group_size, true) ? 1 : 0;
}

SLM_API int slm_w4a16_gemv_norm(const slm_w4_gemm_args* a, const slm_w4_norm_prologue* np,
                                void* stream) {
  if (!a || !np) return SLM_ERR_INVALID_ARG;
  return gemm_impl(a, np, stream);
}

}  // extern "C"
