// stream_timed.hip -- GENERATED from scalellm_amd/csrc/w4_stream.hip: s_memtime stamps around the
// phases of one chunk iteration (A: fragment reads + weight wait + unpack + MFMA issue; B: X reads +
// group epilogue; C: weight refill issue; D: everything between two chunks = stage store + barrier +
// loop overhead), summed over the chunks of ONE wave (wave 1 of the middle workgroup).
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "w4_common.h"
namespace slm {
__device__ __forceinline__ unsigned long long clk() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}
constexpr int ST_RING = 4;
constexpr int ST_WRING = 8;
constexpr int ST_TILE_BYTES = 32 * 256;
template <typename T> struct StOnes;
template <> struct StOnes<bf16_tag> { static constexpr uint32_t bits = 0x3F803F80u; };
template <> struct StOnes<f16_tag> { static constexpr uint32_t bits = 0x3C003C00u; };
template <typename T, int NG, bool SPAN, int NTW>
__global__ void __launch_bounds__(256, 2) stream_timed_kernel(const GemmKParams p, unsigned long long* dbg) {
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
  const char* abase = reinterpret_cast<const char*>(p.a);
  const char* a_src[2];
  int a_dst[2], x_dst[2];
  bool x_wr[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + 256 * i;
    const int row = idx >> 4, slot = idx & 15;
    const int64_t mc = row < p.M ? row : p.M - 1;  // rows >= M: clamped loads, never stored
    a_src[i] = abase + 2 * (mc * p.lda + slot * 8);
    a_dst[i] = row * 256 + ((slot ^ (row & 15)) << 4);
    x_dst[i] = (slot / (16 / NG)) * 32 + row;       // xs[group][row]
    x_wr[i] = (slot & (16 / NG - 1)) == 0;
  }
  // A of one STAGE = ST_RING chunks (32 rows x 512 k = 32 KiB): all 2 * ST_RING loads of stage s+1 are
  // issued at the START of stage s -- before stage s' weight refills in program order, so that by the
  // end of stage s (VMEM completes in order) waiting for them waits for nothing but loads the stage
  // has consumed anyway -- and written to LDS at its end, followed by the ONE barrier of the stage.
  u32x4 areg[ST_RING][2];
  auto a_load_stage = [&](int cfirst) {
#pragma unroll
    for (int ch = 0; ch < ST_RING; ++ch) {
      const uint32_t off = (uint32_t)clampc(cfirst + ch) * 256u;  // < 2 GiB: checked on the host
#pragma unroll
      for (int i = 0; i < 2; ++i) areg[ch][i] = *reinterpret_cast<const u32x4*>(a_src[i] + off);
    }
  };
  uint32_t ones_v = StOnes<T>::bits;
  asm volatile("" : "+v"(ones_v));
  auto a_store_stage = [&](int buf) {
#pragma unroll
    for (int ch = 0; ch < ST_RING; ++ch) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const u32x4 v = areg[ch][i];
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
  const char* wlane[NTW];
  const char* szlane[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    wlane[t] = reinterpret_cast<const char*>(p.wq + (nt[t] * 64 + lane) * 4);
    szlane[t] = reinterpret_cast<const char*>(p.sz + nt[t] * 32 + (lane & 31));
  }
  const uint32_t wstride = (uint32_t)(n_tiles * 1024);  // bytes per 64-deep half chunk
  const uint32_t szstride = (uint32_t)(p.N * 4);        // bytes per scale group
  const int cpg_shift = p.gs_shift >= 30 ? 30 : (p.gs_shift > 7 ? p.gs_shift - 7 : 0);  // log2(chunks per group)
  auto w_load = [&](int c, u32x4 (&w)[NTW][2], uint32_t (&sz)[NTW][NG]) {
    const uint32_t cc = (uint32_t)clampc(c);
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
        w[t][h] = __builtin_nontemporal_load(
            reinterpret_cast<const u32x4*>(wlane[t] + (cc * 2 + h) * wstride));
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const uint32_t grp = NG > 1 ? cc * NG + g : (cc >> cpg_shift);
        sz[t][g] = *reinterpret_cast<const uint32_t*>(szlane[t] + grp * szstride);
      }
    }
  };

  // prologue: A of stage 0, then the weight ring (chunks 0..3), then A of stage 0 -> LDS
  a_load_stage(c0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int d = 0; d < ST_WRING; ++d) {
    w_load(c0 + d, wreg[d], szreg[d]);
    __builtin_amdgcn_sched_barrier(0);
  }
  a_store_stage(0);

  // NCH independent MFMA accumulation chains per tile (k-step j feeds chain j % NCH): an MFMA
  // whose accumulator operand was written by the PREVIOUS MFMA must wait for that result to be
  // written back unless the two issue back to back -- and here 8-16 unpack instructions sit
  // between them.  Measured (profiles/r02_*): one chain costs ~1250 cycles per 128-deep chunk
  // whatever else the kernel does; with 4 chains the dependent distance is 4 MFMAs.
  constexpr int NCH = WPG < 4 ? WPG : 4;
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

  unsigned long long tA = 0, tB = 0, tC = 0, tD = 0, t3 = 0, t2s = 0, t_prev = 0; bool have_prev = false; int n_ch = 0;
  bool group_open = false;  // tmp / xacc hold a partial group (groups wider than a chunk)
  const int n_stage = (nC + ST_RING - 1) / ST_RING;
  for (int stg2 = 0; stg2 < n_stage; stg2 += 2) {
#pragma unroll
   for (int buf = 0; buf < 2; ++buf) {  // two stages per trip: LDS buffer and ring slots are static
    const int stg = stg2 + buf;
    a_load_stage(c0 + (stg + 1) * ST_RING);  // next stage's activations (clamped past the end)
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
        const unsigned long long t0 = clk();
        if (have_prev) tD += t0 - t_prev;
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
          unsigned long long t2 = 0;
          if (g_last) { t2 = clk(); tA += t2 - t0; t2s = t2; }
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
        t3 = clk(); tB += t3 - t2s;
      }
      // refills AFTER the old values are consumed (pinned): each ring slot keeps its registers
      __builtin_amdgcn_sched_barrier(0);
      w_load(c0 + i + ST_WRING, wreg[u], szreg[u]);
      __builtin_amdgcn_sched_barrier(0);
      { const unsigned long long t4 = clk(); if (i < nC) { tC += t4 - t3; t_prev = t4; have_prev = true; ++n_ch; } }
    }
    // next stage -> the buffer everybody finished reading one barrier ago
    a_store_stage(buf ^ 1);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
   }
  }

  if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 64) { dbg[0] = tA; dbg[1] = tB; dbg[2] = tC; dbg[3] = tD; dbg[4] = n_ch; }
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


}  // namespace slm
using namespace slm;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
int main() {
  struct Shape { const char* name; int K, N; };
  const Shape shapes[] = {{"qkv", 4096, 6144}, {"gate_up", 4096, 28672}};
  const int M = 32;
  unsigned long long* dbg; CHECK(hipMalloc(&dbg, 64)); 
  for (const Shape& s : shapes) {
    const size_t bytes = (size_t)s.K * s.N / 2;
    const int n_rot = (int)((400u << 20) / bytes) + 1;
    std::vector<uint32_t*> wq(n_rot);
    for (auto& b : wq) { CHECK(hipMalloc(&b, bytes)); CHECK(hipMemset(b, 0x5a, bytes)); }
    uint32_t* sz; CHECK(hipMalloc(&sz, (size_t)s.K / 128 * s.N * 4)); CHECK(hipMemset(sz, 0x3c, (size_t)s.K / 128 * s.N * 4));
    void *a, *c; CHECK(hipMalloc(&a, (size_t)M * s.K * 2)); CHECK(hipMemset(a, 0x3c, (size_t)M * s.K * 2));
    CHECK(hipMalloc(&c, (size_t)M * s.N * 2));
    GemmKParams p{};
    p.a = a; p.sz = sz; p.bias = nullptr; p.c = c; p.part = nullptr;
    p.M = M; p.K = s.K; p.N = s.N; p.lda = s.K; p.ldc = s.N; p.gs_shift = 7; p.n_chunks = s.K / 128;
    p.split_k = 1; p.chunks_per_split = p.n_chunks; p.n_mblocks = 1; p.n_nblocks = (s.N + 127) / 128;
    constexpr size_t lds = 2 * ST_RING * ST_TILE_BYTES + 2 * ST_RING * 32 * sizeof(float);
    auto kfn = stream_timed_kernel<bf16_tag, 1, false, 1>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) {
      CHECK(hipEventRecord(e0));
      for (auto w : wq) { p.wq = w; hipLaunchKernelGGL(kfn, dim3(p.n_nblocks), dim3(256), lds, 0, p, dbg); }
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long h[5]; CHECK(hipMemcpy(h, dbg, 40, hipMemcpyDeviceToHost));
      const double n = (double)h[4];
      printf("{\"probe\": \"stream_timed\", \"shape\": \"%s\", \"us_per_launch\": %.2f, \"chunks\": %.0f, \"ticks_per_chunk\": {\"A_wait_unpack_mfma\": %.0f, \"B_epilogue\": %.0f, \"C_refill\": %.0f, \"D_between\": %.0f}}\n",
             s.name, ms * 1e3 / wq.size(), n, h[0] / n, h[1] / n, h[2] / n, h[3] / n);
    }
    for (auto& b : wq) CHECK(hipFree(b));
    CHECK(hipFree(sz)); CHECK(hipFree(a)); CHECK(hipFree(c));
  }
  return 0;
}
