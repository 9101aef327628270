// EXPERIMENTS, not part of the library: the split-bf16 predict apply on v_mfma_f32_32x32x16_bf16 in two forms --
//   leaf_tiles_bf16w_kernel  two waves per SIMD taking turns (the structure of leaf_tiles_bf16_kernel), and
//   leaf_tiles_bf16s_kernel  ONE wave per SIMD with the generation of the next k-step software-pipelined into the wave's
//                            own MFMA stream (sched_group_barrier),
// both verified bit-identical to each other and within float rounding of the shipped 16x16x32 kernel against the whole
// parity suite, both SLOWER than it at C3 (1.50 / 1.52 ms against 1.20 ms; profiles/r03_predict_experiments.txt has the
// phase stamps and the ablation builds that explain why).  Kept with their packers and launchers so that the record can
// be re-measured: tools/micro/leaf_bf16_phases.hip includes this file (modes "w" and "s").
#pragma once
namespace gpso {
typedef float f32x16 __attribute__((ext_vector_type(16)));

// LDS reads the compiler does not track: issued and waited for by hand.  lds_wait<N> returns once all but the N youngest
// LDS operations of the wave have completed; the registers named are tied to the wait, so no use of them can be
// scheduled in front of it.
template <int OFF>
__device__ __forceinline__ u32x4 lds_read_b128_async(unsigned lds_addr) {
  static_assert(OFF >= 0 && OFF < 65536, "16-bit unsigned offset");
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF) : "memory");
  return v;
}
__device__ __forceinline__ float lds_read_b32_async(unsigned lds_addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(lds_addr) : "memory");
  return v;
}
template <int N>
__device__ __forceinline__ void lds_wait_f(float (&a)[4], float (&b)[4]) {
  static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
  asm volatile("s_waitcnt lgkmcnt(%8)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])
               : "n"(N)
               : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait(u32x4& r0, u32x4& r1) {
  static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(r0), "+v"(r1) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait(u32x4& r0, u32x4& r1, u32x4& r2) {
  static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
  asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(r0), "+v"(r1), "+v"(r2) : "n"(N) : "memory");
}


// the same apply on the 32x32x16 bf16 MFMA ("wide" form, float generation only; predict.hip: leaf_tiles_bf16w_kernel).
// K = 2 contraction steps over the augmented inputs [x~ | norm | 1], zero-padded to a multiple of four steps:
// dpw = 4 ceil((d + 2) / 8); xw = npad / 32 * dpw * 64 floats from launch_gen_inputs_wide; linv_w = nsplit * npad * npad
// bf16 from launch_pack_linv_bf16w (its own order)
inline int leaf_bf16w_dpw(int d) { return (d + 2 + 7) / 8 * 4; }
inline size_t leaf_bf16w_lds_bytes(int nsplit, int dpw) {
  return (size_t)2 * nsplit * 16 * 1024 + (size_t)3 * (dpw * 256 + 256) + (size_t)8 * dpw * 256 + 256;
}
// the same layouts, ONE wave per SIMD with the generation software-pipelined into the wave's own MFMA stream
// (predict.hip: leaf_tiles_bf16s_kernel)
inline size_t leaf_bf16s_lds_bytes(int nsplit, int dpw) {
  return (size_t)2 * nsplit * 16 * 1024 + (size_t)3 * (dpw * 256 + 256) + (size_t)4 * dpw * 256 + 256;
}

// =============================================================================================
// leaf_tiles, split-bf16 apply on the 32x32x16 bf16 MFMA ("wide" form; float generation)
// =============================================================================================
// leaf_tiles_bf16_kernel above is bound by instruction ISSUE, not by the matrix pipe: a v_mfma_f32_16x16x32_bf16
// holds its SIMD's vector issue for 8 of its 16 clocks, and the other wave's generation VALU work can only issue in
// what is left (profiles/r02h_leaf_bf16_phases.txt: per k-step 2 x 3 400 clocks of generation + 384 x 8 of MFMA
// issue = the 9 900 clocks a step takes, for 6 144 clocks of pipe time).  A v_mfma_f32_32x32x16_bf16 does twice the
// work per instruction and holds the issue for 8 of its 32 clocks.  This kernel is the same algorithm on that
// instruction: a wave's 256 rows x 32 leaves are 8 accumulator tiles of 32 x 32 (16 registers each), a k-step of 32
// training points is 2 x 6 MFMAs per tile (96 per wave instead of 192).
// The generated cross-Gram tile must arrive in the B-operand layout of that MFMA -- lane (j = lane % 32, h = lane / 32)
// holds 8 consecutive k-slots of column j -- which is what the 32x32 ACCUMULATOR layout of v_mfma_f32_32x32x2_f32 gives
// when training points are the rows: register r of lane (j, h) is point 8 (r / 4) + 4 h + r % 4 of the step.  K-slot s
// of MFMA m (the step's two K = 16 halves) of lane (., h) is therefore point 16 m + 8 (s / 4) + 4 h + s % 4, and
// pack_linv_bf16w_kernel lays the A fragments out in that order (any k order is valid as long as A and B agree).
// The contraction yields u = C2 r^2 DIRECTLY: the operands are augmented by two columns,
//     A' = [-2 C2 x~ | C2 |x~|^2 | 1],   B' = [x~* | 1 | C2 |x~*|^2],
// so no norm is loaded and no r^2 is combined on the VALU (one extra K = 2 MFMA when D + 2 crosses an even number).
// Float generation only (there is no 32x32 f64 MFMA): posteriors whose self-test asks for double generation run the
// 16x16x32 kernel above.
template <int NS, typename TF>
__global__ __launch_bounds__(256) void pack_linv_bf16w_kernel(const TF* __restrict__ linv, int64_t n, int64_t npad,
                                                              u32x4* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (rt32, q, m, lane)
  const int64_t nt = npad / 32;
  if (idx >= nt * nt * 2 * 64) return;
  const int lane = (int)(idx & 63), m = (int)((idx >> 6) & 1);
  const int64_t q = (idx >> 7) % nt, rt = (idx >> 7) / nt;
  const int64_t row = rt * 32 + (lane & 31);
  const int h = lane >> 5;
  float v[8], lo[8];
#pragma unroll
  for (int sl = 0; sl < 8; ++sl) {
    const int64_t col = q * 32 + 16 * m + 8 * (sl >> 2) + 4 * h + (sl & 3);
    const TF x = (row < n && col <= row) ? linv[row * npad + col] : (TF)0;
    v[sl] = (float)x;
    lo[sl] = (float)(x - (TF)v[sl]);
  }
#pragma unroll
  for (int sp = 0; sp < NS; ++sp) {
    u32x4 f;
#pragma unroll
    for (int hh = 0; hh < 4; ++hh) f[hh] = bf16_split_pair(v[2 * hh], v[2 * hh + 1]);
    if (sp == 0) {
#pragma unroll
      for (int sl = 0; sl < 8; ++sl) v[sl] += lo[sl];
    }
    out[((((int64_t)sp * nt + rt) * nt + q) * 2 + m) * 64 + lane] = f;
  }
}

// augmented, pre-scaled training inputs as 32x32x2 A fragments: xw[(q * dpw + c) * 64 + lane], lane (i, h) = column
// 2 c + h of A' for training point 32 q + i
__global__ __launch_bounds__(256) void gen_inputs_wide_kernel(const double* __restrict__ xs64, int64_t npad, int d,
                                                              int dp, int dpw, float c2, float* __restrict__ xw) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (npad / 32) * dpw * 64) return;
  const int lane = (int)(idx & 63);
  const int c = (int)((idx >> 6) % dpw);
  const int64_t q = (idx >> 6) / dpw;
  const int64_t nrow = 32 * q + (lane & 31);
  const int col = 2 * c + (lane >> 5);
  float v = 0.0f;
  if (col < d) {
    v = (-2.0f * c2) * (float)xs64[nrow * dp + col];
  } else if (col == d) {  // C2 |x~|^2, the float norm summed as gen_inputs_f32_kernel sums it
    float acc = 0.0f;
    for (int k = 0; k < dp; ++k) {
      const float x = (float)xs64[nrow * dp + k];
      acc += x * x;
    }
    v = c2 * acc;
  } else if (col == d + 1) {
    v = 1.0f;
  }
  xw[idx] = v;
}

// the kernel map on u = C2 r^2 as it leaves the matrix pipe: the floor is a bare v_max_f32 (fmaxf would first
// canonicalise a value the compiler cannot prove quiet: one more VALU instruction per entry), and the variance is
// folded into the polynomial's coefficients (sigma^2 (1 + t + t^2 / 3) = fma(t, fma(t, sigma^2 / 3, sigma^2), sigma^2))
template <int KERNEL>
__device__ __forceinline__ float kern_from_scaled_w(float u, float variance) {
  constexpr float kNegLog2e = -1.4426950408889634f;
  if (KERNEL == 3) return variance * __builtin_amdgcn_exp2f(u * (0.5f * kNegLog2e));
  float uc;
  const float floor_u = (float)(KernScale<KERNEL>::C2 * 1e-36);
  asm("v_max_f32 %0, %1, %2" : "=v"(uc) : "v"(u), "v"(floor_u));
  const float t = __builtin_amdgcn_sqrtf(uc);
  const float e = __builtin_amdgcn_exp2f(t * kNegLog2e);
  if (KERNEL == 0) return fmaf(t, fmaf(t, variance * (1.0f / 3.0f), variance), variance) * e;
  if (KERNEL == 1) return fmaf(t, variance, variance) * e;
  return variance * e;
}

// A k-step's generation is two pieces of very different kind.  (1) The CONTRACTION u = A' B'^T: ceil(dpw / 4) x 4 f32
// MFMAs (32x32x2: 64 clocks each).  It is issued by the wave at the head of its own MFMA stream, in front of the apply
// of the step before: an MFMA of the other wave of the SIMD cannot break into a stream of dependent bf16 MFMAs -- with
// the contraction inside the generation phase it waited for the end of the partner's whole apply (3 000 - 3 500 clocks
// for eight MFMAs; tools/micro/leaf_bf16_phases.hip).  (2) Map, mean and bf16 split: vector ALU work only (~500 clocks),
// which does run under the partner's MFMAs.
__device__ __forceinline__ void leaf_bf16w_contract(int lane, int dpw, const unsigned char* xs_b /* [dpw] X fragments */,
                                                    const float* xb, const float* zfrag /* unused */, f32x16& s, int q) {
  (void)q;
  (void)zfrag;
#pragma unroll
  for (int r = 0; r < 16; ++r) s[r] = 0.0f;
  // dpw is a multiple of 4 (the augmented inputs are zero-padded to it).  Operands from LDS in chunks of four K = 2
  // steps, read and waited for by hand, the next chunk's eight reads in flight during the four (dependent, 64-clock)
  // MFMAs of the current one: left to the compiler this loop is one LDS round trip per MFMA.
  unsigned xa = (unsigned)(size_t)(const __attribute__((address_space(3))) void*)xs_b + (unsigned)lane * 4u;
  unsigned xl = (unsigned)(size_t)(const __attribute__((address_space(3))) void*)xb + (unsigned)lane * 4u;
  float ra[4], rb[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    ra[u] = lds_read_b32_async(xa + (unsigned)u * 256u);
    rb[u] = lds_read_b32_async(xl + (unsigned)u * 256u);
  }
  for (int c = 4; c <= dpw; c += 4) {
    lds_wait_f<0>(ra, rb);
    float ca[4], cb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      ca[u] = ra[u];
      cb[u] = rb[u];
    }
    if (c < dpw) {
      xa += 1024u;
      xl += 1024u;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        ra[u] = lds_read_b32_async(xa + (unsigned)u * 256u);
        rb[u] = lds_read_b32_async(xl + (unsigned)u * 256u);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) s = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[u], cb[u], s, 0, 0, 0);
  }
}

template <int NS, int KERNEL>
__device__ __forceinline__ void leaf_bf16w_finish(bool diag, int lane, const float* alp /* the step's 32 alphas */,
                                                  const f32x16& s, float variance, bf16x8 (&bfrag)[NS][2], float& macc) {
  float p[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) p[r] = kern_from_scaled_w<KERNEL>(s[r], variance);
  if (diag) {  // this k-step lies in the diagonal block (wave-uniform): its share of k*.alpha (f32, before the split)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(alp + 8 * g + 4 * (lane >> 5));
#pragma unroll
      for (int e = 0; e < 4; ++e) macc = fma_t(p[4 * g + e], a4[e], macc);
    }
  }
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int sp = 0; sp < NS; ++sp) {
      u32x4 f;
#pragma unroll
      for (int hh = 0; hh < 4; ++hh) f[hh] = bf16_split_pair(p[8 * m + 2 * hh], p[8 * m + 2 * hh + 1]);
      bfrag[sp][m] = __builtin_bit_cast(bf16x8, f);
    }
}

// DIAG = false (k-steps left of the diagonal block: most of them): straight-line code -- with the per-tile skip test of
// the diagonal block in it every row tile is a basic block of its own, and the compiler then waits for ALL outstanding
// LDS reads (the next tile's prefetch included) in front of every MFMA group: the apply ran at 44 clocks per MFMA
// instead of 32 (tools/micro/leaf_bf16_phases.hip, generation ablated).
template <int NS, bool DIAG>
__device__ __forceinline__ void leaf_bf16w_apply(int q, int q_diag0, int lane, const u32x4* panel_b /* [NS][8][2][64] */,
                                                 const bf16x8 (&bfrag)[NS][2], f32x16 (&acc)[8]) {
  constexpr int RT = 8, NT = 2 * RT;
  // A fragments two (row tile, K-half) groups ahead of their use, in a ring of three register sets, read and waited
  // for by hand (lds_read_b128_async / lds_wait): left to the compiler, every second or third MFMA group waits for
  // lgkmcnt(0) -- the prefetches just issued included -- which exposes a full LDS round trip there (44 clocks per MFMA
  // instead of 32 with generation ablated, tools/micro/leaf_bf16_phases.hip).  All reads are waited for before the
  // function returns, so the compiler's own bookkeeping of the LDS counter stays valid outside it.
  const unsigned base = (unsigned)(size_t)(const __attribute__((address_space(3))) void*)panel_b + (unsigned)lane * 16u;
  u32x4 a[3][NS];
  static_for<0, 2>([&](auto t_) {
    constexpr int t = decltype(t_)::value;
    static_for<0, NS>([&](auto sp_) {
      constexpr int sp = decltype(sp_)::value;
      a[t][sp] = lds_read_b128_async<(sp * NT + t) * 1024>(base);
    });
  });
  static_for<0, NT>([&](auto t_) {  // (row tile rt = t / 2, K-half m = t % 2)
    constexpr int t = decltype(t_)::value;
    constexpr int rt = t >> 1, m = t & 1;
    if constexpr (t + 2 < NT) {
      static_for<0, NS>([&](auto sp_) {
        constexpr int sp = decltype(sp_)::value;
        a[(t + 2) % 3][sp] = lds_read_b128_async<(sp * NT + t + 2) * 1024>(base);
      });
    }
    // everything but the reads of the groups t + 1 and t + 2 has landed
    constexpr int kNewer = (NT - 1 - t >= 2) ? 2 * NS : (NT - 1 - t) * NS;
    if constexpr (NS == 3) lds_wait<kNewer>(a[t % 3][0], a[t % 3][1], a[t % 3][2]);
    else lds_wait<kNewer>(a[t % 3][0], a[t % 3][1]);
    __builtin_amdgcn_sched_barrier(0);
    if (!(DIAG && q - q_diag0 > rt)) {  // diagonal block: a tile whose 32 rows all lie above the step's 32 columns is zero
      f32x16 c = acc[rt];
#define GPSO_BFW(SA, SB) \
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[t % 3][SA]), bfrag[SB][m], c, 0, 0, 0)
      if (NS == 3) {
        GPSO_BFW(2, 0);
        GPSO_BFW(0, 2);
        GPSO_BFW(1, 1);
      }
      GPSO_BFW(1, 0);
      GPSO_BFW(0, 1);
      GPSO_BFW(0, 0);
#undef GPSO_BFW
      acc[rt] = c;
    }
  });
}

template <int NS, int KERNEL>
__global__ __launch_bounds__(512, 2) void leaf_tiles_bf16w_kernel(
    const u32x4* __restrict__ linv_w, const float* __restrict__ xw, const float* __restrict__ alpha,
    const float* __restrict__ leaves_s, const float* __restrict__ lnorm, double* __restrict__ part_var,
    double* __restrict__ part_mean, int npad32, int d, int dp, int dpw, int64_t mpad, int nbi, float variance,
    const int64_t* __restrict__ m_live) {
  constexpr int RT = 8, NW = 8;
  constexpr float C2 = (float)KernScale<KERNEL>::C2;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  if (m_live != nullptr && (int64_t)blockIdx.x * (NW * 32) >= *m_live) return;  // workgroup-uniform
  u32x4* panel = reinterpret_cast<u32x4*>(lds_raw);                                     // [2][NS][RT][2][64]
  unsigned char* xsl = reinterpret_cast<unsigned char*>(panel + 2 * NS * RT * 2 * 64);  // [3] X buffers
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xstride = dpw * 256 + 256;  // X fragments | 32 alphas (padded)
  float* xb = reinterpret_cast<float*>(xsl + 3 * xstride) + (size_t)wave * dpw * 64;  // [dpw][64] per wave
  float* zfrag = reinterpret_cast<float*>(xsl + 3 * xstride) + (size_t)NW * dpw * 64;  // 64 zeros (contraction tails)
  if (tid < 64) zfrag[tid] = 0.0f;

  const int bi = nbi - 1 - (int)blockIdx.y;
  const int64_t col0 = ((int64_t)blockIdx.x * NW + wave) * 32;
  const int q_diag0 = bi * RT, q_end = q_diag0 + RT;

  // LDS-DMA duties as in leaf_tiles_bf16_kernel: ONE window (one M0 value) per wave between two workgroup barriers.
  // Fragment f = 8 wave + j of a buffer = (piece sp = wave / 2, row tile 4 (wave % 2) + j / 2, K-half j % 2); the two
  // K-halves of a row tile are adjacent in memory, so four global bases serve the eight fragments.
  constexpr int NPW = NS * 2;
  static_assert(NPW <= NW - 2, "panel waves and input waves are different waves");
  const int lane16 = lane * 16, lane4 = lane * 4;
  const unsigned char* pgb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int sp = wave >> 1, rt = 4 * (wave & 1) + (j >> 1);
    pgb[j] = reinterpret_cast<const unsigned char*>(linv_w + (((size_t)sp * npad32 + (bi * RT + rt)) * npad32 * 2 + (j & 1)) * 64) -
             (j - 4) * 1024;
  }
  auto uniform = [](const unsigned char* p) {
    const unsigned long long g = (unsigned long long)p;
    unsigned lo = (unsigned)g, hi = (unsigned)(g >> 32);
    asm volatile("" : "+s"(lo), "+s"(hi));
    return reinterpret_cast<const unsigned char*>(((unsigned long long)hi << 32) | lo);
  };
  auto issue_panel = [&](int q, int buf) {
    if (wave >= NPW) return;
    unsigned char* centre = reinterpret_cast<unsigned char*>(panel) + buf * (NS * RT * 2 * 1024) + wave * 8192 + 4096;
    static_for<0, 8>([&](auto j_) {
      constexpr int j = decltype(j_)::value;
      glds16_off<(j - 4) * 1024>(uniform(pgb[j] + (size_t)q * 2048) + lane16, centre);
    });
  };
  const unsigned char* xs_bytes = reinterpret_cast<const unsigned char*>(xw);
  const size_t xstep = (size_t)dpw * 256;
  auto issue_x = [&](int q) {
    unsigned char* xd = xsl + (q % 3) * xstride;
    if (wave == NW - 2) {
      const unsigned char* src = uniform(xs_bytes + (size_t)q * xstep + 4096);
      static_for<0, 32>([&](auto r_) {
        constexpr int r = decltype(r_)::value;
        if (r < dpw) glds4_off<(r - 16) * 256>(src + lane4, xd + 4096);
      });
    } else if (wave == NW - 1) {
      glds4_off<0>(uniform(reinterpret_cast<const unsigned char*>(alpha + 32 * q)) + (lane & 31) * 4, xd + (size_t)dpw * 256);
    }
  };

  issue_panel(0, 0);
  issue_x(0);
  if (1 < q_end) {
    if (wave >= NW - 2) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): a second window for the same wave (once per workgroup)
    issue_x(1);
  }
  {  // this wave's leaf fragments: column 2 c + h of B' for leaf col0 + lane % 32
    const int64_t leaf = col0 + (lane & 31);
    const float cn = C2 * lnorm[leaf];
    for (int c = 0; c < dpw; ++c) {
      const int col = 2 * c + (lane >> 5);
      xb[c * 64 + lane] = (col < d) ? leaves_s[leaf * dp + col] : (col == d) ? 1.0f : (col == d + 1) ? cn : 0.0f;
    }
  }
  f32x16 acc[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rt][r] = 0.0f;
  float macc = 0.0f;
  bf16x8 bfrag[NS][2];
  __syncthreads();

  // The interval scheme of leaf_tiles_bf16_kernel -- between barriers k - 1 and k waves 0-3 finish the generation of step
  // k (vector ALU) and then run their MFMA stream, waves 4-7 run their MFMA stream and then finish step k + 1 -- with the
  // MFMA stream of a wave = contraction of step k + 1, then the apply of step k.  One copy of the code: every wave runs
  // finish(q), contract(q + 1), apply(q) for q = 0, 1, ...; only the place of the barrier differs.  The inputs of step
  // q + 1 (issued in interval q - 1) have landed at barrier q - 1, which both halves pass before contract(q + 1).
  const bool ahead = wave >= NW / 2;
  auto issue_for = [&](int k) {
#ifndef GPSO_ABL_NODMA  // (GPSO_ABL_*: ablation builds of tools/micro/leaf_bf16_phases.hip)
    if (k + 1 < q_end) issue_panel(k + 1, (k + 1) & 1);
    if (k + 2 < q_end) issue_x(k + 2);
#endif
  };
  f32x16 s;
  leaf_bf16w_contract(lane, dpw, xsl, xb, zfrag, s, 0);
#ifdef GPSO_ABL_NOFINISH
#pragma unroll
  for (int sp = 0; sp < NS; ++sp)
#pragma unroll
    for (int mm = 0; mm < 2; ++mm) bfrag[sp][mm] = __builtin_bit_cast(bf16x8, u32x4{(unsigned)lane, 1u, 2u, 3u});
#endif
  auto step = [&](int q, auto diagc) {
    constexpr bool kDiag = decltype(diagc)::value;
    GPSO_BSTAMP(q, 0);
    if (!ahead || q == 0) issue_for(q);
    else if (q >= 2) issue_for(q - 1);
    GPSO_BSTAMP(q, 1);
#ifndef GPSO_ABL_NOFINISH
    leaf_bf16w_finish<NS, KERNEL>(kDiag, lane, reinterpret_cast<const float*>(xsl + (q % 3) * xstride + dpw * 256), s,
                                  variance, bfrag, macc);
#else
    macc += s[0] + s[15];
#endif
    GPSO_BSTAMP(q, 2);
    if (ahead && q > 0) __syncthreads();
    GPSO_BSTAMP(q, 3);
#ifndef GPSO_ABL_NOCONTRACT
    if (q + 1 < q_end) leaf_bf16w_contract(lane, dpw, xsl + ((q + 1) % 3) * xstride, xb, zfrag, s, q);
#endif
    GPSO_BSTAMP(q, 6);
#ifndef GPSO_ABL_NOAPPLY
    leaf_bf16w_apply<NS, kDiag>(q, q_diag0, lane, panel + (q & 1) * NS * RT * 2 * 64, bfrag, acc);
#else
    acc[0][0] += __builtin_bit_cast(float, __builtin_bit_cast(u32x4, bfrag[0][0])[0]) + __builtin_bit_cast(float, __builtin_bit_cast(u32x4, bfrag[NS - 1][1])[3]);
#endif
    GPSO_BSTAMP(q, 4);
    if (!ahead) __syncthreads();
    GPSO_BSTAMP(q, 5);
  };
  for (int q = 0; q < q_diag0; ++q) step(q, std::false_type{});
  for (int q = q_diag0; q < q_end; ++q) step(q, std::true_type{});
  if (ahead) __syncthreads();

  double sq = 0;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) sq = fma((double)acc[rt][r], (double)acc[rt][r], sq);
  sq += __shfl_xor(sq, 32);
  double mm = (double)macc;
  mm += __shfl_xor(mm, 32);
  if (lane < 32) {
    const int64_t col = col0 + lane;
    part_var[(int64_t)bi * mpad + col] = sq;
    part_mean[(int64_t)bi * mpad + col] = mm;
  }
}

template <int NS>
static int launch_leaf_tiles_bf16w_ns(hipStream_t st, const void* linv_w, const float* xw, const float* alpha,
                                      const float* leaves_s, const float* lnorm, double* part_var, double* part_mean,
                                      int64_t npad, int d, int dp, int64_t mpad, const KernParams& kp,
                                      const int64_t* m_live) {
  const int nbi = (int)(npad / 256), dpw = leaf_bf16w_dpw(d);
  const dim3 grid((unsigned)(mpad / 256), (unsigned)nbi);
  const size_t lds = leaf_bf16w_lds_bytes(NS, dpw);
  if (dpw > 32) {  // the X fragments of a k-step are one DMA window of 32 pieces
    note_launch_error("launch_leaf_tiles_bf16w: more than 32 X pieces per k-step");
    return 1;
  }
#define GPSO_L(K)                                                                                      \
  do {                                                                                                 \
    const int rc = ensure_dyn_lds((const void*)leaf_tiles_bf16w_kernel<NS, K>, (int)lds);              \
    if (rc) return rc;                                                                                 \
    hipLaunchKernelGGL((leaf_tiles_bf16w_kernel<NS, K>), grid, dim3(512), lds, st,                     \
                       static_cast<const u32x4*>(linv_w), xw, alpha, leaves_s, lnorm, part_var,        \
                       part_mean, (int)(npad / 32), d, dp, dpw, mpad, nbi, (float)kp.variance, m_live); \
  } while (0)
  switch (kp.kernel) {
    case 0: GPSO_L(0); break;
    case 1: GPSO_L(1); break;
    case 2: GPSO_L(2); break;
    default: GPSO_L(3); break;
  }
#undef GPSO_L
  return 0;
}

int launch_leaf_tiles_bf16w(hipStream_t st, int nsplit, const void* linv_w, const float* xw, const float* alpha,
                            const float* leaves_s, const float* lnorm, double* part_var, double* part_mean,
                            int64_t npad, int d, int dp, int64_t mpad, const KernParams& kp, const int64_t* m_live) {
  if (nsplit == 3)
    return launch_leaf_tiles_bf16w_ns<3>(st, linv_w, xw, alpha, leaves_s, lnorm, part_var, part_mean, npad, d, dp, mpad, kp, m_live);
  return launch_leaf_tiles_bf16w_ns<2>(st, linv_w, xw, alpha, leaves_s, lnorm, part_var, part_mean, npad, d, dp, mpad, kp, m_live);
}

// =============================================================================================
// leaf_tiles, split-bf16 apply, ONE wave per SIMD with the generation software-pipelined into the wave's own
// instruction stream ("stream" form; float generation)
// =============================================================================================
// What the two kernels above taught (tools/micro/leaf_bf16_phases.hip, ablation builds, C3): the apply ALONE runs the
// matrix pipe at its rate -- and everything the OTHER wave of the SIMD does while it runs is paid in full on top:
// vector ALU work costs 2-3x its own issue time (each instruction of the partner that holds the issue port when a
// dependency-paced MFMA becomes ready opens a bubble in the pipe), an MFMA of the partner (the contraction) waits for
// the end of the whole stream.  Two waves per SIMD overlap nothing here.  Inside ONE wave's in-order stream, though,
// up to ~5 vector / scalar / LDS instructions fit into the 32 clocks of a v_mfma_f32_32x32x16_bf16 for free
// (MI355X_MICROARCH.md, "single-issue instructions HIDDEN per gap").  So: one wave per SIMD (4 per workgroup, 512
// registers each), a wave tile of 256 rows x 64 leaves (16 accumulator tiles of 32 x 32: every A fragment read from
// LDS feeds two column tiles), and per k-step of 32 training points ONE stream per wave:
//     contraction of step q + 1 (f32 MFMAs)  |  16 groups of 12 bf16 MFMAs (apply of step q), each carrying a slice of
//     the map / mean / bf16 split of step q + 1 (2-3 vector instructions per MFMA, placed by sched_group_barrier),
//     three hand-issued LDS reads of the A fragments two groups ahead, and one or two LDS-DMA instructions
//     (panel of step q + 1, inputs of step q + 2).
// One workgroup barrier per k-step (the panel buffers swap).  Layouts are those of leaf_tiles_bf16w_kernel.
__device__ __forceinline__ void dma16_m0(unsigned lds_addr /* uniform */, const void* gbase /* uniform */, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(gbase) : "memory");
}
__device__ __forceinline__ void dma4_m0(unsigned lds_addr /* uniform */, const void* gbase /* uniform */, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(lds_addr), "v"(voff), "s"(gbase) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait_v(f32x4& r0, f32x4& r1, f32x4& r2, f32x4& r3) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "n"(N) : "memory");
}
__device__ __forceinline__ f32x4 lds_read_f4_async(unsigned lds_addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(lds_addr) : "memory");
  return v;
}

// The vector work that turns the contraction of a step (sn: u = C2 r^2 of 32 points x 32 leaves) into B fragments rides
// on the 16 MFMA groups of a stream in slices.  There is ONE set of B fragments (bc[piece][K-half]) -- registers decide:
// accumulators 128 + A-fragment ring 36 + B fragments 24 + contraction 16 + ... must stay within the 256 architectural
// VGPRs (the MFMAs are the compiler's, in their VGPR form: a second B set, or accumulators in AGPRs, made hipcc shuttle
// tiles between the two register files with ~5 v_accvgpr moves per MFMA).  A stream applies the K-half 0 of all eight
// row tiles first (groups 0-7), then the K-half 1 (groups 8-15); so
//   groups 0-7   (beside the K-half-0 MFMAs of step q):  split of the K-half 1 of step q itself -- from `carry`, the
//                kernel values of its points 8-15 / 24-31, kept from the previous stream -- into bc[..][1];
//                kernel map (+ mean) of step q + 1, entries 0-7;
//   groups 8-15  (beside the K-half-1 MFMAs): split of the K-half 0 of step q + 1 into bc[..][0] (its readers are
//                done); kernel map (+ mean) of step q + 1, entries 8-15, straight into `carry`.
template <int NS, int KERNEL>
__global__ __launch_bounds__(256, 1) void leaf_tiles_bf16s_kernel(
    const u32x4* __restrict__ linv_w, const float* __restrict__ xw, const float* __restrict__ alpha,
    const float* __restrict__ leaves_s, const float* __restrict__ lnorm, double* __restrict__ part_var,
    double* __restrict__ part_mean, int npad32, int d, int dp, int dpw, int64_t mpad, int nbi, float variance,
    const int64_t* __restrict__ m_live) {
  constexpr int RT = 8, NT = 2 * RT, NW = 4, NF = NS * NT;  // fragments per panel buffer
  constexpr float C2 = (float)KernScale<KERNEL>::C2;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  if (m_live != nullptr && (int64_t)blockIdx.x * (NW * 32) >= *m_live) return;  // workgroup-uniform
  u32x4* panel = reinterpret_cast<u32x4*>(lds_raw);                            // [2][NS][RT][2][64]
  unsigned char* xsl = reinterpret_cast<unsigned char*>(panel + 2 * NF * 64);  // [3] X buffers
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xstride = dpw * 256 + 256;  // X fragments | 32 alphas (padded)
  float* xb = reinterpret_cast<float*>(xsl + 3 * xstride) + (size_t)wave * dpw * 64;   // [dpw][64] per wave
  float* zfrag = nullptr;

  const int bi = nbi - 1 - (int)blockIdx.y;
  const int64_t col0 = ((int64_t)blockIdx.x * NW + wave) * 32;
  const int q_diag0 = bi * RT, q_end = q_diag0 + RT;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds_raw;
  const unsigned xsl0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)xsl;

  // ---- LDS-DMA duties of this wave: of every piece the row tiles 2 wave and 2 wave + 1 (both K-halves: 2 KB per row
  // tile and k-step in memory) = 4 NS fragments; pieces wave, wave + 4, ... of the X fragments; wave 0 the alphas
  const size_t s_rt = (size_t)npad32 * 2048, s_sp = (size_t)npad32 * s_rt;
  const unsigned char* lw = reinterpret_cast<const unsigned char*>(linv_w) + (size_t)(bi * RT + 2 * wave) * s_rt;
  const unsigned lane16 = (unsigned)lane * 16u, lane4 = (unsigned)lane * 4u;
  auto dma_panel = [&](auto j_, int q) {  // j < 4 NS: piece j / 4, row tile 2 wave + (j % 4) / 2, K-half j % 2
    constexpr int j = decltype(j_)::value, sp = j >> 2, jr = (j >> 1) & 1, jm = j & 1;
    const unsigned char* g = lw + (size_t)sp * s_sp + (size_t)jr * s_rt + (size_t)q * 2048 + jm * 1024;
    dma16_m0(lds0 + (unsigned)((q & 1) * NF + sp * NT + 4 * wave + 2 * jr + jm) * 1024u, g, lane16);
  };
  auto dma_x = [&](int j, int q) {  // 1 KB piece r = wave + 4 j of the step's X fragments (dpw / 4 pieces: j < 2)
    const int r = wave + 4 * j;
    if (4 * r < dpw) dma16_m0(xsl0 + (unsigned)((q % 3) * xstride + r * 1024), reinterpret_cast<const unsigned char*>(xw) + ((size_t)q * dpw + 4 * r) * 256, lane16);
  };
  auto dma_alpha = [&](int q) {
    if (wave == 0) dma4_m0(xsl0 + (unsigned)((q % 3) * xstride + dpw * 256), alpha + 32 * (size_t)q, ((unsigned)lane & 31u) * 4u);
  };
  // the inputs of step q + 2 (the panel of step q + 1 goes out one fragment per MFMA group, see the stream)
  auto dma_inputs = [&](int q) {
    dma_x(0, q);
    dma_x(1, q);
    dma_alpha(q);
  };

  // ---- prologue: panel(0), inputs of steps 0 and 1, this wave's leaf fragments --------------------------------
  static_for<0, 4 * NS>([&](auto j_) { dma_panel(j_, 0); });
  for (int qq = 0; qq < 2 && qq < q_end; ++qq) dma_inputs(qq);
  {
    const int64_t leaf = col0 + (lane & 31);
    const float cn = C2 * lnorm[leaf];
    for (int c = 0; c < dpw; ++c) {
      const int col = 2 * c + (lane >> 5);
      xb[c * 64 + lane] = (col < d) ? leaves_s[leaf * dp + col] : (col == d) ? 1.0f : (col == d + 1) ? cn : 0.0f;
    }
  }
  f32x16 acc[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rt][r] = 0.0f;
  float macc = 0.0f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  u32x4 bc[NS][2];  // B fragments [piece][K-half]
  float kv[16];     // kernel values: entries 0-7 of the step whose K-half 0 is split next, entries 8-15 of the step before
                    // it until its K-half 1 has been split (`carry` above)
  f32x16 sn;        // contraction of the next step
  f32x4 al[4];
  auto read_alpha = [&](int q) {  // this lane's 16 alphas of step q: points 8 g + 4 h + e
    const unsigned a0 = xsl0 + (unsigned)((q % 3) * xstride + dpw * 256) + ((unsigned)lane >> 5) * 16u;
#pragma unroll
    for (int g = 0; g < 4; ++g) al[g] = lds_read_f4_async(a0 + 32u * g);
  };
  // kernel map of entry E (+ the mean's share of four entries once they are complete)
  auto map_entry = [&](auto e_, auto diagc) {
    constexpr int E = decltype(e_)::value;
    constexpr bool kD = decltype(diagc)::value;
    // (the empty asm statements pin the slice between the LDS reads of its group and those of the next: plain
    // arithmetic has no side effects, and the compiler otherwise collects the map of all sixteen entries in one place)
    float u = sn[E];
    asm volatile("" : "+v"(u));
    float k = kern_from_scaled_w<KERNEL>(u, variance);
    asm volatile("" : "+v"(k));
    kv[E] = k;
    if constexpr (kD && (E & 3) == 3) {
      constexpr int g = E >> 2;
#pragma unroll
      for (int e = 0; e < 4; ++e) macc = fma_t(kv[4 * g + e], al[g][e], macc);
      asm volatile("" : "+v"(macc));
    }
  };
  auto split_pair = [&](auto pr_, auto m_) {  // pair pr of K-half m -> dword pr of bc[..][m]
    constexpr int pr = decltype(pr_)::value, m = decltype(m_)::value;
    float a = kv[8 * m + 2 * pr], b = kv[8 * m + 2 * pr + 1];
    asm volatile("" : "+v"(a), "+v"(b));
#pragma unroll
    for (int sp = 0; sp < NS; ++sp) {
      unsigned w = bf16_split_pair(a, b);
      asm volatile("" : "+v"(w));
      bc[sp][m][pr] = w;
    }
  };
  const std::integral_constant<int, 0> half0;
  const std::integral_constant<int, 1> half1;

  // step 0's generation, not overlapped with anything
  read_alpha(0);
  leaf_bf16w_contract(lane, dpw, xsl, xb, zfrag, sn, 0);  // (ends with lgkmcnt(0): the alphas have landed too)
  lds_wait_v<0>(al[0], al[1], al[2], al[3]);
  static_for<0, 16>([&](auto e_) {
    if (q_diag0 == 0) map_entry(e_, std::true_type{});
    else map_entry(e_, std::false_type{});
  });
  static_for<0, 4>([&](auto pr_) { split_pair(pr_, half0); });

  // kDiag: step q lies in the diagonal block (tiles above the diagonal are zero); kNDiag: step q + 1 does (its generation
  // accumulates the mean); kMore / kMore2: steps q + 1 / q + 2 exist.  All compile-time: a run-time test inside a group
  // splits it into basic blocks, and neither the MFMA / vector interleave nor the contraction's registers survive that
  auto step = [&](int q, auto diagc, auto ndiagc, auto more_, auto more2_) {
    constexpr bool kDiag = decltype(diagc)::value, kMore = decltype(more_)::value;
    using NDiag = decltype(ndiagc);
    const unsigned pb = lds0 + (unsigned)((q & 1) * NF) * 1024u + lane16;
    GPSO_BSTAMP(q, 0);
    // group order: K-half 0 of the row tiles 0..7, then K-half 1; LDS fragment of group g: 2 (g % 8) + g / 8
    u32x4 a[3][NS];
    static_for<0, 2>([&](auto g_) {
      constexpr int g = decltype(g_)::value, fr = 2 * (g % RT) + g / RT;
      static_for<0, NS>([&](auto sp_) {
        constexpr int sp = decltype(sp_)::value;
        a[g][sp] = lds_read_b128_async<(sp * NT + fr) * 1024>(pb);
      });
    });
    if constexpr (kMore) {
      read_alpha(q + 1);
      leaf_bf16w_contract(lane, dpw, xsl + ((q + 1) % 3) * xstride, xb, zfrag, sn, q);
      lds_wait_v<0>(al[0], al[1], al[2], al[3]);
    }
    GPSO_BSTAMP(q, 1);
    static_for<0, NT>([&](auto g_) {
      constexpr int g = decltype(g_)::value;
      constexpr int rt = g % RT, m = g / RT;
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (g + 2 < NT) {
        constexpr int fr = 2 * ((g + 2) % RT) + (g + 2) / RT;
        static_for<0, NS>([&](auto sp_) {
          constexpr int sp = decltype(sp_)::value;
          a[(g + 2) % 3][sp] = lds_read_b128_async<(sp * NT + fr) * 1024>(pb);
        });
      }
      constexpr int kNewer = (NT - 1 - g >= 2) ? 2 * NS : (NT - 1 - g) * NS;
      if constexpr (NS == 3) lds_wait<kNewer>(a[g % 3][0], a[g % 3][1], a[g % 3][2]);
      else lds_wait<kNewer>(a[g % 3][0], a[g % 3][1]);
      // the slice first in program order (its inputs are ready, its results are needed later): the scheduler deals its
      // instructions between the MFMAs below
      if constexpr (m == 0) {
        if constexpr ((g & 1) == 0) split_pair(std::integral_constant<int, g / 2>{}, half1);  // this step's K-half 1: needed from group 8 on
        if constexpr (kMore) map_entry(std::integral_constant<int, g>{}, NDiag{});
      } else if constexpr (kMore) {
        if constexpr ((g & 1) == 0) split_pair(std::integral_constant<int, (g - RT) / 2>{}, half0);
        map_entry(std::integral_constant<int, g>{}, NDiag{});
      }
      if (!(kDiag && q - q_diag0 > rt)) {
        f32x16 c = acc[rt];
#define GPSO_BFS(SA, SB) \
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[g % 3][SA]), __builtin_bit_cast(bf16x8, bc[SB][m]), c, 0, 0, 0)
        if (NS == 3) {
          GPSO_BFS(2, 0);
          GPSO_BFS(0, 2);
          GPSO_BFS(1, 1);
        }
        GPSO_BFS(1, 0);
        GPSO_BFS(0, 1);
        GPSO_BFS(0, 0);
#undef GPSO_BFS
        acc[rt] = c;
      }
      if constexpr (!kDiag) {  // one MFMA, then up to three of the slice's vector instructions, and so on
#pragma unroll
        for (int i = 0; i < (NS == 3 ? 6 : 3); ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // one fragment of the next step's panel per group, from the first group on: the longest possible flight
      if constexpr (kMore && g < 4 * NS) dma_panel(g_, q + 1);
      if constexpr (g == RT - 1) GPSO_BSTAMP(q, 2);
    });
    GPSO_BSTAMP(q, 3);
    if constexpr (decltype(more2_)::value) dma_inputs(q + 2);
    GPSO_BSTAMP(q, 4);
    // the DMAs of this step have landed on every wave and every wave is done with the panel buffer: swap
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GPSO_BSTAMP(q, 5);
    __syncthreads();
  };
  const std::false_type no;
  const std::true_type yes;
  // steps [0, q_diag0 - 1): left of the diagonal block; q_diag0 - 1: the one whose generation is already the block's;
  // then the block's eight, the last two of which have no step q + 2 / q + 1 to prepare
  for (int q = 0; q + 1 < q_diag0; ++q) step(q, no, no, yes, yes);
  if (q_diag0 > 0) step(q_diag0 - 1, no, yes, yes, yes);
  for (int q = q_diag0; q + 2 < q_end; ++q) step(q, yes, yes, yes, yes);
  step(q_end - 2, yes, yes, yes, no);
  step(q_end - 1, yes, yes, no, no);

  double sq = 0;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) sq = fma((double)acc[rt][r], (double)acc[rt][r], sq);
  sq += __shfl_xor(sq, 32);
  double mm = (double)macc;
  mm += __shfl_xor(mm, 32);
  if (lane < 32) {
    const int64_t col = col0 + lane;
    part_var[(int64_t)bi * mpad + col] = sq;
    part_mean[(int64_t)bi * mpad + col] = mm;
  }
}

inline size_t leaf_bf16s_lds_bytes_impl(int nsplit, int dpw) {
  return (size_t)2 * nsplit * 16 * 1024 + (size_t)3 * (dpw * 256 + 256) + (size_t)4 * dpw * 256 + 256;
}

template <int NS>
static int launch_leaf_tiles_bf16s_ns(hipStream_t st, const void* linv_w, const float* xw, const float* alpha,
                                      const float* leaves_s, const float* lnorm, double* part_var, double* part_mean,
                                      int64_t npad, int d, int dp, int64_t mpad, const KernParams& kp,
                                      const int64_t* m_live) {
  const int nbi = (int)(npad / 256), dpw = leaf_bf16w_dpw(d);
  const dim3 grid((unsigned)(mpad / 128), (unsigned)nbi);  // 4 waves x 32 leaves per workgroup
  const size_t lds = leaf_bf16s_lds_bytes_impl(NS, dpw);
  if (dpw > 32) {
    note_launch_error("launch_leaf_tiles_bf16s: more than 32 X pieces per k-step");
    return 1;
  }
#define GPSO_L(K)                                                                                      \
  do {                                                                                                 \
    const int rc = ensure_dyn_lds((const void*)leaf_tiles_bf16s_kernel<NS, K>, (int)lds);              \
    if (rc) return rc;                                                                                 \
    hipLaunchKernelGGL((leaf_tiles_bf16s_kernel<NS, K>), grid, dim3(256), lds, st,                     \
                       static_cast<const u32x4*>(linv_w), xw, alpha, leaves_s, lnorm, part_var,        \
                       part_mean, (int)(npad / 32), d, dp, dpw, mpad, nbi, (float)kp.variance, m_live); \
  } while (0)
  switch (kp.kernel) {
    case 0: GPSO_L(0); break;
    case 1: GPSO_L(1); break;
    case 2: GPSO_L(2); break;
    default: GPSO_L(3); break;
  }
#undef GPSO_L
  return 0;
}

int launch_leaf_tiles_bf16s(hipStream_t st, int nsplit, const void* linv_w, const float* xw, const float* alpha,
                            const float* leaves_s, const float* lnorm, double* part_var, double* part_mean,
                            int64_t npad, int d, int dp, int64_t mpad, const KernParams& kp, const int64_t* m_live) {
  if (nsplit == 3)
    return launch_leaf_tiles_bf16s_ns<3>(st, linv_w, xw, alpha, leaves_s, lnorm, part_var, part_mean, npad, d, dp, mpad, kp, m_live);
  return launch_leaf_tiles_bf16s_ns<2>(st, linv_w, xw, alpha, leaves_s, lnorm, part_var, part_mean, npad, d, dp, mpad, kp, m_live);
}

template <typename TF>
void launch_pack_linv_bf16w(hipStream_t st, int nsplit, const TF* linv, int64_t n, int64_t npad, void* linv_w) {
  const int64_t total = (npad / 32) * (npad / 32) * 2 * 64;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (nsplit == 3)
    hipLaunchKernelGGL((pack_linv_bf16w_kernel<3, TF>), grid, dim3(256), 0, st, linv, n, npad, static_cast<u32x4*>(linv_w));
  else
    hipLaunchKernelGGL((pack_linv_bf16w_kernel<2, TF>), grid, dim3(256), 0, st, linv, n, npad, static_cast<u32x4*>(linv_w));
}

void launch_gen_inputs_wide(hipStream_t st, const double* xs64, int64_t npad, int d, int dp, int kernel, float* xw) {
  const int dpw = leaf_bf16w_dpw(d);
  const float c2 = kernel == 0 ? 5.0f : kernel == 1 ? 3.0f : 1.0f;
  const int64_t total = (npad / 32) * dpw * 64;
  hipLaunchKernelGGL(gen_inputs_wide_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, xs64, npad, d, dp,
                     dpw, c2, xw);
}

}  // namespace gpso
