// Dense tile GEMM on the gfx950 matrix cores: the workhorse behind the blocked Cholesky
// (syrk / trsm / gemm updates), the triangular solves, K^-1 and the posterior variance.
//
//   C[m x n] = alpha * opA(A) * opB(B)^T + beta * C        (whole TS x TS tiles, TS = 128 or 64)
//
// fp64 uses v_mfma_f64_16x16x4_f64, fp32 uses v_mfma_f32_32x32x2_f32 (exact f32).  A workgroup is
// WM x 2 waves; each wave owns a (TS/WM) x (TS/2) sub-tile.  The k-loop stages TS rows x 128 bytes of
// each operand through LDS with a register-staged software pipeline (one barrier per k-tile, two LDS
// buffers).  Operands may be stored k-contiguous or k-strided; the LDS image keeps the global
// orientation and only the fragment reads differ.  Large grids are walked in 64-tile super-tiles
// interleaved over the 8 XCDs; small grids (which cannot fill 256 CUs with 128 x 128 tiles) use
// 64 x 64 tiles for 4x the parallelism.
#include "gpk_internal.h"

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int LDS_N_STRIDE = 144;    // bytes per row of a k-contiguous tile (128 + 16 pad)

template <typename T> struct Cfg;
template <> struct Cfg<double> { static constexpr int BK = 16; };   // k elements per k-tile (128 bytes)
template <> struct Cfg<float> { static constexpr int BK = 32; };

struct KParams {
  const char* A;
  const char* B;
  char* C;
  long long lda, ldb, ldc;  // in elements
  long long sA, sB, sC;     // byte strides between the nb1 products of an explicit batch (GemmArgs::nbatch) ...
  long long hA, hB, hC;     // ... and between the problems of the handle's batched mode: problem index = e + nb1 * hb
  int nb1;
  int m, n, k;
  double alpha, beta;
  int lower_only, kb0, kb_row, kb_col, ke0, ke_row, ke_col;   // k-range coefficients per 128-row tile index
  int ntm, ntn;
  int heavy_first;            // reverse the tile-row order (k-range grows with the row: longest tiles first)
  int k_super;
  int direct, nst, nsc, sr;   // tile mapping: direct grid, or super-tiles (count, per super-row, rows)
  // balanced persistent schedule (launches whose tiles differ in k-range): bal_wg resident workgroups, bal_tiles tiles per
  // problem enumerated longest k-range first (rows or columns primary, ascending or descending), bal_ny problems
  int balanced, bal_wg, bal_tiles, bal_ny, bal_rows, bal_asc;
  // epilogue 2: max |(float)C_ij| of what this launch stores, per 128-row block of the matrix that starts at amax_base with
  // leading dimension ldc (C lies inside it), as the bits of a non-negative float (atomicMax)
  unsigned* amax;
  const char* amax_base;
};

typedef unsigned int V16 __attribute__((ext_vector_type(4)));   // one 16-byte register quad
template <int V> struct IntC { static constexpr int value = V; };

// ---- global -> registers -> LDS -------------------------------------------------------------------
// An operand k-tile is TS rows x 128 bytes = TS * 8 chunks of 16 bytes, NP = TS * 8 / NT per thread.
// The address of chunk p of k-tile kt is  ubase + kt * step + voff[p]:  voff (32-bit, per lane) is loop
// invariant and ubase + kt * step is wave-uniform, so the loop body needs no vector address arithmetic
// (global_load with a scalar base): the fp32 / fp64 MFMAs issue from the same vector lanes as ordinary VALU
// instructions (measured: they do not overlap), so every v_mul / v_add in the loop costs MFMA time.
template <typename T, bool TR, int NT, int TS>
__device__ __forceinline__ void tile_offsets(long long ld, int tid, unsigned (&voff)[TS * 8 / NT]) {
  constexpr int EPC = 16 / sizeof(T);  // elements per 16-byte chunk
  constexpr int NP = TS * 8 / NT;
  if constexpr (!TR) {
    // stored (rows x k): thread -> chunk c of row (tid >> 3) + (NT / 8) p
    const int c = tid & 7, rr = tid >> 3;
#pragma unroll
    for (int p = 0; p < NP; ++p) voff[p] = (unsigned)(((long long)(rr + (NT / 8) * p) * ld + c * EPC) * (long long)sizeof(T));
  } else {
    // stored (k x rows): BK k-rows of TS elements
    constexpr int CPR = TS / EPC;        // chunks per k-row
    constexpr int RPP = NT / CPR;        // k-rows per pass
    const int c = tid % CPR, kr = tid / CPR;
#pragma unroll
    for (int p = 0; p < NP; ++p) voff[p] = (unsigned)(((long long)(kr + RPP * p) * ld + c * EPC) * (long long)sizeof(T));
  }
}
// buffer_load_dwordx4 with the k-tile's base in a scalar resource descriptor and the lane's 32-bit offset in one
// VGPR (raw buffer, no stride, 2 GB window: the offsets stay below TS rows x ld, < 2^27 bytes)
template <int NP>
__device__ __forceinline__ void load_tile(const char* __restrict__ ubase, const unsigned (&voff)[NP], V16 (&r)[NP]) {
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ubase), 0, 0x7fffffff, 0x00020000);
#pragma unroll
  for (int p = 0; p < NP; ++p) r[p] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[p], 0, 0);
}

template <typename T, bool TR, int NT, int TS>
__device__ __forceinline__ void store_tile(char* lds, int tid, const V16 (&r)[TS * 8 / NT]) {
  constexpr int EPC = 16 / sizeof(T);
  constexpr int NP = TS * 8 / NT;
  if constexpr (!TR) {
    const int c = tid & 7, rr = tid >> 3;
#pragma unroll
    for (int p = 0; p < NP; ++p)
      *reinterpret_cast<V16*>(lds + (rr + (NT / 8) * p) * LDS_N_STRIDE + c * 16) = r[p];
  } else {
    constexpr int CPR = TS / EPC;
    constexpr int RPP = NT / CPR;
    constexpr int RS = (TS + 4) * sizeof(T);   // padded k-row: conflict-free fragment reads
    const int c = tid % CPR, kr = tid / CPR;
#pragma unroll
    for (int p = 0; p < NP; ++p)
      *reinterpret_cast<V16*>(lds + (kr + RPP * p) * RS + c * 16) = r[p];
  }
}

// ---- fp64: AB x NB blocks of 16x16x4 per wave.  Within a k-tile the k index is permuted so that a lane's
// four k values (4 kq .. 4 kq + 3) are contiguous: two ds_read_b128 per block row for k-contiguous images.
template <bool TA, bool TB, int AB, int NB, int TS>
__device__ __forceinline__ void compute_tile(const char* la, const char* lb, int row_w, int col_w, int lane,
                                             d4 (&acc)[AB][NB]) {
  const int r = lane & 15, kq = lane >> 4;
  constexpr int RS = (TS + 4) * 8;
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    double af[AB][2], bf[NB][2];
#pragma unroll
    for (int a = 0; a < AB; ++a) {
      if constexpr (!TA) {
        const double2 v = *reinterpret_cast<const double2*>(la + (row_w + 16 * a + r) * LDS_N_STRIDE +
                                                            (4 * kq + 2 * hh) * 8);
        af[a][0] = v.x; af[a][1] = v.y;
      } else {
        af[a][0] = *reinterpret_cast<const double*>(la + (4 * kq + 2 * hh) * RS + (row_w + 16 * a + r) * 8);
        af[a][1] = *reinterpret_cast<const double*>(la + (4 * kq + 2 * hh + 1) * RS + (row_w + 16 * a + r) * 8);
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if constexpr (!TB) {
        const double2 v = *reinterpret_cast<const double2*>(lb + (col_w + 16 * b + r) * LDS_N_STRIDE +
                                                            (4 * kq + 2 * hh) * 8);
        bf[b][0] = v.x; bf[b][1] = v.y;
      } else {
        bf[b][0] = *reinterpret_cast<const double*>(lb + (4 * kq + 2 * hh) * RS + (col_w + 16 * b + r) * 8);
        bf[b][1] = *reinterpret_cast<const double*>(lb + (4 * kq + 2 * hh + 1) * RS + (col_w + 16 * b + r) * 8);
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int a = 0; a < AB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a][t], bf[b][t], acc[a][b], 0, 0, 0);
  }
}

// ---- fp32: AB x NB blocks of 32x32x2 per wave; a lane's sixteen k values (16 kq .. 16 kq + 15) are
// contiguous: four ds_read_b128 per block row.
template <bool TA, bool TB, int AB, int NB, int TS>
__device__ __forceinline__ void compute_tile(const char* la, const char* lb, int row_w, int col_w, int lane,
                                             f16v (&acc)[AB][NB]) {
  const int r = lane & 31, kq = lane >> 5;
  constexpr int RS = (TS + 4) * 4;
#pragma unroll
  for (int q = 0; q < 4; ++q) {  // 4 groups of 4 k-steps
    float af[AB][4], bf[NB][4];
#pragma unroll
    for (int a = 0; a < AB; ++a) {
      if constexpr (!TA) {
        const float4 v = *reinterpret_cast<const float4*>(la + (row_w + 32 * a + r) * LDS_N_STRIDE +
                                                          (16 * kq + 4 * q) * 4);
        af[a][0] = v.x; af[a][1] = v.y; af[a][2] = v.z; af[a][3] = v.w;
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
          af[a][t] = *reinterpret_cast<const float*>(la + (16 * kq + 4 * q + t) * RS + (row_w + 32 * a + r) * 4);
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if constexpr (!TB) {
        const float4 v = *reinterpret_cast<const float4*>(lb + (col_w + 32 * b + r) * LDS_N_STRIDE +
                                                          (16 * kq + 4 * q) * 4);
        bf[b][0] = v.x; bf[b][1] = v.y; bf[b][2] = v.z; bf[b][3] = v.w;
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
          bf[b][t] = *reinterpret_cast<const float*>(lb + (16 * kq + 4 * q + t) * RS + (col_w + 32 * b + r) * 4);
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int a = 0; a < AB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][t], bf[b][t], acc[a][b], 0, 0, 0);
  }
}

template <int AB, int NB>
__device__ __forceinline__ void zero_acc(d4 (&acc)[AB][NB]) {
#pragma unroll
  for (int a = 0; a < AB; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};
}
template <int AB, int NB>
__device__ __forceinline__ void zero_acc(f16v (&acc)[AB][NB]) {
#pragma unroll
  for (int a = 0; a < AB; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
}

// C/D maps: f64 16x16x4: col = lane & 15, row = (lane >> 4) + 4 * reg;
//           f32 32x32x2: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
// row_g / col_g: global row / column of the wave's sub-tile origin.
template <int AB, int NB>
__device__ __forceinline__ void store_acc(double* __restrict__ C, long long ldc, int row_g, int col_g, int lane,
                                          const d4 (&acc)[AB][NB], double alpha, double beta) {
#pragma unroll
  for (int a = 0; a < AB; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = row_g + 16 * a + (lane >> 4) + 4 * i;
        const int col = col_g + 16 * b + (lane & 15);
        double* p = C + (long long)row * ldc + col;
        double v = alpha * acc[a][b][i];
        if (beta != 0.0) v += beta * (*p);
        *p = v;
      }
}
template <int AB, int NB>
__device__ __forceinline__ void store_acc(float* __restrict__ C, long long ldc, int row_g, int col_g, int lane,
                                          const f16v (&acc)[AB][NB], double alpha, double beta) {
  const float al = (float)alpha, be = (float)beta;
#pragma unroll
  for (int a = 0; a < AB; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = row_g + 32 * a + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
        const int col = col_g + 32 * b + (lane & 31);
        float* p = C + (long long)row * ldc + col;
        float v = al * acc[a][b][i];
        if (be != 0.f) v += be * (*p);
        *p = v;
      }
}

// Epilogue 2: store_acc with beta = 0, returning the lane's max |(float)v| over what it stored.
template <int AB, int NB>
__device__ __forceinline__ float store_acc_amax(double* __restrict__ C, long long ldc, int row_g, int col_g, int lane,
                                                const d4 (&acc)[AB][NB], double alpha) {
  float m = 0.f;
#pragma unroll
  for (int a = 0; a < AB; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = row_g + 16 * a + (lane >> 4) + 4 * i;
        const int col = col_g + 16 * b + (lane & 15);
        const double v = alpha * acc[a][b][i];
        C[(long long)row * ldc + col] = v;
        m = fmaxf(m, fabsf((float)v));
      }
  return m;
}

// Epilogue 1 (column sums of squares): instead of storing the tile, store for each of its TS columns the
// sum over the tile's rows of (alpha * acc)^2, in fp64, at C[tile_row * ldc + col].  WM wave rows each
// contribute a partial per column; col_w = the wave's column offset inside the tile.
template <int AB, int NB, int WM, int TS>
__device__ __forceinline__ void sumsq_acc(char* lds, double* __restrict__ out, long long ldo, int tm, int col0,
                                          int wm, int col_w, int lane, int tid, const d4 (&acc)[AB][NB],
                                          double alpha) {
  double* red = reinterpret_cast<double*>(lds);   // [WM][TS]
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    double s = 0.0;
#pragma unroll
    for (int a = 0; a < AB; ++a)
#pragma unroll
      for (int i = 0; i < 4; ++i) { const double v = alpha * acc[a][b][i]; s = __builtin_fma(v, v, s); }
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (lane < 16) red[wm * TS + col_w + 16 * b + lane] = s;
  }
  __syncthreads();
  if (tid < TS) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < WM; ++w) t += red[w * TS + tid];
    out[(long long)tm * ldo + col0 + tid] = t;
  }
}
template <int AB, int NB, int WM, int TS>
__device__ __forceinline__ void sumsq_acc(char* lds, double* __restrict__ out, long long ldo, int tm, int col0,
                                          int wm, int col_w, int lane, int tid, const f16v (&acc)[AB][NB],
                                          double alpha) {
  double* red = reinterpret_cast<double*>(lds);
  const float al = (float)alpha;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < AB; ++a)
#pragma unroll
      for (int i = 0; i < 16; ++i) { const float v = al * acc[a][b][i]; s = __builtin_fmaf(v, v, s); }
    s += __shfl_xor(s, 32, 64);
    if (lane < 32) red[wm * TS + col_w + 32 * b + lane] = (double)s;
  }
  __syncthreads();
  if (tid < TS) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < WM; ++w) t += red[w * TS + tid];
    out[(long long)tm * ldo + col0 + tid] = t;
  }
}

// accumulator block grid of one wave: (TS/WM) x (TS/2) elements in MFMA blocks
template <typename T, int WM, int TS> struct AccT;
template <int WM, int TS> struct AccT<double, WM, TS> {
  static constexpr int AB = TS / WM / 16, NB = TS / 2 / 16;
  typedef d4 type[AB][NB];
};
template <int WM, int TS> struct AccT<float, WM, TS> {
  static constexpr int AB = TS / WM / 32, NB = TS / 2 / 32;
  typedef f16v type[AB][NB];
};

// TS = 128: WM = 2 -> 256 threads, 64 x 64 per wave, 2 waves/SIMD at 2 workgroups/CU;
//           WM = 4 -> 512 threads, 32 x 64 per wave, 4 waves/SIMD (<= 128 VGPRs)
// TS =  64: WM = 2 -> 256 threads, 32 x 32 per wave, up to 4 workgroups/CU
template <typename T, bool TA, bool TB, int EPI, int WM, int TS>
__global__ __launch_bounds__(WM * 128, TS == 128 ? WM : 4) void gemm_kernel(KParams p) {
  constexpr int LDS_OP_BYTES = TS * LDS_N_STRIDE;   // also >= the k-strided image (BK rows of TS + 4 elements)
  __shared__ __attribute__((aligned(16))) char lds[4 * LDS_OP_BYTES];
  constexpr int BK = Cfg<T>::BK;
  constexpr int NT = WM * 128;
  constexpr int AB = AccT<T, WM, TS>::AB, NB = AccT<T, WM, TS>::NB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int row_w = wm * (TS / WM), col_w = wn * (TS / 2);

  // ---- tile mapping.  Grids that fit one residency round (<= 512 tiles) map block -> tile directly.
  // Larger grids are cut into super-tiles of 64 tiles; hardware workgroup b runs on XCD b % 8 (round-robin
  // dispatch, verified with HW_REG_XCC_ID), so an XCD walks the 64 tiles of one super-tile with its 64 resident
  // workgroups: the 8 + 8 operand panels of a super-tile are shared through that XCD's L2, and every XCD sees
  // an even sample of the tile grid (triangular problems stay balanced).
  // Balanced persistent schedule: a launch whose tiles differ in k-range (triangular operands) and that fits a few
  // residency rounds is bound by where its longest tiles land - resident together on one CU they share its matrix pipe,
  // and a long tile that starts late is the tail of the launch.  Here the tiles are enumerated longest first and dealt to
  // bal_wg resident workgroups in serpentine order (sweep s: positions s W + w going up, then (s + 1) W - 1 - w going
  // down): every workgroup walks a list of the same total length, whatever CU it runs on.  Workgroup ids are dealt
  // round-robin to the XCDs, so w is remapped to give each XCD runs of 8 consecutive positions out of every 64
  // (neighbouring tiles share an operand panel in that XCD's L2; every XCD still samples the whole weight range of a
  // sweep - a contiguous eighth per XCD measured 20 % slower on one-sweep launches, whose heaviest eighth then sits on
  // one XCD).  Per-tile arithmetic is unchanged: results are bit-identical to the static mapping.
  for (int sweep = 0;; ++sweep) {
  int tm, tn;
  long long by = blockIdx.y;
  int rlo = -1, rhi = -1, clo = -1, chi = -1;   // k_super: extreme tile rows / columns of the super-tile
  if (p.balanced) {
    const long long total = (long long)p.bal_tiles * p.bal_ny;
    if ((long long)sweep * p.bal_wg >= total) return;
    const int w = (int)(blockIdx.x >> 6) * 64 + (int)(blockIdx.x & 7) * 8 + (int)((blockIdx.x >> 3) & 7);
    const long long pos = (sweep & 1) ? (long long)(sweep + 1) * p.bal_wg - 1 - w : (long long)sweep * p.bal_wg + w;
    if (pos >= total) continue;
    by = pos % p.bal_ny;
    int q = (int)(pos / p.bal_ny);
    if (p.lower_only) {
      if (!p.bal_asc) q = p.bal_tiles - 1 - q;
      int r = (int)((__builtin_sqrtf(8.0f * (float)q + 1.0f) - 1.0f) * 0.5f);
      while ((r + 1) * (r + 2) / 2 <= q) ++r;
      while (r * (r + 1) / 2 > q) --r;
      tm = r;
      tn = q - r * (r + 1) / 2;
    } else if (p.bal_rows) {
      const int r = q / p.ntn;
      tn = q - r * p.ntn;
      tm = p.bal_asc ? r : p.ntm - 1 - r;
    } else {
      const int c = q / p.ntm;
      tm = q - c * p.ntm;
      tn = p.bal_asc ? c : p.ntn - 1 - c;
    }
  } else if (sweep > 0) {
    return;
  } else if (p.direct) {
    tm = blockIdx.x % p.ntm;
    tn = blockIdx.x / p.ntm;
  } else {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    // super-tiles are dealt to the XCDs in serpentine order (0..7, 7..0, ...): with a k-range that shrinks along
    // the enumeration (heavy_first, triangular operands) plain round-robin always hands XCD 0 the longest of each
    // group of eight (1.2 % more work at the K5 shape, 11 % at 1024 queries)
    const int grp = j >> 6, slot = j & 63;
    const int st = grp * 8 + ((grp & 1) ? 7 - xcd : xcd);
    if (st >= p.nst) return;
    if (p.lower_only) {
      // Lower triangle: first the super-tiles strictly below the diagonal (all 64 tiles active), then the
      // 36 active tiles of each diagonal super-tile packed 64 to a group - no workgroup exits early in the
      // middle of the grid, so the 64 workgroups of an XCD stay in lockstep (early exits on the diagonal let
      // the successors start half a tile apart; they never re-align and stop sharing operand panels in L2).
      const int nsr = (p.ntm + 7) / 8, noff = nsr * (nsr - 1) / 2;
      int R, S;
      if (st < noff) {
        R = (int)((1.0f + __builtin_sqrtf(8.0f * (float)st + 1.0f)) * 0.5f);
        while (R * (R + 1) / 2 <= st) ++R;
        while (R * (R - 1) / 2 > st) --R;
        S = st - R * (R - 1) / 2;
        tm = 8 * R + (slot & 7);
        tn = 8 * S + (slot >> 3);
        if (tm >= p.ntm) return;
      } else {
        const int t = (st - noff) * 64 + slot;
        if (t >= 36 * nsr) return;
        const int d = t / 36, u = t - 36 * d;
        int i = (int)((__builtin_sqrtf(8.0f * (float)u + 1.0f) - 1.0f) * 0.5f);
        while ((i + 1) * (i + 2) / 2 <= u) ++i;
        while (i * (i + 1) / 2 > u) --i;
        tm = 8 * d + i;
        tn = 8 * d + (u - i * (i + 1) / 2);
        R = d; S = d;
        if (tm >= p.ntm) return;
      }
      if (p.k_super) {
        rlo = 8 * R; rhi = min(8 * R + 7, p.ntm - 1);
        clo = 8 * S; chi = min(8 * S + 7, p.ntn - 1);
      }
    } else {
      // Full rectangle: bands of sr tile rows (8; adapted for short or narrow grids), walked column by column,
      // cut into groups of 64 consecutive tiles = one super-tile (8 x 8 when the width is a multiple of 8;
      // otherwise a group runs on into the next band instead of leaving slots idle: 79 tile columns, the K5
      // batch of 10 000 queries, would leave every tenth super-tile 1/8 empty).
      const int sr = p.sr, per = sr * p.ntn, nfull = p.ntm / sr, hlast = p.ntm - nfull * sr;
      const int total = p.ntm * p.ntn;
      auto place = [&](int t, int& row, int& col, int& band) {
        band = min(t / per, nfull);
        const int idx = t - band * per, hh = band < nfull ? sr : hlast;
        col = idx / hh;
        row = band * sr + (idx - col * hh);
      };
      const int t = st * 64 + slot;
      if (t >= total) return;
      int band;
      place(t, tm, tn, band);
      if (p.k_super) {
        // Rows: the whole group, i.e. at most two bands of sr tile rows (what the launch-side gate admits), so that
        // a group that runs on into the next band stays in lockstep.  Columns: only the group's part in this
        // tile's own band -- at most 64 / sr columns -- never the full width between the two parts (the zero band
        // of a triangular operand is only GPK_ZERO_BAND_TILES wide).
        int r0, c0, b0, r1, c1, b1;
        place(st * 64, r0, c0, b0);
        place(min(st * 64 + 63, total - 1), r1, c1, b1);
        if (b1 - b0 <= 1) { rlo = b0 * sr; rhi = min(b1 * sr + sr - 1, p.ntm - 1); }
        else { rlo = band * sr; rhi = min(band * sr + sr - 1, p.ntm - 1); }
        clo = band == b0 ? c0 : 0;
        chi = band == b1 ? c1 : p.ntn - 1;
        // (a short last band makes its groups wide: 64 / 3 columns for three leftover rows)
        if ((chi - clo + 1) * TS > 8 * 128) clo = chi = tn;
      }
    }
    if (p.k_super && p.heavy_first) { const int a0 = p.ntm - 1 - rhi, a1 = p.ntm - 1 - rlo; rlo = a0; rhi = a1; }
  }
  if (p.heavy_first && !p.balanced) tm = p.ntm - 1 - tm;
  if (rlo < 0) { rlo = rhi = tm; clo = chi = tn; }
  if (p.lower_only && tn > tm) return;

  // k-range of this tile; the coefficients are given per 128-row tile index
  // (k_super: the union of the k-ranges of the super-tile's tiles - the coefficients are >= 0, so the lowest
  // row / column gives the start and the highest the end; the operands are zero outside a tile's own range)
  int kb = p.kb0 + p.kb_row * (rlo * TS / 128) + p.kb_col * (clo * TS / 128);
  int ke = p.ke0 < 0 ? p.k : p.ke0 + p.ke_row * (rhi * TS / 128) + p.ke_col * (chi * TS / 128);
  kb = max(kb, 0);
  ke = min(ke, p.k);
  const int nkt = (ke - kb) / BK;

  const long long be = by % p.nb1, bh = by / p.nb1;
  const T* A = reinterpret_cast<const T*>(p.A + be * p.sA + bh * p.hA);
  const T* B = reinterpret_cast<const T*>(p.B + be * p.sB + bh * p.hB);
  T* C = reinterpret_cast<T*>(p.C + be * p.sC + bh * p.hC);
  const int row0 = tm * TS, col0 = tn * TS;

  typename AccT<T, WM, TS>::type acc;
  zero_acc<AB, NB>(acc);

  if (nkt > 0) {
    // Register-staged software pipeline ("write after the barrier"): at the top of iteration kt the
    // registers hold k-tile kt+1 (fetched during iteration kt-1's MFMAs); they are written to the
    // idle LDS buffer, the fetch of k-tile kt+2 is issued, and the MFMAs of k-tile kt run while it
    // is in flight.  One barrier per k-tile; tile indices are clamped so the body is branch-free.
    // (Measured and not kept - DESIGN.md: fetch pinned to the top or the middle of the MFMAs with
    // sched_barrier, s_setprio around the MFMAs: equal or slower on the throughput shapes; 2 or 4 k-tiles per
    // iteration for the 64-tile small-grid configuration: no gain.)
    constexpr int NP = TS * 8 / NT;
    // The 64-tile configuration runs launches of a few hundred workgroups, one or two per CU: with one k-tile of fetch
    // distance such a launch is bound by the latency of its fetches (2.7 us per k-tile of 0.5 us of MFMA work measured at
    // 512 workgroups).  It keeps TWO k-tiles in flight in two register sets (it has the registers: 85 of 128): the set
    // written to LDS at the top of iteration kt (k-tile kt + 1) is refilled with k-tile kt + 3.
    constexpr bool DEEP = TS == 64;
    V16 ra[NP], rb[NP], ra2[NP], rb2[NP];    // (the second set is dead code in the other configurations)
    unsigned offa[NP], offb[NP];
    tile_offsets<T, TA, NT, TS>(p.lda, tid, offa);
    tile_offsets<T, TB, NT, TS>(p.ldb, tid, offb);
    // wave-uniform bases of k-tile 0 and the byte step between k-tiles (k contiguous: BK elements; k strided: BK rows)
    const long long es = sizeof(T);
    const char* ua = reinterpret_cast<const char*>(A) + (TA ? ((long long)kb * p.lda + row0) : ((long long)row0 * p.lda + kb)) * es;
    const char* ub = reinterpret_cast<const char*>(B) + (TB ? ((long long)kb * p.ldb + col0) : ((long long)col0 * p.ldb + kb)) * es;
    const long long sa = (TA ? (long long)BK * p.lda : (long long)BK) * es;
    const long long sb = (TB ? (long long)BK * p.ldb : (long long)BK) * es;
    load_tile<NP>(ua, offa, ra);
    load_tile<NP>(ub, offb, rb);
    store_tile<T, TA, NT, TS>(lds, tid, ra);
    store_tile<T, TB, NT, TS>(lds + LDS_OP_BYTES, tid, rb);
    {
      const int k1 = min(1, nkt - 1);
      load_tile<NP>(ua + k1 * sa, offa, ra);
      load_tile<NP>(ub + k1 * sb, offb, rb);
      if constexpr (DEEP) {
        const int k2 = min(2, nkt - 1);
        load_tile<NP>(ua + k2 * sa, offa, ra2);
        load_tile<NP>(ub + k2 * sb, offb, rb2);
      }
    }
    __syncthreads();
    // two k-tiles per trip so that the LDS buffer index (and, with two register sets, the set) is a compile-time constant
    // (immediate offsets on every ds instruction, no vector address arithmetic in the body)
    auto body = [&](int kt, auto curc) {
      constexpr int cur = decltype(curc)::value;
      const char* la = lds + cur * 2 * LDS_OP_BYTES;
      const char* lb = la + LDS_OP_BYTES;
      if constexpr (DEEP && cur == 1) {
        store_tile<T, TA, NT, TS>(lds + (cur ^ 1) * 2 * LDS_OP_BYTES, tid, ra2);
        store_tile<T, TB, NT, TS>(lds + (cur ^ 1) * 2 * LDS_OP_BYTES + LDS_OP_BYTES, tid, rb2);
        const int kn = min(kt + 3, nkt - 1);
        load_tile<NP>(ua + kn * sa, offa, ra2);
        load_tile<NP>(ub + kn * sb, offb, rb2);
      } else {
        store_tile<T, TA, NT, TS>(lds + (cur ^ 1) * 2 * LDS_OP_BYTES, tid, ra);
        store_tile<T, TB, NT, TS>(lds + (cur ^ 1) * 2 * LDS_OP_BYTES + LDS_OP_BYTES, tid, rb);
        const int kn = min(kt + (DEEP ? 3 : 2), nkt - 1);
        load_tile<NP>(ua + kn * sa, offa, ra);
        load_tile<NP>(ub + kn * sb, offb, rb);
      }
      compute_tile<TA, TB, AB, NB, TS>(la, lb, row_w, col_w, lane, acc);
      __syncthreads();
    };
    int kt = 0;
    for (; kt + 1 < nkt; kt += 2) {
      body(kt, IntC<0>{});
      body(kt + 1, IntC<1>{});
    }
    if (kt < nkt) body(kt, IntC<0>{});
  }
  if constexpr (EPI == 0) {
    store_acc<AB, NB>(C, p.ldc, row0 + row_w, col0 + col_w, lane, acc, p.alpha, p.beta);
  } else if constexpr (EPI == 2) {
    if constexpr (sizeof(T) == 8) {
      float mx = store_acc_amax<AB, NB>(C, p.ldc, row0 + row_w, col0 + col_w, lane, acc, p.alpha);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      // (a wave's rows lie in one 128-row block: tiles are 64 or 128 rows and C starts on a block boundary)
      const long long rowabs = (reinterpret_cast<const char*>(C) - p.amax_base) / (p.ldc * 8) + row0 + row_w;
      if (lane == 0) atomicMax(p.amax + (rowabs >> 7), __float_as_uint(mx));
    }
  } else {
    // the k-loop ended with a barrier: the staging buffers are free for the reduction
    sumsq_acc<AB, NB, WM, TS>(lds, reinterpret_cast<double*>(C), p.ldc, tm, col0, wm, col_w, lane, tid, acc,
                              p.alpha);
    __syncthreads();            // (the reduction used the staging buffers: the next tile of a persistent walk refills them)
  }
  }
}

template <typename T, int WM, int TS>
int launch(gpk_handle h, const GemmArgs& g) {
  KParams p;
  p.A = (const char*)g.A; p.B = (const char*)g.B; p.C = (char*)g.C;
  p.lda = g.lda; p.ldb = g.ldb; p.ldc = g.ldc;
  p.hA = gpk_bstride(h, g.A); p.hB = gpk_bstride(h, g.B); p.hC = gpk_bstride(h, g.C);
  p.sA = p.sB = p.sC = 0;
  p.nb1 = 1;
  unsigned ny = (unsigned)h->batch;
  if (g.nbatch > 0) {
    p.nb1 = g.nbatch; ny *= (unsigned)g.nbatch; p.sA = g.sA; p.sB = g.sB; p.sC = g.sC;
  }
  p.m = g.m; p.n = g.n; p.k = g.k;
  p.alpha = g.alpha; p.beta = g.beta;
  p.lower_only = g.lower_only; p.kb0 = g.kb0; p.kb_row = g.kb_row; p.kb_col = g.kb_col;
  p.ke0 = g.ke0; p.ke_row = g.ke_row; p.ke_col = g.ke_col;
  p.heavy_first = g.heavy_first;
  p.ntm = g.m / TS; p.ntn = g.n / TS;
  const long long ntiles = g.lower_only ? (long long)p.ntm * (p.ntm + 1) / 2 : (long long)p.ntm * p.ntn;
  p.direct = ntiles <= 512 ? 1 : 0;
  p.sr = 8;
  if (!g.lower_only) {
    while (p.sr > p.ntm) p.sr >>= 1;                 // short grids: sr x (64 / sr) super-tiles
    if (p.ntn < 8) { p.sr = 8; while (64 / p.sr > p.ntn && p.sr < 64) p.sr <<= 1; }
  }
  const int sc = 64 / p.sr;
  const int nsr = (p.ntm + p.sr - 1) / p.sr;
  p.nsc = (p.ntn + sc - 1) / sc;
  p.nst = g.lower_only ? nsr * (nsr - 1) / 2 + (36 * nsr + 63) / 64 : (int)(((long long)p.ntm * p.ntn + 63) / 64);
  // k_super widens a tile's k-range to that of its super-tile, which spans at most two bands of sr tile rows:
  // allowed only while that stays inside the zero band the producers of triangular operands guarantee
  p.k_super = (g.k_super && 2 * p.sr * (TS / 64) <= 2 * GPK_ZERO_BAND_TILES) ? 1 : 0;
  long long nblocks = p.direct ? (long long)p.ntm * p.ntn : (long long)((p.nst + 7) / 8) * 512;
  if (nblocks >= (1ll << 31)) { h->err = "gemm: grid too large"; return GPK_BAD_ARG; }
  // Tiles that differ in k-range, few enough residency rounds for the placement of the long ones to matter: the balanced
  // persistent schedule (see the kernel).  dr / dc: how a tile's k-length changes per tile row / column.
  p.balanced = 0; p.bal_wg = 0; p.bal_tiles = 0; p.bal_ny = 1; p.bal_rows = 0; p.bal_asc = 0;
  {
    const int dr = (g.ke0 < 0 ? 0 : g.ke_row) - g.kb_row, dc = (g.ke0 < 0 ? 0 : g.ke_col) - g.kb_col;
    const long long total = ntiles * ny;
    if (h->gemm_balanced && (dr != 0 || (dc != 0 && !g.lower_only)) && total <= h->gemm_balanced_max_tiles) {
      if (h->cus <= 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess || cus <= 0) cus = 256;
        h->cus = cus;
      }
      const long long resident = (long long)h->cus * (TS == 128 ? 2 : TS == 64 ? 4 : 8) / 64 * 64;
      p.balanced = 1;
      p.bal_wg = (int)(total < resident ? (total + 63) / 64 * 64 : resident);
      p.bal_tiles = (int)ntiles;
      p.bal_ny = (int)ny;
      p.bal_rows = (dr != 0 || g.lower_only) ? 1 : 0;
      p.bal_asc = (dr != 0 ? dr < 0 : dc < 0) ? 1 : 0;
      nblocks = p.bal_wg;
      ny = 1;
    }
  }
  dim3 grid((unsigned)nblocks, ny), block(WM * 128);
  p.amax = g.amax; p.amax_base = (const char*)g.amax_base;
  if (g.epilogue == 2) {
    if constexpr (sizeof(T) == 8) {
      if (g.ta || !g.tb || g.beta != 0.0 || !g.amax || !g.amax_base || h->batch != 1) {
        h->err = "gemm: the block-maximum epilogue needs ta == 0, tb == 1, beta == 0, one problem";
        return GPK_BAD_ARG;
      }
      hipLaunchKernelGGL((gemm_kernel<T, false, true, 2, WM, TS>), grid, block, 0, h->stream, p);
      GPK_LAUNCH_CHECK(h);
      return GPK_OK;
    } else {
      h->err = "gemm: the block-maximum epilogue is fp64 only";
      return GPK_BAD_ARG;
    }
  }
  if (g.epilogue == 1) {
    if (g.ta) { h->err = "gemm: the sum-of-squares epilogue needs ta == 0"; return GPK_BAD_ARG; }
    if (!g.tb) hipLaunchKernelGGL((gemm_kernel<T, false, false, 1, WM, TS>), grid, block, 0, h->stream, p);
    else hipLaunchKernelGGL((gemm_kernel<T, false, true, 1, WM, TS>), grid, block, 0, h->stream, p);
    GPK_LAUNCH_CHECK(h);
    return GPK_OK;
  }
  if (!g.ta && !g.tb) hipLaunchKernelGGL((gemm_kernel<T, false, false, 0, WM, TS>), grid, block, 0, h->stream, p);
  else if (!g.ta && g.tb) hipLaunchKernelGGL((gemm_kernel<T, false, true, 0, WM, TS>), grid, block, 0, h->stream, p);
  else if (g.ta && !g.tb) hipLaunchKernelGGL((gemm_kernel<T, true, false, 0, WM, TS>), grid, block, 0, h->stream, p);
  else hipLaunchKernelGGL((gemm_kernel<T, true, true, 0, WM, TS>), grid, block, 0, h->stream, p);
  GPK_LAUNCH_CHECK(h);
  return GPK_OK;
}

}  // namespace

// tile edge gpk_gemm will use for this launch (callers that size per-tile-row outputs need it)
int gpk_gemm_tile(gpk_handle h, const GemmArgs& g) {
  long long t128 = g.lower_only ? (long long)(g.m / 128) * (g.m / 128 + 1) / 2 : (long long)(g.m / 128) * (g.n / 128);
  if (g.nbatch > 0) t128 *= g.nbatch;                  // what counts is how many workgroups the launch has
  t128 *= h->batch;
  const bool aliased = g.C == g.A || g.C == g.B;
  // (fp64 launches of fewer than gemm_tiny_tiles 128-tiles - the products of a 1000-row model: W^T W is 36 tiles - run on
  // 32 x 32 tiles, four waves of 16 x 16: sixteen times the workgroups of the 128-tile form, a sixteenth of the k-loop's matrix
  // work per wave - such a launch is bound by the time of its longest tile; the sum-of-squares and block-maximum epilogues keep 64)
  if (!aliased && g.epilogue == 0 && t128 < h->gemm_tiny_tiles) return 32;
  return (!aliased && t128 < h->gemm_small_tiles) ? 64 : 128;
}

int gpk_gemm(gpk_handle h, int dtype, const GemmArgs& g) {
  const int bk = dtype == GPK_F64 ? 16 : 32;
  if (h->gemm_log)   // option gemm_log = 1: one line per launch, in launch order (joined with a rocprofv3 kernel trace)
    fprintf(stderr, "GPKGEMM %d %d %d %d ta%d tb%d lo%d kb %d %d %d ke %d %d %d\n", dtype, g.m, g.n, g.k, g.ta, g.tb,
            g.lower_only, g.kb0, g.kb_row, g.kb_col, g.ke0, g.ke_row, g.ke_col);
  GPK_REQUIRE(h, g.m > 0 && g.n > 0 && g.m % 128 == 0 && g.n % 128 == 0, "gemm: m, n must be multiples of 128");
  GPK_REQUIRE(h, g.k >= 0 && g.k % bk == 0, "gemm: k must be a multiple of the k-tile");
  GPK_REQUIRE(h, g.kb0 % bk == 0 && g.kb_row % bk == 0 && g.kb_col % bk == 0 && (g.ke0 < 0 || g.ke0 % bk == 0) &&
                     g.ke_row % bk == 0 && g.ke_col % bk == 0,
              "gemm: k-range coefficients must be multiples of the k-tile");
  const int es = dtype == GPK_F64 ? 8 : 4;
  GPK_REQUIRE(h, (g.lda * es) % 16 == 0 && (g.ldb * es) % 16 == 0, "gemm: leading dimensions must be 16-byte multiples");
  GPK_REQUIRE(h, ((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.B % 16) == 0, "gemm: operands must be 16-byte aligned");
  // Small grids: a launch with fewer than h->gemm_small_tiles (1024 = two rounds of the 512 resident workgroups)
  // 128 x 128 tiles is bound by the time of its longest tile, not by the chip; it runs on 64 x 64 tiles instead (4x
  // the workgroups, a quarter of the work each; measured neutral at N = 65 536 and 5-35 % faster for the products of
  // N <= 8192 factorisations).  In-place launches (C aliasing an operand:
  // the 128-wide leaves of the triangular solves) rely on one tile covering everything it reads and keep 128.
  const int tile = gpk_gemm_tile(h, g);
  const bool small = tile == 64;
  if (tile == 32 && dtype == GPK_F64) return launch<double, 2, 32>(h, g);
  // wave rows per 128-tile workgroup (2 -> 256 threads, 4 -> 512 threads), per dtype; tuned on MI355X,
  // overridable through the options gemm_wm_f64 / gemm_wm_f32
  if (dtype == GPK_F64) {
    if (small) return launch<double, 2, 64>(h, g);
    return h->gemm_wm_f64 == 2 ? launch<double, 2, 128>(h, g) : launch<double, 4, 128>(h, g);
  }
  if (small) return launch<float, 2, 64>(h, g);
  return h->gemm_wm_f32 == 2 ? launch<float, 2, 128>(h, g) : launch<float, 4, 128>(h, g);
}
