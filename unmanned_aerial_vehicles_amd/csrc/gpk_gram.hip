// RBF kernel-matrix kernels for gfx950:
//   K1  gram_sym_kernel    symmetric N x N Gram build (HBM write-bound; computes each
//                          64 x 64 tile once and writes it to both triangles)
//       cross_t_kernel     K*^T block (training index major) feeding the variance solve
//   K4  predict_mean_kernel fused k(xq, X) . alpha, K* never stored
//       colsumsq kernels   column sums of squares of V = L^-1 K*^T (fp64 accumulation)
// Squared distances are formed from exact differences of inputs pre-divided by the
// length-scale (the cdist/pdist form of the reference), never by the norm expansion.
#include <cstdlib>

#include "gpk_internal.h"
#include "gpk_math.h"

namespace {

struct LsArr { double v[GPK_MAX_D]; };
struct PArr { double v[GPK_MAX_P]; };
struct Ls16 { double v[16]; };

// ---- exp(x) for x <= 0 ---------------------------------------------------------------------
// fp64: gpk_math.h.
__device__ __forceinline__ double exp_neg(double x) { return gpk_exp_neg(x); }
// fp32: v_exp_f32 on x*log2e with the product's rounding error folded back in.
__device__ __forceinline__ float exp_neg(float x) {
  const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-08f;
  const float hi = x * L2E_HI;
  const float lo = __builtin_fmaf(x, L2E_HI, -hi) + x * L2E_LO;
  const float e = __builtin_amdgcn_exp2f(hi);
  return __builtin_fmaf(e, lo * 0.693147180559945f, e);
}

// type-exact fused multiply-add (the bare __builtin_fma is the double version: on floats it would
// round-trip through fp64)
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

template <typename T> struct Vec16;
template <> struct Vec16<double> { typedef double2 type; static constexpr int N = 2; };
template <> struct Vec16<float> { typedef float4 type; static constexpr int N = 4; };

__device__ __forceinline__ void st16(double* p, double a, double b, bool nt) {
  typedef double dv2 __attribute__((ext_vector_type(2)));
  dv2 v = {a, b};
  if (nt) __builtin_nontemporal_store(v, reinterpret_cast<dv2*>(p));
  else *reinterpret_cast<dv2*>(p) = v;
}
__device__ __forceinline__ void st16(float* p, float a, float b, float c, float d, bool nt) {
  typedef float fv4 __attribute__((ext_vector_type(4)));
  fv4 v = {a, b, c, d};
  if (nt) __builtin_nontemporal_store(v, reinterpret_cast<fv4*>(p));
  else *reinterpret_cast<fv4*>(p) = v;
}
// p = start of the 64-column tile row; writes thread tx's four columns (see colof)
__device__ __forceinline__ void store_row4(double* p, int tx, const double (&v)[4], bool nt) {
  st16(p + 2 * tx, v[0], v[1], nt);
  st16(p + 32 + 2 * tx, v[2], v[3], nt);
}
__device__ __forceinline__ void store_row4(float* p, int tx, const float (&v)[4], bool nt) {
  st16(p + 4 * tx, v[0], v[1], v[2], v[3], nt);
}

constexpr int TS = 64;    // tile edge

// Column c (0..3) of thread tx's 4 x 4 micro-tile.  fp32: 4 consecutive columns (one 16-byte store);
// fp64: two pairs, {2tx, 2tx+1} and {32+2tx, 32+2tx+1}, so that each 16-byte store instruction of
// 16 neighbouring lanes covers 256 contiguous bytes (whole 128-byte lines) instead of every other
// 16-byte piece.
template <typename T> __device__ __forceinline__ int colof(int tx, int c);
template <> __device__ __forceinline__ int colof<double>(int tx, int c) { return (c >> 1) * 32 + 2 * tx + (c & 1); }
template <> __device__ __forceinline__ int colof<float>(int tx, int c) { return 4 * tx + c; }
constexpr int DCH = 16;   // feature chunk held in LDS

// Stage rows [r0, r0+64) of X (n x D) scaled by 1/ls into lds[d][64] for d in [d0, d0+dc).
template <typename T>
__device__ __forceinline__ void stage_x(const T* __restrict__ X, long long n, int D, long long r0, int d0,
                                        int dc, const LsArr& ls, T* lds, int tid) {
  for (int e = tid; e < TS * dc; e += 256) {
    const int i = e / dc, d = e - i * dc;
    const long long gi = r0 + i;
    T v = T(0);
    if (gi < n) v = X[gi * D + d0 + d] / T(ls.v[d0 + d]);
    lds[d * TS + i] = v;
  }
}

// 4 x 4 micro-tile of squared distances accumulated over one feature chunk.
template <typename T>
__device__ __forceinline__ void accum_d2(const T* xi, const T* xj, int dc, int ty, int tx, T (&d2)[4][4]) {
  for (int d = 0; d < dc; ++d) {
    T a[4], b[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) a[r] = xi[d * TS + 4 * ty + r];
#pragma unroll
    for (int c = 0; c < 4; ++c) b[c] = xj[d * TS + colof<T>(tx, c)];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const T df = a[r] - b[c];
        d2[r][c] = fma_t(df, df, d2[r][c]);
      }
  }
}

// ---- K1: symmetric Gram build ---------------------------------------------------------------
template <typename T, bool NT>
__global__ __launch_bounds__(256) void gram_sym_kernel(const T* __restrict__ X, long long N, int D, LsArr ls,
                                                       T sf2, T diag_add, T* __restrict__ K, long long ldk) {
  __shared__ __attribute__((aligned(16))) T xi[DCH * TS];
  __shared__ __attribute__((aligned(16))) T xj[DCH * TS];
  __shared__ __attribute__((aligned(16))) T tb[TS * (TS + 1)];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;

  // linear id -> lower-triangular tile (ti >= tj)
  const long long id = blockIdx.x;
  long long ti = (long long)((__builtin_sqrt(8.0 * (double)id + 1.0) - 1.0) * 0.5);
  while ((ti + 1) * (ti + 2) / 2 <= id) ++ti;
  while (ti * (ti + 1) / 2 > id) --ti;
  const long long tj = id - ti * (ti + 1) / 2;
  const long long i0 = ti * TS, j0 = tj * TS;

  T d2[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) d2[r][c] = T(0);

  for (int dbase = 0; dbase < D; dbase += DCH) {
    const int dc = min(DCH, D - dbase);
    if (dbase) __syncthreads();
    stage_x<T>(X, N, D, i0, dbase, dc, ls, xi, tid);
    stage_x<T>(X, N, D, j0, dbase, dc, ls, xj, tid);
    __syncthreads();
    accum_d2<T>(xi, xj, dc, ty, tx, d2);
  }

  T v[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const long long gi = i0 + 4 * ty + r;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const long long gj = j0 + colof<T>(tx, c);
      T val = sf2 * exp_neg(T(-0.5) * d2[r][c]);
      if (gi == gj) val = sf2 + diag_add;            // RBF diagonal is exactly sf2 (+ white + jitter)
      if (gi >= N || gj >= N) val = (gi == gj) ? T(1) : T(0);   // identity padding
      v[r][c] = val;
    }
    store_row4(K + gi * ldk + j0, tx, v[r], NT);
  }
  if (ti == tj) return;

  // mirrored tile: transpose through LDS, write rows of K[j-block][i-block]
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) tb[(4 * ty + r) * (TS + 1) + colof<T>(tx, c)] = v[r][c];
  __syncthreads();
  constexpr int VEC = Vec16<T>::N;      // elements per 16-byte store
  constexpr int LPR = TS / VEC;         // lanes per output row
  constexpr int RPG = TS / (256 / LPR); // rows per lane group
  const int lp = tid % LPR, g = tid / LPR;
#pragma unroll
  for (int rr = 0; rr < RPG; ++rr) {
    const int jj = g * RPG + rr;
    T w[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) w[e] = tb[(VEC * lp + e) * (TS + 1) + jj];
    T* dst = K + (j0 + jj) * ldk + i0 + VEC * lp;
    if constexpr (VEC == 2) st16(dst, w[0], w[1], NT);
    else st16(dst, w[0], w[1], w[2], w[3], NT);
  }
}

// ---- K1, strip form (D <= 16): one workgroup owns tile row ti and a strip of up to GS consecutive
// column tiles.  The row block's X is staged once; the X rows of the next column tile are fetched into
// registers while the current tile is computed and stored, so the global-load latency of each tile is
// hidden, and the 512-byte row segments of neighbouring tiles are written back to back by one workgroup.
// (GS = 8 at the sizes where the launch fills the chip several times over; small matrices - the reference trains at N = 1000 ..
// 10 000 - get shorter strips, down to one tile per workgroup, so that the launch has enough workgroups: N = 1024 is 24 strips of
// 8 but 136 of 1.  The tiles themselves are computed identically whatever the strip length: bit-identical K.)
template <typename T, bool NT, int GS>
__global__ __launch_bounds__(256) void gram_strip_kernel(const T* __restrict__ X, long long N, int D, LsArr ls,
                                                         T sf2, T diag_add, T* __restrict__ K, long long ldk) {
  __shared__ __attribute__((aligned(16))) T xi[DCH * TS];
  __shared__ __attribute__((aligned(16))) T xj[DCH * TS];
  __shared__ __attribute__((aligned(16))) T tb[TS * (TS + 1)];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  // strip id -> (ti, g): tile row r has floor(r / GS) + 1 strips; rows [GS q, GS q + GS) hold GS (q + 1) strips
  const long long id = blockIdx.x;
  long long q = (long long)((__builtin_sqrt(1.0 + 8.0 * (double)id / GS) - 1.0) * 0.5);   // GS q (q+1) / 2 <= id
  while (GS * (q + 1) * (q + 2) / 2 <= id) ++q;
  while (GS * q * (q + 1) / 2 > id) --q;
  const long long rem_id = id - GS * q * (q + 1) / 2;
  const long long ti = GS * q + rem_id / (q + 1);
  const long long g = rem_id % (q + 1);
  const long long i0 = ti * TS;
  const long long tj0 = g * GS, tj1 = min(ti, tj0 + GS - 1);

  stage_x<T>(X, N, D, i0, 0, D, ls, xi, tid);
  stage_x<T>(X, N, D, tj0 * TS, 0, D, ls, xj, tid);
  // per-thread prefetch registers for the next column tile: element e = tid + 256 p of the (64 x D) block
  constexpr int NPF = (TS * DCH + 255) / 256;   // 4
  T pf[NPF];
  __syncthreads();
  for (long long tj = tj0; tj <= tj1; ++tj) {
    const long long j0 = tj * TS;
    const bool more = tj < tj1;
    if (more) {
#pragma unroll
      for (int p = 0; p < NPF; ++p) {
        const int e = tid + 256 * p;
        const int i = e / D, d = e - i * D;
        const long long gi = j0 + TS + i;
        pf[p] = (e < TS * D && gi < N) ? X[gi * D + d] / T(ls.v[d]) : T(0);
      }
    }
    T d2[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) d2[r][c] = T(0);
    accum_d2<T>(xi, xj, D, ty, tx, d2);
    T v[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long long gi = i0 + 4 * ty + r;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const long long gj = j0 + colof<T>(tx, c);
        T val = sf2 * exp_neg(T(-0.5) * d2[r][c]);
        if (gi == gj) val = sf2 + diag_add;
        if (gi >= N || gj >= N) val = (gi == gj) ? T(1) : T(0);
        v[r][c] = val;
      }
      store_row4(K + gi * ldk + j0, tx, v[r], NT);
    }
    if (ti != tj) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) tb[(4 * ty + r) * (TS + 1) + colof<T>(tx, c)] = v[r][c];
    }
    __syncthreads();          // tb complete; every thread is done reading xj
    if (ti != tj) {
      constexpr int VEC = Vec16<T>::N;
      constexpr int LPR = TS / VEC;
      constexpr int RPG = TS / (256 / LPR);
      const int lp = tid % LPR, gg = tid / LPR;
#pragma unroll
      for (int rr = 0; rr < RPG; ++rr) {
        const int jj = gg * RPG + rr;
        T w[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) w[e] = tb[(VEC * lp + e) * (TS + 1) + jj];
        T* dst = K + (j0 + jj) * ldk + i0 + VEC * lp;
        if constexpr (VEC == 2) st16(dst, w[0], w[1], NT);
        else st16(dst, w[0], w[1], w[2], w[3], NT);
      }
    }
    if (more) {
#pragma unroll
      for (int p = 0; p < NPF; ++p) {
        const int e = tid + 256 * p;
        if (e < TS * D) { const int i = e / D, d = e - i * D; xj[d * TS + i] = pf[p]; }
      }
    }
    __syncthreads();          // xj holds the next tile; tb may be overwritten
  }
}

// ---- K*^T block: B[j][m] = sf2 exp(-0.5 |x_j - xq_m|^2), zero in the padding -------------------
template <typename T>
__global__ __launch_bounds__(256) void cross_t_kernel(const T* __restrict__ X, long long N,
                                                      const T* __restrict__ Xq, long long M, int D, LsArr ls,
                                                      T sf2, T* __restrict__ B, long long ldb) {
  __shared__ __attribute__((aligned(16))) T xi[DCH * TS];
  __shared__ __attribute__((aligned(16))) T xj[DCH * TS];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const long long i0 = (long long)blockIdx.y * TS, j0 = (long long)blockIdx.x * TS;  // i: train, j: query
  T d2[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) d2[r][c] = T(0);
  for (int dbase = 0; dbase < D; dbase += DCH) {
    const int dc = min(DCH, D - dbase);
    if (dbase) __syncthreads();
    stage_x<T>(X, N, D, i0, dbase, dc, ls, xi, tid);
    stage_x<T>(Xq, M, D, j0, dbase, dc, ls, xj, tid);
    __syncthreads();
    accum_d2<T>(xi, xj, dc, ty, tx, d2);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const long long gi = i0 + 4 * ty + r;
    T v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const long long gj = j0 + colof<T>(tx, c);
      T val = sf2 * exp_neg(T(-0.5) * d2[r][c]);
      if (gi >= N || gj >= M) val = T(0);
      v[c] = val;
    }
    store_row4(B + gi * ldb + j0, tx, v, false);
  }
}

// ---- K* written straight into the fp16 x 2 fragment-order layout of the variance launch ------------------------
// Kq[q][j] = sf2 exp(-0.5 |xq_q - x_j|^2) (query-major, j = training point = the k index of V = W Kq^T), zero in the
// padding, never stored as fp32: a lane computes the eight values of one (query, k-half) and writes their two fp16 parts
// as two 16-byte chunks of the layout of k5_direct_kernel (gpk_k5split.hip: chunk (q, k16 block kb, half h, part s) at
// (((q / 32) * KB + kb) * 2 + s) * 64 + h * 32 + q % 32) - a separate fp32 panel (2.65 GB at the headline shape) and a
// pass that splits it (2.65 GB read + 2.65 GB written) never exist.  Same arithmetic per entry as cross_t_kernel (exact
// differences of the length-scale-divided coordinates, FMA accumulation, exp_neg), then x * scale = h0 + h1.
// Workgroup: 64 queries x 128 training points (the staged points are reused 64 times); lane = h * 32 + q % 32 owns ONE
// query and the k-half h of four consecutive k16 blocks; every wave instruction of the store writes 1 KiB of
// contiguous output.
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
constexpr int CS2_QS = 4;
template <int DD>   // DD = D (query coordinates in registers; one instantiation per feature count)
__global__ __launch_bounds__(256) void cross_split2_kernel(const float* __restrict__ Xq, long long M,
                                                           const float* __restrict__ X, long long N, int D, LsArr ls,
                                                           float sf2, float scale, u4_t* __restrict__ dst, long long Np) {
  __shared__ __attribute__((aligned(16))) float xs[DD][128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long j0 = (long long)blockIdx.x * 128;
  for (int e = tid; e < 128 * DD; e += 256) {
    const int d = e >> 7, i = e & 127;
    const long long gj = j0 + i;
    xs[d][i] = (d < D && gj < N) ? X[gj * D + d] / (float)ls.v[d] : 0.f;
  }
  __syncthreads();
  const int r = lane & 31, hh = lane >> 5;
  const long long KB = Np >> 4, Mp = (M + 127) / 128 * 128;
  // CS2_QS sub-blocks of 64 queries per workgroup (the staged points serve 256 queries): two query blocks of 32 per
  // sub-block; a wave takes four of the eight k16 blocks of one of them
#pragma unroll 1
  for (int qs = 0; qs < CS2_QS; ++qs) {
  const long long q0 = ((long long)blockIdx.y * CS2_QS + qs) * 64;
  if (q0 >= Mp) break;
  const long long qblk = (q0 >> 5) + (wave >> 1), q = qblk * 32 + r;
  float xq[DD];
#pragma unroll
  for (int d = 0; d < DD; ++d) xq[d] = (q < M ? Xq[q * DD + d] : 0.f) / (float)ls.v[d];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int kb = 4 * (wave & 1) + t, kbase = 16 * kb + 8 * hh;
    float d2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) d2[i] = 0.f;
#pragma unroll
    for (int d = 0; d < DD; ++d) {
      const float4 lo = *reinterpret_cast<const float4*>(&xs[d][kbase]), hi = *reinterpret_cast<const float4*>(&xs[d][kbase + 4]);
      const float b[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float df = xq[d] - b[i]; d2[i] = __builtin_fmaf(df, df, d2[i]); }
    }
    h8_t p0, p1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float val = sf2 * exp_neg(-0.5f * d2[i]);
      if (q >= M || j0 + kbase + i >= N) val = 0.f;
      const float x = val * scale;
      const _Float16 h0 = (_Float16)x;
      p0[i] = h0;
      p1[i] = (_Float16)(x - (float)h0);
    }
    u4_t* o = dst + ((qblk * KB + (j0 >> 4) + kb) * 2) * 64 + lane;
    o[0] = __builtin_bit_cast(u4_t, p0);
    o[64] = __builtin_bit_cast(u4_t, p1);
  }
  }
}

// ---- K4: fused posterior mean ---------------------------------------------------------------------
// One thread owns QPT queries (coordinates and output accumulators in registers); training rows
// [x_j / ls (16 slots) | alpha_j (16 slots)] are staged through LDS with a fixed 32-element stride and
// read as wave-uniform 16-byte broadcasts.  D and P are rounded up to multiples of 4 at compile time
// (D4, P4 in 1..4; padding slots hold zeros and contribute nothing), so the inner loop is branch-free.
// The training set is split over gridDim.y; partial sums are combined in a fixed order by
// mean_reduce_kernel.
constexpr int PM_DMAX = 16, PM_PMAX = 16, PM_TJ = 128, PM_QPT = 2, PM_RS = PM_DMAX + PM_PMAX;

template <typename T> struct Quad;
template <> struct Quad<float> { typedef float4 type; };
template <> struct Quad<double> { typedef double4 type; };

template <typename T, int D4, int P4>
__global__ __launch_bounds__(256) void predict_mean_kernel(const T* __restrict__ X, const T* __restrict__ alpha,
                                                           long long N, int D, int P, Ls16 ls,
                                                           const T* __restrict__ Xq, long long M,
                                                           long long chunk, T* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) T rows[PM_TJ * PM_RS];
  typedef typename Quad<T>::type Q4;
  const int tid = threadIdx.x;
  constexpr int DD = 4 * D4, PP = 4 * P4;
  T xq[PM_QPT][DD];
  T acc[PM_QPT][PP];
  long long qm[PM_QPT];
#pragma unroll
  for (int q = 0; q < PM_QPT; ++q) {
    qm[q] = ((long long)blockIdx.x * PM_QPT + q) * 256 + tid;
#pragma unroll
    for (int d = 0; d < DD; ++d) {
      xq[q][d] = T(0);
      if (d < D && qm[q] < M) xq[q][d] = Xq[qm[q] * D + d] / T(ls.v[d]);
    }
#pragma unroll
    for (int p = 0; p < PP; ++p) acc[q][p] = T(0);
  }
  // fp32: two-level sums - `acc` runs over one staged round (128 terms) and is then added to `tot`: the rounding of an
  // fp32 chain grows like its length (each add rounds at the magnitude of the running sum), and a single chain over a
  // whole chunk was the largest term of the fp32 mean's error for large query batches (whose chunks are long)
  constexpr bool TWO_LEVEL = sizeof(T) == 4;
  T tot[PM_QPT][PP];
#pragma unroll
  for (int q = 0; q < PM_QPT; ++q)
#pragma unroll
    for (int p = 0; p < PP; ++p) tot[q][p] = T(0);
  const long long n0 = (long long)blockIdx.y * chunk;
  const long long n1 = min(N, n0 + chunk);
  for (long long jb = n0; jb < n1; jb += PM_TJ) {
    const int nj = (int)min((long long)PM_TJ, n1 - jb);
    if constexpr (TWO_LEVEL) {
#pragma unroll
      for (int q = 0; q < PM_QPT; ++q)
#pragma unroll
        for (int p = 0; p < PP; ++p) { tot[q][p] += acc[q][p]; acc[q][p] = T(0); }
    }
    __syncthreads();
    // stage [x/ls | 0.. | alpha | 0..] for the nj <= PM_TJ rows of this round
    for (int e = tid; e < nj * (DD + PP); e += 256) {
      const int j = e / (DD + PP), c = e - j * (DD + PP);
      T v = T(0);
      if (c < DD) { if (c < D) v = X[(jb + j) * D + c] / T(ls.v[c]); }
      else if (c - DD < P) v = alpha[(jb + j) * P + (c - DD)];
      rows[j * PM_RS + (c < DD ? c : PM_DMAX + (c - DD))] = v;
    }
    __syncthreads();
#pragma unroll 2
    for (int j = 0; j < nj; ++j) {
      const Q4* row = reinterpret_cast<const Q4*>(rows + j * PM_RS);
      T xt[DD], al[PP];
#pragma unroll
      for (int g = 0; g < D4; ++g) {
        const Q4 v = row[g];
        xt[4 * g] = v.x; xt[4 * g + 1] = v.y; xt[4 * g + 2] = v.z; xt[4 * g + 3] = v.w;
      }
#pragma unroll
      for (int g = 0; g < P4; ++g) {
        const Q4 v = row[PM_DMAX / 4 + g];
        al[4 * g] = v.x; al[4 * g + 1] = v.y; al[4 * g + 2] = v.z; al[4 * g + 3] = v.w;
      }
#pragma unroll
      for (int q = 0; q < PM_QPT; ++q) {
        T d2 = T(0);
#pragma unroll
        for (int d = 0; d < DD; ++d) {
          const T df = xq[q][d] - xt[d];
          d2 = fma_t(df, df, d2);
        }
        const T e = exp_neg(T(-0.5) * d2);
#pragma unroll
        for (int p = 0; p < PP; ++p) acc[q][p] = fma_t(e, al[p], acc[q][p]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < PM_QPT; ++q) {
    if (qm[q] < M) {
#pragma unroll
      for (int p = 0; p < PP; ++p)
        if (p < P) partial[((long long)blockIdx.y * M + qm[q]) * P + p] = TWO_LEVEL ? tot[q][p] + acc[q][p] : acc[q][p];
    }
  }
}

template <typename T>
using pm_fn = void (*)(const T*, const T*, long long, int, int, Ls16, const T*, long long, long long, T*);
template <typename T, int D4>
pm_fn<T> pm_pick_p(int p4) {
  switch (p4) {
    case 1: return predict_mean_kernel<T, D4, 1>;
    case 2: return predict_mean_kernel<T, D4, 2>;
    case 3: return predict_mean_kernel<T, D4, 3>;
    default: return predict_mean_kernel<T, D4, 4>;
  }
}
template <typename T>
pm_fn<T> pm_pick(int d4, int p4) {
  switch (d4) {
    case 1: return pm_pick_p<T, 1>(p4);
    case 2: return pm_pick_p<T, 2>(p4);
    case 3: return pm_pick_p<T, 3>(p4);
    default: return pm_pick_p<T, 4>(p4);
  }
}

// ---- K4, multi-model form: B independent ARD GPs (own length-scales, signal variance and alpha)
// that share the training inputs X, evaluated in one launch: per (query, training point) pair the
// feature differences are formed once and reused by every model.  Rows are staged as
// [x raw (16 slots) | alpha_0..alpha_{B-1} (16 slots)]; the per-model weights 1/ls^2 live in LDS.
template <typename T, int D4, int B4>
__global__ __launch_bounds__(256) void predict_mean_multi_kernel(const T* __restrict__ X,
                                                                 const T* __restrict__ alpha, long long N, int D,
                                                                 int B, const double* __restrict__ w_dev,
                                                                 const T* __restrict__ Xq, long long M,
                                                                 long long chunk, T* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) T rows[PM_TJ * PM_RS];
  __shared__ __attribute__((aligned(16))) T wl[4 * B4 * 4 * D4];     // [b][d] = 1 / ls_bd^2 (0 in the padding)
  typedef typename Quad<T>::type Q4;
  const int tid = threadIdx.x;
  constexpr int DD = 4 * D4, BB = 4 * B4;
  for (int e = tid; e < BB * DD; e += 256) {
    const int b = e / DD, d = e - b * DD;
    wl[e] = (b < B && d < D) ? T(w_dev[b * D + d]) : T(0);
  }
  T xq[PM_QPT][DD];
  T acc[PM_QPT][BB];
  long long qm[PM_QPT];
#pragma unroll
  for (int q = 0; q < PM_QPT; ++q) {
    qm[q] = ((long long)blockIdx.x * PM_QPT + q) * 256 + tid;
#pragma unroll
    for (int d = 0; d < DD; ++d) {
      xq[q][d] = T(0);
      if (d < D && qm[q] < M) xq[q][d] = Xq[qm[q] * D + d];
    }
#pragma unroll
    for (int b = 0; b < BB; ++b) acc[q][b] = T(0);
  }
  const long long n0 = (long long)blockIdx.y * chunk;
  const long long n1 = min(N, n0 + chunk);
  for (long long jb = n0; jb < n1; jb += PM_TJ) {
    const int nj = (int)min((long long)PM_TJ, n1 - jb);
    __syncthreads();
    for (int e = tid; e < PM_TJ * (DD + BB); e += 256) {
      const int j = e / (DD + BB), c = e - j * (DD + BB);
      T v = T(0);
      if (j < nj) {
        if (c < DD) { if (c < D) v = X[(jb + j) * D + c]; }
        else if (c - DD < B) v = alpha[(jb + j) * B + (c - DD)];
      }
      rows[j * PM_RS + (c < DD ? c : PM_DMAX + (c - DD))] = v;
    }
    __syncthreads();
#pragma unroll 2
    for (int j = 0; j < PM_TJ; ++j) {
      const Q4* row = reinterpret_cast<const Q4*>(rows + j * PM_RS);
      T xt[DD], al[BB];
#pragma unroll
      for (int g = 0; g < D4; ++g) {
        const Q4 v = row[g];
        xt[4 * g] = v.x; xt[4 * g + 1] = v.y; xt[4 * g + 2] = v.z; xt[4 * g + 3] = v.w;
      }
#pragma unroll
      for (int g = 0; g < B4; ++g) {
        const Q4 v = row[PM_DMAX / 4 + g];
        al[4 * g] = v.x; al[4 * g + 1] = v.y; al[4 * g + 2] = v.z; al[4 * g + 3] = v.w;
      }
#pragma unroll
      for (int q = 0; q < PM_QPT; ++q) {
        T sq[DD];
#pragma unroll
        for (int d = 0; d < DD; ++d) { const T df = xq[q][d] - xt[d]; sq[d] = df * df; }
#pragma unroll
        for (int b = 0; b < BB; ++b) {
          T d2 = T(0);
#pragma unroll
          for (int d = 0; d < DD; ++d) d2 = fma_t(sq[d], wl[b * DD + d], d2);
          acc[q][b] = fma_t(exp_neg(T(-0.5) * d2), al[b], acc[q][b]);
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < PM_QPT; ++q) {
    if (qm[q] < M) {
#pragma unroll
      for (int b = 0; b < BB; ++b)
        if (b < B) partial[((long long)blockIdx.y * M + qm[q]) * B + b] = acc[q][b];
    }
  }
}

// mean[m][b] = y_mean[b] + y_std[b] * sf2[b] * sum_s partial[s][m][b]
template <typename T>
__global__ void mean_reduce_multi_kernel(const T* __restrict__ partial, int S, long long M, int B, PArr sf2,
                                         PArr ymean, PArr ystd, T* __restrict__ mean) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= M * B) return;
  const int b = (int)(e % B);
  T s = T(0);
  for (int k = 0; k < S; ++k) s += partial[(long long)k * M * B + e];
  mean[e] = T(ymean.v[b]) + T(ystd.v[b]) * (T(sf2.v[b]) * s);
}

template <typename T>
using pmm_fn = void (*)(const T*, const T*, long long, int, int, const double*, const T*, long long, long long, T*);
template <typename T, int D4>
pmm_fn<T> pmm_pick_b(int b4) {
  return b4 <= 1 ? predict_mean_multi_kernel<T, D4, 1> : predict_mean_multi_kernel<T, D4, 2>;
}
template <typename T>
pmm_fn<T> pmm_pick(int d4, int b4) {
  switch (d4) {
    case 1: return pmm_pick_b<T, 1>(b4);
    case 2: return pmm_pick_b<T, 2>(b4);
    case 3: return pmm_pick_b<T, 3>(b4);
    default: return pmm_pick_b<T, 4>(b4);
  }
}

template <typename T>
__global__ void mean_reduce_kernel(const T* __restrict__ partial, int S, long long M, int P, T sf2, PArr ymean,
                                   PArr ystd, T* __restrict__ mean) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= M * P) return;
  const int p = (int)(e % P);
  double s = 0.0;                     // (fp32 partials too: the S-term chain adds no rounding of its own)
  for (int k = 0; k < S; ++k) s += (double)partial[(long long)k * M * P + e];
  mean[e] = T(ymean.v[p]) + T(ystd.v[p]) * (sf2 * T(s));
}

// ---- column sums of squares (fp64 accumulation) -------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void colsumsq_kernel(const T* __restrict__ B, long long rows_per, long long Np,
                                                       long long ldb, double* __restrict__ partial, long long Mp) {
  // 128 columns per workgroup; the two thread halves take alternate rows
  const int c = threadIdx.x & 127, hh = threadIdx.x >> 7;
  const long long m = (long long)blockIdx.x * 128 + c;
  const long long r0 = (long long)blockIdx.y * rows_per, r1 = min(Np, r0 + rows_per);
  double s0 = 0.0, s1 = 0.0;
  long long r = r0 + hh;
  for (; r + 2 < r1; r += 4) {
    const double a = (double)B[r * ldb + m], b = (double)B[(r + 2) * ldb + m];
    s0 = __builtin_fma(a, a, s0);
    s1 = __builtin_fma(b, b, s1);
  }
  if (r < r1) { const double a = (double)B[r * ldb + m]; s0 = __builtin_fma(a, a, s0); }
  partial[((long long)blockIdx.y * 2 + hh) * Mp + m] = s0 + s1;
}
__global__ void colsum_reduce_kernel(const double* __restrict__ partial, int S, long long Mp,
                                     double* __restrict__ out) {
  const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= Mp) return;
  double s = 0.0;
  for (int k = 0; k < S; ++k) s += partial[(long long)k * Mp + m];
  out[m] = s;
}
// sum of the S partial rows and the clipped variance in one launch: var[m] = max(kss - sum_k partial[k][m], floor)
__global__ void colsum_finalize_kernel(const double* __restrict__ partial, int S, long long Mp, long long M, double kss,
                                       double floor_, double* __restrict__ var) {
  const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  double s = 0.0;
  for (int k = 0; k < S; ++k) s += partial[(long long)k * Mp + m];
  var[m] = fmax(kss - s, floor_);
}
// the same with the un-normalisation and the packing of a serving step's result folded in:
// out[m] = [mean[m][0..P) | max(kss - sum_k partial[k][m], floor) * y_std[p]^2, p = 0..P)   (sklearn/_gpr.py:487-489)
__global__ void colsum_finalize_packed_kernel(const double* __restrict__ partial, int S, long long Mp, long long M, double kss,
                                              double floor_, const float* __restrict__ mean, int P, PArr ystd,
                                              double recheck_below, unsigned* __restrict__ low_count,
                                              double* __restrict__ out) {
  const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  double s = 0.0;
  for (int k = 0; k < S; ++k) s += partial[(long long)k * Mp + m];
  const double v = fmax(kss - s, floor_);
  if (low_count && v < recheck_below) atomicAdd(low_count, 1u);      // (the fp32 serving gate: rows the caller recomputes)
  double* o = out + m * 2 * P;
  for (int p = 0; p < P; ++p) { o[p] = (double)mean[m * P + p]; o[P + p] = v * ystd.v[p] * ystd.v[p]; }
}
// [mean | var y_std^2] rows from a separate mean (M x P, fp32 or fp64) and variance (M, fp64)
template <typename T>
__global__ void pack_mean_var_kernel(const T* __restrict__ mean, const double* __restrict__ var, long long M, int P, PArr ystd,
                                     double* __restrict__ out) {
  const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  double* o = out + m * 2 * P;
  const double v = var[m];
  for (int p = 0; p < P; ++p) { o[p] = (double)mean[m * P + p]; o[P + p] = v * ystd.v[p] * ystd.v[p]; }
}
__global__ void var_finalize_kernel(const double* ss, long long M, double kss, double floor_, double* var) {
  const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  var[m] = fmax(kss - ss[m], floor_);
}

int fill_ls(gpk_handle h, const double* ls, int D, LsArr& out) {
  GPK_REQUIRE(h, D >= 1 && D <= GPK_MAX_D, "D must be in [1, 64]");
  GPK_REQUIRE(h, ls != nullptr, "length-scale array is null");
  for (int d = 0; d < GPK_MAX_D; ++d) out.v[d] = 1.0;
  for (int d = 0; d < D; ++d) {
    GPK_REQUIRE(h, ls[d] > 0.0, "length-scales must be positive");
    out.v[d] = ls[d];
  }
  return GPK_OK;
}

}  // namespace

extern "C" int gpk_gram(gpk_handle h, int dtype, const void* X, int64_t N, int D, const double* ls, double sf2,
                        double diag_add, void* K, int64_t ldk) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, N >= 1 && X && K, "gram: null pointer or N < 1");
  const int64_t Np = gpk_padded(N);
  GPK_REQUIRE(h, ldk >= Np && ldk % 2 == 0, "gram: ldk must be >= gpk_padded(N)");
  GPK_REQUIRE(h, dtype == GPK_F32 || dtype == GPK_F64, "gram: bad dtype");
  LsArr l;
  GPK_TRY(fill_ls(h, ls, D, l));
  const int64_t nt = Np / TS;
  const int64_t tiles = nt * (nt + 1) / 2;
  GPK_REQUIRE(h, tiles < (1ll << 31), "gram: N too large");
  const bool stream_nt = Np >= 16384;   // streaming stores once K exceeds the caches
  GPK_REQUIRE(h, ((uintptr_t)K % 16) == 0, "gram: K must be 16-byte aligned");
  const dim3 block(256);
  // strip kernel for D <= 16 (one chunk of features); the tile-per-workgroup kernel otherwise
  const bool strip = D <= DCH;
  auto count = [&](int gs) { long long c = 0; for (long long r = 0; r < nt; ++r) c += r / gs + 1; return c; };
  const int gs = count(8) >= 1024 ? 8 : count(2) >= 512 ? 2 : 1;     // strips per launch: enough workgroups for 256 CUs
  const long long nstrips = count(gs);
  const dim3 grid((unsigned)(strip ? nstrips : tiles));
#define GPK_GRAM_LAUNCH(T, NT)                                                                              \
  do {                                                                                                      \
    if (strip && gs == 8)                                                                                   \
      hipLaunchKernelGGL((gram_strip_kernel<T, NT, 8>), grid, block, 0, h->stream, (const T*)X, (long long)N, D, l, \
                         (T)sf2, (T)diag_add, (T*)K, (long long)ldk);                                       \
    else if (strip && gs == 2)                                                                              \
      hipLaunchKernelGGL((gram_strip_kernel<T, NT, 2>), grid, block, 0, h->stream, (const T*)X, (long long)N, D, l, \
                         (T)sf2, (T)diag_add, (T*)K, (long long)ldk);                                       \
    else if (strip)                                                                                         \
      hipLaunchKernelGGL((gram_strip_kernel<T, NT, 1>), grid, block, 0, h->stream, (const T*)X, (long long)N, D, l, \
                         (T)sf2, (T)diag_add, (T*)K, (long long)ldk);                                       \
    else                                                                                                    \
      hipLaunchKernelGGL((gram_sym_kernel<T, NT>), grid, block, 0, h->stream, (const T*)X, (long long)N, D, l,  \
                         (T)sf2, (T)diag_add, (T*)K, (long long)ldk);                                       \
  } while (0)
  gpk_time_begin(h, GPK_TIMED_GRAM);
  if (dtype == GPK_F64) { if (stream_nt) GPK_GRAM_LAUNCH(double, true); else GPK_GRAM_LAUNCH(double, false); }
  else { if (stream_nt) GPK_GRAM_LAUNCH(float, true); else GPK_GRAM_LAUNCH(float, false); }
  gpk_time_end(h);
#undef GPK_GRAM_LAUNCH
  GPK_LAUNCH_CHECK(h);
  return GPK_OK;
}

// Kq (gpk_padded(M) x Np) in the fragment-order fp16 x 2 layout, from fp32 coordinates (internal: gpk_predict_var_inv_split2)
int gpk_cross_split2(gpk_handle h, const float* Xq, int64_t M, const float* X, int64_t N, int D, const double* ls,
                     double sf2, double scale, void* dst) {
  GPK_REQUIRE(h, Xq && X && dst && M >= 1 && N >= 1 && D >= 1 && D <= DCH, "cross_split2: bad argument");
  const int64_t Np = gpk_padded(N), Mp = gpk_padded(M);
  LsArr l;
  GPK_TRY(fill_ls(h, ls, D, l));
  GPK_REQUIRE(h, Mp / 64 < 65536, "cross_split2: at most 2^22 queries per call");
  const dim3 grid((unsigned)(Np / 128), (unsigned)((Mp / 64 + CS2_QS - 1) / CS2_QS));
#define GPK_CS2(DD) case DD: hipLaunchKernelGGL(cross_split2_kernel<DD>, grid, dim3(256), 0, h->stream, Xq, (long long)M, X, (long long)N, D, \
                                                l, (float)sf2, (float)scale, (u4_t*)dst, (long long)Np); break
  switch (D) {
    GPK_CS2(1); GPK_CS2(2); GPK_CS2(3); GPK_CS2(4); GPK_CS2(5); GPK_CS2(6); GPK_CS2(7); GPK_CS2(8);
    GPK_CS2(9); GPK_CS2(10); GPK_CS2(11); GPK_CS2(12); GPK_CS2(13); GPK_CS2(14); GPK_CS2(15); GPK_CS2(16);
  }
#undef GPK_CS2
  GPK_LAUNCH_CHECK(h);
  return GPK_OK;
}

extern "C" int gpk_cross_gram_t(gpk_handle h, int dtype, const void* X, int64_t N, const void* Xq, int64_t M,
                                int D, const double* ls, double sf2, void* B, int64_t ldb) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, N >= 1 && M >= 1 && X && Xq && B, "cross_gram: null pointer or empty input");
  const int64_t Np = gpk_padded(N), Mp = gpk_padded(M);
  GPK_REQUIRE(h, ldb >= Mp && ldb % 4 == 0, "cross_gram: ldb must be >= gpk_padded(M)");
  GPK_REQUIRE(h, dtype == GPK_F32 || dtype == GPK_F64, "cross_gram: bad dtype");
  LsArr l;
  GPK_TRY(fill_ls(h, ls, D, l));
  dim3 grid((unsigned)(Mp / TS), (unsigned)(Np / TS));
  GPK_REQUIRE(h, Np / TS < 65536, "cross_gram: N too large");
  if (dtype == GPK_F64)
    hipLaunchKernelGGL(cross_t_kernel<double>, grid, dim3(256), 0, h->stream, (const double*)X, (long long)N,
                       (const double*)Xq, (long long)M, D, l, sf2, (double*)B, (long long)ldb);
  else
    hipLaunchKernelGGL(cross_t_kernel<float>, grid, dim3(256), 0, h->stream, (const float*)X, (long long)N,
                       (const float*)Xq, (long long)M, D, l, (float)sf2, (float*)B, (long long)ldb);
  GPK_LAUNCH_CHECK(h);
  return GPK_OK;
}

// ---- row slab of the padded Gram matrix (multi-GPU build: every rank writes the rows it owns, no exchange) --------
// diagonal of the slab's rows: sf2 + diag_add inside the data, 1 in the identity padding
namespace {
template <typename T>
__global__ void slab_diag_kernel(T* __restrict__ K, long long ldk, long long row0, long long nrows_p, long long N,
                                 long long Np, T dval) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= nrows_p) return;
  const long long i = row0 + r;
  if (i < Np) K[r * ldk + i] = i < N ? dval : T(1);
}
}  // namespace

extern "C" int gpk_gram_rows(gpk_handle h, int dtype, const void* X, int64_t N, int D, const double* ls, double sf2,
                             double diag_add, int64_t row0, int64_t nrows, void* Kslab, int64_t ldk) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, X && Kslab && N >= 1, "gram_rows: null pointer or empty input");
  const int64_t Np = gpk_padded(N), nrp = gpk_padded(nrows);
  GPK_REQUIRE(h, row0 >= 0 && row0 % GPK_TILE == 0 && nrows >= 1 && row0 + nrp <= Np,
              "gram_rows: row0 must be a multiple of 128 and the slab must lie inside the padded matrix");
  GPK_REQUIRE(h, ldk >= Np && ldk % 4 == 0, "gram_rows: ldk must be >= gpk_padded(N)");
  GPK_REQUIRE(h, dtype == GPK_F32 || dtype == GPK_F64, "gram_rows: bad dtype");
  const size_t es = dtype == GPK_F64 ? 8 : 4;
  const int64_t valid = row0 >= N ? 0 : (N - row0 < nrp ? N - row0 : nrp);      // data rows of the slab
  if (valid > 0) {
    // rows = this slab's training points, columns = all training points: the cross-kernel block (zero in the padding)
    GPK_TRY(gpk_cross_gram_t(h, dtype, (const char*)X + (size_t)row0 * D * es, valid, X, N, D, ls, sf2, Kslab, ldk));
    if (gpk_padded(valid) < nrp)
      GPK_CHECK_HIP(h, hipMemsetAsync((char*)Kslab + (size_t)gpk_padded(valid) * ldk * es, 0,
                                      (size_t)(nrp - gpk_padded(valid)) * ldk * es, h->stream));
  } else {
    GPK_CHECK_HIP(h, hipMemsetAsync(Kslab, 0, (size_t)nrp * ldk * es, h->stream));
  }
  const unsigned nb = (unsigned)((nrp + 255) / 256);
  if (dtype == GPK_F64)
    hipLaunchKernelGGL(slab_diag_kernel<double>, dim3(nb), dim3(256), 0, h->stream, (double*)Kslab, (long long)ldk,
                       (long long)row0, (long long)nrp, (long long)N, (long long)Np, sf2 + diag_add);
  else
    hipLaunchKernelGGL(slab_diag_kernel<float>, dim3(nb), dim3(256), 0, h->stream, (float*)Kslab, (long long)ldk,
                       (long long)row0, (long long)nrp, (long long)N, (long long)Np, (float)sf2 + (float)diag_add);
  GPK_LAUNCH_CHECK(h);
  return GPK_OK;
}

extern "C" int gpk_predict_mean(gpk_handle h, int dtype, const void* X, const void* alpha, int64_t N, int D, int P,
                                const double* ls, double sf2, const double* y_mean, const double* y_std,
                                const void* Xq, int64_t M, void* mean) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, X && alpha && Xq && mean && y_mean && y_std, "predict_mean: null pointer");
  GPK_REQUIRE(h, N >= 1 && M >= 1, "predict_mean: empty input");
  GPK_REQUIRE(h, D >= 1 && D <= PM_DMAX, "predict_mean: D must be in [1, 16]");
  GPK_REQUIRE(h, P >= 1 && P <= PM_PMAX, "predict_mean: P must be in [1, 16]");
  GPK_REQUIRE(h, dtype == GPK_F32 || dtype == GPK_F64, "predict_mean: bad dtype");
  LsArr l;
  GPK_TRY(fill_ls(h, ls, D, l));
  PArr ym{}, ys{};
  for (int p = 0; p < P; ++p) { ym.v[p] = y_mean[p]; ys.v[p] = y_std[p]; }
  Ls16 l16;
  for (int d = 0; d < 16; ++d) l16.v[d] = l.v[d];
  const int64_t nqb = (M + 256 * PM_QPT - 1) / (256 * PM_QPT);
  // split the training set so that the grid has >= ~2048 workgroups (8 per CU); a handful of queries against
  // a small training set (the control loop: 1..25 rows at N ~ 1000) is pure latency - a thread walks its
  // chunk serially - so those get chunks of 32 rows instead of 128
  const int64_t gran = (M <= 512 && N <= 16384) ? 32 : PM_TJ;
  int64_t S = (2048 + nqb - 1) / nqb;
  // fp32: chunks of at most 2048 training points, whatever the batch size (a thread's fp32 sums stay short: 128 terms per
  // round, 16 rounds; the partials of the chunks are then added in fp64)
  if (dtype == GPK_F32 && S < (N + 2047) / 2048) S = (N + 2047) / 2048;
  const int64_t maxS = (N + gran - 1) / gran;
  if (S > maxS) S = maxS;
  if (S < 1) S = 1;
  if (S > 65535) S = 65535;
  int64_t chunk = (N + S - 1) / S;
  chunk = (chunk + gran - 1) / gran * gran;
  S = (N + chunk - 1) / chunk;
  const size_t es = dtype == GPK_F64 ? 8 : 4;
  void* partial = nullptr;
  GPK_TRY(gpk_scratch(h, (size_t)S * M * P * es, &partial));
  dim3 grid((unsigned)nqb, (unsigned)S);
  const int64_t tot = M * P;
  const int d4 = (D + 3) / 4, p4 = (P + 3) / 4;
  if (dtype == GPK_F64) {
    hipLaunchKernelGGL(pm_pick<double>(d4, p4), grid, dim3(256), 0, h->stream, (const double*)X,
                       (const double*)alpha, (long long)N, D, P, l16, (const double*)Xq, (long long)M,
                       (long long)chunk, (double*)partial);
    GPK_LAUNCH_CHECK(h);
    hipLaunchKernelGGL(mean_reduce_kernel<double>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream,
                       (const double*)partial, (int)S, (long long)M, P, sf2, ym, ys, (double*)mean);
  } else {
    hipLaunchKernelGGL(pm_pick<float>(d4, p4), grid, dim3(256), 0, h->stream, (const float*)X,
                       (const float*)alpha, (long long)N, D, P, l16, (const float*)Xq, (long long)M,
                       (long long)chunk, (float*)partial);
    GPK_LAUNCH_CHECK(h);
    hipLaunchKernelGGL(mean_reduce_kernel<float>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream,
                       (const float*)partial, (int)S, (long long)M, P, (float)sf2, ym, ys, (float*)mean);
  }
  GPK_LAUNCH_CHECK(h);
  return GPK_OK;
}

extern "C" int gpk_predict_mean_multi(gpk_handle h, int dtype, const void* X, const void* alpha, int64_t N, int D,
                                      int B, const double* ls, const double* sf2, const double* y_mean,
                                      const double* y_std, const void* Xq, int64_t M, void* mean) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, X && alpha && Xq && mean && ls && sf2 && y_mean && y_std, "predict_mean_multi: null pointer");
  GPK_REQUIRE(h, N >= 1 && M >= 1, "predict_mean_multi: empty input");
  GPK_REQUIRE(h, D >= 1 && D <= PM_DMAX, "predict_mean_multi: D must be in [1, 16]");
  GPK_REQUIRE(h, B >= 1 && B <= 8, "predict_mean_multi: B must be in [1, 8]");
  GPK_REQUIRE(h, dtype == GPK_F32 || dtype == GPK_F64, "predict_mean_multi: bad dtype");
  PArr ym{}, ys{}, sf{};
  double w[8 * PM_DMAX];
  for (int b = 0; b < B; ++b) {
    ym.v[b] = y_mean[b]; ys.v[b] = y_std[b]; sf.v[b] = sf2[b];
    for (int d = 0; d < D; ++d) {
      GPK_REQUIRE(h, ls[b * D + d] > 0.0, "predict_mean_multi: length-scales must be positive");
      w[b * D + d] = 1.0 / (ls[b * D + d] * ls[b * D + d]);
    }
  }
  const int64_t nqb = (M + 256 * PM_QPT - 1) / (256 * PM_QPT);
  int64_t S = (2048 + nqb - 1) / nqb;
  const int64_t maxS = (N + PM_TJ - 1) / PM_TJ;
  if (S > maxS) S = maxS;
  if (S < 1) S = 1;
  if (S > 65535) S = 65535;
  int64_t chunk = (N + S - 1) / S;
  chunk = (chunk + PM_TJ - 1) / PM_TJ * PM_TJ;
  S = (N + chunk - 1) / chunk;
  const size_t es = dtype == GPK_F64 ? 8 : 4;
  const size_t wbytes = 8 * PM_DMAX * sizeof(double);
  void* ws = nullptr;
  GPK_TRY(gpk_scratch(h, wbytes + (size_t)S * M * B * es, &ws));
  double* w_dev = (double*)ws;
  void* partial = (char*)ws + wbytes;
  // the weight table is tiny: stage it through the pinned host mirror of the handle
  GPK_CHECK_HIP(h, hipStreamSynchronize(h->stream));
  memcpy(h->h_small, w, (size_t)B * D * sizeof(double));
  GPK_CHECK_HIP(h, hipMemcpyAsync(w_dev, h->h_small, (size_t)B * D * sizeof(double), hipMemcpyHostToDevice, h->stream));
  dim3 grid((unsigned)nqb, (unsigned)S);
  const int64_t tot = M * B;
  const int d4 = (D + 3) / 4, b4 = (B + 3) / 4;
  if (dtype == GPK_F64) {
    hipLaunchKernelGGL(pmm_pick<double>(d4, b4), grid, dim3(256), 0, h->stream, (const double*)X,
                       (const double*)alpha, (long long)N, D, B, (const double*)w_dev, (const double*)Xq,
                       (long long)M, (long long)chunk, (double*)partial);
    GPK_LAUNCH_CHECK(h);
    hipLaunchKernelGGL(mean_reduce_multi_kernel<double>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream,
                       (const double*)partial, (int)S, (long long)M, B, sf, ym, ys, (double*)mean);
  } else {
    hipLaunchKernelGGL(pmm_pick<float>(d4, b4), grid, dim3(256), 0, h->stream, (const float*)X,
                       (const float*)alpha, (long long)N, D, B, (const double*)w_dev, (const float*)Xq,
                       (long long)M, (long long)chunk, (float*)partial);
    GPK_LAUNCH_CHECK(h);
    hipLaunchKernelGGL(mean_reduce_multi_kernel<float>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream,
                       (const float*)partial, (int)S, (long long)M, B, sf, ym, ys, (float*)mean);
  }
  GPK_LAUNCH_CHECK(h);
  return GPK_OK;
}

extern "C" int gpk_colsumsq(gpk_handle h, int dtype, const void* B, int64_t Np, int64_t Mp, int64_t ldb,
                            double* out) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, B && out, "colsumsq: null pointer");
  GPK_REQUIRE(h, Np > 0 && Mp > 0 && Mp % 128 == 0 && ldb >= Mp, "colsumsq: Mp must be a multiple of 128");
  GPK_REQUIRE(h, dtype == GPK_F32 || dtype == GPK_F64, "colsumsq: bad dtype");
  const int64_t nmb = Mp / 128;
  int64_t S = (2048 + nmb - 1) / nmb;
  if (S > Np / 64) S = Np / 64;
  if (S < 1) S = 1;
  if (S > 32768) S = 32768;
  const int64_t rows_per = (Np + S - 1) / S;
  S = (Np + rows_per - 1) / rows_per;
  void* partial = nullptr;
  GPK_TRY(gpk_scratch(h, (size_t)2 * S * Mp * sizeof(double), &partial));
  dim3 grid((unsigned)nmb, (unsigned)S);
  if (dtype == GPK_F64)
    hipLaunchKernelGGL(colsumsq_kernel<double>, grid, dim3(256), 0, h->stream, (const double*)B,
                       (long long)rows_per, (long long)Np, (long long)ldb, (double*)partial, (long long)Mp);
  else
    hipLaunchKernelGGL(colsumsq_kernel<float>, grid, dim3(256), 0, h->stream, (const float*)B, (long long)rows_per,
                       (long long)Np, (long long)ldb, (double*)partial, (long long)Mp);
  GPK_LAUNCH_CHECK(h);
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)((Mp + 255) / 256)), dim3(256), 0, h->stream,
                     (const double*)partial, (int)(2 * S), (long long)Mp, out);
  GPK_LAUNCH_CHECK(h);
  return GPK_OK;
}

int gpk_colsum_finalize(gpk_handle h, const double* partial, int S, int64_t Mp, int64_t M, double kss, double floor_,
                        double* var) {
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, h->stream, partial, S,
                     (long long)Mp, (long long)M, kss, floor_, var);
  GPK_LAUNCH_CHECK(h);
  return GPK_OK;
}
int gpk_colsum_finalize_packed(gpk_handle h, const double* partial, int S, int64_t Mp, int64_t M, double kss, double floor_,
                               const float* mean, int P, const double* y_std, double recheck_below, unsigned* low_count,
                               double* out) {
  PArr ys{};
  for (int p = 0; p < P; ++p) ys.v[p] = y_std[p];
  hipLaunchKernelGGL(colsum_finalize_packed_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, h->stream, partial, S,
                     (long long)Mp, (long long)M, kss, floor_, mean, P, ys, recheck_below, low_count, out);
  GPK_LAUNCH_CHECK(h);
  return GPK_OK;
}
extern "C" int gpk_pack_mean_var(gpk_handle h, int dtype, const void* mean, const double* var, int64_t M, int P,
                                 const double* y_std, double* out) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, mean && var && y_std && out && M >= 1 && P >= 1 && P <= GPK_MAX_P, "pack_mean_var: bad argument");
  GPK_REQUIRE(h, dtype == GPK_F32 || dtype == GPK_F64, "pack_mean_var: bad dtype");
  PArr ys{};
  for (int p = 0; p < P; ++p) ys.v[p] = y_std[p];
  const dim3 grid((unsigned)((M + 255) / 256));
  if (dtype == GPK_F32)
    hipLaunchKernelGGL(pack_mean_var_kernel<float>, grid, dim3(256), 0, h->stream, (const float*)mean, var, (long long)M, P, ys, out);
  else
    hipLaunchKernelGGL(pack_mean_var_kernel<double>, grid, dim3(256), 0, h->stream, (const double*)mean, var, (long long)M, P, ys, out);
  GPK_LAUNCH_CHECK(h);
  return GPK_OK;
}
int gpk_colsum_reduce(gpk_handle h, const double* partial, int S, int64_t Mp, double* out) {
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)((Mp + 255) / 256)), dim3(256), 0, h->stream, partial, S,
                     (long long)Mp, out);
  GPK_LAUNCH_CHECK(h);
  return GPK_OK;
}

int gpk_var_finalize(gpk_handle h, const double* ss, int64_t M, double kss, double floor_, double* var) {
  hipLaunchKernelGGL(var_finalize_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, h->stream, ss,
                     (long long)M, kss, floor_, var);
  GPK_LAUNCH_CHECK(h);
  return GPK_OK;
}
