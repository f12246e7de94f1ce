// K4 on the matrix cores (fp32 serving path): posterior mean  mu = K(Xq, X) alpha  with the pairwise
// squared distances of 32 x 32 (query, training point) blocks formed by MFMAs instead of D subtract/FMA
// pairs per (query, point) on the vector ALU.
//
// With u = sqrt(log2(e)/2) * (x - c) / ls (c = a common centre: the training mean) the kernel value is
//   k = exp2(-d),  d = |u_q|^2 + |u_j|^2 - 2 u_q . u_j
// and d is ONE augmented dot product of depth D + 1 plus an accumulator seed:
//   A (queries, m) = [ u_q (D) | 1 | 0.. ],  B (training, n) = [ -2 u_j (D) | |u_j|^2 | 0.. ],  C_init[m][n] = |u_q|^2
// Per pair only v_exp_f32 + P FMAs remain on the vector ALU.
//
//  * D <= 14 (mean_bf16_kernel; |u_q|^2 rides along as one more component, depth D + 2): the dot product runs on the
//    bf16 matrix pipe at fp32 accuracy.  Every fp32
//    operand is split EXACTLY into three bf16 parts (x = x0 + x1 + x2, 8 significant bits each, rounded to
//    nearest; the remainders are exact in fp32) and the product becomes six v_mfma_f32_32x32x16_bf16
//    (a0 b0, a0 b1, a1 b0, a1 b1, a0 b2, a2 b0; the dropped terms are below 2^-24 of |a||b|; bf16 x bf16
//    products are exact and accumulate in fp32): 192 matrix-pipe cycles per block.
//  * D = 15, 16 (mean_mfma_kernel, depth up to 17): nine v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains).  That
//    instruction runs at the vector rate and was measured NOT to overlap with the exp/FMA work (it shares
//    the fp32 lanes), which is why the bf16 split is preferred wherever the depth fits one instruction.
//
// The expansion loses ~|u|^2 * 2^-23 absolutely in d, so the caller centres the data and uses this path only
// while max |u|^2 is modest (device.py gates on it); the exact-difference VALU kernel in gpk_gram.hip serves
// everything else and all of fp64.
//
// Accumulator layout of a 32 x 32 block (C/D map of the 32x32 MFMAs): lane l, register r holds
// (query row (r & 3) + 8 (r >> 2) + 4 (l >> 5), training point l & 31): a lane keeps ONE training point per
// block, so alpha_j are per-lane values and its 16 x P running sums belong to 16 queries; the 32 lanes of a
// half-wave are combined once, after the loop over the training chunk.  Measured on MI355X (N = 65536,
// D = 9, P = 3, 2^20 queries): 14.1 ms against 43.6 ms for the VALU kernel; vector-ALU bound.
//
// Reference: the mean of sklearn _gpr.py:443-447 / RBF.__call__ kernels.py:1564-1565 and
// quadrotor_gp_mpc/gaussian_process.py:223-226 (same value; fp32 arithmetic).
#include "gpk_internal.h"

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int MM_TJ = 512;        // training points staged per LDS round

struct F16 { float v[16]; };
struct D16 { double v[16]; };

__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// One pipeline step: issue the KP-MFMA chain of the NEXT (points, queries) block and, in between its
// instructions, turn the finished block `cur` into kernel values and accumulate them:
// acc[r][p] += exp2(-cur[r]) * av[p].  Returns the new chain's accumulator.
template <int KP, int PP>
__device__ __forceinline__ f16v mean_step(const float (&a)[KP], const float (&bf)[KP], const f16v& seed, const f16v& cur,
                                          const float (&av)[PP], float (&acc)[16][PP]) {
  f16v nxt = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], bf[0], seed, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < KP; ++i) {
    if (i > 0) nxt = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bf[i], nxt, 0, 0, 0);
#pragma unroll
    for (int r = 16 * i / KP; r < 16 * (i + 1) / KP; ++r) {
      const float e = __builtin_amdgcn_exp2f(-cur[r]);
#pragma unroll
      for (int p = 0; p < PP; ++p) acc[r][p] = __builtin_fmaf(e, av[p], acc[r][p]);
    }
  }
  return nxt;
}

// QB query blocks of 32 per wave (4 waves: 128 * QB queries per workgroup); PP outputs (exact)
template <int KP, int PP, int QB>
__global__ __launch_bounds__(256, 2) void mean_mfma_kernel(const float* __restrict__ X, const float* __restrict__ alpha,
                                                           long long N, int D, F16 sc, F16 ctr,
                                                           const float* __restrict__ Xq, long long M, long long chunk,
                                                           float* __restrict__ partial) {
  __shared__ float xt[2 * KP][MM_TJ];      // row k: B operand component k of every staged point
  __shared__ float al[PP][MM_TJ];
  __shared__ float qn_s[128 * QB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ln = lane & 31, lh = lane >> 5;
  const long long q0 = (long long)blockIdx.x * (128 * QB);

  // ---- query side: norms through LDS, A fragments and accumulator seeds in registers
  for (int t = tid; t < 128 * QB; t += 256) {
    const long long q = q0 + t;
    float s = 0.f;
    if (q < M)
      for (int d = 0; d < D; ++d) {
        const float u = (Xq[q * D + d] - ctr.v[d]) * sc.v[d];
        s = __builtin_fmaf(u, u, s);
      }
    qn_s[t] = s;
  }
  float a[QB][KP];
#pragma unroll
  for (int b = 0; b < QB; ++b) {
    const long long q = q0 + (wave * QB + b) * 32 + ln;
#pragma unroll
    for (int i = 0; i < KP; ++i) {
      const int k = 2 * i + lh;
      float v = 0.f;
      if (q < M) {
        if (k < D) v = (Xq[q * D + k] - ctr.v[k]) * sc.v[k];
        else if (k == D) v = 1.f;
      }
      a[b][i] = v;
    }
  }
  __syncthreads();
  f16v seed[QB];
#pragma unroll
  for (int b = 0; b < QB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) seed[b][r] = qn_s[(wave * QB + b) * 32 + acc_row(r, lane)];

  float acc[QB][16][PP];
#pragma unroll
  for (int b = 0; b < QB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int p = 0; p < PP; ++p) acc[b][r][p] = 0.f;

  const long long n0 = (long long)blockIdx.y * chunk;
  const long long n1 = min(N, n0 + chunk);
  for (long long jb = n0; jb < n1; jb += MM_TJ) {
    __syncthreads();
    // ---- stage MM_TJ training points: [-2 u | |u|^2 | 0] and alpha (zero beyond the chunk)
    for (int t = tid; t < MM_TJ; t += 256) {
      const long long j = jb + t;
      const bool in = j < n1;
      float tn = 0.f;
      for (int d = 0; d < D; ++d) {
        const float u = in ? (X[j * D + d] - ctr.v[d]) * sc.v[d] : 0.f;
        tn = __builtin_fmaf(u, u, tn);
        xt[d][t] = -2.f * u;
      }
      xt[D][t] = tn;
      for (int k = D + 1; k < 2 * KP; ++k) xt[k][t] = 0.f;      // (D = 15: two unused components, not one)
#pragma unroll
      for (int p = 0; p < PP; ++p) al[p][t] = in ? alpha[j * PP + p] : 0.f;
    }
    __syncthreads();
    const int nblk = (int)((min((long long)MM_TJ, n1 - jb) + 31) / 32);

    // ---- software pipeline over (block of 32 points, query block) steps: the MFMA chain of the next step is
    // issued in between the exp/FMA work on the current one (mean_step), so the matrix core and the vector
    // ALU of the SIMD run side by side within a wave
    float bf[KP], av[PP];
#pragma unroll
    for (int i = 0; i < KP; ++i) bf[i] = xt[2 * i + lh][ln];
#pragma unroll
    for (int p = 0; p < PP; ++p) av[p] = al[p][ln];
    f16v cur = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][0], bf[0], seed[0], 0, 0, 0);
#pragma unroll
    for (int i = 1; i < KP; ++i) cur = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][i], bf[i], cur, 0, 0, 0);
    for (int jj = 0; jj < nblk; ++jj) {
      // fragments of the next block of points (clamped on the last one: its chain is discarded)
      const int jn = min(jj + 1, nblk - 1) * 32 + ln;
      float bfn[KP], avn[PP];
#pragma unroll
      for (int i = 0; i < KP; ++i) bfn[i] = xt[2 * i + lh][jn];
#pragma unroll
      for (int p = 0; p < PP; ++p) avn[p] = al[p][jn];
      if constexpr (QB == 2) {
        cur = mean_step<KP, PP>(a[1], bf, seed[1], cur, av, acc[0]);
      }
      cur = mean_step<KP, PP>(a[0], bfn, seed[0], cur, av, acc[QB - 1]);
#pragma unroll
      for (int i = 0; i < KP; ++i) bf[i] = bfn[i];
#pragma unroll
      for (int p = 0; p < PP; ++p) av[p] = avn[p];
    }
  }

  // ---- combine the 32 lanes of each half-wave (they hold different training points of the same 16 queries)
#pragma unroll
  for (int b = 0; b < QB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int p = 0; p < PP; ++p) {
        float v = acc[b][r][p];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        acc[b][r][p] = v;
      }
  if (ln == 0) {
#pragma unroll
    for (int b = 0; b < QB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long q = q0 + (wave * QB + b) * 32 + acc_row(r, lane);
        if (q < M) {
#pragma unroll
          for (int p = 0; p < PP; ++p) partial[((long long)blockIdx.y * M + q) * PP + p] = acc[b][r][p];
        }
      }
  }
}

// ---- D <= 15: distances on the bf16 matrix pipe, exact three-way operand split ----------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// bf16 part of a finite fp32 value, rounded to nearest even (as bits in the upper half of a dword)
__device__ __forceinline__ unsigned bf16_rn_bits(float x) {
  const unsigned u = __float_as_uint(x);
  return (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
}
// x = x0 + x1 + x2 exactly, each part a bf16 obtained by rounding to nearest: |x1| <= 2^-9 |x|, |x2| <= 2^-18 |x|
// and the remainders are exact in fp32; the signs of x1, x2 are not tied to the sign of x, so the cross terms the
// six-product scheme drops (x1 y2, x2 y1, x2 y2 <= 2^-26 |x y|) average out like rounding errors.  (Splitting by
// truncation makes every part share the sign of x: the dropped terms then have the sign of x y and add up
// coherently - measured 17x the error of the fp32 MFMA on the variance launch, whose row sums cancel heavily.)
__device__ __forceinline__ void split3(float x, unsigned& h0, unsigned& h1, unsigned& h2) {
  h0 = bf16_rn_bits(x);
  const float r1 = x - __uint_as_float(h0);
  h1 = bf16_rn_bits(r1);
  const float r2 = r1 - __uint_as_float(h1);
  h2 = __float_as_uint(r2);          // at most 8 significant bits left: the low half is zero
}
// three bf16x8 fragments (k = 0..7 of this lane's half) from eight fp32 values
__device__ __forceinline__ void split_frag(const float (&v)[8], u32x4 (&f)[3]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unsigned a0, a1, a2, b0, b1, b2;
    split3(v[2 * j], a0, a1, a2);
    split3(v[2 * j + 1], b0, b1, b2);
    f[0][j] = (a0 >> 16) | b0;
    f[1][j] = (a1 >> 16) | b1;
    f[2][j] = (a2 >> 16) | (b2 & 0xffff0000u);
  }
}
__device__ __forceinline__ f16v chain6(const u32x4 (&a)[3], const u32x4 (&b)[3], const f16v& seed) {
  const bf16x8 a0 = __builtin_bit_cast(bf16x8, a[0]), a1 = __builtin_bit_cast(bf16x8, a[1]),
               a2 = __builtin_bit_cast(bf16x8, a[2]);
  const bf16x8 b0 = __builtin_bit_cast(bf16x8, b[0]), b1 = __builtin_bit_cast(bf16x8, b[1]),
               b2 = __builtin_bit_cast(bf16x8, b[2]);
  // Smallest terms first, the dominant product last, starting from zero (|u_q|^2 is a component of the dot product,
  // not an accumulator seed): the five partial sums ahead of the last instruction are ~2^-8 of d, so their fp32
  // roundings are negligible and d carries ONE rounding at its own magnitude instead of six.  The exponent's
  // absolute error is the kernel value's relative error: this order took the mean's error constant from 3.5e-7 to
  // 1.4e-7 per term (tools/exp_fp32_gate.py).
  f16v c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, seed, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c, 0, 0, 0);
  return c;
}
typedef float f2 __attribute__((ext_vector_type(2)));
// acc[i][p] holds the running sums of accumulator rows 2i (x) and 2i + 1 (y): one v_pk_fma_f32 per row pair
// and output, with alpha_p broadcast to both halves
template <int PP>
__device__ __forceinline__ void consume(const f16v& cur, const float (&av)[PP], f2 (&acc)[8][PP]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    f2 e;
    e.x = __builtin_amdgcn_exp2f(-cur[2 * i]);
    e.y = __builtin_amdgcn_exp2f(-cur[2 * i + 1]);
#pragma unroll
    for (int p = 0; p < PP; ++p) {
      const f2 a2 = {av[p], av[p]};
      acc[i][p] = __builtin_elementwise_fma(e, a2, acc[i][p]);
    }
  }
}

// D <= 14 (operand depth D + 2 <= 16).  The training side of the product is the same for every query block, so it is
// prepared ONCE per call by mean_prep_kernel - centred, scaled, split, in MFMA-fragment order: block of 32 points jb,
// part s, half h, point t at xb[((jb * 3 + s) * 2 + h) * 32 + t], i.e. one fully coalesced 16-byte load per lane and
// part - and the waves load their B fragments from L2 straight into registers: no LDS, no barriers.  (Staging and
// splitting 512 points per round inside every workgroup, as this kernel did before, was a third of its vector-ALU
// instructions: every workgroup of 256 queries redid the split of the whole training chunk.)
// alpha goes along as al[(jb * PP + p) * 32 + t] (zero beyond N: padded points then contribute exp2(0) * 0).
template <int PP>
__global__ __launch_bounds__(256) void mean_prep_kernel(const float* __restrict__ X, const float* __restrict__ alpha,
                                                        long long N, int D, F16 sc, F16 ctr, u32x4* __restrict__ xb,
                                                        float* __restrict__ al, long long npad) {
  const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
  if (j >= npad) return;
  const bool in = j < N;
  float v[16];
  float tn = 0.f;
#pragma unroll
  for (int d = 0; d < 16; ++d) {
    float u = 0.f;
    if (d < D && in) u = (X[j * D + d] - ctr.v[d]) * sc.v[d];
    tn = __builtin_fmaf(u, u, tn);
    v[d] = -2.f * u;
  }
#pragma unroll
  for (int d = 0; d < 16; ++d) {
    if (d == D) v[d] = tn;
    if (d == D + 1) v[d] = in ? 1.f : 0.f;      // multiplies the queries' |u_q|^2 component
  }
  const long long jb = j >> 5;
  const int t = (int)(j & 31);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float vh[8];
#pragma unroll
    for (int j8 = 0; j8 < 8; ++j8) vh[j8] = v[8 * h + j8];
    u32x4 f[3];
    split_frag(vh, f);
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) xb[((jb * 3 + s3) * 2 + h) * 32 + t] = f[s3];
  }
#pragma unroll
  for (int p = 0; p < PP; ++p) al[(jb * PP + p) * 32 + t] = in ? alpha[j * PP + p] : 0.f;
}

// one halving step per lane-mask bit, MASK, MASK / 2, ...: v[0 .. CNT) -> v[0 .. CNT / 2)
template <int CNT, int MASK, int STEPS>
__device__ __forceinline__ void static_for_halving(float* v, int ln) {
  if constexpr (STEPS > 0) {
    const bool bit = (ln & MASK) != 0;
#pragma unroll
    for (int i = 0; i < CNT / 2; ++i) {
      const float send = bit ? v[i] : v[i + CNT / 2];
      const float keep = bit ? v[i + CNT / 2] : v[i];
      v[i] = keep + __shfl_xor(send, MASK, 64);
    }
    static_for_halving<CNT / 2, MASK / 2, STEPS - 1>(v, ln);
  }
}

template <int PP> struct MeanOperand { u32x4 bf[3]; float av[PP]; };

template <int PP, int QB>
__global__ __launch_bounds__(256, 2) void mean_bf16_kernel(const u32x4* __restrict__ xb, const float* __restrict__ al,
                                                           int nblk, int D, F16 sc, F16 ctr,
                                                           const float* __restrict__ Xq, long long M, int chunk_blocks,
                                                           float* __restrict__ partial) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ln = lane & 31, lh = lane >> 5;
  const long long q0 = (long long)blockIdx.x * (128 * QB);

  // A (queries) = [ u_q (D) | 1 | |u_q|^2 | 0.. ],  B (training) = [ -2 u_j (D) | |u_j|^2 | 1 | 0.. ]: depth D + 2 <= 16
  u32x4 a[QB][3];
#pragma unroll
  for (int b = 0; b < QB; ++b) {
    const long long q = q0 + (wave * QB + b) * 32 + ln;
    float qn = 0.f;
    if (q < M)
      for (int d = 0; d < D; ++d) {
        const float u = (Xq[q * D + d] - ctr.v[d]) * sc.v[d];
        qn = __builtin_fmaf(u, u, qn);
      }
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 8 * lh + j;
      v[j] = 0.f;
      if (q < M) {
        if (k < D) v[j] = (Xq[q * D + k] - ctr.v[k]) * sc.v[k];
        else if (k == D) v[j] = 1.f;
        else if (k == D + 1) v[j] = qn;
      }
    }
    split_frag(v, a[b]);
  }
  f16v zero;                          // the chains start from zero
#pragma unroll
  for (int r = 0; r < 16; ++r) zero[r] = 0.f;

  f2 acc[QB][8][PP];
#pragma unroll
  for (int b = 0; b < QB; ++b)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int p = 0; p < PP; ++p) acc[b][i][p] = f2{0.f, 0.f};

  // this workgroup's chunk of point blocks; two operands in registers: the one in use and the next, whose loads are
  // issued a whole step (12 MFMAs, 32 exponentials) ahead of their first use
  const int b0 = blockIdx.y * chunk_blocks, b1 = min(nblk, b0 + chunk_blocks), nb = b1 - b0;
  MeanOperand<PP> R[2];
  auto ldf = [&](int blk, MeanOperand<PP>& o) {
    blk = min(blk, b1 - 1);
    const u32x4* q = xb + (long long)blk * 192 + lh * 32 + ln;
    o.bf[0] = q[0]; o.bf[1] = q[64]; o.bf[2] = q[128];
  };
  auto lda = [&](int blk, MeanOperand<PP>& o) {
    blk = min(blk, b1 - 1);
#pragma unroll
    for (int p = 0; p < PP; ++p) o.av[p] = al[((long long)blk * PP + p) * 32 + ln];
  };
  ldf(b0, R[0]); lda(b0, R[0]);
  ldf(b0 + 1, R[1]); lda(b0 + 1, R[1]);
  f16v cur = chain6(a[0], R[0].bf, zero);
  // one step = the point block in `c` against the wave's QB query blocks; the chain of the next product is issued before
  // the finished one is turned into kernel values (MFMAs and vector work of one wave overlap)
  auto step = [&](int jj, MeanOperand<PP>& c, const MeanOperand<PP>& n) {
    float av[PP];
#pragma unroll
    for (int p = 0; p < PP; ++p) av[p] = c.av[p];
    if constexpr (QB == 2) {
      const f16v nxt = chain6(a[1], c.bf, zero);
      ldf(b0 + jj + 2, c);                              // the fragments of c have had their last use
      consume<PP>(cur, av, acc[0]);
      cur = nxt;
    }
    const f16v nxt = chain6(a[0], n.bf, zero);          // (discarded after the last block)
    if constexpr (QB == 1) ldf(b0 + jj + 2, c);
    consume<PP>(cur, av, acc[QB - 1]);
    cur = nxt;
    lda(b0 + jj + 2, c);
  };
  int jj = 0;
  for (; jj + 2 <= nb; jj += 2) {
    step(jj, R[0], R[1]);
    step(jj + 1, R[1], R[0]);
  }
  if (jj < nb) step(jj, R[0], R[1]);

  // ---- sum over the 32 points a half-wave holds (lane & 31), by a halving butterfly: at the step with lane mask m a
  // lane keeps one half of its values (by its bit m) and hands the other half to its partner, so the five steps move
  // 48 + 24 + 12 + 6 + 3 values per lane pair instead of 5 x 96.  With the values flattened as k = (b 16 + r) PP + p a lane
  // ends with k = PP (lane & 31) + p: the complete sums of query row r = lane & 15 of block b = (lane & 31) >> 4.
  // (QB = 1: four halving steps and a plain exchange for mask 1; lanes 2 i and 2 i + 1 then both hold row i.)
  constexpr int NV = QB * 16 * PP;
  float v[NV];
#pragma unroll
  for (int b = 0; b < QB; ++b)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int p = 0; p < PP; ++p) {
        v[(b * 16 + 2 * i) * PP + p] = acc[b][i][p].x;
        v[(b * 16 + 2 * i + 1) * PP + p] = acc[b][i][p].y;
      }
  static_for_halving<NV, 16, (QB == 2 ? 5 : 4)>(v, ln);
  if constexpr (QB == 1) {
#pragma unroll
    for (int p = 0; p < PP; ++p) v[p] += __shfl_xor(v[p], 1, 64);
  }
  {
    const int row = QB == 2 ? (ln & 15) : (ln >> 1), b = QB == 2 ? (ln >> 4) : 0;
    const long long q = q0 + (wave * QB + b) * 32 + acc_row(row, lane);
    if ((QB == 2 || (ln & 1) == 0) && q < M) {
#pragma unroll
      for (int p = 0; p < PP; ++p) partial[((long long)blockIdx.y * M + q) * PP + p] = v[p];
    }
  }
}

__global__ void mean_mfma_reduce_kernel(const float* __restrict__ partial, int S, long long M, int P, float sf2,
                                        D16 ymean, D16 ystd, float* __restrict__ mean) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= M * P) return;
  const int p = (int)(e % P);
  double s = 0.0;                     // the partials of the training chunks are added in fp64
  for (int k = 0; k < S; ++k) s += (double)partial[(long long)k * M * P + e];
  mean[e] = (float)ymean.v[p] + (float)ystd.v[p] * (sf2 * (float)s);
}

typedef void (*mm_fn)(const float*, const float*, long long, int, F16, F16, const float*, long long, long long, float*);

template <int KP>
mm_fn mm_pick_p(int P) {
  switch (P) {
    case 1: return mean_mfma_kernel<KP, 1, 2>;
    case 2: return mean_mfma_kernel<KP, 2, 2>;
    case 3: return mean_mfma_kernel<KP, 3, 2>;
    case 4: return mean_mfma_kernel<KP, 4, 1>;
    case 5: return mean_mfma_kernel<KP, 5, 1>;
    case 6: return mean_mfma_kernel<KP, 6, 1>;
    case 7: return mean_mfma_kernel<KP, 7, 1>;
    default: return mean_mfma_kernel<KP, 8, 1>;
  }
}
typedef void (*mb_fn)(const u32x4*, const float*, int, int, F16, F16, const float*, long long, int, float*);
typedef void (*mprep_fn)(const float*, const float*, long long, int, F16, F16, u32x4*, float*, long long);
mb_fn mm_pick_bf16(int P) {
  switch (P) {
    case 1: return mean_bf16_kernel<1, 2>;
    case 2: return mean_bf16_kernel<2, 2>;
    case 3: return mean_bf16_kernel<3, 2>;
    case 4: return mean_bf16_kernel<4, 1>;
    case 5: return mean_bf16_kernel<5, 1>;
    case 6: return mean_bf16_kernel<6, 1>;
    case 7: return mean_bf16_kernel<7, 1>;
    default: return mean_bf16_kernel<8, 1>;
  }
}
mprep_fn mm_pick_prep(int P) {
  switch (P) {
    case 1: return mean_prep_kernel<1>;
    case 2: return mean_prep_kernel<2>;
    case 3: return mean_prep_kernel<3>;
    case 4: return mean_prep_kernel<4>;
    case 5: return mean_prep_kernel<5>;
    case 6: return mean_prep_kernel<6>;
    case 7: return mean_prep_kernel<7>;
    default: return mean_prep_kernel<8>;
  }
}
}  // namespace

extern "C" int gpk_predict_mean_mfma(gpk_handle h, const float* X, const float* alpha, int64_t N, int D, int P,
                                     const double* ls, double sf2, const double* center, const double* y_mean,
                                     const double* y_std, const float* Xq, int64_t M, float* mean) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, X && alpha && Xq && mean && ls && center && y_mean && y_std, "predict_mean_mfma: null pointer");
  GPK_REQUIRE(h, N >= 1 && M >= 1, "predict_mean_mfma: empty input");
  GPK_REQUIRE(h, D >= 1 && D <= 16, "predict_mean_mfma: D must be in [1, 16]");
  GPK_REQUIRE(h, P >= 1 && P <= 8, "predict_mean_mfma: P must be in [1, 8]");
  F16 sc{}, ctr{};
  D16 ym{}, ys{};
  const double s = 0.84932180028801904;   // sqrt(log2(e) / 2)
  for (int d = 0; d < D; ++d) {
    GPK_REQUIRE(h, ls[d] > 0.0, "predict_mean_mfma: length scales must be positive");
    sc.v[d] = (float)(s / ls[d]);
    ctr.v[d] = (float)center[d];
  }
  for (int p = 0; p < P; ++p) { ym.v[p] = y_mean[p]; ys.v[p] = y_std[p]; }
  const int QB = P <= 3 ? 2 : 1;   // must match mm_pick_p / mm_pick_bf16
  const int64_t nqb = (M + 128 * QB - 1) / (128 * QB);
  // split the training set so that the grid has >= ~1024 workgroups (2 resident per CU, 2 rounds)
  int64_t S = (1024 + nqb - 1) / nqb;
  // chunks of at most 2048 training points, whatever the batch size: a lane's fp32 running sums then cover 64 terms each
  // (every add rounds at the magnitude of the running sum, so the rounding of a chain grows like its length: with one
  // chunk = the whole training set, 2048 terms per lane at N = 65 536, that was the largest part of the mean's fp32 error
  // for batches of 10^6 queries - 1.3e-4 against 2.5e-5 for 10^4 queries, whose chunks are short)
  if (S < (N + 2047) / 2048) S = (N + 2047) / 2048;
  const int64_t maxS = (N + MM_TJ - 1) / MM_TJ;
  if (S > maxS) S = maxS;
  if (S < 1) S = 1;
  if (S > 65535) S = 65535;
  int64_t chunk = (N + S - 1) / S;
  chunk = (chunk + MM_TJ - 1) / MM_TJ * MM_TJ;
  S = (N + chunk - 1) / chunk;
  const size_t partial_bytes = ((size_t)S * M * P * sizeof(float) + 255) & ~(size_t)255;
  void* partial = nullptr;
  if (D > 14) {
    // D = 15, 16: depth up to 17, nine fp32 MFMAs per block, points staged through LDS
    GPK_TRY(gpk_scratch(h, partial_bytes, &partial));
    hipLaunchKernelGGL(mm_pick_p<9>(P), dim3((unsigned)nqb, (unsigned)S), dim3(256), 0, h->stream, X, alpha, (long long)N, D, sc,
                       ctr, Xq, (long long)M, (long long)chunk, (float*)partial);
  } else {
    // D <= 14: distances on the bf16 matrix cores (exact three-way operand split, depth D + 2 <= 16); the training operand
    // is prepared once per call (N x 96 bytes + alpha, behind the partial sums in the scratch block)
    const int64_t nblk = (N + 31) / 32, npad = nblk * 32;
    const size_t xb_bytes = (size_t)nblk * 192 * 16;
    GPK_TRY(gpk_scratch(h, partial_bytes + xb_bytes + (size_t)npad * P * sizeof(float), &partial));
    u32x4* xb = reinterpret_cast<u32x4*>((char*)partial + partial_bytes);
    float* al = reinterpret_cast<float*>((char*)partial + partial_bytes + xb_bytes);
    hipLaunchKernelGGL(mm_pick_prep(P), dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, h->stream, X, alpha, (long long)N, D,
                       sc, ctr, xb, al, (long long)npad);
    GPK_LAUNCH_CHECK(h);
    hipLaunchKernelGGL(mm_pick_bf16(P), dim3((unsigned)nqb, (unsigned)S), dim3(256), 0, h->stream, (const u32x4*)xb,
                       (const float*)al, (int)nblk, D, sc, ctr, Xq, (long long)M, (int)(chunk / 32), (float*)partial);
  }
  GPK_LAUNCH_CHECK(h);
  const int64_t tot = M * P;
  hipLaunchKernelGGL(mean_mfma_reduce_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream,
                     (const float*)partial, (int)S, (long long)M, P, (float)sf2, ym, ys, mean);
  GPK_LAUNCH_CHECK(h);
  return GPK_OK;
}
