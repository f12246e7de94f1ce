// K5 on the bf16 matrix pipe at fp32 accuracy: V = W Kq^T with both fp32 operands split EXACTLY into three bf16
// parts (x = x0 + x1 + x2, 8 significant bits each, by truncation; gpk_split3) and every 32 x 32 x 16 block
// product formed by six v_mfma_f32_32x32x16_bf16 (a0 b0, a0 b1, a1 b0, a1 b1, a0 b2, a2 b0: bf16 x bf16 products
// are exact, accumulation is fp32, the dropped cross terms are below 2^-24 of |a||b|, i.e. below the rounding of
// an fp32 FMA).  The fp32 MFMA runs at the vector rate (157 TF); the bf16 pipe is 16x faster, so six instructions
// per fp32-equivalent product raise the ceiling of this launch by 16/6 = 2.7x.
//
// Operand layout (gpk_split3): 16-byte chunks [row / 4][k16 block][row % 4][half h][part s] - the eight bf16 of
// part s for k = 16 kb + 8 h .. + 7 - so that one k-tile of 16 of four consecutive rows is 384 contiguous bytes =
// three whole cache lines (consecutive lanes fetch consecutive chunks, a wave instruction touches 8 lines; with
// plain row-major 96-byte pieces it touched 11-22 partial lines, and variants that fetched 16- or 32-byte pieces per
// row were 20-45 % slower: the fetch granularity is what this kernel is most sensitive to) and a lane's
// MFMA fragment (row r = lane & 31, k half h = lane >> 5) is one ds_read_b128 per part.  LDS rows are padded to
// 112 bytes (28 dwords = 4 x odd: conflict-free b128 reads over 8 consecutive rows).
//
// The kernel is the K5 launch only: 128 x 128 tiles, 4 waves of 64 x 64 (2 x 2 blocks of 32 x 32), k-tiles of 16,
// the same register-staged pipeline, scalar-based buffer loads, band-linear super-tiles in serpentine XCD order,
// heavy-first rows and lockstep k-ranges as gemm_kernel (gpk_gemm.hip), and its epilogue reduces the tile to
// per-column sums of squares (fp64) instead of storing it.
//
// Replaces the same reference lines as gpk_predict_var_inv (sklearn/gaussian_process/_gpr.py:454-485).
#include <cmath>

#include "gpk_internal.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int V16 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int V> struct IntC { static constexpr int value = V; };

constexpr int ROWB = 384;      // bytes of one k-tile (16 k) of a quad of rows: 4 rows x 2 halves x 3 parts x 16 B
// the same two constants for NPART parts per value (3: bf16 x 3, exact; 2: fp16 x 2, 22 significant bits)
template <int NPART> constexpr int rowb_v = 4 * 2 * NPART * 16;
// LDS row of a k-tile: the row's 2 NPART chunks of 16 bytes plus 16 bytes of padding (112 / 80 bytes = 4 x odd dwords:
// conflict-free ds_read_b128).  The padded 80-byte row of the fp16 x 2 split makes every ds_write_b128 2-way (banks
// (a/4) % 32, 8 contiguous lanes = rows r and r+1, which overlap in 4 of the 32 store banks: SQ_LDS_BANK_CONFLICT = 31 % of
// SQ_LDS_IDX_ACTIVE).  The swizzled form (SW = 2) has neither: 64-byte rows, no padding, chunk c of row r stored at
// position c ^ ((r >> 2) & 3) - a ds_read_b128 lane group ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...: four row quads
// whose (r >> 2) & 3 all differ) lands on 16 different 16-byte slots of the 256-byte bank row, and 8 contiguous lanes of a
// store cover 128 contiguous bytes.  Measured on one box (profiles/r02_k5_forms_ab.log): zero conflicts and 2.4 % fewer
// clock cycles, but the chip then holds 1.49 instead of 1.60 GHz under the 256 x 128-tile kernel (+5 % time); under the
// 512 x 128-tile kernel it is 0.8 % faster.  So the swizzle is used with the large tile only.
template <int SW, int NPART> constexpr int lrow_v = SW == 2 ? 64 : 2 * NPART * 16 + 16;
template <bool SWZ> __device__ __forceinline__ int lds_chunk(int row, int chunk) {
  return SWZ ? (chunk ^ ((row >> 2) & 3)) : chunk;
}
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

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

// src: rows x cols fp32 (ld elements per row, cols % 16 == 0) -> dst: rows x (cols / 16) x 96 bytes
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ src, long long rows, long long cols,
                                                     long long ld, V16* __restrict__ dst) {
  const long long hc = cols / 8;                              // half-blocks of 8 k per row
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= rows * hc) return;
  const long long row = e / hc, hb = e - row * hc;            // hb = 2 kb + h
  const float4* s4 = reinterpret_cast<const float4*>(src + row * ld + hb * 8);
  const float4 lo = s4[0], hi = s4[1];
  const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  V16 f[3];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unsigned a0, a1, a2, b0, b1, b2;
    split3(v[2 * j], a0, a1, a2);
    split3(v[2 * j + 1], b0, b1, b2);
    f[0][j] = (a0 >> 16) | b0;
    f[1][j] = (a1 >> 16) | b1;
    f[2][j] = (a2 >> 16) | (b2 & 0xffff0000u);
  }
  // chunk order: [row / 4][k16 block][row % 4][half][part]: the k-tile of four consecutive rows is 384 contiguous bytes
  V16* d = dst + (((row >> 2) * (hc >> 1) + (hb >> 1)) * 4 + (row & 3)) * 6 + (hb & 1) * 3;
  d[0] = f[0]; d[1] = f[1]; d[2] = f[2];
}

// ---- fp16 x 2 split (the optional fast form): x * scale = h0 + h1 + r with h0, h1 fp16 (rounded to nearest), |r| <= 2^-22 |x|
// for values whose second part is a normal fp16 number (|x * scale| >= 2^-3) and <= 2^-25 * 2^15 / scale absolutely
// below that; `scale` (a power of two chosen by the caller so that max |x * scale| <= 2^15) puts the operand's
// largest entries at the top of fp16's range.  Chunk order [row / 4][k16 block][row % 4][half][part]: 64 bytes per row
// and k-tile, 256 contiguous bytes per quad of rows.
__global__ __launch_bounds__(256) void split2_kernel(const float* __restrict__ src, long long rows, long long cols,
                                                     long long ld, float scale, V16* __restrict__ dst) {
  const long long hc = cols / 8;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= rows * hc) return;
  const long long row = e / hc, hb = e - row * hc;
  const float4* s4 = reinterpret_cast<const float4*>(src + row * ld + hb * 8);
  const float4 lo = s4[0], hi = s4[1];
  const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  f16x8 p0, p1;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = v[j] * scale;
    const _Float16 h0 = (_Float16)x;
    p0[j] = h0;
    p1[j] = (_Float16)(x - (float)h0);
  }
  V16* d = dst + (((row >> 2) * (hc >> 1) + (hb >> 1)) * 4 + (row & 3)) * 4 + (hb & 1) * 2;
  d[0] = __builtin_bit_cast(V16, p0);
  d[1] = __builtin_bit_cast(V16, p1);
}

// max |A_ij| over the lower triangle (j <= i) of a square fp32 matrix: one row per workgroup, one atomic per row
// (the bit pattern of a non-negative float orders like the value)
__global__ __launch_bounds__(256) void tril_absmax_kernel(const float* __restrict__ A, long long n, long long lda,
                                                          unsigned* __restrict__ out) {
  const long long i = blockIdx.x;
  float m = 0.f;
  for (long long j = threadIdx.x; j <= i; j += 256) m = fmaxf(m, fabsf(A[i * lda + j]));
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

struct SParams {
  const char* A;        // W, split layout, Np rows
  const char* B;        // Kq, split layout, Mp rows
  double* out;          // [ntm][Mp] partial column sums of squares
  long long rsa, rsb;   // bytes between consecutive quads of rows (Np * 24)
  long long Mp;
  int ntm, ntn, nst;
  float alpha;
};

template <int NP>
__device__ __forceinline__ void load3(const char* __restrict__ ubase, const unsigned (&voff)[NP], V16 (&r)[NP]) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ubase), 0, 0x7fffffff, 0x00020000);
#pragma unroll
  for (int p = 0; p < NP; ++p) r[p] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[p], 0, 0);
}

// WR wave rows x 2 wave columns of 64 x 64 per workgroup: WR = 2 -> 128 x 128 tile, 4 waves, 2 workgroups per CU;
// WR = 4 -> 256 x 128 tile, 8 waves, 1 workgroup per CU (25 % less operand staging per MFMA)
// NPART = 3: bf16 parts, six products per block (exact); NPART = 2: fp16 parts, three products (a1 b0, a0 b1, a0 b0)
// AB: 32-row blocks per wave.  2: 64 x 64 per wave.  4 (with WR = 4): 128 x 64 per wave, 512 x 128 tiles, one
// workgroup per CU at 2 waves per SIMD (256 VGPRs) - 17 % fewer operand bytes from L2 per product and 0.5 instead of
// 0.67 fragment reads per MFMA: the launch is limited by the clock the chip holds under this load, and less data
// movement per product is what raises it (9 % faster at the headline shape).  Needs >= 512 tiles to fill the chip.
template <int WR, int NPART = 3, int AB = 2>
__global__ __launch_bounds__(WR * 128, AB == 4 ? 1 : 2) void k5_split_kernel(SParams p) {
  constexpr bool SWZ = AB == 4;                                   // swizzled 64-byte LDS rows (see lrow_v)
  constexpr int LROW = lrow_v<SWZ ? 2 : 3, NPART>, ROWB = rowb_v<NPART>, CPR = 2 * NPART;      // (shadow the bf16 x 3 constants)
  constexpr int TMR = 32 * AB * WR;                                // tile rows
  constexpr int SSZ = (TMR + 128) * LROW;                          // one LDS stage: [A k-tile | B k-tile]
  constexpr int BOFF = TMR * LROW;                                 // B inside a stage
  // byte offsets of the A / B k-tile of LDS buffer `buf`.  AB = 4: [B0 | B1 | A0 | A1], so that every fragment read is
  // within 64 KiB (the ds offset field) of one base register per operand and swizzle variant; else [A0 | B0 | A1 | B1].
  auto AO = [](int buf) constexpr { return AB == 4 ? 2 * 128 * LROW + buf * TMR * LROW : buf * SSZ; };
  auto BO = [](int buf) constexpr { return AB == 4 ? buf * 128 * LROW : buf * SSZ + BOFF; };
  __shared__ __attribute__((aligned(16))) char lds[2 * SSZ];
  constexpr int NT = WR * 128, NCH = (TMR + 128) * CPR, NQ = (NCH + NT - 1) / NT;   // threads, chunks
  constexpr int GSZ = WR == 2 ? 64 : 32;                           // resident workgroups per XCD = tiles per group
  constexpr int BH = 1024 / TMR;                                   // band height in tile rows (1024 matrix rows)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int row_w = wm * 32 * AB, col_w = wn * 64;

  // ---- tile mapping: bands of 1024 rows walked column by column in groups of GSZ tiles, serpentine over XCDs
  const int ntm = p.ntm * 128 / TMR;                               // tile rows of this configuration
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int grp = j / GSZ, slot = j - grp * GSZ;
  const int st = grp * 8 + ((grp & 1) ? 7 - xcd : xcd);
  if (st >= p.nst) return;
  const int per = BH * p.ntn, nfull = ntm / BH, hlast = ntm - nfull * BH, total = ntm * p.ntn;
  auto place = [&](int t, int& row, int& col, int& band) {
    band = min(t / per, nfull);
    const int idx = t - band * per, hh = band < nfull ? BH : hlast;
    col = idx / hh;
    row = band * BH + (idx - col * hh);
  };
  const int t = st * GSZ + slot;
  if (t >= total) return;
  int tm, tn, band;
  place(t, tm, tn, band);
  // heavy first: row r of the enumeration is tile row ntm-1-r (its k-range is (row + 1) * TMR: longest first);
  // every tile of the group runs to the end of the group's longest row, the one enumerated first (W is zero
  // beyond a row's own range for 16 tiles of 128; a group spans at most two bands = 2048 rows)
  const int band0 = min(st * GSZ / per, nfull), band1 = min(min(st * GSZ + GSZ - 1, total - 1) / per, nfull);
  tm = ntm - 1 - tm;
  // (narrow grids: a group spans more than two bands - beyond the zero band of W - and every tile keeps its own
  // k-range)
  const int rhi = (band1 - band0 <= 1) ? ntm - 1 - band0 * BH : tm;
  const int nkt = (rhi + 1) * (TMR / 16);        // k-tiles of 16

  const int row0 = tm * TMR, col0 = tn * 128;
  f16v acc[AB][2];
#pragma unroll
  for (int a = 0; a < AB; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  // a k-tile is TMR * 6 chunks of 16 B of A and 768 of B; chunk g = tid + NT q of them (A first) is this thread's
  // q-th; which operand it belongs to - and whether it exists at all - is wave-uniform
  unsigned goff[NQ], lofs[NQ];
  bool isb[NQ], live[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int g = tid + NT * q;
    live[q] = __builtin_amdgcn_readfirstlane(g < NCH ? 1 : 0) != 0;
    isb[q] = __builtin_amdgcn_readfirstlane(g >= TMR * CPR ? 1 : 0) != 0;
    const int c = live[q] ? (isb[q] ? g - TMR * CPR : g) : 0, row = c / CPR, w = c - row * CPR;
    goff[q] = (unsigned)((long long)(row >> 2) * p.rsa + ((row & 3) * CPR + w) * 16);     // rsa: bytes between row quads
    lofs[q] = (isb[q] ? BOFF : 0) + row * LROW + lds_chunk<SWZ>(row, w) * 16;
  }
  // fp16 x 2 (four chunks per row, NT a multiple of four, the A / B boundary on a multiple of NT): chunk q of a thread is
  // its chunk 0 moved down NT / 4 rows per q - one vector offset plus a scalar step per q for the global address, one
  // LDS offset plus constants for the store (NQ - 1 fewer address registers of each kind)
  constexpr bool AFF = NPART == 2;
  constexpr int QA = TMR * CPR / NT, QROWS = NT / CPR;
  static_assert(!AFF || ((TMR * CPR) % NT == 0 && NCH % NT == 0 && NT % CPR == 0), "affine chunk addressing");
  const int gstep = (int)((QROWS / 4) * p.rsa);                     // bytes between a thread's consecutive chunks (rsa == rsb)
  const char* ua = p.A + (long long)(row0 >> 2) * p.rsa;
  const char* ub = p.B + (long long)(col0 >> 2) * p.rsb;
  // Register ring of RING k-tiles in flight: the fetch of k-tile kt+1+RING is issued when k-tile kt+1 leaves its
  // ring slot for LDS.  The loop is unrolled by 4: slot and LDS buffer indices are constants.
#ifndef GPK_K5S_SCHED
#define GPK_K5S_SCHED 1
#endif
#ifndef GPK_K5S_RING
#define GPK_K5S_RING 2
#endif
  constexpr int RING = GPK_K5S_RING;
  V16 rr[RING][NQ];
  auto fetch = [&](int tile, auto slotc) {
    constexpr int sl = decltype(slotc)::value;
    const long long ko = (long long)min(tile, nkt - 1) * ROWB;
    if constexpr (AFF) {
      const __amdgpu_buffer_rsrc_t rsa_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ua + ko), 0, 0x7fffffff, 0x00020000);
      const __amdgpu_buffer_rsrc_t rsb_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ub + ko), 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        rr[sl][q] = q < QA ? __builtin_amdgcn_raw_buffer_load_b128(rsa_, goff[0], q * gstep, 0)
                           : __builtin_amdgcn_raw_buffer_load_b128(rsb_, goff[0], (q - QA) * gstep, 0);
      return;
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (NCH % NT != 0 && !live[q]) continue;      // (only the last chunk of the 256-row configuration can be absent)
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>((isb[q] ? ub : ua) + ko), 0, 0x7fffffff, 0x00020000);
      rr[sl][q] = __builtin_amdgcn_raw_buffer_load_b128(rs, goff[q], 0, 0);
    }
  };
  auto stage = [&](auto bufc, auto slotc) {
    constexpr int sl = decltype(slotc)::value, bi = decltype(bufc)::value;
    if constexpr (AFF) {
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        *reinterpret_cast<V16*>(lds + lofs[0] + (q < QA ? AO(bi) + q * QROWS * LROW : BO(bi) + (q - QA) * QROWS * LROW)) = rr[sl][q];
      return;
    }
    static_assert(AFF || AB != 4, "the [B | A] buffer order is used with affine chunk addressing only");
    char* buf = lds + bi * SSZ;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      if (NCH % NT == 0 || live[q]) *reinterpret_cast<V16*>(buf + lofs[q]) = rr[sl][q];
  };
#ifndef GPK_K5S_FPRE
#define GPK_K5S_FPRE 1
#endif
  const int fr = lane & 31, fh = lane >> 5;
  auto mfmas = [&](const bf16x8 (&af)[AB][NPART], const bf16x8 (&bf)[2][NPART]) {
    // smallest terms first
#pragma unroll
    for (int a = 0; a < AB; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        f16v c = acc[a][b];
        if constexpr (NPART == 2) {
          const f16x8 a0 = __builtin_bit_cast(f16x8, af[a][0]), a1 = __builtin_bit_cast(f16x8, af[a][1]);
          const f16x8 b0 = __builtin_bit_cast(f16x8, bf[b][0]), b1 = __builtin_bit_cast(f16x8, bf[b][1]);
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c, 0, 0, 0);
        } else {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][2], bf[b][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][1], bf[b][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][1], bf[b][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][0], c, 0, 0, 0);
        }
        acc[a][b] = c;
      }
  };
  auto frags = [&](auto bufc, bf16x8 (&af)[AB][NPART], bf16x8 (&bf)[2][NPART]) {
    constexpr int bi = decltype(bufc)::value;
#pragma unroll
    for (int a = 0; a < AB; ++a)
#pragma unroll
      for (int s = 0; s < NPART; ++s)
        af[a][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const V16*>(lds + AO(bi) + (row_w + 32 * a + fr) * LROW + lds_chunk<SWZ>(fr, fh * NPART + s) * 16));
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int s = 0; s < NPART; ++s)
        bf[b][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const V16*>(lds + BO(bi) + (col_w + 32 * b + fr) * LROW + lds_chunk<SWZ>(fr, fh * NPART + s) * 16));
  };
#if GPK_K5S_FPRE
  // Fragments one k-tile ahead: while the MFMAs of k-tile kt run out of registers F[kt & 1], the fragments of k-tile
  // kt+1 are read from LDS buffer (kt+1) & 1 into F[(kt+1) & 1] and k-tile kt+2 is written into LDS buffer kt & 1
  // (its previous content, k-tile kt, was read during iteration kt-1; the barrier at the end of each iteration
  // separates the two).  No MFMA waits for LDS.
  static_assert(RING == 2, "the fragment-prefetch pipeline is written for a ring of two k-tiles");
  // (AB = 4: the A fragments are single-buffered - a row block's fragments of the next k-tile replace the ones its
  // MFMAs have just used - which keeps the 128 x 64 wave tile inside 256 registers)
  constexpr int NFA = AB == 4 ? 1 : 2;
  bf16x8 FA[NFA][AB][NPART], FB[2][2][NPART];
  fetch(0, IntC<0>{});
  stage(IntC<0>{}, IntC<0>{});
  fetch(1, IntC<1>{});
  stage(IntC<1>{}, IntC<1>{});
  fetch(2, IntC<0>{});
  fetch(3, IntC<1>{});
  __syncthreads();
  frags(IntC<0>{}, FA[0], FB[0]);
  __syncthreads();
  // (AB = 4: the last row block's A fragments ARE double-buffered, so that an iteration can end with that block's MFMAs
  // and its LDS reads are long complete at the barrier)
  bf16x8 FA3[2][NPART];
  if constexpr (AB == 4) {
#pragma unroll
    for (int sp = 0; sp < NPART; ++sp) FA3[0][sp] = FA[0][AB - 1][sp];
  }
  auto body = [&](int kt, auto ksc) {
    constexpr int KS = decltype(ksc)::value, cur = KS & 1;
    if constexpr (AB == 4) {
      // The 128 x 64 wave tile, issue order written out and fenced (sched_barrier: nothing moves across): MFMAs from the
      // first cycle after the barrier (their operands are in registers); k-tile kt+2 goes from the ring registers to LDS
      // buffer `cur` one store per three MFMAs, each global load of k-tile kt+4 behind the store that frees its
      // registers; a row block's A fragments of k-tile kt+1 are read behind that block's MFMAs, the B fragments and the
      // last block's (double-buffered) A fragments before the last twelve MFMAs, which end the iteration.
      static_assert(AB != 4 || (NPART == 2 && NQ == 5 && AFF), "the 128 x 64 wave tile is written for the fp16 x 2 split");
      const long long ko = (long long)min(kt + 4, nkt - 1) * ROWB;
      const __amdgpu_buffer_rsrc_t rsa_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ua + ko), 0, 0x7fffffff, 0x00020000);
      const __amdgpu_buffer_rsrc_t rsb_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ub + ko), 0, 0x7fffffff, 0x00020000);
      auto st = [&](auto qc) {
        constexpr int q = decltype(qc)::value;
        *reinterpret_cast<V16*>(lds + lofs[0] + (q < QA ? AO(cur) + q * QROWS * LROW : BO(cur) + (q - QA) * QROWS * LROW)) = rr[cur][q];
      };
      auto ld = [&](auto qc) {
        constexpr int q = decltype(qc)::value;
        rr[cur][q] = q < QA ? __builtin_amdgcn_raw_buffer_load_b128(rsa_, goff[0], q * gstep, 0)
                            : __builtin_amdgcn_raw_buffer_load_b128(rsb_, goff[0], (q - QA) * gstep, 0);
      };
      auto rdA = [&](int a, bf16x8 (&dst)[NPART]) {
#pragma unroll
        for (int sp = 0; sp < NPART; ++sp)
          dst[sp] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const V16*>(lds + AO(cur ^ 1) + (row_w + 32 * a + fr) * LROW + lds_chunk<SWZ>(fr, fh * NPART + sp) * 16));
      };
      auto rdB = [&](int b_, bf16x8 (&dst)[NPART]) {
#pragma unroll
        for (int sp = 0; sp < NPART; ++sp)
          dst[sp] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const V16*>(lds + BO(cur ^ 1) + (col_w + 32 * b_ + fr) * LROW + lds_chunk<SWZ>(fr, fh * NPART + sp) * 16));
      };
      auto mm = [&](int a, int b, const bf16x8 (&fa)[NPART], const bf16x8 (&fb)[NPART]) {
        const f16x8 a0 = __builtin_bit_cast(f16x8, fa[0]), a1 = __builtin_bit_cast(f16x8, fa[1]);
        const f16x8 b0 = __builtin_bit_cast(f16x8, fb[0]), b1 = __builtin_bit_cast(f16x8, fb[1]);
        f16v c = acc[a][b];
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c, 0, 0, 0);       // smallest terms first
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c, 0, 0, 0);
        acc[a][b] = c;
      };
      // column block 0 first (its B fragments are single-buffered: FB[0][0], reloaded once its four products are done),
      // then column block 1 (FB[cur][1]; the row blocks' A fragments are reloaded behind their second product)
#define GPK_FENCE() __builtin_amdgcn_sched_barrier(0)
      mm(0, 0, FA[0][0], FB[0][0]); GPK_FENCE();
      st(IntC<0>{}); GPK_FENCE();
      mm(1, 0, FA[0][1], FB[0][0]); GPK_FENCE();
      st(IntC<1>{}); GPK_FENCE();
      mm(2, 0, FA[0][2], FB[0][0]); GPK_FENCE();
      st(IntC<2>{}); GPK_FENCE();
      mm(3, 0, FA3[cur], FB[0][0]); GPK_FENCE();
      st(IntC<3>{}); rdB(0, FB[0][0]); GPK_FENCE();
      mm(0, 1, FA[0][0], FB[cur][1]); GPK_FENCE();
      st(IntC<4>{}); ld(IntC<0>{}); rdA(0, FA[0][0]); GPK_FENCE();
      mm(1, 1, FA[0][1], FB[cur][1]); GPK_FENCE();
      ld(IntC<1>{}); ld(IntC<2>{}); rdA(1, FA[0][1]); GPK_FENCE();
      mm(2, 1, FA[0][2], FB[cur][1]); GPK_FENCE();
      ld(IntC<3>{}); ld(IntC<4>{}); rdA(2, FA[0][2]); rdB(1, FB[cur ^ 1][1]); rdA(3, FA3[cur ^ 1]); GPK_FENCE();
      mm(3, 1, FA3[cur], FB[cur][1]); GPK_FENCE();
#undef GPK_FENCE
      __syncthreads();
      return;
    }
    stage(IntC<cur>{}, IntC<cur>{});            // k-tile kt+2 (ring slot (kt+2) % 2 = cur)
    fetch(kt + 4, IntC<cur>{});
    frags(IntC<(cur ^ 1)>{}, FA[(cur ^ 1) % NFA], FB[cur ^ 1]);
    mfmas(FA[cur % NFA], FB[cur]);
#if GPK_K5S_SCHED
    constexpr int NMF = (NPART == 3 ? 6 : 3) * AB * 2, NRD = NPART * (AB + 2);
    __builtin_amdgcn_sched_group_barrier(0x200, NQ, 0);
#pragma unroll
    for (int g = 0; g < NQ; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, (NMF + NQ - 1) / NQ, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, (NRD + NQ - 1) / NQ, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
#endif
    __syncthreads();
  };
#else
  fetch(0, IntC<0>{});
  stage(IntC<0>{}, IntC<0>{});
  fetch(1, IntC<1 % RING>{});
  if constexpr (RING >= 2) fetch(2, IntC<2 % RING>{});
  if constexpr (RING >= 3) fetch(3, IntC<3 % RING>{});
  if constexpr (RING >= 4) { fetch(4, IntC<0>{}); }
  __syncthreads();
  // iteration kt (kt % RING == KS): k-tile kt+1 goes from ring slot (KS+1) % RING to LDS buffer (kt+1) & 1, that
  // slot is refilled with k-tile kt+1+RING, and k-tile kt is multiplied out of LDS buffer kt & 1
  auto body = [&](int kt, auto ksc) {
    constexpr int KS = decltype(ksc)::value, cur = KS & 1, sl = (KS + 1) % RING;
    stage(IntC<(cur ^ 1)>{}, IntC<sl>{});
    fetch(kt + 1 + RING, IntC<sl>{});
    bf16x8 af[AB][NPART], bf[2][NPART];
    frags(IntC<cur>{}, af, bf);
    mfmas(af, bf);
#if GPK_K5S_SCHED
    // issue order: LDS writes of the next k-tile, all twelve fragment reads, then the MFMAs with the six
    // buffer loads spread among them (the scheduler otherwise trickles the reads between MFMAs and waits five times)
    __builtin_amdgcn_sched_group_barrier(0x200, NQ, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, NPART * (AB + 2), 0);
#pragma unroll
    for (int g = 0; g < NQ; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, ((NPART == 3 ? 6 : 3) * AB * 2 + NQ - 1) / NQ, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
#endif
    __syncthreads();
  };
#endif
  int kt = 0;
  for (; kt + 3 < nkt; kt += 4) {
    body(kt, IntC<0 % RING>{});
    body(kt + 1, IntC<1 % RING>{});
    body(kt + 2, IntC<2 % RING>{});
    body(kt + 3, IntC<3 % RING>{});
  }
  // nkt is a multiple of 8 (k-ranges are whole 128-tiles): no remainder

  // ---- epilogue: per-column sums of squares of the tile (fp64), out[tile row][column]
  double* red = reinterpret_cast<double*>(lds);      // [WR][128]; the k-loop ended with a barrier
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < AB; ++a)
#pragma unroll
      for (int i = 0; i < 16; ++i) { const float v = p.alpha * acc[a][b][i]; s = __builtin_fmaf(v, v, s); }
    s += __shfl_xor(s, 32, 64);
    if (lane < 32) red[wm * 128 + col_w + 32 * b + lane] = (double)s;
  }
  __syncthreads();
  if (tid < 128) {
    double t2 = 0.0;
#pragma unroll
    for (int w = 0; w < WR; ++w) t2 += red[w * 128 + tid];
    p.out[(long long)tm * p.Mp + col0 + tid] = t2;      // one partial row per tile row of TMR matrix rows
  }
}


// ---- second form of the same launch: 16 x 16 x 32 MFMAs, two product terms per instruction, LDS filled by DMA --------
// The six products of a block pair up into three v_mfma_f32_16x16x32_bf16: the instruction's k = 32 is used as
// [k16 of one part | k16 of another part], with the B operand carrying the complementary parts -
//     (a0 b1 + a1 b0):  A = [a0 | a1], B = [b1 | b0]      (a0 b2 + a2 b0):  A = [a0 | a2], B = [b2 | b0]
//     (a0 b0 + a1 b1):  A = [a0 | a1], B = [b0 | b1]
// (lane l of a fragment holds row l & 15 and the eight k of k-group l >> 4: groups 0, 1 = the two halves of the first
// part's k16, groups 2, 3 = those of the second part).  Same exact products, same fp32 accumulation, smallest terms
// first; 3 x 16 cycles per 16 x 16 block and k16 = the 6 x 32 cycles per 32 x 32 block of the first form, but the
// 16 x 16 x 32 shape holds a higher clock under this load (MI355X_MICROARCH.md, DVFS item 7).  Two A fragment
// types and three B types per block: 20 ds_read_b128 per wave and k-tile (64 x 64 per wave) instead of 12.
// With 96-byte LDS rows in the global chunk order ([row][half][part]) every one of those reads is conflict-free
// (checked exhaustively over the four lane groups of ds_read_b128), so the LDS image of a k-tile IS the global
// image of its row quads: the stage is filled by buffer_load_dwordx4 ... lds (1 KiB per wave instruction, no
// registers, no ds_write), 36 of them per 256 x 128 tile and k-tile, into a ring of four 36 KiB stages with three
// k-tiles in flight across the one barrier per k-tile.
constexpr int V2_TM = 256, V2_TN = 128;
constexpr int V2_STAGE = (V2_TM + V2_TN) * 96;      // 36864 bytes: [A k-tile | B k-tile]
constexpr int V2_BOFF = V2_TM * 96;
constexpr int V2_NSTG = 4;
// (36 wave instructions of 1 KiB per stage: 24 of A, 12 of B)

typedef float f4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__global__ __launch_bounds__(512, 2) void k5_split16_kernel(SParams p) {
  __shared__ __attribute__((aligned(1024))) char lds[V2_NSTG * V2_STAGE];
  constexpr int GSZ = 32, BH = 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- tile mapping (as k5_split_kernel<4>): bands of 1024 rows, groups of 32 tiles, serpentine over XCDs, heavy first
  const int ntm = p.ntm / 2;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int grp = j / GSZ, slot = j - grp * GSZ;
  const int st = grp * 8 + ((grp & 1) ? 7 - xcd : xcd);
  if (st >= p.nst) return;
  const int per = BH * p.ntn, nfull = ntm / BH, hlast = ntm - nfull * BH, total = ntm * p.ntn;
  const int t = st * GSZ + slot;
  if (t >= total) return;
  int tm, tn;
  {
    const int band = min(t / per, nfull);
    const int idx = t - band * per, hh = band < nfull ? BH : hlast;
    tn = idx / hh;
    tm = band * BH + (idx - tn * hh);
  }
  const int band0 = min(st * GSZ / per, nfull), band1 = min(min(st * GSZ + GSZ - 1, total - 1) / per, nfull);
  tm = ntm - 1 - tm;
  const int rhi = (band1 - band0 <= 1) ? ntm - 1 - band0 * BH : tm;
  const int nkt = (rhi + 1) * (V2_TM / 16);        // k-tiles of 16: at least 16
  const int row0 = tm * V2_TM, col0 = tn * V2_TN;

  // ---- DMA plan: wave instruction i = wave + 8 q (q = 0 .. 4) of the stage copies 64 consecutive 16-byte chunks;
  // chunk c of an operand's k-tile lives at quad (c / 24) * rs + (c % 24) * 16 in global memory and at c * 16 in LDS
  unsigned dvoff[5];
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    const int i = wave + 8 * q;                                   // wave-uniform
    const int c = ((i < 24 ? i : i - 24) * 64 + lane);
    const int quad = c / 24, within = c - quad * 24;
    dvoff[q] = (unsigned)((long long)quad * (i < 24 ? p.rsa : p.rsb) + within * 16);
  }
  // (instructions 32 .. 35 belong to waves 0 .. 3; waves 4 .. 7 repeat them - same source, same destination, same
  // bytes - so that every wave issues five per k-tile and the loop needs no branch: q = 4 maps wave w to i = 32 + (w & 3))
  {
    const int i = 32 + (wave & 3);
    const int c = (i - 24) * 64 + lane;
    const int quad = c / 24, within = c - quad * 24;
    dvoff[4] = (unsigned)((long long)quad * p.rsb + within * 16);
  }
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(p.A + (long long)(row0 >> 2) * p.rsa), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(p.B + (long long)(col0 >> 2) * p.rsb), 0, 0x7fffffff, 0x00020000);
  // k-tiles beyond the tile's range are clamped to the last one: a redundant copy into a stage nobody reads any more
  // keeps the loop free of branches and the vmcnt bookkeeping uniform (five DMAs per wave and k-tile, always)
  auto dma = [&](int kt) {
    char* stg = lds + (kt & (V2_NSTG - 1)) * V2_STAGE;
    const int so = min(kt, nkt - 1) * ROWB;                       // k-tile kt of every quad: + 384 bytes per k-tile
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(stg + wave * 1024), 16, dvoff[0], so, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(stg + 8192 + wave * 1024), 16, dvoff[1], so, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(stg + 16384 + wave * 1024), 16, dvoff[2], so, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(stg + 24576 + wave * 1024), 16, dvoff[3], so, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(stg + 32768 + (wave & 3) * 1024), 16, dvoff[4], so, 0, 0);
  };

  // ---- fragment addresses: row r = lane & 15, k-group g = lane >> 4 -> chunk (half g & 1, part s(g))
  const int fr = lane & 15, g = lane >> 4, fh = g & 1, hi = g >> 1;
  const int o01 = fr * 96 + (fh * 3 + (hi ? 1 : 0)) * 16;        // [part 0 | part 1]
  const int o02 = fr * 96 + (fh * 3 + (hi ? 2 : 0)) * 16;        // [part 0 | part 2]
  const int o10 = fr * 96 + (fh * 3 + (hi ? 0 : 1)) * 16;        // [part 1 | part 0]
  const int o20 = fr * 96 + (fh * 3 + (hi ? 0 : 2)) * 16;        // [part 2 | part 0]
  const int abase = wm * 64 * 96, bbase = V2_BOFF + wn * 64 * 96;
  auto ld = [&](const char* q) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const V16*>(q)); };

  f4v acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f4v{0.f, 0.f, 0.f, 0.f};

  bf16x8 AX[4], AY[4];                  // A fragments of the current k-tile, one row block each: [a0|a1], [a0|a2]
  bf16x8 BU[2][4], BV[2][4], BZ[2][4];  // B fragments, current and next k-tile: [b1|b0], [b2|b0], [b0|b1]

  dma(0);
  dma(1);
  dma(2);
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");          // k-tile 0 has landed (this wave's part)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const char* q = lds + bbase + b * 16 * 96;
    BU[0][b] = ld(q + o10); BV[0][b] = ld(q + o20); BZ[0][b] = ld(q + o01);
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const char* q = lds + abase + a * 16 * 96;
    AX[a] = ld(q + o01); AY[a] = ld(q + o02);
  }

#ifndef GPK_K5S2_SCHED
#define GPK_K5S2_SCHED 1
#endif
  auto body = [&](int kt, auto curc) {
    constexpr int cur = decltype(curc)::value, nxt = cur ^ 1;
    // k-tile kt+1 must be complete in LDS for every wave before anybody reads it; k-tile kt+2 stays in flight
    asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    dma(kt + 3);                        // into the stage whose fragments were read two barriers ago
    const char* nb = lds + ((kt + 1) & (V2_NSTG - 1)) * V2_STAGE;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      // the next k-tile's B fragments of column block a, read while this row block multiplies
      {
        const char* q = nb + bbase + a * 16 * 96;
        BU[nxt][a] = ld(q + o10); BV[nxt][a] = ld(q + o20); BZ[nxt][a] = ld(q + o01);
      }
      // smallest terms first: (a0 b2 + a2 b0), (a0 b1 + a1 b0), (a0 b0 + a1 b1)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AY[a], BV[cur][b], acc[a][b], 0, 0, 0);
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AX[a], BU[cur][b], acc[a][b], 0, 0, 0);
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AX[a], BZ[cur][b], acc[a][b], 0, 0, 0);
      // this row block's A fragments of the next k-tile replace the ones just used
      {
        const char* q = nb + abase + a * 16 * 96;
        AX[a] = ld(q + o01); AY[a] = ld(q + o02);
      }
    }
#if GPK_K5S2_SCHED
    // issue order: the 20 fragment reads of the next k-tile and the five DMAs spread evenly under the first 40 of the 48
    // MFMAs (left alone, the scheduler puts all the reads behind the last MFMA: every wave then hits LDS at once and
    // the next k-tile starts with a wait)
#pragma unroll
    for (int i = 0; i < 20; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      if (i % 4 == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
#endif
  };
  for (int kt = 0; kt < nkt; kt += 2) {   // nkt is a multiple of 16
    body(kt, IntC<0>{});
    body(kt + 1, IntC<1>{});
  }

  // ---- epilogue: per-column sums of squares of the tile (fp64), out[tile row][column]
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // the clamped tail DMAs and the last fragment reads
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  double* red = reinterpret_cast<double*>(lds);      // [4][128]
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int i = 0; i < 4; ++i) { const float v = p.alpha * acc[a][b][i]; s = __builtin_fmaf(v, v, s); }
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (lane < 16) red[wm * 128 + wn * 64 + b * 16 + lane] = (double)s;
  }
  __syncthreads();
  if (tid < 128) {
    double t2 = 0.0;
#pragma unroll
    for (int w = 0; w < 4; ++w) t2 += red[w * 128 + tid];
    p.out[(long long)tm * p.Mp + col0 + tid] = t2;
  }
}

}  // namespace

extern "C" int gpk_split3(gpk_handle h, const float* src, int64_t rows, int64_t cols, int64_t ld, void* dst) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, src && dst, "split3: null pointer");
  GPK_REQUIRE(h, rows >= 4 && rows % 4 == 0 && cols >= 16 && cols % 16 == 0 && ld >= cols && ld % 4 == 0,
              "split3: rows must be a multiple of 4 and cols a multiple of 16");
  GPK_REQUIRE(h, ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0, "split3: buffers must be 16-byte aligned");
  const long long n = rows * (cols / 8);
  hipLaunchKernelGGL(split3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, src, (long long)rows,
                     (long long)cols, (long long)ld, (V16*)dst);
  GPK_LAUNCH_CHECK(h);
  return GPK_OK;
}

extern "C" int gpk_tril_absmax(gpk_handle h, const float* A, int64_t n, int64_t lda, double* out) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, A && out && n >= 1 && lda >= n && n < (1ll << 31), "tril_absmax: bad argument");
  unsigned* d = reinterpret_cast<unsigned*>(h->d_small);
  GPK_CHECK_HIP(h, hipMemsetAsync(d, 0, sizeof(unsigned), h->stream));
  hipLaunchKernelGGL(tril_absmax_kernel, dim3((unsigned)n), dim3(256), 0, h->stream, A, (long long)n, (long long)lda, d);
  GPK_LAUNCH_CHECK(h);
  unsigned bits = 0;
  GPK_CHECK_HIP(h, hipMemcpyAsync(&bits, d, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
  GPK_CHECK_HIP(h, hipStreamSynchronize(h->stream));
  float f;
  memcpy(&f, &bits, sizeof f);
  *out = (double)f;
  return GPK_OK;
}

extern "C" int gpk_split2(gpk_handle h, const float* src, int64_t rows, int64_t cols, int64_t ld, double scale, void* dst) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, src && dst, "split2: null pointer");
  GPK_REQUIRE(h, rows >= 4 && rows % 4 == 0 && cols >= 16 && cols % 16 == 0 && ld >= cols && ld % 4 == 0,
              "split2: rows must be a multiple of 4 and cols a multiple of 16");
  GPK_REQUIRE(h, ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0, "split2: buffers must be 16-byte aligned");
  int ex = 0;
  GPK_REQUIRE(h, scale > 0.0 && std::frexp(scale, &ex) == 0.5, "split2: scale must be a power of two");
  const long long n = rows * (cols / 8);
  hipLaunchKernelGGL(split2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, src, (long long)rows,
                     (long long)cols, (long long)ld, (float)scale, (V16*)dst);
  GPK_LAUNCH_CHECK(h);
  return GPK_OK;
}

extern "C" int gpk_predict_var_inv_split2(gpk_handle h, const float* X, int64_t N, int D, const double* ls, double sf2,
                                          const void* W2, double w_scale, int64_t Np, const float* Xq, int64_t M,
                                          double kss, double floor_, float* work, void* work2, double* var) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, X && W2 && Xq && work && work2 && var, "predict_var_inv_split2: null pointer");
  GPK_REQUIRE(h, N >= 1 && M >= 1 && Np == gpk_padded(N), "predict_var_inv_split2: Np must equal gpk_padded(N)");
  GPK_REQUIRE(h, h->batch == 1, "predict_var_inv_split2: not available in batched mode");
  GPK_REQUIRE(h, w_scale > 0.0 && sf2 > 0.0, "predict_var_inv_split2: scales must be positive");
  const int64_t Mp = gpk_padded(M);
  const int ntm = (int)(Np / 128), ntn = (int)(Mp / 128);
  // (buffer offsets are 32-bit: a tile's last row quad lies (TMR / 4) * 16 Np bytes beyond its first - at most 2048 Np)
  GPK_REQUIRE(h, (long long)ntm * ntn < (1ll << 30) && Np * 2048 < (1ll << 31), "predict_var_inv_split2: size too large");
  // K* in (0, sf2]: the power of two that puts sf2 just below 2^15
  const double k_scale = std::ldexp(1.0, 14 - (int)std::floor(std::log2(sf2)));
  if (D <= 16 && Mp / 64 < 65536) {
    // K* computed and written in split form in one pass (`work` stays unused)
    GPK_TRY(gpk_cross_split2(h, Xq, M, X, N, D, ls, sf2, k_scale, work2));
  } else {
    GPK_TRY(gpk_cross_gram_t(h, GPK_F32, Xq, M, X, N, D, ls, sf2, work, Np));
    GPK_TRY(gpk_split2(h, work, Mp, Np, Np, k_scale, work2));
  }
  // 512 x 128 tiles (128 x 64 per wave) when they come in at least two rounds of the 256 CUs; 256 x 128 or 128 x 128
  // tiles (64 x 64 per wave) otherwise.  Option "k5_split2_tile": 0 = this rule, 1 = always 64 x 64 per wave, 2 = 512 x 128
  // whenever Np allows.
  const bool big_ok = Np % 512 == 0;
  const bool big = big_ok && (h->k5_split2_tile == 2 || (h->k5_split2_tile == 0 && (Np / 512) * (long long)ntn >= 512));
  const int ab = big ? 4 : 2;
  const int wr = big ? 4 : ((Np % 256 == 0) ? 4 : 2);
  const int ntmT = (int)(Np / (32 * ab * wr)), gsz = wr == 2 ? 64 : 32;
  void* partial = nullptr;
  GPK_TRY(gpk_scratch(h, (size_t)ntmT * Mp * sizeof(double), &partial));
  SParams p;
  p.A = (const char*)W2; p.B = (const char*)work2; p.out = (double*)partial;
  p.rsa = Np * 16; p.rsb = Np * 16; p.Mp = Mp;      // bytes between consecutive quads of rows (4 rows x 4 bytes per entry)
  p.ntm = ntm; p.ntn = ntn;
  p.nst = (int)(((long long)ntmT * ntn + gsz - 1) / gsz);
  p.alpha = (float)(1.0 / (w_scale * k_scale));      // undo both operand scalings (powers of two: exact)
  const long long nblocks = (long long)((p.nst + 7) / 8) * 8 * gsz;
  gpk_time_begin(h, GPK_TIMED_K5);
  if (ab == 4) hipLaunchKernelGGL((k5_split_kernel<4, 2, 4>), dim3((unsigned)nblocks), dim3(512), 0, h->stream, p);
  else if (wr == 4) hipLaunchKernelGGL((k5_split_kernel<4, 2>), dim3((unsigned)nblocks), dim3(512), 0, h->stream, p);
  else hipLaunchKernelGGL((k5_split_kernel<2, 2>), dim3((unsigned)nblocks), dim3(256), 0, h->stream, p);
  gpk_time_end(h);
  GPK_LAUNCH_CHECK(h);
  return gpk_colsum_finalize(h, (const double*)partial, ntmT, Mp, M, kss, floor_, var);
}

extern "C" int gpk_predict_var_inv_split(gpk_handle h, const float* X, int64_t N, int D, const double* ls, double sf2,
                                         const void* W3, int64_t Np, const float* Xq, int64_t M, double kss,
                                         double floor_, float* work, void* work3, double* var) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, X && W3 && Xq && work && work3 && var, "predict_var_inv_split: null pointer");
  GPK_REQUIRE(h, N >= 1 && M >= 1 && Np == gpk_padded(N), "predict_var_inv_split: Np must equal gpk_padded(N)");
  GPK_REQUIRE(h, h->batch == 1, "predict_var_inv_split: not available in batched mode");
  const int64_t Mp = gpk_padded(M);
  const int ntm = (int)(Np / 128), ntn = (int)(Mp / 128);
  GPK_REQUIRE(h, (long long)ntm * ntn < (1ll << 30) && Np * 24 * 64 < (1ll << 31), "predict_var_inv_split: size too large");
  // Kq (Mp x Np, query-major, k contiguous) = k(Xq, X) in fp32, then its exact three-way bf16 split
  GPK_TRY(gpk_cross_gram_t(h, GPK_F32, Xq, M, X, N, D, ls, sf2, work, Np));
  GPK_TRY(gpk_split3(h, work, Mp, Np, Np, work3));
#ifndef GPK_K5S_WR
#define GPK_K5S_WR 4
#endif
  // 256 x 128 tiles (8 waves, one workgroup per CU) when Np is a multiple of 256, else 128 x 128 (4 waves, two per CU)
  const int wr = (GPK_K5S_WR == 4 && Np % 256 == 0) ? 4 : 2;
  const int ntmT = (int)(Np / (64 * wr)), gsz = wr == 2 ? 64 : 32;
  void* partial = nullptr;
  GPK_TRY(gpk_scratch(h, (size_t)ntmT * Mp * sizeof(double), &partial));
  SParams p;
  p.A = (const char*)W3; p.B = (const char*)work3; p.out = (double*)partial;
  p.rsa = Np * 24; p.rsb = Np * 24; p.Mp = Mp;      // bytes between consecutive quads of rows
  p.ntm = ntm; p.ntn = ntn;
  p.nst = (int)(((long long)ntmT * ntn + gsz - 1) / gsz);
  p.alpha = 1.0f;
  const long long nblocks = (long long)((p.nst + 7) / 8) * 8 * gsz;
  gpk_time_begin(h, GPK_TIMED_K5);
  if (wr == 4 && h->k5_split_form == 2) hipLaunchKernelGGL(k5_split16_kernel, dim3((unsigned)nblocks), dim3(512), 0, h->stream, p);
  else if (wr == 4) hipLaunchKernelGGL(k5_split_kernel<4>, dim3((unsigned)nblocks), dim3(512), 0, h->stream, p);
  else hipLaunchKernelGGL(k5_split_kernel<2>, dim3((unsigned)nblocks), dim3(256), 0, h->stream, p);
  gpk_time_end(h);
  GPK_LAUNCH_CHECK(h);
  return gpk_colsum_finalize(h, (const double*)partial, ntmT, Mp, M, kss, floor_, var);
}
