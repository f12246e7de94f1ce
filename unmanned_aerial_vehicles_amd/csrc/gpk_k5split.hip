// K5 on the 16-bit matrix pipe at fp32 accuracy: V = W Kq^T with both fp32 operands split into 16-bit parts, the tile
// reduced to per-column sums of squares (fp64) in the epilogue instead of being stored.  Two forms:
//
//  * fp16 x 2 (k5_direct_kernel; the fp32 serving default): x s = h0 + h1, two fp16 parts rounded to nearest (s: one power
//    of two per 128-row block), block products a1 b0 + a0 b1 + a0 b0 on v_mfma_f32_32x32x16_f16, fp32 accumulation.  The
//    operands are stored in "fragment order" and go from L2 straight into registers: no LDS, no barriers.
//  * bf16 x 3 (k5_split_kernel): x = x0 + x1 + x2 EXACTLY, three bf16 parts, six products per block (a0 b0, a0 b1, a1 b0,
//    a1 b1, a0 b2, a2 b0: bf16 x bf16 products are exact, the dropped cross terms are below 2^-24 of |a||b|); twice the
//    matrix-pipe work of the first form.  Register-staged LDS pipeline on 128 x 128 or 256 x 128 tiles.
//
// The fp32 MFMA runs at the vector rate (157 TF); the 16-bit pipe is 16x faster, so three (six) instructions per
// fp32-equivalent product raise the ceiling of this launch by 5.3x (2.7x).
//
// Operand layout of the bf16 x 3 form (gpk_split3): 16-byte chunks [row / 4][k16 block][row % 4][half h][part s] - the
// eight bf16 of part s for k = 16 kb + 8 h .. + 7 - so that one k-tile of 16 of four consecutive rows is 384 contiguous
// bytes = three whole cache lines (consecutive lanes fetch consecutive chunks, a wave instruction touches 8 lines; with
// plain row-major 96-byte pieces it touched 11-22 partial lines, and variants that fetched 16- or 32-byte pieces per
// row were 20-45 % slower) and a lane's MFMA fragment (row r = lane & 31, k half h = lane >> 5) is one ds_read_b128 per
// part.  LDS rows are padded to 112 bytes (28 dwords = 4 x odd: conflict-free b128 reads over 8 consecutive rows).
// Both kernels share the tile mapping of gemm_kernel (gpk_gemm.hip): band-linear super-tiles in serpentine XCD order,
// heavy-first rows and lockstep k-ranges.
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
constexpr int LROW = 112;      // LDS row of a k-tile: the row's 6 chunks of 16 bytes plus 16 bytes of padding (28 dwords = 4 x odd)
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

struct SParams {
  const char* A;        // W, split layout, Np rows
  const char* B;        // Kq, split layout, Mp rows
  double* out;          // [ntm][Mp] partial column sums of squares
  long long rsa, rsb;   // bytes between consecutive quads of rows (Np * 24)
  long long Mp;
  int ntm, ntn, nst;
};

// WR wave rows x 2 wave columns of 64 x 64 (2 x 2 blocks of 32 x 32) per workgroup: WR = 2 -> 128 x 128 tile, 4 waves,
// 2 workgroups per CU; WR = 4 -> 256 x 128 tile, 8 waves, 1 workgroup per CU (25 % less operand staging per MFMA)
template <int WR>
__global__ __launch_bounds__(WR * 128, 2) void k5_split_kernel(SParams p) {
  constexpr int TMR = 64 * WR;                                     // tile rows
  constexpr int SSZ = (TMR + 128) * LROW;                          // one LDS stage: [A k-tile | B k-tile]
  constexpr int BOFF = TMR * LROW;                                 // B inside a stage
  __shared__ __attribute__((aligned(16))) char lds[2 * SSZ];
  constexpr int NT = WR * 128, NCH = (TMR + 128) * 6, NQ = (NCH + NT - 1) / NT;   // threads, 16-byte chunks per k-tile
  constexpr int GSZ = WR == 2 ? 64 : 32;                           // resident workgroups per XCD = tiles per group
  constexpr int BH = 1024 / TMR;                                   // band height in tile rows (1024 matrix rows)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int row_w = wm * 64, col_w = wn * 64;

  // ---- tile mapping: bands of 1024 rows walked column by column in groups of GSZ tiles, serpentine over XCDs
  const int ntm = p.ntm * 128 / TMR;                               // tile rows of this configuration
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
  f16v acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
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
    isb[q] = __builtin_amdgcn_readfirstlane(g >= TMR * 6 ? 1 : 0) != 0;
    const int c = live[q] ? (isb[q] ? g - TMR * 6 : g) : 0, row = c / 6, w = c - row * 6;
    goff[q] = (unsigned)((long long)(row >> 2) * p.rsa + ((row & 3) * 6 + w) * 16);     // rsa: bytes between row quads
    lofs[q] = (isb[q] ? BOFF : 0) + row * LROW + w * 16;
  }
  const char* ua = p.A + (long long)(row0 >> 2) * p.rsa;
  const char* ub = p.B + (long long)(col0 >> 2) * p.rsb;
  // Register ring of two k-tiles in flight: the fetch of k-tile kt+4 is issued when k-tile kt+2 leaves its ring slot
  // for LDS.  The loop is unrolled by 4: slot and LDS buffer indices are constants.
  V16 rr[2][NQ];
  auto fetch = [&](int tile, auto slotc) {
    constexpr int sl = decltype(slotc)::value;
    const long long ko = (long long)min(tile, nkt - 1) * ROWB;
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
    char* buf = lds + bi * SSZ;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      if (NCH % NT == 0 || live[q]) *reinterpret_cast<V16*>(buf + lofs[q]) = rr[sl][q];
  };
  const int fr = lane & 31, fh = lane >> 5;
  auto mfmas = [&](const bf16x8 (&af)[2][3], const bf16x8 (&bf)[2][3]) {
    // smallest terms first
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        f16v c = acc[a][b];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][2], bf[b][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][1], bf[b][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][1], bf[b][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][0], c, 0, 0, 0);
        acc[a][b] = c;
      }
  };
  auto frags = [&](auto bufc, bf16x8 (&af)[2][3], bf16x8 (&bf)[2][3]) {
    constexpr int bi = decltype(bufc)::value;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int s = 0; s < 3; ++s)
        af[a][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const V16*>(lds + bi * SSZ + (row_w + 32 * a + fr) * LROW + (fh * 3 + s) * 16));
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int s = 0; s < 3; ++s)
        bf[b][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const V16*>(lds + bi * SSZ + BOFF + (col_w + 32 * b + fr) * LROW + (fh * 3 + s) * 16));
  };
  // Fragments one k-tile ahead: while the MFMAs of k-tile kt run out of registers F[kt & 1], the fragments of k-tile
  // kt+1 are read from LDS buffer (kt+1) & 1 into F[(kt+1) & 1] and k-tile kt+2 is written into LDS buffer kt & 1
  // (its previous content, k-tile kt, was read during iteration kt-1; the barrier at the end of each iteration
  // separates the two).  No MFMA waits for LDS.
  bf16x8 FA[2][2][3], FB[2][2][3];
  fetch(0, IntC<0>{});
  stage(IntC<0>{}, IntC<0>{});
  fetch(1, IntC<1>{});
  stage(IntC<1>{}, IntC<1>{});
  fetch(2, IntC<0>{});
  fetch(3, IntC<1>{});
  __syncthreads();
  frags(IntC<0>{}, FA[0], FB[0]);
  __syncthreads();
  auto body = [&](int kt, auto ksc) {
    constexpr int cur = decltype(ksc)::value & 1;
    stage(IntC<cur>{}, IntC<cur>{});            // k-tile kt+2 (ring slot (kt+2) % 2 = cur)
    fetch(kt + 4, IntC<cur>{});
    frags(IntC<(cur ^ 1)>{}, FA[cur ^ 1], FB[cur ^ 1]);
    mfmas(FA[cur], FB[cur]);
    // issue order: the LDS stores of k-tile kt+2 first, then MFMAs with the fragment reads and the global loads spread
    // evenly among them
    constexpr int NMF = 24, NRD = 12;
    __builtin_amdgcn_sched_group_barrier(0x200, NQ, 0);
#pragma unroll
    for (int g = 0; g < NQ; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, (NMF + NQ - 1) / NQ, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, (NRD + NQ - 1) / NQ, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    __syncthreads();
  };
  for (int kt = 0; kt + 3 < nkt; kt += 4) {      // nkt is a multiple of 8 (k-ranges are whole 128-tiles): no remainder
    body(kt, IntC<0>{});
    body(kt + 1, IntC<1>{});
    body(kt + 2, IntC<2>{});
    body(kt + 3, IntC<3>{});
  }

  // ---- epilogue: per-column sums of squares of the tile (fp64), out[tile row][column]
  double* red = reinterpret_cast<double*>(lds);      // [WR][128]; the k-loop ended with a barrier
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < 16; ++i) { const float v = acc[a][b][i]; s = __builtin_fmaf(v, v, s); }
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

// ---- the fp16 x 2 launch: operands straight from L2 into MFMA fragments, no LDS ------------------------------------
// "Fragment order" of a split operand: the 16-byte chunk (row, k16 block kb, half h, part s) lives at chunk
// index (((row / 32) * KB + kb) * 2 + s) * 64 + h * 32 + row % 32 (KB = cols / 16): the 64 chunks of one
// (32-row block, k16 block, part) - exactly one v_mfma_f32_32x32x16_f16 operand, lane = h * 32 + row % 32 - are 1 KiB
// of contiguous memory, so a fragment is ONE fully coalesced buffer_load_dwordx4 per wave (eight whole 128-byte lines),
// and a 32-row block is (cols / 16) * 2 KiB of contiguous memory.
// The launch: tiles of (128 AB) x 128, four waves, one per SIMD with the whole 512-register file; wave w owns rows
// 32 AB w .. of the tile and ALL 128 columns (AB x 4 accumulator blocks of 32 x 32: 256 registers at AB = 4).  A row
// block of W is used by one wave only, so nothing is gained by staging it in LDS: each wave loads its own A fragments
// straight into registers (they come from L2 exactly once per workgroup, as in the LDS-staged kernel of round 2) and the four waves' identical B
// (K*) fragment loads meet in the CU's vector L1 (a barrier every few k-tiles to keep them together measured the same as
// none: there is none).  Per k16 step and workgroup: 40 KiB from L2 as in that kernel, but no LDS writes (40 KiB there)
// and no LDS reads (96 KiB there), no barriers, and 0.33 fragment loads per MFMA against 0.5.  The launch is bound by the
// clock the chip holds under this load (that kernel: MFMA busy 0.89 at 1.39 GHz, profiles/r02_pmc_mfma_k5_fp16x2.txt; this
// one: 0.92 at 1.42 GHz with 8 % less time, profiles/r03_pmc_k5_fp16x2_direct.txt): less data movement per product is what
// raises it.
// A ring of three k-tiles of fragments (3 x 64 registers) keeps two k-tiles in flight; a fragment's registers are
// refilled (k-tile kt + 3) right behind the last MFMA that reads them.
struct DParams {
  const char* A;        // W, fragment order, Np rows
  const char* B;        // K*, fragment order, Mp rows
  double* out;          // [Np / (128 AB)][Mp] partial column sums of squares
  const float* wscale;  // the power-of-two scale of each 128-row block of W (gpk_split2_rows)
  long long rba, rbb;   // bytes of one 32-row block (Np * 128)
  long long Mp;
  float kscale;         // the scale of K*
  int ntm, ntn, nst;    // 128-row tile rows, 128-column tile columns, super-tiles
  int per_pad;          // tile slots per band of the walk (>= band height in tiles x ntn)
};

template <int AB>
__global__ __launch_bounds__(256, 1) void k5_direct_kernel(DParams p) {
  constexpr int TMR = 128 * AB, GSZ = 32, BH = 1024 / TMR;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ---- tile mapping: bands of 1024 rows walked column by column in groups of 32 tiles (one group = the resident
  // workgroups of an XCD: 2 x 512 rows of W and 16 x 128 queries shared through its L2), serpentine over the XCDs, heavy
  // rows first.  Every tile runs over its OWN k-range: the lockstep union of k_ranges that the LDS-staged kernels use costs
  // 0.8 % more MFMA work and measured 0.6 % slower here; bands of 2048 / 4096 rows (all XCDs on one W band, sharing it
  // through the Infinity Cache) measured 2 % / 15 % slower - what the XCD's own L2 misses costs more than HBM saves.
  const int ntm = p.ntm * 128 / TMR;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int grp = j / GSZ, slot = j - grp * GSZ;
  const int st = grp * 8 + ((grp & 1) ? 7 - xcd : xcd);
  if (st >= p.nst) return;
  // (p.per_pad: the BH x ntn tiles of a band rounded up to whole groups when that idles few slots - every group is then
  // 2 tile rows x 16 columns of ONE band instead of straddling two bands, i.e. 3072 instead of up to 4096 operand rows)
  const int nfull = ntm / BH, hlast = ntm - nfull * BH;
  const int t = st * GSZ + slot;
  int tm, tn;
  {
    const int band = min(t / p.per_pad, nfull);
    const int idx = t - band * p.per_pad, hh = band < nfull ? BH : hlast;
    if (idx >= hh * p.ntn) return;               // a padding slot, or beyond the last tile
    tn = idx / hh;
    tm = band * BH + (idx - tn * hh);
  }
  tm = ntm - 1 - tm;
  const int row0 = tm * TMR, col0 = tn * 128;
  // k-tiles of 16 of THIS WAVE: W is lower triangular, so its rows row0 + 32 AB wave .. end at column row0 + 32 AB (wave + 1)
  // (the waves of a workgroup do not wait for each other before the epilogue)
  const int nkt = (row0 + 32 * AB * (wave + 1)) / 16;

  f16v acc[AB][4];
#pragma unroll
  for (int a = 0; a < AB; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  const unsigned vo = lane * 16;
  const char* ua = p.A + (long long)((row0 >> 5) + AB * wave) * p.rba;
  const char* ub = p.B + (long long)(col0 >> 5) * p.rbb;
  const int rba = (int)p.rba, rbb = (int)p.rbb;
  V16 FA[3][AB][2], FB[3][4][2];
  auto rsrc_at = [&](const char* base, int tile) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base + (long long)min(tile, nkt - 1) * 2048), 0, 0x7fffffff, 0x00020000);
  };
  auto ldf = [&](const __amdgpu_buffer_rsrc_t rs, int blk_off, V16 (&f)[2]) {
    f[0] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, blk_off, 0);
    f[1] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, blk_off + 1024, 0);
  };
  auto fill = [&](int tile, auto slc) {          // prologue: a whole k-tile, in the order the MFMAs will want it
    constexpr int S = decltype(slc)::value;
    const __amdgpu_buffer_rsrc_t ra = rsrc_at(ua, tile), rb = rsrc_at(ub, tile);
    ldf(ra, 0, FA[S][0]);
#pragma unroll
    for (int b = 0; b < 4; ++b) ldf(rb, b * rbb, FB[S][b]);
#pragma unroll
    for (int a = 1; a < AB; ++a) ldf(ra, a * rba, FA[S][a]);
  };
  auto mm = [&](f16v& c, const V16 (&fa)[2], const V16 (&fb)[2]) {
    const f16x8 a0 = __builtin_bit_cast(f16x8, fa[0]), a1 = __builtin_bit_cast(f16x8, fa[1]);
    const f16x8 b0 = __builtin_bit_cast(f16x8, fb[0]), b1 = __builtin_bit_cast(f16x8, fb[1]);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c, 0, 0, 0);       // smallest terms first
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c, 0, 0, 0);
  };
  fill(0, IntC<0>{});
  fill(1, IntC<1>{});
  fill(2, IntC<2>{});
  auto body = [&](int kt, auto slc) {
    constexpr int S = decltype(slc)::value;
    const __amdgpu_buffer_rsrc_t ra = rsrc_at(ua, kt + 3), rb = rsrc_at(ub, kt + 3);
#pragma unroll
    for (int a = 0; a < AB; ++a) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        mm(acc[a][b], FA[S][a], FB[S][b]);
        if (a == AB - 1) ldf(rb, b * rbb, FB[S][b]);      // last use of this k-tile's B fragment: refill it
        __builtin_amdgcn_sched_barrier(0);
      }
      ldf(ra, a * rba, FA[S][a]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  int kt = 0;
  for (; kt + 3 <= nkt; kt += 3) {
    body(kt, IntC<0>{});
    body(kt + 1, IntC<1>{});
    body(kt + 2, IntC<2>{});
  }
  if (kt < nkt) {
    body(kt, IntC<0>{});
    if (kt + 1 < nkt) body(kt + 1, IntC<1>{});
  }

  // ---- epilogue: per-column sums of squares of the tile (fp64), out[tile row][column]
  __shared__ double red[4 * 128];
  // powers of two, each reciprocal exact; formed separately so that a large block scale times the K* scale cannot
  // overflow on the way (absmax_to_scale_kernel also keeps block scales at or below 2^100)
  const float alpha = (1.0f / p.wscale[(row0 + 32 * AB * wave) >> 7]) * (1.0f / p.kscale);
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < AB; ++a)
#pragma unroll
      for (int i = 0; i < 16; ++i) { const float v = alpha * acc[a][b][i]; s = __builtin_fmaf(v, v, s); }
    s += __shfl_xor(s, 32, 64);
    if (lane < 32) red[wave * 128 + 32 * b + lane] = (double)s;
  }
  __syncthreads();
  if (tid < 128) p.out[(long long)tm * p.Mp + col0 + tid] = (red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid]);
}

// fragment-order fp16 x 2 split of a row-major fp32 matrix: one wave per (32-row block, k16 block) = one MFMA operand pair;
// lane = h * 32 + r reads the eight values of (row r, k half h) and writes its two chunks: each wave instruction of the
// store writes 1 KiB of contiguous output.  scales: one power of two per 128-row block (device).
__global__ __launch_bounds__(256) void split2_kernel(const float* __restrict__ src, long long rows, long long cols,
                                                      long long ld, const float* __restrict__ scales,
                                                      V16* __restrict__ dst) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long KB = cols >> 4;
  const long long kb = (long long)blockIdx.x * 4 + wave, rb = blockIdx.y;
  if (kb >= KB) return;
  const int r = lane & 31, h = lane >> 5;
  const long long row = rb * 32 + r;
  const float sc = scales[row >> 7];
  const float4* s4 = reinterpret_cast<const float4*>(src + row * ld + kb * 16 + h * 8);
  const float4 lo = s4[0], hi = s4[1];
  const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  f16x8 p0, p1;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = v[j] * sc;
    const _Float16 h0 = (_Float16)x;
    p0[j] = h0;
    p1[j] = (_Float16)(x - (float)h0);
  }
  V16* d = dst + ((rb * KB + kb) * 2) * 64 + lane;
  d[0] = __builtin_bit_cast(V16, p0);
  d[64] = __builtin_bit_cast(V16, p1);
}

// per 128-row block of a square fp32 matrix: max |A_ij| over the lower triangle (one row per workgroup, one atomic per
// wave), then the power of two that puts that maximum in [2^14, 2^15]
__global__ __launch_bounds__(256) void tril_block_absmax_kernel(const float* __restrict__ A, long long n, long long lda,
                                                                unsigned* __restrict__ out) {
  const long long i = blockIdx.x;
  float m = 0.f;
  for (long long jj = threadIdx.x; jj <= i; jj += 256) m = fmaxf(m, fabsf(A[i * lda + jj]));
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out + (i >> 7), __float_as_uint(m));
}
// The same two kernels reading the fp64 inverse factor itself (gpk_trtri output): the fp32 value is formed on the fly,
// (float)W_ij rounded to nearest - what gpk_tril_to_f32 would have stored - so scales and parts are bit-identical to the
// route through an fp32 copy, which then need not exist (17 GB at N = 65 536).  Only the lower tiles are read; the 16-column
// blocks right of a row's diagonal tile are not written (the variance launch stops at the diagonal tile).
__global__ __launch_bounds__(256) void tril_block_absmax_f64_kernel(const double* __restrict__ A, long long n, long long lda,
                                                                    unsigned* __restrict__ out) {
  const long long i = blockIdx.x;
  float m = 0.f;
  for (long long jj = threadIdx.x; jj <= i; jj += 256) m = fmaxf(m, fabsf((float)A[i * lda + jj]));
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out + (i >> 7), __float_as_uint(m));
}
__global__ __launch_bounds__(256) void split2_f64_kernel(const double* __restrict__ src, long long n, long long ld,
                                                          const float* __restrict__ scales, V16* __restrict__ dst) {
  typedef double dv2 __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long KB = n >> 4;
  const long long kb = (long long)blockIdx.x * 4 + wave, rb = blockIdx.y;
  if (kb >= KB || kb * 16 >= ((rb * 32) / 128 + 1) * 128) return;          // right of the diagonal tile: never read
  const int r = lane & 31, h = lane >> 5;
  const long long row = rb * 32 + r, col0 = kb * 16 + h * 8;
  const float sc = scales[row >> 7];
  const dv2* s2 = reinterpret_cast<const dv2*>(src + row * ld + col0);
  const dv2 a = s2[0], b = s2[1], c = s2[2], d = s2[3];
  const double v[8] = {a.x, a.y, b.x, b.y, c.x, c.y, d.x, d.y};
  f16x8 p0, p1;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = (col0 + j <= row ? (float)v[j] : 0.f) * sc;
    const _Float16 h0 = (_Float16)x;
    p0[j] = h0;
    p1[j] = (_Float16)(x - (float)h0);
  }
  V16* o = dst + ((rb * KB + kb) * 2) * 64 + lane;
  o[0] = __builtin_bit_cast(V16, p0);
  o[64] = __builtin_bit_cast(V16, p1);
}
__global__ void absmax_to_scale_kernel(unsigned* __restrict__ io, int nblk) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblk) return;
  const unsigned u = io[b];                     // bits of a non-negative float
  const int e = (int)(u >> 23) - 127;           // max = 2^e * m, 1 <= m < 2
  float s = 1.f;
  if (u != 0u && u < 0x7f800000u) {
    const int se = ((u & 0x7fffffu) == 0u ? 15 : 14) - e;          // largest power of two with s * max <= 2^15
    s = __uint_as_float((unsigned)(min(max(se, -126), 100) + 127) << 23);    // (<= 2^100: s times the K* scale stays finite)
  }
  reinterpret_cast<float*>(io)[b] = s;
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

extern "C" int gpk_split2_rows(gpk_handle h, const float* W, int64_t n, int64_t ld, float* scales, void* dst) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, W && scales && dst, "split2_rows: null pointer");
  GPK_REQUIRE(h, n >= 128 && n % 128 == 0 && ld >= n && ld % 4 == 0 && n < (1ll << 24), "split2_rows: n must be a multiple of 128");
  GPK_REQUIRE(h, ((uintptr_t)W % 16) == 0 && ((uintptr_t)dst % 16) == 0, "split2_rows: buffers must be 16-byte aligned");
  const int nblk = (int)(n / 128);
  GPK_CHECK_HIP(h, hipMemsetAsync(scales, 0, nblk * sizeof(float), h->stream));
  hipLaunchKernelGGL(tril_block_absmax_kernel, dim3((unsigned)n), dim3(256), 0, h->stream, W, (long long)n, (long long)ld,
                     reinterpret_cast<unsigned*>(scales));
  GPK_LAUNCH_CHECK(h);
  hipLaunchKernelGGL(absmax_to_scale_kernel, dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, h->stream,
                     reinterpret_cast<unsigned*>(scales), nblk);
  GPK_LAUNCH_CHECK(h);
  hipLaunchKernelGGL(split2_kernel, dim3((unsigned)((n / 16 + 3) / 4), (unsigned)(n / 32)), dim3(256), 0, h->stream, W,
                     (long long)n, (long long)n, (long long)ld, (const float*)scales, (V16*)dst);
  GPK_LAUNCH_CHECK(h);
  return GPK_OK;
}

int gpk_tril_block_absmax_f64_enqueue(gpk_handle h, const double* W, int64_t n, int64_t ld, unsigned* out) {
  GPK_CHECK_HIP(h, hipMemsetAsync(out, 0, (n / 128) * sizeof(unsigned), h->stream));
  hipLaunchKernelGGL(tril_block_absmax_f64_kernel, dim3((unsigned)n), dim3(256), 0, h->stream, W, (long long)n, (long long)ld, out);
  GPK_LAUNCH_CHECK(h);
  return GPK_OK;
}

static int split2_rows_f64_impl(gpk_handle h, const double* W, int64_t n, int64_t ld, float* scales, void* dst, bool have_absmax) {
  GPK_REQUIRE(h, W && scales && dst, "split2_rows_f64: null pointer");
  GPK_REQUIRE(h, n >= 128 && n % 128 == 0 && ld >= n && ld % 2 == 0 && n < (1ll << 24), "split2_rows_f64: n must be a multiple of 128");
  GPK_REQUIRE(h, ((uintptr_t)W % 16) == 0 && ((uintptr_t)dst % 16) == 0, "split2_rows_f64: buffers must be 16-byte aligned");
  const int nblk = (int)(n / 128);
  if (!have_absmax) GPK_TRY(gpk_tril_block_absmax_f64_enqueue(h, W, n, ld, reinterpret_cast<unsigned*>(scales)));
  hipLaunchKernelGGL(absmax_to_scale_kernel, dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, h->stream,
                     reinterpret_cast<unsigned*>(scales), nblk);
  GPK_LAUNCH_CHECK(h);
  hipLaunchKernelGGL(split2_f64_kernel, dim3((unsigned)((n / 16 + 3) / 4), (unsigned)(n / 32)), dim3(256), 0, h->stream, W,
                     (long long)n, (long long)ld, (const float*)scales, (V16*)dst);
  GPK_LAUNCH_CHECK(h);
  return GPK_OK;
}

extern "C" int gpk_split2_rows_f64(gpk_handle h, const double* W, int64_t n, int64_t ld, float* scales, void* dst) {
  if (!h) return GPK_BAD_ARG;
  return split2_rows_f64_impl(h, W, n, ld, scales, dst, false);
}

extern "C" int gpk_split2_rows_f64_absmax(gpk_handle h, const double* W, int64_t n, int64_t ld, float* scales, void* dst) {
  if (!h) return GPK_BAD_ARG;
  return split2_rows_f64_impl(h, W, n, ld, scales, dst, true);
}

namespace {
// K* in split form + the one launch of the fp16 x 2 form: column sums of squares by tile row in *partial ([*S][Mp])
int split2_launch(gpk_handle h, const char* who, const float* X, int64_t N, int D, const double* ls, double sf2, const void* W2,
                  const float* w_scales, int64_t Np, const float* Xq, int64_t M, void* work2, void** partial, int* S) {
  GPK_REQUIRE(h, X && W2 && w_scales && Xq && work2, (std::string(who) + ": null pointer").c_str());
  GPK_REQUIRE(h, N >= 1 && M >= 1 && Np == gpk_padded(N), (std::string(who) + ": Np must equal gpk_padded(N)").c_str());
  GPK_REQUIRE(h, h->batch == 1, (std::string(who) + ": not available in batched mode").c_str());
  GPK_REQUIRE(h, sf2 > 0.0 && D >= 1 && D <= 16, (std::string(who) + ": sf2 must be positive and D <= 16").c_str());
  const int64_t Mp = gpk_padded(M);
  const int ntm = (int)(Np / 128), ntn = (int)(Mp / 128);
  // (a fragment's byte offset inside a wave's AB row blocks is 32-bit: AB * Np * 128 bytes)
  GPK_REQUIRE(h, (long long)ntm * ntn < (1ll << 30) && Np * 128 * 4 < (1ll << 31) && Mp / 64 < 65536,
              (std::string(who) + ": size too large").c_str());
  // K* in (0, sf2]: the power of two that puts sf2 just below 2^15
  const double k_scale = std::ldexp(1.0, 14 - (int)std::floor(std::log2(sf2)));
  GPK_TRY(gpk_cross_split2(h, Xq, M, X, N, D, ls, sf2, k_scale, work2));
  // the tallest tile (512 / 256 / 128 rows x 128 columns) that still comes in at least two rounds of the 256 CUs
  int ab = 1;
  if (Np % 512 == 0 && (h->k5_split2_tile == 2 || (Np / 512) * (long long)ntn >= 512)) ab = 4;
  else if (Np % 256 == 0 && (Np / 256) * (long long)ntn >= 512) ab = 2;
  if (h->k5_split2_tile == 1) ab = 1;
  const int ntmT = (int)(Np / (128 * ab)), gsz = 32;
  GPK_TRY(gpk_scratch(h, (size_t)ntmT * Mp * sizeof(double), partial));
  DParams p;
  p.A = (const char*)W2; p.B = (const char*)work2; p.out = (double*)*partial; p.wscale = w_scales;
  p.rba = p.rbb = Np * 128;      // bytes of one 32-row block
  p.Mp = Mp;
  p.kscale = (float)k_scale;
  p.ntm = ntm; p.ntn = ntn;
  {
    // the tile walk: bands of 1024 rows; a band's tiles are rounded up to whole groups of 32 when that idles at most an
    // eighth of the slots (headline shape: 158 -> 160)
    const int bh = 1024 / (128 * ab), per = bh * ntn, pad = (per + gsz - 1) / gsz * gsz;
    p.per_pad = (pad - per) * 8 <= per ? pad : per;
    const int nfull = ntmT / bh, hlast = ntmT - nfull * bh;
    p.nst = (int)(((long long)nfull * p.per_pad + (long long)hlast * ntn + gsz - 1) / gsz);
  }
  const long long nblocks = (long long)((p.nst + 7) / 8) * 8 * gsz;
  gpk_time_begin(h, GPK_TIMED_K5);
  if (ab == 4) hipLaunchKernelGGL(k5_direct_kernel<4>, dim3((unsigned)nblocks), dim3(256), 0, h->stream, p);
  else if (ab == 2) hipLaunchKernelGGL(k5_direct_kernel<2>, dim3((unsigned)nblocks), dim3(256), 0, h->stream, p);
  else hipLaunchKernelGGL(k5_direct_kernel<1>, dim3((unsigned)nblocks), dim3(256), 0, h->stream, p);
  gpk_time_end(h);
  GPK_LAUNCH_CHECK(h);
  *S = ntmT;
  return GPK_OK;
}
}  // namespace

extern "C" int gpk_predict_var_inv_split2(gpk_handle h, const float* X, int64_t N, int D, const double* ls, double sf2,
                                          const void* W2, const float* w_scales, int64_t Np, const float* Xq, int64_t M,
                                          double kss, double floor_, void* work2, double* var) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, var, "predict_var_inv_split2: null pointer");
  void* partial = nullptr;
  int S = 0;
  GPK_TRY(split2_launch(h, "predict_var_inv_split2", X, N, D, ls, sf2, W2, w_scales, Np, Xq, M, work2, &partial, &S));
  return gpk_colsum_finalize(h, (const double*)partial, S, gpk_padded(M), M, kss, floor_, var);
}

extern "C" int gpk_predict_mean_var_split2(gpk_handle h, const float* X, const float* alpha, int64_t N, int D, int P,
                                           const double* ls, double sf2, const double* center, const double* y_mean,
                                           const double* y_std, const void* W2, const float* w_scales, int64_t Np,
                                           const float* Xq, int64_t M, double kss, double floor_, void* work2,
                                           float* mean_tmp, double recheck_below, unsigned* low_count, double* out) {
  if (!h) return GPK_BAD_ARG;
  GPK_REQUIRE(h, alpha && y_mean && y_std && mean_tmp && out && P >= 1 && P <= GPK_MAX_P, "predict_mean_var_split2: bad argument");
  // K4: the matrix-core kernel when the caller passes the expansion centre (it has checked max |u|^2), else exact differences
  if (center) GPK_TRY(gpk_predict_mean_mfma(h, X, alpha, N, D, P, ls, sf2, center, y_mean, y_std, Xq, M, mean_tmp));
  else GPK_TRY(gpk_predict_mean(h, GPK_F32, X, alpha, N, D, P, ls, sf2, y_mean, y_std, Xq, M, mean_tmp));
  void* partial = nullptr;
  int S = 0;
  GPK_TRY(split2_launch(h, "predict_mean_var_split2", X, N, D, ls, sf2, W2, w_scales, Np, Xq, M, work2, &partial, &S));
  return gpk_colsum_finalize_packed(h, (const double*)partial, S, gpk_padded(M), M, kss, floor_, mean_tmp, P, y_std,
                                    recheck_below, low_count, out);
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
  // 256 x 128 tiles (8 waves, one workgroup per CU) when Np is a multiple of 256, else 128 x 128 (4 waves, two per CU)
  const int wr = (Np % 256 == 0) ? 4 : 2;
  const int ntmT = (int)(Np / (64 * wr)), gsz = wr == 2 ? 64 : 32;
  void* partial = nullptr;
  GPK_TRY(gpk_scratch(h, (size_t)ntmT * Mp * sizeof(double), &partial));
  SParams p;
  p.A = (const char*)W3; p.B = (const char*)work3; p.out = (double*)partial;
  p.rsa = Np * 24; p.rsb = Np * 24; p.Mp = Mp;      // bytes between consecutive quads of rows
  p.ntm = ntm; p.ntn = ntn;
  p.nst = (int)(((long long)ntmT * ntn + gsz - 1) / gsz);
  const long long nblocks = (long long)((p.nst + 7) / 8) * 8 * gsz;
  gpk_time_begin(h, GPK_TIMED_K5);
  if (wr == 4) hipLaunchKernelGGL(k5_split_kernel<4>, dim3((unsigned)nblocks), dim3(512), 0, h->stream, p);
  else hipLaunchKernelGGL(k5_split_kernel<2>, dim3((unsigned)nblocks), dim3(256), 0, h->stream, p);
  gpk_time_end(h);
  GPK_LAUNCH_CHECK(h);
  return gpk_colsum_finalize(h, (const double*)partial, ntmT, Mp, M, kss, floor_, var);
}
