// Cholesky factor AND inverse of one 16 x 16 diagonal block by ONE wave, in registers (gfx950).
//
// Every 16-lane DPP row holds a copy of the block, lane i = row i in 16 registers (u); the identity rows e_i (x) ride
// along in the same lanes.  Right-looking sweep with unscaled columns, a_ik -= (a_ic / d_c) a_kc: what a step needs from
// another row -- the pivot d_c and the column entries a_kc -- is one v_mov_b64_dpp row_newbcast each (lane k of the own
// 16-lane row to all its lanes): no LDS and no scalar round trips inside the loop.  The identity rows undergo the same
// updates (x L^T = e_r by substitution), which yields W^T = L^-T.  One final scaling of the columns by 1 / sqrt(d_c)
// turns both into L and W^T (d_c / sqrt(d_c) = sqrt(d_c) on the diagonal).  Straight-line and branch-free (a failed
// pivot is handled by selects and reported once); the next pivot's reciprocal -- v_rcp_f64 + two Newton steps, error
// ~1e-16, the loop-carried chain -- is issued right after the one update it depends on.  Entries above the diagonal of
// the block are never read.
#pragma once
#include <hip/hip_runtime.h>

// value of `x` in lane SRC of the caller's 16-lane DPP row, in every lane of that row: one v_mov_b64_dpp
// (the DPP control must be a literal for the builtin to keep its 64-bit type inside a template: one case per lane)
template <int SRC>
__device__ __forceinline__ double gpk_row_bcast(double x) {
  const long long b = __builtin_bit_cast(long long, x);
  long long r = 0;
#define GPK_ROW_NEWBCAST(n) if constexpr (SRC == n) r = __builtin_amdgcn_update_dpp(0ll, b, 0x150 + n, 0xf, 0xf, true);
  GPK_ROW_NEWBCAST(0) GPK_ROW_NEWBCAST(1) GPK_ROW_NEWBCAST(2) GPK_ROW_NEWBCAST(3) GPK_ROW_NEWBCAST(4) GPK_ROW_NEWBCAST(5)
  GPK_ROW_NEWBCAST(6) GPK_ROW_NEWBCAST(7) GPK_ROW_NEWBCAST(8) GPK_ROW_NEWBCAST(9) GPK_ROW_NEWBCAST(10) GPK_ROW_NEWBCAST(11)
  GPK_ROW_NEWBCAST(12) GPK_ROW_NEWBCAST(13) GPK_ROW_NEWBCAST(14) GPK_ROW_NEWBCAST(15)
#undef GPK_ROW_NEWBCAST
  static_assert(SRC >= 0 && SRC < 16, "row_newbcast lane");
  return __builtin_bit_cast(double, r);
}
template <int V> struct GpkP2C { static constexpr int value = V; };
template <int I, int N, class F>
__device__ __forceinline__ void gpk_p2_for(F&& f) {
  if constexpr (I < N) {
    f(GpkP2C<I>{});
    gpk_p2_for<I + 1, N>(f);
  }
}

// src: the block, row-major with row stride ss (doubles), in LDS, every entry finite; only its lower triangle takes part
// (entries above the diagonal ride along as garbage that nothing reads).
// ldst[i * ls + c]: the factor in c <= i (c > i: garbage);  wdst[r * ws + c] = W[c][r] (the inverse, transposed), every entry.
// S: the caller's eight accumulator blocks, used as the sweep's 32 working registers (u = S[0..3], x = S[4..7]) -- a wave
// that factors a diagonal block holds nothing else that is live, and saying so in the code is what keeps the register
// allocator from carrying those blocks through the sweep.  S holds garbage on return.
// Returns 0, or the 1-based column of the first non-positive pivot (the sweep stays finite), in every lane.
typedef double gpk_d4 __attribute__((ext_vector_type(4)));
#define GPK_P2_U(c) S[(c) >> 2][(c) & 3]
#define GPK_P2_X(c) S[4 + ((c) >> 2)][(c) & 3]
__device__ __forceinline__ int gpk_p2_factor(const double* src, int ss, double* ldst, int ls, double* wdst, int ws, int lane,
                                             gpk_d4 (&S)[8]) {
  const int i = lane & 15;
  gpk_p2_for<0, 16>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    double uc = src[i * ss + c], xc = (c == i) ? 1.0 : 0.0;
    // (opaque to the optimiser: knowing that x starts as a row of the identity it rewrites the early columns' updates
    // into selects on values it then keeps alive -- and spills -- for the rest of the sweep)
    asm volatile("" : "+v"(xc));
    GPK_P2_U(c) = uc;
    GPK_P2_X(c) = xc;
  });
  int bad = 0;                             // 1-based column of the first non-positive pivot (wave-uniform: scalar unit)
  double d = gpk_row_bcast<0>(GPK_P2_U(0)), rd;
  auto pivot = [&](auto cc) {              // d holds the (final) pivot of column C, the same value in every lane
    constexpr int C = decltype(cc)::value;
    const bool ok = (__builtin_amdgcn_ballot_w64(d > 0.0) & 1ull) != 0;
    bad = (!ok && bad == 0) ? C + 1 : bad;
    d = ok ? d : 1.0;
    rd = __builtin_amdgcn_rcp(d);
    rd = __builtin_fma(__builtin_fma(-d, rd, 1.0), rd, rd);
    rd = __builtin_fma(__builtin_fma(-d, rd, 1.0), rd, rd);
  };
  pivot(GpkP2C<0>{});
  gpk_p2_for<0, 16>([&](auto cc) {
    constexpr int C = decltype(cc)::value;
    const double nf = -GPK_P2_U(C) * rd, ng = -GPK_P2_X(C) * rd;
    if constexpr (C + 1 < 16) {
      const double s1 = gpk_row_bcast<C + 1>(GPK_P2_U(C));
      GPK_P2_U(C + 1) = __builtin_fma(nf, s1, GPK_P2_U(C + 1));
      GPK_P2_X(C + 1) = __builtin_fma(ng, s1, GPK_P2_X(C + 1));
      d = gpk_row_bcast<C + 1>(GPK_P2_U(C + 1));      // the next pivot is final now
      pivot(GpkP2C<C + 1>{});
    }
    // column C across the row (one v_mov_b64_dpp each), four at a time: read ahead of the updates that use them, but
    // not further (the whole sweep is straight-line code: without the fences the scheduler hoists every broadcast of
    // every column and runs out of registers)
    gpk_p2_for<0, 4>([&](auto gg) {
      constexpr int K0 = C + 2 + 4 * decltype(gg)::value;
      if constexpr (K0 < 16) {
        constexpr int K1 = K0 + 4 < 16 ? K0 + 4 : 16;
        double sk[4];
        gpk_p2_for<K0, K1>([&](auto kk) { sk[decltype(kk)::value - K0] = gpk_row_bcast<decltype(kk)::value>(GPK_P2_U(C)); });
        gpk_p2_for<K0, K1>([&](auto kk) {
          constexpr int K = decltype(kk)::value;
          double uk = __builtin_fma(nf, sk[K - K0], GPK_P2_U(K)), xk = __builtin_fma(ng, sk[K - K0], GPK_P2_X(K));
          // (pins the updates here: instruction selection otherwise sinks the x chains to the column that first reads
          // them and keeps every broadcast alive until then)
          asm volatile("" : "+v"(uk), "+v"(xk));
          GPK_P2_U(K) = uk;
          GPK_P2_X(K) = xk;
        });
        __builtin_amdgcn_sched_barrier(0);
      }
    });
  });
  double pv = 1.0;                         // lane i: its own pivot d_i (1 where the pivot failed)
  gpk_p2_for<0, 16>([&](auto cc) { constexpr int c = decltype(cc)::value; pv = (i == c) ? GPK_P2_U(c) : pv; });
  pv = pv > 0.0 ? pv : 1.0;
  const double rs = 1.0 / __builtin_sqrt(pv);
  gpk_p2_for<0, 16>([&](auto cc) {
    constexpr int C = decltype(cc)::value;
    const double sc = gpk_row_bcast<C>(rs);
    GPK_P2_U(C) *= sc;
    GPK_P2_X(C) *= sc;
  });
  // a failed pivot's diagonal entry: 1 (as the pivot the sweep went on with)
  gpk_p2_for<0, 16>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    GPK_P2_U(c) = (i == c && !(GPK_P2_U(c) > 0.0)) ? 1.0 : GPK_P2_U(c);
  });
  if (lane < 16) {
    gpk_p2_for<0, 16>([&](auto cc) { constexpr int c = decltype(cc)::value; ldst[i * ls + c] = GPK_P2_U(c); });
  }
  if (lane >= 16 && lane < 32) {
    gpk_p2_for<0, 16>([&](auto cc) { constexpr int c = decltype(cc)::value; wdst[i * ws + c] = GPK_P2_X(c); });
  }
  return bad;
}
#undef GPK_P2_U
#undef GPK_P2_X
