// One-launch tile Cholesky: the whole factorisation of an Np x Np matrix (Np <= 16 384 by default), and with it the
// inverses of its 128 x 128 diagonal tiles, as ONE persistent kernel.  Replaces scipy.linalg.cholesky at
// sklearn/gaussian_process/_gpr.py:349,587 for the sizes the reference trains at (src/px4/simple_gp.py:170-177: N <= 10 000;
// src/px4/gp_trainer.py:169-179).
//
// The recursive gpk_potrf (gpk_chol.hip) is a chain of ~3 N / 128 dependent launches; at N = 4096 those launches are
// small and latency-bound and the factorisation runs at 8 % of the fp64 matrix pipe.  Here the matrix is a grid of
// 128 x 128 tiles and every tile is one TASK, handed out in column-major order by a device-side ticket counter:
//
//   T(i, j), i > j :  X = (A_ij - sum_{k<j} L_ik L_jk^T) W_jj^T        (W_jj = L_jj^-1, from T(j, j))
//   T(j, j)        :  A_jj - sum_{k<j} L_jk L_jk^T  ->  L_jj, W_jj      (the leaf, in registers)
//
// A task accumulates its k-sum in registers (left-looking: the tile is read once and written once), waits on a monotone
// per-tile-row counter ready[i] = "tiles (i, 0 .. ready[i] - 1) are final" before each 128-wide k-step and publishes its
// tile by raising ready[i].  Tasks are taken in list order by workgroups that are RUNNING, and a task waits only for
// tasks earlier in the list, so the earliest unfinished task can always proceed: no deadlock whatever the residency.
//
// Register layouts are chosen so that the two dependent steps of the critical path need no data movement:
//   * an off-diagonal task keeps its tile TRANSPOSED, wave w = rows 16 w .. 16 w + 15 of the tile as eight 16 x 16
//     accumulator blocks XT[kb] (element [m][n] = tile[16 w + n][16 kb + m]).  An accumulator block in the f64 C/D map
//     (row = (lane >> 4) + 4 reg, column = lane & 15) IS the B operand of v_mfma_f64_16x16x4_f64 for k-step `reg`,
//     so X^T = W_jj X^^T runs straight from the accumulators (A = blocks of W_jj from LDS), in place.
//   * the diagonal task keeps the lower triangle the same way (wave = one block row r: blocks (r, 0 .. r) transposed,
//     later the blocks of column r of the inverse) and factors it right-looking: the 16 x 16 diagonal block and its
//     inverse by ONE wave on that accumulator itself (gpk_p4.h: four panel steps of 4 x 4 closed forms on wave-uniform
//     scalars and rank-4 MFMA updates), the blocks below it by W_bb * (accumulator), the trailing update and
//     the forward substitution for the inverse by L_(i, b) (one block column of L in LDS) * (accumulator).  While one
//     wave works on a diagonal block the other seven apply the previous block column.
//
// Visibility (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility"): the XCDs' L2s are
// not coherent with each other, so every byte that is handed from one workgroup to another inside the launch (final
// off-diagonal tiles, W_jj) is written with sc1 (write-through) stores and read with sc1 loads; each storing wave waits
// for its stores (s_waitcnt vmcnt(0)), a workgroup barrier follows, and ONE lane raises the counter with an sc1 store;
// consumers poll it with sc1 loads.  A tile is written exactly once and never read before it is final, so no XCD can
// hold a stale copy.  (Keeping a launch on ONE XCD and handing over through its L2 with plain stores was measured: the same
// time - profiles/r05_ptile_one_xcd_ab.log.  What does matter is the SHAPE of a store: whole 128-byte lines per instruction,
// not 32-byte pieces of sixteen lines - the diagonal task and the followers stage their blocks through LDS for that.)
// Every spin loop gives up after GPK_PTILE_TIMEOUT_TICKS of the 100 MHz real-time counter and raises an abort word that
// every other loop polls: the grid always drains.
//
// The hand-overs along the diagonal run 16 columns at a time (p.prog): the critical path is
//   D(j) -> T(j+1, j) -> last k-step of D(j+1) -> D(j+1),
// and each arrow through whole tiles costs a write-through acknowledgement, a flag and 64-128 KB of freshly written lines.
// Instead the diagonal task publishes "block rows < v of L_jj and their 16 x 16 inverses W_bb are final" as it goes
// (wprog; its waves count their own stores - vector memory operations complete in order - so the acknowledgement wait is
// taken one or two steps late and does not stall); the tasks of up to EIGHT tiles under it (p.prog_rows; a third instantiation
// of the task body, PROG) solve  X L_jj^T = X^  by forward substitution behind it, block row by block row, and publish every
// finished 16-column block of their own tile (one progress word per tile row and distance from the diagonal); and whoever
// multiplies those tiles next takes them k-tile by k-tile
// (poll_ktiles).  When a diagonal task is done, the tile under it has one fetch and one block product left, and when that
// tile is done the next diagonal task has one k-tile left.  Deadlock freedom is unchanged: every wait is still for a task
// earlier in the list.
//
// Two later additions (end of round 4): the tiles of the inverse factor's transpose W^T = L^-T as a fourth kind of task in the
// same list (PTParams::wt: tile (i, j), i < j, = -(sum_{k=i}^{j-1} W^T(i,k) L(j,k)^T) W_jj^T - the off-diagonal task on another
// pair of panels; they run in the part of the chip the diagonal chain leaves idle), and the residency: ONE workgroup per CU
// up to ptile_single_max_nt tile columns (every link of the chain is shorter on a CU nobody shares), two above.
// Round 5: eight followers per column instead of two; the diagonal task's 16-column step re-scheduled from its sub-step stamps
// (deferred apply, one-step-late publication from step 2, whole-line stores); a second, 256-register build of the kernel (SR)
// with two k-tiles in flight for launches of up to ptile_sr_max_nt tile columns.  tools/exp_ptile_chain.py shows what bounds
// 6000 - 12 000 rows: every task first streams the backlog of columns that were final when it was taken, and the chain waits
// for its own tasks (four re-orderings of the task list measured, none kept).
#include <cstdlib>

#include "gpk_internal.h"
#include "gpk_p4.h"

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double dv2 __attribute__((ext_vector_type(2)));
typedef unsigned int V16 __attribute__((ext_vector_type(4)));

constexpr int TS = 128;                       // tile edge
constexpr int BK = 16;                        // doubles per k-tile (128 bytes)
constexpr int NT = 512;                       // threads per workgroup (8 waves)
constexpr int ROWB = 144;                     // bytes per staged row (128 + 16 pad: conflict-free 16-byte fragment reads)
constexpr int OPB = TS * ROWB;                // one operand k-tile in LDS
constexpr int BS = 17;                        // row stride (doubles) of a 16 x 16 block in LDS (conflict-free 8-byte reads)
constexpr int BLK = 16 * BS;                  // doubles per block
constexpr int OS = 66;                        // row stride (doubles) of the 128 x 64 output staging image
constexpr int LDS_BYTES = 80 * 1024;          // two workgroups per CU
constexpr int CTL_OFF = LDS_BYTES - 64;       // a few control words at the end
constexpr long long GPK_PTILE_TIMEOUT_TICKS = 400000000ll;   // 4 s of s_memrealtime
constexpr int PAUSE_OFF = 16 + 8 * 512;       // ctrl ints: one word per compute unit (key < 1024)
constexpr int PROG_OFF = PAUSE_OFF + 1024;    // ctrl ints: per diagonal tile, "block rows 0 .. v - 1 of L_jj and their W_bb are final"
constexpr int XPROG_OFF = PROG_OFF + 8 * 512; // ctrl ints: per tile row i, "16-column blocks 0 .. v - 1 of tile (i, i - 1) are final"
constexpr int PT_MAX_FOLLOWERS = 8;           // ... and likewise for the tiles (i, i - 2) .. (i, i - 8): one region of 8 x 512 words per distance
constexpr int WT_OFF = XPROG_OFF + PT_MAX_FOLLOWERS * 8 * 512;   // ctrl ints: per tile row j of W^T, "tiles (j, j .. j + v - 1) are final" (p.wt)
// Build-time choices of the diagonal task (the A/B builds of profiles/r05_ptile_line_stores_ab.log, r05_ptile_defer_ab.log,
// r05_ptile_substamps.log set them on the compiler's command line):
#ifndef PT_LINE_STORES
#define PT_LINE_STORES 1                      // blocks of L_jj and W_jj^T (and the followers' blocks) leave as whole 128-byte lines; 0: 8-byte stores
#endif                                        // straight from the accumulators (LML + gradient at N = 1000 0.465 against 0.449 ms)
#ifndef PT_DEFER
#define PT_DEFER 1                            // the wave on the factoring wave's SIMD applies the previous block column behind the
#endif                                        // step's first barrier: the 16 x 16 factor 4612 -> 3552 cycles
#ifndef PT_LATE1
#define PT_LATE1 2                            // block rows are published two steps late up to this step, one step late from it on
#endif                                        // (0 / 1 / 2 / 4 / 5 / never measured: N = 4096 1.157 / 1.155 / 1.149 / 1.171 / 1.180 / 1.269 ms)
constexpr int XS = 18;                        // row stride (doubles) of a wave's 16 x 16 output staging block
static_assert((22 * BLK + 8 * 16 * XS) * 8 <= CTL_OFF, "forward substitution: two L images, the waves' W blocks and staging blocks");

static_assert(4 * OPB <= CTL_OFF, "staging buffers");
static_assert(36 * BLK * 8 <= CTL_OFF, "W_jj image");
static_assert(TS * OS * 8 <= CTL_OFF, "output staging image");
static_assert((8 + 16) * BLK * 8 <= CTL_OFF, "leaf work area");
static_assert((24 * BLK + 8 * 16 * XS) * 8 <= CTL_OFF, "leaf work area + the waves' staging blocks for the rows of W_jj^T");

struct PTParams {
  double* A; long long lda; long long strideA;     // strides between the problems of a batch, in bytes
  double* winv; long long strideW;
  int* info; int row0;
  int nt, batch, ntasks;
  double* wt; long long strideWt;                  // non-null: the inverse factor's transpose W^T = L^-T by tiles as well (leading
                                                   // dimension lda): nt tasks per tile column instead of nt - column
  int prog_rows;                                   // ... how many tiles under a diagonal tile do so (1 .. PT_MAX_FOLLOWERS)
  int prog;                                        // latency-bound launch: the sub-diagonal tiles follow their diagonal tiles step by step
  int* ctrl;                                       // [0] ticket counter, [1] abort; ready counters from ctrl + 16;
                                                   // "a critical diagonal task runs on this CU" words from ctrl + PAUSE_OFF;
                                                   // progress of the diagonal tiles' factorisations from ctrl + PROG_OFF
  long long* trace;                                // option ptile_trace_path: 16 time stamps per task (100 MHz), or null
  // XCD-aware dealing (null: ONE global ticket, tasks computed from the ticket): the task list cut into PT_QUEUES queues, each in
  // the global (column-major) order; entry = i | j << 9 | problem << 18.  list[0 .. 8] = the queues' offsets, queue q =
  // list[16 + off[q] .. 16 + off[q + 1]); its head is the ticket counter ctrl[QHEAD_OFF + q]; the workgroups of XCD q take from
  // queue q (take_task below).
  const int* list;
};

template <int V> struct IC { static constexpr int value = V; };
template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) { f(IC<I>{}); sfor<I + 1, N>(f); }
}

__device__ __forceinline__ int ld_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// The thread index again, opaque to the optimiser: every section of the kernel derives its per-lane offsets from a
// fresh copy, so that they live for that section only.  (Derived from ONE tid they are loop invariants of the task
// loop; the register allocator then spills them around the register-hungry sections and reloads them inside the
// k-loop, where every reload's s_waitcnt also waits for the operand prefetch just issued.)
__device__ __forceinline__ int fresh_tid() {
  int t = (int)threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}

// one lane: spin until min(*ra, *rb) >= need (rb may be null); -1 when the launch is aborted
__device__ int poll_ready(const int* ra, const int* rb, int need, int* abortp) {
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  for (int it = 0;; ++it) {
    int v = ld_agent(ra);
    if (rb) v = min(v, ld_agent(rb));
    if (v >= need) return v;
    if ((it & 31) == 31) {
      if (ld_agent(abortp) != 0) return -1;
      if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > GPK_PTILE_TIMEOUT_TICKS) { st_agent(abortp, 1); st_agent(abortp + (GPK_PTILE_CTRL_INTS - 1), 1); return -1; }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// one lane: spin until `need` k-tiles (16 columns each) of tile row i are final, as far as a task of tile column j is
// concerned: whole tile columns from *ready, the 16-column blocks of tile (i, i - 1) = (j, j - 1) from *xprog.  Returns the
// count seen (<= 8 j), or -1 when the launch is aborted.
__device__ int poll_ktiles(const int* ready, const int* xprog, int j, int need, int* abortp) {
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  for (int it = 0;; ++it) {
    const int r = ld_agent(ready), x = ld_agent(xprog);      // (both loads in flight together: one latency per look, not two)
    int v = 8 * min(r, j);
    if (r == j - 1) v += min(x, 8);
    if (v >= need) return v;
    if ((it & 31) == 31) {
      if (ld_agent(abortp) != 0) return -1;
      if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > GPK_PTILE_TIMEOUT_TICKS) { st_agent(abortp, 1); st_agent(abortp + (GPK_PTILE_CTRL_INTS - 1), 1); return -1; }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// the same for a task that multiplies two row panels: the k-tiles final in both (xa / xb may be null: whole columns only)
__device__ int poll_ktiles2(const int* ra, const int* xa, const int* rb, const int* xb, int j, int need, int* abortp) {
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  for (int it = 0;; ++it) {
    const int a = ld_agent(ra), bq = ld_agent(rb), xav = xa ? ld_agent(xa) : 0, xbv = xb ? ld_agent(xb) : 0;
    int va = 8 * min(a, j), vb = 8 * min(bq, j);
    if (a == j - 1 && xa) va += min(xav, 8);
    if (bq == j - 1 && xb) vb += min(xbv, 8);
    const int v = min(va, vb);
    if (v >= need) return v;
    if ((it & 31) == 31) {
      if (ld_agent(abortp) != 0) return -1;
      if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > GPK_PTILE_TIMEOUT_TICKS) { st_agent(abortp, 1); st_agent(abortp + (GPK_PTILE_CTRL_INTS - 1), 1); return -1; }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// one lane, task (j, i) of W^T: spin until `need` tile columns from j on are final in row i of L (ready counts from column 0)
// and in row j of W^T (wtready counts from the diagonal tile); -1 when the launch is aborted
__device__ int poll_inv(const int* ready_i, int j, const int* wtready_j, int need, int* abortp) {
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  for (int it = 0;; ++it) {
    const int v = min(ld_agent(ready_i) - j, ld_agent(wtready_j));
    if (v >= need) return v;
    if ((it & 31) == 31) {
      if (ld_agent(abortp) != 0) return -1;
      if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > GPK_PTILE_TIMEOUT_TICKS) { st_agent(abortp, 1); st_agent(abortp + (GPK_PTILE_CTRL_INTS - 1), 1); return -1; }
    }
    __builtin_amdgcn_s_sleep(2);
  }
}

// one lane: a critical task lowers the pause word of its CU - only if it still holds THIS task's value: two critical tasks can
// land on one CU (two workgroups per CU, several problems of a batch), and the later one's raise must outlive the earlier one's end
__device__ __forceinline__ void pause_release(int* p, int tag) {
  int expect = tag;
  __hip_atomic_compare_exchange_strong(p, &expect, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one lane: spin until *p == 0; -1 when the launch is aborted
__device__ int poll_clear(const int* p, int* abortp) {
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  for (int it = 0;; ++it) {
    if (ld_agent(p) == 0) return 0;
    if ((it & 31) == 31) {
      if (ld_agent(abortp) != 0) return -1;
      if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > GPK_PTILE_TIMEOUT_TICKS) { st_agent(abortp, 1); st_agent(abortp + (GPK_PTILE_CTRL_INTS - 1), 1); return -1; }
    }
    __builtin_amdgcn_s_sleep(8);
  }
}

// global -> registers: k-tile of one operand panel (128 rows x 16 doubles), two 16-byte chunks per thread, sc1
__device__ __forceinline__ void load_ktile(const char* base, const unsigned (&voff)[2], V16 (&r)[2]) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 0x7fffffff, 0x00020000);
  r[0] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[0], 0, 16);
  r[1] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[1], 0, 16);
}
__device__ __forceinline__ void store_ktile(char* lds, int tid, const V16 (&r)[2]) {
  const int c = tid & 7, rr = tid >> 3;
  *reinterpret_cast<V16*>(lds + rr * ROWB + c * 16) = r[0];
  *reinterpret_cast<V16*>(lds + (rr + 64) * ROWB + c * 16) = r[1];
}

// 4 MFMAs: acc += A(16 x 16 block in LDS, row-major, stride BS) * B(accumulator block `b` in the C/D map)
// A element [m][kk] is read at ablk[m * BS + kk] (TRANS == false) or ablk[kk * BS + m] (TRANS == true)
template <bool TRANS>
__device__ __forceinline__ d4 blk_mfma(const double* ablk, const d4& b, d4 acc, int lr, int lq) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int kk = lq + 4 * t;
    const double a = TRANS ? ablk[kk * BS + lr] : ablk[lr * BS + kk];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[t], acc, 0, 0, 0);
  }
  return acc;
}

constexpr int PT_QUEUES = 8;                  // one queue per XCD
constexpr int QHEAD_OFF = 2;                  // ctrl ints 2 .. 9: the queues' ticket counters

// One lane: the next task of this workgroup as a position in the task list, or -1 when every queue is exhausted.
// The queues are restrictions of ONE topological order (column-major), and a task waits only for tasks earlier in that
// order.  Workgroup b serves queue b mod 8 (`home`) while that has tasks - workgroups b and b + 8 share an XCD (observed
// placement: round-robin; speed only), so the tasks of one tile ROW run on one XCD, whose L2 then holds the row panels they
// share - and the other queues only once its own is exhausted (load balance at the end).
// Liveness: let t be the earliest unfinished task, in queue q.  The tasks in front of it in q are finished, so t is at q's head
// or already running (then everything it waits for is finished).  A workgroup whose home is q runs tasks of q only while q has
// any - all of them earlier than t, hence finished - so it is free and takes t.  That needs ONE resident workgroup per queue:
// the first eight of the grid (dispatched in order; the grid never exceeds what the device holds).  Should a placement ever break
// that, the spin loops' time-out ends the launch with an error instead of a hang.
__device__ __noinline__ int take_task(const int* list, int* ctrl, int home) {
  int* head = ctrl + QHEAD_OFF;
  for (int d = 0; d < PT_QUEUES; ++d) {
    const int q = (home + d) & (PT_QUEUES - 1);
    const int o = list[q], len = list[q + 1] - o;
    if (d > 0 && ld_agent(head + q) >= len) continue;
    const int t = __hip_atomic_fetch_add(head + q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t < len) return o + t;
  }
  return -1;
}

#ifdef GPK_PTILE_SUBSTAMPS   // (variant build for tools/exp_ptile_trace.py: shader-clock stamps inside step GPK_PTILE_SUBSTAMPS of D(2))
#define PT_SUB(k) do { if (p.trace && j == 2 && JB == GPK_PTILE_SUBSTAMPS && (tl & 63) == 0) p.trace[(long long)p.ntasks * 16 + (k)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define PT_SUB(k) do { } while (0)
#endif
#define PT_STAMP(k) do { if (p.trace && tid == 0) p.trace[(long long)task * 16 + (k)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)

// SR: a second build for the small launches (up to ptile_sr_max_nt tile columns, one resident workgroup per CU): two waves per
// SIMD, 256 registers per lane.  The same tasks, the same arithmetic in the same order (bit-identical factors); what the
// registers buy is a second k-tile in flight in the off-diagonal k-loops (below).  Measured against the 128-register build:
// N = 512 .. 2048 2 - 3 % faster, 4096 1 %, 5120 equal, 6144 - 8192 1 - 3 % slower (profiles/r05_ptile_sr_ab.log).
template <bool SR>
__global__ __launch_bounds__(NT, SR ? 2 : 4) void ptile_potrf_kernel(PTParams p) {
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
  int* ctl = reinterpret_cast<int*>(lds + CTL_OFF);
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int* abortp = p.ctrl + 1;
  // Which compute unit this workgroup runs on (XCC, shader engine / array, CU): the diagonal task - the critical path -
  // raises a word for its CU while its last k-step and its factorisation run, and the OTHER workgroup resident on that CU
  // (two per CU) stands still meanwhile instead of taking half of the CU's matrix pipe and memory queue.  Best effort: the key
  // keeps 7 bits of the hardware id below the XCC (two CUs may share a word - a needless pause, never a missed dependency),
  // and nothing makes the dispatcher place exactly one or two workgroups on every CU.
  const int cu_key = (int)((__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7u) * 128u +
                           ((__builtin_amdgcn_s_getreg((4 << 0) | (8 << 6) | (6 << 11))) & 127u));
  int* pausep = p.ctrl + PAUSE_OFF + cu_key;
  const int nt = p.nt;
  const long long ntp = (long long)nt * (nt + 1) / 2;

  for (;;) {
    __syncthreads();                                  // the previous task is done with LDS and with ctl
    if (tid == 0) {
      int t;
      if (p.list) {
        t = take_task(p.list, p.ctrl, (int)(blockIdx.x & (PT_QUEUES - 1)));
        if (t < 0 || ld_agent(abortp) != 0) t = p.ntasks;
      } else {
        t = __hip_atomic_fetch_add(p.ctrl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t < p.ntasks && ld_agent(abortp) != 0) t = p.ntasks;
      }
      ctl[0] = t;
    }
    __syncthreads();
    const int task = ctl[0];
    if (task >= p.ntasks) return;
    PT_STAMP(0);
    if (p.trace && tid == 0) p.trace[(long long)task * 16 + 14] = (long long)__builtin_readcyclecounter();
    // task -> (problem b, tile row i, tile column j); column-major list: column j starts at j nt - j (j - 1) / 2.
    // With the inverse factor in the launch (p.wt) a column has nt tasks: the diagonal tile, the nt - 1 - c tiles under it, and
    // the c tiles (j, c), j < c, of W^T = L^-T that the diagonal tile c closes - task (i, j) with i < j below:
    //   W^T(i, j) = -(sum_{k=i}^{j-1} W^T(i, k) L(j, k)^T) W_jj^T
    // is the task of a tile under the diagonal on another pair of panels (own rows: row panel i of W^T; the other operand:
    // row panel j of L, both k-contiguous), its k-loop can run as soon as row j of L is final, and its closing product is
    // the same multiplication by W_jj^T.  The factorisation keeps a fifth of the matrix pipe busy at the sizes the reference
    // trains at: these tasks run in the rest.
    int b, i, j;
    if (p.list) {
      const int e = p.list[16 + task];
      i = e & 511; j = (e >> 9) & 511; b = e >> 18;
      if (p.trace && tid == 0) p.trace[(long long)task * 16 + 15] = e;      // (a listed launch: which tile this was)
    } else {
      b = task % p.batch;
      const long long tt = task / p.batch;
      if (p.wt) {
        const int c = (int)(tt / nt), r = (int)(tt - (long long)c * nt);
        j = c;
        i = r < nt - c ? c + r : r - (nt - c);
      } else {
        j = (int)(((double)(2 * nt + 1) - __builtin_sqrt((double)(2 * nt + 1) * (2 * nt + 1) - 8.0 * (double)tt)) * 0.5);
        j = max(0, min(j, nt - 1));
        while (j > 0 && (long long)j * nt - (long long)j * (j - 1) / 2 > tt) --j;
        while ((long long)(j + 1) * nt - (long long)(j + 1) * j / 2 <= tt) ++j;
        i = j + (int)(tt - ((long long)j * nt - (long long)j * (j - 1) / 2));
      }
    }
    (void)ntp;
    double* A = reinterpret_cast<double*>(reinterpret_cast<char*>(p.A) + (long long)b * p.strideA);
    double* Wv = reinterpret_cast<double*>(reinterpret_cast<char*>(p.winv) + (long long)b * p.strideW);
    int* ready = p.ctrl + 16 + b * nt;
    int* wprog = p.ctrl + PROG_OFF + b * nt + j;     // diagonal tile j of problem b: 16-column steps published so far
    int* xprog = p.ctrl + XPROG_OFF + b * nt + i;    // tile (i, i - 1): 16-column blocks published so far
    // the word a follower (i, j), d = i - j rows under the diagonal, publishes its own blocks to - region d behind XPROG_OFF -, and the word that tells it about the blocks of the last tile of its own row panel, (i, j - 1): the tile of the follower
    // one row further from the diagonal in the previous column, if there is one
    const int dfol = i - j;
    int* myprog = p.ctrl + XPROG_OFF + (min(max(dfol, 1), PT_MAX_FOLLOWERS) - 1) * (8 * 512) + b * nt + i;
    const int* rowprog = (dfol >= 1 && dfol + 1 <= p.prog_rows) ? p.ctrl + XPROG_OFF + dfol * (8 * 512) + b * nt + i : nullptr;
    const long long lda = p.lda;
    double* Wt = p.wt ? reinterpret_cast<double*>(reinterpret_cast<char*>(p.wt) + (long long)b * p.strideWt) : nullptr;
    int* wtready = p.ctrl + WT_OFF + b * nt;
    const bool inv = i < j;                              // a tile of W^T (the output, and the own rows, are in Wt)
    double* Atile = (inv ? Wt : A) + (long long)i * TS * lda + (long long)j * TS;
    const bool diag = (i == j);
    const int pause_tag = ((b + 1) << 16) | (j + 1);         // what a critical task of this column and problem raises on its CU
    // One body per task kind, instantiated twice: the two kinds then have their own accumulators.  (As one body with
    // run-time branches the accumulators of both k-loops meet in phi nodes and the register allocator, at its 128-register
    // limit, renames and spills them inside the loops.)
    auto run = [&](auto dc) -> bool {
      constexpr bool DIAG = decltype(dc)::value == 1;
      constexpr bool PROG = decltype(dc)::value == 2;    // the tile under a diagonal tile in a launch bound by the diagonal chain
      constexpr bool INV = decltype(dc)::value == 3;     // a tile of W^T (i < j)
      // the diagonal task IS the critical path: its waves go first wherever they share a SIMD with another workgroup's
      __builtin_amdgcn_s_setprio(DIAG ? 3 : 0);
      // block row owned by this wave: the diagonal task pairs a long and a short row on every SIMD (waves w, w + 4)
      const int rw = DIAG ? (wave < 4 ? wave : 11 - wave) : wave;

      // ---- accumulators start from the tile itself (transposed blocks)
      d4 S[8];
  #pragma unroll
      for (int s = 0; s < 8; ++s) S[s] = d4{0.0, 0.0, 0.0, 0.0};
      {
        const int t0 = fresh_tid(), lr = t0 & 15, lq = (t0 >> 4) & 3;
        const double* src = Atile + (long long)(16 * rw + lr) * lda + lq;
        sfor<0, 8>([&](auto kc) {          // (a diagonal task reads its blocks right of the diagonal too: valid memory,
          constexpr int KB = decltype(kc)::value;   //  they ride along through the k-loop and are dropped after it)
          if constexpr (!INV) {                     // (a tile of W^T starts from zero)
  #pragma unroll
            for (int t = 0; t < 4; ++t) S[KB][t] = src[16 * KB + 4 * t];
          }
        });
      }

      // ---- k-loop over the finished tile columns 0 .. j - 1 (8 k-tiles each), register-staged pipeline as in gpk_gemm.hip
      // (a tile of W^T: tile columns i .. j - 1 of row panel j of L and of row panel i of W^T)
      const int ncol = INV ? j - i : j;
      const int nkt = 8 * ncol;
      if (nkt > 0) {
        const char* pj = reinterpret_cast<const char*>(A + (long long)j * TS * lda + (INV ? (long long)i * TS : 0));   // A operand
        const char* pi = reinterpret_cast<const char*>((INV ? Wt : A) + (long long)i * TS * lda + (INV ? (long long)i * TS : 0));   // B operand: the own rows
        int avail = 0;                                   // tile columns known to be final in rows i and j
        auto need_cols = [&](int need) -> bool {        // uniform; false = aborted
          if (avail >= need) return true;
          if (tid == 0) ctl[1] = INV ? poll_inv(ready + j, i, wtready + i, need, abortp)
                                     : poll_ready(ready + i, DIAG ? nullptr : ready + j, need, abortp);
          __syncthreads();
          const int v = ctl[1];
          __syncthreads();
          if (v < 0) return false;
          avail = min(v, ncol);
          if (avail >= ncol) PT_STAMP(11);

          return true;
        };
        if constexpr (DIAG) {
          // ---- the diagonal task's k-loop: both operands are rows of panel j, and its last k-step - the tile the
          // sub-diagonal task has just published, 2-3 us away in memory - is on the critical path of the whole
          // factorisation.  One operand image per k-tile, brought in by LDS-DMA (buffer_load ... lds: no staging registers)
          // into a ring of FOUR 16 KiB stages: three k-tiles are in flight while one is multiplied, so a k-step of 8 k-tiles
          // costs about one memory latency instead of eight.  Unpadded 128-byte rows, the 16-byte chunks of a row XOR-swizzled
          // by s(r) = (r & 7) ^ (r >> 3), r = row & 15: conflict-free ds_read_b128 fragments.  A wave instruction fills 1 KiB
          // = 8 rows; the compiler knows nothing of these writes: the counted s_waitcnt + s_barrier below are the ordering.
          const int td = fresh_tid(), lane = td & 63, lr = td & 15, lq = (td >> 4) & 3;
          const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(pj), 0, 0x7fffffff, 0x00020000);
          unsigned dvo[2];
  #pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int P = (2 * wave + u) * 64 + lane, row = P >> 3, r = row & 15;
            const int c = (P & 7) ^ ((r & 7) ^ (r >> 3));
            dvo[u] = (unsigned)(((long long)row * lda + 2 * c) * 8);
          }
          auto issue = [&](int kt) {
            char* st = lds + (kt & 3) * 16384 + 2 * wave * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)st, 16, dvo[0], kt * (BK * 8), 0, 16);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(st + 1024), 16, dvo[1], kt * (BK * 8), 0, 16);
          };
          const int frag = lr * 128, sw = (lr & 7) ^ (lr >> 3);
          const int off0 = frag + (((2 * lq) ^ sw) << 4), off1 = frag + (((2 * lq + 1) ^ sw) << 4);
          // The accumulators' initial values (plain loads issued at the start of the task) are claimed HERE: left to the
          // compiler, the wait for them lands in front of the first MFMA inside the loop - as s_waitcnt vmcnt(0), executed
          // on every trip, which would also wait for the k-tiles in flight.
          asm volatile("" : "+v"(S[0]), "+v"(S[1]), "+v"(S[2]), "+v"(S[3]), "+v"(S[4]), "+v"(S[5]), "+v"(S[6]), "+v"(S[7]));
          // `issued` k-tiles have been put in flight so far.  The loop never WAITS for a tile column while it still has
          // landed k-tiles to multiply: when the next column is not final yet it works off what it has (so that nothing
          // stale is left for the moment the fresh column arrives) and blocks only with its ring empty - then three k-tiles
          // go out at once, the fourth behind the next barrier.
          // The last tile column - tile (j, j - 1), whose task solves it 16 columns at a time behind the previous diagonal
          // task - is consumed k-tile by k-tile as it is published (xprog): when that task is done, two k-tiles are left.
          int issued = 0, availk = 0;              // availk: k-tiles of this row panel known to be final
          auto need_kt = [&](int need) -> bool {   // uniform; false = aborted
            if (availk >= need) return true;
            if (tid == 0) ctl[1] = poll_ktiles(ready + i, xprog, j, need, abortp);
            __syncthreads();
            const int v = ctl[1];
            __syncthreads();
            if (v < 0) return false;
            availk = min(v, nkt);
            if (availk >= nkt) PT_STAMP(11);
            return true;
          };
#pragma unroll 1
          for (int kt = 0; kt < nkt; ++kt) {
            if (issued == kt) {                    // ring empty: k-tile kt must be final before anything moves
              // waiting for the LAST column: from here to the end of the task this workgroup is the critical path
              if ((kt >> 3) == j - 1 && availk < nkt && tid == 0) st_agent(pausep, pause_tag);
              if (!need_kt(kt + 1)) return false;
              // (stages kt .. kt + 2 are free: their last readers passed the barrier of iteration kt - 1)
              while (issued < nkt && issued < kt + 3 && issued < availk) issue(issued++);
            }
            // k-tile kt has landed once at most the DMAs of the k-tiles issued after it are outstanding (two per k-tile)
            const int later = issued - 1 - kt;
            if (later >= 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (later == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (later == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // every wave's share is in; stage (kt + 3) & 3 = (kt - 1) & 3 has been read
            if (kt == nkt - 8) PT_STAMP(12);
            if (kt == nkt - 4) PT_STAMP(13);
            while (issued < nkt && issued < kt + 4 && issued < availk) issue(issued++);
            const char* la = lds + (kt & 3) * 16384;
            // only the blocks on and left of the diagonal are part of the task: block row rw multiplies blocks 0 .. rw
            // (waves w and w + 4 share a SIMD and own rows w and 7 - w: nine blocks per SIMD instead of sixteen).  Every
            // fragment is read (LDS reads are cheap), the MFMAs of the other blocks are branched over (wave-uniform).
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
              const int off = hh == 0 ? off0 : off1;
              dv2 bf = *reinterpret_cast<const dv2*>(la + rw * 2048 + off);
              dv2 af[8];
#pragma unroll
              for (int x = 0; x < 8; ++x) af[x] = *reinterpret_cast<const dv2*>(la + x * 2048 + off);
              bf.x = -bf.x; bf.y = -bf.y;
              sfor<0, 8>([&](auto xc) {
                constexpr int KB = decltype(xc)::value;
                if (KB <= rw) {
                  S[KB] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[KB].x, bf.x, S[KB], 0, 0, 0);
                  S[KB] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[KB].y, bf.y, S[KB], 0, 0, 0);
                }
              });
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();            // the last stage has been read (no DMA is outstanding: vmcnt(0) above)
        } else {
        const int tk = fresh_tid(), lr = tk & 15, lq = (tk >> 4) & 3;
        unsigned voff[2];
        {
          const int c = tk & 7, rr = tk >> 3;
          voff[0] = (unsigned)(((long long)rr * lda + c * 2) * 8);
          voff[1] = (unsigned)(((long long)(rr + 64) * lda + c * 2) * 8);
        }
        // (a task that follows its diagonal tile takes the last tile column of its two panels - the tiles of the two tasks that
        // followed the previous diagonal tile - k-tile by k-tile as they are published, like the diagonal task above)
        int availk = 0;
        auto need_kt = [&](int need) -> bool {          // uniform; false = aborted
          if constexpr (!PROG) {
            return need_cols(((need - 1) >> 3) + 1);
          } else {
            if (availk >= need) return true;
            if (tid == 0)
              ctl[1] = poll_ktiles2(ready + i, rowprog, ready + j, p.ctrl + XPROG_OFF + b * nt + j, j, need, abortp);
            __syncthreads();
            const int v = ctl[1];
            __syncthreads();
            if (v < 0) return false;
            availk = min(v, nkt);
            if (availk >= nkt) PT_STAMP(11);
            return true;
          }
        };
        // the products of one k-tile: rows of panel j (la) against this wave's rows of panel i (lb)
        auto mma = [&](const char* la, const char* lb) {
  #pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            dv2 bf = *reinterpret_cast<const dv2*>(lb + (16 * rw + lr) * ROWB + (4 * lq + 2 * hh) * 8);
            bf.x = -bf.x; bf.y = -bf.y;
            sfor<0, 2>([&](auto gc) {
              constexpr int G = decltype(gc)::value;
              dv2 af[4];
  #pragma unroll
              for (int x = 0; x < 4; ++x)
                af[x] = *reinterpret_cast<const dv2*>(la + (16 * (4 * G + x) + lr) * ROWB + (4 * lq + 2 * hh) * 8);
              sfor<0, 4>([&](auto xc) {
                constexpr int KB = 4 * G + decltype(xc)::value;
                S[KB] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[KB - 4 * G].x, bf.x, S[KB], 0, 0, 0);
                S[KB] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[KB - 4 * G].y, bf.y, S[KB], 0, 0, 0);
              });
            });
          }
        };
        if constexpr (SR && !PROG) {
          // TWO k-tiles in flight, in two register sets that take turns (no copies: a copy would wait for the younger set): a
          // k-tile issued in iteration kt is written to LDS in iteration kt + 2 - two k-tiles of matrix work (3.4 us) to arrive
          // instead of one (1.7).  Set A (ra, rb): the odd k-tiles, set B (qa, qb): the even ones.  Worth 2 - 3 % of a launch
          // inside this build (profiles/r05_ptile_sr_ab.log) - not more: with the arithmetic taken out a launch takes the same
          // time with one k-tile in flight and with two (N = 8192: 2.99 / 3.02 ms of 3.83; profiles/r05_ptile_data_path.log) -
          // what bounds these launches is the order of the tasks, not how fast one of them streams.
          // The loop runs in STRETCHES without a poll inside: the compiler's wait-count pass gives up on exact counts in a loop
          // that contains a spin loop with loads of its own (every LDS write then waits for vmcnt(0) - the younger set too; seen in
          // the assembly), so the availability of everything a stretch issues is settled in front of it, and the use of all four
          // registers there starts the stretch from "nothing in flight".  (The followers keep the loop below: they take their
          // last tile column k-tile by k-tile as it is published and cannot wait for three k-tiles ahead.)
          if (!need_kt(min(3, nkt))) return false;
          V16 ra[2], rb[2], qa[2], qb[2];
          load_ktile(pj, voff, ra);
          load_ktile(pi, voff, rb);
          store_ktile(lds, tk, ra);
          store_ktile(lds + OPB, tk, rb);
          {
            const int k1 = min(1, nkt - 1), k2 = min(2, nkt - 1);
            load_ktile(pj + k1 * (BK * 8), voff, ra);
            load_ktile(pi + k1 * (BK * 8), voff, rb);
            load_ktile(pj + k2 * (BK * 8), voff, qa);
            load_ktile(pi + k2 * (BK * 8), voff, qb);
          }
          __syncthreads();
          auto step = [&](int kt, auto curc, V16 (&xa)[2], V16 (&xb)[2]) {
            constexpr int cur = decltype(curc)::value;
            store_ktile(lds + (cur ^ 1) * 2 * OPB, tk, xa);
            store_ktile(lds + (cur ^ 1) * 2 * OPB + OPB, tk, xb);
            const int kn = min(kt + 3, nkt - 1);
            load_ktile(pj + (long long)kn * (BK * 8), voff, xa);
            load_ktile(pi + (long long)kn * (BK * 8), voff, xb);
            mma(lds + cur * 2 * OPB, lds + cur * 2 * OPB + OPB);
            __syncthreads();
          };
          int kt = 0;
  #pragma unroll 1
          while (kt < nkt) {
            if (!need_kt(min(kt + 4, nkt))) return false;          // k-tile kt + 3, the next one to be issued, is final (or all are)
            const int ke = 8 * avail >= nkt ? nkt : 8 * avail - 3;   // iterations < ke issue final k-tiles only
            asm volatile("" : "+v"(ra[0]), "+v"(ra[1]), "+v"(rb[0]), "+v"(rb[1]), "+v"(qa[0]), "+v"(qa[1]), "+v"(qb[0]), "+v"(qb[1]));
            if (kt & 1) { step(kt, IC<1>{}, qa, qb); ++kt; }
  #pragma unroll 1
            for (; kt + 1 < ke; kt += 2) {
              step(kt, IC<0>{}, ra, rb);
              step(kt + 1, IC<1>{}, qa, qb);
            }
            if (kt < ke) { step(kt, IC<0>{}, ra, rb); ++kt; }
          }
        } else {
        if (!need_kt(min(2, nkt))) return false;
        int pause_seen = 0;
        V16 ra[2], rb[2];
        load_ktile(pj, voff, ra);
        load_ktile(pi, voff, rb);
        store_ktile(lds, tk, ra);
        store_ktile(lds + OPB, tk, rb);
        {
          const int k1 = min(1, nkt - 1);
          load_ktile(pj + k1 * (BK * 8), voff, ra);
          load_ktile(pi + k1 * (BK * 8), voff, rb);
        }
        __syncthreads();
        for (int kt = 0; kt < nkt; ++kt) {
          const int cur = kt & 1;
          store_ktile(lds + (cur ^ 1) * 2 * OPB, tk, ra);
          store_ktile(lds + (cur ^ 1) * 2 * OPB + OPB, tk, rb);
          const int kn = min(kt + 2, nkt - 1);
          if (!need_kt(kn + 1)) return false;
          if (tid == 0) {                                   // the CU's pause word, read one iteration ahead of its use
            ctl[2 + cur] = pause_seen;
            pause_seen = ld_agent(pausep);
          }
          load_ktile(pj + (long long)kn * (BK * 8), voff, ra);
          load_ktile(pi + (long long)kn * (BK * 8), voff, rb);
          mma(lds + cur * 2 * OPB, lds + cur * 2 * OPB + OPB);
          __syncthreads();
          // A critical task (the diagonal task of column c, or the tile below it) runs on this CU and has raised c + 1: stand
          // still - but only a task of column >= c does, which nothing on the critical task's dependency chain waits for.
          if (ctl[2 + cur] != 0 && j >= (ctl[2 + cur] & 0xffff) - 1 && i != j + 1) {
            if (tid == 0) ctl[1] = poll_clear(pausep, abortp);
            __syncthreads();
            if (ctl[1] < 0) return false;
            pause_seen = 0;
            __syncthreads();
          }
        }
        }
        }
      }

      PT_STAMP(1);
      if constexpr (!DIAG) {
        // =================================================== off-diagonal task: X^T = W_jj X^^T, in place, then publish
        const int ta = fresh_tid(), lr = ta & 15, lq = (ta >> 4) & 3;
        double* wl = reinterpret_cast<double*>(lds);
        if constexpr (PROG) {
          // ---- the tile below the diagonal is the critical path: it does not wait for W_jj.  X L_jj^T = X^ by forward
          //      substitution, one 16-column step behind the diagonal task's factorisation:
          //          X_c^T = W_cc (X^_c^T - sum_{k<c} L(c, k) X_k^T)
          //      needs block row c of L_jj and the diagonal block W_cc only and runs in place in ascending order; when the
          //      diagonal task is done, one fetch of W_77 and one block product are left.
          // (the tile below that one does the same: the next column's critical tile needs it for its last k-step, and with
          // the whole of W_jj it would be final some 15 us later)
          if (i == j + 1 && tid == 0) st_agent(pausep, pause_tag);   // critical from here to its publication
          const double* Ljj = A + (long long)j * TS * lda + (long long)j * TS;
          const double* Wjj = Wv + (long long)j * TS * TS;
          const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Ljj), 0, 0x7fffffff, 0x00020000);
          const __amdgpu_buffer_rsrc_t rw_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Wjj), 0, 0x7fffffff, 0x00020000);
          // Block row c of the substitution: the blocks L(c, k), k < c - final one step BEFORE W_cc (the diagonal task stores
          // L(c, k) in its step k) and published with the same word: wprog >= c - and the diagonal block W_cc (wprog >= c + 1).
          // Wave w brings in L(c, w) (two 16-byte pieces per lane, negated on the way into the image the step after next reads);
          // EVERY wave brings in its own copy of W_cc (2 KiB: no barrier between the fetch and the one product that needs it).
          // With the update part of step c done under the fetch of W_cc, what is left when the diagonal task is done is one
          // fetch and one block product.  An aborted launch is noticed once, after the steps (whatever they read meanwhile is
          // valid memory).
          auto fetchL = [&](int c, V16 (&v)[2]) {
            if (wave < c) {
              const int ln = fresh_tid() & 63, prow = ln >> 2, pcol = 4 * (ln & 3);   // pieces (prow, pcol) and (prow, pcol + 2)
              const unsigned off = (unsigned)(((long long)(16 * c + prow) * lda + 16 * wave + pcol) * 8);
              v[0] = __builtin_amdgcn_raw_buffer_load_b128(rl, off, 0, 16);
              v[1] = __builtin_amdgcn_raw_buffer_load_b128(rl, off + 16, 0, 16);
            }
          };
          auto putL = [&](int c, double* img, const V16 (&v)[2]) {
            if (wave < c) {
              const int ln = fresh_tid() & 63, prow = ln >> 2, pcol = 4 * (ln & 3);
              double* dstp = img + wave * BLK + prow * BS + pcol;
  #pragma unroll
              for (int u = 0; u < 2; ++u) {
                const dv2 d = __builtin_bit_cast(dv2, v[u]);
                dstp[2 * u] = -d.x;
                dstp[2 * u + 1] = -d.y;
              }
            }
          };
          auto fetchW = [&](int c, V16 (&v)[2]) {
            const int ln = fresh_tid() & 63, prow = ln >> 2, pcol = 4 * (ln & 3);
            const unsigned off = (unsigned)(((16 * c + prow) * TS + 16 * c + pcol) * 8);
            v[0] = __builtin_amdgcn_raw_buffer_load_b128(rw_, off, 0, 16);
            v[1] = __builtin_amdgcn_raw_buffer_load_b128(rw_, off + 16, 0, 16);
          };
          double* wmine = wl + (14 + wave) * BLK;              // this wave's copy of the current W_cc
          auto putW = [&](const V16 (&v)[2]) {
            const int ln = fresh_tid() & 63, prow = ln >> 2, pcol = 4 * (ln & 3);
  #pragma unroll
            for (int u = 0; u < 2; ++u) {
              const dv2 d = __builtin_bit_cast(dv2, v[u]);
              wmine[prow * BS + pcol + 2 * u] = d.x;
              wmine[prow * BS + pcol + 2 * u + 1] = d.y;
            }
          };
          const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(Atile, 0, 0x7fffffff, 0x00020000);
          V16 pl[2] = {}, pw[2] = {};
          int okc = 0;                                        // < 0: the launch was aborted while a poll was waiting
          if (tid == 0) ctl[1] = poll_ready(wprog, nullptr, 1, abortp);
          __syncthreads();
          // `have`: the diagonal task's progress as last seen (monotone).  A task that starts late - the usual case for the second
          // tile under the diagonal, whose k-loop ends when the diagonal task is almost done - finds it far ahead and then polls
          // no more: a poll is one more memory latency in front of every step's fetch (18.1 -> 16.9 us for the eight steps).
          int have = ctl[1];
          okc = min(okc, have);
          PT_STAMP(2);
          fetchW(0, pw);
          fetchL(1, pl);
          sfor<0, 8>([&](auto cc) {
            constexpr int CB = decltype(cc)::value;
            const double* img = wl + (CB & 1) * 7 * BLK;      // L(CB, 0 .. CB - 1), negated: in LDS since the previous step's barrier
            const int tq = fresh_tid(), qr = tq & 15, qq = (tq >> 4) & 3;      // (per step: nothing to keep alive across steps)
            if constexpr (CB == 7) PT_STAMP(3);
            d4 acc = S[CB];
            sfor<0, CB>([&](auto kc) {
              constexpr int KB = decltype(kc)::value;
              acc = blk_mfma<false>(img + KB * BLK, S[KB], acc, qr, qq);
            });
            putW(pw);                                         // (same-wave LDS traffic is ordered: no barrier)
            S[CB] = blk_mfma<false>(wmine, acc, d4{0.0, 0.0, 0.0, 0.0}, qr, qq);
            // ---- column block CB of the tile is final: out it goes (this wave's 16 rows x 16 columns through the wave's own
            //      staging block, two 16-byte write-through stores per lane) - the next diagonal task takes it as one k-tile
            {
              double* stg = wl + 22 * BLK + wave * (16 * XS);
  #pragma unroll
              for (int t = 0; t < 4; ++t) stg[qr * XS + qq + 4 * t] = S[CB][t];
#if PT_LINE_STORES
              // (eight whole 128-byte rows per store instruction - not two 16-byte pieces of every row in each of two)
              const int ln = tq & 63, prow = ln >> 3, pcol = 2 * (ln & 7);
  #pragma unroll
              for (int u = 0; u < 2; ++u) {
                const V16 v = *reinterpret_cast<const V16*>(stg + (8 * u + prow) * XS + pcol);
                __builtin_amdgcn_raw_buffer_store_b128(v, rx, (unsigned)(((long long)(16 * wave + 8 * u + prow) * lda + 16 * CB + pcol) * 8), 0, 16);
              }
#else
              const int ln = tq & 63, prow = ln >> 2, pcol = 4 * (ln & 3);
  #pragma unroll
              for (int u = 0; u < 2; ++u) {
                const V16 v = *reinterpret_cast<const V16*>(stg + prow * XS + pcol + 2 * u);
                __builtin_amdgcn_raw_buffer_store_b128(v, rx, (unsigned)(((long long)(16 * wave + prow) * lda + 16 * CB + pcol + 2 * u) * 8), 0, 16);
              }
#endif
            }
            if constexpr (CB < 7) {
              // ---- the next step: its L blocks go into the other image, then wait for W_(CB+1) (and with it L row CB + 2)
              putL(CB + 1, wl + ((CB + 1) & 1) * 7 * BLK, pl);
              const bool look = have < CB + 2;               // (uniform)
              if (look && tid == 0) ctl[2 + (CB & 1)] = poll_ready(wprog, nullptr, CB + 2, abortp);
              // (vector memory operations complete in order: with this step's two stores the only ones outstanding, this
              // wave's stores of column block CB - 1 have been acknowledged)
              asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
              __syncthreads();                                // image CB + 1 is complete, the poll's result is in
              if (CB >= 1 && tid == 0) st_agent(myprog, CB);
              if (look) {
                have = ctl[2 + (CB & 1)];
                okc = min(okc, have);
              }
              // (fetching W two steps ahead when the diagonal task is that far ahead was measured and LOSES - N = 4096 1.30 against
              // 1.25 ms: the conditional fetches make the compiler wait for everything in flight in front of each use)
              fetchW(CB + 1, pw);
              if constexpr (CB < 6) fetchL(CB + 2, pl);
            }
          });
          if (okc < 0) return false;
          PT_STAMP(4);
          wait_vm0();
          __syncthreads();
          if (tid == 0) {
            st_agent(myprog, 8);
            st_agent(ready + i, j + 1);
            if (i == j + 1) pause_release(pausep, pause_tag);
          }
          PT_STAMP(5);
          return true;
        } else {
        if (tid == 0) {
          if (i == j + 1) st_agent(pausep, pause_tag);          // the tile below the diagonal: critical from here to its publication
          ctl[1] = poll_ready(ready + j, nullptr, j + 1, abortp);
        }
        __syncthreads();
        if (ctl[1] < 0) return false;
        PT_STAMP(2);
        {
          // the 36 lower blocks of W_jj (128 x 128 row-major) -> block-major LDS image, block (mb, kb) at mb (mb + 1) / 2 + kb
          const double* wj = Wv + (long long)j * TS * TS;
          const __amdgpu_buffer_rsrc_t rs =
              __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(wj), 0, 0x7fffffff, 0x00020000);
          V16 v[9];                                                 // all nine 16-byte pieces of a thread in flight at once
  #pragma unroll
          for (int u = 0; u < 9; ++u) {
            const int e = ta + NT * u, blk = e >> 7, rowb = (e & 127) >> 3, cp = e & 7;
            int mb = 0;
            while ((mb + 1) * (mb + 2) / 2 <= blk) ++mb;
            const int kb = blk - mb * (mb + 1) / 2;
            v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(((16 * mb + rowb) * TS + 16 * kb + 2 * cp) * 8), 0, 16);
          }
  #pragma unroll
          for (int u = 0; u < 9; ++u) {
            const int e = ta + NT * u, blk = e >> 7, rowb = (e & 127) >> 3, cp = e & 7;
            const dv2 d = __builtin_bit_cast(dv2, v[u]);
            wl[blk * BLK + rowb * BS + 2 * cp] = d.x;
            wl[blk * BLK + rowb * BS + 2 * cp + 1] = d.y;
          }
        }
        __syncthreads();
        PT_STAMP(3);
        // the 36 block products in the order (MB descending - block MB needs the OLD blocks 0 .. MB only -, KB ascending),
        // the four A values of the next product read from LDS while the current one's MFMAs run
        {
          double an[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) an[t] = wl[(7 * 8 / 2) * BLK + lr * BS + lq + 4 * t];        // (MB, KB) = (7, 0)
          d4 acc = {0.0, 0.0, 0.0, 0.0};
          sfor<0, 36>([&](auto pc) {
            constexpr int PI = decltype(pc)::value;
            constexpr int MB = PI < 8 ? 7 : PI < 15 ? 6 : PI < 21 ? 5 : PI < 26 ? 4 : PI < 30 ? 3 : PI < 33 ? 2 : PI < 35 ? 1 : 0;
            constexpr int KB = PI - (36 - (MB + 1) * (MB + 2) / 2);
            double ac[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) ac[t] = an[t];
            if constexpr (PI < 35) {
              constexpr int MN = KB == MB ? MB - 1 : MB, KN = KB == MB ? 0 : KB + 1;
#pragma unroll
              for (int t = 0; t < 4; ++t) an[t] = wl[(MN * (MN + 1) / 2 + KN) * BLK + lr * BS + lq + 4 * t];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[t], S[KB][t], acc, 0, 0, 0);
            if constexpr (KB == MB) {
              S[MB] = acc;
              acc = d4{0.0, 0.0, 0.0, 0.0};
            }
            __builtin_amdgcn_sched_barrier(0);
          });
        }
        PT_STAMP(4);
        }
        __syncthreads();                                            // every wave is done with the W image
        // (a tile of W^T needs no sign change: the k-loop accumulates MINUS the products, as the factor's update does)
        // S[mb][t] (lane n = lr, q = lq) = L_ij[16 w + n][16 mb + q + 4 t]: through LDS in two column halves, then
        // whole 512-byte row pieces with 16-byte sc1 stores
        double* ob = reinterpret_cast<double*>(lds);
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(Atile, 0, 0x7fffffff, 0x00020000);
        sfor<0, 2>([&](auto hc) {
          constexpr int hf = decltype(hc)::value;
          sfor<0, 4>([&](auto mc) {
            constexpr int ML = decltype(mc)::value;
  #pragma unroll
            for (int t = 0; t < 4; ++t) ob[(16 * wave + lr) * OS + 16 * ML + lq + 4 * t] = S[ML + 4 * hf][t];
          });
          __syncthreads();
  #pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int e = ta + NT * u, row = e >> 5, c2 = e & 31;
            const V16 v = *reinterpret_cast<const V16*>(ob + row * OS + 2 * c2);
            __builtin_amdgcn_raw_buffer_store_b128(v, ro, (unsigned)(((long long)row * lda + 64 * hf + 2 * c2) * 8), 0, 16);
          }
          __syncthreads();
        });
        wait_vm0();
        __syncthreads();
        if (tid == 0) {
          if constexpr (INV) st_agent(wtready + i, j - i + 1);
          else st_agent(ready + i, j + 1);
          if (i == j + 1) pause_release(pausep, pause_tag);
        }
        PT_STAMP(5);
        return true;
      } else {

      // ======================================================= diagonal task: L_jj and W_jj = L_jj^-1 from the accumulators
      // Slot S[k] of the wave that owns block row / column rw:  k <= rw: block (rw, k) of the updated tile, transposed,
      // until block column k is final (then L(rw, k)^T, needed for one more step);  k >= rw: block (k, rw) of the forward
      // substitution for the inverse (R = -sum L W, then W(k, rw)).  Slot rw changes hands when the wave's own diagonal
      // block has gone to the factoring wave.
      const int tl = fresh_tid(), lane = tl & 63, lr = tl & 15, lq = (tl >> 4) & 3;
      double* wd = reinterpret_cast<double*>(lds);                  // wd[b][r][c] = W_bb[c][r]
      double* lcol = wd + 8 * BLK;                                  // two block columns of L: lcol[buf][block row]
      double* Wj = Wv + (long long)j * TS * TS;
      const __amdgpu_buffer_rsrc_t rtile = __builtin_amdgcn_make_buffer_rsrc(Atile, 0, 0x7fffffff, 0x00020000);
      const __amdgpu_buffer_rsrc_t rwt =
          __builtin_amdgcn_make_buffer_rsrc(Wt ? Wt + (long long)j * TS * (lda + 1) : Atile, 0, 0x7fffffff, 0x00020000);
      int* info = p.info + b;
      __syncthreads();                                              // the k-loop's last reads of the staging buffers
      sfor<1, 8>([&](auto kc) {                                     // the blocks right of the diagonal are not part of the task
        constexpr int KB = decltype(kc)::value;
        if (KB > rw) S[KB] = d4{0.0, 0.0, 0.0, 0.0};
      });
      __syncthreads();
      sfor<0, 8>([&](auto jc) {
        constexpr int JB = decltype(jc)::value;
        double* lc = lcol + (JB & 1) * 8 * BLK;                     // block column JB of L
        // ---- phase A: one wave factors the diagonal block; the others apply block column C = JB - 1:
        //      S[X] -= L(X, C) * S[C] for X = C + 1 .. rw (factor rows) or C + 1 .. 7 (inverse columns, rw <= C)
        d4 lbb = {0.0, 0.0, 0.0, 0.0};
        // (the factoring wave shares its SIMD - matrix pipe included - with the wave of block row 7 - JB, which has up to seven
        // block products to apply meanwhile: the factoring wave goes first)
        if (rw == JB) __builtin_amdgcn_s_setprio(3);
        else __builtin_amdgcn_s_setprio(2);
        if (rw == JB) {
          d4 ub = S[JB], xb;                                          // the block itself, as this wave holds it
          PT_SUB(0);
          const int bad = gpk_p4_factor(ub, xb, lane);
          PT_SUB(1);
  #pragma unroll
          for (int t = 0; t < 4; ++t) wd[JB * BLK + lr * BS + lq + 4 * t] = xb[t];   // wd[b][r][c] = W_bb[c][r]
          lbb = ub;                                                 // (stored after the barrier: nobody here waits for it)
          if (bad != 0 && lane == 0) atomicCAS(info, 0, p.row0 + TS * j + 16 * JB + bad);
          PT_SUB(2);
  #pragma unroll
          for (int s = 0; s < 8; ++s) S[s] = d4{0.0, 0.0, 0.0, 0.0};   // nothing of this wave's state is live here
          // (clearing the slots behind the step's first barrier instead - the other waves wait at it - was measured: 1.22 against 1.17 ms)
        }
        // Applying block column C = JB - 1.  The wave that shares its SIMD - and that SIMD's matrix pipe - with the factoring wave
        // (block row 7 - JB: waves w and w + 4 sit on one SIMD) does NOT do it here, next to the factorisation, but behind the
        // step's first barrier: its up to 28 MFMAs otherwise stand in front of the 14 dependent ones of the block routine (the
        // routine took 4 600 cycles inside the kernel against 3 400 alone, profiles/r05_ptile_substamps.log), and in phase B that
        // wave has one product and four stores to do.  (Not in step 3, where that wave owns the NEXT diagonal block.)
        auto apply_prev = [&]() {
          if constexpr (JB > 0) {
            constexpr int C = JB - 1;
            const double* pc = lcol + (C & 1) * 8 * BLK;
            const int hi = rw > C ? rw : 7;
            const d4 nb = -S[C];
            double an[4];
  #pragma unroll
            for (int t = 0; t < 4; ++t) an[t] = pc[(C + 1) * BLK + lr * BS + lq + 4 * t];
            sfor<C + 1, 8>([&](auto xc) {
              constexpr int X = decltype(xc)::value;
              double ac[4];
  #pragma unroll
              for (int t = 0; t < 4; ++t) ac[t] = an[t];
              if constexpr (X < 7) {
  #pragma unroll
                for (int t = 0; t < 4; ++t) an[t] = pc[(X + 1) * BLK + lr * BS + lq + 4 * t];
              }
              if (X <= hi && !(X == C + 1 && rw == C + 1)) {
  #pragma unroll
                for (int t = 0; t < 4; ++t) S[X] = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[t], nb[t], S[X], 0, 0, 0);
              }
              __builtin_amdgcn_sched_barrier(0);
            });
          }
        };
        // The factoring wave (block row JB) shares its SIMD with the wave of block row 7 - JB (waves w and w + 4): that wave's
        // products would take every other issue slot from the serial 16 x 16 factor, so it runs them in phase B, when the
        // factor is done (not in step 3, where the partner is the wave of the NEXT diagonal block and its state is needed first).
        const bool deferred = PT_DEFER && JB > 0 && JB != 3 && rw == 7 - JB && rw != JB;
        if (rw != JB && !deferred) apply_prev();
        // (the last step publishes the previous step's block row already here, a phase early: its stores are a whole sweep old,
        // and the tile under this one then has nothing but W_77 left to fetch when this task is done)
        if constexpr (JB == 7) {
          if (p.prog) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if constexpr (JB == 7) {
          if (p.prog && tid == 0) st_agent(wprog, 7);
        }
        PT_STAMP(2 + JB);
        if (rw == JB) PT_SUB(3);
        if (rw == JB + 1) PT_SUB(8);
        // ---- phase B: S[JB] <- W_bb * S[JB]: below the diagonal block that is L(rw, JB)^T, in the columns of the inverse
        //      W(JB, rw); the wave of the diagonal block itself takes W_bb as it stands
        if (deferred) apply_prev();
        if (rw == JB) {
  #pragma unroll
          for (int t = 0; t < 4; ++t) S[JB][t] = wd[JB * BLK + lr * BS + lq + 4 * t];
        } else {
          S[JB] = blk_mfma<true>(wd + JB * BLK, S[JB], d4{0.0, 0.0, 0.0, 0.0}, lr, lq);
        }
        if (rw == JB + 1) PT_SUB(9);
        if (rw > JB) {
  #pragma unroll
          for (int t = 0; t < 4; ++t) lc[rw * BLK + lr * BS + lq + 4 * t] = S[JB][t];
          // ---- the wave of the next diagonal block goes on at once: L(JB + 1, JB)^T in the C/D map is both the A operand
          //      (lane (m, q), step t: L[m][q + 4 t]) and, negated, the B operand of the block's last update - no LDS round
          //      trip; then the block goes to LDS for the factoring sweep.  Its stores to memory come after that.
          if constexpr (JB < 7) {
            if (rw == JB + 1) {
  #pragma unroll
              for (int t = 0; t < 4; ++t)
                S[JB + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(S[JB][t], -S[JB][t], S[JB + 1], 0, 0, 0);
            }
          }
          // (write-through, as W below: the task of the tile under this one reads L_jj and the W_bb block row by block row
          // while the factorisation is still running)
#if PT_LINE_STORES
          // Whole 128-byte lines: from the accumulators a store instruction writes 32 bytes of each of 16 rows - partial lines,
          // which the memory side turns into read-modify-writes (their acknowledgement took 3.6 us where a task that writes
          // whole rows sees 1.2: profiles/r05_ptile_substamps.log) - so the block is read back from the image this wave has
          // just written (same-wave LDS traffic is ordered), eight rows of 128 bytes per instruction, 16 bytes per lane.
          {
            const int prow = lane >> 3, pcc = lane & 7;
  #pragma unroll
            for (int u = 0; u < 2; ++u) {
              const double* src = lc + rw * BLK + (8 * u + prow) * BS + 2 * pcc;
              const dv2 d = {src[0], src[1]};
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(V16, d), rtile,
                                                     (unsigned)(((long long)(16 * rw + 8 * u + prow) * lda + 16 * JB + 2 * pcc) * 8), 0, 16);
            }
          }
#else
          double* dst = Atile + (long long)(16 * rw + lr) * lda + 16 * JB + lq;
  #pragma unroll
          for (int t = 0; t < 4; ++t) __hip_atomic_store(dst + 4 * t, S[JB][t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        } else {
          double* dst = Wj + (long long)(16 * JB + lq) * TS + 16 * rw + lr;
  #pragma unroll
          for (int t = 0; t < 4; ++t)
            __hip_atomic_store(dst + 4 * t * TS, S[JB][t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (Wt) {                                                 // W_jj^T, the diagonal tile of W^T: entry (16 rw + n, 16 JB + q + 4 t)
#if PT_LINE_STORES
            // (the same, through this wave's own staging block: rows of W_jj^T)
            double* stg = wd + 24 * BLK + wave * (16 * XS);
  #pragma unroll
            for (int t = 0; t < 4; ++t) stg[lr * XS + lq + 4 * t] = S[JB][t];
            const int prow = lane >> 3, pcc = lane & 7;
  #pragma unroll
            for (int u = 0; u < 2; ++u) {
              const V16 v = *reinterpret_cast<const V16*>(stg + (8 * u + prow) * XS + 2 * pcc);
              __builtin_amdgcn_raw_buffer_store_b128(v, rwt, (unsigned)(((long long)(16 * rw + 8 * u + prow) * lda + 16 * JB + 2 * pcc) * 8), 0, 16);
            }
#else
            double* dt = Wt + (long long)j * TS * (lda + 1) + (long long)(16 * rw + lr) * lda + 16 * JB + lq;
  #pragma unroll
            for (int t = 0; t < 4; ++t) __hip_atomic_store(dt + 4 * t, S[JB][t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
          }
          if (rw == JB) {                                           // L_bb: lower triangle only (the tile's upper part stays)
            double* dl = Atile + (long long)(16 * JB + lr) * lda + 16 * JB + lq;
  #pragma unroll
            for (int t = 0; t < 4; ++t)
              if (lq + 4 * t <= lr) __hip_atomic_store(dl + 4 * t, lbb[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        if (rw == JB + 1) PT_SUB(10);
        if (rw == JB) PT_SUB(4);
        // Progress for the tile below, two steps late: every wave has seen its stores of step JB - 2 complete (at most this
        // step's and the previous one's - 4 each, 8 for the wave of the diagonal block - are outstanding: vector memory
        // operations complete in order), then the barrier, then one lane publishes "block rows < JB - 1 are final".  (One
        // step late was measured: the write-through stores take longer than a step to be acknowledged - 6-7 us in the first
        // steps, while the previous column's tiles are being published - and the wait stalled every step by 800 cycles or
        // more.)  The last step publishes a phase early instead (above), and the end of the task everything.
        // (p.prog = 0: a launch without the 16-column hand-overs - the A/B switch ptile_prog_max_nt.)
        // (with W^T in the launch the waves of the inverse columns issue four more stores per step: 8, the factoring wave 12)
        // From step PT_LATE1 on the lag is ONE step: the tile under this one, which cannot start a block row before it is
        // published, is one step closer behind when this task ends - worth more than the stalls it costs now that eight tiles
        // follow the diagonal task (never one step late: 1.27 ms at N = 4096; from step 2 on: 1.15 ms).
        constexpr bool ONE = JB >= PT_LATE1;
        if (p.prog) {
          // stores per step: a wave of the factor's rows NL, a wave of the inverse's columns 4 (+ NT with W^T), the factoring wave
          // 8 (+ NT); a wave waits until only this step's (ONE) or this and the previous step's stores are outstanding
          constexpr int NL = PT_LINE_STORES ? 2 : 4, NT_ = PT_LINE_STORES ? 2 : 4;
          if (Wt) {
            if (ONE) {
              if (rw > JB) wait_vm<NL>();
              else if (rw == JB) wait_vm<8 + NT_>();
              else wait_vm<4 + NT_>();
            } else {
              if (rw > JB) wait_vm<2 * NL>();
              else if (rw == JB) wait_vm<NL + 8 + NT_>();
              else if (rw == JB - 1) wait_vm<8 + NT_ + 4 + NT_>();
              else wait_vm<2 * (4 + NT_)>();
            }
          } else {
            if (ONE) {
              if (rw > JB) wait_vm<NL>();
              else if (rw == JB) wait_vm<8>();
              else wait_vm<4>();
            } else {
              if (rw > JB) wait_vm<2 * NL>();
              else if (rw == JB) wait_vm<NL + 8>();
              else if (rw == JB - 1) wait_vm<12>();
              else wait_vm<8>();
            }
          }
        }
        if (rw == JB + 1) PT_SUB(11);
        __syncthreads();
        if (rw == JB + 1) PT_SUB(12);
        if (p.prog && JB >= 2 && tid == 0) st_agent(wprog, ONE ? JB : JB - 1);
      });
      wait_vm0();
      __syncthreads();
      if (tid == 0) {
        st_agent(wprog, 8);
        if (Wt) st_agent(wtready + j, 1);
        st_agent(ready + j, j + 1);
        pause_release(pausep, pause_tag);
      }
      PT_STAMP(10);
      if (p.trace && tid == 0) p.trace[(long long)task * 16 + 15] = (long long)__builtin_readcyclecounter();
      }
      return true;
    };
    // (one instantiation per task kind: each has its own accumulators, see above)
    if (!(diag ? run(IC<1>{}) : inv ? run(IC<3>{}) : (p.prog && i <= j + p.prog_rows) ? run(IC<2>{}) : run(IC<0>{}))) return;
  }
}

// the diagonal tiles of W^T zeroed: the tasks of W^T read whole tiles, the diagonal task writes only the blocks on and
// right of a diagonal tile's block diagonal
// and, with it in ONE launch, everything else that must be zero before the kernel starts (each was a launch or a memset of its
// own, 5 - 7 us apiece in front of a 0.3 ms factorisation): the control block, the pivot words, winv (the part above its block
// diagonal is never written by the kernel: the tile GEMMs that use winv read whole tiles), and for the caller that goes on to
// the inverse factor the band of zeros right of W's diagonal tiles (GPK_ZERO_BAND_TILES - 1 of them: what the lockstep
// launches read beyond a row's own range).  Workgroup = one matrix row (of problem blockIdx.y).
struct PrepParams {
  int* ctrl; int nctrl;
  int* info; int ninfo;
  double* winv; long long strideW;
  double* wt; long long ldt, strideWt;
  double* wband; long long ldw, strideWb;
  long long Np;
};
__global__ __launch_bounds__(256) void ptile_prepare_kernel(PrepParams p) {
  const long long r = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x;
  if (b == 0) {                                          // the control block, dealt over the launch's workgroups
    for (long long e = r * 256 + tid; e < p.nctrl; e += (long long)gridDim.x * 256) p.ctrl[e] = 0;
    if (r == 0 && tid < p.ninfo) p.info[tid] = 0;
  }
  if (tid < TS) {
    reinterpret_cast<double*>(reinterpret_cast<char*>(p.winv) + b * p.strideW)[r * TS + tid] = 0.0;
    if (p.wt) reinterpret_cast<double*>(reinterpret_cast<char*>(p.wt) + b * p.strideWt)[r * p.ldt + (r / TS) * TS + tid] = 0.0;
  }
  if (p.wband) {
    double* W = reinterpret_cast<double*>(reinterpret_cast<char*>(p.wband) + b * p.strideWb);
    const long long j0 = (r / TS + 1) * TS, j1 = min(p.Np, (r / TS + GPK_ZERO_BAND_TILES) * TS);
    for (long long j = j0 + tid; j < j1; j += 256) W[r * p.ldw + j] = 0.0;
  }
}
}  // namespace

// Factor the Np x Np matrix A (lower triangle, in place) and write the inverses of its diagonal tiles to winv, one launch.
// Returns GPK_OK with *used = 0 when the shape is not served here (the caller then runs the recursion).
// wt (optional): an Np x lda scratch that receives W^T = L^-T by tiles (upper tiles and the diagonal ones; everything else is left
// alone) - only when the whole matrix is this one launch (row0 == 0).
int gpk_potrf_ptile(gpk_handle h, double* A, int64_t Np, int64_t lda, double* winv, int row0, int* used, double* wt,
                    double* wband, int64_t ldw, int zero_info) {
  *used = 0;
  // (three tiles and fewer stay with the launch chain - four launches: nothing to gain, and the factor keeps the bits
  // that the optimiser-path parity test of the 240-row trainer fixture was pinned with)
  if (!h->ptile || Np > h->ptile_max_np || Np < 4 * TS) return GPK_OK;
  if (((uintptr_t)A % 128) != 0 || (lda % 16) != 0 || ((uintptr_t)winv % 128) != 0) return GPK_OK;
  const int nt = (int)(Np / TS);
  const int nb = h->batch;
  const long long ntasks = (wt ? (long long)nt * nt : (long long)nt * (nt + 1) / 2) * nb;
  const size_t ctrl_ints = 16 + (size_t)nb * nt;
  if (ntasks >= (1ll << 30) || ctrl_ints > (size_t)PAUSE_OFF) return GPK_OK;
  if (!h->d_ptile) {
    GPK_CHECK_HIP(h, hipSetDevice(h->device));
    GPK_CHECK_HIP(h, hipMalloc((void**)&h->d_ptile, (GPK_PTILE_CTRL_INTS + 16) * sizeof(int)));
    int cus = 0;
    GPK_CHECK_HIP(h, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device));
    h->ptile_slots = 2 * (cus > 0 ? cus : 256);
  }
  const long long sA = gpk_bstride(h, A), sW = gpk_bstride(h, winv);
  PTParams p;
  p.A = A; p.lda = lda; p.strideA = sA;
  p.winv = winv; p.strideW = sW;
  p.wt = wt; p.strideWt = wt ? gpk_bstride(h, wt) : 0;
  {
    PrepParams q;
    // (the word behind the control block is the sticky "some launch of this gpk_potrf gave up" flag: zeroed by the first launch)
    q.ctrl = h->d_ptile; q.nctrl = (int)GPK_PTILE_CTRL_INTS + (h->ptile_launches == 0 ? 1 : 0);
    q.info = h->d_info; q.ninfo = zero_info ? nb : 0;
    q.winv = winv; q.strideW = sW;
    q.wt = wt; q.ldt = lda; q.strideWt = p.strideWt;
    q.wband = wband; q.ldw = ldw; q.strideWb = wband ? gpk_bstride(h, wband) : 0;
    q.Np = Np;
    hipLaunchKernelGGL(ptile_prepare_kernel, dim3((unsigned)Np, (unsigned)nb), dim3(256), 0, h->stream, q);
    GPK_LAUNCH_CHECK(h);
  }
  p.info = h->d_info; p.row0 = row0;
  p.nt = nt; p.batch = nb; p.ntasks = (int)ntasks;
  p.prog = nt <= h->ptile_prog_max_nt ? 1 : 0;
  p.prog_rows = h->ptile_prog_rows;
  p.ctrl = h->d_ptile;
  p.trace = nullptr;
  p.list = nullptr;
  if (h->ptile_xcd && nt >= h->ptile_xcd_min_nt) {
    // XCD-aware dealing: the column-major task list cut into one queue per XCD by tile ROW (row i of problem b -> queue
    // (i + b) mod 8).  All tasks of a tile row then run on one XCD: tasks (i, j), (i, j + 1), ... share row panel i, the tasks
    // of one tile column that meet there share row panel j, and both come from that XCD's L2 for all but the first reader.
    // Built once per shape and kept on the handle.
    const long long key = ((long long)nt << 32) | ((long long)h->ptile_grp_rows << 24) | ((long long)h->ptile_grp_cols << 16) |
                          ((long long)h->ptile_xcd << 8) | ((long long)nb << 4) | (wt ? 2 : 0) | 1;
    if (key != h->ptile_list_key) {
      std::vector<int> q[PT_QUEUES];
      if (h->ptile_xcd == 2 && !wt) {
        // 2-D groups: super-columns of C tile columns; first the tasks of the diagonal super-tile (rows inside the super-column),
        // then groups of R tile rows x C tile columns, each column-major - a topological order: T(i, j) waits for T(i, k), T(j, k),
        // k < j, and D(j), all in earlier super-columns, in the diagonal super-tile or earlier in its own group.  A whole group goes
        // to ONE queue: its R x C tasks read R + C row panels.
        const int R = h->ptile_grp_rows > 0 ? h->ptile_grp_rows : 8, Cc = h->ptile_grp_cols > 0 ? h->ptile_grp_cols : 4;
        int g = 0;
        for (int c0 = 0; c0 < nt; c0 += Cc) {
          const int c1 = c0 + Cc < nt ? c0 + Cc : nt;
          for (int b = 0; b < nb; ++b) {
            for (int c = c0; c < c1; ++c)
              for (int i = c; i < c1; ++i) q[g % PT_QUEUES].push_back(i | (c << 9) | (b << 18));
            ++g;
          }
          for (int r0 = c1; r0 < nt; r0 += R)
            for (int b = 0; b < nb; ++b) {
              for (int c = c0; c < c1; ++c)
                for (int i = r0; i < r0 + R && i < nt; ++i) q[g % PT_QUEUES].push_back(i | (c << 9) | (b << 18));
              ++g;
            }
        }
      } else
      for (int c = 0; c < nt; ++c) {
        const int rows = wt ? nt : nt - c;
        for (int r = 0; r < rows; ++r) {
          const int i = r < nt - c ? c + r : r - (nt - c);
          for (int b = 0; b < nb; ++b) q[(i + b) % PT_QUEUES].push_back(i | (c << 9) | (b << 18));
        }
      }
      h->ptile_list_host.assign(16, 0);                 // [0 .. 8]: the queues' offsets
      for (int k = 0; k < PT_QUEUES; ++k) {
        h->ptile_list_host[k] = (int)h->ptile_list_host.size() - 16;
        h->ptile_list_host.insert(h->ptile_list_host.end(), q[k].begin(), q[k].end());
      }
      h->ptile_list_host[PT_QUEUES] = (int)h->ptile_list_host.size() - 16;
      if (h->ptile_list_host.size() > h->ptile_list_cap) {
        GPK_CHECK_HIP(h, hipStreamSynchronize(h->stream));
        if (h->d_ptile_list) GPK_CHECK_HIP(h, hipFree(h->d_ptile_list));
        h->d_ptile_list = nullptr; h->ptile_list_cap = 0;
        const size_t want = (h->ptile_list_host.size() + 4095) & ~(size_t)4095;
        GPK_CHECK_HIP(h, hipMalloc((void**)&h->d_ptile_list, want * sizeof(int)));
        h->ptile_list_cap = want;
      }
      // (stream-ordered behind the launches that still read the old list; the host vector lives on the handle)
      GPK_CHECK_HIP(h, hipMemcpyAsync(h->d_ptile_list, h->ptile_list_host.data(), h->ptile_list_host.size() * sizeof(int),
                                      hipMemcpyHostToDevice, h->stream));
      h->ptile_list_key = key;
    }
    p.list = h->d_ptile_list;
  }
  if (!h->ptile_trace_request.empty()) {                   // debugging aid (option "ptile_trace_path"): per-task time stamps to that file
    const std::string tp = h->ptile_trace_request;
    h->ptile_trace_request.clear();
    void* ws = nullptr;
    GPK_TRY(gpk_scratch(h, ((size_t)ntasks * 16 + 64) * sizeof(long long), &ws));
    GPK_CHECK_HIP(h, hipMemsetAsync(ws, 0, ((size_t)ntasks * 16 + 64) * sizeof(long long), h->stream));
    p.trace = (long long*)ws;
    h->ptile_trace_path = tp;
    h->ptile_trace_n = ntasks;
  }
  // Resident workgroups: ONE per CU up to ptile_single_max_nt tile columns, two above.  A workgroup that has its CU to itself
  // runs its task on all four matrix pipes with nobody else in its LDS and memory queues: every link of the diagonal chain is
  // shorter, and while the chain is the bound that beats hiding latency with a second workgroup (measured, 256 against 512
  // workgroups: N = 4096 1.21 / 1.37 ms, 5120 1.57 / 1.85, 8192 3.92 / 4.45, 10 112 6.65 / 7.10; 16 384 25.8 / 25.5: from
  // there on the launch is throughput and two per CU win; 320 or 384 - some CUs with two - lose to both).
  int slots = nt <= h->ptile_single_max_nt ? h->ptile_slots / 2 : h->ptile_slots;
  if (h->ptile_slots_override > 0) slots = h->ptile_slots_override;      // (experiments: option "ptile_slots")
  const unsigned grid = (unsigned)(ntasks < slots ? ntasks : slots);
  // (the small launches: the 256-register build)
  if (h->ptile_sr && nt <= h->ptile_sr_max_nt && slots <= h->ptile_slots / 2) hipLaunchKernelGGL(ptile_potrf_kernel<true>, dim3(grid), dim3(NT), 0, h->stream, p);
  else hipLaunchKernelGGL(ptile_potrf_kernel<false>, dim3(grid), dim3(NT), 0, h->stream, p);
  GPK_LAUNCH_CHECK(h);
  ++h->ptile_launches;
  *used = 1;
  // (the stamps live in the handle's scratch, which the next call of a chain reuses: written out here and now)
  if (p.trace) { GPK_CHECK_HIP(h, hipStreamSynchronize(h->stream)); GPK_TRY(gpk_potrf_ptile_check(h, 0)); }
  return GPK_OK;
}

// after the stream has been synchronised: did the launch give up?
int gpk_potrf_ptile_check(gpk_handle h, int gave_up) {
  if (!h->ptile_trace_path.empty() && h->scratch) {
    std::vector<long long> t((size_t)h->ptile_trace_n * 16 + 64);
    GPK_CHECK_HIP(h, hipMemcpy(t.data(), h->scratch, t.size() * sizeof(long long), hipMemcpyDeviceToHost));
    if (FILE* f = fopen(h->ptile_trace_path.c_str(), "w")) {
      for (long long k = 0; k < h->ptile_trace_n; ++k) {
        for (int c = 0; c < 16; ++c) fprintf(f, "%lld ", t[(size_t)k * 16 + c]);
        fprintf(f, "\n");
      }
      for (int r = 0; r < 4; ++r) {           // per-iteration cycle stamps of one diagonal task's last k-step
        for (int c = 0; c < 16; ++c) fprintf(f, "%lld ", t[(size_t)h->ptile_trace_n * 16 + 16 * r + c]);
        fprintf(f, "\n");
      }
      fclose(f);
    }
    h->ptile_trace_path.clear();
  }
  int ab = gave_up;            // (>= 0: the caller read the flag back with its own results, gpk_status_enqueue)
  if (ab < 0) GPK_CHECK_HIP(h, hipMemcpy(&ab, h->d_ptile + GPK_PTILE_CTRL_INTS, sizeof(int), hipMemcpyDeviceToHost));
  if (ab != 0) {
    h->err = "potrf: the one-launch factorisation timed out waiting for a tile (internal error)";
    return GPK_HIP_ERROR;
  }
  return GPK_OK;
}
