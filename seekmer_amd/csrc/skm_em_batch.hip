// The EM of /root/reference/seekmer/infer.py:133-168 for EM_BATCH problems at once that share
// their class structure and differ only in the class counts and the abundance vector: the
// bootstrap replicates of `-b N` (infer.py:79-82, 108-111 -- every replicate resamples the
// counts of the SAME class table and starts from the same estimate).
//
// A single problem's step is three launches of 5-13 us that move 8 useful bytes per 64-byte
// sector they gather (x[t] per (class, transcript) pair, inner[c] per pair the other way round),
// and a replicate takes ~30 of them: run one by one, a third of the time is gaps between
// launches.  Here the replicates sit side by side -- x[t][r], inner[c][r], count[c][r], r = 0..7
// -- so that one gathered sector carries all eight, the index arrays are read once for the
// eight, and there is one launch where there were eight.
//
// Every replicate still runs ITS OWN iteration: the additions are the single-problem kernels'
// additions in the same order (skm_em.hip), the stopping rule (infer.py:160) is judged per
// replicate, and a replicate that has stopped is frozen (its x is carried from buffer to buffer
// unchanged) while the others go on, so its result and its step count are exactly what the
// one-by-one run gives -- bit for bit (tests/test_gpu_parity.py::test_bootstrap_draw_and_em).
#include "skm_kernels.h"

#include <algorithm>

namespace skm {

namespace {

constexpr int R = EM_BATCH;
static_assert(R == 8, "the row sums pair lanes and replicates eight by eight");
enum { BCTL_ALL_DONE = 0, BCTL_LAST_STEP = 1, BCTL_DONE = 8, BCTL_ITERS = 16, BCTL_UNDEFINED = 24 };

// The stopping rule for each replicate over the partials of finalize pass `steps_done`; returns
// the mask of replicates that have stopped (now or earlier).  Block 0 of the next em_inner launch
// evaluates it; with `publish` it latches what is new.
__device__ unsigned int evaluate_batch(const EmBatchProblem &p, int n_parts, int64_t steps_done, bool publish)
{
    __shared__ double s_max[32][R];
    __shared__ unsigned int s_flags[32][R];
    __shared__ unsigned int s_mask;
    const int r = threadIdx.x & (R - 1), g = threadIdx.x >> 3;        // 256 lanes: 32 groups x 8 replicates
    double m = 0.0;
    unsigned int f = 0;
    for (int b = g; b < n_parts; b += 32) {
        const double o = p.part_max[b * R + r];
        m = o > m ? o : m;
        f |= p.part_flags[b * R + r];
    }
    s_max[g][r] = m;
    s_flags[g][r] = f;
    if (threadIdx.x == 0) s_mask = 0;
    __syncthreads();
    if (threadIdx.x < R) {
        for (int k = 0; k < 32; ++k) {
            m = s_max[k][r] > m ? s_max[k][r] : m;
            f |= s_flags[k][r];
        }
        const bool latched = p.ctl[BCTL_DONE + r] != 0;
        bool done, undefined = false;
        if (!(f & 1u)) {
            undefined = true;                 // numpy raises on max() of an empty selection
            done = true;
        } else {
            done = (f & 2u) || !(m > p.rel_tol);                 // NaN propagates through max()
        }
        if (publish && !latched && done) {
            if (undefined) p.ctl[BCTL_UNDEFINED + r] = 1;
            p.ctl[BCTL_ITERS + r] = (unsigned long long)steps_done;
            p.ctl[BCTL_DONE + r] = 1;
        }
        if (latched || done) atomicOr(&s_mask, 1u << r);
    }
    __syncthreads();
    const unsigned int mask = s_mask;
    // (who says "all done": the judge when the host fills the places -- nothing new can come before it
    // looks --, the assign kernel when the device does: a place that stops is refilled in the same step)
    if (publish && !p.managed && threadIdx.x == 0 && mask == (1u << R) - 1u && !p.ctl[BCTL_ALL_DONE]) {
        p.ctl[BCTL_LAST_STEP] = (unsigned long long)steps_done;
        p.ctl[BCTL_ALL_DONE] = 1;
    }
    return mask;
}

// mgr words: see "the working set kept full by the device" below
enum { MGR_NEXT = 0, MGR_COUNT = 1, MGR_FINISHED = 2, MGR_UNDEFINED = 3, MGR_REP = 8, MGR_SINCE = 16, MGR_TAKE = 24,
       MGR_PUT = 32 };

// What a latched stop means, place by place (one lane): the replicate's step count is recorded, its
// result is to be taken, the next replicate of the group (if any) is to be put in its place; the plan
// goes to mgr[MGR_TAKE ..], mgr[MGR_PUT ..] for em_batch_refill.  `deferred`: called from inside a step
// (block 0 of em_inner_batch, right after it has judged the step before), where the control block
// must stay as the step's other kernels read it -- a stopped place has to be carried through the
// step's finalize pass -- so the control words of a refilled place and the "all done" latch are
// left to em_batch_refill, which runs behind the step.
__device__ void plan_places(const EmBatchProblem &p, unsigned long long *mgr, int64_t *iters_out, int64_t step, bool deferred)
{
    bool any = false;
    for (int r = 0; r < R; ++r) {
        unsigned long long take = 0, put = 0;
        if (mgr[MGR_REP + r] != 0 && p.ctl[BCTL_DONE + r]) {
            const unsigned long long rep = mgr[MGR_REP + r] - 1;
            if (iters_out) iters_out[rep] = (int64_t)(p.ctl[BCTL_ITERS + r] - mgr[MGR_SINCE + r]);
            if (p.ctl[BCTL_UNDEFINED + r]) mgr[MGR_UNDEFINED] = 1;
            take = rep + 1;
            mgr[MGR_FINISHED] += 1;
            if (mgr[MGR_NEXT] < mgr[MGR_COUNT]) {
                put = ++mgr[MGR_NEXT];                              // (replicate mgr[MGR_NEXT] - 1, + 1)
                mgr[MGR_REP + r] = put;
                mgr[MGR_SINCE + r] = (unsigned long long)(step + 1);
                if (!deferred) {
                    p.ctl[BCTL_DONE + r] = 0;
                    p.ctl[BCTL_UNDEFINED + r] = 0;
                    p.ctl[BCTL_ITERS + r] = 0;
                }
            } else {
                mgr[MGR_REP + r] = 0;                               // idle from now on (it stays "stopped")
            }
        }
        mgr[MGR_TAKE + r] = take;
        mgr[MGR_PUT + r] = put;
        any |= mgr[MGR_REP + r] != 0;
    }
    if (!any && !deferred) {
        p.ctl[BCTL_LAST_STEP] = (unsigned long long)(step + 1);
        p.ctl[BCTL_ALL_DONE] = 1;
    }
}

__global__ void __launch_bounds__(256)
em_inner_batch_kernel(EmBatchProblem p, int parity, int eval_parts, int64_t steps_done)
{
    __shared__ int s_all;
    if (threadIdx.x == 0) s_all = p.ctl[BCTL_ALL_DONE] != 0;       // one reading per block
    __syncthreads();
    if (s_all) return;
    // block 0 judges the step before this one and latches what has stopped; the other blocks do not
    // wait (see em_inner_kernel: a pass of `inner` nobody reads, once per batch)
    if (eval_parts > 0 && blockIdx.x == 0
            && evaluate_batch(p, eval_parts, steps_done, true) == (1u << R) - 1u && !p.managed) return;
    // the device-managed working set: what has just been latched is planned for at once (the launch
    // of its own that used to do this after every step is gone); em_batch_refill carries it out
    if (p.managed && p.mgr && blockIdx.x == 0 && threadIdx.x == 0) plan_places(p, p.mgr, p.iters_out, steps_done, true);
    // R lanes per class, lane r = replicate r: the R lanes of a class read one 64-byte sector of x per
    // id together and four ids are in flight per lane (one lane per class with the replicates in
    // registers had ONE dependent 64-byte gather in flight per lane: 86 us per step against 50 for the
    // rows); every (class, replicate) sum still adds its ids in list order, as the single-problem
    // kernel does.  (The rows kernel the same way -- a wave per row, lane = entry residue x replicate --
    // was measured too: 97 ms per 100 replicates instead of 89; it keeps eight lanes per row.)
    const double *__restrict__ x = p.x[parity];
    const int r = threadIdx.x & (R - 1);
    for (int64_t c = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / R; c < p.n_classes;
         c += ((int64_t)gridDim.x * blockDim.x) / R) {
        const int64_t begin = p.cls_offset[c], end = p.cls_offset[c + 1];
        double s = 0.0;
        int64_t j = begin;
        for (; j + 4 <= end; j += 4) {
            const int32_t t0 = p.ids[j], t1 = p.ids[j + 1], t2 = p.ids[j + 2], t3 = p.ids[j + 3];
            const double x0 = x[(int64_t)t0 * R + r], x1 = x[(int64_t)t1 * R + r];
            const double x2 = x[(int64_t)t2 * R + r], x3 = x[(int64_t)t3 * R + r];
            s += x0; s += x1; s += x2; s += x3;
        }
        for (; j < end; ++j) s += x[(int64_t)p.ids[j] * R + r];
        p.inner[c * R + r] = s / p.cls_count[c * R + r];
    }
}

__global__ void __launch_bounds__(256)
em_rows_batch_kernel(EmBatchProblem p, int parity)
{
    if (p.ctl[BCTL_ALL_DONE]) return;
    const double *__restrict__ x = p.x[parity];
    const int sub = threadIdx.x & 7;
    for (int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 3; row < p.n_rows;
         row += ((int64_t)gridDim.x * blockDim.x) >> 3) {
        const int64_t begin = p.row_start[row], end = p.row_start[row + 1];
        const double *__restrict__ xt = x + (int64_t)p.row_tx[row] * R;
        double xr[R], s[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { xr[r] = xt[r]; s[r] = 0.0; }
        for (int64_t e = begin + sub; e < end; e += 8) {
            const double *__restrict__ inner = p.inner + (int64_t)p.tx_cls[e] * R;
#pragma unroll
            for (int r = 0; r < R; ++r) s[r] += xr[r] / inner[r];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            s[r] += __shfl_xor(s[r], 4, 8);
            s[r] += __shfl_xor(s[r], 2, 8);
            s[r] += __shfl_xor(s[r], 1, 8);
        }
        if (sub == 0) {
            double *__restrict__ out = p.row_sum + row * R;
#pragma unroll
            for (int r = 0; r < R; ++r) out[r] = s[r];
        }
    }
}

__global__ void __launch_bounds__(256)
em_finalize_batch_kernel(EmBatchProblem p, int parity)
{
    if (p.ctl[BCTL_ALL_DONE]) return;
    // R lanes per transcript, lane r = replicate r (as in em_inner_batch): a lane follows ONE chain of
    // dependent loads instead of eight, and there are eight times the lanes to hide it.  The
    // arithmetic per (transcript, replicate) is unchanged; maxima and flags do not care about order.
    __shared__ double s_max[4][R];
    __shared__ unsigned int s_flags[4][R];
    const double *__restrict__ x_old = p.x[parity];
    double *__restrict__ x_new = p.x[parity ^ 1];
    const int r = threadIdx.x & (R - 1);
    const bool stopped = p.ctl[BCTL_DONE + r] != 0;           // this lane's replicate is frozen
    double local_max = 0.0;
    unsigned int flags = 0;
    for (int64_t t = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / R; t < p.n_tx;
         t += ((int64_t)gridDim.x * blockDim.x) / R) {
        double a = 0.0;
        for (int64_t row = p.tx_row[t]; row < p.tx_row[t + 1]; ++row) a += p.row_sum[row * R + r];
        const double before = x_old[t * R + r];
        double v = a / p.eff_len[t] / p.n_total;                 // infer.py:158
        if (v != v) v = 0.0;                                     // infer.py:159
        if (stopped) {
            x_new[t * R + r] = before;                           // a stopped replicate keeps its result
        } else {
            x_new[t * R + r] = v;
            if (v > p.x_floor) {                                 // infer.py:160
                const double change = fabs(v - before) / v;
                if (change != change) flags |= 2u;
                else if (change > local_max) local_max = change;
                flags |= 1u;
            }
        }
    }
    const int wave = threadIdx.x >> 6;
    for (int d = 32; d >= R; d >>= 1) {                          // over the lanes of the wave that hold replicate r
        const double o = __shfl_xor(local_max, d, 64);
        local_max = o > local_max ? o : local_max;
        flags |= __shfl_xor(flags, d, 64);
    }
    if ((threadIdx.x & 63) < R) { s_max[wave][r] = local_max; s_flags[wave][r] = flags; }
    __syncthreads();
    if (threadIdx.x < R) {
        double m = s_max[0][r];
        unsigned int f = s_flags[0][r];
        for (int w = 1; w < 4; ++w) { m = s_max[w][r] > m ? s_max[w][r] : m; f |= s_flags[w][r]; }
        // a frozen replicate reports "selected, no change": it stays stopped whatever is judged
        if (stopped) { m = 0.0; f = 1u; }
        p.part_max[blockIdx.x * R + r] = m;
        p.part_flags[blockIdx.x * R + r] = f;
    }
}

// em_rows_batch and em_finalize_batch in ONE launch (skm_em.hip: em_rows_finalize_kernel is the
// single-problem form).  After the row's eight partial sums have been combined every lane of the
// 8-lane group holds all eight replicates' row sums; lane r finalizes replicate r of the row's
// transcript -- all eight lanes busy, with em_finalize_batch_kernel's arithmetic (a = 0.0 + row sum)
// -- when the transcript has this one row (every transcript of the benchmarked tables has), and for a
// transcript of several rows the group whose row arrives last adds them up in row order (row sums
// cross blocks as agent-scope atomic stores / loads, the store completed before the arrival counts).
__global__ void __launch_bounds__(256)
em_rows_finalize_batch_kernel(EmBatchProblem p, int parity)
{
    if (p.ctl[BCTL_ALL_DONE]) return;
    __shared__ double s_max[4][R];
    __shared__ unsigned int s_flags[4][R];
    const double *__restrict__ x = p.x[parity];
    double *__restrict__ x_new = p.x[parity ^ 1];
    const int sub = threadIdx.x & 7;
    const bool stopped = p.ctl[BCTL_DONE + sub] != 0;          // this lane's replicate (r = sub) is frozen
    double local_max = 0.0;
    unsigned int flags = 0;
    for (int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 3; row < p.n_rows;
         row += ((int64_t)gridDim.x * blockDim.x) >> 3) {
        const int64_t begin = p.row_start[row], end = p.row_start[row + 1];
        const int32_t t = p.row_tx[row];
        const double *__restrict__ xt = x + (int64_t)t * R;
        const int64_t first_row = p.tx_row[t], rows_of_t = p.tx_row[t + 1] - first_row;
        const double eff = p.eff_len[t];
        double xr[R], s[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { xr[r] = xt[r]; s[r] = 0.0; }
        for (int64_t e = begin + sub; e < end; e += 8) {
            const double *__restrict__ inner = p.inner + (int64_t)p.tx_cls[e] * R;
#pragma unroll
            for (int r = 0; r < R; ++r) s[r] += xr[r] / inner[r];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            s[r] += __shfl_xor(s[r], 4, 8);
            s[r] += __shfl_xor(s[r], 2, 8);
            s[r] += __shfl_xor(s[r], 1, 8);
        }
        double mine = s[0], before = xr[0];                    // replicate `sub` of this row
#pragma unroll
        for (int r = 1; r < R; ++r) { mine = sub == r ? s[r] : mine; before = sub == r ? xr[r] : before; }
        double a = 0.0;
        if (rows_of_t == 1) {
            a += mine;
        } else {
            unsigned long long *const sums = reinterpret_cast<unsigned long long *>(p.row_sum);
            __hip_atomic_store(&sums[row * R + sub], (unsigned long long)__double_as_longlong(mine), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // (the eight stores of the group are complete: the wait above is per wave, and the group is in one)
            unsigned int arrived = 0;
            if (sub == 0) arrived = atomicAdd(&p.arrivals[t], 1u) + 1u;
            arrived = __shfl(arrived, 0, 8);
            if ((int64_t)arrived != rows_of_t) continue;          // another row of this transcript is still to come
            for (int64_t k = 0; k < rows_of_t; ++k)
                a += __longlong_as_double((long long)__hip_atomic_load(&sums[(first_row + k) * R + sub], __ATOMIC_RELAXED,
                                                                       __HIP_MEMORY_SCOPE_AGENT));
            if (sub == 0) atomicExch(&p.arrivals[t], 0u);         // (for the next step)
        }
        double v = a / eff / p.n_total;                           // infer.py:158
        if (v != v) v = 0.0;                                      // infer.py:159
        if (stopped) {
            x_new[(int64_t)t * R + sub] = before;                 // a stopped replicate keeps its result
        } else {
            x_new[(int64_t)t * R + sub] = v;
            if (v > p.x_floor) {                                  // infer.py:160
                const double change = fabs(v - before) / v;
                if (change != change) flags |= 2u;
                else if (change > local_max) local_max = change;
                flags |= 1u;
            }
        }
    }
    const int wave = threadIdx.x >> 6;
    for (int d = 32; d >= R; d >>= 1) {                          // over the lanes of the wave that hold replicate `sub`
        const double o = __shfl_xor(local_max, d, 64);
        local_max = o > local_max ? o : local_max;
        flags |= __shfl_xor(flags, d, 64);
    }
    if ((threadIdx.x & 63) < R) { s_max[wave][sub] = local_max; s_flags[wave][sub] = flags; }
    __syncthreads();
    if (threadIdx.x < R) {
        double m = s_max[0][sub];
        unsigned int f = s_flags[0][sub];
        for (int w = 1; w < 4; ++w) { m = s_max[w][sub] > m ? s_max[w][sub] : m; f |= s_flags[w][sub]; }
        if (stopped) { m = 0.0; f = 1u; }                        // (see em_finalize_batch_kernel)
        p.part_max[blockIdx.x * R + sub] = m;
        p.part_flags[blockIdx.x * R + sub] = f;
    }
}

__global__ void __launch_bounds__(256)
em_decide_batch_kernel(EmBatchProblem p, int n_parts, int64_t steps_done)
{
    __shared__ int s_all;
    if (threadIdx.x == 0) s_all = p.ctl[BCTL_ALL_DONE] != 0;
    __syncthreads();
    if (s_all) return;
    evaluate_batch(p, n_parts, steps_done, true);
}

// out[t] = x[t][r]
__global__ void __launch_bounds__(256)
em_batch_take_kernel(const double *__restrict__ x, int64_t n_tx, int r, double *__restrict__ out)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_tx;
         i += (int64_t)gridDim.x * blockDim.x) out[i] = x[i * R + r];
}

// ---- the working set kept full by the device (`-b N`): after every step one small kernel decides,
// place by place, what a latched stop means -- the replicate's step count is recorded, its result is
// to be taken, the next replicate of the group (if any) is to be put in its place -- and a grid-wide
// kernel carries the plan out: result out, pre-drawn class counts and the common start vector in,
// the place's partials of "the step before" set to "still changing".  The host only queues steps
// and looks now and then whether everything has finished.  mgr words: [0] next replicate to start,
// [1] replicates in the group, [2] finished, [3] a replicate had no abundance above x_floor,
// [8 + r] replicate in place r + 1 (0: idle), [16 + r] the step it started at, [24 + r] / [32 + r]
// the plan of this step: replicate to take / to put, + 1.
__global__ void em_batch_assign_kernel(EmBatchProblem p, unsigned long long *mgr, int64_t *iters_out, int64_t step)
{
    if (threadIdx.x != 0 || p.ctl[BCTL_ALL_DONE]) return;
    plan_places(p, mgr, iters_out, step, false);
}

// the plan carried out; `into` = the abundance buffer the NEXT step reads (a stopped replicate's
// result is carried there by the finalize pass of this step)
__global__ void __launch_bounds__(256)
em_batch_refill_kernel(EmBatchProblem p, const unsigned long long *__restrict__ mgr, const double *__restrict__ counts_all,
                       const double *__restrict__ x_start, double *__restrict__ out_all, double *__restrict__ into,
                       int64_t step)
{
    __shared__ unsigned long long plan[2 * R];
    if (threadIdx.x < 2 * R) plan[threadIdx.x] = mgr[MGR_TAKE + threadIdx.x];     // (TAKE and PUT are adjacent)
    __syncthreads();
    if (step >= 0 && blockIdx.x == 0 && threadIdx.x == 0 && !p.ctl[BCTL_ALL_DONE]) {
        // what plan_places left for behind the step: a refilled place starts over, and when no place
        // holds a replicate any more everything has finished
        bool live = false;
        for (int r = 0; r < R; ++r) {
            if (plan[R + r]) { p.ctl[BCTL_DONE + r] = 0; p.ctl[BCTL_UNDEFINED + r] = 0; p.ctl[BCTL_ITERS + r] = 0; }
            live |= mgr[MGR_REP + r] != 0;
        }
        if (!live) { p.ctl[BCTL_LAST_STEP] = (unsigned long long)(step + 1); p.ctl[BCTL_ALL_DONE] = 1; }
    }
    bool any = false;
#pragma unroll
    for (int i = 0; i < 2 * R; ++i) any |= plan[i] != 0;
    if (!any) return;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, first = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    double *cls_count = const_cast<double *>(p.cls_count);
    for (int r = 0; r < R; ++r) {
        const unsigned long long take = plan[r], put = plan[R + r];
        if (take)
            for (int64_t t = first; t < p.n_tx; t += stride) out_all[(int64_t)(take - 1) * p.n_tx + t] = into[t * R + r];
        if (put) {
            for (int64_t t = first; t < p.n_tx; t += stride) into[t * R + r] = x_start[t];
            const double *counts = counts_all + (int64_t)(put - 1) * p.n_classes;
            for (int64_t c = first; c < p.n_classes; c += stride) cls_count[c * R + r] = counts[c];
            if (first < EM_FINAL_BLOCKS) { p.part_max[first * R + r] = 1e300; p.part_flags[first * R + r] = 1u; }
        }
    }
}

// fresh control block; the places in `idle` hold no replicate and count as stopped
__global__ void em_batch_ctl_kernel(unsigned long long *ctl, unsigned int idle)
{
    const int i = threadIdx.x;
    if (i < 32) ctl[i] = (i >= BCTL_DONE && i < BCTL_DONE + R && ((idle >> (i - BCTL_DONE)) & 1u)) ? 1ULL : 0ULL;
}

inline unsigned grid_of(int64_t items, int per_block, int64_t cap = 256 * 8)
{
    int64_t blocks = (items + per_block - 1) / per_block;
    if (blocks < 1) blocks = 1;
    if (blocks > cap) blocks = cap;
    return (unsigned)blocks;
}

}  // namespace

int em_batch_final_blocks(const EmBatchProblem &p)
{
    // (fused: the blocks of em_rows_finalize_batch, 32 rows each; else a lane per transcript and replicate)
    int64_t blocks = p.fused ? (p.n_rows + 31) / 32 : (p.n_tx * EM_BATCH + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > EM_FINAL_BLOCKS) blocks = EM_FINAL_BLOCKS;
    return (int)blocks;
}

void launch_em_batch_step(const EmBatchProblem &p, int64_t step, hipStream_t stream)
{
    const int parity = (int)(step & 1);
    hipLaunchKernelGGL(em_inner_batch_kernel, dim3(grid_of(p.n_classes, 256 / EM_BATCH)), dim3(256), 0, stream, p, parity,
                       step > 0 ? em_batch_final_blocks(p) : 0, step);
    if (p.fused) {
        hipLaunchKernelGGL(em_rows_finalize_batch_kernel, dim3((unsigned)em_batch_final_blocks(p)), dim3(256), 0, stream, p,
                           parity);
        return;
    }
    hipLaunchKernelGGL(em_rows_batch_kernel, dim3(grid_of(p.n_rows, 32)), dim3(256), 0, stream, p, parity);
    hipLaunchKernelGGL(em_finalize_batch_kernel, dim3((unsigned)em_batch_final_blocks(p)), dim3(256), 0, stream, p,
                       parity);
}

// the device-managed working set: set up for `n_reps` replicates (places 0 .. min(n_reps, 8) - 1 filled
// from counts_all[0 ..]), then after every step em_batch_manage
void launch_em_batch_manage_init(const EmBatchProblem &p, unsigned long long *mgr, unsigned long long *host, int64_t n_reps,
                                 const double *counts_all, const double *x_start, double *out_all, hipStream_t stream)
{
    for (int i = 0; i < 64; ++i) host[i] = 0;          // (64 words of page-locked memory of the caller's)
    const int filled = (int)std::min<int64_t>(n_reps, EM_BATCH);
    host[MGR_NEXT] = (unsigned long long)filled;
    host[MGR_COUNT] = (unsigned long long)n_reps;
    unsigned int idle = 0;
    for (int r = 0; r < EM_BATCH; ++r) {
        if (r < filled) { host[MGR_REP + r] = (unsigned long long)r + 1; host[MGR_PUT + r] = (unsigned long long)r + 1; }
        else idle |= 1u << r;
    }
    (void)hipMemcpyAsync(mgr, host, 64 * sizeof(unsigned long long), hipMemcpyHostToDevice, stream);
    hipLaunchKernelGGL(em_batch_ctl_kernel, dim3(1), dim3(64), 0, stream, p.ctl, idle);
    hipLaunchKernelGGL(em_batch_refill_kernel, dim3(1024), dim3(256), 0, stream, p, mgr, counts_all, x_start, out_all, p.x[0],
                       (int64_t)-1);
}

// behind a step whose em_inner_batch has planned (p.mgr set): the refill alone; `planned` false (a
// look between chunks, behind em_batch_decide): the plan first, as a launch of its own
void launch_em_batch_manage(const EmBatchProblem &p, unsigned long long *mgr, const double *counts_all, const double *x_start,
                            double *out_all, int64_t *iters_out, int64_t step, bool planned, hipStream_t stream)
{
    if (!planned) hipLaunchKernelGGL(em_batch_assign_kernel, dim3(1), dim3(64), 0, stream, p, mgr, iters_out, step);
    hipLaunchKernelGGL(em_batch_refill_kernel, dim3(1024), dim3(256), 0, stream, p, mgr, counts_all, x_start, out_all,
                       p.x[(step + 1) & 1], step);
}

void launch_em_batch_decide(const EmBatchProblem &p, int64_t steps_done, hipStream_t stream)
{
    hipLaunchKernelGGL(em_decide_batch_kernel, dim3(1), dim3(256), 0, stream, p, em_batch_final_blocks(p), steps_done);
}

void launch_em_batch_take(const double *x, int64_t n_tx, int r, double *out, hipStream_t stream)
{
    hipLaunchKernelGGL(em_batch_take_kernel, dim3(grid_of(n_tx, 256)), dim3(256), 0, stream, x, n_tx, r, out);
}

void launch_em_batch_ctl(unsigned long long *ctl, unsigned int idle, hipStream_t stream)
{
    hipLaunchKernelGGL(em_batch_ctl_kernel, dim3(1), dim3(64), 0, stream, ctl, idle);
}

// (skm_index_create loads every code object of the library before the first sample needs it: the first
// launch out of a translation unit otherwise pays its load, tens of milliseconds, inside the run)
void warm_code_em_batch()
{
    hipFuncAttributes attributes;
    (void)hipFuncGetAttributes(&attributes, reinterpret_cast<const void *>(&em_inner_batch_kernel));
}

}  // namespace skm
