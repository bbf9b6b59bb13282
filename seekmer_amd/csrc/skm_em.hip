// Quantification kernels for gfx950: effective lengths, the EM step, the
// multinomial bootstrap draw.  They replace the numpy loop of
// /root/reference/seekmer/infer.py:133-168 (em), mapper.py:134-141
// (effective_lengths) and the scipy draw of infer.py:108-111.
//
// One EM step is a two-sided gather over two CSR views of the same
// (class, transcript) pairs, four launches on one stream and no
// floating-point atomics, so every step is bitwise reproducible:
//   em_inner     -- one lane per class: S_c = sum of x over the tuple in tuple
//                   order (as numpy.bincount accumulates it), inner_c = S_c /
//                   count_c                                   (infer.py:155-156)
//   em_rows      -- 8 lanes per row (a run of <= 512 classes of ONE transcript):
//                   sum of x_t / inner_c over the run          (infer.py:157)
//   em_finalize  -- one lane per transcript: x'_t = (sum of its rows) / l_t /
//                   n, NaN -> 0, relative change against x_t; block partials
//   (judging)    -- the reference's stopping rule (infer.py:160) over the block
//                   partials is evaluated at the head of the next em_inner
//                   launch (by its block 0) and by a one-block
//                   em_decide launch at the end of each enqueued chunk; it
//                   latches `done`, after which every later launch is a no-op
//                   -- the host enqueues steps in chunks and still stops at
//                   exactly the reference's iteration count.
// All three are gather/stream kernels bound by HBM/L2 bandwidth: no MFMA.
#include "skm_kernels.h"

namespace skm {

enum { CTL_DONE = 0, CTL_ITERS = 1, CTL_UNDEFINED = 3 };

__device__ bool em_evaluate(const EmProblem &p, int n_parts, int64_t steps_done, bool publish);

// eval_parts > 0: the finalize pass before this launch (number `steps_done`) has not been
// judged yet -- do it here, in every block, before starting the next step
__global__ void __launch_bounds__(256)
em_inner_kernel(EmProblem p, int parity, int eval_parts, int64_t steps_done)
{
    // the first class's row is fetched before the verdict on the previous step is known: its
    // latency then runs under the judging instead of after it
    const int64_t c_first = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    int64_t begin_first = 0, end_first = 0;
    double count_first = 1.0;
    if (c_first < p.n_classes) {
        begin_first = p.cls_offset[c_first];
        end_first = p.cls_offset[c_first + 1];
        count_first = p.cls_count[c_first];
    }
    if (eval_parts > 0 && blockIdx.x == 0) {
        // Block 0 judges the step before this one and latches the verdict; the other blocks do not
        // wait for it.  If the EM has just stopped they compute one pass of `inner` that nobody
        // reads (x is not touched by this kernel, and every later launch sees the latch and
        // returns): 13 us once per EM, against every block re-reading all the partials every step
        // (34.5 -> 32.8 us per step).
        if (p.ctl[CTL_DONE]) return;            // (block-uniform)
        if (em_evaluate(p, eval_parts, steps_done, true)) return;
    } else if (p.ctl[CTL_DONE]) {
        return;
    }
    const double *__restrict__ x = p.x[parity];
    for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < p.n_classes;
         c += (int64_t)gridDim.x * blockDim.x) {
        const int64_t begin = c == c_first ? begin_first : p.cls_offset[c];
        const int64_t end = c == c_first ? end_first : p.cls_offset[c + 1];
        const double count = c == c_first ? count_first : p.cls_count[c];
        double s = 0.0;
        int64_t j = begin;
        for (; j + 4 <= end; j += 4) {          // four independent gathers in flight, summed in order
            const int32_t t0 = p.ids[j], t1 = p.ids[j + 1], t2 = p.ids[j + 2], t3 = p.ids[j + 3];
            const double x0 = x[t0], x1 = x[t1], x2 = x[t2], x3 = x[t3];
            s += x0; s += x1; s += x2; s += x3;
        }
        for (; j < end; ++j) s += x[p.ids[j]];
        p.inner[c] = s / count;
    }
}

__global__ void __launch_bounds__(256)
em_rows_kernel(EmProblem p, int parity)
{
    if (p.ctl[CTL_DONE]) return;
    const double *__restrict__ x = p.x[parity];
    const int sub = threadIdx.x & 7;
    for (int64_t r = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 3; r < p.n_rows;
         r += ((int64_t)gridDim.x * blockDim.x) >> 3) {
        const int64_t begin = p.row_start[r], end = p.row_start[r + 1];
        const double xt = x[p.row_tx[r]];
        double s = 0.0;
        int64_t e = begin + sub;
        for (; e + 8 < end; e += 16) {          // two independent gathers in flight per lane
            const int32_t c0 = p.tx_cls[e], c1 = p.tx_cls[e + 8];
            const double i0 = p.inner[c0], i1 = p.inner[c1];
            s += xt / i0;
            s += xt / i1;
        }
        for (; e < end; e += 8) s += xt / p.inner[p.tx_cls[e]];
        s += __shfl_xor(s, 4, 8);
        s += __shfl_xor(s, 2, 8);
        s += __shfl_xor(s, 1, 8);
        if (sub == 0) p.row_sum[r] = s;
    }
}

// em_rows and em_finalize in ONE launch (one rank: nothing to all-reduce between them).  Every
// transcript has at least one row (skm_quant_setup.hip), nearly every transcript exactly one: the
// 8-lane group that has summed such a row finalizes its transcript on the spot -- x'_t, NaN -> 0,
// the relative change -- with the arithmetic of em_finalize_kernel (a = 0.0 + row sum), and the
// block's partials of the stopping rule come out of this kernel.  The rows of a transcript that
// sits in more than EM_ROW_CAP classes may be summed by different blocks; the group whose row
// arrives LAST adds them up in row order and finalizes: row sums cross blocks as 8-byte
// agent-scope atomic stores and loads (write-through / L2-bypassing: the XCDs' L2s are not
// coherent), the store completed (s_waitcnt) before the arrival is counted.  One launch (5.1 us)
// and one launch gap less per step; bit for bit em_rows + em_finalize.
// TO_ACC (several ranks): the transcript's numerator goes to p.acc instead -- em_rows +
// em_rows_to_acc in one launch -- for the all-reduce that sits in front of em_finalize there.
template <bool TO_ACC>
__global__ void __launch_bounds__(256, 8)       // (8 waves per SIMD: the 2048-block grid is resident at once)
em_rows_finalize_kernel(EmProblem p, int parity)
{
    if (p.ctl[CTL_DONE]) return;
    __shared__ double s_max[4];
    __shared__ unsigned int s_flags[4];
    const double *__restrict__ x = p.x[parity];
    double *__restrict__ x_new = p.x[parity ^ 1];
    const int sub = threadIdx.x & 7;
    double local_max = 0.0;
    unsigned int flags = 0;
    for (int64_t r = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 3; r < p.n_rows;
         r += ((int64_t)gridDim.x * blockDim.x) >> 3) {
        const int64_t begin = p.row_start[r], end = p.row_start[r + 1];
        const int32_t t = p.row_tx[r];
        const double xt = x[t];
        // (what the finalize needs is asked for with x[t]: one round trip for all of it)
        const int64_t first_row = p.tx_row[t], rows_of_t = p.tx_row[t + 1] - first_row;
        const double eff = p.eff_len[t];
        double s = 0.0;
        int64_t e = begin + sub;
        for (; e + 8 < end; e += 16) {          // two independent gathers in flight per lane
            const int32_t c0 = p.tx_cls[e], c1 = p.tx_cls[e + 8];
            const double i0 = p.inner[c0], i1 = p.inner[c1];
            s += xt / i0;
            s += xt / i1;
        }
        for (; e < end; e += 8) s += xt / p.inner[p.tx_cls[e]];
        s += __shfl_xor(s, 4, 8);
        s += __shfl_xor(s, 2, 8);
        s += __shfl_xor(s, 1, 8);
        // 1: the transcript's only row; 2: the last of its rows to arrive (this group adds them up); 0: neither
        int mode = 0;
        unsigned long long *const sums = reinterpret_cast<unsigned long long *>(p.row_sum);
        if (sub == 0) {
            if (rows_of_t == 1) {
                mode = 1;
            } else {
                __hip_atomic_store(&sums[r], (unsigned long long)__double_as_longlong(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned int before = atomicAdd(&p.arrivals[t], 1u);
                mode = (int64_t)before + 1 == rows_of_t ? 2 : 0;
            }
        }
        mode = __shfl(mode, 0, 8);
        if (mode == 0) continue;
        double a = 0.0;
        if (mode == 1) {
            a += s;
        } else {
            // (the eight lanes fetch eight row sums at a time; they are added in row order, as
            // em_finalize_kernel adds them: a transcript in 100 000 classes has 200 rows)
            for (int64_t k0 = 0; k0 < rows_of_t; k0 += 8) {
                const int64_t k = k0 + sub;
                const double mine = k < rows_of_t
                    ? __longlong_as_double((long long)__hip_atomic_load(&sums[first_row + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                    : 0.0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const double v = __shfl(mine, j, 8);
                    if (k0 + j < rows_of_t) a += v;
                }
            }
            if (sub == 0) atomicExch(&p.arrivals[t], 0u);             // (for the next step)
        }
        if (sub != 0) continue;
        if (TO_ACC) {
            p.acc[t] = a;
            continue;
        }
        double v = a / eff / p.n_total;                               // infer.py:158
        if (v != v) v = 0.0;                                          // infer.py:159
        x_new[t] = v;
        if (v > p.x_floor) {                                          // infer.py:160
            const double change = fabs(v - xt) / v;
            if (change != change) flags |= 2u;
            else if (change > local_max) local_max = change;
            flags |= 1u;
        }
    }
    for (int d = 32; d > 0; d >>= 1) {
        const double o = __shfl_xor(local_max, d, 64);
        local_max = o > local_max ? o : local_max;
        flags |= __shfl_xor(flags, d, 64);
    }
    if (TO_ACC) return;                         // (em_finalize judges the step there)
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_max[wave] = local_max; s_flags[wave] = flags; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = s_max[0];
        unsigned int f = s_flags[0];
        for (int w = 1; w < 4; ++w) { m = s_max[w] > m ? s_max[w] : m; f |= s_flags[w]; }
        p.part_max[blockIdx.x] = m;             // judged by em_evaluate in the next launch
        p.part_flags[blockIdx.x] = f;
    }
}

// multi-GPU only (the unfused form, SKM_EM_UNFUSED): rows -> per-transcript numerators for the all-reduce
__global__ void __launch_bounds__(256)
em_rows_to_acc_kernel(EmProblem p)
{
    if (p.ctl[CTL_DONE]) return;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < p.n_tx;
         t += (int64_t)gridDim.x * blockDim.x) {
        double a = 0.0;
        for (int64_t r = p.tx_row[t]; r < p.tx_row[t + 1]; ++r) a += p.row_sum[r];
        p.acc[t] = a;
    }
}

template <bool FROM_ACC>
__global__ void __launch_bounds__(256)
em_finalize_kernel(EmProblem p, int parity)
{
    if (p.ctl[CTL_DONE]) return;
    __shared__ double s_max[4];
    __shared__ unsigned int s_flags[4];
    const double *__restrict__ x_old = p.x[parity];
    double *__restrict__ x_new = p.x[parity ^ 1];
    double local_max = 0.0;
    unsigned int flags = 0;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < p.n_tx;
         t += (int64_t)gridDim.x * blockDim.x) {
        double a;
        if (FROM_ACC) {
            a = p.acc[t];
        } else {
            a = 0.0;
            for (int64_t r = p.tx_row[t]; r < p.tx_row[t + 1]; ++r) a += p.row_sum[r];
        }
        double v = a / p.eff_len[t] / p.n_total;                 // infer.py:158
        if (v != v) v = 0.0;                                     // infer.py:159
        x_new[t] = v;
        if (v > p.x_floor) {                                     // infer.py:160
            const double r = fabs(v - x_old[t]) / v;
            if (r != r) flags |= 2u;
            else if (r > local_max) local_max = r;
            flags |= 1u;
        }
    }
    for (int d = 32; d > 0; d >>= 1) {
        const double o = __shfl_xor(local_max, d, 64);
        local_max = o > local_max ? o : local_max;
        flags |= __shfl_xor(flags, d, 64);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_max[wave] = local_max; s_flags[wave] = flags; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = s_max[0];
        unsigned int f = s_flags[0];
        for (int w = 1; w < 4; ++w) { m = s_max[w] > m ? s_max[w] : m; f |= s_flags[w]; }
        p.part_max[blockIdx.x] = m;             // judged by em_evaluate in the next launch
        p.part_flags[blockIdx.x] = f;
    }
}

// The stopping rule (infer.py:160) over the per-block partials of finalize pass number
// `steps_done` (1-based).  Every lane of the calling block takes part and gets the verdict;
// only the caller with `publish` set writes it to the control block.  It is evaluated at the
// head of the NEXT step's em_inner launch by that launch's block 0 (744 partials from L2 are
// cheaper than a launch of their own, 4.4 us), and by a one-block launch at the end of each
// enqueued chunk so that the host can read the state.
__device__ bool em_evaluate(const EmProblem &p, int n_parts, int64_t steps_done, bool publish)
{
    __shared__ double s_max[4];
    __shared__ unsigned int s_flags[4];
    __shared__ int s_done;
    double m = 0.0;
    unsigned int f = 0;
    for (int b = threadIdx.x; b < n_parts; b += blockDim.x) {
        const double o = p.part_max[b];
        m = o > m ? o : m;
        f |= p.part_flags[b];
    }
    for (int d = 32; d > 0; d >>= 1) {
        const double o = __shfl_xor(m, d, 64);
        m = o > m ? o : m;
        f |= __shfl_xor(f, d, 64);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_max[wave] = m; s_flags[wave] = f; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 0; w < 4; ++w) { m = s_max[w] > m ? s_max[w] : m; f |= s_flags[w]; }
        bool done, undefined = false;
        if (p.fixed_iters > 0) {
            done = steps_done >= p.fixed_iters;
        } else if (!(f & 1u)) {
            undefined = true;                 // numpy raises on max() of an empty selection
            done = true;
        } else {
            done = (f & 2u) || !(m > p.rel_tol);                 // NaN propagates through max()
            if (p.max_iters > 0 && steps_done >= p.max_iters) done = true;
        }
        if (publish) {
            if (undefined) p.ctl[CTL_UNDEFINED] = 1;
            p.ctl[CTL_ITERS] = (unsigned long long)steps_done;
            p.ctl[CTL_DONE] = done ? 1ULL : 0ULL;
        }
        s_done = done ? 1 : 0;
    }
    __syncthreads();
    return s_done != 0;
}

__global__ void __launch_bounds__(256)
em_decide_kernel(EmProblem p, int n_parts, int64_t steps_done)
{
    if (p.ctl[CTL_DONE]) return;
    em_evaluate(p, n_parts, steps_done, true);
}

// MapResult.effective_lengths, mapper.py:134-141: p = fld / fld.sum();
// eff_t = sum_i max(len_t - i, 1) * p_i accumulated for i = 0..1999 in order
// (compiled with -ffp-contract=off: separate multiply and add, as numpy).
__global__ void __launch_bounds__(256)
effective_lengths_kernel(const unsigned long long *__restrict__ fld,
                         const double *__restrict__ lengths, int64_t n_tx, double *__restrict__ out)
{
    // Only the bins with p_i != 0 are visited: a term max(len - i, 1) * 0.0 is +0.0 and adding
    // +0.0 leaves the running sum as it is, bit for bit (the sum starts at +0.0 and every term
    // is >= 0); a NaN bin (empty histogram: 0 / 0) is not zero and stays in.  A fragment-length
    // histogram has a few hundred occupied bins out of 2000.
    __shared__ double p[MAX_FRAGMENT_LENGTH];
    __shared__ int bin[MAX_FRAGMENT_LENGTH];
    __shared__ unsigned long long counts[MAX_FRAGMENT_LENGTH];
    __shared__ unsigned long long part[4];
    __shared__ int n_bins;
    // the histogram once into LDS (coalesced), its total by a block reduction (integers: any order)
    unsigned long long mine = 0;
    for (int i = threadIdx.x; i < MAX_FRAGMENT_LENGTH; i += blockDim.x) {
        counts[i] = fld[i];
        mine += counts[i];
    }
    for (int d = 32; d > 0; d >>= 1) mine += __shfl_xor(mine, d, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mine;
    __syncthreads();
    const double total = (double)(long long)(part[0] + part[1] + part[2] + part[3]);
    // wave 0 packs the non-zero bins in bin order (ballot + prefix count per 64 bins)
    if (threadIdx.x < 64) {
        int n = 0;
        for (int first = 0; first < MAX_FRAGMENT_LENGTH; first += 64) {
            const int i = first + (int)threadIdx.x;
            const double v = i < MAX_FRAGMENT_LENGTH ? (double)(long long)counts[i] / total : 0.0;
            const bool keep = i < MAX_FRAGMENT_LENGTH && !(v == 0.0);
            const unsigned long long kept = __ballot(keep);
            if (keep) {
                const int at = n + __popcll(kept & ((1ULL << threadIdx.x) - 1));
                p[at] = v;
                bin[at] = i;
            }
            n += __popcll(kept);
        }
        if (threadIdx.x == 0) n_bins = n;
    }
    __syncthreads();
    const int n = n_bins;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n_tx;
         t += (int64_t)gridDim.x * blockDim.x) {
        const double len = lengths[t];
        double acc = 0.0;
        for (int k = 0; k < n; ++k) {
            double v = len - (double)bin[k];
            if (v < 1.0) v = 1.0;
            acc += v * p[k];
        }
        out[t] = acc;
    }
}

// Counter-based generator for the bootstrap draw (the reference draws from
// numpy's unseeded global generator, so only the distribution can be matched).
__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// multinomial(n, count / n) -- the draw of seekmer/infer.py:108-111 -- as n categorical draws
// over the integer cumulative counts, in two stages so that every draw is decided and counted
// in LDS: the classes are cut into tiles of `tile` classes (at most MN_TILES tiles, at most
// MN_TILES classes each);
//   stage 1  n draws r ~ U[0, total) choose a TILE (LDS counters, one global add per tile and
//            workgroup): the tile totals are multinomial(n, tile masses);
//   stage 2  one workgroup per tile makes its tile's m draws among the tile's classes
//            (cumulative counts relative to the tile in LDS, LDS counters) and writes the counts.
// n iid draws sorted into tiles and then, given the tile totals, iid within the tiles under the
// conditional probabilities ARE n iid draws over the classes: exactly the multinomial, only the
// random stream differs from drawing class by class.  (One draw per class-level binary search
// and one device-scope atomic per draw was 1.45 ms for 20 M draws over 1 M classes: the
// scattered atomics alone are bound at ~20 G/s.)  Counter-based generator: the reference draws
// from numpy's unseeded global generator, so only the distribution can be matched.
//
// A draw costs what its arithmetic costs (20 M draws are ~100 G integer multiplies when every
// draw takes its own 64-bit hash and a 64 x 64 multiply-high; 32-bit multiplies issue at a
// quarter of the rate), so: one 64-bit hash serves TWO draws, each an exactly uniform integer
// below the range from 32 bits by multiply-high with rejection (Lemire; rejected with
// probability range / 2^32, then redrawn from a stream keyed by the draw's number), and the interval holding r
// is found from a guide table over power-of-two buckets of r (one LDS read, then 1-2 steps
// forward) instead of a 12-step binary search.
constexpr int MN_TILES = 4096;
constexpr int MN_GUIDE_BITS = 12;

struct MnStream {
    uint64_t base, spare;      // counter-mode keys of the paired draws and of the redraws
    __device__ __forceinline__ MnStream(uint64_t seed, uint64_t stream_id, uint64_t part)
    {
        base = mix64(seed ^ (stream_id * 0x9E3779B97F4A7C15ULL) ^ (part * 0xC2B2AE3D27D4EB4FULL));
        spare = mix64(base ^ 0xA0761D6478BD642FULL);
    }
    // the two 32-bit words of draw pair number `pair`
    __device__ __forceinline__ uint64_t words(uint64_t pair) const { return mix64(base + pair * 0xD1342543DE82EF95ULL); }
};
// exactly uniform in [0, range), range >= 1, from 32 random bits; `reject_below` = 2^32 mod range.
// A rejected word is replaced from a stream keyed by (pair, which of its two draws, attempt): the
// result depends on the draw's number alone, not on which lane of which launch geometry makes it.
__device__ __forceinline__ uint32_t mn_bounded(uint32_t word, uint32_t range, uint32_t reject_below,
                                               const MnStream &rng, uint64_t pair, uint32_t which)
{
    uint64_t m = (uint64_t)word * range;
    for (uint64_t attempt = 1; (uint32_t)m < reject_below; ++attempt) {
        const uint64_t again = mix64(rng.spare + pair * 0xD1342543DE82EF95ULL + (2 * attempt + which) * 0x9FB21C651E98DF25ULL);
        m = (uint64_t)(uint32_t)(again >> 32) * range;
    }
    return (uint32_t)(m >> 32);
}

// Interval search over ascending cumulative counts cum[0..n) (cum[n-1] = total > r): guide[g] =
// first index whose cumulative count exceeds the smallest r of bucket g = r >> shift.
struct MnGuide {
    int shift;
    __device__ __forceinline__ static int shift_for(uint32_t total)
    {
        const int bits = 32 - __clz(total - 1 | 1u);                 // total <= 2^bits
        return max(0, bits - MN_GUIDE_BITS);
    }
};
// every thread i < n fills the buckets whose smallest r lies in [cum[i-1], cum[i])
__device__ __forceinline__ void mn_fill_guide(const uint32_t *cum, int n, int shift, uint16_t *guide)
{
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t lo = i ? cum[i - 1] : 0u, hi = cum[i];
        const uint32_t w = 1u << shift;
        // buckets g with g * w in [lo, hi)
        uint32_t g = (uint32_t)(((uint64_t)lo + w - 1) >> shift);
        const uint32_t g_end = (uint32_t)(((uint64_t)hi + w - 1) >> shift);
        for (; g < g_end; ++g) guide[g] = (uint16_t)i;
    }
}
__device__ __forceinline__ int mn_find(const uint32_t *cum, const uint16_t *guide, int shift, uint32_t r)
{
    int i = guide[r >> shift];
    while (cum[i] <= r) ++i;
    return i;
}

__global__ void __launch_bounds__(1024)
multinomial_tiles_kernel(const unsigned long long *__restrict__ cum, int64_t n_classes, int tile, int n_tiles,
                         int64_t n_draws, uint64_t seed, uint64_t stream_id, unsigned int *__restrict__ tile_total)
{
    __shared__ uint32_t tile_cum[MN_TILES];
    __shared__ unsigned int count[MN_TILES];
    __shared__ uint16_t guide[1 << MN_GUIDE_BITS];
    for (int k = threadIdx.x; k < n_tiles; k += blockDim.x) {
        tile_cum[k] = (uint32_t)cum[min(n_classes, (int64_t)(k + 1) * tile) - 1];     // (total < 2^32: checked by the caller)
        count[k] = 0;
    }
    __syncthreads();
    const uint32_t total = tile_cum[n_tiles - 1];
    const int shift = MnGuide::shift_for(total);
    mn_fill_guide(tile_cum, n_tiles, shift, guide);
    __syncthreads();
    const uint32_t reject_below = (uint32_t)(0u - total) % total;
    MnStream rng(seed, stream_id, 0);
    // draws 2p and 2p + 1 share hash number p; the workgroups split the pairs evenly
    const int64_t n_pairs = (n_draws + 1) >> 1;
    const int64_t first = n_pairs * blockIdx.x / gridDim.x, last = n_pairs * (blockIdx.x + 1) / gridDim.x;
    for (int64_t p = first + threadIdx.x; p < last; p += blockDim.x) {
        const uint64_t w = rng.words((uint64_t)p);
        const uint32_t r0 = mn_bounded((uint32_t)(w >> 32), total, reject_below, rng, (uint64_t)p, 0u);
        atomicAdd(&count[mn_find(tile_cum, guide, shift, r0)], 1u);
        if (2 * p + 1 < n_draws) {
            const uint32_t r1 = mn_bounded((uint32_t)w, total, reject_below, rng, (uint64_t)p, 1u);
            atomicAdd(&count[mn_find(tile_cum, guide, shift, r1)], 1u);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < n_tiles; k += blockDim.x)
        if (count[k]) atomicAdd(&tile_total[k], count[k]);
}

// counts[(first_class + i) * stride] = draws of class first_class + i, as f8 (the EM's class counts)
// (the guide keeps its 4096 buckets for a tile of a few hundred classes too: with 512 the scans
// through the light classes of a bucket made this kernel 128 us instead of 72; and 256-lane
// workgroups, four per CU by their LDS, ran 353 us)
__global__ void __launch_bounds__(1024)
multinomial_classes_kernel(const unsigned long long *__restrict__ cum, int64_t n_classes, int tile,
                           const unsigned int *__restrict__ tile_total, uint64_t seed, uint64_t stream_id,
                           double *__restrict__ counts, int stride)
{
    __shared__ uint32_t local_cum[MN_TILES];
    __shared__ unsigned int count[MN_TILES];
    __shared__ uint16_t guide[1 << MN_GUIDE_BITS];
    const int64_t first_class = (int64_t)blockIdx.x * tile;
    const int n = (int)min((int64_t)tile, n_classes - first_class);
    const unsigned long long before = first_class ? cum[first_class - 1] : 0ULL;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        local_cum[i] = (uint32_t)(cum[first_class + i] - before);
        count[i] = 0;
    }
    __syncthreads();
    const uint32_t mass = local_cum[n - 1];
    const unsigned int draws = tile_total[blockIdx.x];       // (0 when the tile has no mass)
    if (draws) {                                             // (uniform over the workgroup)
        const int shift = MnGuide::shift_for(mass);
        mn_fill_guide(local_cum, n, shift, guide);
        __syncthreads();
        const uint32_t reject_below = (uint32_t)(0u - mass) % mass;
        MnStream rng(seed, stream_id, 1 + blockIdx.x);
        const unsigned int n_pairs = (unsigned int)(((unsigned long long)draws + 1) >> 1);    // (draws may be 2^32 - 1)
        for (unsigned int p = threadIdx.x; p < n_pairs; p += blockDim.x) {
            const uint64_t w = rng.words(p);
            const uint32_t r0 = mn_bounded((uint32_t)(w >> 32), mass, reject_below, rng, p, 0u);
            atomicAdd(&count[mn_find(local_cum, guide, shift, r0)], 1u);
            if (2ull * p + 1 < draws) {
                const uint32_t r1 = mn_bounded((uint32_t)w, mass, reject_below, rng, p, 1u);
                atomicAdd(&count[mn_find(local_cum, guide, shift, r1)], 1u);
            }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) counts[(first_class + i) * stride] = (double)count[i];
}

// the abundance an EM that ran `ctl[CTL_ITERS]` steps left behind: x0 after an even number of
// steps, x1 after an odd one (the host does not know the count when it queues this)
__global__ void __launch_bounds__(256)
em_result_kernel(const unsigned long long *__restrict__ ctl, const double *__restrict__ x0,
                 const double *__restrict__ x1, int64_t n, double *__restrict__ out)
{
    const double *__restrict__ x = (ctl[CTL_ITERS] & 1ULL) ? x1 : x0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) out[i] = x[i];
}

__global__ void __launch_bounds__(256)
permute_f64_kernel(const double *__restrict__ x, const int32_t *__restrict__ perm, int64_t n,
                   double *__restrict__ y, bool scatter)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        if (scatter) y[perm[i]] = x[i]; else y[i] = x[perm[i]];
    }
}

__global__ void __launch_bounds__(256)
u64_to_double_kernel(const unsigned long long *__restrict__ in, int64_t n, double *__restrict__ out)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (double)in[i];
}

__global__ void __launch_bounds__(256)
double_to_cum_u64_kernel(const double *__restrict__ in, int64_t n, unsigned long long *__restrict__ out)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (unsigned long long)in[i];
}

// ASCII pooled contig bases -> 2-bit words (first base in the top bits)
__global__ void __launch_bounds__(256)
pack_sequences_kernel(const char *__restrict__ bases, int64_t n_bases, uint64_t *__restrict__ seq2,
                      int64_t n_words)
{
    for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < n_words;
         w += (int64_t)gridDim.x * blockDim.x) {
        uint64_t c = 0;
        const int64_t first = w * 32;
        for (int i = 0; i < 32 && first + i < n_bases; ++i)
            c |= (uint64_t)two_bit_encode((uint8_t)bases[first + i]) << (62 - 2 * i);
        seq2[w] = c;
    }
}

static inline unsigned grid_for(int64_t n, int64_t cap = 256 * 16)
{
    int64_t blocks = (n + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > cap) blocks = cap;
    return (unsigned)blocks;
}

static inline unsigned chip_grid(int64_t work_items, int items_per_block)
{
    int64_t blocks = (work_items + items_per_block - 1) / items_per_block;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 8) blocks = 256 * 8;
    return (unsigned)blocks;
}

// blocks of the launch that writes the stopping rule's partials = partials the judge reads:
// em_finalize's (several ranks) or em_rows_finalize's (p.fused)
int em_final_blocks(const EmProblem &p)
{
    int64_t blocks = p.fused ? (p.n_rows + 31) / 32 : (p.n_tx + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > EM_FINAL_BLOCKS) blocks = EM_FINAL_BLOCKS;
    return (int)blocks;
}

void launch_em_inner(const EmProblem &p, int parity, bool judge_previous, int64_t steps_done, hipStream_t stream)
{
    hipLaunchKernelGGL(em_inner_kernel, dim3(chip_grid(p.n_classes, 256)), dim3(256), 0, stream, p, parity,
                       judge_previous ? em_final_blocks(p) : 0, steps_done);
}

void launch_em_decide(const EmProblem &p, int64_t steps_done, hipStream_t stream)
{
    hipLaunchKernelGGL(em_decide_kernel, dim3(1), dim3(256), 0, stream, p, em_final_blocks(p), steps_done);
}

void launch_em_rows(const EmProblem &p, int parity, hipStream_t stream)
{
    hipLaunchKernelGGL(em_rows_kernel, dim3(chip_grid(p.n_rows, 32)), dim3(256), 0, stream, p, parity);
}

void launch_em_rows_finalize(const EmProblem &p, int parity, hipStream_t stream)
{
    hipLaunchKernelGGL(em_rows_finalize_kernel<false>, dim3((unsigned)em_final_blocks(p)), dim3(256), 0, stream, p, parity);
}

void launch_em_rows_acc(const EmProblem &p, int parity, hipStream_t stream)
{
    EmProblem fused = p;
    fused.fused = 1;                            // (the grid of the fused form)
    hipLaunchKernelGGL(em_rows_finalize_kernel<true>, dim3((unsigned)em_final_blocks(fused)), dim3(256), 0, stream, p, parity);
}

void launch_em_rows_to_acc(const EmProblem &p, hipStream_t stream)
{
    hipLaunchKernelGGL(em_rows_to_acc_kernel, dim3(chip_grid(p.n_tx, 256)), dim3(256), 0, stream, p);
}

void launch_em_finalize(const EmProblem &p, int parity, bool from_acc, hipStream_t stream)
{
    const int blocks = em_final_blocks(p);
    if (from_acc)
        hipLaunchKernelGGL(em_finalize_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, p, parity);
    else
        hipLaunchKernelGGL(em_finalize_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, p, parity);
}

// ---- numpy.sum of a contiguous f8 array, bit for bit ----------------------
// numpy adds the array up in blocks of 8192 elements (its reduction buffer),
// block sums accumulated left to right, and each block with pairwise_sum:
// below 8 elements a plain loop, up to 128 eight strided accumulators combined
// as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus the remainder in order, above
// that a split at n/2 rounded down to a multiple of 8 (numpy/_core/src/umath/
// loops_utils.h.src).  The start vector x /= x.sum() (seekmer/infer.py:118-119)
// and the TPM scaling (:127-129) go through it, so doing them on the device
// without it would move the EM input by an ulp.
// The tree is evaluated in three steps per block of 8192: lane 0 walks the
// recursion and lists its leaves (at most 128 elements each), one lane per leaf
// adds its leaf up exactly as numpy does, lane 0 walks the recursion again
// combining the leaf sums.  (The recursion is spelled as a chain of distinct
// functions, one per level, so that the call graph is static: 8192 elements
// split at most 7 times.)
constexpr int NP_MAX_LEAVES = 160;

__device__ __forceinline__ double np_leaf_sum(const double *a, int n)
{
    if (n < 8) {
        double r = 0.0;                      // numpy starts from -0.0 + a[0]; same value for our data
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

template <int LEVELS>
__device__ __attribute__((noinline)) void np_list_leaves(int lo, int n, int *leaf_lo, int *leaf_n, int *count)
{
    if (n <= 128) {
        if (*count < NP_MAX_LEAVES) { leaf_lo[*count] = lo; leaf_n[*count] = n; }
        ++*count;
        return;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    np_list_leaves<LEVELS - 1>(lo, n2, leaf_lo, leaf_n, count);
    np_list_leaves<LEVELS - 1>(lo + n2, n - n2, leaf_lo, leaf_n, count);
}
template <>
__device__ __attribute__((noinline)) void np_list_leaves<0>(int, int, int *, int *, int *count)
{
    *count = NP_MAX_LEAVES + 1;              // not reached for blocks of at most 8192 elements
}

template <int LEVELS>
__device__ __attribute__((noinline)) double np_combine(int n, const double *leaf_sum, int *next)
{
    if (n <= 128) return leaf_sum[(*next)++];
    int n2 = n / 2;
    n2 -= n2 % 8;
    const double left = np_combine<LEVELS - 1>(n2, leaf_sum, next);
    const double right = np_combine<LEVELS - 1>(n - n2, leaf_sum, next);
    return left + right;
}
template <>
__device__ __attribute__((noinline)) double np_combine<0>(int, const double *, int *) { return 0.0; }

// A full block of 8192 elements splits evenly all the way down: 64 leaves of 128 elements,
// combined as a balanced binary tree.  Eight lanes per leaf run numpy's eight strided
// accumulators (lane j adds elements j, j + 8, ... of the leaf in order), the combination
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) is a three-step butterfly over those lanes and the 64 leaf
// sums meet in a six-step butterfly -- the same additions in the same association (a + b = b + a
// exactly, so only the shape of the tree matters), with none of the recursion of the general path.
__device__ __forceinline__ void np_sum_full_block(const double *__restrict__ a, double *__restrict__ out)
{
    __shared__ double leaf[64];
    const int l = threadIdx.x >> 3, j = threadIdx.x & 7;          // 512 lanes: leaf, accumulator
    const double *p = a + l * 128 + j;
    double r = p[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) r += p[8 * i];
    r += __shfl_xor(r, 1, 8);
    r += __shfl_xor(r, 2, 8);
    r += __shfl_xor(r, 4, 8);
    if (j == 0) leaf[l] = r;
    __syncthreads();
    if (threadIdx.x < 64) {
        double s = leaf[threadIdx.x];
        for (int d = 1; d < 64; d <<= 1) s += __shfl_xor(s, d, 64);
        if (threadIdx.x == 0) *out = s;
    }
}

__global__ void __launch_bounds__(512)
np_sum_blocks_kernel(const double *__restrict__ a, int64_t n, double *__restrict__ block_sums, int64_t stride)
{
    // (blockIdx.y: one of several vectors `stride` elements apart, its block sums after the others')
    a += blockIdx.y * stride;
    block_sums += blockIdx.y * (int64_t)gridDim.x;
    const int64_t first = blockIdx.x * (int64_t)8192;
    const int len = (int)min((int64_t)8192, n - first);
    if (len == 8192) {                      // (block-uniform)
        np_sum_full_block(a + first, block_sums + blockIdx.x);
        return;
    }
    __shared__ double staged[8192];          // the block, fetched coalesced; leaves then read LDS
    __shared__ int leaf_lo[NP_MAX_LEAVES], leaf_n[NP_MAX_LEAVES], n_leaves;
    __shared__ double leaf_sum[NP_MAX_LEAVES];
    for (int i = threadIdx.x; i < len; i += blockDim.x) staged[i] = a[first + i];
    if (threadIdx.x == 0) {
        int count = 0;
        np_list_leaves<8>(0, len, leaf_lo, leaf_n, &count);
        n_leaves = count;
    }
    __syncthreads();
    if (n_leaves > NP_MAX_LEAVES) {          // cannot happen; keep the result defined
        if (threadIdx.x == 0) {
            double r = 0.0;
            for (int i = 0; i < len; ++i) r += staged[i];
            block_sums[blockIdx.x] = r;
        }
        return;
    }
    if ((int)threadIdx.x < n_leaves)
        leaf_sum[threadIdx.x] = np_leaf_sum(staged + leaf_lo[threadIdx.x], leaf_n[threadIdx.x]);
    __syncthreads();
    if (threadIdx.x == 0) {
        int next = 0;
        block_sums[blockIdx.x] = np_combine<8>(len, leaf_sum, &next);
    }
}

// out[0] = the sum, out[1] = sum / divisor (vector blockIdx.x of several: its own block sums and pair)
__global__ void np_sum_final_kernel(const double *block_sums, int64_t n_blocks, double divisor, double *out)
{
    if (threadIdx.x != 0) return;
    block_sums += blockIdx.x * n_blocks;
    out += blockIdx.x * 2;
    double acc = 0.0;
    for (int64_t b = 0; b < n_blocks; ++b) acc += block_sums[b];
    out[0] = acc;
    out[1] = acc / divisor;
}

__global__ void __launch_bounds__(256)
reciprocal_kernel(const double *__restrict__ l, int64_t n, double *__restrict__ x)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) x[i] = 1.0 / l[i];
}

// x /= *s, then (TPM) x[x < floor] = 0
__global__ void __launch_bounds__(256)
divide_kernel(double *__restrict__ x, int64_t n, const double *__restrict__ s, bool threshold, double floor,
              int64_t stride)
{
    x += blockIdx.y * stride;                 // (one of several vectors, each with its own divisor pair)
    const double d = s[blockIdx.y * 2];
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        double v = x[i] / d;
        if (threshold && v < floor) v = 0.0;
        x[i] = v;
    }
}

void launch_np_sum(const double *a, int64_t n, double divisor, double *block_sums, double *out,
                   hipStream_t stream)
{
    const int64_t n_blocks = (n + 8191) / 8192;
    if (n_blocks)
        hipLaunchKernelGGL(np_sum_blocks_kernel, dim3((unsigned)n_blocks), dim3(512), 0, stream,
                           a, n, block_sums, (int64_t)0);
    hipLaunchKernelGGL(np_sum_final_kernel, dim3(1), dim3(1), 0, stream, block_sums, n_blocks, divisor, out);
}

// the same for `count` vectors of n elements, `stride` apart: block_sums holds count * ceil(n / 8192)
// doubles, out[2 v] = sum of vector v, out[2 v + 1] = sum / divisor
void launch_np_sum_many(const double *a, int64_t n, int64_t count, int64_t stride, double divisor,
                        double *block_sums, double *out, hipStream_t stream)
{
    const int64_t n_blocks = (n + 8191) / 8192;
    for (int64_t v0 = 0; v0 < count; v0 += 32768) {              // (gridDim.y is limited to 65535)
        const unsigned here = (unsigned)std::min<int64_t>(32768, count - v0);
        if (n_blocks)
            hipLaunchKernelGGL(np_sum_blocks_kernel, dim3((unsigned)n_blocks, here), dim3(512), 0, stream,
                               a + v0 * stride, n, block_sums + v0 * n_blocks, stride);
        hipLaunchKernelGGL(np_sum_final_kernel, dim3(here), dim3(1), 0, stream, block_sums + v0 * n_blocks, n_blocks,
                           divisor, out + 2 * v0);
    }
}

void launch_reciprocal(const double *l, int64_t n, double *x, hipStream_t stream)
{
    hipLaunchKernelGGL(reciprocal_kernel, dim3(grid_for(n)), dim3(256), 0, stream, l, n, x);
}

void launch_divide(double *x, int64_t n, const double *s, bool threshold, double floor, hipStream_t stream)
{
    hipLaunchKernelGGL(divide_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, n, s, threshold, floor, (int64_t)0);
}

// vector v of `count` (n elements, `stride` apart) divided by s[2 v]
void launch_divide_many(double *x, int64_t n, int64_t count, int64_t stride, const double *s, bool threshold,
                        double floor, hipStream_t stream)
{
    for (int64_t v0 = 0; v0 < count; v0 += 32768) {
        const unsigned here = (unsigned)std::min<int64_t>(32768, count - v0);
        hipLaunchKernelGGL(divide_kernel, dim3(std::min<unsigned>(grid_for(n), 64u), here), dim3(256), 0, stream,
                           x + v0 * stride, n, s + 2 * v0, threshold, floor, stride);
    }
}

void launch_effective_lengths(const unsigned long long *fld, const double *lengths, int64_t n_tx,
                              double *out, hipStream_t stream)
{
    hipLaunchKernelGGL(effective_lengths_kernel, dim3(grid_for(n_tx)), dim3(256), 0, stream, fld,
                       lengths, n_tx, out);
}

int multinomial_tile(int64_t n_classes)
{
    int tile = 1;
    while ((n_classes + tile - 1) / tile > MN_TILES) tile <<= 1;
    return tile;
}

// tile_total: MN_TILES unsigned ints of scratch; counts[c * stride] receives class c's draws as f8.
// false when the table is too large for the tiled draw (more than MN_TILES^2 classes)
bool launch_multinomial(const unsigned long long *cum, int64_t n_classes, int64_t n_draws,
                        uint64_t seed, uint64_t stream_id, unsigned int *tile_total, double *counts,
                        int stride, hipStream_t stream)
{
    if (n_classes <= 0) return true;
    const int tile = multinomial_tile(n_classes);
    if (tile > MN_TILES || n_draws >= (1LL << 32)) return false;
    const int n_tiles = (int)((n_classes + tile - 1) / tile);
    (void)hipMemsetAsync(tile_total, 0, MN_TILES * sizeof(unsigned int), stream);
    if (n_draws > 0)
        hipLaunchKernelGGL(multinomial_tiles_kernel, dim3(512), dim3(1024), 0, stream, cum, n_classes, tile,
                           n_tiles, n_draws, seed, stream_id, tile_total);
    hipLaunchKernelGGL(multinomial_classes_kernel, dim3((unsigned)n_tiles), dim3(1024), 0, stream, cum, n_classes,
                       tile, tile_total, seed, stream_id, counts, stride);
    return true;
}

void launch_em_result(const unsigned long long *ctl, const double *x0, const double *x1, int64_t n, double *out,
                      hipStream_t stream)
{
    hipLaunchKernelGGL(em_result_kernel, dim3(grid_for(n)), dim3(256), 0, stream, ctl, x0, x1, n, out);
}

void launch_permute_f64(const double *x, const int32_t *perm, int64_t n, double *y, bool scatter,
                        hipStream_t stream)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(permute_f64_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, perm, n, y, scatter);
}

void launch_u64_to_double(const unsigned long long *in, int64_t n, double *out, hipStream_t stream)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(u64_to_double_kernel, dim3(grid_for(n)), dim3(256), 0, stream, in, n, out);
}

void launch_double_to_u64(const double *in, int64_t n, unsigned long long *out, hipStream_t stream)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(double_to_cum_u64_kernel, dim3(grid_for(n)), dim3(256), 0, stream, in, n, out);
}

void launch_pack_sequences(const char *bases, int64_t n_bases, uint64_t *seq2, int64_t n_words,
                           hipStream_t stream)
{
    hipLaunchKernelGGL(pack_sequences_kernel, dim3(grid_for(n_words)), dim3(256), 0, stream, bases,
                       n_bases, seq2, n_words);
}

void warm_code_em()
{
    hipFuncAttributes attributes;
    (void)hipFuncGetAttributes(&attributes, reinterpret_cast<const void *>(&em_inner_kernel));
}

}  // namespace skm

// ------------------------------------------------------------ diagnostics
// Random 16-byte gathers over a table of `n_slots` slots: the ceiling the
// index probes of the mapper can be priced against.  `chain` = 1:
// every lane's next address depends on the slot it just read (latency under
// load); `chain` = 0: `per_lane` independent gathers (throughput).
namespace skm {
__global__ void __launch_bounds__(256)
gather_probe_kernel(const uint4 *__restrict__ table, uint64_t slot_mask, int per_lane, int chain,
                    unsigned long long *sink)
{
    uint64_t x = mix64(blockIdx.x * (uint64_t)blockDim.x + threadIdx.x + 1);
    unsigned long long acc = 0;
    if (chain) {
        for (int i = 0; i < per_lane; ++i) {
            const uint4 v = table[x & slot_mask];
            acc += v.x;
            x = mix64(x + v.y + i);
        }
    } else {
        for (int i = 0; i < per_lane; i += 4) {
            const uint64_t a = mix64(x + i), b2 = mix64(x + i + 1), c = mix64(x + i + 2), d = mix64(x + i + 3);
            const uint4 v0 = table[a & slot_mask], v1 = table[b2 & slot_mask], v2 = table[c & slot_mask],
                        v3 = table[d & slot_mask];
            acc += v0.x + v1.x + v2.x + v3.x;
        }
    }
    if (acc == 0x123456789ULL) *sink = acc;
}

void launch_gather_probe(const void *table, uint64_t n_slots, int blocks, int per_lane, int chain,
                         unsigned long long *sink, hipStream_t stream)
{
    hipLaunchKernelGGL(gather_probe_kernel, dim3(blocks), dim3(256), 0, stream, (const uint4 *)table,
                       n_slots - 1, per_lane, chain, sink);
}
}  // namespace skm
