// GPU-resident equivalence-class counter: the device replacement of
// MapResult.counter (collections.Counter keyed by the id tuple,
// /root/reference/seekmer/mapper.py:54, 60-70) and of the class enumeration
// in MapResult.summarize (mapper.py:85-92).
//
// Open-addressing table keyed by a 64-bit tag of the tuple.  A batch is
// counted in launches, so that no lane ever spins on another lane and nothing
// one lane stores is read by another in the same launch:
//   insert  -- claim or find the slot by tag (atomicCAS), count++ and
//              first_seen = min(global unit index) (integer atomics, combined per
//              block in LDS first); the lane whose CAS created the class is its
//              creator and commits it on the spot: the block's creators take
//              their registry entries and arena space with ONE atomicAdd per
//              counter and block iteration, store their tuples and the slots'
//              tuple words.  A record that finds a class committed BEFORE this
//              launch compares its full tuple with the stored one at once
//   verify  -- the records that met a class created in the same launch compare
//              their FULL tuple with the stored one; a mismatch is a 64-bit tag
//              collision and raises SKM_ERR_COLLISION, so counts are exact or
//              the call fails -- never silently merged
// The launch boundary orders the tuple stores before the compares; within a
// launch only device-scope atomics touch shared words (a tuple word that a
// finder happens to see already set by a creator of the same launch is treated
// as not yet there: its arena offset is not below the launch's starting cursor).
#include "skm_kernels.h"
#include "../../include/seekmer_hip.h"

namespace skm {

__device__ __forceinline__ uint32_t unsigned_id(int32_t e) { return (uint32_t)(e < 0 ? ~e : e); }

__global__ void __launch_bounds__(256)
class_init_kernel(ClassSlot *slots, uint64_t n_slots)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n_slots;
         i += (uint64_t)gridDim.x * blockDim.x) {
        slots[i].key = 0;
        slots[i].count = 0;
        slots[i].first_seen = ~0ULL;
        slots[i].tuple = -1;
    }
}

// find-or-claim the slot whose tag is `key`; returns the slot index, or ~0
// when `limit` probes did not reach the key or a free slot (table too full:
// the host grows it and retries the deferred units)
// `seen` returns the slot's first_seen as read on the way (possibly stale, i.e. too
// large: it only ever decreases).  The probe reads are plain loads: a key never
// changes once set, and a stale "empty" only costs the CAS that then reports the
// real key -- device-scope atomics are served at the memory side of the fabric
// (the XCDs' L2s are not coherent), so every one avoided is a fabric request less.
__device__ __forceinline__ uint64_t probe_claim(const ClassTable &t, unsigned long long key,
                                                bool &claimed, uint64_t limit,
                                                unsigned long long *seen = nullptr)
{
    uint64_t slot = key & t.slot_mask;
    claimed = false;
    if (limit > t.slot_mask + 1) limit = t.slot_mask + 1;
    for (uint64_t n = 0; n < limit; ++n) {
        const ulonglong2 head = *reinterpret_cast<const ulonglong2 *>(&t.slots[slot]);   // key, first_seen
        unsigned long long cur = head.x;
        if (seen) *seen = head.y;
        if (cur == 0) {
            cur = atomicCAS(&t.slots[slot].key, 0ULL, key);
            if (cur == 0) { claimed = true; return slot; }
        }
        if (cur == key) return slot;
        slot = (slot + 1) & t.slot_mask;
    }
    return ~0ULL;
}

// probe_claim continued from a slot that has been read already (all 32 bytes: key, first_seen,
// count, tuple); `stored` returns the tuple word of the slot the key was found in -- >= 0 once the
// class's tuple is in the arena, which it is for every class of an earlier batch
__device__ __forceinline__ uint64_t probe_claim_from(const ClassTable &t, unsigned long long key,
                                                     ulonglong2 head, long long tuple_word, bool &claimed,
                                                     uint64_t limit, unsigned long long &seen, long long &stored)
{
    uint64_t slot = key & t.slot_mask;
    claimed = false;
    if (limit > t.slot_mask + 1) limit = t.slot_mask + 1;
    for (uint64_t n = 0; n < limit; ++n) {
        if (n) {
            head = *reinterpret_cast<const ulonglong2 *>(&t.slots[slot]);
            tuple_word = t.slots[slot].tuple;
        }
        unsigned long long cur = head.x;
        seen = head.y;
        stored = tuple_word;
        if (cur == 0) {
            cur = atomicCAS(&t.slots[slot].key, 0ULL, key);
            stored = -1;
            if (cur == 0) { claimed = true; return slot; }
        }
        if (cur == key) return slot;
        slot = (slot + 1) & t.slot_mask;
    }
    return ~0ULL;
}

// Records are taken INSERT_WIDTH at a time per lane: everything a record's chain needs from the
// batch (key, unit, tuple word), then the home slots of all of them, are fetched before any is
// looked at -- the chain key -> slot -> atomic of one record runs under the chains of the others.
// A record that lands on a class whose tuple is already in the arena (every class of an earlier
// batch) is verified on the spot: the slot's tuple word came with the probe (same 32-byte
// sector), so class_verify's second random pass over the table is only left with the records that
// met a class created in this very launch (unit_slot >= 0; the others leave -1).
#ifndef SKM_INSERT_WIDTH
#define SKM_INSERT_WIDTH 4
#endif
constexpr int INSERT_WIDTH = SKM_INSERT_WIDTH;
// Timing experiments only (scripts/build_variant.sh; the results of these builds are wrong):
// 1 = no count / first-seen atomics, 2 = no probe (every record "found" in its home slot, no CAS),
// 3 = neither: what is left is the streaming part of the kernel.
#ifndef SKM_CLASS_EXPERIMENT
#define SKM_CLASS_EXPERIMENT 0
#endif
#ifndef SKM_CLASS_AGG_BITS
#define SKM_CLASS_AGG_BITS 11       // entries of the block's combining table (12 bytes of LDS each)
#endif
#ifndef SKM_CLASS_INSERT_BLOCKS
#define SKM_CLASS_INSERT_BLOCKS 2048
#endif

__global__ void __launch_bounds__(256)
class_insert_kernel(ClassTable t, MapBatch b, int64_t unit_base, int64_t *unit_slot, bool retry_deferred)
{
    // tuples stored below this arena offset were committed by an earlier launch (class_totals_kernel
    // publishes the cursor between launches): only those are compared on the spot
    const long long committed = (long long)*t.arena_committed;
    __shared__ unsigned int s_new_classes[4], s_new_ids[4];
    __shared__ unsigned long long s_base[2];
    // Unaligned units are tallied per block (LDS) and reach the device counter with one atomic
    // per block: atomics on one address serialise at ~10 ns apiece.
    __shared__ unsigned int s_unaligned;
    // The count / first-seen updates of a block are combined in LDS before they go to the table: class
    // sizes are skewed (the ten largest classes of a 10 M-pair batch hold 15 400 ... 4 100 units), a
    // slot that is read by a probe and then hit by a device-scope atomic thousands of times is the
    // launch's long pole (scripts/micro/atomic_layout.hip: 10 M read + add pairs take 0.45 ms spread
    // evenly and 0.70 ms with 77 k of them on ten addresses), and a block meets a large class several
    // times.  AGG entries, claimed once and never evicted: a record whose entry is taken by another
    // slot updates the table directly.
    constexpr int AGG = 1 << SKM_CLASS_AGG_BITS;
    __shared__ unsigned int agg_slot[AGG], agg_count[AGG], agg_min[AGG];
    for (int i = threadIdx.x; i < AGG; i += blockDim.x) { agg_slot[i] = 0; agg_count[i] = 0; agg_min[i] = 0xffffffffu; }
    if (threadIdx.x == 0) s_unaligned = 0;
    __syncthreads();
    unsigned int unaligned = 0;
    bool all_same = true;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // (u walks the batch's RECORDS, skm_kernels.h: MapBatch; unit_slot is by record)
    // (every lane of a block runs the same number of iterations: the commit of an iteration is a
    // block-wide step)
    for (int64_t ub = blockIdx.x * (int64_t)blockDim.x; ub < b.n_units; ub += INSERT_WIDTH * stride) {
        const int64_t u0 = ub + threadIdx.x;
        unsigned long long key[INSERT_WIDTH], mine_at[INSERT_WIDTH];
        int64_t made_slot[INSERT_WIDTH];              // slot of a class this lane created in this iteration, or -1
#pragma unroll
        for (int k = 0; k < INSERT_WIDTH; ++k) made_slot[k] = -1;
        ulonglong2 head[INSERT_WIDTH];
        long long tuple_word[INSERT_WIDTH];
        int32_t unit_of[INSERT_WIDTH];
        bool live[INSERT_WIDTH];
#pragma unroll
        for (int k = 0; k < INSERT_WIDTH; ++k) {
            const int64_t u = u0 + k * stride;
            live[k] = u < b.n_units && !(retry_deferred && unit_slot[u] != -2);
            key[k] = live[k] ? b.rec_key[u] : 0;
            unit_of[k] = live[k] ? b.rec_unit[u] : 0;
            mine_at[k] = live[k] ? b.rec_tuple[u] : 0;
        }
#pragma unroll
        for (int k = 0; k < INSERT_WIDTH; ++k) {
            const ClassSlot *home = &t.slots[key[k] & t.slot_mask];          // (key 0: slot 0, unused)
            if (SKM_CLASS_EXPERIMENT & 2) { head[k] = ulonglong2{key[k], 0ULL}; tuple_word[k] = -1; continue; }
            head[k] = *reinterpret_cast<const ulonglong2 *>(home);
            tuple_word[k] = home->tuple;
        }
#pragma unroll
        for (int k = 0; k < INSERT_WIDTH; ++k) {
            const int64_t u = u0 + k * stride;
            if (u >= b.n_units) continue;
            // every record leaves with ONE store to unit_slot, issued at the end: on gfx9 a store issued
            // before the loads would have to drain in front of them
            int64_t where = -1;
            if (live[k]) {
                if (key[k] == 0) {                // empty tuple = unaligned, mapper.py:87
                    ++unaligned;
                } else {
                    bool claimed;
                    unsigned long long seen = ~0ULL;
                    long long stored = -1;
                    const uint64_t slot = probe_claim_from(t, key[k], head[k], tuple_word[k], claimed, CLASS_PROBE_LIMIT,
                                                           seen, stored);
                    if (slot == ~0ULL) {          // deferred: counted after the table has grown
                        atomicAdd(t.n_deferred, 1ULL);
                        where = -2;
                    } else {
                        if (claimed) made_slot[k] = (int64_t)slot;      // committed below, with the block's others
                        const unsigned long long unit = (unsigned long long)(unit_base + unit_of[k]);
                        if (!(SKM_CLASS_EXPERIMENT & 1)) {
                            const bool lower = claimed || seen > unit;
                            const unsigned int tag = (unsigned int)slot + 1u;
                            const unsigned int h = (tag * 0x9E3779B1u) >> (32 - SKM_CLASS_AGG_BITS);
                            unsigned int holder = (slot >> 31) ? ~0u : agg_slot[h];
                            if (holder == 0) holder = atomicCAS(&agg_slot[h], 0u, tag);
                            if (holder == 0 || holder == tag) {
                                atomicAdd(&agg_count[h], 1u);
                                if (lower) atomicMin(&agg_min[h], (unsigned int)unit_of[k]);
                            } else {
                                atomicAdd(&t.slots[slot].count, 1ULL);
                                if (lower) atomicMin(&t.slots[slot].first_seen, unit);
                            }
                        }
                        where = claimed ? -1 : (int64_t)slot;
                        if (stored >= 0 && tuple_offset(stored) < committed) {   // in the arena since an earlier launch: compare now
                            const int n = (int)(mine_at[k] >> 40);
                            bool same = tuple_len(stored) == n;
                            if (same) {
                                const int32_t *mine = b.unit_entries + (mine_at[k] & ((1ULL << 40) - 1));
                                const int32_t *ref = t.arena + tuple_offset(stored);
                                for (int i = 0; same && i < n; ++i) same = (uint32_t)ref[i] == unsigned_id(mine[i]);
                            }
                            all_same &= same;
                            where = -1;           // (nothing left for class_verify)
                        }
                    }
                }
                unit_slot[u] = where;
            }
        }
        // ---- the block's new classes of this iteration: registry entries and arena space with one
        // atomicAdd per counter, then every creator stores its tuple
        unsigned int my_classes = 0, my_ids = 0;
#pragma unroll
        for (int k = 0; k < INSERT_WIDTH; ++k)
            if (made_slot[k] >= 0) { ++my_classes; my_ids += (unsigned int)(mine_at[k] >> 40); }
        unsigned int before_classes = my_classes, before_ids = my_ids;       // inclusive scan over the wave
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned int c = __shfl_up(before_classes, d, 64), i = __shfl_up(before_ids, d, 64);
            if (lane >= d) { before_classes += c; before_ids += i; }
        }
        if (lane == 63) { s_new_classes[wave] = before_classes; s_new_ids[wave] = before_ids; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned int classes = s_new_classes[0] + s_new_classes[1] + s_new_classes[2] + s_new_classes[3];
            const unsigned int ids = s_new_ids[0] + s_new_ids[1] + s_new_ids[2] + s_new_ids[3];
            s_base[0] = s_base[1] = 0;
            if (classes) {
                s_base[0] = atomicAdd(t.n_listed, (unsigned long long)classes);
                s_base[1] = atomicAdd(t.arena_cursor, (unsigned long long)ids);
                atomicAdd(t.n_classes, (unsigned long long)classes);
            }
        }
        __syncthreads();
        if (my_classes) {
            long long registry = (long long)s_base[0] + (before_classes - my_classes);
            long long at = (long long)s_base[1] + (before_ids - my_ids);
            for (int w = 0; w < wave; ++w) { registry += s_new_classes[w]; at += s_new_ids[w]; }
#pragma unroll
            for (int k = 0; k < INSERT_WIDTH; ++k) {
                if (made_slot[k] < 0) continue;
                const int n = (int)(mine_at[k] >> 40);
                if (at + n > t.arena_capacity || registry >= t.class_list_capacity) {
                    atomicExch(t.error, SKM_ERR_STATE);
                } else {
                    const int32_t *entries = b.unit_entries + (mine_at[k] & ((1ULL << 40) - 1));
                    for (int i = 0; i < n; ++i) t.arena[at + i] = (int32_t)unsigned_id(entries[i]);
                    t.slots[made_slot[k]].tuple = tuple_pack(at, n);
                    t.class_list[registry] = made_slot[k];
                }
                at += n;
                ++registry;
            }
        }
        __syncthreads();                              // (s_new_* are rewritten by the next iteration)
    }
    if (!all_same) atomicExch(t.error, SKM_ERR_COLLISION);
    if (unaligned) atomicAdd(&s_unaligned, unaligned);
    __syncthreads();
    for (int i = threadIdx.x; i < AGG; i += blockDim.x) {        // what the block combined
        const unsigned int tag = agg_slot[i];
        if (tag == 0) continue;
        ClassSlot *slot = &t.slots[tag - 1u];
        atomicAdd(&slot->count, (unsigned long long)agg_count[i]);
        if (agg_min[i] != 0xffffffffu) atomicMin(&slot->first_seen, (unsigned long long)(unit_base + (int64_t)agg_min[i]));
    }
    if (threadIdx.x == 0 && s_unaligned) atomicAdd(t.n_unaligned, (unsigned long long)s_unaligned);
    if (!retry_deferred && blockIdx.x == 0 && threadIdx.x == 0)
        atomicAdd(t.n_units, (unsigned long long)b.n_units);
}

// between the launches of a batch: what has been committed so far is published for the next
// launch's on-the-spot compares (and, once per batch, the batch histogram is merged:
// merge_fragment_lengths, mapper.py:106-115)
__global__ void __launch_bounds__(256)
class_totals_kernel(ClassTable t, const unsigned long long *batch_fld)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) *t.arena_committed = *t.arena_cursor;
    if (batch_fld)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < MAX_FRAGMENT_LENGTH;
             i += gridDim.x * blockDim.x)
            t.global_fld[i] += batch_fld[i];
}

__global__ void __launch_bounds__(256)
class_verify_kernel(ClassTable t, MapBatch b, const int64_t *unit_slot)
{
    // (four records per lane at a time: their slots' tuple words, then the first ids of both
    // sides, are in flight together)
    constexpr int WIDTH = 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    bool all_same = true;
    for (int64_t u0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; u0 < b.n_units; u0 += WIDTH * stride) {
        int64_t slot[WIDTH];
        unsigned long long mine_at[WIDTH];
        long long stored[WIDTH];
#pragma unroll
        for (int k = 0; k < WIDTH; ++k) {
            const int64_t u = u0 + k * stride;
            slot[k] = u < b.n_units ? unit_slot[u] : -1;
            mine_at[k] = u < b.n_units ? b.rec_tuple[u] : 0;
        }
#pragma unroll
        for (int k = 0; k < WIDTH; ++k) stored[k] = slot[k] >= 0 ? t.slots[slot[k]].tuple : 0;
#pragma unroll
        for (int k = 0; k < WIDTH; ++k) {
            if (slot[k] < 0) continue;
            const int n = (int)(mine_at[k] >> 40);
            bool same = stored[k] >= 0 && tuple_len(stored[k]) == n;
            if (same) {
                const int32_t *mine = b.unit_entries + (mine_at[k] & ((1ULL << 40) - 1));
                const int32_t *ref = t.arena + tuple_offset(stored[k]);
                for (int i = 0; same && i < n; ++i) same = (uint32_t)ref[i] == unsigned_id(mine[i]);
            }
            all_same &= same;
        }
    }
    if (!all_same) atomicExch(t.error, SKM_ERR_COLLISION);
}

// move every occupied slot of `from` into the (larger, initialised) table `to`
// and record where it went, so that the registry and the unit -> slot map of a
// batch in flight can be redirected
__global__ void __launch_bounds__(256)
class_rehash_kernel(ClassTable from, ClassTable to, int64_t *forward)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i <= from.slot_mask;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const ClassSlot s = from.slots[i];
        if (s.key == 0) { forward[i] = -1; continue; }
        bool claimed;
        const uint64_t slot = probe_claim(to, s.key, claimed, ~0ULL);
        if (slot == ~0ULL || !claimed) { atomicExch(to.error, SKM_ERR_STATE); forward[i] = -1; continue; }
        to.slots[slot].count = s.count;
        to.slots[slot].first_seen = s.first_seen;
        to.slots[slot].tuple = s.tuple;
        forward[i] = (int64_t)slot;
    }
}

__global__ void __launch_bounds__(256)
slot_remap_kernel(int64_t *slots, int64_t n, const int64_t *forward)
{
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n;
         k += (int64_t)gridDim.x * blockDim.x)
        if (slots[k] >= 0) slots[k] = forward[slots[k]];
}

// dense dump of the classes (registry order): arena offset, tuple length,
// count as f8, first-seen unit
__global__ void __launch_bounds__(256)
class_compact_kernel(ClassTable t, int64_t n_classes, int64_t *cls_offset, int64_t *cls_len,
                     double *cls_count, unsigned long long *cls_first_seen)
{
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n_classes;
         k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t.class_list[k];
        const ClassSlot s = t.slots[i];
        cls_offset[k] = s.tuple < 0 ? -1 : tuple_offset(s.tuple);
        cls_len[k] = s.tuple < 0 ? 0 : tuple_len(s.tuple);
        cls_count[k] = (double)s.count;
        if (cls_first_seen) cls_first_seen[k] = s.first_seen;
    }
}

// Counter.update with a foreign table (another GPU's export).  The foreign classes come either as
// the arrays of skm_mapper_export (CSR offsets, int64 counts; class_len == nullptr) or as they lie in
// another mapper's HBM (skm_device_table: class c = ids[class_offsets[c] .. + class_len[c]), counts
// as doubles, registry order).
template <typename Count>
__global__ void __launch_bounds__(256)
class_merge_kernel(ClassTable t, int64_t n_classes, const int64_t *class_offsets, const int64_t *class_len,
                   const int32_t *class_targets, const Count *class_counts,
                   const int64_t *first_seen)
{
    for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < n_classes;
         c += (int64_t)gridDim.x * blockDim.x) {
        const int64_t off = class_offsets[c];
        const int n = (int)(class_len ? class_len[c] : class_offsets[c + 1] - off);
        unsigned long long key = 0x243F6A8885A308D3ULL ^ (unsigned long long)n;
        for (int i = 0; i < n; ++i) {
            key ^= (uint32_t)class_targets[off + i];
            key *= 0x9E3779B97F4A7C15ULL;
            key ^= key >> 32;
        }
        if (key == 0) key = 1;
        bool claimed;
        const uint64_t slot = probe_claim(t, key, claimed, ~0ULL);
        if (slot == ~0ULL) { atomicExch(t.error, SKM_ERR_STATE); continue; }
        // the wave's new classes take their registry entries and arena space with one atomic
        // per counter (same-address atomics serialise: three per class were 30 ms per million)
        const unsigned long long makers = __ballot(claimed);
        long long a = 0, k = 0;
        if (makers) {
            const int lane = (int)(threadIdx.x & 63);
            int before = 0, total = 0;
            for (unsigned long long m = makers; m; m &= m - 1) {
                const int src = __builtin_ctzll(m);
                const int v = __builtin_amdgcn_readlane(n, src);
                if (src < lane) before += v;
                total += v;
            }
            const int leader = __builtin_ctzll(makers);
            unsigned long long base_a = 0, base_k = 0;
            if (lane == leader) {
                base_a = atomicAdd(t.arena_cursor, (unsigned long long)total);
                base_k = atomicAdd(t.n_listed, (unsigned long long)__popcll(makers));
                atomicAdd(t.n_classes, (unsigned long long)__popcll(makers));
            }
            a = (long long)__shfl(base_a, leader, 64) + before;
            k = (long long)__shfl(base_k, leader, 64) + __popcll(makers & ((1ULL << lane) - 1));
        }
        if (claimed) {
            if (a + n > t.arena_capacity || k >= t.class_list_capacity) {
                atomicExch(t.error, SKM_ERR_STATE);
                continue;
            }
            for (int i = 0; i < n; ++i) t.arena[a + i] = class_targets[off + i];
            t.slots[slot].tuple = tuple_pack(a, n);
            t.class_list[k] = (int64_t)slot;
        } else {
            // classes of the resident table are all committed between batches
            const long long stored = t.slots[slot].tuple;
            bool same = stored >= 0 && tuple_len(stored) == n;
            const long long a = tuple_offset(stored);
            for (int i = 0; same && i < n; ++i) same = t.arena[a + i] == class_targets[off + i];
            if (!same) { atomicExch(t.error, SKM_ERR_COLLISION); continue; }
        }
        atomicAdd(&t.slots[slot].count, (unsigned long long)class_counts[c]);
        atomicMin(&t.slots[slot].first_seen, (unsigned long long)first_seen[c]);
    }
}

static inline unsigned grid_for(int64_t n, int64_t cap = 256 * 16)
{
    int64_t blocks = (n + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > cap) blocks = cap;
    return (unsigned)blocks;
}

void launch_class_init(const ClassTable &t, hipStream_t stream)
{
    hipLaunchKernelGGL(class_init_kernel, dim3(grid_for((int64_t)t.slot_mask + 1)), dim3(256), 0,
                       stream, t.slots, t.slot_mask + 1);
}

void launch_class_insert(const ClassTable &t, const MapBatch &b, int64_t unit_base,
                         int64_t *unit_slot, bool retry_deferred, bool merge_fld, hipStream_t stream)
{
    if (b.n_units == 0) return;
    // (one same-address atomic per block for the unaligned tally; a block combines what it meets)
    hipLaunchKernelGGL(class_insert_kernel, dim3(grid_for(b.n_units, SKM_CLASS_INSERT_BLOCKS)), dim3(256), 0, stream, t, b,
                       unit_base, unit_slot, retry_deferred);
    hipLaunchKernelGGL(class_totals_kernel, dim3(8), dim3(256), 0, stream, t, merge_fld ? b.fld : nullptr);
}

void launch_class_verify(const ClassTable &t, const MapBatch &b, const int64_t *unit_slot,
                         hipStream_t stream)
{
    if (b.n_units == 0) return;
    hipLaunchKernelGGL(class_verify_kernel, dim3(grid_for(b.n_units)), dim3(256), 0, stream, t, b,
                       unit_slot);
}

void launch_class_rehash(const ClassTable &from, const ClassTable &to, int64_t *forward,
                         hipStream_t stream)
{
    hipLaunchKernelGGL(class_rehash_kernel, dim3(grid_for((int64_t)from.slot_mask + 1)), dim3(256), 0,
                       stream, from, to, forward);
}

void launch_slot_remap(int64_t *slots, int64_t n, const int64_t *forward, hipStream_t stream)
{
    if (n == 0) return;
    hipLaunchKernelGGL(slot_remap_kernel, dim3(grid_for(n)), dim3(256), 0, stream, slots, n, forward);
}

void launch_class_compact(const ClassTable &t, int64_t n_classes, int64_t *cls_offset,
                          int64_t *cls_len, double *cls_count, unsigned long long *cls_first_seen,
                          hipStream_t stream)
{
    if (n_classes == 0) return;
    hipLaunchKernelGGL(class_compact_kernel, dim3(grid_for(n_classes)), dim3(256), 0, stream, t,
                       n_classes, cls_offset, cls_len, cls_count, cls_first_seen);
}

void launch_class_merge(const ClassTable &t, int64_t n_classes, const int64_t *class_offsets,
                        const int32_t *class_targets, const int64_t *class_counts,
                        const int64_t *first_seen, hipStream_t stream)
{
    if (n_classes == 0) return;
    hipLaunchKernelGGL(class_merge_kernel<int64_t>, dim3(grid_for(n_classes)), dim3(256), 0, stream, t,
                       n_classes, class_offsets, (const int64_t *)nullptr, class_targets, class_counts, first_seen);
    hipLaunchKernelGGL(class_totals_kernel, dim3(1), dim3(64), 0, stream, t, (const unsigned long long *)nullptr);
}

// the unit totals and the histogram of a foreign table, added on the device
__global__ void __launch_bounds__(256)
class_add_totals_kernel(ClassTable t, unsigned long long unaligned, unsigned long long units,
                        const unsigned long long *fld)
{
    if (threadIdx.x == 0) { *t.n_unaligned += unaligned; *t.n_units += units; }
    if (fld)
        for (int i = threadIdx.x; i < MAX_FRAGMENT_LENGTH; i += blockDim.x) t.global_fld[i] += fld[i];
}

void launch_class_merge_device(const ClassTable &t, int64_t n_classes, const int64_t *class_start,
                               const int64_t *class_len, const int32_t *ids, const double *class_counts,
                               const unsigned long long *first_seen, unsigned long long unaligned,
                               unsigned long long units, const unsigned long long *fld, hipStream_t stream)
{
    if (n_classes) {
        hipLaunchKernelGGL(class_merge_kernel<double>, dim3(grid_for(n_classes)), dim3(256), 0, stream, t,
                           n_classes, class_start, class_len, ids, class_counts, (const int64_t *)first_seen);
        hipLaunchKernelGGL(class_totals_kernel, dim3(1), dim3(64), 0, stream, t, (const unsigned long long *)nullptr);
    }
    hipLaunchKernelGGL(class_add_totals_kernel, dim3(1), dim3(256), 0, stream, t, unaligned, units, fld);
}

void warm_code_classes()
{
    hipFuncAttributes attributes;
    (void)hipFuncGetAttributes(&attributes, reinterpret_cast<const void *>(&class_insert_kernel));
}

}  // namespace skm
