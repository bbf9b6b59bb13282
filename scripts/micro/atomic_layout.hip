// Micro-benchmark behind the class counter's layout (class counting): 10 M scattered
// device-scope atomic adds, as class_insert_kernel issues them, over 4 Mi counters laid out
//   (a) as the count word of 32-byte table slots (128 MiB: what the kernel did through round 3's start),
//   (b) as a compact u64 array (32 MiB), (c) as a compact u32 array (16 MiB);
// then the same with the probe's 16-byte read of the slot in front of every add: (d) add into the slot
// that was read, (e) add into the compact u64 array, (f) into the compact u32 array.
//   mkdir -p seekmer_amd/csrc/build_micro && hipcc --offload-arch=gfx950 -O3 \
//       -o seekmer_amd/csrc/build_micro/atomic_layout scripts/micro/atomic_layout.hip      (cross-compiles without a GPU)
//   seekmer_amd/csrc/build_micro/atomic_layout                                            (on a GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

struct alignas(32) Slot { unsigned long long key, first_seen, count; long long tuple; };

template <int MODE, int HOT>
__global__ void __launch_bounds__(256) adds(Slot *slots, unsigned long long *c64, unsigned int *c32, uint32_t mask,
                                            int64_t n_adds, unsigned long long *sink)
{
    unsigned long long seen = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i0 < n_adds; i0 += 4 * stride) {
        uint32_t slot[4];
        ulonglong2 head[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = (uint32_t)(i0 + k * stride);
            slot[k] = mix(i) & mask;
            // the class counter's skew: the ten largest classes of the 10 M-pair table hold 15 400, 8 000,
            // 7 200, 5 500, 4 600 and five times 4 100 units (one add in HOT goes to one of ten addresses)
            if (HOT && mix(i ^ 0x9e3779b9u) % (uint32_t)HOT == 0) slot[k] = (mix(i >> 3) % 10u) * 4099u;
        }
        if (MODE >= 3) {
#pragma unroll
            for (int k = 0; k < 4; ++k) head[k] = *reinterpret_cast<const ulonglong2 *>(&slots[slot[k]]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i0 + k * stride >= n_adds) continue;
            if (MODE >= 3) seen += head[k].x + head[k].y;
            if (MODE == 0 || MODE == 3) atomicAdd(&slots[slot[k]].count, 1ULL);
            else if (MODE == 1 || MODE == 4) atomicAdd(&c64[slot[k]], 1ULL);
            else atomicAdd(&c32[slot[k]], 1u);
        }
    }
    if (seen == 0x1234567) *sink = seen;
}

int main()
{
    const uint32_t n_slots = 1u << 22;
    const int64_t n_adds = 10000000;
    Slot *slots; unsigned long long *c64, *sink; unsigned int *c32;
    CHECK(hipMalloc(&slots, (size_t)n_slots * sizeof(Slot)));
    CHECK(hipMalloc(&c64, (size_t)n_slots * 8));
    CHECK(hipMalloc(&c32, (size_t)n_slots * 4));
    CHECK(hipMalloc(&sink, 8));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const char *names[9] = {"add -> slot.count (32-byte slots, 128 MiB)", "add -> compact u64 (32 MiB)", "add -> compact u32 (16 MiB)",
                            "read slot + add -> slot.count", "read slot + add -> compact u64", "read slot + add -> compact u32",
                            "read slot + add, 77 k adds on 10 addresses", "read slot + add, 7.7 k adds on 10 addresses",
                            "add -> slot.count, 77 k adds on 10 addresses"};
    for (int mode = 0; mode < 9; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipMemset(slots, 0, (size_t)n_slots * sizeof(Slot)));
            CHECK(hipMemset(c64, 0, (size_t)n_slots * 8));
            CHECK(hipMemset(c32, 0, (size_t)n_slots * 4));
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(a));
            switch (mode) {
            case 0: hipLaunchKernelGGL((adds<0, 0>), dim3(2048), dim3(256), 0, 0, slots, c64, c32, n_slots - 1, n_adds, sink); break;
            case 1: hipLaunchKernelGGL((adds<1, 0>), dim3(2048), dim3(256), 0, 0, slots, c64, c32, n_slots - 1, n_adds, sink); break;
            case 2: hipLaunchKernelGGL((adds<2, 0>), dim3(2048), dim3(256), 0, 0, slots, c64, c32, n_slots - 1, n_adds, sink); break;
            case 3: hipLaunchKernelGGL((adds<3, 0>), dim3(2048), dim3(256), 0, 0, slots, c64, c32, n_slots - 1, n_adds, sink); break;
            case 4: hipLaunchKernelGGL((adds<4, 0>), dim3(2048), dim3(256), 0, 0, slots, c64, c32, n_slots - 1, n_adds, sink); break;
            case 5: hipLaunchKernelGGL((adds<5, 0>), dim3(2048), dim3(256), 0, 0, slots, c64, c32, n_slots - 1, n_adds, sink); break;
            case 6: hipLaunchKernelGGL((adds<3, 130>), dim3(2048), dim3(256), 0, 0, slots, c64, c32, n_slots - 1, n_adds, sink); break;
            case 7: hipLaunchKernelGGL((adds<3, 1300>), dim3(2048), dim3(256), 0, 0, slots, c64, c32, n_slots - 1, n_adds, sink); break;
            default: hipLaunchKernelGGL((adds<0, 130>), dim3(2048), dim3(256), 0, 0, slots, c64, c32, n_slots - 1, n_adds, sink); break;
            }
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        printf("%-46s %.3f ms = %.1f G/s\n", names[mode], best, n_adds / best / 1e6);
    }
    return 0;
}
