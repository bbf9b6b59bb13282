// Micro-benchmark behind DESIGN.md's note on the class counter: scattered atomic adds on u32
// counters, (a) device scope on one array (what class_insert does: served at the memory side of
// the fabric), (b) workgroup scope on an array PRIVATE to the issuing XCD (served in that XCD's
// L2; sound only because no other XCD touches the array before the kernel ends).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_scope scripts/micro/atomic_scope.hip && /tmp/atomic_scope
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

__device__ __forceinline__ uint32_t xcc_id()
{
    // HW_REG_XCC_ID = 20, bits 3:0
    return __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)) & 0xf;
}

template <int MODE>
__global__ void __launch_bounds__(256) adds(uint32_t *counters, uint32_t n_slots, int64_t n_adds, uint32_t *xcc_seen)
{
    const uint32_t x = xcc_id();
    if (threadIdx.x == 0) atomicOr(&xcc_seen[0], 1u << x);
    uint32_t *mine = MODE == 1 ? counters + (size_t)x * n_slots : counters;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_adds; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t slot = mix((uint32_t)i) & (n_slots - 1);
        if (MODE == 0) atomicAdd(&mine[slot], 1u);
        else __hip_atomic_fetch_add(&mine[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

__global__ void __launch_bounds__(256) total(const uint32_t *counters, size_t n, unsigned long long *out)
{
    unsigned long long s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += counters[i];
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}

int main()
{
    const uint32_t n_slots = 1u << 21;
    const int64_t n_adds = 10000000;
    uint32_t *counters, *seen;
    unsigned long long *sum;
    CHECK(hipMalloc(&counters, (size_t)8 * n_slots * 4));
    CHECK(hipMalloc(&seen, 4));
    CHECK(hipMalloc(&sum, 8));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipMemset(counters, 0, (size_t)8 * n_slots * 4));
            CHECK(hipMemset(seen, 0, 4));
            CHECK(hipMemset(sum, 0, 8));
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(a));
            if (mode == 0) hipLaunchKernelGGL(adds<0>, dim3(2048), dim3(256), 0, 0, counters, n_slots, n_adds, seen);
            else hipLaunchKernelGGL(adds<1>, dim3(2048), dim3(256), 0, 0, counters, n_slots, n_adds, seen);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, a, b));
            hipLaunchKernelGGL(total, dim3(1024), dim3(256), 0, 0, counters, (size_t)8 * n_slots, sum);
            unsigned long long got = 0;
            uint32_t mask = 0;
            CHECK(hipMemcpy(&got, sum, 8, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(&mask, seen, 4, hipMemcpyDeviceToHost));
            printf("%s rep %d: %.3f ms, %.1f G adds/s, total %llu (%s), XCC ids seen 0x%x\n",
                   mode == 0 ? "device scope, one array      " : "workgroup scope, array per XCD", rep, ms,
                   n_adds / ms / 1e6, got, got == (unsigned long long)n_adds ? "exact" : "WRONG", mask);
        }
    }
    return 0;
}
