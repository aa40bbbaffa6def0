// gather_rates.hip -- microbenchmark (not part of the product): what a wave's scattered table look-up costs on gfx950.
// Every lane reads table[random index] -- 2 / 4 / 8 bytes from a table of 32 KB ... 1 MB in global memory (L1 / L2 hits), or 4 / 8
// bytes from a table in LDS -- with enough independent look-ups in flight to hide the latency; reported: cycles of one CU per
// wave-level look-up instruction (at the clock the device reports).
//   hipcc --offload-arch=gfx950 -O3 gather_rates.hip -o build/gather_rates && build/gather_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)

constexpr int kIters = 256, kUnroll = 8;
__device__ __forceinline__ uint32_t next(uint32_t s) { return s * 1664525u + 1013904223u; }

template <class T, int kSpread> // kSpread 0: every lane its own random element; 1: all lanes of a wave in one 64-byte line
__global__ __launch_bounds__(256) void gather_global(const T* __restrict__ tab, uint32_t mask, uint32_t* out) {
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u, acc = 0;
    for (int it = 0; it < kIters; it++) {
        T v[kUnroll];
#pragma unroll
        for (int k = 0; k < kUnroll; k++) {
            s = next(s);
            uint32_t i = (s >> 8) & mask;
            if (kSpread == 1) i = (i & ~(64u / sizeof(T) - 1u)) & (uint32_t)__builtin_amdgcn_readfirstlane((int)i) | (threadIdx.x & (64u / sizeof(T) - 1u));
            v[k] = tab[i];
        }
#pragma unroll
        for (int k = 0; k < kUnroll; k++) {
            const unsigned char* b = reinterpret_cast<const unsigned char*>(&v[k]);
            for (int j = 0; j < (int)sizeof(T); j += 2) acc += b[j];
        }
    }
    if (acc == 0xfffffff0u) out[0] = acc;
}

template <class T>
__global__ __launch_bounds__(256) void gather_lds(const T* __restrict__ tab, uint32_t n, uint32_t* out) {
    extern __shared__ unsigned char s_raw[];
    T* s_tab = reinterpret_cast<T*>(s_raw);
    for (uint32_t i = threadIdx.x; i < n; i += 256) s_tab[i] = tab[i];
    __syncthreads();
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u, acc = 0;
    const uint32_t mask = n - 1;
    for (int it = 0; it < kIters; it++) {
        T v[kUnroll];
#pragma unroll
        for (int k = 0; k < kUnroll; k++) {
            s = next(s);
            v[k] = s_tab[(s >> 8) & mask];
        }
#pragma unroll
        for (int k = 0; k < kUnroll; k++) {
            const unsigned char* b = reinterpret_cast<const unsigned char*>(&v[k]);
            for (int j = 0; j < (int)sizeof(T); j += 2) acc += b[j];
        }
    }
    if (acc == 0xfffffff0u) out[0] = acc;
}

template <class F>
static double time_ms(F launch) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / 5.0;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const double ghz = prop.clockRate * 1e-6;
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * 8; // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    void* tab;
    uint32_t* out;
    CHECK(hipMalloc(&tab, 4 << 20));
    CHECK(hipMemset(tab, 1, 4 << 20));
    CHECK(hipMalloc(&out, 64));
    const double wave_instr_per_cu = (double)blocks * 4 * kIters * kUnroll / cus;
    printf("%d CUs at %.2f GHz; cycles of one CU per wave-level look-up (8 waves per SIMD, %d look-ups in flight per wave)\n", cus, ghz, kUnroll);
    for (uint32_t kb : {32u, 128u, 512u, 2048u}) {
        const uint32_t bytes = kb << 10;
        const double t2 = time_ms([&] { gather_global<uint16_t, 0><<<blocks, 256>>>((const uint16_t*)tab, bytes / 2 - 1, out); });
        const double t4 = time_ms([&] { gather_global<uint32_t, 0><<<blocks, 256>>>((const uint32_t*)tab, bytes / 4 - 1, out); });
        const double t8 = time_ms([&] { gather_global<uint2, 0><<<blocks, 256>>>((const uint2*)tab, bytes / 8 - 1, out); });
        const double l4 = time_ms([&] { gather_global<uint32_t, 1><<<blocks, 256>>>((const uint32_t*)tab, bytes / 4 - 1, out); });
        printf("global table %5u KB: 2-byte %.1f, 4-byte %.1f, 8-byte %.1f; 4-byte with the wave in one 64-byte line %.1f\n", kb,
               t2 * 1e-3 * ghz * 1e9 / wave_instr_per_cu, t4 * 1e-3 * ghz * 1e9 / wave_instr_per_cu, t8 * 1e-3 * ghz * 1e9 / wave_instr_per_cu,
               l4 * 1e-3 * ghz * 1e9 / wave_instr_per_cu);
    }
    for (uint32_t kb : {16u, 32u, 64u}) {
        const uint32_t bytes = kb << 10;
        const int wg_per_cu = (int)(160u / kb) > 8 ? 8 : (int)(160u / kb);
        const int nb = cus * wg_per_cu;
        const double per_cu = (double)nb * 4 * kIters * kUnroll / cus;
        CHECK(hipFuncSetAttribute((const void*)gather_lds<uint32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        CHECK(hipFuncSetAttribute((const void*)gather_lds<uint2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        const double t4 = time_ms([&] { gather_lds<uint32_t><<<nb, 256, bytes>>>((const uint32_t*)tab, bytes / 4, out); });
        const double t8 = time_ms([&] { gather_lds<uint2><<<nb, 256, bytes>>>((const uint2*)tab, bytes / 8, out); });
        printf("LDS table %3u KB (%d workgroups per CU): 4-byte %.1f, 8-byte %.1f (includes staging the table once per workgroup)\n", kb, wg_per_cu,
               t4 * 1e-3 * ghz * 1e9 / per_cu, t8 * 1e-3 * ghz * 1e9 / per_cu);
    }
    CHECK(hipDeviceSynchronize());
    return 0;
}
