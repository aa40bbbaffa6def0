// valu_rates.hip -- issue-rate microbenchmark for the VALU instructions the reach/distance kernels are
// made of (they are VALU-issue bound).  For each instruction: a wave executes
// ITERS x 32 copies on 8 independent destination registers; every SIMD of the chip holds W such waves
// (W = 1, 2, 4).  Reported: cycles per wave-instruction per SIMD at the clock the device reports.
//
//   hipcc --offload-arch=gfx950 -O2 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string_view>
#include <vector>

#define ITERS 4000

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

// %0 = destination d[i] (read-write), %1, %2 = VGPR sources, %3 = SGPR source
// one instruction on register d[i & 7]
#define ONE(ASM, i) asm volatile(ASM : "+v"(d[i]) : "v"(va), "v"(vb), "s"(s) : "vcc", "s20", "s21");

#define fma_ONE(i) ONE("v_fma_f32 %0, %1, %2, %0", i)
#define fmac_ONE(i) ONE("v_fmac_f32 %0, %1, %2", i)
#define mul_ONE(i) ONE("v_mul_f32 %0, %1, %0", i)
#define add_ONE(i) ONE("v_add_f32 %0, %1, %0", i)
#define sub_ONE(i) ONE("v_sub_f32 %0, %1, %0", i)
#define mul_sgpr_ONE(i) ONE("v_mul_f32 %0, %3, %0", i)
#define fma_sgpr_ONE(i) ONE("v_fma_f32 %0, %1, %3, %0", i)
#define add_abs_ONE(i) ONE("v_add_f32 %0, |%1|, %0", i)
#define fma_neg_ONE(i) ONE("v_fma_f32 %0, -%1, %2, %0", i)
#define max_ONE(i) ONE("v_max_f32 %0, %1, %0", i)
#define min_ONE(i) ONE("v_min_f32 %0, %1, %0", i)
#define max3_ONE(i) ONE("v_max3_f32 %0, %1, %2, %0", i)
#define min3_ONE(i) ONE("v_min3_f32 %0, %1, %2, %0", i)
#define med3_ONE(i) ONE("v_med3_f32 %0, %1, %2, %0", i)
#define and_ONE(i) ONE("v_and_b32 %0, %1, %0", i)
#define or_ONE(i) ONE("v_or_b32 %0, %1, %0", i)
#define xor_ONE(i) ONE("v_xor_b32 %0, %1, %0", i)
#define and_or_ONE(i) ONE("v_and_or_b32 %0, %1, %2, %0", i)
#define bfi_ONE(i) ONE("v_bfi_b32 %0, %1, %2, %0", i)
#define add_u32_ONE(i) ONE("v_add_u32 %0, %1, %0", i)
#define lshl_ONE(i) ONE("v_lshlrev_b32 %0, 1, %0", i)
#define lshl_add_ONE(i) ONE("v_lshl_add_u32 %0, %1, 2, %0", i)
#define mov_ONE(i) ONE("v_mov_b32 %0, %1", i)
#define min_u32_ONE(i) ONE("v_min_u32 %0, %1, %0", i)
#define max_u32_ONE(i) ONE("v_max_u32 %0, %1, %0", i)
#define cndmask_ONE(i) ONE("v_cndmask_b32 %0, %1, %0, vcc", i)
#define cmp_vcc_ONE(i) ONE("v_cmp_lt_f32 vcc, %1, %0", i)
#define cmp_sgpr_ONE(i) ONE("v_cmp_lt_f32 s[20:21], %1, %0", i)
#define cmp_cnd_ONE(i) ONE("v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %0, %2, %0, vcc", i)
#define cmp_class_ONE(i) ONE("v_cmp_class_f32 vcc, %0, %1", i)
#define rsq_ONE(i) ONE("v_rsq_f32 %0, %0", i)
#define sqrt_ONE(i) ONE("v_sqrt_f32 %0, %0", i)
#define rcp_ONE(i) ONE("v_rcp_f32 %0, %0", i)
#define sin_ONE(i) ONE("v_sin_f32 %0, %0", i)
#define mul_lo_ONE(i) ONE("v_mul_lo_u32 %0, %1, %0", i)
#define mul_hi_ONE(i) ONE("v_mul_hi_u32 %0, %1, %0", i)
#define pk_fma_ONE(i) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(dd[i]) : "v"(vd));
#define pk_mul_ONE(i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(dd[i]) : "v"(vd));
#define pk_add_ONE(i) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(dd[i]) : "v"(vd));
#define fma64_ONE(i) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(dd[i]) : "v"(vd));
#define mul64_ONE(i) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(dd[i]) : "v"(vd));
#define add64_ONE(i) asm volatile("v_add_f64 %0, %1, %0" : "+v"(dd[i]) : "v"(vd));
#define cvt_f64_f32_ONE(i) asm volatile("v_cvt_f64_f32 %0, %1" : "+v"(dd[i]) : "v"(va));
#define cvt_f32_f64_ONE(i) asm volatile("v_cvt_f32_f64 %0, %1" : "+v"(d[i]) : "v"(vd));
#define readlane_ONE(i) asm volatile("v_readlane_b32 s20, %0, 3" : "+v"(d[i])::"s20");
#define writelane_ONE(i) asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(d[i]) : "s"(s));
#define div_scale_ONE(i) ONE("v_div_scale_f32 %0, vcc, %1, %2, %0", i)
#define div_fmas_ONE(i) ONE("v_div_fmas_f32 %0, %1, %2, %0", i)
#define div_fixup_ONE(i) ONE("v_div_fixup_f32 %0, %1, %2, %0", i)
#define ldexp_ONE(i) ONE("v_ldexp_f32 %0, %0, 1", i)
#define mov_dpp_ONE(i) ONE("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", i)
#define add_dpp_ONE(i) ONE("v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", i)
#define dep_fma_ONE(i) ONE("v_fma_f32 %0, %1, %2, %0", 0)

#define mul_inl_ONE(i) ONE("v_mul_f32 %0, 2.0, %0", i)
#define mul_lit_ONE(i) ONE("v_mul_f32 %0, 0x40490fdb, %0", i)
#define add_lit_ONE(i) ONE("v_add_f32 %0, 0x40490fdb, %0", i)
#define and_lit_ONE(i) ONE("v_and_b32 %0, 0x7fffffff, %0", i)
#define xor_lit_ONE(i) ONE("v_xor_b32 %0, 0x80000000, %0", i)
#define mul_e64_ONE(i) ONE("v_mul_f32_e64 %0, %1, %0", i)
#define add_e64_ONE(i) ONE("v_add_f32_e64 %0, %1, %0", i)
#define fmac_sgpr_ONE(i) ONE("v_fmac_f32 %0, %3, %2", i)
#define fmaak_ONE(i) ONE("v_fmaak_f32 %0, %1, %0, 0x40490fdb", i)
#define fmamk_ONE(i) ONE("v_fmamk_f32 %0, %1, 0x40490fdb, %0", i)
#define subrev_ONE(i) ONE("v_subrev_f32 %0, %1, %0", i)
#define lshr_ONE(i) ONE("v_lshrrev_b32 %0, 1, %0", i)
#define ashr_ONE(i) ONE("v_ashrrev_i32 %0, 31, %0", i)
#define bfe_ONE(i) ONE("v_bfe_u32 %0, %0, 3, 5", i)
#define sub_u32_ONE(i) ONE("v_sub_u32 %0, %1, %0", i)
#define add3_ONE(i) ONE("v_add3_u32 %0, %1, %2, %0", i)
#define or3_ONE(i) ONE("v_or3_b32 %0, %1, %2, %0", i)
#define cvt_f32_u32_ONE(i) ONE("v_cvt_f32_u32 %0, %0", i)
#define cvt_u32_f32_ONE(i) ONE("v_cvt_u32_f32 %0, %0", i)
#define cnd_sgpr_ONE(i) ONE("v_cndmask_b32 %0, %1, %0, s[20:21]", i)
#define cmp_cnd64_ONE(i) ONE("v_cmp_lt_f32 s[20:21], %1, %0\n v_cndmask_b32 %0, %2, %0, s[20:21]", i)
#define cmp_u32_ONE(i) ONE("v_cmp_lt_u32 vcc, %1, %0", i)
#define not_ONE(i) ONE("v_not_b32 %0, %0", i)
#define mov_lit_ONE(i) ONE("v_mov_b32 %0, 0x12345678", i)
#define mov_sgpr_ONE(i) ONE("v_mov_b32 %0, %3", i)
#define add_sgpr_ONE(i) ONE("v_add_f32 %0, %3, %0", i)
#define fract_ONE(i) ONE("v_fract_f32 %0, %0", i)
#define rndne_ONE(i) ONE("v_rndne_f32 %0, %0", i)
#define mul_legacy_ONE(i) ONE("v_mul_legacy_f32 %0, %1, %0", i)
#define mix_fast_slow_ONE(i) ONE("v_mul_f32 %0, %1, %0\n v_max_f32 %0, %2, %0", i)

// 32-bit kernels
#define K32(NAME)                                                                                 \
    __global__ __launch_bounds__(256) void k_##NAME(float* out, float a, float b, float s) {       \
        float d[8];                                                                               \
        for (int i = 0; i < 8; i++) d[i] = a + (float)i + (float)threadIdx.x;                    \
        float va = a * (float)threadIdx.x, vb = b + (float)threadIdx.x;                          \
        double vd = (double)va;                                                                   \
        (void)vd;                                                                                 \
        for (int it = 0; it < ITERS; it++) {                                                      \
            REP32(NAME##_ONE)                                                                     \
        }                                                                                         \
        float r = 0;                                                                              \
        for (int i = 0; i < 8; i++) r += d[i];                                                    \
        out[blockIdx.x * 256 + threadIdx.x] = r;                                                  \
    }
// 64-bit-register kernels (packed FP32 and FP64)
#define K64(NAME)                                                                                 \
    __global__ __launch_bounds__(256) void k_##NAME(float* out, float a, float b, float s) {       \
        double dd[8];                                                                             \
        for (int i = 0; i < 8; i++) dd[i] = (double)a + i + threadIdx.x;                          \
        float va = a * (float)threadIdx.x;                                                        \
        double vd = (double)b + threadIdx.x;                                                      \
        (void)va;                                                                                 \
        for (int it = 0; it < ITERS; it++) {                                                      \
            REP32(NAME##_ONE)                                                                     \
        }                                                                                         \
        double r = 0;                                                                             \
        for (int i = 0; i < 8; i++) r += dd[i];                                                   \
        out[blockIdx.x * 256 + threadIdx.x] = (float)r;                                           \
    }

#define LIST32(X)                                                                                  \
    X(fma) X(fmac) X(mul) X(add) X(sub) X(mul_sgpr) X(fma_sgpr) X(add_abs) X(fma_neg) X(max) X(min) X(max3)     \
    X(min3) X(med3) X(and) X(or) X(xor) X(and_or) X(bfi) X(add_u32) X(lshl) X(lshl_add) X(mov) X(min_u32)       \
    X(max_u32) X(cndmask) X(cmp_vcc) X(cmp_sgpr) X(cmp_cnd) X(cmp_class) X(rsq) X(sqrt) X(rcp) X(sin) X(mul_lo) \
    X(mul_hi) X(cvt_f32_f64) X(readlane) X(writelane) X(div_scale) X(div_fmas) X(div_fixup) X(ldexp)            \
    X(mov_dpp) X(add_dpp) X(dep_fma) X(mul_inl) X(mul_lit) X(add_lit) X(and_lit) X(xor_lit) X(mul_e64) X(add_e64) \
    X(fmac_sgpr) X(fmaak) X(fmamk) X(subrev) X(lshr) X(ashr) X(bfe) X(sub_u32) X(add3) X(or3) X(cvt_f32_u32) \
    X(cvt_u32_f32) X(cnd_sgpr) X(cmp_cnd64) X(cmp_u32) X(not) X(mov_lit) X(mov_sgpr) X(add_sgpr) X(fract) X(rndne) \
    X(mul_legacy) X(mix_fast_slow)
#define LIST64(X) X(pk_fma) X(pk_mul) X(pk_add) X(fma64) X(mul64) X(add64) X(cvt_f64_f32)

LIST32(K32)
LIST64(K64)

typedef void (*kern_t)(float*, float, float, float);
struct Entry {
    const char* name;
    kern_t fn;
    int per_copy; // instructions per "ONE"
};
#define ENT(NAME) {#NAME, k_##NAME, 1},

int main() {
    std::vector<Entry> list = {LIST32(ENT) LIST64(ENT)};
    for (auto& e : list)
        if (std::string_view(e.name) == "cmp_cnd" || std::string_view(e.name) == "cmp_cnd64" ||
            std::string_view(e.name) == "mix_fast_slow")
            e.per_copy = 2;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) {
        fprintf(stderr, "no HIP device\n");
        return 1;
    }
    const int cus = prop.multiProcessorCount;
    const double mhz = prop.clockRate / 1000.0;
    float* out;
    hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("# %s, %d CUs, %.0f MHz reported; cycles per wave64 instruction per SIMD\n", prop.gcnArchName, cus, mhz);
    printf("%-14s %8s %8s %8s\n", "instruction", "1 wave", "2 waves", "4 waves");
    for (auto& e : list) {
        printf("%-14s", e.name);
        for (int w : {1, 2, 4}) {
            // blocks of 256 threads = one wave on each of the CU's 4 SIMDs
            hipLaunchKernelGGL(e.fn, dim3(cus * w), dim3(256), 0, 0, out, 1.0f, 2.0f, 3.0f);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(e.fn, dim3(cus * w), dim3(256), 0, 0, out, 1.0f, 2.0f, 3.0f);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double instr_per_simd = (double)w * ITERS * 32 * e.per_copy;
            printf(" %8.2f", ms * 1e-3 * mhz * 1e6 / instr_per_simd);
        }
        printf("\n");
    }
    return 0;
}
