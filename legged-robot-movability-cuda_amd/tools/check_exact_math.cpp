// Exhaustive / large-sample comparison of csrc/lrm_exact_math.h (host build) with this machine's libm.
//   g++ -O2 -std=c++17 -mfma -ffp-contract=off -I../csrc -o check_exact_math check_exact_math.cpp -lm
//   ./check_exact_math sincos   every float with |x| < 120 (2.2e9 values), both results
//   ./check_exact_math atan     every one of the 2^32 floats through atanf
//   ./check_exact_math atan2    6e8 pseudo-random and structured (y, x) pairs
// (-mfma: glibc selects its FMA build of sincosf on AVX2 hosts; see the header of lrm_exact_math.h.)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include "lrm_exact_math.h"

static bool same(float a, float b) { return lrm_f2u(a) == lrm_f2u(b) || (a != a && b != b); }

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    unsigned long bad = 0, n = 0;
    if (!strcmp(argv[1], "sincos")) {
        for (uint64_t u = 0; u < 0x100000000ull; u++) {
            const float x = lrm_u2f((uint32_t)u);
            if (!(fabsf(x) < 120.f)) continue;
            float s0, c0, s1, c1;
            sincosf(x, &s0, &c0);
            lrm_sincosf(x, &s1, &c1);
            n++;
            if (!same(s0, s1) || !same(c0, c1)) {
                if (bad < 5) printf("sincosf %a: (%a, %a) vs (%a, %a)\n", x, s0, c0, s1, c1);
                bad++;
            }
        }
    } else if (!strcmp(argv[1], "atan")) {
        for (uint64_t u = 0; u < 0x100000000ull; u++) {
            const float x = lrm_u2f((uint32_t)u);
            n++;
            if (!same(atanf(x), lrm_atanf(x))) {
                if (bad < 5) printf("atanf %a: %a vs %a\n", x, atanf(x), lrm_atanf(x));
                bad++;
            }
        }
    } else {
        uint64_t st = 88172645463325252ull;
        for (long i = 0; i < 600000000L; i++) {
            st ^= st << 13; st ^= st >> 7; st ^= st << 17;
            const uint32_t a = (uint32_t)st, b = (uint32_t)(st >> 32);
            float y, x;
            switch (i % 4) {
            case 0: y = lrm_u2f(a); x = lrm_u2f(b); break;
            case 1: y = (int32_t)a * (1.0f / 4194304.0f); x = (int32_t)b * (1.0f / 4194304.0f); break;
            case 2: y = (int32_t)a * (1.0f / 4194304.0f); x = (b & 1) ? 1.0f : ((b & 2) ? 0.0f : -0.0f); break;
            default:
                y = (a & 7) == 0 ? 0.0f : (int32_t)a * 1e-3f;
                x = (int32_t)b * 1e-3f;
                if ((a & 0xf0) == 0) y = INFINITY;
                if ((b & 0xf00) == 0) x = -INFINITY;
            }
            n++;
            if (!same(atan2f(y, x), lrm_atan2f(y, x))) {
                if (bad < 5) printf("atan2f(%a, %a): %a vs %a\n", y, x, atan2f(y, x), lrm_atan2f(y, x));
                bad++;
            }
        }
    }
    printf("%s: %lu values, %lu mismatches\n", argv[1], n, bad);
    return bad != 0;
}
