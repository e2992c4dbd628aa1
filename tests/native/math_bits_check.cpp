// Host-side exhaustive check of laser_slam_amd/csrc/lsa_math.h against the
// container's libm (glibc 2.35): every float with |x| < 120 for sinf/cosf,
// every float for atanf, and a dense pseudo-random sweep of atan2f(y, x).
// Build/run: see tests/test_math_bits.py.  argv[1] = stride (1 = exhaustive).
#include "../../laser_slam_amd/csrc/lsa_math.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include <atomic>

static std::atomic<unsigned long long> bad_sin{0}, bad_cos{0}, bad_atan{0}, bad_atan2{0};

static bool same(float a, float b)
{
    uint32_t ua = lsa::f2u(a), ub = lsa::f2u(b);
    if (ua == ub) return true;
    return (a != a) && (b != b);
}

int main(int argc, char** argv)
{
    unsigned stride = argc > 1 ? (unsigned)atoi(argv[1]) : 1;
    unsigned nthr = std::thread::hardware_concurrency();
    if (nthr == 0) nthr = 4;
    std::vector<std::thread> th;
    const uint32_t lim = lsa::f2u(120.0f);
    for (unsigned t = 0; t < nthr; ++t) {
        th.emplace_back([=]() {
            unsigned long long bs = 0, bc = 0, ba = 0, b2 = 0;
            for (uint64_t u = t * (uint64_t)stride; u < lim; u += (uint64_t)nthr * stride) {
                for (uint32_t sign = 0; sign < 2; ++sign) {
                    float x = lsa::u2f((uint32_t)u | (sign << 31));
                    if (!same(lsa::sinf_(x), sinf(x))) { if (bs < 3) fprintf(stderr, "sin %a: %a vs %a\n", x, lsa::sinf_(x), sinf(x)); ++bs; }
                    { float sn, cs; lsa::sincosf_fast(x, sn, cs); if (!same(sn, sinf(x)) || !same(cs, cosf(x))) { if (bs < 3) fprintf(stderr, "sincos_fast %a\n", x); ++bs; } }
                    { float sn, cs; lsa::sincosf_(x, sn, cs); if (!same(sn, sinf(x)) || !same(cs, cosf(x))) { if (bs < 3) fprintf(stderr, "sincos %a\n", x); ++bs; } }
                    if (!same(lsa::cosf_(x), cosf(x))) { if (bc < 3) fprintf(stderr, "cos %a: %a vs %a\n", x, lsa::cosf_(x), cosf(x)); ++bc; }
                }
            }
            for (uint64_t u = t * (uint64_t)stride; u < 0x100000000ull; u += (uint64_t)nthr * stride) {
                float x = lsa::u2f((uint32_t)u);
                if (!same(lsa::atan2f_fast(x, 1.0f), atan2f(x, 1.0f))) { if (b2 < 3) fprintf(stderr, "atan2_fast(%a,1)\n", x); ++b2; }
                if (!same(lsa::atanf_(x), atanf(x))) { if (ba < 3) fprintf(stderr, "atan %a: %a vs %a\n", x, lsa::atanf_(x), atanf(x)); ++ba; }
            }
            // atan2f: xorshift over (y,x) bit patterns, plus a structured sweep of
            // scan-like magnitudes (|v| in [1e-3, 1e4]).
            uint64_t s = 0x9E3779B97F4A7C15ull * (t + 1);
            uint64_t cnt = (0x40000000ull / stride) / nthr;
            for (uint64_t i = 0; i < cnt; ++i) {
                s ^= s << 13; s ^= s >> 7; s ^= s << 17;
                float y = lsa::u2f((uint32_t)s), x = lsa::u2f((uint32_t)(s >> 32));
                if (!same(lsa::atan2f_(y, x), atan2f(y, x))) { if (b2 < 3) fprintf(stderr, "atan2 %a %a: %a vs %a\n", y, x, lsa::atan2f_(y, x), atan2f(y, x)); ++b2; }
                if (!same(lsa::atan2f_fast(y, x), atan2f(y, x))) { if (b2 < 3) fprintf(stderr, "atan2_fast %a %a\n", y, x); ++b2; }
                // scan-like: exponents 117..140
                uint32_t ey = 117 + (uint32_t)((s >> 3) % 24), ex = 117 + (uint32_t)((s >> 11) % 24);
                y = lsa::u2f(((uint32_t)s & 0x807fffffu) | (ey << 23));
                x = lsa::u2f(((uint32_t)(s >> 32) & 0x807fffffu) | (ex << 23));
                if (!same(lsa::atan2f_fast(y, x), atan2f(y, x))) { if (b2 < 3) fprintf(stderr, "atan2_fast %a %a\n", y, x); ++b2; }
                if (!same(lsa::atan2f_(y, x), atan2f(y, x))) { if (b2 < 3) fprintf(stderr, "atan2 %a %a: %a vs %a\n", y, x, lsa::atan2f_(y, x), atan2f(y, x)); ++b2; }
            }
            bad_sin += bs; bad_cos += bc; bad_atan += ba; bad_atan2 += b2;
        });
    }
    for (auto& t : th) t.join();
    // seg_foot's degenerate-segment test (match_kernels.hip): (double)sqrtf(s) < 0.0001  <=>  s < 0x1.5798f0p-27f.
    // A million floats either side of the threshold, then every float strided: the two predicates must agree.
    unsigned long long bad_seg = 0;
    {
        const float thr = 0x1.5798f0p-27f;
        const uint32_t ut = lsa::f2u(thr);
        for (uint32_t u = ut - 1000000u; u <= ut + 1000000u; ++u) { const float v = lsa::u2f(u); bad_seg += ((double)sqrtf(v) < 0.0001) != (v < thr); }
        for (uint64_t u = 0; u < 0x7f800000ull; u += 997) { const float v = lsa::u2f((uint32_t)u); bad_seg += ((double)sqrtf(v) < 0.0001) != (v < thr); }
    }
    if (bad_seg) { fprintf(stderr, "segment-length threshold: %llu mismatches\n", bad_seg); bad_sin += bad_seg; }
    printf("{\"stride\": %u, \"sin\": %llu, \"cos\": %llu, \"atan\": %llu, \"atan2\": %llu}\n", stride,
           (unsigned long long)bad_sin, (unsigned long long)bad_cos, (unsigned long long)bad_atan,
           (unsigned long long)bad_atan2);
    return (bad_sin | bad_cos | bad_atan | bad_atan2) ? 1 : 0;
}
