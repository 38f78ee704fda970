// oracle/glibc_rand.h -- TEST INFRASTRUCTURE ONLY.
//
// Model of glibc's default random()/rand() generator (TYPE_3: additive lagged-Fibonacci feedback,
// degree 31, separation 3) as seeded by srand(seed).  The reference draws its parent flags from
// the never-seeded process-global libc rand() (src/cpp_ext/include/base.hpp:44-56), i.e. the
// srand(1) stream of a fresh process; holding the state in a struct lets one process replay that
// stream from any position.  Checked against libc rand() itself in tests/test_rng.py.
#pragma once
#include <stdint.h>

struct GlibcRand {
    uint32_t st[31];
    int f, r;            // front / rear indices into st

    void seed(uint32_t s) {
        if (s == 0) s = 1;
        int32_t word = (int32_t)s;
        st[0] = (uint32_t)word;
        for (int i = 1; i < 31; ++i) {
            // 16807 * word mod 2147483647 without overflow (Schrage)
            int32_t hi = word / 127773;
            int32_t lo = word % 127773;
            word = 16807 * lo - 2836 * hi;
            if (word < 0) word += 2147483647;
            st[i] = (uint32_t)word;
        }
        f = 3; r = 0;
        for (int i = 0; i < 310; ++i) next();
    }

    // one libc rand() value, in [0, 2^31)
    uint32_t next() {
        uint32_t v = (st[f] += st[r]);
        if (++f == 31) f = 0;
        if (++r == 31) r = 0;
        return v >> 1;
    }

    // hem::rand(): eight successive rand() % 16 nibbles, first draw in bits 0..3  (base.hpp:44-50)
    uint32_t hem_rand() {
        uint32_t x = 0;
        for (int i = 0; i < 8; ++i) x |= (next() % 16u) << (4 * i);
        return x;
    }

    // hem::rand01(): float(r) / uint(0xffffffff); the divisor converts to float 4294967296.0f  (base.hpp:53-56)
    float hem_rand01() {
        return (float)hem_rand() / (float)0xffffffffu;
    }
};
