// py_random.h -- CPython's `random` stream, for regenerating what the reference generated from an integer seed.
#pragma once

#include <stdint.h>

namespace tpl {

// The reference draws from Python's `random` module (game/tetris.py:85,93,250,253).  This is that generator:
// MT19937 seeded the way CPython's random.seed(int) seeds it (init_by_array over the 32-bit digits of the seed),
// and randint / shuffle built on _randbelow_with_getrandbits (k = n.bit_length(); draw k bits until < n) -- so a
// configuration can be regenerated from the same integer seed the reference was given.
struct PyRandom {
    uint32_t mt[624];
    int idx = 625;

    void init_genrand(uint32_t s) {
        mt[0] = s;
        for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        idx = 624;
    }
    explicit PyRandom(uint64_t seed) {
        uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
        const int len = key[1] ? 2 : 1;
        init_genrand(19650218u);
        int i = 1, j = 0;
        for (int k = 624 > len ? 624 : len; k; --k) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
            if (++i >= 624) { mt[0] = mt[623]; i = 1; }
            if (++j >= len) j = 0;
        }
        for (int k = 623; k; --k) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
            if (++i >= 624) { mt[0] = mt[623]; i = 1; }
        }
        mt[0] = 0x80000000u;
    }
    uint32_t next() {
        if (idx >= 624) {
            for (int k = 0; k < 624; ++k) {
                const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7FFFFFFFu);
                mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
            }
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= y >> 11; y ^= (y << 7) & 0x9D2C5680u; y ^= (y << 15) & 0xEFC60000u; y ^= y >> 18;
        return y;
    }
    uint32_t randbelow(uint32_t n) {                     // n >= 1
        const int k = 32 - __builtin_clz(n);             // n.bit_length()
        uint32_t r;
        do r = next() >> (32 - k); while (r >= n);
        return r;
    }
    int randint(int lo, int hi) { return lo + (int)randbelow((uint32_t)(hi - lo + 1)); }
};

}  // namespace tpl
