/* philox.c — Philox4x32-R and the draw addressing of include/rt_rng.h (R = RT_PHILOX_ROUNDS).
 *
 * TEST INFRASTRUCTURE (see oracle.h).  The algorithm is the published one
 * (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2,
 * 3", SC'11); it replaces `rand 0.8.5`'s ThreadRng that the reference calls
 * at racer-tracer/src/util.rs:9-23 (that crate is not in /root/reference and
 * is unseedable, so only its distribution is matched: 53-bit uniform [0,1);
 * the three coordinates of a rejection-sphere candidate at 42 bits each, rt_rng.h).
 * Pinned by the Random123 known-answer vectors in
 * tests/test_oracle_reference_vectors.py.
 */
#include "oracle.h"

static inline void mulhilo(uint32_t a, uint32_t b, uint32_t *hi, uint32_t *lo) {
    uint64_t p = (uint64_t)a * (uint64_t)b;
    *hi = (uint32_t)(p >> 32);
    *lo = (uint32_t)p;
}

void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], int rounds, uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < rounds; ++round) {
        uint32_t hi0, lo0, hi1, lo1;
        mulhilo(RT_PHILOX_M0, c0, &hi0, &lo0);
        mulhilo(RT_PHILOX_M1, c2, &hi1, &lo1);
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n1 = lo1;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        uint32_t n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += RT_PHILOX_W0;
        k1 += RT_PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline double u53(uint32_t hi, uint32_t lo) {
    uint64_t w = ((uint64_t)hi << 32) | (uint64_t)lo;
    return (double)(w >> 11) * (1.0 / 9007199254740992.0); /* 2^-53 */
}

double orc_rng_double(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t segment,
                      uint32_t purpose, uint32_t block, int which) {
    uint32_t ctr[4] = { pixel, sample, (segment << 8) | purpose, block };
    uint32_t key[2] = { (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32) };
    uint32_t out[4];
    orc_philox4x32(ctr, key, RT_PHILOX_ROUNDS, out);
    return which ? u53(out[2], out[3]) : u53(out[0], out[1]);
}

/* e_0, e_1, e_2 of the addressed block: three 42-bit uniforms (include/rt_rng.h) */
void orc_rng_triple(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t segment,
                    uint32_t purpose, uint32_t block, double e[3]) {
    uint32_t ctr[4] = { pixel, sample, (segment << 8) | purpose, block };
    uint32_t key[2] = { (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32) };
    uint32_t out[4];
    orc_philox4x32(ctr, key, RT_PHILOX_ROUNDS, out);
    for (int j = 0; j < 3; ++j) {
        uint64_t u = ((uint64_t)out[j] << 10) | (uint64_t)((out[3] >> (10 * j)) & 0x3FFu);
        e[j] = (double)u * (1.0 / 4398046511104.0); /* 2^-42 */
    }
}
