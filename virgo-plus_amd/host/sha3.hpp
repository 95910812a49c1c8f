// SHA3-256 of one 64-byte block (the reference's my_hhash, lib/virgo/src/my_hhash.h:27-33, which calls XKCP's
// SHA3_256): host copy used by the verifier to recompute leaf chains and Merkle paths.  FIPS 202.
#pragma once
#include <cstdint>
#include <cstring>

namespace vph {

struct hhash_digest {
    uint64_t w[4];
    bool operator==(const hhash_digest &o) const { return !memcmp(w, o.w, 32); }
    bool operator!=(const hhash_digest &o) const { return !(*this == o); }
};

inline void keccak_f1600(uint64_t s[25]) {
    static const uint64_t rc[24] = {
        0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
        0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
        0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
        0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
        0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    static const int rot[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
    static const int pil[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    for (int r = 0; r < 24; ++r) {
        uint64_t bc[5];
        for (int i = 0; i < 5; ++i) bc[i] = s[i] ^ s[i + 5] ^ s[i + 10] ^ s[i + 15] ^ s[i + 20];
        for (int i = 0; i < 5; ++i) {
            const uint64_t t = bc[(i + 4) % 5] ^ ((bc[(i + 1) % 5] << 1) | (bc[(i + 1) % 5] >> 63));
            for (int j = 0; j < 25; j += 5) s[j + i] ^= t;
        }
        uint64_t t = s[1];
        for (int i = 0; i < 24; ++i) {           // rho + pi along the single 24-cycle of the lane permutation
            const int j = pil[i];
            const uint64_t b = s[j];
            s[j] = (t << rot[i]) | (t >> (64 - rot[i]));
            t = b;
        }
        for (int j = 0; j < 25; j += 5) {
            uint64_t b[5];
            for (int i = 0; i < 5; ++i) b[i] = s[j + i];
            for (int i = 0; i < 5; ++i) s[j + i] = b[i] ^ (~b[(i + 1) % 5] & b[(i + 2) % 5]);
        }
        s[0] ^= rc[r];
    }
}

// digest of (m[0..4) || prev) — the 64-byte block of the leaf chains and of the Merkle nodes
inline hhash_digest hhash(const uint64_t m[4], const hhash_digest &prev) {
    uint64_t s[25] = {0};
    s[0] = m[0]; s[1] = m[1]; s[2] = m[2]; s[3] = m[3];
    s[4] = prev.w[0]; s[5] = prev.w[1]; s[6] = prev.w[2]; s[7] = prev.w[3];
    s[8] = 0x06; s[16] = 0x8000000000000000ull;
    keccak_f1600(s);
    hhash_digest d; d.w[0] = s[0]; d.w[1] = s[1]; d.w[2] = s[2]; d.w[3] = s[3];
    return d;
}

// Streaming SHA3-256 (FIPS 202 sponge, rate 136 bytes): the statement digest of the Fiat-Shamir mode absorbs the whole
// serialised circuit and its input values through it (verifier::fsInit).
class sha3_256 {
public:
    void update(const void *data, size_t n) {
        const uint8_t *p = static_cast<const uint8_t *>(data);
        while (n) {
            const size_t take = n < 136 - fill ? n : 136 - fill;
            memcpy(buf + fill, p, take);
            fill += take; p += take; n -= take;
            if (fill == 136) { absorb(); fill = 0; }
        }
    }
    void put64(uint64_t x) { update(&x, 8); }
    hhash_digest final() {
        memset(buf + fill, 0, 136 - fill);
        buf[fill] ^= 0x06; buf[135] ^= 0x80;
        absorb();
        hhash_digest d; memcpy(d.w, s, 32);
        return d;
    }
private:
    void absorb() {
        for (int i = 0; i < 17; ++i) { uint64_t w; memcpy(&w, buf + 8 * i, 8); s[i] ^= w; }
        keccak_f1600(s);
    }
    uint64_t s[25] = {0};
    uint8_t buf[136];
    size_t fill = 0;
};

}  // namespace vph
