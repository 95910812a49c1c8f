// TEST INFRASTRUCTURE ONLY.  Sanitizer leg of the CPU build (SURVEY.md §5 "ASan/UBSan CPU targets"): the oracle compiled with
// -fsanitize=address,undefined (make -C oracle asan -> oracle/_build/oracle_asan) and driven through every code path the parity tests use —
// loader + replication, randomize, custom circuits with every gate type, GKR proof, Fiat-Shamir proof, the full protocol with the
// commitment, FRI commit phase, primitives.  Exit code 0 and no sanitizer report = pass (tests/test_sanitizers.py).
#include "../../oracle/vp_oracle.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static int fail(const char *what) { fprintf(stderr, "oracle_asan: %s\n", what); return 1; }

int main(int argc, char **argv) {
    std::vector<uint8_t> buf(1 << 20);
    orc_stats st;
    {   // synthetic circuit: GKR, FS, full protocol (input layer 2^9: the commitment's minimum meaningful size)
        orc_circuit *c = orc_circuit_randomize(4, 9, 3);
        if (!c) return fail("randomize");
        const int64_t n = orc_prove_gkr(c, buf.data(), (int64_t) buf.size(), &st);
        if (n <= 0 || !st.verified) return fail("prove_gkr");
        std::vector<uint8_t> fs(1 << 20);
        const int64_t m = orc_prove_fs(c, fs.data(), (int64_t) fs.size(), &st);
        if (m != n || !st.verified || !memcmp(fs.data(), buf.data(), (size_t) n)) return fail("prove_fs");
        const int64_t k = orc_prove_full(c, buf.data(), (int64_t) buf.size(), &st);
        if (k != n + 32 + 32 + 16 + 65 * 16) return fail("prove_full");
        uint8_t root[32];
        if (orc_commit_private(c, root) != 0 || memcmp(root, buf.data(), 32)) return fail("commit_private");
        const int nb = orc_circuit_layer_bitlen(c, 0);
        std::vector<orc_F> in(1u << nb), pub(1u << nb), r(nb - 6), fin(2048);
        orc_circuit_inputs(c, in.data());
        orc_f_random_seq(9, (int) pub.size(), pub.data());
        orc_f_random_next((int) r.size(), r.data());
        std::vector<uint8_t> roots(32 * r.size());
        if (orc_fri_commit(in.data(), pub.data(), nb, r.data(), roots.data(), fin.data()) != 0) return fail("fri_commit");
        orc_circuit_free(c);
    }
    {   // tiny and ragged shapes: one-gate layers, single-entry tables
        for (int lg = 0; lg < 4; ++lg) {
            orc_circuit *c = orc_circuit_randomize(3, lg, 5 + lg);
            if (orc_prove_gkr(c, buf.data(), (int64_t) buf.size(), &st) <= 0 || !st.verified) return fail("tiny prove_gkr");
            if (orc_prove_fs(c, buf.data(), (int64_t) buf.size(), &st) <= 0 || !st.verified) return fail("tiny prove_fs");
            orc_circuit_free(c);
        }
    }
    if (argc > 1) {   // the reference's data file: loader, levelisation, subsetInit, two replicated blocks
        orc_circuit *c = orc_circuit_from_pws(argv[1], 2, 1);
        if (!c) return fail("from_pws");
        if (orc_prove_gkr(c, buf.data(), (int64_t) buf.size(), &st) <= 0 || !st.verified) return fail("pws prove_gkr");
        uint64_t h[2];
        orc_circuit_hash(c, h);
        orc_circuit_free(c);
    }
    {   // primitives
        orc_F a = {1234567890123456789ull, 987654321987654321ull}, b = {2305843009213693950ull, 1}, o, t;
        orc_f_mul(&a, &b, &o); orc_f_inv(&a, &t); orc_f_mul(&a, &t, &o);
        if (o.real != 1 || o.img != 0) return fail("inv");
        std::vector<orc_F> c8(8), ev(32), back(32);
        for (int i = 0; i < 8; ++i) c8[i] = {(uint64_t) i + 1, (uint64_t) 2 * i + 3};
        orc_fft(c8.data(), 8, 32, ev.data());
        if (ev[0].real != 36 || ev[0].img != 80) return fail("fft");
        orc_ifft(ev.data(), 32, back.data());
        for (int i = 0; i < 8; ++i) if (back[i].real != c8[i].real || back[i].img != c8[i].img) return fail("ifft");
        std::vector<orc_F> rr(5), tab(32);
        orc_f_random_seq(3396, 5, rr.data());
        orc_F one = {1, 0};
        orc_beta_table(rr.data(), 5, &one, tab.data());
        uint8_t z[64] = {0}, d[32];
        orc_sha3_256_64(z, d);
        if (d[0] != 0x07 || d[1] != 0x0f) return fail("sha3");
    }
    puts("oracle_asan ok");
    return 0;
}
