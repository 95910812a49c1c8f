// TEST INFRASTRUCTURE ONLY.  Sanitizer leg of the host library: virgo-plus_amd/host/*.cpp compiled with
// -fsanitize=address,undefined (tests/test_sanitizers.py builds tests/sanitize/_build/host_asan; libvpgpu.so is linked but no device
// call is made) and driven through what runs without a GPU: the .pws loader, DAG levelisation, subsetInit, the replicated-circuit
// builder, randomize, custom circuits, structural hash, statement digest, the verifier's replay of a golden transcript (accept) and of
// a tampered one (reject), the Fiat-Shamir verifier on the committed proof, and a session request that must fail loudly without a GPU.
#include "../../virgo-plus_amd/host/vphost.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static int fail(const char *what) { fprintf(stderr, "host_asan: %s\n", what); return 1; }
static std::vector<uint8_t> slurp(const char *p) {
    std::vector<uint8_t> v;
    FILE *f = fopen(p, "rb");
    if (!f) return v;
    uint8_t b[4096]; size_t n;
    while ((n = fread(b, 1, sizeof b, f)) > 0) v.insert(v.end(), b, b + n);
    fclose(f);
    return v;
}

// argv: SHA256_64.pws  transcript_sha256_x1.bin(gkr slice: bytes 32 .. len-1104)  transcript_randomize_8_12.bin  fs_proof_randomize_6_8_seed5.bin
int main(int argc, char **argv) {
    if (argc < 5) return fail("usage");
    char err[256] = {0};
    {
        vph_circuit *c = vph_circuit_from_pws(argv[1], 1, -1, err, sizeof err);
        if (!c) return fail(err);
        std::vector<uint8_t> t = slurp(argv[2]);
        if (t.size() < 2000) return fail("golden x1 missing");
        const size_t a = 32, b = t.size() - 32 - 16 - 65 * 16;
        if (vph_verify_transcript(c, t.data() + a, b - a, 0) != 0) return fail("golden x1 transcript rejected");
        t[a + 100] ^= 2;
        if (vph_verify_transcript(c, t.data() + a, b - a, 0) == 0) return fail("tampered transcript accepted");
        if (vph_verify_transcript(c, t.data() + a, b - a - 16, 0) == 0) return fail("truncated transcript accepted");
        uint64_t h[2];
        vph_circuit_hash(c, h);
        vph_circuit_free(c);
    }
    {   // replicated builder, two routes
        vph_circuit *c = vph_circuit_from_pws(argv[1], 3, 7, err, sizeof err);
        if (!c) return fail(err);
        if (vph_circuit_layers(c) != 15 || vph_circuit_gates(c) != 3ull * 99949) return fail("x3 shape");
        vph_circuit_free(c);
    }
    {
        vph_circuit *c = vph_circuit_randomize(8, 12, 1);
        std::vector<uint8_t> t = slurp(argv[3]);
        const size_t a = 32, b = t.size() - 32 - 16 - 65 * 16;
        if (vph_verify_transcript(c, t.data() + a, b - a, 0) != 0) return fail("golden randomize(8,12) rejected");
        if (vph_verify_fs(c, t.data() + a, b - a) == 0) return fail("interactive transcript accepted as an FS proof");
        vph_circuit_free(c);
    }
    {
        vph_circuit *c = vph_circuit_randomize(6, 8, 5);
        std::vector<uint8_t> p = slurp(argv[4]);
        if (vph_verify_fs(c, p.data(), p.size()) != 0) return fail("FS fixture rejected");
        p[p.size() / 2] ^= 1;
        if (vph_verify_fs(c, p.data(), p.size()) == 0) return fail("tampered FS proof accepted");
        if (vph_verify_fs(c, p.data(), 0) == 0) return fail("empty FS proof accepted");
        vph_session *s = vph_session_create(c, 0, err, sizeof err);       // no GPU here: must fail with a message, not crash
        if (s) vph_session_free(s);                                        // (on a GPU box it succeeds: fine as well)
        else if (!err[0]) return fail("session failure without a message");
        vph_circuit_free(c);
    }
    {   // custom circuit with every gate type (same generator idea as tests/custom_circuits.py, small)
        const uint64_t sizes[3] = {16, 12, 5};
        std::vector<int32_t> ty, l; std::vector<uint64_t> u, v, cp; std::vector<uint8_t> as;
        for (int i = 0; i < 16; ++i) { ty.push_back(6); l.push_back(-1); u.push_back(1000 + i); v.push_back(0); cp.push_back(0); cp.push_back(0); as.push_back(0); }
        const int kinds[12] = {0, 1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 0};
        for (int i = 0; i < 12; ++i) { ty.push_back(kinds[i]); l.push_back(kinds[i] == 7 || kinds[i] == 8 || kinds[i] >= 10 ? -1 : 0); u.push_back(i); v.push_back((i * 5) % 16); cp.push_back(3 + i); cp.push_back(i); as.push_back(0); }
        for (int i = 0; i < 5; ++i) { ty.push_back(i % 2); l.push_back(i % 2); u.push_back(i * 2); v.push_back(i); cp.push_back(0); cp.push_back(0); as.push_back(0); }
        vph_circuit *c = vph_circuit_custom(3, sizes, ty.data(), l.data(), u.data(), v.data(), cp.data(), as.data());
        if (!c) return fail("custom");
        uint64_t h[2];
        vph_circuit_hash(c, h);
        vph_circuit_free(c);
    }
    puts("host_asan ok");
    return 0;
}
