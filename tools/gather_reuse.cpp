// Development tool: how much operand reuse is there for a workgroup of k_sumfold3b_gen_multi to stage in LDS?
//   g++ -std=c++17 -O2 -o tools/_build/gather_reuse tools/gather_reuse.cpp -Lvirgo-plus_amd/host -lvphost -Lvirgo-plus_amd/csrc -lvpgpu -Wl,-rpath,... ; gather_reuse FILE.pws BLOCKS
// For every layer i the phase-1 init (src/prover.cpp:189-280) generated inside the fold launch walks the target-sorted contribution list: the workgroup's chunk =
// 512 consecutive targets u of layer i-1, a wave's share = 128 of them (two per lane).  Per contribution (gate g of layer i with gate.u = u) the kernel gathers
// eq(r, g) as two half-table entries (bf[g mod 2^h], bs[g >> h]; 16 B each) and, for a binary gate, the other operand V_l[v] (8 B: the witness is real).
// This prints, per layer and for the whole circuit: contributions per chunk, distinct 128-byte lines behind the V gathers of a chunk (and of a wave's 128
// rows), their span (max - min line + 1: what a coalesced window load into LDS would have to fetch), and the same for the two half tables.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <numeric>
#include <set>
#include <string>
#include <vector>
#include "../virgo-plus_amd/host/circuit.hpp"
typedef unsigned int u32;

struct Acc { double n = 0, gathers = 0, vlines = 0, vspan = 0, vlayers = 0, wave_vlines = 0, wave_gathers = 0, waves = 0, bf = 0, bs = 0, gspan = 0; };

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: gather_reuse FILE.pws BLOCKS\n"); return 2; }
    layeredCircuit C; std::string err;
    srandom(1);
    if (!vph::build_replicated(argv[1], atoi(argv[2]), C, &err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    Acc tot;
    printf("%-6s %10s %9s | per 512-target chunk: %9s %9s %9s %7s | per wave (128 targets): %8s %8s | half tables per chunk: %6s %6s %9s\n", "layer", "targets", "contribs",
           "gathersV", "linesV", "spanV", "srcs", "gathersV", "linesV", "bf", "bs", "span_g");
    for (int i = C.size - 1; i >= 1; --i) {
        const layer &cur = C.circuit[i];
        const u64 T = C.circuit[i - 1].size;
        // contributions by target
        std::vector<std::vector<u32>> by_u(T);
        for (u64 g = 0; g < cur.size; ++g) by_u[cur.gates[g].u].push_back((u32) g);
        const int bl = cur.bitLength, h1 = bl >> 1;
        Acc a;
        u64 contribs = 0;
        for (u64 c0 = 0; c0 < T; c0 += 512) {
            std::set<std::pair<int, u64>> lines; std::set<u64> bf, bs; std::map<int, std::pair<u64, u64>> mm;
            u64 gmin = ~0ull, gmax = 0, gathers = 0;
            for (u64 w0 = c0; w0 < std::min(T, c0 + 512); w0 += 128) {
                std::set<std::pair<int, u64>> wl; u64 wg = 0;
                for (u64 u = w0; u < std::min(T, w0 + 128); ++u)
                    for (u32 g : by_u[u]) {
                        const gate &G = cur.gates[g];
                        ++contribs;
                        bf.insert((g & ((1u << h1) - 1)) >> 3); bs.insert(((u64) g >> h1) >> 3);
                        gmin = std::min<u64>(gmin, g); gmax = std::max<u64>(gmax, g);
                        const bool unary = G.ty == Not || G.ty == Copy || G.ty == Addc || G.ty == Mulc;
                        if (unary) continue;
                        const int l = G.l == -1 ? i - 1 : G.l;
                        const u64 line = G.v >> 4;                 // 8-byte real values: 16 per 128-byte line
                        lines.insert({l, line}); wl.insert({l, line}); ++gathers; ++wg;
                        auto it = mm.find(l);
                        if (it == mm.end()) mm[l] = {line, line}; else { it->second.first = std::min(it->second.first, line); it->second.second = std::max(it->second.second, line); }
                    }
                a.wave_vlines += wl.size(); a.wave_gathers += wg; a.waves += 1;
            }
            u64 span = 0;
            for (auto &kv : mm) span += kv.second.second - kv.second.first + 1;
            a.n += 1; a.gathers += gathers; a.vlines += lines.size(); a.vspan += span; a.vlayers += mm.size(); a.bf += bf.size(); a.bs += bs.size();
            a.gspan += gmax >= gmin ? (double) ((gmax - gmin) >> 3) + 1 : 0;
        }
        printf("%-6d %10llu %9llu | %31.1f %9.1f %9.1f %7.2f | %32.1f %8.1f | %29.1f %6.1f %9.1f\n", i, (unsigned long long) T, (unsigned long long) contribs,
               a.gathers / a.n, a.vlines / a.n, a.vspan / a.n, a.vlayers / a.n, a.wave_gathers / a.waves, a.wave_vlines / a.waves, a.bf / a.n, a.bs / a.n, a.gspan / a.n);
        tot.n += a.n; tot.gathers += a.gathers; tot.vlines += a.vlines; tot.vspan += a.vspan; tot.vlayers += a.vlayers; tot.wave_vlines += a.wave_vlines;
        tot.wave_gathers += a.wave_gathers; tot.waves += a.waves; tot.bf += a.bf; tot.bs += a.bs; tot.gspan += a.gspan;
    }
    printf("all: chunks %.0f; V gathers per chunk %.1f, distinct 128-B lines %.1f (reuse %.2f gathers per line; a perfectly contiguous operand run would give 16), span of a window load %.1f lines "
           "(%.2f x the lines actually used), source layers per chunk %.2f; per wave: %.1f gathers over %.1f lines; half tables: %.1f + %.1f lines per chunk, gate-index span %.1f lines\n",
           tot.n, tot.gathers / tot.n, tot.vlines / tot.n, tot.gathers / tot.vlines, tot.vspan / tot.n, tot.vspan / tot.vlines, tot.vlayers / tot.n,
           tot.wave_gathers / tot.waves, tot.wave_vlines / tot.waves, tot.bf / tot.n, tot.bs / tot.n, tot.gspan / tot.n);
    return 0;
}
