// TEST INFRASTRUCTURE ONLY — never linked into the product.
//
// The second forwarding file of INTEGRATION.md: what a reference maintainer puts in place of lib/virgo/src/fri.cpp so that the
// UNMODIFIED lib/virgo verifier (vpd_verifier.cpp: poly_commit_prover::commit_phase :44-74 and
// poly_commit_verifier::verify_poly_commitment :76-328) is served by the MI355X library.  Compiled against the reference's own
// lib/virgo/src/fri.h; every function below keeps the signature and the observable behaviour of its namesake in fri.cpp
// (returned values, the `new_size` proof-size accounting, the globals vpd_verifier.cpp reads afterwards), but the codewords, the
// Merkle trees and the folds live in HBM:
//
//   fri::request_init_commit            (fri.cpp:36-139)   -> the root vp_commit_private / vp_commit_public returned (prover_vpgpu.cpp)
//   fri::commit_phase_step              (fri.cpp:289-424)  -> vp_fri_step
//   fri::commit_phase_final             (fri.cpp:426-431)  -> vp_fri_final (+ the mask codeword through vp_fri_open) into fri::cpd, the
//                                                             object vpd_verifier.cpp:309-324 reads the last codeword from
//   fri::request_init_value_with_merkle (fri.cpp:148-205)  -> vp_fri_open(oracle 0 | 1)
//   fri::request_step_commit            (fri.cpp:229-287)  -> vp_fri_open(oracle 2 + level)
//
// oracle/Makefile (target `integration`) links this file INSTEAD of fri.cpp into oracle/_ref/ref_run_vpgpu; vpd_verifier.cpp,
// poly_commit.cpp, RS_polynomial.cpp, fft_circuit_GKR.cpp, merkle_tree.cpp, fieldElement.cpp stay the reference's objects.
#include "fri.h"                    // the reference's, -I$(REF)/lib/virgo/src
#include "vpgpu_glue.h"
#include <cstring>
#include <cstdlib>
#include <unordered_set>
#include <vector>

namespace virgo {
// ---- the globals fri.h declares (fri.cpp:13-34).  Only the scalars and `cpd` carry state here; the arrays stay NULL: their
// contents are device memory now.
int fri::log_current_witness_size_per_slice, fri::current_step_no, fri::witness_bit_length_per_slice;
fri::commit_phase_data fri::cpd;
double fri::__fri_timer;
__hhash_digest *fri::witness_merkle[2];
fieldElement *fri::witness_rs_codeword_before_arrange[2][slice_number + 1];
fieldElement *fri::witness_rs_codeword_interleaved[2];
int *fri::witness_rs_mapping[2][slice_number + 1];
fieldElement *fri::L_group;
bool *fri::visited[max_bit_length];
bool *fri::visited_init[2];
bool *fri::visited_witness[2];
__hhash_digest *fri::leaf_hash[2];
fieldElement *fri::r_extended;
fieldElement *fri::virtual_oracle_witness, *fri::virtual_oracle_witness_msk;
int *fri::virtual_oracle_witness_mapping, *fri::virtual_oracle_witness_msk_mapping;
}  // namespace virgo

using virgo::fieldElement;
using virgo::__hhash_digest;
namespace fri = virgo::fri;

static_assert(sizeof(fieldElement) == sizeof(vp_F), "virgo::fieldElement is two u64 limbs (fieldElement.hpp:96-97)");
static_assert(sizeof(__hhash_digest) == 32, "digest = 32 bytes (my_hhash.h)");

namespace {
struct OracleState { bool have = false; int bit_len = 0; unsigned char root[32]; };
OracleState g_oracle[2];
int g_log_init = 0;                                     // log2 of a slice's codeword length at commit time (bit_len + rs_code_rate - log_slice_number)
// the reference's `visited_*` bool arrays (proof-size accounting only): sparse here, same semantics
std::unordered_set<unsigned long long> g_vis_witness[2], g_vis_init[2], g_vis_step[virgo::max_bit_length];
std::vector<fieldElement> g_fri_r;                      // challenges of the steps, for VPI_DUMP_FRI

bool mark(std::unordered_set<unsigned long long> &s, unsigned long long k) { return s.insert(k).second; }    // true if it was NOT visited before
}  // namespace

void vpi_oracle_committed(int oracle, int bit_len, const unsigned char root[32]) {
    if (oracle < 0 || oracle > 1) { fprintf(stderr, "vpgpu glue: oracle %d\n", oracle); exit(EXIT_FAILURE); }
    g_oracle[oracle].have = true; g_oracle[oracle].bit_len = bit_len; memcpy(g_oracle[oracle].root, root, 32);
    // what fri::request_init_commit leaves behind for the commit phase (fri.cpp:44-49)
    fri::__fri_timer = 0;
    fri::current_step_no = 0;
    fri::log_current_witness_size_per_slice = bit_len + virgo::rs_code_rate - virgo::log_slice_number;
    fri::witness_bit_length_per_slice = bit_len - virgo::log_slice_number;
    g_log_init = fri::log_current_witness_size_per_slice;
    g_vis_witness[oracle].clear(); g_vis_init[oracle].clear();
    if (oracle == 0) for (auto &s : g_vis_step) s.clear();
}

// fri.cpp:36-139.  The encoding, the interleave, the 65-step leaf chains and the tree were done by vp_commit_private (oracle 0) /
// vp_commit_public (oracle 1) on the device; this returns the root they produced.  (Reached through vpd_prover_init, vpd_prover.cpp:9-13,
// if a caller still goes through poly_commit_prover::commit_private_array; prover_vpgpu.cpp calls the device directly.)
__hhash_digest fri::request_init_commit(const int bit_len, const int oracle_indicator) {
    if (oracle_indicator < 0 || oracle_indicator > 1 || !g_oracle[oracle_indicator].have || g_oracle[oracle_indicator].bit_len != bit_len) {
        fprintf(stderr, "vpgpu glue: request_init_commit(%d, %d) before the device committed that oracle\n", bit_len, oracle_indicator);
        exit(EXIT_FAILURE);
    }
    __hhash_digest d;
    memcpy(&d, g_oracle[oracle_indicator].root, 32);
    return d;
}

void fri::delete_self() { cpd.delete_self(); }

// fri.cpp:148-205: the leaf holding w^pow_0 and w^pow_1 = -w^pow_0 of every slice (+ the mask slice) of oracle l (0) or h (1), and its
// authentication path: com_hhash[k] = sibling at height k, com_hhash[depth] = the leaf digest.
std::pair<std::vector<std::pair<fieldElement, fieldElement> >, std::vector<__hhash_digest> >
fri::request_init_value_with_merkle(long long pow_0, long long pow_1, int &new_size, const int oracle_indicator) {
    if (pow_0 > pow_1) std::swap(pow_0, pow_1);
    const int log_leaf_size = log_slice_number + 1;
    const int depth = log_current_witness_size_per_slice - 1;
    if (pow_0 + (1LL << depth) != pow_1 || oracle_indicator < 0 || oracle_indicator > 1) {
        fprintf(stderr, "vpgpu glue: request_init_value_with_merkle(%lld, %lld)\n", pow_0, pow_1); exit(EXIT_FAILURE);
    }
    vp_F vals[130];
    std::vector<__hhash_digest> com_hhash(depth + 1);
    int len = 0;
    {
        vpi_rand_guard guard("vp_fri_open");
        vpi_must(vp_fri_open(vpi_ctx(), oracle_indicator, (uint64_t) pow_0, vals, reinterpret_cast<uint8_t *>(com_hhash.data()), (depth + 1) * 32, &len),
                 "vp_fri_open(l/h)");
    }
    ++g_vpi_count.open_init;
    if (len != depth + 1) { fprintf(stderr, "vpgpu glue: path of %d digests, expected %d\n", len, depth + 1); exit(EXIT_FAILURE); }
    std::vector<std::pair<fieldElement, fieldElement> > value(slice_number + 1);
    for (int i = 0; i <= slice_number; ++i) {
        memcpy(&value[i].first, &vals[2 * i], 16);
        memcpy(&value[i].second, &vals[2 * i + 1], 16);
    }
    // proof-size accounting exactly as fri.cpp:156-200
    new_size = 0;
    for (int i = 0; i < slice_number; ++i) {
        if (mark(g_vis_witness[oracle_indicator], (unsigned long long) pow_0 << log_leaf_size | i << 1 | 0)) new_size += sizeof(fieldElement);
        if (mark(g_vis_witness[oracle_indicator], (unsigned long long) pow_0 << log_leaf_size | i << 1 | 1)) new_size += sizeof(fieldElement);
    }
    unsigned long long pos = (unsigned long long) pow_0 + (1ULL << depth);
    for (int i = 0; i < depth; ++i) {
        if (!g_vis_init[oracle_indicator].count(pos ^ 1)) new_size += sizeof(__hhash_digest);
        g_vis_init[oracle_indicator].insert(pos);
        g_vis_init[oracle_indicator].insert(pos ^ 1);
        pos /= 2;
    }
    return std::make_pair(value, com_hhash);
}

// fri.cpp:229-287: leaf of FRI level `lvl` that holds position `pow` (the pair pow mod N/2, + N/2 of every slice), path bottom-up, leaf
// digest last.
std::pair<std::vector<std::pair<fieldElement, fieldElement> >, std::vector<__hhash_digest> >
fri::request_step_commit(int lvl, long long pow, int &new_size) {
    if (lvl < 0 || lvl >= current_step_no) { fprintf(stderr, "vpgpu glue: request_step_commit level %d of %d\n", lvl, current_step_no); exit(EXIT_FAILURE); }
    const int log_leaf_size = log_slice_number + 1;
    const unsigned long long n_leaves = (unsigned long long) cpd.merkle_size[lvl];          // = codeword length of the level / 2
    const unsigned long long leaf = (unsigned long long) pow % n_leaves;                    // rs_codeword_mapping[lvl][pow << 6 | i] >> 7 (fri.cpp:348-355)
    int depth = 0;
    while ((1ULL << depth) < n_leaves) ++depth;
    vp_F vals[130];
    std::vector<__hhash_digest> com_hhash(depth + 1);
    int len = 0;
    {
        vpi_rand_guard guard("vp_fri_open");
        vpi_must(vp_fri_open(vpi_ctx(), 2 + lvl, leaf, vals, reinterpret_cast<uint8_t *>(com_hhash.data()), (depth + 1) * 32, &len), "vp_fri_open(step)");
    }
    ++g_vpi_count.open_step;
    if (len != depth + 1) { fprintf(stderr, "vpgpu glue: path of %d digests, expected %d\n", len, depth + 1); exit(EXIT_FAILURE); }
    std::vector<std::pair<fieldElement, fieldElement> > value_vec(slice_number + 1);
    for (int i = 0; i <= slice_number; ++i) {
        memcpy(&value_vec[i].first, &vals[2 * i], 16);
        memcpy(&value_vec[i].second, &vals[2 * i + 1], 16);
    }
    // proof-size accounting as fri.cpp:251-283 (elements and tree nodes share ONE visited array there; elements are tested, never set)
    new_size = 0;
    bool visited_element = false;
    for (int i = 0; i < slice_number; ++i)
        if (g_vis_step[lvl].count(leaf << log_leaf_size | (unsigned long long) i << 1)) visited_element = true;
    if (!visited_element) new_size += sizeof(fieldElement);
    unsigned long long node = leaf + n_leaves;
    while (node != 1) {
        if (!g_vis_step[lvl].count(node ^ 1)) {
            new_size += sizeof(__hhash_digest);
            g_vis_step[lvl].insert(node ^ 1);
            g_vis_step[lvl].insert(node);
        }
        node /= 2;
    }
    return std::make_pair(value_vec, com_hhash);
}

// fri.cpp:289-424: fold every slice (and the mask) by r, re-interleave, hash the leaves, build the tree; the first call builds the
// virtual oracle from what vp_commit_public left in HBM (poly_commit.h:294-318).
__hhash_digest fri::commit_phase_step(fieldElement r) {
    const int nxt_witness_size = (1 << log_current_witness_size_per_slice) / 2;
    __hhash_digest root;
    vp_F rr; rr.real = r.real; rr.img = r.img;
    {
        vpi_rand_guard guard("vp_fri_step");
        vpi_stopwatch sw(current_step_no == 0 ? &g_vpi_sec.first_fri_step : &g_vpi_sec.fri_step);
        vpi_must(vp_fri_step(vpi_ctx(), &rr, reinterpret_cast<uint8_t *>(&root)), "vp_fri_step");
    }
    ++g_vpi_count.fri_step;
    if ((int) g_fri_r.size() > current_step_no) g_fri_r.resize(current_step_no);
    g_fri_r.push_back(r);
    if (FILE *f = vpi_dump_fri_file()) { fwrite(&rr, 16, 1, f); fwrite(&root, 32, 1, f); }
    g_vis_step[current_step_no].clear();
    cpd.merkle_size[current_step_no] = nxt_witness_size / 2;
    log_current_witness_size_per_slice--;
    ++current_step_no;
    return root;
}

// fri.cpp:426-431 returns cpd.rs_codeword[last]; vpd_verifier.cpp:309-324 then reads cpd.rs_codeword[last][j << 7 | i << 1] and
// cpd.rs_codeword_msk[last][j] directly.  Both are fetched from the device here: the 2048 values of the 64 slices by vp_fri_final, the 32
// values of the mask slice as the mask pairs of the level's 16 leaves (interleaved layout tmp[i << 1 | hi], fri.cpp:377-386).
fieldElement *fri::commit_phase_final() {
    if (current_step_no == 0) { fprintf(stderr, "vpgpu glue: commit_phase_final before any step\n"); exit(EXIT_FAILURE); }
    const int last = current_step_no - 1;
    const int n_final = (1 << rs_code_rate) * slice_number;              // 2048
    if (cpd.rs_codeword[last] == NULL) cpd.rs_codeword[last] = new fieldElement[n_final];
    if (cpd.rs_codeword_msk[last] == NULL) cpd.rs_codeword_msk[last] = new fieldElement[1 << rs_code_rate];
    vpi_rand_guard guard("vp_fri_final / vp_fri_open");
    vpi_stopwatch sw(&g_vpi_sec.fri_final);
    vpi_must(vp_fri_final(vpi_ctx(), reinterpret_cast<vp_F *>(cpd.rs_codeword[last])), "vp_fri_final");
    ++g_vpi_count.fri_final;
    for (int i = 0; i < (1 << rs_code_rate) / 2; ++i) {
        vp_F vals[130];
        unsigned char path[8 * 32];
        int len = 0;
        vpi_must(vp_fri_open(vpi_ctx(), 2 + last, (uint64_t) i, vals, path, sizeof path, &len), "vp_fri_open(final mask)");
        memcpy(&cpd.rs_codeword_msk[last][i << 1 | 0], &vals[128], 16);
        memcpy(&cpd.rs_codeword_msk[last][i << 1 | 1], &vals[129], 16);
    }
    if (FILE *f = vpi_dump_fri_file()) {
        fwrite(cpd.rs_codeword[last], 16, n_final, f);
        fwrite(cpd.rs_codeword_msk[last], 16, 1 << rs_code_rate, f);
        fflush(f);
    }
    return cpd.rs_codeword[last];
}
