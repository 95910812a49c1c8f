// libvpgpu.so — host side of the C ABI in include/vpgpu.h: HBM residency of the circuit, the
// bookkeeping tables and the verifier tape, and the launch sequence of the GKR sumcheck kernels
// (vp_kernels.h) on one HIP stream.  gfx950 only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <set>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/vpgpu.h"
#include "vp_kernels.h"

using namespace vp;

static_assert(sizeof(vp_F) == sizeof(F), "vp_F layout");

namespace {

struct Csr {                 // contributions of one layer sorted by target row (see vp_kernels.h K2/K3)
    u32 n_rows = 0, n_entries = 0, n_heavy = 0, n_chunks = 0, n_heavy_entries = 0;
    u32 *rowptr = nullptr, *e_g = nullptr, *e_x = nullptr;
    uint16_t *e_tl = nullptr;
    u32 *heavy_row = nullptr, *heavy_cptr = nullptr, *chunk_beg = nullptr, *chunk_end = nullptr;
    u32 *chunk_h = nullptr, *heavy_cnt = nullptr;      // heavy-row index of every chunk; arrival counters of the rows cut into several chunks (zero between proofs)
    std::vector<u32> h_heavy_row, h_heavy_cptr;      // host copies (a handful of rows): row-range init jobs of the index-split proof
};

struct LayerDev {
    u64 size = 0;
    int bl = 0;
    F *val = nullptr;                       // circuitValue[i]
    unsigned long long *valr = nullptr;     // real parts of val (all-real circuits: operand of the inner products), filled by vp_evaluate
    uint8_t *ty = nullptr; int16_t *gl = nullptr; u32 *gu = nullptr, *gv = nullptr;
    F *gc = nullptr;
    u32 n_assert = 0; u32 *assert_idx = nullptr;
    std::vector<u64> dad_size; std::vector<int> dad_bl; std::vector<u32 *> dad_id;   // j < layer index
    int max_dad_bl = -1;
    std::vector<u32> t_off, t_len;          // phase-2 table layout (slots)
    u32 p2_total = 0;
    u32 n_gather = 0; u32 *g_slot = nullptr, *g_idx = nullptr; uint8_t *g_layer = nullptr;
    Csr c1, c2;
    // Liu: jobs for the layer whose claims are combined on layer (this-1)
    u32 n_jobs = 0; BetaJob *jobs = nullptr; std::vector<int> job_k, job_h1;
    // batched path: per-slot V gather map, Liu gather lists, half tables of this layer's sumchecks
    uint8_t *s_layer = nullptr; u32 *s_idx = nullptr;
    u32 *lrow = nullptr, *l_g = nullptr; uint8_t *l_q = nullptr; u32 l_n = 0;      // l_n: entries of the Liu gather lists
    Half hg{}, hu{};
    Half *liu_H = nullptr;
    int job_hg = -1, job_hu = -1, job_liu0 = -1, job_liun = 0;      // this layer's entries of vp_ctx::all_jobs (eq half tables of its three sumchecks)
    // verifier-side predicates (vp_predicates): gates listed by bucket, pieces of <= 512
    u32 *p_idx = nullptr, *p_cbeg = nullptr, *p_cend = nullptr, *p_bptr = nullptr, *glv = nullptr; uint8_t *p_flag = nullptr;
    u32 p_chunks = 0, p_buckets = 0;
};

struct SumcheckState {
    int phase = 0, layer = 0, n_tab = 0, round = 0, total_rounds = 0, has_a = 1;
    u32 off[VP_MAX_TAB], len0[VP_MAX_TAB], valid0[VP_MAX_TAB];
    int bl[VP_MAX_TAB];
    const F *V0 = nullptr, *M0 = nullptr, *A0 = nullptr;     // round-1 sources
};

struct EvPair { hipEvent_t a, b; u64 bytes; int kind; u32 grid, jobs, rounds, first_round; u64 work; };

}  // namespace

struct PcShard;            // commitment sharded over ranks (vpgpu_pc_shard.inc)
struct FgkState;           // buffers of vp_fft_gkr (vpgpu_fftgkr.inc)
struct VpComm;             // RCCL communicator (vpgpu_pc_shard.inc)
// One in-order chain of the batched proof: its own stream and scratch, so that independent sumchecks
// (all of them, given the tape, except phase 1 -> phase 2 of the same layer) overlap on the device.
struct Lane {
    hipStream_t stream = nullptr;
    F *tab[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    F *part2 = nullptr, *chunk_part = nullptr, *Vu = nullptr, *dot_part = nullptr;
    hipEvent_t done = nullptr;
};

// Batched launch plan of the GKR part (see vp_kernels.h "Batched launches"): recorded once per circuit by running the
// per-sumcheck drivers in record mode (same code that launches directly on the lane path), then merged step by step.
enum { NK_LIGHT = 0, NK_CHUNKS, NK_COMBINE, NK_DOT, NK_DOTFIN, NK_SFGEN, NK_SF, NK_SEG, NK_EMIT, NK_COUNT };
struct PStep { int kind; u32 idx, grid, lds; u64 bytes; int rounds; int xchain; u64 work = 0; int first_round = 0, n_rounds = 0; };   // kind -1: placeholder step; xchain: also wait for that chain's latest node
struct PlanRec {
    std::vector<LightJob> light; std::vector<ChunkJob> chunks; std::vector<CombineJob> combine; std::vector<DotJob> dot;
    std::vector<SfArgs> sf; std::vector<SfGenJob> sfgen; std::vector<SegArgs> seg; std::vector<EmitArgs> emit;
    std::vector<std::vector<PStep>> chains; int cur = -1;
    void push(int kind, u32 idx, u32 grid, u32 lds = 0, u64 bytes = 0, u64 work = 0, int first_round = 0, int n_rounds = 0) {
        PStep st{kind, idx, grid, lds, bytes, 0, -1};
        st.work = work; st.first_round = first_round; st.n_rounds = n_rounds;
        chains[cur].push_back(st);
    }
};
struct PNode { int kind = 0, step = 0, stream = 0; u32 first = 0, count = 0, grid = 0, lds = 0, map_off = 0; u64 bytes = 0, work = 0; int first_round = 0, n_rounds = 0;
               std::vector<int> deps; hipEvent_t ev = nullptr; bool record = false; bool p2 = false; };     // p2: every job belongs to a phase-2 chain
struct Plan {
    std::vector<PNode> nodes;
    DotJob *d_dot = nullptr, *d_dotfin = nullptr; LightJob *d_light = nullptr; ChunkJob *d_chunks = nullptr; CombineJob *d_combine = nullptr;
    SfGenJob *d_sfgen = nullptr; SfArgs *d_sf = nullptr; SegArgs *d_seg = nullptr; EmitArgs *d_emit = nullptr; BlkMap *d_map = nullptr;
    hipStream_t streams[4] = {nullptr, nullptr, nullptr, nullptr};   // [0] = ctx->stream; [1..3] owned: fold (low priority), seg, emit (high)
    hipEvent_t ev_root = nullptr, ev_join[4] = {nullptr, nullptr, nullptr, nullptr};
    u64 rounds = 0; int n_steps = 0;
    int sf3c = 0;                                    // the fold nodes were sized for k_sumfold3c (a wave per chunk): fixed when the plan is recorded
    FixJob *d_fix = nullptr; u32 n_fix = 0;          // k_fixup jobs (when round 1 of the sumchecks leaves its b to the fix-up pass)
    std::vector<void *> allocs;                      // device arrays owned by the plan (freed with it)
    // index-split proof: V_u of the split phase-2 chains ahead of the graph (vu_pre_*): per chain one inner-product job over this rank's share of
    // the previous layer's values (d_vu_slice) or over all of it (d_vu_whole: no exchange available), results side by side in d_vu_sum (one small
    // all-reduce), k_vu_place reduces them mod p into the slots the phase-2 inits read (d_vu_dst)
    std::vector<int> vu_layers;
    DotJob *d_vu_slice = nullptr, *d_vu_whole = nullptr; BlkMap *d_vu_map = nullptr; u32 vu_grid = 0;
    F *d_vu_sum = nullptr; F **d_vu_dst = nullptr;
};

// Every switch of the library, public options (include/vpgpu.h) and internal tuning alike: resolved once per vp_create (defaults <- the caller's vp_options <-
// VP_* environment variables, the latter for tests / benches / A-B runs only).  The internal fields, by the name vp_tuning_get knows them under:
enum { VP_PATH_PLAN = 0, VP_PATH_LANES = 1, VP_PATH_SIMPLE = 3 };
struct VpOpt {
    int32_t gkr_path;               // VP_GKR_PATH=plan|lanes|simple (lanes / simple: -DVP_TEST_DRIVERS builds only)                                     [plan]
    int32_t use_graph;              // public
    int32_t serial;                 // VP_GKR_SERIAL: all chains on one stream (profiling)                                                               [0]
    int32_t fuse_init;              // VP_FUSE_INIT: phase-1 / Liu init inside the first fold launch of large tables                                     [1]
    int32_t fuse_min_log;           // VP_FUSE_MIN_LOG: ... from 2^this entries on; 0 = clamp(largest layer's bit length - 1, 20, 22)                    [0]
    int32_t fuse_dot;               // VP_FUSE_DOT: V_u rides on the fused launch                                                                        [0]
    int32_t drop_y;                 // VP_DROP_Y: rounds >= 2 derive b from the previous claim (five products per pair)                                  [1]
    int32_t drop_y_round1;          // VP_DROP_Y1: round 1 too, restored by k_fixup                                                                      [0]
    int32_t real_values;            // public
    int32_t seg_tiny;               // VP_SEG_TINY: tables <= 2^e entries are folded by the first k_seg launch                                           [1]
    int32_t sf_big_log;             // VP_SF_BIG_LOG: fold kernel from 2^this entries on (plan path)                                                     [14]
    int32_t sf3b_grid;              // VP_SF3B_GRID: its workgroups per launch                                                                           [512]
    int32_t dot_blocks;             // VP_DOT_BLOCKS: workgroups of a stand-alone inner product                                                          [1024]
    int32_t plan_align;             // VP_PLAN_ALIGN=left|right: 0 closing launches aligned at the end, 1 all left, 2 all right                          [0]
    int32_t xcd_map;                // VP_XCD_MAP: XCD-aware block map of the plan nodes (measured: no gain)                                             [0]
    int32_t round_fused_max;        // VP_ROUND_FUSED_MAX: interactive rounds with at most this many pairs take one launch                               [512]
    int32_t persistent_rounds, poll, debug, prefetch_round1, split_cost_percent;          // public
    int32_t kernel_copies;          // VP_KERNEL_COPIES: tape in / transcript out by two small kernels on pinned memory instead of copy-engine commands  [1]
    int32_t fold_branches;          // VP_FOLD_BRANCHES: an independent fold node of the plan runs on another stream                                     [1]
    int32_t ntt_scatter;            // VP_NTT_SCATTER: radix-4 long transforms store in natural order themselves                                         [1]
    int32_t fuse_combine;           // VP_FUSE_COMBINE: heavy rows finished by the chunk launch (2: and the chain keeps the empty step)                   [2]
    int32_t plan_autotune, pc_tensor_pub, persistent_timeout_ms;                          // public
    int32_t graph_explicit;         // VP_GRAPH_EXPLICIT: the plan's hipGraph built node by node (1..3: forms, see plan_graph_explicit)                   [0]
    int32_t ntt_r8;                 // VP_NTT_R8: transforms of 2^13..2^17 points by the radix-8 Stockham pair (0: radix-4 pair, the cross-check)        [1]
    int32_t fri_vo_fused;           // VP_FRI_VO_FUSED: first FRI fold straight from the committed codewords                                              [1]
    int32_t interactive_fast_init;  // public
    int32_t fuse_p2;                // VP_FUSE_P2: phase-2 init inside the first fold launch too                                                          [1]
    int32_t leaf_asm;               // VP_LEAF_ASM: leaf chains by the generated fixed-register block (0: the compiler's Keccak-f, the cross-check)       [1]
    int32_t real_pairs;             // public
    int32_t fft_gkr_batched;        // VP_FFT_GKR_BATCHED: the 2 lg inverse-FFT sumchecks of vp_fft_gkr as one batch                                      [1]
    int32_t fri_fold3;              // VP_FRI_FOLD3: vp_fri_commit folds levels 0, 1, 2 in one pass                                                        [1]
    int32_t sf3c;                   // VP_SF3C: the fold launches of the plan give a WAVE a chunk (k_sumfold3c: permlane swaps, no LDS, no barriers; measured slower) [0]
    int32_t split_vu;               // VP_SPLIT_VU: index-split proof: V_u of a split phase 2 from per-rank partial inner products ahead of the graph      [1]
};

struct vp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err = "";
    int n_layers = 0, max_bl = 0;
    std::vector<LayerDev> L;
    F **d_vals = nullptr;
    unsigned long long **d_valsr = nullptr;      // per layer: the real parts of its values as a dense array (all-real circuits, vp_evaluate), else nullptr
    F *beta_g = nullptr, *beta_u = nullptr, *bf = nullptr, *bs = nullptr, *liu_half = nullptr;
    u32 half_cap = 0;
    F *tab[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    u32 cap = 0;
    F *partials = nullptr;
    unsigned int *round_arrivals = nullptr;       // k_round_main: workgroups that have left their sums (zero between rounds)
    F *small = nullptr;          // [0]=0 [1]=1 [2]=add_term [3]=V_u [4..27]=coef [32..95]=scalarV [96..]=chunk partials ptr elsewhere
    F *chunk_part = nullptr; u32 chunk_cap = 0;
    F *d_tape = nullptr; u64 n_tape = 0;
    F *d_tr = nullptr; u64 n_tr = 0;      // transcript in F units
    F *h_pin = nullptr;                   // pinned: [0..2] poly, [3] vres, [4..4+64) claims
    unsigned long long *h_seq = nullptr, seq = 0;     // pinned ticket the closing kernels of the per-round path publish
    // persistent round kernel of the interactive path (vp_kernels_persist.h): mailbox in pinned host memory
    TailMail *h_req = nullptr; TailReply *h_rep = nullptr; std::atomic<bool> tail_active{false}; unsigned long long tail_seq = 0; int tail_enabled = 1;
    // suspend / resume of the resident round kernel (vp_kernels_persist.h): its last launch arguments, the device buffer it saves a phase
    // into, and whether such a saved phase is waiting for the next vp_round / vp_finalize
    PTailArgs tail_args{}; F *tail_save = nullptr; bool tail_suspended = false, tail_lost = false; u64 tail_resumes = 0;
    TailAux *h_aux = nullptr;            // pinned: per-table data of the resident kernel's launch
    int poll = 1;                                     // VP_POLL=0: wait with hipStreamSynchronize instead
    F *h_io = nullptr; size_t h_io_cap = 0;   // pinned staging of the batched path: tape in, transcript out
    int r1_pending = 0;                // interactive path: round 1 of the phase was queued behind its init (1: in the resident kernel, 2: per-round launch,
                                       // 3: collected into r1_stash when the resident kernel was suspended before the first vp_round)
    vp_F r1_stash[3];
    F *h_stage = nullptr; u32 stage_at = 0;         // pinned ring the challenges are staged through (an init call no longer waits for its copies)
    VpOpt opt{};                       // resolved at vp_create: defaults <- caller's vp_options <- VP_* environment (test-only override)
    int *d_flag = nullptr;
    u32 *d_vcplx = nullptr;            // non-zero: some circuit value of the last vp_evaluate has an imaginary part
    int vreal = 0, plan_vreal = 0;     // 1: every circuit value is real — round 1 of every sumcheck and the phase-1 inits take the half-price products (vp_field.h, f_mad31c_rb)
    bool evaluated = false;
    SumcheckState sc;
    // tape / transcript layout
    std::vector<u64> ru_off, as_off, rv_off, sig_off, rliu_off;
    // stats
    int profiling = 0;
    vp_stats st{};
    std::vector<EvPair> ev_pool; size_t ev_used = 0;
    std::vector<vp_launch_stat> lstats;   // per-launch table of the last profiled call (vp_get_launch_stats)
    std::vector<vp_round_stat> rlog;      // interactive path: one entry per vp_round since the last vp_vres (vp_get_round_stats)
    int rlog_round = 0;                   // vp_round calls since the last phase init
    u64 rec_gen_bytes = 0, rec_gen_work = 0;      // record mode: algorithmic bytes / contributions of the init fused into the next fold launch
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<void *> allocs;
    BetaJob *all_jobs = nullptr; u32 n_all_jobs = 0, beta_bpj = 1; F *half_pool = nullptr;
    // polynomial commitment
    F *pc_rt = nullptr, *pc_coef = nullptr, *pc_cw = nullptr; Dig *pc_tree = nullptr; int pc_lm = -1; double commit_ms = 0;
    std::map<std::pair<const void *, int>, F *> pc_rtc;   // compact root tables of smaller orders, keyed by (source table, log2 order)
    F *pc_q0 = nullptr, *pc_eq = nullptr, *pc_cbuf = nullptr; int pc_cbuf_lm = -1; int *pc_flag = nullptr; bool pc_q_tensor = false;   // tensor public vector: its one encoded slice (commit_public)
    F *pc_pub = nullptr, *pc_qcw = nullptr, *pc_hcw = nullptr, *pc_tmp = nullptr, *pc_small = nullptr; Dig *pc_tree_h = nullptr; bool pc_private_done = false;
    F *pc_scr = nullptr; size_t pc_scr_cap = 0;
    F *pc_fri_all = nullptr; std::vector<size_t> fri_cw_off, fri_tree_off; F *pc_open_buf = nullptr;
    Dig *pc_fri_roots = nullptr;
    F *pc_fri[2] = {nullptr, nullptr}; Dig *pc_fri_tree = nullptr; int fri_step = -1; size_t fri_tree_used = 0; bool pc_public_done = false;
    // the mask slice with content (vp_commit_private_masked / vp_commit_public_masked; 0 = the protocol's zero mask, nothing below is touched): padded mask length,
    // the slice's l / q / h codewords and its FRI levels end to end (M elements each, coset-major), scratch of its small transforms, per-level offsets
    u32 pc_mask_ms = 0; F *pc_lm_cw = nullptr, *pc_qm_cw = nullptr, *pc_hm_cw = nullptr, *pc_fm = nullptr, *pc_mtmp = nullptr; std::vector<size_t> fri_m_off;
    size_t pc_mtmp_cap = 0, pc_mB = 0;                 // scratch capacity; B = max(ms, N): the scratch is laid out in blocks of B elements (pc_mask_scratch)

    // Deferred completion (vp_set_deferred / vp_flush; round 5).  A GPU that goes idle for a fraction of a millisecond — a host synchronisation between two
    // prover calls, the host work between two proofs — runs the NEXT ten milliseconds at a lower clock (tools/leaf_in_step.py: k_leaf_hash alone 10.1 ms
    // back to back, 10.5 ms behind 0.5 ms of idling, 11.7 ms behind 5 ms).  In deferred mode vp_commit_private, vp_prove_gkr (launch plan, unsharded),
    // vp_commit_public_eq, vp_fri_commit and vp_fri_final queue their launches, stage their results in pinned memory, record what is left to do once the
    // stream has got there, and return; vp_flush waits and finishes them in order (copies into the caller's buffers, device times).  Every other entry point
    // flushes first.
    int deferred = 0;
    struct Pending { hipEvent_t a, b; int phase; std::function<int(float)> fin; };
    std::deque<Pending> pending;
    std::vector<hipEvent_t> ev_spare;
    unsigned char *h_ring = nullptr; size_t ring_cap = 0, ring_at = 0;      // pinned ring: results on their way out, small arrays on their way in
    size_t ring_live = 0, ring_call = 0;                                    // bytes behind ring_at that queued calls (ring_live) / the running entry point (ring_call) still use
    double phase_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};                         // device time of the last call of each kind (VP_PH_*)
    std::string plan_cache_path;                                            // VP_PLAN_CACHE at vp_create: the plan tuner's choices, kept across processes (one line per plan shape)
    bool pc_dry = false;                                                    // vp_warm: the transform helpers set up their tables and scratch and launch nothing
    unsigned char *h_pub = nullptr; size_t h_pub_cap = 0;                   // vp_warm: pinned staging of vp_commit_public's vector (uploaded from here in chunks)
    u64 private_epoch = 0;                                                  // vp_commit_private calls queued on this context so far

    PcShard *pcs = nullptr;              // non-null while the commitment is sharded over ranks (vp_pc_set_shard, world > 1)
    FgkState *fgk = nullptr;             // vp_fft_gkr: circuit layers and sumcheck tables of the last size used
    VpComm *cm = nullptr;                // RCCL communicator (vp_comm_init)
    F *part2 = nullptr;                  // [32][MAX_BLOCKS*3] block partials of the batched path
    int simple_path = 0, serial = 0;
    Lane lane0; Lane *ln = nullptr;
    std::vector<Lane> lanes;          // [2*(i-1)] = phases 1+2 of layer i, [2*(i-1)+1] = Liu of layer i
    std::vector<hipStream_t> lane_streams; std::vector<hipEvent_t> lane_events; hipEvent_t ev_fork = nullptr;
    F *pred_r = nullptr, *pred_pool = nullptr, *pred_part = nullptr, *pred_out = nullptr; BetaJob *pred_jobs = nullptr; u32 pred_bpj = 1;
    DotJob *pred_dot = nullptr; BlkMap *pred_map = nullptr;   // vp_predicates scratch
    PlanRec *rec = nullptr;           // non-null while the drivers run in record mode
    DotJob rec_dot{};                 // record mode: the V_u inner product the next phase-1 init job carries
    SfGenJob rec_gen{};               // record mode: init to be fused into the first fold launch of the next sumcheck (mode != 0)
    int fuse_init = 1;                // VP_FUSE_INIT=0: separate init launches for every sumcheck
                                      // Measured (profiles/r02_b_*): stand-alone equal (x64 278 vs 277 us, x1024 1.54 vs 1.56 ms: the init launches are
                                      // bound by writing the mult/add tables and the operand gathers, not by lane divergence), fused into the fold
                                      // launch slower (x1024 4.7 vs 3.5 ms: two more barriers per chunk at 3 workgroups per CU) -> off by default
    Plan *plan = nullptr; int plan_path = 1;   // VP_GKR_PATH=lanes: one stream per sumcheck chain instead of the plan
    // hipGraph of the concurrent GKR submission (per circuit; VP_GKR_GRAPH=0 submits the launches directly)
    std::recursive_mutex mu;         // held by every entry point for the duration of its call (CtxLock)
    bool plan_tuned = false;         // the plan layouts have been tried on this circuit (plan_autotune)
    bool plan_tune_cached = false;   // ... or taken from the process-wide table of an earlier context with the same plan shape
    int leaf_attr_done = 0;          // leaf-hash kernels' dynamic-LDS attribute set on this context's device (1) / failed (-1)
    uint32_t opt_pinned = 0;         // tuner fields the caller (struct or environment) moved off their defaults: bit 0 fuse_combine, 1 fold_branches,
                                     // 2 plan_align, 3 fuse_min_log, 4 sf3b_grid, 5 graph_explicit — plan_autotune leaves those alone
    hipGraphExec_t gkr_graph = nullptr; int use_graph = 1; bool graph_failed = false; u64 graph_launches = 0, graph_rounds = 0;
    // one proof sharded over GPUs by sumcheck chain (vp_set_shard): this rank records and runs only the chains it owns
    int shard_rank = 0, shard_world = 1;
    // index-split proof (vp_set_shard_split): tables of at least 2^(split_lw + split_min_log) entries are cut into 2^split_lw slices, slice s
    // folded by rank s; the entries the slices end in are gathered through the export area behind the transcript and the last split_lw
    // rounds of those tables are finished on the host (split_finish)
    int split_lw = 0, split_min_log = 11;
    struct SplitTab { int j, bl; u32 slot0; };                                  // table j of the chain, log2 of its full length, first export slot (2^split_lw of them)
    struct SplitChain { u64 poly_pos = 0, claims_pos = 0, r_off = 0; int rounds = 0, has_a = 1, n_tab = 0, small_owner = 0; std::vector<SplitTab> tabs; };
    std::vector<SplitChain> split;     // per chain (index as chain_owner); tabs empty = not split
    F *d_trs = nullptr; u64 n_exp = 0; // transcript + export area in one buffer (one all-reduce), export slots
    F *tr_base() const { return (split_lw > 0 && d_trs) ? d_trs : d_tr; }
    std::vector<int> chain_owner;     // per chain of the plan (same indices as `lanes`, + 1 for Vres); empty = everything local
    std::vector<double> chain_cost;
    bool drop_round1 = false;         // plan being recorded: round 1 of every sumcheck also leaves out the product sum (k_fixup restores b)
    std::vector<FixJob> rec_fix;      // record mode: one job per sumcheck, in protocol order
    bool owned(int chain) const { return shard_world <= 1 || chain_owner.empty() || chain_owner[chain] == shard_rank; }
    bool vu_supplied = false; float vu_pre_ms = 0; u64 vu_tape_tag = 0;     // index-split proof, caller-side exchange: vp_shard_vu_set has handed in the summed V_u for the next proof

    F *zero() const { return small; }
    F *one() const { return small + 1; }
    F *add_term() const { return small + 2; }
    F *Vu() const { return small + 3; }
    F *coef() const { return small + 4; }
    F *scalarV() const { return small + 32; }
};

namespace {

#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) {                                                                     \
            ctx->err = std::string(#x) + ": " + hipGetErrorString(e_);                              \
            return VP_EHIP;                                                                         \
        }                                                                                           \
    } while (0)
#define VPCHK(x) do { int r_ = (x); if (r_ != VP_OK) return r_; } while (0)
// every entry point except vp_round / vp_finalize: select the device and, if the persistent round kernel of the interactive path is
// still resident (the caller abandoned a sumcheck), tell it to leave — work submitted to the stream would otherwise queue behind it
int tail_quit(vp_ctx *ctx);
// ... and the resident kernels of the OTHER contexts of this process are suspended (phase saved, resumed by their next vp_round): device-wide
// synchronising HIP calls — hipMalloc / hipFree of an upload, a destroy — would otherwise wait behind them for up to their time-out.
void vp_suspend_others(vp_ctx *ctx);
// Threads.  Every entry point holds its context's lock for the duration of the call: calls on ONE context are serialised, calls on DIFFERENT
// contexts run concurrently (two proofs in flight from two threads).  The outermost entry point of a thread that may synchronise the device first
// suspends the other contexts' resident kernels — BEFORE it takes its own lock, and taking each other context's lock only while it suspends it
// (so it waits for that context's current call to return): no thread ever holds one context's lock while it waits for another's or for the
// registry's, which is what rules out a cycle.
static thread_local int tl_entry_depth = 0;
struct CtxLock {
    vp_ctx *c;
    CtxLock(vp_ctx *ctx, bool suspend_others) : c(ctx) {
        if (suspend_others && tl_entry_depth == 0) vp_suspend_others(ctx);
        c->mu.lock(); ++tl_entry_depth;
    }
    ~CtxLock() { --tl_entry_depth; c->mu.unlock(); }
    CtxLock(const CtxLock &) = delete; CtxLock &operator=(const CtxLock &) = delete;
};
#define VP_LOCK(ctx) CtxLock vp_ctx_lock_(ctx, false)
int flush_pending(vp_ctx *ctx, size_t count);
// VP_ENTER_Q: the entry points that can leave their completion pending (deferred mode); everything else finishes what is pending first
#define VP_ENTER_Q(ctx) CtxLock vp_ctx_lock_(ctx, true); do { HIPCHK(hipSetDevice((ctx)->device)); if ((ctx)->tail_active) (void) tail_quit(ctx); (ctx)->tail_suspended = false; (ctx)->tail_lost = false; (ctx)->ring_call = 0; } while (0)
#define VP_ENTER(ctx) VP_ENTER_Q(ctx); do { if (!(ctx)->pending.empty()) VPCHK(flush_pending((ctx), (size_t) -1)); } while (0)

constexpr u32 MAX_BLOCKS = 2048;     // 256 CUs x 8 resident 256-thread blocks

template <class T>
int dalloc(vp_ctx *ctx, T **p, size_t n) {
    *p = nullptr;
    if (n == 0) n = 1;
    HIPCHK(hipMalloc((void **) p, n * sizeof(T)));
    ctx->allocs.push_back((void *) *p);
    return VP_OK;
}
template <class T>
int dupload(vp_ctx *ctx, T **p, const std::vector<T> &h) {
    VPCHK(dalloc(ctx, p, h.size()));
    if (!h.empty()) HIPCHK(hipMemcpy(*p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return VP_OK;
}
void free_all(vp_ctx *ctx) {
    for (void *p : ctx->allocs) (void) hipFree(p);
    ctx->allocs.clear();
}

inline u32 nblk(u64 n) { return (u32) ((n + VP_BLOCK - 1) / VP_BLOCK); }
inline u32 grid_for(u64 n) { return std::max<u32>(1, std::min<u32>(nblk(n), MAX_BLOCKS)); }
inline int ceil_log2(u64 x) { int b = 0; while ((1ull << b) < x) ++b; return b; }
inline bool is_unary(int ty) { return ty == VP_ADDC || ty == VP_MULC || ty == VP_COPY || ty == VP_NOT; }

void count_launch(vp_ctx *ctx) { ++ctx->st.launches; }

// Profiled calls bracket each launch with a pair of events from the pool (on the stream the launch goes to) and collect
// the table afterwards.  prof_begin returns the pool slot or -1 (not profiling / pool exhausted).
int prof_begin(vp_ctx *ctx, hipStream_t st, int kind, u32 grid, u32 jobs, u64 bytes, u64 work, u32 rounds = 0, u32 first_round = 0) {
    if (!ctx->profiling || ctx->ev_used >= ctx->ev_pool.size()) return -1;
    EvPair &e = ctx->ev_pool[ctx->ev_used];
    e.kind = kind; e.grid = grid; e.jobs = jobs; e.bytes = bytes; e.work = work; e.rounds = rounds; e.first_round = first_round;
    hipEventRecord(e.a, st);
    return (int) ctx->ev_used++;
}
void prof_end(vp_ctx *ctx, hipStream_t st, int slot) { if (slot >= 0) hipEventRecord(ctx->ev_pool[slot].b, st); }
// after the stream has been waited for: durations of the bracketed launches -> ctx->lstats
void prof_collect(vp_ctx *ctx) {
    ctx->lstats.clear();
    for (size_t e = 0; e < ctx->ev_used; ++e) {
        const EvPair &p = ctx->ev_pool[e];
        float t = 0;
        if (hipEventElapsedTime(&t, p.a, p.b) != hipSuccess) t = 0;
        vp_launch_stat s{};
        s.kind = p.kind; s.step = (int) e; s.workgroups = p.grid; s.jobs = p.jobs; s.rounds = p.rounds; s.first_round = p.first_round;
        s.bytes = p.bytes; s.work = p.work; s.us = 1e3 * (double) t;
        ctx->lstats.push_back(s);
    }
}

// ---- internal phase drivers (all arguments already on the device tape) -----------------------------
int run_beta_half(vp_ctx *ctx, const F *r, int n, const F *init) {
    hipLaunchKernelGGL(k_beta_half, dim3(1), dim3(VP_BLOCK), 0, ctx->stream, r, n, init, ctx->bf, ctx->bs);
    count_launch(ctx);
    return VP_OK;
}

template <int PHASE>
int run_init_rows(vp_ctx *ctx, const Csr &c, InitArgs &a) {
    a.rowptr = c.rowptr; a.e_g = c.e_g; a.e_x = c.e_x; a.e_tl = c.e_tl; a.n_rows = c.n_rows;
    if (c.n_rows) {
        hipLaunchKernelGGL(k_init_light<PHASE>, dim3(nblk(c.n_rows)), dim3(VP_BLOCK), 0, ctx->stream, a);
        count_launch(ctx);
    }
    if (c.n_chunks) {
        hipLaunchKernelGGL(k_init_chunks<PHASE>, dim3((c.n_chunks + 3) / 4), dim3(VP_BLOCK), 0, ctx->stream, a,
                           c.chunk_beg, c.chunk_end, c.n_chunks, ctx->chunk_part);
        hipLaunchKernelGGL(k_init_combine, dim3((c.n_heavy + 3) / 4), dim3(VP_BLOCK), 0, ctx->stream, c.heavy_row,
                           c.heavy_cptr, c.n_heavy, ctx->chunk_part, a.M, a.A);
        count_launch(ctx); count_launch(ctx);
    }
    return VP_OK;
}

int do_phase1_init(vp_ctx *ctx, int i, const F *d_rliu, const F *d_assert) {
    LayerDev &cur = ctx->L[i], &pre = ctx->L[i - 1];
    VPCHK(run_beta_half(ctx, d_rliu, cur.bl, ctx->one()));
    hipLaunchKernelGGL(k_beta_expand, dim3(grid_for(cur.size)), dim3(VP_BLOCK), 0, ctx->stream, ctx->bf, ctx->bs,
                       cur.bl >> 1, (u32) cur.size, ctx->beta_g);
    count_launch(ctx);
    if (cur.n_assert) {
        hipLaunchKernelGGL(k_scale_entries, dim3(nblk(cur.n_assert)), dim3(VP_BLOCK), 0, ctx->stream, ctx->beta_g,
                           cur.assert_idx, cur.n_assert, d_assert);
        count_launch(ctx);
    }
    InitArgs a{};
    a.beta_g = ctx->beta_g; a.beta_u = nullptr; a.vals = ctx->d_vals; a.gc = cur.gc; a.coef = nullptr;
    a.M = ctx->tab[0][1]; a.A = ctx->tab[0][2];
    VPCHK(run_init_rows<1>(ctx, cur.c1, a));
    SumcheckState &s = ctx->sc;
    s.phase = 1; s.layer = i; s.n_tab = 1; s.round = 0; s.total_rounds = pre.bl; s.has_a = 1;
    s.off[0] = 0; s.len0[0] = 1u << pre.bl; s.valid0[0] = (u32) pre.size; s.bl[0] = pre.bl;
    s.V0 = pre.val; s.M0 = ctx->tab[0][1]; s.A0 = ctx->tab[0][2];
    return VP_OK;
}

int do_phase2_init(vp_ctx *ctx, int i, const F *d_ru) {
    LayerDev &cur = ctx->L[i], &pre = ctx->L[i - 1];
    VPCHK(run_beta_half(ctx, d_ru, pre.bl, ctx->one()));
    hipLaunchKernelGGL(k_beta_expand, dim3(grid_for(pre.size)), dim3(VP_BLOCK), 0, ctx->stream, ctx->bf, ctx->bs,
                       pre.bl >> 1, (u32) pre.size, ctx->beta_u);
    hipLaunchKernelGGL(k_p2_coef, dim3(1), dim3(64), 0, ctx->stream, ctx->Vu(), ctx->coef());
    hipLaunchKernelGGL(k_zero_f, dim3(1), dim3(64), 0, ctx->stream, ctx->add_term(), 1u);
    count_launch(ctx); count_launch(ctx); count_launch(ctx);
    if (cur.n_gather) {
        hipLaunchKernelGGL(k_p2_gather_v, dim3(nblk(cur.n_gather)), dim3(VP_BLOCK), 0, ctx->stream, cur.g_slot,
                           cur.g_layer, cur.g_idx, cur.n_gather, ctx->d_vals, ctx->tab[0][0]);
        count_launch(ctx);
    }
    InitArgs a{};
    a.beta_g = ctx->beta_g; a.beta_u = ctx->beta_u; a.vals = ctx->d_vals; a.gc = cur.gc; a.coef = ctx->coef();
    a.M = ctx->tab[0][1]; a.A = ctx->tab[0][2];
    VPCHK(run_init_rows<2>(ctx, cur.c2, a));
    SumcheckState &s = ctx->sc;
    s.phase = 2; s.layer = i; s.n_tab = i; s.round = 0; s.total_rounds = cur.max_dad_bl; s.has_a = 1;
    for (int j = 0; j < i; ++j) {
        s.off[j] = cur.t_off[j]; s.len0[j] = cur.t_len[j];
        s.valid0[j] = (u32) cur.dad_size[j];
        s.bl[j] = cur.dad_size[j] ? cur.dad_bl[j] : 0;
    }
    s.V0 = ctx->tab[0][0]; s.M0 = ctx->tab[0][1]; s.A0 = ctx->tab[0][2];
    return VP_OK;
}

int do_liu_init(vp_ctx *ctx, int i) {
    LayerDev &cur = ctx->L[i], &pre = ctx->L[i - 1];
    hipLaunchKernelGGL(k_beta_half_multi, dim3(cur.n_jobs), dim3(VP_BLOCK), 0, ctx->stream, cur.jobs);
    hipLaunchKernelGGL(k_zero_f, dim3(1), dim3(64), 0, ctx->stream, ctx->add_term(), 1u);
    count_launch(ctx); count_launch(ctx);
    F *M = ctx->tab[0][1];
    const u32 hc = ctx->half_cap;
    hipLaunchKernelGGL(k_liu_first, dim3(grid_for(pre.size)), dim3(VP_BLOCK), 0, ctx->stream, ctx->liu_half,
                       ctx->liu_half + hc, cur.job_h1[0], (u32) pre.size, M);
    count_launch(ctx);
    for (u32 q = 1; q < cur.n_jobs; ++q) {
        const int k = cur.job_k[q];
        const LayerDev &Lk = ctx->L[k];
        const u32 n = (u32) Lk.dad_size[i - 1];
        hipLaunchKernelGGL(k_liu_scatter, dim3(grid_for(n)), dim3(VP_BLOCK), 0, ctx->stream,
                           ctx->liu_half + 2 * (size_t) q * hc, ctx->liu_half + (2 * (size_t) q + 1) * hc,
                           cur.job_h1[q], Lk.dad_id[i - 1], n, M);
        count_launch(ctx);
    }
    SumcheckState &s = ctx->sc;
    s.phase = 3; s.layer = i; s.n_tab = 1; s.round = 0; s.total_rounds = pre.bl; s.has_a = 0;
    s.off[0] = 0; s.len0[0] = 1u << pre.bl; s.valid0[0] = (u32) pre.size; s.bl[0] = pre.bl;
    s.V0 = pre.val; s.M0 = M; s.A0 = ctx->tab[0][2];
    return VP_OK;
}

// one sumcheck round; rp (device) or rv (by value) is the previous challenge
int check_stream(vp_ctx *ctx);
int vp_check_collect(vp_ctx *ctx);
// the same three inits through the batched path's kernels (vpgpu_batched.inc): closed-form eq half tables of THIS sumcheck only, eq values as
// products of two half-table entries (no expanded table), the V gather of phase 2 and the assert scaling inside the row kernel
int do_phase1_init_fast(vp_ctx *ctx, int i);
int do_phase2_init_fast(vp_ctx *ctx, int i);
int do_liu_init_fast(vp_ctx *ctx, int i);

// Per-round path: wait for the ticket the closing kernel publishes in pinned memory (a few microseconds sooner than the
// runtime's stream wait, once per round x ~700 rounds per proof); falls back to the stream wait after 2 s or when VP_POLL=0.
int wait_ticket(vp_ctx *ctx) {
    if (!ctx->poll) return check_stream(ctx);
    const unsigned long long want = ctx->seq;
    const auto t0 = std::chrono::steady_clock::now();
    for (u32 spin = 0;; ++spin) {
        if (__atomic_load_n(ctx->h_seq, __ATOMIC_ACQUIRE) == want) return VP_OK;
        if ((spin & 0xfff) == 0xfff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) return check_stream(ctx);
    }
}

int do_round(vp_ctx *ctx, const F *rp, const F &rv, F *poly_dev, F *poly_host) {
    SumcheckState &s = ctx->sc;
    const int k = s.round + 1;
    RoundArgs a{};
    a.rp = rp; a.rv = rv; a.n_tab = s.n_tab; a.fold = (k >= 2); a.has_a = s.has_a;
    if (k <= 2) { a.inV = s.V0; a.inM = s.M0; a.inA = s.A0; }
    else { F **t = ctx->tab[k & 1]; a.inV = t[0]; a.inM = t[1]; a.inA = t[2]; }       // out of round k-1
    { F **t = ctx->tab[(k + 1) & 1]; a.outV = t[0]; a.outM = t[1]; a.outA = t[2]; }
    u32 pairs = 0;
    u64 bytes = 0;
    for (int j = 0; j < s.n_tab; ++j) {
        TabDesc &t = a.t[j];
        t.off = s.off[j];
        t.pair_start = pairs;
        if (k == 1) {
            t.len_in = s.len0[j]; t.valid_in = s.valid0[j];
            if (t.len_in >= 2) { pairs += (t.valid_in + 1) >> 1; bytes += (u64) t.valid_in * (s.has_a ? 48 : 32); }
        } else {
            const int sh = k - 2;
            t.len_in = sh < 32 ? (s.len0[j] >> sh) : 0;
            if (t.len_in < 2) { t.len_in = 0; t.valid_in = 0; continue; }
            t.valid_in = (u32) (((u64) s.valid0[j] + (1ull << sh) - 1) >> sh);
            const u32 vo = (t.valid_in + 1) >> 1;
            if ((t.len_in >> 1) >= 2) { pairs += (vo + 1) >> 1; bytes += (u64) (t.valid_in + vo) * (s.has_a ? 48 : 32); }
        }
    }
    a.total_pairs = pairs;
    const u32 fused_max = (u32) ctx->opt.round_fused_max;
    if (pairs <= fused_max) {            // one workgroup: fold + sums + closing in a single launch
        hipLaunchKernelGGL(k_round_fused, dim3(1), dim3(VP_BLOCK), 0, ctx->stream, a, ctx->add_term(), ctx->scalarV(), poly_dev, poly_host,
                           (poly_host && ctx->poll) ? ctx->h_seq : nullptr, ++ctx->seq);
        count_launch(ctx);
        s.round = k;
        ++ctx->st.rounds;
        return VP_OK;
    }
    const RoundOut ro{ctx->add_term(), ctx->scalarV(), poly_dev, poly_host, (poly_host && ctx->poll) ? ctx->h_seq : nullptr, ++ctx->seq};
    if (pairs) {                         // the workgroup that finishes last closes the round (k_round_main)
        const u32 grid = grid_for(pairs);
        const bool big = pairs >= VP_BIG_PAIRS;
        const int sl = big ? prof_begin(ctx, ctx->stream, VP_K_ROUND, grid, 1, bytes, pairs, 1, (u32) k) : -1;
        if (big) hipLaunchKernelGGL(k_round_main<1>, dim3(grid), dim3(VP_BLOCK), 0, ctx->stream, a, ctx->partials, ctx->round_arrivals, ro);
        else hipLaunchKernelGGL(k_round_main<0>, dim3(grid), dim3(VP_BLOCK), 0, ctx->stream, a, ctx->partials, ctx->round_arrivals, ro);
        prof_end(ctx, ctx->stream, sl);
    } else {
        hipLaunchKernelGGL(k_round_final, dim3(1), dim3(VP_BLOCK), 0, ctx->stream, a, ro);
    }
    count_launch(ctx);
    s.round = k;
    ++ctx->st.rounds;
    return VP_OK;
}

int do_finalize(vp_ctx *ctx, const F *rp, const F &rv, F *claims_dev, F *claims_host) {
    SumcheckState &s = ctx->sc;
    FinArgs a{};
    a.rp = rp; a.rv = rv; a.n_tab = s.n_tab; a.rounds_done = s.round;
    const int R = s.round;
    a.curV = (R <= 1) ? s.V0 : ctx->tab[(R + 1) & 1][0];
    const int sh = R >= 1 ? R - 1 : 0;
    for (int j = 0; j < s.n_tab; ++j) {
        a.off[j] = s.off[j]; a.bl[j] = s.bl[j];
        a.valid[j] = (u32) (((u64) s.valid0[j] + (1ull << sh) - 1) >> sh);
    }
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(64), 0, ctx->stream, a, ctx->scalarV(), claims_dev, claims_host,
                       s.phase == 1 ? ctx->Vu() : nullptr, (claims_host && ctx->poll) ? ctx->h_seq : nullptr, ++ctx->seq);
    count_launch(ctx);
    return VP_OK;
}

// checked build: the first violated device check of the calls since the last collection -> an error of THIS call
int vp_check_collect(vp_ctx *ctx) {
#ifdef VP_CHECKED
    unsigned int e[4] = {0, 0, 0, 0};
    if (hipMemcpyFromSymbol(e, HIP_SYMBOL(g_vp_chk_err), sizeof e) != hipSuccess) return VP_OK;
    if (e[0]) {
        const unsigned int z[4] = {0, 0, 0, 0};
        (void) hipMemcpyToSymbol(HIP_SYMBOL(g_vp_chk_err), z, sizeof z);
        char buf[160];
        snprintf(buf, sizeof buf, "device check failed: site %u (%u, %u, %u) — see vp_check.h", e[0], e[1], e[2], e[3]);
        ctx->err = buf;
        return VP_EHIP;
    }
#endif
    (void) ctx;
    return VP_OK;
}
int check_stream(vp_ctx *ctx) {
    const hipError_t e1 = hipStreamSynchronize(ctx->stream), e2 = e1 == hipSuccess ? hipGetLastError() : e1;
    if (e2 != hipSuccess) {
        // a launch that did not finish may have left workgroups counted in k_round_main's arrival counter: a later round on this context must
        // not close early or never (best effort — after a device fault the memset fails too and every later call reports the fault)
        if (ctx->round_arrivals) (void) hipMemsetAsync(ctx->round_arrivals, 0, 64, ctx->stream);
        ctx->err = std::string("stream: ") + hipGetErrorString(e2);
        return VP_EHIP;
    }
    return vp_check_collect(ctx);
}

// ---- deferred completion (see vp_ctx::deferred) -----------------------------------------------------------------------
enum { VP_PH_PRIVATE = 0, VP_PH_GKR = 1, VP_PH_PUBLIC = 2, VP_PH_FRI = 3, VP_PH_FRI_FINAL = 4 };
hipEvent_t ev_take(vp_ctx *ctx) {
    if (!ctx->ev_spare.empty()) { hipEvent_t e = ctx->ev_spare.back(); ctx->ev_spare.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void) hipEventCreate(&e);
    return e;
}
// pinned bytes that stay untouched until the stream has consumed / produced them: a ring far larger than what two proofs in flight stage (< 100 KB each).
// The ring_live bytes behind ring_at belong to calls that are still queued (deferred completion) or to the running entry point (ring_call of them): a
// request that would run into them first finishes the queued calls in order and lets the stream take what this call has staged, so a caller that
// queues hundreds of calls without a vp_flush gets slower, never wrong bytes.
int ring_alloc(vp_ctx *ctx, size_t bytes, void **out) {
    if (!ctx->h_ring) {
        HIPCHK(hipHostMalloc((void **) &ctx->h_ring, (size_t) 8 << 20, hipHostMallocDefault));
        ctx->ring_cap = (size_t) 8 << 20; ctx->ring_at = 0; ctx->ring_live = ctx->ring_call = 0;
    }
    bytes = (bytes + 63) & ~(size_t) 63;
    if (bytes > ctx->ring_cap / 8) { ctx->err = "internal: staging request too large"; return VP_ELIMIT; }
    const bool wrap = ctx->ring_at + bytes > ctx->ring_cap;
    const size_t skip = wrap ? ctx->ring_cap - ctx->ring_at : 0;                                       // the tail that a wrap leaves unused (possibly none)
    if (ctx->ring_live + skip + bytes > ctx->ring_cap) {
        VPCHK(flush_pending(ctx, (size_t) -1));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        ctx->ring_live = ctx->ring_call;
        if (ctx->ring_live + skip + bytes > ctx->ring_cap) { ctx->err = "staging ring exhausted by one call"; return VP_ELIMIT; }
    }
    if (wrap) ctx->ring_at = 0;
    *out = ctx->h_ring + ctx->ring_at;
    ctx->ring_at += bytes;
    ctx->ring_live += skip + bytes; ctx->ring_call += skip + bytes;
    return VP_OK;
}
// returns the begin event of a deferred bracket to the pool when an entry point leaves early (HIPCHK / VPCHK between defer_begin and defer_end)
struct EvGuard { vp_ctx *ctx; hipEvent_t *ev; ~EvGuard() { if (ev && *ev) { ctx->ev_spare.push_back(*ev); *ev = nullptr; } } };
int defer_begin(vp_ctx *ctx, hipEvent_t *a) {
    *a = ev_take(ctx);
    if (!*a) { ctx->err = "hipEventCreate"; return VP_EHIP; }
    HIPCHK(hipEventRecord(*a, ctx->stream));
    return VP_OK;
}
// fin(ms): what the entry point does once its work is done (ms = device time between defer_begin and here).  Not deferred: waits and runs it now.
constexpr size_t VP_MAX_PENDING = 256;            // queued calls per context; one more finishes them all first
template <class Fn>
int defer_end(vp_ctx *ctx, hipEvent_t &a_ref, int phase, Fn fin) {
    const hipEvent_t a = a_ref;
    a_ref = nullptr;                               // the bracket owns it from here (EvGuard of the caller has nothing left to return)
    hipEvent_t b = ev_take(ctx);
    if (!b) { ctx->ev_spare.push_back(a); ctx->err = "hipEventCreate"; return VP_EHIP; }
    if (hipEventRecord(b, ctx->stream) != hipSuccess) { ctx->ev_spare.push_back(a); ctx->ev_spare.push_back(b); ctx->err = "hipEventRecord"; return VP_EHIP; }
    if (!ctx->deferred || ctx->profiling) {
        const int rc = check_stream(ctx);
        float ms = 0;
        if (rc == VP_OK) { if (ctx->profiling) prof_collect(ctx); (void) hipEventElapsedTime(&ms, a, b); }
        ctx->ev_spare.push_back(a); ctx->ev_spare.push_back(b);
        if (rc != VP_OK) return rc;
        ctx->phase_ms[phase] = ms;
        if (ctx->pending.empty()) ctx->ring_live = 0;             // the stream is idle: nothing staged is still in use once fin has copied
        const int rf = fin(ms);
        ctx->ring_call = 0;
        return rf;
    }
    if (ctx->pending.size() >= VP_MAX_PENDING) {
        const int rc = flush_pending(ctx, (size_t) -1);
        if (rc != VP_OK) { ctx->ev_spare.push_back(a); ctx->ev_spare.push_back(b); return rc; }
    }
    ctx->pending.push_back(vp_ctx::Pending{a, b, phase, std::function<int(float)>(fin)});
    return VP_OK;
}
// finish the first `count` pending calls in order ((size_t) -1: all).  After a failure the rest are dropped (their outputs stay unwritten).
int flush_pending(vp_ctx *ctx, size_t count) {
    int rc = VP_OK;
    while (count-- && !ctx->pending.empty()) {
        vp_ctx::Pending p = std::move(ctx->pending.front());
        ctx->pending.pop_front();
        const hipError_t e = hipEventSynchronize(p.b);          // (its own result only: the sticky last error may belong to a tolerated capture / instantiate failure)
        if (e != hipSuccess) { ctx->err = std::string("stream: ") + hipGetErrorString(e); rc = VP_EHIP; }
        if (rc == VP_OK) {
            float ms = 0;
            (void) hipEventElapsedTime(&ms, p.a, p.b);
            ctx->phase_ms[p.phase] = ms;
            rc = p.fin(ms);
        }
        ctx->ev_spare.push_back(p.a); ctx->ev_spare.push_back(p.b);
        if (rc != VP_OK) break;
    }
    if (rc != VP_OK) {
        (void) hipStreamSynchronize(ctx->stream);
        for (auto &p : ctx->pending) { ctx->ev_spare.push_back(p.a); ctx->ev_spare.push_back(p.b); }
        ctx->pending.clear();
        ctx->ring_live = ctx->ring_call;
        return rc;
    }
    if (ctx->pending.empty()) ctx->ring_live = ctx->ring_call;
    return ctx->pending.empty() ? vp_check_collect(ctx) : VP_OK;
}

// ---- persistent round kernel (vp_kernels_persist.h): host side of the mailbox ---------------------------------------
constexpr int VP_TAIL_SAVED = 1000;                           // internal: the kernel left with its phase saved (status 5)
int tail_wait(vp_ctx *ctx, unsigned long long want) {       // every reply word must carry the tag of message `want` (vp_kernels_persist.h)
    const auto t0 = std::chrono::steady_clock::now();
    const unsigned long long tag = VP_TAG(want), mask = 7ull << 61;
    for (u32 spin = 0;; ++spin) {
        bool all = true;
        for (int q = 0; q < 7; ++q) all &= (__atomic_load_n(&ctx->h_rep->w[q], __ATOMIC_RELAXED) & mask) == tag;
        if (all) { __atomic_thread_fence(__ATOMIC_ACQUIRE); return VP_OK; }
        if ((spin & 0xfff) == 0xfff) {
            if (__atomic_load_n(&ctx->h_rep->dead, __ATOMIC_ACQUIRE)) {
                if (__atomic_load_n(&ctx->h_rep->dead, __ATOMIC_ACQUIRE) == 2) return VP_TAIL_SAVED;     // timed out with the phase saved: the caller relaunches it
                ctx->tail_active = false; ctx->err = "persistent round kernel gave up waiting (a workgroup was not scheduled, or the distributed regime timed out)"; return VP_EHIP;
            }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(15)) { ctx->tail_active = false; ctx->err = "persistent round kernel did not answer"; return VP_EHIP; }
        }
    }
}
int tail_send(vp_ctx *ctx, int cmd, const F &r) {
    TailMail *m = ctx->h_req;
    const unsigned long long tag = VP_TAG(++ctx->tail_seq);
    __atomic_store_n(&m->w[1], r.im | tag, __ATOMIC_RELAXED);
    __atomic_store_n(&m->w[2], (unsigned long long) cmd | tag, __ATOMIC_RELAXED);
    __atomic_store_n(&m->w[0], r.re | tag, __ATOMIC_RELEASE);
    return tail_wait(ctx, ctx->tail_seq);
}
static inline unsigned long long tail_status(const vp_ctx *ctx) { return VP_UNTAG(ctx->h_rep->w[6]); }
static inline void tail_poly(const vp_ctx *ctx, vp_F out[3]) {
    for (int q = 0; q < 3; ++q) { out[q].real = VP_UNTAG(ctx->h_rep->w[2 * q]); out[q].img = VP_UNTAG(ctx->h_rep->w[2 * q + 1]); }
}
// leave the persistent kernel (a call other than vp_round / vp_finalize arrived while it was resident)
int tail_quit(vp_ctx *ctx) {
    if (!ctx->tail_active) return VP_OK;
    int rc = tail_send(ctx, 3, f_zero());
    if (rc == VP_TAIL_SAVED) rc = VP_OK;                      // it had already left by itself
    ctx->tail_active = false;
    (void) hipStreamSynchronize(ctx->stream);
    return rc;
}
// Another context needs the device: the resident kernel saves its phase and leaves; this context's next vp_round / vp_finalize resumes it.
int tail_suspend(vp_ctx *ctx) {
    if (!ctx->tail_active) return VP_OK;
    if (ctx->r1_pending == 1) {                               // round 1 was answered behind the init call and not collected yet: the quit's reply would overwrite it
        if (tail_wait(ctx, ctx->tail_seq) == VP_OK) { tail_poly(ctx, ctx->r1_stash); ctx->r1_pending = 3; ++ctx->sc.round; ++ctx->st.rounds; }
    }
    int rc = tail_send(ctx, 3, f_zero());
    if (rc == VP_TAIL_SAVED) rc = VP_OK;
    ctx->tail_active = false;
    ctx->tail_suspended = rc == VP_OK && tail_status(ctx) == 5;
    ctx->tail_lost = !ctx->tail_suspended;                    // left without saving (distributed regime): the phase cannot be continued
    (void) hipStreamSynchronize(ctx->stream);
    return rc;
}
// Relaunch the resident kernel on the phase it saved.  already_sent: the message the kernel is to answer next is already in the mailbox
// (the host sent it while the kernel was timing out) and carries ctx->tail_seq; otherwise the next tail_send will carry tail_seq + 1.
int tail_resume(vp_ctx *ctx, bool already_sent);
// tail_wait for callers that collect a reply: a kernel that timed out with its phase saved (VP_TAIL_SAVED, internal) is relaunched on the saved
// phase and waited for once more; the internal code never reaches the C ABI's caller.
int tail_wait_resumed(vp_ctx *ctx, unsigned long long want) {
    int rc = tail_wait(ctx, want);
    if (rc == VP_TAIL_SAVED) {
        rc = tail_resume(ctx, true);
        if (rc == VP_OK) rc = tail_wait(ctx, want);
        if (rc == VP_TAIL_SAVED) { ctx->err = "resident round kernel: timed out twice on one message"; rc = VP_EHIP; }
    }
    return rc;
}
int tail_resume(vp_ctx *ctx, bool already_sent) {
    PTailArgs a = ctx->tail_args;
    a.resume = 1;
    a.seq0 = already_sent ? ctx->tail_seq : ctx->tail_seq + 1;
    // the reply words still carry the leaving kernel's message with the tag the relaunched kernel will answer under: give them another one
    if (already_sent) for (int q = 0; q < 7; ++q) __atomic_store_n(&ctx->h_rep->w[q], VP_TAG(ctx->tail_seq + 1), __ATOMIC_RELAXED);
    __atomic_store_n(&ctx->h_rep->dead, 0ull, __ATOMIC_RELEASE);
    hipLaunchKernelGGL(k_phase, dim3(1), dim3(VP_PH_THREADS), (size_t) 3 * a.cap * sizeof(F), ctx->stream, a);
    { const hipError_t e = hipGetLastError(); if (e != hipSuccess) { ctx->err = std::string("k_phase relaunch failed: ") + hipGetErrorString(e); return VP_EHIP; } }
    count_launch(ctx);
    ctx->tail_active = true; ctx->tail_suspended = false; ++ctx->tail_resumes;
    return VP_OK;
}
// every live context of the process (vp_create .. vp_destroy).  Called without any context lock held (CtxLock): the registry lock, then one
// other context's lock at a time — the wait is for that context's current call to return.
static std::mutex g_ctx_mu;
static std::vector<vp_ctx *> g_ctxs;
void vp_suspend_others(vp_ctx *ctx) {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    bool any = false;
    for (vp_ctx *o : g_ctxs) {
        // a context without a resident kernel is nobody's obstacle: passed without its lock (two batched proofs on two threads must not take
        // turns).  Should its owner launch one a moment later, the device-wide call of this thread waits for that phase as it would have
        // before round 3 — at most until the kernel's own time-out suspends it.
        if (o == ctx || !o->tail_active.load(std::memory_order_acquire)) continue;
        std::lock_guard<std::recursive_mutex> lo(o->mu);
        if (o->tail_active) { (void) hipSetDevice(o->device); (void) tail_suspend(o); any = true; }
    }
    if (any) (void) hipSetDevice(ctx->device);
}

// Round k = s.round + 1 with previous challenge rv: if every live table of the phase now fits one CU, launch k_phase for this
// and all later messages of the phase.  Returns 1 when launched (the polynomial of round k is then in h_rep), 0 when not.
int tail_try_launch(vp_ctx *ctx, const F &rv) {
    SumcheckState &s = ctx->sc;
    const int k = s.round + 1;
    if (!ctx->tail_enabled || s.total_rounds - k < 1) return 0;
    PTailArgs a{};
    TailAux &ax = *ctx->h_aux;
    a.aux = ctx->h_aux;
    a.rv = rv; a.fold = (k >= 2) ? 1 : 0;
    if (k <= 2) { a.inV = s.V0; a.inM = s.M0; a.inA = s.A0; }
    else { F **t = ctx->tab[k & 1]; a.inV = t[0]; a.inM = t[1]; a.inA = t[2]; }
    u32 pairs = 0, ents = 0;
    for (int j = 0; j < s.n_tab; ++j) {
        TabDesc &t = ax.t[j];
        t.off = s.off[j]; t.pair_start = pairs;
        u32 len_out;
        if (k == 1) { t.len_in = s.len0[j]; t.valid_in = s.valid0[j]; len_out = t.len_in; }
        else {
            const int sh = k - 2;
            t.len_in = sh < 32 ? (s.len0[j] >> sh) : 0;
            if (t.len_in < 2) { t.len_in = 0; t.valid_in = 0; len_out = 0; }
            else { t.valid_in = (u32) (((u64) s.valid0[j] + (1ull << sh) - 1) >> sh); len_out = t.len_in >> 1; }
        }
        ax.len_out0[j] = len_out;
        ax.loff[j] = ents;
        ax.bl[j] = s.bl[j];
        if (len_out >= 2) { pairs += len_out >> 1; ents += len_out; }
    }
    if (pairs == 0 || pairs > VP_PH_PMAX) return 0;
    a.total_pairs = pairs;
    a.k0 = k; a.R = s.total_rounds; a.n_tab = s.n_tab; a.has_a = s.has_a; a.cap = ents;
    a.add_term = ctx->add_term(); a.scalarV = ctx->scalarV(); a.claims_dev = ctx->d_tr + ctx->n_tr + 3; a.poly_dev = ctx->d_tr + ctx->n_tr;
    a.Vu = s.phase == 1 ? ctx->Vu() : nullptr;
    a.req = ctx->h_req; a.rep = ctx->h_rep; a.claims_host = ctx->h_pin + 4;
    a.seq0 = ++ctx->tail_seq; ctx->h_rep->dead = 0;
    a.save = ctx->tail_save; a.resume = 0; a.timeout_ticks = (unsigned long long) std::max(1, ctx->opt.persistent_timeout_ms) * 100000ull;
    ctx->tail_args = a;
    hipLaunchKernelGGL(k_phase, dim3(1), dim3(VP_PH_THREADS), (size_t) 3 * ents * sizeof(F), ctx->stream, a);
    { const hipError_t e = hipGetLastError(); if (e != hipSuccess) { if ((ctx->opt.debug & 1)) fprintf(stderr, "[vp] k_phase launch failed: %s\n", hipGetErrorString(e)); --ctx->tail_seq; return 0; } }
    count_launch(ctx);
    ctx->tail_active = true;
    return 1;
}

void compute_layout(vp_ctx *ctx) {
    const int n = ctx->n_layers;
    ctx->ru_off.assign(n, 0); ctx->as_off.assign(n, 0); ctx->rv_off.assign(n, 0);
    ctx->sig_off.assign(n, 0); ctx->rliu_off.assign(n + 1, 0);
    u64 off = ctx->L[n - 1].bl;                       // r_0 at offset 0 (src/verifier.cpp:144)
    u64 tr = 1;                                       // Vres
    for (int i = n - 1; i >= 1; --i) {
        ctx->ru_off[i] = off; off += ctx->max_bl;     // verifier.cpp:196
        ctx->as_off[i] = off; off += 1;               // :202
        if (ctx->L[i].max_dad_bl != -1) { ctx->rv_off[i] = off; off += ctx->L[i].max_dad_bl; }   // :236
        ctx->sig_off[i] = off; off += n;              // :278
        ctx->rliu_off[i] = off; off += ctx->max_bl;   // :279
        tr += 3 * (u64) ctx->L[i - 1].bl + 1;
        if (ctx->L[i].max_dad_bl != -1) tr += 3 * (u64) ctx->L[i].max_dad_bl + i;
        tr += 3 * (u64) ctx->L[i - 1].bl + 1;
    }
    ctx->n_tape = off;
    ctx->n_tr = tr;
}
// device address of the point at which layer i's claim lives (prover::r_liu during layer i)
const F *rliu_ptr(vp_ctx *ctx, int i) {
    return i == ctx->n_layers - 1 ? ctx->d_tape : ctx->d_tape + ctx->rliu_off[i + 1];
}

}  // namespace

#include "vpgpu_upload.inc"

extern "C" {

int vp_checked_build(void) {
#ifdef VP_CHECKED
    return 1;
#else
    return 0;
#endif
}
const char *vp_version(void) { return "vpgpu 0.1 (gfx950)"; }
const char *vp_last_error(const vp_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

static void opt_defaults(VpOpt *o) {
    memset(o, 0, sizeof *o);
    o->gkr_path = VP_PATH_PLAN; o->use_graph = 1; o->serial = 0; o->fuse_init = 1; o->fuse_min_log = 0; o->fuse_dot = 0;
    o->drop_y = 1; o->drop_y_round1 = 0; o->real_values = 1; o->seg_tiny = 1; o->sf_big_log = 14;
    o->sf3b_grid = 512; o->dot_blocks = 1024; o->plan_align = 0; o->xcd_map = 0; o->round_fused_max = 512;
    o->persistent_rounds = 1; o->poll = 1; o->debug = 0; o->prefetch_round1 = 1; o->split_cost_percent = 50; o->kernel_copies = 1; o->fold_branches = 1; o->ntt_scatter = 1; o->fuse_combine = 2; o->plan_autotune = 1;
    o->pc_tensor_pub = 1;
    o->persistent_timeout_ms = 10000;
    o->graph_explicit = 0;
    o->ntt_r8 = 1;
    o->fri_vo_fused = 1;
    o->interactive_fast_init = 1;
    o->fuse_p2 = 1;
    o->leaf_asm = 1;
    o->real_pairs = 1;
    o->fft_gkr_batched = 1;
    o->split_vu = 1;
    o->fri_fold3 = 1;
    o->sf3c = 0;
}
static void opt_to_public(const VpOpt &o, vp_options *p) {
    memset(p, 0, sizeof *p);
    p->struct_size = (uint32_t) sizeof *p; p->abi = VP_OPTIONS_ABI;
    p->use_graph = o.use_graph; p->plan_autotune = o.plan_autotune; p->real_values = o.real_values; p->real_pairs = o.real_pairs; p->pc_tensor_pub = o.pc_tensor_pub;
    p->persistent_rounds = o.persistent_rounds; p->persistent_timeout_ms = o.persistent_timeout_ms; p->poll = o.poll; p->prefetch_round1 = o.prefetch_round1;
    p->interactive_fast_init = o.interactive_fast_init; p->split_cost_percent = o.split_cost_percent; p->debug = o.debug;
}
void vp_options_default(vp_options *p) {
    if (!p) return;
    VpOpt o; opt_defaults(&o);
    opt_to_public(o, p);
}
int vp_test_drivers(void) {
#ifdef VP_TEST_DRIVERS
    return 1;
#else
    return 0;
#endif
}
// internal tuning by name (vp_tuning_get; DESIGN.md section 4 lists the names)
struct OptName { const char *name; int32_t VpOpt::*field; };
static const OptName g_opt_names[] = {
    {"gkr_path", &VpOpt::gkr_path}, {"use_graph", &VpOpt::use_graph}, {"serial", &VpOpt::serial}, {"fuse_init", &VpOpt::fuse_init}, {"fuse_min_log", &VpOpt::fuse_min_log},
    {"fuse_dot", &VpOpt::fuse_dot}, {"drop_y", &VpOpt::drop_y}, {"drop_y_round1", &VpOpt::drop_y_round1}, {"real_values", &VpOpt::real_values}, {"seg_tiny", &VpOpt::seg_tiny},
    {"sf_big_log", &VpOpt::sf_big_log}, {"sf3b_grid", &VpOpt::sf3b_grid}, {"dot_blocks", &VpOpt::dot_blocks}, {"plan_align", &VpOpt::plan_align}, {"xcd_map", &VpOpt::xcd_map},
    {"round_fused_max", &VpOpt::round_fused_max}, {"persistent_rounds", &VpOpt::persistent_rounds}, {"poll", &VpOpt::poll}, {"debug", &VpOpt::debug},
    {"prefetch_round1", &VpOpt::prefetch_round1}, {"split_cost_percent", &VpOpt::split_cost_percent}, {"kernel_copies", &VpOpt::kernel_copies},
    {"fold_branches", &VpOpt::fold_branches}, {"ntt_scatter", &VpOpt::ntt_scatter}, {"fuse_combine", &VpOpt::fuse_combine}, {"plan_autotune", &VpOpt::plan_autotune},
    {"pc_tensor_pub", &VpOpt::pc_tensor_pub}, {"persistent_timeout_ms", &VpOpt::persistent_timeout_ms}, {"graph_explicit", &VpOpt::graph_explicit}, {"ntt_r8", &VpOpt::ntt_r8},
    {"fri_vo_fused", &VpOpt::fri_vo_fused}, {"interactive_fast_init", &VpOpt::interactive_fast_init}, {"fuse_p2", &VpOpt::fuse_p2}, {"leaf_asm", &VpOpt::leaf_asm},
    {"real_pairs", &VpOpt::real_pairs}, {"fft_gkr_batched", &VpOpt::fft_gkr_batched}, {"split_vu", &VpOpt::split_vu}, {"fri_fold3", &VpOpt::fri_fold3}, {"sf3c", &VpOpt::sf3c}};
int vp_tuning_get(const vp_ctx *ctx, const char *name, int32_t *value) {
    if (!ctx || !name || !value) return VP_EINVAL;
    for (const OptName &n : g_opt_names) if (!strcmp(n.name, name)) { *value = ctx->opt.*(n.field); return VP_OK; }
    return VP_EINVAL;
}
// defaults <- the caller's struct (refused unless it is THIS header's layout) <- VP_* environment variables (test-only override, read here and nowhere else).
// VP_OK, or VP_EINVAL for a struct of another layout / a driver this build does not have.
static int resolve_options(VpOpt *o, const vp_options *user, uint32_t *pinned) {
    opt_defaults(o);
    if (user) {
        if (user->struct_size != sizeof(vp_options) || user->abi != VP_OPTIONS_ABI) return VP_EINVAL;
        o->use_graph = user->use_graph; o->plan_autotune = user->plan_autotune; o->real_values = user->real_values; o->real_pairs = user->real_pairs;
        o->pc_tensor_pub = user->pc_tensor_pub; o->persistent_rounds = user->persistent_rounds; o->persistent_timeout_ms = user->persistent_timeout_ms;
        o->poll = user->poll; o->prefetch_round1 = user->prefetch_round1; o->interactive_fast_init = user->interactive_fast_init;
        o->split_cost_percent = user->split_cost_percent; o->debug = user->debug;
    }
    auto flag = [](const char *name, int32_t &v) { const char *e = getenv(name); if (e && (e[0] == '0' || e[0] == '1')) v = e[0] - '0'; };
    auto num = [](const char *name, int32_t &v) { const char *e = getenv(name); if (e && *e) v = atoi(e); };
    if (const char *p = getenv("VP_GKR_PATH"))
        o->gkr_path = !strcmp(p, "lanes") ? VP_PATH_LANES : !strcmp(p, "simple") ? VP_PATH_SIMPLE : VP_PATH_PLAN;
#ifndef VP_TEST_DRIVERS
    if (o->gkr_path != VP_PATH_PLAN) return VP_EINVAL;                   // the cross-check drivers are not in this build (vp_test_drivers)
#endif
    if (const char *p = getenv("VP_PLAN_ALIGN")) o->plan_align = !strcmp(p, "left") ? 1 : !strcmp(p, "right") ? 2 : 0;
    flag("VP_GKR_GRAPH", o->use_graph); flag("VP_GKR_SERIAL", o->serial); flag("VP_FUSE_INIT", o->fuse_init); flag("VP_FUSE_DOT", o->fuse_dot);
    flag("VP_DROP_Y", o->drop_y); flag("VP_DROP_Y1", o->drop_y_round1); flag("VP_REAL_V", o->real_values);
    flag("VP_SEG_TINY", o->seg_tiny); flag("VP_XCD_MAP", o->xcd_map); flag("VP_PERSIST", o->persistent_rounds);
    flag("VP_POLL", o->poll); flag("VP_PREFETCH_R1", o->prefetch_round1);
    num("VP_FUSE_MIN_LOG", o->fuse_min_log); num("VP_SF_BIG_LOG", o->sf_big_log);
    num("VP_SF3B_GRID", o->sf3b_grid); num("VP_DOT_BLOCKS", o->dot_blocks);
    num("VP_ROUND_FUSED_MAX", o->round_fused_max);
    num("VP_SPLIT_COST_PERCENT", o->split_cost_percent);
    flag("VP_KERNEL_COPIES", o->kernel_copies);
    flag("VP_FOLD_BRANCHES", o->fold_branches);
    flag("VP_NTT_SCATTER", o->ntt_scatter);
    flag("VP_NTT_R8", o->ntt_r8);
    flag("VP_FRI_VO_FUSED", o->fri_vo_fused);
    flag("VP_FAST_INIT", o->interactive_fast_init);
    flag("VP_FUSE_P2", o->fuse_p2);
    flag("VP_LEAF_ASM", o->leaf_asm);
    flag("VP_REAL_PAIRS", o->real_pairs);
    flag("VP_FFT_GKR_BATCHED", o->fft_gkr_batched);
    flag("VP_SPLIT_VU", o->split_vu);
    flag("VP_FRI_FOLD3", o->fri_fold3);
    flag("VP_SF3C", o->sf3c);
    flag("VP_PC_TENSOR", o->pc_tensor_pub);
    num("VP_PERSIST_TIMEOUT_MS", o->persistent_timeout_ms);
    num("VP_GRAPH_EXPLICIT", o->graph_explicit);
    num("VP_FUSE_COMBINE", o->fuse_combine);
    flag("VP_PLAN_AUTOTUNE", o->plan_autotune);
    if (getenv("VP_DEBUG")) o->debug |= 1;
    if (getenv("VP_DEBUG_UPLOAD")) o->debug |= 2;                       // bit 1: phase times of vp_circuit_upload
    o->dot_blocks = std::max(1, o->dot_blocks); o->sf3b_grid = std::max(1, o->sf3b_grid);
    VpOpt d; opt_defaults(&d);
    *pinned = (o->fuse_combine != d.fuse_combine ? 1u : 0u) | (o->fold_branches != d.fold_branches ? 2u : 0u) | (o->plan_align != d.plan_align ? 4u : 0u) |
              (o->fuse_min_log != d.fuse_min_log ? 8u : 0u) | (o->sf3b_grid != d.sf3b_grid ? 16u : 0u) | (o->graph_explicit != d.graph_explicit ? 32u : 0u);
    return VP_OK;
}

// What plan_autotune chose for this circuit (after the first vp_prove_gkr), and the way to hand such a choice to another process: vp_plan_tuning_set on a
// fresh context before its first proof (the tuner then has nothing left to try; values the library does not accept are refused).
// v = { fuse_combine, fold_branches, plan_align, fuse_min_log, sf3b_grid, graph_explicit }.
int vp_plan_tuning_get(const vp_ctx *ctx, int32_t v[6]) {
    if (!ctx || !v) return VP_EINVAL;
    v[0] = ctx->opt.fuse_combine; v[1] = ctx->opt.fold_branches; v[2] = ctx->opt.plan_align; v[3] = ctx->opt.fuse_min_log; v[4] = ctx->opt.sf3b_grid; v[5] = ctx->opt.graph_explicit;
    return VP_OK;
}
int vp_plan_tuning_set(vp_ctx *ctx, const int32_t v[6]) {
    if (!ctx || !v) return VP_EINVAL;
    if (v[0] < 0 || v[0] > 2 || v[1] < 0 || v[1] > 1 || v[2] < 0 || v[2] > 2 || v[3] < 0 || v[3] > 30 || v[4] < 1 || v[4] > 4096 || v[5] < 0 || v[5] > 3) return VP_EINVAL;
    VP_ENTER(ctx);
    if (ctx->plan || ctx->gkr_graph) { ctx->err = "vp_plan_tuning_set: the context has already recorded its launch plan"; return VP_EINVAL; }
    ctx->opt.fuse_combine = v[0]; ctx->opt.fold_branches = v[1]; ctx->opt.plan_align = v[2]; ctx->opt.fuse_min_log = v[3]; ctx->opt.sf3b_grid = v[4]; ctx->opt.graph_explicit = v[5];
    ctx->plan_tuned = true;
    return VP_OK;
}
int vp_create(int device, vp_ctx **out) { return vp_create_with_options(device, nullptr, out); }
int vp_get_options(const vp_ctx *ctx, vp_options *out) {
    if (!ctx || !out) return VP_EINVAL;
    opt_to_public(ctx->opt, out);
    return VP_OK;
}

// The caller's glibc random() / rand() stream is not ours to consume: the reference verifier draws every challenge and every query position
// from it (lib/virgo/src/fieldElement.cpp:119-124,362-367, vpd_verifier.cpp:121), and the ROCm runtime takes draws from the same process-wide
// generator while it initialises (measured through the reference binary of INTEGRATION.md: vp_create shifted the stream, the proof still
// verified but was no longer the CPU reference's).  The set-up entry points therefore run on a private generator state and hand the caller's
// back untouched.
// (The generator state is process-wide: two threads inside set-up entry points at once would hand each other's private state back.  They take
// turns: one lock around every RandKeep.)
static std::recursive_mutex g_rand_mu;
struct RandKeep {
    char priv[128]; char *caller;
    RandKeep() { g_rand_mu.lock(); memset(priv, 0, sizeof priv); caller = initstate(1u, priv, sizeof priv); }
    ~RandKeep() { if (caller) setstate(caller); g_rand_mu.unlock(); }
};

int vp_create_with_options(int device, const vp_options *user, vp_ctx **out) {
    if (!out) return VP_EINVAL;
    *out = nullptr;
    RandKeep keep_callers_random_stream;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return VP_ENOGPU;
    if (hipSetDevice(device) != hipSuccess) return VP_ENOGPU;
    vp_ctx *ctx = new vp_ctx();
    ctx->device = device;
    if (resolve_options(&ctx->opt, user, &ctx->opt_pinned) != VP_OK) { delete ctx; return VP_EINVAL; }       // a vp_options of another layout, or a driver this build lacks
    if (const char *pcf = getenv("VP_PLAN_CACHE")) ctx->plan_cache_path = pcf;                                // (read here, with the other VP_* variables, and nowhere else)
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return VP_EHIP; }
    if (hipHostMalloc((void **) &ctx->h_pin, (4 + VP_MAX_TAB) * sizeof(F), hipHostMallocDefault) != hipSuccess) {
        delete ctx; return VP_EHIP;
    }
    if (hipHostMalloc((void **) &ctx->h_seq, 64, hipHostMallocDefault) != hipSuccess) { delete ctx; return VP_EHIP; }
    *ctx->h_seq = 0;
    if (hipHostMalloc((void **) &ctx->h_req, sizeof(TailMail), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **) &ctx->h_rep, sizeof(TailReply), hipHostMallocDefault) != hipSuccess) { delete ctx; return VP_EHIP; }
    if (hipHostMalloc((void **) &ctx->h_aux, sizeof(TailAux), hipHostMallocDefault) != hipSuccess) { delete ctx; return VP_EHIP; }
    if (hipHostMalloc((void **) &ctx->h_stage, 4096 * sizeof(F), hipHostMallocDefault) != hipSuccess) { delete ctx; return VP_EHIP; }
    memset(ctx->h_req, 0, sizeof(TailMail)); memset(ctx->h_rep, 0, sizeof(TailReply)); memset(ctx->h_aux, 0, sizeof(TailAux));
    (void) hipFuncSetAttribute(reinterpret_cast<const void *>(k_phase), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 2 * VP_PH_PMAX * (int) sizeof(F));
    ctx->poll = ctx->opt.poll; ctx->tail_enabled = ctx->opt.persistent_rounds;
    // distributed rounds (G workgroups of the resident kernel): measured equal to one launch per round on MI355X (x64 interactive proof
    // 12.6 vs 12.7 ms: profiles/r02_interactive_*), and they need G idle CUs for as long as the verifier takes — opt-in
    hipEventCreate(&ctx->ev0); hipEventCreate(&ctx->ev1);
    (void) hipFuncSetAttribute(reinterpret_cast<const void *>(k_emit_multi), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    (void) hipFuncSetAttribute(reinterpret_cast<const void *>(k_emit), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    (void) hipFuncSetAttribute(reinterpret_cast<const void *>(k_ntt_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    (void) hipFuncSetAttribute(reinterpret_cast<const void *>(k_ntt8_cols<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void) hipFuncSetAttribute(reinterpret_cast<const void *>(k_ntt8_cols<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void) hipFuncSetAttribute(reinterpret_cast<const void *>(k_ntt8_rows<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void) hipFuncSetAttribute(reinterpret_cast<const void *>(k_ntt8_rows<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    { std::lock_guard<std::mutex> lk(g_ctx_mu); g_ctxs.push_back(ctx); }
    *out = ctx;
    return VP_OK;
}

static void free_plan(vp_ctx *ctx);
void vp_free_shard_state(vp_ctx *ctx);
void vp_free_comm(vp_ctx *ctx);
void vp_free_fgk(vp_ctx *ctx);

void vp_destroy(vp_ctx *ctx) {
    if (!ctx) return;
    if (tl_entry_depth == 0) vp_suspend_others(ctx);          // the hipFree calls below synchronise the device
    // out of the registry first (no other thread can reach the context any more), then through its lock once (a suspender that had it is gone)
    { std::lock_guard<std::mutex> lk(g_ctx_mu); g_ctxs.erase(std::remove(g_ctxs.begin(), g_ctxs.end(), ctx), g_ctxs.end()); }
    { std::lock_guard<std::recursive_mutex> lo(ctx->mu); }
    (void) hipSetDevice(ctx->device);
    if (ctx->tail_active) (void) tail_quit(ctx);
    (void) hipStreamSynchronize(ctx->stream);
    vp_free_shard_state(ctx);
    vp_free_comm(ctx);
    vp_free_fgk(ctx);
    if (ctx->gkr_graph) (void) hipGraphExecDestroy(ctx->gkr_graph);
    free_plan(ctx);
    free_all(ctx);
    for (auto &e : ctx->ev_pool) { (void) hipEventDestroy(e.a); (void) hipEventDestroy(e.b); }
    if (ctx->ev0) (void) hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void) hipEventDestroy(ctx->ev1);
    if (ctx->h_pin) (void) hipHostFree(ctx->h_pin);
    if (ctx->h_seq) (void) hipHostFree(ctx->h_seq);
    if (ctx->h_req) (void) hipHostFree(ctx->h_req);
    if (ctx->h_rep) (void) hipHostFree(ctx->h_rep);
    if (ctx->h_aux) (void) hipHostFree(ctx->h_aux);
    if (ctx->h_stage) (void) hipHostFree(ctx->h_stage);
    if (ctx->h_io) (void) hipHostFree(ctx->h_io);
    if (ctx->h_ring) (void) hipHostFree(ctx->h_ring);
    if (ctx->h_pub) (void) hipHostFree(ctx->h_pub);
    for (auto &p : ctx->pending) { (void) hipEventDestroy(p.a); (void) hipEventDestroy(p.b); }
    for (auto e : ctx->ev_spare) (void) hipEventDestroy(e);
    for (auto st : ctx->lane_streams) (void) hipStreamDestroy(st);
    for (auto ev : ctx->lane_events) (void) hipEventDestroy(ev);
    if (ctx->ev_fork) (void) hipEventDestroy(ctx->ev_fork);
    (void) hipStreamDestroy(ctx->stream);
    delete ctx;
}

int vp_circuit_upload(vp_ctx *ctx, int n_layers, const vp_layer_desc *ld) {
    if (!ctx || !ld || n_layers < 2) return VP_EINVAL;
    if (n_layers > VP_MAX_TAB) { ctx->err = "too many layers"; return VP_ELIMIT; }
    RandKeep keep_callers_random_stream;                               // rocPRIM / first-use module loads: see vp_create_with_options
    VP_ENTER(ctx);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->gkr_graph) { (void) hipGraphExecDestroy(ctx->gkr_graph); ctx->gkr_graph = nullptr; }
    ctx->graph_failed = false; ctx->plan_tuned = false; ctx->plan_tune_cached = false;
    free_plan(ctx);
    vp_free_shard_state(ctx);
    free_all(ctx);
    ctx->L.assign(n_layers, LayerDev());
    ctx->n_layers = n_layers;
    ctx->evaluated = false;
    ctx->chain_owner.clear(); ctx->chain_cost.clear();
    ctx->chunk_cap = 0;
    ctx->pred_r = ctx->pred_pool = ctx->pred_part = ctx->pred_out = nullptr; ctx->pred_jobs = nullptr; ctx->pred_dot = nullptr; ctx->pred_map = nullptr;
    ctx->pc_rt = ctx->pc_coef = ctx->pc_cw = nullptr; ctx->pc_tree = nullptr; ctx->pc_lm = -1; ctx->pc_rtc.clear();
    ctx->pc_pub = ctx->pc_qcw = ctx->pc_hcw = ctx->pc_tmp = ctx->pc_small = nullptr; ctx->pc_tree_h = nullptr; ctx->pc_private_done = false;
    ctx->pc_q0 = nullptr; ctx->pc_eq = nullptr; ctx->pc_cbuf = nullptr; ctx->pc_cbuf_lm = -1; ctx->pc_flag = nullptr; ctx->pc_q_tensor = false;
    ctx->pc_scr = nullptr; ctx->pc_scr_cap = 0; ctx->pc_fri_all = nullptr; ctx->pc_open_buf = nullptr; ctx->fri_cw_off.clear(); ctx->fri_tree_off.clear();
    ctx->pc_fri[0] = ctx->pc_fri[1] = nullptr; ctx->pc_fri_tree = nullptr; ctx->pc_fri_roots = nullptr; ctx->fri_step = -1; ctx->pc_public_done = false;
    ctx->pc_mask_ms = 0; ctx->pc_lm_cw = ctx->pc_qm_cw = ctx->pc_hm_cw = ctx->pc_fm = ctx->pc_mtmp = nullptr; ctx->fri_m_off.clear(); ctx->pc_mtmp_cap = 0; ctx->pc_mB = 0;
    int max_bl = 0;
    for (int i = 0; i < n_layers; ++i) {
        if (ld[i].size == 0 || ld[i].size > (1ull << 30) || ld[i].bit_length < 0 || ld[i].bit_length > 30 ||
            (1ull << ld[i].bit_length) < ld[i].size) { ctx->err = "bad layer size"; return VP_ELIMIT; }
        max_bl = std::max(max_bl, (int) ld[i].bit_length);
    }
    ctx->max_bl = max_bl;
    // ---- per-layer uploads: the raw gate arrays go to HBM as they are, every per-gate structure is built there (vpgpu_upload.inc) ----
    UpScratch S;
    const bool up_dbg = (ctx->opt.debug & 2) != 0;
    const auto up_t0 = std::chrono::steady_clock::now();
    auto up_since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - up_t0).count(); };
    u64 *d_sizes = nullptr;
    {
        std::vector<u64> sizes(n_layers);
        for (int i = 0; i < n_layers; ++i) sizes[i] = ld[i].size;
        VPCHK(dupload(ctx, &d_sizes, sizes));
    }
    for (int i = 0; i < n_layers; ++i) {
        LayerDev &D = ctx->L[i];
        const vp_layer_desc &L = ld[i];
        D.size = L.size; D.bl = L.bit_length;
        const u64 nval = i == 0 ? (1ull << L.bit_length) : L.size;     // circuitValue[0] is padded (prover.cpp:30)
        VPCHK(dalloc(ctx, &D.val, nval));
        if (i == 0) continue;
        if (!L.ty || !L.l || !L.u || !L.v || !L.lv || !L.dad_size || !L.dad_bitlen || !L.dad_id) { ctx->err = "layer arrays missing"; return VP_EINVAL; }
        VPCHK(upload_layer_dev(ctx, S, i, ld, d_sizes));
    }
    if (up_dbg) fprintf(stderr, "[vp_circuit_upload] layers %.3f s", up_since());
    // ---- shared buffers ----
    u32 cap = 0;
    for (int i = 1; i < n_layers; ++i) cap = std::max<u32>(cap, std::max<u32>(1u << ctx->L[i - 1].bl, ctx->L[i].p2_total));
    ctx->cap = cap;
    for (int b = 0; b < 2; ++b) for (int t = 0; t < 3; ++t) VPCHK(dalloc(ctx, &ctx->tab[b][t], (size_t) cap));
    VPCHK(dalloc(ctx, &ctx->beta_g, (size_t) 1 << max_bl));
    VPCHK(dalloc(ctx, &ctx->beta_u, (size_t) 1 << max_bl));
    ctx->half_cap = 1u << ((max_bl + 1) / 2);
    VPCHK(dalloc(ctx, &ctx->bf, (size_t) ctx->half_cap));
    VPCHK(dalloc(ctx, &ctx->bs, (size_t) ctx->half_cap));
    VPCHK(dalloc(ctx, &ctx->liu_half, (size_t) 2 * (n_layers + 1) * ctx->half_cap));
    VPCHK(dalloc(ctx, &ctx->partials, (size_t) 3 * MAX_BLOCKS));
    VPCHK(dalloc(ctx, &ctx->round_arrivals, (size_t) 16)); HIPCHK(hipMemsetAsync(ctx->round_arrivals, 0, 64, ctx->stream));
    VPCHK(dalloc(ctx, &ctx->chunk_part, (size_t) 2 * std::max<u32>(1, ctx->chunk_cap)));
    VPCHK(dalloc(ctx, &ctx->small, (size_t) 32 + VP_MAX_TAB));
    VPCHK(dalloc(ctx, &ctx->d_flag, (size_t) 1));
    VPCHK(dalloc(ctx, &ctx->d_vcplx, (size_t) 1));
    {
        std::vector<F> sm(32 + VP_MAX_TAB, f_zero());
        sm[1] = f_one();
        HIPCHK(hipMemcpy(ctx->small, sm.data(), sm.size() * sizeof(F), hipMemcpyHostToDevice));
        std::vector<F *> vals(n_layers);
        for (int i = 0; i < n_layers; ++i) vals[i] = ctx->L[i].val;
        VPCHK(dupload(ctx, &ctx->d_vals, vals));
    }
    if (up_dbg) fprintf(stderr, "  shared %.3f", up_since());
    compute_layout(ctx);
    VPCHK(dalloc(ctx, &ctx->d_tape, (size_t) ctx->n_tape));
    VPCHK(dalloc(ctx, &ctx->d_tr, (size_t) ctx->n_tr + 3 + VP_MAX_TAB));
    HIPCHK(hipMemset(ctx->d_tape, 0, ctx->n_tape * sizeof(F)));
    // ---- Liu jobs (src/prover.cpp:396,402-414): eq tables over r_u and every later layer's r_v ----
    for (int i = 1; i < n_layers; ++i) {
        LayerDev &D = ctx->L[i];
        std::vector<BetaJob> jobs;
        const u32 hc = ctx->half_cap;
        auto add = [&](const F *r, const F *init, int nbits, int k) {
            BetaJob jb{};
            const size_t q = jobs.size();
            jb.r = r; jb.init = init; jb.bf = ctx->liu_half + 2 * q * hc; jb.bs = ctx->liu_half + (2 * q + 1) * hc; jb.n = nbits;
            jobs.push_back(jb); D.job_k.push_back(k); D.job_h1.push_back(nbits >> 1);
        };
        add(ctx->d_tape + ctx->ru_off[i], ctx->d_tape + ctx->sig_off[i], ctx->L[i - 1].bl, -1);
        for (int k = i; k < n_layers; ++k)
            if (ctx->L[k].dad_size[i - 1])
                add(ctx->d_tape + ctx->rv_off[k], ctx->d_tape + ctx->sig_off[i] + (k - i + 1), ctx->L[k].dad_bl[i - 1], k);
        D.n_jobs = (u32) jobs.size();
        VPCHK(dupload(ctx, &D.jobs, jobs));
    }
    if (up_dbg) fprintf(stderr, "  liujobs %.3f", up_since());
    // ---- batched path: per-slot gather map, Liu gather lists, one pool of half tables for the whole proof ----
    {
        std::vector<BetaJob> jobs;
        std::vector<size_t> need;            // pool elements per job (bf + bs)
        auto add_job = [&](const F *r, const F *init, int nbits) {
            BetaJob jb{}; jb.r = r; jb.init = init; jb.n = nbits;
            jobs.push_back(jb);
            need.push_back(((size_t) 1 << (nbits >> 1)) + ((size_t) 1 << (nbits - (nbits >> 1))));
            return (int) jobs.size() - 1;
        };
        std::vector<int> hg_id(n_layers, -1), hu_id(n_layers, -1);
        std::vector<std::vector<int>> liu_id(n_layers);
        for (int i = 1; i < n_layers; ++i) {
            hg_id[i] = add_job(rliu_ptr(ctx, i), ctx->one(), ctx->L[i].bl);
            hu_id[i] = add_job(ctx->d_tape + ctx->ru_off[i], ctx->one(), ctx->L[i - 1].bl);
            liu_id[i].push_back(add_job(ctx->d_tape + ctx->ru_off[i], ctx->d_tape + ctx->sig_off[i], ctx->L[i - 1].bl));
            for (int k = i; k < n_layers; ++k)
                if (ctx->L[k].dad_size[i - 1])
                    liu_id[i].push_back(add_job(ctx->d_tape + ctx->rv_off[k], ctx->d_tape + ctx->sig_off[i] + (k - i + 1),
                                                ctx->L[k].dad_bl[i - 1]));
        }
        size_t total = 0;
        for (size_t q = 0; q < jobs.size(); ++q) total += need[q];
        VPCHK(dalloc(ctx, &ctx->half_pool, total));
        size_t at = 0;
        for (size_t q = 0; q < jobs.size(); ++q) {
            jobs[q].bf = ctx->half_pool + at;
            jobs[q].bs = jobs[q].bf + ((size_t) 1 << (jobs[q].n >> 1));
            at += need[q];
        }
        ctx->n_all_jobs = (u32) jobs.size();
        ctx->beta_bpj = 1;
        for (auto &jb : jobs) ctx->beta_bpj = std::max<u32>(ctx->beta_bpj, nblk(((u64) 1 << (jb.n >> 1)) + ((u64) 1 << (jb.n - (jb.n >> 1)))));
        VPCHK(dupload(ctx, &ctx->all_jobs, jobs));
        auto half_of = [&](int q) { Half h{}; h.bf = jobs[q].bf; h.bs = jobs[q].bs; h.h1 = jobs[q].n >> 1; return h; };
        for (int i = 1; i < n_layers; ++i) {
            LayerDev &D = ctx->L[i];
            D.hg = half_of(hg_id[i]); D.hu = half_of(hu_id[i]);
            D.job_hg = hg_id[i]; D.job_hu = hu_id[i]; D.job_liu0 = liu_id[i].front(); D.job_liun = (int) liu_id[i].size();
            std::vector<Half> hs;
            for (int q : liu_id[i]) hs.push_back(half_of(q));
            VPCHK(dupload(ctx, &D.liu_H, hs));
            VPCHK(upload_liu_dev(ctx, S, i));           // Liu gather lists over the wires of layer i-1 (dadId[k][i-1] inverted)
        }
    if (up_dbg) fprintf(stderr, "  liulists %.3f", up_since());
        VPCHK(dalloc(ctx, &ctx->part2, (size_t) 32 * MAX_BLOCKS * 3));
        VPCHK(dalloc(ctx, &ctx->tail_save, (size_t) 3 * 2 * VP_PH_PMAX + 8 + VP_MAX_TAB));     // a suspended phase of the resident round kernel
        // lanes: per-layer scratch for the concurrent chains
        ctx->lane0.stream = ctx->stream;
        for (int b = 0; b < 2; ++b) for (int t = 0; t < 3; ++t) ctx->lane0.tab[b][t] = ctx->tab[b][t];
        ctx->lane0.part2 = ctx->part2; ctx->lane0.chunk_part = ctx->chunk_part; ctx->lane0.Vu = ctx->Vu();
        ctx->ln = &ctx->lane0;
        const int n_lanes = 2 * (n_layers - 1);
        while ((int) ctx->lane_streams.size() < std::max(n_lanes, 3)) {
            hipStream_t st; hipEvent_t ev;
            HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            ctx->lane_streams.push_back(st); ctx->lane_events.push_back(ev);
        }
        if (!ctx->ev_fork) HIPCHK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        ctx->lanes.assign(n_lanes + (n_layers - 1), Lane());      // [n_lanes + i-1]: phase 2 of layer i when it runs as its own chain (plan path)
        F *vus = nullptr;
        VPCHK(dalloc(ctx, &vus, (size_t) n_layers));
        for (int i = 1; i < n_layers; ++i) {
            for (int h = 0; h < 2; ++h) {
                Lane &ln = ctx->lanes[2 * (i - 1) + h];
                ln.stream = ctx->lane_streams[2 * (i - 1) + h];
                ln.done = ctx->lane_events[2 * (i - 1) + h];
                const size_t capi = h == 0 ? std::max<size_t>((size_t) 1 << ctx->L[i - 1].bl, ctx->L[i].p2_total) : ((size_t) 1 << ctx->L[i - 1].bl);
                for (int b = 0; b < 2; ++b) for (int t = 0; t < 3; ++t) {
                    if (h == 1 && t == 2) { ln.tab[b][t] = ctx->tab[0][2]; continue; }     // Liu never touches the add table
                    VPCHK(dalloc(ctx, &ln.tab[b][t], capi));
                }
                VPCHK(dalloc(ctx, &ln.part2, (size_t) 16 * MAX_BLOCKS * 3));
                const u32 nch = h == 0 ? std::max(ctx->L[i].c1.n_chunks, ctx->L[i].c2.n_chunks) : 0;
                VPCHK(dalloc(ctx, &ln.chunk_part, (size_t) 2 * std::max<u32>(1, nch)));
                ln.Vu = vus + i;
            }
            if (ctx->L[i].max_dad_bl != -1) {
                Lane &ln = ctx->lanes[n_lanes + (i - 1)];
                ln.stream = ctx->stream;
                for (int b = 0; b < 2; ++b) for (int t = 0; t < 3; ++t) VPCHK(dalloc(ctx, &ln.tab[b][t], std::max<size_t>(1, ctx->L[i].p2_total)));
                VPCHK(dalloc(ctx, &ln.part2, (size_t) 16 * MAX_BLOCKS * 3));
                VPCHK(dalloc(ctx, &ln.chunk_part, (size_t) 2 * std::max<u32>(1, ctx->L[i].c2.n_chunks)));
                VPCHK(dalloc(ctx, &ln.dot_part, (size_t) nblk(ctx->L[i - 1].size) + 1));      // V_u block partials of the phase-1 init launch
                ln.Vu = vus + i;
            }
        }
        ctx->serial = ctx->opt.serial; ctx->use_graph = ctx->opt.use_graph;
        ctx->simple_path = ctx->opt.gkr_path == VP_PATH_SIMPLE;
        ctx->plan_path = ctx->opt.gkr_path == VP_PATH_LANES ? 0 : 1;
        ctx->fuse_init = ctx->opt.fuse_init;
    }
    if (up_dbg) fprintf(stderr, "  total %.3f s\n", up_since());
#ifdef VP_CHECKED
    {   // bounds for the device-side checks (vp_check.h): this circuit's layer sizes, the largest table any fold launch may write, the most eq
        // tables a Liu gather selects from.  VP_CHECKED_INJECT=1 (test hook) shrinks layer 0's bound to ONE wire, so that a correct gather trips site 1 / 2.
        VpChkDesc d{};
        d.n_layers = (unsigned) n_layers;
        unsigned long long cap = 1;
        for (int i = 0; i < n_layers; ++i) {
            d.lsize[i] = (unsigned) ctx->L[i].size;
            cap = std::max<unsigned long long>(cap, 1ull << ctx->L[i].bl);
            cap = std::max<unsigned long long>(cap, ctx->L[i].p2_total);
            d.n_liu_tables_max = std::max<unsigned>(d.n_liu_tables_max, (unsigned) ctx->L[i].job_liun);
        }
        d.table_cap = cap;
        if (getenv("VP_CHECKED_INJECT")) d.lsize[0] = 1;
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_vp_chk), &d, sizeof d));
        const unsigned int z[4] = {0, 0, 0, 0};
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_vp_chk_err), z, sizeof z));
    }
#endif
    // event pool for the profiled launches
    if (ctx->ev_pool.empty()) {
        ctx->ev_pool.resize(1024);
        for (auto &e : ctx->ev_pool) { hipEventCreate(&e.a); hipEventCreate(&e.b); e.bytes = 0; e.kind = -1; e.grid = e.jobs = e.rounds = e.first_round = 0; e.work = 0; }
    }
    return VP_OK;
}

int vp_evaluate(vp_ctx *ctx, const vp_F *inputs, uint64_t n_inputs) {
    if (!ctx || !inputs || ctx->n_layers < 2 || n_inputs != ctx->L[0].size) return VP_EINVAL;
    VP_ENTER(ctx);
    LayerDev &L0 = ctx->L[0];
    ctx->pc_private_done = false; ctx->pc_public_done = false; ctx->pc_mask_ms = 0;          // a commitment to the previous witness does not stand for this one
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    HIPCHK(hipMemsetAsync(L0.val, 0, (sizeof(F) << L0.bl), ctx->stream));
    HIPCHK(hipMemcpyAsync(L0.val, inputs, n_inputs * sizeof(F), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_flag, 0, sizeof(int), ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_vcplx, 0, sizeof(u32), ctx->stream));
    hipLaunchKernelGGL(k_mark_complex, dim3(nblk(L0.size)), dim3(VP_BLOCK), 0, ctx->stream, L0.val, (u32) L0.size, ctx->d_vcplx);
    for (int i = 1; i < ctx->n_layers; ++i) {
        LayerDev &D = ctx->L[i];
        hipLaunchKernelGGL(k_evaluate_layer, dim3(nblk(D.size)), dim3(VP_BLOCK), 0, ctx->stream, i, (u32) D.size, D.ty,
                           D.gl, D.gu, D.gv, D.gc, ctx->d_vals, ctx->d_vcplx);
        if (D.n_assert)
            hipLaunchKernelGGL(k_check_asserts, dim3(nblk(D.n_assert)), dim3(VP_BLOCK), 0, ctx->stream, D.assert_idx,
                               D.n_assert, D.val, ctx->d_flag);
    }
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    int flag = 0; u32 vcplx = 1;
    HIPCHK(hipMemcpyAsync(&flag, ctx->d_flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(&vcplx, ctx->d_vcplx, sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
    VPCHK(check_stream(ctx));
    ctx->vreal = (vcplx == 0 && ctx->opt.real_values) ? 1 : 0;
    if (ctx->vreal) {
        std::vector<unsigned long long *> vr(ctx->n_layers);
        for (int i = 0; i < ctx->n_layers; ++i) {
            LayerDev &D = ctx->L[i];
            if (!D.valr) VPCHK(dalloc(ctx, &D.valr, (size_t) D.size));
            hipLaunchKernelGGL(k_real_parts, dim3(nblk(D.size)), dim3(VP_BLOCK), 0, ctx->stream, D.val, (u32) D.size, D.valr);
            vr[i] = D.valr;
        }
        if (!ctx->d_valsr) VPCHK(dalloc(ctx, &ctx->d_valsr, (size_t) ctx->n_layers));
        HIPCHK(hipMemcpyAsync(ctx->d_valsr, vr.data(), vr.size() * sizeof(unsigned long long *), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));      // vr lives on this frame
    }
    float ms = 0;
    hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    ctx->st.evaluate_ms = ms;
    ctx->evaluated = true;
    HIPCHK(hipMemsetAsync(ctx->add_term(), 0, sizeof(F), ctx->stream));
    if (flag) { ctx->err = "assert gate is non-zero"; return VP_EASSERT; }
    return VP_OK;
}

int vp_layer_values(vp_ctx *ctx, int layer, vp_F *out, uint64_t n) {
    if (!ctx || !out || layer < 0 || layer >= ctx->n_layers || n > ctx->L[layer].size) return VP_EINVAL;
    VP_ENTER(ctx);
    VPCHK(check_stream(ctx));
    HIPCHK(hipMemcpy(out, ctx->L[layer].val, n * sizeof(F), hipMemcpyDeviceToHost));
    return VP_OK;
}

static int stage(vp_ctx *ctx, u64 off, const vp_F *src, u64 n) {
    if (!n) return VP_OK;
    if (!src || off + n > ctx->n_tape) return VP_EINVAL;
    // through a pinned ring (4096 elements; a phase stages < 150): the caller's array need not outlive the call and the copy is truly asynchronous
    if (n > 1024) { HIPCHK(hipMemcpy(ctx->d_tape + off, src, n * sizeof(F), hipMemcpyHostToDevice)); return VP_OK; }
    if (ctx->stage_at + n > 4096) ctx->stage_at = 0;
    F *slot = ctx->h_stage + ctx->stage_at;
    memcpy(slot, src, n * sizeof(F));
    ctx->stage_at += (u32) n;
    HIPCHK(hipMemcpyAsync(ctx->d_tape + off, slot, n * sizeof(F), hipMemcpyHostToDevice, ctx->stream));
    return VP_OK;
}
// Round 1 of a sumcheck needs no challenge: it is queued right behind the init kernels, and the init call returns without waiting.
// vp_round's first call of the phase then only waits for the answer (one synchronisation per phase start instead of two).
static int round1_prefetch(vp_ctx *ctx) {
    ctx->r1_pending = 0;
    if (!ctx->opt.prefetch_round1 || ctx->sc.total_rounds < 1 || ctx->profiling) return check_stream(ctx);
    const F zero = f_zero();
    if (tail_try_launch(ctx, zero)) { ctx->r1_pending = 1; return VP_OK; }
    VPCHK(do_round(ctx, nullptr, zero, ctx->d_tr + ctx->n_tr, ctx->h_pin));
    ctx->r1_pending = 2;
    return VP_OK;
}

int vp_predicates(vp_ctx *ctx, int layer, const vp_F *r_g, const vp_F *assert_random, const vp_F *r_u, const vp_F *r_v, int n_v,
                  vp_F *out, uint64_t n_out) {
    if (!ctx || layer < 1 || layer >= ctx->n_layers || !assert_random || !out) return VP_EINVAL;
    LayerDev &D = ctx->L[layer], &pre = ctx->L[layer - 1];
    if ((D.bl > 0 && !r_g) || (pre.bl > 0 && !r_u)) return VP_EINVAL;       // zero-variable layers have no challenges
    if (n_v < 0 || n_v > 31 || (n_v > 0 && !r_v) || n_out != D.p_buckets) { ctx->err = "vp_predicates: bad sizes"; return VP_EINVAL; }
    VP_ENTER(ctx);
    // scratch: challenges, three pairs of half tables, piece sums, bucket sums (sized for the largest layer, once)
    if (!ctx->pred_r) {
        u32 max_chunks = 1, max_buckets = 1;
        for (int i = 1; i < ctx->n_layers; ++i) { max_chunks = std::max(max_chunks, ctx->L[i].p_chunks); max_buckets = std::max(max_buckets, ctx->L[i].p_buckets); }
        VPCHK(dalloc(ctx, &ctx->pred_r, (size_t) 3 * 32 + 2));
        VPCHK(dalloc(ctx, &ctx->pred_pool, (size_t) 3 * 2 * ((size_t) 1 << 16)));
        VPCHK(dalloc(ctx, &ctx->pred_part, (size_t) max_chunks));
        VPCHK(dalloc(ctx, &ctx->pred_out, (size_t) max_buckets));
        VPCHK(dalloc(ctx, &ctx->pred_jobs, (size_t) 3));
    }
    const int n_g = D.bl, n_u = pre.bl;
    if (n_g > 31 || n_u > 31) { ctx->err = "vp_predicates: layer too large"; return VP_ELIMIT; }
    std::vector<F> h(3 * 32 + 2, f_zero());
    if (n_g) memcpy(h.data(), r_g, (size_t) n_g * sizeof(F));
    if (n_u) memcpy(h.data() + 32, r_u, (size_t) n_u * sizeof(F));
    if (n_v) memcpy(h.data() + 64, r_v, (size_t) n_v * sizeof(F));
    h[96] = f_one(); memcpy(&h[97], assert_random, sizeof(F));
    HIPCHK(hipMemcpyAsync(ctx->pred_r, h.data(), h.size() * sizeof(F), hipMemcpyHostToDevice, ctx->stream));
    BetaJob jb[3];
    const int nn[3] = {n_g, n_u, n_v};
    Half hh[3];
    u32 bpj = 1;
    for (int q = 0; q < 3; ++q) {
        jb[q].r = ctx->pred_r + 32 * q; jb[q].init = ctx->pred_r + 96; jb[q].n = nn[q]; jb[q].pad = 0;
        jb[q].bf = ctx->pred_pool + (size_t) q * 2 * ((size_t) 1 << 16); jb[q].bs = jb[q].bf + ((size_t) 1 << (nn[q] >> 1));
        hh[q].bf = jb[q].bf; hh[q].bs = jb[q].bs; hh[q].h1 = nn[q] >> 1; hh[q].pad = 0;
        bpj = std::max<u32>(bpj, nblk(((u64) 1 << (nn[q] >> 1)) + ((u64) 1 << (nn[q] - (nn[q] >> 1)))));
    }
    HIPCHK(hipMemcpyAsync(ctx->pred_jobs, jb, sizeof(jb), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));          // h and jb are stack/heap temporaries
    hipLaunchKernelGGL(k_beta_half_direct, dim3(3 * bpj), dim3(VP_BLOCK), 0, ctx->stream, ctx->pred_jobs, bpj);
    PredArgs a{};
    a.idx = D.p_idx; a.flag = D.p_flag; a.chunk_beg = D.p_cbeg; a.chunk_end = D.p_cend; a.n_chunks = D.p_chunks;
    a.hg = hh[0]; a.hu = hh[1]; a.hv = hh[2]; a.gu = D.gu; a.glv = D.glv; a.gc = D.gc; a.assert_r = ctx->pred_r + 97; a.part = ctx->pred_part;
    if (D.p_chunks) hipLaunchKernelGGL(k_pred_chunks, dim3((D.p_chunks + 3) / 4), dim3(VP_BLOCK), 0, ctx->stream, a);
    hipLaunchKernelGGL(k_pred_combine, dim3((D.p_buckets + 3) / 4), dim3(VP_BLOCK), 0, ctx->stream, D.p_bptr, D.p_buckets, ctx->pred_part, ctx->pred_out);
    HIPCHK(hipMemcpyAsync(out, ctx->pred_out, (size_t) D.p_buckets * sizeof(F), hipMemcpyDeviceToHost, ctx->stream));
    return check_stream(ctx);
}

// <eq(r, .), table> for a device table of `size` entries: the verifier-side building block of vp_liu_gr / vp_layer_mle
static int pred_inner_product(vp_ctx *ctx, const vp_F *r, int n, const F *table, u32 size, vp_F *out) {
    if (n > 31) { ctx->err = "layer too large"; return VP_ELIMIT; }
    if (!ctx->pred_r) {                                  // same scratch as vp_predicates
        u32 max_chunks = 1, max_buckets = 1;
        for (int i = 1; i < ctx->n_layers; ++i) { max_chunks = std::max(max_chunks, ctx->L[i].p_chunks); max_buckets = std::max(max_buckets, ctx->L[i].p_buckets); }
        VPCHK(dalloc(ctx, &ctx->pred_r, (size_t) 3 * 32 + 2));
        VPCHK(dalloc(ctx, &ctx->pred_pool, (size_t) 3 * 2 * ((size_t) 1 << 16)));
        VPCHK(dalloc(ctx, &ctx->pred_part, (size_t) max_chunks));
        VPCHK(dalloc(ctx, &ctx->pred_out, (size_t) max_buckets));
        VPCHK(dalloc(ctx, &ctx->pred_jobs, (size_t) 3));
    }
    if (!ctx->pred_dot) {
        VPCHK(dalloc(ctx, &ctx->pred_dot, (size_t) 1));
        std::vector<BlkMap> mp(128);
        for (u32 b = 0; b < 128; ++b) mp[b] = BlkMap{0, b};
        VPCHK(dupload(ctx, &ctx->pred_map, mp));
    }
    std::vector<F> h(33, f_zero());
    if (n) memcpy(h.data(), r, (size_t) n * sizeof(F));
    h[32] = f_one();
    HIPCHK(hipMemcpyAsync(ctx->pred_r, h.data(), h.size() * sizeof(F), hipMemcpyHostToDevice, ctx->stream));
    BetaJob jb; jb.r = ctx->pred_r; jb.init = ctx->pred_r + 32; jb.n = n; jb.pad = 0; jb.bf = ctx->pred_pool; jb.bs = jb.bf + ((size_t) 1 << (n >> 1));
    DotJob d{}; d.h.bf = jb.bf; d.h.bs = jb.bs; d.h.h1 = n >> 1; d.val = table; d.part = ctx->partials; d.out = ctx->pred_out; d.size = size;
    d.nblk = std::max<u32>(1, std::min<u32>(nblk(size), 128));
    HIPCHK(hipMemcpyAsync(ctx->pred_jobs, &jb, sizeof(jb), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->pred_dot, &d, sizeof(d), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));          // h, jb, d are temporaries
    const u32 bpj = nblk(((u64) 1 << (n >> 1)) + ((u64) 1 << (n - (n >> 1))));
    hipLaunchKernelGGL(k_beta_half_direct, dim3(bpj), dim3(VP_BLOCK), 0, ctx->stream, ctx->pred_jobs, bpj);
    hipLaunchKernelGGL(k_dot_multi, dim3(d.nblk), dim3(VP_BLOCK), 0, ctx->stream, ctx->pred_dot, ctx->pred_map);
    hipLaunchKernelGGL(k_dotfin_multi, dim3(1), dim3(VP_BLOCK), 0, ctx->stream, ctx->pred_dot);
    HIPCHK(hipMemcpyAsync(out, ctx->pred_out, sizeof(F), hipMemcpyDeviceToHost, ctx->stream));
    return check_stream(ctx);
}

int vp_layer_mle(vp_ctx *ctx, int layer, const vp_F *r, int n, vp_F *out) {
    if (!ctx || !ctx->evaluated || layer < 0 || layer >= ctx->n_layers || !out || n != ctx->L[layer].bl || (n && !r)) return VP_EINVAL;
    VP_ENTER(ctx);
    return pred_inner_product(ctx, r, n, ctx->L[layer].val, (u32) ctx->L[layer].size, out);
}

int vp_liu_gr(vp_ctx *ctx, int layer, const vp_F *r_u, const vp_F *const *r_v, const vp_F *s, const vp_F *r_liu, vp_F *out) {
    if (!ctx || !out) return VP_EINVAL;
    const int keep = ctx->opt.prefetch_round1;
    ctx->opt.prefetch_round1 = 0;                                    // only the table is wanted here: no round is coming
    const int rc = vp_liu_init(ctx, layer, r_u, r_v, s);            // the Liu mult table of this layer, as the prover builds it
    ctx->opt.prefetch_round1 = keep;
    if (rc != VP_OK) return rc;
    ctx->sc.phase = 0;                                               // not a sumcheck in progress
    LayerDev &pre = ctx->L[layer - 1];
    if (pre.bl && !r_liu) return VP_EINVAL;
    return pred_inner_product(ctx, r_liu, pre.bl, ctx->tab[0][1], (u32) pre.size, out);
}

int vp_vres(vp_ctx *ctx, const vp_F *r_0, int r_0_size, vp_F *out) {
    if (!ctx || !ctx->evaluated || !out || r_0_size != ctx->L[ctx->n_layers - 1].bl || (r_0_size && !r_0)) return VP_EINVAL;
    VP_ENTER(ctx);
    ctx->rlog.clear();
    VPCHK(stage(ctx, 0, r_0, r_0_size));
    LayerDev &T = ctx->L[ctx->n_layers - 1];
    VPCHK(run_beta_half(ctx, ctx->d_tape, T.bl, ctx->one()));
    hipLaunchKernelGGL(k_vres, dim3(1), dim3(VP_BLOCK), 0, ctx->stream, ctx->bf, ctx->bs, T.bl >> 1, T.val, (u32) T.size,
                       ctx->d_tr, ctx->h_pin + 3);
    VPCHK(check_stream(ctx));
    memcpy(out, ctx->h_pin + 3, sizeof(F));
    return VP_OK;
}

int vp_phase1_init(vp_ctx *ctx, int layer, const vp_F *r_liu, const vp_F *assert_random) {
    if (!ctx || !ctx->evaluated || layer < 1 || layer >= ctx->n_layers || !assert_random) return VP_EINVAL;
    if (ctx->L[layer].bl && !r_liu) return VP_EINVAL;
    VP_ENTER(ctx);
    const u64 off = layer == ctx->n_layers - 1 ? 0 : ctx->rliu_off[layer + 1];
    VPCHK(stage(ctx, off, r_liu, ctx->L[layer].bl));
    VPCHK(stage(ctx, ctx->as_off[layer], assert_random, 1));
    if (ctx->opt.interactive_fast_init) VPCHK(do_phase1_init_fast(ctx, layer));
    else VPCHK(do_phase1_init(ctx, layer, ctx->d_tape + off, ctx->d_tape + ctx->as_off[layer]));
    ctx->rlog_round = 0;
    return round1_prefetch(ctx);
}

int vp_phase2_init(vp_ctx *ctx, int layer, const vp_F *r_u) {
    if (!ctx || !ctx->evaluated || layer < 1 || layer >= ctx->n_layers) return VP_EINVAL;
    if (ctx->L[layer - 1].bl && !r_u) return VP_EINVAL;
    if (ctx->sc.layer != layer || ctx->L[layer].max_dad_bl == -1) { ctx->err = "phase2 out of order"; return VP_EINVAL; }
    VP_ENTER(ctx);
    VPCHK(stage(ctx, ctx->ru_off[layer], r_u, ctx->L[layer - 1].bl));
    if (ctx->opt.interactive_fast_init) VPCHK(do_phase2_init_fast(ctx, layer));
    else VPCHK(do_phase2_init(ctx, layer, ctx->d_tape + ctx->ru_off[layer]));
    ctx->rlog_round = 0;
    return round1_prefetch(ctx);
}

int vp_liu_init(vp_ctx *ctx, int layer, const vp_F *r_u, const vp_F *const *r_v, const vp_F *s) {
    if (!ctx || !ctx->evaluated || layer < 1 || layer >= ctx->n_layers || !s) return VP_EINVAL;
    if (ctx->L[layer - 1].bl && !r_u) return VP_EINVAL;
    VP_ENTER(ctx);
    const int n = ctx->n_layers;
    VPCHK(stage(ctx, ctx->ru_off[layer], r_u, ctx->L[layer - 1].bl));
    VPCHK(stage(ctx, ctx->sig_off[layer], s, n - layer + 1));
    for (int k = layer; k < n; ++k)
        if (ctx->L[k].dad_size[layer - 1] && ctx->L[k].dad_bl[layer - 1]) {
            if (!r_v || !r_v[k]) return VP_EINVAL;
            VPCHK(stage(ctx, ctx->rv_off[k], r_v[k], ctx->L[k].dad_bl[layer - 1]));
        }
    if (ctx->opt.interactive_fast_init) VPCHK(do_liu_init_fast(ctx, layer));
    else VPCHK(do_liu_init(ctx, layer));
    ctx->rlog_round = 0;
    return round1_prefetch(ctx);
}

static int vp_round_impl(vp_ctx *ctx, const vp_F *previous_random, vp_F out_poly[3], int *how);
// The round as the caller sees it: wall time of the call and the ALGORITHMIC bytes of the round (48 B x (L_in + L_out) per table
// family with an add table, 32 B without; round 1 only reads) -> achieved GB/s per sumcheck round (vp_get_round_stats).
int vp_round(vp_ctx *ctx, const vp_F *previous_random, vp_F out_poly[3]) {
    if (!ctx || !previous_random || !out_poly) return VP_EINVAL;
    VP_LOCK(ctx);                                             // no device-wide call in here: other contexts' resident kernels stay where they are
    if (ctx->sc.phase == 0) return VP_EINVAL;
    const auto t0 = std::chrono::steady_clock::now();
    int how = 0;
    const int rc = vp_round_impl(ctx, previous_random, out_poly, &how);
    if (rc == VP_OK && ctx->rlog.size() < (size_t) 1 << 16) {
        const int k = ++ctx->rlog_round;
        const SumcheckState &sc = ctx->sc;
        u64 ent = 0;
        for (int j = 0; j < sc.n_tab; ++j) {
            const u64 v = sc.valid0[j];
            if (k == 1) ent += v;
            else ent += ((v + (1ull << (k - 2)) - 1) >> (k - 2)) + ((v + (1ull << (k - 1)) - 1) >> (k - 1));
        }
        vp_round_stat e;
        e.phase = sc.phase; e.layer = sc.layer; e.round = k; e.how = how; e.tables = sc.n_tab;
        e.bytes = ent * (sc.has_a ? 48 : 32);
        e.us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        ctx->rlog.push_back(e);
    }
    return rc;
}
static int vp_round_impl(vp_ctx *ctx, const vp_F *previous_random, vp_F out_poly[3], int *how) {
    HIPCHK(hipSetDevice(ctx->device));
    F rv; memcpy(&rv, previous_random, sizeof(F));
    // canonical limbs only (header convention); the mailbox of the resident kernel also keeps its sequence tag in bits 61-63 of every word
    if (rv.re >= P61 || rv.im >= P61) { ctx->err = "vp_round: previous_random is not canonical (limb >= 2^61 - 1)"; return VP_EINVAL; }
    if (!ctx->r1_pending && ctx->sc.round >= ctx->sc.total_rounds) { ctx->err = "too many rounds"; return VP_EINVAL; }
    if (ctx->r1_pending) {                                    // round 1 was queued by the init call (it takes no challenge): collect it
        const int how1 = ctx->r1_pending;
        ctx->r1_pending = 0;
        *how = 2;                                             // round 1 was computed behind the init call
        if (how1 == 3) { memcpy(out_poly, ctx->r1_stash, 3 * sizeof(F)); return VP_OK; }      // collected when the kernel was suspended
        if (how1 == 1) {
            VPCHK(tail_wait_resumed(ctx, ctx->tail_seq));
            tail_poly(ctx, out_poly);
            ++ctx->sc.round; ++ctx->st.rounds;
        } else {
            VPCHK(wait_ticket(ctx));
            memcpy(out_poly, ctx->h_pin, 3 * sizeof(F));
        }
        return VP_OK;
    }
    // small rounds: one resident kernel answers every remaining message of the phase through a mailbox (vp_kernels_persist.h)
    if (ctx->tail_lost) { ctx->err = "the resident round kernel was told to leave by another context before it could save its phase"; return VP_EHIP; }
    if (ctx->tail_suspended) VPCHK(tail_resume(ctx, false));  // another context's call (or nothing at all, for longer than the time-out) suspended the phase
    if (ctx->tail_active) {
        *how = 1;                                             // answered by the resident kernel through the mailbox
        int rc = tail_send(ctx, 1, rv);
        if (rc == VP_TAIL_SAVED || (rc == VP_OK && tail_status(ctx) == 5)) {   // the kernel timed out (phase saved) before this message reached it
            VPCHK(tail_resume(ctx, true));
            rc = tail_wait(ctx, ctx->tail_seq);
        }
        if (rc == VP_TAIL_SAVED) { ctx->err = "resident round kernel: timed out twice on one message"; rc = VP_EHIP; }
        VPCHK(rc);
        if (tail_status(ctx) != 0) { ctx->tail_active = false; ctx->err = "persistent round kernel: protocol error"; return VP_EHIP; }
        tail_poly(ctx, out_poly);
        ++ctx->sc.round; ++ctx->st.rounds;
#ifdef VP_TAIL_STAMPS
        { const unsigned long long *t = ctx->h_rep->stamps; fprintf(stderr, "[stamps] sums(before challenge) %llu | wait for challenge %llu | reply chain %llu | fold to regs %llu | write back %llu  (shader clocks)\n",
                  t[5] - t[4], t[0] - t[5], t[1] - t[0], t[2] - t[1], t[3] - t[2]); }
#endif
        return VP_OK;
    }
    if (tail_try_launch(ctx, rv)) {
        *how = 1;
        VPCHK(tail_wait_resumed(ctx, ctx->tail_seq));
        tail_poly(ctx, out_poly);
        ++ctx->sc.round; ++ctx->st.rounds;
        return VP_OK;
    }
    VPCHK(do_round(ctx, nullptr, rv, ctx->d_tr + ctx->n_tr, ctx->h_pin));
    VPCHK(wait_ticket(ctx));
    memcpy(out_poly, ctx->h_pin, 3 * sizeof(F));
    return VP_OK;
}

int vp_finalize(vp_ctx *ctx, const vp_F *previous_random, vp_F *claims, int n_claims) {
    if (!ctx || !previous_random || !claims) return VP_EINVAL;
    VP_LOCK(ctx);
    if (ctx->sc.phase == 0 || n_claims != ctx->sc.n_tab) return VP_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    F rv; memcpy(&rv, previous_random, sizeof(F));
    if (rv.re >= P61 || rv.im >= P61) { ctx->err = "vp_finalize: previous_random is not canonical (limb >= 2^61 - 1)"; return VP_EINVAL; }
    // the live level of a lost phase was in LDS only: the tables in HBM are stale, claims from them would be wrong (vp_round refuses the same way)
    if (ctx->tail_lost) { ctx->err = "the resident round kernel was told to leave by another context before it could save its phase"; return VP_EHIP; }
    if (ctx->tail_suspended) VPCHK(tail_resume(ctx, false));
    if (ctx->tail_active) {                                   // the resident kernel holds the tables: it computes the claims and leaves
        int rc = tail_send(ctx, 2, rv);
        if (rc == VP_TAIL_SAVED || (rc == VP_OK && tail_status(ctx) == 5)) {
            rc = tail_resume(ctx, true);
            if (rc == VP_OK) rc = tail_wait(ctx, ctx->tail_seq);
        }
        ctx->tail_active = false;
        if (rc == VP_TAIL_SAVED) { ctx->err = "resident round kernel: timed out twice on one message"; rc = VP_EHIP; }
        VPCHK(rc);
        if (tail_status(ctx) != 0) { ctx->err = "persistent round kernel: protocol error at finalize"; return VP_EHIP; }
        memcpy(claims, ctx->h_pin + 4, (size_t) n_claims * sizeof(F));
        return VP_OK;
    }
    VPCHK(do_finalize(ctx, nullptr, rv, ctx->d_tr + ctx->n_tr + 3, ctx->h_pin + 4));
    VPCHK(wait_ticket(ctx));
    memcpy(claims, ctx->h_pin + 4, (size_t) n_claims * sizeof(F));
    return VP_OK;
}

static int prove_gkr_fused(vp_ctx *ctx, const vp_F *tape, uint64_t n_tape, uint8_t *transcript, uint64_t *n_written);
static void assign_chains(vp_ctx *ctx);
static void split_layout(vp_ctx *ctx);
bool vp_comm_attached(const vp_ctx *ctx);
bool vp_comm_matches(const vp_ctx *ctx, int rank, int world);

int vp_gkr_sizes(vp_ctx *ctx, uint64_t *n_tape, uint64_t *n_bytes) {
    if (!ctx || ctx->n_layers < 2) return VP_EINVAL;
    if (n_tape) *n_tape = ctx->n_tape;
    if (n_bytes) {
        // index-split proof without a communicator: the call returns the rank's transcript AND its export area (vp_shard_finish)
        u64 n = ctx->n_tr;
        if (ctx->split_lw > 0 && ctx->shard_world > 1 && ctx->plan_path && !vp_comm_attached(ctx)) { split_layout(ctx); n += 3 * ctx->n_exp; }
        *n_bytes = n * sizeof(F);
    }
    return VP_OK;
}

int vp_set_shard(vp_ctx *ctx, int rank, int world) {
    if (!ctx || world < 1 || rank < 0 || rank >= world) return VP_EINVAL;
    if (world > 1 && ctx->n_layers >= 2 && (!ctx->plan_path || ctx->simple_path)) {
        ctx->err = "vp_set_shard: only the launch-plan path shards (unset VP_GKR_PATH)"; return VP_EINVAL;
    }
    if (world > 1 && vp_comm_attached(ctx) && !vp_comm_matches(ctx, rank, world)) { ctx->err = "vp_set_shard: rank/world differ from the attached communicator's"; return VP_EINVAL; }
    VP_ENTER(ctx);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->gkr_graph) { (void) hipGraphExecDestroy(ctx->gkr_graph); ctx->gkr_graph = nullptr; }
    ctx->graph_failed = false;
    free_plan(ctx);
    ctx->shard_rank = rank; ctx->shard_world = world;
    ctx->chain_owner.clear(); ctx->chain_cost.clear();
    if (world <= 1) ctx->split_lw = 0;
    return VP_OK;
}

int vp_set_shard_split(vp_ctx *ctx, int min_log) {
    if (!ctx || min_log < 0 || min_log > 30) return VP_EINVAL;
    VP_ENTER(ctx);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->gkr_graph) { (void) hipGraphExecDestroy(ctx->gkr_graph); ctx->gkr_graph = nullptr; }
    ctx->graph_failed = false;
    free_plan(ctx);
    int lw = 0;
    while ((2 << lw) <= ctx->shard_world) ++lw;                      // largest power of two <= world
    // the 2^lw ranks of a split chain ADD partial sums (each < 2^61) into the same transcript / export slots with a plain u64 all-reduce
    // before anything is reduced mod p: eight addends fit in 64 bits (8 (2^61 - 1) < 2^64), sixteen can wrap — and 2^64 = 8 (mod p) would
    // silently give a wrong transcript.  One node has 8 GPUs; more ranks than that are refused here (chain sharding alone — disjoint
    // slices — has no such bound).
    if (min_log > 0 && lw > 3) { ctx->err = "vp_set_shard_split: at most 8 slices (u64 partial sums of 2^lw ranks must not wrap)"; return VP_ELIMIT; }
    ctx->split_lw = (min_log > 0 && ctx->shard_world > 1) ? lw : 0;
    if (min_log > 0) ctx->split_min_log = std::max(9, min_log);      // a slice keeps at least one fold chunk
    ctx->chain_owner.clear(); ctx->chain_cost.clear();
    return VP_OK;
}

int vp_shard_chains(vp_ctx *ctx, int32_t *owner, double *cost, int capacity, int *n_chains) {
    if (!ctx || ctx->n_layers < 2) return VP_EINVAL;
    split_layout(ctx);                                                // owner -1: the chain is split by index over the ranks
    assign_chains(ctx);
    const int n = (int) ctx->chain_owner.size();
    if (n_chains) *n_chains = n;
    for (int c = 0; c < n && c < capacity; ++c) { if (owner) owner[c] = ctx->chain_owner[c]; if (cost) cost[c] = ctx->chain_cost[c]; }
    return VP_OK;
}

int vp_prove_gkr(vp_ctx *ctx, const vp_F *tape, uint64_t n_tape, uint8_t *transcript, uint64_t capacity,
                 uint64_t *n_written) {
    if (!ctx || !ctx->evaluated || !tape || !transcript || n_tape != ctx->n_tape) return VP_EINVAL;
    { uint64_t need = 0; (void) vp_gkr_sizes(ctx, nullptr, &need); if (ctx->opt.debug & 1) fprintf(stderr, "[vp] vp_prove_gkr: capacity %llu need %llu\n", (unsigned long long) capacity, (unsigned long long) need); if (capacity < need) return VP_EINVAL; }
    if (!ctx->simple_path) return prove_gkr_fused(ctx, tape, n_tape, transcript, n_written);
#ifndef VP_TEST_DRIVERS
    ctx->err = "the per-round driver is not in this build (vp_test_drivers)";        // unreachable: vp_create refuses VP_GKR_PATH in the product build
    return VP_EINVAL;
#else
    VP_ENTER(ctx);
    const int n = ctx->n_layers;
    ctx->st.launches = 0; ctx->st.rounds = 0; ctx->ev_used = 0;
    HIPCHK(hipMemcpyAsync(ctx->d_tape, tape, n_tape * sizeof(F), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->add_term(), 0, sizeof(F), ctx->stream));
    F *tr = ctx->d_tr;
    u64 pos = 0;
    const F zero = f_zero();
    {   // Vres (verifier.cpp:151)
        LayerDev &T = ctx->L[n - 1];
        VPCHK(run_beta_half(ctx, ctx->d_tape, T.bl, ctx->one()));
        hipLaunchKernelGGL(k_vres, dim3(1), dim3(VP_BLOCK), 0, ctx->stream, ctx->bf, ctx->bs, T.bl >> 1, T.val,
                           (u32) T.size, tr + pos, (F *) nullptr);
        count_launch(ctx);
        pos += 1;
    }
    for (int i = n - 1; i >= 1; --i) {
        const int pbl = ctx->L[i - 1].bl;
        const F *ru = ctx->d_tape + ctx->ru_off[i];
        // phase 1 (verifier.cpp:191-229)
        VPCHK(do_phase1_init(ctx, i, rliu_ptr(ctx, i), ctx->d_tape + ctx->as_off[i]));
        for (int j = 0; j < pbl; ++j) { VPCHK(do_round(ctx, j ? ru + (j - 1) : ctx->zero(), zero, tr + pos, nullptr)); pos += 3; }
        VPCHK(do_finalize(ctx, pbl ? ru + (pbl - 1) : ctx->zero(), zero, tr + pos, nullptr));
        pos += 1;
        // phase 2 (verifier.cpp:231-270)
        const int mdb = ctx->L[i].max_dad_bl;
        if (mdb != -1) {
            const F *rv = ctx->d_tape + ctx->rv_off[i];
            VPCHK(do_phase2_init(ctx, i, ru));
            for (int j = 0; j < mdb; ++j) { VPCHK(do_round(ctx, j ? rv + (j - 1) : ctx->zero(), zero, tr + pos, nullptr)); pos += 3; }
            VPCHK(do_finalize(ctx, mdb ? rv + (mdb - 1) : ctx->zero(), zero, tr + pos, nullptr));
            pos += i;
        }
        // Liu (verifier.cpp:272-337)
        const F *rl = ctx->d_tape + ctx->rliu_off[i];
        VPCHK(do_liu_init(ctx, i));
        for (int j = 0; j < pbl; ++j) { VPCHK(do_round(ctx, j ? rl + (j - 1) : ctx->zero(), zero, tr + pos, nullptr)); pos += 3; }
        VPCHK(do_finalize(ctx, pbl ? rl + (pbl - 1) : ctx->zero(), zero, tr + pos, nullptr));
        pos += 1;
    }
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    if (pos != ctx->n_tr) { ctx->err = "internal: transcript size"; return VP_EINVAL; }
    HIPCHK(hipMemcpyAsync(transcript, tr, pos * sizeof(F), hipMemcpyDeviceToHost, ctx->stream));
    VPCHK(check_stream(ctx));
    float ms = 0;
    hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    ctx->st.gkr_ms = ms;
    ctx->st.fold_ms = 0; ctx->st.fold_bytes = 0; ctx->st.fold_launches = ctx->ev_used;
    if (ctx->profiling) prof_collect(ctx);
    for (size_t e = 0; e < ctx->ev_used; ++e) {
        float t = 0;
        hipEventElapsedTime(&t, ctx->ev_pool[e].a, ctx->ev_pool[e].b);
        ctx->st.fold_ms += t; ctx->st.fold_bytes += ctx->ev_pool[e].bytes;
    }
    if (n_written) *n_written = pos * sizeof(F);
    return VP_OK;
#endif
}

int vp_get_stats(vp_ctx *ctx, vp_stats *out) {
    if (!ctx || !out) return VP_EINVAL;
    VP_LOCK(ctx);
    *out = ctx->st;
    return VP_OK;
}
int vp_get_resident_resumes(const vp_ctx *cctx, uint64_t *n) {
    if (!cctx || !n) return VP_EINVAL;
    vp_ctx *ctx = const_cast<vp_ctx *>(cctx);
    VP_LOCK(ctx);
    *n = ctx->tail_resumes;
    return VP_OK;
}
int vp_get_round_stats(vp_ctx *ctx, vp_round_stat *out, int capacity, int *n) {
    if (!ctx || !n || (capacity > 0 && !out)) return VP_EINVAL;
    VP_LOCK(ctx);                                             // a concurrent vp_round on this context appends to (and may reallocate) rlog
    *n = (int) ctx->rlog.size();
    for (int i = 0; i < *n && i < capacity; ++i) out[i] = ctx->rlog[i];
    return VP_OK;
}
int vp_set_deferred(vp_ctx *ctx, int on) {
    if (!ctx) return VP_EINVAL;
    VP_ENTER_Q(ctx);
    ctx->deferred = on ? 1 : 0;                               // (what is pending stays pending: vp_flush, or the next entry point that cannot defer, finishes it)
    return VP_OK;
}
int vp_pending(vp_ctx *ctx, int *n) {
    if (!ctx || !n) return VP_EINVAL;
    VP_LOCK(ctx);
    *n = (int) ctx->pending.size();
    return VP_OK;
}
int vp_flush(vp_ctx *ctx, int count) {
    if (!ctx) return VP_EINVAL;
    VP_ENTER_Q(ctx);
    return flush_pending(ctx, count < 0 ? (size_t) -1 : (size_t) count);
}
int vp_phase_ms(vp_ctx *ctx, double out[5]) {
    if (!ctx || !out) return VP_EINVAL;
    VP_LOCK(ctx);
    for (int i = 0; i < 5; ++i) out[i] = ctx->phase_ms[i];
    return VP_OK;
}
int vp_commit_private_state(vp_ctx *ctx, uint64_t *epoch, int *valid) {
    if (!ctx || !epoch || !valid) return VP_EINVAL;
    VP_LOCK(ctx);
    *epoch = ctx->private_epoch;
    *valid = ctx->pc_private_done && !ctx->pcs ? 1 : 0;
    return VP_OK;
}
int vp_set_profiling(vp_ctx *ctx, int level) {
    if (!ctx) return VP_EINVAL;
    ctx->profiling = level;
    return VP_OK;
}
int vp_get_launch_stats(vp_ctx *ctx, vp_launch_stat *out, int capacity, int *n) {
    if (!ctx || !n || (capacity > 0 && !out)) return VP_EINVAL;
    VP_LOCK(ctx);
    *n = (int) ctx->lstats.size();
    for (int i = 0; i < *n && i < capacity; ++i) out[i] = ctx->lstats[i];
    return VP_OK;
}
const char *vp_kernel_name(int kind) {
    static const char *names[VP_K_COUNT] = {"k_beta_half_direct", "k_light_multi", "k_chunks_multi", "k_combine_multi", "k_dot_multi", "k_dotfin_multi",
        "k_sumfold3b_gen_multi", "k_sumfold3b_multi", "k_seg_multi", "k_emit_multi", "k_fixup", "k_ntt_split", "k_ntt_lds", "k_ntt_unsplit",
        "k_leaf_hash", "k_merkle", "k_pc_pointwise", "k_fri_fold", "k_round", "k_ntt8_cols", "k_ntt8_rows"};
    return (kind >= 0 && kind < VP_K_COUNT) ? names[kind] : "?";
}

int vp_test_field(vp_ctx *ctx, int op, const vp_F *a, const vp_F *b, vp_F *out, uint64_t n) {
    if (!ctx || !a || !b || !out || op < 0 || op > 3) return VP_EINVAL;
    if (n == 0) return VP_OK;
    VP_ENTER(ctx);
    F *da = nullptr, *db = nullptr, *dout = nullptr;
    HIPCHK(hipMalloc((void **) &da, n * sizeof(F)));
    HIPCHK(hipMalloc((void **) &db, n * sizeof(F)));
    HIPCHK(hipMalloc((void **) &dout, n * sizeof(F)));
    HIPCHK(hipMemcpy(da, a, n * sizeof(F), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(db, b, n * sizeof(F), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_test_field, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, ctx->stream, op, da, db, dout, (u64) n);
    int rc = check_stream(ctx);
    if (rc == VP_OK && hipMemcpy(out, dout, n * sizeof(F), hipMemcpyDeviceToHost) != hipSuccess) rc = VP_EHIP;
    (void) hipFree(da); (void) hipFree(db); (void) hipFree(dout);
    return rc;
}

int vp_test_beta(vp_ctx *ctx, const vp_F *r, int n, const vp_F *init, vp_F *out) {
    if (!ctx || !init || !out || n < 0 || n > 28 || (n && !r)) return VP_EINVAL;
    VP_ENTER(ctx);
    F *dr = nullptr, *dbf = nullptr, *dbs = nullptr, *dout = nullptr;
    const size_t half = (size_t) 1 << ((n + 1) / 2);
    HIPCHK(hipMalloc((void **) &dr, (n + 2) * sizeof(F)));
    HIPCHK(hipMalloc((void **) &dbf, half * sizeof(F)));
    HIPCHK(hipMalloc((void **) &dbs, half * sizeof(F)));
    HIPCHK(hipMalloc((void **) &dout, sizeof(F) << n));
    if (n) HIPCHK(hipMemcpy(dr, r, n * sizeof(F), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dr + n, init, sizeof(F), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_beta_half, dim3(1), dim3(VP_BLOCK), 0, ctx->stream, dr, n, dr + n, dbf, dbs);
    hipLaunchKernelGGL(k_beta_expand, dim3(grid_for(1ull << n)), dim3(VP_BLOCK), 0, ctx->stream, dbf, dbs, n >> 1,
                       (u32) (1u << n), dout);
    int rc = check_stream(ctx);
    if (rc == VP_OK && hipMemcpy(out, dout, sizeof(F) << n, hipMemcpyDeviceToHost) != hipSuccess) rc = VP_EHIP;
    (void) hipFree(dr); (void) hipFree(dbf); (void) hipFree(dbs); (void) hipFree(dout);
    return rc;
}

}  // extern "C"

#include "vpgpu_batched.inc"
#include "vpgpu_pc.inc"
#include "vpgpu_pc_shard.inc"
#include "vpgpu_fftgkr.inc"
