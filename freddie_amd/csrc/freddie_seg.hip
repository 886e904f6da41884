// freddie_seg.hip -- the HOST side of libfreddie_seg.so, the gfx950 (MI355X) library of the canonical-segmentation path: contexts,
// slabs and arenas, the launch sequence of a run (enqueue_run), the sized first run, uploads, results, and the C-ABI of
// include/freddie_seg.h.  The kernels live in the stage families' translation units (seg_front / seg_problems /
// seg_score_arena / seg_score_fused + seg_solve16|32|60 / seg_tail / seg_upload .hip; shared definitions seg_common.h,
// declarations seg_kernels.h); freddie_amd/build.py compiles the units side by side.
//
// What each kernel computes is defined by the reference's py/freddie_segment.py (cited per
// kernel as file:line); how it computes it is specific to this implementation:
//   * a batch of independent partitions lives in HBM as flat CSR arrays (include/freddie_seg.h);
//   * every data-dependent size (candidates, DP problems, final positions, label bytes) is
//     produced and consumed on the device; the host only reads one small status record at the
//     end of a run and grows an arena + re-runs when a capacity was exceeded, so the steady
//     state has no host synchronisation inside the pipeline;
//   * the interval-scoring stage never materialises the reference's (N+1)xR uint32 coverage
//     matrix: per DP problem it derives, for 64 reads at a time, the window-local coverage
//     prefix of each read from its exon list, turns the n*(n-1)/2 pair tests into 1-bit planes
//     in LDS, and counts out(i,j,k) with AND + popcount into an LDS-resident table.
//
// No CPU fallback exists in this library: without a GPU fseg_create() fails.
#include "seg_kernels.h"

using namespace fseg;

namespace {

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// A DevBuf is a view into one of the context's slabs (or, for the few buffers with a life of their own, an allocation):
// the buffers of a batch are carved out of three device allocations -- inputs (one host-to-device copy fills it),
// position-sized work arrays, data-dependent arenas -- so a new batch costs no allocator call unless it is larger than
// every batch before it.
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};
struct Slab {
    void *p = nullptr;
    size_t cap = 0;
};
struct HostBuf {     // pinned host memory
    void *p = nullptr;
    size_t cap = 0;
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};
// offsets of consecutive 256-byte-aligned buffers inside a slab: add() them all, reserve the slab, then bind()
struct Carve {
    struct Item { DevBuf *b; size_t off, bytes; };
    std::vector<Item> items;
    size_t total = 0;
    size_t add(DevBuf &b, size_t bytes) {
        const size_t off = total;
        items.push_back(Item{&b, off, bytes});
        total = (off + bytes + 255) & ~(size_t)255;
        return off;
    }
    void bind(const Slab &s) const {
        for (const Item &it : items) { it.b->p = static_cast<char *>(s.p) + it.off; it.b->cap = it.bytes; }
    }
};

enum Stage { ST_HIST, ST_SMOOTH, ST_THRESHOLD, ST_CANDIDATES, ST_FIX, ST_SCORE_PREP, ST_SCORE, ST_DP, ST_REFINE, ST_FINAL, ST_LABEL, ST_COUNT,
             ST_GRAPH_PRE = ST_COUNT, ST_GRAPH_POST, ST_REPORTED };
// Plain launches (the first run of a batch, or FSEG_NO_GRAPH=1): every stage is bracketed by its own pair of events.
// Graph replay with profiling: graph(before scoring) | events around plain launches of the scoring kernel |
// graph(after); the two graphs are reported as graph_pre / graph_post.
const char *kStageNames[ST_REPORTED] = {"histogram", "smooth", "threshold", "candidates", "fix_split", "scoring_prep",
                                        "interval_scoring", "dp", "refine", "final_positions", "labels", "graph_pre",
                                        "graph_post"};

// segments of one run (bit mask of enqueue_run)
enum : unsigned {
    SEG_PRE1 = 1u,     // status reset, histogram .. problem ranges and the problem scan: every arena size is known after it
    SEG_PRE2 = 2u,     // problem list, pair thresholds, window coverage
    SEG_SCORE = 4u,    // the interval-scoring kernels
    SEG_POST1 = 8u,    // DP, refinement, final positions, label columns: the label arena's size is known after it
    SEG_POST2 = 16u,   // label arena fill + per-read labels
    SEG_STATUS = 32u,  // status record to the host
    SEG_ALL = 63u,
};

}  // namespace

hipError_t fseg_sort_pairs(void *tmp, size_t *tmp_bytes, const unsigned long long *keys_in, unsigned long long *keys_out,
                           const int *vals_in, int *vals_out, size_t n, unsigned end_bit, hipStream_t stream);   // freddie_seg_sort.hip

struct fseg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    bool have_params = false, have_batch = false, ran = false, pending = false;
    bool fetched = false;        // the results of the last run are in the pinned result buffers
    fseg_params P{};
    std::vector<double> w_main, w_refine, h_table;
    // batch metadata (host)
    int n_part = 0;
    i64 K = 0, R = 0, I = 0, NPOS = 0, LANES = 0;
    i64 max_part_pos = 0;      // positions of the batch's largest partition (k_thr_part takes partitions of up to kThrPartMaxChunks * 8192)
    int thr_part = -1;         // FSEG_THR_PART=0 / 1: the threshold per partition by one workgroup (k_thr_part) never / whenever possible; -1: batches of many partitions
    i64 max_part_lanes = 0;    // reads of the batch's largest partition (what bounds a DP sum: 32-bit keys below 2^18)
    int n_tiles = 0;
    bool expanded = false;
    std::vector<i64> part_iv_off, part_rep_off;
    // the three slabs and the buffers that live outside them
    Slab slab_in, slab_pos, slab_arena;
    HostBuf h_stage;             // pinned image of the input slab's uploaded part
    HostBuf h_res;               // pinned results: final offsets | final positions | label offsets | labels
    size_t res_off[4] = {0, 0, 0, 0};
    std::vector<i64> res_pfo;    // part_final_off gathered from the per-interval offsets
    DevBuf d_labels;             // label arena (sized after the final positions are known)
    DevBuf d_packed;             // the same at two bits per label, for the trip to the host
    int fetched_packed = -1;     // what the pinned result buffer holds: -1 nothing, 0 label bytes, 1 packed labels
    // Round 5: the labels are WRITTEN at two bits each (k_label_reads ORs the codes of a rep's non-'0' labels into d_packed, a word
    // at a time) whenever no column's default can be '2' (threshold_rate < 1: a read without coverage is never ambiguous) -- the
    // 75 MB byte arena of a 250 k-read batch is then neither filled (15 us) nor packed (25 us); fseg_results / fseg_download unpack
    // it on demand.  FSEG_LABEL_BYTES=1 keeps the byte arena as the primary form (and threshold_rate = 1 always does).
    // "No column's default can be '2'" is a property of the TABLE, not of the rate: smooth_threshold() rounds to two decimals, so a rate
    // in about (0.99972, 1) has a last entry of 1.0 (0.9999: segments of exactly 107 positions have h = 1, l = 0, lo = -1 and every
    // read is '2' there -- golden e_tau9999_len107).  label_has2 is computed from the table by fseg_set_params.
    bool label_packed_ok = true;
    bool label_has2 = false;         // some segment length has h >= 1 (lo < 0): a read without coverage is '2' there -> byte arena
    bool run_label_packed = false;   // the resident run's labels live in d_packed
    bool labels_unpacked = false;    // ... and d_labels holds their byte form (made on demand)
    DevBuf d_sort_tmp;
    // device buffers: inputs (slab_in, uploaded)
    DevBuf d_part_iv_off, d_part_rep_off, d_part_lane_off, d_iv_start, d_iv_end, d_pos_off, d_iv_part,
        d_rep_exon_off, d_rep_weight, d_ex_ts, d_ex_te, d_tile_desc, d_iv_tile0, d_blk_iv0, d_rb_part, d_rb_r0, d_rb_sum, d_rb_base, d_rb_max, d_rb_cmax,
        d_hc_part, d_hc_p0, d_hc_n, d_hc_glo, d_hc_ghi;
    // slab_in, derived on the device by the upload
    DevBuf d_lane_ex, d_lane_start, d_lane_pmax, d_hc_llo, d_hc_lhi, d_key_a, d_key_b, d_val_a, d_val_b, d_rep_last;
    DevBuf d_lane_lx, d_lex, d_rb_esum, d_rb_ebase;   // the exons again as one (ts, te) stream in lane order, and every lane's range in it
    i64 max_rep_exons = 0;       // most exons of one rep in the resident batch
    DevBuf d_w_main, d_w_refine, d_h_table, d_thr_tab;     // parameter tables (own allocations)
    // device buffers: position-sized (slab_pos)
    DevBuf d_bits;               // the three flag masks (Y > 0, candidate, final position), a bit per position each, cleared per run
    DevBuf d_y_raw, d_y, d_v, d_scan_state, d_bsum, d_bsum_side, d_g, d_pk, d_pf, d_kp;
    int n_hist_chunks = 0;
    DevBuf d_voff, d_chunk_off, d_csum, d_mean, d_thr, d_label_off, d_part_has2;
    int n_rep_blocks = 0;
    // candidate-sized (slab_pos)
    DevBuf d_cand_off, d_cand_y, d_fixed0, d_added, d_fixed, d_chosen, d_final_off, d_final_y, d_final_pos, d_final_iv, d_col_thr,
        d_col_zero;
    DevBuf d_tile_defer;        // per smoothing tile: start of the plateau that reaches the tile's end (-1: none)
    DevBuf d_blk_pre, d_tile_tot, d_seg_iv, d_seg_prev, d_rseg_c, d_cand_pn, d_cand_ll, d_cand_ln, d_cand_wide, d_prob_bs;
    // problems / arenas (slab_arena)
    DevBuf d_prob_iv, d_prob_start, d_prob_n, d_prob_pair_off, d_prob_tri_off, d_prob_flags, d_prob_chain,
        d_prob_cov_off, d_prob_lane_lo, d_prob_lane_n;
    DevBuf d_dp_items, d_solve_items, d_solve_desc, d_prob_desc, d_work_pc, d_cls_items, d_work_active, d_pair_thr, d_amb, d_out, d_cov;
    // hand-over arena between k_solve<.., SPLIT> and k_dpw (dpx_slot_bytes per problem of the three solve lists), laid out by
    // alloc_arenas() for the counts it knew: a launch takes the split path only for a list that fits what was laid out
    DevBuf d_wide_items;         // per solve list: the list positions of the problems that see more than kFuseLanes reads (k_prob_emit)
    DevBuf d_wide_all;           // ... of the three lists together (positions from the first list's start)
    DevBuf d_dpx;
    i64 dpx_base[3] = {0, 0, 0}, dpx_stride[3] = {0, 0, 0}, dpx_n[3] = {0, 0, 0};
    int dpx_nm = 0, dpx_cnt[3] = {1, 1, 1};
    static constexpr i64 kWideOneMax = 256;   // plan 'W' takes a batch's wide problems in one launch when there are at most this many
    int split_dp = 7;           // FSEG_SPLIT_DP: bit 0 / 1 / 2 = the small / mid / large class hands its DPs to k_dpw (0: the DP stays the tail of k_solve's workgroups);
                                // bit 3 = the large class's k_dpw by one wave a problem (as the other classes') instead of eight
    i64 prob_cap = 0, work_cap = 0, pair_cap = 0, tri_cap = 0, label_cap = 0, chunk_cap = 0, cov_cap = 0;
    DevBuf d_status, d_prep, d_tacc;
    DevBuf d_sync;               // SyncWords: the scoring stage's device-side fork / join (k_wait_word)
    unsigned sync_gen = 0;       // generation of the last stage enqueued with device-side waiters
    // What a waiter waits at most, in ticks of the 100 MHz clock, per 2^18 reads of the batch: 10 ms, at most 20 (FSEG_SYNC_TICKS; tests
    // force a time-out with 1).  Round 5's 20 ms DID run out, four times in one 8-context trace -- for two reasons, both removed in
    // round 6: a later context's side stream could share its own main stream's hardware queue (probe_side_queues), and two contexts that
    // started together could both fork (claim_device).  What is left is honest waiting: the fork waiters sleep through k_fix ..
    // k_prob_emit (median 49-68 us), the join waiter through what the side chains need beyond the main stream's -- plus whatever is in
    // front of the owner's kernels in its hardware queue: other contexts' kernels (168 and 235 us in two traced 8-context runs) or a
    // large copy to pageable memory (40 MB of a test's taps hold a queue for milliseconds: one time-out at 2 ms in 1 600 forked runs of
    // tests/test_gpu_contexts_stress.py; 200 us, the figure first tried, lost four to six waits per bench run).  A time-out costs the
    // limit AND a rerun of the batch, and with the two causes above gone a waiter only ever waits for work that completes: the limit
    // is the exit every launch must have, not a schedule.
    unsigned sync_ticks = 1000000u;
    unsigned sync_timeouts = 0;  // runs of this context that a waiter gave up on (each was redone with events); FSEG_TAP_SYNC[6]
    unsigned forked_runs = 0;    // runs of this context that owned the device (side streams in use); FSEG_TAP_SYNC[7]
    bool run_events_only = false;   // this run: no device-side waiters (it is the rerun after a timeout)
    bool dev_sync = true;        // FSEG_DEV_SYNC=0: the stage's side streams are forked and joined with events only (also after a waiter timed out)
    Status *h_status = nullptr;   // pinned
    PrepStatus *h_prep = nullptr; // pinned
    bool prep_checked = false;   // the upload's device-side validation has been read back
    bool counted_in_flight = false, counted_live = false;
    // Which side streams may carry a device-side waiter: those whose hardware queue is NOT the main stream's (probe_side_queues).
    // The runtime multiplexes a process's streams onto four hardware queues as they are created: the first context of a device makes
    // its four streams back to back (queues 1 2 3 4: the third side stream shares the main stream's), a later context makes its side
    // streams when it first forks and they land wherever the count of streams then points -- round 5's 8-context trace had four
    // k_wait_word of 20 ms during which NOTHING else ran on the GPU: a waiter in front of its own main stream.
    bool side_probed = false;
    bool side_ok[3] = {false, false, false};
    unsigned probe_gen = 0;
    bool owns_device = false;    // this context holds the device's owner word: the only one that may fork over side streams (claim_device)
    int hsa_agent = -1;         // index into the HSA agent table (-1: not looked up yet, -2: unavailable)
    hsa_signal_t hsa_sig = {0};
    int hsa_cpu = 0;            // index of the CPU agent that owns the pinned result buffer ...
    const void *hsa_dst_base = nullptr;   // ... as asked for this allocation of it
    bool profiling = false;
    bool profile_all = true;    // false (fseg_set_profiling(ctx, 2)): only the interval-scoring stage is bracketed by events
    bool have_huge = false;      // the batch has a problem with more than kNMax candidates: launch the huge kernels
    int nm_giant = 0;            // ... with more than kNHuge: its size (rounded up: the LDS carve-up and scratch piece of the giant kernels); 0: none
    DevBuf d_giant;              // their scratch: kGiantWgs pieces of giant_scratch_bytes(nm_giant) (own allocation)
    bool dp_wide_counts = false; // some problem sees >= 65536 reads: DP stages 32-bit counts
    int nm_big = kNMax;         // LDS carve-up of the big-problem kernels: largest problem of the batch, rounded up
    bool small_batch = false;
    bool tiny_on = false;       // many problems (> tiny_from): problems with <= kTiny candidates go to k_tiny
    bool use_graph = true;      // replay the launch sequence as hipGraphs from the second run of a batch on (FSEG_NO_GRAPH=1 disables)
    // ... when the run keeps to ONE stream (other contexts' batches in flight, small batches, FSEG_NO_FORK=1).  A run that
    // forks is replayed as plain launches: launching a graph with cross-stream edges costs the host 0.5-0.6 ms (ROCm 7.2,
    // the 250 k-read batch: 0.51 ms against 0.17 ms for the ~60 plain launches and 0.03 ms for the one-stream graph), more than
    // the GPU needs for the stages before the scoring stage -- the side streams' packets arrive late and the replay takes
    // 1.13 ms instead of 0.71 (profiles/r04_replay_modes.txt; the forked graph was a switch, FSEG_GRAPH_FORK, until round 6).
    bool run_plain = false;     // this run: plain launches (set by fseg_run)
    bool run_linear = false;    // this run: one stream whatever else holds (a graph is being captured)
    bool use_sized = true;      // the first run of a batch stops twice to size its arenas exactly (FSEG_NO_SIZED=1: guess, and re-run on overflow)
    static constexpr i64 kProbSelfMax = 4 * kProbBlock;   // candidates up to which it does
    bool prob_self_scan = false; // k_prob_emit adds up the candidate blocks itself (few candidates)
    i64 scan_single_max = 512;  // scan blocks up to which the compactions use the single-pass look-back scan (FSEG_SCAN_SINGLE_MAX)
    bool force_scan_stall = false;   // FSEG_FORCE_SCAN_STALL=1 (tests): the first look-back run reports a stall
    hipGraph_t graph[2] = {nullptr, nullptr};            // [0] whole pipeline, or before / after scoring when profiling
    hipGraphExec_t graph_exec[2] = {nullptr, nullptr};
    bool profile_plain = false;   // fseg_set_profiling(3)
    int n_graphs = 0;
    bool last_sized = false;     // the pending run was launched stage by stage (per-stage events valid)
    hipEvent_t ev_b[ST_COUNT] = {}, ev_e[ST_COUNT] = {};
    hipEvent_t ev_g[4] = {};
    float stage_ms[ST_REPORTED] = {};
    // Independent kernels of one run (threshold | candidates, the scoring size classes, the DP classes) go to side
    // streams (pair thresholds | coverage was tried too: the branch cost more than the overlap gave) between a fork and a join, so a captured run becomes a graph with parallel
    // branches (FSEG_NO_FORK=1 keeps everything on the one stream).
    // (four streams in all: the runtime multiplexes a process's streams onto four hardware queues, and two streams that
    // share one run one after the other -- with seven streams the threshold and candidate chains stopped overlapping)
    static constexpr int kSide = 3, kForkEvents = 32;
    hipStream_t side[kSide] = {};
    hipEvent_t fj[kForkEvents] = {};
    bool use_fork = true;
    bool force_key64 = false;
    bool wide_by_seen = false;  // FSEG_WIDE_BY_SEEN=1 (tests)
    char score_plan[32] = "gM|W|hB|gST";   // FSEG_SCORE_PLAN (see enqueue_run; anything that does not name each class once = one stream)
    bool use_wave = true;       // FSEG_NO_WAVE=1: k_tiny / k_solve<16> instead of the wave kernels (k_wave)
    bool use_fuse = true;       // FSEG_NO_FUSE=1: no problem goes to k_solve (everything that is not tiny takes the arena path)
    // Reads the widest problem of a batch may see for the batch's problems to be solved whole (k_solve / k_wave) instead of going
    // through the arena path; FSEG_FUSE_LANES=1023 admits batches of 1 000-read partitions (their widest problems see ~300 reads:
    // the 16-bit-counter instances of k_solve take those).  Measured on 250 x 1 000-read batches (config3 / config5): one batch
    // alone on the GPU is quicker solved whole (0.82 / 0.76 against 0.85 / 0.82 ms per replayed batch), eight contexts taking
    // turns are not (config3: 249 against 301 M reads/s; config5: equal) -- the few wide problems are long, thin launches -- so
    // the default keeps such batches on the arena path.
    int fuse_lanes = kFuseLanesDefault;
    bool wide_solve = true;     // some problem of the batch sees more than kFuseLanes reads: the 16-bit-counter instances of k_solve run too
    bool fuse_on = true;        // this batch's problems go to k_solve: decided per batch -- when its widest problem sees at most
                                // kFuseLanes reads, i.e. all of them qualify (measured: a batch of 500-read partitions gains 8 %, while
                                // batches whose problems straddle the limit run both paths side by side and lose up to 8 %)
    // what the lists of the resident batch hold (read from the status record; exact once the batch has run or been sized):
    // launches over an empty list are skipped
    bool counts_known = false;
    i64 n_solve[3] = {0, 0, 0}, n_cls_work[4] = {0, 0, 0, 0}, n_dp_cls[3] = {0, 0, 0}, n_arena_prob = 0, n_tiny = 0;
    i64 n_wide[3] = {0, 0, 0};  // of n_solve: problems that need the 16-bit-counter instances
    i64 max_ln = 0;             // reads the widest problem of the batch sees
    i64 tiny_from = 256;        // problems above which k_tiny is used (FSEG_TINY_FROM; tests force 0)
    bool trace = false;         // FSEG_TRACE=1: phase timers of upload / run on stderr
    bool force_global_sort = false;   // FSEG_GLOBAL_SORT=1 (tests): the batch-wide radix sort whatever the partition sizes
};

namespace {

int fail(fseg_ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    // A failed HIP call leaves its code as the calling thread's "last error": consumed here, or the next -- successful --
    // upload or run on this host thread would trip over it at its hipGetLastError() check.
    if (code == FSEG_ERR_HIP) (void)hipGetLastError();
    return code;
}
thread_local std::string g_create_error;      // fseg_create has no context to put its message in: the calling thread's own

#define HIP_TRY(c, expr)                                                                         \
    do {                                                                                         \
        hipError_t e__ = (expr);                                                                 \
        if (e__ != hipSuccess) return fail((c), FSEG_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e__)); \
    } while (0)
#define TRY(expr) do { int rc__ = (expr); if (rc__) return rc__; } while (0)

// Contexts of one device that have a batch in flight (uploaded or running, results not yet complete).  A context forks its
// run over side streams only when it is alone on the device: with several contexts taking turns (the CLI, the benchmark's
// timed steps) the other contexts' kernels already fill what one stream leaves idle, and the extra streams only crowd the
// hardware queues (three contexts: 1.44 -> 1.28 ms per 250 k-read batch without them).
// Forking (and with it the device-side waiters of the scoring stage) needs more than that count: two contexts that start together
// could both read "alone", and then each parks one-wave spinners on side streams that share hardware queues with the other's
// main stream -- a cycle that only the waiters' time limit broke (round 5's 8-context trace: four 20 ms k_wait_word).  So a run
// forks only after it has CLAIMED the device: one compare-and-swap on the device's owner word, taken by fseg_run when nobody
// else is in flight, given back when the run has completed (finish_run).  Losers, and whoever arrives while the word is held,
// keep to one stream.  With one forking context per device a waiter can only sit in front of kernels that do not feed it.
static bool g_ablate_dp = false;     // (diagnostic build FSEG_ABLATE_STAGE only: the k_dpw launches are left out)
static std::atomic<int> g_in_flight[64];
static std::atomic<int> g_live[64];
static std::atomic<const fseg_ctx *> g_owner[64];
static void set_in_flight(fseg_ctx *c, bool on);
static bool others_in_flight(const fseg_ctx *c);
static bool would_fork(const fseg_ctx *c);
static void claim_device(fseg_ctx *c);
static void release_device(fseg_ctx *c);

struct Tick {
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    double ms() { auto n = std::chrono::steady_clock::now(); double d = std::chrono::duration<double, std::milli>(n - t).count(); t = n; return d; }
};

// own allocation (parameter tables, label arena, sort scratch): grows, never shrinks
int ensure(fseg_ctx *c, DevBuf &b, size_t bytes) {
    if (bytes <= b.cap && b.p) return FSEG_OK;
    if (b.p) HIP_TRY(c, hipFree(b.p));
    b.p = nullptr; b.cap = 0;
    size_t want = bytes < 256 ? 256 : bytes + bytes / 4;          // headroom: batches of a run are of similar, not equal, size
    HIP_TRY(c, hipMalloc(&b.p, want));
    b.cap = want;
    return FSEG_OK;
}
// keep: the slab's contents must survive growing it (device-to-device copy of the old allocation)
int reserve(fseg_ctx *c, Slab &s, size_t bytes, bool keep = false) {
    if (bytes <= s.cap && s.p) return FSEG_OK;
    const size_t want = bytes + bytes / 4 + 4096;
    void *np = nullptr;
    HIP_TRY(c, hipMalloc(&np, want));
    if (s.p) {
        if (keep) HIP_TRY(c, hipMemcpyAsync(np, s.p, s.cap, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipFree(s.p));
    }
    s.p = np; s.cap = want;
    return FSEG_OK;
}
int reserve_host(fseg_ctx *c, HostBuf &h, size_t bytes) {
    if (bytes <= h.cap && h.p) return FSEG_OK;
    if (h.p) HIP_TRY(c, hipHostFree(h.p));
    h.p = nullptr; h.cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    HIP_TRY(c, hipHostMalloc(&h.p, want, hipHostMallocDefault));
    h.cap = want;
    return FSEG_OK;
}
template <typename T>
int upload_vec(fseg_ctx *c, DevBuf &b, const T *src, size_t n) {
    int rc = ensure(c, b, n * sizeof(T));
    if (rc) return rc;
    if (n) HIP_TRY(c, hipMemcpyAsync(b.p, src, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    return FSEG_OK;
}

// A blocking copy on the context's OWN stream.  (Never hipMemcpy: it runs on the legacy stream, which must not be touched while another
// thread of the process -- another context -- is capturing a graph: "operation would make the legacy stream depend on a capturing blocking
// stream".  Found by tests/test_gpu_contexts_stress.py in round 6: taps, and the side-stream probe of a context that forks for the first
// time while another replays on one stream.)
hipError_t copy_sync(fseg_ctx *c, void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t q = nullptr) {
    if (!q) q = c->stream;
    hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, q);
    if (e == hipSuccess) e = hipStreamSynchronize(q);
    return e;
}

int grid_for(i64 items, int per_block, int max_blocks) {
    i64 g = (items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}


inline size_t dp_lds_for(int nm, int count_bytes) {
    return (size_t)(nm * (nm - 1) / 2) * (8 + 4 + 1) + (size_t)(nm * (nm - 1) * (nm - 2) / 6 + 4) * count_bytes + 16;
}
constexpr size_t kLdsPerWg = 160 * 1024;

void drop_graph(fseg_ctx *c) {
    for (int g = 0; g < 2; ++g) {
        if (c->graph_exec[g]) { (void)hipGraphExecDestroy(c->graph_exec[g]); c->graph_exec[g] = nullptr; }
        if (c->graph[g]) { (void)hipGraphDestroy(c->graph[g]); c->graph[g] = nullptr; }
    }
    c->n_graphs = 0;
}

// (re)bind the data-dependent arenas for the current capacities; the label arena is its own allocation because it is
// sized after everything else of a run has been written
// the label arena for label_cap labels: two bits each (d_packed) when this context's parameters allow it, else bytes (d_labels)
bool labels_packed(const fseg_ctx *c) { return c->label_packed_ok && c->have_params && !c->label_has2; }
int ensure_label_arena(fseg_ctx *c) {
    if (labels_packed(c)) return ensure(c, c->d_packed, (size_t)((c->label_cap + 15) / 16) * 4 + 16);
    return ensure(c, c->d_labels, (size_t)c->label_cap + 16);
}
int alloc_arenas(fseg_ctx *c) {
    drop_graph(c);
    Carve cv;
    cv.add(c->d_prob_iv, (size_t)c->prob_cap * 4);
    cv.add(c->d_prob_start, (size_t)c->prob_cap * 4);
    cv.add(c->d_prob_n, (size_t)c->prob_cap * 4);
    cv.add(c->d_prob_pair_off, (size_t)c->prob_cap * 8);
    cv.add(c->d_prob_tri_off, (size_t)c->prob_cap * 8);
    cv.add(c->d_prob_flags, (size_t)c->prob_cap * 4);
    cv.add(c->d_prob_chain, (size_t)c->prob_cap * 4);
    cv.add(c->d_dp_items, (size_t)c->prob_cap * 4);
    cv.add(c->d_solve_items, (size_t)c->prob_cap * 4);
    cv.add(c->d_wide_items, (size_t)c->prob_cap * 4);
    cv.add(c->d_wide_all, (size_t)c->prob_cap * 4);
    cv.add(c->d_solve_desc, (size_t)c->prob_cap * sizeof(ProbDesc));
    cv.add(c->d_prob_cov_off, (size_t)c->prob_cap * 8);
    cv.add(c->d_prob_lane_lo, (size_t)c->prob_cap * 4);
    cv.add(c->d_prob_lane_n, (size_t)c->prob_cap * 4);
    cv.add(c->d_prob_desc, (size_t)c->prob_cap * sizeof(ProbDesc));
    cv.add(c->d_work_pc, (size_t)c->work_cap * 8);
    cv.add(c->d_cls_items, (size_t)c->work_cap * 16);
    cv.add(c->d_work_active, (size_t)c->work_cap);
    cv.add(c->d_cov, (size_t)c->cov_cap * 4);
    cv.add(c->d_pair_thr, (size_t)c->pair_cap * 8);
    cv.add(c->d_amb, (size_t)c->pair_cap * 4);
    cv.add(c->d_out, (size_t)c->tri_cap * 4);
    {
        size_t bytes = 0;
        const int nms[3] = {kClsSmall, kClsMid, c->nm_big};
        for (int q = 0; q < 3; ++q) {
            const bool on = ((c->split_dp >> q) & 1) && c->counts_known && c->use_fuse && c->fuse_on && c->n_solve[q] > 0;
            c->dpx_cnt[q] = (c->wide_solve && c->n_wide[q] > 0) ? 2 : 1;
            c->dpx_stride[q] = (i64)dpx_slot_bytes(nms[q], c->dpx_cnt[q]);
            c->dpx_n[q] = on ? c->n_solve[q] : 0;
            c->dpx_base[q] = (i64)bytes;
            bytes += (size_t)c->dpx_n[q] * (size_t)c->dpx_stride[q];
        }
        c->dpx_nm = c->nm_big;
        cv.add(c->d_dpx, bytes);
    }
    TRY(reserve(c, c->slab_arena, cv.total));
    cv.bind(c->slab_arena);
    TRY(ensure_label_arena(c));
    return FSEG_OK;
}

// look-back state of the three compactions (values, candidates, final positions): nb words each, zeroed per run
inline i64 scan_blocks(i64 n) { return (n + kScanBlock - 1) / kScanBlock; }

// the wave kernels need the exon stream's pieces to fit their LDS stage (k_wave): a batch with a rep of more exons keeps
// k_tiny / k_solve for its small problems
bool wave_on(const fseg_ctx *c) { return c->use_wave && c->max_rep_exons <= kWaveRepExons; }
ProbSplit split_of(const fseg_ctx *c, bool tiny, bool fuse) {
    const bool wave = wave_on(c);
    (void)wave;
    return ProbSplit{tiny ? kTiny : 0, (c->use_fuse && fuse) ? c->fuse_lanes : -1};
}

// Enqueue the segments `segs` of one run on the context's stream.
//   sized: the run is being launched piecewise with the host reading the sizes in between (first run of a batch):
//          the problem scan always runs as its own kernels (their totals are what the host waits for) and the label
//          arena is filled by its own kernel over exactly label_fill_bytes.
int enqueue_run(fseg_ctx *c, unsigned segs, bool sized = false, i64 label_fill_bytes = 0) {
    hipStream_t s = c->stream;
    const bool do_pre1 = segs & SEG_PRE1, do_pre2 = segs & SEG_PRE2, do_score = segs & SEG_SCORE, do_post1 = segs & SEG_POST1,
               do_post2 = segs & SEG_POST2;
    const bool stage_events = c->profiling && (sized || c->run_plain);
    const int n_part = c->n_part;
    const i64 K = c->K, NPOS = c->NPOS;
    Status *st = c->d_status.as<Status>();
    // the flag masks: Y > 0 | candidate | final position, flag_words(NPOS) words each
    unsigned *flag_pos_bits = c->d_bits.as<unsigned>(), *flag_cand_bits = flag_pos_bits + flag_words(NPOS), *flag_final_bits = flag_cand_bits + flag_words(NPOS);
    auto begin = [&](int i) { if (stage_events && (c->profile_all || i == ST_SCORE)) (void)hipEventRecord(c->ev_b[i], s); };
    auto end = [&](int i) { if (stage_events && (c->profile_all || i == ST_SCORE)) (void)hipEventRecord(c->ev_e[i], s); };
    // fork(k): side stream k continues from here; join(k): the main stream waits for it.  Every fork is joined before
    // the function returns, so a capture of the main stream ends with all branches merged.
    // Small batches (one partition, few problems) are chains of launch-latency-sized kernels: branches only add
    // cross-stream dependencies there (measured: config2 +4 %), so they stay on the one stream.
    const bool forking = would_fork(c) && !c->run_linear;
    const int tiny_max = c->tiny_on ? kTiny : 0;
    const bool wave = wave_on(c);
    const ProbSplit split = split_of(c, c->tiny_on, c->fuse_on);
    // list sizes known (the batch has been sized or has run): launches over an empty list are left out
    const bool known = c->counts_known;
    const bool any_arena = !known || c->n_arena_prob > 0;
    int fj_next = 0;
    hipError_t fj_err = hipSuccess;
    auto fj_event = [&]() { hipEvent_t e = c->fj[fj_next % fseg_ctx::kForkEvents]; ++fj_next; return e; };
    auto fork = [&](int k) -> hipStream_t {
        if (!forking) return s;
        hipEvent_t e = fj_event();
        hipError_t r = hipEventRecord(e, s);
        if (r == hipSuccess) r = hipStreamWaitEvent(c->side[k], e, 0);
        if (r != hipSuccess) fj_err = r;
        return c->side[k];
    };
    auto join = [&](int k) {
        if (!forking) return;
        hipEvent_t e = fj_event();
        hipError_t r = hipEventRecord(e, c->side[k]);
        if (r == hipSuccess) r = hipStreamWaitEvent(s, e, 0);
        if (r != hipSuccess) fj_err = r;
    };
    // The scoring plan of this enqueue (FSEG_SCORE_PLAN, see the scoring stage below) is decided HERE: with the device-side
    // fork (k_wait_word) its side streams are forked in front of the stage's predecessors, so that the events' latency
    // (10-15 us each) is over when k_prob_emit ends.
    const bool any_solve_plan = c->use_fuse && c->fuse_on;
    const char *plan = (do_score && c->prob_cap > 0 && known && !any_arena && !c->small_batch && forking && c->n_solve[2] > 0 && any_solve_plan && wave && c->score_plan[0]) ? c->score_plan : nullptr;
    struct PlanSegs { int n_seg = 1; bool used[4] = {true, false, false, false}; int side_of[4] = {-1, -1, -1, -1}; i64 n_wide_all = 0; bool wide_one = false; } ps;
    if (plan) {
        // W: the problems of every class that see more than kFuseLanes reads -- a handful per batch, most of which keep fewer
        // and are only looked at -- in ONE launch of the large class's 16-bit instance, a workgroup each (as a launch per
        // class on the tiny class's stream they held it back 46-60 us on config3 / config5: tools/run_gaps.py); batches
        // with many such problems keep the per-class instances (in a row on W's stream)
        ps.n_wide_all = c->n_wide[0] + c->n_wide[1] + c->n_wide[2];
        ps.wide_one = ps.n_wide_all <= fseg_ctx::kWideOneMax && strchr(plan, 'W') != nullptr;
        for (const char *p = plan; *p; ++p) {
            if (*p == '|') { ++ps.n_seg; continue; }
            if (ps.n_seg > 4) continue;
            if (*p == 'W') { if (c->wide_solve && ps.n_wide_all > 0) ps.used[ps.n_seg - 1] = true; }
            else ps.used[ps.n_seg - 1] = true;
        }
        if (ps.n_seg > 4) ps.n_seg = 4;
        // (the segments that have something to launch take the side streams in order: the first ones start first)
        for (int k = 1, nx = 0; k < ps.n_seg; ++k) if (ps.used[k]) ps.side_of[k] = nx++;
    }
    // device-side fork / join (k_wait_word): plain launches only (never inside a capture), and only when this enqueue holds
    // k_prob_emit, whose last workgroup is what the side streams wait for; only on side streams whose hardware queue is not the
    // main stream's (side_ok, probed once: a waiter there would sit in front of the kernel it waits for)
    const bool dev_sync = plan && c->dev_sync && !c->run_events_only && do_pre2 && (sized || c->run_plain) && c->d_sync.p != nullptr;
    SyncWords *sw = c->d_sync.as<SyncWords>();
    const unsigned sync_gen = dev_sync ? ++c->sync_gen : 0;
    // (a context's first stages with waiters also load the stage's code objects and make the runtime allocate scratch for the side
    // streams' queues -- hundreds of microseconds, once: tools/waiter_probe.py)
    const i64 sync_scale = std::max<i64>(1, c->LANES >> 18);
    const unsigned kSyncTicks = (unsigned)std::min<i64>((i64)c->sync_ticks * sync_scale, 2000000);
    auto dev_side = [&](int k) { return dev_sync && k >= 1 && k < ps.n_seg && ps.used[k] && ps.side_of[k] >= 0 && ps.side_of[k] < fseg_ctx::kSide && c->side_ok[ps.side_of[k]]; };
    // The fork is EARLY (the event in front of k_fix: its latency hides behind the stage's predecessors), the waiters' launches come
    // LATE on the host -- behind k_prob_emit's --: a waiter then never spins on the device while the host has not yet enqueued the
    // kernel it waits for (a host thread that is descheduled, or held up in the runtime beside another context's large pageable copy,
    // between the two was one way to a time-out: tests/test_gpu_contexts_stress.py); on the device nothing moves -- the host is four
    // launches (~20 us) ahead of the event either way.
    auto early_fork = [&]() {
        if (!dev_sync) return;
        for (int k = 1; k < ps.n_seg; ++k) if (dev_side(k)) (void)fork(ps.side_of[k]);
    };
    auto launch_fork_waiters = [&]() {
        if (!dev_sync) return;
        for (int k = 1; k < ps.n_seg; ++k) if (dev_side(k))
            hipLaunchKernelGGL(k_wait_word, dim3(1), dim3(64), 0, c->side[ps.side_of[k]], st, &sw->emit_gen, 1u, sync_gen, kSyncTicks);
    };
    const i64 avg_len = NPOS / (K > 0 ? K : 1);
    const int iv_threads = avg_len > 65536 ? 1024 : (avg_len > 16384 ? 256 : 64);
    int tile_grid = grid_for(c->n_tiles, 1, 16384);
    const i64 scan_nb = scan_blocks(NPOS);
    const int scan_grid = scan_nb > 0 ? (int)scan_nb : 1;       // exactly one workgroup per scan block (look-back)
    u64 *scan_state = c->d_scan_state.as<u64>();
    // single-pass look-back scan while the chain of blocks is short; block sums + one scanning workgroup beyond that
    const bool scan_single = scan_nb <= c->scan_single_max;
    int *bsum = scan_single ? nullptr : c->d_bsum.as<int>();
    int *bsum_side = scan_single ? nullptr : c->d_bsum_side.as<int>();     // block sums of the scan that runs on a side stream
    auto scan_counts = [&](hipStream_t q, int *bs, const unsigned *flags, u64 *total_dev, i64 *off_last) {
        if (scan_single) return;
        hipLaunchKernelGGL(k_scan1, dim3(grid_for(scan_nb, 1, 4096)), dim3(256), 0, q, flags, NPOS, bs);
        hipLaunchKernelGGL(k_scan2, dim3(1), dim3(kScan2Threads), 0, q, bs, scan_nb, total_dev, off_last);
    };
    int work_grid = grid_for(c->work_cap, 1, 4096);
    ProblemArrays pr{c->d_prob_iv.as<int>(), c->d_prob_start.as<int>(), c->d_prob_n.as<int>(),
                     c->d_prob_pair_off.as<i64>(), c->d_prob_tri_off.as<i64>(), c->d_prob_flags.as<int>(),
                     c->d_prob_chain.as<int>(), c->d_prob_cov_off.as<i64>(), c->d_prob_lane_lo.as<int>(),
                     c->d_prob_lane_n.as<int>()};
    const int pg = grid_for(NPOS / 8 / kProbBlock + 1, 1, 1024);
    // few candidates: the emit kernel scans by itself, one launch instead of three (never in a sized run: there the
    // host reads the scan's totals before the emit kernel is launched)
    i64 *prob_bs = (c->prob_self_scan && !sized) ? nullptr : c->d_prob_bs.as<i64>();
    if (do_pre1) {
    {   // Status and the flag planes (OR-ed into: k_smooth, k_peaks_edges, k_segments, k_refine) by one launch (two memsets were three fill kernels)
        const i64 nw = 3 * (i64)flag_words(NPOS);
        hipLaunchKernelGGL(k_clear, dim3(grid_for(nw / 4 + 1, 256, 2048)), dim3(256), 0, s, st, c->d_bits.as<unsigned>(), nw);
    }
    begin(ST_HIST);
    // S1
    hipLaunchKernelGGL(k_hist, dim3(grid_for(c->n_hist_chunks, 1, 16384)), dim3(512), 0, s, c->n_hist_chunks,
                       c->d_hc_part.as<int>(), c->d_hc_p0.as<i64>(), c->d_hc_n.as<int>(), c->d_hc_glo.as<int>(),
                       c->d_hc_ghi.as<int>(), c->d_hc_llo.as<i64>(), c->d_hc_lhi.as<i64>(), c->d_part_iv_off.as<i64>(), c->d_iv_start.as<int>(), c->d_iv_end.as<int>(),
                       c->d_pos_off.as<i64>(), c->d_part_lane_off.as<i64>(), c->d_lane_lx.as<int2>(),
                       c->d_lane_start.as<int>(), c->d_lane_pmax.as<int>(),
                       c->d_lex.as<int2>(), c->P.ignore_ends, c->d_y_raw.as<int>(), st,
                       scan_state, scan_single ? scan_nb * 3 : 0);
    end(ST_HIST); begin(ST_SMOOTH);
    // S2
#define FSEG_LAUNCH_SMOOTH(RV)                                                                                         \
    hipLaunchKernelGGL(k_smooth<RV>, dim3(tile_grid), dim3(kSmoothThreads), 0, s, c->n_tiles, c->d_tile_desc.as<TileDesc>(),      \
                       c->d_y_raw.as<int>(), c->d_w_main.as<double>(),                                                 \
                       c->P.radius_main, c->d_y.as<double>(), flag_pos_bits,                           \
                       flag_cand_bits, c->d_blk_pre.as<int>(), c->d_tile_tot.as<int>(), c->d_tile_defer.as<int>())
    // sigma = 5 (default) and sigma = 3 (config 5) have their own unrolled instances; any other radius runs the loop
    if (c->P.radius_main == 20) { FSEG_LAUNCH_SMOOTH(20); }
    else if (c->P.radius_main == 12) { FSEG_LAUNCH_SMOOTH(12); }
    else { FSEG_LAUNCH_SMOOTH(0); }
#undef FSEG_LAUNCH_SMOOTH
    end(ST_SMOOTH);
    // S3a threshold: needs only the smoothed signal, like the candidates (S3b) -- the two chains run side by side
    {
    hipStream_t q = fork(0);
    if (stage_events && c->profile_all) (void)hipEventRecord(c->ev_b[ST_THRESHOLD], q);
    // batches of many partitions of moderate size: a workgroup per partition does the whole threshold (k_thr_part); a batch of a
    // few large partitions (config 2: one) keeps the batch-wide compaction and a workgroup per 8192-value chunk
    // (Until the end of round 5 a context that shares the device kept the chunk kernels: host memory -> host memory, eight contexts, the job
    // ran at 368 M reads/s with k_thr_part against 381 M without -- a difference inside that figure's box noise, as it turned out.
    // Over resident batches, three pairs of runs in one call: 527.6 / 530.8 / 531.5 with it, 524.1 / 524.4 / 503.7 without: one launch
    // of 47 us instead of seven that hold a hardware queue for 84.)
    const bool thr_part_fits = c->max_part_pos <= (i64)kThrPartMaxChunks * 8192;
    const bool thr_part = thr_part_fits && (c->thr_part == 1 || (c->thr_part < 0 && n_part >= 64));
    if (thr_part) {
        hipLaunchKernelGGL(k_thr_part, dim3(grid_for(n_part, 1, 4096)), dim3(512), 0, q, n_part, c->d_part_iv_off.as<i64>(), c->d_pos_off.as<i64>(), NPOS,
                           flag_pos_bits, c->d_y.as<double>(), c->d_v.as<double>(), c->P.variance_factor, c->d_mean.as<double>(), c->d_thr.as<double>());
    } else {
    scan_counts(q, bsum_side, flag_pos_bits, &st->n_vals, nullptr);
    hipLaunchKernelGGL(k_scan_emit<kEmitValues>, dim3(scan_grid), dim3(256), 0, q, flag_pos_bits, NPOS,
                       bsum_side, scan_state, &st->n_vals, (i64 *)nullptr, &st->err, c->d_y.as<double>(), c->d_v.as<double>(), K, c->d_pos_off.as<i64>(),
                       c->d_iv_start.as<int>(), c->d_blk_iv0.as<int>(), (int *)nullptr, (int *)nullptr, (i64 *)nullptr,
                       c->force_scan_stall ? 1 : 0, (int *)nullptr);
    hipLaunchKernelGGL(k_voff, dim3(grid_for(n_part + 1, 1, 2048)), dim3(64), 0, q, n_part, c->d_part_iv_off.as<i64>(),
                       c->d_pos_off.as<i64>(), NPOS, flag_pos_bits, bsum_side, scan_state, &st->n_vals,
                       c->d_voff.as<i64>());
    hipLaunchKernelGGL(k_vplan, dim3(1), dim3(256), 0, q, n_part, c->d_part_iv_off.as<i64>(), c->d_pos_off.as<i64>(), NPOS,
                       c->d_voff.as<i64>(), c->d_chunk_off.as<i64>(), st, c->chunk_cap);
    int chunk_grid = grid_for(c->chunk_cap, 1, 4096);
    double *csum0 = c->d_csum.as<double>(), *csum1 = csum0 + c->chunk_cap;
    for (int pass = 0; pass < 2; ++pass)
        hipLaunchKernelGGL(k_vsum_chunks, dim3(chunk_grid), dim3(512), 0, q, n_part, c->d_voff.as<i64>(),
                           c->d_chunk_off.as<i64>(), c->d_v.as<double>(), csum0, pass, pass ? csum1 : csum0, c->chunk_cap);
    hipLaunchKernelGGL(k_vsum_part, dim3(grid_for(n_part, 64, 1024)), dim3(64), 0, q, n_part, c->d_voff.as<i64>(),
                       c->d_chunk_off.as<i64>(), csum0, csum1, c->P.variance_factor, c->d_mean.as<double>(),
                       c->d_thr.as<double>(), c->chunk_cap);
    }
    if (stage_events && c->profile_all) (void)hipEventRecord(c->ev_e[ST_THRESHOLD], q);
    }
    begin(ST_CANDIDATES);
    // S3b candidates
    hipLaunchKernelGGL(k_peaks_edges, dim3(grid_for(c->n_tiles, 256, 4096)), dim3(256), 0, s, c->n_tiles, c->d_tile_desc.as<TileDesc>(),
                       c->d_tile_defer.as<int>(), c->d_y.as<double>(), flag_cand_bits, c->d_part_has2.as<int>(), n_part);
    scan_counts(s, bsum, flag_cand_bits, &st->n_cand, c->d_cand_off.as<i64>() + K);
    hipLaunchKernelGGL(k_scan_emit<kEmitPositions>, dim3(scan_grid), dim3(256), 0, s, flag_cand_bits, NPOS,
                       bsum, scan_state + scan_nb, &st->n_cand, c->d_cand_off.as<i64>() + K, &st->err, (const double *)nullptr, (double *)nullptr, K, c->d_pos_off.as<i64>(),
                       c->d_iv_start.as<int>(), c->d_blk_iv0.as<int>(), c->d_cand_y.as<int>(), (int *)nullptr,
                       c->d_cand_off.as<i64>(), 0, (int *)nullptr);
    end(ST_CANDIDATES);
    join(0);
    early_fork();
    begin(ST_FIX);
    // S4
    hipLaunchKernelGGL(k_fix, dim3(grid_for(K, 1, 8192)), dim3(iv_threads), 0, s, K, c->d_pos_off.as<i64>(),
                       c->d_iv_part.as<int>(), c->d_cand_off.as<i64>(), c->d_cand_y.as<int>(), c->d_y.as<double>(),
                       c->d_thr.as<double>(), c->P.max_problem_size, c->d_fixed0.as<unsigned char>(),
                       c->d_added.as<unsigned char>(), c->d_fixed.as<unsigned char>(), c->d_chosen.as<unsigned char>(),
                       c->d_cand_pn.as<int>(), c->d_seg_iv.as<int>(), st);
    // (with the block sums of the problem scan when the scan is a launch of its own: k_prob_scan1 is the rescan's only)
    i64 *range_bs = prob_bs;
    hipLaunchKernelGGL(k_prob_range, dim3(pg), dim3(kRangeThreads), 0, s, st, c->d_cand_pn.as<int>(),
                       c->d_seg_iv.as<int>(), c->d_cand_y.as<int>(), c->d_iv_part.as<int>(), c->d_iv_start.as<int>(),
                       c->d_part_lane_off.as<i64>(), c->d_lane_start.as<int>(), c->d_lane_pmax.as<int>(),
                       c->d_cand_ll.as<int>(), c->d_cand_ln.as<int>(), c->d_cand_wide.as<unsigned char>(), c->d_lane_lx.as<int2>(),
                       c->d_lex.as<int2>(), c->wide_by_seen ? 1 : 0, split.fuse_lanes, range_bs, split);
    if (prob_bs) {
        if (!range_bs)
            hipLaunchKernelGGL(k_prob_scan1, dim3(pg), dim3(256), 0, s, st, c->d_cand_pn.as<int>(), c->d_cand_ln.as<int>(), c->d_cand_wide.as<unsigned char>(), prob_bs, split);
        hipLaunchKernelGGL(k_prob_scan2, dim3(1), dim3(256), 0, s, st, prob_bs);
    }
    end(ST_FIX);
    }   // do_pre1
    bool score_begun = false;
    if (do_pre2) {
    if (!do_pre1) early_fork();
    begin(ST_SCORE_PREP);
    hipLaunchKernelGGL(k_prob_emit, dim3(pg), dim3(256), 0, s, st, c->d_cand_pn.as<int>(), c->d_cand_ll.as<int>(),
                       c->d_cand_ln.as<int>(), c->d_seg_iv.as<int>(), c->d_cand_off.as<i64>(), prob_bs,
                       pr, c->prob_cap, c->d_work_pc.as<int2>(), c->d_cls_items.as<int4>(),
                       c->work_cap, c->d_dp_items.as<int>(), c->d_prob_desc.as<ProbDesc>(), c->d_iv_start.as<int>(),
                       c->d_iv_part.as<int>(), c->d_part_lane_off.as<i64>(), split, c->d_solve_items.as<int>(), c->d_solve_desc.as<ProbDesc>(),
                       c->d_wide_items.as<int>(), c->d_wide_all.as<int>(), c->d_cand_wide.as<unsigned char>(),
                       dev_sync ? sw : (SyncWords *)nullptr, sync_gen);
    launch_fork_waiters();
    // S5.  The arena path's window coverage (and pair thresholds) are launches of their own in front of k_score: they are
    // interval scoring (get_cumulative_coverage :188-246 -- the solve-list kernels do the same inside their workgroups), so
    // where the stages are bracketed by events the scoring stage's bracket opens here
    if (c->prob_cap > 0 && any_arena) {
        if (stage_events && do_score) { end(ST_SCORE_PREP); begin(ST_SCORE); score_begun = true; }
        const int cov_blocks = work_grid < 2048 ? work_grid : 2048;
        const int pt_blocks = c->small_batch ? grid_for(c->prob_cap, 1, 512) : 0;       // fused only for small batches
        if (!pt_blocks)
            hipLaunchKernelGGL(k_pair_thresholds, dim3(grid_for(c->prob_cap, 1, 2048)), dim3(256), 0, s, st, pr,
                               c->d_prob_desc.as<ProbDesc>(), c->prob_cap, c->d_cand_off.as<i64>(), c->d_cand_y.as<int>(),
                               c->d_h_table.as<double>(), c->P.h_len,
                               c->P.threshold_rate, c->d_pair_thr.as<int2>(), c->pair_cap, c->d_amb.as<unsigned>(),
                               c->d_out.as<unsigned>(), c->tri_cap);
        hipLaunchKernelGGL(k_cov, dim3(cov_blocks + pt_blocks), dim3(kLaneChunk), 0, s, st,
                           c->d_prob_desc.as<ProbDesc>(), c->prob_cap, c->d_work_pc.as<int2>(), c->work_cap, c->d_cand_off.as<i64>(),
                           c->d_cand_y.as<int>(), c->d_iv_start.as<int>(), c->d_lane_lx.as<int2>(),
                           c->d_lex.as<int2>(),
                           c->d_cov.as<unsigned>(), c->cov_cap, c->d_work_active.as<unsigned char>(),
                           cov_blocks, pr, c->d_h_table.as<double>(), c->P.h_len, c->P.threshold_rate, c->d_pair_thr.as<int2>(),
                           c->pair_cap, c->d_amb.as<unsigned>(), c->d_out.as<unsigned>(), c->tri_cap);
    }
    if (!score_begun) end(ST_SCORE_PREP);
    }   // do_pre2
    if (do_score && !score_begun) begin(ST_SCORE);
    if (do_score && c->prob_cap > 0) {
        // How the scoring kernels share the chip is a plan (FSEG_SCORE_PLAN, default "gM|W|hB|gST"): streams separated by '|'
        // (the first is the main stream; the segments that have something to launch take the side streams in order); B M S T =
        // the large / mid / small / tiny class -- a class on the split path is k_solve (rounds) followed by k_dpw (its DPs) --,
        // b m s = the classes' 16-bit-counter instances (launched over their classes' wide problems), g = k_gate (wait until the
        // large class's workgroups, both instances', are placed), h = wait until the 16-bit instance's are, e = wait for the large
        // class to end.  Measured on config4 (250 k-read batch, stage alone, tools/r4_plans.sh):
        //   * everything on the main stream 0.19-0.21 ms; a stream each without a gate 0.21 (the dispatcher runs them in the
        //     reverse of their launch order: a large-class workgroup needs eight wave slots and half a CU's LDS at once);
        //   * round 3's "BM|gTS" 0.161-0.165: the tiny class in the large class's shadow, then small beside mid;
        //   * a cross-stream dependency costs ~9 us (fork or join), so the chain that ends LAST belongs on the main stream,
        //     where the stage's end needs no join: the mid class (gate, rounds, DPs) on main, the large class on a side stream
        //     of its own, tiny + small on a third: 0.144-0.149 with the split path ("B|gM|gTS", the same chains with the large
        //     class on main: 0.154-0.167; per-problem clocks, tools/r4_ticks.sh: 133 us from first start to last end either way);
        //   * the 16-bit instances first (batches of 1 000-read partitions): see the passes below.
        // Anything that does not name each class once, batches with arena-path problems and small batches: one stream.
        const bool sfork = any_arena;                               // (the arena path's work-item kernels keep their streams)
        hipStream_t qt = (tiny_max > 0 && sfork) ? fork(2) : s;     // (forked here: a side stream continues from where it was forked)
#ifdef FSEG_SCORE_TIMING
#define FSEG_TARG , c->d_tacc.as<unsigned long long>()
#else
#define FSEG_TARG
#endif
#ifndef FSEG_WG_SMALL
#define FSEG_WG_SMALL 4096
#endif
#ifndef FSEG_WG_MID
#define FSEG_WG_MID 2048
#endif
#ifndef FSEG_WG_TINY
#define FSEG_WG_TINY 4096
#endif
#define FSEG_LAUNCH_SCORE(Q, NMV, CLS, MAXWG)                                                                           \
        hipLaunchKernelGGL(k_score<NMV>, dim3(work_grid < (MAXWG) ? work_grid : (MAXWG)), dim3(ScoreCfg<NMV>::kThreads),  \
                           score_lds_for((NMV) == kNMax ? c->nm_big : (NMV), (NMV) + 1), Q, st, CLS,                                 \
                           ((NMV) == kNMax ? c->nm_big : (NMV)), pr, c->prob_cap, c->d_cls_items.as<int4>(),              \
                           c->d_prob_desc.as<ProbDesc>(), c->work_cap, c->d_cand_off.as<i64>(),                         \
                           c->d_cand_y.as<int>(), c->d_work_active.as<unsigned char>(), c->d_cov.as<unsigned>(),        \
                           c->cov_cap, c->d_pair_thr.as<int2>(), c->pair_cap, c->d_out.as<unsigned>(), c->tri_cap,      \
                           c->d_amb.as<unsigned>() FSEG_TARG)
        // (a 16-bit-counter instance of a sized batch goes over its class's WIDE problems only: wide_n of them, through wide_items)
        auto wide_n = [&](int cls, size_t cnt_bytes) -> i64 { return (known && cls >= 0 && cls < 3 && cnt_bytes == 2) ? c->n_wide[cls] : -1; };
#define FSEG_SOLVE_N(CNT, CLS, N_ITEMS) (wide_n(CLS, sizeof(CNT)) >= 0 ? wide_n(CLS, sizeof(CNT)) : (i64)(N_ITEMS))
#define FSEG_SOLVE_WIDE(CNT, CLS) (wide_n(CLS, sizeof(CNT)) >= 0 ? c->d_wide_items.as<int>() : (const int *)nullptr)
#define FSEG_SOLVE_ARGS(NMV, CNT, CLS)                                                                                      \
                               st, CLS, ((NMV) == kNMax ? c->nm_big : (NMV)), list_lb(CLS),                                    \
                               (wide_n(CLS, sizeof(CNT)) >= 0 ? wide_n(CLS, sizeof(CNT)) : list_ln(CLS)), pr, c->d_solve_desc.as<ProbDesc>(), \
                               c->prob_cap, c->d_cand_y.as<int>(), c->d_lane_lx.as<int2>(), c->d_lex.as<int2>(),               \
                               c->d_h_table.as<double>(), c->P.h_len, c->P.threshold_rate,               \
                               c->d_thr_tab.as<int2>(), c->P.min_read_support_outside, c->d_chosen.as<unsigned char>()
        // the split path (k_solve<.., SPLIT> then k_dpw on the same stream) for a list that fits the hand-over arena as laid out.
        // (Until round 5 only where the context has the device to itself: host memory -> host memory, eight contexts, the two were
        // within the noise -- 383 against 388 M reads/s; over resident batches the split is 1.8 % faster; FSEG_SPLIT_ALWAYS=0.)
        auto split_ok = [&](int cls, int cnt_bytes) {
            // (k_dpw takes the problem its workgroup index names -- no grid stride --, so a list longer than the grid cap of the
            // two launches keeps the DP as k_solve's tail)
            return known && cls >= 0 && cls < 3 && ((c->split_dp >> cls) & 1) && c->dpx_n[cls] > 0 && c->n_solve[cls] <= c->dpx_n[cls] && c->dpx_nm == c->nm_big &&
                   c->n_solve[cls] <= kSplitGridCap && cnt_bytes <= c->dpx_cnt[cls] && c->d_dpx.p != nullptr;
        };
#define FSEG_LAUNCH_SOLVE(Q, NMV, CNT, VT, CLS, N_ITEMS, MAXWG)                                                              \
            hipLaunchKernelGGL((k_solve<NMV, CNT, VT, false>), dim3(grid_for(FSEG_SOLVE_N(CNT, CLS, N_ITEMS), 1, known ? (1 << 20) : (MAXWG))), dim3(SolveCfg<NMV>::kThreads), \
                               solve_lds_for((NMV) == kNMax ? c->nm_big : (NMV), (NMV) + 1, (int)sizeof(CNT)), Q,               \
                               FSEG_SOLVE_ARGS(NMV, CNT, CLS), (unsigned char *)nullptr, (i64)0, FSEG_SOLVE_WIDE(CNT, CLS) FSEG_TARG)
        // every list's problems that see more than kFuseLanes reads, through wide_all: the large class's 16-bit instance, a workgroup each
#define FSEG_LAUNCH_WIDE_ALL(Q, VT)                                                                                          \
            hipLaunchKernelGGL((k_solve<kNMax, unsigned short, VT, false>), dim3((unsigned)n_wide_all), dim3(SolveCfg<kNMax>::kThreads), \
                               solve_lds_for(c->nm_big, kNMax + 1, 2), Q, st, -1, c->nm_big, (i64)0, n_wide_all, pr, c->d_solve_desc.as<ProbDesc>(), \
                               c->prob_cap, c->d_cand_y.as<int>(), c->d_lane_lx.as<int2>(), c->d_lex.as<int2>(),               \
                               c->d_h_table.as<double>(), c->P.h_len, c->P.threshold_rate,               \
                               c->d_thr_tab.as<int2>(), c->P.min_read_support_outside, c->d_chosen.as<unsigned char>(),       \
                               (unsigned char *)nullptr, (i64)0, c->d_wide_all.as<int>() FSEG_TARG)
#define FSEG_LAUNCH_DPW(Q, NMV, CNT, VT, CLS, N_ITEMS, TT)                                                                   \
            hipLaunchKernelGGL((k_dpw<NMV, CNT, VT, TT>), dim3(grid_for(FSEG_SOLVE_N(CNT, CLS, N_ITEMS), 1, (int)kSplitGridCap)), dim3(TT),      \
                               dpw_lds_for(nm_rt, (int)sizeof(VT), (int)sizeof(CNT)), Q, st, nm_rt, list_lb(CLS),                \
                               FSEG_SOLVE_N(CNT, CLS, list_ln(CLS)), pr,                                                        \
                               c->d_solve_desc.as<ProbDesc>(), dpx0, dstride, \
                               c->P.min_read_support_outside, c->d_chosen.as<unsigned char>(), FSEG_SOLVE_WIDE(CNT, CLS) FSEG_TARG)
        // the split path, one instance: k_solve<.., SPLIT> (set-up and rounds) then k_dpw (the DPs) on the same stream
#define FSEG_LAUNCH_SPLIT(Q, NMV, CNT, VT, CLS, N_ITEMS)                                                                     \
        do { const int nm_rt = (NMV) == kNMax ? c->nm_big : (NMV);                                                            \
            unsigned char *dpx0 = c->d_dpx.as<unsigned char>() + c->dpx_base[(CLS) < 0 ? 0 : (CLS)];                                           \
            const i64 dstride = c->dpx_stride[(CLS) < 0 ? 0 : (CLS)];                                                                       \
            hipLaunchKernelGGL((k_solve<NMV, CNT, int, true>), dim3(grid_for(FSEG_SOLVE_N(CNT, CLS, N_ITEMS), 1, (int)kSplitGridCap)), dim3(SolveCfg<NMV>::kThreads), \
                               solve_lds_for(nm_rt, (NMV) + 1, (int)sizeof(CNT)), Q, FSEG_SOLVE_ARGS(NMV, CNT, CLS), dpx0, dstride, \
                               FSEG_SOLVE_WIDE(CNT, CLS) FSEG_TARG);                                    \
            if (g_ablate_dp) break;                                                                                               \
            /* the large class's DPs by workgroups of eight waves (k_dpw<.., 512>: 27 us alone instead of 38; FSEG_SPLIT_DP bit 3: by one) */ \
            if constexpr ((NMV) == kNMax) { if (!(c->split_dp & 8)) { FSEG_LAUNCH_DPW(Q, NMV, CNT, VT, CLS, N_ITEMS, 512); break; } }      \
            FSEG_LAUNCH_DPW(Q, NMV, CNT, VT, CLS, N_ITEMS, 64); } while (0)
        // 32-bit DP keys (dp_solve_push) when no sum of a chain can reach 2^24: at most 32 links times the reads of the largest partition
        // (WHICH: 1 = the instance with 8-bit counters, 2 = the one with 16-bit counters if the class has problems for it, 3 = both)
#define FSEG_LAUNCH_SPLIT_K(Q, NMV, CNT, CLS, N_ITEMS)                                                                        \
            do { if (key32) FSEG_LAUNCH_SPLIT(Q, NMV, CNT, int, CLS, N_ITEMS); else FSEG_LAUNCH_SPLIT(Q, NMV, CNT, i64, CLS, N_ITEMS); } while (0)
#define FSEG_LAUNCH_SOLVE_X(Q, NMV, CLS, N_ITEMS, MAXWG, WHICH)                                                              \
            do { if (split_ok(CLS, FSEG_WIDE_NEEDED(CLS) ? 2 : 1)) {                                                            \
                         if ((WHICH) & 1) FSEG_LAUNCH_SPLIT_K(Q, NMV, unsigned char, CLS, N_ITEMS);                              \
                         if (((WHICH) & 2) && FSEG_WIDE_NEEDED(CLS)) FSEG_LAUNCH_SPLIT_K(Q, NMV, unsigned short, CLS, N_ITEMS);   \
                 } else if (key32) { if ((WHICH) & 1) FSEG_LAUNCH_SOLVE(Q, NMV, unsigned char, int, CLS, N_ITEMS, MAXWG);              \
                              if (((WHICH) & 2) && FSEG_WIDE_NEEDED(CLS)) FSEG_LAUNCH_SOLVE(Q, NMV, unsigned short, int, CLS, N_ITEMS, MAXWG); } \
                 else { if ((WHICH) & 1) FSEG_LAUNCH_SOLVE(Q, NMV, unsigned char, i64, CLS, N_ITEMS, MAXWG);                    \
                        if (((WHICH) & 2) && FSEG_WIDE_NEEDED(CLS)) FSEG_LAUNCH_SOLVE(Q, NMV, unsigned short, i64, CLS, N_ITEMS, MAXWG); } } while (0)
#define FSEG_LAUNCH_SOLVE_W(Q, NMV, CLS, N_ITEMS, MAXWG) FSEG_LAUNCH_SOLVE_X(Q, NMV, CLS, N_ITEMS, MAXWG, 3)
#define FSEG_WIDE_NEEDED(CLS) (!known || (c->wide_solve && ((CLS) < 0 || c->n_wide[(CLS)] > 0)))
        // Two ways a problem is scored (prob_kind): the arena path's work items (k_score per size class) and the problems that
        // see few reads (at most 255: 8-bit counters), whole, one workgroup each (k_solve per size class).
        // A batch usually holds only one kind; a class's two launches share a stream.
        // one wave per problem on the exon stream (k_wave): the small class's solve list and k_tiny's list
#define FSEG_LAUNCH_WAVE_V(Q, NMV, VT, LIST, N_ITEMS)                                                                        \
            hipLaunchKernelGGL((k_wave<NMV, VT>), dim3(grid_for((N_ITEMS), 4, FSEG_WG_TINY)), dim3(256), 0, Q, st,           \
                               c->d_solve_desc.as<ProbDesc>(), c->prob_cap, LIST, pr, c->d_cand_y.as<int>(), c->d_lane_lx.as<int2>(), \
                               c->d_lex.as<int2>(), c->d_h_table.as<double>(), c->P.h_len, c->P.threshold_rate,               \
                               c->d_thr_tab.as<int2>(), c->P.min_read_support_outside, c->d_chosen.as<unsigned char>(),        \
                               list_lb(LIST), list_ln(LIST) FSEG_TARG)
#define FSEG_LAUNCH_WAVE(Q, NMV, LIST, N_ITEMS)                                                                              \
            do { if (key32) { FSEG_LAUNCH_WAVE_V(Q, NMV, int, LIST, N_ITEMS); }                              \
                 else { FSEG_LAUNCH_WAVE_V(Q, NMV, i64, LIST, N_ITEMS); } } while (0)
        // the bounds of solve list `l` (0..2 the classes, 3 the tiny problems, < 0 the three classes together) when the host knows them
        const bool key32 = c->max_part_lanes < kKey32Reads && !c->force_key64;     // (FSEG_FORCE_KEY64=1: the 64-bit instances whatever the batch)
        auto list_lb = [&](int l) -> i64 { return !known ? -1 : (l <= 0 ? 0 : (l == 1 ? c->n_solve[0] : (l == 2 ? c->n_solve[0] + c->n_solve[1] : c->n_solve[0] + c->n_solve[1] + c->n_solve[2]))); };
        auto list_ln = [&](int l) -> i64 { return !known ? -1 : (l < 0 ? c->n_solve[0] + c->n_solve[1] + c->n_solve[2] : (l == 3 ? c->n_tiny : c->n_solve[l])); };
        const bool any_solve = c->use_fuse && c->fuse_on && (!known || c->n_solve[0] + c->n_solve[1] + c->n_solve[2] > 0);
        const i64 cap = c->prob_cap;
        if (c->small_batch) {
            if (any_arena) FSEG_LAUNCH_SCORE(s, kNMax, -1, 512);    // few work items: one launch for every size class
            if (any_solve) FSEG_LAUNCH_SOLVE_W(s, kNMax, -1, known ? c->n_solve[0] + c->n_solve[1] + c->n_solve[2] : cap, 512);
        } else {                                     // the size classes own disjoint problems: three concurrent chains
            hipStream_t q1 = sfork ? fork(0) : s, q0 = sfork ? fork(1) : s;
            if (plan) {
                // (which segments have something to launch and the side streams they take: decided at the top, `ps`)
                const int n_seg = ps.n_seg;
                const bool *used = ps.used;              // a stream whose kernels have nothing to do is left alone
                const int *side_of = ps.side_of;
                const i64 n_wide_all = ps.n_wide_all;
                const bool wide_one = ps.wide_one;
                // every side stream continues from HERE (the stage's begin event, when it is being recorded, is the first side
                // stream's fork: one marker packet less in front of everything -- each is ~5 us on its queue)
                // (every side stream on that one event: 0.135 -> 0.137-0.149 ms -- the records stagger the streams' starts)
                bool shared = !(stage_events && forking);
                for (int k = 1; k < n_seg; ++k) if (used[k]) {
                    if (dev_side(k)) continue;           // forked early; its waiter (k_wait_word) is what the stream runs first
                    if (!shared) {
                        shared = true;
                        if (hipError_t r = hipStreamWaitEvent(c->side[side_of[k]], c->ev_b[ST_SCORE], 0); r != hipSuccess) fj_err = r;
                    } else (void)fork(side_of[k]);
                }
                int seg = 0;
                // (a plan with W: B M S launch their 8-bit instances only; without: both instances, one after the other)
                const int wb = strchr(plan, 'W') ? 1 : 3, wm = wb, ws = wb;
                // the large class's workgroups a start gate waits for: the 8-bit instance's and the 16-bit instance's (one per wide
                // problem) -- the latter need 90 KB of LDS each and find no room once the other classes are in
                const i64 wide_wgs = wide_one ? (c->wide_solve ? n_wide_all : 0) : (FSEG_WIDE_NEEDED(2) ? c->n_wide[2] : 0);
                const i64 big_wgs = c->n_solve[2] + wide_wgs;
                // Three passes over the plan: the large class's launches that open their stream go out FIRST -- the 16-bit instance
                // (W), then the 8-bit one (B, behind `h` = a gate on W's workgroups having started) --, everything else follows in
                // plan order (a stream's own order is kept).  Why: a workgroup of the 16-bit instance holds up to 120 KB of LDS (n
                // <= 60: planes 28 + coverage 16 + 16-bit counters 68 KB) and fits no CU that has one of the 8-bit instance's
                // (81 KB); enqueued behind it, config3's one real wide problem was placed 135-150 us into the stage, when the 8-bit
                // instance and the classes behind the gate had drained, and the stage took 0.29 ms (tools/stage_timeline.py).  A
                // class has a handful of wide problems: placed first they take a few CUs and the 8-bit instance the rest.
                for (int pass = 0; pass < 3; ++pass) {                       // 0: W   1: h, B   2: the rest
                seg = 0;
                bool opens = true;
                for (const char *p = plan; *p && seg < n_seg; ++p) {
                    if (*p == '|') { ++seg; opens = true; continue; }
                    const int when = !opens ? 2 : (*p == 'W' ? 0 : ((*p == 'B' || *p == 'h') ? 1 : 2));
                    if (*p != 'h') opens = false;
                    if (!used[seg] || when != pass) continue;
                    hipStream_t q = seg == 0 ? s : c->side[side_of[seg]];
                    // (the side streams' waiters are released by k_prob_emit's last workgroup: emit_done)
#ifdef FSEG_ABLATE_STAGE
                    // diagnostic build (wrong results, honest timing): FSEG_ABLATE = bit mask of what the stage leaves out --
                    // 1 the DP launches (k_dpw), 2 the large class, 4 the gates, 8 the mid class, 16 the small class, 32 the tiny class
                    {
                        static const int abl = getenv("FSEG_ABLATE") ? atoi(getenv("FSEG_ABLATE")) : 0;
                        g_ablate_dp = (abl & 1) != 0;
                        if (((abl & 2) && (*p == 'B' || *p == 'W')) || ((abl & 4) && (*p == 'g' || *p == 'h')) || ((abl & 8) && *p == 'M') ||
                            ((abl & 16) && *p == 'S') || ((abl & 32) && *p == 'T')) continue;
                    }
#endif
                    switch (*p) {
                    case 'B': FSEG_LAUNCH_SOLVE_X(q, kNMax, 2, c->n_solve[2], 512, wb); break;
                    case 'M': if (c->n_solve[1] > 0) FSEG_LAUNCH_SOLVE_X(q, kClsMid, 1, c->n_solve[1], FSEG_WG_MID, wm); break;
                    case 'S': if (c->n_solve[0] > 0) FSEG_LAUNCH_SOLVE_X(q, kClsSmall, 0, c->n_solve[0], FSEG_WG_SMALL, ws); break;
                    case 'W':
                        if (!c->wide_solve || n_wide_all == 0) break;
                        if (wide_one) {                          // (the DP stays the workgroup's tail: nothing to hand over, no second placement)
                            if (key32) FSEG_LAUNCH_WIDE_ALL(q, int); else FSEG_LAUNCH_WIDE_ALL(q, i64);
                        } else {
                            FSEG_LAUNCH_SOLVE_X(q, kNMax, 2, c->n_solve[2], 512, 2);
                            if (c->n_solve[1] > 0) FSEG_LAUNCH_SOLVE_X(q, kClsMid, 1, c->n_solve[1], FSEG_WG_MID, 2);
                            if (c->n_solve[0] > 0) FSEG_LAUNCH_SOLVE_X(q, kClsSmall, 0, c->n_solve[0], FSEG_WG_SMALL, 2);
                        }
                        break;
                    case 'T': if (c->n_tiny > 0) FSEG_LAUNCH_WAVE(q, kTiny, 3, c->n_tiny); break;
                    case 'g': hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, q, st, 0, (unsigned)(big_wgs < 512 ? big_wgs : 512), 3000u); break;
                    case 'h': if (wide_wgs > 0)       // the 16-bit instance's workgroups (up to 120 KB of LDS each) take their CUs first
                                  hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, q, st, 2, (unsigned)(wide_wgs < 256 ? wide_wgs : 256), 1500u);
                              break;
                    default: break;
                    }
                }
                }
                // join: a side chain with a device-side fork ends with k_signal and the main stream waits for the words (one wave in
                // front of k_segments) instead of two marker + barrier packets per side stream
                unsigned dev_mask = 0;
                for (int k = 1; k < n_seg; ++k) if (used[k]) {
                    if (dev_side(k)) { hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, c->side[side_of[k]], &sw->side_gen[side_of[k]], sync_gen); dev_mask |= 1u << side_of[k]; }
                    else join(side_of[k]);
                }
                if (dev_mask) hipLaunchKernelGGL(k_wait_word, dim3(1), dim3(64), 0, s, st, sw->side_gen, dev_mask, sync_gen, kSyncTicks);
            } else {
            if (any_arena && (!known || c->n_cls_work[2] > 0)) FSEG_LAUNCH_SCORE(s, kNMax, 2, 512);     // big problems: they are the long poles
            if (any_solve && (!known || c->n_solve[2] > 0)) FSEG_LAUNCH_SOLVE_W(s, kNMax, 2, known ? c->n_solve[2] : cap, 512);
            if (any_arena && (!known || c->n_cls_work[1] > 0)) FSEG_LAUNCH_SCORE(q1, kClsMid, 1, 1280);
            if (any_solve && (!known || c->n_solve[1] > 0)) FSEG_LAUNCH_SOLVE_W(q1, kClsMid, 1, known ? c->n_solve[1] : cap, FSEG_WG_MID);
            if (any_arena && (!known || c->n_cls_work[0] > 0)) FSEG_LAUNCH_SCORE(q0, kClsSmall, 0, 2048);
            if (any_solve && (!known || c->n_solve[0] > 0)) FSEG_LAUNCH_SOLVE_W(q0, kClsSmall, 0, known ? c->n_solve[0] : cap, FSEG_WG_SMALL);
            }
        }
#undef FSEG_LAUNCH_SOLVE_W
#undef FSEG_LAUNCH_SOLVE_X
#undef FSEG_LAUNCH_SPLIT_K
#undef FSEG_LAUNCH_DPW
#undef FSEG_LAUNCH_WIDE_ALL
#undef FSEG_LAUNCH_SPLIT
#undef FSEG_LAUNCH_SOLVE
#undef FSEG_SOLVE_ARGS
#undef FSEG_SOLVE_WIDE
#undef FSEG_SOLVE_N
#undef FSEG_LAUNCH_SCORE
        if (tiny_max > 0 && !plan) {
            // the problems with a handful of candidates, whole (coverage, labels, counts, DP), beside the others (launched
            // after the classes whose workgroups need half a CU's LDS each); joined at the end of this stage, so the
            // stage's time bracket covers all scoring work
            if (wave) FSEG_LAUNCH_WAVE(qt, kTiny, 3, known ? c->n_tiny : c->prob_cap);
            else
            hipLaunchKernelGGL(k_tiny, dim3(grid_for(known ? c->n_tiny : c->prob_cap, 4, FSEG_WG_TINY)), dim3(256), 0, qt, st, c->d_solve_desc.as<ProbDesc>(),
                               c->prob_cap, tiny_max, pr, c->d_cand_y.as<int>(), c->d_lane_ex.as<longlong2>(),
                               c->d_ex_ts.as<int>(), c->d_ex_te.as<int>(), c->d_h_table.as<double>(), c->P.h_len, c->P.threshold_rate,
                               c->d_thr_tab.as<int2>(), c->P.min_read_support_outside, c->d_chosen.as<unsigned char>(), list_lb(3), list_ln(3) FSEG_TARG);
        }
#undef FSEG_LAUNCH_WAVE
#undef FSEG_LAUNCH_WAVE_V
        if (!c->small_batch && sfork) { join(0); join(1); }
        if (c->have_huge && any_arena)
            hipLaunchKernelGGL(k_score_huge, dim3(256), dim3(512), kHugeScoreLds, s, st, c->d_dp_items.as<int>(), pr,
                               c->d_prob_desc.as<ProbDesc>(), c->prob_cap, c->work_cap, c->d_cand_y.as<int>(),
                               c->d_cov.as<unsigned>(), c->cov_cap, c->d_pair_thr.as<int2>(), c->pair_cap,
                               c->d_out.as<unsigned>(), c->tri_cap, c->d_amb.as<unsigned>());
        if (c->have_huge && c->nm_giant > 0 && any_arena && c->d_giant.p)
            hipLaunchKernelGGL(k_score_giant, dim3(kGiantWgs), dim3(512), giant_score_lds(c->nm_giant), s, st, c->d_dp_items.as<int>(), pr,
                               c->d_prob_desc.as<ProbDesc>(), c->prob_cap, c->work_cap, c->d_cand_y.as<int>(),
                               c->d_cov.as<unsigned>(), c->cov_cap, c->d_pair_thr.as<int2>(), c->pair_cap,
                               c->d_out.as<unsigned>(), c->tri_cap, c->d_amb.as<unsigned>(), c->nm_giant,
                               c->d_giant.as<unsigned char>(), (i64)giant_scratch_bytes(c->nm_giant));
        if (tiny_max > 0 && sfork) join(2);    // k_tiny is interval scoring too: inside the stage's time bracket
    }
    if (do_score) end(ST_SCORE);
    const i64 labels_n16 = (c->label_cap + 15) / 16;            // the arena is allocated in multiples of 16 bytes
    // the label arena's '0' fill rides the big-problem DP launch as extra workgroups -- unless the run is being sized
    // (the arena's size is not known yet) or there is no DP launch
    // (not where the stages are bracketed by events: the fill is the labels stage's work, and a DP bracket that holds it says nothing about the DP)
    const bool lab_packed = labels_packed(c);                   // (two-bit labels: the arena is cleared by a memset, nothing rides)
    const bool ride_fill = !sized && c->prob_cap > 0 && c->label_cap > 0 && any_arena && !stage_events && !lab_packed;
    if (do_post1) {
    begin(ST_DP);
    if (c->prob_cap > 0 && any_arena) {
        int dp_grid = grid_for(c->prob_cap, 1, 1024);
        const int fill_blocks = ride_fill ? grid_for(labels_n16 / 8 + 1, 512, 512) : 0;
#define FSEG_LAUNCH_DP(NMV, TV, OUTT, NM_RT, DPCLASS, MAXWG)                                                             \
        hipLaunchKernelGGL((k_dp<NMV, TV, OUTT>), dim3((dp_grid < (MAXWG) ? dp_grid : (MAXWG)) + fill_blocks), dim3(TV),     \
                           dp_lds_for(NM_RT, (int)sizeof(OUTT)), ((NMV) == kDpSmall ? q_small : s), st, DPCLASS, NM_RT, c->d_dp_items.as<int>(), pr,          \
                           c->d_prob_desc.as<ProbDesc>(), c->prob_cap, c->d_cand_off.as<i64>(), c->d_cand_y.as<int>(), c->d_iv_part.as<int>(),               \
                           c->d_part_lane_off.as<i64>(), c->d_out.as<unsigned>(), c->tri_cap, c->d_amb.as<unsigned>(),        \
                           c->d_pair_thr.as<int2>(), c->pair_cap, c->P.min_read_support_outside,                              \
                           c->d_chosen.as<unsigned char>(), tiny_max, (dp_grid < (MAXWG) ? dp_grid : (MAXWG)),                \
                           c->d_labels.as<uint4>(), labels_n16 FSEG_TARG)
#define FSEG_LAUNCH_DP_WAVES(OUTT)                                                                                        \
        hipLaunchKernelGGL((k_dp_waves<OUTT>), dim3(grid_for(c->prob_cap, 4, 2048)), dim3(256),                                \
                           dp_lds_for(kDpSmall, (int)sizeof(OUTT)) > 4 * dp_wave_bytes<OUTT>() ? dp_lds_for(kDpSmall, (int)sizeof(OUTT)) : 4 * dp_wave_bytes<OUTT>(), \
                           q_small, st, c->d_dp_items.as<int>(), pr, c->d_prob_desc.as<ProbDesc>(), c->prob_cap,                \
                           c->d_cand_y.as<int>(), c->d_out.as<unsigned>(), c->tri_cap, c->d_amb.as<unsigned>(),                \
                           c->d_pair_thr.as<int2>(), c->pair_cap, c->P.min_read_support_outside, c->d_chosen.as<unsigned char>(), 0); \
        /* the problems of the list with 17 .. 32 candidates: one workgroup each, so that none waits behind another */      \
        hipLaunchKernelGGL((k_dp<kDpSmall, 256, OUTT>), dim3(grid_for(c->prob_cap, 1, 8192)), dim3(256),                       \
                           dp_lds_for(kDpSmall, (int)sizeof(OUTT)), q_mid, st, 0, kDpSmall, c->d_dp_items.as<int>(), pr,       \
                           c->d_prob_desc.as<ProbDesc>(), c->prob_cap, c->d_cand_off.as<i64>(), c->d_cand_y.as<int>(), c->d_iv_part.as<int>(), \
                           c->d_part_lane_off.as<i64>(), c->d_out.as<unsigned>(), c->tri_cap, c->d_amb.as<unsigned>(),        \
                           c->d_pair_thr.as<int2>(), c->pair_cap, c->P.min_read_support_outside,                              \
                           c->d_chosen.as<unsigned char>(), kDpWave, grid_for(c->prob_cap, 1, 8192), (uint4 *)nullptr, (i64)0 FSEG_TARG)
        // 16-bit count tables unless some problem sees >= 65536 reads;
        // 512 threads (8 waves share the c2 loop) when the tables of the largest problem leave room for their scratch.
        // small_batch: one launch over every problem; otherwise one launch per DP class list.
        hipStream_t q_small = c->small_batch ? s : fork(0);      // the DP classes own disjoint problems
        hipStream_t q_mid = c->small_batch ? s : fork(1);
        if (c->dp_wide_counts) {
            const bool wide_wg = dp_lds_for(c->nm_big, 4) + 8 * 1024 <= kLdsPerWg;
            if (!c->small_batch) { FSEG_LAUNCH_DP_WAVES(unsigned); }
            if (wide_wg) { FSEG_LAUNCH_DP(kNMax, 512, unsigned, c->nm_big, c->small_batch ? -1 : 1, 256); }
            else { FSEG_LAUNCH_DP(kNMax, 256, unsigned, c->nm_big, c->small_batch ? -1 : 1, 256); }
        } else {
            if (!c->small_batch) { FSEG_LAUNCH_DP_WAVES(unsigned short); }
            FSEG_LAUNCH_DP(kNMax, 512, unsigned short, c->nm_big, c->small_batch ? -1 : 1, 512);
        }
#undef FSEG_LAUNCH_DP
#undef FSEG_LAUNCH_DP_WAVES
        if (!c->small_batch) { join(0); join(1); }
        if (c->have_huge)
            hipLaunchKernelGGL(k_dp_huge, dim3(dp_grid < 256 ? dp_grid : 256), dim3(512), kHugeDpLds, s, st, c->d_dp_items.as<int>(),
                               pr, c->d_prob_desc.as<ProbDesc>(), c->prob_cap, c->d_cand_y.as<int>(), c->d_out.as<unsigned>(),
                               c->tri_cap, c->d_amb.as<unsigned>(), c->d_pair_thr.as<int2>(), c->pair_cap,
                               c->P.min_read_support_outside, c->d_chosen.as<unsigned char>());
        if (c->have_huge && c->nm_giant > 0 && c->d_giant.p)
            hipLaunchKernelGGL(k_dp_giant, dim3(kGiantWgs), dim3(512), giant_dp_lds(c->nm_giant), s, st, c->d_dp_items.as<int>(),
                               pr, c->d_prob_desc.as<ProbDesc>(), c->prob_cap, c->d_cand_y.as<int>(), c->d_out.as<unsigned>(),
                               c->tri_cap, c->d_amb.as<unsigned>(), c->d_pair_thr.as<int2>(), c->pair_cap,
                               c->P.min_read_support_outside, c->d_chosen.as<unsigned char>(), c->nm_giant,
                               c->d_giant.as<unsigned char>(), (i64)giant_scratch_bytes(c->nm_giant));
    }
    end(ST_DP); begin(ST_REFINE);
    // S6
    hipLaunchKernelGGL(k_segments, dim3(grid_for(K, 1, 8192)), dim3(iv_threads), 0, s, K, c->d_pos_off.as<i64>(),
                       c->d_cand_off.as<i64>(), c->d_cand_y.as<int>(), c->d_y_raw.as<int>(), c->d_blk_pre.as<int>(), c->d_tile_tot.as<int>(), c->d_iv_tile0.as<int>(),
                       c->d_chosen.as<unsigned char>(), flag_final_bits, c->d_rseg_c.as<int>(),
                       c->d_seg_prev.as<int>(), st);
    hipLaunchKernelGGL(k_refine, dim3(2048), dim3(64), 0, s, st, c->d_seg_iv.as<int>(), c->d_rseg_c.as<int>(),
                       c->d_seg_prev.as<int>(), c->d_cand_y.as<int>(), c->d_pos_off.as<i64>(), c->d_y_raw.as<int>(),
                       c->d_w_refine.as<double>(), c->P.radius_refine, c->P.sigma, c->d_g.as<double>(), c->d_pk.as<int>(),
                       c->d_pf.as<unsigned char>(), c->d_kp.as<unsigned char>(), flag_final_bits);
    end(ST_REFINE); begin(ST_FINAL);
    scan_counts(s, bsum, flag_final_bits, &st->n_final, c->d_final_off.as<i64>() + K);
    hipLaunchKernelGGL(k_scan_emit<kEmitPositions>, dim3(scan_grid), dim3(256), 0, s, flag_final_bits,
                       NPOS, bsum, scan_state + 2 * scan_nb, &st->n_final, c->d_final_off.as<i64>() + K, &st->err, (const double *)nullptr, (double *)nullptr, K, c->d_pos_off.as<i64>(),
                       c->d_iv_start.as<int>(), c->d_blk_iv0.as<int>(), c->d_final_y.as<int>(), c->d_final_pos.as<int>(),
                       c->d_final_off.as<i64>(), 0, c->d_final_iv.as<int>());
    // S7, first half: per-column thresholds and the label arena's plan
    hipLaunchKernelGGL(k_label_cols, dim3(grid_for(NPOS / 8 + 1, 256, 2048)), dim3(256), 0, s, K, c->d_final_off.as<i64>(),
                       c->d_final_y.as<int>(), c->d_final_iv.as<int>(), c->d_iv_part.as<int>(), c->d_h_table.as<double>(), c->P.h_len,
                       c->P.threshold_rate, c->d_thr_tab.as<int2>(), c->d_col_thr.as<int2>(), c->d_col_zero.as<unsigned char>(),
                       c->d_part_has2.as<int>(), n_part, c->d_part_iv_off.as<i64>(), c->d_part_rep_off.as<i64>(),
                       c->d_label_off.as<i64>(), st, c->label_cap);
    end(ST_FINAL);
    }   // do_post1
    if (do_post2) {
    begin(ST_LABEL);
    if (c->label_cap > 0) {
        c->run_label_packed = lab_packed; c->labels_unpacked = false;
        if (lab_packed) {
            const i64 n16 = sized ? (label_fill_bytes + 15) / 16 : labels_n16;
            if (n16 > 0)                                            // (one launch: a hipMemsetAsync of 19 MB is two fill kernels here)
                hipLaunchKernelGGL(k_clear, dim3(grid_for(n16 / 4 + 1, 256, 4096)), dim3(256), 0, s, (Status *)nullptr, c->d_packed.as<unsigned>(), n16);
        } else if (!ride_fill) {                                    // no DP launch carried the fill
            const i64 n16 = sized ? (label_fill_bytes + 15) / 16 : labels_n16;
            if (n16 > 0)
                hipLaunchKernelGGL(k_label_zero, dim3(grid_for(n16 / 8 + 1, 256, 4096)), dim3(256), 0, s, c->d_labels.as<uint4>(), n16);
        }
        hipLaunchKernelGGL(k_label_reads, dim3(grid_for((i64)c->n_rep_blocks * kLabelSplit, 1, 65536)), dim3(256), 0, s, c->n_rep_blocks,
                           c->d_rb_part.as<int>(), c->d_rb_r0.as<int>(), c->d_label_off.as<i64>(), c->label_cap, n_part,
                           c->d_part_iv_off.as<i64>(), c->d_part_rep_off.as<i64>(), c->d_final_off.as<i64>(),
                           c->d_final_pos.as<int>(), c->d_col_thr.as<int2>(), c->d_rep_exon_off.as<i64>(),
                           c->d_ex_ts.as<int>(), c->d_ex_te.as<int>(), c->d_col_zero.as<unsigned char>(),
                           c->d_part_has2.as<int>(), c->d_labels.as<unsigned char>(), lab_packed ? c->d_packed.as<unsigned>() : (unsigned *)nullptr);
    }
    end(ST_LABEL);
    }   // do_post2
    if (segs & SEG_STATUS) HIP_TRY(c, hipMemcpyAsync(c->h_status, st, sizeof(Status), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, fj_err);
    HIP_TRY(c, hipGetLastError());
    return FSEG_OK;
}

// wait for the stream: poll for a while (a run is well under a millisecond on a resident batch and the blocking
// wait's wake-up costs tens of microseconds), then block
hipError_t wait_stream(fseg_ctx *c) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        hipError_t e = hipStreamQuery(c->stream);
        if (e != hipErrorNotReady) { (void)hipGetLastError(); return e; }     // NotReady must not linger as the thread's last error
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
    (void)hipGetLastError();
    return hipStreamSynchronize(c->stream);
}

// the device-side validation of the last upload (read once, with the first status record of the batch)
int check_prep(fseg_ctx *c) {
    if (c->prep_checked) return FSEG_OK;
    c->prep_checked = true;
    const PrepStatus &ps = *c->h_prep;
    if (!ps.err) return FSEG_OK;
    int q = 0;
    for (int i = 1; i < 4; ++i) if (((ps.err >> i) & 1u) && (!((ps.err >> q) & 1u) || ps.bad_rep[i] < ps.bad_rep[q])) q = i;
    if (!((ps.err >> q) & 1u)) for (q = 0; q < 4 && !((ps.err >> q) & 1u); ++q) {}
    const long long r = (long long)ps.bad_rep[q];
    c->have_batch = false;
    switch (1u << q) {
        case kPrepExonEnds: return fail(c, FSEG_ERR_INPUT, "rep %lld: exon with start >= end (py/freddie_segment.py:160)", r);
        case kPrepExonOrder: return fail(c, FSEG_ERR_INPUT, "rep %lld: exons out of order (py/freddie_segment.py:158)", r);
        case kPrepExonInterval: return fail(c, FSEG_ERR_INPUT, "rep %lld: an exon does not lie inside one tint interval (py/freddie_segment.py:668)", r);
        default: return fail(c, FSEG_ERR_INPUT, "rep %lld has no exons", r);
    }
}

// what the status record of a finished run says about the run's input (the reference's assertions)
int run_input_errors(fseg_ctx *c, const Status &s) {
    if (s.err & kErrExonInterval) return fail(c, FSEG_ERR_INPUT, "an exon does not lie inside one tint interval (py/freddie_segment.py:668)");
    if (s.err & kErrBreakAssert) return fail(c, FSEG_ERR_INPUT, "break_large_problems: candidate window out of range or no positive signal (py/freddie_segment.py:640-643)");
    if (s.err & kErrWideMissed) return fail(c, FSEG_ERR_HIP, "internal: a problem keeps more reads than were counted for it (kErrWideMissed)");
    if (s.err & kErrProblemTooLarge) return fail(c, FSEG_ERR_UNSUPPORTED, "a DP problem has more than %d candidates (max_problem_size beyond 1000 is not supported)", kNGiant);
    return FSEG_OK;
}

// the giant-problem kernels' LDS carve-up and scratch for a run whose largest problem has max_n candidates
int prepare_giant(fseg_ctx *c, int max_n) {
    if (max_n <= kNHuge || max_n > kNGiant) { if (max_n <= kNHuge) c->nm_giant = 0; return FSEG_OK; }
    int want = (max_n + 7) & ~7;
    if (want > kNGiant) want = kNGiant;
    if (want > c->nm_giant || c->nm_giant == 0) { c->nm_giant = want; drop_graph(c); }
    TRY(ensure(c, c->d_giant, (size_t)kGiantWgs * giant_scratch_bytes(c->nm_giant)));
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_score_giant), hipFuncAttributeMaxDynamicSharedMemorySize, (int)giant_score_lds(kNGiant));
    if (e != hipSuccess) return fail(c, FSEG_ERR_HIP, "hipFuncSetAttribute(k_score_giant): %s", hipGetErrorString(e));
    return FSEG_OK;
}

// what the lists of the batch hold, from the status record of a run (or of the sizing pass)
void note_counts(fseg_ctx *c, const Status &s) {
    for (int q = 0; q < 3; ++q) { c->n_solve[q] = (i64)s.solve_cls[q]; c->n_dp_cls[q] = (i64)s.dp_cls[q]; }
    c->n_tiny = (i64)s.n_tiny;
    for (int q = 0; q < 3; ++q) c->n_wide[q] = (i64)s.wide_cls[q];
    for (int q = 0; q < 4; ++q) c->n_cls_work[q] = (i64)s.cls_work[q];
    c->n_arena_prob = (i64)s.dp_cls[0] + (i64)s.dp_cls[1] + (i64)s.dp_cls[2];
    if (!c->counts_known) drop_graph(c);
    c->counts_known = true;
}

// launch parameters that follow from the sizes of a run (exact in a sized run, last run's otherwise)
void adapt_to(fseg_ctx *c, const Status &s) {
    const bool old_small = c->small_batch, old_self = c->prob_self_scan, old_tiny = c->tiny_on;
    const int old_nm = c->nm_big;
    c->small_batch = (i64)s.n_prob <= 256 && (i64)s.n_work <= 1024;
    c->tiny_on = (i64)s.n_prob > c->tiny_from;                  // depends on the problem count only, which k_tiny does not change
    const bool old_fuse = c->fuse_on;
    c->fuse_on = (i64)s.max_ln <= c->fuse_lanes;                 // (the widest problem does not depend on the split either)
    c->wide_solve = (i64)s.max_ln > kFuseLanes;                  // some problem needs the 16-bit counters
    c->max_ln = (i64)s.max_ln;
    // (the lists' sizes were counted under the old division of the problems: a run under the new one must not take them from the host --
    // an unsized first run of a batch with more than tiny_from problems turns k_tiny's share on for the replay, whose list bounds then
    // came from the run without it: kErrOverflowNm on every attempt, "arena sizing did not converge"; found in round 5)
    if (old_fuse != c->fuse_on || old_tiny != c->tiny_on) { c->counts_known = false; drop_graph(c); }
    c->prob_self_scan = (i64)s.n_cand <= fseg_ctx::kProbSelfMax;
    {   // the big-problem LDS carve-up: the largest problem (+ headroom, multiple of 4)
        int want = (int)s.max_n + 3;
        want = (want + 3) & ~3;
        if (want < kClsMid + 4) want = kClsMid + 4;
        if (want > kNMax) want = kNMax;
        c->nm_big = want;
    }
    if (old_small != c->small_batch || old_nm != c->nm_big || old_self != c->prob_self_scan || old_tiny != c->tiny_on) drop_graph(c);
}

void collect_stage_times(fseg_ctx *c, int timed_graphs) {
    if (!c->profiling) return;
    for (int i = 0; i < ST_REPORTED; ++i) c->stage_ms[i] = 0.f;
    if (timed_graphs == 2) {
        (void)hipEventElapsedTime(&c->stage_ms[ST_GRAPH_PRE], c->ev_g[0], c->ev_g[1]);
        (void)hipEventElapsedTime(&c->stage_ms[ST_SCORE], c->ev_g[1], c->ev_g[2]);
        (void)hipEventElapsedTime(&c->stage_ms[ST_GRAPH_POST], c->ev_g[2], c->ev_g[3]);
    } else if (c->last_sized || c->run_plain) {
        for (int i = 0; i < ST_COUNT; ++i) {
            if (!c->profile_all && i != ST_SCORE) continue;
            if (hipEventElapsedTime(&c->stage_ms[i], c->ev_b[i], c->ev_e[i]) != hipSuccess) { c->stage_ms[i] = 0.f; (void)hipGetLastError(); }
        }
    }
}

// A device-side waiter gave up: the run is redone with events (run_events_only), the context keeps its waiters for later runs
// unless it keeps happening (three times: events for good).  Counted and exposed (FSEG_TAP_SYNC): a time-out is a stall of the
// waiter's limit plus a rerun, and nothing else would show it.
static void note_sync_timeout(fseg_ctx *c) {
    c->run_events_only = true;
    if (++c->sync_timeouts >= 3 && c->dev_sync) {
        c->dev_sync = false;
        if (c->trace) fprintf(stderr, "[fseg] %u runs lost a device-side waiter to its time limit: this context forks and joins with events from now on\n", c->sync_timeouts);
    } else if (c->trace) fprintf(stderr, "[fseg] a device-side waiter reached its time limit: the run is redone with events\n");
}

// wait for the run; grow arenas and re-run if a capacity was exceeded (never after a sized run: its capacities are exact)
static void set_in_flight(fseg_ctx *c, bool on) {
    if (!on) release_device(c);
    if (c->counted_in_flight == on || c->device < 0 || c->device >= 64) return;
    c->counted_in_flight = on;
    g_in_flight[c->device].fetch_add(on ? 1 : -1, std::memory_order_acq_rel);
}
static void claim_device(fseg_ctx *c) {
    if (c->owns_device || c->device < 0 || c->device >= 64 || !c->use_fork || others_in_flight(c)) return;
    const fseg_ctx *nobody = nullptr;
    if (g_owner[c->device].compare_exchange_strong(nobody, c, std::memory_order_acq_rel)) c->owns_device = true;
}
static void release_device(fseg_ctx *c) {
    if (!c->owns_device) return;
    c->owns_device = false;
    g_owner[c->device].store(nullptr, std::memory_order_release);
}
static bool others_in_flight(const fseg_ctx *c) {
    if (c->device < 0 || c->device >= 64) return false;
    return g_in_flight[c->device].load(std::memory_order_acquire) - (c->counted_in_flight ? 1 : 0) > 0;
}
// the run owns the device and is worth branching (see enqueue_run)
static bool would_fork(const fseg_ctx *c) { return c->use_fork && !c->small_batch && c->owns_device && c->side[0] != nullptr; }
static int finish_run_impl(fseg_ctx *c);
int finish_run(fseg_ctx *c) {
    const int rc = finish_run_impl(c);
    if (!c->pending) set_in_flight(c, false);
    return rc;
}
static int finish_run_impl(fseg_ctx *c) {
    for (int attempt = 0; attempt < 4; ++attempt) {
        HIP_TRY(c, wait_stream(c));
        TRY(check_prep(c));
        const Status &s = *c->h_status;
        unsigned ovf = s.err & (kErrOverflowPairs | kErrOverflowTri | kErrOverflowWork | kErrOverflowLabels |
                                kErrOverflowProblems | kErrOverflowChunks | kErrOverflowCov);
        if (s.err & kErrOverflowNm) { ovf |= kErrOverflowNm; c->nm_big = kNMax; }
        if (s.err & kErrWaveStage) { ovf |= kErrWaveStage; c->use_wave = false; c->counts_known = false; drop_graph(c); }    // k_tiny / k_solve fetch exons read by read
        if (s.err & kErrSyncTimeout) { ovf |= kErrSyncTimeout; note_sync_timeout(c); }   // a device-side waiter gave up (the stage was skipped): this run again, with events
        if ((i64)s.max_ln >= 65536 && !c->dp_wide_counts) { ovf |= kErrNeedWideDp; c->dp_wide_counts = true; drop_graph(c); }
        if (s.err & kErrScanStall) { ovf |= kErrScanStall; c->scan_single_max = 0; c->force_scan_stall = false; drop_graph(c); }
        if (s.dp_cls[2] > 0 && !c->have_huge) { ovf |= kErrProblemTooLarge << 16; c->have_huge = true; drop_graph(c); }   // rerun with the huge-problem kernels
        if ((int)s.max_n > kNHuge && (int)s.max_n <= kNGiant && (int)s.max_n > c->nm_giant) {                             // ... and the giant ones, sized for this run's largest problem
            ovf |= kErrProblemTooLarge << 17; drop_graph(c);
            TRY(prepare_giant(c, (int)s.max_n));
        }
        bool need = ovf != 0 || (i64)s.n_prob > c->prob_cap || (i64)s.n_work > c->work_cap ||
                    (i64)s.pair_used > c->pair_cap || (i64)s.tri_used > c->tri_cap || (i64)s.label_bytes > c->label_cap ||
                    (i64)s.n_vchunks > c->chunk_cap || (i64)s.cov_used > c->cov_cap;
        if (!need) {
            c->pending = false;
            c->ran = true;
            const int timed_graphs = c->run_plain ? 0 : c->n_graphs;
            collect_stage_times(c, timed_graphs);
            note_counts(c, s);
            adapt_to(c, s);
            return run_input_errors(c, s);
        }
        auto grow = [](i64 need_v, i64 cap) { return need_v > cap ? need_v + need_v / 8 + 64 : cap; };
        c->prob_cap = grow((i64)s.n_prob, c->prob_cap);
        c->work_cap = grow((i64)s.n_work, c->work_cap);
        c->pair_cap = grow((i64)s.pair_used, c->pair_cap);
        c->tri_cap = grow((i64)s.tri_used, c->tri_cap);
        c->label_cap = grow((i64)s.label_bytes, c->label_cap);
        c->cov_cap = grow((i64)s.cov_used, c->cov_cap);
        TRY(alloc_arenas(c));
        c->last_sized = false;
        TRY(enqueue_run(c, SEG_ALL));
    }
    {
        const Status &s = *c->h_status;
        return fail(c, FSEG_ERR_HIP, "arena sizing did not converge (status %#x; %llu problems / cap %lld, %llu work items / %lld, pairs %llu / %lld, triples %llu / %lld, "
                    "coverage %llu / %lld, label bytes %llu / %lld, chunks %llu / %lld, widest problem sees %u reads)", s.err,
                    (unsigned long long)s.n_prob, (long long)c->prob_cap, (unsigned long long)s.n_work, (long long)c->work_cap, (unsigned long long)s.pair_used, (long long)c->pair_cap,
                    (unsigned long long)s.tri_used, (long long)c->tri_cap, (unsigned long long)s.cov_used, (long long)c->cov_cap, (unsigned long long)s.label_bytes, (long long)c->label_cap,
                    (unsigned long long)s.n_vchunks, (long long)c->chunk_cap, s.max_ln);
    }
}

// Which of the context's side streams run BESIDE its main stream (see side_ok).  Once per context, by the first run that owns the
// device: per side stream a k_probe_wait on it, then a k_signal on the (drained) main stream.  The limit is generous (2 ms: kernels of
// other contexts may be ahead of the signal in the main stream's hardware queue); a probe that runs out only costs that stream its
// waiters (events instead), never correctness.
int probe_side_queues(fseg_ctx *c) {
    if (c->side_probed || !c->side[0] || !c->d_sync.p) return FSEG_OK;
    c->side_probed = true;
    SyncWords *sw = c->d_sync.as<SyncWords>();
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < fseg_ctx::kSide; ++k) {
        const unsigned gen = ++c->probe_gen;
        unsigned res = 0;
        HIP_TRY(c, hipMemsetAsync(&sw->probe_result, 0, sizeof(unsigned), c->side[k]));
        hipLaunchKernelGGL(k_probe_wait, dim3(1), dim3(64), 0, c->side[k], &sw->probe_word, gen, &sw->probe_result, 200000u);
        hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, c->stream, &sw->probe_word, gen);
        HIP_TRY(c, hipStreamSynchronize(c->side[k]));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, copy_sync(c, &res, &sw->probe_result, sizeof res, hipMemcpyDeviceToHost, c->side[k]));
        c->side_ok[k] = res == 1u;
    }
    if (c->trace) fprintf(stderr, "[fseg] side streams beside the main stream (device-side waiters allowed): %d %d %d\n", (int)c->side_ok[0], (int)c->side_ok[1], (int)c->side_ok[2]);
    return FSEG_OK;
}

// First run of a batch: launched in three pieces with the host reading the status record in between, so every arena is
// sized exactly before anything is written into it -- no guessed capacities, no overflow re-run.
//   A  histogram .. problem scan   -> problems, work items, pairs, triples, coverage elements, largest / widest problem
//   B  problem list .. label plan  -> final positions, label bytes
//   C  labels
// The two waits cost a few tens of microseconds each; with two contexts per GPU (the CLI) another batch's kernels fill them.
int run_sized(fseg_ctx *c) {
    Tick tk;
    double t_a = 0, t_b = 0;
    for (int attempt = 0; attempt < 3; ++attempt) {          // (attempts = redone scans; a waiter's time-out has a retry of its own, once per run)
        // A
        TRY(enqueue_run(c, SEG_PRE1 | SEG_STATUS, true));
        HIP_TRY(c, wait_stream(c));
        TRY(check_prep(c));
        Status s = *c->h_status;
        if (s.err & kErrScanStall) { c->scan_single_max = 0; c->force_scan_stall = false; continue; }     // redo with the three-pass scan
        {   // k_tiny's share of the problems was decided from the previous batch: if this batch decides otherwise, the
            // arena sizes change with it -- redo the (cheap) problem scan under the right setting
            const bool tiny = (i64)s.n_prob > c->tiny_from;
            const bool fuse = (i64)s.max_ln <= c->fuse_lanes;
            if (tiny != c->tiny_on || fuse != c->fuse_on) {
                if (c->trace) fprintf(stderr, "[fseg] problem split changed (tiny %d -> %d, fused %d -> %d; %llu problems, widest sees %u reads): rescan\n",
                                      (int)c->tiny_on, (int)tiny, (int)c->fuse_on, (int)fuse, (unsigned long long)s.n_prob, s.max_ln);
                const bool fuse_turned_on = fuse && !c->fuse_on;
                c->tiny_on = tiny; c->fuse_on = fuse;
                const int pg = grid_for(c->NPOS / 8 / kProbBlock + 1, 1, 1024);
                Status *st = c->d_status.as<Status>();
                if (fuse_turned_on)     // the first pass did not count the reads the wide problems keep: nobody was going to ask
                    hipLaunchKernelGGL(k_prob_range, dim3(pg), dim3(kRangeThreads), 0, c->stream, st, c->d_cand_pn.as<int>(),
                                       c->d_seg_iv.as<int>(), c->d_cand_y.as<int>(), c->d_iv_part.as<int>(), c->d_iv_start.as<int>(),
                                       c->d_part_lane_off.as<i64>(), c->d_lane_start.as<int>(), c->d_lane_pmax.as<int>(),
                                       c->d_cand_ll.as<int>(), c->d_cand_ln.as<int>(), c->d_cand_wide.as<unsigned char>(), c->d_lane_lx.as<int2>(),
                                       c->d_lex.as<int2>(), c->wide_by_seen ? 1 : 0, split_of(c, tiny, fuse).fuse_lanes, (i64 *)nullptr, split_of(c, tiny, fuse));
                // (the scan ADDS to the per-class counts of wide problems: the first scan's must not stay in them)
                HIP_TRY(c, hipMemsetAsync(reinterpret_cast<char *>(st) + offsetof(Status, wide_cls), 0, sizeof(st->wide_cls), c->stream));
                hipLaunchKernelGGL(k_prob_scan1, dim3(pg), dim3(256), 0, c->stream, st, c->d_cand_pn.as<int>(), c->d_cand_ln.as<int>(), c->d_cand_wide.as<unsigned char>(),
                                   c->d_prob_bs.as<i64>(), split_of(c, tiny, fuse));
                hipLaunchKernelGGL(k_prob_scan2, dim3(1), dim3(256), 0, c->stream, st, c->d_prob_bs.as<i64>());
                HIP_TRY(c, hipMemcpyAsync(c->h_status, st, sizeof(Status), hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(c, wait_stream(c));
                s = *c->h_status;
            }
        }
        t_a = tk.ms();
        if ((s.err & (kErrExonInterval | kErrBreakAssert | kErrProblemTooLarge))) {     // the reference would have aborted here
            c->pending = false; c->ran = false;
            return run_input_errors(c, s);
        }
        auto atleast = [](i64 &cap, i64 v) { if (cap < v) cap = v; };
        atleast(c->prob_cap, (i64)s.n_prob); atleast(c->work_cap, (i64)s.n_work); atleast(c->pair_cap, (i64)s.pair_used);
        atleast(c->tri_cap, (i64)s.tri_used); atleast(c->cov_cap, (i64)s.cov_used);
        c->have_huge = s.dp_cls[2] > 0;
        TRY(prepare_giant(c, (int)s.max_n));
        c->dp_wide_counts = (i64)s.max_ln >= 65536;
        note_counts(c, s);
        adapt_to(c, s);
        TRY(alloc_arenas(c));
        // B
        TRY(enqueue_run(c, SEG_PRE2 | SEG_SCORE | SEG_POST1 | SEG_STATUS, true));
        HIP_TRY(c, wait_stream(c));
        s = *c->h_status;
        t_b = tk.ms();
        if (s.err & kErrScanStall) { c->scan_single_max = 0; c->force_scan_stall = false; continue; }
        const unsigned bad = s.err & (kErrOverflowPairs | kErrOverflowTri | kErrOverflowWork | kErrOverflowProblems | kErrOverflowChunks |
                                      kErrOverflowCov | kErrOverflowNm | kErrNeedWideDp);
        if (bad) return fail(c, FSEG_ERR_HIP, "internal: a sized run overflowed an arena (status %#x)", s.err);
        // (what B's scoring kernels can raise besides: a result that is garbage fails HERE, not after the label stage; a wave kernel
        // that met a read it cannot stage turns the wave kernels off and the sized attempt starts over)
        if (s.err & kErrSyncTimeout) {                       // a device-side waiter gave up (the stage was skipped): once more, with events
            if (c->run_events_only) return fail(c, FSEG_ERR_HIP, "internal: a waiter timed out in a run without waiters (status %#x)", s.err);
            note_sync_timeout(c); --attempt; continue;
        }
        if (s.err & kErrWideMissed) { c->pending = false; c->ran = false; return run_input_errors(c, s); }
        if (s.err & kErrWaveStage) { c->use_wave = false; c->counts_known = false; drop_graph(c); continue; }
        if ((s.err & (kErrExonInterval | kErrBreakAssert | kErrProblemTooLarge))) {
            c->pending = false; c->ran = false;
            return run_input_errors(c, s);
        }
        // C
        if ((i64)s.label_bytes > c->label_cap) {
            c->label_cap = (i64)s.label_bytes;
            TRY(ensure_label_arena(c));
            if (!labels_packed(c)) c->label_cap = (i64)c->d_labels.cap - 16;
        }
        TRY(enqueue_run(c, SEG_POST2, true, (i64)s.label_bytes));
        c->h_status->err &= ~kErrOverflowLabels;       // the plan was made against the old capacity; the arena has been grown since
        c->pending = true;
        c->last_sized = true;
        if (c->trace) fprintf(stderr, "[fseg] run (sized): A %.3f ms, B %.3f ms, C enqueued %.3f ms; %llu problems (fused %llu / %llu / %llu, widest sees %u reads), %llu work items, %llu label bytes\n",
                              t_a, t_b, tk.ms(), (unsigned long long)s.n_prob, (unsigned long long)s.solve_cls[0], (unsigned long long)s.solve_cls[1],
                              (unsigned long long)s.solve_cls[2], s.max_ln, (unsigned long long)s.n_work, (unsigned long long)s.label_bytes);
        return FSEG_OK;
    }
    return fail(c, FSEG_ERR_HIP, "the compaction scans stalled repeatedly");
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C-ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

int fseg_abi_version(void) { return FSEG_ABI_VERSION; }

#ifndef FREDDIE_SOURCE_HASH
#define FREDDIE_SOURCE_HASH ""
#endif
static const char freddie_source_stamp[] __attribute__((used)) = "FREDDIE_SRC_HASH=" FREDDIE_SOURCE_HASH;
const char *fseg_source_hash(void) { return freddie_source_stamp + 17; }

const char *fseg_last_error(const fseg_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int fseg_create(int device, fseg_ctx **out) {
    if (!out) return FSEG_ERR_ARG;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_create_error = std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0") +
                         " (this library has no CPU fallback)";
        return FSEG_ERR_HIP;
    }
    if (device < 0 || device >= n) { g_create_error = "device ordinal out of range"; return FSEG_ERR_ARG; }
    fseg_ctx *c = new fseg_ctx();
    c->device = device;
    e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_status, sizeof(Status), hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_prep, sizeof(PrepStatus), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(&c->d_status.p, sizeof(Status));
    if (e == hipSuccess) e = hipMalloc(&c->d_prep.p, sizeof(PrepStatus));
    if (e == hipSuccess) e = hipMalloc(&c->d_tacc.p, kTaccBytes);
    if (e == hipSuccess) e = hipMemsetAsync(c->d_tacc.p, 0, kTaccBytes, c->stream);
    if (e == hipSuccess) e = hipMalloc(&c->d_sync.p, sizeof(SyncWords));
    if (e == hipSuccess) e = hipMemsetAsync(c->d_sync.p, 0, sizeof(SyncWords), c->stream);
    for (int i = 0; e == hipSuccess && i < ST_COUNT; ++i) { e = hipEventCreate(&c->ev_b[i]); if (e == hipSuccess) e = hipEventCreate(&c->ev_e[i]); }
    for (int i = 0; e == hipSuccess && i < 4; ++i) e = hipEventCreate(&c->ev_g[i]);
    // The side streams: at once for the first context of a device (the usual single-context user gets its four streams on
    // four hardware queues), otherwise by the first run that forks: the runtime spreads a process's streams over its few
    // hardware queues in creation order, and contexts that take turns on a device use their main streams only -- created
    // back to back, those land on different queues.
    c->counted_live = device >= 0 && device < 64;
    if (c->counted_live && g_live[device].fetch_add(1) == 0)
        for (int i = 0; e == hipSuccess && i < fseg_ctx::kSide; ++i) e = hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking);
    for (int i = 0; e == hipSuccess && i < fseg_ctx::kForkEvents; ++i) e = hipEventCreateWithFlags(&c->fj[i], hipEventDisableTiming);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_score<kNMax>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)ScoreCfg<kNMax>::kLds);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_dp<kNMax, 256, unsigned>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)dp_lds_for(kNMax, 4));
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_dp<kNMax, 512, unsigned>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kLdsPerWg - 8 * 1024));
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_dp<kNMax, 512, unsigned short>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)dp_lds_for(kNMax, 2));
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_score_huge), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)kHugeScoreLds);
    {
        auto lds_attr = [&](const void *f, size_t bytes) { if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); };
        lds_attr(reinterpret_cast<const void *>(k_solve<kNMax, unsigned char, int, false>), solve_lds_for(kNMax, kNMax + 1, 1));
        lds_attr(reinterpret_cast<const void *>(k_solve<kNMax, unsigned char, i64, false>), solve_lds_for(kNMax, kNMax + 1, 1));
        lds_attr(reinterpret_cast<const void *>(k_solve<kNMax, unsigned short, int, false>), solve_lds_for(kNMax, kNMax + 1, 2));
        lds_attr(reinterpret_cast<const void *>(k_solve<kNMax, unsigned short, i64, false>), solve_lds_for(kNMax, kNMax + 1, 2));
        lds_attr(reinterpret_cast<const void *>(k_solve<kNMax, unsigned char, int, true>), solve_lds_for(kNMax, kNMax + 1, 1));
        lds_attr(reinterpret_cast<const void *>(k_solve<kNMax, unsigned short, int, true>), solve_lds_for(kNMax, kNMax + 1, 2));
        lds_attr(reinterpret_cast<const void *>(k_dpw<kNMax, unsigned char, int, 64>), dpw_lds_for(kNMax, 4, 1));
        lds_attr(reinterpret_cast<const void *>(k_dpw<kNMax, unsigned short, int, 64>), dpw_lds_for(kNMax, 4, 2));
        lds_attr(reinterpret_cast<const void *>(k_dpw<kNMax, unsigned char, i64, 64>), dpw_lds_for(kNMax, 8, 1));
        lds_attr(reinterpret_cast<const void *>(k_dpw<kNMax, unsigned short, i64, 64>), dpw_lds_for(kNMax, 8, 2));
        lds_attr(reinterpret_cast<const void *>(k_dpw<kNMax, unsigned char, int, 512>), dpw_lds_for(kNMax, 4, 1));
        lds_attr(reinterpret_cast<const void *>(k_dpw<kNMax, unsigned short, int, 512>), dpw_lds_for(kNMax, 4, 2));
        lds_attr(reinterpret_cast<const void *>(k_dpw<kNMax, unsigned char, i64, 512>), dpw_lds_for(kNMax, 8, 1));
        lds_attr(reinterpret_cast<const void *>(k_dpw<kNMax, unsigned short, i64, 512>), dpw_lds_for(kNMax, 8, 2));
    }
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_dp_huge), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)kHugeDpLds);
    if (e != hipSuccess) {
        g_create_error = std::string("context creation failed: ") + hipGetErrorString(e);
        if (c->counted_live) g_live[device].fetch_sub(1);
        delete c;
        return FSEG_ERR_HIP;
    }
    c->d_status.cap = sizeof(Status); c->d_prep.cap = sizeof(PrepStatus); c->d_tacc.cap = kTaccBytes; c->d_sync.cap = sizeof(SyncWords);
    auto flag = [](const char *name) { const char *v = getenv(name); return v && v[0] == '1'; };
    { const char *v = getenv("FSEG_DEV_SYNC"); if (v && v[0] == '0') c->dev_sync = false; }
    if (flag("FSEG_LABEL_BYTES")) c->label_packed_ok = false;
    { const char *v = getenv("FSEG_THR_PART"); if (v && (v[0] == '0' || v[0] == '1')) c->thr_part = v[0] - '0'; }
    { const char *v = getenv("FSEG_SYNC_TICKS"); if (v && v[0] && atoll(v) > 0) c->sync_ticks = (unsigned)atoll(v); }
    if (flag("FSEG_NO_GRAPH")) c->use_graph = false;
    if (flag("FSEG_NO_FORK")) c->use_fork = false;
    if (flag("FSEG_NO_FUSE")) c->use_fuse = false;
    if (flag("FSEG_NO_WAVE")) c->use_wave = false;
    if (flag("FSEG_FORCE_KEY64")) c->force_key64 = true;
    if (flag("FSEG_WIDE_BY_SEEN")) c->wide_by_seen = true;
    if (const char *e = getenv("FSEG_SPLIT_DP")) c->split_dp = atoi(e) & 15;
    if (const char *e = getenv("FSEG_SCORE_PLAN")) snprintf(c->score_plan, sizeof c->score_plan, "%s", e);
    {
        int seen[4] = {0, 0, 0, 0};
        static const char kinds[] = "BMST";
        for (const char *q = c->score_plan; *q; ++q) { const char *at = strchr(kinds, *q); if (at) ++seen[at - kinds]; }
        int bars = 0;
        for (const char *q = c->score_plan; *q; ++q) bars += *q == '|';
        if (seen[0] != 1 || seen[1] != 1 || seen[2] != 1 || seen[3] != 1 || bars > fseg_ctx::kSide) c->score_plan[0] = 0;   // (a stream per segment)
    }
    { const char *v = getenv("FSEG_FUSE_LANES"); if (v && v[0] && atoi(v) > 0 && atoi(v) <= kFuseLanesWide) c->fuse_lanes = atoi(v); }
    if (flag("FSEG_NO_SIZED")) c->use_sized = false;
    if (flag("FSEG_TRACE")) c->trace = true;
    if (flag("FSEG_GLOBAL_SORT")) c->force_global_sort = true;
    if (flag("FSEG_FORCE_SCAN_STALL")) c->force_scan_stall = true;
    { const char *tf = getenv("FSEG_TINY_FROM"); if (tf && tf[0]) c->tiny_from = atoll(tf); }
    { const char *sm = getenv("FSEG_SCAN_SINGLE_MAX"); if (sm && sm[0]) c->scan_single_max = atoll(sm); }
    *out = c;
    return FSEG_OK;
}

void fseg_destroy(fseg_ctx *c) {
    if (!c) return;
    set_in_flight(c, false);
    release_device(c);
    if (c->counted_live) { g_live[c->device].fetch_sub(1); c->counted_live = false; }
    if (c->hsa_agent >= 0) { (void)hsa_signal_destroy(c->hsa_sig); c->hsa_agent = -2; }
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    drop_graph(c);
    Slab *slabs[] = {&c->slab_in, &c->slab_pos, &c->slab_arena};
    for (Slab *s : slabs) if (s->p) (void)hipFree(s->p);
    DevBuf *bufs[] = {&c->d_labels, &c->d_packed, &c->d_sort_tmp, &c->d_w_main, &c->d_w_refine, &c->d_h_table, &c->d_thr_tab, &c->d_status, &c->d_prep, &c->d_tacc, &c->d_sync, &c->d_giant};
    for (DevBuf *b : bufs) if (b->p) (void)hipFree(b->p);
    if (c->h_stage.p) (void)hipHostFree(c->h_stage.p);
    if (c->h_res.p) (void)hipHostFree(c->h_res.p);
    if (c->h_status) (void)hipHostFree(c->h_status);
    if (c->h_prep) (void)hipHostFree(c->h_prep);
    for (int i = 0; i < ST_COUNT; ++i) { if (c->ev_b[i]) (void)hipEventDestroy(c->ev_b[i]); if (c->ev_e[i]) (void)hipEventDestroy(c->ev_e[i]); }
    for (int i = 0; i < 4; ++i) if (c->ev_g[i]) (void)hipEventDestroy(c->ev_g[i]);
    for (int i = 0; i < fseg_ctx::kSide; ++i) if (c->side[i]) { (void)hipStreamSynchronize(c->side[i]); (void)hipStreamDestroy(c->side[i]); }
    for (int i = 0; i < fseg_ctx::kForkEvents; ++i) if (c->fj[i]) (void)hipEventDestroy(c->fj[i]);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int fseg_set_params(fseg_ctx *c, const fseg_params *p) {
    if (!c || !p) return FSEG_ERR_ARG;
    // the ranges parse_args() asserts (py/freddie_segment.py:104-109)
    if (!(p->threshold_rate >= 0.5 && p->threshold_rate <= 1.0)) return fail(c, FSEG_ERR_ARG, "threshold_rate must be in [0.5, 1]");
    if (!(p->variance_factor > 0 && p->variance_factor < 10)) return fail(c, FSEG_ERR_ARG, "variance_factor must be in (0, 10)");
    if (!(p->sigma > 0 && p->sigma <= 50)) return fail(c, FSEG_ERR_ARG, "sigma must be in (0, 50]");
    if (!(p->max_problem_size > 3)) return fail(c, FSEG_ERR_ARG, "max_problem_size must be > 3");
    if (p->min_read_support_outside < 0) return fail(c, FSEG_ERR_ARG, "min_read_support_outside must be >= 0");
    if (p->radius_main < 0 || p->radius_main > kMaxRadius || p->radius_refine < 0 || p->radius_refine > kMaxRadius)
        return fail(c, FSEG_ERR_ARG, "Gaussian radius out of range");
    if (!p->w_main || !p->w_refine || !p->h_table || p->h_len <= 0) return fail(c, FSEG_ERR_ARG, "missing weight / threshold tables");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    drop_graph(c);
    c->P = *p;
    c->w_main.assign(p->w_main, p->w_main + p->radius_main + 1);
    c->w_refine.assign(p->w_refine, p->w_refine + p->radius_refine + 1);
    c->h_table.assign(p->h_table, p->h_table + p->h_len);
    c->P.w_main = c->w_main.data(); c->P.w_refine = c->w_refine.data(); c->P.h_table = c->h_table.data();
    TRY(upload_vec(c, c->d_w_main, c->w_main.data(), c->w_main.size()));
    TRY(upload_vec(c, c->d_w_refine, c->w_refine.data(), c->w_refine.size()));
    TRY(upload_vec(c, c->d_h_table, c->h_table.data(), c->h_table.size()));
    TRY(ensure(c, c->d_thr_tab, (size_t)kThrTab * sizeof(int2)));
    hipLaunchKernelGGL(k_thr_table, dim3(kThrTab / 256), dim3(256), 0, c->stream, c->d_h_table.as<double>(), c->P.h_len,
                       c->P.threshold_rate, c->d_thr_tab.as<int2>());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // lo < 0 (label_thresholds: no v >= 0 with v / L < 1 - h) exactly where h >= 1: the rate itself, or an entry of the table that
    // round(y, 2) took to 1.0 (segment lengths start at 1)
    c->label_has2 = !(p->threshold_rate < 1.0);
    for (int L = 1; L < p->h_len; ++L) if (!(c->h_table[(size_t)L] < 1.0)) c->label_has2 = true;
    c->have_params = true;
    c->ran = false;          // results of an earlier run belong to other parameters
    c->counts_known = false; // ... and so do the sizes of its lists
    return FSEG_OK;
}

// One pinned staging image, one host-to-device copy, then the device derives what used to be host work (validation
// of every exon, the per-partition sort of the reps, the lane list, the histogram chunks' lane ranges).  Returns
// without waiting for the device: what the device-side validation finds is reported by the first call that waits
// (fseg_run / fseg_sync / fseg_download ...).
static int upload_impl(fseg_ctx *c, const fseg_batch *b);
int fseg_upload(fseg_ctx *c, const fseg_batch *b) {
    const int rc = upload_impl(c, b);
    if (rc != FSEG_OK && c) set_in_flight(c, false);     // nothing of this batch will run: the device's other contexts may fork again
    return rc;
}
static int upload_impl(fseg_ctx *c, const fseg_batch *b) {
    if (!c || !b) return FSEG_ERR_ARG;
    if (b->n_part <= 0 || !b->part_iv_off || !b->iv_start || !b->iv_end || !b->part_rep_off || !b->rep_weight ||
        !b->rep_exon_off || !b->ex_ts || !b->ex_te)
        return fail(c, FSEG_ERR_ARG, "fseg_upload: null array or empty batch");
    HIP_TRY(c, hipSetDevice(c->device));
    Tick tk;
    // the previous batch's work (and its copy out of the staging image) must be over before its buffers are reused
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->pending = false; c->have_batch = false; c->ran = false; c->fetched = false; c->counts_known = false;
    set_in_flight(c, true);                              // until the run that follows has completed (finish_run)
    const int np = b->n_part;
    const i64 K = b->part_iv_off[np], R = b->part_rep_off[np];
    if (b->part_iv_off[0] != 0 || b->part_rep_off[0] != 0 || b->rep_exon_off[0] != 0)
        return fail(c, FSEG_ERR_ARG, "fseg_upload: offsets must start at 0");
    if (K <= 0 || R < 0) return fail(c, FSEG_ERR_ARG, "fseg_upload: bad offsets");
    const i64 I = b->rep_exon_off[R];
    if (I < 0) return fail(c, FSEG_ERR_ARG, "rep_exon_off not monotone");
    if (I >= 0x7fffffffLL) return fail(c, FSEG_ERR_UNSUPPORTED, "batch has %lld exons; split it (limit 2^31-1 per upload)", (long long)I);
    // ---- host pass 1: the partition / interval level (validation as read_split() asserts it, :138-140) and the counts
    i64 NPOS = 0, n_tiles = 0, lanes = 0, n_rep_blocks = 0, max_part_reps = 0, max_rep_exons = 0;
    bool expanded = false;
    for (int p = 0; p < np; ++p) {
        const i64 k0 = b->part_iv_off[p], k1 = b->part_iv_off[p + 1];
        if (k1 <= k0) return fail(c, FSEG_ERR_INPUT, "partition %d has no intervals", p);
        if (b->part_rep_off[p + 1] < b->part_rep_off[p]) return fail(c, FSEG_ERR_ARG, "part_rep_off not monotone");
        for (i64 k = k0; k < k1; ++k) {
            if (!(b->iv_start[k] < b->iv_end[k])) return fail(c, FSEG_ERR_INPUT, "partition %d: interval with start >= end (py/freddie_segment.py:140)", p);
            if (k > k0 && !(b->iv_end[k - 1] < b->iv_start[k])) return fail(c, FSEG_ERR_INPUT, "partition %d: intervals overlap or are unordered (py/freddie_segment.py:138)", p);
            const i64 len = (i64)b->iv_end[k] - b->iv_start[k] + 1;       // an interval owns positions s..e inclusive (:652-659)
            NPOS += len;
            n_tiles += (len + kSmoothTile - 1) / kSmoothTile;
        }
        n_rep_blocks += (b->part_rep_off[p + 1] - b->part_rep_off[p] + 255) / 256;
        if (b->part_rep_off[p + 1] - b->part_rep_off[p] > max_part_reps) max_part_reps = b->part_rep_off[p + 1] - b->part_rep_off[p];
    }
    if (NPOS >= 0x7fffffffLL) return fail(c, FSEG_ERR_UNSUPPORTED, "batch has %lld positions; split it (limit 2^31-1 per upload)", (long long)NPOS);
    for (i64 r = 0; r < R; ++r) {
        const int w = b->rep_weight[r];
        if (w < 1) return fail(c, FSEG_ERR_INPUT, "rep %lld has weight %d (< 1)", (long long)r, w);
        if (w != 1) expanded = true;
        lanes += w;
        if (b->rep_exon_off[r + 1] < b->rep_exon_off[r]) return fail(c, FSEG_ERR_ARG, "rep_exon_off not monotone");
        if (b->rep_exon_off[r + 1] - b->rep_exon_off[r] > max_rep_exons) max_rep_exons = b->rep_exon_off[r + 1] - b->rep_exon_off[r];
    }
    if (lanes >= 0x7fffffffLL) return fail(c, FSEG_ERR_UNSUPPORTED, "batch has %lld reads; split it (limit 2^31-1 per upload)", (long long)lanes);
    // histogram chunks: consecutive positions of one partition; as large as possible (fewer reads are visited twice)
    // while still giving >= 512 workgroups
    int hist_chunk = kHistChunk;
    while (hist_chunk > 1024 && NPOS / hist_chunk < 512) hist_chunk >>= 1;
    i64 n_chunks = 0;
    {
        i64 k = 0, pos = 0;
        for (int p = 0; p < np; ++p) {
            i64 Pp = 0;
            for (k = b->part_iv_off[p]; k < b->part_iv_off[p + 1]; ++k) Pp += (i64)b->iv_end[k] - b->iv_start[k] + 1;
            n_chunks += (Pp + hist_chunk - 1) / hist_chunk;
            pos += Pp;
        }
        (void)pos;
    }
    const i64 nb = scan_blocks(NPOS);
    if (n_tiles >= 0x7fffffffLL || n_chunks >= 0x7fffffffLL) return fail(c, FSEG_ERR_UNSUPPORTED, "batch too large");
    // ---- the input slab: [uploaded part, mirrored by the pinned staging image][device-derived part]
    Carve in;
    in.add(c->d_part_iv_off, ((size_t)np + 1) * 8);
    in.add(c->d_part_rep_off, ((size_t)np + 1) * 8);
    in.add(c->d_part_lane_off, ((size_t)np + 1) * 8);
    in.add(c->d_iv_start, (size_t)K * 4);
    in.add(c->d_iv_end, (size_t)K * 4);
    in.add(c->d_pos_off, ((size_t)K + 1) * 8);
    in.add(c->d_iv_part, (size_t)K * 4);
    in.add(c->d_iv_tile0, (size_t)K * 4);
    in.add(c->d_tile_desc, (size_t)n_tiles * sizeof(TileDesc));
    in.add(c->d_blk_iv0, ((size_t)nb + 1) * 4);
    in.add(c->d_rb_part, (size_t)n_rep_blocks * 4);
    in.add(c->d_rb_r0, (size_t)n_rep_blocks * 4);
    in.add(c->d_hc_part, (size_t)n_chunks * 4);
    in.add(c->d_hc_p0, (size_t)n_chunks * 8);
    in.add(c->d_hc_n, (size_t)n_chunks * 4);
    in.add(c->d_hc_glo, (size_t)n_chunks * 4);
    in.add(c->d_hc_ghi, (size_t)n_chunks * 4);
    in.add(c->d_rep_exon_off, ((size_t)R + 1) * 8);
    in.add(c->d_rep_weight, (size_t)R * 4);
    in.add(c->d_ex_ts, (size_t)I * 4 + kExonPad);
    in.add(c->d_ex_te, (size_t)I * 4 + kExonPad);
    const size_t up_bytes = in.total;
    in.add(c->d_lane_ex, (size_t)lanes * 16);
    in.add(c->d_lane_start, (size_t)lanes * 4);
    in.add(c->d_lane_pmax, (size_t)lanes * 4);
    in.add(c->d_hc_llo, (size_t)n_chunks * 8);
    in.add(c->d_hc_lhi, (size_t)n_chunks * 8);
    in.add(c->d_key_a, (size_t)R * 8);
    in.add(c->d_key_b, (size_t)R * 8);
    in.add(c->d_val_a, (size_t)R * 4);
    in.add(c->d_val_b, (size_t)R * 4);
    in.add(c->d_rep_last, (size_t)R * 4);
    in.add(c->d_rb_sum, (size_t)n_rep_blocks * 8);
    in.add(c->d_rb_base, (size_t)n_rep_blocks * 8);
    in.add(c->d_rb_max, (size_t)n_rep_blocks * 4);
    in.add(c->d_rb_cmax, (size_t)n_rep_blocks * 4);
    in.add(c->d_rb_esum, (size_t)n_rep_blocks * 8);
    in.add(c->d_rb_ebase, (size_t)n_rep_blocks * 8);
    in.add(c->d_lane_lx, (size_t)lanes * 8 + 64);
    in.add(c->d_lex, (size_t)I * 8 + kLexPad);
    TRY(reserve(c, c->slab_in, in.total));
    in.bind(c->slab_in);
    TRY(reserve_host(c, c->h_stage, up_bytes));
    char *stage = c->h_stage.as<char>();
    auto host_of = [&](const DevBuf &d) { return stage + (static_cast<char *>(d.p) - static_cast<char *>(c->slab_in.p)); };
    const double t_plan = tk.ms();
    // ---- host pass 2: the small derived tables, written straight into the staging image
    {
        i64 *h_pos_off = reinterpret_cast<i64 *>(host_of(c->d_pos_off)), *h_lane_off = reinterpret_cast<i64 *>(host_of(c->d_part_lane_off));
        int *h_iv_part = reinterpret_cast<int *>(host_of(c->d_iv_part)), *h_iv_tile0 = reinterpret_cast<int *>(host_of(c->d_iv_tile0));
        TileDesc *h_tile = reinterpret_cast<TileDesc *>(host_of(c->d_tile_desc));
        int *h_blk = reinterpret_cast<int *>(host_of(c->d_blk_iv0));
        int *h_rb_part = reinterpret_cast<int *>(host_of(c->d_rb_part)), *h_rb_r0 = reinterpret_cast<int *>(host_of(c->d_rb_r0));
        int *h_hc_part = reinterpret_cast<int *>(host_of(c->d_hc_part)), *h_hc_n = reinterpret_cast<int *>(host_of(c->d_hc_n));
        int *h_hc_glo = reinterpret_cast<int *>(host_of(c->d_hc_glo)), *h_hc_ghi = reinterpret_cast<int *>(host_of(c->d_hc_ghi));
        i64 *h_hc_p0 = reinterpret_cast<i64 *>(host_of(c->d_hc_p0));
        h_pos_off[0] = 0; h_lane_off[0] = 0; c->max_part_lanes = 0; c->max_part_pos = 0;
        i64 t = 0, rb = 0, ch = 0, l = 0, bq = 0;
        for (int p = 0; p < np; ++p) {
            const i64 k0 = b->part_iv_off[p], k1 = b->part_iv_off[p + 1];
            for (i64 k = k0; k < k1; ++k) {
                const i64 len = (i64)b->iv_end[k] - b->iv_start[k] + 1;
                h_iv_part[k] = p;
                h_pos_off[k + 1] = h_pos_off[k] + len;
                h_iv_tile0[k] = (int)t;
                for (i64 y = 0; y < len; y += kSmoothTile) h_tile[t++] = TileDesc{h_pos_off[k], (int)y, (int)len};
                // interval of the first position of every scan block
                for (; bq < nb && bq * kScanBlock < h_pos_off[k + 1]; ++bq) h_blk[bq] = (int)k;
            }
            for (i64 r = b->part_rep_off[p]; r < b->part_rep_off[p + 1]; r += 256) { h_rb_part[rb] = p; h_rb_r0[rb] = (int)r; ++rb; }
            for (i64 r = b->part_rep_off[p]; r < b->part_rep_off[p + 1]; ++r) l += b->rep_weight[r];
            h_lane_off[p + 1] = l;
            if (l - h_lane_off[p] > c->max_part_lanes) c->max_part_lanes = l - h_lane_off[p];
            // histogram chunks of the partition, with the genomic position of the chunk's first and last position (a
            // chunk may span several intervals of its partition)
            const i64 P0 = h_pos_off[k0], P1 = h_pos_off[k1];
            if (P1 - P0 > c->max_part_pos) c->max_part_pos = P1 - P0;
            i64 k = k0;
            for (i64 q0 = P0; q0 < P1; q0 += hist_chunk) {
                const i64 q1 = std::min<i64>(q0 + hist_chunk, P1) - 1;
                while (h_pos_off[k + 1] <= q0) ++k;
                i64 kk = k;
                while (h_pos_off[kk + 1] <= q1) ++kk;
                h_hc_part[ch] = p; h_hc_p0[ch] = q0; h_hc_n[ch] = (int)(q1 - q0 + 1);
                h_hc_glo[ch] = b->iv_start[k] + (int)(q0 - h_pos_off[k]);
                h_hc_ghi[ch] = b->iv_start[kk] + (int)(q1 - h_pos_off[kk]);
                ++ch;
            }
        }
        h_blk[nb] = (int)(K - 1);
    }
    const double t_tables = tk.ms();
    // ---- the caller's arrays
    memcpy(host_of(c->d_part_iv_off), b->part_iv_off, ((size_t)np + 1) * 8);
    memcpy(host_of(c->d_part_rep_off), b->part_rep_off, ((size_t)np + 1) * 8);
    memcpy(host_of(c->d_iv_start), b->iv_start, (size_t)K * 4);
    memcpy(host_of(c->d_iv_end), b->iv_end, (size_t)K * 4);
    memcpy(host_of(c->d_rep_exon_off), b->rep_exon_off, ((size_t)R + 1) * 8);
    memcpy(host_of(c->d_rep_weight), b->rep_weight, (size_t)R * 4);
    memcpy(host_of(c->d_ex_ts), b->ex_ts, (size_t)I * 4);
    memcpy(host_of(c->d_ex_te), b->ex_te, (size_t)I * 4);
    const double t_copy = tk.ms();
    c->n_part = np; c->K = K; c->R = R; c->I = I; c->NPOS = NPOS; c->LANES = lanes; c->expanded = expanded;
    if (c->max_part_lanes > lanes) c->max_part_lanes = lanes;
    c->n_tiles = (int)n_tiles; c->n_hist_chunks = (int)n_chunks; c->n_rep_blocks = (int)n_rep_blocks; c->max_rep_exons = max_rep_exons;
    c->part_iv_off.assign(b->part_iv_off, b->part_iv_off + np + 1);
    c->part_rep_off.assign(b->part_rep_off, b->part_rep_off + np + 1);
    hipStream_t s = c->stream;
    HIP_TRY(c, hipMemcpyAsync(c->slab_in.p, stage, up_bytes, hipMemcpyHostToDevice, s));
    // ---- device-side preparation
    {
        PrepStatus init;
        init.err = 0; init.pad = 0;
        for (int q = 0; q < 4; ++q) init.bad_rep[q] = 0x7fffffffffffffffLL;
        *c->h_prep = init;
        HIP_TRY(c, hipMemcpyAsync(c->d_prep.p, c->h_prep, sizeof(PrepStatus), hipMemcpyHostToDevice, s));
        c->prep_checked = false;
    }
    if (R > 0) {
        hipLaunchKernelGGL(k_prep_reps, dim3(grid_for(n_rep_blocks, 1, 65536)), dim3(256), 0, s, (int)n_rep_blocks, c->d_rb_part.as<int>(),
                           c->d_rb_r0.as<int>(), c->d_part_rep_off.as<i64>(), c->d_part_iv_off.as<i64>(), c->d_iv_start.as<int>(),
                           c->d_iv_end.as<int>(), c->d_rep_exon_off.as<i64>(), c->d_ex_ts.as<int>(), c->d_ex_te.as<int>(),
                           c->d_key_a.as<u64>(), c->d_val_a.as<int>(), c->d_rep_last.as<int>(), c->d_prep.as<PrepStatus>());
        // small partitions (the usual case) sort themselves inside k_lanes; a batch with a large one goes through the
        // batch-wide radix sort (FSEG_GLOBAL_SORT=1 forces it: tests)
        const bool sort_here = max_part_reps <= kLaneSortMax && !c->force_global_sort;
        if (!sort_here) {
            unsigned end_bit = 33;
            while (end_bit < 64 && ((u64)np >> (end_bit - 32)) != 0) ++end_bit;
            size_t tmp_bytes = 0;
            HIP_TRY(c, fseg_sort_pairs(nullptr, &tmp_bytes, c->d_key_a.as<u64>(), c->d_key_b.as<u64>(), c->d_val_a.as<int>(), c->d_val_b.as<int>(),
                                       (size_t)R, end_bit, s));
            TRY(ensure(c, c->d_sort_tmp, tmp_bytes));
            HIP_TRY(c, fseg_sort_pairs(c->d_sort_tmp.p, &tmp_bytes, c->d_key_a.as<u64>(), c->d_key_b.as<u64>(), c->d_val_a.as<int>(),
                                       c->d_val_b.as<int>(), (size_t)R, end_bit, s));
        }
        if (sort_here) {
            hipLaunchKernelGGL(k_lanes, dim3(grid_for(np, 1, 65536)), dim3(256), 0, s, np, c->d_part_rep_off.as<i64>(), c->d_part_lane_off.as<i64>(),
                               c->d_key_b.as<u64>(), c->d_val_b.as<int>(), c->d_rep_weight.as<int>(), c->d_rep_last.as<int>(),
                               c->d_rep_exon_off.as<i64>(), c->d_lane_ex.as<longlong2>(), c->d_lane_start.as<int>(), c->d_lane_pmax.as<int>(),
                               1, c->d_key_a.as<u64>(), c->d_ex_ts.as<int>(), c->d_ex_te.as<int>(), c->d_lane_lx.as<int2>(), c->d_lex.as<int2>());
        } else {
            const int rbg = grid_for(n_rep_blocks, 1, 65536);
            hipLaunchKernelGGL(k_lane_blocks, dim3(rbg), dim3(256), 0, s, (int)n_rep_blocks, c->d_rb_part.as<int>(), c->d_rb_r0.as<int>(),
                               c->d_part_rep_off.as<i64>(), c->d_val_b.as<int>(), c->d_rep_weight.as<int>(), c->d_rep_last.as<int>(),
                               c->d_rb_sum.as<i64>(), c->d_rb_max.as<int>(), c->d_rep_exon_off.as<i64>(), c->d_rb_esum.as<i64>());
            hipLaunchKernelGGL(k_lane_block_scan, dim3(grid_for(np, 4, 4096)), dim3(256), 0, s, np, (int)n_rep_blocks, c->d_rb_part.as<int>(),
                               c->d_part_lane_off.as<i64>(), c->d_rb_sum.as<i64>(), c->d_rb_max.as<int>(), c->d_rb_base.as<i64>(),
                               c->d_rb_cmax.as<int>(), c->d_part_rep_off.as<i64>(), c->d_rep_exon_off.as<i64>(), c->d_rb_esum.as<i64>(),
                               c->d_rb_ebase.as<i64>());
            hipLaunchKernelGGL(k_lane_emit, dim3(rbg), dim3(256), 0, s, (int)n_rep_blocks, c->d_rb_part.as<int>(), c->d_rb_r0.as<int>(),
                               c->d_part_rep_off.as<i64>(), c->d_key_b.as<u64>(), c->d_val_b.as<int>(), c->d_rep_weight.as<int>(),
                               c->d_rep_last.as<int>(), c->d_rep_exon_off.as<i64>(), c->d_rb_base.as<i64>(), c->d_rb_cmax.as<int>(),
                               c->d_lane_ex.as<longlong2>(), c->d_lane_start.as<int>(), c->d_lane_pmax.as<int>(), c->d_rb_ebase.as<i64>(),
                               c->d_ex_ts.as<int>(), c->d_ex_te.as<int>(), c->d_lane_lx.as<int2>(), c->d_lex.as<int2>());
        }
    }
    hipLaunchKernelGGL(k_hist_ranges, dim3(grid_for(n_chunks, 256, 4096)), dim3(256), 0, s, (int)n_chunks, c->d_hc_part.as<int>(),
                       c->d_hc_glo.as<int>(), c->d_hc_ghi.as<int>(), c->d_part_lane_off.as<i64>(), c->d_lane_start.as<int>(),
                       c->d_lane_pmax.as<int>(), c->d_hc_llo.as<i64>(), c->d_hc_lhi.as<i64>());
    HIP_TRY(c, hipMemcpyAsync(c->h_prep, c->d_prep.p, sizeof(PrepStatus), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipGetLastError());
    // ---- position- and candidate-sized work buffers (candidates and finals are distinct positions, so NPOS bounds them)
    {
        const size_t np8 = (size_t)NPOS + 64;
        auto atleast = [](i64 &cap, i64 v) { if (cap < v) cap = v; };
        atleast(c->chunk_cap, NPOS / 8192 + np + 8);              // an upper bound, not a guess
        Carve cv;
        cv.add(c->d_y_raw, np8 * 4); cv.add(c->d_y, np8 * 8); cv.add(c->d_bits, 3 * flag_words(np8) * 4);
        cv.add(c->d_v, np8 * 8);
        cv.add(c->d_scan_state, ((size_t)nb * 3 + 1) * 8);
        cv.add(c->d_bsum, ((size_t)nb + 2) * 4);
        cv.add(c->d_bsum_side, ((size_t)nb + 2) * 4);
        cv.add(c->d_g, np8 * 8); cv.add(c->d_pk, np8 * 4); cv.add(c->d_pf, np8); cv.add(c->d_kp, np8);
        cv.add(c->d_part_has2, ((size_t)np + 1) * 4);
        cv.add(c->d_tile_tot, ((size_t)n_tiles + 1) * 4);
        cv.add(c->d_blk_pre, ((size_t)n_tiles + 1) * (kSmoothTile / kSumBlock) * 4);
        cv.add(c->d_tile_defer, ((size_t)n_tiles + 1) * 4);
        cv.add(c->d_voff, ((size_t)np + 2) * 8); cv.add(c->d_chunk_off, ((size_t)np + 2) * 8);
        cv.add(c->d_mean, ((size_t)np + 1) * 8); cv.add(c->d_thr, ((size_t)np + 1) * 8);
        cv.add(c->d_label_off, ((size_t)np + 2) * 8);
        cv.add(c->d_csum, (size_t)c->chunk_cap * 16);     // chunk sums of both passes
        cv.add(c->d_cand_off, ((size_t)K + 2) * 8); cv.add(c->d_final_off, ((size_t)K + 2) * 8);
        cv.add(c->d_cand_y, np8 * 4); cv.add(c->d_fixed0, np8); cv.add(c->d_added, np8);
        cv.add(c->d_fixed, np8); cv.add(c->d_chosen, np8);
        cv.add(c->d_final_y, np8 * 4); cv.add(c->d_final_pos, np8 * 4); cv.add(c->d_final_iv, np8 * 4); cv.add(c->d_col_thr, np8 * 8); cv.add(c->d_col_zero, np8);
        cv.add(c->d_seg_iv, np8 * 4); cv.add(c->d_seg_prev, np8 * 4); cv.add(c->d_rseg_c, np8 * 4);
        cv.add(c->d_cand_pn, np8 * 4); cv.add(c->d_cand_ll, np8 * 4); cv.add(c->d_cand_ln, np8 * 4); cv.add(c->d_cand_wide, np8);
        cv.add(c->d_prob_bs, ((size_t)NPOS / kProbBlock + 2) * kProbCols * 8);
        TRY(reserve(c, c->slab_pos, cv.total));
        cv.bind(c->slab_pos);
        // arena capacities: a sized first run makes them exact; without it (FSEG_NO_SIZED) these are first guesses that
        // finish_run() grows
        atleast(c->prob_cap, 1024); atleast(c->work_cap, 1024); atleast(c->pair_cap, 1 << 16); atleast(c->tri_cap, 1 << 18);
        atleast(c->label_cap, 1 << 16); atleast(c->cov_cap, 1 << 18);
        TRY(alloc_arenas(c));
    }
    drop_graph(c);
    c->have_batch = true;
    if (c->trace)
        fprintf(stderr, "[fseg] upload: plan %.3f ms, tables %.3f ms, copy-in %.3f ms (%.1f MB), enqueue %.3f ms; %lld positions, %lld reps, %lld exons\n",
                t_plan, t_tables, t_copy, up_bytes / 1e6, tk.ms(), (long long)NPOS, (long long)R, (long long)I);
    return FSEG_OK;
}

static int run_impl(fseg_ctx *c);
int fseg_run(fseg_ctx *c) {
    const int rc = run_impl(c);
    if (rc != FSEG_OK && c && !c->pending) set_in_flight(c, false);
    return rc;
}
static int run_impl(fseg_ctx *c) {
    if (!c) return FSEG_ERR_ARG;
    if (!c->have_params || !c->have_batch) return fail(c, FSEG_ERR_ARG, "fseg_run: set parameters and upload a batch first");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->pending) TRY(finish_run(c));
    c->fetched = false;
    c->run_events_only = false;
    set_in_flight(c, true);
    claim_device(c);
    if (c->owns_device) ++c->forked_runs;
    if (c->use_fork && !c->side[0] && c->owns_device) {
        hipStream_t made[fseg_ctx::kSide] = {};
        hipError_t e = hipSuccess;
        for (int i = 0; e == hipSuccess && i < fseg_ctx::kSide; ++i) e = hipStreamCreateWithFlags(&made[i], hipStreamNonBlocking);
        if (e != hipSuccess) {
            for (hipStream_t m : made) if (m) (void)hipStreamDestroy(m);
            return fail(c, FSEG_ERR_HIP, "hipStreamCreateWithFlags: %s", hipGetErrorString(e));
        }
        for (int i = 0; i < fseg_ctx::kSide; ++i) c->side[i] = made[i];
    }
    if (c->owns_device && c->dev_sync && !c->side_probed) TRY(probe_side_queues(c));
    // first run of a batch: piecewise with exact arena sizes; afterwards the sizes are known and the same launch
    // sequence is replayed (as a hipGraph unless disabled)
    if (!c->ran && c->use_sized) return run_sized(c);
    if (c->label_cap > 0) {
        // (the label arena of THIS run's form: a context whose parameters moved between threshold_rate < 1 -- two bits per label --
        // and threshold_rate = 1 -- bytes -- has so far only the other one; nothing to do otherwise)
        void *before[2] = {c->d_labels.p, c->d_packed.p};
        TRY(ensure_label_arena(c));
        if (before[0] != c->d_labels.p || before[1] != c->d_packed.p) drop_graph(c);
    }
    c->last_sized = false;
    c->run_plain = !c->use_graph || c->profile_plain || would_fork(c);
    if (!c->run_plain) {
        c->run_linear = true;
        struct Unset { fseg_ctx *c; ~Unset() { c->run_linear = false; } } unset{c};
        if (c->n_graphs == 0) {
            const int want = c->profiling ? 2 : 1;
            bool ok = true;
            for (int g = 0; g < want && ok; ++g) {
                HIP_TRY(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
                int rc = enqueue_run(c, want == 1 ? SEG_ALL : (g == 0 ? (SEG_PRE1 | SEG_PRE2) : (SEG_POST1 | SEG_POST2 | SEG_STATUS)));
                hipError_t e = hipStreamEndCapture(c->stream, &c->graph[g]);
                if (rc == FSEG_OK && e == hipSuccess) e = hipGraphInstantiate(&c->graph_exec[g], c->graph[g], nullptr, nullptr, 0);
                ok = rc == FSEG_OK && e == hipSuccess;
            }
            if (ok) c->n_graphs = want;
            else {                                         // capture not possible: fall back to plain launches
                (void)hipGetLastError();
                drop_graph(c);
                c->use_graph = false;
                c->run_plain = true;
            }
        }
        if (c->n_graphs == 1) {
            HIP_TRY(c, hipGraphLaunch(c->graph_exec[0], c->stream));
            c->pending = true;
            return FSEG_OK;
        }
        if (c->n_graphs == 2) {
            HIP_TRY(c, hipEventRecord(c->ev_g[0], c->stream));
            HIP_TRY(c, hipGraphLaunch(c->graph_exec[0], c->stream));
            HIP_TRY(c, hipEventRecord(c->ev_g[1], c->stream));
            TRY(enqueue_run(c, SEG_SCORE));
            HIP_TRY(c, hipEventRecord(c->ev_g[2], c->stream));
            HIP_TRY(c, hipGraphLaunch(c->graph_exec[1], c->stream));
            HIP_TRY(c, hipEventRecord(c->ev_g[3], c->stream));
            c->pending = true;
            return FSEG_OK;
        }
    }
    TRY(enqueue_run(c, SEG_ALL));
    c->pending = true;
    return FSEG_OK;
}

int fseg_sync(fseg_ctx *c) {
    if (!c) return FSEG_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->pending) return finish_run(c);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->have_batch) TRY(check_prep(c));
    return FSEG_OK;
}

int fseg_get_sizes(fseg_ctx *c, fseg_sizes *out) {
    if (!c || !out) return FSEG_ERR_ARG;
    TRY(fseg_sync(c));
    if (!c->ran) return fail(c, FSEG_ERR_ARG, "no completed run");
    out->n_final = (int64_t)c->h_status->n_final;
    out->label_bytes = (int64_t)c->h_status->label_bytes;
    out->n_cand = (int64_t)c->h_status->n_cand;
    out->n_problems = (int64_t)c->h_status->n_prob;
    out->n_positions = c->NPOS;
    out->max_problem_size = (int64_t)c->h_status->max_n;
    out->max_problem_reads = (int64_t)c->h_status->max_ln;
    return FSEG_OK;
}

// Results of the last run in the context's pinned host buffers (one device-to-host copy each, no pageable staging):
// valid until the next fseg_results / fseg_results_packed on this context (nothing else writes them: see include/freddie_seg.h).
// Large device-to-host copies go to an SDMA engine through the HSA runtime.  hipMemcpyAsync performs a large copy to pinned
// host memory with a copy KERNEL (256 workgroups that wait on PCIe): it holds its hardware queue for the 340 us the copy
// takes, and the kernels of the contexts that share the queue behind it (six contexts taking turns: 253 -> 280 M reads/s
// with the copy on SDMA).  The HSA agent of a HIP device is found by its PCI address; if anything of this fails the copy
// falls back to hipMemcpyAsync (FSEG_NO_SDMA_D2H=1 forces that).
struct HsaAgents { hsa_agent_t gpu[64]; unsigned bdf[64]; unsigned dom[64]; int n_gpu = 0; hsa_agent_t cpu[16]; int n_cpu = 0; bool ok = false; };
static hsa_status_t hsa_agent_cb(hsa_agent_t a, void *data) {
    HsaAgents *h = static_cast<HsaAgents *>(data);
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    if (t == HSA_DEVICE_TYPE_GPU) {
        if (h->n_gpu < 64) {
            unsigned bdf = 0, dom = 0;
            (void)hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf);
            (void)hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &dom);
            h->gpu[h->n_gpu] = a; h->bdf[h->n_gpu] = bdf & 0xffffu; h->dom[h->n_gpu] = dom; ++h->n_gpu;
        }
    } else if (t == HSA_DEVICE_TYPE_CPU) { if (h->n_cpu < 16) h->cpu[h->n_cpu++] = a; }
    return HSA_STATUS_SUCCESS;
}
static HsaAgents &hsa_agents() {
    static HsaAgents h;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *v = getenv("FSEG_NO_SDMA_D2H");
        if (v && v[0] == '1') return;
        if (hsa_init() == HSA_STATUS_SUCCESS && hsa_iterate_agents(hsa_agent_cb, &h) == HSA_STATUS_SUCCESS) h.ok = h.n_gpu > 0 && h.n_cpu > 0;
    });
    return h;
}
// the copy is issued when the stream's work is over (the caller has waited for it) and waited for here
static bool sdma_d2h(fseg_ctx *c, void *dst_pinned, const void *src_dev, size_t bytes) {
    HsaAgents &h = hsa_agents();
    if (!h.ok) return false;
    if (c->hsa_agent < 0) {                              // once per context: the agent with the device's PCI address
        c->hsa_agent = -2;
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, c->device) == hipSuccess) {
            const unsigned bdf = ((unsigned)pr.pciBusID << 8) | ((unsigned)pr.pciDeviceID << 3);
            for (int i = 0; i < h.n_gpu; ++i)
                if ((h.bdf[i] & 0xfff8u) == bdf && h.dom[i] == (unsigned)pr.pciDomainID) c->hsa_agent = i;
        }
        if (c->hsa_agent >= 0 && hsa_signal_create(1, 0, nullptr, &c->hsa_sig) != HSA_STATUS_SUCCESS) c->hsa_agent = -2;
    }
    if (c->hsa_agent < 0) return false;
    // the destination's agent: the CPU (NUMA node) that owns the pinned buffer, asked of the runtime whenever the buffer has
    // been reallocated; the first CPU agent if it will not say
    if (c->hsa_dst_base != c->h_res.p) {
        c->hsa_dst_base = c->h_res.p;
        c->hsa_cpu = 0;
        hsa_amd_pointer_info_t info;
        memset(&info, 0, sizeof info);
        info.size = sizeof info;
        if (hsa_amd_pointer_info(dst_pinned, &info, nullptr, nullptr, nullptr) == HSA_STATUS_SUCCESS)
            for (int i = 0; i < h.n_cpu; ++i) if (h.cpu[i].handle == info.agentOwner.handle) c->hsa_cpu = i;
    }
    hsa_signal_store_relaxed(c->hsa_sig, 1);
    if (hsa_amd_memory_async_copy(dst_pinned, h.cpu[c->hsa_cpu], src_dev, h.gpu[c->hsa_agent], bytes, 0, nullptr, c->hsa_sig) != HSA_STATUS_SUCCESS) return false;
    // a copy that completes takes the signal from 1 to 0; one that FAILS drives it negative, which also ends the wait:
    // then nothing can be said about the destination -- this context stays off the engine and the caller copies again
    const hsa_signal_value_t v = hsa_signal_wait_scacquire(c->hsa_sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
    if (v != 0) {
        c->hsa_agent = -2;
        if (c->trace) fprintf(stderr, "[fseg] SDMA result copy failed (signal %lld): falling back to the runtime's copy\n", (long long)v);
        return false;
    }
    return true;
}

// the byte form of labels that the run wrote at two bits each (fseg_results / fseg_download; once per run)
static int unpack_labels(fseg_ctx *c, size_t lb) {
    if (!c->run_label_packed || c->labels_unpacked || lb == 0) return FSEG_OK;
    TRY(ensure(c, c->d_labels, lb + 32));
    const i64 n16 = (i64)((lb + 15) / 16);
    hipLaunchKernelGGL(k_unpack_labels, dim3(grid_for(n16, 256 * 4, 2048)), dim3(256), 0, c->stream, c->d_packed.as<unsigned>(), c->d_labels.as<uint4>(), n16);
    c->labels_unpacked = true;
    return FSEG_OK;
}

static int results_impl(fseg_ctx *c, const int64_t **part_final_off, const int32_t **final_pos, const int64_t **label_off,
                        const uint8_t **labels, bool packed) {
    if (!c) return FSEG_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->pending && !c->ran) return fail(c, FSEG_ERR_ARG, "no completed run");
    hipStream_t s = c->stream;
    if (!c->fetched || c->fetched_packed != (packed ? 1 : 0)) {
        // a sized run knows its result sizes before its last kernels have finished: the copies queue up right behind them
        if (!(c->pending && c->last_sized)) TRY(fseg_sync(c));
        if (!c->pending && !c->ran) return fail(c, FSEG_ERR_ARG, "no completed run");
        const size_t nf = (size_t)c->h_status->n_final, lb = (size_t)c->h_status->label_bytes;
        size_t off = 0;
        auto take = [&](int i, size_t bytes) { c->res_off[i] = off; off = (off + bytes + 255) & ~(size_t)255; };
        take(0, ((size_t)c->K + 1) * 8); take(1, nf * 4); take(2, ((size_t)c->n_part + 1) * 8); take(3, packed ? (lb + 15) / 16 * 4 : lb);
        TRY(reserve_host(c, c->h_res, off));
        char *h = c->h_res.as<char>();
        HIP_TRY(c, hipMemcpyAsync(h + c->res_off[0], c->d_final_off.p, ((size_t)c->K + 1) * 8, hipMemcpyDeviceToHost, s));
        if (nf) HIP_TRY(c, hipMemcpyAsync(h + c->res_off[1], c->d_final_pos.p, nf * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipMemcpyAsync(h + c->res_off[2], c->d_label_off.p, ((size_t)c->n_part + 1) * 8, hipMemcpyDeviceToHost, s));
        const void *big_src = nullptr;                       // the label matrix: on SDMA when it is large (see sdma_d2h)
        size_t big_bytes = 0;
        if (lb && !packed) { TRY(unpack_labels(c, lb)); big_src = c->d_labels.p; big_bytes = lb; }
        if (lb && packed) {
            // (the label arena is allocated with 16 spare bytes: the last, partial group of 16 labels is read whole)
            const i64 n16 = (i64)((lb + 15) / 16);
            if (!c->run_label_packed) {                      // the run wrote bytes (threshold_rate = 1, FSEG_LABEL_BYTES=1): pack them
                TRY(ensure(c, c->d_packed, (size_t)n16 * 4));
                hipLaunchKernelGGL(k_pack_labels, dim3(grid_for(n16, 256 * 4, 2048)), dim3(256), 0, s, c->d_labels.as<uint4>(),
                                   c->d_packed.as<unsigned>(), n16);
            }
            big_src = c->d_packed.p; big_bytes = (size_t)n16 * 4;
        }
        if (big_bytes >= (1u << 20) && hsa_agents().ok && c->hsa_agent != -2) {
            if (c->pending) TRY(finish_run(c));
            else HIP_TRY(c, hipStreamSynchronize(s));
            if (!c->ran) return fail(c, FSEG_ERR_ARG, "no completed run");
            if (sdma_d2h(c, h + c->res_off[3], big_src, big_bytes)) big_bytes = 0;
        }
        if (big_bytes) HIP_TRY(c, hipMemcpyAsync(h + c->res_off[3], big_src, big_bytes, hipMemcpyDeviceToHost, s));
        if (c->pending) TRY(finish_run(c));
        else HIP_TRY(c, hipStreamSynchronize(s));
        if (!c->ran) return fail(c, FSEG_ERR_ARG, "no completed run");
        const i64 *fo = reinterpret_cast<const i64 *>(h + c->res_off[0]);
        c->res_pfo.resize((size_t)c->n_part + 1);
        for (int p = 0; p <= c->n_part; ++p) c->res_pfo[(size_t)p] = fo[(size_t)c->part_iv_off[(size_t)p]];
        c->fetched = true;
        c->fetched_packed = packed ? 1 : 0;
    }
    const char *h = c->h_res.as<char>();
    if (part_final_off) *part_final_off = reinterpret_cast<const int64_t *>(c->res_pfo.data());
    if (final_pos) *final_pos = reinterpret_cast<const int32_t *>(h + c->res_off[1]);
    if (label_off) *label_off = reinterpret_cast<const int64_t *>(h + c->res_off[2]);
    if (labels) *labels = reinterpret_cast<const uint8_t *>(h + c->res_off[3]);
    return FSEG_OK;
}

int fseg_results(fseg_ctx *c, const int64_t **part_final_off, const int32_t **final_pos, const int64_t **label_off,
                 const uint8_t **labels) {
    return results_impl(c, part_final_off, final_pos, label_off, labels, false);
}
int fseg_results_packed(fseg_ctx *c, const int64_t **part_final_off, const int32_t **final_pos, const int64_t **label_off,
                        const uint8_t **labels2) {
    return results_impl(c, part_final_off, final_pos, label_off, labels2, true);
}

int fseg_download(fseg_ctx *c, int64_t *part_final_off, int32_t *final_pos, int64_t *label_off, uint8_t *labels) {
    if (!c) return FSEG_ERR_ARG;
    TRY(fseg_sync(c));
    if (!c->ran) return fail(c, FSEG_ERR_ARG, "no completed run");
    hipStream_t s = c->stream;
    if (part_final_off) {
        std::vector<i64> fo((size_t)c->K + 1);
        HIP_TRY(c, hipMemcpyAsync(fo.data(), c->d_final_off.p, fo.size() * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        for (int p = 0; p <= c->n_part; ++p) part_final_off[p] = fo[(size_t)c->part_iv_off[p]];
    }
    if (final_pos) HIP_TRY(c, hipMemcpyAsync(final_pos, c->d_final_pos.p, (size_t)c->h_status->n_final * 4, hipMemcpyDeviceToHost, s));
    if (label_off) HIP_TRY(c, hipMemcpyAsync(label_off, c->d_label_off.p, ((size_t)c->n_part + 1) * 8, hipMemcpyDeviceToHost, s));
    if (labels && c->h_status->label_bytes) {
        TRY(unpack_labels(c, (size_t)c->h_status->label_bytes));
        HIP_TRY(c, hipMemcpyAsync(labels, c->d_labels.p, (size_t)c->h_status->label_bytes, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(c, hipStreamSynchronize(s));
    return FSEG_OK;
}

int fseg_tap(fseg_ctx *c, int what, void *dst, int64_t cap_bytes, int64_t *n_bytes) {
    if (!c || !n_bytes) return FSEG_ERR_ARG;
    TRY(fseg_sync(c));
    if (!c->ran) return fail(c, FSEG_ERR_ARG, "no completed run");
    const Status &st = *c->h_status;
    const void *src = nullptr;
    i64 bytes = 0;
    std::vector<int> packed;
    switch (what) {
        case FSEG_TAP_POS_OFF: src = c->d_pos_off.p; bytes = (c->K + 1) * 8; break;
        case FSEG_TAP_Y_RAW: src = c->d_y_raw.p; bytes = c->NPOS * 4; break;
        case FSEG_TAP_Y: src = c->d_y.p; bytes = c->NPOS * 8; break;
        case FSEG_TAP_THRESHOLD: src = c->d_thr.p; bytes = (i64)c->n_part * 8; break;
        case FSEG_TAP_CAND_OFF: src = c->d_cand_off.p; bytes = (c->K + 1) * 8; break;
        case FSEG_TAP_CAND_Y: src = c->d_cand_y.p; bytes = (i64)st.n_cand * 4; break;
        case FSEG_TAP_FIXED: src = c->d_fixed.p; bytes = (i64)st.n_cand; break;
        case FSEG_TAP_CHOSEN: src = c->d_chosen.p; bytes = (i64)st.n_cand; break;
        case FSEG_TAP_FINAL_OFF: src = c->d_final_off.p; bytes = (c->K + 1) * 8; break;
        case FSEG_TAP_FINAL_Y: src = c->d_final_y.p; bytes = (i64)st.n_final * 4; break;
        case FSEG_TAP_PROBLEMS: {
            size_t n = (size_t)st.n_prob;
            std::vector<int> iv(n), sa(n), nn(n), ch(n);
            if (n) {
                HIP_TRY(c, copy_sync(c, iv.data(), c->d_prob_iv.p, n * 4, hipMemcpyDeviceToHost));
                HIP_TRY(c, copy_sync(c, sa.data(), c->d_prob_start.p, n * 4, hipMemcpyDeviceToHost));
                HIP_TRY(c, copy_sync(c, nn.data(), c->d_prob_n.p, n * 4, hipMemcpyDeviceToHost));
                HIP_TRY(c, copy_sync(c, ch.data(), c->d_prob_chain.p, n * 4, hipMemcpyDeviceToHost));
            }
            packed.resize(n * 4);
            for (size_t i = 0; i < n; ++i) { packed[4 * i] = iv[i]; packed[4 * i + 1] = sa[i]; packed[4 * i + 2] = nn[i]; packed[4 * i + 3] = ch[i]; }
            bytes = (i64)n * 16;
            *n_bytes = bytes;
            if (dst && cap_bytes > 0) memcpy(dst, packed.data(), (size_t)(bytes < cap_bytes ? bytes : cap_bytes));
            return FSEG_OK;
        }
        case FSEG_TAP_LANE_START: src = c->d_lane_start.p; bytes = c->LANES * 4; break;
        case FSEG_TAP_LANE_PMAX: src = c->d_lane_pmax.p; bytes = c->LANES * 4; break;
        case FSEG_TAP_LANE_EXONS: src = c->d_lane_ex.p; bytes = c->LANES * 16; break;
        case FSEG_TAP_LANE_STREAM: src = c->d_lane_lx.p; bytes = c->LANES * 8; break;
        case FSEG_TAP_EXON_STREAM: src = c->d_lex.p; bytes = c->I * 8; break;
        case FSEG_TAP_SYNC: {
            SyncWords w{};
            HIP_TRY(c, copy_sync(c, &w, c->d_sync.p, sizeof w, hipMemcpyDeviceToHost));
            packed = {(int)c->sync_gen, c->dev_sync ? 1 : 0, (int)w.emit_gen, (int)w.side_gen[0], (int)w.side_gen[1], (int)w.emit_ctr,
                      (int)c->sync_timeouts, (int)c->forked_runs, (c->side_probed ? 8 : 0) | (c->side_ok[0] ? 1 : 0) | (c->side_ok[1] ? 2 : 0) | (c->side_ok[2] ? 4 : 0)};
            bytes = (i64)packed.size() * 4;
            *n_bytes = bytes;
            if (dst && cap_bytes > 0) memcpy(dst, packed.data(), (size_t)(bytes < cap_bytes ? bytes : cap_bytes));
            return FSEG_OK;
        }
        default: return fail(c, FSEG_ERR_ARG, "unknown tap %d", what);
    }
    *n_bytes = bytes;
    if (dst && cap_bytes > 0 && bytes > 0)
        HIP_TRY(c, copy_sync(c, dst, src, (size_t)(bytes < cap_bytes ? bytes : cap_bytes), hipMemcpyDeviceToHost));
    return FSEG_OK;
}

int fseg_set_profiling(fseg_ctx *c, int on) {
    if (!c) return FSEG_ERR_ARG;
    if (c->profiling != (on != 0) || c->profile_all != (on != 2) || c->profile_plain != (on == 3)) drop_graph(c);
    c->profiling = on != 0;
    c->profile_all = on != 2;
    c->profile_plain = on == 3;         // every stage bracketed on replays too: plain launches instead of the graph
    return FSEG_OK;
}
int fseg_n_stages(void) { return ST_REPORTED; }
const char *fseg_stage_name(int i) { return (i >= 0 && i < ST_REPORTED) ? kStageNames[i] : ""; }
int fseg_stage_ms(fseg_ctx *c, float *ms) {
    if (!c || !ms) return FSEG_ERR_ARG;
    TRY(fseg_sync(c));
    for (int i = 0; i < ST_REPORTED; ++i) ms[i] = c->stage_ms[i];
    return FSEG_OK;
}

#ifdef FSEG_SCORE_TIMING
int fseg_debug_score_timing(fseg_ctx *c, unsigned long long *out8) {
    if (!c || !out8 || !c->d_tacc.p) return FSEG_ERR_ARG;
    if (copy_sync(c, out8, c->d_tacc.p, 128, hipMemcpyDeviceToHost) != hipSuccess) return FSEG_ERR_HIP;   /* 16 slots */
    (void)hipMemsetAsync(c->d_tacc.p, 0, 120, c->stream); (void)hipStreamSynchronize(c->stream);   /* slot 15 = the k_solve class being timed: kept */
    return FSEG_OK;
}
int fseg_debug_prob_ticks(fseg_ctx *c, unsigned long long *out4, long long n_prob) {     /* 4 values per problem */
    if (!c || !out4 || n_prob < 0 || (size_t)n_prob > kTaccProbs) return FSEG_ERR_ARG;
    if (copy_sync(c, out4, static_cast<char *>(c->d_tacc.p) + 128, (size_t)n_prob * 32, hipMemcpyDeviceToHost) != hipSuccess) return FSEG_ERR_HIP;
    return FSEG_OK;
}
int fseg_debug_dp_ticks(fseg_ctx *c, unsigned long long *out4, long long n_prob) {       /* k_dpw's records: (ticks, -, -, start tick) */
    if (!c || !out4 || n_prob < 0 || (size_t)n_prob > kTaccProbs) return FSEG_ERR_ARG;
    if (copy_sync(c, out4, static_cast<char *>(c->d_tacc.p) + 128 + kTaccProbs * 32, (size_t)n_prob * 32, hipMemcpyDeviceToHost) != hipSuccess) return FSEG_ERR_HIP;
    return FSEG_OK;
}
int fseg_debug_timed_class(fseg_ctx *c, int cls) {
    unsigned long long v = (unsigned long long)(long long)cls;
    if (!c || copy_sync(c, static_cast<char *>(c->d_tacc.p) + 120, &v, 8, hipMemcpyHostToDevice) != hipSuccess) return FSEG_ERR_HIP;
    return FSEG_OK;
}
#endif

int64_t fseg_scoring_algorithmic_bytes(fseg_ctx *c) {
    if (!c || !c->ran) return -1;
    // per partition 4*(N_p + K_p)*R_p + 4*R_p, R_p = read reps
    std::vector<i64> co((size_t)c->K + 1);
    if (copy_sync(c, co.data(), c->d_cand_off.p, co.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    i64 total = 0;
    for (int p = 0; p < c->n_part; ++p) {
        i64 Np = co[(size_t)c->part_iv_off[p + 1]] - co[(size_t)c->part_iv_off[p]];
        i64 Kp = c->part_iv_off[p + 1] - c->part_iv_off[p];
        i64 Rp = c->part_rep_off[p + 1] - c->part_rep_off[p];
        total += 4 * (Np + Kp) * Rp + 4 * Rp;
    }
    return total;
}

}  // extern "C"

