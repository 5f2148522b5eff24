// lps_abi.hip — the C-ABI of liblps_hip.so (include/lps_abi.h): context, device residency, stage orchestration.
//
// HBM layout (one chromosome resident per ctx; all SoA, sized for 288 GB: a 50x human chr1 batch is ~25 GB):
//   variants   pos i32 | ref0,alt0 u8 | ref_len,alt_len u16 | danger,hpoly,erased u8          (<= 16 B / variant)
//   reference  chars [0, lastVariant+5]
//   reads      ref_start,l_qseq i32 | flag u16 | mapq u8 | name_id u32 | seq/qual u64 offsets | cigar u32[] in lane-chunks of 8 (cp_off u32, cp_n i32) | seq 4-bit | qual u8
//   obs rows   var i32 + aq u16 per observation, rows placed by atomic reservation; g_node i32 + g_flag u8 mirror them
//   graph      node-major sorted (key u64, slot u32) list | edge f32[N][A][4] | vote records {f32 w,u32 flags}[N][A] | hp i8[N] | block i32[N]
// Every lps_phase_chromosome() recomputes all stages from the resident reads as the pushes left them: nothing is prepared or cached between a push
// and a call, or from one call to the next.
#include <algorithm>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <memory>
#include <chrono>
#include <cmath>
#include <climits>
#include <cstring>
#include <cerrno>
#include <unistd.h>

#include "lps_graph.h"
#include "lps_bam.h"
#include "lps_inflate.h"
#include "lps_bgzf_walk.h"
#include "lps_deflate.h"
#include "lps_stdsort.h"

static const char *kStageNames[LPS_MAX_STAGES] = {
    "variant_prep", "extract", "names_clips", "overlap_filter", "clip_cnv", "cnv_filter", "graph_rows",
    "edges", "vote_scan", "read_correction", "d2h", nullptr};
// recorded in this order on the stream
enum { ST_PREP, ST_EXTRACT, ST_GROUPS, ST_OVERLAP, ST_CLIP, ST_CNV, ST_NODES, ST_EDGES, ST_SCAN, ST_CORR, ST_D2H, ST_COUNT };

// A few persistent host threads that fill the page-locked pieces of a large upload (h2d_staged): created once per context, woken per piece - a piece
// of 64 MiB lasts 1.3 ms on the link, so the readers must deliver 50 GB/s and must not be created and joined per piece (round 3: four threads per
// piece, 22 GB/s - as long as the inflate beside it).
struct ReaderPool {
    std::vector<std::thread> th; std::mutex m; std::condition_variable cv_go, cv_done;
    std::function<void(int)> job; unsigned long long gen = 0; int pending = 0; bool stop = false;
    explicit ReaderPool(int n) { for (int t = 0; t < n; ++t) th.emplace_back([this, t] { unsigned long long seen = 0;
        for (;;) { std::unique_lock<std::mutex> lk(m); cv_go.wait(lk, [&] { return stop || gen != seen; }); if (stop) return; seen = gen; auto f = job; lk.unlock();
                   f(t); lk.lock(); if (--pending == 0) cv_done.notify_all(); } }); }
    void run(const std::function<void(int)> &f) { std::unique_lock<std::mutex> lk(m); job = f; pending = (int)th.size(); ++gen; cv_go.notify_all(); cv_done.wait(lk, [&] { return pending == 0; }); }
    int size() const { return (int)th.size(); }
    ~ReaderPool() { { std::lock_guard<std::mutex> lk(m); stop = true; } cv_go.notify_all(); for (auto &t : th) t.join(); }
};

struct lps_ctx {
    int device = 0;
    hipStream_t stream = nullptr, copy_stream = nullptr;   // copy_stream: the sorted clip keys leave beside the late stages
    hipEvent_t ev_sorted = nullptr;
    lps_params P{};
    std::string err;
    // variants
    int nV = 0; int last_pos = -1; long long ref_len = 0, ref_len_eff = 0;
    std::vector<int32_t> h_vpos; bool vpos_on_device_only = false;   // lps_set_variants_device: the host copy is fetched when something asks for it
    DevBuf<int32_t> v_pos; DevBuf<uint8_t> v_ref0, v_alt0, v_danger, v_hpoly, v_erased, v_hp1; DevBuf<uint16_t> v_rl, v_al;
    DevBuf<int32_t> v_ps, v_bucket; DevBuf<uint2> v_rec; bool has_hap = false;
    DevBuf<char> ref;
    // SV / MOD rows (lps_set_extra_variants): merged by position; nG / g_vpos = size and positions of the table the stages after the extraction
    // run on (the SNP table itself when there are no such rows, else the union of the three)
    int nX = 0, nSV = 0, nMOD = 0, sv_window = 20; double sv_threshold = 0.1;
    DevBuf<int32_t> x_pos, x_info, x_u, x_snp_u, u_pos, x_x0; DevBuf<uint32_t> x_redo; DevBuf<int4> x_rec; DevBuf<uint32_t> x_mpack; DevBuf<uint8_t> x_kind, x_mflag; DevBuf<uint32_t> x_moff, x_mname;
    std::vector<int32_t> h_snp_u, h_sv_u, h_mod_u, h_res_ps_u; std::vector<uint8_t> h_res_gt_u;
    int nG = 0; const int32_t *g_vpos = nullptr;
    std::vector<int32_t> votes_h1, votes_h2;   // lps_set_read_votes
    // reads
    int nR = 0; uint64_t n_cig = 0, n_seq = 0, n_qual = 0;
    DevBuf<int32_t> r_start, r_lq; DevBuf<uint16_t> r_flag; DevBuf<uint8_t> r_mapq; DevBuf<uint32_t> r_name;
    DevBuf<uint64_t> r_soff, r_qoff; DevBuf<uint8_t> seq, qual;
    // CIGAR words: ONE resident copy, in lane-chunks of 8 words (lps_reads.hip), written in that layout by every kind of push.  n_cig counts the real
    // words (algorithmic bytes), n_chunks the chunks in place; cp_off[r] = first chunk of alignment r (nR + 1 entries), cp_n[r] = its words
    DevBuf<uint32_t> cigar, cp_off, cp_cnt, cp_rel, cig_tmp; DevBuf<int32_t> cp_n; DevBuf<uint64_t> coff_tmp, chunk_off; DevBuf<unsigned long long> cig_words; uint64_t n_chunks = 0;
    double alloc_ms = 0.0;                       // lps_alloc_ms
    DevBuf<int32_t> r_v0; int32_t last_start = 0;   /* start of the last alignment pushed (order across pushes) */
    // raw BAM records (lps_push_bam_records): seq/qual are read in place from the blob
    DevBuf<uint8_t> blob; uint64_t n_blob = 0; int read_mode = 0;   // 0 none yet, 1 SoA batches, 2 BAM records
    DevBuf<uint64_t> rec_off, cig_src; DevBuf<unsigned long long> cig_cnt; DevBuf<unsigned> bam_err;
    // whole BAM file resident on the device (lps_bgzf_load): compressed bytes, block table, inflated stream; survives lps_begin_chromosome
    DevBuf<uint8_t> zfile, file, zscratch; DevBuf<InflateBlock> zblk; uint64_t file_bytes = 0; float bgzf_h2d_ms = 0, bgzf_inflate_ms = 0; bool bgzf_retried = false;
    DevBuf<unsigned long long> tg_len, tg_off; DevBuf<uint2> tg_spans; DevBuf<uint8_t> tg_stream, tg_status, tg_hp; DevBuf<int32_t> tg_ps, tg_pq; int64_t cur_first = -1, cur_count = 0;
    uint8_t *stage[2] = {nullptr, nullptr}; hipEvent_t stage_ev[2] = {nullptr, nullptr}; size_t stage_bytes = 0; std::unique_ptr<ReaderPool> readers;   // pinned staging ring for large pageable uploads
    unsigned long long *up_mark = nullptr;   // upload watermark of lps_bgzf_load: a page-locked host word the inflate kernel polls
    DevBuf<uint8_t> dz_slots, dz_packed, dz_src; DevBuf<uint32_t> dz_bytes; DevBuf<unsigned long long> dz_tmp; DevBuf<uint64_t> dz_off; uint64_t dz_total = 0; float dz_ms = 0;
    DevBuf<uint64_t> rcand; uint64_t n_rec_all = 0; DevBuf<int32_t> r_tid_all; DevBuf<uint32_t> r_lname, r_nameoff, wg_cnt, wg_off, scan_nout; DevBuf<uint8_t> names_d; bool names_ready = false;
    // observations
    DevBuf<RowDesc> rows; DevBuf<int32_t> g_cnt; DevBuf<uint8_t> deleted;
    DevBuf<ObsRec> obs; DevBuf<uint32_t> g_pack, g_rank, t_src, redo_list; DevBuf<int32_t> t_node; DevBuf<uint8_t> t_flag;
    unsigned long long obs_capacity = 0;   // main arenas (LPS_ARENAS equal parts); a tail arena of obs_capacity/4 follows
    DevBuf<unsigned long long> arena_ctr;
    bool in_phase = false; int timing_level = 1;
    uint32_t name_max = 0;        // largest name_id pushed for this chromosome: bounds the digits of the name sort
    size_t z_late_off = 0, z_late_bytes = 0; unsigned long long late_n_keys = 0, late_cap_main = 0, late_tail = 0; bool cnv_skipped = false;
    DevBuf<uint8_t> hap_pool;     // per-read outputs of the scoring kernels, carved like zpool
    DevBuf<uint4> hap_rec; DevBuf<int> pq_tab; bool pq_ready = false; DevBuf<int32_t> d_votes1, d_votes2;   // germline haplotag: packed per-read records, PQ table (host libm), SV / MOD votes
    DevBuf<uint8_t> zpool;        // the zero-initialised arrays of a phase run (arena_ctr, out_ps/gt, deleted, is_node, vtype_key, mrow_cnt, node_end/cur, bsize, cnt4) are carved from it
    // clips / cnv
    DevBuf<ClipEv> clip_ev; size_t clip_capacity = 0;
    DevBuf<unsigned long long> clip_keys, clip_keys_s; DevBuf<uint32_t> clip_tab; bool clips_sorted = false;
    DevBuf<int32_t> cnv_start, cnv_end;
    DevBuf<long long> agg_sum; DevBuf<int32_t> agg_cnt; DevBuf<double> miss;
    DevBuf<uint32_t> cnv_flag, cnv_idx, cnv_list, cnv_nlist; DevBuf<uint8_t> cnv_fn, cnv_pre;
    uint8_t *h_res = nullptr; size_t h_res_bytes = 0;   // pinned landing zone of (phase_set, gt): one copy, then memcpy into the caller's arrays
    unsigned *h_ncnv = nullptr; LpsCounters *h_cnt_pin = nullptr; unsigned *h_stats_pin = nullptr; hipEvent_t ev_cnv = nullptr;   // pinned block + event: the counters reach the host while the GPU keeps working
    unsigned long long *h_clip_keys = nullptr; size_t h_clip_cap = 0; hipEvent_t ev_clip = nullptr;   // sorted clip keys on their way to the host (pinned): the CNV state machine is replayed there
    int32_t *h_cnv_pin = nullptr; size_t h_cnv_pin_bytes = 0;
    std::vector<int32_t> h_cnv_start, h_cnv_end; bool cnv_expect = false; unsigned h_ub_hazard = 0;   // intervals of the current run (each once); cnv_expect: the previous run had intervals
    // groups
    DevBuf<unsigned long long> name_keys, name_keys_s;    // (only when the caller's name ids are not dense: ranks by one sort)
    DevBuf<uint32_t> head, gidx, name_dense, name_link, mm_r, stack, mg_start, mg_cnt, mg_name, mg_plan, mrow_off; DevBuf<int32_t> mrow_cnt;
    const uint32_t *name_p = nullptr; size_t name_cap = 0; bool key64 = false, scan_done = false; GraphView G{};
    // nodes / graph
    DevBuf<uint32_t> name_head, var_cnt, var_del, var_del2, vtype_key, node_of, var_off, node_off, node_cap, node_end, bsum, cnt4; DevBuf<uint8_t> bmulti;
    DevBuf<int32_t> nodes, block; DevBuf<uint8_t> erec; DevBuf<unsigned> clip_stats; DevBuf<int8_t> hp, hp_v; DevBuf<int32_t> blk_v, seg_i32; DevBuf<char> st_b,
            st_e; DevBuf<uint32_t> node_pairs; DevBuf<uint8_t> nstate;
    DevBuf<unsigned long long> nkeys, nkeys_s; DevBuf<uint32_t> nvals, nvals_s;   // node-major lists (keys: 32 bits each when name rank + row index fit, else 64)
    DevBuf<float> edge;
    DevBuf<int32_t> out_ps; DevBuf<uint8_t> out_gt;
    DevBuf<uint8_t> hap_status, hap_nps, v_role, v_derive, v_tkind, read_hp; DevBuf<int32_t> site, t_end, t_len, t_pair_site, t_pair_read,
            t_win_site; DevBuf<uint8_t> t_hp, t_has, t_pair_hp, t_win_allele,
            t_win_base; DevBuf<int16_t> t_win_off; DevBuf<unsigned long long> t_ctr; DevBuf<int4> t_hits; DevBuf<int> t_hit_rp;  DevBuf<int32_t> t_apair_site, t_apair_read; DevBuf<uint8_t> t_apair_hp; size_t t_hit_arena = 0, t_pair_arena = 0, n_pair_arena = 0; bool has_tkind = false; DevBuf<int32_t> hap_h1, hap_h2, hap_psmin, hap_h3, hap_d1,
            hap_d2; bool has_somatic = false;
    DevBuf<char> temp; size_t temp_bytes = 0;
    LpsCounters *d_cnt = nullptr; LpsCounters h_cnt{}; unsigned h_stats[4]{};
    // timing
    hipEvent_t ev[ST_COUNT + 1]{}; bool ev_used[ST_COUNT + 1]{};
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    lps_timings tm{};
    bool phase_valid = false;
    int m_bits = 0, a_bits = 16, n_bits = 0;
};

#define LPS_MAX_ROWS 0xfffffff   /* rows of the table the graph runs on (SNP + SV + MOD): a packed word holds node << 2 | flags; what really bounds a table is memory (560 B per row for the edge matrix) */
static int fail(lps_ctx *c, const std::string &m, int code = -1) { if (c) c->err = m; return code; }

// Large upload from pageable memory (an mmap of the BAM file): the runtime's own path stages through ONE host thread's memcpy; here four threads fill a
// pinned 2 x 64 MiB ring while the DMA engine drains the other half.
// Where a large upload takes its bytes from: memory, or a file read with pread straight into the page-locked pieces (no mapping of the file: an 8 GB
// mapping costs two million page-table entries to set up while it is copied and 0.14 s to tear down when the process ends).
// (ZSource: lps_bgzf_walk.h)
// `mark` (optional): a word in PAGE-LOCKED HOST memory that this (host) thread raises to the number of bytes known to be in place - after the event of
// a piece has been waited for, i.e. two pieces late, and n at the end.  It is what a kernel launched beside the upload polls (k_bgzf_inflate): host
// memory, because a word in device memory written by the copy engine would sit stale in the polling XCD's L2 until the next kernel boundary.
static void h2d_staged(lps_ctx *c, uint8_t *dst, const ZSource &src, size_t n, hipStream_t st = nullptr, unsigned long long *mark = nullptr) {
    const size_t CH = 64u << 20;
    if (!st) st = c->stream;
    auto raise = [&](size_t bytes) { if (mark) __atomic_store_n(mark, (unsigned long long)bytes, __ATOMIC_RELEASE); };
    if (n < (16u << 20)) {
        if (src.mem) { HIP_TRY(hipMemcpyAsync(dst, src.mem, n, hipMemcpyHostToDevice, st)); if (mark) HIP_TRY(hipStreamSynchronize(st)); }
        else { std::vector<uint8_t> tmp(n); if (!src.read(0, n, tmp.data())) throw std::string("cannot read the file"); HIP_TRY(hipMemcpyAsync(dst, tmp.data(), n, hipMemcpyHostToDevice, st)); HIP_TRY(hipStreamSynchronize(st)); }
        raise(n); return;
    }
    if (!c->stage[0]) { for (int k = 0; k < 2; ++k) { HIP_TRY(hipHostMalloc((void **)&c->stage[k], CH,
            hipHostMallocDefault)); HIP_TRY(hipEventCreateWithFlags(&c->stage_ev[k], hipEventDisableTiming)); } c->stage_bytes = CH; }
    int k = 0; bool used[2] = {false, false}; size_t end_of[2] = {0, 0};
    for (size_t off = 0; off < n; off += CH, k ^= 1) {
        const size_t len = std::min(CH, n - off);
        if (used[k]) { HIP_TRY(hipEventSynchronize(c->stage_ev[k])); raise(end_of[k]); }       // (pieces complete in order: everything before end_of[k] is in place)
        if (mark) { const char *thr = getenv("LPS_BGZF_TEST_THROTTLE_MS"); if (thr) usleep((useconds_t)(atof(thr) * 1000.0)); }   // test hook: a slow source (network storage, a cold page cache)
        if (!c->readers) { const char *e = getenv("LPS_UPLOAD_THREADS"); c->readers.reset(new ReaderPool(std::max(1, std::min(64, e ? atoi(e) : 12)))); }
        const int nt = c->readers->size(); const size_t part = ((len + nt - 1) / nt + 4095) & ~(size_t)4095; std::vector<char> ok((size_t)nt, 1);
        c->readers->run([&](int t) { const size_t a = std::min(len, part * (size_t)t), b = std::min(len, a + part); ok[(size_t)t] = b <= a || src.read(off + a, b - a, c->stage[k] + a); });
        for (int t = 0; t < nt; ++t) if (!ok[(size_t)t]) throw std::string("cannot read the file");
        HIP_TRY(hipMemcpyAsync(dst + off, c->stage[k], len, hipMemcpyHostToDevice, st));
        HIP_TRY(hipEventRecord(c->stage_ev[k], st)); used[k] = true; end_of[k] = off + len;
    }
    if (mark) { for (int q = 0; q < 2; ++q) if (used[q]) HIP_TRY(hipEventSynchronize(c->stage_ev[q])); raise(n); }
}
static void h2d_staged(lps_ctx *c, uint8_t *dst, const uint8_t *src, size_t n, hipStream_t st = nullptr, unsigned long long *mark = nullptr) { h2d_staged(c, dst, ZSource{src, -1, 0}, n, st, mark); }

template <class T>
static void upload(lps_ctx *c, DevBuf<T> &b, const T *src, size_t n, size_t at = 0, bool keep = false) {
    b.reserve(at + n, c->stream, keep, at);
    if (n) HIP_TRY(hipMemcpyAsync(b.p + at, src, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
}

template <class T>
static std::vector<T> download(lps_ctx *c, const T *p, size_t n) {
    std::vector<T> h(n);
    if (n) HIP_TRY(hipMemcpyAsync(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return h;
}

extern "C" {

int lps_abi_version(void) { return LPS_ABI_VERSION; }

void lps_debug_std_sort(int32_t *keys, uint8_t *payload, int64_t n) { stdsort_pairs(keys, payload, (int)n); }
int lps_debug_std_sort_gpu(int device, int32_t *keys, uint8_t *payload, const int64_t *row_start, int64_t n_rows) {
    if (n_rows <= 0) return 0;
    if (hipSetDevice(device) != hipSuccess) return -1;
    const size_t n = (size_t)row_start[n_rows]; int32_t *dk = nullptr; uint8_t *dp = nullptr; long long *dr = nullptr; int rc = -1;
    if (hipMalloc(&dk, n * 4 + 4) == hipSuccess && hipMalloc(&dp, n + 4) == hipSuccess && hipMalloc(&dr, (size_t)(n_rows + 1) * 8) == hipSuccess &&
        hipMemcpy(dk, keys, n * 4, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(dp, payload, n, hipMemcpyHostToDevice) == hipSuccess &&
        hipMemcpy(dr, row_start, (size_t)(n_rows + 1) * 8, hipMemcpyHostToDevice) == hipSuccess) {
        launch_debug_std_sort(dk, dp, dr, (int)n_rows, 0);
        if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(keys, dk, n * 4, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(payload, dp, n, hipMemcpyDeviceToHost) == hipSuccess) rc = 0;
    }
    (void)hipFree(dk); (void)hipFree(dp); (void)hipFree(dr);
    return rc;
}

int lps_struct_size(int which) {
    switch (which) {
        case 0: return (int)sizeof(lps_params); case 1: return (int)sizeof(lps_variant_table); case 2: return (int)sizeof(lps_read_batch);
        case 3: return (int)sizeof(lps_phase_result); case 4: return (int)sizeof(lps_haplotag_result); case 5: return (int)sizeof(lps_timings); case 6: return (int)sizeof(lps_somatic_tag_result); case 7: return (int)sizeof(lps_site_counters); case 8: return (int)sizeof(lps_tumor_extract_result); case 9: return (int)sizeof(lps_extra_variants);
    }
    return -1;
}

int lps_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
int lps_device_bus_id(int device, char *buf, int len) { if (!buf || len < 16) return -1; return hipDeviceGetPCIBusId(buf, len, device) == hipSuccess ? 0 : -1; }

void lps_default_params(lps_params *p) {
    memset(p, 0, sizeof *p);
    p->is_ont = 1; p->phase_indel = 0; p->distance = 300000; p->connect_adjacent = 35; p->mapping_quality = 1;
    p->base_quality = 12; p->edge_weight = 0.1; p->snp_confidence = 0.75; p->read_confidence = 0.65;
    p->edge_threshold = 0.7; p->overlap_threshold = 0.2; p->percentage_threshold = 0.6; p->tag_supplementary = 0;
}

const char *lps_stage_name(int i) { return (i >= 0 && i < ST_COUNT) ? kStageNames[i] : nullptr; }

lps_ctx *lps_create(int device, const lps_params *params) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { fprintf(stderr, "lps_create: no HIP device available (the product path has no CPU fallback)\n"); return nullptr; }
    if (device < 0 || device >= n) { fprintf(stderr, "lps_create: device %d out of range (%d devices)\n", device, n); return nullptr; }
    if (params->connect_adjacent < 1 || params->connect_adjacent > LPS_MAX_ADJACENT) { fprintf(stderr, "lps_create: connect_adjacent must be in [1,%d]\n", LPS_MAX_ADJACENT); return nullptr; }
    lps_ctx *c = new lps_ctx();
    try {
        c->device = device; c->P = *params;
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_sorted, hipEventDisableTiming));
        HIP_TRY(hipMalloc((void **)&c->d_cnt, sizeof(LpsCounters)));
        for (auto &e : c->ev) HIP_TRY(hipEventCreate(&e));
        HIP_TRY(hipEventCreate(&c->ev_begin)); HIP_TRY(hipEventCreate(&c->ev_end)); HIP_TRY(hipEventCreate(&c->ev_cnv)); HIP_TRY(hipEventCreate(&c->ev_clip));
        HIP_TRY(hipHostMalloc((void **)&c->h_ncnv, 1024));          // pinned: copies into it do not block the host
        c->h_cnt_pin = (LpsCounters *)(c->h_ncnv + 16); c->h_stats_pin = c->h_ncnv + 192;
        static_assert(sizeof(LpsCounters) <= 512, "pinned block layout");
    } catch (std::string &e) { fprintf(stderr, "lps_create: %s\n", e.c_str()); delete c; return nullptr; }
    return c;
}

void lps_destroy(lps_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); }
    for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
    if (c->ev_begin) (void)hipEventDestroy(c->ev_begin);
    if (c->ev_end) (void)hipEventDestroy(c->ev_end);
    if (c->ev_cnv) (void)hipEventDestroy(c->ev_cnv);
    if (c->ev_clip) (void)hipEventDestroy(c->ev_clip);
    if (c->h_clip_keys) (void)hipHostFree(c->h_clip_keys);
    if (c->h_cnv_pin) (void)hipHostFree(c->h_cnv_pin);
    if (c->h_ncnv) (void)hipHostFree(c->h_ncnv);
    if (c->h_res) (void)hipHostFree(c->h_res);
    if (c->up_mark) (void)hipHostFree(c->up_mark);
    for (int k = 0; k < 2; ++k) { if (c->stage[k]) (void)hipHostFree(c->stage[k]); if (c->stage_ev[k]) (void)hipEventDestroy(c->stage_ev[k]); }
    if (c->d_cnt) (void)hipFree(c->d_cnt);
    if (c->ev_sorted) (void)hipEventDestroy(c->ev_sorted);
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *lps_last_error(lps_ctx *c) { return c ? c->err.c_str() : "null ctx"; }
void *lps_stream(lps_ctx *c) { return c ? (void *)c->stream : nullptr; }

int lps_begin_chromosome(lps_ctx *c) {
    if (!c) return -1;
    c->nV = 0; c->last_pos = -1; c->ref_len = c->ref_len_eff = 0; c->nR = 0; c->n_cig = c->n_seq = c->n_qual = 0; c->n_chunks = 0; c->n_blob = 0; c->read_mode = 0; c->cur_first = -1; c->cur_count = 0;
    c->phase_valid = false; c->has_hap = false; c->h_vpos.clear(); c->vpos_on_device_only = false; c->name_max = 0;
    c->nX = c->nSV = c->nMOD = 0; c->h_snp_u.clear(); c->h_sv_u.clear(); c->h_mod_u.clear(); c->votes_h1.clear(); c->votes_h2.clear();
    return 0;
}

int lps_set_variants(lps_ctx *c, const lps_variant_table *t) {
    if (!c || !t) return -1;
    try {
        HIP_TRY(hipSetDevice(c->device));
        if (t->n > LPS_MAX_ROWS) return fail(c, "variant table larger than 2^28 rows per chromosome");
        for (int64_t i = 1; i < t->n; ++i) if (t->pos[i] <= t->pos[i - 1]) return fail(c, "variant positions must be strictly increasing");
        c->nV = (int)t->n; c->last_pos = t->n ? t->pos[t->n - 1] : -1;
        c->h_vpos.assign(t->pos, t->pos + t->n); c->vpos_on_device_only = false;
        upload(c, c->v_pos, t->pos, t->n); upload(c, c->v_ref0, t->ref0, t->n); upload(c, c->v_alt0, t->alt0, t->n);
        upload(c, c->v_rl, t->ref_len, t->n); upload(c, c->v_al, t->alt_len, t->n);
        c->v_danger.reserve(t->n + 1); c->v_hpoly.reserve(t->n + 1); c->v_erased.reserve(t->n + 1);
        c->has_hap = t->hp1_is_alt && t->phase_set;
        if (c->has_hap) { upload(c, c->v_hp1, t->hp1_is_alt, t->n); upload(c, c->v_ps, t->phase_set, t->n); }
        c->has_somatic = c->has_hap && t->somatic_role && t->derive_hp;
        if (c->has_somatic) { upload(c, c->v_role, t->somatic_role, t->n); upload(c, c->v_derive, t->derive_hp, t->n); }
        c->has_tkind = c->has_hap && t->somatic_role && t->tumor_kind;
        if (c->has_tkind) { upload(c, c->v_role, t->somatic_role, t->n); upload(c, c->v_tkind, t->tumor_kind, t->n); }
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->phase_valid = false;
        c->nX = c->nSV = c->nMOD = 0; c->h_snp_u.clear(); c->h_sv_u.clear(); c->h_mod_u.clear();      // a new SNP table: extra rows must be set again
    } catch (std::string &e) { return fail(c, e); }
    return 0;
}

// lps_set_variants for a table that is already on the ctx's GPU (e.g. the communicator's buffer after lps_comm_bcast_to_device: the table goes
// ncclBroadcast -> context without touching the host).  flags[0] bit0: positions not strictly increasing
__global__ void k_table_check(long long n, const int32_t *pos, unsigned *flags) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if ((i && pos[i] <= pos[i - 1]) || pos[i] < 0) atomicOr(&flags[0], 1u);
}
__global__ void k_fill_u16(long long n, uint16_t v, uint16_t *out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = v;
}
int lps_set_variants_device(lps_ctx *c, const lps_variant_table *t) {
    if (!c || !t) return -1;
    try {
        HIP_TRY(hipSetDevice(c->device));
        if (t->n > LPS_MAX_ROWS) return fail(c, "variant table larger than 2^28 rows per chromosome");
        if (t->hp1_is_alt || t->phase_set || t->somatic_role || t->derive_hp || t->tumor_kind) return fail(c, "lps_set_variants_device takes the phase columns only (pos, ref0, alt0, ref_len, alt_len)");
        if (t->n && (!t->pos || !t->ref0 || !t->alt0)) return fail(c, "lps_set_variants_device: pos / ref0 / alt0 missing");
        hipStream_t s = c->stream; const size_t n = (size_t)t->n;
        c->bam_err.reserve(2);
        HIP_TRY(hipMemsetAsync(c->bam_err.p, 0, sizeof(unsigned), s));
        int32_t last = -1; unsigned flags = 0;
        if (n) {
            hipLaunchKernelGGL(k_table_check, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (long long)n, t->pos, c->bam_err.p);
            HIP_TRY(hipMemcpyAsync(&last, t->pos + n - 1, sizeof last, hipMemcpyDeviceToHost, s));
        }
        HIP_TRY(hipMemcpyAsync(&flags, c->bam_err.p, sizeof flags, hipMemcpyDeviceToHost, s));
        auto d2d = [&](auto &buf, const auto *src) { buf.reserve(n + 1); if (n) HIP_TRY(hipMemcpyAsync(buf.p, src, n * sizeof(*src), hipMemcpyDeviceToDevice, s)); };
        d2d(c->v_pos, t->pos); d2d(c->v_ref0, t->ref0); d2d(c->v_alt0, t->alt0);
        c->v_rl.reserve(n + 1); c->v_al.reserve(n + 1);
        const dim3 g((unsigned)((n + 255) / 256)), b(256);
        if (t->ref_len) d2d(c->v_rl, t->ref_len); else if (n) hipLaunchKernelGGL(k_fill_u16, g, b, 0, s, (long long)n, (uint16_t)1, c->v_rl.p);   // NULL: every row is a SNP
        if (t->alt_len) d2d(c->v_al, t->alt_len); else if (n) hipLaunchKernelGGL(k_fill_u16, g, b, 0, s, (long long)n, (uint16_t)1, c->v_al.p);
        c->v_danger.reserve(n + 1); c->v_hpoly.reserve(n + 1); c->v_erased.reserve(n + 1);
        HIP_TRY(hipStreamSynchronize(s));
        if (flags & 1u) { c->nV = 0; return fail(c, "variant positions must be strictly increasing"); }
        c->nV = (int)n; c->last_pos = last; c->h_vpos.clear(); c->vpos_on_device_only = n != 0;
        c->has_hap = c->has_somatic = c->has_tkind = false;
        c->phase_valid = false;
        c->nX = c->nSV = c->nMOD = 0; c->h_snp_u.clear(); c->h_sv_u.clear(); c->h_mod_u.clear();
    } catch (std::string &e) { return fail(c, e); }
    return 0;
}

// SV_map / currentMod of BamParser (src/phase/ParsingBam.cpp:1207-1235) as one position-sorted list next to the SNP table
int lps_set_extra_variants(lps_ctx *c, const lps_extra_variants *x) {
    if (!c) return -1;
    try {
        HIP_TRY(hipSetDevice(c->device));
        c->nX = c->nSV = c->nMOD = 0; c->h_snp_u.clear(); c->h_sv_u.clear(); c->h_mod_u.clear(); c->phase_valid = false;
        if (!x || (x->n_sv <= 0 && x->n_mod <= 0)) return 0;
        if (c->nV == 0) return fail(c, "lps_set_variants must be called before lps_set_extra_variants");
        if (c->vpos_on_device_only) { c->h_vpos = download(c, c->v_pos.p, (size_t)c->nV); c->vpos_on_device_only = false; }   // the merge below runs on the host
        const int64_t nS = std::max<int64_t>(x->n_sv, 0), nM = std::max<int64_t>(x->n_mod, 0);
        if ((int64_t)c->nV + nS + nM > LPS_MAX_ROWS) return fail(c, "SNP + SV + MOD rows exceed 2^28 per chromosome");
        if (x->sv_window < 0 || !(x->sv_threshold >= 0 && x->sv_threshold <= 1)) return fail(c, "invalid svWindow / svThreshold");     // Phasing.cpp:304-318
        for (int64_t i = 1; i < nS; ++i) if (x->sv_pos[i] <= x->sv_pos[i - 1]) return fail(c, "SV positions must be strictly increasing");
        for (int64_t i = 1; i < nM; ++i) if (x->mod_pos[i] <= x->mod_pos[i - 1]) return fail(c, "MOD positions must be strictly increasing");
        if (nM && x->mod_off[0] != 0) return fail(c, "mod_off[0] must be 0");
        if (nM && x->mod_off[nM] > 0xffffffffull) return fail(c, "more than 2^32 MOD read entries");
        for (int64_t m = 0; m < nM; ++m) {
            if (x->mod_off[m + 1] < x->mod_off[m]) return fail(c, "mod_off must be non-decreasing");
            for (uint64_t k = x->mod_off[m] + 1; k < x->mod_off[m + 1]; ++k) if (x->mod_name[k] <= x->mod_name[k - 1]) return fail(c, "mod_name must be strictly increasing inside a row");
        }
        // merge the three position lists; a position in two of them would make get_snp's inner loop spin forever (none of its three branches
        // serves a row that ties with another cursor, ParsingBam.cpp:1373,1397,1437)
        const size_t nX = (size_t)(nS + nM), nU = nX + (size_t)c->nV;
        std::vector<int32_t> xpos(nX), xinfo(nX), xu(nX), upos(nU), xq(nX); std::vector<uint8_t> xkind(nX);   // xq: position of the last SNP row before the row (INT_MIN: none)
        int32_t last_snp = INT_MIN;
        c->h_snp_u.resize(c->nV); c->h_sv_u.resize(nS); c->h_mod_u.resize(nM);
        size_t a = 0, sv = 0, md = 0, k = 0, u = 0;
        while (a < (size_t)c->nV || sv < (size_t)nS || md < (size_t)nM) {
            const long long pa = a < (size_t)c->nV ? c->h_vpos[a] : LLONG_MAX, ps = sv < (size_t)nS ? x->sv_pos[sv] : LLONG_MAX, pm = md < (size_t)nM ? x->mod_pos[md] : LLONG_MAX;
            const long long lo = std::min(pa, std::min(ps, pm));
            if ((pa == lo) + (ps == lo) + (pm == lo) > 1) return fail(c, "position " + std::to_string(lo) + " occurs in more than one of the SNP / SV / MOD tables: the reference does not terminate on such input");
            upos[u] = (int32_t)lo;
            if (pa == lo) { c->h_snp_u[a++] = (int32_t)u; last_snp = (int32_t)lo; }
            else if (ps == lo) { xq[k] = last_snp; xpos[k] = (int32_t)lo; xinfo[k] = x->sv_len[sv]; xkind[k] = 1; xu[k] = (int32_t)u; c->h_sv_u[sv++] = (int32_t)u; ++k; }
            else { xq[k] = last_snp; xpos[k] = (int32_t)lo; xinfo[k] = (int32_t)md; xkind[k] = 2; xu[k] = (int32_t)u; c->h_mod_u[md++] = (int32_t)u; ++k; }
            ++u;
        }
        std::vector<uint32_t> moff(nM + 1, 0u);
        for (int64_t m = 0; m <= nM && nM; ++m) moff[m] = (uint32_t)x->mod_off[m];
        upload(c, c->x_pos, xpos.data(), nX); upload(c, c->x_info, xinfo.data(), nX); upload(c, c->x_kind, xkind.data(), nX); upload(c, c->x_u, xu.data(), nX);
        upload(c, c->x_snp_u, c->h_snp_u.data(), (size_t)c->nV); upload(c, c->u_pos, upos.data(), nU);
        upload(c, c->x_moff, moff.data(), (size_t)nM + 1);
        const size_t ne = nM ? (size_t)x->mod_off[nM] : 0;
        c->x_mname.reserve(ne + 1); c->x_mflag.reserve(ne + 1);
        if (ne) { upload(c, c->x_mname, x->mod_name, ne); upload(c, c->x_mflag, x->mod_flag, ne); }
        // what k_extra_find reads per row / per listed read in ONE load each: {pos, info, union index | kind << 30, last SNP position before it} and name << 2 | flags
        if (nU >= (1ull << 30)) return fail(c, "lps_set_extra_variants: more than 2^30 rows in the union of the tables");
        std::vector<int4> xrec(nX); for (size_t i = 0; i < nX; ++i) xrec[i] = make_int4(xpos[i], xinfo[i], xu[i] | ((int)xkind[i] << 30), xq[i]);
        std::vector<uint32_t> mpack(ne);
        for (size_t i = 0; i < ne; ++i) { if (x->mod_name[i] >= (1u << 30)) return fail(c, "lps_set_extra_variants: read name ids must be below 2^30"); mpack[i] = (x->mod_name[i] << 2) | (uint32_t)(x->mod_flag[i] & 3u); }
        upload(c, c->x_rec, xrec.data(), (size_t)nX); c->x_mpack.reserve(ne + 1); if (ne) upload(c, c->x_mpack, mpack.data(), ne);
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->nX = (int)nX; c->nSV = (int)nS; c->nMOD = (int)nM; c->sv_window = x->sv_window; c->sv_threshold = x->sv_threshold;
    } catch (std::string &e) { return fail(c, e); }
    return 0;
}

int lps_get_extra_result(lps_ctx *c, lps_phase_result *sv, lps_phase_result *mod) {
    if (!c) return -1;
    if (!c->phase_valid) return fail(c, "no phase result");
    if ((sv && sv->n != c->nSV) || (mod && mod->n != c->nMOD)) return fail(c, "lps_get_extra_result: n must equal the table sizes given to lps_set_extra_variants");
    const bool have = c->nX > 0 && c->h_res_ps_u.size() == (size_t)c->nG;       // nothing ran (no reads): every row unphased
    if (sv) for (int i = 0; i < c->nSV; ++i) { sv->phase_set[i] = have ? c->h_res_ps_u[c->h_sv_u[i]] : 0; sv->gt[i] = have ? c->h_res_gt_u[c->h_sv_u[i]] : 0; }
    if (mod) for (int i = 0; i < c->nMOD; ++i) { mod->phase_set[i] = have ? c->h_res_ps_u[c->h_mod_u[i]] : 0; mod->gt[i] = have ? c->h_res_gt_u[c->h_mod_u[i]] : 0; }
    return 0;
}

int lps_set_reference(lps_ctx *c, const char *seq, int64_t len) {
    if (!c || !seq) return -1;
    try {
        HIP_TRY(hipSetDevice(c->device));
        if (c->nV == 0) return fail(c, "lps_set_variants must be called before lps_set_reference");
        const long long eff = std::min<long long>(len, (long long)c->last_pos + 6);    // ParsingBam.cpp:47
        if (eff <= c->last_pos) return fail(c, "reference shorter than the last variant position");
        c->ref_len = len; c->ref_len_eff = eff;
        upload(c, c->ref, seq, (size_t)eff);
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->phase_valid = false;
    } catch (std::string &e) { return fail(c, e); }
    return 0;
}

// CIGAR words of a batch - dense on the device: words [d_off[i], d_off[i + 1]) of d_src, offsets as the caller numbers them - appended to the resident
// lane-chunk layout behind the chunks already there; fills cp_n[at + i] and cp_off[at .. at + n].  Synchronizes the stream.
static int append_cigar_chunks(lps_ctx *c, size_t at, size_t n, const uint64_t *d_off, const uint32_t *d_src) {
    hipStream_t s = c->stream;
    c->cp_cnt.reserve(n + 2, s); c->cp_rel.reserve(n + 2, s);
    c->cp_n.reserve(at + n + 1, s, true, at); c->cp_off.reserve(at + n + 2, s, true, at);
    const size_t need = GraphTemp::need(n + 2);
    if (need > c->temp_bytes) { c->temp.reserve(need, s); c->temp_bytes = need; }
    unsigned *flag = reinterpret_cast<unsigned *>(c->cp_cnt.p + n + 1);
    HIP_TRY(hipMemsetAsync(flag, 0, 4, s));
    launch_cp_count((int)n, d_off, c->cp_cnt.p, c->cp_n.p + at, flag, s);
    exscan_u32(c->temp.p, c->temp_bytes, c->cp_cnt.p, c->cp_rel.p, n + 1, s);
    uint32_t total = 0, bad = 0;
    HIP_TRY(hipMemcpyAsync(&total, c->cp_rel.p + n, sizeof total, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&bad, flag, sizeof bad, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (bad || c->n_chunks + (uint64_t)total > 0xffffffffull) return fail(c, "CIGAR arrays beyond 2^35 words per chromosome (or 2^31 per alignment) are not supported");
    c->cigar.reserve(8 * (c->n_chunks + total) + 64, s, true, 8 * c->n_chunks);
    launch_cp_pack((int)n, d_off, d_src, c->cp_rel.p, (uint32_t)c->n_chunks, c->cp_off.p + at, c->cigar.p, s);
    HIP_TRY(hipStreamSynchronize(s));
    c->n_chunks += total;
    return 0;
}

int lps_push_reads(lps_ctx *c, const lps_read_batch *b) {
    if (!c || !b) return -1;
    try {
        HIP_TRY(hipSetDevice(c->device));
        const size_t n = (size_t)b->n_reads; if (n == 0) return 0;
        if (c->read_mode >= 2) return fail(c, "lps_push_reads mixed with a BAM-record push in the same chromosome");
        c->read_mode = 1;
        if ((uint64_t)c->nR + n > 0x1fffffffull) return fail(c, "more than 2^29 alignments per chromosome");
        const uint64_t nc = b->cigar_off[n] - b->cigar_off[0], ns = b->seq_off[n] - b->seq_off[0], nq = b->qual_off[n] - b->qual_off[0];
        // operand shapes the kernels rely on
        for (size_t i = 0; i < n; ++i) {
            if (b->cigar_off[i + 1] < b->cigar_off[i] || b->seq_off[i + 1] < b->seq_off[i] || b->qual_off[i + 1] < b->qual_off[i]) return fail(c, "offsets must be non-decreasing");
            if (b->l_qseq[i] < 0 || (uint64_t)((b->l_qseq[i] + 1) / 2) > b->seq_off[i + 1] - b->seq_off[i] || (uint64_t)b->l_qseq[i] > b->qual_off[i + 1] - b->qual_off[i]) return fail(c, "seq/qual shorter than l_qseq");
            if (i && b->ref_start[i] < b->ref_start[i - 1]) return fail(c, "alignments must be coordinate-sorted");
        }
        if (n && c->nR && b->ref_start[0] < c->last_start) return fail(c, "alignments must be coordinate-sorted (a batch starts before the end of the one pushed before it)");
        if (n) c->last_start = b->ref_start[n - 1];
        const size_t at = (size_t)c->nR;
        upload(c, c->r_start, b->ref_start, n, at, true); upload(c, c->r_lq, b->l_qseq, n, at, true);
        upload(c, c->r_flag, b->flag, n, at, true); upload(c, c->r_mapq, b->mapq, n, at, true);
        upload(c, c->r_name, b->name_id, n, at, true);
        for (size_t i = 0; i < n; ++i) c->name_max = std::max(c->name_max, b->name_id[i]);
        std::vector<uint64_t> co(n + 1), so(n + 1), qo(n + 1);
        for (size_t i = 0; i <= n; ++i) { co[i] = b->cigar_off[i] - b->cigar_off[0]; so[i] = b->seq_off[i] - b->seq_off[0] + c->n_seq; qo[i] = b->qual_off[i] - b->qual_off[0] + c->n_qual; }
        upload(c, c->r_soff, so.data(), n + 1, at, true); upload(c, c->r_qoff, qo.data(), n + 1, at, true);
        upload(c, c->coff_tmp, co.data(), n + 1); upload(c, c->cig_tmp, b->cigar + b->cigar_off[0], (size_t)nc);   // the batch's words as they come; they reach the resident layout below
        upload(c, c->seq, b->seq + b->seq_off[0], (size_t)ns, (size_t)c->n_seq, true);
        upload(c, c->qual, b->qual + b->qual_off[0], (size_t)nq, (size_t)c->n_qual, true);
        const int rc = append_cigar_chunks(c, at, n, c->coff_tmp.p, c->cig_tmp.p);     // (synchronizes: co/so/qo are stack-owned)
        if (rc) return rc;
        c->nR += (int)n; c->n_cig += nc; c->n_seq += ns; c->n_qual += nq;
        c->phase_valid = false;
    } catch (std::string &e) { return fail(c, e); }
    return 0;
}

// ---- device-resident batches (lps_push_reads_device): the operand checks of lps_push_reads run as a kernel
// flags[0]: bit0 offsets not non-decreasing, bit1 seq/qual shorter than l_qseq, bit2 not coordinate-sorted; flags[1]: largest name_id
__global__ void k_batch_check(long long n, const int32_t *ref_start, const int32_t *l_qseq, const uint32_t *name_id, const uint64_t *coff, const uint64_t *soff,
                              const uint64_t *qoff, unsigned *flags) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned f = 0;
    if (coff[i + 1] < coff[i] || soff[i + 1] < soff[i] || qoff[i + 1] < qoff[i]) f |= 1u;
    const int lq = l_qseq[i];
    if (lq < 0 || (uint64_t)((lq + 1) / 2) > soff[i + 1] - soff[i] || (uint64_t)lq > qoff[i + 1] - qoff[i]) f |= 2u;
    if (i && ref_start[i] < ref_start[i - 1]) f |= 4u;
    if (f) atomicOr(&flags[0], f);
    unsigned m = name_id[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, d));
    if (lane_id() == 0) atomicMax(&flags[1], m);
}
__global__ void k_rebase_offsets(long long n1, const uint64_t *in, uint64_t add, uint64_t *out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n1) out[i] = in[i] - in[0] + add;
}

// core decode + CIGAR re-alignment of n records described by B (blob already on the device), appended to the ctx's read arrays
static int push_record_view(lps_ctx *c, const BamView &B, size_t n, const uint32_t *name_id) {
    const size_t at = (size_t)c->nR; hipStream_t s = c->stream;
    upload(c, c->r_name, name_id, n, at, true);
    for (size_t i = 0; i < n; ++i) c->name_max = std::max(c->name_max, name_id[i]);
    c->r_start.reserve(at + n, s, true, at); c->r_lq.reserve(at + n, s, true, at); c->r_flag.reserve(at + n, s, true, at); c->r_mapq.reserve(at + n, s, true, at);
    c->r_soff.reserve(at + n + 1, s, true, at); c->r_qoff.reserve(at + n + 1, s, true, at);
    c->cp_n.reserve(at + n + 1, s, true, at); c->cp_off.reserve(at + n + 2, s, true, at);
    c->cig_cnt.reserve(n + 1); c->cig_src.reserve(n + 1); c->chunk_off.reserve(n + 2); c->bam_err.reserve(1); c->cig_words.reserve(1);
    HIP_TRY(hipMemsetAsync(c->bam_err.p, 0, sizeof(unsigned), s)); HIP_TRY(hipMemsetAsync(c->cig_words.p, 0, sizeof(unsigned long long), s));
    launch_bam_core(B, (int)n, (int)at, c->r_start.p, c->r_lq.p, c->r_flag.p, c->r_mapq.p, c->r_soff.p, c->r_qoff.p, c->cig_cnt.p, c->cp_n.p, c->cig_src.p, c->bam_err.p, s);
    bam_cigar_offsets(c->temp, c->temp_bytes, c->cig_cnt.p, c->chunk_off.p, (int)n, c->n_chunks, s);       // lane-chunks: first chunk of every record, behind the chunks in place
    launch_sum_i32(c->cp_n.p + at, (int)n, c->cig_words.p, s);
    uint64_t total = 0; unsigned err = 0; unsigned long long words = 0;
    HIP_TRY(hipMemcpyAsync(&total, c->chunk_off.p + n, sizeof total, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&err, c->bam_err.p, sizeof err, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&words, c->cig_words.p, sizeof words, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (err & LPS_BAM_ERR_BOUNDS) return fail(c, "BAM record does not fit the bytes handed over (truncated or corrupt record)");
    if (err & LPS_BAM_ERR_UNSORTED) return fail(c, "alignments must be coordinate-sorted");
    if (total > 0xffffffffull) return fail(c, "CIGAR arrays beyond 2^35 words per chromosome are not supported");
    c->cigar.reserve(8 * total + 64, s, true, 8 * c->n_chunks);
    launch_bam_cigar(B, (int)n, c->chunk_off.p, c->cp_n.p + at, c->cig_src.p, c->cigar.p, c->cp_off.p + at, s);   // the words re-aligned straight into their lane-chunks
    HIP_TRY(hipStreamSynchronize(s));
    c->nR += (int)n; c->n_chunks = total; c->n_cig += words;
    c->phase_valid = false;
    return 0;
}

int lps_push_bam_records(lps_ctx *c, const uint8_t *records, int64_t n_bytes, const uint64_t *rec_off, int64_t n_records, const uint32_t *name_id) {
    if (!c || (n_records > 0 && (!records || !rec_off || !name_id))) return -1;
    try {
        HIP_TRY(hipSetDevice(c->device));
        const size_t n = (size_t)n_records; if (n == 0) return 0;
        if (c->read_mode == 1 || c->read_mode == 3) return fail(c, "lps_push_bam_records mixed with another kind of push in the same chromosome");
        if ((uint64_t)c->nR + n > 0x1fffffffull) return fail(c, "more than 2^29 alignments per chromosome");
        if (n_bytes < 36) return fail(c, "BAM record bytes too short");
        c->read_mode = 2;
        hipStream_t s = c->stream;
        const uint64_t base = c->n_blob;                               // 16-byte aligned
        c->blob.reserve(base + (uint64_t)n_bytes + 32, s, true, base);
        h2d_staged(c, c->blob.p + base, records, (size_t)n_bytes);
        HIP_TRY(hipMemsetAsync(c->blob.p + base + n_bytes, 0, 32, s));
        upload(c, c->rec_off, rec_off, n);
        BamView B{c->blob.p, base, (uint64_t)n_bytes, c->rec_off.p};
        const int rc = push_record_view(c, B, n, name_id);
        if (rc) return rc;
        c->n_blob = (base + (uint64_t)n_bytes + 15) & ~15ull;
    } catch (std::string &e) { return fail(c, e); }
    return 0;
}

int lps_push_reads_device(lps_ctx *c, const lps_read_batch *b) {
    if (!c || !b) return -1;
    try {
        HIP_TRY(hipSetDevice(c->device));
        const size_t n = (size_t)b->n_reads; if (n == 0) return 0;
        if (c->read_mode >= 2) return fail(c, "lps_push_reads_device mixed with a BAM-record push in the same chromosome");
        if ((uint64_t)c->nR + n > 0x1fffffffull) return fail(c, "more than 2^29 alignments per chromosome");
        hipStream_t s = c->stream;
        uint64_t ends[6];                                                // first and last entry of the three offset arrays
        const uint64_t *offs[3] = {b->cigar_off, b->seq_off, b->qual_off};
        for (int k = 0; k < 3; ++k) { HIP_TRY(hipMemcpyAsync(&ends[2 * k], offs[k], 8, hipMemcpyDeviceToHost, s)); HIP_TRY(hipMemcpyAsync(&ends[2 * k + 1], offs[k] + n,
                8, hipMemcpyDeviceToHost, s)); }
        int32_t start_ends[2] = {0, 0};                                  // first and last start of the batch: the order ACROSS pushes is checked here, inside a batch by k_batch_check
        if (n) { HIP_TRY(hipMemcpyAsync(&start_ends[0], b->ref_start, 4, hipMemcpyDeviceToHost, s)); HIP_TRY(hipMemcpyAsync(&start_ends[1], b->ref_start + (n - 1), 4, hipMemcpyDeviceToHost, s)); }
        c->bam_err.reserve(2);
        HIP_TRY(hipMemsetAsync(c->bam_err.p, 0, 2 * sizeof(unsigned), s));
        hipLaunchKernelGGL(k_batch_check, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (long long)n, b->ref_start, b->l_qseq, b->name_id, b->cigar_off, b->seq_off, b->qual_off, c->bam_err.p);
        unsigned flags[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(flags, c->bam_err.p, sizeof flags, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (flags[0] & 1u) return fail(c, "offsets must be non-decreasing");
        if (flags[0] & 2u) return fail(c, "seq/qual shorter than l_qseq");
        if (flags[0] & 4u) return fail(c, "alignments must be coordinate-sorted");
        if (n && c->nR && start_ends[0] < c->last_start) return fail(c, "alignments must be coordinate-sorted (a batch starts before the end of the one pushed before it)");
        if (n) c->last_start = start_ends[1];
        if (ends[1] < ends[0] || ends[3] < ends[2] || ends[5] < ends[4]) return fail(c, "offsets must be non-decreasing");
        const uint64_t nc = ends[1] - ends[0], ns = ends[3] - ends[2], nq = ends[5] - ends[4];
        c->read_mode = 1;
        const size_t at = (size_t)c->nR;
        auto d2d = [&](auto &buf, const auto *src, size_t cnt, size_t where) {
            buf.reserve(where + cnt, s, true, where);
            if (cnt) HIP_TRY(hipMemcpyAsync(buf.p + where, src, cnt * sizeof(*src), hipMemcpyDeviceToDevice, s));
        };
        d2d(c->r_start, b->ref_start, n, at); d2d(c->r_lq, b->l_qseq, n, at); d2d(c->r_flag, b->flag, n, at); d2d(c->r_mapq, b->mapq, n, at); d2d(c->r_name, b->name_id, n, at);
        c->r_soff.reserve(at + n + 1, s, true, at); c->r_qoff.reserve(at + n + 1, s, true, at);
        const dim3 g((unsigned)((n + 1 + 255) / 256)), bl(256);
        hipLaunchKernelGGL(k_rebase_offsets, g, bl, 0, s, (long long)n + 1, b->seq_off, c->n_seq, c->r_soff.p + at);
        hipLaunchKernelGGL(k_rebase_offsets, g, bl, 0, s, (long long)n + 1, b->qual_off, c->n_qual, c->r_qoff.p + at);
        d2d(c->seq, b->seq + ends[2], (size_t)ns, (size_t)c->n_seq);
        d2d(c->qual, b->qual + ends[4], (size_t)nq, (size_t)c->n_qual);
        const int rc = append_cigar_chunks(c, at, n, b->cigar_off, b->cigar);     // the words' copy into the context, lane-chunked as it is made (synchronizes)
        if (rc) return rc;
        c->name_max = std::max(c->name_max, flags[1]);
        c->nR += (int)n; c->n_cig += nc; c->n_seq += ns; c->n_qual += nq;
        c->phase_valid = false;
    } catch (std::string &e) { return fail(c, e); }
    return 0;
}

int lps_bam_scan_range(lps_ctx *c, int64_t first_record_offset, int64_t end_offset, int32_t n_ref, int64_t *n_records) {
    if (!c || !n_records) return -1;
    if (!c->file_bytes || first_record_offset < 0 || end_offset < first_record_offset || (uint64_t)end_offset > c->file_bytes) return fail(c, "lps_bam_scan: no inflated BAM resident (lps_bgzf_load) or bad offsets");
    try {
        HIP_TRY(hipSetDevice(c->device));
        hipStream_t s = c->stream; c->bam_err.reserve(1); c->scan_nout.reserve(1);
        uint64_t n = 0;
        const int rc = bam_scan_records(c->file.p, (uint64_t)first_record_offset, (uint64_t)end_offset, n_ref, c->rcand, c->wg_cnt, c->wg_off, c->temp, c->temp_bytes,
                c->bam_err.p, c->scan_nout.p, &n, s);
        c->n_rec_all = 0; c->names_ready = false;
        if (rc == -3 || first_record_offset == end_offset) { *n_records = 0; if (rc == -3 && first_record_offset != end_offset) return fail(c, "lps_bam_scan: no BAM record found in a non-empty range"); return 0; }
        if (rc) return fail(c, rc == -4 ? "lps_bam_scan: the BAM record chain is broken (corrupt file)" : "lps_bam_scan: stream too large");
        if (n > 0x7fffffffull) return fail(c, "lps_bam_scan: more than 2^31 records");
        c->r_tid_all.reserve(n, s); c->r_lname.reserve(n + 1, s); c->r_nameoff.reserve(n + 1, s);
        launch_bam_tid_lname(c->file.p, c->rcand.p, (uint32_t)n, c->r_tid_all.p, c->r_lname.p, s);
        HIP_TRY(hipMemsetAsync(c->r_lname.p + n, 0, sizeof(uint32_t), s));
        { const size_t need = GraphTemp::need((size_t)n + 1); if (need > c->temp_bytes) { c->temp.reserve(need, s); c->temp_bytes = need; } }
        exscan_u32(c->temp.p, c->temp_bytes, c->r_lname.p, c->r_nameoff.p, (size_t)n + 1, s);
        HIP_TRY(hipStreamSynchronize(s));
        c->n_rec_all = n; *n_records = (int64_t)n;
    } catch (std::string &e) { return fail(c, e); }
    return 0;
}

int lps_bam_scan(lps_ctx *c, int64_t first_record_offset, int32_t n_ref, int64_t *n_records) {
    if (!c) return -1;
    if (first_record_offset < 12) return fail(c, "lps_bam_scan: bad offset");
    return lps_bam_scan_range(c, first_record_offset, (int64_t)c->file_bytes, n_ref, n_records);
}

int lps_bam_record_tids(lps_ctx *c, int32_t *tid) {
    if (!c || !tid) return -1;
    try {
        HIP_TRY(hipSetDevice(c->device));
        if (c->n_rec_all) HIP_TRY(hipMemcpyAsync(tid, c->r_tid_all.p, c->n_rec_all * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    } catch (std::string &e) { return fail(c, e); }
    return 0;
}

int lps_bam_record_offsets(lps_ctx *c, int64_t first, int64_t count, uint64_t *rec_off) {
    if (!c || first < 0 || count < 0 || (uint64_t)(first + count) > c->n_rec_all || (count > 0 && !rec_off)) return fail(c, "lps_bam_record_offsets: range outside the scanned records");
    try {
        HIP_TRY(hipSetDevice(c->device));
        if (count) HIP_TRY(hipMemcpyAsync(rec_off, c->rcand.p + first, (size_t)count * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    } catch (std::string &e) { return fail(c, e); }
    return 0;
}

int lps_bam_names(lps_ctx *c, int64_t first, int64_t count, uint32_t *name_off, char *names, int64_t names_cap, int64_t *names_bytes) {
    if (!c || first < 0 || count < 0 || (uint64_t)(first + count) > c->n_rec_all || !names_bytes) return fail(c, "lps_bam_names: range outside the scanned records");
    try {
        HIP_TRY(hipSetDevice(c->device));
        hipStream_t s = c->stream;
        uint32_t lo = 0, hi = 0;
        HIP_TRY(hipMemcpyAsync(&lo, c->r_nameoff.p + first, sizeof lo, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(&hi, c->r_nameoff.p + first + count, sizeof hi, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        *names_bytes = (int64_t)(hi - lo);
        if (!names || !name_off) return 0;                              // size query
        if (names_cap < (int64_t)(hi - lo)) return fail(c, "lps_bam_names: buffer too small");
        if (!c->names_ready) {                                          // gather all names once
            uint32_t tot = 0;
            HIP_TRY(hipMemcpyAsync(&tot, c->r_nameoff.p + c->n_rec_all, sizeof tot, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            c->names_d.reserve((size_t)tot + 1, s);
            launch_bam_names(c->file.p, c->rcand.p, (uint32_t)c->n_rec_all, c->r_nameoff.p, c->names_d.p, s);
            c->names_ready = true;
        }
        HIP_TRY(hipMemcpyAsync(name_off, c->r_nameoff.p + first, (size_t)(count + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        if (hi > lo) HIP_TRY(hipMemcpyAsync(names, c->names_d.p + lo, (size_t)(hi - lo), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        for (int64_t i = 0; i <= count; ++i) name_off[i] -= lo;
    } catch (std::string &e) { return fail(c, e); }
    return 0;
}

int lps_push_bam_resident(lps_ctx *c, int64_t first, int64_t count, const uint32_t *name_id) {
    if (!c || first < 0 || count < 0 || (uint64_t)(first + count) > c->n_rec_all || (count > 0 && !name_id)) return fail(c, "lps_push_bam_resident: range outside the scanned records");
    try {
        HIP_TRY(hipSetDevice(c->device));
        if (count == 0) return 0;
        if (c->read_mode == 1 || c->read_mode == 2) return fail(c, "lps_push_bam_resident mixed with another kind of push in the same chromosome");
        if ((uint64_t)c->nR + (uint64_t)count > 0x1fffffffull) return fail(c, "more than 2^29 alignments per chromosome");
        c->cur_first = c->read_mode == 3 ? -2 : first; c->cur_count = count;   // -2: more than one resident push in this chromosome
        c->read_mode = 3;
        BamView B{c->file.p, 0, c->file_bytes, c->rcand.p + first};
        return push_record_view(c, B, (size_t)count, name_id);
    } catch (std::string &e) { return fail(c, e); }
    return 0;
}

// (the header walk itself - bgzf_parse_header, bgzf_walk_piece, bgzf_walk_parallel - is host-only code shared with the command line: lps_bgzf_walk.h)
static int bgzf_load_source(lps_ctx *c, const ZSource &src, int64_t n_bytes, int64_t *inflated_bytes, const InflateBlock *table = nullptr, int64_t n_table = 0);
static_assert(sizeof(lps_bgzf_block) == sizeof(InflateBlock) && offsetof(lps_bgzf_block, in_len) == offsetof(InflateBlock, in_len), "lps_bgzf_block is the kernel's block record");
// host only: needs no GPU and no context (a caller can walk the file while the HIP runtime is still coming up)
int lps_bgzf_walk_fd(int fd, int64_t offset, int64_t n_bytes, lps_bgzf_block **blocks, int64_t *n_blocks, int64_t *inflated_bytes) {
    if (fd < 0 || offset < 0 || n_bytes < 28 || !blocks || !n_blocks) return -1;
    const ZSource src{nullptr, fd, (uint64_t)offset}; std::vector<InflateBlock> blks; uint64_t utot = 0;
    if (!bgzf_walk_parallel(src, (uint64_t)n_bytes, blks, utot)) { blks.clear(); utot = 0; if (!bgzf_walk_piece(src, (uint64_t)n_bytes, 0, (uint64_t)n_bytes, blks, utot) || blks.empty()) return -2; }
    lps_bgzf_block *out = (lps_bgzf_block *)malloc(blks.size() * sizeof(lps_bgzf_block));
    if (!out) return -3;
    memcpy(out, blks.data(), blks.size() * sizeof(lps_bgzf_block));
    *blocks = out; *n_blocks = (int64_t)blks.size(); if (inflated_bytes) *inflated_bytes = (int64_t)utot;
    return 0;
}
void lps_bgzf_blocks_free(lps_bgzf_block *blocks) { free(blocks); }
int lps_bgzf_load_fd_blocks(lps_ctx *c, int fd, int64_t offset, int64_t n_bytes, const lps_bgzf_block *blocks, int64_t n_blocks, int64_t *inflated_bytes) {
    if (!c || fd < 0 || offset < 0 || n_bytes < 28 || !blocks || n_blocks <= 0) return fail(c, "lps_bgzf_load_fd_blocks: not a BGZF file");
    return bgzf_load_source(c, ZSource{nullptr, fd, (uint64_t)offset}, n_bytes, inflated_bytes, reinterpret_cast<const InflateBlock *>(blocks), n_blocks);
}
int lps_bgzf_load(lps_ctx *c, const uint8_t *bgzf, int64_t n_bytes, int64_t *inflated_bytes) {
    if (!c || !bgzf || n_bytes < 28) return fail(c, "lps_bgzf_load: not a BGZF file");
    return bgzf_load_source(c, ZSource{bgzf, -1, 0}, n_bytes, inflated_bytes);
}
int lps_bgzf_load_fd(lps_ctx *c, int fd, int64_t offset, int64_t n_bytes, int64_t *inflated_bytes) {
    if (!c || fd < 0 || offset < 0 || n_bytes < 28) return fail(c, "lps_bgzf_load_fd: not a BGZF file");
    return bgzf_load_source(c, ZSource{nullptr, fd, (uint64_t)offset}, n_bytes, inflated_bytes);
}
static int bgzf_load_source(lps_ctx *c, const ZSource &src, int64_t n_bytes, int64_t *inflated_bytes, const InflateBlock *table, int64_t n_table) {
    try {
        HIP_TRY(hipSetDevice(c->device));
        auto tnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double th0 = tnow();
        // The upload needs nothing but the byte count: it starts now, on the copy stream, and raises a device word behind every 64-MiB piece.  The
        // block headers are walked beside it (pieces on 8 threads); as soon as the table is known the inflate kernel is launched on the main stream,
        // where a wavefront waits for the bytes of its own 32 members only - upload and inflate overlap (LPS_BGZF_SERIAL=1: one after the other).
        static const bool serial = getenv("LPS_BGZF_SERIAL") != nullptr;
        hipStream_t s = c->stream, cs = serial ? c->stream : c->copy_stream; hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
        struct Events { hipEvent_t &a, &b, &c; ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); if (c) (void)hipEventDestroy(c); } } ev_guard{e0, e1, e2};   // every early return below releases them
        HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1)); HIP_TRY(hipEventCreate(&e2));
        c->zfile.reserve((uint64_t)n_bytes + 64, s);
        if (!c->up_mark) HIP_TRY(hipHostMalloc((void **)&c->up_mark, 64, hipHostMallocCoherent));   // coherent (uncached on the GPU side): the polling kernel sees the host's stores
        __atomic_store_n(c->up_mark, 0ull, __ATOMIC_RELEASE);
        HIP_TRY(hipStreamSynchronize(s));                                   // (the buffers exist before another stream writes them)
        HIP_TRY(hipMemsetAsync(c->zfile.p + n_bytes, 0, 64, cs));
        HIP_TRY(hipEventRecord(e0, cs));
        std::string up_err;
        std::thread uploader([&] { try { (void)hipSetDevice(c->device); h2d_staged(c, c->zfile.p, src, (size_t)n_bytes, cs, serial ? nullptr : c->up_mark); HIP_TRY(hipEventRecord(e1, cs)); }
                                   catch (std::string &e) { up_err = e; __atomic_store_n(c->up_mark, ~0ull, __ATOMIC_RELEASE); } });   // (a failed upload must not leave the kernel waiting)
        struct Join { std::thread &t; ~Join() { if (t.joinable()) t.join(); } } join_up{uploader};
        std::vector<InflateBlock> blks; uint64_t utot = 0; const uint64_t n = (uint64_t)n_bytes;
        if (table) {
            // a table the caller walked earlier (lps_bgzf_walk_fd): checked against the byte count, so that the kernel never reads or writes outside
            blks.assign(table, table + n_table);
            for (const InflateBlock &b : blks) {
                if (b.out_len > 65536u || b.out_off != utot || b.in_off < 18 || b.in_off + b.in_len + 8 > n) return fail(c, "lps_bgzf_load_fd_blocks: the block table does not fit the bytes");
                utot += b.out_len;
            }
        } else if (serial || !bgzf_walk_parallel(src, n, blks, utot)) {
            blks.clear(); utot = 0; uint64_t bad_at = 0;
            if (!bgzf_walk_piece(src, n, 0, n, blks, utot, &bad_at) || blks.empty()) {
                // the words for what is wrong at bad_at
                uint8_t h[64]; const size_t hl = (size_t)std::min<uint64_t>(sizeof h, n - bad_at); unsigned xlen = 0;
                if (bad_at >= n || !src.read(bad_at, hl, h)) return fail(c, bad_at >= n ? "lps_bgzf_load: trailing bytes after the last BGZF block" : "lps_bgzf_load: cannot read the file");
                if (hl < 18) return fail(c, "lps_bgzf_load: trailing bytes after the last BGZF block");
                if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) return fail(c, "lps_bgzf_load: not a BGZF block header");
                const uint64_t bsize = bgzf_parse_header(h, hl, bad_at, n, xlen);
                if (!bsize) return fail(c, "lps_bgzf_load: truncated BGZF block");
                return fail(c, "lps_bgzf_load: BGZF block larger than 64 KiB");
            }
        }
        if (blks.size() > 0x7fffffffull) return fail(c, "lps_bgzf_load: too many blocks");
        const double th1 = tnow();
        if (serial) { uploader.join(); if (!up_err.empty()) return fail(c, up_err); }
        c->file.reserve(utot + 64, s); c->zblk.reserve(blks.size(), s); c->bam_err.reserve(1);
        c->zscratch.reserve(bgzf_inflate_scratch_bytes((int)blks.size()), s);
        const double th2 = tnow();
        HIP_TRY(hipMemcpyAsync(c->zblk.p, blks.data(), blks.size() * sizeof(InflateBlock), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemsetAsync(c->bam_err.p, 0, sizeof(unsigned), s));
        if (!serial) {
            // The kernel is launched when the bytes of its first 16 384 members are in place (a quarter of the 65 536 that are resident at a time: about
            // 0.5 GB, 20 ms of upload).  Launched at once - the table may have been walked long before - every resident wavefront sits waiting, and the
            // upload beside them ran at half its rate (measured: 0.37 - 0.40 s instead of 0.20 s for 8.26 GB, on some boxes three times that); launched
            // after a whole round of 65 536 the kernel, which bounds the load at 35 GB/s, starts 60 ms later for nothing (12.4 GB: 0.70 against 0.66 s;
            // 4 096: 0.70 s, the upload slows down again).
            const char *fr_env = getenv("LPS_BGZF_TEST_FIRST_ROUND");                      // test hook: members the launch waits for
            const size_t fr = fr_env ? (size_t)std::max(1, atoi(fr_env)) : 16384;
            const unsigned long long first_round = blks.size() > fr ? blks[fr].in_off : n;
            while (__atomic_load_n(c->up_mark, __ATOMIC_ACQUIRE) < first_round) usleep(200);
        }
        HIP_TRY(hipEventRecord(e2, s));
        const char *to_env = getenv("LPS_BGZF_TEST_TIMEOUT_MS");                           // test hook: how long a wave waits for its bytes
        launch_bgzf_inflate(c->zfile.p, c->zblk.p, (int)blks.size(), c->file.p, c->bam_err.p, c->zscratch.p, s, serial ? nullptr : c->up_mark, n, to_env ? atof(to_env) : 5000.0);
        launch_bgzf_crc(c->zfile.p, c->zblk.p, (int)blks.size(), c->file.p, c->bam_err.p, s);
        HIP_TRY(hipMemsetAsync(c->file.p + utot, 0, 64, s));
        hipEvent_t e3 = nullptr; HIP_TRY(hipEventCreate(&e3)); struct One { hipEvent_t &a; ~One() { if (a) (void)hipEventDestroy(a); } } e3_guard{e3};
        HIP_TRY(hipEventRecord(e3, s));
        unsigned err = 0;
        HIP_TRY(hipMemcpyAsync(&err, c->bam_err.p, sizeof err, hipMemcpyDeviceToHost, s));
        if (uploader.joinable()) uploader.join();
        HIP_TRY(hipStreamSynchronize(cs)); HIP_TRY(hipStreamSynchronize(s));
        if (!up_err.empty()) return fail(c, up_err);
        c->bgzf_retried = false;
        if (err & LPS_INF_ERR_TIMEOUT) {
            // Some wavefront waited longer for its bytes than it is allowed to (a slow source: network storage, a file that fell out of the page cache,
            // another worker's upload on the same link) and left.  The upload itself is complete now - the uploader has been joined, its stream
            // synchronized - so the members are inflated again WITHOUT the watermark: slower than the overlap, never wrong, no failure for the caller.
            HIP_TRY(hipMemsetAsync(c->bam_err.p, 0, sizeof(unsigned), s));
            HIP_TRY(hipEventRecord(e2, s));
            launch_bgzf_inflate(c->zfile.p, c->zblk.p, (int)blks.size(), c->file.p, c->bam_err.p, c->zscratch.p, s);
            launch_bgzf_crc(c->zfile.p, c->zblk.p, (int)blks.size(), c->file.p, c->bam_err.p, s);
            HIP_TRY(hipMemsetAsync(c->file.p + utot, 0, 64, s));
            HIP_TRY(hipEventRecord(e3, s));
            HIP_TRY(hipMemcpyAsync(&err, c->bam_err.p, sizeof err, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            c->bgzf_retried = true;
            if (getenv("LPS_DEBUG")) fprintf(stderr, "[lps_bgzf_load] the inflate kernel outran the upload (timeout): inflated again after the upload\n");
        }
        HIP_TRY(hipEventElapsedTime(&c->bgzf_h2d_ms, e0, e1)); HIP_TRY(hipEventElapsedTime(&c->bgzf_inflate_ms, e2, e3));
        if (getenv("LPS_DEBUG")) fprintf(stderr, "[lps_bgzf_load] %zu blocks: header walk %.1f ms | device buffers %.1f ms | upload %.1f ms beside inflate + crc %.1f ms (from its launch; it waits for its bytes) | host wall %.1f ms\n", blks.size(), th1 - th0, th2 - th1, c->bgzf_h2d_ms, c->bgzf_inflate_ms, tnow() - th0);
        c->file_bytes = 0; c->n_rec_all = 0; c->names_ready = false;
        if (err) return fail(c, err & LPS_INF_ERR_DATA ? "lps_bgzf_load: corrupt deflate stream" : err & (LPS_INF_ERR_SIZE | LPS_INF_ERR_OVERRUN) ? "lps_bgzf_load: a block does not inflate to its ISIZE"
                                                                : "lps_bgzf_load: CRC32 mismatch in a BGZF block");
        c->file_bytes = utot;
        if (inflated_bytes) *inflated_bytes = (int64_t)utot;
    } catch (std::string &e) { return fail(c, e); }
    return 0;
}

int lps_bgzf_read(lps_ctx *c, int64_t offset, int64_t n, uint8_t *dst) {
    if (!c || !dst || offset < 0 || n < 0 || (uint64_t)(offset + n) > c->file_bytes) return fail(c, "lps_bgzf_read: range outside the inflated stream");
    try {
        HIP_TRY(hipSetDevice(c->device));
        if (n) HIP_TRY(hipMemcpyAsync(dst, c->file.p + offset, (size_t)n, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    } catch (std::string &e) { return fail(c, e); }
    return 0;
}

// two events for timing a stretch of the stream; destroyed when the scope is left, whichever way
struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    EventPair() { HIP_TRY(hipEventCreate(&a)); if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); a = nullptr; throw std::string("hipEventCreate failed"); } }
    ~EventPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    EventPair(const EventPair &) = delete; EventPair &operator=(const EventPair &) = delete;
};

int lps_bgzf_deflate(lps_ctx *c, int64_t offset, int64_t n_bytes, int64_t *out_bytes) {
    if (!c || !out_bytes || offset < 0 || n_bytes < 0 || (uint64_t)(offset + n_bytes) > c->file_bytes) return fail(c, "lps_bgzf_deflate: range outside the resident stream");
    try {
        HIP_TRY(hipSetDevice(c->device));
        hipStream_t s = c->stream; EventPair ev; hipEvent_t &e0 = ev.a, &e1 = ev.b;   // (released on every way out, a throw included)
        HIP_TRY(hipEventRecord(e0, s));
        c->dz_total = bgzf_deflate_device(c->file.p + offset, (uint64_t)n_bytes, c->dz_slots, c->dz_bytes, c->dz_tmp, c->dz_off, c->dz_packed, c->temp, c->temp_bytes, s);
        HIP_TRY(hipEventRecord(e1, s)); HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipEventElapsedTime(&c->dz_ms, e0, e1));
        *out_bytes = (int64_t)c->dz_total;
    } catch (std::string &e) { return fail(c, e); }
    return 0;
}

int lps_bgzf_deflate_host(lps_ctx *c, const uint8_t *bytes, int64_t n_bytes, int64_t *out_bytes) {
    if (!c || !out_bytes || n_bytes < 0 || (n_bytes && !bytes)) return -1;
    try {
        HIP_TRY(hipSetDevice(c->device));
        hipStream_t s = c->stream; EventPair ev; hipEvent_t &e0 = ev.a, &e1 = ev.b;   // (released on every way out, a throw included)
        c->dz_src.reserve((size_t)n_bytes + 64, s);
        h2d_staged(c, c->dz_src.p, bytes, (size_t)n_bytes);
        HIP_TRY(hipEventRecord(e0, s));
        c->dz_total = bgzf_deflate_device(c->dz_src.p, (uint64_t)n_bytes, c->dz_slots, c->dz_bytes, c->dz_tmp, c->dz_off, c->dz_packed, c->temp, c->temp_bytes, s);
        HIP_TRY(hipEventRecord(e1, s)); HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipEventElapsedTime(&c->dz_ms, e0, e1));
        *out_bytes = (int64_t)c->dz_total;
    } catch (std::string &e) { return fail(c, e); }
    return 0;
}

static int write_tagged_bgzf(lps_ctx *c, const uint8_t *status, const uint8_t *hp, const int32_t *ps, const int32_t *pq, const uint8_t *prefix, int64_t prefix_bytes,
        int64_t *out_bytes, int somatic_tags) {
    if (!c || !out_bytes || prefix_bytes < 0 || (prefix_bytes && !prefix)) return -1;
    if (c->read_mode != 3 || c->cur_first < 0) return fail(c, "lps_haplotag_write_bgzf / lps_somatic_write_bgzf: needs exactly one lps_push_bam_resident in this chromosome");
    const size_t n = (size_t)c->cur_count;
    if (n && (!status || !hp || !ps || !pq)) return -1;
    try {
        HIP_TRY(hipSetDevice(c->device));
        hipStream_t s = c->stream; EventPair ev; hipEvent_t &e0 = ev.a, &e1 = ev.b;   // (released on every way out, a throw included)
        upload(c, c->tg_status, status, n); upload(c, c->tg_hp, hp, n); upload(c, c->tg_ps, ps, n); upload(c, c->tg_pq, pq, n);
        c->tg_stream.reserve((size_t)prefix_bytes + 64, s); c->bam_err.reserve(1);
        if (prefix_bytes) HIP_TRY(hipMemcpyAsync(c->tg_stream.p, prefix, (size_t)prefix_bytes, hipMemcpyHostToDevice, s));
        HIP_TRY(hipEventRecord(e0, s));
        const int64_t total = bam_tag_stream(c->file.p, c->rcand.p + c->cur_first, (uint32_t)n, c->tg_status.p, c->tg_hp.p, c->tg_ps.p, c->tg_pq.p, somatic_tags, (uint64_t)prefix_bytes,
                                             c->tg_len, c->tg_off, c->tg_spans, c->tg_stream, c->temp, c->temp_bytes, c->bam_err.p, s);
        if (total < 0) return fail(c, "malformed auxiliary field in a BAM record");
        c->dz_total = bgzf_deflate_device(c->tg_stream.p, (uint64_t)total, c->dz_slots, c->dz_bytes, c->dz_tmp, c->dz_off, c->dz_packed, c->temp, c->temp_bytes, s);
        HIP_TRY(hipEventRecord(e1, s)); HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipEventElapsedTime(&c->dz_ms, e0, e1));
        *out_bytes = (int64_t)c->dz_total;
    } catch (std::string &e) { return fail(c, e); }
    return 0;
}
int lps_haplotag_write_bgzf(lps_ctx *c, const uint8_t *status, const uint8_t *hp, const int32_t *ps, const int32_t *pq, const uint8_t *prefix, int64_t prefix_bytes, int64_t *out_bytes) {
    return write_tagged_bgzf(c, status, hp, ps, pq, prefix, prefix_bytes, out_bytes, 0);
}
int lps_somatic_write_bgzf(lps_ctx *c, const uint8_t *status, const uint8_t *hp, const int32_t *ps, const int32_t *pq, const uint8_t *prefix, int64_t prefix_bytes, int64_t *out_bytes) {
    return write_tagged_bgzf(c, status, hp, ps, pq, prefix, prefix_bytes, out_bytes, 1);
}

int lps_bgzf_deflate_fetch(lps_ctx *c, uint8_t *dst, int64_t cap, double *kernel_ms) {
    if (!c || (c->dz_total && !dst) || cap < (int64_t)c->dz_total) return fail(c, "lps_bgzf_deflate_fetch: buffer too small");
    try {
        HIP_TRY(hipSetDevice(c->device));
        if (c->dz_total) HIP_TRY(hipMemcpyAsync(dst, c->dz_packed.p, (size_t)c->dz_total, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (kernel_ms) *kernel_ms = c->dz_ms;
    } catch (std::string &e) { return fail(c, e); }
    return 0;
}

int lps_bgzf_deflate_fetch_range(lps_ctx *c, int64_t offset, int64_t n, uint8_t *dst) {
    if (!c || offset < 0 || n < 0 || (uint64_t)(offset + n) > c->dz_total || (n && !dst)) return fail(c, "lps_bgzf_deflate_fetch_range: range outside the deflated result");
    try {
        HIP_TRY(hipSetDevice(c->device));
        if (n) HIP_TRY(hipMemcpyAsync(dst, c->dz_packed.p + offset, (size_t)n, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    } catch (std::string &e) { return fail(c, e); }
    return 0;
}
void *lps_host_alloc(size_t bytes) { void *p = nullptr; return hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? p : nullptr; }
void lps_host_free(void *p) { if (p) (void)hipHostFree(p); }

int lps_bgzf_retried(lps_ctx *c) { return c ? (c->bgzf_retried ? 1 : 0) : -1; }
int lps_bgzf_timings(lps_ctx *c, double *h2d_ms, double *inflate_ms) {
    if (!c) return -1;
    if (h2d_ms) *h2d_ms = c->bgzf_h2d_ms;
    if (inflate_ms) *inflate_ms = c->bgzf_inflate_ms;
    return 0;
}

// Clip::getCNVInterval (src/phase/PhasingGraph.cpp:1103-1227) replayed on the HOST from the sorted clip keys (pos << 1 | front/back): a serial state
// machine over the clipped positions - a lone GPU lane walked it at ~0.4 us per position (5 ms for a chr20 with pile-ups), a host core does it in
// microseconds while the GPU runs the stages that do not need the intervals.  The reference runs it twice on the same counts (Clip ctor +
// PhasingProcess.cpp:148), which appends the same intervals twice: here it runs once and the list is doubled by the caller.  Unbounded output.
// Every transition that can emit needs a position with >= 5 front or >= 5 back clips (push needs up >= 5; slowUp emits on down >= 5, or on
// down >= curr / 4 with curr > 20, i.e. again down >= 5): without one the walk is skipped.
static void replay_cnv(const unsigned long long *keys, size_t n, std::vector<int32_t> &start, std::vector<int32_t> &end) {
    start.clear(); end.clear();
    if (n == 0) return;
    {   // largest per-position FRONT / BACK count: below 5 nothing can be emitted
        unsigned best = 0, run = 0;
        for (size_t i = 0; i < n; ++i) { run = (i && keys[i] == keys[i - 1]) ? run + 1 : 1; if (run > best) best = run; }
        if (best < 5) return;
    }
    struct St { bool push = false, slowUp = false, slowDown = false; int curr = 0, reject = 0, pullDown = 0, slowDownCount = 0, candStart = -1, candEnd = -1;
                void reset() { *this = St(); }
                void threshold(int up) { reject = up; if (up >= 20) { pullDown = up / 2; slowDownCount = 5; } else if (up >= 10) { pullDown = up / 2; slowDownCount = up / 4; } else { pullDown = 5; slowDownCount = 2; } } } s;
    const int Area = 30000;
    size_t i = 0; bool sentinel_done = false; int last_up = 0, last_down = 0, last_pos = 0;
    while (true) {
        int pos, up = 0, down = 0;
        if (i < n) {
            pos = (int)(keys[i] >> 1);
            while (i < n && (int)(keys[i] >> 1) == pos) { if (keys[i] & 1) ++down; else ++up; ++i; }
            last_up = up; last_down = down; last_pos = pos;
        } else if (!sentinel_done) { pos = last_pos + Area; up = last_up; down = last_down; sentinel_done = true; }   // :1134
        else break;
        if (!s.push && !s.slowDown && !s.slowUp) {
            if (up >= 5 && s.curr == 0) { s.push = true; s.slowUp = false; s.slowDown = true; s.curr = up - down; s.candStart = pos; s.candEnd = pos + Area; s.threshold(up); }
            else if (up > down && s.curr == 0) { s.push = false; s.slowUp = true; s.slowDown = false; s.curr = up - down; s.candStart = pos; s.candEnd = pos + Area; }
        } else if (s.push && s.slowDown) {
            if (up > s.reject) { s.threshold(up); s.candStart = pos; s.candEnd = pos + Area; }
            s.curr = s.curr + up - down;
            if (s.curr > 30) s.candEnd = pos + Area;
            bool emitted = false;
            if (down >= s.pullDown) emitted = true;
            else if (s.curr <= s.slowDownCount && pos <= s.candEnd) emitted = true;
            if (emitted) { start.push_back(s.candStart); end.push_back(pos); s.reset(); }
            if (pos > s.candEnd || s.curr <= 0 || pos - s.candStart >= 200000) s.reset();
        } else if (s.slowUp) {
            if (s.curr > 20 ? down >= s.curr / 4 : down >= 5) { start.push_back(s.candStart); end.push_back(pos); s.reset(); }
            else if (up >= 5) { s.push = true; s.slowUp = false; s.slowDown = true; s.curr = up - down; s.candStart = pos; s.candEnd = pos + Area; s.threshold(up); }
            else {
                s.curr = s.curr + up - down;
                if (s.curr > 30) s.candEnd = pos + Area;
                if (pos > s.candEnd || s.curr <= 0 || pos - s.candStart >= 200000) s.reset();
            }
        }
    }
}

static int bits_for(unsigned long long n) { int b = 1; while ((1ull << b) < n) ++b; return b; }

// stage boundaries on the stream.  Every recorded event drains the queue for a moment (~3.5 us): a phase run records them all only when the
// caller asked for the per-stage breakdown (lps_set_stage_timing level 2), else just the two around the extraction kernel (level 1) or none.
static void mark(lps_ctx *c, int st) {
    static const bool dbg = getenv("LPS_DEBUG_SYNC") != nullptr;        // diagnostic: drain the stream at every stage boundary and say which stage comes next
    if (dbg) { const hipError_t e = hipStreamSynchronize(c->stream); fprintf(stderr, "[lps] before stage %s: %s\n", kStageNames[st], hipGetErrorString(e)); fflush(stderr); }
    if (c->in_phase && (c->timing_level == 0 || (c->timing_level == 1 && st != ST_EXTRACT && st != ST_GROUPS))) return;
    HIP_TRY(hipEventRecord(c->ev[st], c->stream)); c->ev_used[st] = true;
}

static VarView var_view(lps_ctx *c) {
    VarView V{};
    V.n = c->nV; V.pos = c->v_pos.p; V.ref0 = c->v_ref0.p; V.alt0 = c->v_alt0.p; V.ref_len = c->v_rl.p; V.alt_len = c->v_al.p;
    V.danger = c->v_danger.p; V.hpoly = c->v_hpoly.p; V.erased = c->v_erased.p; V.hp1_is_alt = c->has_hap ? c->v_hp1.p : nullptr; V.phase_set = c->v_ps.p;
    V.somatic_role = (c->has_somatic || c->has_tkind) ? c->v_role.p : nullptr; V.derive_hp = c->has_somatic ? c->v_derive.p : nullptr;
    V.tumor_kind = c->has_tkind ? c->v_tkind.p : nullptr;
    V.ref = c->ref.p; V.ref_len_eff = c->ref_len_eff; V.last_pos = c->last_pos;
    V.n_bucket = (int)(((long long)c->last_pos + 1) >> LPS_BUCKET_SHIFT) + 1; V.bucket = c->v_bucket.p; V.rec = c->v_rec.p;
    return V;
}
static ReadView read_view(lps_ctx *c) {
    ReadView R{};
    R.n = c->nR; R.ref_start = c->r_start.p; R.l_qseq = c->r_lq.p; R.flag = c->r_flag.p; R.mapq = c->r_mapq.p; R.name_id = c->r_name.p;
    R.seq_off = c->r_soff.p; R.qual_off = c->r_qoff.p;
    if (c->read_mode == 2) R.seq = R.qual = c->blob.p; else if (c->read_mode == 3) R.seq = R.qual = c->file.p; else { R.seq = c->seq.p; R.qual = c->qual.p; }
    R.v0 = c->r_v0.p; R.cigp = c->cigar.p; R.cp_off = c->cp_off.p; R.cp_n = c->cp_n.p;
    return R;
}

// Stages after the overlap filter and the clip statistics.  with_cnv = false is the first, speculative run (no CNV interval assumed); with_cnv =
// true runs them again with the CNV mismatch filter after clearing what they accumulate (tail of the zero pool, late counters).
static int run_late(lps_ctx *c, bool with_cnv) {
    hipStream_t s = c->stream; const lps_params &P = c->P; const int nR = c->nR, nV = c->nG, A = P.connect_adjacent;   // nV here: rows of the table the graph runs on (SNP rows, or the union with SV / MOD rows)
    const GraphView &G = c->G;
    if (with_cnv) {
        HIP_TRY(hipMemsetAsync(c->zpool.p + c->z_late_off, 0, c->z_late_bytes, s));
        HIP_TRY(hipMemsetAsync(&c->d_cnt->n_pairs, 0, offsetof(LpsCounters, arena_max) - offsetof(LpsCounters, n_pairs), s));
        HIP_TRY(hipMemsetAsync(c->clip_stats.p + 2, 0, 2 * sizeof(unsigned), s));
    }
    // ---- a9 CNV mismatch filter: only when intervals exist (the count arrived while the kernels above were running)
    mark(c, ST_CNV);
    c->cnv_skipped = !with_cnv;
    if (with_cnv) {
        const size_t K = c->h_cnv_start.size();
        const size_t need = (4 * K + 4) * sizeof(int32_t);             // pinned staging [n_cnv | start x2 | end x2]: every interval twice, as the reference's cnvVec holds them
        if (need > c->h_cnv_pin_bytes) { if (c->h_cnv_pin) HIP_TRY(hipHostFree(c->h_cnv_pin)); c->h_cnv_pin = nullptr; c->h_cnv_pin_bytes = need * 2; HIP_TRY(hipHostMalloc((void **)&c->h_cnv_pin, c->h_cnv_pin_bytes)); }
        int32_t *two = c->h_cnv_pin + 4;
        c->h_cnv_pin[0] = (int32_t)(2 * K);
        for (size_t i = 0; i < K; ++i) { two[i] = two[K + i] = c->h_cnv_start[i]; two[2 * K + i] = two[3 * K + i] = c->h_cnv_end[i]; }
        c->cnv_start.reserve(4 * K + 4); c->cnv_end.carve(c->cnv_start.p + 2 * K, 2 * K);
        HIP_TRY(hipMemcpyAsync(c->cnv_start.p, two, 4 * K * sizeof(int32_t), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(&c->d_cnt->n_cnv, c->h_cnv_pin, sizeof(unsigned), hipMemcpyHostToDevice, s));
        c->agg_sum.reserve((size_t)nV * 2 + 2); c->agg_cnt.reserve((size_t)nV * 2 + 2); c->miss.reserve(nV + 1);
        c->cnv_flag.reserve(nR + 1); c->cnv_idx.reserve(nR + 1); c->cnv_list.reserve(nR + 1); c->cnv_nlist.reserve(4);
        c->cnv_fn.reserve(nR + 1); c->cnv_pre.reserve(nR + 1);
        CnvScratch W{c->cnv_flag.p, c->cnv_idx.p, c->cnv_list.p, c->cnv_nlist.p, c->cnv_fn.p, c->cnv_pre.p};
        launch_cnv_filter(c->d_cnt, nR, nV, c->rows.p, c->deleted.p, c->obs.p, c->g_vpos, c->cnv_start.p, c->cnv_end.p, c->agg_sum.p, c->agg_cnt.p, c->miss.p, W,
                c->var_del2.p, c->temp.p, c->temp_bytes, s);
    }
    // ---- a10 node set (numbered before the host looked at the counters when no CNV filter was expected), graph view of the rows, node-major lists,
    //      merged rows of reads with several alignments
    mark(c, ST_NODES);
    if (!c->scan_done) launch_var_scan(G, s);
    c->scan_done = false;
    launch_graph_rows(G, P.base_quality, c->a_bits, c->key64, c->h_cnt.n_multi, s);
    // ---- a11/a12 edges
    mark(c, ST_EDGES);
    launch_edges(G, c->m_bits, c->a_bits, c->key64, P.edge_weight, P.edge_threshold, s);
    // ---- a13 vote scan
    mark(c, ST_SCAN);
    launch_vote_scan(c->d_cnt, nV, c->nodes.p, c->g_vpos, c->erec.p, A, P.distance, c->hp_v.p, c->blk_v.p, c->st_b.p, c->st_e.p, c->seg_i32.p, c->clip_stats.p + 2,
            c->hp.p, c->block.p, c->bmulti.p, c->nX ? 2 : 1, s);
    // ---- a14/a15 read correction + export
    mark(c, ST_CORR);
    launch_correction(G, c->block.p, c->bmulti.p, c->hp.p, c->nstate.p, P.read_confidence, P.snp_confidence, c->cnt4.p, c->out_ps.p, c->out_gt.p, s);
    mark(c, ST_D2H);
    return 0;                                                          // counters + statistics leave with the result (enqueue_result_copy)
}

static int run_phase(lps_ctx *c) {
    const lps_params &P = c->P; hipStream_t s = c->stream;
    const int nR = c->nR, nV = c->nV, A = P.connect_adjacent;
    const int nG = c->nG = c->nV + c->nX;                                      // rows of the table the stages after the extraction run on
    c->g_vpos = c->nX ? c->u_pos.p : c->v_pos.p;
    // read names: the stages below index by name, so the ids must be dense.  The CLI and the bench hand over ranks; anything sparser is ranked here
    const bool dense = (unsigned long long)c->name_max < 4ull * (unsigned long long)nR + 65536ull;
    c->name_cap = dense ? (size_t)c->name_max + 1 : (size_t)nR;
    // ---- capacities
    if (c->obs_capacity == 0) c->obs_capacity = std::max<unsigned long long>(64 * 1024, (unsigned long long)nR * 64);
    for (int attempt = 0; attempt < 3; ++attempt) {
        const int n_blocks = (nR + 3) / 4;                                    // k_extract_phase: a wave of 4 alignments per workgroup
        const int n_arenas = std::max(1, std::min(LPS_ARENAS, n_blocks));
        c->obs_capacity = (c->obs_capacity + n_arenas - 1) / n_arenas * n_arenas;
        const unsigned long long cap_main = c->obs_capacity, arena_size = cap_main / n_arenas, tail_size = cap_main / 4 + 4096;
        const unsigned long long cap = cap_main + tail_size;
        if (cap > 0xffffffffull) { c->err = "observation arena exceeds 2^32 slots"; return -8; }
        c->rows.reserve(nR + 4); c->redo_list.reserve((size_t)nR / 4 + 4);
        c->g_cnt.reserve(nR + 1);
        c->obs.reserve(cap); c->g_pack.reserve(cap); c->g_rank.reserve(cap);
        c->t_node.reserve(tail_size + 64); c->t_flag.reserve(tail_size + 64); c->t_src.reserve(tail_size + 64);
        c->clip_capacity = (size_t)EXT_CLIPS * ((nR + 3) / 4) + (size_t)4 * nR + 64;   // clip events: every extraction job's own slots, then room for what the general walker appends
        c->clip_ev.reserve(c->clip_capacity);
        c->clip_keys.reserve(c->clip_capacity); c->clip_keys_s.reserve(c->clip_capacity);
        c->name_link.reserve(nR + 1); c->mm_r.reserve(nR + 1); c->stack.reserve(nR + 1);
        c->mg_start.reserve(nR / 2 + 2); c->mg_cnt.reserve(nR / 2 + 2); c->mg_name.reserve(nR / 2 + 2); c->mg_plan.reserve(nR / 2 + 2);
        c->mrow_off.reserve(c->name_cap + 1); c->mrow_cnt.reserve(c->name_cap + 1);
        c->node_of.reserve(nG + 1); c->var_off.reserve(nG + 1); c->node_off.reserve(nG + 2); c->node_cap.reserve(nG + 2); c->node_end.reserve(nG + 2);
        c->bsum.reserve(3 * ((size_t)nG / 1024 + 2));
        c->nodes.reserve(nG + 1); c->block.reserve(nG + 1);
        c->hp.reserve(nG + 1);
        c->erec.reserve((size_t)nG * A + 256);
        c->hp_v.reserve(2 * ((size_t)nG + 64)); c->blk_v.reserve(2 * ((size_t)nG + 64)); c->seg_i32.reserve(8 * (size_t)scan_segments(nG) + 16); c->node_pairs.reserve(nG + 1); c->nstate.reserve(nG + 1);
        c->st_b.reserve(scan_state_bytes(nG)); c->st_e.reserve(scan_state_bytes(nG)); c->edge.reserve((size_t)nG * A * 4 + 16);
        // everything that has to start a run as zeros sits in ONE allocation cleared by one fill (a dozen separate fills cost ~4 us each)
        size_t zbytes = 0;
        auto zslot = [&](size_t bytes) { const size_t at = zbytes; zbytes += (bytes + 255) & ~(size_t)255; return at; };
        const size_t z_arena = zslot(LPS_ARENAS * 8 * sizeof(unsigned long long)), z_del = zslot((size_t)nR + 1), z_stats = zslot(4 * sizeof(unsigned)), z_ctab = zslot((size_t)(2u << LPS_CLIP_TAB_BITS) * 4),
                     z_vc = zslot(((size_t)nG + 1) * 4), z_vd = zslot(((size_t)nG + 1) * 4), z_nh = zslot((c->name_cap + 2) * 4),
                     z_ps = zslot(((size_t)nG + 1) * 4), z_gt = zslot((size_t)nG + 1), z_vd2 = zslot(((size_t)nG + 1) * 4), z_vtk = zslot(((size_t)nG + 1) * 4),
                     z_bm = zslot((size_t)nG + 1), z_c4 = zslot(((size_t)nG * 4 + 4) * 4);

        c->zpool.reserve(zbytes);
        c->z_late_off = z_ps; c->z_late_bytes = zbytes - z_ps;       // what the stages after the overlap filter need zeroed (see run_late)
        c->arena_ctr.carve(c->zpool.p + z_arena, LPS_ARENAS * 8); c->out_ps.carve(c->zpool.p + z_ps, (size_t)nG + 1); c->out_gt.carve(c->zpool.p + z_gt, (size_t)nG + 1);
        c->deleted.carve(c->zpool.p + z_del, (size_t)nR + 1); c->clip_stats.carve(c->zpool.p + z_stats, 4); c->clip_tab.carve(c->zpool.p + z_ctab, (size_t)(2u << LPS_CLIP_TAB_BITS));
        c->var_cnt.carve(c->zpool.p + z_vc, (size_t)nG + 1); c->var_del.carve(c->zpool.p + z_vd, (size_t)nG + 1); c->name_head.carve(c->zpool.p + z_nh, c->name_cap + 2);
        c->var_del2.carve(c->zpool.p + z_vd2, (size_t)nG + 1); c->vtype_key.carve(c->zpool.p + z_vtk, (size_t)nG + 1);
        c->bmulti.carve(c->zpool.p + z_bm, (size_t)nG + 1); c->cnt4.carve(c->zpool.p + z_c4, (size_t)nG * 4 + 4);
        const size_t need = GraphTemp::need((size_t)std::max(nR, nG) + 1);
        if (need > c->temp_bytes) { c->temp.reserve(need); c->temp_bytes = need; }

        for (auto &u : c->ev_used) u = false;
        HIP_TRY(hipEventRecord(c->ev_begin, s));
        HIP_TRY(hipMemsetAsync(c->d_cnt, 0, sizeof(LpsCounters), s));
        HIP_TRY(hipMemsetAsync(c->zpool.p, 0, zbytes, s));
        // ---- a4/a5/a6 variant table prep
        c->v_bucket.reserve((size_t)(((long long)c->last_pos + 1) >> LPS_BUCKET_SHIFT) + 8);
        c->v_rec.reserve((size_t)nV + 1);
        c->r_v0.reserve((size_t)nR + 1);
        VarView V = var_view(c); ReadView R = read_view(c);
        mark(c, ST_PREP);
        launch_variant_prep(V, P.is_ont, c->v_bucket.p, c->v_rec.p, s, R.ref_start, R.n, c->r_v0.p);
        if (!dense) {
            c->name_keys.reserve(nR + 1); c->name_keys_s.reserve(nR + 1); c->head.reserve(nR + 1); c->gidx.reserve(nR + 1); c->name_dense.reserve(nR + 1);
            launch_dense_names(nR, c->r_name.p, c->name_max, c->name_keys.p, c->name_keys_s.p, c->head.p, c->gidx.p, c->name_dense.p, c->temp.p, c->temp_bytes, s);
        }
        c->name_p = dense ? c->r_name.p : c->name_dense.p;
        // ---- a1/a2/a3 extraction; with no SV / MOD rows every observation is counted (and ranked inside its variant's list) right there
        ObsView O{c->rows.p, c->obs.p, arena_size, c->arena_ctr.p, n_arenas, c->nX ? c->x_snp_u.p : nullptr};
        ClipView C{c->clip_ev.p, c->clip_stats.p, (unsigned)c->clip_capacity, (unsigned)(EXT_CLIPS * ((nR + 3) / 4))};            // clip_stats[0]: events appended by the general walker, [1]: jobs queued for it (zero pool)
        mark(c, ST_EXTRACT);
        launch_extract_phase(V, R, O, C, P.mapping_quality, c->d_cnt, c->redo_list.p, c->clip_stats.p + 1, c->nX ? nullptr : c->var_cnt.p, c->var_del.p, s);
        {   // diagnostic (profiles/extract_only.py: timing experiments on builds whose extraction is incomplete on purpose): stop here, rc 77
            static const bool extract_only = getenv("LPS_EXTRACT_ONLY") != nullptr;
            if (extract_only) {
                HIP_TRY(hipEventRecord(c->ev_end, s)); HIP_TRY(hipStreamSynchronize(s));
                float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, c->ev[ST_EXTRACT], c->ev_end)); c->tm.ms_kernel[ST_EXTRACT] = ms;
                { unsigned st[2] = {0, 0}; HIP_TRY(hipMemcpy(st, c->clip_stats.p, sizeof st, hipMemcpyDeviceToHost)); static int said = 0; if (!said++) fprintf(stderr, "[extract only] clip events %u, jobs sent to the general walker %u of %d\n", st[0], st[1], (nR + 3) / 4); }
                return 77;
            }
        }
        // ---- SV / MOD rows: served against each alignment's CIGAR, merged into its row; every observation leaves in union indices
        if (c->nX) {
            ExtraView X{c->nX, c->x_pos.p, c->x_info.p, c->x_kind.p, c->x_u.p, c->x_snp_u.p, c->x_moff.p, c->x_mname.p, c->x_mflag.p, c->sv_window, c->sv_threshold, c->x_rec.p, c->x_mpack.p};
            c->x_x0.reserve((size_t)nR + 1); c->x_redo.reserve((size_t)nR + 2);
            launch_extra_merge(V, R, O, X, c->x_x0.p, c->x_redo.p + 1, c->x_redo.p, P.mapping_quality, c->d_cnt, s);   // (x_redo[0]: how many alignments were queued for the general walker)
        }
        GraphView &G = c->G;
        G = GraphView{};
        G.n_reads = nR; G.n_var = nG; G.A = A; G.rows = c->rows.p; G.obs = c->obs.p; G.deleted = c->deleted.p; G.vpos = c->g_vpos; G.name = c->name_p;
        G.name_head = c->name_head.p; G.name_link = c->name_link.p; G.mm_r = c->mm_r.p; G.stack = c->stack.p; G.mg_start = c->mg_start.p; G.mg_cnt = c->mg_cnt.p; G.mg_name = c->mg_name.p; G.mg_plan = c->mg_plan.p;
        G.var_cnt = c->var_cnt.p; G.var_del = c->var_del.p; G.var_del2 = c->var_del2.p; G.vtype_key = c->vtype_key.p; G.bsum = c->bsum.p;
        G.node_of = c->node_of.p; G.var_off = c->var_off.p; G.nodes = c->nodes.p; G.node_off = c->node_off.p; G.node_cap = c->node_cap.p; G.node_end = c->node_end.p;
        G.g_pack = c->g_pack.p; G.g_rank = c->g_rank.p; G.g_cnt = c->g_cnt.p; G.mrow_off = c->mrow_off.p; G.mrow_cnt = c->mrow_cnt.p;
        G.t_node = c->t_node.p; G.t_flag = c->t_flag.p; G.t_src = c->t_src.p; G.tail_lo = cap_main; G.tail_size = tail_size;
        G.edge = c->edge.p; G.erec = c->erec.p; G.node_pairs = c->node_pairs.p; G.cnt = c->d_cnt;
        // ---- read names linked into lists, clip keys, reservation totals: one launch; then the groups of several alignments and their overlap filter (a8)
        mark(c, ST_GROUPS);
        launch_names(G, C, c->clip_keys.p, c->arena_ctr.p, arena_size, c->clip_tab.p, s);
        mark(c, ST_OVERLAP);
        launch_groups(G, P.overlap_threshold, /*counted=*/!c->nX, s);
        if (c->nX) launch_count_ranks(G, s);
        // ---- the counters (sizes of what follows, errors) go to the host; the node numbering does not need them and keeps the GPU busy meanwhile
        //      (unless the CNV filter is expected to drop observations first)
        HIP_TRY(hipMemcpyAsync(c->h_cnt_pin, c->d_cnt, sizeof(LpsCounters), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipEventRecord(c->ev_cnv, s));
        c->scan_done = !c->cnv_expect;
        if (c->scan_done) launch_var_scan(G, s);
        HIP_TRY(hipEventSynchronize(c->ev_cnv));
        c->h_cnt = *c->h_cnt_pin;
        if (c->h_cnt.err & LPS_ERR_BAD_CIGAR) { c->err = "alignment find unsupported CIGAR operation"; return -2; }
        if (c->h_cnt.err & LPS_ERR_CLIP_OVERFLOW) { c->err = "clip event buffer overflow"; return -3; }
        if (c->h_cnt.err & LPS_ERR_OBS_OVERFLOW) { c->obs_capacity = (unsigned long long)n_arenas * (c->h_cnt.arena_max + c->h_cnt.arena_max / 4 + 1024); continue; }
        // keys of the node-major lists: (read name, index in the read's merged row).  The index field is sized by what the rows actually hold - the
        // longest row times the most alignments under one name - so that for real data name rank + index fit ONE 32-bit word (half the entry
        // bytes, one compare per rank step in k_edges); 64-bit keys otherwise
        c->m_bits = bits_for((unsigned long long)c->name_cap + 1);
        c->a_bits = bits_for((unsigned long long)std::max(1u, c->h_cnt.max_row) * (unsigned long long)std::max(1u, c->h_cnt.max_group) + 1);
        c->key64 = c->m_bits + c->a_bits > 31;
        if (c->m_bits + c->a_bits > 63) { c->err = "sort key overflow"; return -4; }
        c->late_n_keys = c->h_cnt.obs_total;
        // (32-bit keys: the unsorted lists are {key, slot} entries of 8 bytes, nvals is not used; the lists k_edges sorts keep keys and slots apart)
        c->nkeys.reserve(c->late_n_keys + 2); c->nkeys_s.reserve(c->late_n_keys / (c->key64 ? 1 : 2) + 2); c->nvals.reserve(c->key64 ? c->late_n_keys + 1 : 1); c->nvals_s.reserve(c->late_n_keys + 1);
        G.ukeys = c->nkeys.p; G.skeys = c->nkeys_s.p; G.uvals = c->nvals.p; G.svals = c->nvals_s.p;
        // places in the lists that no row will ever fill (observations of a job that went to the general walker after they were counted: malformed records)
        if (c->h_cnt.n_abandoned) HIP_TRY(hipMemsetAsync(c->nkeys.p, 0xff, (size_t)c->late_n_keys * 8, s));
        // ---- a7 clips -> CNV intervals: the keys are sorted here and travel to the host (pinned), which replays the state machine (replay_cnv) -
        //      unless no (position, front / back) key occurs five times (clip_mult, k_name_link's count-min bound): then no interval can be emitted
        //      (replay_cnv's own first test) and the sort (nine launches), the copy and the replay are skipped
        mark(c, ST_CLIP);
        const size_t nk = c->h_cnt.n_clips;
        c->late_cap_main = cap_main; c->late_tail = tail_size;
        c->h_ub_hazard = nk == 0 ? 1u : 0u;                               // reference: UB on an empty ClipCount (PhasingGraph.cpp:1134)
        static const bool always_sort = getenv("LPS_CLIP_ALWAYS_SORT") != nullptr;       // (test hook: the sorted path on inputs that would skip it)
        c->clips_sorted = false;
        if (c->h_cnt.clip_mult < 5u && !always_sort) {
            c->h_cnv_start.clear(); c->h_cnv_end.clear();
            return run_late(c, false);
        }
        { const size_t need_k = GraphTemp::need((size_t)c->h_cnt.n_clips + 1); if (need_k > c->temp_bytes) { c->temp.reserve(need_k, s); c->temp_bytes = need_k; } }   // (more clip events than alignments: every read clipped at both ends)
        launch_clip_sort(c->h_cnt.n_clips, c->clip_keys.p, c->clip_keys_s.p, c->temp.p, c->temp_bytes, s);
        c->clips_sorted = true;
        if (nk > c->h_clip_cap) { if (c->h_clip_keys) HIP_TRY(hipHostFree(c->h_clip_keys)); c->h_clip_keys = nullptr; c->h_clip_cap = nk + nk / 2 + 1024; HIP_TRY(hipHostMalloc((void **)&c->h_clip_keys, c->h_clip_cap * sizeof(unsigned long long))); }
        HIP_TRY(hipEventRecord(c->ev_sorted, s));
        HIP_TRY(hipStreamWaitEvent(c->copy_stream, c->ev_sorted, 0));      // the copy rides on its own stream: the late stages do not queue behind it
        if (nk) HIP_TRY(hipMemcpyAsync(c->h_clip_keys, c->clip_keys_s.p, nk * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->copy_stream));
        HIP_TRY(hipEventRecord(c->ev_clip, c->copy_stream));
        // ---- everything after.  Usually the clips give no CNV interval, so the late stages are enqueued on that guess and the host replays the state
        //      machine meanwhile; lps_phase_chromosome runs them again with the filter if the guess was wrong.  When the previous run of this ctx did
        //      have intervals the guess is not made: the host waits for the keys (a bubble of one small copy), replays, and the late stages run once.
        if (c->cnv_expect) {
            HIP_TRY(hipEventSynchronize(c->ev_clip));
            replay_cnv(c->h_clip_keys, nk, c->h_cnv_start, c->h_cnv_end);
            return run_late(c, !c->h_cnv_start.empty());
        }
        const int rc = run_late(c, false);
        HIP_TRY(hipEventSynchronize(c->ev_clip));                          // arrived long ago: the late stages are still running
        replay_cnv(c->h_clip_keys, nk, c->h_cnv_start, c->h_cnv_end);
        return rc;
    }
    c->err = "observation buffer kept overflowing";
    return -5;
}

// out_ps and out_gt are neighbours in the zero pool: ONE kernel writes both, the counters and the scan statistics straight into pinned host memory
// (coalesced 16-byte stores over PCIe) instead of three copies on the DMA queue, each with its fixed cost.
__global__ void k_result_out(const uint4 *src, uint4 *dst, size_t n16, const LpsCounters *cnt, LpsCounters *cnt_out, const unsigned *stats, unsigned *stats_out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) dst[i] = src[i];
    if (i == 0) { *cnt_out = *cnt; stats_out[0] = stats[0]; stats_out[1] = stats[1]; stats_out[2] = stats[2]; stats_out[3] = stats[3]; }
}
static size_t enqueue_result_copy(lps_ctx *c) {
    const size_t span = (size_t)((uint8_t *)c->out_gt.p - (uint8_t *)c->out_ps.p) + (size_t)c->nG;
    const size_t n16 = (span + 15) / 16;                                   // the zero pool's slots are padded to 256 bytes: the rounded span stays inside it
    if (n16 * 16 > c->h_res_bytes) { if (c->h_res) HIP_TRY(hipHostFree(c->h_res)); c->h_res = nullptr; c->h_res_bytes = n16 * 16 + span / 4 + 4096; HIP_TRY(hipHostMalloc((void **)&c->h_res, c->h_res_bytes)); }
    hipLaunchKernelGGL(k_result_out, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, c->stream, (const uint4 *)c->out_ps.p, (uint4 *)c->h_res, n16, c->d_cnt,
            c->h_cnt_pin, c->clip_stats.p, c->h_stats_pin);
    return span;
}
static void deliver_result(lps_ctx *c, lps_phase_result *out) {
    const int32_t *ps = (const int32_t *)c->h_res; const uint8_t *gt = c->h_res + ((uint8_t *)c->out_gt.p - (uint8_t *)c->out_ps.p);
    if (!c->nX) { memcpy(out->phase_set, ps, (size_t)c->nV * sizeof(int32_t)); memcpy(out->gt, gt, (size_t)c->nV); return; }
    // the result covers the union of the three tables: the SNP rows go to the caller, the whole of it stays for lps_get_extra_result
    c->h_res_ps_u.assign(ps, ps + c->nG); c->h_res_gt_u.assign(gt, gt + c->nG);
    for (int i = 0; i < c->nV; ++i) { out->phase_set[i] = ps[c->h_snp_u[i]]; out->gt[i] = gt[c->h_snp_u[i]]; }
}

int lps_debug_set_obs_capacity(lps_ctx *c, int64_t slots) {
    if (!c || slots < 0) return -1;
    c->obs_capacity = (unsigned long long)slots;
    return 0;
}

int lps_set_stage_timing(lps_ctx *c, int level) {
    if (!c || level < 0 || level > 2) return -1;
    c->timing_level = level;
    return 0;
}

// what the call spent growing device buffers (host time; the first call on a chromosome sizes them, later calls find them in place)
struct AllocScope { lps_ctx *c; double a0; explicit AllocScope(lps_ctx *x) : c(x), a0(g_lps_alloc_ms) {} ~AllocScope() { c->alloc_ms = g_lps_alloc_ms - a0; } };
double lps_alloc_ms(lps_ctx *c) { return c ? c->alloc_ms : -1.0; }
int lps_phase_chromosome(lps_ctx *c, lps_phase_result *out) {
    if (!c || !out) return -1;
    AllocScope alloc_scope(c);
    try {
        HIP_TRY(hipSetDevice(c->device));
        if (out->n != c->nV) return fail(c, "lps_phase_result.n must equal the variant table size");
        memset(out->phase_set, 0, (size_t)out->n * sizeof(int32_t)); memset(out->gt, 0, (size_t)out->n);
        c->phase_valid = false; c->h_res_ps_u.clear(); c->h_res_gt_u.clear();
        if (c->nV == 0 || c->nR == 0) { c->h_cnt = LpsCounters{}; memset(c->h_stats, 0, sizeof c->h_stats); c->h_cnv_start.clear(); c->h_cnv_end.clear(); c->phase_valid = c->nV != 0; return 0; }   // (the dump entries then report an empty graph, not the previous chromosome's)
        if (c->ref_len_eff == 0) return fail(c, "lps_set_reference has not been called");
        c->in_phase = true;
        int rc = run_phase(c);
        c->in_phase = false;
        if (rc != 0) return rc;
        (void)enqueue_result_copy(c);
        HIP_TRY(hipEventRecord(c->ev_end, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->h_cnt = *c->h_cnt_pin; memcpy(c->h_stats, c->h_stats_pin, sizeof c->h_stats);
        if (c->cnv_skipped && !c->h_cnv_start.empty() && !(c->h_cnt.err & LPS_ERR_OBS_OVERFLOW)) {     // CNV intervals exist after all: late stages again, with the filter
            rc = run_late(c, true);
            if (rc != 0) return rc;
            (void)enqueue_result_copy(c);
            HIP_TRY(hipEventRecord(c->ev_end, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            c->h_cnt = *c->h_cnt_pin; memcpy(c->h_stats, c->h_stats_pin, sizeof c->h_stats);
        }
        if (c->h_cnt.err & LPS_ERR_OBS_OVERFLOW) { c->obs_capacity = c->obs_capacity * 2 + 64 * 1024; return lps_phase_chromosome(c, out); }
        if (c->h_cnt.err & LPS_ERR_KEY_RANGE) return fail(c, "more than 4 194 304 observations of one variant (the rank inside a variant's list is a 22-bit field)", -6);
        c->cnv_expect = !c->h_cnv_start.empty();
        deliver_result(c, out);
        // timings
        lps_timings &t = c->tm; memset(&t, 0, sizeof t);
        t.n_stages = ST_COUNT;
        int prev = -1;
        if (c->timing_level == 1) { float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, c->ev[ST_EXTRACT], c->ev[ST_GROUPS])); t.ms_kernel[ST_EXTRACT] = ms; }
        for (int i = 0; i <= ST_COUNT && c->timing_level == 2; ++i) {
            const bool last = (i == ST_COUNT);
            if (!last && !c->ev_used[i]) continue;
            if (prev >= 0) { float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, c->ev[prev], last ? c->ev_end : c->ev[i])); t.ms_kernel[prev] = ms; }
            prev = i;
        }
        HIP_TRY(hipEventElapsedTime(&t.ms_total, c->ev_begin, c->ev_end));
        t.n_obs = (int64_t)c->h_cnt.n_obs_final; t.n_nodes = c->h_cnt.n_nodes; t.n_pairs = (int64_t)c->h_cnt.n_pairs; t.n_reads_used = c->h_cnt.n_kept;
        // algorithmic bytes (SURVEY.md §8d closed forms)
        t.algorithmic_bytes[ST_EXTRACT] = 36ll * c->nR + 4ll * (int64_t)c->n_cig + (int64_t)c->h_cnt.n_obs_final * (1 + 1 + 12 + 8);
        t.algorithmic_bytes[ST_EDGES] = 8ll * (int64_t)c->h_cnt.n_pairs + 8ll * (int64_t)c->h_cnt.n_obs_final + 16ll * c->P.connect_adjacent * (int64_t)c->h_cnt.n_nodes;
        t.algorithmic_bytes[ST_SCAN] = (16ll * c->P.connect_adjacent + 64) * (int64_t)c->h_cnt.n_nodes;   // SURVEY.md 8d closed form (the edge matrix); the kernels read 1 B per cell
        t.algorithmic_bytes[ST_CORR] = 16ll * (int64_t)c->h_cnt.n_obs_final + 32ll * (int64_t)c->h_cnt.n_nodes;
        t.n_scan_segments = c->h_cnt.n_nodes ? scan_segments((int)c->h_cnt.n_nodes) - 1 : 0; t.n_scan_replayed = c->h_stats[2];
        c->phase_valid = true;
    } catch (std::string &e) { return fail(c, e); }
    return 0;
}

// shared by haplotag and the somatic tagging pass: variant prep + scoring kernel + D2H of the integer counts
static int run_scorer(lps_ctx *c, bool somatic, uint8_t *status, int32_t *hp1, int32_t *hp2, uint8_t *n_ps, int32_t *ps_min,
                      int32_t *hp3, int32_t *d1, int32_t *d2) {
    const int nR = c->nR, nV = c->nV;
    hipStream_t s = c->stream;
    // per-read outputs side by side in one allocation: they leave in ONE copy into pinned memory (five to eight copies into the caller's pageable
    // arrays are staged by the runtime one after the other and block the host meanwhile)
    size_t hbytes = 0;
    auto hslot = [&](size_t bytes) { const size_t at = hbytes; hbytes += (bytes + 255) & ~(size_t)255; return at; };
    const size_t o_st = hslot(nR), o_h1 = hslot((size_t)nR * 4), o_h2 = hslot((size_t)nR * 4), o_np = hslot(nR), o_pm = hslot((size_t)nR * 4), o_h3 = hslot((size_t)nR * 4),
                 o_d1 = hslot((size_t)nR * 4), o_d2 = hslot((size_t)nR * 4);
    c->hap_pool.reserve(hbytes);
    c->hap_status.carve(c->hap_pool.p + o_st, nR); c->hap_h1.carve(c->hap_pool.p + o_h1, nR); c->hap_h2.carve(c->hap_pool.p + o_h2, nR); c->hap_nps.carve(c->hap_pool.p + o_np, nR);
    c->hap_psmin.carve(c->hap_pool.p + o_pm, nR); c->hap_h3.carve(c->hap_pool.p + o_h3, nR); c->hap_d1.carve(c->hap_pool.p + o_d1, nR); c->hap_d2.carve(c->hap_pool.p + o_d2, nR);
    const size_t span = somatic ? hbytes : o_h3;
    if (span + 1024 > c->h_res_bytes) { if (c->h_res) HIP_TRY(hipHostFree(c->h_res)); c->h_res = nullptr; c->h_res_bytes = span + span / 4 + 4096; HIP_TRY(hipHostMalloc((void **)&c->h_res, c->h_res_bytes)); }
    c->v_bucket.reserve((size_t)(((long long)c->last_pos + 1) >> LPS_BUCKET_SHIFT) + 8); c->v_rec.reserve((size_t)nV + 1); c->r_v0.reserve((size_t)nR + 1);
    HIP_TRY(hipEventRecord(c->ev_begin, s));
    HIP_TRY(hipMemsetAsync(c->d_cnt, 0, sizeof(LpsCounters), s));
    VarView V = var_view(c); ReadView R = read_view(c);
    for (auto &u : c->ev_used) u = false;
    mark(c, ST_PREP);
    launch_variant_prep(V, /*is_ont (filterSNP is a `phase` step)*/ 0, c->v_bucket.p, c->v_rec.p, s, R.ref_start, somatic ? R.n : 0, somatic ? c->r_v0.p : nullptr);   // (the stream walk starts every alignment at its first candidate row)
    mark(c, ST_EXTRACT);
    HapOut H{c->hap_status.p, c->hap_h1.p, c->hap_h2.p, c->hap_nps.p, c->hap_psmin.p, c->hap_h3.p, c->hap_d1.p, c->hap_d2.p, nullptr, nullptr, 0.0};
    for (int attempt = 0; attempt < 2; ++attempt) {
        launch_haplotag(V, R, H, c->P.mapping_quality, c->P.tag_supplementary, somatic ? 1 : 0, c->d_cnt, s, /*general=*/attempt == 1);
        if (attempt == 0) mark(c, ST_D2H);
        HIP_TRY(hipMemcpyAsync(c->h_res, c->hap_pool.p, span, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(c->h_cnt_pin, c->d_cnt, sizeof(LpsCounters), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipEventRecord(c->ev_end, s));
        HIP_TRY(hipStreamSynchronize(s));
        c->h_cnt = *c->h_cnt_pin;
        // the stream walk (somatic tagging pass) met a record outside its arithmetic - an op of 2^24 bases, a job spanning 2^30: the per-op-prefix walker takes any record
        if (!(c->h_cnt.err & LPS_ERR_KEY_RANGE) || attempt == 1) break;
        HIP_TRY(hipMemsetAsync(c->d_cnt, 0, sizeof(LpsCounters), s));
    }
    if (nR < 200000) {
        memcpy(status, c->h_res + o_st, (size_t)nR); memcpy(hp1, c->h_res + o_h1, (size_t)nR * 4); memcpy(hp2, c->h_res + o_h2, (size_t)nR * 4);
        memcpy(n_ps, c->h_res + o_np, (size_t)nR); memcpy(ps_min, c->h_res + o_pm, (size_t)nR * 4);
    } else {
        std::thread t1([&] { memcpy(hp1, c->h_res + o_h1, (size_t)nR * 4); }), t2([&] { memcpy(hp2, c->h_res + o_h2, (size_t)nR * 4); }), t3([&] { memcpy(ps_min, c->h_res + o_pm, (size_t)nR * 4); });
        memcpy(status, c->h_res + o_st, (size_t)nR); memcpy(n_ps, c->h_res + o_np, (size_t)nR);
        t1.join(); t2.join(); t3.join();
    }
    if (somatic) { memcpy(hp3, c->h_res + o_h3, (size_t)nR * 4); memcpy(d1, c->h_res + o_d1, (size_t)nR * 4); memcpy(d2, c->h_res + o_d2, (size_t)nR * 4); }
    if (c->h_cnt.err & LPS_ERR_BAD_CIGAR) return fail(c, "Alignment find unsupported CIGAR operation", -2);
    lps_timings &t = c->tm; memset(&t, 0, sizeof t);
    t.n_stages = ST_COUNT;
    HIP_TRY(hipEventElapsedTime(&t.ms_kernel[ST_PREP], c->ev[ST_PREP], c->ev[ST_EXTRACT]));
    HIP_TRY(hipEventElapsedTime(&t.ms_kernel[ST_EXTRACT], c->ev[ST_EXTRACT], c->ev[ST_D2H]));
    HIP_TRY(hipEventElapsedTime(&t.ms_kernel[ST_D2H], c->ev[ST_D2H], c->ev_end));
    HIP_TRY(hipEventElapsedTime(&t.ms_total, c->ev_begin, c->ev_end));
    t.algorithmic_bytes[ST_EXTRACT] = 36ll * nR + 4ll * (int64_t)c->n_cig + 10ll * nR;   // + observations (unknown here)
    return 0;
}

int lps_set_read_votes(lps_ctx *c, const int32_t *h1, const int32_t *h2, int64_t n_reads) {
    if (!c) return -1;
    c->votes_h1.clear(); c->votes_h2.clear();
    if (!h1 && !h2) return 0;
    if (!h1 || !h2) return fail(c, "lps_set_read_votes: both arrays or none");
    if (n_reads != c->nR) return fail(c, "lps_set_read_votes: n_reads must equal the number of pushed alignments");
    c->votes_h1.assign(h1, h1 + n_reads); c->votes_h2.assign(h2, h2 + n_reads);
    return 0;
}

// PQ = int(-10 log10(min / (max + min))) of (min, max) votes below 64, with the HOST's libm (HaplotagStrategy.cpp:287): what the
// GPU looks up and the host would compute are the same numbers
struct PqSmall { int v[64][64]; };
static const PqSmall &pq_small_table() {
    static const PqSmall tab = [] { PqSmall t{}; for (int mn = 1; mn < 64; ++mn) for (int mx = mn; mx < 64; ++mx) t.v[mn][mx] = -10 * (std::log10((double)mn / double((double)mx + (double)mn))); return t; }();
    return tab;
}

// K consecutive calls behind ONE entry (bench.py's timed region: a caller written in Python re-enters the interpreter between calls, and with several
// contexts driven from several threads a step then waits for the interpreter lock, not for the GPU).  ms_each (may be NULL): wall time of every call.
int lps_phase_chromosome_steps(lps_ctx *c, lps_phase_result *out, int k, double *ms_each) {
    for (int i = 0; i < k; ++i) {
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = lps_phase_chromosome(c, out);
        if (ms_each) ms_each[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (rc) return rc;
    }
    return 0;
}
int lps_haplotag_chromosome(lps_ctx *c, lps_haplotag_result *out);
int lps_haplotag_chromosome_steps(lps_ctx *c, lps_haplotag_result *out, int k, double *ms_each) {
    for (int i = 0; i < k; ++i) {
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = lps_haplotag_chromosome(c, out);
        if (ms_each) ms_each[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (rc) return rc;
    }
    return 0;
}
int lps_haplotag_chromosome(lps_ctx *c, lps_haplotag_result *out) {
    if (!c || !out) return -1;
    AllocScope alloc_scope(c);
    try {
        HIP_TRY(hipSetDevice(c->device));
        const int nR = c->nR, nV = c->nV;
        if (out->n_reads != nR) return fail(c, "lps_haplotag_result.n_reads must equal the number of pushed alignments");
        if (nR == 0) return 0;
        if (nV > 0 && !c->has_hap) return fail(c, "haplotag needs hp1_is_alt and phase_set in the variant table");
        if (nV > 0 && c->ref_len_eff == 0) return fail(c, "lps_set_reference has not been called");
        const bool votes = !c->votes_h1.empty();
        if (votes && (int)c->votes_h1.size() != nR) return fail(c, "lps_set_read_votes was called for another set of alignments");
        hipStream_t s = c->stream;
        // ---- one 16-byte record per read: votes, PS and the read-level decision (judgeReadHap, HaplotagStrategy.cpp:243-300) taken on the GPU
        c->hap_rec.reserve((size_t)nR + 1);
        const size_t span = (size_t)nR * sizeof(uint4);
        if (span + 1024 > c->h_res_bytes) { if (c->h_res) HIP_TRY(hipHostFree(c->h_res)); c->h_res = nullptr; c->h_res_bytes = span + span / 4 + 4096; HIP_TRY(hipHostMalloc((void **)&c->h_res, c->h_res_bytes)); }
        if (!c->pq_ready) { c->pq_tab.reserve(64 * 64); HIP_TRY(hipMemcpyAsync(c->pq_tab.p, &pq_small_table().v[0][0], 64 * 64 * sizeof(int), hipMemcpyHostToDevice, s)); c->pq_ready = true; }
        if (votes) { upload(c, c->d_votes1, c->votes_h1.data(), (size_t)nR); upload(c, c->d_votes2, c->votes_h2.data(), (size_t)nR); }
        c->v_bucket.reserve((size_t)(((long long)c->last_pos + 1) >> LPS_BUCKET_SHIFT) + 8); c->v_rec.reserve((size_t)nV + 1); c->r_v0.reserve((size_t)nR + 1);
        HIP_TRY(hipEventRecord(c->ev_begin, s));
        HIP_TRY(hipMemsetAsync(c->d_cnt, 0, sizeof(LpsCounters), s));
        VarView V = var_view(c); ReadView R = read_view(c);
        for (auto &u : c->ev_used) u = false;
        mark(c, ST_PREP);
        launch_variant_prep(V, /*is_ont (filterSNP is a `phase` step)*/ 0, c->v_bucket.p, c->v_rec.p, s, R.ref_start, R.n, c->r_v0.p);
        mark(c, ST_EXTRACT);
        HapOut H{};
        H.pct_thr = c->P.percentage_threshold; H.rec = c->hap_rec.p; H.pq_tab = c->pq_tab.p; H.votes1 = votes ? c->d_votes1.p : nullptr; H.votes2 = votes ? c->d_votes2.p : nullptr;
        for (int attempt = 0; attempt < 2; ++attempt) {
            launch_haplotag(V, R, H, c->P.mapping_quality, c->P.tag_supplementary, 0, c->d_cnt, s, /*general=*/attempt == 1);
            if (attempt == 0) mark(c, ST_D2H);
            HIP_TRY(hipMemcpyAsync(c->h_res, c->hap_rec.p, span, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(c->h_cnt_pin, c->d_cnt, sizeof(LpsCounters), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipEventRecord(c->ev_end, s));
            HIP_TRY(hipStreamSynchronize(s));
            c->h_cnt = *c->h_cnt_pin;
            // one CIGAR operation of 2^24 bases and more, or the alignments of a job spanning more than 2^30 bases: outside the stream walk's arithmetic
            // (24-bit multiplies, 32-bit stream coordinates).  The chromosome is scored again by the per-op-prefix walker, which takes any BAM record
            if (!(c->h_cnt.err & LPS_ERR_KEY_RANGE) || attempt == 1) break;
            HIP_TRY(hipMemsetAsync(c->d_cnt, 0, sizeof(LpsCounters), s));
        }
        if (c->h_cnt.err & LPS_ERR_BAD_CIGAR) return fail(c, "Alignment find unsupported CIGAR operation", -2);
        // ---- the records into the caller's arrays (PQ of reads with 64 votes or more: libm here)
        const uint4 *rec = (const uint4 *)c->h_res;
        auto unpack = [&](int r0, int r1) -> int64_t {
            int64_t n_tagged = 0;
            for (int r = r0; r < r1; ++r) {
                const uint4 w = rec[r];
                const int a = (int)w.y, b = (int)w.z; const unsigned hp = (w.x >> 16) & 0xffu, nps = (w.x >> 8) & 0xffu; int pq = (int)(w.x >> 24);
                if (pq == 255) { const double mn = a < b ? a : b, mx = a < b ? b : a; pq = -10 * (std::log10(mn / double(mx + mn))); }
                out->status[r] = (uint8_t)(w.x & 0xffu); out->hp1[r] = a; out->hp2[r] = b; out->n_ps[r] = (uint8_t)nps; out->ps_min[r] = (int32_t)w.w;
                out->hp[r] = (uint8_t)hp; out->pq[r] = pq; out->ps[r] = (hp && nps) ? (int32_t)w.w : 0;   // no PS seen (a read tagged by SV / MOD votes alone): the reference reads begin() of an empty map, 0 with libstdc++
                n_tagged += hp != 0;
            }
            return n_tagged;
        };
        int64_t tagged = 0;
        if (nR < 200000) tagged = unpack(0, nR);
        else {                                                            // a whole 50x chromosome: a few host threads share the copy-out
            const int NT = 8, nt = nR >= 400000 ? NT : 4; std::thread th[NT]; int64_t part[NT] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int t = 1; t < nt; ++t) th[t] = std::thread([&, t] { part[t] = unpack((int)((int64_t)nR * t / nt), (int)((int64_t)nR * (t + 1) / nt)); });
            part[0] = unpack(0, (int)((int64_t)nR / nt));                 // (the calling thread takes a share instead of waiting)
            tagged += part[0];
            for (int t = 1; t < nt; ++t) { th[t].join(); tagged += part[t]; }
        }
        lps_timings &t = c->tm; memset(&t, 0, sizeof t);
        t.n_stages = ST_COUNT;
        HIP_TRY(hipEventElapsedTime(&t.ms_kernel[ST_PREP], c->ev[ST_PREP], c->ev[ST_EXTRACT]));
        HIP_TRY(hipEventElapsedTime(&t.ms_kernel[ST_EXTRACT], c->ev[ST_EXTRACT], c->ev[ST_D2H]));
        HIP_TRY(hipEventElapsedTime(&t.ms_kernel[ST_D2H], c->ev[ST_D2H], c->ev_end));
        HIP_TRY(hipEventElapsedTime(&t.ms_total, c->ev_begin, c->ev_end));
        t.algorithmic_bytes[ST_EXTRACT] = 36ll * nR + 4ll * (int64_t)c->n_cig + 16ll * nR;   // + observations (unknown here)
        t.n_reads_used = tagged;
    } catch (std::string &e) { return fail(c, e); }
    return 0;
}

int lps_somatic_tag_chromosome(lps_ctx *c, lps_somatic_tag_result *out) {
    if (!c || !out) return -1;
    try {
        HIP_TRY(hipSetDevice(c->device));
        const int nR = c->nR, nV = c->nV;
        if (out->n_reads != nR) return fail(c, "lps_somatic_tag_result.n_reads must equal the number of pushed alignments");
        if (nR == 0) return 0;
        if (nV > 0 && !c->has_somatic) return fail(c, "somatic tagging needs hp1_is_alt, phase_set, somatic_role and derive_hp in the variant table");
        if (nV > 0 && c->ref_len_eff == 0) return fail(c, "lps_set_reference has not been called");
        int rc = run_scorer(c, true, out->status, out->hp1, out->hp2, out->n_ps, out->ps_min, out->hp3, out->derive_h1, out->derive_h2);
        if (rc) return rc;
        // judgeSomaticReadHap (HaplotagStrategy.cpp:452-602) + inheritHaplotype (SomaticHaplotagProcess.cpp:461-527) + PS rule (:416-434)
        const double thr = c->P.percentage_threshold;
        auto judge = [&](int r_lo, int r_hi) -> int64_t {
        int64_t tagged = 0;
        for (int r = r_lo; r < r_hi; ++r) {
            int hp = 0, pq = 0, ps = -1;
            if (out->status[r] == 0) {
                const int h1 = out->hp1[r], h2 = out->hp2[r], h3 = out->hp3[r], h4 = 0;
                double tMin, tMax, nMin, nMax; int maxT, maxN;
                if (h3 > h4) { tMin = h4; tMax = h3; maxT = 3; } else { tMin = h3; tMax = h4; maxT = 4; }
                if (h1 > h2) { nMin = h2; nMax = h1; maxN = 1; } else { nMin = h1; nMax = h2; maxN = 2; }
                const double tumSim = (tMax == 0) ? 0.0 : tMax / (tMax + tMin);
                const double norSim = (nMax == 0) ? 0.0 : nMax / (nMax + nMin);
                if (tMax != 0) {
                    if (tumSim >= thr) {
                        if (norSim >= thr) hp = (maxT == 3) ? (maxN == 1 ? 5 : 7) : (maxN == 1 ? 6 : 8);
                        else hp = (maxT == 3) ? 3 : 4;
                    }
                } else if (nMax != 0) { if (norSim >= thr) hp = maxN; }
                if (out->n_ps[r] > 1) hp = 0;
                if (nMax == 0 && tMax == 0) pq = 0;
                else if (tMax != 0) { if (tMax == tMax + tMin) pq = 40; else pq = -10 * (std::log10((double)tMin / double(tMax + tMin))); }
                else if (nMax != 0) { if (nMax == nMax + nMin) pq = 40; else pq = -10 * (std::log10((double)nMin / double(nMax + nMin))); }
                if (hp == 3) {
                    const int d1 = out->derive_h1[r], d2 = out->derive_h2[r];
                    int mx, mn, mh;
                    if (d1 > d2) { mx = d1; mn = d2; mh = 1; } else { mx = d2; mn = d1; mh = 2; }
                    const float sim = (mx == 0) ? 0.0f : ((float)mx / ((float)mx + (float)mn));
                    if (sim >= thr) hp = (mh == 1) ? 5 : 7;
                }
                if (hp != 0 && out->n_ps[r] > 0) ps = out->ps_min[r];
            }
            out->hp[r] = (uint8_t)hp; out->pq[r] = pq; out->ps[r] = ps;
            tagged += hp != 0;
        }
        return tagged; };
        int64_t tagged = 0;
        if (nR < 200000) tagged = judge(0, nR);
        else {                                                            // a whole 50x chromosome: the reads are independent, a few host threads share them
            const int nt = 8; std::thread th[nt]; int64_t part[nt] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int t = 1; t < nt; ++t) th[t] = std::thread([&, t] { part[t] = judge((int)((int64_t)nR * t / nt), (int)((int64_t)nR * (t + 1) / nt)); });
            tagged = judge(0, (int)((int64_t)nR / nt));
            for (int t = 1; t < nt; ++t) { th[t].join(); tagged += part[t]; }
        }
        c->tm.n_reads_used = tagged;
    } catch (std::string &e) { return fail(c, e); }
    return 0;
}

int lps_somatic_extract_normal(lps_ctx *c, lps_site_counters *out) {
    if (!c || !out) return -1;
    try {
        HIP_TRY(hipSetDevice(c->device));
        const int nR = c->nR, nV = c->nV;
        if (out->n != nV) return fail(c, "lps_site_counters.n must equal the variant table size");
        if (out->read_hp && out->n_reads != nR) return fail(c, "lps_site_counters.n_reads must equal the number of pushed alignments");
        if (nR == 0 || nV == 0) { memset(out->counters, 0, (size_t)nV * LPS_SITE_COUNTERS * sizeof(int32_t)); return 0; }   // (otherwise every counter is overwritten by the copy below)
        if (!c->has_tkind) return fail(c, "somatic extraction needs hp1_is_alt, phase_set, somatic_role and tumor_kind in the variant table");
        if (c->ref_len_eff == 0) return fail(c, "lps_set_reference has not been called");
        hipStream_t s = c->stream;
        c->site.reserve((size_t)nV * LPS_SITE_COUNTERS); c->read_hp.reserve(nR);
        c->v_bucket.reserve((size_t)(((long long)c->last_pos + 1) >> LPS_BUCKET_SHIFT) + 8); c->v_rec.reserve((size_t)nV + 1); c->r_v0.reserve((size_t)nR + 1);
        HIP_TRY(hipEventRecord(c->ev_begin, s));
        VarView V = var_view(c); ReadView R = read_view(c);
        for (auto &u : c->ev_used) u = false;
        // The stream walk (votes, base counters, the read's haplotype) lists the (tumor row, alignment) pairs it touches in LPS_TARENAS arenas; the rows'
        // ReadHpCount comes from that list (no second walk).  An arena that turns out too short: again with room for the fullest one.  A record outside
        // the stream walk's arithmetic (LPS_ERR_KEY_RANGE): both passes on the per-op-prefix walker, which takes any BAM record.
        { const size_t want = (2 * (size_t)nR + 4096 + LPS_TARENAS - 1) / LPS_TARENAS; if (c->n_pair_arena < want) c->n_pair_arena = want; }
        bool general = false;
        for (int attempt = 0; attempt < 4; ++attempt) {
            const size_t pa = c->n_pair_arena, ipc = pa * LPS_TARENAS;
            c->t_apair_site.reserve(ipc + 1); c->t_apair_read.reserve(ipc + 1); c->t_ctr.reserve(2 * LPS_TARENAS * 16 + 8);
            HIP_TRY(hipMemsetAsync(c->d_cnt, 0, sizeof(LpsCounters), s));
            HIP_TRY(hipMemsetAsync(c->site.p, 0, (size_t)nV * LPS_SITE_COUNTERS * sizeof(int32_t), s));
            HIP_TRY(hipMemsetAsync(c->t_ctr.p, 0, (size_t)LPS_TARENAS * 16 * sizeof(unsigned long long), s));
            if (attempt == 0) { mark(c, ST_PREP); launch_variant_prep(V, 0, c->v_bucket.p, c->v_rec.p, s, R.ref_start, R.n, c->r_v0.p); mark(c, ST_EXTRACT); }
            HapOut H{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, c->site.p, c->read_hp.p, c->P.percentage_threshold};
            if (!general) { H.pair_ctr = c->t_ctr.p; H.pair_arena = (long long)pa; H.apair_site = c->t_apair_site.p; H.apair_read = c->t_apair_read.p; }
            launch_haplotag(V, R, H, c->P.mapping_quality, c->P.tag_supplementary, 2, c->d_cnt, s, general);   // votes + base counters + read haplotype (+ the pair list)
            if (!general) launch_normal_pair_sites(H, s);                                                      // ReadHpCount of the touched sites
            else launch_haplotag(V, R, H, c->P.mapping_quality, c->P.tag_supplementary, 3, c->d_cnt, s, true);
            if (attempt == 0) mark(c, ST_D2H);
            std::vector<unsigned long long> ctr((size_t)LPS_TARENAS * 16, 0ull);
            HIP_TRY(hipMemcpyAsync(&c->h_cnt, c->d_cnt, sizeof(LpsCounters), hipMemcpyDeviceToHost, s));
            if (!general) HIP_TRY(hipMemcpyAsync(ctr.data(), c->t_ctr.p, ctr.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            if (!general && (c->h_cnt.err & LPS_ERR_KEY_RANGE)) { general = true; continue; }
            unsigned long long fullest = 0; for (int a = 0; a < LPS_TARENAS; ++a) fullest = std::max(fullest, ctr[(size_t)a * 16]);
            if (!general && fullest > pa) { if (attempt == 3) return fail(c, "somatic extraction: a list arena kept overflowing"); c->n_pair_arena = (size_t)fullest + (size_t)fullest / 8 + 1024; continue; }
            break;
        }
        HIP_TRY(hipMemcpyAsync(out->counters, c->site.p, (size_t)nV * LPS_SITE_COUNTERS * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        if (out->read_hp) HIP_TRY(hipMemcpyAsync(out->read_hp, c->read_hp.p, (size_t)nR, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipEventRecord(c->ev_end, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (c->h_cnt.err & LPS_ERR_BAD_CIGAR) return fail(c, "Alignment find unsupported CIGAR operation", -2);
        if (out->read_hp) for (int r = 0; r < nR; ++r) if (out->read_hp[r] == 255) out->read_hp[r] = 0;
        lps_timings &t = c->tm; memset(&t, 0, sizeof t);
        t.n_stages = ST_COUNT;
        HIP_TRY(hipEventElapsedTime(&t.ms_kernel[ST_EXTRACT], c->ev[ST_EXTRACT], c->ev[ST_D2H]));
        HIP_TRY(hipEventElapsedTime(&t.ms_total, c->ev_begin, c->ev_end));
    } catch (std::string &e) { return fail(c, e); }
    return 0;
}

int lps_somatic_extract_tumor(lps_ctx *c, lps_tumor_extract_result *out) {
    if (!c || !out) return -1;
    try {
        HIP_TRY(hipSetDevice(c->device));
        const int nR = c->nR, nV = c->nV;
        if (out->n != nV) return fail(c, "lps_tumor_extract_result.n must equal the variant table size");
        if (out->n_reads != nR) return fail(c, "lps_tumor_extract_result.n_reads must equal the number of pushed alignments");
        out->n_pairs = 0; out->n_windows = 0;
        if (nR == 0) { memset(out->site, 0, (size_t)nV * LPS_TSITE_COUNTERS * sizeof(int32_t)); return 0; }   // (otherwise every counter is overwritten by the copy below)
        if (nV > 0 && !c->has_tkind) return fail(c, "somatic extraction needs hp1_is_alt, phase_set, somatic_role and tumor_kind in the variant table");
        if (nV > 0 && c->ref_len_eff == 0) return fail(c, "lps_set_reference has not been called");
        hipStream_t s = c->stream;
        const size_t pc = (size_t)std::max<int64_t>(out->pair_capacity, 1), wc = (size_t)std::max<int64_t>(out->win_capacity, 1);
        c->site.reserve((size_t)nV * LPS_TSITE_COUNTERS + 1);
        c->hap_status.reserve(nR); c->hap_h1.reserve(nR); c->hap_h2.reserve(nR); c->hap_h3.reserve(nR); c->hap_nps.reserve(nR); c->hap_psmin.reserve(nR);
        c->t_hp.reserve(nR); c->t_has.reserve(nR); c->t_end.reserve(nR); c->t_len.reserve(nR); c->t_ctr.reserve(4);
        c->t_pair_site.reserve(pc); c->t_pair_read.reserve(pc); c->t_pair_hp.reserve(pc);
        c->t_win_site.reserve(wc); c->t_win_allele.reserve(wc); c->t_win_off.reserve(wc); c->t_win_base.reserve(wc);
        c->v_bucket.reserve((size_t)(((long long)c->last_pos + 1) >> LPS_BUCKET_SHIFT) + 8); c->v_rec.reserve((size_t)nV + 1); c->r_v0.reserve((size_t)nR + 1);
        // The two lists the passes append to - (site, read, base HP) pairs and window hits (alignment x TUMOR row) - live in LPS_TARENAS arenas each
        // (lps_kernels.h).  They start at two entries per alignment - a tumor VCF holds the somatic sites, an alignment meets one or none - and when
        // an arena turns out too short the passes run again with what the fullest one needs; the size sticks to the context (the per-slot kernels
        // behind the lists cost what the arenas hold, not what they could)
        const size_t want = (2 * (size_t)nR + 4096 + LPS_TARENAS - 1) / LPS_TARENAS;
        if (c->t_pair_arena < want) c->t_pair_arena = want;
        if (c->t_hit_arena < want) c->t_hit_arena = want;
        unsigned long long tot[4] = {0, 0, 0, 0}, n_win = 0;
        bool general = false;                                             // the passes on the per-op-prefix walker: after the stream walk met a record outside its arithmetic
        for (int attempt = 0; attempt < 3; ++attempt) {
            const size_t pa = c->t_pair_arena, ha = c->t_hit_arena, hc = ha * LPS_TARENAS, ipc = pa * LPS_TARENAS;
            if (2 * hc + 1 > 0xfffffff0ull) return fail(c, "somatic extraction: more than 2^31 window hits");
            c->t_hits.reserve(hc + 1); c->t_hit_rp.reserve(hc + 1);
            c->t_apair_site.reserve(ipc + 1); c->t_apair_read.reserve(ipc + 1); c->t_apair_hp.reserve(ipc + 1);
            c->t_ctr.reserve(2 * LPS_TARENAS * 16 + 8);
            HIP_TRY(hipEventRecord(c->ev_begin, s));
            HIP_TRY(hipMemsetAsync(c->d_cnt, 0, sizeof(LpsCounters), s));
            HIP_TRY(hipMemsetAsync(c->site.p, 0, ((size_t)nV * LPS_TSITE_COUNTERS + 1) * sizeof(int32_t), s));
            HIP_TRY(hipMemsetAsync(c->t_ctr.p, 0, (2 * LPS_TARENAS * 16 + 8) * sizeof(unsigned long long), s));
            VarView V = var_view(c); ReadView R = read_view(c);
            for (auto &u : c->ev_used) u = false;
            mark(c, ST_PREP);
            if (nV) launch_variant_prep(V, 0, c->v_bucket.p, c->v_rec.p, s, R.ref_start, R.n, c->r_v0.p);
            mark(c, ST_EXTRACT);
            TumOut T{};
            T.site = c->site.p; T.status = c->hap_status.p; T.hp1 = c->hap_h1.p; T.hp2 = c->hap_h2.p; T.hp3 = c->hap_h3.p; T.hp = c->t_hp.p; T.n_ps = c->hap_nps.p; T.ps_min = c->hap_psmin.p;
            T.end_pos = c->t_end.p; T.read_len = c->t_len.p; T.has_site = c->t_has.p;
            T.pair_ctr = c->t_ctr.p; T.hit_ctr = c->t_ctr.p + LPS_TARENAS * 16; T.tot = c->t_ctr.p + 2 * LPS_TARENAS * 16;
            T.pair_arena = (long long)pa; T.hit_arena = (long long)ha; T.apair_site = c->t_apair_site.p; T.apair_read = c->t_apair_read.p; T.apair_hp = c->t_apair_hp.p;
            T.pair_cap = (long long)out->pair_capacity; T.win_cap = (long long)out->win_capacity;
            T.pair_site = c->t_pair_site.p; T.pair_read = c->t_pair_read.p; T.pair_hp = c->t_pair_hp.p;
            T.win_site = c->t_win_site.p; T.win_allele = c->t_win_allele.p; T.win_offset = c->t_win_off.p; T.win_base = c->t_win_base.p; T.pct_thr = c->P.percentage_threshold;
            T.hits = c->t_hits.p; T.hit_rp = c->t_hit_rp.p; T.win_total = T.tot + 4;
            launch_tumor_extract(V, R, T, c->P.mapping_quality, c->P.tag_supplementary, 0, c->d_cnt, s, general);
            launch_tumor_windows(V, R, T, s);
            launch_tumor_extract(V, R, T, c->P.mapping_quality, c->P.tag_supplementary, 1, c->d_cnt, s, general);
            launch_tumor_pairs_out(T, s);
            mark(c, ST_D2H);
            unsigned walk_err = 0;
            HIP_TRY(hipMemcpyAsync(tot, T.tot, sizeof tot, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(&n_win, T.win_total, sizeof n_win, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(&walk_err, &c->d_cnt->err, sizeof walk_err, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            // one CIGAR operation of 2^24 bases and more, or the alignments of a job spanning more than 2^30 bases: the passes again on the general walker
            if (!general && (walk_err & LPS_ERR_KEY_RANGE)) { general = true; --attempt; continue; }
            if (tot[1] <= pa && tot[3] <= ha) break;
            if (tot[1] > pa) c->t_pair_arena = (size_t)tot[1] + (size_t)tot[1] / 8 + 1024;   // an arena was too short: again with room for the fullest one
            if (tot[3] > ha) c->t_hit_arena = (size_t)tot[3] + (size_t)tot[3] / 8 + 1024;
            if (attempt == 2) return fail(c, "somatic extraction: a list arena kept overflowing");
        }
        const unsigned long long ctr[2] = {tot[0], n_win};
        TumOut T{}; T.status = c->hap_status.p; T.hp1 = c->hap_h1.p; T.hp2 = c->hap_h2.p; T.hp3 = c->hap_h3.p; T.hp = c->t_hp.p; T.n_ps = c->hap_nps.p; T.ps_min = c->hap_psmin.p;
        T.end_pos = c->t_end.p; T.read_len = c->t_len.p; T.has_site = c->t_has.p; T.pair_site = c->t_pair_site.p; T.pair_read = c->t_pair_read.p; T.pair_hp = c->t_pair_hp.p;
        T.win_site = c->t_win_site.p; T.win_allele = c->t_win_allele.p; T.win_offset = c->t_win_off.p; T.win_base = c->t_win_base.p;
        HIP_TRY(hipMemcpyAsync(out->site, c->site.p, (size_t)nV * LPS_TSITE_COUNTERS * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(out->status, T.status, (size_t)nR, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(out->hp1, T.hp1, (size_t)nR * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(out->hp2, T.hp2, (size_t)nR * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(out->hp3, T.hp3, (size_t)nR * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(out->hp, T.hp, (size_t)nR, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(out->n_ps, T.n_ps, (size_t)nR, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(out->ps_min, T.ps_min, (size_t)nR * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(out->end_pos, T.end_pos, (size_t)nR * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(out->read_len, T.read_len, (size_t)nR * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(out->has_site, T.has_site, (size_t)nR, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(&c->h_cnt, c->d_cnt, sizeof(LpsCounters), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (c->h_cnt.err & LPS_ERR_BAD_CIGAR) return fail(c, "Alignment find unsupported CIGAR operation", -2);
        out->n_pairs = (int64_t)ctr[0]; out->n_windows = (int64_t)ctr[1];
        const size_t np_ = (size_t)std::min<int64_t>(out->n_pairs, out->pair_capacity), nw_ = (size_t)std::min<int64_t>(out->n_windows, out->win_capacity);
        if (np_) { HIP_TRY(hipMemcpyAsync(out->pair_site, T.pair_site, np_ * 4, hipMemcpyDeviceToHost, s)); HIP_TRY(hipMemcpyAsync(out->pair_read, T.pair_read, np_ * 4,
                hipMemcpyDeviceToHost, s)); HIP_TRY(hipMemcpyAsync(out->pair_base_hp, T.pair_hp, np_, hipMemcpyDeviceToHost, s)); }
        if (nw_) { HIP_TRY(hipMemcpyAsync(out->win_site, T.win_site, nw_ * 4, hipMemcpyDeviceToHost, s)); HIP_TRY(hipMemcpyAsync(out->win_allele, T.win_allele, nw_, hipMemcpyDeviceToHost, s));
                   HIP_TRY(hipMemcpyAsync(out->win_offset, T.win_offset, nw_ * 2, hipMemcpyDeviceToHost, s)); HIP_TRY(hipMemcpyAsync(out->win_base, T.win_base, nw_, hipMemcpyDeviceToHost, s)); }
        HIP_TRY(hipEventRecord(c->ev_end, s));
        HIP_TRY(hipStreamSynchronize(s));
        lps_timings &t = c->tm; memset(&t, 0, sizeof t);
        t.n_stages = ST_COUNT;
        HIP_TRY(hipEventElapsedTime(&t.ms_kernel[ST_EXTRACT], c->ev[ST_EXTRACT], c->ev[ST_D2H]));
        HIP_TRY(hipEventElapsedTime(&t.ms_total, c->ev_begin, c->ev_end));
        if (out->n_pairs > out->pair_capacity || out->n_windows > out->win_capacity) return fail(c, "pair/window list capacity too small (needed sizes are in n_pairs / n_windows)", -9);
    } catch (std::string &e) { return fail(c, e); }
    return 0;
}

int lps_get_timings(lps_ctx *c, lps_timings *t) { if (!c || !t) return -1; *t = c->tm; return 0; }

// ------------------------------------------------------------------------------------------------ dumps
int64_t lps_dump_observations(lps_ctx *c, int32_t *obs_count, int32_t *var_index, int8_t *allele, int16_t *quality, int64_t capacity) {
    if (!c || !c->phase_valid) return -1;
    if (c->nR == 0) return 0;                                            // nothing ran
    try {
        HIP_TRY(hipSetDevice(c->device));
        auto rows = download(c, c->rows.p, c->nR);
        const size_t tot = (size_t)(c->obs_capacity + c->obs_capacity / 4 + 4096);   // rows are scattered over the arenas
        auto obs = download(c, c->obs.p, tot);
        int64_t n = 0;
        for (int r = 0; r < c->nR; ++r) {
            if (obs_count) obs_count[r] = rows[r].cnt;
            for (int k = 0; k < rows[r].cnt; ++k, ++n) {
                if (var_index && n < capacity) {
                    const ObsRec &o = obs[rows[r].off + k];
                    int v = o.var; if (v < 0) v = -1 - v;       // erased later by the CNV filter
                    var_index[n] = v; allele[n] = (int8_t)aq_allele((uint16_t)o.aq); quality[n] = (int16_t)aq_quality((uint16_t)o.aq);
                }
            }
        }
        return n;
    } catch (std::string &e) { fail(c, e); return -1; }
}

int64_t lps_dump_graph(lps_ctx *c, int32_t *node_var_index, float *edge, int64_t node_capacity) {
    if (!c || !c->phase_valid) return -1;
    try {
        HIP_TRY(hipSetDevice(c->device));
        const int64_t N = c->h_cnt.n_nodes;
        if (node_var_index && N <= node_capacity) {
            auto nd = download(c, c->nodes.p, (size_t)N); memcpy(node_var_index, nd.data(), (size_t)N * 4);
            if (edge) { auto e = download(c, c->edge.p, (size_t)N * c->P.connect_adjacent * 4); memcpy(edge, e.data(), e.size() * 4); }
        }
        return N;
    } catch (std::string &e) { fail(c, e); return -1; }
}

int64_t lps_dump_votes(lps_ctx *c, int8_t *hp, int32_t *block_node, int64_t node_capacity) {
    if (!c || !c->phase_valid) return -1;
    try {
        HIP_TRY(hipSetDevice(c->device));
        const int64_t N = c->h_cnt.n_nodes;
        if (hp && N <= node_capacity) {
            auto h = download(c, c->hp.p, (size_t)N); memcpy(hp, h.data(), (size_t)N);
            auto b = download(c, c->block.p, (size_t)N); memcpy(block_node, b.data(), (size_t)N * 4);
        }
        return N;
    } catch (std::string &e) { fail(c, e); return -1; }
}

int64_t lps_dump_clips(lps_ctx *c, int32_t *pos, uint8_t *front_back, int64_t capacity) {
    if (!c || !c->phase_valid) return -1;
    try {
        HIP_TRY(hipSetDevice(c->device));
        const size_t n = c->h_cnt.n_clips;
        if (!c->clips_sorted && n) { const size_t need_k = GraphTemp::need(n + 1); if (need_k > c->temp_bytes) { c->temp.reserve(need_k, c->stream); c->temp_bytes = need_k; }
            launch_clip_sort((unsigned)n, c->clip_keys.p, c->clip_keys_s.p, c->temp.p, c->temp_bytes, c->stream); c->clips_sorted = true; }   // (the run itself had no use for the order)
        auto k = download(c, c->clip_keys_s.p, n);
        int64_t m = 0;
        for (size_t i = 0; i < n; ++i) { if (k[i] == ~0ull) break; if (pos && m < capacity) { pos[m] = (int32_t)(k[i] >> 1); front_back[m] = (uint8_t)(k[i] & 1); } ++m; }
        return m;
    } catch (std::string &e) { fail(c, e); return -1; }
}

int64_t lps_dump_cnv(lps_ctx *c, int32_t *start, int32_t *end, int64_t capacity, uint8_t *aln_deleted) {
    if (!c || !c->phase_valid) return -1;
    try {
        HIP_TRY(hipSetDevice(c->device));
        const int64_t K = (int64_t)c->h_cnv_start.size();
        for (int64_t i = 0; i < 2 * K && i < capacity && start && end; ++i) { start[i] = c->h_cnv_start[(size_t)(i % K)]; end[i] = c->h_cnv_end[(size_t)(i % K)]; }
        if (aln_deleted) { auto d = download(c, c->deleted.p, (size_t)c->nR); memcpy(aln_deleted, d.data(), (size_t)c->nR); }
        return 2 * K;
    } catch (std::string &e) { fail(c, e); return -1; }
}

}  // extern "C"
