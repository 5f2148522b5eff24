// longphase_amd — host CLI over liblps_hip.so: `longphase_amd phase ...` with the reference's flags and output format.
//
// SURVEY.md §8(f) widening (rows f-1 and f-4): BGZF/BAM decoding and the phased-VCF rewriter, written from scratch on zlib
// (htslib is not available on the GPU box).  It restates, citing the reference (relative to /root/reference/):
//   option surface of `phase`                 src/phase/Phasing.cpp:9-116
//   SnpParser row selection                   src/phase/ParsingBam.cpp:222-359
//   per-chromosome driver                     src/phase/PhasingProcess.cpp:113-173   (the hot path is one lps_phase_chromosome call)
//   SnpParser::writeLine (VCF rewrite rules)  src/phase/ParsingBam.cpp:460-635
//   SVParser / METHParser (--sv-file, --mod-file) src/phase/ParsingBam.cpp:915-1206, 1647-1952   (cli_extra.h)
//   SnpParser::preprocessDeepsomaticVCF (--deepsomatic_output) src/phase/ParsingBam.cpp:651-835, PhasingProcess.cpp:47-61
// --dot: <chr>.dot from the GPU's graph (SNP / indel graphs; with --sv-file / --mod-file the reference path must be used).  Not supported: CRAM.   (--indelQuality: cli_vcf.h IndelQual)
// Split in round 2: cli_common.h (loader), cli_bam.h (BGZF/BAM in, BGZF out), cli_vcf.h (VCF/FASTA in, phased VCF out), cli_purity.h (purity estimator).
#include "cli_common.h"
#include "cli_bam.h"
#include "cli_vcf.h"
#include "cli_purity.h"
#include "cli_extra.h"

// ------------------------------------------------------------------------------------------------ phase
static const char *kUsage =
    "Usage: longphase_amd phase [OPTION] ... READSFILE\n"
    "   -s, --snp-file=NAME   -b, --bam-file=NAME (repeatable)   -r, --reference=NAME   -o, --out-prefix=NAME (result)   -t, --threads=Num (1)\n"
    "   --ont | --pb   --indels   -q MAPQ(1)  -p baseQuality(12)  -e edgeWeight(0.1)  -a connectAdjacent(35)  -d distance(300000)\n"
    "   -1 edgeThreshold(0.7)  -L overlapThreshold(0.2)  -m readConfidence(0.65)  -n snpConfidence(0.75)  --gpu=ID (0)\n"
    "   --deepsomatic_output   the SNP file is a DeepSomatic VCF: keep FILTER=GERMLINE records, genotype them from AD / VAF (writes <prefix>_preprocessed.vcf)\n"
    "   --indelQuality=N       with --indels: indels below this QUAL are left out (logged to <prefix>_removed_indels.log, FILTER INDEL_QUAL_FILTERED)\n"
    "   --dot                  write <chromosome>.dot (the connected pairs of the graph) into the working directory; not with --sv-file / --mod-file\n"
    "   --sv-file=NAME  --mod-file=NAME   co-phase structural variants / modcall records (outputs <prefix>_SV.vcf, <prefix>_mod.vcf)   -w svWindow(20)  -h svThreshold(0.1)\n"
    "   --host-inflate | --gpu-inflate   BGZF inflate with zlib on the -t host threads / on the GPU (default: GPU for one BAM of 256 MiB or more)\n"
    "   --no-index       ignore <bam>.bai: make the whole file resident on the GPU instead of one contig at a time\n"
    "   --group-bytes=N  indexed input: consecutive contigs are uploaded and inflated together up to N compressed bytes (phase: 16 GiB, haplotag: 8 GiB)\n"
    "   --workers-per-gpu=N  indexed input: contig groups in flight per GPU, each with its own context and stream (1)\n"
    "   --gpus=N         deal the contigs onto N GPUs (devices --gpu, --gpu+1, ... modulo the number present); needs the .bai index\n";

// The commands end with _exit once their outputs are closed and the GPU context is destroyed (the ROCm runtime's static teardown costs ~0.1 s;
// LPS_CLI_NO_FAST_EXIT=1 takes the full way out, e.g. under a profiler that reports from an exit handler).  The context IS destroyed first: device
// memory left to the process exit is given back by the driver from a work queue after the exit, and the next process on the GPU waits for it.

static double g_main_entered = 0;

static int phase_main(int argc, char **argv, const std::string &command) {
    std::vector<std::function<void(lps_params &)>> over; bool indels = false;
    std::string snp, ref, prefix = "result", sv_file, mod_file;
    std::vector<std::string> bams;
    int threads = 1, gpu = 0, n_gpus = 1, sv_window = 20, indel_quality = 0; double sv_threshold = 0.1;
    uint64_t group_bytes = 16ull << 30; int workers_per_gpu = 1;       // indexed BAM: contigs are taken in groups of about this many compressed bytes (16 GiB inflate to 50 - 60 GB of the 288; the 12.4 GB sample of bench.py as ONE group: 1.23 s against 1.53 s in two), by this many concurrent workers per GPU
    bool ont = false, pb = false, host_inflate = false, gpu_inflate = false, no_index = false, deepsomatic = false, dot = false;
    auto need = [&](int &i) -> std::string { if (i + 1 >= argc) { std::cerr << kUsage; exit(1); } return argv[++i]; };
    for (int i = 2; i < argc; ++i) {
        std::string a = argv[i], v; size_t eq = a.find('=');
        if (a.rfind("--", 0) == 0 && eq != std::string::npos) { v = a.substr(eq + 1); a = a.substr(0, eq); }
        auto val = [&]() { return v.empty() ? need(i) : v; };
        if (a == "-s" || a == "--snp-file") snp = val();
        else if (a == "-b" || a == "--bam-file") bams.push_back(val());
        else if (a == "-r" || a == "--reference") ref = val();
        else if (a == "-o" || a == "--out-prefix") prefix = val();
        else if (a == "-t" || a == "--threads") threads = std::stoi(val());
        else if (a == "--ont") ont = true; else if (a == "--pb") pb = true;
        else if (a == "--indels") { indels = true; over.push_back([](lps_params &P) { P.phase_indel = 1; }); }
        else if (a == "-q" || a == "--mappingQuality") { const auto x = std::stoi(val());
            over.push_back([x](lps_params &P) { P.mapping_quality = x; });
            }
        else if (a == "-p" || a == "--baseQuality") { const auto x = std::stoi(val()); over.push_back([x](lps_params &P) { P.base_quality = x; }); }
        else if (a == "-e" || a == "--edgeWeight") { const auto x = std::stod(val()); over.push_back([x](lps_params &P) { P.edge_weight = x; }); }
        else if (a == "-a" || a == "--connectAdjacent") { const auto x = std::stoi(val());
            over.push_back([x](lps_params &P) { P.connect_adjacent = x; });
            }
        else if (a == "-d" || a == "--distance") { const auto x = std::stoi(val()); over.push_back([x](lps_params &P) { P.distance = x; }); }
        else if (a == "-1" || a == "--edgeThreshold") { const auto x = std::stod(val());
            over.push_back([x](lps_params &P) { P.edge_threshold = x; });
            }
        else if (a == "-L" || a == "--overlapThreshold") { const auto x = std::stod(val());
            over.push_back([x](lps_params &P) { P.overlap_threshold = x; });
            }
        else if (a == "-m" || a == "--readConfidence") { const auto x = std::stod(val());
            over.push_back([x](lps_params &P) { P.read_confidence = x; });
            }
        else if (a == "-n" || a == "--snpConfidence") { const auto x = std::stod(val());
            over.push_back([x](lps_params &P) { P.snp_confidence = x; });
            }
        else if (a == "-x" || a == "--mismatchRate") (void)val();
        else if (a == "--sv-file") sv_file = val();
        else if (a == "--mod-file") mod_file = val();
        else if (a == "-w" || a == "--svWindow") sv_window = std::stoi(val());
        else if (a == "-h" || a == "--svThreshold") sv_threshold = std::stod(val());
        else if (a == "--gpu") gpu = std::stoi(val());
        else if (a == "--gpus") n_gpus = std::max(1, std::stoi(val()));
        else if (a == "--group-bytes") group_bytes = (uint64_t)std::stoull(val());
        else if (a == "--upload-threads") setenv("LPS_UPLOAD_THREADS", val().c_str(), 1);      // host threads that fill the upload's page-locked pieces (library default 12)
        else if (a == "--workers-per-gpu") workers_per_gpu = std::max(1, std::stoi(val()));
        else if (a == "--host-inflate") host_inflate = true;
        else if (a == "--gpu-inflate") gpu_inflate = true;
        else if (a == "--no-index") no_index = true;
        else if (a == "--help") { std::cout << kUsage; return 0; }
        else if (a == "--deepsomatic_output") deepsomatic = true;
        else if (a == "--indelQuality") indel_quality = std::stoi(val());
        else if (a == "--dot") dot = true;
        else if (a == "--no-walk-ahead") setenv("LPS_CLI_NO_WALK_AHEAD", "1", 1);      // (A/B switch: the BGZF header walk inside the load, not ahead of it)
        else { std::cerr << "longphase_amd: unknown option " << a << "\n" << kUsage; return 1; }
    }
    if (dot && (!sv_file.empty() || !mod_file.empty())) die("longphase_amd: --dot together with --sv-file / --mod-file is not supported by the GPU path; use the reference binary");
    if (!sv_file.empty()) {                                             // Phasing.cpp:304-318
        if (sv_window < 0) { std::cerr << "longphase_amd phase: invalid svWindow. value: " << sv_window << "\n please check -w, --svWindow=Num\n"; return 1; }
        if (sv_threshold < 0 || sv_threshold > 1) { std::cerr << "longphase_amd phase: invalid svThreshold. value: " << sv_threshold << "\n this value need: 0~1, please check -h, --svThreshold=[0~1]\n"; return 1; }
    }
    if (snp.empty() || bams.empty() || ref.empty()) { std::cerr << "longphase_amd phase: missing arguments\n" << kUsage; return 1; }
    if (ont == pb) { std::cerr << "longphase_amd phase: missing arguments. --ont or --pb\n" << kUsage; return 1; }   // Phasing.cpp:175-183
    over.push_back([ont](lps_params &P) { P.is_ont = ont; });

    Lps L; lps_ctx *ctx = nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now(); double t_lib = 0, t_ctx_ready = 0;
    std::thread gpu_init([&] { if (!L.load()) return; t_lib = now(); lps_params P; L.default_params(&P); for (auto &f : over) f(P); ctx = L.create(gpu, &P); if (!ctx) L.error = "cannot create a GPU context (no CPU fallback)"; else L.set_stage_timing(ctx, 0); t_ctx_ready = now(); });
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{gpu_init};
    // One BAM large enough for the GPU path is opened NOW: when the whole file is one contig group anyway (or has no index), its BGZF headers are
    // walked on helper threads beside the VCF parse and the HIP start-up - host-only code compiled into this program (cli_bam.h, csrc/lps_bgzf_walk.h),
    // so the walk waits neither for the library nor for the list of wanted contigs, and the first load finds its table made.
    GpuBam gb;
    if (!host_inflate && !gpu_inflate && bams.size() == 1) host_inflate = file_bytes(bams[0]) < kGpuInflateMinBytes;
    // small file: zlib on the host overlaps the GPU start-up
    if (bams.size() == 1 && !host_inflate) {
        gb.open_file(bams[0], !no_index);
        if (n_gpus == 1 && workers_per_gpu == 1 && (!gb.indexed || (uint64_t)gb.fsz <= group_bytes)) gb.walk_whole_file();
    }
    std::vector<std::string> vcf_lines;
    if (!read_lines(snp, vcf_lines)) die("ERROR: Cannot open vcf file " + snp);
    if (deepsomatic) {                                                  // PhasingProcess.cpp:47-61: the filtered, re-genotyped copy IS the SNP file from here on
        std::vector<std::string> pre; preprocess_deepsomatic(vcf_lines, pre);
        std::ofstream o(prefix + "_preprocessed.vcf"); if (!o) die("Fail to open output VCF: " + prefix + "_preprocessed.vcf");
        for (const std::string &l : pre) o << l << "\n";
        vcf_lines.swap(pre);
    }
    std::vector<std::string> chr_order; std::map<std::string, ChrVariants> vars;
    IndelQual iq;
    if (indels && indel_quality > 0) { iq.threshold = indel_quality; iq.log.open(prefix + "_removed_indels.log"); if (iq.log.is_open()) iq.log << "#CHROM\tPOS\tREF\tALT\tQUAL\n"; }
    parse_vcf(vcf_lines, indels, chr_order, vars, &iq, threads);
    if (iq.log.is_open()) iq.log.close();
    // SV rows, then MOD rows: each reader drops what sits on a row of the tables read before it (PhasingProcess.cpp:69-79)
    std::vector<std::string> sv_lines, mod_lines; SvTable svt; ModTable modt;
    if (!sv_file.empty()) { if (!read_lines(sv_file, sv_lines)) die("ERROR: Cannot open vcf file " + sv_file); svt.parse(sv_lines, vars); }
    if (!mod_file.empty()) { if (!read_lines(mod_file, mod_lines)) die("ERROR: Cannot open vcf file " + mod_file); modt.parse(mod_lines, vars, svt); }
    const bool co_phase = !sv_lines.empty() || !mod_lines.empty();
    std::map<std::string, int> want; for (auto &kv : vars) if (!kv.second.pos.empty()) want[kv.first] = 1;
    // the reference sequences are first needed when a contig's table goes to the GPU: the FASTA is read beside the GPU start-up and the upload +
    // inflate of the first BAM blocks (need_fasta joins the reader)
    std::map<std::string, std::string> seqs; std::thread fasta_thread([&] { read_fasta(ref, vars, seqs); });
    std::once_flag fasta_once; auto need_fasta = [&] { std::call_once(fasta_once, [&] { fasta_thread.join(); }); };
    const double t_text = now();
    // one BAM: BGZF inflate, record discovery and record decode all run on the GPU; several BAMs (or --host-inflate): zlib on `-t` host threads
    const bool gpu_input = bams.size() == 1 && !host_inflate;
    std::vector<BamFile> files(gpu_input ? 0 : bams.size());
    for (size_t b = 0; b < files.size(); ++b) files[b].load(bams[b], threads, want);
    const double t_bam = now();
    // with a .bai next to the BAM only the blocks of one contig are resident at a time (any file size, per-contig sharding); without, the whole file.
    // The BGZF header walk of the first load needs no GPU: it runs on a helper thread while the HIP runtime is still coming up
    if (gpu_input) {
        if (!gb.indexed) gb.walk_ahead(L, 0, gb.fsz);
        else if (n_gpus == 1 && workers_per_gpu == 1) {
            std::vector<std::string> list; for (const std::string &c : chr_order) if (want.count(c)) list.push_back(c);
            std::sort(list.begin(), list.end(), [&](const std::string &a, const std::string &b) { return gb.tid_of(a) < gb.tid_of(b); });
            const auto groups = gb.plan_groups(list, group_bytes);
            if (!groups.empty()) gb.walk_group_ahead(L, groups.front());
        }
    }

    gpu_init.join();
    if (!ctx) die("longphase_amd: " + L.error);
    const double t_ctx = now();
    if (gpu_input && !gb.indexed) gb.load_all(L, ctx);
    const double t_gin = now();
    std::map<std::string, std::map<int32_t, Phased>> res; std::mutex res_mu;
    // packed SNP table of the whole genome (pos i32 | ref0 u8 | alt0 u8 | ref_len u16 | alt_len u16, contig after contig): worker 0 builds it from the parsed
    // VCF; with --gpus N it reaches the other GPUs by one RCCL broadcast (lps_comm_bcast) and every worker reads its contigs' rows from ITS copy
    // dev: the table where ncclBroadcast left it on this worker's GPU (lps_comm_bcast_to_device) - the rows then go to the context as device
    // pointers (lps_set_variants_device), no copy back to the host; else the rows are read from buf (host)
    struct Packed { std::vector<uint8_t> buf; size_t n = 0; std::map<std::string, std::pair<size_t, size_t>> where; const uint8_t *dev = nullptr;
        const uint8_t *base() const { return dev ? dev : buf.data(); }
        const int32_t *pos() const { return (const int32_t *)base();
            } const uint8_t *r0() const { return base() + 4 * n;
            } const uint8_t *a0() const { return base() + 5 * n;
            }
        const uint16_t *rl() const { return (const uint16_t *)(base() + 6 * n);
            } const uint16_t *al() const { return (const uint16_t *)(base() + 8 * n);
            } };
    Packed table0;
    { size_t n = 0; for (const std::string &c : chr_order) { table0.where[c] = {n, vars[c].pos.size()}; n += vars[c].pos.size(); }
      table0.n = n; table0.buf.assign(10 * n + 16, 0);
      int32_t *pp = (int32_t *)table0.buf.data();
      uint8_t *r0 = table0.buf.data() + 4 * n, *a0 = table0.buf.data() + 5 * n;
      uint16_t *rl = (uint16_t *)(table0.buf.data() + 6 * n), *al = (uint16_t *)(table0.buf.data() + 8 * n);
      for (const std::string &c : chr_order) { const ChrVariants &cv = vars[c]; const size_t o = table0.where[c].first;
          for (size_t i = 0; i < cv.pos.size(); ++i) { pp[o + i] = cv.pos[i];
              r0[o + i] = (uint8_t)cv.ref[i][0];
              a0[o + i] = (uint8_t)cv.alt[i][0];
              rl[o + i] = (uint16_t)cv.ref[i].size();
              al[o + i] = (uint16_t)cv.alt[i].size();
              } } }
    // names and name ranks of the contigs of a loaded group, made ahead of the contig loop: the names come off the GPU in one go per contig, the ranks
    // (a sort of the contig's read names: 2.5 ms for 25 k) are computed by a BOUNDED pool of helper threads (at most -t of them, whatever the
    // number of contigs in the group: a genome with hundreds of decoy contigs does not start hundreds of sorts at once) that walks the group in contig
    // order while the contigs before are phased
    struct NamesAhead { std::vector<char> store; std::vector<uint32_t> off; std::vector<std::pair<const char *, size_t>> names; std::vector<uint32_t> id;
                        std::mutex m; std::condition_variable cv; bool done = false;
                        void wait() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return done; }); } };
    struct RankPool { std::vector<NamesAhead *> items; std::atomic<size_t> next{0}; std::vector<std::thread> th;
                      void start(int n) { for (int k = 0; k < n; ++k) th.emplace_back([this] { for (;;) { const size_t i = next++; if (i >= items.size()) break;
                              NamesAhead *a = items[i]; rank_names(a->names, a->id);
                              std::lock_guard<std::mutex> lk(a->m); a->done = true; a->cv.notify_all(); } }); }     // (notified under the lock: the waiter may destroy *a as soon as it sees done)
                      ~RankPool() { for (auto &t : th) if (t.joinable()) t.join(); } };                              // joins on every way out of the scope, an exception included
    std::map<lps_ctx *, std::map<std::string, std::unique_ptr<NamesAhead>>> ahead; std::mutex ahead_mu;
    static std::atomic<long long> ns_names{0}, ns_rank{0}, ns_setup{0}, ns_push{0}, ns_phase{0}, ns_merge{0}, ns_result{0};      // where a contig's host time goes (LPS_CLI_DEBUG)
    auto tick_ns = [] { return std::chrono::steady_clock::now(); };
    auto tock_ns = [](std::atomic<long long> &acc,
            std::chrono::steady_clock::time_point t0) { acc += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); };
    auto run_contig = [&](lps_ctx *ctx, GpuBam &gb, const std::string &chr, const Packed &tab) {   // PhasingProcess.cpp:113-173, one contig on one GPU
        ChrVariants &cv = vars[chr];
        need_fasta();
        if (cv.pos.empty() || !seqs.count(chr)) return;
        // names of all files of this contig ranked together (one read name = one merged row, whatever file it came from)
        std::vector<std::pair<const char *, size_t>> names;
        std::vector<const ContigRecords *> parts;
        std::vector<char> name_store;
        std::vector<uint32_t> name_off;
        std::pair<int64_t, int64_t> gr{0, 0};
        auto t_st = tick_ns();
        std::unique_ptr<NamesAhead> pre;
        if (gpu_input) {
            { auto it = gb.range.find(chr); if (it == gb.range.end()) return; gr = it->second; }   // whole file, or the group loaded by the caller
            { std::lock_guard<std::mutex> lk(ahead_mu); auto a = ahead.find(ctx); if (a != ahead.end()) { auto b = a->second.find(chr); if (b != a->second.end()) { pre = std::move(b->second); a->second.erase(b); } } }
            if (pre) { pre->wait(); names.swap(pre->names); }
            else gb.names(L, ctx, gr.first, gr.second, name_store, name_off, names);
        }
        tock_ns(ns_names, t_st);
        for (BamFile &f : files) { auto it = f.contigs.find(chr);
            if (it == f.contigs.end() || it->second.rec_off.empty()) { parts.push_back(nullptr);
                continue;
                }
            parts.push_back(&it->second);
            for (size_t i = 0; i < it->second.rec_off.size(); ++i) { size_t l;
                const char *nm = f.name_of(it->second, i, l);
                names.emplace_back(nm, l);
                } }
        if (names.empty()) return;
        t_st = tick_ns();
        std::vector<uint32_t> name_id; if (pre) name_id.swap(pre->id); else rank_names(names, name_id);
        tock_ns(ns_rank, t_st); t_st = tick_ns();
        const size_t to = tab.where.at(chr).first;
        lps_variant_table vt{};
        vt.n = (int64_t)cv.pos.size();
        vt.pos = tab.pos() + to;
        vt.ref0 = tab.r0() + to;
        vt.alt0 = tab.a0() + to;
        vt.ref_len = tab.rl() + to;
        vt.alt_len = tab.al() + to;
        const std::string &sq = seqs[chr];
        if (L.begin_chromosome(ctx) || (tab.dev ? L.set_variants_device(ctx, &vt) : L.set_variants(ctx, &vt)) || L.set_reference(ctx, sq.data(), (int64_t)sq.size())) die(std::string("longphase_amd: ") + L.last_error(ctx));
        ExtraRows xr;
        if (co_phase) { xr.build(svt, modt, chr, names, name_id, sv_window, sv_threshold); if (xr.any() && L.set_extra_variants(ctx, &xr.x)) die(std::string("longphase_amd: ") + chr + ": " + L.last_error(ctx)); }
        size_t at = 0;
        tock_ns(ns_setup, t_st); t_st = tick_ns();
        if (gpu_input && L.push_bam_resident(ctx, gr.first, gr.second, name_id.data())) die(std::string("longphase_amd: ") + L.last_error(ctx));
        for (size_t b = 0; b < files.size(); ++b) {                  // BAM files in -b order (ParsingBam.cpp:1252)
            if (!parts[b]) continue;
            const ContigRecords &c = *parts[b];
            if (L.push_bam_records(ctx, files[b].z.data + c.lo, (int64_t)(c.hi - c.lo), c.rec_off.data(), (int64_t)c.rec_off.size(), name_id.data() + at))
                die(std::string("longphase_amd: ") + L.last_error(ctx));
            at += c.rec_off.size();
        }
        std::vector<int32_t> ps(cv.pos.size()); std::vector<uint8_t> gt(cv.pos.size());
        lps_phase_result pr{(int64_t)cv.pos.size(), ps.data(), gt.data()};
        tock_ns(ns_push, t_st); t_st = tick_ns();
        if (L.phase_chromosome(ctx, &pr)) die(std::string("longphase_amd: ") + L.last_error(ctx));
        tock_ns(ns_phase, t_st); t_st = tick_ns();
        if (dot) {
            // --dot: <chr>.dot in the working directory, two lines per CONNECTED pair of edgeConnectResult in the order it visits them
            // (PhasingGraph.cpp:286-418 with findBestEdgePair :166-228; written by writingDotFile :1031-1047).  A node the walk gives no haplotype - a
            // gap beyond `distance`, a tie behind the last connection - is skipped with all its pairs; a pair is connected when its cells do not tie
            // and their similarity ratio does not exceed the threshold.  From the graph the GPU built: the edge matrix and the vote scan's haplotypes
            lps_params P; L.default_params(&P); for (auto &f : over) f(P);
            const int64_t N = L.dump_graph(ctx, nullptr, nullptr, 0);
            if (N < 0) die(std::string("longphase_amd: ") + L.last_error(ctx));
            const int A = P.connect_adjacent;
            std::vector<int32_t> node((size_t)N + 1), blk((size_t)N + 1); std::vector<float> edge((size_t)N * (size_t)A * 4 + 4); std::vector<int8_t> hp((size_t)N + 1);
            if (N && (L.dump_graph(ctx, node.data(), edge.data(), N) != N || L.dump_votes(ctx, hp.data(), blk.data(), N) != N)) die(std::string("longphase_amd: ") + L.last_error(ctx));
            std::ofstream d;
            if (N > 0) d.open(chr + ".dot");                                   // no read reached a variant: the reference leaves the chromosome before the graph (PhasingProcess.cpp:143-146)
            if (N <= 0) {}
            else if (!d) std::cerr << "Fail to open write file: " << chr << ".vcf\n";            // (the reference's message names .vcf)
            else {
                d << "digraph G {\n";
                for (int64_t i = 0; i + 1 < N; ++i) {
                    if (!hp[(size_t)i]) continue;
                    for (int k = 0; k < A && i + 1 + k < N; ++k) {
                        const float *c = &edge[((size_t)i * (size_t)A + (size_t)k) * 4];
                        const float para = c[0] + c[3], cross = c[1] + c[2];
                        if (para == cross) continue;
                        const double esr = (double)std::min(para, cross) / (double)std::max(para, cross);
                        if (esr > P.edge_threshold) continue;
                        const int a1 = para > cross ? 1 : 2, a2 = 3 - a1;
                        const long long sp = (long long)cv.pos[(size_t)node[(size_t)i]] + 1, tp = (long long)cv.pos[(size_t)node[(size_t)(i + 1 + k)]] + 1;
                        d << sp << ".1\t->\t" << tp << "." << a1 << "\n" << sp << ".2\t->\t" << tp << "." << a2 << "\n";
                    }
                }
                d << "}\n";
            }
        }
        tock_ns(ns_merge, t_st); t_st = tick_ns();                     // (ns_merge: --dot)
        std::map<int32_t, Phased> rc;
        for (size_t i = 0; i < cv.pos.size(); ++i) if (ps[i]) rc.emplace_hint(rc.end(), cv.pos[i], Phased{ps[i], gt[i] ? '1' : '0', gt[i] ? '0' : '1'});   // (positions ascend: appended)
        if (xr.any()) {                                              // the reference keeps ONE result map keyed by position for all three files
            std::vector<int32_t> sps(xr.sv_pos.size()), mps(xr.mod_pos.size()); std::vector<uint8_t> sgt(xr.sv_pos.size()), mgt(xr.mod_pos.size());
            lps_phase_result rs{(int64_t)sps.size(), sps.data(), sgt.data()}, rm{(int64_t)mps.size(), mps.data(), mgt.data()};
            if (L.get_extra_result(ctx, &rs, &rm)) die(std::string("longphase_amd: ") + L.last_error(ctx));
            for (size_t i = 0; i < sps.size(); ++i) if (sps[i]) rc[xr.sv_pos[i]] = Phased{sps[i], sgt[i] ? '1' : '0', sgt[i] ? '0' : '1'};
            for (size_t i = 0; i < mps.size(); ++i) if (mps[i]) rc[xr.mod_pos[i]] = Phased{mps[i], mgt[i] ? '1' : '0', mgt[i] ? '0' : '1'};
        }
        { std::lock_guard<std::mutex> lk(res_mu); res[chr].swap(rc); std::cerr << "(" << chr << ")"; }
        tock_ns(ns_result, t_st);
    };
    // contigs never interact: with --gpus N and an indexed BAM they are dealt longest-first onto N contexts, one host thread + one
    // GPU each, every worker uploading only the BGZF blocks of its own contigs.  No data-path collective; results meet in the VCF writer.
    // On ONE GPU, too, several workers (host thread + context + stream each) can take contig groups side by side (--workers-per-gpu): one uploads
    // its next group's BGZF blocks while another's are inflated and a third's kernels phase.  MEASURED on the 16-contig, 8.3 GB sample of bench.py
    // (profiles/e2e_whole_node.py): 1 worker with the whole file as one group 1.44 s, 2 - 8 workers with 0.5 - 1 GiB groups 1.61 - 1.74 s - one
    // large upload and one inflate launch over all blocks beat several small ones that compete for the link and the GPU.  The default stays one.
    int n_workers = 1;
    if (gpu_input && gb.indexed) n_workers = n_gpus * workers_per_gpu;
    else { workers_per_gpu = 1; if (n_gpus > 1) std::cerr << "longphase_amd: --gpus needs one BAM with its .bai index; running on one GPU\n"; n_gpus = 1; }
    { size_t with_rows = 0; for (const std::string &c : chr_order) with_rows += !vars[c].pos.empty();
      while (workers_per_gpu > 1 && (size_t)(n_gpus * workers_per_gpu) > with_rows) --workers_per_gpu;       // no more workers than contigs to deal out
      n_workers = n_gpus * workers_per_gpu; }
    std::vector<std::vector<std::string>> share((size_t)n_workers);
    {
        std::vector<std::string> by_size(chr_order);
        std::stable_sort(by_size.begin(), by_size.end(), [&](const std::string &a, const std::string &b) { return vars[a].pos.size() > vars[b].pos.size(); });
        std::vector<size_t> load((size_t)n_workers, 0);
        if (n_workers == 1) share[0] = chr_order;
        else for (const std::string &c : by_size) { const size_t g = (size_t)(std::min_element(load.begin(), load.end()) - load.begin());
            share[g].push_back(c);
            load[g] += vars[c].pos.size() + 1;
            }
    }
    std::vector<std::thread> workers;
    const int n_dev = std::max(1, L.device_count());
    auto run_share = [&](lps_ctx *cx, GpuBam &g, std::vector<std::string> list, const Packed &tab) {
        if (!gpu_input || !g.indexed) { for (const std::string &c : list) run_contig(cx, g, c, tab); return; }
        std::sort(list.begin(), list.end(), [&](const std::string &a, const std::string &b) { return g.tid_of(a) < g.tid_of(b); });
        // file order, so that neighbours share an upload
        const auto groups = g.plan_groups(list, group_bytes);
        for (size_t gi = 0; gi < groups.size(); ++gi) { const auto &grp = groups[gi];
            g.load_group(L, cx, grp);
            if (gi + 1 < groups.size()) g.walk_group_ahead(L, groups[gi + 1]);       // the next group's header walk beside this group's contigs
            RankPool pool;                                                // (declared before the contig loop: joined when the group is done or on a throw)
            const double tn0 = now();
            if (files.empty()) {                                          // one BAM: all names of a contig are the GPU's
                std::map<std::string, std::unique_ptr<NamesAhead>> mine;
                for (const std::string &c : grp) { auto it = g.range.find(c); if (it == g.range.end() || !vars.count(c) || vars[c].pos.empty()) continue;
                    std::unique_ptr<NamesAhead> a(new NamesAhead());
                    g.names(L, cx, it->second.first, it->second.second, a->store, a->off, a->names);
                    pool.items.push_back(a.get());
                    mine[c] = std::move(a); }
                { std::lock_guard<std::mutex> lk(ahead_mu); ahead[cx] = std::move(mine); }
                pool.start((int)std::min<size_t>((size_t)std::max(1, threads), pool.items.size()));
            }
            const double tn1 = now();
            for (const std::string &c : grp) run_contig(cx, g, c, tab);
            if (getenv("LPS_CLI_DEBUG")) fprintf(stderr, "[cli] group of %zu contigs: names off the GPU %.3f s, contig loop %.3f s\n", grp.size(), tn1 - tn0, now() - tn1);
            for (auto &t : pool.th) if (t.joinable()) t.join();            // (every item has been ranked: each contig waited for its own)
            { std::lock_guard<std::mutex> lk(ahead_mu); ahead.erase(cx); }
        }
    };
    // the one collective: a communicator over the workers' GPUs (ncclCommInitAll); fails when two workers share a device (rehearsal on fewer GPUs
    // than --gpus) - the workers then read the table worker 0 holds, in this one address space
    // ONE communicator rank per GPU: worker k * workers_per_gpu leads GPU k, the workers beside it read the device buffer it received
    auto dev_of = [&](int g) { return (gpu + g / workers_per_gpu) % n_dev; };
    std::vector<lps_comm *> comms((size_t)n_gpus, nullptr); bool have_comm = false;
    // (LPS_CLI_BCAST_ALWAYS: a communicator even for one GPU, so that the broadcast -> lps_set_variants_device path runs on a one-GPU box)
    if (n_gpus > 1 || getenv("LPS_CLI_BCAST_ALWAYS")) { std::vector<int> devs; for (int k = 0; k < n_gpus; ++k) devs.push_back(dev_of(k * workers_per_gpu));
        have_comm = std::set<int>(devs.begin(), devs.end()).size() == devs.size() && L.comm_create_all(n_gpus, devs.data(), comms.data()) == 0;
        if (have_comm) std::cerr << "longphase_amd: RCCL communicator over " << L.comm_size(comms[0]) << " GPUs\n";
        else std::cerr << "longphase_amd: no RCCL communicator (" << (std::set<int>(devs.begin(), devs.end()).size() == devs.size() ? L.comm_last_error() : "workers share a device") << "); workers read the host table\n";
        }
    std::vector<Packed> gpu_tab((size_t)n_gpus); std::vector<char> gpu_tab_ready((size_t)n_gpus, 0); std::mutex tab_mu; std::condition_variable tab_cv;
    auto obtain_table = [&](int g) -> const Packed & {                     // every worker calls this once (collective among the GPUs' leaders)
        if (!have_comm) return table0;
        const int k = g / workers_per_gpu;
        if (g % workers_per_gpu == 0) {
            // root's host table -> the communicator's device buffer on every GPU; each GPU's contexts take their contigs' rows from there
            double ms = 0; void *d = nullptr;
            if (L.comm_bcast_to_device(comms[(size_t)k], k == 0 ? table0.buf.data() : nullptr, (int64_t)table0.buf.size(), 0, &d, k == 0 ? &ms : nullptr)) die(std::string("longphase_amd: ") + L.comm_last_error());
            if (k == 0) fprintf(stderr, "longphase_amd: SNP table broadcast, %zu bytes, %.3f ms\n", table0.buf.size(), ms);
            { std::lock_guard<std::mutex> lk(tab_mu); Packed &m = gpu_tab[(size_t)k]; m.n = table0.n; m.where = table0.where; m.dev = (const uint8_t *)d; gpu_tab_ready[(size_t)k] = 1; }
            tab_cv.notify_all();
        } else { std::unique_lock<std::mutex> lk(tab_mu); tab_cv.wait(lk, [&] { return gpu_tab_ready[(size_t)k] != 0; }); }
        return gpu_tab[(size_t)k];
    };
    for (int g = 1; g < n_workers; ++g) workers.emplace_back([&, g] {
        lps_params P; L.default_params(&P); for (auto &f : over) f(P);
        lps_ctx *cx = L.create(dev_of(g), &P); if (!cx) die("longphase_amd: cannot create a GPU context for worker " + std::to_string(g));
        L.set_stage_timing(cx, 0);
        GpuBam gg; gg.open_file(bams[0], true);
        const Packed &tab = obtain_table(g);
        run_share(cx, gg, share[(size_t)g], tab);
        { std::lock_guard<std::mutex> lk(res_mu); gb.t_inflate += gg.t_inflate; gb.t_scan += gg.t_scan; }     // (summed over the workers: they overlap, the sum can exceed the wall time)
        L.destroy(cx); gg.close_file();
    });
    { const Packed &tab = obtain_table(0); run_share(ctx, gb, share[0], tab); }
    for (auto &w : workers) w.join();
    for (lps_comm *cm : comms) if (cm) L.comm_destroy(cm);
    std::cerr << "\n";
    need_fasta();
    // the context is given back beside the VCF writer (25 ms for the buffers of a 12 GB group.  Left to the process exit, the driver gives the device
    // memory back from a work queue AFTER the exit - and the next process on the GPU pays for it)
    std::thread destroyer([&] { L.destroy(ctx); });
    struct JoinD { std::thread &t; ~JoinD() { if (t.joinable()) t.join(); } } join_destroyer{destroyer};
    const double t_gpu = now();
    write_vcf(vcf_lines, prefix + ".vcf", res, vars, command, &iq);
    if (!sv_file.empty()) write_sv_vcf(sv_lines, prefix + "_SV.vcf", res, svt, command);          // PhasingProcess.cpp:191-203
    if (!mod_file.empty()) write_mod_vcf(mod_lines, prefix + "_mod.vcf", res, modt, command);
    destroyer.join();
    if (gpu_input) fprintf(stderr, "%s | vcf read (fasta beside the gpu start-up) %.3fs | wait for gpu context %.3fs | map bam+header%s %.3fs | upload+gpu inflate %.3fs | gpu record scan %.3fs | names+decode+phase %.3fs | write vcf %.3fs | total %.3fs\n",
                           gb.indexed ? "contig groups (indexed)" : "whole file", t_text - t_begin - gb.t_map, t_ctx - t_bam, gb.indexed ? "+index" : "", gb.t_map, gb.t_inflate, gb.t_scan,
                           t_gpu - t_gin - (gb.indexed ? gb.t_inflate + gb.t_scan : 0.0), now() - t_gpu, now() - t_begin);
    else fprintf(stderr, "vcf+fasta read %.3fs | bam inflate+walk %.3fs | wait for gpu context %.3fs | upload+phase %.3fs | write vcf %.3fs | total %.3fs\n", t_text - t_begin,
                 t_bam - t_text, t_ctx - t_bam, t_gpu - t_ctx, now() - t_gpu, now() - t_begin);
    if (getenv("LPS_CLI_DEBUG")) fprintf(stderr, "[cli] per-contig host time summed: names %.3fs, ranks %.3fs, table + reference %.3fs, push %.3fs, phase %.3fs, result map %.3fs\n", ns_names / 1e9, ns_rank / 1e9, ns_setup / 1e9, ns_push / 1e9, ns_phase / 1e9, ns_result / 1e9);
    if (getenv("LPS_CLI_DEBUG")) fprintf(stderr, "[cli] start-up: library loaded after %.3f s, gpu context ready after %.3f s, text inputs parsed after %.3f s\n", t_lib - t_begin, t_ctx_ready - t_begin, t_text - t_begin);
    if (getenv("LPS_CLI_DEBUG")) fprintf(stderr, "[cli] main entered at %.3f, left at %.3f (epoch seconds: what the caller's clock shows before and after is start-up and exit)\n", g_main_entered, epoch_now());
    fflush(stderr);
    if (getenv("LPS_CLI_NO_FAST_EXIT")) return 0;                       // e.g. under a profiler that writes its report from an exit handler
    _exit(0);   // outputs are closed and flushed; skip the ROCm runtime's static teardown (~0.1 s)
}

static const char *kTagUsage =
    "Usage: longphase_amd haplotag [OPTION] ... READSFILE\n"
    "   -s, --snp-file=NAME   -b, --bam-file=NAME   -r, --reference=NAME   -o, --out-prefix=NAME (result)   -t, --threads=Num (1)\n"
    "   --tagSupplementary   -q qualityThreshold(1)   -p percentageThreshold(0.6)   --gpu=ID (0)\n"
    "   --sv-file=NAME  --mod-file=NAME   phased SV / modcall VCFs: every phased record votes for the reads it lists (RNAMES= / MR=)\n"
    "   --gpus=N (deal the contigs onto N GPUs, devices --gpu, --gpu+1, ...; needs <bam>.bai and the GPU writer; output contigs stay in VCF-header order)\n"
    "   --host-inflate | --gpu-inflate (zlib on the -t threads / GPU inflate + GPU writer; default: GPU for a BAM of 256 MiB or more)   --no-index (ignore <bam>.bai, keep the whole file on the GPU)\n"
    "   --host-deflate (tag splice + zlib deflate on the -t threads instead of the GPU writer; implied by --host-inflate)\n"
    "   --compress-level=N (6)   --compress-strategy=rle|default|huffman (rle: packed bases and qualities hold few LZ77 matches; about 2 % larger\n"
    "                             output than zlib's default strategy at several times the speed; `default` = what htslib writes)\n";

static int haplotag_main(int argc, char **argv, const std::string &command) {
    std::vector<std::function<void(lps_params &)>> over;
    std::string snp, ref, bam, prefix = "result", sv_file, mod_file;
    int threads = 1, gpu = 0, n_gpus = 1, level = 6, strategy = Z_RLE;
    bool host_inflate = false, gpu_inflate = false, no_index = false, host_deflate = false;
    uint64_t group_bytes = 8ull << 30;
    auto need = [&](int &i) -> std::string { if (i + 1 >= argc) { std::cerr << kTagUsage; exit(1); } return argv[++i]; };
    for (int i = 2; i < argc; ++i) {
        std::string a = argv[i], v; size_t eq = a.find('=');
        if (a.rfind("--", 0) == 0 && eq != std::string::npos) { v = a.substr(eq + 1); a = a.substr(0, eq); }
        auto val = [&]() { return v.empty() ? need(i) : v; };
        if (a == "-s" || a == "--snp-file") snp = val();
        else if (a == "-b" || a == "--bam-file") bam = val();
        else if (a == "-r" || a == "--reference") ref = val();
        else if (a == "-o" || a == "--out-prefix") prefix = val();
        else if (a == "-t" || a == "--threads") threads = std::stoi(val());
        else if (a == "--tagSupplementary") over.push_back([](lps_params &P) { P.tag_supplementary = 1; });
        else if (a == "--sv-file") sv_file = val();
        else if (a == "--mod-file") mod_file = val();
        else if (a == "-q" || a == "--qualityThreshold") { const auto x = std::stoi(val());
            over.push_back([x](lps_params &P) { P.mapping_quality = x; });
            }
        else if (a == "-p" || a == "--percentageThreshold") { const auto x = std::stod(val());
            over.push_back([x](lps_params &P) { P.percentage_threshold = x; });
            }
        else if (a == "--gpu") gpu = std::stoi(val());
        else if (a == "--gpus") n_gpus = std::max(1, std::stoi(val()));
        else if (a == "--host-inflate") host_inflate = true;
        else if (a == "--gpu-inflate") gpu_inflate = true;
        else if (a == "--no-index") no_index = true;
        else if (a == "--host-deflate") host_deflate = true;
        else if (a == "--group-bytes") group_bytes = (uint64_t)std::stoull(val());
        else if (a == "--upload-threads") setenv("LPS_UPLOAD_THREADS", val().c_str(), 1);      // host threads that fill the upload's page-locked pieces (library default 12)
        else if (a == "--compress-level") level = std::stoi(val());
        else if (a == "--compress-strategy") { const std::string x = val();
            strategy = x == "default" ? Z_DEFAULT_STRATEGY : x == "rle" ? Z_RLE : x == "huffman" ? Z_HUFFMAN_ONLY : -1;
            if (strategy < 0) die("longphase_amd: --compress-strategy is one of default, rle, huffman");
            }
        else if (a == "--help") { std::cout << kTagUsage; return 0; }
        else if (a == "--cram" || a == "--region" || a == "--log") die("longphase_amd: " + a + " is not supported by the GPU path; use the reference binary");
        else { std::cerr << "longphase_amd: unknown option " << a << "\n" << kTagUsage; return 1; }
    }
    if (snp.empty() || bam.empty() || ref.empty()) { std::cerr << "longphase_amd haplotag: missing arguments\n" << kTagUsage; return 1; }

    Lps L; lps_ctx *ctx = nullptr;
    std::thread gpu_init([&] { if (!L.load()) return; lps_params P; L.default_params(&P); for (auto &f : over) f(P); ctx = L.create(gpu, &P); if (!ctx) L.error = "cannot create a GPU context (no CPU fallback)"; else L.set_stage_timing(ctx, 0); });
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{gpu_init};
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    std::vector<std::string> vcf_lines;
    if (!read_lines(snp, vcf_lines)) die("Fail to open vcf: " + snp);
    std::vector<std::string> chr_vec; std::map<std::string, int> chr_len; std::map<std::string, std::map<int32_t, PhasedRow>> rows;
    parse_phased_vcf(vcf_lines, chr_vec, chr_len, rows);
    // phased SV / MOD files: votes per read NAME, one table for the whole genome (HaplotagProcess.cpp:72-90)
    ReadVotes votes; bool have_votes = false;
    for (int k = 0; k < 2; ++k) { const std::string &fn = k ? mod_file : sv_file; if (fn.empty()) continue;
        std::vector<std::string> ls; if (!read_lines(fn, ls)) die("Fail to open vcf: " + fn);
        parse_read_votes(ls, k ? "MR=" : "RNAMES=", votes); have_votes = true; }
    // per alignment of a contig, in record order: looked up by name; handed to the library before the scoring call
    auto set_votes = [&](lps_ctx *cx, size_t n, const std::function<std::string(size_t)> &name_of) {
        if (!have_votes) return;
        std::vector<int32_t> v1(n, 0), v2(n, 0);
        for (size_t i = 0; i < n; ++i) { auto it = votes.find(name_of(i)); if (it != votes.end()) { v1[i] = it->second[0]; v2[i] = it->second[1]; } }
        if (L.set_read_votes(cx, v1.data(), v2.data(), (int64_t)n)) die(std::string("longphase_amd: ") + L.last_error(cx));
    };
    std::map<std::string, ChrVariants> want_seq; std::map<std::string, int> want;
    for (const std::string &c : chr_vec) { want[c] = 1; want_seq[c]; }
    // (the FASTA is read beside the BAM upload + inflate; need_fasta joins the reader before the first contig's table goes to the GPU)
    std::map<std::string, std::string> seqs; std::thread fasta_thread([&] { read_fasta(ref, want_seq, seqs); });
    std::once_flag fasta_once; auto need_fasta = [&] { std::call_once(fasta_once, [&] { fasta_thread.join(); }); };
    const double t_text = now();
    // default: BGZF inflate + record discovery on the GPU, the inflated stream is copied back once for the writer; --host-inflate: zlib on -t threads
    BamFile in; GpuBam gb; size_t in_cap = 0;
    if (!host_inflate && !gpu_inflate) host_inflate = file_bytes(bam) < kGpuInflateMinBytes;   // small file: the host path overlaps the GPU start-up
    if (host_inflate) host_deflate = true;                             // the GPU writer works on the stream the GPU inflated
    const bool gpu_writer = !host_deflate;
    if (host_inflate) in.load(bam, threads, want);
    std::vector<std::vector<std::string>> groups;
    if (!host_inflate) {                                                // the first load's BGZF header walk runs while the HIP runtime comes up
        gb.open_file(bam, !no_index);
        if (!gb.indexed) gb.walk_ahead(L, 0, gb.fsz);
        else { groups = gb.plan_groups(chr_vec, group_bytes); if (!groups.empty()) gb.walk_group_ahead(L, groups.front()); }
    }
    const double t_bam = now();
    gpu_init.join();
    if (!ctx) die("longphase_amd: " + L.error);
    const double t_ctx = now();
    // copy [0, total) of the stream resident on the GPU into `in.z` (2 MiB pages, touched by a few threads first so the D2H does not fault serially)
    auto copy_back = [&](int64_t total) {
        if ((size_t)total + 64 > in_cap) { free(in.z.data);
            const size_t huge = 2u << 20;
            in_cap = ((size_t)total + 64 + huge - 1) / huge * huge + (in_cap >> 1);
            in.z.data = (uint8_t *)aligned_alloc(huge, in_cap / huge * huge + huge);
            if (!in.z.data) die("ERROR: out of memory");
            madvise(in.z.data, in_cap, MADV_HUGEPAGE);
            std::vector<std::thread> th; const int nt = std::max(1, std::min(threads, 8)); const size_t slice = (in_cap + nt - 1) / nt;
            for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { const size_t a = std::min(in_cap, slice * t), e = std::min(in_cap,
                    a + slice); for (size_t p = a; p < e; p += 4096) in.z.data[p] = 0; });
            for (auto &x : th) x.join(); }
        in.z.size = (size_t)total;
        if (L.bgzf_read(ctx, 0, total, in.z.data)) die(std::string("ERROR: ") + L.last_error(ctx));
    };
    if (!host_inflate) {
        if (!gb.indexed) {
            gb.load_all(L, ctx);
            if (!gpu_writer) copy_back(gb.total);
            for (auto &kv : gb.range) {
                if (!want.count(kv.first)) continue;
                ContigRecords &c = in.contigs[kv.first]; c.rec_off.resize((size_t)kv.second.second);
                if (!gpu_writer && L.bam_record_offsets(ctx, kv.second.first, kv.second.second, c.rec_off.data())) die(std::string("ERROR: ") + L.last_error(ctx));
                c.lo = 0; c.hi = (uint64_t)gb.total;                   // offsets stay absolute (lo = 0)
            }
        }
    }
    const double t_gin = now();

    BgzfWriter w; w.open(prefix + ".bam", threads, level, strategy);
    {   // header: the input's text + one @PG line (BamFileRAII, src/haplotag/HaplotagParsingBam.cpp:45), then the reference table unchanged
        const uint8_t *d = host_inflate ? in.z.data : gb.header.data(); const uint32_t l_text = rd32(d + 4);
        std::string text((const char *)d + 8, l_text); while (!text.empty() && text.back() == '\0') text.pop_back();
        if (!text.empty() && text.back() != '\n') text += '\n';
        std::string last_pg;
        for (size_t p = 0; p < text.size();) { const size_t e = text.find('\n', p);
            const std::string ln = text.substr(p, e - p);
            if (ln.compare(0, 3, "@PG") == 0) { const size_t i = ln.find("\tID:");
                if (i != std::string::npos) last_pg = ln.substr(i + 4, ln.find('\t', i + 4) - i - 4);
                } p = e == std::string::npos ? text.size() : e + 1;
            }
        text += "@PG\tID:longphase_amd\tPN:longphase_amd" + (last_pg.empty() ? std::string() : "\tPP:" + last_pg) + "\tVN:" + kVersion + "\tCL:" + command + "\n";
        std::vector<uint8_t> h;
        h.insert(h.end(), d, d + 4);
        const uint32_t lt = (uint32_t)text.size();
        for (int k = 0; k < 4; ++k) h.push_back((uint8_t)(lt >> (8 * k)));
        h.insert(h.end(), text.begin(), text.end());
        size_t p = 8 + (size_t)l_text;
        const size_t ref_begin = p;
        const uint32_t n_ref = rd32(d + p);
        p += 4;
        for (uint32_t i = 0; i < n_ref; ++i) p += 4 + (size_t)rd32(d + p) + 4;
        h.insert(h.end(), d + ref_begin, d + p);
        w.append(h.data(), h.size());
        if (gpu_writer) w.flush_partial();
        // the header becomes its own BGZF block(s); the GPU writes whole blocks per contig
    }
    unsigned long long st_count[8] = {0}, hp_count[3] = {0};
    double t_score = 0, t_splice = 0, t_deflate = 0, t_load = 0, t_mark = now(); std::vector<uint8_t> zbuf; uint8_t *pin[2] = {nullptr, nullptr};
    // ---- --gpus N (indexed BAM, GPU writer): scoring is per read and the output BGZF blocks of a contig depend on nothing but that contig, so the contigs
    //      are dealt longest-first onto N workers (one host thread + one GPU + its own view of the file each); every worker inflates, scores, re-tags and
    //      deflates its contigs, the main thread writes the finished contigs in VCF-header order (BGZF members concatenate).  No data-path collective.
    if (n_gpus > 1 && !(gpu_writer && !host_inflate && gb.indexed)) std::cerr << "longphase_amd: haplotag --gpus needs the indexed BAM and the GPU writer; running on one GPU\n";
    if (n_gpus > 1 && gpu_writer && !host_inflate && gb.indexed) {
        struct Done { std::vector<uint8_t> z; bool ready = false; unsigned long long st[8] = {0}, hpc[3] = {0}; };
        std::vector<Done> done(chr_vec.size()); std::mutex mu; std::condition_variable cv;
        auto tag_contig = [&](lps_ctx *cx, GpuBam &g, const std::string &chr, Done &d) {
            auto gi = g.range.find(chr);
            if (gi == g.range.end() || gi->second.second == 0) return;
            const size_t n = (size_t)gi->second.second;
            auto ri = rows.find(chr);
            std::vector<uint8_t> status(n, 5), hp(n, 0); std::vector<int32_t> h1(n), h2(n), psmin(n), pq(n), psv(n); std::vector<uint8_t> nps(n);
            std::vector<uint32_t> name_id(n, 0);                        // haplotag does not group by read name
            if (ri != rows.end() && !ri->second.empty()) {
                need_fasta();
                if (!seqs.count(chr)) die("ERROR: contig " + chr + " is missing from the reference FASTA");
                const size_t m = ri->second.size();
                std::vector<int32_t> pos(m), ps(m); std::vector<uint8_t> r0(m), a0(m), hpa(m); std::vector<uint16_t> rl(m), al(m); size_t k = 0;
                for (auto &kv : ri->second) { pos[k] = kv.first;
                    r0[k] = (uint8_t)kv.second.ref[0];
                    a0[k] = (uint8_t)kv.second.alt[0];
                    rl[k] = (uint16_t)kv.second.ref.size();
                    al[k] = (uint16_t)kv.second.alt.size();
                    hpa[k] = kv.second.hp1_is_alt;
                    ps[k] = kv.second.ps;
                    ++k;
                    }
                lps_variant_table vt{};
                vt.n = (int64_t)m;
                vt.pos = pos.data();
                vt.ref0 = r0.data();
                vt.alt0 = a0.data();
                vt.ref_len = rl.data();
                vt.alt_len = al.data();
                vt.hp1_is_alt = hpa.data();
                vt.phase_set = ps.data();
                const std::string &sq = seqs.at(chr);
                lps_haplotag_result hr{(int64_t)n, status.data(), h1.data(), h2.data(), nps.data(), psmin.data(), hp.data(), pq.data(), psv.data()};
                if (L.begin_chromosome(cx) || L.set_variants(cx, &vt) || L.set_reference(cx, sq.data(), (int64_t)sq.size()) || L.push_bam_resident(cx, gi->second.first, (int64_t)n, name_id.data()))
                    die(std::string("longphase_amd: ") + L.last_error(cx));
                if (have_votes) { std::vector<char> store; std::vector<uint32_t> noff; std::vector<std::pair<const char *, size_t>> nm;
                    g.names(L, cx, gi->second.first, gi->second.second, store, noff, nm);
                    set_votes(cx, n, [&](size_t i) { return std::string(nm[i].first, nm[i].second); }); }
                if (L.haplotag_chromosome(cx, &hr)) die(std::string("longphase_amd: ") + L.last_error(cx));
            } else if (L.begin_chromosome(cx) || L.push_bam_resident(cx, gi->second.first, (int64_t)n, name_id.data())) die(std::string("longphase_amd: ") + L.last_error(cx));
            int64_t nb = 0;
            if (L.haplotag_write_bgzf(cx, status.data(), hp.data(), psv.data(), pq.data(), nullptr, 0, &nb)) die(std::string("longphase_amd: ") + L.last_error(cx));
            d.z.resize((size_t)nb);
            if (L.bgzf_deflate_fetch(cx, d.z.data(), (int64_t)d.z.size(), nullptr)) die(std::string("longphase_amd: ") + L.last_error(cx));
            for (size_t i = 0; i < n; ++i) { ++d.st[status[i] & 7]; if (status[i] == 0) ++d.hpc[hp[i] < 3 ? hp[i] : 0]; }
        };
        // deal by compressed size of the contig's blocks (what the index knows), longest first
        std::vector<size_t> order(chr_vec.size()); for (size_t i = 0; i < order.size(); ++i) order[i] = i;
        auto weight = [&](size_t i) -> uint64_t { const int t = gb.tid_of(chr_vec[i]);
            return (t < 0 || (size_t)t >= gb.voff.size()) ? 0 : ((gb.voff[(size_t)t].second >> 16) - (gb.voff[(size_t)t].first >> 16)) + 1;
            };
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return weight(a) > weight(b); });
        std::vector<std::vector<size_t>> share((size_t)n_gpus); std::vector<uint64_t> load((size_t)n_gpus, 0);
        for (size_t i : order) { const size_t g = (size_t)(std::min_element(load.begin(), load.end()) - load.begin());
            share[g].push_back(i);
            load[g] += weight(i);
            }
        const int n_dev = std::max(1, L.device_count());
        auto run_worker = [&](int g, lps_ctx *cx, GpuBam &gg) {
            std::vector<size_t> mine = share[(size_t)g];
            std::sort(mine.begin(), mine.end(), [&](size_t a, size_t b) { return gg.tid_of(chr_vec[a]) < gg.tid_of(chr_vec[b]); });
            // file order: neighbours share an upload
            std::vector<std::string> names; for (size_t i : mine) names.push_back(chr_vec[i]);
            std::map<std::string, size_t> idx; for (size_t i : mine) idx[chr_vec[i]] = i;
            for (auto &grp : gg.plan_groups(names, group_bytes)) {
                gg.load_group(L, cx, grp);
                for (const std::string &c : grp) { Done &d = done[idx[c]];
                    tag_contig(cx, gg, c, d);
                    { std::lock_guard<std::mutex> lk(mu);
                        d.ready = true;
                        } cv.notify_all();
                    }
            }
            for (size_t i : mine) { std::lock_guard<std::mutex> lk(mu);
                if (!done[i].ready) { done[i].ready = true;
                    cv.notify_all();
                    } }     // contigs without records in the file
        };
        std::vector<std::thread> workers;
        for (int g = 1; g < n_gpus; ++g) workers.emplace_back([&, g] {
            lps_params P; L.default_params(&P); for (auto &f : over) f(P);
            lps_ctx *cx = L.create((gpu + g) % n_dev, &P);
            if (!cx) die("longphase_amd: cannot create a GPU context for worker " + std::to_string(g));
            L.set_stage_timing(cx, 0);
            GpuBam gg; gg.open_file(bam, true);
            run_worker(g, cx, gg);
            L.destroy(cx); gg.close_file();
        });
        std::thread first([&] { run_worker(0, ctx, gb); });
        for (size_t i = 0; i < chr_vec.size(); ++i) {                     // the writer: contigs in VCF-header order
            std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return done[i].ready; }); lk.unlock();
            if (!done[i].z.empty()) w.write_raw(done[i].z.data(), done[i].z.size());
            for (int k = 0; k < 8; ++k) st_count[k] += done[i].st[k]; for (int k = 0; k < 3; ++k) hp_count[k] += done[i].hpc[k];
            std::vector<uint8_t>().swap(done[i].z);
            std::cerr << "(" << chr_vec[i] << ")";
        }
        first.join(); for (auto &x : workers) x.join();
        std::cerr << "\n";
        need_fasta(); w.finish();
        L.destroy(ctx);
        unsigned long long total = 0; for (int k = 0; k < 8; ++k) total += st_count[k];
        fprintf(stderr, "total alignment %llu | tagged %llu (HP1 %llu, HP2 %llu) | untagged: low mapq %llu, unmapped %llu, secondary %llu, supplementary %llu, no variant %llu, beyond last variant %llu, judged %llu\n",
                total, hp_count[1] + hp_count[2], hp_count[1], hp_count[2], st_count[1], st_count[2], st_count[3], st_count[4], st_count[5], st_count[6], hp_count[0]);
        fprintf(stderr, "%d workers (one GPU each, contigs dealt by compressed size) | total %.3fs\n", n_gpus, now() - t_begin);
        fflush(stderr);
        if (getenv("LPS_CLI_NO_FAST_EXIT")) return 0;
        _exit(0);
    }

    // output order = chr_vec order; a group = a run of consecutive contigs in it
    for (const std::string &chr : chr_vec) {                          // contigs in VCF-header order (HaplotagProcess.cpp:94-97)
        if (!host_inflate && gb.indexed) {                              // indexed input: the group of consecutive contigs this one belongs to is loaded when its first member comes up
            const double tl = now();
            if (!gb.range.count(chr)) {
                for (size_t gi = 0; gi < groups.size(); ++gi) if (!groups[gi].empty() && groups[gi].front() == chr) { auto &grp = groups[gi];
                    gb.load_group(L, ctx, grp);
                    if (gi + 1 < groups.size()) gb.walk_group_ahead(L, groups[gi + 1]);   // the next group's header walk beside this group's contigs
                    if (!gpu_writer) copy_back(gb.total);
                    for (const std::string &m : grp) { auto it = gb.range.find(m);
                        if (it == gb.range.end()) continue;
                        ContigRecords &cc = in.contigs[m];
                        cc.rec_off.resize((size_t)it->second.second);
                        cc.lo = 0;
                        cc.hi = (uint64_t)gb.total;
                        if (!gpu_writer && L.bam_record_offsets(ctx, it->second.first, it->second.second, cc.rec_off.data())) die(std::string("ERROR: ") + L.last_error(ctx));
                        }
                }
            }
            t_load += now() - tl; t_mark = now();
        }
        auto ci = in.contigs.find(chr);
        if (ci == in.contigs.end() || ci->second.rec_off.empty()) continue;
        const ContigRecords &c = ci->second; const size_t n = c.rec_off.size(); const uint8_t *base = in.z.data + c.lo;
        auto ri = rows.find(chr);
        std::vector<uint8_t> status(n, 5), hp(n, 0); std::vector<int32_t> h1(n), h2(n), psmin(n), pq(n), psv(n); std::vector<uint8_t> nps(n);
        if (ri != rows.end() && !ri->second.empty()) {
            need_fasta();
                if (!seqs.count(chr)) die("ERROR: contig " + chr + " is missing from the reference FASTA");
            const size_t m = ri->second.size();
            std::vector<int32_t> pos(m), ps(m); std::vector<uint8_t> r0(m), a0(m), hpa(m); std::vector<uint16_t> rl(m), al(m); size_t k = 0;
            for (auto &kv : ri->second) { pos[k] = kv.first;
                r0[k] = (uint8_t)kv.second.ref[0];
                a0[k] = (uint8_t)kv.second.alt[0];
                rl[k] = (uint16_t)kv.second.ref.size();
                al[k] = (uint16_t)kv.second.alt.size();
                hpa[k] = kv.second.hp1_is_alt;
                ps[k] = kv.second.ps;
                ++k;
                }
            lps_variant_table vt{};
            vt.n = (int64_t)m;
            vt.pos = pos.data();
            vt.ref0 = r0.data();
            vt.alt0 = a0.data();
            vt.ref_len = rl.data();
            vt.alt_len = al.data();
            vt.hp1_is_alt = hpa.data();
            vt.phase_set = ps.data();
            const std::string &sq = seqs[chr];
            std::vector<uint32_t> name_id(n, 0);                        // haplotag does not group by read name
            lps_haplotag_result hr{(int64_t)n, status.data(), h1.data(), h2.data(), nps.data(), psmin.data(), hp.data(), pq.data(), psv.data()};
            if (L.begin_chromosome(ctx) || L.set_variants(ctx, &vt) || L.set_reference(ctx, sq.data(), (int64_t)sq.size()) ||
                (host_inflate ? L.push_bam_records(ctx, base, (int64_t)(c.hi - c.lo), c.rec_off.data(), (int64_t)n, name_id.data())
                              : L.push_bam_resident(ctx, gb.range[chr].first, (int64_t)n, name_id.data())))
                die(std::string("longphase_amd: ") + L.last_error(ctx));
            if (have_votes) {
                if (host_inflate) set_votes(ctx, n, [&](size_t i) { size_t l; const char *nm = in.name_of(c, i, l); return std::string(nm, l); });
                else { std::vector<char> store; std::vector<uint32_t> noff; std::vector<std::pair<const char *, size_t>> nm;
                    gb.names(L, ctx, gb.range[chr].first, gb.range[chr].second, store, noff, nm);
                    set_votes(ctx, n, [&](size_t i) { return std::string(nm[i].first, nm[i].second); }); }
            }
            if (L.haplotag_chromosome(ctx, &hr)) die(std::string("longphase_amd: ") + L.last_error(ctx));
        } else if (gpu_writer) {                                          // no variants on this contig: its records are still written (untouched)
            std::vector<uint32_t> name_id(n, 0);
            if (L.begin_chromosome(ctx) || L.push_bam_resident(ctx, gb.range[chr].first, (int64_t)n, name_id.data())) die(std::string("longphase_amd: ") + L.last_error(ctx));
        }
        t_score += now() - t_mark; t_mark = now();
        if (gpu_writer) {                                                 // tag splice + BGZF deflate on the GPU; the host only writes the finished blocks
            int64_t nb = 0;
            if (L.haplotag_write_bgzf(ctx, status.data(), hp.data(), psv.data(), pq.data(), nullptr, 0, &nb)) die(std::string("longphase_amd: ") + L.last_error(ctx));
            t_splice += now() - t_mark; t_mark = now();
            {   // the blocks leave the GPU in 64-MiB pieces through two page-locked buffers; a writer thread puts piece k on disk while piece k+1 arrives
                const int64_t piece = 64ll << 20;
                if (!pin[0]) { pin[0] = (uint8_t *)L.host_alloc((size_t)piece); pin[1] = (uint8_t *)L.host_alloc((size_t)piece); if (!pin[0] || !pin[1]) die("longphase_amd: cannot allocate page-locked host memory"); }
                w.wait_writer();
                std::thread wr; int k = 0;
                for (int64_t off = 0; off < nb; off += piece, k ^= 1) {
                    const int64_t len = std::min(piece, nb - off);
                    if (L.bgzf_deflate_fetch_range(ctx, off, len, pin[k])) die(std::string("longphase_amd: ") + L.last_error(ctx));
                    if (wr.joinable()) wr.join();                       // piece k-1 is on disk: its buffer is free for piece k+1
                    uint8_t *src = pin[k];
                    wr = std::thread([&w, src, len] { w.write_raw(src, (size_t)len); });
                }
                if (wr.joinable()) wr.join();
            }
            for (size_t i = 0; i < n; ++i) { ++st_count[status[i] & 7]; if (status[i] == 0) ++hp_count[hp[i] < 3 ? hp[i] : 0]; }
            t_deflate += now() - t_mark; t_mark = now();
            std::cerr << "(" << chr << ")";
            continue;
        }
        // second pass, records stay in input order (the reference's tagRead is single-threaded for that reason, HaplotagProcess.cpp:138):
        // (1) output length of every record, (2) prefix sum, (3) records written into place by a thread pool, (4) block-parallel deflate
        std::vector<uint64_t> out_off(n + 1, 0);
        auto aux_of = [&](const uint8_t *r) { const uint32_t l_name = r[8], n_cig = r[12] | (r[13] << 8), l_seq = rd32(r + 16);
            return r + 32 + l_name + 4ull * n_cig + (l_seq + 1) / 2 + l_seq;
            };
        auto parallel_records = [&](const std::function<void(size_t, size_t)> &fn) {
            const int nt = (int)std::max<size_t>(1, std::min<size_t>(threads, n / 256 + 1)); std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { fn(n * t / nt, n * (t + 1) / nt); });
            for (auto &x : th) x.join();
        };
        std::atomic<int> malformed{0};
        parallel_records([&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i) {
                const uint8_t *r = base + c.rec_off[i]; const uint32_t bs = rd32(r - 4);
                uint64_t len = 4ull + bs;
                if (status[i] == 0) {
                    bool seen[3] = {false, false, false};
                    for (const uint8_t *p = aux_of(r), *end = r + bs; p < end;) {
                        const size_t l = aux_field_len(p, end); if (!l) { malformed = 1; break; }
                        const int which = (p[0] == 'H' && p[1] == 'P') ? 0 : (p[0] == 'P' && p[1] == 'S') ? 1 : (p[0] == 'P' && p[1] == 'Q') ? 2 : -1;
                        if (which >= 0 && !seen[which]) { seen[which] = true; len -= l; }
                        p += l;
                    }
                    if (hp[i]) len += 21;
                }
                out_off[i + 1] = len;
            }
        });
        if (malformed) die("ERROR: malformed auxiliary field in " + bam);
        for (size_t i = 0; i < n; ++i) { out_off[i + 1] += out_off[i];
            ++st_count[status[i] & 7];
            if (status[i] == 0) ++hp_count[hp[i] < 3 ? hp[i] : 0];
            }
        uint8_t *ob = (uint8_t *)malloc(out_off[n] + 64); if (!ob) die("ERROR: out of memory");
        parallel_records([&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i) {
                const uint8_t *r = base + c.rec_off[i]; const uint32_t bs = rd32(r - 4); uint8_t *o = ob + out_off[i];
                if (status[i] != 0) { memcpy(o, r - 4, 4 + (size_t)bs); continue; }             // not scored: written untouched
                const uint8_t *aux = aux_of(r), *end = r + bs; uint8_t *q = o + 4;
                memcpy(q, r, (size_t)(aux - r)); q += aux - r;
                bool seen[3] = {false, false, false};
                // initFlag: the first HP, PS and PQ field each (:337-339)
                for (const uint8_t *p = aux; p < end;) {
                    const size_t l = aux_field_len(p, end);
                    const int which = (p[0] == 'H' && p[1] == 'P') ? 0 : (p[0] == 'P' && p[1] == 'S') ? 1 : (p[0] == 'P' && p[1] == 'Q') ? 2 : -1;
                    if (which >= 0 && !seen[which]) seen[which] = true; else { memcpy(q, p, l); q += l; }
                    p += l;
                }
                if (hp[i]) {                                                                     // addAuxiliaryTags (:357-361)
                    const int32_t vals[3] = {(int32_t)hp[i], psv[i], pq[i]}; const char *tags[3] = {"HP", "PS", "PQ"};
                    for (int k = 0; k < 3; ++k) { *q++ = (uint8_t)tags[k][0];
                        *q++ = (uint8_t)tags[k][1];
                        *q++ = 'i';
                        for (int b = 0; b < 4; ++b) *q++ = (uint8_t)((uint32_t)vals[k] >> (8 * b));
                        }
                }
                const uint32_t nbs = (uint32_t)(q - o) - 4; for (int b = 0; b < 4; ++b) o[b] = (uint8_t)(nbs >> (8 * b));
            }
        });
        t_splice += now() - t_mark; t_mark = now();
        w.append(ob, out_off[n]);
        free(ob);
        t_deflate += now() - t_mark; t_mark = now();
        std::cerr << "(" << chr << ")";
    }
    std::cerr << "\n";
    need_fasta(); w.finish();
    t_deflate += now() - t_mark;
    L.destroy(ctx);
    unsigned long long total = 0; for (int k = 0; k < 8; ++k) total += st_count[k];
    fprintf(stderr, "total alignment %llu | tagged %llu (HP1 %llu, HP2 %llu) | untagged: low mapq %llu, unmapped %llu, secondary %llu, supplementary %llu, no variant %llu, beyond last variant %llu, judged %llu\n",
            total, hp_count[1] + hp_count[2], hp_count[1], hp_count[2], st_count[1], st_count[2], st_count[3], st_count[4], st_count[5], st_count[6], hp_count[0]);
    fprintf(stderr, "vcf+fasta read %.3fs | %s %.3fs | wait for gpu context %.3fs | score %.3fs | %s %.3fs | %s %.3fs (%llu bytes) | total %.3fs\n",
            t_text - t_begin, host_inflate ? "host inflate+walk" : gb.indexed ? "gpu inflate+scan per contig group (indexed)" : "gpu inflate+scan+copy back", host_inflate ? t_bam - t_text : t_gin - t_ctx + t_load, t_ctx - t_bam, t_score, gpu_writer ? "gpu tag splice+deflate" : "tag splice", t_splice, gpu_writer ? "copy out+write (overlapped)" : "deflate+write", t_deflate, w.bytes_out, now() - t_begin);
    fflush(stderr);
    if (getenv("LPS_CLI_NO_FAST_EXIT")) return 0;                       // e.g. under a profiler that writes its report from an exit handler
    _exit(0);
}

// ------------------------------------------------------------------------------------------------ somatic_haplotag
// The three BAM passes run on the GPU (rows a20-a22 of SURVEY.md §8: lps_somatic_extract_normal / _tumor, lps_somatic_tag_chromosome); between them the
// caller's per-site statistics are restated here: SomaticVarCaller::setFilterParamsWithPurity :951-1060, getDenseTumorSnpInterval :1243-1355,
// somaticFeatureFilter :1062-1230, calibrateReadHP :1366-1404, calculateReadSetHP :1418-1439, statisticSomaticPosReadHP :1441-1518, getSomaticFlag :2397-2412
// (src/somatic_haplotag/SomaticVarCaller.cpp), and TumorPurityEstimator.cpp (estimate_purity below) when no --tumor-purity is given.
struct SomaticThr { float norVAF_max;
    int norDepth_min;
    float messy;
    int readCount_min;
    float hap_VAF_max;
    int hap_readCount_max, hap_somaticRead_min;
    float ivl_VAF_max;
    int ivl_readCount_max, ivl_count_min;
    float z_max;
    const char *tier;
    };
static SomaticThr somatic_thresholds(double purity) {                  // setFilterParamsWithPurity: the float members are assigned from double literals, the int locals truncate them
    if (purity >= 0.9 && purity <= 1.0) return {0.13f, 1, 1.0f, 3, 0.144f, 12, 0, 0.189f, 12, 4, 5.233f, "1.0"};
    if (purity >= 0.7 && purity < 0.9) return {0.13f, 1, 1.0f, 3, 0.130f, 10, 1, 0.133f, 10, 4, 2.676f, "0.8"};
    if (purity >= 0.5 && purity < 0.7) return {0.105f, 1, 1.0f, 1, 0.071f, 10, 0, 0.105f, 10, 4, 5.683f, "0.6"};
    if (purity >= 0.3 && purity < 0.5) return {0.117f, 1, 1.0f, 1, 0.035f, 8, 1, 0.049f, 8, 4, 3.043f, "0.4"};
    return {0.130f, 1, 1.0f, 1, 0.020f, 8, 1, 0.025f, 8, 8, 1.953f, "0.2"};
}
static int judge_somatic_read_hap(int h1, int h2, int h3, int n_ps, double thr) {   // judgeSomaticReadHap (src/haplotag/HaplotagStrategy.cpp:452-602), hpCount[4] is always 0 here
    double tMin, tMax, nMin, nMax; int maxN;
    if (h3 > 0) { tMin = 0; tMax = h3; } else { tMin = h3; tMax = 0; }
    if (h1 > h2) { nMin = h2; nMax = h1; maxN = 1; } else { nMin = h1; nMax = h2; maxN = 2; }
    const double tumSim = (tMax == 0) ? 0.0 : tMax / (tMax + tMin), norSim = (nMax == 0) ? 0.0 : nMax / (nMax + nMin);
    int hp = 0;
    if (tMax != 0) { if (tumSim >= thr) hp = norSim >= thr ? (maxN == 1 ? 5 : 7) : 3; }
    else if (nMax != 0) { if (norSim >= thr) hp = maxN; }
    if (n_ps > 1) hp = 0;
    return hp;
}

static const char *kSomUsage =
    "Usage: longphase_amd somatic_haplotag [OPTION] ... READSFILE\n"
    "   -s, --snp-file=NAME (phased normal VCF)   -b, --bam-file=NAME (normal BAM)   --tumor-snv-file=NAME   --tumor-bam-file=NAME   -r, --reference=NAME\n"
    "   --tumor-purity=Num (default: automatic estimation, written to <prefix>_purity.out)   --disableFilter   --somatic-calling-log (writes <prefix>_somatic_filter.log)\n"
    "   --output-somatic-vcf (writes <prefix>_sc.vcf: the tumor VCF with FILTER = PASS for the somatic calls, LowQual otherwise)\n"
    "   --tagSupplementary   -q qualityThreshold(1)   -p percentageThreshold(0.6)   -t threads(1)   -o out-prefix(result)   --gpu=ID (0)\n"
    "   --host-deflate   zlib (level 6, RLE) on the -t threads for the tagged BAM instead of the GPU's BGZF writer\n"
    "   --host-inflate   zlib on the -t threads for both BAMs instead of the GPU's BGZF inflate (the default for files below 256 MiB); --gpu-inflate forces the GPU\n"
    "   --gpus=N (deal the contigs onto N GPU contexts, devices --gpu, --gpu+1, ...: the three BAM passes of a contig run on its worker, purity is estimated\n"
    "             over all contigs, logs and the tagged BAM are merged in contig order; with both BAMs indexed: contig GROUPS onto the workers)\n"
    "   --group-bytes=N  both BAMs indexed (.bai): the pair is walked in groups of consecutive contigs of up to N compressed bytes, tumor + normal together\n"
    "                    (24 GiB) - only one group's inflated records are on the GPU at a time, as the reference walks the pair chromosome by chromosome;\n"
    "                    --no-index: ignore the indexes and keep both whole files on the GPU\n";

static int somatic_main(int argc, char **argv, const std::string &command) {
    std::vector<std::function<void(lps_params &)>> over;
    std::string snp, ref, nbam, tvcf, tbam, prefix = "result";
    int threads = 1, gpu = 0, n_gpus = 1; uint64_t group_bytes = 24ull << 30; bool no_index = false;   // (a pair whose compressed bytes fit one group is loaded once; ~4x that inflated, of 288 GB)
    double purity = -1, pct = 0.6;
    bool enable_filter = true, write_log = false, write_sc_vcf = false; bool host_deflate = false, raw_started = false, host_inflate = false, gpu_inflate = false;
    auto need = [&](int &i) -> std::string { if (i + 1 >= argc) { std::cerr << kSomUsage; exit(1); } return argv[++i]; };
    for (int i = 2; i < argc; ++i) {
        std::string a = argv[i], v; size_t eq = a.find('=');
        if (a.rfind("--", 0) == 0 && eq != std::string::npos) { v = a.substr(eq + 1); a = a.substr(0, eq); }
        auto val = [&]() { return v.empty() ? need(i) : v; };
        if (a == "-s" || a == "--snp-file") snp = val();
        else if (a == "-b" || a == "--bam-file") nbam = val();
        else if (a == "--tumor-snv-file") tvcf = val();
        else if (a == "--tumor-bam-file") tbam = val();
        else if (a == "-r" || a == "--reference") ref = val();
        else if (a == "-o" || a == "--out-prefix") prefix = val();
        else if (a == "-t" || a == "--threads") threads = std::stoi(val());
        else if (a == "--tagSupplementary") over.push_back([](lps_params &P) { P.tag_supplementary = 1; });
        else if (a == "-q" || a == "--qualityThreshold") { const auto x = std::stoi(val());
            over.push_back([x](lps_params &P) { P.mapping_quality = x; });
            }
        else if (a == "-p" || a == "--percentageThreshold") { pct = std::stod(val());
            const double x = pct;
            over.push_back([x](lps_params &P) { P.percentage_threshold = x; });
            }
        else if (a == "--tumor-purity") purity = std::stod(val());
        else if (a == "--disableFilter") enable_filter = false;
        else if (a == "--somatic-calling-log") write_log = true;
        else if (a == "--output-somatic-vcf") write_sc_vcf = true;
        else if (a == "--gpu") gpu = std::stoi(val());
        else if (a == "--gpus") n_gpus = std::max(1, std::stoi(val()));
        else if (a == "--host-deflate") host_deflate = true;
        else if (a == "--host-inflate") host_inflate = true;
        else if (a == "--gpu-inflate") gpu_inflate = true;
        else if (a == "--group-bytes") group_bytes = (uint64_t)std::stoull(val());
        else if (a == "--upload-threads") setenv("LPS_UPLOAD_THREADS", val().c_str(), 1);      // host threads that fill the upload's page-locked pieces (library default 12)
        else if (a == "--no-index") no_index = true;
        else if (a == "--help") { std::cout << kSomUsage; return 0; }
        else if (a == "--cram" || a == "--region" || a == "--log" || a == "--truth-vcf" || a == "--truth-bed" || a == "--benchmark-log") die("longphase_amd: " + a + " is not supported by the GPU path; use the reference binary");
        else { std::cerr << "longphase_amd: unknown option " << a << "\n" << kSomUsage; return 1; }
    }
    if (snp.empty() || nbam.empty() || tvcf.empty() || tbam.empty() || ref.empty()) { std::cerr << "longphase_amd somatic_haplotag: missing arguments\n" << kSomUsage;
        return 1;
        }
    const bool estimate = purity < 0;                                  // default: automatic estimation, as in the reference

    Lps L; lps_ctx *ctx = nullptr;
    std::thread gpu_init([&] { if (!L.load()) return; lps_params P; L.default_params(&P); for (auto &f : over) f(P); ctx = L.create(gpu, &P); if (!ctx) L.error = "cannot create a GPU context (no CPU fallback)"; else L.set_stage_timing(ctx, 0); });
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{gpu_init};
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    std::vector<std::string> nlines, tlines;
    if (!read_lines(snp, nlines)) die("Fail to open vcf: " + snp);
    if (!read_lines(tvcf, tlines)) die("Fail to open vcf: " + tvcf);
    std::vector<std::string> nchr, tchr;
    std::map<std::string, int> nlen, tlen;
    std::map<std::string, std::map<int32_t, PhasedRow>> nrows;
    std::map<std::string, std::map<int32_t, TumorRow>> trows;
    parse_phased_vcf(nlines, nchr, nlen, nrows); parse_tumor_vcf(tlines, tchr, tlen, trows);
    if (const char *dump = getenv("LPS_CLI_DUMP_TABLE")) {                // debugging aid: the parsed rows, before any GPU work
        std::ofstream o(dump);
        for (auto &c : trows) for (auto &r : c.second) o << "T\t" << c.first << "\t" << r.first << "\t" << r.second.ref << "\t" << r.second.alt << "\t" << r.second.kind << "\n";
        for (auto &c : nrows) for (auto &r : c.second) o << "N\t" << c.first << "\t" << r.first << "\t" << r.second.ref << "\t" << r.second.alt << "\t" << r.second.ps << "\t" << (int)r.second.hp1_is_alt << "\n";
    }
    for (auto &kv : tlen) { auto it = nlen.find(kv.first);
        if (it == nlen.end()) die("[ERROR] (setChrVecAndChrLength) :tumor & normal VCFs chromosome count are not the same");
        if (it->second != kv.second) die("[ERROR] (setChrVecAndChrLength) :tumor & normal VCFs chromosome length are not the same => chr: " + kv.first);
        }
    const std::vector<std::string> &chr_vec = tchr.empty() ? nchr : tchr;   // SomaticHaplotagProcess.cpp:161-174
    if (chr_vec.empty()) die("[ERROR] (setChrVecAndChrLength) :tumor & normal VCFs chromosome count are empty");
    std::map<std::string, ChrVariants> want_seq; std::map<std::string, int> want;
    for (const std::string &c : chr_vec) { want[c] = 1; want_seq[c]; }
    // Both BAMs: BGZF inflate + record discovery on the GPU, the inflated stream copied back once (the tag splice below works on host bytes); a small
    // file (or --host-inflate) goes through zlib on -t threads beside the GPU start-up.  The FASTA is read beside either.
    std::map<std::string, std::string> seqs; std::thread fasta_thread([&] { read_fasta(ref, want_seq, seqs); });
    struct FastaJoin { std::thread &t; ~FastaJoin() { if (t.joinable()) t.join(); } } fasta_join{fasta_thread};
    BamFile nin, tin;
    const bool n_gpu = !host_inflate && (gpu_inflate || file_bytes(nbam) >= kGpuInflateMinBytes), t_gpu = !host_inflate && (gpu_inflate || file_bytes(tbam) >= kGpuInflateMinBytes);
    if (!n_gpu) nin.load(nbam, threads, want);
    if (!t_gpu) tin.load(tbam, threads, want);
    bool resident = n_gpu && t_gpu && !host_deflate;
    GpuBam ngb, tgb;
    if (resident) { tgb.open_file(tbam, !no_index); ngb.open_file(nbam, !no_index);
        if (n_gpus > 1 && !(tgb.indexed && ngb.indexed)) { resident = false; tgb.close_file(); ngb.close_file(); } }   // several workers need the groups an index gives; else: records pushed from host memory
    // GROUPED (both BAMs indexed): the pair is walked in groups of consecutive contigs, as the reference walks it chromosome by chromosome
    // (src/somatic_haplotag/SomaticVarCaller.cpp:822, SomaticHaplotagProcess.cpp:54-109) - one group of the tumor BAM and the same contigs of the
    // normal BAM are uploaded, inflated and scanned, their contigs go through the passes, the next group replaces them.  HBM then holds
    // --group-bytes of the pair at a time instead of both whole files (a 50x / 25x whole-genome pair inflates to ~0.5 TB).
    std::vector<std::vector<std::string>> som_groups; bool grouped = false;
    if (resident) {
        grouped = tgb.indexed && ngb.indexed;
        if (grouped) {
            // a group is a run of chr_vec whose members are neighbours in BOTH files (contigs without records in a file do not break its run)
            auto has = [](const GpuBam &g, int t) { return t >= 0 && g.voff[(size_t)t].second > g.voff[(size_t)t].first; };
            auto span = [](const GpuBam &g, int t) -> uint64_t { return (g.voff[(size_t)t].second >> 16) - (g.voff[(size_t)t].first >> 16) + 65536; };
            auto run_ok = [&](const GpuBam &g, int last, int t) { if (t <= last) return false; for (int k = last + 1; k < t; ++k) if (has(g, k)) return false; return true; };
            int last_t = -2, last_n = -2; uint64_t bytes = 0;
            for (const std::string &c : chr_vec) {
                const int tt = tgb.tid_of(c), tn = ngb.tid_of(c);
                if (!has(tgb, tt)) continue;                              // no tumor records: nothing is tagged or written for this contig
                const bool hn = has(ngb, tn);
                const uint64_t sz = span(tgb, tt) + (hn ? span(ngb, tn) : 0);
                const bool ok = !som_groups.empty() && run_ok(tgb, last_t, tt) && (!hn || last_n == -2 || run_ok(ngb, last_n, tn)) && bytes + sz <= group_bytes;
                if (!ok) { som_groups.emplace_back(); bytes = 0; last_n = -2; }
                som_groups.back().push_back(c); bytes += sz; last_t = tt; if (hn) last_n = tn;
            }
            if (!som_groups.empty()) tgb.walk_group_ahead(L, som_groups.front());
        } else { tgb.walk_ahead(L, 0, tgb.fsz); ngb.walk_ahead(L, 0, ngb.fsz); }   // header walks beside the GPU start-up
    }
    gpu_init.join();
    if (!ctx) die("longphase_amd: " + L.error);
    // One worker and the GPU writer: both inflated streams STAY on the GPU (the tumor's in this context, the normal's in a second one that runs pass 1):
    // the passes take their records from there (lps_push_bam_resident), the tagged records are spliced and deflated there (lps_somatic_write_bgzf).
    // Otherwise (--gpus N, --host-deflate): the stream is copied back once and the contigs' records are pushed from host memory as before.
    double t_gpu_inflate = 0; lps_ctx *nctx = nullptr;
    if (resident) {
        lps_params P; L.default_params(&P); for (auto &f : over) f(P);
        nctx = L.create(gpu, &P); if (!nctx) die("longphase_amd: cannot create the GPU context of the normal BAM");
        L.set_stage_timing(nctx, 0);
        if (!grouped) { tgb.load_all(L, ctx); ngb.load_all(L, nctx); t_gpu_inflate = tgb.t_inflate + tgb.t_scan + ngb.t_inflate + ngb.t_scan; }
    } else {
        if (n_gpu) gpu_load_to_host(L, ctx, nbam, want, threads, nin, &t_gpu_inflate);
        if (t_gpu) gpu_load_to_host(L, ctx, tbam, want, threads, tin, &t_gpu_inflate);
    }
    fasta_thread.join();
    const double t_in = now();
    std::atomic<long long> ns_p1{0}, ns_p2{0}, ns_host{0}, ns_p3{0}, ns_splice{0}, ns_append{0}, ns_prep{0}, ns_purity{0}, ns_finish{0}, ns_gpu_deflate{0};      // where the time goes (summed over workers)
    auto tick = [] { return std::chrono::steady_clock::now(); };
    auto tock = [](std::atomic<long long> &acc,
            std::chrono::steady_clock::time_point t0) { acc += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); };
    SomaticThr T = somatic_thresholds(purity);
    auto announce = [&]() { if (purity <= 0 || purity > 1.0) std::cerr << "[WARNING] tumor purity is not in the range of 0.0 to 1.0: " << purity << "\n[WARNING] setting default parameters (tier " << T.tier << ")\n";
        else std::cerr << "setting filter params (tier " << T.tier << ") with tumor purity: " << purity << "\n"; };
    if (!estimate) announce();
    std::vector<PurityDatum> pdata; size_t p_initial = 0; int lcvf[5] = {0, 0, 0, 0, 0};
    std::ofstream flog;
    if (write_log) { flog.open(prefix + "_somatic_filter.log");
        if (!flog) die("Fail to open write file: " + prefix + "_somatic_filter.log");
        flog << "######################################\n# Somatic Filter Evaluation Per-Pos   #\n######################################\n"
             << "#CHROM\tPOS\tNorVAF\tNorDepth\tMixedHpReadRatio\tCaseReadCount\tTumVAF\tIntervalSnpCount\tzScore\tDenseAltSameCount\tFilteredByTINC\tFilteredByMessyRead\tFilteredByReadCount\tFilteredByHapConsistency\tFilteredByVariantCluster\tFilteredByDenseAlt\tisFilterOut\n";
             }

    BgzfWriter w; w.open(prefix + ".bam", threads, 6, Z_RLE);
    {   // header of the TUMOR BAM + one @PG line
        const uint8_t *d = resident ? tgb.header.data() : tin.z.data; const uint32_t l_text = rd32(d + 4);
        std::string text((const char *)d + 8, l_text); while (!text.empty() && text.back() == '\0') text.pop_back();
        if (!text.empty() && text.back() != '\n') text += '\n';
        std::string last_pg;
        for (size_t p = 0; p < text.size();) { const size_t e = text.find('\n', p);
            const std::string ln = text.substr(p, e - p);
            if (ln.compare(0, 3, "@PG") == 0) { const size_t i = ln.find("\tID:");
                if (i != std::string::npos) last_pg = ln.substr(i + 4, ln.find('\t', i + 4) - i - 4);
                } p = e == std::string::npos ? text.size() : e + 1;
            }
        text += "@PG\tID:longphase_amd\tPN:longphase_amd" + (last_pg.empty() ? std::string() : "\tPP:" + last_pg) + "\tVN:" + kVersion + "\tCL:" + command + "\n";
        std::vector<uint8_t> h;
        h.insert(h.end(), d, d + 4);
        const uint32_t lt = (uint32_t)text.size();
        for (int k = 0; k < 4; ++k) h.push_back((uint8_t)(lt >> (8 * k)));
        h.insert(h.end(), text.begin(), text.end());
        size_t p = 8 + (size_t)l_text;
        const size_t ref_begin = p;
        const uint32_t n_ref = rd32(d + p);
        p += 4;
        for (uint32_t i = 0; i < n_ref; ++i) p += 4 + (size_t)rd32(d + p) + 4;
        h.insert(h.end(), d + ref_begin, d + p);
        w.append(h.data(), h.size());
    }
    unsigned long long n_somatic_flag = 0, hp_hist[9] = {0}, st_count[8] = {0};
    std::map<std::string, std::set<int32_t>> somatic_pos;               // isSomaticVariant (getSomaticFlag), for --output-somatic-vcf
    // phase 0 (only when the purity has to be estimated): passes 1 and 2 over every contig feed the estimator, which needs all contigs at once;
    // phase 1: passes 1 and 2 again (milliseconds on the GPU), the caller's statistics and filters, pass 3, the writer.
    // One contig of one phase, on one GPU context.  Everything it adds to run-wide state goes to its own ContigAcc, merged in contig order by the caller - so the
    // contigs can be dealt onto several workers (--gpus N) and the outputs (filter log, tagged BAM, purity inputs) still come out in the reference's order.
    int n_workers_now = 1; uint8_t *pin[2] = {nullptr, nullptr};       // set once the workers are known (below); pin: page-locked pieces of the single-worker writer
    struct ContigAcc { std::vector<PurityDatum> pdata; size_t p_initial = 0; int lcvf[5] = {0, 0, 0, 0, 0}; std::ostringstream flog; std::set<int32_t> som;
                       unsigned long long n_flag = 0, hp_hist[9] = {0}, st_count[8] = {0}; uint8_t *out = nullptr; size_t out_bytes = 0; bool ready = false, deflated = false; };
    // What passes 1 and 2 of a contig produced, kept from the estimation phase (0) for the calling phase (1): the reference, too, extracts once and
    // holds the per-site data of every chromosome while it estimates the purity (SomaticVarCaller.cpp:822-905); the second phase then needs the tumor
    // records only (tagging pass + writer) - the normal BAM is read once, the two extraction passes run once.
    // a worker of the resident path: its context (tumor stream), the second context that holds the normal stream, its own views of the two files
    struct SomWorker { lps_ctx *ctx = nullptr, *nctx = nullptr; GpuBam *tgb = nullptr, *ngb = nullptr; size_t group_loaded = (size_t)-1; };
    struct Saved { bool have = false; std::vector<int32_t> nsite, tsite, h1, h2, h3, psmin, endp, rlen, pr_site, pr_read, wn_site; std::vector<uint8_t> tstat, thp, tnps, has, pr_hp, wn_al, wn_base;
                   std::vector<int16_t> wn_off; int64_t n_pairs = 0, n_windows = 0; };
    std::vector<Saved> saved(chr_vec.size());
    auto do_contig = [&](SomWorker &W, const std::string &chr, int phase, ContigAcc &A, Saved &SV) {
        lps_ctx *ctx = W.ctx, *nctx = W.nctx; GpuBam &tgb = *W.tgb, &ngb = *W.ngb;
        auto t_prep = tick();
        auto fail = [&]() { die(std::string("longphase_amd: ") + L.last_error(ctx)); };
        std::vector<PurityDatum> &pdata = A.pdata; size_t &p_initial = A.p_initial; int (&lcvf)[5] = A.lcvf; std::ostringstream &flog = A.flog;
        unsigned long long &n_somatic_flag = A.n_flag; unsigned long long (&hp_hist)[9] = A.hp_hist; unsigned long long (&st_count)[8] = A.st_count;
        auto ti = tin.contigs.find(chr);
        std::pair<int64_t, int64_t> t_range{0, 0}, n_range{0, 0};          // resident streams: (first record, count) of this contig
        if (resident) { auto a = tgb.range.find(chr); if (a != tgb.range.end()) t_range = a->second; auto b = ngb.range.find(chr); if (b != ngb.range.end()) n_range = b->second; }
        const bool have_t = resident ? t_range.second > 0 : (ti != tin.contigs.end() && !ti->second.rec_off.empty());
        // ---- merged table (MultiGenomeVar map): normal phased-het rows + tumor rows
        std::map<int32_t, PhasedRow> none_n; std::map<int32_t, TumorRow> none_t;
        const std::map<int32_t, PhasedRow> &nr = nrows.count(chr) ? nrows[chr] : none_n;
        const std::map<int32_t, TumorRow> &tr = trows.count(chr) ? trows[chr] : none_t;
        std::vector<int32_t> pos, ps; std::vector<uint8_t> r0, a0, hpa, role, derive, tkind; std::vector<uint16_t> rl, al;
        { auto a = nr.begin(); auto b = tr.begin();
          while (a != nr.end() || b != tr.end()) {
              const bool take_n = a != nr.end() && (b == tr.end() || a->first <= b->first), both = take_n && b != tr.end() && a->first == b->first;
              if (take_n) { pos.push_back(a->first);
                  r0.push_back((uint8_t)a->second.ref[0]);
                  a0.push_back((uint8_t)a->second.alt[0]);
                  rl.push_back((uint16_t)a->second.ref.size());
                  al.push_back((uint16_t)a->second.alt.size());
                  hpa.push_back(a->second.hp1_is_alt);
                  ps.push_back(a->second.ps);
                  role.push_back(0);
                  derive.push_back(0);
                  tkind.push_back(both ? (uint8_t)b->second.kind : 0);
                  if (both) { if (a->second.ref != b->second.ref || a->second.alt != b->second.alt) die("longphase_amd: normal and tumor VCF disagree on the alleles at " + chr + ":" + std::to_string(a->first + 1));
                      ++b;
                      } ++a;
                  }
              else { pos.push_back(b->first);
                  r0.push_back((uint8_t)b->second.ref[0]);
                  a0.push_back((uint8_t)b->second.alt[0]);
                  rl.push_back((uint16_t)b->second.ref.size());
                  al.push_back((uint16_t)b->second.alt.size());
                  hpa.push_back(0); ps.push_back(0); role.push_back(2); derive.push_back(0); tkind.push_back((uint8_t)b->second.kind); ++b; }
          } }
        const size_t nv = pos.size();
        if (!have_t) return;
        static const ContigRecords no_records;
        const ContigRecords &tc = resident ? no_records : ti->second; const size_t nt = resident ? (size_t)t_range.second : tc.rec_off.size(); const uint8_t *tbase = resident ? nullptr : tin.z.data + tc.lo;
        bool tumor_pushed = false;
        std::vector<uint8_t> status(nt, 5), hp(nt, 0); std::vector<int32_t> psv(nt, -1), pq(nt, 0);
        if (phase == 0 && !nv) return;
        if (nv) {
            if (!seqs.count(chr)) die("ERROR: contig " + chr + " is missing from the reference FASTA");
            const std::string &sq = seqs[chr];
            lps_variant_table vt{};
            vt.n = (int64_t)nv;
            vt.pos = pos.data();
            vt.ref0 = r0.data();
            vt.alt0 = a0.data();
            vt.ref_len = rl.data();
            vt.alt_len = al.data();
            vt.hp1_is_alt = hpa.data();
            vt.phase_set = ps.data();
            vt.somatic_role = role.data(); vt.derive_hp = derive.data(); vt.tumor_kind = tkind.data();
            // ---- pass 1: normal BAM (ExtractNorDataBamParser)
            tock(ns_prep, t_prep);
            auto t_stage = tick();
            const bool reuse = phase == 1 && SV.have;                       // passes 1 and 2 ran in the estimation phase
            std::vector<int32_t> nsite; if (reuse) nsite.swap(SV.nsite); else nsite.assign(nv * LPS_SITE_COUNTERS, 0);
            auto ni = nin.contigs.find(chr);
            if (reuse) {}
            else if (resident) {
                if (n_range.second > 0) { std::vector<uint32_t> nid((size_t)n_range.second, 0);
                    lps_site_counters sc{(int64_t)nv, nsite.data(), 0, nullptr};
                    if (L.begin_chromosome(nctx) || L.set_variants(nctx, &vt) || L.set_reference(nctx, sq.data(), (int64_t)sq.size()) ||
                        L.push_bam_resident(nctx, n_range.first, n_range.second, nid.data()) || L.somatic_extract_normal(nctx, &sc)) die(std::string("longphase_amd: ") + L.last_error(nctx)); }
            } else if (ni != nin.contigs.end() && !ni->second.rec_off.empty()) {
                const ContigRecords &nc = ni->second; std::vector<uint32_t> nid(nc.rec_off.size(), 0);
                lps_site_counters sc{(int64_t)nv, nsite.data(), 0, nullptr};
                if (L.begin_chromosome(ctx) || L.set_variants(ctx, &vt) || L.set_reference(ctx, sq.data(), (int64_t)sq.size()) ||
                    L.push_bam_records(ctx, nin.z.data + nc.lo, (int64_t)(nc.hi - nc.lo), nc.rec_off.data(), (int64_t)nc.rec_off.size(), nid.data()) || L.somatic_extract_normal(ctx, &sc)) fail();
            }
            tock(ns_p1, t_stage); t_stage = tick();
            // ---- pass 2: tumor BAM (ExtractTumDataBamParser)
            std::vector<int32_t> tsite, h1, h2, h3, psmin, endp, rlen;
            std::vector<uint8_t> tstat, thp, tnps, has;
            std::vector<int32_t> pr_site, pr_read, wn_site; std::vector<uint8_t> pr_hp, wn_al, wn_base; std::vector<int16_t> wn_off;
            if (reuse) { tsite.swap(SV.tsite); h1.swap(SV.h1); h2.swap(SV.h2); h3.swap(SV.h3); psmin.swap(SV.psmin); endp.swap(SV.endp); rlen.swap(SV.rlen); tstat.swap(SV.tstat); thp.swap(SV.thp);
                tnps.swap(SV.tnps); has.swap(SV.has); pr_site.swap(SV.pr_site); pr_read.swap(SV.pr_read); pr_hp.swap(SV.pr_hp); wn_site.swap(SV.wn_site); wn_al.swap(SV.wn_al); wn_base.swap(SV.wn_base); wn_off.swap(SV.wn_off); }
            else { tsite.assign(nv * LPS_TSITE_COUNTERS, 0); h1.resize(nt); h2.resize(nt); h3.resize(nt); psmin.resize(nt); endp.resize(nt); rlen.resize(nt); tstat.resize(nt); thp.resize(nt); tnps.resize(nt); has.resize(nt); }
            lps_tumor_extract_result te{};
            te.n = (int64_t)nv;
            te.site = tsite.data();
            te.n_reads = (int64_t)nt;
            te.status = tstat.data();
            te.hp1 = h1.data();
            te.hp2 = h2.data();
            te.hp3 = h3.data();
            te.hp = thp.data();
            te.n_ps = tnps.data(); te.ps_min = psmin.data(); te.end_pos = endp.data(); te.read_len = rlen.data(); te.has_site = has.data();
            { std::vector<uint32_t> tid(nt, 0);
              if (L.begin_chromosome(ctx) || L.set_variants(ctx, &vt) || L.set_reference(ctx, sq.data(), (int64_t)sq.size()) ||
                  (resident ? L.push_bam_resident(ctx, t_range.first, t_range.second, tid.data()) : L.push_bam_records(ctx, tbase, (int64_t)(tc.hi - tc.lo),
                          tc.rec_off.data(), (int64_t)nt, tid.data()))) fail();
              tumor_pushed = true;
              }
            size_t pcap = nt * 4 + 1024, wcap = nt * 64 + 4096;
            if (reuse) { te.n_pairs = SV.n_pairs; te.n_windows = SV.n_windows; SV = Saved(); }
            else for (int attempt = 0;; ++attempt) {
                pr_site.resize(pcap);
                pr_read.resize(pcap);
                pr_hp.resize(pcap);
                wn_site.resize(wcap);
                wn_al.resize(wcap);
                wn_off.resize(wcap);
                wn_base.resize(wcap);
                te.pair_capacity = (int64_t)pcap; te.pair_site = pr_site.data(); te.pair_read = pr_read.data(); te.pair_base_hp = pr_hp.data();
                te.win_capacity = (int64_t)wcap;
                te.win_site = wn_site.data();
                te.win_allele = wn_al.data();
                te.win_offset = wn_off.data();
                te.win_base = wn_base.data();
                const int rc = L.somatic_extract_tumor(ctx, &te);
                if (rc == 0) break;
                if (rc != -9 || attempt > 2) fail();
                pcap = (size_t)te.n_pairs + 16; wcap = (size_t)te.n_windows + 16;
            }
            if (phase == 0) {                                                // TumorPurityEstimator::buildPurityFeatureValueVec (LCVF) over the touched sites
                for (size_t v = 0; v < nv; ++v) { if (!tkind[v]) continue;
                    const int32_t *c = &tsite[v * LPS_TSITE_COUNTERS], *n = &nsite[v * LPS_SITE_COUNTERS];
                    long rh = 0;
                    for (int k = 15; k < 24; ++k) rh += c[k];
                    if (!(c[6] > 0 || rh > 0)) continue;
                    ++p_initial;
                    auto imb = [](int a, int b) { const int t = a + b;
                        if (a > 0 && b > 0) return a > b ? (double)a / (double)t : (double)b / (double)t;
                        if (a == 0 && b == 0) return 0.0;
                        return 1.0;
                        };
                    const double tr_ratio = tkind[v] == 4 ? 0.0 : imb(c[16], c[17]), nr_ratio = imb(n[LPS_SC_READHP_H1], n[LPS_SC_READHP_H2]);
                    const int ncount = n[LPS_SC_READHP_H1] + n[LPS_SC_READHP_H2];
                    const double npct = (n[LPS_SC_DEPTH] == 0 || ncount == 0) ? 0.0 : (double)ncount / (double)n[LPS_SC_DEPTH];
                    if (nr_ratio == 0.0f) ++lcvf[0];
                    else if (tr_ratio == 0.0f) ++lcvf[1];
                    else if (nr_ratio >= 0.7f) ++lcvf[2];
                    else if (ncount <= 5) ++lcvf[3];
                    else if (npct <= 0.7f) ++lcvf[4];
                    else pdata.push_back(PurityDatum{tr_ratio, ncount}); }
                // kept for the calling phase (a few bytes per alignment, a few hundred per touched site)
                SV.have = true; SV.n_pairs = te.n_pairs; SV.n_windows = te.n_windows;
                pr_site.resize((size_t)te.n_pairs); pr_read.resize((size_t)te.n_pairs); pr_hp.resize((size_t)te.n_pairs);
                wn_site.resize((size_t)te.n_windows); wn_al.resize((size_t)te.n_windows); wn_off.resize((size_t)te.n_windows); wn_base.resize((size_t)te.n_windows);
                pr_site.shrink_to_fit(); pr_read.shrink_to_fit(); pr_hp.shrink_to_fit(); wn_site.shrink_to_fit(); wn_al.shrink_to_fit(); wn_off.shrink_to_fit(); wn_base.shrink_to_fit();
                SV.nsite.swap(nsite); SV.tsite.swap(tsite); SV.h1.swap(h1); SV.h2.swap(h2); SV.h3.swap(h3); SV.psmin.swap(psmin); SV.endp.swap(endp); SV.rlen.swap(rlen); SV.tstat.swap(tstat); SV.thp.swap(thp);
                SV.tnps.swap(tnps); SV.has.swap(has); SV.pr_site.swap(pr_site); SV.pr_read.swap(pr_read); SV.pr_hp.swap(pr_hp); SV.wn_site.swap(wn_site); SV.wn_al.swap(wn_al); SV.wn_base.swap(wn_base); SV.wn_off.swap(wn_off);
                return;
            }
            tock(ns_p2, t_stage); t_stage = tick();
            // ---- host stages.  "exists": the site was touched by a tumor read (std::map entries of somaticPosInfo)
            std::vector<int> sites; std::vector<int> site_of(nv, -1);
            for (size_t v = 0; v < nv; ++v) { if (!tkind[v]) continue;
                const int32_t *c = &tsite[v * LPS_TSITE_COUNTERS];
                long rh = 0;
                for (int k = 15; k < 24; ++k) rh += c[k];
                if (c[6] > 0 || rh > 0) { site_of[v] = (int)sites.size();
                    sites.push_back((int)v);
                    } }
            const size_t ns = sites.size();
            if (getenv("LPS_CLI_DEBUG")) { int byk[5] = {0}, tab[5] = {0};
                for (size_t v = 0; v < nv; ++v) ++tab[tkind[v] < 5 ? tkind[v] : 0];
                for (int v : sites) ++byk[tkind[(size_t)v]];
                fprintf(stderr, "[debug] %s: table %zu rows (tumor kinds %d/%d/%d/%d), touched sites %zu (%d/%d/%d/%d), pairs %lld, windows %lld, tumor reads %zu\n", chr.c_str(), nv, tab[1], tab[2], tab[3], tab[4], ns, byk[1], byk[2], byk[3], byk[4], (long long)te.n_pairs, (long long)te.n_windows, nt);
                }
            std::vector<std::vector<std::pair<int, int>>> pairs(ns);       // site -> (read, baseHP): tumorPosReadCorrBaseHP
            for (int64_t k = 0; k < te.n_pairs; ++k) { const int sidx = site_of[(size_t)pr_site[(size_t)k]];
                if (sidx < 0) die("[ERROR] pair at a site that does not exist");
                pairs[(size_t)sidx].push_back({pr_read[(size_t)k], pr_hp[(size_t)k]});
                }
            std::vector<float> meanAlt(ns, 0.0f), zScore(ns, 0.0f), tumVAF(ns, 0.0f), norVAF(ns, 0.0f), mixedRatio(ns, 0.0f);
            std::vector<int> ivlCount(ns, 0), caseCount(ns, 0), norDepth(ns, 0), sameCount(ns, 0);
            std::vector<uint8_t> highCon(ns, 0), filt(ns, 0);
            std::vector<int> hp3(h3.begin(), h3.end());
            for (size_t i = 0; i < ns; ++i) {                                // getDenseTumorSnpInterval, first loop: mean HP3 count of the reads that carry the ALT here
                if (pairs[i].empty()) continue;
                float readCount = 0.0f, altMean = 0.0f;
                for (auto &pr : pairs[i]) { if (pr.second != 3) continue;
                    readCount++;
                    if (!has[(size_t)pr.first]) die("[ERROR](getDenseTumorSnpInterval) => readID not found in readHpResultSet");
                    altMean += (float)hp3[(size_t)pr.first];
                    }
                if (altMean != 0) altMean /= readCount;
                meanAlt[i] = altMean;
            }
            {   // intervals of sites at most 5000 bp apart (INTERVAL_SNP_MAX_DISTANCE), z-score of meanAlt inside each
                struct Ivl { std::map<int, double> mean, z; int count = 0; };
                std::vector<Ivl> ivls; Ivl cur; bool rec = false; int startPos = 0;
                auto close_ivl = [&]() { const double sz = (double)cur.mean.size();
                    double sum = 0;
                    for (auto &m : cur.mean) sum += m.second;
                    const double mean = sz == 0 ? 0.0 : sum / sz;
                    double var = 0;
                    for (auto &m : cur.mean) var += (m.second - mean) * (m.second - mean);
                    const double sd = std::sqrt(var / cur.mean.size());
                    for (auto &m : cur.mean) cur.z[m.first] = sd == 0 ? 0.0 : (m.second - mean) / sd; ivls.push_back(cur); };
                for (size_t i = 0; i < ns; ++i) {
                    if (i + 1 < ns) {
                        const int curPos = pos[(size_t)sites[i]], nextPos = pos[(size_t)sites[i + 1]], dist = nextPos - curPos;
                        if (dist <= 5000) { if (!rec) { rec = true;
                                startPos = curPos;
                                cur.mean[(int)i] = meanAlt[i];
                                cur.count++;
                                } cur.mean[(int)(i + 1)] = meanAlt[i + 1];
                            cur.count++;
                            }
                        else if (rec) { close_ivl(); rec = false; startPos = 0; cur = Ivl(); }
                    }
                }
                if (rec && ns && pos[(size_t)sites[ns - 1]] - startPos <= 5000) close_ivl();
                for (const Ivl &iv : ivls) if (iv.count > 1) for (auto &z : iv.z) { zScore[(size_t)z.first] = (float)std::abs(z.second);
                    ivlCount[(size_t)z.first] = iv.count;
                    }
            }
            // offset (-100..100) -> count per allele (PosSomaticOffsetBase; std::map<int,int> per site in the reference: flat counters here - the
            // DenseAlt test below only counts offsets that qualify, up to three, so the order they are visited in does not matter)
            const int WOFF = 100, WN = 2 * WOFF + 1;
            std::vector<int32_t> winRef(ns * (size_t)WN, 0), winAlt(ns * (size_t)WN, 0);
            for (int64_t k = 0; k < te.n_windows; ++k) { const int sidx = site_of[(size_t)wn_site[(size_t)k]];
                if (sidx < 0) continue;
                const int off = wn_off[(size_t)k];
                if (off < -WOFF || off > WOFF) die("[ERROR] difference window offset out of range");
                (wn_al[(size_t)k] ? winAlt : winRef)[(size_t)sidx * WN + (size_t)(off + WOFF)]++;
                }
            for (size_t i = 0; i < ns; ++i) {                                // somaticFeatureFilter
                const size_t v = (size_t)sites[i];
                const int32_t *c = &tsite[v * LPS_TSITE_COUNTERS], *n = &nsite[v * LPS_SITE_COUNTERS];
                const int kind = tkind[v];
                if (kind == 4) continue;                                     // MNP rows never become high-confidence calls
                auto base_count = [&](const int32_t *cc, uint8_t b) { return b == 'A' ? cc[1] : b == 'C' ? cc[2] : b == 'G' ? cc[3] : b == 'T' ? cc[4] : 0;
                    };
                auto vaf = [](int alt, int depth) { return (depth == 0 || alt == 0) ? 0.0f : (float)alt / (float)depth; };
                const int tAlt = kind == 1 ? base_count(c, a0[v]) : c[0], nAlt = kind == 1 ? base_count(n, a0[v]) : n[0];
                tumVAF[i] = vaf(tAlt, c[6]); norVAF[i] = vaf(nAlt, n[6]); norDepth[i] = n[6];
                const int clean = c[25], messy = c[29]; caseCount[i] = clean + messy;
                mixedRatio[i] = caseCount[i] != 0 ? (float)messy / ((float)clean + (float)messy) : 0.0f;
                const bool f_tinc = !(norVAF[i] <= T.norVAF_max && (float)norDepth[i] > (float)T.norDepth_min);
                const bool f_messy = mixedRatio[i] >= T.messy, f_count = caseCount[i] <= T.readCount_min;
                bool f_hap = false;
                if (caseCount[i] <= T.hap_readCount_max && tumVAF[i] <= T.hap_VAF_max) { if (c[35] > T.hap_somaticRead_min && c[37] > T.hap_somaticRead_min) f_hap = true;
                    }
                bool f_z = false;
                if (caseCount[i] <= T.ivl_readCount_max && tumVAF[i] <= T.ivl_VAF_max) { if (ivlCount[i] > T.ivl_count_min && zScore[i] <= T.z_max && zScore[i] >= 0.0f) f_z = true;
                    }
                int same = 0;
                { const float c1 = 0.5f, c2 = 0.6f;
                    const int target = c[0];
                    // DenseAlt: base.altCount
                    for (int w = 0; w < WN; ++w) { const int aa = winAlt[i * WN + (size_t)w]; if (!aa) continue;
                        const int ra = winRef[i * WN + (size_t)w];
                        const double k1 = (double)aa / target, k2 = (double)aa / (ra + aa); if (k1 >= c1 && k2 >= c2) { if (++same == 3) break; } } }
                sameCount[i] = same; const bool f_dense = same >= 3;
                filt[i] = f_tinc || f_messy || f_count || f_hap || f_z || f_dense;
                if (write_log) flog << chr << "\t" << pos[v] + 1 << "\t" << norVAF[i] << "\t" << norDepth[i] << "\t" << mixedRatio[i] << "\t" << caseCount[i] << "\t" << tumVAF[i] << "\t" << ivlCount[i] << "\t"
                                    << zScore[i] << "\t" << sameCount[i] << "\t" << f_tinc << "\t" << f_messy << "\t" << f_count << "\t" << f_hap << "\t" << f_z << "\t" << f_dense << "\t" << (int)filt[i] << "\n";
                if (!(enable_filter && filt[i])) highCon[i] = 1;
            }
            for (size_t i = 0; i < ns; ++i) {                                // calibrateReadHP: reads lose the H3 votes of rejected sites
                if (highCon[i]) continue;
                if (pairs[i].empty()) die("[ERROR](calibrate read HP) => can't find pos in tumorPosReadCorrBaseHP : chr: " + chr + " pos: " + std::to_string(pos[(size_t)sites[i]] + 1));
                for (auto &pr : pairs[i]) if (pr.second == 3) { if (--hp3[(size_t)pr.first] < 0) die("[ERROR](calibrate read HP) => read HP3 or HP4 SNP count < 0 :");
                    }
            }
            std::vector<uint8_t> setHp(nt, 0);                               // calculateReadSetHP
            for (size_t r = 0; r < nt; ++r) if (has[r]) setHp[r] = (uint8_t)judge_somatic_read_hap(h1[r], h2[r], hp3[r], tnps[r], pct);
            for (size_t i = 0; i < ns; ++i) {                                // statisticSomaticPosReadHP + getSomaticFlag
                if (!highCon[i]) continue;
                if (pairs[i].empty()) die("[ERROR](statistic all read HP) => can't find pos in tumorPosReadCorrBaseHP : chr: " + chr + " pos: " + std::to_string(pos[(size_t)sites[i]] + 1));
                int d1 = 0, d2 = 0;
                for (auto &pr : pairs[i]) if (pr.second == 3) { if (setHp[(size_t)pr.first] == 5) ++d1;
                    else if (setHp[(size_t)pr.first] == 7) ++d2;
                    }
                const int tot = d1 + d2;
                float r1 = 0.0f, r2 = 0.0f;
                if (tot > 0) { if (d1 > 0) r1 = (float)d1 / (float)tot;
                    if (d2 > 0) r2 = (float)d2 / (float)tot;
                    }
                const size_t v = (size_t)sites[i]; ++n_somatic_flag; if (write_sc_vcf) A.som.insert(pos[v]);
                if (role[v] != 0) { role[v] = 1;
                    derive[v] = r1 >= 1.0f ? 1 : r2 >= 1.0f ? 2 : 0;
                    }   // a position that also has a normal row keeps its germline role in the tagging pass
            }
            tock(ns_host, t_stage); t_stage = tick();
            // ---- pass 3: tagging (SomaticHaplotagChrProcessor::judgeHaplotype); the tumor reads are still resident
            std::vector<int32_t> g1(nt), g2(nt), g3(nt), dh1(nt), dh2(nt), gmin(nt); std::vector<uint8_t> gnps(nt);
            lps_somatic_tag_result tg{(int64_t)nt, status.data(), g1.data(), g2.data(), g3.data(), dh1.data(), dh2.data(), gnps.data(), gmin.data(), hp.data(), pq.data(), psv.data()};
            if (L.set_variants(ctx, &vt) || L.somatic_tag_chromosome(ctx, &tg)) fail();
            tock(ns_p3, t_stage);
        }
        auto t_splice = tick();
        // ---- writer: HP:Z / PS:i (when the read saw a normal phase set) / PQ:i  (SomaticHaplotagProcess.cpp:529-536), records in input order
        if (resident) {                                                     // records spliced, cut into BGZF blocks and deflated where the inflated stream already is
            for (size_t i = 0; i < nt; ++i) { ++st_count[status[i] & 7]; if (status[i] == 0) ++hp_hist[hp[i] < 9 ? hp[i] : 0]; }
            if (!tumor_pushed) { std::vector<uint32_t> tid(nt, 0);         // a contig without variants: every record is copied untouched
                if (L.begin_chromosome(ctx) || L.push_bam_resident(ctx, t_range.first, t_range.second, tid.data())) fail(); }
            int64_t nb = 0;
            if (L.somatic_write_bgzf(ctx, status.data(), hp.data(), psv.data(), pq.data(), nullptr, 0, &nb)) fail();
            tock(ns_splice, t_splice);
            auto td = tick();
            if (n_workers_now > 1) {                                        // several workers: the merger writes the contigs' members in contig order
                uint8_t *zb = (uint8_t *)malloc((size_t)nb + 64); if (!zb) die("ERROR: out of memory");
                uint8_t *bounce = (uint8_t *)L.host_alloc(32u << 20); if (!bounce) die("longphase_amd: cannot allocate page-locked host memory");
                for (int64_t off = 0; off < nb; off += (32ll << 20)) { const int64_t len = std::min<int64_t>(32ll << 20, nb - off);
                    if (L.bgzf_deflate_fetch_range(ctx, off, len, bounce)) fail();
                    memcpy(zb + off, bounce, (size_t)len); }
                L.host_free(bounce);
                A.out = zb; A.out_bytes = (size_t)nb; A.deflated = true;
                tock(ns_gpu_deflate, td);
                std::cerr << "(" << chr << ")";
                return;
            }
            if (!pin[0]) { pin[0] = (uint8_t *)L.host_alloc(64u << 20); pin[1] = (uint8_t *)L.host_alloc(64u << 20); if (!pin[0] || !pin[1]) die("longphase_amd: cannot allocate page-locked host memory"); }
            if (!raw_started) { w.flush_partial(); raw_started = true; }
            std::thread wr; int k = 0;
            for (int64_t off = 0; off < nb; off += (64ll << 20), k ^= 1) { const int64_t len = std::min<int64_t>(64ll << 20, nb - off);
                if (L.bgzf_deflate_fetch_range(ctx, off, len, pin[k])) fail();
                if (wr.joinable()) wr.join();
                uint8_t *src = pin[k];
                wr = std::thread([&w, src, len] { w.write_raw(src, (size_t)len); }); }
            if (wr.joinable()) wr.join();
            tock(ns_gpu_deflate, td);
            std::cerr << "(" << chr << ")";
            return;
        }
        std::vector<uint64_t> out_off(nt + 1, 0);
        auto aux_of = [&](const uint8_t *r) { const uint32_t l_name = r[8], n_cig = r[12] | (r[13] << 8), l_seq = rd32(r + 16);
            return r + 32 + l_name + 4ull * n_cig + (l_seq + 1) / 2 + l_seq;
            };
        static const char *hp_str[9] = {".", "1", "2", "3", "4", "1-1", "1-2", "2-1", "2-2"};
        auto tag_bytes = [&](size_t i) -> size_t { if (status[i] != 0 || !hp[i]) return 0;
            return 3 + strlen(hp_str[hp[i] < 9 ? hp[i] : 0]) + 1 + (psv[i] != -1 ? 7 : 0) + 7;
            };
        auto parallel_records = [&](const std::function<void(size_t, size_t)> &fn) {      // records are independent: -t threads share them
            const int nth = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads, nt / 256 + 1)); std::vector<std::thread> th;
            for (int t = 0; t < nth; ++t) th.emplace_back([&, t] { fn(nt * t / nth, nt * (t + 1) / nth); });
            for (auto &x : th) x.join();
        };
        std::atomic<int> malformed{0};
        parallel_records([&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            const uint8_t *r = tbase + tc.rec_off[i]; const uint32_t bs = rd32(r - 4); uint64_t len = 4ull + bs;
            if (status[i] == 0) { bool seen[3] = {false, false, false};
                for (const uint8_t *p = aux_of(r), *end = r + bs; p < end;) { const size_t l = aux_field_len(p, end);
                    if (!l) { malformed = 1; break; }
                    const int which = (p[0] == 'H' && p[1] == 'P') ? 0 : (p[0] == 'P' && p[1] == 'S') ? 1 : (p[0] == 'P' && p[1] == 'Q') ? 2 : -1;
                    if (which >= 0 && !seen[which]) { seen[which] = true;
                        len -= l;
                        } p += l;
                    }
                len += tag_bytes(i); }
            out_off[i + 1] = len;
        } });
        if (malformed) die("ERROR: malformed auxiliary field in " + tbam);
        for (size_t i = 0; i < nt; ++i) { out_off[i + 1] += out_off[i]; ++st_count[status[i] & 7]; if (status[i] == 0) ++hp_hist[hp[i] < 9 ? hp[i] : 0]; }
        uint8_t *obp = (uint8_t *)malloc(out_off[nt] + 64); if (!obp) die("ERROR: out of memory");      // (not a vector: 2 GB would be zeroed first)
        parallel_records([&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            const uint8_t *r = tbase + tc.rec_off[i]; const uint32_t bs = rd32(r - 4); uint8_t *o = obp + out_off[i];
            if (status[i] != 0) { memcpy(o, r - 4, 4 + (size_t)bs); continue; }
            const uint8_t *aux = aux_of(r), *end = r + bs;
            uint8_t *q = o + 4;
            memcpy(q, r, (size_t)(aux - r));
            q += aux - r;
            bool seen[3] = {false, false, false};
            for (const uint8_t *p = aux; p < end;) { const size_t l = aux_field_len(p, end);
                const int which = (p[0] == 'H' && p[1] == 'P') ? 0 : (p[0] == 'P' && p[1] == 'S') ? 1 : (p[0] == 'P' && p[1] == 'Q') ? 2 : -1;
                if (which >= 0 && !seen[which]) seen[which] = true; else { memcpy(q, p, l); q += l; } p += l; }
            if (hp[i]) { const char *hs = hp_str[hp[i] < 9 ? hp[i] : 0];
                *q++ = 'H';
                *q++ = 'P';
                *q++ = 'Z';
                const size_t hl = strlen(hs) + 1;
                memcpy(q, hs, hl);
                q += hl;
                auto put_i = [&](char a, char b, int32_t v) { *q++ = (uint8_t)a;
                    *q++ = (uint8_t)b;
                    *q++ = 'i';
                    for (int k = 0; k < 4; ++k) *q++ = (uint8_t)((uint32_t)v >> (8 * k));
                    };
                if (psv[i] != -1) put_i('P', 'S', psv[i]); put_i('P', 'Q', pq[i]); }
            const uint32_t nbs = (uint32_t)(q - o) - 4; for (int k = 0; k < 4; ++k) o[k] = (uint8_t)(nbs >> (8 * k));
        } });
        tock(ns_splice, t_splice);
        if (!host_deflate && out_off[nt]) {                                // BGZF blocks cut and deflated on this worker's GPU (per-block Huffman codes, as the haplotag writer)
            auto td = tick();
            int64_t nb = 0;
            if (L.bgzf_deflate_host(ctx, obp, (int64_t)out_off[nt], &nb)) fail();
            free(obp);
            if (n_workers_now == 1) {                                       // this thread is also the writer: pieces go from two page-locked buffers straight to the file
                if (!pin[0]) { pin[0] = (uint8_t *)L.host_alloc(64u << 20); pin[1] = (uint8_t *)L.host_alloc(64u << 20); if (!pin[0] || !pin[1]) die("longphase_amd: cannot allocate page-locked host memory"); }
                if (!raw_started) { w.flush_partial(); raw_started = true; }
                std::thread wr; int k = 0;
                for (int64_t off = 0; off < nb; off += (64ll << 20), k ^= 1) { const int64_t len = std::min<int64_t>(64ll << 20, nb - off);
                    if (L.bgzf_deflate_fetch_range(ctx, off, len, pin[k])) fail();
                    if (wr.joinable()) wr.join();
                    uint8_t *src = pin[k];
                    wr = std::thread([&w, src, len] { w.write_raw(src, (size_t)len); }); }
                if (wr.joinable()) wr.join();
                tock(ns_gpu_deflate, td);
                std::cerr << "(" << chr << ")";
                return;
            }
            uint8_t *zb = (uint8_t *)malloc((size_t)nb + 64); if (!zb) die("ERROR: out of memory");
            uint8_t *bounce = (uint8_t *)L.host_alloc(32u << 20); if (!bounce) die("longphase_amd: cannot allocate page-locked host memory");
            for (int64_t off = 0; off < nb; off += (32ll << 20)) { const int64_t len = std::min<int64_t>(32ll << 20, nb - off);
                if (L.bgzf_deflate_fetch_range(ctx, off, len, bounce)) fail();
                memcpy(zb + off, bounce, (size_t)len); }
            L.host_free(bounce);
            A.out = zb; A.out_bytes = (size_t)nb; A.deflated = true;
            tock(ns_gpu_deflate, td);
        } else { A.out = obp; A.out_bytes = (size_t)out_off[nt]; }
        std::cerr << "(" << chr << ")";
    };
    // GROUPED: which group a contig belongs to, and the loader that makes a group's records resident (tumor in ctx, normal in nctx) when the
    // contig loop reaches its first member
    std::map<std::string, size_t> group_of;
    for (size_t g = 0; g < som_groups.size(); ++g) for (const std::string &c : som_groups[g]) group_of[c] = g;
    auto enter_group = [&](SomWorker &W, const std::string &chr, int phase) {
        if (!grouped) return;
        GpuBam &tgb = *W.tgb, &ngb = *W.ngb;
        auto it = group_of.find(chr);
        if (it == group_of.end()) { tgb.range.erase(chr); ngb.range.erase(chr); return; }      // no tumor records
        if (it->second == W.group_loaded) return;
        const std::vector<std::string> &grp = som_groups[it->second];
        tgb.load_group(L, W.ctx, grp);
        std::vector<std::string> ngrp; for (const std::string &c : grp) { const int t = ngb.tid_of(c); if (t >= 0 && ngb.voff[(size_t)t].second > ngb.voff[(size_t)t].first) ngrp.push_back(c); }
        if (phase == 0 || !estimate) ngb.load_group(L, W.nctx, ngrp);      // (the calling phase behind an estimation phase reuses that phase's passes: it needs the tumor records only)
        else ngb.range.clear();
        W.group_loaded = it->second;
        // one worker: the next group's header walk (host only) beside this group's passes; at the last group of the estimation phase the first group's again
        if (n_workers_now == 1 && som_groups.size() > 1) tgb.walk_group_ahead(L, som_groups[it->second + 1 < som_groups.size() ? it->second + 1 : 0]);
        if (getenv("LPS_CLI_DEBUG")) fprintf(stderr, "[cli] somatic group %zu / %zu: %zu contig(s) from %s, %zu of them in the normal BAM\n", it->second + 1, som_groups.size(), grp.size(), grp.front().c_str(), ngrp.size());
    };
    // contigs dealt longest-first (tumor records) onto the workers; worker 0 is this thread's context, the others create theirs
    const int n_dev = std::max(1, L.device_count());
    const int n_workers = std::max(1, std::min<int>(n_gpus, grouped ? (int)std::max<size_t>(1, som_groups.size()) : (int)chr_vec.size()));
    std::vector<std::vector<size_t>> share((size_t)n_workers);
    if (grouped && n_workers > 1) {
        // whole GROUPS dealt longest-first (compressed bytes of the tumor BAM) onto the workers: a worker uploads and inflates only its own groups
        std::vector<size_t> gorder(som_groups.size()), gbytes(som_groups.size(), 0), load((size_t)n_workers, 0);
        for (size_t g = 0; g < som_groups.size(); ++g) { gorder[g] = g; for (const std::string &c : som_groups[g]) { const int t = tgb.tid_of(c); gbytes[g] += (size_t)((tgb.voff[(size_t)t].second >> 16) - (tgb.voff[(size_t)t].first >> 16)) + 1; } }
        std::stable_sort(gorder.begin(), gorder.end(), [&](size_t a, size_t b) { return gbytes[a] > gbytes[b]; });
        std::vector<int> worker_of_group(som_groups.size(), 0);
        for (size_t g : gorder) { const size_t k = (size_t)(std::min_element(load.begin(), load.end()) - load.begin()); worker_of_group[g] = (int)k; load[k] += gbytes[g]; }
        for (size_t i = 0; i < chr_vec.size(); ++i) { auto it = group_of.find(chr_vec[i]); share[it == group_of.end() ? 0 : (size_t)worker_of_group[it->second]].push_back(i); }
    } else {
      std::vector<size_t> order(chr_vec.size()); for (size_t i = 0; i < order.size(); ++i) order[i] = i;
      auto weight = [&](size_t i) -> size_t { if (resident) { auto a = tgb.range.find(chr_vec[i]); return a == tgb.range.end() ? 0 : (size_t)a->second.second; }
          auto it = tin.contigs.find(chr_vec[i]); return it == tin.contigs.end() ? 0 : it->second.rec_off.size(); };
      std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return weight(a) > weight(b); });
      std::vector<size_t> load((size_t)n_workers, 0);
      for (size_t i : order) { const size_t g = (size_t)(std::min_element(load.begin(), load.end()) - load.begin()); share[g].push_back(i); load[g] += weight(i) + 1; }
      for (auto &v : share) std::sort(v.begin(), v.end()); }
    n_workers_now = n_workers;
    std::vector<SomWorker> wk((size_t)n_workers); std::vector<std::unique_ptr<GpuBam>> wbams;
    wk[0].ctx = ctx; wk[0].nctx = nctx; wk[0].tgb = &tgb; wk[0].ngb = &ngb;
    for (int g = 1; g < n_workers; ++g) { lps_params P; L.default_params(&P); for (auto &f : over) f(P);
        SomWorker &W = wk[(size_t)g];
        W.ctx = L.create((gpu + g) % n_dev, &P); if (!W.ctx) die("longphase_amd: cannot create a GPU context for worker " + std::to_string(g));
        L.set_stage_timing(W.ctx, 0);
        W.tgb = &tgb; W.ngb = &ngb;                                       // (host-record path: the files' records are in host memory, shared)
        if (resident) {                                                   // grouped: its own second context and its own views of the two files
            W.nctx = L.create((gpu + g) % n_dev, &P); if (!W.nctx) die("longphase_amd: cannot create the normal-BAM context of worker " + std::to_string(g));
            L.set_stage_timing(W.nctx, 0);
            wbams.emplace_back(new GpuBam()); wbams.back()->open_file(tbam, true); W.tgb = wbams.back().get();
            wbams.emplace_back(new GpuBam()); wbams.back()->open_file(nbam, true); W.ngb = wbams.back().get(); } }
    for (int phase = estimate ? 0 : 1; phase < 2; ++phase) {
        if (phase == 1 && estimate) { auto tp = tick(); purity = estimate_purity(pdata, p_initial, lcvf, prefix); T = somatic_thresholds(purity); announce(); tock(ns_purity, tp); }
        std::vector<ContigAcc> acc(chr_vec.size()); std::mutex mu; std::condition_variable cv;
        auto run_share = [&](int g) { for (size_t i : share[(size_t)g]) { enter_group(wk[(size_t)g], chr_vec[i], phase); do_contig(wk[(size_t)g], chr_vec[i], phase,
                acc[i], saved[i]); { std::lock_guard<std::mutex> lk(mu); acc[i].ready = true; } cv.notify_all(); } };
        std::vector<std::thread> workers;
        if (n_workers > 1) for (int g = 0; g < n_workers; ++g) workers.emplace_back(run_share, g);
        for (size_t i = 0; i < chr_vec.size(); ++i) {                    // merge (and write) in contig order
            if (n_workers == 1) { enter_group(wk[0], chr_vec[i], phase); do_contig(wk[0], chr_vec[i], phase, acc[i], saved[i]); }
            else { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return acc[i].ready; }); }
            ContigAcc &A = acc[i];
            pdata.insert(pdata.end(), A.pdata.begin(), A.pdata.end()); p_initial += A.p_initial; for (int k = 0; k < 5; ++k) lcvf[k] += A.lcvf[k];
            if (write_log) flog << A.flog.str();
            if (!A.som.empty()) somatic_pos[chr_vec[i]].swap(A.som);
            n_somatic_flag += A.n_flag; for (int k = 0; k < 9; ++k) hp_hist[k] += A.hp_hist[k]; for (int k = 0; k < 8; ++k) st_count[k] += A.st_count[k];
            { auto ta = tick(); if (A.out_bytes) { if (A.deflated) { if (!raw_started) { w.flush_partial(); raw_started = true; } w.write_raw(A.out,
                    A.out_bytes); } else w.append(A.out, A.out_bytes); } tock(ns_append, ta); }
            free(A.out); A.out = nullptr; A.flog.str(std::string());
        }
        for (auto &x : workers) x.join();
    }
    for (int g = 1; g < n_workers; ++g) { L.destroy(wk[(size_t)g].ctx); if (resident && wk[(size_t)g].nctx) L.destroy(wk[(size_t)g].nctx); }
    if (grouped) for (int g = 1; g < n_workers; ++g) { tgb.t_inflate += wk[(size_t)g].tgb->t_inflate; tgb.t_scan += wk[(size_t)g].tgb->t_scan; ngb.t_inflate += wk[(size_t)g].ngb->t_inflate; ngb.t_scan += wk[(size_t)g].ngb->t_scan; }
    if (n_workers > 1) std::cerr << "\n" << n_workers << " workers (one GPU context each, contigs dealt by tumor record count)";
    std::cerr << "\n";
    { auto tf = tick(); w.finish(); tock(ns_finish, tf); }
    if (write_log) flog.close();                                       // the process leaves through _exit: nothing is flushed implicitly
    if (write_sc_vcf) {                                                // VcfParser::writeProcess (src/haplotag/HaplotagVcfParser.cpp:548-614): the tumor VCF with FILTER rewritten
        std::ofstream o(prefix + "_sc.vcf"); if (!o) die("Fail to open output file: " + prefix + "_sc.vcf");
        bool cmd_done = false; std::set<std::string> in_vec(chr_vec.begin(), chr_vec.end());
        for (const std::string &in : tlines) {
            if (in.size() >= 2 && in.compare(0, 2, "##") == 0) { o << in << std::endl; continue; }
            if (in.size() >= 6 && (in.compare(0, 6, "#CHROM") == 0 || in.compare(0, 6, "#chrom") == 0)) { if (!cmd_done) { o << "##longphase_s_version=" << kVersion << std::endl << "##commandline=" << command << std::endl;
                    cmd_done = true;
                    } o << in << std::endl;
                continue;
                }
            std::istringstream iss(in); std::vector<std::string> f((std::istream_iterator<std::string>(iss)), std::istream_iterator<std::string>());
            if (f.empty()) continue;
            if (f.size() < 7) die("[ERROR](VcfParser::writeProcess) => VCF file format error: " + in);
            if (!in_vec.count(f[0])) continue;
            const int32_t p0 = std::stoi(f[1]) - 1;
            auto ct = trows.find(f[0]);
            if (ct == trows.end()) continue;
            auto rt = ct->second.find(p0);
            if (rt == ct->second.end() || rt->second.kind == 4) continue;
            const bool som = somatic_pos.count(f[0]) && somatic_pos[f[0]].count(p0);
            if (som) { if (f[6] != "PASS") f[6] = "PASS"; } else if (f[6] == "PASS") f[6] = "LowQual";
            std::string line = f[0]; for (size_t i = 1; i < f.size(); ++i) line += "\t" + f[i];
            o << line << std::endl;
        }
    }
    if (nctx) L.destroy(nctx);
    tgb.close_file(); ngb.close_file();
    L.destroy(ctx);
    unsigned long long total = 0; for (int k = 0; k < 8; ++k) total += st_count[k];
    fprintf(stderr, "somatic variant count(Flag): %llu\n", n_somatic_flag);
    fprintf(stderr, "total alignment %llu | HP1 %llu HP2 %llu HP1-1 %llu HP2-1 %llu HP3 %llu | judged untagged %llu | low mapq %llu unmapped %llu secondary %llu supplementary %llu no variant %llu beyond last variant %llu\n",
            total, hp_hist[1], hp_hist[2], hp_hist[5], hp_hist[7], hp_hist[3], hp_hist[0], st_count[1], st_count[2], st_count[3], st_count[4], st_count[5], st_count[6]);
    if (grouped) fprintf(stderr, "contig groups: %zu (both BAMs indexed, --group-bytes %llu): upload + gpu inflate %.3fs, record scan %.3fs, inside the passes' time\n", som_groups.size(), (unsigned long long)group_bytes,
                         tgb.t_inflate + ngb.t_inflate, tgb.t_scan + ngb.t_scan);
    fprintf(stderr, "inputs %.3fs | passes + caller + writer %.3fs (table %.3f, normal pass %.3f, tumor pass %.3f, host stages %.3f, purity %.3f, tagging pass %.3f, tag splice %.3f, gpu deflate + copy out %.3f, %s %.3f + %.3f) | total %.3fs\n", t_in - t_begin, now() - t_in,
            ns_prep / 1e9, ns_p1 / 1e9, ns_p2 / 1e9, ns_host / 1e9, ns_purity / 1e9, ns_p3 / 1e9, ns_splice / 1e9, ns_gpu_deflate / 1e9, host_deflate ? "deflate + write" : "write", ns_append / 1e9, ns_finish / 1e9, now() - t_begin);
    fflush(stderr);
    if (getenv("LPS_CLI_NO_FAST_EXIT")) return 0;                       // e.g. under a profiler that writes its report from an exit handler
    _exit(0);
}

// `longphase_amd view BAM CONTIG` — decoded records as SAM columns 1-11 (CPU-only check of the BGZF reader and the record walk)
static int view_main(int argc, char **argv) {
    if (argc < 4) die("Usage: longphase_amd view <in.bam> <contig> [threads]");
    std::map<std::string, int> want{{argv[3], 1}}; BamFile f;
    f.load(argv[2], argc > 4 ? atoi(argv[4]) : 1, want);
    const ContigRecords &k = f.contigs[argv[3]];
    std::string line;
    for (size_t i = 0; i < k.rec_off.size(); ++i) {
        const uint8_t *r = f.z.data + k.lo + k.rec_off[i];
        const uint32_t l_name = r[8], n_cig = r[12] | (r[13] << 8), l_seq = rd32(r + 16);
        size_t nl; const char *nm = f.name_of(k, i, nl);
        line = std::string(nm, nl) + "\t" + std::to_string(r[14] | (r[15] << 8)) + "\t" + argv[3] + "\t" + std::to_string((int32_t)rd32(r + 4) + 1) + "\t" + std::to_string(r[9]) + "\t";
        const uint8_t *cg = r + 32 + l_name, *sq = cg + 4ull * n_cig, *ql = sq + (l_seq + 1) / 2;
        for (uint32_t c = 0; c < n_cig; ++c) { const uint32_t w = rd32(cg + 4ull * c); line += std::to_string(w >> 4) + "MIDNSHP=XB"[w & 15]; }
        if (!n_cig) line += "*";
        line += "\t*\t0\t0\t";
        for (uint32_t j = 0; j < l_seq; ++j) line += "=ACMGRSVTWYHKDBN"[(sq[j >> 1] >> ((j & 1) ? 0 : 4)) & 15];
        line += "\t";
        for (uint32_t j = 0; j < l_seq; ++j) line += (char)(ql[j] + 33);
        std::cout << line << "\n";
    }
    return 0;
}

// the SNP table `phase` makes of a VCF, as text (no GPU; tests/test_cli_cpu.py holds the threaded parser against a line-by-line restatement):
//   longphase_amd vcf-table <in.vcf[.gz]> <threads> [--indels] [--indelQuality=N]
static int vcf_table_main(int argc, char **argv) {
    if (argc < 4) die("Usage: longphase_amd vcf-table <in.vcf> <threads> [--indels] [--indelQuality=N]");
    bool indels = false; IndelQual iq;
    for (int i = 4; i < argc; ++i) { const std::string a = argv[i]; if (a == "--indels") indels = true; else if (a.rfind("--indelQuality=", 0) == 0) iq.threshold = atoi(a.c_str() + 15); }
    std::vector<std::string> lines; if (!read_lines(argv[2], lines)) die(std::string("ERROR: Cannot open vcf file ") + argv[2]);
    std::vector<std::string> order; std::map<std::string, ChrVariants> vars;
    parse_vcf(lines, indels, order, vars, &iq, atoi(argv[3]));
    for (const std::string &c : order) { const ChrVariants &v = vars[c]; std::cout << "#" << c << "\t" << v.pos.size() << "\n";
        for (size_t i = 0; i < v.pos.size(); ++i) std::cout << c << "\t" << v.pos[i] << "\t" << v.ref[i] << "\t" << v.alt[i] << "\n"; }
    for (auto &kv : iq.filtered) for (int32_t p : kv.second) std::cout << "!" << kv.first << "\t" << p << "\n";
    return 0;
}

int main(int argc, char **argv) {
    g_main_entered = epoch_now();
    std::string command; for (int i = 0; i < argc; ++i) { if (i) command += " "; command += argv[i]; }
    if (argc < 2) { std::cout << "Version: " << kVersion << "\nUsage: longphase_amd <command> [options]\n    phase    run phasing algorithm on the GPU.\n    haplotag tag reads by haplotype on the GPU.\n    somatic_haplotag tag tumor reads (somatic + germline haplotypes) on the GPU; needs --tumor-purity.\n";
        return 0;
        }
    const std::string cmd = argv[1];
    if (cmd == "phase") return phase_main(argc, argv, command);
    if (cmd == "view") return view_main(argc, argv);
    if (cmd == "vcf-table") return vcf_table_main(argc, argv);
    if (cmd == "haplotag") return haplotag_main(argc, argv, command);
    if (cmd == "somatic_haplotag") return somatic_main(argc, argv, command);
    std::cerr << "Unrecognized command: " << cmd << "\n"; return 1;
}
