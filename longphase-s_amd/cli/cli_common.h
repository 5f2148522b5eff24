// cli_common.h — shared includes, error exit and the run-time loader of liblps_hip.so for longphase_amd (split out of longphase_amd.cpp in round 2).
#pragma once
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <iterator>
#include <map>
#include <memory>
#include <set>
#include <mutex>
#include <numeric>
#include <sstream>
#include <condition_variable>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/lps_abi.h"

static const char *kVersion = "1.0.0-mi355x";

// Errors end the process at once, from whatever thread: _exit skips the static destructors and the ROCm runtime's teardown, which would otherwise race
// with the streams of worker threads that are still running (--gpus N) - exit(1) from a worker could hang or crash on the way out.
[[noreturn]] static void die(const std::string &m) { std::cerr << m << "\n"; std::cerr.flush(); fflush(nullptr); _exit(1); }
static const size_t kGpuInflateMinBytes = 256u << 20;
static double epoch_now() { struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
static size_t file_bytes(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0 ? (size_t)st.st_size : 0; }

// ------------------------------------------------------------------------------------------------ the library, loaded at run time
// liblps_hip.so (and with it the ROCm runtime) is dlopen'ed from a helper thread so that loading it and creating the GPU context
// overlap with reading the inputs; the CLI binary itself has no GPU dependency (its `view` subcommand runs anywhere).
struct Lps {
    void *so = nullptr;
    decltype(&lps_default_params) default_params = nullptr; decltype(&lps_create) create = nullptr; decltype(&lps_destroy) destroy = nullptr;
    decltype(&lps_last_error) last_error = nullptr;
    decltype(&lps_begin_chromosome) begin_chromosome = nullptr;
    decltype(&lps_set_variants) set_variants = nullptr;
    decltype(&lps_set_reference) set_reference = nullptr;
    decltype(&lps_set_extra_variants) set_extra_variants = nullptr; decltype(&lps_get_extra_result) get_extra_result = nullptr; decltype(&lps_set_read_votes) set_read_votes = nullptr;
    decltype(&lps_bgzf_deflate_fetch_range) bgzf_deflate_fetch_range = nullptr; decltype(&lps_bgzf_deflate_host) bgzf_deflate_host = nullptr; decltype(&lps_host_alloc) host_alloc = nullptr; decltype(&lps_host_free) host_free = nullptr;
    decltype(&lps_push_bam_records) push_bam_records = nullptr;
    decltype(&lps_phase_chromosome) phase_chromosome = nullptr;
    decltype(&lps_haplotag_chromosome) haplotag_chromosome = nullptr; decltype(&lps_abi_version) abi_version = nullptr;
    decltype(&lps_bgzf_load) bgzf_load = nullptr; decltype(&lps_bgzf_read) bgzf_read = nullptr; decltype(&lps_bam_scan) bam_scan = nullptr;
    decltype(&lps_bam_record_tids) bam_record_tids = nullptr;
    decltype(&lps_bam_names) bam_names = nullptr;
    decltype(&lps_push_bam_resident) push_bam_resident = nullptr;
    decltype(&lps_bam_record_offsets) bam_record_offsets = nullptr;
    decltype(&lps_bam_scan_range) bam_scan_range = nullptr;
    decltype(&lps_device_count) device_count = nullptr;
    decltype(&lps_set_stage_timing) set_stage_timing = nullptr;
    decltype(&lps_haplotag_write_bgzf) haplotag_write_bgzf = nullptr; decltype(&lps_bgzf_deflate_fetch) bgzf_deflate_fetch = nullptr;
    decltype(&lps_somatic_extract_normal) somatic_extract_normal = nullptr; decltype(&lps_somatic_extract_tumor) somatic_extract_tumor = nullptr;
    decltype(&lps_somatic_tag_chromosome) somatic_tag_chromosome = nullptr;
    decltype(&lps_comm_create_all) comm_create_all = nullptr;
    decltype(&lps_comm_bcast) comm_bcast = nullptr; decltype(&lps_comm_bcast_to_device) comm_bcast_to_device = nullptr; decltype(&lps_set_variants_device) set_variants_device = nullptr;
    decltype(&lps_somatic_write_bgzf) somatic_write_bgzf = nullptr; decltype(&lps_bgzf_load_fd) bgzf_load_fd = nullptr; decltype(&lps_bgzf_walk_fd) bgzf_walk_fd = nullptr; decltype(&lps_bgzf_blocks_free) bgzf_blocks_free = nullptr; decltype(&lps_bgzf_load_fd_blocks) bgzf_load_fd_blocks = nullptr; decltype(&lps_dump_graph) dump_graph = nullptr; decltype(&lps_dump_votes) dump_votes = nullptr;
    decltype(&lps_comm_destroy) comm_destroy = nullptr;
    decltype(&lps_comm_size) comm_size = nullptr; decltype(&lps_comm_last_error) comm_last_error = nullptr;
    std::atomic<bool> ready{false};                                     // the symbols are bound (load() has returned true): host-only entry points may be called
    std::string error;
    bool load() {
        char exe[4096]; const ssize_t k = readlink("/proc/self/exe", exe, sizeof exe - 1);
        std::string dir = "."; if (k > 0) { exe[k] = 0; dir = exe; dir = dir.substr(0, dir.find_last_of('/')); }
        const char *env = getenv("LPS_HIP_LIBRARY");
        const std::string path = env ? env : dir + "/../csrc/liblps_hip.so";
        so = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!so) { error = std::string("cannot load ") + path + ": " + dlerror() + " (the GPU library is required; there is no CPU fallback)";
            return false;
            }
#define LPS_SYM(field, name) field = (decltype(field))dlsym(so, #name); if (!field) { error = "liblps_hip.so does not export " #name; return false; }
        LPS_SYM(default_params, lps_default_params) LPS_SYM(create, lps_create) LPS_SYM(destroy, lps_destroy) LPS_SYM(last_error, lps_last_error)
        LPS_SYM(begin_chromosome, lps_begin_chromosome) LPS_SYM(set_variants, lps_set_variants) LPS_SYM(set_reference, lps_set_reference) LPS_SYM(set_extra_variants,
                lps_set_extra_variants) LPS_SYM(get_extra_result, lps_get_extra_result) LPS_SYM(set_read_votes, lps_set_read_votes) LPS_SYM(bgzf_deflate_fetch_range,
                lps_bgzf_deflate_fetch_range) LPS_SYM(bgzf_deflate_host, lps_bgzf_deflate_host) LPS_SYM(host_alloc, lps_host_alloc) LPS_SYM(host_free, lps_host_free)
        LPS_SYM(push_bam_records, lps_push_bam_records) LPS_SYM(phase_chromosome, lps_phase_chromosome) LPS_SYM(haplotag_chromosome, lps_haplotag_chromosome)
        LPS_SYM(abi_version, lps_abi_version) LPS_SYM(bgzf_load, lps_bgzf_load) LPS_SYM(bgzf_read, lps_bgzf_read) LPS_SYM(bam_scan, lps_bam_scan)
        LPS_SYM(bam_record_tids, lps_bam_record_tids) LPS_SYM(bam_names, lps_bam_names) LPS_SYM(push_bam_resident, lps_push_bam_resident) LPS_SYM(bam_record_offsets,
                lps_bam_record_offsets) LPS_SYM(bam_scan_range, lps_bam_scan_range) LPS_SYM(device_count, lps_device_count) LPS_SYM(set_stage_timing,
                lps_set_stage_timing) LPS_SYM(haplotag_write_bgzf, lps_haplotag_write_bgzf) LPS_SYM(bgzf_deflate_fetch, lps_bgzf_deflate_fetch)
        LPS_SYM(somatic_extract_normal, lps_somatic_extract_normal) LPS_SYM(somatic_extract_tumor, lps_somatic_extract_tumor) LPS_SYM(somatic_tag_chromosome, lps_somatic_tag_chromosome)
        LPS_SYM(somatic_write_bgzf, lps_somatic_write_bgzf) LPS_SYM(bgzf_load_fd, lps_bgzf_load_fd) LPS_SYM(bgzf_walk_fd, lps_bgzf_walk_fd) LPS_SYM(bgzf_blocks_free, lps_bgzf_blocks_free) LPS_SYM(bgzf_load_fd_blocks, lps_bgzf_load_fd_blocks) LPS_SYM(dump_graph, lps_dump_graph) LPS_SYM(dump_votes, lps_dump_votes)
        LPS_SYM(comm_create_all, lps_comm_create_all) LPS_SYM(comm_bcast, lps_comm_bcast) LPS_SYM(comm_bcast_to_device,
                lps_comm_bcast_to_device) LPS_SYM(set_variants_device, lps_set_variants_device) LPS_SYM(comm_destroy, lps_comm_destroy) LPS_SYM(comm_size,
                lps_comm_size) LPS_SYM(comm_last_error, lps_comm_last_error)
#undef LPS_SYM
        if (abi_version() != LPS_ABI_VERSION) { error = "liblps_hip.so has a different ABI version than this binary was built for"; return false; }
        ready.store(true, std::memory_order_release);
        return true;
    }
};
