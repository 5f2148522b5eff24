// lps_graph.h — launch wrappers of lps_graph.hip
#pragma once
#include "lps_kernels.h"

struct GraphTemp { static size_t need(size_t n_sort); };

void sort_keys64_range(void *temp, size_t temp_bytes, const unsigned long long *in, unsigned long long *out, size_t n, int begin_bit, int end_bit, hipStream_t s);
void sort_keys64(void *temp, size_t temp_bytes, const unsigned long long *in, unsigned long long *out, size_t n, int bits, hipStream_t s);
void sort_pairs64(void *temp, size_t temp_bytes, const unsigned long long *kin, unsigned long long *kout, const uint32_t *vin, uint32_t *vout, size_t n, int bits, hipStream_t s);
void exscan_u32(void *temp, size_t temp_bytes, const uint32_t *in, uint32_t *out, size_t n, hipStream_t s);

// device arrays of the stages after the extraction (one chromosome resident)
struct GraphView {
    int n_reads, n_var, A;
    const RowDesc *rows; ObsRec *obs; uint8_t *deleted; const int32_t *vpos;
    const uint32_t *name;                         // dense read-name id of every alignment (order-preserving)
    uint32_t *name_head, *name_link;              // alignments of a name as a list: head by name (+1), link by alignment (+1; 0 = end)
    uint32_t *mm_r, *stack, *mg_start, *mg_cnt, *mg_name, *mg_plan;   // names with several alignments: members in BAM order, per-group range / name / merge plan
    uint32_t *var_cnt, *var_del, *var_del2;       // observations counted at the extraction / dropped by the overlap filter / by the CNV filter
    uint32_t *vtype_key, *bsum;
    uint32_t *node_of, *var_off; int32_t *nodes; uint32_t *node_off, *node_cap, *node_end;
    uint32_t *g_pack, *g_rank; int32_t *g_cnt;
    uint32_t *mrow_off; int32_t *mrow_cnt;        // by name: the read's merged row
    void *ukeys, *skeys; uint32_t *uvals, *svals; // node-major lists: (name, index in row) keys (32 or 64 bits), slots
    int32_t *t_node; uint8_t *t_flag; uint32_t *t_src; unsigned long long tail_lo, tail_size;
    float *edge; uint8_t *erec; uint32_t *node_pairs;
    LpsCounters *cnt;
};

void launch_clip_sort(unsigned n_clips, unsigned long long *keys, unsigned long long *keys_sorted, void *temp, size_t temp_bytes, hipStream_t s);
void launch_dense_names(int n_reads, const uint32_t *name_id, uint32_t name_max, unsigned long long *keys, unsigned long long *keys_s, uint32_t *head, uint32_t *gidx,
                        uint32_t *dense, void *temp, size_t temp_bytes, hipStream_t s);
#define LPS_CLIP_TAB_BITS 20   /* two tables of 2^20 counters each (8 MB, part of the zero pool): the count-min bound of the clip keys' multiplicity */
void launch_names(const GraphView &G, const ClipView &C, unsigned long long *clip_keys, const unsigned long long *arena_ctr, unsigned long long arena_size, uint32_t *clip_tab, hipStream_t s);
void launch_groups(const GraphView &G, double overlap_threshold, bool counted, hipStream_t s);
void launch_count_ranks(const GraphView &G, hipStream_t s);
struct CnvScratch { uint32_t *flag, *idx, *list, *n_list; uint8_t *fn, *pre; };
void launch_cnv_filter(const LpsCounters *cnt, int n_reads, int n_var, const RowDesc *rows,
                       const uint8_t *deleted, ObsRec *obs, const int32_t *vpos,
                       const int32_t *cnv_start, const int32_t *cnv_end, long long *agg_sum, int32_t *agg_cnt, double *miss,
                       CnvScratch &W, uint32_t *var_del2, void *temp, size_t temp_bytes, hipStream_t s);
void launch_debug_std_sort(int32_t *keys, uint8_t *payload, const long long *row_start, int n_rows, hipStream_t s);
void launch_var_scan(const GraphView &G, hipStream_t s);
void launch_graph_rows(const GraphView &G, int base_quality, int a_bits, bool key64, unsigned n_multi, hipStream_t s);
void launch_edges(const GraphView &G, int m_bits, int a_bits, bool key64, double edge_weight, double edge_threshold, hipStream_t s);
size_t scan_state_bytes(int n_var);
int scan_segments(int n_var);
void launch_vote_scan(const LpsCounters *cnt, int n_var, const int32_t *nodes, const int32_t *vpos, const uint8_t *erec,
                      int A, int distance, int8_t *hp_v, int32_t *blk_v, void *st_b, void *st_e, int32_t *seg_i32,
                      unsigned *n_replayed, int8_t *hp, int32_t *block, uint8_t *bmulti, int warm_tiles, hipStream_t s);
void launch_correction(const GraphView &G, const int32_t *block, const uint8_t *bmulti, const int8_t *hp, uint8_t *nstate, double read_conf, double snp_conf,
                       uint32_t *cnt4, int32_t *out_ps, uint8_t *out_gt, hipStream_t s);
