// lps_graph.h — launch wrappers of lps_graph.hip
#pragma once
#include "lps_kernels.h"

struct GraphTemp { static size_t need(size_t n_sort); };

void sort_keys64_range(void *temp, size_t temp_bytes, const unsigned long long *in, unsigned long long *out, size_t n, int begin_bit, int end_bit, hipStream_t s);
void sort_keys64(void *temp, size_t temp_bytes, const unsigned long long *in, unsigned long long *out, size_t n, int bits, hipStream_t s);
void sort_pairs64(void *temp, size_t temp_bytes, const unsigned long long *kin, unsigned long long *kout, const uint32_t *vin, uint32_t *vout, size_t n, int bits, hipStream_t s);
void exscan_u32(void *temp, size_t temp_bytes, const uint32_t *in, uint32_t *out, size_t n, hipStream_t s);

void launch_clip_keys(const ClipView &C, const RowDesc *rows, int n_reads, unsigned long long *keys, LpsCounters *cnt, hipStream_t s);
void launch_clip_sort(unsigned n_clips, unsigned long long *keys, unsigned long long *keys_sorted, void *temp, size_t temp_bytes, hipStream_t s);
void launch_name_keys(int n_reads, const uint32_t *name_id, const RowDesc *rows, unsigned long long *keys, LpsCounters *cnt,
                      const unsigned long long *arena_ctr, unsigned long long arena_size, hipStream_t s);
void launch_groups(const unsigned long long *skeys, int n_reads, LpsCounters *cnt, uint32_t *head, uint32_t *gidx,
                   uint32_t *gstart, uint32_t *read_group, void *temp, size_t temp_bytes, hipStream_t s);
void launch_overlap_filter(const unsigned long long *skeys, const uint32_t *gstart, const LpsCounters *cnt, int n_reads,
                           const RowDesc *rows, const ObsRec *obs, const int32_t *vpos,
                           double thr, uint32_t *stack, uint8_t *deleted, hipStream_t s);
struct CnvScratch { uint32_t *flag, *idx, *list, *n_list; uint8_t *fn, *pre; };
void launch_cnv_filter(const LpsCounters *cnt, int n_reads, int n_var, const RowDesc *rows,
                       const uint8_t *deleted, ObsRec *obs, const int32_t *vpos,
                       const int32_t *cnv_start, const int32_t *cnv_end, long long *agg_sum, int32_t *agg_cnt, double *miss,
                       CnvScratch &W, void *temp, size_t temp_bytes, hipStream_t s);
void launch_nodes(int n_reads, int n_var, const RowDesc *rows, const uint8_t *deleted,
                  const ObsRec *obs, uint32_t *is_node, uint32_t *vtype_key, uint32_t *node_of,
                  int32_t *nodes, uint8_t *ntype, int base_quality, int32_t *g_node, uint8_t *g_flag, uint32_t *g_pack, uint16_t *g_rank, int32_t *g_cnt,
                  LpsCounters *cnt, uint32_t *node_cnt, void *temp, size_t temp_bytes, hipStream_t s);
void launch_debug_std_sort(int32_t *keys, uint8_t *payload, const long long *row_start, int n_rows, hipStream_t s);
void launch_merge_rows(const unsigned long long *skeys, const uint32_t *gstart, LpsCounters *cnt, int n_reads,
                       const RowDesc *rows, const int32_t *g_cnt, int32_t *g_node, uint8_t *g_flag,
                       unsigned long long tail_lo, unsigned long long tail_size, uint32_t *mrow_off, int32_t *mrow_cnt, uint32_t *multi_list, uint32_t *g_pack, uint32_t *t_src, hipStream_t s);
void launch_node_lists(LpsCounters *cnt, int n_reads, int n_var, const RowDesc *rows, const int32_t *g_cnt, const uint32_t *read_group, const uint32_t *gstart,
                       const uint32_t *mrow_off, const int32_t *mrow_cnt, const uint32_t *multi_list,
                       const int32_t *g_node, const uint16_t *g_rank, const uint32_t *t_src, uint32_t tail_lo, int a_bits,
                       unsigned long long *keys, uint32_t *vals, uint32_t *node_off, uint32_t *node_cnt, void *temp, size_t temp_bytes, hipStream_t s);
void launch_edges(LpsCounters *cnt, int n_var, const uint32_t *node_off, const uint32_t *node_end,
                  const unsigned long long *ukeys, const uint32_t *uvals, unsigned long long *skeys, uint32_t *svals, const uint32_t *mrow_off, const int32_t *mrow_cnt,
                  int m_bits, int a_bits, const uint32_t *g_pack, uint32_t tail_lo, int A, double edge_weight,
                  double edge_threshold, const uint8_t *ntype, float *edge, uint8_t *erec, uint32_t *node_pairs, hipStream_t s);
size_t scan_state_bytes(int n_var);
int scan_segments(int n_var);
void launch_vote_scan(const LpsCounters *cnt, int n_var, const int32_t *nodes, const int32_t *vpos, const uint8_t *erec,
                      int A, int distance, int8_t *hp_v, int32_t *blk_v, void *st_b, void *st_e, int32_t *seg_i32,
                      unsigned *n_replayed, int8_t *hp, int32_t *block, int warm_tiles, hipStream_t s);
void launch_correction(LpsCounters *cnt, int n_reads, int n_var, const RowDesc *rows, const int32_t *g_cnt,
                       const int32_t *g_node, const uint8_t *g_flag, const int32_t *nodes, const int32_t *vpos,
                       const int32_t *block, uint32_t *bsize, const int8_t *hp, const uint8_t *ntype, const uint32_t *node_pairs,
                       uint8_t *nstate, double read_conf, double snp_conf, uint32_t *cnt4, int32_t *out_ps, uint8_t *out_gt, hipStream_t s);
