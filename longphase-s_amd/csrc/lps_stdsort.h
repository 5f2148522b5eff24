// lps_stdsort.h — what `std::sort(v.begin(), v.end(), less_than_key)` of libstdc++ does to a merged read's variants
// (src/phase/PhasingGraph.cpp:854, comparator src/shared/Util.cpp:3-5 compares positions only), restated step by step so that the order it leaves
// among EQUAL positions is reproduced exactly: the reference's fp32 edge sums depend on that order.
// Algorithm (bits/stl_algo.h): introsort loop until a partition has <= 16 elements (median of first+1 / middle / last-1 moved to the front,
// unguarded Hoare partition, heapsort when 2*floor(log2 n) levels are used up), then one final insertion sort pass.
// Elements are (key, payload) pairs behind an accessor A { K(i), P(i), set(i, key, payload) }; indices play the iterators.  Accessors: plain
// arrays in host memory (test hook) or HBM (rows of more than 1024 elements on the GPU).  Shorter rows are sorted by a whole wavefront in LDS:
// lps_graph.hip wave_std_sort walks the same loop with every partition step done by 64 lanes, and borrows the median and heapsort steps from here.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define LPS_HD __host__ __device__
#else
#define LPS_HD
#endif

struct StdSortArrays {
    int32_t *k; uint8_t *p;
    LPS_HD int32_t K(int i) const { return k[i]; }
    LPS_HD uint8_t P(int i) const { return p[i]; }
    LPS_HD void set(int i, int32_t kk, uint8_t pp) { k[i] = kk; p[i] = pp; }
};

template <class A> LPS_HD inline bool stdsort_less(A &v, int a, int b) { return v.K(a) < v.K(b); }
template <class A> LPS_HD inline void stdsort_swap(A &v, int a, int b) { const int32_t ka = v.K(a), kb = v.K(b); const uint8_t pa = v.P(a), pb = v.P(b); v.set(a, kb, pb); v.set(b, ka, pa); }
template <class A> LPS_HD inline void stdsort_move(A &v, int dst, int src) { v.set(dst, v.K(src), v.P(src)); }

template <class A> LPS_HD inline void stdsort_move_median_to_first(A &v, int result, int a, int b, int c) {
    if (stdsort_less(v, a, b)) { if (stdsort_less(v, b, c)) stdsort_swap(v, result, b); else if (stdsort_less(v, a, c)) stdsort_swap(v, result, c); else stdsort_swap(v, result, a); }
    else if (stdsort_less(v, a, c)) stdsort_swap(v, result, a);
    else if (stdsort_less(v, b, c)) stdsort_swap(v, result, c);
    else stdsort_swap(v, result, b);
}
template <class A> LPS_HD inline int stdsort_unguarded_partition(A &v, int first, int last, int pivot) {
    for (;;) {
        while (stdsort_less(v, first, pivot)) ++first;
        --last;
        while (stdsort_less(v, pivot, last)) --last;
        if (!(first < last)) return first;
        stdsort_swap(v, first, last);
        ++first;
    }
}
// heap helpers with an explicit value (key, payload) as in __adjust_heap / __push_heap
template <class A> LPS_HD inline void stdsort_push_heap(A &v, int first, int hole, int top, int32_t vk, uint8_t vp) {
    int parent = (hole - 1) / 2;
    while (hole > top && v.K(first + parent) < vk) { stdsort_move(v, first + hole, first + parent); hole = parent; parent = (hole - 1) / 2; }
    v.set(first + hole, vk, vp);
}
template <class A> LPS_HD inline void stdsort_adjust_heap(A &v, int first, int hole, int len, int32_t vk, uint8_t vp) {
    const int top = hole; int child = hole;
    while (child < (len - 1) / 2) { child = 2 * (child + 1); if (v.K(first + child) < v.K(first + child - 1)) --child; stdsort_move(v, first + hole, first + child); hole = child; }
    if ((len & 1) == 0 && child == (len - 2) / 2) { child = 2 * (child + 1); stdsort_move(v, first + hole, first + child - 1); hole = child - 1; }
    stdsort_push_heap(v, first, hole, top, vk, vp);
}
template <class A> LPS_HD inline void stdsort_heapsort(A &v, int first, int last) {      // __partial_sort(first, last, last)
    const int len = last - first;
    if (len >= 2) for (int parent = (len - 2) / 2;; --parent) { const int32_t vk = v.K(first + parent); const uint8_t vp = v.P(first + parent); stdsort_adjust_heap(v,
            first, parent, len, vk, vp); if (parent == 0) break; }
    while (last - first > 1) { --last; const int32_t vk = v.K(last); const uint8_t vp = v.P(last); stdsort_move(v, last, first); stdsort_adjust_heap(v, first, 0, last - first, vk, vp); }
}
template <class A> LPS_HD inline void stdsort_unguarded_linear_insert(A &v, int last) {
    const int32_t vk = v.K(last); const uint8_t vp = v.P(last); int next = last - 1;
    while (vk < v.K(next)) { stdsort_move(v, last, next); last = next; --next; }
    v.set(last, vk, vp);
}
template <class A> LPS_HD inline void stdsort_insertion_sort(A &v, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        if (stdsort_less(v, i, first)) { const int32_t vk = v.K(i); const uint8_t vp = v.P(i); for (int j = i; j > first; --j) stdsort_move(v, j, j - 1); v.set(first, vk, vp); }
        else stdsort_unguarded_linear_insert(v, i);
    }
}
// std::sort over [0, n)
// `stk`: 3 * 64 ints of work space for the explicit stack (the GPU passes LDS: a dynamically indexed local array would live in scratch memory)
template <class A> LPS_HD inline void stdsort_run(A &v, int n, int *stk) {
    if (n <= 1) return;
    int lg = 0; for (int m = n; m > 1; m >>= 1) ++lg;
    // __introsort_loop: the recursion on the right part becomes an explicit stack (the parts are independent)
    int *stk_first = stk, *stk_last = stk + 64, *stk_depth = stk + 128; int sp = 0;
    stk_first[0] = 0; stk_last[0] = n; stk_depth[0] = 2 * lg; sp = 1;
    while (sp) {
        --sp; int first = stk_first[sp], last = stk_last[sp], depth = stk_depth[sp];
        while (last - first > 16) {
            if (depth == 0) { stdsort_heapsort(v, first, last); break; }
            --depth;
            const int mid = first + (last - first) / 2;
            stdsort_move_median_to_first(v, first, first + 1, mid, last - 1);
            const int cut = stdsort_unguarded_partition(v, first + 1, last, first);
            if (sp < 64) { stk_first[sp] = cut; stk_last[sp] = last; stk_depth[sp] = depth; ++sp; }
            last = cut;
        }
    }
    // __final_insertion_sort (stable: equal to a stable sort of what the loop above left behind - lps_graph.hip's wave_std_sort uses that)
    if (n > 16) { stdsort_insertion_sort(v, 0, 16); for (int i = 16; i < n; ++i) stdsort_unguarded_linear_insert(v, i); }
    else stdsort_insertion_sort(v, 0, n);
}
LPS_HD inline void stdsort_pairs(int32_t *keys, uint8_t *payload, int n) { StdSortArrays v{keys, payload}; int stk[192]; stdsort_run(v, n, stk); }
