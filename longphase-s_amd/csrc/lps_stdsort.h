// lps_stdsort.h — what `std::sort(v.begin(), v.end(), less_than_key)` of libstdc++ does to a merged read's variants
// (src/phase/PhasingGraph.cpp:854, comparator src/shared/Util.cpp:3-5 compares positions only), restated step by step so that the order it leaves
// among EQUAL positions is reproduced exactly: the reference's fp32 edge sums depend on that order.
// Algorithm (bits/stl_algo.h): introsort loop until a partition has <= 16 elements (median of first+1 / middle / last-1 moved to the front,
// unguarded Hoare partition, heapsort when 2*floor(log2 n) levels are used up), then one final insertion sort pass.
// Elements are (key, payload) pairs held in two parallel arrays; indices play the iterators.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define LPS_HD __host__ __device__
#else
#define LPS_HD
#endif

struct StdSortPairs {
    int32_t *k; uint8_t *p;
    LPS_HD bool less(int a, int b) const { return k[a] < k[b]; }
    LPS_HD void swap(int a, int b) { const int32_t t = k[a]; k[a] = k[b]; k[b] = t; const uint8_t u = p[a]; p[a] = p[b]; p[b] = u; }
    LPS_HD void move(int dst, int src) { k[dst] = k[src]; p[dst] = p[src]; }
};

LPS_HD inline void stdsort_move_median_to_first(StdSortPairs &v, int result, int a, int b, int c) {
    if (v.less(a, b)) { if (v.less(b, c)) v.swap(result, b); else if (v.less(a, c)) v.swap(result, c); else v.swap(result, a); }
    else if (v.less(a, c)) v.swap(result, a);
    else if (v.less(b, c)) v.swap(result, c);
    else v.swap(result, b);
}
LPS_HD inline int stdsort_unguarded_partition(StdSortPairs &v, int first, int last, int pivot) {
    for (;;) {
        while (v.less(first, pivot)) ++first;
        --last;
        while (v.less(pivot, last)) --last;
        if (!(first < last)) return first;
        v.swap(first, last);
        ++first;
    }
}
// heap helpers with an explicit value (key, payload) as in __adjust_heap / __push_heap
LPS_HD inline void stdsort_push_heap(StdSortPairs &v, int first, int hole, int top, int32_t vk, uint8_t vp) {
    int parent = (hole - 1) / 2;
    while (hole > top && v.k[first + parent] < vk) { v.move(first + hole, first + parent); hole = parent; parent = (hole - 1) / 2; }
    v.k[first + hole] = vk; v.p[first + hole] = vp;
}
LPS_HD inline void stdsort_adjust_heap(StdSortPairs &v, int first, int hole, int len, int32_t vk, uint8_t vp) {
    const int top = hole; int child = hole;
    while (child < (len - 1) / 2) { child = 2 * (child + 1); if (v.k[first + child] < v.k[first + child - 1]) --child; v.move(first + hole, first + child); hole = child; }
    if ((len & 1) == 0 && child == (len - 2) / 2) { child = 2 * (child + 1); v.move(first + hole, first + child - 1); hole = child - 1; }
    stdsort_push_heap(v, first, hole, top, vk, vp);
}
LPS_HD inline void stdsort_heapsort(StdSortPairs &v, int first, int last) {      // __partial_sort(first, last, last)
    const int len = last - first;
    if (len >= 2) for (int parent = (len - 2) / 2;; --parent) { const int32_t vk = v.k[first + parent]; const uint8_t vp = v.p[first + parent]; stdsort_adjust_heap(v, first, parent, len, vk, vp); if (parent == 0) break; }
    while (last - first > 1) { --last; const int32_t vk = v.k[last]; const uint8_t vp = v.p[last]; v.move(last, first); stdsort_adjust_heap(v, first, 0, last - first, vk, vp); }
}
LPS_HD inline void stdsort_unguarded_linear_insert(StdSortPairs &v, int last) {
    const int32_t vk = v.k[last]; const uint8_t vp = v.p[last]; int next = last - 1;
    while (vk < v.k[next]) { v.move(last, next); last = next; --next; }
    v.k[last] = vk; v.p[last] = vp;
}
LPS_HD inline void stdsort_insertion_sort(StdSortPairs &v, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        if (v.less(i, first)) { const int32_t vk = v.k[i]; const uint8_t vp = v.p[i]; for (int j = i; j > first; --j) v.move(j, j - 1); v.k[first] = vk; v.p[first] = vp; }
        else stdsort_unguarded_linear_insert(v, i);
    }
}
// std::sort over [0, n)
LPS_HD inline void stdsort_pairs(int32_t *keys, uint8_t *payload, int n) {
    if (n <= 1) return;
    StdSortPairs v{keys, payload};
    int lg = 0; for (int m = n; m > 1; m >>= 1) ++lg;
    // __introsort_loop: the recursion on the right part becomes an explicit stack (the parts are independent)
    int stk_first[64], stk_last[64], stk_depth[64]; int sp = 0;
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
    // __final_insertion_sort
    if (n > 16) { stdsort_insertion_sort(v, 0, 16); for (int i = 16; i < n; ++i) stdsort_unguarded_linear_insert(v, i); }
    else stdsort_insertion_sort(v, 0, n);
}
