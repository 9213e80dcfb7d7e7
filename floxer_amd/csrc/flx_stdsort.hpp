// libstdc++'s std::sort reproduced step for step, for host and device (see seed_select_kernel in flx_device.hip: the reference orders
// hit groups and anchor buckets with std::sort, its comparators tie, and what std::sort does with equal elements shows in the
// output). tests/stdsort_check.cpp compares it with the real std::sort.
#pragma once

#if defined(__HIPCC__)
#define FLX_SORT_HD __host__ __device__
#else
#define FLX_SORT_HD
#endif

namespace flx {

// libstdc++'s std::sort (bits/stl_algo.h: introsort with median-of-three pivots down to 16 elements, then insertion sort), step
// for step: the reference orders hit groups and anchor buckets with it, its comparators tie often, and what it does with equal
// elements shows in the output. Returns false if the depth limit is reached (the heap-sort fallback is left to the host).
// An element held aside while others move. Word by word: a struct copied whole between LDS and a local is a memcpy the GPU compiler
// leaves in scratch memory.
template <class T>
struct SortVal {
    static_assert(sizeof(T) % 4 == 0, "elements are whole words");
    unsigned w[sizeof(T) / 4];
    // (four-byte memcpys: one load / store each, and no aliasing assumptions for the host compiler to act on)
    FLX_SORT_HD explicit SortVal(T const& src) { for (unsigned k = 0; k < sizeof(T) / 4; ++k) __builtin_memcpy(&w[k], reinterpret_cast<const char*>(&src) + 4 * k, 4); }
    FLX_SORT_HD T get() const { T t; for (unsigned k = 0; k < sizeof(T) / 4; ++k) __builtin_memcpy(reinterpret_cast<char*>(&t) + 4 * k, &w[k], 4); return t; }
    FLX_SORT_HD void put(T& dst) const { for (unsigned k = 0; k < sizeof(T) / 4; ++k) __builtin_memcpy(reinterpret_cast<char*>(&dst) + 4 * k, &w[k], 4); }
};
template <class T>
FLX_SORT_HD inline void sort_move(T& dst, T const& src) { SortVal<T>(src).put(dst); }

// (up to 16 elements std::sort is its final insertion sort alone: no pivots, no stack - and stable)
template <class T, class Less>
FLX_SORT_HD void insertion_sort_emulated(T* a, int n, Less less) {
    for (int i = 1; i < n; ++i) {
        SortVal<T> const val(a[i]);
        if (less(val.get(), a[0])) { for (int j = i; j > 0; --j) sort_move(a[j], a[j - 1]); val.put(a[0]); }
        else {
            int last = i, next = i - 1;
            while (less(val.get(), a[next])) { sort_move(a[last], a[next]); last = next; --next; }
            val.put(a[last]);
        }
    }
}

// stacks: 48 ints of working storage for the partitions still to do (a kernel passes LDS: indexed private arrays live in scratch memory)
template <class T, class Less>
FLX_SORT_HD bool std_sort_emulated(T* a, int n, Less less, int* stacks) {
    auto swap_at = [&](int i, int j) { SortVal<T> const t(a[i]); sort_move(a[i], a[j]); t.put(a[j]); };
    auto unguarded_linear_insert = [&](int last) {
        SortVal<T> const val(a[last]);
        int next = last - 1;
        while (less(val.get(), a[next])) { sort_move(a[last], a[next]); last = next; --next; }
        val.put(a[last]);
    };
    auto insertion_sort = [&](int first, int last) {
        for (int i = first + 1; i < last; ++i) {
            if (less(a[i], a[first])) { SortVal<T> const val(a[i]); for (int j = i; j > first; --j) sort_move(a[j], a[j - 1]); val.put(a[first]); }
            else unguarded_linear_insert(i);
        }
    };
    if (n <= 1) return true;
    if (n > 16) {
        // __introsort_loop: recursion on the right part, iteration on the left; at most 2*floor(log2 n) levels
        int depth0 = 0;
        for (int m = n; m > 1; m >>= 1) ++depth0;
        depth0 *= 2;
        int* const stack_first = stacks; int* const stack_last = stacks + 16; int* const stack_depth = stacks + 32;
        int sp = 0;
        int first = 0, last = n, depth = depth0;
        while (true) {
            while (last - first > 16) {
                if (depth == 0) return false;
                --depth;
                int const mid = first + (last - first) / 2;
                // __move_median_to_first(first, first + 1, mid, last - 1)
                int const x = first + 1, y = mid, z = last - 1;
                if (less(a[x], a[y])) {
                    if (less(a[y], a[z])) swap_at(first, y);
                    else if (less(a[x], a[z])) swap_at(first, z);
                    else swap_at(first, x);
                } else if (less(a[x], a[z])) swap_at(first, x);
                else if (less(a[y], a[z])) swap_at(first, z);
                else swap_at(first, y);
                // __unguarded_partition(first + 1, last, pivot = first)
                int lo = first + 1, hi = last;
                while (true) {
                    while (less(a[lo], a[first])) ++lo;
                    --hi;
                    while (less(a[first], a[hi])) --hi;
                    if (!(lo < hi)) break;
                    swap_at(lo, hi);
                    ++lo;
                }
                int const cut = lo;
                if (sp < 16) { stack_first[sp] = cut; stack_last[sp] = last; stack_depth[sp] = depth; ++sp; } else return false;
                last = cut;
            }
            if (sp == 0) break;
            --sp;
            first = stack_first[sp]; last = stack_last[sp]; depth = stack_depth[sp];
        }
        // (the recursion of the original handles the right part before it continues with the left one; the parts are disjoint,
        //  so the order in which they are partitioned does not change the result)
        insertion_sort(0, 16);
        for (int i = 16; i < n; ++i) unguarded_linear_insert(i);
        return true;
    }
    insertion_sort(0, n);
    return true;
}
template <class T, class Less>
FLX_SORT_HD bool std_sort_emulated(T* a, int n, Less less) {
    int stacks[48];
    return std_sort_emulated(a, n, less, stacks);
}

}  // namespace flx
