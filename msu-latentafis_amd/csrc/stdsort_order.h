// stdsort_order.h — the permutation libstdc++'s std::sort produces, restated so that it can run on the device (option "s3_tie_order" 1).
//
// Why.  The reference sorts the nL x nR candidate norms with std::sort on a NON-STRICT key (matching/matcher.cpp:473-476: indices 0 .. n-1, comparator
// norm[a] > norm[b]) and keeps the first 120.  Where norms tie, which index comes first is whatever libstdc++'s introsort does with the whole array — and ties
// are not rare where it matters: a (latent, rolled) pair with fewer than 120 POSITIVE similarities fills its list with zero-norm entries, all tied (every tiny
// latent template; 0.4 % of the lists of the structured workload: tools/tie_site_sweep.py).  The library's default orders tied entries by ascending element index;
// with the option on, the any-shape candidate kernel (minu.hip::k_minu_cands) runs THIS restatement instead and delivers the list the reference binary delivers
// (index type I = uint16_t with the arrays in LDS up to 8192 similarities, uint32_t with the arrays in the workgroup's global scratch beyond).
//
// What is restated (the published algorithm of libstdc++'s <bits/stl_algo.h>, unchanged since GCC 4; written from its description, not copied):
//   sort            = introsort loop with depth limit 2 floor(log2 n), then a final insertion sort
//   introsort loop  : while the range has more than 16 elements — depth limit reached: heap sort of the range (make_heap + sort_heap); otherwise the median of
//                     (first + 1, middle, last - 1) goes to `first`, an UNGUARDED Hoare partition of (first + 1, last) around it, recursion on the right part,
//                     iteration on the left
//   final insertion : plain insertion sort (the first 16 guarded, the rest unguarded: the same moves, because every left neighbour block holds keys >= the block's)
// Only the first K positions are wanted (K = 120): a range that starts at or beyond K can neither receive nor send an element across its left boundary once its
// parent has been partitioned (everything left of it is >= everything in it, and insertion moves an element left only past STRICTLY smaller keys), so such ranges
// are skipped — the work is about 2 n element steps instead of n log n.
//
// tests/test_host.py::test_stdsort_order_equals_libstdcxx compiles this header on the host and compares it with std::sort itself (random and tie-heavy arrays,
// every n up to 300, a sample up to 8192 with 16-bit indices and up to 400 000 with 32-bit ones, forced depth limits for the heap-sort branch).
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define SSO_FN __host__ __device__ inline
#else
#define SSO_FN inline
#endif

namespace afis {

// comp(a, b) == key[a] > key[b]: "a sorts before b" (descending keys), as the reference's lambda
template <typename I> struct SsoCtx { I* A; const uint32_t* key; };   // I: uint16_t (arrays in LDS, n <= 65535) or uint32_t (arrays in global scratch)
template <typename I> SSO_FN bool sso_before(const SsoCtx<I>& c, I a, I b) { return c.key[a] > c.key[b]; }
template <typename I> SSO_FN void sso_swap(I* A, int i, int j) { const I t = A[i]; A[i] = A[j]; A[j] = t; }

// __adjust_heap + __push_heap on the range starting at `f` (positions relative to f), "less" = sso_before
template <typename I> SSO_FN void sso_adjust_heap(const SsoCtx<I>& c, int f, int hole, int len, I value)
{
    I* A = c.A + f;
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (sso_before(c, A[child], A[child - 1])) --child;
        A[hole] = A[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        A[hole] = A[child - 1];
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && sso_before(c, A[parent], value)) {
        A[hole] = A[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    A[hole] = value;
}
// __partial_sort(first, last, last) = make_heap + sort_heap: the depth-limit branch
template <typename I> SSO_FN void sso_heap_sort(const SsoCtx<I>& c, int f, int l)
{
    const int len = l - f;
    if (len < 2) return;
    for (int parent = (len - 2) / 2;; --parent) {
        sso_adjust_heap(c, f, parent, len, c.A[f + parent]);
        if (parent == 0) break;
    }
    for (int last = l; last - f > 1;) {
        --last;
        const I value = c.A[last];
        c.A[last] = c.A[f];
        sso_adjust_heap(c, f, 0, last - f, value);
    }
}
template <typename I> SSO_FN int sso_partition_pivot(const SsoCtx<I>& c, int f, int l)
{
    I* A = c.A;
    const int mid = f + (l - f) / 2;
    const int a = f + 1, b = mid, cc = l - 1;                              // __move_median_to_first(first, first + 1, mid, last - 1)
    if (sso_before(c, A[a], A[b])) {
        if (sso_before(c, A[b], A[cc])) sso_swap(A, f, b);
        else if (sso_before(c, A[a], A[cc])) sso_swap(A, f, cc);
        else sso_swap(A, f, a);
    } else if (sso_before(c, A[a], A[cc])) sso_swap(A, f, a);
    else if (sso_before(c, A[b], A[cc])) sso_swap(A, f, cc);
    else sso_swap(A, f, b);
    int first = f + 1, last = l;                                          // __unguarded_partition(first + 1, last, pivot = *first)
    const I pivot = A[f];
    for (;;) {
        while (sso_before(c, A[first], pivot)) ++first;
        --last;
        while (sso_before(c, pivot, A[last])) --last;
        if (!(first < last)) return first;
        sso_swap(A, first, last);
        ++first;
    }
}
// The same partition in CLOSED FORM — what lets a wave do it in parallel (minu.hip::sso_partition_wave is this function with ballots for the loops).  The unguarded
// Hoare partition moves two pointers towards each other and swaps where both have stopped; until they cross, neither pointer ever reads a position the other has
// written.  So the stops are those of the ORIGINAL array: left stops L_1 < L_2 < ... = the positions of (first + 1, last) whose key is <= the pivot's, right stops
// R_1 > R_2 > ... = those whose key is >= the pivot's; swap i exchanges L_i and R_i while L_i < R_i (say m swaps); and the partition point is
// min(L_{m+1}, R_m) (R_m now holds a left-stop element; R_0 = nothing).  lpos / rpos: scratch of (last - first) entries each.
template <typename I> SSO_FN int sso_partition_pivot_closed(const SsoCtx<I>& c, int f, int l, I* lpos, I* rpos)
{
    I* A = c.A;
    const int mid = f + (l - f) / 2;
    const int a = f + 1, b = mid, cc = l - 1;
    if (sso_before(c, A[a], A[b])) {
        if (sso_before(c, A[b], A[cc])) sso_swap(A, f, b);
        else if (sso_before(c, A[a], A[cc])) sso_swap(A, f, cc);
        else sso_swap(A, f, a);
    } else if (sso_before(c, A[a], A[cc])) sso_swap(A, f, a);
    else if (sso_before(c, A[b], A[cc])) sso_swap(A, f, cc);
    else sso_swap(A, f, b);
    const uint32_t pk = c.key[A[f]];
    int nl = 0, nr = 0;
    for (int p = f + 1; p < l; ++p) {
        const uint32_t k = c.key[A[p]];
        if (k <= pk) lpos[nl++] = (I)p;                              // !before(A[p], pivot): the left pointer stops here
        if (k >= pk) rpos[nr++] = (I)p;                              // !before(pivot, A[p]): the right pointer stops here (ascending; the i-th from the right is rpos[nr - i])
    }
    int m = 0;
    while (m < nl && m < nr && lpos[m] < rpos[nr - 1 - m]) ++m;
    for (int i = 0; i < m; ++i) sso_swap(A, lpos[i], rpos[nr - 1 - i]);
    int cut = m < nl ? (int)lpos[m] : 0x7fffffff;
    if (m > 0 && (int)rpos[nr - m] < cut) cut = rpos[nr - m];
    return cut;
}

SSO_FN int sso_floor_log2(int n) { int k = 0; while (n > 1) { n >>= 1; ++k; } return k; }

// A[0 .. n) = a permutation (normally 0 .. n-1 in order); afterwards A[0 .. min(K, n)) is what std::sort(A, A + n, comp) leaves there.
// stack: 3 * 64 ints of scratch (first, last, depth of the pending right parts; the depth limit bounds its use by 2 log2 n <= 26 entries).
// depth0 < 0: the real limit; tests pass small values to reach the heap-sort branch.
template <typename I> SSO_FN void stdsort_prefix(I* A, int n, int K, const uint32_t* key, int* stack, int depth0 = -1)
{
    const SsoCtx<I> c{A, key};
    if (n < 2) return;
    if (K > n) K = n;
    int sp = 0;
    int kend = 0;                                                         // end of the processed leaf ranges: [0, kend) is partitioned down to blocks of <= 16
    stack[0] = 0; stack[1] = n; stack[2] = depth0 >= 0 ? depth0 : 2 * sso_floor_log2(n); sp = 1;
    while (sp > 0) {
        --sp;
        int f = stack[3 * sp], l = stack[3 * sp + 1], d = stack[3 * sp + 2];
        if (f >= K) continue;                                             // nothing of this range can reach the first K positions
        bool sorted_by_heap = false;
        while (l - f > 16) {
            if (d == 0) { sso_heap_sort(c, f, l); sorted_by_heap = true; break; }
            --d;
            const int cut = sso_partition_pivot(c, f, l);
            stack[3 * sp] = cut; stack[3 * sp + 1] = l; stack[3 * sp + 2] = d; ++sp;
            l = cut;
        }
        (void)sorted_by_heap;
        if (l > kend) kend = l;
    }
    // __final_insertion_sort over the processed prefix (what lies beyond it cannot move into it)
    for (int i = 1; i < kend; ++i) {
        const I val = A[i];
        int j = i;
        while (j > 0 && sso_before(c, val, A[j - 1])) { A[j] = A[j - 1]; --j; }
        A[j] = val;
    }
}

}  // namespace afis
