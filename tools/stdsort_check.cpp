// stdsort_check.cpp — csrc/stdsort_order.h against libstdc++'s std::sort itself (host; tests/test_host.py builds and runs it with g++).
// The comparator is the reference's: indices sorted by key descending, ties in whatever order the algorithm leaves them.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>
#include "../msu-latentafis_amd/csrc/stdsort_order.h"

static long checked = 0, bad = 0;

template <typename I> static void one_t(const std::vector<uint32_t>& key, int K, int depth)
{
    const int n = (int)key.size();
    std::vector<I> a(n), b(n);
    std::iota(a.begin(), a.end(), 0); b = a;
    auto comp = [&key](I x, I y) { return key[x] > key[y]; };
    if (depth < 0) std::sort(a.begin(), a.end(), comp);
    else if (n > 1) {                                                    // the same algorithm with a forced depth limit (libstdc++'s own internals)
        std::__introsort_loop(a.begin(), a.end(), (long)depth, __gnu_cxx::__ops::__iter_comp_iter(comp));
        std::__final_insertion_sort(a.begin(), a.end(), __gnu_cxx::__ops::__iter_comp_iter(comp));
    }
    int stack[3 * 64];
    afis::stdsort_prefix(b.data(), n, K, key.data(), stack, depth);
    ++checked;
    for (int i = 0; i < std::min(K, n); ++i)
        if (a[i] != b[i]) { if (bad < 5) fprintf(stderr, "mismatch: n %d K %d depth %d at %d: %u vs %u\n", n, K, depth, i, a[i], b[i]); ++bad; return; }
}

static void one(const std::vector<uint32_t>& key, int K, int depth) { one_t<uint16_t>(key, K, depth); }

// the closed form of the partition (what the device's wave executes in parallel) against the pointer walk, on the same array
static void partition_forms(const std::vector<uint32_t>& key)
{
    const int n = (int)key.size();
    if (n < 17) return;
    std::vector<uint16_t> a(n), b(n), lp(n), rp(n);
    std::iota(a.begin(), a.end(), 0);
    std::mt19937 r2(n * 7 + 1); std::shuffle(a.begin(), a.end(), r2); b = a;
    const afis::SsoCtx<uint16_t> ca{a.data(), key.data()}, cb{b.data(), key.data()};
    const int c1 = afis::sso_partition_pivot(ca, 0, n), c2 = afis::sso_partition_pivot_closed(cb, 0, n, lp.data(), rp.data());
    ++checked;
    if (c1 != c2 || a != b) { if (bad < 5) fprintf(stderr, "partition forms differ: n %d cut %d vs %d\n", n, c1, c2); ++bad; }
}

int main()
{
    std::mt19937 rng(12345);
    for (int n = 0; n <= 300; ++n)
        for (int rep = 0; rep < 12; ++rep) {
            std::vector<uint32_t> key(n);
            const int kind = rep % 6;                                   // 0: distinct, 1: few distinct values, 2: mostly zero (the zero-fill case), 3: all equal, 4: ascending, 5: organ pipe with ties
            for (int i = 0; i < n; ++i) {
                switch (kind) {
                case 0: key[i] = rng(); break;
                case 1: key[i] = rng() % 4; break;
                case 2: key[i] = (rng() % 100) < 4 ? 1000 + rng() % 1000 : 0; break;
                case 3: key[i] = 7; break;
                case 4: key[i] = i / 3; break;
                default: key[i] = (uint32_t)std::min(i, n - 1 - i) / 2; break;
                }
            }
            one(key, n, -1); one(key, 120, -1); one(key, 17, -1); partition_forms(key);
            for (int d : {0, 1, 2, 3}) { one(key, n, d); one(key, 120, d); }
        }
    for (int rep = 0; rep < 400; ++rep) {                                // the shapes of real tasks: 20..64 x 20..128 entries, a few dozen positive, the rest zero
        const int n = (20 + rng() % 45) * (20 + rng() % 109);
        std::vector<uint32_t> key(n, 0);
        const int pos = rng() % 130;
        for (int i = 0; i < pos; ++i) key[rng() % n] = 1 + rng() % 100000;
        if (rep % 5 == 0) for (int i = 0; i < n; ++i) if (rng() % 3 == 0) key[i] = 1 + rng() % 50;    // many positive ties as well
        one(key, 120, -1); partition_forms(key);
        if (rep % 40 == 0) { one(key, n, -1); one(key, 120, 3); one(key, 120, 5); }
    }
    for (int rep = 0; rep < 24; ++rep) {                                 // pairs beyond 8192 similarities: 32-bit indices (the device keeps these arrays in global scratch)
        const int n = 8193 + rng() % (rep < 20 ? 60000 : 400000);
        std::vector<uint32_t> key(n, 0);
        const int pos = rng() % 130;
        for (int i = 0; i < pos; ++i) key[rng() % n] = 1 + rng() % 100000;
        if (rep % 4 == 0) for (int i = 0; i < n; ++i) if (rng() % 3 == 0) key[i] = 1 + rng() % 50;
        one_t<uint32_t>(key, 120, -1);
        if (rep % 8 == 0) { one_t<uint32_t>(key, 120, 4); one_t<uint32_t>(key, n, -1); }
    }
    printf("stdsort_order: %ld comparisons with libstdc++, %ld mismatches\n", checked, bad);
    return bad ? 1 : 0;
}
