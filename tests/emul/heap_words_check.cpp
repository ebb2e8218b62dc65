// tests/emul/heap_words_check.cpp - csrc/mc_heap_words.h (the heap sort of k_heap_lanes, written against an accessor)
// against the plain libstdc++ loop (mc_sort_impl.h's mc_adjust_heap / mc_heapsort over words) and against the C++ library's own
// make_heap + sort_heap, on 200,000 arrays of 0 .. 600 words with few distinct keys (ties everywhere).  Built and run by tests/test_emul.py.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "mc_heap_words.h"
struct Acc { uint32_t *p; uint32_t get(int e) const { return p[e]; } void set(int e, uint32_t v) { p[e] = v; } };
static void plain_adjust(uint32_t *first, long hole, long len, uint32_t value)
{
    long top = hole, sc = hole;
    while (sc < (len - 1) / 2) { sc = 2 * (sc + 1); if ((first[sc] >> 16) < (first[sc - 1] >> 16)) sc--; first[hole] = first[sc]; hole = sc; }
    if ((len & 1) == 0 && sc == (len - 2) / 2) { sc = 2 * (sc + 1); first[hole] = first[sc - 1]; hole = sc - 1; }
    long parent = (hole - 1) / 2;
    while (hole > top && (first[parent] >> 16) < (value >> 16)) { first[hole] = first[parent]; hole = parent; parent = (hole - 1) / 2; }
    first[hole] = value;
}
static void plain_sort(uint32_t *first, long n)
{
    if (n >= 2) for (long parent = (n - 2) / 2;; parent--) { plain_adjust(first, parent, n, first[parent]); if (parent == 0) break; }
    for (long m = n; m > 1;) { m--; uint32_t v = first[m]; first[m] = first[0]; plain_adjust(first, 0, m, v); }
}
int main()
{
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 11); };
    long bad = 0, total = 0, bad_std = 0;
    for (int it = 0; it < 200000; it++) {
        const int n = it < 600 ? it % 600 : (int)(rnd() % 501);
        const int nk = 1 + (int)(rnd() % (it % 3 == 0 ? 4 : it % 3 == 1 ? 40 : 600));      // few distinct keys: ties everywhere
        std::vector<uint32_t> a(n + 2), b, c;
        for (int i = 0; i < n; i++) a[i] = ((rnd() % nk) << 16) | (uint32_t)i;
        b = a; c = a;
        Acc acc{a.data()};
        mc_heap_words_sort(acc, n);
        plain_sort(b.data(), n);
        auto cmp = [](uint32_t x, uint32_t y) { return (x >> 16) < (y >> 16); };
        std::make_heap(c.begin(), c.begin() + n, cmp); std::sort_heap(c.begin(), c.begin() + n, cmp);
        total++;
        if (a != b) bad++;
        if (a != c) bad_std++;
    }
    printf("arrays %ld differ_from_plain %ld differ_from_libstdcxx %ld\n", total, bad, bad_std);
    return bad != 0;
}
