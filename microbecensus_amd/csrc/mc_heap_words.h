// mc_heap_words.h - MergeRes' heap sort (libstdc++ make_heap + sort_heap, GCC 4.4: mc_sort_impl.h) over 32-bit words whose KEY IS
// THE UPPER HALF (rank << 16 | index), move for move, written against an accessor so that the same text runs on a lane of
// k_heap_lanes (words transposed in LDS) and on the host (tests/test_emul.py checks it against mc_heapsort and against the C++
// library's own make_heap / sort_heap on arrays full of ties).
//
// (Round 5 tried to read both children AND the four grandchildren of the hole together and take two levels of the walk per trip to the
// LDS - word for word the same result on 200,000 arrays, and k_heap_lanes took 0.70 ms per 1 M reads instead of 0.67: a lane's walk
// is bound by the instructions of its loop and the divergence between the 64 heaps of a wave, not by the trip.)
// A: struct with uint32_t get(int e) const; void set(int e, uint32_t v); elements 0-based.
#ifndef MC_HEAP_WORDS_H
#define MC_HEAP_WORDS_H
#include <stdint.h>
#ifndef MC_HD
#define MC_HD inline
#endif
template <class A>
MC_HD void mc_heap_words_adjust(A &w, int hole, int len, uint32_t value)
{
    const int top = hole, half = (len - 1) / 2;
    int sc = hole;
    while (sc < half) {
        sc = 2 * (sc + 1);
        const uint32_t L = w.get(sc - 1), R = w.get(sc);
        uint32_t pick = R;
        if ((R >> 16) < (L >> 16)) { sc--; pick = L; }
        w.set(hole, pick); hole = sc;
    }
    if ((len & 1) == 0 && sc == (len - 2) / 2) { sc = 2 * (sc + 1); w.set(hole, w.get(sc - 1)); hole = sc - 1; }
    int parent = (hole - 1) / 2;
    while (hole > top) {
        const uint32_t p = w.get(parent);
        if (!((p >> 16) < (value >> 16))) break;
        w.set(hole, p); hole = parent; parent = (hole - 1) / 2;
    }
    w.set(hole, value);
}
template <class A>
MC_HD void mc_heap_words_sort(A &w, int n)
{
    if (n < 2) return;
    for (int parent = (n - 2) / 2;; parent--) { mc_heap_words_adjust(w, parent, n, w.get(parent)); if (parent == 0) break; }
    for (int m = n; m > 1;) { m--; const uint32_t v = w.get(m); w.set(m, w.get(0)); mc_heap_words_adjust(w, 0, m, v); }
}
#endif
