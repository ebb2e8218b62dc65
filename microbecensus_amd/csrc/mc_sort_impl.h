// mc_sort_impl.h - libstdc++ (GCC 4.4) std::sort / heap sort, move for move (see mc_finish.h).  Included twice by
// mc_finish.h: once as out-of-line functions (MC_SORT_ATTR = noinline), once force-inlined (suffix _inl) for the kernel
// that sorts items held in LDS - inlining is what lets the compiler address the array as LDS.
// No include guard on purpose.  MC_SORT_FN(x) names the functions, MC_SORT_ATTR is their attribute.
template <class E>
MC_SORT_ATTR void MC_SORT_FN(mc_adjust_heap)(E *first, long hole, long len, E value, int key)
{
    long top = hole, sc = hole;
    while (sc < (len - 1) / 2) {
        sc = 2 * (sc + 1);
        if (mc_hless(first[sc], first[sc - 1], key)) sc--;
        first[hole] = first[sc]; hole = sc;
    }
    if ((len & 1) == 0 && sc == (len - 2) / 2) { sc = 2 * (sc + 1); first[hole] = first[sc - 1]; hole = sc - 1; }
    long parent = (hole - 1) / 2;
    while (hole > top && mc_hless(first[parent], value, key)) { first[hole] = first[parent]; hole = parent; parent = (hole - 1) / 2; }
    first[hole] = value;
}
template <class E>
MC_SORT_ATTR void MC_SORT_FN(mc_heapsort)(E *first, long n, int key)
{
    if (n >= 2) for (long parent = (n - 2) / 2;; parent--) { MC_SORT_FN(mc_adjust_heap)(first, parent, n, first[parent], key); if (parent == 0) break; }
    for (long m = n; m > 1;) { m--; E v = first[m]; first[m] = first[0]; MC_SORT_FN(mc_adjust_heap)(first, 0, m, v, key); }
}
template <class E>
MC_SORT_ATTR void MC_SORT_FN(mc_unguarded_insert)(E *last, E val, int key)
{
    E *next = last - 1;
    while (mc_hless(val, *next, key)) { *last = *next; last = next; --next; }
    *last = val;
}
template <class E>
MC_SORT_ATTR void MC_SORT_FN(mc_insertion_sort)(E *first, E *last, int key)
{
    if (first == last) return;
    for (E *i = first + 1; i != last; ++i) {
        E val = *i;
        if (mc_hless(val, *first, key)) { for (E *p = i; p != first; --p) *p = *(p - 1); *first = val; }
        else MC_SORT_FN(mc_unguarded_insert)(i, val, key);
    }
}
template <class E>
MC_SORT_ATTR void MC_SORT_FN(mc_std_sort)(E *first, long n, int key)
{
    if (n <= 0) return;
    long lg = 0;
    for (long t = n; t > 1; t >>= 1) lg++;
    // explicit stack of (first, last, depth): the recursion of __introsort_loop goes into the right part
    long sf[64], sl[64], sd[64];
    int sp = 0;
    sf[0] = 0; sl[0] = n; sd[0] = 2 * lg; sp = 1;
    while (sp > 0) {
        sp--;
        long f = sf[sp], l = sl[sp], depth = sd[sp];
        while (l - f > 16) {
            if (depth == 0) { MC_SORT_FN(mc_heapsort)(first + f, l - f, key); break; }
            --depth;
            const E &a = first[f], &b = first[f + (l - f) / 2], &c = first[l - 1];
            E pivot;
            if (mc_hless(a, b, key)) { if (mc_hless(b, c, key)) pivot = b; else if (mc_hless(a, c, key)) pivot = c; else pivot = a; }
            else if (mc_hless(a, c, key)) pivot = a;
            else if (mc_hless(b, c, key)) pivot = c;
            else pivot = b;
            long lo = f, hi = l;
            for (;;) {
                while (mc_hless(first[lo], pivot, key)) ++lo;
                --hi;
                while (mc_hless(pivot, first[hi], key)) --hi;
                if (!(lo < hi)) break;
                E t = first[lo]; first[lo] = first[hi]; first[hi] = t;
                ++lo;
            }
            if (sp < 64) { sf[sp] = lo; sl[sp] = l; sd[sp] = depth; sp++; }
            l = lo;
        }
    }
    if (n > 16) { MC_SORT_FN(mc_insertion_sort)(first, first + 16, key); for (E *i = first + 16; i != first + n; ++i) MC_SORT_FN(mc_unguarded_insert)(i, *i, key); }
    else MC_SORT_FN(mc_insertion_sort)(first, first + n, key);
}
