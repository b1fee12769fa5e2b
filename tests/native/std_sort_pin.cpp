// Pins oracle/orb_oracle.c's restatement of libstdc++ std::sort against the real std::sort
// of this container's g++, for ORB-SLAM3's compareNodes (size, then UL.x; not a total
// order, so the permutation of equal keys is implementation-defined and must be restated).
// Test infrastructure.  Exit code 0 = every trial identical.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../../oracle/orb_oracle.h"

static bool compareNodes(const orc_sort_item &e1, const orc_sort_item &e2)
{
    if (e1.size < e2.size) return true;
    else if (e1.size > e2.size) return false;
    else return e1.ulx < e2.ulx;
}

// McIlroy's "A Killer Adversary for Quicksort": builds an input that drives any
// quicksort-family std::sort to its depth limit, so the heapsort fallback is exercised.
static std::vector<int> killer(int n)
{
    std::vector<int> val(n), ptr(n);
    int gas = n - 1, nsolid = 0, candidate = 0;
    for (int i = 0; i < n; i++) { ptr[i] = i; val[i] = gas; }
    auto cmp = [&](int x, int y) {
        if (val[x] == gas && val[y] == gas) {
            if (x == candidate) val[x] = nsolid++; else val[y] = nsolid++;
        }
        if (val[x] == gas) candidate = x; else if (val[y] == gas) candidate = y;
        return val[x] < val[y];
    };
    std::sort(ptr.begin(), ptr.end(), cmp);
    return val;
}

static int check(std::vector<orc_sort_item> v, const char *what)
{
    std::vector<orc_sort_item> a = v, b = v;
    std::sort(a.begin(), a.end(), compareNodes);
    orc_std_sort(b.data(), (int)b.size());
    for (size_t i = 0; i < v.size(); i++)
        if (a[i].id != b[i].id || a[i].size != b[i].size || a[i].ulx != b[i].ulx) {
            std::fprintf(stderr, "MISMATCH (%s) n=%zu at %zu: std id %d vs restated id %d\n", what,
                         v.size(), i, a[i].id, b[i].id);
            return 1;
        }
    return 0;
}

int main()
{
    std::mt19937 rng(12345);
    int bad = 0;
    long trials = 0;
    for (int n = 0; n <= 700 && !bad; n++) {
        for (int rep = 0; rep < 12 && !bad; rep++) {
            std::vector<orc_sort_item> v(n);
            const int size_range = 1 + (int)(rng() % (rep < 4 ? 3 : rep < 8 ? 12 : 200));
            const int x_range = 1 + (int)(rng() % (rep % 2 ? 4 : 40));
            for (int i = 0; i < n; i++) {
                v[i].size = 2 + (int)(rng() % size_range);
                v[i].ulx = 35 * (int)(rng() % x_range);
                v[i].id = i;
            }
            bad |= check(v, "random");
            trials++;
        }
    }
    const int before = orc_std_sort_heap_calls;
    for (int n : {64, 200, 433, 1000, 2171, 5000}) {
        std::vector<int> k = killer(n);
        for (int shift : {0, 1, 2, 3}) { // quantise to create ties that still defeat median-of-3
            std::vector<orc_sort_item> v(n);
            for (int i = 0; i < n; i++) { v[i].size = k[i] >> shift; v[i].ulx = (k[i] * 7) % 5; v[i].id = i; }
            bad |= check(v, "killer");
            trials++;
        }
    }
    std::printf("trials=%ld heap_fallbacks=%d mismatches=%d\n", trials, orc_std_sort_heap_calls - before, bad);
    if (orc_std_sort_heap_calls - before == 0) { std::fprintf(stderr, "heapsort path never taken\n"); return 2; }
    return bad;
}
