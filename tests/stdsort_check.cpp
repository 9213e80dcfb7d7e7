// Compares flx::std_sort_emulated (floxer_amd/csrc/flx_stdsort.hpp) with the std::sort of this toolchain on arrays full of ties,
// with the two comparators the anchor selection uses. Prints "ok <cases>" or the first difference.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#include "../floxer_amd/csrc/flx_stdsort.hpp"

struct Group { uint32_t lb, len, errors; };
struct Anchor { uint64_t pos; uint32_t ref, errors; };

int main() {
    std::mt19937_64 rng(12345);
    long cases = 0, bailed = 0;
    for (int n = 0; n <= 64; ++n)
        for (int rep = 0; rep < 3000; ++rep) {
            int const spread = 1 + (int)(rng() % 6);
            std::vector<Group> g(n), g2;
            for (int i = 0; i < n; ++i) g[i] = Group{(uint32_t)i, 1 + (uint32_t)(rng() % spread), (uint32_t)(rng() % 3)};
            g2 = g;
            auto gl = [](Group const& x, Group const& y) { return x.len != y.len ? x.len < y.len : x.errors < y.errors; };
            std::sort(g.begin(), g.end(), gl);
            bool const ok = flx::std_sort_emulated(g2.data(), n, gl);
            ++cases;
            if (!ok) { ++bailed; continue; }
            for (int i = 0; i < n; ++i)
                if (g[i].lb != g2[i].lb) { std::printf("groups differ: n %d rep %d at %d\n", n, rep, i); return 1; }
            std::vector<Anchor> a(n), a2;
            for (int i = 0; i < n; ++i) a[i] = Anchor{(uint64_t)(rng() % (1 + n / spread)), 0u, (uint32_t)i};
            a2 = a;
            auto al = [](Anchor const& x, Anchor const& y) { return x.pos < y.pos; };
            std::sort(a.begin(), a.end(), al);
            if (!flx::std_sort_emulated(a2.data(), n, al)) { ++bailed; continue; }
            for (int i = 0; i < n; ++i)
                if (a[i].errors != a2[i].errors) { std::printf("anchors differ: n %d rep %d at %d\n", n, rep, i); return 1; }
        }
    std::printf("ok %ld (depth limit reached %ld times)\n", cases, bailed);
    return 0;
}
