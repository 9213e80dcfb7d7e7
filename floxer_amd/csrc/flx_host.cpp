// Host-side integer/double arithmetic of the path that defines seeds and windows: must be bit-identical to the reference,
// so it stays on the host in IEEE double. math.hpp, input.cpp, pex.cpp, search-scheme expansion.
#include <cmath>
#include <algorithm>
#include <cstdlib>
#include <new>
#include <vector>
#include <limits>
#include <map>
#include <mutex>

#include "flx_internal.hpp"

namespace flx {

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
const char* last_error_cstr() { return g_last_error.c_str(); }

// ---------------------------------------------------------------- host block pool (flx_internal.hpp)
namespace {
constexpr size_t POOL_MIN = 256 << 10;
size_t pool_class(size_t bytes) {           // next multiple of an eighth of the enclosing power of two (at most 12.5 % over)
    size_t p2 = POOL_MIN;
    while (p2 < bytes) p2 <<= 1;
    size_t const step = p2 / 16;
    return (bytes + step - 1) / step * step;
}
struct BlockPool {
    std::mutex mu;
    std::map<size_t, std::vector<void*>> free_blocks;
    size_t kept = 0, cap;
    BlockPool() {
        const char* env = getenv("FLX_HOST_POOL_MB");
        cap = (env ? strtoull(env, nullptr, 10) : 16384) << 20;
    }
    ~BlockPool() { for (auto& kv : free_blocks) for (void* p : kv.second) free(p); }
};
BlockPool& block_pool() { static BlockPool* p = new BlockPool(); return *p; }    // never destroyed: lists may outlive static teardown
}  // namespace

std::atomic<void (*)(void* p, size_t bytes, int pin)> host_pool_pin_hook{nullptr};

void* host_pool_get(size_t bytes) {
    if (bytes < POOL_MIN) { void* p = malloc(bytes ? bytes : 1); if (!p) throw std::bad_alloc(); return p; }
    size_t const cls = pool_class(bytes);
    BlockPool& bp = block_pool();
    {
        std::lock_guard<std::mutex> g(bp.mu);
        auto it = bp.free_blocks.find(cls);
        if (it != bp.free_blocks.end() && !it->second.empty()) {
            void* p = it->second.back();
            it->second.pop_back();
            bp.kept -= cls;
            return p;
        }
    }
    void* p = malloc(cls);
    if (!p) throw std::bad_alloc();
    if (auto const hook = host_pool_pin_hook.load(std::memory_order_acquire)) hook(p, cls, 1);
    return p;
}
void host_pool_put(void* p, size_t bytes) {
    if (!p) return;
    if (bytes < POOL_MIN) { free(p); return; }
    size_t const cls = pool_class(bytes);
    BlockPool& bp = block_pool();
    {
        std::lock_guard<std::mutex> g(bp.mu);
        if (bp.kept + cls <= bp.cap) { bp.free_blocks[cls].push_back(p); bp.kept += cls; return; }
    }
    if (auto const hook = host_pool_pin_hook.load(std::memory_order_acquire)) hook(p, cls, 0);
    free(p);
}

// ---------------------------------------------------------------- math.hpp:10-27
u64 ceil_div(u64 a, u64 b) { return (a % b) ? a / b + 1 : a / b; }
u64 fp_aware_ceil(double v) {
    static constexpr double epsilon = 0.000000001;
    return (u64)(std::ceil(v - epsilon) + epsilon);
}
int32_t saturate_i32(u64 v) {
    return v > (u64)std::numeric_limits<int32_t>::max() ? std::numeric_limits<int32_t>::max() : (int32_t)v;
}

// ---------------------------------------------------------------- input.cpp:165-176 (ivs::d_dna5 ranks)
u8 char_to_rank(char c) {
    switch (c) {
        case '$': return 0;
        case 'A': case 'a': return 1;
        case 'C': case 'c': return 2;
        case 'G': case 'g': return 3;
        case 'T': case 't': case 'U': case 'u': return 4;
        default: return 5;
    }
}
char rank_to_char(u8 r) { return r < 6 ? "$ACGTN"[r] : 'N'; }
void reverse_complement(const u8* in, u64 n, u8* out) {
    static const u8 comp[8] = {0, 4, 3, 2, 1, 5, 5, 5};
    for (u64 i = 0; i < n; ++i) out[i] = comp[in[n - 1 - i] & 7];
}

// ---------------------------------------------------------------- pex.cpp:84-256, built iteratively
PexTree build_pex_tree(u64 len, u64 k, u64 s, bool bottom_up) {
    PexTree t;
    if (!bottom_up) {
        // recursive strategy (pex.cpp:110-156) with an explicit work list; children are visited left before right so that
        // inner nodes come out in pre-order and leaves left to right, exactly as the recursion emits them.
        u64 const no_error_seed_length = len / (k + 1);
        struct Item { u64 from1, to1, errors; u32 parent; };
        std::vector<Item> work{{1, len, k, FLX_NULL_ID}};
        while (!work.empty()) {
            Item const it = work.back();
            work.pop_back();
            flx_pex_node node{it.parent, (u32)(it.from1 - 1), (u32)(it.to1 - 1), (u32)it.errors};
            if (it.errors <= s) { t.leaves.push_back(node); continue; }
            u32 const id = (u32)t.inner.size();
            t.inner.push_back(node);
            u64 const left_leaves = ceil_div(it.errors + 1, 2);
            u64 const split = it.from1 + left_leaves * no_error_seed_length;
            u64 const e_left = (left_leaves * it.errors) / (it.errors + 1);
            u64 const e_right = ((it.errors + 1 - left_leaves) * it.errors) / (it.errors + 1);
            work.push_back(Item{split, it.to1, e_right, id});      // popped second
            work.push_back(Item{it.from1, split - 1, e_left, id}); // popped first
        }
        return t;
    }
    // bottom-up strategy (pex.cpp:158-256)
    u64 const num_leaves = ceil_div(k + 1, s + 1);
    if (num_leaves == 1) { t.leaves.push_back(flx_pex_node{FLX_NULL_ID, 0, (u32)(len - 1), (u32)k}); return t; }
    u64 const base = len / num_leaves, rem = len % num_leaves;
    u64 start = 0;
    for (u64 i = 0; i < num_leaves; ++i) {
        u64 const l = base + (i < rem ? 1 : 0);
        t.leaves.push_back(flx_pex_node{0, (u32)start, (u32)(start + l - 1), (u32)s});
        start += l;
    }
    t.inner.reserve(num_leaves);
    t.inner.push_back(flx_pex_node{});
    auto merge = [](flx_pex_node* c, size_t count, u32 parent_id) {
        u32 err = 0;
        for (size_t i = 0; i < count; ++i) { c[i].parent_id = parent_id; err += c[i].num_errors; }
        return flx_pex_node{0, c[0].from, c[count - 1].to, (u32)(err + count - 1)};
    };
    flx_pex_node* level = t.leaves.data();
    size_t level_size = t.leaves.size();
    while (level_size > 3) {
        for (size_t i = 0; i + 1 < level_size; i += 2) {
            size_t const take = (level_size - i == 3) ? 3 : 2;
            flx_pex_node parent = merge(level + i, take, (u32)t.inner.size());
            t.inner.push_back(parent);
            if (take == 3) break;
        }
        level_size /= 2;
        level = t.inner.data() + (t.inner.size() - level_size);
    }
    t.inner[0] = merge(level, level_size, 0);
    t.inner[0].parent_id = FLX_NULL_ID;
    return t;
}

// ---------------------------------------------------------------- search schemes (search.cpp:328-350)
const std::vector<SearchDef>& optimum_scheme(u32 k) {
    static const std::vector<SearchDef> s0{{{0}, {0}, {0}}};
    static const std::vector<SearchDef> s1{{{0, 1}, {0, 0}, {0, 1}}, {{1, 0}, {0, 1}, {0, 1}}};
    static const std::vector<SearchDef> s2{{{0, 1, 2, 3}, {0, 0, 1, 1}, {0, 0, 2, 2}},
                                           {{2, 1, 0, 3}, {0, 0, 0, 0}, {0, 1, 1, 2}},
                                           {{3, 2, 1, 0}, {0, 0, 0, 2}, {0, 1, 2, 2}}};
    static const std::vector<SearchDef> s3{{{0, 1, 2, 3, 4}, {0, 0, 0, 0, 0}, {0, 0, 3, 3, 3}},
                                           {{2, 1, 0, 3, 4}, {0, 0, 1, 1, 1}, {0, 1, 1, 2, 3}},
                                           {{3, 2, 1, 0, 4}, {0, 0, 0, 2, 2}, {0, 1, 2, 2, 3}},
                                           {{4, 3, 2, 1, 0}, {0, 0, 0, 0, 3}, {0, 2, 2, 3, 3}}};
    static const std::vector<SearchDef> none;
    switch (k) { case 0: return s0; case 1: return s1; case 2: return s2; case 3: return s3; default: return none; }
}

std::vector<u64> expanded_scheme(u32 k, u32 len) {
    auto const& scheme = optimum_scheme(k);
    std::vector<u64> out;
    if (scheme.empty()) return out;
    u32 const P = (u32)scheme[0].pi.size();
    if (len < P) return out;
    // part p (by position) covers counts[p] characters starting at starts[p]
    std::vector<u32> counts(P, len / P), starts(P, 0);
    for (u32 p = 0; p < len % P; ++p) counts[p]++;
    for (u32 p = 1; p < P; ++p) starts[p] = starts[p - 1] + counts[p - 1];
    out.reserve((size_t)scheme.size() * len);
    for (auto const& s : scheme) {
        size_t const first = out.size();
        for (u32 i = 0; i < P; ++i) {
            u32 const part = s.pi[i];
            bool const right = i == 0 || s.pi[i - 1] < s.pi[i];
            u32 const lower_before_end = i > 0 ? s.l[i - 1] : 0;
            for (u32 j = 0; j < counts[part]; ++j) {
                u32 const pos = right ? starts[part] + j : starts[part] + counts[part] - 1 - j;
                bool const last = j + 1 == counts[part];
                bool const exact_prefix = i == 0 && s.u[0] == 0 && s.l[0] == 0;       // first part of every optimum search
                out.push_back(sch_pack(pos, last ? s.l[i] : lower_before_end, s.u[i], right, exact_prefix));
            }
        }
        // high word (flx_fm_core.hpp): end of the run of entries that share this one's upper bound and direction | lowest seed
        // position among the entries before this one (the walk's string is seed[lo, lo + x) at entry x while it has no indel)
        u64* const e = out.data() + first;
        u32 run_end = len, lo = (u32)e[0] & SCH_POS_MASK;
        std::vector<u32> lo_before(len);
        for (u32 x = 0; x < len; ++x) { lo_before[x] = lo; lo = std::min(lo, (u32)e[x] & SCH_POS_MASK); }
        for (u32 x = len; x-- > 0;) {
            if (x + 1 < len && ((((u32)e[x] ^ (u32)e[x + 1]) >> 23) & 0xFu) != 0u) run_end = x + 1;      // upper bound (3 bits) or direction differs
            e[x] |= ((u64)(run_end & 0x7FFFu) << 32) | ((u64)(lo_before[x] & 0x3FFFu) << 47);
        }
    }
    return out;
}

}  // namespace flx
