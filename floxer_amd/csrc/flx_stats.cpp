// Search and alignment statistics in the reference's form (include/statistics.hpp:24-172, src/lib/statistics.cpp): one count and
// eighteen fixed-bin histograms with the same names, thresholds ("real_nanopore" / "simulated" scales, statistics.cpp:7-62) and
// the same terminal and TOML renderings (statistics.cpp:64-145, 421-447), so that tooling written for `floxer --stats` reads this
// build's output. The values come from what the batch pipeline holds anyway (flx_pipeline.cpp: seeds, per-seed selection
// counters, the window of every requested DP, the records).
// Deviation: the reference measures wall-clock milliseconds per query inside its worker threads; here a query's search /
// verification time is its chunk's phase time divided by the chunk's reads (the GPU works on whole chunks at a time).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <vector>

#include "flx_stats.hpp"

namespace flx {

namespace {
std::vector<u64> linear_range(u64 num_steps, u64 max) {                               // statistics.cpp:463-469
    std::vector<u64> r;
    for (u64 i = 0; i < num_steps; ++i) r.push_back(i * max / num_steps);
    return r;
}
struct Scales { std::vector<u64> small_values, medium_values, tiny_values, query_length, anchor, kept_anchor_per_seed, edit_distance, time; };
Scales scales_for(bool simulated) {                                                   // statistics.cpp:7-62
    Scales s;
    s.small_values = linear_range(30, 100);
    s.medium_values = linear_range(30, 1000);
    s.tiny_values = {0, 1, 2, 3, 4};
    s.query_length = linear_range(30, simulated ? 10'000 : 150'000);
    s.anchor = linear_range(30, simulated ? 1000 : 30'000);
    s.kept_anchor_per_seed = linear_range(30, 200);
    s.edit_distance = linear_range(30, simulated ? 1000 : 3000);
    s.time = linear_range(30, simulated ? 3000 : 20'000);
    return s;
}
std::string underscored(std::string s) { std::replace(s.begin(), s.end(), ' ', '_'); return s; }
std::string join(std::vector<u64> const& v, const char* sep) {
    std::string out;
    for (size_t i = 0; i < v.size(); ++i) { if (i) out += sep; out += std::to_string(v[i]); }
    return out;
}
std::string two_decimals(double v) { char b[64]; snprintf(b, sizeof(b), "%.2f", v); return b; }
}  // namespace

void StatHistogram::add(u64 value) {                                                  // statistics.cpp:80-94
    ++num_values;
    min = std::min(min, value);
    sum += (double)value;
    max = std::max(max, value);
    for (size_t i = 0; i < thresholds.size(); ++i)
        if (value <= thresholds[i]) { ++data[i]; return; }
    ++data.back();
}
void StatHistogram::merge(StatHistogram const& o) {                                   // statistics.cpp:96-106
    num_values += o.num_values;
    min = std::min(min, o.min);
    sum += o.sum;
    max = std::max(max, o.max);
    for (size_t i = 0; i < data.size(); ++i) data[i] += o.data[i];
}

Stats::Stats(bool simulated_) : simulated(simulated_) {
    Scales const s = scales_for(simulated);
    auto h = [&](std::vector<u64> const& thresholds, const char* name) {
        StatHistogram x;
        x.name = name;
        x.thresholds = thresholds;
        x.data.assign(thresholds.size() + 1, 0);
        histograms.push_back(std::move(x));
    };
    // order and scales of statistics.cpp:223-242
    h(s.query_length, "query lengths");
    h(s.small_values, "seed lengths");
    h(s.tiny_values, "errors per seed");
    h(s.medium_values, "seeds per query");
    h(s.medium_values, "fully excluded seeds per query");
    h(s.anchor, "kept anchors per query");
    h(s.anchor, "excluded raw anchors by soft cap per query");
    h(s.anchor, "excluded raw anchors by erase useless per query");
    h(s.kept_anchor_per_seed, "kept anchors per kept seed");
    h(s.kept_anchor_per_seed, "excluded raw anchors by soft cap per kept seed");
    h(s.kept_anchor_per_seed, "excluded raw anchors by erase useless per kept seed");
    h(s.query_length, "reference span sizes aligned of inner nodes");
    h(s.query_length, "reference span sizes aligned of roots");
    h(s.query_length, "reference span sizes alignment avoided of roots");
    h(s.small_values, "alignments per query");
    h(s.edit_distance, "alignments edit distance");
    h(s.time, "milliseconds spent in search per query");
    h(s.time, "milliseconds spent in verification per query");
}

void Stats::merge(Stats const& o) {                                                   // statistics.cpp:449-459
    completely_excluded_queries += o.completely_excluded_queries;
    for (size_t i = 0; i < histograms.size(); ++i) histograms[i].merge(o.histograms[i]);
}

// statistics.cpp:367-419 for one query: its seeds' selection counters, forward then reverse complement
void Stats::add_search_result(const SeedStatRow* rows, size_t n) {
    u64 fully_excluded = 0, kept = 0, by_soft = 0, by_erase = 0;
    bool all_excluded = true;
    for (size_t i = 0; i < n; ++i) {
        SeedStatRow const& r = rows[i];
        if (r.useful == 0) { ++fully_excluded; continue; }
        all_excluded = false;
        kept += r.useful;
        at(KEPT_ANCHORS_PER_KEPT_SEED).add(r.useful);
        by_soft += r.excluded_soft;
        at(EXCLUDED_SOFT_PER_KEPT_SEED).add(r.excluded_soft);
        u64 const erased = r.raw - r.useful;
        by_erase += erased;
        at(EXCLUDED_ERASE_PER_KEPT_SEED).add(erased);
    }
    at(FULLY_EXCLUDED_SEEDS_PER_QUERY).add(fully_excluded);
    at(KEPT_ANCHORS_PER_QUERY).add(kept);
    at(EXCLUDED_SOFT_PER_QUERY).add(by_soft);
    at(EXCLUDED_ERASE_PER_QUERY).add(by_erase);
    if (all_excluded) ++completely_excluded_queries;
}

std::string Stats::format(bool toml) const {
    std::string out;
    if (toml) {                                                                       // statistics.cpp:70-74, 126-145, 436-447
        out += "completely_excluded_queries = " + std::to_string(completely_excluded_queries) + "\n";
        for (auto const& h : histograms) {
            out += "[" + underscored(h.name) + "]\nnum_values = " + std::to_string(h.num_values) + "\nthresholds = [" + join(h.thresholds, ", ") +
                   "]\noccurrences = [" + join(h.data, ", ") + "]\n";
            if (h.num_values > 0)
                out += "min_value = " + std::to_string(h.min) + "\nmean = " + two_decimals(h.sum / (double)h.num_values) + "\nmax_value = " + std::to_string(h.max) + "\n";
        }
    } else {                                                                          // statistics.cpp:64-68, 108-124, 421-434; one entry per line group
        out += "number of completely excluded queries: " + std::to_string(completely_excluded_queries) + "\n\n";
        for (auto const& h : histograms) {
            out += "histogram for " + h.name + " (total: " + std::to_string(h.num_values) + ")\nthreshold:\t" + join(h.thresholds, "\t") + "\tinf\noccurrences:\t" +
                   join(h.data, "\t");
            if (h.num_values > 0) out += "\nmin = " + std::to_string(h.min) + ", mean = " + two_decimals(h.sum / (double)h.num_values) + ", max = " + std::to_string(h.max);
            out += "\n\n";
        }
    }
    return out;
}

}  // namespace flx

using namespace flx;

struct flx_stats {
    Stats s;
    std::mutex mu;
    explicit flx_stats(bool simulated) : s(simulated) {}
};

namespace flx {
void stats_merge_locked(flx_stats* into, Stats const& local) {
    std::lock_guard<std::mutex> g(into->mu);
    into->s.merge(local);
}
bool stats_simulated(const flx_stats* s) { return s->s.simulated; }
}  // namespace flx

extern "C" int flx_stats_create(const char* input_hint, flx_stats** out) {
    if (!out) { set_error("flx_stats_create: null argument"); return FLX_ERR_INVALID; }
    std::string const hint = input_hint ? input_hint : "";
    if (!hint.empty() && hint != "real_nanopore" && hint != "simulated") { set_error("unknown stats input hint"); return FLX_ERR_INVALID; }   // statistics.cpp:218-220
    *out = new flx_stats(hint == "simulated");
    return FLX_OK;
}
extern "C" void flx_stats_free(flx_stats* s) { delete s; }
extern "C" uint64_t flx_stats_num_queries(const flx_stats* s) { return s ? s->s.histograms[Stats::QUERY_LENGTHS].num_values : 0; }   // statistics.cpp:417-419
extern "C" int flx_stats_merge(flx_stats* into, const flx_stats* other) {
    if (!into || !other || into->s.simulated != other->s.simulated) { set_error("flx_stats_merge: null argument or different scales"); return FLX_ERR_INVALID; }
    stats_merge_locked(into, other->s);
    return FLX_OK;
}
extern "C" int flx_stats_format(const flx_stats* s, int toml, char* buf, uint64_t* len) {
    if (!s || !len) { set_error("flx_stats_format: null argument"); return FLX_ERR_INVALID; }
    std::string text;
    { std::lock_guard<std::mutex> g(const_cast<flx_stats*>(s)->mu); text = s->s.format(toml != 0); }
    uint64_t const cap = *len;
    *len = text.size() + 1;
    if (!buf || cap < text.size() + 1) { set_error("stats text buffer too small"); return FLX_ERR_CAPACITY; }
    memcpy(buf, text.c_str(), text.size() + 1);
    return FLX_OK;
}
