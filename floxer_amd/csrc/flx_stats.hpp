// Statistics accumulator of the batch pipeline (see flx_stats.cpp).
#pragma once

#include <limits>
#include <string>
#include <vector>

#include "flx_internal.hpp"

struct flx_stats;

namespace flx {

struct StatHistogram {
    std::string name;
    std::vector<u64> thresholds, data;
    u64 num_values = 0, min = std::numeric_limits<u64>::max(), max = 0;
    double sum = 0.0;
    void add(u64 value);
    void merge(StatHistogram const& other);
};

struct SeedStatRow { u32 useful, raw, excluded_soft; };

struct Stats {
    enum Id {      // order of statistics.cpp:223-242
        QUERY_LENGTHS, SEED_LENGTHS, ERRORS_PER_SEED, SEEDS_PER_QUERY, FULLY_EXCLUDED_SEEDS_PER_QUERY, KEPT_ANCHORS_PER_QUERY,
        EXCLUDED_SOFT_PER_QUERY, EXCLUDED_ERASE_PER_QUERY, KEPT_ANCHORS_PER_KEPT_SEED, EXCLUDED_SOFT_PER_KEPT_SEED,
        EXCLUDED_ERASE_PER_KEPT_SEED, SPAN_INNER, SPAN_ROOT, SPAN_ROOT_AVOIDED, ALIGNMENTS_PER_QUERY, EDIT_DISTANCE, MS_SEARCH,
        MS_VERIFICATION, N_HISTOGRAMS
    };
    bool simulated;
    u64 completely_excluded_queries = 0;
    std::vector<StatHistogram> histograms;
    explicit Stats(bool simulated);
    StatHistogram& at(Id id) { return histograms[(size_t)id]; }
    void merge(Stats const& other);
    void add_search_result(const SeedStatRow* rows, size_t n);
    std::string format(bool toml) const;
};

void stats_merge_locked(flx_stats* into, Stats const& local);
bool stats_simulated(const flx_stats* s);

}  // namespace flx
