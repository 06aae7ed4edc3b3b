// match.hip -- host-side one-to-one selection for the bootstrapping step (no device code).
//
// Stands in for mwgm_graph_tool (modules/bootstrapping/alignment_finder.py:83-112: graph_tool's
// max_cardinality_matching(heuristic=True, weight=..., minimize=False), a linear-time heuristic maximal matching):
// edges taken by descending weight (ties: smaller left id, then smaller right id) while both endpoints are free --
// the classic 1/2-approximation.  Native because the candidate list reaches several hundred thousand edges at the
// 100K datasets and a python loop over them costs more than the training epochs between two bootstrapping rounds.
#include "common.h"

#include <algorithm>
#include <numeric>
#include <vector>

extern "C" {

int oea_greedy_matching(const int32_t *left, const int32_t *right, const float *weight, int64_t n_edges,
                        uint8_t *selected) {
    OEA_REQUIRE(n_edges >= 0 && (n_edges == 0 || (left && right && weight && selected)), "null pointer");
    int32_t max_l = -1, max_r = -1;
    for (int64_t e = 0; e < n_edges; ++e) {
        OEA_REQUIRE(left[e] >= 0 && right[e] >= 0, "negative node id");
        max_l = std::max(max_l, left[e]);
        max_r = std::max(max_r, right[e]);
    }
    std::vector<int64_t> order((size_t)n_edges);
    std::iota(order.begin(), order.end(), (int64_t)0);
    std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
        if (weight[a] != weight[b]) return weight[a] > weight[b];
        if (left[a] != left[b]) return left[a] < left[b];
        return right[a] < right[b];
    });
    std::vector<uint8_t> used_l((size_t)max_l + 1, 0), used_r((size_t)max_r + 1, 0);
    for (int64_t e = 0; e < n_edges; ++e) selected[e] = 0;
    for (int64_t e : order) {
        if (!used_l[left[e]] && !used_r[right[e]]) {
            used_l[left[e]] = used_r[right[e]] = 1;
            selected[e] = 1;
        }
    }
    return OEA_OK;
}

}  // extern "C"
