// step_plan.hip -- builds the gathered-sum plan of a translational epoch (step_plan.h) from the epoch's positives and the negatives
// drawn ahead for them.  Five launches + three rocPRIM calls into a caller-provided workspace: nothing is allocated, nothing is read
// back, so the whole build can sit on a side stream behind the previous epoch (models/trainer.py:_prefetch_next).
//
// Replaces nothing of the reference by itself: it is the bookkeeping TF's gradient of tf.nn.embedding_lookup does implicitly
// (IndexedSlices -> unsorted_segment_sum over the gathered ids, models/basic_model.py:89-98) done ONCE per epoch instead of per step.
#include <string.h>

#include <rocprim/rocprim.hpp>

#include <algorithm>

#include "common.h"
#include "step_plan.h"

namespace oea {

static size_t al256(size_t x) { return (x + 255) / 256 * 256; }
static int bits_for(uint64_t max_value) { int b = 1; while (b < 64 && (max_value >> b) != 0) ++b; return b; }

size_t step_plan_layout(int64_t n_total, int32_t steps, int64_t max_batch, int64_t n_ent, int32_t ld, void *base, StepPlanView *v) {
    const size_t m = (size_t)std::max<int64_t>(2 * n_total, 1);
    size_t off = 0;
    char *b = static_cast<char *>(base);
    auto take = [&](size_t bytes) { size_t o = off; off += al256(bytes); return b ? b + o : nullptr; };
    StepPlanView w{};
    w.keys_a = (uint64_t *)take(8 * m);
    w.keys_b = (uint64_t *)take(8 * m);
    w.vals_a = (uint32_t *)take(4 * m);
    w.vals_b = (uint32_t *)take(4 * m);
    w.ukeys = (uint64_t *)take(8 * m);
    w.ucount = (uint32_t *)take(4 * (m + 1));
    w.uoff = (uint32_t *)take(4 * (m + 1));
    w.n_unique = (int32_t *)take(4);
    w.step_first = (int32_t *)take(4 * ((size_t)steps + 1));
    w.pflags = (uint32_t *)take(4 * (size_t)std::max<int64_t>(n_total, 1));
    w.recs = (uint4 *)take(16 * m);
    w.inplan = (uint8_t *)take((size_t)std::max<int32_t>(steps, 1) * (size_t)n_ent);
    w.contrib = (float *)take(sizeof(float) * 2 * (size_t)std::max<int64_t>(max_batch, 1) * (size_t)ld);
    // temporary storage of the three device primitives (sizes depend on the element count only)
    size_t t1 = 0, t2 = 0, t3 = 0;
    (void)rocprim::radix_sort_pairs(nullptr, t1, (const uint64_t *)nullptr, (uint64_t *)nullptr, (const uint32_t *)nullptr,
                                    (uint32_t *)nullptr, m, 0, 64, (hipStream_t)0);
    (void)rocprim::run_length_encode(nullptr, t2, (const uint64_t *)nullptr, m, (uint64_t *)nullptr, (uint32_t *)nullptr,
                                     (int32_t *)nullptr, (hipStream_t)0);
    (void)rocprim::exclusive_scan(nullptr, t3, (const uint32_t *)nullptr, (uint32_t *)nullptr, 0u, m + 1, rocprim::plus<uint32_t>(),
                                  (hipStream_t)0);
    w.temp_bytes = std::max(t1, std::max(t2, t3)) + 256;
    w.temp = take(w.temp_bytes);
    w.row_bits = bits_for((uint64_t)std::max<int64_t>(n_ent - 1, 1));
    if (v) *v = w;
    return off;
}

namespace {

// one thread per positive of the epoch: is it inside the rule (every negative a corruption of it, all on one side -- the test
// triple_wave makes), and its two references
__global__ void plan_emit_kernel(const int32_t *__restrict__ pos_all, const int32_t *__restrict__ neg_all, int k,
                                 const int64_t *__restrict__ offsets, int steps, int64_t n_total, int row_bits,
                                 uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_total) return;
    int lo = 0, hi = steps;                                   // largest s with offsets[s] <= p (empty batches repeat an offset)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (offsets[mid] <= p) lo = mid; else hi = mid;
    }
    const uint32_t pl = (uint32_t)(p - offsets[lo]);
    const int h = pos_all[3 * p], r = pos_all[3 * p + 1], t = pos_all[3 * p + 2];
    unsigned bad = 0, xh = 0, xt = 0;
    const int32_t *ng = neg_all + p * k * 3;
    for (int j = 0; j < k; ++j) {
        const unsigned a = (unsigned)(ng[3 * j] ^ h), b = (unsigned)(ng[3 * j + 2] ^ t), c = (unsigned)(ng[3 * j + 1] ^ r);
        bad |= c | min(a, b);
        xh |= a;
        xt |= b;
    }
    if (bad | min(xh, xt)) {                                  // outside the rule: scored as independent triples, atomics + flags
        keys[2 * p] = keys[2 * p + 1] = (uint64_t)steps << row_bits;
        vals[2 * p] = vals[2 * p + 1] = 0u;
        return;
    }
    const bool tails = xh == 0;
    const uint64_t sk = (uint64_t)lo << row_bits;
    keys[2 * p] = sk | (uint64_t)(uint32_t)h;
    vals[2 * p] = 2u * pl + (tails ? 0u : 1u);                        // head row: + A (tail side) / + B (head side)
    keys[2 * p + 1] = sk | (uint64_t)(uint32_t)t;
    vals[2 * p + 1] = 0x80000000u | (2u * pl + (tails ? 1u : 0u));     // tail row: - B (tail side) / - A (head side)
}

__global__ void plan_step_first_kernel(const uint64_t *__restrict__ ukeys, const int32_t *__restrict__ n_unique, int steps, int row_bits,
                                       int32_t *__restrict__ step_first) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > steps) return;
    const uint64_t key = (uint64_t)s << row_bits;
    int lo = 0, hi = *n_unique;                               // first index whose key >= key
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (ukeys[mid] < key) lo = mid + 1; else hi = mid;
    }
    step_first[s] = lo;
}

// one thread per distinct key: its record for the optimiser kernel, and its row's mark in the step's membership map
__global__ void plan_records_kernel(const uint64_t *__restrict__ ukeys, const uint32_t *__restrict__ uoff, const uint32_t *__restrict__ vals,
                                    const int32_t *__restrict__ n_unique, int steps, int row_bits, int64_t n_ent, uint4 *__restrict__ recs,
                                    uint8_t *__restrict__ inplan) {
    const int nu = *n_unique;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nu; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t key = ukeys[i];
        const uint32_t row = (uint32_t)(key & (((uint64_t)1 << row_bits) - 1)), e0 = uoff[i], cnt = uoff[i + 1] - e0;
        recs[i] = make_uint4(row, e0, cnt, vals[e0]);
        const int64_t step = (int64_t)(key >> row_bits);
        if (step < steps && cnt <= kPlanHubEntries) inplan[step * n_ent + row] = 1;
    }
}

// one thread per sorted entry: is its key's run longer than kPlanHubEntries?  (Look at most kPlanHubEntries entries to the left and
// right: a run is a hub exactly when some window of kPlanHubEntries + 1 consecutive entries around the entry carries one key.)  Hub
// entries mark their positive (bit 0: referred to as head, bit 1: as tail) -- the scoring kernel sends those rows' gradient through
// the atomic scratch.  (A wave per distinct key walked 1.3 M keys to find a few thousand hubs: 58 us; this: one pass over the entries.)
__global__ void plan_hub_kernel(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals, int64_t m,
                                const int64_t *__restrict__ offsets, int steps, int row_bits, uint32_t *__restrict__ pflags) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m) return;
    const uint64_t key = keys[e];
    const int64_t step = (int64_t)(key >> row_bits);
    if (step >= steps) return;                                    // the sentinel run of the positives outside the rule
    int left = 0, right = 0;
    while (left < (int)kPlanHubEntries && e - left - 1 >= 0 && keys[e - left - 1] == key) ++left;
    while (left + right < (int)kPlanHubEntries && e + right + 1 < m && keys[e + right + 1] == key) ++right;
    if (left + right < (int)kPlanHubEntries) return;              // run length = left + right + 1 (capped) <= kPlanHubEntries
    const uint32_t v = vals[e];
    atomicOr(&pflags[offsets[step] + ((v & 0x7fffffffu) >> 1)], (v >> 31) ? 2u : 1u);
}

// ---- an epoch's shuffle + batch layout in one place (basic_model.py:234-235: random.shuffle of both KGs' triple lists; batch.py:17-22:
// batch s = KG1's slice s followed by KG2's slice s) ---------------------------------------------------------------------------------------
// The permutation of a list of n triples is a keyed BIJECTION evaluated per element, no sort: a 6-round Feistel network on the
// ceil(log2 n)-bit index (round function: a 32-bit mix of the right half with a round key from Philox4x32-10 of (seed, epoch, list)),
// cycle-walked until the image falls below n (the domain is < 2 n: < 2 evaluations on average).  A Feistel network is a permutation of
// its power-of-two domain for ANY round function, and cycle walking restricts it to [0, n) -- every triple exactly once; which
// permutation comes out is decided by the 6 x 32 key bits.  dall[j] = triples[perm(slot[j])]: ONE launch (a radix sort of Philox keys
// + a gather took 0.21 ms of side-stream time per epoch at the 100K shape, torch.randperm x 2 + two index gathers 0.66 ms).  A fresh
// permutation of the FIXED list every epoch has the distribution of shuffling the previous epoch's order.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t feistel_perm(uint32_t i, uint32_t n, int lbits, int rbits, const uint32_t (&key)[6]) {
    const uint32_t lmask = (1u << lbits) - 1u, rmask = (1u << rbits) - 1u;
    uint32_t x = i;
    do {
        uint32_t l = x >> rbits, r = x & rmask;
#pragma unroll
        for (int q = 0; q < 6; q += 2) {   // alternating rounds: each XORs one half with a function of the other -- invertible whatever mix32 is
            l = (l ^ mix32(r ^ key[q])) & lmask;
            r = (r ^ mix32(l ^ key[q + 1])) & rmask;
        }
        x = (l << rbits) | r;
    } while (x >= n);
    return x;
}
__global__ void layout_perm_kernel(const int32_t *__restrict__ triples, int64_t n1, int64_t n2, const int64_t *__restrict__ slot, int64_t n_slots,
                                   uint32_t k0, uint32_t k1, uint32_t epoch, int32_t *__restrict__ dall) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_slots) return;
    const int64_t sj = slot[j];
    const bool second = sj >= n1;
    const uint32_t n = (uint32_t)(second ? n2 : n1), i = (uint32_t)(second ? sj - n1 : sj);
    int bits = 1;
    while (bits < 32 && (1ull << bits) < (unsigned long long)n) ++bits;
    if (bits < 2) bits = 2;
    const int rbits = bits / 2, lbits = bits - rbits;
    const uint4 wa = philox4x32_10(epoch, second ? 1u : 0u, 0x5eedu, 0u, k0, k1), wb = philox4x32_10(epoch, second ? 1u : 0u, 0x5eedu, 1u, k0, k1);
    const uint32_t key[6] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y};
    const int64_t src = (int64_t)feistel_perm(i, n, lbits, rbits, key) + (second ? n1 : 0);
    dall[3 * j] = triples[3 * src]; dall[3 * j + 1] = triples[3 * src + 1]; dall[3 * j + 2] = triples[3 * src + 2];
}

}  // namespace
}  // namespace oea

extern "C" {

size_t oea_step_plan_bytes(int64_t n_total, int32_t steps, int64_t max_batch, int64_t n_ent, int32_t ld) {
    if (n_total < 0 || steps < 0 || max_batch < 0 || n_ent <= 0 || ld <= 0) return 0;
    return oea::step_plan_layout(n_total, steps, max_batch, n_ent, ld, nullptr, nullptr);
}

/* byte offsets of the arrays a built plan consists of (tests; the optimiser kernel takes them from step_plan_layout):
 * out[0] sorted entry values (uint32 [2 n_total]: sign << 31 | slot), [1] distinct keys (uint64: step << row_bits | row), [2] first
 * entry of every distinct key (uint32 [.. + 1]), [3] number of distinct keys (int32), [4] first distinct key of every step (int32
 * [steps + 1]), [5] contribution rows (float [2 max_batch, ld]), [6] row_bits, [7] total bytes, [8] per-positive hub bits (uint32 [n_total]) */
int oea_step_plan_offsets(int64_t n_total, int32_t steps, int64_t max_batch, int64_t n_ent, int32_t ld, int64_t *out) {
    OEA_REQUIRE(out && n_total >= 0 && steps >= 0 && n_ent > 0 && ld > 0, "arguments");
    oea::StepPlanView v;
    char *base = reinterpret_cast<char *>(4096);                 // fake base: only differences matter
    const size_t total = oea::step_plan_layout(n_total, steps, max_batch, n_ent, ld, base, &v);
    out[0] = (char *)v.vals_b - base; out[1] = (char *)v.ukeys - base; out[2] = (char *)v.uoff - base;
    out[3] = (char *)v.n_unique - base; out[4] = (char *)v.step_first - base; out[5] = (char *)v.contrib - base;
    out[6] = v.row_bits; out[7] = (int64_t)total; out[8] = (char *)v.pflags - base;
    return OEA_OK;
}

size_t oea_epoch_layout_bytes(int64_t n) {
    return n < 0 ? 0 : 256;   // the keyed permutation needs no scratch; the argument stays in the interface for callers that size it
}

int oea_epoch_layout(const int32_t *triples, int64_t n1, int64_t n2, const int64_t *slot, int64_t n_slots, uint64_t seed, uint32_t epoch,
                     int32_t *dall, void *workspace, size_t ws_bytes, void *stream) {
    OEA_REQUIRE(triples && (slot || n_slots == 0) && (dall || n_slots == 0), "null pointer");
    OEA_REQUIRE(n1 >= 0 && n2 >= 0 && n_slots >= 0 && n1 < ((int64_t)1 << 32) && n2 < ((int64_t)1 << 32), "sizes");
    (void)workspace; (void)ws_bytes;
    if (n1 + n2 == 0 || n_slots == 0) return OEA_OK;
    hipStream_t st = oea::as_stream(stream);
    oea::layout_perm_kernel<<<(unsigned)oea::ceil_div(n_slots, 256), 256, 0, st>>>(triples, n1, n2, slot, n_slots, (uint32_t)seed,
                                                                                   (uint32_t)(seed >> 32), epoch, dall);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_step_plan_build(const int32_t *pos_all, const int32_t *neg_all, int32_t k, const int64_t *offsets_dev, int64_t n_total,
                        int32_t steps, int64_t max_batch, int64_t n_ent, int32_t ld, void *plan, size_t plan_bytes, void *stream) {
    OEA_REQUIRE(pos_all && offsets_dev && plan && (neg_all || k == 0), "null pointer");
    OEA_REQUIRE(n_total >= 0 && steps >= 1 && k >= 0 && n_ent > 0 && ld > 0, "sizes");
    OEA_REQUIRE(2 * n_total < ((int64_t)1 << 31) && 2 * max_batch < ((int64_t)1 << 31), "entries and slots are 31-bit");
    oea::StepPlanView v;
    const size_t need = oea::step_plan_layout(n_total, steps, max_batch, n_ent, ld, plan, &v);
    OEA_REQUIRE(plan_bytes >= need, "plan workspace smaller than oea_step_plan_bytes");
    hipStream_t st = oea::as_stream(stream);
    const size_t m = (size_t)(2 * n_total);
    if (m == 0) {
        OEA_CHECK_HIP(hipMemsetAsync(v.n_unique, 0, sizeof(int32_t), st));
        OEA_CHECK_HIP(hipMemsetAsync(v.step_first, 0, sizeof(int32_t) * ((size_t)steps + 1), st));
        return OEA_OK;
    }
    oea::plan_emit_kernel<<<(unsigned)oea::ceil_div(n_total, 256), 256, 0, st>>>(pos_all, neg_all, k, offsets_dev, steps, n_total, v.row_bits,
                                                                                 v.keys_a, v.vals_a);
    const int end_bit = std::min(64, v.row_bits + oea::bits_for((uint64_t)steps));
    size_t tb = v.temp_bytes;
    OEA_CHECK_HIP(rocprim::radix_sort_pairs(v.temp, tb, (const uint64_t *)v.keys_a, v.keys_b, (const uint32_t *)v.vals_a, v.vals_b, m, 0,
                                            (unsigned)end_bit, st));
    OEA_CHECK_HIP(hipMemsetAsync(v.ucount, 0, sizeof(uint32_t) * (m + 1), st));
    tb = v.temp_bytes;
    OEA_CHECK_HIP(rocprim::run_length_encode(v.temp, tb, (const uint64_t *)v.keys_b, m, v.ukeys, v.ucount, v.n_unique, st));
    tb = v.temp_bytes;
    OEA_CHECK_HIP(rocprim::exclusive_scan(v.temp, tb, (const uint32_t *)v.ucount, v.uoff, 0u, m + 1, rocprim::plus<uint32_t>(), st));
    oea::plan_step_first_kernel<<<(unsigned)oea::ceil_div(steps + 1, 256), 256, 0, st>>>(v.ukeys, v.n_unique, steps, v.row_bits, v.step_first);
    OEA_CHECK_HIP(hipMemsetAsync(v.pflags, 0, sizeof(uint32_t) * (size_t)n_total, st));
    OEA_CHECK_HIP(hipMemsetAsync(v.inplan, 0, (size_t)steps * (size_t)n_ent, st));
    oea::plan_records_kernel<<<(unsigned)std::min<int64_t>(oea::ceil_div((int64_t)m, 256), 4096), 256, 0, st>>>(
        v.ukeys, v.uoff, v.vals_b, v.n_unique, steps, v.row_bits, n_ent, v.recs, v.inplan);
    oea::plan_hub_kernel<<<(unsigned)oea::ceil_div((int64_t)m, 256), 256, 0, st>>>(v.keys_b, v.vals_b, (int64_t)m, offsets_dev, steps, v.row_bits,
                                                                                   v.pflags);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

}  // extern "C"
