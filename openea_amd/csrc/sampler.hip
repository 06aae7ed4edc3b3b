// sampler.hip -- truncated / uniform negative sampling on the device.
//
// Replaces generate_neg_triples_fast (modules/train/batch.py:89-119): per positive, up to
// max_try rounds; one Bernoulli(0.5) per ROUND picks head or tail corruption for all
// samples still needed (batch.py:99); `need` distinct candidates are drawn without
// replacement (random.sample, batch.py:101,104); true triples are removed through the
// set difference unless it is the last round (batch.py:106-111).
// candidates = neighbor.get(entity, entities_list) (batch.py:96-97).
//
// The python set `all_triples_set` becomes an open-addressing table of packed 64-bit keys
// in HBM (2x load factor); `neighbor` becomes a dense int32 [N, nbr_k] matrix produced by
// oea_topk_inner and never leaves the device.  RNG = Philox4x32-10, so the CPU oracle
// reproduces every draw bit-for-bit (oracle/c/oracle.c:oracle_sample_negatives).
//
// One lane per positive: the work is a short dependent chain of L2-resident probes, the
// batch has thousands of positives, so lanes give enough parallelism without any
// cross-lane traffic.
#include "common.h"

namespace {

__global__ void tripleset_clear(uint64_t *table, uint64_t capacity) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < capacity; i += (uint64_t)gridDim.x * blockDim.x)
        table[i] = OEA_EMPTY_KEY;
}

__global__ void tripleset_insert(const int32_t *__restrict__ triples, int64_t n, uint64_t *table, uint64_t capacity) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t key = oea::pack_triple((uint32_t)triples[3 * i], (uint32_t)triples[3 * i + 1], (uint32_t)triples[3 * i + 2]);
        uint64_t s = oea::mix64(key) & (capacity - 1);
        for (;;) {
            const unsigned long long prev = atomicCAS((unsigned long long *)&table[s], (unsigned long long)OEA_EMPTY_KEY, (unsigned long long)key);
            if (prev == OEA_EMPTY_KEY || prev == key) break;
            s = (s + 1) & (capacity - 1);
        }
    }
}

__device__ __forceinline__ bool contains(const uint64_t *__restrict__ table, uint64_t capacity, uint32_t h, uint32_t r, uint32_t t) {
    const uint64_t key = oea::pack_triple(h, r, t);
    uint64_t s = oea::mix64(key) & (capacity - 1);
    for (;;) {
        const uint64_t cur = table[s];
        if (cur == key) return true;
        if (cur == OEA_EMPTY_KEY) return false;
        s = (s + 1) & (capacity - 1);
    }
}

constexpr int kMaxK = 64;

__global__ __launch_bounds__(64) void sample_negatives_kernel(
    const int32_t *__restrict__ pos, int64_t n_pos, int k, const uint64_t *__restrict__ table, uint64_t capacity,
    const int32_t *__restrict__ entity_list, int n_ent_list, const int32_t *__restrict__ ent_pos,
    const int32_t *__restrict__ nbr, int nbr_k, uint32_t k0, uint32_t k1, uint32_t step, uint32_t pos_offset,
    int max_try, int32_t *__restrict__ out, int32_t *__restrict__ err_flag) {
    // chosen[] lives in LDS, transposed ([slot][lane]) so a wave's accesses are conflict-free
    // and never spill to scratch.
    __shared__ int32_t s_chosen[kMaxK * 64];
    int32_t *chosen = s_chosen + threadIdx.x;
#define CH(q) chosen[(q) * 64]
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pos) return;
    const int32_t h = pos[3 * p], r = pos[3 * p + 1], t = pos[3 * p + 2];
    // neighbor.get(e, entities_list): entities without a neighbour row (e.g. the other KG's
    // entities inside seed-swapped triples, kgs.py:45-50) fall back to the whole entity list
    const bool h_has = nbr && ent_pos[h] >= 0, t_has = nbr && ent_pos[t] >= 0;
    const int32_t *hc = h_has ? nbr + (int64_t)ent_pos[h] * nbr_k : entity_list;
    const int32_t *tc = t_has ? nbr + (int64_t)ent_pos[t] * nbr_k : entity_list;
    const int hn = h_has ? nbr_k : n_ent_list, tn = t_has ? nbr_k : n_ent_list;
    int got = 0;
    const uint32_t c0 = (uint32_t)p + pos_offset;
    for (int tr = 0; tr < max_try && got < k; ++tr) {
        uint4 w = oea::philox4x32_10(c0, step, (uint32_t)tr, 0u, k0, k1);
        const bool corrupt_head = (w.x & 1u) != 0u;
        const int32_t *cand = corrupt_head ? hc : tc;
        const int nc = corrupt_head ? hn : tn;
        const int need = k - got;
        if (need > nc) { *err_flag = 1; return; }   // random.sample would raise ValueError
        uint32_t draw = 1;
        for (int s = 0; s < need; ++s) {
            for (;;) {
                w = oea::philox4x32_10(c0, step, (uint32_t)tr, draw++, k0, k1);
                const int32_t j = (int32_t)__umulhi(w.x, (uint32_t)nc);
                bool dup = false;
                for (int q = 0; q < s; ++q) dup |= (CH(q) == j);
                if (!dup) { CH(s) = j; break; }
            }
        }
        for (int s = 0; s < need; ++s) {
            const int32_t e = cand[CH(s)];
            const int32_t nh = corrupt_head ? e : h, nt = corrupt_head ? t : e;
            if (tr == max_try - 1 || !contains(table, capacity, (uint32_t)nh, (uint32_t)r, (uint32_t)nt)) {
                int32_t *o = out + ((int64_t)p * k + got) * 3;
                o[0] = nh; o[1] = r; o[2] = nt;
                ++got;
            }
        }
    }
#undef CH
}

}  // namespace

extern "C" {

uint64_t oea_tripleset_capacity(int64_t n) {
    uint64_t cap = 16;
    while (cap < 2ull * (uint64_t)(n > 1 ? n : 1)) cap *= 2;
    return cap;
}

int oea_tripleset_build(const int32_t *triples, int64_t n, uint64_t *table, uint64_t capacity, void *stream) {
    OEA_REQUIRE(table && (triples || n == 0), "null pointer");
    OEA_REQUIRE(capacity >= 2ull * (uint64_t)n && (capacity & (capacity - 1)) == 0, "capacity: power of two >= 2n");
    hipStream_t st = oea::as_stream(stream);
    tripleset_clear<<<1024, 256, 0, st>>>(table, capacity);
    if (n > 0) tripleset_insert<<<(unsigned)std::min<int64_t>(oea::ceil_div(n, 256), 4096), 256, 0, st>>>(triples, n, table, capacity);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_sample_negatives(const int32_t *pos, int64_t n_pos, int32_t k, const uint64_t *table,
                         uint64_t capacity, const int32_t *entity_list, int32_t n_ent_list,
                         const int32_t *ent_pos, const int32_t *nbr, int32_t nbr_k, uint64_t seed,
                         uint32_t step, uint32_t pos_offset, int32_t max_try, int32_t *out,
                         int32_t *err_flag, void *stream) {
    OEA_REQUIRE(pos && table && entity_list && out && err_flag, "null pointer");
    OEA_REQUIRE(k >= 1 && k <= kMaxK, "1 <= k <= 64");
    OEA_REQUIRE(max_try >= 1, "max_try >= 1");
    OEA_REQUIRE(nbr == nullptr || (ent_pos != nullptr && nbr_k > 0), "nbr needs ent_pos and nbr_k");
    OEA_REQUIRE((nbr ? nbr_k : n_ent_list) >= k && n_ent_list >= k, "Sample larger than population");
    if (n_pos == 0) return OEA_OK;
    sample_negatives_kernel<<<(unsigned)oea::ceil_div(n_pos, 64), 64, 0, oea::as_stream(stream)>>>(
        pos, n_pos, k, table, capacity, entity_list, n_ent_list, ent_pos, nbr, nbr_k, (uint32_t)seed,
        (uint32_t)(seed >> 32), step, pos_offset, max_try, out, err_flag);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

}  // extern "C"
