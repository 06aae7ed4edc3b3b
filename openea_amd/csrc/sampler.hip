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
// One 16-lane (k <= 16) or 64-lane group per positive, one lane per sampled slot: candidate
// reads and membership probes of a try go out in parallel (the one-lane-per-positive version
// was a 34 us dependent chain for 2,500 positives -- profiles/r01a_bench_kernel_stats.txt).
#include <stdlib.h>

#include <algorithm>

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

__device__ __forceinline__ uint64_t filter_bit(uint64_t key, uint64_t bits) { return oea::mix64(key ^ 0x9E3779B97F4A7C15ull) & (bits - 1); }

__global__ void tripleset_filter_insert(const int32_t *__restrict__ triples, int64_t n, uint32_t *__restrict__ filter, uint64_t bits) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t b = filter_bit(oea::pack_triple((uint32_t)triples[3 * i], (uint32_t)triples[3 * i + 1], (uint32_t)triples[3 * i + 2]), bits);
        atomicOr(&filter[b >> 5], 1u << (b & 31));
    }
}

// filter (may be NULL): a clear bit = certainly absent, no probe of the key table (include/openea_hip.h)
__device__ __forceinline__ bool contains(const uint64_t *__restrict__ table, uint64_t capacity, const uint32_t *__restrict__ filter,
                                         uint64_t filter_bits, uint32_t h, uint32_t r, uint32_t t) {
    const uint64_t key = oea::pack_triple(h, r, t);
    if (filter) {
        const uint64_t b = filter_bit(key, filter_bits);
        if (!((filter[b >> 5] >> (b & 31)) & 1u)) return false;
    }
    uint64_t s = oea::mix64(key) & (capacity - 1);
    for (;;) {
        const uint64_t cur = table[s];
        if (cur == key) return true;
        if (cur == OEA_EMPTY_KEY) return false;
        s = (s + 1) & (capacity - 1);
    }
}

constexpr int kMaxK = 64;

// G lanes per positive (G = 16 for k <= 16, else 64): slot s of a try lives in lane s, so the
// k candidate reads and the k membership probes of a try are issued in parallel instead of
// as one lane's dependent chain.  Distinctness is resolved in synchronous rounds (see
// oracle_sample_negatives): a slot redraws while an earlier slot holds the same index.
template <int G>
__global__ __launch_bounds__(256) void sample_negatives_kernel(
    const int32_t *__restrict__ pos, int64_t n_pos, int64_t n_split, int k, oea_sampler_side side0,
    oea_sampler_side side1, uint32_t k0, uint32_t k1, uint32_t step, uint32_t pos_offset,
    int max_try, int32_t *__restrict__ out, int32_t *__restrict__ err_flag,
    const int64_t *__restrict__ seg_off, const int64_t *__restrict__ seg_split, int n_seg, const int32_t *__restrict__ replay) {
    const int lane = threadIdx.x % G;
    const uint32_t step_in = step;
    const int64_t n_split_in = n_split;
    // grid-stride over the positives (the launch may be capped: OEA_SAMPLER_CAP, experiments)
    for (int64_t p = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G; p < n_pos; p += (int64_t)gridDim.x * blockDim.x / G) {
    step = step_in;
    n_split = n_split_in;
    // Epoch mode (seg_off != NULL): `pos` holds every batch of the epoch back to back; the batch of
    // row p is found by bisection and the Philox counter is (row within batch, step + batch), i.e.
    // exactly the stream of a per-batch call -- one launch samples the whole epoch.
    int64_t pl = p;                               // row index inside its batch
    if (seg_off) {
        // the wave's first row is looked up on the scalar unit (uniform addresses: s_load, no vector-memory round trips in front of
        // everything else this thread does); a lane whose row lies past that batch's end -- a wave straddling a boundary -- bisects alone
        const int64_t p0 = ((int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(p >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)p);
        int lo = 0, hi = n_seg;                   // invariant: seg_off[lo] <= p0 < seg_off[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (seg_off[mid] <= p0) lo = mid; else hi = mid;
        }
        int64_t first = seg_off[lo], split = seg_split[lo];           // scalar loads
        int seg = lo;
        if (p >= seg_off[lo + 1]) {
            int l2 = lo;
            hi = n_seg;
            while (hi - l2 > 1) {
                const int mid = (l2 + hi) >> 1;
                if (seg_off[mid] <= p) l2 = mid; else hi = mid;
            }
            seg = l2; first = seg_off[l2]; split = seg_split[l2];
        }
        pl = p - first;
        n_split = split;
        step += (uint32_t)seg;
    }
    // positives [0, n_split) belong to KG1, the rest to KG2 (pos_batch1 + pos_batch2, batch.py:45)
    const oea_sampler_side &sd = pl < n_split ? side0 : side1;
    const uint64_t *__restrict__ table = sd.table;
    const uint64_t capacity = sd.capacity;
    const int32_t *__restrict__ entity_list = sd.entity_list;
    const int32_t *__restrict__ ent_pos = sd.ent_pos;
    const int32_t *__restrict__ nbr = sd.nbr;
    const uint32_t *__restrict__ filter = sd.filter;
    const uint64_t filter_bits = sd.filter_bits;
    const int n_ent_list = sd.n_ent_list, nbr_k = sd.nbr_k;
    const int32_t h = pos[3 * p], r = pos[3 * p + 1], t = pos[3 * p + 2];
    // neighbor.get(e, entities_list): entities without a neighbour row (e.g. the other KG's
    // entities inside seed-swapped triples, kgs.py:45-50) fall back to the whole entity list
    const bool h_has = nbr && ent_pos[h] >= 0, t_has = nbr && ent_pos[t] >= 0;
    const int32_t *hc = h_has ? nbr + (int64_t)ent_pos[h] * nbr_k : entity_list;
    const int32_t *tc = t_has ? nbr + (int64_t)ent_pos[t] * nbr_k : entity_list;
    const int hn = h_has ? nbr_k : n_ent_list, tn = t_has ? nbr_k : n_ent_list;
    const uint32_t c0 = (uint32_t)pl + pos_offset;
    // mask of this group's lanes inside the 64-lane ballot
    const int gbase = (threadIdx.x & 63) / G * G;
    const unsigned long long gmask = (G == 64 ? ~0ull : ((1ull << G) - 1ull)) << gbase;
    int got = 0;
    for (int tr = 0; tr < max_try && got < k; ++tr) {
        // Philox block 0 of the round decides the side, block 1 + s is slot s's draw.  With a spare lane in the group (k < G) ONE call
        // per lane serves both: the group's last lane computes block 0 and hands its first word round, the others their own block --
        // the same numbers as two calls per lane at half the integer multiplies (the kernel's main VALU cost)
        const bool spare = k < G;
        uint4 w = oea::philox4x32_10(c0, step, (uint32_t)tr, (spare && lane != G - 1) ? 1u + (uint32_t)lane : 0u, k0, k1);
        const uint32_t side_word = spare ? (uint32_t)__shfl((int)w.x, (int)((threadIdx.x & 63) / G * G) + G - 1, 64) : w.x;
        // replay (oea_sample_negatives_replay): the round's Bernoulli and its draws come from a RECORDED run of the reference
        // (random.sample positions, np.random.binomial) instead of Philox -- same rounds, same filter, same order of acceptance
        const int32_t *rp = replay ? replay + ((int64_t)p * max_try + tr) * (1 + k) : nullptr;
        const bool corrupt_head = rp ? rp[0] != 0 : (side_word & 1u) != 0u;
        const int32_t *cand = corrupt_head ? hc : tc;
        const int nc = corrupt_head ? hn : tn;
        const int need = k - got;
        if (need > nc) { if (lane == 0) *err_flag = 1; break; }    // random.sample would raise ValueError
        const bool active = lane < need;
        uint32_t att = 0;
        int32_t v = -1 - lane;                       // inactive lanes never match anything
        if (active) {
            if (!spare) w = oea::philox4x32_10(c0, step, (uint32_t)tr, 1u + (uint32_t)lane, k0, k1);
            v = rp ? rp[1 + lane] : (int32_t)__umulhi(w.x, (uint32_t)nc);
            if (rp && (v < 0 || v >= nc)) { *err_flag = 2; v = 0; }         // the record does not fit this positive's candidate list
        }
        for (;;) {
            bool conflict = false;
            for (int q = 0; q < need; ++q) {
                const int32_t vq = __shfl(v, gbase + q, 64);
                conflict |= (q < lane) && (vq == v);
            }
            conflict &= active;
            if ((__ballot(conflict) & gmask) == 0ull) break;
            if (conflict) {
                ++att;
                w = oea::philox4x32_10(c0, step, (uint32_t)tr, 1u + (uint32_t)lane + 64u * att, k0, k1);
                v = (int32_t)__umulhi(w.x, (uint32_t)nc);
            }
        }
        bool accept = false;
        int32_t nh = h, nt = t;
        if (active) {
            const int32_t e = cand[v];
            nh = corrupt_head ? e : h;
            nt = corrupt_head ? t : e;
            accept = (tr == max_try - 1) || !contains(table, capacity, filter, filter_bits, (uint32_t)nh, (uint32_t)r, (uint32_t)nt);
        }
        const unsigned long long acc_mask = (__ballot(accept) & gmask) >> gbase;
        if (accept) {
            const int slot = got + __popcll(acc_mask & ((1ull << lane) - 1ull));
            struct __attribute__((packed, aligned(4))) Triple { int32_t h, r, t; };          // one 12-byte store per lane
            *reinterpret_cast<Triple *>(out + ((int64_t)p * k + slot) * 3) = Triple{nh, r, nt};
        }
        got += __popcll(acc_mask);
    }
    }
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

uint64_t oea_tripleset_filter_bits(uint64_t capacity) { return 8ull * capacity; }

int oea_tripleset_filter_build(const int32_t *triples, int64_t n, uint32_t *filter, uint64_t filter_bits, void *stream) {
    OEA_REQUIRE(filter && (triples || n == 0), "null pointer");
    OEA_REQUIRE(filter_bits >= 32 && (filter_bits & (filter_bits - 1)) == 0, "filter_bits: a power of two >= 32");
    hipStream_t st = oea::as_stream(stream);
    OEA_CHECK_HIP(hipMemsetAsync(filter, 0, (size_t)(filter_bits / 8), st));
    if (n > 0) tripleset_filter_insert<<<(unsigned)std::min<int64_t>(oea::ceil_div(n, 256), 4096), 256, 0, st>>>(triples, n, filter, filter_bits);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_sample_negatives(const int32_t *pos, int64_t n_pos, int32_t k, const uint64_t *table,
                         uint64_t capacity, const int32_t *entity_list, int32_t n_ent_list,
                         const int32_t *ent_pos, const int32_t *nbr, int32_t nbr_k, uint64_t seed,
                         uint32_t step, uint32_t pos_offset, int32_t max_try, int32_t *out,
                         int32_t *err_flag, void *stream) {
    oea_sampler_side sd{};
    sd.table = table; sd.capacity = capacity; sd.entity_list = entity_list; sd.ent_pos = ent_pos; sd.nbr = nbr;
    sd.n_ent_list = n_ent_list; sd.nbr_k = nbr_k;
    return oea_sample_negatives_pair(pos, n_pos, n_pos, k, &sd, &sd, seed, step, pos_offset, max_try, out, err_flag,
                                     stream);
}

static int check_side(const oea_sampler_side *s, int k) {
    OEA_REQUIRE(s && s->table && s->entity_list, "sampler side: null pointer");
    OEA_REQUIRE(s->nbr == nullptr || (s->ent_pos != nullptr && s->nbr_k > 0), "nbr needs ent_pos and nbr_k");
    OEA_REQUIRE((s->nbr ? s->nbr_k : s->n_ent_list) >= k && s->n_ent_list >= k, "Sample larger than population");
    return OEA_OK;
}

static int sample_impl(const int32_t *pos, int64_t n_pos, int64_t n_split, int32_t k,
                       const oea_sampler_side *side0, const oea_sampler_side *side1, uint64_t seed,
                       uint32_t step, uint32_t pos_offset, int32_t max_try, int32_t *out,
                       int32_t *err_flag, const int64_t *seg_off_dev, const int64_t *seg_split_dev, int n_seg,
                       void *stream, const int32_t *replay = nullptr) {
    OEA_REQUIRE(pos && out && err_flag, "null pointer");
    OEA_REQUIRE(k >= 1 && k <= kMaxK, "1 <= k <= 64");
    OEA_REQUIRE(max_try >= 1, "max_try >= 1");
    OEA_REQUIRE(n_split >= 0 && n_split <= n_pos, "0 <= n_split <= n_pos");
    int rc = check_side(side0, k);
    if (rc != OEA_OK) return rc;
    rc = check_side(side1, k);
    if (rc != OEA_OK) return rc;
    if (n_pos == 0) return OEA_OK;
    hipStream_t st = oea::as_stream(stream);
    // OEA_SAMPLER_CAP > 0: at most that many workgroups for the epoch launch (side stream).  Measured under the bench's protocol
    // (20-step regions, device synchronised at both ends): no cap 0.095 ms per step, 512 workgroups 0.113-0.125, 128 0.20 -- what the
    // side stream has not finished is waited for at the next synchronisation, so the shortest sampler wins although the step kernels
    // that run beside it take 420 instead of 40 us (tools/r06/e.sh)
    static const int64_t cap_env = [] { const char *e = getenv("OEA_SAMPLER_CAP"); return e ? (int64_t)atoll(e) : (int64_t)0; }();
    const int64_t cap = (seg_off_dev && cap_env > 0) ? cap_env : ((int64_t)1 << 30);
    if (k <= 16)
        sample_negatives_kernel<16><<<(unsigned)std::min<int64_t>(oea::ceil_div(n_pos, 256 / 16), cap), 256, 0, st>>>(
            pos, n_pos, n_split, k, *side0, *side1, (uint32_t)seed, (uint32_t)(seed >> 32), step, pos_offset, max_try,
            out, err_flag, seg_off_dev, seg_split_dev, n_seg, replay);
    else
        sample_negatives_kernel<64><<<(unsigned)std::min<int64_t>(oea::ceil_div(n_pos, 256 / 64), cap), 256, 0, st>>>(
            pos, n_pos, n_split, k, *side0, *side1, (uint32_t)seed, (uint32_t)(seed >> 32), step, pos_offset, max_try,
            out, err_flag, seg_off_dev, seg_split_dev, n_seg, replay);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_sample_negatives_pair(const int32_t *pos, int64_t n_pos, int64_t n_split, int32_t k,
                              const oea_sampler_side *side0, const oea_sampler_side *side1, uint64_t seed,
                              uint32_t step, uint32_t pos_offset, int32_t max_try, int32_t *out,
                              int32_t *err_flag, void *stream) {
    return sample_impl(pos, n_pos, n_split, k, side0, side1, seed, step, pos_offset, max_try, out, err_flag, nullptr,
                       nullptr, 0, stream);
}

int oea_sample_negatives_replay(const int32_t *pos, int64_t n_pos, int32_t k, const uint64_t *table, uint64_t capacity,
                                const int32_t *entity_list, int32_t n_ent_list, const int32_t *ent_pos, const int32_t *nbr,
                                int32_t nbr_k, int32_t max_try, const int32_t *replay, int32_t *out, int32_t *err_flag, void *stream) {
    OEA_REQUIRE(replay, "replay record");
    oea_sampler_side sd{};
    sd.table = table; sd.capacity = capacity; sd.entity_list = entity_list; sd.ent_pos = ent_pos; sd.nbr = nbr;
    sd.n_ent_list = n_ent_list; sd.nbr_k = nbr_k;
    return sample_impl(pos, n_pos, n_pos, k, &sd, &sd, 0, 0u, 0u, max_try, out, err_flag, nullptr, nullptr, 0, stream, replay);
}

int oea_sample_negatives_epoch(const int32_t *pos_all, int64_t n_rows, const int64_t *offsets_dev,
                               const int64_t *splits_dev, int32_t steps, int32_t k, const oea_sampler_side *side0,
                               const oea_sampler_side *side1, uint64_t seed, uint32_t step_base, int32_t max_try,
                               int32_t *out_all, int32_t *err_flag, void *stream) {
    OEA_REQUIRE(offsets_dev && splits_dev && steps >= 1, "offsets / splits / steps");
    return sample_impl(pos_all, n_rows, 0, k, side0, side1, seed, step_base, 0u, max_try, out_all, err_flag, offsets_dev,
                       splits_dev, steps, stream);
}

}  // extern "C"
