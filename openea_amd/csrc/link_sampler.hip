// link_sampler.hip -- negative LINKS for the GNN alignment losses, on the device.
//
// Replaces AliNet.generate_input_batch's negative half (approaches/alinet.py:988-1006):
//   uniform    k rounds of  zip(random.sample(ents1, batch), random.sample(ents2, batch))
//   truncated  per positive link (e1, e2):  (e1, c) for c in random.sample(neighbors1[e1], k)
//                                           (c, e2) for c in random.sample(neighbors2[e2], k)
//   neg_links = set(neg_links) - sup_links_set - new_sup_links_set
// `random.sample` (a uniformly random ordered subset) becomes the first `count` images of a keyed pseudo-random
// PERMUTATION of the list (4-round Feistel network on the index bits + cycle walking): distinct by construction,
// O(1) per draw, no sort, and reproduced bit for bit by oracle/np_oracle.py:link_negatives.  The python set becomes
// an open-addressing table of packed (a, b) keys: of equal pairs the one with the smallest index stays valid.
// Output: all m drawn pairs + a 0/1 mask (no device->host sync for a count; the loss weights by the mask).
#include "common.h"

namespace {

__host__ __device__ __forceinline__ uint32_t feistel_f(uint32_t r, uint32_t key, uint32_t round) {
    uint32_t v = r * 0x9E3779B1u + key + round * 0x85EBCA6Bu;
    v ^= v >> 15; v *= 0x2C1B3C6Du;
    v ^= v >> 12; v *= 0x297A2D39u;
    v ^= v >> 15;
    return v;
}
// image of i under the keyed permutation of [0, n)
__host__ __device__ __forceinline__ uint32_t perm_index(uint32_t i, uint32_t n, uint32_t key) {
    uint32_t bits = 2;
    while ((1u << bits) < n) bits += 2;
    const uint32_t half = bits >> 1, mask = (1u << half) - 1u;
    uint32_t x = i;
    do {
        uint32_t l = x >> half, r = x & mask;
#pragma unroll
        for (uint32_t round = 0; round < 4; ++round) {
            const uint32_t t = l ^ (feistel_f(r, key, round) & mask);
            l = r;
            r = t;
        }
        x = (l << half) | r;
    } while (x >= n);
    return x;
}

__global__ void link_clear_kernel(uint64_t *keys, int32_t *vals, uint64_t cap) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (uint64_t)gridDim.x * blockDim.x) {
        keys[i] = OEA_EMPTY_KEY;
        vals[i] = 0x7fffffff;
    }
}

struct LinkArgs {
    const int32_t *pos_links;
    int64_t n_pos;
    int k;
    const int32_t *ents1, *ents2;
    int n1, n2;
    const int32_t *nbr1, *row1, *nbr2, *row2;
    int nbr_k;
    uint32_t k0, k1, step;
};

__device__ __forceinline__ void link_pair(const LinkArgs &a, int64_t q, int &x, int &y) {
    if (a.nbr1 == nullptr) {                       // uniform: q = round * n_pos + draw
        const uint32_t round = (uint32_t)(q / a.n_pos), i = (uint32_t)(q % a.n_pos);
        const uint4 w = oea::philox4x32_10(round, a.step, 1u, 0u, a.k0, a.k1);
        x = a.ents1[perm_index(i, (uint32_t)a.n1, w.x)];
        y = a.ents2[perm_index(i, (uint32_t)a.n2, w.y)];
    } else {                                       // truncated: q = link * 2k + slot
        const int64_t link = q / (2 * a.k);
        const int slot = (int)(q % (2 * a.k));
        const int e1 = a.pos_links[2 * link], e2 = a.pos_links[2 * link + 1];
        const uint4 w = oea::philox4x32_10((uint32_t)link, a.step, 2u, 0u, a.k0, a.k1);
        if (slot < a.k) {
            x = e1;
            y = a.nbr1[(int64_t)a.row1[e1] * a.nbr_k + perm_index((uint32_t)slot, (uint32_t)a.nbr_k, w.x)];
        } else {
            x = a.nbr2[(int64_t)a.row2[e2] * a.nbr_k + perm_index((uint32_t)(slot - a.k), (uint32_t)a.nbr_k, w.y)];
            y = e2;
        }
    }
}

__global__ void link_generate_kernel(LinkArgs a, int64_t m, int32_t *__restrict__ out, uint64_t *keys, int32_t *vals, uint64_t cap) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= m) return;
    int x, y;
    link_pair(a, q, x, y);
    out[2 * q] = x;
    out[2 * q + 1] = y;
    const uint64_t key = ((uint64_t)(uint32_t)x << 32) | (uint32_t)y;
    uint64_t s = oea::mix64(key) & (cap - 1);
    for (;;) {
        const unsigned long long prev = atomicCAS((unsigned long long *)&keys[s], (unsigned long long)OEA_EMPTY_KEY, (unsigned long long)key);
        if (prev == OEA_EMPTY_KEY || prev == key) break;
        s = (s + 1) & (cap - 1);
    }
    atomicMin(&vals[s], (int32_t)q);
}

__global__ void link_finalize_kernel(const int32_t *__restrict__ out, int64_t m, const uint64_t *__restrict__ keys,
                                     const int32_t *__restrict__ vals, uint64_t cap, const uint64_t *__restrict__ exclude,
                                     uint64_t exclude_cap, float *__restrict__ valid) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= m) return;
    const int x = out[2 * q], y = out[2 * q + 1];
    const uint64_t key = ((uint64_t)(uint32_t)x << 32) | (uint32_t)y;
    uint64_t s = oea::mix64(key) & (cap - 1);
    while (keys[s] != key) s = (s + 1) & (cap - 1);
    bool keep = vals[s] == (int32_t)q;                      // the first of equal pairs (python set)
    if (keep && exclude) {                                  // minus the supervised links
        const uint64_t ek = oea::pack_triple((uint32_t)x, 0u, (uint32_t)y);
        uint64_t t = oea::mix64(ek) & (exclude_cap - 1);
        for (;;) {
            const uint64_t cur = exclude[t];
            if (cur == ek) { keep = false; break; }
            if (cur == OEA_EMPTY_KEY) break;
            t = (t + 1) & (exclude_cap - 1);
        }
    }
    valid[q] = keep ? 1.f : 0.f;
}

}  // namespace

extern "C" {

uint32_t oea_perm_index(uint32_t i, uint32_t n, uint32_t key) { return perm_index(i, n, key); }

int oea_sample_link_negatives(const int32_t *pos_links, int64_t n_pos, int32_t k, const int32_t *ents1, int32_t n1,
                              const int32_t *ents2, int32_t n2, const int32_t *nbr1, const int32_t *row1,
                              const int32_t *nbr2, const int32_t *row2, int32_t nbr_k, const uint64_t *exclude,
                              uint64_t exclude_cap, uint64_t seed, uint32_t step, int32_t *out_pairs, float *out_valid,
                              uint64_t *scratch_keys, int32_t *scratch_vals, uint64_t scratch_cap, void *stream) {
    OEA_REQUIRE(out_pairs && out_valid && scratch_keys && scratch_vals, "null pointer");
    OEA_REQUIRE(n_pos >= 0 && k >= 1, "n_pos >= 0, k >= 1");
    const bool truncated = nbr1 != nullptr;
    if (truncated) {
        OEA_REQUIRE(pos_links && nbr2 && row1 && row2, "truncated mode needs the links, both neighbour tables and row maps");
        if (k > nbr_k) { oea::set_error("Sample larger than population or is negative"); return OEA_EINVAL; }   // random.sample
    } else {
        OEA_REQUIRE(ents1 && ents2, "uniform mode needs both entity lists");
        if (n_pos > n1 || n_pos > n2) { oea::set_error("Sample larger than population or is negative"); return OEA_EINVAL; }
    }
    const int64_t m = (truncated ? 2 : 1) * (int64_t)k * n_pos;
    OEA_REQUIRE(m < 0x7fffffff, "too many negatives for one call");
    OEA_REQUIRE((scratch_cap & (scratch_cap - 1)) == 0 && scratch_cap >= 2 * (uint64_t)std::max<int64_t>(m, 1), "scratch_cap: power of two >= 2 m");
    OEA_REQUIRE(!exclude || (exclude_cap & (exclude_cap - 1)) == 0, "exclude_cap: power of two");
    if (m == 0) return OEA_OK;
    hipStream_t st = oea::as_stream(stream);
    LinkArgs a{pos_links, n_pos, k, ents1, ents2, n1, n2, nbr1, row1, nbr2, row2, nbr_k,
               (uint32_t)(seed & 0xFFFFFFFFu), (uint32_t)(seed >> 32), step};
    link_clear_kernel<<<(unsigned)std::min<uint64_t>(oea::ceil_div((int64_t)scratch_cap, 256), 2048), 256, 0, st>>>(scratch_keys, scratch_vals, scratch_cap);
    const unsigned nb = (unsigned)oea::ceil_div(m, 256);
    link_generate_kernel<<<nb, 256, 0, st>>>(a, m, out_pairs, scratch_keys, scratch_vals, scratch_cap);
    link_finalize_kernel<<<nb, 256, 0, st>>>(out_pairs, m, scratch_keys, scratch_vals, scratch_cap, exclude, exclude_cap, out_valid);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

}  // extern "C"
