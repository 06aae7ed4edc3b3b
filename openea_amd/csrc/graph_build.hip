// graph_build.hip -- the adjacency builders of the GNN approaches on the device.
//
// Replaces the python dictionary walks / pandas joins the reference runs once per model init:
//   approaches/gcn_align.py:610-664 (func / ifunc / get_weighted_adj) + :566-578 (D^-1/2 (A + I) D^-1/2),
//   approaches/alinet.py:155-181 (undirected 0/1 adjacency, same normalisation) and :250-287 (generate_2hop_triples:
//   self-join of the triple table, pattern ranking, cut of the 5 most frequent (r1, r2) patterns),
//   approaches/rdgcn.py:45-72 (get_mat / get_sparse_tensor) and :268-277 (Jaccard overlap of the head / tail sets).
// Everything is a sort / run-length / reduce-by-key over packed 64-bit keys (rocPRIM device primitives) plus small kernels
// for the key construction, the join expansion and the normalisation.  Sums run in a FIXED order (stable sort, then a
// segmented reduction), so replicas under torch.distributed build bit-identical operands.  Outputs come in the entry
// order of the reference's scipy pipeline (normalised operands: sorted by (col, row)); sizes are data dependent: the caller passes capacities and reads the counts back.
#include <string.h>

#include <algorithm>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace {

struct Scratch {                       // stream-ordered temporaries of one call
    hipStream_t st;
    void *p[96];
    int n = 0;
    explicit Scratch(hipStream_t s) : st(s) {}
    template <class T>
    T *get(size_t count) {
        void *q = nullptr;
        if (n >= 96 || hipMallocAsync(&q, std::max<size_t>(count, 1) * sizeof(T), st) != hipSuccess) return nullptr;
        p[n++] = q;
        return static_cast<T *>(q);
    }
    ~Scratch() {
        for (int i = 0; i < n; ++i) (void)hipFreeAsync(p[i], st);
    }
};

#define GB_ALLOC(var, T, count)                                                    \
    T *var = scratch.get<T>(count);                                                \
    if (!var) { oea::set_error("%s:%d: device allocation failed", __FILE__, __LINE__); return OEA_ENOMEM; }

#define GB_PRIM(call_with_temp)                                                    \
    do {                                                                           \
        size_t tb = 0;                                                             \
        void *temp = nullptr;                                                      \
        OEA_CHECK_HIP(call_with_temp);                                             \
        temp = scratch.get<char>(tb);                                              \
        if (!temp) { oea::set_error("%s:%d: device allocation failed", __FILE__, __LINE__); return OEA_ENOMEM; } \
        OEA_CHECK_HIP(call_with_temp);                                             \
    } while (0)

inline unsigned blocks_for(int64_t n, int per = 256) { return (unsigned)std::min<int64_t>(std::max<int64_t>(oea::ceil_div(n, per), 1), 65535 * 16); }

// sorted distinct keys (+ run lengths) of `in`
int sort_unique(Scratch &scratch, const uint64_t *in, int64_t n, uint64_t *out, uint32_t *counts, int64_t *n_out, int bits) {
    hipStream_t st = scratch.st;
    if (n == 0) { OEA_CHECK_HIP(hipMemsetAsync(n_out, 0, sizeof(int64_t), st)); return OEA_OK; }
    GB_ALLOC(sorted, uint64_t, n);
    GB_PRIM(rocprim::radix_sort_keys(temp, tb, in, sorted, (size_t)n, 0, bits, st));
    if (counts)
        GB_PRIM(rocprim::run_length_encode(temp, tb, sorted, (size_t)n, out, counts, n_out, st));
    else
        GB_PRIM(rocprim::run_length_encode(temp, tb, sorted, (size_t)n, out, rocprim::make_discard_iterator(), n_out, st));
    return OEA_OK;
}

// per distinct key (ascending) the sum of its values, added in input order (stable sort + segmented reduction)
int sum_by_key(Scratch &scratch, const uint64_t *keys, const double *vals, int64_t n, uint64_t *out_keys, double *out_vals,
               int64_t *n_out, int bits) {
    hipStream_t st = scratch.st;
    if (n == 0) { OEA_CHECK_HIP(hipMemsetAsync(n_out, 0, sizeof(int64_t), st)); return OEA_OK; }
    GB_ALLOC(sk, uint64_t, n);
    GB_ALLOC(sv, double, n);
    GB_PRIM(rocprim::radix_sort_pairs(temp, tb, keys, sk, vals, sv, (size_t)n, 0, bits, st));
    GB_PRIM(rocprim::reduce_by_key(temp, tb, sk, sv, (size_t)n, out_keys, out_vals, n_out, rocprim::plus<double>(),
                                   rocprim::equal_to<uint64_t>(), st));
    return OEA_OK;
}

int key_bits(uint64_t max_key) {
    int b = 1;
    while (b < 64 && (max_key >> b)) ++b;
    return b;
}

// ---- key construction --------------------------------------------------------------------------------------------------
// both directions of every (h, t) pair; with skip_loops, h == t gives two copies of a key no other entry can equal (the
// caller drops it): rows stay aligned with the triple index
__global__ void sym_keys_kernel(const int32_t *__restrict__ tri, int64_t n, uint64_t ne, int skip_loops, uint64_t *__restrict__ keys) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t h = (uint64_t)tri[3 * i], t = (uint64_t)tri[3 * i + 2];
        const bool drop = skip_loops && h == t;
        keys[2 * i] = drop ? ~0ull : h * ne + t;
        keys[2 * i + 1] = drop ? ~0ull : t * ne + h;
    }
}
__global__ void diag_keys_kernel(int64_t ne, uint64_t *__restrict__ keys, double *__restrict__ vals) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ne; i += (int64_t)gridDim.x * blockDim.x) {
        keys[i] = (uint64_t)i * (uint64_t)ne + (uint64_t)i;
        if (vals) vals[i] = 1.0;
    }
}
__global__ void fill_ones_kernel(double *__restrict__ v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) v[i] = 1.0;
}
// drop a trailing ~0 key (sorted input): n_keep = n - (last == ~0)
__global__ void trim_sentinel_kernel(const uint64_t *__restrict__ keys, int64_t *__restrict__ n) {
    if (*n > 0 && keys[*n - 1] == ~0ull) --*n;
}

// ---- D^-1/2 (A) D^-1/2 on sorted (row-major) entries ---------------------------------------------------------------------
// rows are contiguous: one thread per entry start-of-row would serialise hubs; instead reduce_by_key on the row ids
struct RowOf {
    uint64_t ne;
    __host__ __device__ uint64_t operator()(uint64_t k) const { return k / ne; }
};
__global__ void scatter_rowsum_kernel(const uint64_t *__restrict__ rows, const double *__restrict__ sums, const int64_t *__restrict__ n,
                                      double *__restrict__ dinv) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < *n; i += (int64_t)gridDim.x * blockDim.x) {
        const double s = sums[i];
        // np.power(rowsum, -0.5) with inf -> 0 (gcn_align.py:569-571)
        dinv[rows[i]] = s == 0.0 ? 0.0 : 1.0 / sqrt(s);
    }
}
// entry (i, j) of A (key i * ne + j, value a; keys ascending = row-major) -> entry (j, i) of (A D)^T D with the value
// (a * dinv[j]) * dinv[i] (scipy's order of the two products).  Written in the order of the input, the result is sorted by
// (col, row): the order scipy's csc -> coo conversion hands to the reference (which AliNet's 'runs' softmax grouping sees).
__global__ void normalize_entries_kernel(const uint64_t *__restrict__ keys, const double *__restrict__ a, const int64_t *__restrict__ n,
                                         uint64_t ne, const double *__restrict__ dinv, int32_t *__restrict__ row, int32_t *__restrict__ col,
                                         double *__restrict__ out_vals) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < *n; e += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t i = keys[e] / ne, j = keys[e] % ne;
        out_vals[e] = (a[e] * dinv[j]) * dinv[i];
        row[e] = (int32_t)j;
        col[e] = (int32_t)i;
    }
}
__global__ void split_keys_kernel(const uint64_t *__restrict__ keys, const int64_t *__restrict__ n, uint64_t ne, int32_t *__restrict__ row,
                                  int32_t *__restrict__ col) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < *n; e += (int64_t)gridDim.x * blockDim.x) {
        row[e] = (int32_t)(keys[e] / ne);
        col[e] = (int32_t)(keys[e] % ne);
    }
}

// (A + I) normalised = ((A + I) D^-1/2)^T D^-1/2; `keys` / `vals` hold the n_a entries of A first (unsorted, duplicates
// summed) and room for ne more
int normalized_support(Scratch &scratch, uint64_t *keys, double *vals, int64_t n_a, int64_t ne, int32_t *row, int32_t *col,
                       double *val, int64_t cap, int64_t *nnz_dev) {
    hipStream_t st = scratch.st;
    const int bits = key_bits((uint64_t)ne * (uint64_t)ne);
    diag_keys_kernel<<<blocks_for(ne), 256, 0, st>>>(ne, keys + n_a, vals + n_a);
    const int64_t n = n_a + ne;
    OEA_REQUIRE(cap >= n, "output capacity < entries + entities");
    GB_ALLOC(uk, uint64_t, n);
    GB_ALLOC(uv, double, n);
    int rc = sum_by_key(scratch, keys, vals, n, uk, uv, nnz_dev, bits);
    if (rc != OEA_OK) return rc;
    GB_ALLOC(rows, uint64_t, ne);
    GB_ALLOC(sums, double, ne);
    GB_ALLOC(nrows, int64_t, 1);
    GB_ALLOC(dinv, double, ne);
    OEA_CHECK_HIP(hipMemsetAsync(dinv, 0, sizeof(double) * (size_t)ne, st));
    // the entry count lives on the device: the reductions run over the capacity with the tail masked by a sentinel row
    // (keys past *nnz are not initialised) -- simpler: read the count back; builders run once per model init
    int64_t nnz = 0;
    OEA_CHECK_HIP(hipMemcpyAsync(&nnz, nnz_dev, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    OEA_CHECK_HIP(hipStreamSynchronize(st));
    auto row_it = rocprim::make_transform_iterator(uk, RowOf{(uint64_t)ne});
    GB_PRIM(rocprim::reduce_by_key(temp, tb, row_it, uv, (size_t)nnz, rows, sums, nrows, rocprim::plus<double>(),
                                   rocprim::equal_to<uint64_t>(), st));
    scatter_rowsum_kernel<<<blocks_for(ne), 256, 0, st>>>(rows, sums, nrows, dinv);
    normalize_entries_kernel<<<blocks_for(nnz), 256, 0, st>>>(uk, uv, nnz_dev, (uint64_t)ne, dinv, row, col, val);
    OEA_CHECK_HIP(hipGetLastError());
    OEA_CHECK_HIP(hipStreamSynchronize(st));          // the temporaries go back to the pool after the kernels that read them
    return OEA_OK;
}

// ---- GCN-Align: functionality weights -------------------------------------------------------------------------------------
__global__ void rel_pair_keys_kernel(const int32_t *__restrict__ tri, int64_t n, int side, uint64_t *__restrict__ keys,
                                     int32_t *__restrict__ cnt) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t r = (uint64_t)tri[3 * i + 1], e = (uint64_t)tri[3 * i + side];
        keys[i] = (r << 32) | e;
        if (cnt) atomicAdd(cnt + r, 1);
    }
}
__global__ void count_distinct_kernel(const uint64_t *__restrict__ keys, const int64_t *__restrict__ n, int32_t *__restrict__ distinct) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < *n; i += (int64_t)gridDim.x * blockDim.x)
        atomicAdd(distinct + (keys[i] >> 32), 1);
}
__global__ void ratio_kernel(const int32_t *__restrict__ distinct, const int32_t *__restrict__ cnt, int64_t n, double *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = cnt[i] > 0 ? (double)distinct[i] / (double)cnt[i] : 0.0;
}
// gcn_align.py:650-656: M[(h, t)] += max(r2if[r], .3), M[(t, h)] += max(r2f[r], .3); the COO entry of key (a, b) sits at
// row b, column a (:659-662) -> packed as b * ne + a
__global__ void weighted_keys_kernel(const int32_t *__restrict__ tri, int64_t n, uint64_t ne, const double *__restrict__ r2f,
                                     const double *__restrict__ r2if, uint64_t *__restrict__ keys, double *__restrict__ vals) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t h = (uint64_t)tri[3 * i], t = (uint64_t)tri[3 * i + 2];
        const int r = tri[3 * i + 1];
        const bool drop = h == t;                                   // :652-653
        keys[2 * i] = drop ? ~0ull : t * ne + h;
        vals[2 * i] = drop ? 0.0 : fmax(r2if[r], 0.3);
        keys[2 * i + 1] = drop ? ~0ull : h * ne + t;
        vals[2 * i + 1] = drop ? 0.0 : fmax(r2f[r], 0.3);
    }
}
// compaction of the dropped (~0) pairs: stable partition by flag
struct NotSentinel {
    __host__ __device__ bool operator()(uint64_t k) const { return k != ~0ull; }
};

// ---- RDGCN -------------------------------------------------------------------------------------------------------------
// rdgcn.py:45-52: degree starts at 1; a triple whose head id differs from its RELATION id adds one to degree[head] and to
// degree[relation id] (the reference's indexing, kept as it is)
__global__ void primal_degree_kernel(const int32_t *__restrict__ tri, int64_t n, int64_t ne, int32_t *__restrict__ degree) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int h = tri[3 * i], r = tri[3 * i + 1];
        if (h != r) {
            atomicAdd(degree + h, 1);
            if (r < ne) atomicAdd(degree + r, 1);
        }
    }
}
__global__ void fill_i32_kernel(int32_t *__restrict__ p, int64_t n, int32_t v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
// key fir * ne + sec -> M[sec, fir] = 1 / sqrt(deg[fir]) / sqrt(deg[sec])   (rdgcn.py:63-72)
__global__ void primal_values_kernel(const uint64_t *__restrict__ keys, const int64_t *__restrict__ n, uint64_t ne,
                                     const int32_t *__restrict__ degree, int32_t *__restrict__ row, int32_t *__restrict__ col,
                                     float *__restrict__ val) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < *n; e += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t fir = keys[e] / ne, sec = keys[e] % ne;
        row[e] = (int32_t)sec;
        col[e] = (int32_t)fir;
        val[e] = (float)(1.0 / sqrt((double)degree[fir]) / sqrt((double)degree[sec]));
    }
}
__global__ void swap_halves_kernel(const uint64_t *__restrict__ in, int64_t n, uint64_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (in[i] << 32) | (in[i] >> 32);
}
// entity-major (entity << 32 | relation) distinct keys: one wave per entity walks the pairs of its relations
__global__ void pair_overlap_kernel(const uint64_t *__restrict__ keys, const int64_t *__restrict__ n, int64_t n_rel, int32_t *__restrict__ inter) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t total = *n;
    // a wave owns the segments that START in its slice of the key array
    const int64_t per = (total + nwave - 1) / nwave;
    int64_t i = wave * per;
    const int64_t end = min(total, i + per);
    while (i < end) {
        const uint64_t ent = keys[i] >> 32;
        if (i > 0 && (keys[i - 1] >> 32) == ent) { ++i; continue; }       // not a segment start
        int64_t j = i + 1;
        while (j < total && (keys[j] >> 32) == ent) ++j;
        const int64_t len = j - i;
        for (int64_t p = lane; p < len * len; p += 64) {
            const int64_t a = (int64_t)(keys[i + p / len] & 0xffffffffull), b = (int64_t)(keys[i + p % len] & 0xffffffffull);
            atomicAdd(inter + a * n_rel + b, 1);
        }
        i = j;
    }
}
// rdgcn.py:268-277: |A & B| / |A | B| of the head sets + of the tail sets
__global__ void jaccard_kernel(const int32_t *__restrict__ ih, const int32_t *__restrict__ it, int64_t n_rel, float *__restrict__ out) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_rel * n_rel; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t a = e / n_rel, b = e % n_rel;
        double s = 0.0;
        {
            const double inter = (double)ih[e], uni = (double)ih[a * n_rel + a] + (double)ih[b * n_rel + b] - inter;
            s += uni > 0.0 ? inter / uni : 0.0;
        }
        {
            const double inter = (double)it[e], uni = (double)it[a * n_rel + a] + (double)it[b * n_rel + b] - inter;
            s += uni > 0.0 ? inter / uni : 0.0;
        }
        out[e] = (float)s;
    }
}

// ---- AliNet 2-hop join ---------------------------------------------------------------------------------------------------
__global__ void head_index_kernel(const int32_t *__restrict__ tri, int64_t n, uint32_t *__restrict__ heads, int32_t *__restrict__ idx) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        heads[i] = (uint32_t)tri[3 * i];
        idx[i] = (int32_t)i;
    }
}
__device__ __forceinline__ int64_t lower_bound_u32(const uint32_t *a, int64_t n, uint32_t v) {
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (a[m] < v) lo = m + 1; else hi = m; }
    return lo;
}
__device__ __forceinline__ bool contains_u64(const uint64_t *a, int64_t n, uint64_t v) {
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (a[m] < v) lo = m + 1; else hi = m; }
    return lo < n && a[lo] == v;
}
// matches of left triple i = right triples whose head is tri[i].tail (pd.merge(left_on='t', right_on='h'), alinet.py:256)
__global__ void join_count_kernel(const int32_t *__restrict__ tri, int64_t n, const uint32_t *__restrict__ heads_sorted,
                                  int64_t *__restrict__ lo_out, int64_t *__restrict__ cnt) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t t = (uint32_t)tri[3 * i + 2];
        const int64_t lo = lower_bound_u32(heads_sorted, n, t);
        int64_t hi = (t == 0xffffffffu) ? n : lower_bound_u32(heads_sorted, n, t + 1);
        lo_out[i] = lo;
        cnt[i] = hi - lo;
    }
}
// merged row m (left-major, matches in table order): (h, r1, r2, t); rows whose (h, t) is an edge of the full KG are dropped
// (alinet.py:262-266: "tail not in out[head] and head not in in[tail]").  Kept rows: pattern key + row index for the ranking.
__global__ void join_expand_kernel(const int32_t *__restrict__ tri, int64_t n, const int32_t *__restrict__ by_head,
                                   const int64_t *__restrict__ lo, const int64_t *__restrict__ offs, int64_t total,
                                   const uint64_t *__restrict__ edges, const int64_t *__restrict__ n_edges, uint64_t nn, uint64_t n_rel,
                                   uint64_t *__restrict__ pat, uint64_t *__restrict__ quad, uint64_t *__restrict__ hop) {
    for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < total; m += (int64_t)gridDim.x * blockDim.x) {
        // left triple of merged row m: last i with offs[i] <= m
        int64_t a = 0, b = n;
        while (b - a > 1) { const int64_t c = (a + b) >> 1; if (offs[c] <= m) a = c; else b = c; }
        const int64_t i = a;
        const int32_t right = by_head[lo[i] + (m - offs[i])];
        const uint64_t h = (uint64_t)tri[3 * i], r1 = (uint64_t)tri[3 * i + 1], r2 = (uint64_t)tri[3 * right + 1], t = (uint64_t)tri[3 * right + 2];
        const bool keep = !contains_u64(edges, *n_edges, h * nn + t);
        const uint64_t p = r1 * n_rel + r2;
        pat[m] = keep ? p : ~0ull;
        quad[m] = keep ? (h * (n_rel * n_rel) + p) * nn + t : ~0ull;          // distinct (h, r1, r2, t): the reference's log line
        hop[m] = keep ? (h * (2 * n_rel) + (r1 + r2)) * nn + t : ~0ull;       // (h, r1 + r2, t)
    }
}
__global__ void iota_kernel(int64_t *__restrict__ p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = i;
}
// `sorted(counts.items(), key=count, reverse=True)[:5]` (alinet.py:270-274): python's sort is stable, so ties keep the order
// of first appearance in the merged table.  runs: distinct patterns ascending, their lengths, and the merged-row index of
// their first member (the pairs were sorted stably by pattern).  One workgroup, 5 rounds of arg-max.
__global__ __launch_bounds__(1024) void top_patterns_kernel(const uint64_t *__restrict__ pats, const uint32_t *__restrict__ counts,
                                                            const int64_t *__restrict__ run_start, const int64_t *__restrict__ first_rows,
                                                            const int64_t *__restrict__ n_runs, int n_top, uint64_t *__restrict__ top) {
    __shared__ unsigned long long best_c[1024];
    __shared__ long long best_f[1024];
    __shared__ long long best_i[1024];
    __shared__ long long taken[8];
    const int64_t n = *n_runs - ((*n_runs > 0 && pats[*n_runs - 1] == ~0ull) ? 1 : 0);       // the dropped rows' run sorts last
    for (int round = 0; round < n_top; ++round) {
        unsigned long long bc = 0; long long bf = 0x7fffffffffffffffll, bi = -1;
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
            bool used = false;
            for (int q = 0; q < round; ++q) used |= taken[q] == i;
            if (used) continue;
            const unsigned long long c = counts[i];
            const long long f = first_rows[run_start[i]];
            if (c > bc || (c == bc && f < bf)) { bc = c; bf = f; bi = i; }
        }
        best_c[threadIdx.x] = bc; best_f[threadIdx.x] = bf; best_i[threadIdx.x] = bi;
        __syncthreads();
        for (int s = 512; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) {
                const int o = threadIdx.x + s;
                if (best_i[o] >= 0 && (best_i[threadIdx.x] < 0 || best_c[o] > best_c[threadIdx.x] ||
                                       (best_c[o] == best_c[threadIdx.x] && best_f[o] < best_f[threadIdx.x]))) {
                    best_c[threadIdx.x] = best_c[o]; best_f[threadIdx.x] = best_f[o]; best_i[threadIdx.x] = best_i[o];
                }
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) { taken[round] = best_i[0]; top[round] = best_i[0] >= 0 ? pats[best_i[0]] : ~0ull; }
        __syncthreads();
    }
}
__global__ void exclusive_from_counts_kernel(const uint32_t *__restrict__ counts, const int64_t *__restrict__ n, int64_t *__restrict__ start) {
    // single thread block scan: pattern runs are tens of thousands
    __shared__ long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < *n; base += blockDim.x) {
        const int64_t i = base + threadIdx.x;
        long long v = i < *n ? (long long)counts[i] : 0;
        // block-wide inclusive scan (Hillis-Steele in shared memory)
        __shared__ long long buf[1024];
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int s = 1; s < (int)blockDim.x; s <<= 1) {
            const long long add = (int)threadIdx.x >= s ? buf[threadIdx.x - s] : 0;
            __syncthreads();
            buf[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < *n) start[i] = carry + buf[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == blockDim.x - 1) carry += buf[threadIdx.x];
        __syncthreads();
    }
}
// rows of the selected patterns -> their (h, r1 + r2, t) key and the loop key (h, 0, h); the others -> ~0
__global__ void select_hops_kernel(const uint64_t *__restrict__ pat, const uint64_t *__restrict__ hop, int64_t total,
                                   const uint64_t *__restrict__ top, int n_top, uint64_t nn, uint64_t n_rel, uint64_t *__restrict__ out) {
    for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < total; m += (int64_t)gridDim.x * blockDim.x) {
        bool sel = pat[m] != ~0ull;
        for (int q = 0; q < n_top; ++q) sel &= pat[m] != top[q];
        const uint64_t h = sel ? hop[m] / nn / (2 * n_rel) : 0;
        out[2 * m] = sel ? hop[m] : ~0ull;
        out[2 * m + 1] = sel ? (h * (2 * n_rel)) * nn + h : ~0ull;
    }
}
__global__ void split_hops_kernel(const uint64_t *__restrict__ keys, const int64_t *__restrict__ n, uint64_t nn, uint64_t n_rel,
                                  int32_t *__restrict__ out) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < *n; e += (int64_t)gridDim.x * blockDim.x) {
        out[3 * e] = (int32_t)(keys[e] / nn / (2 * n_rel));
        out[3 * e + 1] = (int32_t)(keys[e] / nn % (2 * n_rel));
        out[3 * e + 2] = (int32_t)(keys[e] % nn);
    }
}
__global__ void edge_keys_kernel(const int32_t *__restrict__ tri, int64_t n, uint64_t nn, uint64_t *__restrict__ keys) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        keys[i] = (uint64_t)tri[3 * i] * nn + (uint64_t)tri[3 * i + 2];
}

}  // namespace

extern "C" {

int oea_build_unweighted_adj(const int32_t *tri, int64_t n_tri, int64_t n_ent, int32_t *row, int32_t *col, double *val, int64_t cap,
                             int64_t *nnz_dev, void *stream) {
    OEA_REQUIRE(tri && row && col && val && nnz_dev && n_tri >= 0 && n_ent > 0 && n_ent < (1ll << 31), "arguments");
    OEA_REQUIRE(cap >= 2 * n_tri + n_ent, "cap >= 2 n_tri + n_ent");
    Scratch scratch(oea::as_stream(stream));
    hipStream_t st = scratch.st;
    const int bits = key_bits((uint64_t)n_ent * (uint64_t)n_ent);
    GB_ALLOC(raw, uint64_t, 2 * n_tri);
    GB_ALLOC(keys, uint64_t, 2 * n_tri + n_ent);
    GB_ALLOC(vals, double, 2 * n_tri + n_ent);
    GB_ALLOC(n_u, int64_t, 1);
    sym_keys_kernel<<<blocks_for(n_tri), 256, 0, st>>>(tri, n_tri, (uint64_t)n_ent, 0, raw);
    int rc = sort_unique(scratch, raw, 2 * n_tri, keys, nullptr, n_u, bits);      // the reference's dict of sets: distinct pairs
    if (rc != OEA_OK) return rc;
    int64_t n_a = 0;
    OEA_CHECK_HIP(hipMemcpyAsync(&n_a, n_u, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    OEA_CHECK_HIP(hipStreamSynchronize(st));
    fill_ones_kernel<<<blocks_for(n_a), 256, 0, st>>>(vals, n_a);
    return normalized_support(scratch, keys, vals, n_a, n_ent, row, col, val, cap, nnz_dev);
}

int oea_build_weighted_adj(const int32_t *tri, int64_t n_tri, int64_t n_ent, int64_t n_rel, double *r2f, double *r2if, int32_t *adj_row,
                           int32_t *adj_col, double *adj_val, int64_t *adj_nnz_dev, int32_t *row, int32_t *col, double *val,
                           int64_t cap, int64_t *nnz_dev, void *stream) {
    OEA_REQUIRE(tri && r2f && r2if && row && col && val && nnz_dev && n_tri >= 0 && n_ent > 0 && n_ent < (1ll << 31) && n_rel > 0,
                "arguments");
    OEA_REQUIRE((adj_row == nullptr) == (adj_col == nullptr) && (adj_row == nullptr) == (adj_val == nullptr) &&
                    (adj_row == nullptr) == (adj_nnz_dev == nullptr), "the raw adjacency outputs go together");
    OEA_REQUIRE(cap >= 2 * n_tri + n_ent, "cap >= 2 n_tri + n_ent");
    Scratch scratch(oea::as_stream(stream));
    hipStream_t st = scratch.st;
    const int bits = key_bits((uint64_t)n_ent * (uint64_t)n_ent);
    // functionality weights (gcn_align.py:610-640)
    GB_ALLOC(cnt, int32_t, n_rel);
    GB_ALLOC(dh, int32_t, n_rel);
    GB_ALLOC(dt, int32_t, n_rel);
    GB_ALLOC(pk, uint64_t, n_tri);
    GB_ALLOC(pu, uint64_t, n_tri);
    GB_ALLOC(n_u, int64_t, 1);
    OEA_CHECK_HIP(hipMemsetAsync(cnt, 0, sizeof(int32_t) * (size_t)n_rel, st));
    OEA_CHECK_HIP(hipMemsetAsync(dh, 0, sizeof(int32_t) * (size_t)n_rel, st));
    OEA_CHECK_HIP(hipMemsetAsync(dt, 0, sizeof(int32_t) * (size_t)n_rel, st));
    rel_pair_keys_kernel<<<blocks_for(n_tri), 256, 0, st>>>(tri, n_tri, 0, pk, cnt);
    int rc = sort_unique(scratch, pk, n_tri, pu, nullptr, n_u, 64);
    if (rc != OEA_OK) return rc;
    count_distinct_kernel<<<blocks_for(n_tri), 256, 0, st>>>(pu, n_u, dh);
    rel_pair_keys_kernel<<<blocks_for(n_tri), 256, 0, st>>>(tri, n_tri, 2, pk, nullptr);
    rc = sort_unique(scratch, pk, n_tri, pu, nullptr, n_u, 64);
    if (rc != OEA_OK) return rc;
    count_distinct_kernel<<<blocks_for(n_tri), 256, 0, st>>>(pu, n_u, dt);
    ratio_kernel<<<blocks_for(n_rel), 256, 0, st>>>(dh, cnt, n_rel, r2f);
    ratio_kernel<<<blocks_for(n_rel), 256, 0, st>>>(dt, cnt, n_rel, r2if);
    // weighted entries (gcn_align.py:642-664), duplicates summed in triple order
    GB_ALLOC(wk, uint64_t, 2 * n_tri);
    GB_ALLOC(wv, double, 2 * n_tri);
    GB_ALLOC(keys, uint64_t, 2 * n_tri + n_ent);
    GB_ALLOC(vals, double, 2 * n_tri + n_ent);
    weighted_keys_kernel<<<blocks_for(n_tri), 256, 0, st>>>(tri, n_tri, (uint64_t)n_ent, r2f, r2if, wk, wv);
    rc = sum_by_key(scratch, wk, wv, 2 * n_tri, keys, vals, n_u, 64);
    if (rc != OEA_OK) return rc;
    trim_sentinel_kernel<<<1, 1, 0, st>>>(keys, n_u);
    int64_t n_a = 0;
    OEA_CHECK_HIP(hipMemcpyAsync(&n_a, n_u, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    OEA_CHECK_HIP(hipStreamSynchronize(st));
    if (adj_row) {
        OEA_CHECK_HIP(hipMemcpyAsync(adj_nnz_dev, n_u, sizeof(int64_t), hipMemcpyDeviceToDevice, st));
        OEA_CHECK_HIP(hipMemcpyAsync(adj_val, vals, sizeof(double) * (size_t)n_a, hipMemcpyDeviceToDevice, st));
        split_keys_kernel<<<blocks_for(n_a), 256, 0, st>>>(keys, n_u, (uint64_t)n_ent, adj_row, adj_col);
    }
    (void)bits;
    // support = normalize_adj(adj + I) = (A' D)^T D  (gcn_align.py:566-578)
    return normalized_support(scratch, keys, vals, n_a, n_ent, row, col, val, cap, nnz_dev);
}

int oea_build_primal_adj(const int32_t *tri, int64_t n_tri, int64_t n_ent, int32_t *row, int32_t *col, float *val, int64_t cap,
                         int64_t *nnz_dev, void *stream) {
    OEA_REQUIRE(tri && row && col && val && nnz_dev && n_tri >= 0 && n_ent > 0 && n_ent < (1ll << 31), "arguments");
    OEA_REQUIRE(cap >= 2 * n_tri + n_ent, "cap >= 2 n_tri + n_ent");
    Scratch scratch(oea::as_stream(stream));
    hipStream_t st = scratch.st;
    GB_ALLOC(degree, int32_t, n_ent);
    GB_ALLOC(raw, uint64_t, 2 * n_tri + n_ent);
    GB_ALLOC(keys, uint64_t, 2 * n_tri + n_ent);
    fill_i32_kernel<<<blocks_for(n_ent), 256, 0, st>>>(degree, n_ent, 1);
    primal_degree_kernel<<<blocks_for(n_tri), 256, 0, st>>>(tri, n_tri, n_ent, degree);
    sym_keys_kernel<<<blocks_for(n_tri), 256, 0, st>>>(tri, n_tri, (uint64_t)n_ent, 1, raw);
    diag_keys_kernel<<<blocks_for(n_ent), 256, 0, st>>>(n_ent, raw + 2 * n_tri, nullptr);
    int rc = sort_unique(scratch, raw, 2 * n_tri + n_ent, keys, nullptr, nnz_dev, 64);
    if (rc != OEA_OK) return rc;
    trim_sentinel_kernel<<<1, 1, 0, st>>>(keys, nnz_dev);
    primal_values_kernel<<<blocks_for(2 * n_tri + n_ent), 256, 0, st>>>(keys, nnz_dev, (uint64_t)n_ent, degree, row, col, val);
    OEA_CHECK_HIP(hipGetLastError());
    OEA_CHECK_HIP(hipStreamSynchronize(st));
    return OEA_OK;
}

int oea_build_dual_adj(const int32_t *tri, int64_t n_tri, int64_t n_rel, float *out, void *stream) {
    OEA_REQUIRE(tri && out && n_tri >= 0 && n_rel > 0 && n_rel <= 46340, "arguments (n_rel^2 < 2^31)");
    Scratch scratch(oea::as_stream(stream));
    hipStream_t st = scratch.st;
    GB_ALLOC(pk, uint64_t, n_tri);
    GB_ALLOC(pu, uint64_t, n_tri);
    GB_ALLOC(n_u, int64_t, 1);
    GB_ALLOC(ih, int32_t, n_rel * n_rel);
    GB_ALLOC(it, int32_t, n_rel * n_rel);
    OEA_CHECK_HIP(hipMemsetAsync(ih, 0, sizeof(int32_t) * (size_t)(n_rel * n_rel), st));
    OEA_CHECK_HIP(hipMemsetAsync(it, 0, sizeof(int32_t) * (size_t)(n_rel * n_rel), st));
    for (int side = 0; side <= 2; side += 2) {
        // distinct (relation << 32 | entity) pairs first, then the same pairs entity-major
        rel_pair_keys_kernel<<<blocks_for(n_tri), 256, 0, st>>>(tri, n_tri, side, pk, nullptr);
        int rc = sort_unique(scratch, pk, n_tri, pu, nullptr, n_u, 64);
        if (rc != OEA_OK) return rc;
        int64_t n_p = 0;
        OEA_CHECK_HIP(hipMemcpyAsync(&n_p, n_u, sizeof(int64_t), hipMemcpyDeviceToHost, st));
        OEA_CHECK_HIP(hipStreamSynchronize(st));
        GB_ALLOC(sw, uint64_t, n_p);
        GB_ALLOC(ss, uint64_t, n_p);
        swap_halves_kernel<<<blocks_for(n_p), 256, 0, st>>>(pu, n_p, sw);
        GB_PRIM(rocprim::radix_sort_keys(temp, tb, sw, ss, (size_t)n_p, 0, 64, st));
        pair_overlap_kernel<<<256, 256, 0, st>>>(ss, n_u, n_rel, side == 0 ? ih : it);
    }
    jaccard_kernel<<<blocks_for(n_rel * n_rel), 256, 0, st>>>(ih, it, n_rel, out);
    OEA_CHECK_HIP(hipGetLastError());
    OEA_CHECK_HIP(hipStreamSynchronize(st));
    return OEA_OK;
}

int oea_build_2hop(const int32_t *tri, int64_t n_tri, const int32_t *full_tri, int64_t n_full, int64_t n_ent, int64_t n_rel, int32_t n_cut,
                   int32_t *out, int64_t cap, int64_t *n_out_dev, int64_t *stats_host, void *stream) {
    OEA_REQUIRE(tri && full_tri && out && n_out_dev && stats_host && n_tri >= 0 && n_full >= 0 && n_ent > 0 && n_rel > 0, "arguments");
    OEA_REQUIRE(n_cut >= 0 && n_cut <= 8, "0 <= n_cut <= 8");
    // key ranges: (h * n_rel^2 + r1 * n_rel + r2) * n_ent + t must fit 64 bits
    const long double span = (long double)n_ent * (long double)n_ent * (long double)n_rel * (long double)n_rel;
    OEA_REQUIRE(span < 1.8e19L, "n_ent^2 * n_rel^2 must stay below 2^64");
    Scratch scratch(oea::as_stream(stream));
    hipStream_t st = scratch.st;
    stats_host[0] = stats_host[1] = stats_host[2] = stats_host[3] = 0;
    OEA_CHECK_HIP(hipMemsetAsync(n_out_dev, 0, sizeof(int64_t), st));
    if (n_tri == 0) { OEA_CHECK_HIP(hipStreamSynchronize(st)); return OEA_OK; }
    const uint64_t nn = (uint64_t)n_ent, nr = (uint64_t)n_rel;
    // right side grouped by head, table order kept inside a group
    GB_ALLOC(heads, uint32_t, n_tri);
    GB_ALLOC(idx, int32_t, n_tri);
    GB_ALLOC(heads_sorted, uint32_t, n_tri);
    GB_ALLOC(by_head, int32_t, n_tri);
    head_index_kernel<<<blocks_for(n_tri), 256, 0, st>>>(tri, n_tri, heads, idx);
    GB_PRIM(rocprim::radix_sort_pairs(temp, tb, heads, heads_sorted, idx, by_head, (size_t)n_tri, 0, 32, st));
    GB_ALLOC(lo, int64_t, n_tri);
    GB_ALLOC(cnt, int64_t, n_tri + 1);
    GB_ALLOC(offs, int64_t, n_tri + 1);
    OEA_CHECK_HIP(hipMemsetAsync(cnt + n_tri, 0, sizeof(int64_t), st));
    join_count_kernel<<<blocks_for(n_tri), 256, 0, st>>>(tri, n_tri, heads_sorted, lo, cnt);
    GB_PRIM(rocprim::exclusive_scan(temp, tb, cnt, offs, (int64_t)0, (size_t)(n_tri + 1), rocprim::plus<int64_t>(), st));
    int64_t total = 0;
    OEA_CHECK_HIP(hipMemcpyAsync(&total, offs + n_tri, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    OEA_CHECK_HIP(hipStreamSynchronize(st));
    if (total == 0) return OEA_OK;
    // edges of the FULL kg
    GB_ALLOC(ek, uint64_t, n_full);
    GB_ALLOC(edges, uint64_t, n_full);
    GB_ALLOC(n_edges, int64_t, 1);
    edge_keys_kernel<<<blocks_for(n_full), 256, 0, st>>>(full_tri, n_full, nn, ek);
    int rc = sort_unique(scratch, ek, n_full, edges, nullptr, n_edges, 64);
    if (rc != OEA_OK) return rc;
    GB_ALLOC(pat, uint64_t, total);
    GB_ALLOC(quad, uint64_t, total);
    GB_ALLOC(hop, uint64_t, total);
    join_expand_kernel<<<blocks_for(total), 256, 0, st>>>(tri, n_tri, by_head, lo, offs, total, edges, n_edges, nn, nr, pat, quad, hop);
    // log line 1: distinct kept (h, r1, r2, t)
    GB_ALLOC(uq, uint64_t, total);
    GB_ALLOC(n_q, int64_t, 1);
    rc = sort_unique(scratch, quad, total, uq, nullptr, n_q, 64);
    if (rc != OEA_OK) return rc;
    trim_sentinel_kernel<<<1, 1, 0, st>>>(uq, n_q);
    // pattern ranking: stable sort of (pattern, merged row), runs = patterns
    GB_ALLOC(rows, int64_t, total);
    GB_ALLOC(sp, uint64_t, total);
    GB_ALLOC(sr, int64_t, total);
    iota_kernel<<<blocks_for(total), 256, 0, st>>>(rows, total);
    GB_PRIM(rocprim::radix_sort_pairs(temp, tb, pat, sp, rows, sr, (size_t)total, 0, 64, st));
    GB_ALLOC(up, uint64_t, total);
    GB_ALLOC(uc, uint32_t, total);
    GB_ALLOC(n_p, int64_t, 1);
    GB_PRIM(rocprim::run_length_encode(temp, tb, sp, (size_t)total, up, uc, n_p, st));
    int64_t h_np = 0, h_nq = 0;
    OEA_CHECK_HIP(hipMemcpyAsync(&h_np, n_p, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    OEA_CHECK_HIP(hipMemcpyAsync(&h_nq, n_q, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    OEA_CHECK_HIP(hipStreamSynchronize(st));
    GB_ALLOC(run_start, int64_t, h_np);
    exclusive_from_counts_kernel<<<1, 1024, 0, st>>>(uc, n_p, run_start);
    GB_ALLOC(top, uint64_t, 8);
    OEA_CHECK_HIP(hipMemsetAsync(top, 0xff, sizeof(uint64_t) * 8, st));
    if (n_cut > 0) top_patterns_kernel<<<1, 1024, 0, st>>>(up, uc, run_start, sr, n_p, n_cut, top);
    // the kept patterns' hops + self loops, distinct, sorted by (h, r, t)
    GB_ALLOC(hk, uint64_t, 2 * total);
    GB_ALLOC(hu, uint64_t, 2 * total);
    select_hops_kernel<<<blocks_for(total), 256, 0, st>>>(pat, hop, total, top, n_cut, nn, nr, hk);
    rc = sort_unique(scratch, hk, 2 * total, hu, nullptr, n_out_dev, 64);
    if (rc != OEA_OK) return rc;
    trim_sentinel_kernel<<<1, 1, 0, st>>>(hu, n_out_dev);
    int64_t n_out = 0;
    uint64_t last_pat = 0;
    OEA_CHECK_HIP(hipMemcpyAsync(&n_out, n_out_dev, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    OEA_CHECK_HIP(hipMemcpyAsync(&last_pat, up + (h_np - 1), sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    OEA_CHECK_HIP(hipStreamSynchronize(st));
    OEA_REQUIRE(cap >= n_out, "output capacity");
    split_hops_kernel<<<blocks_for(n_out), 256, 0, st>>>(hu, n_out_dev, nn, nr, out);
    const int64_t n_patterns = h_np - (last_pat == ~0ull ? 1 : 0);
    stats_host[0] = h_nq;                              // "total 2-hop neighbors"
    stats_host[1] = n_patterns;                        // "total 2-hop relation patterns"
    stats_host[2] = std::max<int64_t>(n_patterns - n_cut, 0);   // "selected relation patterns"
    stats_host[3] = n_out;                             // "selected 2-hop neighbors"
    OEA_CHECK_HIP(hipGetLastError());
    OEA_CHECK_HIP(hipStreamSynchronize(st));
    return OEA_OK;
}

// ---- the endpoints of a pair list grouped by row (oea_pair_grad_rows) -----------------------------------------------------
// pairs int32 [m, 2]: slot i < m = the FIRST element of pair i, slot m + i its SECOND element.  Stable radix sort of the 2 m
// endpoint rows with their slot numbers: inside a row the slots keep pair order (first elements before second ones) -- the
// fixed summation order of the gradient.  rowptr from a histogram + exclusive scan.  One call, no host synchronisation (the
// torch composition -- cat / argsort / bincount / cumsum / two gathers -- cost 0.8 ms per rebuild, mostly host time).
__global__ void pair_ends_kernel(const int32_t *__restrict__ pairs, int64_t m, uint32_t *__restrict__ keys, int32_t *__restrict__ slots,
                                 int64_t *__restrict__ counts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * m) return;
    const int32_t row = i < m ? pairs[2 * i] : pairs[2 * (i - m) + 1];
    keys[i] = (uint32_t)row;
    slots[i] = (int32_t)i;
    atomicAdd(reinterpret_cast<unsigned long long *>(counts + row), 1ull);
}

__global__ void pair_slots_kernel(const int32_t *__restrict__ pairs, int64_t m, const int32_t *__restrict__ sorted_slots,
                                  const int64_t *__restrict__ offs, int64_t n_rows, int32_t *__restrict__ rowptr,
                                  int32_t *__restrict__ other, int32_t *__restrict__ slot_pair) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n_rows) rowptr[i] = (int32_t)offs[i];
    if (i >= 2 * m) return;
    const int32_t sl = sorted_slots[i];
    const int64_t pr = sl < m ? sl : sl - m;
    other[i] = sl < m ? pairs[2 * pr + 1] : pairs[2 * pr];
    slot_pair[i] = (int32_t)pr;
}

int oea_pair_rows_build(const int32_t *pairs, int64_t m, int64_t n_rows, int32_t *rowptr, int32_t *other, int32_t *slot_pair,
                        void *stream) {
    OEA_REQUIRE(pairs && rowptr && other && slot_pair && m >= 0 && n_rows > 0 && 2 * m < 0x7fffffff, "arguments");
    Scratch scratch(oea::as_stream(stream));
    hipStream_t st = scratch.st;
    GB_ALLOC(counts, int64_t, n_rows + 1);
    GB_ALLOC(offs, int64_t, n_rows + 1);
    OEA_CHECK_HIP(hipMemsetAsync(counts, 0, sizeof(int64_t) * (size_t)(n_rows + 1), st));
    const int64_t n = 2 * m;
    GB_ALLOC(keys, uint32_t, n);
    GB_ALLOC(slots, int32_t, n);
    GB_ALLOC(sk, uint32_t, n);
    GB_ALLOC(ss, int32_t, n);
    if (n > 0) {
        pair_ends_kernel<<<blocks_for(n), 256, 0, st>>>(pairs, m, keys, slots, counts);
        GB_PRIM(rocprim::radix_sort_pairs(temp, tb, keys, sk, slots, ss, (size_t)n, 0, key_bits((uint64_t)n_rows), st));
    }
    GB_PRIM(rocprim::exclusive_scan(temp, tb, counts, offs, (int64_t)0, (size_t)(n_rows + 1), rocprim::plus<int64_t>(), st));
    pair_slots_kernel<<<blocks_for(std::max<int64_t>(n, n_rows + 1)), 256, 0, st>>>(pairs, m, ss, offs, n_rows, rowptr, other, slot_pair);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

}  // extern "C"
