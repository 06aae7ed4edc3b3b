// step_plan.h -- the "gathered sum" plan of a translational epoch (round 6), shared by step_plan.hip (build) and triple_step.hip (use).
//
// Why: device-scope fp32 atomics on gfx950 execute MEMORY-SIDE (every request leaves the XCD's L2: TCC_EA0_ATOMIC == TCC_ATOMIC,
// profiles/r06_step_wave_pmc_atomics.csv) at ~20 G requests of 64 B per second -- 24 of triple_wave's 54 us at the 100K shape, most
// of them the rows of the positives' OWN head and tail.  Which rows those are is known before the step runs: the epoch's positives
// and its negatives are drawn ahead.  So the epoch is planned once (on a side stream, behind the previous epoch):
//   * every positive p of step s whose k negatives are corruptions of it on ONE side (the sampler's output but for rounds after a
//     collision, batch.py:101-107) contributes two rows: A_p = sum of dL/d(delta) over its active triples, B_p = the positive's own
//     term.  Tail side: head += A, tail -= B (and relation += A); head side: head += B, tail -= A (relation += A).  triple_wave
//     writes A_p, B_p with PLAIN stores into contrib[2 p_local + {0, 1}] instead of two rows of atomics;
//   * the plan = the (step, row)-sorted list of those references: unique (step, row) keys, their entry ranges, and per entry
//     sign << 31 | slot.  A stable radix sort keeps a row's entries in batch order: the optimiser kernel adds them in THAT order
//     (apply_rows_plan) -- no atomics, no order dependence for these rows, and it visits exactly the rows that received gradient
//     instead of streaming the whole table past the touched flags;
//   * relation rows (a few hundred, shared by the whole batch), the corrupted rows of ACTIVE negatives, positives outside the rule
//     (mixed sides, foreign entries) and HUB rows (more than kPlanHubEntries references in a step) keep the atomic scratch + touched
//     flags and the flag-driven optimiser pass.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace oea {

struct StepPlanView {
    uint64_t *keys_a, *keys_b;     // [2N]  (step << row_bits | row) of every reference: emitted / sorted
    uint32_t *vals_a, *vals_b;     // [2N]  sign << 31 | slot, slot = 2 * (index inside the batch) + {0: A, 1: B}
    uint64_t *ukeys;               // [2N]  distinct keys, ascending
    uint32_t *ucount;              // [2N + 1]
    uint32_t *uoff;                // [2N + 1] first entry of every distinct key (exclusive scan of ucount)
    int32_t *n_unique;             // [1]
    int32_t *step_first;           // [steps + 1] index of a step's first distinct key
    uint32_t *pflags;              // [N] per positive of the epoch: bit 0 / 1 = its head / tail row is a HUB of its step (more than
                                   //     kPlanHubEntries references): that row takes the positive's gradient through the atomic scratch
    uint4 *recs;                   // [2N] per distinct key: {row, first entry, entries, value of the first entry} -- one load instead of three
    uint8_t *inplan;               // [steps][n_ent] 1 = the row is summed from the plan in that step (listed and not a hub): the flag-driven
                                   //     part of the optimiser kernel leaves it alone
    float *contrib;                // [2 * max_batch][ld]
    void *temp;
    size_t temp_bytes;
    int row_bits;
};

// A row with more references in one step than this is left to the atomic scratch and the flag-driven optimiser pass: summing a hub's
// hundreds of contribution rows is a serial chain for ONE lane group (measured: 4.9 ms for the hottest step at the 100K shape), and
// the memory-side adder serialises same-row atomics at ~25 ns each whatever the design.
constexpr uint32_t kPlanHubEntries = 8;

// layout of a plan workspace for an epoch of n_total positives in `steps` batches of at most max_batch rows; base == nullptr: size only
size_t step_plan_layout(int64_t n_total, int32_t steps, int64_t max_batch, int64_t n_ent, int32_t ld, void *base, StepPlanView *v);

}  // namespace oea
