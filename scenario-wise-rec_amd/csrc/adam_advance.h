// One optimizer step of bookkeeping: the step counter, the bias-correction scalars of the step and their entry in the history ring
// the lazy row catch-up replays (csrc/adam.hip).  ONE thread runs it: swr_adam_advance's own launch, or -- as a rider -- the thread
// that finishes the fused loss launch (moe.hip bce_finish), which reads none of these fields: the 5 us launch leaves the step.
#pragma once
#include "common.h"

__device__ __forceinline__ void swr_adam_advance_body(swr_adam_hyper* h, float* hist, int64_t cap) {
    h->step += 1;
    const double t = static_cast<double>(h->step);
    const double bc1 = 1.0 - pow(h->beta1, t), bc2 = 1.0 - pow(h->beta2, t);
    h->step_size = static_cast<float>(h->lr / bc1);
    h->inv_bc2_sqrt = static_cast<float>(1.0 / sqrt(bc2));
    h->one_minus_b1 = static_cast<float>(1.0 - h->beta1);
    h->b2 = static_cast<float>(h->beta2);
    h->one_minus_b2 = static_cast<float>(1.0 - h->beta2);
    h->eps_f = static_cast<float>(h->eps);
    h->wd_f = static_cast<float>(h->weight_decay);
    if (hist) {                              // per-step scalars, replayed later by the lazy row catch-up: a RING of
        const uint32_t mask = static_cast<uint32_t>(cap - 1);      // `cap` (a power of two) steps -- the host flushes
        h->hist_mask = mask;                                        // every lazily updated table before a row can lag that far
        const int64_t slot = h->step & static_cast<int64_t>(mask);
        hist[2 * slot] = h->step_size;
        hist[2 * slot + 1] = h->inv_bc2_sqrt;
    }
}
