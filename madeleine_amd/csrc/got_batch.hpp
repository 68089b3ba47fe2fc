// got_batch.hpp -- up to GOT_MAXP independent GOT problems (one per stain: loss.py:278-302 is called once per stain, trainer.py:40-49)
// in ONE launch sequence.  Every GOT kernel takes a GotBatch by value; a workgroup finds its problem from blockIdx (got_pick in
// got_impl.inc).  Round 4: the four stains of a data-parallel rank used to run as four chains on four HIP streams -- a process has four
// hardware queues, and the first stream that is not ours (a prefetcher's, RCCL's) made two chains share one.  One stream,
// every launch covering all problems, needs no queue at all beyond the caller's.
#pragma once
#include <stdint.h>

namespace mdl {
constexpr int GOT_MAXP = 4;
struct GotProb {
    float* ws;                 // workspace of the problem (mdl_got_ws_bytes(k, n, d))
    const float* V;            // [k, n, d]
    const float* Q;            // [k, n, d]
    float* dV;                 // backward finish
    float* dQ;
    float* out;                // [2] = (WD sum, GWD sum)
    float* mm_out;             // [6] local extrema (may be NULL)
    const float* mm_in;        // [6] thresholds' extrema supplied by the caller (may be NULL: the local ones)
    const float* d_out;        // [2]
    float* d_mm;               // [6] (may be NULL)
    const float* d_mm_total;   // [6] (may be NULL)
    int k, n;
};
struct GotBatch {
    int np, d;
    GotProb p[GOT_MAXP];
};
}  // namespace mdl
