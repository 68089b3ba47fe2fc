// GOT kernels built for 1024-thread workgroups (16 waves x 128 VGPRs): more waves to hide the L2 latency of the
// per-iteration matrix passes -- the faster build for n <= 128 and for every backward sweep.
#define GOT_THREADS 1024
#define GOT_NS got1024
#include "got_impl.inc"
