// GOT kernels built for 1024-thread workgroups (16 waves x 128 VGPRs): enough waves to hide the L2 latency of the
// per-iteration matrix passes.  got_impl.inc is parametrised by GOT_THREADS / GOT_NS so that other geometries can be
// built side by side for A/B measurements (tools/got_ab.py).
#define GOT_THREADS 1024
#define GOT_NS got1024
#include "got_impl.inc"
