// GOT kernels built for 512-thread workgroups (8 waves x 256 VGPRs): the faster forward for n > 128.
#define GOT_THREADS 512
#define GOT_NS got512
#include "got_impl.inc"
