// cluster_kernel<512, false, false, true> with the momentum-conserving angular integration modes compiled in (the sixteen hot-path constraint types, split-island plans).
#define BEPU_VARIANT_THREADS 512
#define BEPU_VARIANT_WIDE 0
#define BEPU_VARIANT_SHARED 1
#define BEPU_VARIANT_CONSERVING 1
#include "bepu_cluster_variant.inc"
