// cluster_kernel<512, false, true, true> with the momentum-conserving angular integration modes compiled in (all 44 type ids, split-island plans).
#define BEPU_VARIANT_THREADS 512
#define BEPU_VARIANT_WIDE 1
#define BEPU_VARIANT_SHARED 1
#define BEPU_VARIANT_CONSERVING 1
#include "bepu_cluster_variant.inc"
