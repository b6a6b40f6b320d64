// cluster_kernel<1024, false, true, false> with the momentum-conserving angular integration modes compiled in (all 44 type ids, whole-island plans).
#define BEPU_VARIANT_THREADS 1024
#define BEPU_VARIANT_WIDE 1
#define BEPU_VARIANT_SHARED 0
#define BEPU_VARIANT_CONSERVING 1
#include "bepu_cluster_variant.inc"
