// cluster_kernel<1024, false, false, false> with the momentum-conserving angular integration modes compiled in (the sixteen hot-path constraint types, whole-island plans).
#define BEPU_VARIANT_THREADS 1024
#define BEPU_VARIANT_WIDE 0
#define BEPU_VARIANT_SHARED 0
#define BEPU_VARIANT_CONSERVING 1
#include "bepu_cluster_variant.inc"
