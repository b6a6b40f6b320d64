// cluster_kernel<512, *, false, SHARED> with the non-temporal row policy: split-island plans on boxes where that is the faster one.
#define BEPU_VARIANT_THREADS 512
#define BEPU_VARIANT_WIDE 0
#define BEPU_VARIANT_SHARED 1
#define BEPU_VARIANT_NT 1
#include "bepu_cluster_variant.inc"
