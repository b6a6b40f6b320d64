// cluster_kernel<1024, *, false, false> with non-temporal constraint-row loads: the second row-load policy of the whole-island schedule (box classes).
#define BEPU_VARIANT_THREADS 1024
#define BEPU_VARIANT_WIDE 0
#define BEPU_VARIANT_NT 1
#include "bepu_cluster_variant.inc"
