// cluster_kernel<768, *, true>: all 44 constraint types, 768 threads per cluster (168 VGPRs per wave).
#define BEPU_VARIANT_THREADS 768
#define BEPU_VARIANT_WIDE 1
#include "bepu_cluster_variant.inc"
