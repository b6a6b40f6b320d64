// cluster_kernel<768, *, false>: the sixteen hot-path constraint types, 768 threads per cluster (168 VGPRs per wave).
#define BEPU_VARIANT_THREADS 768
#define BEPU_VARIANT_WIDE 0
#include "bepu_cluster_variant.inc"
