// cluster_kernel<1024, *, false>: the sixteen hot-path constraint types, 1024 threads per cluster (128 VGPRs per wave).
#define BEPU_VARIANT_THREADS 1024
#define BEPU_VARIANT_WIDE 0
#include "bepu_cluster_variant.inc"
