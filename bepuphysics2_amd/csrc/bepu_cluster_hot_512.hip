// cluster_kernel<512, *, false>: the sixteen hot-path constraint types, 512 threads per cluster (256 VGPRs per wave).
#define BEPU_VARIANT_THREADS 512
#define BEPU_VARIANT_WIDE 0
#include "bepu_cluster_variant.inc"
