// cluster_kernel<1024, *, true>: all 44 constraint types, 1024 threads per cluster (128 VGPRs per wave).
#define BEPU_VARIANT_THREADS 1024
#define BEPU_VARIANT_WIDE 1
#include "bepu_cluster_variant.inc"
