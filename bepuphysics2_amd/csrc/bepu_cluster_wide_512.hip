// cluster_kernel<512, *, true>: all 44 constraint types, 512 threads per cluster (256 VGPRs per wave).
#define BEPU_VARIANT_THREADS 512
#define BEPU_VARIANT_WIDE 1
#include "bepu_cluster_variant.inc"
