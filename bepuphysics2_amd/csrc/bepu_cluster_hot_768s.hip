// cluster_kernel<768, *, false, SHARED>: the sixteen hot-path constraint types, split-island plans, twelve waves per cluster (168 VGPRs per wave).
#define BEPU_VARIANT_THREADS 768
#define BEPU_VARIANT_WIDE 0
#define BEPU_VARIANT_SHARED 1
#include "bepu_cluster_variant.inc"
