// cluster_kernel<512, *, false, SHARED>: the sixteen hot-path constraint types, split-island plans (bodies shared between clusters).
#define BEPU_VARIANT_THREADS 512
#define BEPU_VARIANT_WIDE 0
#define BEPU_VARIANT_SHARED 1
#include "bepu_cluster_variant.inc"
