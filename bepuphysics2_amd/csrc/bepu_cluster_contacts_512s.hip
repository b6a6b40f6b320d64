// cluster_kernel<512, *, false, SHARED>, contacts family: the eight convex contact manifolds only, split-island plans (the 100k-box pile, BASELINE.json configs[1]).
#define BEPU_VARIANT_THREADS 512
#define BEPU_VARIANT_WIDE 0
#define BEPU_VARIANT_CONTACTS 1
#define BEPU_VARIANT_SHARED 1
#include "bepu_cluster_variant.inc"
