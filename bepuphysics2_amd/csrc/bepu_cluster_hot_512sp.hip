// cluster_kernel<512, false, false, true> as a one-sweep-per-launch unit (kPass: the exchanged solves of a scene split across GPUs; the sixteen hot-path constraint types, split-island plans).
#define BEPU_VARIANT_THREADS 512
#define BEPU_VARIANT_WIDE 0
#define BEPU_VARIANT_SHARED 1
#define BEPU_VARIANT_PASS 1
#include "bepu_cluster_variant.inc"
