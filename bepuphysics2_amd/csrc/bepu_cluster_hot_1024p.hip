// cluster_kernel<1024, false, false, false> as a one-sweep-per-launch unit (kPass: the exchanged solves of a scene split across GPUs; the sixteen hot-path constraint types, whole-island plans).
#define BEPU_VARIANT_THREADS 1024
#define BEPU_VARIANT_WIDE 0
#define BEPU_VARIANT_SHARED 0
#define BEPU_VARIANT_PASS 1
#include "bepu_cluster_variant.inc"
