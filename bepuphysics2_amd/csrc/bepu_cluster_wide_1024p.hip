// cluster_kernel<1024, false, true, false> as a one-sweep-per-launch unit (kPass: the exchanged solves of a scene split across GPUs; all 44 type ids, whole-island plans).
#define BEPU_VARIANT_THREADS 1024
#define BEPU_VARIANT_WIDE 1
#define BEPU_VARIANT_SHARED 0
#define BEPU_VARIANT_PASS 1
#include "bepu_cluster_variant.inc"
