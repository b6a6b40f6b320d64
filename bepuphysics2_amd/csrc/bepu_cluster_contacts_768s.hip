// cluster_kernel<768, *, false, SHARED>, contacts family: split-island plans at twelve waves per cluster.
#define BEPU_VARIANT_THREADS 768
#define BEPU_VARIANT_WIDE 0
#define BEPU_VARIANT_CONTACTS 1
#define BEPU_VARIANT_SHARED 1
#include "bepu_cluster_variant.inc"
