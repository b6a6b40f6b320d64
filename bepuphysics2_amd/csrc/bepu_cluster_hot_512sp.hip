// cluster_kernel<512, *, false, SHARED, PREFETCH>: the sixteen hot-path types, split-island plans, rows of the next work item prefetched into LDS (RowAhead).
#define BEPU_VARIANT_THREADS 512
#define BEPU_VARIANT_WIDE 0
#define BEPU_VARIANT_SHARED 1
#define BEPU_VARIANT_PREFETCH 1
#include "bepu_cluster_variant.inc"
