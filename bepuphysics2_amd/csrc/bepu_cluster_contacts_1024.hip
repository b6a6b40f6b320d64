// cluster_kernel<1024, *, false>, contacts family: whole-island plans of scenes made of convex contact manifolds alone (box stacks, pyramids).
#define BEPU_VARIANT_THREADS 1024
#define BEPU_VARIANT_WIDE 0
#define BEPU_VARIANT_CONTACTS 1
#include "bepu_cluster_variant.inc"
