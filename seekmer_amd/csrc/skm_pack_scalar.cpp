// The FASTQ -> 2-bit parser core (skm_pack_core.h) over single characters: the reference for the vector variants and the fallback on a CPU without SSSE3.
#define SKM_PACK_VARIANT 0
#include "skm_pack_core.h"
