// The FASTQ -> 2-bit parser core (skm_pack_core.h) over 16-byte blocks (compiled with -mssse3).
#define SKM_PACK_VARIANT 1
#include "skm_pack_core.h"
