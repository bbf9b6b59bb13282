// The FASTQ -> 2-bit parser core (skm_pack_core.h) over 32-byte blocks (compiled with -mavx2).
#define SKM_PACK_VARIANT 2
#include "skm_pack_core.h"
