// The DPP-row instantiations of the resident PCG kernel (template parameter DR of pcg_resident_kernel, gato_pcg_resident.hip):
// the same source, compiled as a second translation unit so that the two halves build in parallel.
#define GATO_RESIDENT_DPP_PART 1
#include "gato_pcg_resident.hip"
