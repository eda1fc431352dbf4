// The one-workgroup two-rows-per-lane kernels (pcg_single_f32x2_kernel, pcg_single_f32h_kernel, pcg_single_f64m_kernel,
// gato_pcg_resident.hip) and their launcher: the same source, compiled as a third translation unit so that the parts build in
// parallel (the whole file in one unit takes five minutes).
#define GATO_RESIDENT_SINGLE_PART 1
#include "gato_pcg_resident.hip"
