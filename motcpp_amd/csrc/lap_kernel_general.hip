// Variants of the exact assignment kernel for the general association measures (GIoU, DIoU, CIoU, centroid: cost flavour 2). See lap_kernel.hip /
// lap_kernel_body.hpp.
#include "lap_kernel_body.hpp"
#define MOT_LAP_VARIANTS_GENERAL(X) X(64, 0, 0, 2) X(64, 2, 0, 2) X(64, 3, 0, 2) X(256, 0, 0, 2) X(256, 2, 0, 2) X(256, 3, 0, 2)
MOT_LAP_TU_EXPORTS(lap_general, MOT_LAP_VARIANTS_GENERAL)
