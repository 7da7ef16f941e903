// Variants of the exact assignment kernel with 4 or 8 wavefronts per problem (large problems with too few instances to fill the chip; lds_mode 4 / 6:
// OC-SORT's first association with the row lists and the parallel scan steps). See lap_kernel.hip / lap_kernel_body.hpp.
#include "lap_kernel_body.hpp"
#define MOT_LAP_VARIANTS_WIDE(X) X(256, 0, 0, 1) X(256, 2, 0, 1) X(256, 3, 0, 1) X(512, 0, 0, 1) X(512, 2, 0, 1) X(512, 3, 0, 1) \
                                 X(512, 4, 0, 1) X(256, 4, 0, 1) X(512, 6, 0, 1) X(256, 6, 0, 1) \
                                 X(256, 5, 2, 0) X(256, 5, 2, 1)  /* behind the fast path, 257..512 columns: four wavefronts, two register-cached columns per lane */
MOT_LAP_TU_EXPORTS(lap_wide, MOT_LAP_VARIANTS_WIDE)
