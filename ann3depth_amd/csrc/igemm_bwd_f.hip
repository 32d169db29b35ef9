#define A3D_MODE 2
#include "igemm_inst.h"
