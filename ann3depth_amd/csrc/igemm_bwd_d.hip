#define A3D_MODE 1
#include "igemm_inst.h"
