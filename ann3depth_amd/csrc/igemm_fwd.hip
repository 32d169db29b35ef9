#define A3D_MODE 0
#include "igemm_inst.h"
