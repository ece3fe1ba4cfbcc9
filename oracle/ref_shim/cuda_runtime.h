// stub for the host-serial build of the reference kernels (see cuda_serial_shim.h)
#pragma once
#include "cuda_serial_shim.h"
