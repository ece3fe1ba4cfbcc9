// stub: the kernels need nothing from torch; the ATen host code is restated in ref_driver.cpp
#pragma once
#include "../cuda_serial_shim.h"
