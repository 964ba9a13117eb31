// Backward kernel argument blocks + launchers (see k_backward*.hip).
#pragma once
#include "dvs_kernels.h"
