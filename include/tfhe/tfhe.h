/* tfhe/tfhe.h -- umbrella header (reference: include/Math.h:4, include/Client.h:4). */
#ifndef TFHE_HIP_TFHE_H
#define TFHE_HIP_TFHE_H
#include "tfhe_core.h"
#include "tfhe_gate_bootstrapping_functions.h"
#include "../tfhe_hip.h"
#endif
