/* tfhe/tfhe_io.h -- included by the reference (include/Math.h:5, include/Client.h:5)
 * but none of its export_/import_ functions is called there.
 * Client.h uses std::vector while including only tfhe headers, so in
 * C++ this header must pull in <vector> and <iostream> as upstream's does. */
#ifndef TFHE_HIP_TFHE_IO_H
#define TFHE_HIP_TFHE_IO_H
#include "tfhe_core.h"
#ifdef __cplusplus
#include <iostream>
#include <vector>
#endif
#include <stdio.h>
#endif
