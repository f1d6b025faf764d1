/* gdyn_dev.h -- developer-only entry points of libgdyn_dev.so (make -C csrc dev, -DGD_DEV).  Not part of the ABI
 * in include/gdyn.h and absent from the product library libgdyn.so. */
#ifndef GDYN_DEV_H
#define GDYN_DEV_H
#include "../../include/gdyn.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Times `n` back-to-back launches of one piece of the path on the CURRENT state with HIP events, without advancing
 * the trajectory.  what = 0: full neighbour-list build; 1: step kernel (output discarded); 10.. / 30..: section stamps
 * of the -DGD_ABL=30 / 34 timing builds.  Returns mean ms per launch. */
int gd_debug_bench(gd_system *sys, int what, int n, double *mean_ms);
#ifdef __cplusplus
}
#endif
#endif
