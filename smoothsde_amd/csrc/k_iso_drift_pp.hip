// k_iso_drift_pp.hip -- the smooth-drift lanes whose design blocks are evaluated ON THE DEVICE from the blocks' covariates and
// piecewise-cubic tables (ssde_ppbasis: 8 B/row per block instead of 8 K): see k_iso_drift.inc
#define SSDE_DRIFT_PP 1
#include "k_iso_drift.inc"
