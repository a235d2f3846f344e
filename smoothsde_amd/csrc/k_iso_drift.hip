// k_iso_drift.hip -- the smooth-drift lanes with STREAMED design columns (tile channels, 8 K B/row): see k_iso_drift.inc
#define SSDE_DRIFT_PP 0
#include "k_iso_drift.inc"
