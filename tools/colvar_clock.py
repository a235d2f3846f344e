#!/usr/bin/env python3
"""tools/colvar_clock.py FILE -- summarise the per-wave phase clocks a -DSSDE_CV_CLOCK build of k_iso_colvar.hip writes
(SSDE_WAVE_CLOCK=FILE at create): core-clock cycles per row in staging / transition / row / barrier, by part (wave of the workgroup)."""
import sys
import numpy as np
d = np.loadtxt(sys.argv[1], comments="#")
for p in range(8):
    r = d[d[:, 0].astype(int) % 8 == p][:, 1:]
    r = r[r.sum(axis=1) > 0]
    print(f"part {p}: {len(r)} waves; cycles per row: staging {r[:,0].mean():.0f}, transition {r[:,1].mean():.0f}, row {r[:,2].mean():.0f}, barrier {r[:,3].mean():.0f}, total {r.sum(axis=1).mean():.0f}")
