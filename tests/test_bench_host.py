"""Host logic of bench.py (no GPU): which tracks a rank owns under strong / weak scaling, the C5 batch's ragged
lengths and missing rows as functions of the seed and of GLOBAL indices (so that every cut of the batch is the same
batch).  Independence of tracks: /root/reference/src/nllk/nllk_ctcrw.hpp:196-200, 234."""
import numpy as np
import torch

import bench


def test_strong_shards_partition_the_batch_and_weak_shards_replicate_its_size():
    for M in (10_000, 1250, 7, 100_000):
        for world in (1, 2, 3, 4, 8):
            cuts = [bench.shard_range(M, world, r, "strong") for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == M
            assert all(cuts[r][1] == cuts[r + 1][0] for r in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1
            wk = [bench.shard_range(M, world, r, "weak") for r in range(world)]
            assert all(b - a == M for a, b in wk) and wk[-1][1] == M * world


def test_c5_lengths_and_missing_rows_do_not_depend_on_the_cut():
    T, M = 64, 50
    lens = bench.track_lengths(M, T, seed=7)
    assert lens.min() >= T // 2 and lens.max() <= T and np.array_equal(lens, bench.track_lengths(M, T, seed=7))
    row0 = np.concatenate([[0], np.cumsum(lens)])
    n = int(row0[-1])
    whole = torch.zeros(n, 2, dtype=torch.float64)
    k = bench.missing_rows(whole, 0, torch.as_tensor(row0[:-1]))
    frac = k / n
    assert 0.02 < frac < 0.09
    isn = torch.isnan(whole)
    assert not isn[torch.as_tensor(row0[:-1])].any()                       # first rows stay observed
    only0 = (isn[:, 0] & ~isn[:, 1]).sum().item()
    both = (isn[:, 0] & isn[:, 1]).sum().item()
    assert only0 > 0 and both > 0 and (isn[:, 1] & ~isn[:, 0]).sum().item() == 0
    # the same rows when the batch is generated in two shards
    m = 20
    a = torch.zeros(int(row0[m]), 2, dtype=torch.float64)
    b = torch.zeros(n - int(row0[m]), 2, dtype=torch.float64)
    bench.missing_rows(a, 0, torch.as_tensor(row0[:m]))
    bench.missing_rows(b, int(row0[m]), torch.as_tensor(row0[m:-1] - row0[m]))
    assert torch.equal(torch.isnan(torch.cat([a, b])), isn)
