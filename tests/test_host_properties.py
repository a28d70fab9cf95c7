"""Property tests (hypothesis) of the host-side pieces that need no GPU: the exact
sampler against NumPy for arbitrary sizes, the MF level schedule's invariants, the
data-parallel shard bounds."""
import numpy as np
from hypothesis import given, settings, strategies as st

from relevance_factorizationmachine_amd.dist import shard_bounds
from relevance_factorizationmachine_amd.runtime import mf_schedule, mf_schedule_ex, sample_batches

SETTINGS = dict(max_examples=60, deadline=None)


@settings(**SETTINGS)
@given(n=st.integers(1, 6000), frac=st.floats(0.0, 1.0), epoch=st.integers(0, 2 ** 31 - 2),
       threads=st.integers(1, 4))
def test_sampler_equals_numpy_shuffle(n, frac, epoch, threads):
    """resample(..., replace=False, random_state=epoch) = first B of RandomState(epoch).shuffle(arange(n))
    (src/fm.py:72-79), for any n, B <= n, seed and thread count."""
    batch = max(1, int(round(frac * n)))
    got = sample_batches(n, batch, epoch, 2, n_threads=threads)
    for i in range(2):
        order = np.arange(n)
        np.random.RandomState(epoch + i).shuffle(order)
        np.testing.assert_array_equal(got[i], order[:batch])


@settings(**SETTINGS)
@given(data=st.data(), b=st.integers(1, 400), nu=st.integers(1, 40), ni=st.integers(1, 40),
       cap=st.integers(0, 50))
def test_mf_schedule_invariants(data, b, nu, ni, cap):
    """Every example runs one level after the latest earlier example sharing its user or
    item (so the sequential result of src/mf.py:97-108 is reproduced), a level never holds
    two examples of one user or item, and the record form carries the same schedule."""
    users = np.array(data.draw(st.lists(st.integers(0, nu - 1), min_size=b, max_size=b)), dtype=np.int64)
    items = np.array(data.draw(st.lists(st.integers(0, ni - 1), min_size=b, max_size=b)), dtype=np.int64)
    order, lptr = mf_schedule(users, items, nu, ni)
    assert sorted(order.tolist()) == list(range(b)) and lptr[0] == 0 and lptr[-1] == b
    level = np.empty(b, dtype=np.int64)
    for lv in range(len(lptr) - 1):
        grp = order[lptr[lv]:lptr[lv + 1]]
        assert len(grp) >= 1 and np.all(np.diff(grp) > 0)
        assert len(set(users[grp].tolist())) == len(grp) and len(set(items[grp].tolist())) == len(grp)
        level[grp] = lv
    last_u, last_i = {}, {}
    for s in range(b):
        want = max(last_u.get(users[s], -1), last_i.get(items[s], -1)) + 1
        assert level[s] == want
        last_u[users[s]] = last_i[items[s]] = want
    y = np.ones(b)
    p = np.full(b, 0.5)
    ex, lptr2, cache = mf_schedule_ex(users, items, y, p, nu, ni, cap)
    np.testing.assert_array_equal(lptr2, lptr)
    np.testing.assert_array_equal(ex["u"], users[order])
    np.testing.assert_array_equal(ex["i"], items[order])
    assert np.all(ex["ry"] == 2.0) and len(cache) <= cap
    counts = np.bincount(items, minlength=ni)
    assert all(counts[c] >= 2 for c in cache) and len(set(cache.tolist())) == len(cache)


@settings(**SETTINGS)
@given(batch=st.integers(0, 10 ** 6), world=st.integers(1, 64))
def test_shard_bounds_partition_the_batch(batch, world):
    """Contiguous shards that tile [0, batch) with sizes differing by at most one."""
    edges = [shard_bounds(batch, world, r) for r in range(world)]
    assert edges[0][0] == 0 and edges[-1][1] == batch
    sizes = []
    for (lo, hi), nxt in zip(edges, edges[1:] + [(batch, batch)]):
        assert lo <= hi == nxt[0]
        sizes.append(hi - lo)
    assert max(sizes) - min(sizes) <= 1
