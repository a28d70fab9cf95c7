"""Feature assembly and the binary CSR cache.

Not-GPU part: the oracle's SciPy restatement and the product's host helpers against what
the reference's dataset preparers produced (tests/golden/make_golden_features.py), the
table builders against the pandas / scikit-learn calls the reference makes, the cache file
format.  GPU part (``-m gpu``): the device assembly (``rfm_csr_assemble_*``) against the same
fixtures and against SciPy's hstack / row indexing; integer / byte work, so bit-exact."""
import numpy as np
import pytest
from scipy import sparse

from conftest import load_golden, rel_err
from oracle import cpu_ref
from relevance_factorizationmachine_amd import features, synth


def _csr(g, prefix, shape):
    return sparse.csr_matrix((g[prefix + "data"], g[prefix + "indices"], g[prefix + "indptr"]), shape=shape)


def _same_csr(A, B):
    A, B = A.tocsr().copy(), B.tocsr().copy()
    A.sort_indices()
    B.sort_indices()
    assert A.shape == B.shape
    np.testing.assert_array_equal(A.indptr, B.indptr)
    np.testing.assert_array_equal(A.indices, B.indices)
    np.testing.assert_array_equal(A.data, B.data)


def test_oracle_and_host_helpers_match_the_reference_preparers():
    g = load_golden("feature_assembly")
    ut, it = sparse.csr_matrix(g["coat_user_feats"]), sparse.csr_matrix(g["coat_item_feats"])
    want = _csr(g, "coat_", tuple(g["coat_shape"]))
    _same_csr(cpu_ref.fm_features_coat(g["coat_user"], g["coat_item"], ut, it), want)
    # 1:1 negative sampling: oracle (NumPy's own permutation) and product (library shuffle)
    for fn in (cpu_ref.negative_sample, features.negative_sample):
        np.testing.assert_array_equal(fn(g["coat_label"], 12345), g["coat_sampled"])
        np.testing.assert_array_equal(fn(g["kuai_label"], 12345), g["kuai_sampled"])
        np.testing.assert_array_equal(fn(g["kuai_label"], 12345, 2), g["kuai_sampled_x2"])
    feats = _csr(g, "kuai_feat_", (5000, 50))
    _same_csr(feats[g["kuai_sampled"]], _csr(g, "kuai_pick_", (len(g["kuai_sampled"]), 50)))
    assert features.negative_sample(np.zeros(5), 1).shape == (0,)
    np.testing.assert_array_equal(features.negative_sample(np.array([1, 0, 1]), 3), [0, 2, 1])


def test_table_builders_match_pandas_and_sklearn():
    """dummies / standardise / multi_hot restate pd.get_dummies, StandardScaler + fillna(mean)
    and MultiLabelBinarizer as kuairec/_feature.py:90-135 calls them (that module itself is not
    importable here -- omegaconf -- so these three are checked against the libraries)."""
    rng = np.random.default_rng(4)
    n = 300
    cat = rng.integers(0, 7, size=n) * 3 - 4
    words = np.array(["b", "a", "zz", "c"])[rng.integers(0, 4, size=n)]
    num = rng.standard_normal((n, 4)) * np.array([1.0, 1e3, 1e-3, 5.0]) + np.array([0.0, 50.0, 1.0, -2.0])
    num[rng.random((n, 4)) < 0.1] = np.nan
    num[:, 3] = 2.5  # a constant column: scaled by 1, not by 0
    tags = [list(rng.choice(12, size=rng.integers(0, 4), replace=False)) for _ in range(n)]
    want = cpu_ref.feature_table([cat, words], num, [tags])
    np.testing.assert_array_equal(features.dummies(cat), want[0])
    np.testing.assert_array_equal(features.dummies(words), want[1])
    got = features.standardise(num)
    assert not np.isnan(got).any()
    np.testing.assert_allclose(got, want[2], rtol=1e-12, atol=1e-14)
    np.testing.assert_array_equal(features.multi_hot(tags), want[3])
    T = features.table(features.dummies(cat), got, features.multi_hot(tags))
    assert T.shape == (n, 7 + 4 + 12) and T.nnz < n * 23  # exact zeros are not stored


def test_binary_csr_cache_round_trip(tmp_path):
    train, _ = synth.make_log("coat", "FM", "IPS", seed=0)
    X = train["features"]
    path = str(tmp_path / "train.rfmcsr")
    features.save_csr(path, X, train["labels"], train["pscores"])
    for mmap in (True, False):
        d = features.load_csr(path, mmap=mmap)
        assert d["shape"] == X.shape
        np.testing.assert_array_equal(d["indptr"], X.indptr)
        np.testing.assert_array_equal(d["indices"], X.indices)
        np.testing.assert_array_equal(d["values"], X.data)
        np.testing.assert_array_equal(d["labels"], train["labels"])
        np.testing.assert_array_equal(d["pscores"], train["pscores"])
        assert d["indptr"].dtype == np.int64 and d["indices"].dtype == np.int32 and d["values"].dtype == np.float64
    for name in ("indptr", "indices", "values", "labels", "pscores"):  # every section 64-byte aligned
        assert features.load_csr(path)[name].offset % 64 == 0
    features.save_csr(path, X)  # without labels / pscores, and an empty matrix
    assert "labels" not in features.load_csr(path)
    features.save_csr(path, sparse.csr_matrix((0, 5)))
    assert features.load_csr(path)["shape"] == (0, 5)
    features.save_csr(path, X, train["labels"])
    blob = open(path, "rb").read()
    open(path, "wb").write(blob[:-100])
    with pytest.raises(ValueError, match="truncated"):
        features.load_csr(path)
    open(path, "wb").write(b"not a cache" * 10)
    with pytest.raises(ValueError, match="RFMCSR01"):
        features.load_csr(path)
    with pytest.raises(ValueError, match="one value per row"):
        features.save_csr(path, X, train["labels"][:-1])


# ---------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def rt():
    from relevance_factorizationmachine_amd.runtime import Runtime
    return Runtime.get()


@pytest.mark.gpu
def test_device_assembly_matches_the_reference_preparers(rt):
    g = load_golden("feature_assembly")
    ut, it = sparse.csr_matrix(g["coat_user_feats"]), sparse.csr_matrix(g["coat_item_feats"])
    got = features.fm_features_coat(rt, g["coat_user"], g["coat_item"], ut, it)
    assert got.shape == tuple(g["coat_shape"]) and got.nnz == len(g["coat_data"])
    _same_csr(got.to_scipy(), _csr(g, "coat_", tuple(g["coat_shape"])))
    feats = _csr(g, "kuai_feat_", (5000, 50))
    _same_csr(features.take_rows(rt, feats, g["kuai_sampled"]).to_scipy(),
              _csr(g, "kuai_pick_", (len(g["kuai_sampled"]), 50)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(3))
def test_device_assembly_matches_scipy_hstack(rt, seed):
    """KuaiRec-shaped design matrix from random tables -- rows without entries, stored zeros,
    ids at both ends of their ranges -- against SciPy's hstack of the same pieces."""
    rng = np.random.default_rng(seed)
    nu, ni, n = [50, 1, 700][seed], [30, 400, 2][seed], [2000, 333, 5000][seed]
    users, items = rng.integers(0, nu, size=n), rng.integers(0, ni, size=n)
    users[:2], items[:2] = [0, nu - 1], [ni - 1, 0]
    utab = sparse.random(nu, 11, density=0.3, format="csr", random_state=rng)
    itab = sparse.random(ni, 7, density=0.5, format="csr", random_state=rng)
    if itab.nnz:
        itab.data[0] = 0.0  # a stored zero travels as it is (hstack keeps it too)
    inter = sparse.random(n, 3, density=0.6, format="csr", random_state=rng)
    got = features.fm_features_kuairec(rt, users, items, nu, ni, inter, utab, itab)
    want = cpu_ref.fm_features_kuairec(users, items, nu, ni, inter, utab, itab)
    assert got.shape == want.shape == (n, nu + ni + 3 + 11 + 7)
    _same_csr(got.to_scipy(), want)
    # the assembled matrix feeds the training path as it is
    sub = rng.permutation(n)[: n // 2]
    _same_csr(features.take_rows(rt, got, sub).to_scipy(), want[sub])
    with pytest.raises(ValueError, match="outside"):
        features.fm_features_kuairec(rt, users + 1, items, nu, ni, inter, utab, itab)
    with pytest.raises(ValueError):
        features.assemble(rt, n, [features.Rows(utab)])  # per-row segment of the wrong length
    empty = features.take_rows(rt, want, np.zeros(0, np.int64))
    assert empty.shape == (0, want.shape[1]) and empty.nnz == 0


@pytest.mark.gpu
def test_fit_from_the_binary_cache_equals_fit_from_scipy(rt, tmp_path):
    import relevance_factorizationmachine_amd as pkg
    train, val = synth.make_log("coat", "FM", "IPS", seed=0)
    path = str(tmp_path / "train.rfmcsr")
    features.save_csr(path, train["features"], train["labels"], train["pscores"])
    dev, labels, pscores = features.load_csr_to_device(rt, path)
    kw = dict(estimator="IPS", n_epochs=5, n_factors=8, lr=1e-4, batch_size=500, seed=12345,
              n_features=train["features"].shape[1])
    a = pkg.FactorizationMachines(**kw)
    a.hot_min_count = -1  # fixed-order sums: bitwise comparable
    ta, va = a.fit({"features": dev, "labels": labels, "pscores": pscores}, val)
    b = pkg.FactorizationMachines(**kw)
    b.hot_min_count = -1
    tb, vb = b.fit(train, val)
    np.testing.assert_array_equal(a.V(), b.V())
    assert ta == tb and va == vb
    ref = cpu_ref.fm_fit(train, val, **{k: v for k, v in kw.items() if k not in ("estimator", "n_features")})
    assert rel_err(a.V(), ref["V"]) < 1e-9
