"""The oracle's quantification stage against numpy itself (the reference's EM
is a numpy program, seekmer/infer.py:133-168; numpy is present in this image)."""
import numpy as np


def _numpy_em_step(x, l, class_map, class_count, n):
    w = x[class_map[1]]
    inner = np.bincount(class_map[0], weights=w, minlength=class_count.size) / class_count
    with np.errstate(divide='ignore', invalid='ignore'):
        new = np.bincount(class_map[1], weights=w / inner[class_map[0]], minlength=l.size) / l / n
    new[new != new] = 0
    return new


def _random_problem(rng, n_tx, n_classes):
    sizes = rng.integers(1, 7, n_classes)
    cls = np.repeat(np.arange(n_classes), sizes)
    tx = rng.integers(0, n_tx, cls.size)
    class_map = np.vstack([cls, tx]).astype(np.int64)
    class_count = rng.integers(1, 50, n_classes).astype('f8')
    l = rng.uniform(50, 3000, n_tx)
    return class_map, class_count, l


def test_pairwise_sum_matches_numpy(oracle):
    rng = np.random.default_rng(0)
    for n in (0, 1, 7, 8, 9, 127, 128, 129, 1000, 4097, 8192, 8193, 16385, 65537, 190402):
        a = rng.uniform(0, 1, n) * 10.0 ** rng.integers(-8, 8, n)
        assert oracle.pairwise_sum(a) == float(a.sum()), n


def test_effective_lengths_matches_numpy(oracle):
    rng = np.random.default_rng(1)
    fld = np.zeros(2000, dtype=np.int64)
    fld[rng.integers(100, 400, 500)] += 1
    length = rng.integers(30, 5000, 300).astype('f8')
    expected = np.zeros(length.shape, dtype='f8')
    p = fld / fld.sum()
    for i in range(p.size):
        expected += (length - i).clip(min=1) * p[i]
    np.testing.assert_array_equal(oracle.effective_lengths(fld, length), expected)
    hmean = fld.sum() / (fld[1:].astype('f8') / np.arange(1, 2000)).sum()
    assert oracle.harmonic_mean_fragment_length(fld) == hmean


def test_em_steps_match_numpy_bitwise(oracle):
    rng = np.random.default_rng(2)
    class_map, class_count, l = _random_problem(rng, 400, 900)
    x0 = 1.0 / l
    x0 /= x0.sum()
    n = class_count.sum()
    x = x0.copy()
    _, _, trace = oracle.em(x0, l, class_map, class_count, fixed_iters=6, trace_iters=(1, 2, 5, 6))
    for it in range(1, 7):
        x = _numpy_em_step(x, l, class_map, class_count, n)
        if it in (1, 2, 5, 6):
            np.testing.assert_array_equal(trace[(1, 2, 5, 6).index(it)], x)


def test_em_stopping_rule_matches_numpy(oracle):
    rng = np.random.default_rng(3)
    class_map, class_count, l = _random_problem(rng, 200, 500)
    x0 = 1.0 / l
    x0 /= x0.sum()
    n = class_count.sum()
    old, x, iters = x0, _numpy_em_step(x0, l, class_map, class_count, n), 1
    while (np.absolute(x - old) / x)[x > 1e-8].max() > 0.01:
        old, x = x, _numpy_em_step(x, l, class_map, class_count, n)
        iters += 1
    got, got_iters = oracle.em(x0, l, class_map, class_count)
    assert got_iters == iters
    np.testing.assert_array_equal(got, x)
    tpm, _ = oracle.quantify(l, class_map, class_count)
    ref = x.copy()
    ref /= ref.sum() / 1000000
    ref[ref < 0.001] = 0
    ref /= ref.sum() / 1000000
    np.testing.assert_array_equal(tpm, ref)


def test_em_zero_count_and_nan_rules(oracle):
    """class_count 0 (bootstrap) -> inf inner -> zero terms; NaN -> 0."""
    class_map = np.array([[0, 0, 1, 2, 2], [0, 1, 1, 2, 3]], dtype=np.int64)
    class_count = np.array([5.0, 0.0, 7.0])
    l = np.array([100.0, 200.0, 300.0, 400.0])
    x0 = np.full(4, 0.25)
    n = class_count.sum()
    x = _numpy_em_step(x0, l, class_map, class_count, n)
    got, _, trace = oracle.em(x0, l, class_map, class_count, fixed_iters=1, trace_iters=(1,))
    np.testing.assert_array_equal(trace[0], x)
