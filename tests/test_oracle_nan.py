"""-m "not gpu": NaN VALUES (not NULLs) in the rolling extrema of MIDPOINT / MIDPRICE -- hand-derived from the reference's deques.

ref src/talib/overlap.rs:203-217 (midpoint maximum), :218-231 (midpoint minimum: its front is expired by the MAXIMUM deque's front index
-- quirk Q-MID -- so it never expires), :325-345 / :378-398 (midprice, no-bitmap branches).  A deque pops its back while `back <= value`
(`>=` for a minimum): no comparison with a NaN holds, so a NaN is never popped from the back and shields everything older; the front is
then the extremum of the values older than the oldest NaN in the window, the NaN itself once those have expired, and (minimum of
MIDPOINT, which never expires) whatever the minimum was when the first NaN arrived, for ever.  Worked through by hand below, row by row.
"""
import numpy as np

from oracle import pq_oracle as oracle

NAN = float("nan")


def same(a, b):
    a, b = np.asarray(a, float).reshape(-1), np.asarray(b, float)
    return bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def test_midpoint_nan_shields_the_older_maximum():
    # p = 3.  row 3 (NaN): max deque [(2,5),(3,NaN)] -> 5; row 4: [(2,5),(3,NaN),(4,2)] -> 5 (the 2 is hidden);
    # row 5: 5 expires, front = NaN; row 6: NaN expires -> [(5,7),(6,3)] -> 7.  min deque never expires: 1 throughout.
    x = np.array([1.0, 5.0, NAN, 2.0, 7.0, 3.0, 4.0])
    assert same(oracle.call("midpoint", x, timeperiod=3)[0], [1.0, 3.0, 3.0, 3.0, NAN, 4.0, 4.0])


def test_midpoint_nan_freezes_the_minimum_for_ever():
    # p = 2.  the minimum's deque is [(1,4),(2,NaN),...]: nothing behind the NaN reaches the front, 4 stays the minimum although 1 and 2
    # follow.  row 3: the maximum's front is the NaN; row 4: both 4 (max side) and the NaN have expired -> max 2, min 4 -> 3.
    x = np.array([4.0, NAN, 1.0, 2.0])
    assert same(oracle.call("midpoint", x, timeperiod=2)[0], [4.0, 4.0, NAN, 3.0])


def test_midpoint_first_value_nan():
    # the minimum's deque starts with the NaN and keeps it: every row is NaN
    x = np.array([NAN, 1.0, 2.0, 3.0])
    assert same(oracle.call("midpoint", x, timeperiod=2)[0], [NAN, NAN, NAN, NAN])


def test_midprice_nan_in_either_column():
    # p = 2.  highs: [3], [3,NaN] -> 3, 3 expires -> NaN, NaN expires -> 5.   lows: 2, 1, [1,NaN] -> 1, 1 expires -> NaN.
    h = np.array([3.0, NAN, 1.0, 5.0])
    l = np.array([2.0, 1.0, NAN, 4.0])
    assert same(oracle.call("midprice", h, l, timeperiod=2)[0], [2.5, 2.0, NAN, NAN])


def test_no_nan_is_the_plain_window_extremum():
    rng = np.random.default_rng(3)
    x = rng.random(200)
    for p in (1, 2, 7, 30):
        got = np.asarray(oracle.call("midprice", x, x, timeperiod=p)[0], float).reshape(-1)
        exp = np.array([(x[max(0, t - p + 1): t + 1].max() + x[max(0, t - p + 1): t + 1].min()) / 2.0 for t in range(200)])
        assert np.array_equal(got, exp)


def test_polars_rolling_extrema_ignore_nan_values():
    # decision D-14 (py-polars' rolling_min / rolling_max behind STOCH / STOCHF / STOCHRSI and Strategy.breakout): a NaN value is
    # ignored, whichever row of the frame holds it; only a frame of NaNs gives NaN; a NULL row makes the frame NULL
    x = np.array([1.0, NAN, 3.0, 2.0, NAN, NAN, 5.0, oracle.NULL, 4.0, 6.0])
    mx = np.asarray(oracle.call("rolling_max", x, window=2)[0], float).reshape(-1)
    mn = np.asarray(oracle.call("rolling_min", x, window=2)[0], float).reshape(-1)
    nullb = lambda a: np.ascontiguousarray(a).view(np.uint64) == np.uint64(oracle.NULL_BITS)
    assert nullb(mx).tolist() == [True, False, False, False, False, False, False, True, True, False]
    assert same(np.where(nullb(mx), 0.0, mx), [0.0, 1.0, 3.0, 3.0, 2.0, NAN, 5.0, 0.0, 0.0, 6.0])
    assert same(np.where(nullb(mn), 0.0, mn), [0.0, 1.0, 3.0, 2.0, 2.0, NAN, 5.0, 0.0, 0.0, 4.0])
    # STOCHF's fastk on the same semantics: frame 2, row 1 has a NaN high -> the maximum is the other row's
    h = np.array([2.0, NAN, 4.0]); l = np.array([1.0, 1.5, 3.0]); c = np.array([1.5, 1.75, 3.5])
    fk = np.asarray(oracle.call("stochf", h, l, c, fastk_period=2, fastd_period=1, fastd_matype=0)[0], float).reshape(-1)
    assert fk[1] == (1.75 - 1.0) * 100.0 / (2.0 - 1.0) and fk[2] == (3.5 - 1.5) * 100.0 / (4.0 - 1.5)
