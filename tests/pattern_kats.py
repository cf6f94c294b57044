"""Hand-built known-answer candles for all 61 recognisers of src/talib/pattern.rs ("one positive + one negative
per cdl*").  Every sequence below was derived BY HAND from the Rust conditions (file:line cited per entry) -- not from the C
oracle and not from the HIP kernel; tests/test_oracle_kat3.py holds the oracle to them, tests/test_gpu_parity.py feeds the
concatenated positives to the HIP path so that every satisfiable recogniser fires on the GPU.

Candle = (open, high, low, close).  A sequence is exactly look-back + 1 rows; the expected value is the LAST row's output.
Index convention of the source: in a k-row look-back, o1/c1 is the OLDEST row (i - k), the unnumbered o/c the current row.
Predicates (pattern.rs:2067-2143): long_body |o-c| > 0.05*(o+c)*0.5; short_body < 0.1*(o+c)*0.5; doji <= 0.005*(o+c)*0.5;
long_*_shadow > 2*body; short_* < 0.5*body; vshort_* < 0.1*body; vlong_dn > 3*body; near |a-b| < 0.01*(h+l)*0.5; equal < 0.001*(h+l)*0.5
(h, l of the CURRENT row).

cdl2crows is UNSATISFIABLE as written (pattern.rs:30-33): bear2 needs c2 < o2 while open_in2 = (o > o2) && (o < c2) needs
o2 < o < c2.  It can never fire; it has negatives only and UNSAT names it.
"""

# name -> list of (candles, expected value at the last row, what the case shows)
KATS = {
    # :10-40 -- unsatisfiable; the classic two-crows shape (third opens inside the second body) does not fire
    "cdl2crows": [
        ([(10, 11.1, 9.9, 11), (12, 12.1, 11.4, 11.5), (11.8, 11.9, 10.4, 10.5)], 0, "classic shape: o inside body 2, but the source asks o > o2 && o < c2"),
        ([(10, 11.1, 9.9, 11), (12, 12.1, 11.4, 11.5), (12.2, 12.3, 10.4, 10.5)], 0, "o > o2 but then o < c2 fails"),
    ],
    # :43-73
    "cdl3blackcrows": [
        ([(20, 20.1, 18.4, 18.5), (19.5, 19.6, 17.4, 17.5), (18.5, 18.6, 16.4, 16.5)], -100, "three long bears, each opening inside the prior body"),
        ([(20, 20.1, 18.4, 18.5), (19.5, 19.6, 17.4, 17.5), (19.6, 19.7, 16.4, 16.5)], 0, "third opens above the second open"),
    ],
    # :76-111
    "cdl3inside": [
        ([(12, 12.1, 9.9, 10), (10.5, 11.6, 10.4, 11.5), (11.4, 12.6, 11.3, 12.5)], 100, "bear long, bull inside, bull closing above o1"),
        ([(10, 12.1, 9.9, 12), (11.5, 11.6, 10.4, 10.5), (10.6, 10.7, 9.4, 9.5)], -100, "mirror"),
        ([(12, 12.1, 9.9, 10), (10.5, 11.6, 10.4, 11.5), (11.4, 12.0, 11.3, 11.9)], 0, "third close 11.9 not above o1 = 12"),
    ],
    # :114-157
    "cdl3linestrike": [
        ([(20, 20.1, 18.9, 19), (19.5, 19.6, 17.9, 18), (18.5, 18.6, 16.9, 17), (16.5, 20.6, 16.4, 20.5)], 100, "three stepping bears, bull strike from below c3 to above o1"),
        ([(10, 11.1, 9.9, 11), (10.5, 12.1, 10.4, 12), (11.5, 13.1, 11.4, 13), (13.5, 13.6, 9.4, 9.5)], -100, "mirror"),
        ([(20, 20.1, 18.9, 19), (19.5, 19.6, 17.9, 18), (18.5, 18.6, 16.9, 17), (16.5, 20.0, 16.4, 19.9)], 0, "strike closes 19.9 < o1 = 20"),
    ],
    # :160-191
    "cdl3outside": [
        ([(11, 11.1, 9.9, 10), (9.9, 11.3, 9.8, 11.2), (11.1, 12.1, 11, 12)], 100, "bull engulfs bear, then higher close"),
        ([(10, 11.1, 9.9, 11), (11.1, 11.2, 9.7, 9.8), (9.9, 10, 8.9, 9)], -100, "mirror"),
        ([(11, 11.1, 9.9, 10), (9.9, 11.3, 9.8, 11.2), (10.5, 11.2, 10.4, 11.1)], 0, "third close 11.1 < c2 = 11.2"),
    ],
    # :194-231
    "cdl3starsinsouth": [
        ([(10, 10.1, 8, 9.4), (9.9, 10, 8.5, 9.5), (9.7, 9.8, 9.0, 9.6)], 100, "long bear with long lower shadow, bear with higher low and close, small bear inside"),
        ([(10, 10.1, 8, 9.4), (9.9, 10, 8.5, 9.5), (9.7, 9.8, 8.4, 9.6)], 0, "third low 8.4 below l2 = 8.5"),
    ],
    # :234-265
    "cdl3whitesoldiers": [
        ([(10, 11.1, 9.9, 11), (10.5, 12.1, 10.4, 12), (11.5, 13.1, 11.4, 13)], 100, "three long bulls opening inside the prior body"),
        ([(10, 11.1, 9.9, 11), (10.5, 12.1, 10.4, 12), (12.1, 13.7, 12.0, 13.6)], 0, "third opens above c2"),
    ],
    # :268-306
    "cdlabandonedbaby": [
        ([(10, 10.1, 8.9, 9), (8.5, 8.6, 8.4, 8.52), (8.7, 9.3, 8.65, 9.2)], 100, "bear long, doji gapped below l1, bull gapped above h2"),
        ([(9, 10.1, 8.9, 10), (10.5, 10.6, 10.4, 10.52), (10.3, 10.35, 9.7, 9.8)], -100, "mirror"),
        ([(10, 10.1, 8.9, 9), (8.5, 8.6, 8.4, 8.52), (8.7, 9.3, 8.55, 9.2)], 0, "third low 8.55 not above h2 = 8.6"),
    ],
    # :309-342
    "cdladvanceblock": [
        ([(10, 11.1, 9.9, 11), (10.5, 11.9, 10.4, 11.8), (11.5, 12.3, 11.4, 12.2)], -100, "three bulls, third body 0.7 < second 1.3"),
        ([(10, 11.1, 9.9, 11), (10.5, 11.9, 10.4, 11.8), (11.5, 13.1, 11.4, 13.0)], 0, "third body 1.5 not shrinking"),
    ],
    # :345-370
    "cdlbelthold": [
        ([(10, 11.05, 9.99, 11)], 100, "long bull, lower shadow 0.01 < 0.1"),
        ([(11, 11.01, 9.9, 10)], -100, "long bear, upper shadow 0.01 < 0.1"),
        ([(10, 11.05, 9.8, 11)], 0, "lower shadow 0.2"),
    ],
    # :373-411
    "cdlbreakaway": [
        ([(20, 20.1, 18.4, 18.5), (18, 18.1, 17.4, 17.5), (17.4, 17.5, 16.9, 17.0), (17, 17.2, 16.8, 17.1), (17, 18.3, 16.9, 18.2)], 100, "bear long, gap-down bear, lower close, ..., bull closing inside the gap"),
        ([(10, 11.6, 9.9, 11.5), (12, 12.6, 11.9, 12.5), (12.5, 13.1, 12.4, 13.0), (13, 13.2, 12.9, 13.1), (13, 13.1, 11.7, 11.8)], -100, "mirror"),
        ([(20, 20.1, 18.4, 18.5), (18, 18.1, 17.4, 17.5), (17.4, 17.5, 16.9, 17.0), (17, 17.2, 16.8, 17.1), (17, 18.7, 16.9, 18.6)], 0, "close 18.6 above c1 = 18.5"),
    ],
    # :414-439
    "cdlclosingmarubozu": [
        ([(10, 11.01, 9.5, 11)], 100, "long bull closing at the high"),
        ([(11, 11.5, 9.99, 10)], -100, "long bear closing at the low"),
        ([(10, 11.2, 9.5, 11)], 0, "upper shadow 0.2"),
    ],
    # :442-484
    "cdlconcealbabyswall": [
        ([(20, 20, 18, 18), (17.5, 17.5, 16, 16), (15.8, 16.5, 15.4, 15.5), (17, 17.1, 14.9, 15)], 100, "two bear marubozus, bear with high into body 2, engulfing long bear"),
        ([(20, 20, 18, 18), (17.5, 17.5, 16, 16), (15.8, 16.5, 15.4, 15.5), (16.4, 16.45, 14.9, 15)], 0, "fourth opens 16.4 below h3 = 16.5"),
    ],
    # :487-516
    "cdlcounterattack": [
        ([(12, 12.1, 9.9, 10), (8.5, 10.1, 8.4, 10.05)], 100, "long bear, long bull, closes 0.05 apart"),
        ([(10, 12.1, 9.9, 12), (13.5, 13.6, 11.9, 11.95)], -100, "mirror"),
        ([(12, 12.1, 9.9, 10), (8.5, 10.6, 8.4, 10.5)], 0, "closes 0.5 apart"),
    ],
    # :519-550 (penetration: Rust default 0.3)
    "cdldarkcloudcover": [
        ([(10, 12.1, 9.9, 12), (12.5, 12.6, 10.7, 10.8)], -100, "close 10.8 < 12 - 0.5*2 (fires at the Rust default 0.3 and the Python default 0.5)"),
        ([(10, 12.1, 9.9, 12), (12.5, 12.6, 11.4, 11.5)], 0, "close 11.5 not below 11.4"),
    ],
    # :553-575
    "cdldoji": [
        ([(10, 10.5, 9.5, 10.04)], 100, "body 0.04 <= 0.0501"),
        ([(10, 10.5, 9.5, 10.06)], 0, "body 0.06"),
    ],
    # :578-607
    "cdldojistar": [
        ([(12, 12.1, 9.9, 10), (9.5, 9.7, 9.3, 9.52)], 100, "long bear, doji with mid below c1"),
        ([(10, 12.1, 9.9, 12), (12.5, 12.7, 12.3, 12.52)], -100, "mirror"),
        ([(12, 12.1, 9.9, 10), (9.5, 9.9, 9.3, 9.8)], 0, "second is not a doji"),
    ],
    # :610-632
    "cdldragonflydoji": [
        ([(10, 10.043, 8, 10.04)], 100, "doji, lower shadow 2, upper 0.003 < 0.004"),
        ([(10, 10, 8, 10)], 0, "perfect doji: shadow < 0.1*0 is false (D-7)"),
    ],
    # :635-662
    "cdlengulfing": [
        ([(11, 11.1, 9.9, 10), (9.9, 11.3, 9.8, 11.2)], 100, "bull body engulfs bear body"),
        ([(10, 11.1, 9.9, 11), (11.2, 11.3, 9.8, 9.9)], -100, "mirror"),
        ([(11, 11.1, 9.9, 10), (10, 11.1, 9.9, 11)], 0, "equal bodies: neither strict inequality holds"),
    ],
    # :665-700
    "cdleveningdojistar": [
        ([(10, 12.1, 9.9, 12), (12.5, 12.7, 12.4, 12.52), (12.3, 12.4, 10.9, 11)], -100, "long bull, gapped doji, bear closing below 11.4"),
        ([(10, 12.1, 9.9, 12), (12.5, 12.7, 12.4, 12.52), (12.3, 12.4, 11.4, 11.5)], 0, "close 11.5 not below 11.4"),
    ],
    # :703-736
    "cdleveningstar": [
        ([(10, 12.1, 9.9, 12), (12.5, 12.9, 12.4, 12.8), (12.3, 12.4, 10.9, 11)], -100, "long bull, gapped small body, bear closing below 11.4"),
        ([(10, 12.1, 9.9, 12), (11.9, 12.9, 11.8, 12.8), (12.3, 12.4, 10.9, 11)], 0, "no gap: min(o2, c2) = 11.9 <= c1"),
    ],
    # :739-774
    "cdlgapsidesidewhite": [
        ([(10, 11.1, 9.9, 11), (11.5, 12.1, 11.4, 12.0), (11.52, 12.1, 11.4, 12.0)], 100, "bull, gap-up bull, similar bull"),
        ([(11, 11.1, 9.9, 10), (9, 9.6, 8.9, 9.5), (9.02, 9.6, 8.9, 9.5)], -100, "bear, lower bull, similar bull"),
        ([(10, 11.1, 9.9, 11), (11.5, 12.1, 11.4, 12.0), (11.8, 12.1, 11.4, 12.0)], 0, "third body 0.2 vs 0.5, open 0.3 apart"),
    ],
    # :777-799
    "cdlgravestonedoji": [
        ([(10.04, 12, 9.997, 10)], -100, "doji, upper shadow 1.96, lower 0.003 < 0.004 (bearish, :795)"),
        ([(10.04, 12, 9.9, 10)], 0, "lower shadow 0.1"),
    ],
    # :802-829
    "cdlhammer": [
        ([(11, 11.1, 9.9, 10), (9.5, 9.605, 8.5, 9.6)], 100, "after a bear: small body, lower shadow 1 > 0.2, upper 0.005 < 0.01"),
        ([(10, 11.1, 9.9, 11), (9.5, 9.605, 8.5, 9.6)], 0, "previous candle is a bull"),
    ],
    # :832-859
    "cdlhangingman": [
        ([(10, 11.1, 9.9, 11), (11.5, 11.605, 10.5, 11.6)], -100, "the hammer shape after a bull"),
        ([(11, 11.1, 9.9, 10), (11.5, 11.605, 10.5, 11.6)], 0, "previous candle is a bear"),
    ],
    # :862-893
    "cdlharami": [
        ([(15, 15.2, 9.8, 10), (11, 12.2, 10.8, 12)], 100, "small bull inside a long bear"),
        ([(10, 15.2, 9.8, 15), (14, 14.2, 12.8, 13)], -100, "mirror"),
        ([(15, 15.2, 9.8, 10), (9, 16.2, 8.8, 16)], 0, "second body is not inside"),
    ],
    # :896-926
    "cdlharamicross": [
        ([(15, 15.2, 9.8, 10), (12, 12.3, 11.7, 12.03)], 100, "doji inside a long bear"),
        ([(10, 15.2, 9.8, 15), (12, 12.3, 11.7, 12.03)], -100, "doji inside a long bull"),
        ([(15, 15.2, 9.8, 10), (9.5, 9.8, 9.2, 9.53)], 0, "doji below the body"),
    ],
    # :929-953
    "cdlhighwave": [
        ([(10, 11, 9, 10.2)], 100, "small bull, shadows 0.8 and 1.0 > 0.4"),
        ([(10.2, 11, 9, 10)], -100, "small bear"),
        ([(10, 10.5, 9, 10.2)], 0, "upper shadow 0.3"),
    ],
    # :956-984
    "cdlhikkake": [
        ([(10, 12, 9, 11), (10.5, 11.5, 9.5, 11), (11, 12.6, 10.9, 12.5)], 100, "inside bar, bull close above h1"),
        ([(10, 12, 9, 11), (10.5, 11.5, 9.5, 11), (9.5, 9.6, 8.4, 8.5)], -100, "inside bar, bear close below l1"),
        ([(10, 12, 9, 11), (10.5, 11.5, 9.5, 11), (11, 12.0, 10.9, 11.9)], 0, "close 11.9 not above h1"),
    ],
    # :987-1018
    "cdlhikkakemod": [
        ([(10, 12, 9, 11), (10.5, 11.5, 9.5, 11), (10.6, 11.2, 9.8, 10.9), (11, 12.6, 10.9, 12.5)], 100, "two nested inside bars, bull breakout"),
        ([(10, 12, 9, 11), (10.5, 11.5, 9.5, 11), (10.6, 11.2, 9.8, 10.9), (9.5, 9.6, 8.4, 8.5)], -100, "bear breakout"),
        ([(10, 12, 9, 11), (10.5, 11.5, 9.5, 11), (10.6, 11.6, 9.8, 10.9), (11, 12.6, 10.9, 12.5)], 0, "third bar not inside the second"),
    ],
    # :1021-1045
    "cdlhomingpigeon": [
        ([(12, 12.1, 9.9, 10), (11, 11.1, 10.4, 10.5)], 100, "small bear inside a long bear"),
        ([(12, 12.1, 9.9, 10), (11, 11.1, 9.8, 9.9)], 0, "second close below c1"),
    ],
    # :1048-1080
    "cdlidentical3crows": [
        ([(20, 20.1, 18.4, 18.5), (18.5, 18.6, 16.9, 17), (17, 17.1, 15.4, 15.5)], -100, "three long bears, each opening at the prior close"),
        ([(20, 20.1, 18.4, 18.5), (18.6, 18.7, 16.9, 17), (17, 17.1, 15.4, 15.5)], 0, "second opens 0.1 from c1 (limit 0.016)"),
    ],
    # :1083-1108
    "cdlinneck": [
        ([(12, 12.1, 9.9, 10), (9.5, 10.1, 9.4, 10.05)], -100, "bull from below closing at c1"),
        ([(12, 12.1, 9.9, 10), (9.5, 10.6, 9.4, 10.5)], 0, "close 0.5 above c1"),
    ],
    # :1111-1138
    "cdlinvertedhammer": [
        ([(11, 11.1, 9.9, 10), (9.5, 10.6, 9.495, 9.6)], 100, "after a bear: small body, upper shadow 1, lower 0.005"),
        ([(10, 11.1, 9.9, 11), (9.5, 10.6, 9.495, 9.6)], 0, "previous candle is a bull"),
    ],
    # :1141-1180
    "cdlkicking": [
        ([(10, 10, 9, 9), (10.5, 11.5, 10.5, 11.5)], 100, "bear marubozu, bull marubozu opening above o1"),
        ([(9, 10, 9, 10), (8.5, 8.5, 7.5, 7.5)], -100, "mirror"),
        ([(10, 10, 9, 9), (9.5, 10.5, 9.5, 10.5)], 0, "second opens below o1"),
    ],
    # :1183-1226 (the two kicks exclude each other, so the length rule never changes the sign)
    "cdlkickingbylength": [
        ([(10, 10, 9, 9), (10.5, 11.5, 10.5, 11.5)], 100, "as cdlkicking"),
        ([(9, 10, 9, 10), (8.5, 8.5, 7.5, 7.5)], -100, "as cdlkicking"),
        ([(10, 10, 9, 9), (10.5, 12.0, 10.5, 12.0)], 100, "longer second body"),
        ([(10, 10.2, 9, 9), (10.5, 11.5, 10.5, 11.5)], 0, "first has an upper shadow 0.2"),
    ],
    # :1229-1264
    "cdlladderbottom": [
        ([(20, 20.1, 18.4, 18.5), (18.6, 18.7, 17.4, 17.5), (17.6, 17.7, 16.4, 16.5), (16.4, 17.0, 16.1, 16.2), (16.6, 17.6, 16.5, 17.5)], 100, "three lower bears, bear with upper shadow 0.6 > 0.4, bull opening above o4"),
        ([(20, 20.1, 18.4, 18.5), (18.6, 18.7, 17.4, 17.5), (17.6, 17.7, 16.4, 16.5), (16.4, 17.0, 16.1, 16.2), (16.3, 17.6, 16.2, 17.5)], 0, "fifth opens below o4"),
    ],
    # :1267-1289
    "cdllongleggeddoji": [
        ([(10, 11, 9, 10.04)], 100, "doji with two long shadows"),
        ([(10, 10.05, 9, 10.04)], 0, "upper shadow 0.01"),
    ],
    # :1292-1318
    "cdllongline": [
        ([(10, 11.2, 9.8, 11)], 100, "long bull, shadows 0.2 < 0.5"),
        ([(11, 11.2, 9.8, 10)], -100, "long bear"),
        ([(10, 11.6, 9.8, 11)], 0, "upper shadow 0.6"),
    ],
    # :1321-1346
    "cdlmarubozu": [
        ([(10, 12, 10, 12)], 100, "no shadows"),
        ([(12, 12, 10, 10)], -100, "no shadows"),
        ([(10, 13, 9, 12)], 0, "shadows 1.0"),
    ],
    # :1349-1373
    "cdlmatchinglow": [
        ([(12, 12.1, 9.9, 10), (10.8, 10.9, 9.95, 10.005)], 100, "two bears closing 0.005 apart (limit 0.0104)"),
        ([(12, 12.1, 9.9, 10), (10.8, 10.9, 9.95, 10.1)], 0, "closes 0.1 apart"),
    ],
    # :1376-1413
    "cdlmathold": [
        ([(10, 11.6, 9.9, 11.5), (11.8, 11.9, 11.3, 11.4), (11.4, 11.5, 11.0, 11.1), (11.1, 11.3, 10.7, 10.9), (11, 12.1, 10.9, 12)], 100, "long bull, three small bodies holding above o1, bull closing above c1"),
        ([(10, 11.6, 9.9, 11.5), (11.8, 11.9, 11.3, 11.4), (11.4, 11.5, 11.0, 11.1), (11.1, 11.3, 9.9, 10.9), (11, 12.1, 10.9, 12)], 0, "fourth low 9.9 below o1"),
    ],
    # :1416-1451
    "cdlmorningdojistar": [
        ([(12, 12.1, 9.9, 10), (9.5, 9.6, 9.4, 9.52), (9.7, 11.1, 9.6, 11)], 100, "long bear, gapped doji, bull closing above 10.6"),
        ([(12, 12.1, 9.9, 10), (9.5, 9.6, 9.4, 9.52), (9.7, 10.6, 9.6, 10.5)], 0, "close 10.5 not above 10.6"),
    ],
    # :1454-1487
    "cdlmorningstar": [
        ([(12, 12.1, 9.9, 10), (9.5, 9.6, 9.1, 9.2), (9.7, 11.1, 9.6, 11)], 100, "long bear, gapped small body, bull closing above 10.6"),
        ([(12, 12.1, 9.9, 10), (9.5, 9.6, 9.1, 9.2), (9.7, 10.6, 9.6, 10.5)], 0, "close 10.5"),
    ],
    # :1490-1516
    "cdlonneck": [
        ([(12, 12.1, 9.8, 10), (9.3, 9.85, 9.2, 9.82)], -100, "bull closing at the prior low 9.8"),
        ([(12, 12.1, 9.8, 10), (9.3, 10.3, 9.2, 10.2)], 0, "close 0.4 above l1"),
    ],
    # :1519-1550
    "cdlpiercing": [
        ([(12, 12.1, 9.9, 10), (9.5, 11.3, 9.4, 11.2)], 100, "close 11.2 > 10 + 0.5*2, below o1 (fires at 0.3 and 0.5)"),
        ([(12, 12.1, 9.9, 10), (9.5, 10.6, 9.4, 10.5)], 0, "close 10.5 not above 10.6"),
    ],
    # :1553-1578
    "cdlrickshawman": [
        ([(10, 11, 9.04, 10.04)], 100, "doji, both shadows 0.96"),
        ([(10, 11.5, 9.04, 10.04)], 0, "shadows 1.46 and 0.96"),
    ],
    # :1581-1644
    "cdlrisefall3methods": [
        ([(10, 11.6, 9.9, 11.5), (11.2, 11.3, 10.8, 10.9), (10.9, 11.0, 10.5, 10.6), (10.6, 10.8, 10.3, 10.4), (10.5, 12.1, 10.4, 12)], 100, "long bull, three small bodies inside its range, long bull to a new close"),
        ([(11.5, 11.6, 9.9, 10), (10.3, 10.7, 10.2, 10.6), (10.6, 11.0, 10.5, 10.9), (10.9, 11.3, 10.8, 11.2), (11, 11.1, 9.4, 9.5)], -100, "mirror"),
        ([(10, 11.6, 9.9, 11.5), (11.2, 11.3, 10.8, 10.9), (10.9, 11.7, 10.5, 10.6), (10.6, 10.8, 10.3, 10.4), (10.5, 12.1, 10.4, 12)], 0, "third high 11.7 above h1"),
    ],
    # :1647-1676
    "cdlseparatinglines": [
        ([(10, 10.1, 8.9, 9), (10, 11, 10, 11)], 100, "long bear, long bull from the same open"),
        ([(10, 11.1, 9.9, 11), (10, 10, 9, 9)], -100, "mirror"),
        ([(10, 10.1, 8.9, 9), (10.1, 11, 10, 11)], 0, "opens 0.1 apart (limit 0.0105)"),
    ],
    # :1679-1706
    "cdlshootingstar": [
        ([(10, 11.1, 9.9, 11), (11.5, 12.6, 11.495, 11.6)], -100, "the inverted-hammer shape after a bull"),
        ([(11, 11.1, 9.9, 10), (11.5, 12.6, 11.495, 11.6)], 0, "previous candle is a bear"),
    ],
    # :1709-1735
    "cdlshortline": [
        ([(10, 10.6, 9.9, 10.5)], 100, "small bull, shadows 0.1 < 0.25"),
        ([(10.5, 10.6, 9.9, 10)], -100, "small bear"),
        ([(10, 10.9, 9.9, 10.5)], 0, "upper shadow 0.4"),
    ],
    # :1738-1763
    "cdlspinningtop": [
        ([(10, 10.8, 9.4, 10.2)], 100, "small bull, shadows 0.6 > body 0.2"),
        ([(10.2, 10.8, 9.4, 10)], -100, "small bear"),
        ([(10, 10.3, 9.4, 10.2)], 0, "upper shadow 0.1"),
    ],
    # :1766-1794
    "cdlstalledpattern": [
        ([(10, 11.1, 9.9, 11), (11, 12.6, 10.9, 12.5), (12.4, 12.8, 12.3, 12.7)], -100, "two long bulls, small bull opening near c2"),
        ([(10, 11.1, 9.9, 11), (11, 12.6, 10.9, 12.5), (12.6, 13.0, 12.5, 12.9)], 0, "third opens above c2"),
    ],
    # :1797-1828
    "cdlsticksandwich": [
        ([(12, 12.1, 9.9, 10), (10.5, 11.6, 10.4, 11.5), (12, 12.1, 9.95, 10.005)], 100, "bear, gapped bull, bear closing at c1"),
        ([(12, 12.1, 9.9, 10), (10.5, 11.6, 10.4, 11.5), (12, 12.1, 9.95, 10.2)], 0, "third close 0.2 from c1"),
    ],
    # :1831-1853
    "cdltakuri": [
        ([(10, 10.043, 8, 10.04)], 100, "doji, lower shadow 2 > 0.12, upper 0.003"),
        ([(10, 10.1, 8, 10.04)], 0, "upper shadow 0.06"),
    ],
    # :1856-1891
    "cdltasukigap": [
        ([(10, 11.1, 9.9, 11), (11.5, 12.6, 11.4, 12.5), (12, 12.1, 10.4, 10.5)], 100, "two gapped bulls, bear from inside body 2 closing inside body 1"),
        ([(11, 11.1, 9.9, 10), (9.5, 9.6, 8.4, 8.5), (9, 10.6, 8.9, 10.5)], -100, "mirror"),
        ([(10, 11.1, 9.9, 11), (11.5, 12.6, 11.4, 12.5), (12, 12.1, 11.1, 11.2)], 0, "close 11.2 above c1"),
    ],
    # :1894-1919
    "cdlthrusting": [
        ([(12, 12.1, 9.9, 10), (9.5, 10.9, 9.4, 10.8)], -100, "bull closing between c1 and the body midpoint 11"),
        ([(12, 12.1, 9.9, 10), (9.5, 11.3, 9.4, 11.2)], 0, "close above the midpoint"),
    ],
    # :1922-1961
    "cdltristar": [
        ([(10, 10.2, 9.8, 10.02), (9.5, 9.7, 9.3, 9.52), (9.8, 10, 9.6, 9.82)], 100, "three dojis, the middle one lowest"),
        ([(10, 10.2, 9.8, 10.02), (10.5, 10.7, 10.3, 10.52), (10.2, 10.4, 10, 10.22)], -100, "the middle one highest"),
        ([(10, 10.2, 9.8, 10.02), (9.5, 9.8, 9.3, 9.7), (9.8, 10, 9.6, 9.82)], 0, "middle is not a doji"),
    ],
    # :1964-1994
    "cdlunique3river": [
        ([(12, 12.1, 9.9, 10), (11, 11.1, 9.5, 10.5), (10, 10.4, 9.9, 10.3)], 100, "long bear, harami bear with a new low, small bull below c2"),
        ([(12, 12.1, 9.9, 10), (11, 11.1, 9.5, 10.5), (10, 10.7, 9.9, 10.6)], 0, "third close above c2"),
    ],
    # :1997-2024
    "cdlupsidegap2crows": [
        ([(10, 11.1, 9.9, 11), (12, 12.1, 11.4, 11.5), (12.3, 12.4, 11.1, 11.2)], -100, "long bull, gapped bear, larger bear still above c1"),
        ([(10, 11.1, 9.9, 11), (12, 12.1, 11.4, 11.5), (12.3, 12.4, 10.8, 10.9)], 0, "third closes the gap"),
    ],
    # :2027-2062
    "cdlxsidegap3methods": [
        ([(10, 11.1, 9.9, 11), (11.5, 12.6, 11.4, 12.5), (12, 12.1, 10.4, 10.5)], 100, "two gapped bulls, bear filling the gap"),
        ([(11, 11.1, 9.9, 10), (9.5, 9.6, 8.4, 8.5), (9, 10.6, 8.9, 10.5)], -100, "mirror"),
        ([(10, 11.1, 9.9, 11), (11.5, 12.6, 11.4, 12.5), (12, 12.1, 11.1, 11.2)], 0, "close 11.2 above c1"),
    ],
}

# penetration used with the entries above where the function reads one (pattern.rs:529-532 etc.: Rust default 0.3)
KAT_PENETRATION = 0.3
# (name, candles, penetration, expected): the same shapes at another penetration
PEN_CASES = [
    ("cdldarkcloudcover", [(10, 12.1, 9.9, 12), (12.5, 12.6, 10.9, 11)], 0.5, 0),       # 11 < 12 - 1.0 is false
    ("cdlpiercing", [(12, 12.1, 9.9, 10), (9.5, 11.1, 9.4, 11)], 0.5, 0),               # 11 > 11 is false
    ("cdleveningdojistar", [(10, 12.1, 9.9, 12), (12.5, 12.7, 12.4, 12.52), (12.3, 12.4, 10.9, 11)], 0.6, 0),
    ("cdleveningstar", [(10, 12.1, 9.9, 12), (12.5, 12.9, 12.4, 12.8), (12.3, 12.4, 10.9, 11)], 0.6, 0),
    ("cdlmorningdojistar", [(12, 12.1, 9.9, 10), (9.5, 9.6, 9.4, 9.52), (9.7, 11.1, 9.6, 11)], 0.6, 0),
    ("cdlmorningstar", [(12, 12.1, 9.9, 10), (9.5, 9.6, 9.1, 9.2), (9.7, 11.1, 9.6, 11)], 0.6, 0),
    # functions whose Python wrapper passes a penetration the Rust ignores (pattern.rs:268, 1376, 1894): unchanged
    ("cdlabandonedbaby", [(10, 10.1, 8.9, 9), (8.5, 8.6, 8.4, 8.52), (8.7, 9.3, 8.65, 9.2)], 0.9, 100),
    ("cdlthrusting", [(12, 12.1, 9.9, 10), (9.5, 10.9, 9.4, 10.8)], 0.9, -100),
]

UNSAT = ("cdl2crows",)


def positives():
    """[(name, candles, value)] of every firing case"""
    return [(nm, cs, v) for nm, cases in KATS.items() for cs, v, _ in cases if v != 0]


def kat_series(T: int, start: int = 0):
    """One OHLC series of T rows: the firing sequences above from number `start` on laid end to end (cyclically, each followed by a flat separator
    candle), as four float64 arrays.  Returns (open, high, low, close, marks) with marks = [(row, name, value)] for the last row of
    every complete sequence: the recogniser `name` must output `value` there whatever precedes the sequence."""
    import numpy as np
    rows, marks = [], []
    pos = positives()
    k = start
    while len(rows) < T:
        nm, cs, v = pos[k % len(pos)]
        k += 1
        if len(rows) + len(cs) > T:
            break
        rows.extend(cs)
        marks.append((len(rows) - 1, nm, v))
        rows.append((10.0, 10.0, 10.0, 10.0))
    while len(rows) < T:
        rows.append((10.0, 10.0, 10.0, 10.0))
    a = np.asarray(rows[:T], dtype=np.float64)
    return a[:, 0].copy(), a[:, 1].copy(), a[:, 2].copy(), a[:, 3].copy(), [m for m in marks if m[0] < T]
