"""The score kernel decides `sqrt(s) < sqrt(sBest)` (both correctly rounded) without taking roots when
s < fl(sBest * (1 - 2^-21)).  This checks that implication on the CPU, adversarially close to the band edge."""
import numpy as np

K = np.float32(0.999999523162841796875)  # 1 - 2^-21, PF_GUARD_K


def test_guard_band_implies_strict_root_order():
    rng = np.random.RandomState(0)
    for scale in (1e-20, 1e-8, 1e-3, 1.0, 37.5, 1e6, 1e20):
        sb = (rng.uniform(0.5, 2.0, 400000) * scale).astype(np.float32)
        guard = (sb * K).astype(np.float32)
        # candidates just below the guard: the nearest floats under it, where a wrong decision would show first
        for steps in (1, 2, 3, 8, 64):
            s = guard.copy()
            for _ in range(steps):
                s = np.nextafter(s, np.float32(0), dtype=np.float32)
            ok = s < guard
            assert ok.all()
            assert (np.sqrt(s) < np.sqrt(sb)).all(), (scale, steps)
    # and the band itself really contains pairs whose rounded roots tie (so the exact path is needed there)
    sb = np.float32(2.0) + np.arange(1, 2000, dtype=np.float32) * np.float32(2.0 ** -22)
    s = np.nextafter(sb, np.float32(0), dtype=np.float32)
    assert (np.sqrt(s) == np.sqrt(sb)).any()


def test_guard_constant_is_exact():
    assert float(K) == 1.0 - 2.0 ** -21
