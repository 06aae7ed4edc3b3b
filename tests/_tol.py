"""north-star tolerance for floating-point tables: "embedding L2 within 1e-4 fp32"."""
import numpy as np


def assert_rows_close(got, ref, what, tol=1e-4):
    """every ROW of `got` within tol (L2) of the oracle's row, relative to max(1, |row|) -- rows of the tables are unit
    length or close to it, so this is an absolute 1e-4 per embedding, not a fraction of the whole table's norm.
    Prints the measured deviation.  -> (max row deviation, whole-table relative deviation)"""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    dev = np.linalg.norm(got - ref, axis=1)
    scale = np.maximum(np.linalg.norm(ref, axis=1), 1.0)
    worst = float((dev / scale).max()) if len(dev) else 0.0
    rel = float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30))
    print("%s: max row L2 deviation %.3g (tolerance %.0e), whole table relative %.3g" % (what, worst, tol, rel))
    assert worst <= tol, "%s: row deviation %.3g > %.0e" % (what, worst, tol)
    return worst, rel
