"""The 64-entry table of the spacing logarithm of GPF_RESAMPLE_MULTINOMIAL_SORTED (gpf_math.hpp spacing_of / gpf_oracle.c o_spacing): for the
mantissa interval i, INV[i] = fl(1 / (1 + (i + 1/2) / 64)) and LN[i] = fl(-ln(INV[i])) -- ln m = LN[i] + log1p(m INV[i] - 1).  Printed as C
hexadecimal floating literals; the two copies in csrc/gpf_math.hpp and oracle/gpf_oracle_math.h are pasted from this output.  Also checks the
accuracy of the whole function against math.log (needs nothing beyond the standard library: decimal for the table's second column)."""
import math
from decimal import Decimal, getcontext

getcontext().prec = 60
inv = [1.0 / (1.0 + (i + 0.5) / 64.0) for i in range(64)]
ln = [float(-(Decimal(v).ln())) for v in inv]


def emit(name, vals):
    print(f"static const double {name}[64] = {{")
    for r in range(0, 64, 4):
        print("    " + ", ".join(float.hex(v) for v in vals[r:r + 4]) + ",")
    print("};")


def neglog(k):
    """-ln((k + 1/2) 2^-52) the way the spec computes it (Python floats are IEEE doubles, one rounding per operation)"""
    x = float(2 * k + 1)
    m, e = math.frexp(x)            # x = m 2^e, m in [0.5, 1)
    m *= 2.0; e -= 1                # m in [1, 2)
    i = int((m - 1.0) * 64.0)
    r = m * inv[i] - 1.0
    p = r * (1.0 - r * (0.5 - r * (1.0 / 3.0 - r * 0.25)))
    v = float(53 - e) * math.log(2.0) - (ln[i] + p)
    return v if v > 0.0 else 0.0


if __name__ == "__main__":
    emit("SP_INV", inv)
    emit("SP_LN", ln)
    import random
    random.seed(1)
    worst = 0.0
    ks = [0, 1, 2, (1 << 52) - 1, (1 << 52) - 2, 1 << 51, (1 << 51) - 1] + [random.getrandbits(52) for _ in range(200000)] + \
         [random.getrandbits(b) for b in range(1, 52) for _ in range(200)]
    for k in ks:
        want = -math.log((k + 0.5) * 2.0 ** -52)
        worst = max(worst, abs(neglog(k) - want))
    print("/* max |error| over", len(ks), "arguments:", worst, "*/")
