"""No-arithmetic stand-in for the un-vendored `opt_einsum` dependency (requirements: >=3.3.0).

Used ONLY by tests/golden/make_golden.py inside the build container to import the Python
reference and capture golden vectors.  It chooses a fixed left-to-right contraction order;
any valid order yields the same mathematics.  Never shipped, never
imported by the product.
"""


class DynamicProgramming:  # opaque optimiser handle
    def __init__(self, **kwargs):
        self.kwargs = kwargs


def contract_path(expr, *operands, optimize=None):
    n = len(operands)
    return [(0, 1)] * (n - 1), None
