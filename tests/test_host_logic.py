"""CPU tests of the host-side mirror's own logic (no device calls): monomial tables and
snapshot selection agree with the oracle's independent restatement."""
import numpy as np

from koopman_realizations_amd.ksysid import poly_exponent_table
from oracle import koopman_oracle as ko


def test_exponent_tables_agree_with_oracle():
    for nv, d in [(6, 2), (6, 3), (9, 3), (1, 13), (2, 4), (3, 5)]:
        assert (poly_exponent_table(nv, d) == ko.poly_exponents(nv, d)).all()
