"""Workload definitions for the configurations BASELINE.json names.

manual_constraints_circuit  config #1: /root/reference/examples/manual-constraints.rs:15-31 — one public input a, one
                            witness b, one row (a - b) * 1 = 0.
random_sparse_circuit       parity case beyond the BASELINE shapes: multi-term rows, 5 public inputs, |K| != |H|, nnz(B) > nnz(A).
synthetic_r1cs              configs #2-#4: the ark-marlin test/bench circuit shape — witnesses a, b, public c = a*b and
                            d = c*b, n - 1 rows a*b = c and one row c*b = d, padded with copies of a so that
                            |H| = |K| = n exactly (instance [1, c, d, 0-pad], n - 4 witnesses).  Built with numpy so that
                            n = 2^20 .. 2^22 take milliseconds to lay out.
"""
import numpy as np

from .marlin import ConstraintSystem, PackedR1cs, R_MODULUS, _to_mont_limbs


def manual_constraints_circuit(a, b):
    cs = ConstraintSystem()
    va = cs.new_input_variable(a)
    vb = cs.new_witness_variable(b)
    cs.enforce_constraint([(1, va), (R_MODULUS - 1, vb)], [(1, cs.one())], [])
    return cs


def synthetic_circuit(n, a, b):
    """Same circuit through the ConstraintSystem builder (small n)."""
    assert n >= 8 and n & (n - 1) == 0
    cs = ConstraintSystem()
    va = cs.new_witness_variable(a)
    vb = cs.new_witness_variable(b)
    c = a * b % R_MODULUS
    d = c * b % R_MODULUS
    vc = cs.new_input_variable(c)
    vd = cs.new_input_variable(d)
    for _ in range(n - 6):
        cs.new_witness_variable(a)
    for _ in range(n - 1):
        cs.enforce_constraint([(1, va)], [(1, vb)], [(1, vc)])
    cs.enforce_constraint([(1, vc)], [(1, vb)], [(1, vd)])
    return cs


def random_sparse_circuit(seed, num_inputs=5, free_witnesses=4, num_constraints=12, repeated_rows=0):
    """Small circuit with multi-term linear combinations, several public inputs, |K| != |H| and nnz(B) > nnz(A): rows
    (1-3 terms) * (2-4 terms) = fresh product witness, shape drawn from random.Random(seed).  The test suite's
    reference model builds the identical system from the same arguments; tests/golden/marlin.json holds its proofs."""
    import random
    rnd = random.Random(seed)
    cs = ConstraintSystem()
    vars_, vals = [cs.one()], [1]
    for _ in range(num_inputs):
        v = rnd.randrange(R_MODULUS)
        vars_.append(cs.new_input_variable(v))
        vals.append(v)
    for _ in range(free_witnesses):
        v = rnd.randrange(R_MODULUS)
        vars_.append(cs.new_witness_variable(v))
        vals.append(v)
    rows = []
    for _ in range(num_constraints):
        def lc(lo, hi):
            terms, total = [], 0
            for _ in range(rnd.randint(lo, hi)):
                k = rnd.randrange(len(vars_))
                coeff = rnd.choice([1, 2, R_MODULUS - 1, rnd.randrange(R_MODULUS)])
                terms.append((coeff, vars_[k]))
                total = (total + coeff * vals[k]) % R_MODULUS
            return terms, total
        a, va = lc(1, 3)
        b, vb = lc(2, 4)
        prod = va * vb % R_MODULUS
        w = cs.new_witness_variable(prod)
        vars_.append(w)
        vals.append(prod)
        cs.enforce_constraint(a, b, [(1, w)])
        rows.append((a, b, [(1, w)]))
    for i in range(repeated_rows):  # more constraints than variables: |H| comes from the row count
        cs.enforce_constraint(*rows[i % len(rows)])
    return cs


def synthetic_r1cs(n, a, b):
    """Vectorised layout of synthetic_circuit(n, a, b) as a PackedR1cs; returns (packed, public_inputs)."""
    assert n >= 8 and n & (n - 1) == 0
    a %= R_MODULUS
    b %= R_MODULUS
    c = a * b % R_MODULUS
    d = c * b % R_MODULUS
    instance = _to_mont_limbs([1, c, d])
    am = _to_mont_limbs([a, b])
    witness = np.empty((n - 4, 4), dtype=np.uint64)
    witness[:] = am[0]
    witness[1] = am[1]
    one = _to_mont_limbs([1])[0]
    ninst = 3
    col_a, col_b, col_c, col_d = ninst + 0, ninst + 1, 1, 2
    rowptr = np.arange(n + 1, dtype=np.uint32)
    val = np.empty((n, 4), dtype=np.uint64)
    val[:] = one

    def mat(cols_main, col_last):
        col = np.full(n, cols_main, dtype=np.uint32)
        col[-1] = col_last
        return rowptr, col, val

    packed = PackedR1cs(instance, witness, mat(col_a, col_c), mat(col_b, col_b), mat(col_c, col_d))
    return packed, [c, d]
